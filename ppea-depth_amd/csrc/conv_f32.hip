// fp32 dense convolutions and linear layers (groups = 1) as implicit GEMMs on the fp32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: true fp32 products, fp32 accumulation).
//
// Replaces, for the fp32 (parity / BASELINE config 1) step, every library convolution and GEMM the reference's modules
// call: the frozen 1x1 convs of RepLKBlock / ConvFFN and the adapters' 1x1 / 3x3 convs and Linears
// (networks/replknet_adapter.py:20-109, 264-326), the decoder's 3x3 convs and the Stage-2 ConvTranspose2d
// (networks/depth_decoder_v2.py:172-245), the pose ResNet-18 / PoseDecoder convs (networks/resnet_encoder.py:367-409,
// networks/pose_decoder.py:27-31), reduce_conv and stem[0].  The bf16 step has its own layout-specialised kernels
// (pwconv.hip, conv_nhwc.hip, conv_wgrad.hip, conv_image.hip); this file is one family for every shape, stride and memory
// format (NCHW, channels_last, [tokens][features] matrices): operands are addressed through element strides.
//
//   forward        y[n][co][oy][ox]  = bias[co] + sum_{ci,r,s} w[co][ci][r][s] x[n][ci][oy st - pad + r][ox st - pad + s]
//   data gradient  dx[n][ci][iy][ix] = sum_{co,r,s} w[co][ci][r][s] dy[n][co][(iy + pad - r) / st][(ix + pad - s) / st]
//   weight grad.   dw[co][ci][r][s]  = sum_{n,oy,ox} dy[n][co][oy][ox] x[n][ci][oy st - pad + r][ox st - pad + s]
//
// GEMM view: rows = output channels (data gradient: input channels), columns = pixels, contraction = (channel, tap); the
// weight gradient contracts over pixels, split over the grid's z with partial slabs in a caller-owned workspace summed in
// a fixed order (no float atomics: bitwise reproducible).  A workgroup of four waves owns a 64 x 64 tile, a wave a
// 32 x 32 quadrant; K is walked in chunks of 16 staged through LDS k-major, so that a lane's MFMA operand
// (index = lane % 32, k = lane / 32) is one conflict-free 4-byte LDS read.  The im2col matrix is never built: the pixel
// operand is gathered with bounds checks (zero padding) as it is staged.  Correctness first -- this path is timed by
// nobody but `bench.py --dtype f32`; the benchmarked arithmetic is bf16.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TM = 64, TN = 64, KC = 16, PITCH = TN + 4, TPB = 256;

struct CvArgs {
    const void* a;         // weights (fwd / dgrad) or dy (wgrad): float, or bf16 in the T = uint16_t instantiation
    const void* b;         // x (fwd, wgrad) or dy (dgrad)
    const float* bias;     // always fp32
    void* y;               // T (wgrad: fp32 slabs)
    long bs[4];            // element strides (n, c, h, w) of the gathered operand
    long as[4];            // wgrad: element strides of dy
    long ys[4];            // element strides of the result (fwd: y, dgrad: dx)
    int N, Cin, H, W, Cout, R, S, stride, pad, Ho, Wo;
    int M, NC, K;          // GEMM sizes
    int kper;              // wgrad: contraction elements per split
};

__device__ __forceinline__ void mfma_chunk(const float (*As)[PITCH], const float (*Bs)[PITCH], int wm, int wn, int lane,
                                           f32x16& acc) {
    const int idx = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int k2 = 0; k2 < KC; k2 += 2) {
        const float a = As[k2 + kh][wm * 32 + idx];
        const float b = Bs[k2 + kh][wn * 32 + idx];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
}

// MODE 0: forward, 1: data gradient.  T: element type of weights, activations and result (float, or uint16_t = bf16: the
// operands are widened as they are staged, products and sums stay fp32, the result is rounded once).
template <int MODE, typename T>
__global__ __launch_bounds__(TPB) void conv_f32_kernel(CvArgs p) {
    const T* pa = (const T*)p.a;
    const T* pb = (const T*)p.b;
    T* py_out = (T*)p.y;
    __shared__ float As[KC][PITCH];
    __shared__ float Bs[KC][PITCH];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const int RS = p.R * p.S;
    // this thread's column of the pixel operand (fixed for the whole tile)
    const int pcol = n0 + (t & 63);
    const int PH = MODE == 0 ? p.Ho : p.H, PW = MODE == 0 ? p.Wo : p.W;
    const bool pok = pcol < p.NC;
    int pn = 0, py = 0, px = 0;
    if (pok) {
        pn = pcol / (PH * PW);
        const int rem = pcol - pn * PH * PW;
        py = rem / PW;
        px = rem - py * PW;
    }
    const int akk = t & 15, am = t >> 4;            // A tile: k = akk, rows am + 16 i
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k0 = 0; k0 < p.K; k0 += KC) {
        // ---- A: weights ----
        {
            const int k = k0 + akk;
            int c2 = 0, r = 0, s = 0;
            if (MODE == 1 && k < p.K) {
                c2 = k / RS;
                const int rs = k - c2 * RS;
                r = rs / p.S;
                s = rs - r * p.S;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + am + 16 * i;
                float v = 0.f;
                if (m < p.M && k < p.K)
                    v = ld_f32<T>(MODE == 0 ? pa + ((long)m * p.K + k) : pa + ((((long)c2 * p.Cin + m) * p.R + r) * p.S + s));
                As[akk][am + 16 * i] = v;
            }
        }
        // ---- B: gathered pixels ----
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = (t >> 6) + 4 * i;
            const int k = k0 + kk;
            float v = 0.f;
            if (pok && k < p.K) {
                const int c2 = k / RS;
                const int rs = k - c2 * RS;
                const int r = rs / p.S, s = rs - r * p.S;
                if (MODE == 0) {
                    const int iy = py * p.stride - p.pad + r, ix = px * p.stride - p.pad + s;
                    if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                        v = ld_f32<T>(pb + (pn * p.bs[0] + c2 * p.bs[1] + iy * p.bs[2] + ix * p.bs[3]));
                } else {
                    const int ty = py + p.pad - r, tx = px + p.pad - s;
                    if (ty >= 0 && tx >= 0) {
                        const int oy = ty / p.stride, ox = tx / p.stride;
                        if (oy * p.stride == ty && ox * p.stride == tx && oy < p.Ho && ox < p.Wo)
                            v = ld_f32<T>(pb + (pn * p.bs[0] + c2 * p.bs[1] + oy * p.bs[2] + ox * p.bs[3]));
                    }
                }
            }
            Bs[kk][t & 63] = v;
        }
        __syncthreads();
        mfma_chunk(As, Bs, wm, wn, lane, acc);
        __syncthreads();
    }
    // ---- epilogue: D[row][col], row = 8 (i / 4) + 4 (lane / 32) + i % 4, col = lane % 32 ----
    const int col = n0 + wn * 32 + (lane & 31);
    if (col < p.NC) {
        const int n = col / (PH * PW);
        const int rem = col - n * PH * PW;
        const int y = rem / PW, x = rem - y * PW;
        const long base = n * p.ys[0] + y * p.ys[2] + x * p.ys[3];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m0 + wm * 32 + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
            if (m < p.M) st_f32<T>(py_out + (base + m * p.ys[1]), acc[i] + (p.bias != nullptr ? p.bias[m] : 0.f));
        }
    }
}

// weight gradient: rows = output channels, columns = (ci, r, s), contraction = pixels [z kper, (z + 1) kper); fp32 slabs
template <typename T>
__global__ __launch_bounds__(TPB) void conv_f32_wgrad_kernel(CvArgs p) {
    const T* pa = (const T*)p.a;
    const T* pb = (const T*)p.b;
    __shared__ float As[KC][PITCH];
    __shared__ float Bs[KC][PITCH];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const int RS = p.R * p.S, HW = p.Ho * p.Wo;
    const int kbeg = blockIdx.z * p.kper, kend = min(p.K, kbeg + p.kper);
    const int jcol = n0 + (t & 63);
    const bool jok = jcol < p.NC;
    int jc = 0, jr = 0, js = 0;
    if (jok) {
        jc = jcol / RS;
        const int rs = jcol - jc * RS;
        jr = rs / p.S;
        js = rs - jr * p.S;
    }
    const int akk = t & 15, am = t >> 4;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k0 = kbeg; k0 < kend; k0 += KC) {
        {
            const int k = k0 + akk;
            long off = 0;
            const bool kok = k < kend;
            if (kok) {
                const int n = k / HW;
                const int rem = k - n * HW;
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                off = n * p.as[0] + oy * p.as[2] + ox * p.as[3];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + am + 16 * i;
                As[akk][am + 16 * i] = (kok && m < p.M) ? ld_f32<T>(pa + (off + m * p.as[1])) : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = (t >> 6) + 4 * i;
            const int k = k0 + kk;
            float v = 0.f;
            if (jok && k < kend) {
                const int n = k / HW;
                const int rem = k - n * HW;
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                const int iy = oy * p.stride - p.pad + jr, ix = ox * p.stride - p.pad + js;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                    v = ld_f32<T>(pb + (n * p.bs[0] + jc * p.bs[1] + iy * p.bs[2] + ix * p.bs[3]));
            }
            Bs[kk][t & 63] = v;
        }
        __syncthreads();
        mfma_chunk(As, Bs, wm, wn, lane, acc);
        __syncthreads();
    }
    float* out = (float*)p.y + (long)blockIdx.z * p.M * p.NC;
    const int col = n0 + wn * 32 + (lane & 31);
    if (col < p.NC) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m0 + wm * 32 + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
            if (m < p.M) out[(long)m * p.NC + col] = acc[i];
        }
    }
}

__global__ void conv_f32_slab_sum(const float* __restrict__ ws, float* __restrict__ dw, long n, int slabs) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = ws[i];
    for (int z = 1; z < slabs; ++z) s += ws[(long)z * n + i];          // fixed order
    dw[i] = s;
}

int wgrad_splits(int M, int NC, long K) {
    const long tiles = (long)((M + TM - 1) / TM) * ((NC + TN - 1) / TN);
    long splits = (1024 + tiles - 1) / tiles;                           // ~4 workgroups per CU
    const long most = (K + 511) / 512;                                  // >= 512 pixels per split
    if (splits > most) splits = most;
    if (splits > 256) splits = 256;
    if (splits < 1) splits = 1;
    return (int)splits;
}

bool fits_int(long v) { return v > 0 && v < (1L << 31); }

template <typename T>
int fwd_impl(const void* x, const long* xs, const void* w, const float* bias, void* y, const long* ys, int N, int Cin, int H,
             int W, int Cout, int R, int S, int stride, int pad, void* stream) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0) return PPEA_ERR_ARG;
    const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    if (Ho <= 0 || Wo <= 0 || !fits_int((long)N * Ho * Wo) || !fits_int((long)Cin * R * S)) return PPEA_ERR_UNSUPPORTED;
    CvArgs p{};
    p.a = w; p.b = x; p.bias = bias; p.y = y;
    for (int i = 0; i < 4; ++i) { p.bs[i] = xs[i]; p.ys[i] = ys[i]; }
    p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.R = R; p.S = S; p.stride = stride; p.pad = pad; p.Ho = Ho; p.Wo = Wo;
    p.M = Cout; p.NC = N * Ho * Wo; p.K = Cin * R * S;
    dim3 grid((p.NC + TN - 1) / TN, (p.M + TM - 1) / TM);
    if (grid.y > 65535) return PPEA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((conv_f32_kernel<0, T>), grid, dim3(TPB), 0, (hipStream_t)stream, p);
    return launch_status();
}

template <typename T>
int dgrad_impl(const void* dy, const long* dys, const void* w, void* dx, const long* dxs, int N, int Cin, int H, int W, int Cout,
               int R, int S, int stride, int pad, int Ho, int Wo, void* stream) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0 || Ho <= 0 || Wo <= 0) return PPEA_ERR_ARG;
    if (!fits_int((long)N * H * W) || !fits_int((long)Cout * R * S)) return PPEA_ERR_UNSUPPORTED;
    CvArgs p{};
    p.a = w; p.b = dy; p.bias = nullptr; p.y = dx;
    for (int i = 0; i < 4; ++i) { p.bs[i] = dys[i]; p.ys[i] = dxs[i]; }
    p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.R = R; p.S = S; p.stride = stride; p.pad = pad; p.Ho = Ho; p.Wo = Wo;
    p.M = Cin; p.NC = N * H * W; p.K = Cout * R * S;
    dim3 grid((p.NC + TN - 1) / TN, (p.M + TM - 1) / TM);
    if (grid.y > 65535) return PPEA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((conv_f32_kernel<1, T>), grid, dim3(TPB), 0, (hipStream_t)stream, p);
    return launch_status();
}

template <typename T>
int wgrad_impl(const void* x, const long* xs, const void* dy, const long* dys, float* dw, float* workspace, int N, int Cin, int H,
               int W, int Cout, int R, int S, int stride, int pad, int Ho, int Wo, void* stream) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0 || Ho <= 0 || Wo <= 0) return PPEA_ERR_ARG;
    if (!fits_int((long)N * Ho * Wo) || !fits_int((long)Cin * R * S)) return PPEA_ERR_UNSUPPORTED;
    CvArgs p{};
    p.a = dy; p.b = x; p.bias = nullptr;
    for (int i = 0; i < 4; ++i) { p.bs[i] = xs[i]; p.as[i] = dys[i]; }
    p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.R = R; p.S = S; p.stride = stride; p.pad = pad; p.Ho = Ho; p.Wo = Wo;
    p.M = Cout; p.NC = Cin * R * S; p.K = N * Ho * Wo;
    const int splits = wgrad_splits(p.M, p.NC, p.K);
    if (splits > 1 && workspace == nullptr) return PPEA_ERR_ARG;
    p.kper = ((p.K + splits - 1) / splits + KC - 1) / KC * KC;
    p.y = splits > 1 ? workspace : dw;
    dim3 grid((p.NC + TN - 1) / TN, (p.M + TM - 1) / TM, splits);
    if (grid.y > 65535) return PPEA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((conv_f32_wgrad_kernel<T>), grid, dim3(TPB), 0, (hipStream_t)stream, p);
    if (splits > 1) {
        const long n = (long)p.M * p.NC;
        hipLaunchKernelGGL(conv_f32_slab_sum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, workspace,
                           dw, n, splits);
    }
    return launch_status();
}

}  // namespace

extern "C" {

// xs / ys: four element strides (n, c, h, w) of x / y (NCHW, channels_last, or a [tokens][features] matrix viewed as
// [1][features][tokens][1]); w: [Cout][Cin][R][S] contiguous; bias: fp32 [Cout] or NULL.
int ppea_conv2d_f32_fwd(const float* x, const long* xs, const float* w, const float* bias, float* y, const long* ys, int N,
                        int Cin, int H, int W, int Cout, int R, int S, int stride, int pad, void* stream) {
    return fwd_impl<float>(x, xs, w, bias, y, ys, N, Cin, H, W, Cout, R, S, stride, pad, stream);
}
// dx [N][Cin][H][W] (strides dxs) from dy [N][Cout][Ho][Wo] (strides dys): every element of dx is written.
int ppea_conv2d_f32_dgrad(const float* dy, const long* dys, const float* w, float* dx, const long* dxs, int N, int Cin, int H,
                          int W, int Cout, int R, int S, int stride, int pad, int Ho, int Wo, void* stream) {
    return dgrad_impl<float>(dy, dys, w, dx, dxs, N, Cin, H, W, Cout, R, S, stride, pad, Ho, Wo, stream);
}
long ppea_conv2d_f32_wgrad_workspace_bytes(int N, int Cin, int Cout, int R, int S, int Ho, int Wo) {
    const int splits = wgrad_splits(Cout, Cin * R * S, (long)N * Ho * Wo);
    return splits > 1 ? (long)splits * Cout * Cin * R * S * 4 : 0;
}
// dw [Cout][Cin][R][S] contiguous fp32; workspace: ppea_conv2d_f32_wgrad_workspace_bytes(...) bytes (NULL when that is 0).
int ppea_conv2d_f32_wgrad(const float* x, const long* xs, const float* dy, const long* dys, float* dw, float* workspace, int N,
                          int Cin, int H, int W, int Cout, int R, int S, int stride, int pad, int Ho, int Wo, void* stream) {
    return wgrad_impl<float>(x, xs, dy, dys, dw, workspace, N, Cin, H, W, Cout, R, S, stride, pad, Ho, Wo, stream);
}
// The same family for bf16 weights / activations / results (fp32 products and sums on the same fp32 matrix-core
// instruction, one rounding at the store; dw stays fp32): what the bf16 step falls back to for shapes the layout-specialised
// bf16 kernels do not take (maps whose width is not a multiple of 4 pixels, channel counts that are not multiples of 8 /
// 32 -- reduced-size test configurations), so that no shape of either dtype reaches a library convolution.
int ppea_conv2d_bf16_fwd(const void* x, const long* xs, const void* w, const float* bias, void* y, const long* ys, int N,
                         int Cin, int H, int W, int Cout, int R, int S, int stride, int pad, void* stream) {
    return fwd_impl<uint16_t>(x, xs, w, bias, y, ys, N, Cin, H, W, Cout, R, S, stride, pad, stream);
}
int ppea_conv2d_bf16_dgrad(const void* dy, const long* dys, const void* w, void* dx, const long* dxs, int N, int Cin, int H,
                           int W, int Cout, int R, int S, int stride, int pad, int Ho, int Wo, void* stream) {
    return dgrad_impl<uint16_t>(dy, dys, w, dx, dxs, N, Cin, H, W, Cout, R, S, stride, pad, Ho, Wo, stream);
}
int ppea_conv2d_bf16_wgrad(const void* x, const long* xs, const void* dy, const long* dys, float* dw, float* workspace, int N,
                           int Cin, int H, int W, int Cout, int R, int S, int stride, int pad, int Ho, int Wo, void* stream) {
    return wgrad_impl<uint16_t>(x, xs, dy, dys, dw, workspace, N, Cin, H, W, Cout, R, S, stride, pad, Ho, Wo, stream);
}

}  // extern "C"
