// Dense convolution as an implicit GEMM on the matrix cores: channels-last bf16 in, fp32 accumulate, bf16 out.  gfx950.
//
// Replaces the library (MIOpen / CK) convolutions of the hot path:
//   * depth decoders `ConvBlock` / `Conv3x3` (layers.py:103-135; networks/depth_decoder_v2.py:172-245): 3x3, reflection
//     pad 1 (read through reflected indices -- no padded copy), bias + ELU / bias + sigmoid in the epilogue;
//   * pose ResNet-18 (networks/resnet_encoder.py:25-72): 7x7 s2, 3x3 s1/s2, 1x1 s2, zero pad, no bias;
//   * `PoseDecoder` (networks/pose_decoder.py:33-52): 1x1 / 3x3 with bias (+ ReLU);
//   * `reduce_conv` (networks/replk_matching_adapter.py:127-131): 3x3 zero pad, bias + ReLU;
//   * RepLKNet stem[0] (networks/replknet_adapter.py:411): 3x3 s2 on the image.
// and their DATA gradients (the same kernel run with flipped / transposed weights; a strided forward becomes a unit
// stride pass over the zero-dilated output gradient, `dil`).
//
//   Y[n][oh][ow][co] = act( bias[co] + sum_{r,s,ci} Wp[r*S+s][co][ci] * X[n][oh*stride - pad + r][ow*stride - pad + s][ci] )
//
// GEMM view: rows = output pixels, columns = output channels, contraction = (r, s, ci).  A workgroup (4 waves) owns a
// 8 x 16 patch of output pixels and BN output channels.  Per 32-channel slice of the input it stages the patch's HALO
// ((7*stride + R) x (15*stride + S) pixels x 32 channels) in LDS ONCE and reuses it for all R*S taps -- the im2col
// matrix is never materialised and every input byte is fetched once per slice; the weights of one filter row
// ([S][BN][32]) are staged next to it.  Pixels and weight rows sit on an 80-byte pitch (64 B data + 16 B): the sixteen
// lanes of a `ds_read_b128` group then hit sixteen different 16-byte slots (20 i mod 64 is a permutation), so both
// fragment reads are conflict free at stride 1.  `v_mfma_f32_16x16x32_bf16`: for channels-last output the weights are
// the A operand and the pixels the B operand -- a lane then holds 4 consecutive output channels of one pixel and
// stores 8 bytes; for NCHW output (consumers in the RepLKNet trunk) the operands swap and a lane holds 4 consecutive
// pixels of one channel.  Global loads of step t+1 are in flight under the MFMAs of step t; LDS is double buffered
// (one barrier per filter row).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int TH = 8, TW = 16;        // output patch of a workgroup
constexpr int PITCH = 80;             // bytes per pixel / weight row in LDS (32 channels + 16 B)
// staging registers per thread: 16-byte halo chunks / weight chunks for the largest filter an instantiation serves
constexpr int max_a(int stride, int kmax) { return (((TH - 1) * stride + kmax) * ((TW - 1) * stride + kmax) * 4 + 255) / 256; }
constexpr int max_b(int bn, int kmax) { return (kmax * bn * 4 + 255) / 256; }

__device__ __forceinline__ float bf2f(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }

struct ConvArgs {
    const uint16_t* x;      // [N][H][W][Cin]
    const uint16_t* w;      // packed [R*S][Cout][CinP], CinP = Cin rounded up to 32, zero filled
    const void* bias;       // [Cout] fp32 / bf16 or null
    uint16_t* y;            // [N][Ho][Wo][Cout] or [N][Cout][Ho][Wo]
    int N, H, W, Cin, CinP, Cout, R, S, stride, pad, reflect, dil, Ho, Wo, act, bias_bf16;
    int tiles_x, tiles_y;
};

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_ELU = 2, ACT_SIGMOID = 3 };

// BN output channels per workgroup; WM x WN waves; STRIDE compile time (halo geometry); NCHW output flag.
template <int BN, int WM, int WN, int STRIDE, bool OUT_NCHW, int KMAX>
__global__ __launch_bounds__(256) void conv_nhwc_kernel(const ConvArgs a) {
    static_assert(WM * WN == 4, "four waves");
    constexpr int MAX_A = max_a(STRIDE, KMAX), MAX_B = max_b(BN, KMAX);
    constexpr int MT = TH / WM;            // 16-pixel rows per wave
    constexpr int NT = BN / 16 / WN;       // 16-channel column tiles per wave
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    const int R = a.R, S = a.S;
    const int HALO_H = (TH - 1) * STRIDE + R, HALO_W = (TW - 1) * STRIDE + S;
    const int halo_px = HALO_H * HALO_W;
    const int A_BYTES = ((halo_px * PITCH + 127) / 128) * 128;
    const int B_BYTES = S * BN * PITCH;
    const int n_abuf = a.CinP > 32 ? 2 : 1;        // a single channel slice never re-stages the halo
    uint8_t* ldsA[2] = {lds, lds + (n_abuf - 1) * A_BYTES};
    uint8_t* ldsB[2] = {lds + n_abuf * A_BYTES, lds + n_abuf * A_BYTES + B_BYTES};

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int g = lane >> 4, li = lane & 15;
    const int co0 = blockIdx.x * BN;
    int t = blockIdx.y;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const int oh0 = ty * TH, ow0 = tx * TW;
    const int ih0 = oh0 * STRIDE - a.pad, iw0 = ow0 * STRIDE - a.pad;
    const int Hl = (a.H - 1) * a.dil + 1, Wl = (a.W - 1) * a.dil + 1;      // logical (zero-dilated) input size
    const uint16_t* xn = a.x + (long)n * a.H * a.W * a.Cin;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = {0.f, 0.f, 0.f, 0.f};

    // ---- staging plans -------------------------------------------------------------------------------------------
    // halo: chunk q = pixel * 4 + c16; its source offset (in elements, -1 = zero) does not depend on the channel slice
    const int a_chunks = halo_px * 4;
    long a_src[MAX_A];
#pragma unroll
    for (int c = 0; c < MAX_A; ++c) {
        const int q = tid + c * 256;
        a_src[c] = -1;
        if (q < a_chunks) {
            const int px = q >> 2;
            int ih = ih0 + px / HALO_W, iw = iw0 + px % HALO_W;
            if (a.reflect) {               // rows / columns beyond the pad ring only feed masked outputs: clamp them
                ih = ih < 0 ? -ih : (ih >= Hl ? 2 * Hl - 2 - ih : ih);
                iw = iw < 0 ? -iw : (iw >= Wl ? 2 * Wl - 2 - iw : iw);
                ih = ih < 0 ? 0 : (ih >= Hl ? Hl - 1 : ih);
                iw = iw < 0 ? 0 : (iw >= Wl ? Wl - 1 : iw);
            }
            const bool ok = ih >= 0 && ih < Hl && iw >= 0 && iw < Wl && (ih % a.dil) == 0 && (iw % a.dil) == 0;
            if (ok) a_src[c] = ((long)(ih / a.dil) * a.W + iw / a.dil) * a.Cin + (q & 3) * 8;
        }
    }
    const int b_chunks = S * BN * 4;
    uint4 a_reg[MAX_A], b_reg[MAX_B];

    auto load_a = [&](int c0) {
#pragma unroll
        for (int c = 0; c < MAX_A; ++c) {
            const int q = tid + c * 256;
            if (q < a_chunks) {
                const int ch = c0 + (q & 3) * 8;
                a_reg[c] = (a_src[c] >= 0 && ch < a.Cin) ? *reinterpret_cast<const uint4*>(xn + a_src[c] + c0)
                                                         : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto store_a = [&](uint8_t* dst) {
#pragma unroll
        for (int c = 0; c < MAX_A; ++c) {
            const int q = tid + c * 256;
            if (q < a_chunks) *reinterpret_cast<uint4*>(dst + (q >> 2) * PITCH + (q & 3) * 16) = a_reg[c];
        }
    };
    auto load_b = [&](int c0, int r) {
#pragma unroll
        for (int c = 0; c < MAX_B; ++c) {
            const int q = tid + c * 256;
            if (q < b_chunks) {
                const int row = q >> 2;                       // s * BN + column
                const int s = row / BN, co = co0 + row % BN;
                b_reg[c] = co < a.Cout ? *reinterpret_cast<const uint4*>(
                                             a.w + ((long)(r * S + s) * a.Cout + co) * a.CinP + c0 + (q & 3) * 8)
                                       : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto store_b = [&](uint8_t* dst) {
#pragma unroll
        for (int c = 0; c < MAX_B; ++c) {
            const int q = tid + c * 256;
            if (q < b_chunks) *reinterpret_cast<uint4*>(dst + (q >> 2) * PITCH + (q & 3) * 16) = b_reg[c];
        }
    };

    // ---- main loop over (channel slice, filter row) ------------------------------------------------------------------
    const int n_slices = a.CinP / 32, steps = n_slices * R;
    load_a(0);
    load_b(0, 0);
    store_a(ldsA[0]);
    store_b(ldsB[0]);
    __syncthreads();
    int pa = 0;
    for (int st = 0; st < steps; ++st) {
        const int r = st % R;
        const int nxt = st + 1;
        const bool more = nxt < steps;
        const bool new_slice = more && (nxt % R) == 0;
        if (more) {
            load_b((nxt / R) * 32, nxt % R);
            if (new_slice) load_a((nxt / R) * 32);
        }
        const uint8_t* As = ldsA[pa];
        const uint8_t* Bs = ldsB[st & 1];
        const uint8_t* a_base = As + ((wm * MT * STRIDE + r) * HALO_W + li * STRIDE) * PITCH + g * 16;
        const uint8_t* b_base = Bs + (wn * NT * 16 + li) * PITCH + g * 16;
        for (int s = 0; s < S; ++s) {
            bf16x8 af[MT], bfr[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a_base + (i * STRIDE * HALO_W + s) * PITCH));
#pragma unroll
            for (int j = 0; j < NT; ++j)
                bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(b_base + (s * BN + j * 16) * PITCH));
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if constexpr (OUT_NCHW)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                }
        }
        if (more) {
            store_b(ldsB[(st + 1) & 1]);
            if (new_slice) store_a(ldsA[pa ^ 1]);
        }
        __syncthreads();
        if (new_slice) pa ^= 1;
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------
    auto activate = [&](float v) -> float {
        switch (a.act) {
            case ACT_RELU: return v > 0.f ? v : 0.f;
            case ACT_ELU: return v > 0.f ? v : (__expf(v) - 1.f);
            case ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
            default: return v;
        }
    };
    auto bias_at = [&](int co) -> float {
        if (a.bias == nullptr) return 0.f;
        return a.bias_bf16 ? bf2f(reinterpret_cast<const uint16_t*>(a.bias)[co]) : reinterpret_cast<const float*>(a.bias)[co];
    };
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int oh = oh0 + wm * MT + i;
        if (oh >= a.Ho) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int cb = co0 + (wn * NT + j) * 16;
            if constexpr (OUT_NCHW) {
                // C: column = li = channel, row = 4 g + e = pixel ow0 + 4 g + e
                const int co = cb + li, ow = ow0 + 4 * g;
                if (co >= a.Cout || ow >= a.Wo) continue;
                const float bv = bias_at(co);
                uint16_t* dst = a.y + (((long)n * a.Cout + co) * a.Ho + oh) * a.Wo + ow;
                uint16_t v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = f2bf(activate(acc[i][j][e] + bv));
                if (ow + 3 < a.Wo && (a.Wo & 3) == 0) {
                    *reinterpret_cast<uint2*>(dst) = make_uint2(v[0] | ((uint32_t)v[1] << 16), v[2] | ((uint32_t)v[3] << 16));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (ow + e < a.Wo) dst[e] = v[e];
                }
            } else {
                // C: column = li = pixel ow0 + li, row = 4 g + e = channel cb + 4 g + e
                const int ow = ow0 + li, co = cb + 4 * g;
                if (ow >= a.Wo || co >= a.Cout) continue;
                uint16_t* dst = a.y + (((long)n * a.Ho + oh) * a.Wo + ow) * a.Cout + co;
                uint16_t v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (co + e < a.Cout) ? f2bf(activate(acc[i][j][e] + bias_at(co + e))) : 0;
                if (co + 3 < a.Cout && (a.Cout & 3) == 0) {
                    *reinterpret_cast<uint2*>(dst) = make_uint2(v[0] | ((uint32_t)v[1] << 16), v[2] | ((uint32_t)v[3] << 16));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < a.Cout) dst[e] = v[e];
                }
            }
        }
    }
}

template <int BN, int WM, int WN, int STRIDE, bool NCHW, int KMAX>
int launch_conv(const ConvArgs& a, hipStream_t st) {
    const int halo_px = ((TH - 1) * STRIDE + a.R) * ((TW - 1) * STRIDE + a.S);
    const int A_BYTES = ((halo_px * PITCH + 127) / 128) * 128, B_BYTES = a.S * BN * PITCH;
    const size_t smem = (a.CinP > 32 ? 2 : 1) * (size_t)A_BYTES + 2 * (size_t)B_BYTES;
    if (smem > 160 * 1024) return PPEA_ERR_UNSUPPORTED;
    auto kern = conv_nhwc_kernel<BN, WM, WN, STRIDE, NCHW, KMAX>;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    const dim3 grid((a.Cout + BN - 1) / BN, (unsigned)((long)a.tiles_x * a.tiles_y * a.N));
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, a);
    return launch_status();
}

template <int STRIDE, bool NCHW>
int dispatch_conv(const ConvArgs& a, hipStream_t st) {
    const long tiles = (long)a.tiles_x * a.tiles_y * a.N;
    if (a.R > 3 || a.S > 3) {                         // 7x7 (pose conv1) / 5x5: one wide-halo instantiation
        if (a.Cout > 32) return launch_conv<64, 2, 2, STRIDE, NCHW, 7>(a, st);
        return launch_conv<32, 4, 1, STRIDE, NCHW, 7>(a, st);
    }
    if (a.Cout > 64 && tiles * ((a.Cout + 127) / 128) >= 256) return launch_conv<128, 2, 2, STRIDE, NCHW, 3>(a, st);
    if (a.Cout > 32) return launch_conv<64, 2, 2, STRIDE, NCHW, 3>(a, st);
    return launch_conv<32, 4, 1, STRIDE, NCHW, 3>(a, st);
}

// weights [Cout][Cin][R][S] (bf16 or fp32) -> packed bf16 [R*S][Cout][CinP]  (flip = 0)
//                                         -> packed bf16 [R*S][Cin][CoutP] with both taps reversed (flip = 1: dgrad)
template <typename T>
__global__ void conv_pack_kernel(const T* __restrict__ w, uint16_t* __restrict__ out, int Cout, int Cin, int RS, int flip) {
    const int rows = flip ? Cin : Cout, cols = flip ? Cout : Cin;
    const int colsP = (cols + 31) / 32 * 32;
    const long total = (long)RS * rows * colsP;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % colsP);
        const int rw = (int)((i / colsP) % rows);
        const int tap = (int)(i / ((long)colsP * rows));
        float v = 0.f;
        if (c < cols) {
            const int co = flip ? c : rw, ci = flip ? rw : c;
            const int tp = flip ? RS - 1 - tap : tap;
            v = ld_f32<T>(w + ((long)co * Cin + ci) * RS + tp);
        }
        out[i] = f2bf(v);
    }
}

// NCHW fp32 image(s) -> channels-last bf16 with the channel count padded to Cp (zeros), y = (x - sub) / div
// (a true division: resnet_encoder.py:399 computes (x - 0.45) / 0.225 and the bf16 rounding must see the same value)
__global__ void image_to_nhwc_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, long NHW, int HW, int C, int Cp,
                                     float sub, float div) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < NHW; i += (long)gridDim.x * blockDim.x) {
        const long n = i / HW, p = i % HW;
        for (int c = 0; c < Cp; ++c)
            y[i * Cp + c] = c < C ? f2bf((x[(n * C + c) * HW + p] - sub) / div) : (uint16_t)0;
    }
}

}  // namespace

extern "C" {

static long packed_elems(int Cout, int Cin, int R, int S, int flip) {
    const int rows = flip ? Cin : Cout, cols = flip ? Cout : Cin;
    return (long)R * S * rows * ((cols + 31) / 32 * 32);
}

long ppea_conv_packed_bytes(int Cout, int Cin, int R, int S, int flip) { return 2 * packed_elems(Cout, Cin, R, S, flip); }

// w_is_bf16: dtype of the source weight.  flip = 0: forward operand; flip = 1: data-gradient operand.
int ppea_conv_pack_weights(const void* w, int w_is_bf16, void* packed, int Cout, int Cin, int R, int S, int flip,
                           void* stream) {
    if (Cout <= 0 || Cin <= 0 || R <= 0 || S <= 0) return PPEA_ERR_ARG;
    const long total = packed_elems(Cout, Cin, R, S, flip);
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (w_is_bf16)
        hipLaunchKernelGGL(conv_pack_kernel<uint16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t*)w, (uint16_t*)packed, Cout, Cin, R * S, flip);
    else
        hipLaunchKernelGGL(conv_pack_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)w,
                           (uint16_t*)packed, Cout, Cin, R * S, flip);
    return launch_status();
}

int ppea_image_to_nhwc_bf16(const float* x, void* y, int N, int C, int H, int W, int Cp, float sub, float div, void* stream) {
    if (N <= 0 || C <= 0 || Cp < C || (Cp % 8) != 0) return PPEA_ERR_ARG;
    const long NHW = (long)N * H * W;
    const int blocks = (int)((NHW + 255) / 256 > 8192 ? 8192 : (NHW + 255) / 256);
    hipLaunchKernelGGL(image_to_nhwc_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)y, NHW, H * W, C, Cp,
                       sub, div);
    return launch_status();
}

// x [N][H][W][Cin] bf16 channels-last (Cin % 8 == 0); packed weights from ppea_conv_pack_weights (Cin padded to 32);
// y [N][Ho][Wo][Cout] bf16 (out_nchw = 0) or [N][Cout][Ho][Wo] (out_nchw = 1).
// stride in {1, 2}; pad >= 0 zero padding, or reflect != 0: reflection padding (pad <= 1);
// dil > 1: x is read as its zero-dilated image of (H-1)*dil+1 x (W-1)*dil+1 (data gradient of a strided conv);
// act: 0 none, 1 ReLU, 2 ELU, 3 sigmoid, applied after the bias.
int ppea_conv_nhwc_bf16(const void* x, const void* w_packed, const void* bias, int bias_bf16, void* y, int N, int H, int W,
                        int Cin, int Cout, int R, int S, int stride, int pad, int reflect, int dil, int Ho, int Wo, int act,
                        int out_nchw, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Ho <= 0 || Wo <= 0) return PPEA_ERR_ARG;
    if ((Cin % 8) != 0 || R < 1 || S < 1 || R > 7 || S > 7 || (stride != 1 && stride != 2) || dil < 1 || pad < 0)
        return PPEA_ERR_UNSUPPORTED;
    if (reflect && (pad > 1 || dil != 1 || H < 2 || W < 2)) return PPEA_ERR_UNSUPPORTED;
    ConvArgs a;
    a.x = (const uint16_t*)x; a.w = (const uint16_t*)w_packed; a.bias = bias; a.y = (uint16_t*)y;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.CinP = (Cin + 31) / 32 * 32; a.Cout = Cout; a.R = R; a.S = S;
    a.stride = stride; a.pad = pad; a.reflect = reflect; a.dil = dil; a.Ho = Ho; a.Wo = Wo; a.act = act;
    a.bias_bf16 = bias_bf16;
    a.tiles_x = (Wo + TW - 1) / TW; a.tiles_y = (Ho + TH - 1) / TH;
    hipStream_t st = (hipStream_t)stream;
    if (stride == 1) return out_nchw ? dispatch_conv<1, true>(a, st) : dispatch_conv<1, false>(a, st);
    return out_nchw ? dispatch_conv<2, true>(a, st) : dispatch_conv<2, false>(a, st);
}

}  // extern "C"
