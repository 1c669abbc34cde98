// Image-fed convolutions on the matrix cores: RepLKNet stem[0] (networks/replknet_adapter.py:411, 3x3 stride 2 on the
// RGB frame) and the pose ResNet-18 conv1 (networks/resnet_encoder.py:376-388, 7x7 stride 2 on a frame pair).  gfx950.
//
// With 3 / 6 input channels (zero-padded to 8, channels-last) a per-tap contraction would waste 3/4 of every
// 32-deep MFMA.  Instead a whole FILTER ROW is one contraction: for filter row r and output pixel (oh, ow) the S taps
// x 8 channels are S*8 CONTIGUOUS bf16 of the channels-last frame, starting at pixel (oh*stride - pad + r,
// ow*stride - pad) -- so A fragments are plain 16-byte LDS reads at a per-lane pixel offset (overlapping windows, no
// im2col), the weights are packed [r][Cout][s*8 + ci] (zero-padded to a multiple of 32), and a 7x7 conv costs
// 7 x 2 MFMA steps per tile instead of 49, a 3x3 conv 3 instead of 9.
//   forward:  conv_image_kernel   (no bias / activation: both layers are followed by BatchNorm)
//   wgrad:    conv_image_wgrad_kernel (conv1 only; stem[0] is frozen, repdepth.py:47-50) -- contraction over pixels,
//             both operands through `ds_read_b64_tr_b16`, split over output patches, deterministic reduce into the
//             parameter layout [Cout][Cin][K][K].
// Neither layer needs a data gradient (their input is the frame).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

constexpr int TH = 8, TW = 16;

__device__ __forceinline__ uint16_t f2bf(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }
constexpr int kp_of(int ks) { return (ks * 8 + 31) / 32 * 32; }

struct ImgArgs {
    const uint16_t* x;     // [N][H][W][8]
    const uint16_t* w;     // packed [KS][Cout][KP]
    uint16_t* y;           // [N][Ho][Wo][Cout] or [N][Cout][Ho][Wo]
    int N, H, W, Cout, pad, Ho, Wo, tiles_x, tiles_y;
};

template <int KS, int STRIDE, int BN, bool OUT_NCHW>
__global__ __launch_bounds__(256, 2) void conv_image_kernel(const ImgArgs a) {
    constexpr int KP = kp_of(KS), KC = KP / 32;
    constexpr int HALO_H = (TH - 1) * STRIDE + KS, HALO_W = (TW - 1) * STRIDE + KS;
    constexpr int ROW_PX = HALO_W + (KP - KS * 8) / 8;            // + zero pixels the padded contraction runs into
    constexpr int ROWB = ROW_PX * 16;
    constexpr int PITCHB = KP * 2 + 16;                           // weight row pitch: 36 n mod 64 is a permutation
    constexpr int A_BYTES = (HALO_H * ROWB + 127) / 128 * 128;
    constexpr int MT = 4, NT = BN / 32;                           // 2 x 2 waves
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t* ldsA = lds;
    uint8_t* ldsB = lds + A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, li = lane & 15;
    const int co0 = blockIdx.x * BN;
    int t = blockIdx.y;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const int oh0 = ty * TH, ow0 = tx * TW;
    const int ih0 = oh0 * STRIDE - a.pad, iw0 = ow0 * STRIDE - a.pad;
    const uint16_t* xn = a.x + (long)n * a.H * a.W * 8;

    for (int q = tid; q < HALO_H * ROW_PX; q += 256) {
        const int hr = q / ROW_PX, hc = q - hr * ROW_PX;
        const int ih = ih0 + hr, iw = iw0 + hc;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (hc < HALO_W && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W)
            v = *reinterpret_cast<const uint4*>(xn + ((long)ih * a.W + iw) * 8);
        *reinterpret_cast<uint4*>(ldsA + hr * ROWB + hc * 16) = v;
    }
    constexpr int B_CH = KP / 8;                                  // 16-byte chunks per weight row
    for (int q = tid; q < KS * BN * B_CH; q += 256) {
        const int row = q / B_CH, ch = q - row * B_CH;            // row = r * BN + column
        const int r = row / BN, co = co0 + row % BN;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (co < a.Cout) v = *reinterpret_cast<const uint4*>(a.w + ((long)r * a.Cout + co) * KP + ch * 8);
        *reinterpret_cast<uint4*>(ldsB + row * PITCHB + ch * 16) = v;
    }
    __syncthreads();

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = {0.f, 0.f, 0.f, 0.f};
    const uint8_t* a_base = ldsA + (wm * MT * STRIDE) * ROWB + li * STRIDE * 16 + g * 16;
    const uint8_t* b_base = ldsB + (wn * NT * 16 + li) * PITCHB + g * 16;
#pragma unroll
    for (int r = 0; r < KS; ++r)
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            bf16x8 af[MT], bfr[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a_base + (i * STRIDE + r) * ROWB + c * 64));
#pragma unroll
            for (int j = 0; j < NT; ++j)
                bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(b_base + (r * BN + j * 16) * PITCHB + c * 64));
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
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int oh = oh0 + wm * MT + i;
        if (oh >= a.Ho) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int cb = co0 + (wn * NT + j) * 16;
            if constexpr (OUT_NCHW) {
                const int co = cb + li, ow = ow0 + 4 * g;
                if (co >= a.Cout || ow >= a.Wo) continue;
                uint16_t* dst = a.y + (((long)n * a.Cout + co) * a.Ho + oh) * a.Wo + ow;
                if (ow + 3 < a.Wo && (a.Wo & 3) == 0) {
                    *reinterpret_cast<uint2*>(dst) = make_uint2(f2bf(acc[i][j][0]) | ((uint32_t)f2bf(acc[i][j][1]) << 16),
                                                                f2bf(acc[i][j][2]) | ((uint32_t)f2bf(acc[i][j][3]) << 16));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (ow + e < a.Wo) dst[e] = f2bf(acc[i][j][e]);
                }
            } else {
                const int ow = ow0 + li, co = cb + 4 * g;
                if (ow >= a.Wo || co >= a.Cout) continue;
                uint16_t* dst = a.y + (((long)n * a.Ho + oh) * a.Wo + ow) * a.Cout + co;
                if (co + 3 < a.Cout && (a.Cout & 3) == 0) {
                    *reinterpret_cast<uint2*>(dst) = make_uint2(f2bf(acc[i][j][0]) | ((uint32_t)f2bf(acc[i][j][1]) << 16),
                                                                f2bf(acc[i][j][2]) | ((uint32_t)f2bf(acc[i][j][3]) << 16));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < a.Cout) dst[e] = f2bf(acc[i][j][e]);
                }
            }
        }
    }
}

// w [Cout][Cin][KS][KS] (bf16 or fp32, Cin <= 8) -> [KS][Cout][KP] bf16, element s * 8 + ci
template <typename T>
__global__ void image_pack_kernel(const T* __restrict__ w, uint16_t* __restrict__ out, int Cout, int Cin, int KS, int KP) {
    const long total = (long)KS * Cout * KP;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % KP);
        const int co = (int)((i / KP) % Cout);
        const int r = (int)(i / ((long)KP * Cout));
        const int s = k >> 3, ci = k & 7;
        float v = 0.f;
        if (s < KS && ci < Cin) v = ld_f32<T>(w + (((long)co * Cin + ci) * KS + r) * KS + s);
        out[i] = f2bf(v);
    }
}

// ---- weight gradient (7x7 / 3x3, stride 2): dW[r][co][s*8+ci] = sum_px dz[px][co] * x[(oh*S - pad + r)][(ow*S - pad)*8 + s*8 + ci]
struct ImgWgArgs {
    const uint16_t* dz;    // [N][Ho][Wo][Cout]
    const uint16_t* x;     // [N][H][W][8]
    float* ws;             // [splits][KS][CoutP][KP]
    int N, H, W, Cout, CoutP, pad, Ho, Wo, tiles_x, tiles_y, n_patches, splits;
};

__device__ __forceinline__ int swz64(int row) { return 4 * ((row >> 1) & 1) + 8 * ((row >> 3) & 1); }

template <int KS, int STRIDE>
__global__ __launch_bounds__(256) void conv_image_wgrad_kernel(const ImgWgArgs a) {
    constexpr int KP = kp_of(KS), NTW = KP / 32;                  // 32-wide column pieces of the (s, ci) axis
    constexpr int HALO_H = (TH - 1) * STRIDE + KS, HALO_W = (TW - 1) * STRIDE + KS;
    constexpr int ROW_PX = HALO_W + (KP - KS * 8) / 8, ROWB = ROW_PX * 16;
    constexpr int ZROW = 128;                                     // 64 co x 2 B
    constexpr int X_CH = (HALO_H * ROW_PX + 255) / 256;
    static_assert(NTW <= 2, "at most 64 packed columns");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t* ldsZ = lds;                                          // [128 px][64 co], swizzled
    uint8_t* ldsX = lds + TH * TW * ZROW;                         // raw halo rows

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 64 co x KP columns: waves 2 (co) x NTW (columns) [x 2 K-shares when KP == 32]
    constexpr int WK = 4 / (2 * NTW);
    const int wk = wave / (2 * NTW), wmn = wave % (2 * NTW);
    const int wm = wmn / NTW, wn = wmn % NTW;
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    const int co0 = blockIdx.x * 64, split = blockIdx.y;

    f32x4 acc[KS][2][2];
#pragma unroll
    for (int r = 0; r < KS; ++r)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[r][i][j] = {0.f, 0.f, 0.f, 0.f};

    uint4 z_reg[4], x_reg[X_CH];
    auto load_patch = [&](int patch) {
        int t = patch;
        const int tx = t % a.tiles_x; t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        const int n = t / a.tiles_y;
        const int oh0 = ty * TH, ow0 = tx * TW;
        const uint16_t* zn = a.dz + (long)n * a.Ho * a.Wo * a.Cout;
        const uint16_t* xn = a.x + (long)n * a.H * a.W * 8;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int idx = tid + c * 256, px = idx >> 3, ch = co0 + (idx & 7) * 8;
            const int oh = oh0 + px / TW, ow = ow0 + px % TW;
            z_reg[c] = (oh < a.Ho && ow < a.Wo && ch < a.Cout)
                           ? *reinterpret_cast<const uint4*>(zn + ((long)oh * a.Wo + ow) * a.Cout + ch) : make_uint4(0, 0, 0, 0);
        }
        const int ih0 = oh0 * STRIDE - a.pad, iw0 = ow0 * STRIDE - a.pad;
#pragma unroll
        for (int c = 0; c < X_CH; ++c) {
            const int idx = tid + c * 256;
            if (idx < HALO_H * ROW_PX) {
                const int hr = idx / ROW_PX, hc = idx - hr * ROW_PX;
                const int ih = ih0 + hr, iw = iw0 + hc;
                x_reg[c] = (hc < HALO_W && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W)
                               ? *reinterpret_cast<const uint4*>(xn + ((long)ih * a.W + iw) * 8) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int idx = tid + c * 256, row = idx >> 3;
            *reinterpret_cast<uint4*>(ldsZ + row * ZROW + ((((idx & 7) * 2) ^ swz64(row)) << 3)) = z_reg[c];
        }
#pragma unroll
        for (int c = 0; c < X_CH; ++c) {
            const int idx = tid + c * 256;
            if (idx < HALO_H * ROW_PX) *reinterpret_cast<uint4*>(ldsX + idx * 16) = x_reg[c];
        }
    };
    auto tr_pair = [&](const uint8_t* lo_p, const uint8_t* hi_p) -> bf16x8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)lo_p);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)hi_p);
        const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, both);
    };

    int patch = split;
    if (patch < a.n_patches) load_patch(patch);
    for (; patch < a.n_patches; patch += a.splits) {
        __syncthreads();
        store_patch();
        __syncthreads();
        if (patch + a.splits < a.n_patches) load_patch(patch + a.splits);
#pragma unroll 1
        for (int kk = wk; kk < TH / 2; kk += WK) {
            // contraction rows of this lane: tile pixels k = 32 kk + 8 g + q (lo) and + 4 (hi)
            const int prow = 2 * kk + (g >> 1), pcol = 8 * (g & 1) + q;
            const int zr = prow * TW + pcol;
            bf16x8 af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c0 = (wm * 32 + i * 16) >> 2;
                af[i] = tr_pair(ldsZ + zr * ZROW + (((c0 + p) ^ swz64(zr)) << 3),
                                ldsZ + (zr + 4) * ZROW + (((c0 + p) ^ swz64(zr + 4)) << 3));
            }
            // the packed (s, ci) columns of pixel (prow, pcol) and filter row r start at halo pixel (prow*S + r, pcol*S)
            const uint8_t* xlo = ldsX + (prow * STRIDE) * ROWB + pcol * STRIDE * 16 + (wn * 32 + 4 * p) * 2;
            const uint8_t* xhi = xlo + 4 * STRIDE * 16;
#pragma unroll
            for (int r = 0; r < KS; ++r) {
                bf16x8 bfr[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) bfr[j] = tr_pair(xlo + r * ROWB + j * 32, xhi + r * ROWB + j * 32);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[r][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[r][i][j], 0, 0, 0);
            }
        }
    }
    // partial slab (split, K-share): C column = li -> packed column, row = 4 g + e -> co
#pragma unroll
    for (int r = 0; r < KS; ++r) {
        float* base = a.ws + (((long)(split * WK + wk) * KS + r) * a.CoutP) * KP;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = co0 + wm * 32 + i * 16 + 4 * g + e, k = wn * 32 + j * 16 + li;
                    base[(long)co * KP + k] = acc[r][i][j][e];
                }
    }
}

__global__ void image_wgrad_reduce_kernel(const float* __restrict__ ws, void* __restrict__ dw, int dw_bf16, int slabs,
                                          int KS, int KP, int Cout, int CoutP, int Cin) {
    const long total = (long)Cout * Cin * KS * KS;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int s = (int)(i % KS);
        const int r = (int)((i / KS) % KS);
        const int ci = (int)((i / ((long)KS * KS)) % Cin);
        const int co = (int)(i / ((long)KS * KS * Cin));
        float v = 0.f;
        for (int k = 0; k < slabs; ++k) v += ws[(((long)k * KS + r) * CoutP + co) * KP + s * 8 + ci];
        if (dw_bf16) reinterpret_cast<uint16_t*>(dw)[i] = f32_to_bf16(v);
        else reinterpret_cast<float*>(dw)[i] = v;
    }
}

struct ImgPlan { int tiles_x, tiles_y, n_patches, splits, slabs, CoutP, WK; long ws_bytes; };
ImgPlan img_plan(int N, int Cout, int KS, int Ho, int Wo) {
    ImgPlan p;
    p.tiles_x = (Wo + TW - 1) / TW; p.tiles_y = (Ho + TH - 1) / TH;
    p.n_patches = p.tiles_x * p.tiles_y * N;
    p.CoutP = (Cout + 63) / 64 * 64;
    p.WK = 4 / (2 * (kp_of(KS) / 32));
    const long per_slab = (long)KS * p.CoutP * kp_of(KS) * 4;
    long splits = 512 / (p.CoutP / 64);
    const long cap = (16L << 20) / (per_slab * p.WK);
    if (splits > cap) splits = cap;
    if (splits > p.n_patches) splits = p.n_patches;
    if (splits < 1) splits = 1;
    p.splits = (int)splits; p.slabs = p.splits * p.WK; p.ws_bytes = per_slab * p.slabs;
    return p;
}

template <int KS, int STRIDE, int BN, bool NCHW>
int launch_image(const ImgArgs& a, hipStream_t st) {
    constexpr int KP = kp_of(KS), HALO_H = (TH - 1) * STRIDE + KS, HALO_W = (TW - 1) * STRIDE + KS;
    constexpr int ROWB = (HALO_W + (KP - KS * 8) / 8) * 16;
    constexpr size_t smem = (size_t)((HALO_H * ROWB + 127) / 128 * 128) + (size_t)KS * BN * (KP * 2 + 16);
    static_assert(smem <= 160 * 1024, "LDS");
    auto kern = conv_image_kernel<KS, STRIDE, BN, NCHW>;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    const dim3 grid((a.Cout + BN - 1) / BN, (unsigned)((long)a.tiles_x * a.tiles_y * a.N));
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, a);
    return launch_status();
}

}  // namespace

extern "C" {

long ppea_conv_image_packed_bytes(int Cout, int K) { return (long)K * Cout * kp_of(K) * 2; }

// w [Cout][Cin][K][K] (Cin <= 8; bf16 or fp32) -> the row-packed operand of ppea_conv_image_bf16
int ppea_conv_image_pack_weights(const void* w, int w_is_bf16, void* packed, int Cout, int Cin, int K, void* stream) {
    if (Cout <= 0 || Cin <= 0 || Cin > 8 || (K != 3 && K != 7)) return PPEA_ERR_UNSUPPORTED;
    const long total = (long)K * Cout * kp_of(K);
    const int blocks = (int)((total + 255) / 256);
    if (w_is_bf16)
        hipLaunchKernelGGL(image_pack_kernel<uint16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)w,
                           (uint16_t*)packed, Cout, Cin, K, kp_of(K));
    else
        hipLaunchKernelGGL(image_pack_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)w,
                           (uint16_t*)packed, Cout, Cin, K, kp_of(K));
    return launch_status();
}

// x [N][H][W][8] bf16 channels-last frames (ppea_image_to_nhwc_bf16), stride 2, K in {3, 7};
// y [N][Ho][Wo][Cout] bf16, or [N][Cout][Ho][Wo] with out_nchw.
int ppea_conv_image_bf16(const void* x, const void* w_packed, void* y, int N, int H, int W, int Cout, int K, int stride,
                         int pad, int Ho, int Wo, int out_nchw, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Ho <= 0 || Wo <= 0) return PPEA_ERR_ARG;
    if (stride != 2 || (K != 3 && K != 7) || pad < 0) return PPEA_ERR_UNSUPPORTED;
    ImgArgs a;
    a.x = (const uint16_t*)x; a.w = (const uint16_t*)w_packed; a.y = (uint16_t*)y;
    a.N = N; a.H = H; a.W = W; a.Cout = Cout; a.pad = pad; a.Ho = Ho; a.Wo = Wo;
    a.tiles_x = (Wo + TW - 1) / TW; a.tiles_y = (Ho + TH - 1) / TH;
    hipStream_t st = (hipStream_t)stream;
    if (K == 7) return out_nchw ? launch_image<7, 2, 64, true>(a, st) : launch_image<7, 2, 64, false>(a, st);
    if (Cout > 64) return out_nchw ? launch_image<3, 2, 128, true>(a, st) : launch_image<3, 2, 128, false>(a, st);
    return out_nchw ? launch_image<3, 2, 64, true>(a, st) : launch_image<3, 2, 64, false>(a, st);
}

long ppea_conv_image_wgrad_workspace_bytes(int N, int Cout, int K, int Ho, int Wo) {
    if (K != 3 && K != 7) return 0;
    return img_plan(N, Cout, K, Ho, Wo).ws_bytes;
}

// dz [N][Ho][Wo][Cout] (Cout % 8 == 0), x [N][H][W][8]; dw [Cout][Cin][K][K] fp32 or bf16 (Cin <= 8 real channels).
int ppea_conv_image_wgrad_bf16(const void* dz, const void* x, void* dw, int dw_bf16, void* workspace, int N, int H, int W,
                               int Cin, int Cout, int K, int stride, int pad, int Ho, int Wo, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Ho <= 0 || Wo <= 0 || !workspace) return PPEA_ERR_ARG;
    if (stride != 2 || (K != 3 && K != 7) || Cin > 8 || (Cout % 8) != 0 || pad < 0) return PPEA_ERR_UNSUPPORTED;
    const ImgPlan p = img_plan(N, Cout, K, Ho, Wo);
    ImgWgArgs a;
    a.dz = (const uint16_t*)dz; a.x = (const uint16_t*)x; a.ws = (float*)workspace;
    a.N = N; a.H = H; a.W = W; a.Cout = Cout; a.CoutP = p.CoutP; a.pad = pad; a.Ho = Ho; a.Wo = Wo;
    a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.n_patches = p.n_patches; a.splits = p.splits;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(p.CoutP / 64, p.splits);
    if (K == 7) {
        constexpr int HALO_H = 7 * 2 + 7, ROW_PX = 15 * 2 + 7 + (kp_of(7) - 56) / 8;
        const size_t smem = (size_t)TH * TW * 128 + (size_t)HALO_H * ROW_PX * 16;
        hipLaunchKernelGGL((conv_image_wgrad_kernel<7, 2>), grid, dim3(256), smem, st, a);
    } else {
        constexpr int HALO_H = 7 * 2 + 3, ROW_PX = 15 * 2 + 3 + (kp_of(3) - 24) / 8;
        const size_t smem = (size_t)TH * TW * 128 + (size_t)HALO_H * ROW_PX * 16;
        hipLaunchKernelGGL((conv_image_wgrad_kernel<3, 2>), grid, dim3(256), smem, st, a);
    }
    int err = launch_status();
    if (err) return err;
    const long total = (long)Cout * Cin * K * K;
    hipLaunchKernelGGL(image_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const float*)workspace, dw, dw_bf16, p.slabs, K, kp_of(K), Cout, p.CoutP, Cin);
    return launch_status();
}

}  // extern "C"
