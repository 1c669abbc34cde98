// Weight gradient of the channels-last bf16 convolutions of conv_nhwc.hip on the matrix cores.  gfx950.
//
//   dW[r][s][co][ci] = sum_{n, oh, ow} dZ[n][oh][ow][co] * X[n][oh*stride - pad + r][ow*stride - pad + s][ci]
//
// (dZ = gradient at the conv output, after the activation's derivative; X as the forward read it: zero padding or
// reflected indices).  Replaces the library wgrad kernels behind `ConvBlock` / `Conv3x3` (layers.py:103-135), the
// pose ResNet-18 (networks/resnet_encoder.py:25-72), `PoseDecoder`, `reduce_conv`.
//
// GEMM view per tap: rows = co, columns = ci, contraction over PIXELS -- the slow axis of both channels-last
// operands.  Both fragments therefore come out of `ds_read_b64_tr_b16` (the CDNA4 transposing LDS read): the
// dZ patch [128 pixels][64 co] and the X halo patch [(7*stride+R) x (15*stride+S) pixels][64 ci] are staged exactly as
// they lie in memory (coalesced 16-byte rows), the halo ONCE for all taps; a K step is 32 pixels (two patch rows).
// An XOR of the 8-byte chunk index with 4*bit1(row) + 8*bit3(row) puts the eight rows a 32-lane half reads
// ({p..p+3} and {p+8..p+11}, any p) on eight different 32-byte bank windows: conflict free at stride 1.
// A workgroup owns a 64 x 64 (co x ci) tile of every tap of up to three filter rows and a strided subset of the
// output patches (split-K); fp32 partials go to a workspace and a second kernel sums them and writes the gradient
// in the parameter's own layout [Cout][Cin][R][S] and dtype (deterministic: no float atomics).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

constexpr int TH = 8, TW = 16, BM = 64, BN = 64, ROW = 128;     // LDS row = 64 channels x 2 B
constexpr int MAX_TAPS = 9;

__device__ __forceinline__ float bf2f(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
__device__ __forceinline__ int swz(int row) { return 4 * ((row >> 1) & 1) + 8 * ((row >> 3) & 1); }   // 8-byte chunks

struct WgArgs {
    const uint16_t* dz;     // [N][Ho][Wo][Cout]
    const uint16_t* x;      // [N][H][W][Cin]
    float* ws;              // [splits][R*S][CoutP][CinP] fp32, CoutP / CinP = rounded up to 64
    int N, H, W, Cin, Cout, R, S, stride, pad, reflect, Ho, Wo;
    int tiles_x, tiles_y, n_patches, splits, rows_per_block, CoutP, CinP;
};

constexpr int max_x(int stride, int kmax) { return (((TH - 1) * stride + kmax) * ((TW - 1) * stride + kmax) * 8 + 255) / 256; }

template <int STRIDE, int KMAX>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgArgs a) {
    constexpr int MAX_X = max_x(STRIDE, KMAX);
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int R = a.R, S = a.S;
    const int HALO_H = (TH - 1) * STRIDE + R, HALO_W = (TW - 1) * STRIDE + S;
    const int halo_px = HALO_H * HALO_W;
    uint8_t* ldsZ = lds;                               // [128 px][64 co]
    uint8_t* ldsX = lds + TH * TW * ROW;               // [halo px][64 ci]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;           // 2 x 2 waves, 32 x 32 each
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    const int co0 = blockIdx.x * BM, ci0 = blockIdx.y * BN;
    const int rgroups = (R + a.rows_per_block - 1) / a.rows_per_block;
    const int split = blockIdx.z / rgroups;
    const int r0 = (blockIdx.z % rgroups) * a.rows_per_block;
    const int nr = (R - r0) < a.rows_per_block ? (R - r0) : a.rows_per_block;
    const int ntap = nr * S;                           // <= MAX_TAPS

    f32x4 acc[MAX_TAPS][2][2];
#pragma unroll
    for (int t = 0; t < MAX_TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = {0.f, 0.f, 0.f, 0.f};

    const int x_chunks = halo_px * 8;
    uint4 z_reg[4], x_reg[MAX_X];

    auto load_patch = [&](int patch) {
        int t = patch;
        const int tx = t % a.tiles_x; t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        const int n = t / a.tiles_y;
        const int oh0 = ty * TH, ow0 = tx * TW;
        const uint16_t* zn = a.dz + (long)n * a.Ho * a.Wo * a.Cout;
        const uint16_t* xn = a.x + (long)n * a.H * a.W * a.Cin;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int idx = tid + c * 256, px = idx >> 3, ch = co0 + (idx & 7) * 8;
            const int oh = oh0 + px / TW, ow = ow0 + px % TW;
            z_reg[c] = (oh < a.Ho && ow < a.Wo && ch < a.Cout)
                           ? *reinterpret_cast<const uint4*>(zn + ((long)oh * a.Wo + ow) * a.Cout + ch) : make_uint4(0, 0, 0, 0);
        }
        const int ih0 = oh0 * STRIDE - a.pad, iw0 = ow0 * STRIDE - a.pad;
#pragma unroll
        for (int c = 0; c < MAX_X; ++c) {
            const int idx = tid + c * 256;
            if (idx < x_chunks) {
                const int px = idx >> 3, ch = ci0 + (idx & 7) * 8;
                int ih = ih0 + px / HALO_W, iw = iw0 + px % HALO_W;
                if (a.reflect) {
                    ih = ih < 0 ? -ih : (ih >= a.H ? 2 * a.H - 2 - ih : ih);
                    iw = iw < 0 ? -iw : (iw >= a.W ? 2 * a.W - 2 - iw : iw);
                    ih = ih < 0 ? 0 : (ih >= a.H ? a.H - 1 : ih);
                    iw = iw < 0 ? 0 : (iw >= a.W ? a.W - 1 : iw);
                }
                x_reg[c] = (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W && ch < a.Cin)
                               ? *reinterpret_cast<const uint4*>(xn + ((long)ih * a.W + iw) * a.Cin + ch) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int idx = tid + c * 256, row = idx >> 3;
            *reinterpret_cast<uint4*>(ldsZ + row * ROW + ((((idx & 7) * 2) ^ swz(row)) << 3)) = z_reg[c];
        }
#pragma unroll
        for (int c = 0; c < MAX_X; ++c) {
            const int idx = tid + c * 256;
            if (idx < x_chunks) {
                const int row = idx >> 3;
                *reinterpret_cast<uint4*>(ldsX + row * ROW + ((((idx & 7) * 2) ^ swz(row)) << 3)) = x_reg[c];
            }
        }
    };
    auto tr8 = [&](const uint8_t* base, int row_lo, int row_hi, int chunk0) -> bf16x8 {
        // rows row_lo (+q handled by the caller) ... : two transposing reads, 4 contraction rows each
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + row_lo * ROW + (((chunk0 + p) ^ swz(row_lo)) << 3)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + row_hi * ROW + (((chunk0 + p) ^ swz(row_hi)) << 3)));
        const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, both);
    };

    int patch = split;
    if (patch < a.n_patches) load_patch(patch);
    for (; patch < a.n_patches; patch += a.splits) {
        __syncthreads();                               // everyone is done reading the previous patch
        store_patch();
        __syncthreads();
        if (patch + a.splits < a.n_patches) load_patch(patch + a.splits);   // in flight under the MFMAs below
#pragma unroll 1
        for (int kk = 0; kk < TH / 2; ++kk) {          // K steps of 32 pixels = patch rows 2kk, 2kk+1
            // lane (g, q, p): contraction rows k = 8g + q (lo) and 8g + 4 + q (hi): patch row 2kk + (g >> 1),
            // patch columns 8 (g & 1) + q and + 4
            const int prow = 2 * kk + (g >> 1), pcol = 8 * (g & 1) + q;
            bf16x8 af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int zr = prow * TW + pcol;
                af[i] = tr8(ldsZ, zr, zr + 4, (wm * 32 + i * 16) >> 2);
            }
#pragma unroll
            for (int t = 0; t < MAX_TAPS; ++t) {
                if (t < ntap) {                        // wave-uniform
                    const int r = r0 + t / S, s = t % S;
                    const int xr = (prow * STRIDE + r) * HALO_W + pcol * STRIDE + s;
                    bf16x8 bfr[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) bfr[j] = tr8(ldsX, xr, xr + 4 * STRIDE, (wn * 32 + j * 16) >> 2);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[t][i][j], 0, 0, 0);
                }
            }
        }
    }

    // partials: C column = li = ci, row = 4 g + e = co
#pragma unroll
    for (int t = 0; t < MAX_TAPS; ++t) {
        if (t >= ntap) continue;
        const int tap = (r0 + t / S) * S + t % S;
        float* base = a.ws + (((long)split * R * S + tap) * a.CoutP) * a.CinP;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = co0 + wm * 32 + i * 16 + 4 * g + e, ci = ci0 + wn * 32 + j * 16 + li;
                    base[(long)co * a.CinP + ci] = acc[t][i][j][e];
                }
    }
}

// sum over splits, write dW in parameter layout [Cout][Cin][R][S] (fp32 or bf16)
__global__ void conv_wgrad_reduce_kernel(const float* __restrict__ ws, void* __restrict__ dw, int dw_bf16, int splits, int RS,
                                         int Cout, int Cin, int CoutP, int CinP) {
    const long total = (long)Cout * Cin * RS;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // i enumerates (tap, co, ci) with ci fastest: coalesced workspace reads; the store scatters by R*S
        const int ci = (int)(i % Cin);
        const int co = (int)((i / Cin) % Cout);
        const int tap = (int)(i / ((long)Cin * Cout));
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += ws[(((long)k * RS + tap) * CoutP + co) * CinP + ci];
        const long o = ((long)co * Cin + ci) * RS + tap;
        if (dw_bf16) reinterpret_cast<uint16_t*>(dw)[o] = f32_to_bf16(s);
        else reinterpret_cast<float*>(dw)[o] = s;
    }
}

struct WgPlan { int splits, rows_per_block, rgroups, CoutP, CinP, n_patches, tiles_x, tiles_y; long ws_bytes; };

WgPlan plan(int N, int Cin, int Cout, int R, int S, int Ho, int Wo) {
    WgPlan p;
    p.tiles_x = (Wo + TW - 1) / TW; p.tiles_y = (Ho + TH - 1) / TH;
    p.n_patches = p.tiles_x * p.tiles_y * N;
    p.CoutP = (Cout + BM - 1) / BM * BM; p.CinP = (Cin + BN - 1) / BN * BN;
    p.rows_per_block = (R * S <= MAX_TAPS) ? R : (MAX_TAPS / S > 0 ? MAX_TAPS / S : 1);
    p.rgroups = (R + p.rows_per_block - 1) / p.rows_per_block;
    const long tiles = (long)(p.CoutP / BM) * (p.CinP / BN) * p.rgroups;
    const long per_split = (long)R * S * p.CoutP * p.CinP * 4;
    long splits = (768 + tiles - 1) / tiles;                         // enough workgroups to fill 256 CUs three times
    const long cap = (64L << 20) / per_split;                        // workspace <= 64 MB
    if (splits > cap) splits = cap;
    if (splits > p.n_patches) splits = p.n_patches;
    if (splits < 1) splits = 1;
    p.splits = (int)splits;
    p.ws_bytes = per_split * splits;
    return p;
}

}  // namespace

extern "C" {

long ppea_conv_wgrad_workspace_bytes(int N, int Cin, int Cout, int R, int S, int Ho, int Wo) {
    return plan(N, Cin, Cout, R, S, Ho, Wo).ws_bytes;
}

// dz [N][Ho][Wo][Cout], x [N][H][W][Cin] channels-last bf16 (Cin % 8 == 0, Cout % 8 == 0); dw [Cout][Cin][R][S] fp32 or
// bf16 (dw_bf16); workspace of ppea_conv_wgrad_workspace_bytes bytes (caller-owned; the library never allocates).
int ppea_conv_wgrad_nhwc_bf16(const void* dz, const void* x, void* dw, int dw_bf16, void* workspace, int N, int H, int W,
                              int Cin, int Cout, int R, int S, int stride, int pad, int reflect, int Ho, int Wo, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Ho <= 0 || Wo <= 0 || !workspace) return PPEA_ERR_ARG;
    if ((Cin % 8) != 0 || (Cout % 8) != 0 || R < 1 || S < 1 || R > 7 || S > 7 || (stride != 1 && stride != 2) || pad < 0)
        return PPEA_ERR_UNSUPPORTED;
    if (reflect && (pad > 1 || H < 2 || W < 2)) return PPEA_ERR_UNSUPPORTED;
    const WgPlan p = plan(N, Cin, Cout, R, S, Ho, Wo);
    WgArgs a;
    a.dz = (const uint16_t*)dz; a.x = (const uint16_t*)x; a.ws = (float*)workspace;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.R = R; a.S = S; a.stride = stride; a.pad = pad;
    a.reflect = reflect; a.Ho = Ho; a.Wo = Wo; a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.n_patches = p.n_patches;
    a.splits = p.splits; a.rows_per_block = p.rows_per_block; a.CoutP = p.CoutP; a.CinP = p.CinP;
    hipStream_t st = (hipStream_t)stream;
    const int halo_px = ((TH - 1) * stride + R) * ((TW - 1) * stride + S);
    const size_t smem = (size_t)(TH * TW + halo_px) * ROW;
    if (smem > 160 * 1024) return PPEA_ERR_UNSUPPORTED;
    const dim3 grid(p.CoutP / BM, p.CinP / BN, p.splits * p.rgroups);
#define WG_LAUNCH(STRIDE_, KMAX_)                                                                                       \
    do {                                                                                                                \
        auto kern = conv_wgrad_kernel<STRIDE_, KMAX_>;                                                                  \
        if (smem > 64 * 1024) {                                                                                         \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                     \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                   \
            if (e != hipSuccess) return (int)e;                                                                         \
        }                                                                                                               \
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, a);                                                         \
    } while (0)
    const bool wide = R > 3 || S > 3;
    if (stride == 1) { if (wide) WG_LAUNCH(1, 7); else WG_LAUNCH(1, 3); }
    else { if (wide) WG_LAUNCH(2, 7); else WG_LAUNCH(2, 3); }
#undef WG_LAUNCH
    int err = launch_status();
    if (err) return err;
    const long total = (long)Cout * Cin * R * S;
    const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, dw, dw_bf16,
                       p.splits, R * S, Cout, Cin, p.CoutP, p.CinP);
    return launch_status();
}

}  // extern "C"
