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

constexpr int TILE_PX = 128;          // output pixels per staged patch: TR rows x TC columns (4 K steps of 32 pixels)
constexpr int MAX_TAPS = 9;

__device__ __forceinline__ float bf2f(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
// XOR on the 8-byte chunk index of an LDS row of CH channels (bank windows of the transposing reads, see header)
template <int CH> __device__ __forceinline__ int swz(int row) {
    if constexpr (CH == 64) return 4 * ((row >> 1) & 1) + 8 * ((row >> 3) & 1);
    else return 4 * ((row >> 3) & 1);
}

struct WgArgs {
    const uint16_t* dz;     // [N][Ho][Wo][Cout]
    const uint16_t* x;      // [N][H][W][Cin]
    float* ws;              // [slabs][R*S][CoutP][CinP] fp32
    int N, H, W, Cin, Cout, pad, reflect, Ho, Wo;
    int TR, TC, tiles_x, tiles_y, n_patches, splits, rows_per_block, CoutP, CinP;
};

constexpr int halo_cap(int stride, int ks) {
    return ks <= 3 ? (stride == 1 ? 4 * 66 : 9 * 65) : (stride == 1 ? 14 * 22 : 21 * 37);
}

// BM x BN (co x ci) tile per workgroup; the four waves split it in 32 x 32 pieces and, when the tile has fewer than
// four, the K steps of a patch among themselves (each wave then writes its own partial slab).
template <int STRIDE, int KS, int BM, int BN>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgArgs a) {
    constexpr int NMN = (BM / 32) * (BN / 32), WK = 4 / NMN;
    constexpr int ZROW = BM * 2, XROW = BN * 2;                 // LDS row bytes
    constexpr int ZCH = BM / 8, XCH = BN / 8;                   // 16-byte chunks per row
    constexpr int MAX_Z = TILE_PX * ZCH / 256;
    constexpr int MAX_X = (halo_cap(STRIDE, KS) * XCH + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int TR = a.TR, TC = a.TC;
    const int HALO_H = (TR - 1) * STRIDE + KS, HALO_W = (TC - 1) * STRIDE + KS;
    const int halo_px = HALO_H * HALO_W;
    uint8_t* ldsZ = lds;                               // [128 px][BM co]
    uint8_t* ldsX = lds + TILE_PX * ZROW;              // [halo px][BN ci]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave / NMN, wmn = wave % NMN;
    const int wm = wmn / (BN / 32), wn = wmn % (BN / 32);
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    const int co0 = blockIdx.x * BM, ci0 = blockIdx.y * BN;
    const int rgroups = (KS + a.rows_per_block - 1) / a.rows_per_block;
    const int split = blockIdx.z / rgroups;
    const int r0 = (blockIdx.z % rgroups) * a.rows_per_block;
    const int nr = (KS - r0) < a.rows_per_block ? (KS - r0) : a.rows_per_block;
    const int ntap = nr * KS;                          // <= MAX_TAPS

    f32x4 acc[MAX_TAPS][2][2];
#pragma unroll
    for (int t = 0; t < MAX_TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = {0.f, 0.f, 0.f, 0.f};

    const int x_chunks = halo_px * XCH;
    uint4 z_reg[MAX_Z], x_reg[MAX_X];
    // (row, column) of every staged chunk inside the patch: the same for every patch of the loop below -- the divisions by the
    // run-time tile shape are done once, not per patch (they were a third of the loop: 13 chunks x 3 divisions per thread)
    // (the 32 x 32 tile and the stride-2 instantiations keep the in-loop form: with the tables they sit at the 256-VGPR limit
    // and ran slower -- 32 -> 32 @192x640 98 -> 124 us, the three stride-2 ResNet layers 45 / 51 / 43 -> 55 / 62 / 52)
    constexpr bool HOIST = STRIDE == 1 && !(BM == 32 && BN == 32);
    int z_rc[HOIST ? MAX_Z : 1], x_rc[HOIST ? MAX_X : 1];
    auto rc_of = [&](int px, int width) { return ((px / width) << 16) | (px % width); };
    if constexpr (HOIST) {
#pragma unroll
        for (int c = 0; c < MAX_Z; ++c) {
            const int px = (tid + c * 256) / ZCH;
            z_rc[c] = px < TR * TC ? rc_of(px, TC) : -1;
        }
#pragma unroll
        for (int c = 0; c < MAX_X; ++c) {
            const int idx = tid + c * 256;
            x_rc[c] = idx < x_chunks ? rc_of(idx / XCH, HALO_W) : -1;
        }
    }

    auto load_patch = [&](int patch) {
        int t = patch;
        const int tx = t % a.tiles_x; t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        const int n = t / a.tiles_y;
        const int oh0 = ty * TR, ow0 = tx * TC;
        const uint16_t* zn = a.dz + (long)n * a.Ho * a.Wo * a.Cout;
        const uint16_t* xn = a.x + (long)n * a.H * a.W * a.Cin;
#pragma unroll
        for (int c = 0; c < MAX_Z; ++c) {
            const int idx = tid + c * 256, ch = co0 + (idx % ZCH) * 8;
            int oh, ow;
            bool in_tile;
            if constexpr (HOIST) {
                in_tile = z_rc[c] >= 0;
                oh = oh0 + (z_rc[c] >> 16); ow = ow0 + (z_rc[c] & 0xffff);
            } else {
                const int px = idx / ZCH;
                in_tile = px < TR * TC;
                oh = oh0 + px / TC; ow = ow0 + px % TC;
            }
            z_reg[c] = (in_tile && oh < a.Ho && ow < a.Wo && ch < a.Cout)
                           ? *reinterpret_cast<const uint4*>(zn + ((long)oh * a.Wo + ow) * a.Cout + ch) : make_uint4(0, 0, 0, 0);
        }
        const int ih0 = oh0 * STRIDE - a.pad, iw0 = ow0 * STRIDE - a.pad;
#pragma unroll
        for (int c = 0; c < MAX_X; ++c) {
            const int idx = tid + c * 256;
            if (idx < x_chunks) {
                const int ch = ci0 + (idx % XCH) * 8;
                int ih, iw;
                if constexpr (HOIST) {
                    ih = ih0 + (x_rc[c] >> 16); iw = iw0 + (x_rc[c] & 0xffff);
                } else {
                    const int px = idx / XCH;
                    ih = ih0 + px / HALO_W; iw = iw0 + px % HALO_W;
                }
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
        for (int c = 0; c < MAX_Z; ++c) {
            const int idx = tid + c * 256, row = idx / ZCH;
            *reinterpret_cast<uint4*>(ldsZ + row * ZROW + ((((idx % ZCH) * 2) ^ swz<BM>(row)) << 3)) = z_reg[c];
        }
#pragma unroll
        for (int c = 0; c < MAX_X; ++c) {
            const int idx = tid + c * 256;
            if (idx < x_chunks) {
                const int row = idx / XCH;
                *reinterpret_cast<uint4*>(ldsX + row * XROW + ((((idx % XCH) * 2) ^ swz<BN>(row)) << 3)) = x_reg[c];
            }
        }
    };
    auto tr_z = [&](int row_lo, int row_hi, int chunk0) -> bf16x8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ldsZ + row_lo * ZROW + (((chunk0 + p) ^ swz<BM>(row_lo)) << 3)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ldsZ + row_hi * ZROW + (((chunk0 + p) ^ swz<BM>(row_hi)) << 3)));
        const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, both);
    };
    auto tr_x = [&](int row_lo, int row_hi, int chunk0) -> bf16x8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ldsX + row_lo * XROW + (((chunk0 + p) ^ swz<BN>(row_lo)) << 3)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ldsX + row_hi * XROW + (((chunk0 + p) ^ swz<BN>(row_hi)) << 3)));
        const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, both);
    };

    // halo rows of a lane's two contraction pixels for tap (0, 0), per K step of the patch (patch independent as well)
    constexpr int KSTEPS = TILE_PX / 32 / WK;
    auto halo_rows = [&](int kk, int& lo, int& hi) {
        // lane (g, q, p): contraction rows = linear tile pixels k = 32 kk + 8 g + q (lo) and + 4 (hi)
        const int k_lo = 32 * kk + 8 * g + q, k_hi = k_lo + 4;
        // pixels past the tile (TR * TC < 128) carry dz = 0: any row
        const int m_lo = k_lo < TR * TC ? k_lo : 0, m_hi = k_hi < TR * TC ? k_hi : 0;
        lo = (m_lo / TC) * STRIDE * HALO_W + (m_lo % TC) * STRIDE;
        hi = (m_hi / TC) * STRIDE * HALO_W + (m_hi % TC) * STRIDE;
    };
    int xl[HOIST ? KSTEPS : 1], xh[HOIST ? KSTEPS : 1];
    if constexpr (HOIST) {
#pragma unroll
        for (int i = 0; i < KSTEPS; ++i) halo_rows(wk + i * WK, xl[i], xh[i]);
    }

    int patch = split;
    if (patch < a.n_patches) load_patch(patch);
    for (; patch < a.n_patches; patch += a.splits) {
        __syncthreads();                               // everyone is done reading the previous patch
        store_patch();
        __syncthreads();
        if (patch + a.splits < a.n_patches) load_patch(patch + a.splits);   // in flight under the MFMAs below
        // one K step (32 pixels) of the patch for all taps of this workgroup; x_lo / x_hi: halo rows of the lane's two contraction
        // pixels for tap (0, 0).  (A macro, not a lambda: with the body in a lambda the 32 x 32 instantiation took 256 VGPRs
        // instead of 148 and its layers ran 98 -> 116 us.)
#define PPEA_WG_KSTEP(kk_, x_lo_, x_hi_)                                                                                          \
        {                                                                                                                         \
            const int k_lo = 32 * (kk_) + 8 * g + q, k_hi = k_lo + 4;                                                             \
            bf16x8 af[2];                                                                                                         \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) af[i] = tr_z(k_lo, k_hi, (wm * 32 + i * 16) >> 2);                      \
            _Pragma("unroll") for (int t = 0; t < MAX_TAPS; ++t) {                                                                \
                if (t < ntap) { /* wave-uniform */                                                                                \
                    const int off = (r0 + t / KS) * HALO_W + t % KS;                                                              \
                    bf16x8 bfr[2];                                                                                                \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                 \
                        bfr[j] = tr_x((x_lo_) + off, (x_hi_) + off, (wn * 32 + j * 16) >> 2);                                     \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                 \
                        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
                            acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[t][i][j], 0, 0, 0);         \
                }                                                                                                                 \
            }                                                                                                                     \
        }
        if constexpr (HOIST) {
            // (and a lambda here: with the macro body the stride-2 instantiations ran 41 -> 61, 48 -> 68, 39 -> 57 us)
            auto kstep = [&](int kk, int x_lo, int x_hi) { PPEA_WG_KSTEP(kk, x_lo, x_hi) };
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) kstep(wk + ks * WK, xl[ks], xh[ks]);
        } else {
#pragma unroll 1
            for (int kk = wk; kk < TILE_PX / 32; kk += WK) {
                // lane (g, q, p): contraction rows = linear tile pixels k = 32 kk + 8 g + q (lo) and + 4 (hi); pixels past the
                // tile (TR * TC < 128) carry dz = 0: any row
                const int kl = 32 * kk + 8 * g + q, kh = kl + 4;
                const int m_lo = kl < TR * TC ? kl : 0, m_hi = kh < TR * TC ? kh : 0;
                const int x_lo = (m_lo / TC) * STRIDE * HALO_W + (m_lo % TC) * STRIDE;
                const int x_hi = (m_hi / TC) * STRIDE * HALO_W + (m_hi % TC) * STRIDE;
                PPEA_WG_KSTEP(kk, x_lo, x_hi)
            }
        }
#undef PPEA_WG_KSTEP
    }

    // partials: C column = li = ci, row = 4 g + e = co; one slab per (split, K-share of the wave)
#pragma unroll
    for (int t = 0; t < MAX_TAPS; ++t) {
        if (t >= ntap) continue;
        const int tap = (r0 + t / KS) * KS + t % KS;
        float* base = a.ws + (((long)(split * WK + wk) * KS * KS + tap) * a.CoutP) * a.CinP;
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

// sum over slabs, write dW in parameter layout [Cout][Cin][R][S] (fp32 or bf16).  64 consecutive (tap, co, ci) elements
// per workgroup (ci fastest: coalesced slab reads) x 16 slab groups, combined through LDS: small-channel layers have
// few elements and many slabs, so the sum over slabs has to be parallel as well.
__global__ __launch_bounds__(1024) void conv_wgrad_reduce_kernel(const float* __restrict__ ws, void* __restrict__ dw,
                                                                  int dw_bf16, int slabs, int RS, int Cout, int Cin,
                                                                  int CoutP, int CinP) {
    __shared__ float part[16][64];
    const long total = (long)Cout * Cin * RS;
    const long i = (long)blockIdx.x * 64 + threadIdx.x;
    float s = 0.f;
    int ci = 0, co = 0, tap = 0;
    if (i < total) {
        ci = (int)(i % Cin);
        co = (int)((i / Cin) % Cout);
        tap = (int)(i / ((long)Cin * Cout));
        for (int k = threadIdx.y; k < slabs; k += 16) s += ws[(((long)k * RS + tap) * CoutP + co) * CinP + ci];
    }
    part[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && i < total) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += part[k][threadIdx.x];
        const long o = ((long)co * Cin + ci) * RS + tap;
        if (dw_bf16) reinterpret_cast<uint16_t*>(dw)[o] = f32_to_bf16(v);
        else reinterpret_cast<float*>(dw)[o] = v;
    }
}

// few slabs (large-channel layers: many elements, 1-16 slabs): one thread per element, serial sum
__global__ void conv_wgrad_reduce_flat_kernel(const float* __restrict__ ws, void* __restrict__ dw, int dw_bf16, int slabs,
                                              int RS, int Cout, int Cin, int CoutP, int CinP) {
    const long total = (long)Cout * Cin * RS;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        const int co = (int)((i / Cin) % Cout);
        const int tap = (int)(i / ((long)Cin * Cout));
        float s = 0.f;
        for (int k = 0; k < slabs; ++k) s += ws[(((long)k * RS + tap) * CoutP + co) * CinP + ci];
        const long o = ((long)co * Cin + ci) * RS + tap;
        if (dw_bf16) reinterpret_cast<uint16_t*>(dw)[o] = f32_to_bf16(s);
        else reinterpret_cast<float*>(dw)[o] = s;
    }
}

struct WgPlan { int BM, BN, WK, splits, slabs, rows_per_block, rgroups, CoutP, CinP, n_patches, TR, TC, tiles_x, tiles_y; long ws_bytes; };

WgPlan plan(int N, int Cin, int Cout, int K, int stride, int Ho, int Wo) {
    WgPlan p;
    // tile of output pixels with the least padding (same candidates as the forward kernel)
    p.TR = 8; p.TC = 16;
    long best = (long)((Ho + 7) / 8) * ((Wo + 15) / 16);
    const int cand[3][2] = {{4, 32}, {2, 64}, {Wo <= 64 ? TILE_PX / Wo : 0, Wo}};
    for (int c = 0; c < 3; ++c) {
        int tr = cand[c][0], tc = cand[c][1];
        if (tr <= 0 || tc <= 0) continue;
        if (tr > Ho) tr = Ho;
        if (((tr - 1) * stride + K) * ((tc - 1) * stride + K) > halo_cap(stride, K)) continue;
        const long tiles = (long)((Ho + tr - 1) / tr) * ((Wo + tc - 1) / tc);
        if (tiles < best) { best = tiles; p.TR = tr; p.TC = tc; }
    }
    p.tiles_x = (Wo + p.TC - 1) / p.TC; p.tiles_y = (Ho + p.TR - 1) / p.TR;
    p.n_patches = p.tiles_x * p.tiles_y * N;
    p.BM = Cout > 32 ? 64 : 32; p.BN = Cin > 32 ? 64 : 32;
    p.WK = 4 / ((p.BM / 32) * (p.BN / 32));
    p.CoutP = (Cout + p.BM - 1) / p.BM * p.BM; p.CinP = (Cin + p.BN - 1) / p.BN * p.BN;
    p.rows_per_block = (K * K <= MAX_TAPS) ? K : (MAX_TAPS / K > 0 ? MAX_TAPS / K : 1);
    p.rgroups = (K + p.rows_per_block - 1) / p.rows_per_block;
    const long tiles = (long)(p.CoutP / p.BM) * (p.CinP / p.BN) * p.rgroups;
    const long per_slab = (long)K * K * p.CoutP * p.CinP * 4;
    long splits = (512 + tiles - 1) / tiles;                         // two waves of workgroups on 256 CUs
    const long want = splits;
    const long cap = (32L << 20) / (per_slab * p.WK);                // workspace <= 32 MB, or up to 4/3 of that (42.7 MB) ...
    if (splits > cap) splits = cap;
    // ... except that a layer never runs as fewer workgroups than CUs for want of workspace: 1024 -> 512 (19 MB per slab,
    // 128 tiles) ran as 128 workgroups walking 48 patches each -- 307 us against the library's 153; with 4 splits 184 us.
    // (More slabs everywhere -- a 128 MB cap -- was a net loss: every slab is a full-size fp32 gradient written and re-read.)
    // (the same for the layers whose cap leaves them a few workgroups short of one per CU: 512 -> 256 at 24 x 80 ran as
    // 192 workgroups, 256 -> 128 at 48 x 160 as 216: they take `fill` splits when that stays within 4/3 of the cap; 9-16
    // slabs are reduced by the flat kernel -- tests/test_kernels_gpu.py CONV_CASES wgrad_512_256_fill / wgrad_256_128_fill)
    long floor_splits = want < 4 ? want : 4;
    const long fill = (256 + tiles - 1) / tiles;
    if (fill <= want && fill > floor_splits && fill * 3 <= cap * 4) floor_splits = fill;
    if (splits < floor_splits) splits = floor_splits;
    if (splits > p.n_patches) splits = p.n_patches;
    if (splits < 1) splits = 1;
    p.splits = (int)splits;
    p.slabs = p.splits * p.WK;
    p.ws_bytes = per_slab * p.slabs;
    return p;
}

}  // namespace

extern "C" {

long ppea_conv_wgrad_workspace_bytes(int N, int Cin, int Cout, int R, int S, int stride, int Ho, int Wo) {
    if (R != S) return 0;
    return plan(N, Cin, Cout, R, stride, Ho, Wo).ws_bytes;
}

// dz [N][Ho][Wo][Cout], x [N][H][W][Cin] channels-last bf16 (Cin % 8 == 0, Cout % 8 == 0); dw [Cout][Cin][R][S] fp32 or
// bf16 (dw_bf16); workspace of ppea_conv_wgrad_workspace_bytes bytes (caller-owned; the library never allocates).
int ppea_conv_wgrad_nhwc_bf16(const void* dz, const void* x, void* dw, int dw_bf16, void* workspace, int N, int H, int W,
                              int Cin, int Cout, int R, int S, int stride, int pad, int reflect, int Ho, int Wo, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Ho <= 0 || Wo <= 0 || !workspace) return PPEA_ERR_ARG;
    if ((Cin % 8) != 0 || (Cout % 8) != 0 || R != S || (R != 1 && R != 3 && R != 7) || (stride != 1 && stride != 2) || pad < 0)
        return PPEA_ERR_UNSUPPORTED;
    if (reflect && (pad > 1 || H < 2 || W < 2)) return PPEA_ERR_UNSUPPORTED;
    const WgPlan p = plan(N, Cin, Cout, R, stride, Ho, Wo);
    WgArgs a;
    a.dz = (const uint16_t*)dz; a.x = (const uint16_t*)x; a.ws = (float*)workspace;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.pad = pad;
    a.reflect = reflect; a.Ho = Ho; a.Wo = Wo; a.TR = p.TR; a.TC = p.TC; a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y;
    a.n_patches = p.n_patches; a.splits = p.splits; a.rows_per_block = p.rows_per_block; a.CoutP = p.CoutP; a.CinP = p.CinP;
    hipStream_t st = (hipStream_t)stream;
    const int halo_px = ((p.TR - 1) * stride + R) * ((p.TC - 1) * stride + R);
    const size_t smem = (size_t)TILE_PX * p.BM * 2 + (size_t)halo_px * p.BN * 2;
    if (smem > 160 * 1024) return PPEA_ERR_UNSUPPORTED;
    const dim3 grid(p.CoutP / p.BM, p.CinP / p.BN, p.splits * p.rgroups);
#define WG_LAUNCH(STRIDE_, KS_, BM_, BN_)                                                                               \
    do {                                                                                                                \
        auto kern = conv_wgrad_kernel<STRIDE_, KS_, BM_, BN_>;                                                          \
        if (smem > 64 * 1024) {                                                                                         \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                     \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                   \
            if (e != hipSuccess) return (int)e;                                                                         \
        }                                                                                                               \
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, a);                                                         \
    } while (0)
#define WG_TILES(STRIDE_, KS_)                                                                                          \
    do {                                                                                                                \
        if (p.BM == 64 && p.BN == 64) WG_LAUNCH(STRIDE_, KS_, 64, 64);                                                  \
        else if (p.BM == 64) WG_LAUNCH(STRIDE_, KS_, 64, 32);                                                           \
        else if (p.BN == 64) WG_LAUNCH(STRIDE_, KS_, 32, 64);                                                           \
        else WG_LAUNCH(STRIDE_, KS_, 32, 32);                                                                           \
    } while (0)
    if (stride == 1) { if (R == 1) WG_TILES(1, 1); else if (R == 3) WG_TILES(1, 3); else WG_TILES(1, 7); }
    else { if (R == 1) WG_TILES(2, 1); else if (R == 3) WG_TILES(2, 3); else WG_TILES(2, 7); }
#undef WG_TILES
#undef WG_LAUNCH
    int err = launch_status();
    if (err) return err;
    const long total = (long)Cout * Cin * R * S;
    if (p.slabs > 16) {
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64, 16), 0, st,
                           (const float*)workspace, dw, dw_bf16, p.slabs, R * S, Cout, Cin, p.CoutP, p.CinP);
    } else {
        const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
        hipLaunchKernelGGL(conv_wgrad_reduce_flat_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, dw, dw_bf16,
                           p.slabs, R * S, Cout, Cin, p.CoutP, p.CinP);
    }
    return launch_status();
}

}  // extern "C"
