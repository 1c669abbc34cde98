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
// ([S][BN][32]) are staged next to it.  Pixels and weight rows are 64-byte LDS rows whose 16-byte chunks are XOR-swizzled
// by bit 2 of the row (`swz_chunk`), which makes both fragment reads conflict free for the lane groups a
// `ds_read_b128` is actually served in (PMC: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 50 % -> 0 against an 80-byte pitch).  `v_mfma_f32_16x16x32_bf16`: for channels-last output the weights are
// the A operand and the pixels the B operand -- a lane then holds 4 consecutive output channels of one pixel and
// stores 8 bytes; for NCHW output (consumers in the RepLKNet trunk) the operands swap and a lane holds 4 consecutive
// pixels of one channel.  Global loads of step t+1 are in flight under the MFMAs of step t; LDS is double buffered
// (one barrier per filter row).
#include "common.h"
#include <cstdlib>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int TILE_PX = 128;          // output pixels of a workgroup: TR rows x TC columns, TR * TC <= 128
constexpr int PITCH = 64;             // bytes per pixel / weight row in LDS: 32 channels, no padding
// ds_read_b128 is served in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63}
// (MI355X_MICROARCH.md, LDS): each holds all sixteen rows (lane & 15) of a fragment, rows 4..11 with the OTHER 16-byte
// chunk (lane >> 4) of the pair.  On 64-byte rows the four lanes of a group that share a bank base are rows r, r+4,
// r+8, r+12 with chunks (c, c^1, c^1, c): XOR-ing the chunk index with 2 * bit2(row) makes the four land on four
// different 16-byte slots for every alignment of the first row (exhaustive search: the only family of solutions) ->
// conflict free for consecutive rows, i.e. the weight rows and the pixels of a stride-1 tile row.
__device__ __forceinline__ int swz_chunk(int row) { return ((row >> 2) & 1) << 1; }
// largest halo (pixels) an instantiation stages: stride 1, 3x3: (2,64) -> 4 x 66; stride 2, 3x3: (4,32) -> 9 x 65;
// 7x7: (8,16) -> 14 x 22 (stride 1, the dilated data gradient) / 21 x 37 (stride 2)
constexpr int halo_cap(int stride, int ks) {
    return ks <= 3 ? (stride == 1 ? 4 * 66 : 9 * 65) : (stride == 1 ? 14 * 22 : 21 * 37);
}

__device__ __forceinline__ float bf2f(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }

struct ConvArgs {
    const uint16_t* x;      // [N][H][W][Cin]
    const uint16_t* w;      // packed [R*S][Cout][CinP], CinP = Cin rounded up to 32, zero filled
    const void* bias;       // [Cout] fp32 / bf16 or null
    uint16_t* y;            // [N][Ho][Wo][Cout] or [N][Cout][Ho][Wo]
    int N, H, W, Cin, CinP, Cout, stride, pad, reflect, dil, Ho, Wo, act, bias_bf16;
    int TR, TC, tiles_x, tiles_y;
};

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_ELU = 2, ACT_SIGMOID = 3 };

// BN output channels per workgroup; WM x WN waves; STRIDE and the (square) filter size KS compile time.
// RPS: filter rows whose weights are staged per step (1: one row -- two barriers per row; KS: the whole K x K slice at once --
// two barriers per 32-channel slice, K x the weight bytes in LDS).
template <int BN, int WM, int WN, int STRIDE, bool OUT_NCHW, int KS, int RPS = 1>
__global__ __launch_bounds__(256, 2) void conv_nhwc_kernel(const ConvArgs a) {
    static_assert(WM * WN == 4, "four waves");
    static_assert(RPS == 1 || RPS == KS, "one filter row or all of them per step");
    constexpr int MAX_A = (halo_cap(STRIDE, KS) * 4 + 255) / 256, MAX_B = (RPS * KS * BN * 4 + 255) / 256;
    constexpr int MT = 8 / WM;             // 16-pixel M tiles per wave
    constexpr int NT = BN / 16 / WN;       // 16-channel column tiles per wave
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    const int TR = a.TR, TC = a.TC;
    const int HALO_H = (TR - 1) * STRIDE + KS, HALO_W = (TC - 1) * STRIDE + KS;
    const int halo_px = HALO_H * HALO_W;
    uint8_t* ldsA = lds;
    uint8_t* ldsB = lds + ((halo_px * PITCH + 127) / 128) * 128;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int g = lane >> 4, li = lane & 15;
    const int co0 = blockIdx.x * BN;
    int t = blockIdx.y;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const int oh0 = ty * TR, ow0 = tx * TC;
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
    int a_src[MAX_A];
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
            if (ok) a_src[c] = ((ih / a.dil) * a.W + iw / a.dil) * a.Cin + (q & 3) * 8;      // < 2^31 per image
        }
    }
    constexpr int b_chunks = RPS * KS * BN * 4;
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
    auto store_a = [&]() {
#pragma unroll
        for (int c = 0; c < MAX_A; ++c) {
            const int q = tid + c * 256;
            if (q < a_chunks) *reinterpret_cast<uint4*>(ldsA + (q >> 2) * PITCH + (((q & 3) ^ swz_chunk(q >> 2)) << 4)) = a_reg[c];
        }
    };
    auto load_b = [&](int c0, int r) {
#pragma unroll
        for (int c = 0; c < MAX_B; ++c) {
            const int q = tid + c * 256;
            if (q < b_chunks) {
                const int row = q >> 2;                       // (rr * KS + s) * BN + column; rr = 0 for one row per step
                const int s = row / BN, co = co0 + row % BN;
                b_reg[c] = co < a.Cout ? *reinterpret_cast<const uint4*>(
                                             a.w + ((long)(r * KS + s) * a.Cout + co) * a.CinP + c0 + (q & 3) * 8)
                                       : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto store_b = [&]() {
#pragma unroll
        for (int c = 0; c < MAX_B; ++c) {
            const int q = tid + c * 256;
            if (q < b_chunks) *reinterpret_cast<uint4*>(ldsB + (q >> 2) * PITCH + (((q & 3) ^ swz_chunk(q >> 2)) << 4)) = b_reg[c];
        }
    };

    // per-lane fragment bases: M tile i of this wave covers linear tile pixels m = (wm*MT + i)*16 + li -> (m / TC, m % TC)
    int a_px[MT];                                              // halo pixel (LDS row) of tap (0, 0)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int m = (wm * MT + i) * 16 + li;
        if (m >= TR * TC) m = 0;                               // padding rows of the tile: any valid address
        a_px[i] = (m / TC) * STRIDE * HALO_W + (m % TC) * STRIDE;
    }
    // weight rows s * BN + 16 j + li: bit 2 of the row is bit 2 of li (BN and 16 j are multiples of 8)
    const int b_off = (wn * NT * 16 + li) * PITCH + ((g ^ swz_chunk(li)) << 4);

    // ---- main loop over (channel slice, filter row): registers hold the NEXT step's operands while this one computes ----
    constexpr int SPS = KS / RPS;                              // steps per channel slice
    const int steps = (a.CinP / 32) * SPS;
    load_a(0);
    load_b(0, 0);
    for (int st = 0; st < steps; ++st) {
        const int r0 = (st % SPS) * RPS;                       // first filter row of this step
        __syncthreads();                                       // every wave is done with the previous step's LDS image
        store_b();
        if (r0 == 0) store_a();
        __syncthreads();
        const int nxt = st + 1;
        if (nxt < steps) {
            load_b((nxt / SPS) * 32, (nxt % SPS) * RPS);
            if ((nxt % SPS) == 0) load_a((nxt / SPS) * 32);
        }
#pragma unroll
        for (int rr = 0; rr < RPS; ++rr) {
            const int r = r0 + rr;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bf16x8 af[MT], bfr[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int px = a_px[i] + r * HALO_W + s;
                    af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ldsA + px * PITCH + ((g ^ swz_chunk(px)) << 4)));
                }
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ldsB + b_off + ((rr * KS + s) * BN + j * 16) * PITCH));
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
        }
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
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int cb = co0 + (wn * NT + j) * 16;
            if constexpr (OUT_NCHW) {
                // C: column = li = channel, row = 4 g + e = tile pixel m0 + e (TC % 4 == 0: one output row, 4 columns)
                const int m0 = (wm * MT + i) * 16 + 4 * g;
                const int oh = oh0 + m0 / TC, ow = ow0 + m0 % TC, co = cb + li;
                if (m0 >= TR * TC || oh >= a.Ho || co >= a.Cout || ow >= a.Wo) continue;
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
                // C: column = li = tile pixel, row = 4 g + e = channel cb + 4 g + e
                const int m = (wm * MT + i) * 16 + li;
                const int oh = oh0 + m / TC, ow = ow0 + m % TC, co = cb + 4 * g;
                if (m >= TR * TC || oh >= a.Ho || ow >= a.Wo || co >= a.Cout) continue;
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

// tile shape: TR rows x TC columns (TR * TC <= 128) covering the output with the least padding; NCHW stores want TC % 4 == 0
void pick_tile(ConvArgs& a, int stride, int ks, bool nchw) {
    int best_tr = 8, best_tc = 16;
    long best = (long)((a.Ho + 7) / 8) * ((a.Wo + 15) / 16);
    const int cand[4][2] = {{4, 32}, {2, 64}, {a.Wo <= 64 ? TILE_PX / a.Wo : 0, a.Wo}, {0, 0}};
    for (int c = 0; c < 3; ++c) {
        int tr = cand[c][0], tc = cand[c][1];
        if (tr <= 0 || tc <= 0) continue;
        if (tr > a.Ho) tr = a.Ho;
        if (nchw && (tc & 3)) continue;
        if (((tr - 1) * stride + ks) * ((tc - 1) * stride + ks) > halo_cap(stride, ks)) continue;
        const long tiles = (long)((a.Ho + tr - 1) / tr) * ((a.Wo + tc - 1) / tc);
        if (tiles < best) { best = tiles; best_tr = tr; best_tc = tc; }
    }
    a.TR = best_tr; a.TC = best_tc;
    a.tiles_x = (a.Wo + a.TC - 1) / a.TC; a.tiles_y = (a.Ho + a.TR - 1) / a.TR;
}

template <int BN, int WM, int WN, int STRIDE, bool NCHW, int KS, int RPS = 1>
int launch_conv(const ConvArgs& a, hipStream_t st) {
    const int halo_px = ((a.TR - 1) * STRIDE + KS) * ((a.TC - 1) * STRIDE + KS);
    if (halo_px > halo_cap(STRIDE, KS)) return PPEA_ERR_UNSUPPORTED;
    if constexpr (RPS == 1 && KS == 3 && BN <= 64) {
        // Long contractions (>= 256 input channels): the whole 3 x 3 weight slice per step -- 2 barriers per 32-channel slice
        // instead of 6; the waves of these layers were parked at barriers for 35-50 % of their cycles (PMC, round 2).
        // 1024 -> 512 @6x20 forward 106 -> 59 us, @12x40 161 -> 107, ResNet 512 -> 512 55 -> 38 / data gradient 42 -> 29
        // (profiles/r04_conv_layers_vs_library.txt).  With 64 / 128 input channels the 36 KB weight image per 12-17 KB
        // halo costs occupancy and loses (64 -> 64 @96x320: 74 -> 95 us): those keep one filter row per step.
        static const bool one_row = getenv("PPEA_CONV_RPS") != nullptr && getenv("PPEA_CONV_RPS")[0] == '1';
        if (!one_row && a.CinP >= 256) return launch_conv<BN, WM, WN, STRIDE, NCHW, KS, KS>(a, st);
    }
    const size_t smem = (size_t)((halo_px * PITCH + 127) / 128) * 128 + (size_t)RPS * KS * BN * PITCH;
    if (smem > 160 * 1024) return PPEA_ERR_UNSUPPORTED;
    auto kern = conv_nhwc_kernel<BN, WM, WN, STRIDE, NCHW, KS, RPS>;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    const dim3 grid((a.Cout + BN - 1) / BN, (unsigned)((long)a.tiles_x * a.tiles_y * a.N));
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, a);
    return launch_status();
}

template <int STRIDE, bool NCHW, int KS>
int dispatch_bn(const ConvArgs& a, hipStream_t st) {
    const long tiles = (long)a.tiles_x * a.tiles_y * a.N;
    if constexpr (KS <= 3) {
        // 128 channels per workgroup only when that still leaves two waves of workgroups for the 256 CUs
        // (3 x 3 with >= 256 input channels: 64 channels per workgroup with the whole weight slice per step is faster than
        // 128 with one filter row per step -- 256 -> 128 @48x160 forward 105 -> 97 us; launch_conv)
        if (a.Cout > 64 && tiles * ((a.Cout + 127) / 128) >= 512 && !(KS == 3 && a.CinP >= 256))
            return launch_conv<128, 2, 2, STRIDE, NCHW, KS>(a, st);
    }
    // 64 channels per workgroup unless that leaves CUs without one (small maps, e.g. 1024 -> 512 @6x20: 96 workgroups)
    static const bool no_narrow = getenv("PPEA_CONV_NO_NARROW") != nullptr;           // tuning hook (tools/bench_conv.py)
    if (a.Cout > 32 && (no_narrow || tiles * ((a.Cout + 63) / 64) >= 256)) return launch_conv<64, 2, 2, STRIDE, NCHW, KS>(a, st);
    return launch_conv<32, 4, 1, STRIDE, NCHW, KS>(a, st);
}

template <int STRIDE, bool NCHW>
int dispatch_conv(ConvArgs& a, int ks, hipStream_t st) {
    pick_tile(a, STRIDE, ks, NCHW);
    switch (ks) {
        case 1: return dispatch_bn<STRIDE, NCHW, 1>(a, st);
        case 3: return dispatch_bn<STRIDE, NCHW, 3>(a, st);
        case 7: return dispatch_bn<STRIDE, NCHW, 7>(a, st);
    }
    return PPEA_ERR_UNSUPPORTED;
}

// ---- 3x3 stride-1 convs with few channels: persistent workgroups, weights resident in LDS --------------------------------
// The 192x640 / 96x320 levels of the depth decoders (32 and 64 channels; layers.py:103-135, dec.py:172-245) are HBM bound:
// 1.5 M pixels x 64 bytes.  With one tile per workgroup they ran at 1-1.8 TB/s: every tile re-staged the weights row by
// row (18 KB per 8 KB of input) behind two barriers per filter row, and a tile's load latency was only hidden by the two
// or three other workgroups of the CU.  Here a workgroup keeps ALL taps of the filter in LDS, walks over tiles of 8 x 16
// output pixels, and the halo of the NEXT tile is in flight (global -> registers) under the MFMAs and stores of the current
// one: one barrier per tile, no weight traffic after the first tile.
// NTC: 16-channel column tiles (Cout <= 16 * NTC); SL: 32-channel input slices (CinP = 32 * SL).
template <int NTC, int SL>
__global__ __launch_bounds__(256) void conv3x3_resident_kernel(const ConvArgs a) {
    constexpr int TR = 8, TC = 16, HALO_W = TC + 2, HALO_PX = (TR + 2) * HALO_W;      // 10 x 18 = 180 pixels
    constexpr int CO_ROWS = NTC * 16, MT = 2;
    constexpr int W_BYTES = SL * 9 * CO_ROWS * PITCH, H_BYTES = SL * HALO_PX * PITCH;
    constexpr int A_CHUNKS = SL * HALO_PX * 4, MAX_A = (A_CHUNKS + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t* ldsW = lds;
    uint8_t* ldsH = lds + W_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int tiles = a.tiles_x * a.tiles_y * a.N;              // < 2^31 (host check)

    // resident weights: packed [tap][Cout][CinP] -> LDS [slice][tap][CO_ROWS][32 channels], rows >= Cout zero
    for (int q = tid; q < SL * 9 * CO_ROWS * 4; q += 256) {
        const int c = q & 3, row = q >> 2;                       // row = (sl * 9 + tap) * CO_ROWS + co
        const int co = row % CO_ROWS, st = row / CO_ROWS, tap = st % 9, sl = st / 9;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (co < a.Cout) v = *reinterpret_cast<const uint4*>(a.w + ((long)tap * a.Cout + co) * a.CinP + sl * 32 + c * 8);
        *reinterpret_cast<uint4*>(ldsW + row * PITCH + ((c ^ swz_chunk(co)) << 4)) = v;
    }

    // Halo prefetch: `buffer_load` through inline asm.  Written as ordinary loads hipcc sinks them into the branch that
    // stores them to LDS at the END of the tile (no overlap at all); volatile asm keeps them where they are issued.  The
    // hardware range check supplies the zeros of the padding ring (offset past the tensor), so no load sits under a branch;
    // the wait is explicit (`halo_wait`) and names the registers, which ties the LDS stores to it.
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    const unsigned long xb = (unsigned long)a.x;
    const unsigned nbytes = (unsigned)((long)a.N * a.H * a.W * a.Cin * 2);                 // < 2^31: checked by the host
    i32x4 rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(xb & 0xffffffffu));
    rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(xb >> 32));
    rs[2] = __builtin_amdgcn_readfirstlane((int)nbytes);
    rs[3] = __builtin_amdgcn_readfirstlane(0x00020000);
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;     // a register type the asm operands accept
    u32x4 a_reg[MAX_A];
    auto load_halo = [&](int t) {
        const int tx = t % a.tiles_x, ty = (t / a.tiles_x) % a.tiles_y, n = t / (a.tiles_x * a.tiles_y);
        const int ih0 = ty * TR - a.pad, iw0 = tx * TC - a.pad;
#pragma unroll
        for (int k = 0; k < MAX_A; ++k) {
            const int q = tid + k * 256;
            unsigned off = 0x80000000u;                          // past every tensor: reads zeros
            if (q < A_CHUNKS) {
                const int c = q & 3, rest = q >> 2, px = rest % HALO_PX, sl = rest / HALO_PX;
                int ih = ih0 + px / HALO_W, iw = iw0 + px % HALO_W;
                if (a.reflect) {               // beyond the pad ring only masked outputs read: clamp
                    ih = ih < 0 ? -ih : (ih >= a.H ? 2 * a.H - 2 - ih : ih);
                    iw = iw < 0 ? -iw : (iw >= a.W ? 2 * a.W - 2 - iw : iw);
                    ih = ih < 0 ? 0 : (ih >= a.H ? a.H - 1 : ih);
                    iw = iw < 0 ? 0 : (iw >= a.W ? a.W - 1 : iw);
                }
                const int ch = sl * 32 + c * 8;
                if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W && ch < a.Cin)
                    off = (unsigned)((((n * a.H + ih) * a.W + iw) * a.Cin + ch) * 2);
            }
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(a_reg[k]) : "v"(off), "s"(rs));
        }
    };
    auto halo_wait = [&]() {
        static_assert(MAX_A == 3, "one wait operand per staged 16-byte piece");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(a_reg[0]), "+v"(a_reg[1]), "+v"(a_reg[2]));
    };
    auto store_halo = [&](uint8_t* buf) {
#pragma unroll
        for (int k = 0; k < MAX_A; ++k) {
            const int q = tid + k * 256;
            if (q < A_CHUNKS) {
                const int c = q & 3, rest = q >> 2, px = rest % HALO_PX;          // rest = sl * HALO_PX + px
                *reinterpret_cast<u32x4*>(buf + rest * PITCH + ((c ^ swz_chunk(px)) << 4)) = a_reg[k];
            }
        }
    };

    int a_px[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a_px[i] = (wave * MT + i) * HALO_W + li;          // tile row wave * 2 + i, column li
    const int b_off = li * PITCH + ((g ^ swz_chunk(li)) << 4);

    // bias of this lane's channels (j * 16 + 4 g + e), once
    float bias_r[NTC][4];
#pragma unroll
    for (int j = 0; j < NTC; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = j * 16 + 4 * g + e;
            bias_r[j][e] = 0.f;
            if (a.bias != nullptr && co < a.Cout)
                bias_r[j][e] = a.bias_bf16 ? bf2f(reinterpret_cast<const uint16_t*>(a.bias)[co])
                                           : reinterpret_cast<const float*>(a.bias)[co];
        }

    int t = blockIdx.x;
    load_halo(t);                                              // grid <= tiles
    halo_wait();
    store_halo(ldsH);
    __syncthreads();
    int cur = 0;
    for (; t < tiles; t += gridDim.x) {
        const int tn = t + gridDim.x;
        if (tn < tiles) load_halo(tn);                         // in flight under this tile's MFMAs and stores
        const uint8_t* hb = ldsH + cur * H_BYTES;
        f32x4 acc[MT][NTC];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTC; ++j) acc[i][j] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sl = 0; sl < SL; ++sl)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3) {
                    bf16x8 af[MT], bfr[NTC];
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const int px = a_px[i] + r * HALO_W + s3;
                        af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(
                                                               hb + (sl * HALO_PX + px) * PITCH + ((g ^ swz_chunk(px)) << 4)));
                    }
#pragma unroll
                    for (int j = 0; j < NTC; ++j)
                        bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(
                                                                ldsW + b_off + (((sl * 9 + r * 3 + s3) * NTC + j) * 16) * PITCH));
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NTC; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                }
        // epilogue: column = li = tile pixel, row = 4 g + e = channel (weights were the A operand)
        const int tx = t % a.tiles_x, ty = (t / a.tiles_x) % a.tiles_y, n = t / (a.tiles_x * a.tiles_y);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int oh = ty * TR + wave * MT + i, ow = tx * TC + li;
            if (oh >= a.Ho || ow >= a.Wo) continue;
            uint16_t* row = a.y + (((long)n * a.Ho + oh) * a.Wo + ow) * a.Cout;
#pragma unroll
            for (int j = 0; j < NTC; ++j) {
                const int co = j * 16 + 4 * g;
                if (co >= a.Cout) continue;
                uint16_t v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float f = acc[i][j][e];
                    if (co + e < a.Cout) {
                        f += bias_r[j][e];
                        switch (a.act) {
                            case ACT_RELU: f = f > 0.f ? f : 0.f; break;
                            case ACT_ELU: f = f > 0.f ? f : (__expf(f) - 1.f); break;
                            case ACT_SIGMOID: f = 1.f / (1.f + __expf(-f)); break;
                            default: break;
                        }
                    }
                    v[e] = f2bf(f);
                }
                if (co + 3 < a.Cout && (a.Cout & 3) == 0) {
                    *reinterpret_cast<uint2*>(row + co) = make_uint2(v[0] | ((uint32_t)v[1] << 16), v[2] | ((uint32_t)v[3] << 16));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < a.Cout) row[co + e] = v[e];
                }
            }
        }
        if (tn < tiles) {
            halo_wait();
            store_halo(ldsH + (cur ^ 1) * H_BYTES);
        }
        __syncthreads();
        cur ^= 1;
    }
}

template <int NTC, int SL>
int launch_resident(ConvArgs a, hipStream_t st) {
    a.TR = 8; a.TC = 16;
    a.tiles_x = (a.Wo + 15) / 16; a.tiles_y = (a.Ho + 7) / 8;
    constexpr size_t smem = (size_t)SL * 9 * NTC * 16 * PITCH + 2 * (size_t)SL * 180 * PITCH;
    auto kern = conv3x3_resident_kernel<NTC, SL>;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    const long tiles = (long)a.tiles_x * a.tiles_y * a.N;
    int per_cu = (int)((160 * 1024) / smem);
    per_cu = per_cu > 3 ? 3 : (per_cu < 1 ? 1 : per_cu);
    const long grid = tiles < 256L * per_cu ? tiles : 256L * per_cu;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), smem, st, a);
    return launch_status();
}

// few channels, many pixels: the persistent kernel (at least ~4 tiles per workgroup slot, else the per-tile kernel)
bool resident_ok(const ConvArgs& a, int ks, int out_nchw) {
    const long tiles = (long)((a.Wo + 15) / 16) * ((a.Ho + 7) / 8) * a.N;
    // 32 input channels only: with 64 (74 KB of weights + 46 KB of halos = one workgroup per CU) the per-tile kernel was
    // faster (64->64 @96x320: 76 vs 88 us)
    return ks == 3 && a.stride == 1 && a.dil == 1 && !out_nchw && a.CinP == 32 && a.Cout <= 64 && a.pad <= 2 && tiles >= 2048 &&
           (long)a.N * a.H * a.W * a.Cin * 2 < (1L << 31);
}

int dispatch_resident(const ConvArgs& a, hipStream_t st) {
    switch ((a.Cout + 15) / 16) {
        case 1: return launch_resident<1, 1>(a, st);
        case 2: return launch_resident<2, 1>(a, st);
        case 3: case 4: return launch_resident<4, 1>(a, st);
    }
    return PPEA_ERR_UNSUPPORTED;
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
    if ((Cin % 8) != 0 || R != S || (R != 1 && R != 3 && R != 7) || (stride != 1 && stride != 2) || dil < 1 || pad < 0)
        return PPEA_ERR_UNSUPPORTED;
    if (reflect && (pad > 1 || dil != 1 || H < 2 || W < 2)) return PPEA_ERR_UNSUPPORTED;
    if ((long)H * W * Cin >= (1L << 31)) return PPEA_ERR_UNSUPPORTED;
    ConvArgs a;
    a.x = (const uint16_t*)x; a.w = (const uint16_t*)w_packed; a.bias = bias; a.y = (uint16_t*)y;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.CinP = (Cin + 31) / 32 * 32; a.Cout = Cout;
    a.stride = stride; a.pad = pad; a.reflect = reflect; a.dil = dil; a.Ho = Ho; a.Wo = Wo; a.act = act;
    a.bias_bf16 = bias_bf16;
    hipStream_t st = (hipStream_t)stream;
    static const bool no_resident = getenv("PPEA_CONV_NO_RESIDENT") != nullptr;        // tuning hook (tools/bench_conv.py)
    if (!no_resident && resident_ok(a, R, out_nchw)) return dispatch_resident(a, st);
    if (stride == 1) return out_nchw ? dispatch_conv<1, true>(a, R, st) : dispatch_conv<1, false>(a, R, st);
    return out_nchw ? dispatch_conv<2, true>(a, R, st) : dispatch_conv<2, false>(a, R, st);
}

}  // extern "C"
