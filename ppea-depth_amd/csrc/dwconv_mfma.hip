// Large-kernel depthwise convolution on the matrix cores (bf16 in / fp32 accumulate), gfx950.
//
// Banded-Toeplitz formulation (im2col-free).  For one channel and one filter row ky
//     out[y][x] += sum_j in[y + ky - P][j] * T_ky[j][x],      T_ky[j][x] = w[ky][j - x + P]
// i.e. a GEMM  (rows y) x (cols x)  with the contraction over input columns j.  Per 16x16
// output tile only j in [x0 - P, x0 + 15 + P] contributes, which fits NS chunks of 32
// columns, so a tile costs K * NS `v_mfma_f32_16x16x32_bf16` (48 % of their MACs are useful at
// K = 31 -- still ~8x the fp32 vector peak).
//
//   A operand = 16 input rows x 32 input columns: one `ds_read_b128` per lane from the LDS image
//               of the plane (lane l: row l&15, columns 8*(l>>4)..+7), immediate offsets for
//               (ky, chunk); row stride = 288 B (== 32 mod 256 -> conflict-free b128 reads);
//   B operand = Toeplitz fragment of filter row ky; depends on (ky, chunk, lane) only, so ALL
//               K*NS fragments of the wave's channel live in registers for the whole kernel
//               (K = 31: 248 VGPRs; one wave per SIMD owns the 512-entry file) -- weights are
//               read once per workgroup, every MFMA's B comes from registers;
//   C tile    = 4 fp32 per lane, chained over ky (dependent 16x16x32 MFMAs issue back to back).
//
// A wave owns one work item = (group of G stacked planes of one channel with G*H <= 48 rows, or a
// 48-row band of one plane) x (segment of NSEG column tiles).  The four waves of a workgroup share
// the channel and its padded filter image in LDS.  The 5x5 re-param branch (fwd: second output;
// dgrad: second input, same accumulator) costs 5 more MFMAs per tile.
// Replaces nn.Conv2d(groups=C) at networks/replknet_adapter.py:225-239 under bf16 autocast.
#include "common.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

#ifdef DW_PROF
// Phase timing (tools/dwconv_phases.py builds a separate library with -DDW_PROF): s_memtime deltas per wave.
__device__ unsigned long long g_dw_prof[4096][8];
#define PROF_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define PROF_ADD(slot, a, b) prof[slot] += (b) - (a)
#else
#define PROF_T(v)
#define PROF_ADD(slot, a, b)
#endif

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int WAVES = 4;
constexpr int LDS_LIMIT = 160 * 1024;

// LDS row stride in bytes for WL staged columns: smallest S >= 2*WL with S/32 odd, so that the 16
// rows x 4 k-groups of a ds_read_b128 fall on distinct 16-byte slots of the 256-byte bank row.
constexpr int lds_stride_bytes(int wl) {
    int s = (2 * wl + 31) / 32;
    if ((s & 1) == 0) ++s;
    return 32 * s;
}

template <int K>
struct Geo {
    static constexpr int P = K / 2;
    static constexpr int NS = (16 + 2 * P <= 32) ? 1 : 2;      // 32-column chunks per tile
    static constexpr int JOFF = (NS == 1) ? 8 : 16;            // first chunk starts at x0 - JOFF
    static_assert(JOFF >= P && 32 * NS - JOFF >= 16 + P, "band does not fit the chunks");
};

template <int K, int NSEG>
struct Seg {
    static constexpr int WL = 16 * (NSEG - 1) + 32 * Geo<K>::NS;     // staged columns per item
    static constexpr int STRIDE = lds_stride_bytes(WL);
};

// Packed filter image (built once per weight version by ppea_dwconv_lk_pack_bf16): the filter itself, bf16
//   img[c][ky * K + t] = w[c][ky][t]   (dgrad: both axes reversed),  padded to a multiple of 8 elements per channel
// -- 1.9 KB per channel at k = 31.  The Toeplitz fragment of (filter row ky, chunk s) in register layout is
//   frag[lane][0..7] = wpad[ky][i0 + j],  i0 = 32 + 32 s + 8 (lane >> 4) - (lane & 15) + (P - JOFF),
//   wpad[ky][i] = w[ky][i - 32] for 0 <= i - 32 < K, else 0,
// eight consecutive taps at a 2-byte-granular offset.  Every wave builds the K * NS (+ KS) fragments of its channel in its
// prologue through its own (still unused) LDS tile region: the filter rows are written twice, zero-padded, the second copy
// shifted by one element, so that every lane's window starts on a 4-byte boundary of one of the copies and a fragment is
// two `ds_read2_b32`.  Round 1 / 2 gathered the fragments from a 4-shifted-copy image in global memory (134 poorly
// coalesced loads per lane, 12-30 % of a wave's time); the first half of round 3 stored the fragments themselves (62 KB per
// channel: ONE coalesced 16-byte load per fragment, but 30 MB of filter image per stage-2 launch against 18 MB of
// activations -- 25 % of that kernel's time, tools/dwconv_phases.py).
constexpr int frag_ns(int K) { return (16 + 2 * (K / 2) <= 32) ? 1 : 2; }
constexpr int packed_elems(int K) { return (K * K + 7) & ~7; }
constexpr int FR_COPY_B = 192;                                  // one zero-padded filter row: 96 bf16
constexpr int FR_ROW_B = 2 * FR_COPY_B;                         // + the copy shifted by one element
constexpr int frag_scratch_bytes(int K, int KS) { return (K + KS) * FR_ROW_B; }

template <int K, int NS>
__device__ __forceinline__ void build_bfrags(bf16x8 (&bf)[K][NS], const uint16_t* __restrict__ img, uint8_t* scratch,
                                             int lane) {
    constexpr int P = K / 2, JOFF = (NS == 1) ? 8 : 16;
    constexpr int PIECES = packed_elems(K) / 8, PL = (PIECES + 63) / 64;
    static_assert(32 + 32 * (NS - 1) + 24 + (P - JOFF) + 8 <= 96 && 32 - 15 + (P - JOFF) >= 1, "window leaves the padded row");
    uint4 v[PL];
#pragma unroll
    for (int u = 0; u < PL; ++u) {                              // unconditional loads (the last piece again for idle lanes)
        const int p = min(lane + 64 * u, PIECES - 1);
        v[u] = reinterpret_cast<const uint4*>(img)[p];
    }
    for (int off = lane * 16; off < K * FR_ROW_B; off += 64 * 16)
        *reinterpret_cast<uint4*>(scratch + off) = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < PL; ++u) {
        const uint32_t wds[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = (lane + 64 * u) * 8 + q;
            if (e < K * K) {
                const int ky = e / K, t = e - ky * K;
                const uint16_t val = (uint16_t)(wds[q >> 1] >> (16 * (q & 1)));
                uint8_t* row = scratch + ky * FR_ROW_B;
                *reinterpret_cast<uint16_t*>(row + 2 * (32 + t)) = val;
                *reinterpret_cast<uint16_t*>(row + FR_COPY_B + 2 * (31 + t)) = val;
            }
        }
    }
    const int i0 = 32 + 8 * (lane >> 4) - (lane & 15) + (P - JOFF);
    const uint8_t* base = scratch + ((i0 & 1) ? FR_COPY_B + 2 * (i0 - 1) : 2 * i0);
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx) {
            const uint32_t* p = reinterpret_cast<const uint32_t*>(base + ky * FR_ROW_B + sidx * 64);
            bf[ky][sidx] = __builtin_bit_cast(bf16x8, make_uint4(p[0], p[1], p[2], p[3]));
        }
}

__global__ void pack_filter_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int C, int K, int flip) {
    const int PE = packed_elems(K);
    const long total = (long)C * PE;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int e = (int)(i % PE), c = (int)(i / PE);
        float v = 0.f;
        if (e < K * K) v = flip ? w[(long)c * K * K + (K * K - 1 - e)] : w[(long)c * K * K + e];
        out[i] = __builtin_bit_cast(uint16_t, (__bf16)v);
    }
}

struct Item {
    int n0;        // first image of the group
    int G;         // images stacked
    int y0;        // first output row (band) inside each image
    int rows;      // output rows per image in this item
    int x0;        // first output column of the segment
};

// Training-mode BatchNorm (+ ReLU) of the conv's INPUT, applied while the planes are staged (forward only): the 1x1
// conv that produced the input left per-channel partial sums in its epilogue (ppea_pwconv_stats_bf16), every wave
// finalises its channel's statistics in its prologue (same fp64 arithmetic and order as bn_finalize_sums) and stages
// relu(a * z + o) -- the conv_bn_relu between pw1 and the large kernel (replknet_adapter.py:305-308, 182-197) costs no
// launch and no pass over the activation.  Padding stays zero: the reference pads the ACTIVATED tensor.
struct BnIn {
    const float* sums;           // [C][P][2] partial (sum, sum of squares); nullptr: no BatchNorm
    int P;
    float count, eps, momentum;
    const float *gamma, *beta;
    float *running_mean, *running_var, *mean_out, *invstd_out;
};

// This channel's BatchNorm as y = a * x + o from the producer's partial sums (the arithmetic and order of bn_finalize_sums,
// every lane of the wave gets the same values); `writer`: this wave stores the saved statistics and updates the running ones.
__device__ __forceinline__ void bn_channel_affine(const BnIn& bn, int c, int lane, bool writer, float& a, float& o) {
    const float2* sp = reinterpret_cast<const float2*>(bn.sums) + (long)c * bn.P;
    double ds = 0.0, dq = 0.0;
#pragma unroll 8
    for (int i = lane; i < bn.P; i += 64) {
        const float2 v = sp[i];
        ds += (double)v.x; dq += (double)v.y;
    }
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) { ds += __shfl_xor(ds, k, WAVE); dq += __shfl_xor(dq, k, WAVE); }
    const double mean = ds / (double)bn.count;
    double var = dq / (double)bn.count - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float invstd = rsqrtf((float)var + bn.eps);
    a = bn.gamma[c] * invstd;
    o = bn.beta[c] - (float)mean * a;
    if (lane == 0 && writer) {                                   // one writer per channel
        bn.mean_out[c] = (float)mean;
        bn.invstd_out[c] = invstd;
        if (bn.running_mean != nullptr) {
            const float unbiased = (float)(var * (double)bn.count / fmax((double)bn.count - 1.0, 1.0));
            bn.running_mean[c] = (1.f - bn.momentum) * bn.running_mean[c] + bn.momentum * (float)mean;
            bn.running_var[c] = (1.f - bn.momentum) * bn.running_var[c] + bn.momentum * unbiased;
        }
    }
}

__device__ __forceinline__ uint32_t bnrelu2(uint32_t v, float a, float o) {
    const float lo = fmaxf(a * __uint_as_float(v << 16) + o, 0.f), hi = fmaxf(a * __uint_as_float(v & 0xffff0000u) + o, 0.f);
    return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)lo) | ((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)hi) << 16);
}
template <bool BN>
__device__ __forceinline__ uint4 bn_piece(uint4 v, float a, float o) {
    if constexpr (BN) return make_uint4(bnrelu2(v.x, a, o), bnrelu2(v.y, a, o), bnrelu2(v.z, a, o), bnrelu2(v.w, a, o));
    else return v;
}

// Stage G planes' rows [y0 - P, y0 + rows + P) x cols [x0 - JOFF, x0 - JOFF + WL) into LDS (bf16, zeros
// outside).  Lane -> (row within a block of 64/CG rows, 16-byte column group); no divisions in the loop.
template <int K, int NSEG, bool BN = false>
__device__ __forceinline__ void stage_planes(uint8_t* tile, const uint16_t* __restrict__ src, const Item& it,
                                             int C, int c, int H, int W, int lane, float bn_a = 1.f, float bn_o = 0.f) {
    using GE = Geo<K>;
    constexpr int CG = Seg<K, NSEG>::WL / 8;                   // 16-byte groups per staged row
    constexpr int RPI = 64 / CG;                               // rows per wave iteration
    constexpr int STRIDE_B = Seg<K, NSEG>::STRIDE;
    const int r_in = lane / CG, cg = lane - r_in * CG;
    if (r_in >= RPI) return;
    const int rows_l = it.rows + K - 1;                        // LDS rows per image
    const int gx = it.x0 - GE::JOFF + cg * 8;
    const bool vec_ok = ((W & 7) == 0) && gx >= 0 && gx + 8 <= W;
    const bool any_col = gx + 8 > 0 && gx < W;
    constexpr int U = 8;                                       // row pieces in flight per lane
    for (int g = 0; g < it.G; ++g) {
        const uint16_t* plane = src + ((long)(it.n0 + g) * C + c) * (long)H * W;
        uint8_t* dst0 = tile + ((long)g * rows_l + r_in) * STRIDE_B + cg * 16;
        const int gy0 = it.y0 - GE::P + r_in;
        // fast path: every global load is unconditional and in bounds (masked pieces re-read the tensor's first
        // 16 bytes -- NOT the plane's: a 2x3 plane is 12 bytes -- and are zeroed at the LDS store), U of them in flight before the first store -- a load under a
        // branch would make the compiler drain vmcnt in every iteration, one exposed memory latency per piece.
        for (int rb = r_in; rb < rows_l; rb += RPI * U) {
            uint4 v[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int gy = gy0 + (rb - r_in) + u * RPI;
                ok[u] = vec_ok && gy >= 0 && gy < H && (rb + u * RPI) < rows_l;
                v[u] = *reinterpret_cast<const uint4*>(ok[u] ? plane + (long)gy * W + gx : src);   // src: 16 B always in bounds
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (rb + u * RPI < rows_l) {
                    const uint4 t = bn_piece<BN>(v[u], bn_a, bn_o);
                    const uint4 w = make_uint4(ok[u] ? t.x : 0u, ok[u] ? t.y : 0u, ok[u] ? t.z : 0u, ok[u] ? t.w : 0u);
                    *reinterpret_cast<uint4*>(dst0 + (long)(rb - r_in + u * RPI) * STRIDE_B) = w;
                }
            }
        }
        // slow path, only for column groups that straddle the left / right image edge (or W % 8 != 0)
        if (!vec_ok && any_col) {
            uint8_t* dst = dst0;
            int gy = gy0;
            for (int r = r_in; r < rows_l; r += RPI, gy += RPI, dst += RPI * STRIDE_B) {
                if (gy < 0 || gy >= H) continue;
                const uint16_t* rowp = plane + (long)gy * W;
                uint16_t e[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) e[k] = (gx + k >= 0 && gx + k < W) ? rowp[gx + k] : (uint16_t)0;
                uint4 v;
                v.x = e[0] | ((uint32_t)e[1] << 16); v.y = e[2] | ((uint32_t)e[3] << 16);
                v.z = e[4] | ((uint32_t)e[5] << 16); v.w = e[6] | ((uint32_t)e[7] << 16);
                if constexpr (BN) {                              // columns outside the image stay zero (padding)
                    const uint4 t = bn_piece<true>(v, bn_a, bn_o);
                    const bool i0 = gx >= 0 && gx < W, i1 = gx + 1 >= 0 && gx + 1 < W, i2 = gx + 2 >= 0 && gx + 2 < W,
                               i3 = gx + 3 >= 0 && gx + 3 < W, i4 = gx + 4 >= 0 && gx + 4 < W, i5 = gx + 5 >= 0 && gx + 5 < W,
                               i6 = gx + 6 >= 0 && gx + 6 < W, i7 = gx + 7 >= 0 && gx + 7 < W;
                    v.x = (i0 ? t.x & 0xffffu : 0u) | (i1 ? t.x & 0xffff0000u : 0u);
                    v.y = (i2 ? t.y & 0xffffu : 0u) | (i3 ? t.y & 0xffff0000u : 0u);
                    v.z = (i4 ? t.z & 0xffffu : 0u) | (i5 ? t.z & 0xffff0000u : 0u);
                    v.w = (i6 ? t.w & 0xffffu : 0u) | (i7 ? t.w & 0xffff0000u : 0u);
                }
                *reinterpret_cast<uint4*>(dst) = v;
            }
        }
    }
}

// ---- prefetched staging -------------------------------------------------------------------------------
// When an item covers whole planes (one band) the halo rows above / below every staged image are zero for
// EVERY item of the wave: they are cleared once and never written again, and an item stages only the rows
// that exist in the image -- at most SU 16-byte pieces per lane.  Those pieces are loaded into registers
// one item AHEAD (the loads are in flight under the previous item's MFMAs) and written to LDS when the tile
// region is free again.  Needs W % 8 == 0 (every column group is wholly inside or wholly outside the image).
constexpr int SU = 16;

template <int K, int NSEG>
__device__ __forceinline__ int stage_pieces(const Item& it, int H) {       // per-lane slots the item needs
    constexpr int RPI = 64 / (Seg<K, NSEG>::WL / 8);
    const int lo = max(0, it.y0 - Geo<K>::P), hi = min(H, it.y0 + it.rows + Geo<K>::P);
    return it.G * ((hi - lo + RPI - 1) / RPI);
}

template <int K, int NSEG>
__device__ __forceinline__ void stage_load(uint4 (&v)[SU], const uint16_t* __restrict__ src, const Item& it, int C,
                                           int c, int H, int W, int lane) {
    using GE = Geo<K>;
    constexpr int CG = Seg<K, NSEG>::WL / 8, RPI = 64 / CG;
    const int r_in = lane / CG, cg = lane - r_in * CG;
    const int gx = it.x0 - GE::JOFF + cg * 8;
    const bool col_ok = r_in < RPI && gx >= 0 && gx + 8 <= W;
    const int lo = max(0, it.y0 - GE::P), hi = min(H, it.y0 + it.rows + GE::P);
    const int ppv = (hi - lo + RPI - 1) / RPI;
    const uint16_t* base = src + ((long)it.n0 * C + c) * (long)H * W;
    int g = 0, j = 0;
#pragma unroll
    for (int u = 0; u < SU; ++u) {
        const int gy = lo + j * RPI + r_in;
        const bool ok = col_ok && g < it.G && gy < hi;
        v[u] = *reinterpret_cast<const uint4*>(ok ? base + ((long)g * C * H + gy) * W + gx : src);
        if (++j == ppv) { j = 0; ++g; }
    }
}

template <int K, int NSEG, bool BN = false>
__device__ __forceinline__ void stage_store(const uint4 (&v)[SU], uint8_t* tile, const Item& it, int H, int W,
                                            int lane, float bn_a = 1.f, float bn_o = 0.f) {
    using GE = Geo<K>;
    constexpr int CG = Seg<K, NSEG>::WL / 8, RPI = 64 / CG;
    constexpr int STRIDE_B = Seg<K, NSEG>::STRIDE;
    const int r_in = lane / CG, cg = lane - r_in * CG;
    const int gx = it.x0 - GE::JOFF + cg * 8;
    const bool col_ok = gx >= 0 && gx + 8 <= W;
    const int lo = max(0, it.y0 - GE::P), hi = min(H, it.y0 + it.rows + GE::P);
    const int ppv = (hi - lo + RPI - 1) / RPI;
    const int rows_l = it.rows + K - 1;
    int g = 0, j = 0;
#pragma unroll
    for (int u = 0; u < SU; ++u) {
        const int gy = lo + j * RPI + r_in;
        if (r_in < RPI && g < it.G && gy < hi) {
            const uint4 t = bn_piece<BN>(v[u], bn_a, bn_o);
            const uint4 w = make_uint4(col_ok ? t.x : 0u, col_ok ? t.y : 0u, col_ok ? t.z : 0u, col_ok ? t.w : 0u);
            *reinterpret_cast<uint4*>(tile + ((long)g * rows_l + (gy - (it.y0 - GE::P))) * STRIDE_B + cg * 16) = w;
        }
        if (++j == ppv) { j = 0; ++g; }
    }
}

// ---- hand-counted LDS pipeline ---------------------------------------------------------------------
// hipcc sinks every ds_read next to its MFMA (one read in flight, ~100 exposed cycles per MFMA with one
// wave per SIMD).  The A-fragment reads are therefore inline asm (invisible to the scheduler, issued in
// program order) and their completion is counted by hand: LDS returns in order, so after issuing the
// next group's n reads `s_waitcnt lgkmcnt(n)` means "the current group has landed".  The wait statement
// names the current group's registers as in/out operands, which ties every consuming MFMA to it
// (cdna_hip_programming.md 5.7, form (ii)).
typedef __attribute__((address_space(3))) const uint8_t* lds_cptr_t;
__device__ __forceinline__ uint32_t lds_addr(const uint8_t* p) { return (uint32_t)(uintptr_t)(lds_cptr_t)p; }

template <int OFF>
__device__ __forceinline__ void ds_read128_async(bf16x8& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}

// Software pipeline of one tile: the A fragments stream through a ring of RING registers, the read of
// fragment I + DIST is issued right after MFMA I, and every wait names exactly how many younger reads may
// still be in flight (LDS reads of one wave return in order; lgkmcnt is a 4-bit counter, so DIST <= 15).
// Two accumulators alternate so that consecutive MFMAs never wait on each other's result.
constexpr int DIST = 12, RING = 13;

template <int KK, int NS, int ROW0, int STRIDE_B, int I>
__device__ __forceinline__ void issue_read(bf16x8 (&a)[RING], uint32_t abase) {
    if constexpr (I < KK * NS) {
        constexpr int ky = I / NS, sidx = I % NS;
        ds_read128_async<(ky + ROW0) * STRIDE_B + sidx * 64>(a[I % RING], abase);
    }
}

template <int KK, int NS, int ROW0, int STRIDE_B, int... Is>
__device__ __forceinline__ void issue_first(bf16x8 (&a)[RING], uint32_t abase, std::integer_sequence<int, Is...>) {
    (issue_read<KK, NS, ROW0, STRIDE_B, Is>(a, abase), ...);
}

template <int KK, int NS, int ROW0, int STRIDE_B, int I>
__device__ __forceinline__ void mac_step(f32x4 (&acc)[2], bf16x8 (&a)[RING], uint32_t abase,
                                         const bf16x8 (&bf)[KK][NS]) {
    if constexpr (I < KK * NS) {
        constexpr int TOTAL = KK * NS;
        constexpr int younger = (TOTAL - 1 - I) < (DIST - 1) ? (TOTAL - 1 - I) : (DIST - 1);
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a[I % RING]) : "i"(younger));
        acc[I & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[I % RING], bf[I / NS][I % NS], acc[I & 1], 0, 0, 0);
        issue_read<KK, NS, ROW0, STRIDE_B, I + DIST>(a, abase);
        mac_step<KK, NS, ROW0, STRIDE_B, I + 1>(acc, a, abase, bf);
    }
}

// acc += sum over filter rows of A(rows, chunk) * B(ky, chunk) for one 16x16 tile.
// `abase` = LDS byte address of (first input row of this lane's output row, first chunk column of
// the tile) + 16 * (lane >> 4); ROW0 = extra row offset (small kernel inside the big halo).
template <int KK, int NS, int ROW0, int STRIDE_B>
__device__ __forceinline__ void tile_mac(f32x4& acc, uint32_t abase, const bf16x8 (&bf)[KK][NS]) {
    bf16x8 a[RING];
    f32x4 acc2[2] = {acc, {0.f, 0.f, 0.f, 0.f}};
    issue_first<KK, NS, ROW0, STRIDE_B>(a, abase, std::make_integer_sequence<int, DIST>{});
    mac_step<KK, NS, ROW0, STRIDE_B, 0>(acc2, a, abase, bf);
    acc = acc2[0] + acc2[1];
}

// ---- cross-tile stream -----------------------------------------------------------------------------------
// One tile = TOTAL = K*NS (+KS) steps, each one A read + one MFMA.  The read for step I + DIST is issued right
// after MFMA I; for the last DIST steps of a tile those are the FIRST reads of the next tile (into the other
// ring), so the LDS pipe never drains between tiles and a tile's epilogue (convert + stores) runs under the
// next tile's reads.  Step I < K*NS: big filter row I / NS, chunk I % NS at `big`; the KS small-filter steps
// read at `small` (MODE 0: own accumulator / second output; MODE 1: second input, same accumulator).
struct TileBase { uint32_t big, small; };

template <int K, int KS, int NS, int STRIDE_B, int SM_ROW0, int J>
__device__ __forceinline__ void stream_read(bf16x8& slot, const TileBase& tb) {
    if constexpr (J < K * NS) ds_read128_async<(J / NS) * STRIDE_B + (J % NS) * 64>(slot, tb.big);
    else ds_read128_async<(J - K * NS + SM_ROW0) * STRIDE_B>(slot, tb.small);
}

template <int K, int KS, int NS, int STRIDE_B, int SM_ROW0, int... Is>
__device__ __forceinline__ void stream_first(bf16x8 (&ring)[RING], const TileBase& tb, std::integer_sequence<int, Is...>) {
    (stream_read<K, KS, NS, STRIDE_B, SM_ROW0, Is>(ring[Is % RING], tb), ...);
}

// `epi`: the PREVIOUS tile's epilogue (convert + stores), run EPI_AT steps into this tile's stream: by then the previous
// tile's last MFMAs have retired (no wait on the accumulator) and the conversions / address arithmetic / stores issue in
// the shadow of this tile's MFMAs instead of between two tiles.
constexpr int EPI_AT = 6;

template <int K, int KS, int MODE, int NS, int STRIDE_B, int SM_ROW0, int I, typename Epi>
__device__ __forceinline__ void stream_step(f32x4 (&accb)[2], f32x4& accs, bf16x8 (&cur)[RING], bf16x8 (&nxt)[RING],
                                            const TileBase& tc, const TileBase& tn, const bf16x8 (&bfb)[K][NS],
                                            const bf16x8 (&bfs)[(KS > 0 ? KS : 1)][1], Epi&& epi) {
    constexpr int TOTAL = K * NS + KS;
    if constexpr (I < TOTAL) {
        if constexpr (I == EPI_AT) epi();
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(cur[I % RING]) : "i"(DIST - 1));
        if constexpr (I < K * NS)
            accb[I & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfb[I / NS][I % NS], cur[I % RING], accb[I & 1], 0, 0, 0);
        else if constexpr (MODE == 0)
            accs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfs[I - K * NS][0], cur[I % RING], accs, 0, 0, 0);
        else
            accb[I & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfs[I - K * NS][0], cur[I % RING], accb[I & 1], 0, 0, 0);
        constexpr int J = I + DIST;
        if constexpr (J < TOTAL) stream_read<K, KS, NS, STRIDE_B, SM_ROW0, J>(cur[J % RING], tc);
        else stream_read<K, KS, NS, STRIDE_B, SM_ROW0, J - TOTAL>(nxt[(J - TOTAL) % RING], tn);
        stream_step<K, KS, MODE, NS, STRIDE_B, SM_ROW0, I + 1>(accb, accs, cur, nxt, tc, tn, bfb, bfs, epi);
    }
}

// ---- window sharing between vertically stacked tiles (WS) ----------------------------------------------------------
// A 48-row item is a column of MTL = 3 stacked 16-row tiles per column tile.  The A fragment of (tile t, filter row ky)
// is the 16-row window of the staged plane that starts at row 16 t + ky: window w = 16 t + ky serves every tile t with
// 0 <= w - 16 t < K, i.e. up to two tiles.  One COLUMN is therefore one stream of (16 (MTL - 1) + K) x NS window reads
// (+ MTL x KS small-kernel reads) feeding MTL x (K x NS + KS) MFMAs -- 141 LDS reads per 201 MFMAs at k = 31 instead of
// 201, which is what the LDS-bound stream (four waves x 1 KB per ~20 cycles = 80 % of the 256 B/clk) needed.
// The schedule is a compile-time table; tile t's accumulators are final after window 16 t + K - 1 and are converted and
// stored EPI_DELAY steps later, inside the same stream; the last tile's leave with the column and are stored EPI_AT
// steps into the next column's stream (its accumulator slot alternates by column parity).
constexpr int EPI_DELAY = 6;
// a shallower read-ahead than the per-tile stream's: a window feeds up to two MFMAs, and the column's accumulators (up to
// 12 f32x4) need the registers
constexpr int CT_DIST = 8, CT_RING = 9;

template <int K, int KS, int NS, int MTL>
struct CtSched {
    static constexpr int WN = 16 * (MTL - 1) + K;
    static constexpr int TOTAL = WN * NS + MTL * KS;
    static constexpr int SM_ROW0 = K / 2 - (KS > 0 ? KS : 5) / 2;
    struct Step { int kind, w, s, t, ky; };                 // kind 0: big window (w, chunk s); 1: small (tile t, row ky)
    static constexpr Step at(int J) {
        int j = 0;
        for (int w = 0; w < WN; ++w) {
            for (int s = 0; s < NS; ++s) { if (j == J) return {0, w, s, 0, 0}; ++j; }
            if (KS > 0) {
                const int r = w - (SM_ROW0 + KS - 1);
                if (r >= 0 && r % 16 == 0 && r / 16 < MTL)
                    for (int ky = 0; ky < KS; ++ky) { if (j == J) return {1, 0, 0, r / 16, ky}; ++j; }
            }
        }
        return {-1, 0, 0, 0, 0};
    }
    static constexpr int final_idx(int t) {                  // the step after which tile t receives no more MFMAs
        int j = 0, last = 0;
        for (int w = 0; w < WN; ++w) {
            for (int s = 0; s < NS; ++s) { if (w == 16 * t + K - 1) last = j; ++j; }
            if (KS > 0) {
                const int r = w - (SM_ROW0 + KS - 1);
                if (r >= 0 && r % 16 == 0 && r / 16 < MTL) j += KS;
            }
        }
        return last;
    }
};

template <int K, int KS, int NS, int MTL, int STRIDE_B, int J>
__device__ __forceinline__ void ct_read(bf16x8& slot, const TileBase& tb) {
    using S = CtSched<K, KS, NS, MTL>;
    constexpr typename S::Step st = S::at(J);
    if constexpr (st.kind == 0) ds_read128_async<st.w * STRIDE_B + st.s * 64>(slot, tb.big);
    else ds_read128_async<(16 * st.t + S::SM_ROW0 + st.ky) * STRIDE_B>(slot, tb.small);
}

template <int K, int KS, int NS, int MTL, int STRIDE_B, int... Is>
__device__ __forceinline__ void ct_first(bf16x8 (&ring)[CT_RING], const TileBase& tb, std::integer_sequence<int, Is...>) {
    (ct_read<K, KS, NS, MTL, STRIDE_B, Is>(ring[Is % CT_RING], tb), ...);
}

// accb[MTL + 1][NS], accs[MTL + 1]: slot MTL is the last tile's alternate (columns of odd parity)
template <int K, int KS, int MODE, int NS, int MTL, int STRIDE_B, int PAR, int I, typename EpiT, typename EpiPrev>
__device__ __forceinline__ void ct_step(f32x4 (&accb)[MTL + 1][NS], f32x4 (&accs)[MTL + 1], bf16x8 (&cur)[CT_RING],
                                        bf16x8 (&nxt)[CT_RING], const TileBase& tc, const TileBase& tn,
                                        const bf16x8 (&bfb)[K][NS], const bf16x8 (&bfs)[(KS > 0 ? KS : 1)][1],
                                        EpiT&& epi_tile, EpiPrev&& epi_prev) {
    using S = CtSched<K, KS, NS, MTL>;
    if constexpr (I < S::TOTAL) {
        if constexpr (I == EPI_AT) epi_prev();
#define PPEA_CT_EPI(T_)                                                                                     \
        if constexpr (T_ < MTL - 1) { if constexpr (I == S::final_idx(T_) + EPI_DELAY) epi_tile(std::integral_constant<int, T_>{}); }
        PPEA_CT_EPI(0) PPEA_CT_EPI(1) PPEA_CT_EPI(2)
#undef PPEA_CT_EPI
        constexpr typename S::Step st = S::at(I);
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(cur[I % CT_RING]) : "i"(CT_DIST - 1));
        if constexpr (st.kind == 0) {
#define PPEA_CT_MAC(T_)                                                                                     \
            if constexpr (T_ < MTL && st.w - 16 * T_ >= 0 && st.w - 16 * T_ < K) {                          \
                constexpr int sl = (T_ == MTL - 1 && PAR) ? MTL : T_;                                       \
                accb[sl][st.s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfb[st.w - 16 * T_][st.s], cur[I % CT_RING],     \
                                                                          accb[sl][st.s], 0, 0, 0);         \
            }
            PPEA_CT_MAC(0) PPEA_CT_MAC(1) PPEA_CT_MAC(2)
#undef PPEA_CT_MAC
        } else {
            constexpr int sl = (st.t == MTL - 1 && PAR) ? MTL : st.t;
            if constexpr (MODE == 0)
                accs[sl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfs[st.ky][0], cur[I % CT_RING], accs[sl], 0, 0, 0);
            else
                accb[sl][st.ky % NS] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfs[st.ky][0], cur[I % CT_RING],
                                                                                accb[sl][st.ky % NS], 0, 0, 0);
        }
        constexpr int J = I + CT_DIST;
        if constexpr (J < S::TOTAL) ct_read<K, KS, NS, MTL, STRIDE_B, J>(cur[J % CT_RING], tc);
        else ct_read<K, KS, NS, MTL, STRIDE_B, J - S::TOTAL>(nxt[(J - S::TOTAL) % CT_RING], tn);
        ct_step<K, KS, MODE, NS, MTL, STRIDE_B, PAR, I + 1>(accb, accs, cur, nxt, tc, tn, bfb, bfs, epi_tile, epi_prev);
    }
}

// The MFMAs take the Toeplitz fragment as the A operand and the input rows as B (both fragment layouts are
// "index = lane & 15, k = 8 * (lane >> 4) + j", so the registers are the same either way): the accumulator then
// holds the TRANSPOSED tile -- lane (y = lane & 15, g = lane >> 4) owns columns 4g..4g+3 of image row y, four
// consecutive bf16 = ONE 8-byte store per lane instead of four 2-byte ones.
// Per M-tile: element offset of this lane's output row (-1 = padding row).
struct RowOffs { int off; };

__device__ __forceinline__ RowOffs row_offsets(const Item& it, int C, int c, int H, int W, int mt, int lane) {
    RowOffs ro;
    const int m = mt * 16 + (lane & 15);                       // stacked output row
    const int g = m / it.rows, y = it.y0 + (m - g * it.rows);
    ro.off = (g < it.G && y < H) ? (((it.n0 + g) * C + c) * H + y) * W : -1;
    return ro;
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {      // v_cvt_pk_bf16_f32 (RNE, NaN kept)
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ void store_tile(uint16_t* __restrict__ dst, const f32x4& acc, const RowOffs& ro, int W,
                                           int xt, int lane) {
    const int col = xt + 4 * (lane >> 4);
    if (ro.off < 0 || col >= W) return;
    uint16_t* p = dst + ro.off + col;
    if (col + 4 <= W && ((ro.off + col) & 3) == 0) {
        *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]));
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (col + i < W) p[i] = __builtin_bit_cast(uint16_t, (__bf16)acc[i]);
    }
}

// Statistics of what store_tile writes (the values AS STORED: rounded to bf16), for the BatchNorm that follows the conv.
__device__ __forceinline__ void tile_stats(const f32x4& acc, const RowOffs& ro, int W, int xt, int lane, float& s, float& q) {
    const int col = xt + 4 * (lane >> 4);
    if (ro.off < 0) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (col + i < W) {
            const float v = __uint_as_float((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)acc[i]) << 16);
            s += v;
            q += v * v;
        }
}

// MODE 0: fwd  (in0 = x; out0 = y_big, out1 = y_small if KS)
// MODE 1: dgrad (in0 = dy_big, in1 = dy_small if KS; out0 = dx), filters flipped
template <int K, int KS, int MODE, int NSEG, bool BN = false, bool WS = false>
__global__ __launch_bounds__(64 * WAVES, 1) void dwconv_mfma_kernel(
    const uint16_t* __restrict__ in0, const uint16_t* __restrict__ in1, const uint16_t* __restrict__ w_big,
    const uint16_t* __restrict__ w_small, uint16_t* __restrict__ out0, uint16_t* __restrict__ out1, int N, int C,
    int H, int W, int G, int band, int bands, int segs, int items_per_channel, int ipw, int wpc,
    long total_waves, int tile_bytes, int region_bytes, float* __restrict__ stats, BnIn bn) {
    static_assert(!BN || MODE == 0, "the fused input BatchNorm is a forward feature");
    using GE = Geo<K>;
    using GS = Geo<(KS > 0 ? KS : 5)>;
    constexpr int STRIDE_B = Seg<K, NSEG>::STRIDE;
    constexpr int NT_IN = (MODE == 1 && KS > 0) ? 2 : 1;
    constexpr int AGPR_FROM = 6;             // big-filter fragments [AGPR_FROM, K*NS) live in AGPRs
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // One LDS region per wave: the wave's input tile(s).  Nothing is shared between waves, so the kernel has no
    // workgroup barrier.
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * WAVES + wave;
    if (wid >= total_waves) return;
    const int c = (int)(wid / wpc);
    const int first_item = (int)(wid - (long)c * wpc) * ipw;
#ifdef DW_PROF
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    PROF_T(t_begin);
    // The Toeplitz fragments of the whole filter: built from the channel's filter through the wave's LDS region (see
    // build_bfrags), registers for the rest of the kernel.  The region is the wave's own and LDS operations of one wave
    // execute in order, so the staging writes further down need no wait.
    uint8_t* tile0 = smem + (long)wave * region_bytes;
    bf16x8 bf_big[K][GE::NS];
    build_bfrags<K, GE::NS>(bf_big, w_big + (long)c * packed_elems(K), tile0, lane);
    // Park most Toeplitz fragments in the accumulator half of the unified register file (MFMA reads its
    // B operand from AGPRs directly); this keeps the arch VGPRs free for A-fragment prefetch.
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int s = 0; s < GE::NS; ++s)
            if (ky * GE::NS + s >= AGPR_FROM) asm volatile("" : "+a"(bf_big[ky][s]));
    bf16x8 bf_small[(KS > 0 ? KS : 1)][1];
    if constexpr (KS > 0)
        build_bfrags<KS, 1>(bf_small, w_small + (long)c * packed_elems(KS), tile0 + K * FR_ROW_B, lane);

    PROF_T(t_frags);
    PROF_ADD(6, t_begin, t_frags);
    // fused input BatchNorm: this channel's statistics from the producer's partial sums (bn_finalize_sums' arithmetic)
    float bn_a = 1.f, bn_o = 0.f;
    if constexpr (BN) bn_channel_affine(bn, c, lane, wid == (long)c * wpc, bn_a, bn_o);

    uint8_t* tile1 = tile0 + tile_bytes;
    constexpr int SM_ROW0 = GE::P - GS::P;                  // small-kernel rows inside the big halo
    constexpr int SM_COLB = (GE::JOFF - GS::JOFF) * 2;      // byte offset of its first chunk

    auto make_item = [&](int item_id, Item& it) -> bool {    // item = (plane group or row band, column segment)
        if (item_id >= items_per_channel || item_id >= first_item + ipw) return false;
        const int seg = item_id % segs;
        const int gb = item_id / segs;
        it.x0 = seg * 16 * NSEG;
        if (G > 1) {                                        // G small planes stacked along M, whole height
            it.n0 = gb * G;
            it.G = min(G, N - it.n0);
            it.y0 = 0;
            it.rows = H;
        } else {                                            // one plane, bands of `band` rows
            it.n0 = gb / bands;
            it.G = 1;
            it.y0 = (gb - it.n0 * bands) * band;
            it.rows = min(band, H - it.y0);
        }
        return true;
    };
    // MODE 0 with `stats`: per-channel partial sums (sum, sum of squares) of both outputs, one entry per wave of the
    // channel -- stats [2][C][wpc][2] -- so that the BatchNorm pair after the conv needs no statistics pass
    float st_sb = 0.f, st_qb = 0.f, st_ss = 0.f, st_qs = 0.f;
    int done = 0;
    Item it;
    bool have = make_item(first_item, it);
    // prefetched staging (see stage_load): whole planes per item, aligned columns, few enough pieces
    const bool fast = have && bands == 1 && (W & 7) == 0 && stage_pieces<K, NSEG>(it, H) * (G > 1 ? G : 1) / it.G <= SU;
    uint4 pre[SU];
    if (fast) {
        for (int off = lane * 16; off < NT_IN * tile_bytes; off += 64 * 16)
            *reinterpret_cast<uint4*>(tile0 + off) = make_uint4(0, 0, 0, 0);
        stage_load<K, NSEG>(pre, in0, it, C, c, H, W, lane);
    }

    PROF_T(t_setup_done);
    PROF_ADD(0, t_begin, t_setup_done);
    while (have) {
        PROF_T(t_item);
        // The tile is wave-private: LDS operations of one wave execute in order, so the staging writes
        // below are ordered after the previous item's (already waited-for) reads and before this item's.
        Item nxt;
        const bool have_next = make_item(first_item + (++done), nxt);
        if (fast) {
            stage_store<K, NSEG, BN>(pre, tile0, it, H, W, lane, bn_a, bn_o);
            if constexpr (NT_IN == 2) {
                uint4 second[SU];
                stage_load<K, NSEG>(second, in1, it, C, c, H, W, lane);
                stage_store<K, NSEG>(second, tile1, it, H, W, lane);
            }
            if (have_next) stage_load<K, NSEG>(pre, in0, nxt, C, c, H, W, lane);   // lands during the MFMAs below
        } else {
            stage_planes<K, NSEG, BN>(tile0, in0, it, C, c, H, W, lane, bn_a, bn_o);
            if constexpr (NT_IN == 2) stage_planes<K, NSEG>(tile1, in1, it, C, c, H, W, lane);
        }
        asm volatile("" ::: "memory");      // compiler fence: staging stores stay above the asm LDS reads
        PROF_T(t_staged);
        PROF_ADD(1, t_item, t_staged);

        const int rows_l = it.rows + K - 1;                 // LDS rows per stacked image (with halo)
        const int ntiles_x = min(NSEG, (W - it.x0 + 15) / 16);
        const int mrows = it.G * it.rows;
        const int ntiles_m = (mrows + 15) / 16;
        // tiles of the item in (mt, nt) order as ONE stream (see stream_step); two rings alternate by tile parity
        auto tile_base = [&](int mt, int nt) -> TileBase {
            // this lane's A row -> first LDS row of its window (clamped for padding rows)
            const int m = min(mt * 16 + (lane & 15), mrows - 1);
            const int g = m / it.rows, y = m - g * it.rows;
            const long aoff = (long)(g * rows_l + y) * STRIDE_B + 16 * (lane >> 4) + nt * 32;
            TileBase tb;
            tb.big = lds_addr(tile0 + aoff);
            tb.small = lds_addr((MODE == 1 ? tile1 : tile0) + aoff) + SM_COLB;
            return tb;
        };
        const int ntiles = ntiles_m * ntiles_x;
        bf16x8 ringA[WS ? CT_RING : RING], ringB[WS ? CT_RING : RING];
        if constexpr (WS) {
            // window sharing (see CtSched): the item is ONE plane band of exactly 48 rows (the host launches this variant
            // only then), processed column by column
            constexpr int MTL = 3, NS = GE::NS;
            f32x4 accb[MTL + 1][NS], accs[MTL + 1];
            RowOffs ro3[MTL];
#pragma unroll
            for (int t = 0; t < MTL; ++t) ro3[t] = row_offsets(it, C, c, H, W, t, lane);
            TileBase tc = tile_base(0, 0);
            ct_first<K, KS, NS, MTL, STRIDE_B>(ringA, tc, std::make_integer_sequence<int, CT_DIST>{});
            int xt_prev = 0;
            bool pend = false;
            auto store3 = [&](const f32x4 (&pb)[NS], const f32x4& ps, const RowOffs& ro, int xt) {
                f32x4 acc = pb[0];
                if constexpr (NS == 2) acc = pb[0] + pb[1];
                store_tile(out0, acc, ro, W, xt, lane);
                if constexpr (MODE == 0 && KS > 0) store_tile(out1, ps, ro, W, xt, lane);
                if constexpr (MODE == 0) {
                    if (stats != nullptr) {
                        tile_stats(acc, ro, W, xt, lane, st_sb, st_qb);
                        if constexpr (KS > 0) tile_stats(ps, ro, W, xt, lane, st_ss, st_qs);
                    }
                }
            };
            auto one_ct = [&](auto par_c, bf16x8 (&cur)[CT_RING], bf16x8 (&nxt_ring)[CT_RING], int nt, bool last) {
                constexpr int PAR = decltype(par_c)::value;
                constexpr int LS = PAR ? MTL : MTL - 1, PS = PAR ? MTL - 1 : MTL;   // this / the previous column's last-tile slot
                const TileBase tn = last ? tc : tile_base(0, nt + 1);
#pragma unroll
                for (int t = 0; t < MTL - 1; ++t) {
#pragma unroll
                    for (int q = 0; q < NS; ++q) accb[t][q] = {0.f, 0.f, 0.f, 0.f};
                    accs[t] = {0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int q = 0; q < NS; ++q) accb[LS][q] = {0.f, 0.f, 0.f, 0.f};
                accs[LS] = {0.f, 0.f, 0.f, 0.f};
                const int xt = it.x0 + nt * 16;
                PROF_T(t_mac0);
                ct_step<K, KS, MODE, NS, MTL, STRIDE_B, PAR, 0>(
                    accb, accs, cur, nxt_ring, tc, tn, bf_big, bf_small,
                    [&](auto tcst) { constexpr int T = decltype(tcst)::value; store3(accb[T], accs[T], ro3[T], xt); },
                    [&]() { if (pend) store3(accb[PS], accs[PS], ro3[MTL - 1], xt_prev); });
                PROF_T(t_mac1);
                PROF_ADD(2, t_mac0, t_mac1);
                xt_prev = xt;
                pend = true;
                tc = tn;
            };
            for (int nt = 0; nt < ntiles_x; nt += 2) {
                one_ct(std::integral_constant<int, 0>{}, ringA, ringB, nt, nt + 1 >= ntiles_x);
                if (nt + 1 < ntiles_x) one_ct(std::integral_constant<int, 1>{}, ringB, ringA, nt + 1, nt + 2 >= ntiles_x);
            }
            if (pend) {                                          // the last column's last tile
                if (ntiles_x & 1) store3(accb[MTL - 1], accs[MTL - 1], ro3[MTL - 1], xt_prev);
                else store3(accb[MTL], accs[MTL], ro3[MTL - 1], xt_prev);
            }
        } else {
        int mt = 0, nt = 0;
        TileBase tc = tile_base(0, 0);
        stream_first<K, KS, GE::NS, STRIDE_B, SM_ROW0>(ringA, tc, std::make_integer_sequence<int, DIST>{});
        RowOffs ro = row_offsets(it, C, c, H, W, 0, lane);
        // two accumulator sets alternate by tile parity (like the rings); the set a tile leaves behind is converted and
        // stored EPI_AT steps into the next tile's stream (stream_step)
        f32x4 accA[2], accB[2], accsA, accsB;
        RowOffs ro_pend = ro;
        int xt_pend = 0;
        bool pend = false;
        auto epilogue = [&](const f32x4 (&pb)[2], const f32x4& ps) {
            const f32x4 acc = pb[0] + pb[1];
            store_tile(out0, acc, ro_pend, W, xt_pend, lane);
            if constexpr (MODE == 0 && KS > 0) store_tile(out1, ps, ro_pend, W, xt_pend, lane);
            if constexpr (MODE == 0) {
                if (stats != nullptr) {                      // wave-uniform
                    tile_stats(acc, ro_pend, W, xt_pend, lane, st_sb, st_qb);
                    if constexpr (KS > 0) tile_stats(ps, ro_pend, W, xt_pend, lane, st_ss, st_qs);
                }
            }
        };
        auto one_tile = [&](bf16x8 (&cur)[RING], bf16x8 (&nxt_ring)[RING], f32x4 (&mine)[2], f32x4& mine_s,
                            const f32x4 (&other)[2], const f32x4& other_s, int t) {
            int mt2 = mt, nt2 = nt + 1;
            if (nt2 == ntiles_x) { nt2 = 0; ++mt2; }
            const bool last = t + 1 >= ntiles;
            const TileBase tn = last ? tc : tile_base(mt2, nt2);        // last tile: harmless re-reads, drained below
            mine[0] = {0.f, 0.f, 0.f, 0.f};
            mine[1] = {0.f, 0.f, 0.f, 0.f};
            mine_s = {0.f, 0.f, 0.f, 0.f};
            PROF_T(t_mac0);
            stream_step<K, KS, MODE, GE::NS, STRIDE_B, SM_ROW0, 0>(mine, mine_s, cur, nxt_ring, tc, tn, bf_big, bf_small,
                                                                   [&]() { if (pend) epilogue(other, other_s); });
            PROF_T(t_mac1);
            PROF_ADD(2, t_mac0, t_mac1);
            ro_pend = ro;
            xt_pend = it.x0 + nt * 16;
            pend = true;
            if (mt2 != mt && !last) ro = row_offsets(it, C, c, H, W, mt2, lane);
            mt = mt2; nt = nt2; tc = tn;
        };
        for (int t = 0; t < ntiles; t += 2) {
            one_tile(ringA, ringB, accA, accsA, accB, accsB, t);
            if (t + 1 < ntiles) one_tile(ringB, ringA, accB, accsB, accA, accsA, t + 1);
        }
        PROF_T(t_epi0);
        if (pend) {                                              // the item's last tile
            if (ntiles & 1) epilogue(accA, accsA); else epilogue(accB, accsB);
        }
        PROF_T(t_epi);
        PROF_ADD(3, t_epi0, t_epi);
        }                                                        // !WS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the look-ahead reads of the last tile
        PROF_T(t_item_end);
        PROF_ADD(4, t_item, t_item_end);
        asm volatile("" ::: "memory");      // next item's staging stores stay below this item's LDS reads
        it = nxt;
        have = have_next;
    }
    if constexpr (MODE == 0) {
        if (stats != nullptr) {
            const float sb = wave_sum(st_sb), qb = wave_sum(st_qb), ss = wave_sum(st_ss), qs = wave_sum(st_qs);
            if (lane == 0) {
                const long e = ((long)c * wpc + (wid - (long)c * wpc)) * 2;
                stats[e] = sb; stats[e + 1] = qb;
                if constexpr (KS > 0) { stats[(long)C * wpc * 2 + e] = ss; stats[(long)C * wpc * 2 + e + 1] = qs; }
            }
        }
    }
#ifdef DW_PROF
    PROF_T(t_end);
    prof[5] = t_end - t_begin;
    if (lane == 0 && wid < 4096)
        for (int i = 0; i < 8; ++i) g_dw_prof[wid][i] = prof[i];
#endif
}

// ---- batch-major variant for small planes (stages 2 / 3: 12 x 40 and 6 x 20 maps) ----------------------------------------
// On a plane that is shorter than the filter the row-band formulation above wastes most of its work: G planes stacked along
// M carry K - 1 zero halo rows each (12 image rows in 38 LDS rows at k = 27), 56 % of the MFMAs multiply zeros, every MFMA
// needs its own A window from LDS (the stream is LDS-bound), and a channel's 3 items do not divide by its 2 waves.  Here
// the M dimension of the MFMA is the BATCH instead: for one channel, one input row y_in and one 32-column chunk, the A
// fragment is [16 images x 32 input columns]; multiplied by the Toeplitz fragment of filter row ky = y_in - y_out + P it
// contributes to output row y_out of all 16 images at once.  A wave owns NY output rows (NY accumulators per column tile,
// + NY for the 5x5 branch) and walks y_in: ONE A read serves up to NY MFMAs, only (y_in, y_out) pairs inside the plane are
// issued, and a wave needs only the filter rows with |y_in - y_out| <= P that its rows can meet (17 of 27 at H = 12).
// k = 27 on [12,512,12,40]: 522 MFMAs and 108 LDS reads per wave instead of 1 062 / 1 062 on the slower wave.
// The H / NY waves of a channel share its staged tile  [y_in][image][columns]  (row stride as above: conflict-free b128 reads
// for 16 images x 4 k-groups); a workgroup is 4 waves = 4 NY / H channels, one group of up to 16 images.
template <int K, int NTX>
struct BmGeo {
    static constexpr int WL = 16 * (NTX - 1) + 32 * Geo<K>::NS;
    static constexpr int STRIDE = lds_stride_bytes(WL);
    static constexpr int CG = WL / 8;
};

// Stage the channel's rows: piece (y, image, VEC-element column group) -> tile[(y * nimg + image) * STRIDE + 2 VEC * group];
// x0: first output column of the staged segment; `sub` / `nsub`: this lane's index among the lanes that share the tile.  VEC = 8 (16-byte pieces) needs W % 8 == 0,
// VEC = 4 (8-byte pieces: the 6 x 20 maps) W % 4 == 0 -- a piece is wholly inside the plane or wholly padding.
template <int K, int NTX, int H, int VEC, int CPW_>
struct BmStage {
    using B = BmGeo<K, NTX>;
    using V = std::conditional_t<VEC == 8, uint4, uint2>;
    // pieces in (image, row, column group) order: consecutive lanes walk a plane's rows (contiguous in memory), and the
    // decode divides by constants only; U pieces in flight per lane before the first LDS store (2 round trips on the
    // 48 x 160 maps, one on the small ones)
    static constexpr int CGV = B::WL / VEC;
    static constexpr int PPL = (H * 12 * CGV + 64 * (4 / CPW_) - 1) / (64 * (4 / CPW_));   // pieces per lane at 12 images
    static constexpr int U = PPL <= 14 ? PPL : (PPL + 1) / 2;

    __device__ static __forceinline__ void load(V (&v)[U], const uint16_t* __restrict__ src, int n0, int nimg, int C, int c,
                                                int W, int x0, int i0, int nsub) {
        const int total = H * nimg * CGV;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = i0 + u * nsub;
            const int pc = idx % CGV, r2 = idx / CGV, y = r2 % H, n = r2 / H;
            const int gx = x0 + VEC * pc - Geo<K>::JOFF;
            const bool ok = idx < total && gx >= 0 && gx + VEC <= W;
            v[u] = *reinterpret_cast<const V*>(ok ? src + (((long)(n0 + n) * C + c) * H + y) * W + gx : src);
        }
    }
    template <bool BN>
    __device__ static __forceinline__ void store(const V (&v)[U], uint8_t* tile, int nimg, int W, int x0, int i0, int nsub,
                                                 float bn_a, float bn_o) {
        const int total = H * nimg * CGV;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = i0 + u * nsub;
            const int pc = idx % CGV, r2 = idx / CGV, y = r2 % H, n = r2 / H;
            const int gx = x0 + VEC * pc - Geo<K>::JOFF;
            const bool ok = gx >= 0 && gx + VEC <= W;
            if (idx < total) {
                uint8_t* dst = tile + (y * nimg + n) * B::STRIDE + pc * (2 * VEC);
                if constexpr (VEC == 8) {
                    const uint4 t = bn_piece<BN>(v[u], bn_a, bn_o);
                    *reinterpret_cast<uint4*>(dst) = make_uint4(ok ? t.x : 0u, ok ? t.y : 0u, ok ? t.z : 0u, ok ? t.w : 0u);
                } else {
                    uint2 t = v[u];
                    if constexpr (BN) t = make_uint2(bnrelu2(t.x, bn_a, bn_o), bnrelu2(t.y, bn_a, bn_o));
                    *reinterpret_cast<uint2*>(dst) = make_uint2(ok ? t.x : 0u, ok ? t.y : 0u);
                }
            }
        }
    }
};

// Stage the channel's rows: piece (y, image, VEC-element column group) -> tile[(y * nimg + image) * STRIDE + 2 VEC * group];
// x0: first output column of the staged segment; `sub` / `nsub`: this lane's index among the lanes that share the tile.
// VEC = 8 (16-byte pieces) needs W % 8 == 0, VEC = 4 (8-byte pieces: the 6 x 20 maps) W % 4 == 0 -- a piece is wholly inside
// the plane or wholly padding.
template <int K, int NTX, int H, bool BN, int VEC, int CPW_>
__device__ __forceinline__ void stage_bm(uint8_t* tile, const uint16_t* __restrict__ src, int n0, int nimg, int C, int c,
                                         int W, int x0, int sub, int nsub, float bn_a, float bn_o) {
    using St = BmStage<K, NTX, H, VEC, CPW_>;
    const int total = H * nimg * St::CGV;
    for (int i0 = sub; i0 < total; i0 += nsub * St::U) {
        typename St::V v[St::U];
        St::load(v, src, n0, nimg, C, c, W, x0, i0, nsub);
        St::template store<BN>(v, tile, nimg, W, x0, i0, nsub, bn_a, bn_o);
    }
}

template <int OFF>
__device__ __forceinline__ void ds_read128_at(bf16x8& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}

// The rows of one wave: Y0 + RS * j, j < NY (RS = 1: a contiguous block; RS = number of waves: interleaved, which gives
// every wave the same number of in-plane (y_in, y_out) pairs on planes taller than the filter's half width).
template <int K, int KS, int MODE, int H, int NY, int Y0, int RS>
struct BmSched {
    static constexpr int P = K / 2, PS = (KS > 0 ? KS : 1) / 2;
    static constexpr int YLAST = Y0 + RS * (NY - 1);
    static constexpr int YLO = Y0 - P > 0 ? Y0 - P : 0;
    static constexpr int YHI = YLAST + P < H - 1 ? YLAST + P + 1 : H;                     // exclusive
    static constexpr bool SHARE = MODE == 0 && Geo<K>::JOFF == Geo<(KS > 0 ? KS : 5)>::JOFF;   // small chunk == big chunk 0
    static constexpr bool small_at(int yi) {
        if (KS == 0) return false;
        for (int j = 0; j < NY; ++j) {
            const int d = yi - (Y0 + RS * j);
            if (d >= -PS && d <= PS) return true;
        }
        return false;
    }
    static constexpr int reads_at(int yi) { return Geo<K>::NS + ((small_at(yi) && !SHARE) ? 1 : 0); }
};

// ra_big / ra_small: LDS addresses of THIS lane's pieces of tile row YI (the caller steps them row by row: an address that
// is a*YI + b with a run-time stride would be hoisted out of the column-tile loop for all rows at once -- 2 x 48 live values).
template <int K, int KS, int MODE, int H, int NY, int Y0, int RS, int YI>
__device__ __forceinline__ void bm_issue(bf16x8 (&slot)[3], uint32_t ra_big, uint32_t ra_small) {
    using S = BmSched<K, KS, MODE, H, NY, Y0, RS>;
    if constexpr (YI < S::YHI) {
        ds_read128_at<0>(slot[0], ra_big);
        if constexpr (Geo<K>::NS == 2) ds_read128_at<64>(slot[1], ra_big);
        if constexpr (S::small_at(YI) && !S::SHARE) ds_read128_at<0>(slot[2], ra_small);
    }
}

__device__ __forceinline__ void bm_next_row(uint32_t& ra, int stride) {          // opaque to the optimiser on purpose
    asm volatile("v_add_u32 %0, %0, %1" : "+v"(ra) : "s"(stride));
}

// One input row: issue the next row's reads, wait for this row's, then every (output row, chunk) MFMA it feeds.
template <int K, int KS, int MODE, int H, int NY, int Y0, int RS, int YI>
__device__ __forceinline__ void bm_row(f32x4 (&accb)[NY], f32x4 (&accs)[NY], bf16x8 (&ring)[2][3], uint32_t ra_big,
                                       uint32_t ra_small, int ystride, int ystride_s, const bf16x8 (&bfb)[K][Geo<K>::NS],
                                       const bf16x8 (&bfs)[(KS > 0 ? KS : 1)][1]) {
    using S = BmSched<K, KS, MODE, H, NY, Y0, RS>;
    constexpr int NS = Geo<K>::NS;
    if constexpr (YI < S::YHI) {
        constexpr int cur = (YI - S::YLO) & 1;
        bm_next_row(ra_big, ystride);
        if constexpr (KS > 0 && !S::SHARE) bm_next_row(ra_small, ystride_s);
        bm_issue<K, KS, MODE, H, NY, Y0, RS, YI + 1>(ring[cur ^ 1], ra_big, ra_small);
        constexpr int younger = (YI + 1 < S::YHI) ? S::reads_at(YI + 1) : 0;
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(ring[cur][0]), "+v"(ring[cur][1]), "+v"(ring[cur][2]) : "i"(younger));
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx)
#pragma unroll
            for (int j = 0; j < NY; ++j) {
                const int d = YI - (Y0 + RS * j);
                if (d >= -S::P && d <= S::P)
                    accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfb[d + S::P][sidx], ring[cur][sidx], accb[j], 0, 0, 0);
            }
        if constexpr (S::small_at(YI)) {
#pragma unroll
            for (int j = 0; j < NY; ++j) {
                const int d = YI - (Y0 + RS * j);
                if (d >= -S::PS && d <= S::PS) {
                    const bf16x8& a = S::SHARE ? ring[cur][0] : ring[cur][2];
                    if constexpr (MODE == 0) accs[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfs[d + S::PS][0], a, accs[j], 0, 0, 0);
                    else accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfs[d + S::PS][0], a, accb[j], 0, 0, 0);
                }
            }
        }
        bm_row<K, KS, MODE, H, NY, Y0, RS, YI + 1>(accb, accs, ring, ra_big, ra_small, ystride, ystride_s, bfb, bfs);
    }
}

// The `nct` column tiles of the staged segment (first output column x0) for this wave's rows.
// Epilogue: W % 4 == 0 and tiles start at multiples of 16, so a lane's four columns are inside the plane together and its
// 8-byte store is aligned -- one validity test per column tile, then per output row two conversions, one pointer step and
// one store (store_tile's general form re-derives all of that per row: ~200 cycles per row and output, as much time as the
// MFMA rows themselves on the 12 x 40 maps).
template <int K, int KS, int MODE, int H, int NY, int Y0, int RS>
__device__ __forceinline__ void bm_rows(uint32_t a_big, uint32_t a_small, int ystride, int ystride_s,
                                        const bf16x8 (&bfb)[K][Geo<K>::NS],
                                        const bf16x8 (&bfs)[(KS > 0 ? KS : 1)][1], uint16_t* __restrict__ out0,
                                        uint16_t* __restrict__ out1, long plane_off, bool img_ok, int W, int x0, int nct,
                                        int lane, bool want_stats, float (&st)[4], unsigned long long* prof = nullptr) {
    using S = BmSched<K, KS, MODE, H, NY, Y0, RS>;
#pragma unroll 1
    for (int ct = 0; ct < nct; ++ct) {
        PROF_T(t_ct0);
        f32x4 accb[NY], accs[NY];
#pragma unroll
        for (int j = 0; j < NY; ++j) { accb[j] = {0.f, 0.f, 0.f, 0.f}; accs[j] = {0.f, 0.f, 0.f, 0.f}; }
        bf16x8 ring[2][3];
        const uint32_t ab = a_big + 32 * ct + S::YLO * ystride, as = a_small + 32 * ct + S::YLO * ystride_s;
        bm_issue<K, KS, MODE, H, NY, Y0, RS, S::YLO>(ring[0], ab, as);
        bm_row<K, KS, MODE, H, NY, Y0, RS, S::YLO>(accb, accs, ring, ab, as, ystride, ystride_s, bfb, bfs);
        PROF_T(t_ct1);
        PROF_ADD(3, t_ct0, t_ct1);
        // the addresses are derived HERE, after the MFMA rows (opaque copy): hoisted above them they would be live across
        // the rows, spilled, and every reload's vmcnt wait would drain the epilogue's own stores
        int col = x0 + 16 * ct + 4 * (lane >> 4);
        asm volatile("" : "+v"(col));
        if (img_ok && col < W) {
            const long first = plane_off + (long)Y0 * W + col;
            uint16_t* p0 = out0 + first;
            uint16_t* p1 = (MODE == 0 && KS > 0) ? out1 + first : nullptr;
            const int step = RS * W;
#pragma unroll
            for (int j = 0; j < NY; ++j) {
                *reinterpret_cast<uint2*>(p0) = make_uint2(pack_bf16x2(accb[j][0], accb[j][1]), pack_bf16x2(accb[j][2], accb[j][3]));
                p0 += step;
                if constexpr (MODE == 0 && KS > 0) {
                    *reinterpret_cast<uint2*>(p1) = make_uint2(pack_bf16x2(accs[j][0], accs[j][1]), pack_bf16x2(accs[j][2], accs[j][3]));
                    p1 += step;
                }
            }
        }
        if constexpr (MODE == 0) {
            if (want_stats) {
                const int po = img_ok ? (int)plane_off : -1;
#pragma unroll
                for (int j = 0; j < NY; ++j) {
                    const RowOffs ro{po >= 0 ? po + (Y0 + RS * j) * W : -1};
                    tile_stats(accb[j], ro, W, x0 + 16 * ct, lane, st[0], st[1]);
                    if constexpr (KS > 0) tile_stats(accs[j], ro, W, x0 + 16 * ct, lane, st[2], st[3]);
                }
            }
        }
    }
}

// NTX: column tiles per staged segment; a workgroup walks `cpw` column tiles of its channel(s) from tile index wgi * cpw in
// segments of NTX (one segment covers the plane on the small maps; the 48 x 160 maps are cut into 2 workgroups x (3 + 2)
// tiles per channel).  RS: see BmSched.
template <int K, int KS, int MODE, int H, int NY, int NTX, int RS, bool BN>
__global__ __launch_bounds__(256, 1) void dwconv_bm_kernel(
    const uint16_t* __restrict__ in0, const uint16_t* __restrict__ in1, const uint16_t* __restrict__ w_big,
    const uint16_t* __restrict__ w_small, uint16_t* __restrict__ out0, uint16_t* __restrict__ out1, int N, int C, int W,
    int gi, int ngroups, int wgpc, int cpw, int tile_bytes, int tile1_bytes, float* __restrict__ stats, BnIn bn) {
    static_assert(!BN || MODE == 0, "the fused input BatchNorm is a forward feature");
    static_assert(H % NY == 0 && 4 % (H / NY) == 0, "a workgroup holds whole channels");
    static_assert(RS == 1 || RS == H / NY, "rows of a wave: contiguous or interleaved over the channel's waves");
    using GE = Geo<K>;
    using GS = Geo<(KS > 0 ? KS : 5)>;
    using B = BmGeo<K, NTX>;
    constexpr int SPLIT = H / NY, CPW = 4 / SPLIT;
    constexpr int NT_IN = (MODE == 1 && KS > 0) ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    int bid = blockIdx.x;
    const int wgi = bid % wgpc; bid /= wgpc;
    const int grp = bid % ngroups;
    const int cb = bid / ngroups;
    const int cw = cb * CPW + wave / SPLIT, part = wave % SPLIT;
    const bool active = cw < C;                                 // idle waves still meet the barriers
    const int c = active ? cw : C - 1;
    const int n0 = grp * gi, nimg = min(gi, N - n0);
#ifdef DW_PROF
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    PROF_T(t_begin);

    bf16x8 bf_big[K][GE::NS];
    uint8_t* scratch = smem + wave * frag_scratch_bytes(K, KS);
    build_bfrags<K, GE::NS>(bf_big, w_big + (long)c * packed_elems(K), scratch, lane);
    bf16x8 bf_small[(KS > 0 ? KS : 1)][1];
    if constexpr (KS > 0) build_bfrags<KS, 1>(bf_small, w_small + (long)c * packed_elems(KS), scratch + K * FR_ROW_B, lane);
    float bn_a = 1.f, bn_o = 0.f;
    if constexpr (BN) bn_channel_affine(bn, c, lane, active && part == 0 && grp == 0 && wgi == 0, bn_a, bn_o);

    // dgrad: the second input (the 5x5 branch's gradient) is staged with the 5x5 kernel's own, narrower column geometry
    using B1 = BmGeo<(KS > 0 ? KS : 5), NTX>;
    uint8_t* tile0 = smem + (wave / SPLIT) * (tile_bytes + (NT_IN == 2 ? tile1_bytes : 0));
    // this lane's A row = image (lane & 15), clamped: the rows past the group's last image repeat it and are never stored
    const int n = min(lane & 15, nimg - 1);
    const uint32_t a_big = lds_addr(tile0) + (uint32_t)(n * B::STRIDE + 16 * (lane >> 4));
    const uint32_t a_small = MODE == 1 ? lds_addr(tile0 + tile_bytes) + (uint32_t)(n * B1::STRIDE + 16 * (lane >> 4))
                                       : a_big + (GE::JOFF - GS::JOFF) * 2;
    const int ystride = nimg * B::STRIDE, ystride_s = MODE == 1 ? nimg * B1::STRIDE : ystride;
    const bool img_ok = (lane & 15) < nimg;
    const long plane_off = ((long)(n0 + n) * C + c) * (long)H * W;
    float st[4] = {0.f, 0.f, 0.f, 0.f};
    const bool want_stats = stats != nullptr;
#ifdef DW_PROF
#define PPEA_BM_PROF prof
#else
#define PPEA_BM_PROF nullptr
#endif
    const int ct_end = min((W + 15) / 16, (wgi + 1) * cpw);
    PROF_T(t_setup_done);
    PROF_ADD(0, t_begin, t_setup_done);
    for (int cs = wgi * cpw; cs < ct_end; cs += NTX) {
        PROF_T(t_seg);
        // first pass: every wave's fragments are in registers and the scratch is free; later: the segment's reads are done
        __syncthreads();
        const int x0 = 16 * cs;
        if ((W & 7) == 0) {
            stage_bm<K, NTX, H, BN, 8, CPW>(tile0, in0, n0, nimg, C, c, W, x0, part * 64 + lane, SPLIT * 64, bn_a, bn_o);
            if constexpr (NT_IN == 2) stage_bm<(KS > 0 ? KS : 5), NTX, H, false, 8, CPW>(tile0 + tile_bytes, in1, n0, nimg, C, c, W, x0, part * 64 + lane, SPLIT * 64, 1.f, 0.f);
        } else {
            stage_bm<K, NTX, H, BN, 4, CPW>(tile0, in0, n0, nimg, C, c, W, x0, part * 64 + lane, SPLIT * 64, bn_a, bn_o);
            if constexpr (NT_IN == 2) stage_bm<(KS > 0 ? KS : 5), NTX, H, false, 4, CPW>(tile0 + tile_bytes, in1, n0, nimg, C, c, W, x0, part * 64 + lane, SPLIT * 64, 1.f, 0.f);
        }
        __syncthreads();
        PROF_T(t_staged);
        PROF_ADD(1, t_seg, t_staged);
        const int nct = min(NTX, ct_end - cs);
        if (active) {
#define PPEA_BM_PART(P_)                                                                                               \
            if constexpr (P_ < SPLIT) {                                                                                 \
                if (part == P_)                                                                                         \
                    bm_rows<K, KS, MODE, H, NY, (RS == 1 ? P_ * NY : P_), RS>(a_big, a_small, ystride, ystride_s, bf_big, bf_small, out0, \
                                                                             out1, plane_off, img_ok, W, x0, nct, lane, want_stats, st, PPEA_BM_PROF); \
            }
            PPEA_BM_PART(0) PPEA_BM_PART(1) PPEA_BM_PART(2) PPEA_BM_PART(3)
#undef PPEA_BM_PART
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PROF_T(t_mac);
        PROF_ADD(2, t_staged, t_mac);
    }
#ifdef DW_PROF
    PROF_T(t_end);
    prof[5] = t_end - t_begin;
    {
        const long wid = (long)blockIdx.x * 4 + wave;
        if (lane == 0 && wid < 4096)
            for (int i = 0; i < 8; ++i) g_dw_prof[wid][i] = prof[i];
    }
#endif
    if constexpr (MODE == 0) {
        if (want_stats && active) {                              // stats [2][C][wpc][2], wpc = SPLIT * ngroups * wgpc
            const float sb = wave_sum(st[0]), qb = wave_sum(st[1]), ss = wave_sum(st[2]), qs = wave_sum(st[3]);
            if (lane == 0) {
                const int wpc = SPLIT * ngroups * wgpc;
                const long e = ((long)c * wpc + (grp * wgpc + wgi) * SPLIT + part) * 2;
                stats[e] = sb; stats[e + 1] = qb;
                if constexpr (KS > 0) { stats[(long)C * wpc * 2 + e] = ss; stats[(long)C * wpc * 2 + e + 1] = qs; }
            }
        }
    }
}

// returns PPEA_ERR_UNSUPPORTED when the shape is not this variant's (the caller falls back to the row-band kernel)
template <int K, int KS, int MODE, int H, int NY, int NTX, int RS, bool BN>
int launch_bm(const uint16_t* in0, const uint16_t* in1, const uint16_t* wb, const uint16_t* ws, uint16_t* o0, uint16_t* o1,
              int N, int C, int W, hipStream_t st, float* stats, int* wpc_out, const BnIn* bn) {
    using B = BmGeo<K, NTX>;
    using B1 = BmGeo<(KS > 0 ? KS : 5), NTX>;
    constexpr int SPLIT = H / NY, CPW = 4 / SPLIT;
    constexpr int NT_IN = (MODE == 1 && KS > 0) ? 2 : 1;
    if ((long)N * C * H * W >= (1L << 31) || (long)N * C * H * W < 8) return PPEA_ERR_UNSUPPORTED;   // (masked pieces read the tensor's first 16 bytes)
    int gi = 0, tile_bytes = 0, tile1_bytes = 0;
    for (int cand : {16, 12, 8, 4}) {                            // images per group: the largest whose tiles fit
        const int alloc = N < cand ? N : cand;
        const int tb = H * alloc * B::STRIDE, tb1 = NT_IN == 2 ? H * alloc * B1::STRIDE : 0;
        if (CPW * (tb + tb1) <= LDS_LIMIT) { gi = cand; tile_bytes = tb; tile1_bytes = tb1; break; }
    }
    if (!gi) return PPEA_ERR_UNSUPPORTED;
    const int ngroups = (N + gi - 1) / gi;
    // column tiles per workgroup: whole planes on the small maps; on wide maps enough workgroups to fill the chip
    const int ntx_all = (W + 15) / 16;
    int wgpc = 1;
    if (ntx_all > NTX) {
        const long base = (long)((C + CPW - 1) / CPW) * ngroups;
        wgpc = (int)((256 + base - 1) / base);
        if (wgpc < 1) wgpc = 1;
        if (wgpc > ntx_all) wgpc = ntx_all;
    }
    const int cpw = (ntx_all + wgpc - 1) / wgpc;
    wgpc = (ntx_all + cpw - 1) / cpw;
    if (wpc_out != nullptr) { *wpc_out = SPLIT * ngroups * wgpc; return 0; }
    size_t lds = (size_t)CPW * (tile_bytes + tile1_bytes);
    if (lds < (size_t)4 * frag_scratch_bytes(K, KS)) lds = (size_t)4 * frag_scratch_bytes(K, KS);
    auto kern = dwconv_bm_kernel<K, KS, MODE, H, NY, NTX, RS, BN>;
    const BnIn bnv = bn != nullptr ? *bn : BnIn{nullptr, 0, 0.f, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(((C + CPW - 1) / CPW) * ngroups * wgpc)), dim3(256), lds, st, in0, in1, wb, ws,
                       o0, o1, N, C, W, gi, ngroups, wgpc, cpw, tile_bytes, tile1_bytes, stats, bnv);
    return launch_status();
}

template <int K, int KS, int MODE, int H, int NY, int NTX, int RS>
int launch_bm_n(const uint16_t* in0, const uint16_t* in1, const uint16_t* wb, const uint16_t* ws, uint16_t* o0, uint16_t* o1,
                int N, int C, int W, hipStream_t st, float* stats, int* wpc_out, const BnIn* bn) {
    if constexpr (MODE == 0) {
        if (bn != nullptr) return launch_bm<K, KS, MODE, H, NY, NTX, RS, true>(in0, in1, wb, ws, o0, o1, N, C, W, st, stats, wpc_out, bn);
    }
    if (bn != nullptr) return PPEA_ERR_UNSUPPORTED;
    return launch_bm<K, KS, MODE, H, NY, NTX, RS, false>(in0, in1, wb, ws, o0, o1, N, C, W, st, stats, wpc_out, nullptr);
}

// The plane sizes this variant is built for: RepLKNet stages 0-3 of 192-row frames (48 x W in segments, 24 x {64, 80},
// 12 x {8..48}, 6 x {4..48}).
template <int K, int KS, int MODE>
int launch_bm_w(const uint16_t* in0, const uint16_t* in1, const uint16_t* wb, const uint16_t* ws, uint16_t* o0, uint16_t* o1,
                int N, int C, int H, int W, hipStream_t st, float* stats, int* wpc_out, const BnIn* bn) {
    static const bool on = !(getenv("PPEA_DW_BM") != nullptr && getenv("PPEA_DW_BM")[0] == '0');
    if (!on || (W & 3) != 0) return PPEA_ERR_UNSUPPORTED;
    const int ntx = (W + 15) / 16;
#define PPEA_BM_NTX(H_, NTX_)                                                                                          \
    if (H == H_ && ntx == NTX_) return launch_bm_n<K, KS, MODE, H_, 6, NTX_, 1>(in0, in1, wb, ws, o0, o1, N, C, W, st, stats, wpc_out, bn);
    if constexpr (KS == 5 && K == 29) { PPEA_BM_NTX(24, 4) PPEA_BM_NTX(24, 5) }
    if constexpr (KS == 5 && K == 27) { PPEA_BM_NTX(12, 1) PPEA_BM_NTX(12, 2) PPEA_BM_NTX(12, 3) }
    if constexpr (KS == 5 && K == 13) { PPEA_BM_NTX(6, 1) PPEA_BM_NTX(6, 2) PPEA_BM_NTX(6, 3) }
#undef PPEA_BM_NTX
    if constexpr (KS == 5 && K == 31 && MODE == 1) {
        // 48-row planes (stage 0 of 192-row frames), data gradient only: 12 interleaved rows per wave, segments of 2 column
        // tiles (two staged inputs), M = images -- worth it when the image groups fill most of the 16 MFMA rows (66 us
        // against the row-band kernel's 74 at [12,128,48,160]).  The forward stays on the window-sharing row-band kernel:
        // measured 65-73 us here against 58 (tools/bench_dwconv.py) -- all 256 workgroups stage their 129 KB segment at the
        // same time with nothing to overlap it (one workgroup per CU, no registers left for a prefetch) and the epilogue's
        // 32-byte runs per image cost as much as on the small maps.  PPEA_DW_BM48=0 disables the variant.
        static const bool on48 = !(getenv("PPEA_DW_BM48") != nullptr && getenv("PPEA_DW_BM48")[0] == '0');
        const int groups16 = (N + 15) / 16;
        if (on48 && H == 48 && (W & 7) == 0 && 10 * N >= 7 * 16 * groups16)
            return launch_bm_n<K, KS, MODE, 48, 12, 2, 4>(in0, in1, wb, ws, o0, o1, N, C, W, st, stats, wpc_out, bn);
    }
    return PPEA_ERR_UNSUPPORTED;
}

// stats / wpc_out: forward only -- per-wave partial sums for the BatchNorm pair (see the kernel); wpc_out != nullptr:
// do not launch, return the number of waves per channel (= partials per channel) the launch would use
template <int K, int KS, int MODE, int NSEG, bool BN = false, bool WS = false>
int launch(const uint16_t* in0, const uint16_t* in1, const uint16_t* wb, const uint16_t* ws, uint16_t* o0,
           uint16_t* o1, int N, int C, int H, int W, hipStream_t st, float* stats = nullptr, int* wpc_out = nullptr,
           const BnIn* bn = nullptr) {
    constexpr int STRIDE_B = Seg<K, NSEG>::STRIDE;
    constexpr int NT_IN = (MODE == 1 && KS > 0) ? 2 : 1;
    if ((long)N * C * H * W >= (1L << 31)) return PPEA_ERR_UNSUPPORTED;      // 32-bit element offsets
    if ((long)N * C * H * W < 8) return PPEA_ERR_UNSUPPORTED;                // masked lanes read the first 16 bytes
    // largest band / stacking whose per-wave region fits four times into the 160 KB of LDS
    int band = 0, G = 1, tile_bytes = 0, region = 0;
    for (int cand : {48, 32, 16}) {
        int g = (H < cand) ? (cand / H < N ? cand / H : N) : 1;
        if (g < 1) g = 1;
        for (; g >= 1; --g) {
            const int rows = (g > 1) ? g * (H + K - 1) : ((H < cand ? H : cand) + K - 1);
            const int tb = (rows * STRIDE_B + 15) & ~15;
            // (the region also hosts the prologue's fragment construction, see build_bfrags)
            const int reg = NT_IN * tb > frag_scratch_bytes(K, KS) ? NT_IN * tb : frag_scratch_bytes(K, KS);
            if (WAVES * reg <= LDS_LIMIT) { band = cand; G = g; tile_bytes = tb; region = reg; break; }
        }
        if (band) break;
    }
    if (!band) return PPEA_ERR_UNSUPPORTED;
    const int bands = (G > 1) ? 1 : (H + band - 1) / band;
    const int segs = (W + 16 * NSEG - 1) / (16 * NSEG);
    const int groups = (G > 1) ? (N + G - 1) / G : N * bands;
    const int items_per_channel = groups * segs;
    // waves per channel: ~one wave per SIMD over the whole chip (1024), each wave keeps its channel's
    // Toeplitz fragments in registers and walks `ipw` items
    int wpc = 1024 / C;
    if (wpc < 1) wpc = 1;
    if (wpc > items_per_channel) wpc = items_per_channel;
    const int ipw = (items_per_channel + wpc - 1) / wpc;
    wpc = (items_per_channel + ipw - 1) / ipw;
    if (wpc_out != nullptr) { *wpc_out = wpc; return 0; }
    const long total_waves = (long)C * wpc;
    const size_t lds = (size_t)WAVES * region;
    if constexpr (WS) {
        if (!(G == 1 && band == 48 && H % 48 == 0)) return PPEA_ERR_UNSUPPORTED;       // every item: one 48-row band
    }
    auto kern = dwconv_mfma_kernel<K, KS, MODE, NSEG, BN, WS>;
    const BnIn bnv = bn != nullptr ? *bn : BnIn{nullptr, 0, 0.f, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)((total_waves + WAVES - 1) / WAVES)), dim3(64 * WAVES), lds, st, in0,
                       in1, wb, ws, o0, o1, N, C, H, W, G, band, bands, segs, items_per_channel, ipw, wpc,
                       total_waves, tile_bytes, region, stats, bnv);
    return launch_status();
}

// staged columns for a choice of NSEG (cost model: LDS staging traffic)
template <int K>
inline long staged_cols(int W, int nseg) {
    const long segs = (W + 16 * nseg - 1) / (16 * nseg);
    return segs * (16 * (nseg - 1) + 32 * Geo<K>::NS);
}

template <int K, int KS, int MODE>
int launch_k(const uint16_t* in0, const uint16_t* in1, const uint16_t* wb, const uint16_t* ws, uint16_t* o0,
             uint16_t* o1, int N, int C, int H, int W, hipStream_t st, float* stats, int* wpc_out, const BnIn* bn) {
    {                                                            // small planes: the batch-major variant
        const int err = launch_bm_w<K, KS, MODE>(in0, in1, wb, ws, o0, o1, N, C, H, W, st, stats, wpc_out, bn);
        if (err != PPEA_ERR_UNSUPPORTED) return err;
    }
#ifdef PPEA_BM_ONLY                                              // (compile-time experiments with the batch-major variant alone)
    return PPEA_ERR_UNSUPPORTED;
#endif
    const long c5 = staged_cols<K>(W, 5), c3 = staged_cols<K>(W, 3), c2 = staged_cols<K>(W, 2);
    const int nseg = (c5 <= c3 && c5 <= c2) ? 5 : (c3 <= c2 ? 3 : 2);
    if constexpr (K == 31 && KS == 5) {
        // window sharing between the three stacked tiles of a 48-row plane (stage 0 at 192 x 640): launch() refuses the
        // variant unless every item is one 48-row band, and the generic kernel takes over below
        static const bool ws_on = !(getenv("PPEA_DW_WS") != nullptr && getenv("PPEA_DW_WS")[0] == '0');
        if (ws_on && nseg == 5 && wpc_out == nullptr) {
            int err;
            if constexpr (MODE == 0) {
                err = bn != nullptr ? launch<K, KS, MODE, 5, true, true>(in0, in1, wb, ws, o0, o1, N, C, H, W, st, stats, nullptr, bn)
                                    : launch<K, KS, MODE, 5, false, true>(in0, in1, wb, ws, o0, o1, N, C, H, W, st, stats, nullptr, nullptr);
            } else {
                err = bn != nullptr ? PPEA_ERR_UNSUPPORTED
                                    : launch<K, KS, MODE, 5, false, true>(in0, in1, wb, ws, o0, o1, N, C, H, W, st, stats, nullptr, nullptr);
            }
            if (err != PPEA_ERR_UNSUPPORTED) return err;
        }
    }
    if constexpr (MODE == 0 && KS == 5) {
        if (bn != nullptr) {                                     // fused input BatchNorm + ReLU (RepLKBlock forward)
            if (nseg == 5) return launch<K, KS, MODE, 5, true>(in0, in1, wb, ws, o0, o1, N, C, H, W, st, stats, wpc_out, bn);
            if (nseg == 3) return launch<K, KS, MODE, 3, true>(in0, in1, wb, ws, o0, o1, N, C, H, W, st, stats, wpc_out, bn);
            return launch<K, KS, MODE, 2, true>(in0, in1, wb, ws, o0, o1, N, C, H, W, st, stats, wpc_out, bn);
        }
    }
    if (bn != nullptr) return PPEA_ERR_UNSUPPORTED;
    if (nseg == 5) return launch<K, KS, MODE, 5>(in0, in1, wb, ws, o0, o1, N, C, H, W, st, stats, wpc_out);
    if (nseg == 3) return launch<K, KS, MODE, 3>(in0, in1, wb, ws, o0, o1, N, C, H, W, st, stats, wpc_out);
    return launch<K, KS, MODE, 2>(in0, in1, wb, ws, o0, o1, N, C, H, W, st, stats, wpc_out);
}

template <int MODE>
int dispatch(const uint16_t* in0, const uint16_t* in1, const uint16_t* wb, const uint16_t* ws, uint16_t* o0,
             uint16_t* o1, int N, int C, int H, int W, int K, int KS, hipStream_t st, float* stats = nullptr,
             int* wpc_out = nullptr, const BnIn* bn = nullptr) {
#define PPEA_CASE(K_)                                                                             \
    case K_:                                                                                      \
        return KS == 5 ? launch_k<K_, 5, MODE>(in0, in1, wb, ws, o0, o1, N, C, H, W, st, stats, wpc_out, bn)   \
                       : launch_k<K_, 0, MODE>(in0, in1, wb, ws, o0, o1, N, C, H, W, st, stats, wpc_out, bn);
    switch (K) {
        PPEA_CASE(31) PPEA_CASE(29) PPEA_CASE(27) PPEA_CASE(13)
        default: return PPEA_ERR_UNSUPPORTED;
    }
#undef PPEA_CASE
}

}  // namespace

extern "C" {

#ifdef DW_PROF
int ppea_debug_dwconv_prof(unsigned long long* out) {   // host buffer [4096][8]
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dw_prof), sizeof(unsigned long long) * 4096 * 8);
}
#endif

long ppea_dwconv_lk_packed_bytes(int C, int K) {
    if (C <= 0 || K < 3 || K > 31 || (K & 1) == 0) return -1;
    return (long)C * packed_elems(K) * 2;
}

int ppea_dwconv_lk_pack_bf16(const float* w, void* packed, int C, int K, int flip, void* stream) {
    if (C <= 0 || K < 3 || K > 31 || (K & 1) == 0) return PPEA_ERR_UNSUPPORTED;
    const long total = (long)C * packed_elems(K);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_filter_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w,
                       (uint16_t*)packed, C, K, flip);
    return launch_status();
}

int ppea_dwconv_lk_fwd_bf16p(const uint16_t* x, const void* packed_big, const void* packed_small, uint16_t* y_big,
                             uint16_t* y_small, int N, int C, int H, int W, int K, int KS, void* stream) {
    if (N < 0 || C <= 0 || H <= 0 || W <= 0) return PPEA_ERR_UNSUPPORTED;
    if (N == 0) return 0;
    if (packed_small == nullptr || y_small == nullptr) KS = 0;
    if (KS != 0 && KS != 5) return PPEA_ERR_UNSUPPORTED;
    return dispatch<0>(x, nullptr, (const uint16_t*)packed_big, (const uint16_t*)packed_small, y_big, y_small, N, C,
                       H, W, K, KS, (hipStream_t)stream);
}

// Forward as above plus the statistics of the two BatchNorms that follow (rka.py:232-239): stats [2][C][P][2] fp32 =
// per channel and wave of the channel (sum, sum of squares) of the stored y_big (first half) and y_small values,
// P = ppea_dwconv_lk_stats_partials(N, C, H, W, K, KS); reduce each half with ppea_bn_finalize_sums_f32.
int ppea_dwconv_lk_stats_partials(int N, int C, int H, int W, int K, int KS) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || (KS != 0 && KS != 5)) return 0;
    int wpc = 0;
    const int err = dispatch<0>(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, N, C, H, W, K, KS, nullptr, nullptr, &wpc);
    return err == 0 ? wpc : 0;
}
int ppea_dwconv_lk_fwd_stats_bf16p(const uint16_t* x, const void* packed_big, const void* packed_small, uint16_t* y_big,
                                   uint16_t* y_small, float* stats, int N, int C, int H, int W, int K, int KS,
                                   void* stream) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || stats == nullptr) return PPEA_ERR_UNSUPPORTED;
    if (packed_small == nullptr || y_small == nullptr) KS = 0;
    if (KS != 0 && KS != 5) return PPEA_ERR_UNSUPPORTED;
    return dispatch<0>(x, nullptr, (const uint16_t*)packed_big, (const uint16_t*)packed_small, y_big, y_small, N, C,
                       H, W, K, KS, (hipStream_t)stream, stats, nullptr);
}

// Forward with the BatchNorm (+ ReLU) of the INPUT fused into the staging pass (RepLKBlock: pw1 conv_bn_relu -> large
// kernel, replknet_adapter.py:305-308): x = the 1x1 conv's output, `sums` [C][P][2] its epilogue's partial (sum, sum of
// squares) (ppea_pwconv_stats_bf16), count = N*H*W.  Every wave finalises its channel (fp64, the arithmetic of
// ppea_bn_finalize_sums_f32), stages relu(gamma * (x - mean) * invstd + beta) rounded to bf16 (zero padding outside the
// plane), and the channel's first wave writes mean / invstd [C] (saved for backward) and updates the running statistics
// (NULL: no update).  KS must be 5.
int ppea_dwconv_lk_fwd_bn_bf16p(const uint16_t* x, const void* packed_big, const void* packed_small, uint16_t* y_big,
                                uint16_t* y_small, const float* sums, int P, long count, const float* gamma,
                                const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                float* mean, float* invstd, int N, int C, int H, int W, int K, int KS, void* stream) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || KS != 5 || packed_small == nullptr || y_small == nullptr) return PPEA_ERR_UNSUPPORTED;
    if (sums == nullptr || P <= 0 || count <= 0 || gamma == nullptr || beta == nullptr || mean == nullptr || invstd == nullptr)
        return PPEA_ERR_ARG;
    const BnIn bn{sums, P, (float)count, eps, momentum, gamma, beta, running_mean, running_var, mean, invstd};
    return dispatch<0>(x, nullptr, (const uint16_t*)packed_big, (const uint16_t*)packed_small, y_big, y_small, N, C,
                       H, W, K, KS, (hipStream_t)stream, nullptr, nullptr, &bn);
}

int ppea_dwconv_lk_bwd_data_bf16p(const uint16_t* dy_big, const uint16_t* dy_small, const void* packed_big_flip,
                                  const void* packed_small_flip, uint16_t* dx, int N, int C, int H, int W, int K,
                                  int KS, void* stream) {
    if (N < 0 || C <= 0 || H <= 0 || W <= 0) return PPEA_ERR_UNSUPPORTED;
    if (N == 0) return 0;
    if (packed_small_flip == nullptr || dy_small == nullptr) KS = 0;
    if (KS != 0 && KS != 5) return PPEA_ERR_UNSUPPORTED;
    return dispatch<1>(dy_big, dy_small, (const uint16_t*)packed_big_flip, (const uint16_t*)packed_small_flip, dx,
                       nullptr, N, C, H, W, K, KS, (hipStream_t)stream);
}

}  // extern "C"
