// Pointwise (1x1) convolution on the matrix cores, NCHW bf16 in / fp32 accumulate / bf16 out, gfx950.
// Replaces the 1x1 nn.Conv2d of RepLKBlock / ConvFFN / stem[2] / transitions[.][0]
// (networks/replknet_adapter.py:270-271, 296-297, 415, 452) in forward and, with the transposed
// weight matrix, their data gradients.
//
//     Y[n][m][p] = sum_k A[m][k] * X[n][k][p] (+ bias[m])        A = W [Cout][Cin]  (or W^T for dgrad)
//
// Per image a GEMM with the activation operand in its natural NCHW form: k (channel) is the SLOW axis
// of X, pixels are contiguous.  The tile of X is staged into LDS exactly as it lies in memory
// ([32 channels][128 pixels], coalesced 16-byte reads, no transposition) and the MFMA B fragments
// (8 consecutive k for one pixel) come out of `ds_read_b64_tr_b16`, the CDNA4 transposing LDS read.
//   * block = 4 waves, tile BM x 128 pixels, BK = 32; `v_mfma_f32_16x16x32_bf16`;
//   * A tile [BM][32] with a 96-byte row stride, B tile XOR-swizzled in 8-byte chunks: both fragment
//     reads are bank-conflict free (cdna_hip_programming.md 2, T10);
//   * global -> register prefetch of the next K-tile overlaps the MFMAs of the current one; LDS is
//     double buffered so a K step costs one barrier.
#include "common.h"
#include <cstdlib>
#include <cstring>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

constexpr int BN = 128;
constexpr int B_STRIDE = 256;                   // bytes per B row (128 pixels), swizzled
// K step BK = 32 or 64 channels: the long-K, few-tile shapes of stage 2 are bound by the per-step barrier and
// load latency, not by MFMA issue -- twice the work per step halves that overhead.
constexpr int a_stride(int bk) { return bk == 32 ? 96 : 160; }   // bytes per A row in LDS: data + 32 (32 * odd)

__device__ __forceinline__ int b_swz(int row) { return 4 * (row & 3) + 16 * ((row >> 3) & 1); }

__device__ __forceinline__ float bf2f(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// EPI 4: as EPI 0, and per-channel partial sums (sum, sum of squares) of the values AS STORED, per pixel-wave of the
// workgroup, into (float*)Y2 [M][B * pixel tiles * WN][2] -- the BatchNorm that follows needs no statistics pass.
// EPI 0: Y = acc + bias.   EPI 1: Y = pre = acc + bias, Y2 = GELU(pre) (pre rounded to bf16 first, like an
// autocast nn.GELU on the stored tensor).   EPI 2: Y = acc * GELU'(aux[n][m][p]) (data gradient through GELU).
// TA: the matrix is given transposed, At [K][M] (m contiguous) -- its tile is staged like the X tile and the A
// fragments come out of the transposing LDS read as well, so a data gradient uses the forward weight as is.
template <int BM, int WM, int WN, int EPI, bool TA, int BK>
__global__ __launch_bounds__(256) void pwconv_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ X,
                                                     const void* __restrict__ bias, int bias_bf16,
                                                     const uint16_t* __restrict__ aux, uint16_t* __restrict__ Y,
                                                     uint16_t* __restrict__ Y2, int M, int K, int HW) {
    static_assert(WM * WN == 4, "four waves");
    constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 16, NT = TN / 16;
    constexpr int A_STRIDE = a_stride(BK), KH = BK / 32;
    constexpr int A_BYTES = TA ? BK * B_STRIDE : BM * A_STRIDE, B_BYTES = BK * B_STRIDE;
    constexpr int A_CH = (BM * BK / 8 + 255) / 256;  // 16-byte chunks of the A tile per thread
    constexpr int A_CPR = TA ? BM / 8 : BK / 8;      // chunks per staged row: [BK k][BM m] or [BM m][BK k]
    constexpr int B_CH = BK * 16 / 256;              // 16-byte chunks of the X tile per thread
    constexpr int BUF = A_BYTES + B_BYTES;
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * BUF];      // double buffered: one barrier per K step

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int p0 = blockIdx.x * BN, m0 = blockIdx.y * BM, n = blockIdx.z;
    const uint16_t* Xn = X + (long)n * K * HW;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = {0.f, 0.f, 0.f, 0.f};

    // per-thread staging coordinates
    int a_row[A_CH], a_ch[A_CH];
#pragma unroll
    for (int c = 0; c < A_CH; ++c) { const int idx = tid + c * 256; a_row[c] = idx / A_CPR; a_ch[c] = idx % A_CPR; }
    int b_row[B_CH], b_c16[B_CH];
#pragma unroll
    for (int c = 0; c < B_CH; ++c) { const int idx = tid + c * 256; b_row[c] = idx >> 4; b_c16[c] = idx & 15; }

    uint4 a_reg[A_CH], b_reg[B_CH];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int c = 0; c < A_CH; ++c) {
            if constexpr (TA) {
                const int m = m0 + a_ch[c] * 8;
                a_reg[c] = (m < M && a_row[c] < BK && k0 + a_row[c] < K)
                               ? *reinterpret_cast<const uint4*>(A + (long)(k0 + a_row[c]) * M + m) : make_uint4(0, 0, 0, 0);
            } else {
                const int m = m0 + a_row[c];
                a_reg[c] = (m < M && a_row[c] < BM && k0 + a_ch[c] * 8 < K)
                               ? *reinterpret_cast<const uint4*>(A + (long)m * K + k0 + a_ch[c] * 8) : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int c = 0; c < B_CH; ++c) {
            const int p = p0 + b_c16[c] * 8;
            b_reg[c] = (p < HW && k0 + b_row[c] < K) ? *reinterpret_cast<const uint4*>(Xn + (long)(k0 + b_row[c]) * HW + p)
                                                     : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tiles = [&](uint8_t* As) {
        uint8_t* Bs = As + A_BYTES;
#pragma unroll
        for (int c = 0; c < A_CH; ++c) {
            if constexpr (TA) {
                if (a_row[c] < BK) {
                    const int s = b_swz(a_row[c]);
                    uint8_t* rowp = As + a_row[c] * B_STRIDE;
                    // the swizzle is a multiple of 4 chunks: the two 8-byte halves stay adjacent -> one 16-byte store
                    *reinterpret_cast<uint4*>(rowp + (((2 * a_ch[c]) ^ s) << 3)) = a_reg[c];
                }
            } else {
                if (a_row[c] < BM) *reinterpret_cast<uint4*>(As + a_row[c] * A_STRIDE + a_ch[c] * 16) = a_reg[c];
            }
        }
#pragma unroll
        for (int c = 0; c < B_CH; ++c) {
            const int s = b_swz(b_row[c]);
            uint8_t* rowp = Bs + b_row[c] * B_STRIDE;
            *reinterpret_cast<uint4*>(rowp + (((2 * b_c16[c]) ^ s) << 3)) = b_reg[c];
        }
    };

    // fragment offsets inside a buffer
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int b_rowi = 8 * g + q;
    const int a_off = TA ? b_rowi * B_STRIDE : (wm * TM + li) * A_STRIDE + g * 16;
    const int b_off = A_BYTES + b_rowi * B_STRIDE;
    const int b_s = b_swz(b_rowi);

    load_tiles(0);
    store_tiles(lds);
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < K; k0 += BK) {
        const bool more = k0 + BK < K;
        if (more) load_tiles(k0 + BK);            // in flight while the MFMAs of this step run
        const uint8_t* buf = lds + cur * BUF;
#pragma unroll
        for (int h = 0; h < KH; ++h) {                    // 32-wide halves of the K step
            bf16x8 af[MT], bfr[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if constexpr (TA) {
                    const int chunk = (((wm * TM + 16 * i) >> 2) + pp) ^ b_swz(b_rowi);
                    const uint8_t* ap = buf + a_off + h * 32 * B_STRIDE + (chunk << 3);
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ap));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ap + 4 * B_STRIDE));
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    af[i] = __builtin_bit_cast(bf16x8, both);
                } else {
                    af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(buf + a_off + i * 16 * A_STRIDE + h * 64));
                }
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int chunk = (((wn * TN + 16 * j) >> 2) + pp) ^ b_s;
                const uint8_t* bp = buf + b_off + h * 32 * B_STRIDE + (chunk << 3);
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(bp));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(bp + 4 * B_STRIDE));
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                bfr[j] = __builtin_bit_cast(bf16x8, both);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    // operands swapped (the two fragment layouts are the same registers): the accumulator holds the
                    // TRANSPOSED tile -- a lane owns 4 consecutive pixels of one output channel = one 8-byte store
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        if (more) store_tiles(lds + (cur ^ 1) * BUF);   // the other buffer was last read one barrier ago
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: transposed C layout -- col = lane & 15 = output channel, row = 4 * (lane >> 4) + r = pixel
    // (HW % 8 == 0 and the four pixels start at a multiple of 4: all inside the plane or all outside)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + wm * TM + 16 * i + li;
        if (m >= M) continue;
        float bv = 0.f;
        if (bias != nullptr)
            bv = bias_bf16 ? bf2f(reinterpret_cast<const uint16_t*>(bias)[m]) : reinterpret_cast<const float*>(bias)[m];
        float st_s = 0.f, st_q = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int p = p0 + wn * TN + 16 * j + 4 * g;
            if (p >= HW) continue;
            const long o = (long)n * M * HW + (long)m * HW + p;
            uint16_t v[4];
            if constexpr (EPI == 0 || EPI == 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = f2bf(acc[i][j][r] + bv);
                if constexpr (EPI == 4) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float f = bf2f(v[r]); st_s += f; st_q += f * f; }
                }
            } else if constexpr (EPI == 1) {
                uint16_t w[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = f2bf(acc[i][j][r] + bv);
                    w[r] = f2bf(gelu_f(bf2f(v[r])));
                }
                *reinterpret_cast<uint2*>(Y2 + o) = make_uint2(w[0] | ((uint32_t)w[1] << 16), w[2] | ((uint32_t)w[3] << 16));
            } else {
                const uint2 a = *reinterpret_cast<const uint2*>(aux + o);
                const uint16_t x[4] = {(uint16_t)(a.x & 0xffffu), (uint16_t)(a.x >> 16), (uint16_t)(a.y & 0xffffu),
                                       (uint16_t)(a.y >> 16)};
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = f2bf(acc[i][j][r] * dgelu_f(bf2f(x[r])));
            }
            *reinterpret_cast<uint2*>(Y + o) = make_uint2(v[0] | ((uint32_t)v[1] << 16), v[2] | ((uint32_t)v[3] << 16));
        }
        if constexpr (EPI == 4) {
            // the four lane groups g hold the other pixels of channel m: fixed-order butterfly, lane group 0 writes
            st_s += __shfl_xor(st_s, 16, WAVE); st_q += __shfl_xor(st_q, 16, WAVE);
            st_s += __shfl_xor(st_s, 32, WAVE); st_q += __shfl_xor(st_q, 32, WAVE);
            if (g == 0) {
                // channel-major [M][P][2]: the finalize kernel reads a channel's P partials as one contiguous run
                const long P = (long)gridDim.z * gridDim.x * WN;
                float* sp = reinterpret_cast<float*>(Y2) + ((long)m * P + ((long)n * gridDim.x + blockIdx.x) * WN + wn) * 2;
                sp[0] = st_s; sp[1] = st_q;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// v2: the same GEMM with a three-stage LDS ring filled by LDS-DMA (`global_load_lds`, 16 bytes per lane, no staging
// registers), counted `s_waitcnt vmcnt(N)` + ONE raw `s_barrier` per K step (two K tiles in flight across it), an
// XCD-aware tile order and 16-byte epilogue stores.
//   * LDS-DMA writes wave-uniform base + lane * 16, so both tiles are dense images and every swizzle sits on the SOURCE
//     address (the inverse permutation of the fragment reads):
//       X tile [BK k][128 pixels] (256-byte rows): 16-byte slot s of row r holds pixel chunk s ^ (2 (r & 3) + 8 ((r >> 3) & 1))
//         -- the 8-byte swizzle of v1, which is a multiple of two 8-byte chunks; read with `ds_read_b64_tr_b16`;
//       A tile [BM m][BK k] (128- / 64-byte rows): slot s of row r holds k chunk s ^ ((r >> 1) & 7)  /  s ^ (2 ((r >> 2) & 1));
//         read with `ds_read_b128` (conflict free for its four 16-lane service groups, MI355X_MICROARCH.md LDS).
//   * a wave waits for ITS loads of stage t (vmcnt leaves the newer stage in flight), the barrier publishes all four
//     waves' pieces and retires everybody's reads of stage t - 1, whose buffer the loads for stage t + 2 then overwrite.
//   * epilogue: `v_permlane16_swap` between the packed results of two neighbouring 16-pixel tiles gives every lane 8
//     consecutive pixels of one channel: one 16-byte store instead of two 8-byte ones (a wave writes 64 contiguous bytes
//     per channel row).
//   * TA (the matrix given transposed, At [K][M]: data gradients use the forward weight as is): with BM = 128 the A tile
//     [BK k][128 m] has the X tile's shape, so it is staged with the X tile's source swizzle and its fragments come out of
//     the same transposing read -- the adapters' data gradients ran 17-22 us on v1's transposed mode against 5-9 us for
//     the same GEMM with a row-major matrix.
// Requirements: K % BK == 0, HW % 8 == 0; TA: BM = 128, M % 8 == 0.
constexpr int NSTAGE = 3;

template <int BK> __device__ __forceinline__ int a_swz16(int row) {
    return BK == 64 ? ((row >> 1) & 7) : (2 * ((row >> 2) & 1));
}

template <int BM, int BK, int EPI, bool TA = false>
__global__ __launch_bounds__(256) void pwconv2_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ X,
                                                      const void* __restrict__ bias, int bias_bf16,
                                                      const uint16_t* __restrict__ aux, uint16_t* __restrict__ Y,
                                                      uint16_t* __restrict__ Y2, int M, int K, int HW, int nb, int mtiles,
                                                      int total) {
    constexpr int WM = BM >= 64 ? 2 : 1, WN = 4 / WM;
    constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 16, NT = TN / 16;
    static_assert(!TA || BM == 128, "the transposed-A tile reuses the X tile's 256-byte-row layout");
    constexpr int ARB = BK * 2;                          // bytes per A row
    constexpr int A_BYTES = BM * ARB, X_BYTES = BK * B_STRIDE, STAGE = A_BYTES + X_BYTES;
    constexpr int A_INS = A_BYTES / 1024 / 4, X_INS = X_BYTES / 1024 / 4;      // LDS-DMA instructions per wave and stage
    static_assert(A_BYTES % 4096 == 0 && X_BYTES % 4096 == 0, "every wave issues the same number of loads per stage");
    constexpr int LPS = A_INS + X_INS, KH = BK / 32;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    // XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2); give every XCD a
    // contiguous run of logical tiles, channel tiles fastest, so the workgroups that share an X tile share an L2
    const int id = blockIdx.x, xcd = id & 7, q8 = total >> 3, r8 = total & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
    const int mt_i = L % mtiles, pt = L / mtiles;
    const int n = pt / nb, p0 = (pt - n * nb) * BN, m0 = mt_i * BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const uint16_t* Xn = X + (long)n * K * HW;

    // ---- LDS-DMA source addresses (per lane; advanced by BK channels per stage) and destinations (wave uniform) ----------
    const uint16_t* a_src[A_INS];
    const uint16_t* x_src[X_INS];
#pragma unroll
    for (int c = 0; c < A_INS; ++c) {
        const int t = c * 4 + wave;                              // 1-KiB piece of the A tile
        if constexpr (TA) {                                      // [BK k][128 m]: the X tile's layout and source swizzle
            const int row = 4 * t + (lane >> 4), slot = lane & 15;
            int m = m0 + 8 * (slot ^ (2 * (row & 3) + 8 * ((row >> 3) & 1)));
            if (m >= M) m = m0;                                  // chunks past M: any valid chunk (not stored; M % 8 == 0)
            a_src[c] = A + (long)row * M + m;
        } else {
            const int row = t * (1024 / ARB) + lane / (ARB / 16), slot = lane % (ARB / 16);
            const int m = min(m0 + row, M - 1);                  // rows past M: any valid row (their outputs are not stored)
            a_src[c] = A + (long)m * K + 8 * (slot ^ a_swz16<BK>(row));
        }
    }
#pragma unroll
    for (int c = 0; c < X_INS; ++c) {
        const int t = c * 4 + wave;
        const int row = 4 * t + (lane >> 4), slot = lane & 15;
        int p = p0 + 8 * (slot ^ (2 * (row & 3) + 8 * ((row >> 3) & 1)));
        if (p >= HW) p = p0;                                     // chunks past the plane: any valid chunk (not stored)
        x_src[c] = Xn + (long)row * HW + p;
    }
    // LDS-DMA through inline asm: hipcc's wait-count pass would otherwise drain EVERY pending LDS-DMA (`vmcnt(0)`) in front of
    // the first transposing LDS read after an issue -- it cannot tell the stage being filled from the stage being read --
    // which is the whole pipeline.  M0 (the LDS destination base) is written in the statement that uses it and restored.
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    auto glds16 = [&](const uint16_t* src, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    };
    auto issue = [&](int kt, int stage) {
        const unsigned base = __builtin_amdgcn_readfirstlane(lds_base + stage * STAGE + wave * 1024);
        const long ka = TA ? (long)kt * BK * M : (long)kt * BK, kx = (long)kt * BK * HW;
#pragma unroll
        for (int c = 0; c < A_INS; ++c) glds16(a_src[c] + ka, base + c * 4096);
#pragma unroll
        for (int c = 0; c < X_INS; ++c) glds16(x_src[c] + kx, base + A_BYTES + c * 4096);
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = {0.f, 0.f, 0.f, 0.f};

    // fragment read offsets inside a stage
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int b_rowi = 8 * g + q, b_s = b_swz(b_rowi);
    int a_off[MT][KH];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int h = 0; h < KH; ++h) {
            const int row = wm * TM + 16 * i + li;
            a_off[i][h] = row * ARB + 16 * ((4 * h + g) ^ a_swz16<BK>(row));
        }
    const int x_off = A_BYTES + b_rowi * B_STRIDE;

    const int nk = K / BK;
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) issue(kt + 2, cur >= 1 ? cur - 1 : NSTAGE - 1);       // (cur + 2) % 3: the buffer read one step ago
        const uint8_t* buf = lds + cur * STAGE;
#pragma unroll
        for (int h = 0; h < KH; ++h) {
            bf16x8 af[MT], bfr[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if constexpr (TA) {
                    const int chunk = (((wm * TM + 16 * i) >> 2) + pp) ^ b_s;
                    const uint8_t* ap = buf + b_rowi * B_STRIDE + h * 32 * B_STRIDE + (chunk << 3);
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ap));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ap + 4 * B_STRIDE));
                    typedef __attribute__((ext_vector_type(8))) short s16x8a;
                    const s16x8a both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    af[i] = __builtin_bit_cast(bf16x8, both);
                } else {
                    af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(buf + a_off[i][h]));
                }
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int chunk = (((wn * TN + 16 * j) >> 2) + pp) ^ b_s;
                const uint8_t* bp = buf + x_off + h * 32 * B_STRIDE + (chunk << 3);
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(bp));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(bp + 4 * B_STRIDE));
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                bfr[j] = __builtin_bit_cast(bf16x8, both);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        cur = cur == NSTAGE - 1 ? 0 : cur + 1;
    }

    // ---- epilogue: transposed C layout (col = lane & 15 = channel, rows 4 g + r = pixels); pairs of pixel tiles exchange
    // halves so that a lane stores 8 consecutive pixels
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + wm * TM + 16 * i + li;
        const bool m_ok = m < M;
        float bv = 0.f;
        if (bias != nullptr && m_ok)
            bv = bias_bf16 ? bf2f(reinterpret_cast<const uint16_t*>(bias)[m]) : reinterpret_cast<const float*>(bias)[m];
        float st_s = 0.f, st_q = 0.f;
        uint2 v[NT], w[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int p = p0 + wn * TN + 16 * j + 4 * g;
            const bool ok = m_ok && p < HW;
            const long o = (long)n * M * HW + (long)m * HW + p;
            uint16_t e[4], f[4] = {0, 0, 0, 0};
            if constexpr (EPI == 2) {
                uint2 a = make_uint2(0, 0);
                if (ok) a = *reinterpret_cast<const uint2*>(aux + o);
                const uint16_t x[4] = {(uint16_t)(a.x & 0xffffu), (uint16_t)(a.x >> 16), (uint16_t)(a.y & 0xffffu),
                                       (uint16_t)(a.y >> 16)};
#pragma unroll
                for (int r = 0; r < 4; ++r) e[r] = f2bf(acc[i][j][r] * dgelu_f(bf2f(x[r])));
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    e[r] = f2bf(acc[i][j][r] + bv);
                    if constexpr (EPI == 1) f[r] = f2bf(gelu_f(bf2f(e[r])));
                    if constexpr (EPI == 4) {
                        if (ok) { const float t = bf2f(e[r]); st_s += t; st_q += t * t; }
                    }
                }
            }
            v[j] = make_uint2(e[0] | ((uint32_t)e[1] << 16), e[2] | ((uint32_t)e[3] << 16));
            w[j] = make_uint2(f[0] | ((uint32_t)f[1] << 16), f[2] | ((uint32_t)f[3] << 16));
        }
#pragma unroll
        for (int j = 0; j < NT; j += 2) {
            // rows 1 / 3 of tile j <-> rows 0 / 2 of tile j + 1: lane group g then holds pixels 8 (g >> 1) .. + 7 of tile
            // j + (g & 1), the first four in the tile-j register, the next four in the tile-(j + 1) register
            auto sx = __builtin_amdgcn_permlane16_swap(v[j].x, v[j + 1].x, false, false);
            auto sy = __builtin_amdgcn_permlane16_swap(v[j].y, v[j + 1].y, false, false);
            const int p = p0 + wn * TN + 16 * (j + (g & 1)) + 8 * (g >> 1);
            const long o = (long)n * M * HW + (long)m * HW + p;
            if (m_ok && p < HW) *reinterpret_cast<uint4*>(Y + o) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
            if constexpr (EPI == 1) {
                auto tx = __builtin_amdgcn_permlane16_swap(w[j].x, w[j + 1].x, false, false);
                auto ty = __builtin_amdgcn_permlane16_swap(w[j].y, w[j + 1].y, false, false);
                if (m_ok && p < HW) *reinterpret_cast<uint4*>(Y2 + o) = make_uint4(tx[0], ty[0], tx[1], ty[1]);
            }
        }
        if constexpr (EPI == 4) {
            st_s += __shfl_xor(st_s, 16, WAVE); st_q += __shfl_xor(st_q, 16, WAVE);
            st_s += __shfl_xor(st_s, 32, WAVE); st_q += __shfl_xor(st_q, 32, WAVE);
            if (g == 0 && m_ok) {
                const long P = (long)(total / mtiles) * WN;
                float* sp = reinterpret_cast<float*>(Y2) + ((long)m * P + (long)pt * WN + wn) * 2;
                sp[0] = st_s; sp[1] = st_q;
            }
        }
    }
}

// ---- dispatch ----------------------------------------------------------------------------------------------------------
// One place decides the kernel and its tile: the statistics epilogue writes one partial per pixel-wave (WN), and
// ppea_pwconv_stats_partials must size that buffer for the kernel that will run.
struct PwCfg { int v2, bm, bk, wn; };

PwCfg pw_choose(int B, int M, int K, int HW, bool ta) {
    const int nb = (HW + BN - 1) / BN;
    const long blocks128 = (long)nb * ((M + 127) / 128) * B, blocks64 = (long)nb * ((M + 63) / 64) * B;
    PwCfg c;
    // tuning / test hooks, read per call so that a test can walk the tiles in one process
    const char* force = getenv("PPEA_PW_TILE");                 // v1 tile: "128", "64", "32" [+ "d" = 64-channel steps]
    const char* v2env = getenv("PPEA_PW_V2");                   // "0": v1 everywhere; "bm,bk": force a v2 tile
    if (force != nullptr) {
        c.bm = atoi(force) == 128 ? 128 : (atoi(force) == 64 ? 64 : 32);
        c.bk = strchr(force, 'd') != nullptr ? 64 : 32;
    } else {
        c.bm = (M >= 128 && blocks128 >= 512) ? 128 : ((M > 32 && blocks64 >= 256) ? 64 : 32);
        // 64 channels per step only where it pays (measured on v1): long contraction AND too few tiles to hide the
        // per-step latency by occupancy
        c.bk = (c.bm != 128 && K >= 512) ? 64 : 32;
    }
    c.v2 = 0;
    if (!ta && !(v2env != nullptr && v2env[0] == '0' && v2env[1] == 0)) {
        // measured per trunk shape (tools/bench_pw4.py, profiles/r03_pwconv_shapes.txt): enough 128-row tiles to fill the
        // chip once -> 128 x 128 tiles with 32-channel steps (48 KB of LDS: three workgroups per CU hide each other's
        // barriers); else 64-row tiles (64-channel steps for long contractions); else 32-row tiles with 64-channel steps
        int bm, bk;
        if (M >= 128 && blocks128 >= 256) { bm = 128; bk = 32; }
        else if (M > 32 && blocks64 >= 256) { bm = 64; bk = (K >= 512 && K % 64 == 0) ? 64 : 32; }
        else if (K % 64 == 0) { bm = 32; bk = 64; }
        else { bm = 64; bk = 32; }
        if (v2env != nullptr && strchr(v2env, ',') != nullptr) { bm = atoi(v2env); bk = atoi(strchr(v2env, ',') + 1); }
        if (bm == 32 && bk == 32) bk = 64;                       // a 32 x 32 A tile is half a piece per wave
        if ((bm == 128 || bm == 64 || bm == 32) && (bk == 64 || bk == 32) && K % bk == 0) { c.v2 = 1; c.bm = bm; c.bk = bk; }
    }
    c.wn = c.bm == 32 ? 4 : 2;
    return c;
}

template <int BM, int BK, int EPI, bool TA = false>
int launch_pw2_t(const void* A, const void* X, const void* bias, int bias_bf16, const void* aux, void* Y, void* Y2, int B,
                 int M, int K, int HW, hipStream_t st) {
    constexpr int smem = NSTAGE * (BM * BK * 2 + BK * B_STRIDE);
    static bool ready = false;                                   // per instantiation
    auto kern = pwconv2_kernel<BM, BK, EPI, TA>;
    if (!ready) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        ready = true;
    }
    const int nb = (HW + BN - 1) / BN, mtiles = (M + BM - 1) / BM;
    const long total = (long)nb * mtiles * B;
    if (total > 0x7fffffffL) return PPEA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), smem, st, (const uint16_t*)A, (const uint16_t*)X, bias, bias_bf16,
                       (const uint16_t*)aux, (uint16_t*)Y, (uint16_t*)Y2, M, K, HW, nb, mtiles, (int)total);
    return launch_status();
}

template <int EPI, bool TA>
int launch_pw(const void* A, const void* X, const void* bias, int bias_bf16, const void* aux, void* Y, void* Y2, int B,
              int M, int K, int HW, hipStream_t st) {
    const PwCfg c = pw_choose(B, M, K, HW, TA);
    if constexpr (!TA) {
        if (c.v2) {
#define PW2(BM_, BK_) return launch_pw2_t<BM_, BK_, EPI>(A, X, bias, bias_bf16, aux, Y, Y2, B, M, K, HW, st)
            if (c.bm == 128) { if (c.bk == 64) PW2(128, 64); else PW2(128, 32); }
            if (c.bm == 64) { if (c.bk == 64) PW2(64, 64); else PW2(64, 32); }
            PW2(32, 64);
#undef PW2
        }
    }
    if constexpr (TA) {
        // transposed matrix on the LDS-DMA ring: 128-row tiles only (see pwconv2_kernel); PPEA_PW_V2=0 keeps v1
        const char* v2env = getenv("PPEA_PW_V2");
        static const bool ta_v2 = !(getenv("PPEA_PW_TA_V2") != nullptr && getenv("PPEA_PW_TA_V2")[0] == '0');
        // ... where they give at least ~128 workgroups: a 128-channel result on 12 x 40 maps is 48 tiles with a long
        // contraction each, and v1's 32-row tiles (192 workgroups) are faster there (10 against 18 us, tools/bench_pw_ta.py)
        const long tiles128 = (long)((HW + BN - 1) / BN) * ((M + 127) / 128) * B;
        if (ta_v2 && !(v2env != nullptr && v2env[0] == '0' && v2env[1] == 0) && M >= 128 && tiles128 >= 128 && (M % 8) == 0 &&
            (K % 32) == 0)
            return launch_pw2_t<128, 32, EPI, true>(A, X, bias, bias_bf16, aux, Y, Y2, B, M, K, HW, st);
    }
    const int nb = (HW + BN - 1) / BN;
#define PW_LAUNCH(BM_, WM_, WN_, BK_)                                                                                 \
    hipLaunchKernelGGL((pwconv_kernel<BM_, WM_, WN_, EPI, TA, BK_>), dim3(nb, (M + BM_ - 1) / BM_, B), dim3(256), 0, st, \
                       (const uint16_t*)A, (const uint16_t*)X, bias, bias_bf16, (const uint16_t*)aux, (uint16_t*)Y, \
                       (uint16_t*)Y2, M, K, HW)
    if (c.bm == 128) PW_LAUNCH(128, 2, 2, 32);
    else if (c.bm == 64) { if (c.bk == 64) PW_LAUNCH(64, 2, 2, 64); else PW_LAUNCH(64, 2, 2, 32); }
    else { if (c.bk == 64) PW_LAUNCH(32, 1, 4, 64); else PW_LAUNCH(32, 1, 4, 32); }
#undef PW_LAUNCH
    return launch_status();
}

}  // namespace

extern "C" {

// A [M][K] bf16 row-major (k contiguous); X [B][K][HW] bf16; Y [B][M][HW] bf16; bias [M] fp32 or NULL.
// Requirements of the fast path: K % 32 == 0, HW % 8 == 0 (else PPEA_ERR_UNSUPPORTED).
int ppea_pwconv_bf16(const void* A, const void* X, const float* bias, void* Y, int B, int M, int K, int HW,
                     void* stream) {
    if (B <= 0 || M <= 0 || K <= 0 || HW <= 0 || (K % 32) != 0 || (HW % 8) != 0 || B > 65535)
        return PPEA_ERR_UNSUPPORTED;
    return launch_pw<0, false>(A, X, bias, 0, nullptr, Y, nullptr, B, M, K, HW, (hipStream_t)stream);
}

// ppea_pwconv_bf16 plus the statistics of the BatchNorm that follows (conv_bn / conv_bn_relu, rka.py:182-197): per output
// channel the sum and the sum of squares of the stored bf16 values, as P = ppea_pwconv_stats_partials(B, M, K, HW) partial
// pairs, stats [M][P][2] fp32 (every entry written; reduce with ppea_bn_finalize_sums_f32).
int ppea_pwconv_stats_partials(int B, int M, int K, int HW) {
    if (B <= 0 || M <= 0 || K <= 0 || HW <= 0) return 0;
    return B * ((HW + BN - 1) / BN) * pw_choose(B, M, K, HW, false).wn;
}
int ppea_pwconv_stats_bf16(const void* A, const void* X, const float* bias, void* Y, float* stats, int B, int M, int K,
                           int HW, void* stream) {
    if (B <= 0 || M <= 0 || K <= 0 || HW <= 0 || (K % 32) != 0 || (HW % 8) != 0 || B > 65535)
        return PPEA_ERR_UNSUPPORTED;
    if (stats == nullptr) return PPEA_ERR_ARG;
    return launch_pw<4, false>(A, X, bias, 0, nullptr, Y, stats, B, M, K, HW, (hipStream_t)stream);
}

// Same GEMM with an epilogue (adapters, replknet_adapter.py:20-109): epi 0 plain; epi 1 writes the
// pre-activation to Y and GELU(pre) to Y2; epi 2 multiplies by GELU'(aux) (aux, Y: [B][M][HW] bf16).
// `bias` is fp32 or, with bias_bf16 != 0, bf16.  a_transposed != 0: A is given as At [K][M] (M % 8 == 0).
int ppea_pwconv_ex_bf16(const void* A, const void* X, const void* bias, int bias_bf16, int epi, const void* aux,
                        void* Y, void* Y2, int B, int M, int K, int HW, int a_transposed, void* stream) {
    if (B <= 0 || M <= 0 || K <= 0 || HW <= 0 || (K % 32) != 0 || (HW % 8) != 0 || B > 65535)
        return PPEA_ERR_UNSUPPORTED;
    if (a_transposed && (M % 8) != 0) return PPEA_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (epi == 1 && !Y2) return PPEA_ERR_ARG;
    if (epi == 2 && !aux) return PPEA_ERR_ARG;
    if (a_transposed) {
        switch (epi) {
            case 0: return launch_pw<0, true>(A, X, bias, bias_bf16, nullptr, Y, nullptr, B, M, K, HW, st);
            case 1: return launch_pw<1, true>(A, X, bias, bias_bf16, nullptr, Y, Y2, B, M, K, HW, st);
            case 2: return launch_pw<2, true>(A, X, nullptr, 0, aux, Y, nullptr, B, M, K, HW, st);
        }
    } else {
        switch (epi) {
            case 0: return launch_pw<0, false>(A, X, bias, bias_bf16, nullptr, Y, nullptr, B, M, K, HW, st);
            case 1: return launch_pw<1, false>(A, X, bias, bias_bf16, nullptr, Y, Y2, B, M, K, HW, st);
            case 2: return launch_pw<2, false>(A, X, nullptr, 0, aux, Y, nullptr, B, M, K, HW, st);
        }
    }
    return PPEA_ERR_ARG;
}

}  // extern "C"
