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

// tile choice of the dispatch below as (rows per workgroup, waves along the pixels): the statistics epilogue writes one
// partial per pixel-wave, and the caller sizes / reduces that buffer
void pw_config(int B, int M, int K, int HW, int& bm, int& wn) {
    const int nb = (HW + BN - 1) / BN;
    const long blocks128 = (long)nb * ((M + 127) / 128) * B, blocks64 = (long)nb * ((M + 63) / 64) * B;
    static const char* force = getenv("PPEA_PW_TILE");
    if (force != nullptr) bm = atoi(force) == 128 ? 128 : (atoi(force) == 64 ? 64 : 32);
    else bm = (M >= 128 && blocks128 >= 512) ? 128 : ((M > 32 && blocks64 >= 256) ? 64 : 32);
    wn = bm == 32 ? 4 : 2;
}

template <int EPI, bool TA>
int launch_pw(const void* A, const void* X, const void* bias, int bias_bf16, const void* aux, void* Y, void* Y2, int B,
              int M, int K, int HW, hipStream_t st) {
    const int nb = (HW + BN - 1) / BN;
    const long blocks128 = (long)nb * ((M + 127) / 128) * B;
#define PW_LAUNCH(BM_, WM_, WN_, BK_)                                                                                 \
    hipLaunchKernelGGL((pwconv_kernel<BM_, WM_, WN_, EPI, TA, BK_>), dim3(nb, (M + BM_ - 1) / BM_, B), dim3(256), 0, st, \
                       (const uint16_t*)A, (const uint16_t*)X, bias, bias_bf16, (const uint16_t*)aux, (uint16_t*)Y, \
                       (uint16_t*)Y2, M, K, HW)
    const long blocks64 = (long)nb * ((M + 63) / 64) * B;
    // 64 channels per step only where it pays (measured): long contraction AND too few tiles to hide the per-step
    // latency by occupancy; with plenty of tiles the smaller LDS footprint (more workgroups per CU) wins.
    const bool deep = K >= 512;
    static const char* force = getenv("PPEA_PW_TILE");          // tuning hook (tools/bench_pw2.py): "128", "64", "32" [+ "d"]
    if (force != nullptr) {
        const int bm = atoi(force);
        const bool dp = strchr(force, 'd') != nullptr;
        if (bm == 128) PW_LAUNCH(128, 2, 2, 32);
        else if (bm == 64) { if (dp) PW_LAUNCH(64, 2, 2, 64); else PW_LAUNCH(64, 2, 2, 32); }
        else { if (dp) PW_LAUNCH(32, 1, 4, 64); else PW_LAUNCH(32, 1, 4, 32); }
        return launch_status();
    }
    if (M >= 128 && blocks128 >= 512) PW_LAUNCH(128, 2, 2, 32);
    else if (M > 32 && blocks64 >= 256) { if (deep) PW_LAUNCH(64, 2, 2, 64); else PW_LAUNCH(64, 2, 2, 32); }
    else { if (deep) PW_LAUNCH(32, 1, 4, 64); else PW_LAUNCH(32, 1, 4, 32); }   // few tiles: smaller workgroups
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
    int bm, wn;
    pw_config(B, M, K, HW, bm, wn);
    return B * ((HW + BN - 1) / BN) * wn;
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
