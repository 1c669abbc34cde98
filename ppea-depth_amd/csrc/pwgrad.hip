// Pixel-contraction GEMM on the matrix cores: the weight gradients of every 1x1 / Linear / tap-expanded
// 3x3 layer of the adapters (replknet_adapter.py:20-109), NCHW bf16 operands, fp32 result.
//
//     C[m][n] = sum_b sum_p P[b][m][p] * Q[b][n][p]           rowsum[m] = sum_b sum_p P[b][m][p]
//
// Both operands are contracted over their CONTIGUOUS axis (pixels), so the MFMA fragments (one row, 8
// consecutive k) are plain 16-byte reads of the staged tiles -- no transposition anywhere.
//   * work item = (128 x 128 tile of C) x (split of the B*HW contraction); 4 waves, 64 x 64 per wave,
//     `v_mfma_f32_16x16x32_bf16`, 64 pixels per step, global -> register prefetch of the next step;
//   * LDS tiles [128 rows][64 px] with a 160-byte row stride (== 32 * odd: conflict-free b128 reads);
//   * split partial results go to a workspace [S][M*N + M] fp32 with plain coalesced stores and are summed
//     by a second tiny kernel: deterministic, and no cross-XCD atomics;
//   * the row sums (bias gradients) ride along in the n-tile-0 workgroups.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int TM = 128, TN = 128, TK = 64;
constexpr int STRIDE = 160;                      // bytes per LDS row (128 data + 32 pad)

__device__ __forceinline__ float bf_lo(uint32_t v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf_hi(uint32_t v) { return __builtin_bit_cast(float, v & 0xffff0000u); }

__global__ __launch_bounds__(256) void pwgrad_kernel(const uint16_t* __restrict__ P, const uint16_t* __restrict__ Q,
                                                     float* __restrict__ ws, int M, int N, int HW, int spi,
                                                     int total_steps, int sps, int want_rowsum) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * TM * STRIDE];
    uint8_t* Ps = lds;
    uint8_t* Qs = lds + TM * STRIDE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int split = blockIdx.x, m0 = blockIdx.y * TM, n0 = blockIdx.z * TN;
    const int s_begin = split * sps, s_end = min(s_begin + sps, total_steps);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = {0.f, 0.f, 0.f, 0.f};
    float rs[4] = {0.f, 0.f, 0.f, 0.f};

    // staging: 128 rows x 8 chunks of 16 B per operand = 1024 chunks, 4 per thread
    int row[4], ch[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { const int idx = tid + c * 256; row[c] = idx >> 3; ch[c] = idx & 7; }
    uint4 preg[4], qreg[4];
    auto load_tiles = [&](int step) {
        const int b = step / spi, p0 = (step - b * spi) * TK;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int p = p0 + ch[c] * 8;
            const int m = m0 + row[c], n = n0 + row[c];
            preg[c] = (m < M && p < HW) ? *reinterpret_cast<const uint4*>(P + ((long)b * M + m) * HW + p)
                                        : make_uint4(0, 0, 0, 0);
            qreg[c] = (n < N && p < HW) ? *reinterpret_cast<const uint4*>(Q + ((long)b * N + n) * HW + p)
                                        : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            *reinterpret_cast<uint4*>(Ps + row[c] * STRIDE + ch[c] * 16) = preg[c];
            *reinterpret_cast<uint4*>(Qs + row[c] * STRIDE + ch[c] * 16) = qreg[c];
        }
    };

    const int g = lane >> 4, li = lane & 15;
    const uint8_t* a_frag = Ps + (wm * 64 + li) * STRIDE + g * 16;
    const uint8_t* b_frag = Qs + (wn * 64 + li) * STRIDE + g * 16;
    const bool do_rs = want_rowsum && blockIdx.z == 0 && wn == 0;

    if (s_begin < s_end) load_tiles(s_begin);
    for (int step = s_begin; step < s_end; ++step) {
        __syncthreads();
        store_tiles();
        __syncthreads();
        if (step + 1 < s_end) load_tiles(step + 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint4 au[4], bu[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) au[i] = *reinterpret_cast<const uint4*>(a_frag + i * 16 * STRIDE + h * 64);
#pragma unroll
            for (int j = 0; j < 4; ++j) bu[j] = *reinterpret_cast<const uint4*>(b_frag + j * 16 * STRIDE + h * 64);
            if (do_rs) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    rs[i] += (bf_lo(au[i].x) + bf_hi(au[i].x)) + (bf_lo(au[i].y) + bf_hi(au[i].y)) +
                             (bf_lo(au[i].z) + bf_hi(au[i].z)) + (bf_lo(au[i].w) + bf_hi(au[i].w));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, au[i]),
                                                                        __builtin_bit_cast(bf16x8, bu[j]), acc[i][j],
                                                                        0, 0, 0);
        }
    }

    // partial tile -> workspace slice of this split: C layout col = lane & 15 (n), row = 4 * (lane >> 4) + r (m)
    float* out = ws + (long)split * ((long)M * N + M);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * 64 + 16 * i + 4 * g + r;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + 16 * j + li;
                if (n < N) out[(long)m * N + n] = acc[i][j][r];
            }
        }
    if (do_rs) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = rs[i];
            v += __shfl_xor(v, 16, WAVE);
            v += __shfl_xor(v, 32, WAVE);
            const int m = m0 + wm * 64 + 16 * i + li;
            if (g == 0 && m < M) out[(long)M * N + m] = v;
        }
    }
}

// out[i] = sum_s ws[s * pitch + i], i < len.  Block = 32 columns x 8 split groups, combined through LDS.
__global__ __launch_bounds__(256) void pwgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                            long len, long pitch, int S) {
    __shared__ float part[8][33];
    const int x = threadIdx.x & 31, y = threadIdx.x >> 5;
    const long i = (long)blockIdx.x * 32 + x;
    float v = 0.f;
    if (i < len)
        for (int s = y; s < S; s += 8) v += ws[(long)s * pitch + i];
    part[y][x] = v;
    __syncthreads();
    if (y == 0 && i < len) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += part[k][x];
        out[i] = t;
    }
}

// Same reduction, written straight into the parameter gradients: weights as fp32 or bf16, either [M][N] or,
// for the tap-major 3x3 form (taps = 9, M = 9 * Ch rows t * Ch + m), in nn.Conv2d layout [Ch][N][3][3];
// bias = row sums of rows [b_row0, b_row0 + b_rows).
__global__ __launch_bounds__(256) void pwgrad_reduce_ex_kernel(const float* __restrict__ ws, long pitch, int S, int M,
                                                               int N, int taps, void* __restrict__ out_w,
                                                               int w_bf16, void* __restrict__ out_b, int b_bf16,
                                                               int b_row0, int b_rows) {
    __shared__ float part[8][33];
    const int x = threadIdx.x & 31, y = threadIdx.x >> 5;
    const long len = (long)M * N + M;
    const long i = (long)blockIdx.x * 32 + x;
    float v = 0.f;
    if (i < len)
        for (int s = y; s < S; s += 8) v += ws[(long)s * pitch + i];
    part[y][x] = v;
    __syncthreads();
    if (y != 0 || i >= len) return;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += part[k][x];
    if (i < (long)M * N) {
        const int m = (int)(i / N), n = (int)(i - (long)m * N);
        long o = i;
        if (taps > 1) {
            const int Ch = M / taps, tp = m / Ch, mo = m - tp * Ch;
            o = ((long)mo * N + n) * taps + tp;
        }
        if (w_bf16) reinterpret_cast<uint16_t*>(out_w)[o] = __builtin_bit_cast(uint16_t, (__bf16)t);
        else reinterpret_cast<float*>(out_w)[o] = t;
    } else if (out_b != nullptr) {
        const int r = (int)(i - (long)M * N) - b_row0;
        if (r >= 0 && r < b_rows) {
            if (b_bf16) reinterpret_cast<uint16_t*>(out_b)[r] = __builtin_bit_cast(uint16_t, (__bf16)t);
            else reinterpret_cast<float*>(out_b)[r] = t;
        }
    }
}

void plan(int B, int M, int N, int HW, int& spi, int& total, int& sps, int& S) {
    spi = (HW + TK - 1) / TK;
    total = B * spi;
    const int tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
    int want = (384 + tiles - 1) / tiles;
    if (want > (total + 3) / 4) want = (total + 3) / 4;      // at least ~4 steps per work item
    if (want < 1) want = 1;
    sps = (total + want - 1) / want;
    S = (total + sps - 1) / sps;
}

}  // namespace

extern "C" {

// Bytes of workspace ppea_pwgrad_bf16 needs for these shapes.
long ppea_pwgrad_workspace_bytes(int B, int M, int N, int HW) {
    int spi, total, sps, S;
    plan(B, M, N, HW, spi, total, sps, S);
    return (long)S * ((long)M * N + M) * 4;
}

// out [M*N + M] fp32: C row-major followed by the row sums of P (written only when want_rowsum != 0).
// P [B][M][HW], Q [B][N][HW] bf16; HW % 8 == 0.
int ppea_pwgrad_bf16(const void* P, const void* Q, float* out, void* workspace, int B, int M, int N, int HW,
                     int want_rowsum, void* stream) {
    if (B <= 0 || M <= 0 || N <= 0 || HW <= 0 || (HW % 8) != 0) return PPEA_ERR_UNSUPPORTED;
    int spi, total, sps, S;
    plan(B, M, N, HW, spi, total, sps, S);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(S, (M + TM - 1) / TM, (N + TN - 1) / TN);
    if (grid.y > 65535 || grid.z > 65535) return PPEA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pwgrad_kernel, grid, dim3(256), 0, st, (const uint16_t*)P, (const uint16_t*)Q,
                       (float*)workspace, M, N, HW, spi, total, sps, want_rowsum);
    const long len = (long)M * N + (want_rowsum ? M : 0), pitch = (long)M * N + M;
    hipLaunchKernelGGL(pwgrad_reduce_kernel, dim3((unsigned)((len + 31) / 32)), dim3(256), 0, st,
                       (const float*)workspace, out, len, pitch, S);
    return launch_status();
}

// As ppea_pwgrad_bf16, results written straight into the parameter gradients (see pwgrad_reduce_ex_kernel).
int ppea_pwgrad_ex_bf16(const void* P, const void* Q, void* workspace, int B, int M, int N, int HW, void* out_w,
                        int out_w_bf16, int taps, void* out_b, int out_b_bf16, int b_row0, int b_rows, void* stream) {
    if (B <= 0 || M <= 0 || N <= 0 || HW <= 0 || (HW % 8) != 0 || taps < 1 || (M % taps) != 0)
        return PPEA_ERR_UNSUPPORTED;
    if (out_w == nullptr || (out_b != nullptr && (b_row0 < 0 || b_row0 + b_rows > M))) return PPEA_ERR_ARG;
    int spi, total, sps, S;
    plan(B, M, N, HW, spi, total, sps, S);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(S, (M + TM - 1) / TM, (N + TN - 1) / TN);
    if (grid.y > 65535 || grid.z > 65535) return PPEA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pwgrad_kernel, grid, dim3(256), 0, st, (const uint16_t*)P, (const uint16_t*)Q,
                       (float*)workspace, M, N, HW, spi, total, sps, out_b != nullptr ? 1 : 0);
    const long pitch = (long)M * N + M;
    const long len = out_b != nullptr ? pitch : (long)M * N;
    hipLaunchKernelGGL(pwgrad_reduce_ex_kernel, dim3((unsigned)((len + 31) / 32)), dim3(256), 0, st,
                       (const float*)workspace, pitch, S, M, N, taps, out_w, out_w_bf16, out_b, out_b_bf16, b_row0,
                       b_rows);
    return launch_status();
}

}  // extern "C"
