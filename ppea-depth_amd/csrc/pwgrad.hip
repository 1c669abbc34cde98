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
#include <cstdlib>

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

// ---------------------------------------------------------------------------------------------------------------------
// v2: the same contraction with a three-stage LDS ring filled by LDS-DMA (`global_load_lds`, 16 bytes per lane), counted
// `s_waitcnt vmcnt(N)` + ONE raw `s_barrier` per step (two steps in flight across it) -- the pipeline of pwconv v2
// (pwconv.hip).  Both operand tiles are [128 rows][BK pixels] row-major images read with `ds_read_b128`; LDS-DMA writes
// wave-uniform base + lane * 16, so the bank swizzle sits on the SOURCE address: slot s of row r holds pixel chunk
// s ^ ((r >> 1) & 7) (128-byte rows) / s ^ (2 ((r >> 2) & 1)) (64-byte rows).  The MFMA operands are swapped (the
// accumulator holds the transposed tile): a lane owns 4 consecutive columns n of one row m = one 16-byte slab store.
// Requirement: HW % BK == 0 (a step never straddles two images); other shapes stay on v1.
constexpr int NSTAGE2 = 3;

template <int BK> __device__ __forceinline__ int g_swz16(int row) { return BK == 64 ? ((row >> 1) & 7) : (2 * ((row >> 2) & 1)); }

template <int BK>
__device__ __forceinline__ void pwgrad2_body(const uint16_t* __restrict__ P, const uint16_t* __restrict__ Q,
                                             float* __restrict__ ws, int M, int N, int HW, int spi, int total_steps,
                                             int sps, bool rowsum_tile, int split, int m0, int n0) {
    constexpr int RB = BK * 2;                          // bytes per tile row
    constexpr int TILE = TM * RB, STAGE = 2 * TILE;
    constexpr int INS = TILE / 1024 / 4;                // LDS-DMA instructions per wave, operand and stage
    constexpr int LPS = 2 * INS, KH = BK / 32;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int s_begin = split * sps, s_end = min(s_begin + sps, total_steps);

    // per-lane source pointers inside image 0 (advanced per step by a wave-uniform offset)
    const uint16_t* p_src[INS];
    const uint16_t* q_src[INS];
#pragma unroll
    for (int c = 0; c < INS; ++c) {
        const int t = c * 4 + wave;
        const int row = t * (1024 / RB) + lane / (RB / 16), slot = lane % (RB / 16);
        const int chunk = slot ^ g_swz16<BK>(row);
        p_src[c] = P + (long)min(m0 + row, M - 1) * HW + 8 * chunk;          // rows past M / N: any valid row (not stored)
        q_src[c] = Q + (long)min(n0 + row, N - 1) * HW + 8 * chunk;
    }
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    auto glds16 = [&](const uint16_t* src, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    };
    auto issue = [&](int step, int stage) {
        const int b = step / spi, p0 = (step - b * spi) * BK;
        const long po = (long)b * M * HW + p0, qo = (long)b * N * HW + p0;
        const unsigned base = __builtin_amdgcn_readfirstlane(lds_base + stage * STAGE + wave * 1024);
#pragma unroll
        for (int c = 0; c < INS; ++c) glds16(p_src[c] + po, base + c * 4096);
#pragma unroll
        for (int c = 0; c < INS; ++c) glds16(q_src[c] + qo, base + TILE + c * 4096);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = {0.f, 0.f, 0.f, 0.f};
    float rs[4] = {0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, li = lane & 15;
    int a_off[4][KH], b_off[4][KH];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < KH; ++h) {
            const int ra = wm * 64 + 16 * i + li, rb = wn * 64 + 16 * i + li;
            a_off[i][h] = ra * RB + 16 * ((4 * h + g) ^ g_swz16<BK>(ra));
            b_off[i][h] = TILE + rb * RB + 16 * ((4 * h + g) ^ g_swz16<BK>(rb));
        }
    const bool do_rs = rowsum_tile && wn == 0;

    const int nsteps = s_end - s_begin;
    if (nsteps > 0) issue(s_begin, 0);
    if (nsteps > 1) issue(s_begin + 1, 1);
    int cur = 0;
    for (int k = 0; k < nsteps; ++k) {
        if (k + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (k + 2 < nsteps) issue(s_begin + k + 2, cur >= 1 ? cur - 1 : NSTAGE2 - 1);
        const uint8_t* buf = lds + cur * STAGE;
#pragma unroll
        for (int h = 0; h < KH; ++h) {
            uint4 au[4], bu[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) au[i] = *reinterpret_cast<const uint4*>(buf + a_off[i][h]);
#pragma unroll
            for (int j = 0; j < 4; ++j) bu[j] = *reinterpret_cast<const uint4*>(buf + b_off[j][h]);
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
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bu[j]),
                                                                        __builtin_bit_cast(bf16x8, au[i]), acc[i][j],
                                                                        0, 0, 0);
        }
        cur = cur == NSTAGE2 - 1 ? 0 : cur + 1;
    }

    // transposed C layout: col = lane & 15 = row m of the result, rows 4 g + r = 4 consecutive columns n
    float* out = ws + (long)split * ((long)M * N + M);
    const bool vec = (N & 3) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + 16 * i + li;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + 16 * j + 4 * g;
            if (n >= N) continue;
            float* o = out + (long)m * N + n;
            if (vec) {                                           // n % 4 == 0 and N % 4 == 0: all four inside, 16-byte aligned
                *reinterpret_cast<float4*>(o) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < N) o[r] = acc[i][j][r];
            }
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

template <int BK>
__global__ __launch_bounds__(256) void pwgrad2_kernel(const uint16_t* __restrict__ P, const uint16_t* __restrict__ Q,
                                                      float* __restrict__ ws, int M, int N, int HW, int spi,
                                                      int total_steps, int sps, int want_rowsum) {
    pwgrad2_body<BK>(P, Q, ws, M, N, HW, spi, total_steps, sps, want_rowsum && blockIdx.z == 0, blockIdx.x, blockIdx.y * TM,
                     blockIdx.z * TN);
}

// Two weight-gradient GEMMs over the same pixels in ONE launch (an adapter's D_fc2 and D_fc1 gradients: 0.75 GFLOP each, a
// launch's fixed latency several times the arithmetic): blockIdx.y walks the tiles of problem 0, then those of problem 1.
struct PwgProb { const uint16_t *P, *Q; float* ws; int M, N, want_rowsum, mt; };     // mt: tiles along M
template <int BK>
__global__ __launch_bounds__(256) void pwgrad2_pair_kernel(PwgProb a, PwgProb b, int tiles_a, int HW, int spi, int total_steps,
                                                           int sps) {
    int t = blockIdx.y;
    const bool second = t >= tiles_a;
    if (second) t -= tiles_a;
    const PwgProb& pr = second ? b : a;
    const int mi = t % pr.mt, ni = t / pr.mt;
    pwgrad2_body<BK>(pr.P, pr.Q, pr.ws, pr.M, pr.N, HW, spi, total_steps, sps, pr.want_rowsum && ni == 0, blockIdx.x, mi * TM,
                     ni * TN);
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
struct PwgOut { const float* ws; long pitch; int M, N, taps; void* out_w; int w_bf16; void* out_b; int b_bf16, b_row0, b_rows; };

__device__ __forceinline__ void pwgrad_reduce_ex_body(const PwgOut& o_, int S, long block) {
    const float* __restrict__ ws = o_.ws;
    const long pitch = o_.pitch;
    const int M = o_.M, N = o_.N, taps = o_.taps, w_bf16 = o_.w_bf16, b_bf16 = o_.b_bf16, b_row0 = o_.b_row0, b_rows = o_.b_rows;
    void* out_w = o_.out_w;
    void* out_b = o_.out_b;
    __shared__ float part[8][33];
    const int x = threadIdx.x & 31, y = threadIdx.x >> 5;
    const long len = (long)M * N + M;
    const long i = block * 32 + x;
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

__global__ __launch_bounds__(256) void pwgrad_reduce_ex_kernel(const float* __restrict__ ws, long pitch, int S, int M,
                                                               int N, int taps, void* __restrict__ out_w,
                                                               int w_bf16, void* __restrict__ out_b, int b_bf16,
                                                               int b_row0, int b_rows) {
    pwgrad_reduce_ex_body(PwgOut{ws, pitch, M, N, taps, out_w, w_bf16, out_b, b_bf16, b_row0, b_rows}, S, blockIdx.x);
}
__global__ __launch_bounds__(256) void pwgrad_reduce_ex2_kernel(PwgOut a, PwgOut b, int S, int blocks_a) {
    if ((int)blockIdx.x < blocks_a) pwgrad_reduce_ex_body(a, S, blockIdx.x);
    else pwgrad_reduce_ex_body(b, S, (long)blockIdx.x - blocks_a);
}

// v2 (LDS-DMA ring) serves planes whose size is a multiple of its 32-pixel step; PPEA_PWGRAD_V2=0: v1 everywhere
int pwgrad_bk(int HW) {
    const char* e = getenv("PPEA_PWGRAD_V2");
    if (e != nullptr && e[0] == '0') return 0;
    return (HW % 32) == 0 ? 32 : 0;
}

void plan(int B, int M, int N, int HW, int& spi, int& total, int& sps, int& S) {
    const int bk = pwgrad_bk(HW);
    const int tk = bk ? bk : TK;
    spi = (HW + tk - 1) / tk;
    total = B * spi;
    const int tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
    // enough work items to fill the chip once, but every split writes (and the reduce kernel re-reads) a full fp32 copy
    // of the result: v2's deeper pipeline needs fewer, longer items
    int want = ((bk ? 288 : 384) + tiles - 1) / tiles;
    const int min_steps = bk ? 8 : 4;
    if (want > (total + min_steps - 1) / min_steps) want = (total + min_steps - 1) / min_steps;
    if (want < 1) want = 1;
    sps = (total + want - 1) / want;
    S = (total + sps - 1) / sps;
}

int launch_main(const void* P, const void* Q, void* workspace, int B, int M, int N, int HW, int want_rowsum, hipStream_t st) {
    int spi, total, sps, S;
    plan(B, M, N, HW, spi, total, sps, S);
    dim3 grid(S, (M + TM - 1) / TM, (N + TN - 1) / TN);
    if (grid.y > 65535 || grid.z > 65535) return PPEA_ERR_UNSUPPORTED;
    if (pwgrad_bk(HW) == 32) {
        constexpr int smem = NSTAGE2 * 2 * TM * 64;
        hipLaunchKernelGGL(pwgrad2_kernel<32>, grid, dim3(256), smem, st, (const uint16_t*)P, (const uint16_t*)Q,
                           (float*)workspace, M, N, HW, spi, total, sps, want_rowsum);
    } else {
        hipLaunchKernelGGL(pwgrad_kernel, grid, dim3(256), 0, st, (const uint16_t*)P, (const uint16_t*)Q,
                           (float*)workspace, M, N, HW, spi, total, sps, want_rowsum);
    }
    return launch_status();
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
    const int err = launch_main(P, Q, workspace, B, M, N, HW, want_rowsum, st);
    if (err != 0) return err;
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
    const int err = launch_main(P, Q, workspace, B, M, N, HW, out_b != nullptr ? 1 : 0, st);
    if (err != 0) return err;
    const long pitch = (long)M * N + M;
    const long len = out_b != nullptr ? pitch : (long)M * N;
    hipLaunchKernelGGL(pwgrad_reduce_ex_kernel, dim3((unsigned)((len + 31) / 32)), dim3(256), 0, st,
                       (const float*)workspace, pitch, S, M, N, taps, out_w, out_w_bf16, out_b, out_b_bf16, b_row0,
                       b_rows);
    return launch_status();
}


// Two ppea_pwgrad_ex_bf16 problems over the same [B][.][HW] pixels in one GEMM launch and one reduce launch (an adapter's
// two weight gradients).  Arrays of two: P, Q, M, N, out_w, out_w_bf16, taps, out_b, out_b_bf16, b_row0, b_rows; workspace
// of ppea_pwgrad_pair_workspace_bytes (-1: not served).  PPEA_ERR_UNSUPPORTED (HW % 32 != 0 ...): call the single form twice.
static void pair_plan(int B, const int* M, const int* N, int HW, int& tiles, int& spi, int& total, int& sps, int& S) {
    tiles = ((M[0] + TM - 1) / TM) * ((N[0] + TN - 1) / TN) + ((M[1] + TM - 1) / TM) * ((N[1] + TN - 1) / TN);
    spi = HW / 32;
    total = B * spi;
    int want = (288 + tiles - 1) / tiles;
    if (want > (total + 7) / 8) want = (total + 7) / 8;
    if (want < 1) want = 1;
    sps = (total + want - 1) / want;
    S = (total + sps - 1) / sps;
}
long ppea_pwgrad_pair_workspace_bytes(int B, const int* M, const int* N, int HW) {
    if (B <= 0 || HW <= 0 || pwgrad_bk(HW) != 32 || M[0] <= 0 || M[1] <= 0 || N[0] <= 0 || N[1] <= 0) return -1;
    int tiles, spi, total, sps, S;
    pair_plan(B, M, N, HW, tiles, spi, total, sps, S);
    return (long)S * ((long)M[0] * N[0] + M[0] + (long)M[1] * N[1] + M[1]) * 4;
}
int ppea_pwgrad_ex_pair_bf16(const void* const* P, const void* const* Q, void* workspace, int B, const int* M, const int* N,
                             int HW, void* const* out_w, const int* out_w_bf16, const int* taps, void* const* out_b,
                             const int* out_b_bf16, const int* b_row0, const int* b_rows, void* stream) {
    if (B <= 0 || HW <= 0 || pwgrad_bk(HW) != 32) return PPEA_ERR_UNSUPPORTED;
    if (workspace == nullptr) return PPEA_ERR_ARG;
    for (int k = 0; k < 2; ++k) {
        if (P[k] == nullptr || Q[k] == nullptr) return PPEA_ERR_ARG;
        if (M[k] <= 0 || N[k] <= 0 || taps[k] < 1 || (M[k] % taps[k]) != 0) return PPEA_ERR_UNSUPPORTED;
        if (out_w[k] == nullptr || (out_b[k] != nullptr && (b_row0[k] < 0 || b_row0[k] + b_rows[k] > M[k]))) return PPEA_ERR_ARG;
    }
    int tiles, spi, total, sps, S;
    pair_plan(B, M, N, HW, tiles, spi, total, sps, S);
    if (tiles > 65535) return PPEA_ERR_UNSUPPORTED;
    const int mt0 = (M[0] + TM - 1) / TM, nt0 = (N[0] + TN - 1) / TN, mt1 = (M[1] + TM - 1) / TM;
    const long pitch0 = (long)M[0] * N[0] + M[0], pitch1 = (long)M[1] * N[1] + M[1];
    float* ws0 = (float*)workspace;
    float* ws1 = ws0 + (long)S * pitch0;
    hipStream_t st = (hipStream_t)stream;
    constexpr int smem = NSTAGE2 * 2 * TM * 64;
    const PwgProb a{(const uint16_t*)P[0], (const uint16_t*)Q[0], ws0, M[0], N[0], out_b[0] != nullptr ? 1 : 0, mt0};
    const PwgProb b{(const uint16_t*)P[1], (const uint16_t*)Q[1], ws1, M[1], N[1], out_b[1] != nullptr ? 1 : 0, mt1};
    hipLaunchKernelGGL(pwgrad2_pair_kernel<32>, dim3(S, tiles, 1), dim3(256), smem, st, a, b, mt0 * nt0, HW, spi, total, sps);
    const int err = launch_status();
    if (err != 0) return err;
    const long len0 = out_b[0] != nullptr ? pitch0 : (long)M[0] * N[0], len1 = out_b[1] != nullptr ? pitch1 : (long)M[1] * N[1];
    const int nb0 = (int)((len0 + 31) / 32), nb1 = (int)((len1 + 31) / 32);
    const PwgOut oa{ws0, pitch0, M[0], N[0], taps[0], out_w[0], out_w_bf16[0], out_b[0], out_b_bf16[0], b_row0[0], b_rows[0]};
    const PwgOut ob{ws1, pitch1, M[1], N[1], taps[1], out_w[1], out_w_bf16[1], out_b[1], out_b_bf16[1], b_row0[1], b_rows[1]};
    hipLaunchKernelGGL(pwgrad_reduce_ex2_kernel, dim3((unsigned)(nb0 + nb1)), dim3(256), 0, st, oa, ob, S, nb0);
    return launch_status();
}

}  // extern "C"
