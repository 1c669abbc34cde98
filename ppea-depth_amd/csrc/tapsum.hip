// 3x3 convolution of the B_Adapter (replknet_adapter.py:49-109, nn.Conv2d(C, C/4, 3, 1, 1)) split into an
// MFMA GEMM and a shift-and-add stencil:
//     T[b][t*Ch + m][p]   = sum_k W[m][k][t] * x[b][k][p]                 (pwconv.hip, M = 9*Ch rows)
//     pre[b][m][y][x]     = bias[m] + sum_{t=(dy,dx)} T[b][t*Ch + m][y+dy][x+dx]   (this file, zero padded)
//     h                   = GELU(pre)
// and, for the gradients, the adjoint scatter written as a gather:
//     dT[b][t*Ch + m][y][x] = g[b][m][y-dy][x-dx]
// Both kernels are pure data movement at HBM/L2 speed: one thread per 4 consecutive pixels, every vector
// access 8-byte aligned (W % 4 == 0), the +-1 column shifts come from one extra scalar load per side.
#include "common.h"

namespace {

__device__ __forceinline__ float bf2f(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

struct Px4 { uint16_t v[4]; };

__device__ __forceinline__ Px4 ld4(const uint16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    Px4 r;
    r.v[0] = (uint16_t)(u.x & 0xffff); r.v[1] = (uint16_t)(u.x >> 16);
    r.v[2] = (uint16_t)(u.y & 0xffff); r.v[3] = (uint16_t)(u.y >> 16);
    return r;
}
__device__ __forceinline__ void st4(uint16_t* p, uint16_t a, uint16_t b, uint16_t c, uint16_t d) {
    *reinterpret_cast<uint2*>(p) = make_uint2((uint32_t)a | ((uint32_t)b << 16), (uint32_t)c | ((uint32_t)d << 16));
}

__global__ __launch_bounds__(256) void tapsum_fwd_kernel(const uint16_t* __restrict__ T, const void* __restrict__ bias,
                                                         int bias_bf16, uint16_t* __restrict__ pre,
                                                         uint16_t* __restrict__ hout, int Ch, int H, int W, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // over B*Ch*H*(W/4)
    if (idx >= total) return;
    const int W4 = W >> 2;
    const int x0 = (int)(idx % W4) * 4;
    const int y = (int)((idx / W4) % H);
    const int m = (int)((idx / ((long)W4 * H)) % Ch);
    const long b = idx / ((long)W4 * H * Ch);
    float bv = 0.f;
    if (bias != nullptr)
        bv = bias_bf16 ? bf2f(reinterpret_cast<const uint16_t*>(bias)[m]) : reinterpret_cast<const float*>(bias)[m];
    float acc[4] = {bv, bv, bv, bv};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int t = (dy + 1) * 3 + (dx + 1);
            const uint16_t* row = T + (((b * 9 + t) * Ch + m) * H + yy) * (long)W;
            const Px4 c = ld4(row + x0);
            if (dx == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] += bf2f(c.v[i]);
            } else if (dx < 0) {
                const float left = (x0 > 0) ? bf2f(row[x0 - 1]) : 0.f;
                acc[0] += left; acc[1] += bf2f(c.v[0]); acc[2] += bf2f(c.v[1]); acc[3] += bf2f(c.v[2]);
            } else {
                const float right = (x0 + 4 < W) ? bf2f(row[x0 + 4]) : 0.f;
                acc[0] += bf2f(c.v[1]); acc[1] += bf2f(c.v[2]); acc[2] += bf2f(c.v[3]); acc[3] += right;
            }
        }
    }
    uint16_t p[4], h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { p[i] = f2bf(acc[i]); h[i] = f2bf(gelu_f(bf2f(p[i]))); }
    const long o = ((b * Ch + m) * H + y) * (long)W + x0;
    st4(pre + o, p[0], p[1], p[2], p[3]);
    st4(hout + o, h[0], h[1], h[2], h[3]);
}

__global__ __launch_bounds__(256) void tapsum_bwd_kernel(const uint16_t* __restrict__ g, uint16_t* __restrict__ dT,
                                                         int Ch, int H, int W, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // over B*Ch*H*(W/4) output chunks
    if (idx >= total) return;
    const int W4 = W >> 2;
    const int x0 = (int)(idx % W4) * 4;
    const int y = (int)((idx / W4) % H);
    const int m = (int)((idx / ((long)W4 * H)) % Ch);
    const long b = idx / ((long)W4 * H * Ch);
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int ys = y - dy;
        uint16_t v[6] = {0, 0, 0, 0, 0, 0};                 // g[ys][x0-1 .. x0+4]
        if (ys >= 0 && ys < H) {
            const uint16_t* row = g + ((b * Ch + m) * H + ys) * (long)W;
            const Px4 c = ld4(row + x0);
            v[1] = c.v[0]; v[2] = c.v[1]; v[3] = c.v[2]; v[4] = c.v[3];
            if (x0 > 0) v[0] = row[x0 - 1];
            if (x0 + 4 < W) v[5] = row[x0 + 4];
        }
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int t = (dy + 1) * 3 + (dx + 1);
            uint16_t* dst = dT + (((b * 9 + t) * Ch + m) * H + y) * (long)W + x0;
            // dT[..][y][x] = g[y-dy][x-dx]:  x - dx = x0 + i - dx  ->  v[1 + i - dx]
            st4(dst, v[1 - dx], v[2 - dx], v[3 - dx], v[4 - dx]);
        }
    }
}

}  // namespace

extern "C" {

// T [B][9*Ch][H][W] bf16 (tap-major rows t*Ch + m, t = 3*ky + kx) -> pre, h [B][Ch][H][W] bf16.  W % 4 == 0.
int ppea_tapsum_fwd_bf16(const void* T, const void* bias, int bias_bf16, void* pre, void* h, int B, int Ch, int H,
                         int W, void* stream) {
    if (B <= 0 || Ch <= 0 || H <= 0 || W <= 0 || (W % 4) != 0) return PPEA_ERR_UNSUPPORTED;
    const long total = (long)B * Ch * H * (W / 4);
    hipLaunchKernelGGL(tapsum_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)T, bias, bias_bf16, (uint16_t*)pre, (uint16_t*)h, Ch, H, W, total);
    return launch_status();
}

// g [B][Ch][H][W] bf16 -> dT [B][9*Ch][H][W] bf16 (every element written).
int ppea_tapsum_bwd_bf16(const void* g, void* dT, int B, int Ch, int H, int W, void* stream) {
    if (B <= 0 || Ch <= 0 || H <= 0 || W <= 0 || (W % 4) != 0) return PPEA_ERR_UNSUPPORTED;
    const long total = (long)B * Ch * H * (W / 4);
    hipLaunchKernelGGL(tapsum_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)g, (uint16_t*)dT, Ch, H, W, total);
    return launch_status();
}

}  // extern "C"
