// Training-mode batch norm fused with its neighbours, gfx950 (HBM-bound; every tensor crosses HBM once
// per kernel).  Replaces, per call site of networks/replknet_adapter.py (conv_bn / conv_bn_relu :182-197,
// ReparamLargeKernelConv.forward :232-239, RepLKBlock.forward :315-326, ConvFFN.forward :283-289):
//     y = act( BN_a(z1) [+ BN_b(z2)] ) [* mask[n]] [+ r1] [+ s * r2]
// i.e. batch-norm (batch statistics), the optional second re-param branch, ReLU / GELU, the DropPath
// per-sample scale, the residual and the adapter branch -- 6-8 ATen kernels -- in
//   forward : bn_stats (per-plane mean / M2)  ->  bn_finalize (per channel, updates running stats)
//             -> bn_apply (one elementwise pass);
//   backward: bn_bwd_reduce (per-plane sums of g and g*zhat) -> bn_bwd_finalize -> bn_bwd_apply.
// Statistics: the kernels that READ the activation (bn_stats, bn_stats_channel*, the channel kernels) take per-plane /
// per-thread two-pass (mean, M2) and combine with Chan's parallel-variance formula: no E[x^2] - mean^2 cancellation.
// The kernels that take their statistics from a PRODUCER's epilogue (bn_finalize_sums, bn_fwd_channel_sums; bn_sync.hip
// bn_sums_to_packed; the fused-input finalize of dwconv_mfma.hip) receive fp32 per-tile (sum, sum of squares) partials,
// add them in fp64 in a fixed order and form var = E[x^2] - mean^2 in fp64: exact up to the fp32 rounding of each tile's
// partial, i.e. a relative variance error of ~1e-7 * (1 + mean^2 / var) -- 1e-3 for a channel with |mean| = 100 std
// (pinned by tests/test_kernels_gpu.py::test_bn_statistics_from_epilogue_sums_with_a_large_mean; post-BatchNorm
// activations of this network have |mean| / std of order 1).  Layout NCHW, T = float or bf16 (fp32 math).
#include "bn_common.h"
#include <cstdlib>

namespace {

// partial[(c*N + n)*2 + {0,1}] = (mean, M2) of plane (n, c)
template <typename T>
__global__ __launch_bounds__(TPB) void bn_stats(const T* __restrict__ z, float* __restrict__ partial, int N, int C,
                                                int HW) {
    __shared__ float red[4];
    const int n = blockIdx.x / C, c = blockIdx.x - n * C;
    const T* p = z + (long)blockIdx.x * HW;
    const int hv = ((HW % V) == 0) ? HW : 0;            // 16-byte path needs aligned planes
    float s = 0.f;
    for (int i = threadIdx.x * V; i < hv; i += TPB * V) {
        float x[V];
        ld8<T>(p + i, x);
#pragma unroll
        for (int k = 0; k < V; ++k) s += x[k];
    }
    for (int i = hv + threadIdx.x; i < HW; i += TPB) s += ld_f32<T>(p + i);
    const float mean = block_sum(s, red) / (float)HW;
    float m2 = 0.f;
    for (int i = threadIdx.x * V; i < hv; i += TPB * V) {
        float x[V];
        ld8<T>(p + i, x);
#pragma unroll
        for (int k = 0; k < V; ++k) m2 += (x[k] - mean) * (x[k] - mean);
    }
    for (int i = hv + threadIdx.x; i < HW; i += TPB) {
        const float d = ld_f32<T>(p + i) - mean;
        m2 += d * d;
    }
    m2 = block_sum(m2, red);
    if (threadIdx.x == 0) {
        partial[((long)c * N + n) * 2] = mean;
        partial[((long)c * N + n) * 2 + 1] = m2;
    }
}

__global__ void bn_finalize(const float* __restrict__ partial, int N, int C, int HW, float eps, float momentum,
                            float* __restrict__ mean_out, float* __restrict__ var_out, float* __restrict__ invstd_out,
                            float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float* p = partial + (long)c * N * 2;
    float mean = 0.f;
    for (int n = 0; n < N; ++n) mean += p[2 * n];
    mean /= (float)N;
    float m2 = 0.f;
    for (int n = 0; n < N; ++n) {
        const float d = p[2 * n] - mean;
        m2 += p[2 * n + 1] + (float)HW * d * d;
    }
    const float cnt = (float)N * (float)HW;
    const float var = m2 / cnt;
    mean_out[c] = mean;
    var_out[c] = var;
    invstd_out[c] = rsqrtf(var + eps);
    if (running_mean != nullptr) {
        const float unbiased = m2 / fmaxf(cnt - 1.f, 1.f);
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

// Statistics from per-channel partial sums written by the producing GEMM's epilogue (ppea_pwconv_stats_bf16):
// partial [C][P][2] = (sum, sum of squares) of disjoint pixel sets, channel-major.  One wave per channel reads its P pairs
// as one contiguous run; the partials are fp32 sums of <= 64 bf16 values, the totals are taken in fp64 (var = E[x^2] -
// mean^2 without cancellation trouble) in a fixed order.
__global__ __launch_bounds__(256) void bn_finalize_sums(const float* __restrict__ partial, int P, int C, float count,
                                                        float eps, float momentum, float* __restrict__ mean_out,
                                                        float* __restrict__ var_out, float* __restrict__ invstd_out,
                                                        float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    const float2* p = reinterpret_cast<const float2*>(partial) + (long)c * P;
    double s = 0.0, q = 0.0;
    // (unrolled: the loads of eight iterations are in flight together; the additions keep their order)
#pragma unroll 8
    for (int i = lane; i < P; i += 64) {
        const float2 v = p[i];
        s += (double)v.x; q += (double)v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, WAVE); q += __shfl_xor(q, o, WAVE); }
    if (lane == 0) {
        const double mean = s / (double)count;
        double var = q / (double)count - mean * mean;
        var = var > 0.0 ? var : 0.0;
        mean_out[c] = (float)mean;
        var_out[c] = (float)var;
        invstd_out[c] = rsqrtf((float)var + eps);
        if (running_mean != nullptr) {
            const float unbiased = (float)(var * (double)count / fmax((double)count - 1.0, 1.0));
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
        }
    }
}

// SyncBN (several ranks): local statistics in the layout that goes on the wire, packed[2C+1] = mean[C] |
// biased var[C] | count, and the Chan combine of the gathered [world][2C+1] table (+ running statistics).
__global__ void bn_finalize_packed(const float* __restrict__ partial, int N, int C, int HW,
                                   float* __restrict__ packed) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0) packed[2 * C] = (float)N * (float)HW;
    if (c >= C) return;
    const float* p = partial + (long)c * N * 2;
    float mean = 0.f;
    for (int n = 0; n < N; ++n) mean += p[2 * n];
    mean /= (float)N;
    float m2 = 0.f;
    for (int n = 0; n < N; ++n) {
        const float d = p[2 * n] - mean;
        m2 += p[2 * n + 1] + (float)HW * d * d;
    }
    packed[c] = mean;
    packed[C + c] = m2 / ((float)N * (float)HW);
}

__global__ void bn_sync_combine(const float* __restrict__ gathered, int world, int C, float eps, float momentum,
                                float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int pitch = 2 * C + 1;
    float total = 0.f, mean = 0.f;
    for (int r = 0; r < world; ++r) {
        const float cnt = gathered[(long)r * pitch + 2 * C];
        total += cnt;
        mean += cnt * gathered[(long)r * pitch + c];
    }
    mean /= total;
    float m2 = 0.f;
    for (int r = 0; r < world; ++r) {
        const float cnt = gathered[(long)r * pitch + 2 * C];
        const float d = gathered[(long)r * pitch + c] - mean;
        m2 += cnt * (gathered[(long)r * pitch + C + c] + d * d);
    }
    mean_out[c] = mean;
    invstd_out[c] = rsqrtf(m2 / total + eps);
    if (running_mean != nullptr) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (m2 / fmaxf(total - 1.f, 1.f));
    }
}

// one block per (plane, chunk); VEC elements per thread per iteration
template <typename T>
__global__ __launch_bounds__(TPB) void bn_apply(const T* __restrict__ z1, const T* __restrict__ z2, Branch b1,
                                                Branch b2, const float* __restrict__ mask, const T* __restrict__ r1,
                                                const T* __restrict__ r2, float r2_scale, T* __restrict__ y, int act,
                                                int C, int HW, int chunks) {
    const int plane = blockIdx.x / chunks, chunk = blockIdx.x - plane * chunks;
    const int n = plane / C, c = plane - n * C;
    const float a1 = b1.gamma[c] * b1.invstd[c], o1 = b1.beta[c] - b1.mean[c] * a1;
    float a2 = 0.f, o2 = 0.f;
    if (z2 != nullptr) { a2 = b2.gamma[c] * b2.invstd[c]; o2 = b2.beta[c] - b2.mean[c] * a2; }
    const float m = (mask != nullptr) ? mask[n] : 1.f;
    const long base = (long)plane * HW;
    const int per = (((HW + chunks - 1) / chunks) + V - 1) / V * V;
    const int i0 = chunk * per, i1 = min(HW, i0 + per);
    const int iv = ((HW % V) == 0) ? i1 : i0;
    for (int i = i0 + threadIdx.x * V; i < iv; i += TPB * V) {
        float x1[V], x2[V], q1[V], q2[V], o[V];
        ld8<T>(z1 + base + i, x1);
        if (z2 != nullptr) ld8<T>(z2 + base + i, x2);
        if (r1 != nullptr) ld8<T>(r1 + base + i, q1);
        if (r2 != nullptr) ld8<T>(r2 + base + i, q2);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float u = a1 * x1[k] + o1;
            if (z2 != nullptr) u += a2 * x2[k] + o2;
            float v = act_fwd<T>(u, act) * m;
            if (r1 != nullptr) v += q1[k];
            if (r2 != nullptr) v += r2_scale * q2[k];
            o[k] = v;
        }
        st8<T>(y + base + i, o);
    }
    for (int i = iv + threadIdx.x; i < i1; i += TPB) {
        float u = a1 * ld_f32<T>(z1 + base + i) + o1;
        if (z2 != nullptr) u += a2 * ld_f32<T>(z2 + base + i) + o2;
        float v = act_fwd<T>(u, act) * m;
        if (r1 != nullptr) v += ld_f32<T>(r1 + base + i);
        if (r2 != nullptr) v += r2_scale * ld_f32<T>(r2 + base + i);
        st_f32<T>(y + base + i, v);
    }
}

// partial[(c*N + n)*3 + {0,1,2}] = sum g, sum g*zhat1, sum g*zhat2   with g = dy * mask[n] * act'(u)
template <typename T>
__global__ __launch_bounds__(TPB) void bn_bwd_reduce(const T* __restrict__ dy, const T* __restrict__ z1,
                                                     const T* __restrict__ z2, Branch b1, Branch b2,
                                                     const float* __restrict__ mask, float* __restrict__ partial,
                                                     int act, int N, int C, int HW) {
    __shared__ float red[4];
    const int n = blockIdx.x / C, c = blockIdx.x - n * C;
    const float is1 = b1.invstd[c], mu1 = b1.mean[c];
    const float a1 = b1.gamma[c] * is1, o1 = b1.beta[c] - mu1 * a1;
    float is2 = 0.f, mu2 = 0.f, a2 = 0.f, o2 = 0.f;
    if (z2 != nullptr) { is2 = b2.invstd[c]; mu2 = b2.mean[c]; a2 = b2.gamma[c] * is2; o2 = b2.beta[c] - mu2 * a2; }
    const float m = (mask != nullptr) ? mask[n] : 1.f;
    const long base = (long)blockIdx.x * HW;
    float sg = 0.f, s1 = 0.f, s2 = 0.f;
    const int hv = ((HW % V) == 0) ? HW : 0;
    for (int i = threadIdx.x * V; i < hv; i += TPB * V) {
        float x1[V], x2[V], d[V];
        ld8<T>(z1 + base + i, x1);
        if (z2 != nullptr) ld8<T>(z2 + base + i, x2);
        ld8<T>(dy + base + i, d);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float u = a1 * x1[k] + o1;
            float xx2 = 0.f;
            if (z2 != nullptr) { xx2 = x2[k]; u += a2 * xx2 + o2; }
            const float g = d[k] * m * act_bwd<T>(u, act);
            sg += g;
            s1 += g * (x1[k] - mu1) * is1;
            s2 += g * (xx2 - mu2) * is2;
        }
    }
    for (int i = hv + threadIdx.x; i < HW; i += TPB) {
        const float x1 = ld_f32<T>(z1 + base + i);
        float u = a1 * x1 + o1;
        float x2 = 0.f;
        if (z2 != nullptr) { x2 = ld_f32<T>(z2 + base + i); u += a2 * x2 + o2; }
        const float g = ld_f32<T>(dy + base + i) * m * act_bwd<T>(u, act);
        sg += g;
        s1 += g * (x1 - mu1) * is1;
        s2 += g * (x2 - mu2) * is2;
    }
    sg = block_sum(sg, red);
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        float* p = partial + ((long)c * N + n) * 3;
        p[0] = sg; p[1] = s1; p[2] = s2;
    }
}

// sums[0*C + c] = sum g, sums[1*C + c] = sum g*zhat1, sums[2*C + c] = sum g*zhat2
__global__ void bn_bwd_finalize(const float* __restrict__ partial, int N, int C, float* __restrict__ sums) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float* p = partial + (long)c * N * 3;
    float a = 0.f, b = 0.f, d = 0.f;
    for (int n = 0; n < N; ++n) { a += p[3 * n]; b += p[3 * n + 1]; d += p[3 * n + 2]; }
    sums[c] = a; sums[C + c] = b; sums[2 * C + c] = d;
}

// dz_k = gamma_k * invstd_k * (g - sum_g / M - zhat_k * sum_gz_k / M),  M = count (global batch)
template <typename T>
__global__ __launch_bounds__(TPB) void bn_bwd_apply(const T* __restrict__ dy, const T* __restrict__ z1,
                                                    const T* __restrict__ z2, Branch b1, Branch b2,
                                                    const float* __restrict__ mask, const float* __restrict__ sums,
                                                    float inv_count, T* __restrict__ dz1, T* __restrict__ dz2,
                                                    int act, int C, int HW, int chunks) {
    const int plane = blockIdx.x / chunks, chunk = blockIdx.x - plane * chunks;
    const int n = plane / C, c = plane - n * C;
    const float is1 = b1.invstd[c], mu1 = b1.mean[c];
    const float a1 = b1.gamma[c] * is1, o1 = b1.beta[c] - mu1 * a1;
    float is2 = 0.f, mu2 = 0.f, a2 = 0.f, o2 = 0.f;
    if (z2 != nullptr) { is2 = b2.invstd[c]; mu2 = b2.mean[c]; a2 = b2.gamma[c] * is2; o2 = b2.beta[c] - mu2 * a2; }
    const float m = (mask != nullptr) ? mask[n] : 1.f;
    const float mg = sums[c] * inv_count, m1 = sums[C + c] * inv_count, m2 = sums[2 * C + c] * inv_count;
    const long base = (long)plane * HW;
    const int per = (((HW + chunks - 1) / chunks) + V - 1) / V * V;
    const int i0 = chunk * per, i1 = min(HW, i0 + per);
    const int iv = ((HW % V) == 0) ? i1 : i0;
    for (int i = i0 + threadIdx.x * V; i < iv; i += TPB * V) {
        float x1[V], x2[V], d[V], o1v[V], o2v[V];
        ld8<T>(z1 + base + i, x1);
        if (z2 != nullptr) ld8<T>(z2 + base + i, x2);
        ld8<T>(dy + base + i, d);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float u = a1 * x1[k] + o1;
            float xx2 = 0.f;
            if (z2 != nullptr) { xx2 = x2[k]; u += a2 * xx2 + o2; }
            const float g = d[k] * m * act_bwd<T>(u, act);
            o1v[k] = a1 * (g - mg - (x1[k] - mu1) * is1 * m1);
            o2v[k] = a2 * (g - mg - (xx2 - mu2) * is2 * m2);
        }
        st8<T>(dz1 + base + i, o1v);
        if (z2 != nullptr) st8<T>(dz2 + base + i, o2v);
    }
    for (int i = iv + threadIdx.x; i < i1; i += TPB) {
        const float x1 = ld_f32<T>(z1 + base + i);
        float u = a1 * x1 + o1;
        float x2 = 0.f;
        if (z2 != nullptr) { x2 = ld_f32<T>(z2 + base + i); u += a2 * x2 + o2; }
        const float g = ld_f32<T>(dy + base + i) * m * act_bwd<T>(u, act);
        st_f32<T>(dz1 + base + i, a1 * (g - mg - (x1 - mu1) * is1 * m1));
        if (z2 != nullptr) st_f32<T>(dz2 + base + i, a2 * (g - mg - (x2 - mu2) * is2 * m2));
    }
}

// ---- small planes (HW <= 4096, HW % 8 == 0): one WAVE per plane, no LDS, no barriers -----------------
template <typename T>
__global__ __launch_bounds__(TPB) void bn_stats_wave(const T* __restrict__ z, float* __restrict__ partial, int N,
                                                     int C, int HW, long planes) {
    const long plane = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const int lane = threadIdx.x & 63;
    const int n = (int)(plane / C), c = (int)(plane - (long)n * C);
    const T* p = z + plane * HW;
    float s = 0.f;
    for (int i = lane * V; i < HW; i += WAVE * V) {
        float x[V];
        ld8<T>(p + i, x);
#pragma unroll
        for (int k = 0; k < V; ++k) s += x[k];
    }
    const float mean = wave_sum(s) / (float)HW;
    float m2 = 0.f;
    for (int i = lane * V; i < HW; i += WAVE * V) {
        float x[V];
        ld8<T>(p + i, x);
#pragma unroll
        for (int k = 0; k < V; ++k) m2 += (x[k] - mean) * (x[k] - mean);
    }
    m2 = wave_sum(m2);
    if (lane == 0) {
        partial[((long)c * N + n) * 2] = mean;
        partial[((long)c * N + n) * 2 + 1] = m2;
    }
}

template <typename T>
__global__ __launch_bounds__(TPB) void bn_bwd_reduce_wave(const T* __restrict__ dy, const T* __restrict__ z1,
                                                          const T* __restrict__ z2, Branch b1, Branch b2,
                                                          const float* __restrict__ mask,
                                                          float* __restrict__ partial, int act, int N, int C, int HW,
                                                          long planes) {
    const long plane = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const int lane = threadIdx.x & 63;
    const int n = (int)(plane / C), c = (int)(plane - (long)n * C);
    const float is1 = b1.invstd[c], mu1 = b1.mean[c];
    const float a1 = b1.gamma[c] * is1, o1 = b1.beta[c] - mu1 * a1;
    float is2 = 0.f, mu2 = 0.f, a2 = 0.f, o2 = 0.f;
    if (z2 != nullptr) { is2 = b2.invstd[c]; mu2 = b2.mean[c]; a2 = b2.gamma[c] * is2; o2 = b2.beta[c] - mu2 * a2; }
    const float m = (mask != nullptr) ? mask[n] : 1.f;
    const long base = plane * HW;
    float sg = 0.f, s1 = 0.f, s2 = 0.f;
    for (int i = lane * V; i < HW; i += WAVE * V) {
        float x1[V], x2[V], d[V];
        ld8<T>(z1 + base + i, x1);
        if (z2 != nullptr) ld8<T>(z2 + base + i, x2);
        ld8<T>(dy + base + i, d);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float u = a1 * x1[k] + o1;
            float xx2 = 0.f;
            if (z2 != nullptr) { xx2 = x2[k]; u += a2 * xx2 + o2; }
            const float g = d[k] * m * act_bwd<T>(u, act);
            sg += g;
            s1 += g * (x1[k] - mu1) * is1;
            s2 += g * (xx2 - mu2) * is2;
        }
    }
    sg = wave_sum(sg); s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) {
        float* p = partial + ((long)c * N + n) * 3;
        p[0] = sg; p[1] = s1; p[2] = s2;
    }
}

// ---- small channels (N*HW <= 16384 elements): one WORKGROUP per CHANNEL covers all N planes, so the statistics
// are final when it is done -- no partial buffer, no finalize launch (stages 2 and 3 of the trunk: 20 of
// the 24 blocks of each encoder).

template <typename T>
__global__ __launch_bounds__(TPB) void bn_stats_channel(const T* __restrict__ z, int N, int C, int HW, float eps,
                                                        float momentum, float* __restrict__ mean_out,
                                                        float* __restrict__ var_out, float* __restrict__ invstd_out,
                                                        float* __restrict__ running_mean,
                                                        float* __restrict__ running_var,
                                                        float* __restrict__ count_out) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const int hv = HW / V, total = N * hv;              // vectors per plane / per channel
    if (count_out != nullptr && c == 0 && threadIdx.x == 0) *count_out = (float)N * (float)HW;
    // every load of the channel is issued before the first use: one memory latency for the whole pass, and
    // the values stay in registers for the second (centred) pass
    float x[CH_VECS][V];
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < CH_VECS; ++u) {
        const int j = threadIdx.x + u * TPB;
        if (j < total) {
            const int n = j / hv, i = j - n * hv;
            ld8<T>(z + ((long)n * C + c) * HW + i * V, x[u]);
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) x[u][k] = 0.f;
        }
    }
#pragma unroll
    for (int u = 0; u < CH_VECS; ++u)
#pragma unroll
        for (int k = 0; k < V; ++k) s += x[u][k];
    const float cnt = (float)N * (float)HW;
    const float mean = block_sum(s, red) / cnt;
    float m2 = 0.f;
#pragma unroll
    for (int u = 0; u < CH_VECS; ++u)
        if (threadIdx.x + u * TPB < total) {
#pragma unroll
            for (int k = 0; k < V; ++k) m2 += (x[u][k] - mean) * (x[u][k] - mean);
        }
    m2 = block_sum(m2, red);
    if (threadIdx.x == 0) {
        const float var = m2 / cnt;
        mean_out[c] = mean;
        var_out[c] = var;
        if (invstd_out != nullptr) invstd_out[c] = rsqrtf(var + eps);
        if (running_mean != nullptr) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (m2 / fmaxf(cnt - 1.f, 1.f));
        }
    }
}

template <typename T>
__global__ __launch_bounds__(TPB) void bn_bwd_reduce_channel(const T* __restrict__ dy, const T* __restrict__ dyb,
                                                             T* __restrict__ dym, const T* __restrict__ z1,
                                                             const T* __restrict__ z2, Branch b1, Branch b2,
                                                             const float* __restrict__ mask, float* __restrict__ sums,
                                                             int act, int N, int C, int HW) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const int hv = HW / V, total = N * hv;
    const float is1 = b1.invstd[c], mu1 = b1.mean[c];
    const float a1 = b1.gamma[c] * is1, o1 = b1.beta[c] - mu1 * a1;
    float is2 = 0.f, mu2 = 0.f, a2 = 0.f, o2 = 0.f;
    if (z2 != nullptr) { is2 = b2.invstd[c]; mu2 = b2.mean[c]; a2 = b2.gamma[c] * is2; o2 = b2.beta[c] - mu2 * a2; }
    float sg = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll 4
    for (int u = 0; u < CH_VECS; ++u) {
        const int j = threadIdx.x + u * TPB;
        if (j >= total) break;
        const int n = j / hv, i = j - n * hv;
        const float m = (mask != nullptr) ? mask[n] : 1.f;
        const long base = ((long)n * C + c) * HW + i * V;
        float x1[V], x2[V], d[V];
        ld8<T>(z1 + base, x1);
        if (z2 != nullptr) ld8<T>(z2 + base, x2);
        ld8<T>(dy + base, d);
        if (dyb != nullptr) {                                // second consumer's gradient: merge, and leave the sum for the apply launch
            float gb[V];
            ld8<T>(dyb + base, gb);
#pragma unroll
            for (int k = 0; k < V; ++k) d[k] = round_as<T>(d[k] + gb[k]);
            st8<T>(dym + base, d);
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float u2 = a1 * x1[k] + o1;
            float xx2 = 0.f;
            if (z2 != nullptr) { xx2 = x2[k]; u2 += a2 * xx2 + o2; }
            const float g = d[k] * m * act_bwd<T>(u2, act);
            sg += g;
            s1 += g * (x1[k] - mu1) * is1;
            s2 += g * (xx2 - mu2) * is2;
        }
    }
    sg = block_sum(sg, red); s1 = block_sum(s1, red); s2 = block_sum(s2, red);
    if (threadIdx.x == 0) { sums[c] = sg; sums[C + c] = s1; sums[2 * C + c] = s2; }
}

// ---- one launch per BN and direction for small channels ---------------------------------------------------------------
// The workgroup that owns a channel holds ALL of its values in registers, so the statistics pass and the apply pass
// are the same kernel: z is read once (not twice) and the BN costs one launch forward and one backward instead of two
// each (stages 2 and 3 of both encoders: 40 of the 48 blocks).  Same arithmetic as bn_stats_channel + bn_apply_flat
// and bn_bwd_reduce_channel + bn_bwd_apply_flat.
struct FwdPrm { const float *gamma1, *beta1, *gamma2, *beta2; float *rm1, *rv1, *rm2, *rv2, *mean1, *invstd1, *mean2, *invstd2; };

template <typename T, bool TWO, int NV>
__global__ __launch_bounds__(TPB) void bn_fwd_channel(const T* __restrict__ z1, const T* __restrict__ z2, FwdPrm p,
                                                      float eps, float momentum, const float* __restrict__ mask,
                                                      const T* __restrict__ r1, const T* __restrict__ r2,
                                                      float r2_scale, T* __restrict__ y, int act, int N, int C, int HW) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const int hv = HW / V, total = N * hv;
    const float cnt = (float)N * (float)HW;
    float x1[NV][V], x2[TWO ? NV : 1][V];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int j = threadIdx.x + u * TPB;
        if (j < total) {
            const int n = j / hv, i = j - n * hv;
            const long off = ((long)n * C + c) * HW + i * V;
            ld8<T>(z1 + off, x1[u]);
            if constexpr (TWO) ld8<T>(z2 + off, x2[u]);
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) { x1[u][k] = 0.f; if constexpr (TWO) x2[u][k] = 0.f; }
        }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u)
#pragma unroll
        for (int k = 0; k < V; ++k) { s1 += x1[u][k]; if constexpr (TWO) s2 += x2[u][k]; }
    const float mu1 = block_sum(s1, red) / cnt;
    float mu2 = 0.f;
    if constexpr (TWO) mu2 = block_sum(s2, red) / cnt;
    float q1 = 0.f, q2 = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u)
        if (threadIdx.x + u * TPB < total) {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                q1 += (x1[u][k] - mu1) * (x1[u][k] - mu1);
                if constexpr (TWO) q2 += (x2[u][k] - mu2) * (x2[u][k] - mu2);
            }
        }
    q1 = block_sum(q1, red);
    if constexpr (TWO) q2 = block_sum(q2, red);
    const float is1 = rsqrtf(q1 / cnt + eps), is2 = TWO ? rsqrtf(q2 / cnt + eps) : 0.f;
    if (threadIdx.x == 0) {
        p.mean1[c] = mu1; p.invstd1[c] = is1;
        if (p.rm1 != nullptr) {
            p.rm1[c] = (1.f - momentum) * p.rm1[c] + momentum * mu1;
            p.rv1[c] = (1.f - momentum) * p.rv1[c] + momentum * (q1 / fmaxf(cnt - 1.f, 1.f));
        }
        if constexpr (TWO) {
            p.mean2[c] = mu2; p.invstd2[c] = is2;
            if (p.rm2 != nullptr) {
                p.rm2[c] = (1.f - momentum) * p.rm2[c] + momentum * mu2;
                p.rv2[c] = (1.f - momentum) * p.rv2[c] + momentum * (q2 / fmaxf(cnt - 1.f, 1.f));
            }
        }
    }
    const float a1 = p.gamma1[c] * is1, o1 = p.beta1[c] - mu1 * a1;
    float a2 = 0.f, o2 = 0.f;
    if constexpr (TWO) { a2 = p.gamma2[c] * is2; o2 = p.beta2[c] - mu2 * a2; }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int j = threadIdx.x + u * TPB;
        if (j >= total) break;
        const int n = j / hv, i = j - n * hv;
        const long off = ((long)n * C + c) * HW + i * V;
        const float m = (mask != nullptr) ? mask[n] : 1.f;
        float e1[V], e2[V], o[V];
        if (r1 != nullptr) ld8<T>(r1 + off, e1);
        if (r2 != nullptr) ld8<T>(r2 + off, e2);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float w = a1 * x1[u][k] + o1;
            if constexpr (TWO) w += a2 * x2[u][k] + o2;
            float v = act_fwd<T>(w, act) * m;
            if (r1 != nullptr) v += e1[k];
            if (r2 != nullptr) v += r2_scale * e2[k];
            o[k] = v;
        }
        st8<T>(y + off, o);
    }
}

// The same forward with the statistics taken from partial sums the PRODUCING GEMM left in its epilogue
// (ppea_pwconv_stats_bf16: partial [C][P][2] = (sum, sum of squares) of disjoint pixel sets of the stored values): no
// reduction over the activation and no workgroup barrier -- every wave finalises the channel itself (fp64, the arithmetic and
// order of bn_finalize_sums) while its loads of the channel are in flight, then applies.  The FFN's BatchNorm + GELU over the
// 4C-wide hidden tensor (rka.py:264-289) is the user: 47 MB per launch at stage 2.
template <typename T, int NV>
__global__ __launch_bounds__(TPB) void bn_fwd_channel_sums(const T* __restrict__ z1, const float* __restrict__ partial, int P,
                                                           FwdPrm p, float eps, float momentum,
                                                           const float* __restrict__ mask, const T* __restrict__ r1,
                                                           const T* __restrict__ r2, float r2_scale, T* __restrict__ y,
                                                           int act, int N, int C, int HW) {
    const int c = blockIdx.x, lane = threadIdx.x & 63;
    const int hv = HW / V, total = N * hv;
    const double cnt = (double)N * (double)HW;
    float x1[NV][V];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int j = threadIdx.x + u * TPB;
        if (j < total) {
            const int n = j / hv, i = j - n * hv;
            ld8<T>(z1 + ((long)n * C + c) * HW + i * V, x1[u]);
        }
    }
    const float2* sp = reinterpret_cast<const float2*>(partial) + (long)c * P;
    double ds = 0.0, dq = 0.0;
#pragma unroll 8
    for (int i = lane; i < P; i += 64) {
        const float2 v = sp[i];
        ds += (double)v.x; dq += (double)v.y;
    }
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) { ds += __shfl_xor(ds, k, WAVE); dq += __shfl_xor(dq, k, WAVE); }
    const double mean = ds / cnt;
    double var = dq / cnt - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float mu1 = (float)mean, is1 = rsqrtf((float)var + eps);
    if (threadIdx.x == 0) {
        p.mean1[c] = mu1; p.invstd1[c] = is1;
        if (p.rm1 != nullptr) {
            const float unbiased = (float)(var * cnt / fmax(cnt - 1.0, 1.0));
            p.rm1[c] = (1.f - momentum) * p.rm1[c] + momentum * mu1;
            p.rv1[c] = (1.f - momentum) * p.rv1[c] + momentum * unbiased;
        }
    }
    const float a1 = p.gamma1[c] * is1, o1 = p.beta1[c] - mu1 * a1;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int j = threadIdx.x + u * TPB;
        if (j >= total) break;
        const int n = j / hv, i = j - n * hv;
        const long off = ((long)n * C + c) * HW + i * V;
        const float m = (mask != nullptr) ? mask[n] : 1.f;
        float e1[V], e2[V], o[V];
        if (r1 != nullptr) ld8<T>(r1 + off, e1);
        if (r2 != nullptr) ld8<T>(r2 + off, e2);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float v = act_fwd<T>(a1 * x1[u][k] + o1, act) * m;
            if (r1 != nullptr) v += e1[k];
            if (r2 != nullptr) v += r2_scale * e2[k];
            o[k] = v;
        }
        st8<T>(y + off, o);
    }
}

template <typename T, bool TWO, int NV>
__global__ __launch_bounds__(TPB) void bn_bwd_channel(const T* __restrict__ dy, const T* __restrict__ dyb, const T* __restrict__ z1,
                                                      const T* __restrict__ z2, Branch b1, Branch b2,
                                                      const float* __restrict__ mask, float inv_count,
                                                      const T* __restrict__ acc, T* __restrict__ dz1,
                                                      T* __restrict__ dz2, float* __restrict__ sums, int act, int N,
                                                      int C, int HW) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const int hv = HW / V, total = N * hv;
    const float is1 = b1.invstd[c], mu1 = b1.mean[c];
    const float a1 = b1.gamma[c] * is1, o1 = b1.beta[c] - mu1 * a1;
    float is2 = 0.f, mu2 = 0.f, a2 = 0.f, o2 = 0.f;
    if constexpr (TWO) { is2 = b2.invstd[c]; mu2 = b2.mean[c]; a2 = b2.gamma[c] * is2; o2 = b2.beta[c] - mu2 * a2; }
    float x1[NV][V], x2[TWO ? NV : 1][V], gq[NV][V];
    float sg = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int j = threadIdx.x + u * TPB;
        if (j < total) {
            const int n = j / hv, i = j - n * hv;
            const long off = ((long)n * C + c) * HW + i * V;
            ld8<T>(z1 + off, x1[u]);
            if constexpr (TWO) ld8<T>(z2 + off, x2[u]);
            ld8<T>(dy + off, gq[u]);
            if (dyb != nullptr) {                            // the output's second consumer (see ppea_bn_bwd_channel_dup_*)
                float gb[V];
                ld8<T>(dyb + off, gb);
#pragma unroll
                for (int k = 0; k < V; ++k) gq[u][k] = round_as<T>(gq[u][k] + gb[k]);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int j = threadIdx.x + u * TPB;
        if (j < total) {
            const float m = (mask != nullptr) ? mask[j / hv] : 1.f;
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float w = a1 * x1[u][k] + o1;
                if constexpr (TWO) w += a2 * x2[u][k] + o2;
                const float g = gq[u][k] * m * act_bwd<T>(w, act);
                gq[u][k] = g;
                sg += g;
                s1 += g * (x1[u][k] - mu1) * is1;
                if constexpr (TWO) s2 += g * (x2[u][k] - mu2) * is2;
            }
        }
    }
    sg = block_sum(sg, red); s1 = block_sum(s1, red);
    if constexpr (TWO) s2 = block_sum(s2, red);
    if (threadIdx.x == 0) { sums[c] = sg; sums[C + c] = s1; sums[2 * C + c] = s2; }
    const float mg = sg * inv_count, m1 = s1 * inv_count, m2 = s2 * inv_count;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int j = threadIdx.x + u * TPB;
        if (j >= total) break;
        const int n = j / hv, i = j - n * hv;
        const long off = ((long)n * C + c) * HW + i * V;
        float e1[V], e2[V], ea[V];
        if (acc != nullptr) ld8<T>(acc + off, ea);         // gradient reaching z1 through its other consumer
#pragma unroll
        for (int k = 0; k < V; ++k) {
            e1[k] = a1 * (gq[u][k] - mg - (x1[u][k] - mu1) * is1 * m1);
            if constexpr (TWO) e2[k] = a2 * (gq[u][k] - mg - (x2[u][k] - mu2) * is2 * m2);
        }
        if (acc != nullptr) {
#pragma unroll
            for (int k = 0; k < V; ++k) e1[k] = round_as<T>(e1[k]) + ea[k];   // = the two-kernel result (dz stored, then added)
        }
        st8<T>(dz1 + off, e1);
        if constexpr (TWO) st8<T>(dz2 + off, e2);
    }
}

// ---- end of one block + start of the next in ONE launch per direction ---------------------------------------------------
// A block ends with  x' = x + mask * BN_A(z) + s * adapter  (rka.py:283-289, 315-326) and the next block starts with
// out' = BN_B(x') (its prelkb_bn / preffn_bn).  Both are per-channel passes over the same tensor, and at stages 2 / 3 the
// step's wall time is the number of dependent launches: the workgroup that owns the channel keeps x' in registers, takes
// BN_B's statistics of the STORED (rounded) values and writes out' as well.  Backward likewise:
//   dx' = BN_B-backward(d out') + (gradient of x' from its other use, the next block's residual);  dz = BN_A-backward(mask dx').
// Arithmetic, rounding points and running-statistics updates are those of the two separate launches (bit-identical results).
struct NextPrm { const float *gammaA, *betaA, *gammaB, *betaB; float *rmA, *rvA, *rmB, *rvB, *meanA, *invstdA, *meanB, *invstdB; };

template <typename T, int NV>
__global__ __launch_bounds__(TPB) void bn_fwd_channel_next(const T* __restrict__ z, NextPrm p, float eps, float momentum,
                                                           const float* __restrict__ mask, const T* __restrict__ r1,
                                                           const T* __restrict__ r2, float r2_scale, T* __restrict__ y,
                                                           T* __restrict__ y2, int N, int C, int HW) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const int hv = HW / V, total = N * hv;
    const float cnt = (float)N * (float)HW;
    float x[NV][V];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int j = threadIdx.x + u * TPB;
        if (j < total) {
            const int n = j / hv, i = j - n * hv;
            ld8<T>(z + ((long)n * C + c) * HW + i * V, x[u]);
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) x[u][k] = 0.f;
        }
    }
    // two-pass statistics of the register-resident values (valid elements only); returns (mean, sum of squared deviations)
    auto stats = [&](float& mu, float& q) {
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < NV; ++u)
#pragma unroll
            for (int k = 0; k < V; ++k) s += x[u][k];
        mu = block_sum(s, red) / cnt;
        float qq = 0.f;
#pragma unroll
        for (int u = 0; u < NV; ++u)
            if (threadIdx.x + u * TPB < total) {
#pragma unroll
                for (int k = 0; k < V; ++k) qq += (x[u][k] - mu) * (x[u][k] - mu);
            }
        q = block_sum(qq, red);
    };
    float muA, qA;
    stats(muA, qA);
    const float isA = rsqrtf(qA / cnt + eps);
    if (threadIdx.x == 0) {
        p.meanA[c] = muA; p.invstdA[c] = isA;
        if (p.rmA != nullptr) {
            p.rmA[c] = (1.f - momentum) * p.rmA[c] + momentum * muA;
            p.rvA[c] = (1.f - momentum) * p.rvA[c] + momentum * (qA / fmaxf(cnt - 1.f, 1.f));
        }
    }
    const float aA = p.gammaA[c] * isA, oA = p.betaA[c] - muA * aA;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int j = threadIdx.x + u * TPB;
        if (j < total) {
            const int n = j / hv, i = j - n * hv;
            const long off = ((long)n * C + c) * HW + i * V;
            const float m = (mask != nullptr) ? mask[n] : 1.f;
            float e1[V], e2[V];
            if (r1 != nullptr) ld8<T>(r1 + off, e1);
            if (r2 != nullptr) ld8<T>(r2 + off, e2);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float v = (aA * x[u][k] + oA) * m;
                if (r1 != nullptr) v += e1[k];
                if (r2 != nullptr) v += r2_scale * e2[k];
                x[u][k] = round_as<T>(v);                  // what the next block reads back
            }
            st8<T>(y + off, x[u]);
        }
    }
    float muB, qB;
    stats(muB, qB);
    const float isB = rsqrtf(qB / cnt + eps);
    if (threadIdx.x == 0) {
        p.meanB[c] = muB; p.invstdB[c] = isB;
        if (p.rmB != nullptr) {
            p.rmB[c] = (1.f - momentum) * p.rmB[c] + momentum * muB;
            p.rvB[c] = (1.f - momentum) * p.rvB[c] + momentum * (qB / fmaxf(cnt - 1.f, 1.f));
        }
    }
    const float aB = p.gammaB[c] * isB, oB = p.betaB[c] - muB * aB;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int j = threadIdx.x + u * TPB;
        if (j < total) {
            const int n = j / hv, i = j - n * hv;
            float o[V];
#pragma unroll
            for (int k = 0; k < V; ++k) o[k] = aB * x[u][k] + oB;
            st8<T>(y2 + ((long)n * C + c) * HW + i * V, o);
        }
    }
}

// sums [4][C] = d betaA | d gammaA | d betaB | d gammaB
template <typename T, int NV>
__global__ __launch_bounds__(TPB) void bn_bwd_channel_next(const T* __restrict__ dy2, const T* __restrict__ dy2b, const T* __restrict__ dskip,
                                                           const T* __restrict__ z, const T* __restrict__ y, Branch A,
                                                           Branch B, const float* __restrict__ mask, float inv_count,
                                                           T* __restrict__ dz, T* __restrict__ dy, float* __restrict__ sums,
                                                           int N, int C, int HW) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const int hv = HW / V, total = N * hv;
    const float isA = A.invstd[c], muA = A.mean[c], aA = A.gamma[c] * isA;
    const float isB = B.invstd[c], muB = B.mean[c], aB = B.gamma[c] * isB;
    float g[NV][V], xd[NV][V];                    // gradient in flight, input - mean of the current BN
    float sg = 0.f, s1 = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int j = threadIdx.x + u * TPB;
        if (j < total) {
            const int n = j / hv, i = j - n * hv;
            const long off = ((long)n * C + c) * HW + i * V;
            ld8<T>(dy2 + off, g[u]);
            if (dy2b != nullptr) {                           // y2's second consumer (see ppea_bn_bwd_channel_next_dup_*)
                float gb[V];
                ld8<T>(dy2b + off, gb);
#pragma unroll
                for (int k = 0; k < V; ++k) g[u][k] = round_as<T>(g[u][k] + gb[k]);
            }
            ld8<T>(y + off, xd[u]);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                xd[u][k] = xd[u][k] - muB;
                sg += g[u][k];
                s1 += g[u][k] * xd[u][k] * isB;              // operation order of bn_bwd_channel (bit-identical sums)
            }
        }
    }
    sg = block_sum(sg, red); s1 = block_sum(s1, red);
    if (threadIdx.x == 0) { sums[2 * C + c] = sg; sums[3 * C + c] = s1; }
    float mg = sg * inv_count, m1 = s1 * inv_count;
    sg = 0.f; s1 = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int j = threadIdx.x + u * TPB;
        if (j < total) {
            const int n = j / hv, i = j - n * hv;
            const long off = ((long)n * C + c) * HW + i * V;
            const float m = (mask != nullptr) ? mask[n] : 1.f;
            float ea[V], zz[V];
            if (dskip != nullptr) ld8<T>(dskip + off, ea);
            ld8<T>(z + off, zz);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float e = round_as<T>(aB * (g[u][k] - mg - xd[u][k] * isB * m1));   // BN_B backward, as its own launch stores it
                if (dskip != nullptr) e = round_as<T>(e + ea[k]);               // + the residual use's gradient
                g[u][k] = e;
            }
            st8<T>(dy + off, g[u]);                          // d x': to the block's input (residual) and, scaled, to the adapter
#pragma unroll
            for (int k = 0; k < V; ++k) {
                g[u][k] *= m;
                xd[u][k] = zz[k] - muA;
                sg += g[u][k];
                s1 += g[u][k] * xd[u][k] * isA;
            }
        }
    }
    sg = block_sum(sg, red); s1 = block_sum(s1, red);
    if (threadIdx.x == 0) { sums[c] = sg; sums[C + c] = s1; }
    mg = sg * inv_count; m1 = s1 * inv_count;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int j = threadIdx.x + u * TPB;
        if (j < total) {
            const int n = j / hv, i = j - n * hv;
            float o[V];
#pragma unroll
            for (int k = 0; k < V; ++k) o[k] = aA * (g[u][k] - mg - xd[u][k] * isA * m1);
            st8<T>(dz + ((long)n * C + c) * HW + i * V, o);
        }
    }
}


// ---- flat element-wise passes (HW % 8 == 0): 8 elements per thread, channel looked up per thread -------
template <typename T>
__global__ __launch_bounds__(TPB) void bn_apply_flat(const T* __restrict__ z1, const T* __restrict__ z2, Branch b1,
                                                     Branch b2, const float* __restrict__ mask,
                                                     const T* __restrict__ r1, const T* __restrict__ r2,
                                                     float r2_scale, T* __restrict__ y, int act, int C, int HW,
                                                     long total8) {
    const long t = (long)blockIdx.x * TPB + threadIdx.x;
    if (t >= total8) return;
    const long i = t * V;
    // two 64-bit divisions per thread would cost more than the 8 elements of work: 32-bit whenever it fits
    long plane;
    int n, c;
    if (total8 < (1L << 28)) {
        const unsigned p32 = (unsigned)i / (unsigned)HW, n32 = p32 / (unsigned)C;
        plane = p32; n = (int)n32; c = (int)(p32 - n32 * (unsigned)C);
    } else {
        plane = i / HW;
        n = (int)(plane / C); c = (int)(plane - (long)n * C);
    }
    const float a1 = b1.gamma[c] * b1.invstd[c], o1 = b1.beta[c] - b1.mean[c] * a1;
    float a2 = 0.f, o2 = 0.f;
    if (z2 != nullptr) { a2 = b2.gamma[c] * b2.invstd[c]; o2 = b2.beta[c] - b2.mean[c] * a2; }
    const float m = (mask != nullptr) ? mask[n] : 1.f;
    float x1[V], x2[V], q1[V], q2[V], o[V];
    ld8<T>(z1 + i, x1);
    if (z2 != nullptr) ld8<T>(z2 + i, x2);
    if (r1 != nullptr) ld8<T>(r1 + i, q1);
    if (r2 != nullptr) ld8<T>(r2 + i, q2);
#pragma unroll
    for (int k = 0; k < V; ++k) {
        float u = a1 * x1[k] + o1;
        if (z2 != nullptr) u += a2 * x2[k] + o2;
        float v = act_fwd<T>(u, act) * m;
        if (r1 != nullptr) v += q1[k];
        if (r2 != nullptr) v += r2_scale * q2[k];
        o[k] = v;
    }
    st8<T>(y + i, o);
}

template <typename T>
__global__ __launch_bounds__(TPB) void bn_bwd_apply_flat(const T* __restrict__ dy, const T* __restrict__ z1,
                                                         const T* __restrict__ z2, Branch b1, Branch b2,
                                                         const float* __restrict__ mask,
                                                         const float* __restrict__ sums, float inv_count,
                                                         const T* __restrict__ acc, T* __restrict__ dz1,
                                                         T* __restrict__ dz2, float* __restrict__ dgb, float gscale,
                                                         int act, int C, int HW, long total8) {
    const long t = (long)blockIdx.x * TPB + threadIdx.x;
    if (t >= total8) return;
    const long i = t * V;
    // two 64-bit divisions per thread would cost more than the 8 elements of work: 32-bit whenever it fits
    long plane;
    int n, c;
    if (total8 < (1L << 28)) {
        const unsigned p32 = (unsigned)i / (unsigned)HW, n32 = p32 / (unsigned)C;
        plane = p32; n = (int)n32; c = (int)(p32 - n32 * (unsigned)C);
    } else {
        plane = i / HW;
        n = (int)(plane / C); c = (int)(plane - (long)n * C);
    }
    const float is1 = b1.invstd[c], mu1 = b1.mean[c];
    const float a1 = b1.gamma[c] * is1, o1 = b1.beta[c] - mu1 * a1;
    float is2 = 0.f, mu2 = 0.f, a2 = 0.f, o2 = 0.f;
    if (z2 != nullptr) { is2 = b2.invstd[c]; mu2 = b2.mean[c]; a2 = b2.gamma[c] * is2; o2 = b2.beta[c] - mu2 * a2; }
    const float m = (mask != nullptr) ? mask[n] : 1.f;
    const float mg = sums[c] * inv_count, m1 = sums[C + c] * inv_count, m2 = sums[2 * C + c] * inv_count;
    if (dgb != nullptr && n == 0 && i == plane * HW) {      // the thread at the head of plane (0, c): one writer per channel
        dgb[c] = sums[c] * gscale; dgb[C + c] = sums[C + c] * gscale; dgb[2 * C + c] = sums[2 * C + c] * gscale;
    }
    float x1[V], x2[V], d[V], o1v[V], o2v[V];
    ld8<T>(z1 + i, x1);
    if (z2 != nullptr) ld8<T>(z2 + i, x2);
    ld8<T>(dy + i, d);
#pragma unroll
    for (int k = 0; k < V; ++k) {
        float u = a1 * x1[k] + o1;
        float xx2 = 0.f;
        if (z2 != nullptr) { xx2 = x2[k]; u += a2 * xx2 + o2; }
        const float g = d[k] * m * act_bwd<T>(u, act);
        o1v[k] = a1 * (g - mg - (x1[k] - mu1) * is1 * m1);
        o2v[k] = a2 * (g - mg - (xx2 - mu2) * is2 * m2);
    }
    if (acc != nullptr) {                                  // gradient reaching z1 through its other consumer (as bn_bwd_channel)
        float ea[V];
        ld8<T>(acc + i, ea);
#pragma unroll
        for (int k = 0; k < V; ++k) o1v[k] = round_as<T>(o1v[k]) + ea[k];
    }
    st8<T>(dz1 + i, o1v);
    if (z2 != nullptr) st8<T>(dz2 + i, o2v);
}

constexpr int SMALL_PLANE = 4096;

inline int plane_chunks(long planes, int HW) {
    // enough blocks to fill the chip (>= ~2048) without making them tiny
    int chunks = 1;
    while (planes * chunks < 2048 && HW / (chunks * 2) >= 1024) chunks *= 2;
    return chunks;
}

template <typename T>
int stats_impl(const void* z, float* partial, int N, int C, int HW, void* stream) {
    if (N <= 0 || C <= 0 || HW <= 0) return PPEA_ERR_UNSUPPORTED;
    const long planes = (long)N * C;
    if (HW % V == 0 && HW <= SMALL_PLANE)
        hipLaunchKernelGGL(bn_stats_wave<T>, dim3((unsigned)((planes + 3) / 4)), dim3(TPB), 0, (hipStream_t)stream,
                           (const T*)z, partial, N, C, HW, planes);
    else
        hipLaunchKernelGGL(bn_stats<T>, dim3((unsigned)planes), dim3(TPB), 0, (hipStream_t)stream, (const T*)z,
                           partial, N, C, HW);
    return launch_status();
}

template <typename T>
int apply_impl(const void* z1, const void* z2, const float* const* st, const float* mask, const void* r1,
               const void* r2, float r2_scale, void* y, int act, int N, int C, int HW, void* stream) {
    if (N <= 0 || C <= 0 || HW <= 0 || act < 0 || act > 2) return PPEA_ERR_UNSUPPORTED;
    const Branch b1{st[0], st[1], st[2], st[3]}, b2{st[4], st[5], st[6], st[7]};
    if (HW % V == 0) {
        const long total8 = (long)N * C * HW / V;
        hipLaunchKernelGGL(bn_apply_flat<T>, dim3((unsigned)((total8 + TPB - 1) / TPB)), dim3(TPB), 0,
                           (hipStream_t)stream, (const T*)z1, (const T*)z2, b1, b2, mask, (const T*)r1,
                           (const T*)r2, r2_scale, (T*)y, act, C, HW, total8);
        return launch_status();
    }
    const int chunks = plane_chunks((long)N * C, HW);
    hipLaunchKernelGGL(bn_apply<T>, dim3((unsigned)((long)N * C * chunks)), dim3(TPB), 0, (hipStream_t)stream,
                       (const T*)z1, (const T*)z2, b1, b2, mask, (const T*)r1, (const T*)r2, r2_scale, (T*)y, act, C,
                       HW, chunks);
    return launch_status();
}

template <typename T>
int bwd_reduce_impl(const void* dy, const void* z1, const void* z2, const float* const* st, const float* mask,
                    float* partial, int act, int N, int C, int HW, void* stream) {
    if (N <= 0 || C <= 0 || HW <= 0 || act < 0 || act > 2) return PPEA_ERR_UNSUPPORTED;
    const Branch b1{st[0], st[1], st[2], st[3]}, b2{st[4], st[5], st[6], st[7]};
    const long planes = (long)N * C;
    if (HW % V == 0 && HW <= SMALL_PLANE)
        hipLaunchKernelGGL(bn_bwd_reduce_wave<T>, dim3((unsigned)((planes + 3) / 4)), dim3(TPB), 0,
                           (hipStream_t)stream, (const T*)dy, (const T*)z1, (const T*)z2, b1, b2, mask, partial, act,
                           N, C, HW, planes);
    else
        hipLaunchKernelGGL(bn_bwd_reduce<T>, dim3((unsigned)planes), dim3(TPB), 0, (hipStream_t)stream,
                           (const T*)dy, (const T*)z1, (const T*)z2, b1, b2, mask, partial, act, N, C, HW);
    return launch_status();
}

template <typename T>
int bwd_apply_impl(const void* dy, const void* z1, const void* z2, const float* const* st, const float* mask,
                   const float* sums, float inv_count, void* dz1, void* dz2, int act, int N, int C, int HW,
                   void* stream, const void* acc = nullptr, float* dgb = nullptr, float gscale = 1.f) {
    if (N <= 0 || C <= 0 || HW <= 0 || act < 0 || act > 2) return PPEA_ERR_UNSUPPORTED;
    const Branch b1{st[0], st[1], st[2], st[3]}, b2{st[4], st[5], st[6], st[7]};
    if (HW % V == 0) {
        const long total8 = (long)N * C * HW / V;
        hipLaunchKernelGGL(bn_bwd_apply_flat<T>, dim3((unsigned)((total8 + TPB - 1) / TPB)), dim3(TPB), 0,
                           (hipStream_t)stream, (const T*)dy, (const T*)z1, (const T*)z2, b1, b2, mask, sums,
                           inv_count, (const T*)acc, (T*)dz1, (T*)dz2, dgb, gscale, act, C, HW, total8);
        return launch_status();
    }
    if (acc != nullptr || dgb != nullptr) return PPEA_ERR_UNSUPPORTED;
    const int chunks = plane_chunks((long)N * C, HW);
    hipLaunchKernelGGL(bn_bwd_apply<T>, dim3((unsigned)((long)N * C * chunks)), dim3(TPB), 0, (hipStream_t)stream,
                       (const T*)dy, (const T*)z1, (const T*)z2, b1, b2, mask, sums, inv_count, (T*)dz1, (T*)dz2, act,
                       C, HW, chunks);
    return launch_status();
}

}  // namespace

template <typename T>
int stats_final_impl(const void* z, int N, int C, int HW, float eps, float momentum, float* mean, float* var,
                     float* invstd, float* running_mean, float* running_var, void* stream) {
    if (N <= 0 || C < 64 || HW <= 0 || (HW % V) != 0 || (long)N * HW > CHANNEL_ELEMS) return PPEA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(bn_stats_channel<T>, dim3((unsigned)C), dim3(TPB), 0, (hipStream_t)stream,
                       (const T*)z, N, C, HW, eps, momentum, mean, var, invstd, running_mean, running_var,
                       (float*)nullptr);
    return launch_status();
}
// SyncBN wire format packed[2C+1] = mean | biased var | count in one launch (small channels)
template <typename T>
int stats_packed_impl(const void* z, int N, int C, int HW, float* packed, void* stream) {
    if (N <= 0 || C < 64 || HW <= 0 || (HW % V) != 0 || (long)N * HW > CHANNEL_ELEMS) return PPEA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(bn_stats_channel<T>, dim3((unsigned)C), dim3(TPB), 0, (hipStream_t)stream,
                       (const T*)z, N, C, HW, 0.f, 0.f, packed, packed + C, (float*)nullptr, (float*)nullptr,
                       (float*)nullptr, packed + 2 * C);
    return launch_status();
}
template <typename T>
int bwd_reduce_final_impl(const void* dy, const void* dyb, void* dym, const void* z1, const void* z2, const float* const* st,
                          const float* mask, float* sums, int act, int N, int C, int HW, void* stream) {
    if ((dyb == nullptr) != (dym == nullptr)) return PPEA_ERR_ARG;
    if (N <= 0 || C < 64 || HW <= 0 || (HW % V) != 0 || (long)N * HW > CHANNEL_ELEMS) return PPEA_ERR_UNSUPPORTED;
    Branch b1{st[0], st[1], st[2], st[3]}, b2{st[4], st[5], st[6], st[7]};
    hipLaunchKernelGGL(bn_bwd_reduce_channel<T>, dim3((unsigned)C), dim3(TPB), 0, (hipStream_t)stream,
                       (const T*)dy, (const T*)dyb, (T*)dym, (const T*)z1, (const T*)z2, b1, b2, mask, sums, act, N, C, HW);
    return launch_status();
}

template <typename T>
int fwd_channel_impl(const void* z1, const void* z2, const float* const* prm, float* const* outp, float eps, float momentum,
                     const float* mask, const void* r1, const void* r2, float r2_scale, void* y, int act, int N, int C, int HW,
                     void* stream) {
    if (N <= 0 || C < 64 || HW <= 0 || (HW % V) != 0 || (long)N * HW > CHANNEL_ELEMS || act < 0 || act > 2)
        return PPEA_ERR_UNSUPPORTED;
    FwdPrm p{prm[0], prm[1], prm[2], prm[3], outp[0], outp[1], outp[2], outp[3], outp[4], outp[5], outp[6], outp[7]};
#define PPEA_L(NV_)                                                                                                   \
    if (z2 != nullptr)                                                                                                \
        hipLaunchKernelGGL((bn_fwd_channel<T, true, NV_>), dim3((unsigned)C), dim3(TPB), 0, (hipStream_t)stream, \
                           (const T*)z1, (const T*)z2, p, eps, momentum, mask, (const T*)r1, (const T*)r2, r2_scale, (T*)y, act, N, C, HW); \
    else                                                                                                              \
        hipLaunchKernelGGL((bn_fwd_channel<T, false, NV_>), dim3((unsigned)C), dim3(TPB), 0, (hipStream_t)stream, \
                           (const T*)z1, (const T*)nullptr, p, eps, momentum, mask, (const T*)r1, (const T*)r2, r2_scale, (T*)y, act, N, C, HW)
    PPEA_BN_NV(bn_nv(N, HW), PPEA_L);
#undef PPEA_L
    return launch_status();
}
template <typename T>
int fwd_channel_sums_impl(const void* z1, const float* partial, int P, const float* const* prm, float* const* outp, float eps,
                          float momentum, const float* mask, const void* r1, const void* r2, float r2_scale, void* y, int act,
                          int N, int C, int HW, void* stream) {
    if (N <= 0 || C < 64 || HW <= 0 || (HW % V) != 0 || (long)N * HW > CHANNEL_ELEMS || act < 0 || act > 2)
        return PPEA_ERR_UNSUPPORTED;
    if (partial == nullptr || P <= 0) return PPEA_ERR_ARG;
    FwdPrm p{prm[0], prm[1], nullptr, nullptr, outp[0], outp[1], nullptr, nullptr, outp[2], outp[3], nullptr, nullptr};
#define PPEA_L(NV_)                                                                                                   \
    hipLaunchKernelGGL((bn_fwd_channel_sums<T, NV_>), dim3((unsigned)C), dim3(TPB), 0, (hipStream_t)stream, (const T*)z1, partial, \
                       P, p, eps, momentum, mask, (const T*)r1, (const T*)r2, r2_scale, (T*)y, act, N, C, HW)
    PPEA_BN_NV(bn_nv(N, HW), PPEA_L);
#undef PPEA_L
    return launch_status();
}
template <typename T>
int bwd_channel_impl(const void* dy, const void* dyb, const void* z1, const void* z2, const float* const* st, const float* mask,
                     float inv_count, const void* acc, void* dz1, void* dz2, float* sums, int act, int N, int C, int HW,
                     void* stream) {
    if (N <= 0 || C < 64 || HW <= 0 || (HW % V) != 0 || (long)N * HW > CHANNEL_ELEMS || act < 0 || act > 2)
        return PPEA_ERR_UNSUPPORTED;
    Branch b1{st[0], st[1], st[2], st[3]}, b2{st[4], st[5], st[6], st[7]};
#define PPEA_L(NV_)                                                                                                   \
    if (z2 != nullptr)                                                                                                \
        hipLaunchKernelGGL((bn_bwd_channel<T, true, NV_>), dim3((unsigned)C), dim3(TPB), 0, (hipStream_t)stream, \
                           (const T*)dy, (const T*)dyb, (const T*)z1, (const T*)z2, b1, b2, mask, inv_count, (const T*)acc, (T*)dz1, (T*)dz2, sums, act, N, C, HW); \
    else                                                                                                              \
        hipLaunchKernelGGL((bn_bwd_channel<T, false, NV_>), dim3((unsigned)C), dim3(TPB), 0, (hipStream_t)stream, \
                           (const T*)dy, (const T*)dyb, (const T*)z1, (const T*)nullptr, b1, b2, mask, inv_count, (const T*)acc, (T*)dz1, (T*)nullptr, sums, act, N, C, HW)
    PPEA_BN_NV(bn_nv(N, HW), PPEA_L);
#undef PPEA_L
    return launch_status();
}

template <typename T>
int fwd_channel_next_impl(const void* z, const float* const* prm, float* const* outp, float eps, float momentum,
                          const float* mask, const void* r1, const void* r2, float r2_scale, void* y, void* y2, int N, int C,
                          int HW, void* stream) {
    if (N <= 0 || C < 64 || HW <= 0 || (HW % V) != 0 || (long)N * HW > CHANNEL_ELEMS) return PPEA_ERR_UNSUPPORTED;
    NextPrm p{prm[0], prm[1], prm[2], prm[3], outp[0], outp[1], outp[2], outp[3], outp[4], outp[5], outp[6], outp[7]};
#define PPEA_L(NV_)                                                                                                   \
    hipLaunchKernelGGL((bn_fwd_channel_next<T, NV_>), dim3((unsigned)C), dim3(TPB), 0, (hipStream_t)stream, (const T*)z, p, eps, \
                       momentum, mask, (const T*)r1, (const T*)r2, r2_scale, (T*)y, (T*)y2, N, C, HW)
    PPEA_BN_NV(bn_nv(N, HW), PPEA_L);
#undef PPEA_L
    return launch_status();
}
template <typename T>
int bwd_channel_next_impl(const void* dy2, const void* dy2b, const void* dskip, const void* z, const void* y, const float* const* st,
                          const float* mask, float inv_count, void* dz, void* dy, float* sums, int N, int C, int HW,
                          void* stream) {
    if (N <= 0 || C < 64 || HW <= 0 || (HW % V) != 0 || (long)N * HW > CHANNEL_ELEMS) return PPEA_ERR_UNSUPPORTED;
    Branch A{st[0], st[1], st[2], st[3]}, B{st[4], st[5], st[6], st[7]};
#define PPEA_L(NV_)                                                                                                   \
    hipLaunchKernelGGL((bn_bwd_channel_next<T, NV_>), dim3((unsigned)C), dim3(TPB), 0, (hipStream_t)stream, (const T*)dy2, \
                       (const T*)dy2b, (const T*)dskip, (const T*)z, (const T*)y, A, B, mask, inv_count, (T*)dz, (T*)dy, sums, N, C, HW)
    PPEA_BN_NV(bn_nv(N, HW), PPEA_L);
#undef PPEA_L
    return launch_status();
}

extern "C" {

// End of a block + the next block's first BatchNorm in one launch (see bn_fwd_channel_next): y = mask * BN_A(z) + r1 +
// r2_scale * r2, y2 = BN_B(y).  prm = {gammaA, betaA, gammaB, betaB}; out = {running_meanA, running_varA, running_meanB,
// running_varB (NULL: no update), meanA, invstdA, meanB, invstdB (written)}.  Backward: dy2 = gradient of y2, dskip =
// gradient of y from its other use (or NULL), stats = {meanA, invstdA, gammaA, betaA, meanB, invstdB, gammaB, betaB};
// writes dz, dy (the total gradient of y: r1's gradient, and r2's after scaling) and sums [4][C] = d betaA | d gammaA |
// d betaB | d gammaB.  Same limits as ppea_bn_fwd_channel_*.
int ppea_bn_fwd_channel_next_f32(const void* z, const float* const* prm, float* const* out, float eps, float momentum,
                                 const float* mask, const void* r1, const void* r2, float r2_scale, void* y, void* y2,
                                 int N, int C, int HW, void* stream) {
    return fwd_channel_next_impl<float>(z, prm, out, eps, momentum, mask, r1, r2, r2_scale, y, y2, N, C, HW, stream);
}
int ppea_bn_fwd_channel_next_bf16(const void* z, const float* const* prm, float* const* out, float eps, float momentum,
                                  const float* mask, const void* r1, const void* r2, float r2_scale, void* y, void* y2,
                                  int N, int C, int HW, void* stream) {
    return fwd_channel_next_impl<uint16_t>(z, prm, out, eps, momentum, mask, r1, r2, r2_scale, y, y2, N, C, HW, stream);
}
int ppea_bn_bwd_channel_next_f32(const void* dy2, const void* dskip, const void* z, const void* y, const float* const* stats,
                                 const float* mask, float inv_count, void* dz, void* dy, float* sums, int N, int C, int HW,
                                 void* stream) {
    return bwd_channel_next_impl<float>(dy2, nullptr, dskip, z, y, stats, mask, inv_count, dz, dy, sums, N, C, HW, stream);
}
int ppea_bn_bwd_channel_next_bf16(const void* dy2, const void* dskip, const void* z, const void* y, const float* const* stats,
                                  const float* mask, float inv_count, void* dz, void* dy, float* sums, int N, int C, int HW,
                                  void* stream) {
    return bwd_channel_next_impl<uint16_t>(dy2, nullptr, dskip, z, y, stats, mask, inv_count, dz, dy, sums, N, C, HW, stream);
}
// As above for a y2 with TWO consumers (a block's first 1x1 conv and its adapter, rka.py:283-289, 315-326): dy2b = the second
// consumer's gradient; the kernel starts from round(dy2 + dy2b), what the framework's separate element-wise add would have
// stored, so every result is bit-identical to that form and the add launch is gone from the chain.
int ppea_bn_bwd_channel_next_dup_f32(const void* dy2, const void* dy2b, const void* dskip, const void* z, const void* y,
                                     const float* const* stats, const float* mask, float inv_count, void* dz, void* dy,
                                     float* sums, int N, int C, int HW, void* stream) {
    return bwd_channel_next_impl<float>(dy2, dy2b, dskip, z, y, stats, mask, inv_count, dz, dy, sums, N, C, HW, stream);
}
int ppea_bn_bwd_channel_next_dup_bf16(const void* dy2, const void* dy2b, const void* dskip, const void* z, const void* y,
                                      const float* const* stats, const float* mask, float inv_count, void* dz, void* dy,
                                      float* sums, int N, int C, int HW, void* stream) {
    return bwd_channel_next_impl<uint16_t>(dy2, dy2b, dskip, z, y, stats, mask, inv_count, dz, dy, sums, N, C, HW, stream);
}

// One launch per BN for small channels (N * HW <= 16384, HW % 8 == 0, C >= 64; else PPEA_ERR_UNSUPPORTED):
// prm = {gamma1, beta1, gamma2, beta2}; out = {running_mean1, running_var1, running_mean2, running_var2 (NULL: no
// update), mean1, invstd1, mean2, invstd2 (written; kept for backward)}.  Rest as ppea_bn_apply_* / ppea_bn_bwd_*.
int ppea_bn_fwd_channel_f32(const void* z1, const void* z2, const float* const* prm, float* const* out, float eps,
                            float momentum, const float* mask, const void* r1, const void* r2, float r2_scale, void* y,
                            int act, int N, int C, int HW, void* stream) {
    return fwd_channel_impl<float>(z1, z2, prm, out, eps, momentum, mask, r1, r2, r2_scale, y, act, N, C, HW, stream);
}
int ppea_bn_fwd_channel_bf16(const void* z1, const void* z2, const float* const* prm, float* const* out, float eps,
                             float momentum, const float* mask, const void* r1, const void* r2, float r2_scale, void* y,
                             int act, int N, int C, int HW, void* stream) {
    return fwd_channel_impl<uint16_t>(z1, z2, prm, out, eps, momentum, mask, r1, r2, r2_scale, y, act, N, C, HW, stream);
}
// One BatchNorm, statistics from the producing GEMM's epilogue sums (partial [C][P][2], ppea_pwconv_stats_bf16):
// prm = {gamma, beta}; out = {running_mean, running_var (NULL: no update), mean, invstd (written)}.
int ppea_bn_fwd_channel_sums_f32(const void* z, const float* partial, int P, const float* const* prm, float* const* out,
                                 float eps, float momentum, const float* mask, const void* r1, const void* r2, float r2_scale,
                                 void* y, int act, int N, int C, int HW, void* stream) {
    return fwd_channel_sums_impl<float>(z, partial, P, prm, out, eps, momentum, mask, r1, r2, r2_scale, y, act, N, C, HW, stream);
}
int ppea_bn_fwd_channel_sums_bf16(const void* z, const float* partial, int P, const float* const* prm, float* const* out,
                                  float eps, float momentum, const float* mask, const void* r1, const void* r2, float r2_scale,
                                  void* y, int act, int N, int C, int HW, void* stream) {
    return fwd_channel_sums_impl<uint16_t>(z, partial, P, prm, out, eps, momentum, mask, r1, r2, r2_scale, y, act, N, C, HW, stream);
}
int ppea_bn_bwd_channel_f32(const void* dy, const void* z1, const void* z2, const float* const* stats, const float* mask,
                            float inv_count, const void* acc, void* dz1, void* dz2, float* sums, int act, int N, int C, int HW,
                            void* stream) {
    return bwd_channel_impl<float>(dy, nullptr, z1, z2, stats, mask, inv_count, acc, dz1, dz2, sums, act, N, C, HW, stream);
}
int ppea_bn_bwd_channel_bf16(const void* dy, const void* z1, const void* z2, const float* const* stats, const float* mask,
                            float inv_count, const void* acc, void* dz1, void* dz2, float* sums, int act, int N, int C, int HW,
                            void* stream) {
    return bwd_channel_impl<uint16_t>(dy, nullptr, z1, z2, stats, mask, inv_count, acc, dz1, dz2, sums, act, N, C, HW, stream);
}
// As above for an output with TWO consumers: dyb = the second consumer's gradient, the kernel starts from round(dy + dyb)
// (bit-identical to a separate element-wise add followed by ppea_bn_bwd_channel_*).
int ppea_bn_bwd_channel_dup_f32(const void* dy, const void* dyb, const void* z1, const void* z2, const float* const* stats,
                                const float* mask, float inv_count, const void* acc, void* dz1, void* dz2, float* sums,
                                int act, int N, int C, int HW, void* stream) {
    return bwd_channel_impl<float>(dy, dyb, z1, z2, stats, mask, inv_count, acc, dz1, dz2, sums, act, N, C, HW, stream);
}
int ppea_bn_bwd_channel_dup_bf16(const void* dy, const void* dyb, const void* z1, const void* z2, const float* const* stats,
                                 const float* mask, float inv_count, const void* acc, void* dz1, void* dz2, float* sums,
                                 int act, int N, int C, int HW, void* stream) {
    return bwd_channel_impl<uint16_t>(dy, dyb, z1, z2, stats, mask, inv_count, acc, dz1, dz2, sums, act, N, C, HW, stream);
}

// stats[8] = {mean1, invstd1, gamma1, beta1, mean2, invstd2, gamma2, beta2} (branch 2 NULL when z2 is NULL)
int ppea_bn_stats_f32(const void* z, float* partial, int N, int C, int HW, void* stream) {
    return stats_impl<float>(z, partial, N, C, HW, stream);
}
int ppea_bn_stats_bf16(const void* z, float* partial, int N, int C, int HW, void* stream) {
    return stats_impl<uint16_t>(z, partial, N, C, HW, stream);
}
int ppea_bn_finalize_sums_f32(const float* partial, int P, int C, long count, float eps, float momentum, float* mean,
                              float* var, float* invstd, float* running_mean, float* running_var, void* stream) {
    if (P <= 0 || C <= 0 || count <= 0) return PPEA_ERR_ARG;
    hipLaunchKernelGGL(bn_finalize_sums, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, partial, P, C, (float)count,
                       eps, momentum, mean, var, invstd, running_mean, running_var);
    return launch_status();
}
int ppea_bn_finalize_f32(const float* partial, int N, int C, int HW, float eps, float momentum, float* mean,
                         float* var, float* invstd, float* running_mean, float* running_var, void* stream) {
    if (N <= 0 || C <= 0 || HW <= 0) return PPEA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(bn_finalize, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, partial, N, C, HW, eps,
                       momentum, mean, var, invstd, running_mean, running_var);
    return launch_status();
}
int ppea_bn_stats_final_f32(const void* z, int N, int C, int HW, float eps, float momentum, float* mean, float* var,
                            float* invstd, float* running_mean, float* running_var, void* stream) {
    return stats_final_impl<float>(z, N, C, HW, eps, momentum, mean, var, invstd, running_mean, running_var, stream);
}
int ppea_bn_stats_final_bf16(const void* z, int N, int C, int HW, float eps, float momentum, float* mean, float* var,
                             float* invstd, float* running_mean, float* running_var, void* stream) {
    return stats_final_impl<uint16_t>(z, N, C, HW, eps, momentum, mean, var, invstd, running_mean, running_var, stream);
}
int ppea_bn_bwd_reduce_final_f32(const void* dy, const void* z1, const void* z2, const float* const* stats,
                                 const float* mask, float* sums, int act, int N, int C, int HW, void* stream) {
    return bwd_reduce_final_impl<float>(dy, nullptr, nullptr, z1, z2, stats, mask, sums, act, N, C, HW, stream);
}
int ppea_bn_bwd_reduce_final_bf16(const void* dy, const void* z1, const void* z2, const float* const* stats,
                                  const float* mask, float* sums, int act, int N, int C, int HW, void* stream) {
    return bwd_reduce_final_impl<uint16_t>(dy, nullptr, nullptr, z1, z2, stats, mask, sums, act, N, C, HW, stream);
}
// Several ranks, an output with two consumers (see ppea_bn_bwd_channel_dup_*): the reduce launch starts from dym = round(dy +
// dyb), which it also stores for the apply launch that follows the all-reduce.
int ppea_bn_bwd_reduce_final_dup_f32(const void* dy, const void* dyb, void* dym, const void* z1, const void* z2,
                                     const float* const* stats, const float* mask, float* sums, int act, int N, int C,
                                     int HW, void* stream) {
    return bwd_reduce_final_impl<float>(dy, dyb, dym, z1, z2, stats, mask, sums, act, N, C, HW, stream);
}
int ppea_bn_bwd_reduce_final_dup_bf16(const void* dy, const void* dyb, void* dym, const void* z1, const void* z2,
                                      const float* const* stats, const float* mask, float* sums, int act, int N, int C,
                                      int HW, void* stream) {
    return bwd_reduce_final_impl<uint16_t>(dy, dyb, dym, z1, z2, stats, mask, sums, act, N, C, HW, stream);
}
int ppea_bn_stats_packed_f32(const void* z, int N, int C, int HW, float* packed, void* stream) {
    return stats_packed_impl<float>(z, N, C, HW, packed, stream);
}
int ppea_bn_stats_packed_bf16(const void* z, int N, int C, int HW, float* packed, void* stream) {
    return stats_packed_impl<uint16_t>(z, N, C, HW, packed, stream);
}
int ppea_bn_finalize_packed_f32(const float* partial, int N, int C, int HW, float* packed, void* stream) {
    if (N <= 0 || C <= 0 || HW <= 0) return PPEA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(bn_finalize_packed, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, partial, N, C, HW,
                       packed);
    return launch_status();
}
int ppea_bn_sync_combine_f32(const float* gathered, int world, int C, float eps, float momentum, float* mean,
                             float* invstd, float* running_mean, float* running_var, void* stream) {
    if (world <= 0 || C <= 0) return PPEA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(bn_sync_combine, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, gathered, world, C, eps,
                       momentum, mean, invstd, running_mean, running_var);
    return launch_status();
}
int ppea_bn_apply_f32(const void* z1, const void* z2, const float* const* stats, const float* mask, const void* r1,
                      const void* r2, float r2_scale, void* y, int act, int N, int C, int HW, void* stream) {
    return apply_impl<float>(z1, z2, stats, mask, r1, r2, r2_scale, y, act, N, C, HW, stream);
}
int ppea_bn_apply_bf16(const void* z1, const void* z2, const float* const* stats, const float* mask, const void* r1,
                       const void* r2, float r2_scale, void* y, int act, int N, int C, int HW, void* stream) {
    return apply_impl<uint16_t>(z1, z2, stats, mask, r1, r2, r2_scale, y, act, N, C, HW, stream);
}
int ppea_bn_bwd_reduce_f32(const void* dy, const void* z1, const void* z2, const float* const* stats,
                           const float* mask, float* partial, int act, int N, int C, int HW, void* stream) {
    return bwd_reduce_impl<float>(dy, z1, z2, stats, mask, partial, act, N, C, HW, stream);
}
int ppea_bn_bwd_reduce_bf16(const void* dy, const void* z1, const void* z2, const float* const* stats,
                            const float* mask, float* partial, int act, int N, int C, int HW, void* stream) {
    return bwd_reduce_impl<uint16_t>(dy, z1, z2, stats, mask, partial, act, N, C, HW, stream);
}
int ppea_bn_bwd_finalize_f32(const float* partial, int N, int C, float* sums, void* stream) {
    if (N <= 0 || C <= 0) return PPEA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(bn_bwd_finalize, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, partial, N, C, sums);
    return launch_status();
}
int ppea_bn_bwd_apply_f32(const void* dy, const void* z1, const void* z2, const float* const* stats,
                          const float* mask, const float* sums, float inv_count, void* dz1, void* dz2, int act, int N,
                          int C, int HW, void* stream) {
    return bwd_apply_impl<float>(dy, z1, z2, stats, mask, sums, inv_count, dz1, dz2, act, N, C, HW, stream);
}
int ppea_bn_bwd_apply_bf16(const void* dy, const void* z1, const void* z2, const float* const* stats,
                           const float* mask, const float* sums, float inv_count, void* dz1, void* dz2, int act,
                           int N, int C, int HW, void* stream) {
    return bwd_apply_impl<uint16_t>(dy, z1, z2, stats, mask, sums, inv_count, dz1, dz2, act, N, C, HW, stream);
}
// SyncBN backward apply (bn_sync.hip's counterpart): `sums` are the all-reduced sums of the GLOBAL batch, inv_count =
// 1 / (global count); acc (shape of z1, or NULL): dz1 = round(dz1) + acc, the gradient reaching z1 through its other consumer
// (what ppea_bn_bwd_channel_* does in the one-launch form); dgb [3][C] (or NULL) = sums * gscale, the parameter gradients
// d beta | d gamma1 | d gamma2 scaled for the data-parallel mean (gscale = 1 / world).  HW % 8 == 0.
int ppea_bn_sync_bwd_apply_f32(const void* dy, const void* z1, const void* z2, const float* const* stats,
                               const float* mask, const float* sums, float inv_count, const void* acc, void* dz1, void* dz2,
                               float* dgb, float gscale, int act, int N, int C, int HW, void* stream) {
    return bwd_apply_impl<float>(dy, z1, z2, stats, mask, sums, inv_count, dz1, dz2, act, N, C, HW, stream, acc, dgb, gscale);
}
int ppea_bn_sync_bwd_apply_bf16(const void* dy, const void* z1, const void* z2, const float* const* stats,
                                const float* mask, const float* sums, float inv_count, const void* acc, void* dz1, void* dz2,
                                float* dgb, float gscale, int act, int N, int C, int HW, void* stream) {
    return bwd_apply_impl<uint16_t>(dy, z1, z2, stats, mask, sums, inv_count, dz1, dz2, act, N, C, HW, stream, acc, dgb, gscale);
}

}  // extern "C"
