// SyncBatchNorm across ranks on the fused BatchNorm kernels, gfx950.
//
// Reference: both RepLKNet encoders are nn.SyncBatchNorm (networks/replknet_adapter.py:170-180): batch statistics of
// the GLOBAL batch.  With one rank a BatchNorm (+ activation, DropPath scale, residual, adapter add) is one launch
// (bn_fused.hip); with several ranks one exchange of 2C+1 floats has to sit between the statistics and the apply, so it
// is exactly two launches around ONE collective (torch.distributed all-gather on RCCL, issued by the host between
// them):
//
//   bn_stats_channel_pk / producer-epilogue sums -> packed [mean | biased var | count]       (this rank)
//        all_gather_into_tensor -> gathered [world][pitch]
//   bn_fwd_channel_sync / bn_apply_flat_sync: Chan combine of the gathered rows INLINE (every workgroup combines the
//        channels it touches; no combine launch), running statistics, saved (mean, invstd), the fused apply -- and,
//        on request, the local statistics of the values it STORES in wire format, so that the next BatchNorm over that
//        tensor (a block's last BN followed by the next block's first one) needs no statistics launch of its own.
//
// Backward is reduce -> all-reduce of [3][C] sums -> apply (bn_fused.hip; the apply takes the residual path's gradient
// `acc` like the one-launch kernel does).  Arithmetic of the combine is that of torch's
// batch_norm_gather_stats_with_counts (count-weighted mean, M2 = sum cnt (var + d^2)).
#include "bn_common.h"

namespace {

struct Gathered {
    const float* tab;     // [world][pitch]
    int world, pitch;     // count of a row at [pitch - 1]
};

// global (mean, M2, count) of channel c from rows (mean at off + c, biased var at off + C + c)
__device__ __forceinline__ void chan_combine(const Gathered g, int off, int C, int c, float& mean, float& m2, float& total) {
    total = 0.f;
    mean = 0.f;
    for (int r = 0; r < g.world; ++r) {
        const float cnt = g.tab[(long)r * g.pitch + g.pitch - 1];
        total += cnt;
        mean += cnt * g.tab[(long)r * g.pitch + off + c];
    }
    mean /= total;
    m2 = 0.f;
    for (int r = 0; r < g.world; ++r) {
        const float cnt = g.tab[(long)r * g.pitch + g.pitch - 1];
        const float d = g.tab[(long)r * g.pitch + off + c] - mean;
        m2 += cnt * (g.tab[(long)r * g.pitch + off + C + c] + d * d);
    }
}

struct SyncPrm { const float *gamma1, *beta1, *gamma2, *beta2; float *rm1, *rv1, *rm2, *rv2, *mean1, *invstd1, *mean2, *invstd2; };

// ---- local statistics in wire format, one workgroup per channel (N * HW <= 16384) ----------------------------------------
template <typename T, bool TWO, int NV>
__global__ __launch_bounds__(TPB) void bn_stats_channel_pk(const T* __restrict__ z1, const T* __restrict__ z2,
                                                           float* __restrict__ packed, int N, int C, int HW) {
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
    if (threadIdx.x == 0) {
        packed[c] = mu1;
        packed[C + c] = q1 / cnt;
        if constexpr (TWO) { packed[2 * C + c] = mu2; packed[3 * C + c] = q2 / cnt; }
        if (c == 0) packed[(TWO ? 4 : 2) * C] = cnt;
    }
}

// per-plane partials (ppea_bn_stats_*: partial[(c*N + n)*2] = mean, M2) of one or two tensors -> wire format
__global__ void bn_finalize_packed2(const float* __restrict__ partial1, const float* __restrict__ partial2, int N, int C, int HW,
                                    float* __restrict__ packed) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int nb = partial2 != nullptr ? 2 : 1;
    if (c == 0) packed[2 * nb * C] = (float)N * (float)HW;
    if (c >= C) return;
    for (int b = 0; b < nb; ++b) {
        const float* p = (b == 0 ? partial1 : partial2) + (long)c * N * 2;
        float mean = 0.f;
        for (int n = 0; n < N; ++n) mean += p[2 * n];
        mean /= (float)N;
        float m2 = 0.f;
        for (int n = 0; n < N; ++n) {
            const float d = p[2 * n] - mean;
            m2 += p[2 * n + 1] + (float)HW * d * d;
        }
        packed[2 * b * C + c] = mean;
        packed[(2 * b + 1) * C + c] = m2 / ((float)N * (float)HW);
    }
}

// producer-epilogue partial sums [C][P][2] = (sum, sum of squares) (ppea_pwconv_stats_bf16) -> wire format; totals in
// fp64 in a fixed order, as bn_finalize_sums does for one rank
__global__ __launch_bounds__(256) void bn_sums_to_packed(const float* __restrict__ partial, int P, int C, float count,
                                                         float* __restrict__ packed) {
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
        packed[c] = (float)mean;
        packed[C + c] = (float)var;
        if (c == 0) packed[2 * C] = count;
    }
}

// ---- combine + apply, one workgroup per channel; EMIT: also the local statistics of the stored values ---------------------
template <typename T, bool TWO, bool EMIT, int NV>
__global__ __launch_bounds__(TPB) void bn_fwd_channel_sync(const T* __restrict__ z1, const T* __restrict__ z2, Gathered g,
                                                           SyncPrm p, float eps, float momentum, const float* __restrict__ mask,
                                                           const T* __restrict__ r1, const T* __restrict__ r2, float r2_scale,
                                                           T* __restrict__ y, float* __restrict__ packed_next, int act, int N,
                                                           int C, int HW) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const int hv = HW / V, total = N * hv;
    float x1[NV][V], x2[TWO ? NV : 1][V];
#pragma unroll
    for (int u = 0; u < NV; ++u) {                      // the channel's loads are in flight under the combine
        const int j = threadIdx.x + u * TPB;
        if (j < total) {
            const int n = j / hv, i = j - n * hv;
            const long off = ((long)n * C + c) * HW + i * V;
            ld8<T>(z1 + off, x1[u]);
            if constexpr (TWO) ld8<T>(z2 + off, x2[u]);
        }
    }
    float mu1, m21, tot, mu2 = 0.f, m22 = 0.f;
    chan_combine(g, 0, C, c, mu1, m21, tot);
    if constexpr (TWO) chan_combine(g, 2 * C, C, c, mu2, m22, tot);
    const float is1 = rsqrtf(m21 / tot + eps), is2 = TWO ? rsqrtf(m22 / tot + eps) : 0.f;
    if (threadIdx.x == 0) {
        p.mean1[c] = mu1; p.invstd1[c] = is1;
        if (p.rm1 != nullptr) {
            p.rm1[c] = (1.f - momentum) * p.rm1[c] + momentum * mu1;
            p.rv1[c] = (1.f - momentum) * p.rv1[c] + momentum * (m21 / fmaxf(tot - 1.f, 1.f));
        }
        if constexpr (TWO) {
            p.mean2[c] = mu2; p.invstd2[c] = is2;
            if (p.rm2 != nullptr) {
                p.rm2[c] = (1.f - momentum) * p.rm2[c] + momentum * mu2;
                p.rv2[c] = (1.f - momentum) * p.rv2[c] + momentum * (m22 / fmaxf(tot - 1.f, 1.f));
            }
        }
    }
    const float a1 = p.gamma1[c] * is1, o1 = p.beta1[c] - mu1 * a1;
    float a2 = 0.f, o2 = 0.f;
    if constexpr (TWO) { a2 = p.gamma2[c] * is2; o2 = p.beta2[c] - mu2 * a2; }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int j = threadIdx.x + u * TPB;
        if (j < total) {
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
                if constexpr (EMIT) x1[u][k] = round_as<T>(v);          // what the next BatchNorm reads back
            }
            st8<T>(y + off, o);
        } else if constexpr (EMIT) {
#pragma unroll
            for (int k = 0; k < V; ++k) x1[u][k] = 0.f;
        }
    }
    if constexpr (EMIT) {
        const float cnt = (float)N * (float)HW;
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < NV; ++u)
#pragma unroll
            for (int k = 0; k < V; ++k) s += x1[u][k];
        const float mu = block_sum(s, red) / cnt;
        float q = 0.f;
#pragma unroll
        for (int u = 0; u < NV; ++u)
            if (threadIdx.x + u * TPB < total) {
#pragma unroll
                for (int k = 0; k < V; ++k) q += (x1[u][k] - mu) * (x1[u][k] - mu);
            }
        q = block_sum(q, red);
        if (threadIdx.x == 0) {
            packed_next[c] = mu;
            packed_next[C + c] = q / cnt;
            if (c == 0) packed_next[2 * C] = cnt;
        }
    }
}

// ---- combine + apply, flat element-wise pass (HW % 8 == 0, any N * HW): a workgroup covers 2048 consecutive elements, i.e.
// a handful of planes; its first threads combine the channels of those planes into LDS.  The workgroup in which plane
// (n = 0, c) begins writes channel c's saved statistics and running statistics (exactly one writer per channel).
template <typename T>
__global__ __launch_bounds__(TPB) void bn_apply_flat_sync(const T* __restrict__ z1, const T* __restrict__ z2, Gathered g,
                                                          SyncPrm p, float eps, float momentum, const float* __restrict__ mask,
                                                          const T* __restrict__ r1, const T* __restrict__ r2, float r2_scale,
                                                          T* __restrict__ y, int act, int C, int HW, long total8) {
    __shared__ float sa1[TPB], so1[TPB], sa2[TPB], so2[TPB];
    const long t0 = (long)blockIdx.x * TPB;
    const long tl = min(total8, t0 + TPB) - 1;                     // last thread of this workgroup with work
    const long plane0 = t0 * V / HW, plane1 = tl * V / HW;
    const int np = (int)(plane1 - plane0) + 1;                     // <= TPB: a thread's 8 elements lie in one plane
    if ((int)threadIdx.x < np) {
        const long plane = plane0 + threadIdx.x;
        const int n = (int)(plane / C), c = (int)(plane - (long)n * C);
        float mu1, m21, tot;
        chan_combine(g, 0, C, c, mu1, m21, tot);
        const float is1 = rsqrtf(m21 / tot + eps);
        const float a1 = p.gamma1[c] * is1;
        sa1[threadIdx.x] = a1;
        so1[threadIdx.x] = p.beta1[c] - mu1 * a1;
        const bool owner = (n == 0) && (plane * HW >= t0 * V);
        if (owner) {
            p.mean1[c] = mu1; p.invstd1[c] = is1;
            if (p.rm1 != nullptr) {
                p.rm1[c] = (1.f - momentum) * p.rm1[c] + momentum * mu1;
                p.rv1[c] = (1.f - momentum) * p.rv1[c] + momentum * (m21 / fmaxf(tot - 1.f, 1.f));
            }
        }
        if (z2 != nullptr) {
            float mu2, m22;
            chan_combine(g, 2 * C, C, c, mu2, m22, tot);
            const float is2 = rsqrtf(m22 / tot + eps);
            const float a2 = p.gamma2[c] * is2;
            sa2[threadIdx.x] = a2;
            so2[threadIdx.x] = p.beta2[c] - mu2 * a2;
            if (owner) {
                p.mean2[c] = mu2; p.invstd2[c] = is2;
                if (p.rm2 != nullptr) {
                    p.rm2[c] = (1.f - momentum) * p.rm2[c] + momentum * mu2;
                    p.rv2[c] = (1.f - momentum) * p.rv2[c] + momentum * (m22 / fmaxf(tot - 1.f, 1.f));
                }
            }
        }
    }
    __syncthreads();
    const long t = t0 + threadIdx.x;
    if (t >= total8) return;
    const long i = t * V;
    const long plane = i / HW;
    const int n = (int)(plane / C);
    const int k0 = (int)(plane - plane0);
    const float a1 = sa1[k0], o1 = so1[k0];
    float a2 = 0.f, o2 = 0.f;
    if (z2 != nullptr) { a2 = sa2[k0]; o2 = so2[k0]; }
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
int sync_stats_impl(const void* z1, const void* z2, float* packed, float* ws, int N, int C, int HW, void* stream,
                    int (*plane_stats)(const void*, float*, int, int, int, void*)) {
    if (N <= 0 || C <= 0 || HW <= 0) return PPEA_ERR_ARG;
    if (C >= 64 && HW % V == 0 && (long)N * HW <= CHANNEL_ELEMS) {
#define PPEA_L(NV_)                                                                                                   \
        if (z2 != nullptr)                                                                                            \
            hipLaunchKernelGGL((bn_stats_channel_pk<T, true, NV_>), dim3((unsigned)C), dim3(TPB), 0, (hipStream_t)stream, \
                               (const T*)z1, (const T*)z2, packed, N, C, HW);                                         \
        else                                                                                                          \
            hipLaunchKernelGGL((bn_stats_channel_pk<T, false, NV_>), dim3((unsigned)C), dim3(TPB), 0, (hipStream_t)stream, \
                               (const T*)z1, (const T*)nullptr, packed, N, C, HW)
        PPEA_BN_NV(bn_nv(N, HW), PPEA_L);
#undef PPEA_L
        return launch_status();
    }
    if (ws == nullptr) return PPEA_ERR_ARG;
    float* p2 = z2 != nullptr ? ws + (long)N * C * 2 : nullptr;
    int err = plane_stats(z1, ws, N, C, HW, stream);
    if (err == 0 && z2 != nullptr) err = plane_stats(z2, p2, N, C, HW, stream);
    if (err != 0) return err;
    hipLaunchKernelGGL(bn_finalize_packed2, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, ws, p2, N, C, HW, packed);
    return launch_status();
}

template <typename T>
int sync_apply_impl(const void* z1, const void* z2, const float* gathered, int world, const float* const* prm,
                    float* const* outp, float eps, float momentum, const float* mask, const void* r1, const void* r2,
                    float r2_scale, void* y, float* packed_next, int act, int N, int C, int HW, void* stream) {
    if (N <= 0 || C <= 0 || HW <= 0 || world <= 0 || act < 0 || act > 2 || gathered == nullptr) return PPEA_ERR_ARG;
    if (HW % V != 0) return PPEA_ERR_UNSUPPORTED;
    const Gathered g{gathered, world, (z2 != nullptr ? 4 : 2) * C + 1};
    const SyncPrm p{prm[0], prm[1], prm[2], prm[3], outp[0], outp[1], outp[2], outp[3], outp[4], outp[5], outp[6], outp[7]};
    const hipStream_t s = (hipStream_t)stream;
    if (C >= 64 && (long)N * HW <= CHANNEL_ELEMS) {
        const dim3 grid((unsigned)C), blk(TPB);
        const T *a = (const T*)z1, *b = (const T*)z2, *q1 = (const T*)r1, *q2 = (const T*)r2;
#define PPEA_L(NV_)                                                                                                   \
        if (packed_next != nullptr) {                                                                                 \
            if (z2 != nullptr)                                                                                        \
                hipLaunchKernelGGL((bn_fwd_channel_sync<T, true, true, NV_>), grid, blk, 0, s, a, b, g, p, eps, momentum, mask, q1, q2, \
                                   r2_scale, (T*)y, packed_next, act, N, C, HW);                                      \
            else                                                                                                      \
                hipLaunchKernelGGL((bn_fwd_channel_sync<T, false, true, NV_>), grid, blk, 0, s, a, b, g, p, eps, momentum, mask, q1, q2, \
                                   r2_scale, (T*)y, packed_next, act, N, C, HW);                                      \
        } else {                                                                                                      \
            if (z2 != nullptr)                                                                                        \
                hipLaunchKernelGGL((bn_fwd_channel_sync<T, true, false, NV_>), grid, blk, 0, s, a, b, g, p, eps, momentum, mask, q1, q2, \
                                   r2_scale, (T*)y, packed_next, act, N, C, HW);                                      \
            else                                                                                                      \
                hipLaunchKernelGGL((bn_fwd_channel_sync<T, false, false, NV_>), grid, blk, 0, s, a, b, g, p, eps, momentum, mask, q1, q2, \
                                   r2_scale, (T*)y, packed_next, act, N, C, HW);                                      \
        }
        PPEA_BN_NV(bn_nv(N, HW), PPEA_L);
#undef PPEA_L
        return launch_status();
    }
    if (packed_next != nullptr) return PPEA_ERR_UNSUPPORTED;      // statistics of the output: channel-owning workgroups only
    const long total8 = (long)N * C * HW / V;
    hipLaunchKernelGGL(bn_apply_flat_sync<T>, dim3((unsigned)((total8 + TPB - 1) / TPB)), dim3(TPB), 0, s, (const T*)z1,
                       (const T*)z2, g, p, eps, momentum, mask, (const T*)r1, (const T*)r2, r2_scale, (T*)y, act, C, HW, total8);
    return launch_status();
}

}  // namespace

extern "C" {

int ppea_bn_stats_f32(const void* z, float* partial, int N, int C, int HW, void* stream);
int ppea_bn_stats_bf16(const void* z, float* partial, int N, int C, int HW, void* stream);

long ppea_bn_sync_stats_workspace_bytes(int N, int C, int HW, int two) {
    if (N <= 0 || C <= 0 || HW <= 0) return 0;
    if (C >= 64 && HW % V == 0 && (long)N * HW <= CHANNEL_ELEMS) return 0;
    return (long)N * C * 2 * (two ? 2 : 1) * (long)sizeof(float);
}
int ppea_bn_sync_stats_f32(const void* z1, const void* z2, float* packed, float* ws, int N, int C, int HW, void* stream) {
    return sync_stats_impl<float>(z1, z2, packed, ws, N, C, HW, stream, ppea_bn_stats_f32);
}
int ppea_bn_sync_stats_bf16(const void* z1, const void* z2, float* packed, float* ws, int N, int C, int HW, void* stream) {
    return sync_stats_impl<uint16_t>(z1, z2, packed, ws, N, C, HW, stream, ppea_bn_stats_bf16);
}
int ppea_bn_sync_stats_from_sums_f32(const float* sums, int P, int C, long count, float* packed, void* stream) {
    if (P <= 0 || C <= 0 || count <= 0) return PPEA_ERR_ARG;
    hipLaunchKernelGGL(bn_sums_to_packed, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, sums, P, C, (float)count, packed);
    return launch_status();
}
int ppea_bn_sync_apply_f32(const void* z1, const void* z2, const float* gathered, int world, const float* const* prm,
                           float* const* out, float eps, float momentum, const float* mask, const void* r1, const void* r2,
                           float r2_scale, void* y, float* packed_next, int act, int N, int C, int HW, void* stream) {
    return sync_apply_impl<float>(z1, z2, gathered, world, prm, out, eps, momentum, mask, r1, r2, r2_scale, y, packed_next, act,
                                  N, C, HW, stream);
}
int ppea_bn_sync_apply_bf16(const void* z1, const void* z2, const float* gathered, int world, const float* const* prm,
                            float* const* out, float eps, float momentum, const float* mask, const void* r1, const void* r2,
                            float r2_scale, void* y, float* packed_next, int act, int N, int C, int HW, void* stream) {
    return sync_apply_impl<uint16_t>(z1, z2, gathered, world, prm, out, eps, momentum, mask, r1, r2, r2_scale, y, packed_next,
                                     act, N, C, HW, stream);
}

}  // extern "C"
