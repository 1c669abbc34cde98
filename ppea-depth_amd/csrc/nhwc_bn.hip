// Training-mode BatchNorm (+ReLU, + residual added BEFORE the activation) for channels_last tensors: the ResNet-18
// pose trunk (reference networks/resnet_encoder.py:25-72: relu(bn1(conv1 x)), relu(bn2(conv2 .) + identity)), which
// runs in NHWC because the library's implicit-GEMM convolutions are NHWC-native.
//
// The tensor is G consecutive sub-batches, each a [P][C] matrix (P = pixels of the sub-batch, C contiguous), normalised
// with its own statistics and updating the running statistics in turn: this is how the two frame pairs of a step go
// through the trunk as ONE 2B batch with per-pair statistics (see GroupBN in networks/resnet_encoder.py).
//
//   forward : stats    partial[slab][{sum, sumsq}][C]
//             finalize mean, invstd (+ running statistics, unbiased variance), ab[{a, b}][C]: y = a x + b
//             apply    y = act(a x + b + res)
//   backward: g = dy * act'(u), u = a x + b + res
//             reduce   partial[slab][{sum g, sum g xhat}][C]
//             finalize k[{S1/P, S2/P}][C], dgamma = S2, dbeta = S1
//             apply    dx = a (g - S1/P - xhat S2/P),  dres = g
// C % 8 == 0, C <= 2048; threads cover 8 consecutive channels with 16-byte accesses.
#include "common.h"

namespace {

constexpr int TPB = 256, V = 8, MAX_SLABS = 256;

template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&o)[V]);
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&o)[V]) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
template <> __device__ __forceinline__ void ld8<uint16_t>(const uint16_t* p, float (&o)[V]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xffff0000u);
    o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xffff0000u);
    o[4] = __uint_as_float(u.z << 16); o[5] = __uint_as_float(u.z & 0xffff0000u);
    o[6] = __uint_as_float(u.w << 16); o[7] = __uint_as_float(u.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float (&v)[V]);
template <> __device__ __forceinline__ void st8<float>(float* p, const float (&v)[V]) {
    reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void st8<uint16_t>(uint16_t* p, const float (&v)[V]) {
    uint32_t h[V];
#pragma unroll
    for (int i = 0; i < V; ++i) h[i] = f32_to_bf16(v[i]);
    *reinterpret_cast<uint4*>(p) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
}

// Two per-channel sums over the rows of a slab.  MODE 0: (x, x^2).  MODE 1: (g, g * xhat) with g = dy * act'(u).
// block = 256 threads = RL row lanes x CT channel threads (CT = C / 8); LDS reduction over the row lanes.
template <typename T, int MODE>
__global__ __launch_bounds__(TPB) void nhwc_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                          const T* __restrict__ res, const float* __restrict__ stats,
                                                          const float* __restrict__ ab, float* __restrict__ partial,
                                                          int P, int C, int rows_per_slab, int act) {
    extern __shared__ float sh[];                          // [2][RL][C]
    const int grp = blockIdx.y;                            // sub-batch: its rows, its statistics, its partial sums
    x += (long)grp * P * C;
    if (MODE == 1) { dy += (long)grp * P * C; if (res != nullptr) res += (long)grp * P * C; stats += grp * 3 * C; ab += grp * 2 * C; }
    partial += (long)grp * gridDim.x * 2 * C;
    const int CT = C / V, RL = TPB / CT;
    const int ct = threadIdx.x % CT, rl = threadIdx.x / CT;
    const int c0 = ct * V;
    const int r0 = blockIdx.x * rows_per_slab, r1 = min(P, r0 + rows_per_slab);
    float s0[V], s1[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { s0[k] = 0.f; s1[k] = 0.f; }
    float mean[V], istd[V], a[V], b[V];
    if (MODE == 1) {
        ld8<float>(stats + c0, mean); ld8<float>(stats + C + c0, istd);
        ld8<float>(ab + c0, a); ld8<float>(ab + C + c0, b);
    }
    if (rl < RL) {
        for (int r = r0 + rl; r < r1; r += RL) {
            float xv[V];
            ld8<T>(x + (long)r * C + c0, xv);
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < V; ++k) { s0[k] += xv[k]; s1[k] += xv[k] * xv[k]; }
            } else {
                float d[V], q[V];
                ld8<T>(dy + (long)r * C + c0, d);
                if (res != nullptr) ld8<T>(res + (long)r * C + c0, q);
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    float u = a[k] * xv[k] + b[k];
                    if (res != nullptr) u += q[k];
                    const float g = (act == 1 && !(u > 0.f)) ? 0.f : d[k];
                    s0[k] += g;
                    s1[k] += g * ((xv[k] - mean[k]) * istd[k]);
                }
            }
        }
    }
    if (rl < RL) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            sh[(0 * RL + rl) * C + c0 + k] = s0[k];
            sh[(1 * RL + rl) * C + c0 + k] = s1[k];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += TPB) {
        const int which = c / C, cc = c - which * C;
        float t = 0.f;
        for (int j = 0; j < RL; ++j) t += sh[(which * RL + j) * C + cc];
        partial[((long)blockIdx.x * 2 + which) * C + cc] = t;
    }
}

// Finalize kernels: block = 16 channels x 16 slab lanes (a single thread walking 256 slab partials per channel is a
// 60 us chain of dependent loads); the lanes' double-precision sums meet in LDS.
constexpr int FC = 16, FL = 16;

__device__ __forceinline__ void slab_sums(const float* __restrict__ pp, int slabs, int C, int c, int lane, double& s,
                                          double& q, double (*sh)[FL][FC]) {
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int j = lane; j < slabs; j += FL) { a += pp[((long)j * 2) * C + c]; b += pp[((long)j * 2 + 1) * C + c]; }
    __syncthreads();                                       // previous use of sh is over
    sh[0][lane][threadIdx.x] = a; sh[1][lane][threadIdx.x] = b;
    __syncthreads();
    s = 0.0; q = 0.0;
    if (lane == 0)
        for (int l = 0; l < FL; ++l) { s += sh[0][l][threadIdx.x]; q += sh[1][l][threadIdx.x]; }
}

__global__ __launch_bounds__(FC * FL) void nhwc_bn_finalize_kernel(
    const float* __restrict__ partial, int slabs, int P, int C, int G, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float momentum, float* __restrict__ stats /*[G][mean|invstd|unbiased var][C]*/,
    float* __restrict__ ab /*[G][a|b][C]*/, float* __restrict__ running_mean, float* __restrict__ running_var) {
    __shared__ double sh[2][FL][FC];
    const int c = blockIdx.x * FC + threadIdx.x, lane = threadIdx.y;
    for (int g = 0; g < G; ++g) {                          // in order: each sub-batch updates the running statistics
        double s, q;
        slab_sums(partial + (long)g * slabs * 2 * C, slabs, C, c, lane, s, q, sh);
        if (lane != 0 || c >= C) continue;
        const double mean = s / P;
        double var = q / P - mean * mean;
        if (var < 0.0) var = 0.0;
        const float istd = rsqrtf((float)var + eps);
        const float unbiased = (float)(var * ((double)P / (double)max(P - 1, 1)));
        float* st = stats + g * 3 * C;
        st[c] = (float)mean; st[C + c] = istd; st[2 * C + c] = unbiased;
        const float a = gamma[c] * istd;
        ab[g * 2 * C + c] = a; ab[g * 2 * C + C + c] = beta[c] - (float)mean * a;
        if (running_mean != nullptr) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
        }
    }
}

__global__ __launch_bounds__(FC * FL) void nhwc_bn_bwd_finalize_kernel(
    const float* __restrict__ partial, int slabs, int P, int C, int G, float* __restrict__ k /*[G][S1/P | S2/P][C]*/,
    float* __restrict__ dgamma_dbeta /*[S2 | S1][C], summed over sub-batches*/) {
    __shared__ double sh[2][FL][FC];
    const int c = blockIdx.x * FC + threadIdx.x, lane = threadIdx.y;
    double t1 = 0.0, t2 = 0.0;
    for (int g = 0; g < G; ++g) {
        double s1, s2;
        slab_sums(partial + (long)g * slabs * 2 * C, slabs, C, c, lane, s1, s2, sh);
        if (lane != 0 || c >= C) continue;
        k[g * 2 * C + c] = (float)(s1 / P); k[g * 2 * C + C + c] = (float)(s2 / P);
        t1 += s1; t2 += s2;
    }
    if (lane == 0 && c < C) { dgamma_dbeta[c] = (float)t2; dgamma_dbeta[C + c] = (float)t1; }
}

// FWD: y = act(a x + b + res).  BWD: dx = a (g - k1 - xhat k2), dres = g.
template <typename T, bool BWD>
__global__ __launch_bounds__(TPB) void nhwc_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                         const T* __restrict__ res, const float* __restrict__ stats,
                                                         const float* __restrict__ ab, const float* __restrict__ kk,
                                                         T* __restrict__ out, T* __restrict__ dres, int C, int act,
                                                         unsigned total_vec, unsigned P) {
    const unsigned v = blockIdx.x * TPB + threadIdx.x;
    if (v >= total_vec) return;
    const unsigned CT = (unsigned)C / V;
    const unsigned pix = v / CT;
    const unsigned c0 = (v - pix * CT) * V;
    const unsigned grp = pix / P;                          // sub-batch of this pixel
    ab += grp * 2 * C;
    if (BWD) { stats += grp * 3 * C; kk += grp * 2 * C; }
    const long off = (long)v * V;
    float a[V], b[V], xv[V], o[V], q[V];
    ld8<float>(ab + c0, a); ld8<float>(ab + C + c0, b);
    ld8<T>(x + off, xv);
    if (res != nullptr) ld8<T>(res + off, q);
    if (!BWD) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float u = a[k] * xv[k] + b[k];
            if (res != nullptr) u += q[k];
            o[k] = (act == 1) ? fmaxf(u, 0.f) : u;
        }
        st8<T>(out + off, o);
    } else {
        float mean[V], istd[V], k1[V], k2[V], d[V], g[V];
        ld8<float>(stats + c0, mean); ld8<float>(stats + C + c0, istd);
        ld8<float>(kk + c0, k1); ld8<float>(kk + C + c0, k2);
        ld8<T>(dy + off, d);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float u = a[k] * xv[k] + b[k];
            if (res != nullptr) u += q[k];
            g[k] = (act == 1 && !(u > 0.f)) ? 0.f : d[k];
            o[k] = a[k] * (g[k] - k1[k] - ((xv[k] - mean[k]) * istd[k]) * k2[k]);
        }
        st8<T>(out + off, o);
        if (dres != nullptr) st8<T>(dres + off, g);
    }
}

inline bool ok_shape(int P, int C) {                       // C / 8 channel threads must tile the 256-thread block
    if (P <= 0 || C < V || (C % V) != 0) return false;
    const int ct = C / V;
    return ct <= TPB && (TPB % ct) == 0;
}

inline int plan_slabs(int P, int C, int& rows) {
    const int RL = TPB / (C / V);
    int slabs = (P + RL * 8 - 1) / (RL * 8);               // >= 8 rows per row lane
    if (slabs > MAX_SLABS) slabs = MAX_SLABS;
    if (slabs < 1) slabs = 1;
    rows = (P + slabs - 1) / slabs;
    return (P + rows - 1) / rows;
}

template <typename T, int MODE>
int reduce_impl(const void* x, const void* dy, const void* res, const float* stats, const float* ab, float* partial,
                int P, int C, int G, int act, void* stream) {
    if (!ok_shape(P, C) || G < 1 || G > 65535) return PPEA_ERR_UNSUPPORTED;
    int rows;
    const int slabs = plan_slabs(P, C, rows);
    const int RL = TPB / (C / V);
    hipLaunchKernelGGL((nhwc_reduce_kernel<T, MODE>), dim3(slabs, G), dim3(TPB), (size_t)2 * RL * C * sizeof(float),
                       (hipStream_t)stream, (const T*)x, (const T*)dy, (const T*)res, stats, ab, partial, P, C, rows, act);
    return launch_status();
}

template <typename T, bool BWD>
int apply_impl(const void* x, const void* dy, const void* res, const float* stats, const float* ab, const float* kk,
               void* out, void* dres, int P, int C, int G, int act, void* stream) {
    if (!ok_shape(P, C) || G < 1 || (long)G * P * C / V > 0x7fffffffL) return PPEA_ERR_UNSUPPORTED;
    const unsigned total = (unsigned)((long)G * P * C / V);
    hipLaunchKernelGGL((nhwc_apply_kernel<T, BWD>), dim3((total + TPB - 1) / TPB), dim3(TPB), 0, (hipStream_t)stream,
                       (const T*)x, (const T*)dy, (const T*)res, stats, ab, kk, (T*)out, (T*)dres, C, act, total, (unsigned)P);
    return launch_status();
}

}  // namespace

extern "C" {

// x (and dy, res, y, dx, dres) hold G consecutive sub-batches of P pixels each: [G][P][C].
// slabs the reduce kernels write per sub-batch for (P, C): partial must hold G * slabs * 2 * C floats
int ppea_nhwc_bn_slabs(int P, int C) {
    if (!ok_shape(P, C)) return PPEA_ERR_UNSUPPORTED;
    int rows;
    return plan_slabs(P, C, rows);
}
int ppea_nhwc_bn_stats_f32(const void* x, float* partial, int P, int C, int G, void* stream) {
    return reduce_impl<float, 0>(x, nullptr, nullptr, nullptr, nullptr, partial, P, C, G, 0, stream);
}
int ppea_nhwc_bn_stats_bf16(const void* x, float* partial, int P, int C, int G, void* stream) {
    return reduce_impl<uint16_t, 0>(x, nullptr, nullptr, nullptr, nullptr, partial, P, C, G, 0, stream);
}
// stats [G][3][C] = mean | invstd | unbiased var; ab [G][2][C]; running_* (may be NULL) updated once per sub-batch
int ppea_nhwc_bn_finalize_f32(const float* partial, int P, int C, int G, const float* gamma, const float* beta,
                              float eps, float momentum, float* stats, float* ab, float* running_mean,
                              float* running_var, void* stream) {
    if (!ok_shape(P, C) || G < 1) return PPEA_ERR_UNSUPPORTED;
    int rows;
    const int slabs = plan_slabs(P, C, rows);
    hipLaunchKernelGGL(nhwc_bn_finalize_kernel, dim3((C + FC - 1) / FC), dim3(FC, FL), 0, (hipStream_t)stream, partial, slabs, P,
                       C, G, gamma, beta, eps, momentum, stats, ab, running_mean, running_var);
    return launch_status();
}
int ppea_nhwc_bn_apply_f32(const void* x, const void* res, const float* ab, void* y, int P, int C, int G, int act,
                           void* stream) {
    return apply_impl<float, false>(x, nullptr, res, nullptr, ab, nullptr, y, nullptr, P, C, G, act, stream);
}
int ppea_nhwc_bn_apply_bf16(const void* x, const void* res, const float* ab, void* y, int P, int C, int G, int act,
                            void* stream) {
    return apply_impl<uint16_t, false>(x, nullptr, res, nullptr, ab, nullptr, y, nullptr, P, C, G, act, stream);
}
int ppea_nhwc_bn_bwd_reduce_f32(const void* x, const void* dy, const void* res, const float* stats, const float* ab,
                                float* partial, int P, int C, int G, int act, void* stream) {
    return reduce_impl<float, 1>(x, dy, res, stats, ab, partial, P, C, G, act, stream);
}
int ppea_nhwc_bn_bwd_reduce_bf16(const void* x, const void* dy, const void* res, const float* stats, const float* ab,
                                 float* partial, int P, int C, int G, int act, void* stream) {
    return reduce_impl<uint16_t, 1>(x, dy, res, stats, ab, partial, P, C, G, act, stream);
}
// k [G][2][C] = S1/P | S2/P;  dgamma_dbeta [2][C] = S2 | S1 summed over the sub-batches
int ppea_nhwc_bn_bwd_finalize_f32(const float* partial, int P, int C, int G, float* k, float* dgamma_dbeta,
                                  void* stream) {
    if (!ok_shape(P, C) || G < 1) return PPEA_ERR_UNSUPPORTED;
    int rows;
    const int slabs = plan_slabs(P, C, rows);
    hipLaunchKernelGGL(nhwc_bn_bwd_finalize_kernel, dim3((C + FC - 1) / FC), dim3(FC, FL), 0, (hipStream_t)stream, partial,
                       slabs, P, C, G, k, dgamma_dbeta);
    return launch_status();
}
int ppea_nhwc_bn_bwd_apply_f32(const void* x, const void* dy, const void* res, const float* stats, const float* ab,
                               const float* k, void* dx, void* dres, int P, int C, int G, int act, void* stream) {
    return apply_impl<float, true>(x, dy, res, stats, ab, k, dx, dres, P, C, G, act, stream);
}
int ppea_nhwc_bn_bwd_apply_bf16(const void* x, const void* dy, const void* res, const float* stats, const float* ab,
                                const float* k, void* dx, void* dres, int P, int C, int G, int act, void* stream) {
    return apply_impl<uint16_t, true>(x, dy, res, stats, ab, k, dx, dres, P, C, G, act, stream);
}

}  // extern "C"
