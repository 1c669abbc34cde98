// Shared helpers of the BatchNorm kernels (bn_fused.hip: one rank; bn_sync.hip: SyncBN across ranks).
#pragma once
#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int V = 8;         // elements per thread per step on the 16-byte path
constexpr int CH_VECS = 8;   // 16-byte vectors per thread of a channel-owning workgroup: N * HW <= 256 * 8 * 8 = 16384 elements
constexpr long CHANNEL_ELEMS = 16384;

template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&o)[V]);
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&o)[V]) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
template <> __device__ __forceinline__ void ld8<uint16_t>(const uint16_t* p, float (&o)[V]) {
    const uint4 a = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        o[2 * k] = __uint_as_float(w[k] << 16);
        o[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
    }
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float (&v)[V]);
template <> __device__ __forceinline__ void st8<float>(float* p, const float (&v)[V]) {
    reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void st8<uint16_t>(uint16_t* p, const float (&v)[V]) {
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t lo = __builtin_bit_cast(uint16_t, (__bf16)v[2 * k]);
        const uint32_t hi = __builtin_bit_cast(uint16_t, (__bf16)v[2 * k + 1]);
        w[k] = lo | (hi << 16);
    }
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}

// value a store of T would keep (bf16: round to nearest even; fp32: unchanged)
template <typename T> __device__ __forceinline__ float round_as(float v);
template <> __device__ __forceinline__ float round_as<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_as<uint16_t>(float v) { return __uint_as_float((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)v) << 16); }

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

struct Branch {
    const float* mean;
    const float* invstd;
    const float* gamma;
    const float* beta;
};

// erf for the bf16 kernels: Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7 (the bf16 result's own rounding is 2^-9 relative), one
// reciprocal, one exponential and five fused multiply-adds instead of the library's branchy ~35-instruction erff -- the
// BatchNorm + GELU pass over the FFN's hidden tensor was VALU-bound on it ([12,2048,12,40]: 47 MB in 25 us).  Also returns
// e = exp(-x^2), which is the Gaussian of GELU's derivative.  The fp32 kernels keep erff / expf.
__device__ __forceinline__ float erf_fast(float x, float& e) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
    e = __expf(-ax * ax);
    float p = 1.061405429f;
    p = fmaf(p, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    return copysignf(fmaf(-p * t, e, 1.f), x);
}
template <typename T>
__device__ __forceinline__ float act_fwd(float u, int act) {
    if (act == 1) return fmaxf(u, 0.f);
    if (act == 2) {
        if constexpr (sizeof(T) == 2) {
            float e;
            return 0.5f * u * (1.f + erf_fast(u * 0.70710678118654752f, e));
        } else {
            return 0.5f * u * (1.f + erff(u * 0.70710678118654752f));
        }
    }
    return u;
}
template <typename T>
__device__ __forceinline__ float act_bwd(float u, int act) {
    if (act == 1) return u > 0.f ? 1.f : 0.f;
    if (act == 2) {
        if constexpr (sizeof(T) == 2) {
            float e;
            const float cdf = 0.5f * (1.f + erf_fast(u * 0.70710678118654752f, e));
            return cdf + u * (0.39894228040143268f * e);
        } else {
            const float cdf = 0.5f * (1.f + erff(u * 0.70710678118654752f));
            const float pdf = 0.39894228040143268f * expf(-0.5f * u * u);
            return cdf + u * pdf;
        }
    }
    return 1.f;
}

// Vectors per thread of a channel-owning workgroup: the kernels keep the channel in registers, NV x 8 values per array and
// thread -- sized for the channel at hand (12 x 40 planes at batch 12: 3, not the 8 of the 16384-element limit) the register
// count lets 8 workgroups share a CU instead of 3, and their load / reduce / store phases overlap.
#define PPEA_BN_NV(nv_, LAUNCH_)                                                                   \
    do {                                                                                          \
        if ((nv_) <= 1) { LAUNCH_(1); } else if ((nv_) == 2) { LAUNCH_(2); } else if ((nv_) == 3) { LAUNCH_(3); }   \
        else if ((nv_) == 4) { LAUNCH_(4); } else { LAUNCH_(8); }                                  \
    } while (0)
static inline int bn_nv(int N, int HW) { return (int)(((long)N * (HW / V) + TPB - 1) / TPB); }

}  // namespace
