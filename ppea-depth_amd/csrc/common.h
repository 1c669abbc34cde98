// Shared device/host helpers for libppea_depth.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PPEA_ABI_VERSION 12
#define PPEA_ERR_UNSUPPORTED (-1)
#define PPEA_ERR_ARG (-2)
#define WAVE 64

// bf16 <-> f32 (storage type uint16_t).  Plain casts through __hip_bfloat16 would pull in
// headers we do not need; round-to-nearest-even on the bit pattern, NaN kept a NaN.
__device__ __forceinline__ float bf16_to_f32(uint16_t v) {
    return __uint_as_float(((uint32_t)v) << 16);
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

template <typename T> __device__ __forceinline__ float ld_f32(const T* p);
template <> __device__ __forceinline__ float ld_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_f32<uint16_t>(const uint16_t* p) { return bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void st_f32(T* p, float v);
template <> __device__ __forceinline__ void st_f32<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_f32<uint16_t>(uint16_t* p, float v) { *p = f32_to_bf16(v); }

// 64-lane sum via DPP-backed shuffles; result valid in every lane.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

static inline int launch_status() { return (int)hipGetLastError(); }
