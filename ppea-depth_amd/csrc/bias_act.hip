// Bias + ELU after the decoders' 3x3 convolutions (layers.py:103-116 ConvBlock: Conv3x3 -> ELU), NCHW.
//   forward : y = elu(z + b[c])                                   (one pass instead of conv-bias + ELU)
//   backward: g = dy * elu'(u), elu'(u) = 1 for u > 0 else y + 1;  dz = g;  partial[plane][chunk] = sum g
// The bias gradient leaves the backward pass as per-(plane, chunk) partial sums: the library's bias gradient is a
// generic strided reduction over [N,C,H,W] that costs 60-170 us per layer at decoder resolutions.
#include "common.h"

namespace {

constexpr int TPB = 256, V = 8;

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

// grid = planes * chunks blocks; block (plane, chunk) covers elements [chunk * per, min(HW, (chunk+1) * per))
template <typename T, bool BWD>
__global__ __launch_bounds__(TPB) void bias_elu_kernel(const T* __restrict__ a, const T* __restrict__ yb,
                                                       const void* __restrict__ bias, int bias_bf16,
                                                       T* __restrict__ out, float* __restrict__ partial, int C, int HW,
                                                       int chunks, int per) {
    __shared__ float red[4];
    const int plane = blockIdx.x / chunks, chunk = blockIdx.x - plane * chunks;
    const int c = plane % C;
    float bv = 0.f;
    if (!BWD)
        bv = bias_bf16 ? bf16_to_f32(reinterpret_cast<const uint16_t*>(bias)[c]) : reinterpret_cast<const float*>(bias)[c];
    const long base = (long)plane * HW;
    const int i0 = chunk * per, i1 = min(HW, i0 + per);
    const int iv = ((HW % V) == 0) ? i1 : i0;          // per is a multiple of V
    float s = 0.f;
    for (int i = i0 + threadIdx.x * V; i < iv; i += TPB * V) {
        float x[V], o[V];
        ld8<T>(a + base + i, x);
        if (!BWD) {
#pragma unroll
            for (int k = 0; k < V; ++k) { const float u = x[k] + bv; o[k] = u > 0.f ? u : expm1f(u); }
        } else {
            float y[V];
            ld8<T>(yb + base + i, y);
#pragma unroll
            for (int k = 0; k < V; ++k) { o[k] = x[k] * (y[k] > 0.f ? 1.f : y[k] + 1.f); s += o[k]; }
        }
        st8<T>(out + base + i, o);
    }
    for (int i = iv + threadIdx.x; i < i1; i += TPB) {
        const float x = ld_f32<T>(a + base + i);
        float o;
        if (!BWD) { const float u = x + bv; o = u > 0.f ? u : expm1f(u); }
        else { const float y = ld_f32<T>(yb + base + i); o = x * (y > 0.f ? 1.f : y + 1.f); s += o; }
        st_f32<T>(out + base + i, o);
    }
    if (BWD) {
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

inline void plan(long planes, int HW, int& chunks, int& per) {
    chunks = 1;
    while (planes * chunks < 2048 && HW / (chunks * 2) >= 2048) chunks *= 2;
    per = (((HW + chunks - 1) / chunks) + V - 1) / V * V;
}

template <typename T>
int run_fwd(const void* z, const void* bias, int bias_bf16, void* y, int N, int C, int HW, void* stream) {
    if (N <= 0 || C <= 0 || HW <= 0 || bias == nullptr) return PPEA_ERR_UNSUPPORTED;
    int chunks, per;
    plan((long)N * C, HW, chunks, per);
    hipLaunchKernelGGL((bias_elu_kernel<T, false>), dim3((unsigned)((long)N * C * chunks)), dim3(TPB), 0,
                       (hipStream_t)stream, (const T*)z, (const T*)nullptr, bias, bias_bf16, (T*)y, (float*)nullptr, C,
                       HW, chunks, per);
    return launch_status();
}
template <typename T>
int run_bwd(const void* dy, const void* y, void* dz, float* partial, int N, int C, int HW, void* stream) {
    if (N <= 0 || C <= 0 || HW <= 0) return PPEA_ERR_UNSUPPORTED;
    int chunks, per;
    plan((long)N * C, HW, chunks, per);
    hipLaunchKernelGGL((bias_elu_kernel<T, true>), dim3((unsigned)((long)N * C * chunks)), dim3(TPB), 0,
                       (hipStream_t)stream, (const T*)dy, (const T*)y, nullptr, 0, (T*)dz, partial, C, HW, chunks, per);
    return launch_status();
}

}  // namespace

extern "C" {

// number of partial sums per plane the backward writes: partial is [N*C][chunks] fp32
int ppea_bias_elu_chunks(int N, int C, int HW) {
    int chunks, per;
    plan((long)N * C, HW, chunks, per);
    return chunks;
}
int ppea_bias_elu_fwd_f32(const void* z, const void* bias, int bias_bf16, void* y, int N, int C, int HW, void* stream) {
    return run_fwd<float>(z, bias, bias_bf16, y, N, C, HW, stream);
}
int ppea_bias_elu_fwd_bf16(const void* z, const void* bias, int bias_bf16, void* y, int N, int C, int HW, void* stream) {
    return run_fwd<uint16_t>(z, bias, bias_bf16, y, N, C, HW, stream);
}
int ppea_bias_elu_bwd_f32(const void* dy, const void* y, void* dz, float* partial, int N, int C, int HW, void* stream) {
    return run_bwd<float>(dy, y, dz, partial, N, C, HW, stream);
}
int ppea_bias_elu_bwd_bf16(const void* dy, const void* y, void* dz, float* partial, int N, int C, int HW, void* stream) {
    return run_bwd<uint16_t>(dy, y, dz, partial, N, C, HW, stream);
}

}  // extern "C"
