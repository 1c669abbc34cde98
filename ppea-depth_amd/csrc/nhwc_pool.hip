// MaxPool2d(3, 2, 1) of the pose ResNet-18 (networks/resnet_encoder.py:376-392: conv1 -> bn1 -> relu -> maxpool) on
// channels_last tensors, forward and backward.  The library's NHWC max-pool backward scatters with atomics and carries
// int64 indices (190 us on [24,64,96,320] bf16, at the tail of the step's critical stream); here the forward keeps a
// one-byte window index per element and the backward GATHERS: an input pixel belongs to at most four windows, its
// gradient is the sum (fixed order, fp32) of the windows whose maximum it is -- no atomics, bitwise reproducible.
// Tie rule = torch's: the first maximum in (row, column) scan order of the in-bounds window wins (post-ReLU maps are full
// of exact zeros, so the rule decides where most of the gradient goes).  A thread moves 8 consecutive channels.
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

// x [N][H][W][C] -> y [N][Ho][Wo][C], idx [N][Ho][Wo][C] (uint8: 3 r + s of the maximum inside the 3 x 3 window whose
// top-left corner is (2 oy - 1, 2 ox - 1))
template <typename T>
__global__ __launch_bounds__(TPB) void nhwc_maxpool_fwd(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx,
                                                        int H, int W, int C, int Ho, int Wo, unsigned total_vec) {
    const unsigned v = blockIdx.x * TPB + threadIdx.x;
    if (v >= total_vec) return;
    const unsigned CT = (unsigned)C / V;
    const unsigned pix = v / CT, c0 = (v - pix * CT) * V;
    const unsigned row = pix / Wo, ox = pix - row * Wo;
    const unsigned n = row / Ho, oy = row - n * Ho;
    float best[V];
    uint32_t bi[V];
    bool first = true;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int iy = (int)oy * 2 - 1 + r;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int ix = (int)ox * 2 - 1 + s;
            if (ix < 0 || ix >= W) continue;
            float val[V];
            ld8<T>(x + ((((long)n * H + iy) * W + ix) * C + c0), val);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                if (first || val[k] > best[k] || val[k] != val[k]) { best[k] = val[k]; bi[k] = r * 3 + s; }
            }
            first = false;
        }
    }
    st8<T>(y + (long)v * V, best);
    uint2 packed;
    packed.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    packed.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
    *reinterpret_cast<uint2*>(idx + (long)v * V) = packed;
}

// dx [N][H][W][C] (every element written) from dy, idx [N][Ho][Wo][C]
template <typename T>
__global__ __launch_bounds__(TPB) void nhwc_maxpool_bwd(const T* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                        T* __restrict__ dx, int H, int W, int C, int Ho, int Wo,
                                                        unsigned total_vec) {
    const unsigned v = blockIdx.x * TPB + threadIdx.x;
    if (v >= total_vec) return;
    const unsigned CT = (unsigned)C / V;
    const unsigned pix = v / CT, c0 = (v - pix * CT) * V;
    const unsigned row = pix / W, ix = pix - row * W;
    const unsigned n = row / H, iy = row - n * H;
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    // windows that contain row iy: oy with 2 oy - 1 <= iy <= 2 oy + 1
    const int oy_lo = ((int)iy) / 2, oy_hi = ((int)iy + 1) / 2;          // equal for even iy
    const int ox_lo = ((int)ix) / 2, ox_hi = ((int)ix + 1) / 2;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        if (oy >= Ho) continue;
        const int r = (int)iy - (2 * oy - 1);
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            if (ox >= Wo) continue;
            const int s = (int)ix - (2 * ox - 1);
            const uint32_t want = (uint32_t)(r * 3 + s);
            const long o = ((((long)n * Ho + oy) * Wo + ox) * C + c0);
            const uint2 pk = *reinterpret_cast<const uint2*>(idx + o);
            float g[V];
            ld8<T>(dy + o, g);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const uint32_t id = ((k < 4 ? pk.x : pk.y) >> (8 * (k & 3))) & 0xffu;
                if (id == want) acc[k] += g[k];
            }
        }
    }
    st8<T>(dx + (long)v * V, acc);
}

template <typename T>
int fwd_impl(const void* x, void* y, void* idx, int N, int H, int W, int C, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % V) != 0) return PPEA_ERR_UNSUPPORTED;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long total = (long)N * Ho * Wo * (C / V);
    if (total >= (1L << 32)) return PPEA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(nhwc_maxpool_fwd<T>, dim3((unsigned)((total + TPB - 1) / TPB)), dim3(TPB), 0, (hipStream_t)stream,
                       (const T*)x, (T*)y, (uint8_t*)idx, H, W, C, Ho, Wo, (unsigned)total);
    return launch_status();
}

template <typename T>
int bwd_impl(const void* dy, const void* idx, void* dx, int N, int H, int W, int C, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % V) != 0) return PPEA_ERR_UNSUPPORTED;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long total = (long)N * H * W * (C / V);
    if (total >= (1L << 32)) return PPEA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(nhwc_maxpool_bwd<T>, dim3((unsigned)((total + TPB - 1) / TPB)), dim3(TPB), 0, (hipStream_t)stream,
                       (const T*)dy, (const uint8_t*)idx, (T*)dx, H, W, C, Ho, Wo, (unsigned)total);
    return launch_status();
}

}  // namespace

extern "C" {

// x [N][H][W][C] channels-last (C % 8 == 0) -> y, idx [N][Ho][Wo][C], Ho = (H - 1) / 2 + 1.
int ppea_nhwc_maxpool3x3s2_fwd_f32(const void* x, void* y, void* idx, int N, int H, int W, int C, void* stream) {
    return fwd_impl<float>(x, y, idx, N, H, W, C, stream);
}
int ppea_nhwc_maxpool3x3s2_fwd_bf16(const void* x, void* y, void* idx, int N, int H, int W, int C, void* stream) {
    return fwd_impl<uint16_t>(x, y, idx, N, H, W, C, stream);
}
int ppea_nhwc_maxpool3x3s2_bwd_f32(const void* dy, const void* idx, void* dx, int N, int H, int W, int C, void* stream) {
    return bwd_impl<float>(dy, idx, dx, N, H, W, C, stream);
}
int ppea_nhwc_maxpool3x3s2_bwd_bf16(const void* dy, const void* idx, void* dx, int N, int H, int W, int C, void* stream) {
    return bwd_impl<uint16_t>(dy, idx, dx, N, H, W, C, stream);
}

}  // extern "C"
