// ReflectionPad2d(1) forward / backward for the depth decoder's 3x3 convs (layers.py:119-135, Conv3x3).
// ATen's backward scatters with atomics (540 us for [12,32,192,640] bf16 on MI355X); here the backward is
// a gather: every input pixel sums the <= 4 padded positions that mirror onto it.  HBM-bound, one pass.
#include "common.h"

namespace {

__device__ __forceinline__ int refl(int i, int n) {       // padded index -1..n  ->  source index
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

template <typename T>
__global__ __launch_bounds__(256) void reflect_pad1_fwd(const T* __restrict__ in, T* __restrict__ out, long planes,
                                                        int H, int W) {
    const int Ho = H + 2, Wo = W + 2;
    const long total = planes * Ho * Wo;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho);
        const long p = i / ((long)Wo * Ho);
        out[i] = in[(p * H + refl(y - 1, H)) * W + refl(x - 1, W)];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void reflect_pad1_bwd(const T* __restrict__ dout, T* __restrict__ din, long planes,
                                                        int H, int W) {
    const int Ho = H + 2, Wo = W + 2;
    const long total = planes * H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const long p = i / ((long)W * H);
        const T* d = dout + p * Ho * Wo;
        // padded rows/cols that mirror onto (y, x): always (y+1, x+1); plus the border copies
        int ys[2] = {y + 1, -1}, xs[2] = {x + 1, -1};
        if (y == 1) ys[1] = 0; else if (y == H - 2) ys[1] = H + 1;
        if (x == 1) xs[1] = 0; else if (x == W - 2) xs[1] = W + 1;
        // H == 2 (or W == 2): index 0 and H-1 both receive a second copy; y == 1 == H-1 handled above, y == 0 == H-2:
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                if (ys[a] >= 0 && xs[b] >= 0) acc += ld_f32<T>(d + (long)ys[a] * Wo + xs[b]);
        if (H == 3 && y == 1) {            // y == 1 == H-2: both border copies exist
#pragma unroll
            for (int b = 0; b < 2; ++b)
                if (xs[b] >= 0) acc += ld_f32<T>(d + (long)(H + 1) * Wo + xs[b]);
        }
        if (W == 3 && x == 1) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
                if (ys[a] >= 0) acc += ld_f32<T>(d + (long)ys[a] * Wo + (W + 1));
            if (H == 3 && y == 1) acc += ld_f32<T>(d + (long)(H + 1) * Wo + (W + 1));
        }
        st_f32<T>(din + i, acc);
    }
}

template <typename T>
int fwd_impl(const void* in, void* out, long planes, int H, int W, void* stream) {
    if (planes <= 0 || H < 2 || W < 2) return PPEA_ERR_UNSUPPORTED;
    const long total = planes * (H + 2) * (W + 2);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(reflect_pad1_fwd<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const T*)in, (T*)out,
                       planes, H, W);
    return launch_status();
}
template <typename T>
int bwd_impl(const void* dout, void* din, long planes, int H, int W, void* stream) {
    if (planes <= 0 || H < 3 || W < 3) return PPEA_ERR_UNSUPPORTED;   // H, W == 2: use the framework op
    const long total = planes * H * W;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(reflect_pad1_bwd<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const T*)dout, (T*)din,
                       planes, H, W);
    return launch_status();
}

}  // namespace

extern "C" {
int ppea_reflect_pad1_fwd_f32(const void* in, void* out, long planes, int H, int W, void* stream) {
    return fwd_impl<float>(in, out, planes, H, W, stream);
}
int ppea_reflect_pad1_fwd_bf16(const void* in, void* out, long planes, int H, int W, void* stream) {
    return fwd_impl<uint16_t>(in, out, planes, H, W, stream);
}
int ppea_reflect_pad1_bwd_f32(const void* dout, void* din, long planes, int H, int W, void* stream) {
    return bwd_impl<float>(dout, din, planes, H, W, stream);
}
int ppea_reflect_pad1_bwd_bf16(const void* dout, void* din, long planes, int H, int W, void* stream) {
    return bwd_impl<uint16_t>(dout, din, planes, H, W, stream);
}
}
