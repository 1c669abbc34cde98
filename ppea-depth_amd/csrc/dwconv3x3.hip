// Depthwise 3x3 convolution, stride 1 or 2, pad 1, no bias (RepLKNet stem[1], stem[3] and the
// transitions' second conv, networks/replknet_adapter.py:414-416, 451-453).  MIOpen serves these grouped
// strided convs with its naive reference kernels (170-340 us each on MI355X); they are plain HBM-bound
// stencils: one thread per output (fwd) / input (dgrad) pixel, 9 wave-uniform taps per channel from SGPRs.
#include "common.h"

namespace {

template <typename T, int S>
__global__ __launch_bounds__(256) void dw3_fwd(const T* __restrict__ x, const float* __restrict__ w,
                                               T* __restrict__ y, int C, int H, int W, int Ho, int Wo) {
    const int plane = blockIdx.y;                          // n * C + c
    const int c = plane % C;
    const float* wc = w + c * 9;
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = wc[i];
    const T* xp = x + (long)plane * H * W;
    T* yp = y + (long)plane * Ho * Wo;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Ho * Wo; i += gridDim.x * 256) {
        const int oy = i / Wo, ox = i - oy * Wo;
        const int iy0 = oy * S - 1, ix0 = ox * S - 1;
        float acc = 0.f;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int iy = iy0 + u;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const int ix = ix0 + v;
                if (ix >= 0 && ix < W) acc = fmaf(k[u * 3 + v], ld_f32<T>(xp + (long)iy * W + ix), acc);
            }
        }
        st_f32<T>(yp + i, acc);
    }
}

// dx[iy][ix] = sum_{u,v} w[u][v] * dy[oy][ox]  with  oy*S - 1 + u = iy,  ox*S - 1 + v = ix
template <typename T, int S>
__global__ __launch_bounds__(256) void dw3_bwd(const T* __restrict__ dy, const float* __restrict__ w,
                                               T* __restrict__ dx, int C, int H, int W, int Ho, int Wo) {
    const int plane = blockIdx.y;
    const int c = plane % C;
    const float* wc = w + c * 9;
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = wc[i];
    const T* dp = dy + (long)plane * Ho * Wo;
    T* xp = dx + (long)plane * H * W;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
        const int iy = i / W, ix = i - iy * W;
        float acc = 0.f;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int ty = iy + 1 - u;
            if (ty < 0 || (ty % S) != 0) continue;
            const int oy = ty / S;
            if (oy >= Ho) continue;
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const int tx = ix + 1 - v;
                if (tx < 0 || (tx % S) != 0) continue;
                const int ox = tx / S;
                if (ox < Wo) acc = fmaf(k[u * 3 + v], ld_f32<T>(dp + (long)oy * Wo + ox), acc);
            }
        }
        st_f32<T>(xp + i, acc);
    }
}

template <typename T>
int run(bool bwd, const void* a, const float* w, void* o, int N, int C, int H, int W, int stride, void* stream) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2) || (long)N * C > 65535L * 32)
        return PPEA_ERR_UNSUPPORTED;
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const long planes = (long)N * C;
    if (planes > 65535) return PPEA_ERR_UNSUPPORTED;
    const int work = bwd ? H * W : Ho * Wo;
    int bx = (work + 255) / 256;
    if (bx > 64) bx = 64;
    dim3 g(bx, (unsigned)planes);
    hipStream_t st = (hipStream_t)stream;
    if (!bwd) {
        if (stride == 1) hipLaunchKernelGGL((dw3_fwd<T, 1>), g, dim3(256), 0, st, (const T*)a, w, (T*)o, C, H, W, Ho, Wo);
        else hipLaunchKernelGGL((dw3_fwd<T, 2>), g, dim3(256), 0, st, (const T*)a, w, (T*)o, C, H, W, Ho, Wo);
    } else {
        if (stride == 1) hipLaunchKernelGGL((dw3_bwd<T, 1>), g, dim3(256), 0, st, (const T*)a, w, (T*)o, C, H, W, Ho, Wo);
        else hipLaunchKernelGGL((dw3_bwd<T, 2>), g, dim3(256), 0, st, (const T*)a, w, (T*)o, C, H, W, Ho, Wo);
    }
    return launch_status();
}

}  // namespace

extern "C" {
// x [N,C,H,W] -> y [N,C,Ho,Wo], Ho = (H-1)/stride + 1; w [C,1,3,3] fp32
int ppea_dwconv3x3_fwd_f32(const void* x, const float* w, void* y, int N, int C, int H, int W, int stride, void* stream) {
    return run<float>(false, x, w, y, N, C, H, W, stride, stream);
}
int ppea_dwconv3x3_fwd_bf16(const void* x, const float* w, void* y, int N, int C, int H, int W, int stride, void* stream) {
    return run<uint16_t>(false, x, w, y, N, C, H, W, stride, stream);
}
// dy [N,C,Ho,Wo] -> dx [N,C,H,W]   (H, W are the INPUT sizes of the forward conv)
int ppea_dwconv3x3_bwd_data_f32(const void* dy, const float* w, void* dx, int N, int C, int H, int W, int stride, void* stream) {
    return run<float>(true, dy, w, dx, N, C, H, W, stride, stream);
}
int ppea_dwconv3x3_bwd_data_bf16(const void* dy, const float* w, void* dx, int N, int C, int H, int W, int stride, void* stream) {
    return run<uint16_t>(true, dy, w, dx, N, C, H, W, stride, stream);
}
}
