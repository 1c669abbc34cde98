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


// ---- bf16, 8 outputs per thread ---------------------------------------------------------------------------
// The stem runs these at 96x320 / 192x640 on [12,128,...] tensors (94 MB in, 94 MB out): one 16-byte load per
// input row piece (+ one scalar per side), one 16-byte store; the row taps stay in registers.
struct Bf8 { float v[8]; };
__device__ __forceinline__ Bf8 ld8(const uint16_t* p) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    Bf8 r;
    r.v[0] = __uint_as_float(u.x << 16); r.v[1] = __uint_as_float(u.x & 0xffff0000u);
    r.v[2] = __uint_as_float(u.y << 16); r.v[3] = __uint_as_float(u.y & 0xffff0000u);
    r.v[4] = __uint_as_float(u.z << 16); r.v[5] = __uint_as_float(u.z & 0xffff0000u);
    r.v[6] = __uint_as_float(u.w << 16); r.v[7] = __uint_as_float(u.w & 0xffff0000u);
    return r;
}
__device__ __forceinline__ void st8(uint16_t* p, const float (&a)[8]) {
    uint32_t h[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = f32_to_bf16(a[i]);
    *reinterpret_cast<uint4*>(p) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
}

// stride 1 (forward, and data gradient with the taps reversed by the caller flag FLIP): W % 8 == 0
template <bool FLIP>
__global__ __launch_bounds__(256) void dw3v_s1(const uint16_t* __restrict__ x, const float* __restrict__ w,
                                               uint16_t* __restrict__ y, int C, int H, int W, long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int W8 = W >> 3;
    const int x0 = (int)(idx % W8) * 8;
    const int oy = (int)((idx / W8) % H);
    const long plane = idx / ((long)W8 * H);
    const float* wc = w + (plane % C) * 9;
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = FLIP ? wc[8 - i] : wc[i];
    const uint16_t* xp = x + plane * (long)H * W;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int iy = oy - 1 + u;
        if (iy < 0 || iy >= H) continue;
        const uint16_t* row = xp + (long)iy * W;
        const Bf8 c = ld8(row + x0);
        float in[10];
        in[0] = x0 > 0 ? bf16_to_f32(row[x0 - 1]) : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) in[j + 1] = c.v[j];
        in[9] = x0 + 8 < W ? bf16_to_f32(row[x0 + 8]) : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            acc[j] = fmaf(k[u * 3 + 2], in[j + 2], fmaf(k[u * 3 + 1], in[j + 1], fmaf(k[u * 3], in[j], acc[j])));
    }
    st8(y + (plane * H + oy) * (long)W + x0, acc);
}

// stride 2 forward: Wo % 8 == 0, W == 2 * Wo
__global__ __launch_bounds__(256) void dw3v_s2_fwd(const uint16_t* __restrict__ x, const float* __restrict__ w,
                                                   uint16_t* __restrict__ y, int C, int H, int W, int Ho, int Wo,
                                                   long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int W8 = Wo >> 3;
    const int x0 = (int)(idx % W8) * 8;
    const int oy = (int)((idx / W8) % Ho);
    const long plane = idx / ((long)W8 * Ho);
    const float* wc = w + (plane % C) * 9;
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = wc[i];
    const uint16_t* xp = x + plane * (long)H * W;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int iy = 2 * oy - 1 + u;
        if (iy < 0 || iy >= H) continue;
        const uint16_t* row = xp + (long)iy * W + 2 * x0;
        const Bf8 a = ld8(row), b = ld8(row + 8);
        float in[17];
        in[0] = x0 > 0 ? bf16_to_f32(row[-1]) : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { in[j + 1] = a.v[j]; in[j + 9] = b.v[j]; }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            acc[j] = fmaf(k[u * 3 + 2], in[2 * j + 2], fmaf(k[u * 3 + 1], in[2 * j + 1], fmaf(k[u * 3], in[2 * j], acc[j])));
    }
    st8(y + (plane * Ho + oy) * (long)Wo + x0, acc);
}

// stride 2 data gradient: dx[iy][ix] = sum w[u][v] dy[(iy+1-u)/2][(ix+1-v)/2] over even numerators; W % 8 == 0
__global__ __launch_bounds__(256) void dw3v_s2_bwd(const uint16_t* __restrict__ dy, const float* __restrict__ w,
                                                   uint16_t* __restrict__ dx, int C, int H, int W, int Ho, int Wo,
                                                   long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int W8 = W >> 3;
    const int x0 = (int)(idx % W8) * 8;
    const int iy = (int)((idx / W8) % H);
    const long plane = idx / ((long)W8 * H);
    const float* wc = w + (plane % C) * 9;
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = wc[i];
    const uint16_t* dp = dy + plane * (long)Ho * Wo;
    const int ox0 = x0 >> 1;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int ty = iy + 1 - u;
        if (ty < 0 || (ty & 1)) continue;
        const int oy = ty >> 1;
        if (oy >= Ho) continue;
        const uint16_t* row = dp + (long)oy * Wo + ox0;
        const uint2 q = *reinterpret_cast<const uint2*>(row);
        float d[5];
        d[0] = __uint_as_float(q.x << 16); d[1] = __uint_as_float(q.x & 0xffff0000u);
        d[2] = __uint_as_float(q.y << 16); d[3] = __uint_as_float(q.y & 0xffff0000u);
        d[4] = ox0 + 4 < Wo ? bf16_to_f32(row[4]) : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if ((j & 1) == 0) acc[j] = fmaf(k[u * 3 + 1], d[j >> 1], acc[j]);                 // ix even: v = 1
            else acc[j] = fmaf(k[u * 3 + 2], d[(j - 1) >> 1], fmaf(k[u * 3], d[(j + 1) >> 1], acc[j]));   // v = 2, 0
        }
    }
    st8(dx + (plane * H + iy) * (long)W + x0, acc);
}

template <typename T>
int run(bool bwd, const void* a, const float* w, void* o, int N, int C, int H, int W, int stride, void* stream) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2) || (long)N * C > 65535L * 32)
        return PPEA_ERR_UNSUPPORTED;
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const long planes = (long)N * C;
    if (planes > 65535) return PPEA_ERR_UNSUPPORTED;
    if constexpr (sizeof(T) == 2) {                        // bf16: 8 outputs per thread when the rows allow it
        hipStream_t st = (hipStream_t)stream;
        const uint16_t* in = (const uint16_t*)a;
        uint16_t* out = (uint16_t*)o;
        if (stride == 1 && (W & 7) == 0) {
            const long total = planes * H * (W >> 3);
            const unsigned blocks = (unsigned)((total + 255) / 256);
            if (!bwd) hipLaunchKernelGGL(dw3v_s1<false>, dim3(blocks), dim3(256), 0, st, in, w, out, C, H, W, total);
            else hipLaunchKernelGGL(dw3v_s1<true>, dim3(blocks), dim3(256), 0, st, in, w, out, C, H, W, total);
            return launch_status();
        }
        if (stride == 2 && (W & 1) == 0 && (H & 1) == 0 && (Wo & 7) == 0) {
            if (!bwd) {
                const long total = planes * Ho * (Wo >> 3);
                hipLaunchKernelGGL(dw3v_s2_fwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in, w, out, C, H,
                                   W, Ho, Wo, total);
            } else {
                const long total = planes * H * (W >> 3);
                hipLaunchKernelGGL(dw3v_s2_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in, w, out, C, H,
                                   W, Ho, Wo, total);
            }
            return launch_status();
        }
    }
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
