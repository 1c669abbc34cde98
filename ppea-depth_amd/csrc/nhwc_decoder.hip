// channels_last versions of the two element-wise passes around the decoders' 3x3 convolutions (layers.py:103-135:
// ReflectionPad2d(1) -> conv -> (+bias) -> ELU), so that the decoders can hand the library's NHWC-native implicit-GEMM
// kernels NHWC activations (no layout round trip per convolution).  A thread moves 8 consecutive channels (16-byte
// accesses for bf16); C % 8 == 0.
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

__device__ __forceinline__ int refl(int i, int n) {       // padded index -1..n  ->  source index
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

// out[b][Y][X][:] = in[b][refl(Y-1)][refl(X-1)][:]        (a pure 16-byte copy per thread)
template <typename T>
__global__ __launch_bounds__(TPB) void nhwc_pad_fwd(const T* __restrict__ in, T* __restrict__ out, int H, int W, int C,
                                                    unsigned total_vec) {
    const unsigned v = blockIdx.x * TPB + threadIdx.x;
    if (v >= total_vec) return;
    const unsigned CT = (unsigned)C / V, Wo = W + 2, Ho = H + 2;
    const unsigned pix = v / CT, c0 = (v - pix * CT) * V;
    const unsigned row = pix / Wo, X = pix - row * Wo;
    const unsigned b = row / Ho, Y = row - b * Ho;
    const long src = (((long)b * H + refl((int)Y - 1, H)) * W + refl((int)X - 1, W)) * C + c0;
    if (sizeof(T) == 2) *reinterpret_cast<uint4*>(out + (long)v * V) = *reinterpret_cast<const uint4*>(in + src);
    else {
        reinterpret_cast<uint4*>(out + (long)v * V)[0] = reinterpret_cast<const uint4*>(in + src)[0];
        reinterpret_cast<uint4*>(out + (long)v * V)[1] = reinterpret_cast<const uint4*>(in + src)[1];
    }
}

// din[b][y][x][:] = sum of the padded positions that mirror onto (y, x): (y+1, x+1) and the border copies
template <typename T>
__global__ __launch_bounds__(TPB) void nhwc_pad_bwd(const T* __restrict__ dout, T* __restrict__ din, int H, int W,
                                                    int C, unsigned total_vec) {
    const unsigned v = blockIdx.x * TPB + threadIdx.x;
    if (v >= total_vec) return;
    const unsigned CT = (unsigned)C / V, Wo = W + 2, Ho = H + 2;
    const unsigned pix = v / CT, c0 = (v - pix * CT) * V;
    const unsigned row = pix / (unsigned)W;
    const int x = (int)(pix - row * (unsigned)W);
    const unsigned b = row / (unsigned)H;
    const int y = (int)(row - b * (unsigned)H);
    int ys[3] = {y + 1, -1, -1}, xs[3] = {x + 1, -1, -1};
    if (y == 1) ys[1] = 0;
    if (y == H - 2) ys[(y == 1) ? 2 : 1] = H + 1;
    if (x == 1) xs[1] = 0;
    if (x == W - 2) xs[(x == 1) ? 2 : 1] = W + 1;
    float acc[V] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const T* base = dout + (long)b * Ho * Wo * C + c0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (ys[i] < 0) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (xs[j] < 0) continue;
            float t[V];
            ld8<T>(base + ((long)ys[i] * Wo + xs[j]) * C, t);
#pragma unroll
            for (int k = 0; k < V; ++k) acc[k] += t[k];
        }
    }
    st8<T>(din + (long)v * V, acc);
}

// y = elu(z + bias[c])
template <typename T>
__global__ __launch_bounds__(TPB) void nhwc_bias_elu_fwd(const T* __restrict__ z, const void* __restrict__ bias,
                                                         int bias_bf16, T* __restrict__ y, int C, unsigned total_vec) {
    const unsigned v = blockIdx.x * TPB + threadIdx.x;
    if (v >= total_vec) return;
    const unsigned CT = (unsigned)C / V, c0 = (v % CT) * V;
    float bv[V], x[V], o[V];
    if (bias_bf16) ld8<uint16_t>(reinterpret_cast<const uint16_t*>(bias) + c0, bv);
    else ld8<float>(reinterpret_cast<const float*>(bias) + c0, bv);
    ld8<T>(z + (long)v * V, x);
#pragma unroll
    for (int k = 0; k < V; ++k) { const float u = x[k] + bv[k]; o[k] = u > 0.f ? u : expm1f(u); }
    st8<T>(y + (long)v * V, o);
}

// dz = dy * elu'(u) (from the output: 1 for y > 0 else y + 1); partial[slab][C] = per-channel sums of dz over the slab
template <typename T>
__global__ __launch_bounds__(TPB) void nhwc_bias_elu_bwd(const T* __restrict__ dy, const T* __restrict__ y,
                                                         T* __restrict__ dz, float* __restrict__ partial, int P, int C,
                                                         int rows_per_slab) {
    extern __shared__ float sh[];                          // [RL][C]
    const int CT = C / V, RL = TPB / CT;
    const int ct = threadIdx.x % CT, rl = threadIdx.x / CT, c0 = ct * V;
    const int r0 = blockIdx.x * rows_per_slab, r1 = min(P, r0 + rows_per_slab);
    float s[V] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = r0 + rl; r < r1; r += RL) {
        float d[V], yy[V], o[V];
        const long off = (long)r * C + c0;
        ld8<T>(dy + off, d);
        ld8<T>(y + off, yy);
#pragma unroll
        for (int k = 0; k < V; ++k) { o[k] = d[k] * (yy[k] > 0.f ? 1.f : yy[k] + 1.f); s[k] += o[k]; }
        st8<T>(dz + off, o);
    }
#pragma unroll
    for (int k = 0; k < V; ++k) sh[rl * C + c0 + k] = s[k];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += TPB) {
        float t = 0.f;
        for (int j = 0; j < RL; ++j) t += sh[j * C + c];
        partial[(long)blockIdx.x * C + c] = t;
    }
}

inline bool ok_c(int C) {
    if (C < V || (C % V) != 0) return false;
    const int ct = C / V;
    return ct <= TPB && (TPB % ct) == 0;
}
inline int plan_slabs(int P, int C, int& rows) {
    const int RL = TPB / (C / V);
    int slabs = (P + RL * 8 - 1) / (RL * 8);
    if (slabs > 1024) slabs = 1024;
    if (slabs < 1) slabs = 1;
    rows = (P + slabs - 1) / slabs;
    return (P + rows - 1) / rows;
}

template <typename T>
int pad_impl(bool bwd, const void* a, void* o, int B, int H, int W, int C, void* stream) {
    if (B <= 0 || C < V || (C % V) != 0 || H < (bwd ? 3 : 2) || W < (bwd ? 3 : 2)) return PPEA_ERR_UNSUPPORTED;
    const long total = bwd ? (long)B * H * W * (C / V) : (long)B * (H + 2) * (W + 2) * (C / V);
    if (total > 0x7fffffffL) return PPEA_ERR_UNSUPPORTED;
    const unsigned blocks = (unsigned)((total + TPB - 1) / TPB);
    if (!bwd) hipLaunchKernelGGL(nhwc_pad_fwd<T>, dim3(blocks), dim3(TPB), 0, (hipStream_t)stream, (const T*)a, (T*)o, H,
                                 W, C, (unsigned)total);
    else hipLaunchKernelGGL(nhwc_pad_bwd<T>, dim3(blocks), dim3(TPB), 0, (hipStream_t)stream, (const T*)a, (T*)o, H, W, C,
                            (unsigned)total);
    return launch_status();
}

// ---- nearest 2x upsampling fused with the skip concatenation (depth_decoder_v2.py:231-236: upsample(x), then
// torch.cat with the encoder feature) -- one pass instead of two, 16-byte pieces; and its backward: the gradient of the
// concatenated tensor is split, the upsampled part summed over each 2x2 block in fp32 (what upsample_nearest2d_backward
// does) -- one pass instead of slice copies + the library kernel.
//   out[n][y][x][0:C1] = a[n][y/2][x/2][:]      out[n][y][x][C1:C1+C2] = b[n][y][x][:]        (C2 may be 0)
template <typename T>
__global__ __launch_bounds__(TPB) void nhwc_up2cat_fwd(const T* __restrict__ a, const T* __restrict__ b,
                                                       T* __restrict__ out, int H, int W, int C1, int C2, unsigned total_vec) {
    const unsigned v = blockIdx.x * TPB + threadIdx.x;
    if (v >= total_vec) return;
    const unsigned CT = (unsigned)(C1 + C2) / V;
    const unsigned pix = v / CT, c0 = (v - pix * CT) * V;
    const unsigned row = pix / (unsigned)W, x = pix - row * (unsigned)W;
    const unsigned n = row / (unsigned)H, y = row - n * (unsigned)H;
    const T* src = (c0 < (unsigned)C1)
                       ? a + ((long)((n * (unsigned)(H >> 1) + (y >> 1)) * (unsigned)(W >> 1) + (x >> 1))) * C1 + c0
                       : b + (long)pix * C2 + (c0 - (unsigned)C1);
    T* dst = out + (long)v * V;
    reinterpret_cast<uint4*>(dst)[0] = reinterpret_cast<const uint4*>(src)[0];
    if (sizeof(T) == 4) reinterpret_cast<uint4*>(dst)[1] = reinterpret_cast<const uint4*>(src)[1];
}

template <typename T>
__global__ __launch_bounds__(TPB) void nhwc_up2cat_bwd(const T* __restrict__ dout, T* __restrict__ da, T* __restrict__ db,
                                                       int H, int W, int C1, int C2, unsigned total_a, unsigned total_vec) {
    const unsigned v = blockIdx.x * TPB + threadIdx.x;
    if (v >= total_vec) return;
    const int C = C1 + C2;
    if (v < total_a) {
        const unsigned CT = (unsigned)C1 / V, W2 = (unsigned)W >> 1, H2 = (unsigned)H >> 1;
        const unsigned pix = v / CT, c0 = (v - pix * CT) * V;
        const unsigned row = pix / W2, x2 = pix - row * W2;
        const unsigned n = row / H2, y2 = row - n * H2;
        const T* p = dout + ((long)((n * (unsigned)H + 2 * y2) * (unsigned)W + 2 * x2)) * C + c0;
        float s[V], t[V];
        ld8<T>(p, s);
        ld8<T>(p + C, t);
#pragma unroll
        for (int k = 0; k < V; ++k) s[k] += t[k];
        ld8<T>(p + (long)W * C, t);
#pragma unroll
        for (int k = 0; k < V; ++k) s[k] += t[k];
        ld8<T>(p + (long)W * C + C, t);
#pragma unroll
        for (int k = 0; k < V; ++k) s[k] += t[k];
        st8<T>(da + (long)v * V, s);
    } else {
        const unsigned u = v - total_a, CT = (unsigned)C2 / V;
        const unsigned pix = u / CT, c0 = (u - pix * CT) * V;
        const T* src = dout + (long)pix * C + C1 + c0;
        T* dst = db + (long)u * V;
        reinterpret_cast<uint4*>(dst)[0] = reinterpret_cast<const uint4*>(src)[0];
        if (sizeof(T) == 4) reinterpret_cast<uint4*>(dst)[1] = reinterpret_cast<const uint4*>(src)[1];
    }
}

template <typename T>
int up2cat_impl(bool bwd, const void* p0, const void* p1, void* p2, int N, int H, int W, int C1, int C2, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C1 <= 0 || (C1 % V) != 0 || C2 < 0 || (C2 % V) != 0)
        return PPEA_ERR_UNSUPPORTED;
    const long pixels = (long)N * H * W;
    if (pixels * (C1 + C2) > 0x7fffffffL) return PPEA_ERR_UNSUPPORTED;           // 32-bit piece indices
    if (!bwd) {
        if (C2 > 0 && p1 == nullptr) return PPEA_ERR_ARG;
        const unsigned total = (unsigned)(pixels * (C1 + C2) / V);
        hipLaunchKernelGGL(nhwc_up2cat_fwd<T>, dim3((total + TPB - 1) / TPB), dim3(TPB), 0, (hipStream_t)stream,
                           (const T*)p0, (const T*)p1, (T*)p2, H, W, C1, C2, total);
    } else {
        if (C2 > 0 && p2 == nullptr) return PPEA_ERR_ARG;
        const unsigned total_a = (unsigned)(pixels / 4 * C1 / V), total = total_a + (unsigned)(pixels * C2 / V);
        hipLaunchKernelGGL(nhwc_up2cat_bwd<T>, dim3((total + TPB - 1) / TPB), dim3(TPB), 0, (hipStream_t)stream,
                           (const T*)p0, (T*)p1, (T*)p2, H, W, C1, C2, total_a, total);
    }
    return launch_status();
}

}  // namespace

extern "C" {

// x [B][H][W][C] -> out [B][H+2][W+2][C]  (channels_last storage of [B,C,H,W] / [B,C,H+2,W+2])
int ppea_nhwc_reflect_pad1_fwd_f32(const void* x, void* out, int B, int H, int W, int C, void* stream) {
    return pad_impl<float>(false, x, out, B, H, W, C, stream);
}
int ppea_nhwc_reflect_pad1_fwd_bf16(const void* x, void* out, int B, int H, int W, int C, void* stream) {
    return pad_impl<uint16_t>(false, x, out, B, H, W, C, stream);
}
int ppea_nhwc_reflect_pad1_bwd_f32(const void* dout, void* dx, int B, int H, int W, int C, void* stream) {
    return pad_impl<float>(true, dout, dx, B, H, W, C, stream);
}
int ppea_nhwc_reflect_pad1_bwd_bf16(const void* dout, void* dx, int B, int H, int W, int C, void* stream) {
    return pad_impl<uint16_t>(true, dout, dx, B, H, W, C, stream);
}
// slabs of the backward's partial sums for P pixels: partial is [slabs][C] fp32
int ppea_nhwc_bias_elu_slabs(int P, int C) {
    if (P <= 0 || !ok_c(C)) return PPEA_ERR_UNSUPPORTED;
    int rows;
    return plan_slabs(P, C, rows);
}
int ppea_nhwc_bias_elu_fwd_f32(const void* z, const void* bias, int bias_bf16, void* y, int P, int C, void* stream) {
    if (P <= 0 || !ok_c(C) || bias == nullptr || (long)P * C / V > 0x7fffffffL) return PPEA_ERR_UNSUPPORTED;
    const unsigned total = (unsigned)((long)P * C / V);
    hipLaunchKernelGGL(nhwc_bias_elu_fwd<float>, dim3((total + TPB - 1) / TPB), dim3(TPB), 0, (hipStream_t)stream,
                       (const float*)z, bias, bias_bf16, (float*)y, C, total);
    return launch_status();
}
int ppea_nhwc_bias_elu_fwd_bf16(const void* z, const void* bias, int bias_bf16, void* y, int P, int C, void* stream) {
    if (P <= 0 || !ok_c(C) || bias == nullptr || (long)P * C / V > 0x7fffffffL) return PPEA_ERR_UNSUPPORTED;
    const unsigned total = (unsigned)((long)P * C / V);
    hipLaunchKernelGGL(nhwc_bias_elu_fwd<uint16_t>, dim3((total + TPB - 1) / TPB), dim3(TPB), 0, (hipStream_t)stream,
                       (const uint16_t*)z, bias, bias_bf16, (uint16_t*)y, C, total);
    return launch_status();
}
int ppea_nhwc_bias_elu_bwd_f32(const void* dy, const void* y, void* dz, float* partial, int P, int C, void* stream) {
    if (P <= 0 || !ok_c(C)) return PPEA_ERR_UNSUPPORTED;
    int rows;
    const int slabs = plan_slabs(P, C, rows);
    hipLaunchKernelGGL(nhwc_bias_elu_bwd<float>, dim3(slabs), dim3(TPB), (size_t)(TPB / (C / V)) * C * sizeof(float),
                       (hipStream_t)stream, (const float*)dy, (const float*)y, (float*)dz, partial, P, C, rows);
    return launch_status();
}
int ppea_nhwc_bias_elu_bwd_bf16(const void* dy, const void* y, void* dz, float* partial, int P, int C, void* stream) {
    if (P <= 0 || !ok_c(C)) return PPEA_ERR_UNSUPPORTED;
    int rows;
    const int slabs = plan_slabs(P, C, rows);
    hipLaunchKernelGGL(nhwc_bias_elu_bwd<uint16_t>, dim3(slabs), dim3(TPB), (size_t)(TPB / (C / V)) * C * sizeof(float),
                       (hipStream_t)stream, (const uint16_t*)dy, (const uint16_t*)y, (uint16_t*)dz, partial, P, C, rows);
    return launch_status();
}

// a [N][H/2][W/2][C1], b [N][H][W][C2] (or NULL with C2 = 0) -> out [N][H][W][C1+C2]: nearest 2x upsampling of a next to b.
// H, W: the OUTPUT size (even); C1, C2 multiples of 8.
int ppea_nhwc_up2cat_fwd_f32(const void* a, const void* b, void* out, int N, int H, int W, int C1, int C2, void* stream) {
    return up2cat_impl<float>(false, a, b, out, N, H, W, C1, C2, stream);
}
int ppea_nhwc_up2cat_fwd_bf16(const void* a, const void* b, void* out, int N, int H, int W, int C1, int C2, void* stream) {
    return up2cat_impl<uint16_t>(false, a, b, out, N, H, W, C1, C2, stream);
}
// dout [N][H][W][C1+C2] -> da [N][H/2][W/2][C1] (2x2 block sums, fp32 accumulation), db [N][H][W][C2]
int ppea_nhwc_up2cat_bwd_f32(const void* dout, void* da, void* db, int N, int H, int W, int C1, int C2, void* stream) {
    return up2cat_impl<float>(true, dout, da, db, N, H, W, C1, C2, stream);
}
int ppea_nhwc_up2cat_bwd_bf16(const void* dout, void* da, void* db, int N, int H, int W, int C1, int C2, void* stream) {
    return up2cat_impl<uint16_t>(true, dout, da, db, N, H, W, C1, C2, stream);
}

}  // extern "C"
