// Geometry of the photometric warp, gfx950:
//   A18+A19  BackprojectDepth -> Project3D fused        (layers.py:138-199)
//   A20      F.grid_sample bilinear, align_corners=True  (trainer.py:911-914, rkm.py:299)
// All HBM-bound streaming kernels: one thread per output pixel, coalesced NCHW rows,
// no intermediate [B,4,HW] point cloud (the reference writes ~5 MB/img for 1.5 MB
// of algorithmic traffic; here depth is read once and the grid written once).
#include "common.h"

namespace {

struct Cam {            // per-batch matrices broadcast through SGPRs
    float ik[9];        // inv_K[:3,:3]
    float p[12];        // P = (K @ T)[:3,:4]
};

__device__ __forceinline__ Cam load_cam(const float* __restrict__ inv_K, const float* __restrict__ P, int b) {
    Cam c;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) c.ik[i * 3 + j] = inv_K[b * 16 + i * 4 + j];
#pragma unroll
    for (int i = 0; i < 12; ++i) c.p[i] = P[b * 12 + i];
    return c;
}

// ray = inv_K[:3,:3] @ (x, y, 1) with the reference's matmul association (k = 0,1,2).
__device__ __forceinline__ void pixel_ray(const Cam& c, float x, float y, float (&r)[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) r[i] = (c.ik[i * 3] * x + c.ik[i * 3 + 1] * y) + c.ik[i * 3 + 2];
}

__global__ __launch_bounds__(256) void backproject_project_fwd(const float* __restrict__ depth,
                                                               const float* __restrict__ inv_K,
                                                               const float* __restrict__ P,
                                                               float* __restrict__ grid, int H, int W,
                                                               float eps) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const Cam c = load_cam(inv_K, P, b);
    const int py = i / W, px = i - py * W;
    float r[3];
    pixel_ray(c, (float)px, (float)py, r);
    const float d = depth[(long)b * H * W + i];
    const float X = d * r[0], Y = d * r[1], Z = d * r[2];
    float cam[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        cam[k] = ((c.p[k * 4] * X + c.p[k * 4 + 1] * Y) + c.p[k * 4 + 2] * Z) + c.p[k * 4 + 3];
    const float iz = cam[2] + eps;
    float u = cam[0] / iz, v = cam[1] / iz;
    u = u / (float)(W - 1);
    v = v / (float)(H - 1);
    float2 o;
    o.x = (u - 0.5f) * 2.f;
    o.y = (v - 0.5f) * 2.f;
    reinterpret_cast<float2*>(grid)[(long)b * H * W + i] = o;
}

// d_depth (per pixel) and dP (12 sums per batch item: wave shuffle reduce -> LDS -> one partial per block and entry in
// `partial` [B][blocks][12]; bp_reduce_dP adds them in a fixed order.  Round 1 added the block sums with float atomics:
// the pose gradient -- and, after a few optimizer steps, the whole pose branch -- then depended on the order in which
// the 480 blocks of an image happened to finish).
__global__ __launch_bounds__(256) void backproject_project_bwd(const float* __restrict__ depth,
                                                               const float* __restrict__ inv_K,
                                                               const float* __restrict__ P,
                                                               const float* __restrict__ d_grid,
                                                               float* __restrict__ d_depth,
                                                               float* __restrict__ partial, int H, int W,
                                                               float eps) {
    __shared__ float red[4][12];
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const Cam c = load_cam(inv_K, P, b);
    float g[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) g[k] = 0.f;
    if (i < H * W) {
        const int py = i / W, px = i - py * W;
        float r[3];
        pixel_ray(c, (float)px, (float)py, r);
        const float d = depth[(long)b * H * W + i];
        const float Xh[4] = {d * r[0], d * r[1], d * r[2], 1.f};
        float cam[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            cam[k] = ((c.p[k * 4] * Xh[0] + c.p[k * 4 + 1] * Xh[1]) + c.p[k * 4 + 2] * Xh[2]) + c.p[k * 4 + 3];
        const float iz = 1.f / (cam[2] + eps);
        const float u = cam[0] * iz, v = cam[1] * iz;
        const float2 dg = reinterpret_cast<const float2*>(d_grid)[(long)b * H * W + i];
        const float du = dg.x * 2.f / (float)(W - 1), dv = dg.y * 2.f / (float)(H - 1);
        const float dc[3] = {du * iz, dv * iz, -(du * u + dv * v) * iz};
        float dX[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) dX[j] = c.p[j] * dc[0] + c.p[4 + j] * dc[1] + c.p[8 + j] * dc[2];
        d_depth[(long)b * H * W + i] = dX[0] * r[0] + dX[1] * r[1] + dX[2] * r[2];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) g[k * 4 + j] = dc[k] * Xh[j];
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const float s = wave_sum(g[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        const float s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        partial[((long)b * gridDim.x + blockIdx.x) * 12 + threadIdx.x] = s;
    }
}

// dP[b][k] = sum over the image's blocks, always in the same order (one wave per batch item, fixed tree)
__global__ __launch_bounds__(64) void bp_reduce_dP(const float* __restrict__ partial, float* __restrict__ dP, int blocks) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    for (int i = lane; i < blocks; i += 64) {
        const float* p = partial + ((long)b * blocks + i) * 12;
#pragma unroll
        for (int k = 0; k < 12; ++k) acc[k] += p[k];
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const float s = wave_sum(acc[k]);
        if (lane == 0) dP[b * 12 + k] = s;
    }
}

// ---- grid_sample -----------------------------------------------------------------------
struct Tap {
    int x0, y0;
    float tx, ty;       // fractional parts
    float mx, my;       // d(ix)/d(grid.x), d(iy)/d(grid.y) incl. the border clip gate
};

__device__ __forceinline__ Tap make_tap(float gx, float gy, int Wi, int Hi, bool border) {
    // align_corners=True un-normalisation: ix = (x + 1) / 2 * (W - 1)
    float ix = ((gx + 1.f) / 2.f) * (float)(Wi - 1);
    float iy = ((gy + 1.f) / 2.f) * (float)(Hi - 1);
    Tap t;
    t.mx = (float)(Wi - 1) / 2.f;
    t.my = (float)(Hi - 1) / 2.f;
    if (border) {       // clip_coordinates_set_grad: zero gradient at and beyond the rim
        if (!(ix > 0.f)) { ix = 0.f; t.mx = 0.f; } else if (ix >= (float)(Wi - 1)) { ix = (float)(Wi - 1); t.mx = 0.f; }
        if (!(iy > 0.f)) { iy = 0.f; t.my = 0.f; } else if (iy >= (float)(Hi - 1)) { iy = (float)(Hi - 1); t.my = 0.f; }
    }
    const float fx = floorf(ix), fy = floorf(iy);
    t.tx = ix - fx;
    t.ty = iy - fy;
    // far-out-of-range coordinates (zeros mode): keep the int conversion defined; all four
    // corners are then outside the image and contribute 0.
    t.x0 = (int)fminf(fmaxf(fx, -4.f), (float)Wi + 4.f);
    t.y0 = (int)fminf(fmaxf(fy, -4.f), (float)Hi + 4.f);
    return t;
}

__device__ __forceinline__ float fetch(const float* __restrict__ plane, int x, int y, int Wi, int Hi) {
    return (x >= 0 && x < Wi && y >= 0 && y < Hi) ? plane[(long)y * Wi + x] : 0.f;
}

template <int CMAX>
__global__ __launch_bounds__(256) void grid_sample_fwd(const float* __restrict__ src,
                                                       const float* __restrict__ grid,
                                                       float* __restrict__ out, int C, int Hi, int Wi,
                                                       int Ho, int Wo, int border) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Ho * Wo) return;
    const float2 g = reinterpret_cast<const float2*>(grid)[(long)b * Ho * Wo + i];
    const Tap t = make_tap(g.x, g.y, Wi, Hi, border != 0);
    // weights exactly as ATen: nw = (x1 - ix)(y1 - iy) etc.
    const float wx1 = t.tx, wx0 = 1.f - t.tx, wy1 = t.ty, wy0 = 1.f - t.ty;
    for (int c = 0; c < C; ++c) {
        const float* plane = src + ((long)b * C + c) * Hi * Wi;
        const float v00 = fetch(plane, t.x0, t.y0, Wi, Hi), v01 = fetch(plane, t.x0 + 1, t.y0, Wi, Hi);
        const float v10 = fetch(plane, t.x0, t.y0 + 1, Wi, Hi), v11 = fetch(plane, t.x0 + 1, t.y0 + 1, Wi, Hi);
        out[((long)b * C + c) * Ho * Wo + i] = ((v00 * (wx0 * wy0) + v01 * (wx1 * wy0)) + v10 * (wx0 * wy1)) + v11 * (wx1 * wy1);
    }
}

__global__ __launch_bounds__(256) void grid_sample_bwd_grid(const float* __restrict__ src,
                                                            const float* __restrict__ grid,
                                                            const float* __restrict__ d_out,
                                                            float* __restrict__ d_grid, int C, int Hi,
                                                            int Wi, int Ho, int Wo, int border) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Ho * Wo) return;
    const float2 g = reinterpret_cast<const float2*>(grid)[(long)b * Ho * Wo + i];
    const Tap t = make_tap(g.x, g.y, Wi, Hi, border != 0);
    float gx = 0.f, gy = 0.f;
    for (int c = 0; c < C; ++c) {
        const float* plane = src + ((long)b * C + c) * Hi * Wi;
        const float v00 = fetch(plane, t.x0, t.y0, Wi, Hi), v01 = fetch(plane, t.x0 + 1, t.y0, Wi, Hi);
        const float v10 = fetch(plane, t.x0, t.y0 + 1, Wi, Hi), v11 = fetch(plane, t.x0 + 1, t.y0 + 1, Wi, Hi);
        const float go = d_out[((long)b * C + c) * Ho * Wo + i];
        gx += go * ((v01 - v00) * (1.f - t.ty) + (v11 - v10) * t.ty);
        gy += go * ((v10 - v00) * (1.f - t.tx) + (v11 - v01) * t.tx);
    }
    float2 o;
    o.x = gx * t.mx;
    o.y = gy * t.my;
    reinterpret_cast<float2*>(d_grid)[(long)b * Ho * Wo + i] = o;
}


// ---- A15: axis-angle + translation -> 4x4 transformation (layers.py:26-42, 61-100) --------------------------------------
// One thread per sample.  The reference builds the matrix from ~25 tiny element-wise kernels (and autograd adds ~50 more on
// the way back) on the step's critical stream; here the forward is one launch and the backward one launch whose Jacobian
// comes from running the SAME arithmetic on dual numbers (value + 6 partial derivatives): exact, no hand-derived formulas.
struct Dual6 {
    float v, d[6];
};
__device__ __forceinline__ Dual6 dconst(float c) { Dual6 r; r.v = c; for (int i = 0; i < 6; ++i) r.d[i] = 0.f; return r; }
__device__ __forceinline__ Dual6 dvar(float c, int k) { Dual6 r = dconst(c); r.d[k] = 1.f; return r; }
__device__ __forceinline__ Dual6 operator+(const Dual6& a, const Dual6& b) { Dual6 r; r.v = a.v + b.v; for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ __forceinline__ Dual6 operator-(const Dual6& a, const Dual6& b) { Dual6 r; r.v = a.v - b.v; for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ __forceinline__ Dual6 operator*(const Dual6& a, const Dual6& b) { Dual6 r; r.v = a.v * b.v; for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ __forceinline__ Dual6 operator/(const Dual6& a, const Dual6& b) {
    Dual6 r; r.v = a.v / b.v;
    for (int i = 0; i < 6; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v;
    return r;
}
__device__ __forceinline__ Dual6 dsqrt(const Dual6& a) { Dual6 r; r.v = sqrtf(a.v); for (int i = 0; i < 6; ++i) r.d[i] = a.v > 0.f ? a.d[i] / (2.f * r.v) : 0.f; return r; }
__device__ __forceinline__ Dual6 dsin(const Dual6& a) { Dual6 r; r.v = sinf(a.v); const float c = cosf(a.v); for (int i = 0; i < 6; ++i) r.d[i] = c * a.d[i]; return r; }
__device__ __forceinline__ Dual6 dcos(const Dual6& a) { Dual6 r; r.v = cosf(a.v); const float s = -sinf(a.v); for (int i = 0; i < 6; ++i) r.d[i] = s * a.d[i]; return r; }

// T[16] row-major from (axis-angle vx, vy, vz, translation tx, ty, tz); `N` is float (forward) or Dual6 (backward)
template <typename N>
__device__ __forceinline__ void pose_matrix(const N& vx, const N& vy, const N& vz, const N& tx, const N& ty, const N& tz, int invert,
                                            N (&T)[16], N (*cst)(float), N (*sq)(const N&), N (*sn)(const N&), N (*cs)(const N&)) {
    const N angle = sq(vx * vx + vy * vy + vz * vz);
    const N den = angle + cst(1e-7f);
    const N x = vx / den, y = vy / den, z = vz / den;
    const N ca = cs(angle), sa = sn(angle);
    const N C = cst(1.f) - ca;
    const N xC = x * C, yC = y * C, zC = z * C;
    N R[9] = {x * xC + ca, x * yC - z * sa, z * xC + y * sa,
              x * yC + z * sa, y * yC + ca, y * zC - x * sa,
              z * xC - y * sa, y * zC + x * sa, z * zC + ca};
    const N zero = cst(0.f), one = cst(1.f);
    if (!invert) {                                           // T(t) . R  = [R | t]
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) T[4 * i + j] = R[3 * i + j];
        }
        T[3] = tx; T[7] = ty; T[11] = tz;
    } else {                                                 // R^T . T(-t) = [R^T | -R^T t]
        const N nx = zero - tx, ny = zero - ty, nz = zero - tz;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) T[4 * i + j] = R[3 * j + i];
            T[4 * i + 3] = R[i] * nx + R[3 + i] * ny + R[6 + i] * nz;
        }
    }
    T[12] = zero; T[13] = zero; T[14] = zero; T[15] = one;
}
__device__ __forceinline__ float fconst(float c) { return c; }
__device__ __forceinline__ float fsqrt(const float& a) { return sqrtf(a); }
__device__ __forceinline__ float fsin(const float& a) { return sinf(a); }
__device__ __forceinline__ float fcos(const float& a) { return cosf(a); }

__global__ void pose_matrix_fwd(const float* __restrict__ aa, const float* __restrict__ tr, float* __restrict__ T, int B, int invert) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float M[16];
    pose_matrix<float>(aa[3 * b], aa[3 * b + 1], aa[3 * b + 2], tr[3 * b], tr[3 * b + 1], tr[3 * b + 2], invert, M, fconst, fsqrt, fsin, fcos);
    for (int i = 0; i < 16; ++i) T[16 * b + i] = M[i];
}
__global__ void pose_matrix_bwd(const float* __restrict__ aa, const float* __restrict__ tr, const float* __restrict__ dT,
                                float* __restrict__ daa, float* __restrict__ dtr, int B, int invert) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    Dual6 M[16];
    pose_matrix<Dual6>(dvar(aa[3 * b], 0), dvar(aa[3 * b + 1], 1), dvar(aa[3 * b + 2], 2), dvar(tr[3 * b], 3), dvar(tr[3 * b + 1], 4),
                       dvar(tr[3 * b + 2], 5), invert, M, dconst, dsqrt, dsin, dcos);
    float g[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 16; ++i) {
        const float w = dT[16 * b + i];
        for (int k = 0; k < 6; ++k) g[k] += w * M[i].d[k];
    }
    for (int k = 0; k < 3; ++k) { daa[3 * b + k] = g[k]; dtr[3 * b + k] = g[3 + k]; }
}

}  // namespace

extern "C" {

int ppea_backproject_project_fwd_f32(const float* depth, const float* inv_K, const float* P, float* grid,
                                     int B, int H, int W, float eps, void* stream) {
    if (B < 0 || H < 2 || W < 2) return PPEA_ERR_UNSUPPORTED;
    if (B == 0) return 0;
    dim3 g((H * W + 255) / 256, B);
    hipLaunchKernelGGL(backproject_project_fwd, g, dim3(256), 0, (hipStream_t)stream, depth, inv_K, P, grid,
                       H, W, eps);
    return launch_status();
}

long ppea_backproject_project_bwd_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H < 2 || W < 2) return 0;
    return (long)B * ((H * W + 255) / 256) * 12 * (long)sizeof(float);
}

int ppea_backproject_project_bwd_f32(const float* depth, const float* inv_K, const float* P,
                                     const float* d_grid, float* d_depth, float* dP, void* workspace, int B, int H,
                                     int W, float eps, void* stream) {
    if (B < 0 || H < 2 || W < 2) return PPEA_ERR_UNSUPPORTED;
    if (B == 0) return 0;
    if (workspace == nullptr) return PPEA_ERR_ARG;
    dim3 g((H * W + 255) / 256, B);
    hipLaunchKernelGGL(backproject_project_bwd, g, dim3(256), 0, (hipStream_t)stream, depth, inv_K, P,
                       d_grid, d_depth, (float*)workspace, H, W, eps);
    hipLaunchKernelGGL(bp_reduce_dP, dim3(B), dim3(64), 0, (hipStream_t)stream, (const float*)workspace, dP, (int)g.x);
    return launch_status();
}

int ppea_grid_sample_fwd_f32(const float* src, const float* grid, float* out, int B, int C, int Hi, int Wi,
                             int Ho, int Wo, int padding, void* stream) {
    if (B < 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || (padding != 0 && padding != 1))
        return PPEA_ERR_UNSUPPORTED;
    if (B == 0) return 0;
    dim3 g((Ho * Wo + 255) / 256, B);
    hipLaunchKernelGGL(grid_sample_fwd<0>, g, dim3(256), 0, (hipStream_t)stream, src, grid, out, C, Hi, Wi,
                       Ho, Wo, padding);
    return launch_status();
}

int ppea_grid_sample_bwd_grid_f32(const float* src, const float* grid, const float* d_out, float* d_grid,
                                  int B, int C, int Hi, int Wi, int Ho, int Wo, int padding, void* stream) {
    if (B < 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || (padding != 0 && padding != 1))
        return PPEA_ERR_UNSUPPORTED;
    if (B == 0) return 0;
    dim3 g((Ho * Wo + 255) / 256, B);
    hipLaunchKernelGGL(grid_sample_bwd_grid, g, dim3(256), 0, (hipStream_t)stream, src, grid, d_out, d_grid,
                       C, Hi, Wi, Ho, Wo, padding);
    return launch_status();
}


// A15 (layers.py:26-42, 61-100): T [B][4][4] fp32 from axis-angle aa [B][3] and translation tr [B][3] (invert: the inverse
// transformation R^T T(-t)); backward: d aa, d tr from dT (exact Jacobian of the same arithmetic).
int ppea_pose_matrix_fwd_f32(const float* aa, const float* tr, float* T, int B, int invert, void* stream) {
    if (B < 0) return PPEA_ERR_ARG;
    if (B == 0) return 0;
    if (!aa || !tr || !T) return PPEA_ERR_ARG;
    hipLaunchKernelGGL(pose_matrix_fwd, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, aa, tr, T, B, invert);
    return launch_status();
}
int ppea_pose_matrix_bwd_f32(const float* aa, const float* tr, const float* dT, float* daa, float* dtr, int B, int invert,
                             void* stream) {
    if (B < 0) return PPEA_ERR_ARG;
    if (B == 0) return 0;
    if (!aa || !tr || !dT || !daa || !dtr) return PPEA_ERR_ARG;
    hipLaunchKernelGGL(pose_matrix_bwd, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, aa, tr, dT, daa, dtr, B, invert);
    return launch_status();
}

}  // extern "C"
