// Plane-sweep matching cost volume, gfx950 (replk_matching_adapter.py:261-340, 372-387, 446-456).
//
// The reference repeats the lookup feature map 96x ([96,128,48,160] = 377 MB per item), warps it
// with grid_sample, subtracts, reduces -- per batch item, in a Python loop.  Here one kernel
// computes, for every (item, depth bin, pixel), the projected sample position analytically and
// reduces |warp(lookup) - cur| over channels on the fly: nothing but the [B,D,h,w] cost leaves
// the chip (10.8 MB/img algorithmic).  A second kernel does the per-pixel
// work over the bin axis (missing -> max, confidence, argmin, lowest-cost depth).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void cost_volume_fwd(const float* __restrict__ cur,
                                                       const float* __restrict__ lookup,
                                                       const float* __restrict__ P,
                                                       const float* __restrict__ inv_K,
                                                       const float* __restrict__ bins,
                                                       const int32_t* __restrict__ skip,
                                                       float* __restrict__ cost, int C, int h, int w, int D,
                                                       float eps) {
    const int b = blockIdx.z, d = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int hw = h * w;
    if (i >= hw) return;
    float* outp = cost + ((long)b * D + d) * hw + i;
    if (skip != nullptr && skip[b] != 0) { *outp = 0.f; return; }
    const int py = i / w, px = i - py * w;
    // the 2-pixel border of the current frame is masked out (rkm.py:315-317)
    if (px < 2 || px >= w - 2 || py < 2 || py >= h - 2) { *outp = 0.f; return; }
    const float* ik = inv_K + b * 16;
    const float* pm = P + b * 12;
    const float fx = (float)px, fy = (float)py;
    const float depth = bins[d];
    float X[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) X[k] = depth * ((ik[k * 4] * fx + ik[k * 4 + 1] * fy) + ik[k * 4 + 2]);
    float cam[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        cam[k] = ((pm[k * 4] * X[0] + pm[k * 4 + 1] * X[1]) + pm[k * 4 + 2] * X[2]) + pm[k * 4 + 3];
    const float iz = cam[2] + eps;
    // normalised grid exactly as Project3D builds it, then both consumers' un-normalisations
    const float gx = ((cam[0] / iz) / (float)(w - 1) - 0.5f) * 2.f;
    const float gy = ((cam[1] / iz) / (float)(h - 1) - 0.5f) * 2.f;
    const float xv = (gx / 2.f + 0.5f) * (float)(w - 1);          // edge-mask coordinates (:306-308)
    const float yv = (gy / 2.f + 0.5f) * (float)(h - 1);
    if (!(xv >= 2.0f && xv <= (float)(w - 2) && yv >= 2.0f && yv <= (float)(h - 2))) { *outp = 0.f; return; }
    const float ix = ((gx + 1.f) / 2.f) * (float)(w - 1);         // grid_sample coordinates
    const float iy = ((gy + 1.f) / 2.f) * (float)(h - 1);
    const float flx = floorf(ix), fly = floorf(iy);
    const int x0 = (int)flx, y0 = (int)fly;
    const float tx = ix - flx, ty = iy - fly;
    const float w00 = (1.f - tx) * (1.f - ty), w01 = tx * (1.f - ty), w10 = (1.f - tx) * ty, w11 = tx * ty;
    // inside the edge mask x0 >= 1 and x0 + 1 <= w - 1 unless ix == w-2 exactly... keep the checks
    const bool in_x1 = (x0 + 1 < w), in_y1 = (y0 + 1 < h);
    const float* lk = lookup + (long)b * C * hw + (long)y0 * w + x0;
    const float* cu = cur + (long)b * C * hw + i;
    double acc = 0.0;
    for (int c = 0; c < C; ++c) {
        const float* l = lk + (long)c * hw;
        const float v00 = l[0];
        const float v01 = in_x1 ? l[1] : 0.f;
        const float v10 = in_y1 ? l[w] : 0.f;
        const float v11 = (in_x1 && in_y1) ? l[w + 1] : 0.f;
        const float warped = ((v00 * w00 + v01 * w01) + v10 * w10) + v11 * w11;
        acc += (double)fabsf(warped - cu[(long)c * hw]);
    }
    const float diff = (float)(acc / (double)C);
    // single lookup frame: volume = diff / ((diff > 0) + 1e-7)   (:323-326)
    *outp = diff / ((diff > 0.f ? 1.f : 0.f) + 1e-7f);
}

__global__ __launch_bounds__(256) void cost_volume_reduce(const float* __restrict__ cost,
                                                          const float* __restrict__ bins,
                                                          float* __restrict__ cost_out,
                                                          float* __restrict__ confidence,
                                                          int64_t* __restrict__ argmin,
                                                          float* __restrict__ lowest, int D, int hw) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hw) return;
    const float* cp = cost + (long)b * D * hw + i;
    float mx = -INFINITY;
    bool all_pos = true;
    for (int d = 0; d < D; ++d) {
        const float v = cp[(long)d * hw];
        mx = fmaxf(mx, v);
        all_pos = all_pos && (v > 0.f);
    }
    const float conf = all_pos ? 1.f : 0.f;
    float best = INFINITY;
    int bi = 0;
    float* op = cost_out + (long)b * D * hw + i;
    for (int d = 0; d < D; ++d) {
        const float v = cp[(long)d * hw];
        const float miss = (v == 0.f) ? 1.f : 0.f;
        const float filled = v * (1.f - miss) + mx * miss;
        const float viz = (filled == 0.f) ? 100.f : filled;
        if (viz < best) { best = viz; bi = d; }
        op[(long)d * hw] = filled * conf;
    }
    confidence[(long)b * hw + i] = conf;
    argmin[(long)b * hw + i] = bi;
    lowest[(long)b * hw + i] = 1.f / bins[bi];
}

}  // namespace

extern "C" {

int ppea_cost_volume_fwd_f32(const float* cur, const float* lookup, const float* P, const float* inv_K,
                             const float* bins, const int32_t* skip, float* cost, int B, int C, int h, int w,
                             int D, float eps, void* stream) {
    if (B < 0 || C <= 0 || h < 5 || w < 5 || D <= 0 || D > 65535) return PPEA_ERR_UNSUPPORTED;
    if (B == 0) return 0;
    dim3 g((h * w + 255) / 256, D, B);
    hipLaunchKernelGGL(cost_volume_fwd, g, dim3(256), 0, (hipStream_t)stream, cur, lookup, P, inv_K, bins,
                       skip, cost, C, h, w, D, eps);
    return launch_status();
}

int ppea_cost_volume_reduce_f32(const float* cost, const float* bins, float* cost_out, float* confidence,
                                int64_t* argmin, float* lowest, int B, int D, int h, int w, void* stream) {
    if (B < 0 || D <= 0 || h <= 0 || w <= 0) return PPEA_ERR_UNSUPPORTED;
    if (B == 0) return 0;
    dim3 g((h * w + 255) / 256, B);
    hipLaunchKernelGGL(cost_volume_reduce, g, dim3(256), 0, (hipStream_t)stream, cost, bins, cost_out,
                       confidence, argmin, lowest, D, h * w);
    return launch_status();
}

}  // extern "C"
