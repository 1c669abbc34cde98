// Plane-sweep matching cost volume, gfx950 (replk_matching_adapter.py:261-340, 372-387, 446-456).
//
// The reference repeats the lookup feature map 96x ([96,128,48,160] = 377 MB per item), warps it
// with grid_sample, subtracts, reduces -- per batch item, in a Python loop.  Here one kernel
// computes, for every (item, depth bin, pixel), the projected sample position analytically and
// reduces |warp(lookup) - cur| over channels on the fly: nothing but the [B,D,h,w] cost leaves
// the chip (10.8 MB/img algorithmic).  A second kernel does the per-pixel
// work over the bin axis (missing -> max, confidence, argmin, lowest-cost depth).
#include "common.h"
#include <cstdlib>

namespace {

// Two horizontally adjacent samples of a feature row in one 8-byte load: the address is only 4-byte aligned (x0 is any
// column), which gfx950 global loads allow.
struct alignas(4) Pair { float a, b; };

// One thread = one pixel x DB consecutive depth bins.  The kernel is bound by vector-memory instruction issue (gathers
// that hit L1 / L2: the lookup map is 3.9 MB per item), not by bytes, so the loads are what is economised: the current
// frame's feature is read once per channel for all DB bins (it does not depend on the bin), and a bilinear corner pair is
// one 8-byte load -- (1 + 2 * 2 * DB) / DB = 2.25 loads per (pixel, bin, channel) at DB = 4 instead of 5.
// Same arithmetic, in the same order, as the one-bin form (pinned bit for bit by test_cost_volume_golden's argmin).
template <int DB>
__global__ __launch_bounds__(256) void cost_volume_fwd(const float* __restrict__ cur,
                                                       const float* __restrict__ lookup,
                                                       const float* __restrict__ P,
                                                       const float* __restrict__ inv_K,
                                                       const float* __restrict__ bins,
                                                       const int32_t* __restrict__ skip,
                                                       float* __restrict__ cost, int C, int h, int w, int D,
                                                       float eps) {
    const int b = blockIdx.z, d0 = blockIdx.y * DB;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int hw = h * w;
    if (i >= hw) return;
    float* outp = cost + ((long)b * D + d0) * hw + i;
    const int nd = min(DB, D - d0);
    const int py = i / w, px = i - py * w;
    // the 2-pixel border of the current frame is masked out (rkm.py:315-317)
    if ((skip != nullptr && skip[b] != 0) || px < 2 || px >= w - 2 || py < 2 || py >= h - 2) {
        for (int k = 0; k < nd; ++k) outp[(long)k * hw] = 0.f;
        return;
    }
    const float* ik = inv_K + b * 16;
    const float* pm = P + b * 12;
    const float fx = (float)px, fy = (float)py;
    float ray[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) ray[k] = (ik[k * 4] * fx + ik[k * 4 + 1] * fy) + ik[k * 4 + 2];
    long off[DB];                      // y0 * w + x0 of the top-left corner; -1: outside the edge mask (cost 0)
    float w00[DB], w01[DB], w10[DB], w11[DB];
    bool pair_ok[DB];
#pragma unroll
    for (int k = 0; k < DB; ++k) {
        off[k] = -1;
        pair_ok[k] = false;
        w00[k] = w01[k] = w10[k] = w11[k] = 0.f;
        if (k >= nd) continue;
        const float depth = bins[d0 + k];
        float X[3], cam[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) X[q] = depth * ray[q];
#pragma unroll
        for (int q = 0; q < 3; ++q)
            cam[q] = ((pm[q * 4] * X[0] + pm[q * 4 + 1] * X[1]) + pm[q * 4 + 2] * X[2]) + pm[q * 4 + 3];
        const float iz = cam[2] + eps;
        // normalised grid exactly as Project3D builds it, then both consumers' un-normalisations
        const float gx = ((cam[0] / iz) / (float)(w - 1) - 0.5f) * 2.f;
        const float gy = ((cam[1] / iz) / (float)(h - 1) - 0.5f) * 2.f;
        const float xv = (gx / 2.f + 0.5f) * (float)(w - 1);          // edge-mask coordinates (:306-308)
        const float yv = (gy / 2.f + 0.5f) * (float)(h - 1);
        if (!(xv >= 2.0f && xv <= (float)(w - 2) && yv >= 2.0f && yv <= (float)(h - 2))) continue;
        const float ix = ((gx + 1.f) / 2.f) * (float)(w - 1);         // grid_sample coordinates
        const float iy = ((gy + 1.f) / 2.f) * (float)(h - 1);
        const float flx = floorf(ix), fly = floorf(iy);
        const int x0 = (int)flx, y0 = (int)fly;
        const float tx = ix - flx, ty = iy - fly;
        w00[k] = (1.f - tx) * (1.f - ty); w01[k] = tx * (1.f - ty); w10[k] = (1.f - tx) * ty; w11[k] = tx * ty;
        // inside the edge mask the 2 x 2 footprint lies inside the map; a footprint that touches the last column / row
        // (ix == w - 2 rounded up) takes the corner-by-corner path with zero fill, as grid_sample does
        pair_ok[k] = (x0 + 1 < w) && (y0 + 1 < h);
        off[k] = (long)y0 * w + x0;
        if (!pair_ok[k]) {              // rare: resolve here, once, with scalar loads
            const float* lk = lookup + (long)b * C * hw + off[k];
            const float* cu = cur + (long)b * C * hw + i;
            double acc = 0.0;
            for (int c = 0; c < C; ++c) {
                const float* l = lk + (long)c * hw;
                const float v00 = l[0];
                const float v01 = (x0 + 1 < w) ? l[1] : 0.f;
                const float v10 = (y0 + 1 < h) ? l[w] : 0.f;
                const float v11 = 0.f;
                const float warped = ((v00 * w00[k] + v01 * w01[k]) + v10 * w10[k]) + v11 * w11[k];
                acc += (double)fabsf(warped - cu[(long)c * hw]);
            }
            const float diff = (float)(acc / (double)C);
            outp[(long)k * hw] = diff / ((diff > 0.f ? 1.f : 0.f) + 1e-7f);
            off[k] = -2;                // done
        }
    }
    const float* lkb = lookup + (long)b * C * hw;
    const float* cu = cur + (long)b * C * hw + i;
    double acc[DB];
#pragma unroll
    for (int k = 0; k < DB; ++k) acc[k] = 0.0;
    for (int c = 0; c < C; ++c) {
        const float cv = cu[(long)c * hw];
        const float* lc = lkb + (long)c * hw;
#pragma unroll
        for (int k = 0; k < DB; ++k) {
            if (off[k] < 0) continue;
            const Pair top = *reinterpret_cast<const Pair*>(lc + off[k]);
            const Pair bot = *reinterpret_cast<const Pair*>(lc + off[k] + w);
            const float warped = ((top.a * w00[k] + top.b * w01[k]) + bot.a * w10[k]) + bot.b * w11[k];
            acc[k] += (double)fabsf(warped - cv);
        }
    }
#pragma unroll
    for (int k = 0; k < DB; ++k) {
        if (k >= nd || off[k] == -2) continue;
        float v = 0.f;
        if (off[k] >= 0) {
            const float diff = (float)(acc[k] / (double)C);
            // single lookup frame: volume = diff / ((diff > 0) + 1e-7)   (:323-326)
            v = diff / ((diff > 0.f ? 1.f : 0.f) + 1e-7f);
        }
        outp[(long)k * hw] = v;
    }
}

// ---- bf16 features (the bf16 step) ----------------------------------------------------------------------------------------
// The forward kernel is bound by the bytes that cross the L1 (64 B / clk / CU: 4 bilinear corners of 4 bytes per pixel, bin
// and channel; the neighbouring lanes' footprints overlap, the cache lines are fetched once but delivered per lane).  The
// bf16 step's features ARE bf16: packed as channel PAIRS (one dword = channels 2c, 2c + 1 at one position) a corner load
// serves two channels -- half the L1 bytes per (pixel, bin, channel).  The arithmetic is the fp32 kernel's on the widened
// values, channel by channel in the same order: bit-identical to running the fp32 kernel on `feature.float()`.
__global__ __launch_bounds__(256) void cv_pack_pairs(const uint16_t* __restrict__ a, uint32_t* __restrict__ pa,
                                                     const uint16_t* __restrict__ b, uint32_t* __restrict__ pb, int C2,
                                                     int hw, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // over B * C/2 * hw
    if (i >= total) return;
    const long p = i % hw, c2 = (i / hw) % C2, n = i / ((long)hw * C2);
    const long src = (n * 2 * C2 + 2 * c2) * hw + p;
    pa[i] = (uint32_t)a[src] | ((uint32_t)a[src + hw] << 16);
    pb[i] = (uint32_t)b[src] | ((uint32_t)b[src + hw] << 16);
}

struct alignas(4) PairU { uint32_t a, b; };
__device__ __forceinline__ float lo_f(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float hi_f(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }

template <int DB>
__global__ __launch_bounds__(256) void cost_volume_fwd_bf16(const uint32_t* __restrict__ cur,      // [B][C/2][hw] pairs
                                                            const uint32_t* __restrict__ lookup,
                                                            const float* __restrict__ P,
                                                            const float* __restrict__ inv_K,
                                                            const float* __restrict__ bins,
                                                            const int32_t* __restrict__ skip,
                                                            float* __restrict__ cost, int C, int h, int w, int D,
                                                            float eps) {
    const int b = blockIdx.z, d0 = blockIdx.y * DB;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int hw = h * w, C2 = C >> 1;
    if (i >= hw) return;
    float* outp = cost + ((long)b * D + d0) * hw + i;
    const int nd = min(DB, D - d0);
    const int py = i / w, px = i - py * w;
    if ((skip != nullptr && skip[b] != 0) || px < 2 || px >= w - 2 || py < 2 || py >= h - 2) {
        for (int k = 0; k < nd; ++k) outp[(long)k * hw] = 0.f;
        return;
    }
    const float* ik = inv_K + b * 16;
    const float* pm = P + b * 12;
    const float fx = (float)px, fy = (float)py;
    float ray[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) ray[k] = (ik[k * 4] * fx + ik[k * 4 + 1] * fy) + ik[k * 4 + 2];
    long off[DB];
    float w00[DB], w01[DB], w10[DB], w11[DB];
    bool inx[DB], iny[DB];
#pragma unroll
    for (int k = 0; k < DB; ++k) {
        off[k] = -1;
        inx[k] = iny[k] = true;
        w00[k] = w01[k] = w10[k] = w11[k] = 0.f;
        if (k >= nd) continue;
        const float depth = bins[d0 + k];
        float X[3], cam[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) X[q] = depth * ray[q];
#pragma unroll
        for (int q = 0; q < 3; ++q)
            cam[q] = ((pm[q * 4] * X[0] + pm[q * 4 + 1] * X[1]) + pm[q * 4 + 2] * X[2]) + pm[q * 4 + 3];
        const float iz = cam[2] + eps;
        const float gx = ((cam[0] / iz) / (float)(w - 1) - 0.5f) * 2.f;
        const float gy = ((cam[1] / iz) / (float)(h - 1) - 0.5f) * 2.f;
        const float xv = (gx / 2.f + 0.5f) * (float)(w - 1);
        const float yv = (gy / 2.f + 0.5f) * (float)(h - 1);
        if (!(xv >= 2.0f && xv <= (float)(w - 2) && yv >= 2.0f && yv <= (float)(h - 2))) continue;
        const float ix = ((gx + 1.f) / 2.f) * (float)(w - 1);
        const float iy = ((gy + 1.f) / 2.f) * (float)(h - 1);
        const float flx = floorf(ix), fly = floorf(iy);
        const int x0 = (int)flx, y0 = (int)fly;
        const float tx = ix - flx, ty = iy - fly;
        w00[k] = (1.f - tx) * (1.f - ty); w01[k] = tx * (1.f - ty); w10[k] = (1.f - tx) * ty; w11[k] = tx * ty;
        inx[k] = x0 + 1 < w;
        iny[k] = y0 + 1 < h;
        off[k] = (long)y0 * w + x0;
    }
    const uint32_t* lkb = lookup + (long)b * C2 * hw;
    const uint32_t* cu = cur + (long)b * C2 * hw + i;
    double acc[DB];
#pragma unroll
    for (int k = 0; k < DB; ++k) acc[k] = 0.0;
    for (int c = 0; c < C2; ++c) {
        const uint32_t cv = cu[(long)c * hw];
        const float c0 = lo_f(cv), c1 = hi_f(cv);
        const uint32_t* lc = lkb + (long)c * hw;
#pragma unroll
        for (int k = 0; k < DB; ++k) {
            if (off[k] < 0) continue;
            PairU top, bot;
            if (inx[k] && iny[k]) {
                top = *reinterpret_cast<const PairU*>(lc + off[k]);
                bot = *reinterpret_cast<const PairU*>(lc + off[k] + w);
            } else {                         // footprint on the last column / row (ix == w - 2 rounded up): zero fill
                top.a = lc[off[k]];
                top.b = inx[k] ? lc[off[k] + 1] : 0u;
                bot.a = iny[k] ? lc[off[k] + w] : 0u;
                bot.b = 0u;
            }
            const float wa = ((lo_f(top.a) * w00[k] + lo_f(top.b) * w01[k]) + lo_f(bot.a) * w10[k]) + lo_f(bot.b) * w11[k];
            acc[k] += (double)fabsf(wa - c0);
            const float wb = ((hi_f(top.a) * w00[k] + hi_f(top.b) * w01[k]) + hi_f(bot.a) * w10[k]) + hi_f(bot.b) * w11[k];
            acc[k] += (double)fabsf(wb - c1);
        }
    }
#pragma unroll
    for (int k = 0; k < DB; ++k) {
        if (k >= nd) continue;
        float v = 0.f;
        if (off[k] >= 0) {
            const float diff = (float)(acc[k] / (double)C);
            v = diff / ((diff > 0.f ? 1.f : 0.f) + 1e-7f);
        }
        outp[(long)k * hw] = v;
    }
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
    const char* e = getenv("PPEA_CV_DB");
    const int db = e ? atoi(e) : 4;
#define CV_LAUNCH(DB_)                                                                                              \
    hipLaunchKernelGGL(cost_volume_fwd<DB_>, dim3((h * w + 255) / 256, (D + DB_ - 1) / DB_, B), dim3(256), 0,         \
                       (hipStream_t)stream, cur, lookup, P, inv_K, bins, skip, cost, C, h, w, D, eps)
    if (db == 1) CV_LAUNCH(1); else if (db == 2) CV_LAUNCH(2); else if (db == 8) CV_LAUNCH(8); else CV_LAUNCH(4);
#undef CV_LAUNCH
    return launch_status();
}

// bf16 features [B][C][h][w] (C even); `pairs`: caller-owned workspace of 2 * B * C/2 * h * w uint32 (= the two inputs' bytes).
// Same result, bit for bit, as ppea_cost_volume_fwd_f32 on the features widened to fp32.
int ppea_cost_volume_fwd_bf16(const void* cur, const void* lookup, void* pairs, const float* P, const float* inv_K,
                              const float* bins, const int32_t* skip, float* cost, int B, int C, int h, int w, int D,
                              float eps, void* stream) {
    if (B < 0 || C <= 0 || (C & 1) || h < 5 || w < 5 || D <= 0 || D > 65535) return PPEA_ERR_UNSUPPORTED;
    if (B == 0) return 0;
    if (pairs == nullptr) return PPEA_ERR_ARG;
    const long n = (long)B * (C / 2) * h * w;
    uint32_t* pc = (uint32_t*)pairs;
    uint32_t* pl = pc + n;
    hipLaunchKernelGGL(cv_pack_pairs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)cur,
                       pc, (const uint16_t*)lookup, pl, C / 2, h * w, n);
    constexpr int DB = 4;
    hipLaunchKernelGGL(cost_volume_fwd_bf16<DB>, dim3((h * w + 255) / 256, (D + DB - 1) / DB, B), dim3(256), 0,
                       (hipStream_t)stream, pc, pl, P, inv_K, bins, skip, cost, C, h, w, D, eps);
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
