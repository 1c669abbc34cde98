// Photometric loss terms, gfx950 (all HBM-bound; each tensor crosses HBM once per kernel):
//   A21+A22  0.85*mean_C SSIM + 0.15*mean_C |t-p|      (layers.py:226-257, trainer.py:995-1007)
//   A23      edge-aware smoothness                      (layers.py:210-223)
//   A24/A25  per-pixel min / selec_reproj / automask    (trainer.py:1069-1091)
//
// SSIM forward: one wave walks a 62-column strip downwards; the horizontal 3-tap box sums
// of x, y, x^2, y^2, xy come from wavefront shuffles (lane = padded column), the vertical
// 3-tap sum from a 3-row register ring -- no LDS, no temporaries in HBM (the reference
// materialises 5 pooled maps + 2 padded copies + ~15 elementwise temporaries per call).
#include "common.h"

namespace {

constexpr float SSIM_C1 = 0.0001f;   // 0.01^2
constexpr float SSIM_C2 = 0.0009f;   // 0.03^2

__device__ __forceinline__ int reflect1(int i, int n) {   // ReflectionPad2d(1) index map for i in [-1, n]
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

__device__ __forceinline__ float ssim_from_sums(float sx, float sy, float sxx, float syy, float sxy) {
    const float mx = sx * (1.f / 9.f), my = sy * (1.f / 9.f);
    const float vx = sxx * (1.f / 9.f) - mx * mx;
    const float vy = syy * (1.f / 9.f) - my * my;
    const float cxy = sxy * (1.f / 9.f) - mx * my;
    const float n = (2.f * mx * my + SSIM_C1) * (2.f * cxy + SSIM_C2);
    const float d = (mx * mx + my * my + SSIM_C1) * (vx + vy + SSIM_C2);
    return fminf(fmaxf((1.f - n / d) * 0.5f, 0.f), 1.f);
}

constexpr int STRIP = 62;    // output columns per wave (64 lanes incl. one halo column each side)
constexpr int RCH = 8;       // output rows per wave

__global__ __launch_bounds__(256) void ssim_l1_fwd(const float* __restrict__ pred,
                                                   const float* __restrict__ target,
                                                   float* __restrict__ out, long out_bstride, int C,
                                                   int H, int W, float alpha, int nstrips, int nchunks,
                                                   long n_items) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const int per_img = nstrips * nchunks;
    const int b = (int)(item / per_img);
    const int r = (int)(item - (long)b * per_img);
    const int chunk = r / nstrips, strip = r - chunk * nstrips;
    const int pc = strip * STRIP - 1 + lane;                 // padded column of this lane
    const int sc = reflect1(min(pc, W), W);                  // source column (lanes past W+1 idle)
    const int ry0 = chunk * RCH;
    const float wa = alpha / (float)C, wl = (1.f - alpha) / (float)C;

    float acc[RCH];
#pragma unroll
    for (int i = 0; i < RCH; ++i) acc[i] = 0.f;

    for (int c = 0; c < C; ++c) {
        const float* xp = pred + ((long)b * C + c) * H * W;
        const float* yp = target + ((long)b * C + c) * H * W;
        float h1[5], h2[5];                                  // horizontal sums of rows r-1, r-2
        float cx = 0.f, cy = 0.f;                            // centre values of the previous row
#pragma unroll
        for (int k = 0; k < 5; ++k) { h1[k] = 0.f; h2[k] = 0.f; }
#pragma unroll
        for (int it = 0; it < RCH + 2; ++it) {
            const int pr = ry0 - 1 + it;                     // padded row
            const int sr = reflect1(min(pr, H), H);
            const float x = xp[(long)sr * W + sc], y = yp[(long)sr * W + sc];
            float q[5] = {x, y, x * x, y * y, x * y}, h[5];
#pragma unroll
            for (int k = 0; k < 5; ++k)
                h[k] = (__shfl_up(q[k], 1, WAVE) + q[k]) + __shfl_down(q[k], 1, WAVE);
            if (it >= 2) {
                const float s = ssim_from_sums((h2[0] + h1[0]) + h[0], (h2[1] + h1[1]) + h[1],
                                               (h2[2] + h1[2]) + h[2], (h2[3] + h1[3]) + h[3],
                                               (h2[4] + h1[4]) + h[4]);
                acc[it - 2] += wa * s + wl * fabsf(cy - cx);
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) { h2[k] = h1[k]; h1[k] = h[k]; }
            cx = x; cy = y;
        }
    }
    const int oc = pc;
    if (lane >= 1 && lane <= STRIP && oc < W) {
#pragma unroll
        for (int i = 0; i < RCH; ++i) {
            const int orow = ry0 + i;
            if (orow < H) out[(long)b * out_bstride + (long)orow * W + oc] = acc[i];
        }
    }
}

// Backward of the same term w.r.t. `pred`.  Tile of 16 x 32 pixels per block; per channel:
//   (1) stage pred/target with a 2-pixel reflected halo into LDS,
//   (2) per window centre q (tile + 1 halo) compute g_q * dS/d{mu_x, E[x^2], E[xy]},
//   (3) every pixel p gathers its <= 3x3 windows (with the multiplicity reflection gives the
//       second and second-to-last row/column) and adds the L1 sign term.
constexpr int BT_H = 16, BT_W = 32;

__global__ __launch_bounds__(256) void ssim_l1_bwd(const float* __restrict__ pred,
                                                   const float* __restrict__ target,
                                                   const float* __restrict__ d_out, long dout_bstride,
                                                   float* __restrict__ d_pred, int C, int H, int W,
                                                   float alpha) {
    __shared__ float xs[BT_H + 4][BT_W + 5], ys[BT_H + 4][BT_W + 5];
    __shared__ float cm[BT_H + 2][BT_W + 3], ca[BT_H + 2][BT_W + 3], cc[BT_H + 2][BT_W + 3];
    const int b = blockIdx.z;
    const int py0 = blockIdx.y * BT_H, px0 = blockIdx.x * BT_W;
    const int tid = threadIdx.x;
    const float wa = alpha / (float)C, wl = (1.f - alpha) / (float)C;
    const float* go = d_out + (long)b * dout_bstride;

    for (int c = 0; c < C; ++c) {
        const float* xp = pred + ((long)b * C + c) * H * W;
        const float* yp = target + ((long)b * C + c) * H * W;
        for (int i = tid; i < (BT_H + 4) * (BT_W + 4); i += 256) {
            const int r = i / (BT_W + 4), cidx = i - r * (BT_W + 4);
            const int gy = min(max(py0 - 2 + r, -1), H), gx = min(max(px0 - 2 + cidx, -1), W);
            const long off = (long)reflect1(gy, H) * W + reflect1(gx, W);
            xs[r][cidx] = xp[off];
            ys[r][cidx] = yp[off];
        }
        __syncthreads();
        for (int i = tid; i < (BT_H + 2) * (BT_W + 2); i += 256) {
            const int r = i / (BT_W + 2), cidx = i - r * (BT_W + 2);
            const int qy = py0 - 1 + r, qx = px0 - 1 + cidx;
            float vm = 0.f, va = 0.f, vc = 0.f;
            if (qy >= 0 && qy < H && qx >= 0 && qx < W) {
                float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float x = xs[r + dy][cidx + dx], y = ys[r + dy][cidx + dx];
                        sx += x; sy += y; sxx += x * x; syy += y * y; sxy += x * y;
                    }
                const float m = sx * (1.f / 9.f), n = sy * (1.f / 9.f);
                const float vx = sxx * (1.f / 9.f) - m * m, vy = syy * (1.f / 9.f) - n * n;
                const float cxy = sxy * (1.f / 9.f) - m * n;
                const float N1 = 2.f * m * n + SSIM_C1, N2 = 2.f * cxy + SSIM_C2;
                const float D1 = m * m + n * n + SSIM_C1, D2 = vx + vy + SSIM_C2;
                const float inv = 1.f / (D1 * D2);
                const float S = N1 * N2 * inv;
                const float v = (1.f - S) * 0.5f;
                if (v >= 0.f && v <= 1.f) {                  // clamp passes gradient on [0, 1]
                    const float g = go[(long)qy * W + qx] * wa * (-0.5f) * (1.f / 9.f);
                    vm = g * (2.f * n * (N2 - N1) - 2.f * m * S * (D2 - D1)) * inv;
                    va = g * (-S / D2);
                    vc = g * (2.f * N1 * inv);
                }
            }
            cm[r][cidx] = vm; ca[r][cidx] = va; cc[r][cidx] = vc;
        }
        __syncthreads();
        for (int i = tid; i < BT_H * BT_W; i += 256) {
            const int r = i / BT_W, cidx = i - r * BT_W;
            const int py = py0 + r, px = px0 + cidx;
            if (py < H && px < W) {
                float sm = 0.f, sa = 0.f, sc = 0.f;
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy) {
                    const int qy = py + dy;
                    float my = 1.f;
                    if (py == 1 && qy == 0) my += 1.f;
                    if (py == H - 2 && qy == H - 1) my += 1.f;
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int qx = px + dx;
                        float mlt = my;
                        if (px == 1 && qx == 0) mlt += my;
                        if (px == W - 2 && qx == W - 1) mlt += my;
                        sm += mlt * cm[r + 1 + dy][cidx + 1 + dx];
                        sa += mlt * ca[r + 1 + dy][cidx + 1 + dx];
                        sc += mlt * cc[r + 1 + dy][cidx + 1 + dx];
                    }
                }
                const float x = xs[r + 2][cidx + 2], y = ys[r + 2][cidx + 2];
                const float dl = x - y;
                const float sgn = (dl > 0.f) ? 1.f : ((dl < 0.f) ? -1.f : 0.f);
                d_pred[((long)b * C + c) * H * W + (long)py * W + px] =
                    (sm + 2.f * x * sa + y * sc) + wl * sgn * go[(long)py * W + px];
            }
        }
        __syncthreads();
    }
}

// ---- smoothness -----------------------------------------------------------------------------
constexpr int SMOOTH_BLOCKS = 512;

__global__ __launch_bounds__(256) void smooth_fwd(const float* __restrict__ disp,
                                                  const float* __restrict__ img,
                                                  float* __restrict__ partials, int B, int C, int H, int W) {
    __shared__ float red[4][2];
    const long total = (long)B * H * W;
    float sx = 0.f, sy = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const int b = (int)(i / ((long)H * W));
        const float d = disp[i];
        const long ib = (long)b * C * H * W + (long)y * W + x;
        if (x < W - 1) {
            float e = 0.f;
            for (int c = 0; c < C; ++c) e += fabsf(img[ib + (long)c * H * W] - img[ib + (long)c * H * W + 1]);
            sx += fabsf(d - disp[i + 1]) * expf(-e / (float)C);
        }
        if (y < H - 1) {
            float e = 0.f;
            for (int c = 0; c < C; ++c) e += fabsf(img[ib + (long)c * H * W] - img[ib + (long)c * H * W + W]);
            sy += fabsf(d - disp[i + W]) * expf(-e / (float)C);
        }
    }
    sx = wave_sum(sx);
    sy = wave_sum(sy);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave][0] = sx; red[wave][1] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[blockIdx.x * 2] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        partials[blockIdx.x * 2 + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    }
}

__device__ __forceinline__ float sgnf(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void smooth_bwd(const float* __restrict__ disp,
                                                  const float* __restrict__ img, float gx, float gy,
                                                  float* __restrict__ d_disp, int B, int C, int H, int W) {
    const long total = (long)B * H * W;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const int b = (int)(i / ((long)H * W));
    const long ib = (long)b * C * H * W + (long)y * W + x;
    const float d = disp[i];
    float g = 0.f;
    if (x < W - 1) {
        float e = 0.f;
        for (int c = 0; c < C; ++c) e += fabsf(img[ib + (long)c * H * W] - img[ib + (long)c * H * W + 1]);
        g += gx * sgnf(d - disp[i + 1]) * expf(-e / (float)C);
    }
    if (x > 0) {
        float e = 0.f;
        for (int c = 0; c < C; ++c) e += fabsf(img[ib + (long)c * H * W - 1] - img[ib + (long)c * H * W]);
        g -= gx * sgnf(disp[i - 1] - d) * expf(-e / (float)C);
    }
    if (y < H - 1) {
        float e = 0.f;
        for (int c = 0; c < C; ++c) e += fabsf(img[ib + (long)c * H * W] - img[ib + (long)c * H * W + W]);
        g += gy * sgnf(d - disp[i + W]) * expf(-e / (float)C);
    }
    if (y > 0) {
        float e = 0.f;
        for (int c = 0; c < C; ++c) e += fabsf(img[ib + (long)c * H * W - W] - img[ib + (long)c * H * W]);
        g -= gy * sgnf(disp[i - W] - d) * expf(-e / (float)C);
    }
    d_disp[i] = g;
}

// ---- per-pixel selection ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void loss_select(const float* __restrict__ reproj,
                                                   const float* __restrict__ identity,
                                                   const float* __restrict__ wm1,
                                                   const float* __restrict__ wp1,
                                                   const float* __restrict__ noise,
                                                   float* __restrict__ sel, uint8_t* __restrict__ src_idx,
                                                   int64_t* __restrict__ frame_idx,
                                                   int64_t* __restrict__ auto_idx, int B, int C, int H,
                                                   int W, int selec) {
    const long hw = (long)H * W;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * hw) return;
    const long b = i / hw, p = i - b * hw;
    const float r0 = reproj[b * 2 * hw + p], r1 = reproj[b * 2 * hw + hw + p];
    // torch.min over dim=1: first minimum wins; NaN propagates (treated as smaller)
    int fi = (r1 < r0 || (r1 != r1 && r0 == r0)) ? 1 : 0;
    float v = fi ? r1 : r0;
    int si = fi;
    if (selec) {
        float s0 = 0.f, s1 = 0.f;
        for (int c = 0; c < C; ++c) {
            s0 += wm1[(b * C + c) * hw + p];
            s1 += wp1[(b * C + c) * hw + p];
        }
        const bool m0 = s0 < 0.1f, m1 = s1 < 0.1f;
        if (m0) { v = r1; si = 1; }
        if (m1) { v = r0; si = 0; }
        if (m0 && m1) { v = 0.f; si = 2; }
    }
    const float i0 = identity[b * 2 * hw + p], i1 = identity[b * 2 * hw + hw + p];
    float idm = (i1 < i0 || (i1 != i1 && i0 == i0)) ? i1 : i0;
    if (noise != nullptr) idm += noise[i];
    sel[i] = v;
    src_idx[i] = (uint8_t)si;
    frame_idx[i] = fi;
    auto_idx[i] = (idm < v || (idm != idm && v == v)) ? 1 : 0;
}


// ---- the tail of compute_losses (trainer.py:1092-1139) --------------------------------------------------------------------
// After the per-pixel selection: the (auto)mask / motion mask, the masked mean of the reprojection loss
// rl = sum(sel * mask) / (sum(mask) + 1e-7), and for the multi-frame pass the consistency term
// mean(|multi_depth - mono_depth| * (1 - mask)) and its target 1 / (mono * cm + multi * (1 - cm)).  The reference issues ~20
// element-wise / reduction kernels per pass for this (and autograd as many again backward); here: one pass + a finalize
// forward, one pass backward, sums in a fixed order (per-block partials, then one block in fp64).
constexpr int TAIL_TPB = 256, TAIL_PER = 4;

__global__ __launch_bounds__(TAIL_TPB) void loss_tail_fwd(const float* __restrict__ sel, const int64_t* __restrict__ auto_idx,
                                                          const float* __restrict__ cons, const float* __restrict__ aug,
                                                          const float* __restrict__ multi, const float* __restrict__ mono,
                                                          float* __restrict__ mask_out, float* __restrict__ target,
                                                          float* __restrict__ partial, long hw, long total, int is_multi) {
    __shared__ float red[3][TAIL_TPB / 64];
    float s_num = 0.f, s_den = 0.f, s_con = 0.f;
    const long base = ((long)blockIdx.x * TAIL_TPB + threadIdx.x) * TAIL_PER;
#pragma unroll
    for (int k = 0; k < TAIL_PER; ++k) {
        const long i = base + k;
        if (i >= total) break;
        float m;
        if (!is_multi) {
            m = (auto_idx == nullptr) ? 1.f : (auto_idx[i] == 0 ? 1.f : 0.f);
        } else {
            m = 1.f;
            if (cons != nullptr) m = m * cons[i];
            if (aug != nullptr) m = m * (1.f - aug[i / hw]);
        }
        mask_out[i] = m;
        s_num += sel[i] * m;
        s_den += m;
        if (is_multi) {
            const float cm = 1.f - m;
            const float a = multi[i], b = mono[i];
            s_con += fabsf(a - b) * cm;
            target[i] = 1.f / (b * cm + a * (1.f - cm));
        }
    }
    s_num = wave_sum(s_num); s_den = wave_sum(s_den); s_con = wave_sum(s_con);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wv] = s_num; red[1][wv] = s_den; red[2][wv] = s_con; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < TAIL_TPB / 64; ++w) t += red[threadIdx.x][w];
        partial[(long)blockIdx.x * 3 + threadIdx.x] = t;
    }
}

// out[0] = rl, out[1] = consistency loss, out[2] = 1 / (sum(mask) + 1e-7)
__global__ __launch_bounds__(256) void loss_tail_finalize(const float* __restrict__ partial, int nblk, long total,
                                                          float* __restrict__ out) {
    __shared__ double red[3][256];
    double a = 0.0, b = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) { a += partial[i * 3]; b += partial[i * 3 + 1]; c += partial[i * 3 + 2]; }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b; red[2][threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
#pragma unroll
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float num = (float)red[0][0], den = (float)red[1][0] + 1e-7f;
        out[0] = num / den;
        out[1] = (float)(red[2][0] / (double)total);
        out[2] = 1.f / den;
    }
}

// d_reproj[b][c][p] = g_rl * mask * inv_den if src[b][p] == c else 0 (c = 0, 1; src == 2: the selection forced zero);
// d_multi = g_con * sign(multi - mono) * (1 - mask) / total
__global__ __launch_bounds__(256) void loss_tail_bwd(const float* __restrict__ mask, const uint8_t* __restrict__ src,
                                                     const float* __restrict__ out, const float* __restrict__ g_rl,
                                                     const float* __restrict__ g_con, const float* __restrict__ multi,
                                                     const float* __restrict__ mono, float* __restrict__ d_reproj,
                                                     float* __restrict__ d_multi, long hw, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float m = mask[i];
    if (d_reproj != nullptr) {
        const float g = (g_rl != nullptr ? g_rl[0] : 0.f) * m * out[2];
        const long b = i / hw, p = i - b * hw;
        const int sidx = src[i];
        d_reproj[b * 2 * hw + p] = sidx == 0 ? g : 0.f;
        d_reproj[b * 2 * hw + hw + p] = sidx == 1 ? g : 0.f;
    }
    if (d_multi != nullptr) {
        const float d = multi[i] - mono[i];
        const float sg = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
        d_multi[i] = (g_con != nullptr ? g_con[0] : 0.f) * sg * (1.f - m) / (float)total;
    }
}

}  // namespace

extern "C" {

int ppea_ssim_l1_fwd_f32(const float* pred, const float* target, float* out, long out_bstride, int B, int C,
                         int H, int W, float alpha, void* stream) {
    if (B < 0 || C <= 0 || H < 2 || W < 2) return PPEA_ERR_UNSUPPORTED;
    if (B == 0) return 0;
    const int nstrips = (W + STRIP - 1) / STRIP, nchunks = (H + RCH - 1) / RCH;
    const long n_items = (long)B * nstrips * nchunks;
    hipLaunchKernelGGL(ssim_l1_fwd, dim3((unsigned)((n_items + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       pred, target, out, out_bstride, C, H, W, alpha, nstrips, nchunks, n_items);
    return launch_status();
}

int ppea_ssim_l1_bwd_f32(const float* pred, const float* target, const float* d_out, long dout_bstride,
                         float* d_pred, int B, int C, int H, int W, float alpha, void* stream) {
    if (B < 0 || C <= 0 || H < 2 || W < 2) return PPEA_ERR_UNSUPPORTED;
    if (B == 0) return 0;
    dim3 g((W + BT_W - 1) / BT_W, (H + BT_H - 1) / BT_H, B);
    hipLaunchKernelGGL(ssim_l1_bwd, g, dim3(256), 0, (hipStream_t)stream, pred, target, d_out, dout_bstride,
                       d_pred, C, H, W, alpha);
    return launch_status();
}

int ppea_smooth_num_partials(void) { return SMOOTH_BLOCKS; }

int ppea_smooth_fwd_f32(const float* disp, const float* img, float* partials, int B, int C, int H, int W,
                        void* stream) {
    if (B <= 0 || C <= 0 || H < 2 || W < 2) return PPEA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(smooth_fwd, dim3(SMOOTH_BLOCKS), dim3(256), 0, (hipStream_t)stream, disp, img,
                       partials, B, C, H, W);
    return launch_status();
}

int ppea_smooth_bwd_f32(const float* disp, const float* img, float gx, float gy, float* d_disp, int B, int C,
                        int H, int W, void* stream) {
    if (B <= 0 || C <= 0 || H < 2 || W < 2) return PPEA_ERR_UNSUPPORTED;
    const long total = (long)B * H * W;
    hipLaunchKernelGGL(smooth_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       disp, img, gx, gy, d_disp, B, C, H, W);
    return launch_status();
}

int ppea_loss_select_f32(const float* reproj, const float* identity, const float* warped_m1,
                         const float* warped_p1, const float* noise, float* sel, uint8_t* src_idx,
                         int64_t* frame_idx, int64_t* auto_idx, int B, int C, int H, int W, int selec_reproj,
                         void* stream) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return PPEA_ERR_UNSUPPORTED;
    const long total = (long)B * H * W;
    hipLaunchKernelGGL(loss_select, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reproj, identity, warped_m1, warped_p1, noise, sel, src_idx, frame_idx, auto_idx, B, C,
                       H, W, selec_reproj);
    return launch_status();
}


// Tail of compute_losses (trainer.py:1092-1139).  sel, mask, target, multi, mono: [B][1][H][W] fp32; auto_idx int64 or NULL;
// cons [B][H][W] fp32 or NULL; aug [B] fp32 or NULL; partial: workspace of ppea_loss_tail_blocks(B * H * W) * 3 floats;
// out[0] = rl, out[1] = consistency loss (multi), out[2] = 1 / (sum(mask) + 1e-7).
int ppea_loss_tail_blocks(long total) { return (int)((total + TAIL_TPB * TAIL_PER - 1) / (TAIL_TPB * TAIL_PER)); }
int ppea_loss_tail_fwd_f32(const float* sel, const int64_t* auto_idx, const float* cons, const float* aug, const float* multi,
                           const float* mono, float* mask, float* target, float* partial, float* out, int B, int H, int W,
                           int is_multi, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return PPEA_ERR_UNSUPPORTED;
    if (is_multi && (multi == nullptr || mono == nullptr || target == nullptr)) return PPEA_ERR_ARG;
    const long hw = (long)H * W, total = (long)B * hw;
    const int nblk = ppea_loss_tail_blocks(total);
    hipLaunchKernelGGL(loss_tail_fwd, dim3(nblk), dim3(TAIL_TPB), 0, (hipStream_t)stream, sel, auto_idx, cons, aug, multi, mono,
                       mask, target, partial, hw, total, is_multi);
    hipLaunchKernelGGL(loss_tail_finalize, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nblk, total, out);
    return launch_status();
}
int ppea_loss_tail_bwd_f32(const float* mask, const uint8_t* src, const float* out, const float* g_rl, const float* g_con,
                           const float* multi, const float* mono, float* d_reproj, float* d_multi, int B, int H, int W,
                           void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return PPEA_ERR_UNSUPPORTED;
    const long hw = (long)H * W, total = (long)B * hw;
    hipLaunchKernelGGL(loss_tail_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mask, src, out,
                       g_rl, g_con, multi, mono, d_reproj, d_multi, hw, total);
    return launch_status();
}

}  // extern "C"
