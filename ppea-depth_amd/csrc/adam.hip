// Adam over ONE flat fp32 parameter buffer (trainer.py:350 `self.model_optimizer.step()`, torch.optim.Adam with
// the reference's defaults: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad).  The 1 306 trainable
// tensors of the step live in one contiguous buffer (dist.py), so the update is a single streaming pass:
// 7 fp32 words of traffic per parameter, and the bf16 working copy of the first n_lo parameters (the dense conv /
// linear weights the forward reads) is written by the same pass.
//
//   m = b1 m + (1 - b1) g;   v = b2 v + (1 - b2) g^2
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)          (torch's operation order)
//
// state[0] = t (already incremented for this step, as float), state[1] = lr: device scalars, so that a captured
// step graph sees the schedule.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        uint16_t* __restrict__ w16, long n, long n_lo,
                                                        const float* __restrict__ state, float b1, float b2,
                                                        float eps, float gscale) {
    const float t = state[0], lr = state[1];
    const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
    const float step_size = lr / bc1, sq_bc2 = sqrtf(bc2);
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float ga[4] = {gg.x * gscale, gg.y * gscale, gg.z * gscale, gg.w * gscale};
        float ma[4] = {mm.x, mm.y, mm.z, mm.w}, va[4] = {vv.x, vv.y, vv.z, vv.w}, pa[4] = {pp.x, pp.y, pp.z, pp.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ma[k] = ma[k] + (1.f - b1) * (ga[k] - ma[k]);                    // lerp(m, g, 1 - b1)
            va[k] = b2 * va[k] + (1.f - b2) * ga[k] * ga[k];
            pa[k] = pa[k] - step_size * ma[k] / (sqrtf(va[k]) / sq_bc2 + eps);
        }
        reinterpret_cast<float4*>(m)[i] = make_float4(ma[0], ma[1], ma[2], ma[3]);
        reinterpret_cast<float4*>(v)[i] = make_float4(va[0], va[1], va[2], va[3]);
        reinterpret_cast<float4*>(p)[i] = make_float4(pa[0], pa[1], pa[2], pa[3]);
        const long e = i << 2;
        if (w16 != nullptr && e < n_lo) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (e + k < n_lo) w16[e + k] = f32_to_bf16(pa[k]);
        }
    }
    // tail (n % 4 elements)
    const long tail0 = n4 << 2;
    const long j = tail0 + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) {
        const float gj = g[j] * gscale;
        const float mj = m[j] + (1.f - b1) * (gj - m[j]);
        const float vj = b2 * v[j] + (1.f - b2) * gj * gj;
        const float pj = p[j] - step_size * mj / (sqrtf(vj) / sq_bc2 + eps);
        m[j] = mj; v[j] = vj; p[j] = pj;
        if (w16 != nullptr && j < n_lo) w16[j] = f32_to_bf16(pj);
    }
}

}  // namespace

extern "C" {

// p, g, m, v: fp32 [n] (16-byte aligned); w16: bf16 [n_lo] working copy of p[0, n_lo) or NULL;
// state: device float[2] = {step t >= 1, learning rate}.
int ppea_adam_flat_f32(float* p, const float* g, float* m, float* v, void* w16, long n, long n_lo, const float* state,
                       float beta1, float beta2, float eps, void* stream) {
    if (n <= 0 || n_lo < 0 || n_lo > n) return PPEA_ERR_UNSUPPORTED;
    long blocks = ((n >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                       (uint16_t*)w16, n, n_lo, state, beta1, beta2, eps, 1.f);
    return launch_status();
}

// The same step on g * grad_scale (one fp32 multiply per element, i.e. the values a separate scaling pass would have
// stored): data-parallel training sums the gradients over the ranks and divides by their number here instead of in a pass of
// its own over the gradient buffer (trainer.py:215-222: DDP's mean).
int ppea_adam_flat_scaled_f32(float* p, const float* g, float* m, float* v, void* w16, long n, long n_lo, const float* state,
                              float beta1, float beta2, float eps, float grad_scale, void* stream) {
    if (n <= 0 || n_lo < 0 || n_lo > n) return PPEA_ERR_UNSUPPORTED;
    long blocks = ((n >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                       (uint16_t*)w16, n, n_lo, state, beta1, beta2, eps, grad_scale);
    return launch_status();
}

}  // extern "C"
