/*
 * ppea_depth.h -- C ABI of libppea_depth.so (gfx950 / MI355X).
 *
 * The drop-in boundary of the PPEA-Depth training hot path.
 * Every entry point replaces a stock ATen/cuDNN op sequence at the cited
 * reference call site (paths relative to /root/reference/ppeadepth/).
 *
 * Conventions
 *   - all pointers are DEVICE pointers into contiguous row-major (NCHW) buffers
 *     owned by the caller; kernels never allocate or free;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every
 *     call only enqueues work on that stream and returns;
 *   - the return value is a hipError_t (0 = hipSuccess); PPEA_ERR_UNSUPPORTED
 *     (= -1) is returned for argument combinations the library does not serve;
 *   - fp32 I/O entry points end in _f32, bf16 I/O (fp32 accumulate) in _bf16.
 */
#ifndef PPEA_DEPTH_H
#define PPEA_DEPTH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPEA_ERR_UNSUPPORTED (-1)
#define PPEA_ERR_ARG (-2)          /* inconsistent arguments (e.g. a required pointer is NULL) */

/* Library / build identification: returns the ABI version (bumped on any signature change). */
int ppea_abi_version(void);

/* Measurement aid (tools/step_timeline.py): a one-lane launch on `stream` that writes the device's constant-rate clock
 * (wall_clock64, 100 MHz) to *slot (uint64, device memory) -- a timestamp of that point of the stream that also works inside
 * a captured graph, where HIP events cannot be timed. */
int ppea_timestamp(void* slot, void* stream);

/* ------------------------------------------------------------------------------------------
 * A1  Large-kernel depthwise convolution  (networks/replknet_adapter.py:151-168 get_conv2d,
 *     :232-239 ReparamLargeKernelConv.forward).  stride 1, pad K/2, dilation 1, no bias.
 *     x [N,C,H,W]; w_big [C,1,K,K]; w_small [C,1,KS,KS] or NULL (then y_small is ignored).
 *     One launch produces both branches from one LDS-resident input tile.
 *     K odd, 7 <= K <= 31; KS in {0 (absent), 3, 5}.
 * ---------------------------------------------------------------------------------------- */
int ppea_dwconv_lk_fwd_f32(const float* x, const float* w_big, const float* w_small,
                           float* y_big, float* y_small,
                           int N, int C, int H, int W, int K, int KS, void* stream);
int ppea_dwconv_lk_fwd_bf16(const uint16_t* x, const float* w_big, const float* w_small,
                            uint16_t* y_big, uint16_t* y_small,
                            int N, int C, int H, int W, int K, int KS, void* stream);

/* dgrad of the pair: dx = corr(dy_big, flip(w_big)) + corr(dy_small, flip(w_small)).
 * dy_small / w_small may be NULL. */
int ppea_dwconv_lk_bwd_data_f32(const float* dy_big, const float* dy_small,
                                const float* w_big, const float* w_small, float* dx,
                                int N, int C, int H, int W, int K, int KS, void* stream);
int ppea_dwconv_lk_bwd_data_bf16(const uint16_t* dy_big, const uint16_t* dy_small,
                                 const float* w_big, const float* w_small, uint16_t* dx,
                                 int N, int C, int H, int W, int K, int KS, void* stream);

/* bf16 on the matrix cores (banded-Toeplitz MFMA kernel, csrc/dwconv_mfma.hip).  The filters are packed
 * ONCE per weight version (they are frozen in PPEA-Depth Stage 1/2, repdepth.py:47-50) into a bf16 image
 * from which every wave builds its register-resident Toeplitz fragments:
 *   ppea_dwconv_lk_packed_bytes(C, K)            size of the packed buffer for w [C,1,K,K]
 *   ppea_dwconv_lk_pack_bf16(w, packed, C, K, flip)   flip = 0 for fwd, 1 for dgrad (both axes reversed)
 *   ppea_dwconv_lk_fwd_bf16p / _bwd_data_bf16p   as the _bf16 entry points above, with packed filters
 *                                                (packed_small may be NULL; KS in {0, 5}; K in {31,29,27,13}).
 * Returns PPEA_ERR_UNSUPPORTED for shapes it does not serve; callers then use the _bf16 entry points. */
long ppea_dwconv_lk_packed_bytes(int C, int K);
int ppea_dwconv_lk_pack_bf16(const float* w, void* packed, int C, int K, int flip, void* stream);
int ppea_dwconv_lk_fwd_bf16p(const uint16_t* x, const void* packed_big, const void* packed_small,
                             uint16_t* y_big, uint16_t* y_small,
                             int N, int C, int H, int W, int K, int KS, void* stream);
/* Forward with the BatchNorm (+ ReLU) of the INPUT fused into the staging pass: RepLKBlock's pw1 conv_bn_relu -> large
 * kernel (replknet_adapter.py:305-308 with :182-197).  x = the 1x1 conv's output, `sums` [C][P][2] its epilogue's partial
 * (sum, sum of squares) (ppea_pwconv_stats_bf16), count = N*H*W.  Every wave finalises its channel's statistics (fp64, the
 * arithmetic of ppea_bn_finalize_sums_f32) and stages relu(gamma * (x - mean) * invstd + beta) rounded to bf16, zero
 * outside the plane; mean / invstd [C] are written for backward, running statistics updated unless NULL.  KS must be 5. */
int ppea_dwconv_lk_fwd_bn_bf16p(const uint16_t* x, const void* packed_big, const void* packed_small, uint16_t* y_big,
                                uint16_t* y_small, const float* sums, int P, long count, const float* gamma,
                                const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                float* mean, float* invstd, int N, int C, int H, int W, int K, int KS, void* stream);
/* forward + the statistics of the two BatchNorms that follow the re-parameterised pair (replknet_adapter.py:232-239):
 * stats [2][C][P][2] fp32, P = ppea_dwconv_lk_stats_partials(...) partial (sum, sum of squares) pairs per channel of the
 * stored y_big (first half) / y_small values; reduce each half with ppea_bn_finalize_sums_f32. */
int ppea_dwconv_lk_stats_partials(int N, int C, int H, int W, int K, int KS);
int ppea_dwconv_lk_fwd_stats_bf16p(const uint16_t* x, const void* packed_big, const void* packed_small, uint16_t* y_big,
                                   uint16_t* y_small, float* stats, int N, int C, int H, int W, int K, int KS,
                                   void* stream);
int ppea_dwconv_lk_bwd_data_bf16p(const uint16_t* dy_big, const uint16_t* dy_small,
                                  const void* packed_big_flip, const void* packed_small_flip,
                                  uint16_t* dx, int N, int C, int H, int W, int K, int KS, void* stream);

/* Depthwise 3x3, stride 1 or 2, pad 1, no bias (stem[1], stem[3], transitions[.][1];
 * replknet_adapter.py:414-416, 451-453).  w [C,1,3,3] fp32.  bwd_data: H, W are the forward INPUT sizes. */
int ppea_dwconv3x3_fwd_f32(const void* x, const float* w, void* y, int N, int C, int H, int W, int stride, void* stream);
int ppea_dwconv3x3_fwd_bf16(const void* x, const float* w, void* y, int N, int C, int H, int W, int stride, void* stream);
int ppea_dwconv3x3_bwd_data_f32(const void* dy, const float* w, void* dx, int N, int C, int H, int W, int stride, void* stream);
int ppea_dwconv3x3_bwd_data_bf16(const void* dy, const float* w, void* dx, int N, int C, int H, int W, int stride, void* stream);

/* wgrad (only needed with --fullft_reb, repdepth.py:47, and by the plug-in's weight.grad):
 * dw[c,0,u,v] = sum_{n,i,j} dy[n,c,i,j] * x[n,c,i+u-K/2,j+v-K/2].  dw is overwritten. */
int ppea_dwconv_lk_bwd_filter_f32(const float* x, const float* dy, float* dw,
                                  int N, int C, int H, int W, int K, void* stream);

/* ------------------------------------------------------------------------------------------
 * A5/A6  Pointwise (1x1) convolution on MFMA, NCHW bf16 (csrc/pwconv.hip): RepLKBlock.pw1/pw2,
 *     ConvFFN.pw1/pw2, stem[2], transitions[.][0] (replknet_adapter.py:270-271, 296-297, 415, 452).
 *     Y[b][m][p] = sum_k A[m][k] * X[b][k][p] (+ bias[m]);  A [M][K] bf16 row-major = the conv weight
 *     [Cout][Cin] for forward, its transpose for the data gradient.  X [B][K][HW], Y [B][M][HW].
 *     Fast path needs K % 32 == 0 and HW % 8 == 0; otherwise PPEA_ERR_UNSUPPORTED.
 * ---------------------------------------------------------------------------------------- */
int ppea_pwconv_bf16(const void* A, const void* X, const float* bias, void* Y, int B, int M, int K, int HW,
                     void* stream);

/* ------------------------------------------------------------------------------------------
 * A4  Adapters (replknet_adapter.py:20-47 `Adapter`: Linear -> GELU -> Linear over channels;
 *     :49-109 `B_Adapter`, adpt_test 4: Conv2d(C, C/4, 3, 1, 1) -> GELU -> Linear(C/4, C)), forward and
 *     all gradients, NCHW bf16 end to end (no layout changes), built from four entry points:
 *
 *   ppea_pwconv_ex_bf16   the GEMM above with an epilogue. epi 0: Y = A X + bias.  epi 1: Y = pre-activation,
 *                         Y2 = GELU(pre).  epi 2: Y = (A X) * GELU'(aux)  (aux, Y, Y2: [B][M][HW] bf16).
 *                         bias: fp32, or bf16 when bias_bf16 != 0.  a_transposed != 0: the matrix is passed as
 *                         At [K][M] (M % 8 == 0), so a data gradient consumes the forward weight unchanged.
 *   ppea_pwgrad_bf16      weight gradients: out[m*N + n] = sum_{b,p} P[b][m][p] Q[b][n][p] (fp32), followed by
 *                         the row sums of P (bias gradient) at out[M*N + m] when want_rowsum != 0.
 *                         P [B][M][HW], Q [B][N][HW] bf16, HW % 8 == 0.  `workspace`: device scratch of
 *                         ppea_pwgrad_workspace_bytes(B, M, N, HW) bytes (split-K partials; deterministic).
 *   ppea_tapsum_fwd_bf16  the 3x3 conv = GEMM with the tap-major weight matrix [9*Ch][C] (rows t*Ch + m,
 *                         t = 3*ky + kx) followed by this shift-and-add: pre[b][m][y][x] = bias[m] +
 *                         sum_t T[b][t*Ch+m][y+ky-1][x+kx-1] (zero padded), h = GELU(pre).  W % 4 == 0.
 *   ppea_tapsum_bwd_bf16  its adjoint: dT[b][t*Ch+m][y][x] = g[b][m][y-ky+1][x-kx+1].
 * ---------------------------------------------------------------------------------------- */
/* ppea_pwconv_bf16 + the statistics of the BatchNorm that follows it (conv_bn / conv_bn_relu, replknet_adapter.py:182-197):
 * stats [M][P][2] fp32 = per output channel (sum, sum of squares) of the stored bf16 values over P disjoint pixel sets,
 * P = ppea_pwconv_stats_partials(B, M, K, HW); ppea_bn_finalize_sums_f32 turns them into mean / biased var / invstd of
 * `count` values per channel and updates the running statistics (fp64 totals, fixed order) -- no statistics pass over Y. */
int ppea_pwconv_stats_partials(int B, int M, int K, int HW);
int ppea_pwconv_stats_bf16(const void* A, const void* X, const float* bias, void* Y, float* stats, int B, int M, int K,
                           int HW, void* stream);
int ppea_bn_finalize_sums_f32(const float* partial, int P, int C, long count, float eps, float momentum, float* mean,
                              float* var, float* invstd, float* running_mean, float* running_var, void* stream);
int ppea_pwconv_ex_bf16(const void* A, const void* X, const void* bias, int bias_bf16, int epi, const void* aux,
                        void* Y, void* Y2, int B, int M, int K, int HW, int a_transposed, void* stream);
long ppea_pwgrad_workspace_bytes(int B, int M, int N, int HW);
int ppea_pwgrad_bf16(const void* P, const void* Q, float* out, void* workspace, int B, int M, int N, int HW,
                     int want_rowsum, void* stream);
/* ppea_pwgrad_ex_bf16: same GEMM, results written straight into the parameter gradients: out_w fp32 or bf16
 * (out_w_bf16), [M][N] for taps = 1 or, for the tap-major 3x3 form (taps = 9, rows t*Ch + m), nn.Conv2d layout
 * [Ch][N][3][3]; out_b (may be NULL) = row sums of rows [b_row0, b_row0 + b_rows). */
int ppea_pwgrad_ex_bf16(const void* P, const void* Q, void* workspace, int B, int M, int N, int HW, void* out_w,
                        int out_w_bf16, int taps, void* out_b, int out_b_bf16, int b_row0, int b_rows, void* stream);
/* Two such problems over the same [B][.][HW] pixels (an adapter's D_fc2 and D_fc1 weight gradients, replknet_adapter.py:20-109)
 * in ONE GEMM launch and ONE reduce launch; every argument that differs is an array of two.  HW % 32 == 0, else
 * PPEA_ERR_UNSUPPORTED / a workspace size of -1: call ppea_pwgrad_ex_bf16 twice. */
long ppea_pwgrad_pair_workspace_bytes(int B, const int* M, const int* N, int HW);
int ppea_pwgrad_ex_pair_bf16(const void* const* P, const void* const* Q, void* workspace, int B, const int* M, const int* N,
                             int HW, void* const* out_w, const int* out_w_bf16, const int* taps, void* const* out_b,
                             const int* out_b_bf16, const int* b_row0, const int* b_rows, void* stream);
int ppea_tapsum_fwd_bf16(const void* T, const void* bias, int bias_bf16, void* pre, void* h, int B, int Ch, int H,
                         int W, void* stream);
int ppea_tapsum_bwd_bf16(const void* g, void* dT, int B, int Ch, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------
 * A2 (+A5/A6 glue)  Training-mode BatchNorm fused with its neighbours (csrc/bn_fused.hip):
 *     y = act( BN_a(z1) [+ BN_b(z2)] ) [* mask[n]] [+ r1] [+ r2_scale * r2]
 *   replaces conv_bn / conv_bn_relu (replknet_adapter.py:182-197), the two-branch sum of
 *   ReparamLargeKernelConv.forward (:232-239) and the residual / DropPath / adapter adds of
 *   RepLKBlock.forward (:315-326) and ConvFFN.forward (:283-289).  NCHW, HW = H*W.
 *   act: 0 none, 1 ReLU, 2 GELU(erf).  z2, mask [N], r1, r2 may be NULL.
 *   stats = host array of 8 DEVICE pointers {mean1, invstd1, gamma1, beta1, mean2, invstd2, gamma2, beta2}.
 *   forward : ppea_bn_stats_*   per-plane (mean, M2) -> partial [C][N][2]
 *             ppea_bn_finalize_f32  Chan-combine -> mean/var(biased)/invstd [C]; running stats updated
 *                                   (momentum, unbiased var) unless running_mean is NULL
 *             ppea_bn_apply_*
 *   backward: g = dy * mask[n] * act'(u);
 *             ppea_bn_bwd_reduce_*  -> partial [C][N][3] = sum g, sum g*zhat1, sum g*zhat2
 *             ppea_bn_bwd_finalize_f32 -> sums [3][C]  (d_beta = sums[0], d_gamma_k = sums[k])
 *             ppea_bn_bwd_apply_*   dz_k = gamma_k*invstd_k*(g - sums0*inv_count - zhat_k*sums_k*inv_count)
 * ---------------------------------------------------------------------------------------- */
int ppea_bn_stats_f32(const void* z, float* partial, int N, int C, int HW, void* stream);
int ppea_bn_stats_bf16(const void* z, float* partial, int N, int C, int HW, void* stream);
int ppea_bn_finalize_f32(const float* partial, int N, int C, int HW, float eps, float momentum,
                         float* mean, float* var, float* invstd, float* running_mean,
                         float* running_var, void* stream);
/* Small channels (N*HW <= 16384 elements, HW % 8 == 0, C >= 64): statistics (resp. backward sums) final in ONE
 * launch, one wave per channel over all N planes; PPEA_ERR_UNSUPPORTED otherwise (use the two-launch forms). */
int ppea_bn_stats_final_f32(const void* z, int N, int C, int HW, float eps, float momentum, float* mean, float* var,
                            float* invstd, float* running_mean, float* running_var, void* stream);
int ppea_bn_stats_final_bf16(const void* z, int N, int C, int HW, float eps, float momentum, float* mean, float* var,
                             float* invstd, float* running_mean, float* running_var, void* stream);
int ppea_bn_bwd_reduce_final_f32(const void* dy, const void* z1, const void* z2, const float* const* stats,
                                 const float* mask, float* sums, int act, int N, int C, int HW, void* stream);
int ppea_bn_bwd_reduce_final_bf16(const void* dy, const void* z1, const void* z2, const float* const* stats,
                                  const float* mask, float* sums, int act, int N, int C, int HW, void* stream);
/* SyncBatchNorm across ranks (one packed all-gather per BN forward, trainer.py:215-222 + get_bn rka.py:176-180):
 * local statistics in wire layout packed[2C+1] = mean[C] | biased var[C] | count, and the Chan combine of the
 * gathered [world][2C+1] table into mean / invstd (running statistics updated unless running_mean is NULL). */
int ppea_bn_finalize_packed_f32(const float* partial, int N, int C, int HW, float* packed, void* stream);
/* same wire layout straight from the tensor in one launch (small channels; PPEA_ERR_UNSUPPORTED otherwise) */
int ppea_bn_stats_packed_f32(const void* z, int N, int C, int HW, float* packed, void* stream);
int ppea_bn_stats_packed_bf16(const void* z, int N, int C, int HW, float* packed, void* stream);
int ppea_bn_sync_combine_f32(const float* gathered, int world, int C, float eps, float momentum, float* mean,
                             float* invstd, float* running_mean, float* running_var, void* stream);
/* One launch per BN and direction for small channels (N * HW <= 16384, HW % 8 == 0, C >= 64): the workgroup that
 * owns a channel keeps its values in registers, so statistics + apply (forward) and reduce + apply (backward) are one
 * kernel each.  prm = {gamma1, beta1, gamma2, beta2}; out = {running_mean1, running_var1, running_mean2, running_var2
 * (NULL: no update), mean1, invstd1, mean2, invstd2 (written)}; `sums` [3][C] as ppea_bn_bwd_reduce_final_*.
 * Same semantics as rka.py:182-197, 232-239, 283-289, 315-326 (training-mode BN + act + DropPath + residuals). */
int ppea_bn_fwd_channel_f32(const void* z1, const void* z2, const float* const* prm, float* const* out, float eps,
                            float momentum, const float* mask, const void* r1, const void* r2, float r2_scale, void* y,
                            int act, int N, int C, int HW, void* stream);
int ppea_bn_fwd_channel_bf16(const void* z1, const void* z2, const float* const* prm, float* const* out, float eps,
                             float momentum, const float* mask, const void* r1, const void* r2, float r2_scale, void* y,
                             int act, int N, int C, int HW, void* stream);
/* The same forward for ONE BatchNorm whose statistics come from the producing GEMM's epilogue (partial [C][P][2] from
 * ppea_pwconv_stats_bf16): no reduction over the activation, no workgroup barrier.  prm = {gamma, beta}; out =
 * {running_mean, running_var (NULL: no update), mean, invstd}.  Backward: ppea_bn_bwd_channel_*. */
int ppea_bn_fwd_channel_sums_f32(const void* z, const float* partial, int P, const float* const* prm, float* const* out,
                                 float eps, float momentum, const float* mask, const void* r1, const void* r2, float r2_scale,
                                 void* y, int act, int N, int C, int HW, void* stream);
int ppea_bn_fwd_channel_sums_bf16(const void* z, const float* partial, int P, const float* const* prm, float* const* out,
                                  float eps, float momentum, const float* mask, const void* r1, const void* r2, float r2_scale,
                                  void* y, int act, int N, int C, int HW, void* stream);
/* acc (or NULL): a gradient that reaches z1 through another consumer (the block's residual connection, rka.py:289, 326);
 * dz1 = BN-backward(dy) + acc in the same launch instead of a separate element-wise add. */
int ppea_bn_bwd_channel_f32(const void* dy, const void* z1, const void* z2, const float* const* stats, const float* mask,
                            float inv_count, const void* acc, void* dz1, void* dz2, float* sums, int act, int N, int C,
                            int HW, void* stream);
int ppea_bn_bwd_channel_bf16(const void* dy, const void* z1, const void* z2, const float* const* stats, const float* mask,
                             float inv_count, const void* acc, void* dz1, void* dz2, float* sums, int act, int N, int C,
                             int HW, void* stream);
/* An output with TWO consumers (RepLKBlock / ConvFFN: the first 1x1 conv and the adapter both read the block's first
 * BatchNorm, replknet_adapter.py:283-289, 315-326): dyb / dy2b = the second consumer's gradient.  The kernels start from
 * round(dy + dyb) -- what a separate element-wise add would have stored -- so results are bit-identical to that form. */
int ppea_bn_bwd_channel_dup_f32(const void* dy, const void* dyb, const void* z1, const void* z2, const float* const* stats,
                                const float* mask, float inv_count, const void* acc, void* dz1, void* dz2, float* sums,
                                int act, int N, int C, int HW, void* stream);
int ppea_bn_bwd_channel_dup_bf16(const void* dy, const void* dyb, const void* z1, const void* z2, const float* const* stats,
                                 const float* mask, float inv_count, const void* acc, void* dz1, void* dz2, float* sums,
                                 int act, int N, int C, int HW, void* stream);
int ppea_bn_bwd_channel_next_dup_f32(const void* dy2, const void* dy2b, const void* dskip, const void* z, const void* y,
                                     const float* const* stats, const float* mask, float inv_count, void* dz, void* dy,
                                     float* sums, int N, int C, int HW, void* stream);
int ppea_bn_bwd_channel_next_dup_bf16(const void* dy2, const void* dy2b, const void* dskip, const void* z, const void* y,
                                      const float* const* stats, const float* mask, float inv_count, void* dz, void* dy,
                                      float* sums, int N, int C, int HW, void* stream);
/* Several ranks (reduce -> all-reduce -> apply): the reduce launch merges the two gradients and stores dym = round(dy + dyb)
 * for the apply launch.  Limits of ppea_bn_bwd_reduce_final_*. */
int ppea_bn_bwd_reduce_final_dup_f32(const void* dy, const void* dyb, void* dym, const void* z1, const void* z2,
                                     const float* const* stats, const float* mask, float* sums, int act, int N, int C,
                                     int HW, void* stream);
int ppea_bn_bwd_reduce_final_dup_bf16(const void* dy, const void* dyb, void* dym, const void* z1, const void* z2,
                                      const float* const* stats, const float* mask, float* sums, int act, int N, int C,
                                      int HW, void* stream);
/* End of one block and the first BatchNorm of the next in one launch per direction (replknet_adapter.py:283-289,
 * 315-326 followed by the next block's prelkb_bn / preffn_bn, :281, :312): y = mask * BN_A(z) + r1 + r2_scale * r2,
 * y2 = BN_B(y) with BN_B's statistics taken of the stored y.  prm = {gammaA, betaA, gammaB, betaB}; out = {running_meanA,
 * running_varA, running_meanB, running_varB (NULL: no update), meanA, invstdA, meanB, invstdB (written)}.  Backward:
 * stats = {meanA, invstdA, gammaA, betaA, meanB, invstdB, gammaB, betaB}; dskip = gradient of y from its other consumer
 * (NULL: none); writes dz, dy (total gradient of y) and sums [4][C] = dbetaA | dgammaA | dbetaB | dgammaB.
 * Results are bit-identical to ppea_bn_fwd_channel_* / ppea_bn_bwd_channel_* applied twice. */
int ppea_bn_fwd_channel_next_f32(const void* z, const float* const* prm, float* const* out, float eps, float momentum,
                                 const float* mask, const void* r1, const void* r2, float r2_scale, void* y, void* y2,
                                 int N, int C, int HW, void* stream);
int ppea_bn_fwd_channel_next_bf16(const void* z, const float* const* prm, float* const* out, float eps, float momentum,
                                  const float* mask, const void* r1, const void* r2, float r2_scale, void* y, void* y2,
                                  int N, int C, int HW, void* stream);
int ppea_bn_bwd_channel_next_f32(const void* dy2, const void* dskip, const void* z, const void* y, const float* const* stats,
                                 const float* mask, float inv_count, void* dz, void* dy, float* sums, int N, int C, int HW,
                                 void* stream);
int ppea_bn_bwd_channel_next_bf16(const void* dy2, const void* dskip, const void* z, const void* y, const float* const* stats,
                                  const float* mask, float inv_count, void* dz, void* dy, float* sums, int N, int C, int HW,
                                  void* stream);
int ppea_bn_apply_f32(const void* z1, const void* z2, const float* const* stats, const float* mask,
                      const void* r1, const void* r2, float r2_scale, void* y, int act,
                      int N, int C, int HW, void* stream);
int ppea_bn_apply_bf16(const void* z1, const void* z2, const float* const* stats, const float* mask,
                       const void* r1, const void* r2, float r2_scale, void* y, int act,
                       int N, int C, int HW, void* stream);
int ppea_bn_bwd_reduce_f32(const void* dy, const void* z1, const void* z2, const float* const* stats,
                           const float* mask, float* partial, int act, int N, int C, int HW, void* stream);
int ppea_bn_bwd_reduce_bf16(const void* dy, const void* z1, const void* z2, const float* const* stats,
                            const float* mask, float* partial, int act, int N, int C, int HW, void* stream);
int ppea_bn_bwd_finalize_f32(const float* partial, int N, int C, float* sums, void* stream);
int ppea_bn_bwd_apply_f32(const void* dy, const void* z1, const void* z2, const float* const* stats,
                          const float* mask, const float* sums, float inv_count, void* dz1, void* dz2,
                          int act, int N, int C, int HW, void* stream);
int ppea_bn_bwd_apply_bf16(const void* dy, const void* z1, const void* z2, const float* const* stats,
                           const float* mask, const float* sums, float inv_count, void* dz1, void* dz2,
                           int act, int N, int C, int HW, void* stream);
/* ------------------------------------------------------------------------------------------
 * A2 across ranks  SyncBatchNorm on the fused kernels (csrc/bn_sync.hip).  Both encoders are nn.SyncBatchNorm in the
 *   reference (replknet_adapter.py:170-180; DDP step trainer.py:215-222): statistics of the GLOBAL batch.  One BatchNorm
 *   (+ act + DropPath + residual + adapter add) = TWO launches around ONE collective issued by the host:
 *     ppea_bn_sync_stats_* (or ppea_bn_sync_stats_from_sums_f32 after a GEMM that left partial sums)
 *         -> packed = mean1[C] | biased var1[C] [| mean2[C] | var2[C]] | count        (pitch 2C+1, or 4C+1 with z2)
 *     all-gather -> gathered [world][pitch]
 *     ppea_bn_sync_apply_*: Chan combine of the gathered rows inside the apply kernel (no combine launch), running
 *         statistics (momentum, unbiased variance of the global batch), out[4..7] = saved mean / invstd, y as
 *         ppea_bn_fwd_channel_*; packed_next (or NULL; needs N * HW <= 16384, C >= 64): the LOCAL statistics of the stored
 *         y in wire format (pitch 2C+1) -- the next BatchNorm over y (replknet_adapter.py:281, 312) then needs no
 *         statistics launch.  HW % 8 == 0, else PPEA_ERR_UNSUPPORTED (ppea_bn_sync_combine_f32 + ppea_bn_apply_*).
 *   prm / out as ppea_bn_fwd_channel_*.  ws: ppea_bn_sync_stats_workspace_bytes(N, C, HW, z2 != NULL) bytes (0 when one
 *   workgroup covers a channel).  Backward: ppea_bn_bwd_reduce(_final)_* -> all-reduce (sum) of sums [3][C] ->
 *   ppea_bn_sync_bwd_apply_* with inv_count = 1 / (global count); acc (shape of z1, or NULL): dz1 = round(dz1) + acc, the
 *   gradient reaching z1 through its other consumer (the block's residual connection, rka.py:289, 326); dgb [3][C] (or
 *   NULL) = sums * gscale: d beta | d gamma1 | d gamma2 with gscale = 1 / world -- the reference's DDP averages the
 *   per-rank LOCAL sums torch's SyncBatchNorm returns (trainer.py:215-222), which is (global sum) / world.
 * ---------------------------------------------------------------------------------------- */
long ppea_bn_sync_stats_workspace_bytes(int N, int C, int HW, int two);
int ppea_bn_sync_stats_f32(const void* z1, const void* z2, float* packed, float* ws, int N, int C, int HW, void* stream);
int ppea_bn_sync_stats_bf16(const void* z1, const void* z2, float* packed, float* ws, int N, int C, int HW, void* stream);
int ppea_bn_sync_stats_from_sums_f32(const float* sums, int P, int C, long count, float* packed, void* stream);
int ppea_bn_sync_apply_f32(const void* z1, const void* z2, const float* gathered, int world, const float* const* prm,
                           float* const* out, float eps, float momentum, const float* mask, const void* r1, const void* r2,
                           float r2_scale, void* y, float* packed_next, int act, int N, int C, int HW, void* stream);
int ppea_bn_sync_apply_bf16(const void* z1, const void* z2, const float* gathered, int world, const float* const* prm,
                            float* const* out, float eps, float momentum, const float* mask, const void* r1, const void* r2,
                            float r2_scale, void* y, float* packed_next, int act, int N, int C, int HW, void* stream);
int ppea_bn_sync_bwd_apply_f32(const void* dy, const void* z1, const void* z2, const float* const* stats,
                               const float* mask, const float* sums, float inv_count, const void* acc, void* dz1, void* dz2,
                               float* dgb, float gscale, int act, int N, int C, int HW, void* stream);
int ppea_bn_sync_bwd_apply_bf16(const void* dy, const void* z1, const void* z2, const float* const* stats,
                                const float* mask, const float* sums, float inv_count, const void* acc, void* dz1, void* dz2,
                                float* dgb, float gscale, int act, int N, int C, int HW, void* stream);

/* ------------------------------------------------------------------------------------------
 * A12 glue  nn.ReflectionPad2d(1) of the decoder's Conv3x3 (layers.py:119-135).  in [planes,H,W] ->
 *     out [planes,H+2,W+2]; backward is a gather (no atomics): din [planes,H,W] from dout.  H, W >= 3.
 * ---------------------------------------------------------------------------------------- */
int ppea_reflect_pad1_fwd_f32(const void* in, void* out, long planes, int H, int W, void* stream);
int ppea_reflect_pad1_fwd_bf16(const void* in, void* out, long planes, int H, int W, void* stream);
int ppea_reflect_pad1_bwd_f32(const void* dout, void* din, long planes, int H, int W, void* stream);
int ppea_reflect_pad1_bwd_bf16(const void* dout, void* din, long planes, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------
 * A18+A19  BackprojectDepth -> Project3D fused (layers.py:138-199; trainer.py:904-907).
 *     depth [B,1,H,W]; inv_K [B,4,4] (only [:3,:3] read); P [B,3,4] = (K @ T)[:, :3, :];
 *     grid [B,H,W,2] normalised to [-1,1] (x then y); eps added to z (1e-7).
 *     bwd: d_depth [B,1,H,W] and dP [B,3,4] overwritten; dP is summed from per-block partials in a caller-owned
 *     workspace of ppea_backproject_project_bwd_workspace_bytes(B, H, W) bytes in a FIXED order (no float atomics:
 *     the pose gradient is bitwise reproducible).
 * ---------------------------------------------------------------------------------------- */
int ppea_backproject_project_fwd_f32(const float* depth, const float* inv_K, const float* P,
                                     float* grid, int B, int H, int W, float eps, void* stream);
long ppea_backproject_project_bwd_workspace_bytes(int B, int H, int W);
int ppea_backproject_project_bwd_f32(const float* depth, const float* inv_K, const float* P,
                                     const float* d_grid, float* d_depth, float* dP, void* workspace,
                                     int B, int H, int W, float eps, void* stream);

/* ------------------------------------------------------------------------------------------
 * A20  F.grid_sample(bilinear, align_corners=True)  (trainer.py:911-914 border;
 *      replk_matching_adapter.py:299 zeros).  src [B,C,Hi,Wi]; grid [B,Ho,Wo,2];
 *      out [B,C,Ho,Wo].  padding: 0 = zeros, 1 = border.
 *      bwd_grid: gradient w.r.t. the grid only (the source is data).
 * ---------------------------------------------------------------------------------------- */
int ppea_grid_sample_fwd_f32(const float* src, const float* grid, float* out,
                             int B, int C, int Hi, int Wi, int Ho, int Wo, int padding,
                             void* stream);
int ppea_grid_sample_bwd_grid_f32(const float* src, const float* grid, const float* d_out,
                                  float* d_grid, int B, int C, int Hi, int Wi, int Ho, int Wo,
                                  int padding, void* stream);

/* A15: axis-angle + translation -> 4x4 transformation (layers.py:26-42 `transformation_from_parameters`, 61-100
 * `rot_from_axisangle`): aa, tr [B][3] fp32 -> T [B][4][4]; invert != 0: R^T T(-t).  Backward: d aa, d tr from dT. */
int ppea_pose_matrix_fwd_f32(const float* aa, const float* tr, float* T, int B, int invert, void* stream);
int ppea_pose_matrix_bwd_f32(const float* aa, const float* tr, const float* dT, float* daa, float* dtr, int B, int invert,
                             void* stream);

/* ------------------------------------------------------------------------------------------
 * A21+A22  compute_reprojection_loss (trainer.py:995-1007; SSIM layers.py:226-257):
 *      out[b,0,i,j] = alpha * mean_c SSIM(pred,target) + (1-alpha) * mean_c |target-pred|.
 *      pred/target [B,C,H,W] (H,W >= 2); out has batch stride `out_bstride` elements so the
 *      result can be written straight into one channel of a [B,2,H,W] buffer.
 *      bwd: gradient w.r.t. pred (d_out has batch stride dout_bstride).
 * ---------------------------------------------------------------------------------------- */
int ppea_ssim_l1_fwd_f32(const float* pred, const float* target, float* out, long out_bstride,
                         int B, int C, int H, int W, float alpha, void* stream);
int ppea_ssim_l1_bwd_f32(const float* pred, const float* target, const float* d_out,
                         long dout_bstride, float* d_pred, int B, int C, int H, int W,
                         float alpha, void* stream);

/* ------------------------------------------------------------------------------------------
 * A23  get_smooth_loss (layers.py:210-223).  disp [B,1,H,W], img [B,C,H,W].
 *      partials [ppea_smooth_num_partials()][2]: per-workgroup partial sums (deterministic, no
 *      atomics); column 0 = sum |dx disp| e^{-mean_c|dx img|}, column 1 = same for dy.  The caller
 *      adds the rows and divides by the element counts B*H*(W-1) and B*(H-1)*W.
 *      bwd: d_disp = gx * d(sum_x)/d(disp) + gy * d(sum_y)/d(disp) with scalar weights.
 * ---------------------------------------------------------------------------------------- */
int ppea_smooth_num_partials(void);
int ppea_smooth_fwd_f32(const float* disp, const float* img, float* partials,
                        int B, int C, int H, int W, void* stream);
int ppea_smooth_bwd_f32(const float* disp, const float* img, float gx, float gy, float* d_disp,
                        int B, int C, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------
 * A24/A25  per-pixel loss selection (trainer.py:1069-1114): min over the two source frames,
 *      selec_reproj overwrite, identity min + tie-break noise, automask argmin.
 *      reproj, identity [B,2,H,W]; warped_m1/p1 [B,3,H,W]; noise [B,1,H,W] (already * 1e-5) or
 *      NULL.  Outputs: sel [B,1,H,W] selected reprojection loss; src_idx uint8 [B,1,H,W]
 *      (0/1 = which of reproj's two channels feeds `sel`, 2 = forced zero);
 *      frame_idx int64 [B,1,H,W] (argmin over the two frames, first minimum);
 *      auto_idx int64 [B,1,H,W] (argmin over [sel, identity_min+noise], first minimum).
 * ---------------------------------------------------------------------------------------- */
int ppea_loss_select_f32(const float* reproj, const float* identity, const float* warped_m1,
                         const float* warped_p1, const float* noise, float* sel,
                         uint8_t* src_idx, int64_t* frame_idx, int64_t* auto_idx,
                         int B, int C, int H, int W, int selec_reproj, void* stream);

/* Tail of compute_losses (trainer.py:1092-1139) after the per-pixel selection: mask (automask argmin == 0, or for the
 * multi-frame pass consistency_mask * (1 - augmentation_mask)), rl = sum(sel * mask) / (sum(mask) + 1e-7), and for the
 * multi-frame pass the consistency loss mean(|multi - mono| * (1 - mask)) and its target.  One pass + finalize forward, one
 * pass backward (d reproj [B][2][H][W] routed by the selection's source index, d multi_depth); fixed summation order.
 *   sel, mask, target, multi, mono [B][1][H][W] fp32; auto_idx int64 or NULL; cons [B][H][W] or NULL; aug [B] or NULL;
 *   partial: ppea_loss_tail_blocks(B*H*W) * 3 floats; out[0] = rl, out[1] = consistency loss, out[2] = 1/(sum(mask)+1e-7);
 *   g_rl / g_con: device scalars (gradients of out[0] / out[1]) or NULL. */
int ppea_loss_tail_blocks(long total);
int ppea_loss_tail_fwd_f32(const float* sel, const int64_t* auto_idx, const float* cons, const float* aug, const float* multi,
                           const float* mono, float* mask, float* target, float* partial, float* out, int B, int H, int W,
                           int is_multi, void* stream);
int ppea_loss_tail_bwd_f32(const float* mask, const uint8_t* src, const float* out, const float* g_rl, const float* g_con,
                           const float* multi, const float* mono, float* d_reproj, float* d_multi, int B, int H, int W,
                           void* stream);

/* ------------------------------------------------------------------------------------------
 * A12 / A14  Dense convolutions as implicit GEMMs on the matrix cores (csrc/conv_nhwc.hip, conv_wgrad.hip):
 *     bf16 channels-last activations, fp32 accumulation.  Replaces the library convolutions behind
 *       ConvBlock / Conv3x3            layers.py:103-135 (decoder: networks/depth_decoder_v2.py:172-245)
 *       ResNet-18 pose trunk          networks/resnet_encoder.py:25-72, 397-409
 *       PoseDecoder                   networks/pose_decoder.py:33-52
 *       reduce_conv                   networks/replk_matching_adapter.py:127-131
 *       RepLKNet stem[0]              networks/replknet_adapter.py:411
 *     forward and data gradient (one kernel: the data gradient runs it with the flipped / transposed operand image
 *     and, for a strided forward, over the zero-dilated output gradient) and weight gradient.
 *
 *   ppea_conv_packed_bytes(Cout, Cin, R, S, flip)   bytes of the operand image of w [Cout][Cin][R][S]
 *   ppea_conv_pack_weights(w, w_is_bf16, packed, Cout, Cin, R, S, flip)
 *        flip = 0: [R*S][Cout][Cin padded to 32] (forward); flip = 1: [R*S reversed][Cin][Cout padded to 32] (dgrad)
 *   ppea_conv_nhwc_bf16   y = act(bias + conv(x)):  x [N][H][W][Cin] (Cin % 8 == 0), y [N][Ho][Wo][Cout] or, with
 *        out_nchw, [N][Cout][Ho][Wo]; R, S <= 7; stride 1 | 2; pad = zero padding, or reflect != 0: reflection
 *        padding of `pad` <= 1 read through reflected indices (no padded copy); dil > 1: x is read as its zero-dilated
 *        image (data gradient of a strided conv); bias fp32 or (bias_bf16) bf16 or NULL; act 0 none, 1 ReLU, 2 ELU,
 *        3 sigmoid.
 *   ppea_conv_wgrad_nhwc_bf16   dw [Cout][Cin][R][S] (fp32 or bf16) = sum_pixels dz (x) x, split over output patches
 *        with fp32 partials in a caller-owned workspace of ppea_conv_wgrad_workspace_bytes(...) bytes
 *        (deterministic; Cin % 8 == 0, Cout % 8 == 0).
 *   ppea_image_to_nhwc_bf16   fp32 NCHW frames -> bf16 channels-last, channels zero-padded to Cp (% 8 == 0),
 *        y = (x - sub) / div  (resnet_encoder.py:399 normalisation for the pose trunk; sub 0, div 1 for stem[0]).
 * ---------------------------------------------------------------------------------------- */
long ppea_conv_packed_bytes(int Cout, int Cin, int R, int S, int flip);
int ppea_conv_pack_weights(const void* w, int w_is_bf16, void* packed, int Cout, int Cin, int R, int S, int flip,
                           void* stream);
int ppea_conv_nhwc_bf16(const void* x, const void* w_packed, const void* bias, int bias_bf16, void* y,
                        int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int reflect, int dil,
                        int Ho, int Wo, int act, int out_nchw, void* stream);
long ppea_conv_wgrad_workspace_bytes(int N, int Cin, int Cout, int R, int S, int stride, int Ho, int Wo);
int ppea_conv_wgrad_nhwc_bf16(const void* dz, const void* x, void* dw, int dw_bf16, void* workspace,
                              int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int reflect,
                              int Ho, int Wo, void* stream);
int ppea_image_to_nhwc_bf16(const float* x, void* y, int N, int C, int H, int W, int Cp, float sub, float div,
                            void* stream);

/* fp32 dense convolutions / linear layers on the fp32 matrix cores (csrc/conv_f32.hip, v_mfma_f32_32x32x2_f32): the
 * fp32 (parity, BASELINE config 1) step's replacement for every library convolution and GEMM behind
 *   RepLKBlock / ConvFFN 1x1 convs, B_Adapter / Adapter     networks/replknet_adapter.py:20-109, 264-326
 *   ConvBlock / Conv3x3 / ConvTranspose2d (Stage 2)          layers.py:103-135, networks/depth_decoder_v2.py:172-245
 *   ResNet-18 pose trunk, PoseDecoder                        networks/resnet_encoder.py:367-409, networks/pose_decoder.py:27-52
 *   reduce_conv, RepLKNet stem[0]                            networks/replk_matching_adapter.py:127-131, replknet_adapter.py:411
 * One implicit-GEMM family for every kernel size, stride, zero padding and memory format: x / y / dy / dx are addressed
 * through four ELEMENT strides (n, c, h, w) -- NCHW, channels_last, or a [tokens][features] matrix viewed as
 * [1][features][tokens][1] (nn.Linear).  w / dw: [Cout][Cin][R][S] contiguous.  Ho = (H + 2 pad - R) / stride + 1.
 *   ppea_conv2d_f32_fwd     y = bias + conv(x)  (bias [Cout] or NULL)
 *   ppea_conv2d_f32_dgrad   dx (every element written) from dy
 *   ppea_conv2d_f32_wgrad   dw = sum over pixels, split over the grid with fp32 slabs in a caller-owned workspace of
 *                           ppea_conv2d_f32_wgrad_workspace_bytes(...) bytes summed in a fixed order (no atomics)
 * ---------------------------------------------------------------------------------------- */
int ppea_conv2d_f32_fwd(const float* x, const long* xs, const float* w, const float* bias, float* y, const long* ys, int N,
                        int Cin, int H, int W, int Cout, int R, int S, int stride, int pad, void* stream);
int ppea_conv2d_f32_dgrad(const float* dy, const long* dys, const float* w, float* dx, const long* dxs, int N, int Cin, int H,
                          int W, int Cout, int R, int S, int stride, int pad, int Ho, int Wo, void* stream);
long ppea_conv2d_f32_wgrad_workspace_bytes(int N, int Cin, int Cout, int R, int S, int Ho, int Wo);
int ppea_conv2d_f32_wgrad(const float* x, const long* xs, const float* dy, const long* dys, float* dw, float* workspace, int N,
                          int Cin, int H, int W, int Cout, int R, int S, int stride, int pad, int Ho, int Wo, void* stream);
/* The same family for bf16 weights / activations / results (widened as they are staged, fp32 products and sums on the same
 * instruction, one rounding at the store; bias and dw stay fp32): what the bf16 step uses for shapes the layout-specialised
 * bf16 kernels do not take (map widths that are not a multiple of 4 pixels, channel counts that are not multiples of
 * 8 / 32: reduced-size test configurations), so that no shape of either dtype reaches a library convolution or GEMM. */
int ppea_conv2d_bf16_fwd(const void* x, const long* xs, const void* w, const float* bias, void* y, const long* ys, int N,
                         int Cin, int H, int W, int Cout, int R, int S, int stride, int pad, void* stream);
int ppea_conv2d_bf16_dgrad(const void* dy, const long* dys, const void* w, void* dx, const long* dxs, int N, int Cin, int H,
                           int W, int Cout, int R, int S, int stride, int pad, int Ho, int Wo, void* stream);
int ppea_conv2d_bf16_wgrad(const void* x, const long* xs, const void* dy, const long* dys, float* dw, float* workspace, int N,
                           int Cin, int H, int W, int Cout, int R, int S, int stride, int pad, int Ho, int Wo, void* stream);

/* MaxPool2d(3, 2, 1) of the pose ResNet-18 (networks/resnet_encoder.py:376-392) on channels-last tensors
 * (csrc/nhwc_pool.hip): forward keeps a one-byte window index (3 r + s, torch's tie rule: first maximum in scan order), the
 * backward gathers (no atomics).  x [N][H][W][C], C % 8 == 0; y, idx [N][Ho][Wo][C] with Ho = (H - 1) / 2 + 1. */
int ppea_nhwc_maxpool3x3s2_fwd_f32(const void* x, void* y, void* idx, int N, int H, int W, int C, void* stream);
int ppea_nhwc_maxpool3x3s2_fwd_bf16(const void* x, void* y, void* idx, int N, int H, int W, int C, void* stream);
int ppea_nhwc_maxpool3x3s2_bwd_f32(const void* dy, const void* idx, void* dx, int N, int H, int W, int C, void* stream);
int ppea_nhwc_maxpool3x3s2_bwd_bf16(const void* dy, const void* idx, void* dx, int N, int H, int W, int C, void* stream);

/* Image-fed convolutions (csrc/conv_image.hip): RepLKNet stem[0] (networks/replknet_adapter.py:411, 3x3 stride 2) and
 * the pose ResNet-18 conv1 (networks/resnet_encoder.py:376-388, 7x7 stride 2).  The frame is channels-last bf16 with its
 * 3 / 6 channels zero-padded to 8 (ppea_image_to_nhwc_bf16); one MFMA contraction covers a whole filter ROW (S taps x 8
 * channels are contiguous in memory), weights packed [K][Cout][K*8 padded to 32].  Forward and weight gradient
 * (dw [Cout][Cin][K][K], Cin <= 8 real channels); no data gradient (the input is the frame). */
long ppea_conv_image_packed_bytes(int Cout, int K);
int ppea_conv_image_pack_weights(const void* w, int w_is_bf16, void* packed, int Cout, int Cin, int K, void* stream);
int ppea_conv_image_bf16(const void* x, const void* w_packed, void* y, int N, int H, int W, int Cout, int K, int stride,
                         int pad, int Ho, int Wo, int out_nchw, void* stream);
long ppea_conv_image_wgrad_workspace_bytes(int N, int Cout, int K, int Ho, int Wo);
int ppea_conv_image_wgrad_bf16(const void* dz, const void* x, void* dw, int dw_bf16, void* workspace, int N, int H, int W,
                               int Cin, int Cout, int K, int stride, int pad, int Ho, int Wo, void* stream);

/* ------------------------------------------------------------------------------------------
 * A9  match_features (replk_matching_adapter.py:261-340), one lookup frame per item.
 *      cur, lookup [B,C,h,w]; P [B,3,4] = (K @ T)[:, :3, :] at the matching scale;
 *      inv_K [B,4,4]; bins [D]; skip [B] int32 (non-zero = lookup pose was zeroed, item skipped
 *      -> cost 0).  cost [B,D,h,w] = mean_c|warp(lookup) - cur| * edge_mask / (cnt + 1e-7).
 * A10 cost_volume_reduce (:446-456, :372-387): missing -> per-pixel max; confidence mask;
 *      argmin over bins after 0 -> 100 (int64, first minimum); lowest = 1/bins[argmin];
 *      cost_out = filled cost * confidence.
 * ---------------------------------------------------------------------------------------- */
int ppea_cost_volume_fwd_f32(const float* cur, const float* lookup, const float* P,
                             const float* inv_K, const float* bins, const int32_t* skip,
                             float* cost, int B, int C, int h, int w, int D, float eps,
                             void* stream);
/* bf16 features (the bf16 step): the kernel is bound by the bytes crossing the L1, so channel PAIRS are packed into dwords
 * first (`pairs`: caller-owned workspace of 2 * B * C/2 * h * w uint32) and a corner load serves two channels.  Bit-identical
 * to ppea_cost_volume_fwd_f32 on the features widened to fp32.  C even. */
int ppea_cost_volume_fwd_bf16(const void* cur, const void* lookup, void* pairs, const float* P, const float* inv_K,
                              const float* bins, const int32_t* skip, float* cost, int B, int C, int h, int w, int D,
                              float eps, void* stream);
int ppea_cost_volume_reduce_f32(const float* cost, const float* bins, float* cost_out,
                                float* confidence, int64_t* argmin, float* lowest,
                                int B, int D, int h, int w, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training-mode BatchNorm (+ReLU, + residual added before the activation) on channels_last data: the ResNet-18 pose
 * trunk (networks/resnet_encoder.py:25-72).  The tensor is G consecutive sub-batches [G][P][C] (C contiguous,
 * C / 8 a power of two <= 256), each normalised with its own statistics (per-pair statistics of a 2B pose batch).
 *   forward : stats -> partial[G][slabs][2][C]; finalize -> stats[G][3][C] (mean | invstd | unbiased var),
 *             ab[G][2][C] (y = a x + b), running statistics updated once per sub-batch in order;
 *             apply -> y = act(a x + b + res), act 0 none / 1 ReLU
 *   backward: bwd_reduce -> partial; bwd_finalize -> k[G][2][C], dgamma_dbeta[2][C]; bwd_apply -> dx, dres
 * ---------------------------------------------------------------------------------------- */
int ppea_nhwc_bn_slabs(int P, int C);
int ppea_nhwc_bn_stats_f32(const void* x, float* partial, int P, int C, int G, void* stream);
int ppea_nhwc_bn_stats_bf16(const void* x, float* partial, int P, int C, int G, void* stream);
int ppea_nhwc_bn_finalize_f32(const float* partial, int P, int C, int G, const float* gamma, const float* beta,
                              float eps, float momentum, float* stats, float* ab, float* running_mean,
                              float* running_var, void* stream);
int ppea_nhwc_bn_apply_f32(const void* x, const void* res, const float* ab, void* y, int P, int C, int G, int act,
                           void* stream);
int ppea_nhwc_bn_apply_bf16(const void* x, const void* res, const float* ab, void* y, int P, int C, int G, int act,
                            void* stream);
int ppea_nhwc_bn_bwd_reduce_f32(const void* x, const void* dy, const void* res, const float* stats, const float* ab,
                                float* partial, int P, int C, int G, int act, void* stream);
int ppea_nhwc_bn_bwd_reduce_bf16(const void* x, const void* dy, const void* res, const float* stats, const float* ab,
                                 float* partial, int P, int C, int G, int act, void* stream);
int ppea_nhwc_bn_bwd_finalize_f32(const float* partial, int P, int C, int G, float* k, float* dgamma_dbeta,
                                  void* stream);
int ppea_nhwc_bn_bwd_apply_f32(const void* x, const void* dy, const void* res, const float* stats, const float* ab,
                               const float* k, void* dx, void* dres, int P, int C, int G, int act, void* stream);
int ppea_nhwc_bn_bwd_apply_bf16(const void* x, const void* dy, const void* res, const float* stats, const float* ab,
                                const float* k, void* dx, void* dres, int P, int C, int G, int act, void* stream);

/* ------------------------------------------------------------------------------------------
 * Bias + ELU of the decoders' ConvBlock (layers.py:103-116), NCHW: y = elu(z + bias[c]); backward dz = dy * elu'
 * (from the saved OUTPUT: elu' = 1 for y > 0 else y + 1) and partial[N*C][chunks] = per-plane-chunk sums of dz
 * (chunks = ppea_bias_elu_chunks(N, C, HW)); the bias gradient is their sum over N and chunks.
 * ---------------------------------------------------------------------------------------- */
int ppea_bias_elu_chunks(int N, int C, int HW);
int ppea_bias_elu_fwd_f32(const void* z, const void* bias, int bias_bf16, void* y, int N, int C, int HW, void* stream);
int ppea_bias_elu_fwd_bf16(const void* z, const void* bias, int bias_bf16, void* y, int N, int C, int HW, void* stream);
int ppea_bias_elu_bwd_f32(const void* dy, const void* y, void* dz, float* partial, int N, int C, int HW, void* stream);
int ppea_bias_elu_bwd_bf16(const void* dy, const void* y, void* dz, float* partial, int N, int C, int HW, void* stream);

/* Nearest 2x upsampling fused with the skip concatenation of the depth decoder (networks/depth_decoder_v2.py:231-236;
 * layers.py:204-207), channels-last: a [N][H/2][W/2][C1], b [N][H][W][C2] (NULL, C2 = 0: no skip) -> out [N][H][W][C1+C2];
 * backward: dout -> da (2x2 block sums accumulated in fp32), db.  H, W = OUTPUT size (even); C1, C2 % 8 == 0. */
int ppea_nhwc_up2cat_fwd_f32(const void* a, const void* b, void* out, int N, int H, int W, int C1, int C2, void* stream);
int ppea_nhwc_up2cat_fwd_bf16(const void* a, const void* b, void* out, int N, int H, int W, int C1, int C2, void* stream);
int ppea_nhwc_up2cat_bwd_f32(const void* dout, void* da, void* db, int N, int H, int W, int C1, int C2, void* stream);
int ppea_nhwc_up2cat_bwd_bf16(const void* dout, void* da, void* db, int N, int H, int W, int C1, int C2, void* stream);

/* channels_last versions of the decoder passes (x [B][H][W][C], C % 8 == 0; bias_elu: C / 8 a power of two <= 256):
 * the decoders hand the library's NHWC-native convolutions NHWC activations. */
int ppea_nhwc_reflect_pad1_fwd_f32(const void* x, void* out, int B, int H, int W, int C, void* stream);
int ppea_nhwc_reflect_pad1_fwd_bf16(const void* x, void* out, int B, int H, int W, int C, void* stream);
int ppea_nhwc_reflect_pad1_bwd_f32(const void* dout, void* dx, int B, int H, int W, int C, void* stream);
int ppea_nhwc_reflect_pad1_bwd_bf16(const void* dout, void* dx, int B, int H, int W, int C, void* stream);
int ppea_nhwc_bias_elu_slabs(int P, int C);
int ppea_nhwc_bias_elu_fwd_f32(const void* z, const void* bias, int bias_bf16, void* y, int P, int C, void* stream);
int ppea_nhwc_bias_elu_fwd_bf16(const void* z, const void* bias, int bias_bf16, void* y, int P, int C, void* stream);
int ppea_nhwc_bias_elu_bwd_f32(const void* dy, const void* y, void* dz, float* partial, int P, int C, void* stream);
int ppea_nhwc_bias_elu_bwd_bf16(const void* dy, const void* y, void* dz, float* partial, int P, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer step (trainer.py:350, torch.optim.Adam with the reference's defaults) over ONE flat fp32 buffer
 * holding every trainable tensor: p, g, m, v fp32 [n]; w16 (may be NULL) = bf16 working copy of p[0, n_lo);
 * state = device float[2] {step t >= 1, learning rate}.
 * ---------------------------------------------------------------------------------------- */
int ppea_adam_flat_f32(float* p, const float* g, float* m, float* v, void* w16, long n, long n_lo,
                       const float* state, float beta1, float beta2, float eps, void* stream);
/* The same step on g * grad_scale (data-parallel mean of rank-summed gradients without a scaling pass of its own). */
int ppea_adam_flat_scaled_f32(float* p, const float* g, float* m, float* v, void* w16, long n, long n_lo,
                              const float* state, float beta1, float beta2, float eps, float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PPEA_DEPTH_H */
