"""Autograd operators over the C ABI (include/ppea_depth.h).  Host glue only: every
forward/backward below is one (or two) kernel launches on the current HIP stream.

Reference call sites are cited per op (paths relative to /root/reference/ppeadepth/).
"""
import weakref

import torch

from . import _abi
from ._abi import call, ptr, stream_ptr

_F32 = torch.float32
_BF16 = torch.bfloat16

# bench.py sets this to a list; every k=31 launch then appends (kind, start_event, end_event)
# recorded on the launch stream (HIP events; the kernels run on torch's current stream).
PROFILE_DWCONV = None


PROFILE_REPLAY = {}       # kind -> closure re-issuing the first profiled launch of that kind (same arguments)


def _timed(kind, K, fn):
    if PROFILE_DWCONV is None or K != 31:
        return fn()
    PROFILE_REPLAY.setdefault(kind, fn)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = fn()
    e.record()
    PROFILE_DWCONV.append((kind, s, e))
    return r


def _suffix(t):
    if t.dtype == _F32:
        return "f32"
    if t.dtype == _BF16:
        return "bf16"
    raise _abi.PpeaKernelError(f"unsupported dtype {t.dtype}")


# ---------------------------------------------------------------------------------------------
# packed bf16 filter images for the MFMA depthwise kernel, cached per (weight storage, version, flip)
# ---------------------------------------------------------------------------------------------
_PACK_CACHE = {}
_MFMA_K = (31, 29, 27, 13)


def _packed_filter(w, flip):
    """uint8 buffer with the bf16 Toeplitz source image of w [C,1,K,K]; rebuilt when w changes in place.
    Keyed by the tensor object (weak reference), not by its address: a freed weight's address can be handed to
    another model's weight."""
    key = (id(w), int(flip))
    ver = w._version
    # A trainable filter (--fullft_reb) is updated by the flat Adam kernel through raw pointers, which does not
    # bump `_version`: pack it on every use (the pack launch is then part of the captured step as well).
    cacheable = not w.requires_grad
    hit = _PACK_CACHE.get(key) if cacheable else None
    if hit is not None and hit[3]() is w and hit[0] == ver and hit[2] == tuple(w.shape):
        return hit[1]
    C, K = w.shape[0], w.shape[-1]
    nbytes = _abi.lib.ppea_dwconv_lk_packed_bytes(C, K)
    buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    wf = w.detach().to(_F32).contiguous()
    call("ppea_dwconv_lk_pack_bf16", ptr(wf), ptr(buf), C, K, int(flip), stream_ptr())
    if cacheable:
        _PACK_CACHE[key] = (ver, buf, tuple(w.shape), weakref.ref(w, lambda _r, k=key: _PACK_CACHE.pop(k, None)))
    return buf


def _mfma_ok(x, K, KS):
    return x.dtype == _BF16 and K in _MFMA_K and KS in (0, 5)


# ---------------------------------------------------------------------------------------------
# A1  large-kernel depthwise conv (+ fused 5x5 branch)   networks/replknet_adapter.py:151-168, 232-239
# ---------------------------------------------------------------------------------------------
class _DwConvLK(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_big, w_small, want_sums=False):
        x = x.contiguous()
        N, C, H, W = x.shape
        K = w_big.shape[-1]
        KS = 0 if w_small is None else w_small.shape[-1]
        wb = w_big.detach().to(_F32).contiguous()
        ws = None if w_small is None else w_small.detach().to(_F32).contiguous()
        y_big = torch.empty_like(x)
        y_small = torch.empty_like(x) if KS else None
        ctx.packed = None
        done = False
        sums = None
        if _mfma_ok(x, K, KS):
            pb = _packed_filter(w_big, False)
            ps = _packed_filter(w_small, False) if KS else None
            P = _abi.lib.ppea_dwconv_lk_stats_partials(N, C, H, W, K, KS) if (want_sums and KS) else 0
            err = -1
            if P > 0:
                # per-channel partial sums of both outputs from the conv's epilogue: the BatchNorm pair after it needs
                # no statistics pass (batchnorm.fused_bn_act(..., sums=(sums[0], sums[1])))
                sums = torch.empty(2, C, P, 2, device=x.device, dtype=_F32)
                err = _timed("fwd31", K, lambda: _abi.lib.ppea_dwconv_lk_fwd_stats_bf16p(
                    ptr(x), ptr(pb), ptr(ps), ptr(y_big), ptr(y_small), ptr(sums), N, C, H, W, K, KS, stream_ptr()))
                if err != 0:
                    sums = None
            if sums is None:
                err = _timed("fwd31", K, lambda: _abi.lib.ppea_dwconv_lk_fwd_bf16p(
                    ptr(x), ptr(pb), ptr(ps), ptr(y_big), ptr(y_small), N, C, H, W, K, KS, stream_ptr()))
            if err == 0:
                done = True
                ctx.packed = (w_big, w_small)
            elif err != -1:
                _abi.check(err, "ppea_dwconv_lk_fwd_bf16p")
        if not done:
            _timed("fwd31", K, lambda: call(f"ppea_dwconv_lk_fwd_{_suffix(x)}", ptr(x), ptr(wb, _F32), ptr(ws),
                                             ptr(y_big), ptr(y_small), N, C, H, W, K, KS, stream_ptr()))
        ctx.save_for_backward(x, wb, ws)
        ctx.has_small = KS > 0
        ctx.w_dtypes = (w_big.dtype, None if w_small is None else w_small.dtype)
        if want_sums:
            if sums is not None:
                ctx.mark_non_differentiable(sums)
            return y_big, (y_small if KS else None), sums
        if KS:
            return y_big, y_small
        return y_big, None

    @staticmethod
    def backward(ctx, dy_big, dy_small, _dsums=None):
        x, wb, ws = ctx.saved_tensors
        N, C, H, W = x.shape
        K = wb.shape[-1]
        KS = ws.shape[-1] if ctx.has_small else 0
        dx = dwb = dws = None
        if dy_big is None:
            dy_big = torch.zeros_like(x)
        dy_big = dy_big.contiguous().to(x.dtype)
        if ctx.has_small:
            dy_small = (torch.zeros_like(x) if dy_small is None else dy_small.contiguous().to(x.dtype))
        else:
            dy_small = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            done = False
            if ctx.packed is not None:
                pb = _packed_filter(ctx.packed[0], True)
                ps = _packed_filter(ctx.packed[1], True) if KS else None
                err = _timed("bwd31", K, lambda: _abi.lib.ppea_dwconv_lk_bwd_data_bf16p(
                    ptr(dy_big), ptr(dy_small), ptr(pb), ptr(ps), ptr(dx), N, C, H, W, K, KS, stream_ptr()))
                if err == 0:
                    done = True
                elif err != -1:
                    _abi.check(err, "ppea_dwconv_lk_bwd_data_bf16p")
            if not done:
                _timed("bwd31", K, lambda: call(f"ppea_dwconv_lk_bwd_data_{_suffix(x)}", ptr(dy_big),
                                                 ptr(dy_small), ptr(wb), ptr(ws), ptr(dx), N, C, H, W, K, KS,
                                                 stream_ptr()))
        if ctx.needs_input_grad[1]:
            dwb = torch.empty_like(wb)
            call("ppea_dwconv_lk_bwd_filter_f32", ptr(x.float().contiguous()), ptr(dy_big.float().contiguous()),
                 ptr(dwb), N, C, H, W, K, stream_ptr())
            dwb = dwb.to(ctx.w_dtypes[0])
        if ctx.has_small and ctx.needs_input_grad[2]:
            dws = torch.empty_like(ws)
            call("ppea_dwconv_lk_bwd_filter_f32", ptr(x.float().contiguous()),
                 ptr(dy_small.float().contiguous()), ptr(dws), N, C, H, W, KS, stream_ptr())
            dws = dws.to(ctx.w_dtypes[1])
        return dx, dwb, dws, None


class _DwConvLKBn(torch.autograd.Function):
    """(DW_k(t), DW_5(t)) with t = relu(BN(z)) never materialised: the BatchNorm of RepLKBlock's pw1 (training mode,
    statistics from the producing 1x1 conv's epilogue `sums`) and its ReLU are applied while the depthwise kernel stages its
    planes (csrc/dwconv_mfma.hip, BnIn).  Backward: depthwise data gradient, then the BatchNorm + ReLU backward kernels on
    the saved pre-BN tensor (relu' is recomputed from it).   networks/replknet_adapter.py:305-308, 182-197, 232-239"""

    @staticmethod
    def forward(ctx, z, sums, gamma, beta, rm, rv, eps, momentum, w_big, w_small):
        z = z.contiguous()
        N, C, H, W = z.shape
        K = w_big.shape[-1]
        pb, ps = _packed_filter(w_big, False), _packed_filter(w_small, False)
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        st = torch.empty(2, C, device=z.device, dtype=_F32)                     # mean | invstd
        y_big, y_small = torch.empty_like(z), torch.empty_like(z)
        _timed("fwd31", K, lambda: call(
            "ppea_dwconv_lk_fwd_bn_bf16p", ptr(z), ptr(pb), ptr(ps), ptr(y_big), ptr(y_small), ptr(sums, _F32), sums.shape[1],
            N * H * W, ptr(g), ptr(b), float(eps), float(momentum), ptr(rm), ptr(rv), ptr(st[0]), ptr(st[1]), N, C, H, W, K, 5,
            stream_ptr()))
        ctx.save_for_backward(z, st, g, b)
        ctx.packed = (w_big, w_small)
        ctx.pdt = (gamma.dtype, beta.dtype)
        ctx.mark_non_differentiable(st)
        ctx.set_materialize_grads(False)
        return y_big, y_small, st

    @staticmethod
    def backward(ctx, dy_big, dy_small, _dst):
        z, st, g, b = ctx.saved_tensors
        N, C, H, W = z.shape
        K = ctx.packed[0].shape[-1]
        dy_big = torch.zeros_like(z) if dy_big is None else dy_big.contiguous().to(z.dtype)
        dy_small = torch.zeros_like(z) if dy_small is None else dy_small.contiguous().to(z.dtype)
        dt = torch.empty_like(z)
        pb, ps = _packed_filter(ctx.packed[0], True), _packed_filter(ctx.packed[1], True)
        _timed("bwd31", K, lambda: call("ppea_dwconv_lk_bwd_data_bf16p", ptr(dy_big), ptr(dy_small), ptr(pb), ptr(ps),
                                        ptr(dt), N, C, H, W, K, 5, stream_ptr()))
        # BatchNorm + ReLU backward on the saved pre-BN tensor (as _BnActChannel / _BnAct would)
        HW = H * W
        stats = _stats_array((st[0], st[1], g, b, None, None, None, None))
        sums = torch.empty(3, C, device=z.device, dtype=_F32)
        dz = torch.empty_like(z)
        sfx = _suffix(z)
        if bn_channel_ok(z):
            call(f"ppea_bn_bwd_channel_{sfx}", ptr(dt), ptr(z), None, stats, None, 1.0 / float(N * HW), None, ptr(dz), None,
                 ptr(sums), ACT_RELU, N, C, HW, stream_ptr())
        else:
            err = getattr(_abi.lib, f"ppea_bn_bwd_reduce_final_{sfx}")(ptr(dt), ptr(z), None, stats, None, ptr(sums), ACT_RELU,
                                                                       N, C, HW, stream_ptr())
            if err == -1:
                partial = torch.empty(C * N * 3, device=z.device, dtype=_F32)
                call(f"ppea_bn_bwd_reduce_{sfx}", ptr(dt), ptr(z), None, stats, None, ptr(partial), ACT_RELU, N, C, HW,
                     stream_ptr())
                call("ppea_bn_bwd_finalize_f32", ptr(partial), N, C, ptr(sums), stream_ptr())
            else:
                _abi.check(err, "ppea_bn_bwd_reduce_final")
            call(f"ppea_bn_bwd_apply_{sfx}", ptr(dt), ptr(z), None, stats, None, ptr(sums), 1.0 / float(N * HW), ptr(dz), None,
                 ACT_RELU, N, C, HW, stream_ptr())
        dg = sums[1].to(ctx.pdt[0]) if ctx.needs_input_grad[2] else None
        db = sums[0].to(ctx.pdt[1]) if ctx.needs_input_grad[3] else None
        return dz, None, dg, db, None, None, None, None, None, None


def dwconv_lk_bn_supported(shape, K, KS):
    """The fused form needs the MFMA depthwise kernel (bf16 input of `shape`, k in 31/29/27/13, 5x5 branch) and a shape
    that kernel serves."""
    N, C, H, W = shape
    return K in _MFMA_K and KS == 5 and _abi.lib.ppea_dwconv_lk_stats_partials(N, C, H, W, K, KS) > 0


def dwconv_lk_bn(z, sums, bn, w_big, w_small):
    """-> (DW_k(relu(BN(z))), DW_5(relu(BN(z))), stats [2,C] = mean | invstd).  `sums`: the partial sums the 1x1 conv that
    produced z left (pwconv_frozen(..., want_sums=True)).  Frozen depthwise filters only; updates bn's running statistics."""
    # (the filters are passed as they are: the packed fragment images are cached per weight OBJECT)
    assert not w_big.requires_grad and not w_small.requires_grad
    return _DwConvLKBn.apply(z, sums, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum,
                             w_big, w_small)


def dwconv_lk(x, w_big, w_small=None, want_sums=False):
    """(DW_k(x), DW_ks(x)) with stride 1 / pad k//2 / no bias; second is None without w_small.
    want_sums: -> (y_big, y_small, sums [2, C, P, 2] or None): per-channel partial (sum, sum of squares) of both outputs
    from the conv's epilogue, for the two BatchNorms that follow."""
    if want_sums:
        return _DwConvLK.apply(x, w_big, w_small, True)
    return _DwConvLK.apply(x, w_big, w_small)


# ---------------------------------------------------------------------------------------------
# A18+A19  BackprojectDepth -> Project3D                         layers.py:138-199
# ---------------------------------------------------------------------------------------------
class _BackprojectProject(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, inv_K, P, eps):
        depth = depth.contiguous().float()
        inv_K = inv_K.contiguous().float()
        P = P.contiguous().float()
        B, _, H, W = depth.shape
        grid = torch.empty(B, H, W, 2, device=depth.device, dtype=_F32)
        call("ppea_backproject_project_fwd_f32", ptr(depth), ptr(inv_K), ptr(P), ptr(grid), B, H, W,
             float(eps), stream_ptr())
        ctx.save_for_backward(depth, inv_K, P)
        ctx.eps = float(eps)
        return grid

    @staticmethod
    def backward(ctx, d_grid):
        depth, inv_K, P = ctx.saved_tensors
        B, _, H, W = depth.shape
        d_depth = torch.empty_like(depth)
        dP = torch.empty_like(P)
        ws = torch.empty(_abi.lib.ppea_backproject_project_bwd_workspace_bytes(B, H, W) // 4, device=depth.device, dtype=_F32)
        call("ppea_backproject_project_bwd_f32", ptr(depth), ptr(inv_K), ptr(P),
             ptr(d_grid.contiguous().float()), ptr(d_depth), ptr(dP), ptr(ws), B, H, W, ctx.eps, stream_ptr())
        return d_depth, None, dP, None


def backproject_project(depth, inv_K, K, T, eps=1e-7):
    """depth [B,1,H,W] -> sampling grid [B,H,W,2];  P = (K @ T)[:, :3] keeps the pose gradient."""
    P = torch.matmul(K, T)[:, :3, :]
    return _BackprojectProject.apply(depth, inv_K, P, eps)


# ---------------------------------------------------------------------------------------------
# A20  grid_sample (bilinear, align_corners=True)            trainer.py:911-914, rkm.py:299
# ---------------------------------------------------------------------------------------------
class _GridSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, grid, padding):
        src = src.contiguous().float()
        grid = grid.contiguous().float()
        B, C, Hi, Wi = src.shape
        _, Ho, Wo, _ = grid.shape
        out = torch.empty(B, C, Ho, Wo, device=src.device, dtype=_F32)
        call("ppea_grid_sample_fwd_f32", ptr(src), ptr(grid), ptr(out), B, C, Hi, Wi, Ho, Wo, padding,
             stream_ptr())
        ctx.save_for_backward(src, grid)
        ctx.padding = padding
        return out

    @staticmethod
    def backward(ctx, d_out):
        src, grid = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise _abi.PpeaKernelError("grid_sample: gradient w.r.t. the source image is not part of the "
                                       "hot path (sources are input frames)")
        B, C, Hi, Wi = src.shape
        _, Ho, Wo, _ = grid.shape
        d_grid = torch.empty_like(grid)
        call("ppea_grid_sample_bwd_grid_f32", ptr(src), ptr(grid), ptr(d_out.contiguous().float()),
             ptr(d_grid), B, C, Hi, Wi, Ho, Wo, ctx.padding, stream_ptr())
        return None, d_grid, None


def grid_sample(src, grid, padding_mode="border"):
    return _GridSample.apply(src, grid, {"zeros": 0, "border": 1}[padding_mode])


# ---------------------------------------------------------------------------------------------
# A21+A22  reprojection loss                             trainer.py:995-1007, layers.py:226-257
# ---------------------------------------------------------------------------------------------
class _SsimL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, alpha):
        pred = pred.contiguous().float()
        target = target.contiguous().float()
        B, C, H, W = pred.shape
        out = torch.empty(B, 1, H, W, device=pred.device, dtype=_F32)
        call("ppea_ssim_l1_fwd_f32", ptr(pred), ptr(target), ptr(out), H * W, B, C, H, W, float(alpha),
             stream_ptr())
        ctx.save_for_backward(pred, target)
        ctx.alpha = float(alpha)
        return out

    @staticmethod
    def backward(ctx, d_out):
        pred, target = ctx.saved_tensors
        B, C, H, W = pred.shape
        d_pred = torch.empty_like(pred)
        call("ppea_ssim_l1_bwd_f32", ptr(pred), ptr(target), ptr(d_out.contiguous().float()), H * W,
             ptr(d_pred), B, C, H, W, ctx.alpha, stream_ptr())
        return d_pred, None, None


def ssim_l1(pred, target, alpha=0.85):
    """alpha * mean_C SSIM(pred, target) + (1-alpha) * mean_C |target - pred| -> [B,1,H,W].
    Differentiable w.r.t. pred only (target is an input frame)."""
    return _SsimL1.apply(pred, target, alpha)


# ---------------------------------------------------------------------------------------------
# A23  edge-aware smoothness                                         layers.py:210-223
# ---------------------------------------------------------------------------------------------
class _SmoothLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, img):
        disp = disp.contiguous().float()
        img = img.contiguous().float()
        B, C, H, W = img.shape
        nb = _abi.lib.ppea_smooth_num_partials()
        partials = torch.empty(nb, 2, device=disp.device, dtype=_F32)
        call("ppea_smooth_fwd_f32", ptr(disp), ptr(img), ptr(partials), B, C, H, W, stream_ptr())
        s = partials.sum(0)
        ctx.save_for_backward(disp, img)
        ctx.nx = float(B * H * (W - 1))
        ctx.ny = float(B * (H - 1) * W)
        return s[0] / ctx.nx + s[1] / ctx.ny

    @staticmethod
    def backward(ctx, g):
        disp, img = ctx.saved_tensors
        B, C, H, W = img.shape
        d_disp = torch.empty_like(disp)
        call("ppea_smooth_bwd_f32", ptr(disp), ptr(img), 1.0 / ctx.nx, 1.0 / ctx.ny, ptr(d_disp), B, C, H, W,
             stream_ptr())
        return d_disp * g, None      # upstream scalar stays on the device (no host sync)


def smooth_loss(disp, img):
    return _SmoothLoss.apply(disp, img)


# ---------------------------------------------------------------------------------------------
# A24/A25  per-pixel selection                                     trainer.py:1069-1091
# ---------------------------------------------------------------------------------------------
class _LossSelect(torch.autograd.Function):
    @staticmethod
    def forward(ctx, reproj, identity, warped_m1, warped_p1, noise, selec):
        reproj = reproj.contiguous().float()
        B, _, H, W = reproj.shape
        C = warped_m1.shape[1]
        dev = reproj.device
        sel = torch.empty(B, 1, H, W, device=dev, dtype=_F32)
        src = torch.empty(B, 1, H, W, device=dev, dtype=torch.uint8)
        fidx = torch.empty(B, 1, H, W, device=dev, dtype=torch.int64)
        aidx = torch.empty(B, 1, H, W, device=dev, dtype=torch.int64)
        call("ppea_loss_select_f32", ptr(reproj), ptr(identity.contiguous().float()),
             ptr(warped_m1.contiguous().float()), ptr(warped_p1.contiguous().float()),
             ptr(None if noise is None else noise.contiguous().float()), ptr(sel), ptr(src), ptr(fidx),
             ptr(aidx), B, C, H, W, int(bool(selec)), stream_ptr())
        ctx.save_for_backward(src)
        ctx.mark_non_differentiable(src, fidx, aidx)
        ctx.set_materialize_grads(False)
        return sel, src, fidx, aidx

    @staticmethod
    def backward(ctx, d_sel, _a, _b, _c):
        (src,) = ctx.saved_tensors
        if d_sel is None:
            return None, None, None, None, None, None
        d = torch.cat([d_sel * (src == 0), d_sel * (src == 1)], 1)
        return d, None, None, None, None, None


def loss_select(reproj, identity, warped_m1, warped_p1, noise, selec_reproj=True):
    """-> (selected reprojection loss [B,1,H,W], source idx u8, frame argmin i64, automask argmin i64)."""
    return _LossSelect.apply(reproj, identity, warped_m1, warped_p1, noise, selec_reproj)


class _LossTail(torch.autograd.Function):
    """Tail of compute_losses (trainer.py:1092-1139) in one pass per direction (csrc/photometric.hip loss_tail_*):
    -> (rl, consistency_loss, mask, consistency_target).  `reproj` [B,2,H,W] is the differentiable input the selected loss
    `sel` was taken from (`src`: which channel, 2 = forced zero); its gradient is produced directly."""

    @staticmethod
    def forward(ctx, reproj, sel, src, auto_idx, cons, aug, multi, mono, is_multi):
        B, _, H, W = sel.shape
        dev = sel.device
        sel = sel.contiguous().float()
        mask = torch.empty(B, 1, H, W, device=dev, dtype=_F32)
        target = torch.empty(B, 1, H, W, device=dev, dtype=_F32) if is_multi else None
        nblk = _abi.lib.ppea_loss_tail_blocks(B * H * W)
        partial = torch.empty(nblk * 3, device=dev, dtype=_F32)
        out = torch.empty(3, device=dev, dtype=_F32)
        multi_c = None if multi is None else multi.detach().contiguous().float()
        mono_c = None if mono is None else mono.detach().contiguous().float()
        call("ppea_loss_tail_fwd_f32", ptr(sel), ptr(auto_idx), ptr(None if cons is None else cons.contiguous().float()),
             ptr(None if aug is None else aug.reshape(-1).contiguous().float()), ptr(multi_c), ptr(mono_c), ptr(mask),
             ptr(target), ptr(partial), ptr(out), B, H, W, int(bool(is_multi)), stream_ptr())
        ctx.save_for_backward(mask, src, out, multi_c, mono_c)
        ctx.dims = (B, H, W, bool(is_multi), multi is not None and multi.requires_grad)
        ctx.set_materialize_grads(False)
        rl, cl = out[0], out[1]
        if target is not None:
            ctx.mark_non_differentiable(mask, target)
            return rl, cl, mask, target
        ctx.mark_non_differentiable(mask)
        return rl, cl, mask

    @staticmethod
    def backward(ctx, g_rl, g_cl, *_unused):
        mask, src, out, multi, mono = ctx.saved_tensors
        B, H, W, is_multi, multi_grad = ctx.dims
        dev = mask.device
        d_reproj = torch.empty(B, 2, H, W, device=dev, dtype=_F32) if ctx.needs_input_grad[0] else None
        d_multi = torch.empty(B, 1, H, W, device=dev, dtype=_F32) if (is_multi and multi_grad and g_cl is not None) else None
        call("ppea_loss_tail_bwd_f32", ptr(mask), ptr(src), ptr(out), ptr(None if g_rl is None else g_rl.contiguous().float()),
             ptr(None if g_cl is None else g_cl.contiguous().float()), ptr(multi), ptr(mono), ptr(d_reproj), ptr(d_multi),
             B, H, W, stream_ptr())
        return d_reproj, None, None, None, None, None, d_multi, None, None


def loss_tail(reproj, sel, src, auto_idx=None, cons=None, aug=None, multi=None, mono=None, is_multi=False):
    return _LossTail.apply(reproj, sel, src, auto_idx, cons, aug, multi, mono, is_multi)


# ---------------------------------------------------------------------------------------------
# A9/A10  cost volume (runs under no_grad in the reference, rkm.py:427)
# ---------------------------------------------------------------------------------------------
CV_BF16 = True      # bf16 features on the packed-pair kernel (False: widened to fp32 first; bit-identical results)


@torch.no_grad()
def cost_volume(cur, lookup, poses, K, inv_K, bins, eps=1e-7):
    """cur, lookup [B,C,h,w]; poses [B,4,4] (zeroed pose = skipped item); -> raw cost [B,D,h,w]."""
    B, C, h, w = cur.shape
    D = bins.shape[0]
    P = torch.matmul(K, poses)[:, :3, :].contiguous().float()
    skip = (poses.reshape(B, -1).sum(1) == 0).to(torch.int32)
    cost = torch.empty(B, D, h, w, device=cur.device, dtype=_F32)
    if CV_BF16 and cur.dtype == _BF16 and lookup.dtype == _BF16 and C % 2 == 0:
        # bf16 features: channel pairs packed into dwords (half the bytes through the L1), same arithmetic
        pairs = torch.empty(2 * B * (C // 2) * h * w, device=cur.device, dtype=torch.int32)
        call("ppea_cost_volume_fwd_bf16", ptr(cur.contiguous()), ptr(lookup.contiguous()), ptr(pairs), ptr(P),
             ptr(inv_K.contiguous().float()), ptr(bins.contiguous().float()), ptr(skip), ptr(cost), B, C, h, w, D, float(eps),
             stream_ptr())
        return cost
    cur = cur.contiguous().float()
    lookup = lookup.contiguous().float()
    call("ppea_cost_volume_fwd_f32", ptr(cur), ptr(lookup), ptr(P), ptr(inv_K.contiguous().float()),
         ptr(bins.contiguous().float()), ptr(skip), ptr(cost), B, C, h, w, D, float(eps), stream_ptr())
    return cost


@torch.no_grad()
def cost_volume_reduce(cost, bins):
    """-> (masked cost [B,D,h,w], confidence [B,h,w], argmin int64 [B,h,w], lowest-cost 1/depth [B,h,w])."""
    B, D, h, w = cost.shape
    dev = cost.device
    out = torch.empty_like(cost)
    conf = torch.empty(B, h, w, device=dev, dtype=_F32)
    idx = torch.empty(B, h, w, device=dev, dtype=torch.int64)
    low = torch.empty(B, h, w, device=dev, dtype=_F32)
    call("ppea_cost_volume_reduce_f32", ptr(cost.contiguous()), ptr(bins.contiguous().float()), ptr(out),
         ptr(conf), ptr(idx), ptr(low), B, D, h, w, stream_ptr())
    return out, conf, idx, low


# ---------------------------------------------------------------------------------------------
# A2 + block glue: y = act(BN_a(z1) [+ BN_b(z2)]) [* mask[n]] [+ r1] [+ s * r2]   (csrc/bn_fused.hip)
# ---------------------------------------------------------------------------------------------
import ctypes as _ct

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2


def _stats_array(t):
    """host array of 8 device pointers (NULL for the absent second branch)."""
    arr = (_ct.c_void_p * 8)()
    for i, x in enumerate(t):
        arr[i] = None if x is None else x.data_ptr()
    return arr


def bn_batch_stats(z, eps, momentum, running_mean=None, running_var=None):
    """(mean, biased var, invstd) of z over (N,H,W); optionally updates running stats in the same launch."""
    z = z.contiguous()
    N, C = z.shape[0], z.shape[1]
    HW = z.numel() // (N * C)
    dev = z.device
    out = torch.empty(3, C, device=dev, dtype=_F32)
    err = getattr(_abi.lib, f"ppea_bn_stats_final_{_suffix(z)}")(
        ptr(z), N, C, HW, float(eps), float(momentum), ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(running_mean),
        ptr(running_var), stream_ptr())
    if err == 0:                            # small channels: statistics final in one launch
        return out[0], out[1], out[2]
    if err != -1:
        _abi.check(err, "ppea_bn_stats_final")
    partial = torch.empty(C * N * 2, device=dev, dtype=_F32)
    call(f"ppea_bn_stats_{_suffix(z)}", ptr(z), ptr(partial), N, C, HW, stream_ptr())
    call("ppea_bn_finalize_f32", ptr(partial), N, C, HW, float(eps), float(momentum), ptr(out[0]), ptr(out[1]),
         ptr(out[2]), ptr(running_mean), ptr(running_var), stream_ptr())
    return out[0], out[1], out[2]


def bn_batch_stats_from_sums(sums, count, eps, momentum, running_mean=None, running_var=None):
    """(mean, biased var, invstd) from the producing GEMM's partial sums [C, P, 2] (pwconv_frozen(..., want_sums=True)):
    one small launch instead of a statistics pass over the activation; updates the running statistics."""
    C, P = sums.shape[0], sums.shape[1]
    out = torch.empty(3, C, device=sums.device, dtype=_F32)
    call("ppea_bn_finalize_sums_f32", ptr(sums, _F32), P, C, int(count), float(eps), float(momentum), ptr(out[0]), ptr(out[1]),
         ptr(out[2]), ptr(running_mean), ptr(running_var), stream_ptr())
    return out[0], out[1], out[2]


def bn_local_stats_packed(z):
    """SyncBN wire format of the local statistics: [mean(C) | biased var(C) | count] fp32."""
    z = z.contiguous()
    N, C = z.shape[0], z.shape[1]
    HW = z.numel() // (N * C)
    packed = torch.empty(2 * C + 1, device=z.device, dtype=_F32)
    err = getattr(_abi.lib, f"ppea_bn_stats_packed_{_suffix(z)}")(ptr(z), N, C, HW, ptr(packed), stream_ptr())
    if err == 0:
        return packed
    if err != -1:
        _abi.check(err, "ppea_bn_stats_packed")
    partial = torch.empty(C * N * 2, device=z.device, dtype=_F32)
    call(f"ppea_bn_stats_{_suffix(z)}", ptr(z), ptr(partial), N, C, HW, stream_ptr())
    call("ppea_bn_finalize_packed_f32", ptr(partial), N, C, HW, ptr(packed), stream_ptr())
    return packed


def bn_sync_combine(gathered, eps, momentum, running_mean=None, running_var=None):
    """gathered [world, 2C+1] -> (mean, invstd) of the global batch; running statistics updated in the launch."""
    world, C = gathered.shape[0], (gathered.shape[1] - 1) // 2
    out = torch.empty(2, C, device=gathered.device, dtype=_F32)
    call("ppea_bn_sync_combine_f32", ptr(gathered, _F32), world, C, float(eps), float(momentum), ptr(out[0]),
         ptr(out[1]), ptr(running_mean), ptr(running_var), stream_ptr())
    return out[0], out[1]


class _BnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z1, g1, b1, mean1, invstd1, z2, g2, b2, mean2, invstd2, mask, r1, r2, r2_scale, act,
                count, group):
        z1 = z1.contiguous()
        N, C = z1.shape[0], z1.shape[1]
        HW = z1.numel() // (N * C)
        dt = z1.dtype
        z2 = None if z2 is None else z2.contiguous().to(dt)
        r1 = None if r1 is None else r1.contiguous().to(dt)
        r2 = None if r2 is None else r2.contiguous().to(dt)
        g1f, b1f = g1.detach().float().contiguous(), b1.detach().float().contiguous()
        g2f = None if g2 is None else g2.detach().float().contiguous()
        b2f = None if b2 is None else b2.detach().float().contiguous()
        maskf = None if mask is None else mask.detach().reshape(-1).float().contiguous()
        y = torch.empty_like(z1)
        st = (mean1, invstd1, g1f, b1f, mean2, invstd2, g2f, b2f)
        call(f"ppea_bn_apply_{_suffix(z1)}", ptr(z1), ptr(z2), _stats_array(st), ptr(maskf), ptr(r1), ptr(r2),
             float(r2_scale), ptr(y), int(act), N, C, HW, stream_ptr())
        ctx.save_for_backward(z1, z2, mean1, invstd1, g1f, b1f, mean2, invstd2, g2f, b2f, maskf)
        ctx.act, ctx.r2_scale, ctx.count, ctx.group = int(act), float(r2_scale), float(count), group
        ctx.has = (r1 is not None, r2 is not None)
        ctx.pdt = (g1.dtype, b1.dtype, None if g2 is None else g2.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        z1, z2, mean1, invstd1, g1f, b1f, mean2, invstd2, g2f, b2f, maskf = ctx.saved_tensors
        N, C = z1.shape[0], z1.shape[1]
        HW = z1.numel() // (N * C)
        dy = dy.contiguous().to(z1.dtype)
        dev = z1.device
        st = _stats_array((mean1, invstd1, g1f, b1f, mean2, invstd2, g2f, b2f))
        sums = torch.empty(3, C, device=dev, dtype=_F32)
        sfx = _suffix(z1)
        err = getattr(_abi.lib, f"ppea_bn_bwd_reduce_final_{sfx}")(
            ptr(dy), ptr(z1), ptr(z2), st, ptr(maskf), ptr(sums), ctx.act, N, C, HW, stream_ptr())
        if err == -1:                       # large planes: per-plane partials + finalize
            partial = torch.empty(C * N * 3, device=dev, dtype=_F32)
            call(f"ppea_bn_bwd_reduce_{sfx}", ptr(dy), ptr(z1), ptr(z2), st, ptr(maskf), ptr(partial), ctx.act, N, C,
                 HW, stream_ptr())
            call("ppea_bn_bwd_finalize_f32", ptr(partial), N, C, ptr(sums), stream_ptr())
        elif err != 0:
            _abi.check(err, "ppea_bn_bwd_reduce_final")
        inv_count = 1.0 / ctx.count
        gscale = None
        if ctx.group is not None:                      # SyncBN: sums of the global batch (ctx.count is the global count)
            reduce_sums(sums, ctx.group[0])
            gscale = 1.0 / sync_world(ctx.group[0])    # d gamma / d beta: see _SyncBnAct
        dz1 = torch.empty_like(z1)
        dz2 = None if z2 is None else torch.empty_like(z2)
        call(f"ppea_bn_bwd_apply_{sfx}", ptr(dy), ptr(z1), ptr(z2), st, ptr(maskf), ptr(sums), inv_count,
             ptr(dz1), ptr(dz2), ctx.act, N, C, HW, stream_ptr())
        if gscale is not None:
            sums = sums * gscale
        dg1 = sums[1].to(ctx.pdt[0]) if ctx.needs_input_grad[1] else None
        db1 = sums[0].to(ctx.pdt[1]) if ctx.needs_input_grad[2] else None
        dg2 = sums[2].to(ctx.pdt[2]) if (z2 is not None and ctx.needs_input_grad[6]) else None
        db2 = sums[0].to(ctx.pdt[2]) if (z2 is not None and ctx.needs_input_grad[7]) else None
        dr1 = dy if ctx.has[0] else None
        dr2 = (dy if ctx.r2_scale == 1.0 else dy * ctx.r2_scale) if ctx.has[1] else None
        return (dz1, dg1, db1, None, None, dz2, dg2, db2, None, None, None, dr1, dr2, None, None, None, None)


BN_CHANNEL = True          # small channels: statistics + apply (and reduce + apply) as ONE launch each


def bn_channel_ok(z):
    """Whole channel in one workgroup's registers: csrc/bn_fused.hip bn_fwd_channel / bn_bwd_channel."""
    if not (BN_CHANNEL and z.is_cuda and z.dim() == 4 and z.dtype in (_F32, _BF16)):
        return False
    N, C = z.shape[0], z.shape[1]
    HW = z.shape[2] * z.shape[3]
    return C >= 64 and HW % 8 == 0 and N * HW <= 16384


def _ptr_array(ts):
    arr = (_ct.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def _other_stream_grad(g):
    """A gradient that may have been produced on ANOTHER stream (the second consumer of a BatchNorm output is the block's
    adapter, which runs on a forked side stream) and is read by a kernel launched here: mark it as in use by the current
    stream.  The autograd engine makes the consumer stream wait for the producer, but for a gradient that is not accumulated
    with another one it does not record the consumer on the tensor -- the block would go back to the producer stream's pool
    as soon as this backward returns and could be rewritten by that stream's next allocation while the kernel launched here
    has not run yet.  (A precaution: the alias is only handed to consumers on the SAME stream, see rka.ConvFFN.forward.)"""
    if g is not None and g.is_cuda:
        g.record_stream(torch.cuda.current_stream())


class _BnActChannel(torch.autograd.Function):
    """y = act(BN1(z1) [+ BN2(z2)]) [* mask[n]] [+ r1] [+ s * r2] with batch statistics, running-statistics update and
    the saved (mean, invstd) in ONE launch; backward (sums + dz) in one launch."""

    @staticmethod
    def forward(ctx, z1, g1, b1, rm1, rv1, z2, g2, b2, rm2, rv2, mask, r1, r2, r2_scale, act, eps, momentum, skip=False,
                dup=False, sums=None):
        z1_in = z1
        z1 = z1.contiguous()
        N, C = z1.shape[0], z1.shape[1]
        HW = z1.numel() // (N * C)
        dt = z1.dtype
        z2 = None if z2 is None else z2.contiguous().to(dt)
        r1 = None if r1 is None else r1.contiguous().to(dt)
        r2 = None if r2 is None else r2.contiguous().to(dt)
        g1f, b1f = g1.detach().float().contiguous(), b1.detach().float().contiguous()
        g2f = None if g2 is None else g2.detach().float().contiguous()
        b2f = None if b2 is None else b2.detach().float().contiguous()
        maskf = None if mask is None else mask.detach().reshape(-1).float().contiguous()
        st = torch.empty(4, C, device=z1.device, dtype=_F32)          # mean1 | invstd1 | mean2 | invstd2
        y = torch.empty_like(z1)
        if sums is not None and z2 is None:
            # statistics from the producing GEMM's epilogue: [C][P][2] partial (sum, sum of squares) of the stored values
            sums = sums.contiguous()
            call(f"ppea_bn_fwd_channel_sums_{_suffix(z1)}", ptr(z1), ptr(sums, _F32), int(sums.shape[1]), _ptr_array((g1f, b1f)),
                 _ptr_array((rm1, rv1, st[0], st[1])), float(eps), float(momentum), ptr(maskf), ptr(r1), ptr(r2),
                 float(r2_scale), ptr(y), int(act), N, C, HW, stream_ptr())
        else:
            call(f"ppea_bn_fwd_channel_{_suffix(z1)}", ptr(z1), ptr(z2), _ptr_array((g1f, b1f, g2f, b2f)),
                 _ptr_array((rm1, rv1, rm2, rv2, st[0], st[1], st[2], st[3])), float(eps), float(momentum), ptr(maskf),
                 ptr(r1), ptr(r2), float(r2_scale), ptr(y), int(act), N, C, HW, stream_ptr())
        ctx.save_for_backward(z1, z2, st, g1f, b1f, g2f, b2f, maskf)
        ctx.act, ctx.r2_scale = int(act), float(r2_scale)
        ctx.has = (r1 is not None, r2 is not None)
        ctx.pdt = (g1.dtype, b1.dtype, None if g2 is None else g2.dtype)
        ctx.mark_non_differentiable(st)
        ctx.set_materialize_grads(False)         # no zero-filled "gradient" of the statistics output per backward call
        ctx.skip, ctx.dup = bool(skip), bool(dup)
        outs = (y, st)
        if skip:
            # third output: z1 itself, for the block's residual use -- its gradient comes back HERE and is added in the
            # backward launch (autograd would otherwise add the two gradients of z1 with one more element-wise kernel)
            outs += (z1_in,)
        if dup:
            # last output: y once more (same storage) for its SECOND consumer (a block's adapter next to its first 1x1
            # conv): the two gradients arrive separately and are added in the backward launch
            outs += (y.detach(),)
        return outs

    @staticmethod
    def backward(ctx, dy, _dst, *rest):
        rest = list(rest)
        dskip = rest.pop(0) if ctx.skip else None
        dyb = rest.pop(0) if ctx.dup else None
        z1, z2, st, g1f, b1f, g2f, b2f, maskf = ctx.saved_tensors
        N, C = z1.shape[0], z1.shape[1]
        HW = z1.numel() // (N * C)
        _other_stream_grad(dyb)
        if dy is None:
            dy, dyb = (dyb, None) if dyb is not None else (torch.zeros_like(z1), None)
        dy = dy.contiguous().to(z1.dtype)
        if dyb is not None:
            dyb = dyb.contiguous().to(z1.dtype)
        sums = torch.empty(3, C, device=z1.device, dtype=_F32)
        dz1 = torch.empty_like(z1)
        dz2 = None if z2 is None else torch.empty_like(z2)
        stats = _stats_array((st[0], st[1], g1f, b1f, st[2] if z2 is not None else None,
                              st[3] if z2 is not None else None, g2f, b2f))
        if dskip is not None:
            dskip = dskip.contiguous().to(z1.dtype)
        if dyb is not None:
            if ctx.has[0] or ctx.has[1]:
                dy = dy + dyb                                # (r1 / r2 receive dy itself: not a block's first BatchNorm)
                dyb = None
        if dyb is not None:
            call(f"ppea_bn_bwd_channel_dup_{_suffix(z1)}", ptr(dy), ptr(dyb), ptr(z1), ptr(z2), stats, ptr(maskf),
                 1.0 / float(N * HW), ptr(dskip), ptr(dz1), ptr(dz2), ptr(sums), ctx.act, N, C, HW, stream_ptr())
        else:
            call(f"ppea_bn_bwd_channel_{_suffix(z1)}", ptr(dy), ptr(z1), ptr(z2), stats, ptr(maskf), 1.0 / float(N * HW),
                 ptr(dskip), ptr(dz1), ptr(dz2), ptr(sums), ctx.act, N, C, HW, stream_ptr())
        dg1 = sums[1].to(ctx.pdt[0]) if ctx.needs_input_grad[1] else None
        db1 = sums[0].to(ctx.pdt[1]) if ctx.needs_input_grad[2] else None
        dg2 = sums[2].to(ctx.pdt[2]) if (z2 is not None and ctx.needs_input_grad[6]) else None
        db2 = sums[0].to(ctx.pdt[2]) if (z2 is not None and ctx.needs_input_grad[7]) else None
        dr1 = dy if ctx.has[0] else None
        dr2 = (dy if ctx.r2_scale == 1.0 else dy * ctx.r2_scale) if ctx.has[1] else None
        return (dz1, dg1, db1, None, None, dz2, dg2, db2, None, None, None, dr1, dr2, None, None, None, None, None, None, None)


def bn_act_channel(z1, bn1, z2=None, bn2=None, mask=None, r1=None, r2=None, r2_scale=1.0, act=ACT_NONE, skip=False,
                   dup=False, sums=None):
    """-> (y, stats [4,C] = mean1 | invstd1 | mean2 | invstd2 [, z1 for the residual use when `skip`] [, y again for its
    second consumer when `dup`]).  Updates the running statistics of bn1 / bn2."""
    return _BnActChannel.apply(z1, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, z2,
                               None if bn2 is None else bn2.weight, None if bn2 is None else bn2.bias,
                               None if bn2 is None else bn2.running_mean, None if bn2 is None else bn2.running_var,
                               mask, r1, r2, r2_scale, act, bn1.eps, bn1.momentum, skip, dup, sums)


class _BnActChannelNext(torch.autograd.Function):
    """End of one block + first BatchNorm of the next in one launch per direction (csrc/bn_fused.hip, bn_*_channel_next):
    y = mask * BN_A(z) + r1 + s * r2,  y2 = BN_B(y).  Bit-identical to _BnActChannel(z, A, ...) followed by
    _BnActChannel(y, B, skip=True)."""

    @staticmethod
    def forward(ctx, z, gA, bA, rmA, rvA, gB, bB, rmB, rvB, mask, r1, r2, r2_scale, eps, momentum, dup=False):
        z = z.contiguous()
        N, C = z.shape[0], z.shape[1]
        HW = z.numel() // (N * C)
        dt = z.dtype
        r1 = None if r1 is None else r1.contiguous().to(dt)
        r2 = None if r2 is None else r2.contiguous().to(dt)
        gAf, bAf = gA.detach().float().contiguous(), bA.detach().float().contiguous()
        gBf, bBf = gB.detach().float().contiguous(), bB.detach().float().contiguous()
        maskf = None if mask is None else mask.detach().reshape(-1).float().contiguous()
        st = torch.empty(4, C, device=z.device, dtype=_F32)          # meanA | invstdA | meanB | invstdB
        y, y2 = torch.empty_like(z), torch.empty_like(z)
        call(f"ppea_bn_fwd_channel_next_{_suffix(z)}", ptr(z), _ptr_array((gAf, bAf, gBf, bBf)),
             _ptr_array((rmA, rvA, rmB, rvB, st[0], st[1], st[2], st[3])), float(eps), float(momentum), ptr(maskf),
             ptr(r1), ptr(r2), float(r2_scale), ptr(y), ptr(y2), N, C, HW, stream_ptr())
        ctx.save_for_backward(z, y, st, gAf, bAf, gBf, bBf, maskf)
        ctx.r2_scale = float(r2_scale)
        ctx.has = (r1 is not None, r2 is not None)
        ctx.pdt = (gA.dtype, bA.dtype, gB.dtype, bB.dtype)
        ctx.mark_non_differentiable(st)
        ctx.set_materialize_grads(False)
        if dup:                                              # y2 once more for its second consumer (see _BnActChannel)
            return y, y2, st, y2.detach()
        return y, y2, st

    @staticmethod
    def backward(ctx, dy, dy2, _dst, dy2b=None):
        z, y, st, gAf, bAf, gBf, bBf, maskf = ctx.saved_tensors
        N, C = z.shape[0], z.shape[1]
        HW = z.numel() // (N * C)
        _other_stream_grad(dy2b)
        if dy2 is None:
            dy2, dy2b = (dy2b, None) if dy2b is not None else (torch.zeros_like(z), None)
        dy2 = dy2.contiguous().to(z.dtype)
        dskip = None if dy is None else dy.contiguous().to(z.dtype)
        sums = torch.empty(4, C, device=z.device, dtype=_F32)
        dz, dyt = torch.empty_like(z), torch.empty_like(z)
        stats = _stats_array((st[0], st[1], gAf, bAf, st[2], st[3], gBf, bBf))
        if dy2b is not None:
            dy2b = dy2b.contiguous().to(z.dtype)
            call(f"ppea_bn_bwd_channel_next_dup_{_suffix(z)}", ptr(dy2), ptr(dy2b), ptr(dskip), ptr(z),
                 ptr(y), stats, ptr(maskf), 1.0 / float(N * HW), ptr(dz), ptr(dyt), ptr(sums), N, C, HW, stream_ptr())
        else:
            call(f"ppea_bn_bwd_channel_next_{_suffix(z)}", ptr(dy2), ptr(dskip), ptr(z), ptr(y), stats, ptr(maskf),
                 1.0 / float(N * HW), ptr(dz), ptr(dyt), ptr(sums), N, C, HW, stream_ptr())
        n = ctx.needs_input_grad
        dgA = sums[1].to(ctx.pdt[0]) if n[1] else None
        dbA = sums[0].to(ctx.pdt[1]) if n[2] else None
        dgB = sums[3].to(ctx.pdt[2]) if n[5] else None
        dbB = sums[2].to(ctx.pdt[3]) if n[6] else None
        dr1 = dyt if ctx.has[0] else None
        dr2 = (dyt if ctx.r2_scale == 1.0 else dyt * ctx.r2_scale) if ctx.has[1] else None
        return (dz, dgA, dbA, None, None, dgB, dbB, None, None, None, dr1, dr2, None, None, None, None)


def bn_act_channel_next(z, bnA, bnB, mask=None, r1=None, r2=None, r2_scale=1.0, dup=False):
    """-> (y, y2, stats [4,C] = meanA | invstdA | meanB | invstdB [, y2 again for its second consumer when `dup`]).  Updates
    the running statistics of both BNs."""
    return _BnActChannelNext.apply(z, bnA.weight, bnA.bias, bnA.running_mean, bnA.running_var, bnB.weight, bnB.bias,
                                   bnB.running_mean, bnB.running_var, mask, r1, r2, r2_scale, bnA.eps, bnA.momentum, dup)


# ---------------------------------------------------------------------------------------------
# A2 across ranks: SyncBatchNorm on the fused kernels (csrc/bn_sync.hip) -- two launches around ONE collective per
# BatchNorm and direction                                                   networks/replknet_adapter.py:170-180
# ---------------------------------------------------------------------------------------------
SYNC_COUNTERS = None      # bench.py / tests: {"launches": n, "collectives": m} accumulated while set to a dict


def _count(launches=0, collectives=0):
    if SYNC_COUNTERS is not None:
        SYNC_COUNTERS["launches"] = SYNC_COUNTERS.get("launches", 0) + launches
        SYNC_COUNTERS["collectives"] = SYNC_COUNTERS.get("collectives", 0) + collectives


def sync_bn_supported(z):
    return (z.is_cuda and z.dim() == 4 and z.dtype in (_F32, _BF16) and (z.shape[2] * z.shape[3]) % 8 == 0)


def sync_world(group):
    import torch.distributed as dist
    return dist.get_world_size(group)


def sync_rank(group):
    import torch.distributed as dist
    return dist.get_rank(group)


def gather_rows(table, group):
    """ONE all-gather, IN PLACE: `table` [world, pitch] arrives with this rank's row filled (the statistics kernel wrote
    straight into it) and leaves with every rank's row -- no staging copy on either side of the collective."""
    import torch.distributed as dist
    from .dist import log_collective
    mine = table[sync_rank(group)]
    log_collective("all_gather", table, group)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(table, mine, group=group)
    else:
        dist.all_gather(list(table.unbind(0)), mine.clone(), group=group)     # gloo has no tensor form
    _count(collectives=1)


def reduce_sums(sums, group):
    """ONE in-place all-reduce (sum) of a BatchNorm backward's [3][C] sums."""
    import torch.distributed as dist
    from .dist import log_collective
    log_collective("all_reduce", sums, group)
    dist.all_reduce(sums, group=group)
    _count(collectives=1)


class _SyncBnAct(torch.autograd.Function):
    """y = act(BN1(z1) [+ BN2(z2)]) [* mask[n]] [+ r1] [+ s * r2] with statistics of the GLOBAL batch.
    forward : local statistics in wire format written into this rank's row of the gather table (one launch; none when
              `table` comes from the previous BatchNorm's apply launch; a tiny one when the producing GEMM left partial
              `sums`) -> in-place all-gather -> combine + running statistics + apply in one launch (`emit`: that launch
              also leaves the local statistics of y in the next BatchNorm's table);
    backward: reduce -> all-reduce [3][C] -> apply (+ the gradient `z1` receives through its other use when `skip`).
              d gamma / d beta leave as (global sum) / world: what the reference's DDP mean of the per-rank LOCAL sums
              comes to (torch's SyncBatchNorm returns local grad_weight / grad_bias; trainer.py:215-222)."""

    @staticmethod
    def forward(ctx, z1, g1, b1, rm1, rv1, z2, g2, b2, rm2, rv2, mask, r1, r2, r2_scale, act, eps, momentum, group, table,
                sums, skip, emit, dup=False):
        z1_in = z1
        z1 = z1.contiguous()
        N, C = z1.shape[0], z1.shape[1]
        HW = z1.numel() // (N * C)
        dt = z1.dtype
        sfx = _suffix(z1)
        dev = z1.device
        z2 = None if z2 is None else z2.contiguous().to(dt)
        r1 = None if r1 is None else r1.contiguous().to(dt)
        r2 = None if r2 is None else r2.contiguous().to(dt)
        g1f, b1f = g1.detach().float().contiguous(), b1.detach().float().contiguous()
        g2f = None if g2 is None else g2.detach().float().contiguous()
        b2f = None if b2 is None else b2.detach().float().contiguous()
        maskf = None if mask is None else mask.detach().reshape(-1).float().contiguous()
        two = z2 is not None
        pitch = (4 if two else 2) * C + 1
        world, rank = sync_world(group), sync_rank(group)
        if table is None:
            table = torch.empty(world, pitch, device=dev, dtype=_F32)
            row = _ct.c_void_p(table.data_ptr() + 4 * pitch * rank)
            if sums is not None and isinstance(sums, tuple) == two:
                # producer-epilogue partial sums [C][P][2]; a pair fills the two halves of one row (the first call's count
                # lands on mean2[0] and is overwritten by the second call, which runs after it on the stream)
                for k, sk in enumerate(sums if two else (sums,)):
                    call("ppea_bn_sync_stats_from_sums_f32", ptr(sk.contiguous(), _F32), sk.shape[1], C, N * HW,
                         _ct.c_void_p(row.value + 8 * C * k), stream_ptr())
                _count(launches=2 if two else 1)
            else:
                nws = _abi.lib.ppea_bn_sync_stats_workspace_bytes(N, C, HW, int(two)) // 4
                ws = torch.empty(nws, device=dev, dtype=_F32) if nws else None
                call(f"ppea_bn_sync_stats_{sfx}", ptr(z1), ptr(z2), row, ptr(ws), N, C, HW, stream_ptr())
                _count(launches=1 if not nws else (3 if two else 2))
        assert tuple(table.shape) == (world, pitch)
        gather_rows(table, group)
        st = torch.empty(4, C, device=dev, dtype=_F32)          # mean1 | invstd1 | mean2 | invstd2
        y = torch.empty_like(z1)
        nxt = torch.empty(world, 2 * C + 1, device=dev, dtype=_F32) if emit else None
        call(f"ppea_bn_sync_apply_{sfx}", ptr(z1), ptr(z2), ptr(table), world, _ptr_array((g1f, b1f, g2f, b2f)),
             _ptr_array((rm1, rv1, rm2, rv2, st[0], st[1], st[2], st[3])), float(eps), float(momentum), ptr(maskf), ptr(r1),
             ptr(r2), float(r2_scale), ptr(y), None if nxt is None else _ct.c_void_p(nxt.data_ptr() + 4 * (2 * C + 1) * rank),
             int(act), N, C, HW, stream_ptr())
        _count(launches=1)
        ctx.save_for_backward(z1, z2, st, g1f, b1f, g2f, b2f, maskf)
        ctx.act, ctx.r2_scale, ctx.group, ctx.world = int(act), float(r2_scale), group, world
        ctx.has = (r1 is not None, r2 is not None)
        ctx.pdt = (g1.dtype, b1.dtype, None if g2 is None else g2.dtype)
        ctx.skip, ctx.emit, ctx.dup = bool(skip), bool(emit), bool(dup)
        ctx.set_materialize_grads(False)
        outs = [y, st]
        ctx.mark_non_differentiable(st)
        if skip:
            outs.append(z1_in)
        if emit:
            outs.append(nxt)
            ctx.mark_non_differentiable(nxt)
        if dup:
            outs.append(y.detach())      # y again (same storage) for its second consumer: see _BnActChannel
        return tuple(outs)

    @staticmethod
    def backward(ctx, dy, _dst, *rest):
        z1, z2, st, g1f, b1f, g2f, b2f, maskf = ctx.saved_tensors
        N, C = z1.shape[0], z1.shape[1]
        HW = z1.numel() // (N * C)
        sfx = _suffix(z1)
        dev = z1.device
        if dy is None:
            dy = torch.zeros_like(z1)
        dy = dy.contiguous().to(z1.dtype)
        rest = list(rest)
        dskip = rest.pop(0) if ctx.skip else None
        if ctx.emit:
            rest.pop(0)
        dyb = rest.pop(0) if ctx.dup else None
        if dskip is not None:
            dskip = dskip.contiguous().to(z1.dtype)
        _other_stream_grad(dyb)
        if dyb is not None and (ctx.has[0] or ctx.has[1]):
            dy, dyb = dy + dyb.to(z1.dtype), None             # (r1 / r2 receive dy itself: not a block's first BatchNorm)
        two = z2 is not None
        stats = _stats_array((st[0], st[1], g1f, b1f, st[2] if two else None, st[3] if two else None, g2f, b2f))
        sums = torch.empty(3, C, device=dev, dtype=_F32)
        if dyb is not None:
            dyb = dyb.contiguous().to(z1.dtype)
            dym = torch.empty_like(z1)
            err = getattr(_abi.lib, f"ppea_bn_bwd_reduce_final_dup_{sfx}")(
                ptr(dy), ptr(dyb), ptr(dym), ptr(z1), ptr(z2), stats, ptr(maskf), ptr(sums), ctx.act, N, C, HW, stream_ptr())
            dy = dym if err != -1 else dy + dyb
        else:
            err = getattr(_abi.lib, f"ppea_bn_bwd_reduce_final_{sfx}")(
                ptr(dy), ptr(z1), ptr(z2), stats, ptr(maskf), ptr(sums), ctx.act, N, C, HW, stream_ptr())
        if err == -1:                       # large planes: per-plane partials + finalize
            partial = torch.empty(C * N * 3, device=dev, dtype=_F32)
            call(f"ppea_bn_bwd_reduce_{sfx}", ptr(dy), ptr(z1), ptr(z2), stats, ptr(maskf), ptr(partial), ctx.act, N, C,
                 HW, stream_ptr())
            call("ppea_bn_bwd_finalize_f32", ptr(partial), N, C, ptr(sums), stream_ptr())
            _count(launches=2)
        else:
            _abi.check(err, "ppea_bn_bwd_reduce_final")
            _count(launches=1)
        reduce_sums(sums, ctx.group)                            # -> sums of the global batch
        dz1 = torch.empty_like(z1)
        dz2 = torch.empty_like(z2) if two else None
        dgb = torch.empty(3, C, device=dev, dtype=_F32)         # d beta | d gamma1 | d gamma2, already divided by world
        call(f"ppea_bn_sync_bwd_apply_{sfx}", ptr(dy), ptr(z1), ptr(z2), stats, ptr(maskf), ptr(sums),
             1.0 / float(N * HW * ctx.world), ptr(dskip), ptr(dz1), ptr(dz2), ptr(dgb), 1.0 / ctx.world, ctx.act, N, C, HW,
             stream_ptr())
        _count(launches=1)
        n = ctx.needs_input_grad
        dg1 = dgb[1].to(ctx.pdt[0]) if n[1] else None
        db1 = dgb[0].to(ctx.pdt[1]) if n[2] else None
        dg2 = dgb[2].to(ctx.pdt[2]) if (two and n[6]) else None
        db2 = dgb[0].to(ctx.pdt[2]) if (two and n[7]) else None
        dr1 = dy if ctx.has[0] else None
        dr2 = (dy if ctx.r2_scale == 1.0 else dy * ctx.r2_scale) if ctx.has[1] else None
        return (dz1, dg1, db1, None, None, dz2, dg2, db2, None, None, None, dr1, dr2) + (None,) * 10


def sync_bn_act(z1, bn1, z2=None, bn2=None, mask=None, r1=None, r2=None, r2_scale=1.0, act=ACT_NONE, group=None,
                table=None, sums=None, skip=False, emit=False, dup=False):
    """-> (y, stats [4,C] = mean1 | invstd1 | mean2 | invstd2 of the global batch [, z1 for the residual use when `skip`]
    [, the next BatchNorm's gather table with this rank's statistics of y filled in when `emit`] [, y again for its second
    consumer when `dup`]).  `table`: such a table
    from the launch that produced z1.  Updates the running statistics of bn1 / bn2."""
    return _SyncBnAct.apply(z1, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, z2,
                            None if bn2 is None else bn2.weight, None if bn2 is None else bn2.bias,
                            None if bn2 is None else bn2.running_mean, None if bn2 is None else bn2.running_var,
                            mask, r1, r2, r2_scale, act, bn1.eps, bn1.momentum, group, table, sums, skip, emit, dup)


def bn_act_apply(z1, g1, b1, mean1, invstd1, z2=None, g2=None, b2=None, mean2=None, invstd2=None, mask=None,
                 r1=None, r2=None, r2_scale=1.0, act=ACT_NONE, count=None, group=None):
    if count is None:
        count = z1.numel() // z1.shape[1]
    return _BnAct.apply(z1, g1, b1, mean1, invstd1, z2, g2, b2, mean2, invstd2, mask, r1, r2, r2_scale, act, count,
                        group)


# ---------------------------------------------------------------------------------------------
# pose trunk: training-mode BN (+ReLU, + residual before the activation) on channels_last tensors, per sub-batch
# ---------------------------------------------------------------------------------------------
def _raw(t, byte_offset=0):
    return None if t is None else _ct.c_void_p(t.data_ptr() + byte_offset)


def nhwc_bn_supported(x, groups=1):
    if not (x.is_cuda and x.dim() == 4 and x.dtype in (_F32, _BF16)):
        return False
    N, C = x.shape[0], x.shape[1]
    ct = C // 8
    return (C % 8 == 0 and 1 <= ct <= 256 and 256 % ct == 0 and N % groups == 0
            and x.is_contiguous(memory_format=torch.channels_last))


class _NhwcBnAct(torch.autograd.Function):
    """y = act(BN_g(x) + res) with separate batch statistics for each of `groups` consecutive sub-batches; the
    running statistics are updated once per sub-batch, in order.  Returns (y, stats [G,3,C]) with stats =
    mean | invstd | unbiased variance per sub-batch."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, res, act, groups, eps, momentum):
        N, C, H, W = x.shape
        P = (N // groups) * H * W
        sfx = _suffix(x)
        dev = x.device
        gam, bet = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        slabs = _abi.lib.ppea_nhwc_bn_slabs(P, C)
        partial = torch.empty(groups * slabs * 2 * C, device=dev, dtype=_F32)
        stats = torch.empty(groups, 3, C, device=dev, dtype=_F32)
        ab = torch.empty(groups, 2, C, device=dev, dtype=_F32)
        y = torch.empty_like(x)                             # preserves channels_last
        st = stream_ptr()
        call(f"ppea_nhwc_bn_stats_{sfx}", _raw(x), ptr(partial), P, C, groups, st)
        call("ppea_nhwc_bn_finalize_f32", ptr(partial), P, C, groups, ptr(gam), ptr(bet), float(eps), float(momentum),
             ptr(stats), ptr(ab), ptr(running_mean), ptr(running_var), st)
        call(f"ppea_nhwc_bn_apply_{sfx}", _raw(x), _raw(res), ptr(ab), _raw(y), P, C, groups, int(act), st)
        ctx.save_for_backward(x, res, stats, ab)
        ctx.cfg = (int(act), int(groups), P, C, weight.dtype, bias.dtype)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _ds):
        x, res, stats, ab = ctx.saved_tensors
        act, groups, P, C, wdt, bdt = ctx.cfg
        sfx = _suffix(x)
        dev = x.device
        if dy is None:
            dy = torch.zeros_like(x)
        dy = dy.contiguous(memory_format=torch.channels_last).to(x.dtype)
        slabs = _abi.lib.ppea_nhwc_bn_slabs(P, C)
        partial = torch.empty(groups * slabs * 2 * C, device=dev, dtype=_F32)
        kk = torch.empty(groups, 2, C, device=dev, dtype=_F32)
        tot = torch.empty(2, C, device=dev, dtype=_F32)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if res is not None else None
        st = stream_ptr()
        call(f"ppea_nhwc_bn_bwd_reduce_{sfx}", _raw(x), _raw(dy), _raw(res), ptr(stats), ptr(ab), ptr(partial),
             P, C, groups, act, st)
        call("ppea_nhwc_bn_bwd_finalize_f32", ptr(partial), P, C, groups, ptr(kk), ptr(tot), st)
        call(f"ppea_nhwc_bn_bwd_apply_{sfx}", _raw(x), _raw(dy), _raw(res), ptr(stats), ptr(ab), ptr(kk),
             _raw(dx), _raw(dres), P, C, groups, act, st)
        return dx, tot[0].to(wdt), tot[1].to(bdt), None, None, dres, None, None, None, None


def nhwc_bn_act(x, weight, bias, running_mean, running_var, res=None, act=ACT_NONE, groups=1, eps=1e-5, momentum=0.1):
    return _NhwcBnAct.apply(x, weight, bias, running_mean, running_var, res, act, groups, eps, momentum)


# ---------------------------------------------------------------------------------------------
# pose trunk: MaxPool2d(3, 2, 1) on channels_last tensors (csrc/nhwc_pool.hip)          resnet_encoder.py:376-392
# ---------------------------------------------------------------------------------------------
class _MaxPoolNhwc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        N, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty(N, C, Ho, Wo, device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
        idx = torch.empty(N, Ho, Wo, C, device=x.device, dtype=torch.uint8)
        call(f"ppea_nhwc_maxpool3x3s2_fwd_{_suffix(x)}", _raw(x), _raw(y), ptr(idx), N, H, W, C, stream_ptr())
        ctx.save_for_backward(idx)
        ctx.dims = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        N, C, H, W = ctx.dims
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty(N, C, H, W, device=dy.device, dtype=dy.dtype, memory_format=torch.channels_last)
        call(f"ppea_nhwc_maxpool3x3s2_bwd_{_suffix(dy)}", _raw(dy), ptr(idx), _raw(dx), N, H, W, C, stream_ptr())
        return dx


def maxpool3x3s2(x):
    """nn.MaxPool2d(3, 2, 1)(x) for a channels_last HIP tensor (C % 8 == 0, fp32 / bf16) on the gather kernels; None when
    this call is not served."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype in (_F32, _BF16) and x.shape[1] % 8 == 0 and _is_nhwc(x)):
        return None
    return _MaxPoolNhwc.apply(x)


# ---------------------------------------------------------------------------------------------
# A12 glue: bias + ELU of ConvBlock                                         layers.py:103-116
# ---------------------------------------------------------------------------------------------
def _is_nhwc(t):
    """channels_last storage (and not at the same time NCHW-contiguous, as C == 1 or H == W == 1 would be)."""
    return t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last)


def _nhwc_c_ok(C):
    ct = C // 8
    return C % 8 == 0 and 1 <= ct <= 256 and 256 % ct == 0


class _BiasEluNhwc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, bias):
        N, C, H, W = z.shape
        y = torch.empty_like(z)
        bp, bflag = _bias_arg(bias.detach().contiguous())
        call(f"ppea_nhwc_bias_elu_fwd_{_suffix(z)}", _raw(z), bp, bflag, _raw(y), N * H * W, C, stream_ptr())
        ctx.save_for_backward(y)
        ctx.bdt = bias.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        N, C, H, W = y.shape
        dy = dy.contiguous(memory_format=torch.channels_last).to(y.dtype)
        slabs = _abi.lib.ppea_nhwc_bias_elu_slabs(N * H * W, C)
        partial = torch.empty(slabs, C, device=y.device, dtype=_F32)
        dz = torch.empty_like(y)
        call(f"ppea_nhwc_bias_elu_bwd_{_suffix(y)}", _raw(dy), _raw(y), _raw(dz), ptr(partial), N * H * W, C,
             stream_ptr())
        return dz, partial.sum(0).to(ctx.bdt)


class _BiasElu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, bias):
        z = z.contiguous()
        N, C = z.shape[0], z.shape[1]
        HW = z.numel() // (N * C)
        y = torch.empty_like(z)
        bp, bflag = _bias_arg(bias.detach().contiguous())
        call(f"ppea_bias_elu_fwd_{_suffix(z)}", ptr(z), bp, bflag, ptr(y), N, C, HW, stream_ptr())
        ctx.save_for_backward(y)
        ctx.bdt = bias.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        N, C = y.shape[0], y.shape[1]
        HW = y.numel() // (N * C)
        dy = dy.contiguous().to(y.dtype)
        chunks = _abi.lib.ppea_bias_elu_chunks(N, C, HW)
        partial = torch.empty(N, C, chunks, device=y.device, dtype=_F32)
        dz = torch.empty_like(y)
        call(f"ppea_bias_elu_bwd_{_suffix(y)}", ptr(dy), ptr(y), ptr(dz), ptr(partial), N, C, HW, stream_ptr())
        return dz, partial.sum((0, 2)).to(ctx.bdt)


def bias_elu(z, bias):
    """elu(z + bias[None, :, None, None]) in one pass; the bias gradient comes out of the backward pass.
    NCHW or channels_last input (output in the same format)."""
    if _is_nhwc(z) and _nhwc_c_ok(z.shape[1]):
        return _BiasEluNhwc.apply(z, bias)
    return _BiasElu.apply(z, bias)


# ---------------------------------------------------------------------------------------------
# A12 glue: ReflectionPad2d(1)                                            layers.py:119-135
# ---------------------------------------------------------------------------------------------
class _ReflectPad1Nhwc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, C, H, W = x.shape
        out = torch.empty(B, C, H + 2, W + 2, device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
        call(f"ppea_nhwc_reflect_pad1_fwd_{_suffix(x)}", _raw(x), _raw(out), B, H, W, C, stream_ptr())
        ctx.shape = (B, C, H, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, C, H, W = ctx.shape
        dout = dout.contiguous(memory_format=torch.channels_last)
        dx = torch.empty(B, C, H, W, device=dout.device, dtype=dout.dtype, memory_format=torch.channels_last)
        call(f"ppea_nhwc_reflect_pad1_bwd_{_suffix(dout)}", _raw(dout), _raw(dx), B, H, W, C, stream_ptr())
        return dx


class _ReflectPad1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, C, H, W = x.shape
        out = torch.empty(B, C, H + 2, W + 2, device=x.device, dtype=x.dtype)
        call(f"ppea_reflect_pad1_fwd_{_suffix(x)}", ptr(x), ptr(out), B * C, H, W, stream_ptr())
        ctx.shape = (B, C, H, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, C, H, W = ctx.shape
        dout = dout.contiguous()
        din = torch.empty(B, C, H, W, device=dout.device, dtype=dout.dtype)
        call(f"ppea_reflect_pad1_bwd_{_suffix(dout)}", ptr(dout), ptr(din), B * C, H, W, stream_ptr())
        return din


def reflect_pad1(x):
    if x.shape[-1] < 3 or x.shape[-2] < 3 or x.dtype not in (_F32, _BF16) or not x.is_cuda:
        return torch.nn.functional.pad(x, (1, 1, 1, 1), mode="reflect")
    if _is_nhwc(x) and x.shape[1] % 8 == 0:
        return _ReflectPad1Nhwc.apply(x)
    return _ReflectPad1.apply(x)


# ---------------------------------------------------------------------------------------------
# depthwise 3x3 (stride 1 / 2, pad 1) of the stem and the stage transitions   rka.py:414-416, 451-453
# ---------------------------------------------------------------------------------------------
class _DwConv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride):
        x = x.contiguous()
        N, C, H, W = x.shape
        wf = w.detach().to(_F32).contiguous()
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        y = torch.empty(N, C, Ho, Wo, device=x.device, dtype=x.dtype)
        call(f"ppea_dwconv3x3_fwd_{_suffix(x)}", ptr(x), ptr(wf), ptr(y), N, C, H, W, stride, stream_ptr())
        ctx.save_for_backward(wf)
        ctx.meta = (N, C, H, W, stride, x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        (wf,) = ctx.saved_tensors
        N, C, H, W, stride, dt = ctx.meta
        dy = dy.contiguous().to(dt)
        dx = torch.empty(N, C, H, W, device=dy.device, dtype=dt)
        call(f"ppea_dwconv3x3_bwd_data_{_suffix(dy)}", ptr(dy), ptr(wf), ptr(dx), N, C, H, W, stride, stream_ptr())
        return dx, None, None


def dwconv3x3(x, w, stride):
    """Frozen-filter depthwise 3x3 (no weight gradient: the backbone convs are frozen, repdepth.py:47-50)."""
    return _DwConv3x3.apply(x, w.detach(), stride)


# ---------------------------------------------------------------------------------------------
# A5/A6  pointwise conv on MFMA (frozen 1x1 convs: forward with W, data gradient with W^T)
# ---------------------------------------------------------------------------------------------
_PW_CACHE = {}


def _pw_matrices(w):
    """(W [Cout][Cin] bf16, W^T [Cin][Cout] bf16) of a frozen 1x1 conv weight, cached per tensor object/version."""
    key = id(w)
    hit = _PW_CACHE.get(key)
    if hit is not None and hit[4]() is w and hit[0] == w._version and hit[3] == tuple(w.shape):
        return hit[1], hit[2]
    m = w.detach().reshape(w.shape[0], w.shape[1]).to(_BF16).contiguous()
    mt = m.t().contiguous()
    _PW_CACHE[key] = (w._version, m, mt, tuple(w.shape), weakref.ref(w, lambda _r, k=key: _PW_CACHE.pop(k, None)))
    return m, mt


def pwconv_raw(a_mat, x, bias=None):
    """Y[b] = A @ X[b] (+ bias) for x [B,K,H,W] bf16, A [M,K] bf16 -> [B,M,H,W]; None if unsupported."""
    B, K, H, W = x.shape
    M = a_mat.shape[0]
    y = torch.empty(B, M, H, W, device=x.device, dtype=_BF16)
    err = _abi.lib.ppea_pwconv_bf16(ptr(a_mat, _BF16), ptr(x, _BF16), ptr(bias), ptr(y), B, M, K, H * W, stream_ptr())
    if err == -1:
        return None
    _abi.check(err, "ppea_pwconv_bf16")
    return y


class _PwConvFrozen(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, want_sums=False):
        a, at = _pw_matrices(w)
        ctx.at = at
        x = x.contiguous()
        if not want_sums:
            return pwconv_raw(a, x)
        # the epilogue also leaves per-channel partial sums of the stored output: the BatchNorm that follows needs no
        # statistics pass (batchnorm.fused_bn_act(..., sums=...))
        B, K, H, W = x.shape
        M = a.shape[0]
        y = torch.empty(B, M, H, W, device=x.device, dtype=_BF16)
        P = _abi.lib.ppea_pwconv_stats_partials(B, M, K, H * W)
        sums = torch.empty(M, P, 2, device=x.device, dtype=_F32)
        call("ppea_pwconv_stats_bf16", ptr(a, _BF16), ptr(x, _BF16), None, ptr(y), ptr(sums), B, M, K, H * W, stream_ptr())
        ctx.mark_non_differentiable(sums)
        ctx.set_materialize_grads(False)
        return y, sums

    @staticmethod
    def backward(ctx, dy, _dsums=None):
        return pwconv_raw(ctx.at, dy.contiguous().to(_BF16)), None, None


def pwconv_frozen(x, w, want_sums=False):
    """1x1 conv with a frozen weight [Cout,Cin,1,1] on the MFMA kernel; None when the shape is not served
    (caller falls back to the library conv).  want_sums: -> (y, sums [Cout, P, 2]) with the per-channel partial
    (sum, sum of squares) of y for the BatchNorm that follows."""
    B, K, H, W = x.shape
    if (x.dtype != _BF16 or not x.is_cuda or K % 32 != 0 or w.shape[0] % 32 != 0 or (H * W) % 8 != 0
            or w.requires_grad):
        return None
    if want_sums:
        return _PwConvFrozen.apply(x, w, True)
    return _PwConvFrozen.apply(x, w)


# ---------------------------------------------------------------------------------------------
# A4  adapters on the MFMA kernels, NCHW bf16 end to end (replknet_adapter.py:20-109)
# ---------------------------------------------------------------------------------------------
EPI_NONE, EPI_GELU, EPI_DGELU = 0, 1, 2


def _bias_arg(b):
    if b is None:
        return None, 0
    if b.dtype == _BF16:
        return ptr(b, _BF16), 1
    return ptr(b, _F32), 0


def pwconv_ex(a_mat, x, bias=None, epi=EPI_NONE, aux=None, transposed=False):
    """A [M,K] bf16 (or, with transposed=True, At [K,M]) applied over the channel axis of x [B,K,H,W] bf16 with
    an epilogue: EPI_NONE -> y;  EPI_GELU -> (pre, gelu(pre));  EPI_DGELU -> (A x) * gelu'(aux)."""
    B, K, H, W = x.shape
    M = a_mat.shape[1] if transposed else a_mat.shape[0]
    y = torch.empty(B, M, H, W, device=x.device, dtype=_BF16)
    y2 = torch.empty_like(y) if epi == EPI_GELU else None
    bp, bflag = _bias_arg(bias)
    call("ppea_pwconv_ex_bf16", ptr(a_mat, _BF16), ptr(x, _BF16), bp, bflag, epi, ptr(aux, _BF16) if aux is not None
         else None, ptr(y), ptr(y2), B, M, K, H * W, int(transposed), stream_ptr())
    return (y, y2) if epi == EPI_GELU else y


def pwgrad(p, q, want_rowsum=True):
    """(sum_{b,pixels} p[b,m,:] q[b,n,:]  [M,N] fp32,  sum_{b,pixels} p[b,m,:]  [M] fp32 or None)."""
    B, M, H, W = p.shape
    N = q.shape[1]
    ws = torch.empty(_abi.lib.ppea_pwgrad_workspace_bytes(B, M, N, H * W) // 4, device=p.device, dtype=_F32)
    out = torch.empty(M * N + M, device=p.device, dtype=_F32)
    call("ppea_pwgrad_bf16", ptr(p, _BF16), ptr(q, _BF16), ptr(out), ptr(ws), B, M, N, H * W, int(want_rowsum),
         stream_ptr())
    return out[:M * N].view(M, N), (out[M * N:] if want_rowsum else None)


def pwgrad_into(p, q, w_shape, w_dtype, taps=1, b_rows=None, b_dtype=None):
    """Weight (and bias) gradients written straight in parameter layout / dtype by the reduce kernel:
    dW = sum_{b,pixels} p[b,m,:] q[b,n,:] as `w_shape` (taps = 9: p rows are tap-major, dW in Conv2d layout);
    db = row sums of p rows `b_rows` = (first, count)."""
    B, M, H, W = p.shape
    N = q.shape[1]
    ws = torch.empty(_abi.lib.ppea_pwgrad_workspace_bytes(B, M, N, H * W) // 4, device=p.device, dtype=_F32)
    dw = torch.empty(w_shape, device=p.device, dtype=w_dtype)
    db = torch.empty(b_rows[1], device=p.device, dtype=b_dtype) if b_rows is not None else None
    call("ppea_pwgrad_ex_bf16", ptr(p, _BF16), ptr(q, _BF16), ptr(ws), B, M, N, H * W, ptr(dw), int(w_dtype == _BF16),
         taps, ptr(db), int(b_dtype == _BF16), b_rows[0] if b_rows else 0, b_rows[1] if b_rows else 0, stream_ptr())
    return dw, db


PWGRAD_PAIR = __import__("os").environ.get("PPEA_PWGRAD_PAIR", "1") == "1"


def pwgrad_into_pair(a, b):
    """Two pwgrad_into problems over the same pixels -- a, b = (p, q, w_shape, w_dtype, taps, b_rows, b_dtype) -- in ONE GEMM
    launch and ONE reduce launch (an adapter's D_fc2 and D_fc1 weight gradients); falls back to two calls where the pair
    form is not served.  -> ((dw_a, db_a), (dw_b, db_b))."""
    (pa, qa), (pb, qb) = a[:2], b[:2]
    B, _, H, W = pa.shape
    if PWGRAD_PAIR and pb.shape[0] == B and pb.shape[2:] == pa.shape[2:]:
        Ms = (_ct.c_int * 2)(pa.shape[1], pb.shape[1])
        Ns = (_ct.c_int * 2)(qa.shape[1], qb.shape[1])
        nbytes = _abi.lib.ppea_pwgrad_pair_workspace_bytes(B, Ms, Ns, H * W)
        if nbytes > 0:
            dev = pa.device
            ws = torch.empty(nbytes // 4, device=dev, dtype=_F32)
            outs = []
            for (_p, _q, w_shape, w_dtype, taps, b_rows, b_dtype) in (a, b):
                outs.append((torch.empty(w_shape, device=dev, dtype=w_dtype),
                             torch.empty(b_rows[1], device=dev, dtype=b_dtype) if b_rows is not None else None))
            i2 = _ct.c_int * 2

            def vp2(u, v, dt=None):                       # array of two device pointers (None -> NULL)
                return (_ct.c_void_p * 2)(*(None if t_ is None else ptr(t_, dt).value for t_ in (u, v)))
            rows = [x[5] if x[5] is not None else (0, 0) for x in (a, b)]
            call("ppea_pwgrad_ex_pair_bf16", vp2(pa, pb, _BF16), vp2(qa, qb, _BF16), ptr(ws), B,
                 Ms, Ns, H * W, vp2(outs[0][0], outs[1][0]), i2(int(a[3] == _BF16), int(b[3] == _BF16)),
                 i2(a[4], b[4]), vp2(outs[0][1], outs[1][1]), i2(int(a[6] == _BF16), int(b[6] == _BF16)),
                 i2(rows[0][0], rows[1][0]), i2(rows[0][1], rows[1][1]), stream_ptr())
            return outs[0], outs[1]
    return pwgrad_into(*a), pwgrad_into(*b)


class _PoseMatrix(torch.autograd.Function):
    """A15 `transformation_from_parameters` (layers.py:26-42, 61-100): axis-angle [B,3] + translation [B,3] -> [B,4,4], one
    launch forward and one backward (exact Jacobian) instead of ~25 + ~50 tiny element-wise kernels on the step's stream."""

    @staticmethod
    def forward(ctx, aa, tr, invert):
        aa, tr = aa.contiguous(), tr.contiguous()
        B = aa.shape[0]
        T = torch.empty(B, 4, 4, device=aa.device, dtype=_F32)
        call("ppea_pose_matrix_fwd_f32", ptr(aa, _F32), ptr(tr, _F32), ptr(T), B, int(invert), stream_ptr())
        ctx.save_for_backward(aa, tr)
        ctx.invert = int(invert)
        return T

    @staticmethod
    def backward(ctx, dT):
        aa, tr = ctx.saved_tensors
        B = aa.shape[0]
        daa, dtr = torch.empty_like(aa), torch.empty_like(tr)
        call("ppea_pose_matrix_bwd_f32", ptr(aa), ptr(tr), ptr(dT.contiguous().float()), ptr(daa), ptr(dtr), B, ctx.invert,
             stream_ptr())
        return daa, dtr, None


def pose_matrix(axisangle, translation, invert=False):
    """axisangle, translation [B,1,3] (or [B,3]) fp32 on the device -> [B,4,4] fp32."""
    return _PoseMatrix.apply(axisangle.reshape(-1, 3).float(), translation.reshape(-1, 3).float(), bool(invert))


def tapsum_fwd(T, bias, Ch):
    B, _, H, W = T.shape
    pre = torch.empty(B, Ch, H, W, device=T.device, dtype=_BF16)
    h = torch.empty_like(pre)
    bp, bflag = _bias_arg(bias)
    call("ppea_tapsum_fwd_bf16", ptr(T, _BF16), bp, bflag, ptr(pre), ptr(h), B, Ch, H, W, stream_ptr())
    return pre, h


def tapsum_bwd(g):
    B, Ch, H, W = g.shape
    dT = torch.empty(B, 9 * Ch, H, W, device=g.device, dtype=_BF16)
    call("ppea_tapsum_bwd_bf16", ptr(g, _BF16), ptr(dT), B, Ch, H, W, stream_ptr())
    return dT


def adapter_supported(x, hidden):
    """Any hidden width is served: mlp_adapter / conv_adapter zero-pad it to the next multiple of 32 (RepLKNet-31L stage 0:
    hidden 48; the Stage-2 decoder adapter: 148)."""
    B, C, H, W = x.shape
    return (x.is_cuda and x.dtype == _BF16 and C % 32 == 0 and hidden >= 1 and (H * W) % 8 == 0 and W % 4 == 0)


def _pad_hidden(w1, b1, w2):
    """Zero rows for D_fc1 / zero columns for D_fc2 up to a hidden width the GEMM kernels take (a multiple of 32).  The
    padded units compute gelu(0) = 0 and meet zero columns: same function, same gradients (autograd slices the padded
    gradients back); three small concatenations per call."""
    Ch = w1.shape[0]
    pad = (-Ch) % 32
    if pad == 0:
        return w1, b1, w2
    w1p = torch.cat([w1, w1.new_zeros((pad,) + tuple(w1.shape[1:]))], 0)
    b1p = None if b1 is None else torch.cat([b1, b1.new_zeros(pad)], 0)
    w2p = torch.cat([w2, w2.new_zeros(w2.shape[0], pad)], 1)
    return w1p, b1p, w2p


def _grad_dtype(t):
    return t.dtype if t.dtype in (_BF16, _F32) else _F32


class _MlpAdapterFn(torch.autograd.Function):
    """`Adapter` (rka.py:20-47): y = W2 gelu(W1 x + b1) + b2 over the channel axis.  No weight repacking: the
    data gradients consume the forward matrices through the kernel's transposed-A mode, and the weight / bias
    gradients leave the reduce kernel in parameter dtype."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        x = x.contiguous()
        w1m, w2m = w1.to(_BF16).contiguous(), w2.to(_BF16).contiguous()
        pre, h = pwconv_ex(w1m, x, b1, EPI_GELU)
        y = pwconv_ex(w2m, h, b2)
        ctx.save_for_backward(x, pre, h, w1m, w2m)
        ctx.dtypes = tuple(_grad_dtype(t) for t in (w1, b1, w2, b2))
        ctx.shapes = (w1.shape, w2.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, pre, h, w1m, w2m = ctx.saved_tensors
        t = ctx.dtypes
        C, Ch = w2m.shape
        dy = dy.contiguous()
        g = pwconv_ex(w2m, dy, None, EPI_DGELU, pre, transposed=True)          # W2^T dy: At = W2 [C][Ch]
        (dw2, db2), (dw1, db1) = pwgrad_into_pair((dy, h, ctx.shapes[1], t[2], 1, (0, C), t[3]),
                                                  (g, x, ctx.shapes[0], t[0], 1, (0, Ch), t[1]))
        dx = pwconv_ex(w1m, g, transposed=True) if ctx.needs_input_grad[0] else None
        return dx, dw1, db1, dw2, db2


class _ConvAdapterFn(torch.autograd.Function):
    """`B_Adapter`, adpt_test 4 (rka.py:49-109): y = W2 gelu(conv3x3(x; W1) + b1) + b2.  One repack per step
    (W1 -> tap-major [9*Ch][C]); its transpose for the data gradient is the kernel's transposed-A mode."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        x = x.contiguous()
        Ch, C = w1.shape[0], w1.shape[1]
        a1 = w1.to(_BF16).permute(2, 3, 0, 1).reshape(9 * Ch, C).contiguous()     # rows t*Ch + m
        w2m = w2.to(_BF16).contiguous()
        T = pwconv_ex(a1, x)
        pre, h = tapsum_fwd(T, b1, Ch)
        y = pwconv_ex(w2m, h, b2)
        ctx.save_for_backward(x, pre, h, a1, w2m)
        ctx.dtypes = tuple(_grad_dtype(t) for t in (w1, b1, w2, b2))
        ctx.shapes = (w1.shape, w2.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, pre, h, a1, w2m = ctx.saved_tensors
        t = ctx.dtypes
        C, Ch = w2m.shape
        dy = dy.contiguous()
        g = pwconv_ex(w2m, dy, None, EPI_DGELU, pre, transposed=True)
        dT = tapsum_bwd(g)
        (dw2, db2), (dw1, db1) = pwgrad_into_pair((dy, h, ctx.shapes[1], t[2], 1, (0, C), t[3]),
                                                  (dT, x, ctx.shapes[0], t[0], 9, (4 * Ch, Ch), t[1]))   # centre-tap rows == g
        dx = pwconv_ex(a1, dT, transposed=True) if ctx.needs_input_grad[0] else None  # At = a1 [9*Ch][C]
        return dx, dw1, db1, dw2, db2


def mlp_adapter(x, w1, b1, w2, b2):
    w1, b1, w2 = _pad_hidden(w1, b1, w2)
    return _MlpAdapterFn.apply(x, w1, b1, w2, b2)


def conv_adapter(x, w1, b1, w2, b2):
    w1, b1, w2 = _pad_hidden(w1, b1, w2)
    return _ConvAdapterFn.apply(x, w1, b1, w2, b2)


class _PwLinear(torch.autograd.Function):
    """y[b, :, p] = W x[b, :, p] (+ bias) over the channel axis of NCHW bf16 x with a TRAINABLE W [M, K]: forward and data
    gradient on pwconv (the data gradient consumes W through the kernel's transposed-A mode), weight / bias gradients on
    pwgrad, written in parameter dtype."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        wm = w.to(_BF16).contiguous()
        y = pwconv_ex(wm, x, b)
        ctx.save_for_backward(x, wm)
        ctx.meta = (tuple(w.shape), _grad_dtype(w), None if b is None else _grad_dtype(b))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wm = ctx.saved_tensors
        wshape, wdt, bdt = ctx.meta
        dy = dy.contiguous().to(_BF16)
        dx = pwconv_ex(wm, dy, transposed=True) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1] or (bdt is not None and ctx.needs_input_grad[2]):
            dw, db = pwgrad_into(dy, x, wshape, wdt, 1, None if bdt is None else (0, wshape[0]), bdt)
        return dx, dw, db


def pw_linear_supported(x, M):
    B, K, H, W = x.shape
    return x.is_cuda and x.dtype == _BF16 and K % 32 == 0 and M % 8 == 0 and (H * W) % 8 == 0


def pw_linear(x, w, b=None):
    """nn.Linear(K -> M) over the channel axis of x [B,K,H,W] bf16 -> [B,M,H,W] (trainable weight and bias)."""
    return _PwLinear.apply(x, w, b)


# ---------------------------------------------------------------------------------------------
# fp32 dense convolutions / linear layers on the fp32 matrix cores (csrc/conv_f32.hip): the fp32 (parity, BASELINE config 1)
# step's replacement for the library convolution / GEMM behind every groups == 1 conv, nn.Linear and ConvTranspose2d of the
# path (rka.py:20-109, 264-326; depth_decoder_v2.py:172-245; resnet_encoder.py:367-409; pose_decoder.py:27-52).  Operands
# are addressed through element strides: NCHW and channels_last tensors go in as they are.
# ---------------------------------------------------------------------------------------------
CONV_F32_MFMA = True       # False: the library (tests compare the two)


def _strides(t):
    return (_ct.c_long * 4)(*[int(v) for v in t.stride()])


def _like_format(x, shape):
    cl = x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
    return torch.empty(shape, device=x.device, dtype=_F32, memory_format=torch.channels_last if cl else torch.contiguous_format)


def _f32_dense(x):
    """A 4-D fp32 HIP tensor whose strides the kernels may use as they are (any dense or expanded layout)."""
    return x if all(st >= 0 for st in x.stride()) else x.contiguous()


def _gen_sfx(x):
    return "bf16" if x.dtype == _BF16 else "f32"


def _gen_out(x, shape):
    cl = x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
    return torch.empty(shape, device=x.device, dtype=x.dtype, memory_format=torch.channels_last if cl else torch.contiguous_format)


def _f32_vec(b):
    return None if b is None else ptr(b.detach().float().contiguous(), _F32)


class _Conv2dF32(torch.autograd.Function):
    """x, w both fp32 or both bf16 (the caller casts); bias any float dtype (read as fp32)."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad):
        x = _f32_dense(x)
        N, Cin, H, W = x.shape
        Cout, _, R, S = w.shape
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
        wd = w.detach().to(x.dtype).contiguous()
        y = _gen_out(x, (N, Cout, Ho, Wo))
        call(f"ppea_conv2d_{_gen_sfx(x)}_fwd", _raw(x), _strides(x), ptr(wd), _f32_vec(bias), _raw(y), _strides(y), N, Cin, H, W,
             Cout, R, S, stride, pad, stream_ptr())
        ctx.save_for_backward(x, wd)
        ctx.cfg = (stride, pad, Ho, Wo, None if bias is None else bias.dtype, w.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wd = ctx.saved_tensors
        stride, pad, Ho, Wo, bdt, wdt = ctx.cfg
        N, Cin, H, W = x.shape
        Cout, _, R, S = wd.shape
        dy = _f32_dense(dy.to(x.dtype))
        sfx = _gen_sfx(x)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _gen_out(x, x.shape)
            call(f"ppea_conv2d_{sfx}_dgrad", _raw(dy), _strides(dy), ptr(wd), _raw(dx), _strides(dx), N, Cin, H, W, Cout, R, S,
                 stride, pad, Ho, Wo, stream_ptr())
        if ctx.needs_input_grad[1]:
            nbytes = _abi.lib.ppea_conv2d_f32_wgrad_workspace_bytes(N, Cin, Cout, R, S, Ho, Wo)
            ws = torch.empty(nbytes // 4, device=x.device, dtype=_F32) if nbytes else None
            dw = torch.empty(Cout, Cin, R, S, device=x.device, dtype=_F32)
            call(f"ppea_conv2d_{sfx}_wgrad", _raw(x), _strides(x), _raw(dy), _strides(dy), ptr(dw), ptr(ws), N, Cin, H, W, Cout,
                 R, S, stride, pad, Ho, Wo, stream_ptr())
            dw = dw.to(wdt)
        if bdt is not None and ctx.needs_input_grad[2]:
            db = dy.float().sum((0, 2, 3)).to(bdt)
        return dx, dw, db, None, None


def conv2d_f32_ok(x, w, stride=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1, padding_mode="zeros"):
    """This call is served by csrc/conv_f32.hip: fp32 tensors outside autocast (the fp32 step), or bf16 activations (the bf16
    step's shapes that its layout-specialised kernels do not take)."""
    if not (CONV_F32_MFMA and x.is_cuda and x.dim() == 4 and groups == 1 and tuple(dilation) == (1, 1)
            and not isinstance(padding, str) and padding_mode == "zeros" and stride[0] == stride[1]
            and padding[0] == padding[1] and w.dtype in (_F32, _BF16)):
        return False
    if x.dtype == _BF16:
        return True
    return x.dtype == _F32 and w.dtype == _F32 and not torch.is_autocast_enabled()


def conv2d_f32(x, w, bias=None, stride=1, pad=0):
    """F.conv2d(x, w, bias, stride, pad), groups = 1, on the fp32 MFMA kernels (fp32, or bf16 storage with fp32 arithmetic)."""
    return _Conv2dF32.apply(x, w, bias, int(stride), int(pad))


def conv2d(x, w, bias=None, stride=1, pad=0):
    """Dense conv of the path: HIP tensors on this build's kernel, anything else through torch."""
    if conv2d_f32_ok(x, w, (stride, stride), (pad, pad)):
        return conv2d_f32(x, w, bias, stride, pad)
    return torch.nn.functional.conv2d(x, w, bias, stride, pad)


class Conv2d(torch.nn.Conv2d):
    """nn.Conv2d of this package: the bf16 step reaches the layout-specialised kernels through `conv_module` /
    `pwconv_frozen`; whatever falls through to `nn.Conv2d.forward` lands HERE -- HIP tensors run on csrc/conv_f32.hip (fp32
    step: every dense conv; bf16 step: shapes the specialised kernels refuse), everything else (CPU tensors) on torch."""

    def _conv_forward(self, input, weight, bias):
        if conv2d_f32_ok(input, weight, self.stride, self.padding, self.dilation, self.groups, self.padding_mode):
            return conv2d_f32(input, weight, bias, self.stride[0], self.padding[0])
        return super()._conv_forward(input, weight, bias)


def channel_linear_f32(x, w, b):
    """nn.Linear(K -> M) over the channel axis of x [B,K,H,W] = a 1x1 convolution."""
    return conv2d_f32(x, w.view(w.shape[0], w.shape[1], 1, 1), b, 1, 0)


class _ConvTransposeF32(torch.autograd.Function):
    """ConvTranspose2d (weight [Cin][Cout][R][S]) = the data gradient of the conv with that weight; its data gradient is that
    conv's forward and its weight gradient that conv's weight gradient with the two activations exchanged."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, out_pad):
        x = _f32_dense(x)
        N, Cin, H, W = x.shape
        _, Cout, R, S = w.shape
        Ho, Wo = (H - 1) * stride - 2 * pad + R + out_pad, (W - 1) * stride - 2 * pad + S + out_pad
        wd = w.detach().to(x.dtype).contiguous()
        y = _gen_out(x, (N, Cout, Ho, Wo))
        call(f"ppea_conv2d_{_gen_sfx(x)}_dgrad", _raw(x), _strides(x), ptr(wd), _raw(y), _strides(y), N, Cout, Ho, Wo, Cin, R, S,
             stride, pad, H, W, stream_ptr())
        if bias is not None:
            y += bias.detach().to(y.dtype).view(1, -1, 1, 1)
        ctx.save_for_backward(x, wd)
        ctx.cfg = (stride, pad, Ho, Wo, None if bias is None else bias.dtype, w.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wd = ctx.saved_tensors
        stride, pad, Ho, Wo, bdt, wdt = ctx.cfg
        N, Cin, H, W = x.shape
        _, Cout, R, S = wd.shape
        dy = _f32_dense(dy.to(x.dtype))
        sfx = _gen_sfx(x)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _gen_out(x, x.shape)
            call(f"ppea_conv2d_{sfx}_fwd", _raw(dy), _strides(dy), ptr(wd), None, _raw(dx), _strides(dx), N, Cout, Ho, Wo, Cin, R, S,
                 stride, pad, stream_ptr())
        if ctx.needs_input_grad[1]:
            nbytes = _abi.lib.ppea_conv2d_f32_wgrad_workspace_bytes(N, Cout, Cin, R, S, H, W)
            ws = torch.empty(nbytes // 4, device=x.device, dtype=_F32) if nbytes else None
            dw = torch.empty(Cin, Cout, R, S, device=x.device, dtype=_F32)
            call(f"ppea_conv2d_{sfx}_wgrad", _raw(dy), _strides(dy), _raw(x), _strides(x), ptr(dw), ptr(ws), N, Cout, Ho, Wo, Cin,
                 R, S, stride, pad, H, W, stream_ptr())
            dw = dw.to(wdt)
        if bdt is not None and ctx.needs_input_grad[2]:
            db = dy.float().sum((0, 2, 3)).to(bdt)
        return dx, dw, db, None, None, None


def conv_transpose_f32_module(m, x):
    """nn.ConvTranspose2d `m` on the fp32 MFMA kernels, or None when this call is not served."""
    if not (CONV_F32_MFMA and x.is_cuda and m.groups == 1 and tuple(m.dilation) == (1, 1) and m.stride[0] == m.stride[1]
            and m.padding[0] == m.padding[1] and m.output_padding[0] == m.output_padding[1]
            and m.kernel_size[0] == m.kernel_size[1] and m.weight.dtype in (_F32, _BF16)
            and (x.dtype == _BF16 or (x.dtype == _F32 and m.weight.dtype == _F32 and not torch.is_autocast_enabled()))):
        return None
    return _ConvTransposeF32.apply(x, m.weight, m.bias, m.stride[0], m.padding[0], m.output_padding[0])


# ---------------------------------------------------------------------------------------------
# A13  transposed convolution of the Stage-2 decoder adapter (depth_decoder_v2.py:137-139: ConvTranspose2d(c, c, 3, 2, 1,
# output_padding=1)) on the implicit-GEMM kernels: the forward IS the data gradient of a stride-2 conv, its data gradient is
# that conv's forward, its weight gradient that conv's weight gradient with the two activations exchanged.
# ---------------------------------------------------------------------------------------------
class _ConvTransposeNhwc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, out_pad):
        x = _as_nhwc(x)
        N, Cin, H, W = x.shape
        _, Cout, R, S = w.shape                                  # ConvTranspose2d weight: [Cin, Cout, R, S]
        Ho, Wo = (H - 1) * stride - 2 * pad + R + out_pad, (W - 1) * stride - 2 * pad + S + out_pad
        # as a conv weight [Cout_c = Cin][Cin_c = Cout][R][S]: the flipped / transposed operand image of its data gradient
        y = conv_nhwc_raw(x, _conv_packed(w, True), None if bias is None else bias.detach().contiguous(), Cout, R, S, 1,
                          R - 1 - pad, False, stride, Ho, Wo, 0, False)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, Ho, Wo, None if bias is None else bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, Ho, Wo, bdt = ctx.cfg
        N, Cin, H, W = x.shape
        _, Cout, R, S = w.shape
        dz = _as_nhwc(dy.to(_BF16))
        db = dz.float().sum((0, 2, 3)).to(bdt) if (bdt is not None and ctx.needs_input_grad[2]) else None
        dx = dw = None
        if ctx.needs_input_grad[0]:                              # = conv2d(dy, w as [Cin][Cout][R][S], stride, pad)
            dx = conv_nhwc_raw(dz, _conv_packed(w, False), None, Cin, R, S, stride, pad, False, 1, H, W, 0, False)
        if ctx.needs_input_grad[1]:
            if Cin % 8 != 0 or Cout % 8 != 0:
                raise _abi.PpeaKernelError("transposed conv weight gradient: channels must be multiples of 8")
            # weight gradient of that conv: "input" = dy [N,Cout,Ho,Wo], "output gradient" = x [N,Cin,H,W]
            ws = torch.empty(_abi.lib.ppea_conv_wgrad_workspace_bytes(N, Cout, Cin, R, S, stride, H, W) // 4, device=dz.device,
                             dtype=_F32)
            gdt = w.dtype if w.dtype in (_BF16, _F32) else _F32
            dw = torch.empty(Cin, Cout, R, S, device=dz.device, dtype=gdt)
            call("ppea_conv_wgrad_nhwc_bf16", _raw(x), _raw(dz), ptr(dw), int(gdt == _BF16), ptr(ws), N, Ho, Wo, Cout, Cin, R, S,
                 stride, pad, 0, H, W, stream_ptr())
        return dx, dw, db, None, None, None


def conv_transpose_module(m, x):
    """nn.ConvTranspose2d `m` on the implicit-GEMM kernels (bf16 step), or None when this call is not served."""
    if not (CONV_MFMA and x.is_cuda and x.dtype == _BF16 and m.groups == 1 and tuple(m.dilation) == (1, 1)
            and m.stride[0] == m.stride[1] and m.stride[0] in (1, 2) and m.kernel_size[0] == m.kernel_size[1] == 3
            and m.padding[0] == m.padding[1] and m.output_padding[0] == m.output_padding[1]
            and x.shape[1] % 8 == 0 and m.out_channels % 8 == 0):
        return None
    return _ConvTransposeNhwc.apply(x, m.weight, m.bias, m.stride[0], m.padding[0], m.output_padding[0])


# ---------------------------------------------------------------------------------------------
# A12 / A14  dense convolutions as implicit GEMMs on the matrix cores (csrc/conv_nhwc.hip, conv_wgrad.hip):
# decoder ConvBlock / Conv3x3 (layers.py:103-135), pose ResNet-18 + PoseDecoder, reduce_conv, stem[0]
# ---------------------------------------------------------------------------------------------
CONV_ACT = {"none": 0, "relu": 1, "elu": 2, "sigmoid": 3}
CONV_MFMA = True       # dense convs of the bf16 step on this build's implicit-GEMM kernels (False: library convs)


def bf16_autocast():
    return torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == _BF16


def conv_module(conv, x, act="none", reflect=False, out_nchw=False):
    """nn.Conv2d `conv` applied through the implicit-GEMM kernels when the bf16 step is running (x bf16, or an fp32
    image under bf16 autocast for the image-fed layers); None when this call is not served (caller uses the library)."""
    if not (CONV_MFMA and x.is_cuda and conv.groups == 1 and tuple(conv.dilation) == (1, 1)
            and conv.stride[0] == conv.stride[1] and conv.stride[0] in (1, 2)
            and conv.kernel_size[0] == conv.kernel_size[1] and conv.kernel_size[0] in (1, 3, 7)       # kernels built
            and not isinstance(conv.padding, str) and conv.padding[0] == conv.padding[1]
            and getattr(conv, "padding_mode", "zeros") == "zeros"):
        return None
    if x.dtype == _F32 and x.shape[1] < 8 and bf16_autocast():
        x = image_to_nhwc(x, 8)
    if x.dtype != _BF16 or x.shape[1] % 8 != 0:
        return None
    if reflect and (x.shape[2] < 3 or x.shape[3] < 3):        # the fold kernel of the data gradient needs a 3 x 3 interior
        return None
    return conv2d_nhwc(x, conv.weight, conv.bias, conv.stride[0], 1 if reflect else conv.padding[0], reflect, act, out_nchw)
_CONV_PACK_CACHE = {}


def _conv_packed(w, flip):
    """bf16 operand image of w [Cout,Cin,R,S] (flip: the data-gradient operand).  Frozen weights are packed once per
    version; trainable ones on every use (the flat Adam kernel updates them through raw pointers)."""
    Cout, Cin, R, S = w.shape
    cacheable = not w.requires_grad
    key = (id(w), int(flip))
    hit = _CONV_PACK_CACHE.get(key) if cacheable else None
    if hit is not None and hit[2]() is w and hit[0] == w._version:
        return hit[1]
    wd = w.detach()
    if wd.dtype not in (_F32, _BF16) or not wd.is_contiguous():
        wd = wd.float().contiguous()
    buf = torch.empty(_abi.lib.ppea_conv_packed_bytes(Cout, Cin, R, S, int(flip)) // 2, dtype=_BF16, device=w.device)
    call("ppea_conv_pack_weights", ptr(wd), int(wd.dtype == _BF16), ptr(buf), Cout, Cin, R, S, int(flip), stream_ptr())
    if cacheable:
        _CONV_PACK_CACHE[key] = (w._version, buf, weakref.ref(w, lambda _r, k=key: _CONV_PACK_CACHE.pop(k, None)))
    return buf


def _nhwc_raw(t):
    """Pointer of a channels_last (or C == 1 / 1x1 degenerate) 4-D tensor's storage."""
    return _ct.c_void_p(t.data_ptr())


def _as_nhwc(t):
    return t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)


# ---------------------------------------------------------------------------------------------
# decoder glue: nearest 2x upsampling + skip concatenation in one pass (channels_last)
# ---------------------------------------------------------------------------------------------
def up2cat_supported(a, b=None):
    ok = (a.is_cuda and a.dim() == 4 and a.dtype in (_F32, _BF16) and a.shape[1] % 8 == 0
          and a.is_contiguous(memory_format=torch.channels_last))
    if ok and b is not None:
        ok = (b.dtype == a.dtype and b.shape[1] % 8 == 0 and b.shape[0] == a.shape[0]
              and tuple(b.shape[2:]) == (2 * a.shape[2], 2 * a.shape[3])
              and b.is_contiguous(memory_format=torch.channels_last))
    return ok


class _Up2Cat(torch.autograd.Function):
    """cat([interpolate(a, scale_factor=2, mode="nearest"), b], 1) on channels_last tensors in ONE pass; backward
    splits the gradient and sums the upsampled part over each 2x2 block (fp32 accumulation) in one pass."""

    @staticmethod
    def forward(ctx, a, b):
        N, C1, h, w = a.shape
        C2 = 0 if b is None else b.shape[1]
        out = torch.empty(N, C1 + C2, 2 * h, 2 * w, device=a.device, dtype=a.dtype, memory_format=torch.channels_last)
        call(f"ppea_nhwc_up2cat_fwd_{_suffix(a)}", _nhwc_raw(a), None if b is None else _nhwc_raw(b), _nhwc_raw(out),
             N, 2 * h, 2 * w, C1, C2, stream_ptr())
        ctx.dims = (N, C1, C2, h, w)
        return out

    @staticmethod
    def backward(ctx, dout):
        N, C1, C2, h, w = ctx.dims
        dout = _as_nhwc(dout)
        da = torch.empty(N, C1, h, w, device=dout.device, dtype=dout.dtype, memory_format=torch.channels_last)
        db = None if C2 == 0 else torch.empty(N, C2, 2 * h, 2 * w, device=dout.device, dtype=dout.dtype,
                                              memory_format=torch.channels_last)
        call(f"ppea_nhwc_up2cat_bwd_{_suffix(dout)}", _nhwc_raw(dout), _nhwc_raw(da), None if db is None else _nhwc_raw(db),
             N, 2 * h, 2 * w, C1, C2, stream_ptr())
        return da, db


def upsample2x_cat(a, b=None):
    return _Up2Cat.apply(a, b)


def conv_nhwc_raw(x, wp, bias, Cout, R, S, stride, pad, reflect, dil, Ho, Wo, act, out_nchw):
    """One launch of the implicit-GEMM kernel.  x: bf16, channels_last storage, logical [N,Cin,H,W]."""
    N, Cin, H, W = x.shape
    if out_nchw:
        y = torch.empty(N, Cout, Ho, Wo, device=x.device, dtype=_BF16)
    else:
        y = torch.empty(N, Cout, Ho, Wo, device=x.device, dtype=_BF16, memory_format=torch.channels_last)
    bp, bflag = _bias_arg(bias)
    call("ppea_conv_nhwc_bf16", _nhwc_raw(x), ptr(wp), bp, bflag, _nhwc_raw(y), N, H, W, Cin, Cout, R, S, stride, pad,
         int(reflect), dil, Ho, Wo, act, int(out_nchw), stream_ptr())
    return y


def conv_supported(x, w):
    return (x.is_cuda and x.dtype == _BF16 and x.dim() == 4 and x.shape[1] % 8 == 0 and w.shape[2] <= 7
            and w.shape[3] <= 7)


_SIDE = {}


def side_stream_of(main):
    """The side stream that belongs to `main` (created on first use): the student's adapters run on it."""
    key = (main.device.index, main.cuda_stream)
    side = _SIDE.get(key)
    if side is None:
        side = _SIDE[key] = torch.cuda.Stream(main.device)
    return side


class _ConvNhwc(torch.autograd.Function):
    """y = act(conv2d(x, w, stride, pad | reflection pad) + bias) -- forward, data gradient and weight gradient on the
    implicit-GEMM kernels; bias gradient and the activation's derivative in one pass before them."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, reflect, act, out_nchw):
        x = _as_nhwc(x)
        N, Cin, H, W = x.shape
        Cout, _, R, S = w.shape
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
        y = conv_nhwc_raw(x, _conv_packed(w, False), None if bias is None else bias.detach().contiguous(), Cout, R, S, stride,
                          pad, reflect, 1, Ho, Wo, act, out_nchw)
        ctx.save_for_backward(x, w, y if act != 0 else None)
        ctx.cfg = (stride, pad, bool(reflect), act, bool(out_nchw), bias is not None,
                   None if bias is None else bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        stride, pad, reflect, act, out_nchw, has_bias, bdt = ctx.cfg
        N, Cin, H, W = x.shape
        Cout, _, R, S = w.shape
        Ho, Wo = dy.shape[2], dy.shape[3]
        db = None
        # ---- dz = dy * act'(y), channels_last bf16; bias gradient ------------------------------------------------------
        if act == 2 and not out_nchw and _nhwc_c_ok(Cout):
            dy = dy.contiguous(memory_format=torch.channels_last).to(_BF16)
            slabs = _abi.lib.ppea_nhwc_bias_elu_slabs(N * Ho * Wo, Cout)
            partial = torch.empty(slabs, Cout, device=dy.device, dtype=_F32)
            dz = torch.empty_like(dy)
            call("ppea_nhwc_bias_elu_bwd_bf16", _raw(dy), _raw(y), _raw(dz), ptr(partial), N * Ho * Wo, Cout, stream_ptr())
            if has_bias and ctx.needs_input_grad[2]:
                db = partial.sum(0).to(bdt)
        else:
            if act == 1:
                dz = dy * (y > 0)
            elif act == 2:
                dz = dy * torch.where(y > 0, torch.ones_like(y), y + 1)
            elif act == 3:
                dz = dy * (y * (1 - y))
            else:
                dz = dy
            if has_bias and ctx.needs_input_grad[2]:
                db = dz.float().sum((0, 2, 3)).to(bdt)
            dz = _as_nhwc(dz.to(_BF16))
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wt = _conv_packed(w, True)                           # [R*S flipped][Cin][Cout padded to 32]
            if reflect:
                # gradient on the reflection-padded domain, then folded back (layers.py:119-135)
                dpad = conv_nhwc_raw(dz, wt, None, Cin, R, S, 1, R - 1, False, stride, Ho + R - 1, Wo + S - 1, 0, False)
                dx = torch.empty(N, Cin, H, W, device=dz.device, dtype=_BF16, memory_format=torch.channels_last)
                call("ppea_nhwc_reflect_pad1_bwd_bf16", _raw(dpad), _raw(dx), N, H, W, Cin, stream_ptr())
            else:
                dx = conv_nhwc_raw(dz, wt, None, Cin, R, S, 1, R - 1 - pad, False, stride, H, W, 0, False)
        if ctx.needs_input_grad[1]:
            if Cout % 8 != 0:
                raise _abi.PpeaKernelError("conv weight gradient: output channels must be a multiple of 8 "
                                           "(pad the layer, see conv2d_nhwc)")
            ws = torch.empty(_abi.lib.ppea_conv_wgrad_workspace_bytes(N, Cin, Cout, R, S, stride, Ho, Wo) // 4, device=dz.device,
                             dtype=_F32)
            gdt = w.dtype if w.dtype in (_BF16, _F32) else _F32
            dw = torch.empty(Cout, Cin, R, S, device=dz.device, dtype=gdt)
            call("ppea_conv_wgrad_nhwc_bf16", _raw(dz), _raw(x), ptr(dw), int(gdt == _BF16), ptr(ws), N, H, W, Cin, Cout, R, S,
                 stride, pad, int(reflect), Ho, Wo, stream_ptr())
        return dx, dw, db, None, None, None, None, None


_IMG_PACK_CACHE = {}


def _image_packed(w):
    Cout, Cin, K, _ = w.shape
    cacheable = not w.requires_grad
    key = id(w)
    hit = _IMG_PACK_CACHE.get(key) if cacheable else None
    if hit is not None and hit[2]() is w and hit[0] == w._version:
        return hit[1]
    wd = w.detach()
    if wd.dtype not in (_F32, _BF16) or not wd.is_contiguous():
        wd = wd.float().contiguous()
    buf = torch.empty(_abi.lib.ppea_conv_image_packed_bytes(Cout, K) // 2, dtype=_BF16, device=w.device)
    call("ppea_conv_image_pack_weights", ptr(wd), int(wd.dtype == _BF16), ptr(buf), Cout, Cin, K, stream_ptr())
    if cacheable:
        _IMG_PACK_CACHE[key] = (w._version, buf, weakref.ref(w, lambda _r, k=key: _IMG_PACK_CACHE.pop(k, None)))
    return buf


class _ConvImage(torch.autograd.Function):
    """stem[0] / pose conv1 on the row-packed image kernels: x [N,8,H,W] bf16 channels_last frames (3 / 6 real channels),
    w [Cout,Cin,K,K], stride 2.  Weight gradient only (the input is the frame)."""

    @staticmethod
    def forward(ctx, x, w, pad, out_nchw):
        N, _, H, W = x.shape
        Cout, Cin, K, _ = w.shape
        Ho, Wo = (H + 2 * pad - K) // 2 + 1, (W + 2 * pad - K) // 2 + 1
        y = torch.empty(N, Cout, Ho, Wo, device=x.device, dtype=_BF16,
                        memory_format=torch.contiguous_format if out_nchw else torch.channels_last)
        call("ppea_conv_image_bf16", _nhwc_raw(x), ptr(_image_packed(w)), _nhwc_raw(y), N, H, W, Cout, K, 2, pad, Ho, Wo,
             int(out_nchw), stream_ptr())
        ctx.save_for_backward(x, w)
        ctx.pad = pad
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        if not ctx.needs_input_grad[1]:
            return None, None, None, None
        N, _, H, W = x.shape
        Cout, Cin, K, _ = w.shape
        dz = _as_nhwc(dy.to(_BF16))
        Ho, Wo = dz.shape[2], dz.shape[3]
        ws = torch.empty(_abi.lib.ppea_conv_image_wgrad_workspace_bytes(N, Cout, K, Ho, Wo) // 4, device=dz.device, dtype=_F32)
        gdt = w.dtype if w.dtype in (_BF16, _F32) else _F32
        dw = torch.empty(Cout, Cin, K, K, device=dz.device, dtype=gdt)
        call("ppea_conv_image_wgrad_bf16", _raw(dz), _raw(x), ptr(dw), int(gdt == _BF16), ptr(ws), N, H, W, Cin, Cout, K, 2,
             ctx.pad, Ho, Wo, stream_ptr())
        return None, dw, None, None


def conv2d_nhwc(x, w, bias=None, stride=1, pad=0, reflect=False, act="none", out_nchw=False):
    """Dense conv on the matrix cores.  x [N,Cin,H,W] bf16 (channels_last storage preferred), w [Cout,Cin,R,S] (bf16 or
    fp32 parameter), -> [N,Cout,Ho,Wo] bf16 channels_last (or NCHW-contiguous with out_nchw).  Layers whose output
    channel count is not a multiple of 8 (disp conv: 1, pose head: 12) run zero-padded to the next multiple."""
    Cout = w.shape[0]
    if (x.shape[1] == 8 and w.shape[1] <= 8 and stride == 2 and w.shape[2] in (3, 7) and w.shape[2] == w.shape[3]
            and bias is None and act == "none" and not reflect and Cout % 8 == 0 and not x.requires_grad):
        return _ConvImage.apply(_as_nhwc(x), w, pad, out_nchw)        # stem[0] / pose conv1: one contraction per filter row
    if w.shape[1] < x.shape[1]:                 # image-fed layers: x was zero-padded to 8 channels (image_to_nhwc)
        w = torch.cat([w, w.new_zeros(Cout, x.shape[1] - w.shape[1], *w.shape[2:])], 1)
    if Cout % 8 != 0:
        padn = 8 - Cout % 8
        w8 = torch.cat([w, w.new_zeros(padn, *w.shape[1:])], 0)
        b8 = None if bias is None else torch.cat([bias, bias.new_zeros(padn)], 0)
        y = _ConvNhwc.apply(x, w8, b8, stride, pad, reflect, CONV_ACT[act], out_nchw)
        return y[:, :Cout]
    return _ConvNhwc.apply(x, w, bias, stride, pad, reflect, CONV_ACT[act], out_nchw)


def image_to_nhwc(x, cp=8, sub=0.0, div=1.0):
    """fp32 NCHW image(s) -> bf16 channels_last with the channels zero-padded to `cp`; y = (x - sub) / div."""
    x = x.contiguous().float()
    N, C, H, W = x.shape
    y = torch.empty(N, cp, H, W, device=x.device, dtype=_BF16, memory_format=torch.channels_last)
    call("ppea_image_to_nhwc_bf16", ptr(x), _nhwc_raw(y), N, C, H, W, cp, float(sub), float(div), stream_ptr())
    return y
