"""RepLKNet-31B/L backbone with PPEA adapters on gfx950 kernels.

Module tree, attribute names and state_dict keys follow the reference
(networks/replknet_adapter.py) because (a) released checkpoints load by key and (b) the freeze
rule is a substring test on parameter names (repdepth.py:47-50).  What differs is execution:
the large-kernel branch pair (k x k + 5 x 5 depthwise) is one fused HIP launch each way, batch
norm is `ppeadepth.batchnorm.BatchNorm2d`, and `use_checkpoint` never recomputes activations
(see batchnorm.py).
"""
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops, rng
from ..batchnorm import BatchNorm2d, fused_bn_act, fused_bn_act_next, second_use

FUSE_BN = True      # training-mode BN + activation + residual glue on the fused HIP kernels
ADAPTER_CHANNELS_LAST = True

use_sync_bn = False


def enable_sync_bn():
    """replknet_adapter.py:170-174: a module-level switch that, once set, stays set."""
    global use_sync_bn
    use_sync_bn = True


def get_bn(channels):
    return BatchNorm2d(channels, sync=use_sync_bn)


ADAPTER_STREAMS = os.environ.get("PPEA_ADAPTER_STREAMS", "1") == "1"   # adapters of the student on a forked side stream
# The second-consumer alias of a block's first BatchNorm output (batchnorm.second_use) is handed to the FORKED adapters too
# (round 4).  Round 3 recorded "wrong encoder gradients in 2 of 3 captures" for this topology and parked it; the cause was
# not a capture race: tools/graph_edges_dup.py reads the captured hipGraph back (every kernel that touches the alias
# gradient's buffer is ordered with the launch that consumes it), and tools/debug_repro.py shows that with the library
# convolutions gone (fp32: csrc/conv_f32.hip) the step is a pure function of (state, inputs, seeds) -- eager, replayed,
# with or without the alias: 0 of 4 220 tensors differ.  What differed in round 3 were MIOpen's fp32 igemm kernels
# (atomic accumulation: different bits on every run), amplified by the chaotic random-weight fixture.  DESIGN 5.
BN_DUP_FORKED = os.environ.get("PPEA_BN_DUP_FORK", "1") == "1"
NO_FORK_ON = set()       # cuda_stream handles on which adapters run inline (already a forked branch)


side_stream_of = ops.side_stream_of      # the adapter side stream that belongs to a stream (shared with the deferred wgrads)


def _forked_adapter(adapter, inp):
    """Run `adapter(inp)` on a side stream forked from the current one; returns (output, join) where join()
    must be called on the current stream before the output is consumed."""
    main = torch.cuda.current_stream()
    side = side_stream_of(main)
    side.wait_stream(main)
    # `inp` was allocated on `main` but the adapter SAVES it for backward, where side-stream kernels read it (the weight
    # gradient of D_fc1) right before autograd drops the last reference: without this mark the allocator hands the
    # block back to `main` while those kernels may still be running (found in round 2: D_fc1.weight.grad differed
    # in the last bits between the forked and the in-line adapter).
    inp.record_stream(side)
    with torch.cuda.stream(side):
        out = adapter(inp)
    # `out` needs no mark: it returns to the side stream's pool only after the consumer on `main` has been enqueued,
    # and the next fork waits for `main` before reusing it.
    return out, lambda: main.wait_stream(side)


class DropPath(nn.Module):
    """timm DropPath (stochastic depth per sample, scale_by_keep) with an injectable RNG."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = rng.bernoulli_keep(x.shape[0], keep, x)
        if keep > 0.0:
            mask = mask / keep
        return x * mask

    def extra_repr(self):
        return f"drop_prob={self.drop_prob:.3f}"


def _drop_mask(drop_path, x):
    """Per-sample DropPath scale [N] (None for Identity / p = 0), drawn where the reference draws it."""
    if not isinstance(drop_path, DropPath) or drop_path.drop_prob == 0.0 or not drop_path.training:
        return None
    planned = drop_path.__dict__.pop("_planned", None)
    if planned is not None and planned.shape[0] == x.shape[0]:
        return planned
    keep = 1.0 - drop_path.drop_prob
    m = rng.bernoulli_keep(x.shape[0], keep, x).reshape(-1).float()
    return m / keep if keep > 0.0 else m


def _plan_drop_masks(net, batch, device, stages=None):
    """Device-RNG mode: the per-sample DropPath scales of ALL blocks of one encoder pass in two launches (one Bernoulli
    draw with per-block keep probabilities, one division) instead of three tiny launches per block.  The reference draws
    block by block from the global generator (timm DropPath); that order only matters to seeded parity runs, which use
    rng mode "reference" and never come here."""
    which = tuple(range(len(net.stages))) if stages is None else tuple(stages)
    dps = [b.drop_path for si in which for b in net.stages[si].blocks
           if isinstance(b.drop_path, DropPath) and b.drop_path.drop_prob > 0.0 and b.drop_path.training]
    if not dps:
        return
    cache = net.__dict__.setdefault("_keep_probs", {})
    keep = cache.get(which)
    if keep is None or keep.device != device or keep.shape[0] != len(dps):
        keep = torch.tensor([1.0 - d.drop_prob for d in dps], dtype=torch.float32, device=device).view(-1, 1)
        cache[which] = keep
    scales = rng.drop_path_scales(keep, batch)                       # [blocks, batch]: 0 or 1 / keep
    for i, d in enumerate(dps):
        d._planned = scales[i]


class LargeKernelDW(nn.Conv2d):
    """Depthwise k x k conv (k > 5, stride 1, pad k//2) on the HIP kernel -- what `get_conv2d`
    returns through the LARGE_KERNEL_CONV_IMPL hook in the reference (rka.py:157-165)."""

    def __init__(self, channels, kernel_size, bias=False):
        super().__init__(channels, channels, kernel_size, 1, kernel_size // 2, 1, channels, bias)

    def forward(self, x):
        y, _ = ops.dwconv_lk(x, self.weight, None)
        if self.bias is not None:
            y = y + self.bias.view(1, -1, 1, 1).to(y.dtype)
        return y


class SmallDW(nn.Conv2d):
    """Depthwise 3x3, stride 1/2, pad 1 (stem / transitions) on the HIP stencil kernel when frozen."""

    def forward(self, x):
        if x.is_cuda and not self.weight.requires_grad and self.bias is None:
            return ops.dwconv3x3(x, self.weight, self.stride[0])
        return super().forward(x)


class ImageConv(ops.Conv2d):
    """stem[0] (rka.py:411): dense 3x3 stride-2 conv on the RGB frame.  On the bf16 step: frame -> channels_last bf16
    padded to 8 channels, then this build's implicit-GEMM kernel writing NCHW for the trunk."""

    def forward(self, x):
        y = ops.conv_module(self, x, out_nchw=True) if x.is_cuda else None
        return super().forward(x) if y is None else y


class PointwiseConv(ops.Conv2d):
    """1x1 stride-1 conv; frozen + bf16 activations run on the NCHW MFMA kernel (forward with W, data
    gradient with W^T), everything else on the library conv."""

    def forward(self, x):
        if PW_MFMA and x.is_cuda and x.dtype == torch.bfloat16 and not self.weight.requires_grad:
            y = ops.pwconv_frozen(x, self.weight)
            if y is not None:
                return y
        return super().forward(x)

    def forward_sums(self, x, always=False):
        """-> (conv(x), sums or None): on the MFMA path, for activations whose BatchNorm takes separate statistics and
        apply launches (N * HW > 16 384: stages 0 / 1; `always`: any size), the GEMM's epilogue also returns the per-channel
        partial sums that BatchNorm needs (`fused_bn_act(..., sums=sums)`)."""
        from ..batchnorm import _collectives_on
        # (several ranks: BatchNorm statistics are a launch of their own at every size -- the all-gather sits between them
        # and the apply launch -- so the epilogue sums replace a pass over the activation at stages 2 / 3 as well)
        if (PW_MFMA and BN_SUMS and x.is_cuda and x.dtype == torch.bfloat16 and not self.weight.requires_grad
                and (always or x.shape[0] * x.shape[2] * x.shape[3] > 16384 or _collectives_on())):
            r = ops.pwconv_frozen(x, self.weight, want_sums=True)
            if r is not None:
                return r
        return self.forward(x), None


PW_MFMA = True
# ... and from the large-kernel depthwise conv's epilogue: measured neutral for the step, but the 16 conversions + FMAs per
# tile sit in the hand-pipelined kernel's instruction stream (31x31 forward 74 -> 82 us), so it is opt-in (PPEA_DW_SUMS=1)
DW_SUMS = os.environ.get("PPEA_DW_SUMS", "0") == "1"
# the BatchNorm + ReLU between pw1 and the large kernel applied in the depthwise kernel's staging pass (one rank)
DW_BN_FUSE = os.environ.get("PPEA_DW_BN", "1") == "1"
BN_SUMS = os.environ.get("PPEA_BN_SUMS", "1") == "1"   # BatchNorm statistics from the 1x1 conv's epilogue (stages 0 / 1)
BN_SUMS_FFN = os.environ.get("PPEA_BN_SUMS_FFN", "1") == "1"   # ... for the FFN's hidden BatchNorm + GELU at stages 2 / 3 too
ADAPTER_MFMA = True    # adapters (forward + every gradient) on the NCHW MFMA kernels under bf16


def get_conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias):
    k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
    if k == 1 and stride == 1 and padding == 0 and groups == 1 and dilation == 1 and not bias:
        return PointwiseConv(in_channels, out_channels, 1, 1, 0, 1, 1, False)
    if (in_channels == out_channels == groups and k == 3 and stride in (1, 2) and padding == 1 and dilation == 1
            and groups > 1):
        return SmallDW(in_channels, out_channels, 3, stride, 1, 1, groups, bias)
    if (in_channels == out_channels == groups and k > 5 and stride == 1 and padding == k // 2
            and dilation == 1):
        return LargeKernelDW(in_channels, k, bias=bias)
    if groups == 1 and k == 3 and in_channels < 8 and dilation == 1:
        return ImageConv(in_channels, out_channels, 3, stride, padding, dilation, 1, bias)
    cls = ops.Conv2d if groups == 1 else nn.Conv2d           # (dense: fp32 steps run on csrc/conv_f32.hip)
    return cls(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)


class ConvBNAct(nn.Sequential):
    """nn.Sequential(conv, bn[, nonlinear]) of the reference (same child names / state_dict keys); in
    training mode the BN(+ReLU) runs as one fused statistics pass and one fused apply pass."""

    def forward(self, x):
        if not (FUSE_BN and self.training and x.is_cuda):
            return super().forward(x)
        act = ops.ACT_RELU if hasattr(self, "nonlinear") else ops.ACT_NONE
        if isinstance(self.conv, PointwiseConv):
            z, sums = self.conv.forward_sums(x)
            return fused_bn_act(z, self.bn, act=act, sums=sums)
        return fused_bn_act(self.conv(x), self.bn, act=act)


def conv_bn(in_channels, out_channels, kernel_size, stride, padding, groups, dilation=1):
    if padding is None:
        padding = kernel_size // 2
    seq = ConvBNAct()
    seq.add_module("conv", get_conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation,
                                      groups, False))
    seq.add_module("bn", get_bn(out_channels))
    return seq


def conv_bn_relu(in_channels, out_channels, kernel_size, stride, padding, groups, dilation=1):
    seq = conv_bn(in_channels, out_channels, kernel_size, stride, padding, groups, dilation)
    seq.add_module("nonlinear", nn.ReLU())
    return seq


def fuse_bn(conv, bn):
    """rka.py:199-208: fold eval-mode BN into the preceding conv's kernel and bias."""
    std = (bn.running_var + bn.eps).sqrt()
    t = (bn.weight / std).reshape(-1, 1, 1, 1)
    return conv.weight * t, bn.bias - bn.running_mean * bn.weight / std


def channel_linear(x, lin):
    """nn.Linear `lin` applied over the channel axis of x [B,C,H,W] -> [B,C_out,H,W]."""
    B, C, H, W = x.shape
    if ops.conv2d_f32_ok(x, lin.weight):                    # fp32 step: a 1x1 convolution on the fp32 MFMA kernel
        return ops.channel_linear_f32(x, lin.weight, lin.bias)
    w = lin.weight.unsqueeze(0).expand(B, -1, -1)          # stride-0 batch: one strided-batched GEMM, no copy
    xf = x.flatten(2)
    if lin.bias is not None:
        y = torch.baddbmm(lin.bias[None, :, None], w, xf)
    else:
        y = torch.bmm(w, xf)
    return y.view(B, -1, H, W)


class Adapter(nn.Module):
    """rka.py:20-47: token-wise MLP adapter Linear -> GELU -> Linear on [B,HW,C]."""

    def __init__(self, D_features, adpt_test=4, mlp_ratio=0.25, act_layer=nn.GELU):
        super().__init__()
        hidden = int(D_features * mlp_ratio)
        self.feats, self.test_id = D_features, adpt_test
        self.act = act_layer()
        self.D_fc1 = nn.Linear(D_features, hidden)
        self.D_fc2 = nn.Linear(hidden, D_features)

    def forward(self, x):
        if ADAPTER_MFMA and ops.adapter_supported(x, self.D_fc1.out_features):
            return ops.mlp_adapter(x, self.D_fc1.weight, self.D_fc1.bias, self.D_fc2.weight, self.D_fc2.bias)
        # Linear over the channel axis of [B,HW,C] == W @ x[B,C,HW] on NCHW: no permute / reshape copies,
        # one strided-batched GEMM per projection (weight broadcast over the batch)
        return channel_linear(self.act(channel_linear(x, self.D_fc1)), self.D_fc2)


class B_Adapter(nn.Module):
    """rka.py:49-109.  adpt_test == 4 (the default): Conv3x3(C -> C/4) -> GELU -> Linear(C/4 -> C)."""

    def __init__(self, D_features, adpt_test=4, mlp_ratio=0.25, act_layer=nn.GELU):
        super().__init__()
        hidden = int(D_features * mlp_ratio)
        self.feats, self.test_id = D_features, adpt_test
        self.act = act_layer()
        if adpt_test in (1, 2):
            self.D_fc1 = nn.Linear(D_features, hidden)
            self.D_fc2 = nn.Linear(hidden, D_features)
        elif adpt_test == 4:
            self.D_fc1 = ops.Conv2d(D_features, hidden, 3, 1, 1)
            self.D_fc2 = nn.Linear(hidden, D_features)
        else:
            self.D_fc1 = ops.Conv2d(D_features, hidden, 3, 1, 1)
            self.D_fc2 = ops.Conv2d(hidden, D_features, 3, 1, 1)

    def forward(self, x):
        B, C, H, W = x.shape
        if ADAPTER_MFMA and self.test_id in (1, 2, 4) and ops.adapter_supported(x, self.D_fc2.in_features):
            fn = ops.conv_adapter if self.test_id == 4 else ops.mlp_adapter
            return fn(x, self.D_fc1.weight, self.D_fc1.bias, self.D_fc2.weight, self.D_fc2.bias)
        if self.test_id == 4:
            if ADAPTER_CHANNELS_LAST and x.is_cuda and not ops.conv2d_f32_ok(x, self.D_fc1.weight):
                # whole branch in NHWC: MIOpen's implicit-GEMM kernels are NHWC-native (no per-conv layout
                # round trips), and Linear over channels is then a plain [B*HW, C/4] x [C/4, C] GEMM
                xc = x.contiguous(memory_format=torch.channels_last)
                h = self.act(self.D_fc1(xc))                                   # [B,C/4,H,W] channels_last
                y = F.linear(h.permute(0, 2, 3, 1), self.D_fc2.weight, self.D_fc2.bias)    # [B,H,W,C]
                return y.permute(0, 3, 1, 2).contiguous()
            return channel_linear(self.act(self.D_fc1(x)), self.D_fc2)
        if self.test_id in (1, 2):
            x = x.flatten(2).permute(0, 2, 1)
        h = self.D_fc1(x)
        if self.test_id == 4:
            h = h.flatten(2).permute(0, 2, 1)
        h = self.D_fc2(self.act(h))
        if self.test_id > 0:
            h = h.permute(0, 2, 1).reshape(B, -1, H, W)
        return h


class InputAdapter(nn.Module):
    """rka.py:134-149 (`--input` / `--mono_input`): Conv3x3 s2 (C -> C/4) -> BN -> GELU -> Conv3x3 (C/4 -> C) -> BN on the
    output of stem[0], added to the output of the stem.  Plain (not Sync) BatchNorm in the reference."""

    def __init__(self, D_features, mlp_ratio=0.25, act_layer=nn.GELU):
        super().__init__()
        hidden = int(D_features * mlp_ratio)
        self.D_fc1 = ops.Conv2d(D_features, hidden, 3, 2, 1)
        self.D_fc2 = ops.Conv2d(hidden, D_features, 3, 1, 1)
        self.act = act_layer()
        self.bn1 = BatchNorm2d(hidden)
        self.bn2 = BatchNorm2d(D_features)

    def forward(self, x):
        if FUSE_BN and self.training and x.is_cuda:
            h = fused_bn_act(self.D_fc1(x), self.bn1, act=ops.ACT_GELU)
            return fused_bn_act(self.D_fc2(h), self.bn2)
        return self.bn2(self.D_fc2(self.act(self.bn1(self.D_fc1(x)))))


class ReparamLargeKernelConv(nn.Module):
    """rka.py:210-261.  Training form: BN(DW_k(x)) + BN(DW_5(x)); both depthwise convs come out of one
    launch that stages x in LDS once."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, groups, small_kernel,
                 small_kernel_merged=False):
        super().__init__()
        self.kernel_size, self.small_kernel = kernel_size, small_kernel
        pad = kernel_size // 2
        if small_kernel_merged:
            self.lkb_reparam = get_conv2d(in_channels, out_channels, kernel_size, stride, pad, 1, groups, True)
        else:
            self.lkb_origin = conv_bn(in_channels, out_channels, kernel_size, stride, pad, groups)
            if small_kernel is not None:
                assert small_kernel <= kernel_size
                self.small_conv = conv_bn(in_channels, out_channels, small_kernel, stride, small_kernel // 2,
                                          groups)

    def forward(self, x):
        if hasattr(self, "lkb_reparam"):
            return self.lkb_reparam(x)
        if not hasattr(self, "small_conv"):
            return self.lkb_origin(x)
        return self.forward_act(x, ops.ACT_NONE)

    def forward_act(self, x, act):
        """act(BN(DW_k(x)) + BN(DW_5(x))): one conv launch + fused BN/activation."""
        big, small = self.lkb_origin.conv, self.small_conv.conv
        if isinstance(big, LargeKernelDW) and small.kernel_size[0] in (3, 5) and small.stride[0] == 1:
            if (FUSE_BN and BN_SUMS and DW_SUMS and self.training and x.is_cuda and x.dtype == torch.bfloat16
                    and x.shape[0] * x.shape[2] * x.shape[3] > 16384):
                # stages 0 / 1: the conv's epilogue also leaves the per-channel sums the two BatchNorms need
                y_big, y_small, sums = ops.dwconv_lk(x, big.weight, small.weight, want_sums=True)
                return fused_bn_act(y_big, self.lkb_origin.bn, y_small, self.small_conv.bn, act=act,
                                    sums=None if sums is None else (sums[0], sums[1]))
            y_big, y_small = ops.dwconv_lk(x, big.weight, small.weight)
            if FUSE_BN and self.training and x.is_cuda:
                return fused_bn_act(y_big, self.lkb_origin.bn, y_small, self.small_conv.bn, act=act)
            out = self.lkb_origin.bn(y_big) + self.small_conv.bn(y_small)
        else:
            out = self.lkb_origin(x) + self.small_conv(x)
        return F.relu(out) if act == ops.ACT_RELU else out

    def get_equivalent_kernel_bias(self):
        eq_k, eq_b = fuse_bn(self.lkb_origin.conv, self.lkb_origin.bn)
        if hasattr(self, "small_conv"):
            small_k, small_b = fuse_bn(self.small_conv.conv, self.small_conv.bn)
            eq_b = eq_b + small_b
            eq_k = eq_k + F.pad(small_k, [(self.kernel_size - self.small_kernel) // 2] * 4)
        return eq_k, eq_b

    def merge_kernel(self):
        """rka.py:250-261 (inference-time structural re-parameterisation)."""
        eq_k, eq_b = self.get_equivalent_kernel_bias()
        c = self.lkb_origin.conv
        self.lkb_reparam = get_conv2d(c.in_channels, c.out_channels, c.kernel_size, c.stride[0],
                                      c.padding[0], c.dilation[0], c.groups, True)
        self.lkb_reparam.weight.data = eq_k.detach()
        self.lkb_reparam.bias.data = eq_b.detach()
        del self.lkb_origin
        if hasattr(self, "small_conv"):
            del self.small_conv


def _conv_sums(conv, x, always=False):
    """(conv(x), partial sums for the BatchNorm that follows or None)."""
    return conv.forward_sums(x, always) if isinstance(conv, PointwiseConv) else (conv(x), None)


class ConvFFN(nn.Module):
    """rka.py:264-289."""

    def __init__(self, in_channels, internal_channels, out_channels, drop_path, gamma=1.0, adpt_test=0,
                 ratio=0.25):
        super().__init__()
        self.inline_adapter = False        # True: the adapter runs on the block's own stream (RepDepth: stage 0 with POSE_SIDE)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.preffn_bn = get_bn(in_channels)
        self.pw1 = conv_bn(in_channels, internal_channels, 1, 1, 0, 1)
        self.pw2 = conv_bn(internal_channels, out_channels, 1, 1, 0, 1)
        self.nonlinear = nn.GELU()
        if adpt_test >= 0:
            self.mlp_adapter = Adapter(in_channels, adpt_test, 0.5 if adpt_test == 2 else 0.25)
        self.gamma, self.test_id = gamma, adpt_test

    pre_bn = property(lambda self: self.preffn_bn)

    def forward(self, x, pre_out=None, next_bn=None):
        """pre_out: preffn_bn(x), already computed by the previous block's last launch; next_bn: the following block's
        first BatchNorm -- computed here, in this block's last launch: -> (y, next_bn(y)) (see fused_bn_act_next)."""
        if FUSE_BN and self.training and x.is_cuda:
            if pre_out is not None:
                out = pre_out
            else:
                out, x = fused_bn_act(x, self.preffn_bn, skip=True)    # x: the same tensor, routed for the residual use
            adpt, join = None, None
            if self.test_id >= 0:
                if ADAPTER_STREAMS and not self.inline_adapter and torch.cuda.current_stream().cuda_stream not in NO_FORK_ON:
                    adpt, join = _forked_adapter(self.mlp_adapter, second_use(out) if BN_DUP_FORKED else out)
                else:
                    # hand the adapter the alias of `out` whose gradient joins pw1's inside the BatchNorm's backward launch
                    # (batchnorm.second_use)
                    adpt = self.mlp_adapter(second_use(out))
            # (the BatchNorm + GELU over the 4C-wide hidden tensor takes its statistics from the GEMM's epilogue at every size)
            z1, s1 = _conv_sums(self.pw1.conv, out, BN_SUMS_FFN)
            h = fused_bn_act(z1, self.pw1.bn, act=ops.ACT_GELU, sums=s1)
            z, s2 = _conv_sums(self.pw2.conv, h)
            if join is not None:
                join()
            mask = _drop_mask(self.drop_path, x)
            if next_bn is not None:
                pair = fused_bn_act_next(z, self.pw2.bn, next_bn, mask=mask, r1=x, r2=adpt, r2_scale=self.gamma, sums=s2)
                if pair is not None:
                    return pair
            y = fused_bn_act(z, self.pw2.bn, mask=mask, r1=x, r2=adpt, r2_scale=self.gamma, sums=s2)
            return y if next_bn is None else (y, None)
        assert pre_out is None and next_bn is None
        out = self.preffn_bn(x)
        adpt = self.mlp_adapter(out) if self.test_id >= 0 else None
        out = self.pw2(self.nonlinear(self.pw1(out)))
        y = x + self.drop_path(out)
        return y if adpt is None else y + self.gamma * adpt


class RepLKBlock(nn.Module):
    """rka.py:292-326."""

    def __init__(self, in_channels, dw_channels, block_lk_size, small_kernel, drop_path, gamma=1.0,
                 small_kernel_merged=False, adpt_test=0, ratio=0.25):
        super().__init__()
        self.inline_adapter = False        # True: the adapter runs on the block's own stream (RepDepth: stage 0 with POSE_SIDE)
        self.pw1 = conv_bn_relu(in_channels, dw_channels, 1, 1, 0, 1)
        self.pw2 = conv_bn(dw_channels, in_channels, 1, 1, 0, 1)
        self.large_kernel = ReparamLargeKernelConv(dw_channels, dw_channels, block_lk_size, 1, dw_channels,
                                                   small_kernel, small_kernel_merged)
        self.lk_nonlinear = nn.ReLU()
        self.prelkb_bn = get_bn(in_channels)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        if adpt_test >= 0:
            self.adapter = B_Adapter(in_channels, adpt_test, ratio)
        self.gamma, self.test_id = gamma, adpt_test

    pre_bn = property(lambda self: self.prelkb_bn)

    def _pw1_large_kernel_fused(self, out):
        """relu(BN(DW_k(t)) + BN(DW_5(t))) with t = relu(BN(pw1(out))) applied inside the depthwise kernel's staging pass:
        pw1's GEMM leaves the statistics' partial sums, the depthwise kernel finalises and applies them -- the conv_bn_relu
        between the two (rka.py:305-308) costs no launch and no pass over the activation.  One rank, bf16 step; else None."""
        from .. import batchnorm as bnm
        lk = self.large_kernel
        big, small = lk.lkb_origin.conv, lk.small_conv.conv
        conv, bn = self.pw1.conv, self.pw1.bn
        if not (DW_BN_FUSE and isinstance(conv, PointwiseConv) and isinstance(big, LargeKernelDW) and bn.training
                and not (bn.sync and bnm._collectives_on()) and bn.running_mean.dtype == torch.float32
                and not big.weight.requires_grad and not small.weight.requires_grad and small.kernel_size[0] == 5
                and out.dtype == torch.bfloat16 and out.is_cuda):
            return None
        if not ops.dwconv_lk_bn_supported((out.shape[0], conv.out_channels, out.shape[2], out.shape[3]), big.kernel_size[0], 5):
            return None
        z1, sums = conv.forward_sums(out, always=True)
        if sums is None:                                       # this 1x1 conv is not served by the MFMA GEMM: unfused
            return lk.forward_act(fused_bn_act(z1, bn, act=ops.ACT_RELU), ops.ACT_RELU)
        y_big, y_small, st = ops.dwconv_lk_bn(z1, sums, bn, big.weight, small.weight)
        cnt = float(z1.numel() // z1.shape[1])
        if bnm._ACTIVE_DEFERRED is None:
            bn.num_batches_tracked += 1
        else:
            bnm._ACTIVE_DEFERRED.count(bn)
            if bn.replay_update and torch.is_grad_enabled():
                bnm._ACTIVE_DEFERRED.add(bn, st[0], st[1], cnt)
        return fused_bn_act(y_big, lk.lkb_origin.bn, y_small, lk.small_conv.bn, act=ops.ACT_RELU)

    def forward(self, x, pre_out=None, next_bn=None):
        """pre_out / next_bn: see ConvFFN.forward."""
        if FUSE_BN and self.training and x.is_cuda and hasattr(self.large_kernel, "small_conv"):
            if pre_out is not None:
                out = pre_out
            else:
                out, x = fused_bn_act(x, self.prelkb_bn, skip=True)    # x: the same tensor, routed for the residual use
            adpt, join = None, None
            if self.test_id >= 0:
                if ADAPTER_STREAMS and not self.inline_adapter and torch.cuda.current_stream().cuda_stream not in NO_FORK_ON:
                    adpt, join = _forked_adapter(self.adapter, second_use(out) if BN_DUP_FORKED else out)
                else:
                    adpt = self.adapter(second_use(out))         # (see ConvFFN.forward)
            t = self._pw1_large_kernel_fused(out)
            if t is None:
                t = self.large_kernel.forward_act(self.pw1(out), ops.ACT_RELU)
            z, s2 = _conv_sums(self.pw2.conv, t)
            if join is not None:
                join()
            mask = _drop_mask(self.drop_path, x)
            if next_bn is not None:
                pair = fused_bn_act_next(z, self.pw2.bn, next_bn, mask=mask, r1=x, r2=adpt, r2_scale=self.gamma, sums=s2)
                if pair is not None:
                    return pair
            y = fused_bn_act(z, self.pw2.bn, mask=mask, r1=x, r2=adpt, r2_scale=self.gamma, sums=s2)
            return y if next_bn is None else (y, None)
        assert pre_out is None and next_bn is None
        out = self.prelkb_bn(x)
        adpt = self.adapter(out) if self.test_id >= 0 else None
        out = self.pw2(self.lk_nonlinear(self.large_kernel(self.pw1(out))))
        y = x + self.drop_path(out)
        return y if adpt is None else y + self.gamma * adpt


class RepLKNetStage(nn.Module):
    """rka.py:329-369."""

    def __init__(self, channels, num_blocks, stage_lk_size, drop_path, small_kernel, g_blk=1.0, g_ffn=1.0,
                 dw_ratio=1, ffn_ratio=4, use_checkpoint=False, small_kernel_merged=False,
                 norm_intermediate_features=False, adpt_test=0, ratio=0.25):
        super().__init__()
        self.use_checkpoint = use_checkpoint
        blks = []
        for i in range(num_blocks):
            a_r, a_c = adpt_test, adpt_test
            if adpt_test == 5:
                a_r, a_c = -1, 1
            if adpt_test == 6:
                a_r, a_c = 4, -1
            dp = drop_path[i] if isinstance(drop_path, list) else drop_path
            blks.append(RepLKBlock(channels, int(channels * dw_ratio), stage_lk_size, small_kernel, dp, g_blk,
                                   small_kernel_merged, a_r, ratio))
            blks.append(ConvFFN(channels, int(channels * ffn_ratio), channels, dp, g_ffn, a_c, ratio))
        self.blocks = nn.ModuleList(blks)
        self.norm = get_bn(channels) if norm_intermediate_features else nn.Identity()
        if use_checkpoint:
            _mark_replay(self.blocks)

    def forward(self, x):
        if not (FUSE_BN and self.training and x.is_cuda):
            for blk in self.blocks:
                x = blk(x)
            return x
        # training on the HIP kernels: a block's last launch also computes the next block's first BatchNorm
        # (fused_bn_act_next) where the one-launch channel kernels serve the shape
        pre_out = None
        for i, blk in enumerate(self.blocks):
            nxt = self.blocks[i + 1] if i + 1 < len(self.blocks) else None
            chain = nxt is not None and not (isinstance(blk, RepLKBlock) and not hasattr(blk.large_kernel, "small_conv")) \
                and not (isinstance(nxt, RepLKBlock) and not hasattr(nxt.large_kernel, "small_conv"))
            r = blk(x, pre_out, nxt.pre_bn if chain else None)
            x, pre_out = r if isinstance(r, tuple) else (r, None)
        return x


def _mark_replay(module):
    """BNs inside a segment the reference wraps in checkpoint.checkpoint (rka.py:364-368, 515-519)."""
    for m in module.modules():
        if isinstance(m, BatchNorm2d):
            m.replay_update = True


class RepLKNetAdapter(nn.Module):
    """rka.py:381-628."""

    def __init__(self, large_kernel_sizes, layers, channels, drop_path_rate, small_kernel, dw_ratio=1,
                 ffn_ratio=4, in_channels=3, num_classes=1000, out_indices=None, use_checkpoint=False,
                 small_kernel_merged=False, use_sync_bn=True, norm_intermediate_features=False,
                 pretrained=None, g_blk=1, g_ffn=1, trans_adpt=False, input_adpt=False, adpt_test=0,
                 ratio=0.25, num_input_images=1):
        super().__init__()
        if num_classes is not None and out_indices is not None:
            raise ValueError("cannot specify both num_classes (for pretraining) and out_indices")
        self.out_indices = out_indices
        if use_sync_bn:
            enable_sync_bn()
        self.channels = channels
        self.use_checkpoint = use_checkpoint
        self.num_stages = len(layers)
        self.num_input_images = num_input_images
        bw = channels[0]
        self.stem = nn.ModuleList([
            conv_bn_relu(in_channels * num_input_images, bw, 3, 2, 1, 1),
            conv_bn_relu(bw, bw, 3, 1, 1, bw),
            conv_bn_relu(bw, bw, 1, 1, 0, 1),
            conv_bn_relu(bw, bw, 3, 2, 1, bw)])
        if use_checkpoint:
            _mark_replay(self.stem[1:])
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(layers))]
        self.stages = nn.ModuleList()
        self.transitions = nn.ModuleList()
        # rka.py:429-436: `--input` adapter on the stem, `--trans` adapters (+ DropPath) after every transition
        self.input_adpt = bool(input_adpt)
        if input_adpt:
            self.input_adapter = InputAdapter(bw)
        self.trans_adpt = False
        if trans_adpt:
            self.trans_adpt = nn.ModuleList()
            self.trans_drop_path = nn.ModuleList()
        for s in range(self.num_stages):
            self.stages.append(RepLKNetStage(
                channels[s], layers[s], large_kernel_sizes[s], dpr[sum(layers[:s]):sum(layers[:s + 1])],
                small_kernel, g_blk, g_ffn, dw_ratio, ffn_ratio, use_checkpoint, small_kernel_merged,
                norm_intermediate_features, adpt_test, ratio))
            if s < len(layers) - 1:
                self.transitions.append(nn.Sequential(
                    conv_bn_relu(channels[s], channels[s + 1], 1, 1, 0, 1),
                    conv_bn_relu(channels[s + 1], channels[s + 1], 3, 2, 1, channels[s + 1])))
                if trans_adpt:
                    self.trans_adpt.append(Adapter(channels[s + 1], adpt_test))
                    self.trans_drop_path.append(DropPath(dpr[sum(layers[:s])]))
        if num_classes is not None:
            self.norm = get_bn(channels[-1])
            self.avgpool = nn.AdaptiveAvgPool2d(1)
            self.head = nn.Linear(channels[-1], num_classes)
        if pretrained:
            self.load_pretrained(pretrained)
        # adapter output projections start at zero (rka.py:482-493)
        for n, m in self.named_modules():
            if "adapter" in n and n.endswith("D_fc2") and isinstance(m, (nn.Linear, nn.Conv2d)):
                nn.init.constant_(m.weight, 0)
                nn.init.constant_(m.bias, 0)

    def load_pretrained(self, path):
        if not os.path.exists(path):
            print(f"[ppeadepth] pretrained backbone {path} not found: keeping random initialisation", file=sys.stderr)
            return
        weights = torch.load(path, map_location="cpu")
        for key in ("model", "state_dict"):
            if isinstance(weights, dict) and key in weights:
                weights = weights[key]
        if self.num_input_images == 2 and "stem.0.conv.weight" in weights:
            weights["stem.0.conv.weight"] = torch.cat([weights["stem.0.conv.weight"]] * 2, 1) / 2
        self.load_state_dict(weights, strict=False)

    def stem_forward(self, x):
        """rka.py:511-521: the stem, with the `--input` adapter branching off stem[0]'s output."""
        x = self.stem[0](x)
        adpt = self.input_adapter(x) if self.input_adpt else None
        for layer in self.stem[1:]:
            x = layer(x)
        return x if adpt is None else x + adpt

    def transition_forward(self, s, x):
        """rka.py:537-540: transition s followed, with `--trans`, by x + DropPath(Adapter(x))."""
        x = self.transitions[s](x)
        if self.trans_adpt:
            x = x + self.trans_drop_path[s](self.trans_adpt[s](x))
        return x

    def plan_drop_masks(self, x, stages=None):
        """Device-RNG training passes: draw the DropPath scales of the given stages (default all) for one pass now."""
        if FUSE_BN and self.training and x.is_cuda and rng.get_mode() == "device":
            _plan_drop_masks(self, x.shape[0], x.device, stages)

    def forward_features(self, x):
        self.plan_drop_masks(x)
        x = self.stem_forward(x)
        outs = []
        for s in range(self.num_stages):
            x = self.stages[s](x)
            if self.out_indices is not None and s in self.out_indices:
                outs.append(self.stages[s].norm(x))
            if s < self.num_stages - 1:
                x = self.transition_forward(s, x)
        return x if self.out_indices is None else outs

    def forward(self, x):
        return self.forward_features(x)

    def structural_reparam(self):
        """rka.py:544-547 / replknet.py:400-413: every k x k + 5 x 5 branch pair (with its two BNs) becomes ONE k x k
        depthwise conv with bias -- inference only (uses the running statistics)."""
        for m in self.modules():
            if hasattr(m, "merge_kernel"):
                m.merge_kernel()

    def deep_fuse_BN(self):
        """rka.py:563-580: fold the BN of every remaining conv-BN(-ReLU) `nn.Sequential` into its conv (inference)."""
        for m in self.modules():
            if not isinstance(m, nn.Sequential) or len(m) not in (2, 3):
                continue
            if hasattr(m[0], "kernel_size") and hasattr(m[0], "weight") and isinstance(m[1], (BatchNorm2d, nn.BatchNorm2d)):
                conv, bn = m[0], m[1]
                fused_kernel, fused_bias = fuse_bn(conv, bn)
                fused_conv = get_conv2d(conv.in_channels, conv.out_channels, conv.kernel_size[0], conv.stride[0],
                                        conv.padding[0], conv.dilation[0], conv.groups, True)
                fused_conv.weight.data = fused_kernel.detach()
                fused_conv.bias.data = fused_bias.detach()
                m[0] = fused_conv
                m[1] = nn.Identity()


def _factory(channels):
    def create(drop_path_rate=0.3, num_classes=1000, num_input_images=1, out_indices=(0, 1, 2, 3),
               use_checkpoint=True, small_kernel_merged=False, pretrained=None, use_sync_bn=True, g_blk=1.0,
               g_ffn=1.0, ratio=0.25, trans_adpt=False, input_adpt=False, adpt_test=0):
        return RepLKNetAdapter([31, 29, 27, 13], [2, 2, 18, 2], channels, drop_path_rate, 5,
                               num_classes=num_classes, out_indices=out_indices,
                               use_checkpoint=use_checkpoint, small_kernel_merged=small_kernel_merged,
                               use_sync_bn=use_sync_bn, pretrained=pretrained, g_blk=g_blk, g_ffn=g_ffn,
                               trans_adpt=trans_adpt, input_adpt=input_adpt, adpt_test=adpt_test, ratio=ratio,
                               num_input_images=num_input_images)
    return create


create_RepLKNet31B_Adapter = _factory([128, 256, 512, 1024])     # rka.py:630-633
create_RepLKNet31L_Adapter = _factory([192, 384, 768, 1536])     # rka.py:635-638
