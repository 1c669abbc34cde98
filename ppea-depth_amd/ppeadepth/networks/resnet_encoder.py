"""ResNet-18 pose encoder (reference: networks/resnet_encoder.py:25-72, 367-409).

torchvision is not a dependency: the ResNet-18 definition (BasicBlock x [2,2,2,2], torchvision
attribute names so `pose_encoder.encoder.*` state_dict keys match) is stated here.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import ops


CHANNELS_LAST = True
FUSED_NHWC_BN = True      # pose-trunk BN + ReLU + residual on the channels_last HIP kernels (GroupBN.fused)


class GroupBN(nn.BatchNorm2d):
    """nn.BatchNorm2d (same parameters / buffers / state_dict keys) that can treat its batch as `groups`
    consecutive sub-batches, each normalised with ITS OWN batch statistics and each updating the running
    statistics in turn -- exactly what separate forward calls on the sub-batches do.  This lets the pose network
    see both frame pairs of a step as one batch of 2B (convolutions launched once, not twice) without changing
    a number.  With `record` set the sub-batch statistics are kept so that a further call on the same data (the
    reference's no_grad pose pass for the matching frames) can be replayed as a running-statistics update."""
    groups = 1
    record = None           # None, or a list that receives (mean, invstd, count) per sub-batch

    def fused(self, x, act=0, res=None):
        """act(BN(x) + res) on the channels_last HIP kernels (per-sub-batch statistics, ReLU and the residual add in
        the same pass); falls back to the composed ops when the tensor is not served."""
        from .. import ops
        if (FUSED_NHWC_BN and self.training and ops.nhwc_bn_supported(x, self.groups)
                and (res is None or (res.shape == x.shape and res.dtype == x.dtype
                                     and res.is_contiguous(memory_format=torch.channels_last)))):
            y, stats = ops.nhwc_bn_act(x, self.weight, self.bias, self.running_mean, self.running_var, res, act,
                                       self.groups, self.eps, self.momentum)
            _count_batches(self, self.groups)
            if self.record is not None:
                n = (x.shape[0] // self.groups) * x.shape[2] * x.shape[3]
                for g in range(self.groups):
                    self.record.append((stats[g, 0], stats[g, 1], n))
            return y
        y = self.forward(x)
        if res is not None:
            y = y + res
        return torch.relu(y) if act == 1 else y

    def forward(self, x):
        if not self.training or (self.groups == 1 and self.record is None):
            return super().forward(x)
        outs = []
        for chunk in x.chunk(self.groups, 0):
            y, mean, invstd = torch.native_batch_norm(chunk, self.weight, self.bias, self.running_mean,
                                                      self.running_var, True, self.momentum, self.eps)
            self.num_batches_tracked += 1
            if self.record is not None:
                self.record.append((mean.detach(), invstd.detach(), chunk.numel() // chunk.shape[1]))
            outs.append(y)
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)

    def replay_update(self, mean, invstd, count):
        """Running-statistics update of one more training-mode forward on a batch with these statistics."""
        var = (1.0 / (invstd * invstd) - self.eps) * (count / max(count - 1, 1))
        # through .data: like the update inside the batch-norm kernel, this must not bump the version counter of the
        # buffers autograd holds for the backward of the passes above
        self.running_mean.data.lerp_(mean.to(self.running_mean.dtype), self.momentum)
        self.running_var.data.lerp_(var.to(self.running_var.dtype), self.momentum)
        self.num_batches_tracked += 1


def _count_batches(bn, n):
    """num_batches_tracked += n, through the step's deferred multi-tensor update when one is active."""
    from .. import batchnorm
    d = batchnorm._ACTIVE_DEFERRED
    if d is None:
        bn.num_batches_tracked += n
    else:
        d.count(bn, n)


def replay_updates(recorded):
    """`GroupBN.replay_update` for a list of (bn, mean, invstd, count) in a handful of multi-tensor launches (the
    pose trunk has 20 BNs: 8 tiny launches each otherwise).  Same arithmetic, same order of operations."""
    if not recorded:
        return
    bns = [r[0] for r in recorded]
    inv = [r[2] for r in recorded]
    var = torch._foreach_mul(inv, inv)
    var = torch._foreach_reciprocal(var)
    torch._foreach_sub_(var, [bn.eps for bn in bns])
    torch._foreach_mul_(var, [r[3] / max(r[3] - 1, 1) for r in recorded])
    mom = bns[0].momentum
    assert all(bn.momentum == mom for bn in bns)
    torch._foreach_lerp_([bn.running_mean.data for bn in bns],
                         [r[1].to(bn.running_mean.dtype) for r, bn in zip(recorded, bns)], mom)
    torch._foreach_lerp_([bn.running_var.data for bn in bns],
                         [v.to(bn.running_var.dtype) for v, bn in zip(var, bns)], mom)
    for bn in bns:
        _count_batches(bn, 1)


def _conv(c, x):
    """nn.Conv2d of the pose trunk: this build's implicit-GEMM kernels on the bf16 step, the library otherwise."""
    from .. import ops
    y = ops.conv_module(c, x) if x.is_cuda else None
    return c(x) if y is None else y


def _maxpool(m, x):
    """MaxPool2d(3, 2, 1) of the trunk: gather kernels on channels_last HIP tensors, the module otherwise."""
    y = ops.maxpool3x3s2(x) if (x.is_cuda and m.kernel_size == 3 and m.stride == 2 and m.padding == 1) else None
    return m(x) if y is None else y


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = ops.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = GroupBN(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = ops.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = GroupBN(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        if FUSED_NHWC_BN and self.training and x.is_cuda:
            identity = x if self.downsample is None else self.downsample[1].fused(_conv(self.downsample[0], x))
            out = self.bn1.fused(_conv(self.conv1, x), 1)
            return self.bn2.fused(_conv(self.conv2, out), 1, identity)
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


class ResNetMultiImageInput(nn.Module):
    """ResNet-18 trunk whose first conv takes num_input_images * 3 channels."""

    def __init__(self, layers=(2, 2, 2, 2), num_classes=1000, num_input_images=1):
        super().__init__()
        self.inplanes = 64
        self.conv1 = ops.Conv2d(num_input_images * 3, 64, 7, 2, 3, bias=False)
        self.bn1 = GroupBN(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride=1):
        down = None
        if stride != 1 or self.inplanes != planes:
            down = nn.Sequential(ops.Conv2d(self.inplanes, planes, 1, stride, bias=False), GroupBN(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, down)]
        self.inplanes = planes
        layers += [BasicBlock(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


class ResnetEncoder(nn.Module):
    def __init__(self, num_layers, pretrained, num_input_images=1, **kwargs):
        super().__init__()
        if num_layers != 18:
            raise NotImplementedError("the pose network of the hot path is ResNet-18")
        if pretrained:
            raise RuntimeError("ImageNet ResNet-18 weights need a network fetch (resnet_encoder.py:63-71); "
                               "load them through load_state_dict, or pass --weights_init scratch")
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        self.encoder = ResNetMultiImageInput(num_input_images=num_input_images)
        for name, p in self.encoder.named_parameters():
            if "fc" in name:
                p.requires_grad = False

    def forward(self, input_image, groups=1, record=False):
        """groups > 1: the batch is `groups` consecutive sub-batches with separate BatchNorm statistics (see GroupBN);
        record: keep the first sub-batch's statistics in `self.recorded` [(bn, mean, invstd, count)]."""
        e = self.encoder
        bns = [m for m in e.modules() if isinstance(m, GroupBN)]
        for m in bns:
            m.groups = groups
            m.record = [] if record else None
        try:
            feats = self._forward(input_image)
        finally:
            if record:
                self.recorded = [(m,) + m.record[0] for m in bns if m.record]
            for m in bns:
                m.groups, m.record = 1, None
        return feats

    # -- running-statistics bookkeeping for a forward pass that is replayed instead of recomputed -----------------
    def running_buffers(self):
        bns = [m for m in self.encoder.modules() if isinstance(m, nn.BatchNorm2d)]
        return bns, [m.running_mean for m in bns] + [m.running_var for m in bns]

    def snapshot_running(self):
        """Copies of all running means / variances (multi-tensor copy), taken BEFORE a training-mode pass."""
        _, bufs = self.running_buffers()
        with torch.no_grad():
            return [b.clone() for b in bufs]

    def batch_stats_since(self, snap):
        """Batch statistics (mean | unbiased var per layer) of the ONE training-mode pass run since `snap`, recovered
        from the momentum update r1 = (1 - m) r0 + m s  ->  s = (r1 - (1 - m) r0) / m."""
        bns, bufs = self.running_buffers()
        m = bns[0].momentum
        with torch.no_grad():
            scaled = torch._foreach_mul(snap, 1.0 - m)
            diff = torch._foreach_sub([b.detach() for b in bufs], scaled)
            return torch._foreach_div(diff, m)

    def replay_pass(self, stats):
        """Running-statistics effect of one more training-mode pass over a batch with these statistics."""
        bns, bufs = self.running_buffers()
        with torch.no_grad():
            torch._foreach_lerp_([b.data for b in bufs], stats, bns[0].momentum)   # .data: no version bump
            for bn in bns:
                bn.num_batches_tracked += 1

    def _forward(self, input_image):
        e = self.encoder
        from .. import ops
        if (ops.CONV_MFMA and input_image.is_cuda and input_image.dtype == torch.float32 and ops.bf16_autocast()
                and FUSED_NHWC_BN and self.training):
            # normalisation, bf16 rounding (what autocast feeds conv1), 6 -> 8 channels and channels_last in one kernel
            x = ops.image_to_nhwc(input_image, 8, 0.45, 0.225)
            self.features = [e.bn1.fused(_conv(e.conv1, x), 1)]
            self.features.append(e.layer1(_maxpool(e.maxpool, self.features[-1])))
            self.features.append(e.layer2(self.features[-1]))
            self.features.append(e.layer3(self.features[-1]))
            self.features.append(e.layer4(self.features[-1]))
            return self.features
        x = (input_image - 0.45) / 0.225
        if CHANNELS_LAST and x.is_cuda:
            # MIOpen's implicit-GEMM convolutions are NHWC-native: with a channels_last input every activation of the
            # trunk stays NHWC and no conv pays a layout round trip on its activations (the weights keep their
            # storage -- they are views into the optimizer's flat buffer -- and are re-laid-out per call, which is small)
            x = x.contiguous(memory_format=torch.channels_last)
        self.features = [e.bn1.fused(e.conv1(x), 1) if (FUSED_NHWC_BN and self.training and x.is_cuda)
                         else e.relu(e.bn1(e.conv1(x)))]
        self.features.append(e.layer1(_maxpool(e.maxpool, self.features[-1])))
        self.features.append(e.layer2(self.features[-1]))
        self.features.append(e.layer3(self.features[-1]))
        self.features.append(e.layer4(self.features[-1]))
        return self.features
