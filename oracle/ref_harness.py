"""TEST INFRASTRUCTURE ONLY -- never imported by the product package.

Loads the *reference* PPEA-Depth implementation from /root/reference inside THIS
container (CPU only) so that golden vectors can be generated from it
(SURVEY.md section 8(c) / Appendix A).  The reference itself never travels: only the
`.npz` fixtures written by `oracle/gen_golden.py` do.  This file does not exist
functionally on the GPU box (there is no /root/reference there) and nothing
under tests/ -m gpu, smoke() or bench.py imports it.

Third-party modules the reference needs but this image lacks are replaced by
minimal stand-ins *of the third-party libraries* (not of the reference):

  timm.layers.DropPath / trunc_normal_   (replknet_adapter.py:15)
  torchvision.models.ResNet + BasicBlock (resnet_encoder.py:17, 25-61)
  torchvision.transforms, cv2, wandb, skimage.transform (import-time only)
  torchmetrics.Metric                    (trainer.py:41-46)

ResNet-18 / DropPath stand-ins restate the published torchvision / timm
definitions; for those two the pin is "self-consistent only" (SURVEY 8(c)).
"""
import contextlib
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "ppeadepth"))


# --------------------------------------------------------------------------
# third-party stand-ins
# --------------------------------------------------------------------------
class _DropPath(nn.Module):
    """timm.layers.DropPath (stochastic depth per sample, scale_by_keep=True)."""

    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        mask = x.new_empty(shape).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask


def _conv3x3(i, o, stride=1):
    return nn.Conv2d(i, o, 3, stride, 1, bias=False)


class _BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        idt = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            idt = self.downsample(x)
        return self.relu(out + idt)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("only ResNet-18 is on the hot path")


class _ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)


class _Metric(nn.Module):
    """torchmetrics.Metric: only add_state is used (trainer.py:45-46)."""

    def __init__(self):
        super().__init__()

    def add_state(self, name, default, dist_reduce_fx=None):
        setattr(self, name, default)


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_installed = False


def install_stubs():
    global _installed
    if _installed:
        return
    sys.dont_write_bytecode = True  # the reference tree is read-only
    import accelerate  # noqa: F401  (must precede the spec-less stubs)
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot  # noqa: F401

    def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return nn.init.trunc_normal_(t, mean, std, a, b)

    layers = _module("timm.layers", DropPath=_DropPath, trunc_normal_=trunc_normal_)
    _module("timm", layers=layers)

    def _resnet18(pretrained=False, **kw):
        return _ResNet(_BasicBlock, [2, 2, 2, 2])

    def _unsupported(*a, **k):
        raise NotImplementedError

    resnet = _module("torchvision.models.resnet", BasicBlock=_BasicBlock,
                     Bottleneck=_Bottleneck, ResNet=_ResNet)
    models = _module("torchvision.models", ResNet=_ResNet, resnet=resnet,
                     resnet18=_resnet18, resnet34=_unsupported, resnet50=_unsupported,
                     resnet101=_unsupported, resnet152=_unsupported)

    class _T:
        def __init__(self, *a, **k):
            pass

        @staticmethod
        def get_params(*a, **k):
            return None

    transforms = _module("torchvision.transforms", ColorJitter=_T, ToTensor=_T,
                         Resize=_T, InterpolationMode=types.SimpleNamespace(LANCZOS=1))
    _module("torchvision", models=models, transforms=transforms)
    def _cv2_resize(img, dsize, interpolation=None):
        """cv2.resize with the default INTER_LINEAR on a 2-D float array (trainer.py:793): bilinear, half-pixel
        centres, no antialiasing."""
        import numpy as np
        import torch.nn.functional as F
        t = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32))[None, None]
        return F.interpolate(t, (dsize[1], dsize[0]), mode="bilinear", align_corners=False)[0, 0].numpy()

    _module("cv2", setNumThreads=lambda n: None, resize=_cv2_resize)
    _module("wandb", init=lambda *a, **k: None, log=lambda *a, **k: None)
    _module("torchmetrics", Metric=_Metric)
    sk_t = _module("skimage.transform")
    _module("skimage", transform=sk_t)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


@contextlib.contextmanager
def scratch_cwd(path="/tmp/ppea_ref_scratch"):
    """The reference hard-codes ./pretrained/*.pth (replk_matching_adapter.py:46,
    repdepth.py:84).  An empty dict -> load_state_dict(strict=False) no-op."""
    os.makedirs(os.path.join(path, "pretrained"), exist_ok=True)
    for n in ("RepLKNet-31B_ImageNet-1K_224.pth", "RepLKNet-31L_ImageNet-22K.pth"):
        p = os.path.join(path, "pretrained", n)
        if not os.path.exists(p):
            torch.save({}, p)
    old = os.getcwd()
    os.chdir(path)
    try:
        yield
    finally:
        os.chdir(old)


def parse_options(extra_argv=()):
    install_stubs()
    from ppeadepth.options import MonodepthOptions
    old = sys.argv
    sys.argv = ["x", "--adapter", "--weights_init", "scratch"] + list(extra_argv)
    try:
        return MonodepthOptions().parse()
    finally:
        sys.argv = old


class FakeAccelerator:
    """The three Accelerate methods process_batch touches (trainer.py:436,453,465)."""

    def __init__(self, device="cpu"):
        self.device = torch.device(device)
        self.is_main_process = True

    def autocast(self):
        return contextlib.nullcontext()

    def wait_for_everyone(self):
        return None


def build_reference_trainer(opt, model):
    """Trainer.__new__ + the attributes process_batch reads."""
    install_stubs()
    from ppeadepth.trainer import Trainer, DepthBins
    from ppeadepth.layers import SSIM, BackprojectDepth, Project3D
    tr = Trainer.__new__(Trainer)
    tr.opt = opt
    tr.acc = FakeAccelerator()
    tr.device = tr.acc.device
    tr.model = model
    tr.depth_bin_tracker = DepthBins(opt.min_depth)
    tr.freeze_tp = False
    tr.step = 0
    tr.is_main = True
    tr.ssim = SSIM()
    tr.backproject_depth = {0: BackprojectDepth(opt.batch_size, opt.height, opt.width)}
    tr.project_3d = {0: Project3D(opt.batch_size, opt.height, opt.width)}
    return tr
