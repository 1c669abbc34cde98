"""Large-kernel depthwise conv plug-in, MI355X build.

The reference's `get_conv2d` (networks/replknet_adapter.py:151-168) does, when the environment
variable LARGE_KERNEL_CONV_IMPL=<dir> is set,

    sys.path.append(<dir>); from depthwise_conv2d_implicit_gemm import DepthWiseConv2dImplicitGEMM
    return DepthWiseConv2dImplicitGEMM(in_channels, kernel_size, bias=bias)

for depthwise convs with k > 5, stride 1, padding k//2.  Point LARGE_KERNEL_CONV_IMPL at this
directory and the unmodified reference runs its 31/29/27/13 convs on the gfx950 HIP kernels of
libppea_depth.so.  The module is an nn.Conv2d subclass, so `.weight [C,1,k,k]`, `.bias`,
`in_channels/out_channels/kernel_size/stride/padding/dilation/groups` and the state_dict keys are
those the reference's `merge_kernel` / `deep_fuse_BN` (rka.py:250-261, 563-580) expect.

Name collision: inside the reference process `sys.modules["ppeadepth"]` is the REFERENCE's package
(`python -m ppeadepth.train`), so this file must not `import ppeadepth`.  The kernel package that
sits next to this file (also a directory called `ppeadepth`, because it mirrors the reference's module
API) is therefore loaded under the private name `_ppea_depth_amd_kernels` straight from its path; only
its `ops` / `_abi` modules (ctypes over the C ABI) are pulled in.  When the process's `ppeadepth` IS
this build's package, that one is used and nothing is loaded twice.
"""
import importlib
import importlib.util
import os
import sys

import torch.nn as nn

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG_DIR = os.path.join(_HERE, "ppeadepth")
_PRIVATE = "_ppea_depth_amd_kernels"


def _load_ops():
    mine = sys.modules.get("ppeadepth")
    if mine is not None and os.path.dirname(os.path.abspath(getattr(mine, "__file__", "") or "")) == _PKG_DIR:
        return importlib.import_module("ppeadepth.ops")
    pkg = sys.modules.get(_PRIVATE)
    if pkg is None:
        spec = importlib.util.spec_from_file_location(
            _PRIVATE, os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
        pkg = importlib.util.module_from_spec(spec)
        sys.modules[_PRIVATE] = pkg
        spec.loader.exec_module(pkg)
    return importlib.import_module(_PRIVATE + ".ops")       # relative imports inside resolve to the private name


ops = _load_ops()


class DepthWiseConv2dImplicitGEMM(nn.Conv2d):
    def __init__(self, channels, kernel, bias=False):
        super().__init__(channels, channels, kernel, stride=1, padding=kernel // 2, dilation=1,
                         groups=channels, bias=bias)

    def forward(self, x):
        y, _ = ops.dwconv_lk(x, self.weight, None)
        if self.bias is not None:
            y = y + self.bias.view(1, -1, 1, 1).to(y.dtype)
        return y
