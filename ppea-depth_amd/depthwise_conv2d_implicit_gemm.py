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
"""
import os
import sys

import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ppeadepth import ops  # noqa: E402


class DepthWiseConv2dImplicitGEMM(nn.Conv2d):
    def __init__(self, channels, kernel, bias=False):
        super().__init__(channels, channels, kernel, stride=1, padding=kernel // 2, dilation=1,
                         groups=channels, bias=bias)

    def forward(self, x):
        y, _ = ops.dwconv_lk(x, self.weight, None)
        if self.bias is not None:
            y = y + self.bias.view(1, -1, 1, 1).to(y.dtype)
        return y
