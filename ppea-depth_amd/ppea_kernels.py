"""Collision-free entry to this build's kernels from inside a process whose `ppeadepth` is the REFERENCE's package.

    sys.path.append("<repo>/ppea-depth_amd"); import ppea_kernels
    ppea_kernels.ops.ssim_l1(pred, target, 0.85)          # autograd ops over the C ABI (ppeadepth/ops.py)
    ppea_kernels.layers.SSIM()                             # reference-named layer classes (ppeadepth/layers.py)

The kernel package next to this file is loaded under the private name `_ppea_depth_amd_kernels` (see
depthwise_conv2d_implicit_gemm.py for why); `ops` and `layers` only depend on torch and the C-ABI binding.
"""
import importlib
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.append(_HERE)

import depthwise_conv2d_implicit_gemm as _plugin  # noqa: E402

ops = _plugin.ops
_abi = importlib.import_module(ops.__package__ + "._abi")
layers = importlib.import_module(ops.__package__ + ".layers")
DepthWiseConv2dImplicitGEMM = _plugin.DepthWiseConv2dImplicitGEMM
