"""Launches for the PMC passes (rocprofv3 --pmc ... -- python3 tools/pmc_target.py): the 31x31 (+5x5) bf16 depthwise
conv forward / dgrad at stage 0 and the stage-2 pointwise GEMM shapes.  GPU box only."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ppea-depth_amd"))
from ppeadepth import ops  # noqa: E402

dev = torch.device("cuda:0")
B = 12
x = torch.randn(B, 128, 48, 160, device=dev).bfloat16().requires_grad_(True)
wb = torch.randn(128, 1, 31, 31, device=dev) / 31
ws = torch.randn(128, 1, 5, 5, device=dev) / 5
for _ in range(4):
    yb, ys = ops.dwconv_lk(x, wb, ws)
    torch.autograd.grad((yb, ys), x, (torch.ones_like(yb), torch.ones_like(ys)))
# stage 2: the batch-major variant (dwconv_bm_kernel<27,5,...>) forward and data gradient
x2 = torch.randn(B, 512, 12, 40, device=dev).bfloat16().requires_grad_(True)
wb2 = torch.randn(512, 1, 27, 27, device=dev) / 27
ws2 = torch.randn(512, 1, 5, 5, device=dev) / 5
for _ in range(4):
    yb, ys = ops.dwconv_lk(x2, wb2, ws2)
    torch.autograd.grad((yb, ys), x2, (torch.ones_like(yb), torch.ones_like(ys)))
for (K, H, W, M) in [(512, 12, 40, 512), (2048, 12, 40, 512), (512, 12, 40, 2048), (128, 48, 160, 128)]:
    xx = torch.randn(B, K, H, W, device=dev).bfloat16()
    a = (torch.randn(M, K, device=dev) / K ** 0.5).bfloat16()
    for _ in range(3):
        ops.pwconv_raw(a, xx)
# dense convs of the decoder on the implicit-GEMM kernels: forward, data gradient, weight gradient
for (Cin, H, W, Cout) in [(1024, 12, 40, 512), (256, 48, 160, 128), (32, 192, 640, 32)]:
    xc = torch.randn(B, Cin, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wc = (torch.randn(Cout, Cin, 3, 3, device=dev) / (9 * Cin) ** 0.5).bfloat16().requires_grad_(True)
    bc = torch.zeros(Cout, device=dev).bfloat16().requires_grad_(True)
    for _ in range(3):
        y = ops.conv2d_nhwc(xc, wc, bc, 1, 1, True, "elu")
        torch.autograd.grad(y, (xc, wc, bc), torch.ones_like(y))
torch.cuda.synchronize()
print("done")
