#!/usr/bin/env python3
"""pwconv per trunk shape: our kernel (dispatch) vs one library GEMM of the same size on a [K][B*HW] matrix
(the library's best case: no per-image batching, no layout change) -- gives the target for the kernel."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ppea-depth_amd"))
from ppeadepth import ops
dev = torch.device("cuda:0")

def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

shapes = [(12, 128, 48, 160, 128), (12, 128, 48, 160, 512), (12, 512, 48, 160, 128), (12, 256, 24, 80, 256),
          (12, 256, 24, 80, 1024), (12, 1024, 24, 80, 256), (12, 512, 12, 40, 512), (12, 512, 12, 40, 2048),
          (12, 2048, 12, 40, 512), (12, 1024, 6, 20, 1024), (12, 1024, 6, 20, 4096), (12, 4096, 6, 20, 1024)]
print("shape                         ours us  TF/s  GB/s | lib us  TF/s | bmm us")
for (B, Ci, H, W, Co) in shapes:
    x = torch.randn(B, Ci, H, W, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(Co, Ci, device=dev) / Ci ** 0.5).bfloat16()
    x2 = torch.randn(Ci, B * H * W, device=dev, dtype=torch.bfloat16)
    x3 = x.view(B, Ci, H * W)
    t = timeit(lambda: ops.pwconv_raw(w, x))
    tl = timeit(lambda: torch.matmul(w, x2))
    tb = timeit(lambda: torch.matmul(w, x3))
    fl = 2.0 * B * H * W * Ci * Co
    by = 2.0 * (B * H * W * (Ci + Co) + Ci * Co)
    print(f"{B}x{Ci:4d}x{H:2d}x{W:3d} -> {Co:4d}   {t:7.1f} {fl/t/1e6:6.0f} {by/t/1e3:6.0f} | {tl:6.1f} {fl/tl/1e6:6.0f} | {tb:6.1f}", flush=True)
