#!/usr/bin/env python3
"""pwconv_kernel per trunk shape under a forced tile configuration (PPEA_PW_TILE=128|64|64d|32|32d; unset = dispatch)."""
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
out = []
for (B, Ci, H, W, Co) in shapes:
    x = torch.randn(B, Ci, H, W, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(Co, Ci, device=dev) / Ci ** 0.5).bfloat16()
    t = timeit(lambda: ops.pwconv_raw(w, x))
    out.append(f"{t:6.1f}")
print(os.environ.get("PPEA_PW_TILE", "auto").ljust(5), " ".join(out), flush=True)
