#!/usr/bin/env python3
"""pwconv per trunk shape (20 launches inside one hipGraph each): v1, every v2 tile (LDS-DMA ring), the dispatch's own
choice and, with --lib, one library GEMM of the same size on a [K][B*HW] matrix (the library's best case: no per-image
batching, no layout change).      python tools/bench_pw4.py [--lib] [--quick]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ppea-depth_amd"))
from ppeadepth import ops
dev = torch.device("cuda:0")
REP = 20


def graph_time(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(REP): fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * REP) * 1e3


shapes = [(12, 128, 48, 160, 128), (12, 128, 48, 160, 512), (12, 512, 48, 160, 128), (12, 256, 24, 80, 256),
          (12, 256, 24, 80, 1024), (12, 1024, 24, 80, 256), (12, 512, 12, 40, 512), (12, 512, 12, 40, 2048),
          (12, 2048, 12, 40, 512), (12, 1024, 6, 20, 1024), (12, 1024, 6, 20, 4096), (12, 4096, 6, 20, 1024),
          (12, 512, 12, 40, 1152), (12, 512, 12, 40, 128), (12, 128, 12, 40, 512)]
lib = "--lib" in sys.argv
tiles = ["0", "128,64", "128,32", "64,64", "64,32", "32,64"]
print("shape                      " + "".join(f"{('v1' if t == '0' else t):>9s}" for t in tiles) + "     auto  TF/s  GB/s"
      + ("   lib us  TF/s" if lib else ""))
tot = {t: 0.0 for t in tiles + ["auto", "lib"]}
for (B, Ci, H, W, Co) in shapes:
    x = torch.randn(B, Ci, H, W, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(Co, Ci, device=dev) / Ci ** 0.5).bfloat16()
    line = f"{B}x{Ci:4d}x{H:2d}x{W:3d} -> {Co:4d}  "
    for t in tiles:
        if t != "0" and Ci % int(t.split(",")[1]):
            line += "        -"
            continue
        os.environ["PPEA_PW_V2"] = t
        tt = graph_time(lambda: ops.pwconv_raw(w, x))
        tot[t] += tt
        line += f"{tt:9.1f}"
    os.environ.pop("PPEA_PW_V2")
    t = graph_time(lambda: ops.pwconv_raw(w, x))
    tot["auto"] += t
    fl = 2.0 * B * H * W * Ci * Co
    by = 2.0 * (B * H * W * (Ci + Co) + Ci * Co)
    line += f"{t:9.1f} {fl/t/1e6:5.0f} {by/t/1e3:5.0f}"
    if lib:
        x2 = torch.randn(Ci, B * H * W, device=dev, dtype=torch.bfloat16)
        tl = graph_time(lambda: torch.matmul(w, x2))
        tot["lib"] += tl
        line += f" {tl:8.1f} {fl/tl/1e6:5.0f}"
    print(line, flush=True)
print("sum                        " + "".join(f"{tot[t]:9.1f}" for t in tiles) + f"{tot['auto']:9.1f}" + (f"             {tot['lib']:8.1f}" if lib else ""))
