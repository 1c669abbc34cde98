#!/usr/bin/env python3
"""Transposed-matrix 1x1 GEMMs of the adapters' data gradients (20 launches per hipGraph): v1's transposed mode
(PPEA_PW_TA_V2=0) against the v2 ring.      python tools/bench_pw_ta.py"""
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


for (B, M, K, H, W, epi) in [(12, 128, 512, 12, 40, "dgelu"), (12, 512, 128, 12, 40, "none"), (12, 32, 128, 48, 160, "dgelu"),
                             (12, 128, 32, 48, 160, "none"), (12, 64, 256, 24, 80, "dgelu"), (12, 256, 64, 24, 80, "none"),
                             (12, 256, 1024, 6, 20, "dgelu"), (12, 1024, 256, 6, 20, "none"), (12, 512, 512, 12, 40, "none"),
                             (12, 2048, 512, 12, 40, "none"), (12, 512, 2048, 12, 40, "none")]:
    x = torch.randn(B, K, H, W, device=dev).bfloat16()
    at = (torch.randn(K, M, device=dev) / K ** 0.5).bfloat16()
    aux = torch.randn(B, M, H, W, device=dev).bfloat16()
    kw = dict(epi=ops.EPI_DGELU, aux=aux) if epi == "dgelu" else {}
    t = graph_time(lambda: ops.pwconv_ex(at, x, transposed=True, **kw))
    print(f"At[{K}][{M}] x [{B},{K},{H},{W}] {epi:5s}: {t:6.1f} us  {2.0 * B * H * W * M * K / t / 1e6:6.0f} TF/s", flush=True)
