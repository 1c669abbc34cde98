#!/usr/bin/env python3
"""An adapter's two weight gradients (pair launch vs two launches), 20 per hipGraph.      python tools/bench_pwgrad.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ppea-depth_amd"))
from ppeadepth import ops
dev = torch.device("cuda:0")


def graph_time(fn, REP=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(REP): fn()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * REP) * 1e3


for (B, C, Ch, H, W) in [(12, 512, 128, 12, 40), (12, 128, 32, 48, 160), (12, 256, 64, 24, 80), (12, 1024, 256, 6, 20)]:
    dy = torch.randn(B, C, H, W, device=dev).bfloat16()
    h = torch.randn(B, Ch, H, W, device=dev).bfloat16()
    g = torch.randn(B, Ch, H, W, device=dev).bfloat16()
    x = torch.randn(B, C, H, W, device=dev).bfloat16()
    a = (dy, h, (C, Ch), torch.float32, 1, (0, C), torch.float32)
    b = (g, x, (Ch, C), torch.float32, 1, (0, Ch), torch.float32)
    tp = graph_time(lambda: ops.pwgrad_into_pair(a, b))
    ts = graph_time(lambda: (ops.pwgrad_into(*a), ops.pwgrad_into(*b)))
    fl = 4.0 * B * H * W * C * Ch
    print(f"C {C} hidden {Ch} @{H}x{W}: pair {tp:6.1f} us ({fl / tp / 1e6:5.0f} TF/s)   two launches {ts:6.1f} us", flush=True)
