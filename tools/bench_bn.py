#!/usr/bin/env python3
"""One-launch channel BatchNorm kernels at the stage-2 / stage-3 shapes (20 launches per hipGraph): forward (+GELU),
backward.      python tools/bench_bn.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ppea-depth_amd"))
from ppeadepth import ops
from ppeadepth.batchnorm import BatchNorm2d
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


for (N, C, H, W, act) in [(12, 2048, 12, 40, ops.ACT_GELU), (12, 512, 12, 40, ops.ACT_NONE), (12, 4096, 6, 20, ops.ACT_GELU)]:
    z = torch.randn(N, C, H, W, device=dev).bfloat16()
    dy = torch.randn(N, C, H, W, device=dev).bfloat16()
    bn = BatchNorm2d(C).to(dev)
    y, st = ops.bn_act_channel(z, bn, act=act)
    tf = graph_time(lambda: ops.bn_act_channel(z, bn, act=act))
    sums = torch.empty(3, C, device=dev, dtype=torch.float32)
    dz = torch.empty_like(z)
    g1, b1 = bn.weight.detach().float(), bn.bias.detach().float()
    stats = ops._stats_array((st[0], st[1], g1, b1, None, None, None, None))
    tb = graph_time(lambda: ops.call("ppea_bn_bwd_channel_bf16", ops.ptr(dy), ops.ptr(z), None, stats, None, 1.0 / (N * H * W),
                                     None, ops.ptr(dz), None, ops.ptr(sums), int(act), N, C, H * W, ops.stream_ptr()))
    by = z.numel() * 2
    print(f"[{N},{C},{H},{W}] act {act}: fwd {tf:6.1f} us {2 * by / tf / 1e6:5.2f} TB/s | bwd {tb:6.1f} us {3 * by / tb / 1e6:5.2f} TB/s", flush=True)
