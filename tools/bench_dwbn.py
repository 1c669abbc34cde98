#!/usr/bin/env python3
"""RepLKBlock's pw1 -> BN + ReLU -> large-kernel pair, forward: fused (BatchNorm applied in the depthwise kernel's staging
pass) against the separate launches, per stage shape; 20 launches inside one hipGraph each."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ppea-depth_amd"))
from ppeadepth import ops
from ppeadepth.batchnorm import BatchNorm2d, fused_bn_act


def graph_time(fn, REP=20):
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


dev = torch.device("cuda:0")
for (K, C, H, W) in [(31, 128, 48, 160), (29, 256, 24, 80), (27, 512, 12, 40), (13, 1024, 6, 20)]:
    N = 12
    x = torch.randn(N, C, H, W, device=dev).bfloat16()
    wpw = (torch.randn(C, C, 1, 1, device=dev) / C ** 0.5)
    wb = torch.randn(C, 1, K, K, device=dev) / K
    ws = torch.randn(C, 1, 5, 5, device=dev) / 5
    bn = BatchNorm2d(C).to(dev)
    with torch.no_grad():
        def sep():
            z = ops.pwconv_frozen(x, wpw)
            t = fused_bn_act(z, bn, act=ops.ACT_RELU)
            return ops.dwconv_lk(t, wb, ws)

        def sep_sums():
            z, sums = ops.pwconv_frozen(x, wpw, want_sums=True)
            t = fused_bn_act(z, bn, act=ops.ACT_RELU, sums=sums)
            return ops.dwconv_lk(t, wb, ws)

        def fused():
            z, sums = ops.pwconv_frozen(x, wpw, want_sums=True)
            return ops.dwconv_lk_bn(z, sums, bn, wb, ws)

        def only_dw():
            return ops.dwconv_lk(x, wb, ws)

        def only_pw():
            return ops.pwconv_frozen(x, wpw)

        def only_pw_sums():
            return ops.pwconv_frozen(x, wpw, want_sums=True)
        ts = [graph_time(f) for f in (sep, sep_sums, fused, only_dw, only_pw, only_pw_sums)]
    print(f"k{K} [{N},{C},{H},{W}]: separate {ts[0]:6.1f}  separate+sums {ts[1]:6.1f}  fused {ts[2]:6.1f} us | dwconv alone {ts[3]:6.1f}"
          f"  pwconv {ts[4]:6.1f}  pwconv+stats {ts[5]:6.1f}", flush=True)
