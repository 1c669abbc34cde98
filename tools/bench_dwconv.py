"""Micro-benchmark of the large-kernel depthwise conv at the BASELINE config-2 shapes (B=12).
Usage (GPU box): python tools/bench_dwconv.py [--dtype f32|bf16] [--iters 20]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ppea-depth_amd"))
from ppeadepth import ops  # noqa: E402

SHAPES = [(31, 128, 48, 160), (29, 256, 24, 80), (27, 512, 12, 40), (13, 1024, 6, 20)]


def timeit(fn, iters):
    """20 launches captured in ONE hipGraph (device-paced: no host launch gaps), average of 5 replays."""
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (5 * iters) * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--lib", action="store_true")
    a = ap.parse_args()
    dt = torch.float32 if a.dtype == "f32" else torch.bfloat16
    dev = torch.device("cuda:0")
    for K, C, H, W in SHAPES:
        x = torch.randn(a.batch, C, H, W, device=dev).to(dt)
        wb = torch.randn(C, 1, K, K, device=dev) / K
        ws = torch.randn(C, 1, 5, 5, device=dev) / 5
        gy = torch.randn_like(x)
        fl_big = 2.0 * a.batch * C * H * W * K * K
        fl_all = fl_big + 2.0 * a.batch * C * H * W * 25
        by = x.element_size() * a.batch * C * H * W
        t_big = timeit(lambda: ops.dwconv_lk(x, wb, None), a.iters)
        t_fused = timeit(lambda: ops.dwconv_lk(x, wb, ws), a.iters)
        if dt == torch.bfloat16:                                # the data-gradient entry itself (no autograd inside the capture)
            pbf, psf = ops._packed_filter(wb, True), ops._packed_filter(ws, True)
            dx = torch.empty_like(x)

            def bwd():
                ops.call("ppea_dwconv_lk_bwd_data_bf16p", ops.ptr(gy), ops.ptr(gy), ops.ptr(pbf), ops.ptr(psf), ops.ptr(dx),
                         a.batch, C, H, W, K, 5, ops.stream_ptr())
        else:
            xr = x.clone().requires_grad_(True)
            yb, ys = ops.dwconv_lk(xr, wb, ws)

            def bwd():
                torch.autograd.grad((yb, ys), xr, (gy, gy), retain_graph=True)
        t_bwd = timeit(bwd, a.iters)
        xf = x.float()
        t_torch = timeit(lambda: torch.nn.functional.conv2d(xf, wb, None, 1, K // 2, 1, C), 2) if a.lib else float("nan")
        print(f"k{K} [{a.batch},{C},{H},{W}] {a.dtype}: big {t_big*1e6:8.1f} us {fl_big/t_big/1e12:6.1f} TF "
              f"{(2*by + C*K*K*4)/t_big/1e9:7.1f} GB/s | fused big+5x5 {t_fused*1e6:8.1f} us {fl_all/t_fused/1e12:6.1f} TF "
              f"| dgrad(both) {t_bwd*1e6:8.1f} us {fl_all/t_bwd/1e12:6.1f} TF | torch/MIOpen big {t_torch*1e6:9.1f} us",
              flush=True)


if __name__ == "__main__":
    main()
