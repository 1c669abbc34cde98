#!/usr/bin/env python3
"""Back-to-back launch time of the plane-sweep cost volume at the benchmarked shape ([12,128,48,160] features, 96 bins).
GPU box:  python tools/bench_costvol.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ppea-depth_amd"))

import torch  # noqa: E402


def main():
    from ppeadepth import ops, synthetic as synth
    dev = torch.device("cuda:0")
    B, C, h, w, D = 12, 128, 48, 160, 96
    g = torch.Generator().manual_seed(0)
    cur = torch.randn(B, C, h, w, generator=g).to(dev)
    look = torch.randn(B, C, h, w, generator=g).to(dev)
    K, inv_K = synth.kitti_K(192, 640, 2)
    K, inv_K = K[None].repeat(B, 1, 1).to(dev), inv_K[None].repeat(B, 1, 1).to(dev)
    T = torch.eye(4)[None].repeat(B, 1, 1)
    T[:, 2, 3] = 1.0                                  # 1 m along the optical axis
    T[:, 0, 3] = 0.05
    T = T.to(dev)
    bins = torch.exp(torch.linspace(torch.log(torch.tensor(0.1)), torch.log(torch.tensor(10.0)), D)).to(dev)
    ref = None
    for what, a, b in (("fp32 features", cur.bfloat16().float(), look.bfloat16().float()),
                       ("bf16 features, packed channel pairs", cur.bfloat16(), look.bfloat16())):
        for _ in range(3):
            cost = ops.cost_volume(a, b, T, K, inv_K, bins)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        n = 20
        for _ in range(n):
            cost = ops.cost_volume(a, b, T, K, inv_K, bins)
        e.record()
        e.synchronize()
        us = s.elapsed_time(e) / n * 1e3
        nz = float((cost != 0).float().mean())
        alg = 2 * B * C * h * w * a.element_size() + B * D * h * w * 4
        same = "" if ref is None else f"; bit-identical to the fp32-feature result: {bool(torch.equal(cost, ref))}"
        ref = cost if ref is None else ref
        print(f"cost_volume_fwd [12,128,48,160] x 96 bins, {what}: {us:.1f} us per call (incl. ~10 small host-issued ops), "
              f"{nz:.0%} of the (pixel, bin) pairs inside the edge mask; algorithmic bytes {alg / 1e6:.1f} MB -> "
              f"{alg / us / 1e3:.0f} GB/s{same}")


if __name__ == "__main__":
    main()
