#!/usr/bin/env python3
"""Per-key errors of the bf16 engine step (eager and hipGraph replay) against the WELL-CONDITIONED golden
(tests/golden/e2e_render.npz: rendered frames, near-identity trunk, predicted pose = rendered pose), next to plain torch
bf16 autocast of the same model.  GPU box:  python tools/render_parity.py > profiles/r03_bf16_render_parity.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ppea-depth_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import test_e2e_gpu as T  # noqa: E402
from conftest import load_golden  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    golden = load_golden
    name = sys.argv[1] if len(sys.argv) > 1 else "e2e_render"          # e2e_render | e2e_render_l | e2e_render_dc
    cfg = T.CONFIG_OF[name]
    cols = {}
    cols["K/graph"] = T._errors(*T._engine_step(name, golden, dev, bf16=True, graph=True, **cfg))
    cols["K/eager"] = T._errors(*T._engine_step(name, golden, dev, bf16=True, graph=False, **cfg))
    with T._plain_torch_bf16():
        cols["torch"] = T._errors(*T._engine_step(name, golden, dev, bf16=True, graph=False, **cfg))
    cols["fp32"] = T._errors(*T._engine_step(name, golden, dev, bf16=False, graph=True, **cfg))
    names = list(cols)
    print(f"bf16 training step vs the reference's fp32 golden on the well-conditioned rendered fixture ({name})")
    print("K = this build's bf16 kernels (graph replay / eager), torch = plain torch bf16 autocast, fp32 = this build in fp32")
    print("%-66s" % "" + "".join("%11s" % n for n in names))
    for k in sorted(cols[names[0]]):
        if k.startswith("out:") and not k.startswith("out:relative"):
            continue
        print("%-66s" % k[:65] + "".join("%11.2e" % cols[n].get(k, float("nan")) for n in names))


if __name__ == "__main__":
    main()
