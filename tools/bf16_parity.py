#!/usr/bin/env python3
"""Prints how far the benchmarked arithmetic (bf16 autocast + bf16 working weights + MFMA kernels), eager and replayed
from a hipGraph, lands from the fp32 reference goldens -- the numbers behind the tolerances in tests/test_e2e_gpu.py.

    python tools/bf16_parity.py [e2e_small e2e_full ...]
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ppea-depth_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import test_e2e_gpu as T  # noqa: E402
from conftest import load_golden  # noqa: E402


def main():
    names = sys.argv[1:] or ["e2e_small", "e2e_full"]
    dev = torch.device("cuda:0")
    for name in names:
        g = load_golden(name)
        for bf16 in (False, True):
            for graph in (False, True):
                res = T._engine_step(name, lambda n: g, dev, bf16=bf16, graph=graph, **T.CONFIG_OF.get(name, {}))
                errs = T._errors(*res)
                tag = f"{name} {'bf16' if bf16 else 'fp32'} {'graph' if graph else 'eager'}"
                worst = {}
                for k, v in errs.items():
                    grp = k.split(":")[0]
                    worst[grp] = max(worst.get(grp, 0.0), v)
                print(tag, {k: f"{v:.2e}" for k, v in worst.items()}, flush=True)
                top = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
                print("   worst:", [(k, f"{v:.2e}") for k, v in top], flush=True)


if __name__ == "__main__":
    main()
