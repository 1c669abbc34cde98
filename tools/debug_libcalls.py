#!/usr/bin/env python3
"""Which library convolutions / GEMMs does one bf16 engine step still call?  Logs every F.conv2d / F.conv_transpose2d /
F.linear / bmm / baddbmm / matmul call with its shapes.   python tools/debug_libcalls.py [b|l|dc]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ppea-depth_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
import traceback

cfg = sys.argv[1] if len(sys.argv) > 1 else "dc"
import test_e2e_gpu as T
from oracle import synth
from ppeadepth import rng
from ppeadepth.dist import TrainEngine

dev = torch.device("cuda:0")
B, H, W = 1, 192, (512 if cfg == "dc" else 640)
opt, model, tr = T._build(dev, B, H, W, use_checkpoint=True, amp=torch.bfloat16, rep_size="l" if cfg == "l" else "b",
                          dc=(cfg == "dc"), conditioned=True)
rng.set_mode("device")
eng = TrainEngine(tr, lr=1e-4, bf16_params=True)
inputs = {k: v.to(dev) for k, v in synth.make_rendered_inputs(B, H, W, intrinsics="cityscapes" if cfg == "dc" else "kitti").items()}
eng.step(dict(inputs))
seen = {}


def wrap(mod, name):
    orig = getattr(mod, name)

    def f(*a, **k):
        shapes = tuple(tuple(t.shape) for t in a if torch.is_tensor(t))
        if any(len(s) >= 2 and max(s) > 8 for s in shapes):
            key = (name, shapes)
            if key not in seen:
                seen[key] = "".join(traceback.format_stack(limit=6)[:-1])
        return orig(*a, **k)
    setattr(mod, name, f)


for n in ("conv2d", "conv_transpose2d", "linear"):
    wrap(F, n)
for n in ("bmm", "baddbmm", "matmul", "mm", "addmm"):
    wrap(torch, n)
eng.step(dict(inputs))
torch.cuda.synchronize()
for (name, shapes), tb in seen.items():
    print(name, shapes)
    print(tb)
