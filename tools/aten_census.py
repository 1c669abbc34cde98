#!/usr/bin/env python3
"""Census of the device kernels of ONE eager bf16 step that do NOT come from this build's library (ATen element-wise /
copy / reduce kernels, runtime buffer copies, memsets): how many, how long, and which host op launched them -- forward
AND backward (autograd's thread included), grouped by (kernel family, aten op, enclosing autograd node or forward scope,
shapes).  The list VERDICT r3 #6 asks to shrink.

    python tools/aten_census.py [--top 60] > profiles/r04_aten_census.txt
"""
import argparse
import collections
import os
import random
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ppea-depth_amd"))

import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402


def family(name):
    if "at::" not in name and "rocclr" not in name and "Memcpy" not in name and "Memset" not in name:
        return None                                       # this build's kernels (and RCCL)
    if "copyBuffer" in name or "Memcpy" in name:
        return "runtime copyBuffer"
    if "Memset" in name or "fillBuffer" in name:
        return "runtime fill"
    if "direct_copy" in name or "copy_kernel" in name:
        return "aten copy"
    if "reduce_kernel" in name:
        return "aten reduce"
    if "multi_tensor_apply" in name:
        return "aten multi_tensor"
    if "CatArray" in name:
        return "aten cat"
    if "max_pool" in name:
        return "aten max_pool"
    if "upsample" in name:
        return "aten upsample"
    if "elementwise" in name or "vectorized" in name:
        m = re.search(r"(\w+Functor\w*|\w+_kernel_cuda|\w+KernelImpl|CUDAFunctor_\w+)", name)
        return "aten elementwise " + (m.group(1)[:28] if m else "")
    if "at::native" in name or "at_cuda_detail" in name:
        return "aten other"
    return None                                           # this build's kernels (and RCCL)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=70)
    ap.add_argument("--batch", type=int, default=12)
    args = ap.parse_args()
    from ppeadepth import dist as pdist, networks, options, rng, synthetic as synth
    from ppeadepth.trainer import Trainer
    device = torch.device("cuda", 0)
    H, W, B = 192, 640, args.batch
    opt = options.default_options(height=H, width=W, batch_size=B)
    torch.manual_seed(0)
    model = networks.RepDepth(opt)
    synth.fill_state_dict(model)
    model.to(device).train()
    trainer = Trainer(opt, model, device, amp_dtype=torch.bfloat16)
    engine = pdist.TrainEngine(trainer, bf16_params=True)
    rng.set_mode("device")
    inputs = {k: v.to(device) for k, v in synth.make_inputs(B, H, W, seed=1234, smooth=True).items()}
    random.seed(1000)
    for _ in range(2):
        engine.step(dict(inputs))
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        engine.step(dict(inputs))
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    fam_tot = collections.defaultdict(lambda: [0, 0.0])
    ours = [0, 0.0]
    for ev in prof.events():
        if not ev.kernels:
            continue
        scope, p = "forward / host", ev.cpu_parent
        chain = []
        while p is not None:
            chain.append(p.name)
            p = p.cpu_parent
        for nm in chain:
            if nm.startswith("autograd::engine::evaluate_function: "):
                scope = nm.split(": ", 1)[1]
        top_op = ev.name
        for nm in chain:                                   # outermost aten op below the autograd node
            if nm.startswith("aten::"):
                top_op = nm
        shapes = str(ev.input_shapes)[:60] if ev.input_shapes else ""
        for k in ev.kernels:
            fam = family(k.name)
            if fam is None:
                ours[0] += 1
                ours[1] += k.duration
                continue
            key = (fam, top_op, scope, shapes)
            agg[key][0] += 1
            agg[key][1] += k.duration
            fam_tot[fam][0] += 1
            fam_tot[fam][1] += k.duration
    n = sum(v[0] for v in fam_tot.values())
    t = sum(v[1] for v in fam_tot.values())
    print(f"one eager bf16 step, B = {B}, {W}x{H}: {n} non-library device launches, {t / 1e3:.2f} ms kernel-sum "
          f"(this build's kernels: {ours[0]} launches, {ours[1] / 1e3:.2f} ms)")
    for fam, (c, d) in sorted(fam_tot.items(), key=lambda kv: -kv[1][1]):
        print(f"  {fam:46s} {c:5d} launches {d / 1e3:7.3f} ms")
    print()
    print(f"{'launches':>8s} {'ms':>7s}  family | aten op | autograd node / scope | input shapes")
    for (fam, op, scope, shapes), (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]:
        print(f"{c:8d} {d / 1e3:7.3f}  {fam} | {op} | {scope} | {shapes}")


if __name__ == "__main__":
    main()
