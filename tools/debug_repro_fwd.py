#!/usr/bin/env python3
"""Where does a TrainEngine step stop being reproducible?  Two eager steps from the same snapshot with forward hooks on
every module (outputs) and tensor hooks on every module output (gradients): prints the first modules, in execution order,
whose forward output / output gradient differ bit for bit between the two runs.

    python tools/debug_repro_fwd.py [--dtype bf16|f32] [--B 2 --H 64 --W 96]
"""
import argparse
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ppea-depth_amd"))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--B", type=int, default=2)
    ap.add_argument("--H", type=int, default=64)
    ap.add_argument("--W", type=int, default=96)
    ap.add_argument("--streams", type=int, default=1)
    args = ap.parse_args()
    from ppeadepth import dist as pdist, networks, options, rng, synthetic as synth
    from ppeadepth.trainer import Trainer
    if not args.streams:
        networks.repdepth.TWO_STREAMS = False
        networks.replknet_adapter.ADAPTER_STREAMS = False
    dev = torch.device("cuda:0")
    B, H, W = args.B, args.H, args.W
    opt = options.default_options(height=H, width=W, batch_size=B, use_checkpoint=True)
    model = networks.RepDepth(opt)
    synth.fill_state_dict(model)
    model.to(dev).train()
    bf16 = args.dtype == "bf16"
    tr = Trainer(opt, model, dev, amp_dtype=torch.bfloat16 if bf16 else None)
    eng = pdist.TrainEngine(tr, lr=1e-4, bf16_params=bf16)
    rng.set_mode("reference")
    inputs = {k: v.to(dev).contiguous() for k, v in synth.make_inputs(B, H, W, seed=1234, smooth=True).items()}
    snap = eng.snapshot()
    names = {m: n for n, m in model.named_modules()}
    rec = None

    def fwd_hook(mod, inp, out):
        outs = out if isinstance(out, (tuple, list)) else (out,)
        for j, o in enumerate(outs):
            if torch.is_tensor(o) and o.is_floating_point():
                key = f"{names[mod]}[{j}]#{rec['count'].setdefault(names[mod], 0)}"
                rec["fwd"].append((key, o.detach().clone()))
                if o.requires_grad:
                    o.register_hook(lambda g, key=key: rec["bwd"].append((key, g.detach().clone())))
        rec["count"][names[mod]] = rec["count"].get(names[mod], 0) + 1

    for m in model.modules():
        if m is not model:
            m.register_forward_hook(fwd_hook)

    def run():
        nonlocal rec
        rec = {"fwd": [], "bwd": [], "count": {}}
        eng.restore(snap)
        torch.manual_seed(3)
        random.seed(3)
        _, losses = eng.step(dict(inputs))
        torch.cuda.synchronize()
        print(f"loss {float(losses['loss'].detach()):.9f}  ({len(rec['fwd'])} forward outputs, {len(rec['bwd'])} gradients recorded)",
              flush=True)
        return rec

    a, b = run(), run()
    for kind in ("fwd", "bwd"):
        da, db = dict(a[kind]), dict(b[kind])
        shown = 0
        for key, t in a[kind]:
            if key in db and not torch.equal(t, db[key]):
                u = db[key].float()
                print(f"  {kind} differs: {key:80s} shape {tuple(t.shape)} max|d| {float((t.float() - u).abs().max()):.3g} of {float(u.abs().max()):.3g}, "
                      f"{int((t != db[key]).sum())} elements")
                shown += 1
                if shown >= 12:
                    break
        if not shown:
            print(f"  {kind}: all {len(da)} recorded tensors identical")


if __name__ == "__main__":
    main()
