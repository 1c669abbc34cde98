#!/usr/bin/env python3
"""Is one TrainEngine step a pure function of (state, inputs, seeds)?  Runs the eager step twice from the same in-memory
snapshot (and once more replayed from a hipGraph) and reports what differs, bit for bit.

    python tools/debug_repro.py                       # default one-rank path
    PPEA_FORCE_COLLECTIVES=1 python tools/debug_repro.py   # every collective of the multi-rank path through RCCL (1 rank)
Options: --dtype bf16|f32, --rng reference|device, --no-graph
"""
import argparse
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ppea-depth_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29000 + os.getpid() % 500))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--rng", default="reference")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--B", type=int, default=2)
    ap.add_argument("--H", type=int, default=64)
    ap.add_argument("--W", type=int, default=96)
    ap.add_argument("--alias-fork", action="store_true",
                    help="after the default runs: the same step with the second-consumer alias handed to the FORKED adapters too "
                         "(the topology DESIGN 5 recorded as a capture race), eager and replayed, against the default eager step")
    args = ap.parse_args()
    from ppeadepth import dist as pdist, networks, options, rng, synthetic as synth
    from ppeadepth.trainer import Trainer
    pdist.init_distributed("nccl")
    from ppeadepth.networks import replknet_adapter as _rka
    if args.alias_fork:
        _rka.BN_DUP_FORKED = False                 # the "default" runs below: alias for the in-line adapters only
    dev = torch.device("cuda:0")
    B, H, W = args.B, args.H, args.W
    opt = options.default_options(height=H, width=W, batch_size=B, use_checkpoint=True)
    model = networks.RepDepth(opt)
    synth.fill_state_dict(model)
    model.to(dev).train()
    if pdist.collectives_on():
        pdist.assign_groups(model)
    bf16 = args.dtype == "bf16"
    tr = Trainer(opt, model, dev, amp_dtype=torch.bfloat16 if bf16 else None)
    eng = pdist.TrainEngine(tr, lr=1e-4, bf16_params=bf16)
    rng.set_mode(args.rng)
    inputs = {k: v.to(dev).contiguous() for k, v in synth.make_inputs(B, H, W, seed=1234, smooth=True).items()}
    snap = eng.snapshot()
    print(f"collectives_on={pdist.collectives_on()} dtype={args.dtype} rng={args.rng} hooked={eng.flat.hooked}", flush=True)

    def run(tag):
        eng.restore(snap)
        torch.manual_seed(3)
        torch.cuda.manual_seed(3)
        random.seed(3)
        outputs, losses = eng.step(dict(inputs) if eng.graph is None else inputs)
        torch.cuda.synchronize()
        res = {"loss:" + k: v.detach().clone() for k, v in losses.items()}
        for k, v in outputs.items():
            if torch.is_tensor(v):
                res["out:" + str(k)] = v.detach().clone()
        for k, v in eng.named_grads().items():
            res["grad:" + k] = v.detach().clone()
        for k, v in model.state_dict().items():
            res["state:" + k] = v.detach().clone()
        print(f"[{tag}] loss {float(losses['loss']):.9f}", flush=True)
        return res

    def compare(a, b, what):
        diff = [k for k in a if not torch.equal(a[k], b[k])]
        by = {}
        for k in diff:
            by.setdefault(k.split(":")[0], []).append(k)
        print(f"{what}: {len(diff)} of {len(a)} tensors differ: " + ", ".join(f"{g} {len(v)}" for g, v in by.items()), flush=True)
        for g, v in by.items():
            if g != "grad":
                for k in v[:12]:
                    x, y = a[k].float(), b[k].float()
                    print(f"      {k}  max|d| {float((x - y).abs().max()):.3g}  of {float(y.abs().max()):.3g}")
        if "grad" in by:
            order = list(eng.named_grads())
            idx = sorted(order.index(k[5:]) for k in by["grad"])
            print(f"      grads: first differing (in backward order) #{idx[0]} {order[idx[0]]}; equal ones: "
                  f"{[k for k in order if 'grad:' + k not in set(by['grad'])][:6]}")

    e1 = run("eager 1")
    e2 = run("eager 2")
    compare(e1, e2, "eager vs eager")
    if not args.no_graph:
        eng.restore(snap)
        torch.manual_seed(3)
        random.seed(3)
        eng.capture(inputs, warmup=1, restore_state=True)
        g1 = run("graph 1")
        g2 = run("graph 2")
        compare(g1, g2, "graph vs graph")
        compare(g1, e1, "graph vs eager")
    if args.alias_fork:
        from ppeadepth.networks import replknet_adapter as rka
        rka.BN_DUP_FORKED = True
        eng.graph = None
        f1 = run("alias+fork eager")
        compare(f1, e1, "alias+fork eager vs default eager")
        for i in range(3):
            eng.graph = None
            eng.restore(snap)
            torch.manual_seed(3)
            random.seed(3)
            eng.capture(inputs, warmup=1, restore_state=True)
            fg = run(f"alias+fork graph, capture {i}")
            compare(fg, e1, f"alias+fork graph (capture {i}) vs default eager")
            fg2 = run(f"alias+fork graph, capture {i}, second replay")
            compare(fg2, e1, f"alias+fork graph (capture {i}, 2nd replay) vs default eager")
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
