#!/usr/bin/env python3
"""Where the wall time of the CAPTURED step goes, without a profiler: one-lane timestamp launches (ppea_timestamp: the
device's 100 MHz clock) are recorded into the step graph at module boundaries -- forward hooks, gradient hooks on module
outputs (= the moment backward reaches the module), post-accumulate hooks on first-layer parameters (= a branch's backward is
over) -- on whatever stream the point runs on.  rocprofv3's kernel trace serialises the three streams of the step (kernels in
flight > 1 for 10 % of the time against ~1.4x overlap unprofiled), so it cannot show which chain is the critical one; this does.

    python tools/step_timeline.py [--replays 5] > profiles/r04_step_timeline.txt
Environment switches of the step (PPEA_POSE_SIDE, PPEA_MONO_SIDE, ...) apply as in bench.py.
"""
import argparse
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ppea-depth_amd"))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replays", type=int, default=5)
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--sync_each", action="store_true",
                    help="synchronize after every replay (exposes the host's submission time at the start of the step)")
    args = ap.parse_args()
    from ppeadepth import _abi, dist as pdist, networks, options, ops, rng, synthetic as synth
    from ppeadepth.trainer import Trainer
    dev = torch.device("cuda", 0)
    H, W, B = 192, 640, args.batch
    opt = options.default_options(height=H, width=W, batch_size=B)
    torch.manual_seed(0)
    model = networks.RepDepth(opt)
    synth.fill_state_dict(model)
    model.to(dev).train()
    trainer = Trainer(opt, model, dev, amp_dtype=torch.bfloat16)
    engine = pdist.TrainEngine(trainer, bf16_params=True)
    rng.set_mode("device")
    inputs = {k: v.to(dev) for k, v in synth.make_inputs(B, H, W, seed=1234, smooth=True).items()}
    random.seed(1000)

    slots = torch.zeros(512, dtype=torch.int64, device=dev)
    marks = []                                             # (label, cuda_stream handle) in capture (host) order

    def mark(label):
        if not torch.cuda.is_current_stream_capturing():
            return
        i = len(marks)
        st = torch.cuda.current_stream()
        _abi.call("ppea_timestamp", _abi.ptr(slots[i:i + 1]), _abi.stream_ptr())
        marks.append((label, st.cuda_stream))

    def first_tensor(o):
        if torch.is_tensor(o):
            return o if o.requires_grad else None
        if isinstance(o, dict):
            o = list(o.values())
        if isinstance(o, (list, tuple)):
            for v in o:
                t = first_tensor(v)
                if t is not None:
                    return t
        return None

    calls = {}

    def watch(mod, name):
        def pre(_m, _i):
            k = calls[name] = calls.get(name, 0) + 1
            mark(f"{name} #{k} fwd start")

        def post(_m, _i, out):
            k = calls[name]
            mark(f"{name} #{k} fwd end")
            t = first_tensor(out)
            if t is not None:
                t.register_hook(lambda g, k=k: (mark(f"{name} #{k} bwd start"), None)[1])
        mod.register_forward_pre_hook(pre)
        mod.register_forward_hook(post)

    watch(model.mono_encoder, "teacher encoder")
    watch(model.mono_depth, "teacher decoder")
    watch(model.pose_encoder, "pose encoder")
    watch(model.encoder.replk.stem[0], "student stem conv")
    for i, st in enumerate(model.encoder.replk.stages):
        watch(st, f"student stage {i}")
    watch(model.encoder.reduce_conv, "student reduce_conv")
    watch(model.depth, "student decoder")
    for i, st in enumerate(model.mono_encoder.stages):
        watch(st, f"teacher stage {i}")
    names = dict(model.named_parameters())
    for label, key in (("teacher backward end", "mono_encoder.stages.0.blocks.0.adapter.D_fc1.weight"),
                       ("student backward end", "encoder.replk.stages.0.blocks.0.adapter.D_fc1.weight"),
                       ("student stage-1 backward end", "encoder.replk.stages.1.blocks.0.adapter.D_fc1.weight"),
                       ("pose backward end", "pose_encoder.encoder.conv1.weight")):
        names[key].register_post_accumulate_grad_hook(lambda _p, label=label: mark(label))

    def wrap(obj, attr, label):
        fn = getattr(obj, attr)

        def wrapped(*a, **k):
            mark(label + " start")
            r = fn(*a, **k)
            mark(label + " end")
            return r
        setattr(obj, attr, wrapped)

    wrap(model.encoder, "compute_depth_bins", "student depth bins")
    wrap(model.encoder.replk, "plan_drop_masks", "student drop masks")
    wrap(model.mono_encoder, "plan_drop_masks", "teacher drop masks")
    wrap(model, "predict_poses", "predict_poses")
    wrap(trainer.depth_bin_tracker, "compute", "tracker.compute")
    wrap(trainer.depth_bin_tracker, "update", "tracker.update")
    wrap(trainer, "generate_images_pred", "generate_images_pred")

    orig_losses = trainer.compute_losses

    def compute_losses(inp, out, is_multi=False):
        mark(("student" if is_multi else "teacher") + " loss start")
        r = orig_losses(inp, out, is_multi=is_multi)
        mark(("student" if is_multi else "teacher") + " loss end")
        return r
    trainer.compute_losses = compute_losses
    orig_pb = trainer.process_batch

    def process_batch(inp, is_train=False):
        calls.clear()
        mark("step start")
        r = orig_pb(inp, is_train)
        mark("process_batch end (backward starts)")
        return r
    trainer.process_batch = process_batch
    orig_opt = engine._optimizer_phase

    def optimizer_phase():
        mark("backward returned (all streams joined)")
        orig_opt()
        mark("optimizer end")
    engine._optimizer_phase = optimizer_phase

    engine.capture(inputs, warmup=2)
    stream_name = {engine.stream.cuda_stream: "step", ops.side_stream_of(engine.stream).cuda_stream: "side"}
    for _ in range(3):
        engine.step(inputs)
    torch.cuda.synchronize()
    acc = None
    evs = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    evs[0].record()
    if args.sync_each:
        for _ in range(args.replays):
            engine.step(inputs)
            torch.cuda.synchronize()
            t = slots[:len(marks)].cpu().double()
            t = (t - t[0]) / 100.0                        # 100 MHz -> microseconds since "step start"
            acc = t if acc is None else acc + t
        evs[1].record()
        torch.cuda.synchronize()
        acc = acc / args.replays
    else:
        # back to back, as a training loop (and bench.py) replays them: the host submits replay k + 1 while the device runs
        # replay k; the timestamps are those of the LAST replay (each replay overwrites the slots)
        for _ in range(args.replays):
            engine.step(inputs)
        evs[1].record()
        torch.cuda.synchronize()
        t = slots[:len(marks)].cpu().double()
        acc = (t - t[0]) / 100.0
    how = "mean of" if args.sync_each else "last of"
    print(f"captured step, B = {B}, {W}x{H}, bf16: {len(marks)} timestamps, {how} {args.replays} replays "
          f"(replay + synchronize: {evs[0].elapsed_time(evs[1]) / args.replays:.2f} ms each); switches: "
          + " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("PPEA_")))
    print(f"{'ms':>8s}  {'stream':7s} point")
    for i in sorted(range(len(marks)), key=lambda i: float(acc[i])):
        label, st = marks[i]
        print(f"{float(acc[i]) / 1e3:8.3f}  {stream_name.get(st, 'teacher'):7s} {label}")


if __name__ == "__main__":
    main()
