#!/usr/bin/env python3
"""Headline benchmark: training img/s at 640x192, RepLKNet-31B Stage-1 (BASELINE.json config 2/3).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One rank per GPU (RCCL).  A "step" = Trainer.process_batch (teacher + multi-frame forward, both
photometric losses) + backward + gradient all-reduce + Adam step on a synthetic batch of 12 frame
triplets per GPU already resident in HBM.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline      31x31 depthwise-conv kernel (the kernel north_star names): algorithmic bytes per
                launch / average launch duration measured with HIP events inside the timed steps.
  cpu_baseline  the CPU oracle (oracle/ref_model.py, a restatement pinned to the reference's golden
                vectors) timed on this host for one B=2 step (rank 0, N=1 only).
"""
import argparse
import json
import os
import random
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ppea-depth_amd"))
sys.path.insert(0, ROOT)

# (no MIOpen convolutions are left on the step: no find-mode warm-up)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_TF = 157.3        # fp32 vector peak
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 matrix peak


def pmc_traffic(kernel):
    """(bytes per dispatch, file) for `kernel` from the newest committed profiles/r*_pmc_*.csv, or None."""
    import csv
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_*.csv")), reverse=True):
        fetch = write = None
        with open(path) as f:
            for r in csv.DictReader(f):
                if "grid" in r and r.get("kernel", "").startswith(kernel):
                    if r["counter"] == "FETCH_SIZE":
                        fetch = float(r["mean_per_dispatch"])
                    elif r["counter"] == "WRITE_SIZE":
                        write = float(r["mean_per_dispatch"])
        if fetch is not None and write is not None:
            return int((2 * fetch + write) * 1024), os.path.relpath(path, ROOT)
    return None


def cpu_baseline(seconds_budget=40.0, eval_root=None):
    """The oracle's full training step on the host CPU: B=2, 192x640, fp32 (config 1)."""
    import types
    from oracle import model_spec, ref_model, synth          # the CPU baseline IS the oracle
    B, H, W = 2, 192, 640
    opt = types.SimpleNamespace(rep_size="b", g_blk=1.0, g_ffn=1.0, use_checkpoint=False, height=H, width=W,
                                batch_size=B, num_depth_bins=96, min_depth=0.1, max_depth=100.0,
                                disparity_smoothness=1e-3)
    sd = {k: synth.synth_tensor(k, torch.empty(shape, dtype=dt))
          for k, (shape, dt) in model_spec.state_spec("b").items()}
    sd = ref_model.leaf_state_dict(sd, opt)
    tr = ref_model.RefTrainer(ref_model.RefRepDepth(sd, opt), opt)
    torch.set_num_threads(min(torch.get_num_threads(), 16))
    absrel_oracle = None
    if eval_root is not None:      # before the training forwards below move the BN running statistics
        # AbsRel of the oracle's eval-mode inference path on the synthetic eigen_zhou-format split (the checker for
        # the "absrel" object of the bench line; pinned to the reference's Trainer.val by tests/golden/eval.npz)
        from oracle import ref_ops as R
        ds = synth.SynthEigenDataset(eval_root, split="eigen", height=H, width=W)
        batch = synth.collate([ds[i] for i in range(len(ds))])
        model = ref_model.RefRepDepth({k: v.detach() for k, v in sd.items()}, opt)
        model.training = False
        disp, mono = model.predict_val(batch, torch.tensor([0.1]), torch.tensor([10.0]))
        gts = ds.gt_depths()
        absrel_oracle = round(float(torch.stack(
            [R.evaluate_image(disp[i], torch.from_numpy(gts[i])) for i in range(len(gts))]).mean(0)[0]), 5)
    inputs = synth.make_inputs(B, H, W)
    # a GPU box shows every host core but grants a 16-core share: more threads than that only oversubscribe
    torch.set_num_threads(min(torch.get_num_threads(), 16))
    cores = torch.get_num_threads()
    times = []
    t_all = time.time()
    for it in range(3):
        for v in sd.values():
            v.grad = None
        random.seed(it)
        t0 = time.time()
        _, losses = tr.process_batch(dict(inputs))
        losses["loss"].backward()
        times.append(time.time() - t0)
        if time.time() - t_all > seconds_budget:
            break
    best = min(times[1:]) if len(times) > 1 else times[0]
    out = {"value": round(B / best, 4), "unit": "img/s", "cores": cores, "kind": "port",
           "sample": f"{len(times)} x (process_batch + backward), B=2, 192x640, RepLKNet-31B fp32, "
                     f"torch-CPU oracle, best of the non-first iterations ({best:.2f} s/step)"}
    if absrel_oracle is not None:
        out["absrel_oracle"] = absrel_oracle
    # the REFERENCE's own CPU path (unmodified Trainer.process_batch + backward of /root/reference), timed in the build
    # container by the committed harness tools/time_reference_cpu.py: the reference cannot travel to the GPU box, so its
    # timing is quoted from the committed result file next to the port timed live on this host
    ref_path = os.path.join(ROOT, "profiles", "r04_cpu_reference.json")
    if os.path.exists(ref_path):
        with open(ref_path) as f:
            r = json.load(f)
        out["reference_container"] = {
            "file": "profiles/r04_cpu_reference.json (tools/time_reference_cpu.py)",
            "host": f"{r['host']['nproc']} cores, torch {r['host']['torch']}, no GPU",
            "img_per_s_8_threads": r["reference"]["threads_8"]["img_per_s"],
            "s_per_step_8_threads": r["reference"]["threads_8"]["s_per_step_mean"],
            "img_per_s_1_thread": r["reference"]["threads_1"]["img_per_s"],
            "s_per_step_1_thread": r["reference"]["threads_1"]["s_per_step_mean"],
            "port_img_per_s_8_threads_same_host": r["port"]["threads_8"]["img_per_s"]}
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` outside a launcher: start the N ranks ourselves, one per GPU, BEFORE anything touches a
    GPU (torch.cuda.device_count() does not initialise HIP on this image), and leave with the worst exit status.  Never
    benchmarks a configuration other than the one asked for: fewer visible devices than N is an error.
    Reference launcher: `accelerate launch --multi_gpu` (train.py:39-42)."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    if n_dev < n:
        raise SystemExit(f"bench.py --gpus {n}: only {n_dev} HIP device(s) visible; refusing to run (and report) "
                         f"a different configuration")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=12, help="per-GPU batch (reference --batch_size)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--rep_size", default="b", choices=["b", "l"])
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="no hipGraph: launch every kernel from Python")
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--input_pipeline", action="store_true",
                    help="also time the device input pipeline (raw uint8 375x1242 KITTI-size frames -> row-P dict: flip, "
                         "LANCZOS pyramid, ColorJitter; the reference does this on 12 CPU workers per rank) and run the "
                         "timed steps on its output: adds an `input_pipeline` object (its own img/s) to the line")
    ap.add_argument("--dc", action="store_true",
                    help="Stage-2 decoder adapter (BASELINE config 5: --dc after dc_ft_init, Cityscapes intrinsics; the "
                         "reference trains it at 192x512, BASELINE names 512x1024: pass --height/--width)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)                # does not return

    # The contract is ONE JSON line on stdout.  RCCL prints a version banner to stdout when a communicator is created, and
    # libraries may print more: from here on fd 1 is stderr, and the line is written to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    from ppeadepth import dist as pdist
    from ppeadepth import networks, options, ops, rng
    from ppeadepth.trainer import Trainer
    from ppeadepth import synthetic as synth

    rank, local_rank, world = pdist.init_distributed()
    if os.environ.get("PPEA_STREAMS") is not None:       # parallel graph branches on/off (default on)
        on = os.environ["PPEA_STREAMS"] == "1"
        networks.repdepth.TWO_STREAMS = on
        networks.replknet_adapter.ADAPTER_STREAMS = on
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE {world} != --gpus {args.gpus}; refusing to report n_gpus != requested")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    H, W, B = args.height, args.width, args.batch
    opt = options.default_options(height=H, width=W, batch_size=B, rep_size=args.rep_size, dc=args.dc)
    torch.manual_seed(0)
    model = networks.RepDepth(opt)
    if args.dc:
        model.dc_ft_init()                       # reference Trainer.__init__, trainer.py:158-161
    synth.fill_state_dict(model)                 # deterministic random-init weights (no checkpoints offline)
    model.to(device).train()
    pdist.broadcast_module(model)
    if pdist.collectives_on():
        pdist.assign_groups(model)               # one communicator per branch (stream) of the step
    amp = torch.bfloat16 if args.dtype == "bf16" else None
    trainer = Trainer(opt, model, device, amp_dtype=amp)
    engine = pdist.TrainEngine(trainer, bf16_params=(args.dtype == "bf16"))
    rng.set_mode("device")
    inputs = {k: v.to(device) for k, v in synth.make_inputs(B, H, W, seed=1234 + rank, smooth=True,
                                                            intrinsics="cityscapes" if args.dc else "kitti").items()}
    random.seed(1000 + rank)
    pipe_info = None
    if args.input_pipeline:
        # the batch comes from raw uint8 frames through the device pipeline (bit-exact with the Pillow
        # arithmetic of the reference's loader); timed on its own, outside the training step's timed region
        from ppeadepth import input_pipeline as ipl
        gen = torch.Generator().manual_seed(77 + rank)
        raw_hw = (375, 1242)
        base = torch.nn.functional.interpolate(inputs[("color", 0, 0)], raw_hw, mode="bilinear", align_corners=False)
        raw = {f: (torch.nn.functional.interpolate(inputs[("color", f, 0)], raw_hw, mode="bilinear", align_corners=False)
                   * 255).round().clamp(0, 255).to(torch.uint8) for f in (0, -1, 1)}
        del base
        pipe = ipl.DeviceInputPipeline(raw_hw, H, W, device, K=ipl.KITTI_K)
        for _ in range(2):
            out = pipe(raw, generator=gen)
        torch.cuda.synchronize()
        t0 = time.time()
        reps = 5
        for _ in range(reps):
            out = pipe(raw, generator=gen)
        torch.cuda.synchronize()
        dt_pipe = (time.time() - t0) / reps
        pipe_info = {"img_per_s": round(B / dt_pipe, 1), "ms_per_batch": round(dt_pipe * 1e3, 2), "raw_hw": list(raw_hw),
                     "what": "uint8 frame triplets -> flip, 4-level LANCZOS pyramid, ColorJitter per frame and scale, "
                             "intrinsics (row-P dict); one frame triplet = one img"}
        inputs = {k: v.contiguous() for k, v in out.items()}

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        engine.step(dict(inputs))
    barrier()
    # roofline probe: HIP events around every k=31 launch of ONE eager step (events cannot be recorded
    # inside a captured graph); the same kernels with the same arguments are what the graph replays.
    # The probe step is launched behind ~15 ms of queued device work, so that the host stays ahead of the GPU for the
    # whole step: the launches then meet the conditions of the graph replays that the timed region consists of (a busy
    # device at its working clocks, the other streams' kernels in flight) instead of an idle GPU waking up for each one.
    head = None
    if args.dtype == "bf16":
        C0_ = 128 if args.rep_size == "b" else 192
        head = (torch.randn(B, C0_, H // 4, W // 4, device=device).bfloat16(), torch.randn(C0_, 1, 31, 31, device=device) / 31,
                torch.randn(C0_, 1, 5, 5, device=device) / 5)

    def head_start(n):
        if head is not None:
            with torch.cuda.stream(engine.stream), torch.no_grad():
                for _ in range(n):
                    ops.dwconv_lk(*head)

    torch.cuda.synchronize()
    head_start(250)
    ops.PROFILE_DWCONV = []                       # (kind, start_event, end_event) per k=31 launch
    ops.SYNC_COUNTERS = {}                        # SyncBN launches / collectives of this one eager step
    pdist.COLLECTIVE_LOG = []                     # (communicator, op, elements, dtype) of every collective, in issue order
    engine.step(dict(inputs))
    barrier()
    events, ops.PROFILE_DWCONV = ops.PROFILE_DWCONV, None
    sync_counts, ops.SYNC_COUNTERS = ops.SYNC_COUNTERS, None
    coll_log, pdist.COLLECTIVE_LOG = pdist.COLLECTIVE_LOG, None
    # the same launch (same tensors, same stream) 20x back to back between two HIP events: the kernel's duration
    # without the host gaps that events around ONE ~80 us launch include
    replay_us = {}
    with torch.cuda.stream(engine.stream):
        for kind, fn in ops.PROFILE_REPLAY.items():
            for _ in range(3):
                fn()
            s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_ev.record()
            for _ in range(20):
                fn()
            e_ev.record()
            e_ev.synchronize()
            replay_us[kind] = s_ev.elapsed_time(e_ev) / 20 * 1e3
    ops.PROFILE_REPLAY.clear()
    # What an event pair around ONE launch adds to the kernel's duration (the two marker packets between queued kernels):
    # the same pattern around a near-empty kernel, behind the same kind of head start, minus that kernel's own
    # back-to-back cost.  Subtracted from the per-launch figures below.
    event_overhead_us = 0.0
    with torch.cuda.stream(engine.stream):
        tiny = torch.zeros(64, device=device)
        torch.cuda.synchronize()
        head_start(60)
        pairs = []
        for _ in range(24):
            s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_ev.record()
            tiny.zero_()
            e_ev.record()
            pairs.append((s_ev, e_ev))
        s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_ev.record()
        for _ in range(50):
            tiny.zero_()
        e_ev.record()
        e_ev.synchronize()
        singles = sorted(a.elapsed_time(b) * 1e3 for a, b in pairs)
        event_overhead_us = max(0.0, singles[len(singles) // 2] - s_ev.elapsed_time(e_ev) / 50 * 1e3)
    head = None
    # the PLAIN 31x31 (+5x5) forward -- round 1 / 2's roofline kernel; in the step its place is taken by the variant with
    # pw1's BatchNorm + ReLU fused into the staging pass -- 20 launches back to back on the launch stream
    plain_us = None
    if args.dtype == "bf16" and rank == 0:
        with torch.cuda.stream(engine.stream), torch.no_grad():
            C0_ = 128 if args.rep_size == "b" else 192
            xx = torch.randn(B, C0_, H // 4, W // 4, device=device).bfloat16()
            wb_ = torch.randn(C0_, 1, 31, 31, device=device) / 31
            ws_ = torch.randn(C0_, 1, 5, 5, device=device) / 5
            for _ in range(3):
                ops.dwconv_lk(xx, wb_, ws_)
            s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_ev.record()
            for _ in range(20):
                ops.dwconv_lk(xx, wb_, ws_)
            e_ev.record()
            e_ev.synchronize()
            plain_us = s_ev.elapsed_time(e_ev) / 20 * 1e3
            del xx, wb_, ws_
    barrier()
    # second roofline: the kernel family that carries the most wall time of the step (profiles/r02_wall_attribution.txt) --
    # the NCHW 1x1-conv GEMM, on its largest stage-2 shape (ConvFFN pw1, 512 -> 2048 at 12x40, the batch of this run);
    # 20 launches back to back between two HIP events on the launch stream
    pw_roof = None
    if args.dtype == "bf16" and args.rep_size == "b" and rank == 0 and (H, W) == (192, 640):
        with torch.cuda.stream(engine.stream):
            xx = torch.randn(B, 512, 12, 40, device=device).bfloat16()
            aa = (torch.randn(2048, 512, device=device) / 512 ** 0.5).bfloat16()
            for _ in range(3):
                ops.pwconv_raw(aa, xx)
            s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_ev.record()
            for _ in range(20):
                ops.pwconv_raw(aa, xx)
            e_ev.record()
            e_ev.synchronize()
            t_pw = s_ev.elapsed_time(e_ev) / 20 * 1e-3
        flops = 2.0 * B * 480 * 512 * 2048
        pw_roof = {"kernel": "pwconv2_kernel<128,32,0>: 1x1 conv 512 -> 2048 on [%d,512,12,40] (ConvFFN pw1, stage 2)" % B,
                   "bound": "mfma", "achieved": round(flops / t_pw / 1e12, 1), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                   "frac": round(flops / t_pw / 1e12 / MFMA_BF16_PEAK_TF, 4), "traffic": None,
                   "back_to_back_us": round(t_pw * 1e6, 1), "flops_per_launch": flops,
                   "note": "host-paced launches; hipGraph-paced per trunk shape vs the library: profiles/r04_pwconv_shapes.txt; "
                           "algorithmic bytes %.1f MB" % ((B * 480 * (512 + 2048) + 512 * 2048) * 2 / 1e6)}
        del xx, aa
    barrier()
    # Several ranks: the captured step also holds the RCCL calls (SyncBN all-gathers / all-reduces on per-branch
    # communicators, the gradient all-reduce on a side stream).  Capture is attempted on every rank; if any
    # rank fails, all fall back to eager launches (PPEA_MULTI_GRAPH=0 skips the attempt).
    def rccl_census():
        """What RCCL saw in one step (the probe step's host-side log; the captured graph replays the same sequence):
        ranks, communicators, collectives and bytes per step, gradient ranges.  None without a process group."""
        if not pdist.collectives_on():
            return None
        esz = {"float32": 4, "bfloat16": 2, "float16": 2, "float64": 8, "int32": 4, "int64": 8}
        per = {}
        for key, op, numel, dtype in coll_log:
            d = per.setdefault(key, {"collectives": 0, "bytes": 0, "ops": {}})
            d["collectives"] += 1
            d["bytes"] += numel * esz.get(dtype, 4)
            d["ops"][op] = d["ops"].get(op, 0) + 1
        plan = engine.flat.last_plan if (engine.flat is not None and engine.flat.hooked) else []
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:                        # noqa: BLE001
            ver = None
        return {"world": dist.get_world_size(), "backend": dist.get_backend(), "rccl_version": ver,
                "communicators": sorted(per), "per_communicator": per,
                "collectives_per_step": sum(d["collectives"] for d in per.values()),
                "bytes_per_step": sum(d["bytes"] for d in per.values()),
                "grad_ranges": len(plan), "grad_ranges_from_hooks": sum(1 for p in plan if p[3]),
                "grad_bytes_per_step": sum(p[2] for p in plan) * 4,
                "forced_single_rank": bool(pdist.FORCE_COLLECTIVES and world == 1),
                "order_check": "per-communicator issue order equal across ranks and eager == capture: "
                               "tests/test_ddp_gpu.py::test_collective_order_*, test_forced_collectives_*"}

    def make_line(dt, loss_val, launch):
        # ---- roofline of the 31x31 depthwise conv (fwd launches with the fused 5x5 branch) ------
        es = 2 if args.dtype == "bf16" else 4
        C0 = 128 if args.rep_size == "b" else 192
        fwd = [s.elapsed_time(e) * 1e-3 for (kind, s, e) in events if kind == "fwd31"]
        bwd = [s.elapsed_time(e) * 1e-3 for (kind, s, e) in events if kind == "bwd31"]
        roof = None
        if fwd:
            # events around each launch inside one eager step, minus the event pair's own overhead (calibrated above): what
            # rocprofv3 reports for the same command (kernel average over the run) agrees within a few per cent
            t_raw = sum(fwd) / len(fwd)
            t_k = max(t_raw - event_overhead_us * 1e-6, 0.5 * t_raw)
            plane = B * C0 * (H // 4) * (W // 4)
            bytes_alg = plane * es * 3 + C0 * (961 + 25) * 4            # x in, y_big + y_small out, weights
            useful = 2.0 * plane * (961 + 25)
            roof = {"kernel": "31x31 depthwise conv fwd (+ fused 5x5 branch), stage-0 planes [%d,%d,%d,%d]" % (B, C0, H // 4, W // 4),
                    "bound": "hbm", "achieved": round(bytes_alg / t_k / 1e9, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(bytes_alg / t_k / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                    "avg_launch_us": round(t_k * 1e6, 1), "avg_launch_us_raw": round(t_raw * 1e6, 1),
                    "event_pair_overhead_us": round(event_overhead_us, 1), "launches_timed": len(fwd),
                    "back_to_back_us": round(replay_us["fwd31"], 1) if "fwd31" in replay_us else None,
                    "algorithmic_bytes_per_launch": bytes_alg,
                    "dgrad_avg_launch_us": round(sum(bwd) / len(bwd) * 1e6 - event_overhead_us, 1) if bwd else None,
                    "dgrad_back_to_back_us": round(replay_us["bwd31"], 1) if "bwd31" in replay_us else None,
                    "dgrad_name": "dwconv_bm_kernel<31,5,1,48,12,2,4,false> (batch-major variant; M = images)",
                    "timing_note": "avg_launch_us = HIP events around each launch INSIDE the step, the other two streams' "
                                   "kernels sharing the chip; back_to_back_us = the same launch alone.  rocprofv3 --kernel-trace "
                                   "serialises the step's streams (tools/step_timeline.py), so its average for this kernel "
                                   "(profiles/*_bench_n1_kernel_stats.csv) lies between the two"}
            if args.dtype == "bf16":
                # banded-Toeplitz MFMA kernel: (31 rows x 2 chunks + 5) v_mfma_f32_16x16x32_bf16 per 16x16 tile
                executed = (plane / 256.0) * (31 * 2 + 5) * 2.0 * 16 * 16 * 32
                from ppeadepth.networks import replknet_adapter as _rka
                fused_bn = bool(_rka.DW_BN_FUSE and not pdist.collectives_on())
                if plain_us is not None:
                    roof["plain_kernel"] = {
                        "name": "dwconv_mfma_kernel<31,5,0,5,false,true> (no fused input BatchNorm: rounds 1-2's roofline kernel)",
                        "back_to_back_us": round(plain_us, 1),
                        "achieved": round(bytes_alg / plain_us / 1e3, 1), "unit": "GB/s",
                        "frac": round(bytes_alg / plain_us / 1e3 / HBM_PEAK_GBS, 4),
                        "mfma_frac": round(executed / plain_us / 1e6 / MFMA_BF16_PEAK_TF, 3)}
                if fused_bn:
                    roof["kernel"] += "; as launched in the step: pw1's BatchNorm + ReLU applied in the staging pass " \
                                      "(the separate BN pass it replaces would move 47 MB more)"
                roof.update({"name": "dwconv_mfma_kernel<31,5,0,5,%s,true>" % ("true" if fused_bn else "false"),
                             "binding_roof": "bf16 MFMA (useful AI %d F/B; Toeplitz band executes %.2fx the useful MACs)"
                                             % (round(useful / bytes_alg), executed / useful),
                             "mfma_executed_tflops": round(executed / t_k / 1e12, 1),
                             "mfma_useful_tflops": round(useful / t_k / 1e12, 1),
                             "mfma_peak_tflops": MFMA_BF16_PEAK_TF,
                             "mfma_frac": round(executed / t_k / 1e12 / MFMA_BF16_PEAK_TF, 3)})
                if B == 12 and C0 == 128 and (H, W) == (192, 640):
                    # HBM-side bytes per launch: parsed from the committed PMC summary (separate rocprofv3 --pmc
                    # FETCH_SIZE / WRITE_SIZE passes over tools/pmc_target.py, same shape; KB per dispatch;
                    # FETCH_SIZE doubled for 16-byte-per-lane loads on gfx950 as MI355X_MICROARCH.md prescribes)
                    pmc = pmc_traffic("dwconv_mfma_kernel<31, 5, 0, 5")
                    if pmc is not None:
                        roof["traffic"] = pmc[0]
                        roof["traffic_note"] = (f"{pmc[1]}: FETCH_SIZE x2 + WRITE_SIZE per dispatch = "
                                                f"{pmc[0] / bytes_alg:.2f}x the algorithmic bytes")
            else:
                roof.update({"name": "dwconv_lk_kernel<float,31,5,...>",
                             "binding_roof": "fp32 vector FMA (AI ~ %d F/B)" % round(useful / bytes_alg),
                             "valu_tflops": round(useful / t_k / 1e12, 1),
                             "valu_frac": round(useful / t_k / 1e12 / VALU_PEAK_TF, 3)})
        line = {
            "metric": f"training img/s at {W}x{H} RepLKNet-31{args.rep_size.upper()}" + (" Stage-2 (--dc)" if args.dc else ""),
            "value": round(world * B * args.steps / dt, 3), "unit": "img/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic (smoothed uniform frames, KITTI intrinsics; random-init weights)",
            "config": {"workload": f"RepLKNet-31{args.rep_size.upper()} Stage-{2 if args.dc else 1}{' decoder adapter' if args.dc else ''} {W}x{H}, {B} frame triplets/GPU, "
                                   "process_batch + backward + grad all-reduce + Adam",
                       "global_batch": world * B, "per_gpu_batch": B, "parallelism": f"dp{world}",
                       "use_checkpoint": "BN-stat replay, no recompute (288 GB HBM)",
                       "launch": launch},
            "final_loss": round(loss_val, 5),
            "input_pipeline": pipe_info,
            "sync_bn": ({"launches_per_step": sync_counts.get("launches", 0),
                         "collectives_per_step": sync_counts.get("collectives", 0),
                         "forced_single_rank": bool(pdist.FORCE_COLLECTIVES and world == 1)}
                        if pdist.collectives_on() else None),
            "rccl": rccl_census(),
            "roofline": roof,
            "roofline_pwconv": pw_roof,
        }

        return line

    multi = pdist.collectives_on()
    use_graph = (not args.eager) and (not multi or os.environ.get("PPEA_MULTI_GRAPH", "1") == "1")

    def timed(eager):
        """EXACTLY args.steps steps between barriers; max over ranks."""
        barrier()
        t0 = time.time()
        for _ in range(args.steps):
            _, ls = engine.step(dict(inputs) if eager else inputs)
        barrier()
        el = torch.tensor([time.time() - t0], device=device, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el), float(ls["loss"].detach())

    # Several ranks: the captured step holds RCCL calls on two communicators next to the compute branches.  That
    # combination cannot run on the one-GPU development box, so it is fenced twice: the eager measurement is taken
    # FIRST, and a watchdog prints that line and ends the process if capture + replay has not finished in time.
    fallback, watchdog = None, None
    if multi and use_graph:
        fallback = timed(eager=True)
        deadline = float(os.environ.get("PPEA_GRAPH_DEADLINE_S", "240"))

        def on_timeout():
            # A hang is a FAILURE: the eager measurement taken before the attempt is printed for diagnosis, marked
            # `"hang": true`, and the process exits non-zero so the run cannot be mistaken for a successful one.
            print(f"[bench] rank {rank}: captured multi-rank step did not finish within {deadline:.0f} s -- HANG; "
                  f"printing the eager measurement taken before the attempt and exiting with status 3",
                  file=sys.stderr, flush=True)
            if rank == 0:
                emit(fallback[0], fallback[1], True, "eager (captured multi-rank step HUNG; see stderr)", hang=True)
            os._exit(3)
        watchdog = threading.Timer(deadline, on_timeout)
        watchdog.daemon = True

    def absrel(eval_root):
        """AbsRel (BASELINE metric, second half) of this build on a synthetic eigen_zhou-format split with the
        benchmark's random-init weights restored to their initial values is not possible after training steps, so a
        fresh model with the same synthetic weights is evaluated: fp32 and the benchmarked bf16 arithmetic."""
        synth.make_eval_split(eval_root, n=2, height=H, width=W, seed=7, split="eigen")
        ds = synth.SynthEigenDataset(eval_root, split="eigen", height=H, width=W)
        batch = synth.collate([ds[i] for i in range(len(ds))])
        eopt = options.default_options(height=H, width=W, batch_size=len(ds), rep_size=args.rep_size)
        emodel = networks.RepDepth(eopt)
        synth.fill_state_dict(emodel)
        emodel.to(device)
        etr = Trainer(eopt, emodel, device)
        res = {"split": "synthetic eigen_zhou-format (2 frames, 375x1242 sparse ground truth)", "weights": "random init"}
        res["fp32"] = round(float(etr.val([batch], ds.gt_depths())[0][0]), 5)
        etr.amp_dtype = torch.bfloat16
        res["bf16"] = round(float(etr.val([batch], ds.gt_depths())[0][0]), 5)
        return res

    def emit(dt, loss_val, eager, launch, hang=False):
        line = make_line(dt, loss_val, launch)
        if hang:
            line["hang"] = True
        if world == 1 and not args.no_cpu_baseline and (H, W) == (192, 640) and not args.dc:
            import tempfile
            eval_root = tempfile.mkdtemp(prefix="ppea_eval_") if args.rep_size == "b" else None
            try:
                if eval_root is not None:
                    line["absrel"] = absrel(eval_root)
                line["cpu_baseline"] = cpu_baseline(eval_root=eval_root)
                if eval_root is not None and "absrel_oracle" in line["cpu_baseline"]:
                    line["absrel"]["oracle_cpu"] = line["cpu_baseline"]["absrel_oracle"]
                    line["absrel"]["abs_diff_fp32"] = round(abs(line["absrel"]["fp32"] - line["absrel"]["oracle_cpu"]), 6)
            except Exception as ex:          # the baseline must never take the bench line down
                line["cpu_baseline"] = {"error": repr(ex)}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())

    if watchdog is not None:
        watchdog.start()
    if use_graph:
        ok = 1
        try:
            engine.capture(inputs, warmup=1)
        except Exception as ex:                  # noqa: BLE001 -- any capture failure means eager
            ok = 0
            print(f"[bench] rank {rank}: graph capture failed ({type(ex).__name__}: {ex}); eager fallback",
                  file=sys.stderr, flush=True)
        if multi:
            flag = torch.tensor([ok], device=device, dtype=torch.int32)
            torch.cuda.synchronize()
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag)
        if not ok:
            engine.graph = None
            use_graph = False
    args.eager = not use_graph
    if use_graph:
        for _ in range(2):
            engine.step(inputs)
    if use_graph or fallback is None:
        dt, loss_val = timed(eager=args.eager)
    else:
        dt, loss_val = fallback
    if watchdog is not None:
        watchdog.cancel()

    if rank == 0 and engine.flat is not None and engine.flat.hooked:
        plan = engine.flat.last_plan
        total = sum(p[2] for p in plan)
        early = sum(p[2] for p in plan if p[3])
        print(f"[bench] gradient exchange: {len(plan)} ranges, {early / max(total, 1):.0%} of {total * 4 / 1e6:.0f} MB "
              f"launched from autograd hooks during backward: "
              + ", ".join(f"{p[2] * 4 / 1e6:.0f}MB{'*' if p[3] else ''}" for p in plan), file=sys.stderr, flush=True)
    if rank == 0:
        emit(dt, loss_val, args.eager, "eager" if args.eager else "whole step captured in one hipGraph")
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
