#!/usr/bin/env python3
"""Times the REFERENCE's own CPU path (BASELINE.json configs[0]) in the build container.

    python tools/time_reference_cpu.py [--out profiles/r04_cpu_reference.json] [--timed 3]

What is timed: the unmodified `Trainer.process_batch(inputs, True)` of /root/reference
(ppeadepth/trainer.py:420-472) followed by `losses["loss"].backward()` (trainer.py:349), driven
exactly as `oracle/gen_golden.py` drives it for the `e2e_full` fixture: `oracle/ref_harness.py`
(stand-ins for the absent third-party modules only), `--adapter --weights_init scratch`,
RepLKNet-31B with the synthetic weights of `oracle/synth.py`, fp32, `model.train()`, B=2, 192x640,
KITTI intrinsics, seeds `torch.manual_seed(1)` / `random.seed(1)` before every step.
1 warm-up + N timed steps with 8 threads (all cores of this container) and with 1 thread (the
reference pins OMP/MKL_NUM_THREADS=1 at import, trainer.py:8-10).  The build's CPU restatement
(`oracle/ref_model.py`, what `bench.py`'s `cpu_baseline` times on the GPU box, where the reference
does not exist) is timed beside it on the same inputs and thread counts.

Test / measurement infrastructure only: needs /root/reference, never runs on the GPU box.
Writes a JSON file (nproc, torch version, s/step, img/s) that `bench.py` quotes as
`cpu_baseline.reference_container` and that BASELINE.md section 2 is regenerated from.
"""
import argparse
import json
import os
import platform
import random
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ppea-depth_amd"))

import torch  # noqa: E402

B, H, W = 2, 192, 640


def time_steps(step, n_timed, label):
    out = []
    for it in range(n_timed + 1):
        torch.manual_seed(1)
        random.seed(1)
        t0 = time.perf_counter()
        t_fwd = step()
        dt = time.perf_counter() - t0
        print(f"  [{label}] step {it}: {dt:.2f} s (process_batch {t_fwd:.2f} s)", flush=True)
        if it:                         # step 0 = warm-up
            out.append((dt, t_fwd))
    tot = [a for a, _ in out]
    fwd = [b for _, b in out]
    mean = sum(tot) / len(tot)
    return {"s_per_step_mean": round(mean, 3), "s_per_step_min": round(min(tot), 3), "s_per_step_max": round(max(tot), 3),
            "process_batch_s_mean": round(sum(fwd) / len(fwd), 3), "backward_s_mean": round(mean - sum(fwd) / len(fwd), 3),
            "img_per_s": round(B / mean, 4), "timed_steps": len(tot), "warmup_steps": 1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_cpu_reference.json"))
    ap.add_argument("--timed", type=int, default=3)
    ap.add_argument("--threads", type=int, nargs="*", default=[8, 1])
    args = ap.parse_args()

    from oracle import model_spec, ref_harness as rh, ref_model, synth
    if not rh.reference_available():
        raise SystemExit("the reference tree is not present: this harness runs in the build container only")
    rh.install_stubs()
    opt = rh.parse_options(["--height", str(H), "--width", str(W), "--batch_size", str(B)])
    torch.manual_seed(0)
    random.seed(0)
    with rh.scratch_cwd():
        from ppeadepth import networks           # /root/reference/ppeadepth (ref_harness put it first on sys.path)
        model = networks.RepDepth(opt)
    assert networks.__file__.startswith(rh.REFERENCE_ROOT), networks.__file__
    model.train()
    synth.fill_state_dict(model)
    tr = rh.build_reference_trainer(opt, model)
    inputs = synth.make_inputs(B, H, W)

    def ref_step():
        for p in model.parameters():
            p.grad = None
        t0 = time.perf_counter()
        _, losses = tr.process_batch(dict(inputs), True)
        t1 = time.perf_counter()
        losses["loss"].backward()
        ref_step.loss = float(losses["loss"])
        return t1 - t0

    # the build's CPU restatement on the same inputs (bench.py: cpu_baseline.kind == "port")
    popt = types.SimpleNamespace(rep_size="b", g_blk=1.0, g_ffn=1.0, use_checkpoint=False, height=H, width=W,
                                 batch_size=B, num_depth_bins=96, min_depth=0.1, max_depth=100.0,
                                 disparity_smoothness=1e-3)
    sd = {k: synth.synth_tensor(k, torch.empty(shape, dtype=dt))
          for k, (shape, dt) in model_spec.state_spec("b").items()}
    sd = ref_model.leaf_state_dict(sd, popt)
    port = ref_model.RefTrainer(ref_model.RefRepDepth(sd, popt), popt)

    def port_step():
        for v in sd.values():
            v.grad = None
        t0 = time.perf_counter()
        _, losses = port.process_batch(dict(inputs))
        t1 = time.perf_counter()
        losses["loss"].backward()
        port_step.loss = float(losses["loss"])
        return t1 - t0

    res = {"what": "unmodified /root/reference Trainer.process_batch + loss.backward() (trainer.py:420-472, 349), "
                   "BASELINE.json configs[0]: RepLKNet-31B fwd + reprojection loss on CPU, B=2, 192x640, fp32, "
                   "synthetic weights (oracle/synth.py), uniform-random frames seed 1234, KITTI intrinsics",
           "harness": "tools/time_reference_cpu.py (oracle/ref_harness.py: stand-ins for timm / torchvision / torchmetrics / "
                      "wandb / cv2 / skimage only)",
           "host": {"nproc": os.cpu_count(), "machine": platform.machine(), "python": platform.python_version(),
                    "torch": torch.__version__, "gpu_visible": bool(torch.cuda.is_available()),
                    "loadavg_at_start": [round(x, 2) for x in os.getloadavg()]},
           "batch": B, "height": H, "width": W, "reference": {}, "port": {}}
    for nt in args.threads:
        torch.set_num_threads(nt)
        print(f"reference, {nt} thread(s)", flush=True)
        res["reference"][f"threads_{nt}"] = time_steps(ref_step, args.timed, f"reference x{nt}")
        print(f"port (oracle/ref_model.py), {nt} thread(s)", flush=True)
        res["port"][f"threads_{nt}"] = time_steps(port_step, args.timed, f"port x{nt}")
    res["loss_reference"] = round(ref_step.loss, 6)
    res["loss_port"] = round(port_step.loss, 6)
    res["loss_rel_diff"] = abs(ref_step.loss - port_step.loss) / abs(ref_step.loss)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
        f.write("\n")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
