#!/usr/bin/env python3
"""Is one training step a pure function of (state, inputs, seeds)?  Runs the SAME eager bf16 step several times from the
same snapshot (same torch / python seeds) and reports which gradients, outputs and losses differ bitwise between runs."""
import os, random, sys, torch
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(root, "ppea-depth_amd"))
from ppeadepth import dist as pdist, networks, options, rng, synthetic as synth
from ppeadepth.trainer import Trainer
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
H, W, B = 192, 640, int(os.environ.get("B", "12"))
opt = options.default_options(height=H, width=W, batch_size=B, rep_size="b")
torch.manual_seed(0)
model = networks.RepDepth(opt); synth.fill_state_dict(model); model.to(dev).train()
trainer = Trainer(opt, model, dev, amp_dtype=torch.bfloat16)
engine = pdist.TrainEngine(trainer, bf16_params=True)
rng.set_mode("device")
inputs = {k: v.to(dev) for k, v in synth.make_inputs(B, H, W, seed=1234, smooth=True).items()}
random.seed(3)
engine.step(dict(inputs))
snap = engine.snapshot()
runs = []
for r in range(4):
    engine.restore(snap)
    torch.manual_seed(7); random.seed(7)
    trace = []
    for _ in range(int(os.environ.get("STEPS", "1"))):
        outputs, losses = engine.step(dict(inputs))
        trace.append((float(losses["loss"].detach()), float(outputs[("cam_T_cam", 0, -1)].detach().float().abs().sum()),
                      float(engine.named_grads()["pose_encoder.encoder.conv1.weight"].abs().sum()),
                      float(engine.named_grads()["pose.net.3.weight"].abs().sum()),
                      float(engine.named_grads()["depth.upconvs_0.0.conv.conv.weight"].abs().sum())))
    torch.cuda.synchronize()
    print("trace", r, " | ".join(f"{a:.9f} {b:.7f} {c:.7f} {d:.7f} {e:.7f}" for a, b, c, d, e in trace[-4:]))
    g = {k: v.detach().clone() for k, v in engine.named_grads().items()}
    o = {str(k): v.detach().clone() for k, v in outputs.items() if torch.is_tensor(v)}
    runs.append((g, o, {k: float(v) for k, v in losses.items()}))
for r in range(1, 4):
    dg = [k for k in runs[0][0] if not torch.equal(runs[0][0][k], runs[r][0][k])]
    do = [k for k in runs[0][1] if k in runs[r][1] and not torch.equal(runs[0][1][k], runs[r][1][k])]
    print(f"run {r} vs run 0: {len(dg)} of {len(runs[0][0])} gradients differ, {len(do)} of {len(runs[0][1])} outputs differ; loss {runs[r][2]['loss']!r} vs {runs[0][2]['loss']!r}")
    print("   outputs:", do[:24])
    print("   grads  :", dg[:12])
import hashlib
h = hashlib.md5()
for k in sorted(runs[0][0]):
    h.update(runs[0][0][k].float().cpu().numpy().tobytes())
print("across processes: loss", repr(runs[0][2]["loss"]), "grad checksum", h.hexdigest()[:12])
bad = {}
if os.environ.get("DUMP"):
    torch.save({k: v.float().cpu() for k, v in runs[0][0].items()}, os.environ["DUMP"])
if os.environ.get("CMP") and os.path.exists(os.environ["CMP"]):
    ref = torch.load(os.environ["CMP"])
    d = [k for k in ref if not torch.equal(ref[k], runs[0][0][k].float().cpu())]
    print(len(d), "gradients differ from the other process:", d[:20])
