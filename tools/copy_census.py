#!/usr/bin/env python3
"""Where do the glue kernels of one eager bf16 step come from?  torch.profiler with Python stacks: device time of every
aten op that is not one of this build's kernels, grouped by (op, innermost ppeadepth frame), forward AND backward."""
import collections, os, random, sys, torch
from torch.profiler import ProfilerActivity, profile
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(root, "ppea-depth_amd"))
from ppeadepth import dist as pdist, networks, options, rng, synthetic as synth
from ppeadepth.trainer import Trainer
device = torch.device("cuda", 0); torch.cuda.set_device(device)
H, W, B = 192, 640, 12
opt = options.default_options(height=H, width=W, batch_size=B, rep_size="b")
torch.manual_seed(0)
model = networks.RepDepth(opt); synth.fill_state_dict(model); model.to(device).train()
trainer = Trainer(opt, model, device, amp_dtype=torch.bfloat16)
engine = pdist.TrainEngine(trainer, bf16_params=True)
rng.set_mode("device")
inputs = {k: v.to(device) for k, v in synth.make_inputs(B, H, W, seed=1234, smooth=True).items()}
random.seed(1000)
for _ in range(2):
    engine.step(dict(inputs))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    engine.step(dict(inputs))
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
shapes = collections.defaultdict(collections.Counter)
for ev in prof.events():
    t = getattr(ev, "self_device_time_total", 0)
    if not ev.name.startswith("aten::") or t <= 0:
        continue
    frame = "(autograd engine / no python frame)"
    for fr in (ev.stack or []):
        if "ppeadepth/" in fr:
            frame = fr.split("ppeadepth/")[-1][:80]
            break
    key = (ev.name, frame)
    agg[key][0] += 1
    agg[key][1] += t
    shapes[key][str(ev.input_shapes)[:90]] += 1
tot = sum(v[1] for v in agg.values())
print(f"{tot / 1e3:.2f} ms of device time in aten ops (self) in one eager step")
for key, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:50]:
    print(f"{t / 1e3:8.3f} ms {n:5d}  {key[0]:28s} {key[1]}")
    for sh, k in shapes[key].most_common(3):
        print(f"                    {k:4d} x {sh}")
