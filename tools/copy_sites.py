#!/usr/bin/env python3
"""Which Python call sites make the layout / dtype copies of one eager bf16 step?  Wraps Tensor.contiguous / .to / .clone /
.copy_ / .float / .bfloat16 and counts the calls that really copy (by call site and shape), forward and the Python-side
backward functions of this build's autograd Functions alike."""
import collections, os, random, sys, traceback, torch
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
agg = collections.defaultdict(lambda: [0, 0])


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "ppeadepth" in fr.filename and "copy_sites" not in fr.filename:
            return f"{fr.filename.split('ppeadepth/')[-1]}:{fr.lineno}"
    return "?"


def wrap(name):
    orig = getattr(torch.Tensor, name)

    def f(self, *a, **k):
        out = orig(self, *a, **k)
        copied = name in ("clone", "copy_", "zero_", "fill_", "mul_", "add_", "div_") or (torch.is_tensor(out) and out.data_ptr() != self.data_ptr())
        if copied and self.is_cuda and self.numel() >= int(os.environ.get('MIN_NUMEL', '4096')):
            key = (name, site(), tuple(self.shape), str(self.dtype).replace("torch.", ""))
            agg[key][0] += 1
            agg[key][1] += self.numel() * self.element_size()
        return out
    setattr(torch.Tensor, name, f)


for n in ("contiguous", "to", "clone", "copy_", "float", "bfloat16", "detach_", "zero_", "fill_", "mul_", "add_", "div_"):
    wrap(n)
engine.step(dict(inputs))
torch.cuda.synchronize()
tot = sum(v[1] for v in agg.values())
print(f"{sum(v[0] for v in agg.values())} copying calls, {tot / 1e6:.0f} MB read")
for key, (n, b) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{b / 1e6:8.1f} MB {n:4d}  {key[0]:10s} {key[1]:44s} {key[2]} {key[3]}")
