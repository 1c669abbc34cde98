#!/usr/bin/env python3
"""First point of divergence between repeated runs of the same K eager steps from one snapshot: per step, hashes of
every gradient; prints the first step at which a repeat differs from run 0 and which gradients differ there."""
import hashlib, os, random, sys, torch
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(root, "ppea-depth_amd"))
from ppeadepth import dist as pdist, networks, options, rng, synthetic as synth
from ppeadepth.trainer import Trainer
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
H, W, B, K, R = 192, 640, 12, int(os.environ.get("STEPS", "10")), int(os.environ.get("REPEATS", "6"))
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
def run():
    engine.restore(snap)
    torch.manual_seed(7); random.seed(7)
    hist = []
    for _ in range(K):
        outputs, losses = engine.step(dict(inputs))
        torch.cuda.synchronize()
        flat = engine.flat.flat
        hist.append({k: (float(v.double().abs().sum()), hashlib.md5(v.cpu().numpy().tobytes()).hexdigest()[:8])
                     for k, v in engine.named_grads().items()})
    return hist
ref = run()
for r in range(1, R):
    h = run()
    first = next((s for s in range(K) if any(h[s][k][1] != ref[s][k][1] for k in ref[s])), None)
    if first is None:
        print(f"repeat {r}: identical over {K} steps", flush=True)
        continue
    bad = [k for k in ref[first] if h[first][k][1] != ref[first][k][1]]
    print(f"repeat {r}: first difference at step {first + 1}: {len(bad)} gradients", flush=True)
    for k in bad[:40]:
        print(f"    {k}: |g| {ref[first][k][0]:.9g} vs {h[first][k][0]:.9g}")
