#!/usr/bin/env python3
"""Which Python call sites launch the tiny element-wise kernels of one eager bf16 step?  A TorchFunctionMode counts the torch
functions whose result is a small device tensor (<= 4096 elements), by call site -- forward and the Python-side backward
functions of this build's autograd Functions (the autograd engine's own C++ nodes are not visible here)."""
import collections, os, random, sys, traceback, torch
from torch.overrides import TorchFunctionMode
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
agg = collections.Counter()


class Census(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        t = out[0] if isinstance(out, (tuple, list)) and out and torch.is_tensor(out[0]) else out
        if torch.is_tensor(t) and t.is_cuda and t.numel() <= int(os.environ.get('MAX_NUMEL', '4096')):
            name = getattr(func, "__name__", str(func))
            if name not in ("view", "reshape", "__getitem__", "detach", "size", "dim", "to", "float", "contiguous", "unsqueeze",
                            "squeeze", "expand", "permute", "transpose", "t", "view_as", "__get__", "data_ptr", "is_contiguous",
                            "stride", "numel", "type", "requires_grad_", "unbind", "chunk", "split", "flatten", "narrow", "select", "empty",
                            "empty_like", "apply", "record_stream", "is_floating_point", "element_size", "storage_offset"):
                site = "?"
                for fr in reversed(traceback.extract_stack()[:-1]):
                    if "ppeadepth" in fr.filename:
                        site = f"{fr.filename.split('ppeadepth/')[-1]}:{fr.lineno}"
                        break
                agg[(site, name, tuple(t.shape))] += 1
        return out


with Census():
    engine.step(dict(inputs))
torch.cuda.synchronize()
print(sum(agg.values()), "small-result torch calls in one step")
by_site = collections.Counter()
for (site, name, shape), n in agg.items():
    by_site[site] += n
for site, n in by_site.most_common(40):
    ops_ = collections.Counter({k[1]: v for k, v in agg.items() if k[0] == site})
    print(f"{n:5d}  {site:44s} {dict(ops_.most_common(5))}")
