#!/usr/bin/env python3
"""Which host-issued torch (ATen) ops does the FORWARD of one eager step launch, and from where?  A TorchDispatchMode
over one eager step of the bench configuration; ops grouped by (aten op, innermost ppeadepth frame).  (Backward runs
on autograd's thread and is not seen; tools/step_profile.py on a kernel trace covers it by kernel name.)"""
import collections, os, random, sys, traceback, torch
from torch.utils._python_dispatch import TorchDispatchMode
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
VIEWS = ("view", "reshape", "as_strided", "expand", "permute", "transpose", "slice", "select", "unsqueeze", "squeeze",
         "detach", "alias", "empty", "t.", "_unsafe_view", "unbind", "split", "chunk", "unfold", "narrow", "_to_copy_view",
         "is_", "size", "stride", "numel", "dim", "sym_", "lift", "_local_scalar", "item", "record_stream", "resize",
         "set_", "contiguous", "unflatten", "flatten", "movedim", "view_as", "result_type", "new_empty", "zeros_like_")
agg = collections.Counter()
class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        full = str(func)
        if not any(full.split("aten.")[-1].startswith(v) for v in VIEWS):
            frame = "?"
            for fs in reversed(traceback.extract_stack(limit=40)):
                if "ppeadepth" in fs.filename and "_python_dispatch" not in fs.filename:
                    frame = f"{fs.filename.split('ppeadepth/')[-1]}:{fs.lineno} {fs.name}"
                    break
            agg[(full.replace("aten.", ""), frame)] += 1
        return func(*args, **(kwargs or {}))
with Census():
    engine.step(dict(inputs))
torch.cuda.synchronize()
print(sum(agg.values()), "non-view aten ops dispatched from the host thread in one eager step")
for (name, frame), n in agg.most_common(120):
    print(f"{n:5d}  {name:36s} {frame}")
