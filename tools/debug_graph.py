import os, sys, random, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ppea-depth_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch
from ppeadepth import ops
dev = torch.device("cuda:0")
which = sys.argv[1]

def cap(fn, warm=2):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(warm): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    print(which, "captured+replayed ok", flush=True)
    return out

if which == "dw_bf16":
    x = torch.randn(2, 8, 48, 160, device=dev).bfloat16(); wb = torch.randn(8, 1, 31, 31, device=dev); ws = torch.randn(8, 1, 5, 5, device=dev)
    cap(lambda: ops.dwconv_lk(x, wb, ws))
elif which == "dw_f32":
    x = torch.randn(2, 8, 48, 160, device=dev); wb = torch.randn(8, 1, 31, 31, device=dev); ws = torch.randn(8, 1, 5, 5, device=dev)
    cap(lambda: ops.dwconv_lk(x, wb, ws))
elif which == "bn":
    z = torch.randn(2, 8, 12, 40, device=dev).bfloat16(); g = torch.ones(8, device=dev); b = torch.zeros(8, device=dev)
    def f():
        m, v, i = ops.bn_batch_stats(z, 1e-5, 0.1, None, None)
        return ops.bn_act_apply(z, g, b, m, i, act=1)
    cap(f)
elif which == "loss":
    B, H, W = 2, 64, 96
    a = torch.rand(B, 3, H, W, device=dev); t = torch.rand(B, 3, H, W, device=dev)
    cap(lambda: ops.ssim_l1(a, t))
elif which in ("fwd", "fwdbwd", "step"):
    from ppeadepth import networks, options, rng
    from ppeadepth.trainer import Trainer
    from ppeadepth.dist import TrainEngine
    from ppeadepth import synthetic as synth
    B, H, W = 2, 64, 96
    opt = options.default_options(height=H, width=W, batch_size=B)
    model = networks.RepDepth(opt); synth.fill_state_dict(model); model.to(dev).train()
    tr = Trainer(opt, model, dev, amp_dtype=torch.bfloat16)
    eng = TrainEngine(tr)
    inputs = {k: v.to(dev) for k, v in synth.make_inputs(B, H, W, smooth=True).items()}
    rng.set_aug_buffer(torch.rand(B, device=dev))
    for _ in range(2): eng.step(dict(inputs))
    torch.cuda.synchronize()
    if which == "fwd":
        def f():
            with torch.no_grad():
                return tr.process_batch(dict(inputs), True)
        cap(f, warm=1)
    elif which == "fwdbwd":
        def f():
            o, l = tr.process_batch(dict(inputs), True)
            eng.grads.zero(); l["loss"].backward(); return l
        cap(f, warm=1)
    else:
        eng.capture(inputs, warmup=1)
        for _ in range(3): o, l = eng.step(inputs)
        torch.cuda.synchronize(); print("step captured ok", float(l["loss"]))
