"""Which library path is fastest for the 1x1 convs (fwd + dgrad, bf16)?  GPU box only."""
import os, sys, torch, torch.nn.functional as F
os.environ.setdefault("MIOPEN_FIND_MODE", "NORMAL")
dev = torch.device("cuda:0")

def timeit(fn, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

shapes = [(12, 128, 48, 160, 128), (12, 128, 48, 160, 512), (12, 512, 48, 160, 128),
          (12, 512, 12, 40, 512), (12, 512, 12, 40, 2048), (12, 2048, 12, 40, 512),
          (12, 1024, 6, 20, 4096), (12, 256, 24, 80, 1024), (12, 512, 12, 40, 128), (12, 128, 12, 40, 512),
          (12, 1024, 6, 20, 1024)]
for (B, Ci, H, W, Co) in shapes:
    x = torch.randn(B, Ci, H, W, device=dev, dtype=torch.bfloat16, requires_grad=True)
    w = (torch.randn(Co, Ci, 1, 1, device=dev) / Ci ** 0.5).bfloat16()
    w2 = w.view(Co, Ci)
    gy = torch.randn(B, Co, H, W, device=dev, dtype=torch.bfloat16)
    xcl = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gycl = gy.contiguous(memory_format=torch.channels_last)
    xn = x.detach().permute(0, 2, 3, 1).reshape(-1, Ci).contiguous().requires_grad_(True)
    gyn = gy.permute(0, 2, 3, 1).reshape(-1, Co).contiguous()
    fl = 2.0 * B * H * W * Ci * Co
    res = {}
    res["conv_nchw"] = (timeit(lambda: F.conv2d(x, w)), timeit(lambda: torch.autograd.grad(F.conv2d(x, w), x, gy)))
    res["matmul_nchw"] = (timeit(lambda: torch.matmul(w2, x.flatten(2))),
                          timeit(lambda: torch.autograd.grad(torch.matmul(w2, x.flatten(2)), x, gy.flatten(2))))
    res["conv_cl"] = (timeit(lambda: F.conv2d(xcl, w)), timeit(lambda: torch.autograd.grad(F.conv2d(xcl, w), xcl, gycl)))
    res["matmul_nhwc"] = (timeit(lambda: xn @ w2.t()), timeit(lambda: torch.autograd.grad(xn @ w2.t(), xn, gyn)))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ppea-depth_amd"))
    from ppeadepth import ops
    w2c = w2.contiguous(); xd = x.detach()
    res["hip_pw"] = (timeit(lambda: ops.pwconv_raw(w2c, xd)), 0.0)
    print(f"[{B},{Ci},{H},{W}]->{Co}  " + "  ".join(
        f"{k}: f {a:7.1f}us ({fl/a/1e6:5.0f}TF) f+b {b:7.1f}us" for k, (a, b) in res.items()), flush=True)
