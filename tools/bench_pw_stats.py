import os, sys, torch
sys.path.insert(0, "/root/repo/ppea-depth_amd")
from ppeadepth import ops
dev = torch.device("cuda:0")
def graph_time(fn, REP=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(REP): fn()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * REP) * 1e3
for (B, K, H, W, M) in [(12, 512, 12, 40, 2048), (12, 128, 48, 160, 512), (12, 512, 12, 40, 512), (12, 256, 24, 80, 1024)]:
    x = torch.randn(B, K, H, W, device=dev).bfloat16()
    w = (torch.randn(M, K, 1, 1, device=dev) / K ** 0.5)
    with torch.no_grad():
        t0 = graph_time(lambda: ops.pwconv_frozen(x, w))
        t1 = graph_time(lambda: ops.pwconv_frozen(x, w, want_sums=True))
    print(f"{K}->{M} @{H}x{W}: plain {t0:.1f} us, with statistics epilogue {t1:.1f} us")
