"""Upper bound of batching the two pose passes: ResNet-18 pose encoder + decoder forward+backward, bf16 autocast,
channels_last activations: 2 x (B=12) vs 1 x (B=24).  GPU box only."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ppea-depth_amd"))
os.environ.setdefault("MIOPEN_FIND_MODE", "NORMAL")
from ppeadepth.networks import ResnetEncoder, PoseDecoder
dev = torch.device("cuda:0")
enc = ResnetEncoder(18, False, num_input_images=2).to(dev).train()
dec = PoseDecoder(enc.num_ch_enc, 1, 2).to(dev).train()
x12 = [torch.rand(12, 6, 192, 640, device=dev) for _ in range(2)]
x24 = torch.cat(x12, 0)

def run(xs):
    tot = 0
    for x in xs:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            aa, tt = dec([enc(x)])
        tot = tot + aa.float().sum() + tt.float().sum()
    tot.backward()

s = torch.cuda.Stream()


def timeit(fn, n=10):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        for _ in range(4):
            fn()
            for p in list(enc.parameters()) + list(dec.parameters()):
                p.grad = None
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print("2 x B=12: %.2f ms" % timeit(lambda: run(x12)))
print("1 x B=24: %.2f ms" % timeit(lambda: run([x24])))
