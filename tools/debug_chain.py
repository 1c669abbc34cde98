import os, sys, torch
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "ppea-depth_amd")); sys.path.insert(0, os.path.join(root, "tests"))
import test_kernels_gpu as T
from oracle import synth
from ppeadepth import batchnorm, rng
from ppeadepth.networks import replknet_adapter as rka
device = torch.device("cuda:0")
C, K, H, W, B = 128, 13, 12, 20, 3
def run(chain, streams=True):
    batchnorm.BN_CHAIN = chain
    rka.ADAPTER_STREAMS = streams
    st = rka.RepLKNetStage(C, 2, K, [0.1, 0.2], 5, adpt_test=4, ratio=0.25)
    synth.fill_state_dict(st)
    st = st.to(device).train()
    for n, p in st.named_parameters():
        p.requires_grad = "adapter" in n or ".bn" in n or "_bn" in n
    g = T._g(5)
    x = torch.randn(B, C, H, W, generator=g).to(device).requires_grad_(True)
    go = torch.randn(B, C, H, W, generator=g).to(device)
    rng.set_mode("reference"); torch.manual_seed(11)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = st(x.bfloat16())
    y.backward(go.bfloat16())
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in st.named_parameters() if p.grad is not None}
for label, a, b in (("F streams vs F none", run(False, True), run(False, False)), ("T streams vs T none", run(True, True), run(True, False)),
                    ("F vs T (streams)", run(False, True), run(True, True))):
    bad = [n for n in a if not torch.equal(a[n], b[n])]
    print(label, len(bad), "differ:", [(n, float((a[n].float() - b[n].float()).abs().max())) for n in bad[:8]])
