#!/usr/bin/env python3
"""Scratch diagnostics (GPU): (1) fp32 RepLKBlock dx at C=64,K=13,6x20 vs CPU oracle in fp64; (2) eager vs graph loss
sequences."""
import os
import random
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ppea-depth_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from conftest import rel_err  # noqa: E402


def l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def blk(dev):
    from oracle import ref_model as RM, ref_ops as R, synth
    from ppeadepth import ops
    from ppeadepth.networks import replknet_adapter as rka
    for (C, K, H, W, B) in ((64, 13, 6, 20, 3), (64, 13, 6, 20, 2), (32, 13, 6, 20, 3), (64, 13, 12, 20, 3)):
        x = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(C + H))
        go = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(7))
        # (a) the conv alone
        wb = torch.randn(C, 1, K, K, generator=torch.Generator().manual_seed(1)) / K
        ws = torch.randn(C, 1, 5, 5, generator=torch.Generator().manual_seed(2)) / 5
        xd = x.to(dev).requires_grad_(True)
        yb, ys = ops.dwconv_lk(xd, wb.to(dev), ws.to(dev))
        (yb * go.to(dev) + ys * go.to(dev)).sum().backward()
        xr = x.double().requires_grad_(True)
        (R.dwconv(xr, wb.double()) * go.double() + R.dwconv(xr, ws.double()) * go.double()).sum().backward()
        print(f"conv C={C} K={K} {H}x{W} B={B}: y {rel_err(yb.cpu(), R.dwconv(x, wb)):.2e} dx {rel_err(xd.grad.cpu(), xr.grad):.2e}")
        # (b) the block, oracle in fp64 and fp32
        m = rka.RepLKBlock(C, C, K, 5, drop_path=0.0, adpt_test=4, ratio=0.25)
        synth.fill_state_dict(m)
        opt = types.SimpleNamespace(rep_size="b", g_blk=1.0, g_ffn=1.0, use_checkpoint=False)
        res = {}
        for dt in (torch.float64, torch.float32):
            sd = {"m." + k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in m.state_dict().items()}
            ref = RM.RefRepDepth(sd, opt)
            xr = x.clone().to(dt).requires_grad_(True)
            yr = ref._replk_block(xr, "m", K, 0.0)
            yr.backward(go.to(dt))
            res[dt] = (yr.detach(), xr.grad)
        print(f"   oracle fp32 vs fp64: y {rel_err(res[torch.float32][0], res[torch.float64][0]):.2e} "
              f"dx {rel_err(res[torch.float32][1], res[torch.float64][1]):.2e} l2 {l2(res[torch.float32][1], res[torch.float64][1]):.2e}")
        md = m.to(dev).train()
        for n, p in md.named_parameters():
            p.requires_grad = "adapter" in n or ".bn" in n or "_bn" in n
        xd = x.to(dev).requires_grad_(True)
        y = md(xd)
        y.backward(go.to(dev))
        print(f"   product fp32 vs fp64: y {rel_err(y.cpu(), res[torch.float64][0]):.2e} dx {rel_err(xd.grad.cpu(), res[torch.float64][1]):.2e} "
              f"l2 {l2(xd.grad.cpu(), res[torch.float64][1]):.2e}")
        # the oracle's functions on the GPU (library kernels): third opinion
        sdg = {"m." + k: v.clone().to(dev) for k, v in m.state_dict().items()}
        refg = RM.RefRepDepth(sdg, opt)
        xg = x.to(dev).requires_grad_(True)
        yg = refg._replk_block(xg, "m", K, 0.0)
        yg.backward(go.to(dev))
        print(f"   torch-GPU fp32 vs fp64: y {rel_err(yg.cpu(), res[torch.float64][0]):.2e} dx {rel_err(xg.grad.cpu(), res[torch.float64][1]):.2e}")


def seq(dev):
    import test_e2e_gpu as T
    from oracle import synth
    from ppeadepth import rng
    from ppeadepth.dist import TrainEngine
    B, H, W, N = 2, 64, 96, 4
    batches = [{k: v.to(dev) for k, v in synth.make_inputs(B, H, W, seed=50 + i, smooth=True).items()} for i in range(N)]
    for graph in (False, True):
        opt, model, tr = T._build(dev, B, H, W, use_checkpoint=True, amp=None)
        rng.set_mode("reference")
        eng = TrainEngine(tr, lr=1e-4, bf16_params=False)
        if graph:
            eng.capture(batches[0], warmup=1, restore_state=True)
        out = []
        for i in range(N):
            random.seed(100 + i)
            torch.manual_seed(100 + i)
            o, losses = eng.step(dict(batches[i]) if not graph else batches[i])
            out.append((float(losses["loss"]), float(losses["reproj_loss/0"]), float(losses["consistency_loss/0"]),
                        float(tr.depth_bin_tracker.max_depth), float(o["augmentation_mask"].sum()),
                        float(o[("disp", 0)].float().sum()), float(o[("mono_disp", 0)].float().sum())))
        print("graph" if graph else "eager")
        for r in out:
            print("   ", r)
        rng.set_aug_buffer(None)


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    what = sys.argv[1:] or ["blk", "seq"]
    if "blk" in what:
        blk(dev)
    if "seq" in what:
        seq(dev)
