#!/usr/bin/env python3
"""Where does the bf16 step differ from the fp32 goldens, and is it this build's kernels or bf16 itself?

  block   : one RepLKBlock / ConvFFN, dx / y / adapter grads vs the fp32 oracle, with kernel families switched off one
            at a time (plain torch bf16 autocast = all off)
  e2e     : per-key errors of the bf16 engine step vs the golden, for the full kernel set and for plain torch bf16
"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ppea-depth_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from conftest import load_golden, rel_err  # noqa: E402


def switches(**kw):
    from ppeadepth import ops
    from ppeadepth.networks import replknet_adapter as rka
    rka.FUSE_BN = kw.get("fuse_bn", True)
    rka.PW_MFMA = kw.get("pw", True)
    rka.ADAPTER_MFMA = kw.get("adapter", True)
    ops._MFMA_K = (31, 29, 27, 13) if kw.get("dw", True) else ()
    rka.ADAPTER_STREAMS = kw.get("streams", True)
    ops.CONV_MFMA = kw.get("conv", True)


def block(kind, C, K, H, W, dev):
    from oracle import ref_model as RM, synth
    from ppeadepth.networks import replknet_adapter as rka
    B = 3
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, C, H, W, generator=g)
    go = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(7))

    def make():
        if kind == "blk":
            m = rka.RepLKBlock(C, C, K, 5, drop_path=0.0, adpt_test=4, ratio=0.25)
        else:
            m = rka.ConvFFN(C, 4 * C, C, drop_path=0.0, adpt_test=4)
        synth.fill_state_dict(m)
        return m
    m = make()
    sd = {"m." + k: v.clone() for k, v in m.state_dict().items()}
    opt = types.SimpleNamespace(rep_size="b", g_blk=1.0, g_ffn=1.0, use_checkpoint=False)
    ref = RM.RefRepDepth(sd, opt)
    xr = x.clone().requires_grad_(True)
    yr = ref._replk_block(xr, "m", K, 0.0) if kind == "blk" else ref._conv_ffn(xr, "m", 0.0)
    yr.backward(go)
    for name, kw in (("all kernels", {}), ("no dw mfma", dict(dw=False)), ("no pw mfma", dict(pw=False)),
                     ("no adapter mfma", dict(adapter=False)), ("no fused bn", dict(fuse_bn=False)),
                     ("no streams", dict(streams=False)),
                     ("plain torch bf16", dict(dw=False, pw=False, adapter=False, fuse_bn=False, streams=False)),
                     ("fp32 kernels", dict(fp32=True))):
        switches(**kw)
        m = make().to(dev).train()
        for n, p in m.named_parameters():
            p.requires_grad = "adapter" in n or ".bn" in n or "_bn" in n
        xd = x.to(dev).requires_grad_(True)
        if kw.get("fp32"):
            y = m(xd)
            y.backward(go.to(dev))
        else:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(xd.bfloat16())
            y.backward(go.to(dev).bfloat16())
        print(f"  {kind} C={C} K={K} {H}x{W}  {name:18s} y {rel_err(y.float().cpu(), yr.detach()):.2e}  "
              f"dx {rel_err(xd.grad.cpu(), xr.grad):.2e}", flush=True)
    switches()


def e2e(name, dev):
    import test_e2e_gpu as T
    g = load_golden(name)
    for tag, kw in (("all kernels", {}), ("plain torch bf16", dict(dw=False, pw=False, adapter=False, fuse_bn=False, conv=False))):
        switches(**kw)
        res = T._engine_step(name, lambda n: g, dev, bf16=True, graph=False, **T.CONFIG_OF.get(name, {}))
        errs = T._errors(*res)
        print(name, tag)
        for k, v in sorted(errs.items()):
            if k.split(":")[0] in ("l2", "grad_l2", "grad_cos", "loss", "share", "buf", "grad_abs"):
                print(f"    {k:70s} {v:.3e}")
    switches()


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    what = sys.argv[1:] or ["block", "e2e"]
    if "block" in what:
        for cfg in (("blk", 64, 13, 6, 20), ("blk", 128, 31, 24, 40), ("ffn", 64, 0, 12, 40)):
            block(*cfg, dev)
    if "e2e" in what:
        for n in ("e2e_small", "e2e_full", "e2e_l", "e2e_dc"):
            e2e(n, dev)
