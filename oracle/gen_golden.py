"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/*.npz from the REFERENCE itself.

Run in the build container (where /root/reference exists):

    python oracle/gen_golden.py [--only NAME ...]

Each fixture holds inputs and the reference's outputs for one hot-path operator
 at a reduced size, plus end-to-end `Trainer.process_batch`
results with synthetic weights (`oracle/synth.py`).  Fixtures are data only; no
reference source is stored.  Seeds: images Generator(1234); weights crc32(key);
torch.manual_seed / random.seed stated per fixture.
"""
import argparse
import os
import random
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh  # noqa: E402
from oracle import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _np(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (_np(v) if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return scale * torch.randn(*shape, generator=g)


# ---------------------------------------------------------------------------
def gen_layers():
    from ppeadepth import layers as L
    B, H, W = 2, 24, 40
    disp = torch.sigmoid(rnd(B, 1, H, W, seed=1))
    sd, depth = L.disp_to_depth(disp, 0.1, 100.0)
    aa = rnd(B, 1, 3, seed=2, scale=0.05)
    tt = rnd(B, 1, 3, seed=3, scale=0.1)
    T_fwd = L.transformation_from_parameters(aa, tt, invert=False)
    T_inv = L.transformation_from_parameters(aa, tt, invert=True)
    K, inv_K = synth.kitti_K(H, W, 0)
    K = K[None].repeat(B, 1, 1)
    inv_K = inv_K[None].repeat(B, 1, 1)
    bp, pj = L.BackprojectDepth(B, H, W), L.Project3D(B, H, W)
    pts = bp(depth, inv_K)
    grid = pj(pts, K, T_inv)
    src = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(4))
    tgt = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(5))
    warped = torch.nn.functional.grid_sample(src, grid, padding_mode="border", align_corners=True)
    # a grid that leaves the image on all sides exercises the border clamp
    wide = grid * 1.6 + 0.1
    warped_wide = torch.nn.functional.grid_sample(src, wide, padding_mode="border", align_corners=True)
    warped_zeros = torch.nn.functional.grid_sample(src, wide, padding_mode="zeros", mode="bilinear",
                                                  align_corners=True)
    ssim = L.SSIM()(warped, tgt)
    smooth = L.get_smooth_loss(disp, tgt)
    save("layers_geometry", disp=disp, scaled_disp=sd, depth=depth, axisangle=aa, translation=tt,
         T_fwd=T_fwd, T_inv=T_inv, K=K, inv_K=inv_K, points=pts, grid=grid, src=src, tgt=tgt,
         warped=warped, wide_grid=wide, warped_wide=warped_wide, warped_zeros=warped_zeros,
         ssim=ssim, smooth=smooth)


def gen_losses():
    """Trainer.compute_reprojection_loss / compute_loss_masks / compute_matching_mask and a
    whole compute_losses call (mono and multi branch) on hand-made `outputs`."""
    from ppeadepth.trainer import Trainer
    from ppeadepth import layers as L
    B, H, W = 2, 24, 40
    opt = rh.parse_options(["--height", str(H), "--width", str(W), "--batch_size", str(B)])
    tr = Trainer.__new__(Trainer)
    tr.opt, tr.ssim, tr.device, tr.step, tr.is_main = opt, L.SSIM(), torch.device("cpu"), 0, True
    g = torch.Generator().manual_seed(11)
    tgt = torch.rand(B, 3, H, W, generator=g)
    pm1 = torch.rand(B, 3, H, W, generator=g)
    pp1 = torch.rand(B, 3, H, W, generator=g)
    # black holes in the warped frames exercise selec_reproj (trainer.py:1077-1083)
    pm1[:, :, 2:6, 3:9] = 0.0
    pp1[:, :, 4:8, 6:12] = 0.0
    sm1 = torch.rand(B, 3, H, W, generator=g)
    sp1 = torch.rand(B, 3, H, W, generator=g)
    disp = torch.sigmoid(torch.randn(B, 1, H, W, generator=g))
    mono_depth = 1.0 + 5 * torch.rand(B, 1, H, W, generator=g)
    depth = 1.0 + 5 * torch.rand(B, 1, H, W, generator=g)
    cmask = (torch.rand(B, H, W, generator=g) > 0.3).float()
    aug = torch.tensor([0.0, 1.0]).reshape(B, 1, 1, 1)
    rl = tr.compute_reprojection_loss(pm1, tgt)
    a = torch.rand(B, 1, H, W, generator=g)
    b = a.clone()
    b[:, :, ::2] += 0.1        # exact ties on odd rows -> first index must win
    mask = Trainer.compute_loss_masks(a, b)
    lowest = 1.0 / (0.5 + 8 * torch.rand(B, H, W, generator=g))
    mm = tr.compute_matching_mask({("mono_depth", 0, 0): mono_depth, "lowest_cost": lowest})
    inputs = {("color", 0, 0): tgt, ("color", -1, 0): sm1, ("color", 1, 0): sp1}
    res = {}
    for tag, multi in (("mono", False), ("multi", True)):
        outputs = {("disp", 0): disp.clone(), ("color", -1, 0): pm1, ("color", 1, 0): pp1,
                   ("depth", 0, 0): depth, ("mono_depth", 0, 0): mono_depth,
                   "consistency_mask": cmask, "augmentation_mask": aug}
        torch.manual_seed(7)
        noise = torch.randn(B, 1, H, W)          # what compute_losses will draw (trainer.py:1086)
        torch.manual_seed(7)
        losses, _ = tr.compute_losses(inputs, outputs, is_multi=multi)
        for k, v in losses.items():
            res[f"{tag}:{k}"] = v
        res[f"{tag}:noise"] = noise
        if multi:
            res["multi:consistency_target"] = outputs["consistency_target/0"]
    save("losses", tgt=tgt, pred_m1=pm1, pred_p1=pp1, src_m1=sm1, src_p1=sp1, disp=disp,
         mono_depth=mono_depth, depth=depth, consistency_mask=cmask, augmentation_mask=aug,
         reproj_m1=rl, tie_a=a, tie_b=b, tie_mask=mask, lowest=lowest, matching_mask=mm.float(),
         **res)


def gen_replk_blocks():
    """ReparamLargeKernelConv / B_Adapter / Adapter / RepLKBlock / ConvFFN (train mode,
    local batch statistics) at reduced channel counts."""
    from ppeadepth.networks import replknet_adapter as rka
    out = {}
    for k, (h, w) in ((31, (12, 40)), (29, (24, 20)), (27, (12, 40)), (13, (6, 20))):
        C, B = 8, 2
        torch.manual_seed(k)
        m = rka.ReparamLargeKernelConv(C, C, k, 1, C, small_kernel=5).train()
        synth.fill_state_dict(m)
        x = rnd(B, C, h, w, seed=100 + k)
        y = m(x)
        y_big = m.lkb_origin.conv(x)
        y_small = m.small_conv.conv(x)
        out[f"lk{k}:x"] = x
        out[f"lk{k}:y"] = y
        out[f"lk{k}:y_big_conv"] = y_big
        out[f"lk{k}:y_small_conv"] = y_small
        for n, v in m.state_dict().items():
            out[f"lk{k}:sd:{n}"] = v
    C, B, h, w = 16, 2, 12, 20
    x = rnd(B, C, h, w, seed=200)
    ba = rka.B_Adapter(C, adpt_test=4, mlp_ratio=0.25)
    synth.fill_state_dict(ba)
    out["badpt:x"], out["badpt:y"] = x, ba(x)
    for n, v in ba.state_dict().items():
        out[f"badpt:sd:{n}"] = v
    ad = rka.Adapter(C, adpt_test=4, mlp_ratio=0.25)
    synth.fill_state_dict(ad)
    out["adpt:y"] = ad(x)
    for n, v in ad.state_dict().items():
        out[f"adpt:sd:{n}"] = v
    blk = rka.RepLKBlock(C, C, 13, 5, drop_path=0.0, adpt_test=4, ratio=0.25).train()
    synth.fill_state_dict(blk)
    for n, v in blk.state_dict().items():
        out[f"blk:sd:{n}"] = v.clone()
    out["blk:y"] = blk(x)
    # running statistics after one training forward (momentum 0.1, unbiased var)
    for n, v in blk.state_dict().items():
        if "running" in n:
            out[f"blk:after:{n}"] = v.clone()
    ffn = rka.ConvFFN(C, 4 * C, C, drop_path=0.0, adpt_test=4).train()
    synth.fill_state_dict(ffn)
    sd0 = {n: v.clone() for n, v in ffn.state_dict().items()}
    out["ffn:y"] = ffn(x)
    for n, v in sd0.items():
        out[f"ffn:sd:{n}"] = v
    save("replk_blocks", **out)


def gen_cost_volume():
    from ppeadepth.networks.replk_matching_adapter import RepLKMatchingAdapter as RMA
    from ppeadepth import layers as L
    B, C, h, w, D = 3, 16, 16, 24, 96
    me = types.SimpleNamespace(num_depth_bins=D, matching_height=h, matching_width=w,
                               depth_binning="log", device=torch.device("cpu"),
                               set_missing_to_max=True,
                               backprojector=L.BackprojectDepth(D, h, w),
                               projector=L.Project3D(D, h, w))
    mn, mx = torch.Tensor([0.37]), torch.Tensor([14.5])
    RMA.compute_depth_bins(me, mn, mx)
    cur = rnd(B, C, h, w, seed=31).relu()
    look = rnd(B, 1, C, h, w, seed=32).relu()
    from ppeadepth.layers import transformation_from_parameters
    T = transformation_from_parameters(rnd(B, 1, 3, seed=33, scale=0.02),
                                       rnd(B, 1, 3, seed=34, scale=0.3), invert=True)
    T[2] *= 0          # third item: zeroed pose -> frame skipped (rkm.py:294)
    poses = T[:, None]
    K, inv_K = synth.kitti_K(4 * h, 4 * w, 2)
    K, inv_K = K[None].repeat(B, 1, 1), inv_K[None].repeat(B, 1, 1)
    cost, miss = RMA.match_features(me, cur, look, poses, K, inv_K)
    conf = RMA.compute_confidence_mask(me, cost * (1 - miss))
    viz = cost.clone()
    viz[viz == 0] = 100
    mins, argmin = torch.min(viz, 1)
    lowest = RMA.indices_to_disparity(me, argmin)
    save("cost_volume", min_depth=mn, max_depth=mx, bins=me.depth_bins, cur=cur, lookup=look,
         poses=poses, K=K, inv_K=inv_K, cost=cost, missing=miss, confidence=conf,
         argmin=argmin, lowest_cost=lowest)


def gen_depth_bins():
    from ppeadepth.trainer import DepthBins
    tr = DepthBins(0.1)
    seq, mins, maxs = [], [], []
    for i in range(3):
        d = 0.05 + 20 * torch.rand(2, 1, 8, 12, generator=torch.Generator().manual_seed(40 + i))
        seq.append(d)
        tr.update(d)
        mn, mx = tr.compute()
        mins.append(mn.reshape(()))
        maxs.append(mx.reshape(()))
    save("depth_bins", depths=torch.stack(seq), mins=torch.stack(mins), maxs=torch.stack(maxs))


# ---------------------------------------------------------------------------
GRAD_KEYS = [
    "encoder.replk.stages.0.blocks.0.adapter.D_fc1.weight",
    "encoder.replk.stages.0.blocks.1.mlp_adapter.D_fc2.weight",
    "encoder.replk.stages.2.blocks.10.adapter.D_fc2.weight",
    "encoder.replk.stages.3.blocks.3.preffn_bn.weight",
    "encoder.replk.stem.0.bn.bias",
    "encoder.reduce_conv.0.weight",
    "mono_encoder.stages.1.blocks.2.adapter.D_fc1.weight",
    "mono_encoder.stages.2.blocks.35.mlp_adapter.D_fc1.bias",
    "mono_encoder.stages.0.blocks.0.large_kernel.lkb_origin.bn.weight",
    "depth.upconvs_0.0.conv.conv.weight",
    "depth.disp_convs.0.conv.weight",
    "mono_depth.upconvs_1.4.conv.conv.weight",
    "pose_encoder.encoder.conv1.weight",
    "pose_encoder.encoder.layer4.1.bn2.weight",
    "pose.net.3.weight",
]


DC_GRAD_KEYS = [k for k in GRAD_KEYS if not k.startswith(("depth.", "mono_depth."))] + [
    "depth.adapter.D_fc1.weight", "depth.adapter.D_fc2.weight", "depth.adapter.D_fc2.bias",
    "depth.deconv_adpt.weight", "depth.deconv_adpt.bias", "mono_depth.adapter.D_fc1.bias",
    "mono_depth.deconv_adpt.weight"]


TI_GRAD_KEYS = GRAD_KEYS + ["encoder.replk.input_adapter.D_fc1.weight", "encoder.replk.input_adapter.bn2.weight",
                            "encoder.replk.trans_adpt.0.D_fc1.weight", "encoder.replk.trans_adpt.2.D_fc2.bias",
                            "mono_encoder.input_adapter.D_fc2.weight", "mono_encoder.trans_adpt.1.D_fc2.weight"]


def gen_e2e(name, B, H, W, extra=(), stride=1, seed=1, grad_keys=None, intrinsics="kitti", conditioned=False):
    """Unmodified Trainer.process_batch + backward of the reference (config-1 style).
    conditioned: rendered frames (synth.make_rendered_inputs) + the well-conditioned weight variant
    (synth.synth_tensor(..., conditioned=True)) -- the fixture the bf16 step is pinned on."""
    opt = rh.parse_options(["--height", str(H), "--width", str(W), "--batch_size", str(B)]
                           + list(extra))
    torch.manual_seed(0)
    random.seed(0)
    with rh.scratch_cwd():
        from ppeadepth import networks
        model = networks.RepDepth(opt)
    if opt.dc:
        model.dc_ft_init()                 # Trainer.__init__, trainer.py:158-161 (Stage-2: decoder adapter only)
    model.train()
    synth.fill_state_dict(model, conditioned=conditioned)
    tr = rh.build_reference_trainer(opt, model)
    inputs = (synth.make_rendered_inputs(B, H, W, intrinsics=intrinsics) if conditioned
              else synth.make_inputs(B, H, W, intrinsics=intrinsics))
    torch.manual_seed(seed)
    random.seed(seed)
    outputs, losses = tr.process_batch(inputs, True)
    losses["loss"].backward()
    arrays = {"meta": np.array([B, H, W, stride, seed])}
    for k, v in losses.items():
        arrays["loss:" + k] = v
    for k, v in outputs.items():
        key = "out:" + ("|".join(str(s) for s in k) if isinstance(k, tuple) else k)
        if v.dim() >= 3 and stride > 1:
            if v.dim() == 4 and v.shape[-1] == 2:          # sample grids [B,H,W,2]
                v = v[:, ::stride, ::stride]
            else:
                v = v[..., ::stride, ::stride]
        arrays[key] = v
    arrays["in:relative_pose|-1"] = inputs[("relative_pose", -1)]
    params = dict(model.named_parameters())
    for k in (grad_keys or GRAD_KEYS):
        g = params[k].grad
        arrays["grad_sum:" + k] = g.double().sum().float()
        arrays["grad_abs:" + k] = g.double().abs().sum().float()
        arrays["grad_head:" + k] = g.reshape(-1)[:32].clone()
        # up to 4096 evenly strided elements: L2 / cosine comparisons for reduced-precision runs
        flat = g.reshape(-1)
        arrays["grad_sample:" + k] = flat[::max(1, flat.numel() // 4096)][:4096].clone()
    sd = model.state_dict()
    for k in ("encoder.replk.stem.0.bn.running_mean", "encoder.replk.stem.0.bn.running_var",
              "mono_encoder.stages.3.blocks.3.pw2.bn.running_var",
              "pose_encoder.encoder.bn1.running_mean"):
        arrays["buf:" + k] = sd[k]
    mn, mx = tr.depth_bin_tracker.compute()
    arrays["bins_after"] = torch.stack([mn.reshape(()), mx.reshape(())])
    save(name, **arrays)


def gen_dec_designs():
    """The reference's DepthDecoderV2 with every Stage-2 decoder-adapter design (`--dec_id` 1-4, 8, 10;
    depth_decoder_v2.py:135-245) standalone on seeded RepLKNet-31B-shaped features: disparity, the gradients of the four
    feature maps and every parameter gradient's (sum, abs sum, 32-element head) for the back-propagated seed."""
    from ppeadepth.networks.depth_decoder_v2 import DepthDecoderV2
    ch = np.array([128, 256, 512, 1024])
    B, h, w = 2, 16, 24
    arrays = {"meta": np.array([B, h, w])}
    for tid in (1, 2, 3, 4, 8, 10):
        torch.manual_seed(0)
        dec = DepthDecoderV2(ch, range(4), False, dc=True, test_id=tid)
        synth.fill_state_dict(dec)
        dec.train()
        feats = [rnd(B, int(ch[i]), h >> i, w >> i, seed=20 + i).requires_grad_(True) for i in range(4)]
        disp = dec(feats)[("disp", 0)]
        go = rnd(*disp.shape, seed=31)
        disp.backward(go)
        arrays[f"{tid}:disp"] = disp[..., ::2, ::2]
        arrays[f"{tid}:names"] = np.array(list(dec.state_dict().keys()))
        for i, f in enumerate(feats):                  # (spatially subsampled where large: fixtures stay small)
            arrays[f"{tid}:dfeat{i}"] = f.grad[..., ::4, ::4] if i == 0 else (f.grad[..., ::2, ::2] if i == 1 else f.grad)
        for k, p_ in dec.named_parameters():
            g = p_.grad
            arrays[f"{tid}:gsum:{k}"] = g.double().sum().float()
            arrays[f"{tid}:gabs:{k}"] = g.double().abs().sum().float()
            arrays[f"{tid}:ghead:{k}"] = g.reshape(-1)[:32].clone()
    save("dec_designs", **arrays)


def gen_eval():
    """(1) evaluate_depth.compute_errors on fixed arrays; (2) the reference's unmodified `Trainer.val`
    (trainer.py:653-857) on a synthetic eigen_zhou-format split (synth.make_eval_split), synthetic weights, eval
    mode: mean errors of the multi-frame network and of the teacher + the predicted disparities."""
    import tempfile
    from ppeadepth.evaluate_depth import compute_errors
    g = torch.Generator().manual_seed(3)
    gt = (1.0 + 60 * torch.rand(5000, generator=g)).numpy().astype(np.float64)
    pred = (gt * (0.7 + 0.6 * torch.rand(5000, generator=g).numpy())).astype(np.float64)
    arrays = {"ce_gt": gt, "ce_pred": pred, "ce_errors": np.array(compute_errors(gt, pred))}
    n, H, W = 3, 192, 640
    opt = rh.parse_options(["--height", str(H), "--width", str(W), "--batch_size", str(n), "--eval_split", "eigen"])
    torch.manual_seed(0)
    with rh.scratch_cwd():
        from ppeadepth import networks
        model = networks.RepDepth(opt)
    synth.fill_state_dict(model)
    model.eval()
    tr = rh.build_reference_trainer(opt, model)
    tr.model = types.SimpleNamespace(module=model)
    tr.val_frames_to_load = [0, -1]
    root = tempfile.mkdtemp(prefix="ppea_eval_")
    synth.make_eval_split(root, n=n, height=H, width=W, seed=7, split="eigen")
    ds = synth.SynthEigenDataset(root, split="eigen", height=H, width=W)
    tr.val_loader = [synth.collate([ds[i] for i in range(n)])]
    old = os.getcwd()
    os.chdir(root)
    try:
        errors, errors_mono = tr.val()
    finally:
        os.chdir(old)
    arrays.update(val_meta=np.array([n, H, W, 7]), val_errors=np.asarray(errors), val_errors_mono=np.asarray(errors_mono))
    save("eval", **arrays)


def gen_state_spec():
    """Key names / shapes / trainable flags of RepDepth(opt).state_dict() (31B and 31L)."""
    arrays = {}
    for size in ("b", "l"):
        opt = rh.parse_options(["--rep_size", size])
        with rh.scratch_cwd():
            from ppeadepth import networks
            model = networks.RepDepth(opt)
        sd = model.state_dict()
        req = {n: p.requires_grad for n, p in model.named_parameters()}
        arrays[f"{size}:names"] = np.array(list(sd.keys()))
        arrays[f"{size}:shapes"] = np.array([";".join(str(d) for d in v.shape) for v in sd.values()])
        arrays[f"{size}:trainable"] = np.array([int(req.get(n, False)) for n in sd.keys()])
        arrays[f"{size}:bn_types"] = np.array(sorted({type(m).__name__ for m in model.modules()
                                                      if "BatchNorm" in type(m).__name__}))
        del model
    save("state_spec", **arrays)


GENERATORS = {
    "state_spec": gen_state_spec,
    "eval": gen_eval,
    "layers_geometry": gen_layers,
    "losses": gen_losses,
    "replk_blocks": gen_replk_blocks,
    "cost_volume": gen_cost_volume,
    "depth_bins": gen_depth_bins,
    "e2e_small": lambda: gen_e2e("e2e_small", 2, 64, 96),
    "e2e_small_ckpt": lambda: gen_e2e("e2e_small_ckpt", 2, 64, 96, extra=["--use_checkpoint"]),
    "e2e_full": lambda: gen_e2e("e2e_full", 2, 192, 640, stride=8),
    # BASELINE config 4's model (RepLKNet-31L: 1.5x channels) at a reduced frame size
    "e2e_l": lambda: gen_e2e("e2e_l", 2, 64, 96, extra=["--rep_size", "l"], stride=2),
    # BASELINE config 5 (Stage-2 `--dc`: decoder adapter trained, decoder frozen) at the reference-faithful
    # Cityscapes size 192x512 (trainer.py:90-93) with Cityscapes-like intrinsics
    "e2e_dc": lambda: gen_e2e("e2e_dc", 2, 192, 512, extra=["--dc"], stride=8, grad_keys=DC_GRAD_KEYS,
                              intrinsics="cityscapes"),
    # the WELL-CONDITIONED fixture: frames rendered from a known depth map + ego-motion, near-identity
    # trunk, predicted pose = rendered pose; config 1/2's model at full size.  The bf16 step is held to absolute bounds
    # on it (tests/test_e2e_gpu.py::test_engine_step_bf16_on_the_rendered_fixture)
    "e2e_render": lambda: gen_e2e("e2e_render", 2, 192, 640, stride=8, conditioned=True),
    # the same well-conditioned construction for BASELINE config 4's model (RepLKNet-31L) at full size and for config 5
    # (Stage-2 `--dc`, 192x512, Cityscapes intrinsics): absolute bounds for the bf16 step of every benchmarked model
    # the `--trans` / `--input` adapters of both encoders (replknet_adapter.py:112-149, 429-458, 511-540)
    "e2e_ti": lambda: gen_e2e("e2e_ti", 2, 64, 96, extra=["--trans", "--input", "--mono_trans", "--mono_input"],
                              grad_keys=TI_GRAD_KEYS),
    "dec_designs": gen_dec_designs,
    "e2e_render_l": lambda: gen_e2e("e2e_render_l", 2, 192, 640, extra=["--rep_size", "l"], stride=8, conditioned=True),
    "e2e_render_dc": lambda: gen_e2e("e2e_render_dc", 2, 192, 512, extra=["--dc"], stride=8, grad_keys=DC_GRAD_KEYS,
                                     intrinsics="cityscapes", conditioned=True),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    if not rh.reference_available():
        raise SystemExit("reference tree not present: goldens can only be regenerated in the build container")
    rh.install_stubs()
    torch.set_num_threads(8)
    for name, fn in GENERATORS.items():
        if args.only and name not in args.only:
            continue
        print(name)
        fn()


if __name__ == "__main__":
    main()
