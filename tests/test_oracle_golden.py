"""Pins the CPU oracle (oracle/ref_ops.py, oracle/ref_model.py) to golden vectors
produced by the reference implementation itself (oracle/gen_golden.py).

Tolerances: fp32 outputs <= 1e-5 relative, index tensors bit-exact.
"""
import os
import random
import types

import pytest
import torch

from conftest import rel_err
from oracle import ref_ops as R
from oracle import ref_model as RM
from oracle import synth

TOL = 1e-5


def test_disp_to_depth_and_pose(golden):
    g = golden("layers_geometry")
    sd, depth = R.disp_to_depth(g["disp"], 0.1, 100.0)
    assert rel_err(sd, g["scaled_disp"]) < TOL
    assert rel_err(depth, g["depth"]) < TOL
    assert rel_err(R.transformation_from_parameters(g["axisangle"], g["translation"], False), g["T_fwd"]) < TOL
    assert rel_err(R.transformation_from_parameters(g["axisangle"], g["translation"], True), g["T_inv"]) < TOL
    # invert=True composed with invert=False is the identity (layers.py:299-307 smoke)
    eye = g["T_inv"] @ g["T_fwd"]
    assert rel_err(eye, torch.eye(4).expand_as(eye)) < 1e-5


def test_backproject_project_warp(golden):
    g = golden("layers_geometry")
    B, _, H, W = g["depth"].shape
    pts = R.backproject(g["depth"], g["inv_K"])
    assert pts.shape == g["points"].shape
    assert rel_err(pts, g["points"]) < TOL
    grid = R.project3d(pts, g["K"], g["T_inv"], H, W)
    assert (grid - g["grid"]).abs().max() < 1e-5
    assert rel_err(R.grid_sample_border(g["src"], g["grid"]), g["warped"]) < TOL
    assert rel_err(R.grid_sample_border(g["src"], g["wide_grid"]), g["warped_wide"]) < TOL
    assert rel_err(R.grid_sample_zeros(g["src"], g["wide_grid"]), g["warped_zeros"]) < TOL
    # the gather-free restatement of grid_sampler agrees with both padding modes
    assert rel_err(R.grid_sample_manual(g["src"], g["wide_grid"], True), g["warped_wide"]) < TOL
    assert rel_err(R.grid_sample_manual(g["src"], g["wide_grid"], False), g["warped_zeros"]) < TOL


def test_ssim_smooth(golden):
    g = golden("layers_geometry")
    assert rel_err(R.ssim(g["warped"], g["tgt"]), g["ssim"]) < TOL
    assert rel_err(R.smooth_loss(g["disp"], g["tgt"]), g["smooth"]) < TOL


def test_loss_pieces(golden):
    g = golden("losses")
    assert rel_err(R.reprojection_loss(g["pred_m1"], g["tgt"]), g["reproj_m1"]) < TOL
    idx, mask = R.automask(g["tie_a"], g["tie_b"])
    assert torch.equal(mask, g["tie_mask"])                     # ties -> index 0 (first minimum)
    assert idx.dtype == torch.int64
    mm = R.matching_mask(g["lowest"], g["mono_depth"])
    assert torch.equal(mm.float(), g["matching_mask"])


@pytest.mark.parametrize("multi", [False, True])
def test_compute_losses(golden, multi):
    g = golden("losses")
    tag = "multi" if multi else "mono"
    opt = types.SimpleNamespace(height=24, width=40, min_depth=0.1, max_depth=100.0,
                                disparity_smoothness=1e-3)
    tr = RM.RefTrainer(None, opt)
    inputs = {("color", 0, 0): g["tgt"], ("color", -1, 0): g["src_m1"], ("color", 1, 0): g["src_p1"]}
    outputs = {("disp", 0): g["disp"], ("color", -1, 0): g["pred_m1"], ("color", 1, 0): g["pred_p1"],
               ("depth", 0, 0): g["depth"], ("mono_depth", 0, 0): g["mono_depth"],
               "consistency_mask": g["consistency_mask"], "augmentation_mask": g["augmentation_mask"]}
    torch.manual_seed(7)
    losses = tr.compute_losses(inputs, outputs, multi)
    for k, v in losses.items():
        assert rel_err(v, g[f"{tag}:{k}"]) < TOL, k
    if multi:
        assert rel_err(outputs["consistency_target/0"], g["multi:consistency_target"]) < TOL


@pytest.mark.parametrize("k", [31, 29, 27, 13])
def test_reparam_large_kernel(golden, k):
    g = golden("replk_blocks")
    x = g[f"lk{k}:x"]
    sd = {n[len(f"lk{k}:sd:"):]: v for n, v in g.items() if n.startswith(f"lk{k}:sd:")}
    wb, ws = sd["lkb_origin.conv.weight"], sd["small_conv.conv.weight"]
    assert rel_err(R.dwconv(x, wb), g[f"lk{k}:y_big_conv"]) < TOL
    assert rel_err(R.dwconv(x, ws), g[f"lk{k}:y_small_conv"]) < TOL
    # tap-by-tap restatement (no conv primitive) agrees with the reference conv
    assert rel_err(R.dwconv_loops(x, wb), g[f"lk{k}:y_big_conv"]) < TOL
    y = R.reparam_lk(x, wb, sd["lkb_origin.bn.weight"], sd["lkb_origin.bn.bias"],
                     ws, sd["small_conv.bn.weight"], sd["small_conv.bn.bias"])
    assert rel_err(y, g[f"lk{k}:y"]) < TOL


def _sub(g, prefix):
    return {n[len(prefix):]: v.clone() for n, v in g.items() if n.startswith(prefix)}


def test_adapters_and_blocks(golden):
    g = golden("replk_blocks")
    x = g["badpt:x"]
    sd = _sub(g, "badpt:sd:")
    y = R.b_adapter(x, sd["D_fc1.weight"], sd["D_fc1.bias"], sd["D_fc2.weight"], sd["D_fc2.bias"])
    assert rel_err(y, g["badpt:y"]) < TOL
    sd = _sub(g, "adpt:sd:")
    y = R.mlp_adapter(x, sd["D_fc1.weight"], sd["D_fc1.bias"], sd["D_fc2.weight"], sd["D_fc2.bias"])
    assert rel_err(y, g["adpt:y"]) < TOL
    opt = types.SimpleNamespace(rep_size="b", g_blk=1.0, g_ffn=1.0, use_checkpoint=False)
    m = RM.RefRepDepth({"blk." + k: v for k, v in _sub(g, "blk:sd:").items()}, opt)
    assert rel_err(m._replk_block(x, "blk", 13, 0.0), g["blk:y"]) < TOL
    for n, v in _sub(g, "blk:after:").items():
        assert rel_err(m.sd["blk." + n], v) < TOL, n        # running stats: momentum .1, unbiased var
    m = RM.RefRepDepth({"ffn." + k: v for k, v in _sub(g, "ffn:sd:").items()}, opt)
    assert rel_err(m._conv_ffn(x, "ffn", 0.0), g["ffn:y"]) < TOL


def test_bn_train_explicit_stats():
    x = torch.randn(3, 5, 7, 9)
    gmm, bta = torch.rand(5) + 0.5, torch.randn(5)
    rm, rv = torch.zeros(5), torch.ones(5)
    y_ref = torch.nn.functional.batch_norm(x, rm, rv, gmm, bta, True, 0.1, 1e-5)
    y, mean, var = R.bn_train(x, gmm, bta)
    assert rel_err(y, y_ref) < TOL
    m2, v2 = R.bn_running_update(torch.zeros(5), torch.ones(5), mean, var, 3 * 7 * 9)
    assert rel_err(m2, rm) < TOL and rel_err(v2, rv) < TOL


def test_cost_volume(golden):
    g = golden("cost_volume")
    bins = R.depth_bins_log(g["min_depth"], g["max_depth"], 96)
    assert rel_err(bins, g["bins"]) < 1e-6
    cost, miss = R.cost_volume(g["cur"], g["lookup"], g["poses"], g["K"], g["inv_K"], bins)
    assert rel_err(cost, g["cost"]) < TOL
    assert torch.equal(miss, g["missing"])
    conf, idx, lowest, _ = R.cost_volume_reduce(cost, miss, bins)
    assert torch.equal(conf, g["confidence"])
    assert torch.equal(idx, g["argmin"])                       # int64, bit-exact
    assert rel_err(lowest, g["lowest_cost"]) < TOL
    assert float(cost[2].abs().max()) == 0.0                   # zeroed pose -> empty volume


def test_depth_bin_tracker(golden):
    g = golden("depth_bins")
    tr = R.DepthBinTracker(0.1)
    for i in range(g["depths"].shape[0]):
        tr.update(g["depths"][i])
        mn, mx = tr.compute()
        assert rel_err(mn, g["mins"][i]) < 1e-6 and rel_err(mx, g["maxs"][i]) < 1e-6


# ---------------------------------------------------------------------------
def _run_e2e(g, use_checkpoint=False, rep_size="b", dc=False, intrinsics="kitti", conditioned=False):
    B, H, W, stride, seed = (int(v) for v in g["meta"])
    opt = types.SimpleNamespace(
        rep_size=rep_size, g_blk=1.0, g_ffn=1.0, use_checkpoint=use_checkpoint, height=H, width=W,
        batch_size=B, num_depth_bins=96, min_depth=0.1, max_depth=100.0, disparity_smoothness=1e-3, dc=dc)
    from oracle import model_spec
    sd = {k: synth.synth_tensor(k, torch.empty(shape, dtype=dt), conditioned)
          for k, (shape, dt) in model_spec.state_spec(rep_size, dc=dc).items()}
    sd = RM.leaf_state_dict(sd, opt)
    model = RM.RefRepDepth(sd, opt)
    tr = RM.RefTrainer(model, opt)
    inputs = (synth.make_rendered_inputs(B, H, W, intrinsics=intrinsics) if conditioned
              else synth.make_inputs(B, H, W, intrinsics=intrinsics))
    torch.manual_seed(seed)
    random.seed(seed)
    outputs, losses = tr.process_batch(inputs)
    losses["loss"].backward()
    return outputs, losses, sd, tr, inputs, stride


def _check_e2e(g, outputs, losses, sd, tr, inputs, stride, tol=2e-4):
    for k, v in g.items():
        if k.startswith("loss:"):
            assert rel_err(losses[k[5:]], v) < tol, k
    for k, v in g.items():
        if not k.startswith("out:"):
            continue
        parts = k[4:].split("|")
        key = tuple(int(p) if p.lstrip("-").isdigit() else p for p in parts) if len(parts) > 1 else parts[0]
        mine = outputs[key].detach()
        if mine.dim() >= 3 and stride > 1:
            mine = mine[:, ::stride, ::stride] if (mine.dim() == 4 and mine.shape[-1] == 2) \
                else mine[..., ::stride, ::stride]
        if key in ("lowest_cost", "consistency_mask", "augmentation_mask"):
            assert torch.equal(mine, v), k                    # index-derived / mask tensors: exact
        else:
            assert rel_err(mine, v) < tol, k
    assert rel_err(inputs[("relative_pose", -1)], g["in:relative_pose|-1"]) < tol
    for k, v in g.items():
        if k.startswith("grad_sum:"):
            name = k[9:]
            gr = sd[name].grad
            scale = float(g["grad_abs:" + name]) + 1e-12
            assert abs(float(gr.double().sum()) - float(v)) / scale < 5e-4, k
            assert abs(float(gr.double().abs().sum()) - scale) / scale < 5e-4, k
            assert (gr.reshape(-1)[:32] - g["grad_head:" + name]).abs().max() / \
                (g["grad_head:" + name].abs().max() + 1e-12) < 5e-3, k
        if k.startswith("buf:"):
            assert rel_err(sd[k[4:]], v) < tol, k
    mn, mx = tr.bins.compute()
    assert rel_err(torch.stack([mn.reshape(()), mx.reshape(())]), g["bins_after"]) < 1e-6


def test_e2e_small(golden):
    g = golden("e2e_small")
    _check_e2e(g, *_run_e2e(g))


def test_e2e_small_checkpoint_semantics(golden):
    """--use_checkpoint: identical losses; BN running stats updated twice inside blocks."""
    g, g0 = golden("e2e_small_ckpt"), golden("e2e_small")
    assert float(g["loss:loss"]) == float(g0["loss:loss"])
    _check_e2e(g, *_run_e2e(g, use_checkpoint=True))


@pytest.mark.slow
def test_e2e_full(golden):
    g = golden("e2e_full")
    _check_e2e(g, *_run_e2e(g))


@pytest.mark.slow
def test_e2e_rendered_well_conditioned(golden):
    """The well-conditioned fixture (rendered frames, near-identity trunk, predicted pose = rendered pose) the bf16 step
    is pinned on: the oracle reproduces the reference's unmodified process_batch on it like on the others."""
    g = golden("e2e_render")
    _check_e2e(g, *_run_e2e(g, conditioned=True))


def test_e2e_replknet31l(golden):
    """BASELINE config 4's model (RepLKNet-31L, 1.5x channels) at 64x96."""
    g = golden("e2e_l")
    _check_e2e(g, *_run_e2e(g, rep_size="l"))


@pytest.mark.slow
def test_e2e_stage2_decoder_adapter(golden):
    """BASELINE config 5: `--dc` after dc_ft_init (decoder adapter + deconv trainable, decoder frozen) at the
    reference-faithful Cityscapes size 192x512 (trainer.py:90-93)."""
    g = golden("e2e_dc")
    outputs, losses, sd, tr, inputs, stride = _run_e2e(g, dc=True, intrinsics="cityscapes")
    _check_e2e(g, outputs, losses, sd, tr, inputs, stride)
    assert sd["depth.upconvs_0.0.conv.conv.weight"].grad is None          # frozen by dc_ft_init
    assert sd["depth.deconv_adpt.weight"].grad is not None


@pytest.mark.slow
def test_e2e_rendered_replknet31l(golden):
    """The well-conditioned fixture for BASELINE config 4's model (RepLKNet-31L at 192x640): the absolute-bound pin of the
    bf16 step of that config (tests/test_e2e_gpu.py::test_engine_step_bf16_on_the_rendered_fixture[e2e_render_l])."""
    g = golden("e2e_render_l")
    _check_e2e(g, *_run_e2e(g, rep_size="l", conditioned=True))


@pytest.mark.slow
def test_e2e_rendered_stage2_decoder_adapter(golden):
    """The well-conditioned fixture for BASELINE config 5 (`--dc`, 192x512, Cityscapes intrinsics)."""
    g = golden("e2e_render_dc")
    outputs, losses, sd, tr, inputs, stride = _run_e2e(g, dc=True, intrinsics="cityscapes", conditioned=True)
    _check_e2e(g, outputs, losses, sd, tr, inputs, stride)
    assert sd["depth.upconvs_0.0.conv.conv.weight"].grad is None and sd["depth.deconv_adpt.weight"].grad is not None


@pytest.mark.parametrize("size", ["b", "l"])
def test_state_spec_matches_reference(size):
    """Names, shapes and the name-substring freeze rule (repdepth.py:47-50, 121-124)."""
    import numpy as np
    import os
    from conftest import GOLDEN
    from oracle import model_spec
    z = np.load(os.path.join(GOLDEN, "state_spec.npz"))
    names = [str(n) for n in z[f"{size}:names"]]
    shapes = {n: tuple(int(d) for d in str(s).split(";") if d) for n, s in zip(names, z[f"{size}:shapes"])}
    spec = model_spec.state_spec(size)
    assert set(spec) == set(names)
    assert len(names) == 2888
    for n in names:
        assert tuple(spec[n][0]) == shapes[n], n
    opt = types.SimpleNamespace(rep_size=size)
    for n, t in zip(names, z[f"{size}:trainable"]):
        if "running_" in n or "num_batches" in n:
            continue
        assert RM.trainable(n, opt) == bool(t), n
    assert list(z[f"{size}:bn_types"]) == ["BatchNorm2d", "SyncBatchNorm"]


# ---------------------------------------------------------------------------
# AbsRel harness
# ---------------------------------------------------------------------------
def _eval_split(tmp_path, g):
    n, H, W, seed = (int(v) for v in g["val_meta"])
    synth.make_eval_split(str(tmp_path), n=n, height=H, width=W, seed=seed, split="eigen")
    ds = synth.SynthEigenDataset(str(tmp_path), split="eigen", height=H, width=W)
    return ds, synth.collate([ds[i] for i in range(n)])


def test_compute_errors_matches_reference(golden):
    """Oracle restatement and the product's numpy `compute_errors` against the reference's own
    evaluate_depth.compute_errors on fixed arrays (tests/golden/eval.npz)."""
    g = golden("eval")
    assert rel_err(R.compute_errors(g["ce_gt"], g["ce_pred"]), g["ce_errors"]) < 1e-9
    import importlib.util
    import numpy as np
    from conftest import PKG
    spec = importlib.util.spec_from_file_location("_ppea_evaluate", os.path.join(PKG, "ppeadepth", "evaluate.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)                                   # by path: no HIP library needed for the metric code
    mine = np.array(ev.compute_errors(g["ce_gt"].numpy(), g["ce_pred"].numpy()))
    assert np.allclose(mine, g["ce_errors"].numpy(), rtol=1e-12)
    # the per-image protocol (resize, Eigen crop, median scaling, clamp): product numpy == oracle torch
    gen = torch.Generator().manual_seed(5)
    disp = 1 / (2 + 50 * torch.rand(48, 160, generator=gen))
    gt = (1 + 70 * torch.rand(375, 1242, generator=gen)) * (torch.rand(375, 1242, generator=gen) < 0.05)
    a = np.array(ev.evaluate_image(disp.numpy(), gt.numpy())[0])
    assert rel_err(torch.from_numpy(a), R.evaluate_image(disp, gt)) < 1e-5


@pytest.mark.slow
def test_val_absrel_oracle_matches_reference(golden, tmp_path):
    """The oracle's eval-mode inference path + metric protocol reproduce the reference's unmodified `Trainer.val`
    (trainer.py:653-857) on the synthetic eigen_zhou-format split: AbsRel and the other six errors, both networks."""
    g = golden("eval")
    ds, batch = _eval_split(tmp_path, g)
    opt = types.SimpleNamespace(rep_size="b", g_blk=1.0, g_ffn=1.0, use_checkpoint=False, num_depth_bins=96)
    from oracle import model_spec
    sd = {k: synth.synth_tensor(k, torch.empty(shape, dtype=dt)) for k, (shape, dt) in model_spec.state_spec("b").items()}
    model = RM.RefRepDepth(sd, opt)
    model.training = False
    disp, mono = model.predict_val(batch, torch.tensor([0.1]), torch.tensor([10.0]))
    gts = ds.gt_depths()
    err = torch.stack([R.evaluate_image(disp[i], torch.from_numpy(gts[i])) for i in range(len(gts))]).mean(0)
    err_m = torch.stack([R.evaluate_image(mono[i], torch.from_numpy(gts[i])) for i in range(len(gts))]).mean(0)
    assert abs(float(err[0]) - float(g["val_errors"][0])) < 1e-4            # AbsRel
    assert abs(float(err_m[0]) - float(g["val_errors_mono"][0])) < 1e-4
    assert rel_err(err, g["val_errors"]) < 1e-3 and rel_err(err_m, g["val_errors_mono"]) < 1e-3
