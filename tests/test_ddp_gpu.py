"""Two ranks on ONE MI355X (gloo transport, device tensors): the data-parallel path end to end --
fused SyncBN statistics across ranks, flat-gradient mean all-reduce with bf16 working weights and
fp32 masters, depth-bin tracker reduction.  RCCL itself cannot be exercised with two ranks on one
device; the collectives are issued through the same torch.distributed calls."""
import os
import random

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

B_GLOBAL, H, W = 2, 64, 96


def _patch_rng():
    """Deterministic draws (no DropPath, no matching augmentation, no tie-break noise).  Returns the originals:
    the parent process must restore them, later tests in the same process draw through the same module."""
    from ppeadepth import rng
    saved = (rng.bernoulli_keep, rng.aug_draws, rng.randn_like_cpu_order, rng.drop_path_scales)
    rng.bernoulli_keep = lambda batch, keep, like: torch.ones(batch, 1, 1, 1, device=like.device, dtype=like.dtype)
    rng.drop_path_scales = lambda keep, batch: (1.0 / keep).expand(-1, batch)
    rng.aug_draws = lambda batch, device: torch.full((batch,), 0.9, device=device)
    rng.randn_like_cpu_order = lambda shape, device: torch.zeros(shape, device=device)
    return saved


def _restore_rng(saved):
    from ppeadepth import rng
    rng.bernoulli_keep, rng.aug_draws, rng.randn_like_cpu_order, rng.drop_path_scales = saved


def _build(batch):
    from oracle import synth
    from ppeadepth import networks, options
    from ppeadepth.trainer import Trainer
    dev = torch.device("cuda:0")
    opt = options.default_options(height=H, width=W, batch_size=batch, use_checkpoint=True)
    model = networks.RepDepth(opt)
    synth.fill_state_dict(model)
    model.to(dev).train()
    return opt, model, Trainer(opt, model, dev), dev


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "ppea-depth_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import time
    t0 = time.time()
    from oracle import synth
    from ppeadepth import dist as pdist
    pdist.init_distributed("gloo")
    _patch_rng()
    opt, model, tr, dev = _build(1)
    t1 = time.time()
    pdist.broadcast_module(model)
    t2 = time.time()
    eng = pdist.TrainEngine(tr, lr=1e-4)
    full = synth.make_inputs(B_GLOBAL, H, W, smooth=True)
    inputs = {k: v[rank:rank + 1].to(dev).contiguous() for k, v in full.items()}
    outputs, losses = eng.step(dict(inputs))
    torch.cuda.synchronize()
    print(f"[ddp worker {rank}] import+build {t1 - t0:.1f} s, broadcast {t2 - t1:.1f} s, step {time.time() - t2:.1f} s",
          flush=True)
    sd = model.state_dict()
    probe = torch.cat([sd["encoder.replk.stem.0.bn.running_var"].float().cpu(),
                       sd["mono_encoder.stages.2.blocks.7.pw2.bn.running_mean"].float().cpu(),
                       sd["encoder.replk.stages.0.blocks.1.mlp_adapter.D_fc2.weight"].float().flatten()[:64].cpu(),
                       sd["depth.upconvs_0.0.conv.conv.weight"].float().flatten()[:64].cpu()])
    q.put((rank, outputs[("disp", 0)].detach().float().cpu(), outputs[("mono_disp", 0)].detach().float().cpu(),
           probe, float(losses["loss"])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_match_global_batch(device):
    from oracle import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # single process, global batch of 2: SyncBN statistics over both ranks == plain BN over the batch
    saved = _patch_rng()
    try:
        opt, model, tr, dev = _build(B_GLOBAL)
        inputs = {k: v.to(dev) for k, v in synth.make_inputs(B_GLOBAL, H, W, smooth=True).items()}
        outputs, _ = tr.process_batch(dict(inputs), True)
    finally:
        _restore_rng(saved)
    disp = outputs[("disp", 0)].detach().float().cpu()
    mono = outputs[("mono_disp", 0)].detach().float().cpu()
    # teacher path: only SyncBN layers see the batch -> the 2-rank result equals the global-batch result.
    # (the multi-frame disp also depends on the pose network, whose plain BatchNorm2d layers use per-rank
    # statistics in the reference too, resnet_encoder.py:35 -- so it is only required to be finite here)
    for r in range(2):
        assert (res[r][2] - mono[r:r + 1]).abs().max() < 5e-4 * mono.abs().max()
        assert torch.isfinite(res[r][1]).all() and res[r][1].shape == disp[r:r + 1].shape
    # both ranks hold identical running statistics and identical updated weights after the step
    assert torch.equal(res[0][3], res[1][3])
    assert all(l == l for l in (res[0][4], res[1][4]))


# ---------------------------------------------------------------------------------------------
# real RCCL, two devices: skipped on a one-GPU box (the development box has one); runs wherever >= 2 GPUs are visible
# ---------------------------------------------------------------------------------------------
def _nccl_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "ppea-depth_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from oracle import synth
    from ppeadepth import batchnorm, dist as pdist, networks, options, rng
    from ppeadepth.trainer import Trainer
    pdist.init_distributed("nccl")
    dev = torch.device("cuda", rank)
    opt = options.default_options(height=H, width=W, batch_size=1, use_checkpoint=True)
    model = networks.RepDepth(opt)
    synth.fill_state_dict(model)
    model.to(dev).train()
    pdist.broadcast_module(model)
    pdist.assign_groups(model)
    tr = Trainer(opt, model, dev, amp_dtype=torch.bfloat16)
    eng = pdist.TrainEngine(tr, lr=1e-4, bf16_params=True)
    assert eng.flat.hooked                      # gradient exchange overlapped with backward
    rng.set_mode("device")
    full = synth.make_inputs(B_GLOBAL, H, W, smooth=True)
    inputs = {k: v[rank:rank + 1].to(dev).contiguous() for k, v in full.items()}
    random.seed(rank)
    l_eager = float(eng.step(dict(inputs))[1]["loss"])
    eng.capture(inputs, warmup=1)               # the captured step holds the RCCL calls
    l_graph = [float(eng.step(inputs)[1]["loss"]) for _ in range(2)]
    torch.cuda.synchronize()
    sd = model.state_dict()
    probe = torch.cat([sd["encoder.replk.stem.0.bn.running_var"].float().cpu(),
                       eng.P[:4096].float().cpu()])
    q.put((rank, probe, l_eager, l_graph))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_two_ranks_rccl_eager_and_captured(device):
    """Two RCCL ranks, bf16 step, hook-driven gradient all-reduce, SyncBN exchanges on per-branch communicators,
    eager and captured: identical running statistics and master weights on both ranks, finite losses."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + os.getpid() % 90
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert torch.equal(res[0][1], res[1][1])
    for r in res:
        assert all(v == v for v in [r[2]] + r[3])
