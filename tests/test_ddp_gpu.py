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


# ---------------------------------------------------------------------------------------------
# First multi-rank contact, de-risked on one GPU (VERDICT r3 "next" #1): RCCL deadlocks when two ranks issue a
# communicator's collectives in different orders.  ProcessGroupNCCL enqueues a communicator's kernels on its internal
# stream in HOST ISSUE ORDER (eager and under capture), so the property to hold is: per communicator, the ordered list of
# (op, elements, dtype) is (a) the same on every rank whatever the rank's data / random draws, (b) the same in a captured
# step as in an eager one.  Reference: Accelerate -> DDP bucketed all-reduce + SyncBatchNorm (trainer.py:215-222, 350;
# networks/replknet_adapter.py:170-180).
# ---------------------------------------------------------------------------------------------
def _split_by_communicator(log):
    per = {}
    for key, op, numel, dtype in log:
        per.setdefault(key, []).append((op, numel, dtype))
    return per


def _order_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "ppea-depth_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    from oracle import synth
    from ppeadepth import dist as pdist, rng
    pdist.init_distributed("gloo")
    opt, model, tr, dev = _build(1)
    pdist.broadcast_module(model)
    pdist.assign_groups(model, any_backend=True)          # the RCCL communicator layout, on gloo groups
    tr.amp_dtype = torch.bfloat16
    eng = pdist.TrainEngine(tr, lr=1e-4, bf16_params=True)
    assert eng.flat.hooked and eng.flat.scale_in_optimizer
    rng.set_mode("device")                                # every rank draws its own DropPath masks / augmentation
    torch.manual_seed(100 + 17 * rank)
    random.seed(5 + rank)
    # different frames on every rank (different seeds): nothing about the data may move a collective
    inputs = {k: v.to(dev).contiguous() for k, v in synth.make_inputs(1, H, W, seed=1234 + rank, smooth=True).items()}
    logs = []
    for _ in range(2):                                    # two steps: the order must not drift from step to step either
        pdist.COLLECTIVE_LOG = []
        _, losses = eng.step(dict(inputs))
        torch.cuda.synchronize()
        logs.append(pdist.COLLECTIVE_LOG)
        pdist.COLLECTIVE_LOG = None
    plan = [(a, n, e, bool(h)) for a, n, e, h in eng.flat.last_plan]
    q.put((rank, logs, plan, float(losses["loss"])))
    dist.barrier()
    dist.destroy_process_group()


def test_collective_order_is_identical_across_ranks_with_different_data(device):
    """Two gloo ranks on the RCCL communicator layout (per-branch groups, hook-driven gradient ranges, fused SyncBN),
    different frames / seeds / DropPath draws per rank: per communicator the ordered (op, elements, dtype) lists are equal
    across ranks and across steps, every gradient range is launched from a hook on every rank, and the census matches the
    model (one all-gather per SyncBN forward exchange, one all-reduce per backward exchange, 9 gradient ranges at most)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + os.getpid() % 200
    procs = [ctx.Process(target=_order_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (_, logs0, plan0, l0), (_, logs1, plan1, l1) = res
    assert l0 == l0 and l1 == l1
    per = [[_split_by_communicator(step) for step in logs] for logs in (logs0, logs1)]
    assert set(per[0][0]) == {"encoder", "mono_encoder", "adapters", "world"}, set(per[0][0])
    for key in per[0][0]:
        assert per[0][0][key] == per[1][0][key], f"rank-dependent collective order on communicator {key}"
        assert per[0][0][key] == per[0][1][key] == per[1][1][key], f"collective order drifts between steps on {key}"
    assert plan0 == plan1 and all(h for *_, h in plan0), (plan0, plan1)      # same ranges, all launched from hooks
    # the adapters' communicator carries gradient ranges only; the depth-bin tracker is the only user of the world group
    assert all(op == "all_reduce" and dt == "float32" for op, _, dt in per[0][0]["adapters"])
    assert per[0][0]["world"] == [("all_reduce_max", 2, "float32")]
    for key in ("encoder", "mono_encoder"):
        ops_ = [op for op, _, _ in per[0][0][key]]
        assert ops_.count("all_gather") > 50 and ops_.count("all_reduce") > 50


def _forced_worker(idx, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "ppea-depth_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      PPEA_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from oracle import synth
    from ppeadepth import dist as pdist, rng
    assert pdist.FORCE_COLLECTIVES
    pdist.init_distributed("nccl")
    assert pdist.collectives_on()
    opt, model, tr, dev = _build(2)
    pdist.assign_groups(model)
    assert set(pdist.COMM) == {"encoder", "mono_encoder", "adapters"}
    tr.amp_dtype = torch.bfloat16
    eng = pdist.TrainEngine(tr, lr=1e-4, bf16_params=True)
    assert eng.flat.hooked
    rng.set_mode("reference")
    inputs = {k: v.to(dev).contiguous() for k, v in synth.make_inputs(2, H, W, seed=1234 + idx, smooth=True).items()}
    seed = 3 + idx
    snap = eng.snapshot()

    def run():
        torch.manual_seed(seed)
        random.seed(seed)
        pdist.COLLECTIVE_LOG = []
        outputs, losses = eng.step(dict(inputs) if eng.graph is None else inputs)
        torch.cuda.synchronize()
        log, pdist.COLLECTIVE_LOG = pdist.COLLECTIVE_LOG, None
        grads = {k: v.detach().clone() for k, v in eng.named_grads().items()}
        return log, {k: v.detach().clone() for k, v in losses.items()}, outputs[("disp", 0)].detach().clone(), grads

    log_e, loss_e, disp_e, grads_e = run()
    eng.restore(snap)
    _, loss_e2, disp_e2, grads_e2 = run()                  # the eager step itself is reproducible (no float atomics)
    eager_repro = [k for k in grads_e if not torch.equal(grads_e[k], grads_e2[k])]
    eng.restore(snap)
    torch.manual_seed(seed)
    random.seed(seed)
    pdist.COLLECTIVE_LOG = []
    eng.capture(inputs, warmup=1, restore_state=True)      # one eager warm-up step + the captured step
    log_c, pdist.COLLECTIVE_LOG = pdist.COLLECTIVE_LOG, None
    log_r, loss_g, disp_g, grads_g = run()

    def worst(a, b):
        bad = [(float((a[k].float() - b[k].float()).abs().max() / (b[k].float().abs().max() + 1e-30)), k)
               for k in a if not torch.equal(a[k], b[k])]
        return len(bad), (max(bad) if bad else (0.0, None))

    diff = worst(grads_g, grads_e)
    dloss = {k: (float(loss_e[k]), float(loss_g[k])) for k in loss_e if not torch.equal(loss_e[k], loss_g[k])}
    q.put((idx, log_e, log_c, log_r, diff, dloss, bool(torch.equal(disp_e, disp_g)), len(grads_e),
           [(a, n, e, bool(h)) for a, n, e, h in eng.flat.last_plan], len(eager_repro)))
    dist.destroy_process_group()


def test_forced_collectives_captured_step_equals_eager_step_bitwise_and_issues_the_same_collectives(device):
    """PPEA_FORCE_COLLECTIVES=1 on one GPU (every collective of the multi-rank step goes through ProcessGroupNCCL / RCCL
    with a single rank), bf16 step, two independent processes with different frames and seeds: (1) the step replayed from
    the hipGraph equals the eager step BIT FOR BIT (losses, disp, every gradient); (2) the capture issues exactly the
    eager step's collective sequence per communicator, and a replay issues none from the host; (3) the sequence does not
    depend on the process's data."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    base = 29100 + os.getpid() % 200
    procs = [ctx.Process(target=_forced_worker, args=(i, base + i, q)) for i in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=420) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for idx, log_e, log_c, log_r, diff, dloss, disp_eq, n_grads, plan, eager_repro in res:
        assert eager_repro == 0, f"{eager_repro} gradients differ between two eager steps from the same state"
        assert not dloss and disp_eq, (idx, dloss)
        assert n_grads > 1000 and diff[0] == 0, (idx, diff)
        assert log_r == [], "a replay must not issue collectives from the host"
        pe, pc = _split_by_communicator(log_e), _split_by_communicator(log_c)
        assert set(pe) == {"encoder", "mono_encoder", "adapters", "world"}
        for key in pe:
            assert pc[key] == pe[key] * 2, f"captured collective sequence differs from the eager one on {key}"
        assert all(h for *_, h in plan)
    assert _split_by_communicator(res[0][1]) == _split_by_communicator(res[1][1])
