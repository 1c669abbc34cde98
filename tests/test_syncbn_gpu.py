"""SyncBatchNorm on the fused kernels (csrc/bn_sync.hip, ops.sync_bn_act): W simulated ranks in ONE process against the
one-rank fused path on the concatenated (global) batch.

Reference semantics: both encoders are nn.SyncBatchNorm (networks/replknet_adapter.py:170-180) under DDP
(trainer.py:215-222): global-batch statistics, local d gamma / d beta averaged over ranks.

The collectives are simulated by fixed-point replay: every pass runs all ranks' forward + backward with each collective
served from the values the other ranks contributed to the SAME call in the previous pass; after (number of dependent
collectives + 1) passes every value is the true one.  The real collectives run in tests/test_ddp_gpu.py (2 processes).
"""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _g(seed):
    return torch.Generator().manual_seed(seed)


class Replay:
    """Stand-in for ops.gather_rows / ops.reduce_sums / ops.sync_world."""

    def __init__(self, world):
        self.world, self.prev, self.cur, self.rank, self.idx = world, {}, {}, 0, 0

    def begin_pass(self):
        self.prev, self.cur = self.cur, {}

    def begin_rank(self, r):
        self.rank, self.idx = r, 0

    def _slot(self, t):
        from ppeadepth import ops
        ops._count(collectives=1)
        i = self.idx
        self.idx += 1
        self.cur.setdefault(i, {})[self.rank] = t.detach().clone()
        return [t if r == self.rank else self.prev.get(i, {}).get(r, t) for r in range(self.world)]

    def gather(self, table, group):                        # in place: this rank's row arrives filled
        rows = self._slot(table[self.rank])
        for r in range(self.world):
            if r != self.rank:
                table[r].copy_(rows[r])

    def reduce(self, sums, group):
        rows = self._slot(sums)
        sums.copy_(torch.stack(rows).sum(0))

    def install(self, monkeypatch):
        from ppeadepth import batchnorm, ops
        monkeypatch.setattr(ops, "gather_rows", self.gather)
        monkeypatch.setattr(ops, "reduce_sums", self.reduce)
        monkeypatch.setattr(ops, "sync_world", lambda group: self.world)
        monkeypatch.setattr(ops, "sync_rank", lambda group: self.rank)
        monkeypatch.setattr(batchnorm, "_collectives_on", lambda: True)


def _mk_bn(C, device, g, sync):
    from ppeadepth.batchnorm import BatchNorm2d
    bn = BatchNorm2d(C, sync=sync).to(device)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.2)
    return bn


def _copy_bn(bn, sync):
    from ppeadepth.batchnorm import BatchNorm2d
    out = BatchNorm2d(bn.num_features, sync=sync).to(bn.weight.device)
    out.load_state_dict(bn.state_dict())
    return out


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("two,act,res", [(False, 0, False), (False, 1, False), (True, 1, False), (False, 2, False),
                                         (True, 0, True), (False, 0, True)])
@pytest.mark.parametrize("world,shape", [(2, (4, 64, 12, 40)),       # a workgroup owns a channel (stages 2 / 3)
                                         (3, (2, 128, 6, 24)),
                                         (2, (3, 32, 48, 160)),      # flat element-wise pass (stages 0 / 1)
                                         (2, (2, 8, 24, 8))])        # flat pass, many planes per workgroup, C < 64
def test_sync_bn_act_equals_the_global_batch(device, monkeypatch, dtype, two, act, res, world, shape):
    from ppeadepth import ops
    from ppeadepth.batchnorm import fused_bn_act
    n, C, H, W = shape
    N = n * world
    g = _g(N * 100 + C + act)
    dt = torch.float32 if dtype == "f32" else torch.bfloat16
    z1 = (torch.randn(N, C, H, W, generator=g) * 2 + 0.5).to(dt).to(device)
    z1[:n] += 1.5                                              # ranks with different local means: the combine matters
    z2 = (torch.randn(N, C, H, W, generator=g) * 0.7 - 0.2).to(dt).to(device)
    mask = torch.tensor(([0.0] + [1.4] * N)[:N], device=device)
    r1 = torch.randn(N, C, H, W, generator=g).to(dt).to(device)
    r2 = torch.randn(N, C, H, W, generator=g).to(dt).to(device)
    go = torch.randn(N, C, H, W, generator=g).to(dt).to(device)
    bn1, bn2 = _mk_bn(C, device, g, False), _mk_bn(C, device, g, False)

    def run(bna, bnb, sl):
        leaves = [t[sl].clone().requires_grad_(True) for t in (z1, z2, r1, r2)]
        x = leaves[0] * 1                                          # non-leaf, as a block's input is
        kw = dict(z2=leaves[1], bn2=bnb) if two else {}
        if res:
            kw.update(mask=mask[sl], r1=leaves[2], r2=leaves[3], r2_scale=0.5)
        y, xs = fused_bn_act(x, bna, act=act, skip=True, **kw)
        ((y.float() * go[sl].float()).sum() + (xs.float() * r2[sl].float()).sum()).backward()
        return y.detach(), [l.grad for l in leaves]

    y_ref, gr_ref = run(bn1, bn2, slice(0, N))                    # one rank, the global batch

    rep = Replay(world)
    rep.install(monkeypatch)
    for _ in range(3):                                            # 2 dependent collectives (gather, reduce) + 1
        rep.begin_pass()
        outs = []
        for r in range(world):
            rep.begin_rank(r)
            a, b = _copy_bn(bn1, True), _copy_bn(bn2, True)
            with torch.no_grad():
                a.running_mean.zero_(); a.running_var.fill_(1.0); b.running_mean.zero_(); b.running_var.fill_(1.0)
            ops.SYNC_COUNTERS = {}
            outs.append((a, b) + run(a, b, slice(r * n, (r + 1) * n)) + (dict(ops.SYNC_COUNTERS),))
            ops.SYNC_COUNTERS = None
    tol_f, tol_b = (2e-5, 2e-4) if dtype == "f32" else (1e-2, 3e-2)
    chan = ops.bn_channel_ok(z1[:n])
    for r, (a, b, y, gr, counts) in enumerate(outs):
        sl = slice(r * n, (r + 1) * n)
        assert rel_err(y.float(), y_ref[sl].float()) < tol_f
        assert rel_err(a.running_mean, bn1.running_mean) < 1e-5 and rel_err(a.running_var, bn1.running_var) < 1e-4
        for k in range(4):
            if gr_ref[k] is not None:
                assert rel_err(gr[k].float(), gr_ref[k][sl].float()) < tol_b, k
        if two:
            assert rel_err(b.running_var, bn2.running_var) < 1e-4
        # two launches around one collective per direction where a workgroup owns a channel
        assert counts["collectives"] == 2
        assert counts["launches"] == (4 if chan else (7 if two else 6))
    # d gamma / d beta: the DDP mean over ranks == (gradient of the summed loss) / world (trainer.py:215-222)
    for ref_bn, k in ((bn1, 0),) + (((bn2, 1),) if two else ()):
        gw = torch.stack([o[k].weight.grad for o in outs]).mean(0)
        gb = torch.stack([o[k].bias.grad for o in outs]).mean(0)
        assert rel_err(gw * world, ref_bn.weight.grad) < tol_b and rel_err(gb * world, ref_bn.bias.grad) < tol_b


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_sync_bn_chain_equals_the_one_rank_chain(device, monkeypatch, dtype):
    """A block's last BatchNorm + the next block's first one (fused_bn_act_next): several ranks run statistics -> gather ->
    [apply A + local statistics of y] -> gather -> apply B, i.e. 3 launches and 2 collectives forward."""
    from ppeadepth import ops
    from ppeadepth.batchnorm import fused_bn_act_next
    world, n, C, H, W = 2, 3, 64, 12, 40
    N = n * world
    g = _g(5)
    dt = torch.float32 if dtype == "f32" else torch.bfloat16
    z = (torch.randn(N, C, H, W, generator=g) * 2 + 0.5).to(dt).to(device)
    z[:n] -= 1.0
    x = torch.randn(N, C, H, W, generator=g).to(dt).to(device)
    ad = torch.randn(N, C, H, W, generator=g).to(dt).to(device)
    mask = torch.tensor([1.25, 0.0, 1.25, 1.25, 1.25, 0.0], device=device)
    go, gs = (torch.randn(N, C, H, W, generator=g).to(dt).to(device) for _ in range(2))
    bnA, bnB = _mk_bn(C, device, g, False), _mk_bn(C, device, g, False)

    def run(a, b, sl):
        leaves = [t[sl].clone().requires_grad_(True) for t in (z, x, ad)]
        y, y2 = fused_bn_act_next(leaves[0] * 1, a, b, mask=mask[sl], r1=leaves[1], r2=leaves[2], r2_scale=0.5)
        ((y2.float() * go[sl].float()).sum() + (y.float() * gs[sl].float()).sum()).backward()
        return y.detach(), y2.detach(), [l.grad for l in leaves]

    y_ref, y2_ref, gr_ref = run(bnA, bnB, slice(0, N))
    rep = Replay(world)
    rep.install(monkeypatch)
    for _ in range(5):                                            # 4 dependent collectives + 1
        rep.begin_pass()
        outs = []
        for r in range(world):
            rep.begin_rank(r)
            a, b = _copy_bn(bnA, True), _copy_bn(bnB, True)
            with torch.no_grad():
                a.running_mean.zero_(); a.running_var.fill_(1.0); b.running_mean.zero_(); b.running_var.fill_(1.0)
            ops.SYNC_COUNTERS = {}
            outs.append((a, b) + run(a, b, slice(r * n, (r + 1) * n)) + (dict(ops.SYNC_COUNTERS),))
            ops.SYNC_COUNTERS = None
    tol_f, tol_b = (2e-5, 3e-4) if dtype == "f32" else (1e-2, 3e-2)
    for r, (a, b, y, y2, gr, counts) in enumerate(outs):
        sl = slice(r * n, (r + 1) * n)
        assert rel_err(y.float(), y_ref[sl].float()) < tol_f and rel_err(y2.float(), y2_ref[sl].float()) < tol_f
        for bn_s, bn_r in ((a, bnA), (b, bnB)):
            assert rel_err(bn_s.running_mean, bn_r.running_mean) < 1e-5
            assert rel_err(bn_s.running_var, bn_r.running_var) < 1e-4
        for k in range(3):
            assert rel_err(gr[k].float(), gr_ref[k][sl].float()) < tol_b, k
        assert counts == {"launches": 3 + 4, "collectives": 4}
    for k, ref_bn in ((0, bnA), (1, bnB)):
        gw = torch.stack([o[k].weight.grad for o in outs]).mean(0)
        assert rel_err(gw * world, ref_bn.weight.grad) < tol_b


@pytest.mark.parametrize("chained", [True, False])
def test_sync_bn_second_consumer_gradient_is_merged_in_the_reduce_launch(device, monkeypatch, chained):
    """The BatchNorm output that feeds a block's first 1x1 conv AND its adapter: with BN_DUP the two gradients reach the
    backward separately and the reduce launch merges them (round(dy + dyb), stored for the apply launch) -- bit-identical
    to autograd's own element-wise add in front of the two launches, on every simulated rank."""
    from ppeadepth import batchnorm, ops
    from ppeadepth.batchnorm import fused_bn_act, fused_bn_act_next, second_use
    world, n, C, H, W = 2, 3, 64, 12, 40
    N = n * world
    g = _g(9)
    dt = torch.bfloat16
    z = (torch.randn(N, C, H, W, generator=g) * 2 + 0.5).to(dt).to(device)
    x = torch.randn(N, C, H, W, generator=g).to(dt).to(device)
    ga, gb, gs = (torch.randn(N, C, H, W, generator=g).to(dt).to(device) for _ in range(3))
    bnA, bnB = _mk_bn(C, device, g, False), _mk_bn(C, device, g, False)

    def run(a, b, sl):
        leaves = [t[sl].clone().requires_grad_(True) for t in (z, x)]
        if chained:
            y, y2 = fused_bn_act_next(leaves[0] * 1, a, b, r1=leaves[1])
        else:
            y2, y = fused_bn_act(leaves[0] * 1, b, skip=True)
        first, second = y2, second_use(y2)
        ((first.float() * ga[sl].float()).sum() + (second.float() * gb[sl].float()).sum()
         + (y.float() * gs[sl].float()).sum()).backward()
        return [l.grad for l in leaves] + [b.weight.grad.clone(), b.bias.grad.clone()]

    res = {}
    for dup in (False, True):
        monkeypatch.setattr(batchnorm, "BN_DUP", dup)
        rep = Replay(world)
        rep.install(monkeypatch)
        for _ in range(5):
            rep.begin_pass()
            outs = []
            for r in range(world):
                rep.begin_rank(r)
                a, b = _copy_bn(bnA, True), _copy_bn(bnB, True)
                ops.SYNC_COUNTERS = {}
                outs.append(run(a, b, slice(r * n, (r + 1) * n)) + [dict(ops.SYNC_COUNTERS)])
                ops.SYNC_COUNTERS = None
        res[dup] = outs
    for r in range(world):
        for k, (u, v) in enumerate(zip(res[False][r][:-1], res[True][r][:-1])):
            if k == 1 and not chained:
                assert u is None and v is None                    # (x is not an input of the unchained form)
            else:
                assert u is not None and torch.equal(u, v), k
        assert res[False][r][-1] == res[True][r][-1]              # same launches of ours, same collectives


def test_sync_bn_statistics_from_the_gemm_epilogue(device, monkeypatch):
    """Stages 0 / 1: the 1x1 conv's epilogue leaves per-channel partial sums; with several ranks they become the wire-format
    local statistics in one tiny launch (no pass over the activation)."""
    from ppeadepth import ops
    from ppeadepth.batchnorm import fused_bn_act
    world, n, K, M, H, W = 2, 6, 128, 128, 48, 160
    g = _g(9)
    x = torch.randn(world * n, K, H, W, generator=g).bfloat16().to(device)
    x[:n] += 0.5
    w = (torch.randn(M, K, 1, 1, generator=g) / K ** 0.5).to(device)
    bn = _mk_bn(M, device, g, False)
    y_ref = fused_bn_act(ops.pwconv_frozen(x, w), bn, act=1)
    rep = Replay(world)
    rep.install(monkeypatch)
    for _ in range(2):
        rep.begin_pass()
        outs = []
        for r in range(world):
            rep.begin_rank(r)
            b = _copy_bn(bn, True)
            with torch.no_grad():
                b.running_mean.zero_(); b.running_var.fill_(1.0)
            z, sums = ops.pwconv_frozen(x[r * n:(r + 1) * n], w, want_sums=True)
            ops.SYNC_COUNTERS = {}
            outs.append((b, fused_bn_act(z, b, act=1, sums=sums), dict(ops.SYNC_COUNTERS)))
            ops.SYNC_COUNTERS = None
    for r, (b, y, counts) in enumerate(outs):
        assert rel_err(y.float(), y_ref[r * n:(r + 1) * n].float()) < 1e-2
        assert rel_err(b.running_mean, bn.running_mean) < 1e-4 and rel_err(b.running_var, bn.running_var) < 1e-3
        assert counts == {"launches": 2, "collectives": 1}
