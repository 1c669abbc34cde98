"""Training-mode batch normalisation for the RepLKNet encoders.

Reference semantics (networks/replknet_adapter.py:170-180): every BN of both encoders is an
`nn.SyncBatchNorm` (the global `use_sync_bn` flag is set by the matching encoder and never
reset) -- batch statistics over the GLOBAL batch (all ranks), running stats updated with
momentum 0.1 and the unbiased variance.  `state_dict` keys are those of nn.BatchNorm2d.

MI355X design notes
  * one rank: BN (+ activation, DropPath, residual, adapter add) is ONE launch per direction where a
    workgroup can own a channel (`ops.bn_act_channel`), statistics + apply launches otherwise;
  * several ranks: the SAME fused kernels split at the one point where ranks must talk -- local
    statistics in wire format [mean | biased var | count] (one launch, or none: the previous
    BatchNorm's apply launch / the producing GEMM's epilogue left them) -> ONE all-gather over RCCL
    -> Chan combine + running statistics + apply in one launch (`ops.sync_bn_act`, csrc/bn_sync.hip);
    backward: reduce -> ONE all-reduce of [sum_dy | sum_dy_xmu] -> apply.  Same collectives and
    combine arithmetic as torch's SyncBatchNorm, so numerics match;
  * `--use_checkpoint` in the reference re-runs every block in backward (reentrant
    checkpoint), which updates the running statistics of the BNs inside a second time with the
    same batch statistics.  With 288 GB of HBM this build never recomputes activations; the
    second update is replayed instead from the saved statistics, for all BNs at once with
    multi-tensor ops, after the forward pass (`DeferredStats.flush`), which is also when the
    reference's recompute happens.
"""
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F


class DeferredStats:
    """Second running-stat update of checkpointed BNs, applied in a few multi-tensor kernels."""

    def __init__(self):
        self.rm, self.rv, self.mean, self.invstd, self.corr, self.eps, self.mom = [], [], [], [], [], [], []
        self.calls = {}       # id(num_batches_tracked) -> [tensor, forward calls this step]

    def count(self, bn, n=1):
        e = self.calls.setdefault(id(bn.num_batches_tracked), [bn.num_batches_tracked, 0])
        e[1] += n

    def add(self, bn, mean, invstd, count):
        self.rm.append(bn.running_mean)
        self.rv.append(bn.running_var)
        self.mean.append(mean)
        self.invstd.append(invstd)
        self.corr.append(count / max(count - 1.0, 1.0))
        self.eps.append(bn.eps)
        self.mom.append(bn.momentum)
        self.count(bn)

    @torch.no_grad()
    def flush(self):
        by_n = {}
        for t, n in self.calls.values():
            by_n.setdefault(n, []).append(t)
        for n, ts in by_n.items():
            torch._foreach_add_(ts, n)
        self.calls = {}
        if not self.rm:
            return
        var = torch._foreach_pow(self.invstd, -2.0)            # biased var + eps
        torch._foreach_sub_(var, self.eps)
        torch._foreach_mul_(var, self.corr)                    # unbiased
        means = [m.to(r.dtype) for m, r in zip(self.mean, self.rm)]
        var = [v.to(r.dtype) for v, r in zip(var, self.rv)]
        mom = self.mom[0]
        # a BN that ran twice this step (stem / stage 0: current and lookup frames) appears once per
        # grad-enabled call only, so no tensor is listed twice here.
        torch._foreach_lerp_(self.rm, means, mom)
        torch._foreach_lerp_(self.rv, var, mom)
        self.__init__()


_ACTIVE_DEFERRED = None     # set by RepDepth.forward for the duration of a training forward


def set_deferred(d):
    global _ACTIVE_DEFERRED
    _ACTIVE_DEFERRED = d


def _collectives_on():
    from .dist import collectives_on
    return collectives_on()


def _log_collective(op, tensor, group):
    from .dist import log_collective
    log_collective(op, tensor, group)


class _SyncBNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, group):
        x = x.contiguous()
        C = x.shape[1]
        count = x.numel() // C
        mean, invstd = torch.batch_norm_stats(x, eps)
        packed = torch.cat([mean, invstd, mean.new_full((1,), float(count))])
        world = dist.get_world_size(group)
        gathered = torch.empty(world, 2 * C + 1, device=x.device, dtype=packed.dtype)
        _log_collective("all_gather", gathered, group)
        dist.all_gather(list(gathered.unbind(0)), packed, group=group)
        mean_all, invstd_all, counts = gathered[:, :C], gathered[:, C:2 * C], gathered[:, 2 * C]
        mean, invstd = torch.batch_norm_gather_stats_with_counts(
            x, mean_all, invstd_all, running_mean, running_var, momentum, eps, counts)
        ctx.save_for_backward(x, weight, mean, invstd, counts.to(torch.int32))
        ctx.group = group
        out = torch.batch_norm_elemt(x, weight, bias, mean, invstd, eps)
        ctx.mark_non_differentiable(mean, invstd)
        return out, mean, invstd

    @staticmethod
    def backward(ctx, dy, _dm, _di):
        x, weight, mean, invstd, counts = ctx.saved_tensors
        dy = dy.contiguous()
        sum_dy, sum_dy_xmu, gw, gb = torch.batch_norm_backward_reduce(
            dy, x, mean, invstd, weight, True, True, True)
        C = sum_dy.shape[0]
        packed = torch.cat([sum_dy, sum_dy_xmu])
        _log_collective("all_reduce", packed, ctx.group)
        dist.all_reduce(packed, group=ctx.group)
        sum_dy, sum_dy_xmu = packed[:C], packed[C:]
        dx = torch.batch_norm_backward_elemt(dy, x, mean, invstd, weight, sum_dy, sum_dy_xmu, counts)
        return dx, gw, gb, None, None, None, None, None


def _global_stats(bn, z, sums=None):
    """Batch statistics of z for `bn` in training mode: (mean, invstd, count, group).  One rank: two fused
    launches (running stats updated in the second).  Several ranks (sync BN): local stats -> one packed
    all-gather -> Chan combine; running stats updated with the global statistics."""
    from . import ops
    multi = bn.sync and _collectives_on()
    cnt = z.numel() // z.shape[1]
    if not multi:
        if sums is not None and bn.running_mean.dtype == torch.float32:
            mean, _var, invstd = ops.bn_batch_stats_from_sums(sums, cnt, bn.eps, bn.momentum, bn.running_mean,
                                                              bn.running_var)
        else:
            mean, _var, invstd = ops.bn_batch_stats(z, bn.eps, bn.momentum, bn.running_mean, bn.running_var)
        return mean, invstd, float(cnt), None
    # several ranks: stats + packed finalize (2 launches) -> ONE all-gather -> combine (1 launch)
    group = getattr(bn, "group", None)
    world = dist.get_world_size(group)
    packed = ops.bn_local_stats_packed(z)
    gathered = torch.empty(world, packed.numel(), device=z.device, dtype=packed.dtype)
    gathered[dist.get_rank(group)].copy_(packed)
    ops.gather_rows(gathered, group)
    mean, invstd = ops.bn_sync_combine(gathered, bn.eps, bn.momentum, bn.running_mean, bn.running_var)
    return mean, invstd, float(cnt * world), (group,)


PACK_PAIR = __import__("os").environ.get("PPEA_BN_PAIR", "1") == "1"


def _global_stats_pair(bn1, z1, bn2, z2, sums=None):
    """Two BNs over tensors that exist at the same time (the k x k and 5 x 5 branches of a re-parameterised large-kernel
    conv, rka.py:232-239): their SyncBN statistics travel in ONE packed all-gather instead of two."""
    from . import ops
    if not (PACK_PAIR and bn1.sync and bn2.sync and _collectives_on()) \
            or getattr(bn1, "group", None) is not getattr(bn2, "group", None):
        s1, s2 = sums if sums is not None else (None, None)
        return _global_stats(bn1, z1, s1), _global_stats(bn2, z2, s2)
    group = getattr(bn1, "group", None)
    world = dist.get_world_size(group)
    p1, p2 = ops.bn_local_stats_packed(z1), ops.bn_local_stats_packed(z2)
    n1 = p1.numel()
    packed = torch.cat([p1, p2])
    gathered = torch.empty(world, packed.numel(), device=z1.device, dtype=packed.dtype)
    gathered[dist.get_rank(group)].copy_(packed)
    ops.gather_rows(gathered, group)
    out = []
    for bn, z, g in ((bn1, z1, gathered[:, :n1]), (bn2, z2, gathered[:, n1:])):
        mean, invstd = ops.bn_sync_combine(g.contiguous(), bn.eps, bn.momentum, bn.running_mean, bn.running_var)
        out.append((mean, invstd, float(z.numel() // z.shape[1] * world), (group,)))
    return out[0], out[1]


SYNC_FUSED = __import__("os").environ.get("PPEA_SYNC_FUSED", "1") == "1"     # 0: the round-2 multi-rank path (A/B only)


def _sync_path_ok(z, bns):
    """Several ranks and every BN of the call is a SyncBN on one communicator: the fused two-launch form applies."""
    from . import ops
    if not (SYNC_FUSED and _collectives_on() and all(bn.sync and bn.training for bn in bns)):
        return False
    g0 = getattr(bns[0], "group", None)
    return (all(getattr(bn, "group", None) is g0 and bn.eps == bns[0].eps and bn.momentum == bns[0].momentum
                and bn.running_mean.dtype == torch.float32 for bn in bns) and ops.sync_bn_supported(z))


def _book_sync(bns, st, z):
    """num_batches_tracked / checkpoint-replay bookkeeping of a sync_bn_act call (st = mean1 | invstd1 | mean2 | invstd2)."""
    group = getattr(bns[0][1], "group", None)
    from . import ops
    cnt = float(z.numel() // z.shape[1] * ops.sync_world(group))
    for k, (_, bn) in enumerate(bns):
        if _ACTIVE_DEFERRED is None:
            bn.num_batches_tracked += 1
        else:
            _ACTIVE_DEFERRED.count(bn)
            if bn.replay_update and torch.is_grad_enabled():
                _ACTIVE_DEFERRED.add(bn, st[2 * k], st[2 * k + 1], cnt)


def assign_groups(model, any_backend=False):
    """One process group per concurrently running branch: ProcessGroupNCCL runs a group's collectives in order
    on one internal stream, so the teacher's and the student's SyncBN exchanges must not share a group or the
    two branches of the step serialise on it.  Call on every rank, after init_process_group."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_backend() != "nccl" and not any_backend):
        return
    groups = {}
    for top in ("mono_encoder", "encoder"):
        sub = getattr(model, top, None)
        if sub is None:
            continue
        groups[top] = dist.new_group()
        for m in sub.modules():
            if isinstance(m, BatchNorm2d):
                m.group = groups[top]
    return groups


def fused_bn_act(z1, bn1, z2=None, bn2=None, act=0, mask=None, r1=None, r2=None, r2_scale=1.0, skip=False, sums=None):
    """act(BN1(z1) [+ BN2(z2)]) [* mask[n]] [+ r1] [+ r2_scale * r2] on the fused HIP kernels
    (training mode: batch statistics, running stats and checkpoint-replay bookkeeping as BatchNorm2d).
    sums: partial (sum, sum of squares) of z1 per channel from the GEMM that produced it (ops.pwconv_frozen(...,
    want_sums=True)): the statistics pass over z1 is skipped where it would be a separate launch.
    skip: -> (y, z1'), z1' being z1 for the caller's OTHER use of it (a block's residual connection): where the
    one-launch kernels serve the shape, the gradient of that use is added inside this BN's backward launch."""
    from . import ops
    assert bn1.training, "fused_bn_act is the training-mode path; eval goes through BatchNorm2d.forward"
    bns = [(z1, bn1)] + ([(z2, bn2)] if z2 is not None else [])
    if _sync_path_ok(z1, [bn for _, bn in bns]):
        # several ranks: [statistics launch] -> all-gather -> combine + apply launch (same fused neighbours as below)
        want_skip = bool(skip and torch.is_grad_enabled() and z1.requires_grad)
        want_dup = bool(want_skip and BN_DUP and r1 is None and r2 is None and ops.bn_channel_ok(z1))
        outs = ops.sync_bn_act(z1, bn1, z2, bn2, mask=mask, r1=r1, r2=r2, r2_scale=r2_scale, act=act,
                               group=getattr(bn1, "group", None), sums=sums, skip=want_skip, dup=want_dup)
        _book_sync(bns, outs[1], z1)
        if want_dup:
            outs[0]._second_use = outs[-1]
        return (outs[0], outs[2] if want_skip else z1) if skip else outs[0]
    # small channels on one rank: statistics, running-statistics update and apply in ONE launch (backward likewise)
    if (ops.bn_channel_ok(z1) and all(bn.training for _, bn in bns) and not any(bn.sync and _collectives_on() for _, bn in bns)
            and (bn2 is None or (bn2.eps == bn1.eps and bn2.momentum == bn1.momentum))
            and bn1.running_mean.dtype == torch.float32):
        z1_skip = z1
        if skip and torch.is_grad_enabled() and z1.requires_grad:
            if BN_DUP and r1 is None and r2 is None:
                y, st, z1_skip, y_b = ops.bn_act_channel(z1, bn1, z2, bn2, mask=mask, act=act, skip=True, dup=True)
                y._second_use = y_b                      # see second_use()
            else:
                y, st, z1_skip = ops.bn_act_channel(z1, bn1, z2, bn2, mask=mask, r1=r1, r2=r2, r2_scale=r2_scale, act=act,
                                                    skip=True)
        else:
            y, st = ops.bn_act_channel(z1, bn1, z2, bn2, mask=mask, r1=r1, r2=r2, r2_scale=r2_scale, act=act,
                                       sums=sums if (z2 is None and torch.is_tensor(sums)) else None)
        cnt = float(z1.numel() // z1.shape[1])
        for k, (_, bn) in enumerate(bns):
            if _ACTIVE_DEFERRED is None:
                bn.num_batches_tracked += 1
            else:
                _ACTIVE_DEFERRED.count(bn)
                if bn.replay_update and torch.is_grad_enabled():
                    _ACTIVE_DEFERRED.add(bn, st[2 * k], st[2 * k + 1], cnt)
        return (y, z1_skip) if skip else y
    stats = []
    count, group = None, None
    pre = None
    if z2 is not None and bn1.training and bn2.training:
        pre = list(_global_stats_pair(bn1, z1, bn2, z2, sums if isinstance(sums, tuple) else None))
    for z, bn in bns:
        if bn.training:
            mean, invstd, count, group = pre.pop(0) if pre is not None else _global_stats(bn, z, sums if (z is z1 and not isinstance(sums, tuple)) else None)
            if _ACTIVE_DEFERRED is None:
                bn.num_batches_tracked += 1
            else:
                _ACTIVE_DEFERRED.count(bn)
                if bn.replay_update and torch.is_grad_enabled():
                    _ACTIVE_DEFERRED.add(bn, mean, invstd, count)
        else:
            mean = bn.running_mean.float()
            invstd = torch.rsqrt(bn.running_var.float() + bn.eps)
        stats.append((mean, invstd))
    m2, i2 = (stats[1] if z2 is not None else (None, None))
    y = ops.bn_act_apply(z1, bn1.weight, bn1.bias, stats[0][0], stats[0][1], z2,
                         None if bn2 is None else bn2.weight, None if bn2 is None else bn2.bias, m2, i2,
                         mask=mask, r1=r1, r2=r2, r2_scale=r2_scale, act=act, count=count, group=group)
    return (y, z1) if skip else y


BN_CHAIN = __import__("os").environ.get("PPEA_BN_CHAIN", "1") == "1"
# A block's first BatchNorm output feeds its first 1x1 conv AND its adapter: hand the adapter an alias of it whose gradient
# arrives separately at the BatchNorm's backward launch, which adds the two (bit-identical to autograd's own add kernel,
# one launch fewer on the dependent chain per block and network)
BN_DUP = __import__("os").environ.get("PPEA_BN_DUP", "1") == "1"


def second_use(y):
    """The tensor to hand to the SECOND consumer of a fused BatchNorm output (itself where no alias was made)."""
    return getattr(y, "_second_use", y)


def fused_bn_act_next(z, bnA, bnB, mask=None, r1=None, r2=None, r2_scale=1.0, sums=None):
    """A block's last BatchNorm (+ DropPath + residual + adapter) and the NEXT block's first BatchNorm in one launch
    per direction: -> (y, y2) = (mask * BN_A(z) + r1 + r2_scale * r2, BN_B(y)), or None when the shape / mode is not
    served by the one-launch channel kernels (the caller then runs the two BatchNorms separately).  Same numbers, running
    statistics and bookkeeping as fused_bn_act(z, bnA, ...) followed by fused_bn_act(y, bnB, skip=True)."""
    from . import ops
    if (BN_CHAIN and ops.bn_channel_ok(z) and bnA.num_features == bnB.num_features and _sync_path_ok(z, [bnA, bnB])):
        # several ranks: stats(z) -> gather -> [apply A + local statistics of y] -> gather -> apply B: the second
        # BatchNorm needs no statistics launch, and its backward adds the residual use's gradient of y in its apply launch
        group = getattr(bnA, "group", None)
        y, stA, tab = ops.sync_bn_act(z, bnA, mask=mask, r1=r1, r2=r2, r2_scale=r2_scale, group=group, emit=True, sums=sums)
        _book_sync([(z, bnA)], stA, z)
        want_skip = bool(torch.is_grad_enabled() and y.requires_grad)
        outs = ops.sync_bn_act(y, bnB, group=group, table=tab, skip=want_skip, dup=bool(want_skip and BN_DUP))
        _book_sync([(y, bnB)], outs[1], y)
        if want_skip and BN_DUP:
            outs[0]._second_use = outs[-1]
        return (outs[2] if want_skip else y), outs[0]
    if not (BN_CHAIN and ops.bn_channel_ok(z) and bnA.training and bnB.training
            and not ((bnA.sync or bnB.sync) and _collectives_on())
            and bnA.eps == bnB.eps and bnA.momentum == bnB.momentum
            and bnA.running_mean.dtype == torch.float32 and bnB.running_mean.dtype == torch.float32
            and bnA.num_features == bnB.num_features):
        return None
    if BN_DUP and torch.is_grad_enabled() and z.requires_grad:
        y, y2, st, y2_b = ops.bn_act_channel_next(z, bnA, bnB, mask=mask, r1=r1, r2=r2, r2_scale=r2_scale, dup=True)
        y2._second_use = y2_b
    else:
        y, y2, st = ops.bn_act_channel_next(z, bnA, bnB, mask=mask, r1=r1, r2=r2, r2_scale=r2_scale)
    cnt = float(z.numel() // z.shape[1])
    for k, bn in enumerate((bnA, bnB)):
        if _ACTIVE_DEFERRED is None:
            bn.num_batches_tracked += 1
        else:
            _ACTIVE_DEFERRED.count(bn)
            if bn.replay_update and torch.is_grad_enabled():
                _ACTIVE_DEFERRED.add(bn, st[2 * k], st[2 * k + 1], cnt)
    return y, y2


class BatchNorm2d(nn.Module):
    """Drop-in for nn.BatchNorm2d / nn.SyncBatchNorm (same parameters, buffers, state_dict keys)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, sync=False):
        super().__init__()
        self.num_features, self.eps, self.momentum, self.sync = num_features, eps, momentum, sync
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.replay_update = False     # inside a segment the reference checkpoints

    def extra_repr(self):
        return f"{self.num_features}, eps={self.eps}, momentum={self.momentum}, sync={self.sync}"

    def forward(self, x):
        if not self.training:
            return F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias,
                                False, self.momentum, self.eps)
        count = x.numel() // x.shape[1]
        multi = self.sync and _collectives_on()
        if multi:
            out, mean, invstd = _SyncBNFn.apply(x, self.weight, self.bias, self.running_mean,
                                                self.running_var, self.eps, self.momentum,
                                                getattr(self, "group", None))
            count = count * dist.get_world_size()
        else:
            out, mean, invstd = torch.native_batch_norm(x, self.weight, self.bias, self.running_mean,
                                                        self.running_var, True, self.momentum, self.eps)
        if _ACTIVE_DEFERRED is None:
            self.num_batches_tracked += 1
        else:
            _ACTIVE_DEFERRED.count(self)
            if self.replay_update and torch.is_grad_enabled():
                _ACTIVE_DEFERRED.add(self, mean.detach(), invstd.detach(), float(count))
        return out
