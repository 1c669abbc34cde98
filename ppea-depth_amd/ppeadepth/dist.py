"""Data-parallel plumbing: one process per GPU, `torch.distributed` (backend "nccl" == RCCL on ROCm,
over xGMI inside a node; "gloo" for CPU tests).

Reference behaviour (trainer.py:215-222, 350; Accelerate -> DDP): per-rank mini-batch, mean
all-reduce of the trainable gradients (80.3 M fp32 = 321 MB at 31B; 1 306 tensors, DDP's default
25 MB buckets), SyncBatchNorm statistics, torchmetrics min/max of the depth-bin tracker.

MI355X design: xGMI is point-to-point (7 links x ~153 GB/s per GPU), so few, large collectives win.
All trainable gradients live in ONE flat fp32 buffer (`FlatGrads`): `p.grad` of every trainable
parameter is a view into it, autograd accumulates straight into the buffer, zeroing is one
memset, and the exchange is a handful of all-reduces of <= 84 MB each, issued from autograd hooks IN LINE on the
stream that produced a range's gradients -- and on that stream's own communicator -- as soon as backward has produced
the last gradient of the range (`FlatGrads.install_hooks`; ranges = branches of the step, laid out in the order
backward finishes them).
"""
import os

import torch
import torch.distributed as dist


# Test hook: run every collective of the multi-rank path (SyncBN statistics, gradient all-reduce) through the
# process group even when it has a single rank -- exercises ProcessGroupNCCL/RCCL plumbing (incl. graph
# capture) on a one-GPU box.
FORCE_COLLECTIVES = os.environ.get("PPEA_FORCE_COLLECTIVES") == "1"


def collectives_on():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torchrun contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or FORCE_COLLECTIVES) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


# One communicator per concurrently running branch of the step (ProcessGroupNCCL orders a group's collectives on one
# internal stream; RCCL communicators used from several streams at once need a consistent order on every rank).  Every
# stream of the step therefore talks through exactly ONE communicator, and no communicator is used from two streams:
#   step stream    (student trunk SyncBN, decoders, pose network, leftovers after backward) -> COMM["encoder"]
#   teacher stream (teacher SyncBN, teacher decoder / adapters)                            -> COMM["mono_encoder"]
#   adapter stream (student adapters' gradients)                                           -> COMM["adapters"]
COMM = {}
_COMM_OF_BRANCH = {"depth": "encoder", "encoder": "encoder", "pose": "encoder", "mono_depth": "mono_encoder",
                   "mono_encoder": "mono_encoder", "encoder_adapters": "adapters"}
if os.environ.get("PPEA_POSE_SIDE", "1") == "1":      # networks/repdepth.py POSE_SIDE: the pose network lives on the adapter stream
    _COMM_OF_BRANCH["pose"] = "adapters"
_KEY_OF = {}              # id(process group) -> communicator key ("encoder" / "mono_encoder" / "adapters"); else "world"

# Tests / bench.py: while this is a list, every collective the step issues appends (communicator key, op, elements,
# dtype) to it in HOST ISSUE ORDER -- the order in which ProcessGroupNCCL enqueues the communicator's kernels on its
# internal stream, eager or under capture.  RCCL requires that order to be the same on every rank, per communicator
# (tests/test_ddp_gpu.py::test_collective_order_*; bench.py reports the per-step census as the `rccl` object).
COLLECTIVE_LOG = None


def key_of(group):
    return "world" if group is None else _KEY_OF.get(id(group), "world")


def log_collective(op, tensor, group=None):
    if COLLECTIVE_LOG is not None:
        COLLECTIVE_LOG.append((key_of(group), op, int(tensor.numel()), str(tensor.dtype).replace("torch.", "")))


def assign_groups(model, any_backend=False):
    """Create the per-branch communicators and hand the SyncBN layers theirs.  Call on every rank, after
    init_process_group, before the first step.  RCCL only by default (gloo serialises on the host anyway);
    `any_backend` builds the same communicator layout on gloo (the collective-order tests)."""
    from . import batchnorm
    COMM.clear()
    _KEY_OF.clear()
    groups = batchnorm.assign_groups(model, any_backend)
    if groups:
        COMM.update(groups)
        COMM["adapters"] = dist.new_group()
        _KEY_OF.update({id(g): k for k, g in COMM.items()})
    return COMM


def comm_of(branch):
    """The communicator of a gradient range: a pure function of the range's branch (never of where it is launched)."""
    return COMM.get(_COMM_OF_BRANCH.get(branch, "encoder"))


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def broadcast_module(module, src=0):
    """Replicate parameters and buffers from `src` (what DDP does at construction)."""
    if world_size() == 1:
        return
    # one flat broadcast per dtype (2 888 tensors at 31B: per-tensor broadcasts are latency-bound)
    by_dtype = {}
    for t in list(module.parameters()) + list(module.buffers()):
        by_dtype.setdefault(t.dtype, []).append(t.data)
    with torch.no_grad():
        for ts in by_dtype.values():
            flat = torch.cat([t.reshape(-1) for t in ts])
            dist.broadcast(flat, src)
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()


class FlatGrads:
    """One contiguous fp32 gradient buffer for the optimizer's tensors + chunked mean all-reduce.
    `targets[i].grad` is a view into the buffer; `gather(sources)` fills it from the model parameters'
    freshly produced gradients (bf16 or fp32) with ONE multi-tensor copy."""

    def __init__(self, targets, n_chunks=4, align=1):
        self.targets = list(targets)
        self.scale_in_optimizer = False    # True: the exchange leaves the SUM over ranks, the optimizer kernel divides by world
        device = self.targets[0].device
        # every tensor starts at a multiple of `align` elements (flat optimizer: parameter views keep the
        # 256-byte alignment the GEMM / conv kernels expect of a weight pointer)
        self.offsets, off = [], 0
        for p in self.targets:
            self.offsets.append(off)
            off += -(-p.numel() // align) * align
        self.numel = off
        self.flat = torch.zeros(self.numel, device=device, dtype=torch.float32)
        self.views = []
        for p, off in zip(self.targets, self.offsets):
            v = self.flat[off:off + p.numel()].view_as(p)
            self.views.append(v)
            p.grad = v
        n_chunks = max(1, min(n_chunks, len(self.targets)))
        target = self.numel / n_chunks                     # chunk boundaries on tensor boundaries
        self.bounds = [0]
        for off in self.offsets[1:]:
            if off >= target * len(self.bounds) and len(self.bounds) < n_chunks:
                self.bounds.append(off)
        self.bounds.append(self.numel)
        self.comm_stream = torch.cuda.Stream(device) if device.type == "cuda" else None
        self.hooked = False

    def zero(self):
        self.flat.zero_()

    def gather(self, sources):
        """views[i] <- sources[i].grad (zero where a parameter received no gradient)."""
        groups = {}                    # one multi-tensor copy per source dtype (mixed lists take the slow path)
        for v, p in zip(self.views, sources):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                dst, src = groups.setdefault(p.grad.dtype, ([], []))
                dst.append(v)
                src.append(p.grad)
        for dst, src in groups.values():
            torch._foreach_copy_(dst, src)
        for t, v in zip(self.targets, self.views):
            t.grad = v

    # ---- gradient exchange overlapped with backward -------------------------------------------------------------
    # `sources[i]` is the model parameter whose gradient lands in `views[i]` (the parameter itself, or the bf16 working
    # copy of an fp32 master).  The buffer is cut into RANGES of tensors whose gradients autograd produces on the same
    # stream (`groups`: one id per target; TrainEngine derives them from the branch structure of the step -- teacher
    # encoder / decoder on the teacher's stream, student adapters on the adapter stream, the rest on the step stream).
    # A post-accumulate hook on every source counts its range down; the hook that completes a range enqueues, IN LINE
    # on the stream that produced the range, ONE multi-tensor copy per dtype into the flat buffer and the range's mean
    # all-reduce -- while the other branches of backward keep the GPU busy.  No extra stream and no new fork/join edges:
    # measured on MI355X, a separate communication stream that joins several branches mid-backward costs the captured
    # step 9 ms (the hipGraph executor stops overlapping the branches), in-line launches do not.  A range whose
    # gradients turn out to come from more than one stream is not launched from a hook but after backward, where
    # autograd has joined all streams (so correctness never rests on the grouping rule).
    # Reference: DDP's bucketed all-reduce overlapped with backward (trainer.py:220-222, 350).
    def install_hooks(self, sources, groups=None):
        self.sources = list(sources)
        assert len(self.sources) == len(self.views)
        if groups is None:                     # ranges = the size-balanced chunks
            import bisect
            groups = [bisect.bisect_right(self.bounds, off) - 1 for off in self.offsets]
        ranges, start = [], 0                  # maximal runs of equal group id -> [first tensor, last tensor + 1)
        for i in range(1, len(groups) + 1):
            if i == len(groups) or groups[i] != groups[start]:
                ranges.append((start, i))
                start = i
        self._ranges = ranges
        self._range_branch = [groups[a] for a, _ in ranges]        # branch name (or chunk index) of each range
        self._range_of = [k for k, (a, b) in enumerate(ranges) for _ in range(a, b)]
        self._active = False
        self._launched = [True] * len(ranges)
        for i, p in enumerate(self.sources):
            p.register_post_accumulate_grad_hook(lambda _p, i=i: self._on_grad(i))
        self.hooked = True

    def begin_backward(self):
        """Call right before loss.backward(): arms the hooks for one backward pass."""
        self._left = [b - a for a, b in self._ranges]
        self._streams = [set() for _ in self._ranges]
        self._fired = set()
        self._launched = [False] * len(self._ranges)
        self._active = True

    def _on_grad(self, i):
        if not self._active or i in self._fired:
            return
        self._fired.add(i)
        k = self._range_of[i]
        cuda = self.flat.is_cuda
        if cuda:
            self._streams[k].add(torch.cuda.current_stream())
        self._left[k] -= 1
        if self._left[k] == 0 and (not cuda or len(self._streams[k]) == 1):
            self._launch_range(k)              # in line, on the stream that produced every gradient of the range

    @torch.no_grad()
    def _launch_range(self, k, after_backward=False):
        self._launched[k] = True
        a, b = self._ranges[k]
        groups = {}
        for i in range(a, b):
            g = self.sources[i].grad
            if i not in self._fired or g is None:
                self.views[i].zero_()                      # no gradient this step (unused parameter)
            elif g.data_ptr() != self.views[i].data_ptr():
                dst, src = groups.setdefault(g.dtype, ([], []))
                dst.append(self.views[i])
                src.append(g)
        for dst, src in groups.values():
            torch._foreach_copy_(dst, src)
        lo = self.offsets[a]
        hi = self.offsets[b] if b < len(self.offsets) else self.numel
        if collectives_on():
            chunk = self.flat[lo:hi]
            # ALWAYS the communicator of the range's branch -- from a hook (on the stream that produced the range) and
            # after backward (on the step stream, which autograd has joined with every stream it used) alike.  A range that
            # one rank launches from a hook and another from finish() (a rank-dependent unused parameter) thus still meets
            # its peers on the same communicator, and ProcessGroupNCCL orders it on that communicator's internal stream
            # by host issue order either way.
            group = comm_of(self._range_branch[k])
            log_collective("all_reduce", chunk, group)
            dist.all_reduce(chunk, group=group)
            if not self.scale_in_optimizer:        # (the flat Adam kernel multiplies by 1 / world as it reads the sums)
                chunk.mul_(1.0 / world_size())

    def finish(self):
        """After backward (autograd has joined every stream it used with the current one): launch what the hooks did
        not -- ranges with unused parameters or mixed producer streams -- and hand the views to the optimizer."""
        self.last_plan = []                    # (first tensor, tensors, elements, launched from a hook?) per range
        for k, (a, b) in enumerate(self._ranges):
            hi = self.offsets[b] if b < len(self.offsets) else self.numel
            self.last_plan.append((a, b - a, hi - self.offsets[a], bool(self._launched[k])))
            if not self._launched[k]:
                self._launch_range(k, after_backward=True)
        self._active = False
        for t, v in zip(self.targets, self.views):
            t.grad = v

    def all_reduce_mean(self):
        """Mean over ranks (DDP semantics) in a few large chunks on a side stream."""
        w = world_size()
        if not collectives_on():
            return
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                for a, b in zip(self.bounds[:-1], self.bounds[1:]):
                    chunk = self.flat[a:b]
                    log_collective("all_reduce", chunk)
                    dist.all_reduce(chunk)
                    chunk.mul_(1.0 / w)
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            for a, b in zip(self.bounds[:-1], self.bounds[1:]):
                chunk = self.flat[a:b]
                log_collective("all_reduce", chunk)
                dist.all_reduce(chunk)
                chunk.mul_(1.0 / w)


FLAT_ADAM = True     # one-launch Adam over a flat parameter buffer (GPU, fused path)
# several ranks: chunked gradient all-reduce launched from autograd hooks during backward (PPEA_OVERLAP=0: after it)
OVERLAP_ALLREDUCE = os.environ.get("PPEA_OVERLAP", "1") == "1"
N1_RANGE_HOOKS = os.environ.get("PPEA_N1_HOOKS", "1") == "1"     # one rank: pack gradient ranges from the hooks too


# Branches of the step's backward pass, in the order they finish (networks/repdepth.py forks the teacher onto a side
# stream and replknet_adapter.py the student's adapters onto another): gradients of one branch are produced on one stream.
_BRANCH_ORDER = ("depth", "mono_depth", "encoder_adapters", "encoder", "mono_encoder", "pose")


_STAGE0_INLINE = os.environ.get("PPEA_POSE_SIDE", "1") == "1" and os.environ.get("PPEA_POSE_SIDE_INLINE0", "1") == "1"


def _branch_of(name):
    top = name.split(".")[0]
    if top in ("mono_depth", "mono_encoder", "depth"):
        return top
    if top == "encoder":
        if _STAGE0_INLINE and name.startswith("encoder.replk.stages.0."):
            return "encoder"     # (networks/repdepth.py POSE_SIDE: the student's stage-0 adapters run on the step stream)
        return "encoder_adapters" if (".adapter." in name or ".mlp_adapter." in name) else "encoder"
    return "pose"            # pose_encoder, pose (and anything else: launched after backward if streams mix)


class TrainEngine:
    """process_batch -> backward -> gradient exchange -> Adam step (trainer.py:345-351)."""

    def __init__(self, trainer, lr=None, n_chunks=4, fused_adam=None, bf16_params=False):
        self.trainer = trainer
        model = trainer._module()
        self.masters = None
        if bf16_params:
            self._to_bf16_params(model)
        self.params = [p for p in model.parameters() if p.requires_grad]
        # reverse registration order ~ the order in which backward finishes the gradients
        self.params = list(reversed(self.params))
        on_gpu = self.params[0].is_cuda
        # One rank: the same hooks pack each branch's gradients into the flat buffer on the stream that produced them (the
        # teacher's and the adapters' multi-tensor copies then run beside the student's backward instead of after it).
        self.range_hooks = (collectives_on() and OVERLAP_ALLREDUCE) or (N1_RANGE_HOOKS and on_gpu and FLAT_ADAM
                                                                         and (fused_adam is None or fused_adam))
        if self.range_hooks:
            # tensors of one backward branch sit together (stable sort: reverse registration order inside a branch)
            names = {id(p): n for n, p in model.named_parameters()}
            rank_of = {b: i for i, b in enumerate(_BRANCH_ORDER)}
            self.params.sort(key=lambda p: rank_of[_branch_of(names[id(p)])])
        lr = trainer.opt.learning_rate if lr is None else lr
        if fused_adam is None:
            fused_adam = on_gpu
        # the tensors Adam updates: the fp32 master of a bf16 parameter, else the parameter itself
        self.opt_params = [self.masters.get(id(p), p) for p in self.params] if self.masters else self.params
        self._lo = [p for p in self.params if self.masters and id(p) in self.masters]
        self._hi = [self.masters[id(p)] for p in self._lo]
        self.flat_adam = bool(fused_adam and on_gpu and FLAT_ADAM)
        if self.flat_adam:
            # bf16-backed tensors first: their working copies mirror the head of the flat fp32 buffer
            order = [i for i, p in enumerate(self.params) if self.masters and id(p) in self.masters] + \
                    [i for i, p in enumerate(self.params) if not (self.masters and id(p) in self.masters)]
            self.params = [self.params[i] for i in order]
            self.opt_params = [self.opt_params[i] for i in order]
            self._lo = [p for p in self.params if self.masters and id(p) in self.masters]
            self._hi = [self.masters[id(p)] for p in self._lo]
        self.flat = FlatGrads(self.opt_params, n_chunks, align=128 if self.flat_adam else 1) \
            if (collectives_on() or self.flat_adam) else None
        # Gradient scale read by the flat Adam kernel: fixed HERE, with the world size of construction time (it
        # was re-derived from collectives_on() at every step and could disagree with what the exchange had left).
        self.grad_scale = 1.0
        if self.flat is not None and self.flat_adam and collectives_on() and OVERLAP_ALLREDUCE:
            # CONTRACT: with this flag `p.grad` / `FlatGrads.flat` hold the SUM over ranks after a step, not the mean; the
            # optimizer kernel multiplies by `grad_scale` = 1 / world as it reads them.  Readers that want DDP's mean
            # gradients (logging, clipping, export) use `mean_grads()`.
            self.flat.scale_in_optimizer = True
            self.grad_scale = 1.0 / world_size()
        if self.flat is not None and self.range_hooks:
            # gradient exchange overlapped with backward, one range per (branch of the step, dtype class)
            names = {id(p): n for n, p in model.named_parameters()}
            self.flat.install_hooks(self.params, [_branch_of(names[id(p)]) for p in self.params])
        if self.flat is None and self._lo:
            self._hi_grads = [torch.zeros_like(m) for m in self._hi]
            for m, g in zip(self._hi, self._hi_grads):
                m.grad = g
        # torch's optimizer object always exists (learning-rate schedule, param_groups API); with the flat layout
        # its step() is replaced by one launch of ppea_adam_flat_f32 over the buffers built below
        self.optimizer = torch.optim.Adam(self.opt_params, lr, fused=True, capturable=on_gpu) if fused_adam else \
            torch.optim.Adam(self.opt_params, lr, foreach=True)
        if self.flat_adam:
            self._build_flat_state()
        self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, trainer.opt.scheduler_step_size, 0.1)
        self.graph = None
        self.static_inputs = None
        self.static_out = None
        # every step -- eager or captured -- runs on ONE non-default stream: autograd's AccumulateGrad
        # nodes remember the stream they were created on, and a captured backward must not touch the
        # legacy default stream.
        self.stream = torch.cuda.Stream(self.params[0].device) if on_gpu else None

    # ---- bf16 parameters with fp32 masters -----------------------------------------------------------
    # Under autocast every dense conv / linear casts its fp32 weight to bf16 on each use (~730 cast kernels
    # per step) and its bf16 weight gradient back to fp32 (~590).  Dense-conv / linear / deconv weights
    # and biases are therefore stored in bf16 (frozen ones once and for all); trainable ones keep an
    # fp32 master that Adam updates, and two multi-tensor copies per step move gradients up and weights
    # down.  BN affine parameters and depthwise filters stay fp32 (the HIP kernels read them as such).
    def _to_bf16_params(self, model):
        import torch.nn as nn
        self.masters = {}
        for m in model.modules():
            dense = isinstance(m, (nn.Linear, nn.ConvTranspose2d)) or (isinstance(m, nn.Conv2d) and m.groups == 1)
            if not dense:
                continue
            for p in (m.weight, m.bias):
                if p is None or p.dtype != torch.float32:
                    continue
                if p.requires_grad:
                    master = p.detach().clone()
                    master.requires_grad_(True)
                    self.masters[id(p)] = master
                p.data = p.data.to(torch.bfloat16)

    def export_state_dict(self):
        """state_dict with the fp32 masters substituted for the bf16 working copies (checkpoint format)."""
        model = self.trainer._module()
        sd = model.state_dict()
        if self.masters is not None:
            for name, p in model.named_parameters():
                if id(p) in self.masters:
                    sd[name] = self.masters[id(p)].detach().clone()
                elif p.dtype == torch.bfloat16:
                    sd[name] = p.detach().float()
        return sd

    # ---- flat optimizer state -------------------------------------------------------------------------
    # Every tensor Adam updates becomes a view into ONE fp32 buffer (bf16-backed ones first), their bf16 working
    # copies views into one bf16 buffer with the same element order: parameter update, moment update and the
    # master -> bf16 refresh are a single streaming kernel instead of ~90 multi-tensor launches.
    def _build_flat_state(self):
        dev = self.opt_params[0].device
        n = self.flat.numel
        self.P = torch.zeros(n, device=dev, dtype=torch.float32)
        self.M = torch.zeros(n, device=dev, dtype=torch.float32)
        self.V = torch.zeros(n, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for t, off in zip(self.opt_params, self.flat.offsets):
                view = self.P[off:off + t.numel()].view_as(t)
                view.copy_(t)
                t.data = view
            # the bf16-backed tensors come first, so their (aligned) offsets are valid in the bf16 buffer as well
            n_lo_t = len(self._lo)
            self.n_lo = (self.flat.offsets[n_lo_t] if n_lo_t < len(self.opt_params) else n) if n_lo_t else 0
            self.W16 = torch.zeros(max(self.n_lo, 1), device=dev, dtype=torch.bfloat16)
            for p, off in zip(self._lo, self.flat.offsets):
                view = self.W16[off:off + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view
        self.adam_state = torch.tensor([0.0, float(self.optimizer.param_groups[0]["lr"])], device=dev)

    def _flat_adam_step(self):
        from ._abi import call, ptr, stream_ptr
        g = self.optimizer.param_groups[0]
        with torch.no_grad():
            self.adam_state[0] += 1
        gscale = self.grad_scale
        call("ppea_adam_flat_scaled_f32", ptr(self.P), ptr(self.flat.flat), ptr(self.M), ptr(self.V),
             ptr(self.W16) if self.n_lo else None, self.flat.numel, self.n_lo, ptr(self.adam_state),
             float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), gscale, stream_ptr())

    # ---- learning-rate schedule --------------------------------------------------------------------------------
    # Reference: StepLR(optimizer, scheduler_step_size, 0.1) wrapped by `accelerator.prepare` (trainer.py:144, 153) and
    # stepped once per epoch (trainer.py:418).  Accelerate's AcceleratedScheduler (scheduler.py:69-82, same in the pinned
    # 0.18.0) steps the wrapped scheduler `num_processes` times per call unless split_batches is set -- so on 8 GPUs the
    # reference's learning rate drops every ceil(15 / 8) = 2 epochs, not every 15.  PRESERVED by default (a drop-in
    # replacement must train with the reference's schedule under the same launch); `lr_quirk=False` steps once per call
    # (the schedule the options describe).
    def scheduler_step(self, lr_quirk=None):
        if lr_quirk is None:
            lr_quirk = bool(getattr(self.trainer.opt, "lr_quirk", True))
        for _ in range(world_size() if lr_quirk else 1):
            self.scheduler.step()
        self.sync_lr()

    def describe_schedule(self):
        """One line for the start-up log: the learning-rate schedule this launch will actually follow."""
        quirk = bool(getattr(self.trainer.opt, "lr_quirk", True))
        w = world_size() if quirk else 1
        every = -(-self.trainer.opt.scheduler_step_size // w)
        return (f"StepLR x0.1 every {every} epoch(s) (scheduler_step_size {self.trainer.opt.scheduler_step_size}, {world_size()} rank(s), "
                f"lr_quirk={'on: Accelerate steps the schedule once per rank' if quirk else 'off'})")

    def sync_lr(self):
        """Push the optimizer's (scheduled) learning rate into the device scalar the step kernel reads."""
        if self.flat_adam:
            self.adam_state[1].fill_(float(self.optimizer.param_groups[0]["lr"]))

    # ---- whole-step hipGraph -------------------------------------------------------------------------
    # ~10k kernel launches per step make the eager step host-bound (Python + dispatcher ~15 us per
    # launch); the captured graph replays the same launches (ours and the library ones) from the GPU's
    # command processor.  Shapes are static; the only per-step host inputs -- the batch and the matching
    # augmentation draws -- are copied into static buffers before each replay.
    def capture(self, inputs, warmup=3, restore_state=False):
        """Warm up with `warmup` eager steps on a static copy of `inputs`, then record the step into a hipGraph.
        restore_state: put model / optimizer / tracker state back to what it was before the warm-up steps (in place,
        so the graph's addresses stay valid) -- the first replay is then step 1 of the run."""
        from . import rng
        dev = self.params[0].device
        snap = self.snapshot() if restore_state else None
        self.static_inputs = {k: v.clone() for k, v in inputs.items()}
        B = self.static_inputs[("color", 0, 0)].shape[0]
        # the augmentation-draw buffer and the recorded draw plan belong to THIS engine; they are installed in `rng` only
        # while this method runs (eager steps after a capture, and other engines, draw afresh)
        self._aug = torch.zeros(B, device=dev)
        self._rng_plan = None
        rng.set_aug_buffer(self._aug)
        reference_rng = rng.get_mode() == "reference"
        warmup = max(int(warmup), 1)
        import gc
        try:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                for i in range(warmup):
                    rng.refill_aug(self._aug)
                    if reference_rng and i == warmup - 1:
                        rng.static_begin_record()      # this step's host draws become the graph's static draw buffers
                    self._step_body(dict(self.static_inputs))
                    if reference_rng and i == warmup - 1:
                        self._rng_plan = rng.static_end_record()
            torch.cuda.synchronize()
            gc.collect()                        # drop autograd graphs of earlier steps before capturing
            self.graph = torch.cuda.CUDAGraph()
            rng.refill_aug(self._aug)
            torch.cuda.synchronize()
            # the dict the captured step saw: process_batch adds ("relative_pose", f) entries to it (repdepth.py:507)
            self.static_step_inputs = dict(self.static_inputs)
            with rng.serving(self._rng_plan):
                with torch.cuda.graph(self.graph, stream=self.stream):
                    outputs, losses = self._step_body(self.static_step_inputs)
        finally:
            rng.set_aug_buffer(None)
        self.static_out = (outputs, losses)
        if snap is not None:
            self.restore(snap)

    # ---- in-memory state snapshot (model, optimizer, depth-bin tracker) ----------------------------------
    def _state_tensors(self):
        model = self.trainer._module()
        ts = [v for v in model.state_dict().values()]
        if self.flat_adam:
            ts += [self.P, self.M, self.V, self.adam_state, self.flat.flat]
            if self.n_lo:
                ts.append(self.W16)
        tr = self.trainer.depth_bin_tracker
        ts += [tr.min_depth, tr.max_depth]
        return ts

    @torch.no_grad()
    def snapshot(self):
        if not self.flat_adam:
            raise NotImplementedError("snapshot/restore covers the flat-optimizer layout (the GPU path)")
        torch.cuda.synchronize()
        return [t.detach().clone() for t in self._state_tensors()], self.trainer.depth_bin_tracker.updated

    @torch.no_grad()
    def restore(self, snap):
        """Copy a snapshot back IN PLACE (addresses captured in the step graph stay valid)."""
        torch.cuda.synchronize()
        for t, s in zip(self._state_tensors(), snap[0]):
            t.copy_(s)
        self.trainer.depth_bin_tracker.updated = snap[1]
        torch.cuda.synchronize()

    def mean_grads(self):
        """name -> gradient of this step as DDP would leave it (mean over ranks), whatever the exchange stored."""
        return {n: (g * self.grad_scale if self.grad_scale != 1.0 else g) for n, g in self.named_grads().items()}

    def named_grads(self):
        """name -> fp32 gradient of this step for every trainable parameter (the flat buffer's views; the SUM over ranks
        when `flat.scale_in_optimizer` is set -- see `mean_grads`)."""
        model = self.trainer._module()
        names = {id(p): n for n, p in model.named_parameters()}
        if self.flat is not None:
            return {names[id(p)]: v for p, v in zip(self.params, self.flat.views)}
        return {names[id(p)]: (self.masters[id(p)].grad if self.masters and id(p) in self.masters else p.grad)
                for p in self.params}

    def replay(self, inputs=None):
        """Copy the new batch into the graph's static inputs and replay.  Everything -- the input copies, the
        augmentation draws and the graph -- is enqueued on the step stream, ordered after the caller's stream (which
        produced `inputs`); the caller's stream then waits for the step, so reading the returned tensors (or feeding
        the next batch) from it is ordered without a device-wide synchronize."""
        from . import rng
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            if inputs is not None:
                for k, v in inputs.items():
                    if k in self.static_inputs and v.data_ptr() != self.static_inputs[k].data_ptr():
                        self.static_inputs[k].copy_(v, non_blocking=True)
                        if v.is_cuda:
                            v.record_stream(self.stream)
            rng.refill_aug(self._aug)
            rng.refill_plan(self._rng_plan)  # reference-order DropPath / tie-break draws (None in device mode)
            self.graph.replay()
        cur.wait_stream(self.stream)
        self.trainer.step += 1
        return self.static_out

    def _step_body(self, inputs):
        outputs, losses = self.trainer.process_batch(inputs, is_train=True)
        for p in self.params:
            p.grad = None              # autograd then hands each gradient over without an accumulate kernel
        if self.flat is not None and self.flat.hooked:
            self.flat.begin_backward()
        losses["loss"].backward()
        self._optimizer_phase()
        return outputs, losses

    def _optimizer_phase(self):
        if self.flat is not None and self.flat.hooked:
            self.flat.finish()         # chunks were packed and all-reduced from the hooks, during backward
        elif self.flat is not None:    # pack -> (several ranks: few large all-reduces) -> Adam on the flat buffer
            self.flat.gather(self.params)
            self.flat.all_reduce_mean()
        elif self._lo:                 # bf16 working weights: bf16 grads -> fp32 master grads
            with torch.no_grad():
                torch._foreach_copy_(self._hi_grads, [p.grad if p.grad is not None else torch.zeros_like(p)
                                                      for p in self._lo])
        if self.flat_adam:
            self._flat_adam_step()     # also refreshes the bf16 working weights
        else:
            self.optimizer.step()
            if self._lo:
                with torch.no_grad():
                    torch._foreach_copy_(self._lo, self._hi)  # masters -> bf16 working weights

    def step(self, inputs):
        if self.graph is not None:
            return self.replay(inputs)
        if self.stream is None:
            outputs, losses = self._step_body(inputs)
        else:
            cur = torch.cuda.current_stream()
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                outputs, losses = self._step_body(inputs)
            cur.wait_stream(self.stream)          # results are consumed on the caller's stream
            for d in (outputs, losses):
                for v in d.values():
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(cur)
        self.trainer.step += 1
        return outputs, losses
