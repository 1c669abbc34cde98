"""Random draws of the training step, injectable for parity runs.

The reference draws (a) one python `random.random()` per batch item for the matching
augmentation (networks/repdepth.py:561-575), (b) a per-sample Bernoulli mask in every
DropPath (timm), and (c) tie-break noise `torch.randn(...)` on the CPU
(trainer.py:1086-1087).  By default this build draws (b) and (c) on the device (no
host->device copies, no syncs).  `set_mode("reference")` reproduces the reference's
streams exactly -- CPU default generator, same shapes, same order -- so that runs seeded
like the reference give the same numbers; parity tests use it.
"""
import random

import torch

_MODE = "device"


def set_mode(mode: str) -> None:
    global _MODE
    if mode not in ("device", "reference"):
        raise ValueError(mode)
    _MODE = mode


def get_mode() -> str:
    return _MODE


# ---- reference-order draws under hipGraph replay ---------------------------------------------------------------
# A captured step cannot draw on the host.  In "reference" mode the engine therefore records, during one eager
# warm-up step, every host draw of the step (kind, arguments, a static device tensor) in call order: the PLAN.  The
# plan belongs to the engine that recorded it (`static_end_record()` hands it over and clears the module state); it is
# installed here only while that engine captures its graph (`serving(plan)`: the captured step reads the static
# tensors), and before every replay the engine calls `refill_plan(plan)`, which draws fresh values on the CPU default
# generator IN THE SAME ORDER (so a seeded run still sees the reference's stream) and copies them in.  Outside a
# capture every draw is a fresh host draw, whatever engines exist.
_STATIC = None          # list of [kind, args, device tensor] while recording or serving, else None
_RECORDING = False
_POS = 0


def static_begin_record():
    global _STATIC, _RECORDING, _POS
    _STATIC, _RECORDING, _POS = [], True, 0


def static_end_record():
    """-> the recorded plan (owned by the caller); nothing stays installed."""
    global _STATIC, _RECORDING, _POS
    plan, _STATIC, _RECORDING, _POS = _STATIC, None, False, 0
    return plan


def static_clear():
    global _STATIC, _RECORDING, _POS
    _STATIC, _RECORDING, _POS = None, False, 0


class serving:
    """with serving(plan): host draws are served from the plan's static tensors, in order from the first (the body is
    ONE step: the capture of a step graph)."""

    def __init__(self, plan):
        self.plan = plan

    def __enter__(self):
        global _STATIC, _RECORDING, _POS
        if self.plan is not None:
            _STATIC, _RECORDING, _POS = self.plan, False, 0
        return self

    def __exit__(self, *exc):
        if self.plan is not None:
            static_clear()
        return False


def _draw_cpu(kind, args):
    if kind == "bernoulli":
        return torch.empty(args[0], 1, 1, 1, dtype=torch.float32).bernoulli_(args[1])
    return torch.randn(args[0])


def refill_plan(plan):
    """Fresh reference-order draws into a recorded plan's static tensors (before a graph replay)."""
    if plan:
        for kind, args, dev_t in plan:
            dev_t.copy_(_draw_cpu(kind, args), non_blocking=True)


def _host_draw(kind, args, device):
    """One reference-order draw: eager -> CPU draw copied to the device; recording -> the same, kept as a static
    buffer; serving (inside `serving(plan)`) -> the next static buffer, no host work."""
    global _POS
    if _STATIC is not None and not _RECORDING:
        if _POS >= len(_STATIC):
            raise RuntimeError(f"rng: draw #{_POS} {kind}{args}: the recorded step had only {len(_STATIC)} draws")
        k, a, t = _STATIC[_POS]
        if (k, a) != (kind, args):
            raise RuntimeError(f"rng: draw #{_POS} is {kind}{args}, the recorded step had {k}{a}")
        _POS += 1
        return t
    t = _draw_cpu(kind, args).to(device)
    if _RECORDING:
        _STATIC.append([kind, args, t])
    return t


def bernoulli_keep(batch: int, keep_prob: float, like: torch.Tensor) -> torch.Tensor:
    """[batch,1,1,1] mask of 0/1 with P(1)=keep_prob, dtype/device of `like`."""
    if _MODE == "reference":
        return _host_draw("bernoulli", (int(batch), float(keep_prob)), like.device).to(like.dtype)
    return torch.empty(batch, 1, 1, 1, device=like.device, dtype=like.dtype).bernoulli_(keep_prob)


def drop_path_scales(keep: torch.Tensor, batch: int) -> torch.Tensor:
    """Device mode: the DropPath scales (0 or 1 / keep) of several blocks at once -- keep [blocks, 1] on the device ->
    [blocks, batch], ONE Bernoulli draw with per-block probabilities instead of one tiny launch per block."""
    return torch.bernoulli(keep.expand(-1, batch)) / keep


def randn_like_cpu_order(shape, device) -> torch.Tensor:
    if _MODE == "reference":
        return _host_draw("randn", (tuple(int(d) for d in shape),), device)
    return torch.randn(shape, device=device)


def uniform01() -> float:
    return random.random()


_AUG_BUFFER = None      # static device tensor [B] a step reads its matching-augmentation draws from, while installed


def set_aug_buffer(t):
    """Install (or, with None, remove) the static augmentation-draw buffer.  An engine installs its own buffer for the
    duration of its capture() only; outside, every step draws afresh."""
    global _AUG_BUFFER
    _AUG_BUFFER = t
    if t is None:
        static_clear()


def refill_aug(buffer):
    """Host side of the matching augmentation: B uniform draws (python `random`, like the reference) copied into a static
    buffer (before the step graph that reads it is replayed)."""
    if buffer is not None:
        buffer.copy_(torch.tensor([random.random() for _ in range(buffer.shape[0])]), non_blocking=True)


def refill_aug_buffer():
    refill_aug(_AUG_BUFFER)


def aug_draws(batch: int, device) -> torch.Tensor:
    """One uniform draw per batch item (networks/repdepth.py:561-575) as a device tensor."""
    if _AUG_BUFFER is not None and _AUG_BUFFER.shape[0] == batch:
        return _AUG_BUFFER
    return torch.tensor([random.random() for _ in range(batch)], device=device)
