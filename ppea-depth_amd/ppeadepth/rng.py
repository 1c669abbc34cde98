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


def bernoulli_keep(batch: int, keep_prob: float, like: torch.Tensor) -> torch.Tensor:
    """[batch,1,1,1] mask of 0/1 with P(1)=keep_prob, dtype/device of `like`."""
    if _MODE == "reference":
        m = torch.empty(batch, 1, 1, 1, dtype=torch.float32).bernoulli_(keep_prob)
        return m.to(device=like.device, dtype=like.dtype)
    return torch.empty(batch, 1, 1, 1, device=like.device, dtype=like.dtype).bernoulli_(keep_prob)


def randn_like_cpu_order(shape, device) -> torch.Tensor:
    if _MODE == "reference":
        return torch.randn(shape).to(device)
    return torch.randn(shape, device=device)


def uniform01() -> float:
    return random.random()


_AUG_BUFFER = None      # static device tensor [B] the captured step graph reads its draws from


def set_aug_buffer(t):
    global _AUG_BUFFER
    _AUG_BUFFER = t


def refill_aug_buffer():
    """Host side of the matching augmentation: B uniform draws (python `random`, like the reference) copied
    into the static buffer before the step graph is replayed."""
    if _AUG_BUFFER is not None:
        _AUG_BUFFER.copy_(torch.tensor([random.random() for _ in range(_AUG_BUFFER.shape[0])]),
                          non_blocking=True)


def aug_draws(batch: int, device) -> torch.Tensor:
    """One uniform draw per batch item (networks/repdepth.py:561-575) as a device tensor."""
    if _AUG_BUFFER is not None:
        return _AUG_BUFFER
    return torch.tensor([random.random() for _ in range(batch)], device=device)
