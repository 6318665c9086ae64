"""Batch sharding of the sampling loop over the GPUs of one node.

Trajectories are independent (no op of the U-Net couples samples), so the path shards by batch
with NO collective inside the denoise loop; one all-gather (RCCL over xGMI when the backend is
"nccl") of the final trajectories closes ``sample()``.  Rank r owns the contiguous global range
``shard_bounds(B, r, world)``; the device noise stream is keyed by the GLOBAL trajectory index
(``sample_offset``), so results do not depend on the rank count.  The reference has no
distributed code at all: this is the one collective the build adds.

Entry points, from the drop-in surface downwards:
  * ``Diffusion_DDPM.sample(batch, batched=True)`` shards by itself when a process group is
    initialised (``sharded=None``: auto; ``True`` / ``False``: explicit) -- diffusion.py;
  * ``ShardedSampler(engine)``: begin / run / result over this rank's shard (what ``bench.py``
    times step ranges of), ``sample()`` = all three;
  * ``sample_sharded(sample_fn, ...)``: the same split + gather around any per-shard callable.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split: the first ``total % world`` ranks get one extra trajectory."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def world_and_rank(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def all_gather_trajectories(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """Rank-major concatenation of per-rank (b_r, ...) tensors into (total, ...).  Shards may be
    uneven: every rank pads to the largest shard for one fixed-size all_gather, then trims."""
    world, _ = world_and_rank(group)
    if world == 1:
        return local
    mx = (total + world - 1) // world
    if local.shape[0] == mx:
        pad = local.contiguous()
    else:
        pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    if total == world * mx:
        return out
    pieces = []
    for r in range(world):
        s, e = shard_bounds(total, r, world)
        pieces.append(out[r * mx: r * mx + (e - s)])
    return torch.cat(pieces, dim=0)


def shard_inputs(cond: Optional[torch.Tensor], x_T: torch.Tensor, noise: Optional[torch.Tensor] = None,
                 inpaint: Optional[torch.Tensor] = None, group=None):
    """This rank's rows of the GLOBAL inputs: (cond, x_T, noise, inpaint, first global trajectory index).
    ``inpaint`` of leading size 1 is the broadcast form and stays whole."""
    world, rank = world_and_rank(group)
    s, e = shard_bounds(x_T.shape[0], rank, world)
    ip = None
    if inpaint is not None:
        ip = inpaint if inpaint.shape[0] == 1 else inpaint[s:e]
    return (None if cond is None else cond[s:e], x_T[s:e], None if noise is None else noise[:, s:e], ip, s)


def sample_sharded(sample_fn: Callable[..., torch.Tensor], cond: torch.Tensor, x_T: torch.Tensor,
                   noise: Optional[torch.Tensor] = None, inpaint: Optional[torch.Tensor] = None,
                   seed: int = 0, group=None) -> torch.Tensor:
    """Run ``sample_fn(cond, x_T, noise=, inpaint=, seed=, sample_offset=)`` (e.g.
    ``SpdmEngine.sample``) on this rank's slice of the GLOBAL batch and all-gather x_0.
    All ranks pass the same global tensors (or at least their own slice's rows)."""
    c, x, n, ip, first = shard_inputs(cond, x_T, noise, inpaint, group)
    out = sample_fn(c, x, noise=n, inpaint=ip, seed=seed, sample_offset=first)
    return all_gather_trajectories(out, x_T.shape[0], group)


class ShardedSampler:
    """The sampling loop of one engine (``SpdmEngine`` interface: sample_begin / sample_run / sample_result) over this
    rank's shard of a global batch.  ``begin`` takes GLOBAL tensors, ``run(a, b)`` executes loop iterations [a, b) on the
    shard (no communication), ``result`` all-gathers the current iterates of every rank, rank-major."""

    def __init__(self, engine, group=None):
        self.engine, self.group = engine, group
        self.total = 0
        self._hist = None

    def begin(self, cond, x_T, noise=None, inpaint=None, seed: int = 0, history: bool = False):
        c, x, n, ip, first = shard_inputs(cond, x_T, noise, inpaint, self.group)
        self.total = x_T.shape[0]
        kw = {"history": True} if history else {}
        self._hist = self.engine.sample_begin(c, x, noise=n, inpaint=ip, seed=seed, sample_offset=first, **kw)
        return first

    def run(self, step_begin: int, step_end: int) -> None:
        self.engine.sample_run(step_begin, step_end)

    def result(self) -> torch.Tensor:
        return all_gather_trajectories(self.engine.sample_result(), self.total, self.group)

    def history(self) -> Optional[torch.Tensor]:
        """(n_steps + 1, B_total, ...) stack of all ranks' iterates (only after ``begin(history=True)``)."""
        if self._hist is None:
            return None
        h = self._hist.transpose(0, 1).contiguous()                     # trajectory-major for the gather
        return all_gather_trajectories(h, self.total, self.group).transpose(0, 1).contiguous()

    def sample(self, cond, x_T, noise=None, inpaint=None, seed: int = 0, history: bool = False, check_finite: bool = True):
        self.begin(cond, x_T, noise, inpaint, seed, history)
        self.run(0, self.engine.n_steps)
        out = self.result()
        if check_finite and hasattr(self.engine, "nonfinite") and self.engine.nonfinite():
            raise FloatingPointError("non-finite iterate in the sampling loop (this rank's shard)")
        return (out, self.history()) if history else out
