"""Batch sharding of the sampling loop over the GPUs of one node.

Trajectories are independent (no op of the U-Net couples samples), so the path shards by batch
with NO collective inside the denoise loop; one all-gather (RCCL over xGMI when the backend is
"nccl") of the final trajectories closes ``sample()``.  Rank r owns the contiguous global range
``shard_bounds(B, r, world)``; the device noise stream is keyed by the GLOBAL trajectory index
(``sample_offset``), so results do not depend on the rank count.  The reference has no
distributed code at all: this is the one collective the build adds.
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


def all_gather_trajectories(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """Rank-major concatenation of per-rank (b_r, ...) tensors into (total, ...).  Shards may be
    uneven: every rank pads to the largest shard for one fixed-size all_gather, then trims."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mx = (total + world - 1) // world
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    pieces = []
    for r in range(world):
        s, e = shard_bounds(total, r, world)
        pieces.append(out[r * mx: r * mx + (e - s)])
    del rank
    return torch.cat(pieces, dim=0)


def sample_sharded(sample_fn: Callable[..., torch.Tensor], cond: torch.Tensor, x_T: torch.Tensor,
                   noise: Optional[torch.Tensor] = None, inpaint: Optional[torch.Tensor] = None,
                   seed: int = 0, group=None) -> torch.Tensor:
    """Run ``sample_fn(cond, x_T, noise=, inpaint=, seed=, sample_offset=)`` (e.g.
    ``SpdmEngine.sample``) on this rank's slice of the GLOBAL batch and all-gather x_0.
    All ranks pass the same global tensors (or at least their own slice's rows)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = x_T.shape[0]
    s, e = shard_bounds(B, rank, world)
    ip = None
    if inpaint is not None:
        ip = inpaint if inpaint.shape[0] == 1 else inpaint[s:e]
    out = sample_fn(cond[s:e], x_T[s:e], noise=None if noise is None else noise[:, s:e], inpaint=ip,
                    seed=seed, sample_offset=s)
    if world == 1:
        return out
    return all_gather_trajectories(out, B, group)
