"""Replicate sharding across the GPUs of one node.

Replicates are independent units (own panel, own parameters, own EM trajectory): rank k of W owns the
contiguous block [k B / W, (k+1) B / W) and no data-path collective is needed.  The only exchange
north_star prescribes is one all-gather of the per-replicate {log-likelihood, still-iterating} pairs at
the end of every EM iteration, so that every rank sees the global convergence state.  torch.distributed
is plumbing here: backend "nccl" is RCCL over xGMI on the GPU box, "gloo" in the CPU tests.
"""
from __future__ import annotations

from typing import Tuple


def replicate_range(B: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of replicate indices owned by `rank` (sizes differ by at most one)."""
    if B < 0 or world < 1 or not (0 <= rank < world):
        raise ValueError("replicate_range: need B >= 0, world >= 1, 0 <= rank < world")
    return (B * rank) // world, (B * (rank + 1)) // world


def owner_of(b: int, B: int, world: int) -> int:
    """Rank that owns replicate b under replicate_range."""
    if not (0 <= b < B):
        raise ValueError("replicate index out of range")
    # smallest k with (B (k+1)) // world > b
    k = (b * world) // B
    while (B * (k + 1)) // world <= b:
        k += 1
    while (B * k) // world > b:
        k -= 1
    return k


def allgather_replicates(local, B: int, group=None):
    """All-gather a per-replicate tensor whose leading dimension is this rank's shard; returns the
    [B, ...] tensor in global replicate order on every rank (one collective; shards may differ by one
    replicate, so the shards are padded to the largest and trimmed)."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        if local.shape[0] != B:
            raise ValueError("allgather_replicates: not distributed but local shard != B")
        return local
    world = dist.get_world_size(group)
    sizes = [replicate_range(B, world, k)[1] - replicate_range(B, world, k)[0] for k in range(world)]
    rank = dist.get_rank(group)
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"allgather_replicates: shard has {local.shape[0]} replicates, expected {sizes[rank]}")
    mx = max(sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))], dim=0)
    out = local.new_empty((world * mx,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    parts = [out[k * mx:k * mx + sizes[k]] for k in range(world)]
    return torch.cat(parts, dim=0)


def em_convergence_allgather(loglik_local, active_local, B: int, group=None):
    """The per-EM-iteration exchange: every rank contributes {loglik, active} of its replicates and gets
    the global [B] vectors back (one all-gather of a [shard, 2] fp64 tensor)."""
    import torch
    pair = torch.stack([loglik_local.double(), active_local.double()], dim=1)
    full = allgather_replicates(pair, B, group)
    return full[:, 0], full[:, 1] != 0
