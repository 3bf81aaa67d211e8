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


def em_batch_sharded(ctx, panel, Lam, R, A, Q, mu0, P0, B_global: int, max_iter: int = 10, tol: float = 0.0,
                     want_smooth: bool = True, want_P: bool = True, may_have_missing=False, group=None,
                     iterate=None):
    """The EM loop of ONE rank of a replicate-sharded job (north_star: "the replicate batch shards
    embarrassingly across the 8 GPUs of one node with a single RCCL allgather at the end of each EM iteration").

    `panel`, `Lam`, .. `P0`: THIS rank's shard (replicates replicate_range(B_global, world, rank)), tensors on this
    rank's device; parameters are updated in place.  Every iteration runs `iterate` -- by default
    DfmContext.em_iterate_batch, i.e. dfm_em_iterate_batch_dev: E-step + M-step + per-replicate bookkeeping on the
    GPU -- then em_convergence_allgather exchanges {loglik_k, active} of every replicate of the job (one collective),
    and every rank stops at the same iteration, when no replicate anywhere is still iterating (tol > 0) or after
    max_iter iterations.  A replicate that converged keeps the parameters that entered its last iteration, exactly as
    dfm_em_batch_dev; a single rank (or no process group) reproduces dfm_em_batch_dev.

    Returns dict(path [shard, max_iter] (NaN past iters), iters [shard] int32, f, P, loglik_global [B_global,
    iterations] -- the gathered log-likelihoods, identical on every rank --, active_global [B_global] bool,
    iterations).  `iterate` is injectable so that the world-size-2 gloo test can drive this very function on CPU
    ranks with the oracle standing in for the kernel call (tests/ only)."""
    import torch
    if iterate is None:
        iterate = ctx.em_iterate_batch
    Bl, T, N = panel.shape
    r = Lam.shape[2]
    dev = panel.device
    path = torch.full((Bl, max_iter), float("nan"), dtype=torch.float64, device=dev)
    iters = torch.zeros((Bl,), dtype=torch.int32, device=dev)
    active = torch.ones((Bl,), dtype=torch.int32, device=dev)
    f = torch.empty((Bl, T, r), dtype=torch.float64, device=dev) if want_smooth else None
    P = torch.empty((Bl, T, r * (r + 1) // 2), dtype=torch.float64, device=dev) if (want_smooth and want_P) else None
    history = []
    gact = None
    ran = 0
    for k in range(max_iter):
        iterate(panel, Lam, R, A, Q, mu0, P0, k, max_iter, tol, path, iters, active, f, P,
                may_have_missing=may_have_missing)
        ran = k + 1
        gll, gact = em_convergence_allgather(path[:, k], active, B_global, group)   # THE collective of the iteration
        history.append(gll)
        if tol > 0.0 and not bool(gact.any().item()):
            break
    return dict(path=path, iters=iters, f=f, P=P, loglik_global=torch.stack(history, dim=1), active_global=gact,
                iterations=ran)
