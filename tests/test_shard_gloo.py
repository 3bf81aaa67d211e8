"""World-size-2 (and 3) CPU tests of the replicate sharding + the per-EM-iteration all-gather (gloo)."""
import os

import numpy as np
import pytest

from dynamic_factor_models_amd import shard


def test_replicate_range_partitions():
    for B in (0, 1, 7, 8, 1024, 65536, 10001):
        for W in (1, 2, 3, 8):
            edges = [shard.replicate_range(B, W, k) for k in range(W)]
            assert edges[0][0] == 0 and edges[-1][1] == B
            assert all(edges[k][1] == edges[k + 1][0] for k in range(W - 1))
            sizes = [e[1] - e[0] for e in edges]
            assert max(sizes) - min(sizes) <= 1
    assert shard.replicate_range(65536, 8, 3) == (24576, 32768)      # BASELINE config 3: 8192 per GPU
    for B, W in ((10, 3), (7, 2), (1024, 8)):
        for b in range(B):
            k = shard.owner_of(b, B, W)
            lo, hi = shard.replicate_range(B, W, k)
            assert lo <= b < hi
    with pytest.raises(ValueError):
        shard.replicate_range(4, 2, 2)


def _worker(rank, world, port, B, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard.replicate_range(B, world, rank)
        idx = torch.arange(lo, hi, dtype=torch.float64)
        ll_local = -1000.0 - 3.0 * idx                     # a "log-likelihood" that identifies the replicate
        act_local = (torch.arange(lo, hi) % 3) != 0
        ll, act = shard.em_convergence_allgather(ll_local, act_local, B)
        f_local = torch.stack([idx, idx * idx], dim=1).reshape(hi - lo, 1, 2).repeat(1, 4, 1)   # [shard, T=4, r=2]
        f_all = shard.allgather_replicates(f_local, B)
        q.put((rank, ll.numpy().copy(), act.numpy().copy(), f_all.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 10), (2, 7), (3, 8)])
def test_allgather_over_gloo(world, B):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 7 * world + B
    procs = [ctx.Process(target=_worker, args=(k, world, port, B, q)) for k in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    idx = np.arange(B, dtype=float)
    for rank, ll, act, f_all in got:
        np.testing.assert_array_equal(ll, -1000.0 - 3.0 * idx)
        np.testing.assert_array_equal(act, (np.arange(B) % 3) != 0)
        assert f_all.shape == (B, 4, 2)
        np.testing.assert_array_equal(f_all[:, 0, 0], idx)
        np.testing.assert_array_equal(f_all[:, 3, 1], idx * idx)


def test_single_process_passthrough():
    import torch
    x = torch.arange(6.0)
    assert shard.allgather_replicates(x, 6) is x
    with pytest.raises(ValueError):
        shard.allgather_replicates(x, 7)


def _draw_worker(rank, world, port, ndraws, q):
    """Bootstrap draws are sharded like replicates; the one collective is the all-gather of the draws before the
    bands are taken (api.bootstrap_irf_bands(..., rank, world, gather))."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard.replicate_range(ndraws, world, rank)
        # stand-in for the device draws: a pure function of the GLOBAL draw index, as the Philox signs are
        g = np.arange(lo, hi, dtype=float)
        local = np.stack([np.sin(g), g * g], axis=1).reshape(hi - lo, 2, 1)
        full = shard.allgather_replicates(torch.from_numpy(local), ndraws).numpy()
        med = np.quantile(full, 0.5, axis=0, method="inverted_cdf")
        q.put((rank, full.copy(), med.copy()))
    finally:
        dist.destroy_process_group()


def test_bootstrap_draws_gather_over_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, ndraws = 2, 101
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_draw_worker, args=(k, world, port, ndraws, q)) for k in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = np.arange(ndraws, dtype=float)
    want = np.stack([np.sin(g), g * g], axis=1).reshape(ndraws, 2, 1)
    for rank, full, med in got:
        np.testing.assert_array_equal(full, want)                       # every rank: all draws, global order
        np.testing.assert_array_equal(med, np.quantile(want, 0.5, axis=0, method="inverted_cdf"))


# ---- the multi-GPU EM driver itself (shard.em_batch_sharded) on CPU ranks -------------------------------------
# The kernel call (DfmContext.em_iterate_batch = dfm_em_iterate_batch_dev) is replaced by the oracle's EM step with
# the same bookkeeping contract (include/dfm_hip.h): tests only.  What is under test is the driver: sharding, the
# all-gather after EVERY iteration, the global stopping rule, rank-identical global state, results equal to a
# single-rank run.
EM_KEYS = ("Lam", "R", "A", "Q", "mu0", "P0")


def _oracle_iterate(panel, Lam, R, A, Q, mu0, P0, k, max_iter, tol, path, iters, active, f, P, may_have_missing=False):
    from oracle import kalman_oracle as ko
    import torch
    for b in range(panel.shape[0]):
        was = True if k == 0 else bool(active[b])
        par = dict(Lam=Lam[b].numpy(), R=R[b].numpy(), A=A[b].numpy(), Q=Q[b].numpy(), mu0=mu0[b].numpy(), P0=P0[b].numpy())
        new, ll, out = ko.em_step(panel[b].numpy(), **par)
        go = was
        if was and k >= 1 and tol > 0.0:
            llp = float(path[b, k - 1])
            go = not ((ll - llp) / (0.5 * (abs(ll) + abs(llp))) < tol)
        if was:
            path[b, k] = ll
            iters[b] = k + 1
        active[b] = 1 if go else 0
        if go:
            for name, t in zip(EM_KEYS, (Lam, R, A, Q, mu0, P0)):
                t[b] = torch.from_numpy(np.ascontiguousarray(new[name]))
        if f is not None:
            f[b] = torch.from_numpy(out["f_smooth"])
        if P is not None:
            P[b] = torch.from_numpy(ko.pack_sym(out["P_smooth"]))


def _em_problem(B, N, T, r):
    from oracle import kalman_oracle as ko
    panels, starts = [], []
    for b in range(B):
        x, _ = ko.synth_replicate(b, N, T, r, missing=0.05 if b % 2 else 0.0)
        p0, _ = ko.pca_init(np.nan_to_num(x), r)
        panels.append(x); starts.append(p0)
    return np.stack(panels), {k: np.stack([s[k] for s in starts]) for k in EM_KEYS}


def _em_worker(rank, world, port, B, max_iter, tol, q):
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        panel, st = _em_problem(B, 12, 30, 2)
        lo, hi = shard.replicate_range(B, world, rank)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a[lo:hi])).clone()
        par = [t(st[k]) for k in EM_KEYS]
        out = shard.em_batch_sharded(None, t(panel), *par, B_global=B, max_iter=max_iter, tol=tol, iterate=_oracle_iterate)
        q.put((rank, lo, hi, out["iterations"], out["loglik_global"].numpy().copy(), out["active_global"].numpy().copy(),
               out["path"].numpy().copy(), out["iters"].numpy().copy(), [p.numpy().copy() for p in par],
               out["f"].numpy().copy()))
    finally:
        if world > 1:
            dist.destroy_process_group()


def _run_em(world, B, max_iter, tol):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + 11 * world
    procs = [ctx.Process(target=_em_worker, args=(k, world, port, B, max_iter, tol, q)) for k in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda g: g[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize("tol", [0.0, 2e-3])
def test_em_driver_world2_equals_single_rank(tol):
    B, max_iter = 5, 7
    one = _run_em(1, B, max_iter, tol)[0]
    two = _run_em(2, B, max_iter, tol)
    # rank-identical GLOBAL state after every iteration: the gathered {loglik, active} and the stopping iteration
    assert two[0][3] == two[1][3] == one[3]
    np.testing.assert_array_equal(two[0][4], two[1][4])
    np.testing.assert_array_equal(two[0][5], two[1][5])
    np.testing.assert_array_equal(two[0][4], one[4])            # ... and equal to the single-rank run
    np.testing.assert_array_equal(two[0][5], one[5])
    if tol > 0.0:
        assert one[3] < max_iter and not one[5].any()           # stopped globally, before max_iter
        assert len(set(one[7].tolist())) > 1                    # replicates converge at different iterations
    else:
        assert one[3] == max_iter
    # per-replicate results of the shards = the single-rank results (paths, iteration counts, parameters, factors)
    for rank, lo, hi, _, _, _, path, iters, par, f in two:
        np.testing.assert_array_equal(path, one[6][lo:hi])
        np.testing.assert_array_equal(iters, one[7][lo:hi])
        for a, b in zip(par, one[8]):
            np.testing.assert_array_equal(a, b[lo:hi])
        np.testing.assert_array_equal(f, one[9][lo:hi])
    # the gathered column k is the log-likelihood path of every replicate still iterating at k
    for b in range(B):
        n = one[7][b]
        np.testing.assert_array_equal(one[4][b, :min(n, one[3])], one[6][b, :min(n, one[3])])
