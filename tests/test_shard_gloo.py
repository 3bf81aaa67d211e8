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
