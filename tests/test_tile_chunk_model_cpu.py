"""CPU checks of the time chunks of recursion_tile_kernel (csrc/recursion_tile.hip) through their NumPy model
(scripts/dbg/tile_chunk_emul.py: which periods a chunk runs, its guesses, the private table of the extra periods, the boundary
states, the parts of the log-likelihood and of the EM sums) against the oracle.  The kernel itself is compared with the oracle in
tests/test_gpu_tile_chunk.py."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import info_form as inf
from oracle import kalman_oracle as ko

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("tile_chunk_emul", os.path.join(ROOT, "scripts", "dbg", "tile_chunk_emul.py"))
tc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tc)


def test_chunk_geometry():
    assert tc.chunk_geometry(256, 2000) == (2, 1000, 16)              # BASELINE config 4: two workgroups per CU
    assert tc.chunk_geometry(32, 2000)[:2] == (16, 126)               # a small batch: 16 chunks
    assert tc.chunk_geometry(3, 400)[:2] == (6, 68)                   # chunks are at least four warm-ups long
    assert tc.chunk_geometry(2, 40)[0] == 1 and tc.chunk_geometry(512, 2000)[0] == 1 and tc.chunk_geometry(4, 500, nc_req=1)[0] == 1
    for (B, T, nc, W) in [(1, 131, 0, 16), (5, 333, 7, 30), (9, 1000, 3, 8), (2, 77, 2, 2)]:
        n, lc, w = tc.chunk_geometry(B, T, nc, W)
        assert w % 2 == 0 and lc % 2 == 0 or n == 1
        if n > 1:
            assert lc >= 4 * w and T - (n - 1) * lc >= w + 2 and (n - 1) * lc < T


@pytest.mark.parametrize("N,T,r,miss,NC,W", [(60, 150, 6, 0.1, 2, 16), (40, 200, 4, 0.3, 3, 16), (80, 260, 5, 0.0, 4, 16), (50, 90, 3, 0.2, 2, 10)])
def test_chunks_reproduce_the_sequential_pass(N, T, r, miss, NC, W):
    x, p = ko.synth_replicate(3, N, T, r, seed=77, missing=miss)
    n, lc, w = tc.chunk_geometry(1, T, NC, W)
    assert n == NC
    got = tc.chunked_pass(x, p["Lam"], p["R"], p["A"], p["Q"], p["mu0"], p["P0"], n, lc, w)
    ref = inf.kfs_pass_info(x, p["Lam"], p["R"], p["A"], p["Q"], p["mu0"], p["P0"])
    assert not got["fail"] and got["worst"] < 1e-10
    np.testing.assert_allclose(got["loglik"], ref["loglik"], rtol=1e-11)
    assert np.abs(got["f_smooth"] - ref["f_smooth"]).max() <= 1e-9 * np.abs(ref["f_smooth"]).max()
    assert np.abs(got["P_smooth"] - ref["P_smooth"]).max() <= 1e-9 * np.abs(ref["P_smooth"]).max()
    assert np.abs(got["f0_smooth"] - ref["f0_smooth"]).max() <= 1e-9 and np.abs(got["P0_smooth"] - ref["P0_smooth"]).max() <= 1e-9
    np.testing.assert_allclose(got["S_P"], ref["P_smooth"].sum(axis=0), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(got["S_U"], ref["P_lag"].sum(axis=0), rtol=1e-9, atol=1e-12)


def test_a_filter_that_forgets_slowly_fails_the_boundary_check():
    """A near-unit-root factor seen through almost nothing: 16 periods of warm-up do not bring a guess to the exact state -- the check
    says so (the kernel then runs the replicate sequentially), and the chunked numbers are indeed off."""
    N, T, r = 30, 200, 3
    x, p = ko.synth_replicate(1, N, T, r, seed=5, missing=0.1)
    p = dict(p, Lam=p["Lam"] * 1e-3, R=np.ones(N), A=0.999 * np.eye(r), Q=1e-3 * np.eye(r))
    n, lc, w = tc.chunk_geometry(1, T, 2, 16)
    got = tc.chunked_pass(x, p["Lam"], p["R"], p["A"], p["Q"], p["mu0"], p["P0"], n, lc, w)
    ref = inf.kfs_pass_info(x, p["Lam"], p["R"], p["A"], p["Q"], p["mu0"], p["P0"])
    assert got["fail"] and got["worst"] > 1e-3
    assert np.abs(got["P_smooth"] - ref["P_smooth"]).max() > 1e-6 * np.abs(ref["P_smooth"]).max()
