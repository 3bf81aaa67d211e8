"""CPU check of the ALGORITHM of csrc/recursion_tile.hip through its lane-level NumPy model (scripts/dbg/tile_emul.py): the MFMA
operand layouts of v_mfma_f64_16x16x4, the "Y'X" products of matrices held as accumulator tiles, the 4 x 4 block-pivot sweep with
the pivot block published as -I and the LDL' solve, the partial sweep that leaves the identity padding alone, and the mean vectors
riding in padding column 31 -- against oracle/kalman_oracle.py.  The kernel itself is compared with the oracle in the GPU tests
(tests/test_gpu_round3.py config 4, tests/test_gpu_round4.py, tests/test_gpu_fuzz.py); this file keeps the model honest."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import kalman_oracle as ko

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("tile_emul", os.path.join(ROOT, "scripts", "dbg", "tile_emul.py"))
te = importlib.util.module_from_spec(spec)
spec.loader.exec_module(te)


@pytest.mark.parametrize("r", [17, 20, 23, 28, 31])
def test_block_pivot_sweep_inverts_the_leading_block_only(r):
    rng = np.random.default_rng(r)
    Mx = rng.standard_normal((r, r + 7))
    S = Mx @ Mx.T + r * np.eye(r)
    Z, det = te.sweep_inverse(te.to_tl(te.pad(S, r)), (r + 3) // 4)
    Zn = te.from_tl(Z)
    np.testing.assert_allclose(Zn[:r, :r], np.linalg.inv(S), rtol=1e-10, atol=1e-13)
    np.testing.assert_array_equal(Zn[r:, r:], np.eye(te.R - r))
    assert not Zn[:r, r:].any() and not Zn[r:, :r].any()
    np.testing.assert_allclose(det, np.linalg.det(S), rtol=1e-10)


def test_tile_products_are_transposed_left():
    rng = np.random.default_rng(0)
    Y, X = rng.standard_normal((32, 32)), rng.standard_normal((32, 32))
    np.testing.assert_allclose(te.from_tl(te.mm_tn(te.to_tl(Y), te.to_tl(X), 8)), Y.T @ X, rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("N,T,r,miss", [(40, 9, 20, 0.2), (25, 6, 17, 0.0), (50, 7, 31, 0.3), (30, 1, 24, 0.1), (30, 2, 18, 0.0)])
def test_the_model_reproduces_the_oracle_pass(N, T, r, miss):
    x, p = ko.synth_replicate(1, N, max(T, 6), r, missing=miss)     # (standardising a one-period panel divides by zero: cut a longer one)
    x = x[:T]
    out = ko.kfs_pass(x, p["Lam"], p["R"], p["A"], p["Q"], p["mu0"], p["P0"], lag_one=True)
    f, P, ll, em = te.tile_pass(x, p["Lam"], p["R"], p["A"], p["Q"], p["mu0"], p["P0"])
    np.testing.assert_allclose(ll, out["loglik"], rtol=1e-10)
    np.testing.assert_allclose(f, out["f_smooth"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(P, out["P_smooth"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(em["f0"], out["f0_smooth"], atol=1e-10)
    np.testing.assert_allclose(em["P0s"], out["P0_smooth"], atol=1e-10)
    np.testing.assert_allclose(em["SU"], out["P_lag"].sum(0), atol=1e-9)      # sum of the lag-one covariances (EM's S10)
    np.testing.assert_allclose(em["SP"], out["P_smooth"].sum(0), atol=1e-9)
