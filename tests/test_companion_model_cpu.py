"""CPU check of the algorithm behind csrc/recursion_comp.hip (round 6: companion states in information form as a block elimination of the
block-banded posterior precision -- 4 x 4 pivots only): its NumPy model scripts/dbg/r06/companion_emul.py against the covariance-form
oracles (oracle/varp_oracle.py: VAR(p) factor dynamics; oracle/ar_oracle.py: AR idiosyncratic terms, whose quasi-differenced observation
loads on every block of the state).  The kernel itself is compared with the oracles in the GPU tests (tests/test_gpu_varp.py,
tests/test_gpu_ar.py, tests/test_gpu_api.py)."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import ar_oracle as aro
from oracle import kalman_oracle as ko
from oracle import varp_oracle as vo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("companion_emul", os.path.join(ROOT, "scripts", "dbg", "r06", "companion_emul.py"))
ce = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ce)


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("N,T,r,p,miss", [(139, 222, 4, 4, 0.1), (30, 40, 3, 2, 0.3), (20, 50, 2, 3, 0.6), (40, 60, 4, 7, 0.05)])
def test_block_elimination_reproduces_the_varp_oracle(N, T, r, p, miss):
    x = vo.synth_varp(5, N, T, r, p, missing=miss)
    q, _ = vo.varp_init(np.nan_to_num(x), r, p)
    out = vo.kfs_pass_varp(x, q["Lam"], q["R"], q["Avar"], q["Q"], q["mu0"], q["P0"], p)
    k = r * p
    LamK = np.zeros((N, k)); LamK[:, :r] = q["Lam"]
    b, s, n, ld, C = ko.collapse(x, LamK, q["R"])
    o = ce.companion_pass(b, C, s, n, ld, q["Avar"], q["Q"], q["mu0"], q["P0"])
    assert abs(o["loglik"] - out["loglik"]) <= 1e-12 * abs(out["loglik"])
    for key in ("f_smooth", "P_smooth", "P_lag", "f0_smooth", "P0_smooth"):
        assert _rel(o[key], out[key]) <= 1e-11, key


@pytest.mark.parametrize("N,T,r,p,q,miss", [(139, 222, 4, 4, 4, 0.1), (30, 60, 2, 1, 2, 0.2), (25, 50, 3, 2, 1, 0.0)])
def test_block_elimination_reproduces_the_ar_oracle(N, T, r, p, q, miss):
    x, st = aro.synth_ar(3, N, T, r, p, q, missing=miss)
    out = aro.kfs_pass_ar(x, st["Lam"], st["sig2"], st["rho"], st["Avar"], st["Q"], st["mu0"], st["P0"])
    m = aro.state_lags(p, q); k = r * m
    xt = aro.quasi_difference(x, st["rho"]); LamK = aro.ar_loadings(st["Lam"], st["rho"], m)
    b, s, n, ld, C = ko.collapse(xt, LamK, st["sig2"])
    Phi = np.zeros((r, k)); Phi[:, :r * p] = st["Avar"]
    o = ce.companion_pass(b, C, s, n, ld, Phi, st["Q"], st["mu0"], st["P0"])
    assert abs(o["loglik"] - out["loglik"]) <= 1e-12 * abs(out["loglik"])
    for key in ("f_smooth", "P_smooth", "f0_smooth", "P0_smooth"):
        assert _rel(o[key], out[key]) <= 1e-11, key
