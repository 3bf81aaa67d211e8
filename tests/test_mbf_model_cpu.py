"""CPU check of the algorithm behind csrc/recursion_mbf16.hip (round 6: the smoother pass of companion states -- VAR(p) factor dynamics,
singular innovation covariance -- without any k x k inversion): its NumPy model scripts/dbg/r06/mbf_emul.py -- rank-rc covariance-form
update through two rc x rc Cholesky factors, modified Bryson-Frazier backward recursion, lag-one covariances from the adjoint pair --
against the covariance-form oracle (oracle/varp_oracle.py -> kalman_oracle.kfs_pass: textbook filter + RTS smoother).  The kernel itself
is compared with the oracle in the GPU tests (tests/test_gpu_varp.py, tests/test_gpu_api.py)."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import kalman_oracle as ko
from oracle import varp_oracle as vo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("mbf_emul", os.path.join(ROOT, "scripts", "dbg", "r06", "mbf_emul.py"))
mbf = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mbf)


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("N,T,r,p,miss", [(139, 222, 4, 4, 0.1),      # the Stock-Watson window's shape, the model's own n_factorlag
                                          (30, 40, 3, 2, 0.3), (50, 60, 4, 4, 0.0),
                                          (20, 50, 2, 3, 0.6),        # most cells missing: periods with a singular or zero C_t
                                          (25, 30, 4, 1, 0.2)])       # p = 1: a non-singular Q takes the same path
def test_the_mbf_model_reproduces_the_covariance_form_oracle(N, T, r, p, miss):
    x = vo.synth_varp(5, N, T, r, p, missing=miss)
    if miss > 0.5:
        x[7] = np.nan                                                # a period without a single observed cell
    q, _ = vo.varp_init(np.nan_to_num(x), r, p)
    out = vo.kfs_pass_varp(x, q["Lam"], q["R"], q["Avar"], q["Q"], q["mu0"], q["P0"], p)
    b, s, n, ld, C = ko.collapse(x, q["Lam"], q["R"])
    M, Qk = vo.companion(q["Avar"], q["Q"], p)
    o = mbf.mbf_pass(b, C, s, n, ld, M, Qk, q["mu0"], q["P0"], r)
    assert abs(o["loglik"] - out["loglik"]) <= 1e-12 * abs(out["loglik"])
    for key in ("f_smooth", "P_smooth", "P_lag", "f0_smooth", "P0_smooth"):
        assert _rel(o[key], out[key]) <= 1e-11, key
