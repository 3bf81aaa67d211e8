"""Pins oracle/varp_oracle.py (VAR(p) factor dynamics in companion form): brute-force Gaussian
conditioning with a singular state innovation covariance, p = 1 == the VAR(1) oracle, EM monotonicity."""
import numpy as np
import pytest

from oracle import kalman_oracle as ko
from oracle import varp_oracle as vo


def _small(seed, N, T, r, p, miss):
    rng = np.random.default_rng(seed)
    k = r * p
    Lam = rng.standard_normal((N, r)); R = rng.uniform(.5, 1.5, N)
    Avar = 0.3 * rng.standard_normal((r, k)) / p
    G = rng.standard_normal((r, r)); Q = G @ G.T + 0.3 * np.eye(r)
    mu0 = rng.standard_normal(k)
    G0 = rng.standard_normal((k, k)); P0 = G0 @ G0.T + 0.5 * np.eye(k)
    x = rng.standard_normal((T, N))
    x[rng.random((T, N)) < miss] = np.nan
    return x, dict(Lam=Lam, R=R, Avar=Avar, Q=Q, mu0=mu0, P0=P0)


@pytest.mark.parametrize("seed,N,T,r,p,miss", [(0, 5, 6, 2, 2, .2), (1, 4, 5, 1, 3, 0.), (2, 6, 5, 2, 3, .3)])
def test_companion_pass_equals_brute_force(seed, N, T, r, p, miss):
    x, q = _small(seed, N, T, r, p, miss)
    a = vo.kfs_pass_varp(x, p=p, **q)
    M, Qk = vo.companion(q["Avar"], q["Q"], p)
    LamK = np.zeros((N, r * p)); LamK[:, :r] = q["Lam"]
    b = ko.brute_force_gaussian(x, LamK, q["R"], M, Qk, q["mu0"], q["P0"])
    assert abs(a["loglik"] - b["loglik"]) <= 1e-10 * abs(b["loglik"])
    for k in ("f_smooth", "P_smooth", "f0_smooth", "P0_smooth", "P_lag"):
        np.testing.assert_allclose(a[k], b[k], rtol=0, atol=1e-9, err_msg=k)


def test_one_lag_is_the_var1_oracle():
    x, p = ko.synth_replicate(1, 20, 40, 3, missing=0.1)
    a = ko.em(x, p, 3)
    q = dict(Lam=p["Lam"], R=p["R"], Avar=p["A"], Q=p["Q"], mu0=p["mu0"], P0=p["P0"])
    b = vo.em_varp(x, q, 1, 3)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-12)
    for k, kk in (("Lam", "Lam"), ("R", "R"), ("A", "Avar"), ("Q", "Q"), ("mu0", "mu0"), ("P0", "P0")):
        np.testing.assert_allclose(a[0][k], b[0][kk], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("miss", [0.0, 0.1])
def test_em_varp_monotone_and_keeps_structure(miss):
    x = vo.synth_varp(0, 24, 90, 2, 3, missing=miss)
    q0, _ = vo.varp_init(np.nan_to_num(x), 2, 3)
    q, path, out = vo.em_varp(x, q0, 3, 15)
    assert np.all(np.diff(path) > -1e-8 * np.abs(path[:-1])) and path[-1] > path[0]
    # the lagged blocks of the smoothed companion state are the smoothed earlier factors
    z = out["f_smooth"]
    np.testing.assert_allclose(z[1:, 2:4], z[:-1, 0:2], atol=1e-9)
