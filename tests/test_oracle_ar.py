"""Pins oracle/ar_oracle.py (AR idiosyncratic terms by quasi-differencing): against the Gaussian density
built directly from the model's recursion, and against the VAR(p) oracle when q = 0."""
import numpy as np
import pytest

from oracle import ar_oracle as aro
from oracle import varp_oracle as vo


def _small(seed, N, T, r, p, q):
    rng = np.random.default_rng(seed)
    m = aro.state_lags(p, q)
    k = r * m
    Lam = rng.standard_normal((N, r)); sig2 = rng.uniform(.5, 1.5, N)
    rho = 0.5 * rng.uniform(-1, 1, (N, q)) / max(q, 1)
    Avar = 0.4 * rng.standard_normal((r, r * p)) / p
    G = rng.standard_normal((r, r)); Q = G @ G.T + 0.3 * np.eye(r)
    mu0 = rng.standard_normal(k)
    G0 = rng.standard_normal((k, k)); P0 = G0 @ G0.T + 0.5 * np.eye(k)
    x = rng.standard_normal((T, N))
    return x, dict(Lam=Lam, sig2=sig2, rho=rho, Avar=Avar, Q=Q, mu0=mu0, P0=P0)


@pytest.mark.parametrize("seed,N,T,r,p,q", [(0, 4, 7, 2, 1, 1), (1, 3, 8, 1, 2, 2), (2, 5, 7, 2, 3, 1), (3, 3, 9, 2, 1, 3)])
def test_quasi_differenced_pass_is_the_models_conditional_density(seed, N, T, r, p, q):
    x, a = _small(seed, N, T, r, p, q)
    o = aro.kfs_pass_ar(x, **a)
    ll, Ef = aro.direct_conditional_density(x, **a)
    assert abs(o["loglik"] - ll) <= 1e-9 * abs(ll)
    np.testing.assert_allclose(o["f_smooth"][:, :r], Ef, atol=1e-8)


def test_no_ar_lags_is_the_varp_model():
    x, a = _small(5, 6, 12, 2, 2, 0)
    x[3, 1] = np.nan
    o = aro.kfs_pass_ar(x, **a)
    v = vo.kfs_pass_varp(x, a["Lam"], a["sig2"], a["Avar"], a["Q"], a["mu0"], a["P0"], 2)
    assert abs(o["loglik"] - v["loglik"]) <= 1e-12 * abs(v["loglik"])
    np.testing.assert_allclose(o["f_smooth"], v["f_smooth"], atol=1e-12)


def test_missing_cells_propagate_to_their_lags():
    x, a = _small(6, 4, 10, 1, 1, 2)
    x[4, 2] = np.nan
    xt = aro.quasi_difference(x, a["rho"])
    assert np.isnan(xt[2:5, 2]).all() and np.isfinite(np.delete(xt, 2, axis=1)).all() and np.isfinite(xt[:2, 2]).all()
