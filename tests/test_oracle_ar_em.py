"""CPU: the oracle of the joint estimation with AR idiosyncratic terms (oracle/ar_oracle.py em_ar).
PARITY UNPINNED by the reference (no Kalman / EM code there); pinned here by what an ECM algorithm must satisfy:
monotone observed-data log-likelihood, every conditional step a maximiser of its block of the expected complete-data
criterion, and q = 0 == the VAR(p) oracle's EM."""
import numpy as np
import pytest

from oracle import ar_oracle as ao
from oracle import varp_oracle as vo


@pytest.mark.parametrize("N,T,r,p,q,missing", [(12, 60, 2, 1, 1, 0.0), (15, 70, 2, 2, 2, 0.0), (14, 80, 3, 1, 2, 0.08),
                                               (10, 50, 1, 2, 4, 0.0)])
def test_em_ar_likelihood_is_monotone(N, T, r, p, q, missing):
    x, start = ao.synth_ar(0, N, T, r, p, q, missing=missing)
    _, path, _ = ao.em_ar(x, start, max_iter=12)
    assert np.all(np.isfinite(path))
    assert np.all(np.diff(path) >= -1e-8 * np.abs(path[:-1])), path


def test_em_ar_recovers_persistence():
    """Idiosyncratic AR(1) coefficients move from the flat start 0.1 towards the generating ones."""
    x, start = ao.synth_ar(3, 20, 300, 2, 1, 1)
    est, path, _ = ao.em_ar(x, start, max_iter=40)
    rng = np.random.default_rng([20160415, 3, 1, 1, 7])
    true = rng.uniform(-0.3, 0.6, 20)
    err0 = np.abs(0.1 - true).mean()
    err1 = np.abs(est["rho"][:, 0] - true).mean()
    assert err1 < 0.6 * err0, (err0, err1)


def test_em_ar_with_q0_equals_varp_em():
    N, T, r, p = 10, 40, 2, 2
    x = vo.synth_varp(1, N, T, r, p)
    start, _ = vo.varp_init(x, r, p)
    a_start = dict(Lam=start["Lam"], sig2=start["R"], rho=np.zeros((N, 0)), Avar=start["Avar"], Q=start["Q"],
                   mu0=start["mu0"], P0=start["P0"])
    e1, p1, _ = ao.em_ar(x, a_start, max_iter=5)
    e2, p2, _ = vo.em_varp(x, start, p, max_iter=5)
    np.testing.assert_allclose(p1, p2, rtol=1e-10)
    np.testing.assert_allclose(e1["Lam"], e2["Lam"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(e1["sig2"], e2["R"], rtol=1e-8)
    np.testing.assert_allclose(e1["Avar"], e2["Avar"], rtol=1e-7, atol=1e-10)


def test_cm_steps_are_block_maximisers():
    """Perturbing lam_i (rho fixed) or rho_i (lam fixed at its new value) away from the CM solution increases the
    expected sum of squares the step minimises."""
    N, T, r, p, q = 8, 60, 2, 1, 2
    x, start = ao.synth_ar(5, N, T, r, p, q)
    new, _, out = ao.em_step_ar(x, **start)
    m = ao.state_lags(p, q)
    Tq = T - q
    zs, Ps = out["f_smooth"], out["P_smooth"]
    Ez = (zs[:, :, None] * zs[:, None, :] + Ps).reshape(Tq, m, r, m, r)[:, :q + 1, :, :q + 1, :]
    zb = zs.reshape(Tq, m, r)[:, :q + 1]
    X = np.stack([x[q - l:T - l] for l in range(q + 1)], axis=2)

    def crit(i, lam, rho_i):
        a = np.concatenate([[1.0], -rho_i])
        xt = X[:, i] @ a
        g = np.einsum("l,tlc->tc", a, zb)
        EG = np.einsum("l,tlcmd,m->tcd", a, Ez, a)
        return float((xt ** 2).sum() - 2.0 * (xt[:, None] * g).sum(0) @ lam + np.einsum("c,tcd,d->", lam, EG, lam))

    rng = np.random.default_rng(0)
    for i in range(N):
        base_l = crit(i, new["Lam"][i], start["rho"][i])
        base_r = crit(i, new["Lam"][i], new["rho"][i])
        assert base_r <= base_l + 1e-9 * abs(base_l)
        for _ in range(5):
            assert crit(i, new["Lam"][i] + 1e-3 * rng.standard_normal(r), start["rho"][i]) >= base_l - 1e-9 * abs(base_l)
            assert crit(i, new["Lam"][i], new["rho"][i] + 1e-3 * rng.standard_normal(q)) >= base_r - 1e-9 * abs(base_r)
