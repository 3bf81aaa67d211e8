"""The observed-factor oracle (oracle/obs_oracle.py; PARITY UNPINNED BY THE REFERENCE: `nfac_o > 0` is non-functional there,
SURVEY App. D 7) pinned by what the algorithm must satisfy."""
import numpy as np

from oracle import kalman_oracle as ko
from oracle import obs_oracle as oo

KEYS = ("Lam", "R", "A", "Q", "mu0", "P0")


def test_no_observed_factor_is_the_plain_em():
    x, p = ko.synth_replicate(3, 30, 50, 3, missing=0.1)
    G = np.zeros((50, 0))
    a, pa, _ = oo.em_obs(x, G, p, max_iter=4)
    b, pb, _ = ko.em(x, p, max_iter=4)
    np.testing.assert_allclose(pa, pb, rtol=1e-13)
    for k in KEYS:
        np.testing.assert_allclose(a[k], b[k], rtol=1e-12, atol=1e-14)


def test_likelihood_is_monotone_and_loadings_are_recovered():
    for miss in (0.0, 0.15):
        x, G, p = oo.synth_obs(11, 60, 220, 2, 2, missing=miss)
        start = {k: v.copy() for k, v in p.items()}
        rng = np.random.default_rng(0)
        start["Lam"] = p["Lam"] + 0.3 * rng.standard_normal(p["Lam"].shape)      # perturbed start
        est, path, _ = oo.em_obs(x, G, start, max_iter=40)
        assert np.all(np.diff(path) > -1e-8 * np.abs(path[:-1])), np.diff(path).min()
        # the observed block is identified (g is data): recovered up to sampling error
        err = np.abs(est["Lam"][:, :2] - p["Lam"][:, :2]).max()
        assert err < 0.35, err
        assert path[-1] > path[0]


def test_every_m_step_block_maximises_the_expected_complete_data_likelihood():
    """Q(theta) = E[log p(x, f | theta)] at the E-step's moments: the closed-form loadings / R of em_step_obs beat perturbations."""
    x, G, p = oo.synth_obs(5, 12, 80, 2, 1, missing=0.1)
    new, _, out = oo.em_step_obs(x, G, **p)
    fs, Ps = out["f_smooth"], out["P_smooth"]
    ro = 1
    z = np.hstack([G, fs])
    Ez = z[:, :, None] * z[:, None, :]
    Ez[:, ro:, ro:] += Ps
    obs = ~np.isnan(x)

    def q_series(i, lam, R):
        w = obs[:, i]
        xi = x[w, i]
        e2 = (xi ** 2).sum() - 2.0 * lam @ (xi @ z[w]) + lam @ Ez[w].sum(0) @ lam
        return -0.5 * (w.sum() * np.log(R) + e2 / R)

    rng = np.random.default_rng(1)
    for i in range(12):
        best = q_series(i, new["Lam"][i], new["R"][i])
        for _ in range(20):
            assert q_series(i, new["Lam"][i] + 1e-3 * rng.standard_normal(3), new["R"][i]) <= best + 1e-12
            assert q_series(i, new["Lam"][i], new["R"][i] * (1 + 1e-3 * rng.standard_normal())) <= best + 1e-12


def test_loglik_is_the_plain_models_on_the_residual_panel():
    x, G, p = oo.synth_obs(7, 20, 40, 3, 2, missing=0.05)
    ll = oo.loglik_obs(x, G, **p)
    y = x - G @ p["Lam"][:, :2].T
    ref = ko.brute_force_gaussian(y, p["Lam"][:, 2:], p["R"], p["A"], p["Q"], p["mu0"], p["P0"])["loglik"]
    np.testing.assert_allclose(ll, ref, rtol=1e-9)
