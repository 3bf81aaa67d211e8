"""GPU test of the reference-interface mirror: estimate(m, Parametric()) on an unbalanced panel against the
same pipeline assembled from the CPU oracle (PCA start -> complete-case OLS for gappy series -> EM)."""
import numpy as np
import pytest

from oracle import kalman_oracle as ko

pytestmark = pytest.mark.gpu


def test_estimate_parametric_matches_oracle_pipeline():
    from dynamic_factor_models_amd import api
    rng = np.random.default_rng(5)
    T_all, ns, r = 120, 30, 3
    x, _ = ko.synth_replicate(11, ns, T_all, r)
    raw = 3.0 + x * rng.uniform(0.5, 2.0, ns)                 # un-standardised data with level and scale
    raw[rng.random(raw.shape) < 0.04] = np.nan
    raw[:, :18][np.isnan(raw[:, :18])] = 1.0                  # 18 fully observed series
    inclcode = np.ones(ns, dtype=int); inclcode[[4, 9]] = 0
    init, last = 3, 118
    m = api.DFMModel(raw, inclcode, 20, 40, init, last, 0, r, 1e-8, 4, 4)
    path = api.estimate(m, api.Parametric(), max_em_iter=8, tol_em=0.0, factor_lags=1)

    # oracle pipeline
    z, sd = api.standardize_data(raw[init - 1:last][:, inclcode == 1])
    xbal, bal = api.drop_missing_col(z)
    p0, F0 = ko.pca_init(xbal, r)
    N = z.shape[1]
    Lam = np.empty((N, r)); R = np.empty(N)
    Lam[bal] = p0["Lam"]; R[bal] = p0["R"]
    for i in np.nonzero(~bal)[0]:
        ok = ~np.isnan(z[:, i])
        b = np.linalg.lstsq(F0[ok], z[ok, i], rcond=None)[0]
        e = z[ok, i] - F0[ok] @ b
        Lam[i] = b; R[i] = e @ e / ok.sum()
    start = dict(Lam=Lam, R=R, A=p0["A"], Q=p0["Q"], mu0=p0["mu0"], P0=p0["P0"])
    pe, pathe, out = ko.em(z, start, max_iter=8, tol=0.0)

    np.testing.assert_allclose(path, pathe, rtol=1e-7)
    f = m.factor[init - 1:last]
    assert np.isnan(m.factor[:init - 1]).all() and np.isnan(m.factor[last:]).all()
    np.testing.assert_allclose(f, out["f_smooth"], atol=1e-6 * np.abs(out["f_smooth"]).max())
    cols = np.nonzero(inclcode == 1)[0]
    # the path returns the parameters entering the last E-step run (oracle em(): `p` after max_iter-1 M-steps ...)
    assert m.em_iters == 8
    np.testing.assert_allclose(m.lambda_[cols] / sd[0][:, None], m.em_params["Lam"], rtol=1e-12)
    assert np.isnan(m.lambda_[[4, 9]]).all()
    np.testing.assert_allclose(m.factor_var_model.seps, m.em_params["Q"])
    np.testing.assert_allclose(m.factor_var_model.G[:r, :r] @ m.factor_var_model.G[:r, :r].T, m.em_params["Q"], rtol=1e-10)
    assert 0 < m.fes.ssr < m.fes.tss and m.fes.nobs == int((~np.isnan(z)).sum())
    assert np.all(np.diff(path) > -1e-8 * np.abs(path[:-1]))   # EM monotone


def test_estimate_parametric_with_the_models_factor_lags():
    """Default factor_lags = m.n_factorlag (4): the companion-form EM (dfm_em_varp_batch) against the oracle pipeline."""
    from dynamic_factor_models_amd import api
    from oracle import varp_oracle as vo
    rng = np.random.default_rng(8)
    T_all, ns, r, p = 130, 28, 2, 4
    x = vo.synth_varp(3, ns, T_all, r, p)
    raw = 1.0 + x * rng.uniform(0.5, 2.0, ns)
    raw[rng.random(raw.shape) < 0.05] = np.nan
    raw[:, :16][np.isnan(raw[:, :16])] = 0.5
    inclcode = np.ones(ns, dtype=int); inclcode[3] = 0
    init, last = 2, 128
    m = api.DFMModel(raw, inclcode, 20, 40, init, last, 0, r, 1e-8, 4, p)
    path = api.estimate(m, api.Parametric(), max_em_iter=6, tol_em=0.0)

    z, sd = api.standardize_data(raw[init - 1:last][:, inclcode == 1])
    xbal, bal = api.drop_missing_col(z)
    q0, F0 = vo.varp_init(xbal, r, p)
    N = z.shape[1]
    Lam = np.empty((N, r)); R = np.empty(N)
    Lam[bal] = q0["Lam"]; R[bal] = q0["R"]
    for i in np.nonzero(~bal)[0]:
        ok = ~np.isnan(z[:, i])
        b = np.linalg.lstsq(F0[ok], z[ok, i], rcond=None)[0]
        e = z[ok, i] - F0[ok] @ b
        Lam[i] = b; R[i] = e @ e / ok.sum()
    start = dict(q0); start["Lam"] = Lam; start["R"] = R
    qe, pathe, out = vo.em_varp(z, start, p, 6)
    np.testing.assert_allclose(path, pathe, rtol=1e-7)
    f = m.factor[init - 1:last]
    np.testing.assert_allclose(f, out["f_smooth"][:, :r], atol=1e-6 * np.abs(out["f_smooth"]).max())
    var = m.factor_var_model
    np.testing.assert_allclose(var.M[:r], m.em_params["Avar"], rtol=1e-12)
    np.testing.assert_allclose(var.M[r:, :-r], np.eye(r * (p - 1)))
    np.testing.assert_allclose(var.seps, m.em_params["Q"], rtol=1e-12)


C1_TOL = 1e-10      # attained on MI355X (round 6, printed below): log-likelihood path 3.4e-15, smoothed factors 2.3e-14 (VAR(1)) / 3.5e-14 (VAR(4))


@pytest.mark.parametrize("lags", [1, 4])
def test_config1_stock_watson_panel_pca_plus_10_em_iterations(lags):
    """BASELINE configs[0]: the Stock-Watson panel, r = 4, PCA start + 10 EM iterations through estimate(m, Parametric())
    -- unbalanced real data (the :All window: 222 periods, 139 included series) -- against the oracle pipeline, with
    VAR(1) factor dynamics and with the model's own n_factorlag = 4 (companion state 16)."""
    import os
    from dynamic_factor_models_amd import api
    from oracle import varp_oracle as vo
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sw_panel.npz"))
    r, init, last = 4, 3, 224
    m = api.DFMModel(d["bpdata"], d["inclcode"], 20, 40, init, last, 0, r, 1e-8, 4, 4)
    path = api.estimate(m, api.Parametric(), max_em_iter=10, tol_em=0.0, factor_lags=lags)

    incl = d["inclcode"] == 1
    z, sd = api.standardize_data(d["bpdata"][init - 1:last][:, incl])
    enough = (~np.isnan(z)).sum(axis=0) >= 20
    z = z[:, enough]
    xbal, bal = api.drop_missing_col(z)
    q0, F0 = vo.varp_init(xbal, r, lags)
    N = z.shape[1]
    Lam = np.empty((N, r)); R = np.empty(N)
    Lam[bal] = q0["Lam"]; R[bal] = q0["R"]
    for i in np.nonzero(~bal)[0]:
        ok = ~np.isnan(z[:, i])
        b = np.linalg.lstsq(F0[ok], z[ok, i], rcond=None)[0]
        e = z[ok, i] - F0[ok] @ b
        Lam[i] = b; R[i] = e @ e / ok.sum()
    if lags == 1:
        p0, _ = ko.pca_init(xbal, r)
        start = dict(Lam=Lam, R=R, A=p0["A"], Q=p0["Q"], mu0=p0["mu0"], P0=p0["P0"])
        qe, pathe, out = ko.em(z, start, max_iter=10, tol=0.0)
        fo = out["f_smooth"]
    else:
        start = dict(q0); start["Lam"] = Lam; start["R"] = R
        qe, pathe, out = vo.em_varp(z, start, lags, 10)
        fo = out["f_smooth"][:, :r]
    assert m.em_iters == 10 and np.all(np.diff(pathe) > 0)
    f = m.factor[init - 1:last]
    err_path = float(np.max(np.abs(path - pathe) / np.abs(pathe)))
    err_f = float(np.abs(f - fo).max() / np.abs(fo).max())
    print(f"config 1, factor VAR({lags}): attained loglik-path error {err_path:.2e}, factor error {err_f:.2e}")
    assert err_path <= C1_TOL and err_f <= C1_TOL, (err_path, err_f)      # (north_star asks 1e-6 relative)


def test_estimate_parametric_with_bootstrap_replicates_and_short_series():
    """`estimate(m, Parametric(); nrep, seed, ngpu)`: replicates drawn from the fitted model and re-estimated
    in one dfm_em_batch_multi call, each against the oracle's EM on the same replicate panel from the same start; a series
    with fewer than nt_min observed periods is left out (dfm_functions.ipynb:357) and gets no loadings."""
    from dynamic_factor_models_amd import api
    rng = np.random.default_rng(2)
    T_all, ns, r = 90, 20, 2
    x, _ = ko.synth_replicate(4, ns, T_all, r)
    raw = 1.0 + 2.0 * x
    raw[rng.random(raw.shape) < 0.05] = np.nan
    raw[:, :12][np.isnan(raw[:, :12])] = 0.5
    raw[:75, 17] = np.nan                                       # 15 observed periods < nt_min = 20
    m = api.DFMModel(raw, np.ones(ns, dtype=int), 20, 40, 1, T_all, 0, r, 1e-8, 4, 4)
    path = api.estimate(m, api.Parametric(), max_em_iter=6, tol_em=0.0, factor_lags=1, nrep=3, seed=77, ngpu=1)
    assert len(path) == 6 and np.isnan(m.lambda_[17]).all() and np.isfinite(np.delete(m.lambda_, 17, axis=0)).all()
    rep = m.replicates
    assert rep["iterations"] == 6 and rep["panels"].shape == (3, T_all, ns - 1)
    keep = np.delete(np.arange(ns), 17)
    z, _ = api.standardize_data(raw[:, keep])
    np.testing.assert_array_equal(np.isnan(rep["panels"][1]), np.isnan(z))      # the window's own missing pattern
    start = {k: v for k, v in m.em_params.items()}
    for b in range(3):
        pe, pathe, _ = ko.em(rep["panels"][b], start, max_iter=6, tol=0.0)
        np.testing.assert_allclose(rep["loglik_path"][b], pathe, rtol=1e-8)
        for k in ("Lam", "R", "A", "Q"):
            assert np.abs(rep["params"][k][b] - pe[k]).max() <= 1e-8 * max(1.0, np.abs(pe[k]).max()), k
    # same seed -> same replicates; the draws are a function of (seed, fitted model) only
    m2 = api.DFMModel(raw, np.ones(ns, dtype=int), 20, 40, 1, T_all, 0, r, 1e-8, 4, 4)
    api.estimate(m2, api.Parametric(), max_em_iter=6, tol_em=0.0, factor_lags=1, nrep=3, seed=77)
    np.testing.assert_array_equal(np.nan_to_num(m2.replicates["panels"]), np.nan_to_num(rep["panels"]))
