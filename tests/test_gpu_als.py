"""GPU parity tests of the reference's NON-parametric estimator on the HIP path (als.hip): batched ALS
(`estimate_factor!`, dfm_functions.ipynb:328-382), batched complete-case OLS (`ols_skipmissing` :242-252) and the
`estimate!(m, NonParametric())` chain -- against the CPU oracle (oracle/als_oracle.py) and, through the whole chain
PCA (HIP) -> ALS (HIP) -> loadings / VAR (HIP), against every digit the reference's notebook saved
(tests/golden/notebook_goldens.json <- Stock_Watson.ipynb:572-576, 619-628, 991-1017, 1250-1261).  This part of
the path is PARITY PINNED by the reference itself."""
import json
import math
import os

import numpy as np
import pytest

from oracle import als_oracle as ao

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "notebook_goldens.json")))
INIT, LAST = 3, 224


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from dynamic_factor_models_amd import DfmContext
    c = DfmContext()
    yield c
    c.close()


@pytest.fixture(scope="module")
def sw():
    d = np.load(os.path.join(HERE, "golden", "sw_panel.npz"))
    bp, inc, cat = d["bpdata"], d["inclcode"], d["bpcatcode"]
    real = np.isin(np.floor(cat), [1, 2, 3, 5])
    return dict(all=bp, inc_all=inc, real=bp[:, real], inc_real=inc[real], names=[str(s) for s in d["bpnamevec"]])


def _shown(x, digits=3):
    return round(float(x), digits)


def _sig6(x, g):
    if g == 0:
        return abs(x) < 5e-7
    ulp = 10.0 ** (math.floor(math.log10(abs(g))) - 5)
    return abs(x - g) <= 0.5 * ulp * (1 + 1e-6)


def _synthetic_unbalanced(seed, T, N, r, miss):
    g = np.random.default_rng(seed)
    f = g.standard_normal((T, r))
    x = f @ g.standard_normal((r, N)) + 0.7 * g.standard_normal((T, N))
    x[g.random((T, N)) < miss] = np.nan
    x[: T // 3, N - 3:] = np.nan                 # series that start late
    x[:, N - 1] = np.nan
    x[:5, N - 1] = 1.0 + np.arange(5)            # a series with fewer than nt_min observations: no loadings
    x[:, : max(r + 2, N // 3)] = np.where(np.isnan(x[:, : max(r + 2, N // 3)]), 0.3, x[:, : max(r + 2, N // 3)])
    return x


@pytest.mark.parametrize("T,N,r,miss", [(60, 30, 3, 0.1), (45, 24, 1, 0.05), (80, 50, 6, 0.15), (50, 40, 11, 0.05)])
def test_als_batch_matches_oracle(ctx, T, N, r, miss):
    B = 3
    zs, F0s, refs = [], [], []
    for b in range(B):
        x = _synthetic_unbalanced(100 + b, T, N, r, miss)
        o = ao.estimate_factor(x, np.ones(N, int), 1, T, r, nt_min=10, max_iter=40, solver="normal")
        zs.append(o["z"])
        F0s.append(ao.pca_score(o["z"][:, ~np.isnan(o["z"]).any(axis=0)], r))
        refs.append(o)
    got = ctx.als_batch_host(np.stack(zs), np.stack(F0s), nt_min=10, max_iter=40, path_cap=40, want_R2=True)
    for b, o in enumerate(refs):
        k = o["iters"]
        assert got["iters"][b] == k
        np.testing.assert_allclose(got["ssr_path"][b, :k], o["ssr_path"], rtol=1e-10)
        assert np.isnan(got["ssr_path"][b, k:]).all()
        np.testing.assert_allclose(got["ssr"][b], o["ssr"], rtol=1e-10)
        assert np.abs(got["F"][b] - o["f"]).max() <= 1e-8 * np.abs(o["f"]).max()
        assert np.array_equal(np.isnan(got["Lam"][b]), np.isnan(o["lam"]))          # the undefined rows (:357)
        assert np.isnan(got["Lam"][b, N - 1]).all()
        m = ~np.isnan(o["lam"])
        assert np.abs(got["Lam"][b][m] - o["lam"][m]).max() <= 1e-8 * np.abs(o["lam"][m]).max()
        np.testing.assert_allclose(got["R2"][b], o["R2"], rtol=0, atol=1e-9, equal_nan=True)


def test_als_runs_with_different_numbers_of_factors_share_a_panel(ctx):
    """`estimate_factor_numbers` issues one run per r on the same window (dfm_functions.ipynb:698-725)."""
    T, N, rmax = 70, 36, 5
    x = _synthetic_unbalanced(7, T, N, 3, 0.1)
    z, _ = ao.standardize_data(x)
    F0 = ao.pca_score(z[:, ~np.isnan(z).any(axis=0)], rmax)
    got = ctx.als_batch_host(z, np.repeat(F0[None], rmax, axis=0), r_each=np.arange(1, rmax + 1), nt_min=10,
                             path_cap=400)
    for r in range(1, rmax + 1):
        o = ao.estimate_factor(x, np.ones(N, int), 1, T, r, nt_min=10, solver="normal", compute_r2_flag=False)
        assert got["iters"][r - 1] == o["iters"], r
        np.testing.assert_allclose(got["ssr"][r - 1], o["ssr"], rtol=1e-9)
        assert np.isnan(got["F"][r - 1][:, r:]).all() and np.isnan(got["Lam"][r - 1][:, r:]).all()
        assert np.abs(got["F"][r - 1][:, :r] - o["f"]).max() <= 1e-7 * np.abs(o["f"]).max()


def test_als_argument_errors(ctx):
    from dynamic_factor_models_amd import DfmError
    z = np.zeros((30, 8)); F0 = np.zeros((1, 30, 40))
    with pytest.raises(DfmError) as ei:
        ctx.als_batch_host(z, F0)
    assert ei.value.code == -2                                    # r > DFM_MAX_R
    with pytest.raises(DfmError) as ei:
        ctx.als_batch_host(np.zeros((5000, 8)), np.zeros((1, 5000, 4)))
    assert ei.value.code == -1                                    # factors + loadings exceed the LDS


@pytest.mark.parametrize("K,shared", [(5, True), (17, True), (4, False), (33, True)])
def test_ols_batch_matches_lstsq(ctx, K, shared):
    g = np.random.default_rng(K)
    T, P = 120, 23
    X = g.standard_normal((T, K)) if shared else g.standard_normal((P, T, K))
    if shared:
        X[:, -1] = 1.0
    Y = g.standard_normal((T, P)) + (X if shared else X[0]) @ g.standard_normal((K, P))
    Y[g.random((T, P)) < 0.1] = np.nan
    Y[:, 3] = np.nan
    Y[:K - 1, 3] = 1.0                                             # fewer complete rows than regressors
    if not shared:
        X[2, :4, :] = np.nan                                       # missing regressor rows drop out too
    got = ctx.ols_batch_host(X, Y, nt_min=0)
    for p in range(P):
        Xp = X if shared else X[p]
        ok = ~np.isnan(Y[:, p]) & ~np.isnan(Xp).any(axis=1)
        assert got["nobs"][p] == ok.sum()
        if ok.sum() < K:
            assert np.isnan(got["beta"][p]).all() and np.isnan(got["ssr"][p])
            continue
        b, e = ao._ols(Y[ok, p], Xp[ok])
        np.testing.assert_allclose(got["beta"][p], b, rtol=0, atol=1e-9 * np.abs(b).max())
        np.testing.assert_allclose(got["resid"][ok, p], e, rtol=0, atol=1e-9)
        assert np.isnan(got["resid"][~ok, p]).all()
        np.testing.assert_allclose(got["ssr"][p], e @ e, rtol=1e-9)
        d = Y[ok, p] - Y[ok, p].mean()
        np.testing.assert_allclose(got["tss"][p], d @ d, rtol=1e-9)


# ------------------------------------------------------------------------------------------------
# The reference's own numbers, through the HIP path end to end
def _model(api, data, inc, r, nlag=4):
    return api.DFMModel(data, inc, 20, 40, INIT, LAST, 0, r, 1e-8, 4, nlag)


def _check_table2(rows, fn, nrows):
    tr = 1.0 - fn["ssr_static"] / fn["tss"]
    marg = np.concatenate([[tr[0]], np.diff(tr)])
    ah = marg[:-1] / marg[1:]
    for k in range(nrows):
        g = rows[k]
        assert _shown(tr[k]) == g[1] and _shown(marg[k]) == g[2], (k, tr[k], marg[k], g)
        assert _shown(fn["bn_icp"][k]) == g[3] and _shown(ah[k]) == g[4], (k, fn["bn_icp"][k], ah[k], g)


def test_table2A_real_panel_on_the_gpu(ctx, sw):
    """Stock_Watson.ipynb:572-576: trace R2, marginal R2, Bai-Ng ICp2, Ahn-Horenstein ratio, r = 1..5."""
    from dynamic_factor_models_amd import api
    m = _model(api, sw["real"], sw["inc_real"], 4)
    fn = api.estimate_factor_numbers(m, range(1, 7), ctx=ctx)
    assert fn["nobs"] == 12700 and fn["T"] == 222 and fn["tss"] == pytest.approx(12700.0, rel=1e-12)
    _check_table2(GOLD["table2A_real"]["rows"], fn, 5)
    assert fn["iters"][3] == 78                                    # r = 4 converges at sweep 78
    np.testing.assert_allclose(fn["ssr_static"][3], 5531.4888394136, rtol=1e-9)


def test_table2B_full_panel_on_the_gpu(ctx, sw):
    """Stock_Watson.ipynb:619-628, r = 1..10 of the 11 runs the driver issues (139 series, 29 098 cells)."""
    from dynamic_factor_models_amd import api
    m = _model(api, sw["all"], sw["inc_all"], 4)
    fn = api.estimate_factor_numbers(m, range(1, 12), ctx=ctx)
    assert fn["nobs"] == 29098
    _check_table2(GOLD["table2B_all"]["rows"], fn, 10)


def test_ssr_path_of_config1(ctx, sw):
    """BASELINE config 1: Stock-Watson real panel, r = 4, PCA + 10 sweeps (SURVEY section 8(c) values)."""
    from dynamic_factor_models_amd import api
    z, _ = api.standardize_data(sw["real"][INIT - 1:LAST][:, sw["inc_real"] == 1])
    F0 = api.pca_start(ctx, z, 4)
    got = ctx.als_batch_host(z, F0[None], nt_min=20, max_iter=10, path_cap=10)
    want = [5546.5329290692, 5536.8680742366, 5534.6500793947, 5533.7811509890, 5533.3555332866, 5533.1041233104,
            5532.9293170552, 5532.7925260970, 5532.6774431342, 5532.5766246526]
    assert got["iters"][0] == 10
    np.testing.assert_allclose(got["ssr_path"][0], want, rtol=1e-9)


@pytest.mark.parametrize("r", [1, 3, 8, 10])
def test_table3_series_r2_on_the_gpu(ctx, sw, r):
    """Stock_Watson.ipynb:991-1017: r2 of `estimate!(m, NonParametric())` for all 207 series, 6 significant
    digits -- ALS, loadings regression and VAR all on the HIP kernels."""
    from dynamic_factor_models_amd import api
    col = {1: 0, 2: 1, 3: 2, 8: 3, 9: 4, 10: 5}[r]
    m = _model(api, sw["all"], sw["inc_all"], r)
    api.estimate(m, api.NonParametric(), ctx=ctx)
    first, last = GOLD["table3_r2"]["first_rows"], GOLD["table3_r2"]["last_rows"]
    for i, row in enumerate(first):
        assert _sig6(m.r2[i], row[col]), (i, m.r2[i], row[col])
    for i, row in enumerate(last):
        k = 207 - len(last) + i
        assert _sig6(m.r2[k], row[col]), (k, m.r2[k], row[col])
    # and the rest of the model object against the oracle
    o = ao.estimate_factor(sw["all"], sw["inc_all"], INIT, LAST, r, solver="normal", compute_r2_flag=False)
    lam, r2, uc, us = ao.estimate_factor_loading(sw["all"], o["factor"], INIT, LAST)
    # factors agree up to the sign of each PCA start column only when the starts agree: compare invariants
    np.testing.assert_allclose(m.fes.ssr, o["ssr"], rtol=1e-8)
    np.testing.assert_allclose(m.r2, r2, rtol=0, atol=1e-7, equal_nan=True)
    np.testing.assert_allclose(m.uar_ser, us, rtol=1e-6, equal_nan=True)
    np.testing.assert_allclose(m.uar_coef, uc, rtol=0, atol=1e-6, equal_nan=True)
    common_gpu = m.factor[INIT - 1:LAST] @ m.lambda_.T
    common_ref = o["factor"][INIT - 1:LAST] @ lam.T
    ok = ~np.isnan(common_ref)
    assert np.abs(common_gpu[ok] - common_ref[ok]).max() <= 1e-6 * np.abs(common_ref[ok]).max()


def test_table5_canonical_correlations_on_the_gpu(ctx, sw):
    """Stock_Watson.ipynb:1250-1261 (table A): 8 factors, VAR(4) residuals of factors and of observables (33
    regressors: the one-wave-per-problem OLS kernel)."""
    from dynamic_factor_models_amd import api
    bp, names = sw["all"], sw["names"]
    m = _model(api, bp, sw["inc_all"], 8)
    api.estimate(m, api.NonParametric(), ctx=ctx)
    fv = m.factor_var_model
    assert fv.betahat.shape == (33, 8) and fv.M.shape == (32, 32)
    vars_ = ["GDPC96", "PAYEMS", "PCECTPI", "FEDFUNDS"]
    X = np.column_stack([bp[:, names.index(v)] for v in vars_])
    v = api._var_model(X, 4, True, INIT, LAST)
    api.estimate_var(v, ctx=ctx)
    ok = ~np.isnan(np.column_stack([X, m.factor])).any(axis=1)
    lev = ao.canonical_correlations(X[ok], m.factor[ok])
    ok = ~np.isnan(np.column_stack([v.resid, fv.resid])).any(axis=1)
    res = ao.canonical_correlations(v.resid[ok], fv.resid[ok])
    for x, g in zip(res, GOLD["table5"]["A"]["resid"]):
        assert _sig6(x, g), (x, g)
    for x, g in zip(lev, GOLD["table5"]["A"]["level"]):
        assert _sig6(x, g), (x, g)
    np.testing.assert_allclose(fv.G[:8] @ fv.G[:8].T, fv.seps, rtol=1e-10)


def test_table2C_amengual_watson_on_the_gpu(ctx, sw):
    """Stock_Watson.ipynb:673-682: Amengual-Watson ICp2 for the number of dynamic factors given r static factors
    (columns r = 1..10, rows k <= r): static ALS, residual regressions (41 regressors at r = 10) and the ALS runs
    on the residual panels all on the HIP kernels."""
    from dynamic_factor_models_amd import api
    rows = GOLD["table2C_aw"]["rows"]
    for r in (1, 2, 4, 7, 10):
        m = _model(api, sw["all"], sw["inc_all"], r)
        api.estimate_factor(m, computeR2=False, ctx=ctx)
        aw, _ = api.amengual_watson_test(m, 4, ctx=ctx)
        for k in range(r):
            g = rows[k][1 + (r - 1)]
            assert g is not None
            assert _shown(aw[k]) == g or abs(aw[k] - g) <= 5.0001e-4, (r, k, aw[k], g)


def test_table2_all_runs_batched_in_two_als_calls(ctx, sw):
    """`estimate_factor_numbers(m, 1:10)` as the notebook calls it (Stock_Watson.ipynb:516, 643): the 10 static runs in
    ONE dfm_als_batch call and the 55 Amengual-Watson runs (k <= r, each on the residual window of its static run) in a
    SECOND one -- what julia/dfm_hip.jl estimate_factor_numbers_hip does.  Every printed digit of Tables 2A (:572-576) and
    2C (:673-682) must come out of the batched path."""
    from dynamic_factor_models_amd import api
    m = _model(api, sw["all"], sw["inc_all"], 1)
    o = api.estimate_factor_numbers(m, range(1, 11), ctx=ctx, with_aw=True)
    rows = GOLD["table2C_aw"]["rows"]
    for r in range(1, 11):
        for k in range(r):
            g = rows[k][1 + (r - 1)]
            assert g is not None
            v = o["aw_icp"][k, r - 1]
            assert _shown(v) == g or abs(v - g) <= 5.0001e-4, (r, k, v, g)
        assert np.isnan(o["aw_icp"][r:, r - 1]).all()                         # the reference's `missing` above the diagonal
    # the same runs one static count at a time (the unbatched route of test_table2C_...) give the same numbers
    m4 = _model(api, sw["all"], sw["inc_all"], 4)
    api.estimate_factor(m4, computeR2=False, ctx=ctx)
    aw4, ssr4 = api.amengual_watson_test(m4, 4, ctx=ctx)
    # (the ALS stopping rule is |dSSR| < tol T N: two routes to the same fixed point agree to that, not to rounding)
    np.testing.assert_allclose(o["aw_icp"][:4, 3], aw4, rtol=1e-5)
    np.testing.assert_allclose(o["ssr_dynamic"][:4, 3], ssr4, rtol=1e-5)


def test_standardize_batch_matches_reference_semantics(ctx):
    """dfm_functions.ipynb:501-509: mean and POPULATION s.d. over the observed cells, NaN preserved."""
    import torch
    from dynamic_factor_models_amd import api
    g = np.random.default_rng(3)
    x = 3.0 + 2.0 * g.standard_normal((4, 70, 33))
    x[g.random(x.shape) < 0.1] = np.nan
    t = torch.from_numpy(x.copy()).cuda()
    mu, sd = ctx.standardize_batch(t)
    torch.cuda.synchronize()
    for b in range(4):
        z, s = api.standardize_data(x[b])
        np.testing.assert_allclose(t[b].cpu().numpy(), z, rtol=1e-12, atol=1e-13, equal_nan=True)
        np.testing.assert_allclose(sd[b].cpu().numpy(), s[0], rtol=1e-13)
        np.testing.assert_allclose(mu[b].cpu().numpy(), np.nanmean(x[b], axis=0), rtol=1e-13)
