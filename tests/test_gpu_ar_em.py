"""GPU: joint ECM estimation with AR(q) idiosyncratic terms (dfm_em_ar_batch*, mstep_ar.hip + the recursion kernels'
restricted transition step) against oracle/ar_oracle.py em_ar -- SURVEY.md §8 f3.  PARITY UNPINNED by the reference
(no Kalman / EM code there); the oracle is pinned by tests/test_oracle_ar_em.py."""
import numpy as np
import pytest

from oracle import ar_oracle as ao

pytestmark = pytest.mark.gpu

KEYS = ("Lam", "sig2", "rho", "Avar", "Q", "mu0", "P0")


@pytest.fixture(scope="module")
def ctx():
    from dynamic_factor_models_amd import DfmContext
    c = DfmContext(0)
    yield c
    c.close()


def _stack(B, N, T, r, p, q, missing):
    xs, sts = zip(*[ao.synth_ar(b, N, T, r, p, q, missing=missing) for b in range(B)])
    return np.stack(xs), {k: np.stack([s[k] for s in sts]) for k in KEYS}


@pytest.mark.parametrize("B,N,T,r,p,q,missing,iters", [
    (3, 12, 60, 2, 1, 1, 0.0, 4),      # state 4 (wave kernel, Rp = 8 after padding)
    (2, 15, 70, 2, 2, 2, 0.0, 3),      # state 6, VAR(2) inside a 3-lag state: restricted transition step
    (2, 14, 80, 3, 1, 2, 0.08, 3),     # missing cells: per-series sets of usable quasi-differenced rows
    (2, 20, 90, 4, 4, 4, 0.0, 3),      # the Stock-Watson shape: r = 4, p = 4, q = 4 -> state 20 (Grid<32>)
    (2, 10, 50, 1, 2, 4, 0.05, 3),     # one factor, q = 4
    (2, 16, 64, 8, 1, 2, 0.0, 2),      # r = 8, state 24
])
def test_em_ar_matches_oracle(ctx, B, N, T, r, p, q, missing, iters):
    panel, st = _stack(B, N, T, r, p, q, missing)
    est, path, its, f, P = ctx.em_ar_batch_host(panel, *[st[k] for k in KEYS], max_iter=iters)
    for b in range(B):
        ref, opath, out = ao.em_ar(panel[b], {k: st[k][b] for k in KEYS}, max_iter=iters)
        np.testing.assert_allclose(path[b], opath, rtol=1e-8, err_msg=f"loglik path b={b}")
        for k in KEYS:
            tol = 1e-7 * max(1.0, np.abs(ref[k]).max())
            assert np.abs(est[k][b] - ref[k]).max() <= tol, (k, b, np.abs(est[k][b] - ref[k]).max())
        fo = out["f_smooth"][:, :r]
        assert np.abs(f[b] - fo).max() <= 1e-8 * max(1.0, np.abs(fo).max())
    assert np.all(its == iters)


@pytest.mark.parametrize("B,N,T,r,p,q,missing", [
    (2, 50, 150, 4, 4, 4, 0.05),       # three series blocks of 48 / 2 series, two staged chunks of the panel (128 + 18 periods)
    (2, 33, 40, 2, 1, 0, 0.1),         # q = 0: one lag, a single tile product for the lanes to share
    (2, 17, 45, 5, 1, 3, 0.1),         # r = 5, q = 3: 20 states, odd N, one partial series block
])
def test_em_ar_moment_form_shapes(ctx, B, N, T, r, p, q, missing):
    """ar_moments_kernel / ar_solve_kernel beyond the shapes above: several series blocks and staged chunks, q = 0, odd widths."""
    panel, st = _stack(B, N, T, r, p, q, missing)
    est, path, its, f, P = ctx.em_ar_batch_host(panel, *[st[k] for k in KEYS], max_iter=2)
    for b in range(B):
        ref, opath, out = ao.em_ar(panel[b], {k: st[k][b] for k in KEYS}, max_iter=2)
        np.testing.assert_allclose(path[b], opath, rtol=1e-8, err_msg=f"loglik path b={b}")
        for k in KEYS:
            if ref[k].size == 0:                                  # (rho at q = 0)
                continue
            tol = 1e-7 * max(1.0, np.abs(ref[k]).max())
            assert np.abs(est[k][b] - ref[k]).max() <= tol, (k, b, np.abs(est[k][b] - ref[k]).max())


def test_two_sweep_kernel_agrees_with_the_moment_form():
    """Diagnostics build only (DFM_AR_MSTEP_OLD=1: mstep_ar_kernel, a thread per series, two sweeps): the same estimates to rounding."""
    import os
    from conftest import diag_only  # noqa: F401
    if os.environ.get("DFM_LIB") != "diag":
        pytest.skip("switch of the diagnostics build: run with DFM_LIB=diag")
    from dynamic_factor_models_amd import DfmContext
    panel, st = _stack(2, 30, 80, 4, 4, 4, 0.05)
    outs = []
    for old in ("0", "1"):
        os.environ["DFM_AR_MSTEP_OLD"] = old
        try:
            c = DfmContext(0)
            est, path, _, _, _ = c.em_ar_batch_host(panel, *[st[k] for k in KEYS], max_iter=2)
            outs.append((est, path))
            c.close()
        finally:
            os.environ.pop("DFM_AR_MSTEP_OLD", None)
    for k in KEYS:
        assert np.abs(outs[0][0][k] - outs[1][0][k]).max() <= 1e-10 * max(1.0, np.abs(outs[0][0][k]).max()), k


def test_em_ar_likelihood_monotone_and_tol(ctx):
    panel, st = _stack(4, 18, 120, 2, 1, 1, 0.0)
    est, path, its, _, _ = ctx.em_ar_batch_host(panel, *[st[k] for k in KEYS], max_iter=25, tol=1e-5)
    for b in range(4):
        pb = path[b, :its[b]]
        assert np.all(np.diff(pb) >= -1e-8 * np.abs(pb[:-1]))
        assert np.all(np.isnan(path[b, its[b]:]))
    assert its.min() >= 2 and its.max() <= 25


def test_em_ar_device_entry_updates_in_place(ctx):
    import torch
    panel, st = _stack(2, 12, 60, 2, 1, 1, 0.0)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d = {k: t(st[k]) for k in KEYS}
    path, its, f, P = ctx.em_ar_batch(t(panel), *[d[k] for k in KEYS], max_iter=3)
    torch.cuda.synchronize()
    ref, opath, _ = ao.em_ar(panel[1], {k: st[k][1] for k in KEYS}, max_iter=3)
    np.testing.assert_allclose(path[1].cpu().numpy(), opath, rtol=1e-8)
    assert np.abs(d["rho"][1].cpu().numpy() - ref["rho"]).max() <= 1e-7
    assert P.shape == (2, 60 - 1, 3)


def test_estimate_ar_idio_on_the_stock_watson_panel():
    """The reference's two-step estimator on the GPU (estimate(m, NonParametric())) as the start, then the joint ECM:
    library == oracle on the same inputs, likelihood non-decreasing, the model's fields re-estimated in place."""
    import os
    from dynamic_factor_models_amd import api
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sw_panel.npz"))
    m = api.DFMModel(d["bpdata"], d["inclcode"], 20, 40, 3, 224, 0, 4, 1e-8, 4, 4)
    api.estimate(m, api.NonParametric())
    inputs, cols, mu_f, _ = api._ar_model_inputs(m)
    lam0, rho0, ser0 = m.lambda_[cols].copy(), m.uar_coef[cols].copy(), m.uar_ser[cols].copy()
    path = api.estimate_ar_idio(m, max_em_iter=4, tol_em=0.0)
    start = dict(Lam=inputs["Lam"], sig2=inputs["sig2"], rho=inputs["rho"], Avar=inputs["Avar"], Q=inputs["Q"],
                 mu0=inputs["mu0"], P0=inputs["P0"])
    ref, opath, out = ao.em_ar(inputs["x"], start, max_iter=4)
    np.testing.assert_allclose(path, opath, rtol=1e-7)
    assert np.all(np.diff(path) >= -1e-8 * np.abs(path[:-1]))
    assert np.abs(m.lambda_[cols] - ref["Lam"]).max() <= 1e-6 * np.abs(ref["Lam"]).max()
    assert np.abs(m.uar_coef[cols] - ref["rho"]).max() <= 1e-6
    assert np.abs(m.uar_ser[cols] - np.sqrt(ref["sig2"])).max() <= 1e-6
    assert np.abs(m.factor_var_model.seps - ref["Q"]).max() <= 1e-6 * np.abs(ref["Q"]).max()
    q = m.n_uarlag
    fs = m.factor[m.initperiod - 1 + q:m.lastperiod] - mu_f
    assert np.abs(fs - out["f_smooth"][:, :4]).max() <= 1e-6 * np.abs(out["f_smooth"]).max()
    assert np.abs(m.lambda_[cols] - lam0).max() > 1e-4 and np.abs(m.uar_coef[cols] - rho0).max() > 1e-4   # it moved
    assert np.all(m.uar_ser[cols] > 0) and np.all(np.isfinite(ser0))
    k = 4 * 4
    assert m.factor_var_model.M.shape == (k, k)
    np.testing.assert_allclose(m.factor_var_model.M[:4], ref["Avar"], atol=1e-6)
