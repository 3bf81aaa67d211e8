"""AR idiosyncratic terms by quasi-differencing: dfm_ks_pass_ar_batch through the C-ABI against
oracle/ar_oracle.py (1e-9; north_star asks 1e-6)."""
import numpy as np
import pytest

from oracle import ar_oracle as aro
from oracle import varp_oracle as vo

pytestmark = pytest.mark.gpu
KEYS = ("Lam", "sig2", "rho", "Avar", "Q", "mu0", "P0")


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from dynamic_factor_models_amd import DfmContext
    c = DfmContext()
    yield c
    c.close()


def _case(b, N, T, r, p, q, miss):
    rng = np.random.default_rng([77, b, q])
    x = vo.synth_varp(b, N, T, r, p, missing=miss)
    st, _ = vo.varp_init(np.nan_to_num(x), r, p)
    m = aro.state_lags(p, q)
    k = r * m
    G0 = rng.standard_normal((k, k))
    return x, dict(Lam=st["Lam"], sig2=st["R"], rho=0.6 * rng.uniform(-1, 1, (N, q)) / max(q, 1), Avar=st["Avar"], Q=st["Q"],
                   mu0=0.1 * rng.standard_normal(k), P0=G0 @ G0.T / k + 0.5 * np.eye(k))


@pytest.mark.parametrize("B,N,T,r,p,q,miss", [(3, 30, 60, 2, 1, 1, 0.0), (2, 40, 70, 4, 4, 4, 0.0), (2, 25, 50, 3, 2, 2, 0.1),
                                               (2, 139, 222, 4, 4, 4, 0.05), (3, 20, 40, 1, 1, 3, 0.0), (2, 30, 50, 2, 3, 0, 0.1),
                                               (2, 33, 41, 5, 1, 4, 0.02)])
def test_ar_pass(ctx, B, N, T, r, p, q, miss):
    import torch
    cases = [_case(b, N, T, r, p, q, miss) for b in range(B)]
    x = np.stack([c[0] for c in cases])
    a = {k: np.stack([c[1][k] for c in cases]) for k in KEYS}
    dev = torch.device("cuda", ctx.device)
    t = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(dev)
    f, P, ll = ctx.ks_pass_ar_batch(t(x), *[t(a[k]) for k in KEYS])
    torch.cuda.synchronize()
    tri = np.tril_indices(r)
    for b in range(B):
        o = aro.kfs_pass_ar(x[b], **cases[b][1])
        assert abs(ll[b].item() - o["loglik"]) <= 1e-9 * abs(o["loglik"])
        fo = o["f_smooth"][:, :r]
        assert np.abs(f[b].cpu().numpy() - fo).max() <= 1e-9 * max(1.0, np.abs(fo).max())
        Po = o["P_smooth"][:, :r, :r][:, tri[0], tri[1]]
        assert np.abs(P[b].cpu().numpy() - Po).max() <= 1e-9 * max(1.0, np.abs(Po).max())


def test_ar_pass_host_entry_and_dimension_limit(ctx):
    from dynamic_factor_models_amd._lib import DfmError
    x, a = _case(0, 24, 40, 2, 2, 2, 0.05)
    f, P, ll = ctx.ks_pass_ar_batch_host(x[None], *[a[k][None] for k in KEYS])
    o = aro.kfs_pass_ar(x, **a)
    assert abs(ll[0] - o["loglik"]) <= 1e-9 * abs(o["loglik"])
    np.testing.assert_allclose(f[0], o["f_smooth"][:, :2], atol=1e-9)
    x, a = _case(0, 24, 40, 8, 1, 4, 0.0)                      # r (q + 1) = 40 > 32
    with pytest.raises(DfmError) as e:
        ctx.ks_pass_ar_batch_host(x[None], *[a[k][None] for k in KEYS])
    assert e.value.code == -2


def test_smooth_factors_ar_idio_on_the_stock_watson_panel():
    """estimate(m, NonParametric()) (the reference's estimator on the GPU) -> AR-idiosyncratic smoother with its
    uar_coef / uar_ser / factor VAR; library result == oracle on the same inputs; smoothed factors track the ALS ones."""
    import os
    from dynamic_factor_models_amd import api
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sw_panel.npz"))
    m = api.DFMModel(d["bpdata"], d["inclcode"], 20, 40, 3, 224, 0, 4, 1e-8, 4, 4)
    api.estimate(m, api.NonParametric())
    out = api.smooth_factors_ar_idio(m)
    a = out["inputs"]
    o = aro.kfs_pass_ar(a["x"], a["Lam"], a["sig2"], a["rho"], a["Avar"], a["Q"], a["mu0"], a["P0"])
    assert abs(out["loglik"] - o["loglik"]) <= 1e-8 * abs(o["loglik"])
    q = m.n_uarlag
    fs = out["factor"][m.initperiod - 1 + q:m.lastperiod]
    np.testing.assert_allclose(fs - out["mu_f"], o["f_smooth"][:, :4], atol=1e-7 * np.abs(o["f_smooth"]).max())
    assert np.isnan(out["factor"][:m.initperiod - 1 + q]).all()
    fa = m.factor[m.initperiod - 1 + q:m.lastperiod]
    for j in range(4):
        assert abs(np.corrcoef(fs[:, j], fa[:, j])[0, 1]) > 0.9
