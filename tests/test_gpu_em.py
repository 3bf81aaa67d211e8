"""GPU parity tests of the EM loop (E-step = HIP smoother pass, M-step kernels) vs the CPU oracle:
per-iteration log-likelihood and updated parameters on the same seeded panels."""
import numpy as np
import pytest

from oracle import kalman_oracle as ko

pytestmark = pytest.mark.gpu
RTOL = 1e-8


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available()
    from dynamic_factor_models_amd import DfmContext
    c = DfmContext()
    yield c
    c.close()


def _start(B, N, T, r, missing):
    panels, starts = [], []
    for b in range(B):
        x, _ = ko.synth_replicate(b, N, T, r, missing=missing)
        if missing > 0:   # PCA start needs a balanced panel: fill NaN with 0 (= column mean) as DGR do
            p0, _ = ko.pca_init(np.nan_to_num(x), r)
        else:
            p0, _ = ko.pca_init(x, r)
        panels.append(x); starts.append(p0)
    panel = np.stack(panels)
    st = {k: np.stack([s[k] for s in starts]) for k in starts[0]}
    return panel, st


def _dev(ctx, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.device("cuda", ctx.device))


KEYS = ("Lam", "R", "A", "Q", "mu0", "P0")


@pytest.mark.parametrize("B,N,T,r,missing,iters", [
    (4, 200, 500, 8, 0.0, 3),     # BASELINE config-2 shape
    (9, 40, 80, 3, 0.0, 6),       # r padded 3 -> 4  (balanced: fast-path E-step + em_update_kernel)
    (3, 64, 48, 12, 0.0, 4),      # balanced, r padded to 16
    (2, 31, 41, 5, 0.0, 5),       # balanced, odd N: wide collapse
    (5, 40, 80, 4, 0.15, 6),      # missing cells: per-series normal equations
    (3, 139, 222, 4, 0.05, 10),   # config 1 shape: 10 EM iterations
    (2, 60, 50, 12, 0.1, 3),      # r padded to 16: global Dmiss accumulators
    (2, 300, 40, 5, 0.1, 3),      # N > 256: two series per lane
])
def test_em_path_and_params_match_oracle(ctx, B, N, T, r, missing, iters):
    import torch
    panel, st = _start(B, N, T, r, missing)
    dev = {k: _dev(ctx, st[k]) for k in KEYS}
    path, its, f, P = ctx.em_batch(_dev(ctx, panel), *[dev[k] for k in KEYS], max_iter=iters, tol=0.0)
    torch.cuda.synchronize()
    path = path.cpu().numpy(); f = f.cpu().numpy(); P = P.cpu().numpy()
    assert np.all(its.cpu().numpy() == iters)
    for b in range(B):
        p, opath, out = ko.em(panel[b], {k: st[k][b] for k in KEYS}, max_iter=iters, tol=0.0)
        np.testing.assert_allclose(path[b], opath, rtol=RTOL, err_msg=f"loglik path b={b}")
        for k in KEYS:
            got = dev[k][b].cpu().numpy()
            assert np.abs(got - p[k]).max() <= RTOL * max(1.0, np.abs(p[k]).max()), (k, b, np.abs(got - p[k]).max())
        assert np.abs(f[b] - out["f_smooth"]).max() <= RTOL * np.abs(out["f_smooth"]).max()
        assert np.abs(P[b] - ko.pack_sym(out["P_smooth"])).max() <= RTOL * np.abs(out["P_smooth"]).max()
        assert np.all(np.diff(path[b]) > -1e-9 * np.abs(path[b][:-1])), "EM log-likelihood must not decrease"


def test_em_single_step_in_place(ctx):
    import torch
    panel, st = _start(3, 50, 70, 4, 0.1)
    dev = {k: _dev(ctx, st[k]) for k in KEYS}
    ll = ctx.em_step_batch(_dev(ctx, panel), *[dev[k] for k in KEYS])
    torch.cuda.synchronize()
    for b in range(3):
        new, llo, _ = ko.em_step(panel[b], **{k: st[k][b] for k in KEYS})
        assert abs(ll[b].item() - llo) <= RTOL * abs(llo)
        for k in KEYS:
            np.testing.assert_allclose(dev[k][b].cpu().numpy(), new[k], rtol=0, atol=RTOL * max(1.0, np.abs(new[k]).max()))


def test_em_tolerance_stops_each_replicate_like_the_oracle(ctx):
    import torch
    B, N, T, r = 6, 30, 60, 2
    panel, st = _start(B, N, T, r, 0.0)
    tol, max_iter = 2e-4, 40
    dev = {k: _dev(ctx, st[k]) for k in KEYS}
    path, its, f, P = ctx.em_batch(_dev(ctx, panel), *[dev[k] for k in KEYS], max_iter=max_iter, tol=tol)
    torch.cuda.synchronize()
    path = path.cpu().numpy(); its = its.cpu().numpy()
    for b in range(B):
        p, opath, out = ko.em(panel[b], {k: st[k][b] for k in KEYS}, max_iter=max_iter, tol=tol)
        assert its[b] == len(opath), (b, its[b], len(opath))
        np.testing.assert_allclose(path[b, :its[b]], opath, rtol=RTOL)
        assert np.all(np.isnan(path[b, its[b]:]))
        for k in KEYS:
            got = dev[k][b].cpu().numpy()
            assert np.abs(got - p[k]).max() <= 1e-7 * max(1.0, np.abs(p[k]).max()), (k, b)
    assert its.min() < max_iter, "at least one replicate should stop early in this test"


def test_em_host_entry(ctx):
    panel, st = _start(2, 30, 40, 2, 0.1)
    newp, path, its, f, P = ctx.em_batch_host(panel, *[st[k] for k in KEYS], max_iter=4)
    for b in range(2):
        p, opath, out = ko.em(panel[b], {k: st[k][b] for k in KEYS}, max_iter=4)
        np.testing.assert_allclose(path[b], opath, rtol=RTOL)
        for k in KEYS:
            assert np.abs(newp[k][b] - p[k]).max() <= RTOL * max(1.0, np.abs(p[k]).max())
