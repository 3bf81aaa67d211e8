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
    (2, 600, 80, 4, 0.0, 3),      # balanced, narrow state, rows beyond the 4-KB ring: collapse_wide2 / mstep_wide compute 16 columns, keep 4
    (17, 700, 60, 7, 0.0, 2),     # ... r padded 7 -> 8, XCD-ordered queues
    (3, 130, 70, 20, 0.0, 3),     # balanced, Rp = 32: mstep_wide (2 series blocks, the second partial; 3 stages of 32 periods, the last of 6)
    (17, 66, 90, 25, 0.0, 2),     # ... XCD-ordered item queues with B not a multiple of 8, 3 column groups past the first 16
    (2, 260, 110, 32, 0.0, 2),    # ... every factor column in use (two 16-wide tiles)
    (2, 300, 60, 20, 0.1, 2),     # Rp = 32 beyond the register tiling, missing cells: collapse_wide2 (missing-cell variant) + C_t kernel
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


# ---- multi-GPU drivers on one GPU: the stepping entry, the torch.distributed driver and the in-library (RCCL) driver
# are the same function of the inputs as dfm_em_batch_dev -------------------------------------------------------------
@pytest.mark.parametrize("B,N,T,r,missing,tol", [(6, 30, 60, 2, 0.0, 2e-4), (5, 40, 50, 3, 0.1, 0.0), (4, 200, 120, 8, 0.0, 0.0)])
def test_em_iterate_and_sharded_driver_equal_em_batch(ctx, B, N, T, r, missing, tol):
    import torch
    from dynamic_factor_models_amd import shard
    max_iter = 12
    panel, st = _start(B, N, T, r, missing)
    ref = {k: _dev(ctx, st[k]) for k in KEYS}
    path0, its0, f0, P0 = ctx.em_batch(_dev(ctx, panel), *[ref[k] for k in KEYS], max_iter=max_iter, tol=tol,
                                       may_have_missing=missing > 0)
    torch.cuda.synchronize()
    got = {k: _dev(ctx, st[k]) for k in KEYS}
    out = shard.em_batch_sharded(ctx, _dev(ctx, panel), *[got[k] for k in KEYS], B_global=B, max_iter=max_iter, tol=tol,
                                 may_have_missing=missing > 0)
    torch.cuda.synchronize()
    assert torch.equal(out["iters"], its0)
    # r == padded width: the same kernels on the same numbers, bit for bit.  r < padded width: dfm_em_batch_dev keeps the
    # padding states' (decoupled) parameters across iterations while the stepping entry re-embeds every iteration, and the
    # 2 x 2 block pivots of the sweep inverse round a real state together with a padding state: equal to rounding.
    same = torch.equal if r in (2, 4, 8, 16, 32) and missing == 0.0 else (lambda a, b: torch.allclose(a, b, rtol=1e-11, atol=1e-13))
    assert same(torch.nan_to_num(out["path"]), torch.nan_to_num(path0))
    for k in KEYS:
        assert same(got[k], ref[k]), k
    assert same(out["f"], f0) and same(out["P"], P0)
    assert out["iterations"] == int(its0.max().item()) or tol == 0.0
    assert out["loglik_global"].shape == (B, out["iterations"])
    if tol > 0.0:
        assert out["iterations"] < max_iter and not bool(out["active_global"].any())


def test_em_batch_multi_one_gpu_equals_em_batch_host(ctx):
    """dfm_em_batch_multi with ngpu = 1 (what the Julia host binds; RCCL not needed for one GPU) and the same through
    ngpu = 1 of dfm_ks_pass_batch_multi."""
    from dynamic_factor_models_amd import DfmContext
    panel, st = _start(5, 30, 50, 3, 0.1)
    for tol in (0.0, 1e-3):
        p0, path0, its0, f0, P0 = ctx.em_batch_host(panel, *[st[k] for k in KEYS], max_iter=9, tol=tol)
        p1, path1, its1, f1, P1, ran = DfmContext.em_batch_multi_host(1, panel, *[st[k] for k in KEYS], max_iter=9, tol=tol)
        np.testing.assert_array_equal(its0, its1)
        # r = 3 is padded to 4: the stepping entry re-embeds the parameters every iteration (equal to rounding, see above)
        np.testing.assert_allclose(np.nan_to_num(path0), np.nan_to_num(path1), rtol=1e-11)
        for k in KEYS:
            np.testing.assert_allclose(p0[k], p1[k], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(f0, f1, rtol=1e-10, atol=1e-12)
        assert ran == (9 if tol == 0.0 else its0.max())
    f, P, ll = ctx.ks_pass_batch_host(panel, *[st[k] for k in KEYS])
    f2, P2, ll2 = DfmContext.ks_pass_batch_multi_host(1, panel, *[st[k] for k in KEYS])
    np.testing.assert_array_equal(ll, ll2); np.testing.assert_array_equal(f, f2); np.testing.assert_array_equal(P, P2)


def test_em_batch_multi_argument_errors():
    from dynamic_factor_models_amd import DfmContext, DfmError
    panel, st = _start(2, 20, 30, 2, 0.0)
    with pytest.raises(DfmError) as ei:
        DfmContext.em_batch_multi_host(2, panel, *[st[k] for k in KEYS], max_iter=2, device_ids=[0, 0])
    assert ei.value.code == -1
    with pytest.raises(DfmError) as ei:
        DfmContext.em_batch_multi_host(1, panel, *[st[k] for k in KEYS], max_iter=2, device_ids=[99])
    assert ei.value.code == -1


# ---- BASELINE config 4 (N = 1000, T = 2000, r = 20) and config 3's shard (B = 8192) through EM ------------------
def test_config4_em_three_iterations_match_the_oracle(ctx):
    import torch
    B, N, T, r, iters = 2, 1000, 2000, 20, 3
    panel, st = _start(B, N, T, r, 0.0)
    dev = {k: _dev(ctx, st[k]) for k in KEYS}
    path, its, f, P = ctx.em_batch(_dev(ctx, panel), *[dev[k] for k in KEYS], max_iter=iters, tol=0.0, may_have_missing=False)
    torch.cuda.synchronize()
    path = path.cpu().numpy(); f = f.cpu().numpy()
    for b in range(B):
        p, opath, out = ko.em(panel[b], {k: st[k][b] for k in KEYS}, max_iter=iters, tol=0.0)
        np.testing.assert_allclose(path[b], opath, rtol=RTOL, err_msg=f"loglik path b={b}")
        for k in KEYS:
            got = dev[k][b].cpu().numpy()
            assert np.abs(got - p[k]).max() <= RTOL * max(1.0, np.abs(p[k]).max()), (k, b, np.abs(got - p[k]).max())
        assert np.abs(f[b] - out["f_smooth"]).max() <= RTOL * np.abs(out["f_smooth"]).max()


def test_config3_shard_em_scattered_replicates(ctx):
    """B = 8192 replicates (device-generated), PCA start on the device, 3 EM iterations; 12 replicates scattered over the
    batch against the oracle's EM from the same start."""
    import torch
    B, N, T, r, iters = 8192, 200, 500, 8, 3
    panel, _ = ctx.synth_panels(99, 0, B, T, N, r)
    Lam, R, A, Q, mu0, P0, _ = ctx.pca_init_batch(panel, r, want_factors=False)
    torch.cuda.synchronize()
    idx = [0, 3, 4, 2047, 2048, 2051, 4096, 5000, 6143, 6144, 8188, 8191]
    ix = torch.tensor(idx, device=panel.device)
    start = {k: v.index_select(0, ix).cpu().numpy() for k, v in zip(KEYS, (Lam, R, A, Q, mu0, P0))}
    xs = panel.index_select(0, ix).cpu().numpy()
    path, its, _, _ = ctx.em_batch(panel, Lam, R, A, Q, mu0, P0, max_iter=iters, tol=0.0, want_smooth=False,
                                   may_have_missing=False)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(path).all()) and bool((its == iters).all())
    got = {k: v.index_select(0, ix).cpu().numpy() for k, v in zip(KEYS, (Lam, R, A, Q, mu0, P0))}
    pth = path.index_select(0, ix).cpu().numpy()
    for j in range(len(idx)):
        p, opath, _ = ko.em(xs[j], {k: start[k][j] for k in KEYS}, max_iter=iters, tol=0.0)
        np.testing.assert_allclose(pth[j], opath, rtol=RTOL, err_msg=f"replicate {idx[j]}")
        for k in KEYS:
            assert np.abs(got[k][j] - p[k]).max() <= RTOL * max(1.0, np.abs(p[k]).max()), (k, idx[j])
    del panel
    torch.cuda.empty_cache()
