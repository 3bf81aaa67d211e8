"""VAR(p) factor dynamics in companion form and the covariance-form recursion (DFM_F_SINGULAR_Q):
HIP path through the C-ABI against oracle/varp_oracle.py / kalman_oracle.py.  Tolerances: 1e-9 on the pass, 1e-8 on EM
(north_star asks 1e-6)."""
import numpy as np
import pytest

from oracle import kalman_oracle as ko
from oracle import varp_oracle as vo

pytestmark = pytest.mark.gpu
KEYS = ("Lam", "R", "Avar", "Q", "mu0", "P0")


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from dynamic_factor_models_amd import DfmContext
    c = DfmContext()
    yield c
    c.close()


def _batch(B, N, T, r, p, miss):
    xs, qs = [], []
    for b in range(B):
        x = vo.synth_varp(b, N, T, r, p, missing=miss)
        q, _ = vo.varp_init(np.nan_to_num(x), r, p)
        xs.append(x); qs.append(q)
    return np.stack(xs), {k: np.stack([q[k] for q in qs]) for k in KEYS}


def _close(got, want, tol, what):
    assert np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max()), what


@pytest.mark.parametrize("B,N,T,r,p,miss", [(3, 30, 60, 2, 2, 0.0), (2, 40, 80, 4, 4, 0.0), (3, 25, 50, 3, 2, 0.15),
                                             (2, 60, 70, 4, 4, 0.1), (5, 20, 40, 1, 5, 0.0), (2, 30, 45, 8, 4, 0.05),
                                             (2, 24, 40, 3, 1, 0.1), (1, 139, 222, 4, 4, 0.1),
                                             # round 6 (recursion_mbf16_kernel: 9 <= r p <= 16, r <= 4): collapsed observations 2 wide,
                                             # a state narrower than its padding, most cells missing, a batch beyond one wave per SIMD
                                             (3, 30, 50, 2, 6, 0.1), (2, 36, 64, 3, 4, 0.2), (2, 40, 33, 4, 3, 0.0), (2, 20, 90, 2, 8, 0.6),
                                             (1100, 12, 40, 3, 3, 0.1),
                                             # the collapse on collapse_miss_kernel's table (loadings <= 4 wide, missing cells; odd N padded)
                                             (2, 51, 60, 1, 4, 0.1), (3, 33, 70, 2, 3, 0.3), (2, 201, 64, 4, 2, 0.05)])
def test_varp_pass(ctx, B, N, T, r, p, miss):
    import torch
    x, q = _batch(B, N, T, r, p, miss)
    dev = torch.device("cuda", ctx.device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    f, P, ll = ctx.ks_pass_varp_batch(t(x), *[t(q[k]) for k in KEYS])
    torch.cuda.synchronize()
    tri = np.tril_indices(r)
    for b in range(B):
        o = vo.kfs_pass_varp(x[b], p=p, **{k: q[k][b] for k in KEYS})
        assert abs(ll[b].item() - o["loglik"]) <= 1e-9 * abs(o["loglik"])
        _close(f[b].cpu().numpy(), o["f_smooth"][:, :r], 1e-9, "f_smooth")
        _close(P[b].cpu().numpy(), o["P_smooth"][:, :r, :r][:, tri[0], tri[1]], 1e-9, "P_smooth")


@pytest.mark.parametrize("B,N,T,r,p,miss", [(3, 30, 60, 2, 2, 0.0), (2, 40, 80, 4, 4, 0.0), (2, 25, 50, 3, 2, 0.15),
                                             (2, 50, 70, 4, 4, 0.1), (2, 24, 40, 3, 1, 0.1),
                                             (2, 30, 50, 2, 6, 0.1), (2, 36, 64, 3, 4, 0.2), (2, 40, 45, 4, 3, 0.0),
                                             (2, 33, 70, 2, 3, 0.3), (2, 51, 60, 1, 4, 0.1)])
def test_varp_em(ctx, B, N, T, r, p, miss):
    import torch
    x, q = _batch(B, N, T, r, p, miss)
    dev = torch.device("cuda", ctx.device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d = {k: t(q[k]) for k in KEYS}
    path, its, f, P = ctx.em_varp_batch(t(x), *[d[k] for k in KEYS], max_iter=4, tol=0.0)
    torch.cuda.synchronize()
    for b in range(B):
        qo, opath, out = vo.em_varp(x[b], {k: q[k][b] for k in KEYS}, p, 4)
        np.testing.assert_allclose(path[b].cpu().numpy(), opath, rtol=1e-8)
        for k in KEYS:
            _close(d[k][b].cpu().numpy(), qo[k], 1e-8, k)
        _close(f[b].cpu().numpy(), out["f_smooth"][:, :r], 1e-8, "f_smooth")


def test_varp_em_host_entry_and_tolerance_stop(ctx):
    x, q = _batch(2, 30, 60, 2, 3, 0.0)
    new, path, its, f, P = ctx.em_varp_batch_host(x, *[q[k] for k in KEYS], max_iter=30, tol=1e-4)
    for b in range(2):
        qo, opath, _ = vo.em_varp(x[b], {k: q[k][b] for k in KEYS}, 3, 30, 1e-4)
        assert its[b] == len(opath)
        np.testing.assert_allclose(path[b, :its[b]], opath, rtol=1e-8)
        _close(new["Avar"][b], qo["Avar"], 1e-7, "Avar")
    f2, P2, ll = ctx.ks_pass_varp_batch_host(x, *[new[k] for k in KEYS])
    assert np.isfinite(ll).all()


def test_singular_Q_flag_runs_the_start_the_information_form_cannot(ctx):
    """T - 1 < 2r: rank-deficient Q from the PCA start (tests/test_gpu_fuzz.py::test_singular_Q_is_reported);
    with DFM_F_SINGULAR_Q the covariance-form recursion reproduces the oracle."""
    import torch
    B, N, T, r = 2, 52, 18, 11
    xs = [ko.synth_replicate(b, N, T, r, seed=ko.SEED0 + 17 * N + T)[0] for b in range(B)]
    starts = [ko.pca_init(x, r)[0] for x in xs]
    keys = ("Lam", "R", "A", "Q", "mu0", "P0")
    dev = torch.device("cuda", ctx.device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    x = np.stack(xs)
    st = {k: np.stack([s[k] for s in starts]) for k in keys}
    f, P, ll = ctx.ks_pass_batch(t(x), *[t(st[k]) for k in keys], may_have_missing=False, singular_q=True)
    torch.cuda.synchronize()
    for b in range(B):
        o = ko.kfs_pass(x[b], **starts[b])
        assert abs(ll[b].item() - o["loglik"]) <= 1e-8 * abs(o["loglik"])
        _close(f[b].cpu().numpy(), o["f_smooth"], 1e-7, "f_smooth")
    d = {k: t(st[k]) for k in keys}
    path, its, f, P = ctx.em_batch(t(x), *[d[k] for k in keys], max_iter=3, tol=0.0, singular_q=True)
    torch.cuda.synchronize()
    for b in range(B):
        _, opath, _ = ko.em(x[b], starts[b], 3)
        np.testing.assert_allclose(path[b].cpu().numpy(), opath, rtol=1e-7)


@pytest.mark.parametrize("miss", [0.0, 0.1])
def test_covariance_form_equals_information_form(ctx, miss):
    import torch
    B, N, T, r = 4, 40, 70, 5
    reps = [ko.synth_replicate(b, N, T, r, missing=miss) for b in range(B)]
    x = np.stack([a for a, _ in reps])
    st = {k: np.stack([p[k] for _, p in reps]) for k in reps[0][1]}
    dev = torch.device("cuda", ctx.device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    args = [t(st[k]) for k in ("Lam", "R", "A", "Q", "mu0", "P0")]
    a = ctx.ks_pass_batch(t(x), *args)
    b = ctx.ks_pass_batch(t(x), *args, singular_q=True)
    torch.cuda.synchronize()
    for u, v in zip(a, b):
        _close(v.cpu().numpy(), u.cpu().numpy(), 1e-10, "cov vs info")


def test_varp_and_ar_edge_cases(ctx):
    """Edges the reference's own tests would probe: a period with every cell missing, a single replicate, odd N, the
    shortest panel an AR(q) model admits (T = q + 2), q = 0."""
    import torch
    from oracle import ar_oracle as aro
    dev = torch.device("cuda", ctx.device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # VAR(2), r = 3, N odd, one period with no observation at all
    x = vo.synth_varp(5, 27, 40, 3, 2, missing=0.05)
    q, _ = vo.varp_init(np.nan_to_num(x), 3, 2)
    x[11, :] = np.nan
    f, P, ll = ctx.ks_pass_varp_batch(t(x[None]), *[t(q[k][None]) for k in KEYS])
    o = vo.kfs_pass_varp(x, p=2, **q)
    assert abs(ll[0].item() - o["loglik"]) <= 1e-9 * abs(o["loglik"])
    _close(f[0].cpu().numpy(), o["f_smooth"][:, :3], 1e-9, "all-missing period")
    # AR(3) idiosyncratic terms on T = q + 2 = 5 periods: two quasi-differenced rows
    rng = np.random.default_rng(3)
    N, r, qq = 12, 2, 3
    xs = rng.standard_normal((5, N))
    k = r * (qq + 1)
    G0 = rng.standard_normal((k, k))
    a = dict(Lam=rng.standard_normal((N, r)), sig2=rng.uniform(0.5, 1.5, N), rho=0.2 * rng.uniform(-1, 1, (N, qq)),
             Avar=0.5 * np.eye(r), Q=np.eye(r), mu0=np.zeros(k), P0=G0 @ G0.T / k + np.eye(k))
    keys = ("Lam", "sig2", "rho", "Avar", "Q", "mu0", "P0")
    f, P, ll = ctx.ks_pass_ar_batch(t(xs[None]), *[t(a[kk][None]) for kk in keys])
    o = aro.kfs_pass_ar(xs, **a)
    assert f.shape == (1, 2, r)
    assert abs(ll[0].item() - o["loglik"]) <= 1e-9 * abs(o["loglik"])
    _close(f[0].cpu().numpy(), o["f_smooth"][:, :r], 1e-9, "shortest AR panel")


@pytest.mark.parametrize("N,T,r,p", [(20, 40, 1, 12), (16, 70, 1, 16), (30, 24, 2, 5)])
def test_mbf16_edge_shapes(ctx, N, T, r, p):
    """recursion_mbf16_kernel (round 6) at the edges of its domain: one factor with 12 / 16 lags (collapsed observations 2 wide, one of them
    padding), periods without a single observed cell (C_t = 0: the rank-4 update must be the identity), a first and a last period among
    them, and the pass without P_smooth."""
    import torch
    x, q = _batch(2, N, T, r, p, 0.15)
    x[:, [0, 7, 8, T - 1], :] = np.nan
    dev = torch.device("cuda", ctx.device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    f, P, ll = ctx.ks_pass_varp_batch(t(x), *[t(q[k]) for k in KEYS])
    torch.cuda.synchronize()
    tri = np.tril_indices(r)
    for b in range(2):
        o = vo.kfs_pass_varp(x[b], p=p, **{k: q[k][b] for k in KEYS})
        assert abs(ll[b].item() - o["loglik"]) <= 1e-9 * abs(o["loglik"])
        _close(f[b].cpu().numpy(), o["f_smooth"][:, :r], 1e-9, "f_smooth")
        _close(P[b].cpu().numpy(), o["P_smooth"][:, :r, :r][:, tri[0], tri[1]], 1e-9, "P_smooth")


def test_rank_deficient_innovation_block_at_r4(ctx):
    """r = 4: recursion_comp_kernel eliminates the companion state in 4 x 4 blocks and inverts the r x r block Q.  A rank-deficient Q is
    reported (NaN log-likelihood on the device entry, DFM_E_NUMERIC on the host entry) and DFM_F_SINGULAR_Q runs the kernels that never
    invert it -- the oracle's covariance form has no such condition."""
    import torch
    from dynamic_factor_models_amd._lib import DfmError
    r, p, N, T = 4, 3, 30, 60
    x = vo.synth_varp(9, N, T, r, p, missing=0.1)
    q, _ = vo.varp_init(np.nan_to_num(x), r, p)
    G = np.linalg.cholesky(q["Q"])[:, :3]
    q["Q"] = G @ G.T                                                        # rank 3
    dev = torch.device("cuda", ctx.device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    args = [t(q[k][None]) for k in KEYS]
    _, _, ll = ctx.ks_pass_varp_batch(t(x[None]), *args)
    assert not np.isfinite(ll[0].item())
    with pytest.raises(DfmError) as e:
        ctx.ks_pass_varp_batch_host(x[None], *[q[k][None] for k in KEYS])
    assert e.value.code == -5
    f, P, ll = ctx.ks_pass_varp_batch(t(x[None]), *args, singular_q=True)
    o = vo.kfs_pass_varp(x, p=p, **q)
    assert abs(ll[0].item() - o["loglik"]) <= 1e-9 * abs(o["loglik"])
    _close(f[0].cpu().numpy(), o["f_smooth"][:, :r], 1e-9, "f_smooth, singular Q")
    f2, _, ll2 = ctx.ks_pass_varp_batch_host(x[None], *[q[k][None] for k in KEYS], singular_q=True)
    assert abs(ll2[0] - o["loglik"]) <= 1e-9 * abs(o["loglik"])
