"""GPU parity tests of the time-chunked recursion (csrc/recursion_chunk.hip: panels with missing cells at r <= 8; a replicate's
T periods as 64 chunks, one per lane) against the CPU oracle, through the C-ABI: the shapes that were benchmarked
(BASELINE config 2 with 10 % missing cells), samples that do not divide into the chunks, narrow states, panels whose filter
forgets too slowly (the boundary check must hand them to the sequential kernel), and the EM statistics."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import kalman_oracle as ko

pytestmark = pytest.mark.gpu
RTOL = 1e-9


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from dynamic_factor_models_amd import DfmContext
    c = DfmContext()
    yield c
    c.close()


def _batch(B, N, T, r, missing, first=0):
    reps = [ko.synth_replicate(first + b, N, T, r, missing=missing) for b in range(B)]
    panel = np.stack([x for x, _ in reps])
    st = {k: np.stack([p[k] for _, p in reps]) for k in reps[0][1]}
    return panel, st


def _dev(ctx, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.device("cuda", ctx.device))


KEYS = ("Lam", "R", "A", "Q", "mu0", "P0")


def _pass(ctx, panel, st, **kw):
    import torch
    f, P, ll = ctx.ks_pass_batch(_dev(ctx, panel), *[_dev(ctx, st[k]) for k in KEYS], **kw)
    torch.cuda.synchronize()
    return f.cpu().numpy(), None if P is None else P.cpu().numpy(), ll.cpu().numpy()


def _compare(got, ref, tag=""):
    f, P, ll = got
    fo, Po, llo = ref
    assert np.all(np.isfinite(ll)), tag
    np.testing.assert_allclose(ll, llo, rtol=RTOL, err_msg=f"loglik {tag}")
    assert np.abs(f - fo).max() <= RTOL * np.abs(fo).max(), f"f_smooth {tag}: {np.abs(f - fo).max()}"
    if P is not None:
        assert np.abs(P - Po).max() <= RTOL * np.abs(Po).max(), f"P_smooth {tag}: {np.abs(P - Po).max()}"


@pytest.mark.parametrize("B,N,T,r,missing", [
    (24, 200, 500, 8, 0.1),       # BASELINE config 2 with 10 % missing cells: 63 lanes of 8 periods, the last of 4
    (6, 200, 500, 8, 0.5),
    (5, 200, 512, 8, 0.1),        # every lane full
    (5, 200, 513, 8, 0.1),        # L = 9: 57 lanes
    (4, 120, 1100, 8, 0.1),       # L = 18 > W
    (4, 150, 97, 8, 0.1),         # L = 4 (the floor): 25 lanes
    (3, 139, 222, 4, 0.05),       # config 1 shape: state padded 4 -> 8, collapsed observations 4 wide
    (3, 90, 130, 2, 0.1),         # ... 2 wide
    (3, 100, 160, 5, 0.1),        # r padded 5 -> 8 (outputs 5 wide)
    (3, 160, 300, 7, 0.02),       # mostly complete periods: the replicate's full Gram matrix on the scalar path
])
def test_chunked_pass_matches_oracle(ctx, B, N, T, r, missing):
    panel, st = _batch(B, N, T, r, missing)
    ref = co.ks_pass_batch(panel, *[st[k] for k in KEYS])
    got = _pass(ctx, panel, st)
    nf, nt = ctx.chunk_fallbacks()
    assert nt == B, "the pass did not run on recursion_chunk_kernel"
    if (N, r) != (90, 2):      # (90 series on 2 factors, 10 % missing: one replicate of three misses the 1e-10 boundary tolerance -- still equal to the oracle)
        assert nf == 0, f"{nf} of {nt} replicates fell back to the sequential kernel"
    _compare(got, ref, f"B={B} N={N} T={T} r={r}")
    got = _pass(ctx, panel, st, want_P=False)
    _compare(got, ref, "want_P=False")


def test_slowly_forgetting_filters_go_to_the_sequential_kernel(ctx):
    # 12 series on 8 factors: the filter needs far more than 8 periods to forget its start; mixed with well-conditioned replicates
    B, T, r = 6, 300, 8
    panels, sts = [], []
    for b in range(B):
        N = 12 if b % 2 == 0 else 200
        x, p = ko.synth_replicate(b, N, T, r, missing=0.2)
        xx = np.full((T, 200), np.nan); xx[:, :N] = x
        Lam = np.zeros((200, r)); Lam[:N] = p["Lam"]
        R = np.ones(200); R[:N] = p["R"]
        panels.append(xx); sts.append(dict(p, Lam=Lam, R=R))
    panel = np.stack(panels)
    st = {k: np.stack([s[k] for s in sts]) for k in KEYS}
    ref = co.ks_pass_batch(panel, *[st[k] for k in KEYS])
    got = _pass(ctx, panel, st)
    nf, nt = ctx.chunk_fallbacks()
    assert nt == B and nf == 3, (nf, nt)
    _compare(got, ref, "mixed batch")


@pytest.mark.parametrize("B,N,T,r,missing,iters", [
    (6, 200, 500, 8, 0.1, 3),
    (4, 139, 222, 4, 0.05, 4),
    (3, 100, 160, 5, 0.1, 3),
])
def test_em_on_the_chunked_recursion_matches_oracle(ctx, B, N, T, r, missing, iters):
    import torch
    panels, starts = [], []
    for b in range(B):
        x, _ = ko.synth_replicate(b, N, T, r, missing=missing)
        p0, _ = ko.pca_init(np.nan_to_num(x), r)
        panels.append(x); starts.append(p0)
    panel = np.stack(panels)
    st = {k: np.stack([s[k] for s in starts]) for k in KEYS}
    dev = {k: _dev(ctx, st[k]) for k in KEYS}
    path, its, f, P = ctx.em_batch(_dev(ctx, panel), *[dev[k] for k in KEYS], max_iter=iters, tol=0.0)
    torch.cuda.synchronize()
    nf, nt = ctx.chunk_fallbacks()
    assert nt == B and nf == 0, (nf, nt)
    path = path.cpu().numpy(); f = f.cpu().numpy(); P = P.cpu().numpy()
    for b in range(B):
        p, opath, out = ko.em(panel[b], {k: st[k][b] for k in KEYS}, max_iter=iters, tol=0.0)
        np.testing.assert_allclose(path[b], opath, rtol=1e-8, err_msg=f"loglik path b={b}")
        for k in KEYS:
            got = dev[k][b].cpu().numpy()
            assert np.abs(got - p[k]).max() <= 1e-8 * max(1.0, np.abs(p[k]).max()), (k, b, np.abs(got - p[k]).max())
        assert np.abs(f[b] - out["f_smooth"]).max() <= 1e-8 * np.abs(out["f_smooth"]).max()
        assert np.abs(P[b] - ko.pack_sym(out["P_smooth"])).max() <= 1e-8 * np.abs(out["P_smooth"]).max()


@pytest.mark.parametrize("B,N,T,r,missing,blank", [
    (3, 200, 24, 8, 0.1, ()),             # the shortest sample the chunked kernel takes (L = 4, W = 8: two windows)
    (3, 200, 25, 8, 0.1, ()),
    (2, 300, 100, 8, 0.1, ()),            # N > 224: collapse_kernel's per-period arrays + the bridge, loadings as wide as the state
    (3, 200, 120, 8, 0.1, (0, 57, 58, 119)),   # periods without a single observed cell, first and last among them
    (1, 200, 500, 8, 0.1, ()),            # one replicate
])
def test_chunked_pass_edge_shapes(ctx, B, N, T, r, missing, blank):
    panel, st = _batch(B, N, T, r, missing, first=40)
    for t in blank:
        panel[:, t, :] = np.nan
    ref = co.ks_pass_batch(panel, *[st[k] for k in KEYS])
    got = _pass(ctx, panel, st)
    nf, nt = ctx.chunk_fallbacks()
    assert nt == B, "the pass did not run on recursion_chunk_kernel"
    _compare(got, ref, f"B={B} N={N} T={T} r={r} blank={blank}")


def test_em_with_replicates_on_both_kernels(ctx):
    """EM on a batch in which every other replicate fails the boundary check: the EM bookkeeping (log-likelihood path, iteration
    counts, the transition M-step) is done by recursion_chunk_kernel for some replicates and by the sequential kernel for the
    others, iteration after iteration."""
    import torch
    B, T, r, iters = 6, 200, 8, 3
    panels, starts = [], []
    for b in range(B):
        N = 12 if b % 2 == 0 else 200
        x, p = ko.synth_replicate(b, N, T, r, missing=0.15)
        xx = np.full((T, 200), np.nan); xx[:, :N] = x
        Lam = 0.01 * np.random.default_rng(b).standard_normal((200, r)); Lam[:N] = p["Lam"]
        R = np.ones(200); R[:N] = p["R"]
        panels.append(xx); starts.append(dict(p, Lam=Lam, R=R))
    panel = np.stack(panels)
    st = {k: np.stack([s[k] for s in starts]) for k in KEYS}
    dev = {k: _dev(ctx, st[k]) for k in KEYS}
    path, its, f, P = ctx.em_batch(_dev(ctx, panel), *[dev[k] for k in KEYS], max_iter=iters, tol=0.0)
    torch.cuda.synchronize()
    nf, nt = ctx.chunk_fallbacks()
    assert nt == B and 0 < nf < B, (nf, nt)
    path = path.cpu().numpy(); f = f.cpu().numpy()
    assert np.all(its.cpu().numpy() == iters)
    for b in range(B):
        keep = ~np.isnan(panel[b]).all(axis=0)                   # (series without any observed cell keep their parameters: drop them for the oracle)
        sub = {k: (st[k][b][keep] if k in ("Lam", "R") else st[k][b]) for k in KEYS}
        p, opath, out = ko.em(panel[b][:, keep], sub, max_iter=iters, tol=0.0)
        np.testing.assert_allclose(path[b], opath, rtol=1e-8, err_msg=f"loglik path b={b}")
        for k in ("A", "Q", "mu0", "P0"):
            got = dev[k][b].cpu().numpy()
            assert np.abs(got - p[k]).max() <= 1e-7 * max(1.0, np.abs(p[k]).max()), (k, b, np.abs(got - p[k]).max())
        assert np.abs(dev["Lam"][b].cpu().numpy()[keep] - p["Lam"]).max() <= 1e-7 * max(1.0, np.abs(p["Lam"]).max())
        assert np.abs(f[b] - out["f_smooth"]).max() <= 1e-7 * np.abs(out["f_smooth"]).max()


# ---- the boundary check as a BOUND: a sweep over forgetting rates (round 6) ------------------------------------------------------------
def _ctx_with_env(**env):
    import os
    from dynamic_factor_models_amd import DfmContext
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return DfmContext()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _rel_errors(got, ref):
    """Per replicate: the largest relative deviation (max-norm) of the log-likelihood, f_smooth and P_smooth from the oracle."""
    f, P, ll = got
    fo, Po, llo = ref
    ax = tuple(range(1, f.ndim))
    e = np.abs(ll - llo) / np.abs(llo)
    e = np.maximum(e, np.abs(f - fo).max(axis=ax) / np.abs(fo).max(axis=ax))
    return np.maximum(e, np.abs(P - Po).max(axis=ax) / np.abs(Po).max(axis=ax))


SWEEP_N = (10, 12, 16, 24, 32, 48, 64, 96, 160)          # r + 2 ... 4 r at r = 8 (slow forgetting) and on into the regime that forgets
SWEEP_MISS = (0.0, 0.2, 0.4, 0.6)


@pytest.mark.parametrize("W", [2, 4, 8])
def test_forgetting_rate_sweep_every_replicate_is_exact_or_was_redone(W):
    """r = 8, T = 200 (L = 4), N from r + 2 up, 0 ... 60 % missing cells, warm-ups of 2 / 4 / 8 periods: the states at the chunk
    boundaries differ by anything between 1e-1 and 1e-12 (scripts/dbg/chunk_emul.py prints them).  Whatever the rate, EVERY replicate
    must come out within 1e-9 of the oracle -- because its boundaries passed the element-wise check at 1e-10, or because the
    sequential kernel redid it (dfm_chunk_fallbacks counts those).  Both classes must occur."""
    c = _ctx_with_env(DFM_CHUNK_W=W)
    B, T, r = 4, 200, 8
    tot_f = tot = 0
    try:
        for N in SWEEP_N:
            for miss in SWEEP_MISS:
                panel, st = _batch(B, N, T, r, miss, first=100)
                ref = co.ks_pass_batch(panel, *[st[k] for k in KEYS])
                got = _pass(c, panel, st, may_have_missing=True)
                nf, nt = c.chunk_fallbacks()
                assert nt == B
                err = _rel_errors(got, ref)
                assert err.max() <= RTOL, f"W={W} N={N} miss={miss}: {err} ({nf} of {nt} redone)"
                if N <= 24:
                    assert nf == B, f"W={W} N={N} miss={miss}: {nf} of {nt} (a filter that cannot have forgotten passed its check)"
                tot_f += nf; tot += nt
    finally:
        c.close()
    assert tot_f > 0
    if W == 8:
        assert tot_f < tot


@pytest.mark.parametrize("tol", [1e-4, 1e-6, 1e-8])
def test_a_boundary_within_tol_means_a_result_within_tol(tol):
    """The intermediate regime: with the boundary tolerance LOOSENED to `tol` the kernel keeps replicates whose chunks start from
    states that far off.  The check is a bound: every replicate it keeps is within `tol` of the oracle (an error at a chunk's start
    only contracts along the chunk), and the ones it hands to the sequential kernel are exact.  The sweep must contain kept
    replicates whose error is far above the production tolerance -- otherwise it did not reach the regime."""
    c = _ctx_with_env(DFM_CHUNK_TOL=tol)
    B, T, r = 4, 200, 8
    worst, kept = 0.0, 0
    try:
        for N in (32, 40, 48, 64, 96):
            for miss in SWEEP_MISS:
                panel, st = _batch(B, N, T, r, miss, first=100)
                ref = co.ks_pass_batch(panel, *[st[k] for k in KEYS])
                got = _pass(c, panel, st, may_have_missing=True)
                nf, nt = c.chunk_fallbacks()
                err = _rel_errors(got, ref)
                assert err.max() <= tol, f"tol={tol} N={N} miss={miss}: {err} ({nf} of {nt} redone)"
                worst = max(worst, err.max()); kept += nt - nf
    finally:
        c.close()
    assert kept > 0 and worst > 1e-3 * tol, (kept, worst)


def test_narrow_state_replicates_on_both_kernels(ctx):
    """r = 4 on the 8-wide state through collapse_miss_kernel's table (CollapseArgs::lam_w) with every other replicate handed back:
    chunk_unbridge_kernel<4> writes the sequential kernel's 4-wide rows from the 8-wide table.  Odd N: one all-missing series appended."""
    B, T, r, NN = 6, 240, 4, 141
    panels, sts = [], []
    for b in range(B):
        N = 7 if b % 2 == 0 else NN
        x, p = ko.synth_replicate(100 + b, N, T, r, missing=0.1)
        xx = np.full((T, NN), np.nan); xx[:, :N] = x
        Lam = np.zeros((NN, r)); Lam[:N] = p["Lam"]
        R = np.ones(NN); R[:N] = p["R"]
        panels.append(xx); sts.append(dict(p, Lam=Lam, R=R))
    panel = np.stack(panels)
    st = {k: np.stack([s[k] for s in sts]) for k in KEYS}
    ref = co.ks_pass_batch(panel, *[st[k] for k in KEYS])
    got = _pass(ctx, panel, st)
    nf, nt = ctx.chunk_fallbacks()
    assert nt == B and 3 <= nf < B, (nf, nt)
    _compare(got, ref, "narrow mixed batch")


@pytest.mark.parametrize("B,N,T,r,missing,iters", [
    (3, 101, 160, 8, 0.1, 3),             # odd N at Rp = 8: the appended series has no observed cell in any period
    (3, 77, 130, 3, 0.1, 3),              # r = 3 -> collapsed observations 4 wide, odd N
    (2, 51, 120, 2, 0.05, 2),
])
def test_em_odd_n_through_the_table(ctx, B, N, T, r, missing, iters):
    import torch
    panels, starts = [], []
    for b in range(B):
        x, _ = ko.synth_replicate(60 + b, N, T, r, missing=missing)
        p0, _ = ko.pca_init(np.nan_to_num(x), r)
        panels.append(x); starts.append(p0)
    panel = np.stack(panels)
    st = {k: np.stack([s[k] for s in starts]) for k in KEYS}
    ref = co.ks_pass_batch(panel, *[st[k] for k in KEYS])
    _compare(_pass(ctx, panel, st), ref, f"pass N={N} r={r}")
    dev = {k: _dev(ctx, st[k]) for k in KEYS}
    path, its, f, P = ctx.em_batch(_dev(ctx, panel), *[dev[k] for k in KEYS], max_iter=iters, tol=0.0)
    torch.cuda.synchronize()
    nf, nt = ctx.chunk_fallbacks()
    assert nt == B, (nf, nt)
    path = path.cpu().numpy(); f = f.cpu().numpy()
    for b in range(B):
        p, opath, out = ko.em(panel[b], {k: st[k][b] for k in KEYS}, max_iter=iters, tol=0.0)
        np.testing.assert_allclose(path[b], opath, rtol=1e-8, err_msg=f"loglik path b={b}")
        for k in KEYS:
            got = dev[k][b].cpu().numpy()
            assert np.abs(got - p[k]).max() <= 1e-8 * max(1.0, np.abs(p[k]).max()), (k, b, np.abs(got - p[k]).max())
        assert np.abs(f[b] - out["f_smooth"]).max() <= 1e-8 * np.abs(out["f_smooth"]).max()
