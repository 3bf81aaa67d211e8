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
