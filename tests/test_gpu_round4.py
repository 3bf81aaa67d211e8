"""GPU parity tests added in round 4 ("Next round" item 1): the last kernels that were benchmarked at one launch
geometry and compared with the oracle at another.

* PCA start (`gram_xx_dma_kernel` + `pca_kernel<8, 512>` with two workgroups per CU) at the batch `bench.py --mode pca` times:
  B = 1024, N = 200, T = 500, r = 8 -- 32 scattered replicates (b, b + 256, b + 512, b + 768: both workgroups of a CU and all
  four rounds of the grid) vs `ko.pca_init` at 1e-8; and at config 4's occupancy (B = 256, N = 1000, T = 2000, r = 20);
* EM at the headline batch (B = 1024: `pass_fused` + `em_update` + `mstep_mfma`), three iterations from the GPU's PCA start vs
  `ko.em` from the SAME start (1e-8) and, chained, vs the oracle's own PCA start + EM (north_star's 1e-6);
* EM at B = 1024 with 10 % of the cells missing (`collapse_miss` + `recursion_pair` + `mstep_lam` at one replicate per SIMD);
* odd N beyond the register tiling with missing cells at r > 16 (refused until round 4: the library appends one all-missing
  series with zero loadings), pass and EM;
* a series without a single observed cell keeps its parameters in every loadings step.
"""
import numpy as np
import pytest

from oracle import kalman_oracle as ko
from test_gpu_ks_pass import _compare, _oracle

pytestmark = pytest.mark.gpu

KEYS = ("Lam", "R", "A", "Q", "mu0", "P0")


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from dynamic_factor_models_amd import DfmContext
    c = DfmContext()
    yield c
    c.close()


def _em_c(x, start, iters):
    """ko.em(tol = 0) on the C twin of the oracle (oracle/dfm_oracle.c): affordable at config 4's size"""
    from oracle import c_oracle as co
    p = {k: np.array(v, float) for k, v in start.items()}
    path = []
    for _ in range(iters):
        p, ll = co.em_step(x, **p)
        path.append(ll)
    return p, np.array(path)


def _take(t, ix):
    return t.index_select(0, ix).cpu().numpy()


def _scattered(B, stride, n, seed):
    """n residues b < stride (first, last, seeded rest), each with b + stride, b + 2 stride, ... below B."""
    rng = np.random.default_rng(seed)
    base = sorted(set([0, stride - 1] + rng.choice(stride, size=n, replace=False).tolist()))[:n]
    return sorted(b + j * stride for b in base for j in range(B // stride))


def _check_start(got, panel, r, idx, label):
    for j, b in enumerate(idx):
        ref, Fo = ko.pca_init(panel[j], r)
        np.testing.assert_allclose(got["F"][j], Fo, rtol=0, atol=1e-8 * np.abs(Fo).max(), err_msg=f"{label}: scores, replicate {b}")
        for k in KEYS:
            np.testing.assert_allclose(got[k][j], ref[k], rtol=0, atol=1e-8 * max(np.abs(ref[k]).max(), 1e-300),
                                       err_msg=f"{label}: {k}, replicate {b}")


def test_pca_start_at_the_benchmarked_batch(ctx):
    import torch
    B, N, T, r = 1024, 200, 500, 8
    panel, _ = ctx.synth_panels(20160415, 0, B, T, N, r)
    out = ctx.pca_init_batch(panel, r)
    ctx.synchronize()
    torch.cuda.synchronize()
    idx = _scattered(B, 256, 8, seed=4)
    assert len(idx) == 32
    ix = torch.tensor(idx, device=panel.device)
    got = dict(zip(KEYS + ("F",), [_take(t, ix) for t in out]))
    _check_start(got, _take(panel, ix), r, idx, "PCA start, B = 1024")


def test_pca_start_at_config4_occupancy(ctx):
    import torch
    B, N, T, r = 256, 1000, 2000, 20
    panel, _ = ctx.synth_panels(5, 0, B, T, N, r)
    out = ctx.pca_init_batch(panel, r)
    ctx.synchronize()
    torch.cuda.synchronize()
    idx = [3, 40, 77, 101, 130, 171, 222, 254]                     # (eight scattered replicates: a rare-replicate defect, not only a broken kernel)
    ix = torch.tensor(idx, device=panel.device)
    got = dict(zip(KEYS + ("F",), [_take(t, ix) for t in out]))
    _check_start(got, _take(panel, ix), r, idx, "PCA start, config 4")
    del panel, out
    torch.cuda.empty_cache()


def _em_vs_oracle(ctx, panel, par, idx, iters, missing, label, chain_r=None):
    import torch
    ix = torch.tensor(idx, device=panel.device)
    start = dict(zip(KEYS, [_take(p, ix) for p in par]))          # the parameters entering the GPU's EM
    xs = _take(panel, ix)
    path, its, _, _ = ctx.em_batch(panel, *par, max_iter=iters, tol=0.0, want_smooth=False, may_have_missing=missing)
    ctx.synchronize()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(path).all()) and bool((its == iters).all())
    got = dict(zip(KEYS, [_take(p, ix) for p in par]))
    pth = _take(path, ix)
    for j, b in enumerate(idx):
        p, opath, _ = ko.em(xs[j], {k: start[k][j] for k in KEYS}, max_iter=iters, tol=0.0)
        np.testing.assert_allclose(pth[j], opath, rtol=1e-9, err_msg=f"{label}: log-likelihood path, replicate {b}")
        for k in KEYS:
            err = np.abs(got[k][j] - p[k]).max()
            assert err <= 1e-8 * max(1.0, np.abs(p[k]).max()), (label, k, b, err)
        if chain_r is not None:                                  # the oracle's OWN start, then its EM: the whole chain at north_star's 1e-6
            ref, _ = ko.pca_init(xs[j], chain_r)
            p2, opath2, _ = ko.em(xs[j], ref, max_iter=iters, tol=0.0)
            np.testing.assert_allclose(pth[j], opath2, rtol=1e-6, err_msg=f"{label}: chained path, replicate {b}")
            for k in KEYS:
                err = np.abs(got[k][j] - p2[k]).max()
                assert err <= 1e-6 * max(1.0, np.abs(p2[k]).max()), (label, "chained", k, b, err)


def test_em_at_the_headline_batch_from_the_pca_start(ctx):
    B, N, T, r = 1024, 200, 500, 8
    panel, _ = ctx.synth_panels(20160415, 0, B, T, N, r)
    out = ctx.pca_init_batch(panel, r)
    par = [t.clone() for t in out[:6]]
    _em_vs_oracle(ctx, panel, par, _scattered(B, 256, 4, seed=6), 3, False, "EM, B = 1024, balanced", chain_r=r)


def test_em_at_the_headline_batch_with_missing_cells(ctx):
    B, N, T, r = 1024, 200, 500, 8
    panel, par = ctx.synth_panels(20160415, 0, B, T, N, r, missing_prob=0.1)
    _em_vs_oracle(ctx, panel, list(par), _scattered(B, 256, 4, seed=7), 3, True, "EM, B = 1024, 10 % missing")


@pytest.mark.parametrize("B,N,T,r,miss", [(3, 301, 40, 20, 0.1), (2, 257, 25, 17, 0.3), (2, 699, 60, 32, 0.05)])
def test_odd_n_beyond_the_tiling_with_missing_cells(ctx, B, N, T, r, miss):
    """dfm_functions.ipynb:352-366 takes any cross-section; until round 4 the library refused odd N > 256 at r > 16."""
    import torch
    reps = [ko.synth_replicate(b, N, T, r, seed=ko.SEED0 + N, missing=miss) for b in range(B)]
    panel = np.stack([x for x, _ in reps])
    st = {k: np.stack([p[k] for _, p in reps]) for k in KEYS}
    dev = torch.device("cuda", ctx.device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    f, P, ll = ctx.ks_pass_batch(t(panel), *[t(st[k]) for k in KEYS], may_have_missing=True)
    torch.cuda.synchronize()
    _compare((f.cpu().numpy(), P.cpu().numpy(), ll.cpu().numpy()), _oracle(panel, st), f"odd N = {N}, r = {r}")
    if T >= 2 * r:
        par = [t(st[k]) for k in KEYS]
        iters = 2
        path, its, _, _ = ctx.em_batch(t(panel), *par, max_iter=iters, tol=0.0, want_smooth=False, may_have_missing=True)
        ctx.synchronize()
        torch.cuda.synchronize()
        for b in range(B):
            p, opath, _ = ko.em(panel[b], {k: st[k][b] for k in KEYS}, max_iter=iters, tol=0.0)
            np.testing.assert_allclose(path[b].cpu().numpy(), opath, rtol=1e-9)
            for k, g in zip(KEYS, par):
                err = np.abs(g[b].cpu().numpy() - p[k]).max()
                assert err <= 1e-8 * max(1.0, np.abs(p[k]).max()), (k, b, err)


@pytest.mark.parametrize("N,T,r", [(40, 60, 3), (300, 50, 6), (300, 40, 20)])
def test_a_series_without_observed_cells_keeps_its_parameters(ctx, N, T, r):
    """All three loadings steps (mstep_lam_kernel with register / global accumulators, mmw_finish_kernel) on a panel whose
    series 5 has no observed cell: its loadings and R stay, every other series matches the oracle run without that series."""
    import torch
    B = 2
    reps = [ko.synth_replicate(b, N, T, r, seed=77, missing=0.05) for b in range(B)]
    panel = np.stack([x for x, _ in reps])
    panel[:, :, 5] = np.nan
    st = {k: np.stack([p[k] for _, p in reps]) for k in KEYS}
    dev = torch.device("cuda", ctx.device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    par = [t(st[k]) for k in KEYS]
    path, its, _, _ = ctx.em_batch(t(panel), *par, max_iter=1, tol=0.0, want_smooth=False, may_have_missing=True)
    ctx.synchronize()
    torch.cuda.synchronize()
    keep = [i for i in range(N) if i != 5]
    for b in range(B):
        sub = {k: (st[k][b][keep] if k in ("Lam", "R") else st[k][b]) for k in KEYS}
        p, opath, _ = ko.em(panel[b][:, keep], sub, max_iter=1, tol=0.0)
        np.testing.assert_allclose(path[b].cpu().numpy(), opath, rtol=1e-9)
        Lam = par[0][b].cpu().numpy(); R = par[1][b].cpu().numpy()
        assert np.array_equal(Lam[5], st["Lam"][b][5]) and R[5] == st["R"][b][5]
        assert np.abs(Lam[keep] - p["Lam"]).max() <= 1e-8 * max(1.0, np.abs(p["Lam"]).max())
        assert np.abs(R[keep] - p["R"]).max() <= 1e-8


def test_a_stale_status_bit_does_not_fail_the_next_host_call(ctx):
    """the status word is sticky; an unchecked *_dev call that raised a bit (NaN in a panel declared balanced)
    must not make the NEXT, unrelated host-pointer call fail -- host entries open a new status epoch."""
    import torch
    panel, par = ctx.synth_panels(3, 0, 4, 60, 40, 4)
    bad = panel.clone()
    bad[1, 7, 3] = float("nan")
    ctx.ks_pass_batch(bad, *par, may_have_missing=False)          # raises bit 1 on the device, nobody checks
    torch.cuda.synchronize()
    x = panel.cpu().numpy()
    st = [p.cpu().numpy() for p in par]
    f, P, ll = ctx.ks_pass_batch_host(x, *st)                      # used to raise DFM_E_MISSING for a clean panel
    fo, Po, llo = _oracle(x, dict(zip(KEYS, st)))
    np.testing.assert_allclose(ll, llo, rtol=1e-9)
    ctx.check_status()                                             # and nothing is left behind


def test_multi_object_on_every_visible_device(ctx):
    """The library's multi-GPU object over ALL the devices of the box (skips on a one-GPU box: it runs the day a node appears --
    VERDICT r3 item 5).  Shards, per-device LDS opt-in, the RCCL communicator and the per-iteration ncclAllGather over > 1 rank
    must reproduce the single-handle results bit for bit (replicates are independent; the collective carries convergence state only)."""
    import torch
    from dynamic_factor_models_amd import DfmMulti
    G = torch.cuda.device_count()
    if G < 2:
        pytest.skip("one visible device: the multi-rank branch is covered by the forced 1-rank communicator (test_gpu_round3)")
    B, N, T, r, iters = 8 * G + 3, 200, 120, 8, 4                 # (uneven shards)
    seed, first = 99, 10
    m = DfmMulti(G)
    try:
        assert m.ngpu == G and m.has_comm
        m.synth(seed, first, B, T, N, r)
        panel, par = ctx.synth_panels(seed, first, B, T, N, r)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(m.fetch("panel"), panel.cpu().numpy())
        m.ks_pass(want_P=True)
        f, P, ll = ctx.ks_pass_batch(panel, *par, may_have_missing=False)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(m.fetch("loglik"), ll.cpu().numpy())
        np.testing.assert_array_equal(m.fetch("P_smooth"), P.cpu().numpy())
        ran = m.em(max_iter=iters, tol=1e-5, want_smooth=True, want_P=False)
        q = [p.clone() for p in par]
        path, its, f2, _ = ctx.em_batch(panel, *q, max_iter=iters, tol=1e-5, may_have_missing=False)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(m.fetch("iters"), its.cpu().numpy())
        for k, p in zip(KEYS, q):
            np.testing.assert_array_equal(m.fetch(k), p.cpu().numpy())
        np.testing.assert_array_equal(m.fetch("f_smooth"), f2.cpu().numpy())
        assert ran == int(its.max().item())
    finally:
        m.close()


@pytest.mark.parametrize("N", [3000, 6400])
def test_very_wide_cross_section_with_missing_cells(ctx, N):
    """The C_t kernel of the wide path keeps 16 bytes of LDS per series (mask + R): N = 3000 still gets the compact rows the tile
    recursion reads (three stage buffers instead of five), N = 6400 falls back to the round-2 kernel with full rows."""
    import torch
    B, T, r = 2, 10, 18
    reps = [ko.synth_replicate(b, N, T, r, seed=5, missing=0.05) for b in range(B)]
    panel = np.stack([x for x, _ in reps])
    st = {k: np.stack([p[k] for _, p in reps]) for k in KEYS}
    dev = torch.device("cuda", ctx.device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    f, P, ll = ctx.ks_pass_batch(t(panel), *[t(st[k]) for k in KEYS], may_have_missing=True)
    torch.cuda.synchronize()
    _compare((f.cpu().numpy(), P.cpu().numpy(), ll.cpu().numpy()), _oracle(panel, st), f"N = {N}, r = {r}, missing cells")


@pytest.mark.parametrize("T,r,N,miss", [(1, 20, 40, 0.2), (2, 17, 300, 0.1), (5, 31, 64, 0.3), (3, 24, 258, 0.0)])
def test_tile_recursion_edges(ctx, T, r, N, miss):
    """recursion_tile_kernel at the edges of its loops: one and two periods (its operand sets alternate by the parity of the
    period and are re-loaded two periods ahead), the widest state it takes (r = 31: column 31 is the only padding), a panel declared
    with missing cells that has none (every row takes Cfull)."""
    import torch
    B = 3
    reps = [ko.synth_replicate(b, N, max(T, 6), r, seed=1234 + T, missing=miss) for b in range(B)]   # (a one-period panel cannot be standardised: cut a longer one)
    panel = np.ascontiguousarray(np.stack([x for x, _ in reps])[:, :T])
    st = {k: np.stack([p[k] for _, p in reps]) for k in KEYS}
    dev = torch.device("cuda", ctx.device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    f, P, ll = ctx.ks_pass_batch(t(panel), *[t(st[k]) for k in KEYS], may_have_missing=True)
    torch.cuda.synchronize()
    _compare((f.cpu().numpy(), P.cpu().numpy(), ll.cpu().numpy()), _oracle(panel, st), f"T = {T}, r = {r}")


# ---- observed factors beyond r_o + r_u = 8: the ordinary loadings step on the moments of z = (g, f) -------
@pytest.mark.parametrize("N,T,ru,ro,missing", [
    (60, 150, 6, 4, 0.0),      # width 10 -> 16, state Rp = 8, balanced panel (every series takes the shared inverse)
    (80, 160, 9, 3, 0.1),      # width 12 -> 16, state Rp = 16, per-series solves
    (300, 200, 12, 8, 0.05),   # width 20 -> 32, N > 256 (two series per lane)
    (40, 120, 2, 7, 0.0),      # more observed than latent factors
])
def test_em_with_many_observed_factors_matches_the_oracle(ctx, N, T, ru, ro, missing):
    from oracle import obs_oracle as oo
    B, iters = 2, 4
    reps = [oo.synth_obs(300 + b, N, T, ru, ro, missing=missing) for b in range(B)]
    panel = np.stack([x for x, _, _ in reps]); G = np.stack([g for _, g, _ in reps])
    st = {k: np.stack([p[k] for _, _, p in reps]) for k in KEYS}
    new, path, its, f, P = ctx.em_obs_batch_host(panel, G, *[st[k] for k in KEYS], max_iter=iters, tol=0.0)
    for b in range(B):
        p, opath, out = oo.em_obs(panel[b], G[b], {k: st[k][b] for k in KEYS}, max_iter=iters, tol=0.0)
        np.testing.assert_allclose(path[b], opath, rtol=1e-9, err_msg=f"loglik path b={b}")
        for k in KEYS:
            assert np.abs(new[k][b] - p[k]).max() <= 1e-8 * max(1.0, np.abs(p[k]).max()), (k, b, np.abs(new[k][b] - p[k]).max())
        assert np.abs(f[b] - out["f_smooth"]).max() <= 1e-8 * np.abs(out["f_smooth"]).max()
    # a series with fewer than r_o + r_u + 1 observed cells keeps its parameters (as the narrow kernel and the oracle)
    if missing > 0.0:
        x2 = panel.copy()
        x2[:, ro + ru - 2:, 5] = np.nan
        new2, *_ = ctx.em_obs_batch_host(x2, G, *[st[k] for k in KEYS], max_iter=2, tol=0.0)
        assert np.array_equal(new2["Lam"][:, 5], st["Lam"][:, 5]) and np.array_equal(new2["R"][:, 5], st["R"][:, 5])
        for b in range(B):
            p, _, _ = oo.em_obs(x2[b], G[b], {k: st[k][b] for k in KEYS}, max_iter=2, tol=0.0)
            assert np.abs(new2["Lam"][b] - p["Lam"]).max() <= 1e-8 * max(1.0, np.abs(p["Lam"]).max())
    from dynamic_factor_models_amd import DfmError
    with pytest.raises(DfmError):                                  # r_o + r_u = 33
        ctx.em_obs_batch_host(np.zeros((1, 50, 40)), np.zeros((1, 50, 17)), np.zeros((1, 40, 33)), np.ones((1, 40)),
                              np.zeros((1, 16, 16)), np.eye(16)[None], np.zeros((1, 16)), np.eye(16)[None], max_iter=1)


# ---- balanced wide states at every padding width: cov_tile_kernel's block pivots (ceil(r / 4) of 8), the mean scan's state pieces
# (ceil(r / 8) of 4), the unpadded b_t rows of the streaming collapse -- pass and three EM iterations against the oracle ----------
@pytest.mark.parametrize("N,T,r", [(64, 140, 17), (300, 90, 21), (64, 70, 24), (520, 60, 25), (64, 130, 31), (96, 80, 32), (64, 1, 20), (64, 2, 18)])
def test_balanced_wide_states_at_every_width(ctx, N, T, r):
    from test_gpu_ks_pass import _batch, _run_dev
    Tg = max(T, 8)
    panel, st = _batch(3, N, Tg, r, 0.0, seed=ko.SEED0 + 31 * r + N)
    panel = np.ascontiguousarray(panel[:, :T])
    _compare(_run_dev(ctx, panel, st, may_have_missing=False), _oracle(panel, st), f"balanced N={N} T={T} r={r}")
    if T >= 60:
        new, gpath, _, _, _ = ctx.em_batch_host(panel, *[st[k] for k in KEYS], max_iter=3, tol=0.0, may_have_missing=False)
        for b in range(panel.shape[0]):
            p, path, _ = ko.em(panel[b], {k: st[k][b] for k in KEYS}, max_iter=3, tol=0.0)
            np.testing.assert_allclose(gpath[b], path, rtol=1e-9)
            for k in KEYS:
                assert np.abs(new[k][b] - p[k]).max() <= 1e-8 * max(1.0, np.abs(p[k]).max()), (k, b, r, np.abs(new[k][b] - p[k]).max())


# ---- BASELINE config 4 at the BENCHMARKED batch (256 replicates, one workgroup of every persistent kernel per CU): scattered
# replicates with their own parameters against the oracle -- the pass (cov_tile + collapse_wide2 + meanscan_mfma + pfill) and two EM
# iterations (+ em_update_grid + mstep_wide) ---------------------------------------------------------------------------------------
def test_config4_at_the_benchmarked_batch_against_the_oracle(ctx):
    import torch
    B, N, T, r = 256, 1000, 2000, 20
    panel, par = ctx.synth_panels(404, 0, B, T, N, r)
    f, P, ll = ctx.ks_pass_batch(panel, *par, may_have_missing=False)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ll).all())
    ix = torch.tensor([0, 1, 37, 128, 255], device=panel.device)
    x = _take(panel, ix)
    st = dict(zip(KEYS, [_take(p, ix) for p in par]))
    _compare((_take(f, ix), _take(P, ix), _take(ll, ix)), _oracle(x, st), "config 4 at B = 256")
    q = [p.clone() for p in par]
    path, its, _, _ = ctx.em_batch(panel, *q, max_iter=2, tol=0.0, want_smooth=False, may_have_missing=False)
    torch.cuda.synchronize()
    new = dict(zip(KEYS, [_take(p, ix) for p in q]))
    gp = _take(path, ix)
    for k in range(5):                                             # (all five scattered replicates: the oracle's C twin affords it)
        p, opath = _em_c(x[k], {kk: st[kk][k] for kk in KEYS}, 2)
        np.testing.assert_allclose(gp[k], opath, rtol=1e-9)
        for kk in KEYS:
            assert np.abs(new[kk][k] - p[kk]).max() <= 1e-8 * max(1.0, np.abs(p[kk]).max()), (kk, k, np.abs(new[kk][k] - p[kk]).max())
    del panel, f, P
    torch.cuda.empty_cache()


def test_balanced_odd_n_beyond_the_tiling_on_the_general_path(ctx):
    """A BALANCED odd-N panel with singular_q=True takes the general path without the may-have-missing flag; the library appends
    an all-missing series, so the padded problem has a missing cell in every period -- it must carry the flag itself (the C_t
    array used to be absent there: a GPU fault instead of the DFM_E_DIMS of round 3)."""
    import torch
    B, N, T, r = 2, 301, 40, 20
    reps = [ko.synth_replicate(b, N, T, r, seed=ko.SEED0 + 5, missing=0.0) for b in range(B)]
    panel = np.stack([x for x, _ in reps])
    st = {k: np.stack([p[k] for _, p in reps]) for k in KEYS}
    dev = torch.device("cuda", ctx.device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    f, P, ll = ctx.ks_pass_batch(t(panel), *[t(st[k]) for k in KEYS], may_have_missing=False, singular_q=True)
    ctx.synchronize()
    torch.cuda.synchronize()
    _compare((f.cpu().numpy(), P.cpu().numpy(), ll.cpu().numpy()), _oracle(panel, st), "balanced odd N, covariance form")


def test_a_sparse_series_is_updated_on_the_wide_loadings_route(ctx):
    """Plain EM on the matrix-pipe loadings step (Rp > 8: mmw_finish_kernel): a series with FEWER observed cells than r + 1 is
    still updated, as in the oracle (sum E[f f'] includes P_t: positive definite with one observed cell) and in mstep_lam_kernel --
    the r + 1 rule belongs to the observed-factor joint regression only."""
    import torch
    B, N, T, r = 2, 300, 40, 20
    reps = [ko.synth_replicate(b, N, T, r, seed=78, missing=0.05) for b in range(B)]
    panel = np.stack([x for x, _ in reps])
    panel[:, 3:, 7] = np.nan                                     # series 7: three observed cells (< r + 1)
    panel[:, :3, 7] = np.nan_to_num(panel[:, :3, 7])
    st = {k: np.stack([p[k] for _, p in reps]) for k in KEYS}
    dev = torch.device("cuda", ctx.device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    par = [t(st[k]) for k in KEYS]
    path, its, _, _ = ctx.em_batch(t(panel), *par, max_iter=1, tol=0.0, want_smooth=False, may_have_missing=True)
    ctx.synchronize()
    torch.cuda.synchronize()
    for b in range(B):
        p, opath, _ = ko.em(panel[b], {k: st[k][b] for k in KEYS}, max_iter=1, tol=0.0)
        np.testing.assert_allclose(path[b].cpu().numpy(), opath, rtol=1e-9)
        Lam = par[0][b].cpu().numpy(); R = par[1][b].cpu().numpy()
        assert np.abs(Lam - p["Lam"]).max() <= 1e-8 * max(1.0, np.abs(p["Lam"]).max())
        assert np.abs(R - p["R"]).max() <= 1e-8 * max(1.0, np.abs(p["R"]).max())
        assert not np.array_equal(Lam[7], st["Lam"][b][7])
