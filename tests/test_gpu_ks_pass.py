"""GPU parity tests of the hot path proper: HIP Kalman-smoother pass (through the C-ABI) vs the
CPU oracle on the same seeded inputs.  fp64: tolerance 1e-9 relative (north_star asks 1e-6)."""
import numpy as np
import pytest

from conftest import diag_only

from oracle import c_oracle as co
from oracle import kalman_oracle as ko

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def _batch(B, N, T, r, missing=0.0, seed=ko.SEED0, blank_rows=()):
    reps = [ko.synth_replicate(b, N, T, r, seed=seed, missing=missing) for b in range(B)]
    panel = np.stack([x for x, _ in reps])
    for t in blank_rows:
        panel[:, t, :] = np.nan
    st = {k: np.stack([p[k] for _, p in reps]) for k in reps[0][1]}
    return panel, st


def _oracle(panel, st):
    return co.ks_pass_batch(panel, st["Lam"], st["R"], st["A"], st["Q"], st["mu0"], st["P0"])


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from dynamic_factor_models_amd import DfmContext
    c = DfmContext()
    yield c
    c.close()


def _run_dev(ctx, panel, st, **kw):
    import torch
    dev = torch.device("cuda", ctx.device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    f, P, ll = ctx.ks_pass_batch(t(panel), t(st["Lam"]), t(st["R"]), t(st["A"]), t(st["Q"]), t(st["mu0"]),
                                 t(st["P0"]), **kw)
    torch.cuda.synchronize()
    return f.cpu().numpy(), None if P is None else P.cpu().numpy(), ll.cpu().numpy()


def _compare(got, ref, tag=""):
    f, P, ll = got
    fo, Po, llo = ref
    assert np.all(np.isfinite(ll)), tag
    np.testing.assert_allclose(ll, llo, rtol=RTOL, err_msg=f"loglik {tag}")
    scale_f = np.abs(fo).max()
    assert np.abs(f - fo).max() <= RTOL * scale_f, f"f_smooth {tag}: {np.abs(f - fo).max()}"
    if P is not None:
        assert np.abs(P - Po).max() <= RTOL * np.abs(Po).max(), f"P_smooth {tag}: {np.abs(P - Po).max()}"


@pytest.mark.parametrize("B,N,T,r,missing", [
    (16, 200, 500, 8, 0.0),      # BASELINE config 2 shape, host-checked subset
    (13, 200, 500, 8, 0.1),      # 10 %-missing variant; B not a multiple of the 8 replicates per wave
    (5, 30, 60, 4, 0.15),
    (3, 139, 222, 4, 0.05),      # Stock-Watson "All" panel shape (config 1)
    (4, 31, 41, 3, 0.0),         # odd N (8-byte loads), r padded 3 -> 4
    (4, 31, 41, 3, 0.2),
    (3, 20, 25, 1, 0.1),         # r = 1 (padded to 2)
    (3, 50, 40, 5, 0.1),         # r padded 5 -> 8
    (2, 64, 48, 12, 0.1),        # r padded 12 -> 16
    (2, 100, 40, 20, 0.05),      # config 4's r = 20 (padded to 32)
    (2, 256, 30, 8, 0.0),        # widest N of the 2-chunk tiling
    (2, 300, 20, 8, 0.1),        # 4-chunk tiling
    # collapse_miss.hip (Rp = 8, even N <= 224): compacted lists of missing / observed series, partial steps, short panels
    (3, 198, 61, 8, 0.02),       # N not a multiple of 8 (a clamped MFMA tail), T = 1 mod 4
    (3, 200, 64, 8, 0.6),        # more missing than observed: the observed list
    (2, 200, 33, 7, 0.97),       # almost nothing observed (some periods with no cell at all)
    (4, 26, 9, 5, 0.3),          # fewer periods than a wave's segment, r padded 5 -> 8
    (2, 224, 40, 8, 0.1),        # widest cross-section of this kernel; (226: collapse_kernel)
    (2, 226, 40, 8, 0.1),
    (3, 64, 3, 8, 0.25),         # T = 3
    # Rp = 32 beyond collapse_kernel's register tiling (N > 256): collapse_wide2 in its variant for missing cells (NaN -> 0 on the
    # matrix pipe, per-period s_t / n_t / log det R_t) + ct_miss_wide_kernel (C_t of the periods with a missing cell)
    (2, 300, 40, 20, 0.1),       # tiles of 16 periods: 2 full + 8; stages of 32 series: 9 full + 12
    (3, 1000, 130, 20, 0.1),     # config 4's cross-section with missing cells; one tile of 128 periods + 2
    (2, 258, 33, 17, 0.3),       # just past the tiling's limit, r padded 17 -> 32
    (2, 400, 20, 32, 0.05),      # every factor column in use
    (2, 512, 70, 25, 0.002),     # most periods complete (they keep C_full), a few with one missing cell
    (2, 260, 18, 20, 0.97),      # almost nothing observed
])
def test_pass_matches_oracle(ctx, B, N, T, r, missing):
    panel, st = _batch(B, N, T, r, missing)
    got = _run_dev(ctx, panel, st)
    _compare(got, _oracle(panel, st), f"B={B} N={N} T={T} r={r} miss={missing}")


def test_rows_with_every_cell_missing(ctx):
    panel, st = _batch(4, 40, 50, 4, 0.1, blank_rows=(0, 7, 8, 49))
    _compare(_run_dev(ctx, panel, st), _oracle(panel, st), "blank rows")


def test_missing_only_at_the_edges(ctx):
    # unbalanced like the Stock-Watson panel: series that start late / end early; long balanced middle
    panel, st = _batch(9, 60, 120, 4, 0.0)
    panel[:, :10, 50:] = np.nan
    panel[:, -5:, :7] = np.nan
    _compare(_run_dev(ctx, panel, st), _oracle(panel, st), "ragged edges")


def test_nondiagonal_dynamics_from_em(ctx):
    # parameters after a few oracle EM steps: full A, full Q, nonzero mu0, general P0
    B, N, T, r = 3, 40, 80, 3
    panel, st = _batch(B, N, T, r, 0.1)
    for b in range(B):
        p0, _ = ko.pca_init(np.nan_to_num(panel[b]), r)
        p, _, _ = ko.em(panel[b], p0, 4)
        for k in st:
            st[k][b] = p[k]
    _compare(_run_dev(ctx, panel, st), _oracle(panel, st), "EM params")


def test_host_pointer_entry_and_no_P(ctx):
    panel, st = _batch(3, 30, 40, 4, 0.1)
    ref = _oracle(panel, st)
    f, P, ll = ctx.ks_pass_batch_host(panel, st["Lam"], st["R"], st["A"], st["Q"], st["mu0"], st["P0"])
    _compare((f, P, ll), ref, "host entry")
    f2, P2, ll2 = _run_dev(ctx, panel, st, want_P=False)
    assert P2 is None
    _compare((f2, None, ll2), ref, "no P")


def test_nan_without_flag_is_an_error(ctx):
    from dynamic_factor_models_amd import DfmError
    panel, st = _batch(2, 20, 30, 2, 0.1)
    with pytest.raises(DfmError) as ei:
        ctx.ks_pass_batch_host(panel, st["Lam"], st["R"], st["A"], st["Q"], st["mu0"], st["P0"],
                               may_have_missing=False)
    assert ei.value.code == -4


def test_deterministic(ctx):
    panel, st = _batch(8, 64, 100, 8, 0.05)
    a = _run_dev(ctx, panel, st)
    b = _run_dev(ctx, panel, st)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


def test_linearity_in_the_data_full_size(ctx):
    """Size-independent property at BASELINE config-2 size: with mu0 = 0 the smoothed mean is linear
    in the panel and the smoothed covariance does not depend on it."""
    import torch
    B, N, T, r = 64, 200, 500, 8
    g = torch.Generator(device="cuda").manual_seed(1)
    dev = torch.device("cuda", ctx.device)
    panel = torch.randn((B, T, N), dtype=torch.float64, device=dev, generator=g)
    _, st = _batch(1, N, T, r)
    rep = lambda a: torch.from_numpy(a).to(dev).expand(B, *a.shape[1:]).contiguous()
    args = [rep(st[k]) for k in ("Lam", "R", "A", "Q", "mu0", "P0")]
    f1, P1, _ = ctx.ks_pass_batch(panel, *args, may_have_missing=False)
    f2, P2, _ = ctx.ks_pass_batch(2.5 * panel, *args, may_have_missing=False)
    f3, _, _ = ctx.ks_pass_batch(panel + panel.flip(0), *args, may_have_missing=False)
    torch.cuda.synchronize()
    assert torch.allclose(f2, 2.5 * f1, rtol=1e-10, atol=1e-12)
    assert torch.allclose(f3, f1 + f1.flip(0), rtol=1e-10, atol=1e-12)
    assert torch.equal(P1, P2)
    assert torch.allclose(P1[0], P1[-1], rtol=0, atol=0)


# ---- balanced panels: the fast path (collapse on the matrix pipe / VALU LDS-DMA kernel, time-parallel scan) ------
BALANCED_SHAPES = [
    (5, 40, 50, 4), (3, 64, 48, 12), (4, 30, 41, 3), (3, 20, 25, 1), (3, 50, 7, 2), (2, 20, 3, 2), (2, 20, 2, 4),
    (9, 200, 222, 8),          # Stock-Watson window length, headline cross-section
    (3, 38, 33, 8),            # N not a multiple of the 8 series of an MFMA step (clamped last step)
    (2, 130, 37, 5),           # two DMAs per period, r padded 5 -> 8, T = 1 mod 4
    (2, 512, 19, 3),           # four DMAs per period, 16 series per step, 32 steps
    (3, 6, 9, 2), (2, 4, 3, 1),
    (2, 400, 50, 16),          # too many MFMA steps for the register file: VALU kernel
    (2, 1000, 64, 8),          # 8N > 4 KiB rows: VALU kernel, 8-chunk tiling
    (2, 100, 40, 20),          # r padded to 32: VALU kernel
    (2, 600, 70, 12),          # beyond every register tiling (N > 512 at r > 8): wide kernel, r padded to 16
    (2, 300, 37, 20),          # wide kernel, r padded to 32, T not a multiple of the 16-period tile
    (3, 31, 41, 3),            # odd N: wide kernel (8-byte loads)
    (1, 1000, 2000, 20),       # BASELINE config 4's shape (N = 1000, T = 2000, r = 20), one replicate
    # collapse_wide2 (Rp = 32, even N): stages of 32 series / tiles of 128 periods with partial tails, 16x16x4 + NX 4x4x4
    # MFMAs for r = 17..28, two 16x16x4 for r = 29..32; cov_grid_kernel<32>; meanscan_mfma_kernel (128 chunks of L steps)
    (2, 66, 65, 17),           # 2 series past two stages, r padded 17 -> 32 (NX = 1)
    (3, 130, 70, 20),          # 2 series past four stages (a partial MFMA step: 2 of 4 series)
    (2, 64, 64, 32),           # exact stages, every factor column used (NX = 4)
    (2, 34, 3, 24),            # a tile of 3 periods, NX = 2, scan chunks of one step (most of them empty)
    (2, 1280, 40, 20),
    (2, 1400, 40, 20),
    (17, 96, 130, 21),         # XCD-ordered tile queues with B not a multiple of 8; one full tile + 2 periods; NX = 2
    (9, 250, 300, 25),         # one flat tile queue (B < 16); NX = 3; three tiles; scan chunks of 4 steps
    (3, 62, 129, 29),          # NX = 4 with padding columns; stages of 32 + 30 series
    (2, 40, 5, 18),            # T far below the 128 chunks of the scan
    (2, 101, 45, 20),          # odd N: collapse_wide_kernel (8-byte loads) feeding the matrix-pipe scan
    (20, 34, 260, 32),         # uneven replicate counts per XCD queue (8 + 8 + 4), r = 32
    # narrow states (Rp <= 8) with rows beyond the 4-KB row ring of the MFMA collapse: collapse_wide2 computes 16 columns, stores Rp
    (3, 600, 130, 4), (2, 1024, 40, 2), (17, 700, 90, 7), (2, 514, 5, 8),
    # cov_grid_kernel<16> (Rp = 16: 256 threads per replicate)
    (3, 200, 120, 12), (2, 48, 33, 9), (2, 64, 500, 16),
]


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


@pytest.mark.parametrize("B,N,T,r", BALANCED_SHAPES)
def test_balanced_fast_path_matches_oracle(ctx, B, N, T, r):
    panel, st = _batch(B, N, T, r, 0.0)
    ref = _oracle(panel, st)
    _compare(_run_dev(ctx, panel, st, may_have_missing=False), ref, f"fast B={B} N={N} T={T} r={r}")


_D = lambda **env: pytest.param(env, marks=diag_only())           # (switches only the diagnostics build reads)


@pytest.mark.parametrize("env", [_D(DFM_COLLAPSE_VARIANT=199), _D(DFM_COLLAPSE_VARIANT=198), _D(DFM_NO_FUSE_COV=1), _D(DFM_COLLAPSE_WPR=1),
                                 _D(DFM_COLLAPSE_WPR=3), _D(DFM_COLLAPSE_WPR=7), dict(DFM_PASS_FUSED=0), dict(DFM_FORCE_GENERAL=1)])
def test_balanced_kernel_choices_agree(env):
    """The VALU collapse, the wide collapse, the MFMA collapse with its covariance workgroups as separate launches,
    with 1/3/7 period segments per replicate, and the general
    (sequential) path are the same function of the inputs."""
    c = _ctx_with_env(**env)
    try:
        for (B, N, T, r) in [(5, 200, 500, 8), (3, 64, 48, 12), (4, 30, 41, 3), (3, 50, 7, 2), (2, 130, 37, 5)]:
            panel, st = _batch(B, N, T, r, 0.0)
            _compare(_run_dev(c, panel, st, may_have_missing=False), _oracle(panel, st), f"{env} N={N} T={T} r={r}")
    finally:
        c.close()


@pytest.mark.parametrize("env", [dict(), dict(DFM_NO_CHUNK=1), dict(DFM_NO_CHUNK=1, DFM_NO_PAIR=1), _D(DFM_NO_CHUNK=1, DFM_PAIR_BMAX=3),
                                 dict(DFM_CHUNK_W=12), dict(DFM_CHUNK_W=2), dict(DFM_CHUNK_TOL=1e-30)])
def test_sequential_kernel_choices_agree(env):
    """Panels with missing cells at Rp = 8: the time-chunked recursion (recursion_chunk.hip, the default since round 5; W = 12: a
    longer warm-up; W = 2 or a tolerance nothing meets: every replicate fails its boundary check and is redone by the sequential
    kernel behind it), the covariance-wave + mean-wave pair (recursion_pair.hip, DFM_NO_CHUNK=1: the default of rounds 3-4 up to one
    replicate per SIMD) and the one-wave-per-replicate kernel are the same function of the inputs -- dense chunks (every period
    with a missing cell), mixed chunks, fully observed stretches that reach the steady state, r = 3 padded to the 8-wide state, T not
    a multiple of the chunk."""
    c = _ctx_with_env(**env)
    try:
        for (B, N, T, r, miss) in [(5, 200, 500, 8, 0.1), (4, 64, 203, 8, 0.01), (6, 30, 41, 3, 0.3), (3, 120, 257, 5, 0.0005),
                                   (2, 50, 7, 2, 0.2), (2, 20, 5750, 8, 0.05), (2, 20, 5770, 8, 0.05)]:   # (5760 periods: the pair kernel's LDS limit)
            panel, st = _batch(B, N, T, r, miss)
            _compare(_run_dev(c, panel, st, may_have_missing=True), _oracle(panel, st), f"{env} N={N} T={T} r={r} miss={miss}")
    finally:
        c.close()


def test_config4_full_size_properties(ctx):
    """BASELINE config 4 at full size (N = 1000, T = 2000, r = 20, 256 replicates; 4.1 GB of panels): the
    size-independent properties of the pass -- linearity of the smoothed mean in the data (mu0 = 0), data-independent
    smoothed covariance -- plus exact agreement of one replicate with the same replicate run alone."""
    import torch
    B, N, T, r = 256, 1000, 2000, 20
    dev = torch.device("cuda", ctx.device)
    g = torch.Generator(device="cuda").manual_seed(4)
    panel = torch.randn((B, T, N), dtype=torch.float64, device=dev, generator=g)
    _, st = _batch(1, N, T, r)
    rep = lambda a: torch.from_numpy(a).to(dev).expand(B, *a.shape[1:]).contiguous()
    args = [rep(st[k]) for k in ("Lam", "R", "A", "Q", "mu0", "P0")]
    f1, P1, ll1 = ctx.ks_pass_batch(panel, *args, may_have_missing=False)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ll1).all())
    one = [a[:1].contiguous() for a in args]
    f0, P0, ll0 = ctx.ks_pass_batch(panel[17:18].contiguous(), *one, may_have_missing=False)
    assert torch.equal(f0[0], f1[17]) and torch.equal(P0[0], P1[17]) and torch.equal(ll0[0], ll1[17])
    panel.mul_(-2.0)
    f2, P2, _ = ctx.ks_pass_batch(panel, *args, may_have_missing=False)
    torch.cuda.synchronize()
    assert torch.allclose(f2, -2.0 * f1, rtol=1e-10, atol=1e-12)
    assert torch.equal(P1, P2) and torch.equal(P1[0], P1[-1])


# ---- BASELINE config 3's per-GPU shard: 8192 replicates on one GPU (SURVEY row g) ---------------------------------
def _scattered(B, n, seed=3):
    """Replicate indices spread over the batch: both ends, both sides of the boundaries of the launch rounds (at B = 8192
    the MFMA collapse runs one period segment per replicate -- 4 replicates per workgroup, 512 resident workgroups, i.e.
    rounds of 2048 replicates), and a seeded random rest."""
    edge = [0, 1, 2, 3, 4, 7, 8, B - 1, B - 2, B - 4, B - 5, B - 8, B - 9]
    for m in range(2048, B, 2048):
        edge += [m - 5, m - 4, m - 1, m, m + 1, m + 3, m + 4]
    rng = np.random.default_rng(seed)
    rest = rng.choice(B, size=max(0, n - len(set(edge))), replace=False).tolist()
    return sorted(set(i for i in edge + rest if 0 <= i < B))[: max(n, len(set(edge)))]


def test_config3_shard_8192_replicates_against_the_oracle(ctx):
    """B = 8192, N = 200, T = 500, r = 8 balanced (6.6 GB of panels generated on the device): >= 64 replicates scattered
    over the batch against the C oracle at 1e-9, every log-likelihood finite, P_smooth data-independent."""
    import torch
    B, N, T, r = 8192, 200, 500, 8
    panel, par = ctx.synth_panels(20160415, 0, B, T, N, r)
    f, P, ll = ctx.ks_pass_batch(panel, *par, may_have_missing=False)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ll).all())
    idx = _scattered(B, 72)
    assert len(idx) >= 64
    ix = torch.tensor(idx, device=panel.device)
    h = lambda a: a.index_select(0, ix).cpu().numpy()
    st = dict(zip(("Lam", "R", "A", "Q", "mu0", "P0"), [h(p) for p in par]))
    ref = _oracle(h(panel), st)
    _compare((h(f), h(P), h(ll)), ref, "B=8192 scattered replicates")
    # same replicates in a small batch (different launch geometry: 3 segments per replicate instead of 1): same numbers
    f2, P2, ll2 = ctx.ks_pass_batch(panel.index_select(0, ix).contiguous(), *[p.index_select(0, ix).contiguous() for p in par],
                                    may_have_missing=False)
    torch.cuda.synchronize()
    _compare((f2.cpu().numpy(), P2.cpu().numpy(), ll2.cpu().numpy()), ref, "same replicates, small batch")
    del f, P, f2, P2, panel
    torch.cuda.empty_cache()


def test_config3_shard_with_missing_cells(ctx):
    """The same shard size on the sequential path (10 % of the cells missing; lane-group recursion at B > 4096)."""
    import torch
    B, N, T, r = 8192, 200, 500, 8
    panel, par = ctx.synth_panels(7, 100000, B, T, N, r, missing_prob=0.1)
    f, P, ll = ctx.ks_pass_batch(panel, *par, may_have_missing=True)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ll).all())
    idx = _scattered(B, 32, seed=5)[:40]
    ix = torch.tensor(idx, device=panel.device)
    h = lambda a: a.index_select(0, ix).cpu().numpy()
    st = dict(zip(("Lam", "R", "A", "Q", "mu0", "P0"), [h(p) for p in par]))
    _compare((h(f), h(P), h(ll)), _oracle(h(panel), st), "B=8192, 10 % missing")
    del f, P, panel
    torch.cuda.empty_cache()


# ---- BASELINE config 4: several DISTINCT replicates against the oracle -------------------------------------------
def test_config4_eight_replicates_against_the_oracle(ctx):
    B, N, T, r = 8, 1000, 2000, 20
    panel, st = _batch(B, N, T, r, 0.0)
    ref = _oracle(panel, st)
    _compare(_run_dev(ctx, panel, st, may_have_missing=False), ref, "config 4, 8 distinct replicates")


# ---- slowly converging Riccati recursion through the balanced (fixed-point detecting) path ------------------------
def _slow_riccati(B, N, T, r, rho, Rscale, seed=11):
    """Near-unit-root factors observed through very noisy series: C = Lam' R^-1 Lam is small, the closed-loop matrix
    A (I - K Lam) stays close to A, and the covariance recursion needs hundreds of steps to settle (or never does
    within T): the fast path's fixed-point cut E approaches T."""
    rng = np.random.default_rng(seed)
    a = np.linspace(rho, rho - 0.004 * (r - 1), r)
    A = np.diag(a) + 1e-3 * rng.standard_normal((r, r))
    Q = 0.02 * np.eye(r) + 0.002 * np.ones((r, r))
    panels, sts = [], []
    for b in range(B):
        Lam = rng.standard_normal((N, r))
        R = Rscale * rng.uniform(0.5, 1.5, N)
        f = rng.standard_normal(r)
        x = np.empty((T, N))
        for t in range(T):
            f = A @ f + np.linalg.cholesky(Q) @ rng.standard_normal(r)
            x[t] = Lam @ f + np.sqrt(R) * rng.standard_normal(N)
        panels.append(x)
        sts.append(dict(Lam=Lam, R=R, A=A.copy(), Q=Q.copy(), mu0=0.1 * rng.standard_normal(r), P0=np.eye(r) * (1.0 + b)))
    return np.stack(panels), {k: np.stack([s[k] for s in sts]) for k in sts[0]}


@pytest.mark.parametrize("N,T,r,rho,Rscale", [
    (20, 500, 4, 0.999, 1e3),      # E close to (or equal to) T: the whole pass is "transient"
    (20, 500, 8, 0.999, 2e2),      # headline state width
    (200, 500, 8, 0.995, 5e3),     # headline shape, MFMA collapse
    (30, 500, 3, 0.99, 50.0),      # E of a few hundred: long transient + short steady stretch
    (16, 120, 2, 0.9999, 1e4),     # T shorter than the convergence time
    (40, 300, 20, 0.995, 2e2),     # Rp = 32: a long transient on wave 0 of meanscan_mfma_kernel, then its chunked scans
    (40, 150, 18, 0.9999, 1e4),    # Rp = 32 with no steady stretch at all
])
def test_slow_riccati_on_the_balanced_path(ctx, N, T, r, rho, Rscale):
    panel, st = _slow_riccati(3, N, T, r, rho, Rscale)
    ref = _oracle(panel, st)
    _compare(_run_dev(ctx, panel, st, may_have_missing=False), ref, f"slow Riccati N={N} T={T} r={r} rho={rho}")
    c = _ctx_with_env(DFM_FORCE_GENERAL=1)     # and the sequential path on the same inputs
    try:
        _compare(_run_dev(c, panel, st, may_have_missing=False), ref, "slow Riccati, sequential path")
    finally:
        c.close()
