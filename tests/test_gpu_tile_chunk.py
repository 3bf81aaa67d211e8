"""GPU parity tests of the time chunks of recursion_tile_kernel (recursion_tile.hip, wide states padded to 32 with missing cells --
BASELINE config 4's class): a replicate's periods cut into chunks that warm up over a few periods, each chunk a workgroup of its
own, boundaries checked, the sequential instantiation behind them for replicates whose boundaries do not agree.  The chunked and
the sequential kernel must be the same function of the inputs; both are compared with the C oracle (1e-9 / 1e-8 for EM)."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import kalman_oracle as ko

from test_gpu_ks_pass import _batch, _compare, _ctx_with_env, _oracle, _run_dev

pytestmark = pytest.mark.gpu
KEYS = ("Lam", "R", "A", "Q", "mu0", "P0")

SHAPES = [
    (3, 300, 400, 20, 0.1),     # automatic: 6 chunks of 68 periods (the last of 60)
    (2, 64, 333, 17, 0.2),      # narrow cross-section (collapse_kernel's per-period arrays, full C_t rows), odd T: 4 chunks
    (2, 1000, 150, 20, 0.1),    # config 4's cross-section: 2 chunks of 76 / 74
    (2, 40, 301, 31, 0.3),      # the widest state the kernel takes (column 31 is the only padding); 28 observed series for 31 factors
                                # forget slowly: the boundary checks reject the chunks, the sequential kernel redoes both replicates
    (2, 120, 301, 31, 0.3),     # ... enough series: the chunks stand
    (2, 258, 131, 24, 0.0),     # declared with missing cells, has none: every period takes C_full
    (2, 300, 40, 20, 0.1),      # too short for two chunks: the sequential kernel whatever was asked
]


def _num_cu():
    import torch
    return int(torch.cuda.get_device_properties(0).multi_processor_count)


def _chunks(B, T, env, num_cu=256):
    """The chunk plan of launch_recursion_tile (recursion_tile.hip).  Round 6: recursion_tile1_kernel -- one wave per (replicate, chunk), a
    wave on every SIMD -- for batches whose chunks can fill the SIMDs (ceil(4 CUs / B) <= 16) and for every forced count; smaller batches
    keep four waves per chunk (two workgroups per CU).  0 = the sequential kernel, no boundary bookkeeping (DFM_TILE_NC=1, or the
    four-wave plan with one chunk); 1 = one chunk per replicate through tile_chunk_finish_kernel (every replicate counted, none redone)."""
    W = min((int(env.get("DFM_TILE_W", 16)) + 1) & ~1, 32)
    nc = int(env.get("DFM_TILE_NC", 0))
    if nc == 1:
        return 0
    one_wave = nc > 1 or (4 * num_cu + B - 1) // B <= 16
    want = min(nc if nc > 1 else ((4 * num_cu + B - 1) // B if one_wave else (2 * num_cu) // B), 16)
    while want > 1:
        lc = 2 * ((T + 2 * want - 1) // (2 * want))
        if lc >= 4 * W and T - (want - 1) * lc >= W + 2:
            return want
        want -= 1
    return 1 if one_wave else 0


@pytest.mark.parametrize("env,expect_fail", [
    (dict(), 0),                                   # automatic count
    (dict(DFM_TILE_NC=1), None),                   # the sequential kernel
    (dict(DFM_TILE_NC=2), 0),
    (dict(DFM_TILE_NC=3, DFM_TILE_W=8), None),     # odd count, a shorter warm-up (some replicates may not have forgotten: redone)
    (dict(DFM_TILE_NC=7, DFM_TILE_W=30), 0),       # the longest warm-up the scratch holds (chunks of >= 120 periods: fewer than asked)
    (dict(DFM_TILE_W=2, DFM_CHUNK_TOL=1e-30), "some"),   # two periods of warm-up and a tolerance only bit-equal states meet (at N = 300 the
                                                   # default warm-up does arrive at bit-equal states): replicates are redone by the sequential kernel
])
def test_chunked_and_sequential_tile_recursion_agree(env, expect_fail):
    c = _ctx_with_env(**env)
    redone = 0
    try:
        for (B, N, T, r, miss) in SHAPES:
            panel, st = _batch(B, N, T, r, miss)
            _compare(_run_dev(c, panel, st, may_have_missing=True), _oracle(panel, st), f"{env} N={N} T={T} r={r} miss={miss}")
            nf, nt = c.chunk_fallbacks()
            cnt = _chunks(B, T, env, _num_cu())
            if cnt == 0:
                assert nt == 0, (env, T, nf, nt)
            elif cnt == 1:
                assert (nf, nt) == (0, B), (env, T, nf, nt)
            elif expect_fail == "some":
                assert nt == B, (env, T, nf, nt)
                redone += nf
            elif expect_fail == 0 and N == 40:
                assert nt == B and nf > 0, (env, T, nf, nt)
            elif expect_fail == 0:
                assert (nf, nt) == (0, B), (env, T, nf, nt)
            else:
                assert nt == B, (env, T, nf, nt)
        assert expect_fail != "some" or redone >= 4, redone
    finally:
        c.close()


def test_slow_forgetting_goes_to_the_sequential_kernel():
    """A filter that forgets slowly (a near-unit-root factor seen through almost nothing: loadings 1e-3, unit idiosyncratic variance)
    does not meet the boundary check after 16 periods: the replicates concerned are flagged and redone, the others are not."""
    B, N, T, r = 4, 40, 300, 17
    panel, st = _batch(B, N, T, r, 0.1)
    for b in (1, 3):
        st["Lam"][b] *= 1e-3
        st["R"][b][:] = 1.0
        st["A"][b] = 0.999 * np.eye(r)
        st["Q"][b] = 1e-3 * np.eye(r)
    c = _ctx_with_env()
    try:
        _compare(_run_dev(c, panel, st, may_have_missing=True), _oracle(panel, st), "slowly forgetting replicates")
        nf, nt = c.chunk_fallbacks()
        assert nt == B and nf == 2, (nf, nt)
    finally:
        c.close()


@pytest.mark.parametrize("env", [dict(), dict(DFM_TILE_NC=1), dict(DFM_TILE_NC=3), dict(DFM_TILE_W=2, DFM_CHUNK_TOL=1e-30)])
def test_em_on_chunks(env):
    """Two EM iterations at Rp = 32 with missing cells: the chunks' parts of the sums (sum P_t, sum U_t), the products over
    f_smooth in the finishing kernel, the EM bookkeeping there -- against the C oracle."""
    import torch
    B, N, T, r, iters = 3, 300, 280, 20, 2
    panel, st = _batch(B, N, T, r, 0.1)
    start = {}
    for b in range(B):
        p0, _ = ko.pca_init(np.nan_to_num(panel[b]), r)
        for k in KEYS:
            start.setdefault(k, []).append(p0[k])
    start = {k: np.stack(v) for k, v in start.items()}
    c = _ctx_with_env(**env)
    try:
        dev = torch.device("cuda", c.device)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        par = {k: t(start[k]) for k in KEYS}
        path, its, f, P = c.em_batch(t(panel), *[par[k] for k in KEYS], max_iter=iters, tol=0.0, may_have_missing=True)
        torch.cuda.synchronize()
        path = path.cpu().numpy()
        for b in range(B):
            p = {k: start[k][b].copy() for k in KEYS}
            ref = []
            for _ in range(iters):
                p, ll = co.em_step(panel[b], **p)
                ref.append(ll)
            np.testing.assert_allclose(path[b], ref, rtol=1e-8, err_msg=f"{env} loglik path, replicate {b}")
            for k in KEYS:
                got = par[k][b].cpu().numpy()
                assert np.abs(got - p[k]).max() <= 1e-8 * max(1.0, np.abs(p[k]).max()), (env, k, b, np.abs(got - p[k]).max())
    finally:
        c.close()


def test_config4_small_batch_sixteen_chunks_per_replicate():
    """The shape `secondary.c4_missing10_b32` times (N = 1000, T = 2000, r = 20, 10 % missing, 32 replicates: 16 chunks of 126 periods
    per replicate, time chunks of 250 periods for the C_t kernel): scattered replicates against the C oracle, no boundary rejected."""
    import torch
    from dynamic_factor_models_amd import DfmContext
    B, N, T, r = 32, 1000, 2000, 20
    c = DfmContext()
    try:
        panel, par = c.synth_panels(31, 0, B, T, N, r, missing_prob=0.1)
        f, P, ll = c.ks_pass_batch(panel, *par, may_have_missing=True)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(ll).all())
        assert c.chunk_fallbacks() == ((0, B) if _chunks(B, T, {}, _num_cu()) >= 1 else (0, 0))
        ix = torch.tensor([0, 13, 31], device=panel.device)
        take = lambda t: t.index_select(0, ix).cpu().numpy()
        st = dict(zip(KEYS, [take(p) for p in par]))
        _compare((take(f), take(P), take(ll)), _oracle(take(panel), st), "config 4's shape, 32 replicates")
    finally:
        c.close()
