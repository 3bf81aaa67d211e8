"""GPU parity tests of the ONE-LAUNCH balanced pass (pass_fused.hip: persistent workgroups, stream / covariance / scan
waves, b_t and the covariance tables kept on chip) and of the one-wave-per-replicate covariance recursion (dfm_cov8.h)
against the CPU oracle, whatever the library's default path is: the contexts below select them explicitly."""
import numpy as np
import pytest

from conftest import diag_only

from oracle import kalman_oracle as ko
from test_gpu_ks_pass import _batch, _compare, _ctx_with_env, _oracle, _run_dev, _slow_riccati

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fused():
    c = _ctx_with_env(DFM_PASS_FUSED=1)
    yield c
    c.close()


SHAPES = [
    (16, 200, 500, 8),        # BASELINE config 2 shape
    (5, 200, 222, 8),         # Stock-Watson window length
    (600, 40, 60, 8),         # more replicates than workgroups: the persistent loop (3 replicates per workgroup)
    (257, 24, 33, 7),         # one workgroup with two replicates, the rest with one; r padded 7 -> 8
    (3, 38, 33, 8),           # N not a multiple of the 8 series of an MFMA step
    (2, 130, 37, 5),          # two DMAs per period, r padded 5 -> 8, T = 1 mod 4
    (2, 512, 19, 6),          # four DMAs per period, 64 steps? (N / 8 = 64 > 32: falls back to the two-launch path)
    (2, 256, 40, 8),          # 32 MFMA steps, two DMAs per period
    (3, 50, 7, 8), (2, 20, 3, 8), (2, 20, 2, 6),   # panels shorter than a wave's ring
    (4, 64, 1000, 8),         # long panel: 64 KB of b_t in LDS, fewer stream waves
]


@pytest.mark.parametrize("B,N,T,r", SHAPES)
def test_fused_pass_matches_oracle(fused, B, N, T, r):
    panel, st = _batch(B, N, T, r, 0.0)
    _compare(_run_dev(fused, panel, st, may_have_missing=False), _oracle(panel, st), f"fused B={B} N={N} T={T} r={r}")


@diag_only()
@pytest.mark.parametrize("nsw", [1, 2, 3, 4, 5, 6])
def test_fused_pass_with_other_stream_wave_counts(nsw):
    c = _ctx_with_env(DFM_PASS_FUSED=1, DFM_PASS_NSW=nsw)
    try:
        for (B, N, T, r) in [(5, 200, 500, 8), (300, 30, 41, 8), (2, 130, 37, 5)]:
            panel, st = _batch(B, N, T, r, 0.0)
            _compare(_run_dev(c, panel, st, may_have_missing=False), _oracle(panel, st), f"nsw={nsw} N={N} T={T}")
    finally:
        c.close()


@pytest.mark.parametrize("N,T,r,rho,Rscale", [(20, 500, 8, 0.999, 2e2), (200, 500, 8, 0.995, 5e3), (30, 300, 6, 0.99, 50.0),
                                              (16, 120, 8, 0.9999, 1e4)])
def test_fused_pass_slow_riccati(fused, N, T, r, rho, Rscale):
    """More transient covariance steps than the 8 kept in LDS (the rest go through the global table), up to E = T."""
    panel, st = _slow_riccati(3, N, T, r, rho, Rscale)
    _compare(_run_dev(fused, panel, st, may_have_missing=False), _oracle(panel, st), f"fused slow Riccati N={N} T={T} r={r}")


def test_fused_pass_without_P_and_nan_detection(fused):
    from dynamic_factor_models_amd import DfmError
    panel, st = _batch(4, 64, 80, 8, 0.0)
    ref = _oracle(panel, st)
    f, P, ll = _run_dev(fused, panel, st, may_have_missing=False, want_P=False)
    assert P is None
    _compare((f, None, ll), ref, "fused, no P")
    bad = panel.copy(); bad[2, 17, 5] = np.nan
    with pytest.raises(DfmError) as ei:
        fused.ks_pass_batch_host(bad, st["Lam"], st["R"], st["A"], st["Q"], st["mu0"], st["P0"], may_have_missing=False)
    assert ei.value.code == -4


def test_fused_pass_full_size_equals_two_launch_pass(fused):
    """1024 replicates (config 2): the one-launch and the two-launch pass agree to rounding on every replicate."""
    import torch
    from dynamic_factor_models_amd import DfmContext
    panel, par = fused.synth_panels(5, 0, 1024, 500, 200, 8)
    f1, P1, l1 = fused.ks_pass_batch(panel, *par, may_have_missing=False)
    two = _ctx_with_env(DFM_PASS_FUSED=0)
    try:
        f2, P2, l2 = two.ks_pass_batch(panel, *par, may_have_missing=False)
        torch.cuda.synchronize()
    finally:
        two.close()
    assert torch.allclose(l1, l2, rtol=1e-11)
    assert torch.allclose(f1, f2, rtol=0, atol=1e-10 * float(f2.abs().max()))
    assert torch.allclose(P1, P2, rtol=0, atol=1e-10 * float(P2.abs().max()))


@pytest.mark.parametrize("B,N,T,r,iters", [(4, 200, 500, 8, 3), (9, 40, 80, 5, 6), (300, 30, 50, 8, 2)])
def test_fused_em_matches_oracle(fused, B, N, T, r, iters):
    import torch
    from test_gpu_em import KEYS, _dev, _start
    panel, st = _start(B, N, T, r, 0.0)
    dev = {k: _dev(fused, st[k]) for k in KEYS}
    path, its, f, P = fused.em_batch(_dev(fused, panel), *[dev[k] for k in KEYS], max_iter=iters, tol=0.0, may_have_missing=False)
    torch.cuda.synchronize()
    path = path.cpu().numpy(); f = f.cpu().numpy(); P = P.cpu().numpy()
    for b in list(range(min(B, 6))) + ([B - 1] if B > 6 else []):
        p, opath, out = ko.em(panel[b], {k: st[k][b] for k in KEYS}, max_iter=iters, tol=0.0)
        np.testing.assert_allclose(path[b], opath, rtol=1e-8, err_msg=f"loglik path b={b}")
        for k in KEYS:
            got = dev[k][b].cpu().numpy()
            assert np.abs(got - p[k]).max() <= 1e-8 * max(1.0, np.abs(p[k]).max()), (k, b)
        assert np.abs(f[b] - out["f_smooth"]).max() <= 1e-8 * np.abs(out["f_smooth"]).max()
        assert np.abs(P[b] - ko.pack_sym(out["P_smooth"])).max() <= 1e-8 * np.abs(out["P_smooth"]).max()


@diag_only()
def test_cov_wave_kernel_matches_oracle():
    """dfm_cov8.h as a drop-in for cov_kernel on the separate-launch path (gram -> cov || collapse -> pfill -> scan)."""
    c = _ctx_with_env(DFM_NO_FUSE_COV=1, DFM_COV_WAVE=1, DFM_PASS_FUSED=0)
    try:
        for (B, N, T, r) in [(16, 200, 500, 8), (7, 64, 100, 6), (3, 50, 7, 8), (2, 20, 2, 5)]:
            panel, st = _batch(B, N, T, r, 0.0)
            _compare(_run_dev(c, panel, st, may_have_missing=False), _oracle(panel, st), f"cov_wave N={N} T={T} r={r}")
        panel, st = _slow_riccati(3, 20, 500, 8, 0.999, 2e2)
        _compare(_run_dev(c, panel, st, may_have_missing=False), _oracle(panel, st), "cov_wave slow Riccati")
    finally:
        c.close()
