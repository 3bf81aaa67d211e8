"""GPU tests of the PCA initialisation (vs oracle pca_init = the reference's pca_score + OLS start) and of
the device-side synthetic-panel generator (determinism, standardisation, DGP moments)."""
import numpy as np
import pytest

from oracle import kalman_oracle as ko

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available()
    from dynamic_factor_models_amd import DfmContext
    c = DfmContext()
    yield c
    c.close()


def _dev(ctx, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.device("cuda", ctx.device))


@pytest.mark.parametrize("B,N,T,r", [(6, 200, 500, 8), (4, 40, 80, 3), (3, 139, 222, 4), (2, 64, 50, 12),
                                     (2, 30, 41, 1), (2, 300, 60, 5),
                                     # LDS-resident iteration (N <= 256, Rp <= 8) at its edges, matrix-pipe X'X with partial tiles
                                     (3, 256, 100, 8), (2, 255, 33, 7), (3, 64, 50, 1), (2, 17, 30, 2), (2, 257, 40, 8),
                                     (2, 33, 12, 2)])
def test_pca_init_matches_oracle(ctx, B, N, T, r):
    import torch
    panels = np.stack([ko.synth_replicate(b, N, T, r)[0] for b in range(B)])
    Lam, R, A, Q, mu0, P0, F = ctx.pca_init_batch(_dev(ctx, panels), r)
    torch.cuda.synchronize()
    got = dict(Lam=Lam, R=R, A=A, Q=Q, mu0=mu0, P0=P0)
    for b in range(B):
        ref, Fo = ko.pca_init(panels[b], r)
        np.testing.assert_allclose(F[b].cpu().numpy(), Fo, rtol=0, atol=1e-8 * np.abs(Fo).max(), err_msg="scores")
        for k in ("Lam", "R", "A", "Q", "mu0", "P0"):
            g = got[k][b].cpu().numpy()
            np.testing.assert_allclose(g, ref[k], rtol=0, atol=1e-8 * max(np.abs(ref[k]).max(), 1e-300), err_msg=k)


def test_pca_host_entry_and_nan_refused(ctx):
    from dynamic_factor_models_amd import DfmError
    x = np.stack([ko.synth_replicate(b, 30, 40, 2)[0] for b in range(2)])
    p, F = ctx.pca_init_batch_host(x, 2)
    ref, Fo = ko.pca_init(x[1], 2)
    np.testing.assert_allclose(F[1], Fo, atol=1e-9 * np.abs(Fo).max())
    np.testing.assert_allclose(p["Lam"][1], ref["Lam"], atol=1e-9)
    x[0, 3, 4] = np.nan
    with pytest.raises(DfmError):
        ctx.pca_init_batch_host(x, 2)


def test_synth_panels_deterministic_standardised_and_consistent(ctx):
    import torch
    B, T, N, r = 8, 400, 60, 4
    p1, par1 = ctx.synth_panels(1234, 0, B, T, N, r)
    p2, par2 = ctx.synth_panels(1234, 0, B, T, N, r)
    p3, _ = ctx.synth_panels(1234, 4, 4, T, N, r)           # replicates 4..7 again: same numbers
    torch.cuda.synchronize()
    x = p1.cpu().numpy()
    assert np.array_equal(x, p2.cpu().numpy())
    assert np.array_equal(x[4:], p3.cpu().numpy())
    assert not np.array_equal(x[0], x[1])
    np.testing.assert_allclose(x.mean(axis=1), 0.0, atol=1e-12)
    np.testing.assert_allclose(x.std(axis=1), 1.0, atol=1e-12)
    Lam, R, A, Q, mu0, P0 = [t.cpu().numpy() for t in par1]
    a = np.linspace(0.5, 0.9, r)
    np.testing.assert_allclose(A[3], np.diag(a)); np.testing.assert_allclose(Q[3], np.diag(1 - a * a))
    # model-implied variance of every standardised series: lam' lam + R (unit-variance factors) ~ 1
    implied = (Lam ** 2).sum(-1) + R
    assert np.all(np.abs(implied - 1.0) < 0.7) and abs(implied.mean() - 1.0) < 0.08   # sampling noise of persistent factors
    # the DGP parameters give a far better likelihood than a wrong model on these panels
    f, P, ll = ctx.ks_pass_batch(p1, *par1, may_have_missing=False)
    f2, P2, ll2 = ctx.ks_pass_batch(p1, par1[0] * 0.3, *par1[1:], may_have_missing=False)
    torch.cuda.synchronize()
    assert bool((ll > ll2).all())


def test_synth_missing_cells(ctx):
    import torch
    p, _ = ctx.synth_panels(7, 0, 4, 200, 50, 3, missing_prob=0.1)
    torch.cuda.synchronize()
    frac = float(torch.isnan(p).double().mean().item())
    assert 0.08 < frac < 0.12


@pytest.mark.parametrize("B,N,T,r,missing,first", [(5, 60, 81, 4, 0.0, 0), (3, 200, 500, 8, 0.0, 8189), (4, 31, 40, 3, 0.15, 2 ** 33),
                                                  (2, 20, 30, 1, 0.05, 7)])
def test_device_generator_equals_its_host_restatement(ctx, B, N, T, r, missing, first):
    """The panels the throughput runs use (dfm_synth_panels_dev: Philox4x32-10 keyed by (seed, global replicate index))
    cell by cell against oracle/synth_oracle.py -- the §8(d) DGP on the Random123-pinned generator.  Differences are the
    last bits of log / sincospi and of the summation order in the standardisation."""
    import torch
    from oracle import synth_oracle as so
    seed = 20160415
    panel, par = ctx.synth_panels(seed, first, B, T, N, r, missing_prob=missing)
    torch.cuda.synchronize()
    x = panel.cpu().numpy()
    got = dict(zip(("Lam", "R", "A", "Q", "mu0", "P0"), [p.cpu().numpy() for p in par]))
    for b in range(B):
        xo, po = so.synth_replicate_device(seed, first + b, N, T, r, missing)
        np.testing.assert_array_equal(np.isnan(x[b]), np.isnan(xo))          # the same cells are missing
        np.testing.assert_allclose(np.nan_to_num(x[b]), np.nan_to_num(xo), rtol=0, atol=2e-11)
        for k in po:
            np.testing.assert_allclose(got[k][b], po[k], rtol=1e-11, atol=1e-12, err_msg=k)
