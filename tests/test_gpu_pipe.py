"""Large batches of panels with missing cells (states 8 wide; from 16 replicates per CU on) run as sub-batches of 8 replicates per CU
on two streams with two workspace slots (capi.hip pipe_run): the results must be those of the sub-batches run one call at a time, whatever the batch does not
divide into (to rounding: which sequential kernel takes the replicates that fail the chunk boundary check depends on the size of the
launch -- the wave pair up to one replicate per SIMD, one wave beyond), and equal to the oracle's."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import kalman_oracle as ko

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


def _synth(ctx, B, N, T, r, missing, seed=11):
    panel, par = ctx.synth_panels(seed, 0, B, T, N, r, missing_prob=missing)
    return panel, par


@pytest.mark.parametrize("B,N,T,r,missing", [(4200, 40, 64, 8, 0.1), (4096, 36, 50, 6, 0.2), (6145, 24, 40, 8, 0.05)])
def test_piped_pass_equals_its_sub_batches(ctx, B, N, T, r, missing):
    import torch
    panel, par = _synth(ctx, B, N, T, r, missing)
    f, P, ll = ctx.ks_pass_batch(panel, *par, may_have_missing=True)
    torch.cuda.synchronize()
    nf, nt = ctx.chunk_fallbacks()
    assert nt == B, (nf, nt)
    Bs = 2000                                                     # (below two sub-batches: one batch, one stream)
    for b0 in range(0, B, Bs):
        sl = slice(b0, min(B, b0 + Bs))
        f1, P1, ll1 = ctx.ks_pass_batch(panel[sl].contiguous(), *[p[sl].contiguous() for p in par], may_have_missing=True)
        torch.cuda.synchronize()
        for a, c, what in ((ll[sl], ll1, "loglik"), (f[sl], f1, "f_smooth"), (P[sl], P1, "P_smooth")):
            assert (a - c).abs().max().item() <= 1e-11 * c.abs().max().item(), f"{what} of the sub-batch at {b0}"
    for b in (0, 2047, 2048, B - 1):
        st = [p[b:b + 1].cpu().numpy() for p in par]
        fo, Po, llo = co.ks_pass_batch(panel[b:b + 1].cpu().numpy(), *st)
        np.testing.assert_allclose(ll[b:b + 1].cpu().numpy(), llo, rtol=1e-9)
        assert np.abs(f[b:b + 1].cpu().numpy() - fo).max() <= 1e-9 * np.abs(fo).max()
        assert np.abs(P[b:b + 1].cpu().numpy() - Po).max() <= 1e-9 * np.abs(Po).max()


@pytest.mark.parametrize("tol", [0.0, 1e-4])
def test_piped_em_equals_its_sub_batches(ctx, tol):
    import torch
    B, N, T, r, iters = 4200, 40, 64, 8, 4
    panel, par = _synth(ctx, B, N, T, r, 0.1, seed=5)
    start = [p.clone() for p in par]
    path, its, f, P = ctx.em_batch(panel, *par, max_iter=iters, tol=tol)
    torch.cuda.synchronize()
    Bs = 2000
    for b0 in range(0, B, Bs):
        sl = slice(b0, min(B, b0 + Bs))
        sub = [p[sl].clone() for p in start]
        path1, its1, f1, P1 = ctx.em_batch(panel[sl].contiguous(), *sub, max_iter=iters, tol=tol)
        torch.cuda.synchronize()
        assert torch.equal(its[sl], its1), f"iteration counts at {b0}"
        pa, pb = torch.nan_to_num(path[sl]), torch.nan_to_num(path1)
        assert (pa - pb).abs().max().item() <= 1e-9 * pb.abs().max().item(), f"loglik path at {b0}"
        for a, c in zip(par, sub):
            assert (a[sl] - c).abs().max().item() <= 1e-8 * max(1.0, c.abs().max().item()), f"parameters at {b0}"
        assert (f[sl] - f1).abs().max().item() <= 1e-8 * f1.abs().max().item()
    b = 2500
    p0 = {k: start[i][b].cpu().numpy() for i, k in enumerate(KEYS)}
    po, opath, out = ko.em(panel[b].cpu().numpy(), p0, max_iter=iters, tol=tol)
    got = path[b].cpu().numpy()
    np.testing.assert_allclose(got[:len(opath)], opath, rtol=1e-8)


def _fresh_ctx(env):
    import os
    from dynamic_factor_models_amd import DfmContext
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return DfmContext()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_route_switches_of_this_file_agree():
    """Diagnostics build only: DFM_PIPE=0 (one batch, one stream), DFM_NARROW_TAB=0 / DFM_ODD_PAD8=0 (r <= 4 and odd N through
    collapse_kernel + chunk_bridge_kernel) give the default routes' results to rounding."""
    import os
    import torch
    if os.environ.get("DFM_LIB") != "diag":
        pytest.skip("switches of the diagnostics build: run with DFM_LIB=diag")
    base = _fresh_ctx({})
    try:
        panel, par = base.synth_panels(3, 0, 4200, 40, 36, 8, missing_prob=0.1)
        pn, parn = base.synth_panels(4, 0, 6, 120, 77, 4, missing_prob=0.1)
        ref = base.ks_pass_batch(panel, *par, may_have_missing=True)
        refn = base.ks_pass_batch(pn, *parn, may_have_missing=True)
        torch.cuda.synchronize()
    finally:
        pass
    for env, (x, p, want) in (({"DFM_PIPE": "0"}, (panel, par, ref)), ({"DFM_NARROW_TAB": "0", "DFM_ODD_PAD8": "0"}, (pn, parn, refn))):
        c = _fresh_ctx(env)
        try:
            got = c.ks_pass_batch(x, *p, may_have_missing=True)
            torch.cuda.synchronize()
            for a, b_ in zip(got, want):
                assert (a - b_).abs().max().item() <= 1e-10 * b_.abs().max().item(), env
        finally:
            c.close()
    base.close()


def test_piped_em_iterate_equals_em_batch(ctx):
    """dfm_em_iterate_batch_dev (the unit the multi-GPU drivers step; the caller owns path / iters / active) over a sub-batched batch:
    the per-replicate state a handle keeps between the calls (the chunk kernel's count of consecutive boundary failures) lives in a
    workspace slot that other sub-batches overwrite -- it only steers which kernel computes a replicate, never what comes out."""
    import torch
    B, N, T, r, iters = 4200, 40, 64, 8, 4
    panel, par = _synth(ctx, B, N, T, r, 0.1, seed=9)
    a = [p.clone() for p in par]
    b = [p.clone() for p in par]
    path, its, f, P = ctx.em_batch(panel, *a, max_iter=iters, tol=0.0, may_have_missing=True)
    dev = panel.device
    path2 = torch.empty((B, iters), dtype=torch.float64, device=dev)
    its2 = torch.empty((B,), dtype=torch.int32, device=dev)
    act2 = torch.empty((B,), dtype=torch.int32, device=dev)
    for k in range(iters):
        ctx.em_iterate_batch(panel, *b, k, iters, 0.0, path2, its2, act2, may_have_missing=True)
    torch.cuda.synchronize()
    assert torch.equal(its, its2)
    assert (path - path2).abs().max().item() <= 1e-9 * path.abs().max().item()
    for x, y in zip(a, b):
        assert (x - y).abs().max().item() <= 1e-8 * max(1.0, x.abs().max().item())
