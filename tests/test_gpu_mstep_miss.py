"""GPU parity of the loadings M-step with missing cells on the matrix pipe (mstep_miss.hip) against the CPU oracle's EM, one
case per class of its geometry: tile slots per wave (4 / 7 / 10 / 13 / 16 / 18 / 21), one and two column groups, stages of
32 / 16 / 8 periods, partial series blocks, partial last stage, XCD-ordered and flat item order, a replicate that has stopped
iterating.  DFM_MSTEP_MISS=2 sends EVERY shape the kernel supports through it (the default keeps mstep_lam_kernel at Rp = 8,
N <= 256); the switch is read when a handle is created, so each mode gets its own handle."""
import os

import numpy as np
import pytest

from conftest import diag_only
from oracle import kalman_oracle as ko

pytestmark = pytest.mark.gpu
RTOL = 1e-8
KEYS = ("Lam", "R", "A", "Q", "mu0", "P0")


def _ctx(mode, kp=None, lst=None):
    import torch
    assert torch.cuda.is_available()
    from dynamic_factor_models_amd import DfmContext
    old = {k: os.environ.get(k) for k in ("DFM_MSTEP_MISS", "DFM_MM_KP", "DFM_MSTEP_LIST")}
    os.environ["DFM_MSTEP_MISS"] = str(mode)
    if kp:
        os.environ["DFM_MM_KP"] = str(kp)
    if lst is not None:
        os.environ["DFM_MSTEP_LIST"] = str(lst)
    try:
        return DfmContext()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _start(B, N, T, r, missing):
    panels, starts = [], []
    for b in range(B):
        x, _ = ko.synth_replicate(100 + b, N, T, r, missing=missing)
        p0, _ = ko.pca_init(np.nan_to_num(x), r)
        panels.append(x); starts.append(p0)
    return np.stack(panels), {k: np.stack([s[k] for s in starts]) for k in starts[0]}


def _dev(ctx, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.device("cuda", ctx.device))


@pytest.mark.parametrize("B,N,T,r,missing", [
    (3, 18, 37, 1, 0.2),        # Rp = 2 is outside the kernel: falls back to mstep_lam_kernel (must still be right)
    (2, 30, 50, 5, 0.15),       # Rp = 8, 2 + 1 tiles, one partial series block, stages of 32 periods (the last of 18)
    (17, 130, 65, 8, 0.1),      # Rp = 8, 4 tile slots, two series blocks, XCD-ordered items with B not a multiple of 8
    (2, 40, 40, 12, 0.1),       # Rp = 16: 5 + 1 tiles in 7 slots
    (2, 48, 57, 16, 0.1),       # Rp = 16: 9 + 1 tiles = 10 slots
    (2, 34, 61, 17, 0.1),       # Rp = 32: 10 + 2 tiles in 13 slots, stages of 16 periods (T > 3 r: the start's Q must be regular)
    (2, 300, 60, 20, 0.1),      # config 4's class: 14 + 2 tiles = 16 slots, three series blocks (the last of 44)
    (2, 50, 83, 24, 0.1),       # 19 + 2 = 21 slots: stages of 8 periods in three buffers
    (2, 70, 92, 27, 0.1),       # two column groups (24 + 2 tiles, 13 per wave), 64 series per item
    (2, 66, 107, 32, 0.05),     # 33 + 2 tiles, 18 slots per wave, every factor column in use
])
@pytest.mark.parametrize("lst", [None, 0])     # the default route (the list form where it exists and few cells are missing) | the dense product
def test_loadings_step_with_missing_cells_matches_the_oracle(B, N, T, r, missing, lst):
    import torch
    ctx = _ctx(2, lst=lst)
    try:
        panel, st = _start(B, N, T, r, missing)
        dev = {k: _dev(ctx, st[k]) for k in KEYS}
        iters = 2
        path, its, f, P = ctx.em_batch(_dev(ctx, panel), *[dev[k] for k in KEYS], max_iter=iters, tol=0.0)
        torch.cuda.synchronize()
        path = path.cpu().numpy()
        for b in range(B):
            p, opath, _ = ko.em(panel[b], {k: st[k][b] for k in KEYS}, max_iter=iters, tol=0.0)
            np.testing.assert_allclose(path[b], opath, rtol=RTOL, err_msg=f"loglik path b={b}")
            for k in KEYS:
                got = dev[k][b].cpu().numpy()
                assert np.abs(got - p[k]).max() <= RTOL * max(1.0, np.abs(p[k]).max()), (k, b, np.abs(got - p[k]).max())
    finally:
        ctx.close()


@pytest.mark.parametrize("lst", [0, 2])
@pytest.mark.parametrize("B,N,T,r,missing", [
    (2, 30, 50, 5, 0.15),       # Rp = 8: one 1-KB read per row of V, one partial series block, last stage of 2 periods
    (17, 130, 65, 8, 0.1),      # two series blocks (the second of 2 series), XCD-ordered items
    (2, 48, 57, 16, 0.1),       # Rp = 16, r = 16: two reads per row, one tile of f
    (2, 300, 60, 20, 0.1),      # config 4's class: two reads per row, two tiles of f, three series blocks (the last of 44)
    (2, 140, 45, 22, 0.02),     # the widest row of the list form (253 columns, stages in two buffers), hardly a missing cell
    (2, 300, 120, 20, 0.6),     # most cells missing: walks of ten rows per series and stage
])
def test_list_form_and_dense_product_both_match_the_oracle(B, N, T, r, missing, lst):
    """DFM_MSTEP_LIST = 0: the dense product (mstep_miss_kernel) everywhere; 2: the walk over the missing cells
    (mstep_miss_list_kernel) wherever its geometry exists -- the default decides between them on the device from a sampled share
    of missing cells, so each is pinned here on its own."""
    import torch
    ctx = _ctx(2, lst=lst)
    try:
        panel, st = _start(B, N, T, r, missing)
        dev = {k: _dev(ctx, st[k]) for k in KEYS}
        path, its, f, P = ctx.em_batch(_dev(ctx, panel), *[dev[k] for k in KEYS], max_iter=2, tol=0.0)
        torch.cuda.synchronize()
        path = path.cpu().numpy()
        for b in range(B):
            p, opath, _ = ko.em(panel[b], {k: st[k][b] for k in KEYS}, max_iter=2, tol=0.0)
            np.testing.assert_allclose(path[b], opath, rtol=RTOL, err_msg=f"loglik path b={b}")
            for k in KEYS:
                got = dev[k][b].cpu().numpy()
                assert np.abs(got - p[k]).max() <= RTOL * max(1.0, np.abs(p[k]).max()), (k, b, np.abs(got - p[k]).max())
    finally:
        ctx.close()


def test_route_follows_the_share_of_missing_cells():
    """The default route: 10 % missing -> the list form, 60 % -> the dense product (the kernel that is not chosen exits at once, so
    the library's kernel log shows both launches; the RESULT must be the oracle's either way -- checked above -- and the two
    routes' results differ by rounding only)."""
    import torch
    B, N, T, r = 2, 300, 120, 20
    for missing in (0.1, 0.6):
        panel, st = _start(B, N, T, r, missing)
        outs = []
        for lst in (None, 0, 2):
            ctx = _ctx(2, lst=lst)
            try:
                dev = {kk: _dev(ctx, st[kk]) for kk in KEYS}
                ctx.em_batch(_dev(ctx, panel), *[dev[kk] for kk in KEYS], max_iter=1, tol=0.0)
                torch.cuda.synchronize()
                outs.append({kk: dev[kk].cpu().numpy() for kk in ("Lam", "R")})
            finally:
                ctx.close()
        want = 2 if missing < 0.4 else 1                  # index into outs of the route the default must have taken
        other = 3 - want
        for kk in ("Lam", "R"):
            assert np.array_equal(outs[0][kk], outs[want][kk]), (missing, kk)       # bit-identical to the chosen route
            assert np.abs(outs[0][kk] - outs[other][kk]).max() <= 1e-11 * max(1.0, np.abs(outs[0][kk]).max()), (missing, kk)


@diag_only()
def test_lds_column_solve_agrees_with_the_row_per_lane_solve():
    """Diagnostics build only (DFM_MM_FINISH=1: mmw_finish_kernel): the same loadings to rounding as mmw_solve_kernel."""
    import torch
    B, N, T, r = 3, 140, 75, 20
    panel, st = _start(B, N, T, r, 0.12)
    outs = []
    for fin in (None, "1"):
        if fin:
            os.environ["DFM_MM_FINISH"] = fin
        try:
            ctx = _ctx(2)
            dev = {kk: _dev(ctx, st[kk]) for kk in KEYS}
            ctx.em_batch(_dev(ctx, panel), *[dev[kk] for kk in KEYS], max_iter=1, tol=0.0)
            torch.cuda.synchronize()
            outs.append({kk: dev[kk].cpu().numpy() for kk in ("Lam", "R")})
            ctx.close()
        finally:
            os.environ.pop("DFM_MM_FINISH", None)
    for kk in ("Lam", "R"):
        assert np.abs(outs[0][kk] - outs[1][kk]).max() <= 1e-11 * max(1.0, np.abs(outs[0][kk]).max()), kk


@diag_only()
@pytest.mark.parametrize("kp", [8, 16, 32])
def test_stage_depths_agree(kp):
    """The same EM step through every stage depth that fits (8 periods x 3 buffers, 16 x 2, 32 x 2): identical to rounding."""
    import torch
    B, N, T, r = 3, 140, 75, 8
    panel, st = _start(B, N, T, r, 0.12)
    outs = []
    for k in (None, kp):
        ctx = _ctx(2, k, lst=0)
        try:
            dev = {kk: _dev(ctx, st[kk]) for kk in KEYS}
            ctx.em_batch(_dev(ctx, panel), *[dev[kk] for kk in KEYS], max_iter=1, tol=0.0)
            torch.cuda.synchronize()
            outs.append({kk: dev[kk].cpu().numpy() for kk in ("Lam", "R")})
        finally:
            ctx.close()
    for kk in ("Lam", "R"):
        assert np.abs(outs[0][kk] - outs[1][kk]).max() <= 1e-12 * max(1.0, np.abs(outs[0][kk]).max()), kk


def test_a_stopped_replicate_keeps_its_loadings():
    """tol > 0: a replicate that has converged is inactive in later iterations -- its items are skipped, its parameters stay."""
    import torch
    ctx = _ctx(2)
    try:
        B, N, T, r = 4, 60, 50, 12
        panel, st = _start(B, N, T, r, 0.1)
        dev = {k: _dev(ctx, st[k]) for k in KEYS}
        path, its, _, _ = ctx.em_batch(_dev(ctx, panel), *[dev[k] for k in KEYS], max_iter=12, tol=1e-4)
        torch.cuda.synchronize()
        its = its.cpu().numpy()
        for b in range(B):
            p, opath, _ = ko.em(panel[b], {k: st[k][b] for k in KEYS}, max_iter=12, tol=1e-4)
            assert its[b] == len(opath), (b, its[b], len(opath))
            for k in ("Lam", "R"):
                got = dev[k][b].cpu().numpy()
                assert np.abs(got - p[k]).max() <= 1e-7 * max(1.0, np.abs(p[k]).max()), (k, b)
    finally:
        ctx.close()


@pytest.mark.parametrize("B,N,T,r,missing", [
    (2, 260, 67, 17, 0.1),      # Rp = 32 beyond the register tiling of collapse_kernel: collapse_wide2 (missing-cell mode) + ct_miss_wide2
    (3, 300, 110, 32, 0.3),     # every factor column in use, heavy missingness, T not a multiple of 16
    (17, 270, 70, 20, 0.05),    # XCD-ordered queues with B not a multiple of 8
])
def test_wide_pass_with_missing_cells_matches_the_oracle(B, N, T, r, missing):
    """The pass (not the EM) of the wide sequential path: C_t of the periods with missing cells from ct_miss_wide2_kernel."""
    import torch
    ctx = _ctx(1)
    try:
        panel, st = _start(B, N, T, r, missing)
        f, P, ll = ctx.ks_pass_batch(_dev(ctx, panel), *[_dev(ctx, st[k]) for k in KEYS], may_have_missing=True)
        torch.cuda.synchronize()
        f = f.cpu().numpy(); P = P.cpu().numpy(); ll = ll.cpu().numpy()
        for b in range(min(B, 4)):
            out = ko.kfs_pass(panel[b], *[st[k][b] for k in KEYS], lag_one=False)
            assert abs(ll[b] - out["loglik"]) <= 1e-9 * abs(out["loglik"]), (b, ll[b], out["loglik"])
            assert np.abs(f[b] - out["f_smooth"]).max() <= 1e-9 * np.abs(out["f_smooth"]).max()
            assert np.abs(P[b] - ko.pack_sym(out["P_smooth"])).max() <= 1e-9 * np.abs(out["P_smooth"]).max()
    finally:
        ctx.close()
