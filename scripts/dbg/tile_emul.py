"""Lane-level NumPy model of recursion_tile.hip (round 4): the 32 x 32 sequential Kalman recursion with the matrices held as four
16 x 16 tiles of v_mfma_f64_16x16x4 accumulators (one wave per tile), 4 x 4 block-pivot sweep inverse as rank-4 MFMA updates,
products in the "Y'X" form the tile layout gives for free, and the mean vectors riding in (padding) column 31 of the products.

This file is a DESIGN CHECK, not test infrastructure of the product: it follows the kernel instruction by instruction (same
operand layouts, same exchange buffers) and compares the result with oracle/kalman_oracle.py on a small case, so that index
mistakes are found on the CPU instead of on the GPU box.  Run: python scripts/dbg/tile_emul.py
"""
import sys, os
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import kalman_oracle as ko

R = 32
L = np.arange(64)
Q4 = L // 16      # q
C16 = L % 16      # c


def mfma(a, b, c):
    """v_mfma_f64_16x16x4: a[l] = A[l%16][l//16], b[l] = B[l//16][l%16], c/d[l][v] = D[l//16 + 4 v][l%16]."""
    A = np.zeros((16, 4)); Bm = np.zeros((4, 16))
    A[C16, Q4] = a
    Bm[Q4, C16] = b
    D = A @ Bm
    d = c.copy()
    for v in range(4):
        d[:, v] += D[Q4 + 4 * v, C16]
    return d


def to_tl(M):
    """regs[w][l][v] = M[16 I + q + 4 v][16 J + c], w = 2 I + J."""
    regs = np.zeros((4, 64, 4))
    for w in range(4):
        I, J = w >> 1, w & 1
        for v in range(4):
            regs[w, :, v] = M[16 * I + Q4 + 4 * v, 16 * J + C16]
    return regs


def from_tl(regs):
    M = np.zeros((R, R))
    for w in range(4):
        I, J = w >> 1, w & 1
        for v in range(4):
            M[16 * I + Q4 + 4 * v, 16 * J + C16] = regs[w, :, v]
    return M


def mm_tn(Y, X, nks, acc=None):
    """out = Y' X (+ acc), all in TL; wave (I, J) uses Y tiles (kb, I) as A operands and X tiles (kb, J) as B operands."""
    out = np.zeros((4, 64, 4)) if acc is None else acc.copy()
    for w in range(4):
        I, J = w >> 1, w & 1
        for ks in range(nks):
            kb, s = ks >> 2, ks & 3
            out[w] = mfma(Y[2 * kb + I][:, s], X[2 * kb + J][:, s], out[w])
    return out


def inv4(D):
    """4 x 4 SPD inverse by 2 x 2 blocks (what every lane computes redundantly); returns (Dinv, det)."""
    a00, a01, a11 = D[0, 0], D[0, 1], D[1, 1]
    detA = a00 * a11 - a01 * a01
    rA = 1.0 / detA
    i00, i01, i11 = a11 * rA, -a01 * rA, a00 * rA
    B = D[0:2, 2:4]; C = D[2:4, 2:4]
    w00 = i00 * B[0, 0] + i01 * B[1, 0]; w01 = i00 * B[0, 1] + i01 * B[1, 1]
    w10 = i01 * B[0, 0] + i11 * B[1, 0]; w11 = i01 * B[0, 1] + i11 * B[1, 1]
    s00 = C[0, 0] - (B[0, 0] * w00 + B[1, 0] * w10)
    s01 = C[0, 1] - (B[0, 0] * w01 + B[1, 0] * w11)
    s11 = C[1, 1] - (B[0, 1] * w01 + B[1, 1] * w11)
    detS = s00 * s11 - s01 * s01
    rS = 1.0 / detS
    t00, t01, t11 = s11 * rS, -s01 * rS, s00 * rS
    # X = W Si
    x00 = w00 * t00 + w01 * t01; x01 = w00 * t01 + w01 * t11
    x10 = w10 * t00 + w11 * t01; x11 = w10 * t01 + w11 * t11
    E = np.zeros((4, 4))
    E[0, 0] = i00 + x00 * w00 + x01 * w01
    E[0, 1] = E[1, 0] = i01 + x00 * w10 + x01 * w11
    E[1, 1] = i11 + x10 * w10 + x11 * w11
    E[0, 2] = E[2, 0] = -x00; E[0, 3] = E[3, 0] = -x01
    E[1, 2] = E[2, 1] = -x10; E[1, 3] = E[3, 1] = -x11
    E[2, 2] = t00; E[2, 3] = E[3, 2] = t01; E[3, 3] = t11
    return E, detA * detS


def sweep_inverse(m, npiv):
    """In place on TL regs m[w][l][v]: the leading 4 npiv x 4 npiv block is inverted (the rest -- identity padding -- is left
    alone); returns det.  One LDS exchange (raw pivot rows with the pivot block replaced by -I, and the raw block) per pivot."""
    det = 1.0
    m = m.copy()
    for p in range(npiv):
        Ik, vk, ck = p >> 2, p & 3, 4 * (p & 3)
        praw = np.zeros((4, 32)); pD = np.zeros((4, 4))
        for w in range(4):                                   # a. waves of the pivot row block publish their rows
            I, J = w >> 1, w & 1
            if I != Ik:
                continue
            val = m[w, :, vk].copy()
            if J == Ik:
                inblk = (C16 >= ck) & (C16 < ck + 4)
                pD[Q4[inblk], C16[inblk] - ck] = val[inblk]
                val = np.where(inblk, np.where(C16 - ck == Q4, -1.0, 0.0), val)
            praw[Q4, 16 * J + C16] = val
        Dinv, dd = inv4(pD)                                   # (first version: explicit D^-1 by 2 x 2 blocks)
        det *= dd
        # the kernel: D = L diag(e) L' and two substitutions per column (uses the UPPER triangle of the published block)
        D00, D10, D20, D30, D11, D21, D31, D22, D32, D33 = pD[0, 0], pD[0, 1], pD[0, 2], pD[0, 3], pD[1, 1], pD[1, 2], pD[1, 3], pD[2, 2], pD[2, 3], pD[3, 3]
        i0 = 1 / D00; l10, l20, l30 = D10 * i0, D20 * i0, D30 * i0
        e1 = D11 - l10 * D10; i1 = 1 / e1
        u21 = D21 - l20 * D10; u31 = D31 - l30 * D10; l21, l31 = u21 * i1, u31 * i1
        e2 = D22 - l20 * D20 - l21 * u21; i2 = 1 / e2
        u32 = D32 - l30 * D20 - l31 * u21; l32 = u32 * i2
        e3 = D33 - l30 * D30 - l31 * u31 - l32 * u32; i3 = 1 / e3
        assert np.isclose(D00 * e1 * e2 * e3, dd, rtol=1e-12)

        def solve(pc):
            y1 = pc[1] - l10 * pc[0]
            y2 = pc[2] - l20 * pc[0] - l21 * y1
            y3 = pc[3] - l30 * pc[0] - l31 * y1 - l32 * y2
            t3 = y3 * i3
            t2 = y2 * i2 - l32 * t3
            t1 = y1 * i1 - l21 * t2 - l31 * t3
            t0 = pc[0] * i0 - l10 * t1 - l20 * t2 - l30 * t3
            return np.array([t0, t1, t2, t3])
        for w in range(4):
            I, J = w >> 1, w & 1
            col = 16 * J + C16
            T = solve(praw[:, col])                           # all four rows for the lane's column ...
            assert np.allclose(T, Dinv @ praw[:, col], rtol=1e-9, atol=1e-12)
            bop = -T[Q4, L]                                   # ... the lane keeps row q:  B operand = -T~[q][16 J + c]
            aop = praw[Q4, 16 * I + C16]                      # A operand = R~[q][16 I + c]
            acc = m[w].copy()
            if I == Ik:
                acc[:, vk] = 0.0
            if J == Ik:
                inblk = (C16 >= ck) & (C16 < ck + 4)
                acc[inblk, :] = 0.0
            m[w] = mfma(aop, bop, acc)
    lim = 4 * npiv
    for w in range(4):
        I, J = w >> 1, w & 1
        for v in range(4):
            row = 16 * I + Q4 + 4 * v; col = 16 * J + C16
            m[w, :, v] = np.where((row < lim) & (col < lim), -m[w, :, v], m[w, :, v])
    return m, det


def pad(M, r, eye=True):
    out = np.eye(R) if eye else np.zeros((R, R))
    out[:r, :r] = M
    return out


def set_col31(tl, vec32):
    """column 31 of a TL matrix := vec (lanes c == 15 of the waves J == 1)."""
    tl = tl.copy()
    for w in (1, 3):
        I = w >> 1
        for v in range(4):
            sel = C16 == 15
            tl[w, sel, v] = vec32[16 * I + Q4[sel] + 4 * v]
    return tl


def get_col31(tl):
    vec = np.zeros(R)
    for w in (1, 3):
        I = w >> 1
        for v in range(4):
            sel = C16 == 15
            vec[16 * I + Q4[sel] + 4 * v] = tl[w, sel, v]
    return vec


def tile_pass(x, Lam, Rv, A, Q, mu0, P0):
    """The kernel's algorithm end to end (collapse done in NumPy): returns f_smooth, P_smooth, loglik, EM sums."""
    T, N = x.shape
    r = Lam.shape[1]
    assert 16 < r <= 31
    npiv = (r + 3) // 4
    nks = npiv
    obs = ~np.isnan(x)
    x0 = np.where(obs, x, 0.0)
    W = Lam / Rv[:, None]
    # collapsed inputs (what the collapse kernels hand over)
    bcol = np.zeros((T, R)); bcol[:, :r] = x0 @ W
    scol = (x0 * x0 / Rv).sum(1)
    nobs = obs.sum(1)
    ldrow = np.array([np.log(Rv[obs[t]]).sum() for t in range(T)])
    Ct = np.zeros((T, R, R))
    for t in range(T):
        Ct[t, :r, :r] = (Lam[obs[t]].T * (1.0 / Rv[obs[t]])) @ Lam[obs[t]]
    Ap = pad(A, r, eye=False); Qp = pad(Q, r); P0p = pad(P0, r)
    mu = np.zeros(R); mu[:r] = mu0

    # ---- prologue
    Qi, detQ = sweep_inverse(to_tl(Qp), npiv)
    Omf, detP0 = sweep_inverse(to_tl(P0p), npiv)
    Atl = to_tl(Ap)
    Kt = mm_tn(Atl, Qi, nks)                 # K' = A' Qi
    Ktl = mm_tn(Qi, Atl, nks)                # K  = Qi A
    Phi = mm_tn(Ktl, Atl, nks)               # Phi = K' A = A' Qi A
    assert np.allclose(from_tl(Kt), (np.linalg.inv(Qp) @ Ap).T)
    assert np.allclose(from_tl(Phi), Ap.T @ np.linalg.inv(Qp) @ Ap)
    Xk = set_col31(Kt, mu)
    xi = get_col31(mm_tn(Omf, Xk, 8))        # xi_0 = P0^-1 mu0  (all 8 k-steps: mu0's padding is zero anyway)
    q0 = mu @ xi
    sum_xw = 0.0; ssum = 0.0; nsum = 0.0; ldsum = 0.0; logdet = 0.0
    Ztab = []; Jttab = []; wtab = []
    eye31 = np.zeros(R); eye31[31] = 1.0
    for t in range(T):
        M = Omf + Phi
        Z, dM = sweep_inverse(M, npiv)
        logdet += np.log(dM)
        Xk = set_col31(Kt, xi)
        Jaug = mm_tn(Z, Xk, nks)             # [J | w] = Z' [K' | xi]
        Jt = mm_tn(Kt, Z, nks)               # J' = K Z
        prod = mm_tn(Kt, Jaug, nks)          # K [J | w]
        w = get_col31(Jaug); kw = get_col31(prod)
        sum_xw += xi @ w
        Omp = Qi - prod
        Omp = set_col31(Omp, eye31)          # column 31 is padding: restore it
        Ztab.append(Z); Jttab.append(Jt); wtab.append(w)
        xi = kw + bcol[t]
        Omf = Omp + to_tl(Ct[t])
        ssum += scol[t]; nsum += nobs[t]; ldsum += ldrow[t]
    # ---- terminal
    Ps, detOmT = sweep_inverse(Omf, npiv)
    fT = get_col31(mm_tn(Ps, set_col31(Kt, xi), 8))
    qd = q0 - xi @ fT - sum_xw
    LD = np.log(detOmT) + np.log(detP0) + T * np.log(detQ) + logdet
    ll = -0.5 * (nsum * np.log(2 * np.pi) + ldsum + LD + ssum + qd)
    # ---- backward
    f = np.zeros((T, R)); P = np.zeros((T, R, R))
    f[T - 1] = fT; P[T - 1] = from_tl(Ps)
    fs = fT
    SP = Ps.copy(); SU = np.zeros_like(Ps)
    for t in range(T - 1, -1, -1):
        U = mm_tn(Ps, Jttab[t], nks)                         # U = P_s J'
        Uaug = set_col31(U, fs)
        Pn = mm_tn(Jttab[t], Uaug, nks, acc=Ztab[t])         # Z + J [U | f+]
        fs = wtab[t] + get_col31(Pn)
        Ps = set_col31(Pn, eye31)
        SU += U
        if t > 0:
            f[t - 1] = fs; P[t - 1] = from_tl(Ps); SP += Ps
    return f[:, :r], P[:, :r, :r], ll, dict(f0=fs[:r], P0s=from_tl(Ps)[:r, :r], SP=from_tl(SP)[:r, :r], SU=from_tl(SU)[:r, :r])


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    # 1. the sweep inverse alone
    for r in (17, 20, 24, 31):
        Mx = rng.standard_normal((r, r + 5)); S = Mx @ Mx.T + r * np.eye(r)
        Z, det = sweep_inverse(to_tl(pad(S, r)), (r + 3) // 4)
        Zn = from_tl(Z)
        assert np.allclose(Zn[:r, :r], np.linalg.inv(S), rtol=1e-10, atol=1e-12), r
        assert np.allclose(Zn[r:, r:], np.eye(R - r)) and np.allclose(Zn[:r, r:], 0)
        assert np.isclose(det, np.linalg.det(S), rtol=1e-10)
    # 2. Y'X
    Y = rng.standard_normal((R, R)); X = rng.standard_normal((R, R))
    assert np.allclose(from_tl(mm_tn(to_tl(Y), to_tl(X), 8)), Y.T @ X)
    # 3. the whole pass against the oracle
    for (N, T, r, miss) in ((40, 9, 20, 0.2), (25, 6, 17, 0.0), (50, 7, 31, 0.3)):
        x, p = ko.synth_replicate(1, N, T, r, missing=miss)
        out = ko.kfs_pass(x, p["Lam"], p["R"], p["A"], p["Q"], p["mu0"], p["P0"], lag_one=True)
        f, P, ll, em = tile_pass(x, p["Lam"], p["R"], p["A"], p["Q"], p["mu0"], p["P0"])
        assert np.isclose(ll, out["loglik"], rtol=1e-10), (ll, out["loglik"])
        assert np.allclose(f, out["f_smooth"], rtol=1e-8, atol=1e-10)
        assert np.allclose(P, out["P_smooth"], rtol=1e-8, atol=1e-10)
        assert np.allclose(em["f0"], out["f0_smooth"], atol=1e-10) and np.allclose(em["P0s"], out["P0_smooth"], atol=1e-10)
        assert np.allclose(em["SU"], out["P_lag"].sum(0), atol=1e-9)
        assert np.allclose(em["SP"], out["P_smooth"].sum(0), atol=1e-9)
        print("ok", N, T, r, miss, ll)
