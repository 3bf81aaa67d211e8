"""NumPy model of the time chunks of recursion_tile_kernel<true> (csrc/recursion_tile.hip) at the level of
whole matrices: which periods a chunk runs, what it starts from, what it writes, what is compared at the boundaries and how the
parts of the log-likelihood and of the EM sums add up.  The lane-level model of the kernel's arithmetic is tile_emul.py; this file
is about the orchestration only.  TEST / DEVELOPMENT INFRASTRUCTURE (tests/test_tile_chunk_model_cpu.py)."""
import numpy as np

from oracle.kalman_oracle import LOG2PI, collapse


def chunk_geometry(B, T, nc_req=0, W=16, num_cu=256, wmax=32, ncmax=16):
    """recursion_tile_chunks(): (chunks per replicate, periods per chunk, warm-up)"""
    W = min((W + 1) & ~1, wmax)
    if nc_req == 1:
        return 1, T, W
    want = min(nc_req if nc_req > 1 else (2 * num_cu) // B, ncmax)
    while want > 1:
        lc = 2 * ((T + 2 * want - 1) // (2 * want))
        if lc >= 4 * W and T - (want - 1) * lc >= W + 2:
            return want, lc, W
        want -= 1
    return 1, T, W


def chunked_pass(x, Lam, R, A, Q, mu0, P0, NC, Lc, W, tol=1e-10):
    """One smoother pass as NC chunks.  Returns dict(loglik, f_smooth, P_smooth, f0_smooth, P0_smooth, S_P, S_U, fail, worst) where
    S_P = sum_{t=1..T} P_t|T, S_U = sum of the lag-one covariances (the EM sums the chunks' parts add up to), `worst` the largest
    relative boundary difference."""
    T, N = x.shape
    r = Lam.shape[1]
    b, s, n, ld, C = collapse(x, Lam, R)
    Cfull = (Lam.T / R) @ Lam
    ldfull = np.log(R).sum()
    Qi = np.linalg.inv(Q); Psi = A.T @ Qi; Phi = A.T @ Qi @ A
    P0i = np.linalg.inv(P0)
    Z = np.zeros((T, r, r)); J = np.zeros((T, r, r)); w = np.zeros((T, r))       # THE table (own periods only)
    f_s = np.full((T, r), np.nan); P_s = np.full((T, r, r), np.nan)               # rows t = period t + 1
    bst = {}                                                                      # (chunk, kind) -> (matrix, vector)
    parts, SP, SU = [], np.zeros((r, r)), np.zeros((r, r))
    out = {}
    for c in range(NC):
        first, last = c == 0, c == NC - 1
        s0 = c * Lc
        e0 = T if last else s0 + Lc
        tb = 0 if first else s0 - W
        te = T + 1 if last else e0 + W
        if first:
            Om, xi, qacc = P0i.copy(), P0i @ mu0, mu0 @ P0i @ mu0
        else:
            Om, xi, qacc = Qi + Cfull, np.zeros(r), 0.0                           # the guess the warm-up forgets
        xZ = {}; part = 0.0
        sum_ldz = 0.0
        for t in range(tb, te):
            own = s0 <= t < e0
            if not first and t == s0:
                bst[(c, 0)] = (Om.copy(), xi.copy())
            if not last and t == e0:
                bst[(c, 1)] = (Om.copy(), xi.copy())
            if t == T:                                                            # terminal step (last chunk)
                PT = np.linalg.inv(Om); fT = PT @ xi
                part += np.linalg.slogdet(Om)[1]
                qacc -= xi @ fT
                break
            Zt = np.linalg.inv(Om + Phi)
            Jt = Zt @ Psi
            wt = Zt @ xi
            if own:
                Z[t], J[t], w[t] = Zt, Jt, wt
                sum_ldz += np.linalg.slogdet(Zt)[1]
                qacc -= xi @ wt
                part += s[t] + n[t] * LOG2PI + ld[t]
            elif t >= e0:
                xZ[t] = (Zt, Jt, wt)                                              # the private table of the extra periods
            xi = Psi.T @ wt + b[t]
            Om = Qi - Psi.T @ Jt + (Cfull if n[t] == N else C[t])
        part += qacc - sum_ldz
        if first:
            part += np.linalg.slogdet(P0)[1] + T * np.linalg.slogdet(Q)[1]
        parts.append(part)
        # backward
        if last:
            Ps, fs = PT, fT
            f_s[T - 1], P_s[T - 1] = fs, Ps
            SPc = Ps.copy()
            tl = T - 1
        else:
            Ps, fs = xZ[e0 + W - 1][0].copy(), xZ[e0 + W - 1][2].copy()           # the guess: (Z, w) of the last extra period
            SPc = np.zeros((r, r))
            tl = e0 + W - 1
        SUc = np.zeros((r, r))
        for t in range(tl, s0 - 1, -1):
            Zt, Jt, wt = (Z[t], J[t], w[t]) if t < e0 else xZ[t]
            U = Ps @ Jt.T
            Ps = Zt + Jt @ U
            fs = wt + Jt @ fs
            if t < e0:
                SUc += U
                if t > 0:
                    SPc += Ps
                    f_s[t - 1], P_s[t - 1] = fs, Ps
            if not last and t == e0:
                bst[(c, 2)] = (Ps.copy(), fs.copy())
        if not first:
            bst[(c, 3)] = (Ps.copy(), fs.copy())
        else:
            out["f0_smooth"], out["P0_smooth"] = fs, Ps
        SP += SPc; SU += SUc
    worst = 0.0
    for c in range(1, NC):
        for got, ref in ((bst[(c, 0)], bst[(c - 1, 1)]), (bst[(c - 1, 2)], bst[(c, 3)])):
            worst = max(worst, np.abs(got[0] - ref[0]).max() / np.abs(ref[0]).max(),
                        np.abs(got[1] - ref[1]).max() / max(np.abs(ref[1]).max(), 1.0))
    out.update(loglik=-0.5 * float(sum(parts)), f_smooth=f_s, P_smooth=P_s, S_P=SP, S_U=SU, fail=not (worst <= tol), worst=worst)
    return out
