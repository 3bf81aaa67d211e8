"""Lane-level NumPy model of csrc/recursion_chunk.hip: the information-form filter + Z-smoother of oracle/info_form.py cut into
64 time chunks of L periods, one chunk per LANE of a wave, every lane running the plain sequential recursion on its own 8 x 8
matrices (held in its registers on the GPU).  A lane starts W periods before its chunk from a guess (forward: Om_p = Q^-1,
xi = b; backward: P = 0, f = 0) and relies on the filter forgetting its initial condition; the forgetting is CHECKED (a lane's state
after its warm-up against the neighbouring lane's state at the end of its chunk), and a replicate whose check fails is handed to
the sequential kernel.  TEST INFRASTRUCTURE ONLY (tests/test_chunk_model_cpu.py keeps the model honest against
oracle/kalman_oracle.py; the kernel itself is compared with the oracle in the GPU tests)."""
import numpy as np

from oracle.kalman_oracle import LOG2PI, collapse

NL = 64


def chunk_len(T, Lmin=4):
    return max((T + NL - 1) // NL, Lmin)


def chunk_pass(x, Lam, R, A, Q, mu0, P0, W=8, L=None, tol=1e-11):
    """Returns (f_smooth, P_smooth, loglik, info) -- info has `ok` (all boundary checks passed) and the check residuals."""
    T, N = x.shape
    r = Lam.shape[1]
    b, s, n, ld, C = collapse(x, Lam, R)
    L = L or chunk_len(T)
    NS = L + W
    Qi = np.linalg.inv(Q); K = Qi @ A; Phi = K.T @ A
    QPhi = Qi + Phi
    P0i = np.linalg.inv(P0)
    lanes = np.arange(NL)
    c0 = L * lanes                                    # first counted period of a lane
    jtop = (T - 1) // L

    # ---------------------------------------------------------------- forward
    M = np.zeros((NL, r, r)); xi = np.zeros((NL, r))
    Zs = np.full((T, r, r), np.nan); ws = np.full((T, r), np.nan)     # the scratch table, written by the owning lane only
    ldz = np.zeros(NL); sxw = np.zeros(NL)
    M_start = np.zeros((NL, r, r)); xi_start = np.zeros((NL, r))
    Mcap = None
    for u in range(NS):
        t = c0 - W + u
        if u == 0:                                    # guess: a step from "J = 0" with the data of period t - 1
            tm = np.clip(t - 1, 0, T - 1)
            M = QPhi[None] + C[tm]
            xi = b[tm].copy()
        exact = t == 0                                # lane(s) whose window reaches the initial state
        M[exact] = P0i + Phi
        xi[exact] = P0i @ mu0
        if u == W:
            M_start = M.copy(); xi_start = xi.copy()
        valid = (t >= 0) & (t < T)
        tc = np.clip(t, 0, T - 1)
        Z = np.linalg.inv(M)
        w = np.einsum("lij,lj->li", Z, xi)
        counted = valid & (u >= W)
        for l in np.nonzero(counted)[0]:
            Zs[t[l]] = Z[l]; ws[t[l]] = w[l]
        ldz += np.where(counted, np.linalg.slogdet(M)[1], 0.0)
        sxw += np.where(counted, np.einsum("li,li->l", xi, w), 0.0)
        J = Z @ K.T
        Ct = np.where(valid[:, None, None], C[tc], 0.0)
        bt = np.where(valid[:, None], b[tc], 0.0)
        M = QPhi[None] + Ct - K[None] @ J
        xi = np.einsum("ij,lj->li", K, w) + bt
        if u == W + (T - 1) % L:                      # the top lane has just taken period T - 1
            Mcap = M[jtop].copy(); xicap = xi[jtop].copy()
    M_end, xi_end = M, xi
    # boundary checks: lane j's state after its warm-up against lane j - 1's state at the end of its chunk
    res_f = np.zeros(NL)
    for j in range(1, jtop + 1):
        dM = np.abs(M_start[j] - M_end[j - 1]).max() / np.abs(M_end[j - 1]).max()
        dx = np.abs(xi_start[j] - xi_end[j - 1]).max() / max(np.abs(xi_end[j - 1]).max(), 1e-300)
        res_f[j] = max(dM, dx)
    OmT = Mcap - Phi
    PT = np.linalg.inv(OmT); fT = PT @ xicap
    LD = np.linalg.slogdet(OmT)[1] + np.linalg.slogdet(P0)[1] + T * np.linalg.slogdet(Q)[1] + ldz.sum()
    QD = s.sum() + mu0 @ P0i @ mu0 - xicap @ fT - sxw.sum()
    loglik = -0.5 * (n.sum() * LOG2PI + ld.sum() + LD + QD)

    # ---------------------------------------------------------------- backward
    P = np.zeros((NL, r, r)); f = np.zeros((NL, r))
    f_s = np.full((T + 1, r), np.nan); P_s = np.full((T + 1, r, r), np.nan)
    f_s[T], P_s[T] = fT, PT
    P_startb = np.zeros((NL, r, r)); f_startb = np.zeros((NL, r))
    for u in range(NS):
        t = c0 + NS - 1 - u                            # the step that produces state t from state t + 1
        term = t + 1 == T
        P[term] = PT; f[term] = fT
        if u == W:
            P_startb = P.copy(); f_startb = f.copy()
        valid = (t >= 0) & (t < T)
        tc = np.clip(t, 0, T - 1)
        Z = np.where(valid[:, None, None], Zs[tc], 0.0)
        w = np.where(valid[:, None], ws[tc], 0.0)
        G = K.T[None] @ P @ K[None]
        y = np.einsum("ji,lj->li", K, f)               # K' f
        P = Z + Z @ G @ Z
        f = w + np.einsum("lij,lj->li", Z, y)
        counted = valid & (u >= W)
        for l in np.nonzero(counted)[0]:
            f_s[t[l]] = f[l]; P_s[t[l]] = P[l]
    res_b = np.zeros(NL)
    for j in range(0, jtop):
        if L * (j + 1) >= T:
            continue
        dP = np.abs(P_startb[j] - P_s[L * (j + 1)]).max() / np.abs(P_s[L * (j + 1)]).max()
        df = np.abs(f_startb[j] - f_s[L * (j + 1)]).max() / max(np.abs(f_s[L * (j + 1)]).max(), 1e-300)
        res_b[j] = max(dP, df)
    ok = bool(res_f.max() <= tol and res_b.max() <= tol)
    info = dict(ok=ok, res_f=res_f, res_b=res_b, f0=f_s[0], P0s=P_s[0], L=L, W=W)
    return f_s[1:], P_s[1:], float(loglik), info
