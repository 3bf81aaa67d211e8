"""CPU fp64 oracle for the parametric (state-space) dynamic-factor-model path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (dynamic_factor_models_amd/)
may import this module; only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg do, and only as the checker.

PARITY UNPINNED (by the reference).  The reference declares the dispatch tag
`Parametric` (dfm_functions.ipynb:21-23) and writes down the state-space form
y_t = Q z_t, z_t = M z_{t-1} + G u_t (dfm_functions.ipynb:30-34, 477-492) but contains
no Kalman filter, RTS smoother, log-likelihood or EM code.  This file therefore
restates the *published* algorithm (Shumway & Stoffer 1982; Watson & Engle 1983;
Doz, Giannone & Reichlin 2011/2012; Banbura & Modugno 2014 for missing cells;
Jungbacker & Koopman 2015 for the collapsed observation vector), and is pinned by
(i) `brute_force_gaussian` below (joint-Gaussian conditioning, no recursion at all),
(ii) `kfs_pass_textbook` (N x N innovation-covariance filter, independent algebra),
(iii) EM log-likelihood monotonicity, (iv) oracle/dfm_oracle.c (independent C twin).
See tests/test_oracle_kalman.py.

Model:
    x_t = Lam f_t + eps_t,  eps_t ~ N(0, diag R)          t = 1..T   (NaN cell = missing)
    f_t = A f_{t-1} + eta_t, eta_t ~ N(0, Q);  f_0 ~ N(mu0, P0)

Packed symmetric storage used at the C boundary: lower triangle, row-major,
idx(i, j) = i (i + 1) / 2 + j for j <= i.
"""
from __future__ import annotations

import numpy as np

LOG2PI = float(np.log(2.0 * np.pi))


# ----------------------------------------------------------------------------- helpers
def pack_sym(P: np.ndarray) -> np.ndarray:
    """(..., r, r) symmetric -> (..., r(r+1)/2) lower-triangular row-major."""
    r = P.shape[-1]
    il = np.tril_indices(r)
    return P[..., il[0], il[1]]


def unpack_sym(p: np.ndarray, r: int) -> np.ndarray:
    il = np.tril_indices(r)
    out = np.zeros(p.shape[:-1] + (r, r))
    out[..., il[0], il[1]] = p
    out[..., il[1], il[0]] = p
    return out


def collapse(x: np.ndarray, Lam: np.ndarray, R: np.ndarray):
    """Collapsed observation quantities, one row per period.

    b_t = sum_{i in W_t} lam_i x_it / R_i          (T, r)
    s_t = sum_{i in W_t} x_it^2 / R_i              (T,)
    n_t = |W_t|                                    (T,)
    ld_t = sum_{i in W_t} log R_i                  (T,)
    C_t = sum_{i in W_t} lam_i lam_i' / R_i        (T, r, r)
    """
    obs = ~np.isnan(x)
    x0 = np.where(obs, x, 0.0)
    Rinv = 1.0 / R
    b = (x0 * Rinv) @ Lam
    s = (x0 * x0) @ Rinv
    n = obs.sum(axis=1)
    ld = obs.astype(float) @ np.log(R)
    LR = Lam * Rinv[:, None]
    C = np.einsum("ti,ij,ik->tjk", obs.astype(float), LR, Lam)
    C = 0.5 * (C + C.transpose(0, 2, 1))
    return b, s, n, ld, C


# ----------------------------------------------------------------------------- filter + smoother
def kfs_pass(x, Lam, R, A, Q, mu0, P0, lag_one: bool = True):
    """One full Kalman-smoother pass in collapsed form.

    Returns dict with
      loglik              scalar
      f_filt, P_filt      (T, r), (T, r, r)   filtered moments f_{t|t}
      f_pred, P_pred      (T, r), (T, r, r)   predicted moments f_{t|t-1}
      f_smooth, P_smooth  (T, r), (T, r, r)   smoothed moments, t = 1..T
      f0_smooth, P0_smooth                    smoothed moments of the initial state f_0
      P_lag               (T, r, r)           Cov(f_t, f_{t-1} | X), t = 1..T  (if lag_one)
      ll_t                (T,)                log-likelihood increments
    """
    x = np.asarray(x, float)
    T, N = x.shape
    r = Lam.shape[1]
    b, s, n, ld, C = collapse(x, Lam, R)
    I = np.eye(r)

    f_pred = np.empty((T, r)); P_pred = np.empty((T, r, r))
    f_filt = np.empty((T, r)); P_filt = np.empty((T, r, r))
    ll_t = np.empty(T)
    ff, Pf = np.asarray(mu0, float), np.asarray(P0, float)
    for t in range(T):
        fp = A @ ff
        Pp = A @ Pf @ A.T + Q
        Pp = 0.5 * (Pp + Pp.T)
        D = I + C[t] @ Pp
        Pf = np.linalg.solve(D.T, Pp.T).T          # Pp D^{-1}
        Pf = 0.5 * (Pf + Pf.T)
        u = b[t] - C[t] @ fp
        ff = fp + Pf @ u
        sign, logdetD = np.linalg.slogdet(D)
        quad = s[t] - 2.0 * fp @ b[t] + fp @ C[t] @ fp - u @ Pf @ u
        ll_t[t] = -0.5 * (n[t] * LOG2PI + ld[t] + logdetD + quad)
        f_pred[t], P_pred[t], f_filt[t], P_filt[t] = fp, Pp, ff, Pf

    # RTS backward sweep; index t = T-1 .. 0 over (f_1..f_T), then the initial state f_0
    f_s = np.empty((T, r)); P_s = np.empty((T, r, r))
    P_lag = np.empty((T, r, r)) if lag_one else None
    f_s[T - 1], P_s[T - 1] = f_filt[T - 1], P_filt[T - 1]
    for t in range(T - 2, -1, -1):
        J = np.linalg.solve(P_pred[t + 1], A @ P_filt[t]).T      # P_f A' P_p^{-1}
        f_s[t] = f_filt[t] + J @ (f_s[t + 1] - f_pred[t + 1])
        Ps = P_filt[t] + J @ (P_s[t + 1] - P_pred[t + 1]) @ J.T
        P_s[t] = 0.5 * (Ps + Ps.T)
        if lag_one:
            P_lag[t + 1] = P_s[t + 1] @ J.T
    J0 = np.linalg.solve(P_pred[0], A @ P0).T
    f0_s = mu0 + J0 @ (f_s[0] - f_pred[0])
    P0_s = P0 + J0 @ (P_s[0] - P_pred[0]) @ J0.T
    P0_s = 0.5 * (P0_s + P0_s.T)
    if lag_one:
        P_lag[0] = P_s[0] @ J0.T
    return dict(loglik=float(ll_t.sum()), ll_t=ll_t, f_pred=f_pred, P_pred=P_pred,
                f_filt=f_filt, P_filt=P_filt, f_smooth=f_s, P_smooth=P_s,
                f0_smooth=f0_s, P0_smooth=P0_s, P_lag=P_lag)


def kfs_pass_textbook(x, Lam, R, A, Q, mu0, P0):
    """Independent cross-check: the standard filter with the n_t x n_t innovation
    covariance F_t = Lam_t P Lam_t' + R_t (no collapsing, no Woodbury)."""
    x = np.asarray(x, float)
    T, N = x.shape
    r = Lam.shape[1]
    f_pred = np.empty((T, r)); P_pred = np.empty((T, r, r))
    f_filt = np.empty((T, r)); P_filt = np.empty((T, r, r))
    ll = 0.0
    ff, Pf = mu0, P0
    for t in range(T):
        fp = A @ ff
        Pp = A @ Pf @ A.T + Q
        w = ~np.isnan(x[t])
        Z = Lam[w]
        F = Z @ Pp @ Z.T + np.diag(R[w])
        v = x[t, w] - Z @ fp
        Fi_v = np.linalg.solve(F, v)
        K = Pp @ Z.T @ np.linalg.inv(F)
        ff = fp + K @ v
        Pf = Pp - K @ Z @ Pp
        sign, ldF = np.linalg.slogdet(F)
        ll += -0.5 * (w.sum() * LOG2PI + ldF + v @ Fi_v)
        f_pred[t], P_pred[t], f_filt[t], P_filt[t] = fp, Pp, ff, Pf
    f_s = f_filt.copy(); P_s = P_filt.copy()
    for t in range(T - 2, -1, -1):
        J = P_filt[t] @ A.T @ np.linalg.inv(P_pred[t + 1])
        f_s[t] = f_filt[t] + J @ (f_s[t + 1] - f_pred[t + 1])
        P_s[t] = P_filt[t] + J @ (P_s[t + 1] - P_pred[t + 1]) @ J.T
    return dict(loglik=float(ll), f_smooth=f_s, P_smooth=P_s)


def brute_force_gaussian(x, Lam, R, A, Q, mu0, P0):
    """Oracle for the oracle: build the joint Gaussian of the stacked
    state (f_0, f_1, .., f_T) explicitly, stack the observed cells, and get the
    log-density, E[f | X], Cov[f | X] by plain conditioning.  O((T r)^3): tiny sizes only."""
    x = np.asarray(x, float)
    T, N = x.shape
    r = Lam.shape[1]
    # means and covariances of f_0..f_T
    m = [np.asarray(mu0, float)]
    V = [np.asarray(P0, float)]
    for t in range(T):
        m.append(A @ m[-1])
        V.append(A @ V[-1] @ A.T + Q)
    S = np.zeros(((T + 1) * r, (T + 1) * r))
    for a in range(T + 1):
        for c in range(a, T + 1):
            blk = np.linalg.matrix_power(A, c - a) @ V[a]      # Cov(f_c, f_a)
            S[c * r:(c + 1) * r, a * r:(a + 1) * r] = blk
            S[a * r:(a + 1) * r, c * r:(c + 1) * r] = blk.T
    mf = np.concatenate(m)
    rows = []; xs = []; Rs = []
    for t in range(T):
        for i in range(N):
            if not np.isnan(x[t, i]):
                z = np.zeros((T + 1) * r)
                z[(t + 1) * r:(t + 2) * r] = Lam[i]
                rows.append(z); xs.append(x[t, i]); Rs.append(R[i])
    Z = np.array(rows); xv = np.array(xs)
    Sxx = Z @ S @ Z.T + np.diag(Rs)
    v = xv - Z @ mf
    L = np.linalg.cholesky(Sxx)
    w = np.linalg.solve(L, v)
    loglik = -0.5 * (len(xv) * LOG2PI + 2.0 * np.log(np.diag(L)).sum() + w @ w)
    G = S @ Z.T
    cm = mf + G @ np.linalg.solve(Sxx, v)
    cV = S - G @ np.linalg.solve(Sxx, G.T)
    f_all = cm.reshape(T + 1, r)
    P_all = np.stack([cV[t * r:(t + 1) * r, t * r:(t + 1) * r] for t in range(T + 1)])
    P_lag = np.stack([cV[t * r:(t + 1) * r, (t - 1) * r:t * r] for t in range(1, T + 1)])
    return dict(loglik=float(loglik), f_smooth=f_all[1:], P_smooth=P_all[1:],
                f0_smooth=f_all[0], P0_smooth=P_all[0], P_lag=P_lag)


# ----------------------------------------------------------------------------- EM
def em_step(x, Lam, R, A, Q, mu0, P0):
    """One EM iteration: E-step = kfs_pass at the current parameters,
    M-step closed form.  Returns (new_params, loglik_at_current_params, pass_output)."""
    x = np.asarray(x, float)
    T, N = x.shape
    r = Lam.shape[1]
    out = kfs_pass(x, Lam, R, A, Q, mu0, P0, lag_one=True)
    fs, Ps, Pl = out["f_smooth"], out["P_smooth"], out["P_lag"]
    f0, P0s = out["f0_smooth"], out["P0_smooth"]
    Ef = fs[:, :, None] * fs[:, None, :] + Ps                    # E[f_t f_t' | X], t=1..T
    S11 = Ef.sum(0)
    S00 = S11 - Ef[-1] + (np.outer(f0, f0) + P0s)
    fprev = np.vstack([f0[None, :], fs[:-1]])
    S10 = (fs[:, :, None] * fprev[:, None, :] + Pl).sum(0)
    A_new = np.linalg.solve(S00.T, S10.T).T                      # S10 S00^{-1}
    Q_new = (S11 - A_new @ S10.T) / T
    Q_new = 0.5 * (Q_new + Q_new.T)
    obs = ~np.isnan(x)
    x0 = np.where(obs, x, 0.0)
    Lam_new = np.empty_like(Lam); R_new = np.empty_like(R)
    Sxf = x0.T @ fs                                              # (N, r)
    if obs.all():
        Lam_new = np.linalg.solve(S11.T, Sxf.T).T
        Sxx = (x0 * x0).sum(0)
        R_new = (Sxx - 2.0 * np.einsum("ik,ik->i", Lam_new, Sxf)
                 + np.einsum("ik,kl,il->i", Lam_new, S11, Lam_new)) / T
    else:
        for i in range(N):
            w = obs[:, i]
            Sff_i = Ef[w].sum(0)
            lam = np.linalg.solve(Sff_i, Sxf[i])
            Lam_new[i] = lam
            Ti = w.sum()
            R_new[i] = ((x0[w, i] ** 2).sum() - 2.0 * lam @ Sxf[i] + lam @ Sff_i @ lam) / Ti
    new = dict(Lam=Lam_new, R=R_new, A=A_new, Q=Q_new, mu0=f0.copy(), P0=P0s.copy())
    return new, out["loglik"], out


def em(x, params, max_iter=10, tol=0.0):
    """EM loop.  loglik_path[k] is the log-likelihood at the parameters *entering* iteration k
    (so loglik_path[0] is the likelihood at the initial parameters).  Stops after iteration k>=1
    when (ll_k - ll_{k-1}) / (0.5 (|ll_k| + |ll_{k-1}|)) < tol.  Returns (params, path, last pass)."""
    p = {k: np.array(v, float) for k, v in params.items()}
    path = []
    out = None
    for k in range(max_iter):
        new, ll, out = em_step(x, **p)
        path.append(ll)
        if k >= 1 and tol > 0.0:
            if (path[-1] - path[-2]) / (0.5 * (abs(path[-1]) + abs(path[-2]))) < tol:
                break
        p = new
    return p, np.array(path), out


# ----------------------------------------------------------------------------- PCA initialisation
def standardize(x):
    """dfm_functions.ipynb:501-509 (standardize_data): per-series mean and *population* s.d.
    over observed cells."""
    mu = np.nanmean(x, axis=0)
    sd = np.nanstd(x, axis=0)          # ddof=0 == std * sqrt((n-1)/n)
    return (x - mu) / sd, sd


def pca_score(xbal, r):
    """dfm_functions.ipynb:179-183 (pca_score): score = (X V)[:, 1:r] from svd(X).
    Sign fixed here by making the largest-|.| entry of each right singular vector positive
    (LAPACK's sign is arbitrary; all downstream quantities checked are sign-invariant)."""
    _, _, Vt = np.linalg.svd(xbal, full_matrices=False)
    V = Vt[:r].T
    sg = np.sign(V[np.abs(V).argmax(axis=0), np.arange(r)])
    V = V * sg
    return xbal @ V, V


def pca_init(x, r):
    """PCA + OLS initial parameters for EM on a *balanced, standardised* panel (Doz, Giannone &
    Reichlin two-step start; the PCA stage is the reference's pca_score):
      F = pca_score(x, r);  Lam = OLS(x on F) (= V_r);  R_i = mean squared residual;
      A = OLS VAR(1) of F without constant;  Q = residual covariance (divisor T-1);
      mu0 = 0;  P0 = F'F / T."""
    T, N = x.shape
    F, V = pca_score(x, r)
    FtF = F.T @ F
    Lam = np.linalg.solve(FtF, F.T @ x).T
    res = x - F @ Lam.T
    R = (res * res).sum(0) / T
    F0, F1 = F[:-1], F[1:]
    A = np.linalg.solve(F0.T @ F0, F0.T @ F1).T
    e = F1 - F0 @ A.T
    Q = e.T @ e / (T - 1)
    Q = 0.5 * (Q + Q.T)
    return dict(Lam=Lam, R=R, A=A, Q=Q, mu0=np.zeros(r), P0=FtF / T), F


# ----------------------------------------------------------------------------- synthetic DGP
SEED0 = 20160415


def synth_replicate(b, N, T, r, seed=SEED0, missing=0.0):
    """SURVEY.md §8(d) DGP, host-checked subset: lam_ij ~ N(0,1); A = diag(linspace(.5,.9,r));
    Q = I - A A'; R_i ~ U(.5,1.5); f_0 ~ N(0,I); x_t = Lam f_t + sqrt(R) eps_t; then columns
    standardised as standardize_data.  Returns (x, truth-params rescaled to the standardised
    panel).  `missing` = iid probability that a cell is NaN (after standardisation)."""
    rng = np.random.default_rng([seed, b])
    Lam = rng.standard_normal((N, r))
    a = np.linspace(0.5, 0.9, r)
    A = np.diag(a)
    Q = np.eye(r) - A @ A.T
    R = rng.uniform(0.5, 1.5, N)
    f = rng.standard_normal(r)
    x = np.empty((T, N))
    for t in range(T):
        f = A @ f + np.sqrt(np.diag(Q)) * rng.standard_normal(r)
        x[t] = Lam @ f + np.sqrt(R) * rng.standard_normal(N)
    mu = x.mean(0); sd = x.std(0)
    x = (x - mu) / sd
    params = dict(Lam=Lam / sd[:, None], R=R / sd ** 2, A=A, Q=Q, mu0=np.zeros(r), P0=np.eye(r))
    if missing > 0.0:
        m = rng.random((T, N)) < missing
        x = np.where(m, np.nan, x)
    return x, params
