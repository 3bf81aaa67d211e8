"""TEST INFRASTRUCTURE ONLY (CPU oracle) -- the parametric DFM with VAR(p) factor dynamics:

    x_t = Lam f_t + e_t,  e_it ~ N(0, R_i);    f_t = A_1 f_{t-1} + ... + A_p f_{t-p} + eta_t,  eta_t ~ N(0, Q)

in companion form z_t = (f_t, ..., f_{t-p+1}) (k = r p), z_t = M z_{t-1} + G eta_t, the layout the reference's
`fill_matrices!` builds for its factor VAR (dfm_functions.ipynb:477-492: M = [betahat[2:end,:]'; I 0], Q = [I 0]).
SURVEY.md §8(f3).  The reference declares `Parametric` without implementing it (dfm_functions.ipynb:21-23) and
its DFMModel carries `n_factorlag` (:120-146): this is the model that slot would fit.

PARITY UNPINNED against the reference (no implementation there).  Pinned instead by (tests/test_oracle_varp.py):
the brute-force joint-Gaussian log-density and conditional moments on tiny problems, and p = 1 == kalman_oracle.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
from __future__ import annotations

import numpy as np

from . import kalman_oracle as ko


def companion(Avar, Q, p):
    """Avar (r, r p) = [A_1 .. A_p], Q (r, r)  ->  M (k, k), Qk (k, k) (singular for p > 1)."""
    r = Avar.shape[0]
    k = r * p
    M = np.zeros((k, k))
    M[:r] = Avar
    if p > 1:
        M[r:, :k - r] = np.eye(k - r)                     # dfm_functions.ipynb:484-486 (shift block)
    Qk = np.zeros((k, k))
    Qk[:r, :r] = Q
    return M, Qk


def kfs_pass_varp(x, Lam, R, Avar, Q, mu0, P0, p):
    """Full smoother pass of the companion model through the covariance-form oracle (ko.kfs_pass), loadings
    [Lam, 0].  mu0 (k,), P0 (k, k): moments of z_0.  Returns ko.kfs_pass' dict (k-dimensional state)."""
    N, r = Lam.shape
    k = r * p
    M, Qk = companion(Avar, Q, p)
    LamK = np.zeros((N, k))
    LamK[:, :r] = Lam
    return ko.kfs_pass(x, LamK, R, M, Qk, mu0, P0, lag_one=True)


def em_step_varp(x, Lam, R, Avar, Q, mu0, P0, p):
    """One EM iteration.  E-step: kfs_pass_varp.  M-step (closed form, constraints imposed):
      [A_1..A_p] = S10[:r, :] S00^-1,  Q = (S11[:r,:r] - Avar S10[:r,:]') / T     (z-moments, k-dimensional)
      Lam_i, R_i from the moments of f_t = z_t[:r] exactly as ko.em_step;  mu0, P0 = smoothed moments of z_0."""
    x = np.asarray(x, float)
    T, N = x.shape
    r = Lam.shape[1]
    out = kfs_pass_varp(x, Lam, R, Avar, Q, mu0, P0, p)
    zs, Ps, Pl = out["f_smooth"], out["P_smooth"], out["P_lag"]
    z0, P0s = out["f0_smooth"], out["P0_smooth"]
    Ez = zs[:, :, None] * zs[:, None, :] + Ps
    S11 = Ez.sum(0)
    S00 = S11 - Ez[-1] + (np.outer(z0, z0) + P0s)
    zprev = np.vstack([z0[None, :], zs[:-1]])
    S10 = (zs[:, :, None] * zprev[:, None, :] + Pl).sum(0)
    A_new = np.linalg.solve(S00.T, S10[:r].T).T
    Q_new = (S11[:r, :r] - A_new @ S10[:r].T) / T
    Q_new = 0.5 * (Q_new + Q_new.T)
    fs = zs[:, :r]
    Ef = Ez[:, :r, :r]
    F11 = S11[:r, :r]
    obs = ~np.isnan(x)
    x0 = np.where(obs, x, 0.0)
    Sxf = x0.T @ fs
    Lam_new = np.empty_like(Lam); R_new = np.empty_like(R)
    if obs.all():
        Lam_new = np.linalg.solve(F11.T, Sxf.T).T
        Sxx = (x0 * x0).sum(0)
        R_new = (Sxx - 2.0 * np.einsum("ik,ik->i", Lam_new, Sxf) + np.einsum("ik,kl,il->i", Lam_new, F11, Lam_new)) / T
    else:
        for i in range(N):
            w = obs[:, i]
            Sff_i = Ef[w].sum(0)
            lam = np.linalg.solve(Sff_i, Sxf[i])
            Lam_new[i] = lam
            R_new[i] = ((x0[w, i] ** 2).sum() - 2.0 * lam @ Sxf[i] + lam @ Sff_i @ lam) / w.sum()
    new = dict(Lam=Lam_new, R=R_new, Avar=A_new, Q=Q_new, mu0=z0.copy(), P0=0.5 * (P0s + P0s.T))
    return new, out["loglik"], out


def em_varp(x, params, p, max_iter=10, tol=0.0):
    """EM loop with ko.em's bookkeeping (path[k] = log-likelihood at the parameters entering iteration k)."""
    q = {k: np.array(v, float) for k, v in params.items()}
    path = []
    out = None
    for k in range(max_iter):
        new, ll, out = em_step_varp(x, p=p, **q)
        path.append(ll)
        if k >= 1 and tol > 0.0:
            if (path[-1] - path[-2]) / (0.5 * (abs(path[-1]) + abs(path[-2]))) < tol:
                break
        q = new
    return q, np.array(path), out


def varp_init(x, r, p):
    """Start for a balanced standardised panel: ko.pca_init's PCA / loadings / R; [A_1..A_p] = OLS VAR(p) of the
    PCA factors without constant (rows p..T-1), Q = residual covariance (divisor T - p); z_0 ~ N(0, P0) with
    P0 = sample second moment of the stacked lag vectors (f_t, .., f_{t-p+1}), t = p-1..T-1."""
    T, N = x.shape
    base, F = ko.pca_init(x, r)
    Z = np.hstack([F[p - 1 - l:T - l] for l in range(p)])          # row j: (f_{p-1+j}, ..., f_j)
    Y, Xl = F[p:], Z[:-1]
    Avar = np.linalg.solve(Xl.T @ Xl, Xl.T @ Y).T
    e = Y - Xl @ Avar.T
    Q = e.T @ e / (T - p)
    Q = 0.5 * (Q + Q.T)
    P0 = Z.T @ Z / Z.shape[0]
    return dict(Lam=base["Lam"], R=base["R"], Avar=Avar, Q=Q, mu0=np.zeros(r * p), P0=0.5 * (P0 + P0.T)), F


def synth_varp(b, N, T, r, p, seed=ko.SEED0, missing=0.0):
    """Seeded VAR(p) panel: stationary by construction (A_l = a_l-diagonal with sum |a_l| < 1)."""
    rng = np.random.default_rng([seed, b, p])
    Lam = rng.standard_normal((N, r))
    w = 0.5 ** np.arange(1, p + 1)
    w = 0.85 * w / w.sum()
    sg = np.where(np.arange(p) % 2 == 0, 1.0, -1.0)
    Avar = np.hstack([np.diag(np.linspace(0.6, 1.0, r)) * w[l] * sg[l] for l in range(p)])
    Q = np.diag(np.linspace(0.5, 1.0, r))
    R = rng.uniform(0.5, 1.5, N)
    f = np.zeros((T + 50 + p, r))
    for t in range(p, f.shape[0]):
        z = np.concatenate([f[t - 1 - l] for l in range(p)])
        f[t] = Avar @ z + np.sqrt(np.diag(Q)) * rng.standard_normal(r)
    f = f[-T:]
    x = f @ Lam.T + np.sqrt(R) * rng.standard_normal((T, N))
    x = (x - x.mean(0)) / x.std(0)
    if missing > 0.0:
        x = np.where(rng.random((T, N)) < missing, np.nan, x)
    return x
