"""Oracle for the parametric DFM WITH OBSERVED FACTORS.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED BY THE REFERENCE.  The reference's DFMModel carries `nfac_o` observed factors beside the `nfac_u`
unobserved ones (dfm_functions.ipynb:89-146: `factor` is T x nfac_t, observed columns first -- `lambda[:, nfac_o+1:end]`
are the unobserved loadings, :364) but its estimator is non-functional for nfac_o > 0 (:358-359 assigns an nfac_u-vector to
a row of nfac_t loadings, :371 writes nfac_u columns into nfac_t -- SURVEY App. D 7), and there is no parametric estimator
at all (:21-23).  So there is no behaviour to match; the semantics chosen here are the ones the reference's data layout
implies -- a FAVAR-style measurement equation in which the observed factors are KNOWN REGRESSORS the caller has put into the
first nfac_o columns of `m.factor`:

    x_it = lam_o,i' g_t + lam_u,i' f_t + e_it,   e_it ~ N(0, R_i),        f_t = A f_{t-1} + eta_t,  eta_t ~ N(0, Q)

with g_t (T x r_o) observed without gaps.  The joint dynamics of (g, f) are the reference's own second stage (`estimate_var!`
on the whole `factor` matrix, :444-468), not part of this likelihood.

Algorithm = the EM of oracle/kalman_oracle.py on y_t = x_t - Lam_o g_t for the E-step and the transition M-step, and ONE joint
regression of every series on z_t = (g_t, f_t) for the loadings:  lam_i = [sum_t E z_t z_t']^-1 sum_t x_it E z_t  with
E z z' = [[g g', g f^'], [f^ g', f^ f^' + P_t]] over the periods where x_it is observed;  R_i = E-residual variance.
Pinned by: r_o = 0 reproduces kalman_oracle.em exactly; the log-likelihood is non-decreasing; every M-step block maximises
the expected complete-data log-likelihood (perturbations never increase it); data generated with known (lam_o, lam_u) are
recovered (tests/test_oracle_obs.py)."""
from __future__ import annotations

import numpy as np

from . import kalman_oracle as ko


def residual_panel(x, G, Lam_o):
    """y_t = x_t - Lam_o g_t (NaN stays NaN)."""
    return np.asarray(x, float) - np.asarray(G, float) @ np.asarray(Lam_o, float).T


def loglik_obs(x, G, Lam, R, A, Q, mu0, P0):
    ro = G.shape[1]
    return ko.kfs_pass(residual_panel(x, G, Lam[:, :ro]), Lam[:, ro:], R, A, Q, mu0, P0, lag_one=False)["loglik"]


def em_step_obs(x, G, Lam, R, A, Q, mu0, P0):
    """One EM iteration.  Lam: N x (r_o + r_u), observed-factor loadings FIRST.  Returns (new params, loglik at the current
    parameters, pass output of the E-step)."""
    x = np.asarray(x, float); G = np.asarray(G, float)
    T, N = x.shape
    ro = G.shape[1]
    ru = Lam.shape[1] - ro
    y = residual_panel(x, G, Lam[:, :ro])
    out = ko.kfs_pass(y, Lam[:, ro:], R, A, Q, mu0, P0, lag_one=True)
    fs, Ps, Pl = out["f_smooth"], out["P_smooth"], out["P_lag"]
    f0, P0s = out["f0_smooth"], out["P0_smooth"]
    Ef = fs[:, :, None] * fs[:, None, :] + Ps
    S11 = Ef.sum(0)
    S00 = S11 - Ef[-1] + (np.outer(f0, f0) + P0s)
    fprev = np.vstack([f0[None, :], fs[:-1]])
    S10 = (fs[:, :, None] * fprev[:, None, :] + Pl).sum(0)
    A_new = np.linalg.solve(S00.T, S10.T).T
    Q_new = (S11 - A_new @ S10.T) / T
    Q_new = 0.5 * (Q_new + Q_new.T)
    # joint loadings on z = (g, f): E z z' per period
    re = ro + ru
    z = np.hstack([G, fs])                                        # (T, re)
    Ez = z[:, :, None] * z[:, None, :]
    Ez[:, ro:, ro:] += Ps
    obs = ~np.isnan(x)
    x0 = np.where(obs, x, 0.0)
    Lam_new = np.array(Lam, float); R_new = np.array(R, float)
    for i in range(N):
        w = obs[:, i]
        Ti = int(w.sum())
        if Ti < re + 1:                                           # too few cells for the joint regression: the series keeps its parameters
            continue
        Szz = Ez[w].sum(0)
        Sxz = x0[w, i] @ z[w]
        lam = np.linalg.solve(Szz, Sxz)
        Lam_new[i] = lam
        R_new[i] = ((x0[w, i] ** 2).sum() - 2.0 * lam @ Sxz + lam @ Szz @ lam) / Ti
    new = dict(Lam=Lam_new, R=R_new, A=A_new, Q=Q_new, mu0=f0.copy(), P0=P0s.copy())
    return new, out["loglik"], out


def em_obs(x, G, params, max_iter=10, tol=0.0):
    """EM loop with the bookkeeping of kalman_oracle.em: path[k] = log-likelihood at the parameters ENTERING iteration k; a
    replicate that stops keeps the parameters that entered its last iteration."""
    p = {k: np.array(v, float) for k, v in params.items()}
    path, out = [], None
    for k in range(max_iter):
        new, ll, out = em_step_obs(x, G, **p)
        path.append(ll)
        if k >= 1 and tol > 0.0:
            if (path[-1] - path[-2]) / (0.5 * (abs(path[-1]) + abs(path[-2]))) < tol:
                break
        p = new
    return p, np.array(path), out


def synth_obs(seed, N, T, ru, ro, missing=0.0):
    """A replicate with observed factors: g_t a stationary VAR(1) the caller 'observes', f_t latent."""
    rng = np.random.default_rng(seed)
    x, p = ko.synth_replicate(seed % 1000, N, T, ru, missing=0.0)
    Ag = 0.6 * np.eye(ro) + 0.1 * rng.standard_normal((ro, ro))
    G = np.zeros((T, ro))
    g = rng.standard_normal(ro)
    for t in range(T):
        g = Ag @ g + rng.standard_normal(ro)
        G[t] = g
    G = (G - G.mean(0)) / G.std(0)
    Lam_o = 0.5 * rng.standard_normal((N, ro))
    x = x + G @ Lam_o.T
    if missing > 0.0:
        x = np.where(rng.uniform(size=x.shape) < missing, np.nan, x)
    p = dict(p)
    p["Lam"] = np.hstack([Lam_o, p["Lam"]])
    return x, G, p
