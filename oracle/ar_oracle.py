"""TEST INFRASTRUCTURE ONLY (CPU oracle) -- the parametric DFM with autoregressive idiosyncratic terms:

    x_it = lam_i' f_t + e_it,   e_it = rho_i1 e_i,t-1 + .. + rho_iq e_i,t-q + eps_it,  eps_it ~ N(0, sig2_i)
    f_t  = A_1 f_{t-1} + .. + A_p f_{t-p} + eta_t,  eta_t ~ N(0, Q)

with rho / sig2 in the role of the reference's `uar_coef` / `uar_ser`^2 (AR(n_uarlag) of the loading-regression
residuals without intercept, dfm_functions.ipynb:305-311, 405-412).  Quasi-differencing keeps the state r m wide,
m = max(p, q + 1):   x~_it = x_it - sum_l rho_il x_i,t-l = lam_i' (f_t - sum_l rho_il f_{t-l}) + eps_it,  t = q+1..T,
i.e. loadings [lam_i, -rho_i1 lam_i, .., -rho_iq lam_i, 0..] on z_t = (f_t, .., f_{t-m+1}), VAR(p) companion
transition, singular innovation covariance.  The likelihood is conditional on the first q rows; a cell of x~ is
missing when x_it or any of its q lags is.  z_q ~ N(mu0, P0).

PARITY UNPINNED against the reference (`Parametric` is declared, not implemented: dfm_functions.ipynb:21-23).  Pinned by
tests/test_oracle_ar.py: an independent construction of the Gaussian density of x_{q+1..T} given x_{1..q} from the
model's own recursion (no quasi-differencing), and q = 0 == varp_oracle.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
from __future__ import annotations

import numpy as np

from . import kalman_oracle as ko
from . import varp_oracle as vo


def quasi_difference(x, rho):
    """x (T, N) with NaN for missing, rho (N, q)  ->  (T - q, N); NaN where x_it or one of its q lags is NaN."""
    T, N = x.shape
    q = rho.shape[1]
    out = x[q:].copy()
    for l in range(1, q + 1):
        out = out - rho[:, l - 1][None, :] * x[q - l:T - l]
    return out


def ar_loadings(Lam, rho, m):
    """(N, r), (N, q) -> (N, r m): [lam_i, -rho_i1 lam_i, .., -rho_iq lam_i, 0 ..]."""
    N, r = Lam.shape
    q = rho.shape[1]
    out = np.zeros((N, r * m))
    out[:, :r] = Lam
    for l in range(1, q + 1):
        out[:, l * r:(l + 1) * r] = -rho[:, l - 1][:, None] * Lam
    return out


def state_lags(p, q):
    return max(p, q + 1)


def kfs_pass_ar(x, Lam, sig2, rho, Avar, Q, mu0, P0):
    """Smoother pass; Avar (r, r p), mu0 (r m,), P0 (r m, r m) for z_q.  Returns ko.kfs_pass' dict for the T - q
    quasi-differenced rows (state r m wide; f_t = first r components)."""
    N, r = Lam.shape
    p = Avar.shape[1] // r
    q = rho.shape[1]
    m = state_lags(p, q)
    k = r * m
    Ak = np.zeros((r, k)); Ak[:, :r * p] = Avar
    M, Qk = vo.companion(Ak, Q, m)
    return ko.kfs_pass(quasi_difference(x, rho), ar_loadings(Lam, rho, m), sig2, M, Qk, mu0, P0, lag_one=False)


def direct_conditional_density(x, Lam, sig2, rho, Avar, Q, mu0, P0):
    """Independent check (balanced x only, tiny sizes): x_{q+1..T} given x_{1..q} is an affine map of the Gaussian
    vector (z_q, eta_{q+1..T}, eps_{q+1..T}) through the model's OWN recursion (e_t from its AR, f_t from its VAR);
    mean and covariance by evaluating that map on the basis vectors.  Returns (loglik, E[f | x], stacked (T-q, r))."""
    T, N = x.shape
    r = Lam.shape[1]
    p = Avar.shape[1] // r
    q = rho.shape[1]
    m = state_lags(p, q)
    k = r * m
    n = T - q
    dim = k + n * r + n * N

    def run(u):
        z = u[:k]
        eta = u[k:k + n * r].reshape(n, r)
        eps = u[k + n * r:].reshape(n, N)
        # factor history f_{q-m+1..q} from z_q = (f_q, f_{q-1}, ..)
        hist = [z[l * r:(l + 1) * r] for l in range(m)]          # hist[l] = f_{q-l}
        # idiosyncratic history e_{q-l} = x_{q-l} - Lam f_{q-l}, l = 0..q-1
        eh = [x[q - 1 - l] - Lam @ hist[l] for l in range(q)]
        xs = np.empty((n, N)); fs = np.empty((n, r))
        for t in range(n):
            f = sum(Avar[:, l * r:(l + 1) * r] @ hist[l] for l in range(p)) + eta[t]
            e = sum(rho[:, l] * eh[l] for l in range(q)) + eps[t] if q else eps[t]
            xs[t] = Lam @ f + e
            fs[t] = f
            hist = [f] + hist[:-1]
            if q:
                eh = [e] + eh[:-1]
        return xs.ravel(), fs.ravel()

    mean_u = np.zeros(dim); mean_u[:k] = mu0
    S = np.zeros((dim, dim))
    S[:k, :k] = P0
    for t in range(n):
        S[k + t * r:k + (t + 1) * r, k + t * r:k + (t + 1) * r] = Q
    S[k + n * r:, k + n * r:] = np.diag(np.tile(sig2, n))
    x0, f0 = run(np.zeros(dim))
    Jx = np.empty((n * N, dim)); Jf = np.empty((n * r, dim))
    for j in range(dim):
        u = np.zeros(dim); u[j] = 1.0
        a, b = run(u)
        Jx[:, j] = a - x0; Jf[:, j] = b - f0
    mx = x0 + Jx @ mean_u
    mf = f0 + Jf @ mean_u
    Sxx = Jx @ S @ Jx.T
    Sfx = Jf @ S @ Jx.T
    y = x[q:].ravel()
    L = np.linalg.cholesky(Sxx)
    w = np.linalg.solve(L, y - mx)
    ll = -0.5 * (y.size * ko.LOG2PI + 2.0 * np.log(np.diag(L)).sum() + w @ w)
    Ef = mf + Sfx @ np.linalg.solve(Sxx, y - mx)
    return float(ll), Ef.reshape(n, r)


# ------------------------------------------------------------------------------------------------------------------
# Joint estimation of (lam, rho, sig2, A_1..A_p, Q) by ECM (the re-estimation of the reference's
# `uar_coef` / `uar_ser`, dfm_functions.ipynb:305-311, 405-412, inside the parametric model).
#
# E-step: kfs_pass_ar at the current parameters -> smoothed moments of z_t = (f_t, .., f_{t-m+1}), t = q+1..T, and
# the log-likelihood conditional on the first q rows.
# CM-steps (each maximises the expected complete-data log-likelihood over one block with the others fixed, so the
# observed-data likelihood cannot decrease -- Meng & Rubin 1993):
#   (1) transition: [A_1..A_p] = S10[:r, :rp] S00[:rp, :rp]^-1 (the VAR has p lags although the state carries m >= p),
#       Q = (S11[:r,:r] - A S10[:r, :rp]') / T~,  mu0, P0 = smoothed moments of z_q;
#   (2) loadings given rho: with a_i = (1, -rho_i1, .., -rho_iq), g_it = sum_l a_il f_{t-l}, x~_it = sum_l a_il x_i,t-l,
#       lam_i = [sum_t E g g']^-1 sum_t x~_it E g_it          (t over the periods where x~_it is observed);
#   (3) rho given the NEW loadings: u_itl = x_i,t-l - lam_i' f_{t-l}, U_i = sum_t E[u_it u_it'] ((q+1) x (q+1)),
#       rho_i = U_i[1:,1:]^-1 U_i[1:,0];
#   (4) sig2_i = a_i' U_i a_i / n_i with the new a_i.
# ------------------------------------------------------------------------------------------------------------------
def em_step_ar(x, Lam, sig2, rho, Avar, Q, mu0, P0):
    x = np.asarray(x, float)
    T, N = x.shape
    r = Lam.shape[1]
    p = Avar.shape[1] // r
    q = rho.shape[1]
    m = state_lags(p, q)
    k = r * m
    Tq = T - q
    Ak = np.zeros((r, k)); Ak[:, :r * p] = Avar
    M, Qk = vo.companion(Ak, Q, m)
    out = ko.kfs_pass(quasi_difference(x, rho), ar_loadings(Lam, rho, m), sig2, M, Qk, mu0, P0, lag_one=True)
    zs, Ps, Pl = out["f_smooth"], out["P_smooth"], out["P_lag"]
    z0, P0s = out["f0_smooth"], out["P0_smooth"]
    Ez = zs[:, :, None] * zs[:, None, :] + Ps                      # E[z_t z_t'], t = q+1..T
    # (1) transition
    S11 = Ez.sum(0)
    S00 = S11 - Ez[-1] + (np.outer(z0, z0) + P0s)
    zprev = np.vstack([z0[None, :], zs[:-1]])
    S10 = (zs[:, :, None] * zprev[:, None, :] + Pl).sum(0)
    ka = r * p
    A_new = np.linalg.solve(S00[:ka, :ka].T, S10[:r, :ka].T).T
    Q_new = (S11[:r, :r] - A_new @ S10[:r, :ka].T) / Tq
    Q_new = 0.5 * (Q_new + Q_new.T)
    # (2)-(4) per series
    Lam_new = Lam.copy(); rho_new = rho.copy(); sig2_new = sig2.copy()
    X = np.stack([x[q - l:T - l] for l in range(q + 1)], axis=2)   # X[t, i, l] = x_{i, t+q-l}
    obs = ~np.isnan(X).any(axis=2)                                 # x~_it observed
    zb = zs.reshape(Tq, m, r)[:, :q + 1]                           # E f_{t-l}
    Eb = Ez.reshape(Tq, m, r, m, r)[:, :q + 1, :, :q + 1, :]       # E[f_{t-l} f_{t-l'}']  [t, l, :, l', :]
    Pb = Ps.reshape(Tq, m, r, m, r)[:, :q + 1, :, :q + 1, :]
    for i in range(N):
        w = obs[:, i]
        n = int(w.sum())
        if n < r + q + 1:
            continue                                               # too few quasi-differenced cells: series left as is
        a = np.concatenate([[1.0], -rho[i]])
        Xi = X[w, i]                                               # (n, q+1)
        xt = Xi @ a
        g = np.einsum("l,tlc->tc", a, zb[w])
        EG = np.einsum("l,tlcmd,m->cd", a, Eb[w], a)
        lam = np.linalg.solve(EG, g.T @ xt)
        Lam_new[i] = lam
        u = Xi - np.einsum("tlc,c->tl", zb[w], lam)                # E u_itl
        U = u.T @ u + np.einsum("c,tlcmd,d->lm", lam, Pb[w], lam)
        if q > 0:
            rr = np.linalg.solve(U[1:, 1:], U[1:, 0])
            rho_new[i] = rr
            a = np.concatenate([[1.0], -rr])
        sig2_new[i] = (a @ U @ a) / n
    new = dict(Lam=Lam_new, sig2=sig2_new, rho=rho_new, Avar=A_new, Q=Q_new, mu0=z0.copy(), P0=0.5 * (P0s + P0s.T))
    return new, out["loglik"], out


def em_ar(x, params, max_iter=10, tol=0.0):
    """ECM loop with ko.em's bookkeeping (path[k] = conditional log-likelihood at the parameters entering iteration k)."""
    cur = {k: np.array(v, float) for k, v in params.items()}
    path = []
    out = None
    for it in range(max_iter):
        new, ll, out = em_step_ar(x, **cur)
        path.append(ll)
        if it >= 1 and tol > 0.0:
            if (path[-1] - path[-2]) / (0.5 * (abs(path[-1]) + abs(path[-2]))) < tol:
                break
        cur = new
    return cur, np.array(path), out


def synth_ar(b, N, T, r, p, q, seed=ko.SEED0, missing=0.0):
    """Seeded panel with VAR(p) factors and AR(q) idiosyncratic terms, and a (deliberately rough) start for em_ar."""
    rng = np.random.default_rng([seed, b, p, q, 7])
    x0 = vo.synth_varp(b, N, T + 30, r, p, seed=seed)              # factor part + white noise ...
    rho = np.zeros((N, q))
    if q:
        rho[:, 0] = rng.uniform(-0.3, 0.6, N)
        for l in range(1, q):
            rho[:, l] = rng.uniform(-0.15, 0.15, N) / l
    e = np.zeros((T + 30, N))
    eps = 0.6 * rng.standard_normal((T + 30, N))
    for t in range(T + 30):                                        # ... plus an AR(q) component
        e[t] = eps[t] + sum(rho[:, l] * e[t - 1 - l] for l in range(q) if t - 1 - l >= 0)
    x = (x0 + e)[-T:]
    x = (x - x.mean(0)) / x.std(0)
    if missing > 0.0:
        x = np.where(rng.random((T, N)) < missing, np.nan, x)
    m = state_lags(p, q)
    xb = np.where(np.isnan(x), 0.0, x)
    base, F = vo.varp_init(xb, r, p)
    k = r * m
    Z = np.hstack([F[m - 1 - l:T - l] for l in range(m)])
    P0 = Z.T @ Z / Z.shape[0] + 1e-3 * np.eye(k)
    start = dict(Lam=base["Lam"], sig2=np.maximum(base["R"], 0.05), rho=np.full((N, q), 0.1), Avar=base["Avar"], Q=base["Q"],
                 mu0=np.zeros(k), P0=0.5 * (P0 + P0.T))
    return x, start
