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
