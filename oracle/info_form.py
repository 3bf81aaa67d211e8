"""NumPy statement of the *kernel's* algorithm (information-form filter + "Z-smoother"), used by
tests to pin the algebra the HIP recursion kernel implements against the covariance-form oracle
(oracle/kalman_oracle.py).  TEST INFRASTRUCTURE ONLY.

One SPD inversion per time step serves both filter and smoother.  With constants
    Qi = Q^{-1},  Psi = A' Qi,  Phi = A' Qi A,
and per step t = 0..T-1 (t = 0 is the initial state: Om_f,0 = P0^{-1}, xi_f,0 = P0^{-1} mu0):
    Z_t    = (Om_f,t + Phi)^{-1}                      # = (I - J_t A) P_f,t
    J_t    = Z_t Psi                                  # = P_f,t A' P_p,t+1^{-1}   (RTS gain)
    Om_p   = Qi - Psi' J_t                            # = P_p,t+1^{-1}
    w_t    = Z_t xi_f,t                               # = f_f,t - J_t f_p,t+1
    xi_f,t+1 = Psi' w_t + b_{t+1};   Om_f,t+1 = Om_p + C_{t+1}
terminal:  P_f,T = Om_f,T^{-1},  f_f,T = P_f,T xi_f,T
backward:  P_s,t = Z_t + J_t P_s,t+1 J_t';  f_s,t = w_t + J_t f_s,t+1;  Cov(f_t+1,f_t|X) = P_s,t+1 J_t'
log-likelihood (telescoped):
    sum_t logdet D_t = logdet Om_f,T + logdet P0 + T logdet Q - sum_{t<T} logdet Z_t
    sum_t quad_t     = sum_t s_t + mu0' P0^{-1} mu0 - xi_f,T' f_f,T - sum_{t<T} xi_f,t' w_t
Requires Q and P0 positive definite.
"""
import numpy as np

from .kalman_oracle import LOG2PI, collapse


def kfs_pass_info(x, Lam, R, A, Q, mu0, P0):
    T, N = x.shape
    r = Lam.shape[1]
    b, s, n, ld, C = collapse(x, Lam, R)
    Qi = np.linalg.inv(Q); Psi = A.T @ Qi; Phi = A.T @ Qi @ A
    P0i = np.linalg.inv(P0)
    Om_f = P0i.copy(); xi = P0i @ mu0
    Zs = np.empty((T, r, r)); Js = np.empty((T, r, r)); ws = np.empty((T, r))
    sum_ldz = 0.0; sum_xw = 0.0
    for t in range(T):
        Z = np.linalg.inv(Om_f + Phi)
        sum_ldz += np.linalg.slogdet(Z)[1]
        J = Z @ Psi
        Om_p = Qi - Psi.T @ J
        w = Z @ xi
        sum_xw += xi @ w
        Zs[t], Js[t], ws[t] = Z, J, w
        xi = Psi.T @ w + b[t]
        Om_f = Om_p + C[t]
    PfT = np.linalg.inv(Om_f); ffT = PfT @ xi
    LD = np.linalg.slogdet(Om_f)[1] + np.linalg.slogdet(P0)[1] + T * np.linalg.slogdet(Q)[1] - sum_ldz
    QD = s.sum() + mu0 @ P0i @ mu0 - xi @ ffT - sum_xw
    loglik = -0.5 * (n.sum() * LOG2PI + ld.sum() + LD + QD)
    f_s = np.empty((T + 1, r)); P_s = np.empty((T + 1, r, r)); P_lag = np.empty((T, r, r))
    f_s[T], P_s[T] = ffT, PfT
    for t in range(T - 1, -1, -1):
        U = P_s[t + 1] @ Js[t].T
        P_lag[t] = U
        P_s[t] = Zs[t] + Js[t] @ U
        f_s[t] = ws[t] + Js[t] @ f_s[t + 1]
    return dict(loglik=float(loglik), f_smooth=f_s[1:], P_smooth=P_s[1:], f0_smooth=f_s[0],
                P0_smooth=P_s[0], P_lag=P_lag)


def gj_inverse(M):
    """In-place Gauss-Jordan inverse without pivoting exactly as the kernel does it (row i owned
    by lane i; sweep k broadcasts row k).  Returns (inverse, log det M)."""
    M = np.array(M, float); r = M.shape[0]; logdet = 0.0
    for k in range(r):
        p = M[k].copy(); d = 1.0 / p[k]; logdet += np.log(p[k])
        q = p * d; q[k] = d
        c = M[:, k].copy()
        M -= np.outer(c, q)
        M[:, k] = -c * d
        M[k] = q
    return M, logdet
