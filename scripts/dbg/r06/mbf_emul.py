"""NumPy model of csrc/recursion_mbf16.hip (round 6): the smoother pass of a model with a SINGULAR innovation covariance (companion
states of VAR(p) factor dynamics) WITHOUT any k x k inversion:
  forward   covariance form; the collapsed observation (b_t, C_t on the first rc state components) enters through a rank-rc update,
            P_f = P_p - P1 W P1',  W = (I + C P11)^-1 C = C - Y G^-1 Y'  with  P11 = L L', Y = C L, G = I + L' C L  (two rc x rc Cholesky
            factors; W and log det(I + C P11) = log det G stay well defined when C is singular or zero);
  backward  the modified Bryson-Frazier recursion (de Jong 1989; Durbin & Koopman 2012, sec. 4.4-4.7) on the adjoint pair (r_t, N_t):
            r_{t-1} = E u_t + L_t' r_t,  N_{t-1} = E W_t E' + L_t' N_t L_t,  L_t = A (I - P1 W E'),  u_t = (I + C P11)^-1 (b - C m_p1);
            s_t|T = m_p + P_p r_{t-1},  V_t = P_p - P_p N_{t-1} P_p,  Cov(s_t, s_{t-1} | X) = (I - P_p,t N_{t-1}) A P_f,t-1.
TEST INFRASTRUCTURE ONLY (tests/test_mbf_model_cpu.py keeps it honest against oracle/varp_oracle.py)."""
import numpy as np

LOG2PI = np.log(2.0 * np.pi)


def mbf_pass(b, C, s, n, ld, A, Q, mu0, P0, rc):
    """b [T, rc], C [T, rc, rc] collapsed observations; s, n, ld [T]; A, Q, P0 [k, k], mu0 [k].
    Returns dict(loglik, f_smooth [T, k], P_smooth [T, k, k], P_lag [T, k, k] (Cov(s_t, s_{t-1})), f0, P0s)."""
    T = b.shape[0]
    k = A.shape[0]
    E = np.zeros((k, rc)); E[:rc] = np.eye(rc)
    mf, Pf = mu0.copy(), P0.copy()
    mp_all = np.empty((T, k)); Pp_all = np.empty((T, k, k)); Pf_all = np.empty((T + 1, k, k))
    W_all = np.empty((T, rc, rc)); u_all = np.empty((T, rc))
    Pf_all[0] = P0
    ll = 0.0
    for t in range(T):
        mp = A @ mf
        Pp = A @ Pf @ A.T + Q
        Pp = 0.5 * (Pp + Pp.T)
        P11 = Pp[:rc, :rc]
        L = np.linalg.cholesky(P11)
        Y = C[t] @ L
        G = np.eye(rc) + L.T @ Y
        Lg = np.linalg.cholesky(G)
        Z = np.linalg.solve(Lg, Y.T)                      # Lg^-1 Y'
        W = C[t] - Z.T @ Z
        e = b[t] - C[t] @ mp[:rc]
        u = e - Z.T @ np.linalg.solve(Lg, L.T @ e)        # (I + C P11)^-1 e = e - C L G^-1 L' e
        P1 = Pp[:, :rc]
        Pf = Pp - P1 @ W @ P1.T
        Pf = 0.5 * (Pf + Pf.T)
        mf = mp + P1 @ u
        logdet = 2.0 * np.log(np.diag(Lg)).sum()
        quad = s[t] - b[t] @ mp[:rc] - e @ mf[:rc]
        ll += -0.5 * (n[t] * LOG2PI + ld[t] + logdet + quad)
        mp_all[t], Pp_all[t], W_all[t], u_all[t], Pf_all[t + 1] = mp, Pp, W, u, Pf
    # backward
    r = np.zeros(k); N = np.zeros((k, k))
    fs = np.empty((T, k)); Ps = np.empty((T, k, k)); Pl = np.empty((T, k, k))
    Phi_next = None
    for t in range(T - 1, -1, -1):
        Pp, W, u = Pp_all[t], W_all[t], u_all[t]
        D = np.eye(k) - Pp[:, :rc] @ W @ E.T
        Lt = A @ D
        r = E @ u + Lt.T @ r
        N = E @ W @ E.T + Lt.T @ N @ Lt
        N = 0.5 * (N + N.T)
        fs[t] = mp_all[t] + Pp @ r
        Phi = np.eye(k) - Pp @ N
        Ps[t] = Phi @ Pp
        Ps[t] = 0.5 * (Ps[t] + Ps[t].T)
        Pl[t] = Phi @ A @ Pf_all[t]                       # Cov(s_t, s_{t-1}): L_{t-1} P_p,t-1 = A P_f,t-1 (t = 0: A P0)
    # initial state: s_1 = A s_0 + eta, no observation at t = 0
    r0 = A.T @ r
    N0 = A.T @ N @ A
    f0 = mu0 + P0 @ r0
    P0s = P0 - P0 @ N0 @ P0
    return dict(loglik=ll, f_smooth=fs, P_smooth=Ps, P_lag=Pl, f0_smooth=f0, P0_smooth=0.5 * (P0s + P0s.T))
