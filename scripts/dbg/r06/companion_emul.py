"""NumPy model of csrc/recursion_comp.hip (round 6): the smoother pass of a COMPANION state s_t = (f_t, .., f_{t-m+1}) (k = r m; VAR(p)
factor dynamics, AR idiosyncratic terms by quasi-differencing: the collapsed observation may load on every block) in INFORMATION
form, as the block elimination of the block-banded posterior precision -- no k x k inversion per period, only r x r ones:

  filter    posterior of s_t: (Om_f, xi_f).  Joint precision of (f_{t+1}, s_t) after the transition f_{t+1} = Phi s_t + eta:
            [[Qi, -Qi Phi], [-Phi' Qi, M]],  M = Om_f + Phi' Qi Phi.  s_{t+1} = (f_{t+1}, a) keeps a = s_t[:k-r] and drops the oldest
            block d = s_t[k-r:]: a Schur complement on the r x r block M_dd,
            Om_p = [[Qi, -Qi Phi_a], [., M_aa]] - U M_dd^-1 U',  U = [-Qi Phi_d; M_ad],   xi_p = [0; xi_a] - U M_dd^-1 xi_d;
            then the observation adds: Om_f' = Om_p + C_{t+1}, xi_f' = xi_p + b_{t+1}.
  smoother  d | s_{t+1}, X ~ N(g + G s_{t+1}, M_dd^-1),  G = -M_dd^-1 U',  g = M_dd^-1 xi_d: the moments of s_t are those of s_{t+1}
            shifted by one block plus one new block row;  Cov(s_{t+1}, s_t | X) = [V_{t+1}[:, r:], V_{t+1} G'].
  likelihood  -2 ll = sum_t (n_t log 2 pi + ld_t + s_t) + mu0' P0^-1 mu0 + log det P0 + T log det Q + sum_t log det M_dd,t
              + log det Om_f,T - sum_t xi_d' M_dd^-1 xi_d - xi_T' Om_f,T^-1 xi_T     (the eliminations' pivots and the last marginal).
TEST INFRASTRUCTURE ONLY (tests/test_companion_model_cpu.py keeps it honest against oracle/varp_oracle.py and oracle/ar_oracle.py)."""
import numpy as np

LOG2PI = np.log(2.0 * np.pi)


def companion_pass(b, C, s, n, ld, Phi, Q, mu0, P0):
    """b [T, k], C [T, k, k] collapsed observations of periods 1..T (zero where the observation does not load); Phi [r, k] = the free rows
    of the transition, Q [r, r]; s_0 ~ N(mu0, P0).  Returns dict(loglik, f_smooth [T, k], P_smooth [T, k, k], P_lag [T, k, k], f0, P0s)."""
    T, k = b.shape
    r = Phi.shape[0]
    ka = k - r
    Qi = np.linalg.inv(Q)
    PQP = Phi.T @ Qi @ Phi
    QiPhi = Qi @ Phi
    Om = np.linalg.inv(P0); xi = Om @ mu0
    acc = float(mu0 @ Om @ mu0) + np.linalg.slogdet(P0)[1] + T * np.linalg.slogdet(Q)[1]
    Gs = np.empty((T, r, k)); gs = np.empty((T, r)); Mi = np.empty((T, r, r))
    for t in range(T):
        M = Om + PQP
        Mdd = M[ka:, ka:]
        Mdi = np.linalg.inv(Mdd)
        U = np.vstack([-QiPhi[:, ka:], M[:ka, ka:]])               # [k, r]: rows of (f', a) against d
        Op = np.zeros((k, k))
        Op[:r, :r] = Qi; Op[:r, r:] = -QiPhi[:, :ka]; Op[r:, :r] = -QiPhi[:, :ka].T; Op[r:, r:] = M[:ka, :ka]
        Op -= U @ Mdi @ U.T
        xp = np.concatenate([np.zeros(r), xi[:ka]]) - U @ Mdi @ xi[ka:]
        acc += np.linalg.slogdet(Mdd)[1] - xi[ka:] @ Mdi @ xi[ka:]
        Gs[t] = -Mdi @ U.T; gs[t] = Mdi @ xi[ka:]; Mi[t] = Mdi
        Om = Op + C[t]; Om = 0.5 * (Om + Om.T)
        xi = xp + b[t]
        acc += n[t] * LOG2PI + ld[t] + s[t]
    V = np.linalg.inv(Om); m = V @ xi
    acc += np.linalg.slogdet(Om)[1] - xi @ m
    fs = np.empty((T, k)); Ps = np.empty((T, k, k)); Pl = np.empty((T, k, k))
    for t in range(T - 1, -1, -1):
        fs[t] = m; Ps[t] = V
        G, g = Gs[t], gs[t]
        VG = V @ G.T                                               # Cov(s_t, d | X), d = the block of s_{t-1} that s_t dropped
        Pl[t] = np.hstack([V[:, r:], VG])                          # Cov(s_t, s_{t-1} | X)
        md = g + G @ m
        Vdd = G @ VG + Mi[t]
        Vn = np.empty((k, k))
        Vn[:ka, :ka] = V[r:, r:]; Vn[:ka, ka:] = VG[r:]; Vn[ka:, :ka] = VG[r:].T; Vn[ka:, ka:] = Vdd
        m = np.concatenate([m[r:], md]); V = 0.5 * (Vn + Vn.T)
    return dict(loglik=-0.5 * acc, f_smooth=fs, P_smooth=Ps, P_lag=Pl, f0_smooth=m, P0_smooth=V)
