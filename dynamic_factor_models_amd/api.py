"""Host-side mirror of the reference's Julia interface for the parametric path.

The reference (QuantEcon/dynamic_factor_models, dfm_functions.ipynb) is Julia; Julia is not installed in
the build image, so the layer that a Julia user would call -- `DFMModel(...)`, `estimate!(m, Parametric())`
-- is mirrored here in Python with the same names, argument meaning and error behaviour, on top of the
same C-ABI (include/dfm_hip.h) the Julia shim (julia/dfm_hip.jl) binds with `ccall`.  Index arguments
(`initperiod`, `lastperiod`) are 1-based and inclusive exactly as in the reference.

What runs where: everything O(B T N) -- PCA initialisation, Kalman filter, RTS smoother, EM -- runs in the
hand-written HIP kernels of libdfmhip.so.  The host does what the reference's Julia host code does around
its numerical kernels: slicing the estimation window, `standardize_data`, the complete-case column filter,
and copying results into the model object.  There is no CPU implementation of the hot path in this
package: without a HIP device `estimate` raises.

Reference objects mirrored (file:line of dfm_functions.ipynb):
  EstimationMethod / NonParametric / Parametric   :21-23
  VARModel                                         :43-57, ctor :424-435
  FactorEstimateStats                              :66-73
  DFMModel                                         :89-111, ctor and its three `error(...)` checks :120-146
  standardize_data                                 :501-509
  drop_missing_col                                 :167-170
  estimate!(m, ::NonParametric)                    :530-543  (the only method the reference implements;
                                                              `Parametric` is the declared-but-empty slot)
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np


# ----------------------------------------------------------------------------- dispatch tags (:21-23)
class EstimationMethod:
    pass


class NonParametric(EstimationMethod):
    pass


class Parametric(EstimationMethod):
    pass


# ----------------------------------------------------------------------------- containers
@dataclass
class FactorEstimateStats:          # dfm_functions.ipynb:66-73
    T: int
    ns: int
    nobs: Optional[int] = None
    tss: Optional[float] = None
    ssr: Optional[float] = None
    R2: np.ndarray = field(default_factory=lambda: np.empty(0))


@dataclass
class VARModel:                     # dfm_functions.ipynb:43-57 (y_t = Q z_t, z_t = M z_{t-1} + G u_t, :30-34)
    y: np.ndarray
    nlag: int
    withconst: bool
    initperiod: int
    lastperiod: int
    T: int
    ns: int
    resid: np.ndarray
    betahat: np.ndarray
    M: np.ndarray
    Q: np.ndarray
    G: np.ndarray
    seps: np.ndarray


def _var_model(y: np.ndarray, nlag: int = 1, withconst: bool = True, initperiod: int = 1,
               lastperiod: Optional[int] = None) -> VARModel:
    """VARModel(y, nlag; withconst, initperiod, lastperiod) -- dfm_functions.ipynb:424-435."""
    T, ns = y.shape
    lastperiod = T if lastperiod is None else lastperiod
    k = ns * nlag
    return VARModel(y=y, nlag=nlag, withconst=withconst, initperiod=initperiod, lastperiod=lastperiod, T=T, ns=ns,
                    resid=np.full((T, ns), np.nan), betahat=np.full((k + (1 if withconst else 0), ns), np.nan),
                    M=np.zeros((k, k)), Q=np.zeros((ns, k)), G=np.zeros((k, ns)), seps=np.full((ns, ns), np.nan))


class DFMModel:
    """DFMModel(data, inclcode, nt_min_factor_estimation, nt_min_factorloading_estimation, initperiod,
    lastperiod, nfac_o, nfac_u, tol, n_uarlag, n_factorlag) -- dfm_functions.ipynb:89-146.

    `data` is T x ns with NaN for the reference's `missing`.  `factor` and `factor_var_model.y` alias the
    same array, as in the reference (:80, :138).  The reference's field `lambda` is `lambda_` here."""

    def __init__(self, data, inclcode, nt_min_factor_estimation: int, nt_min_factorloading_estimation: int,
                 initperiod: int, lastperiod: int, nfac_o: int, nfac_u: int, tol: float, n_uarlag: int,
                 n_factorlag: int):
        data = np.asarray(data, dtype=np.float64)
        inclcode = np.asarray(inclcode).astype(int).ravel()
        if data.ndim != 2 or data.shape[1] != inclcode.shape[0]:
            raise ValueError("length of inclcode must equal to number of data series")          # :124
        if not (initperiod < lastperiod):
            raise ValueError("initperiod must be smaller than lastperiod")                      # :125
        if not (n_uarlag > 0 and n_factorlag > 0):
            raise ValueError("n_uarlag and n_factorlag must be positive")                       # :126
        T, ns = data.shape
        if not (1 <= initperiod and lastperiod <= T):
            raise ValueError("estimation window must lie inside the data (1 <= initperiod, lastperiod <= T)")
        self.data = data
        self.inclcode = inclcode
        self.T, self.ns = T, ns
        self.nt_min_factor_estimation = int(nt_min_factor_estimation)
        self.nt_min_factorloading_estimation = int(nt_min_factorloading_estimation)
        self.initperiod, self.lastperiod = int(initperiod), int(lastperiod)
        self.nfac_o, self.nfac_u = int(nfac_o), int(nfac_u)
        self.nfac_t = self.nfac_o + self.nfac_u
        self.tol = float(tol)
        n_incl = int(np.count_nonzero(inclcode == 1))
        self.fes = FactorEstimateStats(self.lastperiod - self.initperiod + 1, n_incl, None, None, None,
                                       np.full(n_incl, np.nan))
        self.factor = np.full((T, self.nfac_t), np.nan)
        self.lambda_ = np.full((ns, self.nfac_t), np.nan)
        self.uar_coef = np.full((ns, n_uarlag), np.nan)
        self.uar_ser = np.full(ns, np.nan)
        self.n_uarlag, self.n_factorlag = int(n_uarlag), int(n_factorlag)
        self.factor_var_model = _var_model(self.factor, self.n_factorlag, True, self.initperiod, self.lastperiod)
        self.r2 = np.full(ns, np.nan)
        # results of the parametric path that have no field in the reference struct
        self.loglik_path: Optional[np.ndarray] = None
        self.em_iters: Optional[int] = None
        self.em_params: Optional[dict] = None


# ----------------------------------------------------------------------------- host helpers
def standardize_data(x: np.ndarray):
    """dfm_functions.ipynb:501-509: per-series mean and *population* s.d. over the observed cells;
    returns ((x - mean) / sd, sd [1 x ns])."""
    x = np.asarray(x, dtype=np.float64)
    n = np.count_nonzero(~np.isnan(x), axis=0)
    with np.errstate(invalid="ignore", divide="ignore"):      # a column without observations stays all-NaN
        mu = np.nansum(x, axis=0) / n
        d = np.where(np.isnan(x), 0.0, x - mu)
        sd = np.sqrt((d * d).sum(axis=0) / n)
        return (x - mu) / sd, sd[None, :]


def drop_missing_col(A: np.ndarray):
    """dfm_functions.ipynb:167-170: keep the columns with no missing cell; returns (sub-matrix, mask)."""
    keep = ~np.isnan(A).any(axis=0)
    return A[:, keep], keep


# ----------------------------------------------------------------------------- estimate!(m, ::Parametric)
def estimate(m: DFMModel, method: EstimationMethod = None, *, max_em_iter: int = 50, tol_em: float = 1e-6,
             factor_lags: Optional[int] = None, ctx=None, lam_constr_f=None, lam_constr_fl=None,
             nrep: int = 0, seed: int = 20160415, ngpu: int = 1):
    """`estimate!(m::DFMModel, ::Parametric; max_em_iter, tol_em)`: PCA-initialised EM for the exact
    Gaussian state-space DFM  x_t = Lam f_t + e_t,  f_t = A f_{t-1} + eta_t  on the standardised
    estimation window (rows initperiod..lastperiod, series with inclcode == 1), fitted with the HIP
    library.  Mutates `m` in place like the reference's `estimate!` and returns the per-iteration
    log-likelihood vector (the reference's own methods return `nothing`).

      m.factor[initperiod:lastperiod, :]  <- smoothed factors E[f_t | X]
      m.lambda_[incl, :]                  <- loadings in data units (Lam_i * sd_i)
      m.uar_ser[incl]                     <- idiosyncratic s.d. in data units; m.uar_coef[incl, :] = 0
      m.factor_var_model.M / G / seps     <- A (companion block), chol(Q) lower, Q      (cf. :477-492)
      m.fes.tss / nobs / ssr / R2         <- as estimate_factor! defines them (:342-343, :366, :372-380),
                                             with the common component Lam f_t|T in place of the ALS fit
    `factor_lags` = p of the factor VAR, f_t = A_1 f_{t-1} + .. + A_p f_{t-p} + eta_t; default: the model's own
    `n_factorlag` (dfm_functions.ipynb:120-146), run in the companion form `fill_matrices!` builds (:477-492)
    through dfm_em_varp_batch (r p <= 32); then M, G, seps, betahat hold [A_1 .. A_p], chol(Q), Q.
    `nrep` > 0 (`estimate!(m, ::Parametric; ..., nrep, seed, ngpu)`): after the point estimate, `nrep`
    parametric-bootstrap replicates of the standardised window are drawn from the fitted model (the window's own missing
    pattern; NumPy generator seeded with `seed`) and re-estimated by EM in ONE batched call on `ngpu` GPUs of this node
    (dfm_em_batch_multi: replicates sharded over the GPUs, one RCCL all-gather of {loglik, active} per EM iteration);
    the replicate estimates land in `m.replicates`.  Needs factor_lags = 1.
    `NonParametric()` runs the reference's own estimator (ALS, loadings, VAR) on the HIP kernels of als.hip:
    see estimate_nonparametric below."""
    method = Parametric() if method is None else method
    if isinstance(method, NonParametric):
        return estimate_nonparametric(m, ctx=ctx, lam_constr_f=lam_constr_f, lam_constr_fl=lam_constr_fl)
    if not isinstance(method, Parametric):
        raise TypeError("method must be Parametric() or NonParametric()")
    if lam_constr_f is not None or lam_constr_fl is not None:
        raise NotImplementedError("loading constraints are not supported on the parametric path")
    if m.nfac_o != 0:
        if nrep:
            raise ValueError("bootstrap replicates (nrep > 0) are not available with observed factors")
        return _estimate_parametric_observed(m, max_em_iter, tol_em, ctx)
    r = m.nfac_u
    nlag = m.n_factorlag if factor_lags is None else int(factor_lags)
    if nlag < 1 or r * nlag > 32:
        raise ValueError("need 1 <= factor_lags and nfac_u * factor_lags <= 32 (DFM_MAX_R)")
    if nrep and nlag != 1:
        raise ValueError("bootstrap replicates (nrep > 0) need factor_lags = 1")
    incl = m.inclcode == 1
    xdata = m.data[m.initperiod - 1:m.lastperiod, :][:, incl]           # :335-336
    z, sd = standardize_data(xdata)                                     # :339
    obs = ~np.isnan(z)
    m.fes.tss = float(np.nansum(z * z))                                 # :342
    m.fes.nobs = int(obs.sum())                                         # :343
    enough = obs.sum(axis=0) >= m.nt_min_factor_estimation              # :357 (series too short are left out)
    if not enough.all():
        z = z[:, enough]
    xbal, balmask = drop_missing_col(z)                                 # :345
    T, N = z.shape
    if xbal.shape[1] < r:
        raise ValueError("fewer fully observed series than factors: cannot initialise by PCA")

    own_ctx = ctx is None
    if own_ctx:
        from .kalman import DfmContext
        ctx = DfmContext()                                              # raises without a HIP device
    try:
        p0, F0 = ctx.pca_init_batch_host(xbal[None, :, :], r)          # pca_score (:179-183) + OLS start, on the GPU
        F0 = F0[0]
        Lam = np.empty((N, r)); R = np.empty(N)
        Lam[balmask] = p0["Lam"][0]; R[balmask] = p0["R"][0]
        gap = np.nonzero(~balmask)[0]
        if gap.size:                                                    # series with gaps: complete-case OLS on F0, no
            o = ctx.ols_batch_host(F0, z[:, gap], want_resid=False)     # intercept (`ols_skipmissing`, :242-252): dfm_ols_batch
            Lam[gap] = o["beta"]
            R[gap] = o["ssr"] / np.maximum(o["nobs"], 1)
        if nlag == 1:
            start = dict(Lam=Lam[None], R=R[None], A=p0["A"], Q=p0["Q"], mu0=p0["mu0"], P0=p0["P0"])
            from ._lib import DfmError
            args = (z[None], start["Lam"], start["R"], start["A"], start["Q"], start["mu0"], start["P0"])
            kw = dict(max_iter=max_em_iter, tol=tol_em, may_have_missing=bool((~obs).any()))
            used_singular_q = False
            try:
                params, path, iters, f, P = ctx.em_batch_host(*args, **kw)
            except DfmError as err:
                # the information-form recursion inverts Q: a PCA start on fewer than 2r + 1 periods has a rank-deficient
                # VAR residual covariance.  Run the covariance-form recursion (DFM_F_SINGULAR_Q) instead of failing.
                if err.code != -5:
                    raise
                params, path, iters, f, P = ctx.em_batch_host(*args, singular_q=True, **kw)
                used_singular_q = True
        else:
            # VAR(p) start (oracle/varp_oracle.py varp_init): OLS of the PCA factors on their p lags without constant
            # (dfm_ols_batch), Q = residual covariance / (T - p), z_0 ~ N(0, second moment of the stacked lags)
            if T - nlag <= r * nlag:
                raise ValueError("too few periods for a VAR(factor_lags) start")
            Z = np.hstack([F0[nlag - 1 - l:T - l] for l in range(nlag)])
            o = ctx.ols_batch_host(Z[:-1], F0[nlag:], want_resid=True)  # one regression per factor
            Avar = o["beta"].copy()                                     # [A_1 .. A_p]  (r, r p)
            e = o["resid"]
            Qv = e.T @ e / (T - nlag); Qv = 0.5 * (Qv + Qv.T)
            P0v = Z.T @ Z / Z.shape[0]; P0v = 0.5 * (P0v + P0v.T)
            from ._lib import DfmError
            vargs = (z[None], Lam[None], R[None], Avar[None], Qv[None], np.zeros((1, r * nlag)), P0v[None])
            vkw = dict(max_iter=max_em_iter, tol=tol_em, may_have_missing=bool((~obs).any()))
            try:
                params, path, iters, f, P = ctx.em_varp_batch_host(*vargs, **vkw)
            except DfmError as err:                     # r = 4: recursion_comp.hip inverts the r x r block Q (as above)
                if err.code != -5:
                    raise
                params, path, iters, f, P = ctx.em_varp_batch_host(*vargs, singular_q=True, **vkw)
            params = dict(params)
            params["A"] = params["Avar"]
    finally:
        if own_ctx:
            ctx.close()
    k = int(iters[0])
    f = f[0]
    Lam, R, A, Q = params["Lam"][0], params["R"][0], params["A"][0], params["Q"][0]
    m.loglik_path = path[0, :k].copy()
    m.em_iters = k
    m.em_params = {kk: v[0].copy() for kk, v in params.items()}
    m.factor[m.initperiod - 1:m.lastperiod, :] = f                      # in place: aliases factor_var_model.y (:371, :80)
    cols = np.nonzero(incl)[0][enough] if not enough.all() else np.nonzero(incl)[0]
    sdv = sd[0][enough] if not enough.all() else sd[0]
    m.lambda_[cols, :] = Lam * sdv[:, None]
    m.uar_ser[cols] = np.sqrt(R) * sdv
    m.uar_coef[cols, :] = 0.0
    common = f @ Lam.T
    e = np.where(np.isnan(z), 0.0, z - common)
    m.fes.ssr = float((e * e).sum())                                    # :366
    zc = z - np.nanmean(z, axis=0)
    R2 = 1.0 - (e * e).sum(axis=0) / np.nansum(zc * zc, axis=0)         # compute_r2 (:565-569)
    m.fes.R2 = np.full(m.fes.ns, np.nan)
    m.fes.R2[np.nonzero(enough)[0]] = R2
    m.r2[cols] = R2
    var = m.factor_var_model                                            # fill_matrices! (:477-492) for VAR(1)
    var.M[:] = 0.0; var.Q[:] = 0.0; var.G[:] = 0.0
    ka = min(A.shape[1], var.M.shape[1])                                # [A_1 .. A_p] into the model's companion (:484-486)
    var.M[:r, :ka] = A[:, :ka]
    if var.nlag > 1:
        var.M[r:, :-r] = np.eye(r * (var.nlag - 1))
    var.Q[:, :r] = np.eye(r)
    var.seps[:] = Q
    var.G[:r, :r] = _psd_sqrt(Q)            # lower Cholesky factor (:489); a rank-deficient Q has only a symmetric root
    var.betahat[:] = 0.0
    c0 = 1 if var.withconst else 0
    var.betahat[c0:c0 + ka, :] = A[:, :ka].T
    if nrep:
        m.replicates = _bootstrap_replicates(z, params, int(nrep), int(seed), int(ngpu), max_em_iter, tol_em,
                                             singular_q=(nlag == 1 and used_singular_q))
    return m.loglik_path


def _estimate_parametric_observed(m: DFMModel, max_em_iter, tol_em, ctx):
    """`estimate(m, Parametric())` with OBSERVED factors (nfac_o > 0).  The reference's estimator is
    non-functional there (dfm_functions.ipynb:358-359, :371; App. D 7), so the semantics are those its data layout implies
    (include/dfm_hip.h, oracle/obs_oracle.py): the caller has put the observed factors g_t into the FIRST nfac_o columns of
    `m.factor` (rows initperiod..lastperiod, no gaps); they enter the measurement equation as known regressors,
        x_it = lam_o,i' g_t + lam_u,i' f_t + e_it,
    and only f_t (nfac_u columns, VAR(1)) is latent.  Start: per-series OLS on g (dfm_ols_batch), PCA + OLS start of the
    residual panel (dfm_pca_init_batch); EM: dfm_em_obs_batch.  Afterwards `m.factor[:, nfac_o:]` holds E[f_t | X],
    `m.lambda_` the nfac_t loadings in data units, and the factor VAR of ALL nfac_t factors -- the reference's own second
    stage -- is `estimate_var(m.factor_var_model)` (dfm_functions.ipynb:444-492), run here on the GPU as well."""
    ro, ru = m.nfac_o, m.nfac_u
    incl = m.inclcode == 1
    w0, w1 = m.initperiod - 1, m.lastperiod
    G = np.array(m.factor[w0:w1, :ro], float)
    if np.isnan(G).any():
        raise ValueError("observed factors: fill m.factor[initperiod:lastperiod, :nfac_o] (no gaps) before estimate()")
    z, sd = standardize_data(m.data[w0:w1, :][:, incl])                 # :335-339
    obs = ~np.isnan(z)
    m.fes.tss = float(np.nansum(z * z)); m.fes.nobs = int(obs.sum())    # :342-343
    enough = obs.sum(axis=0) >= m.nt_min_factor_estimation              # :357
    if not enough.all():
        z = z[:, enough]
    T, N = z.shape
    ctx, own = _own(ctx)
    try:
        og = ctx.ols_batch_host(G, z, want_resid=True)                  # series on the observed factors, complete cases
        Lam_o = og["beta"]
        res = og["resid"]                                               # [T, N], NaN where the cell is missing
        rbal, balmask = drop_missing_col(res)
        if rbal.shape[1] < ru:
            raise ValueError("fewer fully observed series than unobserved factors: cannot initialise by PCA")
        p0, F0 = ctx.pca_init_batch_host(rbal[None, :, :], ru)
        F0 = F0[0]
        Lam_u = np.empty((N, ru)); R = np.empty(N)
        Lam_u[balmask] = p0["Lam"][0]; R[balmask] = p0["R"][0]
        gap = np.nonzero(~balmask)[0]
        if gap.size:
            o = ctx.ols_batch_host(F0, res[:, gap], want_resid=False)
            Lam_u[gap] = o["beta"]; R[gap] = o["ssr"] / np.maximum(o["nobs"], 1)
        Lam = np.hstack([Lam_o, Lam_u])
        params, path, iters, f, P = ctx.em_obs_batch_host(z[None], G[None], Lam[None], R[None], p0["A"], p0["Q"], p0["mu0"], p0["P0"],
                                                          max_iter=max_em_iter, tol=tol_em, may_have_missing=bool((~obs).any()))
        k = int(iters[0]); f = f[0]
        Lam, R = params["Lam"][0], params["R"][0]
        m.loglik_path = path[0, :k].copy(); m.em_iters = k
        m.em_params = {kk: v[0].copy() for kk, v in params.items()}
        m.factor[w0:w1, ro:] = f                                        # in place (aliases factor_var_model.y, :80)
        cols = np.nonzero(incl)[0][enough] if not enough.all() else np.nonzero(incl)[0]
        sdv = sd[0][enough] if not enough.all() else sd[0]
        m.lambda_[cols, :] = Lam * sdv[:, None]
        m.uar_ser[cols] = np.sqrt(R) * sdv
        m.uar_coef[cols, :] = 0.0
        e = np.where(np.isnan(z), 0.0, z - np.hstack([G, f]) @ Lam.T)
        m.fes.ssr = float((e * e).sum())
        zc = z - np.nanmean(z, axis=0)
        R2 = 1.0 - (e * e).sum(axis=0) / np.nansum(zc * zc, axis=0)
        m.fes.R2 = np.full(m.fes.ns, np.nan); m.fes.R2[np.nonzero(enough)[0]] = R2
        m.r2[cols] = R2
        estimate_var(m.factor_var_model, ctx=ctx)                       # VAR of (g, f) jointly: the reference's second stage
    finally:
        if own:
            ctx.close()
    return m.loglik_path


def _psd_sqrt(S):
    """A square root L (L L' = S) of a symmetric positive SEMI-definite matrix: Cholesky when it exists, else the symmetric
    eigen square root (a fit that needed the covariance-form recursion has a rank-deficient Q)."""
    S = 0.5 * (S + S.T)
    try:
        return np.linalg.cholesky(S)
    except np.linalg.LinAlgError:
        w, V = np.linalg.eigh(S)
        return (V * np.sqrt(np.maximum(w, 0.0))) @ V.T


def _bootstrap_replicates(z, params, nrep, seed, ngpu, max_em_iter, tol_em, singular_q=False):
    """`nrep` parametric-bootstrap panels from the fitted model on the standardised window z (NaN where z is NaN),
    re-estimated from the point estimate in one dfm_em_batch_multi call (julia/dfm_hip.jl estimate!: same steps)."""
    from .kalman import DfmContext
    Lam, R, A, Q, mu0, P0 = (params[k][0] for k in ("Lam", "R", "A", "Q", "mu0", "P0"))
    T, N = z.shape
    r = Lam.shape[1]
    rng = np.random.default_rng(seed)
    LQ = _psd_sqrt(Q)                      # (never raises after the point estimate has been written into the model)
    LS = _psd_sqrt(P0)
    sq = np.sqrt(R)
    panels = np.empty((nrep, T, N))
    for b in range(nrep):
        f = mu0 + LS @ rng.standard_normal(r)
        for t in range(T):
            f = A @ f + LQ @ rng.standard_normal(r)
            panels[b, t] = Lam @ f + sq * rng.standard_normal(N)
    panels[:, np.isnan(z)] = np.nan
    rep = lambda a: np.repeat(a[None], nrep, axis=0)
    new, path, iters, _, _, ran = DfmContext.em_batch_multi_host(ngpu, panels, rep(Lam), rep(R), rep(A), rep(Q), rep(mu0),
                                                                 rep(P0), max_iter=max_em_iter, tol=tol_em,
                                                                 singular_q=singular_q)
    return dict(params=new, loglik_path=path, iters=iters, iterations=ran, panels=panels)


# ============================================================================= the NON-parametric path
# `estimate!(m, ::NonParametric)` (dfm_functions.ipynb:530-543) = estimate_factor! -> estimate_factor_loading!
# -> estimate_var!, with every regression run by the batched HIP kernels of als.hip (dfm_als_batch /
# dfm_ols_batch) and the PCA start by pca.hip.  The host code below is what the reference's Julia host code is:
# slicing, standardising, building lag matrices, copying results into the model object.
def _lagmat(X: np.ndarray, lags) -> np.ndarray:
    """dfm_functions.ipynb:295-303."""
    X = X.reshape(X.shape[0], -1)
    T, nc = X.shape
    lags = list(lags)
    out = np.full((T, nc * len(lags)), np.nan)
    for k, lag in enumerate(lags):
        out[lag:, nc * k: nc * (k + 1)] = X[: T - lag]
    return out


def _own(ctx):
    if ctx is not None:
        return ctx, False
    from .kalman import DfmContext
    return DfmContext(), True                                           # raises without a HIP device


def pca_start(ctx, z: np.ndarray, r: int) -> np.ndarray:
    """`pca_score` (dfm_functions.ipynb:179-183) of the columns of z without a missing cell (:345-348)."""
    xbal, _ = drop_missing_col(z)
    if xbal.shape[1] < r:
        raise ValueError("fewer fully observed series than factors: cannot initialise by PCA")
    _, F0 = ctx.pca_init_batch_host(xbal[None, :, :], r)
    return F0[0]


def estimate_factor(m: DFMModel, max_iter: int = 100000000, computeR2: bool = True, *, lam_constr=None, ctx=None):
    """`estimate_factor!(m, max_iter, computeR2)` -- dfm_functions.ipynb:328-382 (nfac_o = 0, no constraint)."""
    if lam_constr is not None:
        raise NotImplementedError("loading constraints are not supported on the HIP path")
    if m.nfac_o != 0:
        return _estimate_factor_observed(m, max_iter, computeR2, ctx)
    r = m.nfac_u
    xdata = m.data[m.initperiod - 1:m.lastperiod, :][:, m.inclcode == 1]   # :335-336
    z, _ = standardize_data(xdata)                                         # :339
    m.fes.tss = float(np.nansum(z * z))                                    # :342
    m.fes.nobs = int((~np.isnan(z)).sum())                                 # :343
    ctx, own = _own(ctx)
    try:
        F0 = pca_start(ctx, z, r)                                          # :345-348
        o = ctx.als_batch_host(z, F0[None], nt_min=m.nt_min_factor_estimation, max_iter=max_iter, tol=m.tol,
                               want_R2=computeR2)                          # :352-370 (+ :372-380)
    finally:
        if own:
            ctx.close()
    m.factor[m.initperiod - 1:m.lastperiod, :] = o["F"][0]                 # :371
    m.fes.ssr = float(o["ssr"][0])                                         # :366
    if computeR2:
        m.fes.R2 = o["R2"][0].copy()
    m.als_iters = int(o["iters"][0])
    return None


def _estimate_factor_observed(m: DFMModel, max_iter, computeR2, ctx):
    """`estimate_factor!` with OBSERVED factors, as the reference's loop is evidently meant (dfm_functions.ipynb:352-371: the
    factor step already regresses on `lambda[:, nfac_o+1:end]` only, :364; what is broken is that the loading step regresses
    on the nfac_u estimated columns alone, :358-359, and :371 writes nfac_u columns into nfac_t -- App. D 7).  g_t = the first
    nfac_o columns of `m.factor` (filled by the caller).  Per sweep TWO batched complete-case regressions on the GPU
    (dfm_ols_batch): every series on [g, f] (N problems), then every period's x_t - Lam_o g_t on Lam_u (T problems sharing
    the regressors) -- `ols_skipmissing(.., Unbalanced())` of :364; the stopping rule is the reference's (:366-368)."""
    ro, ru = m.nfac_o, m.nfac_u
    w0, w1 = m.initperiod - 1, m.lastperiod
    G = np.array(m.factor[w0:w1, :ro], float)
    if np.isnan(G).any():
        raise ValueError("observed factors: fill m.factor[initperiod:lastperiod, :nfac_o] (no gaps) before estimate_factor()")
    z, _ = standardize_data(m.data[w0:w1, :][:, m.inclcode == 1])          # :335-339
    T, N = z.shape
    m.fes.tss = float(np.nansum(z * z)); m.fes.nobs = int((~np.isnan(z)).sum())
    ctx, own = _own(ctx)
    try:
        res = ctx.ols_batch_host(G, z, want_resid=True)["resid"]           # start: PCA of what g does not explain
        F = pca_start(ctx, res, ru)
        ssr, it = 0.0, 0
        for it in range(1, int(min(max_iter, 10 ** 8)) + 1):
            ssr_old = ssr
            lam = ctx.ols_batch_host(np.hstack([G, F]), z, nt_min=m.nt_min_factor_estimation, want_resid=False)["beta"]   # :355-361
            y = z - G @ lam[:, :ro].T                                      # NaN rows of lam (short series) drop out below
            o = ctx.ols_batch_host(lam[:, ro:], y.T, want_resid=False)     # :364, one problem per period
            F = o["beta"]
            ssr = float(o["ssr"].sum())                                    # :366
            if not abs(ssr_old - ssr) >= m.tol * m.fes.T * m.fes.ns:       # :367-368
                break
        m.factor[w0:w1, ro:] = F                                           # :371 (the observed columns stay the caller's)
        m.fes.ssr = ssr
        m.als_iters = it
        if computeR2:                                                      # :372-380
            o = ctx.ols_batch_host(np.hstack([G, F]), z, nt_min=m.nt_min_factor_estimation, want_resid=False)
            m.fes.R2 = np.where(np.isnan(o["beta"][:, 0]), np.nan, 1.0 - o["ssr"] / o["tss"])
    finally:
        if own:
            ctx.close()
    return None


def estimate_factor_loading(m: DFMModel, *, lam_constr=None, ctx=None):
    """`estimate_factor_loading!(m)` -- dfm_functions.ipynb:391-415: every series (raw units) on [F 1] over the
    complete cases of the window, r2, then an AR(n_uarlag) of the residuals."""
    if lam_constr is not None:
        raise NotImplementedError("loading constraints are not supported on the HIP path")
    F = m.factor[m.initperiod - 1:m.lastperiod, :]
    Y = m.data[m.initperiod - 1:m.lastperiod, :]
    T, r = F.shape
    X = np.column_stack([F, np.ones(T)])
    ctx, own = _own(ctx)
    try:
        o = ctx.ols_batch_host(X, Y, nt_min=m.nt_min_factorloading_estimation)
        ok = ~np.isnan(o["beta"][:, 0])
        r2 = 1.0 - o["ssr"] / o["tss"]
        m.lambda_[ok, :] = o["beta"][ok, :r]
        m.r2[ok] = r2[ok]
        # AR(n) of the residuals with the gaps closed up (the reference hands `uar` the residual VECTOR of the
        # complete cases, :404-407); one regression problem per series, own lag matrix each
        nlag = m.n_uarlag
        idx = np.nonzero(ok & (r2 < 0.9999))[0]
        if idx.size:
            U = np.full((T, idx.size), np.nan)
            XL = np.full((idx.size, T, nlag), np.nan)
            nu = np.zeros(idx.size, dtype=int)
            for q, i in enumerate(idx):
                u = o["resid"][:, i]
                u = u[~np.isnan(u)]
                nu[q] = u.size
                U[:u.size, q] = u
                XL[q, :u.size] = _lagmat(u, range(1, nlag + 1))
            a = ctx.ols_batch_host(XL, U, nt_min=0, want_resid=False)
            m.uar_coef[idx, :] = a["beta"]
            m.uar_ser[idx] = np.sqrt(a["ssr"] / (nu - nlag))                # :310
        hi = np.nonzero(ok & ~(r2 < 0.9999))[0]
        m.uar_coef[hi, :] = 0.0
        m.uar_ser[hi] = 0.0
    finally:
        if own:
            ctx.close()
    return None


def estimate_var(varm: VARModel, compute_matrices: bool = True, *, ctx=None):
    """`estimate_var!(varm, compute_matrices)` + `fill_matrices!` -- dfm_functions.ipynb:444-492."""
    yr = varm.y[varm.initperiod - 1:varm.lastperiod, :]
    T, ns = yr.shape
    x = _lagmat(yr, range(1, varm.nlag + 1))
    if varm.withconst:
        x = np.column_stack([np.ones(T), x])
    rows_ok = ~np.isnan(x).any(axis=1) & ~np.isnan(yr).any(axis=1)         # the reference drops rows jointly (:455)
    Y = np.where(rows_ok[:, None], yr, np.nan)
    ctx, own = _own(ctx)
    try:
        o = ctx.ols_batch_host(x, Y, nt_min=0)
    finally:
        if own:
            ctx.close()
    K = x.shape[1]
    varm.betahat = o["beta"].T.copy()
    e = o["resid"][rows_ok]
    T_used = int(rows_ok.sum())
    varm.seps = e.T @ e / (T_used - K)
    varm.resid[:] = np.nan
    varm.resid[varm.initperiod - 1 + np.nonzero(rows_ok)[0]] = e
    if compute_matrices:
        _fill_matrices(varm)
    return None


def _fill_matrices(varm: VARModel):
    """`fill_matrices!` -- dfm_functions.ipynb:477-492: companion M, selection Q, G = lower Cholesky factor of seps."""
    ns = varm.seps.shape[0]
    b = varm.betahat[1:].T if varm.withconst else varm.betahat.T
    k = ns * varm.nlag
    varm.M = np.zeros((k, k)); varm.Q = np.zeros((ns, k)); varm.G = np.zeros((k, ns))
    varm.M[:ns] = b
    if k > ns:
        varm.M[ns:, :-ns] = np.eye(k - ns)
    varm.Q[:, :ns] = np.eye(ns)
    varm.G[:ns] = np.linalg.cholesky(varm.seps)


def estimate_nonparametric(m: DFMModel, *, ctx=None, lam_constr_f=None, lam_constr_fl=None):
    """`estimate!(m, NonParametric())` -- dfm_functions.ipynb:530-543."""
    if lam_constr_f is not None or lam_constr_fl is not None:
        raise NotImplementedError("loading constraints are not supported on the HIP path")
    ctx, own = _own(ctx)
    try:
        estimate_factor(m, lam_constr=lam_constr_f, ctx=ctx)
        estimate_factor_loading(m, lam_constr=lam_constr_fl, ctx=ctx)
        m.factor_var_model.y = m.factor
        estimate_var(m.factor_var_model, ctx=ctx)
    finally:
        if own:
            ctx.close()
    return None


def bai_ng_criterion(ssr: float, nobs: int, T: int, r: int) -> float:
    """dfm_functions.ipynb:648-654 (ICp2 with nbar = nobs / T)."""
    nbar = nobs / T
    g = np.log(min(nbar, T)) * (nbar + T) / nobs
    return float(np.log(ssr / nobs) + r * g)


def estimate_factor_numbers(m: DFMModel, nfacs, *, ctx=None, with_aw: bool = False):
    """`estimate_factor_numbers(m, nfacs)` -- dfm_functions.ipynb:698-725: the static-factor runs for every r in
    `nfacs` go through ONE dfm_als_batch call (shared panel, r_each); Bai-Ng ICp2 per r.  Returns
    dict(bn_icp, ssr_static, tss, nobs, T, iters, factors).

    with_aw: also the `amengual_watson_test` (:734-768) the reference runs inside every static run (:716-717): per static
    count i one dfm_ols_batch (every series on [1, lags of the i factors]) and one PCA start of the residual window,
    then ALL the dynamic runs (k = 1..i for every i) in ONE dfm_als_batch call, each run on its own residual window --
    adds aw_icp / ssr_dynamic [max r, n runs] (NaN above the diagonal, the reference's `missing`).  julia/dfm_hip.jl
    estimate_factor_numbers_hip is the same sequence of calls."""
    nfacs = [int(k) for k in nfacs]
    rmax = max(nfacs)
    xdata = m.data[m.initperiod - 1:m.lastperiod, :][:, m.inclcode == 1]
    z, _ = standardize_data(xdata)
    T = z.shape[0]
    tss = float(np.nansum(z * z)); nobs = int((~np.isnan(z)).sum())
    ctx, own = _own(ctx)
    try:
        F0 = pca_start(ctx, z, rmax)                       # scores are nested: run r starts from the first r columns
        o = ctx.als_batch_host(z, np.repeat(F0[None], len(nfacs), axis=0), r_each=nfacs,
                               nt_min=m.nt_min_factor_estimation, tol=m.tol)
        out = dict(bn_icp=np.array([bai_ng_criterion(s, nobs, T, k) for s, k in zip(o["ssr"], nfacs)]),
                   ssr_static=o["ssr"].copy(), tss=tss, nobs=nobs, T=T, iters=o["iters"].copy(), factors=o["F"])
        if with_aw:
            est = m.data[:, m.inclcode == 1]
            T_all = est.shape[0]
            nlag = m.factor_var_model.nlag
            init, last = m.initperiod + 4, m.lastperiod                       # :761 (the reference hard-codes the 4)
            Tw = last - init + 1
            zs, F0s, r_each, owner, meta = [], [], [], [], []
            for col, i in enumerate(nfacs):
                fac = np.full((T_all, i), np.nan)
                fac[m.initperiod - 1:m.lastperiod] = o["F"][col][:, :i]
                x = np.column_stack([np.ones(T_all), _lagmat(fac, range(1, nlag + 1))])
                res = ctx.ols_batch_host(x, est, nt_min=x.shape[1] + m.nt_min_factor_estimation)["resid"]
                zi, _ = standardize_data(res[init - 1:last])
                Fi = np.zeros((Tw, rmax)); Fi[:, :i] = pca_start(ctx, zi, i)
                meta.append((int((~np.isnan(zi)).sum()), zi.shape[0]))
                for k in range(1, i + 1):
                    zs.append(zi); F0s.append(Fi); r_each.append(k); owner.append((k, col))
            a = ctx.als_batch_host(np.stack(zs), np.stack(F0s), r_each=r_each, nt_min=m.nt_min_factor_estimation, tol=m.tol)
            aw = np.full((rmax, len(nfacs)), np.nan); ssr_dyn = np.full((rmax, len(nfacs)), np.nan)
            for b, (k, col) in enumerate(owner):
                aw[k - 1, col] = bai_ng_criterion(a["ssr"][b], meta[col][0], meta[col][1], k)
                ssr_dyn[k - 1, col] = a["ssr"][b]
            out.update(aw_icp=aw, ssr_dynamic=ssr_dyn)
    finally:
        if own:
            ctx.close()
    return out


def impulse_response(varm: VARModel, shock_ids, T: int) -> np.ndarray:
    """`impulse_response(varm, shock_ids, T)` -- dfm_functions.ipynb:793-816: irf[:, t, k] = Q M^t G[:, shock_k]
    (point estimate; host arithmetic on the 16 x 16 companion, as in the reference).  The reference's three methods:
      * a vector of shock ids  -> [ny, T, len(shock_ids)]                                   (:793-799)
      * ONE shock id (a number) -> the [ny, T] matrix of that shock                         (:817-821: the reference's method
        passes an undefined `x` and six arguments to the five-argument `compute_irf_single_shock!` and cannot run; this is
        what it evidently means -- the same recursion written into a matrix; julia/dfm_hip.jl repairs it the same way)
      * "all" (the reference's `:all`) -> every column of G                                 (:822-825)
    Shock ids are 0-based here (1-based in Julia)."""
    if isinstance(shock_ids, str):
        if shock_ids != "all":
            raise ValueError("shock_ids: a shock index, a sequence of them, or 'all'")
        shock_ids = range(varm.G.shape[1])
    scalar = np.isscalar(shock_ids)
    ids = [int(s) for s in np.atleast_1d(shock_ids)]
    out = np.empty((varm.Q.shape[0], T, len(ids)))
    for k, s in enumerate(ids):
        if not 0 <= s < varm.G.shape[1]:
            raise IndexError(f"shock id {s} out of range (G has {varm.G.shape[1]} columns)")
        x = varm.G[:, s].copy()
        for t in range(T):
            out[:, t, k] = varm.Q @ x
            x = varm.M @ x
    return out[:, :, 0] if scalar else out


def bootstrap_irf_bands(varm: VARModel, H: int, ndraws: int = 10000, quantiles=(0.05, 0.16, 0.5, 0.84, 0.95),
                        seed: int = 20160415, signs=None, ctx=None, rank: int = 0, world: int = 1, gather=None):
    """Wild-bootstrap bands of the impulse responses of an estimated VARModel (BASELINE config 5; no reference
    counterpart -- the reference stops at the point estimate).  Draws, re-estimation, Cholesky, IRF recursion and
    the quantiles all run in boot.hip.  Returns dict(point [ns,H,ns], bands [len(q),ns,H,ns], draws [B,ns,H,ns]).

    Multi-GPU (one process per GPU): rank k of `world` computes the draws shard.replicate_range(ndraws, world, k)
    -- the device-drawn signs depend on the global draw index only -- and `gather` (e.g. a function wrapping
    shard.allgather_replicates) assembles the [ndraws, ...] array before the bands are taken; that all-gather is
    the path's one collective."""
    rows = np.nonzero(~np.isnan(varm.resid).any(axis=1))[0]
    if rows.size == 0:
        raise ValueError("estimate_var(varm) first")
    first = rows[0] - varm.nlag
    if first < 0 or not np.array_equal(rows, np.arange(rows[0], rows[-1] + 1)):
        raise ValueError("the VAR's estimation rows must be one contiguous block (no missing factors inside the window)")
    y = varm.y[first:rows[-1] + 1]
    resid = np.zeros_like(y)
    resid[varm.nlag:] = varm.resid[rows]
    if not varm.withconst:
        raise NotImplementedError("bootstrap_irf_bands needs a VAR with constant (the reference's default)")
    ctx, own = _own(ctx)
    try:
        from .shard import replicate_range
        lo, hi = replicate_range(int(ndraws), world, rank)
        draws = ctx.var_bootstrap_irf_host(y, varm.betahat, resid, varm.nlag, H, hi - lo, seed=seed, first_draw=lo,
                                           signs=None if signs is None else np.asarray(signs)[lo:hi])
        if world > 1:
            if gather is None:
                raise ValueError("world > 1 needs a gather function")
            draws = np.asarray(gather(draws))
        bands = ctx.quantile_bands_host(draws, np.asarray(quantiles, float))
    finally:
        if own:
            ctx.close()
    return dict(point=impulse_response(varm, range(varm.ns), H), bands=bands, draws=draws)


def _ar_model_inputs(m: DFMModel):
    """The parametric model with AR(n_uarlag) idiosyncratic terms assembled from what the reference's own estimator leaves
    in the model (`estimate!(m)`):

        x_it = c_i + lam_i' f_t + e_it,   e_it = sum_l uar_coef[i,l] e_i,t-l + eps_it,  sd(eps_it) = uar_ser[i]   (:391-415)
        f_t  = c_f + A_1 f_{t-1} + .. + A_p f_{t-p} + eta_t,  Var(eta_t) = seps      (factor_var_model, :444-492)

    in deviations from the VAR's mean (the reference does not keep c_i: it is re-derived as the mean residual of the
    loading regression).  Series used: included (inclcode == 1) with loadings, AR coefficients and uar_ser > 0.  Runs
    dfm_ks_pass_ar_batch (quasi-differenced observation equation, state (f_t .. f_{t-m+1}), m = max(p, n_uarlag + 1),
    nfac_u * m <= 32; likelihood conditional on the first n_uarlag window rows).  z_q ~ N(0, stationary covariance of
    the companion VAR) -- or the sample second moment of the stacked factor estimates when the VAR is not stable.
    Returns (inputs dict for the library, column indices used, factor mean mu_f, series intercepts c_i)."""
    if m.nfac_o != 0:
        raise NotImplementedError("observed factors (nfac_o > 0) are not supported (non-functional in the reference too)")
    var = m.factor_var_model
    r, p, q = m.nfac_u, var.nlag, m.n_uarlag
    mm = max(p, q + 1)
    k = r * mm
    if k > 32:
        raise ValueError("nfac_u * max(n_factorlag, n_uarlag + 1) must not exceed 32 (DFM_MAX_R)")
    if np.isnan(var.betahat).any() or np.isnan(var.seps).any():
        raise ValueError("factor_var_model is not estimated: run estimate(m, NonParametric()) first")
    rows = slice(m.initperiod - 1, m.lastperiod)
    F = m.factor[rows]
    use = (m.inclcode == 1) & ~np.isnan(m.lambda_).any(axis=1) & ~np.isnan(m.uar_coef).any(axis=1) & (m.uar_ser > 0)
    cols = np.nonzero(use)[0]
    if cols.size == 0:
        raise ValueError("no series with loadings and AR coefficients: run estimate(m, NonParametric()) first")
    Y = m.data[rows][:, cols]
    lam, rho, sig2 = m.lambda_[cols], m.uar_coef[cols], m.uar_ser[cols] ** 2
    c0 = 1 if var.withconst else 0
    Avar = np.ascontiguousarray(var.betahat[c0:c0 + r * p].T)            # [A_1 .. A_p]
    c_f = var.betahat[0] if var.withconst else np.zeros(r)
    Asum = sum(Avar[:, l * r:(l + 1) * r] for l in range(p))
    mu_f = np.linalg.solve(np.eye(r) - Asum, c_f)
    c_i = np.nanmean(Y - F @ lam.T, axis=0)                              # intercept of the loading regression (:396-400)
    x = Y - c_i - lam @ mu_f
    M = np.zeros((k, k)); M[:r, :r * p] = Avar
    M[r:, :k - r] = np.eye(k - r)
    Qk = np.zeros((k, k)); Qk[:r, :r] = var.seps
    if np.abs(np.linalg.eigvals(M)).max() < 0.999:
        P0 = np.linalg.solve(np.eye(k * k) - np.kron(M, M), Qk.ravel()).reshape(k, k)
    else:
        Fd = F - mu_f
        Z = np.hstack([Fd[mm - 1 - l:Fd.shape[0] - l] for l in range(mm)])
        P0 = Z.T @ Z / Z.shape[0]
    P0 = 0.5 * (P0 + P0.T) + 1e-10 * np.eye(k)
    inputs = dict(x=x, Lam=lam, sig2=sig2, rho=rho, Avar=Avar, Q=np.array(var.seps), mu0=np.zeros(k), P0=P0)
    return inputs, cols, mu_f, c_i


def smooth_factors_ar_idio(m: DFMModel, *, ctx=None):
    """Kalman-smoothed factors and Gaussian log-likelihood of the parametric model with AR(n_uarlag)
    idiosyncratic terms at the parameters the reference's own estimator leaves in the model (`_ar_model_inputs`).
    Returns dict(loglik, factor [T_all, r] (NaN outside rows initperiod + n_uarlag .. lastperiod), P [T - q, r(r+1)/2],
    series (column indices used), inputs (the arrays handed to the library, for the parity test))."""
    inputs, cols, mu_f, _ = _ar_model_inputs(m)
    r, q = m.nfac_u, m.n_uarlag
    ctx, own = _own(ctx)
    try:
        f, P, ll = ctx.ks_pass_ar_batch_host(inputs["x"][None], inputs["Lam"][None], inputs["sig2"][None], inputs["rho"][None],
                                             inputs["Avar"][None], inputs["Q"][None], inputs["mu0"][None], inputs["P0"][None])
    finally:
        if own:
            ctx.close()
    factor = np.full((m.T_all if hasattr(m, "T_all") else m.data.shape[0], r), np.nan)
    factor[m.initperiod - 1 + q:m.lastperiod] = f[0] + mu_f
    return dict(loglik=float(ll[0]), factor=factor, P=P[0], series=cols, inputs=inputs, mu_f=mu_f)


def estimate_ar_idio(m: DFMModel, *, max_em_iter: int = 20, tol_em: float = 1e-6, ctx=None):
    """JOINT estimation of the parametric model with AR(n_uarlag) idiosyncratic terms -- the
    re-estimation of `lambda`, `uar_coef`, `uar_ser` and the factor VAR that the reference's two-step estimator
    (:391-415, :444-468) never does.  Started from `estimate(m, NonParametric())` (`_ar_model_inputs`: the reference's own
    loadings, AR coefficients, innovation s.d. and VAR), iterated by ECM on the device (dfm_em_ar_batch: smoother pass of
    the quasi-differenced model, transition step, loadings | rho, rho | loadings, sig2); the series intercepts and the
    factor mean stay at their two-step values.  Mutates m IN PLACE like the reference's estimators: `lambda`, `uar_coef`,
    `uar_ser` of the series used, `factor` (smoothed, rows initperiod + n_uarlag .. lastperiod), `factor_var_model`
    (`betahat` slope rows, `seps`, `M`, `Q`, `G` through `fill_matrices`).  Returns the log-likelihood path (conditional on
    the first n_uarlag window rows; non-decreasing)."""
    inputs, cols, mu_f, _ = _ar_model_inputs(m)
    var = m.factor_var_model
    r, p, q = m.nfac_u, var.nlag, m.n_uarlag
    ctx, own = _own(ctx)
    try:
        est, path, iters, f, _ = ctx.em_ar_batch_host(inputs["x"][None], inputs["Lam"][None], inputs["sig2"][None],
                                                      inputs["rho"][None], inputs["Avar"][None], inputs["Q"][None],
                                                      inputs["mu0"][None], inputs["P0"][None], max_iter=max_em_iter, tol=tol_em)
    finally:
        if own:
            ctx.close()
    m.lambda_[cols] = est["Lam"][0]
    m.uar_coef[cols] = est["rho"][0]
    m.uar_ser[cols] = np.sqrt(est["sig2"][0])
    m.factor[m.initperiod - 1 + q:m.lastperiod] = f[0] + mu_f
    c0 = 1 if var.withconst else 0
    A = est["Avar"][0]
    var.betahat[c0:c0 + r * p] = A.T
    if var.withconst:                                            # the factor mean is held: c_f = (I - sum A_l) mu_f
        var.betahat[0] = (np.eye(r) - sum(A[:, l * r:(l + 1) * r] for l in range(p))) @ mu_f
    var.seps = est["Q"][0]
    _fill_matrices(var)
    return path[0, :int(iters[0])].copy()


def amengual_watson_test(m: DFMModel, nper: int = 4, *, ctx=None):
    """`amengual_watson_test(m, nper)` -- dfm_functions.ipynb:734-768: the number of DYNAMIC factors.  Every
    included series is regressed on [1, lags 1..p of the r estimated static factors], p = the factor VAR's
    `nlag` (:737, :741), over all rows of the data (one dfm_ols_batch call, 1 + p r <= 64 regressors); the ALS
    estimator is then run on the residual panel for k = 1..r dynamic factors over rows initperiod + 4 .. lastperiod
    (:761 -- the reference hard-codes the 4 and never reads its `nper` argument; kept for signature
    parity and ignored here too), one dfm_als_batch call, r runs on a shared panel.  `m.factor` must hold the
    static factors (estimate_factor first).  Returns (aw_icp [r], ssr [r])."""
    r = m.nfac_t
    est = m.data[:, m.inclcode == 1]
    T_all, ns = est.shape
    nlag = m.factor_var_model.nlag
    x = np.column_stack([np.ones(T_all), _lagmat(m.factor, range(1, nlag + 1))])
    ctx, own = _own(ctx)
    try:
        # the reference keeps a series when it has at least nt_min rows MORE than regressors (:744)
        o = ctx.ols_batch_host(x, est, nt_min=x.shape[1] + m.nt_min_factor_estimation)
        res = o["resid"]                                                   # NaN where a row was not used
        init, last = m.initperiod + 4, m.lastperiod
        z, _ = standardize_data(res[init - 1:last])
        T = z.shape[0]
        nobs = int((~np.isnan(z)).sum())
        F0 = pca_start(ctx, z, r)
        a = ctx.als_batch_host(z, np.repeat(F0[None], r, axis=0), r_each=np.arange(1, r + 1),
                               nt_min=m.nt_min_factor_estimation, tol=m.tol)
    finally:
        if own:
            ctx.close()
    aw = np.array([bai_ng_criterion(s, nobs, T, k + 1) for k, s in enumerate(a["ssr"])])
    return aw, a["ssr"].copy()


# ============================================================================= structural breaks
def break_tests(m: DFMModel, T_break: int, ccut: float = 0.15, q: int = 6, min_obs: int = 80, *, ctx=None):
    """Chow statistic at `T_break` and HAC QLR statistic of every series of `m.data` regressed on the estimated
    factors -- the loop of the driver's Table 4 (Stock_Watson.ipynb:1085-1098) over `compute_chow` / `compute_qlr`
    (dfm_functions.ipynb:891-902, 1019-1047), all (series x break date x bandwidth) problems in ONE dfm_chow_batch
    call.  As in the driver, a series takes part when it has at least `min_obs` observations before and after
    row `T_break` (1-based, inclusive), regressions run on the complete cases of [y X], and the break dates index
    the complete-case rows.  Returns (chow [ns], qlr [ns]) with NaN for the series left out."""
    X = m.factor
    ns = m.data.shape[1]
    chow = np.full(ns, np.nan); qlr = np.full(ns, np.nan)
    ys, Xs, who = [], [], []
    for i in range(ns):
        y = m.data[:, i]
        if (~np.isnan(y[:T_break])).sum() >= min_obs and (~np.isnan(y[T_break:])).sum() >= min_obs:
            ok = ~np.isnan(y) & ~np.isnan(X).any(axis=1)
            ys.append(y[ok]); Xs.append(X[ok]); who.append(i)
    if not who:
        return chow, qlr
    ps, pb, pq, tag = [], [], [], []
    for s, yv in enumerate(ys):
        T = len(yv)
        ps.append(s); pb.append(T_break); pq.append(q); tag.append(0)                 # the Chow test of the driver
        n1 = int(np.floor(ccut * T))
        for tb in range(n1, T - n1 + 1):                                               # compute_qlr's HAC leg
            ps.append(s); pb.append(tb); pq.append(q); tag.append(1)
    ctx, own = _own(ctx)
    try:
        stat = ctx.chow_batch_host(ys, Xs, ps, pb, pq)
    finally:
        if own:
            ctx.close()
    ps = np.asarray(ps); tag = np.asarray(tag)
    for s, i in enumerate(who):
        sel = ps == s
        chow[i] = stat[sel & (tag == 0)][0]
        qlr[i] = stat[sel & (tag == 1)].max()
    return chow, qlr
