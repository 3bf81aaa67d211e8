"""CPU fp64 oracle of the wild-bootstrap impulse-response draws (BASELINE config 5).

TEST INFRASTRUCTURE ONLY: nothing under dynamic_factor_models_amd/ may import it.

The reference has no bootstrap ("parity unpinned" for the draw loop); what it does have -- and what every
draw re-runs -- is pinned: `estimate_var!` + `fill_matrices!` (dfm_functions.ipynb:444-492) and
`impulse_response` (:793-816), restated in oracle/als_oracle.py and pinned by the notebook's Table 5.  The
draw itself is the textbook recursive-design wild bootstrap (Goncalves & Kilian 2004):
    e*_t = s_t e_t,  s_t = +-1;  y*_t = c + sum_l A_l y*_{t-l} + e*_t,  y*_t = y_t for t < p.
"""
from __future__ import annotations

import numpy as np

from . import als_oracle as ao


def var_bootstrap_irf(y, p, H, signs):
    """y [T, ns] (no NaN); signs [B, T] of +-1.  Returns (irf [B, ns, H, ns], beta [B, 1 + ns p, ns],
    point estimate dict from als_oracle.estimate_var)."""
    y = np.asarray(y, float)
    T, ns = y.shape
    v = ao.estimate_var(y, p, 1, T)
    beta = v["betahat"]
    e = np.zeros((T, ns))
    e[p:] = v["resid"][p:]
    B = signs.shape[0]
    irf = np.empty((B, ns, H, ns))
    betas = np.empty((B,) + beta.shape)
    for d in range(B):
        ys = y.copy()
        for t in range(p, T):
            x = np.concatenate([[1.0]] + [ys[t - 1 - l] for l in range(p)])
            ys[t] = x @ beta + signs[d, t] * e[t]
        vd = ao.estimate_var(ys, p, 1, T)
        betas[d] = vd["betahat"]
        irf[d] = ao.impulse_response(vd["M"], vd["Q"], vd["G"], range(ns), H)
    return irf, betas, v
