#!/usr/bin/env python
"""Golden vectors of the two estimators added in round 3, computed by the CPU oracle (NOT by the reference: it has no parametric
estimator and its observed-factor path does not run -- see oracle/obs_oracle.py).  They pin the ORACLE against silent drift
(tests/test_oracle_round3_goldens.py, CPU) and give the HIP path a committed target (tests/test_gpu_round3_goldens.py).
  1. observed factors: 3 EM iterations, N = 24, T = 60, r_o = 2, r_u = 3, 8 % missing
  2. EM with missing cells at a wide state: 2 iterations, N = 40, T = 70, r = 12 (Rp = 16), 10 % missing
Run from the repo root:  python tests/golden/make_round3_goldens.py   -> tests/golden/round3_goldens.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import kalman_oracle as ko   # noqa: E402
from oracle import obs_oracle as oo      # noqa: E402

KEYS = ("Lam", "R", "A", "Q", "mu0", "P0")


def case_obs():
    x, G, p = oo.synth_obs(31, 24, 60, 3, 2, missing=0.08)
    new, path, _ = oo.em_obs(x, G, p, max_iter=3, tol=0.0)
    out = {"obs_x": x, "obs_G": G, "obs_path": path}
    for k in KEYS:
        out["obs_start_" + k] = np.asarray(p[k], float)
        out["obs_end_" + k] = np.asarray(new[k], float)
    return out


def case_miss():
    x, _ = ko.synth_replicate(77, 40, 70, 12, missing=0.1)
    p0, _ = ko.pca_init(np.nan_to_num(x), 12)
    new, path, o = ko.em(x, {k: p0[k] for k in KEYS}, max_iter=2, tol=0.0)
    out = {"miss_x": x, "miss_path": path, "miss_f_smooth": o["f_smooth"]}
    for k in KEYS:
        out["miss_start_" + k] = np.asarray(p0[k], float)
        out["miss_end_" + k] = np.asarray(new[k], float)
    return out


if __name__ == "__main__":
    d = {}
    d.update(case_obs())
    d.update(case_miss())
    path = os.path.join(ROOT, "tests", "golden", "round3_goldens.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, {k: v.shape for k, v in d.items() if k.endswith("path")})
