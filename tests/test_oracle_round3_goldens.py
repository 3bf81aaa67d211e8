"""CPU: the oracle against the committed golden vectors of the round-3 estimators (tests/golden/round3_goldens.npz, made by
tests/golden/make_round3_goldens.py).  The vectors were computed by this same oracle: the test catches DRIFT of the checker, it
is not an independent pin (the reference has nothing to pin these paths with -- DESIGN.md section 0)."""
import os

import numpy as np

from oracle import kalman_oracle as ko
from oracle import obs_oracle as oo

G_ = np.load(os.path.join(os.path.dirname(__file__), "golden", "round3_goldens.npz"))
KEYS = ("Lam", "R", "A", "Q", "mu0", "P0")


def test_observed_factor_em_reproduces_the_golden_path():
    p = {k: G_["obs_start_" + k] for k in KEYS}
    new, path, _ = oo.em_obs(G_["obs_x"], G_["obs_G"], p, max_iter=3, tol=0.0)
    np.testing.assert_allclose(path, G_["obs_path"], rtol=1e-12)
    for k in KEYS:
        np.testing.assert_allclose(new[k], G_["obs_end_" + k], rtol=1e-10, atol=1e-12)
    assert np.all(np.diff(path) > 0)


def test_missing_cell_em_at_a_wide_state_reproduces_the_golden_path():
    p = {k: G_["miss_start_" + k] for k in KEYS}
    new, path, out = ko.em(G_["miss_x"], p, max_iter=2, tol=0.0)
    np.testing.assert_allclose(path, G_["miss_path"], rtol=1e-12)
    np.testing.assert_allclose(out["f_smooth"], G_["miss_f_smooth"], rtol=1e-9, atol=1e-11)
    for k in KEYS:
        np.testing.assert_allclose(new[k], G_["miss_end_" + k], rtol=1e-10, atol=1e-12)
