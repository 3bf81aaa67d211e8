"""GPU: the HIP path against the COMMITTED golden vectors of the round-3 estimators (tests/golden/round3_goldens.npz) -- nothing
of oracle/ is executed here, the targets are data.  (The vectors come from the CPU oracle, tests/golden/make_round3_goldens.py;
the oracle itself is held to them by tests/test_oracle_round3_goldens.py.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G_ = np.load(os.path.join(os.path.dirname(__file__), "golden", "round3_goldens.npz"))
KEYS = ("Lam", "R", "A", "Q", "mu0", "P0")


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available()
    from dynamic_factor_models_amd import DfmContext
    c = DfmContext()
    yield c
    c.close()


def test_observed_factor_em_matches_the_golden_vectors(ctx):
    st = [G_["obs_start_" + k][None] for k in KEYS]
    new, path, its, f, P = ctx.em_obs_batch_host(G_["obs_x"][None], G_["obs_G"][None], *st, max_iter=3, tol=0.0)
    np.testing.assert_allclose(path[0], G_["obs_path"], rtol=1e-9)
    for k in KEYS:
        want = G_["obs_end_" + k]
        assert np.abs(new[k][0] - want).max() <= 1e-8 * max(1.0, np.abs(want).max()), k


def test_missing_cell_em_at_a_wide_state_matches_the_golden_vectors(ctx):
    """Rp = 16 with missing cells: collapse_kernel + the sequential recursion + the loadings step of mstep_miss.hip."""
    import torch
    dev = torch.device("cuda", ctx.device)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    par = [up(G_["miss_start_" + k][None]) for k in KEYS]
    path, its, f, P = ctx.em_batch(up(G_["miss_x"][None]), *par, max_iter=2, tol=0.0)
    torch.cuda.synchronize()
    np.testing.assert_allclose(path.cpu().numpy()[0], G_["miss_path"], rtol=1e-9)
    fs = f.cpu().numpy()[0]
    assert np.abs(fs - G_["miss_f_smooth"]).max() <= 1e-8 * np.abs(G_["miss_f_smooth"]).max()
    for k, t in zip(KEYS, par):
        want = G_["miss_end_" + k]
        assert np.abs(t.cpu().numpy()[0] - want).max() <= 1e-8 * max(1.0, np.abs(want).max()), k
