"""Pins the CPU oracle of the (reference-absent) Kalman/RTS/EM path: brute-force Gaussian
conditioning, the independent textbook filter, EM monotonicity, and the
independent C twin."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import kalman_oracle as ko


def _small(seed, N=5, T=7, r=2, miss=0.2):
    rng = np.random.default_rng(seed)
    Lam = rng.standard_normal((N, r)); R = rng.uniform(.5, 1.5, N)
    A = 0.6 * np.eye(r) + 0.15 * rng.standard_normal((r, r))
    G = rng.standard_normal((r, r)); Q = G @ G.T + 0.3 * np.eye(r)
    mu0 = rng.standard_normal(r)
    G0 = rng.standard_normal((r, r)); P0 = G0 @ G0.T + 0.5 * np.eye(r)
    x = rng.standard_normal((T, N))
    x[rng.random((T, N)) < miss] = np.nan
    return x, dict(Lam=Lam, R=R, A=A, Q=Q, mu0=mu0, P0=P0)


@pytest.mark.parametrize("seed,N,T,r,miss", [(0, 5, 7, 2, .2), (1, 4, 6, 3, 0.), (2, 6, 5, 1, .3),
                                              (3, 3, 8, 3, .4)])
def test_collapsed_pass_equals_brute_force(seed, N, T, r, miss):
    x, p = _small(seed, N, T, r, miss)
    a = ko.kfs_pass(x, **p); b = ko.brute_force_gaussian(x, **p)
    assert abs(a["loglik"] - b["loglik"]) <= 1e-10 * abs(b["loglik"])
    for k in ("f_smooth", "P_smooth", "f0_smooth", "P0_smooth", "P_lag"):
        np.testing.assert_allclose(a[k], b[k], rtol=0, atol=1e-10, err_msg=k)


def test_row_with_every_cell_missing():
    x, p = _small(5, 4, 6, 2, 0.1)
    x[2, :] = np.nan
    a = ko.kfs_pass(x, **p); b = ko.brute_force_gaussian(x, **p)
    assert abs(a["loglik"] - b["loglik"]) <= 1e-10 * abs(b["loglik"])
    np.testing.assert_allclose(a["f_smooth"], b["f_smooth"], atol=1e-10)


@pytest.mark.parametrize("miss", [0.0, 0.15])
def test_textbook_filter_agrees(miss):
    x, p = ko.synth_replicate(3, 30, 50, 4, missing=miss)
    a = ko.kfs_pass(x, **p); c = ko.kfs_pass_textbook(x, **p)
    assert abs(a["loglik"] - c["loglik"]) <= 1e-11 * abs(a["loglik"])
    np.testing.assert_allclose(a["f_smooth"], c["f_smooth"], atol=1e-11)
    np.testing.assert_allclose(a["P_smooth"], c["P_smooth"], atol=1e-11)


@pytest.mark.parametrize("miss", [0.0, 0.1])
def test_em_monotone(miss):
    x, _ = ko.synth_replicate(0, 30, 80, 3, missing=miss)
    p0, _ = ko.pca_init(np.nan_to_num(x), 3)
    _, path, _ = ko.em(x, p0, 25)
    assert np.all(np.diff(path) > -1e-8 * np.abs(path[:-1]))
    assert path[-1] > path[0]


@pytest.mark.parametrize("miss", [0.0, 0.15])
def test_c_twin_matches_numpy(miss):
    x, p = ko.synth_replicate(1, 40, 60, 4, missing=miss)
    a = ko.kfs_pass(x, **p); c = co.ks_pass(x, **p)
    assert abs(a["loglik"] - c["loglik"]) <= 1e-12 * abs(a["loglik"])
    np.testing.assert_allclose(c["f_smooth"], a["f_smooth"], atol=1e-12)
    np.testing.assert_allclose(c["P_smooth_packed"], ko.pack_sym(a["P_smooth"]), atol=1e-12)
    np.testing.assert_allclose(c["P_lag"], a["P_lag"], atol=1e-12)
    np.testing.assert_allclose(c["f0_smooth"], a["f0_smooth"], atol=1e-12)
    new, ll, _ = ko.em_step(x, **p); new_c, ll_c = co.em_step(x, **p)
    assert abs(ll - ll_c) <= 1e-12 * abs(ll)
    for k in new:
        np.testing.assert_allclose(new_c[k], new[k], atol=1e-11, err_msg=k)


def test_c_batch_driver():
    B, N, T, r = 3, 20, 30, 2
    reps = [ko.synth_replicate(b, N, T, r) for b in range(B)]
    panel = np.stack([x for x, _ in reps])
    st = {k: np.stack([p[k] for _, p in reps]) for k in reps[0][1]}
    fs, Ps, ll = co.ks_pass_batch(panel, st["Lam"], st["R"], st["A"], st["Q"], st["mu0"], st["P0"])
    for b in range(B):
        a = ko.kfs_pass(reps[b][0], **reps[b][1])
        np.testing.assert_allclose(fs[b], a["f_smooth"], atol=1e-12)
        np.testing.assert_allclose(Ps[b], ko.pack_sym(a["P_smooth"]), atol=1e-12)
        assert abs(ll[b] - a["loglik"]) <= 1e-12 * abs(a["loglik"])


def test_pack_roundtrip():
    P = np.arange(16.).reshape(4, 4); P = P + P.T
    np.testing.assert_array_equal(ko.unpack_sym(ko.pack_sym(P), 4), P)
    assert list(ko.pack_sym(P)[:3]) == [P[0, 0], P[1, 0], P[1, 1]]
