"""CPU checks of the time-chunked recursion (csrc/recursion_chunk.hip) without a GPU:
  * the per-lane algebra the kernel's lanes execute (csrc/dfm_chunk_core.h: symmetric sweep, forward information-filter step,
    backward Z-smoother step with the EM accumulators) compiled for the host (tests/host/chunk_core_host.cpp) and run over a whole
    sample by one lane, against oracle/kalman_oracle.py;
  * the orchestration (64 chunks per replicate, warm-up from a guess, boundary checks) through its lane-level NumPy model
    (scripts/dbg/chunk_emul.py), against the same oracle.
The kernel itself is compared with the oracle in the GPU tests (tests/test_gpu_chunk.py)."""
import importlib.util
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import kalman_oracle as ko

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("chunk_emul", os.path.join(ROOT, "scripts", "dbg", "chunk_emul.py"))
ce = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ce)


@pytest.fixture(scope="module", params=["em", "pass"])
def host_exe(request, tmp_path_factory):
    """(executable, with_sums): the EM instantiation of the backward step (accumulators on) and the plain pass's (NoAcc)."""
    exe = str(tmp_path_factory.mktemp("chunk") / ("chunk_core_host_" + request.param))
    src = os.path.join(ROOT, "tests", "host", "chunk_core_host.cpp")
    flags = ["-DCHUNK_HOST_NOACC"] if request.param == "pass" else []
    subprocess.run(["g++", "-O1", "-std=c++17", *flags, "-o", exe, src], check=True)
    return exe, request.param == "em"


def _pad8(M, r, eye=False):
    out = np.eye(8) if eye else np.zeros((8, 8))
    out[:r, :r] = M
    return out


@pytest.mark.parametrize("N,T,r,miss", [(40, 30, 8, 0.2), (25, 17, 5, 0.0), (60, 40, 3, 0.4)])
def test_the_lane_algebra_reproduces_the_oracle(host_exe, N, T, r, miss):
    x, p = ko.synth_replicate(2, N, T, r, missing=miss)
    out = ko.kfs_pass(x, p["Lam"], p["R"], p["A"], p["Q"], p["mu0"], p["P0"], lag_one=True)
    b, s, n, ld, C = ko.collapse(x, p["Lam"], p["R"])
    A, Q, P0 = _pad8(p["A"], r), _pad8(p["Q"], r, True), _pad8(p["P0"], r, True)
    mu0 = np.zeros(8); mu0[:r] = p["mu0"]
    Qi = np.linalg.inv(Q); K = Qi @ A; Phi = K.T @ A
    P0i = np.linalg.inv(P0)
    C8 = np.zeros((T, 8, 8)); C8[:, :r, :r] = C
    b8 = np.zeros((T, 8)); b8[:, :r] = b
    il = np.tril_indices(8)
    blob = struct.pack("i", T) + np.concatenate([K.ravel(), Phi.ravel(), (Qi + Phi).ravel(), (P0i + Phi)[il], P0i @ mu0,
                                                   C8[:, il[0], il[1]].ravel(), b8.ravel()]).tobytes()
    host_exe, with_sums = host_exe
    res = subprocess.run([host_exe], input=blob, capture_output=True, check=True)
    o = np.frombuffer(res.stdout, dtype=np.float64)
    ldsum, xwsum, ldT, xfT = o[:4]
    Ps = o[4:4 + (T + 1) * 36].reshape(T + 1, 36)
    fs = o[4 + (T + 1) * 36:4 + (T + 1) * 44].reshape(T + 1, 8)
    S10 = o[4 + (T + 1) * 44:4 + (T + 1) * 44 + 64].reshape(8, 8)
    S11 = ko.unpack_sym(o[4 + (T + 1) * 44 + 64:], 8)
    LD = ldT + np.linalg.slogdet(P0)[1] + T * np.linalg.slogdet(Q)[1] + ldsum
    QD = s.sum() + mu0 @ P0i @ mu0 - xfT - xwsum
    ll = -0.5 * (n.sum() * ko.LOG2PI + ld.sum() + LD + QD)
    np.testing.assert_allclose(ll, out["loglik"], rtol=1e-11)
    Pfull = ko.unpack_sym(Ps, 8)
    np.testing.assert_allclose(Pfull[1:, :r, :r], out["P_smooth"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(fs[1:, :r], out["f_smooth"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(Pfull[0, :r, :r], out["P0_smooth"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(fs[0, :r], out["f0_smooth"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(Pfull[:, r:, r:], np.broadcast_to(np.eye(8 - r), (T + 1, 8 - r, 8 - r)), atol=1e-13)   # padding states
    if not with_sums:
        return
    fa = np.vstack([out["f0_smooth"][None], out["f_smooth"]])
    S10o = out["P_lag"].sum(0) + fa[1:].T @ fa[:-1]
    S11o = out["P_smooth"].sum(0) + fa[1:].T @ fa[1:]
    np.testing.assert_allclose(S10[:r, :r], S10o, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(S11[:r, :r], S11o, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("N,T,r,miss,ok", [(200, 500, 8, 0.1, True), (139, 222, 4, 0.1, True), (200, 100, 8, 0.1, True),
                                           (200, 1100, 8, 0.05, True), (10, 300, 8, 0.2, False)])
def test_the_chunk_model_reproduces_the_oracle_pass_or_says_it_did_not(N, T, r, miss, ok):
    x, p = ko.synth_replicate(3, N, T, r, missing=miss)
    out = ko.kfs_pass(x, p["Lam"], p["R"], p["A"], p["Q"], p["mu0"], p["P0"])
    f, P, ll, info = ce.chunk_pass(x, p["Lam"], p["R"], p["A"], p["Q"], p["mu0"], p["P0"], W=8, tol=1e-10)
    assert info["ok"] == ok                       # a slowly forgetting filter (few series) fails its boundary checks: sequential kernel
    if ok:
        np.testing.assert_allclose(ll, out["loglik"], rtol=1e-11)
        np.testing.assert_allclose(f, out["f_smooth"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(P, out["P_smooth"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(info["f0"], out["f0_smooth"], atol=1e-11)


# ---- the boundary check itself (dfm_chunk_core.h state_gap / gap_close) ---------------------------------------------------------------
def _old_weights():
    """The weight vectors of the check this one replaces (round 5: two weighted sums of the packed matrix, one of the vector)."""
    k = np.arange(36, dtype=np.uint64)
    g1 = 1.0 + ((k * 40503 + 12345) & 0xFFFF) / 65536.0
    g2 = 1.0 + ((k * 30011 + 54321) & 0xFFFF) / 65536.0
    i = np.arange(8, dtype=np.uint64)
    g = 1.0 + ((i * 50021 + 777) & 0xFFFF) / 65536.0
    return g1, g2, g


def _gap_verdicts(tmp_path, records):
    exe = str(tmp_path / "chunk_gap_host")
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "host", "chunk_gap_host.cpp")], check=True)
    blob = struct.pack("i", len(records))
    for tol, m, x, m2, x2 in records:
        blob += struct.pack("d", tol) + np.concatenate([m, x, m2, x2]).astype(np.float64).tobytes()
    res = subprocess.run([exe], input=blob, capture_output=True, check=True)
    return [int(v) for v in res.stdout.split()]


def test_the_boundary_check_is_a_bound_not_a_projection(tmp_path):
    rng = np.random.default_rng(7)
    A = rng.standard_normal((8, 8)); S = A @ A.T + 8 * np.eye(8)
    m = S[np.tril_indices(8)]
    x = rng.standard_normal(8)
    g1, g2, g = _old_weights()
    # an error of 1e-3 (relative) in the null space of BOTH matrix weight vectors, and one orthogonal to the vector's weights:
    # the weighted sums of round 5 are unchanged to rounding, every entry is off
    _, _, vt = np.linalg.svd(np.vstack([g1, g2]))
    em = vt[-1] * 1e-3 * np.abs(m).max() / np.abs(vt[-1]).max()
    ex = rng.standard_normal(8); ex -= g * (g @ ex) / (g @ g); ex *= 1e-3 * np.abs(x).max() / np.abs(ex).max()
    assert abs(g1 @ em) < 1e-12 * np.abs(m).sum() and abs(g2 @ em) < 1e-12 * np.abs(m).sum() and abs(g @ ex) < 1e-12 * np.abs(x).sum()
    tiny_m = 1e-12 * np.abs(m).max() * rng.uniform(-1, 1, 36)
    tiny_x = 1e-12 * np.abs(x).max() * rng.uniform(-1, 1, 8)
    nan_m = m.copy(); nan_m[17] = np.nan
    one = m.copy(); one[30] += 3e-10 * np.abs(m).max()                       # ONE entry off by three tolerances
    recs = [(1e-10, m, x, m, x),                                             # identical
            (1e-10, m, x, m + tiny_m, x + tiny_x),                           # inside the tolerance, every entry
            (1e-10, m, x, m + em, x),                                        # orthogonal to the old matrix weights
            (1e-10, m, x, m, x + ex),                                        # orthogonal to the old vector weights
            (1e-10, m, x, one, x),
            (1e-10, m, x, nan_m, x),                                         # NaN never passes
            (1e-10, nan_m, x, nan_m, x),
            (1e-10, m, x, m, x + np.inf),
            (1e-2, m, x, m + em, x + ex)]                                    # ... and a tolerance above the error accepts it
    assert _gap_verdicts(tmp_path, recs) == [1, 1, 0, 0, 0, 0, 0, 0, 1]


@pytest.mark.parametrize("W,N,miss", [(8, 32, 0.2), (8, 48, 0.0), (8, 48, 0.4), (8, 64, 0.2), (8, 96, 0.6), (4, 64, 0.0), (4, 96, 0.4), (2, 32, 0.6)])
def test_the_boundary_residual_bounds_the_error_of_the_result(W, N, miss):
    """The intermediate forgetting regime (boundary states off by 1e-10 ... 1e-2): the largest element-wise boundary residual of a
    replicate is an upper bound on the error of its smoothed moments -- so a replicate the kernel keeps at tolerance tol is within
    tol (tests/test_gpu_chunk.py runs the same sweep through the kernel)."""
    T, r = 200, 8
    for b in range(3):
        x, p = ko.synth_replicate(100 + b, N, T, r, missing=miss)
        out = ko.kfs_pass(x, p["Lam"], p["R"], p["A"], p["Q"], p["mu0"], p["P0"])
        f, P, ll, info = ce.chunk_pass(x, p["Lam"], p["R"], p["A"], p["Q"], p["mu0"], p["P0"], W=W, tol=1e-10)
        res = max(info["res_f"].max(), info["res_b"].max())
        err = max(np.abs(f - out["f_smooth"]).max() / np.abs(out["f_smooth"]).max(),
                  np.abs(P - out["P_smooth"]).max() / np.abs(out["P_smooth"]).max(), abs(ll - out["loglik"]) / abs(out["loglik"]))
        assert 1e-12 < res and err <= res, (res, err)
