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
