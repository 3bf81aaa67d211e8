"""The Julia shim's marshalling, executed (VERDICT r4 "parity residue" (a)): julia/dfm_hip.jl cannot run here (no Julia in the
image), so tests/host/cabi_colmajor.c -- a C host with the shim caller's memory layout (column-major T x ns data with NaN for
`missing`, column-major parameters) -- performs the shim's permutedims / reshape / flag logic line by line and calls
dfm_pca_init_batch and dfm_em_batch through include/dfm_hip.h.  Its column-major outputs must equal the oracle's."""
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import kalman_oracle as ko

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("Lam", "R", "A", "Q", "mu0", "P0")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cabi") / "cabi_colmajor")
    lib = os.path.join(ROOT, "dynamic_factor_models_amd", "lib")
    subprocess.run(["gcc", "-O1", "-std=c11", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host", "cabi_colmajor.c"),
                    "-o", out, "-L", lib, "-ldfmhip", "-lm", f"-Wl,-rpath,{lib}"], check=True)
    return out


def cm(a):
    """bytes of a matrix in Julia's (column-major) memory order"""
    return np.asfortranarray(a, dtype=np.float64).tobytes(order="F")


def _run(exe, tmp_path, blob):
    fi, fo = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fi, "wb") as f:
        f.write(blob)
    subprocess.run([exe, fi, fo], check=True)
    return open(fo, "rb").read()


def test_pca_start_through_the_shims_column_major_marshalling(exe, tmp_path):
    T, N, r = 120, 47, 4                                     # odd N, r padded to 4
    x, _ = ko.synth_replicate(11, N, T, r)
    ref, F = ko.pca_init(x, r)
    raw = _run(exe, tmp_path, struct.pack("5i", 0, T, N, r, 0) + cm(x))
    o = np.frombuffer(raw, dtype=np.float64)
    sizes = [N * r, N, r * r, r * r, r, r * r, T * r]
    shapes = [(N, r), (N,), (r, r), (r, r), (r,), (r, r), (T, r)]
    got, off = [], 0
    for n, sh in zip(sizes, shapes):
        got.append(o[off:off + n].reshape(sh, order="F")); off += n
    sgn = np.sign(np.sum(got[6] * F, axis=0))                # (a principal component's sign is the SVD routine's)
    np.testing.assert_allclose(got[6] * sgn, F, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(got[0] * sgn, ref["Lam"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(got[1], ref["R"], rtol=1e-8)
    np.testing.assert_allclose(got[2] * np.outer(sgn, sgn), ref["A"], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(got[3] * np.outer(sgn, sgn), ref["Q"], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(got[5] * np.outer(sgn, sgn), ref["P0"], rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("T,N,r,miss,iters", [(90, 58, 4, 0.1, 4), (64, 30, 3, 0.0, 3)])
def test_em_through_the_shims_column_major_marshalling(exe, tmp_path, T, N, r, miss, iters):
    x, _ = ko.synth_replicate(5, N, T, r, missing=miss)
    start, _ = ko.pca_init(np.nan_to_num(x), r)
    p, path, out = ko.em(x, start, max_iter=iters, tol=0.0)
    blob = struct.pack("5i", 1, T, N, r, iters) + cm(x) + b"".join(cm(start[k]) if np.ndim(start[k]) == 2 else np.asarray(start[k], np.float64).tobytes()
                                                                    for k in KEYS)
    raw = _run(exe, tmp_path, blob)
    its = struct.unpack("i", raw[:4])[0]
    assert its == iters
    o = np.frombuffer(raw[4:], dtype=np.float64)
    sizes = [N * r, N, r * r, r * r, r, r * r, iters, T * r]
    shapes = [(N, r), (N,), (r, r), (r, r), (r,), (r, r), (iters,), (T, r)]
    got, off = [], 0
    for n, sh in zip(sizes, shapes):
        got.append(o[off:off + n].reshape(sh, order="F")); off += n
    np.testing.assert_allclose(got[6], path, rtol=1e-9)
    for k, g in zip(KEYS, got[:6]):
        assert np.abs(g - p[k]).max() <= 1e-8 * max(1.0, np.abs(p[k]).max()), k
    assert np.abs(got[7] - out["f_smooth"]).max() <= 1e-8 * np.abs(out["f_smooth"]).max()
