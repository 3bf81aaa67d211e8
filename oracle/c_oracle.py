"""ctypes loader for oracle/libdfm_oracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdfm_oracle.so")
_dp = ctypes.POINTER(ctypes.c_double)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "dfm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ks_pass(x, Lam, R, A, Q, mu0, P0):
    T, N = x.shape
    r = Lam.shape[1]
    x, Lam, R, A, Q, mu0, P0 = map(_c, (x, Lam, R, A, Q, mu0, P0))
    fs = np.empty((T, r)); Ps = np.empty((T, r * (r + 1) // 2)); ll = ctypes.c_double()
    f0 = np.empty(r); P0s = np.empty((r, r)); Pl = np.empty((T, r, r))
    rc = lib().dfm_oracle_ks_pass(T, N, r, _p(x), _p(Lam), _p(R), _p(A), _p(Q), _p(mu0), _p(P0),
                                  _p(fs), _p(Ps), ctypes.byref(ll), _p(f0), _p(P0s), _p(Pl))
    if rc:
        raise RuntimeError(f"dfm_oracle_ks_pass rc={rc}")
    return dict(f_smooth=fs, P_smooth_packed=Ps, loglik=ll.value, f0_smooth=f0, P0_smooth=P0s, P_lag=Pl)


def ks_pass_batch(panel, Lam, R, A, Q, mu0, P0, want_P=True, nthreads=0, out=None):
    """`out` = (f_smooth, P_smooth, loglik) persistent C-contiguous buffers (bench.py's cpu_baseline: no allocation and
    no first-touch page faults inside the timed region)."""
    B, T, N = panel.shape
    r = Lam.shape[2]
    panel, Lam, R, A, Q, mu0, P0 = map(_c, (panel, Lam, R, A, Q, mu0, P0))
    if out is not None:
        fs, Ps, ll = out
        assert fs.flags.c_contiguous and ll.flags.c_contiguous and (Ps is None or Ps.flags.c_contiguous)
    else:
        fs = np.empty((B, T, r)); Ps = np.empty((B, T, r * (r + 1) // 2)) if want_P else None
        ll = np.empty(B)
    rc = lib().dfm_oracle_ks_pass_batch(B, T, N, r, _p(panel), _p(Lam), _p(R), _p(A), _p(Q),
                                        _p(mu0), _p(P0), _p(fs), _p(Ps), _p(ll), int(nthreads))
    if rc:
        raise RuntimeError(f"dfm_oracle_ks_pass_batch rc={rc}")
    return fs, Ps, ll


def em_step(x, Lam, R, A, Q, mu0, P0):
    """Returns (new params dict, loglik at the input params)."""
    T, N = x.shape
    r = Lam.shape[1]
    x = _c(x)
    p = {k: _c(v).copy() for k, v in dict(Lam=Lam, R=R, A=A, Q=Q, mu0=mu0, P0=P0).items()}
    ll = ctypes.c_double()
    rc = lib().dfm_oracle_em_step(T, N, r, _p(x), _p(p["Lam"]), _p(p["R"]), _p(p["A"]), _p(p["Q"]),
                                  _p(p["mu0"]), _p(p["P0"]), ctypes.byref(ll))
    if rc:
        raise RuntimeError(f"dfm_oracle_em_step rc={rc}")
    return p, ll.value


def num_threads() -> int:
    return int(lib().dfm_oracle_num_threads())
