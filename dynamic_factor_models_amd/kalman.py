"""Host-side wrappers of the C-ABI (include/dfm_hip.h) over torch device tensors / NumPy arrays.

torch is plumbing only (device memory, the current HIP stream, torch.distributed); every number is
produced by the hand-written gfx950 kernels inside libdfmhip.so.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np

from . import _lib


def _check(h, rc: int):
    if rc != 0:
        raise _lib.DfmError(rc, _lib.load().dfm_last_error(h).decode())


class DfmMulti:
    """The library's multi-GPU object (dfm_multi, csrc/multi.hip): `ngpu` GPUs of this node driven from THIS process -- one
    handle, stream and workspace per GPU and ONE RCCL communicator, created once; the job's replicates stay resident in
    the GPUs' HBM between calls.  What a host without torch.distributed (Julia) uses; `bench.py --driver lib` times it.
    force_comm: build the (1-rank) communicator also for ngpu = 1, so that the all-gather path runs on one GPU."""

    def __init__(self, ngpu: int = 1, device_ids=None, force_comm: bool = False):
        self._lib = _lib.load()
        ids = None if device_ids is None else np.ascontiguousarray(device_ids, dtype=np.int32)
        h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(700)
        rc = self._lib.dfm_multi_create(ctypes.byref(h), int(ngpu), None if ids is None else ctypes.c_void_p(ids.ctypes.data),
                                        _lib.DFM_MULTI_F_FORCE_COMM if force_comm else 0, err, 700)
        if rc != 0:
            raise _lib.DfmError(rc, err.value.decode())
        self._m = h
        self.shape = None
        self._max_iter = 0

    def close(self):
        if getattr(self, "_m", None):
            self._lib.dfm_multi_destroy(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise _lib.DfmError(rc, self._lib.dfm_multi_last_error(self._m).decode())

    @property
    def ngpu(self):
        return self._lib.dfm_multi_ngpu(self._m)

    @property
    def has_comm(self):
        return bool(self._lib.dfm_multi_has_comm(self._m))

    def load(self, panel, Lam, R, A, Q, mu0, P0):
        """Upload a job (NumPy, layouts of em_batch_host) to the GPUs that own its replicates."""
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        panel, Lam, R, A, Q, mu0, P0 = map(c, (panel, Lam, R, A, Q, mu0, P0))
        B, T, N = panel.shape
        r = Lam.shape[2]
        p = lambda a: ctypes.c_void_p(a.ctypes.data)
        self._ck(self._lib.dfm_multi_load(self._m, B, T, N, r, p(panel), p(Lam), p(R), p(A), p(Q), p(mu0), p(P0)))
        self.shape = (B, T, N, r)

    def synth(self, seed: int, first_replicate: int, B: int, T: int, N: int, r: int, missing_prob: float = 0.0,
              pca_start: bool = False):
        """Generate the job where it lives: GPU g draws replicates first_replicate + [lo_g, hi_g) (dfm_synth_panels_dev);
        pca_start replaces the DGP parameters by the PCA + OLS start."""
        self._ck(self._lib.dfm_multi_synth(self._m, int(seed), int(first_replicate), B, T, N, r, float(missing_prob),
                                           1 if pca_start else 0))
        self.shape = (B, T, N, r)

    def ks_pass(self, want_P: bool = True, may_have_missing: bool = False, singular_q: bool = False):
        flags = (_lib.DFM_F_MAY_HAVE_MISSING if may_have_missing else 0) | (_lib.DFM_F_SINGULAR_Q if singular_q else 0)
        self._ck(self._lib.dfm_multi_ks_pass(self._m, 1 if want_P else 0, flags))

    def em(self, max_iter: int = 10, tol: float = 0.0, want_smooth: bool = True, want_P: bool = True,
           may_have_missing: bool = False, singular_q: bool = False) -> int:
        """The EM loop on the resident job (parameters updated in place on the GPUs); returns the iterations run."""
        flags = (_lib.DFM_F_MAY_HAVE_MISSING if may_have_missing else 0) | (_lib.DFM_F_SINGULAR_Q if singular_q else 0)
        ran = ctypes.c_int(0)
        self._max_iter = int(max_iter)
        self._ck(self._lib.dfm_multi_em(self._m, int(max_iter), float(tol), 1 if want_smooth else 0, 1 if want_P else 0, flags,
                                        ctypes.cast(ctypes.byref(ran), ctypes.c_void_p)))
        return ran.value

    def fetch(self, what: str):
        """One resident array of the whole job, global replicate order (NumPy)."""
        B, T, N, r = self.shape
        npk = r * (r + 1) // 2
        table = {"Lam": (_lib.DFM_MULTI_LAM, (B, N, r), np.float64), "R": (_lib.DFM_MULTI_R, (B, N), np.float64),
                 "A": (_lib.DFM_MULTI_A, (B, r, r), np.float64), "Q": (_lib.DFM_MULTI_Q, (B, r, r), np.float64),
                 "mu0": (_lib.DFM_MULTI_MU0, (B, r), np.float64), "P0": (_lib.DFM_MULTI_P0, (B, r, r), np.float64),
                 "f_smooth": (_lib.DFM_MULTI_F_SMOOTH, (B, T, r), np.float64),
                 "P_smooth": (_lib.DFM_MULTI_P_SMOOTH, (B, T, npk), np.float64),
                 "loglik": (_lib.DFM_MULTI_LOGLIK, (B,), np.float64),
                 "loglik_path": (_lib.DFM_MULTI_LOGLIK_PATH, (B, self._max_iter), np.float64),
                 "iters": (_lib.DFM_MULTI_ITERS, (B,), np.int32), "panel": (_lib.DFM_MULTI_PANEL, (B, T, N), np.float64)}
        code, shape, dt = table[what]
        out = np.empty(shape, dtype=dt)
        self._ck(self._lib.dfm_multi_fetch(self._m, code, ctypes.c_void_p(out.ctypes.data)))
        return out


class DfmContext:
    """One libdfmhip handle bound to a HIP device and (by default) torch's current stream."""

    def __init__(self, device: Optional[int] = None, use_torch_stream: bool = True):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("DfmContext needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback for this path")
        self._lib = _lib.load()
        self.device = torch.cuda.current_device() if device is None else int(device)
        self._torch = torch
        self._use_torch_stream = bool(use_torch_stream)
        stream = torch.cuda.current_stream(self.device).cuda_stream if use_torch_stream else None
        h = ctypes.c_void_p()
        rc = self._lib.dfm_create(ctypes.byref(h), self.device, ctypes.c_void_p(stream))
        if rc != 0:
            raise _lib.DfmError(rc, "dfm_create failed")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dfm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _dev(self, t, name, shape=None):
        torch = self._torch
        if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float64 or not t.is_contiguous():
            raise TypeError(f"{name}: expected a contiguous float64 tensor on the HIP device")
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name}: shape {tuple(t.shape)} != expected {tuple(shape)}")
        return ctypes.c_void_p(t.data_ptr())

    def _sync_stream(self):
        if not self._use_torch_stream:      # the handle keeps the stream it created for itself
            return
        s = self._torch.cuda.current_stream(self.device).cuda_stream
        self._lib.dfm_set_stream(self._h, ctypes.c_void_p(s))

    def synchronize(self):
        """Wait for the handle's stream AND surface the status word of the last call (NaN in a panel declared balanced,
        an expired bounded wait of the one-launch pass, PCA start not converged): the device-pointer entry points only
        enqueue, so this is where their failures become exceptions."""
        _check(self._h, self._lib.dfm_synchronize(self._h))

    check_status = synchronize

    def hbm_probe(self, nbytes: int = 1 << 30, iters: int = 10):
        """dfm_hbm_probe: {"read_dma": GB/s, "copy": GB/s, "write": GB/s} of this device, measured now."""
        self._torch.cuda.synchronize(self.device)
        out = {}
        for name, mode in (("read_dma", 0), ("copy", 1), ("write", 2)):
            g = ctypes.c_double(); ms = ctypes.c_double()
            with self._torch.cuda.device(self.device):
                _check(self._h, self._lib.dfm_hbm_probe(self._h, int(nbytes), mode, int(iters), ctypes.byref(g), ctypes.byref(ms)))
            out[name] = g.value
        return out

    def chunk_fallbacks(self):
        """dfm_chunk_fallbacks: (failed, total) replicates of the last pass that ran on a time-chunked recursion (total = 0:
        it did not); `failed` of them were redone by the sequential kernel."""
        nf = ctypes.c_int(); nt = ctypes.c_int()
        _check(self._h, self._lib.dfm_chunk_fallbacks(self._h, ctypes.byref(nf), ctypes.byref(nt)))
        return nf.value, nt.value

    def profile_enable(self, on: bool = True):
        self._sync_stream()
        _check(self._h, self._lib.dfm_profile_enable(self._h, 1 if on else 0))

    def profile_read(self):
        """{kernel name: (total ms, launches)} since profile_enable(True)."""
        out = {}
        idx = 0
        while True:
            name = ctypes.create_string_buffer(64)
            ms = ctypes.c_double(); n = ctypes.c_int()
            rc = self._lib.dfm_profile_read(self._h, idx, name, 64, ctypes.byref(ms), ctypes.byref(n))
            if rc != 0:
                break
            if n.value:
                out[name.value.decode()] = (ms.value, n.value)
            idx += 1
        return out

    @staticmethod
    def _flags(panel, may_have_missing, singular_q=False):
        import torch
        if may_have_missing is None:
            may_have_missing = bool(torch.isnan(panel).any().item())
        return (_lib.DFM_F_MAY_HAVE_MISSING if may_have_missing else 0) | (_lib.DFM_F_SINGULAR_Q if singular_q else 0)

    # ------------------------------------------------------------------ smoother pass
    def ks_pass_batch(self, panel, Lam, R, A, Q, mu0, P0, want_P: bool = True,
                      may_have_missing: Optional[bool] = None, out=None, singular_q: bool = False):
        """One Kalman-smoother pass per replicate (device tensors in, device tensors out).
        Returns (f_smooth [B,T,r], P_smooth [B,T,r(r+1)/2] or None, loglik [B]).  Asynchronous on
        torch's current stream."""
        torch = self._torch
        B, T, N = panel.shape
        r = Lam.shape[2]
        flags = self._flags(panel, may_have_missing, singular_q)   # singular_q: DFM_F_SINGULAR_Q (covariance form)
        if out is None:
            f = torch.empty((B, T, r), dtype=torch.float64, device=panel.device)
            P = torch.empty((B, T, r * (r + 1) // 2), dtype=torch.float64, device=panel.device) if want_P else None
            ll = torch.empty((B,), dtype=torch.float64, device=panel.device)
        else:
            f, P, ll = out
        self._sync_stream()
        rc = self._lib.dfm_ks_pass_batch_dev(
            self._h, B, T, N, r, self._dev(panel, "panel"), self._dev(Lam, "Lam", (B, N, r)),
            self._dev(R, "R", (B, N)), self._dev(A, "A", (B, r, r)), self._dev(Q, "Q", (B, r, r)),
            self._dev(mu0, "mu0", (B, r)), self._dev(P0, "P0", (B, r, r)), self._dev(f, "f_smooth"),
            self._dev(P, "P_smooth") if P is not None else None, self._dev(ll, "loglik"), flags)
        _check(self._h, rc)
        return f, P, ll

    def ks_pass_batch_host(self, panel, Lam, R, A, Q, mu0, P0, want_P: bool = True,
                           may_have_missing: Optional[bool] = None, singular_q: bool = False):
        """Same through the HOST-pointer entry point (what Julia's ccall binds): NumPy in/out."""
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        panel, Lam, R, A, Q, mu0, P0 = map(c, (panel, Lam, R, A, Q, mu0, P0))
        B, T, N = panel.shape
        r = Lam.shape[2]
        if may_have_missing is None:
            may_have_missing = bool(np.isnan(panel).any())
        flags = (_lib.DFM_F_MAY_HAVE_MISSING if may_have_missing else 0) | (_lib.DFM_F_SINGULAR_Q if singular_q else 0)
        f = np.empty((B, T, r)); P = np.empty((B, T, r * (r + 1) // 2)) if want_P else None
        ll = np.empty(B)
        p = lambda a: None if a is None else ctypes.c_void_p(a.ctypes.data)
        rc = self._lib.dfm_ks_pass_batch(self._h, B, T, N, r, p(panel), p(Lam), p(R), p(A), p(Q), p(mu0),
                                         p(P0), p(f), p(P), p(ll), flags)
        _check(self._h, rc)
        return f, P, ll

    # ------------------------------------------------------------------ EM
    def em_step_batch(self, panel, Lam, R, A, Q, mu0, P0, may_have_missing: Optional[bool] = None):
        """One EM iteration per replicate; parameters (device tensors) are updated IN PLACE.
        Returns loglik [B] at the parameters passed in."""
        torch = self._torch
        B, T, N = panel.shape
        r = Lam.shape[2]
        flags = self._flags(panel, may_have_missing)
        ll = torch.empty((B,), dtype=torch.float64, device=panel.device)
        self._sync_stream()
        rc = self._lib.dfm_em_step_batch_dev(
            self._h, B, T, N, r, self._dev(panel, "panel"), self._dev(Lam, "Lam", (B, N, r)),
            self._dev(R, "R", (B, N)), self._dev(A, "A", (B, r, r)), self._dev(Q, "Q", (B, r, r)),
            self._dev(mu0, "mu0", (B, r)), self._dev(P0, "P0", (B, r, r)), self._dev(ll, "loglik"), flags)
        _check(self._h, rc)
        return ll

    def em_batch(self, panel, Lam, R, A, Q, mu0, P0, max_iter: int = 10, tol: float = 0.0,
                 want_smooth: bool = True, want_P: bool = True, may_have_missing: Optional[bool] = None,
                 singular_q: bool = False):
        """max_iter EM iterations (parameters updated in place).  Returns
        (loglik_path [B,max_iter] (NaN past iters[b]), iters [B] int32, f_smooth, P_smooth)."""
        torch = self._torch
        B, T, N = panel.shape
        r = Lam.shape[2]
        flags = self._flags(panel, may_have_missing, singular_q)
        dev = panel.device
        path = torch.empty((B, max_iter), dtype=torch.float64, device=dev)
        iters = torch.empty((B,), dtype=torch.int32, device=dev)
        f = torch.empty((B, T, r), dtype=torch.float64, device=dev) if want_smooth else None
        P = torch.empty((B, T, r * (r + 1) // 2), dtype=torch.float64, device=dev) if (want_smooth and want_P) else None
        self._sync_stream()
        rc = self._lib.dfm_em_batch_dev(
            self._h, B, T, N, r, self._dev(panel, "panel"), self._dev(Lam, "Lam", (B, N, r)),
            self._dev(R, "R", (B, N)), self._dev(A, "A", (B, r, r)), self._dev(Q, "Q", (B, r, r)),
            self._dev(mu0, "mu0", (B, r)), self._dev(P0, "P0", (B, r, r)), int(max_iter), float(tol),
            self._dev(path, "loglik_path"), ctypes.c_void_p(iters.data_ptr()),
            self._dev(f, "f_smooth") if f is not None else None,
            self._dev(P, "P_smooth") if P is not None else None, flags)
        _check(self._h, rc)
        return path, iters, f, P

    def em_iterate_batch(self, panel, Lam, R, A, Q, mu0, P0, k: int, max_iter: int, tol: float, path, iters, active,
                         f=None, P=None, may_have_missing: Optional[bool] = False, singular_q: bool = False):
        """EM iteration number k of max_iter (dfm_em_iterate_batch_dev): parameters updated in place, bookkeeping in the
        CALLER's device tensors path [B,max_iter] f64, iters [B] i32, active [B] i32 (they persist between calls; the
        k = 0 call initialises them).  The unit a multi-GPU driver steps: shard.em_batch_sharded."""
        B, T, N = panel.shape
        r = Lam.shape[2]
        flags = self._flags(panel, may_have_missing, singular_q)
        self._sync_stream()
        rc = self._lib.dfm_em_iterate_batch_dev(
            self._h, B, T, N, r, self._dev(panel, "panel"), self._dev(Lam, "Lam", (B, N, r)),
            self._dev(R, "R", (B, N)), self._dev(A, "A", (B, r, r)), self._dev(Q, "Q", (B, r, r)),
            self._dev(mu0, "mu0", (B, r)), self._dev(P0, "P0", (B, r, r)), int(k), int(max_iter), float(tol),
            self._dev(path, "loglik_path", (B, max_iter)), ctypes.c_void_p(iters.data_ptr()),
            ctypes.c_void_p(active.data_ptr()), self._dev(f, "f_smooth") if f is not None else None,
            self._dev(P, "P_smooth") if P is not None else None, flags)
        _check(self._h, rc)

    def em_batch_host(self, panel, Lam, R, A, Q, mu0, P0, max_iter: int = 10, tol: float = 0.0,
                      may_have_missing: Optional[bool] = None, singular_q: bool = False):
        """Host-pointer EM entry (what Julia's ccall binds).  Returns (params dict, loglik_path, iters,
        f_smooth, P_smooth); inputs are not modified."""
        c = lambda a: np.array(a, dtype=np.float64, order="C", copy=True)
        panel = np.ascontiguousarray(panel, dtype=np.float64)
        Lam, R, A, Q, mu0, P0 = map(c, (Lam, R, A, Q, mu0, P0))
        B, T, N = panel.shape
        r = Lam.shape[2]
        if may_have_missing is None:
            may_have_missing = bool(np.isnan(panel).any())
        flags = (_lib.DFM_F_MAY_HAVE_MISSING if may_have_missing else 0) | (_lib.DFM_F_SINGULAR_Q if singular_q else 0)
        path = np.empty((B, max_iter)); iters = np.empty(B, dtype=np.int32)
        f = np.empty((B, T, r)); P = np.empty((B, T, r * (r + 1) // 2))
        p = lambda a: ctypes.c_void_p(a.ctypes.data)
        rc = self._lib.dfm_em_batch(self._h, B, T, N, r, p(panel), p(Lam), p(R), p(A), p(Q), p(mu0), p(P0),
                                    int(max_iter), float(tol), p(path), p(iters), p(f), p(P), flags)
        _check(self._h, rc)
        return dict(Lam=Lam, R=R, A=A, Q=Q, mu0=mu0, P0=P0), path, iters, f, P

    def em_obs_batch_host(self, panel, G, Lam, R, A, Q, mu0, P0, max_iter: int = 10, tol: float = 0.0,
                          may_have_missing: Optional[bool] = None):
        """dfm_em_obs_batch: EM of the model with OBSERVED factors g_t as known regressors (include/dfm_hip.h).
        panel [B,T,N], G [B,T,r_o] (no NaN), Lam [B,N,r_o+r_u] (observed-factor loadings first), A / Q / P0 [B,r_u,r_u],
        mu0 [B,r_u].  Returns (params dict, loglik_path, iters, f_smooth [B,T,r_u], P_smooth); inputs are not modified."""
        c = lambda a: np.array(a, dtype=np.float64, order="C", copy=True)
        panel = np.ascontiguousarray(panel, dtype=np.float64)
        G = np.ascontiguousarray(G, dtype=np.float64)
        Lam, R, A, Q, mu0, P0 = map(c, (Lam, R, A, Q, mu0, P0))
        B, T, N = panel.shape
        ro = G.shape[2]
        ru = Lam.shape[2] - ro
        if G.shape[:2] != (B, T) or A.shape != (B, ru, ru):
            raise ValueError("em_obs_batch_host: G must be [B,T,r_o] and A [B,r_u,r_u] with Lam [B,N,r_o+r_u]")
        if may_have_missing is None:
            may_have_missing = bool(np.isnan(panel).any())
        flags = _lib.DFM_F_MAY_HAVE_MISSING if may_have_missing else 0
        path = np.empty((B, max_iter)); iters = np.empty(B, dtype=np.int32)
        f = np.empty((B, T, ru)); P = np.empty((B, T, ru * (ru + 1) // 2))
        p = lambda a: ctypes.c_void_p(a.ctypes.data)
        rc = self._lib.dfm_em_obs_batch(self._h, B, T, N, ru, ro, p(panel), p(G), p(Lam), p(R), p(A), p(Q), p(mu0), p(P0),
                                        int(max_iter), float(tol), p(path), p(iters), p(f), p(P), flags)
        _check(self._h, rc)
        return dict(Lam=Lam, R=R, A=A, Q=Q, mu0=mu0, P0=P0), path, iters, f, P

    # ------------------------------------------------------------------ several GPUs from one process (multi.hip)
    @staticmethod
    def em_batch_multi_host(ngpu, panel, Lam, R, A, Q, mu0, P0, max_iter: int = 10, tol: float = 0.0,
                            may_have_missing: Optional[bool] = None, device_ids=None, singular_q: bool = False):
        """dfm_em_batch_multi: the EM loop on `ngpu` GPUs from THIS process (one host thread per GPU, library-owned
        RCCL communicator, one all-gather of {loglik, active} per iteration) -- what the Julia host binds.  NumPy in /
        out as em_batch_host; also returns the number of iterations every GPU ran."""
        lib = _lib.load()
        c = lambda a: np.array(a, dtype=np.float64, order="C", copy=True)
        panel = np.ascontiguousarray(panel, dtype=np.float64)
        Lam, R, A, Q, mu0, P0 = map(c, (Lam, R, A, Q, mu0, P0))
        B, T, N = panel.shape
        r = Lam.shape[2]
        if may_have_missing is None:
            may_have_missing = bool(np.isnan(panel).any())
        flags = (_lib.DFM_F_MAY_HAVE_MISSING if may_have_missing else 0) | (_lib.DFM_F_SINGULAR_Q if singular_q else 0)
        path = np.empty((B, max_iter)); iters = np.empty(B, dtype=np.int32)
        f = np.empty((B, T, r)); P = np.empty((B, T, r * (r + 1) // 2))
        ids = None if device_ids is None else np.ascontiguousarray(device_ids, dtype=np.int32)
        ran = ctypes.c_int(0)
        err = ctypes.create_string_buffer(700)
        p = lambda a: None if a is None else ctypes.c_void_p(a.ctypes.data)
        rc = lib.dfm_em_batch_multi(int(ngpu), p(ids), B, T, N, r, p(panel), p(Lam), p(R), p(A), p(Q), p(mu0), p(P0),
                                    int(max_iter), float(tol), p(path), p(iters), p(f), p(P), flags,
                                    ctypes.cast(ctypes.byref(ran), ctypes.c_void_p), err, 700)
        if rc != 0:
            raise _lib.DfmError(rc, err.value.decode())
        return dict(Lam=Lam, R=R, A=A, Q=Q, mu0=mu0, P0=P0), path, iters, f, P, ran.value

    @staticmethod
    def ks_pass_batch_multi_host(ngpu, panel, Lam, R, A, Q, mu0, P0, may_have_missing: Optional[bool] = None,
                                 device_ids=None):
        """dfm_ks_pass_batch_multi: the smoother pass with the replicates split over `ngpu` GPUs (no exchange)."""
        lib = _lib.load()
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        panel, Lam, R, A, Q, mu0, P0 = map(c, (panel, Lam, R, A, Q, mu0, P0))
        B, T, N = panel.shape
        r = Lam.shape[2]
        if may_have_missing is None:
            may_have_missing = bool(np.isnan(panel).any())
        flags = _lib.DFM_F_MAY_HAVE_MISSING if may_have_missing else 0
        f = np.empty((B, T, r)); P = np.empty((B, T, r * (r + 1) // 2)); ll = np.empty(B)
        ids = None if device_ids is None else np.ascontiguousarray(device_ids, dtype=np.int32)
        err = ctypes.create_string_buffer(700)
        p = lambda a: None if a is None else ctypes.c_void_p(a.ctypes.data)
        rc = lib.dfm_ks_pass_batch_multi(int(ngpu), p(ids), B, T, N, r, p(panel), p(Lam), p(R), p(A), p(Q), p(mu0), p(P0),
                                         p(f), p(P), p(ll), flags, err, 700)
        if rc != 0:
            raise _lib.DfmError(rc, err.value.decode())
        return f, P, ll

    # ------------------------------------------------------------------ VAR(p) factor dynamics (companion form)
    def ks_pass_varp_batch(self, panel, Lam, R, Avar, Q, mu0, P0, want_P: bool = True,
                           may_have_missing: Optional[bool] = None, singular_q: bool = False):
        """Smoother pass of x_t = Lam f_t + e_t, f_t = A_1 f_{t-1} + .. + A_p f_{t-p} + eta_t (device tensors).
        Avar [B,r,r p] = [A_1 .. A_p], Q [B,r,r], mu0 [B,r p], P0 [B,r p,r p].  Returns (f_smooth, P_smooth, loglik).
        singular_q (here and in the other VAR(p) / AR entry points): the r x r block Q itself may be rank deficient -- DFM_F_SINGULAR_Q,
        the kernels that never invert it (include/dfm_hip.h)."""
        torch = self._torch
        B, T, N = panel.shape
        r = Lam.shape[2]
        k = Avar.shape[2]
        p = k // r
        flags = self._flags(panel, may_have_missing, singular_q)
        f = torch.empty((B, T, r), dtype=torch.float64, device=panel.device)
        P = torch.empty((B, T, r * (r + 1) // 2), dtype=torch.float64, device=panel.device) if want_P else None
        ll = torch.empty((B,), dtype=torch.float64, device=panel.device)
        self._sync_stream()
        rc = self._lib.dfm_ks_pass_varp_batch_dev(
            self._h, B, T, N, r, p, self._dev(panel, "panel"), self._dev(Lam, "Lam", (B, N, r)),
            self._dev(R, "R", (B, N)), self._dev(Avar, "Avar", (B, r, r * p)), self._dev(Q, "Q", (B, r, r)),
            self._dev(mu0, "mu0", (B, k)), self._dev(P0, "P0", (B, k, k)), self._dev(f, "f_smooth"),
            self._dev(P, "P_smooth") if P is not None else None, self._dev(ll, "loglik"), flags)
        _check(self._h, rc)
        return f, P, ll

    def em_varp_batch(self, panel, Lam, R, Avar, Q, mu0, P0, max_iter: int = 10, tol: float = 0.0,
                      want_smooth: bool = True, want_P: bool = True, may_have_missing: Optional[bool] = None, singular_q: bool = False):
        """EM for the VAR(p) model, parameters (device tensors) updated in place.
        Returns (loglik_path [B,max_iter], iters [B] int32, f_smooth, P_smooth)."""
        torch = self._torch
        B, T, N = panel.shape
        r = Lam.shape[2]
        k = Avar.shape[2]
        p = k // r
        flags = self._flags(panel, may_have_missing, singular_q)
        dev = panel.device
        path = torch.empty((B, max_iter), dtype=torch.float64, device=dev)
        iters = torch.empty((B,), dtype=torch.int32, device=dev)
        f = torch.empty((B, T, r), dtype=torch.float64, device=dev) if want_smooth else None
        P = torch.empty((B, T, r * (r + 1) // 2), dtype=torch.float64, device=dev) if (want_smooth and want_P) else None
        self._sync_stream()
        rc = self._lib.dfm_em_varp_batch_dev(
            self._h, B, T, N, r, p, self._dev(panel, "panel"), self._dev(Lam, "Lam", (B, N, r)),
            self._dev(R, "R", (B, N)), self._dev(Avar, "Avar", (B, r, r * p)), self._dev(Q, "Q", (B, r, r)),
            self._dev(mu0, "mu0", (B, k)), self._dev(P0, "P0", (B, k, k)), int(max_iter), float(tol),
            self._dev(path, "loglik_path"), ctypes.c_void_p(iters.data_ptr()),
            self._dev(f, "f_smooth") if f is not None else None,
            self._dev(P, "P_smooth") if P is not None else None, flags)
        _check(self._h, rc)
        return path, iters, f, P

    def em_varp_batch_host(self, panel, Lam, R, Avar, Q, mu0, P0, max_iter: int = 10, tol: float = 0.0,
                           may_have_missing: Optional[bool] = None, singular_q: bool = False):
        """Host-pointer entry (what Julia's ccall binds).  Returns (params dict, loglik_path, iters, f_smooth,
        P_smooth); inputs are not modified."""
        c = lambda a: np.array(a, dtype=np.float64, order="C", copy=True)
        panel = np.ascontiguousarray(panel, dtype=np.float64)
        Lam, R, Avar, Q, mu0, P0 = map(c, (Lam, R, Avar, Q, mu0, P0))
        B, T, N = panel.shape
        r = Lam.shape[2]
        p_lag = Avar.shape[2] // r
        if may_have_missing is None:
            may_have_missing = bool(np.isnan(panel).any())
        flags = (_lib.DFM_F_MAY_HAVE_MISSING if may_have_missing else 0) | (_lib.DFM_F_SINGULAR_Q if singular_q else 0)
        path = np.empty((B, max_iter)); iters = np.empty(B, dtype=np.int32)
        f = np.empty((B, T, r)); P = np.empty((B, T, r * (r + 1) // 2))
        p = lambda a: ctypes.c_void_p(a.ctypes.data)
        rc = self._lib.dfm_em_varp_batch(self._h, B, T, N, r, p_lag, p(panel), p(Lam), p(R), p(Avar), p(Q), p(mu0),
                                         p(P0), int(max_iter), float(tol), p(path), p(iters), p(f), p(P), flags)
        _check(self._h, rc)
        return dict(Lam=Lam, R=R, Avar=Avar, Q=Q, mu0=mu0, P0=P0), path, iters, f, P

    def ks_pass_varp_batch_host(self, panel, Lam, R, Avar, Q, mu0, P0, may_have_missing: Optional[bool] = None, singular_q: bool = False):
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        panel, Lam, R, Avar, Q, mu0, P0 = map(c, (panel, Lam, R, Avar, Q, mu0, P0))
        B, T, N = panel.shape
        r = Lam.shape[2]
        p_lag = Avar.shape[2] // r
        if may_have_missing is None:
            may_have_missing = bool(np.isnan(panel).any())
        flags = (_lib.DFM_F_MAY_HAVE_MISSING if may_have_missing else 0) | (_lib.DFM_F_SINGULAR_Q if singular_q else 0)
        f = np.empty((B, T, r)); P = np.empty((B, T, r * (r + 1) // 2)); ll = np.empty(B)
        p = lambda a: ctypes.c_void_p(a.ctypes.data)
        rc = self._lib.dfm_ks_pass_varp_batch(self._h, B, T, N, r, p_lag, p(panel), p(Lam), p(R), p(Avar), p(Q), p(mu0),
                                              p(P0), p(f), p(P), p(ll), flags)
        _check(self._h, rc)
        return f, P, ll

    # ------------------------------------------------------------------ AR idiosyncratic terms (quasi-differencing)
    def ks_pass_ar_batch(self, panel, Lam, sig2, rho, Avar, Q, mu0, P0, want_P: bool = True,
                         may_have_missing: Optional[bool] = None, singular_q: bool = False):
        """Smoother pass with AR(q) idiosyncratic terms (rho [B,N,q], sig2 [B,N]: the reference's uar_coef, uar_ser^2)
        and VAR(p) factors (Avar [B,r,r p]); mu0 [B,r m], P0 [B,r m,r m], m = max(p, q+1).  Device tensors.
        Returns (f_smooth [B,T-q,r], P_smooth or None, loglik [B]) for rows q+1..T."""
        torch = self._torch
        B, T, N = panel.shape
        r = Lam.shape[2]
        p = Avar.shape[2] // r
        q = rho.shape[2]
        k = r * max(p, q + 1)
        flags = self._flags(panel, may_have_missing, singular_q)
        f = torch.empty((B, T - q, r), dtype=torch.float64, device=panel.device)
        P = torch.empty((B, T - q, r * (r + 1) // 2), dtype=torch.float64, device=panel.device) if want_P else None
        ll = torch.empty((B,), dtype=torch.float64, device=panel.device)
        self._sync_stream()
        rc = self._lib.dfm_ks_pass_ar_batch_dev(
            self._h, B, T, N, r, p, q, self._dev(panel, "panel"), self._dev(Lam, "Lam", (B, N, r)),
            self._dev(sig2, "sig2", (B, N)), self._dev(rho, "rho", (B, N, q)) if q else None,
            self._dev(Avar, "Avar", (B, r, r * p)), self._dev(Q, "Q", (B, r, r)), self._dev(mu0, "mu0", (B, k)),
            self._dev(P0, "P0", (B, k, k)), self._dev(f, "f_smooth"), self._dev(P, "P_smooth") if P is not None else None,
            self._dev(ll, "loglik"), flags)
        _check(self._h, rc)
        return f, P, ll

    def ks_pass_ar_batch_host(self, panel, Lam, sig2, rho, Avar, Q, mu0, P0, may_have_missing: Optional[bool] = None, singular_q: bool = False):
        """Host-pointer entry (what Julia's ccall binds)."""
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        panel, Lam, sig2, rho, Avar, Q, mu0, P0 = map(c, (panel, Lam, sig2, rho, Avar, Q, mu0, P0))
        B, T, N = panel.shape
        r = Lam.shape[2]
        p_lag = Avar.shape[2] // r
        q = rho.shape[2]
        if may_have_missing is None:
            may_have_missing = bool(np.isnan(panel).any())
        flags = (_lib.DFM_F_MAY_HAVE_MISSING if may_have_missing else 0) | (_lib.DFM_F_SINGULAR_Q if singular_q else 0)
        f = np.empty((B, T - q, r)); P = np.empty((B, T - q, r * (r + 1) // 2)); ll = np.empty(B)
        p = lambda a: ctypes.c_void_p(a.ctypes.data) if a.size else None
        rc = self._lib.dfm_ks_pass_ar_batch(self._h, B, T, N, r, p_lag, q, p(panel), p(Lam), p(sig2), p(rho), p(Avar), p(Q),
                                            p(mu0), p(P0), p(f), p(P), p(ll), flags)
        _check(self._h, rc)
        return f, P, ll

    def em_ar_batch(self, panel, Lam, sig2, rho, Avar, Q, mu0, P0, max_iter: int = 10, tol: float = 0.0,
                    want_smooth: bool = True, want_P: bool = True, may_have_missing: Optional[bool] = None, singular_q: bool = False):
        """Joint ECM estimation with AR(q) idiosyncratic terms (include/dfm_hip.h: dfm_em_ar_batch_dev).  Device tensors;
        Lam [B,N,r], sig2 [B,N], rho [B,N,q], Avar [B,r,r p], Q, mu0 [B,r m], P0 [B,r m,r m] are UPDATED IN PLACE.
        Returns (loglik_path [B,max_iter], iters [B], f_smooth [B,T-q,r] or None, P_smooth or None)."""
        torch = self._torch
        B, T, N = panel.shape
        r = Lam.shape[2]
        p = Avar.shape[2] // r
        q = rho.shape[2]
        k = r * max(p, q + 1)
        flags = self._flags(panel, may_have_missing, singular_q)
        dev = panel.device
        path = torch.empty((B, max_iter), dtype=torch.float64, device=dev)
        iters = torch.empty((B,), dtype=torch.int32, device=dev)
        f = torch.empty((B, T - q, r), dtype=torch.float64, device=dev) if want_smooth else None
        P = torch.empty((B, T - q, r * (r + 1) // 2), dtype=torch.float64, device=dev) if (want_smooth and want_P) else None
        self._sync_stream()
        rc = self._lib.dfm_em_ar_batch_dev(
            self._h, B, T, N, r, p, q, self._dev(panel, "panel"), self._dev(Lam, "Lam", (B, N, r)),
            self._dev(sig2, "sig2", (B, N)), self._dev(rho, "rho", (B, N, q)) if q else None,
            self._dev(Avar, "Avar", (B, r, r * p)), self._dev(Q, "Q", (B, r, r)), self._dev(mu0, "mu0", (B, k)),
            self._dev(P0, "P0", (B, k, k)), int(max_iter), float(tol), self._dev(path, "loglik_path"),
            ctypes.c_void_p(iters.data_ptr()), self._dev(f, "f_smooth") if f is not None else None,
            self._dev(P, "P_smooth") if P is not None else None, flags)
        _check(self._h, rc)
        return path, iters, f, P

    def em_ar_batch_host(self, panel, Lam, sig2, rho, Avar, Q, mu0, P0, max_iter: int = 10, tol: float = 0.0,
                         may_have_missing: Optional[bool] = None, singular_q: bool = False):
        """Host-pointer entry (what Julia's ccall binds).  Returns (params dict, loglik_path, iters, f_smooth, P_smooth);
        inputs are not modified."""
        c = lambda a: np.array(a, dtype=np.float64, order="C", copy=True)
        panel = np.ascontiguousarray(panel, dtype=np.float64)
        Lam, sig2, rho, Avar, Q, mu0, P0 = map(c, (Lam, sig2, rho, Avar, Q, mu0, P0))
        B, T, N = panel.shape
        r = Lam.shape[2]
        p_lag = Avar.shape[2] // r
        q = rho.shape[2]
        if may_have_missing is None:
            may_have_missing = bool(np.isnan(panel).any())
        flags = (_lib.DFM_F_MAY_HAVE_MISSING if may_have_missing else 0) | (_lib.DFM_F_SINGULAR_Q if singular_q else 0)
        path = np.empty((B, max_iter)); iters = np.empty(B, dtype=np.int32)
        f = np.empty((B, T - q, r)); P = np.empty((B, T - q, r * (r + 1) // 2))
        p = lambda a: ctypes.c_void_p(a.ctypes.data) if a.size else None
        rc = self._lib.dfm_em_ar_batch(self._h, B, T, N, r, p_lag, q, p(panel), p(Lam), p(sig2), p(rho), p(Avar), p(Q),
                                       p(mu0), p(P0), int(max_iter), float(tol), p(path), p(iters), p(f), p(P), flags)
        _check(self._h, rc)
        return dict(Lam=Lam, sig2=sig2, rho=rho, Avar=Avar, Q=Q, mu0=mu0, P0=P0), path, iters, f, P

    # ------------------------------------------------------------------ PCA initialisation / synthetic panels
    def pca_init_batch(self, panel, r: int, want_factors: bool = True):
        """PCA + OLS start of EM on balanced standardised panels (device tensor [B,T,N], no NaN).
        Returns (Lam, R, A, Q, mu0, P0, factors or None) as device tensors.  Asynchronous."""
        torch = self._torch
        B, T, N = panel.shape
        f64 = dict(dtype=torch.float64, device=panel.device)
        Lam = torch.empty((B, N, r), **f64); R = torch.empty((B, N), **f64)
        A = torch.empty((B, r, r), **f64); Q = torch.empty((B, r, r), **f64)
        mu0 = torch.empty((B, r), **f64); P0 = torch.empty((B, r, r), **f64)
        F = torch.empty((B, T, r), **f64) if want_factors else None
        self._sync_stream()
        rc = self._lib.dfm_pca_init_batch_dev(
            self._h, B, T, N, int(r), self._dev(panel, "panel"), self._dev(Lam, "Lam"), self._dev(R, "R"),
            self._dev(A, "A"), self._dev(Q, "Q"), self._dev(mu0, "mu0"), self._dev(P0, "P0"),
            self._dev(F, "factors") if F is not None else None)
        _check(self._h, rc)
        return Lam, R, A, Q, mu0, P0, F

    def pca_init_batch_host(self, panel, r: int):
        """Host-pointer PCA entry (what Julia's ccall binds): NumPy in / out."""
        panel = np.ascontiguousarray(panel, dtype=np.float64)
        B, T, N = panel.shape
        Lam = np.empty((B, N, r)); R = np.empty((B, N)); A = np.empty((B, r, r)); Q = np.empty((B, r, r))
        mu0 = np.empty((B, r)); P0 = np.empty((B, r, r)); F = np.empty((B, T, r))
        p = lambda a: ctypes.c_void_p(a.ctypes.data)
        rc = self._lib.dfm_pca_init_batch(self._h, B, T, N, int(r), p(panel), p(Lam), p(R), p(A), p(Q), p(mu0),
                                          p(P0), p(F))
        _check(self._h, rc)
        return dict(Lam=Lam, R=R, A=A, Q=Q, mu0=mu0, P0=P0), F

    def synth_panels(self, seed: int, first_replicate: int, B: int, T: int, N: int, r: int,
                     missing_prob: float = 0.0):
        """Synthetic replicates of the SURVEY §8(d) DGP generated on the device.  Returns
        (panel [B,T,N], (Lam, R, A, Q, mu0, P0)) -- the DGP parameters rescaled to the standardised panel."""
        torch = self._torch
        dev = torch.device("cuda", self.device)
        f64 = dict(dtype=torch.float64, device=dev)
        panel = torch.empty((B, T, N), **f64)
        Lam = torch.empty((B, N, r), **f64); R = torch.empty((B, N), **f64)
        A = torch.empty((B, r, r), **f64); Q = torch.empty((B, r, r), **f64)
        mu0 = torch.empty((B, r), **f64); P0 = torch.empty((B, r, r), **f64)
        self._sync_stream()
        rc = self._lib.dfm_synth_panels_dev(
            self._h, ctypes.c_uint64(seed), ctypes.c_int64(first_replicate), B, T, N, r,
            ctypes.c_double(missing_prob), self._dev(panel, "panel"), self._dev(Lam, "Lam"), self._dev(R, "R"),
            self._dev(A, "A"), self._dev(Q, "Q"), self._dev(mu0, "mu0"), self._dev(P0, "P0"))
        _check(self._h, rc)
        return panel, (Lam, R, A, Q, mu0, P0)

    # ------------------------------------------------------------------ non-parametric estimator (als.hip)
    def als_batch_host(self, z, F0, r_each=None, nt_min: int = 20, max_iter: int = 10 ** 8, tol: float = 1e-8,
                       path_cap: int = 0, want_R2: bool = False, shared_panel: Optional[bool] = None):
        """Batched `estimate_factor!` sweeps (dfm_functions.ipynb:352-370) through the host-pointer entry.

        z: [T,N] (one panel shared by every run) or [B,T,N]; F0: [B,T,r] starting factors; r_each: [B] ints
        or None.  Returns dict(F [B,T,r], Lam [B,N,r], iters [B], ssr [B], ssr_path [B,path_cap] or None,
        R2 [B,N] or None)."""
        z = np.ascontiguousarray(z, dtype=np.float64)
        F = np.array(F0, dtype=np.float64, order="C", copy=True)
        B, T, r = F.shape
        if shared_panel is None:
            shared_panel = z.ndim == 2
        N = z.shape[-1]
        if z.shape[-2] != T or (not shared_panel and z.shape[0] != B):
            raise ValueError("z and F0 disagree on B or T")
        stride = 0 if shared_panel else T * N
        Lam = np.empty((B, N, r)); iters = np.empty(B, dtype=np.int32); ssr = np.empty(B)
        path = np.empty((B, path_cap)) if path_cap > 0 else None
        R2 = np.empty((B, N)) if want_R2 else None
        re = None if r_each is None else np.ascontiguousarray(r_each, dtype=np.int32)
        if re is not None and (re.shape != (B,) or re.min() < 1 or re.max() > r):
            raise ValueError("r_each must hold B values in 1..r")
        p = lambda a: None if a is None else ctypes.c_void_p(a.ctypes.data)
        rc = self._lib.dfm_als_batch(self._h, B, T, N, r, p(z), stride, p(re), p(F), p(Lam), int(nt_min),
                                     int(min(max_iter, 2 ** 31 - 1)), float(tol), p(path), int(path_cap), p(iters),
                                     p(ssr), p(R2))
        _check(self._h, rc)
        return dict(F=F, Lam=Lam, iters=iters, ssr=ssr, ssr_path=path, R2=R2)

    def ols_batch_host(self, X, Y, nt_min: int = 0, shared_X: Optional[bool] = None, want_resid: bool = True):
        """Batched complete-case OLS (`ols_skipmissing`, dfm_functions.ipynb:242-252) through the host entry.

        X: [T,K] (shared regressors) or [P,T,K]; Y: [T,P] -- one problem per COLUMN (the reference's data
        layout) -- with NaN for missing.  Returns dict(beta [P,K], resid [T,P] or None, ssr, tss, nobs [P])."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        Y = np.ascontiguousarray(Y, dtype=np.float64)
        T, P = Y.shape
        if shared_X is None:
            shared_X = X.ndim == 2
        K = X.shape[-1]
        if X.shape[-2] != T or (not shared_X and X.shape[0] != P):
            raise ValueError("X and Y disagree on T or P")
        beta = np.empty((P, K)); resid = np.empty((P, T)) if want_resid else None
        ssr = np.empty(P); tss = np.empty(P); nobs = np.empty(P, dtype=np.int32)
        p = lambda a: None if a is None else ctypes.c_void_p(a.ctypes.data)
        rc = self._lib.dfm_ols_batch(self._h, P, T, K, p(X), 0 if shared_X else T * K, p(Y), 1, P, int(nt_min),
                                     p(beta), p(resid), p(ssr), p(tss), p(nobs))
        _check(self._h, rc)
        return dict(beta=beta, resid=None if resid is None else resid.T.copy(), ssr=ssr, tss=tss, nobs=nobs)

    # ------------------------------------------------------------------ wild-bootstrap IRF bands (boot.hip)
    def var_bootstrap_irf_host(self, y, betahat, resid, p: int, H: int, ndraws: int, signs=None, seed: int = 0,
                               want_beta: bool = False, first_draw: int = 0):
        """B recursive-design wild-bootstrap draws of VAR(p) -> Cholesky -> impulse responses.
        y, resid: [T,ns] over the estimation window; betahat: [1 + ns p, ns].  Returns irf [B,ns,H,ns]
        (variable, horizon, shock) and, if asked, the re-estimated coefficients [B,1+ns p,ns]."""
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        y, betahat = c(y), c(betahat)
        resid = c(np.nan_to_num(resid))
        T, ns = y.shape
        B = int(ndraws)
        sg = None if signs is None else c(signs)
        if sg is not None and sg.shape != (B, T):
            raise ValueError("signs must be [ndraws, T]")
        irf = np.empty((B, ns, H, ns)); bo = np.empty((B, 1 + ns * p, ns)) if want_beta else None
        ptr = lambda a: None if a is None else ctypes.c_void_p(a.ctypes.data)
        rc = self._lib.dfm_var_bootstrap_irf(self._h, B, T, ns, int(p), int(H), ptr(y), ptr(betahat), ptr(resid),
                                             ptr(sg), ctypes.c_uint64(seed), ctypes.c_int64(first_draw), ptr(bo), ptr(irf))
        _check(self._h, rc)
        return (irf, bo) if want_beta else irf

    def quantile_bands_host(self, x, q):
        """Nearest-rank quantiles over the first axis: x [B, ...] -> [len(q), ...]."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        q = np.ascontiguousarray(q, dtype=np.float64)
        B = x.shape[0]
        S = int(np.prod(x.shape[1:]))
        out = np.empty((q.size, S))
        ptr = lambda a: ctypes.c_void_p(a.ctypes.data)
        rc = self._lib.dfm_quantile_bands(self._h, B, S, int(q.size), ptr(x), ptr(q), ptr(out))
        _check(self._h, rc)
        return out.reshape((q.size,) + x.shape[1:])

    # ------------------------------------------------------------------ Chow / QLR with HAC covariance (breaks.hip)
    def chow_batch_host(self, ys, Xs, prob_series, prob_break, prob_q):
        """P Chow statistics with HAC covariance (`compute_chow`, dfm_functions.ipynb:891-902).  ys: list of S
        complete-case vectors, Xs: list of S matrices (T_s x k); problem p = (series, break date, bandwidth)."""
        S = len(ys)
        k = Xs[0].shape[1]
        Tlen = np.array([len(v) for v in ys], dtype=np.int32)
        Tmax = int(Tlen.max())
        y = np.zeros((S, Tmax)); X = np.zeros((S, Tmax, k))
        for s in range(S):
            y[s, :Tlen[s]] = ys[s]; X[s, :Tlen[s]] = Xs[s]
        ps = np.ascontiguousarray(prob_series, dtype=np.int32); pb = np.ascontiguousarray(prob_break, dtype=np.int32)
        pq = np.ascontiguousarray(prob_q, dtype=np.int32)
        P = ps.size
        out = np.empty(P)
        ptr = lambda a: ctypes.c_void_p(a.ctypes.data)
        rc = self._lib.dfm_chow_batch(self._h, S, Tmax, k, ptr(y), ptr(X), ptr(Tlen), P, ptr(ps), ptr(pb), ptr(pq), ptr(out))
        _check(self._h, rc)
        return out

    def standardize_batch(self, panel, want_stats: bool = True):
        """`standardize_data` (dfm_functions.ipynb:501-509) of a device tensor [B,T,N] IN PLACE; returns (mean, sd)
        [B,N] device tensors (or None)."""
        torch = self._torch
        B, T, N = panel.shape
        mu = torch.empty((B, N), dtype=torch.float64, device=panel.device) if want_stats else None
        sd = torch.empty((B, N), dtype=torch.float64, device=panel.device) if want_stats else None
        self._sync_stream()
        rc = self._lib.dfm_standardize_batch_dev(self._h, B, T, N, self._dev(panel, "panel"),
                                                 self._dev(mu, "mean") if want_stats else None,
                                                 self._dev(sd, "sd") if want_stats else None)
        _check(self._h, rc)
        return mu, sd
