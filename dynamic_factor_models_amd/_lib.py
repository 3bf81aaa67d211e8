"""ctypes binding of libdfmhip.so -- exactly the symbols include/dfm_hip.h declares.

There is NO fallback: if the shared library is missing or a symbol is absent this module raises.
"""
from __future__ import annotations

import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, "lib", "libdfmhip.so")
# The diagnostics build (`python -m dynamic_factor_models_amd.build --diag`: -DDFM_DIAG, reads the ablation / stamp / *_OLD switches,
# csrc/dfm_kernels.h diag_env) is a separate file that only scripts/dbg ask for: DFM_LIB=diag.  The product never loads it.
# (scripts/dbg also pass the path of a development build: in-kernel span stamps, the previous round's library for an A/B)
if os.environ.get("DFM_LIB") == "diag":
    SO_PATH = os.path.join(HERE, "lib", "libdfmhip_diag.so")
elif os.environ.get("DFM_LIB"):
    SO_PATH = os.path.abspath(os.environ["DFM_LIB"])

c_dp = ctypes.POINTER(ctypes.c_double)
c_ip = ctypes.POINTER(ctypes.c_int)
c_vp = ctypes.c_void_p
c_int = ctypes.c_int
c_uint = ctypes.c_uint

DFM_F_MAY_HAVE_MISSING = 1
DFM_F_SINGULAR_Q = 2
DFM_MAX_R = 32
DFM_MULTI_F_FORCE_COMM = 1
# dfm_multi_fetch `what` (enum in include/dfm_hip.h)
(DFM_MULTI_LAM, DFM_MULTI_R, DFM_MULTI_A, DFM_MULTI_Q, DFM_MULTI_MU0, DFM_MULTI_P0, DFM_MULTI_F_SMOOTH, DFM_MULTI_P_SMOOTH,
 DFM_MULTI_LOGLIK, DFM_MULTI_LOGLIK_PATH, DFM_MULTI_ITERS, DFM_MULTI_PANEL) = range(12)
ERRORS = {-1: "DFM_E_DIMS", -2: "DFM_E_R_UNSUPPORTED", -3: "DFM_E_NULL", -4: "DFM_E_MISSING",
          -5: "DFM_E_NUMERIC", -6: "DFM_E_NO_DEVICE", -7: "DFM_E_COMM"}

_PASS_ARGS = [c_vp, c_int, c_int, c_int, c_int] + [c_vp] * 10 + [c_uint]
_EMSTEP_ARGS = [c_vp, c_int, c_int, c_int, c_int] + [c_vp] * 8 + [c_uint]
_EM_ARGS = [c_vp, c_int, c_int, c_int, c_int] + [c_vp] * 7 + [c_int, ctypes.c_double] + [c_vp] * 4 + [c_uint]
_VPASS_ARGS = [c_vp, c_int, c_int, c_int, c_int, c_int] + [c_vp] * 10 + [c_uint]
_VEM_ARGS = [c_vp, c_int, c_int, c_int, c_int, c_int] + [c_vp] * 7 + [c_int, ctypes.c_double] + [c_vp] * 4 + [c_uint]
_ARPASS_ARGS = [c_vp, c_int, c_int, c_int, c_int, c_int, c_int] + [c_vp] * 11 + [c_uint]
_AREM_ARGS = [c_vp, c_int, c_int, c_int, c_int, c_int, c_int] + [c_vp] * 8 + [c_int, ctypes.c_double] + [c_vp] * 4 + [c_uint]
_OBSEM_ARGS = [c_vp, c_int, c_int, c_int, c_int, c_int] + [c_vp] * 8 + [c_int, ctypes.c_double] + [c_vp] * 4 + [c_uint]
_PCA_ARGS = [c_vp, c_int, c_int, c_int, c_int] + [c_vp] * 8
c_ll = ctypes.c_longlong
_ALS_ARGS = [c_vp, c_int, c_int, c_int, c_int, c_vp, c_ll, c_vp, c_vp, c_vp, c_int, c_int, ctypes.c_double,
             c_vp, c_int, c_vp, c_vp, c_vp]
_OLS_ARGS = [c_vp, c_int, c_int, c_int, c_vp, c_ll, c_vp, c_ll, c_ll, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]

# name -> (restype, argtypes); every prototype of include/dfm_hip.h
SYMBOLS = {
    "dfm_create": (c_int, [ctypes.POINTER(c_vp), c_int, c_vp]),
    "dfm_destroy": (c_int, [c_vp]),
    "dfm_set_stream": (c_int, [c_vp, c_vp]),
    "dfm_synchronize": (c_int, [c_vp]),
    "dfm_check_status": (c_int, [c_vp]),
    "dfm_chunk_fallbacks": (c_int, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "dfm_last_error": (ctypes.c_char_p, [c_vp]),
    "dfm_version": (ctypes.c_char_p, []),
    "dfm_profile_enable": (c_int, [c_vp, c_int]),
    "dfm_profile_read": (c_int, [c_vp, c_int, ctypes.c_char_p, c_int, c_dp, c_ip]),
    "dfm_hbm_probe": (c_int, [c_vp, ctypes.c_size_t, c_int, c_int, c_dp, c_dp]),
    "dfm_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int, c_int, c_uint]),
    "dfm_ks_pass_batch_dev": (c_int, _PASS_ARGS),
    "dfm_ks_pass_batch": (c_int, _PASS_ARGS),
    "dfm_em_step_batch_dev": (c_int, _EMSTEP_ARGS),
    "dfm_em_batch_dev": (c_int, _EM_ARGS),
    "dfm_em_batch": (c_int, _EM_ARGS),
    "dfm_em_iterate_batch_dev": (c_int, [c_vp, c_int, c_int, c_int, c_int] + [c_vp] * 7 + [c_int, c_int, ctypes.c_double]
                                 + [c_vp] * 5 + [c_uint]),
    "dfm_multi_create": (c_int, [ctypes.POINTER(c_vp), c_int, c_vp, c_uint, ctypes.c_char_p, c_int]),
    "dfm_multi_destroy": (c_int, [c_vp]),
    "dfm_multi_ngpu": (c_int, [c_vp]),
    "dfm_multi_has_comm": (c_int, [c_vp]),
    "dfm_multi_last_error": (ctypes.c_char_p, [c_vp]),
    "dfm_multi_load": (c_int, [c_vp, c_int, c_int, c_int, c_int] + [c_vp] * 7),
    "dfm_multi_synth": (c_int, [c_vp, ctypes.c_uint64, ctypes.c_int64, c_int, c_int, c_int, c_int, ctypes.c_double, c_int]),
    "dfm_multi_ks_pass": (c_int, [c_vp, c_int, c_uint]),
    "dfm_multi_em": (c_int, [c_vp, c_int, ctypes.c_double, c_int, c_int, c_uint, c_vp]),
    "dfm_multi_fetch": (c_int, [c_vp, c_int, c_vp]),
    "dfm_em_batch_multi": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_int] + [c_vp] * 7 + [c_int, ctypes.c_double]
                           + [c_vp] * 4 + [c_uint, c_vp, ctypes.c_char_p, c_int]),
    "dfm_ks_pass_batch_multi": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_int] + [c_vp] * 10
                                + [c_uint, ctypes.c_char_p, c_int]),
    "dfm_ks_pass_varp_batch_dev": (c_int, _VPASS_ARGS),
    "dfm_ks_pass_varp_batch": (c_int, _VPASS_ARGS),
    "dfm_em_varp_batch_dev": (c_int, _VEM_ARGS),
    "dfm_em_varp_batch": (c_int, _VEM_ARGS),
    "dfm_ks_pass_ar_batch_dev": (c_int, _ARPASS_ARGS),
    "dfm_ks_pass_ar_batch": (c_int, _ARPASS_ARGS),
    "dfm_em_ar_batch_dev": (c_int, _AREM_ARGS),
    "dfm_em_ar_batch": (c_int, _AREM_ARGS),
    "dfm_em_obs_batch_dev": (c_int, _OBSEM_ARGS),
    "dfm_em_obs_batch": (c_int, _OBSEM_ARGS),
    "dfm_pca_init_batch_dev": (c_int, _PCA_ARGS),
    "dfm_pca_init_batch": (c_int, _PCA_ARGS),
    "dfm_standardize_batch_dev": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "dfm_als_batch_dev": (c_int, _ALS_ARGS),
    "dfm_als_batch": (c_int, _ALS_ARGS),
    "dfm_ols_batch_dev": (c_int, _OLS_ARGS),
    "dfm_ols_batch": (c_int, _OLS_ARGS),
    "dfm_var_bootstrap_irf_dev": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp,
                                          ctypes.c_uint64, ctypes.c_int64, c_vp, c_vp]),
    "dfm_var_bootstrap_irf": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp,
                                      ctypes.c_uint64, ctypes.c_int64, c_vp, c_vp]),
    "dfm_quantile_bands_dev": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "dfm_quantile_bands": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "dfm_chow_batch_dev": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp]),
    "dfm_chow_batch": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp]),
    "dfm_synth_panels_dev": (c_int, [c_vp, ctypes.c_uint64, ctypes.c_int64, c_int, c_int, c_int, c_int,
                                     ctypes.c_double] + [c_vp] * 7),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load libdfmhip.so and bind every symbol.  Raises (never falls back) when it cannot."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's).  The HIP runtime that is
    # loaded FIRST becomes the process-wide one, and device pointers / streams are only meaningful
    # inside one runtime -- so torch must be imported before libdfmhip.so pulls in a runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            f"{SO_PATH} not found: build it with `python -m dynamic_factor_models_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for this path.")
    lib = ctypes.CDLL(SO_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class DfmError(RuntimeError):
    def __init__(self, code: int, text: str):
        self.code = code
        name = ERRORS.get(code, f"hipError_t {code}" if code > 0 else str(code))
        super().__init__(f"libdfmhip: {name}: {text}")
