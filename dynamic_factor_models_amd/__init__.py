"""dynamic_factor_models_amd -- MI355X (gfx950) native batched Kalman filter / RTS smoother / EM
for the dynamic factor model of QuantEcon/dynamic_factor_models, behind a C-ABI shared library
(include/dfm_hip.h) and a host-side mirror of the reference's Julia API (api.py)."""
from ._lib import DFM_F_MAY_HAVE_MISSING, DFM_MAX_R, DfmError, SO_PATH  # noqa: F401

__all__ = ["DfmContext", "DfmMulti", "DfmError", "SO_PATH"]


def __getattr__(name):
    if name == "DfmContext":
        from .kalman import DfmContext
        return DfmContext
    if name == "DfmMulti":
        from .kalman import DfmMulti
        return DfmMulti
    raise AttributeError(name)
