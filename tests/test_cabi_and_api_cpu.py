"""CPU-only checks of the boundary: libdfmhip.so loads, exports exactly the symbols include/dfm_hip.h
declares, reports argument errors without a device; and the host-side mirror of the reference interface
(api.py) behaves like the reference's constructors / helpers.  No kernel is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest

from dynamic_factor_models_amd import _lib, api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_prototypes():
    src = open(os.path.join(ROOT, "include", "dfm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dfm_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_table_agree():
    names = _header_prototypes()
    assert names, "no prototypes parsed from include/dfm_hip.h"
    assert sorted(_lib.SYMBOLS) == names


def test_library_exports_every_declared_symbol():
    lib = _lib.load()                       # raises if the .so is missing: there is no fallback
    for name in _header_prototypes():
        assert hasattr(lib, name), name
    assert lib.dfm_version().decode().startswith("dfmhip")


def test_argument_errors_without_a_device():
    import torch
    lib = _lib.load()
    assert lib.dfm_workspace_bytes(0, 10, 10, 2, 0) == 0
    assert lib.dfm_workspace_bytes(4, 10, 10, 40, 0) == 0          # r > DFM_MAX_R
    assert lib.dfm_workspace_bytes(4, 50, 20, 3, 1) > lib.dfm_workspace_bytes(4, 50, 20, 3, 0) > 0
    assert lib.dfm_create(None, 0, None) == -3                     # DFM_E_NULL
    assert lib.dfm_last_error(None) == b"null handle"
    assert lib.dfm_synchronize(None) == -3
    if not torch.cuda.is_available():
        h = ctypes.c_void_p()
        assert lib.dfm_create(ctypes.byref(h), 0, None) == -6      # DFM_E_NO_DEVICE
        assert not h.value


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dynamic_factor_models_amd import DfmContext
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        DfmContext()
    m = api.DFMModel(np.random.default_rng(0).standard_normal((30, 6)), np.ones(6), 5, 5, 1, 30, 0, 2, 1e-8, 4, 4)
    with pytest.raises(RuntimeError):
        api.estimate(m, api.Parametric())


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "dynamic_factor_models_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "dfm_oracle" not in text, f


# ------------------------------------------------------------------ reference-interface mirror
def _data(T=40, ns=7, seed=0):
    x = np.random.default_rng(seed).standard_normal((T, ns))
    x[3, 2] = np.nan
    return x


def test_dfmmodel_constructor_checks_mirror_the_reference():
    x = _data()
    with pytest.raises(ValueError, match="length of inclcode"):        # dfm_functions.ipynb:124
        api.DFMModel(x, np.ones(6), 20, 40, 3, 40, 0, 2, 1e-8, 4, 4)
    with pytest.raises(ValueError, match="initperiod must be smaller"):  # :125
        api.DFMModel(x, np.ones(7), 20, 40, 10, 10, 0, 2, 1e-8, 4, 4)
    with pytest.raises(ValueError, match="must be positive"):          # :126
        api.DFMModel(x, np.ones(7), 20, 40, 3, 40, 0, 2, 1e-8, 0, 4)
    m = api.DFMModel(x, [1, 1, 0, 1, 1, 2, 1], 20, 40, 3, 40, 0, 2, 1e-8, 4, 4)
    assert (m.T, m.ns, m.nfac_t) == (40, 7, 2)
    assert m.fes.T == 38 and m.fes.ns == 5                             # :130-131
    assert m.factor.shape == (40, 2) and np.isnan(m.factor).all()
    assert m.lambda_.shape == (7, 2) and m.uar_coef.shape == (7, 4)
    assert m.factor_var_model.y is m.factor                            # aliasing note :80
    assert m.factor_var_model.M.shape == (8, 8) and m.factor_var_model.G.shape == (8, 2)


def test_standardize_data_is_population_sd_over_observed_cells():
    x = _data()
    z, sd = api.standardize_data(x)
    assert sd.shape == (1, 7)
    col = x[~np.isnan(x[:, 2]), 2]
    np.testing.assert_allclose(sd[0, 2], col.std(ddof=0))
    np.testing.assert_allclose(np.nanmean(z, axis=0), 0.0, atol=1e-14)
    np.testing.assert_allclose(np.nansum(z * z), np.count_nonzero(~np.isnan(x)))    # tss == nobs
    xb, keep = api.drop_missing_col(z)
    assert xb.shape == (40, 6) and not keep[2]


def test_estimate_rejects_what_the_path_does_not_cover():
    m = api.DFMModel(_data(), np.ones(7), 5, 5, 1, 40, 0, 2, 1e-8, 4, 4)
    with pytest.raises(NotImplementedError):
        api.estimate(m, api.NonParametric(), lam_constr_f=np.eye(2))     # loading constraints: not on the HIP path
    with pytest.raises(RuntimeError, match="HIP device"):
        api.estimate(m, api.NonParametric())                             # no CPU fallback
    with pytest.raises(TypeError):
        api.estimate(m, object())
    # observed factors (round 3: supported on the parametric path): the caller must have filled the observed columns of
    # `factor` -- a host-side check that runs before any device work
    m2 = api.DFMModel(_data(), np.ones(7), 5, 5, 1, 40, 1, 2, 1e-8, 4, 4)
    with pytest.raises(ValueError, match="observed factors"):
        api.estimate(m2, api.Parametric())
    m2.factor[:, :1] = 0.5
    with pytest.raises(RuntimeError, match="HIP device"):
        api.estimate(m2, api.Parametric())                               # ... and then there is no CPU fallback
    with pytest.raises(RuntimeError, match="HIP device"):
        api.estimate_factor(m2)                                          # the ALS estimator takes them too (two dfm_ols_batch calls per sweep)


def test_parametric_argument_checks_run_before_any_device_work():
    """factor_lags / state-width validation and the AR-idiosyncratic smoother's preconditions are host logic: they
    must raise their own errors on a machine without a GPU (not 'no HIP device')."""
    m = api.DFMModel(np.random.default_rng(1).standard_normal((40, 7)), np.ones(7), 5, 5, 1, 40, 0, 2, 1e-8, 4, 4)
    with pytest.raises(ValueError, match="factor_lags"):
        api.estimate(m, api.Parametric(), factor_lags=0)
    with pytest.raises(ValueError, match="factor_lags"):
        api.estimate(m, api.Parametric(), factor_lags=17)            # 2 * 17 > 32
    with pytest.raises(ValueError, match="not estimated|no series"):
        api.smooth_factors_ar_idio(m)                                 # nothing estimated yet
    m8 = api.DFMModel(np.random.default_rng(2).standard_normal((60, 12)), np.ones(12), 5, 5, 1, 60, 0, 8, 1e-8, 4, 4)
    with pytest.raises(ValueError, match="must not exceed 32"):
        api.smooth_factors_ar_idio(m8)                                # 8 * max(4, 5) = 40


def test_impulse_response_methods_mirror_the_reference():
    """dfm_functions.ipynb:793-825: a vector of shocks -> [ny, T, k]; ONE shock -> the [ny, T] matrix (the reference's method is
    broken, :817-821; repaired here and in julia/dfm_hip.jl); 'all' -> every column of G."""
    rng = np.random.default_rng(3)
    v = api._var_model(rng.standard_normal((30, 3)), nlag=2)
    v.M = 0.3 * rng.standard_normal((6, 6)); v.Q = rng.standard_normal((3, 6)); v.G = rng.standard_normal((6, 3))
    full = api.impulse_response(v, [0, 1, 2], 7)
    assert full.shape == (3, 7, 3)
    for s in range(3):
        ref = np.stack([v.Q @ np.linalg.matrix_power(v.M, t) @ v.G[:, s] for t in range(7)], axis=1)
        np.testing.assert_allclose(full[:, :, s], ref, rtol=1e-12, atol=1e-14)
        one = api.impulse_response(v, s, 7)
        assert one.shape == (3, 7) and np.array_equal(one, full[:, :, s])
        assert np.array_equal(api.impulse_response(v, np.int64(s), 7), one)
    assert np.array_equal(api.impulse_response(v, "all", 7), full)
    assert np.array_equal(api.impulse_response(v, [2], 7)[:, :, 0], full[:, :, 2])
    with pytest.raises(IndexError):
        api.impulse_response(v, 3, 7)
    with pytest.raises(ValueError):
        api.impulse_response(v, "some", 7)
