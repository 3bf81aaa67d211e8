import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def diag_only(reason="switch of the diagnostics build (csrc/dfm_kernels.h diag_env): run with DFM_LIB=diag"):
    """Mark for tests / parameters that steer the library with a switch only the DIAGNOSTICS build reads (python -m
    dynamic_factor_models_amd.build --diag; DFM_LIB=diag): the production library ignores the switch, so the case would only repeat
    the default route."""
    return pytest.mark.skipif(os.environ.get("DFM_LIB") != "diag", reason=reason)


@pytest.fixture(scope="session")
def repo_root():
    return ROOT
