import os
import sys

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        from pywfa_amd import _native
        return _native.lib().wfa_hip_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must not silently pass without the native path: fail loudly instead of skipping."""
    from pywfa_amd import _native
    n = _native.lib().wfa_hip_device_count()
    assert n > 0, "no HIP device visible: -m gpu tests need a GPU (" + _native.lib().wfa_hip_global_error().decode() + ")"
    return n
