import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        return False


@pytest.fixture(scope="session")
def gpu_available():
    return _have_gpu()


def pytest_collection_modifyitems(config, items):
    # GPU tests never silently pass on a box without a GPU: they are skipped unless selected with -m gpu,
    # and when selected they fail loudly if the device or the HIP library is missing.
    selected = "gpu" in (config.getoption("-m") or "")
    if selected:
        return
    skip = pytest.mark.skip(reason="GPU test (select with -m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
