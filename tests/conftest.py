import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_unavailable_reason():
    """None when a HIP device and the built product library are both present."""
    lib = os.path.join(ROOT, "highs_amd", "lib", "libpdlp_mi355x.so")
    if not os.path.exists(lib):
        return "highs_amd/lib/libpdlp_mi355x.so is not built"
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        if hip.hipGetDeviceCount(ctypes.byref(n)) != 0 or n.value <= 0:
            return "no HIP device visible"
    except OSError:
        return "libamdhip64.so not loadable"
    return None


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without a GPU skips the gpu-marked tests instead of failing them one by one
    (an explicit `-m gpu` run on such a box still skips loudly: nothing passes silently)."""
    reason = _gpu_unavailable_reason()
    if reason is None:
        return
    skip = pytest.mark.skip(reason="needs an MI355X: " + reason)
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
