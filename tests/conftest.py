import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
# the tests drive layout / variant / fault switches of the library: those are development switches, read only with the
# master switch set (highs_amd/csrc/pdlp_env.hpp); child processes (mesh workers, the reference CLI) inherit it
os.environ.setdefault("PDLP_MI355X_DEV", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a multi-million-iteration solve (a minute on the device); deselect with -m 'gpu and not slow'")


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
    """A plain `pytest` on a box without a GPU skips the gpu-marked tests instead of failing them one by one.  An
    EXPLICIT `-m gpu` selection (what the driver runs on the GPU box) or PDLP_REQUIRE_GPU=1 never skips: a box with a
    broken driver or an unbuilt library must not report green with zero GPU tests run."""
    reason = _gpu_unavailable_reason()
    if reason is None:
        return
    expr = (config.getoption("-m") or "").replace(" ", "")
    if os.environ.get("PDLP_REQUIRE_GPU") == "1" or (expr and "gpu" in expr and "notgpu" not in expr):
        return
    skip = pytest.mark.skip(reason="needs an MI355X: " + reason)
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def dropin_build():
    """integration/_build (the reference's libhighs with this library's TUs swapped in, its CLI and Catch2 cases):
    made by `make -C integration` (what __graft_entry__.build() runs wherever the reference tree exists) and shipped to
    the GPU box with the snapshot.  Where the reference tree is present and the build is not, that is a FAILURE, not a
    skip; only a box that has neither (nothing to build from, nothing shipped) skips."""
    build = os.path.join(ROOT, "integration", "_build")
    if not os.path.exists(os.path.join(build, "libhighs.so.1")):
        if os.path.isdir("/root/reference/highs") or os.environ.get("PDLP_REQUIRE_DROPIN") == "1":
            pytest.fail("integration/_build is missing although the reference tree is present: run `make -C integration` "
                        "(or __graft_entry__.build())")
        pytest.skip("integration/_build not present and no reference tree to build it from")
    return build


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
