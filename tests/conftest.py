import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """Build the in-tree libraries when they are missing (fresh checkout): the product library
    cross-compiles without a GPU; on the GPU box the prebuilt .so files travel with the snapshot."""
    import subprocess
    lib_so = os.path.join(ROOT, "roadsurf_amd", "lib", "libroadsurf_hip.so")
    if not os.path.exists(lib_so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "roadsurf_amd")], stdout=subprocess.DEVNULL)
    chk = os.path.join(ROOT, "roadsurf_amd", "lib", "libroadsurf_hip_divcheck.so")
    if not os.path.exists(chk):
        subprocess.call(["make", "-C", os.path.join(ROOT, "roadsurf_amd"), "divcheck"], stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src") and not os.path.exists(
            os.path.join(ROOT, "oracle", "_ref", "libroadrunner_tools_ref.so")):
        subprocess.check_call(["bash", os.path.join(ROOT, "oracle", "build_ref.sh")], stdout=subprocess.DEVNULL)


def pytest_sessionstart(session):
    _ensure_built()


@pytest.fixture(scope="session")
def hip_lib():
    from roadsurf_amd import lib
    return lib.load()


@pytest.fixture(autouse=True)
def _plan_order_for_small_batches(monkeypatch):
    """rs_driver_run leaves blocks of fewer than 4 096 points in natural order with launches of eight hours (they
    are a handful of wavefronts: the latency of their dependent steps whatever the order).  The suite's batches
    are that small, and it is the plan-order path with its re-sorts that they are there to hold to the
    reference's bits: ask for it.  Tests that want natural order set the variable to 0 themselves;
    tests/test_hip_operational.py removes it to run the library's own default."""
    if "ROADSURF_HIP_CLUSTER" not in os.environ:
        monkeypatch.setenv("ROADSURF_HIP_CLUSTER", "1")
