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
    """Build the in-tree libraries when they are missing OR STALE: every product library carries the hash of
    the sources it was built from (roadsurf_amd/Makefile, provenance.build_sha16()); one whose stamp is not the
    hash of the sources beside it is rebuilt before any test maps it (round 5 ran a suite against a
    libroadsurf_hip_divcheck.so older than csrc/).  The product library cross-compiles without a GPU; on the GPU
    box the prebuilt .so files travel with the snapshot together with their sources, so the stamps match and
    nothing is rebuilt there."""
    import subprocess
    from roadsurf_amd import provenance
    amd = os.path.join(ROOT, "roadsurf_amd")
    want = provenance.build_sha16()
    rebuilt = False
    for so, target in (("libroadsurf_hip.so", []), ("libroadsurf_hip_divcheck.so", ["divcheck"])):
        path = os.path.join(amd, "lib", so)
        if provenance.stamp_of(path) == want:
            continue
        subprocess.check_call(["make", "-C", amd, "-j4"] + target, stdout=subprocess.DEVNULL)
        if provenance.stamp_of(path) != want:  # (make saw nothing to do: time stamps that lie) - from scratch
            subprocess.check_call(["make", "-C", amd, "-j4", "-B"] + target, stdout=subprocess.DEVNULL)
        if provenance.stamp_of(path) != want:
            raise RuntimeError(f"{so}: build stamp {provenance.stamp_of(path)} != sources {want} after a rebuild")
        rebuilt = True
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], stdout=subprocess.DEVNULL)
    else:  # the checker's own sources are three C files: let make compare the time stamps
        subprocess.call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], stdout=subprocess.DEVNULL,
                        stderr=subprocess.DEVNULL)
    # oracle/_ref: the reference's libraries, and the reference's Simulation.f90 over THIS library's module
    # surface (libsimulation_over_hip.so: stale when the product's modules have changed since - build_ref.sh
    # leaves the product's stamp beside it).  Only where the reference's sources exist.
    if os.path.isdir("/root/reference/src"):
        ref_dir = os.path.join(ROOT, "oracle", "_ref")
        over_stamp = os.path.join(ref_dir, "over_hip.stamp")
        have = open(over_stamp).read().strip() if os.path.exists(over_stamp) else None
        if (rebuilt or have != want or not os.path.exists(os.path.join(ref_dir, "libroadrunner_tools_ref.so"))
                or not os.path.exists(os.path.join(ref_dir, "libsimulation_over_hip.so"))):
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


# ---- kernel reachability map (tools/kernel_reachability.py): with RS_TEST_WINDOWS=<file> every test's wall-clock
# window is appended to the file (CLOCK_MONOTONIC, the clock rocprofv3 stamps kernel dispatches with), so that a
# `rocprofv3 --kernel-trace -- python3 -m pytest tests -m gpu` run can be reduced to "kernel -> tests that launched it"
@pytest.fixture(autouse=True)
def _test_window(request):
    path = os.environ.get("RS_TEST_WINDOWS")
    if not path:
        yield
        return
    import time
    t0 = time.monotonic_ns()
    yield
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:
        pass
    with open(path, "a") as f:
        f.write("%s,%d,%d\n" % (request.node.nodeid.replace(",", ";"), t0, time.monotonic_ns()))
