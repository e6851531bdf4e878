"""Test-side loader for the CPU oracles (TEST INFRASTRUCTURE).

Two libraries share one harness ABI (``oracle/harness.c``):

* ``oracle/_ref/libroadsurf_ref.so`` — the reference's own Fortran, built by
  ``oracle/build_ref.sh`` (only where ``/root/reference`` exists, or prebuilt);
* ``oracle/liboracle.so`` — our C restatement (``oracle/roadsurf_oracle.c``).

Nothing under ``roadsurf_amd/`` imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from roadsurf_amd import abi  # noqa: E402

ORACLE_DIR = os.path.join(ROOT, "oracle")
PORT_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libroadsurf_ref.so")
#: same sources, `allocator`'s coupling dummy INTENT(INOUT): coupling as gfortran runs it
REF_CPL_SO = os.path.join(ORACLE_DIR, "_ref", "libroadsurf_ref_cpl.so")

F64_IN = ("tair", "tdew", "vz", "rhz", "prec", "sw", "lw", "sw_dir", "lw_net", "tsurfobs", "depth")
I32_AXIS = ("year", "month", "day", "hour", "minute", "second")
F64_OUT = ("tsurf", "snow", "water", "ice", "deposit", "ice2")


class HarnessArrays(C.Structure):
    _fields_ = (
        [(n, abi.c_double_p) for n in F64_IN]
        + [("precphase", abi.c_int32_p)]
        + [(n, abi.c_int32_p) for n in I32_AXIS]
        + [("local_horizons", abi.c_double_p)]
        + [(n, abi.c_double_p) for n in F64_OUT]
    )


def build_port() -> None:
    subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)


def build_ref() -> None:
    subprocess.check_call(["bash", os.path.join(ORACLE_DIR, "build_ref.sh")], stdout=subprocess.DEVNULL)


_libs: dict[str, C.CDLL] = {}


def load(kind: str) -> C.CDLL:
    """kind: 'port' (C restatement), 'ref' (reference Fortran, strict build) or 'ref_cpl'
    (reference Fortran with working coupling, see oracle/build_ref.sh)."""
    if kind in _libs:
        return _libs[kind]
    path = {"port": PORT_SO, "ref": REF_SO, "ref_cpl": REF_CPL_SO}[kind]
    if kind == "port" and not os.path.exists(path):
        build_port()
    if kind.startswith("ref") and not os.path.exists(path):
        if os.path.isdir("/root/reference/src"):
            build_ref()
        else:
            raise FileNotFoundError(path)
    lib = C.CDLL(path)
    lib.harness_run_points.restype = C.c_int
    lib.harness_run_points.argtypes = [
        C.c_int32, C.POINTER(HarnessArrays), C.POINTER(abi.InputSettings),
        C.POINTER(abi.InputParameters), C.POINTER(abi.LocalParameters), C.c_int32,
    ]
    lib.harness_max_threads.restype = C.c_int
    _libs[kind] = lib
    return lib


def have_ref() -> bool:
    return os.path.exists(REF_SO) or os.path.isdir("/root/reference/src")


def time_axis(simlen: int, dtsecs: float = 30.0, start=(2024, 1, 10, 0, 0, 0)):
    """Shared time axis, all points (start 2024-01-10 00:00)."""
    from roadsurf_amd import synth
    return synth.time_axis(simlen, dtsecs, start)


def synth_forcing(n: int, simlen: int, seed: int = 1234, point_offset: int = 0,
                  steps_per_knot: int = 120, start_hour: int = 0) -> dict[str, np.ndarray]:
    """The synthetic workload as per-point [n][simlen] host arrays: the PRODUCT's own host twin of its device
    generator (roadsurf_amd/synth.py over rs_synth_fill_points) - the checker only checks."""
    from roadsurf_amd import synth
    return synth.synth_forcing(n, simlen, seed, point_offset, steps_per_knot, start_hour)


def run_oracle(kind: str, forcing: dict[str, np.ndarray], settings: abi.InputSettings,
               params: abi.InputParameters, local, nthreads: int = 0,
               copy_inputs: bool = True):
    """Run n points through an oracle.  Returns (outputs dict [n][simlen], mutated inputs, threads)."""
    lib = load(kind)
    n, simlen = forcing["tair"].shape
    assert simlen == settings.SimLen
    f = {k: (np.array(v, copy=True) if copy_inputs else v) for k, v in forcing.items()}
    for k in F64_IN:
        assert f[k].dtype == np.float64 and f[k].flags.c_contiguous and f[k].shape == (n, simlen), k
    assert f["precphase"].dtype == np.int32 and f["precphase"].shape == (n, simlen)
    out = {k: np.full((n, simlen), np.nan) for k in F64_OUT}
    a = HarnessArrays()
    for k in F64_IN:
        setattr(a, k, f[k].ctypes.data_as(abi.c_double_p))
    a.precphase = f["precphase"].ctypes.data_as(abi.c_int32_p)
    for k in I32_AXIS:
        assert f[k].dtype == np.int32 and f[k].shape == (simlen,)
        setattr(a, k, f[k].ctypes.data_as(abi.c_int32_p))
    hz = forcing.get("local_horizons")
    a.local_horizons = hz.ctypes.data_as(abi.c_double_p) if hz is not None else None
    for k in F64_OUT:
        setattr(a, k, out[k].ctypes.data_as(abi.c_double_p))
    if isinstance(local, abi.LocalParameters):
        local = [local] * n
    larr = (abi.LocalParameters * n)(*local)
    # the reference reports every bad value, failed coupling and non-converged loop on unit 6, and
    # the Fortran runtime keeps part of it until the process ends - behind the test summary
    if os.environ.get("ORACLE_VERBOSE"):
        used = lib.harness_run_points(n, C.byref(a), C.byref(settings), C.byref(params), larr, nthreads)
    else:
        with quiet_stdout():
            used = lib.harness_run_points(n, C.byref(a), C.byref(settings), C.byref(params), larr, nthreads)
    return out, f, used


def point_pointers(f: dict, p: int, out: dict | None = None):
    """InputPointers/OutputPointers for point p of reference-layout arrays (keeps arrays alive)."""
    from roadsurf_amd import synth
    return synth.point_pointers(f, p, out)


class quiet_stdout:
    """Silence what the reference's Fortran prints to unit 6 (file descriptor 1) inside the block:
    it reports every bad value and every non-converged boundary-layer loop
    (src/BoundaryLayer.f90:71-74,98-101, src/InputOutput.f90:63-65), millions of lines on
    adversarial inputs."""

    target = os.devnull

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        self._null = os.open(self.target, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        os.dup2(self._null, 1)
        return self

    def __exit__(self, *exc):
        for so in (REF_SO, REF_CPL_SO):  # the Fortran runtime buffers unit 6: empty it first
            if os.path.exists(so):
                try:
                    C.CDLL(so).ref_flush_stdout()
                except AttributeError:
                    pass
        C.CDLL(None).fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._null)
        os.close(self._saved)
        return False


class capture_stdout(quiet_stdout):
    """What the code inside the block writes to file descriptor 1 (the reference's and the product's Fortran
    diagnostics included), kept in ``path``; ``text()`` afterwards."""

    def __init__(self, path: str):
        self.target = path

    def text(self) -> str:
        with open(self.target) as fh:
            return fh.read()
