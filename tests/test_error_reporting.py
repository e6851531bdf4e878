"""How the reference's own entry point reports that it could not run.

`runsimulation` has no status argument (examples/example1/src/Simulation.f90:4-6).  When the call
cannot run at all, this library prints one line to standard error, fills the outputs with -9999.0 -
indistinguishable from a point that failed CheckValues at index 1 - and leaves the reason in
rs_last_error() ("Errors").  On a box without a GPU that is every call: the library
has no CPU path, and says so."""
import ctypes as C

import numpy as np
import pytest

import oracle_helpers as oh
from roadsurf_amd import abi, lib


def _one_point(L):
    f = oh.synth_forcing(1, L, seed=3)
    out = {k: np.full((1, L), np.nan) for k in oh.F64_OUT}
    ip, op, keep = oh.point_pointers(f, 0, out)
    return f, out, ip, op, keep


def test_bad_settings_leave_minus_9999_and_a_message():
    """NLayers outside 5..32: refused on the host before any device is touched (runs anywhere)."""
    Lb = lib.load()
    L = 61
    f, out, ip, op, keep = _one_point(L)
    s = abi.default_settings(L); s.NLayers = 3
    p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    Lb.runsimulation(C.byref(op), C.byref(ip), C.byref(s), C.byref(p), C.byref(l))
    for k in oh.F64_OUT:
        assert (out[k] == -9999.0).all(), k
    assert "NLayers" in lib.last_error()
    st = C.c_int32(0)
    ips = (abi.InputPointers * 1)(ip); ops = (abi.OutputPointers * 1)(op); ls = (abi.LocalParameters * 1)(l)
    Lb.runsimulation_batch(1, ops, ips, C.byref(s), C.byref(p), ls, C.byref(st))
    assert st.value == -1


def test_without_a_gpu_the_call_fails_loudly_not_silently():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the no-device path cannot be reached")
    Lb = lib.load()
    L = 61
    f, out, ip, op, keep = _one_point(L)
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    Lb.runsimulation(C.byref(op), C.byref(ip), C.byref(s), C.byref(p), C.byref(l))
    for k in oh.F64_OUT:
        assert (out[k] == -9999.0).all(), k
    assert "no HIP device" in lib.last_error() and "no CPU path" in lib.last_error()


@pytest.mark.gpu
def test_a_successful_call_clears_the_message():
    Lb = lib.load()
    L = 61
    f, out, ip, op, keep = _one_point(L)
    s = abi.default_settings(L); s.NLayers = 3
    p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    Lb.runsimulation(C.byref(op), C.byref(ip), C.byref(s), C.byref(p), C.byref(l))
    assert lib.last_error() != ""
    s.NLayers = 15
    Lb.runsimulation(C.byref(op), C.byref(ip), C.byref(s), C.byref(p), C.byref(l))
    assert lib.last_error() == ""
    ora, _, _ = oh.run_oracle("port", f, s, p, l)
    for k in oh.F64_OUT:
        assert np.array_equal(out[k], ora[k]), k
