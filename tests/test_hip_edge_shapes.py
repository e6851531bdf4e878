"""Edge shapes of the batched boundary (runsimulation_batch, the extension beside the reference's
runsimulation, examples/example1/src/Simulation.f90:4-115): ragged batch sizes around the wavefront
and workgroup widths, series of one to a few time indices, an empty batch, batches whose every point
fails CheckValues at the first index, initialization phases that cover nothing or everything.  Every
case is compared with the reference (built from its own sources, oracle/_ref; the C restatement when
that is absent) bit for bit."""
import ctypes as C

import numpy as np
import pytest

import oracle_helpers as oh
from roadsurf_amd import abi, lib
from test_hip_boundary import _kind, _pointers

pytestmark = pytest.mark.gpu


def _run_batch(f, s, p, ls, first_failed=False):
    L = lib.load()
    n, SL = f["tair"].shape
    g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
    out = {k: np.full((n, SL), np.nan) for k in oh.F64_OUT}
    ips = (abi.InputPointers * max(n, 1))(); ops = (abi.OutputPointers * max(n, 1))(); keep = []
    for pt in range(n):
        ip, op, hz = _pointers(g, out, pt)
        ips[pt], ops[pt] = ip, op
        keep.append(hz)
    larr = (abi.LocalParameters * max(n, 1))(*ls)
    st = C.c_int32(99)
    ff = np.full(max(n, 1), -7, np.int32)
    if first_failed:
        L.runsimulation_batch_ex(n, ops, ips, C.byref(s), C.byref(p), larr, C.byref(st),
                                 ff.ctypes.data_as(abi.c_int32_p))
    else:
        L.runsimulation_batch(n, ops, ips, C.byref(s), C.byref(p), larr, C.byref(st))
    return out, g, st.value, ff


def _locals(n, initlen=1):
    ls = []
    for _ in range(n):
        li = abi.default_local(); li.InitLenI = initlen
        ls.append(li)
    return ls


def _same(out, ora):
    for k in oh.F64_OUT:
        assert np.array_equal(out[k], ora[k]), k


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 257, 513])
def test_ragged_batch_sizes(n):
    """point counts on both sides of the wavefront (64) and workgroup (256) widths"""
    SL = 361
    f = oh.synth_forcing(n, SL, seed=100 + n)
    s = abi.default_settings(SL); p = abi.default_parameters(); ls = _locals(n)
    ora, _, _ = oh.run_oracle(_kind(), f, s, p, ls)
    out, _, st, _ = _run_batch(f, s, p, ls)
    assert st == 0, lib.last_error()
    _same(out, ora)


@pytest.mark.parametrize("SL", [1, 2, 3, 5, 121, 122])
def test_short_series(SL):
    """SimLen 1 runs only the final step (lastValues, src/InputOutput.f90:169-198); 2 and 3 the
    shortest loops; 121/122 one index on each side of a 120-index launch"""
    n = 70
    f = oh.synth_forcing(n, SL, seed=40 + SL)
    s = abi.default_settings(SL); p = abi.default_parameters(); ls = _locals(n)
    ora, _, _ = oh.run_oracle(_kind(), f, s, p, ls)
    out, _, st, _ = _run_batch(f, s, p, ls)
    assert st == 0, lib.last_error()
    _same(out, ora)


def test_empty_batch_is_a_noop():
    f = oh.synth_forcing(1, 11, seed=1)
    s = abi.default_settings(11); p = abi.default_parameters()
    L = lib.load()
    st = C.c_int32(99)
    L.runsimulation_batch(0, None, None, C.byref(s), C.byref(p), None, C.byref(st))
    assert st.value == 0, lib.last_error()


def test_every_point_fails_at_the_first_index():
    """CheckValues raises the flag at index 1 for the whole batch: the row of that index is still
    written (SaveOutput runs before the loop is left, Simulation.f90:100), everything after it
    stays -9999.0, and the extension reports index 1 for every point"""
    n, SL = 130, 241
    f = oh.synth_forcing(n, SL, seed=9)
    f["tair"][:, 0] = -200.0
    s = abi.default_settings(SL); p = abi.default_parameters(); ls = _locals(n)
    with oh.quiet_stdout():
        ora, _, _ = oh.run_oracle(_kind(), f, s, p, ls)
    out, _, st, ff = _run_batch(f, s, p, ls, first_failed=True)
    assert st == 0, lib.last_error()
    _same(out, ora)
    assert (out["tsurf"][:, 1:] == -9999.0).all()
    assert (ff[:n] == 1).all()


@pytest.mark.parametrize("initlen", [0, 1, 200, 241, 500])
def test_initialization_phase_lengths(initlen):
    """InitLenI from nothing to beyond the series, surface observations present throughout
    (SetCurrentValues forces Tmp(1:2) while i <= InitLenI, src/InputOutput.f90:116-148)"""
    n, SL = 96, 241
    f = oh.synth_forcing(n, SL, seed=77)
    f["tsurfobs"][:] = f["tair"] + 0.5
    f["tsurfobs"][3, 50:60] = -9999.9  # a gap inside the phase
    s = abi.default_settings(SL); p = abi.default_parameters(); ls = _locals(n, initlen)
    ora, _, _ = oh.run_oracle(_kind(), f, s, p, ls)
    out, _, st, _ = _run_batch(f, s, p, ls)
    assert st == 0, lib.last_error()
    _same(out, ora)


def test_no_surface_observation_at_all():
    """TSurfObs missing everywhere: the profile starts from the air temperature
    (src/Initialization.f90) and nothing is forced"""
    n, SL = 80, 181
    f = oh.synth_forcing(n, SL, seed=31)
    f["tsurfobs"][:] = -9999.9
    s = abi.default_settings(SL); p = abi.default_parameters(); ls = _locals(n)
    ora, _, _ = oh.run_oracle(_kind(), f, s, p, ls)
    out, _, st, _ = _run_batch(f, s, p, ls)
    assert st == 0, lib.last_error()
    _same(out, ora)
