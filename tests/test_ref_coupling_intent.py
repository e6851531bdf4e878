"""CPU: the claim behind every "coupling is bit-identical to the reference" in this repository.

oracle/build_ref.sh builds the reference twice: as it is (libroadsurf_ref.so) and with ONE token of
`allocator` changed - its coupling dummy INTENT(OUT) -> INTENT(INOUT) (libroadsurf_ref_cpl.so).  The script
says why: `setInputParam` stores the observation in the coupling object (src/InputOutput.f90:30-33) and
`allocator` then receives that object as INTENT(OUT) (src/Initialization.f90:96,150-157); flang
re-initialises an INTENT(OUT) object of a type with default initialisation, so in the build as compiled
here the observation is gone, the coupling window is [1, 0] and coupling never runs.  These tests hold
that explanation to what the two builds actually do, through oracle/ref_probe.f90's
`ref_probe_coupling_init` (the reference's own ConnectFortran2Carrays + Initialization, nothing else)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_helpers as oh
from roadsurf_amd import abi

pytestmark = pytest.mark.skipif(
    not (os.path.exists(oh.REF_SO) and os.path.exists(oh.REF_CPL_SO)),
    reason="needs oracle/_ref (built where /root/reference exists; travels prebuilt to the GPU box)")

L_, CI, OBS = 1441, 700, -2.5


def _inputs(seed=5):
    f = oh.synth_forcing(1, L_, seed=seed)
    f["tsurfobs"][:, :] = -3.0
    s = abi.default_settings(L_)
    s.use_coupling = 1
    p = abi.default_parameters()
    l = abi.default_local()
    l.InitLenI = CI
    l.couplingIndexI = CI
    l.couplingTsurf = OBS
    return f, s, p, l


def _probe(kind, f, s, p, l):
    lib = oh.load(kind)
    ip, op, keep = oh.point_pointers(f, 0)
    iv = (C.c_int32 * 8)()
    rv = (C.c_double * 4)()
    with oh.quiet_stdout():
        lib.ref_probe_coupling_init(C.byref(ip), C.byref(op), C.byref(s), C.byref(p), C.byref(l), iv, rv)
    names = ("use_coupling", "obsI1", "startI1", "endI1", "NObs", "CoupPhaseN", "failed", "use_relaxation")
    out = dict(zip(names, list(iv)))
    out.update(lastTsurfObs=rv[0], obsTsurf1=rv[1], RadCoeff=rv[3])
    return out


def test_strict_build_wipes_the_observation_after_setinputparam():
    """The reference as amdflang compiles it: NObs / obsI(1) / obsTsurf(1) are back at their default
    initialisation behind `allocator`, the coupling window is [1, 0]."""
    q = _probe("ref", *_inputs())
    assert q["use_coupling"] == 1
    assert (q["NObs"], q["obsI1"], q["obsTsurf1"], q["lastTsurfObs"]) == (0, 0, 0.0, 0.0)
    assert (q["startI1"], q["endI1"]) == (1, 0)


def test_patched_build_keeps_what_setinputparam_stored():
    """Same sources, the dummy INTENT(INOUT): the observation setInputParam stored survives and
    initCouplingTimes (src/Coupling.f90:486-534) places the window coupling_minutes before it."""
    f, s, p, l = _inputs()
    q = _probe("ref_cpl", f, s, p, l)
    assert (q["NObs"], q["obsI1"], q["obsTsurf1"], q["lastTsurfObs"]) == (1, CI, OBS, OBS)
    win = int(s.coupling_minutes * 60 / s.DTSecs)
    assert (q["startI1"], q["endI1"]) == (CI - win, CI)
    assert q["failed"] == 0 and q["RadCoeff"] == 1.0


def test_coupling_is_inert_in_the_strict_build_and_acts_in_the_patched_one():
    """Strict build, use_coupling = 1: the VALUE of the observation never reaches the run (two observations
    8 K apart give the same bits), and with no surface observations to force after index 1 the run equals the
    one with the coupling observation removed altogether.  Patched build: the same pairs differ - coupling acts -
    and without a usable observation the two builds are the same program."""
    f, s, p, l = _inputs(seed=11)
    f["tsurfobs"][:, 1:] = -9999.9  # (with use_coupling on, SetCurrentValues forces no observation inside the
    #                                  window [1, 0] either, src/InputOutput.f90:120-121: keep that out of the comparison)
    l_none = abi.default_local()
    l_none.InitLenI = CI
    l_none.couplingIndexI = 0
    l_none.couplingTsurf = -9999.0
    base, _, _ = oh.run_oracle("ref", f, s, p, l_none)
    # observations well away from what the model simulates, so that a working coupling has to act
    l.couplingTsurf = float(base["tsurf"][0, CI - 1]) + 4.0
    l2 = abi.default_local()
    l2.InitLenI = CI
    l2.couplingIndexI = CI
    l2.couplingTsurf = l.couplingTsurf - 8.0
    strict_a, _, _ = oh.run_oracle("ref", f, s, p, l)
    strict_b, _, _ = oh.run_oracle("ref", f, s, p, l2)
    for k in oh.F64_OUT:
        assert np.array_equal(strict_a[k], strict_b[k]), k
        assert np.array_equal(strict_a[k], base[k]), k
    cpl_a, _, _ = oh.run_oracle("ref_cpl", f, s, p, l)
    cpl_b, _, _ = oh.run_oracle("ref_cpl", f, s, p, l2)
    cpl_none, _, _ = oh.run_oracle("ref_cpl", f, s, p, l_none)
    assert np.abs(cpl_a["tsurf"] - cpl_none["tsurf"]).max() > 0.1
    assert np.abs(cpl_a["tsurf"] - cpl_b["tsurf"]).max() > 0.1
    for k in oh.F64_OUT:
        assert np.array_equal(cpl_none[k], base[k]), k
