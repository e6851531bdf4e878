"""GPU: the reference's OWN time loop (examples/example1/src/Simulation.f90, compiled unchanged by
oracle/build_ref.sh against this library's `module RoadSurf` / `module RoadSurfVariables` and linked against
libroadsurf_hip.so -> oracle/_ref/libsimulation_over_hip.so) drives the fourteen module procedures point by
point and index by index (roadsurf_amd/fortran/RoadSurfCompat.f90 over roadsurf_amd/csrc/rs_compat.hip).
Its outputs must carry the bits of the reference build and of this library's `runsimulation`."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_helpers as oh
from roadsurf_amd import abi, lib

pytestmark = pytest.mark.gpu

SIM_SO = os.path.join(oh.ORACLE_DIR, "_ref", "libsimulation_over_hip.so")


def _sim():
    if not os.path.exists(SIM_SO):
        if os.path.isdir("/root/reference/src"):
            oh.build_ref()
        else:
            pytest.skip("oracle/_ref/libsimulation_over_hip.so was not built (needs /root/reference at build time)")
    lib.load()  # the product first: the compiled time loop binds to it
    sim = C.CDLL(SIM_SO)
    sim.runsimulation.restype = None
    return sim


def _run_loop(sim, f, s, p, locs):
    """every point through the compiled Simulation.f90::runsimulation; returns outputs and the (edited) inputs"""
    n, L = f["tair"].shape
    g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
    out = {k: np.full((n, L), np.nan) for k in oh.F64_OUT}
    for pt in range(n):
        ip, op, keep = oh.point_pointers(g, pt, out)
        l = locs[pt] if isinstance(locs, (list, tuple)) else locs
        sim.runsimulation(C.byref(op), C.byref(ip), C.byref(s), C.byref(p), C.byref(l))
    return out, g


def _kind(coupled=False):
    if coupled:
        return "ref_cpl" if os.path.exists(oh.REF_CPL_SO) else "port"
    return "ref" if oh.have_ref() else "port"


def _same(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.int64), np.ascontiguousarray(b).view(np.int64))


def test_time_loop_over_the_module_surface_lean_48h():
    """BASELINE config 1's shape: one point (here three), 48 h, the reference's loop over our procedures"""
    sim = _sim()
    n, L = 3, 5761
    f = oh.synth_forcing(n, L, seed=21)
    f["vz"][1, 0] = 0.1  # Initialization raises VZ(1) to 0.4 in the caller's array
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    ora, fmut, _ = oh.run_oracle(_kind(), f, s, p, l)
    out, g = _run_loop(sim, f, s, p, l)
    for k in oh.F64_OUT:
        assert _same(out[k], ora[k]), (k, float(np.abs(out[k] - ora[k]).max()))
    assert g["vz"][1, 0] == fmut["vz"][1, 0] == np.float64(np.float32(0.4))


def test_time_loop_full_feature_set_and_a_failing_point():
    """initialization phase with observation forcing, relaxation behind it, an output depth, a point that
    CheckValues stops inside the series (its later rows keep Initialization's -9999.0)"""
    sim = _sim()
    n, L = 4, 1441
    f = oh.synth_forcing(n, L, seed=5)
    f["tsurfobs"][:, :300] = f["tair"][:, :300] + 0.4
    f["tair"][2, 700] = -200.0
    for depth in (-9999.9, 0.03):
        s = abi.default_settings(L); s.use_relaxation = 1; s.tsurfOutputDepth = depth
        p = abi.default_parameters()
        ls = []
        for i in range(n):
            li = abi.default_local(); li.InitLenI = 300
            li.tair_relax = float(f["tair"][i, 300]) - 1.5; li.VZ_relax = 2.5; li.RH_relax = 88.0
            ls.append(li)
        ora, _, _ = oh.run_oracle(_kind(), f, s, p, ls)
        out, _ = _run_loop(sim, f, s, p, ls)
        for k in oh.F64_OUT:
            assert _same(out[k], ora[k]), (depth, k, int((out[k] != ora[k]).sum()))
        assert (out["tsurf"][2] == -9999.0).sum() == L - 701


def test_time_loop_sky_view_edits_the_callers_arrays():
    """per-point sky view and local horizons: outputs, and SW / SW_dir / LW as ModRadiationBySurroundings and
    CheckValues leave them in the caller's arrays (src/ModRadiation.f90:57-71, src/InputOutput.f90:75-77)"""
    sim = _sim()
    n, L = 3, 1441
    f = oh.synth_forcing(n, L, seed=9, start_hour=6)
    rs = np.random.RandomState(2)
    f["local_horizons"] = rs.uniform(0, 25, (n, 360))
    f["sw_dir"][:, 200:260] = f["sw"][:, 200:260] + 5.0  # above SW: clamped in place
    s = abi.default_settings(L); p = abi.default_parameters()
    ls = []
    for i in range(n):
        li = abi.default_local(); li.InitLenI = 1
        li.lat, li.lon, li.sky_view = 60.0 + i, 22.0 + 2 * i, 0.45 + 0.2 * i
        ls.append(li)
    ora, fmut, _ = oh.run_oracle(_kind(), f, s, p, ls)
    out, g = _run_loop(sim, f, s, p, ls)
    for k in oh.F64_OUT:
        assert _same(out[k], ora[k]), (k, int((out[k] != ora[k]).sum()))
    for k in ("sw", "sw_dir", "lw"):
        assert _same(g[k], fmut[k]), k
    assert not _same(fmut["sw"], f["sw"])  # the sky view really edited something


def test_time_loop_with_coupling():
    """coupling: the device replays the window at its end (CheckEndCoupling), the loop's index never goes back;
    the outputs are the reference's after its last replay (reference build with allocator's coupling dummy
    INTENT(INOUT): oracle/build_ref.sh)"""
    sim = _sim()
    n, L = 4, 1441
    f = oh.synth_forcing(n, L, seed=13)
    f["tsurfobs"][:, :500] = f["tair"][:, :500] - 0.7
    s = abi.default_settings(L); s.use_coupling = 1; s.use_relaxation = 1; s.coupling_minutes = 120
    p = abi.default_parameters()
    ls = []
    for i in range(n):
        li = abi.default_local(); li.InitLenI = 500
        li.tair_relax = float(f["tair"][i, 500]); li.VZ_relax = 3.0; li.RH_relax = 85.0
        li.couplingIndexI = 500 if i != 3 else -9999           # the last point has no observation to couple to
        li.couplingTsurf = float(f["tair"][i, 499]) + (1.5 if i % 2 else -2.0) if i != 3 else -9999.9
        ls.append(li)
    ora, _, _ = oh.run_oracle(_kind(True), f, s, p, ls)
    base, _, _ = oh.run_oracle(_kind(True), f, abi.default_settings(L), p, ls)
    out, _ = _run_loop(sim, f, s, p, ls)
    for k in oh.F64_OUT:
        assert _same(out[k], ora[k]), (k, int((out[k] != ora[k]).sum()))
    assert np.abs(ora["tsurf"][:3] - base["tsurf"][:3]).max() > 0.05  # coupling acted


def test_time_loop_equals_the_librarys_own_runsimulation():
    sim = _sim()
    L_ = lib.load()
    n, L = 2, 2881
    f = oh.synth_forcing(n, L, seed=33)
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    a, _ = _run_loop(sim, f, s, p, l)
    b, _ = _run_loop(L_, f, s, p, l)
    for k in oh.F64_OUT:
        assert _same(a[k], b[k]), k
