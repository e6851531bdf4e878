"""GPU: coupling (src/Coupling.f90) — per-point replay of the coupling window with scaled
radiation until the simulated surface temperature meets the last observation.

Oracle: the reference built with working coupling (oracle/build_ref.sh explains why the
strict amdflang build has inert coupling: an INTENT(OUT) dummy wipes the observation), or the
C restatement, which is bit-identical to that build (tests/test_oracle_vs_golden.py)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_helpers as oh
from roadsurf_amd import abi, lib

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _kind():
    return "ref_cpl" if os.path.exists(oh.REF_CPL_SO) else "port"


def _cases(n, L, seed):
    f = oh.synth_forcing(n, L, seed=seed)
    s0 = abi.default_settings(L); p = abi.default_parameters(); l0 = abi.default_local(); l0.InitLenI = 1
    base, _, _ = oh.run_oracle("port", f, s0, p, l0)
    rs = np.random.RandomState(seed)
    out = []
    for case in range(4):
        s = abi.default_settings(L); s.use_coupling = 1
        if case == 1:
            s.use_relaxation = 1
        if case == 2:
            s.coupling_minutes = 60
        ls = []
        f2 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
        for i in range(n):
            li = abi.default_local()
            ci = L // 2 if case != 3 else int(rs.randint(50, L - 60))
            li.InitLenI = ci if case != 2 else 1
            li.couplingIndexI = ci
            off = rs.choice([0.0, 0.05, 0.5, -0.5, 2.0, -2.0, 6.0, -6.0, 15.0, -15.0])
            li.couplingTsurf = float(base["tsurf"][i, ci - 1] + off)
            if i % 50 == 0:
                li.couplingTsurf = -9999.0       # no usable observation: coupling off for the point
            if i % 51 == 0:
                li.couplingIndexI = 0
            li.tair_relax = float(f["tair"][i, min(ci, L - 1)]) + 1.0; li.VZ_relax = 3.0; li.RH_relax = 80.0
            ls.append(li)
        if case != 2:
            f2["tsurfobs"][:, :] = base["tsurf"] + 0.3
        out.append((f2, s, p, ls))
    return out, base


def test_coupled_runs_match_oracle_bitwise():
    from roadsurf_amd import device
    n, L = 384, 2881
    cases, base = _cases(n, L, 4242)
    for k, (f2, s, p, ls) in enumerate(cases):
        ora, _, _ = oh.run_oracle(_kind(), f2, s, p, ls)
        res, _ = device.run_points(f2, s, p, ls)
        worst = max(float(np.abs(res[q] - ora[q]).max()) for q in oh.F64_OUT)
        nonid = sum(int((res[q] != ora[q]).sum()) for q in oh.F64_OUT)
        moved = int((np.abs(ora["tsurf"] - base["tsurf"]).max(1) > 1e-3).sum())
        print(f"case {k}: max|diff| {worst:.3e}, non-identical values {nonid}, points moved by coupling {moved}")
        assert worst < TOL
        assert nonid == 0
        assert moved > n // 2  # coupling really acts in these cases


def test_coupling_through_the_c_abi_batch_entry():
    L = lib.load()
    n, SL = 300, 1441
    cases, _ = _cases(n, SL, 77)
    f2, s, p, ls = cases[1]
    ora, _, _ = oh.run_oracle(_kind(), f2, s, p, ls)
    g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f2.items()}
    out = {k: np.full((n, SL), np.nan) for k in oh.F64_OUT}
    ips = (abi.InputPointers * n)(); ops = (abi.OutputPointers * n)(); keep = []
    for pt in range(n):
        ip, op, kp = oh.point_pointers(g, pt, out)
        ips[pt], ops[pt] = ip, op
        keep.append(kp)
    larr = (abi.LocalParameters * n)(*ls)
    st = C.c_int32(99)
    L.runsimulation_batch(n, ops, ips, C.byref(s), C.byref(p), larr, C.byref(st))
    assert st.value == 0, lib.last_error()
    for k in oh.F64_OUT:
        assert np.array_equal(out[k], ora[k]), k


def test_coupling_window_must_be_whole_series():
    from roadsurf_amd import device
    n, L = 64, 721
    cases, _ = _cases(n, L, 5)
    f2, s, p, ls = cases[0]
    plan = device.Plan(n, s, p, 0)
    import torch
    dev = plan.device
    win = device.ForcingWindow.empty(100, plan.np_pad, dev)
    out = device.OutputWindow.empty(100, plan.np_pad, dev)
    z = torch.zeros(plan.np_pad, dtype=torch.int32, device=dev)
    zd = torch.zeros(plan.np_pad, dtype=torch.float64, device=dev)
    pp = plan.point_params(0.0, z, None, None, None, z, zd)
    with pytest.raises(RuntimeError, match="whole series"):
        plan.step(win, out, pp, 1, 100)
    plan.close()


def _set_replay_mode(monkeypatch, replay):
    """general / lockstep: the kernel of the replay rounds; lockstep-collapsed: every listed point runs
    all its replays in the first round's launch (ROADSURF_HIP_CPL_COLLAPSE=1); lockstep-rounds: one
    launch per round throughout (=0).  The default collapses from round 7 on."""
    if not replay:
        return
    kind, _, how = replay.partition("-")
    monkeypatch.setenv("ROADSURF_HIP_CPL_REPLAY", kind)
    if how:
        monkeypatch.setenv("ROADSURF_HIP_CPL_COLLAPSE", "1" if how == "collapsed" else "0")


@pytest.mark.parametrize("replay", [None, "general", "lockstep", "lockstep-collapsed", "lockstep-rounds"])
@pytest.mark.parametrize("chunk", [97, 256])
def test_chunked_coupling_equals_whole_series_and_the_reference(chunk, replay, monkeypatch):
    """rs_hip_step_cpl / rs_hip_cpl_replay: lock-step chunks that park a point behind its coupling
    window, replay rounds over the compacted list of parked points, lock-step chunks again.  Every
    case of _cases - relaxation on, a 60-minute window, coupling switched off for some points, and
    (case 3) a different coupling index for every point, where points that are ahead wait for the
    others - must give the whole-series run's bits, which are the reference's.  The replay rounds
    run in lock step over the list where the block is compact, with the general kernel (a time index
    per lane) where it is not; both are forced on every case here."""
    from roadsurf_amd import device
    _set_replay_mode(monkeypatch, replay)
    n, L = 384, 2881
    cases, base = _cases(n, L, 4242)
    for k, (f2, s, p, ls) in enumerate(cases):
        ora, _, _ = oh.run_oracle(_kind(), f2, s, p, ls)
        res, nfail = device.run_points(f2, s, p, ls, chunk=chunk)
        for q in oh.F64_OUT:
            assert np.array_equal(res[q], ora[q]), (k, q, int((res[q] != ora[q]).sum()))


@pytest.mark.parametrize("replay", ["general", "lockstep", "lockstep-collapsed"])
def test_chunked_coupling_with_failing_points(replay, monkeypatch):
    """A point that fails inside its coupling window - in the first pass or in a replay - keeps the
    outputs earlier passes saved beyond the failure (SaveOutput only ever overwrites).  A bad value
    at the index BEHIND the window end is seen by CheckValues every time the loop arrives there,
    i.e. right before each rewind: the step at the window start still runs, then the loop exits."""
    from roadsurf_amd import device
    _set_replay_mode(monkeypatch, replay)
    n, L = 256, 1441
    cases, _ = _cases(n, L, 11)
    f2, s, p, ls = cases[0]
    ci = L // 2
    f2["tair"][3, ci - 100] = -200.0      # bad value inside the window: fails in the first pass
    f2["tair"][9, ci + 50] = 150.0        # behind the window
    f2["tair"][17, 5] = -200.0            # before the window
    for q in (25, 26, 27, 28, 29, 30):    # at the index behind the window end (1-based ci + 1)
        f2["tair"][q, ci] = -200.0
    # AT the window end (1-based ci): the step still runs and Coupling_control still decides
    # (CheckEndCoupling does not look at simulation_failed), then the loop exits - no rewind, the
    # rows behind the window stay -9999.0 (a lane that parked there would leave them unwritten)
    for q in range(33, 49):
        f2["tair"][q, ci - 1] = -200.0
    ora, _, _ = oh.run_oracle(_kind(), f2, s, p, ls)
    whole, _ = device.run_points(f2, s, p, ls)
    parts, _ = device.run_points(f2, s, p, ls, chunk=128)
    for q in oh.F64_OUT:
        assert np.array_equal(whole[q], ora[q]), q
        assert np.array_equal(parts[q], ora[q]), q
    assert (ora["tsurf"][33:49, ci:] == -9999.0).all() and (ora["tsurf"][33:49, ci - 1] != -9999.0).all()


def test_replay_window_that_misses_a_coupling_window_is_refused():
    """rs_hip_cpl_replay checks the caller's window against the coupling windows of the points that ask
    for a replay: the rewind reads the forcing of the index behind the window end, and a point whose
    window start lies before the window would never step.  (In-tree callers compute the block from
    the points' couplingIndexI; this is the guard of the public entry.)"""
    import torch
    from roadsurf_amd import device
    n, L = 256, 121
    s = abi.default_settings(L); s.use_coupling = 1
    p = abi.default_parameters()
    plan = device.Plan(n, s, p, 0)
    dev, npad = plan.device, plan.np_pad
    win = device.ForcingWindow.empty(L, npad, dev)
    for k, v in (("tair", -2.0), ("tdew", -4.0), ("vz", 3.0), ("rhz", 80.0), ("prec", 0.0), ("sw", 0.0),
                 ("lw", 250.0), ("tsurfobs", -9999.9), ("depth", -9999.9)):
        win.tensors[k].fill_(v)
    win.tensors["precphase"].fill_(-9999)
    win.tensors["hour"].fill_(12)
    win.tensors["tsurfobs"][0].fill_(-2.0)
    out = device.OutputWindow.empty(L, npad, dev)
    ci = torch.full((npad,), 50, dtype=torch.int32, device=dev)
    ct = torch.full((npad,), 4.0, dtype=torch.float64, device=dev)   # far from what the model reaches
    il = torch.ones((npad,), dtype=torch.int32, device=dev)
    pp = plan.point_params(plan.uniform_tbottom(2024, 1, 10), il, None, None, None, ci, ct)
    plan.init_state(win, pp)
    plan.step_cpl(win, out, pp, 1, 60, window_row=0, out_row0=0)   # every point parks behind index 50
    with pytest.raises(RuntimeError, match="does not cover"):
        plan.cpl_replay(win, out, pp, 30, 31, window_row=29, out_row0=0)   # window starts behind couplingStartI = 1
    with pytest.raises(RuntimeError, match="does not cover"):
        plan.cpl_replay(win, out, pp, 1, 50, window_row=0, out_row0=0)     # ends at the window end, not behind it
    rounds = plan.cpl_replay(win, out, pp, 1, 51, window_row=0, out_row0=0)
    assert rounds is None or rounds >= 1
    plan.close()
