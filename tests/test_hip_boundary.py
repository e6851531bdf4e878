"""GPU tests of the drop-in boundary (runsimulation / runsimulation_batch, the
reference's BIND(C) entry, examples/example1/src/Simulation.f90:4-6) and of the
synthetic-workload kernels."""
import ctypes as C

import numpy as np
import pytest

import oracle_helpers as oh
from roadsurf_amd import abi, lib

pytestmark = pytest.mark.gpu


def _pointers(f, out, p):
    ip = abi.InputPointers()
    ip.inputLen = f["tair"].shape[1]
    for name, key in (("c_tair", "tair"), ("c_tdew", "tdew"), ("c_VZ", "vz"), ("c_Rhz", "rhz"),
                      ("c_prec", "prec"), ("c_SW", "sw"), ("c_LW", "lw"), ("c_SW_dir", "sw_dir"),
                      ("c_LW_net", "lw_net"), ("c_TSurfObs", "tsurfobs"), ("c_Depth", "depth")):
        setattr(ip, name, f[key][p].ctypes.data_as(abi.c_double_p))
    ip.c_PrecPhase = f["precphase"][p].ctypes.data_as(abi.c_int32_p)
    hz = np.zeros(360)
    ip.c_local_horizons = hz.ctypes.data_as(abi.c_double_p)
    for name, key in (("c_year", "year"), ("c_month", "month"), ("c_day", "day"), ("c_hour", "hour"),
                      ("c_minute", "minute"), ("c_second", "second")):
        setattr(ip, name, f[key].ctypes.data_as(abi.c_int32_p))
    op = abi.OutputPointers()
    op.outputLen = ip.inputLen
    for name, key in (("c_TsurfOut", "tsurf"), ("c_SnowOut", "snow"), ("c_WaterOut", "water"),
                      ("c_IceOut", "ice"), ("c_DepositOut", "deposit"), ("c_Ice2Out", "ice2")):
        setattr(op, name, out[key][p].ctypes.data_as(abi.c_double_p))
    return ip, op, hz


def _kind():
    return "ref" if oh.have_ref() else "port"


def test_runsimulation_single_point_dropin():
    L = lib.load()
    n, SL = 3, 2881
    f = oh.synth_forcing(n, SL, seed=11)
    f["vz"][1, 0] = 0.1  # exercises the VZ(1) >= 0.4 side effect on the caller's array
    s = abi.default_settings(SL); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    ora, fmut, _ = oh.run_oracle(_kind(), f, s, p, l)
    g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
    out = {k: np.full((n, SL), np.nan) for k in oh.F64_OUT}
    for pt in range(n):
        ip, op, keep = _pointers(g, out, pt)
        L.runsimulation(C.byref(op), C.byref(ip), C.byref(s), C.byref(p), C.byref(l))
    for k in oh.F64_OUT:
        assert np.abs(out[k] - ora[k]).max() < 1e-6, k
    assert g["vz"][1, 0] == fmut["vz"][1, 0] == np.float64(np.float32(0.4))


def test_runsimulation_batch_matches_oracle(monkeypatch):
    L = lib.load()
    n, SL = 700, 1441
    # small tiles/chunks so the tiling logic is exercised (3 point tiles, 6 time chunks)
    monkeypatch.setenv("ROADSURF_HIP_TILE_POINTS", "300")
    monkeypatch.setenv("ROADSURF_HIP_CHUNK_STEPS", "250")
    f = oh.synth_forcing(n, SL, seed=3)
    f["tsurfobs"][:, :400] = f["tair"][:, :400] + 0.3
    f["tair"][5, 700] = -200.0  # one failing point
    s = abi.default_settings(SL); s.use_relaxation = 1
    p = abi.default_parameters()
    ls = []
    for i in range(n):
        li = abi.default_local(); li.InitLenI = 400
        li.tair_relax = float(f["tair"][i, 400]) - 1.0; li.VZ_relax = 2.0; li.RH_relax = 90.0
        ls.append(li)
    ora, _, _ = oh.run_oracle(_kind(), f, s, p, ls)
    g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
    out = {k: np.full((n, SL), np.nan) for k in oh.F64_OUT}
    ips = (abi.InputPointers * n)(); ops = (abi.OutputPointers * n)(); keep = []
    for pt in range(n):
        ip, op, hz = _pointers(g, out, pt)
        ips[pt], ops[pt] = ip, op
        keep.append(hz)
    larr = (abi.LocalParameters * n)(*ls)
    st = C.c_int32(99)
    L.runsimulation_batch(n, ops, ips, C.byref(s), C.byref(p), larr, C.byref(st))
    assert st.value == 0, lib.last_error()
    for k in oh.F64_OUT:
        miss = ora[k] == -9999.0
        assert np.array_equal(miss, out[k] == -9999.0), k
        assert np.abs(np.where(miss, 0, out[k] - ora[k])).max() < 1e-6, k
    assert (out["tsurf"][5] == -9999.0).sum() == SL - 701


def test_batch_points_with_their_own_calendars():
    """runsimulation_batch hands every point's OWN hour array to the kernels (RsForcing::hour_pstride = 1):
    points of one wavefront whose local hour differs take different branches of SetDayDependendVariables
    (src/BalanceModel.f90:354-387) at the same index.  A batch small enough for the two-wavefront flavour,
    two calendars interleaved lane by lane (the traffic friction used to travel as one word per
    index, lane 0's)."""
    L = lib.load()
    n, SL = 200, 1441
    fa = oh.synth_forcing(n, SL, seed=5, start_hour=0)
    fb = oh.synth_forcing(n, SL, seed=5, start_hour=11)  # same values, the calendar eleven hours on
    s = abi.default_settings(SL); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    oa, _, _ = oh.run_oracle(_kind(), fa, s, p, l)
    ob, _, _ = oh.run_oracle(_kind(), fb, s, p, l)
    assert not np.array_equal(oa["tsurf"], ob["tsurf"])  # the calendar matters
    out = {k: np.full((n, SL), np.nan) for k in oh.F64_OUT}
    ga = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in fa.items()}
    gb = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in fb.items()}
    ips = (abi.InputPointers * n)(); ops = (abi.OutputPointers * n)(); keep = []
    odd = np.arange(n) % 3 == 1
    for pt in range(n):
        ips[pt], ops[pt], hz = _pointers(gb if odd[pt] else ga, out, pt)
        keep.append(hz)
    larr = (abi.LocalParameters * n)(*([l] * n))
    st = C.c_int32(99)
    L.runsimulation_batch(n, ops, ips, C.byref(s), C.byref(p), larr, C.byref(st))
    assert st.value == 0, lib.last_error()
    for k in oh.F64_OUT:
        want = np.where(odd[:, None], ob[k], oa[k])
        assert np.array_equal(out[k], want), k


def test_batch_rejects_bad_arguments_loudly():
    L = lib.load()
    n, SL = 1, 241
    f = oh.synth_forcing(n, SL, seed=3)
    out = {k: np.full((n, SL), np.nan) for k in oh.F64_OUT}
    ip, op, hz = _pointers(f, out, 0)
    s = abi.default_settings(SL); p = abi.default_parameters(); l = abi.default_local()
    st = C.c_int32(0)
    s.NLayers = 40
    L.runsimulation_batch(1, C.byref(op), C.byref(ip), C.byref(s), C.byref(p), C.byref(l), C.byref(st))
    assert st.value == -1 and "NLayers" in lib.last_error()
    s.NLayers = 15
    ip.inputLen = 100
    L.runsimulation_batch(1, C.byref(op), C.byref(ip), C.byref(s), C.byref(p), C.byref(l), C.byref(st))
    assert st.value == -4 and "shorter than SimLen" in lib.last_error()


def test_device_synth_equals_host_twin():
    """knots + expand kernels produce bit-identical forcing to the host generator
    that feeds the oracle."""
    import torch
    from roadsurf_amd import device
    n, SL, spk = 1000, 1441, 120
    host = oh.synth_forcing(n, SL, seed=99, point_offset=12345)
    s = abi.default_settings(SL); p = abi.default_parameters()
    plan = device.Plan(n, s, p, 0)
    spec, knots = plan.synth_knots(99, SL // spk + 2, point_offset=12345, steps_per_knot=spk)
    dev = plan.device
    # odd window boundaries on purpose
    bounds = [1, 2, 100, 121, 500, 1201, SL + 1]
    got = {k: np.empty((SL, n)) for k in ("tair", "tdew", "vz", "rhz", "prec", "sw", "lw", "tsurfobs")}
    got["precphase"] = np.empty((SL, n), np.int32)
    hours = np.empty(SL, np.int32)
    for a, b in zip(bounds[:-1], bounds[1:]):
        w = device.ForcingWindow.empty(b - a, plan.np_pad, dev, optional=("tdew", "tsurfobs"))
        plan.expand(spec, knots, w, a, b - a)
        plan.sync()
        for k in got:
            got[k][a - 1:b - 1] = w.tensors[k][:, :n].cpu().numpy()
        hours[a - 1:b - 1] = w.tensors["hour"].cpu().numpy()
    for k in got:
        assert np.array_equal(got[k].T, host[k]), k
    assert np.array_equal(hours, host["hour"])
    plan.close()


def test_runsimulation_is_reentrant_from_driver_threads():
    """The reference driver runs one runsimulation per worker thread
    (examples/example1/src/roadrunner.cpp:490-497); so must the drop-in."""
    import threading
    L = lib.load()
    n, SL = 12, 721
    f = oh.synth_forcing(n, SL, seed=8)
    s = abi.default_settings(SL); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    ora, _, _ = oh.run_oracle(_kind(), f, s, p, l)
    g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
    out = {k: np.full((n, SL), np.nan) for k in oh.F64_OUT}
    args = [_pointers(g, out, pt) for pt in range(n)]
    errs = []

    def work(lo, hi):
        try:
            for pt in range(lo, hi):
                ip, op, _ = args[pt]
                L.runsimulation(C.byref(op), C.byref(ip), C.byref(s), C.byref(p), C.byref(l))
        except Exception as e:  # pragma: no cover
            errs.append(e)
    ts = [threading.Thread(target=work, args=(i * 3, i * 3 + 3)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs
    for k in oh.F64_OUT:
        assert np.array_equal(out[k], ora[k]), k


def test_more_concurrent_callers_than_any_fixed_plan_pool():
    """roadrunner.cpp:490-497 starts `options.jobs` workers, each inside runsimulation at the
    same time.  A plan holds its constants in its own device block (no table of slots), so the
    number of simultaneous callers is unbounded: 24 threads, one point each, all released
    together; every output must be the reference's (a call that could not get a plan would
    leave -9999.0 everywhere)."""
    import threading
    L = lib.load()
    n, SL = 24, 361
    f = oh.synth_forcing(n, SL, seed=31)
    s = abi.default_settings(SL); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    ora, _, _ = oh.run_oracle(_kind(), f, s, p, l)
    g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
    out = {k: np.full((n, SL), np.nan) for k in oh.F64_OUT}
    args = [_pointers(g, out, pt) for pt in range(n)]
    gate = threading.Barrier(n)
    errs = []

    def work(pt):
        try:
            ip, op, _ = args[pt]
            gate.wait()
            L.runsimulation(C.byref(op), C.byref(ip), C.byref(s), C.byref(p), C.byref(l))
        except Exception as e:  # pragma: no cover
            errs.append(e)
    ts = [threading.Thread(target=work, args=(pt,)) for pt in range(n)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs
    for k in oh.F64_OUT:
        assert np.array_equal(out[k], ora[k]), k


def test_batch_fans_out_over_the_device_list(monkeypatch):
    """north_star: points shard over the GPUs of a node with per-GPU streams and no collective.
    runsimulation_batch cuts the batch into contiguous blocks over ROADSURF_HIP_DEVICES, one host
    thread + stream + plan per block (the worker pool of examples/example1/src/roadrunner.cpp:
    423-501, per device instead of per point).  On a one-GPU box the list "0,0,0" gives three
    concurrent plans on the same device: results must equal the single-plan run bit for bit, and
    the reference."""
    L = lib.load()
    n, SL = 1000, 721
    f = oh.synth_forcing(n, SL, seed=17)
    f["tair"][777, 300] = -200.0  # a failing point in the last block
    s = abi.default_settings(SL); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    ora, _, _ = oh.run_oracle(_kind(), f, s, p, l)

    def run():
        g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
        out = {k: np.full((n, SL), np.nan) for k in oh.F64_OUT}
        ips = (abi.InputPointers * n)(); ops = (abi.OutputPointers * n)(); keep = []
        for pt in range(n):
            ips[pt], ops[pt], hz = _pointers(g, out, pt)
            keep.append(hz)
        larr = (abi.LocalParameters * n)(*([l] * n))
        st = C.c_int32(99)
        L.runsimulation_batch(n, ops, ips, C.byref(s), C.byref(p), larr, C.byref(st))
        assert st.value == 0, lib.last_error()
        return out, int(L.rs_last_fanout())

    monkeypatch.setenv("ROADSURF_HIP_DEVICES", "0")
    one, k1 = run()
    assert k1 == 1
    monkeypatch.setenv("ROADSURF_HIP_DEVICES", "0,0,0")
    monkeypatch.setenv("ROADSURF_HIP_MIN_SHARD", "100")
    three, k3 = run()
    assert k3 == 3
    for k in oh.F64_OUT:
        assert np.array_equal(one[k], three[k]), k
        assert np.array_equal(three[k], ora[k]), k
    # an out-of-range entry is ignored, a batch below the minimum block is not split
    monkeypatch.setenv("ROADSURF_HIP_DEVICES", "0,99")
    monkeypatch.setenv("ROADSURF_HIP_MIN_SHARD", "4096")
    small, k = run()
    assert k == 1
    for key in oh.F64_OUT:
        assert np.array_equal(one[key], small[key])


_COALESCE_CHILD = r"""
import ctypes as C, json, sys, threading
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import oracle_helpers as oh
from roadsurf_amd import abi, lib
from test_hip_boundary import _pointers
L = lib.load()
n, SL, T = 96, 721, 32
f = oh.synth_forcing(n, SL, seed=77)
f["tair"][5, 300] = 250.0          # one point fails CheckValues inside the series
f["vz"][9, 0] = 0.1                # one gets the VZ(1) edit written back
s = abi.default_settings(SL); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
s2 = abi.default_settings(SL); s2.tsurfOutputDepth = 0.05   # a second group of callers: other settings
ora, fm, _ = oh.run_oracle("ref" if oh.have_ref() else "port", f, s, p, l)
ora2, _, _ = oh.run_oracle("ref" if oh.have_ref() else "port", f, s2, p, l)
g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
g2 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
out = {k: np.full((n, SL), np.nan) for k in oh.F64_OUT}
out2 = {k: np.full((n, SL), np.nan) for k in oh.F64_OUT}
ptrs = [_pointers(g, out, pt) for pt in range(n)] + [_pointers(g2, out2, pt) for pt in range(n)]
nxt = [0]; lock = threading.Lock()
def worker():
    while True:
        with lock:
            k = nxt[0]; nxt[0] += 1
        if k >= 2 * n: return
        ip, op, _ = ptrs[k]
        L.runsimulation(C.byref(op), C.byref(ip), C.byref(s if k < n else s2), C.byref(p), C.byref(l))
th = [threading.Thread(target=worker) for _ in range(T)]
[x.start() for x in th]; [x.join() for x in th]
b = C.c_int64(0); q = C.c_int64(0)
L.rs_coalesce_stats(C.byref(b), C.byref(q))
same = all(np.array_equal(out[k], ora[k]) for k in oh.F64_OUT)
same2 = all(np.array_equal(out2[k], ora2[k]) for k in oh.F64_OUT)
print(json.dumps({"same": bool(same), "same2": bool(same2), "batches": b.value, "points": q.value,
                  "vz_edit": bool(g["vz"][9, 0] == fm["vz"][9, 0] == np.float64(np.float32(0.4)))}))
"""


@pytest.mark.parametrize("max_batch", [None, 5, "auto"])
def test_runsimulation_coalesces_concurrent_callers(max_batch):
    """ROADSURF_HIP_COALESCE_US: 32 threads call runsimulation point by point, two groups with different
    settings interleaved - every point carries the reference's bits, the points went through far fewer
    batches than calls, and a batch never mixes settings (the second group's output depth differs).
    max_batch = 5: ROADSURF_HIP_COALESCE_MAX below the thread count - queues longer than a batch, the
    collector's own point must be in the batch it runs."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ROADSURF_HIP_COALESCE_US="3000")
    if max_batch == "auto":  # the default since round 5: a second caller inside a first call switches it on
        env.pop("ROADSURF_HIP_COALESCE_US")
        max_batch = None
        auto = True
    else:
        auto = False
    if max_batch:
        env["ROADSURF_HIP_COALESCE_MAX"] = str(max_batch)
    r = subprocess.run([sys.executable, "-c", _COALESCE_CHILD, root], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["same"] and d["same2"] and d["vz_edit"], d
    if auto:  # the very first caller may have run alone (a batch of one that the statistics do not count)
        assert 180 <= d["points"] <= 192 and d["batches"] < 96, d
        return
    assert d["points"] == 192 and d["batches"] < (96 if not max_batch else 193), d
    if max_batch:
        assert d["batches"] >= 192 // max_batch, d
