"""GPU: sky-view / local-horizon radiation (src/ModRadiation.f90, src/SunPosition.f90),
alone and together with coupling."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_helpers as oh
from roadsurf_amd import abi, lib

pytestmark = pytest.mark.gpu


def _sky_case(n, L, seed, summer=False, world=False):
    rs = np.random.RandomState(seed)
    f = oh.synth_forcing(n, L, seed=seed)
    if summer:
        f.update(oh.time_axis(L, 30.0, (2024, 6, 20, 0, 0, 0)))
    f["sw"] *= 2.0
    f["sw_dir"] = np.ascontiguousarray(f["sw"] * rs.uniform(0.2, 1.2, (n, 1)))  # some SW_dir > SW: clamp
    f["lw_net"] = np.ascontiguousarray(-40.0 - 30 * rs.rand(n, L))
    hz = np.ascontiguousarray(np.round(rs.uniform(0, 25, (n, 360)), 1)); hz[::7] = 0.0
    f["local_horizons"] = hz
    ls = []
    for i in range(n):
        li = abi.default_local(); li.InitLenI = 1
        li.lat = float(rs.uniform(-70, 70) if world else rs.uniform(59, 70))
        li.lon = float(rs.uniform(-180, 180) if world else rs.uniform(19, 31))
        li.sky_view = float(rs.choice([0.0, 0.3, 0.75, 0.99, 1.0]))
        ls.append(li)
    return f, ls


@pytest.mark.parametrize("summer,world", [(False, False), (True, False), (True, True)])
def test_sky_view_matches_reference_bitwise(summer, world):
    from roadsurf_amd import device
    n, L = 256, 2881
    f, ls = _sky_case(n, L, 99, summer, world)
    s = abi.default_settings(L); p = abi.default_parameters()
    ora, _, _ = oh.run_oracle("ref" if oh.have_ref() else "port", f, s, p, ls)
    res, _ = device.run_points(f, s, p, ls)
    for k in oh.F64_OUT:
        assert np.abs(res[k] - ora[k]).max() < 1e-6, k
        assert np.array_equal(res[k], ora[k]), k
    plain, _, _ = oh.run_oracle("port", f, s, p, abi.default_local())
    assert np.abs(plain["tsurf"] - ora["tsurf"]).max() > 0.5  # the branch really changes the result


def test_sky_view_with_coupling_and_c_abi():
    L_ = lib.load()
    n, SL = 200, 1441
    f, ls = _sky_case(n, SL, 7, summer=True)
    p = abi.default_parameters()
    base, _, _ = oh.run_oracle("port", f, abi.default_settings(SL), p, ls)
    rs = np.random.RandomState(1)
    for i, li in enumerate(ls):
        li.couplingIndexI = 900; li.InitLenI = 900
        li.couplingTsurf = float(base["tsurf"][i, 899] + rs.choice([0.0, 1.0, -2.0, 5.0]))
    f["tsurfobs"][:, :] = base["tsurf"] + 0.2
    s = abi.default_settings(SL); s.use_coupling = 1
    kind = "ref_cpl" if os.path.exists(oh.REF_CPL_SO) else "port"
    ora, _, _ = oh.run_oracle(kind, f, s, p, ls)
    g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
    out = {k: np.full((n, SL), np.nan) for k in oh.F64_OUT}
    ips = (abi.InputPointers * n)(); ops = (abi.OutputPointers * n)(); keep = []
    for pt in range(n):
        ip, op, kp = oh.point_pointers(g, pt, out)
        hzrow = np.ascontiguousarray(g["local_horizons"][pt])
        ip.c_local_horizons = hzrow.ctypes.data_as(abi.c_double_p)
        ips[pt], ops[pt] = ip, op
        keep.append((kp, hzrow))
    larr = (abi.LocalParameters * n)(*ls)
    st = C.c_int32(99)
    L_.runsimulation_batch(n, ops, ips, C.byref(s), C.byref(p), larr, C.byref(st))
    assert st.value == 0, lib.last_error()
    for k in oh.F64_OUT:
        assert np.array_equal(out[k], ora[k]), k


@pytest.mark.parametrize("chunk,windows,nlayers", [(97, "spread", 15), (400, "spread", 15), (400, "compact", 15),
                                                   (400, "compact", 12)])
def test_sky_view_with_coupling_time_chunked(chunk, windows, nlayers):
    """Sky view and coupling together through the lock-step kernels (rs_hip_step_cpl + the replay rounds
    of rs_hip_cpl_replay, time_loop<SKY, CPL>): coupling windows that end at different indices, points
    with and without a sky view, launch boundaries inside the windows - against the reference.  compact: every
    window ends within a few indices of the others, so the replay rounds run in LOCK STEP over the list
    (step_kernel_cpl_replay_h<3, true>; with another layer count the LDS-profile twin step_kernel_cpl_replay<true>)
    instead of the per-lane kernel - unreached by any test before round 6 (profiles/r06_kernel_reachability.txt)."""
    from roadsurf_amd import device
    n, SL = 200, 1441
    f, ls = _sky_case(n, SL, 11, summer=True)
    p = abi.default_parameters()
    s0 = abi.default_settings(SL); s0.NLayers = nlayers
    base, _, _ = oh.run_oracle("port", f, s0, p, ls)
    rs = np.random.RandomState(3)
    for i, li in enumerate(ls):
        ce = int(rs.choice([700, 900, 905] if windows == "spread" else [900, 902, 905]))
        li.couplingIndexI = ce; li.InitLenI = ce
        li.couplingTsurf = float(base["tsurf"][i, ce - 1] + rs.choice([0.0, 1.0, -2.0, 5.0]))
    f["tsurfobs"][:, :] = base["tsurf"] + 0.2
    s = abi.default_settings(SL); s.use_coupling = 1; s.NLayers = nlayers
    kind = "ref_cpl" if os.path.exists(oh.REF_CPL_SO) else "port"
    ora, _, _ = oh.run_oracle(kind, f, s, p, ls)
    whole, _ = device.run_points(f, s, p, ls)
    parts, _ = device.run_points(f, s, p, ls, chunk=chunk)
    for k in oh.F64_OUT:
        assert np.array_equal(whole[k], ora[k]), k
        assert np.array_equal(parts[k], ora[k]), k
    off = abi.default_settings(SL); off.NLayers = nlayers
    plain, _, _ = oh.run_oracle("port", f, off, p, ls)
    assert np.abs(plain["tsurf"] - ora["tsurf"]).max() > 0.3  # coupling really acts on this case


def _batch_arrays(g, out, n):
    ips = (abi.InputPointers * n)(); ops = (abi.OutputPointers * n)(); keep = []
    for pt in range(n):
        ip, op, kp = oh.point_pointers(g, pt, out)
        hzrow = np.ascontiguousarray(g["local_horizons"][pt])
        ip.c_local_horizons = hzrow.ctypes.data_as(abi.c_double_p)
        ips[pt], ops[pt] = ip, op
        keep.append((kp, hzrow))
    return ips, ops, keep


@pytest.mark.parametrize("coupled", [False, True], ids=["plain", "coupling"])
def test_writeback_of_the_in_place_input_edits(coupled, monkeypatch):
    """The reference edits the caller's input arrays (Ownership): SW_dir is clamped to
    SW at every checked index (src/InputOutput.f90:75-77) and the sky-view correction rewrites SW,
    SW_dir and LW (src/ModRadiation.f90:57-71); a point that fails keeps its arrays from there on.
    With ROADSURF_HIP_WRITEBACK=1 runsimulation_batch leaves the same bits in those arrays as the
    reference's run does (the harness hands its mutated inputs back); without it they are untouched."""
    L_ = lib.load()
    n, SL = 96, 1441
    f, ls = _sky_case(n, SL, 21, summer=True)
    f["tair"][5, 700] = -200.0          # fails at index 701: arrays beyond stay as they were
    ls[7].sky_view = 1.0                # no sky view for this one: only the SW_dir clamp
    p = abi.default_parameters()
    s = abi.default_settings(SL)
    kind = "ref" if oh.have_ref() else "port"
    if coupled:
        base, _, _ = oh.run_oracle("port", f, s, p, ls)
        rs = np.random.RandomState(3)
        for i, li in enumerate(ls):
            li.couplingIndexI = 900; li.InitLenI = 900
            li.couplingTsurf = float(base["tsurf"][i, 899] + rs.choice([0.0, 1.5, -2.0]))
        f["tsurfobs"][:, :] = np.where(base["tsurf"] > -9000, base["tsurf"] + 0.2, -9999.9)
        s = abi.default_settings(SL); s.use_coupling = 1
        kind = "ref_cpl" if os.path.exists(oh.REF_CPL_SO) else "port"
    ora, fmut, _ = oh.run_oracle(kind, f, s, p, ls)
    assert (fmut["sw"] != f["sw"]).mean() > 0.05 and (fmut["sw_dir"] != f["sw_dir"]).mean() > 0.05

    def run():
        g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
        out = {k: np.full((n, SL), np.nan) for k in oh.F64_OUT}
        ips, ops, keep = _batch_arrays(g, out, n)
        larr = (abi.LocalParameters * n)(*ls)
        st = C.c_int32(99)
        ff = np.full(n, -1, np.int32)
        L_.runsimulation_batch_ex(n, ops, ips, C.byref(s), C.byref(p), larr, C.byref(st),
                                  C.c_void_p(ff.ctypes.data))
        assert st.value == 0, lib.last_error()
        return g, out, ff

    monkeypatch.setenv("ROADSURF_HIP_WRITEBACK", "1")
    monkeypatch.setenv("ROADSURF_HIP_TILE_POINTS", "40")   # three tiles
    g, out, ff = run()
    for k in oh.F64_OUT:
        assert np.array_equal(out[k], ora[k]), k
    for k in ("sw", "sw_dir", "lw", "vz"):
        assert np.array_equal(g[k], fmut[k]), (k, int((g[k] != fmut[k]).sum()))
    for k in ("tair", "rhz", "prec", "lw_net", "tsurfobs"):
        assert np.array_equal(g[k], f[k]), k
    want = np.zeros(n, np.int32); want[5] = 701
    assert np.array_equal(ff, want)
    monkeypatch.setenv("ROADSURF_HIP_WRITEBACK", "0")
    g, out, _ = run()
    for k in ("sw", "sw_dir", "lw"):
        assert np.array_equal(g[k], f[k]), k


def test_writeback_of_the_sw_dir_clamp_without_sky_view(monkeypatch):
    L_ = lib.load()
    n, SL = 50, 721
    f = oh.synth_forcing(n, SL, seed=12)
    rs = np.random.RandomState(4)
    f["sw"] += 50.0
    f["sw_dir"] = np.ascontiguousarray(f["sw"] * rs.uniform(0.5, 1.5, (n, SL)))
    f["local_horizons"] = np.zeros((n, 360))
    f["tair"][3, 100] = 150.0           # fails at index 101
    f["rhz"][9, 0] = 130.0              # fails at index 1
    s = abi.default_settings(SL); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    ora, fmut, _ = oh.run_oracle("ref" if oh.have_ref() else "port", f, s, p, l)
    monkeypatch.setenv("ROADSURF_HIP_WRITEBACK", "1")
    g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
    out = {k: np.full((n, SL), np.nan) for k in oh.F64_OUT}
    ips, ops, keep = _batch_arrays(g, out, n)
    larr = (abi.LocalParameters * n)(*([l] * n))
    st = C.c_int32(99)
    ff = np.full(n, -1, np.int32)
    L_.runsimulation_batch_ex(n, ops, ips, C.byref(s), C.byref(p), larr, C.byref(st), C.c_void_p(ff.ctypes.data))
    assert st.value == 0, lib.last_error()
    for k in oh.F64_OUT:
        assert np.array_equal(out[k], ora[k]), k
    assert np.array_equal(g["sw_dir"], fmut["sw_dir"]) and (fmut["sw_dir"] != f["sw_dir"]).mean() > 0.2
    assert ff[3] == 101 and ff[9] == 1 and (np.delete(ff, [3, 9]) == 0).all()
    # the same numbers from the reference's outputs: the first index it left at -9999.0
    first_missing = np.where((ora["tsurf"] == -9999.0).any(1), (ora["tsurf"] == -9999.0).argmax(1), 0)
    assert np.array_equal(ff, first_missing.astype(np.int32))


def test_writeback_is_the_default_of_the_one_point_entry(monkeypatch):
    """`runsimulation` is the reference's own entry: with NO environment variable set it leaves the caller's
    input arrays as the reference's run leaves them - the SW_dir clamp (src/InputOutput.f90:75-77), the sky
    view's SW / SW_dir / LW (src/ModRadiation.f90:57-71), VZ(1) - for a point with sky view, one without, and
    one that fails half way (its arrays stay as they were from there on).  ROADSURF_HIP_WRITEBACK=0 opts out;
    the batch entries keep them out unless asked (test_writeback_of_the_in_place_input_edits)."""
    L_ = lib.load()
    n, SL = 6, 1441
    f, ls = _sky_case(n, SL, 33, summer=True)
    ls[0].sky_view = 0.5
    ls[1].sky_view = 1.0                # no sky view: only the SW_dir clamp
    ls[2].sky_view = 0.75
    f["tair"][2, 500] = -200.0          # fails at index 501
    f["vz"][3, 0] = 0.1
    s = abi.default_settings(SL); p = abi.default_parameters()
    ora, fmut, _ = oh.run_oracle("ref" if oh.have_ref() else "port", f, s, p, ls)
    assert (fmut["sw"] != f["sw"]).any() and (fmut["sw_dir"] != f["sw_dir"]).any() and (fmut["lw"] != f["lw"]).any()

    def run():
        g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
        out = {k: np.full((n, SL), np.nan) for k in oh.F64_OUT}
        for pt in range(n):
            ip, op, kp = oh.point_pointers(g, pt, out)
            hzrow = np.ascontiguousarray(g["local_horizons"][pt])
            ip.c_local_horizons = hzrow.ctypes.data_as(abi.c_double_p)
            L_.runsimulation(C.byref(op), C.byref(ip), C.byref(s), C.byref(p), C.byref(ls[pt]))
        return g, out

    monkeypatch.delenv("ROADSURF_HIP_WRITEBACK", raising=False)
    g, out = run()
    for k in oh.F64_OUT:
        assert np.array_equal(out[k], ora[k]), k
    for k in ("sw", "sw_dir", "lw", "vz"):
        assert np.array_equal(g[k], fmut[k]), (k, int((g[k] != fmut[k]).sum()))
    for k in ("tair", "rhz", "prec", "lw_net", "tsurfobs"):
        assert np.array_equal(g[k], f[k]), k
    monkeypatch.setenv("ROADSURF_HIP_WRITEBACK", "0")
    g, out = run()
    for k in oh.F64_OUT:
        assert np.array_equal(out[k], ora[k]), k
    for k in ("sw", "sw_dir", "lw"):
        assert np.array_equal(g[k], f[k]), k
    assert np.array_equal(g["vz"], fmut["vz"])  # VZ(1): always


def test_sky_view_with_two_time_axes_in_one_batch():
    """The reference takes the solar position from each point's OWN year(i)..second(i)
    (src/SunPosition.f90:196-260): a batch may mix points whose series start on different dates.
    runsimulation_batch_ex groups the points by time axis (one sun table and one device call per
    distinct axis; axes given as separate but equal arrays count as one): winter and midsummer
    points interleaved in one batch equal the reference run of each group on its own axis, the
    failure indices included."""
    L_ = lib.load()
    n, SL = 192, 1441
    f, ls = _sky_case(n, SL, 31, summer=False)
    axis_w = {k: f[k] for k in oh.I32_AXIS}
    axis_s = oh.time_axis(SL, 30.0, (2024, 6, 20, 3, 0, 0))
    summer = (np.arange(n) % 3 == 1)           # interleaved: the groups are not contiguous
    f["tair"][5, 700] = 150.0                    # one failing point in each group
    f["tair"][7, 900] = -150.0
    assert not summer[5] and summer[7]
    s = abi.default_settings(SL); p = abi.default_parameters()
    kind = "ref" if oh.have_ref() else "port"
    want = {k: np.full((n, SL), np.nan) for k in oh.F64_OUT}
    for grp, axis in ((~summer, axis_w), (summer, axis_s)):
        idx = np.nonzero(grp)[0]
        g = {k: (np.ascontiguousarray(v[idx]) if isinstance(v, np.ndarray) and v.ndim == 2 and v.shape[0] == n else v)
             for k, v in f.items()}
        g.update(axis)
        o, _, _ = oh.run_oracle(kind, g, s, p, [ls[i] for i in idx])
        for k in oh.F64_OUT:
            want[k][idx] = o[k]
    out = {k: np.full((n, SL), np.nan) for k in oh.F64_OUT}
    g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
    ips, ops, keep = _batch_arrays(g, out, n)
    # every summer point gets its OWN copy of the summer axis (equal values, different arrays)
    axes = []
    for pt in np.nonzero(summer)[0]:
        own = {k: np.ascontiguousarray(v.copy()) for k, v in axis_s.items()}
        axes.append(own)
        for k in oh.I32_AXIS:
            setattr(ips[pt], "c_" + k, own[k].ctypes.data_as(abi.c_int32_p))
    larr = (abi.LocalParameters * n)(*ls)
    st = C.c_int32(99)
    ff = np.full(n, -1, np.int32)
    L_.runsimulation_batch_ex(n, ops, ips, C.byref(s), C.byref(p), larr, C.byref(st),
                              ff.ctypes.data_as(abi.c_int32_p))
    assert st.value == 0, lib.last_error()
    for k in oh.F64_OUT:
        assert np.array_equal(out[k], want[k]), k
    assert ff[5] == 701 and ff[7] == 901 and (np.delete(ff, [5, 7]) == 0).all()
