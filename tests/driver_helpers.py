"""Helpers for the driver-data-path tests: scenario builder and the CPU checker pipeline
(oracle/driver_oracle.c for the input side, the existing oracles for the simulation)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

import oracle_helpers as oh
from roadsurf_amd import abi, driver

TOOLS_REF_SO = os.path.join(oh.ORACLE_DIR, "_ref", "libroadrunner_tools_ref.so")
START = 1704844800  # 2024-01-10 00:00:00 UTC


def oracle_read_input(sources, settings, start_time, forecast_time, local=None):
    """oracle/driver_oracle.c: same argument meaning and result layout as driver.read_input."""
    lib = oh.load("port")
    inp, keep = driver.make_input(sources, start_time, forecast_time)
    n, L = inp.n_points, settings.SimLen
    larr = driver._locals(n, local)
    merged = np.empty((len(driver.MERGED_FIELDS), n, L), np.float64)
    status = np.empty(n, np.int32)
    mi = np.empty(n, np.int32)
    lib.oracle_driver_expand.argtypes = [C.POINTER(driver.RsDriverInput), C.POINTER(abi.InputSettings),
                                         C.POINTER(abi.LocalParameters), abi.c_double_p,
                                         abi.c_int32_p, abi.c_int32_p]
    rc = lib.oracle_driver_expand(C.byref(inp), C.byref(settings), larr,
                                  merged.ctypes.data_as(abi.c_double_p),
                                  status.ctypes.data_as(abi.c_int32_p), mi.ctypes.data_as(abi.c_int32_p))
    assert rc == 0
    del keep
    return {"merged": {k: merged[i] for i, k in enumerate(driver.MERGED_FIELDS)}, "status": status,
            "missing_index": mi, "local": larr}


def oracle_run(kind, sources, settings, params, start_time, forecast_time, local=None, cal=None,
               horizons=None):
    """read_input (C restatement) -> runsimulation per accepted point (oracle `kind`) ->
    save_output's decimation.  Same result layout as driver.run."""
    ri = oracle_read_input(sources, settings, start_time, forecast_time, local)
    n, L = ri["status"].shape[0], settings.SimLen
    if cal is None:
        cal = driver.calendar(start_time, L, int(settings.DTSecs))
    step, n_out = driver.output_rows(settings)
    res = {k: np.full((n, n_out), -9999.0) for k in driver.OUT_FIELDS}
    ok = np.nonzero(ri["status"] == 0)[0]
    if len(ok):
        f = {k: np.ascontiguousarray(ri["merged"][k][ok]) for k in driver.MERGED_FIELDS}
        f["depth"] = np.full((len(ok), L), -9999.9)          # InputData.cpp:18
        f["precphase"] = np.full((len(ok), L), -9999, np.int32)  # InputData.cpp:16
        f.update({k: np.ascontiguousarray(cal[k], np.int32) for k in driver.CALENDAR})
        if horizons is not None:
            f["local_horizons"] = np.ascontiguousarray(horizons[ok])
        ls = [ri["local"][int(p)] for p in ok]
        out, _, _ = oh.run_oracle(kind, f, settings, params, ls)
        for k in driver.OUT_FIELDS:
            res[k][ok] = out[k][:, ::step]
    res.update(status=ri["status"], missing_index=ri["missing_index"], local=ri["local"], step=step)
    return res


def scenario(n, hours=12, seed=7, obs_hours=6, gaps=True):
    """Two sources over one batch: road-weather-station observations (10-minute data, no Tdew,
    with gaps) for the first `obs_hours`, and an hourly forecast (no RH: Tdew only) that
    starts an hour before the simulation and ends an hour after it."""
    L = hours * 120 + 1
    f = oh.synth_forcing(n, (hours + 2) * 120 + 1, seed=seed)
    rs = np.random.RandomState(seed)
    # forecast: hourly, from -1 h to hours+1 h; the synthetic series is shifted by one hour
    fc_idx = np.arange(0, (hours + 2) * 120 + 1, 120)
    fc_t = START - 3600 + fc_idx.astype(np.int64) * 30
    fc = {k: np.ascontiguousarray(f[k][:, fc_idx]) for k in
          ("tair", "tdew", "vz", "prec", "sw", "lw", "sw_dir", "lw_net")}
    # observations: every 10 minutes from the start to obs_hours
    ob_idx = 120 + np.arange(0, obs_hours * 120 + 1, 20)
    ob_t = START - 3600 + ob_idx.astype(np.int64) * 30
    ob = {k: np.ascontiguousarray(f[k][:, ob_idx]) for k in ("tair", "rhz", "vz", "prec")}
    ob["tair"] = ob["tair"] + 0.25
    ob["tsurfobs"] = np.ascontiguousarray(f["tair"][:, ob_idx] - 0.5 + rs.uniform(-1, 1, (n, 1)))
    if gaps:
        for k in ob:
            m = rs.rand(*ob[k].shape) < 0.08
            ob[k][m] = -9999.9
        ob["tair"][:, 0] = f["tair"][:, ob_idx[0]]          # index 0 stays observed
        ob["tsurfobs"][:, 0] = f["tair"][:, ob_idx[0]] - 0.5
        # some points lose the tail of their observations
        for p in range(0, n, 7):
            cut = rs.randint(len(ob_idx) // 2, len(ob_idx))
            for k in ob:
                ob[k][p, cut:] = -9999.9
    # a few points with a hole in the forecast: read_input rejects them
    names = ("tair", "prec", "sw", "lw", "vz")
    for j, p in enumerate(range(5, n, 29)):
        fc[names[j % 5]][p, 3 + j % 4] = -9999.9
    src = [driver.RawSource(fc_t, fc, False), driver.RawSource(ob_t, ob, True)]
    return src, L, START, START + obs_hours * 3600


def ragged(src: driver.RawSource, seed=3, drop=0.25, pad_value=-9999.9):
    """Per-point time axes from a shared-axis source: every point loses a random subset of its
    time stamps (the station did not report then), rows are compacted and padded to a common
    width, like the per-station "time" arrays of the reference's JSON input."""
    rs = np.random.RandomState(seed)
    t = np.asarray(src.times, np.int64)
    n = next(iter(src.fields.values())).shape[0]
    nt = t.shape[0]
    keep = rs.rand(n, nt) >= drop
    keep[:, 0] |= rs.rand(n) < 0.7          # most series keep their first stamp
    keep[n // 3] = True                      # one complete series
    keep[n // 2] = False                     # one empty series
    if n > 5:
        keep[5] = False
        keep[5, nt // 2] = True             # one series with a single stamp
    lengths = keep.sum(axis=1).astype(np.int32)
    width = int(max(1, lengths.max()))
    times = np.full((n, width), np.iinfo(np.int64).min, np.int64)
    fields = {k: np.full((n, width), pad_value) for k in src.fields}
    for p in range(n):
        idx = np.nonzero(keep[p])[0]
        times[p, :len(idx)] = t[idx]
        for k, a in src.fields.items():
            fields[k][p, :len(idx)] = a[p, idx]
    return driver.RawSource(times, fields, src.is_observation, lengths)


# ---- the reference's operational shape (examples/example1/example_config.json:8-22) -----------------
OPER_ANALYSIS_H, OPER_FORECAST_H = 48, 26
OPER_SEED = 2026


def operational_case(z, case: str):
    """The run the reference's example configuration describes: 48 h analysis + 26 h forecast
    (SimLen 8 881 at DTSecs 30), coupling and relaxation on, InitLenI / couplingIndexI at the end of the
    analysis (forecast_time = start + 48 h), the stations of example_skyview.txt.  `z` = the fixture
    tests/golden/e2e_operational.npz (latitudes, longitudes, sky-view factors and local horizons are the
    NUMBERS of the reference's two data files; the raw series are this module's seeded scenario).
    case "files": sky view and horizons as the files hold them (all 1.0 / 0.0: the branch is never taken);
    case "sky": synthetic sky-view factors and horizons on the same stations.
    Returns (sources, settings, params, start, forecast_time, local, horizons)."""
    lat, lon = z["lat"], z["lon"]
    n = len(lat)
    hours = OPER_ANALYSIS_H + OPER_FORECAST_H
    src, L, t0, tf = scenario(n, hours=hours, seed=OPER_SEED, obs_hours=OPER_ANALYSIS_H)
    assert L == hours * 120 + 1 == 8881
    s = abi.default_settings(L)
    s.use_relaxation = 1
    s.use_coupling = 1
    p = abi.default_parameters()
    sv = z["sky_view_files"] if case == "files" else z["sky_view_sky"]
    hz = (z["horizons_files_tenths"] if case == "files" else z["horizons_sky_tenths"]).astype(np.float64) / 10.0
    local = []
    for i in range(n):
        lp = abi.default_local()
        lp.lat, lp.lon, lp.sky_view = float(lat[i]), float(lon[i]), float(sv[i])
        local.append(lp)
    return src, s, p, t0, tf, local, np.ascontiguousarray(hz)
