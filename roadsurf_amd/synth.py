"""Host arrays of the BASELINE synthetic workload, in the layout the reference driver hands
``runsimulation`` (per-point ``[n][SimLen]`` series, one shared time axis): the host twin of the device
generator, ``rs_synth_fill_points`` (csrc/rs_synth_host.hip over csrc/rs_synth.h - the arithmetic the HIP
kernels compile).  bench.py's host-array leg, ``__graft_entry__.smoke()`` and the parity tests take their
inputs from here; the CPU checker only ever *checks* them."""
from __future__ import annotations

import ctypes as C
import datetime as dt

import numpy as np

from . import abi, lib

F64_IN = ("tair", "tdew", "vz", "rhz", "prec", "sw", "lw", "sw_dir", "lw_net", "tsurfobs", "depth")
I32_AXIS = ("year", "month", "day", "hour", "minute", "second")
F64_OUT = ("tsurf", "snow", "water", "ice", "deposit", "ice2")


def time_axis(simlen: int, dtsecs: float = 30.0, start=(2024, 1, 10, 0, 0, 0)) -> dict[str, np.ndarray]:
    """Shared time axis, all points (start 2024-01-10 00:00)."""
    t0 = dt.datetime(*start)
    ax = {k: np.empty(simlen, np.int32) for k in I32_AXIS}
    for i in range(simlen):
        t = t0 + dt.timedelta(seconds=i * dtsecs)
        ax["year"][i], ax["month"][i], ax["day"][i] = t.year, t.month, t.day
        ax["hour"][i], ax["minute"][i], ax["second"][i] = t.hour, t.minute, t.second
    return ax


def synth_forcing(n: int, simlen: int, seed: int = 1234, point_offset: int = 0,
                  steps_per_knot: int = 120, start_hour: int = 0) -> dict[str, np.ndarray]:
    """Per-point ``[n][simlen]`` arrays of the synthetic workload + the shared calendar arrays."""
    L = lib.load()
    f = {k: np.empty((n, simlen), np.float64) for k in F64_IN}
    f["precphase"] = np.empty((n, simlen), np.int32)
    hour = np.empty(simlen, np.int32)
    L.rs_synth_fill_points.restype = None
    L.rs_synth_fill_points.argtypes = (
        [C.c_uint64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        + [abi.c_double_p] * 11 + [abi.c_int32_p, abi.c_int32_p]
    )
    L.rs_synth_fill_points(
        seed, point_offset, n, simlen, steps_per_knot, start_hour,
        *[f[k].ctypes.data_as(abi.c_double_p) for k in F64_IN],
        f["precphase"].ctypes.data_as(abi.c_int32_p), hour.ctypes.data_as(abi.c_int32_p),
    )
    ax = time_axis(simlen, 3600.0 / steps_per_knot, (2024, 1, 10, start_hour, 0, 0))
    assert np.array_equal(ax["hour"], hour)
    f.update(ax)
    f["hour"] = hour
    return f


def point_pointers(f: dict, p: int, out: dict | None = None):
    """InputPointers / OutputPointers of point ``p`` of reference-layout arrays, as the reference driver builds
    them per point (examples/example1/src/roadrunner.cpp:404-406).  Returns (ip, op, keep-alive)."""
    n, L = f["tair"].shape
    ip = abi.InputPointers()
    ip.inputLen = L
    for name, key in (("c_tair", "tair"), ("c_tdew", "tdew"), ("c_VZ", "vz"), ("c_Rhz", "rhz"),
                      ("c_prec", "prec"), ("c_SW", "sw"), ("c_LW", "lw"), ("c_SW_dir", "sw_dir"),
                      ("c_LW_net", "lw_net"), ("c_TSurfObs", "tsurfobs"), ("c_Depth", "depth")):
        setattr(ip, name, f[key][p].ctypes.data_as(abi.c_double_p))
    ip.c_PrecPhase = f["precphase"][p].ctypes.data_as(abi.c_int32_p)
    hz = f["local_horizons"][p] if f.get("local_horizons") is not None else np.zeros(360)
    ip.c_local_horizons = hz.ctypes.data_as(abi.c_double_p)
    for name in I32_AXIS:
        setattr(ip, "c_" + name, f[name].ctypes.data_as(abi.c_int32_p))
    if out is None:
        out = {k: np.full((1, L), np.nan) for k in F64_OUT}
        row = 0
    else:
        row = p
    op = abi.OutputPointers()
    op.outputLen = L
    for name, key in (("c_TsurfOut", "tsurf"), ("c_SnowOut", "snow"), ("c_WaterOut", "water"),
                      ("c_IceOut", "ice"), ("c_DepositOut", "deposit"), ("c_Ice2Out", "ice2")):
        setattr(op, name, out[key][row].ctypes.data_as(abi.c_double_p))
    return ip, op, (hz, out)
