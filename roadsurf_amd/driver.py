"""Host mirror of the reference driver's data path (layer 4 of ``include/roadsurf.h``).

The names follow ``examples/example1/src``: a *source* is what ``JsonSource`` holds after
parsing (raw series on the source's own time axis), ``read_input`` is what
``roadrunner.cpp:156-278`` returns for a point (series at simulation resolution plus the
relaxation / coupling decisions) and ``run`` is ``read_input`` + ``runsimulation`` +
``save_output``'s decimation for a whole batch, with the per-value work on the GPU
(``roadsurf_amd/csrc/rs_driver.hip``).  Only file formats are handled here on the host:
``read_json_source`` / ``save_output`` read and write the reference's JSON schema
(``JsonSource.cpp:182-316``, ``roadrunner.cpp:285-347``).
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import json
import time as _time

import numpy as np

from . import abi
from . import lib as rslib

RS_MAX_SOURCES = 4
#: order of ``merged`` in rs_driver_expand
MERGED_FIELDS = ("tair", "tdew", "vz", "rhz", "prec", "sw", "lw", "sw_dir", "lw_net", "tsurfobs")
#: member order of RsRawSource
RAW_FIELDS = ("tair", "rhz", "tdew", "vz", "prec", "lw_net", "lw", "sw", "sw_dir", "tsurfobs")
OUT_FIELDS = ("tsurf", "snow", "water", "ice", "deposit", "ice2")
CALENDAR = ("year", "month", "day", "hour", "minute", "second")
#: variable names of the reference's JSON input (JsonSource.cpp:192-194) -> our field names;
#: "PrecipitationForm" is read by the reference but never handed on (JsonSource.cpp:323-373)
JSON_VARIABLES = {
    "Temperature 2m": "tair", "Humidity": "rhz", "DewPoint": "tdew", "WindSpeed": "vz",
    "Precipitation": "prec", "RadiationNetSurfaceLW": "lw_net", "RadiationLW": "lw",
    "RadiationGlobal": "sw", "RadiationDirectSW": "sw_dir", "RoadTemperature": "tsurfobs",
}

c_int64_p = C.POINTER(C.c_int64)


class RsRawSource(C.Structure):
    _fields_ = [("n_times", C.c_int32), ("is_observation", C.c_int32), ("times", c_int64_p)] + [
        (n, abi.c_double_p) for n in RAW_FIELDS
    ] + [("times_per_point", C.c_int32), ("lengths", abi.c_int32_p)]


class RsDriverInput(C.Structure):
    _fields_ = [
        ("n_points", C.c_int32), ("n_sources", C.c_int32), ("sources", C.POINTER(RsRawSource)),
        ("start_time", C.c_int64), ("forecast_time", C.c_int64),
    ] + [(n, abi.c_int32_p) for n in CALENDAR] + [("horizons", abi.c_double_p)]


class RsDriverOutput(C.Structure):
    _fields_ = [("n_out", C.c_int32)] + [(n, abi.c_double_p) for n in OUT_FIELDS] + [
        ("status", abi.c_int32_p), ("missing_index", abi.c_int32_p)
    ]


@dataclasses.dataclass
class RawSource:
    """One data source: ``fields`` name -> [n_points][n_times] float64 (absent name = variable
    not in the source) and ``times`` epoch seconds, either [n_times] shared by all points or
    [n_points][n_times] with one axis per point; ``lengths`` [n_points] then says how many
    leading entries of each row are real (rows are padded to a common width)."""
    times: np.ndarray
    fields: dict
    is_observation: bool = False
    lengths: np.ndarray | None = None


def calendar(start_time: int, simlen: int, dtsecs: int, utc: bool = True) -> dict:
    """Calendar arrays of the simulation times (JsonSource.cpp:297-308 uses localtime)."""
    conv = _time.gmtime if utc else _time.localtime
    ax = {k: np.empty(simlen, np.int32) for k in CALENDAR}
    for i in range(simlen):
        tt = conv(start_time + i * dtsecs)
        ax["year"][i], ax["month"][i], ax["day"][i] = tt.tm_year, tt.tm_mon, tt.tm_mday
        ax["hour"][i], ax["minute"][i], ax["second"][i] = tt.tm_hour, tt.tm_min, tt.tm_sec
    return ax


def output_rows(settings: abi.InputSettings) -> tuple[int, int]:
    """(step, n_out) of save_output, roadrunner.cpp:290,303."""
    step = int(settings.outputStep * 60 / settings.DTSecs)
    if step < 1:
        raise ValueError("outputStep*60/DTSecs < 1")
    return step, (settings.SimLen + step - 1) // step


def make_input(sources, start_time: int, forecast_time: int, cal: dict | None = None,
               horizons: np.ndarray | None = None):
    """Build the C struct.  Returns (RsDriverInput, keepalive list)."""
    if not 1 <= len(sources) <= RS_MAX_SOURCES:
        raise ValueError(f"1..{RS_MAX_SOURCES} sources")
    keep = []
    n_points = None
    arr = (RsRawSource * len(sources))()
    for k, s in enumerate(sources):
        t = np.ascontiguousarray(s.times, np.int64)
        keep.append(t)
        width = t.shape[-1]
        arr[k].n_times = width
        arr[k].is_observation = 1 if s.is_observation else 0
        arr[k].times = t.ctypes.data_as(c_int64_p)
        arr[k].times_per_point = 1 if t.ndim == 2 else 0
        if t.ndim == 2:
            if n_points is None:
                n_points = t.shape[0]
            elif t.shape[0] != n_points:
                raise ValueError("all sources must hold the same points")
            if s.lengths is not None:
                ln = np.ascontiguousarray(s.lengths, np.int32)
                if ln.shape != (t.shape[0],) or ln.min() < 0 or ln.max() > width:
                    raise ValueError("lengths: [n_points] values in 0..n_times")
                keep.append(ln)
                arr[k].lengths = ln.ctypes.data_as(abi.c_int32_p)
        elif s.lengths is not None:
            raise ValueError("lengths needs per-point time axes")
        for name, a in s.fields.items():
            if name not in RAW_FIELDS:
                raise KeyError(name)
            a = np.ascontiguousarray(a, np.float64)
            if a.ndim != 2 or a.shape[1] != width:
                raise ValueError(f"{name}: expected [n_points][{width}], got {a.shape}")
            if n_points is None:
                n_points = a.shape[0]
            elif a.shape[0] != n_points:
                raise ValueError("all sources must hold the same points")
            keep.append(a)
            setattr(arr[k], name, a.ctypes.data_as(abi.c_double_p))
    if n_points is None:
        raise ValueError("no data")
    inp = RsDriverInput()
    inp.n_points = n_points
    inp.n_sources = len(sources)
    inp.sources = arr
    inp.start_time = int(start_time)
    inp.forecast_time = int(forecast_time)
    keep.append(arr)
    if cal is not None:
        for k in CALENDAR:
            a = np.ascontiguousarray(cal[k], np.int32)
            keep.append(a)
            setattr(inp, k, a.ctypes.data_as(abi.c_int32_p))
    if horizons is not None:
        h = np.ascontiguousarray(horizons, np.float64)
        if h.shape != (n_points, 360):
            raise ValueError("horizons: [n_points][360]")
        keep.append(h)
        inp.horizons = h.ctypes.data_as(abi.c_double_p)
    return inp, keep


def _locals(n: int, local) -> C.Array:
    """LocalParameters[n]: None -> defaults, one struct -> replicated, a list, or a ready
    ctypes array (used as is: it is also where the decisions are written back)."""
    if isinstance(local, C.Array):
        if len(local) != n:
            raise ValueError("local: wrong length")
        return local
    if local is None:
        local = abi.default_local()
    if isinstance(local, abi.LocalParameters):
        arr = (abi.LocalParameters * n)()
        tmpl = bytes(local)
        C.memmove(arr, tmpl * n, len(tmpl) * n)
        return arr
    return (abi.LocalParameters * n)(*local)


def _bind(L):
    P = C.POINTER
    L.rs_driver_run.argtypes = [P(RsDriverInput), P(abi.InputSettings), P(abi.InputParameters),
                                P(abi.LocalParameters), P(RsDriverOutput), C.c_int32]
    L.rs_driver_expand.argtypes = [P(RsDriverInput), P(abi.InputSettings), P(abi.LocalParameters),
                                   abi.c_double_p, abi.c_int32_p, abi.c_int32_p, C.c_int32]
    return L


def read_input(sources, settings: abi.InputSettings, start_time: int, forecast_time: int,
               local=None, device: int = 0) -> dict:
    """What ``read_input`` (roadrunner.cpp:156-278) produces for every point, computed on the
    GPU: ``merged`` name -> [n][SimLen], ``status``, ``missing_index``, ``local``."""
    L = _bind(rslib.load())
    inp, keep = make_input(sources, start_time, forecast_time)
    n, simlen = inp.n_points, settings.SimLen
    larr = _locals(n, local)
    merged = np.empty((len(MERGED_FIELDS), n, simlen), np.float64)
    status = np.empty(n, np.int32)
    mi = np.empty(n, np.int32)
    rslib.check(L.rs_driver_expand(C.byref(inp), C.byref(settings), larr,
                                   merged.ctypes.data_as(abi.c_double_p),
                                   status.ctypes.data_as(abi.c_int32_p),
                                   mi.ctypes.data_as(abi.c_int32_p), device), "rs_driver_expand")
    del keep
    return {"merged": {k: merged[i] for i, k in enumerate(MERGED_FIELDS)}, "status": status,
            "missing_index": mi, "local": larr}


def run(sources, settings: abi.InputSettings, params: abi.InputParameters, start_time: int,
        forecast_time: int, local=None, cal: dict | None = None,
        horizons: np.ndarray | None = None, device: int = 0, out: dict | None = None) -> dict:
    """read_input + runsimulation + save_output's decimation for all points.  Returns the six
    outputs as [n][n_out] arrays plus ``status``, ``missing_index``, ``local`` and ``step``.
    ``device`` < 0 fans the points out over ROADSURF_HIP_DEVICES; ``out`` = a result dict of an
    earlier call with the same shapes, whose arrays are written again (no fresh allocation)."""
    L = _bind(rslib.load())
    if cal is None:
        cal = calendar(start_time, settings.SimLen, int(settings.DTSecs))
    inp, keep = make_input(sources, start_time, forecast_time, cal, horizons)
    n = inp.n_points
    step, n_out = output_rows(settings)
    larr = _locals(n, local)
    if out is not None and out["tsurf"].shape == (n, n_out):
        res = {k: out[k] for k in OUT_FIELDS + ("status", "missing_index")}
    else:
        res = {k: np.full((n, n_out), np.nan) for k in OUT_FIELDS}
        res["status"] = np.empty(n, np.int32)
        res["missing_index"] = np.empty(n, np.int32)
    out = RsDriverOutput()
    out.n_out = n_out
    for k in OUT_FIELDS:
        setattr(out, k, res[k].ctypes.data_as(abi.c_double_p))
    out.status = res["status"].ctypes.data_as(abi.c_int32_p)
    out.missing_index = res["missing_index"].ctypes.data_as(abi.c_int32_p)
    rslib.check(L.rs_driver_run(C.byref(inp), C.byref(settings), C.byref(params), larr,
                                C.byref(out), device), "rs_driver_run")
    del keep
    res["local"] = larr
    res["step"] = step
    return res


# ---- file formats (host only) ------------------------------------------------------------

def read_json_source(path: str, is_observation: bool = False, utc: bool = True):
    """Parse one input file of the reference's JSON schema (JsonSource.cpp:206-286): a list of
    stations ``{"statId", "lat", "lon", "time": ["%Y-%m-%d %H:%M", ...], "<variable>": [...]}``.
    Returns (RawSource, station ids, lats, lons).  If all stations carry the same time stamps
    the source gets one shared axis, otherwise per-point axes (padded rows + lengths); values
    absent or null become -9999.9."""
    import calendar as _cal

    with open(path) as fh:
        stations = json.load(fh)
    ids, lats, lons, axes = [], [], [], []
    cols = {v: [] for v in JSON_VARIABLES.values()}
    present = set()
    for st in stations:
        tt = []
        for s in st.get("time", []):
            tm = _time.strptime(s, "%Y-%m-%d %H:%M")
            tt.append(_cal.timegm(tm) if utc else int(_time.mktime(tm)))
        axes.append(np.asarray(tt, np.int64))
        ids.append(int(st["statId"]))
        lats.append(float(st["lat"]))
        lons.append(float(st["lon"]))
        for jname, name in JSON_VARIABLES.items():
            v = st.get(jname)
            if v is None:
                cols[name].append(np.full(len(tt), -9999.9))
            else:
                present.add(name)
                cols[name].append(np.asarray([-9999.9 if x is None else float(x) for x in v]))
    shared = all(np.array_equal(axes[0], a) for a in axes[1:])
    if shared:
        fields = {k: np.stack(v) for k, v in cols.items() if k in present}
        src = RawSource(axes[0], fields, is_observation)
    else:
        lengths = np.asarray([len(a) for a in axes], np.int32)
        width = int(max(1, lengths.max()))
        times = np.full((len(axes), width), np.iinfo(np.int64).min, np.int64)
        fields = {k: np.full((len(axes), width), -9999.9) for k in cols if k in present}
        for p, a in enumerate(axes):
            times[p, :len(a)] = a
            for k in fields:
                fields[k][p, :len(a)] = cols[k][p]
        src = RawSource(times, fields, is_observation, lengths)
    return src, ids, np.asarray(lats), np.asarray(lons)


def save_output(path: str, result: dict, ids, lats, lons, start_time: int, dtsecs: int) -> None:
    """Write the forecast like save_output/write_output (roadrunner.cpp:285-347): one object per
    simulated point with the kept times and RoadTemperature/Water/Ice/Snow/Deposit.  Numbers are
    written with Python's shortest round-trip repr (the reference asks jsoncpp for 7 digits)."""
    step = result["step"]
    n_out = result["tsurf"].shape[1]
    tstr = [_time.strftime("%Y-%m-%dT%H:%M", _time.gmtime(start_time + r * step * dtsecs))
            for r in range(n_out)]
    forecast = []
    for p in range(len(ids)):
        if result["status"][p] != 0:
            continue  # roadrunner.cpp:393: nothing is written for a rejected point
        forecast.append({
            "statId": int(ids[p]), "lat": float(lats[p]), "lon": float(lons[p]), "time": tstr,
            "RoadTemperature": result["tsurf"][p].tolist(), "Water": result["water"][p].tolist(),
            "Ice": result["ice"][p].tolist(), "Snow": result["snow"][p].tolist(),
            "Deposit": result["deposit"][p].tolist(),
        })
    with open(path, "w") as fh:
        json.dump(forecast, fh, indent=3)
