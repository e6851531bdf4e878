"""ctypes binding of ``roadsurf_amd/lib/libroadsurf_hip.so`` (``include/roadsurf.h``).

The library holds the HIP kernels, the C-ABI shim and the Fortran host
orchestration.  There is no CPU implementation behind this binding: if the
shared object is missing, ``load()`` raises; if no GPU is visible, every
compute entry point returns an error that the wrappers turn into
``RuntimeError``.
"""
from __future__ import annotations

import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ROADSURF_HIP_LIB") or os.path.join(_HERE, "lib", "libroadsurf_hip.so")

RS_MAX_LAYERS = abi.RS_MAX_LAYERS
RS_KNOT_FIELDS = 9
RS_NSTATE = RS_MAX_LAYERS + 17 + 21 + 2 * RS_MAX_LAYERS

_tbl = C.c_double * (RS_MAX_LAYERS + 2)


class RsConstants(C.Structure):
    _fields_ = (
        [("NLayers", C.c_int32), ("SimLen", C.c_int32), ("use_relaxation", C.c_int32),
         ("force_tsurf", C.c_int32), ("use_coupling", C.c_int32), ("cplLenI", C.c_int32),
         ("cplLenR", C.c_double), ("cplReduction", C.c_double),
         ("DTSecs", C.c_double), ("Tph", C.c_double), ("tsurfOutputDepth", C.c_double),
         ("twoDT", C.c_double),
         ("ZDpth", _tbl), ("DyC", _tbl), ("condDZ", _tbl), ("WCont", _tbl), ("dryCap", _tbl)]
        + [(n, C.c_double) for n in (
            "HSfac1", "logMom", "logHeat", "logCond", "logUstar",
            "VK_Const", "ZRefT", "Grav", "LVap", "LFus",
            "Emiss", "SB_Const", "Albedo0",
            "NightOn", "NightOff", "CalmLimDay", "CalmLimNgt", "TrfFricNgt", "TrFfricDay",
            "MaxPormms", "MissValI", "MinPrecmm", "MinWatmms", "MinSnowmms", "MinDepmms",
            "MinIcemms", "MaxSnowmms", "MaxDepmms", "MaxIcemms", "MaxWatmms", "AlbDry", "AlbSnow",
            "WatDens", "WatMHeat", "PorEvaF", "DampWearF", "TLimFreeze", "TLimMeltSnow",
            "TLimMeltIce", "TLimMeltDep", "TLimDew", "TLimColdH", "TLimColdL", "WetSnowFormR",
            "WetSnowMeltR", "PLimSnow", "PLimRain", "WWetLim", "WWearLim", "T4Melt0",
            "wSnowTran", "wSnow2Ice", "wIce", "wIce2", "wDep", "wWat",
        )]
    )


class RsForcing(C.Structure):
    _fields_ = (
        [(n, C.c_void_p) for n in ("tair", "tdew", "vz", "rhz", "prec", "sw", "lw",
                                   "tsurfobs", "depth", "precphase", "hour")]
        + [("t_stride", C.c_int64), ("hour_pstride", C.c_int32)]
        + [(n, C.c_void_p) for n in ("sw_dir", "lw_net", "sun")]
    )


class RsOutputs(C.Structure):
    _fields_ = (
        [(n, C.c_void_p) for n in ("tsurf", "snow", "water", "ice", "deposit", "ice2")]
        + [("t_stride", C.c_int64), ("decimate", C.c_int32), ("row0", C.c_int64)]
    )


class RsPointParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("tbottom", "initlen", "tair_relax", "vz_relax", "rh_relax",
                                          "coupling_index", "coupling_tsurf", "sky_view",
                                          "sin_lat", "cos_lat", "lon_rad", "horizons")] + \
               [("albedo_surroundings", C.c_double), ("horizon_index", C.c_void_p), ("horizons_by_point", C.c_int32)]


class RsHostExtras(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("sun", "sin_lat", "cos_lat", "lon_rad")] + \
               [("albedo_surroundings", C.c_double), ("first_failed", C.c_void_p), ("writeback", C.c_int32),
                ("diagnostics", C.c_void_p)]


RS_PREVIEW_MAX = 8


class RsPreview(C.Structure):
    _fields_ = [("n", C.c_int32), ("tair", C.c_void_p * RS_PREVIEW_MAX), ("vz", C.c_void_p * RS_PREVIEW_MAX),
                ("hour", C.c_int32 * RS_PREVIEW_MAX), ("tair_now", C.c_void_p), ("alpha", C.c_double),
                ("mode", C.c_int32), ("index", C.c_void_p), ("prec", C.c_void_p * RS_PREVIEW_MAX),
                ("tair_b", C.c_void_p * RS_PREVIEW_MAX), ("vz_b", C.c_void_p * RS_PREVIEW_MAX),
                ("w", C.c_double * RS_PREVIEW_MAX)]


class RsSynthSpec(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("point_offset", C.c_int64),
                ("steps_per_knot", C.c_int32), ("start_hour", C.c_int32), ("order", C.c_void_p)]


#: every symbol ``include/roadsurf.h`` declares
EXPORTS = (
    "rs_default_parameters", "rs_default_settings", "rs_default_local",
    "runsimulation", "runsimulation_batch", "runsimulation_batch_ex", "rs_coalesce_run", "rs_coalesce_stats", "rs_runsimulation_gathered", "rs_synth_fill_points", "rs_build_sha16", "rs_build_constants", "rs_bottom_temperature",
    "rs_sun_table", "rs_point_geometry",
    "rs_last_error", "rs_hip_device_count", "rs_hip_plan_create", "rs_hip_plan_destroy",
    "rs_hip_plan_npoints", "rs_hip_plan_npoints_padded", "rs_hip_plan_state_bytes",
    "rs_hip_init_state", "rs_hip_step", "rs_hip_step_cpl", "rs_hip_cpl_replay", "rs_hip_set_output_by_point", "rs_hip_state_download", "rs_hip_state_upload",
    "rs_hip_failed_count", "rs_hip_clock_probe", "rs_hip_first_failed_index", "rs_hip_set_diagnostics", "rs_hip_diagnostics", "rs_hip_sync", "rs_hip_synth_knots", "rs_hip_expand_forcing", "rs_hip_expand_forcing_ordered", "rs_hip_step_knots", "rs_hip_expand_forcing_on",
    "rs_hip_plan_order", "rs_hip_recluster", "rs_hip_recluster_forecast", "rs_hip_set_history_score", "rs_hip_coupling_windows_closed", "rs_hip_set_writeback", "rs_hip_plan_order_copy", "rs_hip_outputs_by_point", "rs_hip_plan_reset_order", "rs_hip_set_variant", "rs_hip_set_precision", "rs_hip_test_math", "rs_hip_division_mode", "rs_hip_div_mismatch_count", "rs_hip_div_special_count", "rs_hip_div_samples", "rs_hip_timing_reset", "rs_hip_timing_step_ms", "rs_hip_timing_intervals",
    "rs_host_run_batch", "rs_last_fanout", "rs_driver_run", "rs_driver_last_tiles", "rs_driver_last_raw_launches", "rs_hip_bl_stats", "rs_compat_begin", "rs_compat_step", "rs_compat_replay", "rs_compat_failed_index", "rs_compat_last_state", "rs_compat_outputs", "rs_compat_end", "rs_driver_expand", "rs_driver_release_cache", "rs_abi_version", "rs_abi_sizeof", "rs_fortran_sizeof",
)

_lib = None


def load() -> C.CDLL:
    """Load the HIP library or raise: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels bundle their own libamdhip64.so; two HIP runtimes in one
    # process do not share the device.  Import torch FIRST so that our library's
    # DT_NEEDED libamdhip64.so.7 resolves to the copy torch has already mapped.
    # (A pure C/C++/Fortran host, e.g. the reference's roadrunner, simply gets
    # /opt/rocm/lib/libamdhip64.so.7 via the rpath.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing - build it with `make -C roadsurf_amd` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "roadsurf_amd has no CPU implementation."
        )
    L = C.CDLL(LIB_PATH)
    P = C.POINTER
    L.rs_last_error.restype = C.c_char_p
    L.rs_abi_version.restype = C.c_int
    L.rs_hip_device_count.restype = C.c_int
    L.rs_default_parameters.argtypes = [P(abi.InputParameters), C.c_double]
    L.rs_default_settings.argtypes = [P(abi.InputSettings), C.c_int32]
    L.rs_default_local.argtypes = [P(abi.LocalParameters)]
    L.rs_build_constants.argtypes = [P(abi.InputSettings), P(abi.InputParameters), P(RsConstants),
                                     P(C.c_int32)]
    L.rs_build_constants.restype = None
    L.rs_bottom_temperature.argtypes = [P(abi.InputParameters), P(RsConstants), C.c_int32,
                                        C.c_int32, C.c_int32]
    L.rs_bottom_temperature.restype = C.c_double
    L.runsimulation.argtypes = [P(abi.OutputPointers), P(abi.InputPointers), P(abi.InputSettings),
                                P(abi.InputParameters), P(abi.LocalParameters)]
    L.runsimulation.restype = None
    L.runsimulation_batch.argtypes = [C.c_int32, P(abi.OutputPointers), P(abi.InputPointers),
                                      P(abi.InputSettings), P(abi.InputParameters),
                                      P(abi.LocalParameters), P(C.c_int32)]
    L.runsimulation_batch.restype = None
    L.runsimulation_batch_ex.argtypes = [C.c_int32, P(abi.OutputPointers), P(abi.InputPointers),
                                         P(abi.InputSettings), P(abi.InputParameters),
                                         P(abi.LocalParameters), P(C.c_int32), C.c_void_p]
    L.runsimulation_batch_ex.restype = None
    L.rs_hip_plan_create.argtypes = [C.c_int32, C.c_int64, P(RsConstants), C.c_void_p]
    L.rs_hip_plan_create.restype = C.c_void_p
    L.rs_hip_plan_destroy.argtypes = [C.c_void_p]
    L.rs_hip_plan_destroy.restype = None
    for n in ("rs_hip_plan_npoints", "rs_hip_plan_npoints_padded", "rs_hip_failed_count"):
        getattr(L, n).argtypes = [C.c_void_p]
        getattr(L, n).restype = C.c_int64
    L.rs_hip_clock_probe.argtypes = [C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p]
    L.rs_hip_plan_state_bytes.argtypes = [C.c_void_p]
    L.rs_hip_plan_state_bytes.restype = C.c_size_t
    L.rs_hip_init_state.argtypes = [C.c_void_p, P(RsForcing), P(RsPointParams)]
    L.rs_hip_step.argtypes = [C.c_void_p, P(RsForcing), P(RsOutputs), P(RsPointParams),
                              C.c_int32, C.c_int32]
    L.rs_hip_step_cpl.argtypes = [C.c_void_p, P(RsForcing), P(RsOutputs), P(RsPointParams),
                                  C.c_int32, C.c_int32]
    L.rs_hip_set_output_by_point.argtypes = [C.c_void_p, C.c_int32]
    L.rs_hip_cpl_replay.argtypes = [C.c_void_p, P(RsForcing), P(RsOutputs), P(RsPointParams),
                                    C.c_int32, C.c_int32, P(C.c_int32)]
    L.rs_hip_state_download.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.rs_hip_state_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.rs_hip_sync.argtypes = [C.c_void_p]
    L.rs_hip_first_failed_index.argtypes = [C.c_void_p, C.c_void_p]
    L.rs_hip_set_diagnostics.argtypes = [C.c_void_p, C.c_int32]
    L.rs_hip_diagnostics.argtypes = [C.c_void_p, C.c_void_p]
    L.rs_hip_synth_knots.argtypes = [C.c_void_p, P(RsSynthSpec), C.c_void_p, C.c_int32, C.c_int32]
    L.rs_hip_expand_forcing.argtypes = [C.c_void_p, P(RsSynthSpec), C.c_void_p, C.c_int32,
                                        C.c_int32, P(RsForcing), C.c_int32, C.c_int32]
    L.rs_hip_expand_forcing_ordered.argtypes = L.rs_hip_expand_forcing.argtypes
    L.rs_hip_expand_forcing_on.argtypes = [C.c_void_p, P(RsSynthSpec), C.c_void_p, C.c_int32,
                                           C.c_int32, P(RsForcing), C.c_int32, C.c_int32, C.c_void_p]
    L.rs_hip_step_knots.argtypes = [C.c_void_p, P(RsSynthSpec), C.c_void_p, C.c_int32, C.c_int32, P(RsOutputs),
                                    P(RsPointParams), C.c_int32, C.c_int32]
    L.rs_hip_plan_order.argtypes = [C.c_void_p]
    L.rs_hip_plan_order.restype = C.c_void_p
    L.rs_hip_recluster.argtypes = [C.c_void_p]
    L.rs_hip_recluster_forecast.argtypes = [C.c_void_p, P(RsPreview)]
    L.rs_hip_set_history_score.argtypes = [C.c_void_p, C.c_int32]
    L.rs_hip_coupling_windows_closed.argtypes = [C.c_void_p, C.c_int32]
    L.rs_hip_set_writeback.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.rs_hip_plan_order_copy.argtypes = [C.c_void_p, C.c_void_p]
    L.rs_hip_outputs_by_point.argtypes = [C.c_void_p, P(RsOutputs), C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    L.rs_hip_plan_reset_order.argtypes = [C.c_void_p]
    L.rs_hip_set_variant.argtypes = [C.c_void_p, C.c_int32]
    L.rs_hip_set_precision.argtypes = [C.c_void_p, C.c_int32]
    L.rs_hip_div_mismatch_count.argtypes = [C.c_void_p]
    L.rs_hip_div_mismatch_count.restype = C.c_int64
    L.rs_hip_div_special_count.argtypes = [C.c_void_p]
    L.rs_hip_div_special_count.restype = C.c_int64
    L.rs_hip_div_samples.argtypes = [C.c_void_p, C.c_void_p]
    L.rs_hip_test_math.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]
    L.rs_hip_timing_reset.argtypes = [C.c_void_p]
    L.rs_hip_timing_step_ms.argtypes = [C.c_void_p, P(C.c_int32)]
    L.rs_hip_timing_step_ms.restype = C.c_double
    L.rs_hip_timing_intervals.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    L.rs_hip_timing_intervals.restype = C.c_int32
    L.rs_host_run_batch.argtypes = [C.c_int32, P(abi.OutputPointers), P(abi.InputPointers),
                                    P(RsConstants), P(abi.LocalParameters), P(C.c_double),
                                    P(RsHostExtras), C.c_int32]
    L.rs_sun_table.argtypes = [C.c_int32] + [C.c_void_p] * 7
    L.rs_sun_table.restype = None
    L.rs_point_geometry.argtypes = [C.c_int32, P(abi.LocalParameters)] + [C.c_void_p] * 3
    L.rs_point_geometry.restype = None
    for n in ("rs_abi_sizeof", "rs_fortran_sizeof"):
        getattr(L, n).argtypes = [C.c_int]
        getattr(L, n).restype = C.c_int64
    _lib = L
    return L


def last_error() -> str:
    return (load().rs_last_error() or b"").decode()


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {last_error()}")


def build_constants(settings: abi.InputSettings, params: abi.InputParameters) -> RsConstants:
    """Host-side (Fortran) constants table; needs no GPU."""
    c = RsConstants()
    st = C.c_int32(0)
    load().rs_build_constants(C.byref(settings), C.byref(params), C.byref(c), C.byref(st))
    if st.value != 0:
        raise ValueError("rs_build_constants rejected the settings (NLayers in 5..32, SimLen>=1, DTSecs>0)")
    return c


def bottom_temperature(params: abi.InputParameters, consts: RsConstants, year: int, month: int,
                       day: int) -> float:
    return float(load().rs_bottom_temperature(C.byref(params), C.byref(consts), year, month, day))
