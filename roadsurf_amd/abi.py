"""ctypes mirror of ``include/roadsurf.h`` — the C-ABI of the RoadSurf hot path.

Struct layouts are the reference's ``Bind(C)`` types
(``src/InputPointers.f90.inc:4-27``, ``src/OutputPointers.f90.inc:4-17``,
``src/InputSettings.f90.inc:4-18``, ``src/InputParameters.f90.inc:4-91``,
``src/LocalParameters.f90.inc:4-15`` in fmidev/RoadSurf) and are checked
field-by-field against SURVEY.md Appendix D in ``tests/test_abi_layout.py``.
"""
from __future__ import annotations

import ctypes as C

RS_MAX_LAYERS = 32

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class InputPointers(C.Structure):
    _fields_ = [("inputLen", C.c_int32)] + [
        (n, c_double_p)
        for n in (
            "c_tair", "c_tdew", "c_VZ", "c_Rhz", "c_prec", "c_SW", "c_LW",
            "c_SW_dir", "c_LW_net", "c_TSurfObs",
        )
    ] + [
        ("c_PrecPhase", c_int32_p),
        ("c_local_horizons", c_double_p),
        ("c_Depth", c_double_p),
    ] + [(n, c_int32_p) for n in ("c_year", "c_month", "c_day", "c_hour", "c_minute", "c_second")]


class OutputPointers(C.Structure):
    _fields_ = [("outputLen", C.c_int32)] + [
        (n, c_double_p)
        for n in ("c_TsurfOut", "c_SnowOut", "c_WaterOut", "c_IceOut", "c_DepositOut", "c_Ice2Out")
    ]


class InputSettings(C.Structure):
    _fields_ = [
        ("SimLen", C.c_int32),
        ("use_coupling", C.c_int32),
        ("use_relaxation", C.c_int32),
        ("force_tsurf", C.c_int32),
        ("DTSecs", C.c_double),
        ("tsurfOutputDepth", C.c_double),
        ("NLayers", C.c_int32),
        ("coupling_minutes", C.c_int32),
        ("couplingEffectReduction", C.c_double),
        ("outputStep", C.c_int32),
    ]


INPUT_PARAMETER_NAMES = (
    "NightOn", "NightOff", "CalmLimDay", "CalmLimNgt", "TrfFricNgt", "TrFfricDay",
    "Grav", "SB_Const", "VK_Const", "LVap", "LFus", "WatDens", "SnowDens", "IceDens",
    "DepDens", "WatMHeat", "PorEvaF",
    "ZRefW", "ZRefT", "ZeroDisp", "ZMom", "ZHeat", "Emiss", "Albedo",
    "Albedo_surroundings", "MaxPormms", "TClimG", "DampDpth", "Omega", "AZ", "DampWearF",
    "AlbDry", "AlbSnow", "vsh1", "vsh2", "Poro1", "Poro2", "RhoB1", "RhoB2", "Silt1", "Silt2",
    "freezing_limit_normal", "snow_melting_limit_normal", "ice_melting_limit_normal",
    "frost_melting_limit_normal", "frost_formation_limit_normal", "T4Melt_normal",
    "TLimColdH", "TLimColdL", "WetSnowFormR", "WetSnowMeltR",
    "PLimSnow", "PLimRain", "MaxSnowmms", "MaxDepmms", "MaxIcemms", "MaxExtmms",
    "MissValI", "MissValR",
    "Snow2IceFac",
    "MinPrecmm", "MinWatmms", "MinSnowmms",
    "MaxWatmms",
    "WDampLim", "WWetLim",
    "WWearLim",
    "MinDepmms", "MinIcemms",
)


class InputParameters(C.Structure):
    _fields_ = [(n, C.c_double) for n in INPUT_PARAMETER_NAMES]


class LocalParameters(C.Structure):
    _fields_ = [
        ("tair_relax", C.c_double),
        ("VZ_relax", C.c_double),
        ("RH_relax", C.c_double),
        ("couplingIndexI", C.c_int32),
        ("couplingTsurf", C.c_double),
        ("lat", C.c_double),
        ("lon", C.c_double),
        ("sky_view", C.c_double),
        ("InitLenI", C.c_int32),
    ]


def default_parameters(dtsecs: float = 30.0) -> InputParameters:
    """Reference defaults, ``examples/example1/src/InputParameters.h:18-94`` and the
    DTSecs-derived limits of ``InputParameters.cpp:13-21`` (same expressions, in
    double, as the C++ driver evaluates them)."""
    import math

    p = InputParameters()
    p.NightOn, p.NightOff = 19.0, 4.0
    p.CalmLimDay, p.CalmLimNgt = 1.5, 0.4
    p.TrfFricNgt, p.TrFfricDay = 5.0, 10.0
    p.Grav, p.SB_Const, p.VK_Const = 9.81, 5.67e-8, 0.4
    p.LVap, p.LFus = 2.452e6, 0.334e6
    p.WatDens, p.SnowDens, p.IceDens, p.DepDens = 999.87, 100.0, 920.0, 920.0
    p.WatMHeat, p.PorEvaF = 333000.0, 1.0
    p.ZRefW, p.ZRefT, p.ZeroDisp, p.ZMom, p.ZHeat = 10.0, 2.0, 0.0, 0.4, 0.001
    p.Emiss, p.Albedo, p.Albedo_surroundings = 0.95, 0.10, 0.15
    p.MaxPormms, p.TClimG, p.DampDpth = 1.0, 6.4, 2.7
    p.Omega, p.AZ, p.DampWearF = 2.0 * math.pi / 365.0, 0.6, 0.5
    p.AlbDry, p.AlbSnow = 0.1, 0.6
    p.vsh1, p.vsh2 = 1.94e06, 1.28e06
    p.Poro1, p.Poro2, p.RhoB1, p.RhoB2, p.Silt1, p.Silt2 = 0.1, 0.4, 2.11, 1.6, 0.1, 0.8
    p.freezing_limit_normal = -0.25
    p.snow_melting_limit_normal = 0.25
    p.ice_melting_limit_normal = 0.25
    p.frost_melting_limit_normal = 1.25
    p.frost_formation_limit_normal = 0.25
    p.T4Melt_normal = 0.25
    p.TLimColdH, p.TLimColdL = -19.0, -21.0
    p.WetSnowFormR, p.WetSnowMeltR = 0.1, 0.6
    p.PLimSnow, p.PLimRain = 0.3, 0.7
    p.MaxSnowmms, p.MaxDepmms, p.MaxIcemms, p.MaxExtmms = 100.0, 2.0, 50.0, 1.0
    p.MissValI, p.MissValR = -9999.0, -99.99
    p.Snow2IceFac = 0.5
    p.MinPrecmm = 0.05 * dtsecs / 3600.0
    p.MinWatmms = 0.01 * dtsecs / 3600.0
    p.MinSnowmms = 0.1 * dtsecs / 3600.0
    p.MaxWatmms = p.MaxPormms + p.MaxExtmms
    p.WDampLim = 0.1 * p.MaxPormms
    p.WWetLim = 0.9 * p.MaxPormms
    p.WWearLim = 0.1 * p.MaxPormms
    p.MinDepmms = 0.01 * dtsecs / 3600.0
    p.MinIcemms = 0.05 * dtsecs / 3600.0
    return p


def default_settings(simlen: int, dtsecs: float = 30.0) -> InputSettings:
    """``examples/example1/src/InputSettings.h:13-23`` with ``force_tsurf = 0``."""
    s = InputSettings()
    s.SimLen = simlen
    s.use_coupling = 0
    s.use_relaxation = 0
    s.force_tsurf = 0
    s.DTSecs = dtsecs
    s.tsurfOutputDepth = -9999.9
    s.NLayers = 15
    s.coupling_minutes = 180
    s.couplingEffectReduction = 4.0 * 3600
    s.outputStep = 60
    return s


def default_local() -> LocalParameters:
    """``examples/example1/src/LocalParameters.h:17-25``."""
    l = LocalParameters()
    l.tair_relax = l.VZ_relax = l.RH_relax = -9999.0
    l.couplingIndexI = -9999
    l.couplingTsurf = -9999.0
    l.lat = l.lon = -9999.0
    l.sky_view = 1.0
    l.InitLenI = 0
    return l
