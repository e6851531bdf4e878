"""The C-ABI: struct layouts, the shared library loads and
exports every symbol include/roadsurf.h declares, C / Fortran / ctypes agree on
sizes, and the product refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from roadsurf_amd import abi, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _offsets(cls):
    return {n: getattr(cls, n).offset for n, _ in cls._fields_}


def test_reference_struct_offsets():
    o = _offsets(abi.InputPointers)
    want = dict(inputLen=0, c_tair=8, c_tdew=16, c_VZ=24, c_Rhz=32, c_prec=40, c_SW=48, c_LW=56,
                c_SW_dir=64, c_LW_net=72, c_TSurfObs=80, c_PrecPhase=88, c_local_horizons=96,
                c_Depth=104, c_year=112, c_month=120, c_day=128, c_hour=136, c_minute=144, c_second=152)
    assert o == want and C.sizeof(abi.InputPointers) == 160
    o = _offsets(abi.OutputPointers)
    assert o == dict(outputLen=0, c_TsurfOut=8, c_SnowOut=16, c_WaterOut=24, c_IceOut=32,
                     c_DepositOut=40, c_Ice2Out=48) and C.sizeof(abi.OutputPointers) == 56
    o = _offsets(abi.InputSettings)
    assert o == dict(SimLen=0, use_coupling=4, use_relaxation=8, force_tsurf=12, DTSecs=16,
                     tsurfOutputDepth=24, NLayers=32, coupling_minutes=36,
                     couplingEffectReduction=40, outputStep=48)
    o = _offsets(abi.LocalParameters)
    assert o == dict(tair_relax=0, VZ_relax=8, RH_relax=16, couplingIndexI=24, couplingTsurf=32,
                     lat=40, lon=48, sky_view=56, InitLenI=64) and C.sizeof(abi.LocalParameters) == 72
    assert C.sizeof(abi.InputParameters) == 69 * 8
    assert len(abi.INPUT_PARAMETER_NAMES) == 69 and abi.INPUT_PARAMETER_NAMES[0] == "NightOn" \
        and abi.INPUT_PARAMETER_NAMES[-1] == "MinIcemms"


def test_library_exports_every_declared_symbol(hip_lib):
    header = open(os.path.join(ROOT, "include", "roadsurf.h")).read()
    declared = set(re.findall(r"\b((?:rs_[a-z0-9_]+|runsimulation(?:_batch)?))\s*\(", header))
    declared -= {"rs_last_error()"}
    assert declared == set(lib.EXPORTS), declared ^ set(lib.EXPORTS)
    for sym in declared:
        assert hasattr(hip_lib, sym), sym
    assert hip_lib.rs_abi_version() == 1


def test_c_fortran_ctypes_sizes_agree(hip_lib):
    py = [abi.InputPointers, abi.OutputPointers, abi.InputSettings, abi.InputParameters,
          abi.LocalParameters, lib.RsConstants]
    for i, cls in enumerate(py):
        assert hip_lib.rs_abi_sizeof(i) == hip_lib.rs_fortran_sizeof(i) == C.sizeof(cls), cls.__name__


def test_defaults_match_reference_headers(hip_lib):
    p = abi.InputParameters()
    hip_lib.rs_default_parameters(C.byref(p), 30.0)
    q = abi.default_parameters(30.0)
    for n in abi.INPUT_PARAMETER_NAMES:
        assert getattr(p, n) == getattr(q, n), n
    assert p.MinPrecmm == 0.05 * 30.0 / 3600.0 and p.MaxWatmms == 2.0 and p.WWetLim == 0.9
    s = abi.InputSettings()
    hip_lib.rs_default_settings(C.byref(s), 5761)
    assert (s.SimLen, s.NLayers, s.DTSecs, s.outputStep, s.force_tsurf) == (5761, 15, 30.0, 60, 0)
    l = abi.LocalParameters()
    hip_lib.rs_default_local(C.byref(l))
    assert l.sky_view == 1.0 and l.InitLenI == 0 and l.couplingIndexI == -9999


def test_no_cpu_fallback_without_gpu(hip_lib):
    """On a box without a HIP device the product path must fail loudly."""
    if hip_lib.rs_hip_device_count() > 0:
        pytest.skip("a GPU is visible here")
    s = abi.default_settings(100)
    c = lib.build_constants(s, abi.default_parameters())
    h = hip_lib.rs_hip_plan_create(0, 16, C.byref(c), None)
    assert not h and "no HIP device" in lib.last_error()
    from roadsurf_amd import device
    with pytest.raises(RuntimeError):
        device.Plan(16, s, abi.default_parameters())


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "roadsurf_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".f90", ".cpp")) or fn == "Makefile":
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "oracle/" not in txt.replace("the oracle", "") or fn in ("rs_synth.h",), \
                    os.path.join(dp, fn)
                assert "liboracle" not in txt and "libroadsurf_ref" not in txt, os.path.join(dp, fn)
