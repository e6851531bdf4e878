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
    declared = set(re.findall(r"\b((?:rs_[a-z0-9_]+|runsimulation(?:_batch(?:_ex)?)?))\s*\(", header))
    declared -= {"rs_last_error()"}
    assert declared == set(lib.EXPORTS), declared ^ set(lib.EXPORTS)
    for sym in declared:
        assert hasattr(hip_lib, sym), sym
    assert hip_lib.rs_abi_version() == 11


def test_c_fortran_ctypes_sizes_agree(hip_lib):
    py = [abi.InputPointers, abi.OutputPointers, abi.InputSettings, abi.InputParameters,
          abi.LocalParameters, lib.RsConstants]
    for i, cls in enumerate(py):
        assert hip_lib.rs_abi_sizeof(i) == hip_lib.rs_fortran_sizeof(i) == C.sizeof(cls), cls.__name__


def test_device_api_structs_have_the_size_the_binding_assumes(hip_lib):
    """The ctypes mirrors of the device-resident API's structs (roadsurf_amd/lib.py) against sizeof in the
    library: a field added on one side only (RsPointParams grew twice in round 4) shows here, without a GPU."""
    from roadsurf_amd import driver
    py = [lib.RsForcing, lib.RsOutputs, lib.RsPointParams, lib.RsHostExtras, lib.RsPreview, lib.RsSynthSpec,
          driver.RsRawSource, driver.RsDriverInput, driver.RsDriverOutput]
    for i, cls in enumerate(py, start=6):
        assert hip_lib.rs_abi_sizeof(i) == C.sizeof(cls), cls.__name__


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


REF_SRC = "/root/reference/examples/example1/src"


def _run(cmd, cwd):
    import subprocess
    r = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True)
    assert r.returncode == 0, " ".join(cmd) + "\n" + r.stdout[-1500:] + r.stderr[-3000:]
    return r


@pytest.mark.skipif(not os.path.exists(REF_SRC), reason="build container only: needs /root/reference")
def test_layout_against_the_reference_headers_compiled(tmp_path):
    """The reference's own InputPointers.h / OutputPointers.h (they include nothing) compiled beside
    include/roadsurf.h (ours in a namespace): sizeof and offsetof of every member must agree, checked
    by the compiler.  (InputSettings.h / InputParameters.h / LocalParameters.h need jsoncpp, which
    the image lacks; their layouts are pinned by the Fortran side: rs_fortran_sizeof and the
    reference's .f90.inc files compiled into oracle/_ref.)"""
    src = tmp_path / "layout.cpp"
    members_in = ["inputLen", "c_tair", "c_tdew", "c_VZ", "c_Rhz", "c_prec", "c_SW", "c_LW", "c_SW_dir",
                  "c_LW_net", "c_TSurfObs", "c_PrecPhase", "c_local_horizons", "c_Depth", "c_year",
                  "c_month", "c_day", "c_hour", "c_minute", "c_second"]
    members_out = ["outputLen", "c_TsurfOut", "c_SnowOut", "c_WaterOut", "c_IceOut", "c_DepositOut",
                   "c_Ice2Out"]
    lines = ["#include <cstddef>", "#include <cstdint>", "#include <stddef.h>", "#include <stdint.h>",
             f'#include "{REF_SRC}/InputPointers.h"', f'#include "{REF_SRC}/OutputPointers.h"',
             "namespace ours {", f'#include "{ROOT}/include/roadsurf.h"', "}",
             "#define SAME(T, m) static_assert(offsetof(::T, m) == offsetof(ours::T, m) && "
             "sizeof(((::T *)0)->m) == sizeof(((ours::T *)0)->m), #T \".\" #m)",
             "static_assert(sizeof(::InputPointers) == sizeof(ours::InputPointers), \"InputPointers\");",
             "static_assert(sizeof(::OutputPointers) == sizeof(ours::OutputPointers), \"OutputPointers\");"]
    lines += [f"SAME(InputPointers, {m});" for m in members_in]
    lines += [f"SAME(OutputPointers, {m});" for m in members_out]
    lines += ["int main() { return 0; }"]
    src.write_text("\n".join(lines) + "\n")
    _run(["g++", "-std=c++17", "-Wall", "-fsyntax-only", str(src)], tmp_path)
    # and the check does bite: a shifted member must not compile
    bad = tmp_path / "bad.cpp"
    bad.write_text(src.read_text().replace("SAME(InputPointers, c_tdew);",
                                           "static_assert(offsetof(::InputPointers, c_tdew) == "
                                           "offsetof(ours::InputPointers, c_VZ), \"shifted\");"))
    import subprocess
    assert subprocess.run(["g++", "-std=c++17", "-fsyntax-only", str(bad)], capture_output=True).returncode != 0


@pytest.mark.skipif(not os.path.exists(REF_SRC), reason="build container only: needs /root/reference")
def test_reference_declaration_of_runsimulation_links_against_the_library(tmp_path, hip_lib):
    """A translation unit that declares `runsimulation` exactly as the reference's driver does
    (examples/example1/src/roadrunner.cpp:22-29), over the reference's own pointer structs, links
    against libroadsurf_hip.so: same symbol, C linkage, five by-reference arguments.  (Only the
    address is taken: no GPU here.)"""
    src = tmp_path / "link.cpp"
    src.write_text(f'''
#include "{REF_SRC}/InputPointers.h"
#include "{REF_SRC}/OutputPointers.h"
struct InputSettings; struct InputParameters; struct LocalParameters;
// The fortran API
extern "C"
{{
  void runsimulation(OutputPointers* pOutputPointers,
                     const InputPointers* pInputPointers,
                     const InputSettings* pSettings,
                     const InputParameters* pInputParams,
                     const LocalParameters* lParameters);
}}
#include <cstdio>
int main() {{
  void (*f)(OutputPointers*, const InputPointers*, const InputSettings*, const InputParameters*,
            const LocalParameters*) = &runsimulation;
  std::printf("%d\\n", f != nullptr);
  return 0;
}}
''')
    libdir = os.path.join(ROOT, "roadsurf_amd", "lib")
    exe = tmp_path / "link"
    _run(["g++", "-std=c++17", str(src), "-o", str(exe), f"-L{libdir}", "-lroadsurf_hip",
          f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined"], tmp_path)
    r = _run([str(exe)], tmp_path)
    assert r.stdout.strip() == "1"


def test_reference_module_names_resolve_for_fortran_callers(tmp_path, hip_lib):
    """Caller code written against the reference says `use RoadSurfVariables` (and `use RoadSurf`):
    it must compile against this library's module files and see the five Bind(C) types with the
    reference's component names, plus the many-point entry; `runsimulation` itself comes from
    `module RoadSurfHipEntry` (module RoadSurf cannot export it: the reference's own Simulation.f90
    DEFINES that name while using the module)."""
    import shutil
    if shutil.which("amdflang") is None:
        pytest.skip("no Fortran compiler")
    moddir = os.path.join(ROOT, "roadsurf_amd", "build")
    src = tmp_path / "caller.f90"
    src.write_text('''
subroutine caller(outp, inp, n)
   use, intrinsic :: iso_c_binding
   use RoadSurfVariables
   use RoadSurf
   use RoadSurfHipEntry, only: runsimulation
   implicit none
   integer(c_int), value :: n
   type(OutputPointers), intent(inout) :: outp(n)
   type(InputPointers), intent(in) :: inp(n)
   type(InputSettings) :: s
   type(InputParameters) :: p
   type(LocalParameters) :: l(n)
   integer(c_int) :: status
   s%SimLen = inp(1)%inputLen; s%use_coupling = 0; s%use_relaxation = 0; s%force_tsurf = 0
   s%DTSecs = 30.0d0; s%tsurfOutputDepth = -9999.9d0; s%NLayers = 15; s%coupling_minutes = 180
   s%couplingEffectReduction = 14400.0d0; s%outputStep = 60
   p%NightOn = 19.0d0; p%MinIcemms = 0.0d0
   l(:)%sky_view = 1.0d0; l(:)%InitLenI = 1; l(:)%couplingIndexI = -9999
   call runsimulation(outp(1), inp(1), s, p, l(1))
   call runsimulation_batch(n, outp, inp, s, p, l, status)
end subroutine caller
''')
    _run(["amdflang", "-c", f"-I{moddir}", str(src), "-o", str(tmp_path / "caller.o")], tmp_path)


@pytest.mark.skipif(not os.path.exists(REF_SRC), reason="build container only: needs /root/reference")
def test_reference_time_loop_compiles_and_links_unchanged_against_the_module_surface(tmp_path, hip_lib):
    """Row b' of the coverage table, decided by a compiler: the reference's own examples/example1/src/
    Simulation.f90 - runsimulation's time loop over the fourteen public procedures of `module RoadSurf` and the
    eleven state types of `module RoadSurfVariables` (src/RoadSurf.f90:6-270, src/RoadSurfVariables.f90:13-28),
    plus the external lastValues - compiles UNCHANGED against this library's module files and links against
    libroadsurf_hip.so with no undefined symbol.  (tests/test_hip_module_surface.py runs it on a GPU.)"""
    import shutil
    if shutil.which("amdflang") is None:
        pytest.skip("no Fortran compiler")
    ref = "/root/reference"
    sim = os.path.join(REF_SRC, "Simulation.f90")
    moddir = os.path.join(ROOT, "roadsurf_amd", "build")
    libdir = os.path.join(ROOT, "roadsurf_amd", "lib")
    # flang's preprocessor rejects `#pragma once` (src/Constants.h:1): the same header without that line,
    # and the source through a symlink so that its `#include "Constants.h"` finds it (oracle/build_ref.sh)
    hdr = open(os.path.join(ref, "src", "Constants.h")).read().splitlines()
    (tmp_path / "Constants.h").write_text("\n".join(l for l in hdr if "#pragma once" not in l) + "\n")
    os.symlink(sim, tmp_path / "Simulation.f90")
    _run(["amdflang", "-cpp", "-O2", "-fPIC", "-w", f"-I{tmp_path}", f"-I{moddir}", "-module-dir", str(tmp_path),
          "-c", str(tmp_path / "Simulation.f90"), "-o", str(tmp_path / "Simulation.o")], tmp_path)
    so = tmp_path / "libsim.so"
    _run(["amdflang", "-shared", "-o", str(so), str(tmp_path / "Simulation.o"), f"-L{libdir}", "-lroadsurf_hip",
          f"-Wl,-rpath,{libdir}", "-Wl,--no-undefined"], tmp_path)
    syms = _run(["nm", "-D", "--defined-only", str(so)], tmp_path).stdout
    assert " T runsimulation" in syms and "roadmodelonestep_" in syms
    und = _run(["nm", "-D", "--undefined-only", str(so)], tmp_path).stdout
    for name in ("initialization", "checkvalues", "couplingoperations1", "relaxationoperations", "setcurrentvalues",
                 "balancemodelonestep", "saveoutput", "checkendcoupling", "precipitationtostorage",
                 "modradiationbysurroundings", "wearfactors", "roadcond", "calcalbedo", "connectfortran2carrays"):
        assert f"_QMroadsurfP{name}" in und, name     # the time loop really calls the module's procedures
    assert "lastvalues_" in und
