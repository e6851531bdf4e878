#!/usr/bin/env bash
# oracle/build_ref.sh — TEST INFRASTRUCTURE.
#
# Builds the reference's own Fortran (RoadSurf library + example1's
# Simulation.f90, which owns the time loop) into oracle/_ref/libroadsurf_ref.so,
# straight from the sources where they lie under /root/reference.  Nothing from
# the reference is copied into this repository: the build happens in a mktemp
# directory of SYMLINKS to the reference files, the objects are linked into
# oracle/_ref/ (git-ignored), and the temp directory is deleted.
#
# Compiler: amdflang (AMD flang 22, ROCm 7.2) -O2, no fast-math.  Upstream's
# documented build is gfortran -O2 -Ofast + unsafe-math (Makefile:26,35), which
# is neither available here nor bit-reproducible; parity in this project is
# defined against THIS build.
#
# One generated file: flang's preprocessor rejects `#pragma once`
# (src/Constants.h:1), so the temp dir gets a Constants.h that is the
# reference's with that one line filtered out (9 #defines, semantics unchanged).
set -euo pipefail

REF="${ROADSURF_REFERENCE:-/root/reference}"
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
FC="${FC:-amdflang}"
FFLAGS="${FFLAGS:--cpp -O2 -fPIC -w}"

if [ ! -d "$REF/src" ]; then
  echo "build_ref: $REF not present (GPU box?) - using prebuilt $OUT if any" >&2
  exit 0
fi

mkdir -p "$OUT"
TMP="$(mktemp -d /tmp/roadsurf_ref.XXXXXX)"
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$TMP/src" "$TMP/obj"

for f in "$REF"/src/*; do
  b="$(basename "$f")"
  [ "$b" = "Constants.h" ] && continue
  ln -s "$f" "$TMP/src/$b"
done
ln -s "$REF/examples/example1/src/Simulation.f90" "$TMP/src/Simulation.f90"
grep -v '#pragma once' "$REF/src/Constants.h" > "$TMP/src/Constants.h"

cd "$TMP/src"
# module first, then interface module, then submodules / external procedures
# (same order constraint as the reference Makefile:75-86)
for m in RoadSurfVariables RoadSurf BalanceModel BoundaryLayer Cond \
         ConnectFortran2Carrays Coupling Initialization InputOutput \
         ModRadiation Relaxation Storage SunPosition Simulation; do
  $FC $FFLAGS -module-dir "$TMP/obj" -I. -c "$m.f90" -o "$TMP/obj/$m.o"
done

# optional per-subroutine probe (our own Fortran, uses the reference's modules)
if [ -f "$HERE/ref_probe.f90" ]; then
  $FC $FFLAGS -module-dir "$TMP/obj" -I"$TMP/obj" -I. -c "$HERE/ref_probe.f90" -o "$TMP/obj/ref_probe.o"
fi

gcc -O2 -fPIC -fopenmp -c "$HERE/harness.c" -o "$TMP/obj/harness.o"
$FC -shared -o "$OUT/libroadsurf_ref.so" "$TMP"/obj/*.o -fopenmp -lgomp 2>/dev/null || \
$FC -shared -o "$OUT/libroadsurf_ref.so" "$TMP"/obj/*.o -L/usr/lib/gcc/x86_64-linux-gnu/11 -lgomp
echo "built $OUT/libroadsurf_ref.so"

# ---- second variant: coupling as the reference INTENDS it --------------------------
# The reference relies on undefined behaviour for coupling: setInputParam stores the
# observation (coupling%obsI(1), obsTsurf(1), NObs; src/InputOutput.f90:30-33) and THEN
# `allocator` is called with `type(couplingVariables), intent(OUT) :: coupling`
# (src/Initialization.f90:96,150-157).  An INTENT(OUT) dummy becomes undefined on entry;
# gfortran (the documented compiler) happens to leave the non-allocatable components alone,
# flang re-initialises the whole object, so in the strict build above obsI(1) = 0,
# obsTsurf(1) = 0 and the coupling window is [1,0]: coupling never runs (and, because
# use_coupling stays .true., observation forcing is disabled too, src/InputOutput.f90:120-121).
# That is faithfully what libroadsurf_ref.so does.  To pin the coupling ALGORITHM we also
# build libroadsurf_ref_cpl.so from the same sources with ONE token changed in the temp
# copy of Initialization.f90: that dummy's INTENT(OUT) -> INTENT(INOUT), i.e. gfortran's
# observable behaviour.  Nothing else differs; non-coupled runs are bit-identical between
# the two libraries (tests/test_oracle_vs_golden.py).
rm "$TMP/src/Initialization.f90"
sed '/^Subroutine allocator/,/^end Subroutine/ s/type(couplingVariables), intent(OUT) :: coupling/type(couplingVariables), intent(INOUT) :: coupling/' \
    "$REF/src/Initialization.f90" > "$TMP/src/Initialization.f90"
if cmp -s "$REF/src/Initialization.f90" "$TMP/src/Initialization.f90"; then
  echo "build_ref: allocator intent line not found - reference changed?" >&2; exit 1
fi
[ "$(diff "$REF/src/Initialization.f90" "$TMP/src/Initialization.f90" | grep -c '^>')" = "1" ]
cd "$TMP/src"
$FC $FFLAGS -module-dir "$TMP/obj" -I. -c Initialization.f90 -o "$TMP/obj/Initialization.o"
$FC -shared -o "$OUT/libroadsurf_ref_cpl.so" "$TMP"/obj/*.o -fopenmp -lgomp 2>/dev/null || \
$FC -shared -o "$OUT/libroadsurf_ref_cpl.so" "$TMP"/obj/*.o -L/usr/lib/gcc/x86_64-linux-gnu/11 -lgomp
echo "built $OUT/libroadsurf_ref_cpl.so"

# ---- the reference's own time loop over THIS library's module surface -------------------
# examples/example1/src/Simulation.f90 (runsimulation + roadModelOneStep) compiled UNCHANGED against the
# product's `module RoadSurfVariables` / `module RoadSurf` (roadsurf_amd/fortran/RoadSurfCompat.f90; the .mod
# files of roadsurf_amd/build) and linked against libroadsurf_hip.so: the proof that the module surface is the
# reference's at the source level (tests/test_abi_layout.py) and, on a GPU, that it computes the reference's
# bits (tests/test_hip_module_surface.py).  Skipped while the product has not been built.
PROD="$HERE/../roadsurf_amd"
if [ -f "$PROD/build/roadsurf.mod" ] && [ -f "$PROD/lib/libroadsurf_hip.so" ]; then
  mkdir -p "$TMP/over"
  # (through the symlink in the temp directory: its Constants.h is the one without `#pragma once`)
  $FC $FFLAGS -I"$TMP/src" -I"$PROD/build" -module-dir "$TMP/over" -c "$TMP/src/Simulation.f90" \
      -o "$TMP/over/Simulation.o"
  $FC -shared -o "$OUT/libsimulation_over_hip.so" "$TMP/over/Simulation.o" -L"$PROD/lib" -lroadsurf_hip \
      -Wl,-rpath,'$ORIGIN/../../roadsurf_amd/lib'
  echo "built $OUT/libsimulation_over_hip.so"
  # which product sources its module files came from (tests/conftest.py rebuilds when they have moved on)
  python3 "$PROD/provenance.py" --build > "$OUT/over_hip.stamp"
fi

# ---- the driver's one self-contained C++ file -----------------------------------------
# examples/example1/src/MeteorologyTools.cpp (CalcTdewOrRH) needs only <cmath>; it pins
# oracle/driver_oracle.c's restatement.  The rest of the driver (JsonSource.cpp,
# roadrunner.cpp) needs jsoncpp, which this image lacks: unbuildable, see driver_oracle.c.
# Strict flags (the reference's own are -funsafe-math-optimizations ..., Makefile:6).
g++ -O2 -fPIC -shared -ffp-contract=off -I"$REF/examples/example1/src" \
    "$REF/examples/example1/src/MeteorologyTools.cpp" -o "$OUT/libroadrunner_tools_ref.so"
echo "built $OUT/libroadrunner_tools_ref.so"
