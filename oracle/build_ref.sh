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
