#!/usr/bin/env bash
# same-box A/B of two builds of the library: default against roadsurf_amd/lib/libroadsurf_hip_$1.so
set -e
ALT=$1; shift
OUT=gpurun_out/r3_ab_$ALT
mkdir -p $OUT
run() { # name, bench args
  local name=$1; shift
  python bench.py --no-cpu-baseline --no-natural-leg "$@" > $OUT/$name.json 2> $OUT/$name.err || { tail -5 $OUT/$name.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$OUT/$name.json")); r=d["roofline"]
print("$name value %.4e ms/pass %.1f"%(d["value"],d["ms_per_step"]), flush=True)
PY
}
for rep in 1 2; do
  for cfg in "1000000 0 4" "250000 0 8" "125000 1 8"; do
    set -- $cfg
    unset ROADSURF_HIP_LIB
    run head_$1_$rep --total-points $1 --variant $2 --steps $3
    export ROADSURF_HIP_LIB=$PWD/roadsurf_amd/lib/libroadsurf_hip_$ALT.so
    run ${ALT}_$1_$rep --total-points $1 --variant $2 --steps $3
  done
done
