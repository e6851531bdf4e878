#!/usr/bin/env bash
# round 6: the library against an experiment build at the three sizes:  r6_ab_lib.sh <suffix of lib/libroadsurf_hip<suffix>.so> [label]
SUF=$1; LABEL=${2:-$1}
for pts in 125000 250000 1000000; do
  for lib in "" $SUF; do
    ROADSURF_HIP_LIB=$PWD/roadsurf_amd/lib/libroadsurf_hip$lib.so python3 bench.py --total-points $pts --steps 6 --warmup 2 --no-cpu-baseline --no-natural-leg --no-extra-legs > gpurun_out/r6_ab_${pts}$lib.json 2>/dev/null || exit 1
    python3 - <<PY
import json
d=json.load(open("gpurun_out/r6_ab_${pts}$lib.json"))
print("%8d points %-22s %.4e point-timesteps/s  avg launch %.3f ms"%($pts, "$LABEL" if "$lib" else "library", d["value"], d["roofline"]["avg_launch_ms"]))
PY
  done
done
