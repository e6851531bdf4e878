#!/usr/bin/env bash
# round 6: an UPPER BOUND of what a shorter dependent chain in the boundary-layer loop's stable arm
# buys the small shard - a build whose Stab is numerator x a^3 / (den0 vkvz^3) (wrong bits: no Newton step, remainder or
# correction; three dependent operations fewer than the exact early-start variant could have) against the library.
#   make -C roadsurf_amd OBJ=build_stabx LIB=lib/libroadsurf_hip_stabx.so EXTRA=-DRS_EXP_STAB_CHAIN -j8
for pts in 125000 250000 1000000; do
  for lib in "" _stabx; do
    ROADSURF_HIP_LIB=$PWD/roadsurf_amd/lib/libroadsurf_hip$lib.so python3 bench.py --total-points $pts --steps 6 --warmup 2 --no-cpu-baseline --no-natural-leg --no-extra-legs > gpurun_out/r6_stab_${pts}$lib.json 2>/dev/null
    python3 - <<PY
import json
d=json.load(open("gpurun_out/r6_stab_${pts}$lib.json"))
print("%8d points %-22s %.4e point-timesteps/s  avg launch %.3f ms"%($pts, "approximate Stab" if "$lib" else "library", d["value"], d["roofline"]["avg_launch_ms"]))
PY
  done
done
