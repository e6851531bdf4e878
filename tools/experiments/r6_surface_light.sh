#!/usr/bin/env bash
# round 6: an UPPER BOUND of what a third wavefront per 64 points could buy a small shard - a build whose surface
# wavefront does not run road_condition, the surface vapour pressure's exp and the heat capacities of layers 1-2 (wrong
# physics; about 200 of its ~750 instructions per step: what a third wavefront could take over) against the library.
#   make -C roadsurf_amd OBJ=build_surfx LIB=lib/libroadsurf_hip_surfx.so EXTRA=-DRS_EXP_SURFACE_LIGHT -j8
for pts in 125000 250000 1000000; do
  for lib in "" _surfx; do
    ROADSURF_HIP_LIB=$PWD/roadsurf_amd/lib/libroadsurf_hip$lib.so python3 bench.py --total-points $pts --steps 6 --warmup 2 --no-cpu-baseline --no-natural-leg --no-extra-legs > gpurun_out/r6_surf_${pts}$lib.json 2>/dev/null
    python3 - <<PY
import json
d=json.load(open("gpurun_out/r6_surf_${pts}$lib.json"))
print("%8d points %-22s %.4e point-timesteps/s  avg launch %.3f ms"%($pts, "light surface wave" if "$lib" else "library", d["value"], d["roofline"]["avg_launch_ms"]))
PY
  done
done
