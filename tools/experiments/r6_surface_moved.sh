#!/usr/bin/env bash
# round 6: what MOVING road_condition and the surface's vapour pressure from the surface to the ground wavefront could buy
# a small shard: the light-surface build (r6_surface_light.sh) whose ground wavefront does the same amount of work on
# stand-in values (results discarded), against the library and the light-surface build.
#   make -C roadsurf_amd OBJ=build_surfx LIB=lib/libroadsurf_hip_surfx.so EXTRA=-DRS_EXP_SURFACE_LIGHT -j8
#   make -C roadsurf_amd OBJ=build_surfm LIB=lib/libroadsurf_hip_surfm.so EXTRA="-DRS_EXP_SURFACE_LIGHT -DRS_EXP_SURFACE_MOVED" -j8
for pts in 125000 250000 1000000; do
  for lib in "" _surfx _surfm; do
    ROADSURF_HIP_LIB=$PWD/roadsurf_amd/lib/libroadsurf_hip$lib.so python3 bench.py --total-points $pts --steps 6 --warmup 2 --no-cpu-baseline --no-natural-leg --no-extra-legs > gpurun_out/r6_surfm_${pts}$lib.json 2>/dev/null || exit 1
    python3 - <<PY
import json
d=json.load(open("gpurun_out/r6_surfm_${pts}$lib.json"))
name={"":"library","_surfx":"work removed","_surfm":"work moved to the ground wave"}["$lib"]
print("%8d points %-30s %.4e point-timesteps/s  avg launch %.3f ms"%($pts, name, d["value"], d["roofline"]["avg_launch_ms"]))
PY
  done
done
