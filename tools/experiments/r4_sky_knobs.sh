#!/bin/bash
# rs_driver_run with sky view at 1 M points: fan-out knobs
mkdir -p gpurun_out
OUT=gpurun_out/r4_sky_knobs.txt
: > $OUT
run() { # label, env...
  L=$1; shift
  v=$(env "$@" BENCH_UNIQUE=65536 BENCH_REPS=3 python3 tools/bench_driver_path.py 1000000 48 skyview 2>&1 | grep best)
  echo "$L: $v" | tee -a $OUT
}
run default X=1
run first50 ROADSURF_HIP_FIRST_BLOCK_PCT=50
run first30 ROADSURF_HIP_FIRST_BLOCK_PCT=30
run plans6 ROADSURF_HIP_PLANS_PER_DEVICE=6 GPU_MAX_HW_QUEUES=6
run plans8 ROADSURF_HIP_PLANS_PER_DEVICE=8 GPU_MAX_HW_QUEUES=8
run plans3 ROADSURF_HIP_PLANS_PER_DEVICE=3
run default X=1
