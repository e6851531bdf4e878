#!/bin/bash
# rs_driver_run at 1 M points: blocks per device, with and without more hardware queues
mkdir -p gpurun_out
OUT=gpurun_out/r4_blocks.txt
: > $OUT
run() { # mode label env...
  M=$1; L=$2; shift; shift
  v=$(env "$@" BENCH_UNIQUE=65536 BENCH_REPS=3 python3 tools/bench_driver_path.py 1000000 48 $M 2>&1 | grep best)
  echo "$M $L: $v" | tee -a $OUT
}
for M in skyview relax coupling; do
  run $M blocks4 X=1
  run $M blocks6 ROADSURF_HIP_PLANS_PER_DEVICE=6
  run $M blocks8 ROADSURF_HIP_PLANS_PER_DEVICE=8
  run $M blocks8_hwq8 ROADSURF_HIP_PLANS_PER_DEVICE=8 GPU_MAX_HW_QUEUES=8
done
