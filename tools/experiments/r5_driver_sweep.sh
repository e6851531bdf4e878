#!/usr/bin/env bash
# rs_driver_run (raw-series step kernel), 1 M points x 48 h: blocks per device x indices per launch.
# usage: r5_driver_sweep.sh mode "blocks..." "chunks..."
MODE=${1:-relax}; BLOCKS=${2:-"4 6 8"}; CHUNKS=${3:-"120 240 480"}
export ROADSURF_HIP_DEVICE=0 BENCH_UNIQUE=65536 BENCH_REPS=3
for b in $BLOCKS; do for c in $CHUNKS; do
  q=4; [ $b -gt 4 ] && q=8
  r=$(GPU_MAX_HW_QUEUES=$q ROADSURF_HIP_PLANS_PER_DEVICE=$b ROADSURF_HIP_CHUNK_STEPS=$c python3 tools/bench_driver_path.py 1000000 48 $MODE 2>&1 | grep best)
  echo "mode $MODE blocks $b queues $q chunk $c: $r"
done; done
