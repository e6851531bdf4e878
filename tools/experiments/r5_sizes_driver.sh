#!/usr/bin/env bash
# rs_driver_run across batch sizes, default (raw-series step kernel) against round 4's organisation (forcing
# windows): no size should have become slower
for N in 2048 4096 16384 65536 262144; do for M in relax coupling skyview; do
  A=$(BENCH_REPS=3 timeout -k 10 120 python3 tools/bench_driver_path.py $N 48 $M 2>&1 | grep best | sed 's/best //')
  B=$(ROADSURF_HIP_DRIVER_WINDOWS=1 BENCH_REPS=3 timeout -k 10 120 python3 tools/bench_driver_path.py $N 48 $M 2>&1 | grep best | sed 's/best //')
  echo "n $N $M: $A | windows: $B"
done; done
