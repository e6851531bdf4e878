#!/usr/bin/env bash
# rs_driver_run at HEAD, 1 M points x 48 h, distinct series: the three modes of bench.py's driver legs (+ sky view
# with coupling), each against round 4's organisation (forcing windows) on the same box and inputs.
for M in relax skyview coupling skycoupling; do
  BENCH_REPS=3 timeout -k 10 250 python3 tools/bench_driver_path.py 1000000 48 $M 2>&1 | grep -E "rep|best" | sed "s/^/$M: /"
  ROADSURF_HIP_DRIVER_WINDOWS=1 BENCH_REPS=2 timeout -k 10 250 python3 tools/bench_driver_path.py 1000000 48 $M 2>&1 | grep best | sed "s/^/$M, forcing windows (round 4): /"
done
