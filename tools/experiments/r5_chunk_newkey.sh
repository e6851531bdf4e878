#!/usr/bin/env bash
# launch length of the driver path with the class key (round 5's second pass): 60 / 120 / 180 / 240 indices
for T in 240 120 60 180 240 120; do
  for M in relax coupling; do
    ROADSURF_HIP_CHUNK_STEPS=$T BENCH_REPS=3 timeout -k 10 200 python3 tools/bench_driver_path.py 1000000 48 $M 2>&1 | grep best | sed "s/^/chunk $T $M /"
  done
done
