#!/usr/bin/env bash
# the precipitation bit on the driver path (distinct series), alternating on one box
export ROADSURF_HIP_DEVICE=0 BENCH_REPS=3
for rep in 1 2; do for m in relax skyview coupling; do
  a=$(ROADSURF_HIP_PRECIP_BIT=1 timeout -k 5 200 python3 tools/bench_driver_path.py 1000000 48 $m 2>&1 | grep best)
  b=$(ROADSURF_HIP_PRECIP_BIT=0 timeout -k 5 200 python3 tools/bench_driver_path.py 1000000 48 $m 2>&1 | grep best)
  echo "$m: with the bit: $a | without: $b"
done; done
