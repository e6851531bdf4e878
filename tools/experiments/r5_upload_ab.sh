#!/usr/bin/env bash
# A/B on one box (distinct series): uploads inline on the worker's stream (default) vs on a copy stream
export ROADSURF_HIP_DEVICE=0 BENCH_REPS=3
for rep in 1 2; do for m in relax coupling skyview; do
  a=$(timeout -k 5 200 python3 tools/bench_driver_path.py 1000000 48 $m 2>&1 | grep best)
  b=$(ROADSURF_HIP_UPLOAD_STREAM=1 timeout -k 5 200 python3 tools/bench_driver_path.py 1000000 48 $m 2>&1 | grep best)
  echo "$m: inline: $a | copy stream: $b"
done; done
