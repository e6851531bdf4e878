#!/usr/bin/env bash
# A/B on one box: uploads on a copy stream with eager transposes (default) vs inline on the worker's stream
export ROADSURF_HIP_DEVICE=0 BENCH_UNIQUE=65536 BENCH_REPS=3
for rep in 1 2 3; do for m in relax skyview; do
  a=$(python3 tools/bench_driver_path.py 1000000 48 $m 2>&1 | grep best)
  b=$(ROADSURF_HIP_UPLOAD_INLINE=1 python3 tools/bench_driver_path.py 1000000 48 $m 2>&1 | grep best)
  echo "$m: copy stream: $a | inline: $b"
done; done
