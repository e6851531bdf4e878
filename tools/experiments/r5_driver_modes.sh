#!/usr/bin/env bash
# rs_driver_run on distinct series for every point (as bench.py times it), every mode, raw-series stepping vs windows
export ROADSURF_HIP_DEVICE=0 BENCH_REPS=3
for m in ${1:-relax coupling skyview}; do
  a=$(timeout -k 5 200 python3 tools/bench_driver_path.py 1000000 48 $m 2>&1 | grep -E "^rep [12]|best" | tr '\n' ' ')
  echo "$m raw: $a"
  if [ -n "$2" ]; then
    b=$(ROADSURF_HIP_DRIVER_WINDOWS=1 timeout -k 5 200 python3 tools/bench_driver_path.py 1000000 48 $m 2>&1 | grep best)
    echo "$m windows: $b"
  fi
done
