#!/usr/bin/env bash
# coupling on distinct series, alternating on one box: replay rounds from the raw series (two wavefronts per 64
# listed points) vs a forcing window + the one-point-per-lane replay kernel; and the round-4 organisation
export ROADSURF_HIP_DEVICE=0 BENCH_REPS=3
for rep in 1 2; do
  echo "replays from the raw series: $(timeout -k 5 200 python3 tools/bench_driver_path.py 1000000 48 coupling 2>&1 | grep best)"
  echo "replays from a window:       $(ROADSURF_HIP_CPL_REPLAY_WINDOWS=1 timeout -k 5 200 python3 tools/bench_driver_path.py 1000000 48 coupling 2>&1 | grep best)"
  echo "windows throughout (r04):    $(ROADSURF_HIP_DRIVER_WINDOWS=1 timeout -k 5 200 python3 tools/bench_driver_path.py 1000000 48 coupling 2>&1 | grep best)"
done
