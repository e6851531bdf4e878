#!/usr/bin/env bash
# relax mode on distinct series, alternating on one box: raw-series stepping (chunk 120 / 240) vs windows
export ROADSURF_HIP_DEVICE=0 BENCH_REPS=3
for rep in 1 2; do
  for c in 120 240; do
    echo "raw chunk $c: $(ROADSURF_HIP_CHUNK_STEPS=$c timeout -k 5 200 python3 tools/bench_driver_path.py 1000000 48 relax 2>&1 | grep best)"
  done
  echo "windows: $(ROADSURF_HIP_DRIVER_WINDOWS=1 timeout -k 5 200 python3 tools/bench_driver_path.py 1000000 48 relax 2>&1 | grep best)"
done
echo "tiled raw: $(BENCH_UNIQUE=65536 timeout -k 5 200 python3 tools/bench_driver_path.py 1000000 48 relax 2>&1 | grep best)"
echo "tiled windows: $(BENCH_UNIQUE=65536 ROADSURF_HIP_DRIVER_WINDOWS=1 timeout -k 5 200 python3 tools/bench_driver_path.py 1000000 48 relax 2>&1 | grep best)"
