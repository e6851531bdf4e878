#!/usr/bin/env bash
# blocks of one device shrinking from first to last (ROADSURF_HIP_BLOCK_TAPER_PCT), distinct series, one box
export ROADSURF_HIP_DEVICE=0 BENCH_REPS=3
for m in relax coupling skyview; do for t in 0 15 30 45; do
  echo "$m taper $t: $(ROADSURF_HIP_BLOCK_TAPER_PCT=$t timeout -k 5 200 python3 tools/bench_driver_path.py 1000000 48 $m 2>&1 | grep best)"
done; done
