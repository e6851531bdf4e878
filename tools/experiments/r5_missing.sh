#!/usr/bin/env bash
# rs_driver_run on observation series with holes (BENCH_MISSING=f: a fraction f of the stations without an air
# temperature / humidity / wind sensor each, a fraction f of the road-temperature observations missing): the lanes
# of a wavefront disagree on the supplying source, the raw-series step kernel takes its per-index path there
for F in 0 0.1; do for M in relax coupling; do
  BENCH_MISSING=$F BENCH_REPS=3 timeout -k 10 250 python3 tools/bench_driver_path.py 1000000 48 $M 2>&1 | grep best | sed "s/^/missing $F $M: /"
  BENCH_MISSING=$F ROADSURF_HIP_DRIVER_WINDOWS=1 BENCH_REPS=2 timeout -k 10 250 python3 tools/bench_driver_path.py 1000000 48 $M 2>&1 | grep best | sed "s/^/missing $F $M, forcing windows: /"
done; done
