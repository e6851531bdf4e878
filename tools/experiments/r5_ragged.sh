#!/usr/bin/env bash
# coupling where the stations' road-temperature observations - and so their coupling windows - end hours apart
# (BENCH_RAGGED=f: a fraction f of the stations 1-3 h early): the replay block is no longer compact
for N in 262144 1000000; do for F in 0 0.2; do
  BENCH_RAGGED=$F BENCH_REPS=2 ROADSURF_HIP_DRIVER_TIMING=1 timeout -k 10 280 python3 tools/bench_driver_path.py $N 48 coupling 2>&1 | grep -E "best|coupling round 1:|tiles" | tail -3 | sed "s/^/n $N ragged $F: /"
done; done
