#!/usr/bin/env bash
# round 6: rs_driver_run on the weather of bench.py's device-resident legs (BENCH_WEATHER=bench) against its own
# synthetic workload - how much of the distance to the FULL leg (2.26e10) is the workload, how much the code path.
OUT=gpurun_out/r6_driver_weather
mkdir -p $OUT
export BENCH_REPS=3
for W in driver bench; do
  for M in relax coupling skyview; do
    for K in 4 1; do
      [ $K = 1 ] && [ $M != relax ] && continue
      BENCH_WEATHER=$W ROADSURF_HIP_PLANS_PER_DEVICE=$K timeout -k 10 300 python3 tools/bench_driver_path.py 1000000 48 $M 2> $OUT/${W}_${M}_$K.err | grep -E "best" | sed "s/^/weather $W, $M, $K block(s): /" || exit 1
    done
  done
done
