#!/usr/bin/env bash
# round 6: kernel timelines of rs_driver_run (relax) on both weathers, four blocks and one
OUT=gpurun_out/r6_driver_weather_trace
mkdir -p $OUT
for W in driver bench; do
  for K in 4 1; do
    BENCH_WEATHER=$W ROADSURF_HIP_PLANS_PER_DEVICE=$K bash tools/profile_driver_r05.sh relax > $OUT/log_${W}_$K.txt 2>&1 || { tail -5 $OUT/log_${W}_$K.txt; exit 1; }
    { echo "## weather $W, $K block(s)"; head -16 gpurun_out/profiles_r05/r05_driver_path_relax_timeline.txt; } >> $OUT/timelines.txt
  done
done
cat $OUT/timelines.txt
