#!/usr/bin/env bash
# kernel trace of a small rs_driver_run call (latency regime): usage r5_small_trace.sh points mode
export TMPDIR=/tmp
N=${1:-2048}; MODE=${2:-relax}
OUT=gpurun_out/r5_small_trace; rm -rf $OUT; mkdir -p $OUT
BENCH_REPS=2 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_driver_path.py $N 48 $MODE > $OUT/log.txt 2> $OUT/err.txt || { tail $OUT/err.txt; exit 1; }
grep best $OUT/log.txt
STATS=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cut -c1-150 $STATS | head -8
rm -rf $OUT/trace
