#!/usr/bin/env bash
# Kernel trace + stats of the raw-series driver path (tools/bench_driver_path.py).
set -e
TAG=${1:-r01_driver}
MODE=${2:-relax}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_driver_path.py 262144 48 $MODE > $OUT/bench.log 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
cat $OUT/kernel_stats.csv
cat $OUT/bench.log
