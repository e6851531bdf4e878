#!/usr/bin/env bash
# rocprofv3 kernel stats of the raw-series path (rs_driver_run), 1 M points x 48 h
set -e
TAG=${1:-r02}
OUT=gpurun_out/prof_${TAG}_driver
mkdir -p $OUT
export TMPDIR=/tmp
for mode in relax coupling; do
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$mode -- python3 tools/bench_driver_path.py 1000000 48 $mode > $OUT/$mode.txt 2> $OUT/$mode.err || { tail -20 $OUT/$mode.err; exit 1; }
find $OUT/$mode -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${mode}_kernel_stats.csv
grep "^rep" $OUT/$mode.txt | tail -2
head -8 $OUT/${mode}_kernel_stats.csv | cut -c1-150
rm -rf $OUT/$mode
done
