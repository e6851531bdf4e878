#!/usr/bin/env bash
# VALU instruction counts of the raw-series path's kernels: rocprofv3 --pmc on tools/bench_driver_path.py
# usage: pmc_driver.sh [mode] [points]
set -e
MODE=${1:-coupling}; N=${2:-1000000}
OUT=gpurun_out/pmcd_$MODE
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 tools/bench_driver_path.py $N 48 $MODE > $OUT/bench.txt 2> $OUT/err.txt || { tail -20 $OUT/err.txt; exit 1; }
python3 tools/summarize_pmc.py $OUT | grep -E "step_kernel|expand_raw|again_flags|scan_raw" | grep -E "SQ_INSTS_VALU|SQ_WAVES|SQ_ACTIVE_INST_VALU"
grep "^rep" $OUT/bench.txt | tail -2
rm -rf $OUT/pmc_sq
