#!/bin/bash
# SQ counters of the driver path's step kernels (one mode per run): usage r5_pmc_driver.sh mode
export TMPDIR=/tmp
MODE=${1:-skyview}
OUT=gpurun_out/r5_pmc_drv_$MODE; rm -rf $OUT; mkdir -p $OUT
export BENCH_REPS=1  # (distinct series for every point, as the timed legs)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 tools/bench_driver_path.py 1000000 48 $MODE > $OUT/log.txt 2> $OUT/err.txt || { tail $OUT/err.txt; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT --output-format csv -d $OUT/pmc_sq2 -- python3 tools/bench_driver_path.py 1000000 48 $MODE > $OUT/log2.txt 2> $OUT/err2.txt || { tail $OUT/err2.txt; exit 1; }
python3 tools/summarize_pmc.py $OUT > $OUT/pmc_summary.txt
grep "step_kernel" $OUT/pmc_summary.txt | cut -c1-150
rm -rf $OUT/pmc_sq $OUT/pmc_sq2
