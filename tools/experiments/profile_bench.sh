#!/usr/bin/env bash
# Profiles the default bench.py run: kernel trace + stats, then PMC passes (separate runs, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes).  usage: profile_bench.sh r02 [bench flags]
set -e
TAG=${1:-r02}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="--steps 2 --warmup 1 --no-cpu-baseline --no-natural-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $B "$@" > $OUT/bench_trace.json 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
python3 tools/trace_timeline.py $OUT/trace > $OUT/timeline.txt
cat $OUT/kernel_stats.csv | cut -c1-160 | head -12
B1="--steps 1 --warmup 0 --no-cpu-baseline --no-natural-leg"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py $B1 "$@" > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err || { tail -20 $OUT/pmc_sq.err; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $B1 "$@" > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err || { tail -20 $OUT/pmc_fetch.err; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $B1 "$@" > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err || { tail -20 $OUT/pmc_write.err; exit 1; }
python3 tools/summarize_pmc.py $OUT > $OUT/pmc_summary.txt
grep step_kernel $OUT/pmc_summary.txt
rm -rf $OUT/trace $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write   # raw traces are large; the summaries stay
