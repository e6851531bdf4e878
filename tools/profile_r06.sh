#!/usr/bin/env bash
# Round-6 evidence, all from the kernels at HEAD: rocprofv3 kernel stats + timeline + PMC passes
# (separate runs, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) for
#   lean   bench.py default (1 M points x 48 h, fp64, 3 plans x 60)      -> profiles/r06_*
#   f32    BASELINE config 5 shape (1.25 M points x 7 d, fp32: two points per lane, 2 plans x 360) -> profiles/r06_f32_*
#   small  125 000 points (config 4's per-GPU shard), one point per lane
#          against two wavefronts per 64 points                          -> profiles/r06_small_shard_*
#   full   FULL feature set (bench.py --full)                            -> profiles/r06_full_*
# usage: profile_r06.sh [lean] [f32] [small] [full]   (default: all)
set -e
export TMPDIR=/tmp
WHAT="${*:-lean f32 small full}"
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
B1="--steps 1 --warmup 0 --no-cpu-baseline --no-natural-leg --no-extra-legs"

profile() { # tag, bench flags...
  local TAG=$1; shift
  local OUT=gpurun_out/prof_$TAG
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-natural-leg --no-extra-legs "$@" > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
  find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
  python3 tools/trace_timeline.py $OUT/trace > $OUT/timeline.txt
  cut -c1-150 $OUT/kernel_stats.csv | head -8
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/pmc_sq -- python3 bench.py $B1 "$@" > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err || { tail -20 $OUT/pmc_sq.err; exit 1; }
  rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $B1 "$@" > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err || { tail -20 $OUT/pmc_fetch.err; exit 1; }
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $B1 "$@" > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err || { tail -20 $OUT/pmc_write.err; exit 1; }
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $B1 "$@" > $OUT/bench_pmc_sq2.json 2> $OUT/pmc_sq2.err || { tail -20 $OUT/pmc_sq2.err; exit 1; }
  python3 tools/summarize_pmc.py $OUT > $OUT/pmc_summary.txt
  grep step_kernel $OUT/pmc_summary.txt | cut -c1-140
  rm -rf $OUT/trace $OUT/pmc_sq $OUT/pmc_sq2 $OUT/pmc_fetch $OUT/pmc_write
}

for W in $WHAT; do case $W in
lean)
  profile r06
  python3 bench.py --steps 20 --warmup 5 > gpurun_out/prof_r06/bench.json 2> gpurun_out/prof_r06/bench.err
  ;;
f32)
  profile r06_f32 --f32 --points 1250000 --hours 168
  python3 bench.py --f32 --points 1250000 --hours 168 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_r06_f32/bench.json 2> gpurun_out/prof_r06_f32/bench.err
  ;;
small)
  for V in 1 3; do
    OUT=gpurun_out/prof_r06_small_v$V
    rm -rf $OUT; mkdir -p $OUT
    python3 bench.py --total-points 125000 --variant $V --steps 10 --warmup 2 --no-cpu-baseline --no-natural-leg --no-extra-legs > $OUT/bench.json 2> $OUT/bench.err
    rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/pmc_sq -- python3 bench.py $B1 --total-points 125000 --variant $V > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err || { tail -20 $OUT/pmc_sq.err; exit 1; }
    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $B1 --total-points 125000 --variant $V > $OUT/bench_pmc_sq2.json 2> $OUT/pmc_sq2.err || { tail -20 $OUT/pmc_sq2.err; exit 1; }
    python3 tools/summarize_pmc.py $OUT > $OUT/pmc_summary.txt
    grep step_kernel $OUT/pmc_summary.txt | cut -c1-140
    rm -rf $OUT/pmc_sq $OUT/pmc_sq2
  done
  python3 bench.py --total-points 250000 --steps 10 --warmup 2 --no-cpu-baseline --no-natural-leg --no-extra-legs > gpurun_out/prof_r06_small_v1/bench_250k.json 2>/dev/null
  ;;
full)
  profile r06_full --full
  python3 bench.py --full --steps 5 --warmup 2 --no-cpu-baseline --no-natural-leg --no-extra-legs > gpurun_out/prof_r06_full/bench.json 2> gpurun_out/prof_r06_full/bench.err
  ;;
esac; done
