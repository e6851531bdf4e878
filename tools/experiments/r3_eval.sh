#!/usr/bin/env bash
# round 3 evaluation of a kernel change: [tests] + bench at 1 M / 250 k / 125 k + instruction counts per wave-step
# usage: r3_eval.sh NAME [notest] [nopmc]
set -e
NAME=${1:-x}; shift || true
OUT=gpurun_out/r3_$NAME
mkdir -p $OUT
export TMPDIR=/tmp
if [[ " $* " != *" notest "* ]]; then
  python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
  tail -2 $OUT/tests.log
fi
run() { # name, bench args
  local name=$1; shift
  python bench.py --no-cpu-baseline --no-natural-leg "$@" > $OUT/$name.json 2> $OUT/$name.err || { tail -5 $OUT/$name.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$OUT/$name.json")); r=d["roofline"]
print("$name value %.4e ms/pass %.1f kernel-only %.4e avg launch %.2f ms conc %.2f"%(d["value"],d["ms_per_step"],r["step_kernel_only_value"],r["avg_launch_ms"],r["concurrent_launches"]), flush=True)
PY
}
run n1m --steps 5
run n250k --total-points 250000 --steps 10
run n125k --total-points 125000 --steps 10
G="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"
if [[ " $* " != *" nopmc "* ]]; then
for N in 1000000 125000; do
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/pmc_$N -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-natural-leg --total-points $N > $OUT/pmc_bench_$N.json 2> $OUT/pmc_err_$N.txt || { tail -20 $OUT/pmc_err_$N.txt; echo "pmc $N failed"; continue; }
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(float); n=0
for fn in glob.glob("$OUT/pmc_$N/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        if "step_kernel" not in row["Kernel_Name"]: continue
        acc[row["Counter_Name"]]+=float(row["Counter_Value"])
ws=$N/64.0*5761  # wave-steps of one pass (padding ignored)
print("N=$N per wave-step:", {k: round(v/ws,1) for k,v in sorted(acc.items()) if k!="SQ_WAVES"}, flush=True)
PY
  rm -rf $OUT/pmc_$N
done
fi
# A/B: environment switches given as ab:VAR=VALUE ... -> bench + counters at 1 M with each
for arg in "$@"; do
  case $arg in ab:*)
    kv=${arg#ab:}
    export "$kv"
    run n1m_${kv%%=*} --steps 3
    N=1000000
    rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/pmc_ab -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-natural-leg --total-points $N > $OUT/pmc_bench_ab.json 2> $OUT/pmc_err_ab.txt || { tail -20 $OUT/pmc_err_ab.txt; }
    python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(float)
for fn in glob.glob("$OUT/pmc_ab/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        if "step_kernel" not in row["Kernel_Name"]: continue
        acc[row["Counter_Name"]]+=float(row["Counter_Value"])
ws=$N/64.0*5761
print("$kv N=$N per wave-step:", {k: round(v/ws,1) for k,v in sorted(acc.items()) if k!="SQ_WAVES"}, flush=True)
PY
    rm -rf $OUT/pmc_ab
    unset "${kv%%=*}"
  ;; esac
done
