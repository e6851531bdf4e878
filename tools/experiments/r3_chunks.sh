#!/usr/bin/env bash
# how much coherence does a fresher sort key buy?  vector instructions per wave-step by launch length
set -e
OUT=gpurun_out/r3_chunks
mkdir -p $OUT
export TMPDIR=/tmp
G="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
for C in 30 60 120 240 480; do
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-natural-leg --chunk $C > $OUT/pmc_bench.json 2> $OUT/pmc_err.txt || { tail -20 $OUT/pmc_err.txt; }
  python3 - <<PY
import csv,glob,collections,json
acc=collections.defaultdict(float)
for fn in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        if "step_kernel" not in row["Kernel_Name"]: continue
        acc[row["Counter_Name"]]+=float(row["Counter_Value"])
ws=1000000/64.0*5761
print("chunk $C", {k: round(v/ws,1) for k,v in sorted(acc.items()) if k!="SQ_WAVES"}, flush=True)
PY
  rm -rf $OUT/pmc
done
rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-natural-leg --cluster 0 > $OUT/pmc_bench.json 2> $OUT/pmc_err.txt
python3 - <<PY
import csv,glob,collections,json
acc=collections.defaultdict(float)
for fn in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        if "step_kernel" not in row["Kernel_Name"]: continue
        acc[row["Counter_Name"]]+=float(row["Counter_Value"])
ws=1000000/64.0*5761
print("natural order", {k: round(v/ws,1) for k,v in sorted(acc.items()) if k!="SQ_WAVES"}, flush=True)
PY
rm -rf $OUT/pmc
