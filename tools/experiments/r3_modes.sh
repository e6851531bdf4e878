#!/usr/bin/env bash
# sort key: which fields, in which order (bench.py --forecast-mode; forecast_key_kernel)
set -e
OUT=gpurun_out/r3_modes
mkdir -p $OUT
export TMPDIR=/tmp
G="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
for M in ${MODES:-3124 3786 37865 3784 3126}; do
  python bench.py --no-cpu-baseline --no-natural-leg --steps 3 --forecast-mode $M > $OUT/m$M.json 2>/dev/null
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-natural-leg --forecast-mode $M > $OUT/pmc_bench.json 2> $OUT/pmc_err.txt || { tail -20 $OUT/pmc_err.txt; }
  python3 - <<PY
import csv,glob,collections,json
acc=collections.defaultdict(float)
for fn in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        if "step_kernel" not in row["Kernel_Name"]: continue
        acc[row["Counter_Name"]]+=float(row["Counter_Value"])
ws=1000000/64.0*5761
d=json.load(open("$OUT/m$M.json"))
print("mode $M value %.4e"%d["value"], {k: round(v/ws,1) for k,v in sorted(acc.items()) if k!="SQ_WAVES"}, flush=True)
PY
  rm -rf $OUT/pmc
done
