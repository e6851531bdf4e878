#!/usr/bin/env bash
set -e
OUT=gpurun_out/r3_full2
mkdir -p $OUT
export TMPDIR=/tmp
ROADSURF_HIP_VARIANT=4 timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_hip_random_configs.py -x -q -m gpu -k "full_variant or random_configuration" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
run() { # name, bench args
  local name=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-natural-leg "$@" > $OUT/$name.json 2> $OUT/$name.err || { tail -5 $OUT/$name.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$OUT/$name.json")); r=d["roofline"]
print("$name value %.4e ms/pass %.1f kernel-only %.4e avg launch %.2f ms conc %.2f"%(d["value"],d["ms_per_step"],r["step_kernel_only_value"],r["avg_launch_ms"],r["concurrent_launches"]), flush=True)
PY
}
run full_v31 --full --steps 3 --variant 31
run full_v4 --full --steps 3 --variant 4
run full_v4_c240 --full --steps 3 --variant 4 --chunk 240
run full_v31_c240 --full --steps 3 --variant 31 --chunk 240
run full_v4_k2 --full --steps 3 --variant 4 --plans-per-gpu 2
G="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"
for V in 4; do
rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-natural-leg --full --variant $V > $OUT/pmc_bench.json 2> $OUT/pmc_err.txt || { tail -20 $OUT/pmc_err.txt; }
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(float)
for fn in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        if "step_kernel" not in row["Kernel_Name"]: continue
        acc[row["Counter_Name"]]+=float(row["Counter_Value"])
ws=1000000/64.0*5761
print("FULL variant $V per wave-step:", {k: round(v/ws,1) for k,v in sorted(acc.items()) if k!="SQ_WAVES"}, flush=True)
PY
rm -rf $OUT/pmc
done
python tools/bench_driver_path.py 1000000 48 relax 2>&1 | tail -3
