#!/usr/bin/env bash
set -e
OUT=gpurun_out/r3_k
mkdir -p $OUT
run() { # name, bench args
  local name=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-natural-leg "$@" > $OUT/$name.json 2> $OUT/$name.err || { tail -5 $OUT/$name.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$OUT/$name.json")); r=d["roofline"]
print("$name value %.4e ms/pass %.1f kernel-only %.4e avg launch %.2f ms conc %.2f"%(d["value"],d["ms_per_step"],r["step_kernel_only_value"],r["avg_launch_ms"],r["concurrent_launches"]), flush=True)
PY
}
for K in 1 2 3 4 5; do for C in 120 240; do
  run lean_k${K}_c${C} --steps 3 --plans-per-gpu $K --chunk $C
done; done
run lean_k8_c120 --steps 3 --plans-per-gpu 8 --chunk 120
run lean_k4_c60 --steps 3 --plans-per-gpu 4 --chunk 60
