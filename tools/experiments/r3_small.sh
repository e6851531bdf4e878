#!/usr/bin/env bash
# plans x launch length x flavour at config 4's shard sizes
set -e
OUT=gpurun_out/r3_small
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
for K in 2 3 4 6; do for C in 120 240 480; do
  run n250k_v1_k${K}_c${C} --total-points 250000 --steps 8 --variant 1 --plans-per-gpu $K --chunk $C
done; done
for V in 1 3; do for K in 2 3; do for C in 240 480 960; do
  run n125k_v${V}_k${K}_c${C} --total-points 125000 --steps 8 --variant $V --plans-per-gpu $K --chunk $C
done; done; done
run n250k_v3_k3_c240 --total-points 250000 --steps 8 --variant 3 --plans-per-gpu 3 --chunk 240
