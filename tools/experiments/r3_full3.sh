#!/usr/bin/env bash
set -e
OUT=gpurun_out/r3_full3
mkdir -p $OUT
export TMPDIR=/tmp
run() { # name, bench args
  local name=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-natural-leg "$@" > $OUT/$name.json 2> $OUT/$name.err || { tail -5 $OUT/$name.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$OUT/$name.json")); r=d["roofline"]
print("$name value %.4e ms/pass %.1f kernel-only %.4e avg launch %.2f ms conc %.2f"%(d["value"],d["ms_per_step"],r["step_kernel_only_value"],r["avg_launch_ms"],r["concurrent_launches"]), flush=True)
PY
}
for V in 4 31; do for K in 1 2 3; do for C in 120 240; do
  run full_v${V}_k${K}_c${C} --full --steps 3 --variant $V --plans-per-gpu $K --chunk $C
done; done; done
for P in 1 2 3 4; do
  echo "driver path relax PLANS_PER_DEVICE=$P"
  ROADSURF_HIP_PLANS_PER_DEVICE=$P python tools/bench_driver_path.py 1000000 48 relax 2>&1 | grep "rep [123]"
done
ROADSURF_HIP_DRIVER_TIMING=1 ROADSURF_HIP_PLANS_PER_DEVICE=2 python tools/bench_driver_path.py 1000000 48 relax 2>&1 | tail -30
