#!/usr/bin/env bash
set -e
OUT=gpurun_out/r3_small2
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
for C in 120 180 240; do
  run n250k_k4_c$C --total-points 250000 --steps 8 --chunk $C
  run n125k_duo_k2_c$C --total-points 125000 --steps 8 --chunk $C
  run n500k_k4_c$C --total-points 500000 --steps 5 --chunk $C
done
run n1m --steps 8
run full --full --steps 5
run full_c120 --full --steps 5 --chunk 120 --plans-per-gpu 4
run full_k3_c120 --full --steps 5 --chunk 120 --plans-per-gpu 3
run f32 --f32 --points 1250000 --hours 168 --steps 2
