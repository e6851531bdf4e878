#!/usr/bin/env bash
# two-wavefront flavour: parity (forced through AUTO for every launch that admits it), then rates by shard size
set -e
OUT=gpurun_out/r3_duo
mkdir -p $OUT
export TMPDIR=/tmp
ROADSURF_HIP_DUO_MAX=1000000000 timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_random_configs.py tests/test_hip_edge_shapes.py tests/test_hip_golden_and_scale.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
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
for N in 125000 250000 500000 1000000; do
  run n${N}_v1 --total-points $N --steps 5 --variant 1
  run n${N}_v3 --total-points $N --steps 5 --variant 3
done
run n125k_v3_k1 --total-points 125000 --steps 5 --variant 3 --plans-per-gpu 1
run n125k_v3_k4 --total-points 125000 --steps 5 --variant 3 --plans-per-gpu 4
run n125k_v3_k2_c120 --total-points 125000 --steps 5 --variant 3 --chunk 120
run n250k_v3_k2 --total-points 250000 --steps 5 --variant 3 --plans-per-gpu 2
