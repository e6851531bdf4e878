#!/usr/bin/env bash
set -e
OUT=gpurun_out/r3_sort
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_cluster.py tests/test_hip_golden_and_scale.py tests/test_hip_driver.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
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
run own_c120 --steps 5
ROADSURF_HIP_LIBRARY_SORT=1 run lib_c120 --steps 5
run own_c60 --steps 5 --chunk 60
ROADSURF_HIP_LIBRARY_SORT=1 run lib_c60 --steps 5 --chunk 60
run own_c90 --steps 5 --chunk 90
run own_c80 --steps 5 --chunk 80
run own_c120b --steps 5
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-natural-leg > $OUT/bench_trace.json 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
python3 tools/trace_timeline.py $OUT/trace | head -16
rm -rf $OUT/trace
