#!/usr/bin/env bash
# round 6: the fp32 flavour's gates + its bench (config 5 shape) in one GPU call
# usage: r6_f32_check.sh TAG [extra bench flags]
TAG=${1:-x}; shift || true
OUT=gpurun_out/r6_$TAG
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_hip_f32.py -x -q -s > $OUT/test_f32.txt 2>&1
tail -12 $OUT/test_f32.txt | grep -v "warning\|amdgpu.ids"
for cfg in "4 120" "4 240" "2 240"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --f32 --points 1250000 --hours 168 --no-natural-leg --no-extra-legs --no-cpu-baseline --plans-per-gpu $1 --chunk $2 "${@:3}" > $OUT/f32_p$1_c$2.json 2> $OUT/f32_p$1_c$2.err
  python - <<PY
import json
d=json.load(open("$OUT/f32_p$1_c$2.json")); r=d["roofline"]
print("plans $1 chunk $2: %.3e  ms/pass %.1f  avg launch %.2f ms  concurrent %.2f"%(d["value"], d["ms_per_step"], r["avg_launch_ms"], r["concurrent_launches"]))
PY
done
