#!/usr/bin/env bash
# round 6: plans x launch length (and the surface wave's issue priority) of the fp32 flavour at config 5's shape
OUT=gpurun_out/r6_sweep_${1:-x}
mkdir -p $OUT
run() { # plans chunk tag [env...]
  local K=$1 CH=$2 TAG=$3; shift 3
  env "$@" timeout -k 10 200 python bench.py --f32 --points 1250000 --hours 168 --no-natural-leg --no-extra-legs --no-cpu-baseline --plans-per-gpu $K --chunk $CH > $OUT/f32_${TAG}_p${K}_c${CH}.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$OUT/f32_${TAG}_p${K}_c${CH}.json")); r=d["roofline"]
print("%-8s plans $K chunk $CH: %.3e  ms/pass %.1f  avg launch %.2f ms  concurrent %.2f"%("$TAG", d["value"], d["ms_per_step"], r["avg_launch_ms"], r["concurrent_launches"]))
PY
}
run 2 240 base A=1
run 2 240 prio1 ROADSURF_HIP_F32_SURFACE_PRIO=1
run 2 240 prio0 ROADSURF_HIP_F32_SURFACE_PRIO=0
run 2 480 base A=1
run 3 240 base A=1
run 2 360 base A=1
run 3 360 base A=1
run 1 240 base A=1
