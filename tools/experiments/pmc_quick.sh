#!/usr/bin/env bash
# quick VALU instruction count of the step kernel: rocprofv3 --pmc on one bench pass
set -e
OUT=gpurun_out/pmcq_${1:-x}
shift || true
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-natural-leg "$@" > $OUT/bench.json 2> $OUT/err.txt || { tail -20 $OUT/err.txt; exit 1; }
python3 tools/summarize_pmc.py $OUT | grep -E "step_kernel" 
python3 - <<PY
import json
d=json.load(open("$OUT/bench.json")); r=d["roofline"]
print("under profiler: kernel-only %.3e avg launch ms %.2f"%(r["step_kernel_only_value"], r["avg_launch_ms"]))
PY
