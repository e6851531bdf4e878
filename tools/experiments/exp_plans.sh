# sweep: plans per GPU x launch length (bench.py --plans-per-gpu / --chunk); prints one line per config
set -e
mkdir -p gpurun_out/exp1
# usage: exp_plans.sh [K:CHUNK ...]
[ $# -gt 0 ] || set -- 1:240 2:240 2:120 3:120 4:240 4:120 4:60 6:120 8:120
for cfg in "$@"; do
IFS=: read K CH <<< "$cfg"
python bench.py --no-cpu-baseline --no-natural-leg --plans-per-gpu $K --chunk $CH $EXTRA > gpurun_out/exp1/k${K}_c${CH}.json 2> gpurun_out/exp1/k${K}_c${CH}.err || { tail -5 gpurun_out/exp1/k${K}_c${CH}.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/exp1/k${K}_c${CH}.json")); r=d["roofline"]
print("K=$K chunk=$CH value %.4e ms/pass %.1f kernel-only %.4e avg launch %.2f ms busy %.1f conc %.2f frac %.3f"%(d["value"],d["ms_per_step"],r["step_kernel_only_value"],r["avg_launch_ms"],r["busy_ms"],r["concurrent_launches"],r["frac"]))
PY
done
