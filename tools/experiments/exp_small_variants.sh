# kernel flavour for small shards: fewer resident waves -> a flavour with more registers per wave
set -e
mkdir -p gpurun_out/exp6
for cfg in "$@"; do
IFS=: read N K CH V <<< "$cfg"
python bench.py --no-cpu-baseline --no-natural-leg --total-points $N --plans-per-gpu $K --chunk $CH --variant $V > gpurun_out/exp6/x.json 2> gpurun_out/exp6/err.txt || { tail -5 gpurun_out/exp6/err.txt; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/exp6/x.json")); r=d["roofline"]
print("N=$N K=$K chunk=$CH variant=$V value %.4e ms/pass %.1f avg launch %.2f ms"%(d["value"],d["ms_per_step"],r["avg_launch_ms"]))
PY
done
