# small shards (strong scaling: 1 M points over 8 GPUs = 125 000 per GPU): launch length and plans
# usage: exp_small.sh N:K:CHUNK ...
set -e
mkdir -p gpurun_out/exp5
[ $# -gt 0 ] || set -- 125000:1:120 125000:1:240 125000:2:240 250000:1:120 250000:2:120 500000:2:240 500000:3:120
for cfg in "$@"; do
IFS=: read N K CH <<< "$cfg"
python bench.py --no-cpu-baseline --no-natural-leg --total-points $N --plans-per-gpu $K --chunk $CH > gpurun_out/exp5/n${N}_k${K}_c${CH}.json 2> gpurun_out/exp5/err.txt || { tail -5 gpurun_out/exp5/err.txt; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/exp5/n${N}_k${K}_c${CH}.json")); r=d["roofline"]
print("N=$N K=$K chunk=$CH value %.4e ms/pass %.1f kernel-only %.4e avg launch %.2f ms"%(d["value"],d["ms_per_step"],r["step_kernel_only_value"],r["avg_launch_ms"]))
PY
done
