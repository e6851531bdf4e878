# step-kernel launch duration against the number of points (one plan): fixed cost + tail vs work
set -e
mkdir -p gpurun_out/exp4
for N in ${SIZES:-125000 250000 500000 1000000 2000000}; do
python bench.py --no-cpu-baseline --no-natural-leg --steps 2 --total-points $N $EXTRA > gpurun_out/exp4/n$N.json 2> gpurun_out/exp4/n$N.err || { tail -5 gpurun_out/exp4/n$N.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/exp4/n$N.json")); r=d["roofline"]
print("N=%8d value %.4e ms/pass %.1f kernel-only %.4e avg launch %.3f ms"%($N,d["value"],d["ms_per_step"],r["step_kernel_only_value"],r["avg_launch_ms"]))
PY
done
