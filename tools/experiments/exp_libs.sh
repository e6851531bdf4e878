# A/B of library builds on the same box: bench.py with ROADSURF_HIP_LIB pointing at each
set -e
mkdir -p gpurun_out/exp3
for rep in 1 2; do for lib in hip hip_fx hip_ng; do
ROADSURF_HIP_LIB=$PWD/roadsurf_amd/lib/libroadsurf_$lib.so python bench.py --no-cpu-baseline --no-natural-leg --steps 2 --forecast-mode 3 $EXTRA > gpurun_out/exp3/$lib.json 2> gpurun_out/exp3/$lib.err || { tail -5 gpurun_out/exp3/$lib.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/exp3/$lib.json"))
print("%-10s value %.4e ms/pass %.1f kernel-only %.4e avg launch %.2f ms"%("$lib",d["value"],d["ms_per_step"],d["roofline"]["step_kernel_only_value"],d["roofline"]["avg_launch_ms"]))
PY
done; done
