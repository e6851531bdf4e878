# A/B of library builds on the same box: bench.py with ROADSURF_HIP_LIB pointing at each (LIBS="hip hip_head")
set -e
mkdir -p gpurun_out/exp_ab
for rep in 1 2 3; do for lib in ${LIBS:-hip hip_head}; do
ROADSURF_HIP_LIB=$PWD/roadsurf_amd/lib/libroadsurf_$lib.so python bench.py --no-cpu-baseline --no-natural-leg --steps 2 $EXTRA > gpurun_out/exp_ab/$lib.json 2> gpurun_out/exp_ab/$lib.err || { tail -5 gpurun_out/exp_ab/$lib.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/exp_ab/$lib.json"))
print("%-10s value %.4e ms/pass %.1f kernel-only %.4e avg launch %.2f ms failed %s"%("$lib",d["value"],d["ms_per_step"],d["roofline"]["step_kernel_only_value"],d["roofline"]["avg_launch_ms"],d["config"]["failed_points"]))
PY
done; done
