# sweep of the plan-order sort key on one plan (kernel-only figures are clean with K=1)
set -e
mkdir -p gpurun_out/exp2
run() { tag=$1; shift
python bench.py --no-cpu-baseline --no-natural-leg --steps 2 "$@" > gpurun_out/exp2/$tag.json 2> gpurun_out/exp2/$tag.err || { tail -5 gpurun_out/exp2/$tag.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/exp2/$tag.json"))
print("%-28s value %.4e ms/pass %.1f kernel-only %.4e avg launch %.2f ms"%("$tag",d["value"],d["ms_per_step"],d["roofline"]["step_kernel_only_value"],d["roofline"]["avg_launch_ms"]))
PY
}
run hist_c240 --sort-key history --chunk 240
for CH in ${CHUNKS:-240 120}; do for M in ${MODES:-1234 1324 3124 1243 124 134 14 314 4 34}; do
run fc_c${CH}_m${M} --chunk $CH --forecast-mode $M
done; done
