#!/usr/bin/env bash
# round 6: the driver path's step kernel against the FULL leg's, launch shape for launch shape: one plan / one block of
# 1 M points, launches of 120 indices, nothing beside them on the device.
OUT=gpurun_out/r6_driver_vs_full
mkdir -p $OUT
B="--full --no-natural-leg --no-extra-legs --no-cpu-baseline --chunk 120"
for K in 1 3; do
  python3 bench.py $B --plans-per-gpu $K > $OUT/full_K$K.json 2> $OUT/full_K$K.err || exit 1
  python3 -c "
import json
l=json.loads(open('$OUT/full_K$K.json').read().strip().splitlines()[-1])
r=l['roofline']
print('bench --full, $K plan(s) x 120 indices: value %.3e, step kernels only %.3e, launches %d, avg launch %.3f ms, concurrent %.2f, clock %s MHz'%(l['value'], r['step_kernel_only_value'], r.get('launches',0), r.get('avg_launch_ms',0), r.get('concurrent_launches',0), l.get('sclk_mhz')))
"
done
export BENCH_REPS=3
for K in 1 4; do
  ROADSURF_HIP_PLANS_PER_DEVICE=$K python3 tools/bench_driver_path.py 1000000 48 relax 2> $OUT/drv_K$K.err | grep -E "best" | sed "s/^/rs_driver_run relax, $K block(s): /"
done
