#!/bin/bash
# 125 000 points with the kernels at HEAD: launch length, plans, wave-table class bits
mkdir -p gpurun_out
OUT=gpurun_out/r4_small_final.txt
: > $OUT
B="--total-points 125000 --no-extra-legs --no-natural-leg --no-cpu-baseline --steps 10 --warmup 2"
run() { # label env args
  L=$1; E=$2; shift; shift
  v=$(env $E python bench.py $B "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g launch %.3f ms in flight %.2f'%(d['value'], d['roofline']['avg_launch_ms'], d['roofline']['concurrent_launches']))")
  echo "$L: $v" | tee -a $OUT
}
run "2x240" X=1
run "2x180" X=1 --chunk 180
run "2x300" X=1 --chunk 300
run "2x360" X=1 --chunk 360
run "3x240" X=1 --plans-per-gpu 3
run "2x240 class bits 4" ROADSURF_HIP_WAVE_CLASS_BITS=4
run "2x240 class bits 6" ROADSURF_HIP_WAVE_CLASS_BITS=6
run "2x240 class bits 0" ROADSURF_HIP_WAVE_CLASS_BITS=0
run "2x240 alpha 0.3" X=1 --forecast-alpha 0.3
run "2x240 alpha 0.7" X=1 --forecast-alpha 0.7
run "2x240" X=1
