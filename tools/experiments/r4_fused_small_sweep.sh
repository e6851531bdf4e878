#!/bin/bash
# small shards with the knot-reading two-wavefront flavour: plans x launch length
mkdir -p gpurun_out
OUT=gpurun_out/r4_fused_small_sweep.txt
: > $OUT
B="--no-extra-legs --no-natural-leg --no-cpu-baseline --steps 10 --warmup 2"
run() { # points plans chunk
  v=$(python bench.py --total-points $1 --plans-per-gpu $2 --chunk $3 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g launch %.3f ms in flight %.2f'%(d['value'], d['roofline']['avg_launch_ms'], d['roofline']['concurrent_launches']))")
  echo "points $1 plans $2 chunk $3: $v" | tee -a $OUT
}
for K in 2 3 4; do for C in 240 360 480; do run 125000 $K $C; done; done
for K in 3 4 6; do for C in 120 180 240; do run 250000 $K $C; done; done
