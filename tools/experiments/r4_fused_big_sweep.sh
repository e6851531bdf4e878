#!/bin/bash
# 1 M and 500 000 points with the knot-reading two-wavefront flavour: plans x launch length
mkdir -p gpurun_out
OUT=gpurun_out/r4_fused_big_sweep.txt
: > $OUT
B="--no-extra-legs --no-natural-leg --no-cpu-baseline --steps 10 --warmup 2"
run() { # points plans chunk
  v=$(python bench.py --total-points $1 --plans-per-gpu $2 --chunk $3 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g launch %.3f ms in flight %.2f'%(d['value'], d['roofline']['avg_launch_ms'], d['roofline']['concurrent_launches']))")
  echo "points $1 plans $2 chunk $3: $v" | tee -a $OUT
}
for KC in "2 60" "2 48" "2 90" "3 60" "3 90" "4 120" "2 60"; do run 1000000 $KC; done
for KC in "3 90" "2 60" "3 60" "4 90" "2 90"; do run 500000 $KC; done
