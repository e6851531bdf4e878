#!/bin/bash
# FULL feature set: the hybrid flavour (layers 8-15 in LDS, one point per lane) against two wavefronts per 64 points
mkdir -p gpurun_out
OUT=gpurun_out/r4_full_duo.txt
: > $OUT
B="--full --no-extra-legs --no-natural-leg --no-cpu-baseline --steps 5 --warmup 2"
run() { # points variant plans chunk
  v=$(python bench.py --total-points $1 --variant $2 --plans-per-gpu $3 --chunk $4 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g launch %.3f ms in flight %.2f'%(d['value'], d['roofline']['avg_launch_ms'], d['roofline']['concurrent_launches']))")
  echo "points $1 variant $2 plans $3 chunk $4: $v" | tee -a $OUT
}
run 1000000 0 3 240
run 1000000 3 3 240
run 1000000 3 3 120
run 1000000 3 3 90
run 1000000 3 4 120
run 1000000 3 2 120
run 250000 0 4 240
run 250000 3 4 240
run 125000 0 2 240
run 125000 3 2 240
