#!/bin/bash
# forcing window + step kernel (what rs_hip_step callers get): one point per lane against two wavefronts per
# 64 points, natural order and FULL/LEAN, by launch size
mkdir -p gpurun_out
OUT=gpurun_out/r4_duo_windowed.txt
: > $OUT
B="--no-extra-legs --no-natural-leg --no-cpu-baseline --steps 5 --warmup 2 --cluster 0"
for N in 1000000 250000; do for F in "" "--full"; do for V in 1 4 3; do
  [ "$F" = "" ] && [ $V = 4 ] && continue
  v=$(python bench.py --total-points $N --variant $V --plans-per-gpu 4 --chunk 120 $F $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g launch %.3f ms'%(d['value'], d['roofline']['avg_launch_ms']))")
  echo "points $N ${F:-lean} variant $V: $v" | tee -a $OUT
done; done; done
