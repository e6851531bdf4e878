#!/bin/bash
# forecast key with and without the ground digit (field 9: frost depth), whole-job rate by shard size
mkdir -p gpurun_out
OUT=gpurun_out/r4_ground_digit.txt
: > $OUT
B="--no-extra-legs --no-natural-leg --no-cpu-baseline --steps 10 --warmup 2"
for N in ${SIZES:-1000000 125000}; do
  for M in ${MODES:-37865 378659 37865 378659}; do
    v=$(python bench.py --total-points $N --forecast-mode $M $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g launch %.3f ms in flight %.2f'%(d['value'], d['roofline']['avg_launch_ms'], d['roofline']['concurrent_launches']))")
    echo "points $N mode $M: $v" | tee -a $OUT
  done
done
