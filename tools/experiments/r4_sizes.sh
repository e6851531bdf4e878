#!/bin/bash
# bench.py's automatic configuration at the four shard sizes of the strong-scaling table
mkdir -p gpurun_out
OUT=gpurun_out/r4_sizes_${1:-x}.txt
: > $OUT
B="--no-extra-legs --no-natural-leg --no-cpu-baseline --steps 10 --warmup 2"
for N in ${SIZES:-125000 250000 500000 1000000}; do
  v=$(python bench.py --total-points $N $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g plans %d chunk %d launch %.3f ms in flight %.2f'%(d['value'], d['config']['plans_per_gpu'], d['config']['chunk_steps'], d['roofline']['avg_launch_ms'], d['roofline']['concurrent_launches']))")
  echo "points $N: $v" | tee -a $OUT
done
