#!/bin/bash
# issue priority by progress in the two-wavefront flavour: whole-job rate by shard size and unit
# (needs patches/r4_progress_priority.patch applied: it adds ROADSURF_HIP_DUO_PRIO_UNIT; result: slower, not kept)
mkdir -p gpurun_out
OUT=gpurun_out/r4_prio_sweep.txt
: > $OUT
B="--no-extra-legs --no-natural-leg --no-cpu-baseline --steps 10 --warmup 2"
for N in ${SIZES:-125000 250000 1000000}; do
  for U in ${UNITS:-0 1 2 4 0}; do
    v=$(ROADSURF_HIP_DUO_PRIO_UNIT=$U python bench.py --total-points $N $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g launch %.3f ms'%(d['value'], d['roofline']['avg_launch_ms']))")
    echo "points $N unit $U: $v" | tee -a $OUT
  done
done
