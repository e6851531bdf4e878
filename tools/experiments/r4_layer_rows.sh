#!/bin/bash
# per-layer constants side by side (default) against one table per constant (library build -DRS_NO_LAYER_ROWS)
mkdir -p gpurun_out
OUT=gpurun_out/r4_layer_rows.txt
: > $OUT
B="--no-extra-legs --no-natural-leg --no-cpu-baseline --steps 10 --warmup 2"
for rep in 1 2 3; do for L in nl default; do
  LIBP=$PWD/roadsurf_amd/lib/libroadsurf_hip_$L.so; [ $L = default ] && LIBP=$PWD/roadsurf_amd/lib/libroadsurf_hip.so
  for N in 1000000 125000; do
    v=$(ROADSURF_HIP_LIB=$LIBP python bench.py --total-points $N $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g launch %.3f ms'%(d['value'], d['roofline']['avg_launch_ms']))")
    echo "lib $L points $N: $v" | tee -a $OUT
  done
done; done
