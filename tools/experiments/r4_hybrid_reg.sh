#!/bin/bash
# hybrid profile: how many of the 15 layers live in registers (the rest in LDS columns): 5 / 7 (default) / 9
mkdir -p gpurun_out
OUT=gpurun_out/r4_hybrid_reg.txt
: > $OUT
for rep in 1 2; do for L in h5 default h9; do
  LIBP=$PWD/roadsurf_amd/lib/libroadsurf_hip_$L.so; [ $L = default ] && LIBP=$PWD/roadsurf_amd/lib/libroadsurf_hip.so
  for m in relax skyview; do
    v=$(ROADSURF_HIP_LIB=$LIBP BENCH_UNIQUE=65536 BENCH_REPS=3 python3 tools/bench_driver_path.py 1000000 48 $m 2>&1 | grep best)
    echo "lib $L $m: $v" | tee -a $OUT
  done
done; done
