#!/usr/bin/env bash
set -e
OUT=gpurun_out/r3_cpl
mkdir -p $OUT
for M in 4 3; do
ROADSURF_HIP_CPL_PROFILE=$M timeout -k 10 900 python -m pytest tests/test_hip_coupling.py tests/test_hip_driver.py tests/test_hip_random_configs.py -x -q -m gpu -k "coupl or random_conf or chunked" > $OUT/tests$M.log 2>&1 || { tail -30 $OUT/tests$M.log; exit 1; }
tail -1 $OUT/tests$M.log
done
for M in 0 3 4; do
  echo "CPL_PROFILE=$M"
  ROADSURF_HIP_CPL_PROFILE=$M python tools/bench_driver_path.py 1000000 48 coupling 2>&1 | grep "rep [123]"
done
ROADSURF_HIP_DRIVER_TIMING=1 ROADSURF_HIP_PLANS_PER_DEVICE=1 python tools/bench_driver_path.py 1000000 48 coupling 2>&1 | grep -v "amdgpu.ids" | tail -45
