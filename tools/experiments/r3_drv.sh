#!/usr/bin/env bash
# driver path (rs_driver_run, relaxation on): blocks per device x chunk x FULL flavour
mkdir -p gpurun_out/r3_drv
for V in 0 4; do for P in 2 3 4 6; do for C in 256 512; do
  echo "VARIANT=$V PLANS_PER_DEVICE=$P CHUNK=$C"
  ROADSURF_HIP_VARIANT=$V ROADSURF_HIP_PLANS_PER_DEVICE=$P ROADSURF_HIP_CHUNK_STEPS=$C python tools/bench_driver_path.py 1000000 48 relax 2>&1 | grep "rep [23]"
done; done; done
