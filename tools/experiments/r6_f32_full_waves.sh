#!/usr/bin/env bash
# round 6: the FULL instances of step_kernel_f32duo at five (96 registers, 119 spilled) or four (128, 42 spilled)
# wavefronts per SIMD.  Build first: make -C roadsurf_amd OBJ=build_fw4 LIB=lib/libroadsurf_hip_fw4.so EXTRA=-DRS_X2D_FULL_WAVES=4
B="--f32 --full --points 1250000 --hours 168 --no-natural-leg --no-extra-legs --no-cpu-baseline --plans-per-gpu 2 --chunk 360"
for L in "" _fw4; do
  [ -f roadsurf_amd/lib/libroadsurf_hip$L.so ] || continue
  for rep in 1 2; do
    ROADSURF_HIP_LIB=$PWD/roadsurf_amd/lib/libroadsurf_hip$L.so python3 bench.py $B 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FULL fp32, lib \"$L\": %.3e (kernels only %.3e)'%(l['value'], l['roofline']['step_kernel_only_value']))"
  done
done
