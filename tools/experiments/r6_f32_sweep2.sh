#!/usr/bin/env bash
# round 6, last pass: plans x launch length of the fp32 flavour at config 5's shape again, with the storages for a point
# pair in the kernel (tools/experiments/r6_x2road_ab.sh), and four wavefronts per SIMD against five
# (make -C roadsurf_amd OBJ=build_w4 LIB=lib/libroadsurf_hip_w4.so EXTRA=-DRS_X2D_WAVES=4)
OUT=gpurun_out/r6_f32_sweep2
mkdir -p $OUT
B="--f32 --points 1250000 --hours 168 --no-natural-leg --no-extra-legs --no-cpu-baseline --steps 2 --warmup 1"
run() { # tag, lib suffix, flags
  export ROADSURF_HIP_LIB=$PWD/roadsurf_amd/lib/libroadsurf_hip$2.so
  python3 bench.py $B $3 > $OUT/$1.json 2>/dev/null
  python3 -c "import json;d=json.load(open('$OUT/$1.json'));r=d['roofline'];print('%-22s %.4e  ms/pass %.1f  avg launch %.2f ms  concurrent %.2f'%('$1',d['value'],d['ms_per_step'],r['avg_launch_ms'],r['concurrent_launches']))"
}
run p2_c360 "" "--plans-per-gpu 2 --chunk 360"
run p2_c480 "" "--plans-per-gpu 2 --chunk 480"
run p2_c240 "" "--plans-per-gpu 2 --chunk 240"
run p3_c360 "" "--plans-per-gpu 3 --chunk 360"
run p3_c480 "" "--plans-per-gpu 3 --chunk 480"
run p4_c480 "" "--plans-per-gpu 4 --chunk 480"
run p2_c720 "" "--plans-per-gpu 2 --chunk 720"
run w4_p2_c360 "_w4" "--plans-per-gpu 2 --chunk 360"
run w4_p3_c360 "_w4" "--plans-per-gpu 3 --chunk 360"
run p2_c360_again "" "--plans-per-gpu 2 --chunk 360"
