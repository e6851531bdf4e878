#!/usr/bin/env bash
# road_condition: SnowStorage / IceStorage skipped for wavefronts without snow / ice (default) against the A/B
# build (make OBJ=build_noskip LIB=lib/libroadsurf_hip_noskip.so EXTRA=-DRS_NO_STORAGE_SKIP), same box
B="--no-cpu-baseline --no-natural-leg --no-extra-legs --steps 6 --warmup 2"
for L in noskip default noskip default; do
  if [ $L = noskip ]; then export ROADSURF_HIP_LIB=$PWD/roadsurf_amd/lib/libroadsurf_hip_noskip.so; else unset ROADSURF_HIP_LIB; fi
  echo "== $L"
  timeout -k 10 200 python3 bench.py $B | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lean', d['value'])"
  timeout -k 10 200 python3 bench.py $B --full | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full', d['value'])"
  for M in relax coupling; do
    BENCH_REPS=3 timeout -k 10 200 python3 tools/bench_driver_path.py 1000000 48 $M 2>&1 | grep best | sed "s/^/$M /"
  done
done
