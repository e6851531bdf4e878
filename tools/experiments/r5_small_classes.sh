#!/usr/bin/env bash
# config 4's per-GPU shards on one GPU with the class key against round 4's field (ROADSURF_HIP_EXTRA_CLASSES=0)
B="--no-cpu-baseline --no-natural-leg --no-extra-legs --steps 10 --warmup 2"
for X in 0 1 0 1; do
  export ROADSURF_HIP_EXTRA_CLASSES=$X
  for N in 125000 250000 500000; do
    timeout -k 10 200 python3 bench.py $B --total-points $N | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('classes $X points $N', d['value'])"
  done
done
