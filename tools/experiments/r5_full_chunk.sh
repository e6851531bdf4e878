#!/usr/bin/env bash
# FULL feature set: launch length with previews inside the window (60 / 90 were behind 120 in round 4)
B="--no-cpu-baseline --no-natural-leg --no-extra-legs --steps 6 --warmup 2 --full"
for C in 120 60 90 120 60; do
  timeout -k 10 200 python3 bench.py $B --chunk $C | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full chunk $C', d['value'])"
done
for C in 60 30 40 60 30; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-natural-leg --no-extra-legs --steps 6 --warmup 2 --chunk $C | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lean chunk $C', d['value'])"
done
