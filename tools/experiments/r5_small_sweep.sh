#!/usr/bin/env bash
# 125 000 and 250 000 points: plans x launch length again, with round 5's key (previews inside the window for
# launches that are not whole hours)
B="--no-cpu-baseline --no-natural-leg --no-extra-legs --steps 10 --warmup 2"
for N in 125000 250000; do
for KC in "2 240" "2 120" "2 60" "3 120" "3 60" "4 120" "4 60" "2 240"; do
  set -- $KC
  timeout -k 10 200 python3 bench.py $B --total-points $N --plans-per-gpu $1 --chunk $2 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('points $N plans $1 x $2:', d['value'])"
done; done
