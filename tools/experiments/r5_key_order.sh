#!/usr/bin/env bash
# field order of the forecast key after round 5's changes (classes, precipitation bit, previews inside the window)
B="--no-cpu-baseline --no-natural-leg --no-extra-legs --steps 6 --warmup 2"
for M in 378059 307859 370859 378509 38059 37059 378059 307859; do
  timeout -k 10 200 python3 bench.py $B --forecast-mode $M | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lean mode $M', d['value'])"
done
