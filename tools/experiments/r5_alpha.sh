#!/usr/bin/env bash
# alpha of the forecast key's surface-temperature predictor with previews inside the window
B="--no-cpu-baseline --no-natural-leg --no-extra-legs --steps 6 --warmup 2"
for A in 0.5 0.3 0.7 0.9 0.5 0.7; do
  timeout -k 10 200 python3 bench.py $B --forecast-alpha $A | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lean alpha $A', d['value'])"
done
