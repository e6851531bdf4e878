#!/usr/bin/env bash
# 125 000 points (config 4's per-GPU shard): which fields of the forecast key pay on an underfilled device, where
# a launch is as long as its slowest wavefront (the extra-passes field GATHERS the slow band's points)
B="--no-cpu-baseline --no-natural-leg --no-extra-legs --steps 10 --warmup 2 --total-points ${1:-125000}"
for M in 378059 37859 3759 359 59 9 378059 37859 3759; do
  timeout -k 10 200 python3 bench.py $B --forecast-mode $M | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mode $M', d['value'])"
done
