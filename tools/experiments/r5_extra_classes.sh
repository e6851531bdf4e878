#!/usr/bin/env bash
# Field 0 of the forecast key: classes of the longest expected boundary-layer loop (default) against round 4's
# saturating sum (ROADSURF_HIP_EXTRA_CLASSES=0): bench.py LEAN / FULL and the driver path, same box.
set -e
B="--no-cpu-baseline --no-natural-leg --no-extra-legs --steps 6 --warmup 2"
for X in 0 1 0 1; do
  export ROADSURF_HIP_EXTRA_CLASSES=$X
  echo "== EXTRA_CLASSES=$X"
  timeout -k 10 200 python3 bench.py $B | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lean', d['value'])"
  timeout -k 10 200 python3 bench.py $B --full | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full', d['value'])"
  for M in relax skyview coupling; do
    BENCH_REPS=3 timeout -k 10 200 python3 tools/bench_driver_path.py 1000000 48 $M 2>&1 | grep best | sed "s/^/$M /"
  done
done
