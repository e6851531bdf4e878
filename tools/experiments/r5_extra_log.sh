#!/usr/bin/env bash
# The forecast key's extra-passes field: sum over the previews saturating at 7 (0) against the longest
# preview's trip count in classes (ROADSURF_HIP_EXTRA_LOG=1) - wave statistics and the rate.
set -e
B="--no-cpu-baseline --no-natural-leg --no-extra-legs --steps 6 --warmup 2"
for X in 3 4 3 4; do
  export ROADSURF_HIP_EXTRA_LOG=$X
  echo "== EXTRA_LOG=$X"
  timeout -k 10 200 python3 bench.py $B | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lean', d['value'])"
  timeout -k 10 200 python3 bench.py $B --full | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full', d['value'])"
done
for X in 4; do
  export ROADSURF_HIP_EXTRA_LOG=$X
  echo "== EXTRA_LOG=$X"
  timeout -k 10 170 python3 tools/wave_stats.py bench 250000 2>&1 | grep -E "passes|wave-steps by"
  timeout -k 10 170 python3 tools/wave_stats.py driver-relax 250000 2>&1 | grep -E "passes|wave-steps by"
done
