#!/usr/bin/env bash
# bench.py's workload: previews of the forecast key AT the next window's first / middle / last index (between two
# knots, RsPreview::tair_b; default) against previews at the knots around the window (round 4,
# ROADSURF_HIP_PREVIEWS_AT_KNOTS=1), same box
B="--no-cpu-baseline --no-natural-leg --no-extra-legs --steps 6 --warmup 2"
for X in 1 0 1 0; do
  export ROADSURF_HIP_PREVIEWS_AT_KNOTS=$X
  echo "== previews at the knots: $X"
  timeout -k 10 200 python3 bench.py $B | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lean', d['value'])"
  timeout -k 10 200 python3 bench.py $B --full | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full', d['value'])"
  for N in 500000 250000 125000; do
    timeout -k 10 200 python3 bench.py $B --total-points $N | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('points $N', d['value'])"
  done
done
for X in 1 0; do
  export ROADSURF_HIP_PREVIEWS_AT_KNOTS=$X
  echo "== previews at the knots: $X"
  timeout -k 10 170 python3 tools/wave_stats.py bench 250000 2>&1 | grep -E "passes"
done
