#!/usr/bin/env bash
# the forecast key's precipitation bit (RsPreview::prec), A/B on one box: headline, FULL, 250 000 and 125 000 points
B="--steps 6 --warmup 2 --no-cpu-baseline --no-natural-leg --no-extra-legs"
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e' % d['value'])"; }
for rep in 1 2; do
  for args in "" "--full" "--total-points 250000" "--total-points 125000"; do
    a=$(ROADSURF_HIP_PRECIP_BIT=1 python3 bench.py $B $args 2>/dev/null | val)
    b=$(ROADSURF_HIP_PRECIP_BIT=0 python3 bench.py $B $args 2>/dev/null | val)
    echo "bench.py $args: with the bit $a, without $b"
  done
done
