#!/usr/bin/env bash
# round 6: three libraries side by side, alternating, at the three sizes:  r6_ab3.sh <suffix a> <suffix b> [reps]
A=$1; B=$2; REPS=${3:-2}
for pts in 1000000 250000 125000; do
  for rep in $(seq $REPS); do
    for lib in "" $A $B; do
      ROADSURF_HIP_LIB=$PWD/roadsurf_amd/lib/libroadsurf_hip$lib.so python3 bench.py --total-points $pts --steps 6 --warmup 2 --no-cpu-baseline --no-natural-leg --no-extra-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8d points lib%-8s %.4e  avg launch %.3f ms'%($pts, '$lib' or '(head)', d['value'], d['roofline']['avg_launch_ms']))"
    done
  done
done
