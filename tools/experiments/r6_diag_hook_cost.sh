#!/usr/bin/env bash
# round 6: what the (not taken) diagnostics call costs the one-point-per-lane kernels with the profile in LDS
# (historical: the run-time hook and -DRS_NO_DIAG_HOOK existed for this measurement only; the library now has DIAG instances)
for full in "" "--full"; do
  for lib in "" _nodiag; do
    for rep in 1 2; do
      ROADSURF_HIP_LIB=$PWD/roadsurf_amd/lib/libroadsurf_hip$lib.so python3 bench.py --variant 2 $full --no-cpu-baseline --no-natural-leg --no-extra-legs 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('LDS profile ${full:-lean} lib \"$lib\": %.4e'%l['value'])"
    done
  done
done
