#!/usr/bin/env bash
# the literal drop-in under the C++ harness (tools/dropin_harness.cpp; `make -C roadsurf_amd harness`): threads x coalescing mode
H="timeout -k 5 120 tools/bin/dropin_harness"
export ROADSURF_HIP_DEVICE=0
echo "default (automatic): $($H 1 48)"
echo "ROADSURF_HIP_COALESCE_US=0: $(ROADSURF_HIP_COALESCE_US=0 $H 16 384)"
for cfg in "16 1536" "64 3072" "256 6144"; do
  set -- $cfg
  echo "default (automatic): $($H $1 $2)"
  echo "ROADSURF_HIP_COALESCE_US=2000: $(ROADSURF_HIP_COALESCE_US=2000 $H $1 $2)"
done
