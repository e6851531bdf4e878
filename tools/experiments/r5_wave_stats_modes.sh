for X in 0 3 4; do export ROADSURF_HIP_EXTRA_LOG=$X; echo "== $X"; timeout -k 10 170 python3 tools/wave_stats.py bench 250000 2>&1 | grep -E "passes|by how"; done
