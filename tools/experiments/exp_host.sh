# host-array path (runsimulation_batch): point-tile x time-chunk sweep; usage: exp_host.sh [n] [P:TC ...]
set -e
N=${1:-16384}; shift || true
[ $# -gt 0 ] || set -- 16384:256 4096:1024 2048:2048 1024:5761 4096:256 8192:720
for cfg in "$@"; do
IFS=: read P TC <<< "$cfg"
echo "P=$P TC=$TC: $(ROADSURF_HIP_TILE_POINTS=$P ROADSURF_HIP_CHUNK_STEPS=$TC python tools/bench_host_path.py $N 2>&1 | grep summary | tail -1)"
done
