# rs_driver_run: window length sweep; usage: exp_driver_chunk.sh mode chunk...
MODE=$1; shift
for TC in "$@"; do
echo "$MODE chunk $TC: $(ROADSURF_HIP_CHUNK_STEPS=$TC python tools/bench_driver_path.py 1000000 48 $MODE 2>&1 | grep '^rep' | tail -2 | awk '{print $6, $9}' | tr '\n' ' ')"
done
