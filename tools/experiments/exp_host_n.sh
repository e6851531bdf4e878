# host-array path at the library's defaults for several batch sizes
for N in "$@"; do echo "n=$N: $(REPS=${REPS:-5} python tools/bench_host_path.py $N 2>&1 | grep summary | tail -1)"; done
