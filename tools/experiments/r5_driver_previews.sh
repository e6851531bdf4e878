for P in 3 2 3 2; do for M in relax coupling; do
  ROADSURF_HIP_DRIVER_PREVIEWS=$P BENCH_REPS=3 timeout -k 10 200 python3 tools/bench_driver_path.py 1000000 48 $M 2>&1 | grep best | sed "s/^/previews $P $M: /"
done; done
