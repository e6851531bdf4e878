#!/bin/bash
# driver-path rows: parity tests + rates at 1 M points (tiled inputs); usage: check_driver_path.sh tag [modes...]
TAG=${1:-drv}; shift
MODES=${@:-relax coupling skyview skycoupling}
mkdir -p gpurun_out/r4_$TAG
python -m pytest tests/test_hip_skyview.py tests/test_hip_driver.py tests/test_hip_coupling.py -m gpu -x -q > gpurun_out/r4_$TAG/tests.log 2>&1
rc=$?
tail -3 gpurun_out/r4_$TAG/tests.log
[ $rc -ne 0 ] && { tail -40 gpurun_out/r4_$TAG/tests.log; exit $rc; }
for m in $MODES; do
  N=1000000; [ $m = skycoupling ] && N=262144
  BENCH_UNIQUE=65536 BENCH_REPS=3 python3 tools/bench_driver_path.py $N 48 $m > gpurun_out/r4_$TAG/$m.txt 2>&1
  grep best gpurun_out/r4_$TAG/$m.txt | sed "s/^/$m $N: /"
done
