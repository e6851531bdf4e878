export TMPDIR=/tmp
OUT=gpurun_out/prof_full250k; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --full --variant 3 --total-points 250000 --steps 2 --warmup 1 --no-cpu-baseline --no-natural-leg --no-extra-legs > $OUT/bench.json 2> $OUT/err.txt || { tail $OUT/err.txt; exit 1; }
python3 tools/trace_timeline.py $OUT/trace > $OUT/timeline.txt; head -24 $OUT/timeline.txt
rm -rf $OUT/trace
