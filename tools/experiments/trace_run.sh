# rocprofv3 kernel trace of one bench configuration + timeline statistics
set -e
TAG=$1; shift
OUT=gpurun_out/trace_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-natural-leg "$@" > $OUT/bench.json 2> $OUT/err.txt || { tail -20 $OUT/err.txt; exit 1; }
python3 tools/trace_timeline.py $OUT/trace
