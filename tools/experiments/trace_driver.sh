# rocprofv3 kernel trace of the raw-series path + timeline statistics; usage: trace_driver.sh [mode] [points]
set -e
MODE=${1:-coupling}; N=${2:-1000000}
OUT=gpurun_out/traced_$MODE
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 tools/bench_driver_path.py $N 48 $MODE > $OUT/bench.txt 2> $OUT/err.txt || { tail -20 $OUT/err.txt; exit 1; }
grep "^rep" $OUT/bench.txt
python3 tools/trace_timeline.py $OUT/trace
python3 - <<PY
import csv, glob
rows=[]
for fn in glob.glob("$OUT/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((r.get("Direction", r.get("Name","?")), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
import collections
d=collections.defaultdict(lambda:[0,0])
for k,a,b in rows: d[k][0]+=1; d[k][1]+=b-a
for k,(n,t) in d.items(): print(f"memcpy {k:30s} n={n:5d} sum {t/1e6:9.2f} ms")
PY
rm -rf $OUT/trace
