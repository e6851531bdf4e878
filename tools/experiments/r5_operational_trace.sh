#!/usr/bin/env bash
# kernel trace of the operational shape (401 stations x 8 881, coupling): where a 300 ms call goes
export TMPDIR=/tmp
OUT=gpurun_out/r5_op_trace; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_operational.py files 2 > $OUT/log.txt 2> $OUT/err.txt || { tail $OUT/err.txt; exit 1; }
cat $OUT/log.txt | tail -1
STATS=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cut -c1-160 $STATS | head -25
python3 - "$OUT" <<'PY'
import csv, glob, sys
rows=[]
for fn in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((r["Kernel_Name"].split("(")[0][-70:], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort(key=lambda r: r[1])
# last call: after the last gap > 50 ms
gaps=[i for i in range(1,len(rows)) if rows[i][1]-rows[i-1][2] > 50e6]
rows=rows[gaps[-1]:] if gaps else rows
t0=rows[0][1]; t1=rows[-1][2]
busy=sum(b-a for _,a,b in rows)
print(f"last call: {len(rows)} dispatches, span {(t1-t0)/1e6:.1f} ms, sum of kernel times {busy/1e6:.1f} ms")
step=[(n,a,b) for n,a,b in rows if "step_kernel" in n]
print(f"step kernels {len(step)}: sum {sum(b-a for _,a,b in step)/1e6:.1f} ms")
for n,a,b in step[:6]: print(f"  {n[-60:]} {(b-a)/1e3:.0f} us at {(a-t0)/1e6:.2f} ms")
# gaps between consecutive kernels
g=sorted(((rows[i][1]-rows[i-1][2])/1e3, rows[i-1][0][-40:], rows[i][0][-40:]) for i in range(1,len(rows)))
print("largest gaps (us):", [(round(x), p, q) for x,p,q in g[-8:]])
print("sum of gaps", sum(x for x,_,_ in g)/1e3, "ms")
PY
rm -rf $OUT/trace
