#!/usr/bin/env bash
# kernel timeline of the driver path with coupling (1 M points x 48 h, default fan-out), last repetition
set -e
export TMPDIR=/tmp
OUT=gpurun_out/r3_cpl_trace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/bench_driver_path.py 1000000 48 ${MODE:-coupling} > $OUT/bench.log 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
grep "rep " $OUT/bench.log
python3 - <<'PY'
import csv, glob, collections
rows=[]
for fn in glob.glob("gpurun_out/r3_cpl_trace/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((r["Kernel_Name"].replace("void ","").split("(")[0], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Grid_Size",0) or 0)))
rows.sort(key=lambda r:r[1])
# last repetition = last quarter of the step launches: find big gaps (> 50 ms) between dispatches
gaps=[i for i in range(1,len(rows)) if rows[i][1]-max(r[2] for r in rows[max(0,i-50):i])>30e6]
start=gaps[-1] if gaps else 0
rows=rows[start:]
t0=rows[0][1]; t1=max(r[2] for r in rows)
print(f"last repetition: {len(rows)} dispatches over {(t1-t0)/1e6:.1f} ms")
fam=collections.defaultdict(list)
for n,a,b,g in rows: fam[n.split("::")[-1][:48]].append((a,b,g))
def union(iv):
    tot=0; end=-1
    for a,b in sorted(iv):
        if b>end: tot+=b-max(a,end); end=b
    return tot
for k,iv in sorted(fam.items(), key=lambda kv:-sum(b-a for a,b,g in kv[1]))[:14]:
    print(f"{k:50s} n={len(iv):5d} sum {sum(b-a for a,b,g in iv)/1e6:8.2f} ms union {union([(a,b) for a,b,g in iv])/1e6:8.2f} ms")
allk=[(a,b) for n,a,b,g in rows]
print(f"any kernel running {union(allk)/1e6:.1f} ms, idle {(t1-t0-union(allk))/1e6:.1f} ms")
# occupancy-weighted: time with small grids only
small=[(a,b) for n,a,b,g in rows if "replay" in n and g<=64*1024]
big=[(a,b) for n,a,b,g in rows if not ("replay" in n and g<=64*1024)]
print(f"replay launches of <= 65536 threads: n={len(small)} union {union(small)/1e6:.1f} ms; everything else union {union(big)/1e6:.1f} ms; both {union(small+big)/1e6:.1f}")
# time where ONLY small replay launches run
ev=[]
for a,b in big: ev.append((a,1)); ev.append((b,-1))
ev.sort(); lvl=0; last=t0; busy_big=[]
cur=None
for t,d in ev:
    if lvl==0 and d==1: cur=t
    lvl+=d
    if lvl==0: busy_big.append((cur,t))
only_small=union(small+busy_big)-union(busy_big)
print(f"time with only small replay launches running: {only_small/1e6:.1f} ms")
PY
rm -rf $OUT/trace
