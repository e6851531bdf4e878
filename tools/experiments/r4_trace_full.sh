#!/bin/bash
# kernel timeline of bench.py --full (1 M points, knot-reading two-wavefront flavour): the chain between two step launches of a plan
export TMPDIR=/tmp
OUT=gpurun_out/r4_trace_full
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py ${FULLFLAG---full} --steps 1 --warmup 1 --no-cpu-baseline --no-natural-leg --no-extra-legs > $OUT/bench.json 2> $OUT/err.txt || { tail -20 $OUT/err.txt; exit 1; }
python3 - <<'PY'
import csv, glob, collections
rows=[]
for fn in glob.glob("gpurun_out/r4_trace_full/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((r["Kernel_Name"].replace("void ","").replace("(anonymous namespace)::","").split("(")[0].split("::")[-1][:28], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id","?")))
rows.sort(key=lambda r:r[1])
rows=rows[int(len(rows)*0.6):]
byq=collections.defaultdict(list)
for n,a,b,q in rows: byq[q].append((n,a,b))
for q,v in byq.items():
    steps=[i for i,(n,a,b) in enumerate(v) if n.startswith("step_kernel")]
    if len(steps)<6: continue
    print("queue",q,"dispatches",len(v))
    for i,j in zip(steps[2:5],steps[3:6]):
        end=v[i][2]; start=v[j][1]
        seq=[(n,(a-end)/1e3,(b-a)/1e3) for n,a,b in v[i+1:j]]
        print("  step %.0f us | gap %.0f us:"%((v[i][2]-v[i][1])/1e3,(start-end)/1e3), " ".join("%s@%.0f+%.0f"%s for s in seq))
PY
rm -rf $OUT/trace
