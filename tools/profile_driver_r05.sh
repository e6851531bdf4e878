#!/usr/bin/env bash
# Kernel trace of the driver data path (rs_driver_run, 1 M points x 48 h, default fan-out) for profiles/:
#   profiles/r05_driver_path_<mode>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the run
#   profiles/r05_driver_path_<mode>_timeline.txt       per-kernel sums/unions of the LAST call, idle time
# usage: profile_driver_r05.sh mode [points]      (run on the GPU box; python3 directly behind `--`)
set -e
MODE=${1:-relax}; N=${2:-1000000}
export TMPDIR=/tmp
OUT=gpurun_out/r5_prof_drv_$MODE
PROF=gpurun_out/profiles_r05   # (gpurun merges gpurun_out/ back; copy to profiles/ afterwards)
rm -rf $OUT; mkdir -p $OUT $PROF
export BENCH_REPS=2 BENCH_PAUSE_S=0.25  # (distinct series for every point, as bench.py times them since round 5)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_driver_path.py $N 48 $MODE > $OUT/bench.log 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
grep -E "rep |best" $OUT/bench.log
STATS=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
SHA=$(python3 -c "from roadsurf_amd import provenance; print(provenance.csrc_sha16())")
{ echo "# rocprofv3 --kernel-trace --stats -- python3 tools/bench_driver_path.py $N 48 $MODE (distinct series, 3 calls: 1 warm + 2 timed); kernel sources $SHA"; cat "$STATS"; } > $PROF/r05_driver_path_${MODE}_kernel_stats.csv
python3 - "$OUT" "$MODE" "$N" "$SHA" > $PROF/r05_driver_path_${MODE}_timeline.txt <<'PY'
import csv, glob, collections, sys
out, mode, n, sha = sys.argv[1:5]
rows=[]
for fn in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((r["Kernel_Name"].replace("void ","").replace("(anonymous namespace)::","").split("(")[0], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Grid_Size",0) or 0), r.get("Queue_Id","?")))
rows.sort(key=lambda r:r[1])
gaps=[i for i in range(1,len(rows)) if rows[i][1]-max(r[2] for r in rows[max(0,i-50):i])>100e6]
start=gaps[-1] if gaps else 0
rows=rows[start:]
t0=rows[0][1]; t1=max(r[2] for r in rows)
import os
blocks = os.environ.get("ROADSURF_HIP_PLANS_PER_DEVICE", "4")
print(f"# rs_driver_run, mode {mode}, {n} points x 48 h, default fan-out ({blocks} blocks on one GPU, 4 hardware queues); kernel sources {sha}")
print(f"last call: {len(rows)} dispatches over {(t1-t0)/1e6:.1f} ms of kernel span")
fam=collections.defaultdict(list)
for nme,a,b,g,q in rows: fam[nme.split("::")[-1][:56]].append((a,b,g))
def union(iv):
    tot=0; end=-1
    for a,b in sorted(iv):
        if b>end: tot+=b-max(a,end); end=b
    return tot
for k,iv in sorted(fam.items(), key=lambda kv:-sum(b-a for a,b,g in kv[1]))[:22]:
    print(f"{k:58s} n={len(iv):5d} sum {sum(b-a for a,b,g in iv)/1e6:8.2f} ms union {union([(a,b) for a,b,g in iv])/1e6:8.2f} ms")
allk=[(a,b) for nme,a,b,g,q in rows]
print(f"any kernel running {union(allk)/1e6:.1f} ms, idle {(t1-t0-union(allk))/1e6:.1f} ms")
step=[(a,b) for nme,a,b,g,q in rows if "step_kernel" in nme]
print(f"step kernels: n={len(step)} union {union(step)/1e6:.1f} ms; first starts {(min(a for a,b in step)-t0)/1e6:.1f} ms into the span, last ends {(t1-max(b for a,b in step))/1e6:.1f} ms before its end")
small=[(a,b) for nme,a,b,g,q in rows if "step_kernel" not in nme]
print(f"other kernels: n={len(small)} sum {sum(b-a for a,b in small)/1e6:.1f} ms union {union(small)/1e6:.1f} ms")
byq=collections.defaultdict(list)
for nme,a,b,g,q in rows: byq[q].append((nme,a,b))
print("per hardware queue (one block of the fan-out each): first kernel / first step kernel / last step kernel end / last kernel end, ms into the span")
for q,v in sorted(byq.items(), key=lambda kv: min(a for n_,a,b in kv[1])):
    st=[(a,b) for n_,a,b in v if "step_kernel" in n_]
    if not st: continue
    print(f"  queue {q}: {(min(a for n_,a,b in v)-t0)/1e6:7.1f} {(min(a for a,b in st)-t0)/1e6:7.1f} {(max(b for a,b in st)-t0)/1e6:7.1f} {(max(b for n_,a,b in v)-t0)/1e6:7.1f}   step kernels {len(st)}, sum {sum(b-a for a,b in st)/1e6:.1f} ms")
PY
cat $PROF/r05_driver_path_${MODE}_timeline.txt
rm -rf $OUT/trace
