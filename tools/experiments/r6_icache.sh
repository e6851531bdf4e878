#!/usr/bin/env bash
# round 6: does the raw-series step kernel (64-92 KB of code) miss the instruction cache more than the knot-reading
# one (FULL bench leg)?  Instruction-cache counters of both, per launch.
OUT=gpurun_out/r6_icache
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 -L > $OUT/avail_all.txt 2>&1
grep -i -E "icache|ifetch|inst_cache|SQC_" $OUT/avail_all.txt | cut -c1-200 | sort -u > $OUT/avail.txt
C1="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES"
C2="SQ_IFETCH SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU"
export BENCH_REPS=1
for pass in 1 2; do
  eval C=\$C$pass
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/drv$pass -- python3 tools/bench_driver_path.py 1000000 48 relax > $OUT/drv$pass.log 2> $OUT/drv$pass.err || { tail -5 $OUT/drv$pass.err; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/full$pass -- python3 bench.py --full --steps 1 --warmup 0 --no-natural-leg --no-extra-legs --no-cpu-baseline > $OUT/full$pass.log 2> $OUT/full$pass.err || { tail -5 $OUT/full$pass.err; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections
for tag in ("drv", "full"):
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for p in (1, 2):
        for f in glob.glob(f"gpurun_out/r6_icache/{tag}{p}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "step_kernel_duo" not in r["Kernel_Name"]: continue
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(tag, {k: (f"{v:.4g}", n[k]) for k, v in sorted(tot.items())})
    if tot.get("SQC_ICACHE_REQ"):
        print("   miss rate %.4f; misses per VALU instruction %.5f" % (tot["SQC_ICACHE_MISSES"] / tot["SQC_ICACHE_REQ"], tot["SQC_ICACHE_MISSES"] / max(tot["SQ_INSTS_VALU"], 1)))
PY
rm -rf $OUT/drv? $OUT/full?
