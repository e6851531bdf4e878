#!/usr/bin/env bash
# round 6: instructions per 64 point-steps of the raw-series step kernel on its own workload and on bench.py's weather
OUT=gpurun_out/r6_driver_weather_pmc
mkdir -p $OUT
export TMPDIR=/tmp BENCH_REPS=1
for W in driver bench; do
  BENCH_WEATHER=$W timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $OUT/$W -- python3 tools/bench_driver_path.py 1000000 48 relax > $OUT/$W.log 2> $OUT/$W.err || { tail -5 $OUT/$W.err; exit 1; }
  python3 - $W <<'PY'
import csv, glob, collections, sys
w = sys.argv[1]
tot = collections.defaultdict(float); n = 0
for f in glob.glob(f"gpurun_out/r6_driver_weather_pmc/{w}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "step_kernel_duo" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
calls = 2  # 1 warm + 1 timed
ws = calls * 1000000 * 5761 / 64
print(f"weather {w}: per 64 point-steps " + "  ".join(f"{k[3:]} {v / ws:.0f}" for k, v in sorted(tot.items()) if k != "SQ_WAVES"))
PY
  rm -rf $OUT/$W
done
