#!/usr/bin/env bash
# round 6: the GPU suite under rocprofv3 --kernel-trace with per-test windows -> what tools/kernel_reachability.py reads
export TMPDIR=/tmp
OUT=gpurun_out/r6_reach
rm -rf $OUT; mkdir -p $OUT
RS_TEST_WINDOWS=$PWD/$OUT/windows.csv timeout -k 10 1100 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 -m pytest tests -m gpu -q -x > $OUT/suite.txt 2>&1
grep -E "passed|failed" $OUT/suite.txt
# name + start time of every dispatch, all processes of the run
python3 - <<PY
import csv, glob, gzip
n = 0
with gzip.open("$OUT/dispatches.csv.gz", "wt") as g:
    w = csv.writer(g)
    w.writerow(["Kernel_Name", "Start_Timestamp"])
    for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            w.writerow([r["Kernel_Name"], r["Start_Timestamp"]])
            n += 1
print(n, "dispatches")
PY
rm -rf $OUT/trace
ls -la $OUT
