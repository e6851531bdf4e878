#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs per kernel: mean counter value per dispatch."""
import csv, glob, os, sys, collections
out = sys.argv[1]
for sub in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(sub):
        continue
    files = glob.glob(os.path.join(sub, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in files:
        with open(fn) as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name", "?").replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(f"== {os.path.basename(sub)}")
    for k, cs in acc.items():
        for c, v in cs.items():
            print(f"{k[:70]:70s} {c:22s} n={len(v):4d} mean={sum(v)/len(v):.6g} sum={sum(v):.6g}")
