#!/usr/bin/env python3
"""Timeline statistics from a rocprofv3 --kernel-trace CSV: per kernel family the summed duration
and the union of its intervals; for the step kernels the time with 0, 1, 2, ... of them in flight."""
import csv, glob, sys, collections
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for fn in files:
    with open(fn) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
            rows.append((name.split("(")[0], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort(key=lambda r: r[1])
t0 = rows[0][1]; t1 = max(r[2] for r in rows)
print(f"{len(rows)} dispatches over {(t1 - t0) / 1e6:.1f} ms")
fam = collections.defaultdict(list)
for n, a, b in rows:
    key = "step" if "step_kernel" in n else n.split("::")[-1][:40]
    fam[key].append((a, b))
def union(iv):
    tot = 0; end = -1
    for a, b in sorted(iv):
        if b > end:
            tot += b - max(a, end); end = b
    return tot
for k, iv in sorted(fam.items(), key=lambda kv: -sum(b - a for a, b in kv[1])):
    print(f"{k:42s} n={len(iv):5d} sum {sum(b - a for a, b in iv) / 1e6:9.2f} ms  union {union(iv) / 1e6:9.2f} ms")
allk = [(a, b) for v in fam.values() for a, b in v]
print(f"any kernel running: {union(allk) / 1e6:.2f} ms; idle {(t1 - t0 - union(allk)) / 1e6:.2f} ms")
ev = []
for a, b in fam["step"]:
    ev.append((a, 1)); ev.append((b, -1))
ev.sort()
lvl = 0; last = ev[0][0]; hist = collections.Counter()
for t, d in ev:
    hist[lvl] += t - last; last = t; lvl += d
first_step = min(a for a, b in fam["step"]); last_step = max(b for a, b in fam["step"])
print("step kernels in flight (ms):", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())},
      f"over {(last_step - first_step) / 1e6:.1f} ms")
