"""Diagnostic: where does the adversarial case of tests/test_hip_fastdiv.py first leave the reference?"""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_helpers as oh
import test_hip_fastdiv as T
from roadsurf_amd import device, lib

f, s, p, l = T._adversarial_case()
with oh.quiet_stdout():
    ora, _, _ = oh.run_oracle("ref", f, s, p, l)
res, nfail = device.run_points(f, s, p, l)
print("division mode", lib.load().rs_hip_division_mode(), "failed", nfail)
bad = ~np.isclose(res["tsurf"], ora["tsurf"], rtol=0, atol=0, equal_nan=True)
pts = np.where(bad.any(1))[0]
print("points that differ:", len(pts), "by group:", [(int((pts < 256).sum())), int(((pts >= 256) & (pts < 512)).sum()), int((pts >= 512).sum())])
for q in pts[:6]:
    i = int(np.argmax(bad[q]))
    print(f"point {q}: first difference at index {i} (0-based): gpu {res['tsurf'][q, i]!r} ref {ora['tsurf'][q, i]!r}")
    for k in ("tair", "vz", "rhz", "prec", "sw", "lw", "precphase", "tsurfobs"):
        print("    ", k, f[k][q, max(0, i - 1):i + 2])
    print("     prev tsurf gpu/ref", res["tsurf"][q, i - 1], ora["tsurf"][q, i - 1], "hour", f["hour"][i])
    for k in oh.F64_OUT:
        print("     ", k, res[k][q, i], ora[k][q, i])
