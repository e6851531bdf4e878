"""Diagnostic: where do rs_driver_run and the checker differ?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import driver_helpers as dh, oracle_helpers as oh
from roadsurf_amd import abi, driver

n = 384
src, L, t0, tf = dh.scenario(n, hours=12, seed=23)
s = abi.default_settings(L); s.outputStep = 20
p = abi.default_parameters()
g = driver.run(src, s, p, t0, tf)
o = dh.oracle_run("port", src, s, p, t0, tf)
for k in driver.OUT_FIELDS:
    d = g[k] != o[k]
    pts = np.nonzero(d.any(1))[0]
    print(k, "points differing:", len(pts), pts[:10], "status", o["status"][pts[:10]])
    if len(pts):
        q = pts[0]
        rows = np.nonzero(d[q])[0]
        print("  point", q, "rows", rows[:10], "gpu", g[k][q, rows[:5]], "ora", o[k][q, rows[:5]])
