"""Diagnostic: distribution of the CalcBLCondAndLE trip count (src/BoundaryLayer.f90:64-96) on the
synthetic workload, per point-step, from the CPU checker.  Design input for the kernel: a wave
runs the loop until its slowest lane is done."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import oracle_helpers as oh
from roadsurf_amd import abi

n, L = 2048, 5761
lib = oh.load("port")
hist = (C.c_long * 64).in_dll(lib, "oracle_bl_hist")
C.c_int.in_dll(lib, "oracle_bl_hist_on").value = 1
f = oh.synth_forcing(n, L, seed=20240110)
s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
oh.run_oracle("port", f, s, p, l)
h = np.array(hist[:], dtype=np.float64)
tot = h.sum()
print("calls", int(tot), "mean trip count %.3f" % ((h * np.arange(64)).sum() / tot))
cdf = np.cumsum(h) / tot
for j in range(5, 41):
    if h[j]:
        # probability that the max over 64 i.i.d. lanes is <= j
        print(f"j={j:2d}  share {h[j] / tot:9.6f}  cdf {cdf[j]:.6f}  P(max of 64 <= j) {cdf[j] ** 64:.4f}")
pm = np.diff(np.concatenate([[0.0], cdf ** 64]))
print("expected max over 64 i.i.d. lanes: %.3f" % (pm * np.arange(64)).sum())
