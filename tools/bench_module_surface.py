#!/usr/bin/env python3
"""What the reference's own time loop costs over this library's `module RoadSurf` (one device step per time
index): oracle/_ref/libsimulation_over_hip.so = /root/reference/examples/example1/src/Simulation.f90 compiled
unchanged against roadsurf_amd/build/*.mod (oracle/build_ref.sh), timed per point next to the library's own
`runsimulation`.  usage: bench_module_surface.py [points=8] [hours=48]"""
import ctypes as C, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_helpers as oh   # (the synthetic forcing generator only; nothing of the oracle is timed here)
from roadsurf_amd import abi, lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hours = int(sys.argv[2]) if len(sys.argv) > 2 else 48
SL = hours * 120 + 1
L = lib.load()
sim = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsimulation_over_hip.so"))
f = oh.synth_forcing(n, SL, seed=3)
s = abi.default_settings(SL); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
for name, fn in (("module surface (Simulation.f90 over module RoadSurf)", sim.runsimulation),
                 ("runsimulation (one fused call per point)", L.runsimulation)):
    out = {k: np.full((n, SL), np.nan) for k in oh.F64_OUT}
    g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
    ptrs = [oh.point_pointers(g, pt, out) for pt in range(n)]
    ip, op, _ = ptrs[0]
    fn(C.byref(op), C.byref(ip), C.byref(s), C.byref(p), C.byref(l))  # warm-up: device, context
    t = time.perf_counter()
    for ip, op, _ in ptrs:
        fn(C.byref(op), C.byref(ip), C.byref(s), C.byref(p), C.byref(l))
    dt = (time.perf_counter() - t) / n
    print(f"{name}: {dt * 1e3:.1f} ms per point x {SL} indices = {dt / SL * 1e6:.1f} us per index, {1 / dt:.1f} points/s")
