#!/usr/bin/env python3
"""Per-wavefront start/end times inside one step launch (library built with -DRS_WAVE_TIMING:
`make -C roadsurf_amd OBJ=build_wt LIB=lib/libroadsurf_hip_wt.so EXTRA=-DRS_WAVE_TIMING`, run with
ROADSURF_HIP_LIB pointing at it).  One plan of N points in plan order (forecast re-sort), launches of
`chunk` indices; after launch number `which` the state rows that carry the ticks are read back.

usage: wave_times.py N [chunk] [which]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from roadsurf_amd import abi, device, lib, workload

n = int(sys.argv[1]); chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 240
which = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [3, 8, 14]
hours = 48
simlen = hours * workload.SPK + 1
pl = device.Plan(n, abi.default_settings(simlen), abi.default_parameters(), 0)
pl.set_variant(1)
run = workload.SyntheticRun(pl, 20240110, hours, chunk, plan_order=True)
VZ, RH = lib.RS_MAX_LAYERS + 14, lib.RS_MAX_LAYERS + 15  # RS_ST_VZ_END, RS_ST_RH_END
for c in run.iter_pass():
    if c in which:
        torch.cuda.synchronize()
        st = pl.state()
        t0 = st[VZ, :n].numpy(); t1 = st[RH, :n].numpy()
        w0 = t0[::64]; w1 = t1[::64]           # one value per wavefront
        base = w0.min()
        dur = (w1 - w0) / 100.0                # us
        span = (w1.max() - base) / 100.0
        print(f"launch {c}: {len(w0)} waves, launch span {span:.0f} us; wave duration us: mean {dur.mean():.0f} "
              f"median {np.median(dur):.0f} p90 {np.percentile(dur, 90):.0f} p99 {np.percentile(dur, 99):.0f} max {dur.max():.0f}; "
              f"latest start {((w0 - base).max()) / 100.0:.0f} us")
        # by slot decile (plan order = expensive first)
        k = max(len(dur) // 10, 1)
        print("   mean duration by decile of the slot order:", " ".join(f"{dur[i*k:(i+1)*k].mean():.0f}" for i in range(10)))
        print(f"   sum of wave durations / (1024 SIMDs x span) = {dur.sum() / (1024 * span):.2f} waves busy per SIMD on average")
