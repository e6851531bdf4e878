#!/usr/bin/env python3
"""How wave-uniform is the sign of the ground temperatures?  (The heat capacity of a layer is a constant
below 0 C and two polynomials above: a wavefront whose 64 points all have layer j frozen could take
capDZ(j) from a table.)  Runs the synthetic workload in plan order and, every few launches, looks at the
state block: per layer the fraction of wavefronts that are all-frozen / all-thawed / mixed in the plan's
order, and in the order a stable pre-sort by the number of frozen layers would give."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from roadsurf_amd import abi, device, workload

n, hours, chunk = 250_000, 48, 240
L = hours * workload.SPK + 1
s = abi.default_settings(L); p = abi.default_parameters()
plan = device.Plan(n, s, p, 0)
plan.set_variant(3)
mode = int(os.environ.get("MODE", "378659"))
run = workload.SyntheticRun(plan, 20240110, hours, chunk, plan_order=True, forecast_mode=mode)
print("forecast mode", mode)

def look(c, t0, ns):
    if c % 4 != 3:
        return
    plan.sync()
    st = plan.state().numpy()          # [slots][np_pad], slot order = plan order at this moment
    T = st[0:15, :n]                   # Tmp(1..15)
    frozen = T < 0
    d = frozen.sum(0)
    prefix = (np.cumsum(~frozen, 0)[::-1][::-1] * 0 == 0).all()  # placeholder
    is_prefix = np.all(frozen[:-1] >= frozen[1:], axis=0).mean()
    nw = n // 64
    def stats(fr):
        w = fr[:, :nw * 64].reshape(15, nw, 64)
        allf = w.all(2); allt = (~w).all(2)
        return allf.mean(1), allt.mean(1)
    a, b = stats(frozen)
    # hypothetical: stable sort by d inside classes of the current key = current order is by key; emulate
    # "pre-sort by d then key" by sorting blocks of 4096 consecutive slots by d
    idx = np.arange(n)
    blk = 4096
    for i in range(0, n, blk):
        seg = idx[i:i + blk]
        idx[i:i + blk] = seg[np.argsort(d[seg], kind="stable")]
    a2, b2 = stats(frozen[:, idx])
    print(f"launch {c} (index {t0+ns-1}): frozen layers per point mean {d.mean():.2f}, prefix-shaped {is_prefix:.3f}")
    print("  layer      " + " ".join(f"{j:5d}" for j in range(1, 16)))
    print("  frozen frac" + " ".join(f"{x:5.2f}" for x in frozen.mean(1)))
    print("  wave allF  " + " ".join(f"{x:5.2f}" for x in a))
    print("  wave allT  " + " ".join(f"{x:5.2f}" for x in b))
    print("  sorted allF" + " ".join(f"{x:5.2f}" for x in a2))
    print("  sorted allT" + " ".join(f"{x:5.2f}" for x in b2))
    print(f"  layers 3-15 skippable per wave-step: now {a[2:].sum():.2f}, with the pre-sort {a2[2:].sum():.2f} of 13")

run.run_pass(look)
plan.sync()
