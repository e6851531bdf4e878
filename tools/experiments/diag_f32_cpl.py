"""fp32 coupling (step_kernel_f32_coupled) against the fp64 reference on the cases of tests/test_hip_coupling.py:
the distribution of the differences, per case, and where the largest ones come from."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import oracle_helpers as oh
from roadsurf_amd import abi, device
from test_hip_coupling import _cases, _kind
n, L = 384, 2881
cases, base = _cases(n, L, 4242)
for k, (f2, s, p, ls) in enumerate(cases):
    ora, _, _ = oh.run_oracle(_kind(), f2, s, p, ls)
    res, nfail = device.run_points(f2, s, p, ls, precision=32)
    d = np.abs(res["tsurf"] - ora["tsurf"])
    pm = d.max(1)
    print("case %d: failed %d rms %.2e p99 %.2e p99.9 %.2e max %.3f frac>0.05 %.1e; points with max > 0.01: %d, > 0.05: %d, > 0.2: %d" %
          (k, nfail, np.sqrt((d ** 2).mean()), np.percentile(d, 99), np.percentile(d, 99.9), d.max(), (d > 0.05).mean(),
           (pm > 0.01).sum(), (pm > 0.05).sum(), (pm > 0.2).sum()))
    moved = np.abs(ora["tsurf"] - base["tsurf"]).max(1) > 1e-3
    print("   points moved by coupling: %d" % moved.sum())
    for q in np.argsort(pm)[-5:]:
        ci = ls[q].couplingIndexI
        t1 = int((d[q] > 5e-3).argmax())
        print("   point %d ci %d obs-off %.2f max %.3f at %d; first > 5e-3 at %d; at ci-1: ora %.4f f32 %.4f obs %.4f" %
              (q, ci, ls[q].couplingTsurf - base["tsurf"][q, ci - 1], pm[q], d[q].argmax(), t1, ora["tsurf"][q, ci - 1], res["tsurf"][q, ci - 1], ls[q].couplingTsurf))
    # the window end: how close both runs came to the observation
    ci = np.array([l.couplingIndexI for l in ls]); ct = np.array([l.couplingTsurf for l in ls])
    on = (ci >= 1) & (ct > -100)
    e64 = np.array([ora["tsurf"][i, ci[i] - 1] - ct[i] for i in range(n) if on[i]])
    e32 = np.array([res["tsurf"][i, ci[i] - 1] - ct[i] for i in range(n) if on[i]])
    print("   |Tsurf(couplingEnd) - obs| <= 0.1001: fp64 %d fp32 %d of %d" % ((np.abs(e64) <= 0.1001).sum(), (np.abs(e32) <= 0.1001).sum(), on.sum()))
