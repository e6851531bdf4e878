import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_helpers as oh
import test_hip_coupling as T
from roadsurf_amd import device
n, L = 384, 2881
cases, base = T._cases(n, L, 4242)
for k in (0,):
    f2, s, p, ls = cases[k]
    with oh.quiet_stdout():
        ora, _, _ = oh.run_oracle(T._kind(), f2, s, p, ls)
    res, _ = device.run_points(f2, s, p, ls, chunk=int(os.environ.get("CHUNK", 97)))
    bad = res["tsurf"] != ora["tsurf"]
    pts = np.where(bad.any(1))[0]
    print("case", k, "points differing", len(pts), "of", n)
    first = bad.argmax(1)
    ci = np.array([l.couplingIndexI for l in ls])
    for q in pts[:8]:
        print(" point", q, "ci", ci[q], "cs", ci[q] - 360, "first diff at 0-based", first[q], "gpu", res["tsurf"][q, first[q]], "ref", ora["tsurf"][q, first[q]], "ndiff", bad[q].sum())
    import collections
    print(" histogram of first-diff index:", collections.Counter(first[pts].tolist()).most_common(5))
