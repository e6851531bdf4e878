#!/usr/bin/env python3
"""Small shards: what bounds a launch whose wavefronts all have a SIMD (slot) of their own - the SLOWEST
wavefront, not the sum.  From the CPU checker's trajectory (tools/bl_costmodel.py: per point-step which
boundary-layer passes ran and which branch they took) this replays launches of CH indices with the
forecast sort key and reports, per launch, the modelled instruction chain of
  - the mean wavefront,
  - the slowest wavefront (64 lanes),
  - the slowest wavefront if the most expensive W0 wavefronts of the order are dealt 16 lanes to a wave,
  - the slowest single POINT (what no regrouping can beat).
Chain of a wavefront per step = FIXED + the union cost of its lanes' boundary-layer passes.
usage: N=16384 python tools/bl_makespan.py [CH=240]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(__file__))
import bl_costmodel as cm
from bl_costmodel import C_COMMON, C_FAR, C_NEAR, C_SQRT, C_STABLE, popcount

FIXED = 900  # vector+scalar instructions of a step outside the boundary-layer loop (one point per lane)
CH = int(sys.argv[1]) if len(sys.argv) > 1 else 240


def union_cost(d, rows, sl, width):
    """per-wave chain (sum over the steps of sl) for consecutive groups of `width` rows"""
    def u(x):
        return np.bitwise_or.reduce(x[rows][:, sl].reshape(len(rows) // width, width, -1), axis=1)
    a, n_, f_, s_ = u(d["act"]), u(d["near"]), u(d["far"]), u(d["stb"])
    c = (C_COMMON * popcount(a) + C_SQRT * popcount(n_ | f_) + C_NEAR * popcount(n_) + C_FAR * popcount(f_) +
         C_STABLE * popcount(s_))
    return c.sum(1) + FIXED * (sl.stop - sl.start)


def main():
    from roadsurf_amd import abi, lib
    d = cm.build()
    N, L = cm.N, cm.L
    consts = lib.build_constants(abi.default_settings(L), abi.default_parameters())
    key = cm.make_forecast_key(consts, 3, 0.5, "unst+near+extra")
    order = np.arange(N)
    print(f"{N} points, launches of {CH} indices, forecast sort key; chain = {FIXED} + boundary-layer union cost per step")
    print("launch  mean-wave  p99-wave  max-wave  max-wave(top 8 waves dealt 16 lanes)  max-point   max/mean")
    acc = []
    for c0 in range(0, L - 1, CH):
        sl = slice(c0, min(L, c0 + CH))
        w64 = union_cost(d, order, sl, 64)
        top = order[:8 * 64]
        w16 = union_cost(d, top, sl, 16)
        rest = w64[8:]
        pt = union_cost(d, order, sl, 1)
        acc.append((w64.mean(), np.percentile(w64, 99), w64.max(), max(w16.max(), rest.max() if len(rest) else 0), pt.max()))
        nxt = c0 + CH
        if nxt < L:
            order = np.argsort(-key(d, sl, nxt, CH), kind="stable")
    a = np.array(acc) / CH
    for i, r in enumerate(a):
        if i % 3 == 0:
            print(f"{i:5d}  {r[0]:9.0f} {r[1]:9.0f} {r[2]:9.0f} {r[3]:12.0f} {r[4]:30.0f} {r[2] / r[0]:9.2f}")
    m = a.mean(0)
    print(f"mean over launches: mean-wave {m[0]:.0f}, p99 {m[1]:.0f}, max-wave {m[2]:.0f} ({m[2] / m[0]:.2f} x mean), "
          f"top waves narrow {m[3]:.0f} ({m[3] / m[0]:.2f} x), slowest point {m[4]:.0f} ({m[4] / m[0]:.2f} x) instructions per step")


if __name__ == "__main__":
    main()
