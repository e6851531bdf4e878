"""Where the fp32 flavour with sky view differs from the fp64 reference on the 'midsummer anywhere on the globe'
case of tests/test_hip_f32.py, and what the same forcing gives WITHOUT sky view (is it the feature or the flavour?)."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import oracle_helpers as oh
from roadsurf_amd import abi, device
from test_hip_skyview import _sky_case
n, L = 512, 2881
for summer, world in ((True, False), (True, True)):
    f, ls = _sky_case(n, L, 41, summer, world)
    for li in ls:
        li.InitLenI = 1
    s = abi.default_settings(L); p = abi.default_parameters()
    ora, _, _ = oh.run_oracle("port", f, s, p, ls)
    res, _ = device.run_points(f, s, p, ls, precision=32)
    r64, _ = device.run_points(f, s, p, ls)
    print("case summer=%s world=%s: fp64 device == oracle: %s" % (summer, world, np.array_equal(r64["tsurf"], ora["tsurf"])))
    d = np.abs(res["tsurf"] - ora["tsurf"])
    print("  sky : rms %.2e p99 %.2e p99.9 %.2e max %.3f frac>0.05 %.1e" % (np.sqrt((d**2).mean()), np.percentile(d, 99), np.percentile(d, 99.9), d.max(), (d > 0.05).mean()))
    l0 = abi.default_local(); l0.InitLenI = 1
    o0, _, _ = oh.run_oracle("port", f, s, p, l0)
    r0, _ = device.run_points(f, s, p, l0, precision=32, lean_if_possible=False)
    d0 = np.abs(r0["tsurf"] - o0["tsurf"])
    print("  none: rms %.2e p99 %.2e p99.9 %.2e max %.3f frac>0.05 %.1e" % (np.sqrt((d0**2).mean()), np.percentile(d0, 99), np.percentile(d0, 99.9), d0.max(), (d0 > 0.05).mean()))
    pm = d.max(1)
    for q in np.argsort(pm)[-6:]:
        t = int(d[q].argmax()); t1 = int((d[q] > 2e-3).argmax())
        print("  point", q, "sky_view", ls[q].sky_view, "lat %.1f lon %.1f" % (ls[q].lat, ls[q].lon), "max %.3f at %d" % (pm[q], t), "first > 2e-3 at", t1,
              "tsurf there", ora["tsurf"][q, t1], "range of tsurf", ora["tsurf"][q].min(), ora["tsurf"][q].max())
        for k in ("tsurf", "snow", "water", "ice", "deposit", "ice2"):
            print("     ", k, "ora", np.round(ora[k][q, t1 - 2:t1 + 3], 5), "f32", np.round(res[k][q, t1 - 2:t1 + 3], 5))
    # how many points ever exceed 0.05 K, and the sum over sky-view classes
    sv = np.array([l.sky_view for l in ls])
    for v in (0.0, 0.3, 0.75, 0.99, 1.0):
        m = sv == v
        print("  sky_view", v, "points", int(m.sum()), "rms %.2e max %.3f" % (np.sqrt((d[m]**2).mean()), d[m].max()))
