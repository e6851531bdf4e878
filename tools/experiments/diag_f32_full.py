import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import oracle_helpers as oh
from roadsurf_amd import abi, device
n, L, seed = 1024, 2881, 17
f = oh.synth_forcing(n, L, seed=seed)
s = abi.default_settings(L); s.use_relaxation = 1
p = abi.default_parameters()
lean = abi.default_settings(L)
l0 = abi.default_local(); l0.InitLenI = 1
base, _, _ = oh.run_oracle("port", f, lean, p, l0)
rs = np.random.RandomState(5)
ls = []
for i in range(n):
    li = abi.default_local()
    li.InitLenI = int(rs.choice([1, 240, 600, 721, 1000]))
    li.tair_relax = float(f["tair"][i, min(li.InitLenI, L - 1)] + rs.uniform(-2, 2))
    li.VZ_relax = float(rs.uniform(0.5, 6.0)); li.RH_relax = float(rs.uniform(60, 99))
    if i % 37 == 0: li.tair_relax = -9999.0
    ls.append(li)
f["tsurfobs"][:, :] = base["tsurf"] + rs.uniform(-1.5, 1.5, (n, 1))
f["tsurfobs"][::5, 300:500] = -9999.9
ora, _, _ = oh.run_oracle("port", f, s, p, ls)
res, nfail = device.run_points(f, s, p, ls, chunk=0, precision=32)
d = np.abs(res["tsurf"] - ora["tsurf"])
print("rms %.2e p99 %.2e p99.9 %.2e max %.3f frac>0.05 %.1e" % (np.sqrt((d**2).mean()), np.percentile(d,99), np.percentile(d,99.9), d.max(), (d>0.05).mean()))
pm = d.max(1)
worst = np.argsort(pm)[-8:]
for q in worst:
    t = d[q].argmax()
    print("point", q, "initlen", ls[q].InitLenI, "max", pm[q], "at", t, "first >1e-3 at", int((d[q] > 1e-3).argmax()), "relax", ls[q].tair_relax)
il = np.array([l.InitLenI for l in ls])
for v in (1, 240, 600, 721, 1000):
    m = il == v
    print("initlen", v, "rms", np.sqrt((d[m]**2).mean()), "max", d[m].max())
# time profile of rms
for a, b in ((0, 240), (240, 600), (600, 1000), (1000, 1500), (1500, 2881)):
    print("indices", a, b, "rms %.2e" % np.sqrt((d[:, a:b]**2).mean()))
q = 1
t = int((d[q] > 1e-3).argmax())
print("point", q, "first deviation at index", t + 1)
for k in ("tsurf", "snow", "water", "ice", "deposit", "ice2"):
    print(k, "ora", ora[k][q, t - 2:t + 4], "f32", res[k][q, t - 2:t + 4])
print("prec", f["prec"][q, t - 2:t + 4], "tair", f["tair"][q, t - 2:t + 4], "rh", f["rhz"][q, t - 2:t + 4], "phase", f["precphase"][q, t - 2:t + 4])
