"""Rate of the raw-series boundary (rs_driver_run): hourly forecast + 10-minute observations in,
hourly outputs back, everything else on the GPU.  PCIe-inclusive, host arrays pageable.
usage: python tools/bench_driver_path.py [n_points] [hours] [mode: plain|relax|coupling|skyview|skycoupling] [tsurfOutputDepth]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from roadsurf_amd import abi, driver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
hours = int(sys.argv[2]) if len(sys.argv) > 2 else 48
mode = sys.argv[3] if len(sys.argv) > 3 else "relax"
L = hours * 120 + 1
START = 1704844800
rs = np.random.RandomState(1)

def series(nt, dt, lo, hi, amp, period=86400.0):
    base = rs.uniform(lo, hi, (n, 1))
    ph = rs.uniform(0, 2 * np.pi, (n, 1))
    t = np.arange(nt)[None, :] * dt
    return base + amp * np.sin(2 * np.pi * t / period + ph)

nt_fc = hours + 3
fc_t = START - 3600 + np.arange(nt_fc, dtype=np.int64) * 3600
tair = series(nt_fc, 3600, -12, 6, 4.0)
fc = dict(tair=tair, tdew=tair - rs.uniform(0.5, 4, (n, 1)), vz=np.abs(series(nt_fc, 3600, 1, 8, 2.0, 43200.0)) + 0.2,
          prec=np.where(rs.rand(n, nt_fc) < 0.1, rs.uniform(0, 2, (n, nt_fc)), 0.0),
          sw=np.maximum(0.0, series(nt_fc, 3600, -50, 150, 200.0)), lw=series(nt_fc, 3600, 230, 320, 15.0))
obs_h = 6
nt_ob = obs_h * 6 + 1
ob_t = START + np.arange(nt_ob, dtype=np.int64) * 600
ob = dict(tair=series(nt_ob, 600, -12, 6, 1.0), rhz=np.clip(series(nt_ob, 600, 70, 95, 5.0), 5, 100),
          vz=np.abs(series(nt_ob, 600, 1, 8, 1.0)) + 0.2, tsurfobs=series(nt_ob, 600, -10, 4, 1.0))
src = [driver.RawSource(fc_t, fc, False), driver.RawSource(ob_t, ob, True)]
s = abi.default_settings(L)
s.use_relaxation = 1 if mode in ("relax", "coupling", "skyview", "skycoupling") else 0
s.use_coupling = 1 if mode in ("coupling", "skycoupling") else 0
if len(sys.argv) > 4:  # TsurfAve at this depth below the surface instead of the top layers' mean
    s.tsurfOutputDepth = float(sys.argv[4])
p = abi.default_parameters()
cal = driver.calendar(START, L, 30)
raw_bytes = sum(a.nbytes for d in (fc, ob) for a in d.values())
local = driver._locals(n, None)
hz = None
if mode in ("skyview", "skycoupling"):
    fc["sw_dir"] = 0.6 * fc["sw"]
    fc["lw_net"] = np.full_like(fc["lw"], -40.0)
    sv = rs.uniform(0.3, 1.0, n)
    for q in range(n):
        local[q].lat, local[q].lon, local[q].sky_view = 60.0 + (q % 97) * 0.05, 22.0 + (q % 89) * 0.05, sv[q]
    hz = rs.uniform(0, 20, (n, 360))
r = None
for rep in range(4):
    t0 = time.time()
    # device -1: the library's own fan-out (ROADSURF_HIP_DEVICES / ROADSURF_HIP_PLANS_PER_DEVICE);
    # the caller's result arrays are reused from the second call on
    r = driver.run(src, s, p, START, START + obs_h * 3600, cal=cal, local=local, horizons=hz,
                   device=int(os.environ.get("BENCH_DEVICE", "-1")), out=r)
    dt = time.time() - t0
    out_bytes = sum(r[k].nbytes for k in driver.OUT_FIELDS)
    print(f"rep {rep}: n={n} L={L} mode={mode}: {dt:.3f} s  -> {n * L / dt:.3e} point-timesteps/s "
          f"(raw in {raw_bytes / 1e9:.2f} GB, out {out_bytes / 1e9:.2f} GB, ok={int((r['status'] == 0).sum())})", flush=True)
print("tsurf sample", r["tsurf"][0, :4], r["tsurf"][n // 2, -3:])
