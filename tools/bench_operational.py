"""The reference's own operational shape (tests/golden/e2e_operational.npz: 401 stations, 48 h analysis + 26 h
forecast = SimLen 8 881, coupling + relaxation) through rs_driver_run: milliseconds per call.  A latency case -
seven wavefronts, 8 881 + up to 25 x 361 dependent steps - quoted in DESIGN.md 4.
usage: python tools/bench_operational.py [files|sky] [reps]   (ROADSURF_HIP_DRIVER_TIMING=1: phase times)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import driver_helpers as dh
import golden_helpers as gh
from roadsurf_amd import driver

case = sys.argv[1] if len(sys.argv) > 1 else "files"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
z = gh.load("e2e_operational.npz")
src, s, p, t0, tf, local, hz = dh.operational_case(z, case)
n = len(z["lat"])
ts = []
for r in range(reps + 1):
    a = time.perf_counter()
    g = driver.run(src, s, p, t0, tf, local=local, horizons=hz)
    ts.append(time.perf_counter() - a)
print(f"{case}: {n} stations x {s.SimLen} indices: first call {ts[0] * 1e3:.1f} ms, then best {min(ts[1:]) * 1e3:.1f} ms, "
      f"mean {sum(ts[1:]) / reps * 1e3:.1f} ms -> {n * s.SimLen / min(ts[1:]):.3e} point-timesteps/s; ok {int((g['status'] == 0).sum())}")
