"""Rate of the raw-series boundary (rs_driver_run): hourly forecast + 10-minute observations in,
hourly outputs back, everything else on the GPU.  PCIe-inclusive, host arrays pageable.
(The workload is roadsurf_amd/driver_workload.py, the one bench.py's driver-path legs time.)
usage: python tools/bench_driver_path.py [n_points] [hours] [mode: plain|relax|coupling|skyview|skycoupling] [tsurfOutputDepth]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from roadsurf_amd import driver_workload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
hours = int(sys.argv[2]) if len(sys.argv) > 2 else 48
mode = sys.argv[3] if len(sys.argv) > 3 else "relax"
depth = float(sys.argv[4]) if len(sys.argv) > 4 else None
unique = int(os.environ["BENCH_UNIQUE"]) if "BENCH_UNIQUE" in os.environ else None
w = driver_workload.DriverWorkload(n, hours, unique=unique, pinned=bool(int(os.environ.get("BENCH_PINNED", "0"))),
                                   missing=float(os.environ.get("BENCH_MISSING", "0")),  # BENCH_MISSING=0.1: stations without sensors, gaps
                                   ragged=float(os.environ.get("BENCH_RAGGED", "0")),    # BENCH_RAGGED=0.2: observations that end hours apart
                                   weather=os.environ.get("BENCH_WEATHER", "driver"))   # BENCH_WEATHER=bench: bench.py's weather as raw series (A/B)
best, times, r = w.time_calls(mode, reps=int(os.environ.get("BENCH_REPS", "3")), warm=1,
                              device=int(os.environ.get("BENCH_DEVICE", "-1")), tsurf_output_depth=depth,
                              verbose=True, pause=float(os.environ.get("BENCH_PAUSE_S", "0")))
print(f"best {best:.3f} s -> {n * w.simlen / best:.3e} point-timesteps/s")
print("tsurf sample", r["tsurf"][0, :4], r["tsurf"][n // 2, -3:])
