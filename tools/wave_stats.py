"""What a wavefront of the step kernel pays for on a workload: boundary-layer passes (the wavefront's trip count is
its slowest lane's), which of road_condition's storage blocks any lane needs, how often the precipitation branch
runs.  Experiment build `make -C roadsurf_amd blstats` (counters in rs_math.hpp g_bl_stats, read by
rs_hip_bl_stats); the product build carries none of this.
usage: wave_stats.py bench|bench-full|driver-relax|driver-coupling [points] [hours]"""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
os.environ.setdefault("ROADSURF_HIP_LIB", os.path.join(ROOT, "roadsurf_amd", "lib", "libroadsurf_hip_blstats.so"))
sys.path.insert(0, ROOT)
import torch
from roadsurf_amd import abi, device, lib, workload

what = sys.argv[1] if len(sys.argv) > 1 else "bench"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 250000
hours = int(sys.argv[3]) if len(sys.argv) > 3 else 48
simlen = hours * 120 + 1
s = abi.default_settings(simlen); p = abi.default_parameters()
if what == "bench-full":
    s.use_relaxation = 1  # (as bench.py --full)
if what.startswith("driver"):
    from roadsurf_amd import driver_workload
    w = driver_workload.DriverWorkload(n, hours)
    w.time_calls(what.split("-", 1)[1], reps=1, warm=0)
    plan = device.Plan(256, s, p, 0)  # (only to reach the counters)
else:
    plan = device.Plan(n, s, p, 0)
    plan.set_variant(3)
    run = workload.SyntheticRun(plan, 1, hours, 60, plan_order=True, forecast=True,
                                forecast_mode=workload.DEFAULT_FORECAST_MODE, full=(what == "bench-full"))
    run.run_pass(None)
torch.cuda.synchronize()
L = lib.load()
out = (C.c_int64 * 56)()
L.rs_hip_bl_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
assert L.rs_hip_bl_stats(plan._h, out) == 0
ws, ls = max(out[0], 1), max(out[1], 1)
print(f"{what}: {n} points x {hours} h: {out[0]} wave-steps, {out[1]} lane-steps of the boundary-layer loop")
print(f"  passes per wave-step {out[8] / ws:.3f} (its slowest lane), per lane-step {out[9] / ls:.3f}")
names = ["5", "6", "7", "8", "9-12", "13-20", "21-39", "40"]
print("  wave-steps by trip count: " + ", ".join(f"{k}: {out[24 + i] / ws:.4f}" for i, k in enumerate(names)))
print("  lane-steps by trip count: " + ", ".join(f"{k}: {out[32 + i] / ls:.4f}" for i, k in enumerate(names)))
print("  wave-steps with lanes of more than 20 passes, by how many: " + ", ".join(f"{k}: {out[40 + i] / ws:.4f}" for i, k in enumerate(["1", "2-3", "4-7", "8-15", "16-31", "32-64"])))
wp = max(out[19], 1)
print(f"  passes as issued: {out[19]} ({out[19] / ws:.3f} per wave-step); with a lane on the unstable arm {out[20] / wp:.4f}, on both arms {out[22] / wp:.4f}, "
      f"on log's table path {out[23] / wp:.4f}, on both paths of log {out[21] / wp:.4f}")
lp = max(out[48], 1)
print(f"  lane-passes: {out[48]}; on the unstable arm {out[46] / lp:.4f}, on log's table path {out[47] / lp:.4f}")
rc = max(out[10], 1)
print(f"  road_condition: {out[10]} wave-steps; bare-road shortcut {out[11] / rc:.4f}; every lane without snow {out[12] / rc:.4f}, "
      f"without ice {out[13] / rc:.4f}, without deposit or condensation {out[14] / rc:.4f}, without water {out[15] / rc:.4f}, "
      f"without snow, ice and deposit {out[16] / rc:.4f}")
fp = max(out[17], 1)
print(f"  forcing's share: {out[17]} wave-steps, precipitation branch in {out[18] / fp:.4f}")
