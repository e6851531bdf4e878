"""How often does the boundary-layer fixed point (src/BoundaryLayer.f90:64-96) repeat its bits before the
fifth mandatory pass?  Runs the synthetic bench workload (plan order, knot-reading two-wavefront flavour) in
the experiment build `make -C roadsurf_amd blstats` (ROADSURF_HIP_LIB=.../libroadsurf_hip_blstats.so) and
prints the counters of rs_hip_bl_stats.  usage: bl_fixpoint_stats.py [points] [hours]"""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
os.environ.setdefault("ROADSURF_HIP_LIB", os.path.join(ROOT, "roadsurf_amd", "lib", "libroadsurf_hip_blstats.so"))
sys.path.insert(0, ROOT)
import torch
from roadsurf_amd import abi, device, lib, workload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
hours = int(sys.argv[2]) if len(sys.argv) > 2 else 48
simlen = hours * 120 + 1
s = abi.default_settings(simlen); p = abi.default_parameters()
plan = device.Plan(n, s, p, 0)
plan.set_variant(3)
run = workload.SyntheticRun(plan, 1, hours, 60, plan_order=True, forecast=True, forecast_mode=workload.DEFAULT_FORECAST_MODE)
run.run_pass(None)
torch.cuda.synchronize()
L = lib.load()
out = (C.c_int64 * 8)()
L.rs_hip_bl_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
assert L.rs_hip_bl_stats(plan._h, out) == 0
w, l = out[0], out[1]
print(f"{n} points x {hours} h: {w} wave-steps, {l} lane-steps")
for j in (2, 3, 4):
    print(f"  fixed point reached at pass {j}: whole wavefront {out[j] / w:.4f} of wave-steps, lanes {out[3 + j] / l:.4f} of lane-steps")
