"""PCIe-inclusive rate of the drop-in host-array entry (runsimulation_batch)."""
import ctypes as C, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle_helpers as oh
from roadsurf_amd import abi, lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
SL = 5761
L = lib.load()
f = oh.synth_forcing(n, SL, seed=1)
out = {k: np.empty((n, SL)) for k in oh.F64_OUT}
s = abi.default_settings(SL); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
ips = (abi.InputPointers * n)(); ops = (abi.OutputPointers * n)(); keep = []
for pt in range(n):
    ip, op, kp = oh.point_pointers(f, pt, out)
    ips[pt], ops[pt] = ip, op; keep.append(kp)
larr = (abi.LocalParameters * n)(*([l] * n))
st = C.c_int32(0)
rates = []
for rep in range(int(os.environ.get("REPS", "3"))):
    t = time.perf_counter()
    L.runsimulation_batch(n, ops, ips, C.byref(s), C.byref(p), larr, C.byref(st))
    dt = time.perf_counter() - t
    assert st.value == 0, lib.last_error()
    rates.append(n * SL / dt)
    print(f"runsimulation_batch: {n} points x {SL}: {dt:.3f} s -> {n*SL/dt:.3e} point-timesteps/s "
          f"({n*SL*(11*8+2*4+6*8)/dt/1e9:.1f} GB/s over the boundary)")
rates = sorted(rates[1:]) or rates
print(f"summary: median {rates[len(rates)//2]:.3e} best {rates[-1]:.3e} worst {rates[0]:.3e} (first call excluded)")
