"""How fast the fp32 flavour's GENERAL kernel is (step_kernel_f32_coupled: one point per lane, a time index per lane;
coupling, output depth, other layer counts) next to the fp64 kernels for the same launch: 65 536 points x 24 h, the step
call alone (HIP-synchronised wall clock around rs_hip_step)."""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import torch
import oracle_helpers as oh
from roadsurf_amd import abi, device

times = []
orig = device.Plan.step
def timed(self, *a, **k):
    torch.cuda.synchronize(); t = time.perf_counter()
    orig(self, *a, **k); self.sync(); torch.cuda.synchronize()
    times.append(time.perf_counter() - t)
device.Plan.step = timed

n, L = 65536, 2881
f = oh.synth_forcing(n, L, seed=11)
p = abi.default_parameters()
rs = np.random.RandomState(1)
def locals_(cpl):
    ls = []
    for i in range(n):
        li = abi.default_local(); li.InitLenI = L // 2 if cpl else 1
        if cpl:
            li.couplingIndexI = L // 2
            li.couplingTsurf = float(f["tair"][i, L // 2 - 1] + rs.choice([0.0, 0.5, -0.5, 2.0, -2.0]))
        ls.append(li)
    return ls
f["tsurfobs"][:, :] = f["tair"] + 0.5
for what in ("coupling", "depth-setting", "lean"):
    s = abi.default_settings(L)
    if what == "coupling":
        s.use_coupling = 1
    if what == "depth-setting":
        s.tsurfOutputDepth = 0.05
    ls = locals_(what == "coupling")
    for prec in (32, 64):
        for rep in range(2):
            times.clear()
            device.run_points(f, s, p, ls, precision=prec)
        t = sum(times)
        print("%-14s fp%d: %.1f ms  %.3e point-timesteps/s" % (what, prec, 1e3 * t, n * L / t), flush=True)
