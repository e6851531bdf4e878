"""GPU: the bare (scale/fixup-free) division and square root of rs_math.hpp must return
the IEEE result at EVERY call-site evaluation of the BASELINE workload.  The cross-check
build (-DRS_DIV_CHECK) evaluates both forms and counts disagreements on the device; the
full 1 M-point x 48 h synthetic workload (~2.7e11 divisions) must produce 0."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBCHK = os.path.join(ROOT, "roadsurf_amd", "lib", "libroadsurf_hip_divcheck.so")

SCRIPT = r'''
import sys, ctypes as C
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
from roadsurf_amd import abi, device, lib
import oracle_helpers as oh
L = lib.load()
assert L.rs_hip_division_mode() == 2, "not the cross-check build"
n, hours, spk, chunk = %(points)d, 48, 120, 240
simlen = hours * spk + 1
s = abi.default_settings(simlen); p = abi.default_parameters()
plan = device.Plan(n, s, p, 0)
dev = plan.device
spec, knots = plan.synth_knots(20240110, hours + 2, steps_per_knot=spk)
win = device.ForcingWindow.empty(chunk, plan.np_pad, dev, optional=())
win0 = device.ForcingWindow.empty(1, plan.np_pad, dev, optional=("tsurfobs",))
out = device.OutputWindow.empty(chunk, plan.np_pad, dev)
pp = plan.point_params(plan.uniform_tbottom(2024, 1, 10))
plan.expand(spec, knots, win0, 1, 1); plan.init_state(win0, pp)
t0 = 1
while t0 <= simlen:
    ns = min(chunk, simlen - t0 + 1)
    plan.expand(spec, knots, win, t0, ns); plan.step(win, out, pp, t0, ns, out_row0=t0 - 1)
    t0 += ns
plan.sync()
m1 = L.rs_hip_div_mismatch_count(plan._h)
plan.close()
# the FULL variant too (relaxation, observation forcing, output depth), smaller
n2, L2 = 2048, 2881
f = oh.synth_forcing(n2, L2, seed=9)
f["tsurfobs"][:, :1440] = f["tair"][:, :1440] - 0.5
f["depth"][::2] = 0.07
ls = []
for i in range(n2):
    li = abi.default_local(); li.InitLenI = 1440; li.tair_relax = float(f["tair"][i, 1440]) + 1.0
    li.VZ_relax = 3.0; li.RH_relax = 80.0; ls.append(li)
s2 = abi.default_settings(L2); s2.use_relaxation = 1
res, _ = device.run_points(f, s2, p, ls)
plan = device.Plan(16, s2, p, 0)
m2 = L.rs_hip_div_mismatch_count(plan._h)
print("DIVCHECK", m1, m2)
'''


@pytest.mark.skipif(not os.path.exists(LIBCHK), reason="build with `make -C roadsurf_amd divcheck`")
def test_bare_division_equals_ieee_on_full_workload():
    env = dict(os.environ, ROADSURF_HIP_LIB=LIBCHK)
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "points": 1_000_000}],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("DIVCHECK")][-1]
    _, m1, m2 = line.split()
    print(line)
    assert int(m1) == 0 and int(m2) == 0
