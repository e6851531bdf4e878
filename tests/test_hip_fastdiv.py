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


# ---- adversarial operands ----------------------------------------------------------------------
# Everything CheckValues admits (src/InputOutput.f90:45-84) and every parameter set plan creation
# admits must give the reference's bits, not only the synthetic workload.

def _adversarial_case():
    import numpy as np
    import oracle_helpers as oh
    from roadsurf_amd import abi
    n, L = 768, 1441
    f = oh.synth_forcing(n, L, seed=77)
    rng = np.random.default_rng(5)
    g = n // 3
    # A: |Ts - Ta| ~ 100 K in calm air (VZ at or below the calm limit): PSIM/PSIH run to about -4,
    #    the loop's denominators logUstar + PSIM, logCond + PSIH pass through zero (the reference
    #    prints 'UStar negative' a million times on this case), up to 40 passes
    f["tair"][:g] = -60.0 + 30.0 * rng.random((g, 1))
    f["tsurfobs"][:g, 0] = 10.0 + 5.0 * rng.random(g)
    f["vz"][:g] = np.where(rng.random((g, L)) < 0.5, 0.0, f["vz"][:g] * 0.05)
    f["rhz"][:g] = 0.0
    f["prec"][:g] = 0.0
    # B: the stable mirror image, and wind at the top of the accepted range
    f["tair"][g:2 * g] = 30.0 + 40.0 * rng.random((g, 1))
    f["tsurfobs"][g:2 * g, 0] = -30.0
    f["vz"][g:2 * g] = 100.0 * rng.random((g, L))
    f["rhz"][g:2 * g] = 120.0
    # C: forcing parked at the edges of its accepted interval, hopping between them every 2 h
    sl = slice(2 * g, n)
    m = n - 2 * g
    edge = lambda lo, hi: np.repeat(np.where(rng.random((m, L // 240 + 1)) < 0.5, lo, hi), 240, axis=1)[:, :L]
    f["tair"][sl] = np.where(rng.random((m, 1)) < 0.5, -90.0, 100.0) * np.ones((1, L))
    f["tair"][sl][:, 1:] *= 0.999  # the edge itself at index 1, a hair inside afterwards
    f["vz"][sl] = edge(-1.0, 100.0)
    f["rhz"][sl] = edge(-0.1, 120.0)
    f["prec"][sl] = edge(0.0, 500.0) * (rng.random((m, L)) < 0.05)
    f["sw"][sl] = edge(-0.1, 1400.0)
    f["lw"][sl] = edge(-0.1, 1000.0)
    f["precphase"][sl] = rng.integers(-1, 8, (m, L)).astype(np.int32)
    f["tsurfobs"][sl, 0] = f["tair"][sl, 0]
    f["tdew"][:] = f["tair"] - 1.0
    s = abi.default_settings(L)
    p = abi.default_parameters()
    # extreme but accepted parameters: calm limits near zero, rough momentum / smooth heat
    # surface, black-body road
    p.CalmLimDay = 0.1; p.CalmLimNgt = 0.02; p.ZMom = 1.0; p.ZHeat = 1e-5
    p.Emiss = 1.0
    l = abi.default_local(); l.InitLenI = 1
    return f, s, p, l


ADV_SCRIPT = r'''
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
from roadsurf_amd import device, lib
import test_hip_fastdiv as T
L = lib.load()
assert L.rs_hip_division_mode() == 2, "not the cross-check build"
f, s, p, l = T._adversarial_case()
res, nfail = device.run_points(f, s, p, l, lean_if_possible=False)
plan = device.Plan(16, s, p, 0)
print("DIVCHECK", L.rs_hip_div_mismatch_count(plan._h), L.rs_hip_div_special_count(plan._h), nfail)
import numpy as np
smp = np.zeros((64, 4))
L.rs_hip_div_samples(plan._h, smp.ctypes.data)
for r in smp[:12]:
    if r[2] != r[3]:
        print("SAMPLE a=%%r b=%%r ieee=%%r bare=%%r" %% tuple(float(x) for x in r))
'''


def _same(a, b):
    import numpy as np
    # NaN payloads are outside the contract (x86 and gfx950 disagree on the sign of a generated NaN)
    return np.array_equal(a, b, equal_nan=True)


@pytest.mark.skipif(not os.path.exists(LIBCHK), reason="build with `make -C roadsurf_amd divcheck`")
def test_adversarial_operands_match_the_reference_in_both_builds():
    import numpy as np
    import oracle_helpers as oh
    from roadsurf_amd import device
    f, s, p, l = _adversarial_case()
    with oh.quiet_stdout():  # 'UStar negative' x 3 million
        ora, _, _ = oh.run_oracle("ref" if oh.have_ref() else "port", f, s, p, l)
    for lean in (True, False):
        res, nfail = device.run_points(f, s, p, l, lean_if_possible=lean)
        for k in oh.F64_OUT:
            assert _same(res[k], ora[k]), (k, lean, int((res[k] != ora[k]).sum()))
    alive = (ora["tsurf"][:, -1] != -9999.0).sum()
    assert alive > 150 and nfail > 0   # some points blow up and are failed, like in the reference
    assert np.isfinite(ora["tsurf"]).all()
    env = dict(os.environ, ROADSURF_HIP_LIB=LIBCHK)
    r = subprocess.run([sys.executable, "-c", ADV_SCRIPT % {"root": ROOT}], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("DIVCHECK")][-1]
    print(line)
    print("\n".join(x for x in r.stdout.splitlines() if x.startswith("SAMPLE")))
    # no evaluation with two finite results may differ; the non-finite ones (x/0 and the like,
    # where the bare sequence gives NaN) are what the boundary-layer guard redoes in the product
    assert int(line.split()[1]) == 0


def test_infinite_latent_heat_is_absorbed_like_in_the_reference():
    """ZRefW = 0 makes logMom = logHeat = log(1) = 0 (src/Initialization.f90:330-331); with
    Ts == Ta the unstable branch returns PSIM = PSIH = -0, the aerodynamic resistance is 0 and
    LE = x/0 = +inf, which `LE > 0 and no water -> LE = 0` turns back into a finite state
    (src/BoundaryLayer.f90:126-128,183-187).  The bare division sequence yields NaN for x/0; the
    guard of rs_physics_body.inc redoes such a lane with IEEE division."""
    import numpy as np
    import oracle_helpers as oh
    from roadsurf_amd import abi, device
    n, L = 300, 241
    f = oh.synth_forcing(n, L, seed=3)
    f["tsurfobs"][:, 0] = f["tair"][:, 0]
    f["prec"][:] = 0.0
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    p.ZRefW = 0.0; p.ZeroDisp = -1.0
    ora, _, _ = oh.run_oracle("ref" if oh.have_ref() else "port", f, s, p, l)
    assert np.isfinite(ora["tsurf"]).all()
    assert (ora["tsurf"][:, -1] != -9999.0).sum() > n // 2   # LE = +inf was absorbed, the run goes on
    for variant in (1, 2):
        res, nfail = device.run_points(f, s, p, l, variant=variant)
        assert nfail == (ora["tsurf"][:, -1] == -9999.0).sum()
        for k in oh.F64_OUT:
            assert np.array_equal(res[k], ora[k]), (k, variant)


def test_parameters_outside_the_models_domain_are_refused():
    """A divisor that is zero, negative or non-finite (the reference would divide by it anyway)
    fails plan creation loudly instead of running the bare sequences outside their domain."""
    import oracle_helpers as oh  # noqa: F401
    from roadsurf_amd import abi, device
    s = abi.default_settings(100)
    for name, val in (("WatMHeat", 0.0), ("LVap", 0.0), ("VK_Const", 0.0)):
        p = abi.default_parameters()
        setattr(p, name, val)
        with pytest.raises(RuntimeError, match="domain"):
            device.Plan(8, s, p, 0)
    p = abi.default_parameters(); p.ZMom = 10.0; p.ZRefW = 0.0   # logUstar = log(1) = 0
    with pytest.raises(RuntimeError, match="logUstar"):
        device.Plan(8, s, p, 0)


FEATURE_SCRIPT = r'''
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
from roadsurf_amd import abi, device, driver, lib
import oracle_helpers as oh
import driver_helpers as dh
from test_oracle_vs_golden import _coupling_case, _skyview_case
import test_hip_coupling as TC
import test_hip_skyview as TS
L = lib.load()
assert L.rs_hip_division_mode() == 2, "not the cross-check build"
z, f, s, p, ls = _coupling_case(); device.run_points(f, s, p, ls)
z, f, s, p, ls = _skyview_case(); device.run_points(f, s, p, ls)
cases, _ = TC._cases(256, 1441, 4242)
for f2, s2, p2, ls2 in cases:
    device.run_points(f2, s2, p2, ls2)
    device.run_points(f2, s2, p2, ls2, chunk=200)          # lock-step coupling kernel + replay rounds
f3, ls3 = TS._sky_case(192, 1441, 99, summer=True, world=True)
device.run_points(f3, abi.default_settings(1441), abi.default_parameters(), ls3)
# calm wind exactly 0.0, RH 0, missing dew point: everything CheckValues admits
f4 = oh.synth_forcing(512, 1441, seed=5)
f4["vz"][:, 1:] = np.where(np.arange(1440)[None, :] %% 7 < 3, 0.0, f4["vz"][:, 1:])
f4["rhz"][::2] = 0.0
f4["tdew"][:] = -9999.9 * 0 + f4["tair"] - 2.0
l = abi.default_local(); l.InitLenI = 1
device.run_points(f4, abi.default_settings(1441), abi.default_parameters(), l, lean_if_possible=False)
# the driver data path: relaxation, coupling, rejected points (their lanes step on missing values)
src, Ld, t0, tf = dh.scenario(384, hours=12, seed=23)
for kw in (dict(use_relaxation=1), dict(use_relaxation=1, use_coupling=1)):
    sd = abi.default_settings(Ld)
    for k, v in kw.items():
        setattr(sd, k, v)
    driver.run(src, sd, abi.default_parameters(), t0, tf)
plan = device.Plan(16, abi.default_settings(100), abi.default_parameters(), 0)
print("DIVCHECK", L.rs_hip_div_mismatch_count(plan._h), L.rs_hip_div_special_count(plan._h))
'''


@pytest.mark.skipif(not os.path.exists(LIBCHK), reason="build with `make -C roadsurf_amd divcheck`")
def test_bare_division_equals_ieee_on_every_feature_path():
    """the synthetic LEAN workload is not the only operand distribution.  The
    cross-check build over the golden coupling and sky-view cases, the coupling cases in one launch
    and in chunks (lock-step coupling kernel + replay rounds), sky view around the globe, calm wind
    0.0 / RH 0, and the driver data path with relaxation, coupling and rejected points (whose lanes
    step on -9999.9): no evaluation with two finite results may differ."""
    env = dict(os.environ, ROADSURF_HIP_LIB=LIBCHK)
    r = subprocess.run([sys.executable, "-c", FEATURE_SCRIPT % {"root": ROOT}], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("DIVCHECK")][-1]
    print(line)
    assert int(line.split()[1]) == 0
