"""GPU: randomly drawn settings through the C-ABI batch entry against the checker, bit for bit.
Each seed combines layer count, time step, series length, relaxation, coupling, sky view with
local horizons, output depth (setting and per-index array), force_tsurf and per-point initialisation
lengths: the interactions the per-feature tests do not cover."""
import ctypes as C

import numpy as np
import pytest

import oracle_helpers as oh
from roadsurf_amd import abi, lib

pytestmark = pytest.mark.gpu


def _draw(seed):
    rs = np.random.RandomState(1000 + seed)
    n = int(rs.choice([97, 130, 256]))
    dt = float(rs.choice([20.0, 30.0, 60.0]))
    L = int(rs.choice([721, 1201, 1441]))
    s = abi.default_settings(L, dt)
    s.NLayers = int(rs.choice([6, 15, 15, 27]))
    s.use_relaxation = int(rs.rand() < 0.6)
    s.use_coupling = int(rs.rand() < 0.5)
    s.coupling_minutes = int(rs.choice([30, 60, 180]))
    s.force_tsurf = int(rs.rand() < 0.15)
    s.tsurfOutputDepth = float(rs.choice([-9999.9, -9999.9, 0.0, 0.04, 0.5]))
    p = abi.default_parameters(dt)
    f = oh.synth_forcing(n, L, seed=77 + seed, steps_per_knot=int(3600 / dt))
    base_s = abi.default_settings(L, dt); base_s.NLayers = s.NLayers
    l0 = abi.default_local(); l0.InitLenI = 1
    base, _, _ = oh.run_oracle("port", f, base_s, p, l0)
    sky = rs.rand() < 0.5
    cpl_len = int(s.coupling_minutes * 60 / dt)
    ls = []
    for i in range(n):
        li = abi.default_local()
        li.InitLenI = int(rs.randint(1, L // 2))
        if rs.rand() < 0.8:
            li.tair_relax = float(f["tair"][i, li.InitLenI] + rs.uniform(-2, 2))
            li.VZ_relax = float(rs.uniform(0.5, 8)); li.RH_relax = float(rs.uniform(50, 100))
        if rs.rand() < 0.85:
            ci = int(rs.randint(cpl_len + 2, L - 5))
            li.couplingIndexI = ci
            li.couplingTsurf = float(base["tsurf"][i, ci - 1] + rs.choice([0.0, 0.3, -0.7, 3.0, -4.0]))
        li.lat, li.lon = float(rs.uniform(59, 70)), float(rs.uniform(20, 30))
        li.sky_view = float(rs.uniform(0.2, 1.0)) if (sky and rs.rand() < 0.7) else 1.0
        ls.append(li)
    # road temperature observations: present in stretches, missing elsewhere
    obs = base["tsurf"] + rs.uniform(-0.5, 0.5, (n, 1))
    obs[rs.rand(n, L) < 0.3] = -9999.9
    f["tsurfobs"] = np.ascontiguousarray(obs)
    if rs.rand() < 0.4:   # output depth given per index for some points
        d = np.full((n, L), -9999.9)
        rows = rs.rand(n) < 0.4
        d[rows] = rs.choice([0.0, 0.02, 0.1, 1.0], (int(rows.sum()), 1))
        f["depth"] = d
    if sky:
        f["local_horizons"] = np.ascontiguousarray(rs.uniform(0, 30, (n, 360)))
    return f, s, p, ls


@pytest.mark.parametrize("seed", range(24))
def test_random_configuration_is_bit_identical(seed):
    L = lib.load()
    f, s, p, ls = _draw(seed)
    n, SL = f["tair"].shape
    ora, _, _ = oh.run_oracle("port", f, s, p, ls)
    g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
    out = {k: np.full((n, SL), np.nan) for k in oh.F64_OUT}
    ips = (abi.InputPointers * n)(); ops = (abi.OutputPointers * n)(); keep = []
    for pt in range(n):
        ip, op, kp = oh.point_pointers(g, pt, out)
        ips[pt], ops[pt] = ip, op
        keep.append(kp)
    larr = (abi.LocalParameters * n)(*ls)
    st = C.c_int32(99)
    L.runsimulation_batch(n, ops, ips, C.byref(s), C.byref(p), larr, C.byref(st))
    assert st.value == 0, lib.last_error()
    desc = (f"NL{s.NLayers} dt{s.DTSecs:g} L{SL} relax{s.use_relaxation} cpl{s.use_coupling}/{s.coupling_minutes} "
            f"force{s.force_tsurf} depth{s.tsurfOutputDepth:g} sky{'local_horizons' in f}")
    for k in oh.F64_OUT:
        bad = int((out[k] != ora[k]).sum())
        assert bad == 0, (desc, k, bad, float(np.nanmax(np.abs(out[k] - ora[k]))))
    assert (ora["tsurf"] > -100).mean() > 0.9, desc


@pytest.mark.parametrize("variant", [1, 2, 3, 4])
@pytest.mark.parametrize("seed", [1, 4, 9])
def test_kernel_flavours_agree_on_random_configurations(seed, variant, monkeypatch):
    """The same draws with every kernel flavour forced (ROADSURF_HIP_VARIANT): register profile,
    LDS profile, two wavefronts per 64 points, hybrid profile (where the draw's feature set admits the
    flavour; the other launches of such a plan run the default flavour)."""
    f, s, p, ls = _draw(seed)
    if s.NLayers != 15 and variant in (1, 3):
        pytest.skip("register flavours are built for NLayers = 15")
    monkeypatch.setenv("ROADSURF_HIP_VARIANT", str(variant))
    test_random_configuration_is_bit_identical(seed)
