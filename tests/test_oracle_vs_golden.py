"""The CPU restatement (oracle/roadsurf_oracle.c) against fixtures captured from the
REFERENCE ITSELF (tests/golden/make_golden.py).  Bit-for-bit: on x86-64 both sides do
the same IEEE operations in the same order and call the same glibc exp/log."""
import ctypes as C
import os

import numpy as np
import pytest

import golden_helpers as gh
import oracle_helpers as oh
from roadsurf_amd import abi

SPK = 120


def _knots(z, n=None):
    return {k[5:]: z[k] for k in z.files if k.startswith("knot_")}


def test_scenarios_48h_bit_exact():
    z = gh.load("e2e_scenarios.npz")
    K = _knots(z)
    L = 48 * SPK + 1
    f = gh.expand_knots(K, L, SPK)
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    out, _, _ = oh.run_oracle("port", f, s, p, l)
    idx = z["out_index"]
    for k in oh.F64_OUT:
        assert np.array_equal(out[k][:, idx], z[f"out_{k}"]), k
    # the scenarios do what they were built for
    assert (z["out_tsurf"][5] == -9999.0).any() and not (z["out_tsurf"][0] == -9999.0).any()
    assert z["out_snow"][7].max() > 5 and z["out_ice"][1].max() > 0.5 and z["out_deposit"][3].max() > 0.05


def test_feature_cases_bit_exact():
    z = gh.load("e2e_features.npz")
    K = _knots(z)
    L = 12 * SPK + 1
    f2 = gh.expand_knots(K, L, SPK)
    f2["tsurfobs"][:, :360] = f2["tair"][:, :360] - 0.7
    f2["tsurfobs"][::2, 100:150] = -9999.9
    p = abi.default_parameters()
    ls = []
    for i in range(6):
        li = abi.default_local(); li.InitLenI = 360
        li.tair_relax = float(z["tair_relax"][i]); li.VZ_relax = 3.0; li.RH_relax = 85.0
        ls.append(li)
    idx = z["out_index"]

    def check(tag, f, s):
        out, _, _ = oh.run_oracle("port", f, s, p, ls)
        for k in oh.F64_OUT:
            assert np.array_equal(out[k][:, idx], z[f"{tag}_{k}"]), (tag, k)

    s = abi.default_settings(L); s.use_relaxation = 1
    check("relax", f2, s)
    s = abi.default_settings(L); s.tsurfOutputDepth = 0.05
    check("depthset", f2, s)
    f3 = {k: v.copy() for k, v in f2.items()}
    f3["depth"][:] = 0.0; f3["depth"][::2] = 0.12; f3["depth"][1::4] = 7.0
    check("deptharr", f3, abi.default_settings(L))
    s = abi.default_settings(L); s.force_tsurf = 1
    check("force", f2, s)
    s = abi.default_settings(L); s.NLayers = 9
    check("nl9", f2, s)


def _coupling_case():
    z = gh.load("e2e_coupling.npz")
    K = _knots(z)
    L = 24 * SPK + 1
    f = gh.expand_knots(K, L, SPK)
    base = z["base_tsurf_full"]
    f["tsurfobs"][:, :] = np.where(base == -9999.0, -9999.9, base + 0.3)
    ls = []
    for i in range(8):
        li = abi.default_local(); li.InitLenI = 1440; li.couplingIndexI = 1440
        li.couplingTsurf = float(z["coupling_tsurf"][i])
        li.tair_relax = float(z["tair_relax"][i]); li.VZ_relax = 3.0; li.RH_relax = 80.0
        ls.append(li)
    s = abi.default_settings(L); s.use_coupling = 1; s.use_relaxation = 1
    return z, f, s, abi.default_parameters(), ls


def test_coupling_bit_exact():
    """Fixture from the reference built with working coupling (see oracle/build_ref.sh)."""
    z, f, s, p, ls = _coupling_case()
    out, _, _ = oh.run_oracle("port", f, s, p, ls)
    idx = z["out_index"]
    for k in oh.F64_OUT:
        assert np.array_equal(out[k][:, idx], z[f"cpl_{k}"]), k
    # coupling pulled the simulated temperature at the observation index onto the observation
    got = out["tsurf"][:, 1439]; obs = z["coupling_tsurf"]
    ok = [i for i in range(8) if obs[i] > -100 and abs(obs[i] - z["base_tsurf_full"][i, 1439]) < 7 and got[i] != -9999.0]
    assert len(ok) >= 4 and all(abs(got[i] - obs[i]) <= 0.1 + 1e-9 for i in ok)


def _skyview_case():
    z = gh.load("e2e_skyview.npz")
    K = _knots(z)
    L = 24 * SPK + 1
    f = gh.expand_knots(K, L, SPK, start=(2024, 5, 15, 0, 0, 0))
    f["sw_dir"] = np.ascontiguousarray(f["sw"] * z["sw_dir_factor"][:, None])
    f["lw_net"] = np.full((8, L), -55.0)
    f["local_horizons"] = np.ascontiguousarray(z["horizons"])
    ls = []
    for i in range(8):
        li = abi.default_local(); li.InitLenI = 1; li.lat = float(z["lat"][i]); li.lon = float(z["lon"][i])
        li.sky_view = float(z["sky_view"][i]); ls.append(li)
    return z, f, abi.default_settings(L), abi.default_parameters(), ls


def test_skyview_bit_exact():
    z, f, s, p, ls = _skyview_case()
    out, fm, _ = oh.run_oracle("port", f, s, p, ls)
    idx = z["out_index"]
    for k in oh.F64_OUT:
        assert np.array_equal(out[k][:, idx], z[f"sky_{k}"]), k
    assert np.array_equal(fm["sw"][:, ::12], z["sw_after"])  # same in-place edit of SW
    assert (fm["sw"] != f["sw"]).mean() > 0.05


def test_init_products_bit_exact():
    z = gh.load("init_products.npz")
    zs = gh.load("e2e_scenarios.npz")
    f = gh.expand_knots(_knots(zs), 48 * SPK + 1, SPK)
    port = oh.load("port")
    l = abi.default_local(); l.InitLenI = 1
    for tag, nl, mod in (("nl15", 15, False), ("nl8", 8, False), ("nl32", 32, False), ("nl15mod", 15, True)):
        s = abi.default_settings(48 * SPK + 1); s.NLayers = nl
        p = abi.default_parameters()
        if mod:
            p.RhoB1 = 1.9; p.Silt2 = 0.0; p.ZMom = 0.2; p.ZeroDisp = 0.5; p.Poro1 = 0.15; p.TClimG = 4.0
        ip, op, keep = oh.point_pointers(f, 1)
        arrs = [np.zeros(nl + 2) for _ in range(6)]
        logs = np.zeros(4); ts = C.c_double()
        port.oracle_probe_init(C.byref(ip), C.byref(op), C.byref(s), C.byref(p), C.byref(l),
                               *[a.ctypes.data_as(abi.c_double_p) for a in arrs],
                               logs.ctypes.data_as(abi.c_double_p), C.byref(ts))
        for nm, a in zip(("zdpth", "dyc", "dyk", "cc", "conddz", "tmp"), arrs):
            assert np.array_equal(a, z[f"{tag}_{nm}"]), (tag, nm)
        assert np.array_equal(logs, z[f"{tag}_logs"]) and ts.value == z[f"{tag}_tsurf"][0]
    # the oracle's layer grid for NLayers = 15
    survey = [0, 3.02999997511506081e-02, 6.47199992090463638e-02, 1.04907998815178871e-01,
              1.53171198442578316e-01, 2.12739674374461174e-01, 2.88135541602969170e-01,
              3.85689759626984596e-01, 5.14265662059187889e-01, 6.86271915212273598e-01,
              9.19080657884478569e-01, 1.23701289482414722e+00, 1.67411803267896175e+00,
              2.27806521393358707e+00, 3.11559125594794750e+00, 4.28012775070965290e+00]
    assert list(z["nl15_zdpth"][:16]) == survey


def test_blcond_known_answers_bit_exact():
    z = gh.load("blcond_known_answers.npz")
    port = oh.load("port")
    port.oracle_probe_blcond.argtypes = [C.POINTER(abi.InputSettings), C.POINTER(abi.InputParameters)] + \
        [C.c_double] * 5 + [abi.c_double_p] * 3 + [abi.c_int32_p]
    s = abi.default_settings(100); p = abi.default_parameters()
    iters = []
    for i in range(len(z["tsurf"])):
        b, le, ev, it = C.c_double(), C.c_double(), C.c_double(), C.c_int32()
        port.oracle_probe_blcond(C.byref(s), C.byref(p), z["tsurf"][i], z["tair"][i], z["vz"][i],
                                 z["rh"][i], z["wat"][i], C.byref(b), C.byref(le), C.byref(ev), C.byref(it))
        assert (b.value, le.value, ev.value) == (z["blcond"][i], z["le"][i], z["evap"][i]), i
        iters.append(it.value)
    # both regimes of the fixed-point iteration are covered: exit at the minimum 5 and later
    assert min(iters) == 5 and max(iters) > 6


@pytest.mark.skipif(not oh.have_ref(), reason="reference build not available")
def test_port_equals_reference_on_random_workload():
    n, L = 400, 2881
    f = oh.synth_forcing(n, L, seed=31337)
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    a, fa, _ = oh.run_oracle("ref", f, s, p, l)
    b, fb, _ = oh.run_oracle("port", f, s, p, l)
    for k in oh.F64_OUT:
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(fa["vz"], fb["vz"])  # VZ(1) side effect


@pytest.mark.skipif(not oh.have_ref(), reason="reference build not available")
def test_port_equals_reference_on_random_parameter_sets():
    """the restatement away from the reference's defaults: every physical parameter the path reads,
    the time step, the layer count, the output depth, initialization length and relaxation drawn at
    random (the cases of tests/test_hip_param_fuzz.py, which the HIP path must reproduce)"""
    from test_hip_param_fuzz import _case
    for seed in range(16):
        f, s, p, ls = _case(seed)
        a, _, _ = oh.run_oracle("ref", f, s, p, ls)
        b, _, _ = oh.run_oracle("port", f, s, p, ls)
        for k in oh.F64_OUT:
            assert np.array_equal(a[k], b[k]), (seed, k)


@pytest.mark.skipif(not os.path.exists(oh.REF_CPL_SO), reason="coupling-enabled reference build not available")
def test_port_equals_reference_with_coupling():
    n, L = 300, 2881
    f = oh.synth_forcing(n, L, seed=4242)
    p = abi.default_parameters(); l0 = abi.default_local(); l0.InitLenI = 1
    base, _, _ = oh.run_oracle("port", f, abi.default_settings(L), p, l0)
    rs = np.random.RandomState(3)
    ls = []
    for i in range(n):
        li = abi.default_local(); ci = int(rs.randint(50, L - 60))
        li.InitLenI = ci; li.couplingIndexI = ci
        li.couplingTsurf = float(base["tsurf"][i, ci - 1] + rs.choice([0.0, 0.5, -2.0, 6.0, -15.0]))
        li.tair_relax = float(f["tair"][i, ci]) + 1.0; li.VZ_relax = 3.0; li.RH_relax = 80.0
        ls.append(li)
    f["tsurfobs"][:, :] = base["tsurf"] + 0.3
    s = abi.default_settings(L); s.use_coupling = 1; s.use_relaxation = 1
    a, _, _ = oh.run_oracle("ref_cpl", f, s, p, ls)
    b, _, _ = oh.run_oracle("port", f, s, p, ls)
    for k in oh.F64_OUT:
        assert np.array_equal(a[k], b[k]), k
    # and the two reference builds agree wherever coupling is off
    s2 = abi.default_settings(L); s2.use_relaxation = 1
    c, _, _ = oh.run_oracle("ref", f, s2, p, ls)
    d, _, _ = oh.run_oracle("ref_cpl", f, s2, p, ls)
    for k in oh.F64_OUT:
        assert np.array_equal(c[k], d[k]), k


@pytest.mark.parametrize("case", ["files", "sky"])
def test_operational_shape_bit_exact(case):
    """The reference's operational shape (examples/example1/example_config.json:8-22: SimLen 8 881,
    coupling + relaxation, the stations of its data files): the C restatement behind the restated
    read_input equals the fixture the reference itself produced (tests/golden/make_operational.py)."""
    import driver_helpers as dh
    from roadsurf_amd import driver
    z = gh.load("e2e_operational.npz")
    src, s, p, t0, tf, local, hz = dh.operational_case(z, case)
    o = dh.oracle_run("port", src, s, p, t0, tf, local=local, horizons=hz)
    assert np.array_equal(o["status"], z[f"{case}_status"])
    rows = z["rows"]
    for k in driver.OUT_FIELDS:
        assert np.array_equal(o[k][:, rows], z[f"{case}_{k}"]), (case, k)
    if case == "files":  # the example's own sky-view file switches the branch off everywhere
        assert (z["sky_view_files"] == 1.0).all() and (z["horizons_files_tenths"] == 0).all()
