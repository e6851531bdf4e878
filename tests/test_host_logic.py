"""Host-side logic that needs no GPU: the Fortran constants builder against the
reference's own Initialization products (golden), sharding, argument validation."""
import ctypes as C

import numpy as np
import pytest

import golden_helpers as gh
import oracle_helpers as oh
from roadsurf_amd import abi, lib, sharding


@pytest.mark.parametrize("tag,nl,mod", [("nl15", 15, False), ("nl8", 8, False), ("nl32", 32, False),
                                        ("nl15mod", 15, True)])
def test_fortran_constants_equal_reference_initialization(tag, nl, mod):
    z = gh.load("init_products.npz")
    s = abi.default_settings(5761); s.NLayers = nl
    p = abi.default_parameters()
    if mod:
        p.RhoB1 = 1.9; p.Silt2 = 0.0; p.ZMom = 0.2; p.ZeroDisp = 0.5; p.Poro1 = 0.15; p.TClimG = 4.0
    c = lib.build_constants(s, p)
    assert np.array_equal(np.array(c.ZDpth)[1:nl + 2], z[f"{tag}_zdpth"][:nl + 1])
    assert np.array_equal(np.array(c.DyC)[1:nl + 1], z[f"{tag}_dyc"][:nl])
    assert np.array_equal(np.array(c.condDZ)[1:nl + 1], z[f"{tag}_conddz"][:nl])
    assert [c.logMom, c.logHeat, c.logCond, c.logUstar] == list(z[f"{tag}_logs"])
    # bottom boundary temperature = the reference's Tmp(NLayers+1) for 2024-01-10
    assert lib.bottom_temperature(p, c, 2024, 1, 10) == z[f"{tag}_tmp"][nl + 1]
    assert c.HSfac1 == z[f"{tag}_zdpth"][1] - z[f"{tag}_zdpth"][0]
    assert c.dryCap[1] == (np.float64(np.float32(1.0)) - p.Poro1) * p.vsh1
    assert c.WCont[1] == np.float64(np.float32(0.01)) and c.WCont[3] == np.float64(np.float32(0.3))


def test_wear_constants_fold_in_single_precision():
    c = lib.build_constants(abi.default_settings(10), abi.default_parameters())
    f = np.float32
    assert c.wSnowTran == float(f(0.2) + f(0.25))
    assert c.wSnow2Ice == float(f(0.25) / (f(0.2) + f(0.25)))
    assert c.wIce == float(f(1.1) * f(2.0) * f(0.145))
    assert c.wIce2 == float(f(1.1) * f(2.0) * (f(4.0) * f(0.290)))
    assert c.wDep == float(f(0.5) * f(2.0) * (f(4.0) * f(0.290)))
    assert c.wWat == float(f(0.145))
    assert c.Tph == 30.0 / 3600.0 and c.twoDT == 60.0


def test_leap_years_in_bottom_temperature():
    p = abi.default_parameters()
    c = lib.build_constants(abi.default_settings(10), p)
    z = c.ZDpth[16]

    def want(doy):
        return p.TClimG + p.AZ * np.sin(p.Omega * doy + p.Omega * (-170) - (z / p.DampDpth))

    assert lib.bottom_temperature(p, c, 2023, 3, 1) == want(60)
    assert lib.bottom_temperature(p, c, 2024, 3, 1) == want(61)
    assert lib.bottom_temperature(p, c, 1900, 3, 1) == want(60)
    assert lib.bottom_temperature(p, c, 2000, 12, 31) == want(366)


def test_constants_reject_bad_settings():
    p = abi.default_parameters()
    for nl in (4, 33):
        s = abi.default_settings(10); s.NLayers = nl
        with pytest.raises(ValueError):
            lib.build_constants(s, p)
    s = abi.default_settings(0)
    with pytest.raises(ValueError):
        lib.build_constants(s, p)


def test_shards_partition_the_points():
    for total, world in ((1_000_000, 8), (10, 3), (7, 8)):
        spans = [sharding.strong_shard(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == total
        for (o1, c1), (o2, _) in zip(spans[:-1], spans[1:]):
            assert o1 + c1 == o2
    assert sharding.weak_shard(1000, 3) == (3000, 1000)


def test_fortran_sun_table_against_oracle_solar_position():
    """rs_sun_table + rs_point_geometry (Fortran host) recombine to the oracle's elevation:
    cos(zenith) = sin(decl) sin(lat) + cos(decl) cos(lat) cos(stG + lon - ra)."""
    import oracle_helpers as oh
    L = lib.load()
    port = oh.load("port")
    port.oracle_probe_sun.argtypes = [C.c_int] * 6 + [C.c_double] * 2 + [abi.c_double_p] * 3
    rs = np.random.RandomState(0)
    n = 200
    y = np.full(n, 2024, np.int32); mo = rs.randint(1, 13, n).astype(np.int32)
    d = rs.randint(1, 29, n).astype(np.int32); h = rs.randint(0, 24, n).astype(np.int32)
    mi = rs.randint(0, 60, n).astype(np.int32); se = (30 * rs.randint(0, 2, n)).astype(np.int32)
    tab = np.zeros((n, 6))  # RS_SUN_COLS
    L.rs_sun_table(n, *[C.c_void_p(a.ctypes.data) for a in (y, mo, d, h, mi, se)], C.c_void_p(tab.ctypes.data))
    ls = []
    for k in range(n):
        l = abi.default_local(); l.lat = float(rs.uniform(-80, 80)); l.lon = float(rs.uniform(-180, 180)); ls.append(l)
    larr = (abi.LocalParameters * n)(*ls)
    g = [np.zeros(n) for _ in range(3)]
    L.rs_point_geometry(n, larr, *[C.c_void_p(a.ctypes.data) for a in g])
    for k in range(n):
        el, az, jde = C.c_double(), C.c_double(), C.c_double()
        assert port.oracle_probe_sun(int(y[k]), int(mo[k]), int(d[k]), int(h[k]), int(mi[k]), int(se[k]),
                                     ls[k].lat, ls[k].lon, C.byref(el), C.byref(az), C.byref(jde)) == 0
        ra, stg, sd, cd, ch, sh = tab[k]
        assert abs(ch - np.cos(stg - ra)) < 1e-15 and abs(sh - np.sin(stg - ra)) < 1e-15
        assert 0.0 <= ra <= 2 * np.pi + 1e-12
        cosz = sd * g[0][k] + (cd * g[1][k]) * np.cos((stg + g[2][k]) - ra)
        elev = 90.0 - np.degrees(np.arccos(np.clip(cosz, -1, 1)))
        if el.value > -9000:
            assert abs(elev - el.value) < 1e-9
        else:
            assert elev <= 1e-9


def test_uniform_reciprocal_division_is_ieee_division():
    """The kernels divide by uniform constants with a host-made reciprocal and two fused
    multiply-adds (rs_math.hpp, rs_div_u; constants in rs_consts_dev.h).  The same formula on the
    host (libm fma is correctly rounded like v_fma_f64) must give the IEEE quotient for every
    constant the kernels use, on numerators spanning the magnitudes the model produces."""
    ol = oh.load("port")
    ol.oracle_div_u_mismatches.restype = C.c_long
    ol.oracle_div_u_mismatches.argtypes = [C.c_double, C.c_void_p, C.c_long]
    s = abi.default_settings(5761); p = abi.default_parameters()
    c = lib.build_constants(s, p)
    dens = {"3600": 3600.0, "3364": 3364.0, "1000": 1000.0, "IceMax": 1.5, "twoDT": c.twoDT,
            "DTSecs": c.DTSecs, "meltDen": c.WatMHeat * c.WatDens, "logUstar": c.logUstar,
            "logCond": c.logCond}
    for dt in (10.0, 60.0, 7.0):  # other time steps, other roughness lengths
        s2 = abi.default_settings(100); s2.DTSecs = dt
        p2 = abi.default_parameters(); p2.ZMom = 0.03 * dt; p2.ZHeat = 0.0007 * dt
        c2 = lib.build_constants(s2, p2)
        dens.update({f"twoDT{dt}": c2.twoDT, f"DTSecs{dt}": c2.DTSecs, f"logUstar{dt}": c2.logUstar,
                     f"logCond{dt}": c2.logCond})
    # spans of the forcing interpolation (expand_kernel: steps per knot = 3600 / DTSecs and others)
    dens.update({f"spk{k}": float(k) for k in (2, 3, 7, 12, 60, 120, 240, 360, 3600)})
    rng = np.random.default_rng(7)
    n = 1_000_000
    mant = rng.uniform(1.0, 2.0, n)
    expo = rng.integers(-40, 41, n)
    a = np.ascontiguousarray(np.ldexp(mant, expo) * rng.choice([-1.0, 1.0], n))
    a[:3] = [0.0, 1.0, 3600.0]
    for name, b in dens.items():
        bad = ol.oracle_div_u_mismatches(b, a.ctypes.data, n)
        assert bad == 0, (name, b, bad)
    # the one operand the short sequences (rs_div_u and rs_div alike) do not reproduce: -0.0 / b
    # gives +0.0 where IEEE gives -0.0.  No call site can tell: every numerator is a product or sum
    # of non-negative quantities behind a `> 0` guard, except prec/3600, whose sign of zero is lost
    # in `<= MinPrecmm` and `wat + rain` (rs_math.hpp).
    mz = np.array([-0.0])
    assert ol.oracle_div_u_mismatches(3600.0, mz.ctypes.data, 1) == 1
