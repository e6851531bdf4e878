#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running THE REFERENCE ITSELF
(oracle/_ref/libroadsurf_ref.so, built from /root/reference by oracle/build_ref.sh,
amdflang -O2).  The reference ships no tests or golden outputs, so
these vectors are what pins the CPU restatement (oracle/roadsurf_oracle.c) and,
through it, the HIP path.  Only data is stored: inputs (hourly knots / parameters)
and the reference's outputs.

    python tests/golden/make_golden.py        # needs /root/reference
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import golden_helpers as gh  # noqa: E402
import oracle_helpers as oh  # noqa: E402
from roadsurf_amd import abi  # noqa: E402

SPK, HOURS = 120, 48
SIMLEN = HOURS * SPK + 1
NK = HOURS + 2
KEEP = 12  # outputs stored every KEEP-th index (+ the last one)


def scenario_knots():
    """Eight hand-built 48 h scenarios: each exercises a different part of
    the storage/energy-balance logic."""
    rs = np.random.RandomState(20240110)
    h = np.arange(NK, dtype=np.float64)
    hod = h % 24
    day = np.sin((hod - 9.0) / 24.0 * 2 * np.pi)
    sun = np.maximum(0.0, np.sin((hod - 6.0) / 12.0 * np.pi)) * (hod >= 6) * (hod <= 18)
    n = 8
    K = {k: np.zeros((n, NK)) for k in gh.KNOT_FIELDS}
    K["phase"] = np.full((n, NK), -9999, np.int32)
    noise = lambda s: s * (2 * rs.rand(NK) - 1)
    # 0 always cold, dry, windy
    K["tair"][0] = -15 + 3 * day + noise(0.5); K["rhz"][0] = 70 + noise(3); K["vz"][0] = 6 + noise(1)
    K["sw"][0] = 60 * sun; K["lw"][0] = 210 + noise(5)
    # 1 freeze-thaw crossing with every precipitation form
    K["tair"][1] = 0.5 + 4 * day + noise(0.3); K["rhz"][1] = 92 + noise(3); K["vz"][1] = 3 + noise(1)
    K["sw"][1] = 120 * sun; K["lw"][1] = 300 + noise(10)
    K["prec"][1, 4:30] = 1.5 + noise(0.5)[4:30]; K["phase"][1, 4:30] = np.arange(26) % 7
    # 2 warm and wet
    K["tair"][2] = 9 + 3 * day + noise(0.3); K["rhz"][2] = 95 + noise(2); K["vz"][2] = 4 + noise(1)
    K["sw"][2] = 200 * sun; K["lw"][2] = 340 + noise(5)
    K["prec"][2, 10:20] = 2.5; K["phase"][2, 10:20] = 1
    # 3 calm clear night: wind below both CalmLim values, strong long-wave loss, dew/deposit
    K["tair"][3] = -3 + 5 * day + noise(0.2); K["rhz"][3] = 97 + noise(1); K["vz"][3] = 0.05 + 0.1 * rs.rand(NK)
    K["sw"][3] = 150 * sun; K["lw"][3] = 205 + noise(3)
    # 4 strongly unstable: high sun, weak wind, surface much warmer than air
    K["tair"][4] = 2 + 2 * day + noise(0.2); K["rhz"][4] = 60 + noise(3); K["vz"][4] = 0.6 + 0.3 * rs.rand(NK)
    K["sw"][4] = 400 * sun; K["lw"][4] = 260 + noise(5)
    # 5 a point whose input goes out of range at hour 30 (simulation_failed)
    K["tair"][5] = -5 + 3 * day; K["rhz"][5] = 80; K["vz"][5] = 3; K["sw"][5] = 100 * sun; K["lw"][5] = 250
    K["rhz"][5, 30:] = 130.0
    # 6 precipitation with missing phase around 0 C at high humidity: in-built interpretation
    K["tair"][6] = 0.8 + 1.5 * day + noise(0.4); K["rhz"][6] = 96 + noise(2); K["vz"][6] = 2.5 + noise(0.5)
    K["sw"][6] = 50 * sun; K["lw"][6] = 305 + noise(5)
    K["prec"][6, 2:40] = 1.0 + rs.rand(38)
    # 7 heavy snowfall, then a warm spell melts it
    K["tair"][7] = np.where(h < 24, -4 + noise(0.5), -4 + (h - 24) * 0.6); K["rhz"][7] = 90 + noise(3)
    K["vz"][7] = 4 + noise(1); K["sw"][7] = 180 * sun; K["lw"][7] = 290 + (h >= 24) * 40.0
    K["prec"][7, 1:14] = 3.0; K["phase"][7, 1:14] = 3
    K["prec"][7, 14:16] = 1.0; K["phase"][7, 14:16] = 6
    K["vz"] = np.maximum(K["vz"], 0.01)
    K["rhz"] = np.clip(K["rhz"], 5, None)
    K["tdew"] = K["tair"] - 2.0
    K["tsurf0"] = K["tair"][:, 0] - 0.5
    return K


def thin(a):
    idx = np.unique(np.r_[np.arange(0, a.shape[1], KEEP), a.shape[1] - 1])
    return idx, a[:, idx]


def main():
    if not oh.have_ref():
        raise SystemExit("the reference build (oracle/_ref) is needed")
    p = abi.default_parameters()
    # ---- (i) end-to-end scenarios ------------------------------------------------
    K = scenario_knots()
    f = gh.expand_knots(K, SIMLEN, SPK)
    s = abi.default_settings(SIMLEN)
    l = abi.default_local(); l.InitLenI = 1
    out, _, _ = oh.run_oracle("ref", f, s, p, l)
    save = {f"knot_{k}": v for k, v in K.items()}
    for k in oh.F64_OUT:
        idx, save[f"out_{k}"] = thin(out[k])
    save["out_index"] = idx
    np.savez_compressed(os.path.join(HERE, "e2e_scenarios.npz"), **save)
    # ---- (i') optional features on a shorter run ---------------------------------
    L2 = 12 * SPK + 1
    K2 = {k: (v[:6, :14].copy() if v.ndim == 2 else v[:6].copy()) for k, v in K.items()}
    f2 = gh.expand_knots(K2, L2, SPK)
    f2["tsurfobs"][:, :360] = f2["tair"][:, :360] - 0.7
    f2["tsurfobs"][::2, 100:150] = -9999.9
    cases = {}
    ls = []
    for i in range(6):
        li = abi.default_local(); li.InitLenI = 360
        li.tair_relax = float(f2["tair"][i, 360]) + 1.5; li.VZ_relax = 3.0; li.RH_relax = 85.0
        if i == 5:
            li.tair_relax = -9999.0
        ls.append(li)
    s2 = abi.default_settings(L2); s2.use_relaxation = 1
    cases["relax"] = oh.run_oracle("ref", f2, s2, p, ls)[0]
    s2 = abi.default_settings(L2); s2.tsurfOutputDepth = 0.05
    cases["depthset"] = oh.run_oracle("ref", f2, s2, p, ls)[0]
    f3 = {k: v.copy() for k, v in f2.items()}
    f3["depth"][:] = 0.0; f3["depth"][::2] = 0.12; f3["depth"][1::4] = 7.0
    cases["deptharr"] = oh.run_oracle("ref", f3, abi.default_settings(L2), p, ls)[0]
    s2 = abi.default_settings(L2); s2.force_tsurf = 1
    cases["force"] = oh.run_oracle("ref", f2, s2, p, ls)[0]
    s2 = abi.default_settings(L2); s2.NLayers = 9
    cases["nl9"] = oh.run_oracle("ref", f2, s2, p, ls)[0]
    save = {f"knot_{k}": v for k, v in K2.items()}
    save["tair_relax"] = np.array([x.tair_relax for x in ls])
    for c, o in cases.items():
        for k in oh.F64_OUT:
            idx, save[f"{c}_{k}"] = thin(o[k])
    save["out_index"] = idx
    np.savez_compressed(os.path.join(HERE, "e2e_features.npz"), **save)
    # ---- (i'') coupling: from the reference built with working coupling -------------
    # (oracle/build_ref.sh: the strict amdflang build wipes the observation in `allocator`)
    L3 = 24 * SPK + 1
    K3 = {k: (v[:, :26].copy() if v.ndim == 2 else v.copy()) for k, v in K.items()}
    f3 = gh.expand_knots(K3, L3, SPK)
    base3 = oh.run_oracle("ref", f3, abi.default_settings(L3), p, l)[0]
    f3["tsurfobs"][:, :] = np.where(base3["tsurf"] == -9999.0, -9999.9, base3["tsurf"] + 0.3)
    offs = np.array([0.0, 0.5, -2.0, 6.0, -6.0, 15.0, 0.05, -0.5])
    ls3 = []
    for i in range(8):
        li = abi.default_local(); li.InitLenI = 1440; li.couplingIndexI = 1440
        li.couplingTsurf = float(base3["tsurf"][i, 1439] + offs[i])
        li.tair_relax = float(f3["tair"][i, 1440]) + 1.0; li.VZ_relax = 3.0; li.RH_relax = 80.0
        ls3.append(li)
    ls3[4].couplingTsurf = -9999.0  # a point without a usable observation
    s3 = abi.default_settings(L3); s3.use_coupling = 1; s3.use_relaxation = 1
    o3 = oh.run_oracle("ref_cpl", f3, s3, p, ls3)[0]
    save = {f"knot_{k}": v for k, v in K3.items()}
    save["coupling_tsurf"] = np.array([x.couplingTsurf for x in ls3])
    save["tair_relax"] = np.array([x.tair_relax for x in ls3])
    save["tsurfobs"] = f3["tsurfobs"][:, ::KEEP].copy()
    save["base_tsurf_full"] = base3["tsurf"]
    for k in oh.F64_OUT:
        idx, save[f"cpl_{k}"] = thin(o3[k])
    save["out_index"] = idx
    np.savez_compressed(os.path.join(HERE, "e2e_coupling.npz"), **save)
    # ---- (i-sky) sky view / local horizons (src/ModRadiation.f90, src/SunPosition.f90) ---
    L4 = 24 * SPK + 1
    K4 = {k: (v[:, :26].copy() if v.ndim == 2 else v.copy()) for k, v in K.items()}
    K4["sw"] = K4["sw"] * 2.5
    f4 = gh.expand_knots(K4, L4, SPK, start=(2024, 5, 15, 0, 0, 0))
    rs4 = np.random.RandomState(11)
    fac4 = np.array([0.2, 0.5, 0.8, 1.1, 0.6, 0.9, 0.4, 0.7])
    f4["sw_dir"] = np.ascontiguousarray(f4["sw"] * fac4[:, None])
    f4["lw_net"] = np.full((8, L4), -55.0)
    hz4 = np.round(rs4.uniform(0, 30, (8, 360)), 1); hz4[0] = 0.0
    f4["local_horizons"] = np.ascontiguousarray(hz4)
    lat4 = np.array([60.43, 69.9, 61.0, -33.9, 0.5, 45.0, 64.2, 59.9])
    lon4 = np.array([22.84, 27.0, 25.7, 151.2, -78.5, 7.6, 29.1, 10.7])
    sv4 = np.array([0.99, 0.75, 0.3, 0.5, 0.0, 1.0, 0.6, 0.85])
    ls4 = []
    for i in range(8):
        li = abi.default_local(); li.InitLenI = 1; li.lat = float(lat4[i]); li.lon = float(lon4[i])
        li.sky_view = float(sv4[i]); ls4.append(li)
    o4, fm4, _ = oh.run_oracle("ref", f4, abi.default_settings(L4), p, ls4)
    save = {f"knot_{k}": v for k, v in K4.items()}
    save.update(lat=lat4, lon=lon4, sky_view=sv4, horizons=hz4, sw_dir_factor=fac4)
    for k in oh.F64_OUT:
        idx, save[f"sky_{k}"] = thin(o4[k])
    save["out_index"] = idx
    save["sw_after"] = fm4["sw"][:, ::KEEP].copy()  # the reference's in-place edit of the input
    np.savez_compressed(os.path.join(HERE, "e2e_skyview.npz"), **save)
    # ---- (iii) init products ------------------------------------------------------
    ref = oh.load("ref")
    save = {}
    for tag, nl, mod in (("nl15", 15, False), ("nl8", 8, False), ("nl32", 32, False), ("nl15mod", 15, True)):
        s3 = abi.default_settings(SIMLEN); s3.NLayers = nl
        p3 = abi.default_parameters()
        if mod:
            p3.RhoB1 = 1.9; p3.Silt2 = 0.0; p3.ZMom = 0.2; p3.ZeroDisp = 0.5; p3.Poro1 = 0.15; p3.TClimG = 4.0
        ip, op, keep = oh.point_pointers(f, 1)
        arrs = [np.zeros(nl + 2) for _ in range(6)]
        logs = np.zeros(4); ts = C.c_double()
        ref.ref_probe_init(C.byref(ip), C.byref(op), C.byref(s3), C.byref(p3), C.byref(l),
                           *[a.ctypes.data_as(abi.c_double_p) for a in arrs],
                           logs.ctypes.data_as(abi.c_double_p), C.byref(ts))
        for nm, a in zip(("zdpth", "dyc", "dyk", "cc", "conddz", "tmp"), arrs):
            save[f"{tag}_{nm}"] = a
        save[f"{tag}_logs"] = logs
        save[f"{tag}_tsurf"] = np.array([ts.value])
    np.savez_compressed(os.path.join(HERE, "init_products.npz"), **save)
    # ---- (ii) CalcBLCondAndLE known answers ----------------------------------------
    rs = np.random.RandomState(7)
    m = 400
    tsurf = rs.uniform(-25, 30, m); tair = tsurf + rs.uniform(-12, 12, m)
    vz = np.r_[rs.uniform(0.4, 15, m - 40), rs.uniform(0.4, 0.6, 40)]
    rh = rs.uniform(20, 105, m); wat = np.where(rs.rand(m) < 0.5, 0.0, rs.uniform(0, 2, m))
    res = np.zeros((m, 3))
    ref.ref_probe_blcond.argtypes = [C.POINTER(abi.InputParameters)] + [C.c_double] * 6 + [abi.c_double_p] * 3
    for i in range(m):
        b, le, ev = C.c_double(), C.c_double(), C.c_double()
        ref.ref_probe_blcond(C.byref(p), 30.0, tsurf[i], tair[i], vz[i], rh[i], wat[i],
                             C.byref(b), C.byref(le), C.byref(ev))
        res[i] = b.value, le.value, ev.value
    np.savez_compressed(os.path.join(HERE, "blcond_known_answers.npz"), tsurf=tsurf, tair=tair, vz=vz,
                        rh=rh, wat=wat, blcond=res[:, 0], le=res[:, 1], evap=res[:, 2])
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith(".npz"):
            print(fn, os.path.getsize(os.path.join(HERE, fn)), "bytes")


if __name__ == "__main__":
    main()
