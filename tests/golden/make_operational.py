#!/usr/bin/env python3
"""Generates tests/golden/e2e_operational.npz: the reference's own operational shape
(examples/example1/example_config.json:8-22 - 48 h analysis + 26 h forecast, SimLen 8 881, coupling and
relaxation on) on the stations of its two data files, run by THE REFERENCE (oracle/_ref/
libroadsurf_ref_cpl.so: the reference with `allocator`'s coupling dummy INTENT(INOUT), see
oracle/build_ref.sh) behind the C restatement of the driver's read_input (oracle/driver_oracle.c).

Stored: the NUMBERS of example_skyview.txt / example_local_horizons.txt (latitude, longitude, sky-view
factor, 360 horizon angles per station - data, not the files), a synthetic sky-view variant on the same
stations, and the reference's hourly outputs at the rows OPER_ROWS.  The raw series come from the seeded
scenario of tests/driver_helpers.py and are not stored.

    python tests/golden/make_operational.py        # needs /root/reference and oracle/_ref
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import driver_helpers as dh  # noqa: E402
import oracle_helpers as oh  # noqa: E402
from roadsurf_amd import driver  # noqa: E402

REF_EX = "/root/reference/examples/example1"
# hourly output rows kept: every third hour, the hours around the end of the analysis (row 48), the last
OPER_ROWS = sorted(set(range(0, 75, 3)) | {44, 45, 46, 47, 48, 49, 50, 51, 52, 74})


def read_station_files():
    ids, lat, lon, sv = [], [], [], []
    for line in open(os.path.join(REF_EX, "example_skyview.txt")):
        w = line.split()
        if len(w) < 5:
            continue
        ids.append(w[0]); lat.append(float(w[2])); lon.append(float(w[3])); sv.append(float(w[4]))
    hz = []
    for k, line in enumerate(open(os.path.join(REF_EX, "example_local_horizons.txt"))):
        w = line.split()
        if len(w) < 364:
            continue
        assert w[0] == ids[len(hz)]
        hz.append([float(x) for x in w[4:364]])
    return np.array(lat), np.array(lon), np.array(sv), np.array(hz)


def main():
    if not os.path.exists(oh.REF_CPL_SO):
        raise SystemExit("the reference build with working coupling (oracle/_ref) is needed")
    lat, lon, sv, hz = read_station_files()
    n = len(lat)
    assert hz.shape == (n, 360)
    tenths = np.rint(hz * 10.0)
    assert np.array_equal(tenths / 10.0, hz) and np.abs(tenths).max() < 30000
    rs = np.random.RandomState(dh.OPER_SEED)
    sv_sky = np.round(rs.uniform(0.35, 1.0, n), 3)
    sv_sky[::4] = 1.0  # a quarter of the stations without the sky-view branch
    hz_sky = np.rint(rs.uniform(0.0, 25.0, (n, 360)) * 10.0)
    save = dict(lat=lat, lon=lon, sky_view_files=sv, horizons_files_tenths=tenths.astype(np.int16),
                sky_view_sky=sv_sky, horizons_sky_tenths=hz_sky.astype(np.int16),
                rows=np.array(OPER_ROWS, np.int32))
    for case in ("files", "sky"):
        src, s, p, t0, tf, local, hzc = dh.operational_case(save, case)
        o = dh.oracle_run("ref_cpl", src, s, p, t0, tf, local=local, horizons=hzc)
        assert o["step"] == 120 and o["tsurf"].shape == (n, 75)
        for k in driver.OUT_FIELDS:
            save[f"{case}_{k}"] = np.ascontiguousarray(o[k][:, OPER_ROWS])
        save[f"{case}_status"] = o["status"]
        save[f"{case}_coupling_index"] = np.array([o["local"][q].couplingIndexI for q in range(n)], np.int32)
        save[f"{case}_initlen"] = np.array([o["local"][q].InitLenI for q in range(n)], np.int32)
        print(case, "accepted", int((o["status"] == 0).sum()), "of", n, "coupling indices",
              sorted(set(save[f"{case}_coupling_index"].tolist()))[-5:])
    np.savez_compressed(os.path.join(HERE, "e2e_operational.npz"), **save)
    print("wrote", os.path.join(HERE, "e2e_operational.npz"), os.path.getsize(os.path.join(HERE, "e2e_operational.npz")), "bytes")


if __name__ == "__main__":
    main()
