"""GPU: the reference's own operational shape (examples/example1/example_config.json:8-22): 48 h analysis +
26 h forecast = SimLen 8 881, coupling and relaxation on, initialization and coupling window ending with
the analysis, the stations of example_skyview.txt / example_local_horizons.txt (their numbers are in the
fixture tests/golden/e2e_operational.npz, made by tests/golden/make_operational.py with the reference
itself).  Two cases: sky view and horizons as the files hold them (1.0 / 0.0 everywhere: the sky-view
branch is never taken in the reference's example) and synthetic ones on the same stations.  Both
boundaries must give the reference's bits: rs_driver_run from the raw series, and runsimulation_batch
(the reference's own boundary: step-resolution host arrays per point)."""
import ctypes as C

import numpy as np
import pytest

import driver_helpers as dh
import golden_helpers as gh
import oracle_helpers as oh
from roadsurf_amd import abi, driver, lib
from test_hip_boundary import _pointers

pytestmark = pytest.mark.gpu


def _bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.int64), np.ascontiguousarray(b).view(np.int64))


@pytest.mark.parametrize("case", ["files", "sky"])
def test_operational_shape_through_rs_driver_run(case, monkeypatch):
    monkeypatch.delenv("ROADSURF_HIP_CLUSTER", raising=False)  # the library's own default for a block of 401 points:
    #                                                            natural order, launches of eight hours
    z = gh.load("e2e_operational.npz")
    src, s, p, t0, tf, local, hz = dh.operational_case(z, case)
    assert s.SimLen == 8881
    g = driver.run(src, s, p, t0, tf, local=local, horizons=hz)
    rows = z["rows"]
    assert g["step"] == 120 and g["tsurf"].shape[1] == 75
    assert np.array_equal(g["status"], z[f"{case}_status"])
    n = len(z["lat"])
    assert np.array_equal(np.array([g["local"][q].couplingIndexI for q in range(n)]), z[f"{case}_coupling_index"])
    assert np.array_equal(np.array([g["local"][q].InitLenI for q in range(n)]), z[f"{case}_initlen"])
    for k in driver.OUT_FIELDS:
        assert _bits(g[k][:, rows], z[f"{case}_{k}"]), (case, k, float(np.abs(g[k][:, rows] - z[f"{case}_{k}"]).max()))
    ok = g["status"] == 0
    assert ok.sum() > 350 and (g["tsurf"][ok] > -100).all()
    # the window ends with the analysis (index 5 761 for a station that reported to the end) and coupling acts
    assert int(z[f"{case}_coupling_index"].max()) >= 5700
    s0 = abi.default_settings(s.SimLen); s0.use_relaxation = 1
    base = driver.run(src, s0, p, t0, tf, local=local, horizons=hz)
    moved = (np.abs(base["tsurf"] - g["tsurf"]).max(1) > 1e-3)[ok].sum()
    assert moved > ok.sum() // 2
    if case == "sky":  # ... and so does the sky view
        flat = dh.operational_case(z, "files")
        g0 = driver.run(src, s, p, t0, tf, local=flat[5], horizons=flat[6])
        assert (np.abs(g0["tsurf"] - g["tsurf"]).max(1) > 1e-3)[ok].sum() > ok.sum() // 3


@pytest.mark.parametrize("case", ["files", "sky"])
def test_operational_shape_through_runsimulation_batch(case):
    """The reference's own boundary: step-resolution arrays per point (made by the C restatement of the
    driver's read_input), runsimulation_batch_ex -> hourly rows equal the reference's."""
    z = gh.load("e2e_operational.npz")
    src, s, p, t0, tf, local, hz = dh.operational_case(z, case)
    ri = dh.oracle_read_input(src, s, t0, tf, local)
    L = s.SimLen
    ok = np.nonzero(ri["status"] == 0)[0]
    n = len(ok)
    f = {k: np.ascontiguousarray(ri["merged"][k][ok]) for k in driver.MERGED_FIELDS}
    f["depth"] = np.full((n, L), -9999.9)
    f["precphase"] = np.full((n, L), -9999, np.int32)
    f.update({k: np.ascontiguousarray(v, np.int32) for k, v in driver.calendar(t0, L, 30).items()})
    f["local_horizons"] = np.ascontiguousarray(hz[ok])
    out = {k: np.full((n, L), np.nan) for k in oh.F64_OUT}
    ips = (abi.InputPointers * n)(); ops = (abi.OutputPointers * n)(); keep = []
    for pt in range(n):
        ip, op, hk = _pointers(f, out, pt)
        hzrow = np.ascontiguousarray(f["local_horizons"][pt])  # (_pointers hands over a row of zeros)
        ip.c_local_horizons = hzrow.ctypes.data_as(abi.c_double_p)
        ips[pt], ops[pt] = ip, op
        keep.append((hk, hzrow))
    larr = (abi.LocalParameters * n)(*[ri["local"][int(q)] for q in ok])
    st = C.c_int32(99)
    ff = np.full(n, -7, np.int32)
    lib.load().runsimulation_batch_ex(n, ops, ips, C.byref(s), C.byref(p), larr, C.byref(st),
                                      ff.ctypes.data_as(abi.c_int32_p))
    assert st.value == 0, lib.last_error()
    rows = z["rows"].astype(np.int64) * 120
    for k in oh.F64_OUT:
        assert _bits(out[k][:, rows], z[f"{case}_{k}"][ok]), (case, k)
