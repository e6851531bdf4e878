"""GPU: the reference's run-time messages beyond CheckValues (VERDICT r05 "missing" 4) - CalcBLCondAndLE's
" ERROR : UStar negative" and " Max number of BLCond iterations" (src/BoundaryLayer.f90:69-74,98-101) and
Coupling_control's "coupling coefficient too small / too big, coupling failed" (src/Coupling.f90:400-401,451-452) -
with ROADSURF_HIP_DIAGNOSTICS=1, against what the reference itself prints on the same inputs (its unit 6, captured)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_helpers as oh
from roadsurf_amd import abi, lib

pytestmark = pytest.mark.gpu


def _product(f, s, p, ls, tmp_path, monkeypatch, name):
    """runsimulation_batch with diagnostics on: outputs and what it printed."""
    L = lib.load()
    n, SL = f["tair"].shape
    g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
    out = {k: np.full((n, SL), np.nan) for k in oh.F64_OUT}
    ips = (abi.InputPointers * n)(); ops = (abi.OutputPointers * n)(); keep = []
    for pt in range(n):
        ip, op, kp = oh.point_pointers(g, pt, out)
        ips[pt], ops[pt] = ip, op
        keep.append(kp)
    larr = (abi.LocalParameters * n)(*ls)
    st = C.c_int32(99)
    monkeypatch.setenv("ROADSURF_HIP_DIAGNOSTICS", "1")
    cap = oh.capture_stdout(str(tmp_path / name))
    with cap:
        L.runsimulation_batch(n, ops, ips, C.byref(s), C.byref(p), larr, C.byref(st))
    assert st.value == 0, lib.last_error()
    return out, cap.text().splitlines()


def _reference(kind, f, s, p, ls, tmp_path, monkeypatch):
    """The reference one point at a time (its messages carry no point): outputs and each point's lines."""
    monkeypatch.setenv("ORACLE_VERBOSE", "1")  # run_oracle: do not silence unit 6
    n = f["tair"].shape[0]
    outs, lines = [], []
    for i in range(n):
        fi = {k: (v[i:i + 1].copy() if isinstance(v, np.ndarray) and v.ndim == 2 else v) for k, v in f.items()}
        cap = oh.capture_stdout(str(tmp_path / "ref_one.txt"))
        with cap:
            o, _, _ = oh.run_oracle(kind, fi, s, p, ls[i], nthreads=1)
        outs.append(o)
        lines.append(cap.text().splitlines())
    return {k: np.concatenate([o[k] for o in outs]) for k in oh.F64_OUT}, lines


def _by_point(lines):
    """The product's lines -> {point: [(message lines, tag line)]} (a tag line closes every message)."""
    got, cur = {}, []
    for ln in lines:
        m = re.match(r" \(roadsurf_hip: point (\d+)", ln)
        if m:
            got.setdefault(int(m.group(1)) - 1, []).append((cur, ln))
            cur = []
        elif ln.strip():
            cur.append(ln)
    assert not cur, cur
    return got


@pytest.mark.skipif(not oh.have_ref(), reason="needs the compiled reference (oracle/_ref)")
@pytest.mark.parametrize("mode", ["plain", "skyview"])
def test_boundary_layer_messages_as_the_reference_prints_them(mode, tmp_path, monkeypatch):
    """plain: step_kernel_lds<true, true>; skyview: step_kernel_sky<true> (per-point sky view on half the points)."""
    n, SL = 6, 61
    f = oh.synth_forcing(n, SL, seed=5)
    f["prec"][:] = 0.0
    f["vz"][:] = 0.02       # calm far below the default limits: the loop's denominators change sign
    for q in range(n):
        f["tair"][q, :] = -20.0 + q
        f["tdew"][q, :] = f["tair"][q, :] - 2.0
        f["tsurfobs"][q, 0] = f["tair"][q, 0] + 2.0 + 3 * q
    f["vz"][n - 1, :] = 4.0  # ... and one ordinary point: no message
    s = abi.default_settings(SL); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    p.CalmLimDay = 0.01; p.CalmLimNgt = 0.01
    ls = [l] * n
    if mode == "skyview":
        ls = []
        for q in range(n):
            lq = abi.default_local(); lq.InitLenI = 1
            lq.lat, lq.lon = 60.0 + q, 25.0
            if q % 2 == 0:
                lq.sky_view = 0.6
            ls.append(lq)
        f["sw_dir"] = np.ascontiguousarray(0.5 * f["sw"])
        f["lw_net"] = np.full_like(f["lw"], -40.0)
        f["local_horizons"] = np.zeros((n, 360))
    ref_out, ref_lines = _reference("ref", f, s, p, ls, tmp_path, monkeypatch)
    out, lines = _product(f, s, p, ls, tmp_path, monkeypatch, "hip_bl.txt")
    for k in oh.F64_OUT:  # the kernels that carry the diagnostics return the reference's bits too
        assert np.array_equal(out[k], ref_out[k]), k
    got = _by_point(lines)
    seen = 0
    for q in range(n):
        rl = ref_lines[q]
        us = [i for i, x in enumerate(rl) if "UStar negative" in x]
        mx = [x for x in rl if "Max number of BLCond iterations" in x]
        mine = got.get(q, [])
        my_us = [m for m in mine if "UStar negative" in m[0][0]]
        my_mx = [m for m in mine if "Max number" in m[0][0]]
        assert len(my_us) == (1 if us else 0) and len(my_mx) == (1 if mx else 0), (q, mine)
        if us:  # the first occurrence, both lines, character for character; and how often it was printed
            assert my_us[0][0] == rl[us[0]:us[0] + 2], (q, my_us[0][0], rl[us[0]:us[0] + 2])
            assert int(re.search(r"passes with this message: (\d+)", my_us[0][1]).group(1)) == len(us), q
            seen += 1
        if mx:
            assert my_mx[0][0] == [mx[0]], (q, my_mx[0][0], mx[0])
            assert int(re.search(r"time indices with this message: (\d+)", my_mx[0][1]).group(1)) == len(mx), q
            seen += 1
    assert seen >= 6 and not got.get(n - 1)


@pytest.mark.skipif(not os.path.exists(oh.REF_CPL_SO), reason="needs the reference with working coupling")
def test_coupling_messages_as_the_reference_prints_them(tmp_path, monkeypatch):
    n, SL = 32, 1441
    f = oh.synth_forcing(n, SL, seed=4242)
    s0 = abi.default_settings(SL); p = abi.default_parameters(); l0 = abi.default_local(); l0.InitLenI = 1
    base, _, _ = oh.run_oracle("port", f, s0, p, l0)
    s = abi.default_settings(SL); s.use_coupling = 1
    ci = SL // 2
    ls = []
    for i, off in enumerate(np.linspace(-30.0, 30.0, n)):  # observations far below / above what the weather allows
        li = abi.default_local(); li.InitLenI = ci; li.couplingIndexI = ci
        li.couplingTsurf = float(base["tsurf"][i, ci - 1] + off)
        ls.append(li)
    f["tsurfobs"][:, :] = base["tsurf"] + 0.3
    ref_out, ref_lines = _reference("ref_cpl", f, s, p, ls, tmp_path, monkeypatch)
    out, lines = _product(f, s, p, ls, tmp_path, monkeypatch, "hip_cpl.txt")
    for k in oh.F64_OUT:
        assert np.array_equal(out[k], ref_out[k]), k
    got = _by_point(lines)
    kinds = set()
    for q in range(n):
        want = [x for x in ref_lines[q] if "coupling failed" in x]
        mine = [m[0][0] for m in got.get(q, []) if "coupling failed" in m[0][0]]
        assert mine == want, (q, mine, want)
        kinds.update("small" if "too small" in x else "big" for x in want)
    assert kinds == {"small", "big"}
