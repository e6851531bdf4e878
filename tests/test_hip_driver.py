"""GPU: the reference driver's data path on the device (layer 4, rs_driver_* through the C-ABI)
against the CPU checker: oracle/driver_oracle.c for read_input, the model oracles for the
simulation.  Everything is compared bit for bit."""
import os

import numpy as np
import pytest

import driver_helpers as dh
import oracle_helpers as oh
from roadsurf_amd import abi, driver, lib

pytestmark = pytest.mark.gpu
M = -9999.9
LP_FIELDS = ("tair_relax", "VZ_relax", "RH_relax", "couplingIndexI", "couplingTsurf", "InitLenI")


def _kind(coupled):
    if coupled:
        return "ref_cpl" if os.path.exists(oh.REF_CPL_SO) else "port"
    return "ref" if os.path.exists(oh.REF_SO) else "port"


def _settings(L, **kw):
    s = abi.default_settings(L)
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def _same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.int64), np.ascontiguousarray(b).view(np.int64))


def _compare_read_input(g, o, n):
    assert np.array_equal(g["status"], o["status"])
    assert np.array_equal(g["missing_index"], o["missing_index"])
    for k in driver.MERGED_FIELDS:
        assert _same_bits(g["merged"][k], o["merged"][k]), k
    for p in range(n):
        for f in LP_FIELDS:
            assert getattr(g["local"][p], f) == getattr(o["local"][p], f), (p, f)


def test_read_input_matches_checker_bitwise():
    n = 700
    src, L, t0, tf = dh.scenario(n, hours=12, seed=11)
    for kw in (dict(), dict(use_relaxation=1), dict(use_relaxation=1, use_coupling=1),
               dict(use_coupling=1, coupling_minutes=60)):
        s = _settings(L, **kw)
        g = driver.read_input(src, s, t0, tf)
        o = dh.oracle_read_input(src, s, t0, tf)
        _compare_read_input(g, o, n)
    # the scenario really exercises the branches
    assert (o["status"] == 0).sum() > n // 2
    assert len({int(o["local"][p].couplingIndexI) for p in range(n)}) > 5
    m = o["merged"]
    assert (m["rhz"] > -100).all(axis=1).sum() > n // 2      # RH completed from Tdew
    assert (m["tsurfobs"] > -100).any() and (m["tsurfobs"] < -9000).any()


def test_segment_scan_equals_the_index_by_index_scan(monkeypatch):
    """Shared time axes: the decisions come from a scan over the runs of constant (kind, rawPos)
    (scan_seg_kernel); ROADSURF_HIP_SCAN_FULL=1 walks every simulation index like the per-point-axis
    path.  Same decisions - also with an infinite raw value, above every threshold but poisonous to the
    interpolation (the runs that touch it are walked index by index)."""
    n = 600
    src, L, t0, tf = dh.scenario(n, hours=12, seed=19)
    src[0].fields["tair"][7, 3] = np.inf
    src[0].fields["sw"][9, 5] = np.inf
    src[1].fields["tair"][11, 2] = np.inf
    s = _settings(L, use_relaxation=1, use_coupling=1)
    fast = driver.read_input(src, s, t0, tf)
    monkeypatch.setenv("ROADSURF_HIP_SCAN_FULL", "1")
    full = driver.read_input(src, s, t0, tf)
    monkeypatch.delenv("ROADSURF_HIP_SCAN_FULL")
    _compare_read_input(fast, full, n)
    o = dh.oracle_read_input(src, s, t0, tf)
    _compare_read_input(fast, o, n)


def test_read_input_edge_cases_match_checker():
    t0 = dh.START

    def one(times, obs=False, **fields):
        return driver.RawSource(np.asarray(times, np.int64),
                                {k: np.tile(np.asarray(v, np.float64), (3, 1)) for k, v in fields.items()}, obs)

    full = dict(tair=[1.0, 2.5, 3.0, 7.0], rhz=[80.0, 85.0, 90.0, 70.0], prec=[0.0, 0.3, 0.0, 0.1],
                sw=[0.0, 10.0, 50.0, 0.0], lw=[300.0, 310.0, 290.0, 305.0], vz=[2.0, 0.1, 5.0, 3.0])
    hourly = [t0, t0 + 3600, t0 + 7200, t0 + 10800]
    cases = [
        ([one(hourly, **full)], t0, 241),                       # on grid
        ([one(hourly, **full)], t0 - 90, 241),                  # simulation starts before the data
        ([one(hourly, **full)], t0 + 1830, 121),                # ... after the first raw time
        ([one(hourly, **full)], t0 + 20000, 61),                # data entirely in the past
        ([one([t0, t0 + 3616, t0 + 7200, t0 + 9000], **full)], t0, 241),   # off-grid raw time
        ([one([t0], tair=[1.0])], t0, 11),                      # single raw time: nothing usable
        ([one(hourly, **full), one([t0 + 600, t0 + 1200, t0 + 1800], True, tair=[M, 9.0, 9.5],
                                   tsurfobs=[-1.0, M, -2.0])], t0, 241),
    ]
    for src, start, L in cases:
        for kw in (dict(), dict(use_relaxation=1, use_coupling=1, coupling_minutes=10)):
            s = _settings(L, **kw)
            g = driver.read_input(src, s, start, start + 900)
            o = dh.oracle_read_input(src, s, start, start + 900)
            _compare_read_input(g, o, 3)


@pytest.mark.parametrize("order", ["plan_order", "library_default", "natural"])
@pytest.mark.parametrize("mode", ["plain", "relaxation", "coupling", "skyview", "skyview_coupling"])
def test_run_matches_checker_bitwise(mode, order, monkeypatch):
    """order: tests/conftest.py asks rs_driver_run for plan order whatever the block size (ROADSURF_HIP_CLUSTER=1);
    the library's own default for blocks under 4 096 points - natural order, launches of eight hours, no re-sorts -
    and natural order with the ordinary launch length are held to the same bits here."""
    if order == "library_default":
        monkeypatch.delenv("ROADSURF_HIP_CLUSTER", raising=False)
    elif order == "natural":
        monkeypatch.setenv("ROADSURF_HIP_CLUSTER", "0")
    n = 384
    src, L, t0, tf = dh.scenario(n, hours=12, seed=23)
    kw = {"plain": dict(), "relaxation": dict(use_relaxation=1),
          "coupling": dict(use_relaxation=1, use_coupling=1), "skyview": dict(use_relaxation=1),
          "skyview_coupling": dict(use_relaxation=1, use_coupling=1)}[mode]
    if mode == "skyview_coupling":  # launch boundaries inside the coupling windows
        monkeypatch.setenv("ROADSURF_HIP_CHUNK_STEPS", "97")
    s = _settings(L, outputStep=20, **kw)
    p = abi.default_parameters()
    local, hz = None, None
    if mode.startswith("skyview"):
        rs = np.random.RandomState(3)
        local = []
        for i in range(n):
            lp = abi.default_local()
            lp.lat, lp.lon = 60.0 + rs.uniform(0, 8), 21.0 + rs.uniform(0, 8)
            lp.sky_view = float(rs.uniform(0.3, 1.0)) if i % 3 else 1.0
            local.append(lp)
        hz = rs.uniform(0, 25, (n, 360))
    g = driver.run(src, s, p, t0, tf, local=local, horizons=hz)
    o = dh.oracle_run(_kind(mode.endswith("coupling")), src, s, p, t0, tf, local=local, horizons=hz)
    assert g["step"] == o["step"] == 40
    assert np.array_equal(g["status"], o["status"])
    assert np.array_equal(g["missing_index"], o["missing_index"])
    rejected = o["status"] != 0
    assert 0 < rejected.sum() < n // 2
    for k in driver.OUT_FIELDS:
        assert _same_bits(g[k], o[k]), (mode, k, float(np.abs(g[k] - o[k]).max()))
        assert (g[k][rejected] == -9999.0).all()
    assert (g["tsurf"][~rejected] > -100).all()
    for q in range(n):
        for f in LP_FIELDS:
            assert getattr(g["local"][q], f) == getattr(o["local"][q], f), (q, f)
    if mode.endswith("coupling"):
        base = driver.run(src, _settings(L, outputStep=20, use_relaxation=1), p, t0, tf, local=local, horizons=hz)
        moved = (np.abs(base["tsurf"] - g["tsurf"]).max(1) > 1e-3)[~rejected].sum()
        assert moved > (~rejected).sum() // 2     # coupling really acts


@pytest.mark.parametrize("mode", ["relaxation", "skyview"])
def test_step_kernel_reads_the_raw_series(mode, monkeypatch):
    """rs_driver_run's blocks step with the two-wavefront kernel whose ground wave makes the forcing from the
    raw series itself (JsonSource::interpolate + the GetWeather overlay in registers, rs_raw.hpp; with sky view
    a third wavefront for the radiation): no forcing window, no expansion kernel.  Same bits as the window
    flavour (ROADSURF_HIP_DRIVER_WINDOWS=1) and as the checker - also where the short form cannot be used:
    lanes of a wavefront whose observations are missing in different places, an infinite raw value, a raw
    -0.0 that is copied, and every launch boundary position relative to the raw times."""
    n = 500
    src, L, t0, tf = dh.scenario(n, hours=12, seed=29)
    src[0].fields["tair"][7, 3] = np.inf       # poisons two raw intervals of one point
    src[0].fields["prec"][9, 4] = -0.0          # copied at a raw time: the sign must survive
    src[0].fields["sw"][11, 5] = np.inf
    src[1].fields["vz"][13, 2] = np.inf
    s = _settings(L, outputStep=10, use_relaxation=1)
    p = abi.default_parameters()
    local, hz = None, None
    if mode == "skyview":
        rs = np.random.RandomState(5)
        local = []
        for i in range(n):
            lp = abi.default_local()
            lp.lat, lp.lon = 60.0 + rs.uniform(0, 8), 21.0 + rs.uniform(0, 8)
            lp.sky_view = float(rs.uniform(0.3, 1.0)) if i % 4 else 1.0
            local.append(lp)
        hz = rs.uniform(0, 25, (n, 360))
    o = dh.oracle_run("port", src, s, p, t0, tf, local=local, horizons=hz)
    L_ = lib.load()
    for chunk in (256, 97, 1):
        if chunk == 1 and mode == "skyview":
            continue
        monkeypatch.setenv("ROADSURF_HIP_CHUNK_STEPS", str(chunk) if chunk > 1 else str(L))
        g = driver.run(src, s, p, t0, tf, local=local, horizons=hz, device=0)
        assert L_.rs_driver_last_raw_launches() == (-(-L // chunk) if chunk > 1 else 1)
        monkeypatch.setenv("ROADSURF_HIP_DRIVER_WINDOWS", "1")
        w = driver.run(src, s, p, t0, tf, local=local, horizons=hz, device=0)
        assert L_.rs_driver_last_raw_launches() == 0
        monkeypatch.delenv("ROADSURF_HIP_DRIVER_WINDOWS")
        assert np.array_equal(g["status"], o["status"]) and np.array_equal(w["status"], o["status"])
        for k in driver.OUT_FIELDS:
            assert _same_bits(g[k], w[k]), (mode, chunk, k, int((g[k] != w[k]).sum()))
            assert _same_bits(g[k], o[k]), (mode, chunk, k, int((g[k] != o[k]).sum()))


def test_tiling_does_not_change_results(monkeypatch):
    n = 333
    src, L, t0, tf = dh.scenario(n, hours=6, seed=5, obs_hours=3)
    s = _settings(L, use_relaxation=1)
    p = abi.default_parameters()
    a = driver.run(src, s, p, t0, tf)
    monkeypatch.setenv("ROADSURF_HIP_TILE_POINTS", "100")
    monkeypatch.setenv("ROADSURF_HIP_CHUNK_STEPS", "97")
    b = driver.run(src, s, p, t0, tf)
    for k in driver.OUT_FIELDS:
        assert _same_bits(a[k], b[k]), k
    assert np.array_equal(a["status"], b["status"])


@pytest.mark.parametrize("coupled", [False, True], ids=["relaxation", "coupling"])
def test_tiles_are_cut_to_what_32_bit_offsets_address(coupled, monkeypatch):
    """The raw-series step kernels address a tile's whole output window with 32-bit offsets and refuse a window of
    2^29 elements per stream or more (250 000 points x 48 h with outputStep = 1 min would be 7.2e8): rs_driver_run
    cuts its tiles to fit.  The limit lowered to megabytes (ROADSURF_HIP_A32_LIMIT), a block that fitted one tile
    takes several, with the same bits; before round 6 the call failed with -13."""
    n = 1500
    src, L, t0, tf = dh.scenario(n, hours=6, seed=8, obs_hours=3)
    s = _settings(L, outputStep=1, use_relaxation=1, use_coupling=1 if coupled else 0)
    p = abi.default_parameters()
    monkeypatch.setenv("ROADSURF_HIP_DEVICES", "0")  # one block: the tiles are what is counted
    a = driver.run(src, s, p, t0, tf)
    assert lib.load().rs_driver_last_raw_launches() > 0 and lib.load().rs_driver_last_tiles() == 1
    n_out = a["tsurf"].shape[1]
    assert n_out == (L + 1) // 2
    monkeypatch.setenv("ROADSURF_HIP_A32_LIMIT", str(600 * n_out))  # tiles of 512 points at most
    b = driver.run(src, s, p, t0, tf)
    assert lib.load().rs_driver_last_tiles() == 3 and lib.load().rs_driver_last_raw_launches() > 0
    for k in driver.OUT_FIELDS:
        assert _same_bits(a[k], b[k]), k
    assert np.array_equal(a["status"], b["status"])
    o = dh.oracle_run(_kind(coupled), src, s, p, t0, tf)
    for k in driver.OUT_FIELDS:
        assert _same_bits(b[k], o[k]), k


def test_argument_errors_are_reported():
    src, L, t0, tf = dh.scenario(4, hours=1, seed=1, obs_hours=1)
    with pytest.raises(ValueError, match="outputStep"):
        driver.run(src, _settings(L, outputStep=0), abi.default_parameters(), t0, tf)
    s = _settings(L)
    s.NLayers = 99
    with pytest.raises(RuntimeError, match="bad settings"):
        driver.run(src, s, abi.default_parameters(), t0, tf)


def test_per_point_time_axes_match_checker():
    """Observation series with their own time stamps per point (ragged, padded rows)."""
    n = 300
    src, L, t0, tf = dh.scenario(n, hours=12, seed=31)
    pp = [src[0], dh.ragged(src[1], seed=4)]
    assert pp[1].times.ndim == 2 and pp[1].lengths.min() == 0 and pp[1].lengths.max() > 10
    for kw in (dict(use_relaxation=1), dict(use_relaxation=1, use_coupling=1, coupling_minutes=60)):
        s = _settings(L, **kw)
        g = driver.read_input(pp, s, t0, tf)
        o = dh.oracle_read_input(pp, s, t0, tf)
        _compare_read_input(g, o, n)
    # dropping stamps changes what read_input returns (gaps are bridged instead of staying gaps)
    shared = dh.oracle_read_input(src, s, t0, tf)
    assert not np.array_equal(shared["merged"]["tsurfobs"], o["merged"]["tsurfobs"])
    # both sources per point, forecast too; simulation chunked so that the walks are carried on
    both = [dh.ragged(src[0], seed=9, drop=0.1), pp[1]]
    s = _settings(L, use_relaxation=1, outputStep=30)
    p = abi.default_parameters()
    o = dh.oracle_run(_kind(False), both, s, p, t0, tf)
    g = driver.run(both, s, p, t0, tf)
    assert np.array_equal(g["status"], o["status"]) and (o["status"] == 0).sum() > n // 4
    assert np.array_equal(g["missing_index"], o["missing_index"])
    for k in driver.OUT_FIELDS:
        assert _same_bits(g[k], o[k]), k


def test_per_point_time_axes_with_chunked_coupling(monkeypatch):
    """Coupling is time-chunked in rs_driver_run: lock-step chunks, the replay block (which starts
    BEFORE the last chunk ended), chunks again from behind the first coupling window.  With
    per-point time axes the raw-series walks have to be re-positioned for every such jump; points
    lose the tail of their observations, so their coupling windows end at different indices."""
    n = 260
    src, L, t0, tf = dh.scenario(n, hours=8, seed=37, obs_hours=5)
    both = [dh.ragged(src[0], seed=9, drop=0.1), dh.ragged(src[1], seed=4)]
    s = _settings(L, use_relaxation=1, use_coupling=1, coupling_minutes=90, outputStep=10)
    p = abi.default_parameters()
    o = dh.oracle_run(_kind(True), both, s, p, t0, tf)
    ci = np.array([o["local"][q].couplingIndexI for q in range(n)])
    ok = o["status"] == 0
    assert len(set(ci[ok & (ci > 0)].tolist())) > 3      # windows end at different indices
    for chunk, tile in ((64, 4096), (200, 100)):
        monkeypatch.setenv("ROADSURF_HIP_CHUNK_STEPS", str(chunk))
        monkeypatch.setenv("ROADSURF_HIP_TILE_POINTS", str(tile))
        g = driver.run(both, s, p, t0, tf)
        assert np.array_equal(g["status"], o["status"])
        for k in driver.OUT_FIELDS:
            assert _same_bits(g[k], o[k]), (chunk, k)
    monkeypatch.setenv("ROADSURF_HIP_CLUSTER", "0")        # natural order (plan order is the default)
    g = driver.run(both, s, p, t0, tf)
    for k in driver.OUT_FIELDS:
        assert _same_bits(g[k], o[k]), ("natural order", k)
    monkeypatch.delenv("ROADSURF_HIP_CLUSTER")
    monkeypatch.setenv("ROADSURF_HIP_CPL_WHOLE", "1")     # round-1 organisation: same bits
    g = driver.run(both, s, p, t0, tf)
    for k in driver.OUT_FIELDS:
        assert _same_bits(g[k], o[k]), ("whole", k)


def test_shared_axis_chunked_coupling_in_plan_order(monkeypatch):
    """Shared time axes: the lock-step chunks of the coupled run are re-sorted by the forecast key
    (previews of the next chunk from the raw series), the replays run over the slots as sorted at
    the end of stage 1, outputs go to their point's column.  Same bits as the reference, as in
    natural order and as with the history key."""
    n = 300
    src, L, t0, tf = dh.scenario(n, hours=9, seed=41, obs_hours=5)
    s = _settings(L, use_relaxation=1, use_coupling=1, coupling_minutes=120, outputStep=5)
    p = abi.default_parameters()
    o = dh.oracle_run(_kind(True), src, s, p, t0, tf)
    ok = o["status"] == 0
    assert ok.sum() > n // 2
    monkeypatch.setenv("ROADSURF_HIP_CHUNK_STEPS", "60")
    for env in (dict(), dict(ROADSURF_HIP_CLUSTER="0")):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g = driver.run(src, s, p, t0, tf)
        for k in env:
            monkeypatch.delenv(k)
        assert np.array_equal(g["status"], o["status"]), env
        for k in driver.OUT_FIELDS:
            assert _same_bits(g[k], o[k]), (env, k, int((g[k] != o[k]).sum()))


def test_spread_coupling_windows_are_cut_to_the_window_budget(monkeypatch):
    """Chunked coupling sizes its replay block from the tile's couplingIndexI values: a station whose
    observations stopped early stretches the block, and with it the forcing windows.  A tile whose
    windows would exceed ROADSURF_HIP_WINDOW_BUDGET_MB is cut in halves (down to 4 096 points) and
    started again; the results do not change."""
    n = 9000
    src, L, t0, tf = dh.scenario(n, hours=6, seed=43, obs_hours=4)
    s = _settings(L, use_relaxation=1, use_coupling=1, coupling_minutes=30, outputStep=10)
    p = abi.default_parameters()
    o = dh.oracle_run(_kind(True), src, s, p, t0, tf)
    ci = np.array([o["local"][q].couplingIndexI for q in range(n)])[o["status"] == 0]
    assert ci.max() - ci.min() > 100          # the replay block is much longer than a window
    monkeypatch.setenv("ROADSURF_HIP_CHUNK_STEPS", "64")
    monkeypatch.setenv("ROADSURF_HIP_WINDOW_BUDGET_MB", "40")
    g = driver.run(src, s, p, t0, tf, device=0)
    assert np.array_equal(g["status"], o["status"])
    for k in driver.OUT_FIELDS:
        assert _same_bits(g[k], o[k]), (k, int((g[k] != o[k]).sum()))
    assert lib.load().rs_driver_last_tiles() >= 3   # 9 000 points: 4 096 + 4 096 + the rest


def test_identical_per_point_axes_equal_the_shared_axis(monkeypatch):
    n = 200
    src, L, t0, tf = dh.scenario(n, hours=6, seed=8, obs_hours=3)
    tiled = [driver.RawSource(np.tile(x.times, (n, 1)), x.fields, x.is_observation) for x in src]
    s = _settings(L, use_relaxation=1, use_coupling=1, coupling_minutes=30)
    p = abi.default_parameters()
    monkeypatch.setenv("ROADSURF_HIP_TILE_POINTS", "128")
    a = driver.run(src, s, p, t0, tf)
    b = driver.run(tiled, s, p, t0, tf)
    for k in driver.OUT_FIELDS:
        assert _same_bits(a[k], b[k]), k
    assert np.array_equal(a["status"], b["status"])
    for q in range(n):
        for f in LP_FIELDS:
            assert getattr(a["local"][q], f) == getattr(b["local"][q], f)


def _random_case(seed):
    rs = np.random.RandomState(500 + seed)
    n = int(rs.choice([70, 257, 400]))
    hours = int(rs.choice([4, 6, 12]))
    obs_hours = int(rs.randint(2, hours))
    src, L, t0, tf = dh.scenario(n, hours=hours, seed=40 + seed, obs_hours=obs_hours, gaps=bool(rs.rand() < 0.8))
    if rs.rand() < 0.5:      # a third source: radiation from another model, every 3 hours
        fc = src[0]
        idx = np.arange(0, len(fc.times), 3)
        rad = driver.RawSource(fc.times[idx], {k: np.ascontiguousarray(fc.fields[k][:, idx] * 1.05)
                                               for k in ("sw", "lw")}, False)
        src = [fc, rad, src[1]]
    if rs.rand() < 0.5:
        src[-1] = dh.ragged(src[-1], seed=seed, drop=float(rs.uniform(0.05, 0.4)))
    if rs.rand() < 0.3:
        src[0] = dh.ragged(src[0], seed=seed + 1, drop=0.08)
    kw = dict(use_relaxation=int(rs.rand() < 0.7), use_coupling=int(rs.rand() < 0.5),
              coupling_minutes=int(rs.choice([30, 60, 120])), outputStep=int(rs.choice([10, 20, 60])),
              NLayers=int(rs.choice([7, 15, 15, 22])),
              tsurfOutputDepth=float(rs.choice([-9999.9, -9999.9, 0.0, 0.03])))
    s = _settings(L, **kw)
    local, hz = None, None
    if rs.rand() < 0.4:
        local = []
        for i in range(n):
            lp = abi.default_local()
            lp.lat, lp.lon = 60.0 + rs.uniform(0, 8), 21.0 + rs.uniform(0, 8)
            lp.sky_view = float(rs.uniform(0.3, 1.0)) if rs.rand() < 0.7 else 1.0
            local.append(lp)
        hz = rs.uniform(0, 25, (n, 360)) if rs.rand() < 0.7 else None
    tile = int(rs.choice([64, 200, 4096]))
    chunk = int(rs.choice([50, 97, 256]))
    return src, s, t0, tf, local, hz, tile, chunk


@pytest.mark.parametrize("seed", range(12))
def test_random_driver_case_matches_checker(seed, monkeypatch):
    src, s, t0, tf, local, hz, tile, chunk = _random_case(seed)
    monkeypatch.setenv("ROADSURF_HIP_TILE_POINTS", str(tile))
    monkeypatch.setenv("ROADSURF_HIP_CHUNK_STEPS", str(chunk))
    p = abi.default_parameters()
    g = driver.run(src, s, p, t0, tf, local=local, horizons=hz)
    o = dh.oracle_run("port", src, s, p, t0, tf, local=local, horizons=hz)
    desc = (f"n{len(g['status'])} L{s.SimLen} src{len(src)} pp{[x.times.ndim == 2 for x in src]} relax{s.use_relaxation} "
            f"cpl{s.use_coupling}/{s.coupling_minutes} out{s.outputStep} NL{s.NLayers} depth{s.tsurfOutputDepth:g} "
            f"sky{local is not None} hz{hz is not None} tile{tile} chunk{chunk}")
    assert np.array_equal(g["status"], o["status"]), desc
    assert np.array_equal(g["missing_index"], o["missing_index"]), desc
    for k in driver.OUT_FIELDS:
        assert _same_bits(g[k], o[k]), (desc, k, int((g[k] != o[k]).sum()))
    for q in range(len(g["status"])):
        for f in LP_FIELDS:
            assert getattr(g["local"][q], f) == getattr(o["local"][q], f), (desc, q, f)


def test_driver_run_fans_out_over_the_device_list(monkeypatch):
    """rs_driver_run with device < 0 cuts the points into blocks over ROADSURF_HIP_DEVICES (one
    worker thread, stream, window block and plans per entry).  "0,0" = two concurrent workers on
    the one GPU of the test box: same bits as the single-device call, decisions included."""
    from roadsurf_amd import lib
    n = 600
    src, L, t0, tf = dh.scenario(n, hours=6, seed=9, obs_hours=3)
    s = _settings(L, use_relaxation=1)
    p = abi.default_parameters()
    a = driver.run(src, s, p, t0, tf, device=0)
    monkeypatch.setenv("ROADSURF_HIP_DEVICES", "0,0")
    monkeypatch.setenv("ROADSURF_HIP_MIN_SHARD", "64")
    b = driver.run(src, s, p, t0, tf, device=-1)
    assert lib.load().rs_last_fanout() == 2
    for k in driver.OUT_FIELDS:
        assert _same_bits(a[k], b[k]), k
    assert np.array_equal(a["status"], b["status"])
    assert np.array_equal(a["missing_index"], b["missing_index"])
    for q in range(n):
        for f in LP_FIELDS:
            assert getattr(a["local"][q], f) == getattr(b["local"][q], f), (q, f)


@pytest.mark.parametrize("mode", ["relax", "coupling", "skyview"])
def test_observations_with_holes_on_the_benchmark_workload(mode):
    """The raw-series benchmark workload (roadsurf_amd/driver_workload.py) with observation series as real
    networks have them: stations without an air-temperature / humidity / wind sensor, gaps in the road
    temperature, series that end hours apart.  The lanes of a wavefront then disagree on the source that
    supplies a variable and the step kernel's ground wave keeps every interpolating source's line per lane
    (rs_kernels.hip raw_values, MIXED).  Same bits as the checker, in plan order, several wavefronts per class."""
    from roadsurf_amd import driver_workload
    n, hours = 3000, 8
    w = driver_workload.DriverWorkload(n, hours, seed=11, missing=0.15, ragged=0.2)
    src, s, p = w.sources(mode), w.settings(mode), abi.default_parameters()
    loc = w.local(mode)
    hz = w.horizons() if mode == "skyview" else None
    t0, tf = driver_workload.START, driver_workload.START + driver_workload.OBS_HOURS * 3600
    o = dh.oracle_run(_kind(mode == "coupling"), src, s, p, t0, tf, local=loc, cal=w.cal, horizons=hz)
    g = driver.run(src, s, p, t0, tf, cal=w.cal, local=w.local(mode), horizons=hz)
    assert np.array_equal(g["status"], o["status"]) and (o["status"] == 0).sum() > n // 2
    for k in driver.OUT_FIELDS:
        assert _same_bits(g[k], o[k]), (mode, k, int((g[k] != o[k]).sum()))
    assert lib.load().rs_driver_last_raw_launches() > 0
