"""GPU: the fp32 flavour (BASELINE.json configs[4], "fp32 kernels with fp64 tolerance gate").

fp32 cannot follow the fp64 reference point by point: the storage logic branches on the SIGN of
rounding residuals (roadsurf_amd/csrc/rs_math.hpp), so a fraction of the points takes the other
side of a melt-out branch and carries a transient of up to ~0.2 K.  The gate is therefore on
the DISTRIBUTION of |fp32 - fp64 oracle| over 4 096 points x 48 h (measured values in
brackets, tolerances ~3x above them):
    Tsurf  rms < 1e-3 K [3e-4]   99.9th percentile < 1e-3 K [2.8e-4]   max < 0.5 K [0.17]
    fraction of point-steps off by more than 0.05 K < 1e-4 [9e-6]
    storages  rms < 5e-4 mm [8e-5]   99.9th percentile < 5e-4 mm [8e-5]   max < 0.1 mm [0.024]"""
import os
import sys

import numpy as np  # noqa: E402
import pytest

import oracle_helpers as oh
from roadsurf_amd import abi

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_fp32_distribution_gate_against_fp64_oracle():
    from f32_experiment import run_f32
    n, L, seed = 4096, 5761, 20240110
    f = oh.synth_forcing(n, L, seed=seed)
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    ora, _, _ = oh.run_oracle("port", f, s, p, l)
    res = run_f32(n, L, seed)
    d = np.abs(res["tsurf"] - ora["tsurf"])
    print("tsurf rms %.2e p99.9 %.2e max %.2e frac>0.05K %.2e" %
          (np.sqrt((d ** 2).mean()), np.percentile(d, 99.9), d.max(), (d > 0.05).mean()))
    assert np.sqrt((d ** 2).mean()) < 1e-3 and np.percentile(d, 99.9) < 1e-3 and d.max() < 0.5
    assert (d > 0.05).mean() < 1e-4
    for k in ("snow", "water", "ice", "deposit", "ice2"):
        e = np.abs(res[k] - ora[k])
        assert np.sqrt((e ** 2).mean()) < 5e-4 and np.percentile(e, 99.9) < 5e-4 and e.max() < 0.1, k


def test_fp32_rejects_what_it_does_not_have():
    """What the fp32 flavour still refuses, with a message: time-chunked coupling (rs_hip_step_cpl / rs_hip_cpl_replay; a
    coupled fp32 plan runs its series whole), diagnostics, the write-back of the in-place input edits."""
    import torch
    from roadsurf_amd import device
    p = abi.default_parameters()
    # coupling: the whole series in one launch (rs_hip_step); the time-chunked pair stays with the fp64 flavour
    s = abi.default_settings(100); s.use_coupling = 1
    plan = device.Plan(64, s, p, 0)
    plan.set_precision(32)
    win = device.ForcingWindow.empty(100, plan.np_pad, plan.device, optional=("tdew", "tsurfobs"), dtype=torch.float32)
    out = device.OutputWindow.empty(100, plan.np_pad, plan.device, dtype=torch.float32)
    z = torch.zeros(plan.np_pad, dtype=torch.int32, device=plan.device)
    pp = plan.point_params(5.0, z + 1, None, None, None, z + 50, torch.zeros(plan.np_pad, dtype=torch.float64, device=plan.device))
    with pytest.raises(RuntimeError, match="time-chunked coupling needs the fp64 flavour"):
        plan.step_cpl(win, out, pp, 1, 50)
    with pytest.raises(RuntimeError, match="whole series"):
        plan.step(win, out, pp, 1, 50)
    from roadsurf_amd import lib
    assert plan.L.rs_hip_set_diagnostics(plan._h, 1) != 0 and "fp64 flavour only" in lib.last_error()
    plan.close()


def test_fp32_plan_order_changes_no_value():
    """rs_hip_recluster on an fp32 plan: the re-sorted run equals the natural-order run bit for bit."""
    from f32_experiment import run_f32
    n, L, seed = 2000, 1441, 7
    a = run_f32(n, L, seed, chunk=120)
    b = run_f32(n, L, seed, chunk=120, cluster=True)
    for k in ("tsurf", "snow", "water", "ice", "deposit", "ice2"):
        assert np.array_equal(a[k], b[k]), k


# ---- BASELINE config 5 at its shape: 7-day hindcast (SimLen 20 161), 1.25 M points per GPU -------
# (10 M points over 8 GPUs; examples/example1/src/InputSettings.cpp:98 gives SimLen = 1 + 168*3600/30)
#
# SURVEY.md 8d proposes 0.05 K / 0.05 mm as the tolerance "to be justified empirically".  Measured
# (tools/f32_7d.py, 2 048 points x 7 d, two seeds; in brackets) and gated here at a few times that:
#   * 99.999 % of the point-steps are within 0.05 K of the fp64 reference  [beyond: 2.1e-6, 7.8e-7]
#   * the rest are short transients after a melt-out branch went the other way (rs_math.hpp):
#     0.2-0.3 % of the points ever leave the 0.05 K band, for a median of 3-6 minutes, the longest
#     12 minutes, never further than 0.17 K; they decay - days 6-7 are at rms 3e-6 K: no drift
#   * Tsurf rms 1.5e-4 K, 99.9th percentile 6e-5 K; storages rms < 4e-5 mm, max 0.06 mm
# A pointwise 0.05 K bound cannot hold for arithmetic that differs at all (the reference's own
# storage logic amplifies a last-bit difference to 0.2 K); the distribution does.
HOURS7, L7 = 168, 168 * 120 + 1


@pytest.mark.parametrize("seed", [20240110, 777])
def test_fp32_seven_day_distribution_gate(seed):
    from f32_experiment import run_f32
    n = 2048
    f = oh.synth_forcing(n, L7, seed=seed)
    s = abi.default_settings(L7); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    ora, _, _ = oh.run_oracle("ref" if oh.have_ref() else "port", f, s, p, l)
    res = run_f32(n, L7, seed)
    d = np.abs(res["tsurf"] - ora["tsurf"])
    over = d > 0.05
    runs = [0]
    for q in np.where(over.any(1))[0]:
        x = np.flatnonzero(np.diff(np.concatenate([[0], over[q].astype(np.int8), [0]])))
        runs += list(x[1::2] - x[0::2])
    print("seed %d: rms %.2e p99.9 %.2e max %.3f frac>0.05K %.2e points ever %.4f longest %.0f min" %
          (seed, np.sqrt((d ** 2).mean()), np.percentile(d, 99.9), d.max(), over.mean(),
           over.any(1).mean(), max(runs) * 0.5))
    assert over.mean() < 1e-5                     # 99.999 % of the point-steps within 0.05 K
    assert over.any(1).mean() < 0.015 and max(runs) * 30 < 2 * 3600   # few points, short transients
    assert d.max() < 0.5 and np.sqrt((d ** 2).mean()) < 5e-4 and np.percentile(d, 99.9) < 3e-4
    last2 = d[:, -2 * 2880:]
    assert np.sqrt((last2 ** 2).mean()) < 2e-4    # no drift: the last two days are as good as the first
    for k in ("snow", "water", "ice", "deposit", "ice2"):
        e = np.abs(res[k] - ora[k])
        assert np.sqrt((e ** 2).mean()) < 2e-4 and e.max() < 0.2 and (e > 0.05).mean() < 1e-6, k


def test_fp32_config5_shape_properties_1250000_points_7_days():
    """The shape itself: 1.25 M points (one GPU's share of 10 M over 8) x SimLen 20 161 in fp32
    through the object bench.py times - no NaN, storages within their limits, deterministic,
    plan order value-neutral (checksum and sampled points equal to the natural-order pass)."""
    import torch
    from roadsurf_amd import device, workload
    from test_hip_golden_and_scale import _synthetic_pass
    n, seed, chunk = 1_250_000, 20240110, 240
    s = abi.default_settings(L7); p = abi.default_parameters()
    plan = device.Plan(n, s, p, 0)
    plan.set_precision(32)
    cols = np.concatenate([np.arange(b, b + 64) for b in (0, 600_000, 1_249_920)])
    run = workload.SyntheticRun(plan, seed, HOURS7, chunk, plan_order=False, f32=True)
    assert run.simlen == L7
    c1, samp1, mins, maxs = _synthetic_pass(run, cols, itype=torch.int32)
    del run
    torch.cuda.empty_cache()
    assert plan.failed_count() == 0
    assert -100.0 <= mins["tsurf"] and maxs["tsurf"] <= 100.0
    for k, hi in (("snow", p.MaxSnowmms), ("water", p.MaxWatmms), ("ice", p.MaxIcemms),
                  ("deposit", p.MaxDepmms), ("ice2", p.MaxIcemms)):
        assert mins[k] >= 0.0 and maxs[k] <= hi * (1 + 1e-6), (k, mins[k], maxs[k])
    assert maxs["snow"] > 1 and maxs["ice"] > 1
    run = workload.SyntheticRun(plan, seed, HOURS7, chunk, plan_order=True, f32=True)
    c2, samp2, mins2, maxs2 = _synthetic_pass(run, cols, itype=torch.int32)
    assert c2 == c1 and mins2 == mins and maxs2 == maxs
    for k in device.OUT_FIELDS:
        assert np.array_equal(samp1[k], samp2[k]), k
    del run
    plan.close()
    # the sampled blocks against the fp64 reference: same gate as above, on this run's own points
    l = abi.default_local(); l.InitLenI = 1
    f = oh.synth_forcing(64, L7, seed=seed, point_offset=600_000)
    ora, _, _ = oh.run_oracle("ref" if oh.have_ref() else "port", f, s, p, l)
    d = np.abs(samp1["tsurf"][:, 64:128].T.astype(np.float64) - ora["tsurf"])
    assert (d > 0.05).mean() < 1e-4 and d.max() < 0.5


# ---- round 6: two points per lane, two wavefronts per 128 points (step_kernel_f32duo) ----------------

@pytest.mark.parametrize("n", [1000, 1001, 129])
def test_fp32_knot_reading_launch_equals_window_launch(n):
    """The fp32 step kernel interpolates its forcing from the resident hourly knots itself (rs_hip_step_knots on an
    fp32 plan: no forcing window, no expansion kernel) with the arithmetic expand_kernel_f32 writes a window with
    (rs32_lerp): the two launches hand the model the same bits - in natural order and in plan order, for a point
    count that leaves the last lane one point (a lane owns two) and the last workgroup mostly empty, with launch
    boundaries on, before and between knots."""
    from f32_experiment import run_f32
    L, seed = 1441, 5
    for chunk, cluster in ((97, False), (120, True), (250, True)):
        a = run_f32(n, L, seed, chunk=chunk, cluster=cluster)
        b = run_f32(n, L, seed, chunk=chunk, cluster=cluster, fused=True)
        for k in ("tsurf", "snow", "water", "ice", "deposit", "ice2"):
            assert np.array_equal(a[k], b[k]), (n, chunk, cluster, k, int((a[k] != b[k]).sum()))
        assert np.isfinite(a["tsurf"]).all() and (a["tsurf"] > -9000).all()
    # the history score of rs_hip_recluster is bookkeeping: the instances without it write the same values
    a = run_f32(n, L, seed, chunk=97)
    b = run_f32(n, L, seed, chunk=97, history_score=False)
    for k in ("tsurf", "snow", "water", "ice", "deposit", "ice2"):
        assert np.array_equal(a[k], b[k]), k


def test_fp32_other_layer_counts_and_the_one_point_per_lane_flavour():
    """NLayers != 15 (and variant 2 for any count) keep round 2-5's organisation: one point per lane, the profile in
    LDS (step_kernel_f32_lds) over the one-point physics source.  Held to the fp64 oracle by the distribution, and
    - for NLayers = 15 - to the two-points-per-lane flavour within the same bounds (the two are not bit-equal:
    other roundings in the boundary-layer loop and the layer polynomials, rs_kernels_f32.hip)."""
    from f32_experiment import run_f32
    n, L, seed = 512, 1441, 3
    f = oh.synth_forcing(n, L, seed=seed)
    p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    for nl, variant in ((12, 0), (15, 2), (15, 0)):
        s = abi.default_settings(L); s.NLayers = nl
        ora, _, _ = oh.run_oracle("port", f, s, p, l)
        res = run_f32(n, L, seed, chunk=240, variant=variant, nlayers=nl)
        d = np.abs(res["tsurf"] - ora["tsurf"])
        assert np.sqrt((d ** 2).mean()) < 1e-3 and np.percentile(d, 99.9) < 2e-3 and d.max() < 0.5, (nl, variant, d.max())
        for k in ("snow", "water", "ice", "deposit", "ice2"):
            e = np.abs(res[k] - ora[k])
            assert np.sqrt((e ** 2).mean()) < 5e-4 and e.max() < 0.1, (nl, variant, k)


def test_fp32_failed_points_leave_the_loop_at_the_reference_index():
    """CheckValues in the fp32 flavour (src/InputOutput.f90:45-84): a point whose forcing leaves the limits fails at
    the first such index - that index is still stepped and saved, every later row reads -9999.0 - and its lane
    partner and its neighbours are untouched.  The knot-reading launch tests an hour at a time where both knots
    keep a margin to the limits and index by index otherwise: same indices, same bits as the window launch."""
    from f32_experiment import run_f32
    n, L, seed, spk = 300, 1441, 9, 120
    bad = {7: (5, 0, 150.0),      # air temperature above 100 from some index before knot 5 on
           8: (0, 3, 130.0),      # relative humidity above 120 at index 1 (the lane partner of point 9)
           200: (9, 2, 101.0),    # wind speed
           299: (3, 4, -5.0)}     # precipitation below -0.1 (last point: the last lane's second point)

    def edit(knots):
        for pnt, (k, fld, v) in bad.items():
            knots[k, fld, pnt] = v

    clean = run_f32(n, L, seed, chunk=120)
    win, iw = run_f32(n, L, seed, chunk=120, edit_knots=edit, return_plan_info=True)
    kn, ik = run_f32(n, L, seed, chunk=120, edit_knots=edit, fused=True, return_plan_info=True)
    srt, isr = run_f32(n, L, seed, chunk=120, edit_knots=edit, fused=True, cluster=True, return_plan_info=True)
    assert iw["failed"] == ik["failed"] == isr["failed"] == len(bad)
    assert np.array_equal(iw["first_failed"], ik["first_failed"]) and np.array_equal(iw["first_failed"], isr["first_failed"])
    ff = iw["first_failed"]
    assert ff[8] == 1 and (ff[[p for p in range(n) if p not in bad]] == 0).all()
    assert 4 * spk + 1 < ff[7] <= 5 * spk + 1 and 8 * spk + 1 < ff[200] <= 9 * spk + 1 and 2 * spk + 1 < ff[299] <= 3 * spk + 1
    for k in ("tsurf", "snow", "water", "ice", "deposit", "ice2"):
        assert np.array_equal(win[k], kn[k]) and np.array_equal(win[k], srt[k]), k
        for pnt in bad:
            f = int(ff[pnt])                       # rows [0, f) saved (index f included), the rest blank
            assert (win[k][pnt, f:] == -9999.0).all() and (win[k][pnt, :f] > -9000).all(), (k, pnt)
        ok = [p for p in range(n) if p not in bad]
        assert np.array_equal(win[k][ok], clean[k][ok]), k
    # up to the hour in which its forcing starts to differ a failing point follows the clean run
    assert np.array_equal(win["tsurf"][7, :4 * spk], clean["tsurf"][7, :4 * spk])


@pytest.mark.parametrize("chunk", [0, 97])
def test_fp32_full_feature_set_against_the_fp64_reference(chunk):
    """Round 6: the fp32 flavour with the FULL feature set of the two-wavefront kernels - CheckValues' dew-point test,
    the observation SetCurrentValues forces on Tmp(1:2) during an initialization phase that ends at a different index
    for every point (src/InputOutput.f90:116-148), RelaxationOperations behind it (src/Relaxation.f90:10-47) - against
    the fp64 reference on the same inputs: the distribution gate of the LEAN flavour, and a point whose dew point
    leaves the limits fails at the reference's index.  Launches that cut the initialization phases (chunk 97)."""
    from roadsurf_amd import device
    n, L, seed = 1024, 2881, 17
    f = oh.synth_forcing(n, L, seed=seed)
    s = abi.default_settings(L); s.use_relaxation = 1
    p = abi.default_parameters()
    lean = abi.default_settings(L)
    l0 = abi.default_local(); l0.InitLenI = 1
    base, _, _ = oh.run_oracle("port", f, lean, p, l0)
    rs = np.random.RandomState(5)
    ls = []
    for i in range(n):
        li = abi.default_local()
        li.InitLenI = int(rs.choice([1, 240, 600, 721, 1000]))
        li.tair_relax = float(f["tair"][i, min(li.InitLenI, L - 1)] + rs.uniform(-2, 2))
        li.VZ_relax = float(rs.uniform(0.5, 6.0)); li.RH_relax = float(rs.uniform(60, 99))
        if i % 37 == 0:
            li.tair_relax = -9999.0          # an invalid target: no relaxation for the point
        ls.append(li)
    f["tsurfobs"][:, :] = base["tsurf"] + rs.uniform(-1.5, 1.5, (n, 1))   # observations off the model's own track
    f["tsurfobs"][::5, 300:500] = -9999.9                                # gaps inside the initialization phase
    f["tdew"][3, 1500] = 120.0                                           # fails at index 1501
    ora, _, _ = oh.run_oracle("ref" if oh.have_ref() else "port", f, s, p, ls)
    res, nfail = device.run_points(f, s, p, ls, chunk=chunk, precision=32)
    assert nfail == 1
    assert (res["tsurf"][3, 1501:] == -9999.0).all() and (ora["tsurf"][3, 1501:] == -9999.0).all()
    assert res["tsurf"][3, 1500] > -9000 and ora["tsurf"][3, 1500] > -9000
    ok = np.ones(n, bool); ok[3] = False
    d = np.abs(res["tsurf"][ok] - ora["tsurf"][ok])
    print("FULL fp32: tsurf rms %.2e p99.9 %.2e max %.3f frac>0.05K %.1e" %
          (np.sqrt((d ** 2).mean()), np.percentile(d, 99.9), d.max(), (d > 0.05).mean()))
    # Gate, as for the seven-day runs above: the bulk within microkelvins, a handful of points on the other side of a
    # melt-out branch for a short while.  (Measured: p99.9 2e-5 K, 3e-7 of the point-steps beyond 0.05 K - ONE point
    # whose last 0.002 mm of snow melt out an index later than in fp64: melting() then pins Tmp(1:2) to T4Melt, a 3 K
    # step at 7.5 C (src/Storage.f90:376-388), for that one index.  The rms is that point's: not gated.)
    over = d > 0.05
    runs = [0]
    for q in np.where(over.any(1))[0]:
        x = np.flatnonzero(np.diff(np.concatenate([[0], over[q].astype(np.int8), [0]])))
        runs += list(x[1::2] - x[0::2])
    assert np.percentile(d, 99.9) < 3e-4 and over.mean() < 1e-5
    assert over.any(1).mean() < 0.015 and max(runs) * 30 < 2 * 3600 and d.max() < 5.0
    for k in ("snow", "water", "ice", "deposit", "ice2"):
        e = np.abs(res[k][ok] - ora[k][ok])
        assert np.percentile(e, 99.9) < 5e-4 and (e > 0.05).mean() < 1e-5 and e.max() < 0.2, k
    # the features really act: the LEAN run of the same forcing is somewhere else
    assert np.abs(ora["tsurf"][ok] - base["tsurf"][ok]).max() > 0.5


@pytest.mark.parametrize("summer,world,chunk,history", [(False, False, 0, None), (True, False, 97, True), (True, True, 0, False)])
def test_fp32_sky_view_against_the_fp64_reference(summer, world, chunk, history):
    """Round 6, last pass (VERDICT r05 "missing" 6): sky view and local horizons in the fp32 flavour (step_kernel_f32duo<.,
    ., true, true>: src/ModRadiation.f90:7-73, src/SunPosition.f90:123-193, CheckValues' sky-view tests and the SW_dir
    clamp, src/InputOutput.f90:68-77) against the fp64 reference on the inputs of tests/test_hip_skyview.py - winter in
    Finland, midsummer, midsummer anywhere on the globe; horizons up to 25 degrees, sky-view factors 0 ... 1 (1: the
    point has no sky view), SW_dir above SW for some points.  The sun's position and the decisions taken from it are
    fp64 in this flavour too, so the gate is the LEAN flavour's distribution gate (below: or what the flavour does on
    the same forcing without the feature); a point whose SW_dir leaves CheckValues' limits fails at the reference's index.  Both instances (with and without the history score), whole
    series and launches of 97 indices."""
    from roadsurf_amd import device
    from test_hip_skyview import _sky_case
    n, L = 512, 2881
    f, ls = _sky_case(n, L, 41, summer, world)
    for li in ls:
        li.InitLenI = 1
    bad = next(i for i, li in enumerate(ls) if 0.0 <= li.sky_view < 1.0)
    f["sw_dir"][bad, 1200] = 5000.0   # fails at index 1201 (a point without sky view would not: the test is the sky view's)
    s = abi.default_settings(L); p = abi.default_parameters()
    ora, _, _ = oh.run_oracle("ref" if oh.have_ref() else "port", f, s, p, ls)
    res, nfail = device.run_points(f, s, p, ls, chunk=chunk, precision=32, history_score=history)
    assert nfail == 1
    assert (res["tsurf"][bad, 1201:] == -9999.0).all() and (ora["tsurf"][bad, 1201:] == -9999.0).all()
    assert res["tsurf"][bad, 1200] > -9000 and ora["tsurf"][bad, 1200] > -9000
    ok = np.ones(n, bool); ok[bad] = False
    d = np.abs(res["tsurf"][ok] - ora["tsurf"][ok])
    print("sky view fp32: tsurf rms %.2e p99.9 %.2e max %.3f frac>0.05K %.1e" %
          (np.sqrt((d ** 2).mean()), np.percentile(d, 99.9), d.max(), (d > 0.05).mean()))
    # Gate: the LEAN flavour's distribution gate - or, where this forcing (short wave doubled, a midsummer time axis
    # over winter weather) is harder on the fp32 flavour than the bench workload whatever the features, 1.5 x what the
    # flavour does on the SAME forcing without sky view.  Measured (tools/experiments/diag_f32_sky.py): Finland rms
    # 2.0-2.4e-4 K, max 0.08 K; anywhere on the globe rms 1.13e-3 K, max 0.52 K (one point whose last ice melts an index
    # apart) with sky view against 0.96e-3 K, 0.53 K without: the flavour, not the feature.
    l0 = abi.default_local(); l0.InitLenI = 1
    plain, _, _ = oh.run_oracle("port", f, s, p, l0)
    r0, _ = device.run_points(f, s, p, l0, precision=32, lean_if_possible=False)
    d0 = np.abs(r0["tsurf"][ok] - plain["tsurf"][ok])
    print("   no sky view: tsurf rms %.2e p99.9 %.2e max %.3f frac>0.05K %.1e" %
          (np.sqrt((d0 ** 2).mean()), np.percentile(d0, 99.9), d0.max(), (d0 > 0.05).mean()))
    assert np.sqrt((d ** 2).mean()) < max(1e-3, 1.5 * np.sqrt((d0 ** 2).mean()))
    assert np.percentile(d, 99.9) < max(1e-3, 1.5 * np.percentile(d0, 99.9))
    assert d.max() < max(0.5, 1.5 * d0.max()) and (d > 0.05).mean() < max(1e-4, 1.5 * (d0 > 0.05).mean())
    for k in ("snow", "water", "ice", "deposit", "ice2"):
        e = np.abs(res[k][ok] - ora[k][ok]); e0 = np.abs(r0[k][ok] - plain[k][ok])
        assert np.sqrt((e ** 2).mean()) < max(5e-4, 1.5 * np.sqrt((e0 ** 2).mean())), k
        assert np.percentile(e, 99.9) < max(5e-4, 1.5 * np.percentile(e0, 99.9)) and e.max() < max(0.1, 1.5 * e0.max()), k
    # the sky view really acts: without it the same forcing ends somewhere else
    assert np.abs(plain["tsurf"][ok] - ora["tsurf"][ok]).max() > 0.5


@pytest.mark.parametrize("nl,chunk", [(15, 0), (12, 97)])
def test_fp32_output_depth_and_the_full_feature_set_at_any_layer_count(nl, chunk):
    """The general fp32 kernel (step_kernel_f32_coupled, here without coupling) takes what the two-wavefront kernels do not
    have: an output depth - tsurfOutputDepth, or a depth stream with values inside the grid, below it and exactly 0
    (getTempAtDepth, src/BalanceModel.f90:390-417; src/Initialization.f90:129-136 for the first index) - and the FULL
    feature set (initialization phase with observations, relaxation) at NLayers != 15, whole series and launches of 97
    indices.  Against the fp64 reference, the FULL flavour's gate."""
    from roadsurf_amd import device
    n, L, seed = 512, 2881, 23
    f = oh.synth_forcing(n, L, seed=seed)
    p = abi.default_parameters()
    s0 = abi.default_settings(L); s0.NLayers = nl
    l0 = abi.default_local(); l0.InitLenI = 1
    base, _, _ = oh.run_oracle("port", f, s0, p, l0)
    rs = np.random.RandomState(3)
    ls = []
    for i in range(n):
        li = abi.default_local(); li.InitLenI = int(rs.choice([1, 240, 600]))
        li.tair_relax = float(f["tair"][i, min(li.InitLenI, L - 1)] + rs.uniform(-2, 2))
        li.VZ_relax = float(rs.uniform(0.5, 6.0)); li.RH_relax = float(rs.uniform(60, 99))
        ls.append(li)
    f["tsurfobs"][:, :] = base["tsurf"] + rs.uniform(-1.0, 1.0, (n, 1))
    for what in ("depth-setting", "depth-array"):
        g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
        s = abi.default_settings(L); s.NLayers = nl; s.use_relaxation = 1
        if what == "depth-setting":
            s.tsurfOutputDepth = 0.05
        else:
            g["depth"][:] = 0.0; g["depth"][::2] = 0.12; g["depth"][1::4] = 7.0
        g["tdew"][5, 1500] = 120.0  # CheckValues fails the point at index 1501, in the middle of a launch
        ora, _, _ = oh.run_oracle("ref" if oh.have_ref() else "port", g, s, p, ls)
        res, nfail = device.run_points(g, s, p, ls, chunk=chunk, precision=32)
        assert nfail == 1
        assert (res["tsurf"][5, 1501:] == -9999.0).all() and (ora["tsurf"][5, 1501:] == -9999.0).all()
        assert res["tsurf"][5, 1500] > -9000 and ora["tsurf"][5, 1500] > -9000
        for k in res:
            res[k][5] = ora[k][5]
        d = np.abs(res["tsurf"] - ora["tsurf"])
        print("fp32 %s NLayers %d: tsurf rms %.2e p99.9 %.2e max %.3f frac>0.05K %.1e" %
              (what, nl, np.sqrt((d ** 2).mean()), np.percentile(d, 99.9), d.max(), (d > 0.05).mean()))
        assert np.percentile(d, 99.9) < 1e-3 and (d > 0.05).mean() < 1e-4 and d.max() < 5.0
        for k in ("snow", "water", "ice", "deposit", "ice2"):
            e = np.abs(res[k] - ora[k])
            assert np.percentile(e, 99.9) < 5e-4 and (e > 0.05).mean() < 1e-5 and e.max() < 0.2, k
        # the depth really acts: the surface temperature at 5 cm / at the stream's depths is not the two-layer mean
        s1 = abi.default_settings(L); s1.NLayers = nl; s1.use_relaxation = 1
        plain, _, _ = oh.run_oracle("port", f, s1, p, ls)
        assert np.abs(plain["tsurf"] - ora["tsurf"]).max() > 0.1


@pytest.mark.parametrize("precision", [32, 64])
def test_sky_view_reads_the_horizons_through_the_index_row(precision):
    """RsPointParams::horizon_index (a plan in another order than the caller's horizon table): the points handed over in a
    permuted order with the horizon table left as it was and the permutation as index row give every point the values of
    the plain run, bit for bit - two-wavefront kernels of either precision."""
    from roadsurf_amd import device
    from test_hip_skyview import _sky_case
    n, L = 200, 2881  # (a whole day: the sun has to stand above some horizons and below others)
    f, ls = _sky_case(n, L, 5, summer=True)
    for li in ls:
        li.InitLenI = 1
    s = abi.default_settings(L); p = abi.default_parameters()
    plain, _ = device.run_points(f, s, p, ls, precision=precision)
    perm = np.random.RandomState(2).permutation(n)
    g = {k: (v[perm].copy() if isinstance(v, np.ndarray) and v.ndim == 2 and v.shape[0] == n and k != "local_horizons" else v)
         for k, v in f.items()}
    moved, _ = device.run_points(g, s, p, [ls[i] for i in perm], precision=precision, horizon_index=perm)
    for k in plain:
        assert np.array_equal(moved[k], plain[k][perm]), k
    wrong, _ = device.run_points(g, s, p, [ls[i] for i in perm], precision=precision)  # without the index row: other horizons
    assert not np.array_equal(wrong["tsurf"], plain["tsurf"][perm])


def _coupling_gate(res, ora, ls, what):
    """fp32 against fp64 with coupling: the bulk within microkelvins as without coupling; Coupling_control stops a point's
    replays on |Tsurf - observation| <= 0.1 K, so where the two runs straddle that limit one of them replays once more
    and the point differs by up to about that for the rest of the series (measured, tools/experiments/diag_f32_cpl.py:
    one point of 384 in two of the four cases, 0.09 and 0.17 K; rms 5e-6 ... 6e-4 K) - and as many points end their
    window within the limit as in the reference."""
    n = len(ls)
    d = np.abs(res["tsurf"] - ora["tsurf"])
    pm = d.max(1)
    print("%s: tsurf rms %.2e p99.9 %.2e max %.3f frac>0.05K %.1e points > 0.05 K: %d" %
          (what, np.sqrt((d ** 2).mean()), np.percentile(d, 99.9), d.max(), (d > 0.05).mean(), int((pm > 0.05).sum())))
    # (p99.9: with sky view on the doubled short wave of tests/test_hip_skyview.py 5e-3 K - a point or two a few mK off for
    # the rest of a 1 441-index series; everything far inside the 0.1 K Coupling_control itself works to)
    assert np.sqrt((d ** 2).mean()) < 2e-3 and np.percentile(d, 99.9) < 1e-2 and d.max() < 0.5
    assert (d > 0.05).mean() < 5e-4 and (pm > 0.05).mean() < 0.02
    for k in ("snow", "water", "ice", "deposit", "ice2"):
        e = np.abs(res[k] - ora[k])
        assert np.percentile(e, 99.9) < 1e-2 and e.max() < 0.2, k
    ci = np.array([l.couplingIndexI for l in ls]); ct = np.array([l.couplingTsurf for l in ls])
    on = np.flatnonzero((ci >= 1) & (ct > -100))
    met64 = sum(abs(ora["tsurf"][i, ci[i] - 1] - ct[i]) <= 0.1001 for i in on)
    met32 = sum(abs(res["tsurf"][i, ci[i] - 1] - ct[i]) <= 0.1001 for i in on)
    assert met64 > len(on) // 2 and abs(int(met32) - int(met64)) <= max(2, len(on) // 100), (met32, met64)


def test_fp32_coupling_against_the_fp64_reference():
    """Round 6, last pass (VERDICT r05 "missing" 6): coupling in the fp32 flavour (step_kernel_f32_coupled: src/Coupling.f90
    :10-141,172-289,292-481 with every lane on its own time index) on the four configurations of tests/test_hip_coupling.py
    - plain, with relaxation, a one-hour window without initialization phase, windows spread over the series; offsets of
    the observation from 0 to +-15 K, points without a usable observation or index - against the fp64 reference."""
    from roadsurf_amd import device
    from test_hip_coupling import _cases, _kind
    n, L = 384, 2881
    cases, base = _cases(n, L, 4242)
    for k, (f2, s, p, ls) in enumerate(cases):
        ora, _, _ = oh.run_oracle(_kind(), f2, s, p, ls)
        res, nfail = device.run_points(f2, s, p, ls, precision=32)
        assert nfail == 0
        _coupling_gate(res, ora, ls, "fp32 coupling, case %d" % k)
        assert (np.abs(ora["tsurf"] - base["tsurf"]).max(1) > 1e-3).sum() > n // 2  # coupling really acts


def test_fp32_coupling_with_sky_view_and_other_layer_counts():
    """... together with sky view (long-wave scaling only: src/Coupling.f90:68-76), and for a profile of 12 layers (the
    kernel keeps the profile in LDS: any NLayers)."""
    from roadsurf_amd import device
    from test_hip_coupling import _kind
    from test_hip_skyview import _sky_case
    n, SL = 200, 1441
    f, ls = _sky_case(n, SL, 7, summer=True)
    for nl in (15, 12):
        p = abi.default_parameters()
        s0 = abi.default_settings(SL); s0.NLayers = nl
        basel = [abi.LocalParameters.from_buffer_copy(li) for li in ls]
        base, _, _ = oh.run_oracle("port", f, s0, p, basel)
        rs = np.random.RandomState(1)
        ls2 = []
        for i, li in enumerate(basel):
            li.couplingIndexI = 900; li.InitLenI = 900
            li.couplingTsurf = float(base["tsurf"][i, 899] + rs.choice([0.0, 1.0, -2.0, 5.0]))
            ls2.append(li)
        g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
        g["tsurfobs"][:, :] = base["tsurf"] + 0.2
        s = abi.default_settings(SL); s.use_coupling = 1; s.NLayers = nl
        ora, _, _ = oh.run_oracle(_kind(), g, s, p, ls2)
        res, nfail = device.run_points(g, s, p, ls2, precision=32)
        assert nfail == 0
        _coupling_gate(res, ora, ls2, "fp32 coupling + sky view, NLayers %d" % nl)


@pytest.mark.parametrize("forecast", [True, False], ids=["forecast-key", "history-key"])
def test_fp32_full_feature_set_reads_the_knots_like_a_window(forecast):
    """(history-key: the re-sort by the last window's passes - the instances that keep the score.)
    ... and through the object bench.py times: the knot-reading launch of the FULL fp32 flavour (rs_hip_step_knots:
    dew point interpolated like the other variables, the observation of index 1, initialization phase, relaxation
    towards per-plan targets) equals its window launch bit for bit, in plan order."""
    import torch
    from roadsurf_amd import device, workload
    n, hours, chunk = 3000, 8, 97
    L = hours * 120 + 1
    s = abi.default_settings(L); s.use_relaxation = 1
    p = abi.default_parameters()
    series = {}
    for variant in (2, 0):  # 2: windows + one point per lane is LEAN only -> windows through the duo kernel: variant 0 unfused
        plan = device.Plan(n, s, p, 0)
        plan.set_precision(32)
        run = workload.SyntheticRun(plan, 9, hours, chunk, point_offset=777, plan_order=True, f32=True, full=True, initlen=300,
                                    forecast=forecast)
        if variant == 2:
            run.fused = False  # the same kernel from a forcing window (expand_kernel_f32 + rs_hip_step)
            run.win = device.ForcingWindow.empty(run.chunk, plan.np_pad, plan.device, optional=("tdew", "tsurfobs"),
                                                 dtype=torch.float32)
        else:
            assert run.fused
        full_out = {k: torch.full((L, n), float("nan"), dtype=torch.float32, device=plan.device) for k in device.OUT_FIELDS}

        def on_launch(c, t0, ns):
            o = run.orders[c][:n].long()
            for k in device.OUT_FIELDS:
                full_out[k][t0 - 1:t0 - 1 + ns, o] = run.out.tensors[k][:ns, :n]

        run.run_pass(on_launch)
        plan.sync()
        assert plan.failed_count() == 0
        series[variant] = {k: v.cpu().numpy() for k, v in full_out.items()}
        del run
        plan.close()
    for k in series[0]:
        assert not np.isnan(series[0][k]).any()
        assert np.array_equal(series[0][k], series[2][k]), k
