"""GPU: the HIP path against the committed golden fixtures (captured from the
reference), and size-independent properties at BASELINE.json's full size
(1 000 000 points x 48 h) where no oracle run is affordable:
  * shard independence: a point's series does not depend on which batch / GPU
    shard it is computed in (bit-identical),
  * determinism: two passes give identical bits (wrap-around checksum of checksums),
  * range invariants: storages within their configured limits, |Tsurf| <= 100, no NaN,
  * a sampled oracle check at the full-size run's global point ids.
Tolerance vs the reference: 1e-6 K / 1e-6 mm (BASELINE.json north_star)."""
import numpy as np
import pytest

import golden_helpers as gh
import oracle_helpers as oh
from roadsurf_amd import abi

pytestmark = pytest.mark.gpu
SPK = 120
TOL = 1e-6


def test_golden_scenarios_on_gpu():
    from roadsurf_amd import device
    z = gh.load("e2e_scenarios.npz")
    K = {k[5:]: z[k] for k in z.files if k.startswith("knot_")}
    L = 48 * SPK + 1
    f = gh.expand_knots(K, L, SPK)
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    idx = z["out_index"]
    for variant in (1, 2, 3):
        res, nfail = device.run_points(f, s, p, l, variant=variant)
        assert nfail == 1
        for k in oh.F64_OUT:
            want = z[f"out_{k}"]
            got = res[k][:, idx]
            assert np.array_equal(want == -9999.0, got == -9999.0), k
            assert np.abs(np.where(want == -9999.0, 0, got - want)).max() < TOL, k
            assert np.array_equal(want, got), (k, "within tolerance but not bit-identical")


def test_golden_feature_cases_on_gpu():
    from roadsurf_amd import device
    z = gh.load("e2e_features.npz")
    K = {k[5:]: z[k] for k in z.files if k.startswith("knot_")}
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
        res, _ = device.run_points(f, s, p, ls)
        for k in oh.F64_OUT:
            assert np.abs(res[k][:, idx] - z[f"{tag}_{k}"]).max() < TOL, (tag, k)
            assert np.array_equal(res[k][:, idx], z[f"{tag}_{k}"]), (tag, k, "not bit-identical")

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


def test_golden_coupling_on_gpu():
    from roadsurf_amd import device
    from test_oracle_vs_golden import _coupling_case
    z, f, s, p, ls = _coupling_case()
    res, _ = device.run_points(f, s, p, ls)
    idx = z["out_index"]
    for k in oh.F64_OUT:
        assert np.array_equal(res[k][:, idx], z[f"cpl_{k}"]), k


def test_golden_skyview_on_gpu():
    from roadsurf_amd import device
    from test_oracle_vs_golden import _skyview_case
    z, f, s, p, ls = _skyview_case()
    res, _ = device.run_points(f, s, p, ls)
    idx = z["out_index"]
    for k in oh.F64_OUT:
        assert np.array_equal(res[k][:, idx], z[f"sky_{k}"]), k


def _synthetic_pass(run, sample_points, itype=None):
    """One full pass of roadsurf_amd.workload.SyntheticRun - the object bench.py times.
    Returns (order-independent wrap-around checksum of all six outputs, the sampled points'
    series [field][simlen][npoints_sampled] mapped back through the per-launch order rows,
    mins, maxs)."""
    import torch
    from roadsurf_amd import device
    plan = run.plan
    dev = plan.device
    n = plan.npoints
    itype = itype or torch.int64   # integer view of the output element: int32 for fp32 plans
    acc = {"sum": torch.zeros((), dtype=torch.int64, device=dev)}
    mins = {k: float("inf") for k in device.OUT_FIELDS}
    maxs = {k: float("-inf") for k in device.OUT_FIELDS}
    pts = torch.as_tensor(sample_points, device=dev)
    sampled = {k: [] for k in device.OUT_FIELDS}

    def on_launch(c, t0, ns):
        cols = run.slots_of(c, pts)
        if run.plan_order:  # the kept order row is a permutation of the shard's points
            o = run.orders[c][:n].long()
            assert int(o.min()) == 0 and int(o.max()) == n - 1
            assert int(torch.bincount(o, minlength=n).max()) == 1
        for k in device.OUT_FIELDS:
            o = run.out.tensors[k][:ns, :n]
            acc["sum"] += o.view(itype).sum(dtype=torch.int64)
            mins[k] = min(mins[k], float(o.min())); maxs[k] = max(maxs[k], float(o.max()))
            assert not torch.isnan(o).any()
            sampled[k].append(o[:, cols].clone())

    run.run_pass(on_launch)
    plan.sync()
    return (int(acc["sum"].item()), {k: torch.cat(v).cpu().numpy() for k, v in sampled.items()},
            mins, maxs)


def _interleaved_plans(K, chunk, variants, n, s, p, seed, hours, cols, c_want, samp_want, dev):
    """n points as K plans (contiguous blocks at their global offsets, sharding.strong_shard) on K
    streams whose launches interleave, in plan order: the K wrap-around checksums must add up to the
    single plan's and the sampled points - found in whichever plan holds them, through that plan's
    order rows - must carry the same bits."""
    import torch
    from roadsurf_amd import device, sharding, workload
    L = hours * SPK + 1
    plans, runs, offs = [], [], []
    for j in range(K):
        off, cnt = sharding.strong_shard(n, K, j)
        pl = device.Plan(cnt, s, p, dev.index, stream=torch.cuda.Stream(dev))
        if variants[j]:
            pl.set_variant(variants[j])
        plans.append(pl); offs.append(off)
        runs.append(workload.SyntheticRun(pl, seed, hours, chunk, point_offset=off, plan_order=True))
    sums = [torch.zeros((), dtype=torch.int64, device=dev) for _ in range(K)]
    got = {k: np.full((L, len(cols)), np.nan) for k in device.OUT_FIELDS}
    pending = []

    def hook(j):
        run_j, off, cnt = runs[j], offs[j], plans[j].npoints
        mine = [(q, c - off) for q, c in enumerate(cols) if off <= c < off + cnt]
        local = torch.as_tensor([c for _, c in mine], device=dev, dtype=torch.long)

        def on_launch(c, t0, ns):
            with torch.cuda.stream(plans[j].stream):
                slots = run_j.slots_of(c, local) if len(mine) else None
                for k in device.OUT_FIELDS:
                    o = run_j.out.tensors[k][:ns, :cnt]
                    sums[j] += o.view(torch.int64).sum()
                    if len(mine):
                        pending.append((k, t0, ns, [q for q, _ in mine], o[:, slots].clone()))
        return on_launch

    its = [r.iter_pass(hook(j)) for j, r in enumerate(runs)]
    while its:
        its = [it for it in its if next(it, None) is not None]
    torch.cuda.synchronize()
    assert (sum(int(x.item()) for x in sums) - c_want) % (1 << 64) == 0, f"{K} plans: checksum"   # wrap-around sums
    for k, t0, ns, qs, block in pending:
        got[k][t0 - 1:t0 - 1 + ns, qs] = block.cpu().numpy()
    for k in device.OUT_FIELDS:
        assert np.array_equal(got[k], samp_want[k]), (f"{K} interleaved plans", k)
    assert sum(pl.failed_count() for pl in plans) == 0
    del runs
    for pl in plans:
        pl.close()
    torch.cuda.empty_cache()


def test_full_size_properties_1M_points_48h():
    """BASELINE config 3 at its size, in BOTH orders bench.py times: natural order, and plan order
    (slots re-sorted after every launch, windows generated in slot order, order rows kept)."""
    import torch
    from roadsurf_amd import device, workload
    n, hours, seed, chunk = 1_000_000, 48, 20240110, 240
    L = hours * SPK + 1
    s = abi.default_settings(L); p = abi.default_parameters()
    plan = device.Plan(n, s, p, 0)
    # sampled global ids: a contiguous block per region so that they can be re-run as shards
    blocks = [0, 333_312, 999_744]
    cols = np.concatenate([np.arange(b, b + 64) for b in blocks])
    run = workload.SyntheticRun(plan, seed, hours, chunk, plan_order=False)
    c1, samp1, mins, maxs = _synthetic_pass(run, cols)
    c2, samp2, _, _ = _synthetic_pass(run, cols)
    assert c1 == c2, "two passes over the same inputs differ"
    for k in device.OUT_FIELDS:
        assert np.array_equal(samp1[k], samp2[k])
    assert plan.failed_count() == 0
    # range invariants (storage limits: examples/example1/src/InputParameters.h:78-81, MaxWatmms)
    assert -100.0 <= mins["tsurf"] and maxs["tsurf"] <= 100.0
    for k, hi in (("snow", p.MaxSnowmms), ("water", p.MaxWatmms), ("ice", p.MaxIcemms),
                  ("deposit", p.MaxDepmms), ("ice2", p.MaxIcemms)):
        assert mins[k] >= 0.0 and maxs[k] <= hi, (k, mins[k], maxs[k])
    assert maxs["snow"] > 1 and maxs["ice"] > 1 and maxs["deposit"] > 0.1  # workload is not trivial
    del run
    torch.cuda.empty_cache()
    # plan order, exactly as bench.py's headline leg runs it.  Points do not interact, so the
    # multiset of outputs per launch is the same: the order-independent checksum must match, and
    # the sampled points, found through the kept order rows, must carry the same bits
    run = workload.SyntheticRun(plan, seed, hours, chunk, plan_order=True)
    c3, samp3, mins3, maxs3 = _synthetic_pass(run, cols)
    assert c3 == c1, "plan-order pass: checksum of all outputs differs from the natural-order pass"
    assert mins3 == mins and maxs3 == maxs
    for k in device.OUT_FIELDS:
        assert np.array_equal(samp3[k], samp1[k]), ("plan order", k)
    moved = int((run.orders[-1][:n].long() != torch.arange(n, device=plan.device)).sum())
    assert moved > n // 2, "the slots were never re-sorted: this pass did not test plan order"
    c4, samp4, _, _ = _synthetic_pass(run, cols)  # a second pass starts from the identity again
    assert c4 == c1
    assert plan.failed_count() == 0
    del run
    plan.close()
    torch.cuda.empty_cache()
    # bench.py's round-3 default at this size (and the shape of its natural-order leg): FOUR plans of
    # 250 000 points on four streams whose launches interleave (one plan's window expansion and re-sort
    # under another's step kernel), one point per lane.  Same
    # points, same values: the four checksums add up to the single plan's, the sampled blocks -
    # found in whichever plan holds them, through that plan's order rows - carry the same bits
    _interleaved_plans(4, 120, [1] * 4, n, s, p, seed, hours, cols, c1, samp1, plan.device)
    # bench.py's default at this size since round 4: THREE plans of 333 333 / 333 334 points, launches of 60
    # indices (windows that start between two knots), stepped by the two-wavefront flavour whose ground wave
    # makes the forcing from the knots (no expansion kernel, no forcing window: rs_hip_step_knots)
    _interleaved_plans(3, 60, [3] * 3, n, s, p, seed, hours, cols, c1, samp1, plan.device)
    _interleaved_plans(2, 90, [3] * 2, n, s, p, seed, hours, cols, c1, samp1, plan.device)
    # BASELINE config 4's partition on the hardware at hand: the EIGHT blocks of 125 000 points the
    # eight ranks of a node would hold, at their global offsets, as eight plans on this one GPU -
    # four of them stepped by the two-wavefront flavour (what bench.py picks at that shard size),
    # four with one point per lane
    _interleaved_plans(8, 240, [3, 1] * 4, n, s, p, seed, hours, cols, c1, samp1, plan.device)
    # shard independence + reference spot check: re-run each sampled block as its own tiny batch
    l = abi.default_local(); l.InitLenI = 1
    for bi, b in enumerate(blocks):
        f = oh.synth_forcing(64, L, seed=seed, point_offset=b)
        res, _ = device.run_points(f, s, p, l)
        ora, _, _ = oh.run_oracle("ref" if oh.have_ref() else "port", f, s, p, l)
        for k in device.OUT_FIELDS:
            big = samp1[k][:, bi * 64:(bi + 1) * 64].T  # [64][L]
            assert np.array_equal(big, res[k]), ("shard independence", k, b)
            assert np.abs(big - ora[k]).max() < TOL, ("oracle", k, b)
            assert np.array_equal(samp3[k][:, bi * 64:(bi + 1) * 64].T, ora[k]), \
                ("plan order vs reference, bit for bit", k, b)


@pytest.mark.parametrize("full", [False, True], ids=["lean", "full"])
@pytest.mark.parametrize("n,hours,chunk", [(5000, 6, 7), (5000, 6, 60), (4097, 5, 121), (64, 3, 240), (70000, 3, 90)])
def test_knot_reading_flavour_matches_window_flavour(n, hours, chunk, full):
    _knots_against_windows(n, hours, chunk, full)


@pytest.mark.parametrize("full", [False, True], ids=["lean", "full"])
@pytest.mark.parametrize("key", ["history", "wave_table"])
def test_knot_reading_flavour_under_the_other_sort_keys(key, full):
    """The same comparison with the plan re-sorted by the HISTORY of the last launch (rs_hip_recluster: the step
    kernels then keep the history score - the SCORE instances of step_kernel_duo with the knots as source) and by
    a forecast key of twelve bits (fields 3, 1, 2, 5 + the precipitation bit), for which every class of the key starts a wavefront of its own
    (rs_cluster_wave_table, cs_wave_table_kernel: the default 10/11-bit keys take no table)."""
    kw = dict(forecast=False) if key == "history" else dict(forecast_mode=31259)
    _knots_against_windows(5000, 6, 60, full, **kw)


def _knots_against_windows(n, hours, chunk, full, **run_kw):
    """rs_hip_step_knots (the two-wavefront flavour's ground wave interpolates the forcing from the resident
    knots) against the expansion kernel + forcing window + one point per lane, both in plan order: every
    output of every point at every index carries the same bits, whatever the launch length (windows that
    start on, before and between knots; ragged last launch; point counts off the wavefront size).  full:
    the FULL feature set of bench.py's extra leg - dew point and observation streams, an initialization
    phase that ends inside the run, relaxation behind it."""
    import torch
    from roadsurf_amd import device, workload
    L = hours * SPK + 1
    s = abi.default_settings(L); p = abi.default_parameters()
    if full:
        s.use_relaxation = 1
    series = {}
    for variant in (1, 3):
        plan = device.Plan(n, s, p, 0)
        plan.set_variant(variant)
        run = workload.SyntheticRun(plan, 77, hours, chunk, point_offset=12345, plan_order=True, full=full,
                                    initlen=200, **run_kw)
        assert run.fused == (variant == 3)
        full_out = {k: torch.full((L, n), float("nan"), dtype=torch.float64, device=plan.device) for k in device.OUT_FIELDS}

        def on_launch(c, t0, ns):
            o = run.orders[c][:n].long()
            for k in device.OUT_FIELDS:
                full_out[k][t0 - 1:t0 - 1 + ns, o] = run.out.tensors[k][:ns, :n]

        run.run_pass(on_launch)
        plan.sync()
        assert plan.failed_count() == 0
        series[variant] = {k: v.cpu().numpy() for k, v in full_out.items()}
        if full:  # the relaxation anchors went to the state block (in plan order: compare as multisets)
            st = plan.state().numpy()
            series[variant]["anchors"] = np.sort(st[abi.RS_MAX_LAYERS + 13:abi.RS_MAX_LAYERS + 16, :n], axis=1)
        moved = int((run.orders[-1][:n].long() != torch.arange(n, device=plan.device)).sum())
        assert n < 1000 or moved > 0
        del run
        plan.close()
    for k in series[1]:
        assert not np.isnan(series[3][k]).any()
        assert np.array_equal(series[1][k], series[3][k]), k
