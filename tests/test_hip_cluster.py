"""GPU: plan order (rs_hip_recluster).  Re-sorting the slots of a plan between launches and
generating the windows in slot order must not change any value: outputs mapped back through
the order array are bit-identical to the checker's natural-order run."""
import numpy as np
import pytest
import torch

import oracle_helpers as oh
from roadsurf_amd import abi, device

pytestmark = pytest.mark.gpu


def test_reclustered_run_is_bit_identical_and_really_reorders():
    n, L, chunk, seed, spk = 3000, 1441, 120, 424242, 120
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    f = oh.synth_forcing(n, L, seed=seed)
    ora, _, _ = oh.run_oracle("port", f, s, p, l)

    plan = device.Plan(n, s, p, 0)
    dev, npad = plan.device, plan.np_pad
    spec, _ = plan.synth_knots(seed, 2, steps_per_knot=spk)
    kbuf = torch.empty((chunk // spk + 3, 9, npad), dtype=torch.float64, device=dev)
    win = device.ForcingWindow.empty(chunk, npad, dev, optional=())
    win0 = device.ForcingWindow.empty(1, npad, dev, optional=("tsurfobs",))
    out = device.OutputWindow.empty(chunk, npad, dev)
    pp = plan.point_params(plan.uniform_tbottom(2024, 1, 10))
    res = {k: np.full((n, L), np.nan) for k in oh.F64_OUT}
    names = dict(tsurf="tsurf", snow="snow", water="water", ice="ice", deposit="deposit", ice2="ice2")

    order0 = plan.order().clone()
    assert torch.equal(order0[:n].cpu(), torch.arange(n, dtype=torch.int32))
    plan.synth_knots_range(spec, kbuf, 0, 2, ordered=True)
    plan.expand_range(spec, kbuf, 0, 2, win0, 1, 1)
    plan.init_state(win0, pp)
    moved = 0
    for t0 in range(1, L + 1, chunk):
        ns = min(chunk, L - t0 + 1)
        k0 = (t0 - 1) // spk
        nk = (t0 + ns - 2) // spk + 1 - k0 + 1
        order = plan.order().clone()          # the order this window is produced and stepped in
        plan.synth_knots_range(spec, kbuf, k0, nk, ordered=True)
        plan.expand_range(spec, kbuf, k0, nk, win, t0, ns)
        plan.step(win, out, pp, t0, ns, out_row0=t0 - 1)
        plan.sync()
        idx = order[:n].cpu().numpy()
        assert sorted(idx.tolist()) == list(range(n))          # a permutation of the points
        moved += int((idx != np.arange(n)).sum())
        for k, attr in names.items():
            rows = out.tensors[attr][:ns, :n].cpu().numpy()    # [t][slot]
            res[k][idx, t0 - 1:t0 - 1 + ns] = rows.T
        plan.recluster()
    assert moved > n          # the order did change along the way
    for k in oh.F64_OUT:
        assert np.array_equal(res[k], ora[k]), (k, int((res[k] != ora[k]).sum()))
    plan.close()


def test_recluster_sorts_by_the_score_slot():
    from roadsurf_amd import lib
    n = 1000
    s = abi.default_settings(10); p = abi.default_parameters()
    plan = device.Plan(n, s, p, 0)
    st = plan.state()
    rs = np.random.RandomState(0)
    score = rs.randint(0, 500, n).astype(np.float64)
    nst = st.shape[0]
    slot = lib.RS_MAX_LAYERS + 16          # RS_ST_BLSCORE
    st[:] = 0
    st[slot, :n] = torch.from_numpy(score)
    st[0, :n] = torch.arange(n, dtype=torch.float64) * 10   # a model-state row travels with its point
    plan.load_state(st)
    plan.order()
    plan.recluster()
    plan.sync()
    st2 = plan.state()
    order = plan.order()[:n].cpu().numpy()
    # descending (the expensive points get the low slots: their workgroups start first), stable
    want = np.argsort(-score, kind="stable")
    assert np.array_equal(st2[slot, :n].numpy(), score[want])
    assert np.array_equal(order, want)
    assert np.array_equal(st2[0, :n].numpy(), order * 10.0)
    assert nst == st2.shape[0]
    plan.close()


def test_full_feature_run_in_plan_order_with_per_point_parameters():
    """The documented contract for callers: after a recluster, windows AND per-point parameter
    arrays are indexed by slot.  Relaxation + observation forcing + Tdew check (FULL kernel),
    windows gathered through the order on the device, recluster after every chunk."""
    n, L, chunk = 1500, 1441, 97
    rs = np.random.RandomState(12)
    s = abi.default_settings(L); s.use_relaxation = 1
    p = abi.default_parameters()
    f = oh.synth_forcing(n, L, seed=99)
    base_l = abi.default_local(); base_l.InitLenI = 1
    base, _, _ = oh.run_oracle("port", f, abi.default_settings(L), p, base_l)
    f["tsurfobs"] = np.ascontiguousarray(base["tsurf"] + rs.uniform(-1, 1, (n, 1)))
    ls = []
    for i in range(n):
        li = abi.default_local()
        li.InitLenI = int(rs.randint(2, L // 2))
        li.tair_relax = float(f["tair"][i, li.InitLenI] + rs.uniform(-3, 3))
        li.VZ_relax = float(rs.uniform(0.5, 9)); li.RH_relax = float(rs.uniform(40, 100))
        ls.append(li)
    ora, _, _ = oh.run_oracle("port", f, s, p, ls)

    plan = device.Plan(n, s, p, 0)
    dev, npad = plan.device, plan.np_pad

    def dev_rows(a, dtype):      # host [n][L] -> device [L][npad] in NATURAL order
        t = torch.zeros((L, npad), dtype=dtype, device=dev)
        t[:, :n] = torch.from_numpy(np.ascontiguousarray(a)).to(dev).T
        return t
    nat = {k: dev_rows(f[k], torch.float64) for k in ("tair", "tdew", "vz", "rhz", "prec", "sw", "lw", "tsurfobs")}
    nat["precphase"] = dev_rows(f["precphase"], torch.int32)
    hour = torch.from_numpy(np.ascontiguousarray(f["hour"])).to(dev)

    def vec(vals, dtype):
        t = torch.zeros((npad,), dtype=dtype, device=dev)
        t[:n] = torch.tensor(vals, dtype=dtype)
        return t
    pnat = dict(initlen=vec([l.InitLenI for l in ls], torch.int32),
                tair=vec([l.tair_relax for l in ls], torch.float64),
                vz=vec([l.VZ_relax for l in ls], torch.float64),
                rh=vec([l.RH_relax for l in ls], torch.float64))
    tbot = plan.uniform_tbottom(int(f["year"][0]), int(f["month"][0]), int(f["day"][0]))
    out = device.OutputWindow.empty(chunk, npad, dev)
    res = {k: np.full((n, L), np.nan) for k in oh.F64_OUT}
    for t0 in range(1, L + 1, chunk):
        ns = min(chunk, L - t0 + 1)
        order = plan.order().clone().long()
        tens = {k: v[t0 - 1:t0 - 1 + ns].index_select(1, order).contiguous() for k, v in nat.items()}
        tens["hour"] = hour[t0 - 1:t0 - 1 + ns].contiguous()
        tens["depth"] = None
        win = device.ForcingWindow(ns, npad, tens)
        pp = plan.point_params(tbot, pnat["initlen"][order].contiguous(), pnat["tair"][order].contiguous(),
                               pnat["vz"][order].contiguous(), pnat["rh"][order].contiguous())
        if t0 == 1:
            plan.init_state(win, pp)
        plan.step(win, out, pp, t0, ns, out_row0=t0 - 1)
        plan.sync()
        idx = order[:n].cpu().numpy()
        for k in oh.F64_OUT:
            res[k][idx, t0 - 1:t0 - 1 + ns] = out.tensors[k][:ns, :n].cpu().numpy().T
        plan.recluster()
    for k in oh.F64_OUT:
        assert np.array_equal(res[k], ora[k]), (k, int((res[k] != ora[k]).sum()))
    plan.close()


def test_forecast_recluster_is_a_value_neutral_permutation():
    """rs_hip_recluster_forecast (the sort key bench.py uses): whatever the key says, slots are
    permuted, never changed - outputs mapped back through the order rows equal the checker's
    natural-order run bit for bit - and the most expensive points come first."""
    import torch
    import oracle_helpers as oh
    from roadsurf_amd import abi, device, workload
    n, hours = 3000, 12
    L = hours * 120 + 1
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    f = oh.synth_forcing(n, L, seed=99)
    ora, _, _ = oh.run_oracle("ref" if oh.have_ref() else "port", f, s, p, l)
    plan = device.Plan(n, s, p, 0)
    run = workload.SyntheticRun(plan, 99, hours, 120, plan_order=True, forecast=True)
    got = {k: np.full((n, L), np.nan) for k in device.OUT_FIELDS}
    moved = []

    def on_launch(c, t0, ns):
        order = run.orders[c][:n].long()
        assert int(torch.bincount(order, minlength=n).max()) == 1
        moved.append(int((order != torch.arange(n, device=order.device)).sum()))
        idx = order.cpu().numpy()
        for k in device.OUT_FIELDS:
            got[k][idx, t0 - 1:t0 - 1 + ns] = run.out.tensors[k][:ns, :n].cpu().numpy().T
    run.run_pass(on_launch)
    plan.sync()
    for k in device.OUT_FIELDS:
        assert np.array_equal(got[k], ora[k]), k
    assert moved[0] == 0 and max(moved) > n // 2
    # a plan that gave up its history score cannot be sorted by history
    with pytest.raises(RuntimeError, match="history score"):
        plan.recluster()
    plan.close()


@pytest.mark.parametrize("n", [1000, 70_001])
def test_the_plans_own_sort_equals_the_library_sort(n, monkeypatch):
    """A forecast key of at most 12 bits (the default field set) is sorted by the plan's own stable
    counting pass - per-tile histograms, one scan, a scatter that ranks equal keys by ballot - instead
    of the library's merge sort.  Both are stable, so every launch must see the SAME slot order with
    either (ROADSURF_HIP_LIBRARY_SORT=1 forces the library), whatever the batch size does to the tiles
    of 1 024 keys (1 000 points: one partial tile; 70 001: 69 tiles, the last with 369 keys)."""
    import torch
    from roadsurf_amd import abi, device, workload
    hours = 4
    L = hours * 120 + 1
    s = abi.default_settings(L); p = abi.default_parameters()
    rows = {}
    for tag, env in (("own", None), ("library", "1")):
        if env:
            monkeypatch.setenv("ROADSURF_HIP_LIBRARY_SORT", env)
        plan = device.Plan(n, s, p, 0)
        run = workload.SyntheticRun(plan, 5, hours, 60, plan_order=True, forecast=True)
        assert run.forecast_mode == workload.DEFAULT_FORECAST_MODE
        run.run_pass()
        plan.sync()
        rows[tag] = run.orders[:, :n].cpu().numpy().copy()
        cs = plan.failed_count()
        assert cs == 0
        plan.close()
    assert (rows["own"][0] == np.arange(n)).all() and (rows["own"][-1] != np.arange(n)).any()
    assert np.array_equal(rows["own"], rows["library"])
    for r in rows["own"]:
        assert np.array_equal(np.sort(r), np.arange(n))


def test_previews_between_two_rows_sort_like_the_blended_rows():
    """RsPreview::tair_b / vz_b / w (ABI 9): a preview between two rows - the caller's hourly knots - is the
    straight line between them, so the slots must come out in the order that rows blended by the caller give;
    and that order differs from the one the knots themselves give (the previews really moved)."""
    import torch
    from roadsurf_amd import abi, device, workload
    n, hours = 5000, 3
    L = hours * 120 + 1
    s = abi.default_settings(L); p = abi.default_parameters()
    orders = {}
    for tag in ("between", "blended", "knots"):
        plan = device.Plan(n, s, p, 0)
        run = workload.SyntheticRun(plan, 7, hours, 60, plan_order=True, forecast=True)
        run.run_pass()  # some history in the state, and an order that is not the identity (the same for all three)
        plan.sync()
        kn = run.knots
        ws = (0.0, 0.25, 0.4916666666666667)
        hrs = [3, 3, 3]
        if tag == "between":
            plan.recluster_forecast([kn[1, 0]] * 3, [kn[1, 2]] * 3, hrs, None, 0.5, workload.DEFAULT_FORECAST_MODE,
                                    point_order=True, between=[(kn[2, 0], kn[2, 2], w) for w in ws])
        elif tag == "blended":
            ta = [kn[1, 0] + w * (kn[2, 0] - kn[1, 0]) for w in ws]
            vz = [kn[1, 2] + w * (kn[2, 2] - kn[1, 2]) for w in ws]
            torch.cuda.synchronize()
            plan.recluster_forecast(ta, vz, hrs, ta[0], 0.5, workload.DEFAULT_FORECAST_MODE, point_order=True)
        else:
            plan.recluster_forecast([kn[1, 0], kn[2, 0]], [kn[1, 2], kn[2, 2]], hrs[:2], kn[1, 0], 0.5,
                                    workload.DEFAULT_FORECAST_MODE, point_order=True)
        plan.sync()
        orders[tag] = plan.order()[:n].cpu().numpy().copy()
        plan.close()
    assert np.array_equal(np.sort(orders["between"]), np.arange(n))
    assert np.array_equal(orders["between"], orders["blended"])
    assert not np.array_equal(orders["between"], orders["knots"])


def test_outputs_by_point_are_the_slot_rows_in_point_major_order():
    """rs_hip_outputs_by_point (ABI 10): the rows of a launch, [row][slot] in plan order, as per-point series
    [point][row] - every value where its point's series has it, ragged sizes (a last partial wavefront, more
    rows than one tile of 32), rows placed at an offset of the series, a kept order row or the plan's own."""
    import torch
    from roadsurf_amd import abi, device, workload
    n, hours, chunk = 2500, 3, 90
    L = hours * 120 + 1
    s = abi.default_settings(L); p = abi.default_parameters()
    plan = device.Plan(n, s, p, 0)
    run = workload.SyntheticRun(plan, 3, hours, chunk, plan_order=True, forecast=True)
    series = {k: torch.full((plan.np_pad, L + 7), -1.0, dtype=torch.float64, device=plan.device) for k in device.OUT_FIELDS}
    kept = {}

    def on_launch(c, t0, ns):
        # the plan's current order is still the launch's; every second launch through the kept row instead
        order = run.orders[c] if c % 2 else None
        plan.outputs_by_point(run.out, ns, series, dst_row0=t0 - 1 + 7, order=order)
        kept[c] = {k: run.out.tensors[k][:ns, :n].clone() for k in device.OUT_FIELDS}
    run.run_pass(on_launch)
    plan.sync()
    assert len(kept) > 3
    for c, t0 in enumerate(run.starts):
        order = run.orders[c][:n].long()
        ns = min(chunk, L - t0 + 1)
        for k in device.OUT_FIELDS:
            got = series[k][order, t0 - 1 + 7:t0 - 1 + 7 + ns].T
            assert torch.equal(got.contiguous(), kept[c][k]), (c, k)
    for k in device.OUT_FIELDS:
        assert bool((series[k][:n, :7] == -1.0).all()) and bool((series[k][n:] == -1.0).all())  # nothing else was touched
    plan.close()
