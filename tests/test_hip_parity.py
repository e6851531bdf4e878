"""GPU parity: HIP path (through the C-ABI) vs the CPU oracle on identical forcing.

Contract (BASELINE.json north_star): max|dTsurf| < 1e-6 K over 48 h (we hold the storages,
mm, to the same 1e-6).  What we actually require is stronger: BIT EQUALITY.  Every operation
on the path is IEEE-exact in the reference's evaluation order, and the device exp/log return
glibc's bits (roadsurf_amd/csrc/rs_math.hpp explains why nothing less survives 1e6 points:
melt-out steps branch on the sign of a rounding residual).  Bit equality presumes the host
libm runs its FMA variants (any x86-64 CPU since 2013); on another host the 1e-6 gate still
has to hold and the count of non-identical points is printed.
"""
import numpy as np
import pytest

import oracle_helpers as oh
from roadsurf_amd import abi

pytestmark = pytest.mark.gpu

TOL_K = 1e-6
TOL_MM = 1e-6
try:
    HOST_HAS_FMA = " fma " in open("/proc/cpuinfo").read()
except OSError:
    HOST_HAS_FMA = False


def _oracle_kind():
    return "ref" if oh.have_ref() else "port"


def _compare(res, ora, tag):
    worst = {}
    for k in oh.F64_OUT:
        a, b = res[k], ora[k]
        miss = (b == -9999.0)
        assert np.array_equal(miss, a == -9999.0), f"{tag}: missing-value pattern differs in {k}"
        d = np.abs(np.where(miss, 0.0, a - b))
        worst[k] = float(d.max())
    print(tag, {k: f"{v:.3e}" for k, v in worst.items()})
    assert worst["tsurf"] < TOL_K, (tag, worst)
    for k in ("snow", "water", "ice", "deposit", "ice2"):
        assert worst[k] < TOL_MM, (tag, worst)
    if HOST_HAS_FMA:
        for k in oh.F64_OUT:
            assert np.array_equal(res[k], ora[k]), f"{tag}: {k} is within tolerance but not bit-identical"
    return worst


@pytest.mark.parametrize("variant", [1, 2, 3], ids=["reg", "lds", "duo"])
def test_lean_48h_vs_oracle(variant):
    from roadsurf_amd import device
    n, L = 512, 5761
    f = oh.synth_forcing(n, L, seed=1234)
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    ora, _, _ = oh.run_oracle(_oracle_kind(), f, s, p, l)
    res, nfail = device.run_points(f, s, p, l, variant=variant)
    assert nfail == 0
    _compare(res, ora, f"lean-{variant}")
    # storages must actually be exercised by this workload
    assert ora["snow"].max() > 0.5 and ora["ice"].max() > 0.5 and ora["deposit"].max() > 0.1


@pytest.mark.parametrize("variant", [1, 2, 3], ids=["reg", "lds", "duo"])
def test_chunked_equals_whole(variant):
    """Time-chunked stepping (state parked in HBM between launches) is bit-identical
    to one launch over the whole series."""
    from roadsurf_amd import device
    n, L = 300, 1441
    f = oh.synth_forcing(n, L, seed=77)
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    whole, _ = device.run_points(f, s, p, l, variant=variant)
    parts, _ = device.run_points(f, s, p, l, variant=variant, chunk=97)
    for k in oh.F64_OUT:
        assert np.array_equal(whole[k], parts[k]), k


@pytest.mark.parametrize("variant", [1, 2, 3], ids=["reg", "lds", "duo"])
def test_failed_points_in_every_flavour(variant):
    """CheckValues failures (src/InputOutput.f90:45-84) at the first index, inside a launch, at a launch
    boundary and at the last checked index, in the first and in the second half of a wavefront's
    columns and in the ragged last workgroup: the failing index keeps its row, the rows behind it read
    -9999.0, the point's neighbours are untouched, launch by launch (chunk 97) as in one launch."""
    from roadsurf_amd import device
    n, L = 300, 721
    f = oh.synth_forcing(n, L, seed=31)
    bad = {0: 0, 5: 96, 70: 97, 131: 350, 200: 98, 257: 719, 299: 193, 64: 194, 190: 1}
    for pt, idx in bad.items():
        f["tair"][pt, idx] = 250.0
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    ora, _, _ = oh.run_oracle(_oracle_kind(), f, s, p, l)
    for chunk in (0, 97):
        res, nfail = device.run_points(f, s, p, l, variant=variant, chunk=chunk)
        assert nfail == len(bad)
        _compare(res, ora, f"failed-{variant}-{chunk}")
    for pt, idx in bad.items():
        assert (ora["tsurf"][pt, idx + 1:] == -9999.0).all() and ora["tsurf"][pt, idx] != -9999.0


@pytest.mark.parametrize("variant", [1, 2, 3], ids=["reg", "lds", "duo"])
def test_full_variant_features(variant):
    """Init-phase observation forcing, relaxation, output depth, failures.  (The two-wavefront flavour has
    the FULL feature set without an output depth: those two cases fall back to one point per lane.)"""
    from roadsurf_amd import device
    n, L = 200, 5761
    f = oh.synth_forcing(n, L, seed=7)
    f["tsurfobs"][:, :2880] = f["tair"][:, :2880] - 0.7
    f["tsurfobs"][::3, 100:200] = -9999.9
    ls = []
    for i in range(n):
        li = abi.default_local(); li.InitLenI = 2880
        li.tair_relax = float(f["tair"][i, 2880]) + 1.5; li.VZ_relax = 3.0; li.RH_relax = 85.0
        if i % 5 == 0:
            li.tair_relax = -9999.0
        ls.append(li)
    p = abi.default_parameters()
    kind = _oracle_kind()
    # relaxation
    s = abi.default_settings(L); s.use_relaxation = 1
    ora, _, _ = oh.run_oracle(kind, f, s, p, ls)
    res, _ = device.run_points(f, s, p, ls, variant=variant)
    _compare(res, ora, "relax")
    # output depth from settings
    s = abi.default_settings(L); s.tsurfOutputDepth = 0.05
    ora, _, _ = oh.run_oracle(kind, f, s, p, ls)
    res, _ = device.run_points(f, s, p, ls, variant=variant)
    _compare(res, ora, "depth-setting")
    # output depth from the input array (incl. below the grid and exactly 0)
    f3 = {k: v.copy() for k, v in f.items()}
    f3["depth"][:] = 0.0; f3["depth"][::2] = 0.12; f3["depth"][1::4] = 7.0
    s = abi.default_settings(L)
    ora, _, _ = oh.run_oracle(kind, f3, s, p, ls)
    res, _ = device.run_points(f3, s, p, ls, variant=variant)
    _compare(res, ora, "depth-array")
    # force_tsurf
    s = abi.default_settings(L); s.force_tsurf = 1
    ora, _, _ = oh.run_oracle(kind, f, s, p, ls)
    res, _ = device.run_points(f, s, p, ls, variant=variant)
    _compare(res, ora, "force_tsurf")
    # failures: sticky flag, -9999.0 afterwards, failing index itself still written
    f4 = {k: v.copy() for k, v in f.items()}
    f4["tair"][3, 1000] = 150.0; f4["rhz"][10, 5760] = 500.0; f4["prec"][11, 0] = -5.0
    f4["tdew"][12, 17] = -95.0
    s = abi.default_settings(L)
    ora, _, _ = oh.run_oracle(kind, f4, s, p, ls)
    res, nfail = device.run_points(f4, s, p, ls, variant=variant)
    assert nfail == 3
    _compare(res, ora, "fail")
    # tair[3, 1000] is time index i = 1001: that step still runs and is saved (the loop
    # condition is only re-evaluated afterwards, Simulation.f90:58-95), later ones are not
    assert res["tsurf"][3, 1000] != -9999.0 and res["tsurf"][3, 1001] == -9999.0


def test_other_layer_counts_and_timestep():
    from roadsurf_amd import device
    kind = _oracle_kind()
    for nl in (8, 20, 32):
        n, L = 128, 2881
        f = oh.synth_forcing(n, L, seed=5)
        s = abi.default_settings(L); s.NLayers = nl
        p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
        ora, _, _ = oh.run_oracle(kind, f, s, p, l)
        res, _ = device.run_points(f, s, p, l)
        _compare(res, ora, f"NL{nl}")
    L = 2881
    f = oh.synth_forcing(128, L, seed=5, steps_per_knot=60)
    s = abi.default_settings(L, 60.0); p = abi.default_parameters(60.0)
    l = abi.default_local(); l.InitLenI = 1
    ora, _, _ = oh.run_oracle(kind, f, s, p, l)
    res, _ = device.run_points(f, s, p, l)
    _compare(res, ora, "dt60")


def test_bit_identity_at_scale_8192_points():
    """The melt-out branch (src/Storage.f90:149-162) turns any last-bit difference into
    1e-3..0.2 K; with OCML's exp/log 9 of these 8192 points left the 1e-6 K gate.  With
    glibc-exact exp/log every one of them is bit-identical to the reference."""
    from roadsurf_amd import device
    n, L = 8192, 5761
    f = oh.synth_forcing(n, L, seed=20240110, point_offset=100000)
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    ora, _, _ = oh.run_oracle(_oracle_kind(), f, s, p, l)
    res, _ = device.run_points(f, s, p, l)
    d = np.maximum.reduce([np.abs(res[k] - ora[k]) for k in oh.F64_OUT])
    nonident = int((d.max(1) != 0).sum())
    print("points not bit-identical:", nonident, "max |dTsurf|:", np.abs(res["tsurf"] - ora["tsurf"]).max())
    assert np.abs(res["tsurf"] - ora["tsurf"]).max() < TOL_K
    if HOST_HAS_FMA:
        assert nonident == 0


def test_output_decimation_matches_driver_semantics():
    """decimate = k keeps indices 1, 1+k, 1+2k, ... (what the reference driver writes with
    outputStep, examples/example1/src/roadrunner.cpp:290,303) and equals the full series there."""
    import torch
    from roadsurf_amd import device
    n, L, k = 300, 1441, 120
    f = oh.synth_forcing(n, L, seed=21)
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    full, _ = device.run_points(f, s, p, l)
    plan = device.Plan(n, s, p, 0)
    dev = plan.device
    npad = plan.np_pad

    def pad_t(a, dtype):
        t = torch.zeros((L, npad), dtype=dtype, device=dev)
        t[:, :n] = torch.from_numpy(np.ascontiguousarray(a)).to(dev).T
        return t
    tens = {q: pad_t(f[q], torch.float64) for q in ("tair", "vz", "rhz", "prec", "sw", "lw", "tsurfobs")}
    tens["tdew"] = None; tens["depth"] = None
    tens["precphase"] = pad_t(f["precphase"], torch.int32)
    tens["hour"] = torch.from_numpy(f["hour"]).to(dev)
    win = device.ForcingWindow(L, npad, tens)
    nrows = (L - 1) // k + 1
    out = device.OutputWindow.empty(nrows, npad, dev, decimate=k)
    pp = plan.point_params(plan.uniform_tbottom(2024, 1, 10))
    plan.init_state(win, pp)
    # two launches, boundaries not aligned with the output stride
    plan.step(win, out, pp, 1, 500, window_row=0, out_row0=0)
    plan.step(win, out, pp, 501, L - 500, window_row=500, out_row0=0)
    plan.sync()
    for q in oh.F64_OUT:
        got = out.tensors[q][:, :n].T.cpu().numpy()
        assert np.array_equal(got, full[q][:, ::k]), q
    plan.close()


def _stepwise(f, s, p, n, variant, chunks, t_stride):
    """Step host arrays through the device API with windows whose row stride is `t_stride` columns
    (>= n: include/roadsurf.h only asks for that), in launches of the given lengths; returns the
    outputs [n, L] and the carried state block."""
    import torch
    from roadsurf_amd import device
    L = s.SimLen
    plan = device.Plan(n, s, p, 0)
    plan.set_variant(variant)
    dev = plan.device

    def rows(a, dtype):
        t = torch.zeros((L, t_stride), dtype=dtype, device=dev)
        t[:, :n] = torch.from_numpy(np.ascontiguousarray(a)).to(dev).T
        return t
    tens = {k: rows(f[k], torch.float64) for k in ("tair", "vz", "rhz", "prec", "sw", "lw", "tsurfobs")}
    tens["tdew"] = tens["depth"] = None
    tens["precphase"] = rows(f["precphase"], torch.int32)
    tens["hour"] = torch.from_numpy(np.ascontiguousarray(f["hour"])).to(dev)
    win = device.ForcingWindow(L, t_stride, tens)
    out = device.OutputWindow.empty(L, t_stride, dev)
    pp = plan.point_params(plan.uniform_tbottom(int(f["year"][0]), int(f["month"][0]), int(f["day"][0])))
    plan.init_state(win, pp)
    t0 = 1
    for ns in chunks:
        plan.step(win, out, pp, t0, ns, window_row=t0 - 1, out_row0=0)
        t0 += ns
    assert t0 == L + 1
    plan.sync()
    res = {k: out.tensors[k][:, :n].T.contiguous().cpu().numpy() for k in device.OUT_FIELDS}
    st = plan.state().numpy().copy()
    plan.close()
    return res, st


@pytest.mark.parametrize("variant", [1, 2, 3], ids=["reg", "lds", "duo"])
def test_unpadded_window_stride_and_one_index_launch(variant):
    """Windows that span exactly npoints columns (npoints not a multiple of the wavefront width) and a
    series that ends in a launch of ONE index, SimLen = k * 120 + 1 as the reference driver makes them
    (examples/example1/src/InputSettings.cpp:98): lanes beyond npoints must not touch the windows."""
    n, L = 203, 241
    f = oh.synth_forcing(n, L, seed=404)
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    ora, _, _ = oh.run_oracle(_oracle_kind(), f, s, p, l)
    res, _ = _stepwise(f, s, p, n, variant, [120, 120, 1], t_stride=n)
    _compare(res, ora, f"unpadded-{variant}")


def test_state_of_failed_points_is_the_same_in_every_flavour():
    """A point that fails CheckValues takes the step of that index and none after it
    (examples/example1/src/Simulation.f90:58): the carried profile rs_hip_state_download returns for it
    is the one that index left, whichever kernel flavour stepped it (the two-wavefront flavour's ground
    wave used to go on integrating Tmp(3..N) behind the failure)."""
    n, L = 200, 361
    f = oh.synth_forcing(n, L, seed=58)
    for pt, idx in {3: 0, 70: 100, 131: 119, 150: 120, 199: 300}.items():
        f["tair"][pt, idx] = 250.0
    s = abi.default_settings(L); p = abi.default_parameters()
    states = {}
    for variant in (1, 2, 3):
        _, st = _stepwise(f, s, p, n, variant, [120, 120, 121], t_stride=256)
        states[variant] = st[:, :n]
    NL = s.NLayers
    failed = states[1][abi.RS_MAX_LAYERS + 12] != 0  # RS_ST_FAILED
    assert failed.sum() == 5
    for v in (2, 3):
        for row in range(NL):
            assert np.array_equal(states[1][row], states[v][row]), (v, row)
        for row in range(abi.RS_MAX_LAYERS, abi.RS_MAX_LAYERS + 13):
            assert np.array_equal(states[1][row], states[v][row]), (v, row)


@pytest.mark.parametrize("full", [False, True], ids=["lean", "full"])
def test_every_instance_of_the_one_point_per_lane_kernels_has_the_same_bits(full, monkeypatch):
    """step_kernel_reg / _hybrid / _lds are compiled with and without the history score of rs_hip_recluster and with
    32- or 64-bit window offsets (64 where a stream of a window spans 4 GiB or more: rs_a32_limit).  The 64-bit
    instances never ran in a test before round 6 (a window that large is tens of gigabytes); with the limit lowered
    (ROADSURF_HIP_A32_LIMIT) a window of megabytes takes them: every combination against the reference, bit for
    bit (profiles/r06_kernel_reachability.txt)."""
    from roadsurf_amd import device
    n, L = 300, 721
    f = oh.synth_forcing(n, L, seed=31)
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    if full:
        s.use_relaxation = 1
        l.InitLenI = 240
        l.tair_relax, l.VZ_relax, l.RH_relax = -2.0, 3.0, 85.0
        f["tsurfobs"][:, :240] = f["tair"][:, :240] - 0.5
    ora, _, _ = oh.run_oracle("ref" if oh.have_ref() else "port", f, s, p, l)
    for variant in (1, 2, 4) if full else (1, 2):
        for score in (True, False):
            for limit in (None, "100000"):  # 512 padded columns x 721 rows = 369 152 elements per stream
                if limit:
                    monkeypatch.setenv("ROADSURF_HIP_A32_LIMIT", limit)
                else:
                    monkeypatch.delenv("ROADSURF_HIP_A32_LIMIT", raising=False)
                res, nfail = device.run_points(f, s, p, l, variant=variant, history_score=score, chunk=0)
                for k in oh.F64_OUT:
                    assert np.array_equal(res[k], ora[k]), (variant, score, limit, k)
