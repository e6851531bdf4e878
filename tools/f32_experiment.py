"""fp32 flavour vs fp64 oracle: error distribution + throughput (BASELINE config 5 study)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import oracle_helpers as oh
from roadsurf_amd import abi, device

def run_f32(n, L, seed, offset=0, spk=120, chunk=480, cluster=False, fused=False, edit_knots=None, variant=0,
            return_plan_info=False, history_score=None, nlayers=15):
    """fp32 run of the synthetic workload; with `cluster` the plan is re-sorted after every launch
    (rs_hip_recluster) and the outputs are mapped back through the order of each launch.  fused: the step kernel
    reads the knots itself (rs_hip_step_knots) instead of a forcing window; edit_knots(knots): change the resident
    knot block before the run (tests put values outside CheckValues' limits there)."""
    s = abi.default_settings(L); p = abi.default_parameters()
    s.NLayers = nlayers
    plan = device.Plan(n, s, p, 0); plan.set_precision(32)
    if variant:
        plan.set_variant(variant)
    if history_score is not None:
        plan.set_history_score(history_score)
    dev, npad = plan.device, plan.np_pad
    spec, knots = plan.synth_knots(seed, (L - 1) // spk + 2, point_offset=offset, steps_per_knot=spk)
    if edit_knots is not None:
        plan.sync()
        edit_knots(knots)
    win = device.ForcingWindow.empty(chunk, npad, dev, optional=(), dtype=torch.float32)
    win0 = device.ForcingWindow.empty(1, npad, dev, optional=("tsurfobs",), dtype=torch.float32)
    out = device.OutputWindow.empty(chunk if cluster else L, npad, dev, dtype=torch.float32)
    pp = plan.point_params(plan.uniform_tbottom(2024, 1, 10))
    res = {k: np.full((n, L), np.nan) for k in device.OUT_FIELDS}
    if cluster and not fused and edit_knots is None:
        kbuf = torch.empty((chunk // spk + 3, 9, npad), dtype=torch.float64, device=dev)
        plan.synth_knots_range(spec, kbuf, 0, 2, ordered=True)
        plan.expand_range(spec, kbuf, 0, 2, win0, 1, 1)
    else:
        kbuf = None
        plan.expand(spec, knots, win0, 1, 1)
    plan.init_state(win0, pp)
    t0 = 1
    while t0 <= L:
        ns = min(chunk, L - t0 + 1)
        if cluster:
            k0 = (t0 - 1) // spk
            nk = (t0 + ns - 2) // spk + 1 - k0 + 1
            order = plan.order().clone()
            if fused:
                plan.step_knots(spec, knots, out, pp, t0, ns, out_row0=t0 - 1)
            elif kbuf is None:  # the resident (edited) knots through the plan's order row
                plan.expand_ordered(spec, knots, win, t0, ns)
                plan.step(win, out, pp, t0, ns, out_row0=t0 - 1)
            else:
                plan.synth_knots_range(spec, kbuf, k0, nk, ordered=True)
                plan.expand_range(spec, kbuf, k0, nk, win, t0, ns)
                plan.step(win, out, pp, t0, ns, out_row0=t0 - 1)
            plan.sync()
            idx = order[:n].cpu().numpy()
            for k in device.OUT_FIELDS:
                res[k][idx, t0 - 1:t0 - 1 + ns] = out.tensors[k][:ns, :n].cpu().numpy().T
            plan.recluster()
        elif fused:
            plan.step_knots(spec, knots, out, pp, t0, ns, out_row0=0)
        else:
            plan.expand(spec, knots, win, t0, ns); plan.step(win, out, pp, t0, ns, out_row0=0)
        t0 += ns
    plan.sync()
    if not cluster:
        res = {k: out.tensors[k][:, :n].T.contiguous().cpu().numpy().astype(np.float64) for k in device.OUT_FIELDS}
    info = {"failed": plan.failed_count(), "first_failed": plan.first_failed_index()} if return_plan_info else None
    plan.close()
    return (res, info) if return_plan_info else res

if __name__ == "__main__":
    n, L, seed = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 5761, 20240110
    f = oh.synth_forcing(n, L, seed=seed)
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    ora, _, _ = oh.run_oracle("port", f, s, p, l)
    res = run_f32(n, L, seed)
    for k in oh.F64_OUT:
        d = np.abs(res[k] - ora[k])
        per_point = d.max(1)
        print(f"{k:8s} max {d.max():.3e}  rms {np.sqrt((d**2).mean()):.3e}  p50 {np.percentile(d,50):.2e} "
              f"p99 {np.percentile(d,99):.2e} p99.9 {np.percentile(d,99.9):.2e}  points with max>0.05: {(per_point>0.05).mean()*100:.2f}%  >0.5: {(per_point>0.5).mean()*100:.2f}%")
    dt = np.abs(res['tsurf'] - ora['tsurf'])
    print('fraction of (point,step) with |dTsurf| > 0.05 K:', (dt > 0.05).mean())
