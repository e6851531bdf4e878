"""What the headline's slot-order outputs cost a consumer that wants POINT order at every index (VERDICT r04,
weak point 10).  bench.py's default flavour keeps each launch's output rows in the plan's slot order together
with the order row of the launch; here every launch's six streams x 60 indices are also written into point-major
arrays [point][row] in point order (rs_hip_outputs_by_point, on the plan's own stream, between the launch and its
re-sort) and the pass is timed both ways.  A scatter from inside the step kernel is not an option: 64 lanes
storing 8 bytes each into 64 different lines would ask the memory for eight times the bytes the outputs have.
usage: python tools/bench_point_order_outputs.py [points] [passes]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
from roadsurf_amd import abi, device, sharding, workload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
hours, K, chunk = 48, 3, 60
simlen = hours * 120 + 1
dev = torch.device("cuda", 0)
s = abi.default_settings(simlen); p = abi.default_parameters()
plans, runs = [], []
for j in range(K):
    off, nj = sharding.strong_shard(n, K, j)
    pl = device.Plan(nj, s, p, 0, stream=torch.cuda.Stream(dev))
    pl.set_variant(3)
    plans.append(pl)
    runs.append(workload.SyntheticRun(pl, 20240110, hours, chunk, point_offset=off, plan_order=True, forecast=True,
                                      forecast_mode=workload.DEFAULT_FORECAST_MODE))
dst = [{k: torch.empty((r.plan.np_pad, chunk), dtype=torch.float64, device=dev) for k in device.OUT_FIELDS} for r in runs]
# second mode: the transposition on a stream of its own, two output windows in turn - it runs beside the next launch
side = [torch.cuda.Stream(dev) for _ in runs]
wins = [[r.out, device.OutputWindow.empty(chunk, r.plan.np_pad, dev)] for r in runs]
done = [[None, None] for _ in runs]


def gather(j, overlapped):
    r = runs[j]
    def on_launch(c, t0, ns):
        if not overlapped:  # between the launch and its re-sort: the plan's current order is the launch's
            r.plan.outputs_by_point(r.out, ns, dst[j])
            return
        w = c & 1
        ev = torch.cuda.Event()
        ev.record(r.plan.stream)
        side[j].wait_event(ev)                      # behind the launch that filled window w
        r.plan.outputs_by_point(wins[j][w], ns, dst[j], order=r.orders[c], stream=side[j])
        done[j][w] = torch.cuda.Event()
        done[j][w].record(side[j])
        r.out = wins[j][w ^ 1]                      # the next launch writes the other window ...
        if done[j][w ^ 1] is not None:
            r.plan.stream.wait_event(done[j][w ^ 1])  # ... once its last reader is through
    return on_launch


def one_pass(mode):
    for j, r in enumerate(runs):
        r.out = wins[j][0]
        done[j][0] = done[j][1] = None
    its = [r.iter_pass(gather(j, mode == 2) if mode else None) for j, r in enumerate(runs)]
    while its:
        its = [it for it in its if next(it, None) is not None]


res = {}
for tag, po in (("slot order + order rows (bench.py)", 0), ("per-point series behind every launch (rs_hip_outputs_by_point)", 1),
                ("per-point series, on a second stream with two output windows in turn", 2)) * 2:
    one_pass(po)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(passes):
        one_pass(po)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / passes
    print(f"{tag}: {dt * 1e3:.1f} ms per pass -> {n * simlen / dt:.3e} point-timesteps/s", flush=True)
# the point-major rows are the slot rows, permuted and transposed (last launch of plan 0, last mode)
r = runs[0]
c = len(r.starts) - 1
order = r.orders[c][:r.plan.npoints].long()
ns = min(chunk, simlen - r.starts[c] + 1)
src = wins[0][c & 1]
ok = all(torch.equal(dst[0][k][order, :ns].T.contiguous(), src.tensors[k][:ns, :r.plan.npoints].contiguous()) for k in device.OUT_FIELDS)
print("last launch of plan 0: dst[order[slot], row] == src[row, slot]:", ok)
