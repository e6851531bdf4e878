#!/usr/bin/env python3
"""bench.py — throughput of the RoadSurf hot path on MI355X.

Metric (BASELINE.json): point-timesteps/s at 1 M points x 48 h, fp64.

One bench "step" = one full pass of the hot path over the workload:
    init kernel  ->  for every time chunk:  hourly knots -> DTSecs grid (the device
    twin of the reference driver's interpolation)  ->  step kernel (the model:
    5 761 time indices per point in all).
Inputs resident in HBM before the timed region: the hourly forcing knots of every
point (what an NWP source delivers).  Step-resolution forcing for 1 M points x
5 761 indices is 300 GB (> 288 GB HBM), so it is produced per chunk
on the device: by an expansion kernel into a forcing window the step kernel reads
from HBM (--variant 1, 2; FULL; fp32; natural order), or - the default LEAN fp64
flavour since round 4 - inside the step kernel itself, whose ground wavefront
interpolates the next index's forcing from the knots while the surface wavefront
works on the current one (rs_hip_step_knots; no forcing window exists).  The six
outputs are written every time index (reference SaveOutput semantics) into a
chunk buffer.

The pass itself lives in roadsurf_amd/workload.py (SyntheticRun) and is the code the
parity tests step at the same size (tests/test_hip_golden_and_scale.py).  Outputs stay
attributable to points: with plan order on, the order row of every launch is kept
(inside the timed region).  A second timed leg runs the same passes in natural order and
is reported as `natural_order_value`.

Multi-GPU: one process per GPU (torch.distributed.run), points sharded with no
data-path collective; the only collectives are the timing barrier and the MAX
over ranks.  Default is STRONG scaling (BASELINE config 4: --total-points 1 000 000
sharded over the GPUs); `--points N` gives every GPU N points (weak scaling).

The default run (N = 1) adds short extra legs under the same clock, reported as extra keys of the
one JSON line: `full_feature_value` (the FULL feature set on the same points) and
`driver_path_{relax,coupling,sky}_value` (rs_driver_run from raw series in host arrays, PCIe
inclusive), `driver_path_relax_holes_value` (the same on observation series with missing sensors,
gaps and early ends), `driver_path_relax_bench_weather_value` (the same on the headline's weather), `host_batch_value` (runsimulation_batch from step-resolution host arrays);
`--no-extra-legs` skips them.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
ALGO_BYTES_PER_UNIT = 100.0    # 52 B read + 48 B written per point-timestep
# the knot-reading flavour's own compulsory HBM bytes per point-timestep: 48 B of outputs + 9 knot doubles per
# 120 indices (0.6 B) + the state block once per launch (134 slots x 8 B over >= 60 indices, read and written: <= 36 B
# at 60 indices per launch, ~18 B at 120) - reported beside the contract's 100 B figure, never instead of it
FUSED_BYTES_PER_UNIT = 48.0 + 9 * 8 / 120.0 + 2 * 49 * 8 / 60.0
TRAFFIC_FILE = "profiles/r06_traffic.json"  # committed PMC summary the roofline's `traffic` is read from
TRAFFIC_FILE_F32 = "profiles/r06_f32_traffic.json"


def effective_cpus() -> int:
    """CPUs this process may actually use: affinity mask AND cgroup v2 quota (the GPU box gives a
    16-CPU share of a 256-thread host; OpenMP's default of 256 threads would oversubscribe it)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(sample_points: int, simlen: int, seed: int):
    """Reference Fortran (oracle/_ref, kind 'reference') or, if absent, the C port,
    OpenMP over points on this box's host cores, on a bounded sample of the workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_helpers as oh  # the checker: timed here as the CPU baseline, nothing else of it is used
    from roadsurf_amd import abi, synth

    kind = "ref" if os.path.exists(oh.REF_SO) else "port"
    s = abi.default_settings(simlen)
    p = abi.default_parameters()
    l = abi.default_local()
    l.InitLenI = 1
    f = synth.synth_forcing(sample_points, simlen, seed=seed)
    oh.run_oracle(kind, {k: (v[:64] if v.ndim == 2 else v) for k, v in f.items()}, s, p, l,
                  nthreads=effective_cpus())  # warm
    t = time.perf_counter()
    _, _, threads = oh.run_oracle(kind, f, s, p, l, nthreads=effective_cpus(), copy_inputs=False)
    dt = time.perf_counter() - t
    return {
        "value": sample_points * simlen / dt,
        "unit": "point-timesteps/s",
        "cores": int(threads),
        "kind": "reference" if kind == "ref" else "port",
        "sample": f"{sample_points} points x {simlen} time indices of the same synthetic workload "
                  f"(seed {seed}), OpenMP over points, {dt:.2f} s wall",
    }


def measured_traffic(points: int, chunk: int, plans: int, f32: bool):
    """HBM bytes per launch of the dominant kernel and the counter-derived figures from the committed
    rocprofv3 PMC summary - only if it was collected on THIS configuration and from THESE kernel
    sources (the summary is stamped with roadsurf_amd.provenance.csrc_sha16() when it is made; a
    summary that predates the last kernel change is not reported)."""
    from roadsurf_amd import provenance

    name = TRAFFIC_FILE_F32 if f32 else TRAFFIC_FILE
    try:
        t = json.load(open(os.path.join(ROOT, name)))
    except (OSError, ValueError):
        return None, None, name, "no committed counter summary"
    if t.get("points") != points or t.get("chunk_steps") != chunk or t.get("plans_per_gpu") != plans:
        return None, None, name, "counter summary was collected on another configuration"
    if t.get("csrc_sha16") != provenance.csrc_sha16():
        return None, None, name, "counter summary predates the kernel sources of this run"
    return t.get("traffic_bytes_per_launch"), t.get("derived"), name, None


def host_batch_leg(n: int, simlen: int, seed: int) -> dict:
    """runsimulation_batch (the Fortran entry over the C-ABI shim) on `n` synthetic points held as the
    reference holds them: one array per point and variable at step resolution, outputs at every index."""
    import ctypes as C
    import numpy as np

    from roadsurf_amd import abi, lib as rslib, synth

    L = rslib.load()
    f = synth.synth_forcing(n, simlen, seed=seed)  # the product's own host generator (rs_synth_fill_points)
    out = {k: np.empty((n, simlen)) for k in synth.F64_OUT}
    s = abi.default_settings(simlen)
    p = abi.default_parameters()
    l = abi.default_local()
    l.InitLenI = 1
    ips = (abi.InputPointers * n)()
    ops = (abi.OutputPointers * n)()
    keep = []
    for pt in range(n):
        ips[pt], ops[pt], kp = synth.point_pointers(f, pt, out)
        keep.append(kp)
    larr = (abi.LocalParameters * n)(*([l] * n))
    st = C.c_int32(0)
    times = []
    for rep in range(4):
        t = time.perf_counter()
        L.runsimulation_batch(n, ops, ips, C.byref(s), C.byref(p), larr, C.byref(st))
        dt = time.perf_counter() - t
        if st.value != 0:
            raise RuntimeError(f"runsimulation_batch: {rslib.last_error()}")
        if rep:
            times.append(dt)
    mean = sum(times) / len(times)
    bytes_per_unit = 11 * 8 + 2 * 4 + 6 * 8
    return {"value": n * simlen / mean, "unit": "point-timesteps/s", "seconds_per_call": mean,
            "seconds_per_call_all": times, "calls_timed": len(times), "calls_warm": 1,
            "boundary_GBps": n * simlen * bytes_per_unit / mean / 1e9,
            "config": {"workload": f"runsimulation_batch: {n} synthetic points x SimLen {simlen} from step-resolution "
                                   "host arrays (one per point and variable, as the reference driver holds them), "
                                   "LEAN feature set, outputs at every index back in host arrays",
                       "value_is": "mean of the timed calls", "pcie_inclusive": True,
                       "bytes_over_the_boundary_per_unit": bytes_per_unit}}


F32_POINTS, F32_HOURS, F32_PLANS, F32_CHUNK = 1_250_000, 168, 2, 360


def f32_config5_leg(make_plans, timed_leg, params, args, full: bool = False) -> dict:
    """BASELINE config 5's per-GPU share through the fp32 flavour (rs_kernels_f32.hip step_kernel_f32duo: two
    points per lane, two wavefronts per 128 points, forcing interpolated from the resident knots).  `full`: the
    FULL feature set on the same points (dew-point and observation streams, 6 h initialization phase, relaxation)."""
    from roadsurf_amd import abi, workload

    simlen = F32_HOURS * workload.SPK + 1
    s32 = abi.default_settings(simlen)
    if full:
        s32.use_relaxation = 1
    plans, offs = make_plans(F32_PLANS, s32, 0, npoints=F32_POINTS, f32=True)
    steps = 3
    el, step_ms, nl, chunk, busy = timed_leg(True, plans, offs, F32_CHUNK, full, steps=steps, warmup=1,
                                             hours=F32_HOURS, f32=True)
    nfail = sum(pl.failed_count() for pl in plans)
    for pl in plans:
        pl.close()
    units = F32_POINTS * simlen
    achieved = 52.0 * units * steps / (busy / 1e3) / 1e9
    traffic, valu, traffic_file, traffic_note = ((None, None, None, "not collected for the FULL instance") if full else
                                                 measured_traffic(F32_POINTS, chunk, F32_PLANS, True))
    return {
        "value": units * steps / el, "unit": "point-timesteps/s", "ms_per_step": el / steps * 1e3, "steps": steps,
        "warmup": 1, "dtype": "f32", "failed_points": int(nfail),
        "config": {"workload": f"{F32_POINTS} synthetic points x {F32_HOURS} h (SimLen {simlen}, DTSecs 30, NLayers 15), "
                               f"fp32 state / forcing / outputs / arithmetic, {'FULL (dew point and observation streams, 6 h initialization phase, relaxation)' if full else 'LEAN'} feature set, outputs every time "
                               "index, attributable to points; hourly knots of every point resident in HBM",
                   "plans_per_gpu": F32_PLANS, "chunk_steps": chunk, "plan_order": True,
                   "gate": "distribution of |fp32 - fp64 oracle| (tests/test_hip_f32.py): 99.999 % of the point-steps "
                           "within 0.05 K over 7 days, rms < 5e-4 K"},
        "roofline": {"bound": "hbm", "kernel": "rs32::step_kernel_f32duo" + ("<FULL>" if full else ""), "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": (str(traffic_file) + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, mean per launch; "
                                        "stamped with the hash of the kernel sources)") if traffic is not None else None,
                     "traffic_note": traffic_note, "valu": valu,
                     "algorithmic_bytes_per_unit": 52.0, "avg_launch_ms": step_ms / max(nl, 1), "launches": nl,
                     "busy_ms": busy, "concurrent_launches": step_ms / busy,
                     "step_kernel_only_value": units * steps / (busy / 1e3),
                     "note": "the kernel neither reads nor writes a forcing window (the ground wavefront interpolates "
                             "the knots): what HBM moves is the 24 B of outputs per point-timestep; the binding "
                             "roofline is vector issue (DESIGN.md 3.9)"},
    }


class ClockProbe:
    """Engine clock of the GPU while the timed passes run: rs_hip_clock_probe kernels (one wavefront
    that reads the shader-clock counter and the constant 100 MHz counter about 0.2 ms apart) enqueued
    on a side stream after every pass, i.e. beside the step kernels of the next one.  MHz = 100 x
    shader ticks / 100 MHz ticks, averaged over the probes; None if the two counters run in step (a
    chip whose shader counter is the constant clock)."""

    def __init__(self, torch, dev, dev_index: int, nprobes: int):
        from roadsurf_amd import lib as rslib

        self.torch, self.dev_index = torch, dev_index
        self.L = rslib.load()
        self.stream = torch.cuda.Stream(dev)
        self.buf = torch.zeros((max(nprobes, 1), 2), dtype=torch.int64, device=dev)
        self.n = 0

    def probe(self) -> None:
        if self.n < self.buf.shape[0]:
            import ctypes as C

            self.L.rs_hip_clock_probe(self.dev_index, C.c_void_p(self.buf[self.n].data_ptr()), 200,
                                      C.c_void_p(self.stream.cuda_stream))
            self.n += 1

    def mean_mhz(self):
        self.stream.synchronize()
        b = self.buf[:self.n].cpu().numpy()
        ok = [(c, r) for c, r in b if r > 0 and c > 0]
        if not ok:
            return None
        mhz = [100.0 * c / r for c, r in ok]
        m = sum(mhz) / len(mhz)
        return None if abs(m - 100.0) < 1.0 else m


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scaling", choices=("strong", "weak"), default=None,
                    help="strong (default): --total-points are sharded over the GPUs (BASELINE config 4: "
                         "1 M points over 1/2/4/8); weak: every GPU gets --points")
    ap.add_argument("--total-points", type=int, default=1_000_000)
    ap.add_argument("--points", type=int, default=None, help="points per GPU (implies --scaling weak)")
    ap.add_argument("--hours", type=int, default=48)
    ap.add_argument("--chunk", type=int, default=0,
                    help="time indices per step-kernel launch; 0 = auto by flavour and shard size (see below "
                         "--plans-per-gpu in the source)")
    ap.add_argument("--control", choices=["gloo", "nccl"], default="gloo",
                    help="backend of the barriers / MAX / checksum SUM between ranks (no data-path collective)")
    ap.add_argument("--variant", type=int, default=0,
                    help="0 auto (3 for the LEAN fp64 workload), 1 register profile (one point per lane), 2 LDS "
                         "profile, 3 two wavefronts per 64 points - in plan order without a forcing window: the "
                         "ground wave makes the forcing from the resident knots (rs_hip_step_knots)")
    ap.add_argument("--seed", type=int, default=20240110)
    ap.add_argument("--f32", action="store_true",
                    help="BASELINE config 5 flavour: fp32 state/forcing/outputs/arithmetic "
                         "(tolerance-gated, not the parity path); default is fp64")
    ap.add_argument("--cluster", type=int, default=1,
                    help="1: plan order - re-sort the plan's slots by boundary-layer passes after every "
                         "launch (rs_hip_recluster; windows are generated in slot order, the order row "
                         "of every launch is kept), 0: natural order only")
    ap.add_argument("--sort-key", choices=("forecast", "history"), default="forecast",
                    help="plan order: sort by a forecast of the next window's boundary-layer regime and "
                         "passes (rs_hip_recluster_forecast) or by the passes of the last launch")
    ap.add_argument("--forecast-alpha", type=float, default=0.5)
    ap.add_argument("--forecast-mode", type=int, default=378059,
                    help="fields of the forecast key, most significant first (roadsurf_amd/workload.py)")
    ap.add_argument("--plans-per-gpu", type=int, default=0,
                    help="cut this GPU's points into K plans on K streams whose launches interleave: "
                         "one plan's HBM-bound window expansion and re-sort run beside another's "
                         "VALU-bound step kernel, and the tail of one launch under the head of the next. "
                         "0 = auto by flavour and shard size")
    ap.add_argument("--no-natural-leg", action="store_true",
                    help="skip the second timed leg (natural order) that gives natural_order_value")
    ap.add_argument("--full", action="store_true",
                    help="the FULL feature set instead of the BASELINE workload's LEAN one: Tdew / TsurfObs / "
                         "depth streams present, a 6-hour initialization phase and relaxation behind it (what "
                         "an operational run uses; reported as config.feature_set)")
    ap.add_argument("--checksum", action="store_true",
                    help="after the timed legs run ONE more untimed pass and report config.checksum: the "
                         "order-independent wrap-around sum of the bit patterns of all six outputs of every "
                         "point and index, summed over the ranks (what tests/test_hip_config4.py compares "
                         "between a one-rank and a two-rank launch)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the short extra legs of the default run (N = 1, fp64, LEAN headline only): the FULL "
                         "feature set on the same synthetic points (full_feature_value) and the driver data path "
                         "rs_driver_run with relaxation / coupling / sky view (driver_path_*_value), and runsimulation_batch "
                         "from step-resolution host arrays (host_batch_value)")
    ap.add_argument("--extra-points", type=int, default=1_000_000,
                    help="points of the driver-path legs (host arrays in, hourly outputs back: PCIe inclusive)")
    ap.add_argument("--host-batch-points", type=int, default=32768,
                    help="points of the runsimulation_batch leg (step-resolution host arrays in and out)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=16384)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from roadsurf_amd import abi, device, sharding, workload

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    device.require_gpu()
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        # The data path has no collective (points are independent: SURVEY.md 8e); what the ranks
        # exchange is the control plane of this script - the barriers around the timed region, the MAX
        # of the elapsed times, the SUM of the checksums.  gloo by default: it is what the two-rank test
        # exercises (tests/test_hip_config4.py) and it cannot fail on the GPU side; --control nccl
        # runs the same three calls over RCCL.
        if args.control == "nccl" and ndev >= world:
            dist.init_process_group("nccl", device_id=dev)
        else:
            if ndev < world:  # rehearsal: more ranks than GPUs on this box -> ranks share cards
                print(f"[bench] {world} ranks on {ndev} GPU(s): sharing devices", file=sys.stderr)
            dist.init_process_group("gloo")

    scaling = args.scaling or ("weak" if args.points is not None else "strong")
    if scaling == "weak":
        per_gpu = args.points if args.points is not None else args.total_points
        offset, n = sharding.weak_shard(per_gpu, rank)
        total_points = per_gpu * world
    else:
        total_points = args.total_points if args.points is None else args.points * world
        offset, n = sharding.strong_shard(total_points, world, rank)
    simlen = args.hours * workload.SPK + 1  # examples/example1/src/InputSettings.cpp:98
    settings = abi.default_settings(simlen)
    if args.full:
        settings.use_relaxation = 1
    params = abi.default_parameters()
    if args.variant == 0 and not args.f32:
        # Round 4: the two-wavefront flavour whose ground wave makes the forcing from the resident knots itself
        # (roadsurf_amd/workload.py: no window expansion, no forcing window) is the faster one at every size -
        # 1 M points: 1.93-1.97e10 with one point per lane (4 plans x 120), 2.11-2.13e10 with this flavour
        # 
        args.variant = 3
    fused = args.variant == 3 and not args.f32 and bool(args.cluster)
    if fused:
        # measured (round 4, tools/experiments/r4_fused_{big,small}_sweep.sh): with the expansion gone a launch
        # cycle is short, so fewer plans hide it and shorter launches (a fresher sort key) pay: 3 x 90 from
        # 400 000 points on the GPU (1 M, r4_ground_sweep.sh: 2.29e10 for 3 x 90, 3 x 120 and 2 x 120 alike; 2.27e10
        # for 4 x 90 / 4 x 120), 4 x 120 from 200 000, 2 x 240 below (last pass with the chain kernels at raised
        # priority, tools/experiments/r4_prio_chain_sweep.sh: 2.35e10 / 2.33e10 / 2.11e10 / 1.37e10 at 1 M / 500 000 /
        # 250 000 / 125 000 points; with the 10-bit key, r4_key10_sweep.sh: 3 x 60 2.39e10 / 2.37e10 at 1 M / 500 000,
        # 3 x 90 2.38e10 / 2.36e10)
        K, ch = (3, 60) if n >= 400_000 else (4, 120) if n >= 200_000 else (2, 240) if n >= 100_000 else (1, 240)
    elif args.f32 and args.variant not in (1, 2):
        # round 6 (tools/experiments/r6_f32_sweep.sh): with two points per lane a step launch costs half as much
        # against the same re-sort chain - two plans, launches of three hours
        K, ch = (2 if n >= 100_000 else 1), 360
    else:
        # measured on MI355X (tools/experiments/exp_plans.sh, r3_small2.sh; DESIGN_HISTORY.md 6)
        K, ch = (4 if n >= 200_000 else 2 if n >= 100_000 else 1), (120 if n >= 400_000 else 240)
    if args.plans_per_gpu > 0:
        K = args.plans_per_gpu
    chunk_auto = args.chunk <= 0
    if chunk_auto:
        args.chunk = ch
    if args.full and n >= 400_000:
        # measured: three plans; launches of 120 indices with the knot-reading flavour (tools/experiments/
        # r4_full_duo.sh: 2.01e10 for 3 x 120, 2.00e10 for 3 x 240, 1.97e10 for 3 x 90 and 2 x 120), of 240
        # with a forcing window (tools/experiments/r3_full3.sh)
        K = args.plans_per_gpu if args.plans_per_gpu > 0 else 3
        if chunk_auto:
            args.chunk = 120 if fused else 240
    def make_plans(K, settings, variant, npoints=None, f32=None):
        plans, offsets = [], []
        for j in range(K):
            off_j, n_j = sharding.strong_shard(n if npoints is None else npoints, K, j)
            st = torch.cuda.current_stream(dev) if K == 1 else torch.cuda.Stream(dev)
            pl = device.Plan(n_j, settings, params, dev_index, stream=st)
            if variant:
                pl.set_variant(variant)
            if args.f32 if f32 is None else f32:
                pl.set_precision(32)
            plans.append(pl)
            offsets.append(offset + off_j)
        return plans, offsets

    plans, offsets = make_plans(K, settings, args.variant)
    plans_main, offsets_main = plans, offsets

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed_leg(plan_order: bool, plans=None, offsets=None, chunk_steps=None, full=None, steps=None,
                  warmup=None, hours=None, f32=None):
        """W untimed + exactly K timed passes, barrier + synchronize on both sides, MAX over ranks."""
        plans = plans_main if plans is None else plans
        offsets = offsets_main if offsets is None else offsets
        chunk_steps = args.chunk if chunk_steps is None else chunk_steps
        full = args.full if full is None else full
        steps = args.steps if steps is None else steps
        warmup = args.warmup if warmup is None else warmup
        runs = [workload.SyntheticRun(pl, args.seed, args.hours if hours is None else hours, chunk_steps, point_offset=o,
                                      plan_order=plan_order, f32=args.f32 if f32 is None else f32,
                                      forecast=args.sort_key == "forecast",
                                      forecast_alpha=args.forecast_alpha, forecast_mode=args.forecast_mode,
                                      full=full)
                for pl, o in zip(plans, offsets)]
        run = runs[0]

        def one_pass():
            its = [r.iter_pass() for r in runs]
            while its:  # launch c of every plan, then launch c+1 ...
                its = [it for it in its if next(it, None) is not None]

        for _ in range(warmup):
            one_pass()
        fence()
        for pl in plans:
            pl.timing_reset()
        ref = torch.cuda.Event(enable_timing=True)
        ref.record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        clk = ClockProbe(torch, dev, dev_index, steps)
        t_start = time.perf_counter()
        for _ in range(steps):
            one_pass()
            clk.probe()  # runs beside the launches still in flight
        fence()
        elapsed = time.perf_counter() - t_start
        sclk.append(clk.mean_mhz())
        elapsed = sharding.max_over_ranks(
            elapsed, dist if world > 1 else None,
            dev if (world > 1 and dist.get_backend() == "nccl") else None)
        # HIP events around every step launch, on the stream it was launched on.  With K > 1 the
        # launches of different plans overlap in time: `busy_ms` is the union of their intervals
        # (the time the device spent running step kernels), `step_ms` the plain sum of durations
        iv = []
        for pl in plans:
            iv += pl.timing_intervals(ref)
        step_ms, nlaunch = sum(b - a for a, b in iv), len(iv)
        busy_ms, end = 0.0, float("-inf")
        for a, b in sorted(iv):
            if b > end:
                busy_ms += b - max(a, end)
                end = b
        chunk = run.chunk
        del run, runs
        torch.cuda.empty_cache()
        return elapsed, step_ms, nlaunch, chunk, busy_ms

    cluster = bool(args.cluster)
    sclk = []  # mean engine clock of each timed leg (MHz), None where not readable
    elapsed, step_ms, nlaunch, chunk, busy_ms = timed_leg(cluster)
    natural = None
    if cluster and not args.no_natural_leg:
        if fused:
            # natural order has no knot-reading flavour (nothing to gather through): its best shape is the
            # round-3 one - one point per lane, expansion kernel + forcing window, 4 x 120 / 4 x 240 / 2 x 240
            Kn, chn = (4 if n >= 200_000 else 2 if n >= 100_000 else 1), (120 if n >= 400_000 else 240)
            nplans, noffs = make_plans(Kn, settings, 1)
            natural = timed_leg(False, nplans, noffs, chn) + (Kn,)
            for pl in nplans:
                pl.close()
            del nplans
            torch.cuda.empty_cache()
        else:
            natural = timed_leg(False) + (K,)
    nfail = sum(pl.failed_count() for pl in plans)
    checksum = None
    if args.checksum:
        itype = torch.int32 if args.f32 else torch.int64
        acc = torch.zeros((), dtype=torch.int64, device=dev)
        for pl, o in zip(plans, offsets):
            run = workload.SyntheticRun(pl, args.seed, args.hours, args.chunk, point_offset=o,
                                        plan_order=cluster, f32=args.f32, forecast=args.sort_key == "forecast",
                                        forecast_alpha=args.forecast_alpha, forecast_mode=args.forecast_mode,
                                        full=args.full)

            def on_launch(c, t0, ns, run=run, pl=pl):
                with torch.cuda.stream(pl.stream):
                    for k in device.OUT_FIELDS:
                        acc.add_(run.out.tensors[k][:ns, :pl.npoints].view(itype).sum(dtype=torch.int64))
            torch.cuda.synchronize(dev)
            run.run_pass(on_launch)
            torch.cuda.synchronize(dev)
            del run
        total = acc.cpu()
        if world > 1:
            if dist.get_backend() == "nccl":
                t = acc.clone()
                dist.all_reduce(t)  # SUM of int64 wraps around like the per-rank sums
                total = t.cpu()
            else:
                dist.all_reduce(total)
        checksum = int(total.item())

    units_per_pass_job = total_points * simlen          # whole job, all ranks
    units_per_pass_rank = n * simlen
    value = units_per_pass_job * args.steps / elapsed

    # ---- extra legs of the default run: the rows behind the headline, under the same clock ----------
    # (N = 1, fp64, LEAN headline only; each a few seconds; --no-extra-legs skips them)
    extra = {}
    if world == 1 and not (args.no_extra_legs or args.f32 or args.full) and cluster:
        t_x = time.perf_counter()
        # the headline's plans are done (its timing events and failure counts are read): every leg below has the
        # GPU to itself, as the headline had (the library counts a device's live plans: rs_api.hip `underfilled`)
        for pl in plans:
            pl.close()
        # (1) the FULL feature set (Tdew / TsurfObs / depth streams, 6 h initialization phase, relaxation
        # behind it) on the same synthetic points: what `bench.py --full` reports as its headline
        s_full = abi.default_settings(simlen)
        s_full.use_relaxation = 1
        Kf, chunk_f = (3, 120) if n >= 400_000 else (K, args.chunk)  # measured: tools/experiments/r4_full_duo.sh
        fplans, foffs = make_plans(Kf, s_full, args.variant)
        f_steps = 2
        f_elapsed, _, f_nl, f_chunk, f_busy = timed_leg(True, fplans, foffs, chunk_f, True, steps=f_steps, warmup=1)
        extra["full_feature"] = {
            "value": units_per_pass_job * f_steps / f_elapsed, "unit": "point-timesteps/s",
            "ms_per_step": f_elapsed / f_steps * 1e3, "steps": f_steps, "warmup": 1,
            "step_kernel_only_value": units_per_pass_rank * f_steps / (f_busy / 1e3),
            "config": {"workload": f"{total_points} synthetic points x {args.hours} h (SimLen {simlen}), fp64, FULL "
                                   "feature set (dew point and observation streams - no depth stream, as in the "
                                   "reference driver -, 6 h initialization phase, relaxation), outputs every time "
                                   "index, inputs resident in HBM",
                       "kernel_variant": args.variant,
                       "plans_per_gpu": Kf, "chunk_steps": f_chunk, "plan_order": True, "launches": f_nl},
        }
        for pl in fplans:
            pl.close()
        del fplans
        torch.cuda.empty_cache()
        # (2) the driver data path: raw hourly forecast + 10-minute observations in HOST arrays ->
        # rs_driver_run -> hourly outputs back in host arrays (PCIe inclusive: a whole-call rate, never
        # comparable with `value`, whose inputs are resident in HBM)
        from roadsurf_amd import driver_workload

        if not os.environ.get("ROADSURF_HIP_DEVICES"):
            # the library's fan-out stays on THIS GPU, four blocks on it (the default of a one-GPU process)
            os.environ["ROADSURF_HIP_DEVICES"] = ",".join([str(dev_index)] * int(os.environ.get("ROADSURF_HIP_PLANS_PER_DEVICE", "4")))
        # distinct series for every point (tiles of 65 536 series put identical lanes side by
        # side after the forecast sort) and the MEAN of three timed calls, like the headline's mean over passes
        dw = driver_workload.DriverWorkload(args.extra_points, args.hours, unique=None)
        for mode in ("relax", "coupling", "skyview"):
            best, times, r = dw.time_calls(mode, reps=3, warm=1, device=-1)
            mean = sum(times) / len(times)
            extra["driver_path_" + ("sky" if mode == "skyview" else mode)] = {
                "value": dw.n * dw.simlen / mean, "unit": "point-timesteps/s", "seconds_per_call": mean,
                "seconds_per_call_all": times, "best_call_value": dw.n * dw.simlen / best,
                "calls_timed": len(times), "calls_warm": 1, "points_ok": int((r["status"] == 0).sum()),
                "config": {"workload": f"rs_driver_run: {dw.n} points x {args.hours} h (SimLen {dw.simlen}) from raw series "
                                       f"in pageable host arrays (hourly forecast + 10-minute observations over the "
                                       f"first {driver_workload.OBS_HOURS} h), interpolation / overlay / Tdew<->RH / "
                                       f"simulation / hourly decimation on the GPU, outputs back in host arrays; "
                                       f"mode {mode}: relaxation"
                                       + (", coupling" if mode == "coupling" else "")
                                       + (", per-point sky view and local horizons" if mode == "skyview" else ""),
                           "series": "distinct for every point (generated once, before the timed calls)",
                           "value_is": "mean of the timed calls",
                           "pcie_inclusive": True, "raw_input_bytes": dw.raw_bytes(mode),
                           "forcing_window": "none: the step kernel's ground wavefront interpolates and overlays the "
                                             "raw series itself (rs_step_raw), coupling's replay rounds included",
                           "blocks_per_device": int(os.environ.get("ROADSURF_HIP_PLANS_PER_DEVICE", "4"))},
            }
            del r
        del dw
        # ... and the same call on observation series as real networks have them (round 5): 10 % of the stations
        # without an air-temperature / humidity / wind sensor each, 10 % of the road-temperature observations
        # missing, 20 % of the series ending one to three hours early - the lanes of a wavefront then disagree on
        # the source that supplies a variable
        dw = driver_workload.DriverWorkload(args.extra_points, args.hours, unique=None, missing=0.1, ragged=0.2)
        best, times, r = dw.time_calls("relax", reps=3, warm=1, device=-1)
        mean = sum(times) / len(times)
        extra["driver_path_relax_holes"] = {
            "value": dw.n * dw.simlen / mean, "unit": "point-timesteps/s", "seconds_per_call": mean,
            "seconds_per_call_all": times, "best_call_value": dw.n * dw.simlen / best,
            "calls_timed": len(times), "calls_warm": 1, "points_ok": int((r["status"] == 0).sum()),
            "config": {"workload": f"rs_driver_run as driver_path_relax, {dw.n} points x {args.hours} h, on observation "
                                   "series with holes: 10 % of the stations without an air-temperature / humidity / "
                                   "wind sensor each, 10 % of the road-temperature observations missing, 20 % of the "
                                   "series ending 1-3 h early",
                       "series": "distinct for every point", "value_is": "mean of the timed calls", "pcie_inclusive": True},
        }
        del r, dw
        # ... and on the weather the device-resident legs above run on, put into raw series (round 6: what of the
        # distance between driver_path_relax and full_feature_value is the workload - here precipitation is one event
        # on 30 % of the points, there it is independent from hour to hour on all of them - and what the code path;
        # profiles/r06_driver_weather.txt)
        dw = driver_workload.DriverWorkload(args.extra_points, args.hours, unique=None, weather="bench", seed=args.seed)
        best, times, r = dw.time_calls("relax", reps=3, warm=1, device=-1)
        mean = sum(times) / len(times)
        extra["driver_path_relax_bench_weather"] = {
            "value": dw.n * dw.simlen / mean, "unit": "point-timesteps/s", "seconds_per_call": mean,
            "seconds_per_call_all": times, "best_call_value": dw.n * dw.simlen / best,
            "calls_timed": len(times), "calls_warm": 1, "points_ok": int((r["status"] == 0).sum()),
            "config": {"workload": f"rs_driver_run as driver_path_relax, {dw.n} points x {args.hours} h, the raw series "
                                   "holding the synthetic weather of the headline and full_feature legs (csrc/rs_synth.h: "
                                   "hourly knots as the forecast source, the observations on the same lines)",
                       "series": "distinct for every point", "value_is": "mean of the timed calls", "pcie_inclusive": True},
        }
        del r, dw
        # (3) the drop-in batch entry from STEP-RESOLUTION host arrays (runsimulation_batch: 11 f64 + 2 i32 in,
        # 6 f64 out per point and index over the boundary): PCIe-bound by construction
        extra["host_batch"] = host_batch_leg(args.host_batch_points, simlen, args.seed)
        # (4) BASELINE config 4's per-GPU shards on this one GPU (1 M points over 8 / 4 GPUs: 125 000 / 250 000
        # points, the headline's flavour and launch shapes for those sizes): what DESIGN.md 3.2's projection of the
        # strong-scaling curve rests on, under the driver's clock
        for pts in (125_000, 250_000):
            Ks, chs = (4, 120) if pts >= 200_000 else (2, 240)
            splans, soffs = make_plans(Ks, settings, args.variant, npoints=pts)
            s_steps = 3
            s_el, s_step_ms, s_nl, s_chunk, s_busy = timed_leg(True, splans, soffs, chs, False, steps=s_steps, warmup=1)
            extra["shard_%dk" % (pts // 1000)] = {
                "value": pts * simlen * s_steps / s_el, "unit": "point-timesteps/s", "ms_per_step": s_el / s_steps * 1e3,
                "steps": s_steps, "warmup": 1, "avg_launch_ms": s_step_ms / max(s_nl, 1),
                "concurrent_launches": s_step_ms / s_busy,
                "config": {"workload": f"{pts} synthetic points x {args.hours} h (SimLen {simlen}), fp64, LEAN: the shard "
                                       f"one GPU holds when config 4's 1 000 000 points are cut over {1_000_000 // pts} GPUs",
                           "plans_per_gpu": Ks, "chunk_steps": s_chunk, "kernel_variant": args.variant, "plan_order": True},
            }
            for pl in splans:
                pl.close()
            del splans
            torch.cuda.empty_cache()
        # (5) BASELINE config 5 at its per-GPU shape: fp32, 1.25 M points (10 M over 8 GPUs) x 7 days (SimLen 20 161),
        # tolerance-gated against the fp64 oracle (tests/test_hip_f32.py), its own roofline block (52 algorithmic
        # bytes per point-timestep: 6 x 4 + 4 read, 6 x 4 written)
        extra["f32_config5"] = f32_config5_leg(make_plans, timed_leg, params, args)
        # ... and with the FULL feature set (a hindcast has observations to start from)
        extra["f32_config5_full"] = f32_config5_leg(make_plans, timed_leg, params, args, full=True)
        extra["seconds"] = time.perf_counter() - t_x
    # dominant kernel: step kernel, HIP events on its own stream around every launch (this rank).
    # achieved = algorithmic bytes of the launches / time the device spent in them.  With one plan
    # that is bytes per launch / average launch duration; with K plans on K streams the launches
    # overlap (that is the point: one plan's window expansion hides under another's step kernel),
    # so the denominator is the union of the launch intervals, not their sum
    avg_launch_s = step_ms / 1e3 / max(nlaunch, 1)
    units_per_launch = units_per_pass_rank * args.steps / max(nlaunch, 1)
    algo_bytes = 52.0 if args.f32 else ALGO_BYTES_PER_UNIT  # fp32: 6 x 4 + 4 read, 6 x 4 written
    achieved = algo_bytes * units_per_pass_rank * args.steps / (busy_ms / 1e3) / 1e9
    concurrency = step_ms / busy_ms

    traffic, valu, traffic_file, traffic_note = measured_traffic(n, chunk, K, args.f32)
    if args.full:
        traffic, valu, traffic_note = None, None, "counter summaries are kept for the LEAN feature set only"

    # the binding roofline, from this run: vector-issue time of one pass (SQ_ACTIVE_INST_VALU of the
    # matching counter summary, x4 cycles, over the chip's 1 024 SIMDs) at the clock sampled DURING the
    # timed passes, against the measured pass time
    valu_issue_frac = clock_mhz = None
    if valu and valu.get("valu_active_quadcycles_per_pass") and not args.f32:
        # (fp64 only: a wave64 fp64 instruction holds its SIMD for four cycles; fp32 instructions issue
        # at twice that rate and the counter does not tell the two apart)
        clock_mhz = sclk[0] or 2400.0
        issue_s = valu["valu_active_quadcycles_per_pass"] * 4.0 / 1024.0 / (clock_mhz * 1e6)
        valu_issue_frac = issue_s / (elapsed / args.steps)
    if rank == 0:
        line = {
            "metric": "point_timesteps_per_s",
            "value": value,
            "unit": "point-timesteps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32" if args.f32 else "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{total_points} synthetic points x {args.hours} h (SimLen {simlen}, DTSecs 30, "
                            f"NLayers 15), {'fp32' if args.f32 else 'fp64'}, {'FULL' if args.full else 'LEAN'} feature set, outputs every time index, "
                            f"attributable to points"
                            + (" (per-launch order rows kept inside the timed region)" if cluster else ""),
                "total_points": total_points,
                "points_per_gpu": n,
                "plans_per_gpu": K,
                "feature_set": "FULL (dew point and observation streams, 6 h initialization phase, relaxation)" if args.full else "LEAN",

                "simlen": simlen,
                "chunk_steps": chunk,
                "plan_order": cluster,
                "sort_key": args.sort_key if cluster else None,
                "order_rows_kept": cluster,
                "inputs_resident": "hourly knots of every point, made once before the timed region (since round "
                                   "3; round 2 regenerated the windows' and previews' knots inside it)",
                "kernel_variant": args.variant,
                "forcing_window": ("none: the step kernel's ground wavefront interpolates the forcing from the "
                                   "resident knots (rs_hip_step_knots)" if fused else
                                   "made per launch by the expansion kernel, read from HBM by the step kernel"),
                "parallelism": f"points sharded over {world} GPU(s) ({scaling} scaling), no collectives",
                "failed_points": int(nfail),
                "checksum": checksum,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "rs::step_kernel_*",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_file + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, "
                                  "mean per launch, averaged over the launches of a pass like "
                                  "avg_launch_ms; stamped with the hash of the kernel sources)" if traffic else None,
                "traffic_note": traffic_note,
                "valu": valu,
                "valu_issue_frac": valu_issue_frac,
                "valu_issue_clock_mhz": clock_mhz,
                "valu_issue_clock_source": (None if valu_issue_frac is None else
                                            "rs_hip_clock_probe: shader-clock / 100 MHz counter ratio, beside the timed passes" if sclk[0]
                                            else "not readable on this box: the 2.4 GHz peak clock assumed"),
                "avg_launch_ms": avg_launch_s * 1e3,
                "launches": nlaunch,
                "units_per_launch": units_per_launch,
                "busy_ms": busy_ms,
                "concurrent_launches": concurrency,
                "per_launch_achieved": algo_bytes * units_per_launch / avg_launch_s / 1e9,
                # in the knot-reading flavour the 52 B/unit of forcing never exist in HBM - what the
                # flavour itself has to move is the outputs plus the knots and the state, amortised
                "flavour_compulsory_bytes_per_unit": (FUSED_BYTES_PER_UNIT if fused and not args.f32 else algo_bytes),
                "flavour_compulsory_achieved": (FUSED_BYTES_PER_UNIT if fused and not args.f32 else algo_bytes)
                                               * units_per_pass_rank * args.steps / (busy_ms / 1e3) / 1e9,
                "method": "achieved = algorithmic bytes of all step launches / union of their HIP-event "
                          "intervals (busy_ms); = units_per_launch x bytes / avg_launch_ms x "
                          "concurrent_launches; per_launch_achieved is the single-launch figure",
                "step_kernel_only_value": units_per_pass_rank * args.steps / (busy_ms / 1e3),
                "note": ("fp32" if args.f32 else "fp64") + "-VALU-bound kernel (SURVEY.md 8d): the HBM fraction "
                        "is reported as the contract asks, the binding roofline is vector-ALU issue"
                        + ("; in this flavour the 52 B/unit of forcing are never read from HBM (they are "
                           "made from the knots in registers), so HBM traffic is below the algorithmic bytes" if fused else ""),
            },
        }
        for k, v in extra.items():
            if isinstance(v, dict):
                line[k + "_value"] = v["value"]
        if extra:
            line["extra_legs"] = extra
            if "f32_config5" in extra:
                line["f32_config5_roofline"] = extra["f32_config5"]["roofline"]
        if natural is not None:
            n_elapsed, n_step_ms, n_nlaunch, n_chunk, n_busy, n_K = natural
            line["natural_order_value"] = units_per_pass_job * args.steps / n_elapsed
            line["natural_order"] = {
                "ms_per_step": n_elapsed / args.steps * 1e3,
                "avg_launch_ms": n_step_ms / max(n_nlaunch, 1),
                "step_kernel_only_value": units_per_pass_rank * args.steps / (n_busy / 1e3),
                "plans_per_gpu": n_K, "chunk_steps": n_chunk,
                "kernel_variant": 1 if fused else args.variant,
                "note": "second timed leg, same W/K and fences: points in natural order, no re-sort"
                        + (", expansion kernel + forcing window, one point per lane" if fused else ""),
            }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample, simlen, args.seed)
        print(json.dumps(line), flush=True)
    for pl in plans:
        pl.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
