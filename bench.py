#!/usr/bin/env python3
"""bench.py — throughput of the RoadSurf hot path on MI355X.

Metric (BASELINE.json): point-timesteps/s at 1 M points x 48 h, fp64.

One bench "step" = one full pass of the hot path over the workload:
    init kernel  ->  for every time chunk:  expand kernel (hourly knots -> DTSecs
    grid, the device twin of the reference driver's interpolation)  ->  step
    kernel (the model: 5 761 time indices per point in all).
Inputs resident in HBM before the timed region: the hourly forcing knots of every
point (what an NWP source delivers).  Step-resolution forcing for 1 M points x
5 761 indices is 300 GB (> 288 GB HBM), so it is produced per chunk
on the device and consumed from HBM by the step kernel; the six outputs are
written every time index (reference SaveOutput semantics) into a chunk buffer.

Multi-GPU: one process per GPU (torch.distributed.run), points sharded with no
data-path collective; the only collectives are the timing barrier and the MAX
over ranks.  Weak scaling: every GPU gets --points points.

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
    import oracle_helpers as oh
    from roadsurf_amd import abi

    kind = "ref" if os.path.exists(oh.REF_SO) else "port"
    s = abi.default_settings(simlen)
    p = abi.default_parameters()
    l = abi.default_local()
    l.InitLenI = 1
    f = oh.synth_forcing(sample_points, simlen, seed=seed)
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


def measured_traffic(points: int, chunk: int):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary
    (profiles/r01_traffic.json), valid for the configuration it was collected on."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        t = json.load(open(path))
    except (OSError, ValueError):
        return None, None
    if t.get("points") != points or t.get("chunk_steps") != chunk:
        return None, None
    return t.get("traffic_bytes_per_launch"), t.get("derived")


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", type=int, default=1_000_000, help="points per GPU")
    ap.add_argument("--hours", type=int, default=48)
    ap.add_argument("--chunk", type=int, default=240, help="time indices per step-kernel launch")
    ap.add_argument("--variant", type=int, default=0, help="0 auto, 1 register profile, 2 LDS profile")
    ap.add_argument("--seed", type=int, default=20240110)
    ap.add_argument("--overlap", action="store_true",
                    help="double-buffer: expand window c+1 on a side stream while stepping window c "
                         "(measured: no gain, the step kernel owns the whole register file; DESIGN.md 6)")
    ap.add_argument("--f32", action="store_true",
                    help="BASELINE config 5 flavour: fp32 state/forcing/outputs/arithmetic "
                         "(tolerance-gated, not the parity path); default is fp64")
    ap.add_argument("--cluster", type=int, default=1,
                    help="1: re-sort the plan's slots by boundary-layer passes after every launch "
                         "(rs_hip_recluster; windows are generated in slot order), 0: natural order")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=16384)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from roadsurf_amd import abi, device, sharding

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
        if ndev >= world:
            dist.init_process_group("nccl", device_id=dev)  # RCCL; used for barrier + MAX only
        else:  # rehearsal: more ranks than GPUs on this box -> ranks share cards, gloo for control
            print(f"[bench] {world} ranks on {ndev} GPU(s): sharing devices, gloo control plane",
                  file=sys.stderr)
            dist.init_process_group("gloo")

    spk = 120                      # 3600 s / DTSecs 30 s
    simlen = args.hours * spk + 1  # examples/example1/src/InputSettings.cpp:98
    settings = abi.default_settings(simlen)
    params = abi.default_parameters()
    n = args.points
    plan = device.Plan(n, settings, params, dev_index)
    if args.variant:
        plan.set_variant(args.variant)
    wdtype = torch.float32 if args.f32 else torch.float64
    if args.f32:
        plan.set_precision(32)
    npad = plan.np_pad
    nknots = args.hours + 2
    offset, _ = sharding.weak_shard(n, rank)
    spec, knots = plan.synth_knots(args.seed, nknots, point_offset=offset, steps_per_knot=spk)
    chunk = min(args.chunk, simlen)
    overlap = args.overlap
    nbuf = 2 if overlap else 1
    wins = [device.ForcingWindow.empty(chunk, npad, dev, optional=(), dtype=wdtype) for _ in range(nbuf)]
    out = device.OutputWindow.empty(chunk, npad, dev, dtype=wdtype)
    # index-1 window for the init kernel: needs TsurfObs(1)
    win0 = device.ForcingWindow.empty(1, npad, dev, optional=("tsurfobs",), dtype=wdtype)
    pp = plan.point_params(plan.uniform_tbottom(2024, 1, 10))
    main = plan.stream
    side = torch.cuda.Stream(dev) if overlap else main
    starts = list(range(1, simlen + 1, chunk))
    ev_filled = [torch.cuda.Event() for _ in range(nbuf)]    # window b holds fresh forcing
    ev_consumed = [torch.cuda.Event() for _ in range(nbuf)]  # step kernel is done with window b

    cluster = bool(args.cluster) and not overlap
    kbuf = torch.empty((chunk // spk + 3, 9, npad), dtype=torch.float64, device=dev) if cluster else None

    def clustered_pass():
        """Like the plain pass, with the plan's slots re-sorted after every launch: the knots
        of each window are generated in the current slot order, so the window is born coalesced
        in that order; the re-sort (hipCUB radix sort + state permutation) is inside the timing."""
        plan.synth_knots_range(spec, kbuf, 0, 2, ordered=True)
        plan.expand_range(spec, kbuf, 0, 2, win0, 1, 1)
        plan.init_state(win0, pp)
        for t0 in starts:
            ns = min(chunk, simlen - t0 + 1)
            k0 = (t0 - 1) // spk
            nk = (t0 + ns - 2) // spk + 1 - k0 + 1
            plan.synth_knots_range(spec, kbuf, k0, nk, ordered=True)
            plan.expand_range(spec, kbuf, k0, nk, wins[0], t0, ns)
            plan.step(wins[0], out, pp, t0, ns, out_row0=t0 - 1)
            plan.recluster()

    def one_pass():
        """init -> per window: expand (HBM-bound, side stream) || step (VALU-bound, main stream).
        Double-buffered: expansion of window c+1 overlaps stepping of window c."""
        if cluster:
            return clustered_pass()
        plan.expand(spec, knots, win0, 1, 1)
        plan.init_state(win0, pp)
        if not overlap:
            for t0 in starts:
                ns = min(chunk, simlen - t0 + 1)
                plan.expand(spec, knots, wins[0], t0, ns)
                plan.step(wins[0], out, pp, t0, ns, out_row0=t0 - 1)
            return
        side.wait_stream(main)
        for c, t0 in enumerate(starts):
            b = c % 2
            ns = min(chunk, simlen - t0 + 1)
            if c >= 2:
                side.wait_event(ev_consumed[b])
            plan.expand(spec, knots, wins[b], t0, ns, stream=side)
            ev_filled[b].record(side)
            main.wait_event(ev_filled[b])
            plan.step(wins[b], out, pp, t0, ns, out_row0=t0 - 1)
            ev_consumed[b].record(main)
        main.wait_stream(side)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        one_pass()
    fence()
    plan.timing_reset()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    fence()
    elapsed = time.perf_counter() - t_start
    elapsed = sharding.max_over_ranks(
        elapsed, dist if world > 1 else None,
        dev if (world > 1 and dist.get_backend() == "nccl") else None)
    step_ms, nlaunch = plan.timing_step_ms()
    nfail = plan.failed_count()

    units_per_pass = n * simlen
    value = world * units_per_pass * args.steps / elapsed
    # dominant kernel: step kernel, HIP events on its own stream around every launch
    avg_launch_s = step_ms / 1e3 / max(nlaunch, 1)
    units_per_launch = units_per_pass * args.steps / max(nlaunch, 1)
    algo_bytes = 52.0 if args.f32 else ALGO_BYTES_PER_UNIT  # fp32: 6 x 4 + 4 read, 6 x 4 written
    achieved = algo_bytes * units_per_launch / avg_launch_s / 1e9

    traffic, valu = (None, None) if args.f32 else measured_traffic(n, chunk)
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
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.f32 else "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{n} synthetic points per GPU x {args.hours} h (SimLen {simlen}, "
                            f"DTSecs 30, NLayers 15), {'fp32' if args.f32 else 'fp64'}, outputs every time index",
                "points_per_gpu": n,
                "simlen": simlen,
                "chunk_steps": chunk,
                "overlap_expand_with_step": overlap, "recluster_after_every_launch": cluster,
                "kernel_variant": args.variant,
                "parallelism": f"points sharded over {world} GPU(s), no collectives",
                "failed_points": int(nfail),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "rs::step_kernel_*",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, "
                                  "mean per launch, averaged over the launches of a pass like "
                                  "avg_launch_ms)" if traffic else None,
                "valu": valu,
                "avg_launch_ms": avg_launch_s * 1e3,
                "launches": nlaunch,
                "units_per_launch": units_per_launch,
                "step_kernel_only_value": units_per_pass * args.steps / (step_ms / 1e3),
                "note": "fp64-VALU-bound kernel (SURVEY.md 8d): the HBM fraction is reported "
                        "as the contract asks, the binding roofline is vector fp64 issue",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample, simlen, args.seed)
        print(json.dumps(line), flush=True)
    plan.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
