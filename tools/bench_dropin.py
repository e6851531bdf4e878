#!/usr/bin/env python3
"""The literal drop-in, measured: `runsimulation` called ONCE PER POINT from T threads (what an unchanged
reference driver does: examples/example1/src/roadrunner.cpp:454-497, `-j T`), without and with the
coalescer (ROADSURF_HIP_COALESCE_US), next to the reference's own Fortran on the same host cores.

usage: bench_dropin.py [points=256] [threads=1,16,64] [hours=48]      (prints a table for INTEGRATION.md)
Each configuration runs in a child process (the coalescing window is read once per process)."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(n, T, hours):
    import threading

    import numpy as np
    import oracle_helpers as oh
    from roadsurf_amd import abi, lib
    from test_hip_boundary import _pointers

    L = lib.load()
    SL = hours * 120 + 1
    f = oh.synth_forcing(n, SL, seed=7)
    s = abi.default_settings(SL); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    out = {k: np.full((n, SL), np.nan) for k in oh.F64_OUT}
    ptrs = [_pointers(f, out, pt) for pt in range(n)]

    def one(pt):
        ip, op, _ = ptrs[pt]
        L.runsimulation(C.byref(op), C.byref(ip), C.byref(s), C.byref(p), C.byref(l))

    one(0)  # first call: library and device warm-up
    nxt = [1]
    lock = threading.Lock()
    lat = []

    def worker():
        while True:
            with lock:
                pt = nxt[0]
                nxt[0] += 1
            if pt >= n:
                return
            t = time.perf_counter()
            one(pt)
            lat.append(time.perf_counter() - t)

    t0 = time.perf_counter()
    th = [threading.Thread(target=worker) for _ in range(T)]
    [x.start() for x in th]
    [x.join() for x in th]
    dt = time.perf_counter() - t0
    b = C.c_int64(0); q = C.c_int64(0)
    L.rs_coalesce_stats(C.byref(b), C.byref(q))
    ora, _, _ = oh.run_oracle("ref" if oh.have_ref() else "port", {k: (v[:8] if getattr(v, "ndim", 0) == 2 else v) for k, v in f.items()}, s, p, l)
    same = all(np.array_equal(out[k][:8], ora[k]) for k in oh.F64_OUT)
    print(json.dumps({"points_per_s": (n - 1) / dt, "ms_per_call": 1e3 * sum(lat) / len(lat), "batches": b.value,
                      "points_batched": q.value, "bit_identical_sample": bool(same)}))


def reference(n, hours, threads):
    import oracle_helpers as oh
    from roadsurf_amd import abi
    SL = hours * 120 + 1
    f = oh.synth_forcing(n, SL, seed=7)
    s = abi.default_settings(SL); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    kind = "ref" if oh.have_ref() else "port"
    oh.run_oracle(kind, {k: (v[:8] if getattr(v, "ndim", 0) == 2 else v) for k, v in f.items()}, s, p, l, nthreads=threads)
    t = time.perf_counter()
    oh.run_oracle(kind, f, s, p, l, nthreads=threads, copy_inputs=False)
    dt = time.perf_counter() - t
    return n / dt, 1e3 * dt * threads / n, kind


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
        sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    threads = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,16,64").split(",")]
    hours = int(sys.argv[3]) if len(sys.argv) > 3 else 48
    print(f"# runsimulation called once per point, {n} points x {hours} h (SimLen {hours * 120 + 1}), host arrays in and out")
    print("| caller threads | coalescing window | points/s | ms per call | batches | same bits as the reference |")
    print("|---|---|---|---|---|---|")
    for T in threads:
        for w in (0, 2000):
            if w and T == 1:
                continue
            env = dict(os.environ, ROADSURF_HIP_COALESCE_US=str(w))
            nn = n if T > 1 else min(n, 24)
            r = subprocess.run([sys.executable, __file__, "--child", str(nn), str(T), str(hours)], env=env,
                               capture_output=True, text=True)
            try:
                d = json.loads(r.stdout.strip().splitlines()[-1])
                print(f"| {T} | {'off' if not w else str(w) + ' us'} | {d['points_per_s']:.0f} | {d['ms_per_call']:.1f} | "
                      f"{d['batches'] if w else '-'} | {d['bit_identical_sample']} |", flush=True)
            except Exception:
                print(f"| {T} | {w} | failed: {r.stderr[-300:]!r} |", flush=True)
    import bench
    cores = bench.effective_cpus()
    for th in sorted({1, cores}):
        pps, ms, kind = reference(max(n, 64 * th) if th > 1 else 64, hours, th)
        print(f"| reference Fortran ({kind}), {th} OpenMP thread(s) on the host | - | {pps:.0f} | {ms:.2f} | - | - |", flush=True)
