"""Diagnostic for the plan order (rs_hip_recluster): which sort key / launch length leaves a
wavefront the fewest boundary-layer passes?  Runs the CPU checker with its per-call trace on N
synthetic points (cached in /tmp), then replays the launch/re-sort schedule on the host:
    cost(wave, step) = max over the wave's 64 lanes of the passes of that step
for several keys.  The kernel's measured figures on 1 M points: natural order 13.9,
sum-of-extra-passes key with 240-index launches 9.7, per-lane mean 5.5."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))

N = int(os.environ.get("N", 16384)); L = 5761
CACHE = f"/tmp/bl_trace_{N}.npz"


def trace():
    if os.path.exists(CACHE):
        z = np.load(CACHE)
        return z["t"], z["u"], z["cover"]
    import oracle_helpers as oh
    from roadsurf_amd import abi
    lib = oh.load("port")
    C.c_int.in_dll(lib, "oracle_bl_hist_on").value = 1
    f = oh.synth_forcing(N, L, seed=20240110)
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    tr = np.zeros((N, 2 * (L + 8)), np.uint8)
    lib.runsimulation.argtypes = [C.POINTER(abi.OutputPointers), C.POINTER(abi.InputPointers),
                                  C.POINTER(abi.InputSettings), C.POINTER(abi.InputParameters),
                                  C.POINTER(abi.LocalParameters)]
    cnt = np.zeros(N, np.int64)
    cover = np.zeros((N, L), bool)
    for q in range(N):
        ip, op, keep = oh.point_pointers(f, q)
        C.c_void_p.in_dll(lib, "oracle_bl_trace").value = tr[q].ctypes.data
        C.c_long.in_dll(lib, "oracle_bl_trace_pos").value = 0
        C.c_long.in_dll(lib, "oracle_bl_trace_cap").value = 2 * (L + 8)
        lib.runsimulation(C.byref(op), C.byref(ip), C.byref(s), C.byref(p), C.byref(l))
        cnt[q] = C.c_long.in_dll(lib, "oracle_bl_trace_pos").value
        o = keep["out"] if isinstance(keep, dict) and "out" in keep else None
    cnt //= 2
    off = int(cnt.min()) - L
    t = tr[:, 2 * off:2 * (off + L):2].copy()
    u = tr[:, 2 * off + 1:2 * (off + L) + 1:2].copy()
    np.savez_compressed(CACHE, t=t, u=u, cover=cover)
    return t, u, cover


def wave_cost(tt):  # [points][steps] -> mean over waves and steps of the wave's max
    g = tt.reshape(tt.shape[0] // 64, 64, tt.shape[1])
    return g.max(axis=1).mean()


def replay(t, u, CH, keyfn, label):
    n = t.shape[0]
    order = np.arange(n)
    hist = []  # per-launch per-point summaries
    tot = 0.0; steps = 0
    tot_u = 0.0
    for c0 in range(0, L, CH):
        cur = t[:, c0:c0 + CH]; curu = u[:, c0:c0 + CH]
        w = cur[order].reshape(n // 64, 64, -1)
        tot += w.max(axis=1).sum(); steps += cur.shape[1] * (n // 64)
        wu = (curu[order] > 0).reshape(n // 64, 64, -1)
        tot_u += wu.any(axis=1).sum()
        hist.append(dict(extra=(cur.astype(np.int32) - 5).sum(1), mx=cur.max(1).astype(np.int32),
                         unst_end=(curu[:, -30:] > 0).any(1), last=cur[:, -1].astype(np.int32),
                         tail=(cur[:, -60:].astype(np.int32) - 5).sum(1),
                         unst_frac=(curu > 0).mean(1)))
        key = keyfn(hist)
        order = np.argsort(-key, kind="stable")
    print(f"{label:70s} passes/wave-step {tot / steps:6.3f}   waves with an unstable lane {tot_u / steps:5.3f}")


if __name__ == "__main__" and not os.environ.get("FORECAST"):
    t, u, _ = trace()
    print(f"{N} points: per-lane mean {t.mean():.3f}, natural order {wave_cost(t):.3f}, "
          f"share of point-steps with no unstable pass {(u == 0).mean():.3f}")
    big = 1 << 20
    for CH in (480, 240, 120, 60):
        replay(t, u, CH, lambda h: h[-1]["unst_end"] * big + np.minimum(h[-1]["extra"], big - 1),
               f"launch {CH}: regime bit + sum of extra passes (round 1 key, no cover bit)")
    for CH in (240, 120):
        replay(t, u, CH, lambda h: h[-1]["unst_end"] * big + h[-1]["mx"] * 4096 + np.minimum(h[-1]["extra"], 4095),
               f"launch {CH}: regime + max passes, ties by sum")
        replay(t, u, CH, lambda h: h[-1]["unst_end"] * big + np.minimum(h[-1]["tail"], big - 1),
               f"launch {CH}: regime + extra passes of the last 60 indices")
        replay(t, u, CH, lambda h: h[-1]["unst_end"] * big + h[-1]["last"] * 4096 + np.minimum(h[-1]["tail"], 4095),
               f"launch {CH}: regime + passes of the last index, ties by last 60")
        replay(t, u, CH, lambda h: h[-1]["unst_end"] * big + np.minimum(2 * h[-1]["extra"] + (h[-2]["extra"] if len(h) > 1 else 0), big - 1),
               f"launch {CH}: regime + 2*last launch + the one before")
        replay(t, u, CH, lambda h: np.minimum(h[-1]["extra"], big - 1),
               f"launch {CH}: sum of extra passes only")
    # bound: perfect foresight of the launch
    for CH in (240, 120, 60):
        n = t.shape[0]; tot = 0.0; steps = 0
        for c0 in range(0, L, CH):
            cur = t[:, c0:c0 + CH]
            order = np.argsort(-(cur.astype(np.int32) - 5).sum(1), kind="stable")
            tot += cur[order].reshape(n // 64, 64, -1).max(axis=1).sum(); steps += cur.shape[1] * (n // 64)
        print(f"launch {CH}: sorted by the launch's own sum (foresight bound): {tot / steps:.3f}")


# ---- forecast key: predict the passes of the NEXT launch from its (known) forcing ----------
def bl_trips(consts, tsurf, tair, vz):
    """CalcBLCondAndLE's trip count (src/BoundaryLayer.f90:64-96), vectorised; a predictor, so
    plain double literals are good enough."""
    f32 = lambda x: float(np.float32(x))
    TaK = tair + f32(273.15)
    dens = 100000.0 / (f32(287.05) * TaK)
    hcap = 1005.0 + (TaK - 250.0) ** 2 / 3364.0
    avc = hcap * dens
    dT = tsurf - tair
    den0 = avc * TaK
    vkvz = consts.VK_Const * vz
    avk = avc * consts.VK_Const
    num = -consts.VK_Const * consts.ZRefT * consts.Grav
    psim = np.zeros_like(tair); psih = np.zeros_like(tair); bl = np.zeros_like(tair)
    trips = np.zeros(tair.shape, np.int32)
    active = np.ones(tair.shape, bool)
    for j in range(1, 41):
        old = bl
        us = vkvz / (consts.logUstar + psim)
        bln = avk * us / (consts.logCond + psih)
        stab = np.minimum(num * bln * dT / (den0 * us ** 3), 1.0)
        st = stab > 0
        ph_s = f32(4.7) * stab
        with np.errstate(invalid="ignore"):
            ph_u = -2.0 * np.log((1.0 + np.sqrt(1.0 - 16.0 * np.minimum(stab, 0.0))) / 2.0)
        nph = np.where(st, ph_s, ph_u); npm = np.where(st, ph_s, f32(0.6) * ph_u)
        psih = np.where(active, nph, psih); psim = np.where(active, npm, psim)
        bl = np.where(active, bln, bl)
        done = active & (j >= 5) & (np.abs(bln - old) < f32(0.001))
        trips[done] = j
        active &= ~done
        if not active.any():
            break
    trips[active] = 40
    return trips


def forecast_eval():
    import oracle_helpers as oh
    from roadsurf_amd import abi, lib
    n = int(os.environ.get("NF", 4096))
    t, u, _ = trace()
    t = t[:n]; u = u[:n]
    f = oh.synth_forcing(n, L, seed=20240110)
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    out, _, _ = oh.run_oracle("port", f, s, p, l)
    consts = lib.build_constants(s, p)
    ts = out["tsurf"]  # [n][L] AFTER each step; Ts seen by step i is ts[:, i-1]
    hour = f["hour"]
    night = (hour >= p.NightOn) | (hour <= p.NightOff)
    calm = np.where(night, p.CalmLimNgt, p.CalmLimDay)
    vz = np.maximum(f["vz"], calm[None, :])
    # sanity: the predictor reproduces the traced passes when fed the true Ts
    i = 3000
    tr = bl_trips(consts, ts[:, i - 1], f["tair"][:, i], vz[:, i])
    print("predictor vs trace at one index: equal for %.4f of the points" % (tr == t[:, i]).mean())
    big = 1 << 20
    for CH in (240, 120):
        for nsamp in (1, 3, 5):
            order = np.arange(n); tot = 0.0; steps = 0
            for c0 in range(0, L, CH):
                cur = t[:, c0:c0 + CH]
                tot += cur[order].reshape(n // 64, 64, -1).max(axis=1).sum(); steps += cur.shape[1] * (n // 64)
                nxt = c0 + CH
                if nxt >= L:
                    break
                ts_now = ts[:, nxt - 1]
                key = np.zeros(n)
                for q in range(nsamp):
                    tau = min(L - 1, nxt + (2 * q + 1) * CH // (2 * nsamp))
                    # Ts is carried along with the air temperature change since the launch start
                    key += bl_trips(consts, ts_now, f["tair"][:, tau], vz[:, tau]) - 5
                order = np.argsort(-key, kind="stable")
            print(f"launch {CH}: forecast key, Ts frozen at the launch start, {nsamp} sample(s): {tot / steps:.3f}")
        for alpha in (0.5, 1.0):
            nsamp = 3
            order = np.arange(n); tot = 0.0; steps = 0
            for c0 in range(0, L, CH):
                cur = t[:, c0:c0 + CH]
                tot += cur[order].reshape(n // 64, 64, -1).max(axis=1).sum(); steps += cur.shape[1] * (n // 64)
                nxt = c0 + CH
                if nxt >= L:
                    break
                ts_now = ts[:, nxt - 1]; ta_now = f["tair"][:, nxt - 1]
                key = np.zeros(n)
                for q in range(nsamp):
                    tau = min(L - 1, nxt + (2 * q + 1) * CH // (2 * nsamp))
                    ta = f["tair"][:, tau]
                    key += bl_trips(consts, ts_now + alpha * (ta - ta_now), ta, vz[:, tau]) - 5
                order = np.argsort(-key, kind="stable")
            print(f"launch {CH}: forecast key, Ts follows the air temperature change x {alpha}, 3 samples: {tot / steps:.3f}")
        # history key on the same subset for comparison
        order = np.arange(n); tot = 0.0; steps = 0
        for c0 in range(0, L, CH):
            cur = t[:, c0:c0 + CH]; curu = u[:, c0:c0 + CH]
            tot += cur[order].reshape(n // 64, 64, -1).max(axis=1).sum(); steps += cur.shape[1] * (n // 64)
            key = (curu[:, -30:] > 0).any(1) * big + np.minimum((cur.astype(np.int32) - 5).sum(1), big - 1)
            order = np.argsort(-key, kind="stable")
        print(f"launch {CH}: history key on the same {n} points: {tot / steps:.3f}")
        tot = 0.0; steps = 0
        for c0 in range(0, L, CH):
            cur = t[:, c0:c0 + CH]
            order = np.argsort(-(cur.astype(np.int32) - 5).sum(1), kind="stable")
            tot += cur[order].reshape(n // 64, 64, -1).max(axis=1).sum(); steps += cur.shape[1] * (n // 64)
        print(f"launch {CH}: foresight bound on the same points: {tot / steps:.3f}")


if __name__ == "__main__" and os.environ.get("FORECAST"):
    forecast_eval()
