"""Instruction-cost model of the boundary-layer loop under a plan order (rs_hip_recluster).

For N synthetic points the CPU checker gives Ts(t); a vectorised restatement of the loop
(src/BoundaryLayer.f90:64-96; reproduces the checker's trip counts exactly) then yields, per
point-step, which passes were active and which branch each took: stable, unstable with the log
argument near 1 (glibc's polynomial path), unstable with the table path.  A wavefront pays for the
UNION of its lanes' branches at every pass:

    cost(wave, step) = sum_j  C_COMMON [any lane active at pass j]
                            + C_SQRT   [any unstable lane active at j]
                            + C_NEAR   [any lane on the near-1 log path at j]
                            + C_FAR    [any lane on the table log path at j]

(instruction counts read off the gfx950 ISA of step_kernel_reg).  Different sort keys / launch
lengths are replayed against that cost.  Results go to stdout; a cache of the masks to /tmp."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))

N = int(os.environ.get("N", 4096))
L = 5761
C_COMMON, C_SQRT, C_NEAR, C_FAR, C_STABLE = 45, 19, 38, 35, 3
CACHE = f"/tmp/bl_masks_{N}.npz"


def f32(x):
    return float(np.float32(x))


def bl_masks(consts, tsurf, tair, vz):
    """-> (act, near, far, stab) uint64 masks over the passes (bit j-1 = pass j), unstable flag"""
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
    active = np.ones(tair.shape, bool)
    act = np.zeros(tair.shape, np.uint64); near = np.zeros(tair.shape, np.uint64)
    far = np.zeros(tair.shape, np.uint64); stb = np.zeros(tair.shape, np.uint64)
    for j in range(1, 41):
        bit = np.uint64(1 << (j - 1))
        old = bl
        us = vkvz / (consts.logUstar + psim)
        bln = avk * us / (consts.logCond + psih)
        stab = np.minimum(num * bln * dT / (den0 * us ** 3), 1.0)
        st = stab > 0
        ph_s = f32(4.7) * stab
        arg = (1.0 + np.sqrt(1.0 - 16.0 * np.minimum(stab, 0.0))) / 2.0
        ph_u = -2.0 * np.log(arg)
        isnear = (~st) & (arg >= 0.9375) & (arg < 1.064453125)  # glibc log: 1-2^-4 <= x < 1+0x1.09p-4
        act[active] |= bit
        near[active & isnear] |= bit
        far[active & ~st & ~isnear] |= bit
        stb[active & st] |= bit
        nph = np.where(st, ph_s, ph_u); npm = np.where(st, ph_s, f32(0.6) * ph_u)
        psih = np.where(active, nph, psih); psim = np.where(active, npm, psim)
        bl = np.where(active, bln, bl)
        done = active & (j >= 5) & (np.abs(bln - old) < f32(0.001))
        active &= ~done
        if not active.any():
            break
    return act, near, far, stb


def popcount(x):
    x = x.copy()
    c = np.zeros(x.shape, np.int32)
    for _ in range(40):
        c += (x & np.uint64(1)).astype(np.int32)
        x >>= np.uint64(1)
    return c


def build():
    if os.path.exists(CACHE):
        z = np.load(CACHE)
        return {k: z[k] for k in z.files}
    import oracle_helpers as oh
    from roadsurf_amd import abi, lib
    f = oh.synth_forcing(N, L, seed=20240110)
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    out, _, _ = oh.run_oracle("port", f, s, p, l)
    consts = lib.build_constants(s, p)
    ts = out["tsurf"]
    hour = f["hour"]
    night = (hour >= p.NightOn) | (hour <= p.NightOff)
    calm = np.where(night, p.CalmLimNgt, p.CalmLimDay)
    vz = np.maximum(f["vz"], calm[None, :])
    vz[:, 0] = np.maximum(vz[:, 0], f32(0.4))
    # Ts seen by step i (0-based index i) is the output of step i-1; step 0 sees the initial value
    ts_in = np.concatenate([f["tsurfobs"][:, :1], ts[:, :-1]], axis=1)
    act = np.zeros((N, L), np.uint64); near = act.copy(); far = act.copy(); stb = act.copy()
    B = 256
    for c0 in range(0, L, B):
        sl = slice(c0, min(L, c0 + B))
        a, n_, f_, s_ = bl_masks(consts, ts_in[:, sl], f["tair"][:, sl], vz[:, sl])
        act[:, sl], near[:, sl], far[:, sl], stb[:, sl] = a, n_, f_, s_
    wat = out["water"] + out["snow"] + out["ice"] + out["ice2"] + out["deposit"]
    d = dict(act=act, near=near, far=far, stb=stb, ts_in=ts_in, tair=f["tair"], vz=vz,
             cover=(wat > 0))
    np.savez_compressed(CACHE, **d)
    return d


def wave_cost(d, order, sl):
    """mean modelled instructions per wave-step of the boundary-layer loop for columns sl"""
    def u(x):
        return np.bitwise_or.reduce(x[order, sl].reshape(len(order) // 64, 64, -1), axis=1)
    a, n_, f_, s_ = u(d["act"]), u(d["near"]), u(d["far"]), u(d["stb"])
    c = (C_COMMON * popcount(a) + C_SQRT * popcount(n_ | f_) + C_NEAR * popcount(n_) +
         C_FAR * popcount(f_) + C_STABLE * popcount(s_))
    return c.sum(), popcount(a).sum(), a.size


def replay(d, CH, keyfn, label):
    order = np.arange(N)
    tot = pas = cnt = 0
    for c0 in range(0, L, CH):
        sl = slice(c0, min(L, c0 + CH))
        c, p_, k = wave_cost(d, order, sl)
        tot += c; pas += p_; cnt += k
        nxt = c0 + CH
        if nxt >= L:
            break
        key = keyfn(d, sl, nxt, CH)
        order = np.argsort(-key, kind="stable")
    print(f"{label:84s} instr/wave-step {tot / cnt:7.1f}   passes {pas / cnt:6.3f}")
    return tot / cnt


def lane_cost(d):
    a, n_, f_, s_ = d["act"], d["near"], d["far"], d["stb"]
    c = (C_COMMON * popcount(a) + C_SQRT * popcount(n_ | f_) + C_NEAR * popcount(n_) +
         C_FAR * popcount(f_) + C_STABLE * popcount(s_))
    return c.mean()


# ---- keys -----------------------------------------------------------------------------------
BIG = 1 << 20


def key_history(d, sl, nxt, CH):
    """round-1 key: unstable near the end of the launch | cover | extra passes of the launch"""
    extra = (popcount(d["act"][:, sl]) - 5).sum(1)
    unst = ((d["near"][:, sl] | d["far"][:, sl])[:, -30:] != 0).any(1)
    cover = d["cover"][:, sl.stop - 1]
    return unst * (2 * BIG) + cover * BIG + np.minimum(extra, BIG - 1)


def make_forecast_key(consts, nsamp, alpha, mode):
    def key(d, sl, nxt, CH):
        ts_now = d["ts_in"][:, nxt]
        ta_now = d["tair"][:, nxt]
        cost = np.zeros(N); unst = np.zeros(N); extra = np.zeros(N); nearc = np.zeros(N)
        for q in range(nsamp):
            tau = min(L - 1, nxt + (2 * q + 1) * CH // (2 * nsamp))
            ta = d["tair"][:, tau]
            a, n_, f_, s_ = bl_masks(consts, ts_now + alpha * (ta - ta_now), ta, d["vz"][:, tau])
            extra += popcount(a) - 5
            unst += ((n_ | f_) != 0)
            nearc += popcount(n_) > popcount(f_)
            cost += (C_COMMON * popcount(a) + C_SQRT * popcount(n_ | f_) + C_NEAR * popcount(n_) +
                     C_FAR * popcount(f_))
        if mode == "cost":
            return cost
        if mode == "unst+extra":
            return unst * BIG + np.minimum(extra, BIG - 1)
        if mode == "unst+near+extra":
            return unst * (4 * BIG) + (nsamp - nearc) * BIG * (unst > 0) + np.minimum(extra, BIG - 1)
        raise ValueError(mode)
    return key


if __name__ == "__main__":
    from roadsurf_amd import abi, lib
    d = build()
    consts = lib.build_constants(abi.default_settings(L), abi.default_parameters())
    print(f"{N} points x {L} steps; per-lane modelled cost {lane_cost(d):.1f} instr, "
          f"per-lane passes {popcount(d['act']).mean():.3f}")
    ident = np.arange(N)
    c, p_, k = wave_cost(d, ident, slice(0, L))
    print(f"{'natural order':84s} instr/wave-step {c / k:7.1f}   passes {p_ / k:6.3f}")
    for CH in (240, 120, 60):
        replay(d, CH, key_history, f"launch {CH}: history key (round 1)")
    for CH in (240, 120):
        for nsamp, alpha, mode in ((3, 0.5, "unst+extra"), (3, 0.5, "cost"), (3, 0.5, "unst+near+extra"),
                                   (5, 0.5, "cost"), (4, 0.5, "unst+extra")):
            replay(d, CH, make_forecast_key(consts, nsamp, alpha, mode),
                   f"launch {CH}: forecast key {mode}, {nsamp} samples, Ts follows dTa x {alpha}")
        # foresight bound: the launch's own modelled per-lane cost
        order = np.arange(N); tot = cnt = 0
        for c0 in range(0, L, CH):
            sl = slice(c0, min(L, c0 + CH))
            a, n_, f_, s_ = d["act"][:, sl], d["near"][:, sl], d["far"][:, sl], d["stb"][:, sl]
            own = (C_COMMON * popcount(a) + C_SQRT * popcount(n_ | f_) + C_NEAR * popcount(n_) +
                   C_FAR * popcount(f_)).sum(1)
            order = np.argsort(-own, kind="stable")
            c, p_, k = wave_cost(d, order, sl)
            tot += c; cnt += k
        print(f"{'launch %d: sorted by the launch own modelled cost (foresight)' % CH:84s} instr/wave-step {tot / cnt:7.1f}")
