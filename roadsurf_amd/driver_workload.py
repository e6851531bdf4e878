"""The synthetic raw-series workload of the driver data path (``rs_driver_run``): what
``tools/bench_driver_path.py`` times and ``bench.py`` reports as its extra legs.

A batch of road-weather points the way the reference's driver sees them
(examples/example1/src/roadrunner.cpp:156-278): an hourly forecast source from one hour before the
simulation start to one hour behind its end, and a 10-minute observation source over the first six
hours (air temperature, humidity, wind, road temperature).  Modes: ``plain``, ``relax`` (relaxation
behind the observations), ``coupling`` (relaxation + coupling against the last road-temperature
observation), ``skyview`` (relaxation + per-point sky view and local horizons), ``skycoupling``.
"""
from __future__ import annotations

import time

import numpy as np

from . import abi, driver

START = 1704844800  # 2024-01-10 00:00:00 UTC
OBS_HOURS = 6
MODES = ("plain", "relax", "coupling", "skyview", "skycoupling")


def _pin(a: np.ndarray) -> np.ndarray:
    """A copy of `a` in page-locked host memory (what a C++ caller gets from hipHostMalloc): copies
    from and to it are DMA transfers the runtime does not have to stage."""
    import torch

    t = torch.from_numpy(np.ascontiguousarray(a)).pin_memory()
    b = t.numpy()
    _PINNED_KEEP.append(t)
    return b


_PINNED_KEEP: list = []


def _bench_weather(u: int, nt_fc: int, nt_ob: int, seed: int):
    """bench.py's synthetic weather (the device generator's hourly knots, rs_hip_synth_knots) as the two raw sources."""
    import torch

    from . import device

    plan = device.Plan(u, abi.default_settings(121), abi.default_parameters(), torch.cuda.current_device())
    _, kn = plan.synth_knots(seed, nt_fc)
    k = kn[:, :, :u].permute(1, 2, 0).contiguous().cpu().numpy()  # [field][point][knot]: tair tdew vz rhz prec sw lw tsurf0
    plan.close()
    del kn
    fc = dict(tair=k[0], tdew=k[1], vz=k[2], prec=k[4], sw=k[5], lw=k[6])
    pos = 1.0 + np.arange(nt_ob) / 6.0  # the forecast's axis starts an hour before the observations'
    i0 = np.minimum(pos.astype(np.int64), nt_fc - 2)
    w = pos - i0

    def at_obs(a):
        return np.ascontiguousarray(a[:, i0] + w[None, :] * (a[:, i0 + 1] - a[:, i0]))

    ob = dict(tair=at_obs(k[0]), rhz=at_obs(k[3]), vz=at_obs(k[2]), tsurfobs=at_obs(k[7]))
    return fc, ob


class DriverWorkload:
    """Inputs of one batch.  ``unique``: the series are generated for that many points and tiled up
    to ``n`` (bounds the host time spent making inputs; the regimes inside a tile are what a batch
    of that size has)."""

    def __init__(self, n: int, hours: int = 48, seed: int = 1, unique: int | None = None, pinned: bool = False,
                 missing: float = 0.0, ragged: float = 0.0, weather: str = "driver"):
        self.n, self.hours, self.pinned = n, hours, pinned
        self.simlen = hours * 120 + 1
        u = n if unique is None else min(unique, n)
        rs = np.random.RandomState(seed)
        reps = -(-n // u)

        def tile(a):
            a = np.ascontiguousarray(np.tile(a, (reps, 1))[:n]) if u < n else a
            return _pin(a) if pinned else a

        def series(nt, dt, lo, hi, amp, period=86400.0):
            base = rs.uniform(lo, hi, (u, 1))
            ph = rs.uniform(0, 2 * np.pi, (u, 1))
            t = np.arange(nt)[None, :] * dt
            return base + amp * np.sin(2 * np.pi * t / period + ph)

        nt_fc = hours + 3
        self.fc_t = START - 3600 + np.arange(nt_fc, dtype=np.int64) * 3600
        tair = series(nt_fc, 3600, -12, 6, 4.0)
        fc = dict(tair=tair, tdew=tair - rs.uniform(0.5, 4, (u, 1)),
                  vz=np.abs(series(nt_fc, 3600, 1, 8, 2.0, 43200.0)) + 0.2,
                  prec=np.where(rs.rand(u, nt_fc) < 0.1, rs.uniform(0, 2, (u, nt_fc)), 0.0),
                  sw=np.maximum(0.0, series(nt_fc, 3600, -50, 150, 200.0)),
                  lw=series(nt_fc, 3600, 230, 320, 15.0))
        nt_ob = OBS_HOURS * 6 + 1
        self.ob_t = START + np.arange(nt_ob, dtype=np.int64) * 600
        ob = dict(tair=series(nt_ob, 600, -12, 6, 1.0), rhz=np.clip(series(nt_ob, 600, 70, 95, 5.0), 5, 100),
                  vz=np.abs(series(nt_ob, 600, 1, 8, 1.0)) + 0.2, tsurfobs=series(nt_ob, 600, -10, 4, 1.0))
        if weather == "bench":
            # A/B (tools/experiments/r6_driver_weather.sh): the weather bench.py's device-resident legs run on
            # (csrc/rs_synth.h: a daily cycle and a synoptic wave per point, one precipitation event on 30 % of the
            # points) as hourly forecast series, the observations on the same lines - what of the distance between
            # the driver legs and the FULL leg is the workload, and what the code path
            fc, ob = _bench_weather(u, nt_fc, nt_ob, seed)
        elif weather != "driver":
            raise ValueError("weather: 'driver' or 'bench'")
        if missing > 0.0:
            # stations as real networks have them: some without an air-temperature / humidity / wind sensor (no
            # value of that variable in the observation source at all), and gaps in the road-temperature series -
            # the lanes of a wavefront then disagree on which source supplies a variable
            for name in ("tair", "rhz", "vz"):
                ob[name][rs.rand(u) < missing] = -9999.9
            ob["tsurfobs"][rs.rand(u, nt_ob) < missing] = -9999.9
            ob["tsurfobs"][:, -1] = np.where(ob["tsurfobs"][:, -1] < -9000, -3.0, ob["tsurfobs"][:, -1])  # (coupling needs the last one)
        if ragged > 0.0:
            # ... and some whose road-temperature observations end one to three hours before the others': their
            # coupling windows end that much earlier (the reference's operational example has such stations)
            late = rs.rand(u) < ragged
            cut = nt_ob - 6 * rs.randint(1, 4, u)
            ob["tsurfobs"][late[:, None] & (np.arange(nt_ob)[None, :] >= cut[:, None])] = -9999.9
        self.fc = {k: tile(v) for k, v in fc.items()}
        self.ob = {k: tile(v) for k, v in ob.items()}
        self.sky_view = tile(rs.uniform(0.3, 1.0, (u, 1)))[:, 0]
        self._hz_unique = rs.uniform(0, 20, (u, 360))
        self._hz = None
        self._tile = tile
        self.cal = driver.calendar(START, self.simlen, 30)

    def horizons(self) -> np.ndarray:
        if self._hz is None:
            self._hz = self._tile(self._hz_unique)  # (pinned with the rest if the workload is)
        return self._hz

    def sources(self, mode: str):
        fc = dict(self.fc)
        if mode in ("skyview", "skycoupling"):
            fc["sw_dir"] = 0.6 * fc["sw"]
            fc["lw_net"] = np.full_like(fc["lw"], -40.0)
            if self.pinned:
                fc["sw_dir"], fc["lw_net"] = _pin(fc["sw_dir"]), _pin(fc["lw_net"])
        return [driver.RawSource(self.fc_t, fc, False), driver.RawSource(self.ob_t, self.ob, True)]

    def settings(self, mode: str, tsurf_output_depth: float | None = None) -> abi.InputSettings:
        if mode not in MODES:
            raise ValueError(f"mode: one of {MODES}")
        s = abi.default_settings(self.simlen)
        s.use_relaxation = 1 if mode != "plain" else 0
        s.use_coupling = 1 if mode in ("coupling", "skycoupling") else 0
        if tsurf_output_depth is not None:
            s.tsurfOutputDepth = float(tsurf_output_depth)
        return s

    def local(self, mode: str):
        loc = driver._locals(self.n, None)
        if mode in ("skyview", "skycoupling"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)  # numpy guesses the struct's dtype: it guesses right
                a = np.ctypeslib.as_array(loc)
            q = np.arange(self.n)
            a["lat"], a["lon"], a["sky_view"] = 60.0 + (q % 97) * 0.05, 22.0 + (q % 89) * 0.05, self.sky_view
        return loc

    def raw_bytes(self, mode: str) -> int:
        b = sum(a.nbytes for d in (self.fc, self.ob) for a in d.values())
        if mode in ("skyview", "skycoupling"):
            b += 2 * self.fc["sw"].nbytes + self.n * 360 * 8
        return b

    def time_calls(self, mode: str, reps: int = 3, warm: int = 1, device: int = -1,
                   tsurf_output_depth: float | None = None, verbose: bool = False, pause: float = 0.0):
        """``warm`` untimed calls (the first one allocates), then ``reps`` timed ones.
        Returns (best seconds, all timed seconds, last result).  device -1: the library's own fan-out
        (ROADSURF_HIP_DEVICES / ROADSURF_HIP_PLANS_PER_DEVICE)."""
        src, s, p = self.sources(mode), self.settings(mode, tsurf_output_depth), abi.default_parameters()
        loc = self.local(mode)
        hz = self.horizons() if mode in ("skyview", "skycoupling") else None
        r, times = None, []
        if self.pinned:  # result arrays in page-locked memory too
            step, n_out = driver.output_rows(s)
            r = {k: _pin(np.full((self.n, n_out), np.nan)) for k in driver.OUT_FIELDS}
            r["status"] = np.empty(self.n, np.int32)
            r["missing_index"] = np.empty(self.n, np.int32)
        for rep in range(warm + reps):
            if pause and rep:
                time.sleep(pause)  # (profiling: lets a kernel trace tell the calls apart)
            t0 = time.perf_counter()
            r = driver.run(src, s, p, START, START + OBS_HOURS * 3600, cal=self.cal, local=loc, horizons=hz,
                           device=device, out=r)
            dt = time.perf_counter() - t0
            if rep >= warm:
                times.append(dt)
            if verbose:
                print(f"rep {rep}: n={self.n} L={self.simlen} mode={mode}: {dt:.3f} s  -> "
                      f"{self.n * self.simlen / dt:.3e} point-timesteps/s (raw in {self.raw_bytes(mode) / 1e9:.2f} GB, "
                      f"ok={int((r['status'] == 0).sum())})", flush=True)
        return min(times), times, r
