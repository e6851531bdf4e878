"""Device-resident driver of the HIP path (layer 1 of ``include/roadsurf.h``).

PyTorch is plumbing here: it owns the HBM buffers (``torch.empty`` on the
``cuda`` device), the stream and, in ``bench.py``, ``torch.distributed``.  All
arithmetic happens in ``libroadsurf_hip.so``; tensors cross the boundary as raw
device pointers.

Layouts (points are the fastest axis, see ``include/roadsurf.h``)::

    forcing / outputs   tensor[t, p]   shape [nsteps, npoints_padded]
    per-point params    tensor[p]
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import torch

from . import abi, lib

F64_FORCING = ("tair", "tdew", "vz", "rhz", "prec", "sw", "lw", "tsurfobs", "depth")
OUT_FIELDS = ("tsurf", "snow", "water", "ice", "deposit", "ice2")


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def require_gpu() -> None:
    if not torch.cuda.is_available():
        raise RuntimeError("roadsurf_amd needs a HIP device (torch.cuda.is_available() is False); "
                           "there is no CPU path")


@dataclass
class ForcingWindow:
    """Step-resolution forcing for ``nsteps`` consecutive time indices on the device."""
    nsteps: int
    t_stride: int
    tensors: dict = field(default_factory=dict)  # name -> torch tensor or None
    hour_per_point: bool = False

    @classmethod
    def empty(cls, nsteps: int, np_pad: int, device, optional=("tdew", "tsurfobs", "depth"),
              dtype=torch.float64):
        t = {}
        for n in F64_FORCING:
            if n in ("tdew", "tsurfobs", "depth") and n not in optional:
                t[n] = None
            else:
                t[n] = torch.empty((nsteps, np_pad), dtype=dtype, device=device)
        t["precphase"] = torch.empty((nsteps, np_pad), dtype=torch.int32, device=device)
        t["hour"] = torch.empty((nsteps,), dtype=torch.int32, device=device)
        return cls(nsteps, np_pad, t)

    def struct(self, row: int = 0) -> lib.RsForcing:
        f = lib.RsForcing()
        for n in F64_FORCING + ("precphase",):
            t = self.tensors.get(n)
            setattr(f, n, None if t is None else C.c_void_p(t[row].data_ptr()))
        h = self.tensors["hour"]
        f.hour = C.c_void_p(h[row].data_ptr())
        f.t_stride = self.t_stride
        f.hour_pstride = 1 if self.hour_per_point else 0
        for n in ("sw_dir", "lw_net"):
            t = self.tensors.get(n)
            setattr(f, n, None if t is None else C.c_void_p(t[row].data_ptr()))
        sun = self.tensors.get("sun")  # [nsteps, RS_SUN_COLS = 6]
        f.sun = None if sun is None else C.c_void_p(sun[row].data_ptr())
        return f


@dataclass
class OutputWindow:
    nrows: int
    t_stride: int
    tensors: dict
    decimate: int = 1

    @classmethod
    def empty(cls, nrows: int, np_pad: int, device, decimate: int = 1, dtype=torch.float64):
        # rows the simulation never saves read -9999.0, as in the reference (OutputData.cpp:5-13)
        t = {n: torch.full((nrows, np_pad), -9999.0, dtype=dtype, device=device) for n in OUT_FIELDS}
        return cls(nrows, np_pad, t, decimate)

    def struct(self, row0: int) -> lib.RsOutputs:
        o = lib.RsOutputs()
        for n in OUT_FIELDS:
            setattr(o, n, C.c_void_p(self.tensors[n].data_ptr()))
        o.t_stride = self.t_stride
        o.decimate = self.decimate
        o.row0 = row0
        return o


class Plan:
    """RAII wrapper of ``RsPlan``: one shard of points on one GPU, one stream."""

    def __init__(self, npoints: int, settings: abi.InputSettings, params: abi.InputParameters,
                 device: int | None = None, stream: torch.cuda.Stream | None = None):
        require_gpu()
        self.L = lib.load()
        self.device_index = torch.cuda.current_device() if device is None else device
        self.device = torch.device("cuda", self.device_index)
        self.stream = stream or torch.cuda.current_stream(self.device)
        self.settings, self.params = settings, params
        self.consts = lib.build_constants(settings, params)
        self.npoints = npoints
        self._h = self.L.rs_hip_plan_create(self.device_index, npoints, C.byref(self.consts),
                                            C.c_void_p(self.stream.cuda_stream))
        if not self._h:
            raise RuntimeError("rs_hip_plan_create failed: " + lib.last_error())
        self.np_pad = int(self.L.rs_hip_plan_npoints_padded(self._h))
        self._pp_keep = None

    def close(self):
        if getattr(self, "_h", None):
            self.L.rs_hip_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    # -- per-point parameters -------------------------------------------------
    def point_params(self, tbottom, initlen=None, tair_relax=None, vz_relax=None, rh_relax=None,
                     coupling_index=None, coupling_tsurf=None, sky=None):
        """tbottom: float (uniform) or tensor[np_pad]; others tensors or None."""
        if not torch.is_tensor(tbottom):
            tbottom = torch.full((self.np_pad,), float(tbottom), dtype=torch.float64,
                                 device=self.device)
        keep = (tbottom, initlen, tair_relax, vz_relax, rh_relax, coupling_index, coupling_tsurf)
        pp = lib.RsPointParams()
        pp.tbottom = _ptr(tbottom)
        pp.initlen = _ptr(initlen)
        pp.tair_relax = _ptr(tair_relax)
        pp.vz_relax = _ptr(vz_relax)
        pp.rh_relax = _ptr(rh_relax)
        pp.coupling_index = _ptr(coupling_index)
        pp.coupling_tsurf = _ptr(coupling_tsurf)
        if sky is not None:  # dict: sky_view, sin_lat, cos_lat, lon_rad, horizons (tensors)
            pp.sky_view = _ptr(sky["sky_view"]); pp.sin_lat = _ptr(sky["sin_lat"])
            pp.cos_lat = _ptr(sky["cos_lat"]); pp.lon_rad = _ptr(sky["lon_rad"])
            pp.horizons = _ptr(sky.get("horizons"))
            pp.horizon_index = _ptr(sky.get("horizon_index"))  # int32[np_pad]: column of slot s (plan order)
            pp.albedo_surroundings = float(self.params.Albedo_surroundings)
            keep = keep + (sky,)
        self._pp_keep = keep
        return pp

    def uniform_tbottom(self, year=2024, month=1, day=10) -> float:
        return lib.bottom_temperature(self.params, self.consts, year, month, day)

    # -- kernels ----------------------------------------------------------------
    def init_state(self, window: ForcingWindow, pp) -> None:
        f = window.struct(0)
        lib.check(self.L.rs_hip_init_state(self._h, C.byref(f), C.byref(pp)), "rs_hip_init_state")

    def step(self, window: ForcingWindow, out: OutputWindow, pp, t0: int, nsteps: int,
             window_row: int = 0, out_row0: int | None = None) -> None:
        f = window.struct(window_row)
        o = out.struct((t0 - 1 + out.decimate - 1) // out.decimate if out_row0 is None else out_row0)
        lib.check(self.L.rs_hip_step(self._h, C.byref(f), C.byref(o), C.byref(pp), t0, nsteps),
                  "rs_hip_step")

    def step_knots(self, spec, knots: torch.Tensor, out: OutputWindow, pp, t0: int, nsteps: int,
                   out_row0: int | None = None) -> None:
        """expand_ordered + step in one launch without a forcing window (rs_hip_step_knots: the
        two-wavefront flavour's ground wave interpolates the forcing from the knots)."""
        o = out.struct((t0 - 1 + out.decimate - 1) // out.decimate if out_row0 is None else out_row0)
        lib.check(self.L.rs_hip_step_knots(self._h, C.byref(spec), C.c_void_p(knots.data_ptr()), 0, knots.shape[0],
                                           C.byref(o), C.byref(pp), t0, nsteps), "rs_hip_step_knots")

    def step_cpl(self, window: ForcingWindow, out: OutputWindow, pp, t0: int, nsteps: int,
                 window_row: int = 0, out_row0: int | None = None) -> None:
        """A chunk of a coupled run in lock step (rs_hip_step_cpl): no replays, points park."""
        f = window.struct(window_row)
        o = out.struct((t0 - 1 + out.decimate - 1) // out.decimate if out_row0 is None else out_row0)
        lib.check(self.L.rs_hip_step_cpl(self._h, C.byref(f), C.byref(o), C.byref(pp), t0, nsteps),
                  "rs_hip_step_cpl")

    def cpl_replay(self, window: ForcingWindow, out: OutputWindow, pp, t0: int, nsteps: int,
                   window_row: int = 0, out_row0: int | None = None) -> int:
        """The replay rounds of the parked points over [t0, t0+nsteps) (rs_hip_cpl_replay)."""
        f = window.struct(window_row)
        o = out.struct((t0 - 1 + out.decimate - 1) // out.decimate if out_row0 is None else out_row0)
        rounds = C.c_int32(0)
        lib.check(self.L.rs_hip_cpl_replay(self._h, C.byref(f), C.byref(o), C.byref(pp), t0, nsteps,
                                           C.byref(rounds)), "rs_hip_cpl_replay")
        return int(rounds.value)

    def set_precision(self, bits: int) -> None:
        """32: fp32 flavour (LEAN features, float windows); 64: the parity path."""
        lib.check(self.L.rs_hip_set_precision(self._h, bits), "rs_hip_set_precision")

    def set_history_score(self, on: bool) -> None:
        lib.check(self.L.rs_hip_set_history_score(self._h, 1 if on else 0), "rs_hip_set_history_score")

    def coupling_windows_closed(self, closed: bool = True) -> None:
        """Every coupling window (replays included) is behind the plan: re-sorts move the coupling
        scalars only (rs_hip_coupling_windows_closed)."""
        lib.check(self.L.rs_hip_coupling_windows_closed(self._h, 1 if closed else 0),
                  "rs_hip_coupling_windows_closed")

    def set_variant(self, v: int) -> None:
        lib.check(self.L.rs_hip_set_variant(self._h, v), "rs_hip_set_variant")
        self.variant = v

    def sync(self) -> None:
        lib.check(self.L.rs_hip_sync(self._h), "rs_hip_sync")

    def failed_count(self) -> int:
        return int(self.L.rs_hip_failed_count(self._h))

    def first_failed_index(self):
        """numpy int32[npoints]: 0, or the 1-based index at which the point's run was failed."""
        import numpy as np
        out = np.zeros(self.npoints, np.int32)
        lib.check(self.L.rs_hip_first_failed_index(self._h, C.c_void_p(out.ctypes.data)),
                  "rs_hip_first_failed_index")
        return out

    def timing_reset(self) -> None:
        self.L.rs_hip_timing_reset(self._h)

    def timing_step_ms(self):
        n = C.c_int32(0)
        ms = self.L.rs_hip_timing_step_ms(self._h, C.byref(n))
        return float(ms), int(n.value)

    def timing_intervals(self, ref_event: torch.cuda.Event):
        """[(start_ms, stop_ms)] of the step launches since timing_reset(), after ref_event."""
        import numpy as np
        cap = 65536
        a = np.zeros(cap); b = np.zeros(cap)
        n = self.L.rs_hip_timing_intervals(self._h, C.c_void_p(ref_event.cuda_event), C.c_void_p(a.ctypes.data),
                                           C.c_void_p(b.ctypes.data), cap)
        if n < 0:
            raise RuntimeError("rs_hip_timing_intervals failed: " + lib.last_error())
        return list(zip(a[:n].tolist(), b[:n].tolist()))

    def state(self) -> torch.Tensor:
        """Carried state block as a host tensor [RS_NSTATE, np_pad]."""
        nb = self.L.rs_hip_plan_state_bytes(self._h)
        t = torch.empty((nb // 8 // self.np_pad, self.np_pad), dtype=torch.float64)
        lib.check(self.L.rs_hip_state_download(self._h, C.c_void_p(t.data_ptr()), nb),
                  "rs_hip_state_download")
        return t

    def load_state(self, t: torch.Tensor) -> None:
        t = t.contiguous()
        lib.check(self.L.rs_hip_state_upload(self._h, C.c_void_p(t.data_ptr()), t.numel() * 8),
                  "rs_hip_state_upload")

    # -- synthetic workload -----------------------------------------------------
    def synth_knots(self, seed: int, nknots: int, point_offset: int = 0, steps_per_knot: int = 120,
                    start_hour: int = 0):
        spec = lib.RsSynthSpec(seed, point_offset, steps_per_knot, start_hour)
        knots = torch.empty((nknots, lib.RS_KNOT_FIELDS, self.np_pad), dtype=torch.float64,
                            device=self.device)
        lib.check(self.L.rs_hip_synth_knots(self._h, C.byref(spec), C.c_void_p(knots.data_ptr()),
                                            0, nknots), "rs_hip_synth_knots")
        return spec, knots

    # -- plan order (rs_hip_recluster) ---------------------------------------------
    def order(self) -> torch.Tensor:
        """Zero-copy view of the plan's slot -> local point array (int32 [np_pad]).  Valid until
        the next recluster(): clone it to keep the order a window was produced in."""
        ptr = self.L.rs_hip_plan_order(self._h)
        if not ptr:
            raise RuntimeError("rs_hip_plan_order failed: " + lib.last_error())

        class _View:
            __cuda_array_interface__ = {"shape": (self.np_pad,), "typestr": "<i4",
                                        "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(_View(), device=self.device)

    def copy_order_to(self, dst: torch.Tensor) -> None:
        """Keep the order the last launch was stepped in: dst int32[np_pad] on this device
        (asynchronous on the plan's stream; call before recluster())."""
        assert dst.dtype == torch.int32 and dst.numel() == self.np_pad and dst.is_contiguous()
        lib.check(self.L.rs_hip_plan_order_copy(self._h, C.c_void_p(dst.data_ptr())),
                  "rs_hip_plan_order_copy")

    def outputs_by_point(self, out: "OutputWindow", nrows: int, dst: dict, dst_row0: int = 0, order=None,
                         stream: torch.cuda.Stream | None = None) -> None:
        """The first ``nrows`` rows of the output window, [row][slot], into point-major tensors
        ``dst[name][npoints, dst_rows]`` at columns ``dst_row0 ...`` (rs_hip_outputs_by_point); ``order``: a kept
        order row, default the plan's current order (call between the launch and the next re-sort)."""
        o = out.struct(0)
        rows = next(iter(dst.values())).shape[1]
        ptrs = (C.c_void_p * 6)(*[dst[n].data_ptr() for n in OUT_FIELDS])
        lib.check(self.L.rs_hip_outputs_by_point(self._h, C.byref(o), int(nrows),
                                                 C.c_void_p(order.data_ptr()) if order is not None else None,
                                                 ptrs, C.c_int64(rows), C.c_int64(dst_row0),
                                                 C.c_void_p(stream.cuda_stream) if stream is not None else None),
                  "rs_hip_outputs_by_point")

    def reset_order(self) -> None:
        lib.check(self.L.rs_hip_plan_reset_order(self._h), "rs_hip_plan_reset_order")

    def recluster_forecast(self, tair_rows, vz_rows, hours, tair_now, alpha: float = 0.5,
                           mode: int = 1, point_order: bool = False, prec_rows=None, between=None) -> None:
        """Sort the slots by a forecast of the next launch (rs_hip_recluster_forecast): rows are
        tensors [np_pad] at the preview times, hours the hour of day.  Rows in the CURRENT slot order,
        or - ``point_order`` - in point order, read through the plan's order row.  ``between``: per preview
        (tair_b, vz_b, w) - the preview lies between its row and that one (RsPreview::tair_b); tair_now may
        then be None."""
        pv = lib.RsPreview()
        if between is not None:
            for q, (tb, vb, w) in enumerate(between):
                pv.tair_b[q] = tb.data_ptr(); pv.vz_b[q] = vb.data_ptr(); pv.w[q] = float(w)
        pv.index = self.L.rs_hip_plan_order(self._h) if point_order else None
        pv.n = len(tair_rows)
        for q, (ta, vz, h) in enumerate(zip(tair_rows, vz_rows, hours)):
            pv.tair[q] = ta.data_ptr(); pv.vz[q] = vz.data_ptr(); pv.hour[q] = int(h)
        if prec_rows is not None:  # one more key bit: precipitation somewhere in the next window
            for q, pr in enumerate(prec_rows):
                pv.prec[q] = pr.data_ptr()
        pv.tair_now = tair_now.data_ptr() if tair_now is not None else None
        pv.alpha = alpha
        pv.mode = mode
        lib.check(self.L.rs_hip_recluster_forecast(self._h, C.byref(pv)), "rs_hip_recluster_forecast")

    def recluster(self) -> None:
        """Sort the slots by the boundary-layer passes of the last launch; later windows,
        per-point parameters and outputs are in the new slot order."""
        lib.check(self.L.rs_hip_recluster(self._h), "rs_hip_recluster")

    def synth_knots_range(self, spec, knots: torch.Tensor, k0: int, nknots: int, ordered: bool):
        """Knots k0..k0+nknots-1 into the first rows of ``knots``; with ``ordered`` column p
        belongs to point order()[p]."""
        spec.order = self.L.rs_hip_plan_order(self._h) if ordered else None
        lib.check(self.L.rs_hip_synth_knots(self._h, C.byref(spec), C.c_void_p(knots.data_ptr()),
                                            k0, nknots), "rs_hip_synth_knots")

    def expand_range(self, spec, knots: torch.Tensor, k0: int, nknots: int, window: ForcingWindow,
                     t0: int, nsteps: int) -> None:
        f = window.struct(0)
        lib.check(self.L.rs_hip_expand_forcing(self._h, C.byref(spec), C.c_void_p(knots.data_ptr()),
                                               k0, nknots, C.byref(f), t0, nsteps),
                  "rs_hip_expand_forcing")

    def expand_ordered(self, spec, knots: torch.Tensor, window: ForcingWindow, t0: int, nsteps: int) -> None:
        """Knots [nknots][9][np_pad] in POINT order (all of the series, knot 0 first) -> the window in
        the plan's current slot order (rs_hip_expand_forcing_ordered)."""
        f = window.struct(0)
        lib.check(self.L.rs_hip_expand_forcing_ordered(self._h, C.byref(spec), C.c_void_p(knots.data_ptr()),
                                                       0, knots.shape[0], C.byref(f), t0, nsteps),
                  "rs_hip_expand_forcing_ordered")

    def expand(self, spec, knots, window: ForcingWindow, t0: int, nsteps: int,
               stream: torch.cuda.Stream | None = None) -> None:
        f = window.struct(0)
        st = (stream or self.stream).cuda_stream
        lib.check(self.L.rs_hip_expand_forcing_on(self._h, C.byref(spec),
                                                  C.c_void_p(knots.data_ptr()), 0, knots.shape[0],
                                                  C.byref(f), t0, nsteps, C.c_void_p(st)),
                  "rs_hip_expand_forcing_on")


def run_points(forcing: dict, settings: abi.InputSettings, params: abi.InputParameters,
               local, chunk: int = 0, variant: int = 0, device: int = 0,
               lean_if_possible: bool = True, year_month_day=None, history_score: bool | None = None,
               precision: int = 64, horizon_index=None):
    """Run host arrays ``forcing[name][n, SimLen]`` (numpy, reference layout) through the
    device-resident API and return outputs ``[n, SimLen]`` as numpy.  Test/bench helper:
    transposes with torch on the device, windows of ``chunk`` steps (0 = whole series)."""
    import numpy as np

    require_gpu()
    n, L = forcing["tair"].shape
    assert L == settings.SimLen
    dev = torch.device("cuda", device)
    plan = Plan(n, settings, params, device)
    if variant:
        plan.set_variant(variant)
    if history_score is not None:
        plan.set_history_score(history_score)
    if precision == 32:  # the fp32 flavour: windows and outputs hold floats (rs_hip_set_precision)
        plan.set_precision(32)
    wdt = torch.float32 if precision == 32 else torch.float64
    npad = plan.np_pad
    if isinstance(local, abi.LocalParameters):
        local = [local] * n
    initlen = np.array([l.InitLenI for l in local], np.int32)
    relax_on = settings.use_relaxation == 1
    coupled = settings.use_coupling == 1
    skyv = np.array([l.sky_view for l in local])
    sky_on = bool(((skyv < 1.0) & (skyv > -0.01)).any())
    need_full = (not lean_if_possible) or initlen.max() > 1 or settings.force_tsurf == 1 or \
        relax_on or coupled or sky_on or settings.tsurfOutputDepth >= 0 or (forcing["depth"] >= 0).any() \
        or bool(((forcing["tdew"] < -90.0) | (forcing["tdew"] > 100.0)).any())  # the LEAN kernels skip the Tdew check

    def pad_t(a, dtype):  # [n, L] -> device [L, npad]
        t = torch.zeros((L, npad), dtype=dtype, device=dev)
        t[:, :n] = torch.from_numpy(np.ascontiguousarray(a)).to(dev).T
        return t

    tens = {k: pad_t(forcing[k], wdt) for k in ("tair", "vz", "rhz", "prec", "sw", "lw")}
    tens["tsurfobs"] = pad_t(forcing["tsurfobs"], wdt)
    tens["tdew"] = pad_t(forcing["tdew"], wdt) if need_full else None
    # no depth stream where no value of it can act (depth(i) >= 0 takes the surface temperature from the
    # profile at that depth): the kernels read a missing stream as -9999.9, and the two-wavefront flavour
    # has the FULL feature set only without one
    tens["depth"] = pad_t(forcing["depth"], wdt) if need_full and bool((forcing["depth"] >= 0).any()) else None
    tens["precphase"] = pad_t(forcing["precphase"], torch.int32)
    tens["hour"] = torch.from_numpy(np.ascontiguousarray(forcing["hour"])).to(dev)
    sky = None
    if sky_on:
        Lh = lib.load()
        tens["sw_dir"] = pad_t(forcing["sw_dir"], wdt)  # (window streams: the plan's precision; the geometry stays fp64)
        tens["lw_net"] = pad_t(forcing["lw_net"], wdt)
        sun = np.zeros((L, 6))  # RS_SUN_COLS
        ax = [np.ascontiguousarray(forcing[k], np.int32) for k in ("year", "month", "day", "hour", "minute", "second")]
        Lh.rs_sun_table(L, *[C.c_void_p(a.ctypes.data) for a in ax], C.c_void_p(sun.ctypes.data))
        tens["sun"] = torch.from_numpy(sun).to(dev)
        larr = (abi.LocalParameters * n)(*local)
        geo = [np.zeros(n) for _ in range(3)]
        Lh.rs_point_geometry(n, larr, *[C.c_void_p(g.ctypes.data) for g in geo])

        def vec(a):
            t = torch.zeros((npad,), dtype=torch.float64, device=dev)
            t[:n] = torch.from_numpy(np.ascontiguousarray(a))
            return t
        sky = {"sky_view": vec(skyv), "sin_lat": vec(geo[0]), "cos_lat": vec(geo[1]), "lon_rad": vec(geo[2])}
        hz = forcing.get("local_horizons")
        if hz is not None:
            h = torch.zeros((360, npad), dtype=torch.float64, device=dev)
            h[:, :n] = torch.from_numpy(np.ascontiguousarray(hz)).to(dev).T
            sky["horizons"] = h
            if horizon_index is not None:  # point i of this call reads column horizon_index[i] of local_horizons
                hi = torch.zeros((npad,), dtype=torch.int32, device=dev)
                hi[:n] = torch.from_numpy(np.ascontiguousarray(horizon_index, np.int32))
                sky["horizon_index"] = hi
    win = ForcingWindow(L, npad, tens)
    if year_month_day is None:
        year_month_day = (int(forcing["year"][0]), int(forcing["month"][0]), int(forcing["day"][0]))
    tb = plan.uniform_tbottom(*year_month_day)

    def pp_vec(vals, dtype):
        t = torch.zeros((npad,), dtype=dtype, device=dev)
        t[:n] = torch.tensor(vals, dtype=dtype)
        return t

    if need_full:
        pp = plan.point_params(
            tb, pp_vec(initlen.tolist(), torch.int32),
            pp_vec([l.tair_relax for l in local], torch.float64) if relax_on else None,
            pp_vec([l.VZ_relax for l in local], torch.float64) if relax_on else None,
            pp_vec([l.RH_relax for l in local], torch.float64) if relax_on else None,
            pp_vec([l.couplingIndexI for l in local], torch.int32) if coupled else None,
            pp_vec([l.couplingTsurf for l in local], torch.float64) if coupled else None,
            sky)
    else:
        pp = plan.point_params(tb)
    out = OutputWindow.empty(L, npad, dev, dtype=wdt)
    plan.init_state(win, pp)
    if coupled and chunk:
        # time-chunked coupling: lock-step chunks up to the last coupling-window end, the replay
        # rounds over the window block, then the chunks from the first window end on (points whose
        # window ends later wait there; include/roadsurf.h, rs_hip_step_cpl)
        ci = np.array([l.couplingIndexI for l in local]); ct = np.array([l.couplingTsurf for l in local])
        on = ~((ct < -100) | (ci < 1))
        cpl_len = int(settings.coupling_minutes * 60 / settings.DTSecs)
        cs = np.where(ci <= settings.coupling_minutes * 60 / settings.DTSecs, 1, ci - cpl_len)
        stages = [(1, L)]
        if on.any():
            ce_max, ce_min, cs_min = int(ci[on].max()), int(ci[on].min()), int(cs[on].min())
            # the replay window reaches one index beyond the last window end: a point that replays
            # has CheckValues run on index couplingEndI+1 before it rewinds (Simulation.f90:59-66)
            stages = [(1, min(ce_max, L)), ("replay", cs_min, min(ce_max + 1, L)), (ce_min + 1, L)]
        for st in stages:
            if st[0] == "replay":
                plan.cpl_replay(win, out, pp, st[1], st[2] - st[1] + 1, window_row=st[1] - 1, out_row0=0)
                continue
            t0 = st[0]
            while t0 <= st[1]:
                ns = min(chunk, st[1] - t0 + 1)
                plan.step_cpl(win, out, pp, t0, ns, window_row=t0 - 1, out_row0=0)
                t0 += ns
    else:
        chunk = L if coupled else (chunk or L)
        t0 = 1
        while t0 <= L:
            ns = min(chunk, L - t0 + 1)
            plan.step(win, out, pp, t0, ns, window_row=t0 - 1, out_row0=0)
            t0 += ns
    plan.sync()
    res = {k: out.tensors[k][:, :n].T.contiguous().double().cpu().numpy() for k in OUT_FIELDS}
    nfail = plan.failed_count()
    plan.close()
    return res, nfail
