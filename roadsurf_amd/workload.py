"""The BASELINE synthetic workload as ONE object, so that ``bench.py`` times exactly the code
path the parity tests check (``tests/test_hip_golden_and_scale.py``).

One pass = the hot path over every time index of the shard's points::

    init kernel -> for every window of `chunk` indices:
        hourly knots (in the plan's current slot order) -> expand to the DTSecs grid ->
        step kernel -> [plan order: keep the order row of this launch, re-sort the slots]

With the two-wavefront flavour in plan order the expansion is part of the step kernel (its ground
wave interpolates the forcing from the knots, rs_hip_step_knots) and no forcing window exists.

Outputs are attributable to points (reference ``SaveOutput`` is per point,
src/InputOutput.f90:151-165): with plan order on, launch ``c`` wrote column ``s`` of its
output window for local point ``orders[c, s]``; the row is copied (4 B per point and launch,
inside the timed region) before the slots are re-sorted for the next launch.
"""
from __future__ import annotations

import torch

from . import device, lib

SPK = 120  # 3600 s / DTSecs 30 s: time indices per hourly knot
# fields of the forecast sort key, most significant first (forecast_key_kernel): 3 something on the
# road, 7 unstable previews, 8 of which on the table path of log, 6 predicted extra passes (saturating
# at 31), 5 storage class - 12 bits, sorted by the plan's own counting pass.  Measured equal in vector
# instructions per wave-step to the 21-bit key 3124 of round 2 (+ the storage class: -9).
# 9 (round 4): the ground digit - which layers are frozen, 7 bits, sorted by a pass of its own below the
# others: a wavefront whose 64 points all have a layer frozen reads that layer's capDZ from the plan's
# constants instead of evaluating the heat capacity of water (layer_step): about half of the 15 x 64
# layer updates of a winter wave-step (1 M points: 2.11e10 -> 2.29e10).
# 0 instead of 6 (round 4, last pass): the predicted extra passes saturating at 7 (3 bits) - a 10-bit key, whose
# classes hold more points (the frost depth lines up better inside them), whose sort is cheaper and for which
# the wave table of the two-wavefront flavour switches itself off (it costs a small shard's chain more than
# class-aligned wavefronts save once the chain kernels run at raised priority): +1 % at 1 M and 125 000
# points, +3 % at 250 000 (profiles/r04_key_layouts.txt).
DEFAULT_FORECAST_MODE = 378059


class SyntheticRun:
    def __init__(self, plan: device.Plan, seed: int, hours: int, chunk: int, point_offset: int = 0,
                 plan_order: bool = True, f32: bool = False, year_month_day=(2024, 1, 10),
                 forecast: bool = True, forecast_alpha: float = 0.5, forecast_mode: int = DEFAULT_FORECAST_MODE,
                 full: bool = False, initlen: int = 720, depth_stream: bool = False):
        self.plan, self.seed, self.hours = plan, seed, hours
        self.simlen = hours * SPK + 1  # examples/example1/src/InputSettings.cpp:98
        self.chunk = min(chunk, self.simlen)
        self.plan_order = plan_order
        # Small shards in plan order (the plan was given the two-wavefront flavour): no forcing window - the
        # kernel's ground wave interpolates the forcing of the next index from the resident knots itself, with
        # the expansion kernel's arithmetic (rs_hip_step_knots).  The expansion is the longest link of the chain
        # between two step launches of a plan, and a small shard has nothing to hide it behind.
        self.fused = bool(plan_order and not f32 and not (full and depth_stream) and getattr(plan, "variant", 0) == 3
                          and plan.consts.NLayers == 15)
        # fp32 (round 6): the two-points-per-lane kernel reads the knots itself too, in either order (variants 1
        # and 2 keep round 2-5's one point per lane with a forcing window, for A/B)
        if f32 and not (full and depth_stream) and plan.consts.NLayers == 15 and getattr(plan, "variant", 0) not in (1, 2):
            self.fused = True
        # sort key of the re-sort: forecast of the next window (rs_hip_recluster_forecast) or the
        # history of the last one (rs_hip_recluster)
        self.forecast, self.forecast_alpha, self.forecast_mode = forecast, forecast_alpha, forecast_mode
        # round 5: one more key bit for "precipitation in the next window" (profiles/r05_ab_precip_bit.txt) and
        # previews placed inside the window (profiles/r05_ab_previews_in_window.txt); attributes, not environment
        # knobs, since round 6
        self.precip_bit = True
        self.previews_in_window = True
        plan.set_history_score(not (plan_order and forecast))  # nobody reads it then
        dev, npad = plan.device, plan.np_pad
        wdtype = torch.float32 if f32 else torch.float64
        # full: the FULL feature set as an operational run has it - Tdew and TsurfObs streams present, an
        # initialization phase of `initlen` indices and (if the plan's settings say so) relaxation
        # towards per-point targets behind it.  No depth stream unless asked for: the reference driver
        # never fills it (examples/example1/src/InputData.cpp:18), and rs_driver_run passes none; with
        # depth_stream the window has one (all -9999.9: rounds 2-4a timed the leg that way), which keeps
        # the launch away from the two-wavefront flavour.  The per-point parameters are the same for every
        # point here, so the plan order does not have to move them.
        self.full = full
        opt = (("tdew", "tsurfobs", "depth") if depth_stream else ("tdew", "tsurfobs")) if full else ()
        self.win = (None if self.fused else
                    device.ForcingWindow.empty(self.chunk, npad, dev, optional=opt, dtype=wdtype))
        self.out = device.OutputWindow.empty(self.chunk, npad, dev, dtype=wdtype)
        # index-1 window for the init kernel: needs TsurfObs(1)
        self.win0 = device.ForcingWindow.empty(1, npad, dev, optional=("tsurfobs",), dtype=wdtype)
        if full:
            il = torch.full((npad,), int(initlen), dtype=torch.int32, device=dev)
            rel = [torch.full((npad,), v, dtype=torch.float64, device=dev) for v in (-3.0, 2.5, 85.0)]
            self.pp = plan.point_params(plan.uniform_tbottom(*year_month_day), il, *rel)
        else:
            self.pp = plan.point_params(plan.uniform_tbottom(*year_month_day))
        self.starts = list(range(1, self.simlen + 1, self.chunk))
        # the hourly knots of every point (what an NWP source delivers): resident in HBM in POINT order,
        # made once.  In plan order the windows are produced in the current SLOT order by reading the
        # knots through the plan's order row (rs_hip_expand_forcing_ordered), and the re-sort's previews
        # are knot rows read the same way - nothing is regenerated after a re-sort (round 2 regenerated
        # the window's and the previews' knots in slot order for every launch: 0.7 ms per launch cycle of
        # 250 000 points beside the step kernels)
        self.spec, self.knots = plan.synth_knots(seed, hours + 2, point_offset=point_offset,
                                                 steps_per_knot=SPK)
        self.orders = (torch.empty((len(self.starts), npad), dtype=torch.int32, device=dev)
                       if plan_order else None)

    def run_pass(self, on_launch=None) -> None:
        """Enqueue one pass on the plan's stream.  ``on_launch(c, t0, ns)`` is called after launch
        ``c`` has been enqueued (tests read ``self.out`` and ``self.orders[c]`` there)."""
        for _ in self.iter_pass(on_launch):
            pass

    def iter_pass(self, on_launch=None):
        """run_pass as a generator that yields after every launch has been enqueued, so that a
        caller can interleave the launches of several plans (one stream each) from one thread."""
        plan, spec = self.plan, self.spec
        if self.plan_order:
            plan.reset_order()
            plan.expand_ordered(spec, self.knots, self.win0, 1, 1)
        else:
            plan.expand(spec, self.knots, self.win0, 1, 1)
        plan.init_state(self.win0, self.pp)
        for c, t0 in enumerate(self.starts):
            ns = min(self.chunk, self.simlen - t0 + 1)
            if self.fused:
                plan.step_knots(spec, self.knots, self.out, self.pp, t0, ns, out_row0=t0 - 1)
            else:
                if self.plan_order:
                    plan.expand_ordered(spec, self.knots, self.win, t0, ns)
                else:
                    plan.expand(spec, self.knots, self.win, t0, ns)
                plan.step(self.win, self.out, self.pp, t0, ns, out_row0=t0 - 1)
            if self.plan_order:
                plan.copy_order_to(self.orders[c])  # which point each column of this launch is
                if on_launch:
                    on_launch(c, t0, ns)
                self._resort(t0 + ns)
            elif on_launch:
                on_launch(c, t0, ns)
            yield c

    def _resort(self, t_next: int) -> None:
        """Re-sort the slots for the window that starts at index t_next."""
        plan = self.plan
        if t_next > self.simlen:
            return
        if not self.forecast:
            plan.recluster()
            return
        # previews = the hourly knots that fall into the next window, rows of the resident knot block
        # (point order, read through the order row); field 0 is Tair, field 2 is VZ (rs_synth.h); the
        # window starts on a knot when chunk is a multiple of SPK, else the nearest earlier knot stands
        # in for "now"
        ns = min(self.chunk, self.simlen - t_next + 1)
        k0 = (t_next - 1) // SPK
        k1 = (t_next + ns - 2) // SPK + 1
        nk = min(k1 - k0 + 1, 8)
        hours = [(self.spec.start_hour + k0 + q) % 24 for q in range(nk)]
        kn = self.knots
        # ... field 4 the precipitation: a window has precipitation at some index iff one of its knots has
        prec = [kn[k0 + q, 4] for q in range(nk)] if self.precip_bit else None
        if self.previews_in_window and kn.shape[0] >= 2 and ((t_next - 1) % SPK != 0 or ns % SPK != 0):
            # a window that does not start and end on knots: previews AT its first and last index, on the
            # knots' straight line as the forcing itself (RsPreview::tair_b), instead of at the knots around it - a
            # half-hour window is not forecast from a knot half an hour old.  (Windows of whole hours keep their
            # knots: measured level or 0.5 % better, profiles/r05_ab_previews_in_window.txt)
            # (two: the middle index, and five or seven previews, sort no better, profiles/r05_sweep_previews.txt)
            idx = [t_next - 1, t_next + ns - 2]  # (with the middle index as a third preview: 0.4 % slower)
            ks = [min(i // SPK, kn.shape[0] - 2) for i in idx]
            plan.recluster_forecast([kn[k, 0] for k in ks], [kn[k, 2] for k in ks],
                                    [(self.spec.start_hour + i // SPK) % 24 for i in idx], None, self.forecast_alpha,
                                    self.forecast_mode, point_order=True, prec_rows=prec,
                                    between=[(kn[k + 1, 0], kn[k + 1, 2], min(1.0, (i - k * SPK) / SPK)) for i, k in zip(idx, ks)])
            return
        plan.recluster_forecast([kn[k0 + q, 0] for q in range(nk)], [kn[k0 + q, 2] for q in range(nk)],
                                hours, kn[k0, 0], self.forecast_alpha, self.forecast_mode, point_order=True,
                                prec_rows=prec)

    def slots_of(self, c: int, points: torch.Tensor) -> torch.Tensor:
        """Columns of launch ``c``'s output window that hold the given local points."""
        if not self.plan_order:
            return points
        order = self.orders[c].long()
        inv = torch.empty_like(order)
        inv[order] = torch.arange(order.numel(), device=order.device)
        return inv[points]
