/*
 * rs_synth_host.hip - host twin of the device synthetic-forcing generator (rs_synth.h: the same inline
 * arithmetic synth_knots_kernel / expand_kernel compile), as the reference driver would hand the series over:
 * per-point [SimLen] arrays in the layout of InputPointers (examples/example1/src/InputData.cpp:5-26).
 * bench.py's host-array leg, __graft_entry__.smoke() and the parity tests take their inputs from here, so
 * that the CPU checker and the GPU path see bit-identical forcing ("the SAME generator (host
 * C++) feeds the oracle").  Host code only; no device call.
 */
#include <cstdint>
#include <vector>

#include "rs_synth.h"

extern "C" void rs_synth_fill_points(uint64_t seed, int64_t point_offset, int32_t n, int32_t simlen,
                                     int32_t steps_per_knot, int32_t start_hour, double *tair, double *tdew,
                                     double *vz, double *rhz, double *prec, double *sw, double *lw,
                                     double *sw_dir, double *lw_net, double *tsurfobs, double *depth,
                                     int32_t *precphase, int32_t *hour) {
  const int32_t nk = (simlen - 1) / steps_per_knot + 2;
#pragma omp parallel
  {
    std::vector<RsSynthKnot> K((size_t)nk);
#pragma omp for schedule(static)
    for (int32_t p = 0; p < n; ++p) {
      const int64_t o = (int64_t)p * simlen;
      for (int32_t k = 0; k < nk; ++k) K[k] = rs_sy_knot(seed, point_offset + p, k, start_hour);
      for (int32_t t = 0; t < simlen; ++t) {
        /* rs_sy_step()'s arithmetic with the knots made once per point */
        const int32_t k = t / steps_per_knot, r = t - k * steps_per_knot;
        const RsSynthKnot &a = K[k], &b = K[k + 1];
        const bool on_knot = r == 0;
        auto at = [&](double va, double vb) { return on_knot ? va : rs_sy_lerp(va, vb, r, steps_per_knot); };
        const double swv = at(a.sw, b.sw);
        tair[o + t] = at(a.tair, b.tair);
        tdew[o + t] = at(a.tdew, b.tdew);
        vz[o + t] = at(a.vz, b.vz);
        rhz[o + t] = at(a.rhz, b.rhz);
        prec[o + t] = at(a.prec, b.prec);
        sw[o + t] = swv;
        lw[o + t] = at(a.lw, b.lw);
        sw_dir[o + t] = 0.6 * swv;
        lw_net[o + t] = -40.0;
        tsurfobs[o + t] = (t == 0) ? a.tsurf0 : -9999.9;
        depth[o + t] = -9999.9;
        precphase[o + t] = on_knot ? a.phase : b.phase;
      }
    }
  }
  for (int32_t i = 1; i <= simlen; ++i) hour[i - 1] = rs_sy_hour(i, steps_per_knot, start_hour);
}
