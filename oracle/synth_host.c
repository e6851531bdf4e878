/*
 * oracle/synth_host.c — TEST INFRASTRUCTURE.
 * Host twin of the device synthetic-forcing generator: fills the reference's
 * per-point [SimLen] arrays from roadsurf_amd/csrc/rs_synth.h (same inline
 * arithmetic the HIP kernels compile), so the CPU oracle and the GPU path see
 * bit-identical forcing.
 */
#include "../roadsurf_amd/csrc/rs_synth.h"
#include <stdint.h>
#include <stdlib.h>

void synth_fill_points(uint64_t seed, int64_t point_offset, int32_t n, int32_t simlen,
                       int32_t steps_per_knot, int32_t start_hour, double *tair, double *tdew,
                       double *vz, double *rhz, double *prec, double *sw, double *lw,
                       double *sw_dir, double *lw_net, double *tsurfobs, double *depth,
                       int32_t *precphase, int32_t *hour) {
  const int32_t nk = (simlen - 1) / steps_per_knot + 2;
#pragma omp parallel
  {
    RsSynthKnot *K = (RsSynthKnot *)malloc(sizeof(RsSynthKnot) * (size_t)nk);
#pragma omp for schedule(static)
    for (int32_t p = 0; p < n; ++p) {
      const int64_t o = (int64_t)p * simlen;
      for (int32_t k = 0; k < nk; ++k) K[k] = rs_sy_knot(seed, point_offset + p, k, start_hour);
      for (int32_t i = 1; i <= simlen; ++i) {
        /* same arithmetic as rs_sy_step(), knots hoisted */
        const int32_t t = i - 1, k = t / steps_per_knot, r = t - k * steps_per_knot;
        const RsSynthKnot a = K[k], b = K[k + 1];
        RsSynthStep s;
        if (r == 0) {
          s.tair = a.tair; s.tdew = a.tdew; s.vz = a.vz; s.rhz = a.rhz;
          s.prec = a.prec; s.sw = a.sw; s.lw = a.lw; s.phase = a.phase;
        } else {
          s.tair = rs_sy_lerp(a.tair, b.tair, r, steps_per_knot);
          s.tdew = rs_sy_lerp(a.tdew, b.tdew, r, steps_per_knot);
          s.vz = rs_sy_lerp(a.vz, b.vz, r, steps_per_knot);
          s.rhz = rs_sy_lerp(a.rhz, b.rhz, r, steps_per_knot);
          s.prec = rs_sy_lerp(a.prec, b.prec, r, steps_per_knot);
          s.sw = rs_sy_lerp(a.sw, b.sw, r, steps_per_knot);
          s.lw = rs_sy_lerp(a.lw, b.lw, r, steps_per_knot);
          s.phase = b.phase;
        }
        s.tsurfobs = (i == 1) ? a.tsurf0 : -9999.9;
        tair[o + t] = s.tair; tdew[o + t] = s.tdew; vz[o + t] = s.vz;
        rhz[o + t] = s.rhz; prec[o + t] = s.prec; sw[o + t] = s.sw;
        lw[o + t] = s.lw; sw_dir[o + t] = 0.6 * s.sw; lw_net[o + t] = -40.0;
        tsurfobs[o + t] = s.tsurfobs; depth[o + t] = -9999.9;
        precphase[o + t] = s.phase;
      }
    }
    free(K);
  }
  for (int32_t i = 1; i <= simlen; ++i) hour[i - 1] = rs_sy_hour(i, steps_per_knot, start_hour);
}
