/*
 * rs_synth.h — deterministic synthetic forcing, identical on host and device.
 *
 * The reference ships no forcing data (examples/example1/example_forecast.json
 * is a missing blob, /root/reference/.MISSING_LARGE_BLOBS), so the workload of
 * BASELINE.json configs 2-5 is synthetic.  Shape: HOURLY knots
 * per point (what an NWP source delivers) expanded to the model's DTSecs grid
 * with the linear rule of the reference driver
 * (examples/example1/src/JsonSource.cpp:115-170):
 *     v(t) = k0 + (secs_since_k0 * (k1 - k0)) / secs_between_knots
 * and PrecPhase taken from the NEXT knot between knots (JsonSource.cpp:171-172).
 *
 * Only + - * / on doubles and integer hashing: no libm, no FMA contraction
 * (both sides compile with -ffp-contract=off), so the host twin (used to feed
 * the CPU oracle) and the device generator produce bit-identical arrays.
 *
 * Plain C99 / HIP.  RS_HD expands to __host__ __device__ under hipcc.
 */
#ifndef RS_SYNTH_H
#define RS_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RS_HD __host__ __device__ static inline
#else
#define RS_HD static inline
#endif

/* field ids for the hash */
enum {
  RS_SY_TMEAN = 1, RS_SY_TAMP, RS_SY_TPH, RS_SY_TSYN, RS_SY_TSYNPH, RS_SY_RHMEAN,
  RS_SY_VZMEAN, RS_SY_VZPH, RS_SY_S0, RS_SY_LWMEAN, RS_SY_PFLAG, RS_SY_PSTART,
  RS_SY_PDUR, RS_SY_PRATE, RS_SY_PMODE, RS_SY_TDEWD, RS_SY_TS0,
  RS_SY_N_TAIR = 32, RS_SY_N_RH, RS_SY_N_VZ, RS_SY_N_LW, RS_SY_N_PHASE
};

RS_HD uint64_t rs_sy_mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

/* uniform in [0,1), exact in double */
RS_HD double rs_sy_u(uint64_t seed, int64_t point, int32_t knot, int32_t field) {
  uint64_t h = rs_sy_mix(seed ^ rs_sy_mix((uint64_t)point * 0xD1342543DE82EF95ull +
                                          (uint64_t)(uint32_t)knot * 0x2545F4914F6CDD1Dull +
                                          (uint64_t)(uint32_t)field));
  return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

/* parabolic pseudo-sine of period 1: psin(0)=0, psin(.25)=1, psin(.75)=-1 */
RS_HD double rs_sy_psin(double u) {
  u = u - (double)(int64_t)u;
  if (u < 0.0) u += 1.0;
  if (u < 0.5) {
    double v = 2.0 * u;
    return 4.0 * v * (1.0 - v);
  } else {
    double v = 2.0 * (u - 0.5);
    return -(4.0 * v * (1.0 - v));
  }
}

typedef struct RsSynthKnot {
  double tair, tdew, vz, rhz, prec, sw, lw, tsurf0;
  int32_t phase;
} RsSynthKnot;

/* What a point's knots share: the draws with knot index -1.  Split from rs_sy_knot so that a
 * caller that makes several knots of one point (the device generator) hashes them once. */
typedef struct RsSynthPoint {
  double tmean, tamp, tph, tsyn, tsynph, rhmean, vzmean, vzph, s0, lwmean, tdewd, pflag, prate, pmode, ts0;
  int32_t pstart, pdur;
} RsSynthPoint;

RS_HD RsSynthPoint rs_sy_point(uint64_t seed, int64_t gp) {
  RsSynthPoint c;
  c.tmean = -15.0 + 25.0 * rs_sy_u(seed, gp, -1, RS_SY_TMEAN);
  c.tamp = 1.0 + 7.0 * rs_sy_u(seed, gp, -1, RS_SY_TAMP);
  c.tph = 0.125 * (rs_sy_u(seed, gp, -1, RS_SY_TPH) - 0.5);
  c.tsyn = 6.0 * rs_sy_u(seed, gp, -1, RS_SY_TSYN);
  c.tsynph = rs_sy_u(seed, gp, -1, RS_SY_TSYNPH);
  c.rhmean = 60.0 + 38.0 * rs_sy_u(seed, gp, -1, RS_SY_RHMEAN);
  c.vzmean = 0.2 + 7.8 * rs_sy_u(seed, gp, -1, RS_SY_VZMEAN);
  c.vzph = rs_sy_u(seed, gp, -1, RS_SY_VZPH);
  c.s0 = 400.0 * rs_sy_u(seed, gp, -1, RS_SY_S0);
  c.lwmean = 200.0 + 150.0 * rs_sy_u(seed, gp, -1, RS_SY_LWMEAN);
  c.tdewd = 0.5 + 4.5 * rs_sy_u(seed, gp, -1, RS_SY_TDEWD);
  c.pflag = rs_sy_u(seed, gp, -1, RS_SY_PFLAG);
  c.pstart = (int32_t)(42.0 * rs_sy_u(seed, gp, -1, RS_SY_PSTART));
  c.pdur = 1 + (int32_t)(6.0 * rs_sy_u(seed, gp, -1, RS_SY_PDUR));
  c.prate = 3.0 * rs_sy_u(seed, gp, -1, RS_SY_PRATE);
  c.pmode = rs_sy_u(seed, gp, -1, RS_SY_PMODE);
  c.ts0 = 2.0 * rs_sy_u(seed, gp, -1, RS_SY_TS0);
  return c;
}

/* Hourly knot `k` (k = 0 is absolute time index 1) of the point whose shared draws are `c`. */
RS_HD RsSynthKnot rs_sy_knot_of(const RsSynthPoint *c, uint64_t seed, int64_t gp, int32_t k,
                                int32_t start_hour) {
  RsSynthKnot q;
  const double hod = (double)((k + start_hour) % 24);
  const double day = rs_sy_psin((hod - 9.0) / 24.0 + c->tph);

  q.tair = c->tmean + c->tamp * day + c->tsyn * rs_sy_psin((double)k / 48.0 + c->tsynph) +
           0.5 * (2.0 * rs_sy_u(seed, gp, k, RS_SY_N_TAIR) - 1.0);
  q.tdew = q.tair - c->tdewd;
  double rh = c->rhmean - 10.0 * day + 3.0 * (2.0 * rs_sy_u(seed, gp, k, RS_SY_N_RH) - 1.0);
  if (rh < 20.0) rh = 20.0;
  if (rh > 100.0) rh = 100.0;
  q.rhz = rh;
  double vz = c->vzmean * (1.0 + 0.5 * rs_sy_psin(hod / 12.0 + c->vzph)) +
              (2.0 * rs_sy_u(seed, gp, k, RS_SY_N_VZ) - 1.0);
  if (vz < 0.05) vz = 0.05;
  q.vz = vz;
  double sun = rs_sy_psin((hod - 6.0) / 24.0);
  q.sw = sun > 0.0 ? c->s0 * sun : 0.0;
  q.lw = c->lwmean + 20.0 * (2.0 * rs_sy_u(seed, gp, k, RS_SY_N_LW) - 1.0);

  /* one precipitation event on ~30 % of the points */
  q.prec = 0.0;
  q.phase = -9999;
  if (c->pflag < 0.3) {
    if (k >= c->pstart && k < c->pstart + c->pdur) {
      q.prec = c->prate;
      /* 40 % of the wet points leave the phase missing (model interprets it,
       * src/Cond.f90:221-245), the rest give an explicit form 0..6 per hour */
      if (c->pmode >= 0.4) q.phase = (int32_t)(7.0 * rs_sy_u(seed, gp, k, RS_SY_N_PHASE));
    }
  }
  /* initial surface temperature observation (index 1 only) */
  q.tsurf0 = q.tair - 1.0 + c->ts0;
  return q;
}

/* Hourly knot `k` (k = 0 is absolute time index 1) for global point id `gp`. */
RS_HD RsSynthKnot rs_sy_knot(uint64_t seed, int64_t gp, int32_t k, int32_t start_hour) {
  const RsSynthPoint c = rs_sy_point(seed, gp);
  return rs_sy_knot_of(&c, seed, gp, k, start_hour);
}

typedef struct RsSynthStep {
  double tair, tdew, vz, rhz, prec, sw, lw, tsurfobs;
  int32_t phase;
} RsSynthStep;

RS_HD double rs_sy_lerp(double k0, double k1, int32_t secs, int32_t span) {
  return k0 + ((double)secs * (k1 - k0)) / (double)span;
}

/* Step-resolution forcing at absolute (1-based) time index i. */
RS_HD RsSynthStep rs_sy_step(uint64_t seed, int64_t gp, int32_t i,
                             int32_t steps_per_knot, int32_t start_hour) {
  const int32_t t = i - 1;
  const int32_t k = t / steps_per_knot;
  const int32_t r = t - k * steps_per_knot;
  RsSynthKnot a = rs_sy_knot(seed, gp, k, start_hour);
  RsSynthStep s;
  if (r == 0) {
    s.tair = a.tair; s.tdew = a.tdew; s.vz = a.vz; s.rhz = a.rhz;
    s.prec = a.prec; s.sw = a.sw; s.lw = a.lw; s.phase = a.phase;
  } else {
    RsSynthKnot b = rs_sy_knot(seed, gp, k + 1, start_hour);
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
  return s;
}

/* hour of day at absolute index i (shared axis) */
RS_HD int32_t rs_sy_hour(int32_t i, int32_t steps_per_knot, int32_t start_hour) {
  return (((i - 1) / steps_per_knot) + start_hour) % 24;
}

#endif /* RS_SYNTH_H */
