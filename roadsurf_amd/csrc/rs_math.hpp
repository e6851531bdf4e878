/*
 * rs_math.hpp — exp and log that return glibc's bits.
 *
 * Why bit-exactness matters here (and nowhere else on the path, where IEEE
 * already defines every result): when a snow layer finally melts away the
 * reference evaluates  snow - 1000*Melted  with Melted = Q2Melt*DT/(WatMHeat*WatDens)
 * and Q2Melt = WatMHeat*WatDens*(snow/1000)/DT computed from that SAME snow
 * (src/Storage.f90:149-153, 422).  The residual is 0 or +-1 ulp and its SIGN
 * decides whether the wear branch (src/Storage.f90:156-162) moves 1.39e-4 mm
 * into the ice storage, which in turn shifts Tsurf by 1e-3..0.2 K for hours.
 * Any last-bit difference in an upstream exp/log therefore flips a branch:
 * measured with OCML's exp/log (<= 1 ulp) 9 of 8192 points left the 1e-6 K
 * gate after 48 h, with a plain table-driven 1-ulp exp 29 of 8192.  At 1e6
 * points "accurate" is not enough; the functions have to agree with the
 * reference's libm bit for bit.
 *
 * The reference build calls glibc 2.35's exp/log; on every FMA-capable x86-64
 * CPU their ifunc resolvers select __exp_fma / __log_fma.  Those are Szabolcs
 * Nagy's table-driven routines (published as ARM Optimized Routines, MIT):
 *   exp:  2^(k/128) table with tail correction, degree-5 polynomial
 *   log:  128-entry {1/c, log c} table, degree-5 polynomial, and a separate
 *         degree-11 path with a double-double head for 0.9375 <= x < 1.0645
 * Below, each is re-implemented OPERATION FOR OPERATION as that libm executes
 * it (every fused multiply-add of the x86 code is a __builtin_fma here, every
 * separate multiply/add stays separate: the file is compiled with
 * -ffp-contract=off), with the data read out of the library
 * (tools/extract_glibc_math.py -> rs_glibc_tables.h).  On gfx950 v_fma_f64 and
 * v_add/mul_f64 are IEEE-754 correctly rounded, so the results are the same
 * bits.  tests/test_hip_math.py checks bit equality against libm on 1e6 arguments.
 * Cost: exp ~22, log ~30-60 VALU instructions (OCML: 45 / 97).
 *
 * Out-of-domain arguments cannot be produced by the model (CheckValues bounds
 * the inputs): exp saturates to inf/0 for |x| >= 512 instead of glibc's gradual
 * over/underflow; log handles 0, negatives, inf, nan and subnormals like glibc.
 */
#pragma once
#include <hip/hip_runtime.h>
#include "rs_glibc_tables.h"

namespace rs {

/* static: this header is included by more than one translation unit */
static __constant__ uint64_t c_gl_exp_tab[256] = {0};
static __constant__ uint64_t c_gl_log_tab[256] = {0};

/* Polynomial coefficients that are the ADDEND of a fused multiply-add whose multiplier is a
 * constant too (fma(r, C3, C2)).  gfx950 is a GFX9-family target: a VOP3 instruction may
 * read ONE scalar operand (SGPR or literal), so the second constant has to sit in VGPRs, and
 * with loop-invariant code motion disabled (register budget) that was two v_mov_b32 per use,
 * ~75 vector instructions per point-step.  From LDS the same value arrives by a broadcast
 * ds_read_b64, which does not occupy the vector ALU. */
enum { RS_K_EXP_SHIFT, RS_K_EXP_C2, RS_K_EXP_C4, RS_K_LOG_A1, RS_K_LOG_A3, RS_K_LOG_B1,
       RS_K_LOG_B4, RS_K_LOG_B7, RS_K_COUNT };
static __constant__ uint64_t c_gl_coef[RS_K_COUNT] = {
    RS_GL_EXP_SHIFT, RS_GL_EXP_C2, RS_GL_EXP_C4, RS_GL_LOG_A1,
    RS_GL_LOG_A3,    RS_GL_LOG_B1, RS_GL_LOG_B4, RS_GL_LOG_B7};

/* LDS copies of the tables; filled by fill_math_tables() at kernel start. */
/* The coefficients that enter as MULTIPLIERS (a scalar operand of the fma): as literals they cost two
 * scalar moves each at every use - 10 per exp, 10 / 22 per log -, from this table in constant memory
 * the compiler fetches a group with one wide scalar load (-DRS_NO_SMEM_COEF: literals, round 1-3). */
enum { RS_S_EXP_INVLN2N, RS_S_EXP_NEGLN2HIN, RS_S_EXP_NEGLN2LON, RS_S_EXP_C3, RS_S_EXP_C5, RS_S_PAD0, RS_S_PAD1, RS_S_PAD2,
       RS_S_LOG_LN2HI, RS_S_LOG_A2, RS_S_LOG_LN2LO, RS_S_LOG_A4, RS_S_LOG_A0, RS_S_PAD3, RS_S_PAD4, RS_S_PAD5,
       RS_S_LOG_B2, RS_S_LOG_B5, RS_S_LOG_B8, RS_S_LOG_B3, RS_S_LOG_B6, RS_S_LOG_B9, RS_S_LOG_B10, RS_S_LOG_B0,
       RS_S_COUNT };
static __constant__ uint64_t c_gl_scoef[RS_S_COUNT] = {
    RS_GL_EXP_INVLN2N, RS_GL_EXP_NEGLN2HIN, RS_GL_EXP_NEGLN2LON, RS_GL_EXP_C3, RS_GL_EXP_C5, 0, 0, 0,
    RS_GL_LOG_LN2HI, RS_GL_LOG_A2, RS_GL_LOG_LN2LO, RS_GL_LOG_A4, RS_GL_LOG_A0, 0, 0, 0,
    RS_GL_LOG_B2, RS_GL_LOG_B5, RS_GL_LOG_B8, RS_GL_LOG_B3, RS_GL_LOG_B6, RS_GL_LOG_B9, RS_GL_LOG_B10, RS_GL_LOG_B0};
typedef const double __attribute__((address_space(4))) *MathCoef;

struct MathTab {
  const uint64_t *expT; /* [128][2]  {tail, sbits}  */
  const double *logT;   /* [128][2]  {invc, logc}   */
  const double *K;      /* [RS_K_COUNT] */
  MathCoef S;           /* [RS_S_COUNT], constant memory */
};

#define RS_MATH_LDS_DOUBLES (512 + RS_K_COUNT)

/* All threads of the workgroup call this (before any early return), then
 * __syncthreads().  lds must hold RS_MATH_LDS_DOUBLES 8-byte words. */
__device__ __forceinline__ MathTab fill_math_tables(double *lds) {
  uint64_t *w = reinterpret_cast<uint64_t *>(lds);
  for (int i = threadIdx.x; i < RS_MATH_LDS_DOUBLES; i += blockDim.x)
    w[i] = (i < 256) ? c_gl_exp_tab[i] : (i < 512) ? c_gl_log_tab[i - 256] : c_gl_coef[i - 512];
  MathTab t;
  t.expT = w;
  t.logT = lds + 256;
  t.K = lds + 512;
  t.S = (MathCoef)c_gl_scoef;
  asm volatile("" : "+s"(t.S)); /* opaque: a load from a table whose initialiser the compiler sees is folded back into literals */
  return t;
}

/* ---- correctly rounded division and square root for NORMAL-RANGE operands ----
 *
 * hipcc expands an IEEE fp64 `a / b` to: 2 x v_div_scale_f64, v_rcp_f64, two
 * Newton steps on the reciprocal (4 fma), q = a*r, rem = fma(-b, q, a),
 * v_div_fmas_f64 (= fma(rem, r, q) plus un-scaling), v_div_fixup_f64 (0, inf,
 * nan) — 11 VALU instructions + wait states, ~46 times per point-step.
 * v_div_scale only rescales when an operand or the quotient nears the ends of
 * the exponent range, and v_div_fixup only patches 0/inf/nan operands.  Every
 * division on this path has operands of moderate magnitude (temperatures,
 * fluxes, storages, O(1e-10..1e10); denominators bounded away from 0 by the
 * model's own guards and by CheckValues), so the bare sequence below returns
 * the same bits in 8 instructions.  Likewise sqrt (argument 1 - 16*Stab >= 1).
 *
 * (One operand the short sequence does not reproduce even in range: -0.0 / b gives +0.0.  No
 * call site can tell: numerators are sums or products of non-negative quantities behind `> 0`
 * guards, except prec/3600, where the sign of a zero is lost in `<= MinPrecmm` and `wat + rain`,
 * and the stability parameter's -VK*ZRefT*g*BLCond*(Ts - Ta), which is -0.0 for Ts == Ta: both
 * zeros fail `Stab > 0`, and 1 - 16*Stab is 1.0 for either.)
 *
 * This is an assumption about the DATA, so it is checked, not trusted: the
 * library built with -DRS_DIV_CHECK evaluates both forms at every call site and
 * counts disagreements in a device counter; tests/test_hip_fastdiv.py runs the
 * 1 M-point x 48 h workload through that build and requires the count to be 0.
 * -DRS_IEEE_DIV switches every call site back to the compiler's expansion. */
/* [0]: both results finite, different VALUES - the kind that would silently change a result;
 * [2]: +0.0 against -0.0 (the bare sequence loses the sign of a zero numerator, see above);
 * [1]: one of them inf/NaN - x/0, x/inf, inf/x, where the bare sequence
 * gives NaN: the operands the boundary-layer guard exists for (rs_physics_body.inc) */
static __device__ unsigned long long g_div_mismatch[3] = {0ull, 0ull, 0ull};
/* -DRS_BL_STATS (an experiment build, `make blstats`): how often the boundary-layer fixed point repeats its
 * bits before the fifth pass - [0] wave-steps, [1] lane-steps, [2..4] wave-steps in which EVERY active lane's
 * (PSIM, PSIH) came out of pass 2 / 3 / 4 as they went in, [5..7] the same counted per lane */
/* [8] sum over wave-steps of the loop's trip count (the wavefront's: its slowest lane's), [9] the same summed
 * per lane; road_condition: [10] wave-steps, [11] the bare-road shortcut taken, every lane with [12] no snow,
 * [13] no ice of either kind, [14] no deposit and no condensation, [15] no water, [16] none of snow / ice / deposit;
 * forcing_prep_tail: [17] wave-steps, [18] the precipitation branch taken; boundary-layer passes as wavefronts issue
 * them: [19] all, [20] with a lane on the unstable arm, [21] with lanes on both paths of its log, [22] with lanes on
 * both arms, [23] with a lane on log's table path; [48] lane-passes, [46] of them on the unstable arm, [47] on log's
 * table path */
#define RS_BL_NSTATS 56 /* [24..31] wave-steps by the wavefront's trip count 5, 6, 7, 8, 9-12, 13-20, 21-39, 40; [32..39] lane-steps likewise; [40..45] wave-steps with 1, 2-3, 4-7, 8-15, 16-31, 32-64 lanes of more than 20 passes */
static __device__ unsigned long long g_bl_stats[RS_BL_NSTATS];
/* the first RS_DIV_SAMPLES finite mismatches: {numerator (or sqrt argument), denominator (0 for
 * sqrt), IEEE result, bare result} */
#define RS_DIV_SAMPLES 64
static __device__ double g_div_samples[RS_DIV_SAMPLES][4];
__device__ __forceinline__ void div_check_note(double q, double f, double a = 0.0, double b = 0.0) {
  if (__double_as_longlong(q) == __double_as_longlong(f) || (q != q && f != f)) return;
  const bool special = !(__builtin_fabs(q) < __builtin_inf()) || !(__builtin_fabs(f) < __builtin_inf());
  const int kind = special ? 1 : (q == f ? 2 : 0); /* q == f with different bits: two zeros */
  const unsigned long long k = atomicAdd(&g_div_mismatch[kind], 1ull);
  if (kind == 0 && k < RS_DIV_SAMPLES) {
    g_div_samples[k][0] = a;
    g_div_samples[k][1] = b;
    g_div_samples[k][2] = q;
    g_div_samples[k][3] = f;
  }
}

/* v_rcp_f64 is good to 2^-24.4 (measured over 4e9 mantissas, tools/rcp_accuracy.hip): one Newton
 * step leaves 2^-48.6, so the compiler's expansion takes two (error e^4).  One third-order step
 * r(1 + e + e^2) leaves e^3 = 2^-73 - as far below the rounding of r as e^4 is - in three
 * instructions instead of four.  (-DRS_DIV_TWO_NEWTON: the two-step form.) */
__device__ __forceinline__ double div_bare(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, r, 1.0);
#ifdef RS_DIV_TWO_NEWTON
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
#else
  e = __builtin_fma(e, e, e);
  r = __builtin_fma(r, e, r);
#endif
  const double q = a * r;
  const double rem = __builtin_fma(-b, q, a);
  return __builtin_fma(rem, r, q);
}

__device__ __forceinline__ double sqrt_bare(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
}

__device__ __forceinline__ double rs_div(double a, double b) {
#if defined(RS_IEEE_DIV)
  return a / b;
#elif defined(RS_DIV_CHECK)
  const double q = a / b, f = div_bare(a, b);
  div_check_note(q, f, a, b);
  return q;
#else
  return div_bare(a, b);
#endif
}

__device__ __forceinline__ double rs_sqrt(double x) {
#if defined(RS_IEEE_DIV)
  return ::sqrt(x);
#elif defined(RS_DIV_CHECK)
  const double q = ::sqrt(x), f = sqrt_bare(x);
  div_check_note(q, f, x, 0.0);
  return q;
#else
  return sqrt_bare(x);
#endif
}

/* a / b for a UNIFORM denominator b whose reciprocal rb = RN(1/b) was computed once on the host
 * (rs_consts_dev.h): the last three steps of the sequence above with rb in place of the refined
 * v_rcp_f64 - q = a*rb is within an ulp of a/b, rem = a - b*q is exact in the fma, and
 * q + rem*rb rounds to the correctly rounded quotient (Markstein's final-step theorem; rb is
 * closer to 1/b than the two-step Newton iterate it replaces).  3 instructions instead of 8 and
 * no quarter-rate v_rcp_f64.  Checked like rs_div: the RS_DIV_CHECK build compares every
 * evaluation with a / b, and tests/test_host_logic.py checks the same formula on the host for
 * every constant on 1e6 numerators. */
__device__ __forceinline__ double rs_div_u(double a, double b, double rb) {
#if defined(RS_IEEE_DIV)
  return a / b;
#else
  const double q0 = a * rb;
  const double rem = __builtin_fma(-b, q0, a);
  const double f = __builtin_fma(rem, rb, q0);
#if defined(RS_DIV_CHECK)
  const double q = a / b;
  div_check_note(q, f, a, b);
  return q;
#else
  return f;
#endif
#endif
}

/* IEEE flavour on request (the boundary-layer guard, rs_physics_body.inc): IEEE = true is the
 * compiler's full expansion (v_div_scale / v_div_fmas / v_div_fixup), exact for every operand */
template <bool IEEE>
__device__ __forceinline__ double rs_dv(double a, double b) {
  if (IEEE) return a / b;
  return rs_div(a, b);
}
/* the data-dependent divisions of the boundary-layer loop (rs_physics_body.inc): with
 * -DRS_BL_FIXUP the bare sequence is followed by v_div_fixup_f64, which supplies IEEE's results for
 * zero / infinite / NaN operands and the sign of a zero quotient */
template <bool IEEE>
__device__ __forceinline__ double rs_dvb(double a, double b) {
  if (IEEE) return a / b;
#if defined(RS_BL_FIXUP) && !defined(RS_IEEE_DIV) && !defined(RS_DIV_CHECK)
  return __builtin_amdgcn_div_fixup(div_bare(a, b), b, a);
#else
  return rs_div(a, b);
#endif
}
template <bool IEEE>
__device__ __forceinline__ double rs_sq(double x) {
  if (IEEE) return ::sqrt(x);
  return rs_sqrt(x);
}

/* x is +0.0, by its bits (x == 0 would admit -0.0) */
__device__ __forceinline__ bool rs_is_pos_zero(double x) { return __double_as_longlong(x) == 0; }
__device__ __forceinline__ bool rs_is_pos_zero(float x) { return __float_as_int(x) == 0; }
__device__ __forceinline__ bool rs_is_neg_zero(double x) { return __double_as_longlong(x) == (long long)0x8000000000000000ull; }
/* true in every ACTIVE lane of the wavefront (lanes that have left the time step do not vote) */
__device__ __forceinline__ bool rs_wave_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0ull; }

/* IEEE maxNum / minNum: the other operand where one is a (quiet) NaN.  The instruction itself:
 * __builtin_fmax makes the compiler canonicalise every operand that comes from memory first
 * (v_max_f64 x, x, x), which doubles the cost. */
__device__ __forceinline__ double rs_fmax(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double rs_fmin(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float rs_fmax(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ float rs_fmin(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ double rs_fabs(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ float rs_fabs(float x) { return __builtin_fabsf(x); }

/* fp32 flavour (BASELINE config 5, tolerance-gated, NOT the parity path): hardware
 * transcendental and reciprocal approximations (~1 ulp), no tables. */
struct MathTab;
__device__ __forceinline__ float rs_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float rs_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
template <bool IEEE>
__device__ __forceinline__ float rs_dv(float a, float b) { return rs_div(a, b); }
template <bool IEEE>
__device__ __forceinline__ float rs_dvb(float a, float b) { return rs_div(a, b); }
template <bool IEEE>
__device__ __forceinline__ float rs_sq(float x) { return rs_sqrt(x); }
__device__ __forceinline__ float rs_exp(const MathTab &, float x) { return __expf(x); }
__device__ __forceinline__ float rs_log(const MathTab &, float x) { return __logf(x); }

__device__ __forceinline__ double gl_d(uint64_t bits) { return __longlong_as_double((long long)bits); }
#ifdef RS_NO_SMEM_COEF
#define RS_GLS(mt, NAME) gl_d(RS_GL_##NAME)
#else
#define RS_GLS(mt, NAME) ((mt).S[RS_S_##NAME])
#endif

/* glibc 2.35 sysdeps/ieee754/dbl-64/e_exp.c as built for x86-64 + FMA. */
__device__ __forceinline__ double rs_exp(const MathTab &mt, double x) {
#ifdef RS_OCML_EXP
  return ::exp(x);
#endif
  const uint64_t ix = (uint64_t)__double_as_longlong(x);
  const uint32_t abstop = (uint32_t)(ix >> 52) & 0x7ffu;
  if (__builtin_expect(abstop - 0x3c9u > 0x3eu, 0)) {
    if (abstop < 0x3c9u) return 1.0 + x; /* |x| < 2^-54 */
    /* |x| >= 512, inf, nan: outside the model's domain (CheckValues keeps every
     * argument within +-300).  Saturate instead of glibc's gradual over/underflow. */
    if (x != x) return x;
    return (x > 0.0) ? __builtin_inf() : 0.0;
  }
  /* x = ln2/N*k + r, k integer, |r| <= ln2/2N */
  const double shift = mt.K[RS_K_EXP_SHIFT];
  double kd = __builtin_fma(x, RS_GLS(mt, EXP_INVLN2N), shift);
  const uint64_t ki = (uint64_t)__double_as_longlong(kd);
  kd = kd - shift;
  double r = __builtin_fma(kd, RS_GLS(mt, EXP_NEGLN2HIN), x);
  r = __builtin_fma(kd, RS_GLS(mt, EXP_NEGLN2LON), r);
  const uint32_t idx = 2u * ((uint32_t)ki & 127u);
  const double tail = gl_d(mt.expT[idx]);
  const uint64_t sbits = mt.expT[idx + 1] + (ki << 45);
  const double r2 = r * r;
  const double p23 = __builtin_fma(r, RS_GLS(mt, EXP_C3), mt.K[RS_K_EXP_C2]);
  const double t0 = r + tail;
  const double p45 = __builtin_fma(r, RS_GLS(mt, EXP_C5), mt.K[RS_K_EXP_C4]);
  const double a = __builtin_fma(p23, r2, t0);
  const double r4 = r2 * r2;
  const double tmp = __builtin_fma(r4, p45, a);
  const double scale = gl_d(sbits);
  return __builtin_fma(scale, tmp, scale);
}

/* glibc 2.35 sysdeps/ieee754/dbl-64/e_log.c as built for x86-64 + FMA. */
__device__ __forceinline__ double rs_log(const MathTab &mt, double x) {
#ifdef RS_OCML_LOG
  return ::log(x);
#endif
  uint64_t ix = (uint64_t)__double_as_longlong(x);
  uint32_t hi32 = (uint32_t)(ix >> 32);
  if (hi32 - 0x3fee0000u < 0x00030900u) {
    /* 1 - 2^-4 <= x < 1 + 0x1.09p-4: polynomial in r = x - 1 with a double-double head.
     * (glibc returns 0 for x == 1 up front, for the sake of directed rounding modes; in
     * round-to-nearest the path below gives the same +0.0 - r, hi, lo and y are all +0 - so the
     * test is not spent here; tests/test_hip_math.py has log(1.0) among its arguments.) */
    const double r = x - 1.0;
    const double p12 = __builtin_fma(r, RS_GLS(mt, LOG_B2), mt.K[RS_K_LOG_B1]);
    const double p45 = __builtin_fma(r, RS_GLS(mt, LOG_B5), mt.K[RS_K_LOG_B4]);
    const double r2 = r * r;
    const double p78 = __builtin_fma(r, RS_GLS(mt, LOG_B8), mt.K[RS_K_LOG_B7]);
    const double p123 = __builtin_fma(r2, RS_GLS(mt, LOG_B3), p12);
    const double p456 = __builtin_fma(r2, RS_GLS(mt, LOG_B6), p45);
    const double r3 = r * r2;
    double p = __builtin_fma(r2, RS_GLS(mt, LOG_B9), p78);
    p = __builtin_fma(r3, RS_GLS(mt, LOG_B10), p);
    p = __builtin_fma(p, r3, p456);
    p = __builtin_fma(p, r3, p123);
    const double t = __builtin_fma(r, 0x1p27, r);
    const double rhi = __builtin_fma(-0x1p27, r, t);
    const double rhi2 = rhi * rhi;
    const double rlo = r - rhi;
    const double hi = __builtin_fma(rhi2, RS_GLS(mt, LOG_B0), r);
    const double d = r - hi;
    const double s = r + rhi;
    double lo = __builtin_fma(rhi2, RS_GLS(mt, LOG_B0), d);
    const double m = RS_GLS(mt, LOG_B0) * rlo;
    lo = __builtin_fma(m, s, lo);
    const double y = __builtin_fma(p, r3, lo);
    return hi + y;
  }
  if (__builtin_expect((uint32_t)(hi32 >> 16) - 0x0010u >= 0x7ff0u - 0x0010u, 0)) {
    /* x <= 0, subnormal, inf, nan: outside the model's domain (its argument is >= 1) */
    if (ix * 2 == 0) return -__builtin_inf();
    if (ix == 0x7ff0000000000000ull) return x;
    if ((hi32 & 0x80000000u) || (hi32 & 0x7ff00000u) == 0x7ff00000u) return __builtin_nan("");
    ix = (uint64_t)__double_as_longlong(x * 0x1p52) - (52ull << 52); /* subnormal: normalise */
    hi32 = (uint32_t)(ix >> 32);
  }
  /* x = 2^k z, z in [OFF, 2 OFF), OFF = 0x3fe6000000000000 */
  const uint32_t tmp_hi = hi32 - 0x3fe60000u;
  const uint32_t i = (tmp_hi >> 13) & 127u;
  const int32_t k = (int32_t)tmp_hi >> 20;
  const uint64_t iz = ix - ((uint64_t)(tmp_hi & 0xfff00000u) << 32);
  const double invc = mt.logT[2 * i], logc = mt.logT[2 * i + 1];
  const double z = gl_d(iz);
  const double r = __builtin_fma(z, invc, -1.0);
  const double kd = (double)k;
  const double w = __builtin_fma(kd, RS_GLS(mt, LOG_LN2HI), logc);
  const double p12 = __builtin_fma(r, RS_GLS(mt, LOG_A2), mt.K[RS_K_LOG_A1]);
  const double hi = r + w;
  const double r2 = r * r;
  double lo = (w - hi) + r;
  lo = __builtin_fma(kd, RS_GLS(mt, LOG_LN2LO), lo);
  const double r3 = r * r2;
  const double p34 = __builtin_fma(r, RS_GLS(mt, LOG_A4), mt.K[RS_K_LOG_A3]);
  const double q = __builtin_fma(r2, RS_GLS(mt, LOG_A0), lo);
  const double p = __builtin_fma(p34, r2, p12);
  const double y = __builtin_fma(r3, p, q);
  return y + hi;
}

}  // namespace rs
