/*
 * rs_kernels_f32.hip — single-precision flavour of the model (BASELINE.json configs[4]:
 * "fp32 kernels with fp64 tolerance gate").  NOT the parity path: state, forcing, outputs and
 * arithmetic are fp32, exp/log/rcp/sqrt are the hardware approximations.  The reference's storage
 * logic branches on rounding residuals (rs_math.hpp), so an fp32 run cannot track an fp64 run point
 * by point; tests/test_hip_f32.py gates the DISTRIBUTION of the differences against the fp64 oracle,
 * with the tolerance written there.  Every feature of the model (round 6): LEAN and, for NLayers = 15, the FULL set
 * of the two-wavefront kernels - dew-point test, observation forcing during an initialization phase, relaxation, and
 * from a forcing window sky view with local horizons (step_kernel_f32duo<., ., true, true>); coupling, an output
 * depth and the FULL set / sky view at other layer counts through the general kernel at the end of the file
 * (step_kernel_f32_coupled: one point per lane, a time index per lane).
 *
 * Round 6: step_kernel_f32duo, TWO POINTS PER LANE and two wavefronts per 128 points (NLayers = 15).
 * What decides the organisation is how a gfx950 SIMD issues fp32 (tools/f32_issue.hip,
 * profiles/r06_f32_issue_rates.txt): every vector instruction of a wavefront's DEPENDENT stream takes a
 * four-cycle issue slot, fp32 or fp64, packed or not (v_fma_f32 4.3-4.6 cycles each however many wavefronts
 * share the SIMD; 2.35 only behind an independent twin; v_pk_fma_f32 4.1-4.7).  The model is a dependent
 * chain, so one point per lane in fp32 issues at the fp64 rate (round 2-5's flavour: 885 vector instructions
 * per 64 point-steps, 1.6 x the fp64 rate only because rcp / exp / log are single instructions).  Here a lane
 * owns the points 2l and 2l+1 of its workgroup's 128: state, forcing and temporaries are float2
 * (ext_vector_type), the arithmetic is v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 - two points per slot -,
 * compares, selects and transcendentals come in pairs (the 15-layer update alone: 1.52 x per point,
 * tools/f32_layer_probe.hip, profiles/r06_f32_layer_probe.txt).  Beside that:
 *   - two wavefronts per 128 points as in the fp64 flavour (ground wave: layers 3-15 and the forcing's share
 *     of a step one index ahead; surface wave: the chain from the surface state), 96 registers each, five
 *     wavefronts per SIMD.  One wavefront holding everything needed 168-250 registers, two or three per SIMD,
 *     and ran at 4.1e10 - what one point per lane had given;
 *   - no forcing window and no expansion kernel: the ground wave interpolates its points' forcing from the
 *     resident hourly knots (rs_hip_step_knots on an fp32 plan; rs_hip_step still reads a window);
 *   - explicit fused multiply-adds (the translation unit is still compiled with -ffp-contract=off: what is
 *     contracted is what is written, so a point's bits cannot depend on how the compiler scheduled the wavefront
 *     it sits in - tests/test_hip_f32.py::test_fp32_plan_order_changes_no_value);
 *   - the boundary-layer fixed point (src/BoundaryLayer.f90:64-96) with ONE reciprocal per pass instead of
 *     three: UStar = VK VZ / a and BLCond = avk UStar / b give BLCond = (avk VK VZ) / (a b) and
 *     1 / (den0 UStar^3) = a^3 / (den0 (VK VZ)^3), whose constant part leaves the loop - the same iteration,
 *     the same exit test, other roundings (this flavour is gated by a distribution, not by bits); which points
 *     are still iterating is two wavefront masks on the scalar unit;
 *   - frozen layers from constants where all 128 points of the workgroup have the layer frozen, made ON THE
 *     DEVICE by the lanes' own expression (prepare_constants_f32: a host-made constant differs in the last
 *     bit from v_rcp_f32's, and a point's bits would depend on its wavefront); the water polynomials in Horner
 *     form; a layer's four constants by one scalar load issued a layer ahead;
 *   - CheckValues' forcing tests an hour at a time where both knots keep a margin to the limits;
 *   - the storages' state machine (src/Storage.f90, src/Cond.f90: compares and selects, nothing to pack) for the
 *     lane's two points in ONE basic block (x2_road_condition, x2_melting: the one-point source's operations, as
 *     selects), so that the two independent streams issue behind one another; the wavefront-uniform shortcuts stay.
 * Other layer counts keep one point per lane with the profile in LDS (step_kernel_f32_lds).
 */
#include <hip/hip_runtime.h>
#include "rs_math.hpp"
#include "rs_skyview.hpp"
#include "rs_const_f32.h"
#include "rs_state.h"
#include "rs_synth.h"
#include "rs_kernels.h"

#define RS_REAL float
#define RS_NS rs32
#define RS_CONSTS RsConstantsF __attribute__((address_space(4)))
#define R4(x) (x##f)
#define RS_DIVC(a, b, rb) rs_div(a, b)
#define RS_BL_GUARD 0
#define RS_MELTDEN (c.WatMHeat * c.WatDens)
#define RS_CHK(c, i, lit) (lit)
#define RS_PREC_FAST(c) ((c).MinPrecmm >= 0.f)
#define RS_BARE_FAST(c) ((c).MaxWatmms >= 0.f && (c).MaxSnowmms >= 0.f && (c).MaxIcemms >= 0.f && (c).MaxDepmms >= 0.f)
#define RS_LK(c, j, name) ((c).name[j])
/* (no RS_FROZEN_TABLE here: the fp32 division of this flavour is not the correctly rounded one, so a
 * host-made capDZ of a frozen layer would differ from the kernel's in the last bit and a point's values
 * would depend on which wavefront it shares - measured: the plan-order pass lost its checksum equality) */
namespace rs32 {
using rs::MathTab;
using rs::rs_div;
using rs::rs_dv;
using rs::rs_dvb;
using rs::rs_sq;
using rs::rs_exp;
using rs::rs_fabs;
using rs::rs_fmax;
using rs::rs_fmin;
using rs::rs_is_pos_zero;
using rs::rs_wave_all;
using rs::rs_log;
using rs::rs_sqrt;
}  // namespace rs32
#include "rs_physics_body.inc"

namespace rs32 {

constexpr int kBlock = RS_BLOCK;
typedef RsConstantsF __attribute__((address_space(4))) ConstsAS;
template <class Args>
__device__ __forceinline__ const ConstsAS &consts_of(Args a) {
  return *(const ConstsAS *)a->consts;
}

template <int NL>
struct RegProfile {
  float v[NL];
  static constexpr bool kUnrolled = true; /* the layer count is a compile-time constant */
  __device__ __forceinline__ constexpr int nlayers() const { return NL; }
  __device__ __forceinline__ float get(int j) const { return v[j - 1]; }
  __device__ __forceinline__ void set(int j, float x) { v[j - 1] = x; }
  __device__ __forceinline__ void pin() {}
};
struct LdsProfile {
  float *col;
  int n;
  static constexpr bool kUnrolled = false;
  __device__ __forceinline__ int nlayers() const { return n; }
  __device__ __forceinline__ float get(int j) const { return col[(j - 1) * kBlock]; }
  __device__ __forceinline__ void set(int j, float x) { col[(j - 1) * kBlock] = x; }
  __device__ __forceinline__ void pin() {}
};

typedef const rs::StepArgs __attribute__((address_space(4))) *KernArgs;

template <class Prof>
__device__ __forceinline__ void run(const rs::StepArgs &a, Prof &T) {
  KernArgs ka = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
  const uint32_t lane = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * kBlock;
  const int64_t p = row0 + lane;
  const int64_t np = a.np_pad;
  float *st = reinterpret_cast<float *>(a.state);
  const int N = T.nlayers();
  Scalars s;
  for (int j = 1; j <= N; ++j) T.set(j, st[(int64_t)(RS_ST_TMP0 + j - 1) * np + p]);
  s.tsurf = st[(int64_t)RS_ST_TSURF * np + p];
  s.wat = st[(int64_t)RS_ST_WAT * np + p]; s.snow = st[(int64_t)RS_ST_SNOW * np + p];
  s.ice = st[(int64_t)RS_ST_ICE * np + p]; s.ice2 = st[(int64_t)RS_ST_ICE2 * np + p];
  s.dep = st[(int64_t)RS_ST_DEP * np + p]; s.q2melt = st[(int64_t)RS_ST_Q2MELT * np + p];
  s.t4melt = st[(int64_t)RS_ST_T4MELT * np + p]; s.albedo = st[(int64_t)RS_ST_ALBEDO * np + p];
  s.verycold = st[(int64_t)RS_ST_VERYCOLD * np + p] != 0.f;
  s.failed = st[(int64_t)RS_ST_FAILED * np + p] != 0.f;
  s.tair_end = s.vz_end = s.rh_end = 0.f;
  const float tbot = (float)(ka->pp.tbottom + row0)[lane];
  const int32_t nsteps = ka->nsteps, t0 = ka->t0;
  rs::MathTab mt{nullptr, nullptr, nullptr}; /* fp32 exp/log take no tables */
  int32_t score = 0, regime = 0;

  for (int32_t k = 0; k < nsteps; ++k) {
    asm volatile("" : "+s"(ka));
    const ConstsAS &c = consts_of(ka);
    const int32_t i = t0 + k;
    const int64_t row = (int64_t)k * ka->f.t_stride + row0;
    int64_t r = (int64_t)(i - 1);
    const int32_t dec = ka->o.decimate;
    bool write = true;
    if (dec > 1) {
      write = (r % dec == 0);
      r /= dec;
    }
    const int64_t orow = (r - ka->o.row0) * ka->o.t_stride + row0;
#define F32IN(ptr) (reinterpret_cast<const float *>(ka->f.ptr) + row)[lane]
#define F32OUT(ptr) (reinterpret_cast<float *>(ka->o.ptr) + orow)[lane]
    if (s.failed) {
      if (write) {
        F32OUT(tsurf) = -9999.0f; F32OUT(snow) = -9999.0f; F32OUT(water) = -9999.0f;
        F32OUT(ice) = -9999.0f; F32OUT(deposit) = -9999.0f; F32OUT(ice2) = -9999.0f;
      }
      continue;
    }
    Forcing f;
    f.tair = F32IN(tair); f.vz = F32IN(vz); f.rhz = F32IN(rhz); f.prec = F32IN(prec);
    f.sw = F32IN(sw); f.lw = F32IN(lw);
    f.phase = (ka->f.precphase + row)[lane];
    f.hour = ka->f.hour_pstride ? (ka->f.hour + row)[lane] : ka->f.hour[k];
    f.tdew = 0.f; f.tsurfobs = -9999.9f; f.depth = -9999.9f;
    if (i == 1 && f.vz < 0.4f) f.vz = 0.4f;
    const float prec_ts = rs_div(f.prec, 3600.0f) * c.DTSecs;
    if (i < c.SimLen && check_values(c, f, s.tsurf, false)) {
      s.failed = true;
      st[(int64_t)RS_ST_FAILED * np + p] = (float)i; /* the index it was raised at */
    }
    s.tnw1 = T.get(1);
    s.tnw2 = T.get(2);
    const Fluxes fx = model_step_fluxes(c, mt, s, f.tair, f.vz, f.rhz, prec_ts, f.sw, f.lw, f.phase,
                                        f.hour);
    /* sort key of rs_hip_recluster, as in the fp64 kernels (rs_kernels.hip, bl_score_key) */
    score += (fx.trips & 63) - 5;
    if ((fx.trips & 64) && k >= nsteps - 30) regime = 1;
    model_step_ground(c, s, T, tbot, f.tair, fx, f.depth);
    if (write) {
      F32OUT(tsurf) = s.tsurf; F32OUT(snow) = s.snow; F32OUT(water) = s.wat;
      F32OUT(ice) = s.ice; F32OUT(deposit) = s.dep; F32OUT(ice2) = s.ice2;
    }
  }
  for (int j = 1; j <= N; ++j) st[(int64_t)(RS_ST_TMP0 + j - 1) * np + p] = T.get(j);
  st[(int64_t)RS_ST_TSURF * np + p] = s.tsurf;
  st[(int64_t)RS_ST_WAT * np + p] = s.wat; st[(int64_t)RS_ST_SNOW * np + p] = s.snow;
  st[(int64_t)RS_ST_ICE * np + p] = s.ice; st[(int64_t)RS_ST_ICE2 * np + p] = s.ice2;
  st[(int64_t)RS_ST_DEP * np + p] = s.dep; st[(int64_t)RS_ST_Q2MELT * np + p] = s.q2melt;
  st[(int64_t)RS_ST_T4MELT * np + p] = s.t4melt; st[(int64_t)RS_ST_ALBEDO * np + p] = s.albedo;
  st[(int64_t)RS_ST_VERYCOLD * np + p] = s.verycold ? 1.f : 0.f;
  {
    const int32_t lo = score > 0x7ffff ? 0x7ffff : (score < 0 ? 0 : score);
    const int32_t covered = (s.wat > 0.f || s.snow > 0.f || s.ice > 0.f || s.ice2 > 0.f || s.dep > 0.f) ? 1 : 0;
    st[(int64_t)RS_ST_BLSCORE * np + p] = (float)(lo | (covered << 19) | (regime << 20));
  }
}

__global__ void __launch_bounds__(kBlock, 4) step_kernel_f32_lds(const rs::StepArgs a) {
  extern __shared__ float ldsf[];
  if ((int64_t)blockIdx.x * kBlock + threadIdx.x >= a.npoints) return;
  LdsProfile T{ldsf + threadIdx.x, consts_of(&a).NLayers};
  run(a, T);
}

/* fp32 twin of init_kernel (src/Initialization.f90:238-308). */
__global__ void __launch_bounds__(kBlock) init_kernel_f32(const rs::InitArgs a) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= a.npoints) return;
  const ConstsAS &c = consts_of(&a);
  const int N = c.NLayers;
  const float tair = reinterpret_cast<const float *>(a.f.tair)[p];
  const float tobs = a.f.tsurfobs ? reinterpret_cast<const float *>(a.f.tsurfobs)[p] : -9999.9f;
  const float tbot = (float)a.pp.tbottom[p];
  const float t4 = (tobs > -100) ? tobs : tair;
  const int64_t np = a.np_pad;
  float *st = reinterpret_cast<float *>(a.state);
  for (int i = 1; i <= 4; ++i) st[(int64_t)(RS_ST_TMP0 + i - 1) * np + p] = t4;
  for (int i = 5; i <= N; ++i)
    st[(int64_t)(RS_ST_TMP0 + i - 1) * np + p] =
        t4 + (tbot - t4) / (c.ZDpth[N + 1] - c.ZDpth[4]) * (c.ZDpth[i] - c.ZDpth[4]);
  st[(int64_t)RS_ST_TNW1 * np + p] = t4; st[(int64_t)RS_ST_TNW2 * np + p] = t4;
  float tsurf0 = (t4 + t4) / 2.0f;
  if (a.f.depth) { /* getTempAtDepth on the fresh profile, src/Initialization.f90:129-136 */
    const float depth = reinterpret_cast<const float *>(a.f.depth)[p];
    if (depth >= 0.f) {
      if (__builtin_fabsf(depth) < 0.00001f) tsurf0 = t4;
      else if (depth > c.ZDpth[N + 1]) tsurf0 = tbot;
      else {
        tsurf0 = 0.f;
        for (int k = 1; k <= N; ++k)
          if (depth > c.ZDpth[k] && depth <= c.ZDpth[k + 1]) {
            const float tk = st[(int64_t)(RS_ST_TMP0 + k - 1) * np + p];
            const float tk1 = (k == N) ? tbot : st[(int64_t)(RS_ST_TMP0 + k) * np + p];
            tsurf0 = tk + (depth - c.ZDpth[k]) * (tk1 - tk) / (c.ZDpth[k + 1] - c.ZDpth[k]);
            break;
          }
      }
    }
  }
  st[(int64_t)RS_ST_TSURF * np + p] = tsurf0;
  st[(int64_t)RS_ST_WAT * np + p] = 0.f; st[(int64_t)RS_ST_SNOW * np + p] = 0.f;
  st[(int64_t)RS_ST_ICE * np + p] = 0.f; st[(int64_t)RS_ST_ICE2 * np + p] = 0.f;
  st[(int64_t)RS_ST_DEP * np + p] = 0.f; st[(int64_t)RS_ST_Q2MELT * np + p] = 0.f;
  st[(int64_t)RS_ST_T4MELT * np + p] = c.T4Melt0; st[(int64_t)RS_ST_ALBEDO * np + p] = c.Albedo0;
  st[(int64_t)RS_ST_VERYCOLD * np + p] = 0.f; st[(int64_t)RS_ST_FAILED * np + p] = 0.f;
  st[(int64_t)RS_ST_BLSCORE * np + p] = 0.f;
  st[(int64_t)RS_ST_TAIR_END * np + p] = 0.f; st[(int64_t)RS_ST_VZ_END * np + p] = 0.f;
  st[(int64_t)RS_ST_RH_END * np + p] = 0.f;
  if (c.use_coupling) { /* initCoupling, src/Coupling.f90:144-169 */
    st[(int64_t)RS_ST_CPL_ITER * np + p] = 0.f; st[(int64_t)RS_ST_CPL_FLAGS * np + p] = 0.f;
    st[(int64_t)RS_ST_CPL_TABOVE * np + p] = -9999.0f; st[(int64_t)RS_ST_CPL_TBELOW * np + p] = -9999.0f;
    st[(int64_t)RS_ST_CPL_RADCOEFF * np + p] = 1.0f;
    st[(int64_t)RS_ST_CPL_RCABOVE * np + p] = -9999.0f; st[(int64_t)RS_ST_CPL_RCBELOW * np + p] = -9999.0f;
    st[(int64_t)RS_ST_CPL_RCPREV * np + p] = 1.0f;
    st[(int64_t)RS_ST_CPL_SWCOF * np + p] = 1.0f; st[(int64_t)RS_ST_CPL_LWCOF * np + p] = 1.0f;
    st[(int64_t)RS_ST_CPL_SWCORR * np + p] = 0.0f; st[(int64_t)RS_ST_CPL_LWCORR * np + p] = 0.0f;
    st[(int64_t)RS_ST_CPL_TEND1 * np + p] = 0.f;
    st[(int64_t)RS_ST_CPL_LASTOBS * np + p] = a.pp.coupling_tsurf ? (float)a.pp.coupling_tsurf[p] : -9999.0f;
    st[(int64_t)RS_ST_CPL_RESUME * np + p] = 1.0f;
  }
}

/* The forcing between two hourly knots in single precision: v0 + w (v1 - v0) with the knots' difference taken
 * in fp64 and rounded once, w = r / spk as ONE fp32 product, the sum as one fused multiply-add (the reference
 * driver's linear rule, examples/example1/src/JsonSource.cpp:115-172, in this flavour's arithmetic).  At a knot
 * (r = 0) the value is the knot's, rounded. */
__device__ __forceinline__ float rs32_lerp_weight(int32_t r, double r_spk) { return (float)r * (float)r_spk; }
__device__ __forceinline__ float rs32_lerp(float v0, float dv, float w) { return __builtin_fmaf(w, dv, v0); }

/* fp32 twin of expand_kernel (rs_kernels.hip: one basic block per time index, stores with a scalar
 * row base): same knots (fp64), interpolated in single precision (rs32_lerp). */
__global__ void __launch_bounds__(kBlock) expand_kernel_f32(const rs::ExpandArgs a) {
  const bool OBS = a.f.tsurfobs != nullptr, TDEW = a.f.tdew != nullptr; /* (uniform) */
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= a.npoints) return;
  const int32_t k = a.kfirst + (int32_t)blockIdx.y;
  int32_t tlo = k * a.spk, thi = tlo + a.spk;
  if (tlo < a.t0 - 1) tlo = a.t0 - 1;
  if (thi > a.t0 - 1 + a.nsteps) thi = a.t0 - 1 + a.nsteps;
  if (tlo >= thi) return;
  const int64_t kcol = a.gather ? (int64_t)a.gather[p] : p;
  const double *ka = a.knots + ((int64_t)(k - a.k0) * RS_KNOT_FIELDS) * a.np_pad + kcol;
  const double *kb = ka + (int64_t)RS_KNOT_FIELDS * a.np_pad;
  const bool need_b = (thi - 1) > k * a.spk;
  double v0[7], dv[7];
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    v0[q] = ka[(int64_t)q * a.np_pad];
    const double v1 = need_b ? kb[(int64_t)q * a.np_pad] : v0[q];
    dv[q] = v1 - v0[q];
  }
  const double ts0 = ka[7 * a.np_pad];
  const int32_t ph0 = (int32_t)ka[8 * a.np_pad];
  const int32_t ph1 = need_b ? (int32_t)kb[8 * a.np_pad] : ph0;
  const int64_t col0 = (int64_t)blockIdx.x * kBlock;
  for (int32_t t = tlo; t < thi; ++t) {
    const int32_t r = t - k * a.spk;
    const int64_t row = (int64_t)(t - (a.t0 - 1)) * a.f.t_stride + col0;
    uint32_t b4; /* the lane's byte offset, formed in this block (rs_kernels.hip, LaneOff) */
    asm volatile("v_lshlrev_b32 %0, 2, %1" : "=v"(b4) : "v"(threadIdx.x));
    auto st = [&](const void *base, float v) { *(float *)((char *)((float *)base + row) + b4) = v; };
    /* rs32_lerp: the arithmetic step_kernel_f32duo's ground wave uses when it reads the knots itself - a window and the
     * knot-reading launch hand the model the same bits */
    float v[7];
    const float w = rs32_lerp_weight(r, a.r_spk);
#pragma unroll
    for (int q = 0; q < 7; ++q) v[q] = rs32_lerp((float)v0[q], (float)dv[q], w);
    st(a.f.tair, v[0]);
    if (TDEW) st(a.f.tdew, v[1]);
    st(a.f.vz, v[2]);
    st(a.f.rhz, v[3]);
    st(a.f.prec, v[4]);
    st(a.f.sw, v[5]);
    st(a.f.lw, v[6]);
    if (OBS) st(a.f.tsurfobs, (t == 0) ? (float)ts0 : -9999.9f);
    *(int32_t *)((char *)((int32_t *)a.f.precphase + row) + b4) = (r == 0) ? ph0 : ph1;
  }
  if (p == 0 && !a.f.hour_pstride)
    for (int32_t t = tlo; t < thi; ++t)
      ((int32_t *)a.f.hour)[t - (a.t0 - 1)] = rs_sy_hour(t + 1, a.spk, a.start_hour);
}


/* ==== two points per lane (NLayers = 15) ============================================================ */
typedef float f2 __attribute__((ext_vector_type(2)));
typedef int32_t i2 __attribute__((ext_vector_type(2)));
struct b2 {
  bool x, y;
};
__device__ __forceinline__ f2 S2(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 rcp2(f2 a) { return f2{__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)}; }
__device__ __forceinline__ f2 sqrt2(f2 a) { return f2{__builtin_amdgcn_sqrtf(a.x), __builtin_amdgcn_sqrtf(a.y)}; }
__device__ __forceinline__ f2 exp2v(f2 a) { return f2{__expf(a.x), __expf(a.y)}; }
__device__ __forceinline__ f2 log2v(f2 a) { return f2{__logf(a.x), __logf(a.y)}; }
__device__ __forceinline__ f2 min2(f2 a, f2 b) { return f2{__builtin_fminf(a.x, b.x), __builtin_fminf(a.y, b.y)}; }
__device__ __forceinline__ f2 max2(f2 a, f2 b) { return f2{__builtin_fmaxf(a.x, b.x), __builtin_fmaxf(a.y, b.y)}; }
__device__ __forceinline__ f2 abs2(f2 a) { return f2{__builtin_fabsf(a.x), __builtin_fabsf(a.y)}; }
__device__ __forceinline__ b2 gt2(f2 a, f2 b) { return b2{a.x > b.x, a.y > b.y}; }
__device__ __forceinline__ b2 ge2(f2 a, f2 b) { return b2{a.x >= b.x, a.y >= b.y}; }
__device__ __forceinline__ b2 lt2(f2 a, f2 b) { return b2{a.x < b.x, a.y < b.y}; }
__device__ __forceinline__ b2 le2(f2 a, f2 b) { return b2{a.x <= b.x, a.y <= b.y}; }
__device__ __forceinline__ b2 and2(b2 a, b2 b) { return b2{a.x && b.x, a.y && b.y}; }
__device__ __forceinline__ b2 or2(b2 a, b2 b) { return b2{a.x || b.x, a.y || b.y}; }
__device__ __forceinline__ b2 not2(b2 a) { return b2{!a.x, !a.y}; }
__device__ __forceinline__ f2 sel2(b2 m, f2 a, f2 b) { return f2{m.x ? a.x : b.x, m.y ? a.y : b.y}; }
/* over the wavefront's active lanes, both points of each */
/* (one ballot per point of the pair, OR-ed on the scalar unit: a compare's mask is a ballot already; `m.x || m.y`
 * as one ballot makes the compiler materialise the OR in a vector register first) */
__device__ __forceinline__ bool wave_any2(b2 m) {
  return (__builtin_amdgcn_ballot_w64(m.x) | __builtin_amdgcn_ballot_w64(m.y)) != 0ull;
}
__device__ __forceinline__ bool wave_all2(b2 m) {
  return (__builtin_amdgcn_ballot_w64(!m.x) | __builtin_amdgcn_ballot_w64(!m.y)) == 0ull;
}

/* a select whose condition is a wavefront mask held on the scalar unit (a loop-carried per-point flag kept as a
 * lane-bool costs a v_cndmask + v_cmp_ne to turn it back into a mask at every vote) */
__device__ __forceinline__ float selm(uint64_t mask, float a, float b) {
  float r;
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
  return r;
}
__device__ __forceinline__ int32_t addm(uint64_t mask, int32_t t) { /* t + 1 in the lanes of the mask */
  int32_t r;
  asm("v_addc_co_u32 %0, vcc, 0, %1, %2" : "=v"(r) : "v"(t), "s"(mask) : "vcc");
  return r;
}

enum { X2_WINDOW = 0, X2_KNOTS = 1 };
constexpr float kCHF = 920.0f * 2100.0f; /* density x specific heat of ice: the frozen branch of CalcHCapHCond */

/* capDZ of a layer from the heat capacity of its water (src/BalanceModel.f90:239,146: VSH = dryCap + WCont CHWT,
 * capDZ = -1 / (DyC VSH)) with DyC multiplied in beforehand: -1 / (A chwt + B), A = DyC WCont, B = DyC dryCap.
 * ONE definition for the lanes and for the frozen-layer constants (prepare_constants_f32). */
__device__ __forceinline__ float x2_vsh(float wcont, float drycap, float chwt) { return __builtin_fmaf(wcont, chwt, drycap); }
__device__ __forceinline__ float x2_rcap(float A, float B, float chwt) { return __builtin_amdgcn_rcpf(__builtin_fmaf(A, chwt, B)); }
__device__ __forceinline__ float x2_hs1(float vsh, float hsfac1, float r_twodt) { return (vsh * hsfac1) * r_twodt; }

/* fills the device-made members of RsConstantsF, by the lanes' own expressions */
__global__ void prepare_constants_f32(RsConstantsF *c) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float r_twodt = __builtin_amdgcn_rcpf(c->twoDT);
  c->r_twoDT = r_twodt;
  for (int j = 1; j <= c->NLayers; ++j) {
    const float A = c->DyC[j] * c->WCont[j], B = c->DyC[j] * c->dryCap[j];
    c->lk4[j][0] = A;
    c->lk4[j][1] = B;
    c->lk4[j][2] = c->condDZ[j];
    c->lk4[j][3] = x2_rcap(A, B, kCHF); /* without capDZ's sign: the update subtracts */
    c->capDZF[j] = -c->lk4[j][3];
    if (j == 1) c->hs1F = x2_hs1(x2_vsh(c->WCont[1], c->dryCap[1], kCHF), c->HSfac1, r_twodt);
  }
}

struct X2State { /* SURVEY.md Appendix B, two points */
  f2 tsurf, wat, snow, ice, ice2, dep, q2melt, t4melt, albedo;
  b2 verycold, failed;
};

/* one point of the pair as the one-point physics source sees it, and back */
__device__ __forceinline__ Scalars x2_scalars(const X2State &s, int comp, float t1, float t2) {
  Scalars q;
  q.tnw1 = t1; q.tnw2 = t2;
  q.tsurf = comp ? s.tsurf.y : s.tsurf.x; q.wat = comp ? s.wat.y : s.wat.x; q.snow = comp ? s.snow.y : s.snow.x;
  q.ice = comp ? s.ice.y : s.ice.x; q.ice2 = comp ? s.ice2.y : s.ice2.x; q.dep = comp ? s.dep.y : s.dep.x;
  q.q2melt = comp ? s.q2melt.y : s.q2melt.x; q.t4melt = comp ? s.t4melt.y : s.t4melt.x;
  q.albedo = comp ? s.albedo.y : s.albedo.x;
  q.tair_end = q.vz_end = q.rh_end = 0.f;
  q.verycold = comp ? s.verycold.y : s.verycold.x;
  q.failed = false;
  return q;
}

/* ---- the storages for a point PAIR ---------------------------------------------------------------------
 * melting (src/Storage.f90:319-402) and RoadCond with the four storages, NewMeltFreezeHeat and CalcAlbedo
 * (src/Cond.f90:9-139, src/Storage.f90:33-314,409-432) for the lane's two points at once: the statements of the
 * one-point source (rs_physics_body.inc melting / road_condition) in their order, operation for operation - a
 * point's bits are what the one-point functions give it (A/B: -DRS_X2_ROAD_SCALAR) - with every `if (cond) x = e`
 * as a select.  Compares and selects do not pack, but the two points' streams are independent of each other, and
 * in ONE basic block the SIMD issues an instruction of the one behind an instruction of the other (2.35 cycles
 * each instead of 4.5, profiles/r06_f32_issue_rates.txt); run one point after the other through the one-point
 * source, each behind its own wavefront-uniform branches, they never met.  The wavefront-uniform shortcuts of the
 * one-point source stay, over both points of every lane. */
__device__ __forceinline__ b2 is_pz2(f2 a) { return b2{rs_is_pos_zero(a.x), rs_is_pos_zero(a.y)}; }
__device__ __forceinline__ f2 div2(f2 a, f2 b) { return a * rcp2(b); } /* rs_div(float, float), rs_math.hpp */

__device__ __forceinline__ void x2_melting(X2State &s, f2 &T1, f2 &T2, f2 hstor, f2 hs1) {
  const b2 cover = or2(or2(gt2(s.snow, S2(0.f)), gt2(s.ice, S2(0.f))), gt2(s.ice2, S2(0.f)));
  const b2 A = or2(or2(le2(hstor, S2(0.00001f)), le2(s.tsurf, s.t4melt)), le2(s.q2melt, S2(0.f)));
  const f2 QAvail = hs1 * (T1 - s.t4melt);
  const b2 cold = lt2(s.tsurf, S2(0.5f));
  const b2 r1 = and2(and2(cover, A), cold);                                      /* Q2Melt = 0, the profile stays */
  const b2 r2 = and2(and2(and2(cover, A), not2(cold)), gt2(s.tsurf, S2(2.0f)));  /* Q2Melt = min(Q2Melt, QAvail) */
  const b2 mainb = and2(cover, not2(or2(r1, r2)));
  const b2 all_in = and2(mainb, ge2(s.q2melt, QAvail));                          /* everything available goes into melting */
  const b2 part = and2(mainb, not2(ge2(s.q2melt, QAvail)));
  const f2 t4p = s.t4melt + S2(0.01f);
  const f2 t1part = s.t4melt + div2(QAvail - s.q2melt, hs1);
  f2 q = s.q2melt;
  q = sel2(and2(r2, lt2(QAvail, s.q2melt)), QAvail, q);
  q = sel2(all_in, QAvail, q);
  q = sel2(or2(r1, not2(cover)), S2(0.f), q);
  s.q2melt = q;
  T1 = sel2(all_in, t4p, sel2(part, t1part, T1));
  T2 = sel2(mainb, t4p, T2);
}

__device__ __forceinline__ void x2_road_condition(const ConstsAS &c, X2State &s, f2 evap) {
  const f2 Z = S2(0.f);
  auto pos = [&](f2 a) { return gt2(a, Z); };
  auto nonpos = [&](f2 a) { return le2(a, Z); };
  {
    const b2 bare = and2(and2(and2(is_pz2(s.wat), is_pz2(s.snow)), and2(is_pz2(s.ice), is_pz2(s.ice2))),
                         and2(is_pz2(s.dep), b2{evap.x == 0.f, evap.y == 0.f}));
    if (RS_BARE_FAST(c) && wave_all2(bare)) {
      s.verycold = and2(s.verycold, not2(gt2(s.tsurf, S2(c.TLimColdH))));
      s.verycold = or2(s.verycold, lt2(s.tsurf, S2(c.TLimColdL)));
      s.q2melt = Z;
      s.albedo = S2(c.AlbDry);
      return;
    }
  }
  const bool snow_here = !(RS_BARE_FAST(c) && wave_all2(is_pz2(s.snow)));
  /* WearFactors */
  auto floor_at = [&](f2 v, float lo) { return sel2(gt2(v, S2(lo)), v, S2(lo)); };
  f2 SnowTran = Z;
  if (snow_here) {
    SnowTran = floor_at(S2(c.wSnowTran) * s.snow, 0.01f);
    SnowTran = sel2(lt2(s.snow, S2(0.2f)), SnowTran * S2(3.f), SnowTran);
    SnowTran = SnowTran * S2(c.Tph);
  }
  const f2 IceWear = floor_at(S2(c.wIce) * s.ice, 0.01f) * S2(c.Tph);
  const f2 IceWear2 = floor_at(S2(c.wIce2) * s.ice2, 0.01f) * S2(c.Tph);
  const f2 DepWear = floor_at(S2(c.wDep) * s.dep, 0.01f) * S2(c.Tph);
  f2 WatWear = (S2(10.f) * floor_at(S2(c.wWat) * s.wat, 0.06f)) * S2(c.Tph);
  /* RoadCond head: the VeryCold hysteresis (src/Cond.f90:34-39) */
  s.verycold = and2(s.verycold, not2(gt2(s.tsurf, S2(c.TLimColdH))));
  s.verycold = or2(s.verycold, lt2(s.tsurf, S2(c.TLimColdL)));
  /* WaterStorage, src/Storage.f90:33-84 */
  {
    const b2 evaporates = and2(and2(nonpos(s.snow), nonpos(s.ice)), and2(nonpos(s.dep), gt2(s.tsurf, S2(c.TLimDew))));
    const f2 w1 = sel2(gt2(s.wat, S2(c.MaxPormms)), s.wat - evap, s.wat - S2(c.PorEvaF) * evap);
    s.wat = sel2(evaporates, w1, s.wat);
    const b2 wp = pos(s.wat);
    WatWear = sel2(and2(wp, lt2(s.wat, S2(c.WWearLim))), Z, WatWear);
    const f2 w2 = sel2(gt2(s.wat, S2(c.WWetLim)), s.wat - WatWear, s.wat - S2(c.DampWearF) * WatWear);
    s.wat = sel2(wp, w2, s.wat);
    s.wat = sel2(lt2(s.wat, S2(c.MinWatmms)), Z, s.wat);
    s.wat = sel2(gt2(s.wat, S2(c.MaxWatmms)), S2(c.MaxWatmms), s.wat);
  }
  f2 ext = s.wat - S2(c.MaxPormms);
  ext = sel2(pos(ext), ext, Z);
  /* SnowStorage, src/Storage.f90:88-196 */
  if (snow_here) {
    const f2 RDummy = ext + s.snow;
    const f2 WatSnowRat = sel2(gt2(RDummy, S2(0.001f)), div2(ext, RDummy), Z);
    const b2 sn0 = pos(s.snow);
    const b2 wet = and2(sn0, gt2(WatSnowRat, S2(c.WetSnowFormR)));
    {
      const b2 m = and2(sn0, pos(s.dep));
      s.ice = sel2(m, s.ice + s.dep, s.ice);
      s.dep = sel2(m, Z, s.dep);
      const b2 melts = and2(sn0, and2(pos(s.q2melt), ge2(s.tsurf, S2(c.TLimMeltSnow))));
      const f2 M1000 = S2(1000.f) * div2(s.q2melt * S2(c.DTSecs), S2(RS_MELTDEN));
      s.snow = sel2(melts, s.snow - M1000, s.snow);
      s.wat = sel2(melts, s.wat + M1000, s.wat);
    }
    {
      const b2 m = pos(s.snow);
      const f2 toice = S2(c.wSnow2Ice) * SnowTran;
      s.snow = sel2(m, s.snow - SnowTran, s.snow);
      s.ice = sel2(m, s.ice + toice, s.ice);
      s.ice2 = sel2(m, s.ice2 + toice, s.ice2);
    }
    {
      const b2 sw = and2(pos(s.snow), wet);
      const b2 m1 = and2(sw, gt2(WatSnowRat, S2(c.WetSnowMeltR)));
      s.wat = sel2(m1, s.wat + s.snow, s.wat);
      s.snow = sel2(m1, Z, s.snow);
      const b2 m2 = and2(sw, lt2(s.tsurf, S2(c.TLimFreeze)));
      s.ice = sel2(m2, (s.ice + s.snow) + s.wat, s.ice);
      s.ice2 = sel2(m2, (s.ice2 + s.snow) + s.wat, s.ice2);
      s.snow = sel2(m2, Z, s.snow);
      s.wat = sel2(m2, Z, s.wat);
    }
    s.snow = sel2(lt2(s.snow, S2(c.MinSnowmms)), Z, s.snow);
    s.snow = sel2(gt2(s.snow, S2(c.MaxSnowmms)), s.snow - S2(c.MaxSnowmms * 0.5f), s.snow);
  }
  /* IceStorage, src/Storage.f90:199-267 */
  const b2 freezes = and2(lt2(s.tsurf, S2(c.TLimFreeze)), pos(s.wat));
  const bool ice_here = !(RS_BARE_FAST(c) && wave_all2(and2(and2(is_pz2(s.ice), is_pz2(s.ice2)), not2(freezes))));
  if (ice_here) {
    s.ice = sel2(freezes, s.ice + s.wat, s.ice);
    s.ice2 = sel2(freezes, s.ice2 + s.wat, s.ice2);
    s.wat = sel2(freezes, Z, s.wat);
    const b2 melts = and2(and2(nonpos(s.snow), pos(s.ice)), and2(pos(s.q2melt), ge2(s.tsurf, S2(c.TLimMeltIce))));
    const f2 M1000 = S2(1000.f) * div2(s.q2melt * S2(c.DTSecs), S2(RS_MELTDEN));
    s.ice = sel2(melts, s.ice - M1000, s.ice);
    s.ice2 = sel2(melts, s.ice2 - M1000, s.ice2);
    s.wat = sel2(melts, s.wat + M1000, s.wat);
    s.ice = sel2(pos(s.ice), s.ice - IceWear, s.ice);
    s.ice2 = sel2(pos(s.ice2), s.ice2 - IceWear2, s.ice2);
    s.ice = sel2(lt2(s.ice, S2(c.MinIcemms)), Z, s.ice);
    s.ice = sel2(gt2(s.ice, S2(c.MaxIcemms)), S2(c.MaxIcemms), s.ice);
    s.ice2 = sel2(lt2(s.ice2, S2(c.MinIcemms)), Z, s.ice2);
    s.ice2 = sel2(gt2(s.ice2, S2(c.MaxIcemms)), S2(c.MaxIcemms), s.ice2);
  }
  /* DepositStorage, src/Storage.f90:271-314 */
  s.dep = sel2(lt2(evap, Z), s.dep - evap, s.dep);
  {
    const b2 m = gt2(s.tsurf, S2(c.TLimMeltDep));
    s.wat = sel2(m, s.wat + s.dep, s.wat);
    s.dep = sel2(m, Z, s.dep);
  }
  s.dep = sel2(and2(nonpos(s.snow), pos(s.dep)), s.dep - DepWear, s.dep);
  s.dep = sel2(lt2(s.dep, S2(c.MinDepmms)), Z, s.dep);
  {
    const b2 m = gt2(s.dep, S2(c.MaxDepmms));
    s.wat = sel2(m, s.wat + (s.dep - S2(c.MaxDepmms)), s.wat);
    s.dep = sel2(m, S2(c.MaxDepmms), s.dep);
  }
  /* RoadCond tail, src/Cond.f90:61-62 */
  s.wat = sel2(lt2(s.wat, S2(c.MinWatmms)), Z, s.wat);
  s.wat = sel2(gt2(s.wat, S2(c.MaxWatmms)), S2(c.MaxWatmms), s.wat);
  const float IceMax = 1.5f;
  const f2 span = S2(c.AlbSnow - c.AlbDry);
  if (!snow_here && !ice_here) { /* no melt heat, the deposit alone in the albedo (rs_physics_body.inc) */
    s.q2melt = Z;
    const f2 mid = S2(c.AlbDry) + div2(s.dep, S2(IceMax)) * span;
    s.albedo = sel2(gt2(s.dep, S2(0.01f)), sel2(lt2(s.dep, S2(IceMax)), mid, S2(c.AlbSnow)), S2(c.AlbDry));
    return;
  }
  /* NewMeltFreezeHeat, src/Storage.f90:409-432 */
  {
    const b2 sn = pos(s.snow), ic = and2(nonpos(s.snow), pos(s.ice));
    auto heat = [&](f2 mm) { return div2(S2(RS_MELTDEN) * div2(mm, S2(1000.f)), S2(c.DTSecs)); };
    f2 q = sel2(sn, heat(s.snow), Z);
    s.t4melt = sel2(sn, S2(c.TLimMeltSnow), s.t4melt);
    q = sel2(ic, heat(s.ice), q);
    s.t4melt = sel2(ic, S2(c.TLimMeltIce), s.t4melt);
    s.q2melt = sel2(lt2(q, Z), Z, q);
  }
  /* CalcAlbedo, src/Cond.f90:105-139 */
  {
    f2 IceSum = S2(0.5f) * (s.ice + s.ice2) + s.dep;
    IceSum = sel2(lt2(IceSum, Z), Z, IceSum);
    const b2 snowy = and2(gt2(s.snow, S2(0.01f)), gt2(s.snow, s.ice));
    const b2 icy = or2(gt2(s.ice, S2(0.01f)), gt2(s.dep, S2(0.01f)));
    const f2 mid = S2(c.AlbDry) + div2(IceSum, S2(IceMax)) * span;
    const f2 a_icy = sel2(lt2(IceSum, S2(IceMax)), mid, S2(c.AlbSnow));
    s.albedo = sel2(snowy, S2(c.AlbSnow), sel2(icy, a_icy, S2(c.AlbDry)));
  }
}

/* The three literals of a layer update that cannot ride on an instruction as its one scalar operand, kept in vector
 * registers for a whole time step (left to itself the compiler re-materialises them in every layer) */
struct X2Lit {
  float a1, c3, chf;
  __device__ __forceinline__ void make() {
    a1 = 0.0079f;
    c3 = -0.0017169f;
    chf = kCHF;
    asm volatile("" : "+v"(a1), "+v"(c3), "+v"(chf));
  }
};

/* 1 / (DyC VSH) of layer j for the lane's two points from their (stale) temperatures tj - capDZ without its sign
 * (CalcHCapHCond + calcCapDZCondDZ, src/BalanceModel.f90:189-251,132-155; the water polynomials in Horner form);
 * hs1: where to leave HS(1) (layer 1 only).  A wavefront all of whose 128 points have the layer frozen takes the
 * plan's constant - the same bits, by construction (prepare_constants_f32). */
__device__ __forceinline__ f2 x2_layer_rcap(const ConstsAS &c, const X2Lit &L, float kA, float kB, float kF, f2 tj, f2 *hs1) {
  const b2 thawed = ge2(tj, S2(0.f));
#ifndef RS_ABL_NOFROZEN
  if (!wave_any2(thawed)) {
    if (hs1) *hs1 = S2(c.hs1F);
    return S2(kF);
  }
#endif
  const f2 roow = fma2(fma2(tj, S2(-0.0050f), S2(L.a1)), tj, S2(1000.0028f));
  f2 cw = fma2(tj, S2(0.0000102f), S2(L.c3));
  cw = fma2(cw, tj, S2(0.11516f));
  cw = fma2(cw, tj, S2(-3.4739f));
  cw = fma2(cw, tj, S2(4217.2f));
  const f2 chwt = sel2(thawed, roow * cw, S2(L.chf));
  /* x2_vsh / x2_hs1 / x2_rcap, packed: the operations prepare_constants_f32 runs for a frozen layer */
  if (hs1) *hs1 = (fma2(S2(c.WCont[1]), chwt, S2(c.dryCap[1])) * S2(c.HSfac1)) * S2(c.r_twoDT);
  return rcp2(fma2(S2(kA), chwt, S2(kB)));
}
__device__ __forceinline__ f2 x2_layer_rcap(const ConstsAS &c, const X2Lit &L, int j, f2 tj, f2 *hs1) {
  return x2_layer_rcap(c, L, c.lk4[j][0], c.lk4[j][1], c.lk4[j][3], tj, hs1);
}

/* A layer's four constants (RsConstantsF::lk4 row) by ONE scalar load issued a layer AHEAD: the compiler loads each
 * member where it is first used, behind the frozen / thawed branch, and waits for it on the spot - thirteen layers,
 * three exposed scalar-cache round trips each.  (The wait takes the value as an operand, so nothing that reads it
 * can move above it.) */
typedef float f4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4s x2_sload4(const ConstsAS *c, uint32_t byte_off) {
  f4s r;
  asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(r) : "s"(c), "s"(byte_off));
  return r;
}
__device__ __forceinline__ void x2_swait(f4s &r) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r)); }

/* ==== two points per lane, two wavefronts per 128 points (step_kernel_f32duo) ==========================
 * One wavefront that keeps a point's whole state, the knots of its forcing and every temporary of a time step in
 * its registers needs 168-250 of them: two or three wavefronts per SIMD, each of which offers the vector unit work
 * only about half of the time it is resident (scalar loads, branches, waits) - measured 4.1e10 point-timesteps/s,
 * what one point per lane gave (the kernel is in the history, commit "fp32 flavour: two points per
 * lane").  As in the fp64 flavour (rs_kernels.hip step_kernel_duo) a workgroup is therefore TWO wavefronts that
 * share 128 points and meet once per time index:
 *   ground wave:  layers 3..15 of the explicit update (they read only OLD neighbours) and everything a step
 *                 needs of its forcing alone, one index AHEAD - the interpolation from the knots (or the window
 *                 row), SetCurrentValues' VZ(1) floor, CheckValues' forcing tests, PrecipitationToStorage's
 *                 amounts, SetDayDependendVariables, the air properties and loop invariants of CalcBLCondAndLE,
 *                 the vapour pressure of the air: eleven floats per point through the LDS mailbox;
 *   surface wave: the chain that starts at the surface state - boundary-layer loop, CalcLE, CalcRNet, layers
 *                 1-2, melting, the storages, the outputs.  It never touches the forcing.
 * Half the registers per wavefront, twice the wavefronts for the same points, two different instruction
 * streams per SIMD. */
#define RS_X2D_NPREP 12
enum { XP_TAIR = 0, XP_C1, XP_K3, XP_RRA, XP_AVCAP, XP_PSYCH, XP_EAIR, XP_SW, XP_ELW, XP_RAIN, XP_SNOW,
       XP_OBS /* FULL: the observation SetCurrentValues forces on Tmp(1:2) at the index, or missing */ };
struct X2Mail {
  float v[2][2][128];              /* [buffer][0: Tmp(2) from the surface wave, 1: Tmp(3) from the ground wave][point] */
  float prep[2][RS_X2D_NPREP][128]; /* [buffer = index parity][value][point] */
  uint32_t flags[2][128];          /* bit 0: CheckValues' verdict on the forcing; bit 1: night (SetDayDependendVariables);
                                      bit 2 (SKY): SunPosition would `stop` (rs_skyview.hpp) */
};
#ifndef RS_X2D_WAVES
#define RS_X2D_WAVES 5 /* wavefronts per SIMD: 96 registers; the mailbox (15.4 KB per workgroup) allows ten workgroups per CU */
#endif
__device__ __forceinline__ void x2_meet() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ f2 lds_ld2(const float *row, uint32_t lane) { return *reinterpret_cast<const f2 *>(row + 2 * lane); }
__device__ __forceinline__ void lds_st2(float *row, uint32_t lane, f2 v) { *reinterpret_cast<f2 *>(row + 2 * lane) = v; }

/* FULL (round 6): the FULL feature set as the two-wavefront fp64 flavour has it (rs_kernels.hip duo_ground / duo_surface):
 * CheckValues' dew-point test, the observation SetCurrentValues forces on Tmp(1:2) during the initialization phase
 * (src/InputOutput.f90:116-148; force_tsurf: always), RelaxationOperations behind it (src/Relaxation.f90:10-47).  A
 * launch with an output depth or coupling goes to the general kernel (step_kernel_f32_coupled).
 * SKY (with FULL, from a forcing window that carries SW_dir and LW_net): sky view and local horizons
 * (examples/example1/src/Simulation.f90:154-156, src/ModRadiation.f90:7-73, src/SunPosition.f90:123-193) on the ground
 * wave, which owns the forcing: CheckValues' sky-view tests and the SW_dir clamp (src/InputOutput.f90:68-77), then
 * ModRadiationBySurroundings on the short and long wave the surface wave receives.  The sun's position and what is
 * decided from it - elevation > 0, which degree of azimuth, horizon above the sun or not - stay in fp64, the fp64
 * flavours' own function on the same table of the host (rs_sun_table): an fp32 azimuth would pick the neighbouring
 * degree of the horizon now and then, a difference of another kind than rounding. */
template <int SRC, bool FULL, bool SKY = false>
__device__ __forceinline__ void x2d_ground(X2Mail &mail, const rs::StepArgs &a) {
  static_assert(!SKY || (FULL && SRC == X2_WINDOW), "sky view: the FULL feature set, from a forcing window");
  KernArgs ka = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
  const uint32_t lane = threadIdx.x & 63u;
  const int64_t p = 2 * ((int64_t)blockIdx.x * 64 + lane); /* < np_pad: every array below has np_pad columns */
  const bool liveX = p < a.npoints, liveY = p + 1 < a.npoints;
  const int64_t np = a.np_pad;
  const float *st = reinterpret_cast<const float *>(a.state);
  f2 T[13]; /* Tmp(3..15) */
#pragma unroll
  for (int j = 3; j <= 15; ++j) T[j - 3] = *reinterpret_cast<const f2 *>(st + (int64_t)(RS_ST_TMP0 + j - 1) * np + p);
  const f2 tbot = f2{(float)ka->pp.tbottom[p], (float)ka->pp.tbottom[p + 1]};
  const int32_t nsteps = ka->nsteps, t0 = ka->t0;
  rs::MathTab mt{nullptr, nullptr, nullptr};
  constexpr int NK = FULL ? 7 : 6; /* tair, vz, rhz, prec, sw, lw [, tdew] */
  f2 kv0[NK], kdv[NK];
  i2 kph0 = i2{0, 0}, kph1 = i2{0, 0};
  int32_t kcur = -1;
  bool knots_safe = false; /* uniform: no index of the current knot interval can fail CheckValues' forcing tests */
  /* FULL: as the fp64 flavours set a point up (time_loop, duo_ground) */
  i2 initlen = i2{0, 0};
  b2 relax = b2{false, false};
  f2 relax_dt = S2(0.f), relax_dv = S2(0.f), relax_dr = S2(0.f), tairR = S2(0.f), vzR = S2(0.f), rhR = S2(0.f);
  f2 obs_next = S2(-9999.9f); /* the observation forced at the index the mailbox is being filled for */
  if (FULL) {
    if (ka->pp.initlen) initlen = i2{ka->pp.initlen[p], ka->pp.initlen[p + 1]};
    if (consts_of(ka).use_relaxation && ka->pp.tair_relax) {
      /* the targets through REAL(4), src/InputOutput.f90:19-26 */
      tairR = f2{(float)ka->pp.tair_relax[p], (float)ka->pp.tair_relax[p + 1]};
      vzR = f2{(float)ka->pp.vz_relax[p], (float)ka->pp.vz_relax[p + 1]};
      rhR = f2{(float)ka->pp.rh_relax[p], (float)ka->pp.rh_relax[p + 1]};
      auto valid = [](float t, float v, float r) {
        return !(t < -100.0f || t > 100.0f || v < 0.0f || v > 100.0f || r < 0.0f || r > 110.f);
      };
      relax = b2{valid(tairR.x, vzR.x, rhR.x), valid(tairR.y, vzR.y, rhR.y)};
      relax_dt = tairR - *reinterpret_cast<const f2 *>(st + (int64_t)RS_ST_TAIR_END * np + p);
      relax_dv = vzR - *reinterpret_cast<const f2 *>(st + (int64_t)RS_ST_VZ_END * np + p);
      relax_dr = rhR - *reinterpret_cast<const f2 *>(st + (int64_t)RS_ST_RH_END * np + p);
    }
  }
  /* SKY: as the fp64 flavours set a point up (rs_kernels.hip time_loop<SKY>) */
  double skyv[2] = {1.0, 1.0}, sinlat[2] = {0, 0}, coslat[2] = {0, 0}, lonrad[2] = {0, 0}, coslon[2] = {1.0, 1.0}, sinlon[2] = {0, 0};
  bool sky_on[2] = {false, false};
  int64_t hcol[2] = {p, p + 1};
  if (SKY) {
#pragma unroll
    for (int comp = 0; comp < 2; ++comp) {
      if (!(comp ? liveY : liveX)) continue;
      const int64_t q = p + comp;
      skyv[comp] = ka->pp.sky_view[q];
      sky_on[comp] = skyv[comp] < (double)1.0f && skyv[comp] > (double)-0.01f;
      if (sky_on[comp]) {
        sinlat[comp] = ka->pp.sin_lat[q];
        coslat[comp] = ka->pp.cos_lat[q];
        lonrad[comp] = ka->pp.lon_rad[q];
        coslon[comp] = ::cos(lonrad[comp]);
        sinlon[comp] = ::sin(lonrad[comp]);
      }
      if (ka->pp.horizon_index) hcol[comp] = ka->pp.horizon_index[q];
    }
  }
  int64_t kcolx = p, kcoly = p + 1;
  if (SRC == X2_KNOTS && ka->knot_gather) {
    kcolx = ka->knot_gather[p];
    kcoly = ka->knot_gather[p + 1];
  }
  const bool fvec = SRC == X2_WINDOW && !(ka->f.t_stride & 1) && liveY;
  /* the forcing's share of index `in` -> mailbox buffer `buf` */
  auto prep = [&](int buf, int32_t in) {
    const ConstsAS &c = consts_of(ka);
    f2 tair, vz, rhz, prec, sw, lw, tdew = S2(0.f), tsobs = S2(-9999.9f);
    i2 phase;
    int32_t hour_u = 0;
    i2 hour_p = i2{0, 0};
    bool hour_per_point = false;
    bool has_tdew = false;
    if (SRC == X2_KNOTS) {
      const int32_t spk = ka->spk;
      const int32_t t = in - 1;
      const int32_t kk = __builtin_amdgcn_readfirstlane(t / spk);
      const int32_t rr = t - kk * spk;
      if (kk != kcur) { /* uniform: a new knot interval - expand_kernel_f32's loads and differences */
        kcur = kk;
        const int fld[7] = {0, 2, 3, 4, 5, 6, 1};
        const double *ka_ = ka->knots + ((int64_t)(kk - ka->knot_k0) * RS_KNOT_FIELDS) * np;
        const bool has_b = (kk + 1 - ka->knot_k0) < ka->knot_n;
        const double *kb_ = ka_ + (int64_t)RS_KNOT_FIELDS * np;
#pragma unroll
        for (int q = 0; q < NK; ++q) {
          const double ax = ka_[(int64_t)fld[q] * np + kcolx], ay = ka_[(int64_t)fld[q] * np + kcoly];
          const double bx = has_b ? kb_[(int64_t)fld[q] * np + kcolx] : ax, by = has_b ? kb_[(int64_t)fld[q] * np + kcoly] : ay;
          kv0[q] = f2{(float)ax, (float)ay};
          kdv[q] = f2{(float)(bx - ax), (float)(by - ay)};
        }
        kph0 = i2{(int32_t)ka_[8 * np + kcolx], (int32_t)ka_[8 * np + kcoly]};
        kph1 = has_b ? i2{(int32_t)kb_[8 * np + kcolx], (int32_t)kb_[8 * np + kcoly]} : kph0;
        /* CheckValues' forcing tests for the whole interval at once: a value between two knots lies between them
         * (to a rounding), so where both ends of every variable keep a margin of 0.01 to its limits (src/InputOutput.f90:
         * 55-66) no index of the interval can fail them - for all 128 points, or the tests run index by index as
         * before (a NaN end compares false: index by index) */
        {
          auto inside = [&](int comp) -> bool {
            auto v0 = [&](int q) { return comp ? kv0[q].y : kv0[q].x; };
            auto v1 = [&](int q) { return comp ? kv0[q].y + kdv[q].y : kv0[q].x + kdv[q].x; };
            const float m = 0.01f;
            bool ok = v0(0) > -90.f + m && v1(0) > -90.f + m && v0(0) < 100.f - m && v1(0) < 100.f - m;   /* tair */
            ok = ok && v0(1) > -1.f + m && v1(1) > -1.f + m && v0(1) < 100.f - m && v1(1) < 100.f - m;     /* vz */
            ok = ok && v0(2) > -0.1f + m && v1(2) > -0.1f + m && v0(2) < 120.f - m && v1(2) < 120.f - m;   /* rhz */
            ok = ok && v0(3) > -0.1f + m && v1(3) > -0.1f + m && v0(3) < 500.f - m && v1(3) < 500.f - m;   /* prec */
            ok = ok && v0(4) > -0.1f + m && v1(4) > -0.1f + m && v0(4) < 4000.f - m && v1(4) < 4000.f - m; /* sw */
            ok = ok && v0(5) > -0.1f + m && v1(5) > -0.1f + m && v0(5) < 1000.f - m && v1(5) < 1000.f - m; /* lw */
            if (FULL) ok = ok && v0(NK - 1) > -90.f + m && v1(NK - 1) > -90.f + m && v0(NK - 1) < 100.f - m && v1(NK - 1) < 100.f - m; /* tdew */
            return ok;
          };
          knots_safe = wave_all2(b2{inside(0), inside(1)});
        }
      }
      const f2 w = S2(rs32_lerp_weight(rr, ka->r_spk));
      tair = fma2(w, kdv[0], kv0[0]); vz = fma2(w, kdv[1], kv0[1]); rhz = fma2(w, kdv[2], kv0[2]);
      prec = fma2(w, kdv[3], kv0[3]); sw = fma2(w, kdv[4], kv0[4]); lw = fma2(w, kdv[5], kv0[5]);
      phase = (rr == 0) ? kph0 : kph1;
      hour_u = rs_sy_hour(in, spk, ka->start_hour);
      if (FULL) { /* expand_kernel_f32: the dew point like the others, the observation at index 1 only */
        tdew = fma2(w, kdv[NK - 1], kv0[NK - 1]);
        has_tdew = (ka->duo_full_ok & 2) != 0;
        if (t == 0) {
          const double *k7 = ka->knots + ((int64_t)(kk - ka->knot_k0) * RS_KNOT_FIELDS + 7) * np;
          tsobs = f2{(float)k7[kcolx], (float)k7[kcoly]};
        }
      }
    } else {
      const int64_t row = (int64_t)(in - t0) * ka->f.t_stride + p;
      auto in2 = [&](const void *base) -> f2 {
        const float *q = reinterpret_cast<const float *>(base) + row;
        if (fvec) return *reinterpret_cast<const f2 *>(q);
        return f2{liveX ? q[0] : 0.f, liveY ? q[1] : 0.f};
      };
      tair = in2(ka->f.tair); vz = in2(ka->f.vz); rhz = in2(ka->f.rhz);
      prec = in2(ka->f.prec); sw = in2(ka->f.sw); lw = in2(ka->f.lw);
      phase = i2{liveX ? (ka->f.precphase + row)[0] : 0, liveY ? (ka->f.precphase + row)[1] : 0};
      hour_per_point = ka->f.hour_pstride != 0;
      if (hour_per_point) hour_p = i2{liveX ? (ka->f.hour + row)[0] : 0, liveY ? (ka->f.hour + row)[1] : 0};
      else hour_u = ka->f.hour[in - t0];
      if (FULL) {
        has_tdew = ka->f.tdew != nullptr;
        if (has_tdew) tdew = in2(ka->f.tdew);
        if (ka->f.tsurfobs) tsobs = in2(ka->f.tsurfobs);
        if (ka->f.tsurfobs && !liveY) tsobs.y = -9999.9f;
      }
    }
    if (in == 1) vz = max2(vz, S2(0.4f)); /* src/Initialization.f90:121-123 */
    uint32_t flx = 0u, fly = 0u;
    if (in < c.SimLen && !(SRC == X2_KNOTS && knots_safe)) { /* CheckValues' forcing tests (src/InputOutput.f90:45-84); the surface temperature's are the surface wave's */
      Forcing fa, fb;
      fa.tair = tair.x; fa.vz = vz.x; fa.rhz = rhz.x; fa.prec = prec.x; fa.sw = sw.x; fa.lw = lw.x;
      fb.tair = tair.y; fb.vz = vz.y; fb.rhz = rhz.y; fb.prec = prec.y; fb.sw = sw.y; fb.lw = lw.y;
      fa.tdew = fb.tdew = 0.f; fa.tsurfobs = fb.tsurfobs = -9999.9f; fa.depth = fb.depth = -9999.9f;
      fa.phase = phase.x; fb.phase = phase.y; fa.hour = fb.hour = 0;
      flx = check_values_forcing(c, fa) ? 1u : 0u;
      fly = check_values_forcing(c, fb) ? 1u : 0u;
      if (FULL && has_tdew) { /* :59-62 */
        flx |= (tdew.x < -90.f || tdew.x > 100.0f) ? 1u : 0u;
        fly |= (tdew.y < -90.f || tdew.y > 100.0f) ? 1u : 0u;
      }
    }
    if (SKY) {
      const int64_t row = (int64_t)(in - t0) * ka->f.t_stride + p;
      auto in2s = [&](const void *base) -> f2 {
        const float *q = reinterpret_cast<const float *>(base) + row;
        return f2{liveX ? q[0] : 0.f, liveY ? q[1] : 0.f};
      };
      f2 sw_dir = in2s(ka->f.sw_dir);
      const f2 lw_net = in2s(ka->f.lw_net);
      if (in < c.SimLen) {
        /* CheckValues' sky-view tests and the clamp of the direct short wave, src/InputOutput.f90:68-77 */
        auto outside = [](float d, float l) { return d < -0.1f || d > 4000.0f || l < -1000.0f || l > 1000.0f; };
        flx |= (sky_on[0] && outside(sw_dir.x, lw_net.x)) ? 1u : 0u;
        fly |= (sky_on[1] && outside(sw_dir.y, lw_net.y)) ? 1u : 0u;
        sw_dir = f2{sw_dir.x > sw.x ? sw.x : sw_dir.x, sw_dir.y > sw.y ? sw.y : sw_dir.y};
      }
      const double *sunrow = ka->f.sun + (int64_t)(in - t0) * RS_SUN_COLS;
#pragma unroll
      for (int comp = 0; comp < 2; ++comp) {
        if (!sky_on[comp]) continue;
        double dsw = comp ? sw.y : sw.x, dsd = comp ? sw_dir.y : sw_dir.x, dlw = comp ? lw.y : lw.x;
        const double dln = comp ? lw_net.y : lw_net.x;
        const double *hz = ka->pp.horizons ? ka->pp.horizons + (ka->pp.horizons_by_point ? hcol[comp] * 360 : hcol[comp]) : nullptr;
        const bool ok = rs::sky_view_radiation(sunrow, sinlat[comp], coslat[comp], lonrad[comp], coslon[comp], sinlon[comp], skyv[comp],
                                               ka->pp.albedo_surroundings, hz, ka->pp.horizons_by_point ? (int64_t)1 : np, dsw, dsd,
                                               dlw, dln);
        if (comp) { sw.y = (float)dsw; lw.y = (float)dlw; fly |= ok ? 0u : 4u; }
        else { sw.x = (float)dsw; lw.x = (float)dlw; flx |= ok ? 0u : 4u; }
      }
    }
    f2 obs = S2(-9999.9f);
    if (FULL && in < c.SimLen) {
      /* SetCurrentValues' observation forcing (src/InputOutput.f90:116-124): decided here, applied by both waves */
      const b2 forced = b2{(in <= initlen.x || c.force_tsurf) && tsobs.x > -100.0f, (in <= initlen.y || c.force_tsurf) && tsobs.y > -100.0f};
      obs = sel2(forced, tsobs, obs);
      if (wave_any2(relax)) { /* RelaxationOperations, src/Relaxation.f90:10-47 */
        const b2 at = and2(relax, b2{in == initlen.x, in == initlen.y});
        if (wave_any2(at)) { /* the anchors: once per point (atm%TairInitEnd ...), the forcing as the index has it */
          relax_dt = sel2(at, tairR - tair, relax_dt);
          relax_dv = sel2(at, vzR - vz, relax_dv);
          relax_dr = sel2(at, rhR - rhz, relax_dr);
          float *sw_ = reinterpret_cast<float *>(ka->state);
          if (at.x && liveX) { sw_[(int64_t)RS_ST_TAIR_END * np + p] = tair.x; sw_[(int64_t)RS_ST_VZ_END * np + p] = vz.x; sw_[(int64_t)RS_ST_RH_END * np + p] = rhz.x; }
          if (at.y && liveY) { sw_[(int64_t)RS_ST_TAIR_END * np + p + 1] = tair.y; sw_[(int64_t)RS_ST_VZ_END * np + p + 1] = vz.y; sw_[(int64_t)RS_ST_RH_END * np + p + 1] = rhz.y; }
        }
        const b2 behind = and2(relax, b2{in > initlen.x, in > initlen.y});
        if (wave_any2(behind)) {
          /* exp(-(DTSecs in - DTSecs InitLenI) / (4 x 3600)) */
          const f2 d = f2{(float)(in - initlen.x), (float)(in - initlen.y)};
          const f2 e = exp2v(S2(-c.DTSecs * (1.0f / 14400.0f)) * d);
          tair = sel2(behind, tair - relax_dt * e, tair);
          vz = sel2(behind, vz - relax_dv * e, vz);
          const f2 rh = min2(rhz - relax_dr * e, S2(100.0f));
          rhz = sel2(behind, rh, rhz);
        }
      }
    }
    if (FULL) obs_next = obs;
    /* PrecipitationToStorage without the storages (src/Storage.f90:9-29, src/Cond.f90:143-249): the amounts */
    const f2 prec_ts = (prec * S2(__builtin_amdgcn_rcpf(3600.0f))) * S2(c.DTSecs);
    f2 rain = S2(0.f), snowfall = S2(0.f);
    if (!(RS_PREC_FAST(c) && wave_all2(b2{prec_ts.x == 0.f, prec_ts.y == 0.f}))) {
      Scalars z;
      float pt = prec_ts.x;
      z.wat = 0.f; z.snow = 0.f;
      precipitation_to_storage(c, mt, z, phase.x, pt, tair.x, rhz.x);
      rain.x = z.wat; snowfall.x = z.snow;
      pt = prec_ts.y;
      z.wat = 0.f; z.snow = 0.f;
      precipitation_to_storage(c, mt, z, phase.y, pt, tair.y, rhz.y);
      rain.y = z.wat; snowfall.y = z.snow;
    }
    /* SetDayDependendVariables (src/BalanceModel.f90:354-387): the calm limit here, the traffic friction from the
     * night bit on the surface wave */
    {
      const float calmN = c.CalmLimNgt, calmD = c.CalmLimDay;
      f2 calm;
      if (hour_per_point) {
        const b2 night = b2{((float)hour_p.x >= c.NightOn) || ((float)hour_p.x <= c.NightOff),
                            ((float)hour_p.y >= c.NightOn) || ((float)hour_p.y <= c.NightOff)};
        calm = sel2(night, S2(calmN), S2(calmD));
        flx |= night.x ? 2u : 0u;
        fly |= night.y ? 2u : 0u;
      } else {
        const bool night = ((float)hour_u >= c.NightOn) || ((float)hour_u <= c.NightOff);
        calm = S2(night ? calmN : calmD);
        flx |= night ? 2u : 0u;
        fly |= night ? 2u : 0u;
      }
      vz = max2(vz, calm);
    }
    /* air properties and the loop's invariants (src/BoundaryLayer.f90:50-62,78-79; the algebra: x2d_surface) */
    const f2 TaK = tair + S2(273.15f);
    const f2 AirDens = S2(100000.0f) * rcp2(S2(287.05f) * TaK);
    const f2 dK = (TaK - S2(250.0f)) * (TaK - S2(250.0f));
    const f2 AirHCap = fma2(dK, S2(1.0f / 3364.0f), S2(1005.0f));
    const f2 AirVCap = AirHCap * AirDens;
    const f2 PsychC = S2(0.1f) * fma2(S2(0.00063f), TaK, S2(0.47496f));
    const f2 vkvz = S2(c.VK_Const) * vz;
    const f2 C1 = (AirVCap * S2(c.VK_Const)) * vkvz;
    const float stab_num = -c.VK_Const * c.ZRefT * c.Grav;
    const f2 K3 = S2(stab_num) * rcp2((AirVCap * TaK) * (vkvz * vkvz * vkvz));
    const f2 rRA = rcp2(S2(c.VK_Const * c.VK_Const) * vz);
    /* the vapour pressure of the air (CalcLE, :160-170) */
    const b2 aneg = lt2(tair, S2(0.f));
    const f2 aa = sel2(aneg, S2(21.875f), S2(17.269f)), ba = sel2(aneg, S2(265.5f), S2(237.3f));
    const f2 ESat = S2(0.61078f) * exp2v((aa * tair) * rcp2(tair + ba));
    f2 hum = S2(0.01f) * rhz;
    hum = min2(hum, S2(1.0f));
    float (*w)[128] = mail.prep[buf];
    lds_st2(w[XP_TAIR], lane, tair); lds_st2(w[XP_C1], lane, C1); lds_st2(w[XP_K3], lane, K3);
    lds_st2(w[XP_RRA], lane, rRA); lds_st2(w[XP_AVCAP], lane, AirVCap); lds_st2(w[XP_PSYCH], lane, PsychC);
    lds_st2(w[XP_EAIR], lane, hum * ESat); lds_st2(w[XP_SW], lane, sw); lds_st2(w[XP_ELW], lane, S2(c.Emiss) * lw);
    lds_st2(w[XP_RAIN], lane, rain); lds_st2(w[XP_SNOW], lane, snowfall);
    if (FULL) lds_st2(w[XP_OBS], lane, obs);
    *reinterpret_cast<uint2 *>(&mail.flags[buf][2 * lane]) = uint2{flx, fly};
  };
  lds_st2(mail.v[0][1], lane, T[0]);
  prep(0, t0);
  x2_meet();
  for (int32_t kv = 0; kv < nsteps; ++kv) {
    asm volatile("" : "+s"(ka));
    const ConstsAS &c = consts_of(ka);
    const int32_t k = __builtin_amdgcn_readfirstlane(kv);
    f2 t2 = lds_ld2(mail.v[k & 1][0], lane); /* Tmp(2) as the last step left it (melting included) */
    if (FULL) { /* SetCurrentValues has forced Tmp(1:2) at this index: the flux into layer 3 sees the forced Tmp(2) */
      const f2 obs_cur = obs_next;
      t2 = sel2(gt2(obs_cur, S2(-100.0f)), obs_cur, t2);
    }
    /* (a point that has failed keeps stepping here: its layers are never read again) */
    f2 Gprev = S2(c.lk4[2][2]) * (T[0] - t2); /* G(2), the expression layer 2 itself evaluates */
    const f2 dts = S2(c.DTSecs);
    X2Lit lit;
    lit.make();
    constexpr uint32_t lk0 = (uint32_t)offsetof(RsConstantsF, lk4);
    f4s kn = x2_sload4(&c, lk0 + 3u * 16u);
#ifdef RS_ABL_NOLAYERS
#pragma unroll
    for (int j = 3; j <= 3; ++j) {
#else
#pragma unroll
    for (int j = 3; j <= 15; ++j) {
#endif
      f4s kc = kn;
      x2_swait(kc);
      if (j < 15) kn = x2_sload4(&c, lk0 + (uint32_t)(j + 1) * 16u);
      const f2 tj = T[j - 3];
      const f2 tnext = (j == 15) ? tbot : T[j - 2];
      const f2 rcap = x2_layer_rcap(c, lit, kc.x, kc.y, kc.w, tj, nullptr);
      const f2 G = S2(kc.z) * (tnext - tj);
      T[j - 3] = fma2(dts, rcap * (Gprev - G), tj);
      Gprev = G;
      /* the update belongs HERE: left alone, the compiler sinks the thirteen updates behind the last layer's branch and
       * keeps thirteen reciprocals alive until then (26 registers) */
      asm volatile("" : "+v"(T[j - 3]), "+v"(Gprev));
    }
    lds_st2(mail.v[(k & 1) ^ 1][1], lane, T[0]);
    if (k + 1 < nsteps) prep((k & 1) ^ 1, t0 + k + 1);
    x2_meet();
  }
  if (liveX) {
    float *sw_ = reinterpret_cast<float *>(a.state);
#pragma unroll
    for (int j = 3; j <= 15; ++j) {
      float *q = sw_ + (int64_t)(RS_ST_TMP0 + j - 1) * np + p;
      if (liveY) *reinterpret_cast<f2 *>(q) = T[j - 3];
      else q[0] = T[j - 3].x;
    }
  }
}

template <bool SCORE, bool FULL, bool SKY = false>
__device__ __forceinline__ void x2d_surface(X2Mail &mail, const rs::StepArgs &a) {
  KernArgs ka = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
  const uint32_t lane = threadIdx.x & 63u;
  const int64_t p = 2 * ((int64_t)blockIdx.x * 64 + lane);
  const bool liveX = p < a.npoints, liveY = p + 1 < a.npoints;
  const int64_t np = a.np_pad;
  float *st = reinterpret_cast<float *>(a.state);
  auto ldst = [&](int slot) -> f2 { return *reinterpret_cast<const f2 *>(st + (int64_t)slot * np + p); };
  auto stst = [&](int slot, f2 v) {
    float *q = st + (int64_t)slot * np + p;
    if (liveY) *reinterpret_cast<f2 *>(q) = v;
    else if (liveX) q[0] = v.x;
  };
  f2 T1 = ldst(RS_ST_TMP0), T2 = ldst(RS_ST_TMP0 + 1);
  X2State s;
  s.tsurf = ldst(RS_ST_TSURF);
  s.wat = ldst(RS_ST_WAT); s.snow = ldst(RS_ST_SNOW); s.ice = ldst(RS_ST_ICE); s.ice2 = ldst(RS_ST_ICE2);
  s.dep = ldst(RS_ST_DEP); s.q2melt = ldst(RS_ST_Q2MELT); s.t4melt = ldst(RS_ST_T4MELT);
  s.albedo = ldst(RS_ST_ALBEDO);
  {
    const f2 vc = ldst(RS_ST_VERYCOLD), fl = ldst(RS_ST_FAILED);
    s.verycold = b2{vc.x != 0.f, vc.y != 0.f};
    s.failed = b2{fl.x != 0.f || !liveX, fl.y != 0.f || !liveY}; /* a point beyond npoints never steps */
  }
  const int32_t nsteps = ka->nsteps, t0 = ka->t0;
  i2 score = i2{0, 0}, regime = i2{0, 0}, last_trips = i2{0, 0};
  const bool ovec = !(ka->o.t_stride & 1) && liveY;
  lds_st2(mail.v[0][0], lane, T2);
  x2_meet();
  for (int32_t kv = 0; kv < nsteps; ++kv) {
    asm volatile("" : "+s"(ka));
    const ConstsAS &c = consts_of(ka);
    const int32_t k = __builtin_amdgcn_readfirstlane(kv);
    const int32_t i = t0 + k;
    int64_t r = (int64_t)(i - 1);
    const int32_t dec = ka->o.decimate;
    bool write = true;
    if (dec > 1) {
      write = (r % dec == 0);
      r /= dec;
    }
    const int64_t orow = (r - ka->o.row0) * ka->o.t_stride + p;
    auto out2 = [&](void *base, f2 v) {
      float *q = reinterpret_cast<float *>(base) + orow;
      if (ovec) {
        *reinterpret_cast<f2 *>(q) = v;
      } else {
        if (liveX) q[0] = v.x;
        if (liveY) q[1] = v.y;
      }
    };
    const b2 was_failed = s.failed;
    const f2 t3 = lds_ld2(mail.v[k & 1][1], lane); /* Tmp(3) as the last step left it */
    const float (*w)[128] = mail.prep[k & 1];
    const uint2 fl = *reinterpret_cast<const uint2 *>(&mail.flags[k & 1][2 * lane]);
    if (!wave_all2(was_failed)) {
      const f2 tair = lds_ld2(w[XP_TAIR], lane);
      if (i < c.SimLen) { /* CheckValues: the forcing's verdict | the surface temperature's ("Abnormal surface temperature") */
        const bool badx = !was_failed.x && ((fl.x & 1u) || check_values_tsurf(c, s.tsurf.x));
        const bool bady = !was_failed.y && ((fl.y & 1u) || check_values_tsurf(c, s.tsurf.y));
        if (badx) { s.failed.x = true; st[(int64_t)RS_ST_FAILED * np + p] = (float)i; }
        if (bady) { s.failed.y = true; st[(int64_t)RS_ST_FAILED * np + p + 1] = (float)i; }
      }
      if (SKY && ((fl.x | fl.y) & 4u)) { /* where the reference would `stop` in SunPosition: flagged failed, at any index */
        if (!was_failed.x && (fl.x & 4u)) { s.failed.x = true; st[(int64_t)RS_ST_FAILED * np + p] = (float)i; }
        if (!was_failed.y && (fl.y & 4u)) { s.failed.y = true; st[(int64_t)RS_ST_FAILED * np + p + 1] = (float)i; }
      }
      /* TmpNw(1:2) as CalcHCapHCond sees them (src/BalanceModel.f90:215): SetCurrentValues forces Tmp, not TmpNw */
      const f2 stale1 = T1, stale2 = T2;
      if (FULL) {
        if (i < c.SimLen) {
          const f2 obs = lds_ld2(w[XP_OBS], lane);
          const b2 forced = gt2(obs, S2(-100.0f));
          if (wave_any2(forced)) {
            T1 = sel2(forced, obs, T1);
            T2 = sel2(forced, obs, T2);
            s.tsurf = sel2(forced, (T1 + T2) * S2(0.5f), s.tsurf);
          }
        } else { /* lastValues (src/InputOutput.f90:169-198): no output depth in this kernel's launches */
          s.tsurf = (T1 + T2) * S2(0.5f);
        }
      }
      s.wat = s.wat + lds_ld2(w[XP_RAIN], lane);
      s.snow = s.snow + lds_ld2(w[XP_SNOW], lane);
      const f2 trffric = f2{(fl.x & 2u) ? c.TrfFricNgt : c.TrFfricDay, (fl.y & 2u) ? c.TrfFricNgt : c.TrFfricDay};
      /* ---- the boundary-layer fixed point (src/BoundaryLayer.f90:64-96) from the handed-over invariants, with one
       * reciprocal per pass: with a = logUstar + PSIM, b = logCond + PSIH
       *   UStar = vkvz / a,  BLCond = avk UStar / b = C1 / (a b),                       C1 = avk vkvz
       *   Stab  = stab_num BLCond dT / (den0 UStar^3) = C2 BLCond a^3,                  C2 = dT stab_num / (den0 vkvz^3)
       * Which points are still in the loop is kept as two wavefront masks on the scalar unit. */
      f2 blcond, le, evap;
      i2 trips = i2{5, 5};
      uint64_t unsx = 0ull, unsy = 0ull; /* (SCORE) some pass of the point took the unstable arm */
      {
        const f2 C1 = lds_ld2(w[XP_C1], lane);
        const f2 C2 = lds_ld2(w[XP_K3], lane) * (s.tsurf - tair);
        f2 PSIM = S2(0.f), PSIH = S2(0.f), BL = S2(0.f);
        const f2 lU = S2(c.logUstar), lC = S2(c.logCond);
        const uint64_t livex = __builtin_amdgcn_ballot_w64(!was_failed.x), livey = __builtin_amdgcn_ballot_w64(!was_failed.y);
        /* the first pass starts from PSIM = PSIH = 0 (:62): a = logUstar and b = logCond for every point - the
         * reciprocal and the cube once per wavefront, by the general pass's own operations on the same values */
        const float K0 = __builtin_amdgcn_rcpf(c.logUstar * c.logCond), U3 = (c.logUstar * c.logUstar) * c.logUstar;
        auto pass = [&](f2 &psim, f2 &psih, f2 &bl, uint64_t actx, uint64_t acty, bool first = false) {
          f2 Stab;
          if (first) {
            bl = C1 * S2(K0);
            Stab = (C2 * bl) * S2(U3);
          } else {
            const f2 av = lU + psim, bv = lC + psih;
            bl = C1 * rcp2(av * bv);
            Stab = (C2 * bl) * (av * av * av);
          }
          Stab = min2(Stab, S2(1.0f)); /* (`if (Stab > 1) Stab = 1`; v_min_f32 differs for a NaN only) */
          const bool stx = Stab.x > 0.f, sty = Stab.y > 0.f;
          const f2 ps = S2(4.7f) * Stab;
          const uint64_t ux = __builtin_amdgcn_ballot_w64(!stx) & actx, uy = __builtin_amdgcn_ballot_w64(!sty) & acty;
          if ((ux | uy) != 0ull) { /* (the plan order keeps the regimes together) */
            const f2 arg = (S2(1.0f) + sqrt2(fma2(S2(-16.0f), Stab, S2(1.0f)))) * S2(0.5f);
            const f2 pu = S2(-2.0f * 0.69314718056f) * f2{__builtin_amdgcn_logf(arg.x), __builtin_amdgcn_logf(arg.y)};
            const f2 pm = S2(0.6f) * pu;
            psih = f2{stx ? ps.x : pu.x, sty ? ps.y : pu.y};
            psim = f2{stx ? ps.x : pm.x, sty ? ps.y : pm.y};
            if (SCORE) {
              unsx |= ux;
              unsy |= uy;
            }
          } else {
            psih = ps;
            psim = ps;
          }
        };
#ifdef RS_ABL_NOBL
        const int npre = 1;
#else
        const int npre = 4;
#endif
        /* passes 1-4 never test (j >= 5 in the exit condition), the fifth is the first that may end the loop */
        pass(PSIM, PSIH, BL, livex, livey, true);
#pragma unroll 1
        for (int j = 2; j <= npre; ++j) pass(PSIM, PSIH, BL, livex, livey);
        uint64_t actx, acty;
        {
          const f2 old = BL;
          pass(PSIM, PSIH, BL, livex, livey);
          const f2 d = abs2(BL - old);
          actx = livex & ~__builtin_amdgcn_ballot_w64(d.x < 0.001f);
          acty = livey & ~__builtin_amdgcn_ballot_w64(d.y < 0.001f);
#ifdef RS_ABL_NOBL
          actx = acty = 0ull;
#endif
        }
        /* the tail: a point that has left the loop keeps its values (what a lane's exit does in the one-point flavours) */
        const bool count = SCORE || k == nsteps - 1;
        for (int j = 6; j <= RS_BL_MAXIT && (actx | acty) != 0ull; ++j) {
          f2 pm = PSIM, ph = PSIH, bl = BL;
          pass(pm, ph, bl, actx, acty);
          const f2 d = abs2(bl - BL);
          PSIM = f2{selm(actx, pm.x, PSIM.x), selm(acty, pm.y, PSIM.y)};
          PSIH = f2{selm(actx, ph.x, PSIH.x), selm(acty, ph.y, PSIH.y)};
          BL = f2{selm(actx, bl.x, BL.x), selm(acty, bl.y, BL.y)};
          if (count) trips = i2{addm(actx, trips.x), addm(acty, trips.y)};
          actx &= ~__builtin_amdgcn_ballot_w64(d.x < 0.001f);
          acty &= ~__builtin_amdgcn_ballot_w64(d.y < 0.001f);
        }
        blcond = BL;
        /* calcRaero (:112-131), CalcLE (:134-190) */
        f2 RAero = ((S2(c.logMom) + PSIM) * (S2(c.logHeat) + PSIH)) * lds_ld2(w[XP_RRA], lane);
        RAero = min2(RAero, S2(30.0f));
        const b2 sneg = lt2(s.tsurf, S2(0.f));
        const f2 as = sel2(sneg, S2(21.875f), S2(17.269f)), bs = sel2(sneg, S2(265.5f), S2(237.3f));
        const f2 ESurf = S2(0.61078f) * exp2v((as * s.tsurf) * rcp2(s.tsurf + bs));
        const f2 WatDen = fma2(S2(-0.0050f) * s.tsurf, s.tsurf, fma2(S2(0.0079f), s.tsurf, S2(1000.0028f)));
        f2 le_ = (lds_ld2(w[XP_AVCAP], lane) * (ESurf - lds_ld2(w[XP_EAIR], lane))) * rcp2(lds_ld2(w[XP_PSYCH], lane) * RAero);
        const f2 lat = sel2(ge2(s.tsurf, S2(0.f)), S2(c.LVap), S2(c.LFus));
        const f2 ev = ((le_ * rcp2(lat * WatDen)) * S2(1000.0f)) * S2(c.DTSecs);
        const b2 nowater = and2(gt2(le_, S2(0.f)), le2(s.wat, S2(0.f)));
        le = sel2(nowater, S2(0.f), le_);
        evap = sel2(nowater, S2(0.f), ev);
      }
      if (SCORE) {
        score = i2{score.x + trips.x - 5, score.y + trips.y - 5};
        if (k >= nsteps - 30) regime = i2{regime.x | (int32_t)selm(unsx, 1.f, 0.f), regime.y | (int32_t)selm(unsy, 1.f, 0.f)};
      } else {
        last_trips = trips;
      }
      /* CalcRNet (src/BalanceModel.f90:282-307) */
      f2 rnet;
      {
        const f2 TK = s.tsurf + S2(273.15f);
        const f2 TK2 = TK * TK;
        const f2 RBB = S2(c.Emiss * c.SB_Const) * (TK2 * TK2);
        rnet = fma2(S2(1.0f) - s.albedo, lds_ld2(w[XP_SW], lane), lds_ld2(w[XP_ELW], lane) - RBB);
      }
      /* layers 1-2 with Tmp(3) where a two-layer column has its lower boundary */
      const f2 t1old = T1, t2old = T2;
      f2 hs1 = S2(0.f);
      {
        const f2 dts = S2(c.DTSecs);
        f2 Gprev = ((rnet - le) + trffric) + blcond * (tair - t1old);
        X2Lit lit;
        lit.make();
        const f2 rcap1 = x2_layer_rcap(c, lit, 1, FULL ? stale1 : t1old, &hs1);
        const f2 G1 = S2(c.lk4[1][2]) * (t2old - t1old);
        T1 = fma2(dts, rcap1 * (Gprev - G1), t1old);
        const f2 rcap2 = x2_layer_rcap(c, lit, 2, FULL ? stale2 : t2old, nullptr);
        const f2 G2 = S2(c.lk4[2][2]) * (t3 - t2old);
        T2 = fma2(dts, rcap2 * (G1 - G2), t2old);
      }
      /* calcHStor (src/BalanceModel.f90:311-322), melting (src/Storage.f90:319-402), the new surface temperature, RoadCond
       * with the four storages, NewMeltFreezeHeat, CalcAlbedo (src/Cond.f90:9-139): per point through the one-point source
       * (compares and selects: nothing to pack), behind its wavefront-uniform shortcuts */
      {
        const f2 T1Ave = (t1old + S2(3.f) * t2old) * S2(0.25f);
        const f2 TN1Ave = (T1 + S2(3.f) * T2) * S2(0.25f);
        const f2 hstor = hs1 * (TN1Ave - T1Ave);
        const b2 frozen_cover = b2{(s.snow.x > 0.f) || (s.ice.x > 0.f) || (s.ice2.x > 0.f),
                                   (s.snow.y > 0.f) || (s.ice.y > 0.f) || (s.ice2.y > 0.f)};
        const bool melt_here = wave_any2(frozen_cover);
#ifndef RS_X2_ROAD_SCALAR
#ifndef RS_ABL_NOROAD /* (ablation builds, tools/experiments/r6_ablate.sh: what a part costs) */
        if (melt_here) x2_melting(s, T1, T2, hstor, hs1);
        else s.q2melt = S2(0.f);
#endif
        s.tsurf = (T1 + T2) * S2(0.5f);
#ifndef RS_ABL_NOROAD
        x2_road_condition(c, s, evap);
#endif
#else /* the one-point source, one point after the other (A/B) */
#pragma unroll
        for (int comp = 0; comp < 2; ++comp) {
          RegProfile<2> TT;
          TT.set(1, comp ? T1.y : T1.x);
          TT.set(2, comp ? T2.y : T2.x);
          Scalars q = x2_scalars(s, comp, TT.get(1), TT.get(2));
          if (melt_here) melting(q, TT, comp ? hstor.y : hstor.x, comp ? hs1.y : hs1.x, false, 0.f);
          else q.q2melt = 0.f;
          q.tsurf = (TT.get(1) + TT.get(2)) / 2.0f;
          road_condition(c, q, comp ? evap.y : evap.x);
          if (comp) {
            T1.y = TT.get(1); T2.y = TT.get(2);
            s.tsurf.y = q.tsurf; s.wat.y = q.wat; s.snow.y = q.snow; s.ice.y = q.ice; s.ice2.y = q.ice2; s.dep.y = q.dep;
            s.q2melt.y = q.q2melt; s.t4melt.y = q.t4melt; s.albedo.y = q.albedo; s.verycold.y = q.verycold;
          } else {
            T1.x = TT.get(1); T2.x = TT.get(2);
            s.tsurf.x = q.tsurf; s.wat.x = q.wat; s.snow.x = q.snow; s.ice.x = q.ice; s.ice2.x = q.ice2; s.dep.x = q.dep;
            s.q2melt.x = q.q2melt; s.t4melt.x = q.t4melt; s.albedo.x = q.albedo; s.verycold.x = q.verycold;
          }
        }
#endif
      }
    }
    /* SaveOutput (src/InputOutput.f90:151-165); -9999.0 for a point that failed before this index */
    if (write) {
      if (wave_any2(was_failed)) {
        const f2 m = S2(-9999.0f);
        out2(ka->o.tsurf, sel2(was_failed, m, s.tsurf)); out2(ka->o.snow, sel2(was_failed, m, s.snow));
        out2(ka->o.water, sel2(was_failed, m, s.wat)); out2(ka->o.ice, sel2(was_failed, m, s.ice));
        out2(ka->o.deposit, sel2(was_failed, m, s.dep)); out2(ka->o.ice2, sel2(was_failed, m, s.ice2));
      } else {
        out2(ka->o.tsurf, s.tsurf); out2(ka->o.snow, s.snow); out2(ka->o.water, s.wat);
        out2(ka->o.ice, s.ice); out2(ka->o.deposit, s.dep); out2(ka->o.ice2, s.ice2);
      }
    }
    lds_st2(mail.v[(k & 1) ^ 1][0], lane, T2);
    x2_meet();
  }
  stst(RS_ST_TMP0, T1);
  stst(RS_ST_TMP0 + 1, T2);
  stst(RS_ST_TSURF, s.tsurf);
  stst(RS_ST_WAT, s.wat); stst(RS_ST_SNOW, s.snow); stst(RS_ST_ICE, s.ice); stst(RS_ST_ICE2, s.ice2);
  stst(RS_ST_DEP, s.dep); stst(RS_ST_Q2MELT, s.q2melt); stst(RS_ST_T4MELT, s.t4melt);
  stst(RS_ST_ALBEDO, s.albedo);
  stst(RS_ST_VERYCOLD, f2{s.verycold.x ? 1.f : 0.f, s.verycold.y ? 1.f : 0.f});
  if (SCORE) { /* sort key of rs_hip_recluster, as in the other flavours (rs_kernels.hip, bl_score_key) */
    auto key = [&](int32_t sc, int32_t rg, float w_, float sn, float ic, float i2_, float dp) -> float {
      const int32_t lo = sc > 0x7ffff ? 0x7ffff : (sc < 0 ? 0 : sc);
      const int32_t covered = (w_ > 0.f || sn > 0.f || ic > 0.f || i2_ > 0.f || dp > 0.f) ? 1 : 0;
      return (float)(lo | (covered << 19) | (rg << 20));
    };
    stst(RS_ST_BLSCORE, f2{key(score.x, regime.x, s.wat.x, s.snow.x, s.ice.x, s.ice2.x, s.dep.x),
                           key(score.y, regime.y, s.wat.y, s.snow.y, s.ice.y, s.ice2.y, s.dep.y)});
  } else { /* the pass count of the launch's last index: one more preview for forecast_key_kernel */
    stst(RS_ST_BLSCORE, f2{(float)last_trips.x, (float)last_trips.y});
  }
}

template <int SRC, bool SCORE, bool FULL = false, bool SKY = false>
#ifndef RS_X2D_FULL_WAVES /* measured (tools/experiments/r6_f32_full_waves.sh, config 5's shape): the FULL knot-reading
                             instance at five wavefronts per SIMD - 96 registers, 119 spilled - 5.86e10; at four - 128
                             registers, 42 spilled - 5.59e10 */
#define RS_X2D_FULL_WAVES RS_X2D_WAVES
#endif
/* (SKY: four wavefronts per SIMD - the ground wave keeps a point pair's geometry in fp64 beside its thirteen layers) */
__global__ void __launch_bounds__(128, SKY ? 4 : FULL ? RS_X2D_FULL_WAVES : RS_X2D_WAVES) step_kernel_f32duo(const rs::StepArgs a) {
  __shared__ X2Mail mail;
  /* no early return: both wavefronts walk to every barrier; points beyond npoints are dead weight */
  if (threadIdx.x < 64) {
    if (a.surface_prio) __builtin_amdgcn_s_setprio(1); /* the longer chain of the two issues first (StepArgs::surface_prio) */
    x2d_surface<SCORE, FULL, SKY>(mail, a);
  } else {
    x2d_ground<SRC, FULL, SKY>(mail, a);
  }
}

/* ==== the general kernel in single precision: every feature, coupling included (src/Coupling.f90) ===========
 * fp32 twin of rs_kernels.hip's general kernel (Coupling, coupling_control, time_loop_coupled, step_kernel_coupled):
 * one point per lane, the profile in LDS (any NLayers), every lane with its OWN time index - a point that
 * Coupling_control sends back rewinds to its window start inside the loop (src/Coupling.f90:61-78), up to 25 times,
 * so a coupled launch is the whole series (rs_hip_step demands t0 = 1, nsteps = SimLen of a coupled plan) and
 * forcing reads and output writes are per-lane.  The FULL feature set rides along as in the fp64 kernel: dew-point
 * test, observation forcing (never inside or behind a coupling window: src/InputOutput.f90:116-124), relaxation, sky
 * view (the sun's position in fp64: rs_skyview.hpp), an output depth (tsurfOutputDepth or a depth stream:
 * getTempAtDepth, src/BalanceModel.f90:390-417).  It is also what an fp32 plan WITHOUT coupling launches for what the
 * two-wavefront kernels do not have - an output depth, the FULL set or sky view at NLayers != 15 - in chunks like
 * any other launch.  No diagnostics, no write-back of the in-place input edits; the time-chunked coupling pair
 * rs_hip_step_cpl / rs_hip_cpl_replay stays with the fp64 flavour.
 * Tolerance, not bits: Coupling_control stops on |Tsurf - observation| <= 0.1 K, so a replay more or less than the
 * fp64 run is possible where the two straddle that limit - tests/test_hip_f32.py says what is gated. */
struct Coupling32 {
  float tabove, tbelow, radcoeff, rcabove, rcbelow, rcprev, swcof, lwcof, swcorr, lwcorr, tend1, lastobs;
  int32_t iter, cs, ce, msg;
  bool again, failed, on;
};

/* Coupling_control, src/Coupling.f90:292-481 (the Kelvin round trip of TsurfAve and LastTsurfObs included) */
__device__ __forceinline__ void coupling_control32(Coupling32 &q, float &tsurf) {
  auto reset_cof = [&]() { q.swcof = 1.0f; q.lwcof = 1.0f; q.swcorr = 0.0f; q.lwcorr = 0.0f; };
  auto secant = [&]() {
    const float da = q.tabove - q.lastobs, db = q.lastobs - q.tbelow;
    return q.rcabove - rs_div(da, da + db) * (q.rcabove - q.rcbelow);
  };
  q.again = false;
  tsurf = tsurf + 273.16f;
  q.lastobs = q.lastobs + 273.16f;
  if (q.iter == 0) q.tend1 = tsurf;
  if (q.iter == 25) {
    if (__builtin_fabsf(q.tend1 - q.lastobs) < __builtin_fabsf(tsurf - q.lastobs)) q.again = true;
    reset_cof();
    q.radcoeff = 1.0f;
    q.failed = true;
  } else if (q.lastobs < -100.f) {
    reset_cof();
    q.radcoeff = 1.0f;
    q.failed = true;
    q.again = true;
  } else if (tsurf < 170.0f || tsurf > 400.0f) {
    reset_cof();
    q.failed = true;
    q.again = true;
    q.radcoeff = 1.0f;
  } else if (tsurf - q.lastobs > 0.1f) {
    if (q.tabove < -100.f || q.tabove - q.lastobs > tsurf - q.lastobs) {
      q.tabove = tsurf;
      q.rcabove = q.radcoeff;
    }
    q.again = true;
    q.radcoeff = (q.tabove > -100.f && q.tbelow > -100.f) ? secant() : 0.5f * q.radcoeff;
    if (__builtin_fabsf(q.radcoeff - q.rcprev) < 0.00005f) {
      q.tabove = -9999.f;
      q.tbelow = -9999.f;
    }
    if (q.radcoeff < 0.01f) { /* "coupling coefficient too small, coupling failed" (:400-401) */
      q.msg |= RS_CPL_MSG_SMALL;
      q.radcoeff = 1.0f;
      q.failed = true;
      reset_cof();
    }
    q.rcprev = q.radcoeff;
  } else if (q.lastobs - tsurf > 0.1f) {
    if (q.tbelow < -100.f || q.tbelow - q.lastobs < tsurf - q.lastobs) {
      q.tbelow = tsurf;
      q.rcbelow = q.radcoeff;
    }
    q.again = true;
    q.radcoeff = (q.tabove > -100.f && q.tbelow > -100.f) ? secant() : 2.0f * q.radcoeff;
    if (__builtin_fabsf(q.radcoeff - q.rcprev) < 0.00005f) {
      q.tabove = -9999.f;
      q.tbelow = -9999.f;
    }
    q.rcprev = q.radcoeff;
  } else {
    if (q.radcoeff > 3.0f) { /* "coupling coefficient too big, coupling failed" (:451-452) */
      q.msg |= RS_CPL_MSG_BIG;
      q.failed = true;
      q.radcoeff = 1.0f;
      reset_cof();
    }
    q.swcorr = q.swcof - 1.0f;
    q.lwcorr = q.lwcof - 1.0f;
    q.failed = false;
    q.iter = -1;
    q.tabove = -9999.0f; q.tbelow = -9999.0f;
    q.radcoeff = 1.0f;
    q.rcabove = -9999.0f; q.rcbelow = -9999.0f;
    q.rcprev = 1.0f;
  }
  tsurf = tsurf - 273.16f;
  q.lastobs = q.lastobs - 273.16f;
}

struct GlobalProfile32 { /* the stale TmpNw of a replay's first step, parked in the state block */
  const float *col;
  int64_t stride;
  __device__ __forceinline__ float get(int j) const { return col[(int64_t)(j - 1) * stride]; }
};

__device__ __forceinline__ Forcing gather_forcing32(KernArgs ka, int64_t p, int32_t i, int32_t t0) {
  Forcing o;
  const int64_t off = (int64_t)(i - t0) * ka->f.t_stride + p;
  auto F = [&](const double *base) { return reinterpret_cast<const float *>(base)[off]; };
  o.tair = F(ka->f.tair); o.vz = F(ka->f.vz); o.rhz = F(ka->f.rhz);
  o.prec = F(ka->f.prec); o.sw = F(ka->f.sw); o.lw = F(ka->f.lw);
  o.phase = ka->f.precphase[off];
  o.hour = ka->f.hour_pstride ? ka->f.hour[off] : ka->f.hour[i - t0];
  o.tdew = ka->f.tdew ? F(ka->f.tdew) : 0.f;
  o.tsurfobs = ka->f.tsurfobs ? F(ka->f.tsurfobs) : -9999.9f;
  o.depth = ka->f.depth ? F(ka->f.depth) : -9999.9f;
  return o;
}

__global__ void __launch_bounds__(kBlock, 2) step_kernel_f32_coupled(const rs::StepArgs a) {
  extern __shared__ float ldsf[]; /* [NLayers][kBlock] */
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= a.npoints) return; /* no barriers below: each lane owns its column */
  KernArgs ka = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
  const ConstsAS &c = consts_of(ka);
  const int64_t np = a.np_pad;
  float *st = reinterpret_cast<float *>(a.state);
  auto S = [&](int slot) -> float & { return st[(int64_t)slot * np + p]; };
  LdsProfile T{ldsf + threadIdx.x, c.NLayers};
  const int N = T.nlayers();
  rs::MathTab mt{nullptr, nullptr, nullptr};
  Scalars s;
  for (int j = 1; j <= N; ++j) T.set(j, S(RS_ST_TMP0 + j - 1));
  s.tnw1 = T.get(1); s.tnw2 = T.get(2);
  s.tsurf = S(RS_ST_TSURF); s.wat = S(RS_ST_WAT); s.snow = S(RS_ST_SNOW); s.ice = S(RS_ST_ICE); s.ice2 = S(RS_ST_ICE2);
  s.dep = S(RS_ST_DEP); s.q2melt = S(RS_ST_Q2MELT); s.t4melt = S(RS_ST_T4MELT); s.albedo = S(RS_ST_ALBEDO);
  s.verycold = S(RS_ST_VERYCOLD) != 0.f; s.failed = S(RS_ST_FAILED) != 0.f;
  s.tair_end = S(RS_ST_TAIR_END); s.vz_end = S(RS_ST_VZ_END); s.rh_end = S(RS_ST_RH_END);
  Coupling32 q{};
  q.swcof = q.lwcof = q.radcoeff = q.rcprev = 1.0f; /* (a plan without coupling: CouplingInputs' defaults) */
  if (c.use_coupling) {
    q.iter = (int32_t)S(RS_ST_CPL_ITER);
    const int32_t fl = (int32_t)S(RS_ST_CPL_FLAGS);
    q.again = fl & 1; q.failed = (fl >> 1) & 1; q.msg = fl & (RS_CPL_MSG_SMALL | RS_CPL_MSG_BIG);
    q.tabove = S(RS_ST_CPL_TABOVE); q.tbelow = S(RS_ST_CPL_TBELOW); q.radcoeff = S(RS_ST_CPL_RADCOEFF);
    q.rcabove = S(RS_ST_CPL_RCABOVE); q.rcbelow = S(RS_ST_CPL_RCBELOW); q.rcprev = S(RS_ST_CPL_RCPREV);
    q.swcof = S(RS_ST_CPL_SWCOF); q.lwcof = S(RS_ST_CPL_LWCOF); q.swcorr = S(RS_ST_CPL_SWCORR); q.lwcorr = S(RS_ST_CPL_LWCORR);
    q.tend1 = S(RS_ST_CPL_TEND1); q.lastobs = S(RS_ST_CPL_LASTOBS);
  }
  const int32_t t0 = ka->t0, tend = ka->t0 + ka->nsteps;
  const float tbot = (float)ka->pp.tbottom[p];
  const int32_t initlen = ka->pp.initlen ? ka->pp.initlen[p] : 0;
  bool relax = false;
  float tairR = 0.f, vzR = 0.f, rhR = 0.f;
  if (c.use_relaxation && ka->pp.tair_relax) {
    tairR = (float)ka->pp.tair_relax[p]; vzR = (float)ka->pp.vz_relax[p]; rhR = (float)ka->pp.rh_relax[p];
    relax = !(tairR < -100.0f || tairR > 100.0f || vzR < 0.0f || vzR > 100.0f || rhR < 0.0f || rhR > 110.f);
  }
  /* setInputParam + initCouplingTimes, src/InputOutput.f90:30-36, src/Coupling.f90:486-534 */
  const int32_t cidx = ka->pp.coupling_index ? ka->pp.coupling_index[p] : 0;
  q.on = c.use_coupling && ka->pp.coupling_index && !(ka->pp.coupling_tsurf[p] < -100 || cidx < 1);
  q.cs = -99; q.ce = -99;
  if (q.on) {
    q.ce = cidx;
    q.cs = ((float)cidx <= c.cplLenR) ? 1 : cidx - c.cplLenI;
  }
  double skyv = 1.0, sinlat = 0, coslat = 0, lonrad = 0, coslon = 1.0, sinlon = 0;
  bool sky_on = false;
  if (ka->pp.sky_view) {
    skyv = ka->pp.sky_view[p];
    sky_on = skyv < (double)1.0f && skyv > (double)-0.01f;
    if (sky_on) {
      sinlat = ka->pp.sin_lat[p]; coslat = ka->pp.cos_lat[p]; lonrad = ka->pp.lon_rad[p];
      coslon = ::cos(lonrad); sinlon = ::sin(lonrad);
    }
  }
  auto fail_at = [&](int32_t idx) {
    s.failed = true;
    S(RS_ST_FAILED) = (float)idx;
  };
  auto out_row = [&](int32_t i, int64_t &row) -> bool { /* SaveOutput's decimation (rs_kernels.hip output_row) */
    int32_t r = i - 1;
    const int32_t dec = ka->o.decimate;
    if (dec > 1) {
      if (r % dec != 0) return false;
      r /= dec;
    }
    row = ((int64_t)r - ka->o.row0) * ka->o.t_stride + p;
    return true;
  };
  auto store = [&](int64_t row, bool valid) {
    auto O = [&](double *base, float v) { reinterpret_cast<float *>(base)[row] = valid ? v : -9999.0f; };
    O(ka->o.tsurf, s.tsurf); O(ka->o.snow, s.snow); O(ka->o.water, s.wat);
    O(ka->o.ice, s.ice); O(ka->o.deposit, s.dep); O(ka->o.ice2, s.ice2);
  };
  auto sky_streams = [&](int32_t i, float &sw_dir, float &lw_net) {
    const int64_t off = (int64_t)(i - t0) * ka->f.t_stride + p;
    sw_dir = reinterpret_cast<const float *>(ka->f.sw_dir)[off];
    lw_net = reinterpret_cast<const float *>(ka->f.lw_net)[off];
  };
  const int32_t resume = c.use_coupling ? (int32_t)S(RS_ST_CPL_RESUME) : t0;
  int32_t i = resume > t0 ? resume : t0;
  int32_t written_hi = i - 1; /* highest index the point has saved an output for */
  bool stale_all = false;     /* first step after a restore: TmpNw is the pre-restore profile */
  while (i < tend) {
    if (s.failed) {
      /* the reference's loop has exited: rows it never saved stay -9999.0; a point that fails in the middle of a replay
       * keeps, beyond the failure, what the EARLIER passes saved there (src/InputOutput.f90:151-165 only overwrites) */
      int64_t orow;
      if (i > written_hi && out_row(i, orow)) store(orow, false);
      ++i;
      continue;
    }
    Forcing f = gather_forcing32(ka, p, i, t0);
    if (i == 1 && f.vz < 0.4f) f.vz = 0.4f;
    CouplingInputs cp;
    float sw_dir = 0.f, lw_net = 0.f;
    if (ka->f.sw_dir) sky_streams(i, sw_dir, lw_net);
    if (i < c.SimLen) {
      if (check_values(c, f, s.tsurf, ka->f.tdew != nullptr)) fail_at(i);
      if (sky_on && (sw_dir < -0.1f || sw_dir > 4000.0f || lw_net < -1000.0f || lw_net > 1000.0f)) fail_at(i); /* src/InputOutput.f90:68-74 */
      if (sw_dir > f.sw) sw_dir = f.sw;                                                                          /* :75-77 */
      if (q.on) { /* CouplingOperations1, src/Coupling.f90:10-96 */
        const bool in_phase = (i >= q.cs && i <= q.ce);
        if (i == q.cs && q.iter == 0) { /* saveDataForCoupling :172-210 */
          S(RS_ST_CPL_SAVE_TSURF) = s.tsurf; S(RS_ST_CPL_SAVE_WAT) = s.wat; S(RS_ST_CPL_SAVE_ICE2) = s.ice2;
          S(RS_ST_CPL_SAVE_DEP) = s.dep; S(RS_ST_CPL_SAVE_SNOW) = s.snow; S(RS_ST_CPL_SAVE_ALBEDO) = s.albedo;
          const int32_t fl = ((int32_t)S(RS_ST_CPL_FLAGS)) & (3 | RS_CPL_MSG_SMALL | RS_CPL_MSG_BIG);
          S(RS_ST_CPL_FLAGS) = (float)(fl | (s.verycold ? 4 : 0));
          for (int j = 1; j <= N; ++j) S(RS_ST_CPL_SAVE_TMP0 + j - 1) = T.get(j);
          q.swcof = 1.0f; q.lwcof = 1.0f; q.swcorr = 0.0f; q.lwcorr = 0.0f;
        }
        if (q.again) { /* uploadDataForCoupling :213-255: back to the window start; SrfIcemms, Q2Melt, T4Melt and TmpNw are NOT restored */
          i = q.cs;
          s.tsurf = S(RS_ST_CPL_SAVE_TSURF); s.wat = S(RS_ST_CPL_SAVE_WAT); s.ice2 = S(RS_ST_CPL_SAVE_ICE2);
          s.dep = S(RS_ST_CPL_SAVE_DEP); s.snow = S(RS_ST_CPL_SAVE_SNOW); s.albedo = S(RS_ST_CPL_SAVE_ALBEDO);
          s.verycold = (((int32_t)S(RS_ST_CPL_FLAGS)) & 4) != 0;
          for (int j = 1; j <= N; ++j) {
            S(RS_ST_CPL_STALE_TMP0 + j - 1) = T.get(j); /* TmpNw keeps the end-of-window profile */
            T.set(j, S(RS_ST_CPL_SAVE_TMP0 + j - 1));
          }
          stale_all = true;
          q.again = false;
          f = gather_forcing32(ka, p, i, t0);
          if (i == 1 && f.vz < 0.4f) f.vz = 0.4f;
          if (ka->f.sw_dir) { /* the restored window holds SW_dir as CheckValues left it in the first pass: clamped (:204-208,249-253) */
            sky_streams(i, sw_dir, lw_net);
            if (sw_dir > f.sw) sw_dir = f.sw;
          }
          if (f.sw > f.lw && !sky_on) { /* short-wave scaling by day, long-wave by night - and always with sky view (:68-76) */
            q.swcof = q.radcoeff;
            q.lwcof = 1.0f;
          } else {
            q.swcof = 1.0f;
            q.lwcof = q.radcoeff;
          }
        }
        if (i > q.ce) { /* the correction decays behind the window (:80-88) */
          const float e = __expf(rs_div(-((c.DTSecs * (float)i) - (c.DTSecs * (float)q.ce)), c.cplReduction));
          q.swcof = 1.0f + q.swcorr * e;
          q.lwcof = 1.0f + q.lwcorr * e;
        }
        if (in_phase) { /* snowIceCheck :259-289 */
          if (q.lastobs > c.TLimMeltSnow && s.snow > 0.f) { s.wat = s.wat + s.snow; s.snow = 0.f; }
          if (q.lastobs > c.TLimMeltIce && s.ice > 0.f) { s.wat = s.wat + s.ice; s.ice = 0.f; }
          if (q.lastobs > c.TLimMeltIce && s.ice2 > 0.f) s.ice2 = 0.f;
          if (q.lastobs > c.TLimMeltDep && s.dep > 0.f) { s.wat = s.wat + s.dep; s.dep = 0.f; }
        }
        cp.in_phase = in_phase;
      }
      /* SetCurrentValues' observation forcing, src/InputOutput.f90:116-148 */
      if ((i <= initlen || c.force_tsurf) && f.tsurfobs > -100.0f && (!q.on || i < q.cs)) {
        T.set(1, f.tsurfobs);
        T.set(2, f.tsurfobs);
        s.tsurf = surface_temperature(c, T, tbot, (c.tsurfOutputDepth >= 0.0f) ? c.tsurfOutputDepth : f.depth);
      }
    } else { /* lastValues, src/InputOutput.f90:169-198; coupling%inCouplingPhase keeps the value of index SimLen - 1 */
      s.tsurf = surface_temperature(c, T, tbot, f.depth);
      cp.in_phase = q.on && (c.SimLen - 1 >= q.cs && c.SimLen - 1 <= q.ce);
    }
    float tair = f.tair, vz = f.vz, rhz = f.rhz;
    const float prec_ts = rs_div(f.prec, 3600.0f) * c.DTSecs;
    if (i < c.SimLen && relax) { /* RelaxationOperations, src/Relaxation.f90:10-47 */
      if (i == initlen) { s.tair_end = tair; s.vz_end = vz; s.rh_end = rhz; }
      if (i > initlen) {
        const float e = __expf(rs_div(-((c.DTSecs * (float)i) - (c.DTSecs * (float)initlen)), 4.f * 3600.f));
        tair = tair - (tairR - s.tair_end) * e;
        vz = vz - (vzR - s.vz_end) * e;
        rhz = rhz - (rhR - s.rh_end) * e;
        if (rhz > 100.f) rhz = 100.0f;
      }
    }
    cp.sw_cof = q.swcof; cp.lw_cof = q.lwcof; cp.last_tsurf_obs = q.lastobs;
    float sw_in = f.sw, lw_in = f.lw;
    if (sky_on) { /* ModRadiationBySurroundings (examples/example1/src/Simulation.f90:151-162) */
      double dsw = sw_in, dsd = sw_dir, dlw = lw_in;
      const int64_t hcol = ka->pp.horizon_index ? (int64_t)ka->pp.horizon_index[p] : p;
      if (!rs::sky_view_radiation(ka->f.sun + (int64_t)(i - t0) * RS_SUN_COLS, sinlat, coslat, lonrad, coslon, sinlon, skyv,
                                  ka->pp.albedo_surroundings,
                                  ka->pp.horizons ? ka->pp.horizons + hcol * (ka->pp.horizons_by_point ? 360 : 1) : nullptr,
                                  ka->pp.horizons_by_point ? (int64_t)1 : np, dsw, dsd, dlw, (double)lw_net))
        fail_at(i); /* the reference would `stop` the process here */
      sw_in = (float)dsw;
      lw_in = (float)dlw;
    }
    const Fluxes fx = model_step_fluxes(c, mt, s, tair, vz, rhz, prec_ts, sw_in, lw_in, f.phase, f.hour, cp);
    if (stale_all) { /* observation forcing cannot follow a restore (i >= couplingStartI): TmpNw(1:2) are the stale values too */
      const GlobalProfile32 Tstale{st + (int64_t)RS_ST_CPL_STALE_TMP0 * np + p, np};
      model_step_ground<LdsProfile, GlobalProfile32, true>(c, s, T, tbot, tair, fx, f.depth, cp, &Tstale);
      stale_all = false;
    } else {
      model_step_ground<LdsProfile, LdsProfile, true>(c, s, T, tbot, tair, fx, f.depth, cp);
    }
    int64_t orow;
    if (out_row(i, orow)) store(orow, true);
    if (i > written_hi) written_hi = i;
    /* CheckEndCoupling + CouplingOperations2, src/Coupling.f90:98-141 */
    if (i < c.SimLen && q.on && i == q.ce && !q.failed) {
      if (q.iter == 0) q.tend1 = s.tsurf;
      coupling_control32(q, s.tsurf);
      q.iter = q.iter + 1;
      if (s.failed) q.again = false; /* failed at this index: the loop exits, no rewind */
    }
    ++i;
  }
  if (c.use_coupling) S(RS_ST_CPL_RESUME) = (float)i;
  for (int j = 1; j <= N; ++j) S(RS_ST_TMP0 + j - 1) = T.get(j);
  S(RS_ST_TNW1) = s.tnw1; S(RS_ST_TNW2) = s.tnw2;
  S(RS_ST_TSURF) = s.tsurf; S(RS_ST_WAT) = s.wat; S(RS_ST_SNOW) = s.snow; S(RS_ST_ICE) = s.ice; S(RS_ST_ICE2) = s.ice2;
  S(RS_ST_DEP) = s.dep; S(RS_ST_Q2MELT) = s.q2melt; S(RS_ST_T4MELT) = s.t4melt; S(RS_ST_ALBEDO) = s.albedo;
  S(RS_ST_VERYCOLD) = s.verycold ? 1.f : 0.f;
  S(RS_ST_TAIR_END) = s.tair_end; S(RS_ST_VZ_END) = s.vz_end; S(RS_ST_RH_END) = s.rh_end;
  if (c.use_coupling) {
    S(RS_ST_CPL_ITER) = (float)q.iter;
    const int32_t keep = ((int32_t)S(RS_ST_CPL_FLAGS)) & 4;
    S(RS_ST_CPL_FLAGS) = (float)(keep | (q.again ? 1 : 0) | (q.failed ? 2 : 0) | q.msg);
    S(RS_ST_CPL_TABOVE) = q.tabove; S(RS_ST_CPL_TBELOW) = q.tbelow; S(RS_ST_CPL_RADCOEFF) = q.radcoeff;
    S(RS_ST_CPL_RCABOVE) = q.rcabove; S(RS_ST_CPL_RCBELOW) = q.rcbelow; S(RS_ST_CPL_RCPREV) = q.rcprev;
    S(RS_ST_CPL_SWCOF) = q.swcof; S(RS_ST_CPL_LWCOF) = q.lwcof; S(RS_ST_CPL_SWCORR) = q.swcorr; S(RS_ST_CPL_LWCORR) = q.lwcorr;
    S(RS_ST_CPL_TEND1) = q.tend1; S(RS_ST_CPL_LASTOBS) = q.lastobs;
  }
}

}  // namespace rs32

static inline dim3 grid_for32(int64_t n) { return dim3((unsigned)((n + RS_BLOCK - 1) / RS_BLOCK)); }

size_t rs32_constants_bytes(void) { return sizeof(RsConstantsF); }

hipError_t rs32_upload_constants(void *dst, const RsConstants *c, hipStream_t stream) {
  RsConstantsF f;
  rs_constants_to_f32(*c, f);
  hipError_t e = hipMemcpyAsync(dst, &f, sizeof(f), hipMemcpyHostToDevice, stream);
  if (e != hipSuccess) return e;
  /* the frozen-layer constants, by the lanes' own instructions */
  hipLaunchKernelGGL(rs32::prepare_constants_f32, dim3(1), dim3(64), 0, stream, static_cast<RsConstantsF *>(dst));
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(stream); /* f is stack scratch */
}

static inline dim3 grid_x2(int64_t n) { return dim3((unsigned)((n + 127) / 128)); } /* a workgroup steps 128 points */

#define RS32_DUO(SRC)                                                                                                  \
  do {                                                                                                                 \
    const dim3 g2 = grid_x2(a.npoints);                                                                                \
    if (full && score) hipLaunchKernelGGL((rs32::step_kernel_f32duo<SRC, true, true>), g2, dim3(128), 0, stream, a);   \
    else if (full) hipLaunchKernelGGL((rs32::step_kernel_f32duo<SRC, false, true>), g2, dim3(128), 0, stream, a);      \
    else if (score) hipLaunchKernelGGL((rs32::step_kernel_f32duo<SRC, true, false>), g2, dim3(128), 0, stream, a);     \
    else hipLaunchKernelGGL((rs32::step_kernel_f32duo<SRC, false, false>), g2, dim3(128), 0, stream, a);               \
  } while (0)

/* full: the launch carries the FULL feature set (dew point, observation forcing, relaxation): the two-points-per-lane
 * kernel only (NLayers = 15); sky: and per-point sky view (a window with SW_dir and LW_net, the sun table) */
hipError_t rs32_launch_step(const rs::StepArgs &a, int NL, int variant, bool score, bool full, bool sky, hipStream_t stream) {
  const dim3 g = grid_for32(a.npoints), b(RS_BLOCK);
  const int v = variant;
  if (sky) {
    if (NL != 15) return hipErrorInvalidValue;
    const dim3 g2 = grid_x2(a.npoints);
    if (score) hipLaunchKernelGGL((rs32::step_kernel_f32duo<rs32::X2_WINDOW, true, true, true>), g2, dim3(128), 0, stream, a);
    else hipLaunchKernelGGL((rs32::step_kernel_f32duo<rs32::X2_WINDOW, false, true, true>), g2, dim3(128), 0, stream, a);
    return hipGetLastError();
  }
  if (NL == 15 && (full || (v != RS_VARIANT_REG && v != RS_VARIANT_LDS))) {
    /* two points per lane, two wavefronts per 128 points (round 6); RS_VARIANT_REG / _LDS: round 2-5's one point
     * per lane with the profile in LDS, for A/B (and what other layer counts take) */
    RS32_DUO(rs32::X2_WINDOW);
  } else {
    if (full) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rs32::step_kernel_f32_lds, g, b, (size_t)NL * RS_BLOCK * sizeof(float), stream, a);
  }
  return hipGetLastError();
}

/* the two-points-per-lane kernel reading the hourly knots itself (StepArgs::knots): no forcing window */
hipError_t rs32_launch_step_knots(const rs::StepArgs &a, bool score, bool full, hipStream_t stream) {
  RS32_DUO(rs32::X2_KNOTS);
  return hipGetLastError();
}
#undef RS32_DUO

/* the general kernel: a coupled plan's whole series (every point replays its coupling window inside the launch), or a
 * chunk of a plan without coupling whose features the two-wavefront kernels do not have */
hipError_t rs32_launch_step_coupled(const rs::StepArgs &a, int NL, hipStream_t stream) {
  hipLaunchKernelGGL(rs32::step_kernel_f32_coupled, grid_for32(a.npoints), dim3(RS_BLOCK), (size_t)NL * RS_BLOCK * sizeof(float), stream, a);
  return hipGetLastError();
}

hipError_t rs32_launch_init(const rs::InitArgs &a, hipStream_t stream) {
  hipLaunchKernelGGL(rs32::init_kernel_f32, grid_for32(a.npoints), dim3(RS_BLOCK), 0, stream, a);
  return hipGetLastError();
}

hipError_t rs32_launch_expand(const rs::ExpandArgs &a, int32_t nintervals, hipStream_t stream) {
  dim3 g = grid_for32(a.npoints);
  g.y = (unsigned)nintervals;
  hipLaunchKernelGGL(rs32::expand_kernel_f32, g, dim3(RS_BLOCK), 0, stream, a);
  return hipGetLastError();
}
