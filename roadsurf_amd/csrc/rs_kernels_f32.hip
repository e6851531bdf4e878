/*
 * rs_kernels_f32.hip — single-precision flavour of the model (BASELINE.json configs[4]:
 * "fp32 kernels with fp64 tolerance gate").  NOT the parity path: state, forcing, outputs and
 * arithmetic are fp32, exp/log/rcp/sqrt are the hardware approximations.  It exists to put a
 * number on what fp32 buys (2x less HBM traffic, ~2x the VALU rate) and what it costs: the
 * reference's storage logic branches on rounding residuals (rs_math.hpp), so an fp32 run
 * cannot track an fp64 run point by point; tests/test_hip_f32.py gates the DISTRIBUTION of
 * the differences against the fp64 oracle, with the tolerance written there.
 *
 * Same physics source as the fp64 path (rs_physics_body.inc instantiated with float); LEAN
 * feature set only (no observation forcing after index 1, output depth, relaxation, coupling,
 * sky view).  One point per lane, profile in registers (NLayers = 15) or LDS.
 */
#include <hip/hip_runtime.h>
#include "rs_math.hpp"
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

__global__ void __launch_bounds__(kBlock, 4) step_kernel_f32_reg15(const rs::StepArgs a) {
  if ((int64_t)blockIdx.x * kBlock + threadIdx.x >= a.npoints) return;
  RegProfile<15> T;
  run(a, T);
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
  st[(int64_t)RS_ST_TSURF * np + p] = (t4 + t4) / 2.0f;
  st[(int64_t)RS_ST_WAT * np + p] = 0.f; st[(int64_t)RS_ST_SNOW * np + p] = 0.f;
  st[(int64_t)RS_ST_ICE * np + p] = 0.f; st[(int64_t)RS_ST_ICE2 * np + p] = 0.f;
  st[(int64_t)RS_ST_DEP * np + p] = 0.f; st[(int64_t)RS_ST_Q2MELT * np + p] = 0.f;
  st[(int64_t)RS_ST_T4MELT * np + p] = c.T4Melt0; st[(int64_t)RS_ST_ALBEDO * np + p] = c.Albedo0;
  st[(int64_t)RS_ST_VERYCOLD * np + p] = 0.f; st[(int64_t)RS_ST_FAILED * np + p] = 0.f;
  st[(int64_t)RS_ST_BLSCORE * np + p] = 0.f;
}

/* fp32 twin of expand_kernel (rs_kernels.hip: one basic block per time index, stores with a scalar
 * row base): same knots (fp64), rounded once at the end. */
template <bool TDEW, bool OBS>
__global__ void __launch_bounds__(kBlock) expand_kernel_f32(const rs::ExpandArgs a) {
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
  const double span = (double)a.spk;
  const int64_t col0 = (int64_t)blockIdx.x * kBlock;
  for (int32_t t = tlo; t < thi; ++t) {
    const int32_t r = t - k * a.spk;
    const double secs = (double)r;
    const int64_t row = (int64_t)(t - (a.t0 - 1)) * a.f.t_stride + col0;
    uint32_t b4; /* the lane's byte offset, formed in this block (rs_kernels.hip, LaneOff) */
    asm volatile("v_lshlrev_b32 %0, 2, %1" : "=v"(b4) : "v"(threadIdx.x));
    auto st = [&](const void *base, float v) { *(float *)((char *)((float *)base + row) + b4) = v; };
    float v[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) v[q] = (float)(v0[q] + rs::rs_div_u(secs * dv[q], span, a.r_spk));
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

}  // namespace rs32

static inline dim3 grid_for32(int64_t n) { return dim3((unsigned)((n + RS_BLOCK - 1) / RS_BLOCK)); }

size_t rs32_constants_bytes(void) { return sizeof(RsConstantsF); }

hipError_t rs32_upload_constants(void *dst, const RsConstants *c, hipStream_t stream) {
  RsConstantsF f;
  rs_constants_to_f32(*c, f);
  hipError_t e = hipMemcpyAsync(dst, &f, sizeof(f), hipMemcpyHostToDevice, stream);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(stream); /* f is stack scratch */
}

hipError_t rs32_launch_step(const rs::StepArgs &a, int NL, int variant, hipStream_t stream) {
  const dim3 g = grid_for32(a.npoints), b(RS_BLOCK);
  /* fp32 default: LDS profile (66 VGPRs, 7 waves/SIMD; measured 10 % faster than registers) */
  if (variant % 10 == RS_VARIANT_REG) {
    if (NL != 15) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rs32::step_kernel_f32_reg15, g, b, 0, stream, a);
  } else {
    hipLaunchKernelGGL(rs32::step_kernel_f32_lds, g, b, (size_t)NL * RS_BLOCK * sizeof(float), stream, a);
  }
  return hipGetLastError();
}

hipError_t rs32_launch_init(const rs::InitArgs &a, hipStream_t stream) {
  hipLaunchKernelGGL(rs32::init_kernel_f32, grid_for32(a.npoints), dim3(RS_BLOCK), 0, stream, a);
  return hipGetLastError();
}

hipError_t rs32_launch_expand(const rs::ExpandArgs &a, int32_t nintervals, hipStream_t stream) {
  dim3 g = grid_for32(a.npoints);
  g.y = (unsigned)nintervals;
  if (a.f.tdew && a.f.tsurfobs)
    hipLaunchKernelGGL((rs32::expand_kernel_f32<true, true>), g, dim3(RS_BLOCK), 0, stream, a);
  else if (a.f.tdew)
    hipLaunchKernelGGL((rs32::expand_kernel_f32<true, false>), g, dim3(RS_BLOCK), 0, stream, a);
  else if (a.f.tsurfobs)
    hipLaunchKernelGGL((rs32::expand_kernel_f32<false, true>), g, dim3(RS_BLOCK), 0, stream, a);
  else
    hipLaunchKernelGGL((rs32::expand_kernel_f32<false, false>), g, dim3(RS_BLOCK), 0, stream, a);
  return hipGetLastError();
}
