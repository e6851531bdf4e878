/*
 * rs_kernels.hip — HIP kernels (gfx950) of the RoadSurf hot path.
 *
 * Mapping: ONE POINT PER LANE, persistent over a window of time steps.  A
 * 256-thread workgroup owns 256 consecutive points; lane l of a wavefront reads
 * forcing[t][p0 + l], so each field-step is one coalesced 512-B wave access,
 * and writes the six outputs the same way.  The carried state (SURVEY.md
 * Appendix B, ~30 doubles) is loaded from the SoA state block once per launch,
 * lives in VGPRs for the whole window and is stored back once.
 *
 * Two flavours of the same physics (rs_physics.hpp):
 *   step_kernel_reg<NL>  NLayers known at compile time: the ground temperature
 *                        profile is a fully unrolled register array.
 *   step_kernel_lds      any NLayers <= RS_MAX_LAYERS: the profile is staged in
 *                        LDS as [layer][lane] columns (conflict-free: lane l
 *                        touches 8-byte word l of a 2-KiB row).
 * Each comes in a LEAN variant (no observation forcing beyond index 1, no output
 * depth, no relaxation, no Tdew check: the BASELINE workload) and a FULL variant.
 *
 * No MFMA: there is no contraction anywhere in this path.  The time loop is
 * sequential per point (explicit Euler in time), points are independent.
 */
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "rs_physics.hpp"
#include "rs_state.h"
#include "rs_synth.h"
#include "rs_kernels.h"

#ifndef RS_CPL_WAVES
#define RS_CPL_WAVES 2 /* waves per SIMD the general (per-lane time index) kernel is compiled for */
#endif
#ifndef RS_REGIME_WINDOW
#define RS_REGIME_WINDOW 30 /* indices at the end of a launch that define a point's regime
                               (rs_hip_recluster; 8 ... 90 measured equal) */
#endif

namespace rs {

constexpr int kBlock = RS_BLOCK;

/* Model constants: one RsConstants block per plan in HBM, reached through a pointer in the
 * kernel arguments and read as constant memory (address space 4): every access is a scalar
 * load through the scalar cache.  (A per-plan block, not a __constant__ table: the reference
 * driver calls runsimulation from as many threads as it likes,
 * examples/example1/src/roadrunner.cpp:490-497, so the number of live plans is unbounded.)
 * The time loop re-derives the pointer through an empty asm every iteration so that the
 * compiler cannot hoist ~220 uniform doubles (= 440 SGPRs, against 102 available) out of the
 * loop and then spill them into VGPR lanes; loaded at the point of use they cost one s_load
 * each and no live range. */
typedef RsConstantsDev __attribute__((address_space(4))) ConstsAS;
template <class Args>
__device__ __forceinline__ const ConstsAS &consts_of(Args a) {
  return *(const ConstsAS *)a->consts;
}

template <int NL>
struct RegProfile {
  double v[NL];
  static constexpr bool kUnrolled = true; /* the layer count is a compile-time constant */
  __device__ __forceinline__ constexpr int nlayers() const { return NL; }
  __device__ __forceinline__ double get(int j) const { return v[j - 1]; }
  __device__ __forceinline__ void set(int j, double x) { v[j - 1] = x; }
  /* "the new profile exists here": without it the compiler sinks the update of the layers nothing in
   * the rest of the step reads (3..N) behind the storages into the loop latch, where the layer
   * constants no longer fit the scalar registers (124 v_readlane/v_writelane per step) */
  __device__ __forceinline__ void pin() {
#pragma unroll
    for (int j = 0; j < NL; ++j) asm volatile("" : "+v"(v[j]));
  }
};

/* The upper NREG layers in registers, the rest in LDS columns: what lets the FULL feature set (more
 * live values than the LEAN one) keep four waves per SIMD without scratch spills.  Layer indices are
 * compile-time constants after unrolling, so the split costs no selects. */
template <int NL, int NREG>
struct HybridProfile {
  double v[NREG];
  double *col; /* &lds[threadIdx.x]; layer NREG + 1 + r is col[r * kBlock] */
  static constexpr bool kUnrolled = true; /* the layer count is a compile-time constant */
  __device__ __forceinline__ constexpr int nlayers() const { return NL; }
  __device__ __forceinline__ double get(int j) const { return j <= NREG ? v[j - 1] : col[(j - NREG - 1) * kBlock]; }
  __device__ __forceinline__ void set(int j, double x) {
    if (j <= NREG) v[j - 1] = x;
    else col[(j - NREG - 1) * kBlock] = x;
  }
  __device__ __forceinline__ void pin() {
#pragma unroll
    for (int j = 0; j < NREG; ++j) asm volatile("" : "+v"(v[j]));
  }
};

struct LdsProfile {
  double *col; /* &lds[threadIdx.x]; layer stride = kBlock doubles */
  int n;
  static constexpr bool kUnrolled = false;
  __device__ __forceinline__ int nlayers() const { return n; }
  __device__ __forceinline__ double get(int j) const { return col[(j - 1) * kBlock]; }
  __device__ __forceinline__ void set(int j, double x) { col[(j - 1) * kBlock] = x; }
  __device__ __forceinline__ void pin() {}
};

template <bool FULL, class Prof>
__device__ __forceinline__ void load_state(const double *__restrict__ st, int64_t np, int64_t p,
                                           Prof &T, Scalars &s) {
  const int N = T.nlayers();
#pragma unroll
  for (int j = 1; j <= N; ++j) T.set(j, st[(int64_t)(RS_ST_TMP0 + j - 1) * np + p]);
  s.tnw1 = st[(int64_t)RS_ST_TNW1 * np + p];
  s.tnw2 = st[(int64_t)RS_ST_TNW2 * np + p];
  s.tsurf = st[(int64_t)RS_ST_TSURF * np + p];
  s.wat = st[(int64_t)RS_ST_WAT * np + p];
  s.snow = st[(int64_t)RS_ST_SNOW * np + p];
  s.ice = st[(int64_t)RS_ST_ICE * np + p];
  s.ice2 = st[(int64_t)RS_ST_ICE2 * np + p];
  s.dep = st[(int64_t)RS_ST_DEP * np + p];
  s.q2melt = st[(int64_t)RS_ST_Q2MELT * np + p];
  s.t4melt = st[(int64_t)RS_ST_T4MELT * np + p];
  s.albedo = st[(int64_t)RS_ST_ALBEDO * np + p];
  s.verycold = st[(int64_t)RS_ST_VERYCOLD * np + p] != 0.0;
  s.failed = st[(int64_t)RS_ST_FAILED * np + p] != 0.0;
  if (FULL) { /* relaxation anchors: untouched by the LEAN variant */
    s.tair_end = st[(int64_t)RS_ST_TAIR_END * np + p];
    s.vz_end = st[(int64_t)RS_ST_VZ_END * np + p];
    s.rh_end = st[(int64_t)RS_ST_RH_END * np + p];
  } else {
    s.tair_end = s.vz_end = s.rh_end = 0.0;
  }
}

/* ANCHORS: the relaxation anchors are part of the Scalars (the general coupled kernel); the
 * lock-step loop writes them to the state block at the index that sets them and carries differences */
template <bool FULL, class Prof, bool ANCHORS = false>
__device__ __forceinline__ void store_state(double *__restrict__ st, int64_t np, int64_t p,
                                            const Prof &T, const Scalars &s) {
  const int N = T.nlayers();
#pragma unroll
  for (int j = 1; j <= N; ++j) st[(int64_t)(RS_ST_TMP0 + j - 1) * np + p] = T.get(j);
  st[(int64_t)RS_ST_TNW1 * np + p] = s.tnw1;
  st[(int64_t)RS_ST_TNW2 * np + p] = s.tnw2;
  st[(int64_t)RS_ST_TSURF * np + p] = s.tsurf;
  st[(int64_t)RS_ST_WAT * np + p] = s.wat;
  st[(int64_t)RS_ST_SNOW * np + p] = s.snow;
  st[(int64_t)RS_ST_ICE * np + p] = s.ice;
  st[(int64_t)RS_ST_ICE2 * np + p] = s.ice2;
  st[(int64_t)RS_ST_DEP * np + p] = s.dep;
  st[(int64_t)RS_ST_Q2MELT * np + p] = s.q2melt;
  st[(int64_t)RS_ST_T4MELT * np + p] = s.t4melt;
  st[(int64_t)RS_ST_ALBEDO * np + p] = s.albedo;
  st[(int64_t)RS_ST_VERYCOLD * np + p] = s.verycold ? 1.0 : 0.0;
  /* RS_ST_FAILED is written where the failure is detected (the index it happened at) */
  if (FULL && ANCHORS) {
    st[(int64_t)RS_ST_TAIR_END * np + p] = s.tair_end;
    st[(int64_t)RS_ST_VZ_END * np + p] = s.vz_end;
    st[(int64_t)RS_ST_RH_END * np + p] = s.rh_end;
  }
}

/* The kernel arguments are read through the kernarg segment pointer (constant
 * address space: scalar loads) and that pointer is laundered once per time
 * step.  Effect: the 17 stream pointers + strides are s_load-ed at their point
 * of use instead of living in ~45 SGPRs for the whole time loop (SGPR spills go
 * to VGPR lanes and every use of a spilled value costs a v_readlane). */
typedef const StepArgs __attribute__((address_space(4))) *KernArgs;

__device__ __forceinline__ KernArgs kernargs() {
  return (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
}

/* Addressing: `base` pointers below are wave-uniform (row pointer + first point
 * of the workgroup), `lane` is the 32-bit index of the point inside the
 * workgroup, so every access is global_load/store saddr + 32-bit voffset with no
 * 64-bit per-lane arithmetic. */
/* base[lane] as saddr + 32-bit voffset.  Instruction selection works a basic block at a time and
 * takes that form only when it sees the 32-bit lane offset being widened in the block of the
 * access; a `lane * 8` from the kernel's entry block arrives as a 64-bit register pair and costs a
 * 64-bit vector add per access.  LaneOff therefore forms the byte offset behind an asm barrier in
 * the block that uses it (one v_lshlrev per group of accesses). */
struct LaneOff {
  uint32_t b8;
  __device__ __forceinline__ explicit LaneOff(uint32_t lane) {
    asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(b8) : "v"(lane));
  }
  template <class T>
  __device__ __forceinline__ T ld(const T *base) const {
    static_assert(sizeof(T) == 8 || sizeof(T) == 4, "");
    return *(const T *)((const char *)base + (sizeof(T) == 8 ? b8 : b8 >> 1));
  }
  template <class T>
  __device__ __forceinline__ void st(T *base, T v) const {
    static_assert(sizeof(T) == 8 || sizeof(T) == 4, "");
    *(T *)((char *)base + (sizeof(T) == 8 ? b8 : b8 >> 1)) = v;
  }
};

/* A32 windows (the launcher's choice: every stream of the window spans < 4 GiB): the whole offset
 * of (row, point) fits the 32-bit vector offset, so the stream pointers are used as they come out
 * of the kernel arguments - no 64-bit scalar add per stream and step (2 x 13 streams), one scalar
 * multiply-add for the row and one vector add.  `slot` = row * t_stride + first point of the
 * workgroup (uniform, < 2^29). */
struct WinOff {
  uint32_t b8;
  __device__ __forceinline__ WinOff(uint32_t lane, uint32_t slot) {
    const uint32_t sb = slot << 3;
    asm volatile("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(b8) : "v"(lane), "s"(sb));
  }
  template <class T>
  __device__ __forceinline__ T ld(const T *base) const {
    static_assert(sizeof(T) == 8 || sizeof(T) == 4, "");
    return *(const T *)((const char *)base + (sizeof(T) == 8 ? b8 : b8 >> 1));
  }
  template <class T>
  __device__ __forceinline__ void st(T *base, T v) const {
    static_assert(sizeof(T) == 8 || sizeof(T) == 4, "");
    *(T *)((char *)base + (sizeof(T) == 8 ? b8 : b8 >> 1)) = v;
  }
};

template <bool FULL, bool A32 = false>
__device__ __forceinline__ Forcing load_forcing(KernArgs ka, int64_t row0, uint32_t lane,
                                                int32_t k) {
  Forcing o;
  if (A32) {
    const WinOff L(lane, (uint32_t)k * (uint32_t)ka->f.t_stride + (uint32_t)row0);
    o.tair = L.ld(ka->f.tair);
    o.vz = L.ld(ka->f.vz);
    o.rhz = L.ld(ka->f.rhz);
    o.prec = L.ld(ka->f.prec);
    o.sw = L.ld(ka->f.sw);
    o.lw = L.ld(ka->f.lw);
    o.phase = L.ld(ka->f.precphase);
    if (ka->f.hour_pstride) o.hour = L.ld(ka->f.hour);
    else o.hour = ka->f.hour[k];
    if (FULL) {
      o.tdew = ka->f.tdew ? L.ld(ka->f.tdew) : 0.0;
      o.tsurfobs = ka->f.tsurfobs ? L.ld(ka->f.tsurfobs) : R4(-9999.9);
      o.depth = ka->f.depth ? L.ld(ka->f.depth) : R4(-9999.9);
    } else {
      o.tdew = 0.0;
      o.tsurfobs = R4(-9999.9);
      o.depth = R4(-9999.9);
    }
    return o;
  }
  const int64_t row = (int64_t)k * ka->f.t_stride + row0;
  const LaneOff L(lane);
  o.tair = L.ld(ka->f.tair + row);
  o.vz = L.ld(ka->f.vz + row);
  o.rhz = L.ld(ka->f.rhz + row);
  o.prec = L.ld(ka->f.prec + row);
  o.sw = L.ld(ka->f.sw + row);
  o.lw = L.ld(ka->f.lw + row);
  o.phase = L.ld(ka->f.precphase + row);
  if (ka->f.hour_pstride) o.hour = L.ld(ka->f.hour + row);
  else o.hour = ka->f.hour[k];
  if (FULL) {
    o.tdew = ka->f.tdew ? L.ld(ka->f.tdew + row) : 0.0;
    o.tsurfobs = ka->f.tsurfobs ? L.ld(ka->f.tsurfobs + row) : R4(-9999.9);
    o.depth = ka->f.depth ? L.ld(ka->f.depth + row) : R4(-9999.9);
  } else {
    o.tdew = 0.0;
    o.tsurfobs = R4(-9999.9);
    o.depth = R4(-9999.9);
  }
  return o;
}

/* Output row of time index i: false when the decimation skips it.  UNI: the index is the same in
 * every lane (lock-step kernels) and the row is formed once per step, ahead of the divergent
 * branches, so that it stays in scalar registers (a row computed at each store site is merged by
 * the compiler into a per-lane value). */
template <bool UNI>
__device__ __forceinline__ bool output_row(KernArgs ka, int32_t i, int64_t &row) {
  int32_t r = i - 1;
  const int32_t dec = ka->o.decimate;
  if (dec > 1) {
    if (r % dec != 0) return false;
    r /= dec;
  }
  if (UNI) r = __builtin_amdgcn_readfirstlane(r);
  row = ((int64_t)r - ka->o.row0) * ka->o.t_stride;
  return true;
}

template <bool SCATTER = false, bool A32 = false>
__device__ __forceinline__ void store_outputs(KernArgs ka, int64_t row, int64_t row0, uint32_t lane,
                                              const Scalars &s, bool valid) {
  const double miss = R4(-9999.0); /* src/Initialization.f90:404-411 */
  if (A32 && !SCATTER) {
    const WinOff L(lane, (uint32_t)row + (uint32_t)row0);
    L.st(ka->o.tsurf, valid ? s.tsurf : miss);
    L.st(ka->o.snow, valid ? s.snow : miss);
    L.st(ka->o.water, valid ? s.wat : miss);
    L.st(ka->o.ice, valid ? s.ice : miss);
    L.st(ka->o.deposit, valid ? s.dep : miss);
    L.st(ka->o.ice2, valid ? s.ice2 : miss);
    return;
  }
  if (SCATTER && ka->out_index) { /* column = out_index[slot]: point order whatever the plan order */
    row += (int64_t)ka->out_index[row0 + lane];
    ka->o.tsurf[row] = valid ? s.tsurf : miss;
    ka->o.snow[row] = valid ? s.snow : miss;
    ka->o.water[row] = valid ? s.wat : miss;
    ka->o.ice[row] = valid ? s.ice : miss;
    ka->o.deposit[row] = valid ? s.dep : miss;
    ka->o.ice2[row] = valid ? s.ice2 : miss;
    return;
  }
  row += row0;
  const LaneOff L(lane);
  L.st(ka->o.tsurf + row, valid ? s.tsurf : miss);
  L.st(ka->o.snow + row, valid ? s.snow : miss);
  L.st(ka->o.water + row, valid ? s.wat : miss);
  L.st(ka->o.ice + row, valid ? s.ice : miss);
  L.st(ka->o.deposit + row, valid ? s.dep : miss);
  L.st(ka->o.ice2 + row, valid ? s.ice2 : miss);
}

/* ---- coupling: src/Coupling.f90 -------------------------------------------------
 * Per-point state machine that re-runs the point's coupling window with a scaled
 * short- or long-wave input until the simulated surface temperature at the end of
 * the window matches the last observation (secant / halving / doubling, <= 25
 * replays).  On the device every lane carries its OWN time index: a replay sets it
 * back to the window start (src/Coupling.f90:61-78), so forcing reads and output
 * writes are per-lane gathers/scatters into the (whole-series) windows, and a wave
 * runs until its slowest lane is through.  The saved state of
 * saveDataForCoupling (:172-210) is parked in the HBM state block. */
struct Coupling {
  double tabove, tbelow, radcoeff, rcabove, rcbelow, rcprev, swcof, lwcof, swcorr, lwcorr,
      tend1, lastobs;
  int32_t iter, cs, ce; /* Coupling_iterations, couplingStartI, couplingEndI */
  int32_t msg;          /* bits 3-4 of RS_ST_CPL_FLAGS: what Coupling_control has printed for the point (rs_state.h) */
  bool again, failed, on;
};

/* Coupling_control, src/Coupling.f90:292-481 (called with Coupling_failed == .false.).
 * TsurfAve and LastTsurfObs make a round trip through Kelvin (+273.16, -273.16); the
 * rounding that leaves behind is part of the reference's result. */
__device__ __forceinline__ void coupling_control(Coupling &q, double &tsurf) {
  q.again = false;
  tsurf = tsurf + R4(273.16);
  q.lastobs = q.lastobs + R4(273.16);
  if (q.iter == 0) q.tend1 = tsurf;
  if (q.iter == 25) {
    if (fabs(q.tend1 - q.lastobs) < fabs(tsurf - q.lastobs)) q.again = true;
    q.swcof = R4(1.0); q.lwcof = R4(1.0); q.swcorr = R4(0.0); q.lwcorr = R4(0.0);
    q.radcoeff = R4(1.0);
    q.failed = true;
  } else if (q.lastobs < -100) {
    q.swcof = R4(1.0); q.lwcof = R4(1.0); q.swcorr = R4(0.0); q.lwcorr = R4(0.0);
    q.radcoeff = R4(1.0);
    q.failed = true;
    q.again = true;
  } else if (tsurf < R4(170.0) || tsurf > R4(400.0)) {
    q.swcof = R4(1.0); q.lwcof = R4(1.0); q.swcorr = R4(0.0); q.lwcorr = R4(0.0);
    q.failed = true;
    q.again = true;
    q.radcoeff = R4(1.0);
  } else if (tsurf - q.lastobs > R4(0.1)) {
    if (q.tabove < -100) {
      q.tabove = tsurf;
      q.rcabove = q.radcoeff;
    } else if (q.tabove - q.lastobs > tsurf - q.lastobs) {
      q.tabove = tsurf;
      q.rcabove = q.radcoeff;
    }
    q.again = true;
    if (q.tabove > -100 && q.tbelow > -100) {
      const double da = q.tabove - q.lastobs, db = q.lastobs - q.tbelow;
      q.radcoeff = q.rcabove - da / (da + db) * (q.rcabove - q.rcbelow);
    } else {
      q.radcoeff = R4(0.5) * q.radcoeff;
    }
    if (fabs(q.radcoeff - q.rcprev) < R4(0.00005)) {
      q.tabove = -9999;
      q.tbelow = -9999;
    }
    if (q.radcoeff < R4(0.01)) { /* "coupling coefficient too small, coupling failed" (:400-401) */
      q.msg |= RS_CPL_MSG_SMALL;
      q.radcoeff = R4(1.0);
      q.failed = true;
      q.swcof = R4(1.0); q.lwcof = R4(1.0); q.swcorr = R4(0.0); q.lwcorr = R4(0.0);
    }
    q.rcprev = q.radcoeff;
  } else if (q.lastobs - tsurf > R4(0.1)) {
    if (q.tbelow < -100) {
      q.tbelow = tsurf;
      q.rcbelow = q.radcoeff;
    } else if (q.tbelow - q.lastobs < tsurf - q.lastobs) {
      q.tbelow = tsurf;
      q.rcbelow = q.radcoeff;
    }
    q.again = true;
    if (q.tabove > -100 && q.tbelow > -100) {
      const double da = q.tabove - q.lastobs, db = q.lastobs - q.tbelow;
      q.radcoeff = q.rcabove - da / (da + db) * (q.rcabove - q.rcbelow);
    } else {
      q.radcoeff = R4(2.0) * q.radcoeff;
    }
    if (fabs(q.radcoeff - q.rcprev) < R4(0.00005)) {
      q.tabove = -9999;
      q.tbelow = -9999;
    }
    q.rcprev = q.radcoeff;
  } else {
    if (q.radcoeff > R4(3.0)) { /* "coupling coefficient too big, coupling failed" (:451-452) */
      q.msg |= RS_CPL_MSG_BIG;
      q.failed = true;
      q.radcoeff = R4(1.0);
      q.swcof = R4(1.0); q.lwcof = R4(1.0); q.swcorr = R4(0.0); q.lwcorr = R4(0.0);
    }
    q.swcorr = q.swcof - R4(1.0);
    q.lwcorr = q.lwcof - R4(1.0);
    q.failed = false;
    q.iter = -1;
    q.tabove = R4(-9999.0); q.tbelow = R4(-9999.0);
    q.radcoeff = R4(1.0);
    q.rcabove = R4(-9999.0); q.rcbelow = R4(-9999.0);
    q.rcprev = R4(1.0);
  }
  tsurf = tsurf - R4(273.16);
  q.lastobs = q.lastobs - R4(273.16);
}

__device__ __forceinline__ void load_coupling(const double *st, int64_t np, int64_t p, Coupling &q) {
  q.iter = (int32_t)st[(int64_t)RS_ST_CPL_ITER * np + p];
  const int32_t fl = (int32_t)st[(int64_t)RS_ST_CPL_FLAGS * np + p];
  q.again = fl & 1; q.failed = (fl >> 1) & 1; q.msg = fl & (RS_CPL_MSG_SMALL | RS_CPL_MSG_BIG);
  q.tabove = st[(int64_t)RS_ST_CPL_TABOVE * np + p]; q.tbelow = st[(int64_t)RS_ST_CPL_TBELOW * np + p];
  q.radcoeff = st[(int64_t)RS_ST_CPL_RADCOEFF * np + p];
  q.rcabove = st[(int64_t)RS_ST_CPL_RCABOVE * np + p]; q.rcbelow = st[(int64_t)RS_ST_CPL_RCBELOW * np + p];
  q.rcprev = st[(int64_t)RS_ST_CPL_RCPREV * np + p];
  q.swcof = st[(int64_t)RS_ST_CPL_SWCOF * np + p]; q.lwcof = st[(int64_t)RS_ST_CPL_LWCOF * np + p];
  q.swcorr = st[(int64_t)RS_ST_CPL_SWCORR * np + p]; q.lwcorr = st[(int64_t)RS_ST_CPL_LWCORR * np + p];
  q.tend1 = st[(int64_t)RS_ST_CPL_TEND1 * np + p];
  q.lastobs = st[(int64_t)RS_ST_CPL_LASTOBS * np + p];
}

__device__ __forceinline__ void store_coupling(double *st, int64_t np, int64_t p, const Coupling &q) {
  st[(int64_t)RS_ST_CPL_ITER * np + p] = (double)q.iter;
  const int32_t keep = ((int32_t)st[(int64_t)RS_ST_CPL_FLAGS * np + p]) & 4;
  st[(int64_t)RS_ST_CPL_FLAGS * np + p] = (double)(keep | (q.again ? 1 : 0) | (q.failed ? 2 : 0) | q.msg);
  st[(int64_t)RS_ST_CPL_TABOVE * np + p] = q.tabove; st[(int64_t)RS_ST_CPL_TBELOW * np + p] = q.tbelow;
  st[(int64_t)RS_ST_CPL_RADCOEFF * np + p] = q.radcoeff;
  st[(int64_t)RS_ST_CPL_RCABOVE * np + p] = q.rcabove; st[(int64_t)RS_ST_CPL_RCBELOW * np + p] = q.rcbelow;
  st[(int64_t)RS_ST_CPL_RCPREV * np + p] = q.rcprev;
  st[(int64_t)RS_ST_CPL_SWCOF * np + p] = q.swcof; st[(int64_t)RS_ST_CPL_LWCOF * np + p] = q.lwcof;
  st[(int64_t)RS_ST_CPL_SWCORR * np + p] = q.swcorr; st[(int64_t)RS_ST_CPL_LWCORR * np + p] = q.lwcorr;
  st[(int64_t)RS_ST_CPL_TEND1 * np + p] = q.tend1;
  st[(int64_t)RS_ST_CPL_LASTOBS * np + p] = q.lastobs;
}

/* the stale TmpNw of a replay's first step, parked in the state block (one column per point) */
struct GlobalProfile {
  const double *col;
  int64_t stride;
  __device__ __forceinline__ double get(int j) const { return col[(int64_t)(j - 1) * stride]; }
};

__device__ __forceinline__ Forcing gather_forcing(KernArgs ka, int64_t p, int32_t i, int32_t t0);

/* exp(-(DT*i - DT*couplingEndI)/couplingEffectReduction), src/Coupling.f90:80-88: from the plan's
 * table where the argument is a function of i - couplingEndI alone (rs_consts_dev.h), else here */
__device__ __forceinline__ double cpl_decay(const ConstsAS &c, const MathTab &mt, int32_t i, int32_t ce) {
  const uint32_t d = (uint32_t)(i - ce);
  if (c.cpl_tab && d <= (uint32_t)c.SimLen) return ((const double *)c.cpl_tab)[d];
  return rs_exp(mt, rs_div(-((c.DTSecs * i) - (c.DTSecs * ce)), c.cplReduction));
}

/* Diagnostics (rs_hip_set_diagnostics; ROADSURF_HIP_DIAGNOSTICS=1 on the host paths): what CalcBLCondAndLE prints
 * besides computing - "ERROR : UStar negative" inside the loop and "Max number of BLCond iterations" behind it
 * (src/BoundaryLayer.f90:69-74,98-101).  The loop of the step about to be taken is run once more, out of line, with
 * IEEE division and square root (the bits the bare sequences return, rs_math.hpp), and the first occurrence and the
 * count of either message go to the plan's diagnostics block (rs_state.h RsDiagRow; one lane owns a point: plain
 * read-modify-write).  Only the DIAG instances of the one-point-per-lane kernels with the profile in LDS call it
 * (step_kernel_lds<., true>, step_kernel_sky<true>, step_kernel_coupled<true>) - the launchers send a plan with
 * diagnostics there - so no other kernel carries the call. */
__device__ __attribute__((noinline)) void bl_diagnose(const ConstsAS *cp, const uint64_t *expT, const double *logT,
                                                      const double *K, double tsurf, double tair, double vz,
                                                      double rhz, int32_t hour, int32_t i, double *dg, int64_t np,
                                                      int64_t p) {
  const ConstsAS &c = *cp;
  MathTab mt;
  mt.expT = expT;
  mt.logT = logT;
  mt.K = K;
  mt.S = (rs::MathCoef)rs::c_gl_scoef;
  asm volatile("" : "+s"(mt.S));
  /* SetDayDependendVariables' calm limit (fluxes_pre) */
  const double calmN = c.CalmLimNgt, calmD = c.CalmLimDay;
  const bool night = ((double)hour >= c.NightOn) || ((double)hour <= c.NightOff);
  const double calm = night ? calmN : calmD;
  if (vz < calm) vz = calm;
  BlInv v;
  BlVar x;
  BlAux a;
  bl_setup<true>(c, tsurf, tair, vz, v, x, a);
  const double stab_num = bl_stab_num(c);
  double old = 0.0;
  int j = 1;
  for (; j <= RS_BL_MAXIT; ++j) {
    old = x.BLCond;
    if (rs_dvb<true>(v.vkvz, c.logUstar + x.PSIM) < 0.0) { /* :69-73, with BLCond as the pass before left it */
      const double n = dg[(int64_t)RS_DG_USTAR_N * np + p];
      if (n == 0.0) {
        dg[(int64_t)RS_DG_USTAR_I * np + p] = (double)i;
        dg[(int64_t)RS_DG_USTAR_TAIR * np + p] = tair;
        dg[(int64_t)RS_DG_USTAR_VZ * np + p] = vz;
        dg[(int64_t)RS_DG_USTAR_RHZ * np + p] = rhz;
        dg[(int64_t)RS_DG_USTAR_BL * np + p] = old;
        dg[(int64_t)RS_DG_USTAR_TSURF * np + p] = tsurf;
      }
      dg[(int64_t)RS_DG_USTAR_N * np + p] = n + 1.0;
    }
    if (bl_iteration<true>(c, mt, v, x, j, stab_num)) break;
  }
  if ((fabs(x.BLCond - old) > 10 * R4(0.001)) && (j >= 5)) { /* :98-101; j = MaxIter + 1 behind a DO loop that ran out */
    const double n = dg[(int64_t)RS_DG_MAXIT_N * np + p];
    if (n == 0.0) {
      dg[(int64_t)RS_DG_MAXIT_I * np + p] = (double)i;
      dg[(int64_t)RS_DG_MAXIT_J * np + p] = (double)j;
      dg[(int64_t)RS_DG_MAXIT_OLD * np + p] = old;
      dg[(int64_t)RS_DG_MAXIT_BL * np + p] = x.BLCond;
    }
    dg[(int64_t)RS_DG_MAXIT_N * np + p] = n + 1.0;
  }
}

/* The time loop of runsimulation (examples/example1/src/Simulation.f90:57-115)
 * for one point over absolute indices [t0, t0+nsteps). */
/* SKY: sky view / local horizons (src/ModRadiation.f90, examples/example1/src/Simulation.f90:
 * 151-162) in the lock-step loop; with coupling the general kernel below does it. */
/* CPL: coupling in lock step (src/Coupling.f90), for everything except the replays themselves:
 * the state is saved at the window start (:172-210), the window's first pass runs in the coupling
 * phase, Coupling_control decides at the window end (:98-141,292-481) - a point that has to replay
 * PARKS there (start_coupling_again set, RS_ST_CPL_RESUME = window end + 1) and takes no further
 * step in lock step until the replay rounds (step_kernel_coupled over the compacted list of parked
 * points, rs_hip_cpl_replay) have cleared it - and behind the window the radiation corrections
 * decay (:80-88).  A lane steps index i only if i == its RS_ST_CPL_RESUME, so launches may
 * overlap in time: points that are ahead wait for the others. */
/* REPLAY (with CPL): one replay ROUND in lock step, for the points of the compacted list that asked
 * for another replay (start_coupling_again): the launch covers [first window start, last window
 * end + 1]; a lane waits for its window start, rewinds there - CheckValues of the index behind its
 * window end, as the reference's loop does before CouplingOperations1 takes it back
 * (examples/example1/src/Simulation.f90:62-71), then uploadDataForCoupling (src/Coupling.f90:
 * 213-255) and the radiation coefficient (:61-78) -, runs its window in the coupling phase and lets
 * Coupling_control decide again at the end.  `point` is the lane's slot: the list breaks the tie
 * between thread and point, so accesses are base[point] with the row base on the scalar unit. */
template <bool FULL, class Prof, bool SKY = false, bool SCORE = true, bool CPL = false,
          bool REPLAY = false, bool A32 = false, bool DIAG = false>
__device__ __forceinline__ void time_loop(const MathTab &mt, Prof &T, Scalars &s, int32_t &score,
                                          uint32_t point = 0u) {
  static_assert(!A32 || (!SKY && !CPL), "32-bit window offsets: the plain lock-step kernels only");
  static_assert(FULL || !SKY, "sky view belongs to the FULL feature set");
  static_assert(FULL || !CPL, "coupling belongs to the FULL feature set");
  static_assert(CPL || !REPLAY, "replays belong to coupling");
  KernArgs ka = kernargs();
  const uint32_t lane = REPLAY ? point : threadIdx.x;
  const int64_t row0 = REPLAY ? 0 : (int64_t)blockIdx.x * kBlock; /* first point of this workgroup */
  const int32_t nsteps = ka->nsteps, t0 = ka->t0;
  const double tbot = (ka->pp.tbottom + row0)[lane];
  double skyv = R4(1.0), sinlat = 0, coslat = 0, lonrad = 0, coslon = 1.0, sinlon = 0;
  bool sky_on = false;
  uint32_t hcol = 0; /* the point's column of the local-horizon table (RsPointParams::horizon_index) */
  if (SKY) {
    skyv = (ka->pp.sky_view + row0)[lane];
    sky_on = (skyv < R4(1.0) && skyv > R4(-0.01));
    if (sky_on) {
      sinlat = (ka->pp.sin_lat + row0)[lane];
      coslat = (ka->pp.cos_lat + row0)[lane];
      lonrad = (ka->pp.lon_rad + row0)[lane];
      coslon = ::cos(lonrad); /* once per launch: the addition theorem's point half (sky_view_radiation) */
      sinlon = ::sin(lonrad);
    }
    hcol = ka->pp.horizon_index ? (uint32_t)(ka->pp.horizon_index + row0)[lane] : (uint32_t)row0 + lane;
  }
  int32_t initlen = 0;
  bool relax = false;
  /* RelaxationOperations (src/Relaxation.f90:34-43) only ever uses target - anchor: the three
   * differences are what the loop carries (the targets are read again at the one index that sets
   * the anchors, the anchors go to the state block there and then) */
  double relax_dt = 0, relax_dv = 0, relax_dr = 0;
  auto relax_targets = [&](double &tairR, double &vzR, double &rhR) {
    /* setInputParam, src/InputOutput.f90:19-26: targets pass through REAL(4) */
    tairR = (double)(float)(ka->pp.tair_relax + row0)[lane];
    vzR = (double)(float)(ka->pp.vz_relax + row0)[lane];
    rhR = (double)(float)(ka->pp.rh_relax + row0)[lane];
  };
  if (FULL) {
    initlen = ka->pp.initlen ? (ka->pp.initlen + row0)[lane] : 0;
    if (consts_of(ka).use_relaxation && ka->pp.tair_relax) {
      double tairR, vzR, rhR;
      relax_targets(tairR, vzR, rhR);
      relax = !(tairR < R4(-100.0) || tairR > R4(100.0) || vzR < R4(0.0) || vzR > R4(100.0) ||
                rhR < R4(0.0) || rhR > 110);
      relax_dt = tairR - s.tair_end;
      relax_dv = vzR - s.vz_end;
      relax_dr = rhR - s.rh_end;
    }
  }

  /* settings%simulation_failed (src/InputOutput.f90:66): sticky; the state block keeps the index
   * it was raised at (rs_hip_first_failed_index) */
  auto fail_at = [&](int32_t idx) {
    s.failed = true;
    ka->state[(int64_t)RS_ST_FAILED * ka->np_pad + row0 + lane] = (double)idx;
  };
  /* coupling, lock-step part: what a lane needs every step stays in registers, the rest of
   * CouplingVariables lives in the state block and is touched at the two events only */
  bool cpl_on = false, parked = false;
  int32_t cpl_cs = -99, cpl_ce = -99, next_i = t0;
  double cpl_lastobs = 0.0, cpl_swcorr = 0.0, cpl_lwcorr = 0.0;
  if (CPL) {
    double *st = ka->state;
    const int64_t np = ka->np_pad, p = row0 + lane;
    const int32_t cidx = ka->pp.coupling_index ? (ka->pp.coupling_index + row0)[lane] : 0;
    /* setInputParam + initCouplingTimes, src/InputOutput.f90:30-36, src/Coupling.f90:486-534 */
    cpl_on = consts_of(ka).use_coupling && ka->pp.coupling_index &&
             !((ka->pp.coupling_tsurf + row0)[lane] < -100 || cidx < 1);
    if (cpl_on) {
      cpl_ce = cidx;
      cpl_cs = ((double)cidx <= consts_of(ka).cplLenR) ? 1 : cidx - consts_of(ka).cplLenI;
    }
    cpl_lastobs = st[(int64_t)RS_ST_CPL_LASTOBS * np + p];
    cpl_swcorr = st[(int64_t)RS_ST_CPL_SWCORR * np + p];
    cpl_lwcorr = st[(int64_t)RS_ST_CPL_LWCORR * np + p];
    parked = (((int32_t)st[(int64_t)RS_ST_CPL_FLAGS * np + p]) & 1) != 0;
    next_i = (int32_t)st[(int64_t)RS_ST_CPL_RESUME * np + p];
    if (REPLAY) { /* a listed point: parked behind its window, wanting it again */
      parked = false;
      next_i = cpl_cs;
    }
  }
  /* REPLAY: SWRadCof / LWRadCof of the window being replayed, and "TmpNw is the stale profile" */
  double r_swcof = R4(1.0), r_lwcof = R4(1.0);
  bool stale_now = false;
  /* A failed point's rows read -9999.0 from behind the index that failed it (the reference's loop has
   * exited: src/Initialization.f90:404-411).  Without coupling they are written HERE - the rest of the
   * window when the failure is detected, the whole window for a point that arrives failed - and not by
   * a second store site in the time loop: the compiler merges two store sites of a loop into one and
   * pays for it with ~25 register moves on every step of every lane. */
  auto blank_rows = [&](int32_t i_from) {
    for (int32_t ii = i_from; ii < t0 + nsteps; ++ii) {
      int64_t r;
      if (output_row<false>(ka, ii, r)) store_outputs<false, false>(ka, r, row0, lane, s, false);
    }
  };
  if (!CPL && s.failed) blank_rows(t0);
  Forcing nxt = load_forcing<FULL, A32>(ka, row0, lane, 0);
  for (int32_t kv = 0; kv < nsteps; ++kv) {
    asm volatile("" : "+s"(ka));
    const ConstsAS &c = consts_of(ka);
    /* the loop counter is wave-uniform, but lanes that `continue` (failed, parked) make the
     * compiler keep it in a vector register, and with it every row offset: 64-bit vector
     * multiplies and two 64-bit vector adds per load and store.  Read back as a scalar, the row
     * arithmetic runs on the scalar unit and the accesses are saddr + 32-bit lane offset */
    const int32_t k = __builtin_amdgcn_readfirstlane(kv);
    const int32_t i = t0 + k;
    const Forcing f = nxt;
    int64_t orow = 0;
    const bool owrite = output_row<true>(ka, i, orow);

    if (CPL && (parked || i != next_i)) { /* parked behind its window, or ahead of this launch */
      if (k + 1 < nsteps) nxt = load_forcing<FULL, A32>(ka, row0, lane, k + 1);
      continue;
    }
    if (CPL) next_i = i + 1;
    if (s.failed) { /* loop has exited in the reference: outputs stay -9999.0 */
      /* (no forcing is fetched for a failed point: it never steps again, and a second fetch site would
       * be merged with the one in the middle of the step at the price of moves at the loop's end) */
      /* a point that fails in the middle of a replay keeps, up to its window end, what the earlier
       * passes saved there (src/InputOutput.f90:151-165 only ever overwrites) */
      if (CPL && owrite && (!REPLAY || i > cpl_ce)) store_outputs<CPL, A32>(ka, orow, row0, lane, s, false);
      continue;
    }
    double tair = f.tair, vz = f.vz, rhz = f.rhz;
    /* TmpNw(1:2) as CalcHCapHCond will see them (src/BalanceModel.f90:215): the values the last step
     * left - Tmp = TmpNw at its end (:60-62) - whatever observation forcing does to Tmp(1:2) below
     * (src/InputOutput.f90:122-124 does not touch TmpNw) */
    s.tnw1 = T.get(1);
    s.tnw2 = T.get(2);
    /* src/Initialization.f90:121-123: VZ(1) is raised to 0.4 in the input array */
    if (i == 1 && vz < R4(0.4)) vz = R4(0.4);
    const double prec_ts = RS_DIVC(f.prec, 3600.0, r_3600) * c.DTSecs; /* src/InputOutput.f90:111,186 */

    double sw_dir = 0.0, lw_net = 0.0;
    if (SKY) {
      const int64_t row = (int64_t)k * ka->f.t_stride + row0;
      const LaneOff L(lane);
      sw_dir = L.ld(ka->f.sw_dir + row);
      lw_net = L.ld(ka->f.lw_net + row);
    }
    CouplingInputs cp;
    if (i < c.SimLen) {
      if (REPLAY && i == cpl_cs) {
        /* the rewind: the reference's loop is at the index behind the window end when
         * CouplingOperations1 takes it back, and that is the index CheckValues has just seen -
         * with the surface temperature of the end of the window */
        const Forcing g = gather_forcing(ka, row0 + lane, cpl_ce + 1, t0);
        if (check_values(c, g, s.tsurf, FULL && ka->f.tdew != nullptr)) fail_at(cpl_ce + 1);
      } else {
        Forcing chk = f;
        chk.vz = vz;
        if (check_values(c, chk, s.tsurf, FULL && ka->f.tdew != nullptr)) fail_at(i);
      }
      if (SKY) {
        if (REPLAY && i == cpl_cs) {
          /* the rewind: CheckValues has just seen the index behind the window end (above) */
          if (sky_on) {
            const int64_t off = (int64_t)(cpl_ce + 1 - t0) * ka->f.t_stride + row0 + lane;
            const double sd = ka->f.sw_dir[off], ln = ka->f.lw_net[off];
            if (sd < R4(-0.1) || sd > R4(4000.0) || ln < R4(-1000.0) || ln > R4(1000.0)) fail_at(cpl_ce + 1);
          }
        } else if (sky_on && (sw_dir < R4(-0.1) || sw_dir > R4(4000.0) || lw_net < R4(-1000.0) ||
                              lw_net > R4(1000.0)))
          fail_at(i);                           /* src/InputOutput.f90:68-74 */
        if (sw_dir > f.sw) sw_dir = f.sw;       /* :75-77 */
      }
      if (CPL && cpl_on) {
        /* CouplingOperations1, src/Coupling.f90:10-96, first pass of the window */
        cp.in_phase = (i >= cpl_cs && i <= cpl_ce);
        /* inCouplingPhase is set from the loop index BEFORE a rewind takes it back (:22-27 against
         * :61-66): the step at the window start of a replay runs outside the coupling phase */
        if (REPLAY && i == cpl_cs) cp.in_phase = false;
        if (REPLAY && i == cpl_cs) {
          /* uploadDataForCoupling :213-255: SrfIcemms, Q2Melt, T4Melt and TmpNw are NOT restored;
           * TmpNw keeps the end-of-window profile for the first step */
          double *st = ka->state;
          const int64_t np = ka->np_pad, p = row0 + lane;
          s.tsurf = st[(int64_t)RS_ST_CPL_SAVE_TSURF * np + p];
          s.wat = st[(int64_t)RS_ST_CPL_SAVE_WAT * np + p];
          s.ice2 = st[(int64_t)RS_ST_CPL_SAVE_ICE2 * np + p];
          s.dep = st[(int64_t)RS_ST_CPL_SAVE_DEP * np + p];
          s.snow = st[(int64_t)RS_ST_CPL_SAVE_SNOW * np + p];
          s.albedo = st[(int64_t)RS_ST_CPL_SAVE_ALBEDO * np + p];
          const int32_t fl = (int32_t)st[(int64_t)RS_ST_CPL_FLAGS * np + p];
          s.verycold = (fl & 4) != 0;
          st[(int64_t)RS_ST_CPL_FLAGS * np + p] = (double)(fl & ~1); /* start_coupling_again = .false. */
          const int N = T.nlayers();
          for (int j = 1; j <= N; ++j) {
            st[(int64_t)(RS_ST_CPL_STALE_TMP0 + j - 1) * np + p] = T.get(j);
            T.set(j, st[(int64_t)(RS_ST_CPL_SAVE_TMP0 + j - 1) * np + p]);
          }
          stale_now = true;
          /* short-wave scaling by day, long-wave by night - and always with sky view (:68-76) */
          const double radcoeff = st[(int64_t)RS_ST_CPL_RADCOEFF * np + p];
          if (f.sw > f.lw && !(SKY && sky_on)) {
            r_swcof = radcoeff;
            r_lwcof = R4(1.0);
          } else {
            r_swcof = R4(1.0);
            r_lwcof = radcoeff;
          }
          st[(int64_t)RS_ST_CPL_SWCOF * np + p] = r_swcof; /* Coupling_control reads them at the end */
          st[(int64_t)RS_ST_CPL_LWCOF * np + p] = r_lwcof;
        }
        if (!REPLAY && i == cpl_cs) { /* Coupling_iterations == 0 here: saveDataForCoupling :172-210 */
          double *st = ka->state;
          const int64_t np = ka->np_pad, p = row0 + lane;
          st[(int64_t)RS_ST_CPL_SAVE_TSURF * np + p] = s.tsurf;
          st[(int64_t)RS_ST_CPL_SAVE_WAT * np + p] = s.wat;
          st[(int64_t)RS_ST_CPL_SAVE_ICE2 * np + p] = s.ice2;
          st[(int64_t)RS_ST_CPL_SAVE_DEP * np + p] = s.dep;
          st[(int64_t)RS_ST_CPL_SAVE_SNOW * np + p] = s.snow;
          st[(int64_t)RS_ST_CPL_SAVE_ALBEDO * np + p] = s.albedo;
          const int32_t fl = ((int32_t)st[(int64_t)RS_ST_CPL_FLAGS * np + p]) & (3 | RS_CPL_MSG_SMALL | RS_CPL_MSG_BIG);
          st[(int64_t)RS_ST_CPL_FLAGS * np + p] = (double)(fl | (s.verycold ? 4 : 0));
          const int N = T.nlayers();
          for (int j = 1; j <= N; ++j) st[(int64_t)(RS_ST_CPL_SAVE_TMP0 + j - 1) * np + p] = T.get(j);
          /* SW/LWRadCof = 1, SW/LW_correction = 0: what the registers hold before the window end */
        }
        if (i > cpl_ce) {
          const double e = cpl_decay(c, mt, i, cpl_ce);
          cp.sw_cof = R4(1.0) + cpl_swcorr * e;
          cp.lw_cof = R4(1.0) + cpl_lwcorr * e;
        }
        if (REPLAY && i >= cpl_cs && i <= cpl_ce) {
          cp.sw_cof = r_swcof;
          cp.lw_cof = r_lwcof;
        }
        if (cp.in_phase) {
          /* snowIceCheck :259-289 */
          if (cpl_lastobs > c.TLimMeltSnow && s.snow > R4(0.00)) { s.wat = s.wat + s.snow; s.snow = R4(0.00); }
          if (cpl_lastobs > c.TLimMeltIce && s.ice > R4(0.00)) { s.wat = s.wat + s.ice; s.ice = R4(0.00); }
          if (cpl_lastobs > c.TLimMeltIce && s.ice2 > R4(0.00)) s.ice2 = R4(0.00);
          if (cpl_lastobs > c.TLimMeltDep && s.dep > R4(0.00)) { s.wat = s.wat + s.dep; s.dep = R4(0.00); }
        }
      }
      if (FULL) {
        /* SetCurrentValues obs forcing, src/InputOutput.f90:116-148 */
        if ((i <= initlen || c.force_tsurf) && f.tsurfobs > R4(-100.0) &&
            (!CPL || !cpl_on || i < cpl_cs)) {
          T.set(1, f.tsurfobs);
          T.set(2, f.tsurfobs);
          const double depth = (c.tsurfOutputDepth >= R4(0.0)) ? c.tsurfOutputDepth : f.depth;
          s.tsurf = surface_temperature(c, T, tbot, depth);
        }
        /* RelaxationOperations, src/Relaxation.f90:10-47 */
        if (relax) {
          if (i == initlen) { /* the anchors: once per point */
            double tairR, vzR, rhR;
            relax_targets(tairR, vzR, rhR);
            relax_dt = tairR - tair;
            relax_dv = vzR - vz;
            relax_dr = rhR - rhz;
            double *st = ka->state;
            const int64_t np = ka->np_pad, p = row0 + lane;
            st[(int64_t)RS_ST_TAIR_END * np + p] = tair;
            st[(int64_t)RS_ST_VZ_END * np + p] = vz;
            st[(int64_t)RS_ST_RH_END * np + p] = rhz;
          }
          if (i > initlen) {
            /* exp(-(DT*i - DT*InitLenI)/14400): for an integral time step the argument is a function
             * of i - InitLenI alone and the plan holds the table (rs_consts_dev.h, host libm = the
             * bits rs_exp returns); otherwise evaluated here */
            double e;
            const uint32_t d = (uint32_t)(i - initlen);
            if (c.relax_tab && d <= (uint32_t)c.SimLen) {
              e = ((const double *)c.relax_tab)[d];
            } else {
              const double den = (double)(4.f * 3600.f);
              e = rs_exp(mt, rs_div(-((c.DTSecs * i) - (c.DTSecs * initlen)), den));
            }
            tair = tair - relax_dt * e;
            vz = vz - relax_dv * e;
            rhz = rhz - relax_dr * e;
            if (rhz > R4(100.)) rhz = R4(100.0);
          }
        }
      }
    } else {
      /* lastValues, src/InputOutput.f90:169-198: no checks, no obs forcing, no
       * relaxation; the pre-step surface temperature uses depth(SimLen) only */
      if (FULL) s.tsurf = surface_temperature(c, T, tbot, f.depth);
      /* coupling%inCouplingPhase and SW/LWRadCof keep their last values: those of index SimLen-1 */
      if (CPL && cpl_on) {
        const int32_t j = c.SimLen - 1;
        cp.in_phase = (j >= cpl_cs && j <= cpl_ce);
        if (j > cpl_ce) {
          const double e = cpl_decay(c, mt, j, cpl_ce);
          cp.sw_cof = R4(1.0) + cpl_swcorr * e;
          cp.lw_cof = R4(1.0) + cpl_lwcorr * e;
        } else if (j >= cpl_cs) { /* the window reaches the end of the series: as the last pass left them */
          cp.sw_cof = ka->state[(int64_t)RS_ST_CPL_SWCOF * ka->np_pad + row0 + lane];
          cp.lw_cof = ka->state[(int64_t)RS_ST_CPL_LWCOF * ka->np_pad + row0 + lane];
        }
      }
    }
    if (CPL) cp.last_tsurf_obs = cpl_lastobs;
    double sw_in = f.sw, lw_in = f.lw;
    if (SKY && sky_on) {
      /* the reference runs this between PrecipitationToStorage and BalanceModelOneStep
       * (Simulation.f90:151-162); the two do not share data, so the order is free */
      if (!sky_view_radiation(ka->f.sun + (int64_t)k * RS_SUN_COLS, sinlat, coslat, lonrad, coslon, sinlon, skyv,
                              ka->pp.albedo_surroundings,
                              ka->pp.horizons ? ka->pp.horizons + (ka->pp.horizons_by_point ? (int64_t)hcol * 360 : (int64_t)hcol) : nullptr,
                              ka->pp.horizons_by_point ? (int64_t)1 : ka->np_pad,
                              sw_in, sw_dir, lw_in, lw_net))
        fail_at(i); /* the reference would `stop` the process here */
    }
    if (SKY && ka->wb.sw_dir) {
      /* the caller's arrays as the reference leaves them: SW_dir clamped by CheckValues
       * (src/InputOutput.f90:75-77), SW / SW_dir / LW edited by ModRadiationBySurroundings
       * (src/ModRadiation.f90:57-71) */
      const int64_t wrow = (int64_t)k * ka->wb.t_stride + row0;
      (ka->wb.sw_dir + wrow)[lane] = sw_dir;
      if (sky_on) {
        (ka->wb.sw + wrow)[lane] = sw_in;
        (ka->wb.lw + wrow)[lane] = lw_in;
      }
    }
    if (DIAG) /* (instances of their own, step_kernel_lds<., true> and step_kernel_sky<true>: see bl_diagnose) */
      bl_diagnose(&c, mt.expT, mt.logT, mt.K, s.tsurf, tair, vz, rhz, f.hour, i, ka->diag, ka->np_pad, row0 + lane);
    const Fluxes fx =
        model_step_fluxes<SCORE>(c, mt, s, tair, vz, rhz, prec_ts, sw_in, lw_in, f.phase, f.hour, cp);
    /* scheduling hint (bl_score_key): extra passes of this launch; bit 30 of the counter = the
     * point was in the unstable regime at some index of the launch's last RS_REGIME_WINDOW */
    if (SCORE) {
      score += (fx.trips & 63) - 5;
      if ((fx.trips & 64) && k >= nsteps - RS_REGIME_WINDOW) score |= 1 << 30;
    }
    /* next index's forcing: issued here, half a step before its first use, so the
     * HBM latency hides under the ground/storage half without holding 14 VGPRs
     * across the boundary-layer iteration */
    if (k + 1 < nsteps) nxt = load_forcing<FULL, A32>(ka, row0, lane, k + 1);
    if (REPLAY) {
      /* first step after the restore: CalcHCapHCond sees the pre-restore TmpNw in every layer
       * (observation forcing cannot follow a restore, so layers 1-2 are stale too) */
      const GlobalProfile Tstale{ka->state + (int64_t)RS_ST_CPL_STALE_TMP0 * ka->np_pad + row0 + lane,
                                 ka->np_pad};
      model_step_ground<Prof, GlobalProfile>(c, s, T, tbot, tair, fx, f.depth, cp,
                                             stale_now ? &Tstale : nullptr);
      stale_now = false;
    } else {
      model_step_ground<Prof, Prof, FULL>(c, s, T, tbot, tair, fx, f.depth, cp);
    }
    if (owrite) store_outputs<CPL, A32>(ka, orow, row0, lane, s, true);
    if (!CPL && s.failed) blank_rows(i + 1); /* failed at this index: its own row is saved, the rest is not */
    if (CPL && cpl_on && i < c.SimLen && i == cpl_ce) {
      /* CheckEndCoupling + CouplingOperations2, src/Coupling.f90:98-141 (Coupling_failed is
       * .false. before the first decision) */
      double *st = ka->state;
      const int64_t np = ka->np_pad, p = row0 + lane;
      Coupling q;
      load_coupling(st, np, p, q);
      q.cs = cpl_cs; q.ce = cpl_ce; q.on = true;
      if (!q.failed) {
        if (q.iter == 0) q.tend1 = s.tsurf;
        coupling_control(q, s.tsurf);
        q.iter = q.iter + 1;
        /* a point that failed CheckValues at this very index still ran Coupling_control
         * (CheckEndCoupling does not look at simulation_failed), but its loop exits before any rewind:
         * it must not park - the rows behind its window stay -9999.0 - and must not be listed */
        if (s.failed) q.again = false;
        store_coupling(st, np, p, q);
        cpl_swcorr = q.swcorr;
        cpl_lwcorr = q.lwcorr;
        cpl_lastobs = q.lastobs; /* went to Kelvin and back in Coupling_control */
        parked = q.again; /* replays its window in the rounds; RS_ST_CPL_RESUME = i + 1 */
      }
    }
  }
  if (CPL) ka->state[(int64_t)RS_ST_CPL_RESUME * ka->np_pad + row0 + lane] = (double)next_i;
}

/* Per-lane gather of the forcing of absolute index i (window row i - t0). */
__device__ __forceinline__ Forcing gather_forcing(KernArgs ka, int64_t p, int32_t i, int32_t t0) {
  Forcing o;
  const int64_t off = (int64_t)(i - t0) * ka->f.t_stride + p;
  o.tair = ka->f.tair[off]; o.vz = ka->f.vz[off]; o.rhz = ka->f.rhz[off];
  o.prec = ka->f.prec[off]; o.sw = ka->f.sw[off]; o.lw = ka->f.lw[off];
  o.phase = ka->f.precphase[off];
  o.hour = ka->f.hour_pstride ? ka->f.hour[off] : ka->f.hour[i - t0];
  o.tdew = ka->f.tdew ? ka->f.tdew[off] : 0.0;
  o.tsurfobs = ka->f.tsurfobs ? ka->f.tsurfobs[off] : R4(-9999.9);
  o.depth = ka->f.depth ? ka->f.depth[off] : R4(-9999.9);
  return o;
}

/* runsimulation's loop with coupling (examples/example1/src/Simulation.f90:57-115):
 * CheckValues -> CouplingOperations1 (may rewind i) -> SetCurrentValues -> relaxation ->
 * roadModelOneStep -> SaveOutput -> CheckEndCoupling. */
template <class Prof, bool DIAG = false>
__device__ __forceinline__ void time_loop_coupled(const MathTab &mt, Prof &T, Scalars &s,
                                                  Coupling &q, double *st, int64_t np, int64_t p) {
  KernArgs ka = kernargs();
  const uint32_t lane = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * kBlock;
  const ConstsAS &c = consts_of(ka);
  const int N = T.nlayers();
  const int32_t t0 = ka->t0, tend = ka->t0 + ka->nsteps;
  const double tbot = ka->pp.tbottom[p];
  const int32_t initlen = ka->pp.initlen ? ka->pp.initlen[p] : 0;
  bool relax = false;
  double tairR = 0, vzR = 0, rhR = 0;
  if (c.use_relaxation && ka->pp.tair_relax) {
    tairR = (double)(float)ka->pp.tair_relax[p];
    vzR = (double)(float)ka->pp.vz_relax[p];
    rhR = (double)(float)ka->pp.rh_relax[p];
    relax = !(tairR < R4(-100.0) || tairR > R4(100.0) || vzR < R4(0.0) || vzR > R4(100.0) ||
              rhR < R4(0.0) || rhR > 110);
  }
  /* setInputParam + initCouplingTimes, src/InputOutput.f90:30-36, src/Coupling.f90:486-534 */
  const int32_t cidx = ka->pp.coupling_index ? ka->pp.coupling_index[p] : 0;
  q.on = c.use_coupling && ka->pp.coupling_index &&
         !(ka->pp.coupling_tsurf[p] < -100 || cidx < 1);
  q.cs = -99; q.ce = -99;
  if (q.on) {
    q.ce = cidx;
    q.cs = ((double)cidx <= c.cplLenR) ? 1 : cidx - c.cplLenI;
  }
  (void)lane; (void)row0;
  /* sky view, examples/example1/src/Simulation.f90:154-156 */
  double skyv = R4(1.0), sinlat = 0, coslat = 0, lonrad = 0, coslon = 1.0, sinlon = 0;
  bool sky_on = false;
  if (ka->pp.sky_view) {
    skyv = ka->pp.sky_view[p];
    sky_on = (skyv < R4(1.0) && skyv > R4(-0.01));
    if (sky_on) {
      sinlat = ka->pp.sin_lat[p];
      coslat = ka->pp.cos_lat[p];
      lonrad = ka->pp.lon_rad[p];
      coslon = ::cos(lonrad);
      sinlon = ::sin(lonrad);
    }
  }

  auto fail_at = [&](int32_t idx) {
    s.failed = true;
    st[(int64_t)RS_ST_FAILED * np + p] = (double)idx;
  };
  /* rounds (rs_hip_step): the point resumes where it stopped; a parked point (resume = window
   * end + 1, start_coupling_again set) rewinds to its window start in the first iteration */
  const int32_t resume = (int32_t)st[(int64_t)RS_ST_CPL_RESUME * np + p];
  int32_t i = resume > t0 ? resume : t0;
  int32_t written_hi = i - 1; /* highest index the point has saved an output for */
  bool stale_all = false; /* first step after a restore: TmpNw is the pre-restore profile */
  /* the forcing of index i+1 is fetched in the middle of step i (a replay that rewinds fetches its
   * row again): with two waves per SIMD a load issued at its point of use would stall every step */
  Forcing nxt;
  int32_t nxt_i = -1;
  if (i < tend && !s.failed) {
    nxt = gather_forcing(ka, p, i, t0);
    nxt_i = i;
  }
  while (i < tend) {
    if (s.failed) {
      /* the reference's loop has exited: outputs it never saved stay -9999.0 - but a point
       * that fails in the middle of a coupling replay keeps, beyond the failure, what the
       * EARLIER passes saved there (src/InputOutput.f90:151-165 only ever overwrites) */
      int64_t orow;
      if (i > written_hi && output_row<false>(ka, i, orow)) store_outputs<true>(ka, orow, p, 0u, s, false);
      ++i;
      continue;
    }
    Forcing f = (nxt_i == i) ? nxt : gather_forcing(ka, p, i, t0);
    if (i == 1 && f.vz < R4(0.4)) f.vz = R4(0.4);
    CouplingInputs cp;
    double sw_dir = 0.0, lw_net = 0.0;
    if (ka->f.sw_dir) {
      const int64_t off = (int64_t)(i - t0) * ka->f.t_stride + p;
      sw_dir = ka->f.sw_dir[off];
      lw_net = ka->f.lw_net[off];
    }
    if (i < c.SimLen) {
      if (check_values(c, f, s.tsurf, ka->f.tdew != nullptr)) fail_at(i);
      if (sky_on && (sw_dir < R4(-0.1) || sw_dir > R4(4000.0) || lw_net < R4(-1000.0) ||
                     lw_net > R4(1000.0)))
        fail_at(i); /* src/InputOutput.f90:68-74 */
      if (sw_dir > f.sw) sw_dir = f.sw; /* :75-77 */
      if (q.on) {
        /* CouplingOperations1, src/Coupling.f90:10-96 */
        bool in_phase = (i >= q.cs && i <= q.ce);
        if (i == q.cs && q.iter == 0) {
          /* saveDataForCoupling :172-210 */
          st[(int64_t)RS_ST_CPL_SAVE_TSURF * np + p] = s.tsurf;
          st[(int64_t)RS_ST_CPL_SAVE_WAT * np + p] = s.wat;
          st[(int64_t)RS_ST_CPL_SAVE_ICE2 * np + p] = s.ice2;
          st[(int64_t)RS_ST_CPL_SAVE_DEP * np + p] = s.dep;
          st[(int64_t)RS_ST_CPL_SAVE_SNOW * np + p] = s.snow;
          st[(int64_t)RS_ST_CPL_SAVE_ALBEDO * np + p] = s.albedo;
          const int32_t fl = ((int32_t)st[(int64_t)RS_ST_CPL_FLAGS * np + p]) & (3 | RS_CPL_MSG_SMALL | RS_CPL_MSG_BIG);
          st[(int64_t)RS_ST_CPL_FLAGS * np + p] = (double)(fl | (s.verycold ? 4 : 0));
          for (int j = 1; j <= N; ++j) st[(int64_t)(RS_ST_CPL_SAVE_TMP0 + j - 1) * np + p] = T.get(j);
          q.swcof = R4(1.0); q.lwcof = R4(1.0); q.swcorr = R4(0.0); q.lwcorr = R4(0.0);
        }
        if (q.again) {
          /* uploadDataForCoupling :213-255: back to the window start; SrfIcemms, Q2Melt,
           * T4Melt and TmpNw are NOT restored */
          i = q.cs;
          s.tsurf = st[(int64_t)RS_ST_CPL_SAVE_TSURF * np + p];
          s.wat = st[(int64_t)RS_ST_CPL_SAVE_WAT * np + p];
          s.ice2 = st[(int64_t)RS_ST_CPL_SAVE_ICE2 * np + p];
          s.dep = st[(int64_t)RS_ST_CPL_SAVE_DEP * np + p];
          s.snow = st[(int64_t)RS_ST_CPL_SAVE_SNOW * np + p];
          s.albedo = st[(int64_t)RS_ST_CPL_SAVE_ALBEDO * np + p];
          s.verycold = (((int32_t)st[(int64_t)RS_ST_CPL_FLAGS * np + p]) & 4) != 0;
          for (int j = 1; j <= N; ++j) {
            /* TmpNw keeps the end-of-window profile */
            st[(int64_t)(RS_ST_CPL_STALE_TMP0 + j - 1) * np + p] = T.get(j);
            T.set(j, st[(int64_t)(RS_ST_CPL_SAVE_TMP0 + j - 1) * np + p]);
          }
          stale_all = true;
          q.again = false;
          f = gather_forcing(ka, p, i, t0);
          if (i == 1 && f.vz < R4(0.4)) f.vz = R4(0.4);
          if (ka->f.sw_dir) {
            /* the restored window holds the arrays as CheckValues left them in the first
             * pass: SW_dir already clamped to SW (src/Coupling.f90:204-208,249-253) */
            const int64_t off = (int64_t)(i - t0) * ka->f.t_stride + p;
            sw_dir = ka->f.sw_dir[off];
            lw_net = ka->f.lw_net[off];
            if (sw_dir > f.sw) sw_dir = f.sw;
          }
          /* short-wave scaling only without sky view (:68-76) */
          if (f.sw > f.lw && !sky_on) {
            q.swcof = q.radcoeff;
            q.lwcof = R4(1.0);
          } else {
            q.swcof = R4(1.0);
            q.lwcof = q.radcoeff;
          }
        }
        if (i > q.ce) {
          const double e = cpl_decay(c, mt, i, q.ce);
          q.swcof = R4(1.0) + q.swcorr * e;
          q.lwcof = R4(1.0) + q.lwcorr * e;
        }
        if (in_phase) {
          /* snowIceCheck :259-289 */
          if (q.lastobs > c.TLimMeltSnow && s.snow > R4(0.00)) { s.wat = s.wat + s.snow; s.snow = R4(0.00); }
          if (q.lastobs > c.TLimMeltIce && s.ice > R4(0.00)) { s.wat = s.wat + s.ice; s.ice = R4(0.00); }
          if (q.lastobs > c.TLimMeltIce && s.ice2 > R4(0.00)) s.ice2 = R4(0.00);
          if (q.lastobs > c.TLimMeltDep && s.dep > R4(0.00)) { s.wat = s.wat + s.dep; s.dep = R4(0.00); }
        }
        cp.in_phase = in_phase;
      }
      /* SetCurrentValues obs forcing, src/InputOutput.f90:116-148 */
      if ((i <= initlen || c.force_tsurf) && f.tsurfobs > R4(-100.0) && (!q.on || i < q.cs)) {
        T.set(1, f.tsurfobs);
        T.set(2, f.tsurfobs);
        const double depth = (c.tsurfOutputDepth >= R4(0.0)) ? c.tsurfOutputDepth : f.depth;
        s.tsurf = surface_temperature(c, T, tbot, depth);
      }
    } else {
      /* lastValues, src/InputOutput.f90:169-198 */
      s.tsurf = surface_temperature(c, T, tbot, f.depth);
      cp.in_phase = false; /* coupling%inCouplingPhase keeps its last value: index SimLen-1 */
      if (q.on) cp.in_phase = (c.SimLen - 1 >= q.cs && c.SimLen - 1 <= q.ce);
    }
    double tair = f.tair, vz = f.vz, rhz = f.rhz;
    const double prec_ts = RS_DIVC(f.prec, 3600.0, r_3600) * c.DTSecs;
    if (i < c.SimLen && relax) {
      if (i == initlen) { s.tair_end = tair; s.vz_end = vz; s.rh_end = rhz; }
      if (i > initlen) {
        const double den = (double)(4.f * 3600.f);
        const double e = rs_exp(mt, rs_div(-((c.DTSecs * i) - (c.DTSecs * initlen)), den));
        tair = tair - (tairR - s.tair_end) * e;
        vz = vz - (vzR - s.vz_end) * e;
        rhz = rhz - (rhR - s.rh_end) * e;
        if (rhz > R4(100.)) rhz = R4(100.0);
      }
    }
    cp.sw_cof = q.swcof; cp.lw_cof = q.lwcof; cp.last_tsurf_obs = q.lastobs;
    double sw_in = f.sw, lw_in = f.lw;
    if (sky_on) {
      /* the reference runs this between PrecipitationToStorage and BalanceModelOneStep
       * (Simulation.f90:151-162); the two do not share data, so the order is free */
      if (!sky_view_radiation(ka->f.sun + (int64_t)(i - t0) * RS_SUN_COLS, sinlat, coslat, lonrad, coslon, sinlon, skyv,
                              ka->pp.albedo_surroundings,
                              ka->pp.horizons
                                  ? ka->pp.horizons + (ka->pp.horizon_index ? (int64_t)ka->pp.horizon_index[p] : p) *
                                                          (ka->pp.horizons_by_point ? 360 : 1)
                                  : nullptr,
                              ka->pp.horizons_by_point ? (int64_t)1 : np, sw_in, sw_dir, lw_in, lw_net))
        fail_at(i); /* the reference would `stop` the process here */
    }
    if (ka->wb.sw_dir) { /* in-place input edits of the reference; a replay overwrites them */
      const int64_t woff = (int64_t)(i - t0) * ka->wb.t_stride + p;
      ka->wb.sw_dir[woff] = sw_dir;
      if (sky_on) {
        ka->wb.sw[woff] = sw_in;
        ka->wb.lw[woff] = lw_in;
      }
    }
    if (DIAG) bl_diagnose(&c, mt.expT, mt.logT, mt.K, s.tsurf, tair, vz, rhz, f.hour, i, ka->diag, np, p);
    const Fluxes fx = model_step_fluxes(c, mt, s, tair, vz, rhz, prec_ts, sw_in, lw_in, f.phase,
                                        f.hour, cp);
    if (i + 1 < tend) { /* next index's forcing, half a step ahead of its use */
      nxt = gather_forcing(ka, p, i + 1, t0);
      nxt_i = i + 1;
    }
    if (stale_all) {
      /* observation forcing cannot follow a restore (i >= couplingStartI), so TmpNw(1:2)
       * are the stale values too */
      const GlobalProfile Tstale{st + (int64_t)RS_ST_CPL_STALE_TMP0 * np + p, np};
      model_step_ground(c, s, T, tbot, tair, fx, f.depth, cp, &Tstale);
      stale_all = false;
    } else {
      model_step_ground(c, s, T, tbot, tair, fx, f.depth, cp);
    }
    int64_t orow;
    if (output_row<false>(ka, i, orow)) store_outputs<true>(ka, orow, p, 0u, s, true);
    if (i > written_hi) written_hi = i;
    /* CheckEndCoupling + CouplingOperations2, src/Coupling.f90:98-141 */
    if (i < c.SimLen && q.on && i == q.ce && !q.failed) {
      if (q.iter == 0) q.tend1 = s.tsurf;
      coupling_control(q, s.tsurf);
      q.iter = q.iter + 1;
      if (s.failed) q.again = false; /* failed at this index: the loop exits, no rewind, not listed */
      if (ka->cpl_stop) { /* park: the next round decides between a replay and going on */
        ++i;
        break;
      }
    }
    ++i;
  }
  st[(int64_t)RS_ST_CPL_RESUME * np + p] = (double)i;
}

/* RS_ST_BLSCORE from the loop's counter: bits 0-18 extra passes (saturating), bit 19 cover,
 * bit 20 regime (the counter itself carries the regime flag in bit 30).
 * Sorted (descending: expensive first, rs_cluster.hip) the unstable-regime points come first,
 * by the passes they needed, then the stable-regime points, whose waves never enter the log/sqrt
 * branch (tools/bl_persistence.py). */
__device__ __forceinline__ double bl_score_key(int32_t score, const Scalars &s) {
  const int32_t extra = score & 0x3fffffff;
  const int32_t lo = extra > 0x7ffff ? 0x7ffff : extra;
  /* bit 20: something lies on the road (the storage and melting branches are skipped by a
   * wavefront whose lanes are all bare) */
  const int32_t covered = (s.wat > 0.0 || s.snow > 0.0 || s.ice > 0.0 || s.ice2 > 0.0 || s.dep > 0.0) ? 1 : 0;
  return (double)(lo | (covered << 19) | (((score >> 30) & 1) << 20));
}

/* (waves per SIMD: LEAN four, FULL three - measured, rs_launch_step; the choice was a digit of the variant until
 * round 6) */
template <int NL, bool FULL, bool SCORE = true, bool A32 = false>
__global__ void __launch_bounds__(kBlock, FULL ? 3 : 4) step_kernel_reg(const StepArgs a) {
  __shared__ double math_lds[RS_MATH_LDS_DOUBLES];
  const MathTab mt = fill_math_tables(math_lds);
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= a.npoints) return;
  RegProfile<NL> T;
  Scalars s;
  int32_t score = 0;
#ifdef RS_WAVE_TIMING /* tools/wave_times.py: when each wavefront of a launch starts and ends (100 MHz ticks) */
  const uint64_t wt0 = __builtin_amdgcn_s_memrealtime();
#endif
  load_state<FULL>(a.state, a.np_pad, p, T, s);
  time_loop<FULL, RegProfile<NL>, false, SCORE, false, false, A32>(mt, T, s, score);
  store_state<FULL>(a.state, a.np_pad, p, T, s);
  if (SCORE) a.state[(int64_t)RS_ST_BLSCORE * a.np_pad + p] = bl_score_key(score, s);
#ifdef RS_WAVE_TIMING
  if (!FULL) {
    a.state[(int64_t)RS_ST_VZ_END * a.np_pad + p] = (double)(wt0 & 0xffffffffffffull);
    a.state[(int64_t)RS_ST_RH_END * a.np_pad + p] = (double)(__builtin_amdgcn_s_memrealtime() & 0xffffffffffffull);
  }
#endif
}

/* FULL feature set, NLayers = 15, four waves per SIMD: layers 1..RS_HYBRID_REG in registers, the
 * rest in LDS (HybridProfile). */
#ifndef RS_HYBRID_REG
#define RS_HYBRID_REG 7
#endif
template <bool SCORE, bool A32>
__global__ void __launch_bounds__(kBlock, 4) step_kernel_hybrid(const StepArgs a) {
  __shared__ double math_lds[RS_MATH_LDS_DOUBLES];
  __shared__ double prof_lds[(15 - RS_HYBRID_REG) * kBlock];
  const MathTab mt = fill_math_tables(math_lds);
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= a.npoints) return; /* no barriers below: each lane owns its column */
  HybridProfile<15, RS_HYBRID_REG> T;
  T.col = prof_lds + threadIdx.x;
  Scalars s;
  int32_t score = 0;
  load_state<true>(a.state, a.np_pad, p, T, s);
  time_loop<true, HybridProfile<15, RS_HYBRID_REG>, false, SCORE, false, false, A32>(mt, T, s, score);
  store_state<true>(a.state, a.np_pad, p, T, s);
  if (SCORE) a.state[(int64_t)RS_ST_BLSCORE * a.np_pad + p] = bl_score_key(score, s);
}

/* DIAG (here, step_kernel_sky and step_kernel_coupled): the instance a plan with diagnostics runs (bl_diagnose): a
 * call in the middle of the step costs the kernel around it 11 % even when it is not taken (measured, FULL: 1.70e10
 * -> 1.52e10, tools/experiments/r6_diag_hook_cost.sh), so the regular instances do not contain it */
template <bool FULL, bool DIAG = false>
__global__ void __launch_bounds__(kBlock, 4) step_kernel_lds(const StepArgs a) {
  extern __shared__ double lds[]; /* [NLayers][kBlock] */
  __shared__ double math_lds[RS_MATH_LDS_DOUBLES];
  const MathTab mt = fill_math_tables(math_lds);
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= a.npoints) return; /* no barriers below: each lane owns its column */
  LdsProfile T{lds + threadIdx.x, consts_of(&a).NLayers};
  Scalars s;
  int32_t score = 0;
  load_state<FULL>(a.state, a.np_pad, p, T, s);
  time_loop<FULL, LdsProfile, false, true, false, false, false, DIAG>(mt, T, s, score);
  store_state<FULL>(a.state, a.np_pad, p, T, s);
  a.state[(int64_t)RS_ST_BLSCORE * a.np_pad + p] = bl_score_key(score, s);
}

/* ---- two wavefronts per 64 points: the flavour for small shards --------------------------------
 * A launch of one point per lane needs ~260 000 points to put four wavefronts on every SIMD; the
 * per-GPU shard of BASELINE config 4 (1 M points over 8 GPUs) has 125 000: under two waves per
 * SIMD, each of which must issue its ~2 300 instructions per time step one after the other, with
 * nothing to hide its waits behind.  Here a workgroup is TWO wavefronts that share 64 points:
 *   wave 0, "surface": everything of a time step that forms the serial chain - forcing, checks,
 *           precipitation, boundary layer, radiation, layers 1-2, melting, storages, outputs;
 *   wave 1, "ground":  layers 3..N of the explicit profile update (src/BalanceModel.f90:112-128),
 *           which only read the OLD temperatures of their neighbours.
 * They meet once per time step: the surface wave publishes its new Tmp(2), the ground wave its
 * new Tmp(3) (two LDS words per point, double-buffered, one s_barrier).  The serial chain of a
 * step shrinks from ~1 460 to ~1 020 vector instructions and the shard occupies twice as many
 * wave slots.  Same arithmetic in the same order per point: same bits (layer_step is the one
 * function both flavours call).  LEAN feature set, NLayers = 15, 32-bit window offsets. */
/* what a step needs of its forcing alone (ForcingPrep), [value][lane]: nine values, ten with the FULL feature set */
enum { PR_TAIR = 0, PR_VZ, PR_RHZ, PR_RAIN, PR_SNOW, PR_AVC, PR_EAIR, PR_SW, PR_LW, PR_OBS, PR_LEAN = PR_OBS, PR_FULL = PR_OBS + 1 };
template <int NPREP>
struct DuoMailT {
  double v[2][2][64]; /* [buffer][0: Tmp(2) from the surface wave, 1: Tmp(3) from the ground wave][lane] */
  uint32_t failed[64]; /* sticky, set by the surface wave: the point's loop has exited (the ground wave
                          then leaves Tmp(3..N) alone, as the one-point-per-lane flavours do) */
  /* round 4: what a step needs of its forcing alone (ForcingPrep, rs_physics_body.inc), worked out by
   * the ground wave one index ahead: [buffer = index parity][value][lane].  The traffic friction is the same for
   * every point of an index (one bit per buffer), VK x VZ and the products with the air's volumetric heat capacity
   * are multiplications for the surface wave.  Round 6: the air's density and specific heat travel as their product
   * (all the surface wave ever forms of them; a product is the same bits in either order) and the psychrometric
   * constant is three operations on the air temperature the surface wave has anyway: nine values instead of eleven
   * (+ 1.2 % at 1 M points; 16 192 B per workgroup of the LEAN instances, see RS_DUO_LEAN_WAVES). */
  double prep[2][NPREP][64];
  /* bit 0: CheckValues' verdict on the forcing; bit 1: the index falls in the night of
   * SetDayDependendVariables (src/BalanceModel.f90:354-387) - per LANE: with RsForcing::hour_pstride = 1
   * (runsimulation_batch: every point brings its own calendar) the hour is the point's, not the index's */
  uint32_t prep_bad[2][64];
};

template <bool FULL, bool SKYG = false, class Mail>
__device__ __forceinline__ void duo_put_prep(Mail &mail, int buf, uint32_t lane, const ForcingPrep &q) {
  double (*w)[64] = mail.prep[buf];
  w[PR_TAIR][lane] = q.tair; w[PR_VZ][lane] = q.vz; w[PR_RHZ][lane] = q.rhz; w[PR_RAIN][lane] = q.rain;
  w[PR_SNOW][lane] = q.snow;
  w[PR_AVC][lane] = q.AirHCap * q.AirDens; /* AirVCap, forcing_prep_tail's expression */
  w[PR_EAIR][lane] = q.EAir;
  if (!SKYG) { /* (SKYG: the radiation is the sky wave's to hand over, duo_sky) */
    w[PR_SW][lane] = q.sw;
    w[PR_LW][lane] = q.lw;
  }
  if (FULL) w[PR_OBS][lane] = q.tsurfobs;
  mail.prep_bad[buf][lane] = (q.bad ? 1u : 0u) | (q.night ? 2u : 0u) | (q.bad_rw ? 8u : 0u);
}
template <bool FULL, bool SKYG = false, class C, class Mail>
__device__ __forceinline__ ForcingPrep duo_get_prep(const C &c, const Mail &mail, int buf, uint32_t lane,
                                                    const uint32_t *skyfl = nullptr) {
  const double (*w)[64] = mail.prep[buf];
  ForcingPrep q;
  q.tair = w[PR_TAIR][lane]; q.vz = w[PR_VZ][lane]; q.rhz = w[PR_RHZ][lane]; q.rain = w[PR_RAIN][lane];
  q.snow = w[PR_SNOW][lane];
  const double AirVCap = w[PR_AVC][lane];
  q.AirVCap = AirVCap;
  q.EAir = w[PR_EAIR][lane]; q.sw = w[PR_SW][lane]; q.lw = w[PR_LW][lane];
  if (FULL) q.tsurfobs = w[PR_OBS][lane];
  uint32_t flags = mail.prep_bad[buf][lane];
  if (SKYG) flags |= skyfl[buf * 64 + lane]; /* the sky wave's share of CheckValues (bit 0) and its `stop` (bit 2) */
  { /* both constants first (scalar loads), then the lane picks a VALUE: see fluxes_pre */
    const double fricN = c.TrfFricNgt, fricD = c.TrFfricDay;
    q.night = (flags & 2u) != 0u;
    q.trffric = q.night ? fricN : fricD;
  }
  /* forcing_prep_tail's expressions on forcing_prep_tail's values */
  q.PsychC = R4(0.1) * (R4(0.00063) * (q.tair + R4(273.15)) + R4(0.47496));
  q.den0 = AirVCap * (q.tair + R4(273.15));
  q.vkvz = c.VK_Const * q.vz;
  q.avk = AirVCap * c.VK_Const;
  q.bad = (flags & 1u) != 0u;
  q.stop = SKYG && (flags & 4u) != 0u; /* only an instance with a sky wave raises it */
  q.bad_rw = (flags & 8u) != 0u;       /* (only a replay instance's ground wave raises it) */
  return q;
}

/* LDS writes done, then the workgroup barrier.  Not __syncthreads(): that also waits for the global
 * memory counter, i.e. for the six output stores of the step and the prefetched forcing loads. */
__device__ __forceinline__ void duo_meet() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

/* SKY (with FULL): sky view and local horizons (src/ModRadiation.f90, examples/example1/src/Simulation.f90:
 * 151-162) as the lock-step loop has them - the surface wave reads SW_dir and LW_net of the index itself (two
 * coalesced loads), runs CheckValues' sky-view tests and ModRadiationBySurroundings on the short- and long-wave
 * radiation the ground wave handed over, and writes the reference's in-place edits back where asked to. */
/* SKYG: a third wavefront does the sky view (duo_sky, the raw-series flavour): this wave gets the radiation as
 * the sky view leaves it and the sky wave's flags beside the ground wave's. */
/* OUTIDX: the launch may ask for its output rows in POINT order (StepArgs::out_index, the plan's order row:
 * the decimated rows of rs_driver_run go straight to their point's column, no copy kernel behind the launch). */
/* CPL: coupling in lock step, as time_loop<CPL> has it (src/Coupling.f90; everything but the replays): the state
 * saved at the window start (this wave its scalars and layers 1-2, the ground wave layers 3..N), the coupling
 * phase, Coupling_control at the window end - a point that has to replay PARKS -, the decaying corrections
 * behind the window.  A lane steps index i only if it is the index it is due for (RS_ST_CPL_RESUME); parked or
 * ahead, it is frozen for that index in both wavefronts (DuoMail::failed is rewritten before every meeting). */
/* REPLAY (with CPL): one replay ROUND of the coupling windows in lock step, as time_loop<REPLAY> has it, for the
 * points of the compacted list that asked for another replay (StepArgs::cpl_list): the launch covers [first
 * window start, last window end + 1]; a lane waits for its window start, rewinds there - CheckValues of the
 * index behind its window end, as the reference's loop does before CouplingOperations1 takes it back
 * (examples/example1/src/Simulation.f90:62-71: the forcing's verdict comes from the ground wave, ForcingPrep::
 * bad_rw), then uploadDataForCoupling (src/Coupling.f90:213-255) and the radiation coefficient (:61-78) -, runs
 * its window in the coupling phase and lets Coupling_control decide again at the end.  The list breaks the tie
 * between lane and point: `lane` below is the POINT (row0 = 0), `ml` the lane of the mailbox. */
template <int NL, bool SCORE, bool FULL = false, bool SKY = false, bool SKYG = false, bool OUTIDX = false,
          bool CPL = false, bool REPLAY = false>
__device__ __forceinline__ void duo_surface(const MathTab &mt, DuoMailT<FULL ? PR_FULL : PR_LEAN> &mail, const StepArgs &a,
                                            const uint32_t *skyfl = nullptr) {
  static_assert(FULL || !SKY, "sky view belongs to the FULL feature set");
  static_assert(!(SKY && SKYG), "the sky view is this wave's or the ground wave's");
  static_assert(FULL || !CPL, "coupling belongs to the FULL feature set");
  static_assert(CPL || !REPLAY, "replays belong to coupling");
  KernArgs ka = kernargs();
  const uint32_t ml = threadIdx.x & 63u;
  /* the workgroup's slots: 64 from 64 * blockIdx.x, or what the wave table says (rs_cluster_wave_table);
   * REPLAY: entries 64 * blockIdx.x ... of the list */
  const int64_t listed = REPLAY ? (int64_t)blockIdx.x * 64 + ml : 0;
  const int64_t row0 = REPLAY ? 0 : a.wave_start ? (int64_t)a.wave_start[blockIdx.x] : (int64_t)blockIdx.x * 64;
  const uint32_t lane = REPLAY ? (listed < a.cpl_nlist ? (uint32_t)a.cpl_list[listed] : 0u) : ml;
  const int64_t p = row0 + lane;
  const bool live = REPLAY ? listed < a.cpl_nlist
                           : a.wave_start ? (int32_t)lane < a.wave_cnt[blockIdx.x] : p < a.npoints; /* a dead lane still walks to every barrier */
  RegProfile<2> T;
  Scalars s;
  {
    RegProfile<NL> all;
    if (live) {
      load_state<false>(a.state, a.np_pad, p, all, s);
    } else {
      for (int j = 1; j <= NL; ++j) all.set(j, 0.0);
      s = Scalars();
      s.failed = true;
    }
    T.set(1, all.get(1));
    T.set(2, all.get(2));
  }
  double skyv = R4(1.0), sinlat = 0, coslat = 0, lonrad = 0, coslon = 1.0, sinlon = 0;
  bool sky_on = false;
  uint32_t hcol = 0;
  if (SKY && live) { /* as time_loop<SKY> */
    skyv = ka->pp.sky_view[p];
    sky_on = (skyv < R4(1.0) && skyv > R4(-0.01));
    if (sky_on) {
      sinlat = ka->pp.sin_lat[p];
      coslat = ka->pp.cos_lat[p];
      lonrad = ka->pp.lon_rad[p];
      coslon = ::cos(lonrad);
      sinlon = ::sin(lonrad);
    }
    hcol = ka->pp.horizon_index ? (uint32_t)ka->pp.horizon_index[p] : (uint32_t)p;
  }
  const int32_t nsteps = ka->nsteps, t0 = ka->t0;
  /* coupling, lock-step part (time_loop<CPL>): what a lane needs every step in registers, the rest of
   * CouplingVariables in the state block, touched at the two events only */
  bool cpl_on = false, parked = false;
  int32_t cpl_cs = -99, cpl_ce = -99, next_i = t0;
  double cpl_lastobs = 0.0, cpl_swcorr = 0.0, cpl_lwcorr = 0.0;
  if (CPL && live) {
    const double *st = a.state;
    const int64_t np = a.np_pad;
    const int32_t cidx = ka->pp.coupling_index ? ka->pp.coupling_index[p] : 0;
    /* setInputParam + initCouplingTimes, src/InputOutput.f90:30-36, src/Coupling.f90:486-534 */
    cpl_on = consts_of(ka).use_coupling && ka->pp.coupling_index && !(ka->pp.coupling_tsurf[p] < -100 || cidx < 1);
    if (cpl_on) {
      cpl_ce = cidx;
      cpl_cs = ((double)cidx <= consts_of(ka).cplLenR) ? 1 : cidx - consts_of(ka).cplLenI;
    }
    cpl_lastobs = st[(int64_t)RS_ST_CPL_LASTOBS * np + p];
    cpl_swcorr = st[(int64_t)RS_ST_CPL_SWCORR * np + p];
    cpl_lwcorr = st[(int64_t)RS_ST_CPL_LWCORR * np + p];
    parked = (((int32_t)st[(int64_t)RS_ST_CPL_FLAGS * np + p]) & 1) != 0;
    next_i = (int32_t)st[(int64_t)RS_ST_CPL_RESUME * np + p];
    if (REPLAY) { /* a listed point: parked behind its window, wanting it again */
      parked = false;
      next_i = cpl_cs;
    }
  }
  /* REPLAY: SWRadCof / LWRadCof of the window being replayed */
  double r_swcof = R4(1.0), r_lwcof = R4(1.0);
  mail.v[0][0][ml] = T.get(2);
  /* arrives failed (or a dead lane): frozen from the first index; CPL: or not due for it */
  mail.failed[ml] = (s.failed || (CPL && (parked || t0 != next_i))) ? 1u : 0u;
  duo_meet();
  auto blank_rows = [&](int32_t i_from) { /* as in time_loop */
    for (int32_t ii = i_from; ii < t0 + nsteps; ++ii) {
      int64_t r;
      if (output_row<false>(ka, ii, r)) {
        if (OUTIDX && ka->out_index) store_outputs<true, false>(ka, r, row0, lane, s, false);
        else store_outputs<false, false>(ka, r, row0, lane, s, false);
      }
    }
  };
  /* (CPL: from the index the point is due for - rows before it were written when it was stepped there, by a
   * replay launch that ran ahead of this chunk) */
  if (live && s.failed && !REPLAY) blank_rows(CPL && next_i > t0 ? next_i : t0);
  int32_t score = 0; /* scheduling hint of rs_hip_recluster, as in time_loop */
  /* (the forcing windows are the ground wave's to read: this wave gets what a step needs of them
   * through the mailbox, worked out one index ahead - ForcingPrep) */
  for (int32_t kv = 0; kv < nsteps; ++kv) {
    asm volatile("" : "+s"(ka));
    const ConstsAS &c = consts_of(ka);
    const int32_t k = __builtin_amdgcn_readfirstlane(kv);
    const int32_t i = t0 + k;
    int64_t orow = 0;
    const bool owrite = output_row<true>(ka, i, orow);
    double t3 = mail.v[k & 1][1][ml]; /* Tmp(3) as the last step left it */
    bool go = true; /* CPL: the lane is due for this index (a failed lane keeps counting: its rows are blanked) */
    if (CPL) {
      go = !(parked || i != next_i);
      if (go) next_i = i + 1;
    }
    if (go && !s.failed) {
      const ForcingPrep q = duo_get_prep<FULL, SKYG>(c, mail, k & 1, ml, skyfl);
      if (i < c.SimLen) { /* CheckValues: the forcing's verdict | the surface temperature's */
        /* REPLAY, at the rewind: the index CheckValues has just seen is the one behind the window end - with
         * the surface temperature of the end of the window */
        const bool rewind = REPLAY && i == cpl_cs;
        if ((rewind ? q.bad_rw : q.bad) | check_values_tsurf(c, s.tsurf)) {
          s.failed = true;
          ka->state[(int64_t)RS_ST_FAILED * ka->np_pad + row0 + lane] = (double)(rewind ? cpl_ce + 1 : i);
          mail.failed[ml] = 1u; /* this index still steps (the ground wave is in it already); the next does not */
        }
      }
      if (SKYG && q.stop) { /* as duo_surface<SKY> flags it: at any index */
        s.failed = true;
        ka->state[(int64_t)RS_ST_FAILED * ka->np_pad + row0 + lane] = (double)i;
        mail.failed[ml] = 1u;
      }
      s.tnw1 = T.get(1);
      s.tnw2 = T.get(2);
      CouplingInputs cp;
      if (CPL && cpl_on) { /* CouplingOperations1, src/Coupling.f90:10-96, first pass of the window: time_loop<CPL> */
        if (i < c.SimLen) {
          cp.in_phase = (i >= cpl_cs && i <= cpl_ce);
          if (REPLAY && i == cpl_cs) {
            /* inCouplingPhase is set from the loop index BEFORE a rewind takes it back (:22-27 against :61-66):
             * the step at the window start of a replay runs outside the coupling phase.  uploadDataForCoupling
             * :213-255: SrfIcemms, Q2Melt, T4Melt and TmpNw are NOT restored - s.tnw1/2 above hold the
             * end-of-window values CalcHCapHCond sees in this step (the ground wave does the same for 3..N) */
            cp.in_phase = false;
            double *st = ka->state;
            const int64_t np = ka->np_pad, pp_ = row0 + lane;
            s.tsurf = st[(int64_t)RS_ST_CPL_SAVE_TSURF * np + pp_];
            s.wat = st[(int64_t)RS_ST_CPL_SAVE_WAT * np + pp_];
            s.ice2 = st[(int64_t)RS_ST_CPL_SAVE_ICE2 * np + pp_];
            s.dep = st[(int64_t)RS_ST_CPL_SAVE_DEP * np + pp_];
            s.snow = st[(int64_t)RS_ST_CPL_SAVE_SNOW * np + pp_];
            s.albedo = st[(int64_t)RS_ST_CPL_SAVE_ALBEDO * np + pp_];
            const int32_t fl = (int32_t)st[(int64_t)RS_ST_CPL_FLAGS * np + pp_];
            s.verycold = (fl & 4) != 0;
            st[(int64_t)RS_ST_CPL_FLAGS * np + pp_] = (double)(fl & ~1); /* start_coupling_again = .false. */
            T.set(1, st[(int64_t)(RS_ST_CPL_SAVE_TMP0 + 0) * np + pp_]);
            T.set(2, st[(int64_t)(RS_ST_CPL_SAVE_TMP0 + 1) * np + pp_]);
            t3 = st[(int64_t)(RS_ST_CPL_SAVE_TMP0 + 2) * np + pp_]; /* layer 2's lower boundary: the restored Tmp(3) */
            /* short-wave scaling by day, long-wave by night (:68-76; no sky view in this instance) */
            const double radcoeff = st[(int64_t)RS_ST_CPL_RADCOEFF * np + pp_];
            if (q.sw > q.lw) {
              r_swcof = radcoeff;
              r_lwcof = R4(1.0);
            } else {
              r_swcof = R4(1.0);
              r_lwcof = radcoeff;
            }
            st[(int64_t)RS_ST_CPL_SWCOF * np + pp_] = r_swcof; /* Coupling_control reads them at the end */
            st[(int64_t)RS_ST_CPL_LWCOF * np + pp_] = r_lwcof;
          }
          if (!REPLAY && i == cpl_cs) { /* saveDataForCoupling :172-210 - this wave's share (the ground wave saves layers 3..N) */
            double *st = ka->state;
            const int64_t np = ka->np_pad, pp_ = row0 + lane;
            st[(int64_t)RS_ST_CPL_SAVE_TSURF * np + pp_] = s.tsurf;
            st[(int64_t)RS_ST_CPL_SAVE_WAT * np + pp_] = s.wat;
            st[(int64_t)RS_ST_CPL_SAVE_ICE2 * np + pp_] = s.ice2;
            st[(int64_t)RS_ST_CPL_SAVE_DEP * np + pp_] = s.dep;
            st[(int64_t)RS_ST_CPL_SAVE_SNOW * np + pp_] = s.snow;
            st[(int64_t)RS_ST_CPL_SAVE_ALBEDO * np + pp_] = s.albedo;
            const int32_t fl = ((int32_t)st[(int64_t)RS_ST_CPL_FLAGS * np + pp_]) & (3 | RS_CPL_MSG_SMALL | RS_CPL_MSG_BIG);
            st[(int64_t)RS_ST_CPL_FLAGS * np + pp_] = (double)(fl | (s.verycold ? 4 : 0));
            st[(int64_t)(RS_ST_CPL_SAVE_TMP0 + 0) * np + pp_] = T.get(1);
            st[(int64_t)(RS_ST_CPL_SAVE_TMP0 + 1) * np + pp_] = T.get(2);
          }
          if (i > cpl_ce) {
            const double e = cpl_decay(c, mt, i, cpl_ce);
            cp.sw_cof = R4(1.0) + cpl_swcorr * e;
            cp.lw_cof = R4(1.0) + cpl_lwcorr * e;
          }
          if (REPLAY && i >= cpl_cs && i <= cpl_ce) {
            cp.sw_cof = r_swcof;
            cp.lw_cof = r_lwcof;
          }
          if (cp.in_phase) { /* snowIceCheck :259-289 */
            if (cpl_lastobs > c.TLimMeltSnow && s.snow > R4(0.00)) { s.wat = s.wat + s.snow; s.snow = R4(0.00); }
            if (cpl_lastobs > c.TLimMeltIce && s.ice > R4(0.00)) { s.wat = s.wat + s.ice; s.ice = R4(0.00); }
            if (cpl_lastobs > c.TLimMeltIce && s.ice2 > R4(0.00)) s.ice2 = R4(0.00);
            if (cpl_lastobs > c.TLimMeltDep && s.dep > R4(0.00)) { s.wat = s.wat + s.dep; s.dep = R4(0.00); }
          }
        } else { /* lastValues: inCouplingPhase and the coefficients keep the values of index SimLen - 1 */
          const int32_t j = c.SimLen - 1;
          cp.in_phase = (j >= cpl_cs && j <= cpl_ce);
          if (j > cpl_ce) {
            const double e = cpl_decay(c, mt, j, cpl_ce);
            cp.sw_cof = R4(1.0) + cpl_swcorr * e;
            cp.lw_cof = R4(1.0) + cpl_lwcorr * e;
          } else if (j >= cpl_cs) {
            cp.sw_cof = ka->state[(int64_t)RS_ST_CPL_SWCOF * ka->np_pad + row0 + lane];
            cp.lw_cof = ka->state[(int64_t)RS_ST_CPL_LWCOF * ka->np_pad + row0 + lane];
          }
        }
      }
      if (CPL) cp.last_tsurf_obs = cpl_lastobs;
      if (FULL) {
        /* SetCurrentValues' observation forcing (src/InputOutput.f90:116-148; the ground wave decided
         * whether it applies at this index and uses the same value for Tmp(2)), and lastValues' surface
         * temperature (:169-198) - no depth stream and no tsurfOutputDepth in this flavour: the mean of
         * the two top layers (surface_temperature) */
        if (i < c.SimLen) {
          if (q.tsurfobs > R4(-100.0)) {
            T.set(1, q.tsurfobs);
            T.set(2, q.tsurfobs);
            s.tsurf = (T.get(1) + T.get(2)) / R4(2.0);
          }
        } else {
          s.tsurf = (T.get(1) + T.get(2)) / R4(2.0);
        }
      }
      const double tair = q.tair;
      ForcingPrep qs = q;
      if (SKY) {
        const int64_t row = (int64_t)k * ka->f.t_stride + row0;
        const LaneOff L(lane);
        double sw_dir = L.ld(ka->f.sw_dir + row), lw_net = L.ld(ka->f.lw_net + row);
        if (i < c.SimLen) {
          if (sky_on && (sw_dir < R4(-0.1) || sw_dir > R4(4000.0) || lw_net < R4(-1000.0) || lw_net > R4(1000.0))) {
            s.failed = true; /* src/InputOutput.f90:68-74 */
            ka->state[(int64_t)RS_ST_FAILED * ka->np_pad + row0 + lane] = (double)i;
            mail.failed[ml] = 1u;
          }
          if (sw_dir > q.sw) sw_dir = q.sw; /* :75-77 */
        }
        if (sky_on) {
          if (!sky_view_radiation(ka->f.sun + (int64_t)k * RS_SUN_COLS, sinlat, coslat, lonrad, coslon, sinlon, skyv,
                                  ka->pp.albedo_surroundings,
                                  ka->pp.horizons ? ka->pp.horizons + (ka->pp.horizons_by_point ? (int64_t)hcol * 360 : (int64_t)hcol) : nullptr,
                                  ka->pp.horizons_by_point ? (int64_t)1 : ka->np_pad, qs.sw, sw_dir, qs.lw, lw_net)) {
            s.failed = true; /* the reference would `stop` the process here */
            ka->state[(int64_t)RS_ST_FAILED * ka->np_pad + row0 + lane] = (double)i;
            mail.failed[ml] = 1u;
          }
        }
        if (ka->wb.sw_dir) { /* the caller's arrays as the reference leaves them (time_loop<SKY>) */
          const int64_t wrow = (int64_t)k * ka->wb.t_stride + row0;
          L.st(ka->wb.sw_dir + wrow, sw_dir);
          if (sky_on) {
            L.st(ka->wb.sw + wrow, qs.sw);
            L.st(ka->wb.lw + wrow, qs.lw);
          }
        }
      }
      const Fluxes fx = model_step_fluxes_prepped<SCORE ? 1 : 2>(c, mt, s, qs, cp);
      if (SCORE) {
        score += (fx.trips & 63) - 5;
        if ((fx.trips & 64) && k >= nsteps - RS_REGIME_WINDOW) score |= 1 << 30;
      } else if (k == nsteps - 1) {
        score = fx.trips; /* the launch's last index: what forecast_key_kernel takes as one more preview */
      }
      /* layers 1-2 with Tmp(3) where a two-layer column has its lower boundary */
      model_step_ground<RegProfile<2>, RegProfile<2>, false>(c, s, T, t3, tair, fx, R4(-9999.9), cp);
      if (owrite) {
        if (OUTIDX && ka->out_index) store_outputs<true, false>(ka, orow, row0, lane, s, true);
        else store_outputs<false, true>(ka, orow, row0, lane, s, true);
      }
      /* (REPLAY: a point that fails in the middle of a replay keeps, up to its window end, what the earlier
       * passes saved there: src/InputOutput.f90:151-165 only ever overwrites) */
      if (s.failed) blank_rows(REPLAY && cpl_ce + 1 > i + 1 ? cpl_ce + 1 : i + 1);
      if (CPL && cpl_on && i < c.SimLen && i == cpl_ce) {
        /* CheckEndCoupling + CouplingOperations2, src/Coupling.f90:98-141: as time_loop<CPL> */
        double *st = ka->state;
        const int64_t np = ka->np_pad, pp_ = row0 + lane;
        Coupling cq;
        load_coupling(st, np, pp_, cq);
        cq.cs = cpl_cs; cq.ce = cpl_ce; cq.on = true;
        if (!cq.failed) {
          if (cq.iter == 0) cq.tend1 = s.tsurf;
          coupling_control(cq, s.tsurf);
          cq.iter = cq.iter + 1;
          if (s.failed) cq.again = false; /* its loop exits before any rewind: not parked, not listed */
          store_coupling(st, np, pp_, cq);
          cpl_swcorr = cq.swcorr;
          cpl_lwcorr = cq.lwcorr;
          cpl_lastobs = cq.lastobs;
          parked = cq.again;
        }
      }
    }
    /* CPL: frozen at the NEXT index? (failed, parked, or not due for it) */
    if (CPL) mail.failed[ml] = (s.failed || parked || (i + 1) != next_i) ? 1u : 0u;
    mail.v[(k & 1) ^ 1][0][ml] = T.get(2);
    duo_meet();
  }
  if (CPL && live) a.state[(int64_t)RS_ST_CPL_RESUME * a.np_pad + p] = (double)next_i;
  if (live) {
    double *st = a.state;
    const int64_t np = a.np_pad;
    st[(int64_t)(RS_ST_TMP0 + 0) * np + p] = T.get(1);
    st[(int64_t)(RS_ST_TMP0 + 1) * np + p] = T.get(2);
    st[(int64_t)RS_ST_TNW1 * np + p] = s.tnw1;
    st[(int64_t)RS_ST_TNW2 * np + p] = s.tnw2;
    st[(int64_t)RS_ST_TSURF * np + p] = s.tsurf;
    st[(int64_t)RS_ST_WAT * np + p] = s.wat;
    st[(int64_t)RS_ST_SNOW * np + p] = s.snow;
    st[(int64_t)RS_ST_ICE * np + p] = s.ice;
    st[(int64_t)RS_ST_ICE2 * np + p] = s.ice2;
    st[(int64_t)RS_ST_DEP * np + p] = s.dep;
    st[(int64_t)RS_ST_Q2MELT * np + p] = s.q2melt;
    st[(int64_t)RS_ST_T4MELT * np + p] = s.t4melt;
    st[(int64_t)RS_ST_ALBEDO * np + p] = s.albedo;
    st[(int64_t)RS_ST_VERYCOLD * np + p] = s.verycold ? 1.0 : 0.0;
    st[(int64_t)RS_ST_BLSCORE * np + p] = SCORE ? bl_score_key(score, s) : (double)score;
  }
}

/* The forcing of (1-based) index i from the hourly knots: expand_kernel's arithmetic, value for value -
 * v0 + (secs * (k1 - k0)) / span with the difference taken once per knot interval and the division by
 * the uniform span as rs_div_u; PrecPhase from the later knot between knots; the hour of rs_sy_hour.
 * (The LEAN feature set reads neither Tdew nor, after the initialization, TsurfObs.)  Knot columns are
 * per point: gathered through the plan's order row, once per interval. */
template <bool FULL>
struct KnotLerp {
  double v0[FULL ? 7 : 6], dv[FULL ? 7 : 6]; /* tair, vz, rhz, prec, sw, lw [, tdew] */
  int32_t ph0, ph1;
};
template <bool FULL>
__device__ __forceinline__ Forcing knot_forcing(KernArgs ka, int64_t col, bool live, KnotLerp<FULL> &K,
                                                int32_t &kcur, int32_t i) {
  const int32_t spk = ka->spk;
  const int32_t t = i - 1;
  const int32_t k = __builtin_amdgcn_readfirstlane(t / spk);
  const int32_t r = t - k * spk;
  if (k != kcur) { /* uniform: a new knot interval */
    kcur = k;
    const int fld[7] = {0, 2, 3, 4, 5, 6, 1};
    const int64_t np = ka->np_pad;
    const double *ka_ = ka->knots + ((int64_t)(k - ka->knot_k0) * RS_KNOT_FIELDS) * np + col;
    const bool has_b = (k + 1 - ka->knot_k0) < ka->knot_n;
    const double *kb_ = ka_ + (int64_t)RS_KNOT_FIELDS * np;
#pragma unroll
    for (int q = 0; q < (FULL ? 7 : 6); ++q) {
      K.v0[q] = live ? ka_[(int64_t)fld[q] * np] : 0.0;
      const double v1 = (live && has_b) ? kb_[(int64_t)fld[q] * np] : K.v0[q];
      K.dv[q] = v1 - K.v0[q];
    }
    K.ph0 = live ? (int32_t)ka_[8 * np] : 0;
    K.ph1 = (live && has_b) ? (int32_t)kb_[8 * np] : K.ph0;
  }
  const double secs = (double)r, span = (double)spk;
  Forcing f;
  f.tair = K.v0[0] + rs_div_u(secs * K.dv[0], span, ka->r_spk);
  f.vz = K.v0[1] + rs_div_u(secs * K.dv[1], span, ka->r_spk);
  f.rhz = K.v0[2] + rs_div_u(secs * K.dv[2], span, ka->r_spk);
  f.prec = K.v0[3] + rs_div_u(secs * K.dv[3], span, ka->r_spk);
  f.sw = K.v0[4] + rs_div_u(secs * K.dv[4], span, ka->r_spk);
  f.lw = K.v0[5] + rs_div_u(secs * K.dv[5], span, ka->r_spk);
  f.phase = (r == 0) ? K.ph0 : K.ph1;
  f.hour = rs_sy_hour(i, spk, ka->start_hour);
  f.tdew = 0.0;
  f.tsurfobs = R4(-9999.9);
  if (FULL) { /* expand_kernel<TDEW, OBS>: the dew point like the others, the observation at index 1 only */
    f.tdew = K.v0[6] + rs_div_u(secs * K.dv[6], span, ka->r_spk);
    if (t == 0 && live) f.tsurfobs = (ka->knots + ((int64_t)(k - ka->knot_k0) * RS_KNOT_FIELDS + 7) * ka->np_pad)[col];
  }
  f.depth = R4(-9999.9);
  return f;
}

/* ---- the forcing from the RAW series (rs_driver_run's blocks, rs_step_raw) ---------------------------
 * What the driver path's expansion kernel wrote into forcing windows, the ground wave now forms in
 * registers, one index ahead of the surface wave: per variable JsonSource::interpolate's value of every
 * source (examples/example1/src/JsonSource.cpp:86-170), GetWeather's own test of it (:337-356) and the
 * overlay in source order (DataHandler.cpp:75-84), value for value as rs_raw.hpp raw_source_value has them.
 * The sources share their time axes, so which raw interval an index falls into and whether it copies or
 * interpolates there is the same for every point: a SEGMENT (RawSeg) is a run of indices over which every
 * source keeps its (kind, rawPos).  At the first index of a segment the wavefront RESOLVES each variable:
 * every lane loads its two raw ends of every source that has the variable (its column of the series, through
 * the plan's order row), applies the supply test (a copy: raw > threshold; an interpolation: both ends >
 * threshold - the value then lies between them, RawPlanStep::rden) and keeps the ends of the LAST source that
 * supplies.  Where all 64 lanes agree on that source - the rule, not the exception: whether a source has a
 * variable at a time is a property of the data set - the variable costs five instructions per index,
 * v0 + RN(num (v1 - v0) / den) with the source's uniform (num, den, 1/den) on the scalar unit; a copy, or a
 * variable nobody supplies, is v0 + 0.  Anything else - lanes that disagree, a non-finite end, a difference
 * so large or so small that the division by reciprocal is not the IEEE one, a raw -0.0 to be copied, a plan
 * entry with rden = 0 - is evaluated index by index with raw_source_value itself (RAW_SLOW). */
/* Who evaluates which variables: the ground wave alone {tair, vz, rhz, prec, tdew, tsurfobs, sw, lw}
 * (FSET_ALL); with per-point sky view the ground wave the first six of them (FSET_GROUND) and the sky wave
 * {sw, lw, sw_dir, lw_net} (FSET_SKY, duo_sky). */
enum { FSET_ALL = 0, FSET_GROUND = 1, FSET_SKY = 2 };
__device__ __forceinline__ constexpr int raw_nfields(int fset) { return fset == FSET_ALL ? 8 : fset == FSET_GROUND ? 6 : 4; }
__device__ __forceinline__ constexpr int raw_field_of(int fset, int q) {
  return fset == FSET_SKY ? (q == 0 ? RAW_SW : q == 1 ? RAW_LW : q == 2 ? RAW_SWDIR : RAW_LWNET)
                          : (q == 0 ? RAW_TAIR : q == 1 ? RAW_VZ : q == 2 ? RAW_RHZ : q == 3 ? RAW_PREC
                             : q == 4 ? RAW_TDEW : q == 5 ? RAW_OBS : q == 6 ? RAW_SW : RAW_LW);
}
constexpr int kRawObs = 5; /* tsurfobs in FSET_ALL / FSET_GROUND */
template <int NF>
struct RawLerp {
  double v0[NF], dv[NF];
  /* uniform: bit 10 s + q: variable q is interpolated from source s in this segment (s = 4: copied, or
   * missing: v0 + 0); bit 50 + q: index by index (RAW_SLOW) */
  uint64_t mode;
  /* uniform: >= 0: nothing is evaluated index by index in this segment and at most ONE source interpolates
   * (that one; 4: none - every variable is copied or missing): the whole set is one straight line of
   * v0 + RN(num dv / den) with that source's scalars - for a copied or missing variable dv = 0 and the term is
   * +0, as in the general form.  The rule outside the hours in which observations overlay the forecast. */
  int32_t single;
  int32_t seg, seg_end; /* current segment and its end (0-based, exclusive) */
  /* MIXED variables (uniform mask): the lanes disagree on the supplying source - stations without a sensor, gaps
   * in an observation series - but every lane's ends allow the short form: each interpolating source's line is
   * evaluated with that source's scalars and the lane keeps its own (`wsel`, per lane: three bits per variable,
   * the source or 4).  `imask`: the sources that interpolate in this segment. */
  uint32_t mixed, imask;
  uint32_t wsel;
};

/* one variable of one point at 0-based index i, the long way: every source, every test */
__device__ __forceinline__ double raw_slow_value(KernArgs ka, int fld, int64_t col, int32_t i) {
  const double thr = raw_threshold(fld);
  double v = raw_miss();
  const int nsrc = ka->raw.nsrc;
  const int64_t np = ka->raw.np_pad;
  for (int s = 0; s < nsrc; ++s) {
    const double *x = ka->raw.src[s].fld[fld];
    if (!x) continue;
    const RawPlanStep st = raw_plan_at(ka->raw.src[s].plan, i);
    if (st.kind == RAW_NONE) continue;
    const double a = x[(int64_t)st.rp * np + col], b = x[(int64_t)(st.rp + 1) * np + col];
    double vs;
    if (raw_source_value(st, a, b, thr, vs)) v = vs;
  }
  return v;
}

template <int FSET>
__device__ __forceinline__ void raw_resolve(KernArgs ka, int64_t col, bool live, RawLerp<raw_nfields(FSET)> &R) {
  constexpr int NF = raw_nfields(FSET);
  const RawSeg __attribute__((address_space(4))) *sg =
      (const RawSeg __attribute__((address_space(4))) *)ka->raw.segs + R.seg;
  R.seg_end = sg->i1;
  const int32_t i0 = sg->i0;
  const int nsrc = ka->raw.nsrc;
  const int64_t np = ka->raw.np_pad;
  uint64_t mode = 0;
  uint32_t mixed = 0u, imask = 0u, wsel = 0u;
  for (int s = 0; s < nsrc; ++s)
    if (sg->kind[s] == RAW_INTERP) imask |= 1u << s;
#pragma unroll
  for (int q = 0; q < NF; ++q) {
    const int fld = raw_field_of(FSET, q);
    const double thr = raw_threshold(fld);
    int32_t wl = 4;  /* the lane's winner: a source that interpolates, or 4 (copy / nobody) */
    bool okl = true; /* ... and its ends allow the short form */
    double a_w = raw_miss(), d_w = 0.0;
    for (int s = 0; s < nsrc; ++s) {
      const double *x = ka->raw.src[s].fld[fld];
      const int32_t kind = sg->kind[s], rp = sg->rp[s];
      if (!x || kind == RAW_NONE) continue; /* uniform */
      const double *xa = x + (int64_t)rp * np + col;
      const double a = *xa; /* (a dead lane reads column 0) */
      if (kind == RAW_COPY) {
        if (a > thr) {
          wl = 4;
          a_w = a;
          d_w = 0.0;
          okl = !rs_is_neg_zero(a); /* v0 + 0 would lose the sign */
        }
      } else {
        const double b = xa[np];
        if (a > thr && b > thr) {
          const double d = b - a;
          const double den = raw_plan_at(ka->raw.src[s].plan, i0).den;
          const double ad = __builtin_fabs(d);
          wl = s;
          a_w = a;
          d_w = d;
          /* |num d| stays inside the range in which the division by reciprocal is the IEEE one for every
           * 1 <= num < den of the segment (raw_source_value's own test, once per segment) */
          okl = (__builtin_fabs(a) < __builtin_inf()) && (__builtin_fabs(b) < __builtin_inf()) &&
                (den * ad < 1e290) && (d == 0.0 || ad >= 1e-290);
        }
      }
    }
    const int32_t w0 = __builtin_amdgcn_readfirstlane(wl); /* lane 0 is never a dead lane */
    const bool all_ok = __builtin_amdgcn_ballot_w64(live && !okl) == 0ull;
    const bool same = __builtin_amdgcn_ballot_w64(live && wl != w0) == 0ull;
    if (all_ok && same) mode |= 1ull << (10 * w0 + q);
    else if (all_ok) mixed |= 1u << q;
    else mode |= 1ull << (50 + q);
    wsel |= (uint32_t)wl << (3 * q);
    R.v0[q] = a_w;
    R.dv[q] = d_w;
  }
  R.mode = mode;
  R.mixed = mixed;
  R.imask = imask;
  R.wsel = wsel;
  int32_t single = (((mode >> 50) & 1023ull) || mixed) ? -1 : 4;
  for (int s = 0; s < RS_MAX_SOURCES; ++s)
    if ((mode >> (10 * s)) & 1023ull) single = (single == 4) ? s : -1;
  R.single = single;
}

/* The merged raw values of the set's variables at (1-based) index i, in val[].  `need_obs` (FSET_ALL /
 * FSET_GROUND): the step can use the road-temperature observation (initialization phase or force_tsurf),
 * else val[kRawObs] is left unset. */
template <int FSET>
__device__ __forceinline__ void raw_values(KernArgs ka, int64_t col, bool live, RawLerp<raw_nfields(FSET)> &R,
                                           int32_t i, bool need_obs, double (&val)[raw_nfields(FSET)]) {
  constexpr int NF = raw_nfields(FSET);
  const int32_t t = i - 1;
  if (t >= R.seg_end) { /* uniform: a new segment (a segment is never empty) */
    R.seg += 1;
    raw_resolve<FSET>(ka, col, live, R);
  }
  if (R.single >= 0) { /* uniform */
    double num = 0.0, den = 1.0, rden = 1.0;
    if (R.single < RS_MAX_SOURCES) {
      const RawPlanStep st = raw_plan_at(ka->raw.src[R.single].plan, t);
      num = st.num;
      den = st.den;
      rden = st.rden;
    }
    if (rden != 0.0) {
#pragma unroll
      for (int q = 0; q < NF; ++q) val[q] = R.v0[q] + raw_quot(num * R.dv[q], den, rden);
      return;
    }
  }
#pragma unroll
  for (int q = 0; q < NF; ++q) asm volatile("" : "=v"(val[q])); /* each is written by exactly one block below */
  const uint32_t keep = (FSET != FSET_SKY && !need_obs) ? ~(1u << kRawObs) : ~0u;
  uint32_t slow = (uint32_t)(R.mode >> 50) & 1023u & keep;
#pragma nounroll
  for (int s = 0; s < RS_MAX_SOURCES; ++s) {
    const uint32_t m = (uint32_t)(R.mode >> (10 * s)) & 1023u & keep;
    if (m == 0u) continue;
    const RawPlanStep st = raw_plan_at(ka->raw.src[s].plan, t);
    if (st.rden == 0.0) { /* this entry promises nothing (rs_raw.hpp): the long way for this index */
      slow |= m;
      continue;
    }
#pragma unroll
    for (int q = 0; q < NF; ++q)
      if (m & (1u << q)) val[q] = R.v0[q] + raw_quot(st.num * R.dv[q], st.den, st.rden);
  }
  {
    const uint32_t m = (uint32_t)(R.mode >> 40) & 1023u & keep;
#pragma unroll
    for (int q = 0; q < NF; ++q)
      if (m & (1u << q)) val[q] = R.v0[q] + 0.0; /* a copy, or missing: what the short form gives for dv = 0 */
  }
  uint32_t mix = R.mixed & keep;
  if (mix) { /* uniform */
    for (int s = 0; s < RS_MAX_SOURCES; ++s) /* an entry that promises nothing: the long way for this index */
      if ((R.imask & (1u << s)) && raw_plan_at(ka->raw.src[s].plan, t).rden == 0.0) {
        slow |= mix;
        mix = 0u;
      }
  }
  if (mix) {
#pragma unroll
    for (int q = 0; q < NF; ++q)
      if (mix & (1u << q)) val[q] = R.v0[q] + 0.0; /* the lanes that copy, or have nobody: dv = 0 */
#pragma nounroll
    for (int s = 0; s < RS_MAX_SOURCES; ++s) {
      if (!(R.imask & (1u << s))) continue;
      const RawPlanStep st = raw_plan_at(ka->raw.src[s].plan, t);
#pragma unroll
      for (int q = 0; q < NF; ++q)
        if (mix & (1u << q)) {
          const double cand = R.v0[q] + raw_quot(st.num * R.dv[q], st.den, st.rden);
          if (((R.wsel >> (3 * q)) & 7u) == (uint32_t)s) val[q] = cand;
        }
    }
  }
  if (slow) {
#pragma unroll
    for (int q = 0; q < NF; ++q)
      if (slow & (1u << q)) val[q] = raw_slow_value(ka, raw_field_of(FSET, q), col, t);
  }
}

/* The forcing of index i for the ground wave.  FSET_GROUND: the short- and long-wave radiation are the
 * sky wave's (duo_sky: it tests them and hands them on); here they read 0, inside every bound of CheckValues. */
template <int FSET>
__device__ __forceinline__ Forcing raw_forcing(KernArgs ka, int64_t col, bool live, bool rejected,
                                               RawLerp<raw_nfields(FSET)> &R, int32_t i, bool need_obs) {
  double val[raw_nfields(FSET)];
  raw_values<FSET>(ka, col, live, R, i, need_obs, val);
  Forcing f;
  f.tair = rejected ? raw_miss() : val[0];
  f.vz = val[1];
  f.rhz = val[2];
  f.prec = val[3];
  f.tdew = val[4];
  f.tsurfobs = need_obs ? val[kRawObs] : raw_miss();
  f.sw = FSET == FSET_ALL ? val[FSET == FSET_ALL ? 6 : 0] : 0.0;
  f.lw = FSET == FSET_ALL ? val[FSET == FSET_ALL ? 7 : 0] : 0.0;
  f.phase = -9999; /* InputData.cpp:16: the driver never hands PrecPhase on */
  f.hour = ((const int32_t __attribute__((address_space(4))) *)ka->raw.hour)[i - 1]; /* uniform: a scalar load */
  f.depth = raw_miss(); /* InputData.cpp:18 */
  return f;
}

/* FULL: the FULL feature set without sky view, coupling, a depth stream or tsurfOutputDepth (what the
 * launcher checks): the optional streams, the initialization phase and relaxation.  The ground wave owns
 * what they add to the forcing's share of a step - CheckValues' dew-point test, whether the observation
 * is forced on Tmp(1:2) at an index (it needs the forced Tmp(2) itself, for the flux into layer 3), and
 * RelaxationOperations (src/Relaxation.f90:10-47: the anchors at the end of the initialization, which go
 * to the state block there and then, and the decaying correction behind it). */
enum { SRC_WINDOW = 0, SRC_KNOTS = 1, SRC_RAW = 2 }; /* where the ground wave's forcing comes from */
/* SKYG (with SRC_RAW): the launch has per-point sky view and a THIRD wavefront for it (duo_sky): the short- and
 * long-wave radiation, their share of CheckValues and ModRadiationBySurroundings are that wave's; this one
 * makes the other six variables. */
/* REPLAY (with CPL, SRC_RAW): see duo_surface - this wave restores layers 3..N at the rewind and computes the
 * flux and heat capacities of that step from the restored profile and the end-of-window one (the stale TmpNw),
 * and hands over CheckValues' verdict on the forcing of the index behind the window (ForcingPrep::bad_rw). */
template <int NL, int SRC = SRC_WINDOW, bool FULL = false, bool SKYG = false, bool CPL = false, bool REPLAY = false>
__device__ __forceinline__ void duo_ground(const MathTab &mt, DuoMailT<FULL ? PR_FULL : PR_LEAN> &mail, const StepArgs &a) {
  constexpr bool KNOTS = SRC == SRC_KNOTS, RAW = SRC == SRC_RAW;
  static_assert(FULL || !CPL, "coupling belongs to the FULL feature set");
  static_assert(!RAW || FULL, "the driver's series carry the FULL feature set");
  static_assert(!SKYG || RAW, "a sky wave exists only where the forcing is made from the raw series");
  static_assert(!REPLAY || (CPL && RAW && !SKYG), "replay rounds in this flavour: raw series, no sky view");
  KernArgs ka = kernargs();
  const uint32_t ml = threadIdx.x & 63u; /* lane of the mailbox; `lane` + row0 = the point (REPLAY: through the list) */
  const int64_t listed = REPLAY ? (int64_t)blockIdx.x * 64 + ml : 0;
  const int64_t row0 = REPLAY ? 0 : a.wave_start ? (int64_t)a.wave_start[blockIdx.x] : (int64_t)blockIdx.x * 64;
  const uint32_t lane = REPLAY ? (listed < a.cpl_nlist ? (uint32_t)a.cpl_list[listed] : 0u) : ml;
  const int64_t p = row0 + lane;
  const bool live = REPLAY ? listed < a.cpl_nlist
                           : a.wave_start ? (int32_t)lane < a.wave_cnt[blockIdx.x] : p < a.npoints;
  double Tg[NL - 2]; /* Tmp(3..NL) */
#pragma unroll
  for (int j = 3; j <= NL; ++j) Tg[j - 3] = live ? a.state[(int64_t)(RS_ST_TMP0 + j - 1) * a.np_pad + p] : 0.0;
  const double tbot = live ? ka->pp.tbottom[p] : 0.0;
  mail.v[0][1][ml] = Tg[0];
  const int32_t nsteps = ka->nsteps, t0 = ka->t0;
  /* FULL: as time_loop sets them up */
  int32_t initlen = 0;
  bool relax = false;
  double relax_dt = 0, relax_dv = 0, relax_dr = 0;
  auto relax_targets = [&](double &tairR, double &vzR, double &rhR) { /* through REAL(4): src/InputOutput.f90:19-26 */
    tairR = (double)(float)ka->pp.tair_relax[p];
    vzR = (double)(float)ka->pp.vz_relax[p];
    rhR = (double)(float)ka->pp.rh_relax[p];
  };
  if (FULL && live) {
    initlen = ka->pp.initlen ? ka->pp.initlen[p] : 0;
    if (consts_of(ka).use_relaxation && ka->pp.tair_relax) {
      double tairR, vzR, rhR;
      relax_targets(tairR, vzR, rhR);
      relax = !(tairR < R4(-100.0) || tairR > R4(100.0) || vzR < R4(0.0) || vzR > R4(100.0) ||
                rhR < R4(0.0) || rhR > 110);
      relax_dt = tairR - a.state[(int64_t)RS_ST_TAIR_END * a.np_pad + p];
      relax_dv = vzR - a.state[(int64_t)RS_ST_VZ_END * a.np_pad + p];
      relax_dr = rhR - a.state[(int64_t)RS_ST_RH_END * a.np_pad + p];
    }
  }
  /* CPL: where the point's coupling window starts - from there on the observation is no longer forced on the
   * two top layers (src/InputOutput.f90:120-121), and at that index the layers are saved (src/Coupling.f90:172-210) */
  int32_t cpl_cs = 0x7fffffff, cpl_ce = 0x7ffffff0;
  if (CPL && live) {
    const int32_t cidx = ka->pp.coupling_index ? ka->pp.coupling_index[p] : 0;
    if (consts_of(ka).use_coupling && ka->pp.coupling_index && !(ka->pp.coupling_tsurf[p] < -100 || cidx < 1)) {
      cpl_cs = ((double)cidx <= consts_of(ka).cplLenR) ? 1 : cidx - consts_of(ka).cplLenI;
      cpl_ce = cidx;
    }
  }
  const int64_t rcol_ = (RAW && live) ? (ka->raw.col ? (int64_t)ka->raw.col[p] : p) : 0; /* the point's column of the raw series */
  /* the anchors of an index are stored at the START of that index's step, when the surface wave's
   * verdict on the index before is in the mailbox (a point that has failed never reaches it) */
  bool anchor_due = false;
  double anchor_t = 0, anchor_v = 0, anchor_r = 0;
  double obs_cur = R4(-9999.9); /* the observation forced on Tmp(1:2) at the current index, or missing */
  /* the forcing's share of index `in`, from the forcing in `f` */
  auto prep = [&](const ConstsAS &c, const Forcing &f, int32_t in, double &obs) -> ForcingPrep {
    if (!FULL) return forcing_prep(c, mt, f, in, in < c.SimLen);
    bool bad;
    const bool has_tdew = RAW ? true : KNOTS ? (ka->duo_full_ok & 2) != 0 : ka->f.tdew != nullptr;
    double vz = forcing_prep_head(c, f, in, in < c.SimLen, has_tdew, bad);
    double tair = f.tair, rhz = f.rhz;
    obs = R4(-9999.9);
    if (in < c.SimLen) {
      if ((in <= initlen || c.force_tsurf) && f.tsurfobs > R4(-100.0) && (!CPL || in < cpl_cs)) obs = f.tsurfobs;
      if (relax) {
        if (in == initlen) { /* the anchors: once per point */
          double tairR, vzR, rhR;
          relax_targets(tairR, vzR, rhR);
          relax_dt = tairR - tair;
          relax_dv = vzR - vz;
          relax_dr = rhR - rhz;
          anchor_due = true;
          anchor_t = tair;
          anchor_v = vz;
          anchor_r = rhz;
        }
        if (in > initlen) {
          double e;
          const uint32_t d = (uint32_t)(in - initlen);
          if (c.relax_tab && d <= (uint32_t)c.SimLen) {
            e = ((const double *)c.relax_tab)[d];
          } else {
            const double den = (double)(4.f * 3600.f);
            e = rs_exp(mt, rs_div(-((c.DTSecs * in) - (c.DTSecs * initlen)), den));
          }
          tair = tair - relax_dt * e;
          vz = vz - relax_dv * e;
          rhz = rhz - relax_dr * e;
          if (rhz > R4(100.)) rhz = R4(100.0);
        }
      }
    }
    ForcingPrep q = forcing_prep_tail(c, mt, f, tair, vz, rhz, bad);
    q.tsurfobs = obs;
    if (REPLAY && live && in == cpl_cs && in < c.SimLen) {
      /* the rewind: CheckValues has just seen the index behind the window end (0-based: cpl_ce) - its forcing
       * the long way, once per replay */
      Forcing g = Forcing();
      const int32_t tb = cpl_ce;
      g.tair = ka->raw.status && ka->raw.status[rcol_] != 0 ? raw_miss() : raw_slow_value(ka, RAW_TAIR, rcol_, tb);
      g.vz = raw_slow_value(ka, RAW_VZ, rcol_, tb);
      g.rhz = raw_slow_value(ka, RAW_RHZ, rcol_, tb);
      g.prec = raw_slow_value(ka, RAW_PREC, rcol_, tb);
      g.sw = raw_slow_value(ka, RAW_SW, rcol_, tb);
      g.lw = raw_slow_value(ka, RAW_LW, rcol_, tb);
      g.tdew = raw_slow_value(ka, RAW_TDEW, rcol_, tb);
      q.bad_rw = check_values_forcing(c, g) | (g.tdew < -90) | (g.tdew > R4(100.0));
    }
    return q;
  };
  /* the forcing of the launch's first index, prepared before the first meeting (a lane beyond npoints
   * reads nothing: the windows need only span npoints columns) */
  Forcing nxt = Forcing();
  KnotLerp<FULL> klerp;
  int32_t kcur = -1;
  const int64_t kcol = (KNOTS && live) ? (ka->knot_gather ? (int64_t)ka->knot_gather[p] : p) : 0;
  /* RAW: the point's column of the raw series, read_input's verdict on it, the segment of the first index */
  constexpr int FSET = SKYG ? FSET_GROUND : FSET_ALL;
  RawLerp<raw_nfields(FSET)> rlerp;
  const int64_t rcol = rcol_;
  const bool rejected = RAW && live && ka->raw.status && ka->raw.status[rcol] != 0;
  /* the observation can act at index `in`: SetCurrentValues' own condition (src/InputOutput.f90:116-121),
   * for the whole wavefront */
  auto obs_wanted = [&](const ConstsAS &c, int32_t in) -> bool {
    return c.force_tsurf || __builtin_amdgcn_ballot_w64(live && in <= initlen) != 0ull;
  };
  if (RAW) {
    rlerp.seg = ka->raw.seg0;
    raw_resolve<FSET>(ka, rcol, live, rlerp);
  }
  {
    const ConstsAS &c0 = consts_of(ka);
    if (KNOTS) nxt = knot_forcing<FULL>(ka, kcol, live, klerp, kcur, t0);
    else if (RAW) nxt = raw_forcing<FSET>(ka, rcol, live, rejected, rlerp, t0, obs_wanted(c0, t0));
    else if (live) nxt = load_forcing<FULL, true>(ka, row0, lane, 0);
    duo_put_prep<FULL, SKYG>(mail, 0, ml, prep(c0, nxt, t0, obs_cur));
  }
  duo_meet();
  for (int32_t kv = 0; kv < nsteps; ++kv) {
    asm volatile("" : "+s"(ka));
    const ConstsAS &c = consts_of(ka);
    const int32_t k = __builtin_amdgcn_readfirstlane(kv);
    /* next index's forcing: fetched here, used behind the layers */
    if (!KNOTS && !RAW && k + 1 < nsteps && live) nxt = load_forcing<FULL, true>(ka, row0, lane, k + 1);
    double t2 = mail.v[k & 1][0][ml]; /* Tmp(2) as the last step left it (melting included) */
    /* a failed point takes no further step in any flavour: its Tmp(3..N) stay as the failing index left
     * them (the flag was raised before the barrier that ended that index) */
    if (REPLAY && !mail.failed[ml] && t0 + k == cpl_cs && t0 + k < c.SimLen && live) {
      /* the rewind (uploadDataForCoupling, src/Coupling.f90:245-247): Tmp is restored, TmpNw is not - this
       * step's heat capacities see the end-of-window profile (Tg as it stands), its fluxes the restored one */
      const double *sv = ka->state + (int64_t)RS_ST_CPL_SAVE_TMP0 * ka->np_pad + p;
      const int64_t np = ka->np_pad;
      double cur = sv[(int64_t)2 * np]; /* restored Tmp(3) */
      double Gprev = c.lk[2].condDZ * (cur - sv[np]); /* ... against the restored Tmp(2) */
#pragma unroll
      for (int j = 3; j <= NL; ++j) {
        const double tnext = (j == NL) ? tbot : sv[(int64_t)j * np];
        const double stale = Tg[j - 3];
        Tg[j - 3] = layer_step(c, j, cur, stale, tnext, Gprev, nullptr);
        cur = tnext;
      }
    } else
    if (!mail.failed[ml]) {
      if (CPL && !REPLAY && t0 + k == cpl_cs && t0 + k < c.SimLen && live) { /* saveDataForCoupling: layers 3..N as they stand */
#pragma unroll
        for (int j = 3; j <= NL; ++j) ka->state[(int64_t)(RS_ST_CPL_SAVE_TMP0 + j - 1) * ka->np_pad + p] = Tg[j - 3];
      }
      if (FULL) {
        if (obs_cur > R4(-100.0)) t2 = obs_cur; /* SetCurrentValues has forced Tmp(1:2) at this index */
        if (anchor_due && live) {
          double *st = ka->state;
          st[(int64_t)RS_ST_TAIR_END * ka->np_pad + p] = anchor_t;
          st[(int64_t)RS_ST_VZ_END * ka->np_pad + p] = anchor_v;
          st[(int64_t)RS_ST_RH_END * ka->np_pad + p] = anchor_r;
        }
      }
      double Gprev = c.lk[2].condDZ * (Tg[0] - t2); /* G(2), the expression layer 2 itself evaluates */
#pragma unroll
      for (int j = 3; j <= NL; ++j) {
        const double tj = Tg[j - 3];
        const double tnext = (j == NL) ? tbot : Tg[j - 2];
        Tg[j - 3] = layer_step(c, j, tj, tj, tnext, Gprev, nullptr);
      }
    }
    if (FULL) anchor_due = false;
    mail.v[(k & 1) ^ 1][1][ml] = Tg[0];
    if (k + 1 < nsteps) {
      const int32_t in = t0 + k + 1;
      if (KNOTS) nxt = knot_forcing<FULL>(ka, kcol, live, klerp, kcur, in);
      if (RAW) nxt = raw_forcing<FSET>(ka, rcol, live, rejected, rlerp, in, obs_wanted(c, in));
      duo_put_prep<FULL, SKYG>(mail, (k & 1) ^ 1, ml, prep(c, nxt, in, obs_cur));
    }
    duo_meet();
  }
  if (live) {
#pragma unroll
    for (int j = 3; j <= NL; ++j) a.state[(int64_t)(RS_ST_TMP0 + j - 1) * a.np_pad + p] = Tg[j - 3];
  }
}

/* The sky wave: third wavefront of the workgroup in launches with per-point sky view whose forcing comes from
 * the raw series.  CheckValues' tests of the radiation, the SW_dir clamp and ModRadiationBySurroundings with the
 * sun's position (src/InputOutput.f90:45-84, src/ModRadiation.f90:7-73, src/SunPosition.f90:123-193; examples/
 * example1/src/Simulation.f90:151-162) depend on the forcing and the point's geometry alone: like the rest of
 * the forcing's share of a step they are worked out one index AHEAD, here - four variables from the raw series
 * (global and direct short wave, long wave, net long wave), the tests, the sky view - and the surface wave
 * receives the radiation as the sky view leaves it (mailbox values 9 and 10) plus this wave's flags.  Done on
 * the ground wave, beside thirteen layers and the other six variables, the sky view spilled 86-106 registers;
 * on the surface wave it lengthens the step's serial chain. */
__device__ __forceinline__ void duo_sky(DuoMailT<PR_FULL> &mail, uint32_t *skyfl, const StepArgs &a) {
  KernArgs ka = kernargs();
  const uint32_t lane = threadIdx.x & 63u;
  const int64_t row0 = a.wave_start ? (int64_t)a.wave_start[blockIdx.x] : (int64_t)blockIdx.x * 64;
  const int64_t p = row0 + lane;
  const bool live = a.wave_start ? (int32_t)lane < a.wave_cnt[blockIdx.x] : p < a.npoints;
  double skyv = R4(1.0), sinlat = 0, coslat = 0, lonrad = 0, coslon = 1.0, sinlon = 0;
  bool sky_on = false;
  uint32_t hcol = 0;
  if (live) { /* as time_loop<SKY> sets the point up */
    skyv = ka->pp.sky_view[p];
    sky_on = (skyv < R4(1.0) && skyv > R4(-0.01));
    if (sky_on) {
      sinlat = ka->pp.sin_lat[p];
      coslat = ka->pp.cos_lat[p];
      lonrad = ka->pp.lon_rad[p];
      coslon = ::cos(lonrad);
      sinlon = ::sin(lonrad);
    }
    hcol = ka->pp.horizon_index ? (uint32_t)ka->pp.horizon_index[p] : (uint32_t)p;
  }
  const int64_t rcol = live ? (ka->raw.col ? (int64_t)ka->raw.col[p] : p) : 0;
  RawLerp<4> R;
  R.seg = ka->raw.seg0;
  raw_resolve<FSET_SKY>(ka, rcol, live, R);
  const int32_t nsteps = ka->nsteps, t0 = ka->t0;
  auto put = [&](int buf, int32_t in) {
    const ConstsAS &c = consts_of(ka);
    double val[4];
    raw_values<FSET_SKY>(ka, rcol, live, R, in, false, val);
    double sw = val[0], lw = val[1], sw_dir = val[2], lw_net = val[3];
    bool bad = false, stop = false;
    if (in < c.SimLen) {
      /* check_values_forcing's tests of SW and LW (the ground wave runs the others with 0 in their place) */
      bad = (sw < c.chk[1]) | (lw < c.chk[1]) | (sw > c.chk[5]) | (lw > c.chk[6]);
      if (sky_on && (sw_dir < R4(-0.1) || sw_dir > R4(4000.0) || lw_net < R4(-1000.0) || lw_net > R4(1000.0)))
        bad = true; /* src/InputOutput.f90:68-74 */
      if (sw_dir > sw) sw_dir = sw; /* :75-77 */
    }
    if (sky_on) {
      const double *sunrow = ka->f.sun + (int64_t)(in - t0) * RS_SUN_COLS;
      if (!sky_view_radiation(sunrow, sinlat, coslat, lonrad, coslon, sinlon, skyv, ka->pp.albedo_surroundings,
                              ka->pp.horizons ? ka->pp.horizons + (ka->pp.horizons_by_point ? (int64_t)hcol * 360 : (int64_t)hcol) : nullptr,
                              ka->pp.horizons_by_point ? (int64_t)1 : ka->np_pad, sw, sw_dir, lw, lw_net))
        stop = true; /* the reference would `stop` the process here: the point is failed at this index, checked or not */
    }
    mail.prep[buf][PR_SW][lane] = sw;
    mail.prep[buf][PR_LW][lane] = lw;
    skyfl[buf * 64 + lane] = (bad ? 1u : 0u) | (stop ? 4u : 0u);
  };
  put(0, t0);
  duo_meet();
  for (int32_t kv = 0; kv < nsteps; ++kv) {
    asm volatile("" : "+s"(ka));
    const int32_t k = __builtin_amdgcn_readfirstlane(kv);
    if (k + 1 < nsteps) put((k & 1) ^ 1, t0 + k + 1);
    duo_meet();
  }
}

/* SRC: SRC_WINDOW / SRC_KNOTS / SRC_RAW.  SKY: per-point sky view - on the surface wave (forcing windows)
 * or, with SRC_RAW, on a third wavefront (duo_sky). */
template <int NL, bool SCORE, int SRC = SRC_WINDOW, bool FULL = false, bool SKY = false, bool CPL = false,
          bool REPLAY = false>
#ifndef RS_DUO_LEAN_WAVES
/* The LEAN instances' 16 192 B of LDS would let ten workgroups onto a CU, and at five wavefronts per SIMD they fit 96
 * registers with 9-14 spilled - measured (tools/experiments/r6_ab3.sh, profiles/r06_lean_waves_per_simd.txt): 1 M
 * points 2.513e10 at five against 2.560e10 at four (2.529e10 with round 5's mailbox of eleven values), level at
 * 250 000 and 125 000 points.  The pass is bound by what the wavefronts issue, not by how many of them wait. */
#define RS_DUO_LEAN_WAVES 4
#endif
__global__ void __launch_bounds__((SKY && SRC == SRC_RAW) ? 192 : 128, REPLAY ? 3 : FULL ? 4 : RS_DUO_LEAN_WAVES) step_kernel_duo(const StepArgs a) {
  /* (REPLAY at three waves per SIMD: with the rewind's code the instance spilled 50 registers at four) */
  static_assert(!CPL || SRC == SRC_RAW, "coupling in the two-wavefront flavour: the driver path's lock-step chunks");
  constexpr bool SKYG = SKY && SRC == SRC_RAW;
  __shared__ double math_lds[RS_MATH_LDS_DOUBLES];
  __shared__ DuoMailT<FULL ? PR_FULL : PR_LEAN> mail;
  __shared__ uint32_t skyfl[SKYG ? 2 * 64 : 1]; /* duo_sky's flags, [buffer][lane] */
  const MathTab mt = fill_math_tables(math_lds);
  __syncthreads();
  /* no early return: the wavefronts walk to every barrier, lanes beyond npoints are dead weight.
   * (Dealing the roles by the parity of the hardware wave slot or of the workgroup index, so that every
   * SIMD hosts both kinds, was measured: 1.08e10 and 1.13e10 against 1.12e10 with fixed roles at
   * 125 000 points - nothing to gain.) */
  if (a.wave_start && a.wave_cnt[blockIdx.x] == 0) return; /* a spare workgroup of the wave table: all wavefronts leave */
  if (threadIdx.x < 64) {
    if (a.surface_prio) __builtin_amdgcn_s_setprio(1); /* the longer chain of the two issues first (StepArgs::surface_prio) */
    duo_surface<NL, SCORE, FULL, SKY && !SKYG, SKYG, SRC == SRC_RAW, CPL, REPLAY>(mt, mail, a, skyfl);
  } else if (!SKYG || threadIdx.x < 128) {
    duo_ground<NL, SRC, FULL, SKYG, CPL, REPLAY>(mt, mail, a);
  } else {
    if constexpr (SKYG) duo_sky(mail, skyfl, a);
  }
}

/* FULL feature set + sky view in lock step, LDS profile (any NLayers). */
template <bool DIAG = false>
__global__ void __launch_bounds__(kBlock, 3) step_kernel_sky(const StepArgs a) {
  extern __shared__ double lds[]; /* [NLayers][kBlock] */
  __shared__ double math_lds[RS_MATH_LDS_DOUBLES];
  const MathTab mt = fill_math_tables(math_lds);
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= a.npoints) return;
  LdsProfile T{lds + threadIdx.x, consts_of(&a).NLayers};
  Scalars s;
  int32_t score = 0;
  load_state<true>(a.state, a.np_pad, p, T, s);
  time_loop<true, LdsProfile, true, true, false, false, false, DIAG>(mt, T, s, score);
  store_state<true>(a.state, a.np_pad, p, T, s);
  a.state[(int64_t)RS_ST_BLSCORE * a.np_pad + p] = bl_score_key(score, s);
}

/* The same for NLayers = 15 with the hybrid profile at W waves per SIMD (rs_launch_step_sky). */
template <int W>
__global__ void __launch_bounds__(kBlock, W) step_kernel_sky_h(const StepArgs a) {
  __shared__ double math_lds[RS_MATH_LDS_DOUBLES];
  __shared__ double prof_lds[(15 - RS_HYBRID_REG) * kBlock];
  const MathTab mt = fill_math_tables(math_lds);
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= a.npoints) return;
  HybridProfile<15, RS_HYBRID_REG> T;
  T.col = prof_lds + threadIdx.x;
  Scalars s;
  int32_t score = 0;
  load_state<true>(a.state, a.np_pad, p, T, s);
  time_loop<true, HybridProfile<15, RS_HYBRID_REG>, true>(mt, T, s, score);
  store_state<true>(a.state, a.np_pad, p, T, s);
  a.state[(int64_t)RS_ST_BLSCORE * a.np_pad + p] = bl_score_key(score, s);
}

/* FULL feature set + coupling in lock step (time_loop<CPL>): everything of a coupled run except
 * the replays.  LDS profile (any NLayers).  SKY: with the sky-view radiation of the points that have
 * one (src/ModRadiation.f90:7-73) - the forcing windows are read-only here, so what the reference
 * saves and restores of SW / SW_dir / LW around a window (src/Coupling.f90:204-208,249-253) is simply
 * the window as it stands. */
template <bool SKY>
__global__ void __launch_bounds__(kBlock, 3) step_kernel_cpl(const StepArgs a) {
  extern __shared__ double lds[]; /* [NLayers][kBlock] */
  __shared__ double math_lds[RS_MATH_LDS_DOUBLES];
  const MathTab mt = fill_math_tables(math_lds);
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= a.npoints) return;
  LdsProfile T{lds + threadIdx.x, consts_of(&a).NLayers};
  Scalars s;
  int32_t score = 0;
  load_state<true>(a.state, a.np_pad, p, T, s);
  time_loop<true, LdsProfile, SKY, true, true>(mt, T, s, score);
  store_state<true>(a.state, a.np_pad, p, T, s);
  a.state[(int64_t)RS_ST_BLSCORE * a.np_pad + p] = bl_score_key(score, s); /* parked lanes: cheap */
}

/* The same two kernels for NLayers = 15 with the hybrid profile (layers 1-RS_HYBRID_REG in
 * registers, the rest in LDS) at W waves per SIMD. */
template <int W, bool SKY = false>
__global__ void __launch_bounds__(kBlock, W) step_kernel_cpl_h(const StepArgs a) {
  __shared__ double math_lds[RS_MATH_LDS_DOUBLES];
  __shared__ double prof_lds[(15 - RS_HYBRID_REG) * kBlock];
  const MathTab mt = fill_math_tables(math_lds);
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= a.npoints) return;
  HybridProfile<15, RS_HYBRID_REG> T;
  T.col = prof_lds + threadIdx.x;
  Scalars s;
  int32_t score = 0;
  load_state<true>(a.state, a.np_pad, p, T, s);
  time_loop<true, HybridProfile<15, RS_HYBRID_REG>, SKY, true, true>(mt, T, s, score);
  store_state<true>(a.state, a.np_pad, p, T, s);
  a.state[(int64_t)RS_ST_BLSCORE * a.np_pad + p] = bl_score_key(score, s);
}

template <int W, bool SKY = false>
__global__ void __launch_bounds__(kBlock, W) step_kernel_cpl_replay_h(const StepArgs a) {
  __shared__ double math_lds[RS_MATH_LDS_DOUBLES];
  __shared__ double prof_lds[(15 - RS_HYBRID_REG) * kBlock];
  const MathTab mt = fill_math_tables(math_lds);
  __syncthreads();
  const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (g >= (int64_t)a.cpl_nlist) return;
  const int64_t p = (int64_t)a.cpl_list[g];
  if (a.cpl_prio) __builtin_amdgcn_s_setprio(2);
  HybridProfile<15, RS_HYBRID_REG> T;
  T.col = prof_lds + threadIdx.x;
  Scalars s;
  int32_t score = 0;
  /* cpl_inner replays of the point's window in this launch, as long as its Coupling_control asks for
   * another (start_coupling_again, bit 0 of the flags it has just stored): the secant iteration of
   * src/Coupling.f90:363-449 is per point - only the launches were sequential.  The state goes through
   * the state block between two replays exactly as it does between two rounds. */
  for (int32_t rnd = 0;; ++rnd) {
    load_state<true>(a.state, a.np_pad, p, T, s);
    time_loop<true, HybridProfile<15, RS_HYBRID_REG>, SKY, false, true, true>(mt, T, s, score, (uint32_t)p);
    store_state<true>(a.state, a.np_pad, p, T, s);
    if (rnd + 1 >= a.cpl_inner) break;
    if ((((int32_t)a.state[(int64_t)RS_ST_CPL_FLAGS * a.np_pad + p]) & 1) == 0) break;
  }
}

/* One replay round in lock step over the compacted list (time_loop<REPLAY>). */
template <bool SKY>
__global__ void __launch_bounds__(kBlock, 3) step_kernel_cpl_replay(const StepArgs a) {
  extern __shared__ double lds[]; /* [NLayers][kBlock] */
  __shared__ double math_lds[RS_MATH_LDS_DOUBLES];
  const MathTab mt = fill_math_tables(math_lds);
  __syncthreads();
  const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (g >= (int64_t)a.cpl_nlist) return;
  const int64_t p = (int64_t)a.cpl_list[g];
  if (a.cpl_prio) __builtin_amdgcn_s_setprio(2);
  LdsProfile T{lds + threadIdx.x, consts_of(&a).NLayers};
  Scalars s;
  int32_t score = 0;
  for (int32_t rnd = 0;; ++rnd) { /* as in step_kernel_cpl_replay_h */
    load_state<true>(a.state, a.np_pad, p, T, s);
    time_loop<true, LdsProfile, SKY, false, true, true>(mt, T, s, score, (uint32_t)p);
    store_state<true>(a.state, a.np_pad, p, T, s);
    if (rnd + 1 >= a.cpl_inner) break;
    if ((((int32_t)a.state[(int64_t)RS_ST_CPL_FLAGS * a.np_pad + p]) & 1) == 0) break;
  }
}

/* Coupled variant: LDS profile (any NLayers), FULL feature set + coupling. */
template <bool DIAG = false>
__global__ void __launch_bounds__(kBlock, RS_CPL_WAVES) step_kernel_coupled(const StepArgs a) {
  extern __shared__ double lds[]; /* [NLayers][kBlock] */
  __shared__ double math_lds[RS_MATH_LDS_DOUBLES];
  const MathTab mt = fill_math_tables(math_lds);
  __syncthreads();
  const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (g >= (a.cpl_list ? (int64_t)a.cpl_nlist : a.npoints)) return;
  const int64_t p = a.cpl_list ? (int64_t)a.cpl_list[g] : g;
  if (a.cpl_prio) __builtin_amdgcn_s_setprio(2);
  const int NLc = consts_of(&a).NLayers;
  LdsProfile T{lds + threadIdx.x, NLc};
  Scalars s;
  Coupling q;
  load_state<true>(a.state, a.np_pad, p, T, s);
  load_coupling(a.state, a.np_pad, p, q);
  time_loop_coupled<LdsProfile, DIAG>(mt, T, s, q, a.state, a.np_pad, p);
  store_state<true, LdsProfile, true>(a.state, a.np_pad, p, T, s);
  store_coupling(a.state, a.np_pad, p, q);
}

/* Device part of Initialization (src/Initialization.f90:65-147): initial
 * profile (initTemp :238-287), surface state (initSurf :290-308), T4Melt
 * (condInit :518), albedo (InitParam :339).  The first CalcBLCondAndLE call
 * (:138-139) and the first CalcHCapHCond (:109) only produce values that are
 * overwritten before they are read, so they are not executed. */
__global__ void __launch_bounds__(kBlock) init_kernel(const InitArgs a) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= a.npoints) return;
  const ConstsAS &c = consts_of(&a);
  const int N = c.NLayers;
  const double tair = a.f.tair[p];
  const double tobs = a.f.tsurfobs ? a.f.tsurfobs[p] : R4(-9999.9);
  const double depth = a.f.depth ? a.f.depth[p] : R4(-9999.9);
  const double tbot = a.pp.tbottom[p];
  const double t4 = (tobs > -100) ? tobs : tair;
  const int64_t np = a.np_pad;
  double *st = a.state;
  for (int i = 1; i <= 4; ++i) st[(int64_t)(RS_ST_TMP0 + i - 1) * np + p] = t4;
  for (int i = 5; i <= N; ++i)
    st[(int64_t)(RS_ST_TMP0 + i - 1) * np + p] =
        t4 + (tbot - t4) / (c.ZDpth[N + 1] - c.ZDpth[4]) * (c.ZDpth[i] - c.ZDpth[4]);
  for (int i = N + 1; i <= RS_MAX_LAYERS; ++i) st[(int64_t)(RS_ST_TMP0 + i - 1) * np + p] = 0.0;
  double tsurf;
  if (depth >= 0) {
    /* getTempAtDepth on the fresh profile, src/Initialization.f90:129-136 */
    if (fabs(depth - R4(0.0)) < R4(0.00001)) {
      tsurf = t4;
    } else if (depth > c.ZDpth[N + 1]) {
      tsurf = tbot;
    } else {
      tsurf = 0.0;
      for (int k = 1; k <= N; ++k) {
        if (depth > c.ZDpth[k] && depth <= c.ZDpth[k + 1]) {
          const double tk = st[(int64_t)(RS_ST_TMP0 + k - 1) * np + p];
          const double tk1 = (k == N) ? tbot : st[(int64_t)(RS_ST_TMP0 + k) * np + p];
          tsurf = tk + (depth - c.ZDpth[k]) * (tk1 - tk) / (c.ZDpth[k + 1] - c.ZDpth[k]);
          break;
        }
      }
    }
  } else {
    tsurf = (t4 + t4) / R4(2.0);
  }
  st[(int64_t)RS_ST_TNW1 * np + p] = t4;
  st[(int64_t)RS_ST_TNW2 * np + p] = t4;
  st[(int64_t)RS_ST_TSURF * np + p] = tsurf;
  st[(int64_t)RS_ST_WAT * np + p] = R4(0.0);
  st[(int64_t)RS_ST_SNOW * np + p] = R4(0.0);
  st[(int64_t)RS_ST_ICE * np + p] = R4(0.0);
  st[(int64_t)RS_ST_ICE2 * np + p] = R4(0.0);
  st[(int64_t)RS_ST_DEP * np + p] = R4(0.0);
  st[(int64_t)RS_ST_Q2MELT * np + p] = R4(0.0);
  st[(int64_t)RS_ST_T4MELT * np + p] = c.T4Melt0;
  st[(int64_t)RS_ST_ALBEDO * np + p] = c.Albedo0;
  st[(int64_t)RS_ST_VERYCOLD * np + p] = 0.0;
  st[(int64_t)RS_ST_FAILED * np + p] = 0.0;
  st[(int64_t)RS_ST_TAIR_END * np + p] = R4(-99.9); /* src/Initialization.f90:377-379 */
  st[(int64_t)RS_ST_VZ_END * np + p] = R4(-99.9);
  st[(int64_t)RS_ST_RH_END * np + p] = R4(-99.9);
  st[(int64_t)RS_ST_BLSCORE * np + p] = 0.0;
  /* initCoupling, src/Coupling.f90:144-169 */
  st[(int64_t)RS_ST_CPL_ITER * np + p] = 0.0;
  st[(int64_t)RS_ST_CPL_FLAGS * np + p] = 0.0;
  st[(int64_t)RS_ST_CPL_TABOVE * np + p] = R4(-9999.0);
  st[(int64_t)RS_ST_CPL_TBELOW * np + p] = R4(-9999.0);
  st[(int64_t)RS_ST_CPL_RADCOEFF * np + p] = R4(1.0);
  st[(int64_t)RS_ST_CPL_RCABOVE * np + p] = R4(-9999.0);
  st[(int64_t)RS_ST_CPL_RCBELOW * np + p] = R4(-9999.0);
  st[(int64_t)RS_ST_CPL_RCPREV * np + p] = R4(1.0);
  st[(int64_t)RS_ST_CPL_SWCOF * np + p] = R4(1.0);
  st[(int64_t)RS_ST_CPL_LWCOF * np + p] = R4(1.0);
  st[(int64_t)RS_ST_CPL_SWCORR * np + p] = R4(0.0);
  st[(int64_t)RS_ST_CPL_LWCORR * np + p] = R4(0.0);
  st[(int64_t)RS_ST_CPL_TEND1 * np + p] = 0.0;
  st[(int64_t)RS_ST_CPL_LASTOBS * np + p] = a.pp.coupling_tsurf ? a.pp.coupling_tsurf[p] : R4(-9999.0);
  st[(int64_t)RS_ST_CPL_RESUME * np + p] = 1.0;
}

/* Hourly knots of the synthetic workload, [knot][RS_KNOT_FIELDS][np_pad]. */
__global__ void __launch_bounds__(kBlock) synth_knots_kernel(const KnotArgs a) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= a.npoints) return;
  /* with a plan order (rs_hip_plan_order) slot p holds point order[p] */
  const int64_t gid = a.spec.point_offset + (a.spec.order ? (int64_t)a.spec.order[p] : p);
  /* one thread makes all the knots of its point: the 17 draws the knots share are hashed once */
  const RsSynthPoint pc = rs_sy_point(a.spec.seed, gid);
  for (int32_t y = 0; y < a.nknots; ++y) {
    const RsSynthKnot q = rs_sy_knot_of(&pc, a.spec.seed, gid, a.k0 + y, a.spec.start_hour);
    double *base = a.knots + ((int64_t)y * RS_KNOT_FIELDS) * a.np_pad + p;
    base[0 * a.np_pad] = q.tair;
    base[1 * a.np_pad] = q.tdew;
    base[2 * a.np_pad] = q.vz;
    base[3 * a.np_pad] = q.rhz;
    base[4 * a.np_pad] = q.prec;
    base[5 * a.np_pad] = q.sw;
    base[6 * a.np_pad] = q.lw;
    base[7 * a.np_pad] = q.tsurf0;
    base[8 * a.np_pad] = (double)q.phase;
  }
}

/* Knots -> step resolution, the device twin of the reference driver's
 * interpolation (examples/example1/src/JsonSource.cpp:115-172): linear between
 * knots, PrecPhase from the later knot.  One thread = one point over one knot
 * interval (blockIdx.y): the two knot rows are read once, then up to
 * steps_per_knot rows are streamed out, each a coalesced 2-KiB store per block
 * and field. */
/* TDEW/OBS/DEPTH: which optional streams the window has (compile time, so that the loop over the
 * time indices is ONE basic block: the stores then take the scalar row base + 32-bit lane offset
 * form, LaneOff).  At r = 0 the interpolation adds +-0.0 to the knot value, which returns it
 * unchanged (no knot value is -0.0: rs_synth.h), so the first index needs no case of its own. */
/* (the optional streams - dew point, surface observation, output depth - are there or not for the whole launch:
 * uniform tests of the window's pointers; they were three template parameters, eight instances, until round 6) */
__global__ void __launch_bounds__(kBlock) expand_kernel(const ExpandArgs a) {
  const bool TDEW = a.f.tdew != nullptr, OBS = a.f.tsurfobs != nullptr, DEPTH = a.f.depth != nullptr;
  __builtin_amdgcn_s_setprio(3); /* a link of the chain between two step launches of a block: all of it is waited for */
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= a.npoints) return;
  const int32_t k = a.kfirst + (int32_t)blockIdx.y;        /* knot interval [k, k+1) */
  int32_t tlo = k * a.spk, thi = tlo + a.spk;              /* 0-based time range */
  if (tlo < a.t0 - 1) tlo = a.t0 - 1;
  if (thi > a.t0 - 1 + a.nsteps) thi = a.t0 - 1 + a.nsteps;
  if (tlo >= thi) return;
  const int64_t kcol = a.gather ? (int64_t)a.gather[p] : p; /* knots kept in point order: gather */
  const double *ka = a.knots + ((int64_t)(k - a.k0) * RS_KNOT_FIELDS) * a.np_pad + kcol;
  const double *kb = ka + (int64_t)RS_KNOT_FIELDS * a.np_pad;
  const bool need_b = (thi - 1) > k * a.spk; /* some r > 0 in range */
  double v0[7], dv[7];
  /* rs_sy_lerp, k0 + (secs * (k1 - k0)) / span, with the difference taken once per interval and
   * the division by the uniform span as rs_div_u (rs_math.hpp: exact for a denominator whose
   * reciprocal is correctly rounded; the numerator is never -0.0 here because k1 - k0 is not) */
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
  const int64_t col0 = (int64_t)blockIdx.x * kBlock; /* the workgroup's first column: uniform */
  for (int32_t t = tlo; t < thi; ++t) {
    const int32_t r = t - k * a.spk;
    const double secs = (double)r;
    const int64_t row = (int64_t)(t - (a.t0 - 1)) * a.f.t_stride + col0;
    const LaneOff L(threadIdx.x);
    double v[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) v[q] = v0[q] + rs_div_u(secs * dv[q], span, a.r_spk);
    L.st((double *)a.f.tair + row, v[0]);
    if (TDEW) L.st((double *)a.f.tdew + row, v[1]);
    L.st((double *)a.f.vz + row, v[2]);
    L.st((double *)a.f.rhz + row, v[3]);
    L.st((double *)a.f.prec + row, v[4]);
    L.st((double *)a.f.sw + row, v[5]);
    L.st((double *)a.f.lw + row, v[6]);
    if (OBS) L.st((double *)a.f.tsurfobs + row, (t == 0) ? ts0 : -9999.9);
    if (DEPTH) L.st((double *)a.f.depth + row, -9999.9);
    L.st((int32_t *)a.f.precphase + row, (r == 0) ? ph0 : ph1);
  }
  if (p == 0 && !a.f.hour_pstride)
    for (int32_t t = tlo; t < thi; ++t)
      ((int32_t *)a.f.hour)[t - (a.t0 - 1)] = rs_sy_hour(t + 1, a.spk, a.start_hour);
}

/* Unit-test kernel for rs_math.hpp (tests/test_hip_math.py). fn: 0 exp, 1 log, 2 the bare division
 * x[i] / x[n + i], 3 the bare square root. */
__global__ void __launch_bounds__(kBlock) math_test_kernel(int fn, int64_t n, const double *x,
                                                           double *y) {
  __shared__ double math_lds[RS_MATH_LDS_DOUBLES];
  const MathTab mt = fill_math_tables(math_lds);
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  if (fn == 2) y[i] = rs_div(x[i], x[n + i]);
  else if (fn == 3) y[i] = rs_sqrt(x[i]);
  else y[i] = fn == 0 ? rs_exp(mt, x[i]) : rs_log(mt, x[i]);
}

/* Tdew <-> RH completion of a raw series, examples/example1/src/JsonSource.cpp:288-295 with
 * CalcTdewOrRH of MeteorologyTools.cpp:12-51.  C++ source: the literals are true doubles here
 * (unlike the Fortran side's REAL(4) ones).  exp/log are the glibc-exact ones; the divisions
 * are the compiler's IEEE expansion (denominators can approach 0 on the RH branch). */
__device__ __forceinline__ double calc_tdew_or_rh(const MathTab &mt, double t, double tdew,
                                                  double rh) {
  const double Alphaw = 17.269, Alphai = 21.875, Betaw = 237.3, Betai = 265.5, AFact = 0.61078;
  const double Alpha = (t >= 0.0) ? Alphaw : Alphai;
  const double Beta = (t >= 0.0) ? Betaw : Betai;
  const double EsatT = AFact * rs_exp(mt, Alpha * t / (t + Beta));
  if (!(tdew != tdew) && tdew > -1000) {
    const double EsatTD = AFact * rs_exp(mt, Alpha * tdew / (tdew + Beta));
    const double x = (EsatTD / EsatT) * 100.0;
    return (100.0 < x) ? 100.0 : x; /* std::min(x, 100.0) */
  }
  if (!(rh != rh) && rh > -1) {
    const double Epr = 0.01 * rh * EsatT;
    const double XX = rs_log(mt, Epr / AFact);
    return Beta * XX / (Alpha - XX);
  }
  return __builtin_nan("");
}

__global__ void __launch_bounds__(kBlock) humidity_fill_kernel(const double *__restrict__ tair,
                                                               double *tdew, double *rhz,
                                                               int64_t n) {
  __shared__ double math_lds[RS_MATH_LDS_DOUBLES];
  const MathTab mt = fill_math_tables(math_lds);
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const double t = tair[i];
  double td = tdew[i], rh = rhz[i];
  if (td < -100 && rh > -100 && t > -100) td = calc_tdew_or_rh(mt, t, -9999.9, rh);
  if (rh < -100 && td > -100 && t > -100) rh = calc_tdew_or_rh(mt, t, td, -9999.9);
  tdew[i] = td;
  rhz[i] = rh;
}

__global__ void __launch_bounds__(kBlock) count_failed_kernel(const double *st, int64_t np_pad,
                                                              int64_t npoints,
                                                              unsigned long long *out) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool bad = p < npoints && st[(int64_t)RS_ST_FAILED * np_pad + p] != 0.0;
  const unsigned long long m = __ballot(bad);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, (unsigned long long)__popcll(m));
}

/* Engine clock as the shader sees it: one wavefront reads the shader-clock counter (s_memtime) and
 * the constant 100 MHz counter (s_memrealtime) `spin_us` apart; out[0..1] = the two deltas.  Enqueued
 * beside the step kernels, it measures the clock the chip holds UNDER that load (bench.py). */
__global__ void __launch_bounds__(64) clock_probe_kernel(uint64_t *out, uint32_t spin_us) {
  const uint64_t c0 = __builtin_readcyclecounter();
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  uint64_t r1 = r0;
  /* bounded: at most RS_CLOCK_PROBE_MAX_US (the launcher clamps spin_us) and at most 1 << 16 naps of
   * ~0.9 us - on a part whose real-time counter stands still the wave leaves with out[1] = 0, which
   * bench.py reads as "clock not readable" */
  bool ok = false;
  for (int it = 0; it < (1 << 16); ++it) {
    if (r1 - r0 >= (uint64_t)spin_us * 100ull) {
      ok = true;
      break;
    }
    __builtin_amdgcn_s_sleep(32);
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const uint64_t c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    out[0] = c1 - c0;
    out[1] = ok ? r1 - r0 : 0;
  }
}

/* Coupling windows of the points that ask for a replay (start_coupling_again): out[0] = min
 * couplingStartI, out[1] = max couplingEndI (initialised by the caller to INT_MAX / 0).  What
 * rs_hip_cpl_replay checks the caller's window against. */
__global__ void __launch_bounds__(kBlock) cpl_window_bounds_kernel(const StepArgs a, int32_t *out) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= a.npoints) return;
  const ConstsAS &c = consts_of(&a);
  if ((((int32_t)a.state[(int64_t)RS_ST_CPL_FLAGS * a.np_pad + p]) & 1) == 0) return;
  const int32_t cidx = a.pp.coupling_index[p];
  if (a.pp.coupling_tsurf[p] < -100 || cidx < 1) return;
  const int32_t cs = ((double)cidx <= c.cplLenR) ? 1 : cidx - c.cplLenI;
  atomicMin(&out[0], cs);
  atomicMax(&out[1], cidx);
}

/* Sort key of rs_hip_recluster_forecast (include/roadsurf.h): CalcBLCondAndLE's fixed point
 * (src/BoundaryLayer.f90:64-96) run at the preview times of the NEXT window, with the carried
 * surface temperature moved along with the air temperature.  A predictor: single precision,
 * hardware reciprocal/sqrt/log - it only orders the slots, no model value depends on it. */
__global__ void __launch_bounds__(kBlock) forecast_key_kernel(const ForecastArgs a) {
  __builtin_amdgcn_s_setprio(3); /* a link of the chain between two step launches of a plan: little work, all of it waited for */
  const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (s >= a.npoints) return;
  const ConstsAS &c = consts_of(&a);
  const int64_t np = a.np_pad;
  auto st = [&](int row) -> double {
    return a.f32 ? (double)((const float *)a.state)[(int64_t)row * np + s]
                 : ((const double *)a.state)[(int64_t)row * np + s];
  };
  const double ts_now = st(RS_ST_TSURF);
  const bool has_snow = st(RS_ST_SNOW) > 0.0, has_ice = st(RS_ST_ICE) > 0.0 || st(RS_ST_ICE2) > 0.0,
             has_wet = st(RS_ST_WAT) > 0.0 || st(RS_ST_DEP) > 0.0;
  const int32_t cover = (has_snow || has_ice || has_wet) ? 1 : 0;
  /* which storage branches the point will take (src/Storage.f90): snow is the longest chain */
  const int32_t sclass = has_snow ? 3 : has_ice ? 2 : has_wet ? 1 : 0;
  const int64_t pq = a.pv.index ? (int64_t)a.pv.index[s] : s; /* preview rows in point order */
  auto preview = [&](const double *const *row, const double *const *row_b, int q) -> double {
    const double v = row[q][pq];
    return row_b[q] ? v + a.pv.w[q] * (row_b[q][pq] - v) : v; /* RsPreview::tair_b: between two rows */
  };
  const double ta_now = a.pv.tair_now ? a.pv.tair_now[pq] : preview(a.pv.tair, a.pv.tair_b, 0);
  const double stab_num = -c.VK_Const * c.ZRefT * c.Grav;
  int32_t unst = 0, farc = 0, extra = 0, maxtrip = 5;
  float stab_lo = 2.f, stab_hi = -2.f; /* the converged stability parameter of the previews, after its clamp */
  for (int q = 0; q < a.pv.n; ++q) {
    const double ta = preview(a.pv.tair, a.pv.tair_b, q);
    double vz = preview(a.pv.vz, a.pv.vz_b, q);
    const double hour = (double)a.pv.hour[q];
    const double calm = (hour >= c.NightOn || hour <= c.NightOff) ? c.CalmLimNgt : c.CalmLimDay;
    if (vz < calm) vz = calm;
    const double ts = ts_now + a.pv.alpha * (ta - ta_now);
    const double TaK = ta + R4(273.15);
    const double dens = R4(100000.0) / (R4(287.05) * TaK);
    const double hcap = R4(1005.0) + (TaK - R4(250.0)) * (TaK - R4(250.0)) / R4(3364.);
    const double avc = hcap * dens;
    /* the iteration itself in single precision with the hardware's reciprocal, square root and
     * logarithm: a predictor needs the regime and the pass count, not the last bit (the exit test
     * is 1e-3 on a conductance of order 1..50) */
    const float dT = (float)(ts - ta), den0 = (float)(avc * TaK), vkvz = (float)(c.VK_Const * vz),
                avk = (float)(avc * c.VK_Const), snum = (float)stab_num, lU = (float)c.logUstar,
                lC = (float)c.logCond;
    float psim = 0.f, psih = 0.f, bl = 0.f, stab_last = 0.f;
    int32_t nnear = 0, nfar = 0, j = 1;
    for (; j <= RS_BL_MAXIT; ++j) {
      const float old = bl;
      const float us = vkvz * __builtin_amdgcn_rcpf(lU + psim);
      bl = avk * us * __builtin_amdgcn_rcpf(lC + psih);
      float stab = snum * bl * dT * __builtin_amdgcn_rcpf(den0 * (us * us * us));
      if (stab > 1.f) stab = 1.f;
      stab_last = stab;
      if (stab > 0.f) {
        psih = 4.7f * stab;
        psim = psih;
      } else {
        const float arg = (1.f + __builtin_amdgcn_sqrtf(1.f - 16.f * stab)) * 0.5f;
        /* glibc's log takes its polynomial path for 1 - 2^-4 <= x < 1 + 0x1.09p-4, the table
         * path otherwise (rs_math.hpp): a wavefront with both kinds of lanes pays for both */
        if (arg < 1.064453125f) ++nnear; else ++nfar;
        psih = -2.f * __logf(arg);
        psim = 0.6f * psih;
      }
      if (j >= 5 && __builtin_fabsf(bl - old) < 0.001f) break;
    }
    if (j > RS_BL_MAXIT) j = RS_BL_MAXIT;
    extra += j - 5;
    maxtrip = j > maxtrip ? j : maxtrip;
    stab_lo = stab_last < stab_lo ? stab_last : stab_lo;
    stab_hi = stab_last > stab_hi ? stab_last : stab_hi;
    if (nnear + nfar > 0) {
      ++unst;
      if (nfar >= nnear) ++farc;
    }
  }
  /* Field 0 in classes (ForecastArgs::extra_log): the LONGEST loop the point is expected to run in the window.
   * The loop's slow band is a narrow curve in the stable regime (Tsurf - Tair about -2.6 VZ^2 for the default
   * heights: the fixed point's slope goes through 1 there, 20-39 passes; profiles/r05_wave_stats.txt: 0.2 % of
   * the lane-steps, but a wavefront runs as many passes as its slowest lane).  On one side of it the stability
   * parameter sits at its clamp (PSI = 4.7, two passes and it stands), on the other below: a point whose
   * previews lie on both sides passes through the band inside the window ... */
  if (a.extra_log && stab_hi >= 1.f && stab_lo < 1.f) maxtrip = 31;
  if (a.extra_log) { /* ... and the pass count of the last index stepped is one more preview (the two-wavefront
                        flavour without the history score leaves it in the score's slot; other flavours leave 0
                        or a score there: ignored unless 5..RS_BL_MAXIT) */
    const double last = st(RS_ST_BLSCORE);
    if (last >= 5.0 && last <= (double)RS_BL_MAXIT) {
      const int32_t lt = (int32_t)last;
      maxtrip = lt > maxtrip ? lt : maxtrip;
    }
  }
  if (extra > 4095) extra = 4095;
  /* field 9, the ground digit: which layers are frozen.  The heat capacity of a layer is a constant below
   * 0 C and two polynomials above (src/BalanceModel.f90:215-236), and a wavefront whose 64 points all have
   * layer j frozen takes that layer's capDZ from a table (layer_step, rs_physics_body.inc).  The frozen
   * layers of a point are nearly always one run: 4 bits for its deepest layer, 3 for the thawed layers on
   * top of it (the daily thaw reaches a handful of the thin upper layers).  Sorted as a digit of its own BELOW the others (rs_cluster_count_sort, low_bits): the
   * classes of the boundary-layer fields stay as they are, their points line up by frost depth. */
  uint32_t ground = 0u;
  {
    int32_t m9 = a.pv.mode;
    bool want = false;
    for (; m9 > 0; m9 /= 10) want |= (m9 % 10) == 9;
    if (want) {
      const int NL = c.NLayers;
      int32_t deepest = 0, first = 0;
      for (int j = NL; j >= 1; --j)
        if (st(RS_ST_TMP0 + j - 1) < 0.0) {
          if (!deepest) deepest = j;
          first = j;
        }
      if (NL > 15) deepest = (deepest * 15) / NL;
      const int32_t top = first > 0 ? first - 1 : 0;
      ground = ((uint32_t)deepest << 3) | (uint32_t)(top > 7 ? 7 : top);
    }
  }
  /* key fields, most significant first, named by the decimal digits of `mode`:
   * 1 unstable previews (4 bits), 2 of which on the table path of log (4 bits), 3 cover (1 bit),
   * 4 predicted extra passes (12 bits), 5 storage class snow > ice > wet > bare (2 bits).  Modes 0..3 are shorthands for 14, 124, 134, 1234. */
  int32_t m = a.pv.mode;
  m = (m == 0) ? 14 : (m == 1) ? 124 : (m == 2) ? 134 : (m == 3) ? 1234 : m;
  int32_t div = 1;
  while (m / div >= 10) div *= 10;
  uint32_t key = 0;
  int bits = 0;
  for (; div >= 1; div /= 10) {
    const int d = (m / div) % 10;
    if (d == 1) { key = (key << 4) | (uint32_t)unst; bits += 4; }
    else if (d == 2) { key = (key << 4) | (uint32_t)farc; bits += 4; }
    else if (d == 3) { key = (key << 1) | (uint32_t)cover; bits += 1; }
    else if (d == 4) { key = (key << 12) | (uint32_t)extra; bits += 12; }
    else if (d == 5) { key = (key << 2) | (uint32_t)sclass; bits += 2; }
    /* compact forms: extra passes saturating at 31 (5 bits), the two preview counts at 3 (2 bits) */
    else if (d == 6) { key = (key << 5) | (uint32_t)(extra > 31 ? 31 : extra); bits += 5; }
    else if (d == 0) { /* (inside a list) extra passes saturating at 7, or the longest preview in classes */
      const int32_t t = maxtrip;
      const uint32_t cls = (uint32_t)(t <= 8 ? t - 5 : t <= 12 ? 4 : t <= 20 ? 5 : t <= 30 ? 6 : 7);
      key = (key << 3) | (a.extra_log ? cls : (uint32_t)(extra > 7 ? 7 : extra));
      bits += 3;
    }
    /* (more than three previews: the counts scaled to 0..3) */
    else if (d == 7) { key = (key << 2) | (uint32_t)(a.pv.n > 3 ? (unst * 3 + a.pv.n / 2) / a.pv.n : unst > 3 ? 3 : unst); bits += 2; }
    else if (d == 8) { key = (key << 2) | (uint32_t)(a.pv.n > 3 ? (farc * 3 + a.pv.n / 2) / a.pv.n : farc > 3 ? 3 : farc); bits += 2; }
  }
  /* RsPreview::prec (ABI 8): the most significant bit - some preview has precipitation.  Points with
   * precipitation are a few per cent of a batch at any time, but in an order that ignores it they sit in four
   * wavefronts out of five, and every such wavefront runs PrecipitationToStorage / CalcPrecType (the logistic
   * exp where the phase is missing) for all of its lanes; gathered, they fill one wavefront in thirty. */
  if (a.pv.prec[0]) {
    uint32_t wet = 0u;
    for (int q = 0; q < RS_PREVIEW_MAX; ++q)
      if (a.pv.prec[q] && a.pv.prec[q][pq] > 0.0) wet = 1u;
    key |= wet << bits;
    bits += 1;
  }
  /* descending: the expensive points get the low slots (longest job first, rs_cluster.hip) */
  if (a.compact) { /* right-aligned in its own bits: the plan's counting sort */
    const int low = a.low_bits; /* the ground digit below them, if the mode has one */
    a.keys[s] = ((((1u << bits) - 1u) - key) << low) | (low ? ground : 0u);
  } else if (a.low_bits && bits + a.low_bits <= RS_SORT_KEY_BITS) { /* the same order from the library sort */
    const int low = a.low_bits;
    a.keys[s] = (((((1u << bits) - 1u) - key) << low) | ground) << (RS_SORT_KEY_BITS - bits - low);
  } else {
    if (bits < RS_SORT_KEY_BITS) key <<= (RS_SORT_KEY_BITS - bits); /* left-aligned in the sorted bits */
    else key >>= (bits - RS_SORT_KEY_BITS);
    a.keys[s] = ((1u << RS_SORT_KEY_BITS) - 1u) - key;
  }
  a.slots[s] = (uint32_t)s;
}

}  // namespace rs

/* ---- launchers (host) --------------------------------------------------- */

static inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + RS_BLOCK - 1) / RS_BLOCK)); }

hipError_t rs_read_div_mismatch(unsigned long long *out /*[3]*/, hipStream_t stream) {
  hipError_t e = hipMemcpyFromSymbolAsync(out, HIP_SYMBOL(rs::g_div_mismatch), 3 * sizeof(*out), 0,
                                          hipMemcpyDeviceToHost, stream);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(stream);
}

hipError_t rs_read_bl_stats(unsigned long long *out /*[RS_BL_NSTATS]*/, hipStream_t stream) {
  hipError_t e = hipMemcpyFromSymbolAsync(out, HIP_SYMBOL(rs::g_bl_stats), RS_BL_NSTATS * sizeof(*out), 0,
                                          hipMemcpyDeviceToHost, stream);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(stream);
}

hipError_t rs_read_div_samples(double *out /*[64][4]*/, hipStream_t stream) {
  hipError_t e = hipMemcpyFromSymbolAsync(out, HIP_SYMBOL(rs::g_div_samples), 64 * 4 * sizeof(double), 0,
                                          hipMemcpyDeviceToHost, stream);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(stream);
}

hipError_t rs_launch_math_test(int fn, int64_t n, const double *x, double *y, hipStream_t stream) {
  hipLaunchKernelGGL(rs::math_test_kernel, grid_for(n), dim3(RS_BLOCK), 0, stream, fn, n, x, y);
  return hipGetLastError();
}

hipError_t rs_launch_humidity_fill(const double *tair, double *tdew, double *rhz, int64_t n,
                                   hipStream_t stream) {
  hipLaunchKernelGGL(rs::humidity_fill_kernel, grid_for(n), dim3(RS_BLOCK), 0, stream, tair, tdew,
                     rhz, n);
  return hipGetLastError();
}

hipError_t rs_upload_math_tables(hipStream_t stream) {
  hipError_t e = hipMemcpyToSymbolAsync(HIP_SYMBOL(rs::c_gl_exp_tab), rs_gl_exp_tab,
                                        sizeof(rs_gl_exp_tab), 0, hipMemcpyHostToDevice, stream);
  if (e != hipSuccess) return e;
  return hipMemcpyToSymbolAsync(HIP_SYMBOL(rs::c_gl_log_tab), rs_gl_log_tab, sizeof(rs_gl_log_tab), 0,
                                hipMemcpyHostToDevice, stream);
}

int rs_forecast_key_low_bits(int32_t m) {
  for (; m > 0; m /= 10)
    if (m % 10 == 9) return 7;
  return 0;
}

/* the same field widths as forecast_key_kernel */
int rs_forecast_key_bits(int32_t m) {
  m = (m == 0) ? 14 : (m == 1) ? 124 : (m == 2) ? 134 : (m == 3) ? 1234 : m;
  int bits = 0;
  for (; m > 0; m /= 10) {
    const int d = m % 10;
    bits += d == 1 ? 4 : d == 2 ? 4 : d == 3 ? 1 : d == 4 ? 12 : d == 5 ? 2 : d == 6 ? 5 : d == 7 ? 2 : d == 8 ? 2 : d == 0 ? 3 : 0;
  }
  return bits;
}

hipError_t rs_launch_forecast_keys(const rs::ForecastArgs &a, hipStream_t stream) {
  hipLaunchKernelGGL(rs::forecast_key_kernel, grid_for(a.npoints), dim3(RS_BLOCK), 0, stream, a);
  return hipGetLastError();
}

hipError_t rs_launch_step_sky(const rs::StepArgs &a, int NL, bool score, hipStream_t stream) {
  const size_t lds = (size_t)NL * RS_BLOCK * sizeof(double);
  /* NLayers = 15: the hybrid profile at four waves per SIMD (measured, rs_driver_run with sky view, 262 144 points
   * in four blocks: LDS profile at three waves 6.9e9, hybrid at three 7.1e9, at four 7.4e9); other layer counts:
   * the LDS profile.
   * Two wavefronts per 64 points (no output depth, 32-bit window offsets: what rs_hip_step checked, StepArgs::
   * duo_full_ok bit 2) - no spills (128 registers) where the one-point-per-lane sky kernels spill 62-106 - for
   * launches of at most RS_DUO_MAX_POINTS points, like the other two-wavefront instances: rs_driver_run with sky
   * view, 65 536 points 3.5e9 -> 4.8e9, 200 000 points (four blocks) 7.6e9 -> 8.6e9; at 1 M points (blocks of
   * 250 000) 1.06e10 -> 1.03e10 - four of its wavefronts leave a SIMD no register for the other blocks' window
   * expansion, which then queues. */
  if (NL == 15 && a.npoints <= RS_DUO_MAX_POINTS && (a.duo_full_ok & 4) && !a.diag) {
    const dim3 gd(a.wave_start ? (unsigned)a.wave_n : (unsigned)((a.npoints + 63) / 64));
    if (score) hipLaunchKernelGGL((rs::step_kernel_duo<15, true, rs::SRC_WINDOW, true, true>), gd, dim3(128), 0, stream, a);
    else hipLaunchKernelGGL((rs::step_kernel_duo<15, false, rs::SRC_WINDOW, true, true>), gd, dim3(128), 0, stream, a);
    return hipGetLastError();
  }
  /* (a plan with diagnostics: the instance that carries bl_diagnose - here and below) */
  if (a.diag) hipLaunchKernelGGL(rs::step_kernel_sky<true>, grid_for(a.npoints), dim3(RS_BLOCK), lds, stream, a);
  else if (NL == 15) hipLaunchKernelGGL((rs::step_kernel_sky_h<4>), grid_for(a.npoints), dim3(RS_BLOCK), 0, stream, a);
  else hipLaunchKernelGGL(rs::step_kernel_sky<false>, grid_for(a.npoints), dim3(RS_BLOCK), lds, stream, a);
  return hipGetLastError();
}


hipError_t rs_launch_step_duo_knots(const rs::StepArgs &a, bool score, hipStream_t stream) {
  const dim3 gd(a.wave_start ? (unsigned)a.wave_n : (unsigned)((a.npoints + 63) / 64));
  const bool full = (a.duo_full_ok & 1) != 0;
  if (full && score) hipLaunchKernelGGL((rs::step_kernel_duo<15, true, rs::SRC_KNOTS, true>), gd, dim3(128), 0, stream, a);
  else if (full) hipLaunchKernelGGL((rs::step_kernel_duo<15, false, rs::SRC_KNOTS, true>), gd, dim3(128), 0, stream, a);
  else if (score) hipLaunchKernelGGL((rs::step_kernel_duo<15, true, rs::SRC_KNOTS>), gd, dim3(128), 0, stream, a);
  else hipLaunchKernelGGL((rs::step_kernel_duo<15, false, rs::SRC_KNOTS>), gd, dim3(128), 0, stream, a);
  return hipGetLastError();
}

hipError_t rs_launch_step_duo_raw(const rs::StepArgs &a, bool score, bool sky, bool cpl, hipStream_t stream) {
  const dim3 gd(a.wave_start ? (unsigned)a.wave_n : (unsigned)((a.npoints + 63) / 64));
  if (cpl) { /* lock-step chunk of a coupled plan (rs_step_raw with coupling): the history score is kept */
    if (sky) hipLaunchKernelGGL((rs::step_kernel_duo<15, true, rs::SRC_RAW, true, true, true>), gd, dim3(192), 0, stream, a);
    else hipLaunchKernelGGL((rs::step_kernel_duo<15, true, rs::SRC_RAW, true, false, true>), gd, dim3(128), 0, stream, a);
    return hipGetLastError();
  }
  if (sky && score) hipLaunchKernelGGL((rs::step_kernel_duo<15, true, rs::SRC_RAW, true, true>), gd, dim3(192), 0, stream, a);
  else if (sky) hipLaunchKernelGGL((rs::step_kernel_duo<15, false, rs::SRC_RAW, true, true>), gd, dim3(192), 0, stream, a);
  else if (score) hipLaunchKernelGGL((rs::step_kernel_duo<15, true, rs::SRC_RAW, true>), gd, dim3(128), 0, stream, a);
  else hipLaunchKernelGGL((rs::step_kernel_duo<15, false, rs::SRC_RAW, true>), gd, dim3(128), 0, stream, a);
  return hipGetLastError();
}

/* one replay round of the coupling windows over the listed points, forcing from the raw series */
hipError_t rs_launch_step_duo_raw_replay(const rs::StepArgs &a, hipStream_t stream) {
  if (!a.cpl_list || a.cpl_nlist < 1) return hipSuccess;
  const dim3 gd((unsigned)((a.cpl_nlist + 63) / 64));
  hipLaunchKernelGGL((rs::step_kernel_duo<15, true, rs::SRC_RAW, true, false, true, true>), gd, dim3(128), 0, stream, a);
  return hipGetLastError();
}

hipError_t rs_launch_step_cpl(const rs::StepArgs &a, int NL, hipStream_t stream) {
  /* NLayers = 15: the hybrid profile at three waves per SIMD (measured, tools/r3_cpl.sh, rs_driver_run with
   * coupling, 1 M points: 7.75e9; LDS profile at 3 waves 7.0e9, hybrid at 4 waves - spills - 6.6e9; a profile wholly
   * in registers was twice as slow); other layer counts: the LDS profile */
  const size_t lds = (size_t)NL * RS_BLOCK * sizeof(double);
  const dim3 g = grid_for(a.npoints);
  if (NL == 15 && a.pp.sky_view) hipLaunchKernelGGL((rs::step_kernel_cpl_h<3, true>), g, dim3(RS_BLOCK), 0, stream, a);
  else if (NL == 15) hipLaunchKernelGGL((rs::step_kernel_cpl_h<3, false>), g, dim3(RS_BLOCK), 0, stream, a);
  else if (a.pp.sky_view) hipLaunchKernelGGL(rs::step_kernel_cpl<true>, g, dim3(RS_BLOCK), lds, stream, a);
  else hipLaunchKernelGGL(rs::step_kernel_cpl<false>, g, dim3(RS_BLOCK), lds, stream, a);
  return hipGetLastError();
}

hipError_t rs_launch_step_cpl_replay(const rs::StepArgs &a, int NL, hipStream_t stream) {
  if (!a.cpl_list || a.cpl_nlist < 1) return hipSuccess;
  const size_t lds = (size_t)NL * RS_BLOCK * sizeof(double);
  const dim3 g = grid_for(a.cpl_nlist);
  if (NL == 15 && a.pp.sky_view) hipLaunchKernelGGL((rs::step_kernel_cpl_replay_h<3, true>), g, dim3(RS_BLOCK), 0, stream, a);
  else if (NL == 15) hipLaunchKernelGGL((rs::step_kernel_cpl_replay_h<3, false>), g, dim3(RS_BLOCK), 0, stream, a);
  else if (a.pp.sky_view) hipLaunchKernelGGL(rs::step_kernel_cpl_replay<true>, g, dim3(RS_BLOCK), lds, stream, a);
  else hipLaunchKernelGGL(rs::step_kernel_cpl_replay<false>, g, dim3(RS_BLOCK), lds, stream, a);
  return hipGetLastError();
}

hipError_t rs_launch_step_coupled(const rs::StepArgs &a, int NL, hipStream_t stream) {
  const size_t lds = (size_t)NL * RS_BLOCK * sizeof(double);
  const int64_t n = a.cpl_list ? (int64_t)a.cpl_nlist : a.npoints;
  if (n < 1) return hipSuccess;
  if (a.diag) hipLaunchKernelGGL(rs::step_kernel_coupled<true>, grid_for(n), dim3(RS_BLOCK), lds, stream, a);
  else hipLaunchKernelGGL(rs::step_kernel_coupled<false>, grid_for(n), dim3(RS_BLOCK), lds, stream, a);
  return hipGetLastError();
}

hipError_t rs_launch_step(const rs::StepArgs &a, int NL, bool full, int variant, bool score,
                          hipStream_t stream) {
  const dim3 g = grid_for(a.npoints), b(RS_BLOCK);
  const bool auto_variant = variant == RS_VARIANT_AUTO;
  /* measured (tools/r3_full2.sh, 1 M points): FULL feature set - layers 8-15 in LDS at 4 waves/SIMD
   * 1.36e10, all in registers at 3 waves/SIMD (167 VGPRs) 1.34e10, at 4 waves (14 doubles spilled)
   * 1.23e10, all in LDS 1.28e10.  LEAN: registers, 4 waves.  (tools/bench_driver_path.py relax, 1 M points, one
   * plan: the FULL feature set in the register flavour at 3 waves/SIMD 0.745 s, at 2 waves 0.80 s, at 4 waves -
   * 130 spilled VGPRs - 0.87 s; with the profile in LDS 0.86 s (3 waves) / 0.88 s (4 waves).)  The waves-per-SIMD
   * bound was a digit of the variant until round 6; the measured choices are the kernels' launch bounds now. */
  if (a.diag) variant = RS_VARIANT_LDS;
  else if (auto_variant) variant = (NL != 15) ? RS_VARIANT_LDS : full ? RS_VARIANT_HYBRID : RS_VARIANT_REG;
  /* 32-bit window offsets (WinOff) where every stream of both windows spans < 4 GiB (rs_a32_limit: the tests
   * lower it to reach the 64-bit instances with windows of megabytes) */
  const int64_t out_rows = ((int64_t)a.t0 + a.nsteps - 2) / a.o.decimate - a.o.row0 + 1;
  const bool a32 = (uint64_t)a.f.t_stride * (uint64_t)a.nsteps < rs_a32_limit() &&
                   (uint64_t)a.o.t_stride * (uint64_t)(out_rows > 0 ? out_rows : 1) < rs_a32_limit();
  /* small shards: two wavefronts per 64 points (step_kernel_duo).  Measured on MI355X (tools/r3_eval.sh): faster
   * than one point per lane below RS_DUO_MAX_POINTS points per launch */
  const bool duo_ok = (!full || a.duo_full_ok) && NL == 15 && a32;
  if (variant == RS_VARIANT_DUO && !duo_ok) { /* not this launch: as AUTO */
    variant = (NL != 15) ? RS_VARIANT_LDS : full ? RS_VARIANT_HYBRID : RS_VARIANT_REG;
  } else if (variant == RS_VARIANT_DUO || (auto_variant && !a.diag && duo_ok && a.npoints <= RS_DUO_MAX_POINTS)) {
    const dim3 gd(a.wave_start ? (unsigned)a.wave_n : (unsigned)((a.npoints + 63) / 64));
    if (full && score) hipLaunchKernelGGL((rs::step_kernel_duo<15, true, rs::SRC_WINDOW, true>), gd, dim3(128), 0, stream, a);
    else if (full) hipLaunchKernelGGL((rs::step_kernel_duo<15, false, rs::SRC_WINDOW, true>), gd, dim3(128), 0, stream, a);
    else if (score) hipLaunchKernelGGL((rs::step_kernel_duo<15, true>), gd, dim3(128), 0, stream, a);
    else hipLaunchKernelGGL((rs::step_kernel_duo<15, false>), gd, dim3(128), 0, stream, a);
    return hipGetLastError();
  }
  if (variant == RS_VARIANT_HYBRID && (NL != 15 || !full)) /* not this launch: as AUTO */
    variant = (NL == 15) ? RS_VARIANT_REG : RS_VARIANT_LDS;
  if (variant == RS_VARIANT_HYBRID) {
#define RS_HYB(S, A) \
  if (score == S && a32 == A) hipLaunchKernelGGL((rs::step_kernel_hybrid<S, A>), g, b, 0, stream, a);
    RS_HYB(false, false) RS_HYB(false, true) RS_HYB(true, false) RS_HYB(true, true)
#undef RS_HYB
    return hipGetLastError();
  }
  if (variant == RS_VARIANT_REG) {
    if (NL != 15) return hipErrorInvalidValue;
#define RS_REG(F, S, A) \
  if (full == F && score == S && a32 == A) hipLaunchKernelGGL((rs::step_kernel_reg<15, F, S, A>), g, b, 0, stream, a);
#define RS_REG_A(F, S) RS_REG(F, S, false) RS_REG(F, S, true)
    RS_REG_A(false, true) RS_REG_A(false, false) RS_REG_A(true, true) RS_REG_A(true, false)
#undef RS_REG_A
#undef RS_REG
  } else {
    const size_t lds = (size_t)NL * RS_BLOCK * sizeof(double);
    /* (diagnostics: the FULL instance whatever the launch's feature set - it reads a missing optional stream as
     * "no value" and is exact for a LEAN launch too) */
    if (a.diag) hipLaunchKernelGGL((rs::step_kernel_lds<true, true>), g, b, lds, stream, a);
    else if (full) hipLaunchKernelGGL((rs::step_kernel_lds<true>), g, b, lds, stream, a);
    else hipLaunchKernelGGL((rs::step_kernel_lds<false>), g, b, lds, stream, a);
  }
  return hipGetLastError();
}

hipError_t rs_launch_init(const rs::InitArgs &a, hipStream_t stream) {
  hipLaunchKernelGGL(rs::init_kernel, grid_for(a.npoints), dim3(RS_BLOCK), 0, stream, a);
  return hipGetLastError();
}

hipError_t rs_launch_knots(const rs::KnotArgs &a, int32_t nknots, hipStream_t stream) {
  rs::KnotArgs b = a;
  b.nknots = nknots;
  hipLaunchKernelGGL(rs::synth_knots_kernel, grid_for(a.npoints), dim3(RS_BLOCK), 0, stream, b);
  return hipGetLastError();
}

hipError_t rs_launch_expand(const rs::ExpandArgs &a, int32_t nintervals, hipStream_t stream) {
  dim3 g = grid_for(a.npoints);
  g.y = (unsigned)nintervals;
  hipLaunchKernelGGL(rs::expand_kernel, g, dim3(RS_BLOCK), 0, stream, a);
  return hipGetLastError();
}

hipError_t rs_launch_clock_probe(uint64_t *out, uint32_t spin_us, hipStream_t stream) {
  if (spin_us > RS_CLOCK_PROBE_MAX_US) spin_us = RS_CLOCK_PROBE_MAX_US; /* a probe, not a way to park a wave */
  hipLaunchKernelGGL(rs::clock_probe_kernel, dim3(1), dim3(64), 0, stream, out, spin_us);
  return hipGetLastError();
}

hipError_t rs_launch_cpl_window_bounds(const rs::StepArgs &a, int32_t *out, hipStream_t stream) {
  hipLaunchKernelGGL(rs::cpl_window_bounds_kernel, grid_for(a.npoints), dim3(RS_BLOCK), 0, stream, a, out);
  return hipGetLastError();
}

hipError_t rs_launch_count_failed(const double *st, int64_t np_pad, int64_t npoints,
                                  unsigned long long *out, hipStream_t stream) {
  hipLaunchKernelGGL(rs::count_failed_kernel, grid_for(npoints), dim3(RS_BLOCK), 0, stream, st,
                     np_pad, npoints, out);
  return hipGetLastError();
}
