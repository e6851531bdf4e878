/*
 * rs_raw.hpp - the raw-series rule of the reference driver, shared by the driver kernels (rs_driver.hip:
 * scan, previews, the window expansion of the one-point-per-lane flavours) and by the ground wave of the
 * two-wavefront step kernel (rs_kernels.hip, duo_ground<..., SRC_RAW>), which makes its forcing from the
 * raw series itself:
 *   JsonSource::interpolate   examples/example1/src/JsonSource.cpp:49-176
 *   GetWeather's own test     JsonSource.cpp:323-373 (a source hands a value on only where it is > -100,
 *                             LW_net > -1000)
 *   DataHandler::GetWeather   DataHandler.cpp:75-84: later sources overwrite earlier ones
 * Sources with a time axis shared by all points: the walk over the two time axes (which raw interval a
 * simulation index falls into, copy or interpolate) is run once per source on the host (rs_driver.hip
 * build_plan) and handed over as one RawPlanStep per simulation index; a RawSeg is a run of simulation
 * indices over which every source keeps its (kind, rawPos).
 */
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/roadsurf.h"

namespace rs {

constexpr int RAW_NFLD = 10;
/* order of the raw fields (and of `merged` in rs_driver_expand) */
enum { RAW_TAIR, RAW_TDEW, RAW_VZ, RAW_RHZ, RAW_PREC, RAW_SW, RAW_LW, RAW_SWDIR, RAW_LWNET, RAW_OBS };
enum { RAW_NONE = 0, RAW_COPY = 1, RAW_INTERP = 2 };

struct RawPlanStep {
  int32_t kind; /* RAW_* */
  int32_t rp;   /* rawPos */
  double num;   /* simtime[simPos] - rawtime[rawPos]      (JsonSource.cpp:116 ff.) */
  double den;   /* rawtime[rawPos+1] - rawtime[rawPos] */
  double rden;  /* RN(1 / den) where the quotient may be formed with it (rs_math.hpp rs_div_u: a uniform
                   denominator that is a whole number of seconds in [1, 2^40), 0 < num < den); 0: IEEE
                   division - and no promise that the value lies between the two raw ends */
};

struct RawSeg {
  int32_t i0, i1; /* simulation indices [i0, i1), 0-based */
  int32_t kind[RS_MAX_SOURCES], rp[RS_MAX_SOURCES];
};

/* What the step kernel's ground wave needs to make the forcing of a launch from the raw series (device
 * pointers; series [n_times][np_pad], columns in POINT order). */
struct RawSrc {
  const double *fld[RAW_NFLD]; /* nullptr: the source does not have the variable */
  const RawPlanStep *plan;     /* [SimLen] */
};
struct RawForcing {
  RawSrc src[RS_MAX_SOURCES];
  int32_t nsrc;
  int32_t nseg, seg0;     /* the segment table and the segment the launch's first index lies in */
  const RawSeg *segs;
  int64_t np_pad;         /* elements between two raw times of a series */
  const int32_t *col;     /* slot -> column of the raw series (the plan's order row); nullptr: column = slot */
  const int32_t *status;  /* [column] read_input's verdict: != 0, the point is not simulated (its air
                             temperature reads missing: CheckValues stops it at the first index); or nullptr */
  const int32_t *hour;    /* [SimLen] local hour of the simulation times (JsonSource.cpp:297-308) */
};

/* examples/example1/src/InputData.cpp:5-26: every series starts out missing */
__device__ __forceinline__ double raw_miss() { return -9999.9; }
/* JsonSource.cpp:92-111,323-345: `> -100.0`, except LW_net `> -1000.0` */
__device__ __forceinline__ double raw_threshold(int fld) { return fld == RAW_LWNET ? -1000.0 : -100.0; }

/* A shared-axis plan entry, read as constant memory (address space 4): one scalar load of the
 * 32-byte entry.  Through a generic pointer the compiler loads the two doubles with a VECTOR load
 * from the uniform address (it cannot prove the array apart from the window it is writing) and
 * the lane waits a vector-memory round trip per source and time index.  The plans are uploaded
 * before any kernel of the run and never written on the device. */
__device__ __forceinline__ RawPlanStep raw_plan_at(const RawPlanStep *plan, int32_t i) {
  const RawPlanStep __attribute__((address_space(4))) *q =
      (const RawPlanStep __attribute__((address_space(4))) *)plan + i;
  RawPlanStep st;
  st.kind = q->kind;
  st.rp = q->rp;
  st.num = q->num;
  st.den = q->den;
  st.rden = q->rden;
  return st;
}

/* x / den with den uniform and rden = RN(1 / den): two fused multiply-adds (Markstein; rs_math.hpp
 * rs_div_u: the IEEE quotient for a numerator of moderate exponent - a zero numerator gives +-0 like the
 * division) */
__device__ __forceinline__ double raw_quot(double x, double den, double rden) {
  const double q0 = x * rden;
  const double rem = __builtin_fma(-den, q0, x);
  return __builtin_fma(rem, rden, q0);
}

/* Value of one variable of one source at one simulation index: JsonSource::interpolate
 * (JsonSource.cpp:86-170) followed by GetWeather's own test (JsonSource.cpp:337-356).
 * a, b = raw[rawPos], raw[rawPos+1]. */
__device__ __forceinline__ bool raw_source_value(const RawPlanStep &st, double a, double b, double thr,
                                                 double &v) {
  if (st.kind == RAW_COPY) {
    v = a;
    return a > thr;
  }
  if (!(a > thr && b > thr)) return false;
  /* raw[rawPos] + (simtime-rawtime[rawPos]) * (raw[rawPos+1]-raw[rawPos]) / (rawtime[rawPos+1]-rawtime[rawPos]) */
  const double x = st.num * (b - a);
  double q;
  /* shared axis: the denominator is uniform and comes with its correctly rounded reciprocal, so
   * the quotient is two fused multiply-adds (b - a cannot be -0.0 unless b is, and then a + q is the
   * same for either zero).  Anything else: IEEE division. */
  const double ax = __builtin_fabs(x);
  if (st.rden != 0.0 && !(ax >= 1e290) && !(ax > 0.0 && ax < 1e-290)) {
    q = raw_quot(x, st.den, st.rden);
  } else {
    q = x / st.den;
  }
  v = a + q;
  return v > thr;
}

}  // namespace rs
