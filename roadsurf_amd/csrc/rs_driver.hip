/*
 * rs_driver.hip — layer 4 of include/roadsurf.h: the reference DRIVER's data path
 * (examples/example1/src: JsonSource.cpp, DataHandler.cpp, MeteorologyTools.cpp,
 * roadrunner.cpp read_input/save_output) with the per-value work on the device.
 *
 * Data flow per tile of points:
 *   raw host series [point][time] --H2D--> LDS-tiled transpose --> raw[time][point] in HBM
 *   humidity_fill_kernel       Tdew <-> RH completion           (JsonSource.cpp:288-295)
 *   scan_raw_kernel            per variable: first missing index, latest observation,
 *                              latest road-temperature observation (one pass over the series)
 *   finalize_kernel            read_input's decisions per point    (roadrunner.cpp:186-275)
 *   per time chunk:
 *     expand_raw_kernel        JsonSource::interpolate + GetWeather overlay, written as the
 *                              [t][point] windows the step kernels read
 *     step kernel              (layers 1-3), outputs decimated in the kernel
 *   outputs [row][point] --transpose--> [point][row] --D2H--> caller
 *
 * JsonSource::interpolate walks the raw and the simulation time axes together; which raw
 * interval a simulation index falls into, and whether it copies or interpolates there,
 * depends on the TIMES only.  All points of a source share its time axis, so the walk is
 * run once per source on the host (build_plan: the reference's loop, statement for
 * statement, quirks included) and the device applies the resulting per-index plan to every
 * point's data with the reference's arithmetic and missing-value tests.
 *
 * The kernels are streaming (HBM-bound): one variable per blockIdx.y, one point per lane,
 * time in the loop, the two raw neighbours cached in registers and re-read only when the
 * plan moves to another raw interval.
 */
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <vector>

#include "../../include/roadsurf.h"
#include "rs_devutil.hpp"
#include "rs_state.h"
#include "rs_devices.hpp"
#include "rs_kernels.h"
#include "rs_raw.hpp"

extern "C" void rs_host_set_error(const char *msg);

namespace {

using rsu::Dev;
using rsu::transpose;

#ifndef RS_DRIVER_RAW_CHUNK
#define RS_DRIVER_RAW_CHUNK 120 /* indices per launch of a block whose step kernel reads the raw series (round 5, with the class key: 120 beats 240 by 1 % without coupling, 3-4 % with; 60 and 180 between) */
#endif
constexpr int NFLD = rs::RAW_NFLD;
/* order of `merged` in rs_driver_expand and of the window buffers (rs_raw.hpp) */
enum { R_TAIR = rs::RAW_TAIR, R_TDEW = rs::RAW_TDEW, R_VZ = rs::RAW_VZ, R_RHZ = rs::RAW_RHZ, R_PREC = rs::RAW_PREC,
       R_SW = rs::RAW_SW, R_LW = rs::RAW_LW, R_SWDIR = rs::RAW_SWDIR, R_LWNET = rs::RAW_LWNET, R_OBS = rs::RAW_OBS };
enum { K_NONE = rs::RAW_NONE, K_COPY = rs::RAW_COPY, K_INTERP = rs::RAW_INTERP };
using PlanStep = rs::RawPlanStep; /* one per source and simulation index (rs_raw.hpp) */
__device__ __forceinline__ PlanStep plan_at(const PlanStep *plan, int32_t i) { return rs::raw_plan_at(plan, i); }

struct SrcDev {
  const double *fld[NFLD]; /* [n_times][np_pad], nullptr = variable absent */
  const PlanStep *plan;    /* [SimLen]; shared time axis only */
  /* per-point time axes (RsRawSource.times_per_point): the walk runs per lane */
  const int64_t *ptimes;   /* [n_times][np_pad] or nullptr */
  const int32_t *plen;     /* [np_pad] series lengths */
  int32_t *prp;            /* [np_pad] rawPos of the walk at the start of the current window */
  int32_t n_times;
  int32_t is_obs;
};

struct SrcSet {
  SrcDev src[RS_MAX_SOURCES];
  int32_t nsrc;
  int32_t simlen;
  int64_t np_pad;
  int64_t npoints;
  int64_t sim0; /* simulation index i is at sim0 + i*dt seconds (JsonSource.cpp:199-205) */
  int32_t dt;
};

__device__ __forceinline__ double miss_r() { return rs::raw_miss(); }
__device__ __forceinline__ double threshold(int fld) { return rs::raw_threshold(fld); }
__device__ __forceinline__ bool source_value(const PlanStep &st, double a, double b, double thr, double &v) {
  return rs::raw_source_value(st, a, b, thr, v);
}

/* ---- per-point time axes: JsonSource::interpolate's walk (JsonSource.cpp:57-85,113-114,171)
 * evaluated per lane instead of once per source on the host.  State of the walk = rawPos. */
constexpr int32_t PP_DEAD = 0x3fffffff; /* the while loop can never run (again) for this series */

/* rawPos before the first simulation index (JsonSource.cpp:60-82) */
__device__ __forceinline__ int32_t pp_initial(const SrcDev &sd, int64_t np_pad, int64_t p,
                                              int64_t sim0) {
  const int32_t len = sd.plen[p];
  if (len <= 0) return PP_DEAD; /* JsonSource.cpp:233-237 */
  if (sd.ptimes[p] < sim0) {
    int32_t k = 0;
    for (; k < len; ++k)
      if (sd.ptimes[(int64_t)k * np_pad + p] >= sim0) break;
    return k - 1;
  }
  return 0;
}

struct PpWalk {
  int32_t rp, len, cached; /* cached: the rawPos tr/tr1 belong to, -1 none */
  int64_t tr, tr1;
};

__device__ __forceinline__ void pp_begin(PpWalk &w, const SrcDev &sd, int64_t p, int32_t rp0) {
  w.rp = rp0;
  w.len = sd.plen[p];
  w.cached = -1;
  w.tr = w.tr1 = 0;
}

/* One pass of the reference's while loop for simulation time ts: what happens to this index. */
__device__ __forceinline__ PlanStep pp_step(PpWalk &w, const SrcDev &sd, int64_t np_pad, int64_t p,
                                            int64_t ts) {
  PlanStep st;
  st.kind = K_NONE;
  st.rp = 0;
  st.num = 0.0;
  st.den = 1.0;
  st.rden = 0.0; /* per-lane denominators: IEEE division */
  if (w.rp + 1 >= w.len) return st; /* `while (rawPos+1 < rawLen ...)` is over */
  if (w.cached != w.rp) {
    w.tr = sd.ptimes[(int64_t)w.rp * np_pad + p];
    w.tr1 = sd.ptimes[(int64_t)(w.rp + 1) * np_pad + p];
    w.cached = w.rp;
  }
  if (ts < w.tr) return st; /* simulation starts before the data: the walk has not begun */
  if (ts == w.tr) {
    st.kind = K_COPY;
    st.rp = w.rp;
  } else if (ts == w.tr1) {
    /* rawPos++ and the loop condition is tested again for the same simulation index */
    w.rp += 1;
    if (w.rp + 1 >= w.len) {
      w.rp = PP_DEAD;
      return st;
    }
    w.tr = w.tr1;
    w.tr1 = sd.ptimes[(int64_t)(w.rp + 1) * np_pad + p];
    w.cached = w.rp;
    st.kind = K_COPY;
    st.rp = w.rp;
  } else {
    st.kind = K_INTERP;
    st.rp = w.rp;
    st.num = (double)(ts - w.tr);
    st.den = (double)(w.tr1 - w.tr);
  }
  return st;
}

/* Sequential pass over simulation indices [i0, i1) of one variable of one point.
 * visit(i, merged value, bitmask of the sources that supplied a value).
 * rp0[s]: rawPos at i0 for sources with per-point time axes. */
template <bool PP, class Visit> /* PP: some source has per-point time axes */
__device__ __forceinline__ void walk_field(const SrcSet &S, int fld, int64_t p, int32_t i0,
                                           int32_t i1, const int32_t *rp0, Visit &&visit) {
  const double thr = threshold(fld);
  double a[RS_MAX_SOURCES], b[RS_MAX_SOURCES];
  int32_t cur[RS_MAX_SOURCES];
  PpWalk pw[RS_MAX_SOURCES];
#pragma unroll
  for (int s = 0; s < RS_MAX_SOURCES; ++s) {
    a[s] = b[s] = 0.0;
    cur[s] = -2;
    pw[s].rp = PP_DEAD;
    pw[s].len = 0;
    pw[s].cached = -1;
    pw[s].tr = pw[s].tr1 = 0;
    if (PP && s < S.nsrc && S.src[s].ptimes) pp_begin(pw[s], S.src[s], p, rp0[s]);
  }
  /* the variable's column and the plan of every source, once (fld indexes the kernel arguments
   * dynamically: left inside, that is a scalar load per source and time index) */
  const double *xs[RS_MAX_SOURCES];
  const PlanStep *plans[RS_MAX_SOURCES];
#pragma unroll
  for (int s = 0; s < RS_MAX_SOURCES; ++s) {
    xs[s] = (s < S.nsrc) ? S.src[s].fld[fld] : nullptr;
    plans[s] = (s < S.nsrc) ? S.src[s].plan : nullptr;
  }
  const int64_t np_pad = S.np_pad;
  for (int32_t i = i0; i < i1; ++i) {
    double v = miss_r();
    uint32_t mask = 0;
#pragma unroll
    for (int s = 0; s < RS_MAX_SOURCES; ++s) {
      if (s >= S.nsrc) continue;
      const double *x = xs[s];
      PlanStep st;
      if (PP && S.src[s].ptimes) {
        /* the walk advances whether or not this source has the variable */
        st = pp_step(pw[s], S.src[s], np_pad, p, S.sim0 + (int64_t)i * S.dt);
      } else {
        if (!x) continue;
        st = plan_at(plans[s], i); /* uniform: one scalar load */
      }
      if (!x || st.kind == K_NONE) continue;
      if (st.rp != cur[s]) {
        a[s] = x[(int64_t)st.rp * np_pad + p];
        b[s] = x[(int64_t)(st.rp + 1) * np_pad + p];
        cur[s] = st.rp;
      }
      double vs;
      if (source_value(st, a[s], b[s], thr, vs)) { /* DataHandler.cpp:75-84: later sources win */
        v = vs;
        mask |= 1u << s;
      }
    }
    visit(i, v, mask);
  }
}

/* rawPos at simulation index 0 of every per-point source */
__device__ __forceinline__ void initial_positions(const SrcSet &S, int64_t p, int32_t *rp0) {
#pragma unroll
  for (int s = 0; s < RS_MAX_SOURCES; ++s)
    rp0[s] = (s < S.nsrc && S.src[s].ptimes) ? pp_initial(S.src[s], S.np_pad, p, S.sim0) : 0;
}

/* Random access to the merged value (used for the relaxation targets). */
__device__ __forceinline__ double merged_at(const SrcSet &S, int fld, int64_t p, int32_t i) {
  const double thr = threshold(fld);
  double v = miss_r();
  for (int s = 0; s < S.nsrc; ++s) {
    const double *x = S.src[s].fld[fld];
    if (!x) continue;
    PlanStep st;
    if (S.src[s].ptimes) { /* replay the walk up to i */
      PpWalk w;
      pp_begin(w, S.src[s], p, pp_initial(S.src[s], S.np_pad, p, S.sim0));
      st.kind = K_NONE;
      for (int32_t k = 0; k <= i; ++k) st = pp_step(w, S.src[s], S.np_pad, p, S.sim0 + (int64_t)k * S.dt);
    } else {
      st = plan_at(S.src[s].plan, i);
    }
    if (st.kind == K_NONE) continue;
    const double a = x[(int64_t)st.rp * S.np_pad + p], b = x[(int64_t)(st.rp + 1) * S.np_pad + p];
    double vs;
    if (source_value(st, a, b, thr, vs)) v = vs;
  }
  return v;
}

/* roadrunner.cpp:42-45 */
__device__ __forceinline__ bool is_missing(double v) { return (v != v) || v < -9000; }

struct ScanArgs {
  SrcSet S;
  int32_t *first_missing; /* [6][np_pad]: tair, Rhz, prec, SW, LW, VZ (roadrunner.cpp:188-229) */
  int32_t *last_obs;      /* [np_pad] DataHandler::GetLatestObsIndex, -1 if none */
  int32_t *cpl_i;         /* [np_pad] last index with a road temperature observation, -1 */
  double *cpl_t;          /* [np_pad] that observation */
};

template <bool PP>
__global__ void __launch_bounds__(RS_BLOCK) scan_raw_kernel(const ScanArgs A) {
  __builtin_amdgcn_s_setprio(3); /* before a block's first time step: all of it is waited for */
  const int64_t p = (int64_t)blockIdx.x * RS_BLOCK + threadIdx.x;
  if (p >= A.S.npoints) return;
  const int y = blockIdx.y;
  const int L = A.S.simlen;
  int32_t rp0[RS_MAX_SOURCES] = {0, 0, 0, 0};
  if (PP) initial_positions(A.S, p, rp0);
  if (y < 6) {
    const int fld = (y == 0) ? R_TAIR : (y == 1) ? R_RHZ : (y == 2) ? R_PREC : (y == 3) ? R_SW
                  : (y == 4) ? R_LW : R_VZ;
    uint32_t obsmask = 0;
    for (int s = 0; s < A.S.nsrc; ++s)
      if (A.S.src[s].is_obs) obsmask |= 1u << s;
    int32_t first = L, last = -1;
    walk_field<PP>(A.S, fld, p, 0, L, rp0, [&](int32_t i, double v, uint32_t mask) {
      if (first == L && is_missing(v)) first = i;
      /* JsonSource.cpp:412-416: `for i = SimLen..1: if tair[i-1] > -100 return i`, on the
       * source's OWN interpolated series; DataHandler.cpp:118-137 takes the max over the
       * observation sources */
      if (mask & obsmask) last = i + 1;
    });
    A.first_missing[(int64_t)y * A.S.np_pad + p] = first;
    if (y == 0) A.last_obs[p] = last;
  } else {
    int32_t ci = -1;
    double ct = miss_r();
    /* roadrunner.cpp:256-261: last index whose TSurfObs is neither missing nor < -100 */
    walk_field<PP>(A.S, R_OBS, p, 0, L, rp0, [&](int32_t i, double v, uint32_t) {
      if (!(is_missing(v) || v < -100)) {
        ci = i;
        ct = v;
      }
    });
    A.cpl_i[p] = ci;
    A.cpl_t[p] = ct;
  }
}

/* Shared time axes: a SEGMENT is a run of simulation indices over which every source keeps its
 * (kind, rawPos).  Whether a source supplies a value is then the same at every index of the run:
 * K_COPY: raw[rawPos] > threshold; K_INTERP: both ends > threshold - the interpolated value lies
 * between them (the weight num/den is at most 1 - 1/den, so the rounded increment is smaller in
 * magnitude than the rounded difference of the ends) and passes GetWeather's own test with them.
 * That holds for finite ends; a run with a non-finite end above the threshold is walked index by
 * index like the general scan.  ~250 segments instead of SimLen = 5 761 indices per point and
 * variable: the scan was 18 ms of a block's 35 ms before its first time step. */
using ScanSeg = rs::RawSeg;

__device__ __forceinline__ bool rs_finite(double x) { return __builtin_fabs(x) < __builtin_inf(); }

/* merged value at one index, shared axes (walk_field's body for a single i) */
__device__ __forceinline__ double merged_one(const SrcSet &S, int fld, int64_t p, int32_t i, uint32_t &mask) {
  const double thr = threshold(fld);
  double v = miss_r();
  mask = 0;
  for (int s = 0; s < S.nsrc; ++s) {
    const double *x = S.src[s].fld[fld];
    if (!x) continue;
    const PlanStep st = plan_at(S.src[s].plan, i);
    if (st.kind == K_NONE) continue;
    const double a = x[(int64_t)st.rp * S.np_pad + p], b = x[(int64_t)(st.rp + 1) * S.np_pad + p];
    double vs;
    if (source_value(st, a, b, thr, vs)) {
      v = vs;
      mask |= 1u << s;
    }
  }
  return v;
}

__global__ void __launch_bounds__(RS_BLOCK) scan_seg_kernel(const ScanArgs A, const ScanSeg *segs, int32_t nseg) {
  __builtin_amdgcn_s_setprio(3); /* before a block's first time step: all of it is waited for */
  const int64_t p = (int64_t)blockIdx.x * RS_BLOCK + threadIdx.x;
  if (p >= A.S.npoints) return;
  const int y = blockIdx.y;
  const int L = A.S.simlen;
  const int fld = (y == 0) ? R_TAIR : (y == 1) ? R_RHZ : (y == 2) ? R_PREC : (y == 3) ? R_SW
                : (y == 4) ? R_LW : (y == 5) ? R_VZ : R_OBS;
  const double thr = threshold(fld);
  uint32_t obsmask = 0;
  for (int s = 0; s < A.S.nsrc; ++s)
    if (A.S.src[s].is_obs) obsmask |= 1u << s;
  const double *xs[RS_MAX_SOURCES];
#pragma unroll
  for (int s = 0; s < RS_MAX_SOURCES; ++s) xs[s] = (s < A.S.nsrc) ? A.S.src[s].fld[fld] : nullptr;
  const int64_t np_pad = A.S.np_pad;
  int32_t first = L, last = -1, ci = -1;
  const ScanSeg __attribute__((address_space(4))) *sg = (const ScanSeg __attribute__((address_space(4))) *)segs;
  for (int32_t k = 0; k < nseg; ++k) {
    const int32_t i0 = sg[k].i0, i1 = sg[k].i1;
    uint32_t mask = 0;
    bool slow = false;
#pragma unroll
    for (int s = 0; s < RS_MAX_SOURCES; ++s) {
      if (s >= A.S.nsrc || !xs[s]) continue;
      const int32_t kind = sg[k].kind[s], rp = sg[k].rp[s];
      if (kind == K_NONE) continue;
      const double a = xs[s][(int64_t)rp * np_pad + p];
      if (kind == K_COPY) {
        if (a > thr) mask |= 1u << s;
      } else {
        const double b = xs[s][(int64_t)(rp + 1) * np_pad + p];
        if (a > thr && b > thr) {
          mask |= 1u << s;
          if (!rs_finite(a) || !rs_finite(b)) slow = true;
        }
      }
    }
    if (slow) { /* an infinite end: index by index, as the general scan does */
      for (int32_t i = i0; i < i1; ++i) {
        uint32_t mi;
        const double v = merged_one(A.S, fld, p, i, mi);
        if (y < 6) {
          if (first == L && is_missing(v)) first = i;
          if (mi & obsmask) last = i + 1;
        } else if (!(is_missing(v) || v < -100)) {
          ci = i;
        }
      }
      continue;
    }
    if (y < 6) {
      if (first == L && mask == 0) first = i0; /* nobody supplies a value: the merged series is missing */
      if (mask & obsmask) last = i1;           /* (the run's last index) + 1 */
    } else if (mask != 0) {
      ci = i1 - 1; /* a supplied value is > -100: roadrunner.cpp:256-261 takes it */
    }
  }
  if (y < 6) {
    A.first_missing[(int64_t)y * np_pad + p] = first;
    if (y == 0) A.last_obs[p] = last;
  } else {
    double ct = miss_r();
    if (ci >= 0) {
      uint32_t mi;
      ct = merged_one(A.S, R_OBS, p, ci, mi);
    }
    A.cpl_i[p] = ci;
    A.cpl_t[p] = ct;
  }
}

struct FinalArgs {
  SrcSet S;
  const int32_t *first_missing, *last_obs, *cpl_i;
  const double *cpl_t;
  int32_t use_relaxation, use_coupling, cplLen, default_initlen;
  /* out, [np_pad] */
  int32_t *status, *missing_index, *initlen, *cpl_index, *cpl_hi;
  double *tair_relax, *vz_relax, *rh_relax, *cpl_tsurf;
};

/* read_input after GetWeather, roadrunner.cpp:186-275 */
__global__ void __launch_bounds__(RS_BLOCK) finalize_kernel(const FinalArgs A) {
  __builtin_amdgcn_s_setprio(3); /* before a block's first time step: all of it is waited for */
  const int64_t p = (int64_t)blockIdx.x * RS_BLOCK + threadIdx.x;
  if (p >= A.S.np_pad) return;
  const int L = A.S.simlen;
  int32_t status = 0, mi = L;
  if (p < A.S.npoints) {
    for (int k = 0; k < 6; ++k) {
      const int32_t fm = A.first_missing[(int64_t)k * A.S.np_pad + p];
      if (fm < mi) { /* strict: at equal index the earlier test in the reference's order wins */
        mi = fm;
        status = k + 1;
      }
    }
  }
  int32_t initlen = A.default_initlen; /* roadrunner.cpp:168-169 */
  double tr = miss_r(), vr = miss_r(), rr = miss_r();
  int32_t cidx = -9999, chi = -1;
  double ctsurf = miss_r();
  if (p < A.S.npoints && status == 0) {
    if (A.use_relaxation == 1) { /* roadrunner.cpp:235-250 */
      const int32_t idx = A.last_obs[p];
      if (idx > -1) {
        initlen = idx;
        if (idx >= L) {
          status = 7; /* the reference reads data.tair[SimLen] here */
        } else {
          tr = merged_at(A.S, R_TAIR, p, idx);
          vr = merged_at(A.S, R_VZ, p, idx);
          rr = merged_at(A.S, R_RHZ, p, idx);
        }
      }
    }
    if (A.use_coupling == 1 && status == 0) { /* roadrunner.cpp:253-275 */
      const int32_t i = A.cpl_i[p];
      if (i >= A.cplLen) {
        ctsurf = A.cpl_t[p];
        cidx = i;
        chi = i;
      }
    }
  }
  A.status[p] = status;
  A.missing_index[p] = (status >= 1 && status <= 6) ? mi : -1;
  A.initlen[p] = initlen;
  A.tair_relax[p] = tr;
  A.vz_relax[p] = vr;
  A.rh_relax[p] = rr;
  A.cpl_index[p] = cidx;
  A.cpl_tsurf[p] = ctsurf;
  A.cpl_hi[p] = chi;
}

struct ExpandRawArgs {
  SrcSet S;
  double *out[NFLD]; /* [nsteps][stride] windows; nullptr = not wanted */
  const int32_t *status; /* [np_pad] or nullptr */
  const int32_t *cpl_hi; /* [np_pad] or nullptr */
  int32_t cplLen;
  int32_t i0, nsteps; /* 0-based first simulation index of the window */
  int64_t stride;
  const int32_t *order; /* plan order (rs_hip_recluster): window column s holds point order[s];
                           nullptr = natural order */
};

template <bool PP>
__global__ void __launch_bounds__(RS_BLOCK) expand_raw_kernel(const ExpandRawArgs A) {
  __builtin_amdgcn_s_setprio(3); /* a link of the chain between two step launches of a block: all of it is waited for */
  const int64_t slot = (int64_t)blockIdx.x * RS_BLOCK + threadIdx.x;
  const int fld = blockIdx.y;
  double *out = A.out[fld];
  if (slot >= A.S.npoints || !out) return;
  out += slot;
  /* everything per point (raw columns, walk positions, decisions) is read at the point's own
   * index: the raw series are ~100x smaller than the window, gathering them is cheap */
  const int64_t p = A.order ? (int64_t)A.order[slot] : slot;
  /* a point read_input rejects is not simulated by the reference (roadrunner.cpp:393): a
   * missing air temperature makes CheckValues stop its lane at the first index (its output
   * rows are blanked afterwards, blank_rejected_kernel) */
  const bool rejected = A.status && A.status[p] != 0 && fld == R_TAIR;
  /* roadrunner.cpp:266-273: no road temperature input inside the coupling window */
  int32_t clr_lo = 0, clr_hi = -1;
  if (fld == R_OBS && A.cpl_hi) {
    clr_hi = A.cpl_hi[p];
    clr_lo = clr_hi - A.cplLen; /* exclusive */
    if (clr_hi < 0) clr_lo = clr_hi;
  }
  const int32_t i0 = A.i0;
  const int64_t stride = A.stride;
  int32_t rp0[RS_MAX_SOURCES];
#pragma unroll
  for (int s = 0; s < RS_MAX_SOURCES; ++s)
    rp0[s] = (PP && s < A.S.nsrc && A.S.src[s].ptimes) ? A.S.src[s].prp[p] : 0;
  walk_field<PP>(A.S, fld, p, i0, i0 + A.nsteps, rp0, [&](int32_t i, double v, uint32_t) {
    if (rejected) v = miss_r();
    if (i > clr_lo && i <= clr_hi) v = miss_r();
    out[(int64_t)(i - i0) * stride] = v;
  });
}

/* A point read_input rejects is never handed to runsimulation (roadrunner.cpp:393): all its
 * output rows read -9999.0 (OutputData.cpp:5-13).  Its lanes did run - stopped by CheckValues
 * at the first index, which still writes that index's row (src/InputOutput.f90:55-82 sets the
 * flag, examples/example1/src/Simulation.f90:100 leaves the loop after SaveOutput). */
__global__ void __launch_bounds__(RS_BLOCK) blank_rejected_kernel(double *out, int64_t stride,
                                                                  int32_t nrows, int64_t npoints,
                                                                  const int32_t *status) {
  const int64_t p = (int64_t)blockIdx.x * RS_BLOCK + threadIdx.x;
  if (p >= npoints || status[p] == 0) return;
  for (int f = 0; f < 6; ++f)
    for (int32_t r = 0; r < nrows; ++r) out[((int64_t)f * nrows + r) * stride + p] = -9999.0;
}

/* A few (variable, index) rows of the merged series in slot order - what the driver path still needs as
 * ROWS once the step kernel makes its own forcing from the raw series (rs_step_raw): air temperature and
 * road-temperature observation of index 1 for the initial profile (src/Initialization.f90:256-259), air
 * temperature and wind speed at the three preview indices of the forecast sort key.  Shared time axes. */
constexpr int RAWROWS_MAX = 12;
struct RawRowsArgs {
  SrcSet S;
  const int32_t *status, *order; /* as ExpandRawArgs */
  int32_t nrows;
  int32_t fld[RAWROWS_MAX], idx[RAWROWS_MAX]; /* variable and 0-based simulation index of row y */
  double *out[RAWROWS_MAX];                   /* [np_pad] each */
};
__global__ void __launch_bounds__(RS_BLOCK) raw_rows_kernel(const RawRowsArgs A) {
  __builtin_amdgcn_s_setprio(3); /* a link of the chain between two step launches of a block: all of it is waited for */
  const int64_t slot = (int64_t)blockIdx.x * RS_BLOCK + threadIdx.x;
  const int y = blockIdx.y;
  if (slot >= A.S.npoints || y >= A.nrows) return;
  const int64_t p = A.order ? (int64_t)A.order[slot] : slot;
  const int fld = A.fld[y];
  uint32_t mask;
  double v = merged_one(A.S, fld, p, A.idx[y], mask);
  if (A.status && A.status[p] != 0 && fld == R_TAIR) v = miss_r(); /* a rejected point: as expand_raw_kernel */
  A.out[y][slot] = v;
}

/* Per-point walks: position at simulation index 0 ... */
__global__ void __launch_bounds__(RS_BLOCK) pp_init_kernel(const SrcSet S) {
  const int64_t p = (int64_t)blockIdx.x * RS_BLOCK + threadIdx.x;
  if (p >= S.npoints) return;
  for (int s = 0; s < S.nsrc; ++s)
    if (S.src[s].ptimes) S.src[s].prp[p] = pp_initial(S.src[s], S.np_pad, p, S.sim0);
}
/* ... and moved past the window [i0, i0+nsteps) that has just been expanded. */
__global__ void __launch_bounds__(RS_BLOCK) pp_advance_kernel(const SrcSet S, int32_t i0,
                                                              int32_t nsteps) {
  const int64_t p = (int64_t)blockIdx.x * RS_BLOCK + threadIdx.x;
  if (p >= S.npoints) return;
  for (int s = 0; s < S.nsrc; ++s) {
    if (!S.src[s].ptimes) continue;
    PpWalk w;
    pp_begin(w, S.src[s], p, S.src[s].prp[p]);
    for (int32_t i = i0; i < i0 + nsteps; ++i)
      (void)pp_step(w, S.src[s], S.np_pad, p, S.sim0 + (int64_t)i * S.dt);
    S.src[s].prp[p] = w.rp;
  }
}

/* per-point parameters into slot order */
__global__ void __launch_bounds__(RS_BLOCK) gather_params_kernel(
    const int32_t *__restrict__ order, int64_t npoints, const int32_t *initlen_p, int32_t *initlen_s,
    const double *tair_p, double *tair_s, const double *vz_p, double *vz_s, const double *rh_p,
    double *rh_s, const int32_t *cidx_p = nullptr, int32_t *cidx_s = nullptr,
    const double *ctsurf_p = nullptr, double *ctsurf_s = nullptr,
    const double *geo_p = nullptr, double *geo_s = nullptr, int64_t geo_stride = 0) {
  __builtin_amdgcn_s_setprio(3); /* a link of the chain between two step launches of a block: all of it is waited for */
  const int64_t s = (int64_t)blockIdx.x * RS_BLOCK + threadIdx.x;
  if (s >= npoints) return;
  const int64_t p = order[s];
  initlen_s[s] = initlen_p[p];
  tair_s[s] = tair_p[p];
  vz_s[s] = vz_p[p];
  rh_s[s] = rh_p[p];
  if (cidx_s) { /* coupling index and observation travel with the slot too */
    cidx_s[s] = cidx_p[p];
    ctsurf_s[s] = ctsurf_p[p];
  }
  if (geo_s) /* sky view factor, sin/cos of the latitude, longitude: four scalars; the 360-column horizon
                table stays in point order and is read through the order row (RsPointParams::horizon_index) */
    for (int q = 0; q < 4; ++q) geo_s[q * geo_stride + s] = geo_p[q * geo_stride + p];
}

/* output rows of one launch, written in slot order, into the natural-order result */
__global__ void __launch_bounds__(RS_BLOCK) unpermute_rows_kernel(
    const int32_t *__restrict__ order, int64_t npoints, const double *__restrict__ chunk_out,
    int64_t chunk_rows, double *final_out, int64_t final_rows, int64_t row0, int32_t nrows,
    int64_t stride) {
  __builtin_amdgcn_s_setprio(3); /* a link of the chain between two step launches of a block: all of it is waited for */
  const int64_t s = (int64_t)blockIdx.x * RS_BLOCK + threadIdx.x;
  if (s >= npoints) return;
  const int64_t p = order[s];
  const int f = blockIdx.y;
  for (int32_t r = 0; r < nrows; ++r)
    final_out[((int64_t)f * final_rows + row0 + r) * stride + p] =
        chunk_out[((int64_t)f * chunk_rows + r) * stride + s];
}

__global__ void __launch_bounds__(RS_BLOCK) fill_i32_kernel(int32_t *x, int64_t n, int32_t v) {
  const int64_t i = (int64_t)blockIdx.x * RS_BLOCK + threadIdx.x;
  if (i < n) x[i] = v;
}
__global__ void __launch_bounds__(RS_BLOCK) fill_f64_kernel(double *x, int64_t n, double v) {
  const int64_t i = (int64_t)blockIdx.x * RS_BLOCK + threadIdx.x;
  if (i < n) x[i] = v;
}

void launch_expand_raw(bool any_pp, int64_t mp, const ExpandRawArgs &ea, hipStream_t stream) {
  const dim3 g((unsigned)(mp / RS_BLOCK), NFLD), b(RS_BLOCK);
  if (any_pp)
    hipLaunchKernelGGL(expand_raw_kernel<true>, g, b, 0, stream, ea);
  else
    hipLaunchKernelGGL(expand_raw_kernel<false>, g, b, 0, stream, ea);
}

inline dim3 grid1(int64_t n) { return dim3((unsigned)((n + RS_BLOCK - 1) / RS_BLOCK)); }

/* ---- host side ------------------------------------------------------------------ */

int fail_msg(const char *msg, int code) {
  rs_host_set_error(msg);
  return code;
}
int fail_hip(const char *what, hipError_t e) {
  char buf[300];
  snprintf(buf, sizeof(buf), "rs_driver: %s: %s", what, hipGetErrorString(e));
  rs_host_set_error(buf);
  return -10;
}
#define HOK(expr)                                     \
  do {                                                \
    hipError_t e_ = (expr);                           \
    if (e_ != hipSuccess) return fail_hip(#expr, e_); \
  } while (0)

/* The time walk of JsonSource::interpolate (JsonSource.cpp:49-85,113-114,171-175) without
 * the data: which raw interval each simulation index uses and how. */
void build_plan(const int64_t *rawtime, int rawLen, const std::vector<int64_t> &simtime,
                std::vector<PlanStep> &plan) {
  const int simLen = (int)simtime.size();
  plan.assign(simLen, PlanStep{K_NONE, 0, 0.0, 1.0, 0.0});
  if (rawLen == 0) return; /* JsonSource.cpp:233-237 */
  int rawPos = 0, simPos = 0;
  if (rawtime[0] < simtime[0]) {
    for (rawPos = 0; rawPos < rawLen; ++rawPos)
      if (rawtime[rawPos] >= simtime[0]) break;
    rawPos = rawPos - 1;
    simPos = 0;
  } else if (simtime[0] < rawtime[0]) {
    for (simPos = 0; simPos < simLen; ++simPos)
      if (simtime[simPos] >= rawtime[0]) break;
    rawPos = 0;
  }
  while (rawPos + 1 < rawLen && simPos < simLen) {
    if (std::llabs(simtime[simPos] - rawtime[rawPos]) < 0.01) {
      plan[simPos] = PlanStep{K_COPY, rawPos, 0.0, 1.0, 0.0};
      simPos++;
    } else if (std::llabs(simtime[simPos] - rawtime[rawPos + 1]) < 0.01) {
      rawPos++;
    } else {
      const double den = (double)(rawtime[rawPos + 1] - rawtime[rawPos]);
      /* a positive whole number of seconds below 2^53: its significand is never all ones, the
       * one case rs_div_u's reciprocal does not cover; anything else divides the IEEE way */
      /* ... and for 0 < num < den < 2^40 (times that increase, as the reference assumes) the interpolated
       * value provably lies between the two raw ends (rs_raw.hpp): what the segment scan and the step
       * kernel's own interpolation (rs_kernels.hip raw_forcing) rely on; an entry outside that is marked
       * by rden = 0 and handled index by index */
      const double num = (double)(simtime[simPos] - rawtime[rawPos]);
      const double rden = (den >= 1.0 && den < 1.0995e12 && num > 0.0 && num < den) ? 1.0 / den : 0.0;
      plan[simPos] = PlanStep{K_INTERP, rawPos, num, den, rden};
      simPos++;
    }
  }
}

const double *raw_field(const RsRawSource &s, int fld) {
  switch (fld) {
    case R_TAIR: return s.tair;
    case R_TDEW: return s.tdew;
    case R_VZ: return s.vz;
    case R_RHZ: return s.rhz;
    case R_PREC: return s.prec;
    case R_SW: return s.sw;
    case R_LW: return s.lw;
    case R_SWDIR: return s.sw_dir;
    case R_LWNET: return s.lw_net;
    default: return s.tsurfobs;
  }
}

struct Common {
  int n = 0, nsrc = 0, L = 0, DT = 0;
  int default_initlen = 0, cplLen = 0;
  std::vector<int64_t> simtime;
  std::vector<std::vector<PlanStep>> plans;
  std::vector<ScanSeg> segs; /* shared axes only (scan_seg_kernel); empty: some source has per-point axes */
  std::vector<std::vector<int32_t>> active_prefix; /* [source][i]: plan entries != K_NONE among indices < i */
};

int prepare(const RsDriverInput *in, const InputSettings *st, Common &c) {
  if (!in || !st || in->n_points < 1 || in->n_sources < 1 || in->n_sources > RS_MAX_SOURCES ||
      !in->sources)
    return fail_msg("rs_driver: bad arguments (n_points >= 1, 1 <= n_sources <= RS_MAX_SOURCES)", -1);
  if (st->SimLen < 1 || !(st->DTSecs >= 1.0))
    return fail_msg("rs_driver: SimLen >= 1 and DTSecs >= 1 required", -1);
  c.n = in->n_points;
  c.nsrc = in->n_sources;
  c.L = st->SimLen;
  c.DT = (int)st->DTSecs; /* JsonSource takes `const int DTSecs` */
  c.simtime.resize(c.L);
  for (int k = 0; k < c.L; ++k) c.simtime[k] = in->start_time + (int64_t)k * c.DT;
  /* roadrunner.cpp:168-169: time_t / double, truncated */
  c.default_initlen = 1 + (int)((double)(in->forecast_time - in->start_time) / st->DTSecs);
  /* roadrunner.cpp:263: static_cast<int>(coupling_minutes * 60 / DTSecs) */
  c.cplLen = (int)((double)(st->coupling_minutes * 60) / st->DTSecs);
  c.plans.resize(c.nsrc);
  for (int s = 0; s < c.nsrc; ++s) {
    const RsRawSource &rs = in->sources[s];
    if (rs.n_times < 0 || (rs.n_times > 0 && !rs.times))
      return fail_msg("rs_driver: source without a time axis", -1);
    if (rs.times_per_point) {
      if (rs.lengths)
        for (int p = 0; p < c.n; ++p)
          if (rs.lengths[p] < 0 || rs.lengths[p] > rs.n_times)
            return fail_msg("rs_driver: lengths[p] outside 0..n_times", -1);
      c.plans[s].assign(c.L, PlanStep{K_NONE, 0, 0.0, 1.0, 0.0}); /* unused: the walk runs on the device */
    } else {
      if (rs.lengths) return fail_msg("rs_driver: lengths given without times_per_point", -1);
      build_plan(rs.times, rs.n_times, c.simtime, c.plans[s]);
    }
  }
  c.active_prefix.assign(c.nsrc, std::vector<int32_t>(c.L + 1, 0));
  for (int s = 0; s < c.nsrc; ++s)
    for (int i = 0; i < c.L; ++i) c.active_prefix[s][i + 1] = c.active_prefix[s][i] + (c.plans[s][i].kind != K_NONE ? 1 : 0);
  bool any_pp = false;
  for (int s = 0; s < c.nsrc; ++s) any_pp = any_pp || (in->sources[s].times_per_point && in->sources[s].n_times > 0);
  c.segs.clear();
  if (!any_pp && !getenv("ROADSURF_HIP_SCAN_FULL")) {
    for (int i = 0; i < c.L; ++i) {
      ScanSeg g{};
      g.i0 = i;
      g.i1 = i + 1;
      for (int s = 0; s < RS_MAX_SOURCES; ++s) {
        g.kind[s] = s < c.nsrc ? c.plans[s][i].kind : K_NONE;
        g.rp[s] = s < c.nsrc ? c.plans[s][i].rp : 0;
      }
      bool same = !c.segs.empty();
      if (same)
        for (int s = 0; s < RS_MAX_SOURCES; ++s)
          same = same && c.segs.back().kind[s] == g.kind[s] && (g.kind[s] == K_NONE || c.segs.back().rp[s] == g.rp[s]);
      if (same) c.segs.back().i1 = i + 1;
      else c.segs.push_back(g);
    }
  }
  return 0;
}

/* Device copies of one tile's raw data + plans. */
struct TileRaw {
  Dev plan[RS_MAX_SOURCES];
  Dev ptimes[RS_MAX_SOURCES], plen[RS_MAX_SOURCES], prp[RS_MAX_SOURCES];
  Dev fld[RS_MAX_SOURCES][NFLD];
  bool any_pp = false;
  Dev stage; /* landing block of the H2D copies */
  Dev segs;  /* Common::segs */
  SrcSet S{};
};

/* The tile's raw series onto the device, in two halves.  `upload_copies` lands every host array of the
 * tile - point-major rows [m][n_times], as the caller holds them - in ONE landing block with back-to-
 * back copies and waits for them: that is the part that owns the PCIe link, and the part the workers of
 * a device take in turns (rs_devices.hpp: copy_gate).  `upload_finish` turns the rows into the
 * [n_times][mp] columns the kernels read (LDS-tiled transposes), completes Tdew / RH and positions the
 * per-point walks - device work on the worker's own stream, beside the next worker's copies.  (Round 3
 * had one landing buffer per field, so copy and transpose alternated inside the gate and the blocks of a
 * call started ~38 ms apart; the copies alone take less than half of that.) */
struct TileLanding {
  size_t off[RS_MAX_SOURCES][NFLD + 1] = {}; /* byte offset of (source, field) in the landing block; NFLD: the times */
  bool has[RS_MAX_SOURCES][NFLD + 1] = {};
  /* copies issued on a stream of their own (the worker's copy stream): landed[s][f] is recorded behind the
   * copy of (source, field), and the worker's stream waits for it before it touches the piece - the
   * transposes of the first fields run while the last fields are still on the link (round 5: a block's first
   * step launch used to wait for all copies AND then all transposes, 16 + 6 ms at 250 000 points) */
  hipEvent_t landed[RS_MAX_SOURCES][NFLD + 1] = {};
  bool transposed[RS_MAX_SOURCES][NFLD] = {}; /* upload_copies has issued the piece's transpose already */
  hipEvent_t extra[2] = {nullptr, nullptr};    /* tile start, horizons landed */
  bool async = false;
  ~TileLanding() {
    for (auto &row : landed)
      for (hipEvent_t e : row)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : extra)
      if (e) (void)hipEventDestroy(e);
  }
};

int upload_copies(const RsDriverInput *in, const Common &c, int64_t p0, int m, int64_t mp,
                  TileRaw &T, TileLanding &Ld, hipStream_t stream, hipStream_t work = nullptr) {
  size_t total = 0;
  for (int s = 0; s < c.nsrc; ++s) {
    const RsRawSource &rs = in->sources[s];
    if (rs.n_times < 1) continue;
    const size_t piece = ((size_t)m * rs.n_times * sizeof(double) + 255) & ~(size_t)255;
    if (rs.times_per_point) {
      Ld.off[s][NFLD] = total;
      Ld.has[s][NFLD] = true;
      total += piece;
    }
    for (int f = 0; f < NFLD; ++f)
      if (raw_field(rs, f)) {
        Ld.off[s][f] = total;
        Ld.has[s][f] = true;
        total += piece;
      }
  }
  HOK(T.stage.alloc(total));
  char *base = T.stage.as<char>();
  for (int s = 0; s < c.nsrc; ++s) {
    const RsRawSource &rs = in->sources[s];
    if (Ld.has[s][NFLD]) {
      HOK(hipMemcpyAsync(base + Ld.off[s][NFLD], rs.times + (size_t)p0 * rs.n_times,
                         (size_t)m * rs.n_times * sizeof(int64_t), hipMemcpyHostToDevice, stream));
      if (Ld.async) {
        HOK(hipEventCreateWithFlags(&Ld.landed[s][NFLD], hipEventDisableTiming));
        HOK(hipEventRecord(Ld.landed[s][NFLD], stream));
      }
    }
    for (int f = 0; f < NFLD; ++f)
      if (Ld.has[s][f]) {
        HOK(hipMemcpyAsync(base + Ld.off[s][f], raw_field(rs, f) + (size_t)p0 * rs.n_times,
                           (size_t)m * rs.n_times * sizeof(double), hipMemcpyHostToDevice, stream));
        if (Ld.async) {
          HOK(hipEventCreateWithFlags(&Ld.landed[s][f], hipEventDisableTiming));
          HOK(hipEventRecord(Ld.landed[s][f], stream));
          if (work) { /* the piece's transpose right behind it, on the worker's stream, beside the next copy
                         (a copy from pageable memory returns when its bytes are on their way) */
            HOK(T.fld[s][f].alloc((size_t)rs.n_times * mp * sizeof(double)));
            HOK(hipStreamWaitEvent(work, Ld.landed[s][f], 0));
            HOK(transpose(reinterpret_cast<const double *>(base + Ld.off[s][f]), T.fld[s][f].as<double>(), m,
                          rs.n_times, rs.n_times, mp, work));
            Ld.transposed[s][f] = true;
          }
        }
      }
  }
  return 0;
}

int upload_finish(const RsDriverInput *in, const Common &c, int64_t p0, int m, int64_t mp,
                  TileRaw &T, const TileLanding &Ld, hipStream_t stream) {
  T.S.nsrc = c.nsrc;
  T.S.simlen = c.L;
  T.S.np_pad = mp;
  T.S.npoints = m;
  T.S.sim0 = in->start_time;
  T.S.dt = c.DT;
  const char *base = T.stage.as<char>();
  for (int s = 0; s < c.nsrc; ++s) {
    const RsRawSource &rs = in->sources[s];
    SrcDev &d = T.S.src[s];
    d.n_times = rs.n_times;
    d.is_obs = rs.is_observation;
    d.plan = nullptr;
    d.ptimes = nullptr;
    d.plen = nullptr;
    d.prp = nullptr;
    if (rs.times_per_point && rs.n_times > 0) {
      /* per-point axes: times [m][n_times] -> [n_times][mp], lengths, walk positions */
      HOK(T.ptimes[s].alloc((size_t)rs.n_times * mp * sizeof(int64_t)));
      if (Ld.landed[s][NFLD]) HOK(hipStreamWaitEvent(stream, Ld.landed[s][NFLD], 0));
      HOK(transpose(reinterpret_cast<const int64_t *>(base + Ld.off[s][NFLD]), T.ptimes[s].as<int64_t>(), m,
                    rs.n_times, rs.n_times, mp, stream));
      HOK(T.plen[s].alloc(mp * sizeof(int32_t)));
      if (rs.lengths) {
        HOK(hipMemsetAsync(T.plen[s].p, 0, mp * sizeof(int32_t), stream));
        HOK(hipMemcpyAsync(T.plen[s].p, rs.lengths + p0, (size_t)m * sizeof(int32_t),
                           hipMemcpyHostToDevice, stream));
      } else {
        hipLaunchKernelGGL(fill_i32_kernel, grid1(mp), dim3(RS_BLOCK), 0, stream,
                           T.plen[s].as<int32_t>(), mp, (int32_t)rs.n_times);
        HOK(hipGetLastError());
      }
      HOK(T.prp[s].alloc(mp * sizeof(int32_t)));
      d.ptimes = T.ptimes[s].as<int64_t>();
      d.plen = T.plen[s].as<int32_t>();
      d.prp = T.prp[s].as<int32_t>();
      T.any_pp = true;
    } else {
      HOK(T.plan[s].alloc((size_t)c.L * sizeof(PlanStep)));
      HOK(hipMemcpyAsync(T.plan[s].p, c.plans[s].data(), (size_t)c.L * sizeof(PlanStep),
                         hipMemcpyHostToDevice, stream));
      d.plan = T.plan[s].as<PlanStep>();
    }
    for (int f = 0; f < NFLD; ++f) {
      const bool h = Ld.has[s][f];
      d.fld[f] = nullptr;
      /* Tdew and RH can be completed from each other: both exist if either does */
      const bool derived = (f == R_TDEW && rs.rhz && rs.tair) || (f == R_RHZ && rs.tdew && rs.tair);
      if ((!h && !derived) || rs.n_times == 0) continue;
      const size_t ne = (size_t)rs.n_times * mp;
      if (h && Ld.transposed[s][f]) {
        d.fld[f] = T.fld[s][f].as<double>();
        continue;
      }
      HOK(T.fld[s][f].alloc(ne * sizeof(double)));
      double *dst = T.fld[s][f].as<double>();
      if (h) {
        if (Ld.landed[s][f]) HOK(hipStreamWaitEvent(stream, Ld.landed[s][f], 0));
        HOK(transpose(reinterpret_cast<const double *>(base + Ld.off[s][f]), dst, m, rs.n_times, rs.n_times,
                      mp, stream));
      } else {
        hipLaunchKernelGGL(fill_f64_kernel, grid1((int64_t)ne), dim3(RS_BLOCK), 0, stream, dst,
                           (int64_t)ne, -9999.9);
        HOK(hipGetLastError());
      }
      d.fld[f] = dst;
    }
    /* JsonSource.cpp:288-295 (needs the math tables: the caller has created a plan) */
    if (d.fld[R_TAIR] && d.fld[R_TDEW] && d.fld[R_RHZ])
      HOK(rs_launch_humidity_fill(d.fld[R_TAIR], const_cast<double *>(d.fld[R_TDEW]),
                                  const_cast<double *>(d.fld[R_RHZ]), (int64_t)rs.n_times * mp,
                                  stream));
  }
  if (T.any_pp) {
    hipLaunchKernelGGL(pp_init_kernel, grid1(mp), dim3(RS_BLOCK), 0, stream, T.S);
    HOK(hipGetLastError());
  }
  if (!c.segs.empty()) {
    HOK(T.segs.alloc(c.segs.size() * sizeof(ScanSeg)));
    HOK(hipMemcpyAsync(T.segs.p, c.segs.data(), c.segs.size() * sizeof(ScanSeg), hipMemcpyHostToDevice, stream));
  }
  return 0;
}

int upload_tile(const RsDriverInput *in, const Common &c, int64_t p0, int m, int64_t mp,
                TileRaw &T, hipStream_t stream) {
  TileLanding Ld;
  if (int rc = upload_copies(in, c, p0, m, mp, T, Ld, stream)) return rc;
  return upload_finish(in, c, p0, m, mp, T, Ld, stream);
}

/* Per-point decisions of read_input for one tile (device arrays, [mp]). */
struct TileDecisions {
  Dev first_missing, last_obs, cpl_i, cpl_t;
  Dev status, missing_index, initlen, cpl_index, cpl_hi, tair_relax, vz_relax, rh_relax, cpl_tsurf;
};

int decide_tile(const Common &c, const InputSettings *st, const TileRaw &T, TileDecisions &D,
                hipStream_t stream) {
  const int64_t mp = T.S.np_pad;
  HOK(D.first_missing.alloc((size_t)6 * mp * sizeof(int32_t)));
  HOK(D.last_obs.alloc(mp * sizeof(int32_t)));
  HOK(D.cpl_i.alloc(mp * sizeof(int32_t)));
  HOK(D.cpl_t.alloc(mp * sizeof(double)));
  for (Dev *d : {&D.status, &D.missing_index, &D.initlen, &D.cpl_index, &D.cpl_hi})
    HOK(d->alloc(mp * sizeof(int32_t)));
  for (Dev *d : {&D.tair_relax, &D.vz_relax, &D.rh_relax, &D.cpl_tsurf})
    HOK(d->alloc(mp * sizeof(double)));
  ScanArgs sa;
  sa.S = T.S;
  sa.first_missing = D.first_missing.as<int32_t>();
  sa.last_obs = D.last_obs.as<int32_t>();
  sa.cpl_i = D.cpl_i.as<int32_t>();
  sa.cpl_t = D.cpl_t.as<double>();
  if (T.any_pp)
    hipLaunchKernelGGL(scan_raw_kernel<true>, dim3((unsigned)(mp / RS_BLOCK), 7), dim3(RS_BLOCK), 0,
                       stream, sa);
  else if (!c.segs.empty() && T.segs.p)
    hipLaunchKernelGGL(scan_seg_kernel, dim3((unsigned)(mp / RS_BLOCK), 7), dim3(RS_BLOCK), 0, stream, sa,
                       (const ScanSeg *)T.segs.as<ScanSeg>(), (int32_t)c.segs.size());
  else
    hipLaunchKernelGGL(scan_raw_kernel<false>, dim3((unsigned)(mp / RS_BLOCK), 7), dim3(RS_BLOCK), 0,
                       stream, sa);
  HOK(hipGetLastError());
  FinalArgs fa;
  fa.S = T.S;
  fa.first_missing = sa.first_missing;
  fa.last_obs = sa.last_obs;
  fa.cpl_i = sa.cpl_i;
  fa.cpl_t = sa.cpl_t;
  fa.use_relaxation = st->use_relaxation;
  fa.use_coupling = st->use_coupling;
  fa.cplLen = c.cplLen;
  fa.default_initlen = c.default_initlen;
  fa.status = D.status.as<int32_t>();
  fa.missing_index = D.missing_index.as<int32_t>();
  fa.initlen = D.initlen.as<int32_t>();
  fa.cpl_index = D.cpl_index.as<int32_t>();
  fa.cpl_hi = D.cpl_hi.as<int32_t>();
  fa.tair_relax = D.tair_relax.as<double>();
  fa.vz_relax = D.vz_relax.as<double>();
  fa.rh_relax = D.rh_relax.as<double>();
  fa.cpl_tsurf = D.cpl_tsurf.as<double>();
  hipLaunchKernelGGL(finalize_kernel, grid1(mp), dim3(RS_BLOCK), 0, stream, fa);
  HOK(hipGetLastError());
  return 0;
}

/* Copy the decisions back into the caller's LocalParameters / status arrays the way
 * read_input leaves them.  In two halves: `issue` enqueues the device-to-host copies (into page-locked
 * memory the worker thread keeps) behind the decision kernels and returns; `finish` waits for them and
 * writes the caller's arrays.  A run without coupling finishes after it has enqueued the whole
 * simulation - nothing on the host needs the decisions before - so the time loop starts without
 * waiting for the upload and the scan to drain. */
struct TileReport {
  int m = 0;
  int64_t p0 = 0;
  hipEvent_t ev = nullptr;
  char *h = nullptr; /* [4][m] int32 status, missing_index, initlen, cpl_index; [4][m] double relax x 3, cpl_tsurf */
  bool pending = false;
  ~TileReport() {
    if (ev) (void)hipEventDestroy(ev);
  }
};

inline char *report_staging(size_t bytes) {
  static thread_local rsu::Pinned buf;
  static thread_local size_t cap = 0;
  if (cap < bytes) {
    if (buf.p) (void)hipHostFree(buf.p);
    buf.p = nullptr;
    cap = 0;
    if (buf.alloc(bytes) != hipSuccess) return nullptr;
    cap = bytes;
  }
  return static_cast<char *>(buf.p);
}

int report_issue(const TileDecisions &D, int64_t p0, int m, TileReport &R, hipStream_t stream) {
  R.m = m;
  R.p0 = p0;
  R.h = report_staging((size_t)m * (4 * sizeof(int32_t) + 4 * sizeof(double)));
  if (!R.h) return fail_msg("rs_driver_run: no page-locked memory for the decisions", -10);
  int32_t *hi = reinterpret_cast<int32_t *>(R.h);
  double *hd = reinterpret_cast<double *>(R.h + (size_t)4 * m * sizeof(int32_t));
  const void *si[4] = {D.status.p, D.missing_index.p, D.initlen.p, D.cpl_index.p};
  const void *sd[4] = {D.tair_relax.p, D.vz_relax.p, D.rh_relax.p, D.cpl_tsurf.p};
  for (int k = 0; k < 4; ++k) {
    HOK(hipMemcpyAsync(hi + (size_t)k * m, si[k], (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    HOK(hipMemcpyAsync(hd + (size_t)k * m, sd[k], (size_t)m * sizeof(double), hipMemcpyDeviceToHost, stream));
  }
  if (!R.ev) HOK(hipEventCreateWithFlags(&R.ev, hipEventDisableTiming));
  HOK(hipEventRecord(R.ev, stream));
  R.pending = true;
  return 0;
}

int report_finish(const Common &c, const InputSettings *st, TileReport &R, LocalParameters *local,
                  int32_t *status, int32_t *missing_index) {
  if (!R.pending) return 0;
  HOK(hipEventSynchronize(R.ev));
  R.pending = false;
  const int m = R.m;
  const int64_t p0 = R.p0;
  const int32_t *hs = reinterpret_cast<const int32_t *>(R.h), *hm = hs + m, *hi = hs + 2 * (size_t)m,
                *hc = hs + 3 * (size_t)m;
  const double *tr = reinterpret_cast<const double *>(R.h + (size_t)4 * m * sizeof(int32_t)), *vr = tr + m,
               *rr = tr + 2 * (size_t)m, *ct = tr + 3 * (size_t)m;
  for (int p = 0; p < m; ++p) {
    if (status) status[p0 + p] = hs[p];
    if (missing_index) missing_index[p0 + p] = hm[p];
    if (!local) continue;
    LocalParameters &l = local[p0 + p];
    l.InitLenI = c.default_initlen; /* roadrunner.cpp:169, before anything can fail */
    if (hs[p] >= 1 && hs[p] <= 6) continue; /* read_input returned early */
    if (st->use_relaxation == 1) {
      l.tair_relax = tr[p];
      l.VZ_relax = vr[p];
      l.RH_relax = rr[p];
      l.InitLenI = hi[p];
    }
    if (st->use_coupling == 1 && hs[p] == 0) {
      l.couplingTsurf = ct[p];
      l.couplingIndexI = hc[p];
    }
  }
  return 0;
}

int report_tile(const Common &c, const InputSettings *st, const TileDecisions &D, int64_t p0, int m,
                LocalParameters *local, int32_t *status, int32_t *missing_index,
                hipStream_t stream) {
  TileReport R;
  if (int rc = report_issue(D, p0, m, R, stream)) return rc;
  return report_finish(c, st, R, local, status, missing_index);
}

int check_device(int32_t device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail_msg("rs_driver: no HIP device visible - this library has no CPU path", -9);
  if (device < 0 || device >= ndev) return fail_msg("rs_driver: device index out of range", -9);
  return 0;
}

/* ROADSURF_HIP_DRIVER_TIMING=1: wall time per phase of rs_driver_run on stderr (synchronises) */
struct PhaseTimer {
  bool on;
  hipStream_t s;
  double t0;
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  static double now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
  }
  explicit PhaseTimer(hipStream_t st) : on(getenv("ROADSURF_HIP_DRIVER_TIMING") != nullptr), s(st), t0(now()) {}
  void lap(int k) {
    if (!on) return;
    (void)hipStreamSynchronize(s);
    const double t = now();
    acc[k] += t - t0;
    t0 = t;
  }
  void report() const {
    if (!on) return;
    fprintf(stderr,
            "rs_driver_run phases [s]: setup %.3f  upload+transpose %.3f  scan/decide %.3f  alloc/params %.3f  "
            "expand+step %.3f  outputs %.3f  window alloc %.3f  free %.3f\n", acc[0], acc[1], acc[2], acc[3],
            acc[4], acc[5], acc[6], acc[7]);
  }
};

/* The forcing windows are the one large allocation (up to 64 GB with coupling).  The amdgpu
 * driver wipes VRAM when it is released, and a hipMalloc that lands on pages still being
 * wiped waits for them: measured here, a 60 GB hipMalloc right after a 60 GB hipFree takes
 * 3-6 s, against 0.6 s for the whole simulation.  So the window block is kept per process and
 * reused by the next call (rs_driver_release_cache frees it); a concurrent second caller gets a
 * private allocation. */
constexpr int WINCACHE_SLOTS = 8; /* concurrent rs_driver_run workers per device that keep a block */
struct WindowCache {
  std::mutex m;
  void *p[WINCACHE_SLOTS] = {};
  size_t bytes[WINCACHE_SLOTS] = {};
  bool busy[WINCACHE_SLOTS] = {};
} g_wincache[64], /* per device: the fan-out of rs_driver_run has several workers on each */
    g_arenacache[64]; /* the same for the workers' arenas (rs_devutil.hpp): every other buffer of a tile */

struct WindowLease {
  void *p = nullptr;
  bool cached = false;
  hipStream_t stream = nullptr; /* the stream whose kernels use the block */
  ~WindowLease() { release(); }
  int device = 0, slot = -1;
  WindowCache *cache = g_wincache;
  hipError_t acquire(size_t bytes, int dev) {
    device = dev & 63;
    WindowCache &c = cache[device];
    std::lock_guard<std::mutex> lk(c.m);
    /* a free slot that is large enough, else a free slot to (re)allocate */
    int pick = -1;
    for (int k = 0; k < WINCACHE_SLOTS && pick < 0; ++k)
      if (!c.busy[k] && c.p[k] && c.bytes[k] >= bytes) pick = k;
    for (int k = 0; k < WINCACHE_SLOTS && pick < 0; ++k)
      if (!c.busy[k] && !c.p[k]) pick = k;
    for (int k = 0; k < WINCACHE_SLOTS && pick < 0; ++k)
      if (!c.busy[k]) pick = k;
    if (pick >= 0) {
      if (c.p[pick] && c.bytes[pick] < bytes) {
        (void)hipFree(c.p[pick]);
        c.p[pick] = nullptr;
        c.bytes[pick] = 0;
      }
      if (!c.p[pick]) {
        hipError_t e = hipMalloc(&c.p[pick], bytes);
        if (e != hipSuccess) {
          c.p[pick] = nullptr;
          return e;
        }
        c.bytes[pick] = bytes;
      }
      c.busy[pick] = true;
      cached = true;
      slot = pick;
      p = c.p[pick];
      return hipSuccess;
    }
    cached = false;
    return hipMalloc(&p, bytes);
  }
  void release() {
    if (!p) return;
    /* error returns leave kernels in flight on the call's stream: nobody else may get the
     * block before they have drained (the lease is declared after the stream guard, so the
     * stream is still alive here) */
    if (stream) (void)hipStreamSynchronize(stream);
    if (cached) {
      std::lock_guard<std::mutex> lk(cache[device].m);
      cache[device].busy[slot] = false;
    } else {
      (void)hipFree(p);
    }
    p = nullptr;
  }
};

struct StreamGuard {
  hipStream_t s = nullptr;
  ~StreamGuard() {
    if (s) (void)hipStreamDestroy(s);
  }
};
struct PlanGuard {
  RsPlan *p = nullptr;
  ~PlanGuard() {
    if (p) rs_hip_plan_destroy(p);
  }
};

}  // namespace

extern "C" {

int rs_driver_expand(const RsDriverInput *in, const InputSettings *st, LocalParameters *local,
                     double *merged, int32_t *status, int32_t *missing_index, int32_t device) {
  Common c;
  if (int rc = prepare(in, st, c)) return rc;
  if (!merged) return fail_msg("rs_driver_expand: merged is required", -1);
  if (int rc = check_device(device)) return rc;
  HOK(hipSetDevice(device));
  StreamGuard sg;
  HOK(hipStreamCreate(&sg.s));
  /* the math tables live with the plans' constants: make one (any valid constants do) */
  InputSettings s15 = *st;
  InputParameters prm;
  rs_default_parameters(&prm, st->DTSecs);
  RsConstants consts;
  int32_t rc32 = 0;
  rs_build_constants(&s15, &prm, &consts, &rc32);
  if (rc32 != 0) return fail_msg("rs_driver_expand: bad settings", -1);
  const int P = std::min(c.n, 4096);
  for (int64_t p0 = 0; p0 < c.n; p0 += P) {
    const int m = (int)std::min<int64_t>(P, c.n - p0);
    PlanGuard pg;
    pg.p = rs_hip_plan_create(device, m, &consts, sg.s);
    if (!pg.p) return -11;
    const int64_t mp = rs_hip_plan_npoints_padded(pg.p);
    TileRaw T;
    if (int rc = upload_tile(in, c, p0, m, mp, T, sg.s)) return rc;
    TileDecisions D;
    if (int rc = decide_tile(c, st, T, D, sg.s)) return rc;
    if (int rc = report_tile(c, st, D, p0, m, local, status, missing_index, sg.s)) return rc;
    Dev win, pt;
    HOK(win.alloc((size_t)NFLD * c.L * mp * sizeof(double)));
    HOK(pt.alloc((size_t)m * c.L * sizeof(double)));
    ExpandRawArgs ea;
    ea.S = T.S;
    for (int f = 0; f < NFLD; ++f) ea.out[f] = win.as<double>() + (size_t)f * c.L * mp;
    ea.status = nullptr; /* the test hook shows what read_input returns, rejected or not */
    ea.order = nullptr;
    ea.cpl_hi = st->use_coupling == 1 ? D.cpl_hi.as<int32_t>() : nullptr;
    ea.cplLen = c.cplLen;
    ea.i0 = 0;
    ea.nsteps = c.L;
    ea.stride = mp;
    launch_expand_raw(T.any_pp, mp, ea, sg.s);
    HOK(hipGetLastError());
    for (int f = 0; f < NFLD; ++f) {
      HOK(transpose((const double *)ea.out[f], pt.as<double>(), c.L, m, mp, c.L, sg.s));
      HOK(hipMemcpyAsync(merged + ((size_t)f * c.n + p0) * c.L, pt.p, (size_t)m * c.L * sizeof(double),
                         hipMemcpyDeviceToHost, sg.s));
    }
    HOK(hipStreamSynchronize(sg.s));
  }
  return 0;
}

void rs_driver_release_cache(void) {
  for (int d = 0; d < 128; ++d) {
    WindowCache &c = d < 64 ? g_wincache[d] : g_arenacache[d - 64];
    std::lock_guard<std::mutex> lk(c.m);
    for (int k = 0; k < WINCACHE_SLOTS; ++k) {
      if (c.busy[k] || !c.p[k]) continue;
      if (hipSetDevice(d & 63) != hipSuccess) break;
      (void)hipFree(c.p[k]);
      c.p[k] = nullptr;
      c.bytes[k] = 0;
    }
  }
}

static int driver_run_range(const RsDriverInput *in, const InputSettings *st,
                            const InputParameters *params, LocalParameters *local,
                            const RsDriverOutput *out, int32_t device, int64_t pbeg, int64_t pend);

/* tiles the calling thread's last single-device rs_driver_run stepped (tests: the window budget) */
static thread_local int g_last_tiles = 0, g_last_raw_launches = 0;
int rs_driver_last_tiles(void) { return g_last_tiles; }
/* ... and how many of its step launches made their forcing from the raw series (rs_step_raw) */
int rs_driver_last_raw_launches(void) { return g_last_raw_launches; }

/* device >= 0: that device.  device < 0: the points are cut into contiguous blocks over the
 * device list (rs_devices.hpp: ROADSURF_HIP_DEVICES, default every visible device), one host
 * thread + stream + plans per device, no collective - the in-process counterpart of the
 * reference driver's worker pool (examples/example1/src/roadrunner.cpp:423-501). */
int rs_driver_run(const RsDriverInput *in, const InputSettings *st, const InputParameters *params,
                  LocalParameters *local, const RsDriverOutput *out, int32_t device) {
  if (!in || in->n_points < 1) return fail_msg("rs_driver_run: bad arguments", -1);
  if (device >= 0) {
    rsu::g_last_fanout = 1;
    return driver_run_range(in, st, params, local, out, device, 0, in->n_points);
  }
  /* four blocks per device.  (Six for batches with local horizons were 4 % faster while the horizon table
   * was transposed on the device, tools/experiments/r4_blocks.sh; with the table left in the caller's layout
   * - RsPointParams::horizons_by_point - four and six are level: 1.047e10 / 1.046e10 over three alternating
   * runs each.) */
  /* With local horizons a block uploads twice the bytes, the blocks start 38 instead of 19 ms apart and - of equal
   * size - end that far apart too: there the blocks shrink, the last to 70 % of the first (+1.4 % over three
   * tapers, profiles/r05_ab_block_taper.txt; without horizons equal blocks are as good). */
  const std::vector<rsu::Shard> shards =
      rsu::make_shards(in->n_points, rsu::device_list(), in->horizons ? 30 : RS_BLOCK_TAPER_PCT_DEFAULT);
  return rsu::fan_out(shards, [&](const rsu::Shard &sh, int) {
    return driver_run_range(in, st, params, local, out, sh.device, sh.off, sh.off + sh.cnt);
  });
}

} /* extern "C" */

/* points [pbeg, pend) of the input on one device */
static int driver_run_range(const RsDriverInput *in, const InputSettings *st,
                            const InputParameters *params, LocalParameters *local,
                            const RsDriverOutput *out, int32_t device, int64_t pbeg, int64_t pend) {
  Common c;
  if (int rc = prepare(in, st, c)) return rc;
  if (!params || !local || !out) return fail_msg("rs_driver_run: params, local and out are required", -1);
  if (!in->year || !in->month || !in->day || !in->hour || !in->minute || !in->second)
    return fail_msg("rs_driver_run: the calendar arrays of the simulation times are required", -1);
  /* roadrunner.cpp:290: int step = outputStep*60/DTSecs */
  const int step = (int)((double)(st->outputStep * 60) / st->DTSecs);
  if (step < 1) return fail_msg("rs_driver_run: outputStep*60/DTSecs < 1", -1);
  const int n_out = (c.L + step - 1) / step;
  if (out->n_out != n_out) {
    char b[160];
    snprintf(b, sizeof(b), "rs_driver_run: n_out must be ceil(SimLen/step) = %d (step %d)", n_out, step);
    return fail_msg(b, -1);
  }
  RsConstants consts;
  int32_t rc32 = 0;
  rs_build_constants(st, params, &consts, &rc32);
  if (rc32 != 0)
    return fail_msg("rs_driver_run: bad settings (NLayers in 5..32, SimLen >= 1, DTSecs > 0)", -1);
  if (int rc = check_device(device)) return rc;
  HOK(hipSetDevice(device));
  StreamGuard sg, sg_copy;
  HOK(hipStreamCreate(&sg.s));
  /* The uploads of a tile on a stream of their own, the transposes behind their pieces (TileLanding): measured
   * and switched off - a process has four hardware queues, and with two streams per worker two blocks' compute
   * streams can land on one queue, their step kernels then take turns (one queue with 102 of a call's 204 step
   * launches, the call 0.64 s instead of 0.58 s; round 5 again: 0.35 -> 0.43 s, profiles/r05_ab_upload_stream.txt).
   * A constant since round 6 (it was ROADSURF_HIP_UPLOAD_STREAM). */
  constexpr bool upload_stream = false;
  if (upload_stream) HOK(hipStreamCreate(&sg_copy.s));
  hipStream_t stream = sg.s, copy_stream = upload_stream ? sg_copy.s : sg.s;

  const int L = c.L;
  const bool coupled = st->use_coupling == 1;
  bool skyview = false;
  for (int p = 0; p < c.n; ++p)
    if (local[p].sky_view < 1.0 && local[p].sky_view > (double)-0.01f) skyview = true;
  const double tbottom = rs_bottom_temperature(params, &consts, in->year[0], in->month[0], in->day[0]);

  const char *ep = getenv("ROADSURF_HIP_TILE_POINTS"), *et = getenv("ROADSURF_HIP_CHUNK_STEPS");
  /* Tile of points: large enough to fill the chip (256 CUs x 4 workgroups of 256 points is
   * 262144 points per round).  With coupling the windows hold the whole series (a point
   * replays its coupling window), so the tile follows from a 64 GB window budget. */
  /* Coupling runs time-chunked too (rs_hip_step_cpl / rs_hip_cpl_replay): lock-step
   * chunks that park a point behind its coupling window, replay rounds over a window-sized block,
   * lock-step chunks again - with sky view too (in natural order: the per-point geometry is not
   * gathered into a plan order). */
  const bool cpl_chunked = coupled && !getenv("ROADSURF_HIP_CPL_WHOLE");
  /* The blocks' step kernel makes its forcing from the raw series itself (rs_step_raw: the two-wavefront
   * flavour, ground wave = JsonSource::interpolate + overlay one index ahead) wherever it can: sources on
   * shared time axes (the segment table exists), NLayers = 15, no output depth.  No forcing window, no
   * expansion kernel - with coupling for the lock-step chunks; the replay rounds keep a window over the
   * coupling windows of the tile (rs_hip_cpl_replay).  ROADSURF_HIP_DRIVER_WINDOWS=1: the windows and the one-point-per-lane
   * kernels as before (tests compare the two). */
  const bool use_raw = !c.segs.empty() && (!coupled || cpl_chunked) && consts.NLayers == 15 &&
                       !(st->tsurfOutputDepth >= 0.0) && !getenv("ROADSURF_HIP_DRIVER_WINDOWS");
  int64_t Pdef = 524288;
  if (coupled && !cpl_chunked) {
    Pdef = (int64_t)(64e9 / ((double)L * NFLD * sizeof(double)));
    Pdef = std::max<int64_t>(4096, std::min<int64_t>(262144, Pdef / 4096 * 4096));
  }
  int64_t Pcap = INT64_MAX;
  if (use_raw) {
    /* the raw-series step kernels address a tile's whole output window (every decimated row of the series,
     * padded stride) with 32-bit offsets: rs_step_raw / rs_cpl_replay_raw refuse a window of rs_a32_limit()
     * elements per stream or more (e.g. 250 000 points x 48 h with outputStep = 1 min: 7.2e8), so the tile is
     * cut to fit - whatever ROADSURF_HIP_TILE_POINTS asks for */
    Pcap = std::max<int64_t>(RS_BLOCK, (int64_t)((rs_a32_limit() - 1) / (uint64_t)n_out) / RS_BLOCK * RS_BLOCK);
  }
  const int P = (int)std::min<int64_t>(std::min<int64_t>(pend - pbeg, Pcap), ep ? std::max(1, atoi(ep)) : Pdef);
  /* A block of a few wavefronts (the reference's operational example: 401 stations) is the latency of its
   * dependent steps whatever the order of its points: no re-sorts, and launches of eight hours (154 against 165 ms
   * per call of that example, profiles/r05_operational_shape.txt) */
  const bool small_block = pend - pbeg < 4096;
  const char *ec = getenv("ROADSURF_HIP_CLUSTER"); /* 0 / 1: natural / plan order whatever the size */
  const bool want_cluster = ec ? atoi(ec) != 0 : !small_block;
  const int TC = (coupled && !cpl_chunked) ? L : std::min(L, et ? std::max(1, atoi(et)) : use_raw ? (want_cluster ? RS_DRIVER_RAW_CHUNK : 960) : 256);

  /* Every buffer of a tile other than the forcing windows comes out of one block this worker keeps
   * across calls (rs_devutil.hpp: Arena): no hipMalloc / hipFree inside the tile loop.  The size is
   * an estimate from the tile's shape; what does not fit is allocated the old way. */
  WindowLease arena_lease;
  arena_lease.stream = stream;
  arena_lease.cache = g_arenacache;
  rsu::Arena arena;
  struct ArenaScope {
    rsu::Arena *prev;
    explicit ArenaScope(rsu::Arena *a) : prev(rsu::tls_arena()) { rsu::tls_arena() = a; }
    ~ArenaScope() { rsu::tls_arena() = prev; }
  };
  {
    const size_t mpx = ((size_t)P + RS_BLOCK - 1) / RS_BLOCK * RS_BLOCK;
    size_t raw = 0;
    for (int k = 0; k < c.nsrc; ++k) {
      const size_t nt = (size_t)std::max(in->sources[k].n_times, 1);
      raw += nt * mpx * 8 * (NFLD + 1) + (size_t)L * sizeof(PlanStep) + 3 * mpx * 8;
    }
    const int step_e = std::max(1, (int)((double)(st->outputStep * 60) / st->DTSecs));
    const size_t n_out_e = ((size_t)L + step_e - 1) / step_e;
    const size_t rows_e = cpl_chunked ? (size_t)std::max(TC, std::min(L, c.cplLen + 2)) : (size_t)TC;
    size_t need = 2 * raw                                /* raw series + their landing block */
                  + mpx * 160                            /* decisions, bottom temperature, slot-order copies */
                  + mpx * rows_e * 4                     /* PrecPhase window */
                  + 7 * mpx * n_out_e * 8                /* outputs + their point-major copy */
                  + 6 * mpx * ((size_t)TC / step_e + 2) * 8 /* one launch's rows in slot order */
                  + 6 * mpx * 8 + (size_t)L * 40;        /* previews, hour, sun */
    if (skyview) need += 2 * (size_t)360 * mpx * 8 + 8 * mpx * 8;
    need += mpx * ((size_t)2 * RS_NSTATE * 8 + 64) + ((size_t)16 << 20); /* the tile's plan: two state blocks, order rows, sort scratch */
    need += need / 16 + ((size_t)64 << 10) * 64; /* alignment of ~60 pieces, slack */
    if (arena_lease.acquire(need, device) == hipSuccess) {
      arena.base = static_cast<char *>(arena_lease.p);
      arena.cap = need;
    }
  }
  ArenaScope arena_scope(arena.base ? &arena : nullptr);

  /* shared axes */
  Dev d_hour, d_sun;
  HOK(d_hour.alloc((size_t)L * sizeof(int32_t)));
  HOK(hipMemcpyAsync(d_hour.p, in->hour, (size_t)L * sizeof(int32_t), hipMemcpyHostToDevice, stream));
  std::vector<double> sun, slat, clat, lrad;
  if (skyview) {
    sun.resize((size_t)L * RS_SUN_COLS);
    rs_sun_table(L, in->year, in->month, in->day, in->hour, in->minute, in->second, sun.data());
    HOK(d_sun.alloc(sun.size() * sizeof(double)));
    HOK(hipMemcpyAsync(d_sun.p, sun.data(), sun.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    /* (this worker's points only: index q of the three vectors is point pbeg + q) */
    slat.resize(pend - pbeg);
    clat.resize(pend - pbeg);
    lrad.resize(pend - pbeg);
    rs_point_geometry((int32_t)(pend - pbeg), local + pbeg, slat.data(), clat.data(), lrad.data());
  }

  PhaseTimer pt(stream);
  pt.lap(0);
  WindowLease win;
  win.stream = stream;
  const int nwin = skyview ? NFLD : NFLD - 2;
  size_t win_bytes = 0;
  {
    const int64_t Ppad = ((int64_t)P + RS_BLOCK - 1) / RS_BLOCK * RS_BLOCK;
    /* chunked coupling: the replay block spans a coupling window plus the index behind it
     * (usually more rows than a chunk); a tile whose windows are spread further re-leases below */
    const int rows0 = cpl_chunked ? std::max(TC, std::min(L, c.cplLen + 2)) : TC;
    /* (a block that steps from the raw series needs windows for the replay rounds of coupling only: sized
     * per tile, below) */
    win_bytes = use_raw ? 0 : (size_t)nwin * Ppad * rows0 * sizeof(double);
    if (win_bytes) HOK(win.acquire(win_bytes, device));
  }
  pt.lap(6);
  /* Budget of one worker's forcing windows.  Chunked coupling sizes its replay block from the
   * tile's couplingIndexI values (known only after read_input has run on the device): one station
   * that stopped reporting hours before the others stretches the block towards SimLen, and at the
   * default tile that is > 100 GB per worker.  Such a tile is cut in halves until it fits. */
  const char *eb = getenv("ROADSURF_HIP_WINDOW_BUDGET_MB");
  const size_t win_budget = eb ? (size_t)std::max(1, atoi(eb)) << 20 : (size_t)24 << 30;
  int Pcur = P;
  g_last_tiles = 0;
  g_last_raw_launches = 0;
  const size_t arena_mark = arena.off; /* the shared axes stay; a tile's buffers go when it is done */
  for (int64_t p0 = pbeg, m_done = 0; p0 < pend; p0 += m_done) {
    m_done = 0; /* a tile that has to be cut is started again at the same p0 */
    arena.rewind(arena_mark); /* the last tile's buffers are gone (same stream: what still runs there runs first) */
    const int m = (int)std::min<int64_t>(Pcur, pend - p0);
    PlanGuard pg;
    pg.p = rs_hip_plan_create(device, m, &consts, stream);
    if (!pg.p) return -11;
    const int64_t mp = rs_hip_plan_npoints_padded(pg.p);
    TileRaw T;
    TileLanding landing;
    Dev d_hzpt;
    {
      const double tg0 = PhaseTimer::now();
      std::lock_guard<std::mutex> turn(rsu::copy_gate(device)); /* rs_devices.hpp: uploads take turns */
      const double tg1 = PhaseTimer::now();
      /* default: copies and transposes one after the other on the worker's stream */
      const bool inline_upload = !upload_stream;
      landing.async = !inline_upload;
      hipStream_t cs = inline_upload ? stream : copy_stream;
      if (!inline_upload) { /* the landing block comes out of the arena the last tile's kernels may still be reading */
        HOK(hipEventCreateWithFlags(&landing.extra[0], hipEventDisableTiming));
        HOK(hipEventRecord(landing.extra[0], stream));
        HOK(hipStreamWaitEvent(copy_stream, landing.extra[0], 0));
      }
      if (int rc = upload_copies(in, c, p0, m, mp, T, landing, cs, inline_upload ? nullptr : stream)) return rc;
      /* the tile's local horizons (2.9 KB per point: as many bytes as all the series together) in the same
       * turn on the link, so that the block's first launch waits for ITS bytes only - enqueued later, the
       * copy shared the link with the next block's series and the first step started 58 ms into the call */
      if (skyview && in->horizons) {
        HOK(d_hzpt.alloc((size_t)m * 360 * sizeof(double)));
        HOK(hipMemcpyAsync(d_hzpt.p, in->horizons + (size_t)p0 * 360, (size_t)m * 360 * sizeof(double),
                           hipMemcpyHostToDevice, cs));
      }
      if (inline_upload) HOK(hipStreamSynchronize(stream));
      /* the worker's own stream starts on the pieces as they land (upload_finish); the turn on the link
       * ends when the last byte has */
      if (int rc = upload_finish(in, c, p0, m, mp, T, landing, stream)) return rc;
      const double tg2 = PhaseTimer::now();
      HOK(hipStreamSynchronize(cs));
      if (!inline_upload && skyview && in->horizons) { /* (the copy is done: the event orders the worker's stream behind it) */
        HOK(hipEventCreateWithFlags(&landing.extra[1], hipEventDisableTiming));
        HOK(hipEventRecord(landing.extra[1], copy_stream));
        HOK(hipStreamWaitEvent(stream, landing.extra[1], 0));
      }
      if (pt.on)
        fprintf(stderr, "rs_driver_run upload: waited %.1f ms for the link, issued the copies in %.1f ms, "
                        "drained in %.1f ms\n", 1e3 * (tg1 - tg0), 1e3 * (tg2 - tg1), 1e3 * (PhaseTimer::now() - tg2));
    }
    pt.lap(1);
    TileDecisions D;
    if (int rc = decide_tile(c, st, T, D, stream)) return rc;
    TileReport rep;
    if (int rc = report_issue(D, p0, m, rep, stream)) return rc;
    /* chunked coupling sizes its replay block from the decisions (below): it needs them now */
    if (cpl_chunked || pt.on)
      if (int rc = report_finish(c, st, rep, local, out->status, out->missing_index)) return rc;
    pt.lap(2);

    /* per-point parameters */
    Dev d_tb, d_geo;
    HOK(d_tb.alloc(mp * sizeof(double)));
    hipLaunchKernelGGL(fill_f64_kernel, grid1(mp), dim3(RS_BLOCK), 0, stream, d_tb.as<double>(), mp,
                       tbottom);
    HOK(hipGetLastError());
    RsPointParams pp;
    std::memset(&pp, 0, sizeof(pp));
    pp.tbottom = d_tb.as<double>();
    pp.initlen = D.initlen.as<int32_t>();
    if (st->use_relaxation == 1) {
      pp.tair_relax = D.tair_relax.as<double>();
      pp.vz_relax = D.vz_relax.as<double>();
      pp.rh_relax = D.rh_relax.as<double>();
    }
    if (coupled) {
      pp.coupling_index = D.cpl_index.as<int32_t>();
      pp.coupling_tsurf = D.cpl_tsurf.as<double>();
    }
    if (skyview) {
      HOK(d_geo.alloc((size_t)4 * mp * sizeof(double)));
      std::vector<double> g((size_t)4 * mp, 1.0);
      for (int p = 0; p < m; ++p) {
        g[p] = local[p0 + p].sky_view;
        g[(size_t)mp + p] = slat[p0 - pbeg + p];
        g[(size_t)2 * mp + p] = clat[p0 - pbeg + p];
        g[(size_t)3 * mp + p] = lrad[p0 - pbeg + p];
      }
      HOK(hipMemcpyAsync(d_geo.p, g.data(), g.size() * sizeof(double), hipMemcpyHostToDevice, stream));
      HOK(hipStreamSynchronize(stream)); /* g goes out of scope */
      pp.sky_view = d_geo.as<double>();
      pp.sin_lat = d_geo.as<double>() + mp;
      pp.cos_lat = d_geo.as<double>() + 2 * mp;
      pp.lon_rad = d_geo.as<double>() + 3 * mp;
      pp.albedo_surroundings = params->Albedo_surroundings;
      /* the horizon table as the caller holds it, [point][360] (RsPointParams::horizons_by_point): no
       * transpose, half the device memory, and a point's neighbouring degrees in one cache line; no table
       * at all where the caller has none (the kernels read a missing table as 0) */
      pp.horizons = nullptr;
      pp.horizons_by_point = 1;
      if (in->horizons) pp.horizons = d_hzpt.as<double>(); /* (uploaded with the series, above) */
    }

    /* chunked coupling: where the tile's coupling windows lie (the decisions are back in `local`) */
    int cs_min = 0, ce_min = 0, ce_max = 0;
    bool any_on = false;
    if (cpl_chunked) {
      for (int p = 0; p < m; ++p) {
        const LocalParameters &lp = local[p0 + p];
        if (lp.couplingTsurf < -100 || lp.couplingIndexI < 1) continue; /* src/InputOutput.f90:34-36 */
        const int ce = lp.couplingIndexI;
        /* initCouplingTimes, src/Coupling.f90:512-517 */
        const int cs = ((double)ce <= (double)(st->coupling_minutes * 60) / st->DTSecs) ? 1 : ce - c.cplLen;
        if (!any_on) { cs_min = cs; ce_min = ce_max = ce; any_on = true; }
        cs_min = std::min(cs_min, cs); ce_min = std::min(ce_min, ce); ce_max = std::max(ce_max, ce);
      }
    }
    const int r_lo = cs_min, r_hi = std::min(ce_max + 1, L); /* replay block, 1-based inclusive */
    /* the replay rounds read the raw series too (rs_cpl_replay_raw) where the block ends before SimLen and
     * there is no sky view */
    /* ... and is COMPACT - not much longer than one coupling window, rs_hip_cpl_replay's own rule: stations whose
     * observations end hours apart (the reference's operational example) make a block in which a lock-step
     * replay would step every listed point through all of it, round after round - those keep the forcing window
     * and the per-lane replay kernel */
    const bool replay_raw = use_raw && cpl_chunked && any_on && !skyview && std::min(ce_max + 1, L) < L &&
                            (int64_t)(r_hi - r_lo + 1) * 4 <= ((int64_t)c.cplLen + 2) * 5;
    const bool need_win = !use_raw || (cpl_chunked && any_on && !replay_raw); /* raw-series stepping: windows for such replays only */
    const int WR = (cpl_chunked && any_on) ? (use_raw ? r_hi - r_lo + 1 : std::max(TC, r_hi - r_lo + 1)) : TC;
    if (cpl_chunked && WR > TC && (size_t)nwin * mp * WR * sizeof(double) > win_budget && m > 4096) {
      Pcur = std::max(4096, (m / 2 + 4095) / 4096 * 4096);
      continue;
    }
    if (need_win && (size_t)nwin * mp * WR * sizeof(double) > win_bytes) {
      win.release();
      win_bytes = (size_t)nwin * mp * WR * sizeof(double);
      HOK(win.acquire(win_bytes, device));
    }

    /* windows */
    const size_t fs = (size_t)mp * WR;
    Dev d_phase, d_out, d_outpt;
    if (need_win) {
      HOK(d_phase.alloc(fs * sizeof(int32_t)));
      hipLaunchKernelGGL(fill_i32_kernel, grid1((int64_t)fs), dim3(RS_BLOCK), 0, stream,
                         d_phase.as<int32_t>(), (int64_t)fs, -9999); /* InputData.cpp:16 */
      HOK(hipGetLastError());
    }
    const size_t os = (size_t)mp * n_out;
    HOK(d_out.alloc((size_t)6 * os * sizeof(double)));
    /* OutputData.cpp:5-13: rows the simulation never saves read -9999.0 */
    hipLaunchKernelGGL(fill_f64_kernel, grid1((int64_t)(6 * os)), dim3(RS_BLOCK), 0, stream,
                       d_out.as<double>(), (int64_t)(6 * os), -9999.0);
    HOK(hipGetLastError());
    HOK(d_outpt.alloc((size_t)m * n_out * sizeof(double)));

    ExpandRawArgs ea;
    ea.S = T.S;
    /* window f lives at slot wslot[f] of the leased block (SW_dir / LW_net only with sky view) */
    double *wb = static_cast<double *>(win.p);
    for (int f = 0, k = 0; f < NFLD; ++f) {
      const bool used = need_win && (skyview || (f != R_SWDIR && f != R_LWNET));
      ea.out[f] = used ? wb + (size_t)(k++) * fs : nullptr;
    }
    ea.status = D.status.as<int32_t>();
    ea.cpl_hi = coupled ? D.cpl_hi.as<int32_t>() : nullptr;
    ea.cplLen = c.cplLen;
    ea.stride = mp;
    ea.order = nullptr;

    RsOutputs oo;
    double *ob = d_out.as<double>();
    oo.tsurf = ob; oo.snow = ob + os; oo.water = ob + 2 * os; oo.ice = ob + 3 * os;
    oo.deposit = ob + 4 * os; oo.ice2 = ob + 5 * os;
    oo.t_stride = mp;
    oo.decimate = step;
    oo.row0 = 0;

    /* Plan order (rs_hip_recluster): with more than one launch per tile the slots
     * are re-sorted by regime after every launch; windows and per-point parameters are then
     * produced in slot order and each launch's output rows are mapped back.  Time-chunked
     * coupling runs that way too: the lock-step chunks are re-sorted like the uncoupled launches,
     * and the coupling kernels' outputs - the replays' included - go straight to their point's
     * column (rs_hip_set_output_by_point).  Measured at 1 M points, four plans: 0.90 s against
     * 0.96 s in natural order with the history key (-12 % vector instructions in the lock-step
     * kernel), see DESIGN.md 6 for the forecast key.  ROADSURF_HIP_CLUSTER=0 switches the order
     * off.  Sky view (round 4): the four geometry scalars are gathered like the other per-point
     * parameters, the local-horizon table is read through the order row. */
    const bool cluster = (!coupled || cpl_chunked) && TC < L && want_cluster;
    const int rows_c = TC / step + 2; /* output rows one launch can produce */
    Dev d_outc, d_pp_s, d_geo_s;
    RsOutputs oc = oo;
    RsPointParams pps = pp;
    if (cluster && skyview) { /* the four geometry scalars in slot order (padding: sky view 1.0 = off) */
      HOK(d_geo_s.alloc((size_t)4 * mp * sizeof(double)));
      hipLaunchKernelGGL(fill_f64_kernel, grid1((int64_t)(4 * mp)), dim3(RS_BLOCK), 0, stream,
                         d_geo_s.as<double>(), (int64_t)(4 * mp), 1.0);
      HOK(hipGetLastError());
      pps.sky_view = d_geo_s.as<double>();
      pps.sin_lat = d_geo_s.as<double>() + mp;
      pps.cos_lat = d_geo_s.as<double>() + 2 * mp;
      pps.lon_rad = d_geo_s.as<double>() + 3 * mp;
    }
    if (cluster) {
      if (!use_raw) { /* (the raw-series step kernel writes its rows straight into point order) */
        HOK(d_outc.alloc((size_t)6 * rows_c * mp * sizeof(double)));
        double *cb = d_outc.as<double>();
        const size_t cs = (size_t)rows_c * mp;
        oc.tsurf = cb; oc.snow = cb + cs; oc.water = cb + 2 * cs; oc.ice = cb + 3 * cs;
        oc.deposit = cb + 4 * cs; oc.ice2 = cb + 5 * cs;
      }
      /* slot-order copies: 4 doubles (3 relaxation targets, coupling observation), 2 int32 */
      HOK(d_pp_s.alloc((size_t)mp * (2 * sizeof(int32_t) + 4 * sizeof(double))));
      double *pd = d_pp_s.as<double>();
      pps.tair_relax = st->use_relaxation == 1 ? pd : nullptr;
      pps.vz_relax = st->use_relaxation == 1 ? pd + mp : nullptr;
      pps.rh_relax = st->use_relaxation == 1 ? pd + 2 * mp : nullptr;
      pps.initlen = reinterpret_cast<int32_t *>(pd + 4 * mp);
      if (coupled) {
        pps.coupling_tsurf = pd + 3 * mp;
        pps.coupling_index = reinterpret_cast<int32_t *>(pd + 4 * mp) + mp;
      }
      HOK(hipMemsetAsync(d_pp_s.p, 0, (size_t)mp * (2 * sizeof(int32_t) + 4 * sizeof(double)), stream));
    }
    /* per-point parameters into the plan's current slot order */
    auto gather_params = [&]() -> int {
      ea.order = rs_hip_plan_order(pg.p); /* identity until the first recluster */
      if (!ea.order) return -14;
      double *pd = d_pp_s.as<double>();
      hipLaunchKernelGGL(gather_params_kernel, grid1(m), dim3(RS_BLOCK), 0, stream, ea.order, (int64_t)m,
                         pp.initlen, const_cast<int32_t *>(pps.initlen), D.tair_relax.as<double>(), pd,
                         D.vz_relax.as<double>(), pd + mp, D.rh_relax.as<double>(), pd + 2 * mp,
                         coupled ? pp.coupling_index : nullptr,
                         coupled ? const_cast<int32_t *>(pps.coupling_index) : nullptr,
                         coupled ? pp.coupling_tsurf : nullptr,
                         coupled ? const_cast<double *>(pps.coupling_tsurf) : nullptr,
                         skyview ? pp.sky_view : nullptr, skyview ? d_geo_s.as<double>() : nullptr, (int64_t)mp);
      HOK(hipGetLastError());
      if (skyview) pps.horizon_index = ea.order; /* slot -> column of the horizon table */
      return 0;
    };
    if (cluster)
      if (int rc = gather_params()) return rc;
    pt.lap(3);
    /* The sources that can supply a value somewhere in [i0, i0+n): a shared-axis source whose plan is
     * K_NONE over the whole range - the observations behind their last report, i.e. for seven windows
     * out of eight of a 48 h run - is left out of the launch (it would cost a plan load and a branch
     * per index and variable for nothing).  Order kept: later sources win. */
    const SrcSet S_full = T.S;
    auto sources_for = [&](int i0, int n) -> SrcSet {
      SrcSet r = S_full;
      if (T.any_pp) return r;
      r.nsrc = 0;
      for (int k = 0; k < S_full.nsrc; ++k) {
        const std::vector<int32_t> &act = c.active_prefix[k];
        if (act[std::min(i0 + n, L)] - act[std::min(i0, L)] > 0) r.src[r.nsrc++] = S_full.src[k];
      }
      for (int k = r.nsrc; k < RS_MAX_SOURCES; ++k) r.src[k] = SrcDev{};
      return r;
    };
    /* one window [t0, t0+len): raw series -> step-resolution forcing on the device */
    int walk_at = 0; /* 0-based index the per-point raw walks are positioned at */
    auto expand_window = [&](int t0, int len, RsForcing &fo) -> int {
      if (T.any_pp && walk_at != t0 - 1) { /* not the continuation of the last window: re-position */
        hipLaunchKernelGGL(pp_init_kernel, grid1(mp), dim3(RS_BLOCK), 0, stream, T.S);
        HOK(hipGetLastError());
        if (t0 > 1) {
          hipLaunchKernelGGL(pp_advance_kernel, grid1(mp), dim3(RS_BLOCK), 0, stream, T.S, (int32_t)0,
                             (int32_t)(t0 - 1));
          HOK(hipGetLastError());
        }
        walk_at = t0 - 1;
      }
      ea.i0 = t0 - 1;
      ea.nsteps = len;
      ea.S = sources_for(t0 - 1, len);
      launch_expand_raw(T.any_pp, mp, ea, stream);
      HOK(hipGetLastError());
      if (T.any_pp && t0 + len <= L) { /* per-point walks: move to the start of the next window */
        hipLaunchKernelGGL(pp_advance_kernel, grid1(mp), dim3(RS_BLOCK), 0, stream, T.S,
                           (int32_t)(t0 - 1), (int32_t)len);
        HOK(hipGetLastError());
        walk_at = t0 - 1 + len;
      }
      std::memset(&fo, 0, sizeof(fo));
      fo.tair = ea.out[R_TAIR]; fo.tdew = ea.out[R_TDEW]; fo.vz = ea.out[R_VZ];
      fo.rhz = ea.out[R_RHZ]; fo.prec = ea.out[R_PREC]; fo.sw = ea.out[R_SW]; fo.lw = ea.out[R_LW];
      fo.tsurfobs = ea.out[R_OBS];
      fo.depth = nullptr; /* InputData.cpp:18: Depth is never filled by the driver */
      fo.precphase = d_phase.as<int32_t>();
      fo.hour = d_hour.as<int32_t>() + (t0 - 1);
      fo.t_stride = mp;
      fo.hour_pstride = 0;
      if (skyview) {
        fo.sw_dir = ea.out[R_SWDIR];
        fo.lw_net = ea.out[R_LWNET];
        fo.sun = d_sun.as<double>() + (size_t)(t0 - 1) * RS_SUN_COLS;
      }
      return 0;
    };
    /* Re-sort of the slots for the window [t_next, t_next+len_next): by a FORECAST of that window
     * (rs_hip_recluster_forecast) - air temperature and wind speed at three of its
     * indices, produced from the raw series in the CURRENT slot order by the expansion kernel
     * itself (only those two fields, one index each) - or, where the raw series have per-point time
     * axes (their walks would have to be re-positioned for every preview), by the history of the
     * last launch. */
    Dev d_prev;
    const bool forecast_key = !T.any_pp;
    /* nobody reads the step kernels' history score then: run the instances without it */
    if (rs_hip_set_history_score(pg.p, (cluster && forecast_key && !cpl_chunked) ? 0 : 1) != 0) return -14;
    auto resort_for = [&](int t_next, int len_next) -> int {
      if (!forecast_key) {
        if (rs_hip_recluster(pg.p) != 0) return -14;
        return gather_params();
      }
      if (!d_prev.p) HOK(d_prev.alloc((size_t)9 * mp * sizeof(double)));
      const int idx[3] = {t_next, t_next + len_next / 2, t_next + len_next - 1};
      ExpandRawArgs pe = ea; /* current order, same sources and decisions */
      RsPreview pv;
      std::memset(&pv, 0, sizeof(pv));
      pv.n = 3;
      if (use_raw) { /* the six rows in one launch */
        RawRowsArgs ra;
        std::memset(&ra, 0, sizeof(ra));
        ra.S = S_full;
        ra.status = ea.status;
        ra.order = ea.order;
        /* ... nine with the precipitation of the three indices (RsPreview::prec: the key's precipitation bit) */
        constexpr bool wet_bit = true; /* (+1.7 % / +4 %: profiles/r05_ab_precip_bit.txt; a constant since round 6) */
        ra.nrows = wet_bit ? 9 : 6;
        for (int q = 0; q < 3; ++q) {
          ra.fld[2 * q] = R_TAIR;
          ra.fld[2 * q + 1] = R_VZ;
          ra.idx[2 * q] = ra.idx[2 * q + 1] = idx[q] - 1;
          ra.out[2 * q] = d_prev.as<double>() + (size_t)(2 * q) * mp;
          ra.out[2 * q + 1] = d_prev.as<double>() + (size_t)(2 * q + 1) * mp;
          pv.tair[q] = ra.out[2 * q];
          pv.vz[q] = ra.out[2 * q + 1];
          pv.hour[q] = in->hour[idx[q] - 1];
          if (wet_bit) {
            ra.fld[6 + q] = R_PREC;
            ra.idx[6 + q] = idx[q] - 1;
            ra.out[6 + q] = d_prev.as<double>() + (size_t)(6 + q) * mp;
            pv.prec[q] = ra.out[6 + q];
          }
        }
        hipLaunchKernelGGL(raw_rows_kernel, dim3((unsigned)(mp / RS_BLOCK), (unsigned)ra.nrows), dim3(RS_BLOCK), 0, stream, ra);
        HOK(hipGetLastError());
      } else
      for (int q = 0; q < 3; ++q) {
        for (int f = 0; f < NFLD; ++f) pe.out[f] = nullptr;
        pe.out[R_TAIR] = d_prev.as<double>() + (size_t)(2 * q) * mp;
        pe.out[R_VZ] = d_prev.as<double>() + (size_t)(2 * q + 1) * mp;
        pe.i0 = idx[q] - 1;
        pe.nsteps = 1;
        pe.S = sources_for(idx[q] - 1, 1);
        launch_expand_raw(false, mp, pe, stream);
        HOK(hipGetLastError());
        pv.tair[q] = pe.out[R_TAIR];
        pv.vz[q] = pe.out[R_VZ];
        pv.hour[q] = in->hour[idx[q] - 1];
      }
      /* the boundary-layer regime at the window's first and last index; the middle one makes the key's count
       * fields finer and the order no better (relaxation 0.346 -> 0.343 s without it, as bench.py's FULL leg with
       * its windows of whole hours: profiles/r05_ab_driver_previews.txt).
       * Its precipitation row stays: the key's precipitation bit reads every row it is given. */
      {
        pv.n = 2;
        pv.tair[1] = pv.tair[2];
        pv.vz[1] = pv.vz[2];
        pv.hour[1] = pv.hour[2];
      }
      pv.tair_now = pv.tair[0];
      pv.alpha = 0.5;
      pv.mode = 378059; /* 10 bits + the ground digit: the plan's own counting sort (rs_cluster.hip) */
      if (rs_hip_recluster_forecast(pg.p, &pv) != 0) return -14;
      return gather_params();
    };
    if (cpl_chunked) {
      /* stage 1: lock step to the last window end; stage 2: the replay rounds over the block
       * [first window start, last window end + 1]; stage 3: lock step from behind the first
       * window end (points whose window ends later wait there: they step only the index they
       * are due for) */
      RsForcing fo;
      /* plan order: the slots are re-sorted after every lock-step chunk; windows and per-point
       * parameters are produced in slot order, outputs go to their point's column */
      const RsPointParams &ppx = cluster ? pps : pp;
      if (cluster && rs_hip_set_output_by_point(pg.p, 1) != 0) return -14;
      auto resort = [&](int t_next) -> int {
        if (!cluster) return 0;
        if (forecast_key) return resort_for(t_next, std::min(TC, L - t_next + 1));
        if (rs_hip_recluster(pg.p) != 0) return -14;
        return gather_params();
      };
      /* a lock-step chunk: from the raw series (rs_step_raw: no window) where the block steps that way */
      rs::RawForcing rf;
      std::memset(&rf, 0, sizeof(rf));
      Dev d_row1;
      size_t seg = 0;
      if (use_raw) {
        rf.nsrc = S_full.nsrc;
        for (int k = 0; k < S_full.nsrc; ++k) {
          for (int f = 0; f < NFLD; ++f) rf.src[k].fld[f] = S_full.src[k].fld[f];
          rf.src[k].plan = S_full.src[k].plan;
        }
        rf.nseg = (int32_t)c.segs.size();
        rf.segs = T.segs.as<ScanSeg>();
        rf.np_pad = mp;
        rf.status = D.status.as<int32_t>();
        rf.hour = d_hour.as<int32_t>();
        HOK(d_row1.alloc((size_t)2 * mp * sizeof(double)));
      }
      auto lockstep = [&](int t0, int len) -> int {
        if (!use_raw) {
          if (int rc = expand_window(t0, len, fo)) return rc;
          if (t0 == 1 && rs_hip_init_state(pg.p, &fo, &ppx) != 0) return -12;
          if (rs_hip_step_cpl(pg.p, &fo, &oo, &ppx, t0, len) != 0) return -13;
          return 0;
        }
        if (t0 == 1) { /* index 1's air temperature and observation, for the initial profile */
          RawRowsArgs ra;
          std::memset(&ra, 0, sizeof(ra));
          ra.S = S_full;
          ra.status = ea.status;
          ra.order = cluster ? ea.order : nullptr;
          ra.nrows = 2;
          ra.fld[0] = R_TAIR;
          ra.fld[1] = R_OBS;
          ra.out[0] = d_row1.as<double>();
          ra.out[1] = d_row1.as<double>() + mp;
          hipLaunchKernelGGL(raw_rows_kernel, dim3((unsigned)(mp / RS_BLOCK), 2), dim3(RS_BLOCK), 0, stream, ra);
          HOK(hipGetLastError());
          RsForcing f1;
          std::memset(&f1, 0, sizeof(f1));
          f1.tair = f1.vz = f1.rhz = f1.prec = f1.sw = f1.lw = ra.out[0]; /* (only tair and tsurfobs are read) */
          f1.tsurfobs = ra.out[1];
          f1.precphase = reinterpret_cast<const int32_t *>(ra.out[0]);
          f1.hour = d_hour.as<int32_t>();
          f1.t_stride = mp;
          if (rs_hip_init_state(pg.p, &f1, &ppx) != 0) return -12;
        }
        while (seg > 0 && c.segs[seg].i0 > t0 - 1) --seg; /* (stage 3 starts behind the first window end: back) */
        while (seg + 1 < c.segs.size() && c.segs[seg].i1 <= t0 - 1) ++seg;
        rf.seg0 = (int32_t)seg;
        rf.col = cluster ? ea.order : nullptr;
        const double *sunrows = skyview ? d_sun.as<double>() + (size_t)(t0 - 1) * RS_SUN_COLS : nullptr;
        if (rs_step_raw(pg.p, &rf, sunrows, &oo, &ppx, t0, len, cluster) != 0) return -13;
        ++g_last_raw_launches;
        return 0;
      };
      const int s1_hi = any_on ? std::min(ce_max, L) : L;
      for (int t0 = 1; t0 <= s1_hi; t0 += TC) {
        const int len = std::min(TC, s1_hi - t0 + 1);
        if (int rc = lockstep(t0, len)) return rc;
        /* the next lock-step chunk: the one behind this, or stage 3's first */
        const int t_next = (t0 + len <= s1_hi) ? t0 + len : (any_on && ce_min + 1 <= L) ? ce_min + 1 : 0;
        if (t_next > 0)
          if (int rc = resort(t_next)) return rc;
      }
      if (any_on) {
        int32_t rounds = 0;
        if (replay_raw) {
          while (seg > 0 && c.segs[seg].i0 > r_lo - 1) --seg;
          while (seg + 1 < c.segs.size() && c.segs[seg].i1 <= r_lo - 1) ++seg;
          rf.seg0 = (int32_t)seg;
          rf.col = cluster ? ea.order : nullptr;
          if (rs_cpl_replay_raw(pg.p, &rf, &oo, &ppx, r_lo, r_hi - r_lo + 1, cluster, &rounds) != 0) return -13;
        } else {
          if (int rc = expand_window(r_lo, r_hi - r_lo + 1, fo)) return rc;
          if (rs_hip_cpl_replay(pg.p, &fo, &oo, &ppx, r_lo, r_hi - r_lo + 1, &rounds) != 0) return -13;
        }
        /* every window is behind the plan: stage 3's re-sorts need not move the saved state */
        if (rs_hip_coupling_windows_closed(pg.p, 1) != 0) return -14;
        for (int t0 = ce_min + 1; t0 <= L; t0 += TC) {
          const int len = std::min(TC, L - t0 + 1);
          if (int rc = lockstep(t0, len)) return rc;
          if (t0 + len <= L)
            if (int rc = resort(t0 + len)) return rc;
        }
      }
    } else if (use_raw) {
      /* no windows: the step kernel's ground wave reads the raw series (rs_step_raw) */
      rs::RawForcing rf;
      std::memset(&rf, 0, sizeof(rf));
      rf.nsrc = S_full.nsrc;
      for (int k = 0; k < S_full.nsrc; ++k) {
        for (int f = 0; f < NFLD; ++f) rf.src[k].fld[f] = S_full.src[k].fld[f];
        rf.src[k].plan = S_full.src[k].plan;
      }
      rf.nseg = (int32_t)c.segs.size();
      rf.segs = T.segs.as<ScanSeg>();
      rf.np_pad = mp;
      rf.status = D.status.as<int32_t>();
      rf.hour = d_hour.as<int32_t>();
      Dev d_row1;
      HOK(d_row1.alloc((size_t)2 * mp * sizeof(double)));
      { /* index 1's air temperature and observation, for the initial profile */
        RawRowsArgs ra;
        std::memset(&ra, 0, sizeof(ra));
        ra.S = S_full;
        ra.status = ea.status;
        ra.order = cluster ? ea.order : nullptr;
        ra.nrows = 2;
        ra.fld[0] = R_TAIR;
        ra.fld[1] = R_OBS;
        ra.out[0] = d_row1.as<double>();
        ra.out[1] = d_row1.as<double>() + mp;
        hipLaunchKernelGGL(raw_rows_kernel, dim3((unsigned)(mp / RS_BLOCK), 2), dim3(RS_BLOCK), 0, stream, ra);
        HOK(hipGetLastError());
        RsForcing f1;
        std::memset(&f1, 0, sizeof(f1));
        f1.tair = f1.vz = f1.rhz = f1.prec = f1.sw = f1.lw = ra.out[0]; /* (only tair and tsurfobs are read) */
        f1.tsurfobs = ra.out[1];
        f1.precphase = reinterpret_cast<const int32_t *>(ra.out[0]);
        f1.hour = d_hour.as<int32_t>();
        f1.t_stride = mp;
        if (rs_hip_init_state(pg.p, &f1, cluster ? &pps : &pp) != 0) return -12;
      }
      size_t seg = 0;
      for (int t0 = 1; t0 <= L; t0 += TC) {
        const int len = std::min(TC, L - t0 + 1);
        while (seg + 1 < c.segs.size() && c.segs[seg].i1 <= t0 - 1) ++seg;
        rf.seg0 = (int32_t)seg;
        rf.col = cluster ? ea.order : nullptr;
        const double *sunrows = skyview ? d_sun.as<double>() + (size_t)(t0 - 1) * RS_SUN_COLS : nullptr;
        /* the (decimated) rows of a launch go straight to their point's column of the result */
        if (rs_step_raw(pg.p, &rf, sunrows, &oo, cluster ? &pps : &pp, t0, len, cluster) != 0) return -13;
        ++g_last_raw_launches;
        if (!cluster) continue;
        if (t0 + len <= L)
          if (int rc = resort_for(t0 + len, std::min(TC, L - (t0 + len) + 1))) return rc;
      }
    } else
    for (int t0 = 1; t0 <= L; t0 += TC) {
      const int len = std::min(TC, L - t0 + 1);
      RsForcing fo;
      if (int rc = expand_window(t0, len, fo)) return rc;
      if (!cluster) {
        if (t0 == 1 && rs_hip_init_state(pg.p, &fo, &pp) != 0) return -12;
        if (rs_hip_step(pg.p, &fo, &oo, &pp, t0, len) != 0) return -13;
        continue;
      }
      /* this launch's rows go to the launch buffer in slot order, then home */
      const int64_t r_first = ((int64_t)t0 - 1 + step - 1) / step;
      const int64_t r_last = ((int64_t)t0 + len - 2) / step;
      oc.row0 = r_first;
      if (t0 == 1 && rs_hip_init_state(pg.p, &fo, &pps) != 0) return -12;
      if (rs_hip_step(pg.p, &fo, &oc, &pps, t0, len) != 0) return -13;
      if (r_last >= r_first) {
        hipLaunchKernelGGL(unpermute_rows_kernel, dim3((unsigned)(mp / RS_BLOCK), 6), dim3(RS_BLOCK), 0,
                           stream, ea.order, (int64_t)m, (const double *)d_outc.as<double>(),
                           (int64_t)rows_c, ob, (int64_t)n_out, r_first, (int32_t)(r_last - r_first + 1),
                           (int64_t)mp);
        HOK(hipGetLastError());
      }
      if (t0 + len <= L)
        if (int rc = resort_for(t0 + len, std::min(TC, L - (t0 + len) + 1))) return rc;
    }
    /* everything is enqueued: the decisions go to the caller's arrays while the device works */
    if (int rc = report_finish(c, st, rep, local, out->status, out->missing_index)) return rc;
    pt.lap(4);
    hipLaunchKernelGGL(blank_rejected_kernel, grid1(m), dim3(RS_BLOCK), 0, stream, ob, (int64_t)mp,
                       (int32_t)n_out, (int64_t)m, (const int32_t *)D.status.as<int32_t>());
    HOK(hipGetLastError());
    double *dst[6] = {out->tsurf, out->snow, out->water, out->ice, out->deposit, out->ice2};
    for (int f = 0; f < 6; ++f) {
      if (!dst[f]) continue;
      HOK(transpose((const double *)ob + (size_t)f * os, d_outpt.as<double>(), n_out, m, mp, n_out, stream));
      HOK(hipMemcpyAsync(dst[f] + (size_t)p0 * n_out, d_outpt.p, (size_t)m * n_out * sizeof(double),
                         hipMemcpyDeviceToHost, stream));
    }
    HOK(hipStreamSynchronize(stream));
    pt.lap(5);
    d_phase.release();
    d_outc.release();
    d_pp_s.release();
    d_geo_s.release();
    d_out.release();
    d_outpt.release();
    d_prev.release();
    pt.lap(7);
    m_done = m;
    ++g_last_tiles;
    Pcur = P; /* the next tile starts at full size again */
  }
  pt.report();
  return 0;
}
