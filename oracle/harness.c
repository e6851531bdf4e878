/*
 * oracle/harness.c — TEST INFRASTRUCTURE, not product code.
 *
 * OpenMP-over-points driver around a single-point `runsimulation` with the
 * reference signature (examples/example1/src/Simulation.f90:4-6).  The
 * reference itself has no OpenMP: its examples run one
 * `runsimulation` per work-queue thread (examples/example1/src/WorkQueue.h:16-129,
 * roadrunner.cpp:490-497).  This file is that loop, nothing else.
 *
 * It is compiled twice:
 *   - linked with the reference's own Fortran objects -> oracle/_ref/libroadsurf_ref.so
 *   - linked with oracle/roadsurf_oracle.c (our C restatement) -> oracle/liboracle.so
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * either library.
 *
 * Array layout here is the reference's: one contiguous [SimLen] array per
 * point and field, i.e. field[p*SimLen + t].
 */
#include "../include/roadsurf.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void runsimulation(OutputPointers *, const InputPointers *, const InputSettings *,
                   const InputParameters *, const LocalParameters *);

typedef struct HarnessArrays {
  /* inputs, each [n][SimLen] (mutated like the reference mutates them) */
  double *tair, *tdew, *vz, *rhz, *prec, *sw, *lw, *sw_dir, *lw_net, *tsurfobs,
      *depth;
  int32_t *precphase;
  /* shared time axis, each [SimLen] */
  int32_t *year, *month, *day, *hour, *minute, *second;
  /* optional [n][360], NULL -> zeros */
  double *local_horizons;
  /* outputs, each [n][SimLen] */
  double *tsurf, *snow, *water, *ice, *deposit, *ice2;
} HarnessArrays;

int harness_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* Run n points; nthreads <= 0 -> OpenMP default. Returns threads used. */
int harness_run_points(int32_t n, const HarnessArrays *a,
                       const InputSettings *settings,
                       const InputParameters *params,
                       const LocalParameters *local, int32_t nthreads) {
  const int64_t L = settings->SimLen;
  int used = 1;
  static double zero_horizons[360];
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
  {
#pragma omp single
    used = omp_get_num_threads();
#pragma omp for schedule(dynamic, 16)
#endif
    for (int32_t p = 0; p < n; ++p) {
      InputPointers in;
      OutputPointers out;
      const int64_t o = (int64_t)p * L;
      in.inputLen = (int32_t)L;
      in.c_tair = a->tair + o;
      in.c_tdew = a->tdew + o;
      in.c_VZ = a->vz + o;
      in.c_Rhz = a->rhz + o;
      in.c_prec = a->prec + o;
      in.c_SW = a->sw + o;
      in.c_LW = a->lw + o;
      in.c_SW_dir = a->sw_dir + o;
      in.c_LW_net = a->lw_net + o;
      in.c_TSurfObs = a->tsurfobs + o;
      in.c_PrecPhase = a->precphase + o;
      in.c_local_horizons =
          a->local_horizons ? a->local_horizons + (int64_t)p * 360 : zero_horizons;
      in.c_Depth = a->depth + o;
      in.c_year = a->year;
      in.c_month = a->month;
      in.c_day = a->day;
      in.c_hour = a->hour;
      in.c_minute = a->minute;
      in.c_second = a->second;
      out.outputLen = (int32_t)L;
      out.c_TsurfOut = a->tsurf + o;
      out.c_SnowOut = a->snow + o;
      out.c_WaterOut = a->water + o;
      out.c_IceOut = a->ice + o;
      out.c_DepositOut = a->deposit + o;
      out.c_Ice2Out = a->ice2 + o;
      runsimulation(&out, &in, settings, params, &local[p]);
    }
#ifdef _OPENMP
  }
#endif
  return used;
}
