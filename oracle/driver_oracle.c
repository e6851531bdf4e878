/*
 * oracle/driver_oracle.c — TEST INFRASTRUCTURE, not part of the product.
 *
 * CPU restatement of the reference driver's INPUT side for one batch of points, the
 * checker for layer 4 of include/roadsurf.h (rs_driver_run / rs_driver_expand):
 *
 *   CalcTdewOrRH                examples/example1/src/MeteorologyTools.cpp:12-51
 *   JsonSource::Impl ctor       examples/example1/src/JsonSource.cpp:182-316  (after parsing)
 *   JsonSource::Impl::interpolate                       JsonSource.cpp:49-176
 *   JsonSource::Impl::GetWeather / GetLatestObsIndex    JsonSource.cpp:323-373, 401-420
 *   DataHandler::GetWeather / GetLatestObsIndex         DataHandler.cpp:75-84, 118-137
 *   read_input                  examples/example1/src/roadrunner.cpp:156-278
 *
 * It is written the way the reference is written (a two-pointer walk over raw and
 * simulation times with the data handled inside the walk, one source after the other into
 * one InputData), NOT the way the device code is organised (time plan + per-variable
 * streaming), so the two are independent statements of the same behaviour.
 *
 * PINNING.  MeteorologyTools.cpp compiles on its own: oracle/build_ref.sh builds it with
 * g++ into oracle/_ref/libroadrunner_tools_ref.so and tests/test_driver_oracle.py holds
 * oracle_calc_tdew_or_rh to it bit for bit.  JsonSource.cpp and roadrunner.cpp need jsoncpp,
 * which this image does not have, and the reference ships no tests or fixtures for them:
 * for interpolate / GetWeather / read_input this restatement is PARITY UNPINNED (checked
 * only against hand-worked cases in tests/test_driver_oracle.py).
 * The reference builds its C++ with -funsafe-math-optimizations -freciprocal-math
 * (examples/example1/Makefile:6); like the Fortran side, parity here is defined against the
 * strict evaluation of the source expressions (no reassociation, IEEE division).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/roadsurf.h"

#define NF 10
enum { R_TAIR, R_TDEW, R_VZ, R_RHZ, R_PREC, R_SW, R_LW, R_SWDIR, R_LWNET, R_OBS };

/* MeteorologyTools.cpp:12-51 */
double oracle_calc_tdew_or_rh(double t, double tdew, double rh) {
  const double Alphaw = 17.269, Alphai = 21.875, Betaw = 237.3, Betai = 265.5, AFact = 0.61078;
  double Alpha = 0.0, Beta = 0.0;
  if (t >= 0.0) {
    Alpha = Alphaw;
    Beta = Betaw;
  } else {
    Alpha = Alphai;
    Beta = Betai;
  }
  double EsatT = AFact * exp(Alpha * t / (t + Beta));
  if (!isnan(tdew) && tdew > -1000) {
    double EsatTD = AFact * exp(Alpha * tdew / (tdew + Beta));
    double x = (EsatTD / EsatT) * 100.0;
    return (100.0 < x) ? 100.0 : x; /* std::min(x, 100.0) */
  }
  if (!isnan(rh) && rh > -1) {
    double Epr = 0.01 * rh * EsatT;
    double XX = log(Epr / AFact);
    return Beta * XX / (Alpha - XX);
  }
  return NAN;
}

/* InputData (examples/example1/src/InputData.cpp:5-26): the ten real series we carry */
typedef struct {
  int len;
  double *v[NF];
} Series;

static int series_init(Series *s, int len) {
  s->len = len;
  for (int f = 0; f < NF; ++f) {
    s->v[f] = (double *)malloc(sizeof(double) * (size_t)(len > 0 ? len : 1));
    if (!s->v[f]) return -1;
    for (int i = 0; i < len; ++i) s->v[f][i] = -9999.9;
  }
  return 0;
}
static void series_free(Series *s) {
  for (int f = 0; f < NF; ++f) free(s->v[f]);
}

static double thr_of(int f) { return f == R_LWNET ? -1000.0 : -100.0; }

/* JsonSource.cpp:49-176 */
static void interpolate(const Series *raw, Series *ip, const int64_t *rawtime, int rawLen,
                        const int64_t *simtime, int simLen) {
  int rawPos = 0, simPos = 0;
  if (rawtime[0] < simtime[0]) {
    for (rawPos = 0; rawPos < rawLen; ++rawPos) {
      if (rawtime[rawPos] >= simtime[0]) break;
    }
    rawPos = rawPos - 1;
    simPos = 0;
  } else if (simtime[0] < rawtime[0]) {
    for (simPos = 0; simPos < simLen; ++simPos) {
      if (simtime[simPos] >= rawtime[0]) break;
    }
    rawPos = 0;
  }
  while (rawPos + 1 < rawLen && simPos < simLen) {
    if (llabs(simtime[simPos] - rawtime[rawPos]) < 0.01) {
      for (int f = 0; f < NF; ++f)
        if (raw->v[f][rawPos] > thr_of(f)) ip->v[f][simPos] = raw->v[f][rawPos];
      simPos++;
    } else if (llabs(simtime[simPos] - rawtime[rawPos + 1]) < 0.01) {
      rawPos++;
    } else {
      for (int f = 0; f < NF; ++f)
        if (raw->v[f][rawPos] > thr_of(f) && raw->v[f][rawPos + 1] > thr_of(f))
          ip->v[f][simPos] = raw->v[f][rawPos] +
                             (simtime[simPos] - rawtime[rawPos]) *
                                 (raw->v[f][rawPos + 1] - raw->v[f][rawPos]) /
                                 (rawtime[rawPos + 1] - rawtime[rawPos]);
      simPos++;
    }
  }
}

static const double *raw_field(const RsRawSource *s, int f) {
  switch (f) {
    case R_TAIR: return s->tair;
    case R_TDEW: return s->tdew;
    case R_VZ: return s->vz;
    case R_RHZ: return s->rhz;
    case R_PREC: return s->prec;
    case R_SW: return s->sw;
    case R_LW: return s->lw;
    case R_SWDIR: return s->sw_dir;
    case R_LWNET: return s->lw_net;
    default: return s->tsurfobs;
  }
}

static int is_missing(double v) { return isnan(v) || v < -9000; }

/* One point: everything between the parsed sources and the call of runsimulation.
 * merged[f] -> [SimLen] of this point. */
static int read_input_point(int nsrc, const RsRawSource *src, int64_t p, const InputSettings *st,
                            int64_t start_time, int64_t forecast_time, const int64_t *simtime,
                            double *const merged[NF], LocalParameters *lp, int32_t *status,
                            int32_t *missing_index) {
  const int L = st->SimLen;
  /* roadrunner.cpp:166-169 */
  Series data;
  if (series_init(&data, L)) return -1;
  const int64_t init_secs = forecast_time - start_time;
  lp->InitLenI = 1 + (int)(init_secs / st->DTSecs);

  int maxIndex = -1; /* DataHandler.cpp:118-137 */
  for (int s = 0; s < nsrc; ++s) {
    /* JsonSource.cpp:226-311 for this point */
    Series ip, raw;
    if (series_init(&ip, L)) return -1;
    /* this point's series: its own "time" array or the source's shared one */
    const int width = src[s].n_times;
    const int dataLen = (src[s].times_per_point && src[s].lengths) ? src[s].lengths[p] : width;
    const int64_t *rawtime = src[s].times_per_point ? src[s].times + (size_t)p * width : src[s].times;
    if (dataLen > 0) {
      if (series_init(&raw, dataLen)) return -1;
      for (int f = 0; f < NF; ++f) {
        const double *h = raw_field(&src[s], f);
        if (h) memcpy(raw.v[f], h + (size_t)p * width, sizeof(double) * (size_t)dataLen);
      }
      for (int i = 0; i < dataLen; ++i) { /* JsonSource.cpp:288-295 */
        if (raw.v[R_TDEW][i] < -100 && raw.v[R_RHZ][i] > -100 && raw.v[R_TAIR][i] > -100)
          raw.v[R_TDEW][i] = oracle_calc_tdew_or_rh(raw.v[R_TAIR][i], -9999.9, raw.v[R_RHZ][i]);
        if (raw.v[R_RHZ][i] < -100 && raw.v[R_TDEW][i] > -100 && raw.v[R_TAIR][i] > -100)
          raw.v[R_RHZ][i] = oracle_calc_tdew_or_rh(raw.v[R_TAIR][i], raw.v[R_TDEW][i], -9999.9);
      }
      interpolate(&raw, &ip, rawtime, dataLen, simtime, L);
      series_free(&raw);
    }
    /* JsonSource::Impl::GetWeather, JsonSource.cpp:337-356 */
    for (int i = 0; i < L; ++i)
      for (int f = 0; f < NF; ++f)
        if (ip.v[f][i] > thr_of(f)) data.v[f][i] = ip.v[f][i];
    /* JsonSource::Impl::GetLatestObsIndex, JsonSource.cpp:401-420 */
    if (src[s].is_observation) {
      int tmp = -9999;
      for (int i = L; i > 0; i--) {
        if (ip.v[R_TAIR][i - 1] > -100) {
          tmp = i;
          break;
        }
      }
      if (tmp > -1) {
        if (maxIndex < 0 || (tmp > maxIndex)) maxIndex = tmp;
      }
    }
    series_free(&ip);
  }

  *status = 0;
  *missing_index = -1;
  /* roadrunner.cpp:182-231 */
  for (int i = 0; i < L && *status == 0; i++) {
    if (is_missing(data.v[R_TAIR][i])) *status = 1;
    else if (is_missing(data.v[R_RHZ][i])) *status = 2;
    else if (is_missing(data.v[R_PREC][i])) *status = 3;
    else if (is_missing(data.v[R_SW][i])) *status = 4;
    else if (is_missing(data.v[R_LW][i])) *status = 5;
    else if (is_missing(data.v[R_VZ][i])) *status = 6;
    if (*status) *missing_index = i;
  }
  if (*status == 0) {
    if (st->use_relaxation == 1) { /* roadrunner.cpp:235-250 */
      lp->tair_relax = -9999.9;
      lp->VZ_relax = -9999.9;
      lp->RH_relax = -9999.9;
      const int lastTairObsIndex = maxIndex;
      if (lastTairObsIndex > -1) {
        lp->InitLenI = lastTairObsIndex;
        if (lastTairObsIndex >= L) {
          *status = 7; /* the reference indexes its vectors with SimLen here */
        } else {
          lp->tair_relax = data.v[R_TAIR][lastTairObsIndex];
          lp->VZ_relax = data.v[R_VZ][lastTairObsIndex];
          lp->RH_relax = data.v[R_RHZ][lastTairObsIndex];
        }
      }
    }
    if (st->use_coupling == 1 && *status == 0) { /* roadrunner.cpp:253-275 */
      lp->couplingTsurf = -9999.9;
      lp->couplingIndexI = -9999;
      int i = L - 1;
      while (i >= 0 && (is_missing(data.v[R_OBS][i]) || data.v[R_OBS][i] < -100)) i = i - 1;
      const int cl = (int)(st->coupling_minutes * 60 / st->DTSecs);
      if (i >= cl) {
        lp->couplingTsurf = data.v[R_OBS][i];
        lp->couplingIndexI = i;
        const int couplingStartI = i - cl;
        for (int j = i; j > couplingStartI; j--) data.v[R_OBS][j] = -9999.9;
      }
    }
  }
  for (int f = 0; f < NF; ++f) memcpy(merged[f], data.v[f], sizeof(double) * (size_t)L);
  series_free(&data);
  return 0;
}

/* Batch form with the argument meaning of rs_driver_expand (include/roadsurf.h):
 * merged = [10][n_points][SimLen]. */
int oracle_driver_expand(const RsDriverInput *in, const InputSettings *st, LocalParameters *local,
                         double *merged, int32_t *status, int32_t *missing_index) {
  const int L = st->SimLen, n = in->n_points;
  const int DT = (int)st->DTSecs; /* JsonSource takes `const int DTSecs` */
  int64_t *simtime = (int64_t *)malloc(sizeof(int64_t) * (size_t)L);
  if (!simtime) return -1;
  int64_t t = in->start_time; /* JsonSource.cpp:199-205 */
  for (int i = 0; i < L; i++) {
    simtime[i] = t;
    t += DT;
  }
  int rc = 0;
#pragma omp parallel for schedule(dynamic, 16)
  for (int p = 0; p < n; ++p) {
    double *m[NF];
    for (int f = 0; f < NF; ++f) m[f] = merged + ((size_t)f * n + p) * L;
    if (read_input_point(in->n_sources, in->sources, p, st, in->start_time, in->forecast_time,
                         simtime, m, &local[p], &status[p], &missing_index[p]))
      rc = -1;
  }
  free(simtime);
  return rc;
}
