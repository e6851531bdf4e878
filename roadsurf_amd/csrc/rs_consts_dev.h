/* rs_consts_dev.h - what a plan uploads as its constant block for the fp64 kernels: RsConstants
 * (include/roadsurf.h, built by the Fortran host) plus derived values the kernels would otherwise
 * recompute every time step: the reciprocals of the uniform denominators (rs_math.hpp, rs_div_u)
 * and one uniform product.  Filled in C++ by rs_hip_plan_create; not part of the C-ABI. */
#pragma once
#include <cstdlib>
#include "../../include/roadsurf.h"

struct RsConstantsDev : RsConstants {
  double meltDen;   /* WatMHeat*WatDens, the denominator of `Melted` (src/Storage.f90:149,230) */
  /* RN(1/x) of: 3600 (src/InputOutput.f90:111), 3364 (src/BoundaryLayer.f90:53), 1000
   * (src/Storage.f90:422,427), IceMax 1.5 (src/Cond.f90:131), twoDT (src/BalanceModel.f90:241),
   * DTSecs (src/Storage.f90:422), meltDen, logUstar and logCond (first pass of the boundary-layer
   * loop, where PSIM = PSIH = 0: src/BoundaryLayer.f90:62,69-70) */
  double r_3600, r_3364, r_1000, r_IceMax, r_twoDT, r_DTSecs, r_meltDen, r_logUstar, r_logCond;
  /* 1: a bare, dry road stays bare whatever the limits say (road_condition's fast path,
   * rs_physics_body.inc): no upper storage limit is negative */
  int32_t bareFastOk;
  /* 1: an index without precipitation leaves the storages alone (fluxes_pre's shortcut): MinPrecmm >= 0 */
  int32_t precFastOk;
  /* CheckValues' bounds (src/InputOutput.f90:45-84; REAL(4) literals): 100, -0.1, -90, 120, -1,
   * 4000, 1000, 500 - in the order check_values reads them */
  double chk[8];
  /* relax_tab[d] = exp(-(DTSecs*d)/14400), d = 0..SimLen: the relaxation factor
   * exp(-(DTSecs*i - DTSecs*InitLenI)/(4*3600)) of src/Relaxation.f90:34-37 for i - InitLenI = d.
   * Device pointer, or NULL where the argument is not a function of the difference alone (a time
   * step that is not an integer: DTSecs*i then rounds) or relaxation is off.  Filled by
   * rs_hip_plan_create with the host's exp - glibc's, whose bits rs_exp reproduces
   * (tests/test_hip_math.py), and IEEE division, which rs_div reproduces. */
  const double *relax_tab;
  /* cpl_tab[d] = exp(-(DTSecs*d)/couplingEffectReduction): the decay of the radiation corrections
   * behind a coupling window, exp(-(DTSecs*i - DTSecs*couplingEndI)/reduction) of
   * src/Coupling.f90:80-88, for i - couplingEndI = d.  Same conditions as relax_tab. */
  const double *cpl_tab;
  /* A frozen layer (Tmp < 0) has the constant heat capacity dryCap + WCont*(920*2100)
   * (src/BalanceModel.f90:215-236), so its -1/(DyC*VSH) (calcCapDZCondDZ, :132-155) and, for layer 1,
   * HS(1) = VSH*HSfac1/twoDT (:241) are constants of the plan: the same IEEE operations on the host.  A
   * wavefront whose 64 points all have the layer frozen reads them instead of evaluating the water
   * polynomials and a division (layer_step, rs_physics_body.inc). */
  double capDZF[RS_MAX_LAYERS + 2];
  double hs1F;
  /* the REAL(4) coefficients of the density and heat capacity of water (src/BalanceModel.f90:222-229) as
   * doubles, in the order layer_vsh uses them: as literals they cost the kernels two scalar moves each,
   * per thawed layer and step */
  double hcw[8];
  /* what layer_step reads of layer j, side by side: a frozen layer takes the first two with one scalar load,
   * a thawed one the last four (the same values as the tables of RsConstants and capDZF above) */
  struct LayerRow {
    double capDZF, condDZ, DyC, dryCap, WCont, pad[3];
  } lk[RS_MAX_LAYERS + 2];
};

static inline void rs_consts_dev_fill(const RsConstants &c, RsConstantsDev &d) {
  static_cast<RsConstants &>(d) = c;
  d.meltDen = c.WatMHeat * c.WatDens;
  d.r_3600 = 1.0 / 3600.0;
  d.r_3364 = 1.0 / 3364.0;
  d.r_1000 = 1.0 / 1000.0;
  d.r_IceMax = 1.0 / 1.5;
  d.r_twoDT = 1.0 / c.twoDT;
  d.r_DTSecs = 1.0 / c.DTSecs;
  d.r_meltDen = 1.0 / d.meltDen;
  d.r_logUstar = 1.0 / c.logUstar;
  d.r_logCond = 1.0 / c.logCond;
  const float chk[8] = {100.0f, -0.1f, -90.0f, 120.0f, -1.0f, 4000.0f, 1000.0f, 500.0f};
  for (int i = 0; i < 8; ++i) d.chk[i] = (double)chk[i];
  d.relax_tab = nullptr;
  d.cpl_tab = nullptr;
  for (int j = 0; j < RS_MAX_LAYERS + 2; ++j) {
    d.lk[j].capDZF = 0.0; /* filled below */
    d.lk[j].condDZ = c.condDZ[j];
    d.lk[j].DyC = c.DyC[j];
    d.lk[j].dryCap = c.dryCap[j];
    d.lk[j].WCont = c.WCont[j];
    d.lk[j].pad[0] = d.lk[j].pad[1] = d.lk[j].pad[2] = 0.0;
  }
  {
    const float h[8] = {-0.0050f, 0.0079f, 1000.0028f, 0.0000102f, 0.0017169f, 0.11516f, 3.4739f, 4217.2f};
    for (int i = 0; i < 8; ++i) d.hcw[i] = (double)h[i];
  }
  for (int j = 0; j < RS_MAX_LAYERS + 2; ++j) d.capDZF[j] = 0.0;
  d.hs1F = 0.0;
  for (int j = 1; j <= c.NLayers && j <= RS_MAX_LAYERS; ++j) {
    const volatile double chwt = 920.0 * 2100.0; /* REAL(4) literals, exact product */
    const volatile double wc = c.WCont[j] * chwt;  /* volatile: no contraction into an fma, whatever the flags */
    const volatile double vsh = c.dryCap[j] + wc;
    const volatile double den = c.DyC[j] * vsh;
    d.capDZF[j] = -(1.0 / den);
    d.lk[j].capDZF = d.capDZF[j];
    if (j == 1) {
      const volatile double num = vsh * c.HSfac1;
      d.hs1F = num / c.twoDT;
    }
  }
  d.bareFastOk = (c.MaxWatmms >= 0.0 && c.MaxSnowmms >= 0.0 && c.MaxIcemms >= 0.0 && c.MaxDepmms >= 0.0) ? 1 : 0;
  d.precFastOk = (c.MinPrecmm >= 0.0) ? 1 : 0;
}

/* Operand domain the bare division/sqrt sequences of rs_math.hpp rely on, as far as it is set by
 * the PARAMETERS (the forcing side is bounded by CheckValues; the data-dependent denominators of
 * the boundary-layer loop are covered by the guard in rs_physics_body.inc).  Returns NULL or the
 * name of the first offending quantity; rs_hip_plan_create refuses such a plan. */
static inline const char *rs_consts_domain_error(const RsConstants &c) {
  const double lo = 1e-30, hi = 1e30;
  auto pos = [&](double x) { return x > lo && x < hi; };          /* a positive denominator */
  auto mag = [&](double x) { return (x > lo && x < hi) || (x < -lo && x > -hi); }; /* nonzero, finite */
  if (!pos(c.DTSecs) || !pos(c.twoDT)) return "DTSecs";
  if (!mag(c.logUstar)) return "logUstar = log(ZRefW/ZMom)";
  if (!mag(c.logCond)) return "logCond = log(ZRefT/ZHeat)";
  if (!(c.logMom > -hi && c.logMom < hi) || !(c.logHeat > -hi && c.logHeat < hi)) return "logMom/logHeat";
  if (!pos(c.VK_Const)) return "VK_Const";
  if (!pos(c.LVap) || !pos(c.LFus)) return "LVap/LFus";
  if (!pos(c.WatMHeat * c.WatDens)) return "WatMHeat*WatDens";
  if (!pos(c.HSfac1)) return "layer grid (ZDpth(2)-ZDpth(1))";
  for (int j = 1; j <= c.NLayers; ++j) {
    if (!pos(c.DyC[j])) return "DyC (layer grid)";
    /* VSH = dryCap + WCont*CHWT with CHWT in [1.9e6, 4.3e6] (src/BalanceModel.f90:215-236) */
    if (!(c.dryCap[j] >= 0.0) || !(c.WCont[j] >= 0.0) || !pos(c.dryCap[j] + c.WCont[j] * 1.9e6))
      return "layer heat capacity (vsh/Poro/WCont)";
    if (!(c.ZDpth[j + 1] - c.ZDpth[j] > lo)) return "layer grid (ZDpth)";
  }
  return nullptr;
}

/* The relaxation table of RsConstantsDev::relax_tab on the host, or empty where it does not apply. */
#include <cmath>
#include <vector>
static inline std::vector<double> rs_decay_table(const RsConstants &c, bool on, double den) {
  std::vector<double> t;
  const double dt = c.DTSecs;
  if (!on || !(dt == std::floor(dt)) || !(dt * ((double)c.SimLen + 1.0) < 9.0e15)) return t;
  t.resize((size_t)c.SimLen + 1);
  for (int32_t d = 0; d <= c.SimLen; ++d) t[(size_t)d] = std::exp(-((dt * d) - (dt * 0)) / den);
  return t;
}
static inline std::vector<double> rs_relax_table(const RsConstants &c) {
  return rs_decay_table(c, c.use_relaxation != 0, (double)(4.f * 3600.f));
}
static inline std::vector<double> rs_cpl_table(const RsConstants &c) {
  return rs_decay_table(c, c.use_coupling != 0 && c.cplReduction > 1e-30 && c.cplReduction < 1e30,
                        c.cplReduction);
}
