/*
 * rs_physics.hpp — one RoadSurf time step for one point, as device code.
 *
 * One point per lane.  The ground temperature profile is reached through a
 * small accessor (registers for the NLayers-specialised kernel, an LDS column
 * for the generic one); every other piece of carried state is a scalar in
 * VGPRs.  Layer constants come from RsConstants (kernel argument -> SGPRs).
 *
 * Arithmetic contract: fp64, IEEE + - * / sqrt in the
 * reference's evaluation order, no FMA contraction (the translation unit is
 * compiled with -ffp-contract=off), every unsuffixed Fortran literal is REAL(4)
 * and enters as R4(x) = (double)x##f.  exp/log are the only operations that are
 * not bit-defined by IEEE; see rs_math.hpp.
 *
 * The reference splits this work over ~20 subroutines that communicate through
 * derived types; here it is fused into one pass with the dead stores removed:
 *   - GCond, HS(2:), GroundFlux, SnowIceRat, WetSnowFrozen, Rain/SnowIntensity,
 *     PrecType, Tdew, SensibleHeatFlux are never read on this path;
 *   - condDZ is constant in time and comes from RsConstants;
 *   - VSH/capDZ/flux/update of a layer are one loop body with a rolling flux
 *     instead of three loops over heap/array temporaries
 *     (src/BalanceModel.f90:109 allocates Gflux every call);
 *   - atm%BLCond need not be carried: it only seeds BLCond_Old at j = 1 and the
 *     loop cannot exit before j = 5 (src/BoundaryLayer.f90:67,92).
 * Each block cites the reference lines it restates.
 */
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/roadsurf.h"
#include "rs_math.hpp"

#define R4(x) ((double)(x##f))

namespace rs {

/* Scalar carried state of one point. */
struct Scalars {
  double tnw1, tnw2; /* TmpNw(1:2) */
  double tsurf;      /* surf%TsurfAve */
  double wat, snow, ice, ice2, dep;
  double q2melt, t4melt, albedo;
  double tair_end, vz_end, rh_end; /* relaxation anchors */
  bool verycold, failed;
};

/* Forcing of one time index. */
struct Forcing {
  double tair, tdew, vz, rhz, prec, sw, lw, tsurfobs, depth;
  int32_t phase, hour;
};

/* Temperature at depth (src/BalanceModel.f90:390-417); T(k) = Tmp(k), k=1..N,
 * tbot = Tmp(N+1).  Selects instead of a data-dependent index so the register
 * profile never needs dynamic indexing. */
template <class Prof>
__device__ __forceinline__ double temp_at_depth(const RsConstants &c, const Prof &T, double tbot,
                                                double depth) {
  const int N = T.nlayers();
  if (fabs(depth - R4(0.0)) < R4(0.00001)) return T.get(1);
  if (depth > c.ZDpth[N + 1]) return tbot;
  double r = 0.0;
  bool found = false;
#pragma unroll
  for (int k = 1; k <= N; ++k) {
    const double zk = c.ZDpth[k], zk1 = c.ZDpth[k + 1];
    if (!found && depth > zk && depth <= zk1) {
      const double tk = T.get(k);
      const double tk1 = (k == N) ? tbot : T.get(k + 1);
      r = tk + rs_div((depth - zk) * (tk1 - tk), zk1 - zk);
      found = true;
    }
  }
  return r;
}

template <class Prof>
__device__ __forceinline__ double surface_temperature(const RsConstants &c, const Prof &T,
                                                      double tbot, double depth) {
  if (depth >= 0) return temp_at_depth(c, T, tbot, depth);
  return (T.get(1) + T.get(2)) / R4(2.0);
}

/* src/Cond.f90:143-249 + src/Storage.f90:9-29 */
__device__ __forceinline__ void precipitation_to_storage(const RsConstants &c, const MathTab &mt,
                                                         Scalars &s, int32_t phase,
                                                         double &prec_ts, double tair, double rhz) {
  double rain = R4(0.0), snow = R4(0.0);
  bool interpret = true;
  if ((double)phase > c.MissValI) {
    interpret = false;
    if (prec_ts <= c.MinPrecmm) {
      prec_ts = R4(0.0);
    } else {
      switch (phase) {
        case 0: case 1: case 4: case 5: rain = prec_ts; break;
        case 2: snow = prec_ts / R4(2.); rain = snow; break;
        case 3: case 6: snow = prec_ts; break;
        default: interpret = true;
      }
    }
  }
  if (interpret) {
    if (prec_ts <= c.MinPrecmm) {
      prec_ts = R4(0.0);
      rain = R4(0.0);
      snow = R4(0.0);
    } else {
      const double pexp = R4(22.0) - R4(2.7) * tair - R4(0.20) * rhz;
      const double prain = rs_div(R4(1.0), R4(1.0) + rs_exp(mt, pexp));
      if (prain < c.PLimSnow) {
        snow = prec_ts;
        rain = R4(0.0);
      } else if (prain > c.PLimRain) {
        rain = prec_ts;
        snow = R4(0.0);
      } else {
        snow = prec_ts / R4(2.);
        rain = snow;
      }
    }
  }
  s.wat = s.wat + rain;
  s.snow = s.snow + snow;
}

/* src/BoundaryLayer.f90:3-109 (+ calcRaero :112-131, CalcLE :134-190).
 * Outputs: blcond, le (LE_Flux), evap (EvapmmTS). */
__device__ __forceinline__ void boundary_layer(const RsConstants &c, const MathTab &mt,
                                               double tsurf, double tair, double vz, double rhz,
                                               double wat, double &blcond, double &le,
                                               double &evap) {
  const double ConvLim = R4(0.001);
  const double TaK = tair + R4(273.15);
  const double AirDens = rs_div(R4(100000.0), R4(287.05) * TaK);
  const double AirHCap = R4(1005.0) + rs_div((TaK - R4(250.0)) * (TaK - R4(250.0)), R4(3364.));
  const double AirVCap = AirHCap * AirDens;
  const double PsychC = R4(0.1) * (R4(0.00063) * TaK + R4(0.47496));
  const double WatDen = R4(-0.0050) * tsurf * tsurf + R4(0.0079) * tsurf + R4(1000.0028);
  /* loop invariants of :78-79, same association as the reference */
  const double stab_num = -c.VK_Const * c.ZRefT * c.Grav;
  const double dT = tsurf - tair;
  const double stab_den0 = AirVCap * (tair + R4(273.15));
  const double vkvz = c.VK_Const * vz;
  const double avk = AirVCap * c.VK_Const;

  double PSIM = R4(0.0), PSIH = R4(0.0);
  double BLCond = 0.0, BLCond_Old;
  for (int j = 1; j <= 40; ++j) {
    BLCond_Old = BLCond;
    const double UStar = rs_div(vkvz, c.logUstar + PSIM);
    BLCond = rs_div(avk * UStar, c.logCond + PSIH);
    double Stab = rs_div(stab_num * BLCond * dT, stab_den0 * (UStar * UStar * UStar));
    if (Stab > 1) Stab = 1;
    if (Stab > 0) {
      PSIH = R4(4.7) * Stab;
      PSIM = PSIH;
    } else {
      PSIH = R4(-2.0) * rs_log(mt, (R4(1.0) + rs_sqrt(R4(1.0) - R4(16.0) * Stab)) / R4(2.0));
      PSIM = R4(0.6) * PSIH;
    }
    if ((j >= 5) && (fabs(BLCond - BLCond_Old) < ConvLim)) break;
  }
  blcond = BLCond;

  double RAero = rs_div((c.logMom + PSIM) * (c.logHeat + PSIH), c.VK_Const * c.VK_Const * vz);
  if (RAero > R4(30.0)) RAero = R4(30.);

  /* Magnus formula over ice (T < 0) or water: one exp per temperature, the
   * coefficients are selected instead of the whole expression being branched */
  const double as = (tsurf < 0) ? R4(21.875) : R4(17.269);
  const double bs = (tsurf < 0) ? R4(265.5) : R4(237.3);
  const double ESurf = R4(0.61078) * rs_exp(mt, rs_div(as * tsurf, tsurf + bs));
  const double aa = (tair < 0) ? R4(21.875) : R4(17.269);
  const double ba = (tair < 0) ? R4(265.5) : R4(237.3);
  const double ESat = R4(0.61078) * rs_exp(mt, rs_div(aa * tair, tair + ba));
  double hum = R4(0.01) * rhz;
  if (hum > R4(1.0)) hum = R4(1.0);
  const double EAir = hum * ESat;
  le = rs_div(AirDens * AirHCap * (ESurf - EAir), PsychC * RAero);
  if (tsurf >= R4(0.0))
    evap = rs_div(le, c.LVap * WatDen) * R4(1000.0) * c.DTSecs;
  else
    evap = rs_div(le, c.LFus * WatDen) * R4(1000.0) * c.DTSecs;
  if ((le > R4(0.0)) && (wat <= R4(0.0))) {
    le = R4(0.0);
    evap = R4(0.0);
  }
}

/* Volumetric heat capacity of layer j from its (stale) TmpNw value
 * (src/BalanceModel.f90:215-236). */
__device__ __forceinline__ double layer_vsh(const RsConstants &c, int j, double T) {
  /* both branches are a dozen flops: evaluate the water polynomials
   * unconditionally and select, so 15 layers cost no branches */
  const double tmp2 = T * T;
  const double RooW = R4(-0.0050) * tmp2 + R4(0.0079) * T + R4(1000.0028);
  const double CW = R4(0.0000102) * tmp2 * tmp2 - R4(0.0017169) * tmp2 * T + R4(0.11516) * tmp2 -
                    R4(3.4739) * T + R4(4217.2);
  const bool water = (T >= 0);
  const double RooWT = water ? RooW : R4(920.0);
  const double CWT = water ? CW : R4(2100.0);
  const double CHWT = RooWT * CWT;
  return c.dryCap[j] + c.WCont[j] * CHWT;
}

/* src/Storage.f90:319-402.  Writes T(1), T(2) (the new profile) and q2melt.
 * The TsurfAve assignment at :389-394 is overwritten unconditionally at
 * src/BalanceModel.f90:78-84 and is dropped. */
template <class Prof>
__device__ __forceinline__ void melting(Scalars &s, Prof &T, double hstor, double hs1,
                                        bool in_coupling_phase, double last_tsurf_obs) {
  if ((s.snow > R4(0.0)) || (s.ice > R4(0.0)) || (s.ice2 > R4(0.0))) {
    if ((hstor <= R4(0.00001)) || (s.tsurf <= s.t4melt) || (s.q2melt <= 0) ||
        (in_coupling_phase && last_tsurf_obs < s.t4melt)) {
      if (s.tsurf < R4(0.5)) {
        s.q2melt = R4(0.0);
        return;
      } else if (s.tsurf > R4(2.0)) {
        const double QAvail = hs1 * (T.get(1) - s.t4melt);
        if (QAvail < s.q2melt) s.q2melt = QAvail;
        return;
      }
    }
    const double QAvail = hs1 * (T.get(1) - s.t4melt);
    if (s.q2melt >= QAvail) {
      s.q2melt = QAvail;
      T.set(1, s.t4melt + R4(0.01));
      T.set(2, s.t4melt + R4(0.01));
    } else {
      const double QLeftOver = QAvail - s.q2melt;
      T.set(1, s.t4melt + rs_div(QLeftOver, hs1));
      T.set(2, s.t4melt + R4(0.01));
    }
  } else {
    s.q2melt = R4(0.0);
  }
}

/* src/Cond.f90:69-103 + src/Cond.f90:9-65 + src/Storage.f90:33-314,409-432 +
 * src/Cond.f90:105-139: wear factors, the four storages, melt heat for the next
 * step, albedo for the next step. */
__device__ __forceinline__ void road_condition(const RsConstants &c, Scalars &s, double evap) {
  /* WearFactors */
  double SnowTran = c.wSnowTran * s.snow;
  SnowTran = (SnowTran > R4(0.01)) ? SnowTran : R4(0.01);
  if (s.snow < R4(0.2)) SnowTran = SnowTran * 3;
  SnowTran = SnowTran * c.Tph;
  double IceWear = c.wIce * s.ice;
  IceWear = (IceWear > R4(0.01)) ? IceWear : R4(0.01);
  IceWear = IceWear * c.Tph;
  double IceWear2 = c.wIce2 * s.ice2;
  IceWear2 = (IceWear2 > R4(0.01)) ? IceWear2 : R4(0.01);
  IceWear2 = IceWear2 * c.Tph;
  double DepWear = c.wDep * s.dep;
  DepWear = (DepWear > R4(0.01)) ? DepWear : R4(0.01);
  DepWear = DepWear * c.Tph;
  double WatWear = c.wWat * s.wat;
  WatWear = (WatWear > R4(0.06)) ? WatWear : R4(0.06);
  WatWear = 10 * WatWear * c.Tph;

  /* RoadCond head: hysteresis (:34-39); SnowType reset to dry (:32) */
  bool wet = false;
  if (s.verycold && (s.tsurf > c.TLimColdH)) s.verycold = false;
  if (!s.verycold && (s.tsurf < c.TLimColdL)) s.verycold = true;

  /* WaterStorage, src/Storage.f90:33-84 */
  if ((s.snow <= R4(0.0)) && (s.ice <= R4(0.0)) && (s.dep <= R4(0.0)) && (s.tsurf > c.TLimDew)) {
    if (s.wat > c.MaxPormms)
      s.wat = s.wat - evap;
    else
      s.wat = s.wat - c.PorEvaF * evap;
  }
  if (s.wat > R4(0.0)) {
    if (s.wat < c.WWearLim) WatWear = R4(0.0);
    if (s.wat > c.WWetLim)
      s.wat = s.wat - WatWear;
    else
      s.wat = s.wat - c.DampWearF * WatWear;
  }
  if (s.wat < c.MinWatmms) s.wat = R4(0.0);
  if (s.wat > c.MaxWatmms) s.wat = c.MaxWatmms;
  double ext = s.wat - c.MaxPormms;
  ext = (ext > R4(0.)) ? ext : R4(0.);

  /* SnowStorage, src/Storage.f90:88-196 */
  {
    double WatSnowRat;
    const double RDummy = ext + s.snow;
    if (RDummy > R4(0.001))
      WatSnowRat = rs_div(ext, RDummy);
    else
      WatSnowRat = R4(0.0);
    if (s.snow > R4(0.0)) {
      if (WatSnowRat > c.WetSnowFormR) wet = true;
      if (s.dep > R4(0.0)) {
        s.ice = s.ice + s.dep;
        s.dep = R4(0.0);
      }
      if ((s.q2melt > R4(0.0)) && (s.tsurf >= c.TLimMeltSnow)) {
        const double Melted = rs_div(s.q2melt * c.DTSecs, c.WatMHeat * c.WatDens);
        s.snow = s.snow - R4(1000.) * Melted;
        s.wat = s.wat + R4(1000.) * Melted;
      }
    }
    if (s.snow > R4(0.0)) {
      s.snow = s.snow - SnowTran;
      s.ice = s.ice + c.wSnow2Ice * SnowTran;
      s.ice2 = s.ice2 + c.wSnow2Ice * SnowTran;
    }
    if ((s.snow > R4(0.0)) && wet) {
      if (WatSnowRat > c.WetSnowMeltR) {
        s.wat = s.wat + s.snow;
        s.snow = R4(0.0);
      }
      if (s.tsurf < c.TLimFreeze) {
        s.ice = s.ice + s.snow + s.wat;
        s.ice2 = s.ice2 + s.snow + s.wat;
        s.snow = R4(0.0);
        s.wat = R4(0.0);
      }
    }
    if (s.snow < c.MinSnowmms) s.snow = R4(0.0);
    if (s.snow > c.MaxSnowmms) s.snow = s.snow - (c.MaxSnowmms / R4(2.));
  }

  /* IceStorage, src/Storage.f90:199-267 */
  if (s.tsurf < c.TLimFreeze && s.wat > R4(0.0)) {
    s.ice = s.ice + s.wat;
    s.ice2 = s.ice2 + s.wat;
    s.wat = R4(0.0);
  }
  if ((s.snow <= R4(0.)) && (s.ice > R4(0.))) {
    if ((s.q2melt > R4(0.0)) && (s.tsurf >= c.TLimMeltIce)) {
      const double Melted = rs_div(s.q2melt * c.DTSecs, c.WatMHeat * c.WatDens);
      s.ice = s.ice - R4(1000.) * Melted;
      s.ice2 = s.ice2 - R4(1000.) * Melted;
      s.wat = s.wat + R4(1000.) * Melted;
    }
  }
  if (s.ice > R4(0.)) s.ice = s.ice - IceWear;
  if (s.ice2 > R4(0.)) s.ice2 = s.ice2 - IceWear2;
  if (s.ice < c.MinIcemms) s.ice = R4(0.0);
  if (s.ice > c.MaxIcemms) s.ice = c.MaxIcemms;
  if (s.ice2 < c.MinIcemms) s.ice2 = R4(0.0);
  if (s.ice2 > c.MaxIcemms) s.ice2 = c.MaxIcemms;

  /* DepositStorage, src/Storage.f90:271-314 */
  if (evap < R4(0.0)) s.dep = s.dep - evap;
  if (s.tsurf > c.TLimMeltDep) {
    s.wat = s.wat + s.dep;
    s.dep = R4(0.0);
  }
  if ((s.snow <= R4(0.0)) && (s.dep > 0)) s.dep = s.dep - DepWear;
  if (s.dep < c.MinDepmms) s.dep = R4(0.0);
  if (s.dep > c.MaxDepmms) {
    s.wat = s.wat + (s.dep - c.MaxDepmms);
    s.dep = c.MaxDepmms;
  }

  /* RoadCond tail, src/Cond.f90:61-62 */
  if (s.wat < c.MinWatmms) s.wat = R4(0.0);
  if (s.wat > c.MaxWatmms) s.wat = c.MaxWatmms;

  /* NewMeltFreezeHeat, src/Storage.f90:409-432 */
  s.q2melt = R4(0.0);
  if (s.snow > R4(0.0)) {
    s.q2melt = rs_div(c.WatMHeat * c.WatDens * rs_div(s.snow, R4(1000.)), c.DTSecs);
    s.t4melt = c.TLimMeltSnow;
  }
  if ((s.snow <= R4(0.0)) && (s.ice > R4(0.0))) {
    s.q2melt = rs_div(c.WatMHeat * c.WatDens * rs_div(s.ice, R4(1000.)), c.DTSecs);
    s.t4melt = c.TLimMeltIce;
  }
  if (s.q2melt < R4(0.0)) s.q2melt = R4(0.0);

  /* CalcAlbedo, src/Cond.f90:105-139 */
  {
    double IceSum = R4(0.5) * (s.ice + s.ice2) + s.dep;
    const double IceMax = R4(1.5);
    if (IceSum < R4(0.0)) IceSum = R4(0.0);
    double alb = c.AlbDry;
    if (s.snow > R4(0.01) && s.snow > s.ice) {
      alb = c.AlbSnow;
    } else if (s.ice > R4(0.01) || s.dep > R4(0.01)) {
      if (IceSum < IceMax)
        alb = c.AlbDry + rs_div(IceSum, IceMax) * (c.AlbSnow - c.AlbDry);
      else
        alb = c.AlbSnow;
    }
    s.albedo = alb;
  }
}

/* roadModelOneStep (examples/example1/src/Simulation.f90:120-172) with
 * BalanceModelOneStep (src/BalanceModel.f90:7-86) inlined, in two halves so that
 * the caller can issue the next time index's forcing loads between them.
 * On entry: tair/vz/rhz/prec_ts are the current atm values (after
 * SetCurrentValues / relaxation / lastValues), T is Tmp(1..N) possibly with obs
 * forcing applied, s.tsurf is up to date. */
struct Fluxes {
  double blcond, le, evap, rnet, trffric;
};

/* What the coupling machinery feeds into a step (src/BalanceModel.f90:44-45,71-73);
 * off the coupling path: coefficients 1.0, not in a coupling phase. */
struct CouplingInputs {
  double sw_cof = 1.0, lw_cof = 1.0, last_tsurf_obs = 0.0;
  bool in_phase = false;
};

/* first half: precipitation -> storages, day/night, boundary layer, net radiation */
__device__ __forceinline__ Fluxes model_step_fluxes(const RsConstants &c, const MathTab &mt,
                                                    Scalars &s, double tair, double vz, double rhz,
                                                    double prec_ts, double sw, double lw,
                                                    int32_t phase, int32_t hour,
                                                    const CouplingInputs &cp = CouplingInputs()) {
  Fluxes fx;
  precipitation_to_storage(c, mt, s, phase, prec_ts, tair, rhz);

  /* SetDayDependendVariables, src/BalanceModel.f90:354-387 */
  double calm;
  if (((double)hour >= c.NightOn) || ((double)hour <= c.NightOff)) {
    calm = c.CalmLimNgt;
    fx.trffric = c.TrfFricNgt;
  } else {
    calm = c.CalmLimDay;
    fx.trffric = c.TrFfricDay;
  }
  if (vz < calm) vz = calm;

  boundary_layer(c, mt, s.tsurf, tair, vz, rhz, s.wat, fx.blcond, fx.le, fx.evap);

  /* CalcRNet, src/BalanceModel.f90:282-307 (SwRadCof = LwRadCof = 1.0 off the
   * coupling path; x*1.0 is exact, so the compiler drops the factors there) */
  const double TsurfK = s.tsurf + R4(273.15);
  const double TsurfK2 = TsurfK * TsurfK;
  const double RBB = c.Emiss * c.SB_Const * (TsurfK2 * TsurfK2);
  fx.rnet = (R4(1.) - s.albedo) * sw * cp.sw_cof + c.Emiss * lw * cp.lw_cof - RBB;
  return fx;
}

/* second half: ground profile, melting, new surface temperature, storages.
 * depth_i is modelInput%depth(i). */
template <class Prof>
__device__ __forceinline__ void model_step_ground(const RsConstants &c, Scalars &s, Prof &T,
                                                  double tbot, double tair, const Fluxes &fx,
                                                  double depth_i,
                                                  const CouplingInputs &cp = CouplingInputs(),
                                                  const Prof *stale_all = nullptr) {
  const int N = T.nlayers();
  /* CalcHCapHCond + calcCapDZCondDZ + calcProfile fused
   * (src/BalanceModel.f90:189-251, 132-155, 90-129) */
  const double t1old = T.get(1), t2old = T.get(2);
  const double Sens = fx.blcond * (tair - t1old);
  double Gprev = fx.rnet - fx.le + fx.trffric + Sens;
  double hs1 = 0.0;
#pragma unroll
  for (int j = 1; j <= N; ++j) {
    const double tj = T.get(j);
    /* TmpNw(j) as CalcHCapHCond sees it (src/BalanceModel.f90:215): equal to Tmp(j) except
     * for layers 1-2 after observation forcing, and for EVERY layer on the first step after
     * a coupling restore (Tmp is restored, TmpNw is not: src/Coupling.f90:245-247) */
    const double tstale = stale_all ? stale_all->get(j) : (j == 1) ? s.tnw1 : (j == 2) ? s.tnw2 : tj;
    const double vsh = layer_vsh(c, j, tstale);
    if (j == 1) hs1 = rs_div(vsh * c.HSfac1, c.twoDT);
    const double capDZ = -rs_div(1.0, c.DyC[j] * vsh);
    const double tnext = (j == N) ? tbot : T.get(j + 1);
    const double G = c.condDZ[j] * (tnext - tj);
    T.set(j, tj + c.DTSecs * (capDZ * (G - Gprev)));
    Gprev = G;
  }

  /* calcHStor, src/BalanceModel.f90:311-322 */
  const double T1Ave = (t1old + R4(3.) * t2old) / R4(4.);
  const double TN1Ave = (T.get(1) + R4(3.) * T.get(2)) / R4(4.);
  const double hstor = hs1 * (TN1Ave - T1Ave);

  melting(s, T, hstor, hs1, cp.in_phase, cp.last_tsurf_obs);

  /* Tmp = TmpNw; new TsurfAve (src/BalanceModel.f90:60-84) */
  s.tnw1 = T.get(1);
  s.tnw2 = T.get(2);
  const double depth = (c.tsurfOutputDepth >= R4(0.0)) ? c.tsurfOutputDepth : depth_i;
  s.tsurf = surface_temperature(c, T, tbot, depth);

  road_condition(c, s, fx.evap);
}

/* Sky view / local horizon: the per-point, per-step remainder of calcElevationAzimuth
 * (src/SunPosition.f90:123-193) and ModRadiationBySurroundings (src/ModRadiation.f90:7-73).
 * sun[4] = {ra, stG, sin decl, cos decl} comes from the host (rs_sun_table, libm).  The
 * solar position only ever acts through discrete outcomes (sun above the horizon line or
 * not, which degree of azimuth, elevation > 0), so the device cos/acos need not reproduce
 * libm's last bit: a different outcome needs the elevation within ~1e-14 deg of the horizon
 * value or the azimuth within ~1e-13 deg of a half degree.  Returns false where the
 * reference would `stop` (|cos| >= 1.001: cannot happen for real inputs). */
__device__ __forceinline__ bool sky_view_radiation(const double *sun, double sin_lat,
                                                   double cos_lat, double lon_rad, double sky_view,
                                                   double albedo_surr, const double *horizons,
                                                   int64_t hstride, double &sw, double &sw_dir,
                                                   double &lw, double lw_net) {
  const double pi = 3.141592653589793; /* 4*atan(1.0_8) */
  const double ra = sun[0], stG = sun[1], sin_decl = sun[2], cos_decl = sun[3];
  const double cos_dec_lat = cos_decl * cos_lat;
  const double sin_dec_lat = sin_decl * sin_lat;
  double hac = (stG + lon_rad - ra);
  const double cosah = ::cos(hac);
  const double cos_elev = sin_dec_lat + cos_dec_lat * cosah;
  double chi;
  if (cos_elev >= R4(1.0) && cos_elev < R4(1.001)) {
    chi = R4(0.);
  } else if (cos_elev >= R4(1.001)) {
    return false;
  } else if (cos_elev > R4(-1.001) && cos_elev <= R4(-1.0)) {
    chi = pi;
  } else {
    chi = ::acos(cos_elev);
  }
  double elevation = R4(90.0) - chi * (R4(180.) / pi);
  if (hac < R4(0.))
    hac = 2 * pi + hac;
  else if (hac > 2 * pi)
    hac = hac - 2 * pi;
  double azimuth;
  if (elevation > 0) {
    const double cosele = ::cos((pi / R4(2.0)) - chi);
    if (cosele >= R4(-0.0001) && cosele < R4(0.0001)) {
      azimuth = R4(-9999.9);
    } else {
      const double precos = (sin_decl * cos_lat - cos_decl * sin_lat * cosah) / cosele;
      if (precos >= R4(1.0) && precos < R4(1.001))
        azimuth = R4(0.0);
      else if (precos >= R4(1.001))
        return false;
      else if (precos > R4(-1.001) && precos <= R4(-1.0))
        azimuth = pi;
      else
        azimuth = ::acos(precos);
    }
    if (hac < pi) azimuth = 2 * pi - azimuth;
    azimuth = azimuth * (R4(180.) / pi);
  } else {
    azimuth = R4(-9999.9);
    elevation = R4(-9999.9);
  }
  /* ModRadiationBySurroundings */
  double dif_sw = sw - sw_dir;
  const double lw_surroundings = lw_net - lw;
  int azim_idx = (int)__builtin_round(azimuth); /* NINT */
  if (azim_idx == 360) azim_idx = 0;
  double horizon = R4(0.);
  if (horizons && azim_idx >= 0 && azim_idx < 360) horizon = horizons[(int64_t)azim_idx * hstride];
  const double shadow_fac = (horizon > elevation) ? R4(0.0) : R4(1.0);
  if (elevation > R4(0.0)) {
    sw_dir = sw_dir * shadow_fac;
    const double sw_ref = albedo_surr * sw_dir + albedo_surr * dif_sw;
    dif_sw = sky_view * dif_sw + (R4(1.0) - sky_view) * sw_ref;
    sw = dif_sw + sw_dir;
  }
  lw = sky_view * lw + (R4(1.0) - sky_view) * (-lw_surroundings);
  return true;
}

/* CheckValues, src/InputOutput.f90:45-84 (sky-view checks: see the general kernel) */
__device__ __forceinline__ bool check_values(const Forcing &f, double tsurf, bool has_tdew) {
  bool bad = f.tair < R4(-90.0) || f.tair > R4(100.0) || f.rhz < R4(-0.1) || f.rhz > R4(120.0) ||
             f.vz < R4(-1.0) || f.vz > R4(100.0) || f.sw < R4(-0.1) || f.sw > R4(4000.0) ||
             f.lw < R4(-0.1) || f.lw > R4(1000.0) || f.prec < R4(-0.1) || f.prec > R4(500.0);
  if (has_tdew) bad = bad || f.tdew < -90 || f.tdew > R4(100.0);
  bad = bad || tsurf < R4(-100.0) || tsurf > R4(100.0);
  return bad;
}

}  // namespace rs
