/* rs_const_f32.h - single-precision mirror of RsConstants for the fp32 kernels
 * (BASELINE config 5).  Same member names, so the type-parameterised physics
 * (rs_physics_body.inc) compiles against either. */
#pragma once
#include "../../include/roadsurf.h"

#define RS_CONST_SCALARS(X)                                                                      \
  X(DTSecs) X(Tph) X(tsurfOutputDepth) X(twoDT) X(HSfac1) X(logMom) X(logHeat) X(logCond)        \
  X(logUstar) X(VK_Const) X(ZRefT) X(Grav) X(LVap) X(LFus) X(Emiss) X(SB_Const) X(Albedo0)       \
  X(NightOn) X(NightOff) X(CalmLimDay) X(CalmLimNgt) X(TrfFricNgt) X(TrFfricDay) X(MaxPormms)    \
  X(MissValI) X(MinPrecmm) X(MinWatmms) X(MinSnowmms) X(MinDepmms) X(MinIcemms) X(MaxSnowmms)    \
  X(MaxDepmms) X(MaxIcemms) X(MaxWatmms) X(AlbDry) X(AlbSnow) X(WatDens) X(WatMHeat) X(PorEvaF)  \
  X(DampWearF) X(TLimFreeze) X(TLimMeltSnow) X(TLimMeltIce) X(TLimMeltDep) X(TLimDew)            \
  X(TLimColdH) X(TLimColdL) X(WetSnowFormR) X(WetSnowMeltR) X(PLimSnow) X(PLimRain) X(WWetLim)   \
  X(WWearLim) X(T4Melt0) X(wSnowTran) X(wSnow2Ice) X(wIce) X(wIce2) X(wDep) X(wWat)

struct RsConstantsF {
  int32_t NLayers, SimLen, use_relaxation, force_tsurf;
  float ZDpth[RS_MAX_LAYERS + 2], DyC[RS_MAX_LAYERS + 2], condDZ[RS_MAX_LAYERS + 2],
      WCont[RS_MAX_LAYERS + 2], dryCap[RS_MAX_LAYERS + 2];
#define X(n) float n;
  RS_CONST_SCALARS(X)
#undef X
  /* made ON THE DEVICE from the members above, by the expressions the lanes evaluate (prepare_constants_f32,
   * rs_kernels_f32.hip): capDZ of a frozen layer and HS(1) of a frozen top layer (src/BalanceModel.f90:132-155,
   * 215-241: the heat capacity of a layer below 0 C is a constant), 1 / twoDT */
  float capDZF[RS_MAX_LAYERS + 2], hs1F, r_twoDT;
  /* ... and per layer, side by side (one scalar load per layer): A = DyC WCont, B = DyC dryCap - capDZ =
   * -1 / (A chwt + B) -, condDZ, capDZ of the frozen layer */
  float lk4[RS_MAX_LAYERS + 2][4];
  /* coupling (step_kernel_f32_coupled): settings%use_coupling, the window length as an integer and as a real
   * (src/Coupling.f90:512-517), couplingEffectReduction */
  int32_t use_coupling, cplLenI;
  float cplLenR, cplReduction;
};

static inline void rs_constants_to_f32(const RsConstants &c, RsConstantsF &f) {
  f.NLayers = c.NLayers; f.SimLen = c.SimLen; f.use_relaxation = c.use_relaxation;
  f.force_tsurf = c.force_tsurf;
  for (int i = 0; i < RS_MAX_LAYERS + 2; ++i) {
    f.ZDpth[i] = (float)c.ZDpth[i]; f.DyC[i] = (float)c.DyC[i]; f.condDZ[i] = (float)c.condDZ[i];
    f.WCont[i] = (float)c.WCont[i]; f.dryCap[i] = (float)c.dryCap[i];
  }
#define X(n) f.n = (float)c.n;
  RS_CONST_SCALARS(X)
#undef X
  for (int i = 0; i < RS_MAX_LAYERS + 2; ++i) f.capDZF[i] = 0.f;
  f.hs1F = 0.f;
  f.r_twoDT = 0.f;
  for (int i = 0; i < RS_MAX_LAYERS + 2; ++i)
    for (int q = 0; q < 4; ++q) f.lk4[i][q] = 0.f;
  f.use_coupling = c.use_coupling; f.cplLenI = c.cplLenI;
  f.cplLenR = (float)c.cplLenR; f.cplReduction = (float)c.cplReduction;
}
