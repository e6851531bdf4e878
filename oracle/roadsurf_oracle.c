/*
 * oracle/roadsurf_oracle.c — TEST INFRASTRUCTURE, not product code.
 *
 * A scalar, one-point-at-a-time CPU restatement in C of the RoadSurf hot path
 * (reference: fmidev/RoadSurf v1.6.1, Fortran).  It exists to CHECK the HIP
 * path; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.  The product path never calls into this file.
 *
 * Parity status: PINNED.  The reference ships no tests or golden vectors
 * (SURVEY.md 4), so this restatement is pinned against outputs of the
 * reference itself, built here by oracle/build_ref.sh into
 * oracle/_ref/libroadsurf_ref.so (amdflang -O2) and captured as fixtures in
 * tests/golden/ by tests/golden/make_golden.py.  tests/test_oracle_vs_golden.py
 * requires bit-for-bit equality on those fixtures.
 *
 * Numeric rules followed everywhere:
 *  - every unsuffixed Fortran real literal is REAL(4): written R4(x) = (double)x##f;
 *  - REAL(4) (op) REAL(4) sub-expressions are evaluated in float first;
 *  - evaluation order is the Fortran one (left to right within a precedence
 *    level), no FMA contraction (-ffp-contract=off), IEEE / and sqrt;
 *  - exp/log/pow/sin are glibc's, as in the reference build.
 *
 * Each function cites the reference file:line it restates.
 */
#include "../include/roadsurf.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define R4(x) ((double)(x##f))
#define MAXL RS_MAX_LAYERS

enum { SNOW_DRY = 1, SNOW_WET = 2 }; /* src/Constants.h:10-11 */

typedef struct {
  /* settings (src/ModelSettings.f90.inc) */
  int SimLen, InitLenI, NLayers, use_coupling, use_relaxation, force_tsurf, failed;
  double DTSecs, tsurfOutputDepth, Tph;
  double NightOn, NightOff, CalmLimDay, CalmLimNgt, TrfFricNgt, TrFfricDay;
  /* phy (src/PhysicalParameters.f90.inc) */
  double VK, SB, ZRefT, logMom, logHeat, logCond, logUstar, Grav, Emiss;
  double Poro1, Poro2, vsh1, vsh2, LVap, LFus, MaxPormms;
  /* ground (src/GroundVariables.f90.inc); arrays 0..N+1 like the reference */
  double Albedo, HStor;
  double condDZ[MAXL + 2], capDZ[MAXL + 2], WCont[MAXL + 2], VSH[MAXL + 2],
      HS[MAXL + 2], CC[MAXL + 2], Tmp[MAXL + 2], TmpNw[MAXL + 2], DyC[MAXL + 2],
      DyK[MAXL + 2], ZDpth[MAXL + 2];
  /* surf (src/SurfaceVariables.f90.inc) */
  double TsurfAve, Wat, Snow, Ice, Ice2, Dep, Q2Melt, T4Melt, TrfFric, Evap;
  int VeryCold, WearSurf;
  /* atm (src/AtmVariables.f90.inc) */
  double Tair, VZ, Rhz, PrecInTStep, BLCond, RNet, LE, CalmLim, Rainmm, Snowmm;
  double TairR, VZR, RhzR, TairInitEnd, VZInitEnd, RhzInitEnd;
  int SnowType;
  /* cond (src/RoadCondParameters.f90.inc), the members that are read */
  double MissValI, MinPrecmm, MinWatmms, MinSnowmms, MinDepmms, MinIcemms,
      MaxSnowmms, MaxDepmms, MaxIcemms, MaxWatmms, AlbDry, AlbSnow, WatDens,
      WatMHeat, PorEvaF, DampWearF, TLimFreeze, TLimMeltSnow, TLimMeltIce,
      TLimMeltDep, TLimDew, TLimColdH, TLimColdL, WetSnowFormR, WetSnowMeltR,
      PLimSnow, PLimRain, WWetLim, WWearLim, Snow2IceFac;
  /* coupling (src/CouplingVariables.f90.inc); NObs = CoupPhaseN = 1 in this model version */
  double SwRadCof, LwRadCof, lastTsurfObs;
  int inCouplingPhase;
  int Coupling_iterations, start_coupling_again, Coupling_failed, couplingStartI, couplingEndI,
      saveDatai, VeryColdSave, coupling_minutes;
  double TsurfNearestAbove, TsurfNearestBelow, RadCoeff, RadCoefNearestAbove, RadCoefNearestBelow,
      RadCoeffPrevious, SW_correction, LW_correction, Tsurf_end_coup1, couplingEffectReduction;
  double TsurfAveSave, SrfWatmmsSave, SrfIce2mmsSave, SrfDepmmsSave, SrfSnowmmsSave, AlbedoSave;
  double TmpSave[MAXL + 2];
  double *SWSave, *SWDirSave, *LWSave; /* window copies, src/Coupling.f90:204-208 */
  int skyview_on;
  const InputParameters *P;
  const LocalParameters *lp;
  /* wear factors (src/WearingFactors.f90.inc) */
  double SnowTran, DepWear, IceWear, IceWear2, WatWear;
} Model;

/* src/BalanceModel.f90:390-417.  ZDpth has NLayers+1 entries (1-based). */
static double getTempAtDepth(const Model *m, double depth) {
  const int zlen = m->NLayers + 1;
  int idx;
  if (fabs(depth - R4(0.0)) < R4(0.00001)) return m->Tmp[1];
  if (depth > m->ZDpth[zlen]) return m->Tmp[zlen];
  for (idx = 1; idx <= zlen - 1; ++idx)
    if (depth > m->ZDpth[idx] && depth <= m->ZDpth[idx + 1]) break;
  return m->Tmp[idx] + (depth - m->ZDpth[idx]) * (m->Tmp[idx + 1] - m->Tmp[idx]) /
                           (m->ZDpth[idx + 1] - m->ZDpth[idx]);
}

/* src/BalanceModel.f90:325-351 */
static int JulDay(int syear, int smon, int sday) {
  static const int MonEnd[24] = {0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334,
                                 0, 31, 60, 91, 121, 152, 182, 213, 244, 274, 305, 335};
#define IMIN(a, b) ((a) < (b) ? (a) : (b))
  int leapcorr = 1 - IMIN(syear % 4, 1) + IMIN(syear % 100, 1) - IMIN(syear % 400, 1);
  return MonEnd[smon + leapcorr * 12 - 1] + sday;
}

/* compiler-rt __powisf2, which is what flang lowers REAL(4)**INTEGER to
 * (SURVEY.md Appendix C): square-and-multiply in float. */
static float powi_f32(float a, int b) {
  const int recip = b < 0;
  float r = 1.0f;
  while (1) {
    if (b & 1) r *= a;
    b /= 2;
    if (b == 0) break;
    a *= a;
  }
  return recip ? 1.0f / r : r;
}

/* src/BalanceModel.f90:189-251 */
static void CalcHCapHCond(Model *m) {
  for (int i = 1; i <= m->NLayers; ++i) {
    double RooWT, CWT, CHWT;
    const double T = m->TmpNw[i];
    if (T >= 0) {
      const double tmp2 = T * T;
      RooWT = R4(-0.0050) * tmp2 + R4(0.0079) * T + R4(1000.0028);
      CWT = R4(0.0000102) * tmp2 * tmp2 - R4(0.0017169) * tmp2 * T + R4(0.11516) * tmp2 -
            R4(3.4739) * T + R4(4217.2);
    } else {
      RooWT = R4(920.0);
      CWT = R4(2100.0);
    }
    CHWT = RooWT * CWT;
    if (i <= 2)
      m->VSH[i] = (R4(1.0) - m->Poro1) * m->vsh1 + m->WCont[i] * CHWT;
    else
      m->VSH[i] = (R4(1.0) - m->Poro2) * m->vsh2 + m->WCont[i] * CHWT;
    if (i == 1)
      m->HS[i] = m->VSH[i] * (m->ZDpth[i + 1] - m->ZDpth[i]) / (R4(2.0) * m->DTSecs);
    else
      m->HS[i] = m->VSH[i] * (m->ZDpth[i + 1] - m->ZDpth[i - 1]) / (R4(2.0) * m->DTSecs);
    /* GCond is a dead store in the reference (:248) */
  }
}

/* src/BalanceModel.f90:132-155 */
static void calcCapDZCondDZ(Model *m) {
  for (int j = 1; j <= m->NLayers; ++j) {
    m->condDZ[j] = -(m->CC[j] / m->DyK[j]);
    m->capDZ[j] = -(1.0 / (m->DyC[j] * m->VSH[j]));
  }
}

/* src/BoundaryLayer.f90:112-131 */
static double calcRaero(const Model *m, double PSIM, double PSIH, double VZ) {
  double RAero = (m->logMom + PSIM) * (m->logHeat + PSIH) / (m->VK * m->VK * VZ);
  if (RAero > R4(30.0)) RAero = R4(30.);
  return RAero;
}

/* src/BoundaryLayer.f90:134-190 */
static void CalcLE(Model *m, double TSurfAve, double TAmb, double Rhz, double AirDens,
                   double AirHCap, double PsychC, double RAero, double WatDen) {
  double ESat, ESurf, EAir, hum;
  if (TSurfAve < 0)
    ESat = R4(0.61078) * exp(R4(21.875) * TSurfAve / (TSurfAve + R4(265.5)));
  else
    ESat = R4(0.61078) * exp(R4(17.269) * TSurfAve / (TSurfAve + R4(237.3)));
  ESurf = ESat;
  if (TAmb < 0)
    ESat = R4(0.61078) * exp(R4(21.875) * TAmb / (TAmb + R4(265.5)));
  else
    ESat = R4(0.61078) * exp(R4(17.269) * TAmb / (TAmb + R4(237.3)));
  hum = R4(0.01) * Rhz;
  if (hum > R4(1.0)) hum = R4(1.0); /* Min((0.01*Rhz), 1.0) */
  EAir = hum * ESat;
  m->LE = (AirDens * AirHCap * (ESurf - EAir)) / (PsychC * RAero);
  if (TSurfAve >= R4(0.0))
    m->Evap = (m->LE / (m->LVap * WatDen)) * R4(1000.0) * m->DTSecs;
  else
    m->Evap = (m->LE / (m->LFus * WatDen)) * R4(1000.0) * m->DTSecs;
  if ((m->LE > R4(0.0)) && (m->Wat <= R4(0.0))) {
    m->LE = R4(0.0);
    m->Evap = R4(0.0);
  }
}

/* diagnostic (tools/bl_iterations.py): histogram of the trip count of the loop below */
long oracle_bl_hist[64];
int oracle_bl_hist_on = 0;
/* optional per-call trace for ONE point at a time (single-threaded use) */
unsigned char *oracle_bl_trace = 0;
long oracle_bl_trace_pos = 0, oracle_bl_trace_cap = 0;

/* src/BoundaryLayer.f90:3-109.  Returns the iteration count (known-answer tests). */
static int CalcBLCondAndLE(Model *m) {
  const double ConvLim = R4(0.001);
  const int MaxIter = 40;
  const double TSurfAve = m->TsurfAve;
  const double Tair = m->Tair, VZ = m->VZ, Rhz = m->Rhz;
  double BLCond = m->BLCond, BLCond_Old = BLCond;
  double PSIM = R4(0.0), PSIH = R4(0.0), UStar, Stab, RAero;
  int j;
  const double TaK = Tair + R4(273.15);
  const double AirDens = R4(100000.0) / (R4(287.05) * TaK);
  const double AirHCap = R4(1005.0) + ((TaK - R4(250.0)) * (TaK - R4(250.0))) / R4(3364.);
  const double AirVCap = AirHCap * AirDens;
  const double PsychC = R4(0.1) * (R4(0.00063) * TaK + R4(0.47496));
  const double WatDen = R4(-0.0050) * TSurfAve * TSurfAve + R4(0.0079) * TSurfAve + R4(1000.0028);

  int n_unstable = 0;
  for (j = 1; j <= MaxIter; ++j) {
    BLCond_Old = BLCond;
    UStar = m->VK * VZ / (m->logUstar + PSIM);
    BLCond = AirVCap * m->VK * UStar / (m->logCond + PSIH);
    Stab = -m->VK * m->ZRefT * m->Grav * BLCond * (TSurfAve - Tair) /
           (AirVCap * (Tair + R4(273.15)) * (UStar * UStar * UStar));
    if (Stab > 1) Stab = 1;
    if (Stab > 0) {
      PSIH = R4(4.7) * Stab;
      PSIM = PSIH;
    } else {
      PSIH = R4(-2.0) * log((R4(1.0) + sqrt(R4(1.0) - R4(16.0) * Stab)) / R4(2.0));
      PSIM = R4(0.6) * PSIH;
      n_unstable++;
    }
    if ((fabs(BLCond - BLCond_Old) < ConvLim) && (j >= 5)) break;
  }
  RAero = calcRaero(m, PSIM, PSIH, VZ);
  CalcLE(m, TSurfAve, Tair, Rhz, AirDens, AirHCap, PsychC, RAero, WatDen);
  m->BLCond = BLCond;
  if (oracle_bl_hist_on) {
    const int jj = j > MaxIter ? MaxIter : j;
#pragma omp atomic
    oracle_bl_hist[jj]++;
    if (oracle_bl_trace && oracle_bl_trace_pos + 1 < oracle_bl_trace_cap) {
      oracle_bl_trace[oracle_bl_trace_pos++] = (unsigned char)jj;
      oracle_bl_trace[oracle_bl_trace_pos++] = (unsigned char)n_unstable; /* passes through the log/sqrt branch */
    }
  }
  return j > MaxIter ? MaxIter : j;
}

/* src/BalanceModel.f90:282-307 */
static void CalcRNet(Model *m, double SW, double LW) {
  const double TsurfK = m->TsurfAve + R4(273.15);
  const double TsurfK2 = TsurfK * TsurfK;
  const double RBB = m->Emiss * m->SB * (TsurfK2 * TsurfK2);
  m->RNet = (R4(1.) - m->Albedo) * SW * m->SwRadCof + m->Emiss * LW * m->LwRadCof - RBB;
}

/* src/BalanceModel.f90:90-129 */
static void calcProfile(Model *m) {
  double G[MAXL + 2];
  const int N = m->NLayers;
  const double Sens = m->BLCond * (m->Tmp[0] - m->Tmp[1]);
  G[0] = m->RNet - m->LE + m->TrfFric + Sens;
  memcpy(m->TmpNw, m->Tmp, sizeof(double) * (N + 2));
  for (int j = 1; j <= N; ++j) G[j] = m->condDZ[j] * (m->Tmp[j + 1] - m->Tmp[j]);
  for (int j = 1; j <= N; ++j)
    m->TmpNw[j] = m->Tmp[j] + m->DTSecs * (m->capDZ[j] * (G[j] - G[j - 1]));
}

/* src/BalanceModel.f90:311-322 */
static void calcHStor(Model *m) {
  const double T1Ave = (m->Tmp[1] + R4(3.) * m->Tmp[2]) / R4(4.);
  const double TN1Ave = (m->TmpNw[1] + R4(3.) * m->TmpNw[2]) / R4(4.);
  m->HStor = m->HS[1] * (TN1Ave - T1Ave);
}

/* src/Storage.f90:319-402.  CanMeltingChangeTemperature is .true. on this path
 * (src/Initialization.f90:556; only coupling's snowIceCheck machinery toggles
 * force flags and nothing sets it false). */
static void melting(Model *m, double depth) {
  double QAvail, QLeftOver;
  if ((m->Snow > R4(0.0)) || (m->Ice > R4(0.0)) || (m->Ice2 > R4(0.0))) {
    if ((m->HStor <= R4(0.00001)) || (m->TsurfAve <= m->T4Melt) || (m->Q2Melt <= 0) ||
        (m->inCouplingPhase && m->lastTsurfObs < m->T4Melt)) {
      if (m->TsurfAve < R4(0.5)) {
        m->Q2Melt = R4(0.0);
        return;
      } else if (m->TsurfAve > R4(2.0)) {
        QAvail = m->HS[1] * (m->TmpNw[1] - m->T4Melt);
        if (QAvail < m->Q2Melt) m->Q2Melt = QAvail;
        return;
      }
    }
    QAvail = m->HS[1] * (m->TmpNw[1] - m->T4Melt);
    if (m->Q2Melt >= QAvail) {
      m->Q2Melt = QAvail;
      m->TmpNw[1] = m->T4Melt + R4(0.01);
      m->TmpNw[2] = m->T4Melt + R4(0.01);
    } else {
      QLeftOver = QAvail - m->Q2Melt;
      m->TmpNw[1] = m->T4Melt + (QLeftOver / m->HS[1]);
      m->TmpNw[2] = m->T4Melt + R4(0.01);
    }
    /* :389-394 reads the OLD Tmp through getTempAtDepth; the value is
     * overwritten unconditionally at src/BalanceModel.f90:78-84 */
    if (depth >= 0)
      m->TsurfAve = getTempAtDepth(m, depth);
    else
      m->TsurfAve = R4(0.5) * (m->TmpNw[1] + m->TmpNw[2]);
  } else {
    m->Q2Melt = R4(0.0);
  }
}

/* src/Cond.f90:143-249 (RainIntensity/SnowIntensity/PrecType are dead stores) */
static void CalcPrecType(Model *m, int PrecPhase) {
  int UseInterpr = 1;
  m->Rainmm = R4(0.0);
  m->Snowmm = R4(0.0);
  if ((double)PrecPhase > m->MissValI) {
    UseInterpr = 0;
    if (m->PrecInTStep <= m->MinPrecmm) {
      m->PrecInTStep = R4(0.0);
    } else {
      switch (PrecPhase) {
        case 0: case 1: case 4: case 5:
          m->Rainmm = m->PrecInTStep;
          m->Snowmm = R4(0.0);
          m->SnowType = SNOW_WET;
          break;
        case 2:
          m->Snowmm = m->PrecInTStep / R4(2.);
          m->Rainmm = m->Snowmm;
          m->SnowType = SNOW_WET;
          break;
        case 3: case 6:
          m->Snowmm = m->PrecInTStep;
          m->Rainmm = R4(0.0);
          break;
        default:
          UseInterpr = 1;
      }
    }
  }
  if (UseInterpr) {
    if (m->PrecInTStep <= m->MinPrecmm) {
      m->PrecInTStep = R4(0.0);
      m->Rainmm = R4(0.0);
      m->Snowmm = R4(0.0);
    } else {
      double PExp, PRain;
      m->Snowmm = R4(0.0);
      PExp = R4(22.0) - R4(2.7) * m->Tair - R4(0.20) * m->Rhz;
      PRain = R4(1.0) / (R4(1.0) + exp(PExp));
      if (PRain < m->PLimSnow) {
        m->Snowmm = m->PrecInTStep;
      } else if (PRain > m->PLimRain) {
        m->Rainmm = m->PrecInTStep;
        m->SnowType = SNOW_WET;
      } else {
        m->Snowmm = m->PrecInTStep / R4(2.);
        m->Rainmm = m->Snowmm;
        m->SnowType = SNOW_WET;
      }
    }
  }
}

/* src/Storage.f90:9-29 */
static void PrecipitationToStorage(Model *m, int PrecPhase) {
  CalcPrecType(m, PrecPhase);
  m->Wat = m->Wat + m->Rainmm;
  m->Snow = m->Snow + m->Snowmm;
}

/* src/BalanceModel.f90:354-387 */
static void SetDayDependendVariables(Model *m, int shour) {
  if (((double)shour >= m->NightOn) || ((double)shour <= m->NightOff)) {
    m->CalmLim = m->CalmLimNgt;
    m->TrfFric = m->TrfFricNgt;
  } else {
    m->CalmLim = m->CalmLimDay;
    m->TrfFric = m->TrFfricDay;
  }
  if (m->VZ < m->CalmLim) m->VZ = m->CalmLim;
}

/* src/BalanceModel.f90:7-86 */
static void BalanceModelOneStep(Model *m, double SWi, double LWi, int hour, double depth_i) {
  double depth;
  SetDayDependendVariables(m, hour);
  CalcBLCondAndLE(m);
  CalcRNet(m, SWi, LWi);
  CalcHCapHCond(m);
  calcCapDZCondDZ(m);
  calcProfile(m);
  calcHStor(m);
  if (m->tsurfOutputDepth >= R4(0.0))
    depth = m->tsurfOutputDepth;
  else
    depth = depth_i;
  melting(m, depth);
  memcpy(m->Tmp, m->TmpNw, sizeof(double) * (m->NLayers + 2));
  if (depth >= 0)
    m->TsurfAve = getTempAtDepth(m, depth);
  else
    m->TsurfAve = (m->Tmp[1] + m->Tmp[2]) / R4(2.0);
}

/* src/Cond.f90:69-103.  The REAL(4) literal products fold in float. */
static void WearFactors(Model *m) {
  m->SnowTran = (double)(0.2f + 0.25f) * m->Snow;
  m->SnowTran = fmax(m->SnowTran, R4(0.01));
  if (m->Snow < R4(0.2)) m->SnowTran = m->SnowTran * 3;
  m->Snow2IceFac = (double)(0.25f / (0.2f + 0.25f));
  m->SnowTran = m->SnowTran * m->Tph;
  m->IceWear = (double)(1.1f * 2.0f * 0.145f) * m->Ice;
  m->IceWear = fmax(m->IceWear, R4(0.01));
  m->IceWear = m->IceWear * m->Tph;
  m->IceWear2 = (double)(1.1f * 2.0f * (4.0f * 0.290f)) * m->Ice2;
  m->IceWear2 = fmax(m->IceWear2, R4(0.01));
  m->IceWear2 = m->IceWear2 * m->Tph;
  m->DepWear = (double)(0.5f * 2.0f * (4.0f * 0.290f)) * m->Dep;
  m->DepWear = fmax(m->DepWear, R4(0.01));
  m->DepWear = m->DepWear * m->Tph;
  m->WatWear = R4(0.145) * m->Wat;
  m->WatWear = fmax(m->WatWear, R4(0.06));
  m->WatWear = 10 * m->WatWear * m->Tph;
}

/* src/Storage.f90:33-84 */
static void WaterStorage(Model *m, double *SrfExtmms, double *SrfPormms) {
  const double MaxPormms = m->MaxPormms;
  if ((m->Snow <= R4(0.0)) && (m->Ice <= R4(0.0)) && (m->Dep <= R4(0.0)) &&
      (m->TsurfAve > m->TLimDew)) {
    if (m->Wat > MaxPormms)
      m->Wat = m->Wat - m->Evap;
    else
      m->Wat = m->Wat - m->PorEvaF * m->Evap;
  }
  if (m->WearSurf && m->Wat > R4(0.0)) {
    if (m->Wat < m->WWearLim) m->WatWear = R4(0.0);
    if (m->Wat > m->WWetLim)
      m->Wat = m->Wat - m->WatWear;
    else
      m->Wat = m->Wat - m->DampWearF * m->WatWear;
  }
  if (m->Wat < m->MinWatmms) m->Wat = R4(0.0);
  if (m->Wat > m->MaxWatmms) m->Wat = m->MaxWatmms;
  *SrfExtmms = fmax(m->Wat - MaxPormms, R4(0.));
  *SrfPormms = fmin(m->Wat, MaxPormms);
}

/* src/Storage.f90:88-196.  forceSnowMelting is .false. off the coupling path. */
static void SnowStorage(Model *m, double *SrfExtmms, double *Melted, double *SrfPormms) {
  double WatSnowRat, RDummy;
  RDummy = *SrfExtmms + m->Snow;
  if (RDummy > R4(0.001))
    WatSnowRat = *SrfExtmms / RDummy;
  else
    WatSnowRat = R4(0.0);
  /* CP%SnowIceRat (:122-127) is written and never read */
  if (m->Snow > R4(0.0)) {
    if (WatSnowRat > m->WetSnowFormR) m->SnowType = SNOW_WET;
  } else {
    m->SnowType = SNOW_DRY;
  }
  if (m->Snow > R4(0.0)) {
    if (m->Dep > R4(0.0)) {
      m->Ice = m->Ice + m->Dep;
      m->Dep = R4(0.0);
    }
  }
  if (m->Snow > R4(0.0)) {
    if ((m->Q2Melt > R4(0.0)) && (m->TsurfAve >= m->TLimMeltSnow)) {
      *Melted = (m->Q2Melt * m->DTSecs) / (m->WatMHeat * m->WatDens);
      m->Snow = m->Snow - R4(1000.) * *Melted;
      m->Wat = m->Wat + R4(1000.) * *Melted;
    }
  }
  if (m->WearSurf && m->Snow > R4(0.0)) {
    m->Snow = m->Snow - m->SnowTran;
    m->Ice = m->Ice + m->Snow2IceFac * m->SnowTran;
    m->Ice2 = m->Ice2 + m->Snow2IceFac * m->SnowTran;
  }
  if ((m->Snow > R4(0.0)) && (m->SnowType == SNOW_WET)) {
    if (WatSnowRat > m->WetSnowMeltR) {
      m->Wat = m->Wat + m->Snow;
      m->Snow = R4(0.0);
      m->SnowType = SNOW_DRY;
    }
    if (m->TsurfAve < m->TLimFreeze) {
      m->Ice = m->Ice + m->Snow + m->Wat;
      m->Ice2 = m->Ice2 + m->Snow + m->Wat;
      m->SnowType = SNOW_DRY;
      m->Snow = R4(0.0);
      m->Wat = R4(0.0);
    }
  }
  *SrfExtmms = fmax(m->Wat - m->MaxPormms, R4(0.));
  *SrfPormms = fmin(m->Wat, m->MaxPormms);
  if (m->Snow < m->MinSnowmms) m->Snow = R4(0.0);
  if (m->Snow > m->MaxSnowmms) m->Snow = m->Snow - (m->MaxSnowmms / R4(2.));
}

/* src/Storage.f90:199-267.  forceIceMelting is .false. off the coupling path. */
static void IceStorage(Model *m, double *Melted, double *SrfExtmms, double *SrfPormms) {
  if (m->TsurfAve < m->TLimFreeze && m->Wat > R4(0.0)) {
    m->Ice = m->Ice + m->Wat;
    m->Ice2 = m->Ice2 + m->Wat;
    m->Wat = R4(0.0);
  }
  if ((m->Snow <= R4(0.)) && (m->Ice > R4(0.))) {
    if ((m->Q2Melt > R4(0.0)) && (m->TsurfAve >= m->TLimMeltIce)) {
      *Melted = (m->Q2Melt * m->DTSecs) / (m->WatMHeat * m->WatDens);
      m->Ice = m->Ice - R4(1000.) * *Melted;
      m->Ice2 = m->Ice2 - R4(1000.) * *Melted;
      m->Wat = m->Wat + R4(1000.) * *Melted;
    }
  }
  if (m->WearSurf && m->Ice > R4(0.)) m->Ice = m->Ice - m->IceWear;
  if (m->WearSurf && m->Ice2 > R4(0.)) m->Ice2 = m->Ice2 - m->IceWear2;
  *SrfExtmms = fmax(m->Wat - m->MaxPormms, R4(0.));
  *SrfPormms = fmin(m->Wat, m->MaxPormms);
  if (m->Ice < m->MinIcemms) m->Ice = R4(0.0);
  if (m->Ice > m->MaxIcemms) m->Ice = m->MaxIcemms;
  if (m->Ice2 < m->MinIcemms) m->Ice2 = R4(0.0);
  if (m->Ice2 > m->MaxIcemms) m->Ice2 = m->MaxIcemms;
}

/* src/Storage.f90:271-314 */
static void DepositStorage(Model *m, double *SrfExtmms, double *SrfPormms) {
  if (m->Evap < R4(0.0)) m->Dep = m->Dep - m->Evap;
  if (m->TsurfAve > m->TLimMeltDep) {
    m->Wat = m->Wat + m->Dep;
    m->Dep = R4(0.0);
  }
  if (m->WearSurf && (m->Snow <= R4(0.0)) && (m->Dep > 0)) m->Dep = m->Dep - m->DepWear;
  *SrfExtmms = fmax(m->Wat - m->MaxPormms, R4(0.));
  *SrfPormms = fmin(m->Wat, m->MaxPormms);
  if (m->Dep < m->MinDepmms) m->Dep = R4(0.0);
  if (m->Dep > m->MaxDepmms) {
    m->Wat = m->Wat + (m->Dep - m->MaxDepmms);
    m->Dep = m->MaxDepmms;
  }
}

/* src/Storage.f90:409-432 */
static void NewMeltFreezeHeat(Model *m) {
  m->Q2Melt = R4(0.0);
  if (m->Snow > R4(0.0)) {
    m->Q2Melt = m->WatMHeat * m->WatDens * (m->Snow / R4(1000.)) / m->DTSecs;
    m->T4Melt = m->TLimMeltSnow;
  }
  if ((m->Snow <= R4(0.0)) && (m->Ice > R4(0.0))) {
    m->Q2Melt = m->WatMHeat * m->WatDens * (m->Ice / R4(1000.)) / m->DTSecs;
    m->T4Melt = m->TLimMeltIce;
  }
  if (m->Q2Melt < R4(0.0)) m->Q2Melt = R4(0.0);
}

/* src/Cond.f90:9-65 */
static void RoadCond(Model *m) {
  double SrfExtmms, SrfPormms, Melted = R4(0.0);
  m->SnowType = SNOW_DRY;
  if (m->VeryCold && (m->TsurfAve > m->TLimColdH)) m->VeryCold = 0;
  if (!m->VeryCold && (m->TsurfAve < m->TLimColdL)) m->VeryCold = 1;
  WaterStorage(m, &SrfExtmms, &SrfPormms);
  SnowStorage(m, &SrfExtmms, &Melted, &SrfPormms);
  IceStorage(m, &Melted, &SrfExtmms, &SrfPormms);
  DepositStorage(m, &SrfExtmms, &SrfPormms);
  if (m->Wat < m->MinWatmms) m->Wat = R4(0.0);
  if (m->Wat > m->MaxWatmms) m->Wat = m->MaxWatmms;
  NewMeltFreezeHeat(m);
}

/* src/Cond.f90:105-139 */
static void CalcAlbedo(Model *m) {
  if (m->WearSurf) {
    double IceSum = R4(0.5) * (m->Ice + m->Ice2) + m->Dep;
    const double IceMax = R4(1.5);
    if (IceSum < R4(0.0)) IceSum = R4(0.0);
    m->Albedo = m->AlbDry;
    if (m->Snow > R4(0.01) && m->Snow > m->Ice) {
      m->Albedo = m->AlbSnow;
    } else if (m->Ice > R4(0.01) || m->Dep > R4(0.01)) {
      if (IceSum < IceMax)
        m->Albedo = m->AlbDry + (IceSum / IceMax) * (m->AlbSnow - m->AlbDry);
      else
        m->Albedo = m->AlbSnow;
    }
  }
}

/* ---- coupling: src/Coupling.f90 ------------------------------------------------ */

/* src/Coupling.f90:172-210.  SrfIcemms is NOT saved (the reference saves Ice2 twice,
 * :194-195); the SW/SW_dir/LW window copies only matter when something mutates those
 * inputs (sky view), see uploadDataForCoupling. */
static void saveDataForCoupling(Model *m, const InputPointers *in, int datai) {
  const int len = m->couplingEndI - m->couplingStartI + 1;
  if (len > 0 && !m->SWSave) {
    m->SWSave = (double *)malloc(sizeof(double) * len);
    m->SWDirSave = (double *)malloc(sizeof(double) * len);
    m->LWSave = (double *)malloc(sizeof(double) * len);
  }
  for (int k = 0; k < len; ++k) { /* :204-208 */
    m->SWSave[k] = in->c_SW[m->couplingStartI + k - 1];
    m->SWDirSave[k] = in->c_SW_dir[m->couplingStartI + k - 1];
    m->LWSave[k] = in->c_LW[m->couplingStartI + k - 1];
  }
  m->saveDatai = datai;
  m->TsurfAveSave = m->TsurfAve;
  m->SrfWatmmsSave = m->Wat;
  m->SrfIce2mmsSave = m->Ice2;
  m->SrfDepmmsSave = m->Dep;
  m->SrfSnowmmsSave = m->Snow;
  m->AlbedoSave = m->Albedo;
  m->VeryColdSave = m->VeryCold;
  memcpy(m->TmpSave, m->Tmp, sizeof(double) * (m->NLayers + 2));
}

/* src/Coupling.f90:213-255 (Tmp restored, TmpNw not; SrfIcemms not restored) */
static void uploadDataForCoupling(Model *m, const InputPointers *in, int *datai) {
  const int len = m->couplingEndI - m->couplingStartI + 1;
  for (int k = 0; k < len && m->SWSave; ++k) { /* :249-253: undo in-place sky-view edits */
    in->c_SW[m->couplingStartI + k - 1] = m->SWSave[k];
    in->c_SW_dir[m->couplingStartI + k - 1] = m->SWDirSave[k];
    in->c_LW[m->couplingStartI + k - 1] = m->LWSave[k];
  }
  *datai = m->saveDatai;
  m->TsurfAve = m->TsurfAveSave;
  m->Wat = m->SrfWatmmsSave;
  m->Ice2 = m->SrfIce2mmsSave;
  m->Dep = m->SrfDepmmsSave;
  m->Snow = m->SrfSnowmmsSave;
  m->Albedo = m->AlbedoSave;
  m->VeryCold = m->VeryColdSave;
  memcpy(m->Tmp, m->TmpSave, sizeof(double) * (m->NLayers + 2));
}

/* src/Coupling.f90:259-289 */
static void snowIceCheck(Model *m) {
  const double obs = m->lastTsurfObs;
  if (obs > m->TLimMeltSnow && m->Snow > R4(0.00)) {
    m->Wat = m->Wat + m->Snow;
    m->Snow = R4(0.00);
  }
  if (obs > m->TLimMeltIce && m->Ice > R4(0.00)) {
    m->Wat = m->Wat + m->Ice;
    m->Ice = R4(0.00);
  }
  if (obs > m->TLimMeltIce && m->Ice2 > R4(0.00)) m->Ice2 = R4(0.00);
  if (obs > m->TLimMeltDep && m->Dep > R4(0.00)) {
    m->Wat = m->Wat + m->Dep;
    m->Dep = R4(0.00);
  }
}

/* src/Coupling.f90:10-96 */
static void CouplingOperations1(Model *m, const InputPointers *in, int *i) {
  const double DTs = m->DTSecs;
  m->inCouplingPhase = 0;
  if (*i >= m->couplingStartI && *i <= m->couplingEndI) m->inCouplingPhase = 1;
  if (*i == m->couplingStartI && m->Coupling_iterations == 0) {
    saveDataForCoupling(m, in, *i);
    m->SwRadCof = R4(1.0);
    m->LwRadCof = R4(1.0);
    m->SW_correction = R4(0.0);
    m->LW_correction = R4(0.0);
  }
  if (m->start_coupling_again) {
    uploadDataForCoupling(m, in, i);
    m->start_coupling_again = 0;
    if (in->c_SW[*i - 1] > in->c_LW[*i - 1] && !m->skyview_on) {
      m->SwRadCof = m->RadCoeff;
      m->LwRadCof = R4(1.0);
    } else {
      m->SwRadCof = R4(1.0);
      m->LwRadCof = m->RadCoeff;
    }
  }
  if (*i > m->couplingEndI) {
    m->SwRadCof = R4(1.0) + m->SW_correction * exp(-((DTs * *i) - (DTs * m->couplingEndI)) /
                                                       m->couplingEffectReduction);
    m->LwRadCof = R4(1.0) + m->LW_correction * exp(-((DTs * *i) - (DTs * m->couplingEndI)) /
                                                       m->couplingEffectReduction);
  }
  if (m->inCouplingPhase) snowIceCheck(m);
}

/* src/Coupling.f90:292-481.  TsurfAve and LastTsurfObs make a round trip through
 * Kelvin (+273.16, -273.16): the rounding it leaves behind is part of the result. */
static void Coupling_control(Model *m) {
  double TDifAbove, TDifBelow;
  m->start_coupling_again = 0;
  m->TsurfAve = m->TsurfAve + R4(273.16);
  m->lastTsurfObs = m->lastTsurfObs + R4(273.16);
  if (!m->Coupling_failed) {
    if (m->Coupling_iterations == 0) m->Tsurf_end_coup1 = m->TsurfAve;
    if (m->Coupling_iterations == 25) {
      if (fabs(m->Tsurf_end_coup1 - m->lastTsurfObs) < fabs(m->TsurfAve - m->lastTsurfObs))
        m->start_coupling_again = 1;
      m->SwRadCof = R4(1.0); m->LwRadCof = R4(1.0);
      m->SW_correction = R4(0.0); m->LW_correction = R4(0.0);
      m->RadCoeff = R4(1.0);
      m->Coupling_failed = 1;
    } else if (m->lastTsurfObs < -100) {
      m->SwRadCof = R4(1.0); m->LwRadCof = R4(1.0);
      m->SW_correction = R4(0.0); m->LW_correction = R4(0.0);
      m->RadCoeff = R4(1.0);
      m->Coupling_failed = 1;
      m->start_coupling_again = 1;
    } else if (m->TsurfAve < R4(170.0) || m->TsurfAve > R4(400.0) || m->Coupling_failed) {
      m->SwRadCof = R4(1.0); m->LwRadCof = R4(1.0);
      m->SW_correction = R4(0.0); m->LW_correction = R4(0.0);
      m->Coupling_failed = 1;
      m->start_coupling_again = 1;
      m->RadCoeff = R4(1.0);
    } else if (m->TsurfAve - m->lastTsurfObs > R4(0.1)) {
      if (m->TsurfNearestAbove < -100) {
        m->TsurfNearestAbove = m->TsurfAve;
        m->RadCoefNearestAbove = m->RadCoeff;
      } else if (m->TsurfNearestAbove - m->lastTsurfObs > m->TsurfAve - m->lastTsurfObs) {
        m->TsurfNearestAbove = m->TsurfAve;
        m->RadCoefNearestAbove = m->RadCoeff;
      }
      m->start_coupling_again = 1;
      if (m->TsurfNearestAbove > -100 && m->TsurfNearestBelow > -100) {
        TDifAbove = m->TsurfNearestAbove - m->lastTsurfObs;
        TDifBelow = m->lastTsurfObs - m->TsurfNearestBelow;
        m->RadCoeff = m->RadCoefNearestAbove -
                      TDifAbove / (TDifAbove + TDifBelow) * (m->RadCoefNearestAbove - m->RadCoefNearestBelow);
      } else {
        m->RadCoeff = R4(0.5) * m->RadCoeff;
      }
      if (fabs(m->RadCoeff - m->RadCoeffPrevious) < R4(0.00005)) {
        m->TsurfNearestAbove = -9999;
        m->TsurfNearestBelow = -9999;
      }
      if (m->RadCoeff < R4(0.01)) {
        m->RadCoeff = R4(1.0);
        m->Coupling_failed = 1;
        m->SwRadCof = R4(1.0); m->LwRadCof = R4(1.0);
        m->SW_correction = R4(0.0); m->LW_correction = R4(0.0);
      }
      m->RadCoeffPrevious = m->RadCoeff;
    } else if (m->lastTsurfObs - m->TsurfAve > R4(0.1)) {
      if (m->TsurfNearestBelow < -100) {
        m->TsurfNearestBelow = m->TsurfAve;
        m->RadCoefNearestBelow = m->RadCoeff;
      } else if (m->TsurfNearestBelow - m->lastTsurfObs < m->TsurfAve - m->lastTsurfObs) {
        m->TsurfNearestBelow = m->TsurfAve;
        m->RadCoefNearestBelow = m->RadCoeff;
      }
      m->start_coupling_again = 1;
      if (m->TsurfNearestAbove > -100 && m->TsurfNearestBelow > -100) {
        TDifAbove = m->TsurfNearestAbove - m->lastTsurfObs;
        TDifBelow = m->lastTsurfObs - m->TsurfNearestBelow;
        m->RadCoeff = m->RadCoefNearestAbove -
                      TDifAbove / (TDifAbove + TDifBelow) * (m->RadCoefNearestAbove - m->RadCoefNearestBelow);
      } else {
        m->RadCoeff = R4(2.0) * m->RadCoeff;
      }
      if (fabs(m->RadCoeff - m->RadCoeffPrevious) < R4(0.00005)) {
        m->TsurfNearestAbove = -9999;
        m->TsurfNearestBelow = -9999;
      }
      m->RadCoeffPrevious = m->RadCoeff;
    } else {
      if (m->RadCoeff > R4(3.0)) {
        m->Coupling_failed = 1;
        m->RadCoeff = R4(1.0);
        m->SwRadCof = R4(1.0); m->LwRadCof = R4(1.0);
        m->SW_correction = R4(0.0); m->LW_correction = R4(0.0);
      }
      m->SW_correction = m->SwRadCof - R4(1.0);
      m->LW_correction = m->LwRadCof - R4(1.0);
      m->Coupling_failed = 0;
      m->Coupling_iterations = -1;
      m->TsurfNearestAbove = R4(-9999.0);
      m->TsurfNearestBelow = R4(-9999.0);
      m->RadCoeff = R4(1.0);
      m->RadCoefNearestAbove = R4(-9999.0);
      m->RadCoefNearestBelow = R4(-9999.0);
      m->RadCoeffPrevious = R4(1.0);
      /* CoupPhaseN < NObs never holds (NObs = 1) */
    }
  }
  m->TsurfAve = m->TsurfAve - R4(273.16);
  m->lastTsurfObs = m->lastTsurfObs - R4(273.16);
}

/* src/Coupling.f90:98-141 */
static void CheckEndCoupling(Model *m, int i) {
  if (m->use_coupling && i == m->couplingEndI && !m->Coupling_failed) {
    if (m->Coupling_iterations == 0) m->Tsurf_end_coup1 = m->TsurfAve;
    Coupling_control(m);
    m->Coupling_iterations = m->Coupling_iterations + 1;
  }
}

/* ---- sky view: src/SunPosition.f90, src/ModRadiation.f90 -------------------------- */

/* src/SunPosition.f90:196-260.  REAL() without a kind is REAL(4): the day fraction is
 * accumulated in single precision. */
static double JulianEphemerisDay(const InputPointers *in, int idx /*1-based*/) {
  const int mmyr = in->c_year[idx - 1], mmmon = in->c_month[idx - 1], mmday = in->c_day[idx - 1];
  const int mmhr = in->c_hour[idx - 1], mmmin = in->c_minute[idx - 1], mmsec = in->c_second[idx - 1];
  const double Dyr = R4(365.25);
  double yr, mo, A, B, day;
  if (mmmon <= 2) {
    yr = (double)(float)(mmyr - 1);
    mo = (double)(float)(mmmon + 12);
  } else {
    yr = (double)(float)mmyr;
    mo = (double)(float)mmmon;
  }
  {
    float d = (float)mmday + (float)mmhr / 24.f;
    d = d + (float)mmmin / (24.f * 60.f);
    d = d + (float)mmsec / (24.f * 60.f * 60.f);
    day = (double)d;
  }
  A = trunc(yr / R4(100.));
  B = R4(2.) - A + trunc(A / R4(4.));
  return trunc(Dyr * (yr + 4716)) + trunc(R4(30.6001) * (mo + R4(1.))) + day + B - 1.5245e3;
}

/* src/SunPosition.f90:20-194.  Returns 0, or 1 where the reference would `stop`. */
static int calcElevationAzimuth(double JDE, double lat, double lon, double *elevation_angle,
                                double *azimuth_angle) {
  const double pi = 4 * atan(1.0);
  const double Dyr = R4(365.25);
  double T, ml, ma, ecc, sunc, al, tilt, eps, ra, declination, stG;
  double cos_declination, sin_declination, lat_radians, sin_lat, cos_lat, cos_dec_lat, sin_dec_lat;
  double hour_angle_corr, cosah, cos_elev, chi, cosele, precos;
  T = (JDE - R4(2451545.0)) / (Dyr * R4(100.));
  ml = R4(280.46645) + R4(36000.76983) * T + R4(0.0003032) * T * T;
  if (ml < R4(0.)) ml = ml - R4(360.) * (trunc(ml / R4(360.)) - R4(1.));
  if (ml > R4(360.)) ml = ml - R4(360.) * trunc(ml / R4(360.));
  ma = R4(357.52910) + R4(35999.05030) * T - R4(0.0001559) * T * T - R4(0.00000048) * T * T * T;
  if (ma < R4(0.)) ma = ma - R4(360.) * (trunc(ma / R4(360.)) - R4(1.));
  if (ma > R4(360.)) ma = ma - R4(360.) * trunc(ma / R4(360.));
  ecc = R4(0.016708617) - R4(0.000042037) * T - R4(0.0000001236) * T * T;
  (void)ecc;
  sunc = (R4(1.913600) - R4(0.004817) * T - R4(0.000014) * T * T) * sin(ma * pi / R4(180.)) +
         (R4(0.019993) - R4(0.000101) * T) * sin(R4(2.) * ma * pi / R4(180.)) +
         R4(0.000290) * sin(R4(3.) * ma * pi / R4(180.));
  al = ml + sunc - R4(0.00569) - R4(0.00478) * sin((R4(125.04) - R4(1934.136) * T) * pi / R4(180.));
  al = al * pi / R4(180.);
  tilt = R4(23.43929111) - R4(0.013004166) * T - R4(0.001638888) * T * T + R4(0.005036111) * T * T * T;
  eps = tilt + R4(0.00256) * cos((R4(125.04) - R4(1934.136) * T) * pi / R4(180.));
  eps = eps * pi / R4(180.);
  ra = atan2(cos(eps) * sin(al), cos(al));
  if (ra < R4(0.)) ra = ra - R4(2.) * pi * (trunc(ra / (R4(2.) * pi)) - R4(1.));
  if (ra > R4(2.) * pi) ra = ra - R4(2.) * pi * trunc(ra / (R4(2.) * pi));
  declination = asin(sin(eps) * sin(al));
  stG = R4(280.46061837) + R4(360.98564736629) * (JDE - R4(2451545.0)) + R4(0.000387933) * T * T -
        T * T * T / R4(38710000.);
  if (stG < R4(0.)) stG = stG - R4(360.) * (trunc(stG / R4(360.)) - R4(1.));
  if (stG > R4(360.)) stG = stG - R4(360.) * trunc(stG / R4(360.));
  stG = stG * pi / R4(180.);
  cos_declination = cos(declination);
  sin_declination = sin(declination);
  lat_radians = pi * lat / R4(180.);
  sin_lat = sin(lat_radians);
  cos_lat = cos(lat_radians);
  cos_dec_lat = cos_declination * cos_lat;
  sin_dec_lat = sin_declination * sin_lat;
  hour_angle_corr = (stG + lon * pi / R4(180.) - ra);
  if (ra < R4(0.))
    hour_angle_corr = hour_angle_corr - R4(2.) * pi * (trunc(hour_angle_corr / (R4(2.) * pi)) - R4(1.));
  if (ra > R4(2.) * pi)
    hour_angle_corr = hour_angle_corr - R4(2.) * pi * trunc(hour_angle_corr / (R4(2.) * pi));
  cosah = cos(hour_angle_corr);
  cos_elev = sin_dec_lat + cos_dec_lat * cosah;
  if (cos_elev >= R4(1.0) && cos_elev < R4(1.001)) {
    chi = R4(0.);
  } else if (cos_elev >= R4(1.001)) {
    return 1;
  } else if (cos_elev > R4(-1.001) && cos_elev <= R4(-1.0)) {
    chi = pi;
  } else {
    chi = acos(cos_elev);
  }
  *elevation_angle = R4(90.0) - chi * (R4(180.) / pi);
  if (hour_angle_corr < R4(0.))
    hour_angle_corr = 2 * pi + hour_angle_corr;
  else if (hour_angle_corr > 2 * pi)
    hour_angle_corr = hour_angle_corr - 2 * pi;
  if (*elevation_angle > 0) {
    cosele = cos((pi / R4(2.0)) - chi);
    if (cosele >= R4(-0.0001) && cosele < R4(0.0001)) {
      *azimuth_angle = R4(-9999.9);
    } else {
      precos = (sin_declination * cos_lat - cos_declination * sin_lat * cosah) / cosele;
      if (precos >= R4(1.0) && precos < R4(1.001)) {
        *azimuth_angle = R4(0.0);
      } else if (precos >= R4(1.001)) {
        return 1;
      } else if (precos > R4(-1.001) && precos <= R4(-1.0)) {
        *azimuth_angle = pi;
      } else {
        *azimuth_angle = acos(precos);
      }
    }
    if (hour_angle_corr < pi) *azimuth_angle = 2 * pi - *azimuth_angle;
    *azimuth_angle = *azimuth_angle * (R4(180.) / pi);
  } else {
    *azimuth_angle = R4(-9999.9);
    *elevation_angle = R4(-9999.9);
  }
  return 0;
}

/* src/ModRadiation.f90:7-73: writes SW(i), SW_dir(i), LW(i) in the caller's arrays. */
static void ModRadiationBySurroundings(const InputPointers *in, const InputParameters *P,
                                       const LocalParameters *lp, int i) {
  const int k = i - 1;
  double sun_elevation = 0, sun_azim = 0, dif_SW, LW_surroundings, SW_ref, shadow_fac, horizon;
  int azim_idx;
  dif_SW = in->c_SW[k] - in->c_SW_dir[k];
  LW_surroundings = in->c_LW_net[k] - in->c_LW[k];
  if (calcElevationAzimuth(JulianEphemerisDay(in, i), lp->lat, lp->lon, &sun_elevation, &sun_azim)) {
    fprintf(stderr, "roadsurf_oracle: the reference would STOP here (zenith/azimuth problem)\n");
    abort();
  }
  azim_idx = (int)lround(sun_azim); /* NINT */
  if (azim_idx == 360) azim_idx = 0;
  /* the reference indexes local_horizons(azim_idx+1) also for the missing value -9999.9
   * (index -9999: out of bounds, whatever memory holds); the value is unused then because
   * sun_elevation is -9999.9 too.  We read nothing in that case. */
  horizon = (azim_idx >= 0 && azim_idx < 360) ? in->c_local_horizons[azim_idx] : R4(0.);
  shadow_fac = (horizon > sun_elevation) ? R4(0.0) : R4(1.0);
  if (sun_elevation > R4(0.0)) {
    in->c_SW_dir[k] = in->c_SW_dir[k] * shadow_fac;
    SW_ref = P->Albedo_surroundings * in->c_SW_dir[k] + P->Albedo_surroundings * dif_SW;
    dif_SW = lp->sky_view * dif_SW + (R4(1.0) - lp->sky_view) * SW_ref;
    in->c_SW[k] = dif_SW + in->c_SW_dir[k];
  }
  in->c_LW[k] = lp->sky_view * in->c_LW[k] + (R4(1.0) - lp->sky_view) * (-LW_surroundings);
}

/* examples/example1/src/Simulation.f90:120-172 */
static void roadModelOneStep(Model *m, const InputPointers *in, int i /*1-based*/) {
  PrecipitationToStorage(m, in->c_PrecPhase[i - 1]);
  if (m->skyview_on) ModRadiationBySurroundings(in, m->P, m->lp, i);
  BalanceModelOneStep(m, in->c_SW[i - 1], in->c_LW[i - 1], in->c_hour[i - 1],
                      in->c_Depth[i - 1]);
  WearFactors(m);
  RoadCond(m);
  CalcAlbedo(m);
}

/* src/InputOutput.f90:45-84 */
static void CheckValues(Model *m, const InputPointers *in, const LocalParameters *lp, int i) {
  const int k = i - 1;
  if (in->c_tair[k] < R4(-90.0) || in->c_tair[k] > R4(100.0) || in->c_tdew[k] < -90 ||
      in->c_tdew[k] > R4(100.0) || in->c_Rhz[k] < R4(-0.1) || in->c_Rhz[k] > R4(120.0) ||
      in->c_VZ[k] < R4(-1.0) || in->c_VZ[k] > R4(100.0) || in->c_SW[k] < R4(-0.1) ||
      in->c_SW[k] > R4(4000.0) || in->c_LW[k] < R4(-0.1) || in->c_LW[k] > R4(1000.0) ||
      in->c_prec[k] < R4(-0.1) || in->c_prec[k] > R4(500.0))
    m->failed = 1;
  if (lp->sky_view < R4(1.0) && lp->sky_view > R4(-0.01)) {
    if (in->c_SW_dir[k] < R4(-0.1) || in->c_SW_dir[k] > R4(4000.0) ||
        in->c_LW_net[k] < R4(-1000.0) || in->c_LW_net[k] > R4(1000.0))
      m->failed = 1;
  }
  if (in->c_SW_dir[k] > in->c_SW[k]) in->c_SW_dir[k] = in->c_SW[k];
  if (m->TsurfAve < R4(-100.0) || m->TsurfAve > R4(100.0)) m->failed = 1;
}

/* src/InputOutput.f90:86-149 */
static void SetCurrentValues(Model *m, const InputPointers *in, int i) {
  const int k = i - 1;
  m->Tair = in->c_tair[k];
  m->VZ = in->c_VZ[k];
  m->Rhz = in->c_Rhz[k];
  m->PrecInTStep = in->c_prec[k] / 3600 * m->DTSecs;
  m->Tmp[0] = m->Tair;
  if (i <= m->InitLenI || m->force_tsurf) {
    if (in->c_TSurfObs[k] > R4(-100.0) && (!m->use_coupling || i < m->couplingStartI)) {
      double depth;
      m->Tmp[1] = in->c_TSurfObs[k];
      m->Tmp[2] = in->c_TSurfObs[k];
      if (m->tsurfOutputDepth >= R4(0.0))
        depth = m->tsurfOutputDepth;
      else
        depth = in->c_Depth[k];
      if (depth >= 0)
        m->TsurfAve = getTempAtDepth(m, depth);
      else
        m->TsurfAve = (m->Tmp[1] + m->Tmp[2]) / R4(2.0);
    }
  }
}

/* src/Relaxation.f90:10-47 (the trailing CalcTDew result is never read) */
static void RelaxationOperations(Model *m, int i) {
  const double DTs = m->DTSecs;
  const int initLI = m->InitLenI;
  if (i == initLI) {
    m->TairInitEnd = m->Tair;
    m->VZInitEnd = m->VZ;
    m->RhzInitEnd = m->Rhz;
  }
  if (i > initLI) {
    const double den = (double)(4.f * 3600.f);
    m->Tair = m->Tair - (m->TairR - m->TairInitEnd) * exp(-((DTs * i) - (DTs * initLI)) / den);
    m->Tmp[0] = m->Tair;
    m->VZ = m->VZ - (m->VZR - m->VZInitEnd) * exp(-((DTs * i) - (DTs * initLI)) / den);
    m->Rhz = m->Rhz - (m->RhzR - m->RhzInitEnd) * exp(-((DTs * i) - (DTs * initLI)) / den);
    if (m->Rhz > R4(100.)) m->Rhz = R4(100.0);
  }
}

/* src/InputOutput.f90:151-165 */
static void SaveOutput(const Model *m, OutputPointers *out, int i) {
  out->c_SnowOut[i - 1] = m->Snow;
  out->c_WaterOut[i - 1] = m->Wat;
  out->c_IceOut[i - 1] = m->Ice;
  out->c_Ice2Out[i - 1] = m->Ice2;
  out->c_DepositOut[i - 1] = m->Dep;
  out->c_TsurfOut[i - 1] = m->TsurfAve;
}

/* Init products shared by every point with the same settings/parameters.
 * Exposed for known-answer tests. */
void oracle_init_tables(const InputSettings *s, const InputParameters *P, double *ZDpth /*N+2*/,
                        double *DyC, double *DyK, double *CC, double *condDZ, double *logs4) {
  const int N = s->NLayers;
  /* src/Initialization.f90:217-235 */
  const double ZAdd = R4(0.02);
  ZDpth[0] = 0.0;
  ZDpth[1] = R4(0.0);
  for (int I = 1; I <= N; ++I)
    ZDpth[I + 1] = ZDpth[I] + (double)(0.0103f * powi_f32(1.4f, I - 1)) + ZAdd;
  /* src/Initialization.f90:181-214 */
  DyC[1] = (ZDpth[2] - ZDpth[1]) / R4(2.0);
  for (int j = 2; j <= N; ++j) DyC[j] = (ZDpth[j + 1] - ZDpth[j - 1]) / R4(2.0);
  for (int j = 1; j <= N; ++j) DyK[j] = ZDpth[j + 1] - ZDpth[j];
  /* src/BalanceModel.f90:158-186, 254-279 */
  {
    const double Afc1 = R4(0.65) - R4(0.78) * P->RhoB1 + R4(0.60) * P->RhoB1 * P->RhoB1;
    const double Bfc1 = R4(1.06) * P->RhoB1;
    const double Cfc1 = (P->Silt1 > R4(0.00001)) ? 1 + R4(2.6) / sqrt(P->Silt1) : R4(0.);
    const double Dfc1 = R4(0.03) + R4(0.1) * P->RhoB1 * P->RhoB1;
    const double Afc2 = R4(0.65) - R4(0.78) * P->RhoB2 + R4(0.60) * P->RhoB2 * P->RhoB2;
    const double Bfc2 = R4(1.06) * P->RhoB2;
    const double Cfc2 = (P->Silt2 > R4(0.00001)) ? 1 + R4(2.6) / sqrt(P->Silt2) : R4(0.);
    const double Dfc2 = R4(0.03) + R4(0.1) * P->RhoB2 * P->RhoB2;
    const double Efc = 4;
    for (int I = 1; I <= N; ++I) {
      const double W = (I <= 2) ? R4(0.01) : R4(0.3);
      if (I <= 2)
        CC[I] = Afc1 + Bfc1 * W - (Afc1 - Dfc1) * exp(-pow(Cfc1 * W, Efc));
      else
        CC[I] = Afc2 + Bfc2 * W - (Afc2 - Dfc2) * exp(-pow(Cfc2 * W, Efc));
      condDZ[I] = -(CC[I] / DyK[I]);
    }
  }
  /* src/Initialization.f90:330-337 */
  logs4[0] = log((P->ZRefW + P->ZMom) / P->ZMom);
  logs4[1] = log((P->ZRefW + P->ZHeat) / P->ZHeat);
  logs4[2] = log((P->ZRefW - P->ZeroDisp + P->ZHeat) / P->ZHeat);
  logs4[3] = log((P->ZRefW - P->ZeroDisp + P->ZMom) / P->ZMom);
}

/* src/Initialization.f90:9-147 (+ :442-557, src/InputOutput.f90:4-39) */
static void Initialization(Model *m, OutputPointers *out, const InputPointers *in,
                           const InputSettings *s, const InputParameters *P,
                           const LocalParameters *lp) {
  double logs4[4];
  const int N = s->NLayers;
  memset(m, 0, sizeof(*m));
  /* initSettings :442-476 */
  m->SimLen = s->SimLen;
  m->InitLenI = lp->InitLenI;
  m->DTSecs = s->DTSecs;
  m->tsurfOutputDepth = s->tsurfOutputDepth;
  m->NLayers = N;
  m->NightOn = P->NightOn; m->NightOff = P->NightOff;
  m->CalmLimDay = P->CalmLimDay; m->CalmLimNgt = P->CalmLimNgt;
  m->TrfFricNgt = P->TrfFricNgt; m->TrFfricDay = P->TrFfricDay;
  m->use_coupling = (s->use_coupling == 1);
  m->use_relaxation = (s->use_relaxation == 1);
  m->force_tsurf = (s->force_tsurf == 1);
  /* initOutputArrays :397-412 */
  for (int i = 0; i < s->SimLen; ++i) {
    out->c_SnowOut[i] = R4(-9999.0); out->c_WaterOut[i] = R4(-9999.0);
    out->c_IceOut[i] = R4(-9999.0); out->c_Ice2Out[i] = R4(-9999.0);
    out->c_DepositOut[i] = R4(-9999.0); out->c_TsurfOut[i] = R4(-9999.0);
  }
  /* setInputParam, src/InputOutput.f90:4-39: relaxation targets go through REAL(4) */
  m->TairR = (double)(float)lp->tair_relax;
  m->VZR = (double)(float)lp->VZ_relax;
  m->RhzR = (double)(float)lp->RH_relax;
  if (m->TairR < R4(-100.0) || m->TairR > R4(100.0) || m->VZR < R4(0.0) ||
      m->VZR > R4(100.0) || m->RhzR < R4(0.0) || m->RhzR > 110)
    m->use_relaxation = 0;
  m->lastTsurfObs = lp->couplingTsurf;
  if (lp->couplingTsurf < -100 || lp->couplingIndexI < 1) m->use_coupling = 0;
  m->coupling_minutes = s->coupling_minutes;
  m->couplingEffectReduction = s->couplingEffectReduction;
  m->skyview_on = (lp->sky_view < R4(1.0) && lp->sky_view > R4(-0.01));
  m->P = P;
  m->lp = lp;
  /* initVariablesAndParameters :65-147 */
  m->failed = 0;
  m->Tph = m->DTSecs / R4(3600.0);
  oracle_init_tables(s, P, m->ZDpth, m->DyC, m->DyK, m->CC, m->condDZ, logs4);
  /* initSurf :290-308 */
  m->Q2Melt = R4(0.0); m->VeryCold = 0; m->WearSurf = 1; m->TrfFric = R4(5.0);
  m->Evap = R4(0.0); m->Wat = m->Snow = m->Ice = m->Ice2 = m->Dep = R4(0.0);
  /* InitParam :310-358 */
  m->Grav = P->Grav; m->SB = P->SB_Const; m->VK = P->VK_Const; m->ZRefT = P->ZRefT;
  m->logMom = logs4[0]; m->logHeat = logs4[1]; m->logCond = logs4[2]; m->logUstar = logs4[3];
  m->Emiss = P->Emiss; m->Albedo = P->Albedo; m->MaxPormms = P->MaxPormms;
  m->LVap = P->LVap; m->LFus = P->LFus;
  m->vsh1 = P->vsh1; m->vsh2 = P->vsh2; m->Poro1 = P->Poro1; m->Poro2 = P->Poro2;
  /* initTemp :238-287 */
  {
    const double Tsurf = in->c_TSurfObs[0], Tair = in->c_tair[0];
    int juld;
    m->Tmp[0] = Tair;
    for (int i = 1; i <= 4; ++i) m->Tmp[i] = (Tsurf > -100) ? Tsurf : Tair;
    juld = JulDay(in->c_year[0], in->c_month[0], in->c_day[0]);
    m->Tmp[N + 1] = P->TClimG + P->AZ * sin(P->Omega * juld + P->Omega * (-170) -
                                              (m->ZDpth[N + 1] / P->DampDpth));
    for (int i = 5; i <= N; ++i)
      m->Tmp[i] = m->Tmp[4] + (m->Tmp[N + 1] - m->Tmp[4]) / (m->ZDpth[N + 1] - m->ZDpth[4]) *
                                  (m->ZDpth[i] - m->ZDpth[4]);
    memcpy(m->TmpNw, m->Tmp, sizeof(double) * (N + 2));
  }
  /* initVariables :361-394 (values that can matter) */
  m->BLCond = R4(-99.9); m->TairInitEnd = R4(-99.9); m->VZInitEnd = R4(-99.9);
  m->RhzInitEnd = R4(-99.9); m->SnowType = SNOW_DRY; m->CalmLim = R4(0.4);
  /* ground_prop_init :207-213 */
  for (int I = 1; I <= N; ++I) m->WCont[I] = (I <= 2) ? R4(0.01) : R4(0.3);
  CalcHCapHCond(m);
  calcCapDZCondDZ(m);
  /* initCoupling src/Coupling.f90:144-169 */
  m->Coupling_iterations = 0;
  m->TsurfNearestAbove = R4(-9999.0); m->TsurfNearestBelow = R4(-9999.0);
  m->RadCoeff = R4(1.0);
  m->RadCoefNearestAbove = R4(-9999.0); m->RadCoefNearestBelow = R4(-9999.0);
  m->RadCoeffPrevious = R4(1.0);
  m->SwRadCof = R4(1.0); m->LwRadCof = R4(1.0);
  m->start_coupling_again = 0; m->Coupling_failed = 0;
  m->SW_correction = R4(0.0); m->LW_correction = R4(0.0);
  m->inCouplingPhase = 0;
  /* initCouplingTimes src/Coupling.f90:486-534 */
  m->couplingStartI = -99; m->couplingEndI = -99;
  if (m->use_coupling && lp->couplingIndexI > -1) {
    m->couplingEndI = lp->couplingIndexI;
    if (lp->couplingIndexI <= (m->coupling_minutes * 60) / m->DTSecs)
      m->couplingStartI = 1;
    else
      m->couplingStartI = lp->couplingIndexI - (int)((m->coupling_minutes * 60) / m->DTSecs);
  } else {
    m->use_coupling = 0;
  }
  /* condInit :479-557 */
  m->WatDens = P->WatDens; m->WatMHeat = P->WatMHeat; m->PorEvaF = P->PorEvaF;
  m->DampWearF = P->DampWearF;
  m->TLimFreeze = P->freezing_limit_normal; m->TLimMeltSnow = P->snow_melting_limit_normal;
  m->TLimMeltIce = P->ice_melting_limit_normal; m->TLimMeltDep = P->frost_melting_limit_normal;
  m->TLimDew = P->frost_formation_limit_normal; m->T4Melt = P->T4Melt_normal;
  m->TLimColdH = P->TLimColdH; m->TLimColdL = P->TLimColdL;
  m->WetSnowFormR = P->WetSnowFormR; m->WetSnowMeltR = P->WetSnowMeltR;
  m->PLimSnow = P->PLimSnow; m->PLimRain = P->PLimRain;
  m->MinPrecmm = P->MinPrecmm; m->MinWatmms = P->MinWatmms; m->MinSnowmms = P->MinSnowmms;
  m->MinDepmms = P->MinDepmms; m->MinIcemms = P->MinIcemms;
  m->MaxSnowmms = P->MaxSnowmms; m->MaxDepmms = P->MaxDepmms; m->MaxIcemms = P->MaxIcemms;
  m->MaxWatmms = P->MaxWatmms; m->AlbDry = P->AlbDry; m->AlbSnow = P->AlbSnow;
  m->MissValI = P->MissValI; m->WWetLim = P->WWetLim; m->WWearLim = P->WWearLim;
  m->Snow2IceFac = P->Snow2IceFac;
  /* :121-123 — writes into the caller's array */
  if (in->c_VZ[0] < R4(0.4)) in->c_VZ[0] = R4(0.4);
  m->Tair = in->c_tair[0];
  m->VZ = in->c_VZ[0];
  m->Rhz = in->c_Rhz[0];
  if (in->c_Depth[0] >= 0)
    m->TsurfAve = getTempAtDepth(m, in->c_Depth[0]);
  else
    m->TsurfAve = (m->Tmp[1] + m->Tmp[2]) / R4(2.0);
  CalcBLCondAndLE(m); /* :138-139 */
  if (m->lastTsurfObs < -100) m->Coupling_failed = 1; /* :142-144 */
}

/* examples/example1/src/Simulation.f90:4-117 */
void runsimulation(OutputPointers *out, const InputPointers *in, const InputSettings *s,
                   const InputParameters *P, const LocalParameters *lp) {
  Model M, *m = &M;
  int i;
  if (s->NLayers < 5 || s->NLayers > MAXL) {
    fprintf(stderr, "roadsurf_oracle: NLayers out of range\n");
    abort();
  }
  Initialization(m, out, in, s, P, lp);
  i = 1;
  while (i < m->SimLen && !m->failed) {
    CheckValues(m, in, lp, i);
    if (m->use_coupling) CouplingOperations1(m, in, &i);
    SetCurrentValues(m, in, i);
    if (m->use_relaxation) RelaxationOperations(m, i);
    roadModelOneStep(m, in, i);
    SaveOutput(m, out, i);
    CheckEndCoupling(m, i);
    i = i + 1;
  }
  if (!m->failed) {
    /* lastValues, src/InputOutput.f90:169-198 */
    const int L = m->SimLen;
    double depth;
    m->Tair = in->c_tair[L - 1];
    m->VZ = in->c_VZ[L - 1];
    m->Rhz = in->c_Rhz[L - 1];
    m->PrecInTStep = in->c_prec[L - 1] / 3600 * m->DTSecs;
    m->Tmp[0] = m->Tair;
    depth = in->c_Depth[L - 1];
    if (depth >= 0)
      m->TsurfAve = getTempAtDepth(m, depth);
    else
      m->TsurfAve = (m->Tmp[1] + m->Tmp[2]) / R4(2.0);
    roadModelOneStep(m, in, L);
    SaveOutput(m, out, i);
  }
  free(m->SWSave); free(m->SWDirSave); free(m->LWSave);
}

/* ---- known-answer probes (tests only) ---------------------------------- */

/* Products of Initialization; same signature as ref_probe_init (oracle/ref_probe.f90). */
void oracle_probe_init(const InputPointers *in, OutputPointers *out, const InputSettings *s,
                       const InputParameters *P, const LocalParameters *lp, double *zdpth,
                       double *dyc, double *dyk, double *cc, double *conddz, double *tmp,
                       double *logs, double *tsurfave) {
  Model M;
  Initialization(&M, out, in, s, P, lp);
  for (int i = 1; i <= s->NLayers + 1; ++i) zdpth[i - 1] = M.ZDpth[i];
  for (int i = 1; i <= s->NLayers; ++i) {
    dyc[i - 1] = M.DyC[i]; dyk[i - 1] = M.DyK[i]; cc[i - 1] = M.CC[i]; conddz[i - 1] = M.condDZ[i];
  }
  for (int i = 0; i <= s->NLayers + 1; ++i) tmp[i] = M.Tmp[i];
  logs[0] = M.logMom; logs[1] = M.logHeat; logs[2] = M.logCond; logs[3] = M.logUstar;
  *tsurfave = M.TsurfAve;
}

/* One CalcBLCondAndLE call on default-initialised parameters. */
void oracle_probe_blcond(const InputSettings *s, const InputParameters *P, double TsurfAve,
                         double Tair, double VZ, double Rhz, double SrfWatmms, double *BLCond,
                         double *LE, double *Evap, int *iters) {
  Model M;
  double Z[MAXL + 2], DyC[MAXL + 2], DyK[MAXL + 2], CC[MAXL + 2], cdz[MAXL + 2], logs4[4];
  memset(&M, 0, sizeof(M));
  oracle_init_tables(s, P, Z, DyC, DyK, CC, cdz, logs4);
  M.DTSecs = s->DTSecs;
  M.VK = P->VK_Const; M.ZRefT = P->ZRefT; M.Grav = P->Grav; M.LVap = P->LVap; M.LFus = P->LFus;
  M.logMom = logs4[0]; M.logHeat = logs4[1]; M.logCond = logs4[2]; M.logUstar = logs4[3];
  M.TsurfAve = TsurfAve; M.Tair = Tair; M.VZ = VZ; M.Rhz = Rhz; M.Wat = SrfWatmms;
  M.BLCond = R4(-99.9);
  *iters = CalcBLCondAndLE(&M);
  *BLCond = M.BLCond; *LE = M.LE; *Evap = M.Evap;
}

/* y[i] = libm exp (fn 0) / log (fn 1) of x[i]: the very functions the reference
 * calls.  (numpy may use its own SIMD exp/log, so tests go through this.) */
void oracle_libm_map(int fn, long n, const double *x, double *y) {
  for (long i = 0; i < n; ++i) y[i] = fn == 0 ? exp(x[i]) : log(x[i]);
}

/* Solar elevation/azimuth for one time stamp and location (known-answer tests). */
int oracle_probe_sun(int year, int month, int day, int hour, int minute, int second, double lat,
                     double lon, double *elevation, double *azimuth, double *jde) {
  int32_t y = year, mo = month, d = day, h = hour, mi = minute, se = second;
  InputPointers in;
  memset(&in, 0, sizeof(in));
  in.c_year = &y; in.c_month = &mo; in.c_day = &d; in.c_hour = &h; in.c_minute = &mi; in.c_second = &se;
  *jde = JulianEphemerisDay(&in, 1);
  return calcElevationAzimuth(*jde, lat, lon, elevation, azimuth);
}

/* Host check of the product's uniform-denominator division (roadsurf_amd/csrc/rs_math.hpp,
 * rs_div_u): q0 = a*rb, rem = fma(-b, q0, a), q = fma(rem, rb, q0) with rb = RN(1/b) must be
 * the IEEE quotient a/b.  libm's fma is correctly rounded, like v_fma_f64.  Returns how many of
 * the n numerators disagree (test infrastructure, tests/test_host_logic.py). */
long oracle_div_u_mismatches(double b, const double *a, long n) {
  const double rb = 1.0 / b;
  long bad = 0;
  for (long i = 0; i < n; ++i) {
    const double q0 = a[i] * rb;
    const double rem = fma(-b, q0, a[i]);
    const double q = fma(rem, rb, q0);
    const double want = a[i] / b;
    if (memcmp(&q, &want, sizeof q) != 0 && !(q != q && want != want)) ++bad;
  }
  return bad;
}
