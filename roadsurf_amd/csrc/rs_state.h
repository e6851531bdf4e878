/*
 * rs_state.h — layout of the per-point carried state in HBM.
 *
 * state[slot * npoints_padded + p], doubles.  Slots 0..RS_MAX_LAYERS-1 hold the
 * ground temperature profile Tmp(1..NLayers) (reference: ground%Tmp,
 * src/GroundVariables.f90.inc); the named slots hold what SURVEY.md Appendix B
 * lists as the state that must persist across time steps.  Everything else the
 * reference keeps in its derived types is either constant (RsConstants) or
 * recomputed every step.
 */
#ifndef RS_STATE_H
#define RS_STATE_H

#include "../../include/roadsurf.h"

enum RsStateSlot {
  RS_ST_TMP0 = 0,                /* Tmp(1) ... Tmp(NLayers) at slots 0..NLayers-1 */
  RS_ST_TNW1 = RS_MAX_LAYERS,    /* TmpNw(1): differs from Tmp(1) only after obs forcing */
  RS_ST_TNW2,                    /* TmpNw(2)   (src/InputOutput.f90:122-124 vs BalanceModel.f90:215) */
  RS_ST_TSURF,                   /* surf%TsurfAve */
  RS_ST_WAT,                     /* surf%SrfWatmms */
  RS_ST_SNOW,                    /* surf%SrfSnowmms */
  RS_ST_ICE,                     /* surf%SrfIcemms */
  RS_ST_ICE2,                    /* surf%SrfIce2mms */
  RS_ST_DEP,                     /* surf%SrfDepmms */
  RS_ST_Q2MELT,                  /* surf%Q2Melt */
  RS_ST_T4MELT,                  /* surf%T4Melt */
  RS_ST_ALBEDO,                  /* ground%Albedo */
  RS_ST_VERYCOLD,                /* surf%VeryCold as 0.0/1.0 */
  RS_ST_FAILED,                  /* settings%simulation_failed: 0.0, or the (1-based) time index at which
                                    CheckValues raised it (the step of that index still runs, the loop
                                    exits after it: examples/example1/src/Simulation.f90:58) */
  RS_ST_TAIR_END,                /* atm%TairInitEnd (relaxation) */
  RS_ST_VZ_END,                  /* atm%VZInitEnd */
  RS_ST_RH_END,                  /* atm%RhzInitEnd */
  RS_ST_BLSCORE,                 /* not model state: sort key of rs_hip_recluster (bl_score_key,
                                    rs_kernels.hip): bits 0-18 boundary-layer passes beyond the
                                    mandatory 5 during the last launch (saturating), bit 19 something
                                    lies on the road, bit 20 the point was in the unstable regime near
                                    the end of the launch; rs_cluster.hip sorts these 21 bits */
  /* ---- coupling (src/CouplingVariables.f90.inc); touched only by the coupled kernel ---- */
  RS_ST_CPL_ITER,                /* Coupling_iterations */
  RS_ST_CPL_FLAGS,               /* bit0 start_coupling_again, bit1 Coupling_failed, bit2 VeryColdSave, bits 3-4 what
                                    Coupling_control has printed for the point (RS_CPL_MSG_*) */
  RS_ST_CPL_TABOVE,              /* TsurfNearestAbove */
  RS_ST_CPL_TBELOW,              /* TsurfNearestBelow */
  RS_ST_CPL_RADCOEFF,            /* RadCoeff */
  RS_ST_CPL_RCABOVE,             /* RadCoefNearestAbove */
  RS_ST_CPL_RCBELOW,             /* RadCoefNearestBelow */
  RS_ST_CPL_RCPREV,              /* RadCoeffPrevious */
  RS_ST_CPL_SWCOF,               /* SWRadCof */
  RS_ST_CPL_LWCOF,               /* LWRadCof */
  RS_ST_CPL_SWCORR,              /* SW_correction */
  RS_ST_CPL_LWCORR,              /* LW_correction */
  RS_ST_CPL_TEND1,               /* Tsurf_end_coup1 */
  RS_ST_CPL_LASTOBS,             /* lastTsurfObs */
  RS_ST_CPL_RESUME,              /* next time index the point will step (the rounds of rs_hip_step with
                                    coupling: a point parks behind its coupling window until its replays
                                    are through) */
  RS_ST_CPL_SAVE_TSURF,          /* TSurfAveSave */
  RS_ST_CPL_SAVE_WAT,            /* SrfWatmmsSave */
  RS_ST_CPL_SAVE_ICE2,           /* SrfIce2mmsSave (SrfIcemms is never saved, src/Coupling.f90:194-195) */
  RS_ST_CPL_SAVE_DEP,            /* SrfDepmmsSave */
  RS_ST_CPL_SAVE_SNOW,           /* SrfSnowmmsSave */
  RS_ST_CPL_SAVE_ALBEDO,         /* AlbedoSave */
  RS_ST_CPL_SAVE_TMP0,           /* TmpSave(1..NLayers) at the next RS_MAX_LAYERS slots */
  RS_ST_CPL_STALE_TMP0 = RS_ST_CPL_SAVE_TMP0 + RS_MAX_LAYERS,
                                 /* TmpNw as a restore leaves it: the profile of the END of the
                                    window (Tmp is restored, TmpNw is not: src/Coupling.f90:245-247),
                                    read by the first step of a replay only; RS_MAX_LAYERS slots */
  RS_NSTATE = RS_ST_CPL_STALE_TMP0 + RS_MAX_LAYERS
};

/* the reference's messages from Coupling_control (src/Coupling.f90:400-401,451-452), kept as bits of
 * RS_ST_CPL_FLAGS for rs_hip_diagnostics */
#define RS_CPL_MSG_SMALL 8  /* "coupling coefficient too small, coupling failed" */
#define RS_CPL_MSG_BIG 16   /* "coupling coefficient too big, coupling failed" */

/* Rows of a plan's diagnostics block (rs_hip_set_diagnostics: [RS_DIAG_ROWS][np_pad] doubles, by slot): what the
 * reference prints from CalcBLCondAndLE (src/BoundaryLayer.f90:69-74,98-101), first occurrence and count per point. */
enum RsDiagRow {
  RS_DG_MAXIT_N = 0, /* "Max number of BLCond iterations": how many time indices printed it */
  RS_DG_MAXIT_I,     /* ... the first of them (1-based), */
  RS_DG_MAXIT_J,     /* its j (MaxIter + 1: the DO loop ran out), */
  RS_DG_MAXIT_OLD,   /* BLCond_Old */
  RS_DG_MAXIT_BL,    /* and BLCond */
  RS_DG_USTAR_N,     /* "ERROR : UStar negative": how many passes printed it */
  RS_DG_USTAR_I,     /* the time index of the first, */
  RS_DG_USTAR_TAIR,  /* and the values of its two lines: vz, then Tair, VZ, Rhz, BLCond, TSurfAve */
  RS_DG_USTAR_VZ,
  RS_DG_USTAR_RHZ,
  RS_DG_USTAR_BL,
  RS_DG_USTAR_TSURF,
  RS_DIAG_ROWS
};
/* rs_hip_diagnostics: the rows above per point, then the coupling messages (RS_CPL_MSG_* bits) */
#ifdef __cplusplus
static_assert(RS_DIAG_COLS == RS_DIAG_ROWS + 1, "include/roadsurf.h: RS_DIAG_COLS");
#endif

#endif
