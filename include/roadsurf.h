/*
 * roadsurf.h — C-ABI of the MI355X-native RoadSurf hot path.
 *
 * Everything here is plain C: pointers, sizes, PODs.  No torch types, no C++.
 * Citations are file:line into the reference tree (fmidev/RoadSurf v1.6.1).
 *
 * Three layers, bottom up:
 *   1. rs_hip_*   device-resident SoA API (HIP kernels, streams).  This is what
 *                 the Fortran host orchestration binds through ISO_C_BINDING
 *                 and what bench.py drives through ctypes.
 *   2. rs_host_*  host-array convenience entry (pack -> H2D -> kernels -> D2H),
 *                 used by the Fortran `runsimulation_batch`.
 *   3. runsimulation / runsimulation_batch   the reference's own BIND(C) entry
 *                 (examples/example1/src/Simulation.f90:4-6, declared on the
 *                 C++ side at examples/example1/src/roadrunner.cpp:22-29) and
 *                 its batched extension.  Implemented in Fortran
 *                 (roadsurf_amd/fortran/RoadSurfHip.f90) on top of layers 1-2.
 */
#ifndef ROADSURF_H
#define ROADSURF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------
 * The five Bind(C) structs of the reference boundary, byte-identical.
 * ---------------------------------------------------------------------- */

/* src/InputPointers.f90.inc:4-27 == examples/example1/src/InputPointers.h:7-30
 * sizeof == 160.  The reference declares the arrays `const` but mutates VZ[0]
 * (src/Initialization.f90:121-123) and SW_dir[i] (src/InputOutput.f90:75-77);
 * so do we, hence no const here. */
typedef struct InputPointers {
  int32_t inputLen;
  double *c_tair;
  double *c_tdew;
  double *c_VZ;
  double *c_Rhz;
  double *c_prec;
  double *c_SW;
  double *c_LW;
  double *c_SW_dir;
  double *c_LW_net;
  double *c_TSurfObs;
  int32_t *c_PrecPhase;
  double *c_local_horizons; /* [360] */
  double *c_Depth;
  int32_t *c_year;
  int32_t *c_month;
  int32_t *c_day;
  int32_t *c_hour;
  int32_t *c_minute;
  int32_t *c_second;
} InputPointers;

/* src/OutputPointers.f90.inc:4-17 == examples/example1/src/OutputPointers.h:5-16
 * sizeof == 56 */
typedef struct OutputPointers {
  int32_t outputLen;
  double *c_TsurfOut;
  double *c_SnowOut;
  double *c_WaterOut;
  double *c_IceOut;
  double *c_DepositOut;
  double *c_Ice2Out;
} OutputPointers;

/* Fortran view: src/InputSettings.f90.inc:4-18.  The C++ struct
 * (examples/example1/src/InputSettings.h:13-26) has no force_tsurf member; the
 * Fortran side reads the padding at offset 12.  We expose the Fortran view
 * (the one that is actually read) and keep the C++ tail out of it. */
typedef struct InputSettings {
  int32_t SimLen;
  int32_t use_coupling;
  int32_t use_relaxation;
  int32_t force_tsurf;
  double DTSecs;
  double tsurfOutputDepth;
  int32_t NLayers;
  int32_t coupling_minutes;
  double couplingEffectReduction;
  int32_t outputStep;
} InputSettings;

/* src/InputParameters.f90.inc:4-91 == examples/example1/src/InputParameters.h:18-110
 * 69 consecutive doubles (552 B). */
typedef struct InputParameters {
  double NightOn, NightOff, CalmLimDay, CalmLimNgt, TrfFricNgt, TrFfricDay;
  double Grav, SB_Const, VK_Const, LVap, LFus, WatDens, SnowDens, IceDens,
      DepDens, WatMHeat, PorEvaF;
  double ZRefW, ZRefT, ZeroDisp, ZMom, ZHeat, Emiss, Albedo,
      Albedo_surroundings, MaxPormms, TClimG, DampDpth, Omega, AZ, DampWearF,
      AlbDry, AlbSnow, vsh1, vsh2, Poro1, Poro2, RhoB1, RhoB2, Silt1, Silt2;
  double freezing_limit_normal, snow_melting_limit_normal,
      ice_melting_limit_normal, frost_melting_limit_normal,
      frost_formation_limit_normal, T4Melt_normal;
  double TLimColdH, TLimColdL, WetSnowFormR, WetSnowMeltR;
  double PLimSnow, PLimRain, MaxSnowmms, MaxDepmms, MaxIcemms, MaxExtmms;
  double MissValI, MissValR;
  double Snow2IceFac;
  double MinPrecmm, MinWatmms, MinSnowmms;
  double MaxWatmms;
  double WDampLim, WWetLim;
  double WWearLim;
  double MinDepmms, MinIcemms;
} InputParameters;

/* src/LocalParameters.f90.inc:4-15 == examples/example1/src/LocalParameters.h:17-25
 * sizeof == 72 */
typedef struct LocalParameters {
  double tair_relax;
  double VZ_relax;
  double RH_relax;
  int32_t couplingIndexI;
  double couplingTsurf;
  double lat;
  double lon;
  double sky_view;
  int32_t InitLenI;
} LocalParameters;

/* Fill *p with the reference defaults (examples/example1/src/InputParameters.h:18-94)
 * and the DTSecs-derived limits (examples/example1/src/InputParameters.cpp:13-21). */
void rs_default_parameters(InputParameters *p, double DTSecs);
/* examples/example1/src/InputSettings.h:13-23 defaults, force_tsurf = 0. */
void rs_default_settings(InputSettings *s, int32_t SimLen);
/* examples/example1/src/LocalParameters.h:17-25 defaults. */
void rs_default_local(LocalParameters *l);

/* ------------------------------------------------------------------------
 * Layer 3: the reference entry point and its batched extension (Fortran).
 * ---------------------------------------------------------------------- */

/* Drop-in for examples/example1/src/Simulation.f90:4-117.  One point, whole
 * time series, synchronous, no return value; failure leaves -9999.0 in the
 * unwritten outputs (src/Initialization.f90:404-411).  Runs on the GPU. */
void runsimulation(OutputPointers *outPointers, const InputPointers *inPointers,
                   const InputSettings *inSettings,
                   const InputParameters *inputParam,
                   const LocalParameters *localParam);

/* Callers of runsimulation on several threads at once (the reference driver's worker pool,
 * examples/example1/src/roadrunner.cpp:454-497): with the environment variable
 * ROADSURF_HIP_COALESCE_US = w > 0 their points are gathered - the first caller waits up to w
 * microseconds (or for ROADSURF_HIP_COALESCE_MAX callers, default 4096) and runs everything queued with
 * its settings and parameters as ONE batch; the others sleep until their point is done.  Same bits as
 * calls of their own.  rs_coalesce_run is what runsimulation calls (returns the batch status),
 * rs_coalesce_stats the batches run and points served so far. */
int32_t rs_coalesce_run(OutputPointers *outPointers, const InputPointers *inPointers,
                        const InputSettings *inSettings, const InputParameters *inputParam,
                        const LocalParameters *localParam);
void rs_coalesce_stats(int64_t *batches, int64_t *points);
/* What rs_coalesce_run hands its gathered points to: runsimulation_batch with the reference's in-place input
 * edits written back by default (below).  Exported for that caller only. */
void rs_runsimulation_gathered(int32_t n, OutputPointers *outPointers, const InputPointers *inPointers,
                               const InputSettings *inSettings, const InputParameters *inputParam,
                               const LocalParameters *localParam, int32_t *status);

/* Extension: n independent points with shared settings/parameters, one
 * OutputPointers/InputPointers/LocalParameters per point (exactly what the
 * reference driver builds per point, examples/example1/src/roadrunner.cpp:404-406).
 * status: 0 ok, <0 error (rs_last_error() has the text). */
void runsimulation_batch(int32_t n, OutputPointers *outPointers,
                         const InputPointers *inPointers,
                         const InputSettings *inSettings,
                         const InputParameters *inputParam,
                         const LocalParameters *localParam, int32_t *status);
/* Same, plus what the reference only prints: first_failed[n] (or NULL) receives per point 0, or the
 * 1-based time index at which CheckValues failed it ("BAD input value!" / "Abnormal surface
 * temperature", src/InputOutput.f90:63-65,80-81); outputs after that index read -9999.0.
 * In-place edits of the INPUT arrays (Ownership): the reference clamps SW_dir to SW at
 * every checked index (src/InputOutput.f90:75-77) and, with sky view, rewrites SW / SW_dir / LW
 * (src/ModRadiation.f90:57-71) in the CALLER's arrays.  Who writes them back:
 *   runsimulation                      by default (it is the reference's entry and leaves the caller's
 *                                      arrays as the reference does); ROADSURF_HIP_WRITEBACK=0 opts out
 *   runsimulation_batch / _batch_ex    only with ROADSURF_HIP_WRITEBACK=1: three more arrays per point
 *                                      back over PCIe, which a batch caller rarely reads
 * The VZ(1) >= 0.4 edit of src/Initialization.f90:121-123 is always written, by every entry. */
void runsimulation_batch_ex(int32_t n, OutputPointers *outPointers,
                            const InputPointers *inPointers,
                            const InputSettings *inSettings,
                            const InputParameters *inputParam,
                            const LocalParameters *localParam, int32_t *status,
                            int32_t *first_failed);

/* ------------------------------------------------------------------------
 * Model constants: everything that is uniform over points once settings and
 * parameters are shared.  Built on the HOST by Fortran (same compiler, same
 * REAL(4)-literal semantics as the reference's init code) and uploaded once.
 * Restates src/Initialization.f90:181-235,310-358,479-557,
 * src/BalanceModel.f90:132-186,254-279 and the REAL(4) folds of
 * src/Cond.f90:78-102.
 * ---------------------------------------------------------------------- */
#define RS_MAX_LAYERS 32

typedef struct RsConstants {
  int32_t NLayers;
  int32_t SimLen;
  int32_t use_relaxation;
  int32_t force_tsurf;
  int32_t use_coupling;    /* settings%use_coupling; per point it is switched off without a
                              usable observation (src/InputOutput.f90:34-36) */
  int32_t cplLenI;         /* int(coupling_minutes*60/DTs)   src/Coupling.f90:516-517 */
  double cplLenR;          /* coupling_minutes*60/DTs        src/Coupling.f90:512 */
  double cplReduction;     /* couplingEffectReduction        src/Coupling.f90:84-87 */
  double DTSecs;
  double Tph;              /* DTSecs/3600.0  src/Initialization.f90:92 */
  double tsurfOutputDepth; /* <0: use depth(i) */
  double twoDT;            /* 2.0*DTSecs, divisor of HS  src/BalanceModel.f90:241 */
  /* layer tables, 1-based like the reference (index 0 unused) */
  double ZDpth[RS_MAX_LAYERS + 2];  /* 1..NLayers+1  src/Initialization.f90:217-235 */
  double DyC[RS_MAX_LAYERS + 2];    /* 1..NLayers    src/Initialization.f90:193-196 */
  double condDZ[RS_MAX_LAYERS + 2]; /* 1..NLayers  -(CC/DyK): constant in time because
                                       CC and DyK never change (src/BalanceModel.f90:145,150) */
  double WCont[RS_MAX_LAYERS + 2];  /* 1..NLayers    src/Initialization.f90:207-213 */
  double dryCap[RS_MAX_LAYERS + 2]; /* 1..NLayers  (1.0-Poro)*vsh of the layer
                                       (src/BalanceModel.f90:233,235) */
  double HSfac1;                    /* ZDpth(2)-ZDpth(1), HS(1)  src/BalanceModel.f90:240 */
  /* boundary layer  src/Initialization.f90:330-337 */
  double logMom, logHeat, logCond, logUstar;
  double VK_Const, ZRefT, Grav, LVap, LFus;
  /* radiation */
  double Emiss, SB_Const, Albedo0;
  /* day/night  src/Initialization.f90:461-467 */
  double NightOn, NightOff, CalmLimDay, CalmLimNgt, TrfFricNgt, TrFfricDay;
  /* storage parameters: the RoadCondParameters members the path reads
   * (src/Initialization.f90:479-557) */
  double MaxPormms, MissValI, MinPrecmm, MinWatmms, MinSnowmms, MinDepmms,
      MinIcemms, MaxSnowmms, MaxDepmms, MaxIcemms, MaxWatmms, AlbDry, AlbSnow,
      WatDens, WatMHeat, PorEvaF, DampWearF, TLimFreeze, TLimMeltSnow,
      TLimMeltIce, TLimMeltDep, TLimDew, TLimColdH, TLimColdL, WetSnowFormR,
      WetSnowMeltR, PLimSnow, PLimRain, WWetLim, WWearLim, T4Melt0;
  /* REAL(4)-folded wear constants  src/Cond.f90:78-102:
   * (0.2+0.25), 0.25/(0.2+0.25), 1.1*2.0*0.145, 1.1*2.0*(4.0*0.290),
   * 0.5*2.0*(4.0*0.290), 0.145 */
  double wSnowTran, wSnow2Ice, wIce, wIce2, wDep, wWat;
} RsConstants;

/* Fortran (roadsurf_amd/fortran/RoadSurfHip.f90).  status 0 ok. */
void rs_build_constants(const InputSettings *inSettings,
                        const InputParameters *inputParam, RsConstants *out,
                        int32_t *status);
/* Time-only part of the solar position (src/SunPosition.f90:196-260 and :70-121,124-125):
 * for n time stamps writes table[k*RS_SUN_COLS + {0,1,2,3}] = apparent right ascension (rad, in
 * [0,2pi]), mean sidereal time at Greenwich (rad), sin and cos of the declination, and (ABI 5)
 * {4,5} = cos and sin of (sidereal time - right ascension): the device forms the cosine of the hour
 * angle of a point from them and the point's longitude by the addition theorem instead of a cosine
 * per point-step.  Fortran, host libm. */
#define RS_SUN_COLS 6
void rs_sun_table(int32_t n, const int32_t *year, const int32_t *month,
                  const int32_t *day, const int32_t *hour, const int32_t *minute,
                  const int32_t *second, double *table);
/* Per-point geometry (src/SunPosition.f90:126-128,133): sin/cos of the latitude and the
 * longitude in radians, evaluated as the reference does.  Fortran, host libm. */
void rs_point_geometry(int32_t n, const LocalParameters *localParam,
                       double *sin_lat, double *cos_lat, double *lon_rad);

/* Bottom boundary temperature Tmp(NLayers+1) for a start date
 * (src/Initialization.f90:266-268, src/BalanceModel.f90:325-351). Fortran. */
double rs_bottom_temperature(const InputParameters *inputParam,
                             const RsConstants *consts, int32_t year,
                             int32_t month, int32_t day);

/* ------------------------------------------------------------------------
 * Layer 1: device-resident SoA API.
 *
 * Layout in HBM.  Points are the fastest axis everywhere:
 *   forcing / outputs:  field[t * t_stride + p],  t_stride >= npoints
 *   carried state:      state[v * npoints_padded + p]
 * so a wavefront (64 consecutive points) reads/writes 512 contiguous bytes
 * per field per time step.
 * ---------------------------------------------------------------------- */

typedef struct RsPlan RsPlan; /* opaque */

/* One step-resolution forcing window resident on the device (device pointers).
 * Index t_local = 0 corresponds to absolute (1-based, reference) time index
 * `t0`.  Optional streams may be NULL:
 *   tsurfobs NULL -> all missing (-9999.9);  depth NULL -> all missing;
 *   tdew NULL -> not range-checked (treated as in range).
 * hour: int32, either shared [t] (hour_pstride = 0) or per point
 * [t*t_stride + p] (hour_pstride = 1). */
typedef struct RsForcing {
  const double *tair, *tdew, *vz, *rhz, *prec, *sw, *lw;
  const double *tsurfobs, *depth;
  const int32_t *precphase;
  const int32_t *hour;
  int64_t t_stride;     /* elements between consecutive time indices */
  int32_t hour_pstride; /* 0 shared axis, 1 per point */
  /* sky view (src/ModRadiation.f90): direct short-wave and net long-wave streams, and the
   * time-only solar quantities of rs_sun_table, [nsteps][RS_SUN_COLS] = {ra, stG, sin decl, cos decl,
   * cos(stG - ra), sin(stG - ra)}
   * on a time axis shared by all points.  NULL when no point has 0 <= sky_view < 1. */
  const double *sw_dir, *lw_net;
  const double *sun;
} RsForcing;

typedef struct RsOutputs {
  double *tsurf, *snow, *water, *ice, *deposit, *ice2;
  int64_t t_stride;
  int32_t decimate; /* 1: every step (reference SaveOutput, src/InputOutput.f90:151-165);
                       k>1: only indices i with (i-1) % k == 0 are written (what the
                       reference driver keeps, examples/example1/src/roadrunner.cpp:290,303) */
  int64_t row0;     /* absolute output row r = (i-1)/decimate is stored at buffer
                       row r - row0 */
} RsOutputs;

/* Per-point parameters (device pointers, [npoints]). */
typedef struct RsPointParams {
  const double *tbottom;  /* Tmp(NLayers+1), src/Initialization.f90:267 */
  const int32_t *initlen; /* LocalParameters.InitLenI */
  const double *tair_relax, *vz_relax, *rh_relax; /* may be NULL if !use_relaxation */
  /* LocalParameters.couplingIndexI / couplingTsurf; may be NULL if !use_coupling.  With
   * coupling the step window must be the whole series (t0 = 1, nsteps = SimLen): a point
   * replays its coupling window up to 25 times (src/Coupling.f90:61-78,324), so the window
   * has to stay addressable. */
  const int32_t *coupling_index;
  const double *coupling_tsurf;
  /* sky view: LocalParameters.sky_view, the geometry of rs_point_geometry and the local
   * horizon table [360][npoints_padded] (degree of azimuth is the slow axis; NULL = all 0).
   * NULL sky_view = no point uses the sky-view branch. */
  const double *sky_view, *sin_lat, *cos_lat, *lon_rad;
  const double *horizons;
  double albedo_surroundings; /* InputParameters.Albedo_surroundings */
  /* Column of `horizons` that belongs to slot s: horizons[deg * npoints_padded + horizon_index[s]].
   * NULL = column s.  What a caller that re-sorts the plan's slots (rs_hip_recluster*) passes instead
   * of a re-ordered copy of the 2.9 KB-per-point table: the scalars above are gathered into slot
   * order, the table stays where it is and the kernels read it through this row (the plan's own
   * order row, rs_hip_plan_order, is exactly that). */
  const int32_t *horizon_index;
  /* Layout of `horizons` (ABI 5).  0: [360][npoints_padded], the degree of azimuth is the slow axis (above).
   * 1: [npoints][360], the caller's own layout (LocalParameters' c_local_horizons rows, RsDriverInput::
   * horizons) - element horizons[horizon_index[s] * 360 + deg]: no transpose on the way in, and a point's
   * consecutive degrees share cache lines (the azimuth of the sun moves a degree in four minutes). */
  int32_t horizons_by_point;
} RsPointParams;

const char *rs_last_error(void);
int rs_hip_device_count(void);

/* Create a plan for `npoints` points on HIP device `device`; uploads constants,
 * allocates the carried-state block.  `stream` is a hipStream_t (0 = default),
 * all rs_hip_* work of this plan is enqueued on it. */
RsPlan *rs_hip_plan_create(int32_t device, int64_t npoints,
                           const RsConstants *consts, void *stream);
void rs_hip_plan_destroy(RsPlan *plan);
int64_t rs_hip_plan_npoints(const RsPlan *plan);
size_t rs_hip_plan_state_bytes(const RsPlan *plan);

/* Initialization (src/Initialization.f90:65-147, device part): builds the
 * initial profile and storages from forcing index 1 (t_local 0 of `f`, which
 * must have t0 == 1).  Also applies the VZ(1) >= 0.4 clamp logically (the
 * kernel clamps at i == 1; device forcing is not modified). */
int rs_hip_init_state(RsPlan *plan, const RsForcing *f, const RsPointParams *pp);

/* Advance all points over absolute time indices [t0, t0+nsteps) using the
 * window `f` (whose t_local 0 is absolute index t0) and write outputs for
 * those indices into `o` (row 0 = index t0, or decimated as described).
 * Handles the reference's final-step semantics when the range includes
 * SimLen (lastValues, src/InputOutput.f90:169-198).
 * Asynchronous on the plan's stream.  Returns 0 or <0. */
int rs_hip_step(RsPlan *plan, const RsForcing *f, const RsOutputs *o,
                const RsPointParams *pp, int32_t t0, int32_t nsteps);

/* Coupling with time-chunked windows.  rs_hip_step with coupling wants the whole series in one
 * window, because a point replays its coupling window (src/Coupling.f90:61-78).  The pair below
 * lets a caller keep windows of a few hundred indices:
 *   rs_hip_step_cpl    like rs_hip_step for a chunk [t0, t0+nsteps) of a coupled plan, in lock step:
 *                      saves the state at a point's couplingStartI, runs the first pass of its
 *                      window, decides at couplingEndI (Coupling_control), applies the decaying
 *                      corrections behind it - but never replays: a point that must replay PARKS
 *                      behind its window (and takes no step in later rs_hip_step_cpl calls) until
 *   rs_hip_cpl_replay  has run its replays: the window `f` must cover every parked point's
 *                      [couplingStartI, couplingEndI + 1] (t0 <= min start, t0+nsteps-1 >= max end + 1 unless that
 *                      is beyond SimLen: the reference runs CheckValues on the index behind the
 *                      window before it rewinds, examples/example1/src/Simulation.f90:59-66);
 *                      rounds of the general kernel over the compacted list of points that still
 *                      ask for a replay, until none does (`rounds` = how many, <= 25).
 * A point only ever steps the index it is due for (it remembers it), so the caller then simply
 * continues - or, if the points' windows end at different indices, re-issues - the chunks from
 * min(couplingEndI) + 1 on: points that are ahead wait.  Sky view (RsPointParams::sky_view with
 * sin_lat, cos_lat, lon_rad, RsForcing::sw_dir, lw_net and sun rows of the window) is honoured by both
 * calls; the write-back of rs_hip_set_writeback is not (it follows whole-series windows: rs_hip_step).
 * Outputs of replayed indices are overwritten, as in the reference. */
/* With plan order (rs_hip_recluster) AND coupling: rs_hip_step_cpl / rs_hip_cpl_replay can write
 * the outputs of slot s into column order[s] of the output window, i.e. in POINT order whatever
 * the slots' order is (a replay rewrites rows of earlier launches, so a per-launch buffer in slot
 * order does not do).  Meant for decimated outputs: the stores are scattered. */
int rs_hip_set_output_by_point(RsPlan *plan, int32_t on);
/* closed = 1: every point's coupling window, replays included, is behind the plan (the caller has
 * run rs_hip_cpl_replay over the last of them).  A re-sort then moves the coupling scalars only, not
 * the state saved at the window start (saveDataForCoupling, src/Coupling.f90:172-210) nor the stale
 * TmpNw profile: 32 instead of 68 rows at NLayers = 15.  Reset to 0 by rs_hip_init_state. */
int rs_hip_coupling_windows_closed(RsPlan *plan, int32_t closed);
int rs_hip_step_cpl(RsPlan *plan, const RsForcing *f, const RsOutputs *o,
                    const RsPointParams *pp, int32_t t0, int32_t nsteps);
int rs_hip_cpl_replay(RsPlan *plan, const RsForcing *f, const RsOutputs *o,
                      const RsPointParams *pp, int32_t t0, int32_t nsteps, int32_t *rounds);

/* The reference edits its INPUT arrays in place (SURVEY.md 8b "Ownership"): CheckValues clamps
 * SW_dir(i) to SW(i) (src/InputOutput.f90:75-77) and the sky-view correction rewrites SW(i),
 * SW_dir(i), LW(i) (src/ModRadiation.f90:57-71).  The device forcing windows are never modified;
 * a caller that wants those edits gives three device streams laid out like the forcing window of
 * the NEXT rs_hip_step calls (row 0 = index t0, `t_stride` elements per row), pre-filled with the
 * original values: the sky-view kernels then store SW_dir (every stepped index) and SW, LW (points
 * with 0 <= sky_view < 1) as the reference leaves them; with coupling the last replay wins, as in
 * the reference.  All NULL switches it off.  Without sky view the only edit is the SW_dir clamp,
 * which needs no device work (rs_host_run_batch does it on the host). */
int rs_hip_set_writeback(RsPlan *plan, double *sw, double *sw_dir, double *lw, int64_t t_stride);

/* Copy the carried state block device->host / host->device (checkpointing,
 * tests).  Layout: [RS_NSTATE][npoints_padded] doubles. */
int rs_hip_state_download(RsPlan *plan, double *host, size_t bytes);
int rs_hip_state_upload(RsPlan *plan, const double *host, size_t bytes);
/* Number of points with the sticky failure flag set (src/InputOutput.f90:66). */
int64_t rs_hip_failed_count(RsPlan *plan);
/* Measurement aid: one wavefront that reads the shader-clock counter and the constant 100 MHz
 * counter about spin_us microseconds apart and leaves the two deltas in out[0], out[1] (device
 * memory, 2 x uint64).  Enqueued on a side stream beside the step kernels it tells the engine clock
 * the chip holds under that load: MHz = 100 * out[0] / out[1].  Asynchronous on `stream`.  spin_us is
 * clamped to 10 ms and the wait is bounded: out[1] = 0 means the 100 MHz counter did not advance. */
int rs_hip_clock_probe(int32_t device, void *out, uint32_t spin_us, void *stream);
/* Per point (host int32[npoints], in local point order whatever the plan order is): 0, or the
 * 1-based time index at which CheckValues raised simulation_failed - the step of that index was
 * still taken and saved, later outputs read -9999.0 (examples/example1/src/Simulation.f90:58,
 * src/InputOutput.f90:55-82).  The reference only prints this ("BAD input value!"). */
int rs_hip_first_failed_index(RsPlan *plan, int32_t *first_failed);
/* The reference's other run-time messages (it prints them and carries on): from CalcBLCondAndLE
 * " ERROR : UStar negative,vz" + the values line inside the boundary-layer loop and " Max number of BLCond
 * iterations (MaxIter,BLCond_Old,BLCond) :" behind it (src/BoundaryLayer.f90:69-74,98-101), from Coupling_control
 * "coupling coefficient too small / too big, coupling failed" (src/Coupling.f90:400-401,451-452).
 * rs_hip_set_diagnostics(plan, 1) - fp64 plans in natural order; launches of such a plan take the one-point-per-
 * lane kernels with the profile in LDS, which re-run the loop of every step out of line for the two tests: a
 * debugging switch, not a production setting - zeroes the plan's record; rs_hip_diagnostics copies it out,
 * out[npoints][RS_DIAG_COLS]: 0 time indices that printed "Max number ...", 1 the first of them, 2-4 its j,
 * BLCond_Old, BLCond; 5 passes that printed "UStar negative", 6 the time index of the first, 7-11 its Tair, VZ,
 * Rhz, BLCond, TSurfAve; 12 the coupling messages (8 = too small, 16 = too big; these are kept by every kernel
 * flavour, with or without the switch). */
#define RS_DIAG_COLS 13
int rs_hip_set_diagnostics(RsPlan *plan, int32_t on);
int rs_hip_diagnostics(RsPlan *plan, double *out);
int rs_hip_sync(RsPlan *plan);

/* Synthetic forcing: hourly knots from a counter-based hash,
 * expanded to step resolution by the same linear rule the reference driver
 * uses (examples/example1/src/JsonSource.cpp:115-170).  Device kernels;
 * host twins with identical arithmetic are in roadsurf_amd/csrc/rs_synth.h. */
typedef struct RsSynthSpec {
  uint64_t seed;
  int64_t point_offset;   /* global id of local point 0 (multi-GPU shards) */
  int32_t steps_per_knot; /* 3600/DTSecs = 120 */
  int32_t start_hour;     /* hour of day at absolute index 1 */
  const int32_t *order;   /* NULL, or rs_hip_plan_order(): knot column p then belongs to local
                             point order[p] (plan-order windows, see rs_hip_recluster) */
} RsSynthSpec;

/* Number of doubles per point and knot in a knot buffer. */
#define RS_KNOT_FIELDS 9
/* The same generator on the host, in the layout the reference driver hands runsimulation: per-point
 * [n][simlen] series (InputPointers' eleven f64 arrays + PrecPhase) and the hour of every index (the shared
 * calendar; examples/example1/src/InputData.cpp:5-26).  Host code, no device call: bench.py's host-array leg,
 * __graft_entry__.smoke() and the parity tests take their inputs from it, so that the CPU checker and the GPU
 * path see bit-identical forcing (csrc/rs_synth_host.hip). */
void rs_synth_fill_points(uint64_t seed, int64_t point_offset, int32_t n, int32_t simlen,
                          int32_t steps_per_knot, int32_t start_hour, double *tair, double *tdew,
                          double *vz, double *rhz, double *prec, double *sw, double *lw,
                          double *sw_dir, double *lw_net, double *tsurfobs, double *depth,
                          int32_t *precphase, int32_t *hour);

/* Generate hourly knots k0..k0+nknots-1 into `knots` (device,
 * [nknots][RS_KNOT_FIELDS][npoints_padded] doubles; npoints_padded from
 * rs_hip_plan_npoints_padded). */
int rs_hip_synth_knots(RsPlan *plan, const RsSynthSpec *spec, double *knots,
                       int32_t k0, int32_t nknots);
/* Expand knots to step-resolution forcing for absolute indices
 * [t0, t0+nsteps) into the caller's device buffers `f` (written here).  The
 * knot buffer must hold knots (t0-1)/spk .. (t0+nsteps-2)/spk + 1, the first
 * of them being knot k0.  hour is written as a shared [nsteps] axis. */
int rs_hip_expand_forcing(RsPlan *plan, const RsSynthSpec *spec,
                          const double *knots, int32_t k0, int32_t nknots,
                          const RsForcing *f, int32_t t0, int32_t nsteps);
/* Same with the knot buffer in POINT order (generated once, spec->order NULL) and the window produced
 * in the plan's current SLOT order: column p of the window is interpolated from knot column
 * rs_hip_plan_order()[p].  What a plan that is re-sorted between launches uses instead of
 * regenerating the knots in slot order for every window. */
int rs_hip_expand_forcing_ordered(RsPlan *plan, const RsSynthSpec *spec,
                                  const double *knots, int32_t k0, int32_t nknots,
                                  const RsForcing *f, int32_t t0, int32_t nsteps);

/* rs_hip_expand_forcing_ordered + rs_hip_step in one launch, without the window: the two-wavefront
 * flavour's ground wave interpolates the forcing of the next index from the knots itself (same arithmetic,
 * same values).  knots in POINT order (spec->order NULL), read through the plan's order row.  LEAN feature
 * set, NLayers = 15, fp64 only (anything else: an error - use the two calls).  What a small shard uses:
 * there the window expansion is the longest link of the chain between two step launches of a plan. */
int rs_hip_step_knots(RsPlan *plan, const RsSynthSpec *spec, const double *knots, int32_t k0,
                      int32_t nknots, const RsOutputs *o, const RsPointParams *pp, int32_t t0,
                      int32_t nsteps);

/* Same, enqueued on another HIP stream (hipStream_t) than the plan's: lets a caller
 * overlap the HBM-bound expansion of window c+1 with the VALU-bound stepping of
 * window c (double-buffered windows, dependencies by HIP events on the caller's side). */
int rs_hip_expand_forcing_on(RsPlan *plan, const RsSynthSpec *spec,
                             const double *knots, int32_t k0, int32_t nknots,
                             const RsForcing *f, int32_t t0, int32_t nsteps,
                             void *stream);

/* Test hook: y[i] = device exp (fn 0) / log (fn 1) / bare square root (fn 3) of x[i], or the bare
 * division x[i] / x[n + i] (fn 2; x holds 2n values); device pointers
 * (roadsurf_amd/csrc/rs_math.hpp; tests/test_hip_math.py). */
int rs_hip_test_math(RsPlan *plan, int32_t fn, int64_t n, const double *x, double *y);

/* Arithmetic flavour of a plan: 64 (default; the parity path) or 32 (BASELINE config 5:
 * fp32 state/forcing/outputs/arithmetic, tolerance-gated against fp64).  Every feature of the
 * model runs in either flavour.  Which fp32 kernel a launch takes: NLayers == 15 without an output
 * depth or coupling - two points per lane, two wavefronts per 128 points (LEAN, the FULL set, and
 * through rs_hip_step sky view with local horizons); anything else - coupling (rs_hip_step over
 * the whole series: every point replays its coupling window inside the launch), tsurfOutputDepth
 * or a depth stream, the FULL set or sky view at another layer count - the general kernel with one
 * point per lane and the profile in LDS; LEAN launches at another layer count likewise one point
 * per lane.  Refused on an fp32 plan: rs_hip_step_cpl / rs_hip_cpl_replay (time-chunked coupling),
 * rs_hip_set_diagnostics, the write-back of the in-place input edits (rs_hip_set_writeback).
 * With 32 the `double *` members of RsForcing/RsOutputs point to FLOAT arrays of the same
 * [t][p] layout, depth, sw_dir and lw_net included (precphase/hour stay int32; tbottom, the sun
 * table and the per-point members of RsPointParams - relaxation targets, coupling observation,
 * sky view, latitude / longitude terms, horizons - stay double: the sun's position is worked out
 * in fp64 in either flavour), and the state block holds floats.  Set before rs_hip_init_state. */
int rs_hip_set_precision(RsPlan *plan, int32_t bits);

/* How this build divides: 0 = compiler's IEEE expansion everywhere (-DRS_IEEE_DIV),
 * 1 = bare Newton sequence for normal-range operands (default, same bits, see
 * rs_math.hpp), 2 = both evaluated and compared (-DRS_DIV_CHECK).  In mode 2
 * rs_hip_div_mismatch_count returns how many call-site evaluations disagreed
 * with two finite results since the library was loaded (0 in the other modes). */
int rs_hip_division_mode(void);
/* Experiment builds (-DRS_BL_STATS): out[56] = {wave-steps, lane-steps, wave-steps in which every active lane's
 * boundary-layer fixed point repeated its bits at pass 2, 3, 4, the same per lane, trip counts and which
 * wave-uniform shortcuts applied (rs_math.hpp g_bl_stats)}; zeros in the product. */
int rs_hip_bl_stats(RsPlan *plan, int64_t *out);
int64_t rs_hip_div_mismatch_count(RsPlan *plan);
/* mode 2: evaluations where one of the two results is inf/NaN (x/0, x/inf, inf/x: the bare
 * sequence gives NaN there) - the operands the boundary-layer guard handles in the shipped build
 * (roadsurf_amd/csrc/rs_physics_body.inc) - plus those that differ only in the sign of a zero
 * result (-0.0/b; harmless, rs_math.hpp); rs_hip_div_mismatch_count counts different VALUES */
int64_t rs_hip_div_special_count(RsPlan *plan);
/* mode 2: the first 64 finite mismatches as {numerator or sqrt argument, denominator (0 for a
 * square root), IEEE result, bare result}; out = host double[64][4] */
int rs_hip_div_samples(RsPlan *plan, double *out);

/* Kernel flavour: 0 auto, 1 register profile (NLayers == 15), 2 LDS profile (any NLayers),
 * 3 two wavefronts per 64 points - one steps the surface (forcing, boundary layer, layers 1-2,
 * storages), the other the layers below, meeting once per time step: the flavour for launches too
 * small to fill the chip with one point per lane (LEAN feature set, NLayers == 15, windows under
 * 4 GiB per stream; any other launch of such a plan runs as 0).  0 picks 3 for launches of at most
 * 65 536 points (the measured break-even), else 1 for NLayers == 15 with the LEAN feature set, 4 with
 * the FULL one, else 2.
 * 4: layers 1-7 in registers, 8-15 in LDS columns - the FULL feature set (NLayers == 15) at four
 * waves per SIMD without scratch spills; a LEAN launch of such a plan runs as 0.
 * (Until round 6 a tens digit bounded the waves per SIMD of flavours 1 and 2; the measured choices -
 * LEAN four, FULL three - are the kernels' launch bounds now.)
 * All flavours return the same bits.  An fp32 plan (rs_hip_set_precision): 0 = two points per lane, two
 * wavefronts per 128 points (NLayers == 15); 1 or 2 = one point per lane with the profile in LDS. */
int rs_hip_set_variant(RsPlan *plan, int32_t variant);

/* ---- plan order: load balancing by regime ----------------------------------------------
 * The boundary-layer fixed point (src/BoundaryLayer.f90:64-96) stops after its 5 mandatory passes
 * for most point-steps but needs up to ~35 for some, and a wavefront runs until its slowest lane
 * is done: with independent points in arbitrary order a wave averages 13.9 passes where a lane
 * averages 5.5.  Slow convergence is a property of the weather regime and persists for hours.
 * rs_hip_recluster sorts the plan's SLOTS by the extra passes each point needed during the last
 * launch and moves the carried state accordingly, so that slow points share wavefronts.  From
 * then on everything indexed by "point" in this API - forcing and output windows, RsPointParams
 * arrays - is indexed by SLOT: slot s holds local point rs_hip_plan_order(plan)[s].  Producers
 * generate their windows in that order (the synthetic generator takes the order in RsSynthSpec;
 * raw series are 130x smaller than the windows, so gathering them is cheap) and consumers map
 * the outputs back with the same array.  Points do not interact: the order changes no value.
 *   rs_hip_plan_order   device pointer to order[npoints_padded] (identity until the first
 *                       recluster; allocated by the first call), valid until the next recluster
 *   rs_hip_recluster    asynchronous on the plan's stream; the order changes for the NEXT launch */
const int32_t *rs_hip_plan_order(RsPlan *plan);
int rs_hip_recluster(RsPlan *plan);
/* Outputs stay attributable to points (SaveOutput is per point, src/InputOutput.f90:151-165):
 *   rs_hip_plan_order_copy   copies the current order (int32[npoints_padded]) to a device buffer of
 *                            the caller, asynchronously on the plan's stream - call it after the
 *                            rs_hip_step of a launch and before rs_hip_recluster, and column s of
 *                            that launch's output window is local point dst[s] (4 bytes per point and
 *                            launch against the 48 bytes per point and time index of the outputs)
 *   rs_hip_plan_reset_order  back to the identity (a new run over the same plan) */
/* rs_hip_recluster_forecast: like rs_hip_recluster, but the sort key is a FORECAST of the next
 * launch instead of the history of the last one.  What a point costs a wavefront is decided by its
 * boundary-layer regime (stable: no sqrt/log; unstable: which path of log) and by how many passes
 * its fixed point needs - functions of (Tsurf - Tair, wind speed) only (src/BoundaryLayer.f90:64-96).
 * The forcing of the next window is known before it is stepped, so the key kernel runs the fixed
 * point for each point at a few PREVIEW times of the next window (surface temperature = the carried
 * one, moved along with the air temperature by `alpha`) and sorts by
 *     (how many previews are unstable, how many of those take the table path of log, [cover],
 *      predicted passes beyond the mandatory five).
 * Rows are device pointers in the plan's CURRENT slot order, [npoints_padded] each. */
#define RS_PREVIEW_MAX 8
typedef struct RsPreview {
  int32_t n;                             /* 1..RS_PREVIEW_MAX preview times */
  const double *tair[RS_PREVIEW_MAX];    /* air temperature at preview time q */
  const double *vz[RS_PREVIEW_MAX];      /* wind speed at preview time q */
  int32_t hour[RS_PREVIEW_MAX];          /* hour of day at preview time q (calm limit day/night) */
  const double *tair_now;                /* air temperature at the first index of the next window */
  double alpha;                          /* Tsurf(preview) = Tsurf + alpha*(Tair(preview) - Tair_now) */
  int32_t mode;                          /* key fields in priority order as decimal digits: 1 unstable
                                            previews, 2 table-path previews, 3 cover, 4 predicted extra
                                            passes (e.g. 1234); 0..3 = 14, 124, 134, 1234; 5 storage
                                            class, 6/7/8 = 4/1/2 in fewer bits, 0 (inside a list) = three
                                            bits for the LONGEST loop expected in the window - over the previews,
                                            the last index stepped and a passage through the loop's slow band -
                                            in classes 5, 6, 7, 8, 9-12, 13-20, 21-30, 31+ passes (round 4, a
                                            device whose live plans hold at most 131 072 points, and
                                            ROADSURF_HIP_EXTRA_CLASSES=0: field 4 saturating at 7); 9 (anywhere in the list)
                                            = the ground digit, which layers are frozen, always the LEAST
                                            significant field, honoured for keys of at most 12 bits without
                                            it (the plan's counting sort; 378059 is what bench.py and
                                            rs_driver_run use) */
  const int32_t *index;                  /* NULL: the preview rows are in SLOT order; else row element
                                            index[slot] belongs to that slot (rows kept in point order,
                                            index = rs_hip_plan_order(): no regeneration after a re-sort) */
  const double *prec[RS_PREVIEW_MAX];    /* ABI 8.  precipitation at preview time q, or all NULL.  With rows: the
                                            key gets one more bit, its MOST significant - "some preview of the
                                            next window has precipitation" - whatever `mode` says.  A wavefront
                                            none of whose points has precipitation at an index skips
                                            PrecipitationToStorage / CalcPrecType there (exact: they add zeros);
                                            precipitating points are a few per cent of a batch at any time but, in
                                            arbitrary order, sit in most wavefronts.  Read for every q < RS_PREVIEW_MAX
                                            with a row, whatever n says (the rows need not be at the preview times) */
  const double *tair_b[RS_PREVIEW_MAX];  /* ABI 9.  Previews BETWEEN two rows: where tair_b[q] is not NULL, preview q is */
  const double *vz_b[RS_PREVIEW_MAX];    /* tair[q] + w[q] * (tair_b[q] - tair[q]), likewise vz - the straight line the */
  double w[RS_PREVIEW_MAX];              /* forcing itself follows between two hourly knots: a caller whose rows are the
                                            knots places its previews AT the first and last index of the next
                                            window instead of at the knots around it (bench.py's workload: a wavefront's
                                            boundary-layer passes per step 6.50 -> 6.21).  tair_now may then be NULL: the
                                            air temperature of preview 0 */
} RsPreview;
int rs_hip_recluster_forecast(RsPlan *plan, const RsPreview *preview);
/* A plan that is only ever sorted by forecast can tell the step kernels not to keep the history
 * score (a few vector instructions per boundary-layer pass); rs_hip_recluster then refuses. */
int rs_hip_set_history_score(RsPlan *plan, int32_t on);
int rs_hip_plan_order_copy(RsPlan *plan, int32_t *dst_device);
/* ABI 10.  The output rows of a launch in POINT order at every index, for a consumer that keeps one series per
 * point (SaveOutput fills a point's arrays, src/InputOutput.f90:151-165; OutputData.cpp:5-13): the first
 * `nrows` rows of the window `src` ([row][slot], fp64) are written into point-major arrays
 *     dst[stream][point * dst_rows + dst_row0 + row],   stream = Tsurf, Snow, Water, Ice, Deposit, Ice2,
 * point = order[slot] (`order`: a device row kept with rs_hip_plan_order_copy, or NULL = the plan's current
 * order: then call it between the step launch and the next re-sort).  Asynchronous on the plan's stream, or - with
 * a kept order row - on `stream` (a hipStream_t of the caller's, who orders it behind the launch and in front of
 * the window's next use: with two windows in turn the pass overlaps the next launch).  One
 * pass over the rows (read once, written once in whole lines), and not a cheap one - the outputs are most of
 * the bytes this path moves: at 1 M points the pass with it after every launch runs at 1.4-1.5e10 point-timesteps/s
 * against 2.5e10 without; natural order, whose rows are [row][point], not per-point series, runs at 1.46e10
 * (tools/bench_point_order_outputs.py). */
int rs_hip_outputs_by_point(RsPlan *plan, const RsOutputs *src, int32_t nrows, const int32_t *order_device,
                            double *const *dst_device, int64_t dst_rows, int64_t dst_row0, void *stream);
int rs_hip_plan_reset_order(RsPlan *plan);

/* Device timing of the step kernel with HIP events recorded on the plan's
 * stream around every rs_hip_step launch since the last reset.  Returns the
 * summed milliseconds (synchronises on the last event) and the launch count. */
int rs_hip_timing_reset(RsPlan *plan);
double rs_hip_timing_step_ms(RsPlan *plan, int32_t *nlaunches);
/* The same launches as intervals [start, stop] in milliseconds after `ref_event` (a hipEvent_t
 * with timing enabled that the caller recorded on this device before the launches).  With several
 * plans stepping concurrently on one device (one stream each) their step kernels overlap in time;
 * the union of all plans' intervals is the time during which the device ran step kernels.  Returns
 * the number of launches (at most `cap` are written) or < 0. */
int32_t rs_hip_timing_intervals(RsPlan *plan, void *ref_event, double *start_ms, double *stop_ms,
                                int32_t cap);
int64_t rs_hip_plan_npoints_padded(const RsPlan *plan);

/* ------------------------------------------------------------------------
 * Layer 2: host-array batch (used by the Fortran runsimulation_batch).
 * Inputs are the per-point arrays of the reference boundary; packing to SoA,
 * H2D, kernels and D2H happen inside, tiled over points.  tbottom[n] is
 * computed by the Fortran caller.  Returns 0 or <0.
 * ---------------------------------------------------------------------- */
typedef struct RsHostExtras {
  /* sky view: NULL/0 when no point has 0 <= sky_view < 1 */
  const double *sun;      /* rs_sun_table of the shared time axis, [SimLen][RS_SUN_COLS] */
  const double *sin_lat;  /* rs_point_geometry, [n] each */
  const double *cos_lat;
  const double *lon_rad;
  double albedo_surroundings;
  /* out, [n] or NULL: per point 0, or the 1-based time index at which its run was failed
   * (rs_hip_first_failed_index) */
  int32_t *first_failed;
  /* 1: write the reference's in-place edits of the caller's input arrays back
   * (src/InputOutput.f90:75-77: SW_dir(i) = min(SW_dir(i), SW(i)) at every checked index;
   * src/ModRadiation.f90:57-71: SW, SW_dir, LW as the sky-view correction leaves them) */
  int32_t writeback;
  /* out, [n][RS_DIAG_COLS] or NULL: rs_hip_diagnostics' record of every point (the plans of the call then run
   * with rs_hip_set_diagnostics) */
  double *diagnostics;
} RsHostExtras;

/* `device` >= 0: that device.  `device` < 0 (what runsimulation_batch / runsimulation pass): the
 * batch is cut into contiguous blocks of points over the device list - environment
 * ROADSURF_HIP_DEVICES ("0,1,2,3", a device may repeat; default "all" visible devices), blocks of
 * at least ROADSURF_HIP_MIN_SHARD points (default 4096), batches too small to split take the
 * devices in turn - one host thread + stream + plan per block, no collective: the in-process
 * counterpart of the reference driver's worker pool (examples/example1/src/roadrunner.cpp:423-501).
 * rs_last_fanout(): number of blocks the calling thread's last call used. */
int rs_last_fanout(void);
int rs_host_run_batch(int32_t n, OutputPointers *outPointers,
                      const InputPointers *inPointers,
                      const RsConstants *consts,
                      const LocalParameters *localParam, const double *tbottom,
                      const RsHostExtras *extras, int32_t device);

/* ------------------------------------------------------------------------
 * Layer 4: the reference DRIVER's data path on the device.
 *
 * What examples/example1 does on the host for every point before and after
 * `runsimulation` — and what makes a step-resolution boundary PCIe-bound — happens here
 * on the GPU, so only the RAW series (hourly NWP / observations) cross the bus inbound and
 * only the decimated outputs come back:
 *   JsonSource ctor      Tdew <-> RH completion of the raw series
 *                        (JsonSource.cpp:288-295, MeteorologyTools.cpp:12-51)
 *   JsonSource::interpolate   raw times -> simulation times, per variable, with the
 *                        reference's missing-value rules (JsonSource.cpp:49-176)
 *   DataHandler::GetWeather   later sources overwrite earlier ones where they have data
 *                        (DataHandler.cpp:75-84, JsonSource.cpp:323-373; PrecPhase and
 *                        Depth are NOT carried over there, so they stay missing)
 *   read_input           completeness check, relaxation targets, coupling index/observation
 *                        and the blanking of TSurfObs inside the coupling window
 *                        (roadrunner.cpp:156-278)
 *   runsimulation        the hot path (layers 1-3)
 *   save_output          every (outputStep*60/DTSecs)-th index (roadrunner.cpp:285-315)
 * Out of scope: JSON/text parsing and writing (host work, roadsurf_amd/driver.py has a
 * reader/writer for the reference's schema).
 * A source's points either share one raw time axis or have one each (RsRawSource).
 * ---------------------------------------------------------------------- */
#define RS_MAX_SOURCES 4

/* One data source after parsing (JsonSource.cpp:182-316).  Arrays are HOST pointers,
 * [n_points][n_times] row-major (one contiguous series per point, as the reference's
 * InputData holds them); NULL = the variable is absent from this source (all missing,
 * -9999.9).  PrecipitationForm is not listed: the reference reads it but never hands it
 * on (JsonSource.cpp:323-373 does not copy PrecPhase).
 * Time axis: either one axis shared by all points (times[n_times], lengths NULL: gridded NWP
 * data), or one per point like the reference's per-station "time" arrays
 * (times_per_point = 1: times[n_points][n_times], and lengths[n_points] gives how many leading
 * entries of a point's row are real: rows are padded to the common width n_times). */
typedef struct RsRawSource {
  int32_t n_times;          /* row width of every array below */
  int32_t is_observation;   /* DataHandler.cpp:65-66: counts for GetLatestObsIndex */
  const int64_t *times;     /* epoch seconds, strictly increasing within a series */
  const double *tair, *rhz, *tdew, *vz, *prec, *lw_net, *lw, *sw, *sw_dir, *tsurfobs;
  int32_t times_per_point;  /* 0 shared axis, 1 per-point axes */
  const int32_t *lengths;   /* per-point series lengths (times_per_point = 1), or NULL = n_times */
} RsRawSource;

typedef struct RsDriverInput {
  int32_t n_points;
  int32_t n_sources;            /* 1..RS_MAX_SOURCES, applied in this order */
  const RsRawSource *sources;
  int64_t start_time;           /* InputSettings.start_time; simulation index k (0-based) is at
                                   start_time + k*int(DTSecs)  (JsonSource.cpp:199-205) */
  int64_t forecast_time;        /* InputSettings.forecast_time (roadrunner.cpp:168-169) */
  /* local calendar of the simulation times, [SimLen] each (JsonSource.cpp:297-308) */
  const int32_t *year, *month, *day, *hour, *minute, *second;
  const double *horizons;       /* [n_points][360] local horizon angles, or NULL (all 0) */
} RsDriverInput;

/* Per-point status: 0 simulated; 1..6 a mandatory variable is missing at `missing_index`
 * (tair, Rhz, prec, SW, LW, VZ: the order roadrunner.cpp:188-229 tests them in) and the
 * point is skipped like the reference skips it (outputs stay -9999.0); 7 the relaxation
 * index equals SimLen (the reference reads one past its arrays there, roadrunner.cpp:246-248). */
typedef struct RsDriverOutput {
  int32_t n_out;                /* number of kept indices: ceil(SimLen / step) */
  double *tsurf, *snow, *water, *ice, *deposit, *ice2; /* host [n_points][n_out]; NULL = not wanted */
  int32_t *status;              /* [n_points] */
  int32_t *missing_index;       /* [n_points] 0-based index of the first missing value, else -1 */
} RsDriverOutput;

/* local[n_points]: in lat, lon, sky_view (others ignored); out InitLenI, tair_relax,
 * VZ_relax, RH_relax, couplingIndexI, couplingTsurf as read_input leaves them.
 * device >= 0: that device; device < 0: fan-out over the device list like rs_host_run_batch.
 * Returns 0 or <0 (rs_last_error). */
int rs_driver_run(const RsDriverInput *in, const InputSettings *settings,
                  const InputParameters *params, LocalParameters *local,
                  const RsDriverOutput *out, int32_t device);
/* Tiles: a call steps its points in tiles of ROADSURF_HIP_TILE_POINTS (default 524 288).  With
 * coupling the forcing windows of a tile span [first coupling-window start, last window end + 1]
 * of ITS points; a tile whose windows would exceed ROADSURF_HIP_WINDOW_BUDGET_MB (default 24 576)
 * - stations whose observations ended hours apart - is cut in halves, down to 4 096 points.
 * rs_driver_last_tiles(): tiles the calling thread's last call with device >= 0 stepped. */
int rs_driver_last_tiles(void);
/* Since ABI 6 the blocks of rs_driver_run step with a kernel that makes its forcing from the raw series itself
 * (no forcing windows, no expansion kernel) wherever it can: sources on shared time axes, NLayers = 15, no
 * coupling, no output depth; ROADSURF_HIP_DRIVER_WINDOWS=1 keeps the windows.  Step launches of that kind in
 * the calling thread's last call with device >= 0 (tests). */
int rs_driver_last_raw_launches(void);
/* rs_driver_run keeps its forcing-window block (up to 64 GB of HBM with coupling) for the next
 * call, because releasing and re-acquiring that much VRAM costs seconds (the driver wipes it).
 * This frees it. */
void rs_driver_release_cache(void);
/* Test hook: only the input side.  merged = host [10][n_points][SimLen] in the order tair,
 * tdew, VZ, Rhz, prec, SW, LW, SW_dir, LW_net, TSurfObs: what read_input returns. */
int rs_driver_expand(const RsDriverInput *in, const InputSettings *settings,
                     LocalParameters *local, double *merged, int32_t *status,
                     int32_t *missing_index, int32_t device);

/* ------------------------------------------------------------------------
 * Layer 5: the device side of `module RoadSurf`'s per-step procedures
 * (roadsurf_amd/fortran/RoadSurfCompat.f90 over roadsurf_amd/csrc/rs_compat.hip; reference:
 * src/RoadSurf.f90:6-270, driven by examples/example1/src/Simulation.f90:57-115).  ONE point, ONE time
 * index per call: a compatibility path for callers that own the time loop, not a fast one.
 * ---------------------------------------------------------------------- */
typedef struct RsCompat RsCompat;
/* the caller's series of one point (HOST pointers, [SimLen] each, reference layout; NULL = absent) */
typedef struct RsCompatArrays {
  const double *tair, *tdew, *vz, *rhz, *prec, *sw, *lw, *sw_dir, *lw_net, *tsurfobs, *depth;
  const int32_t *precphase, *hour;
  const double *horizons; /* [360] */
  double *out[6];         /* Tsurf, Snow, Water, Ice, Deposit, Ice2: rows a coupling replay rewrites */
} RsCompatArrays;
/* state_out: the point's state column, RS_COMPAT_NSTATE doubles in the order of roadsurf_amd/csrc/rs_state.h */
#define RS_COMPAT_NSTATE (RS_MAX_LAYERS + 17 + 21 + 2 * RS_MAX_LAYERS)
RsCompat *rs_compat_begin(const RsConstants *consts, const LocalParameters *local, double tbottom,
                          const RsCompatArrays *arrays, const double *sun, const double *geo,
                          double albedo_surroundings, double *state_out);
int rs_compat_step(RsCompat *ctx, int32_t i, double *state_out, double *edits);
int rs_compat_replay(RsCompat *ctx, int32_t i, double *state_out, int32_t *rewritten);
int rs_compat_last_state(const RsCompat *ctx, double *state_out);
int rs_compat_outputs(const RsCompat *ctx, int32_t i, double *out6);
int32_t rs_compat_failed_index(const RsCompat *ctx);
void rs_compat_end(RsCompat *ctx);

/* Hash of the sources this library was built from (16 hex digits; roadsurf_amd/provenance.py build_sha16()):
 * tests/conftest.py rebuilds a prebuilt library whose stamp is not the hash of the sources beside it. */
const char *rs_build_sha16(void);
#define RS_ABI_VERSION 11 /* 2: round 2 additions (forecast re-sort, coupling rounds, fan-out, writeback, failure index); 3: RsPreview::index, rs_hip_expand_forcing_ordered, rs_hip_clock_probe, rs_driver_last_tiles; 4: RsPointParams::horizon_index, rs_hip_step_knots; 5: RS_SUN_COLS 6, RsPointParams::horizons_by_point; 6: rs_driver_last_raw_launches (rs_driver_run without forcing windows); 7: rs_compat_* (module RoadSurf's per-step procedures); 8: RsPreview::prec; 9: RsPreview::tair_b / vz_b / w; 10: rs_hip_outputs_by_point; 11: rs_hip_set_diagnostics / rs_hip_diagnostics, RsHostExtras::diagnostics */
int rs_abi_version(void);
/* sizeof of the boundary structs as the C side / the Fortran side see them
 * (0 InputPointers, 1 OutputPointers, 2 InputSettings, 3 InputParameters,
 * 4 LocalParameters, 5 RsConstants); bindings assert that they agree. */
int64_t rs_abi_sizeof(int which);
int64_t rs_fortran_sizeof(int which);

#ifdef __cplusplus
}
#endif
#endif /* ROADSURF_H */
