/*
 * rs_api.hip — the C-ABI of include/roadsurf.h, layer 1 (device-resident SoA).
 *
 * A plan owns: the HIP device, a stream (the caller's, e.g. torch's current
 * stream, or the default), the uploaded constants, the carried-state block
 * [RS_NSTATE][npoints_padded] in HBM and a pool of HIP events used to time
 * the step kernel on that stream.
 */
#include <hip/hip_runtime.h>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/roadsurf.h"
#include "rs_kernels.h"
#include "rs_devutil.hpp"
#include <cstdint>
#include <algorithm>
#include "rs_consts_dev.h"
#include "rs_state.h"

static thread_local char g_err[512] = "";

static int set_err(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return -1;
}

#define HIP_OK(expr)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess)                                                              \
      return set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

/* exp/log tables are per-device globals, uploaded once per device; plans are created
 * concurrently from driver threads (examples/example1/src/roadrunner.cpp:490-497) */
static bool g_tables_up[64];
static std::mutex g_tables_mutex;

struct RsPlan {
  int device = 0;
  int64_t npoints = 0, np_pad = 0;
  RsConstants c{};
  hipStream_t stream = nullptr;
  double *state = nullptr;
  unsigned long long *counter = nullptr;
  /* plan order (rs_hip_recluster): slot -> local point, second state block and sort scratch */
  int32_t *order = nullptr, *order_alt = nullptr;
  double *state_alt = nullptr;
  void *sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  uint32_t *sort_keys = nullptr; /* [4][np_pad]: keys in/out, slots in/out */
  int variant = RS_VARIANT_AUTO;
  /* the plan's constants on the device (no table of slots: any number of plans may be alive) */
  void *consts_dev = nullptr;   /* RsConstantsDev (rs_consts_dev.h) */
  double *relax_tab = nullptr;  /* RsConstantsDev::relax_tab */
  double *cpl_tab = nullptr;    /* RsConstantsDev::cpl_tab */
  void *consts32_dev = nullptr; /* RsConstantsF, allocated by rs_hip_set_precision(32) */
  rs::Writeback wb{nullptr, nullptr, nullptr, 0};
  /* coupling rounds: scratch for the list of points that replay their window */
  int32_t *cpl_flags = nullptr, *cpl_list = nullptr, *cpl_count = nullptr;
  void *cpl_tmp = nullptr;
  size_t cpl_tmp_bytes = 0;
  int32_t cpl_rounds_last = 0; /* replay rounds of the last coupled rs_hip_step (diagnostics) */
  bool output_by_point = false; /* coupling kernels scatter their outputs through the plan order */
  bool cpl_windows_closed = false; /* rs_hip_coupling_windows_closed: re-sorts leave the saved state */
  bool history_score = true; /* the step kernels leave the sort key of rs_hip_recluster */
  bool f32 = false; /* single-precision flavour: windows and state hold floats */
  double *diag = nullptr; /* rs_hip_set_diagnostics: [RS_DIAG_ROWS][np_pad], by slot; NULL = off */
  bool diag_on = false;
  bool resorted = false; /* the order row has been something else than the identity since the last reset */
  /* wave table of the two-wavefront flavour (rs_cluster_wave_table): [2][wave_n] start, count; valid
   * for the slot order the last forecast re-sort left */
  int32_t *wave_tab = nullptr;
  int32_t wave_n = 0;
  bool wave_tab_valid = false;
  std::vector<void *> owned; /* hipMalloc-ed pieces (plan_malloc): what rs_hip_plan_destroy frees */
  /* the arena the plan's first pieces came out of (nullptr: hipMalloc) and where it stood: a later piece is
   * taken from an arena only if it is still THAT arena and it has not been rewound since (a plan
   * that outlives its tile, or is touched from another thread, must not alias another tile's buffers) */
  rsu::Arena *arena = nullptr;
  uint64_t arena_epoch = 0;
  bool arena_set = false;
  std::vector<hipEvent_t> ev; /* start/stop pairs */
  size_t ev_used = 0;
  bool timing = false;
};

/* Device memory of a plan: out of the calling thread's arena where one is installed (rs_devutil.hpp:
 * the workers of rs_driver_run create and destroy a plan per tile), else hipMalloc, kept in
 * RsPlan::owned for rs_hip_plan_destroy. */
template <class T>
static hipError_t plan_malloc(RsPlan *pl, T **p, size_t bytes) {
  rsu::Arena *a = rsu::tls_arena();
  if (!pl->arena_set) { /* the plan's first allocation decides */
    pl->arena_set = true;
    pl->arena = a;
    pl->arena_epoch = a ? a->epoch : 0;
  }
  if (a && a == pl->arena && a->epoch == pl->arena_epoch)
    if (void *q = a->take(bytes ? bytes : 8)) {
      *p = static_cast<T *>(q);
      return hipSuccess;
    }
  void *q = nullptr;
  const hipError_t e = hipMalloc(&q, bytes ? bytes : 8);
  if (e == hipSuccess) {
    pl->owned.push_back(q);
    *p = static_cast<T *>(q);
  }
  return e;
}

uint64_t rs_a32_limit(void) {
  /* (read per call: a test lowers it for one case) */
  const char *e = getenv("ROADSURF_HIP_A32_LIMIT");
  const uint64_t hard = 1ull << 29;
  if (!e) return hard;
  const unsigned long long v = strtoull(e, nullptr, 10);
  return (v >= 1024 && v < hard) ? (uint64_t)v : hard;
}

extern "C" {

const char *rs_last_error(void) { return g_err; }
void rs_host_set_error(const char *msg) { set_err("%s", msg ? msg : ""); }
int rs_abi_version(void) { return RS_ABI_VERSION; }
/* layout cross-check for bindings: 0 InputPointers 1 OutputPointers 2 InputSettings
 * 3 InputParameters 4 LocalParameters 5 RsConstants; the device-resident API's structs: 6 RsForcing
 * 7 RsOutputs 8 RsPointParams 9 RsHostExtras 10 RsPreview 11 RsSynthSpec; the raw-series boundary's:
 * 12 RsRawSource 13 RsDriverInput 14 RsDriverOutput */
int64_t rs_abi_sizeof(int which) {
  switch (which) {
    case 0: return sizeof(InputPointers);
    case 1: return sizeof(OutputPointers);
    case 2: return sizeof(InputSettings);
    case 3: return sizeof(InputParameters);
    case 4: return sizeof(LocalParameters);
    case 5: return sizeof(RsConstants);
    case 6: return sizeof(RsForcing);
    case 7: return sizeof(RsOutputs);
    case 8: return sizeof(RsPointParams);
    case 9: return sizeof(RsHostExtras);
    case 10: return sizeof(RsPreview);
    case 11: return sizeof(RsSynthSpec);
    case 12: return sizeof(RsRawSource);
    case 13: return sizeof(RsDriverInput);
    case 14: return sizeof(RsDriverOutput);
    default: return -1;
  }
}

int rs_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void rs_default_parameters(InputParameters *p, double DTSecs) {
  /* examples/example1/src/InputParameters.h:18-94 */
  std::memset(p, 0, sizeof(*p));
  p->NightOn = 19.0; p->NightOff = 4.0; p->CalmLimDay = 1.5; p->CalmLimNgt = 0.4;
  p->TrfFricNgt = 5.0; p->TrFfricDay = 10.0;
  p->Grav = 9.81; p->SB_Const = 5.67e-8; p->VK_Const = 0.4; p->LVap = 2.452E6;
  p->LFus = 0.334E6; p->WatDens = 999.87; p->SnowDens = 100.0; p->IceDens = 920.0;
  p->DepDens = 920.0; p->WatMHeat = 333000.0; p->PorEvaF = 1.0;
  p->ZRefW = 10.0; p->ZRefT = 2.0; p->ZeroDisp = 0.0; p->ZMom = 0.4000; p->ZHeat = 0.0010;
  p->Emiss = 0.95; p->Albedo = 0.10; p->Albedo_surroundings = 0.15; p->MaxPormms = 1.0;
  p->TClimG = 6.4; p->DampDpth = 2.7; p->Omega = 2.0 * M_PI / 365.0; p->AZ = 0.6;
  p->DampWearF = 0.5; p->AlbDry = 0.1; p->AlbSnow = 0.6; p->vsh1 = 1.94e+06;
  p->vsh2 = 1.28e+06; p->Poro1 = 0.1; p->Poro2 = 0.4; p->RhoB1 = 2.11; p->RhoB2 = 1.6;
  p->Silt1 = 0.1; p->Silt2 = 0.8;
  p->freezing_limit_normal = -0.25; p->snow_melting_limit_normal = 0.25;
  p->ice_melting_limit_normal = 0.25; p->frost_melting_limit_normal = 1.25;
  p->frost_formation_limit_normal = 0.25; p->T4Melt_normal = 0.25;
  p->TLimColdH = -19.0; p->TLimColdL = -21.0; p->WetSnowFormR = 0.1; p->WetSnowMeltR = 0.6;
  p->PLimSnow = 0.3; p->PLimRain = 0.7; p->MaxSnowmms = 100.0; p->MaxDepmms = 2.0;
  p->MaxIcemms = 50.0; p->MaxExtmms = 1.0;
  p->MissValI = -9999; p->MissValR = -99.99;
  p->Snow2IceFac = 0.5;
  /* examples/example1/src/InputParameters.cpp:13-21 */
  p->MinPrecmm = 0.05 * DTSecs / 3600.0;
  p->MinWatmms = 0.01 * DTSecs / 3600.0;
  p->MinSnowmms = 0.1 * DTSecs / 3600.0;
  p->MaxWatmms = p->MaxPormms + p->MaxExtmms;
  p->WDampLim = 0.1 * p->MaxPormms;
  p->WWetLim = 0.9 * p->MaxPormms;
  p->WWearLim = 0.1 * p->MaxPormms;
  p->MinDepmms = 0.01 * DTSecs / 3600.0;
  p->MinIcemms = 0.05 * DTSecs / 3600.0;
}

void rs_default_settings(InputSettings *s, int32_t SimLen) {
  /* examples/example1/src/InputSettings.h:13-23 */
  std::memset(s, 0, sizeof(*s));
  s->SimLen = SimLen;
  s->DTSecs = 30.0;
  s->tsurfOutputDepth = -9999.9;
  s->NLayers = 15;
  s->coupling_minutes = 180;
  s->couplingEffectReduction = 4.0 * 3600;
  s->outputStep = 60;
}

void rs_default_local(LocalParameters *l) {
  /* examples/example1/src/LocalParameters.h:17-25 */
  std::memset(l, 0, sizeof(*l));
  l->tair_relax = l->VZ_relax = l->RH_relax = -9999.0;
  l->couplingIndexI = -9999;
  l->couplingTsurf = -9999.0;
  l->lat = l->lon = -9999.0;
  l->sky_view = 1.0;
  l->InitLenI = 0;
}

/* Points of the live plans per device.  While ALL of them fit the device at once as wavefront pairs of the
 * two-wavefront flavour (4 wavefronts x 4 SIMDs x 256 CUs = 4 096 slots = 131 072 points), the SIMDs are
 * underfilled and a launch is as long as its slowest workgroup's chain: the surface wave - the longer chain of
 * the two - then runs at raised issue priority (StepArgs::surface_prio; +2.6 % at 125 000 points, and -0.6 ...
 * -5.5 % where more points are resident: profiles/r04_surface_wave_priority.txt). */
static std::atomic<int64_t> g_live_points[64];
static inline int32_t underfilled(const RsPlan *pl);

RsPlan *rs_hip_plan_create(int32_t device, int64_t npoints, const RsConstants *consts,
                           void *stream) {
  if (!consts || npoints <= 0) {
    set_err("rs_hip_plan_create: bad arguments");
    return nullptr;
  }
  if (consts->NLayers < 5 || consts->NLayers > RS_MAX_LAYERS) {
    set_err("rs_hip_plan_create: NLayers=%d outside [5,%d]", consts->NLayers, RS_MAX_LAYERS);
    return nullptr;
  }
  if (const char *bad = rs_consts_domain_error(*consts)) {
    /* the kernels divide by these without IEEE's special-case handling (rs_math.hpp); the
     * reference itself would be dividing by zero or by a non-finite number */
    set_err("rs_hip_plan_create: parameter outside the model's domain (zero, negative or non-finite "
            "where it is a divisor): %s", bad);
    return nullptr;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_err("rs_hip_plan_create: no HIP device visible - this library has no CPU path");
    return nullptr;
  }
  if (device < 0 || device >= ndev || device >= 64) {
    set_err("rs_hip_plan_create: device %d out of range (%d visible)", device, ndev);
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) {
    set_err("rs_hip_plan_create: hipSetDevice(%d) failed", device);
    return nullptr;
  }
  RsPlan *pl = new RsPlan();
  pl->device = device;
  pl->npoints = npoints;
  pl->np_pad = (npoints + RS_BLOCK - 1) / RS_BLOCK * RS_BLOCK;
  pl->c = *consts;
  pl->stream = (hipStream_t)stream;
  if (const char *ev = getenv("ROADSURF_HIP_VARIANT")) { /* tuning: default flavour of new plans */
    const int v = atoi(ev);
    if (v >= 0 && v <= 4 && !(v == RS_VARIANT_REG && consts->NLayers != 15))
      pl->variant = v;
  }
  const size_t bytes = (size_t)RS_NSTATE * pl->np_pad * sizeof(double);
  if (plan_malloc(pl, &pl->state, bytes) != hipSuccess ||
      plan_malloc(pl, &pl->counter, sizeof(unsigned long long)) != hipSuccess) {
    set_err("rs_hip_plan_create: hipMalloc of %zu state bytes failed", bytes);
    for (void *q : pl->owned) (void)hipFree(q);
    delete pl;
    return nullptr;
  }
  (void)hipMemsetAsync(pl->state, 0, bytes, pl->stream);
  RsConstantsDev cd;
  rs_consts_dev_fill(pl->c, cd);
  hipError_t ce = hipSuccess;
  {
    const std::vector<double> tab = rs_relax_table(pl->c);
    if (!tab.empty()) {
      ce = plan_malloc(pl, &pl->relax_tab, tab.size() * sizeof(double));
      if (ce == hipSuccess)
        ce = hipMemcpy(pl->relax_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice);
      cd.relax_tab = pl->relax_tab;
    }
    const std::vector<double> ctab = rs_cpl_table(pl->c);
    if (ce == hipSuccess && !ctab.empty()) {
      ce = plan_malloc(pl, &pl->cpl_tab, ctab.size() * sizeof(double));
      if (ce == hipSuccess)
        ce = hipMemcpy(pl->cpl_tab, ctab.data(), ctab.size() * sizeof(double), hipMemcpyHostToDevice);
      cd.cpl_tab = pl->cpl_tab;
    }
  }
  if (ce == hipSuccess) ce = plan_malloc(pl, &pl->consts_dev, sizeof(RsConstantsDev));
  if (ce == hipSuccess)
    ce = hipMemcpyAsync(pl->consts_dev, &cd, sizeof(RsConstantsDev), hipMemcpyHostToDevice, pl->stream);
  if (ce == hipSuccess) ce = hipStreamSynchronize(pl->stream); /* cd is stack scratch */
  if (ce == hipSuccess) {
    std::lock_guard<std::mutex> lock(g_tables_mutex);
    if (!g_tables_up[device]) {
      ce = rs_upload_math_tables(pl->stream);
      if (ce == hipSuccess) ce = hipStreamSynchronize(pl->stream);
      if (ce == hipSuccess) g_tables_up[device] = true;
    }
  }
  if (ce != hipSuccess) {
    set_err("rs_hip_plan_create: upload of the constants failed: %s", hipGetErrorString(ce));
    for (void *q : pl->owned) (void)hipFree(q);
    delete pl;
    return nullptr;
  }
  g_live_points[device & 63] += npoints;
  return pl;
}

static inline int32_t underfilled(const RsPlan *pl) {
  return g_live_points[pl->device & 63].load() <= 131072 ? 1 : 0;
}

void rs_hip_plan_destroy(RsPlan *pl) {
  if (!pl) return;
  g_live_points[pl->device & 63] -= pl->npoints;
  (void)hipSetDevice(pl->device);
  (void)hipStreamSynchronize(pl->stream);
  for (hipEvent_t e : pl->ev) (void)hipEventDestroy(e);
  for (void *q : pl->owned) (void)hipFree(q); /* pieces of a worker's arena are the worker's to reclaim */
  delete pl;
}

const int32_t *rs_hip_plan_order(RsPlan *pl) {
  if (!pl) {
    set_err("rs_hip_plan_order: null plan");
    return nullptr;
  }
  if (!pl->order) {
    if (hipSetDevice(pl->device) != hipSuccess ||
        plan_malloc(pl, &pl->order, pl->np_pad * sizeof(int32_t)) != hipSuccess ||
        rs_cluster_identity(pl->order, pl->np_pad, pl->stream) != hipSuccess) {
      set_err("rs_hip_plan_order: allocation failed");
      return nullptr;
    }
  }
  return pl->order;
}

static int recluster_buffers(RsPlan *pl) {
  const size_t state_bytes = (size_t)RS_NSTATE * pl->np_pad * sizeof(double);
  if (!pl->state_alt) {
    HIP_OK(plan_malloc(pl, &pl->state_alt, state_bytes));
    HIP_OK(plan_malloc(pl, &pl->order_alt, pl->np_pad * sizeof(int32_t)));
    HIP_OK(plan_malloc(pl, &pl->sort_keys, (size_t)4 * pl->np_pad * sizeof(uint32_t)));
    pl->sort_tmp_bytes = std::max(rs_cluster_scratch_bytes(pl->npoints),
                                  rs_cluster_count_scratch_bytes(pl->npoints, 12));
    HIP_OK(plan_malloc(pl, &pl->sort_tmp, pl->sort_tmp_bytes ? pl->sort_tmp_bytes : 8));
  }
  return 0;
}

static int recluster_apply(RsPlan *pl) {
  pl->resorted = true;
  HIP_OK(rs_cluster_apply(pl->state, pl->state_alt, pl->f32, pl->order, pl->order_alt,
                          pl->sort_keys + 3 * pl->np_pad, pl->np_pad, pl->npoints, pl->c.NLayers,
                          pl->c.use_coupling == 0 ? 0 : pl->cpl_windows_closed ? 1 : 2, pl->stream));
  std::swap(pl->state, pl->state_alt);
  std::swap(pl->order, pl->order_alt);
  return 0;
}

int rs_hip_recluster_forecast(RsPlan *pl, const RsPreview *pv) {
  if (!pl || !pv) return set_err("rs_hip_recluster_forecast: bad arguments");
  if (pv->n < 1 || pv->n > RS_PREVIEW_MAX || (!pv->tair_now && !pv->tair_b[0]))
    return set_err("rs_hip_recluster_forecast: 1 <= n <= %d previews and tair_now (or a preview 0 between two rows) are required",
                   RS_PREVIEW_MAX);
  for (int q = 0; q < pv->n; ++q) {
    if (!pv->tair[q] || !pv->vz[q]) return set_err("rs_hip_recluster_forecast: preview row %d is null", q);
    if ((pv->tair_b[q] != nullptr) != (pv->vz_b[q] != nullptr) || (pv->tair_b[q] && !(pv->w[q] >= 0.0 && pv->w[q] <= 1.0)))
      return set_err("rs_hip_recluster_forecast: preview %d: tair_b and vz_b come together, with 0 <= w <= 1", q);
  }
  if (pl->diag_on)
    return set_err("rs_hip_recluster_forecast: a plan with diagnostics keeps its order (rs_hip_set_diagnostics)");
  if (!rs_hip_plan_order(pl)) return -1;
  HIP_OK(hipSetDevice(pl->device));
  if (recluster_buffers(pl)) return -1;
  rs::ForecastArgs a;
  a.consts = pl->consts_dev;
  a.state = pl->state;
  a.f32 = pl->f32 ? 1 : 0;
  a.npoints = pl->npoints;
  a.np_pad = pl->np_pad;
  a.pv = *pv;
  a.keys = pl->sort_keys;
  a.slots = pl->sort_keys + 2 * pl->np_pad;
  /* a key of at most 12 bits is sorted by the plan's own counting pass (3 kernels), a longer one by
   * the library (ROADSURF_HIP_LIBRARY_SORT=1 forces the library: A/B) */
  const int wet_bit = pv->prec[0] ? 1 : 0; /* RsPreview::prec: one more bit, the most significant */
  const int bits = rs_forecast_key_bits(pv->mode) + wet_bit;
  const bool lib_sort = getenv("ROADSURF_HIP_LIBRARY_SORT") != nullptr; /* read per call: the tests switch it */
  a.compact = (bits >= 1 && bits <= 12 && !lib_sort) ? 1 : 0;
  /* the ground digit (field 9 of the mode): below the others - a pass of its own in the plan's counting
   * sort, the low bits of the key in the library's (where they fit) */
  const int lowb = rs_forecast_key_low_bits(pv->mode);
  const int low = (a.compact || (bits >= 1 && bits + lowb <= RS_SORT_KEY_BITS)) ? lowb : 0;
  a.low_bits = low;
  /* field 0 of the mode in classes of the longest expected loop, or - round 4's form - the previews' extra passes
   * summed and saturating at 7, which stays the form of an underfilled device: there a launch is as long as its
   * slowest wavefront, and gathering the points of the slow band makes that wavefront slower (125 000 points:
   * 1.40e10 against 1.37e10; 500 000: 2.39e10 against 2.44e10, profiles/r05_ab_extra_pass_classes_small_shards.txt) */
  a.extra_log = underfilled(pl) ? 0 : 1;
  HIP_OK(rs_launch_forecast_keys(a, pl->stream));
  pl->wave_tab_valid = false;
  if (a.compact) {
    /* For the two-wavefront flavour: the classes of the key - its five most significant
     * bits: cover, unstable previews, table-path previews of the default field set - each start a wavefront
     * of their own (rs_cluster_wave_table; classes hold at least 64 bins of the key, so the default 10/11-bit keys
     * take no table: profiles/r04_key_layouts.txt). */
    const int cb = 5;
    const bool table = cb >= 1 && cb <= 6 && bits - wet_bit - cb >= 6 && !pl->f32;
    uint32_t *class_total = nullptr;
    int32_t maxw = 0;
    if (table) {
      maxw = (int32_t)((pl->npoints + 63) / 64) + (1 << cb);
      if (!pl->wave_tab || pl->wave_n != maxw) {
        HIP_OK(plan_malloc(pl, &pl->wave_tab, ((size_t)2 * maxw + 64) * sizeof(int32_t)));
        pl->wave_n = maxw;
        HIP_OK(hipMemsetAsync(pl->wave_tab + 2 * (size_t)maxw, 0, 64 * sizeof(int32_t), pl->stream));
      }
      class_total = reinterpret_cast<uint32_t *>(pl->wave_tab + 2 * (size_t)maxw);
    }
    HIP_OK(rs_cluster_count_sort(pl->np_pad, pl->npoints, bits, pl->sort_keys, pl->sort_tmp,
                                 pl->sort_tmp_bytes, pl->stream, class_total, cb, low));
    if (table) {
      HIP_OK(rs_cluster_wave_table(cb, class_total, pl->wave_tab, pl->wave_tab + maxw, maxw, pl->stream));
      pl->wave_tab_valid = true;
    }
  } else
    HIP_OK(rs_cluster_sort_keys(pl->np_pad, pl->npoints, pl->sort_keys, pl->sort_tmp, pl->sort_tmp_bytes,
                                pl->stream));
  return recluster_apply(pl);
}

int rs_hip_recluster(RsPlan *pl) {
  if (!pl) return set_err("rs_hip_recluster: null plan");
  if (!pl->history_score)
    return set_err("rs_hip_recluster: the plan's history score is switched off (rs_hip_set_history_score)");
  if (pl->diag_on) return set_err("rs_hip_recluster: a plan with diagnostics keeps its order (rs_hip_set_diagnostics)");
  if (!rs_hip_plan_order(pl)) return -1;
  pl->wave_tab_valid = false;
  HIP_OK(hipSetDevice(pl->device));
  if (recluster_buffers(pl)) return -1;
  HIP_OK(rs_cluster_sort(pl->state, pl->f32, pl->np_pad, pl->npoints, pl->sort_keys, pl->sort_tmp,
                         pl->sort_tmp_bytes, pl->stream));
  return recluster_apply(pl);
}

int rs_hip_plan_order_copy(RsPlan *pl, int32_t *dst) {
  if (!pl || !dst) return set_err("rs_hip_plan_order_copy: bad arguments");
  if (!rs_hip_plan_order(pl)) return -1;
  HIP_OK(hipSetDevice(pl->device));
  HIP_OK(hipMemcpyAsync(dst, pl->order, (size_t)pl->np_pad * sizeof(int32_t), hipMemcpyDeviceToDevice,
                        pl->stream));
  return 0;
}

int rs_hip_outputs_by_point(RsPlan *pl, const RsOutputs *src, int32_t nrows, const int32_t *order,
                            double *const *dst, int64_t dst_rows, int64_t dst_row0, void *stream) {
  if (!pl || !src || !dst || nrows < 1 || dst_rows < 1 || dst_row0 < 0 || dst_row0 + nrows > dst_rows)
    return set_err("rs_hip_outputs_by_point: bad arguments (rows [dst_row0, dst_row0 + nrows) must lie inside a point's dst_rows)");
  if (pl->f32) return set_err("rs_hip_outputs_by_point: fp64 output windows only");
  const double *in[6] = {src->tsurf, src->snow, src->water, src->ice, src->deposit, src->ice2};
  double *out[6];
  for (int f = 0; f < 6; ++f) {
    if (!in[f] || !dst[f]) return set_err("rs_hip_outputs_by_point: all six streams are required on both sides");
    out[f] = dst[f];
  }
  if (src->t_stride < pl->npoints) return set_err("rs_hip_outputs_by_point: t_stride below the plan's points");
  if (!order && stream) return set_err("rs_hip_outputs_by_point: on a stream of the caller's the order row must be a kept one");
  if (!order && !rs_hip_plan_order(pl)) return -1;
  HIP_OK(hipSetDevice(pl->device));
  HIP_OK(rs_cluster_outputs_by_point(in, out, order ? order : pl->order, pl->npoints, src->t_stride, nrows, dst_rows,
                                     dst_row0, stream ? (hipStream_t)stream : pl->stream));
  return 0;
}

int rs_hip_plan_reset_order(RsPlan *pl) {
  if (!pl) return set_err("rs_hip_plan_reset_order: null plan");
  pl->wave_tab_valid = false;
  if (!pl->order) return rs_hip_plan_order(pl) ? 0 : -1; /* allocated as the identity */
  HIP_OK(hipSetDevice(pl->device));
  HIP_OK(rs_cluster_identity(pl->order, pl->np_pad, pl->stream));
  pl->resorted = false;
  return 0;
}

int64_t rs_hip_plan_npoints(const RsPlan *pl) { return pl ? pl->npoints : 0; }
int64_t rs_hip_plan_npoints_padded(const RsPlan *pl) { return pl ? pl->np_pad : 0; }
size_t rs_hip_plan_state_bytes(const RsPlan *pl) {
  return pl ? (size_t)RS_NSTATE * pl->np_pad * sizeof(double) : 0;
}

int rs_hip_set_precision(RsPlan *pl, int32_t bits) {
  if (!pl || (bits != 32 && bits != 64)) return set_err("rs_hip_set_precision: bits must be 32 or 64");
  HIP_OK(hipSetDevice(pl->device));
  if (bits == 32) {
    if (!pl->consts32_dev) HIP_OK(plan_malloc(pl, &pl->consts32_dev, rs32_constants_bytes()));
    HIP_OK(rs32_upload_constants(pl->consts32_dev, &pl->c, pl->stream));
  }
  pl->f32 = (bits == 32);
  return 0;
}

int rs_hip_set_writeback(RsPlan *pl, double *sw, double *sw_dir, double *lw, int64_t t_stride) {
  if (!pl) return set_err("rs_hip_set_writeback: null plan");
  if ((sw || sw_dir || lw) && (!sw || !sw_dir || !lw || t_stride < pl->npoints))
    return set_err("rs_hip_set_writeback: all three streams and t_stride >= npoints, or all NULL");
  pl->wb = rs::Writeback{sw, sw_dir, lw, t_stride};
  return 0;
}

int rs_hip_set_diagnostics(RsPlan *pl, int32_t on) {
  if (!pl) return set_err("rs_hip_set_diagnostics: null plan");
  if (on && pl->f32) return set_err("rs_hip_set_diagnostics: the fp64 flavour only");
  if (on && pl->resorted)
    return set_err("rs_hip_set_diagnostics: the block is kept by slot - not for a plan that has been re-sorted");
  HIP_OK(hipSetDevice(pl->device));
  if (on) {
    const size_t bytes = (size_t)RS_DIAG_ROWS * pl->np_pad * sizeof(double);
    if (!pl->diag) HIP_OK(plan_malloc(pl, &pl->diag, bytes));
    HIP_OK(hipMemsetAsync(pl->diag, 0, bytes, pl->stream)); /* (every call: a cached plan starts a new batch) */
  }
  pl->diag_on = on != 0;
  return 0;
}

int rs_hip_diagnostics(RsPlan *pl, double *out) {
  if (!pl || !out) return set_err("rs_hip_diagnostics: bad arguments");
  if (!pl->diag_on || !pl->diag) return set_err("rs_hip_diagnostics: rs_hip_set_diagnostics(plan, 1) first");
  HIP_OK(hipSetDevice(pl->device));
  std::vector<double> rows((size_t)(RS_DIAG_ROWS + 1) * pl->np_pad);
  HIP_OK(hipMemcpyAsync(rows.data(), pl->diag, (size_t)RS_DIAG_ROWS * pl->np_pad * sizeof(double), hipMemcpyDeviceToHost,
                        pl->stream));
  HIP_OK(hipMemcpyAsync(rows.data() + (size_t)RS_DIAG_ROWS * pl->np_pad, pl->state + (size_t)RS_ST_CPL_FLAGS * pl->np_pad,
                        (size_t)pl->np_pad * sizeof(double), hipMemcpyDeviceToHost, pl->stream));
  HIP_OK(hipStreamSynchronize(pl->stream));
  for (int64_t p = 0; p < pl->npoints; ++p) {
    for (int r = 0; r < RS_DIAG_ROWS; ++r) out[(size_t)p * RS_DIAG_COLS + r] = rows[(size_t)r * pl->np_pad + p];
    const int32_t fl = (int32_t)rows[(size_t)RS_DIAG_ROWS * pl->np_pad + p];
    out[(size_t)p * RS_DIAG_COLS + RS_DIAG_ROWS] = (double)(fl & (RS_CPL_MSG_SMALL | RS_CPL_MSG_BIG));
  }
  return 0;
}

int rs_hip_set_output_by_point(RsPlan *pl, int32_t on) {
  if (!pl) return set_err("rs_hip_set_output_by_point: null plan");
  pl->output_by_point = on != 0;
  return 0;
}

int rs_hip_coupling_windows_closed(RsPlan *pl, int32_t closed) {
  if (!pl) return set_err("rs_hip_coupling_windows_closed: null plan");
  pl->cpl_windows_closed = closed != 0;
  return 0;
}

int rs_hip_set_history_score(RsPlan *pl, int32_t on) {
  if (!pl) return set_err("rs_hip_set_history_score: null plan");
  pl->history_score = on != 0;
  return 0;
}

int rs_hip_set_variant(RsPlan *pl, int32_t variant) {
  /* (until round 6 a tens digit bounded the waves per SIMD of flavours 1 and 2: the measured choices are the
   * kernels' own launch bounds now) */
  if (!pl || variant < 0 || variant > 4) return set_err("rs_hip_set_variant: the flavour is 0 (automatic) ... 4");
  if (variant == RS_VARIANT_REG && pl->c.NLayers != 15)
    return set_err("register-profile kernel is built for NLayers == 15 only (got %d)",
                   pl->c.NLayers);
  pl->variant = variant;
  return 0;
}

static int check_forcing(const RsPlan *pl, const RsForcing *f, const char *who) {
  if (!f || !f->tair || !f->vz || !f->rhz || !f->prec || !f->sw || !f->lw || !f->precphase ||
      !f->hour)
    return set_err("%s: forcing streams tair,vz,rhz,prec,sw,lw,precphase,hour are required", who);
  if (f->t_stride < pl->npoints)
    return set_err("%s: forcing t_stride %lld < npoints %lld", who, (long long)f->t_stride,
                   (long long)pl->npoints);
  return 0;
}

int rs_hip_init_state(RsPlan *pl, const RsForcing *f, const RsPointParams *pp) {
  if (!pl) return set_err("rs_hip_init_state: null plan");
  if (check_forcing(pl, f, "rs_hip_init_state")) return -1;
  if (!pp || !pp->tbottom) return set_err("rs_hip_init_state: tbottom is required");
  HIP_OK(hipSetDevice(pl->device));
  rs::InitArgs a;
  a.consts = pl->f32 ? pl->consts32_dev : pl->consts_dev;
  a.f = *f;
  a.pp = *pp;
  a.state = pl->state;
  a.npoints = pl->npoints;
  a.np_pad = pl->np_pad;
  pl->cpl_windows_closed = false; /* a new run: its coupling windows lie ahead */
  if (pl->f32) {
    HIP_OK(rs32_launch_init(a, pl->stream));
  } else {
    HIP_OK(rs_launch_init(a, pl->stream));
  }
  return 0;
}

/* The replay rounds of a coupled run: while some points ask for another replay of their coupling
 * window (start_coupling_again), those points - compacted into full wavefronts - rewind, replay
 * the window and park again (step_kernel_coupled with cpl_stop).  `a` describes a window that
 * covers every such point's [couplingStartI, couplingEndI]. */
/* lockstep: the window is compact (every point's coupling window fills most of it) and ends before
 * the final index, no sky view: the rounds run the lock-step loop over the list (time_loop<REPLAY>:
 * scalar row arithmetic, coalesced outputs of the first pass' quality) instead of the general
 * kernel, which carries a time index per lane. */
static int cpl_replay_rounds(RsPlan *pl, rs::StepArgs a, bool lockstep = false, bool raw = false) {
  if (!pl->cpl_list) {
    HIP_OK(plan_malloc(pl, &pl->cpl_flags, (size_t)2 * pl->np_pad * sizeof(int32_t)));
    HIP_OK(plan_malloc(pl, &pl->cpl_list, (size_t)pl->np_pad * sizeof(int32_t)));
    HIP_OK(plan_malloc(pl, &pl->cpl_count, sizeof(int32_t)));
    pl->cpl_tmp_bytes = rs_cpl_select_scratch_bytes(pl->npoints);
    HIP_OK(plan_malloc(pl, &pl->cpl_tmp, pl->cpl_tmp_bytes ? pl->cpl_tmp_bytes : 8));
  }
  a.cpl_stop = 1;
  pl->cpl_rounds_last = 0;
  for (int round = 0; round < 64; ++round) { /* the reference stops at 25 */
    HIP_OK(rs_cpl_select_again(pl->state, pl->np_pad, pl->npoints, pl->cpl_flags, pl->cpl_list,
                               pl->cpl_count, pl->cpl_tmp, pl->cpl_tmp_bytes, pl->stream));
    int32_t n_again = 0;
    HIP_OK(hipMemcpyAsync(&n_again, pl->cpl_count, sizeof(n_again), hipMemcpyDeviceToHost, pl->stream));
    HIP_OK(hipStreamSynchronize(pl->stream));
    if (n_again == 0) break;
    if (getenv("ROADSURF_HIP_DRIVER_TIMING"))
      fprintf(stderr, "coupling round %d: %d of %lld points replay\n", round + 1, n_again,
              (long long)pl->npoints);
    a.cpl_list = pl->cpl_list;
    a.cpl_nlist = n_again;
    /* Late rounds hold a few per cent of the points each (on the driver benchmark 98 % of the points
     * replay once, 30 % seven times, 2 % all 25) and follow one another as launches of a few dozen
     * wavefronts.  ONE lock-step launch can let every listed point replay until its Coupling_control is
     * content (cpl_inner) - but a wavefront then runs for as many replays as its slowest lane, where
     * the rounds re-compact the list every time: measured at 1 M points (four blocks sharing the GPU),
     * collapsing from round 1 / 4 / 7 on costs 25 / 10 / 6 % of the whole call (the half-empty wavefronts
     * take issue slots from the other blocks' launches).  So it is done only for the tail of the tail:
     * once the list is down to 1/64 of the plan's points.  ROADSURF_HIP_CPL_COLLAPSE=r collapses from
     * round r on instead (tests; 0 = never). */
    bool collapse = lockstep && !raw && (int64_t)n_again * 64 <= pl->npoints;
    if (const char *e = getenv("ROADSURF_HIP_CPL_COLLAPSE"))
      collapse = lockstep && !raw && atoi(e) > 0 && round + 1 >= atoi(e);
    a.cpl_inner = collapse ? 64 : 1;
    a.cpl_prio = ((int64_t)n_again * 4 <= pl->npoints) ? 1 : 0;
    if (raw) /* two wavefronts per 64 listed points, forcing from the raw series (one replay per round) */
      HIP_OK(rs_launch_step_duo_raw_replay(a, pl->stream));
    else if (lockstep)
      HIP_OK(rs_launch_step_cpl_replay(a, pl->c.NLayers, pl->stream));
    else
      HIP_OK(rs_launch_step_coupled(a, pl->c.NLayers, pl->stream));
    pl->cpl_rounds_last = round + 1;
    if (round == 63) { /* Coupling_control gives up after 25 passes: a point still listed now never will */
      HIP_OK(rs_cpl_select_again(pl->state, pl->np_pad, pl->npoints, pl->cpl_flags, pl->cpl_list,
                                 pl->cpl_count, pl->cpl_tmp, pl->cpl_tmp_bytes, pl->stream));
      HIP_OK(hipMemcpyAsync(&n_again, pl->cpl_count, sizeof(n_again), hipMemcpyDeviceToHost, pl->stream));
      HIP_OK(hipStreamSynchronize(pl->stream));
      if (n_again != 0)
        return set_err("coupling replays: %d points still ask for a replay after 64 rounds - the window "
                       "passed does not cover their coupling windows", n_again);
    }
  }
  return 0;
}

int rs_hip_step(RsPlan *pl, const RsForcing *f, const RsOutputs *o, const RsPointParams *pp,
                int32_t t0, int32_t nsteps) {
  if (!pl) return set_err("rs_hip_step: null plan");
  if (check_forcing(pl, f, "rs_hip_step")) return -1;
  if (!pp || !pp->tbottom) return set_err("rs_hip_step: tbottom is required");
  if (!o || !o->tsurf || !o->snow || !o->water || !o->ice || !o->deposit || !o->ice2)
    return set_err("rs_hip_step: all six output streams are required");
  if (o->t_stride < pl->npoints) return set_err("rs_hip_step: output t_stride < npoints");
  if (o->decimate < 1) return set_err("rs_hip_step: decimate must be >= 1");
  if (t0 < 1 || nsteps < 1 || (int64_t)t0 + nsteps - 1 > pl->c.SimLen)
    return set_err("rs_hip_step: window [%d,%d) outside [1,SimLen=%d]", t0, t0 + nsteps,
                   pl->c.SimLen);
  {
    const int64_t first = ((int64_t)t0 - 1 + o->decimate - 1) / o->decimate;
    if (first < o->row0) return set_err("rs_hip_step: output row0 %lld beyond first row %lld",
                                        (long long)o->row0, (long long)first);
  }
  HIP_OK(hipSetDevice(pl->device));
  /* LEAN kernel is exact when nothing optional can act: no observation forcing
   * after index 1 (at index 1 it is a no-op: the profile was initialised from
   * the same observation, src/Initialization.f90:256-259), no output depth, no
   * relaxation, no Tdew check. */
  const bool full = (pp->initlen != nullptr) || pl->c.force_tsurf || (f->depth != nullptr) ||
                    (pl->c.tsurfOutputDepth >= 0.0) ||
                    (pl->c.use_relaxation && pp->tair_relax != nullptr) || (f->tdew != nullptr);
  if (pl->c.use_relaxation && pp->tair_relax && (!pp->vz_relax || !pp->rh_relax || !pp->initlen))
    return set_err("rs_hip_step: relaxation needs tair_relax, vz_relax, rh_relax and initlen");
  const bool skyview = pp->sky_view != nullptr;
  if (skyview) {
    if (!pp->sin_lat || !pp->cos_lat || !pp->lon_rad || !f->sw_dir || !f->lw_net || !f->sun)
      return set_err("rs_hip_step: sky view needs sin_lat, cos_lat, lon_rad, sw_dir, lw_net and sun");
    if (f->hour_pstride)
      return set_err("rs_hip_step: sky view needs a time axis shared by all points");
  }
  const bool coupled = pl->c.use_coupling && pp->coupling_index != nullptr;
  if (coupled) {
    if (!pp->coupling_tsurf) return set_err("rs_hip_step: coupling needs coupling_index and coupling_tsurf");
    if (t0 != 1 || nsteps != pl->c.SimLen)
      return set_err("rs_hip_step: with coupling the window must be the whole series "
                     "(t0 = 1, nsteps = SimLen = %d): coupling windows are replayed", pl->c.SimLen);
  }
  /* every argument / feature check comes before the event pool is touched: a start event
   * without its stop event would poison rs_hip_timing_step_ms */
  if (pl->c.use_coupling && !pp->coupling_index)
    return set_err("rs_hip_step: use_coupling is set: pass coupling_index/coupling_tsurf");
  if (pl->f32 && pl->diag_on) return set_err("rs_hip_step: diagnostics: the fp64 flavour only");
  if (pl->f32 && (skyview || coupled) && pl->wb.sw_dir)
    return set_err("rs_hip_step: the fp32 flavour does not write the in-place input edits back (fp64 arrays)");
  rs::StepArgs a;
  a.consts = pl->f32 ? pl->consts32_dev : pl->consts_dev;
  a.f = *f;
  a.o = *o;
  a.pp = *pp;
  a.state = pl->state;
  a.npoints = pl->npoints;
  a.np_pad = pl->np_pad;
  a.t0 = t0;
  a.nsteps = nsteps;
  a.wb = pl->wb;
  if (a.wb.sw_dir && !skyview) a.wb = rs::Writeback{nullptr, nullptr, nullptr, 0};
  a.cpl_list = nullptr;
  a.cpl_nlist = 0;
  a.cpl_stop = 0;
  a.cpl_inner = a.cpl_prio = 0;
  a.out_index = nullptr;
  a.wave_start = pl->wave_tab_valid ? pl->wave_tab : nullptr;
  a.wave_cnt = pl->wave_tab_valid ? pl->wave_tab + pl->wave_n : nullptr;
  a.wave_n = pl->wave_n;
  a.duo_full_ok = (full && !skyview && !coupled && !f->depth && !(pl->c.tsurfOutputDepth >= 0.0)) ? 1 : 0;
  {
    /* bit 2: a sky-view launch the two-wavefront flavour can take (rs_launch_step_sky): no output depth, and
     * every stream of the windows within 32-bit offsets */
    const int64_t orows = ((int64_t)t0 + nsteps - 2) / o->decimate - o->row0 + 1;
    const bool a32 = (uint64_t)f->t_stride * (uint64_t)nsteps < rs_a32_limit() &&
                     (uint64_t)o->t_stride * (uint64_t)(orows > 0 ? orows : 1) < rs_a32_limit() &&
                     (!pl->wb.sw_dir || (uint64_t)pl->wb.t_stride * (uint64_t)nsteps < rs_a32_limit());
    if (skyview && !coupled && !f->depth && !(pl->c.tsurfOutputDepth >= 0.0) && a32) a.duo_full_ok |= 4;
  }
  a.surface_prio = underfilled(pl);
  a.knots = nullptr;
  a.knot_gather = nullptr;
  a.knot_k0 = a.knot_n = a.spk = a.start_hour = 0;
  a.r_spk = 0.0;
  std::memset(&a.raw, 0, sizeof(a.raw));
  a.diag = pl->diag_on ? pl->diag : nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (pl->timing) {
    if (pl->ev_used + 2 > pl->ev.size()) {
      hipEvent_t x, y;
      HIP_OK(hipEventCreate(&x));
      if (hipEventCreate(&y) != hipSuccess) {
        (void)hipEventDestroy(x);
        return set_err("rs_hip_step: hipEventCreate failed");
      }
      pl->ev.push_back(x);
      pl->ev.push_back(y);
    }
    e0 = pl->ev[pl->ev_used];
    e1 = pl->ev[pl->ev_used + 1];
    HIP_OK(hipEventRecord(e0, pl->stream));
  }
  hipError_t le;
  if (pl->f32 && (coupled || f->depth || pl->c.tsurfOutputDepth >= 0.0 || ((full || skyview) && pl->c.NLayers != 15)))
    /* the general fp32 kernel (rs_kernels_f32.hip): every point replays its coupling window inside the one launch; or
     * an output depth, or the FULL feature set / sky view at a layer count the two-wavefront kernels are not built for */
    le = rs32_launch_step_coupled(a, pl->c.NLayers, pl->stream);
  else if (pl->f32)
    le = rs32_launch_step(a, pl->c.NLayers, pl->variant, pl->history_score, full || skyview, skyview, pl->stream);
  else if (skyview && !coupled)
    le = rs_launch_step_sky(a, pl->c.NLayers, pl->history_score, pl->stream); /* lock-step FULL + sky view */
  else if (coupled) {
    /* Rounds instead of "every wavefront replays until its slowest lane is through"
     * (src/Coupling.f90:61-78,324: up to 25 replays of a window of up to 360 indices, per point):
     *   1. every point steps to the end of its coupling window and parks there;
     *   2. while some points ask for another replay: those points, compacted into full
     *      wavefronts, rewind, replay their window and park again;
     *   3. every point goes on from behind its window to the end of the series.
     * A point's arithmetic is the same sequence whichever round executes it. */
    a.cpl_stop = 1;
    le = rs_launch_step_coupled(a, pl->c.NLayers, pl->stream);
    if (le == hipSuccess && cpl_replay_rounds(pl, a) != 0) return -1;
    if (le == hipSuccess) {
      a.cpl_list = nullptr;
      a.cpl_nlist = 0;
      a.cpl_stop = 0;
      le = rs_launch_step_coupled(a, pl->c.NLayers, pl->stream);
    }
  } else
    le = rs_launch_step(a, pl->c.NLayers, full, pl->variant, pl->history_score, pl->stream);
  if (le != hipSuccess) /* the pair stays unused: ev_used has not advanced */
    return set_err("rs_hip_step: kernel launch failed: %s", hipGetErrorString(le));
  if (pl->timing) {
    HIP_OK(hipEventRecord(e1, pl->stream));
    pl->ev_used += 2;
  }
  return 0;
}

/* The LEAN step of the two-wavefront flavour WITHOUT a forcing window: the ground wave of every
 * workgroup makes the forcing of the next index from the hourly knots itself (expand_kernel's arithmetic,
 * value for value), one index ahead of the surface wave.  For a small shard the window expansion is the
 * longest link of the chain between two step launches (0.19 of 0.36 ms at 62 500 points, HBM-bound) and
 * nothing of the same plan can run beside it; here it does not exist.  knots: [nknots][RS_KNOT_FIELDS]
 * [np_pad] in POINT order (rs_hip_synth_knots), knot k0 first; the plan's order row maps slots to columns. */
int rs_hip_step_knots(RsPlan *pl, const RsSynthSpec *spec, const double *knots, int32_t k0, int32_t nknots,
                      const RsOutputs *o, const RsPointParams *pp, int32_t t0, int32_t nsteps) {
  if (!pl || !spec || !knots) return set_err("rs_hip_step_knots: bad arguments");
  if (!pp || !pp->tbottom) return set_err("rs_hip_step_knots: tbottom is required");
  if (!o || !o->tsurf || !o->snow || !o->water || !o->ice || !o->deposit || !o->ice2)
    return set_err("rs_hip_step_knots: all six output streams are required");
  if (o->t_stride < pl->npoints) return set_err("rs_hip_step_knots: output t_stride < npoints");
  if (o->decimate < 1) return set_err("rs_hip_step_knots: decimate must be >= 1");
  if (t0 < 1 || nsteps < 1 || (int64_t)t0 + nsteps - 1 > pl->c.SimLen)
    return set_err("rs_hip_step_knots: window [%d,%d) outside [1,SimLen=%d]", t0, t0 + nsteps, pl->c.SimLen);
  const int32_t spk = spec->steps_per_knot;
  if (spk < 1) return set_err("rs_hip_step_knots: steps_per_knot < 1");
  {
    const int32_t kfirst = (t0 - 1) / spk, tlast = t0 + nsteps - 2;
    const int32_t klast = tlast / spk + ((tlast % spk) ? 1 : 0);
    if (kfirst < k0 || klast >= k0 + nknots)
      return set_err("rs_hip_step_knots: need knots %d..%d, buffer has %d..%d", kfirst, klast, k0, k0 + nknots - 1);
    const int64_t first = ((int64_t)t0 - 1 + o->decimate - 1) / o->decimate;
    if (first < o->row0) return set_err("rs_hip_step_knots: output row0 beyond first row");
  }
  const int64_t out_rows = ((int64_t)t0 + nsteps - 2) / o->decimate - o->row0 + 1;
  if (pl->c.NLayers != 15 || pl->c.tsurfOutputDepth >= 0.0 || pp->sky_view ||
      (pl->c.use_coupling && pp->coupling_index) ||
      (!pl->f32 && (uint64_t)o->t_stride * (uint64_t)(out_rows > 0 ? out_rows : 1) >= rs_a32_limit()))
    return set_err("rs_hip_step_knots: NLayers = 15, no output depth, sky view or coupling and (fp64) an output "
                   "window below 4 GiB per stream only - use rs_hip_expand_forcing_ordered + rs_hip_step");
  /* the FULL feature set as far as the knots carry it: the dew point (CheckValues' test), the observation
   * of index 1, an initialization phase, relaxation - what rs_hip_step calls `full` for a window with the
   * Tdew and TsurfObs streams and no depth stream */
  const bool full = (pp->initlen != nullptr) || pl->c.force_tsurf || (pl->c.use_relaxation && pp->tair_relax != nullptr);
  if (pl->c.use_relaxation && pp->tair_relax && (!pp->vz_relax || !pp->rh_relax || !pp->initlen))
    return set_err("rs_hip_step_knots: relaxation needs tair_relax, vz_relax, rh_relax and initlen");
  const int32_t *order = rs_hip_plan_order(pl);
  if (!order) return -1;
  HIP_OK(hipSetDevice(pl->device));
  rs::StepArgs a;
  std::memset(&a, 0, sizeof(a));
  a.consts = pl->f32 ? pl->consts32_dev : pl->consts_dev;
  a.f.t_stride = pl->np_pad; /* no window: nothing of `f` is read */
  a.o = *o;
  a.pp = *pp;
  a.state = pl->state;
  a.npoints = pl->npoints;
  a.np_pad = pl->np_pad;
  a.t0 = t0;
  a.nsteps = nsteps;
  a.wave_start = pl->wave_tab_valid ? pl->wave_tab : nullptr;
  a.wave_cnt = pl->wave_tab_valid ? pl->wave_tab + pl->wave_n : nullptr;
  a.wave_n = pl->wave_n;
  a.duo_full_ok = full ? 3 : 0; /* bit 1: the dew-point test (the knots always carry a dew point) */
  a.surface_prio = underfilled(pl);
  a.knots = knots;
  a.knot_gather = order;
  a.knot_k0 = k0;
  a.knot_n = nknots;
  a.spk = spk;
  a.start_hour = spec->start_hour;
  a.r_spk = 1.0 / (double)spk;
  a.diag = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (pl->timing) {
    if (pl->ev_used + 2 > pl->ev.size()) {
      hipEvent_t x, y;
      HIP_OK(hipEventCreate(&x));
      if (hipEventCreate(&y) != hipSuccess) {
        (void)hipEventDestroy(x);
        return set_err("rs_hip_step_knots: hipEventCreate failed");
      }
      pl->ev.push_back(x);
      pl->ev.push_back(y);
    }
    e0 = pl->ev[pl->ev_used];
    e1 = pl->ev[pl->ev_used + 1];
    HIP_OK(hipEventRecord(e0, pl->stream));
  }
  /* fp32: two points per lane, each lane interpolating its own forcing (rs_kernels_f32.hip) */
  const hipError_t le = pl->f32 ? rs32_launch_step_knots(a, pl->history_score, full, pl->stream)
                                : rs_launch_step_duo_knots(a, pl->history_score, pl->stream);
  if (le != hipSuccess) return set_err("rs_hip_step_knots: kernel launch failed: %s", hipGetErrorString(le));
  if (pl->timing) {
    HIP_OK(hipEventRecord(e1, pl->stream));
    pl->ev_used += 2;
  }
  return 0;
}

} /* extern "C" */

/* The step of rs_driver_run's blocks (internal, declared in rs_kernels.h): the two-wavefront flavour whose
 * ground wave makes the forcing from the RAW series - JsonSource::interpolate + the GetWeather overlay per
 * variable, one index ahead (examples/example1/src/JsonSource.cpp:49-176, DataHandler.cpp:75-84; rs_raw.hpp)
 * - so that neither a forcing window nor its expansion kernel exists on that path.  FULL feature set as the
 * driver has it (dew point, observations, initialization phase, relaxation), optionally per-point sky view
 * (pp->sky_view with the geometry, the horizon table and `sun` = rs_sun_table rows from index t0 on): the
 * ground wave runs CheckValues' sky-view tests and ModRadiationBySurroundings too (they depend on the
 * forcing and the geometry alone, src/ModRadiation.f90:7-73).  pp in SLOT order; raw series in point order
 * behind raw->col; out_by_point: the output rows leave in point order whatever the slots' order. */
int rs_step_raw(RsPlan *pl, const rs::RawForcing *raw, const double *sun, const RsOutputs *o,
                const RsPointParams *pp, int32_t t0, int32_t nsteps, bool out_by_point) {
  if (!pl || !raw || raw->nsrc < 1 || raw->nsrc > RS_MAX_SOURCES || !raw->segs || !raw->hour)
    return set_err("rs_step_raw: bad arguments");
  if (!pp || !pp->tbottom || !pp->initlen) return set_err("rs_step_raw: tbottom and initlen are required");
  if (!o || !o->tsurf || !o->snow || !o->water || !o->ice || !o->deposit || !o->ice2)
    return set_err("rs_step_raw: all six output streams are required");
  if (o->t_stride < pl->npoints || o->decimate < 1) return set_err("rs_step_raw: bad output window");
  if (t0 < 1 || nsteps < 1 || (int64_t)t0 + nsteps - 1 > pl->c.SimLen)
    return set_err("rs_step_raw: window [%d,%d) outside [1,SimLen=%d]", t0, t0 + nsteps, pl->c.SimLen);
  const int64_t first = ((int64_t)t0 - 1 + o->decimate - 1) / o->decimate;
  if (first < o->row0) return set_err("rs_step_raw: output row0 beyond first row");
  const int64_t out_rows = ((int64_t)t0 + nsteps - 2) / o->decimate - o->row0 + 1;
  if (!rs_step_raw_ok(pl) || (uint64_t)o->t_stride * (uint64_t)(out_rows > 0 ? out_rows : 1) >= rs_a32_limit())
    return set_err("rs_step_raw: NLayers = 15, fp64, no output depth and an output window below 4 GiB per stream only");
  const bool coupled = pl->c.use_coupling != 0;
  if (coupled && (!pp->coupling_index || !pp->coupling_tsurf))
    return set_err("rs_step_raw: use_coupling is set: pass coupling_index and coupling_tsurf");
  if (pl->c.use_relaxation && pp->tair_relax && (!pp->vz_relax || !pp->rh_relax))
    return set_err("rs_step_raw: relaxation needs tair_relax, vz_relax, rh_relax and initlen");
  const bool sky = pp->sky_view != nullptr;
  if (sky && (!pp->sin_lat || !pp->cos_lat || !pp->lon_rad || !sun))
    return set_err("rs_step_raw: sky view needs sin_lat, cos_lat, lon_rad and the sun table");
  if (raw->seg0 < 0 || raw->seg0 >= raw->nseg) return set_err("rs_step_raw: seg0 outside the segment table");
  HIP_OK(hipSetDevice(pl->device));
  rs::StepArgs a;
  std::memset(&a, 0, sizeof(a));
  a.consts = pl->consts_dev;
  a.f.t_stride = pl->np_pad; /* no window: nothing of `f` but the sun rows is read */
  a.f.sun = sun;
  a.o = *o;
  a.pp = *pp;
  a.state = pl->state;
  a.npoints = pl->npoints;
  a.np_pad = pl->np_pad;
  a.t0 = t0;
  a.nsteps = nsteps;
  a.wave_start = pl->wave_tab_valid ? pl->wave_tab : nullptr;
  a.wave_cnt = pl->wave_tab_valid ? pl->wave_tab + pl->wave_n : nullptr;
  a.wave_n = pl->wave_n;
  a.duo_full_ok = 3; /* the driver's series always carry a dew point (completed from the humidity where absent) */
  a.surface_prio = underfilled(pl);
  a.raw = *raw;
  a.diag = nullptr;
  /* out_by_point: the rows of slot s go to column order[s] of `o` (scattered stores: meant for decimated rows) */
  a.out_index = (out_by_point && pl->order) ? pl->order : nullptr;
  const hipError_t le = rs_launch_step_duo_raw(a, pl->history_score, sky, coupled, pl->stream);
  if (le != hipSuccess) return set_err("rs_step_raw: kernel launch failed: %s", hipGetErrorString(le));
  return 0;
}

/* The replay rounds over the raw series (internal, rs_kernels.h): what rs_hip_cpl_replay does with a forcing
 * window, with step_kernel_duo<..., CPL, REPLAY> making its forcing itself.  The block [t0, t0 + nsteps) must
 * cover every parked point's [couplingStartI, couplingEndI + 1] and end before SimLen; no sky view. */
int rs_cpl_replay_raw(RsPlan *pl, const rs::RawForcing *raw, const RsOutputs *o, const RsPointParams *pp,
                      int32_t t0, int32_t nsteps, bool out_by_point, int32_t *rounds) {
  if (!pl || !raw || raw->nsrc < 1 || raw->nsrc > RS_MAX_SOURCES || !raw->segs || !raw->hour)
    return set_err("rs_cpl_replay_raw: bad arguments");
  if (!pp || !pp->tbottom || !pp->initlen || !pp->coupling_index || !pp->coupling_tsurf)
    return set_err("rs_cpl_replay_raw: tbottom, initlen, coupling_index and coupling_tsurf are required");
  if (!pl->c.use_coupling || !rs_step_raw_ok(pl) || pp->sky_view)
    return set_err("rs_cpl_replay_raw: a coupled plan with NLayers = 15, fp64, no output depth, no sky view only");
  if (!o || !o->tsurf || !o->snow || !o->water || !o->ice || !o->deposit || !o->ice2 || o->decimate < 1)
    return set_err("rs_cpl_replay_raw: all six output streams are required");
  if (t0 < 1 || nsteps < 1 || (int64_t)t0 + nsteps - 1 >= pl->c.SimLen)
    return set_err("rs_cpl_replay_raw: the block [%d,%d] must lie inside [1, SimLen - 1]", t0, t0 + nsteps - 1);
  const int64_t out_rows = ((int64_t)t0 + nsteps - 2) / o->decimate - o->row0 + 1;
  if ((uint64_t)o->t_stride * (uint64_t)(out_rows > 0 ? out_rows : 1) >= rs_a32_limit())
    return set_err("rs_cpl_replay_raw: output window of 4 GiB per stream or more");
  if (raw->seg0 < 0 || raw->seg0 >= raw->nseg) return set_err("rs_cpl_replay_raw: seg0 outside the segment table");
  HIP_OK(hipSetDevice(pl->device));
  rs::StepArgs a;
  std::memset(&a, 0, sizeof(a));
  a.consts = pl->consts_dev;
  a.f.t_stride = pl->np_pad;
  a.o = *o;
  a.pp = *pp;
  a.state = pl->state;
  a.npoints = pl->npoints;
  a.np_pad = pl->np_pad;
  a.t0 = t0;
  a.nsteps = nsteps;
  a.duo_full_ok = 3;
  a.raw = *raw;
  a.diag = nullptr;
  a.out_index = (out_by_point && pl->order) ? pl->order : nullptr;
  { /* the block must cover the coupling windows of the points that replay (as rs_hip_cpl_replay checks) */
    if (!pl->cpl_list) {
      HIP_OK(plan_malloc(pl, &pl->cpl_flags, (size_t)2 * pl->np_pad * sizeof(int32_t)));
      HIP_OK(plan_malloc(pl, &pl->cpl_list, (size_t)pl->np_pad * sizeof(int32_t)));
      HIP_OK(plan_malloc(pl, &pl->cpl_count, sizeof(int32_t)));
      pl->cpl_tmp_bytes = rs_cpl_select_scratch_bytes(pl->npoints);
      HIP_OK(plan_malloc(pl, &pl->cpl_tmp, pl->cpl_tmp_bytes ? pl->cpl_tmp_bytes : 8));
    }
    int32_t b[2] = {INT32_MAX, 0};
    HIP_OK(hipMemcpyAsync(pl->cpl_flags, b, sizeof(b), hipMemcpyHostToDevice, pl->stream));
    HIP_OK(rs_launch_cpl_window_bounds(a, pl->cpl_flags, pl->stream));
    HIP_OK(hipMemcpyAsync(b, pl->cpl_flags, sizeof(b), hipMemcpyDeviceToHost, pl->stream));
    HIP_OK(hipStreamSynchronize(pl->stream));
    if (b[1] > 0 && (b[0] < t0 || b[1] + 1 > t0 + nsteps - 1))
      return set_err("rs_cpl_replay_raw: the block [%d,%d] does not cover the coupling windows of the points that "
                     "replay, [%d,%d]", t0, t0 + nsteps - 1, b[0], b[1] + 1);
  }
  if (cpl_replay_rounds(pl, a, true, true)) return -1;
  if (rounds) *rounds = pl->cpl_rounds_last;
  return 0;
}

bool rs_step_raw_ok(const RsPlan *pl) {
  return pl && !pl->f32 && pl->c.NLayers == 15 && !(pl->c.tsurfOutputDepth >= 0.0);
}

extern "C" {

/* common validation + argument block of the two chunked-coupling entry points */
static int cpl_args(RsPlan *pl, const RsForcing *f, const RsOutputs *o, const RsPointParams *pp,
                    int32_t t0, int32_t nsteps, const char *who, rs::StepArgs &a) {
  if (!pl) return set_err("%s: null plan", who);
  if (check_forcing(pl, f, who)) return -1;
  if (!pp || !pp->tbottom || !pp->coupling_index || !pp->coupling_tsurf)
    return set_err("%s: tbottom, coupling_index and coupling_tsurf are required", who);
  if (!pl->c.use_coupling) return set_err("%s: the plan's settings have use_coupling = 0", who);
  if (pl->f32) return set_err("%s: time-chunked coupling needs the fp64 flavour (an fp32 plan runs a coupled series whole: rs_hip_step)", who);
  if (pp->sky_view) {
    if (!pp->sin_lat || !pp->cos_lat || !pp->lon_rad || !f->sw_dir || !f->lw_net || !f->sun)
      return set_err("%s: sky view needs sin_lat, cos_lat, lon_rad, sw_dir, lw_net and sun", who);
    if (f->hour_pstride) return set_err("%s: sky view needs a time axis shared by all points", who);
    if (pl->wb.sw_dir)
      return set_err("%s: the write-back of the sky-view edits follows whole-series windows: use "
                     "rs_hip_step", who);
  }
  if (pl->c.use_relaxation && pp->tair_relax && (!pp->vz_relax || !pp->rh_relax || !pp->initlen))
    return set_err("%s: relaxation needs tair_relax, vz_relax, rh_relax and initlen", who);
  if (!o || !o->tsurf || !o->snow || !o->water || !o->ice || !o->deposit || !o->ice2)
    return set_err("%s: all six output streams are required", who);
  if (o->t_stride < pl->npoints || o->decimate < 1) return set_err("%s: bad output window", who);
  if (t0 < 1 || nsteps < 1 || (int64_t)t0 + nsteps - 1 > pl->c.SimLen)
    return set_err("%s: window [%d,%d) outside [1,SimLen=%d]", who, t0, t0 + nsteps, pl->c.SimLen);
  a.consts = pl->consts_dev;
  a.f = *f;
  a.o = *o;
  a.pp = *pp;
  a.state = pl->state;
  a.npoints = pl->npoints;
  a.np_pad = pl->np_pad;
  a.t0 = t0;
  a.nsteps = nsteps;
  a.wb = rs::Writeback{nullptr, nullptr, nullptr, 0};
  a.cpl_list = nullptr;
  a.cpl_nlist = 0;
  a.cpl_stop = 0;
  a.cpl_inner = a.cpl_prio = 0;
  a.out_index = (pl->output_by_point && pl->order) ? pl->order : nullptr;
  a.wave_start = a.wave_cnt = nullptr;
  a.wave_n = 0;
  a.duo_full_ok = 0;
  a.surface_prio = 0;
  a.diag = nullptr; /* (the lock-step coupling launches have no instance with diagnostics) */
  a.knots = nullptr;
  a.knot_gather = nullptr;
  a.knot_k0 = a.knot_n = a.spk = a.start_hour = 0;
  a.r_spk = 0.0;
  std::memset(&a.raw, 0, sizeof(a.raw));
  return 0;
}

int rs_hip_step_cpl(RsPlan *pl, const RsForcing *f, const RsOutputs *o, const RsPointParams *pp,
                    int32_t t0, int32_t nsteps) {
  rs::StepArgs a;
  if (cpl_args(pl, f, o, pp, t0, nsteps, "rs_hip_step_cpl", a)) return -1;
  HIP_OK(hipSetDevice(pl->device));
  HIP_OK(rs_launch_step_cpl(a, pl->c.NLayers, pl->stream));
  return 0;
}

int rs_hip_cpl_replay(RsPlan *pl, const RsForcing *f, const RsOutputs *o, const RsPointParams *pp,
                      int32_t t0, int32_t nsteps, int32_t *rounds) {
  rs::StepArgs a;
  if (cpl_args(pl, f, o, pp, t0, nsteps, "rs_hip_cpl_replay", a)) return -1;
  HIP_OK(hipSetDevice(pl->device));
  /* windows of cplLenI + 1 indices plus the index behind them: compact if the block is not much
   * longer than one window; ROADSURF_HIP_CPL_REPLAY=general|lockstep overrides (tests) */
  bool lockstep = (int64_t)nsteps * 4 <= ((int64_t)pl->c.cplLenI + 2) * 5;
  if (const char *e = getenv("ROADSURF_HIP_CPL_REPLAY")) {
    if (strcmp(e, "general") == 0) lockstep = false;
    if (strcmp(e, "lockstep") == 0) lockstep = true;
  }
  if ((int64_t)t0 + nsteps - 1 >= pl->c.SimLen) lockstep = false;
  {
    /* the window must cover [couplingStartI, couplingEndI + 1] of every point that replays: the rewind
     * reads the forcing of the index behind the window end (CheckValues, Simulation.f90:59-66), and a
     * point whose window start lies before t0 would never step */
    if (!pl->cpl_list) { /* scratch of the rounds (allocated here so that cpl_count exists) */
      HIP_OK(plan_malloc(pl, &pl->cpl_flags, (size_t)2 * pl->np_pad * sizeof(int32_t)));
      HIP_OK(plan_malloc(pl, &pl->cpl_list, (size_t)pl->np_pad * sizeof(int32_t)));
      HIP_OK(plan_malloc(pl, &pl->cpl_count, sizeof(int32_t)));
      pl->cpl_tmp_bytes = rs_cpl_select_scratch_bytes(pl->npoints);
      HIP_OK(plan_malloc(pl, &pl->cpl_tmp, pl->cpl_tmp_bytes ? pl->cpl_tmp_bytes : 8));
    }
    int32_t b[2] = {INT32_MAX, 0};
    HIP_OK(hipMemcpyAsync(pl->cpl_flags, b, sizeof(b), hipMemcpyHostToDevice, pl->stream));
    HIP_OK(rs_launch_cpl_window_bounds(a, pl->cpl_flags, pl->stream));
    HIP_OK(hipMemcpyAsync(b, pl->cpl_flags, sizeof(b), hipMemcpyDeviceToHost, pl->stream));
    HIP_OK(hipStreamSynchronize(pl->stream));
    if (b[1] > 0) {
      const int32_t need_hi = b[1] + 1 < pl->c.SimLen ? b[1] + 1 : pl->c.SimLen;
      if (b[0] < t0 || need_hi > t0 + nsteps - 1)
        return set_err("rs_hip_cpl_replay: the window [%d,%d] does not cover the coupling windows of the "
                       "points that replay, [%d,%d] (window start to the index behind the window end)",
                       t0, t0 + nsteps - 1, b[0], need_hi);
    }
  }
  if (cpl_replay_rounds(pl, a, lockstep)) return -1;
  if (rounds) *rounds = pl->cpl_rounds_last;
  return 0;
}

int rs_hip_timing_reset(RsPlan *pl) {
  if (!pl) return set_err("rs_hip_timing_reset: null plan");
  pl->timing = true;
  pl->ev_used = 0;
  return 0;
}

double rs_hip_timing_step_ms(RsPlan *pl, int32_t *nlaunches) {
  if (!pl) return -1.0;
  double total = 0.0;
  if (nlaunches) *nlaunches = (int32_t)(pl->ev_used / 2);
  if (pl->ev_used == 0) return 0.0;
  if (hipEventSynchronize(pl->ev[pl->ev_used - 1]) != hipSuccess) return -1.0;
  for (size_t i = 0; i + 1 < pl->ev_used; i += 2) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pl->ev[i], pl->ev[i + 1]) != hipSuccess) return -1.0;
    total += ms;
  }
  return total;
}

int32_t rs_hip_timing_intervals(RsPlan *pl, void *ref_event, double *start_ms, double *stop_ms,
                                int32_t cap) {
  if (!pl || !ref_event || !start_ms || !stop_ms) return set_err("rs_hip_timing_intervals: bad arguments");
  if (hipSetDevice(pl->device) != hipSuccess) return -1;
  const int32_t n = (int32_t)(pl->ev_used / 2);
  if (n == 0) return 0;
  if (hipEventSynchronize(pl->ev[pl->ev_used - 1]) != hipSuccess) return -1;
  for (int32_t i = 0; i < n && i < cap; ++i) {
    float a = 0.f, b = 0.f;
    if (hipEventElapsedTime(&a, (hipEvent_t)ref_event, pl->ev[2 * (size_t)i]) != hipSuccess ||
        hipEventElapsedTime(&b, (hipEvent_t)ref_event, pl->ev[2 * (size_t)i + 1]) != hipSuccess)
      return set_err("rs_hip_timing_intervals: hipEventElapsedTime failed");
    start_ms[i] = a;
    stop_ms[i] = b;
  }
  return n;
}

int rs_hip_state_download(RsPlan *pl, double *host, size_t bytes) {
  if (!pl || !host || bytes != rs_hip_plan_state_bytes(pl))
    return set_err("rs_hip_state_download: bad arguments");
  HIP_OK(hipSetDevice(pl->device));
  HIP_OK(hipMemcpyAsync(host, pl->state, bytes, hipMemcpyDeviceToHost, pl->stream));
  HIP_OK(hipStreamSynchronize(pl->stream));
  return 0;
}

int rs_hip_state_upload(RsPlan *pl, const double *host, size_t bytes) {
  if (!pl || !host || bytes != rs_hip_plan_state_bytes(pl))
    return set_err("rs_hip_state_upload: bad arguments");
  HIP_OK(hipSetDevice(pl->device));
  HIP_OK(hipMemcpyAsync(pl->state, host, bytes, hipMemcpyHostToDevice, pl->stream));
  HIP_OK(hipStreamSynchronize(pl->stream));
  return 0;
}

/* out (device, 2 x uint64): see rs_launch_clock_probe; asynchronous on `stream` */
int rs_hip_clock_probe(int32_t device, void *out, uint32_t spin_us, void *stream) {
  if (!out) return set_err("rs_hip_clock_probe: null output");
  HIP_OK(hipSetDevice(device));
  HIP_OK(rs_launch_clock_probe(static_cast<uint64_t *>(out), spin_us, (hipStream_t)stream));
  return 0;
}

int64_t rs_hip_failed_count(RsPlan *pl) {
  if (!pl) return -1;
  if (hipSetDevice(pl->device) != hipSuccess) return -1;
  unsigned long long h = 0;
  if (pl->f32) { /* the state block holds floats: the row is counted on the host (a diagnostic, not a hot path) */
    std::vector<float> row((size_t)pl->np_pad);
    if (hipMemcpyAsync(row.data(), (const char *)pl->state + (size_t)RS_ST_FAILED * pl->np_pad * sizeof(float),
                       row.size() * sizeof(float), hipMemcpyDeviceToHost, pl->stream) != hipSuccess)
      return -1;
    if (hipStreamSynchronize(pl->stream) != hipSuccess) return -1;
    for (int64_t q = 0; q < pl->npoints; ++q) h += row[(size_t)q] != 0.f ? 1 : 0;
    return (int64_t)h;
  }
  if (hipMemsetAsync(pl->counter, 0, sizeof(h), pl->stream) != hipSuccess) return -1;
  if (rs_launch_count_failed(pl->state, pl->np_pad, pl->npoints, pl->counter, pl->stream) !=
      hipSuccess)
    return -1;
  if (hipMemcpyAsync(&h, pl->counter, sizeof(h), hipMemcpyDeviceToHost, pl->stream) != hipSuccess)
    return -1;
  if (hipStreamSynchronize(pl->stream) != hipSuccess) return -1;
  return (int64_t)h;
}

int rs_hip_first_failed_index(RsPlan *pl, int32_t *out) {
  if (!pl || !out) return set_err("rs_hip_first_failed_index: bad arguments");
  HIP_OK(hipSetDevice(pl->device));
  const size_t esz = pl->f32 ? sizeof(float) : sizeof(double);
  std::vector<char> row((size_t)pl->np_pad * esz);
  HIP_OK(hipMemcpyAsync(row.data(), (const char *)pl->state + (size_t)RS_ST_FAILED * pl->np_pad * esz,
                        row.size(), hipMemcpyDeviceToHost, pl->stream));
  std::vector<int32_t> order;
  if (pl->order) { /* slot -> local point */
    order.resize((size_t)pl->np_pad);
    HIP_OK(hipMemcpyAsync(order.data(), pl->order, order.size() * sizeof(int32_t), hipMemcpyDeviceToHost,
                          pl->stream));
  }
  HIP_OK(hipStreamSynchronize(pl->stream));
  for (int64_t s = 0; s < pl->npoints; ++s) {
    const double v = pl->f32 ? (double)reinterpret_cast<const float *>(row.data())[s]
                             : reinterpret_cast<const double *>(row.data())[s];
    const int64_t p = pl->order ? order[(size_t)s] : s;
    if (p >= 0 && p < pl->npoints) out[p] = (int32_t)v;
  }
  return 0;
}

int rs_hip_sync(RsPlan *pl) {
  if (!pl) return set_err("rs_hip_sync: null plan");
  HIP_OK(hipSetDevice(pl->device));
  HIP_OK(hipStreamSynchronize(pl->stream));
  return 0;
}

static int64_t div_mismatch(RsPlan *pl, int which) {
  if (!pl) return -1;
  if (hipSetDevice(pl->device) != hipSuccess) return -1;
  unsigned long long n[3] = {0, 0, 0};
  if (rs_read_div_mismatch(n, pl->stream) != hipSuccess) return -1;
  return which == 1 ? (int64_t)(n[1] + n[2]) : (int64_t)n[0];
}
int64_t rs_hip_div_mismatch_count(RsPlan *pl) { return div_mismatch(pl, 0); }
int64_t rs_hip_div_special_count(RsPlan *pl) { return div_mismatch(pl, 1); }
int rs_hip_div_samples(RsPlan *pl, double *out) {
  if (!pl || !out) return set_err("rs_hip_div_samples: bad arguments");
  HIP_OK(hipSetDevice(pl->device));
  HIP_OK(rs_read_div_samples(out, pl->stream));
  return 0;
}

/* experiment builds with -DRS_BL_STATS (`make blstats`): rs_math.hpp g_bl_stats; zeros otherwise */
int rs_hip_bl_stats(RsPlan *pl, int64_t *out) {
  if (!pl || !out) return set_err("rs_hip_bl_stats: bad arguments");
  HIP_OK(hipSetDevice(pl->device));
  unsigned long long v[56];
  HIP_OK(rs_read_bl_stats(v, pl->stream));
  for (int k = 0; k < 56; ++k) out[k] = (int64_t)v[k];
  return 0;
}

int rs_hip_division_mode(void) {
#if defined(RS_IEEE_DIV)
  return 0;
#elif defined(RS_DIV_CHECK)
  return 2;
#else
  return 1;
#endif
}

int rs_hip_test_math(RsPlan *pl, int32_t fn, int64_t n, const double *x, double *y) {
  if (!pl || !x || !y || n < 1 || fn < 0 || fn > 3) return set_err("rs_hip_test_math: bad arguments");
  HIP_OK(hipSetDevice(pl->device));
  HIP_OK(rs_launch_math_test(fn, n, x, y, pl->stream));
  return 0;
}

int rs_hip_synth_knots(RsPlan *pl, const RsSynthSpec *spec, double *knots, int32_t k0,
                       int32_t nknots) {
  if (!pl || !spec || !knots || nknots < 1 || k0 < 0)
    return set_err("rs_hip_synth_knots: bad arguments");
  if (nknots > 65535) return set_err("rs_hip_synth_knots: at most 65535 knots per call");
  HIP_OK(hipSetDevice(pl->device));
  rs::KnotArgs a;
  a.spec = *spec;
  a.knots = knots;
  a.npoints = pl->npoints;
  a.np_pad = pl->np_pad;
  a.k0 = k0;
  HIP_OK(rs_launch_knots(a, nknots, pl->stream));
  return 0;
}

static int expand_forcing(RsPlan *pl, const RsSynthSpec *spec, const double *knots, int32_t k0,
                          int32_t nknots, const RsForcing *f, int32_t t0, int32_t nsteps, void *stream,
                          const int32_t *gather) {
  if (!pl || !spec || !knots || !f) return set_err("rs_hip_expand_forcing: bad arguments");
  if (check_forcing(pl, f, "rs_hip_expand_forcing")) return -1;
  if (nsteps < 1 || nsteps > 65535 || t0 < 1)
    return set_err("rs_hip_expand_forcing: 1 <= nsteps <= 65535, t0 >= 1");
  const int32_t spk = spec->steps_per_knot;
  if (spk < 1) return set_err("rs_hip_expand_forcing: steps_per_knot < 1");
  const int32_t kfirst = (t0 - 1) / spk;
  const int32_t tlast = t0 + nsteps - 2; /* 0-based */
  const int32_t klast = tlast / spk + ((tlast % spk) ? 1 : 0);
  if (kfirst < k0 || klast >= k0 + nknots)
    return set_err("rs_hip_expand_forcing: need knots %d..%d, buffer has %d..%d", kfirst, klast,
                   k0, k0 + nknots - 1);
  if (f->hour_pstride) return set_err("rs_hip_expand_forcing: hour must be a shared axis");
  HIP_OK(hipSetDevice(pl->device));
  rs::ExpandArgs a;
  a.f = *f;
  a.knots = knots;
  a.npoints = pl->npoints;
  a.np_pad = pl->np_pad;
  a.k0 = k0;
  a.t0 = t0;
  a.spk = spk;
  a.start_hour = spec->start_hour;
  a.kfirst = kfirst;
  a.nsteps = nsteps;
  a.r_spk = 1.0 / (double)spk;
  a.gather = gather;
  if (pl->f32) {
    if (f->depth) return set_err("rs_hip_expand_forcing: fp32 windows carry no depth stream");
    HIP_OK(rs32_launch_expand(a, (t0 + nsteps - 2) / spk - kfirst + 1, (hipStream_t)stream));
  } else {
    HIP_OK(rs_launch_expand(a, (t0 + nsteps - 2) / spk - kfirst + 1, (hipStream_t)stream));
  }
  return 0;
}

int rs_hip_expand_forcing_on(RsPlan *pl, const RsSynthSpec *spec, const double *knots, int32_t k0,
                             int32_t nknots, const RsForcing *f, int32_t t0, int32_t nsteps,
                             void *stream) {
  return expand_forcing(pl, spec, knots, k0, nknots, f, t0, nsteps, stream, nullptr);
}

int rs_hip_expand_forcing(RsPlan *pl, const RsSynthSpec *spec, const double *knots, int32_t k0,
                          int32_t nknots, const RsForcing *f, int32_t t0, int32_t nsteps) {
  if (!pl) return set_err("rs_hip_expand_forcing: bad arguments");
  return expand_forcing(pl, spec, knots, k0, nknots, f, t0, nsteps, pl->stream, nullptr);
}

int rs_hip_expand_forcing_ordered(RsPlan *pl, const RsSynthSpec *spec, const double *knots, int32_t k0,
                                  int32_t nknots, const RsForcing *f, int32_t t0, int32_t nsteps) {
  if (!pl) return set_err("rs_hip_expand_forcing_ordered: bad arguments");
  const int32_t *order = rs_hip_plan_order(pl);
  if (!order) return -1;
  return expand_forcing(pl, spec, knots, k0, nknots, f, t0, nsteps, pl->stream, order);
}

} /* extern "C" */
