/*
 * rs_compat.hip - the device side of `module RoadSurf`'s per-step procedures
 * (roadsurf_amd/fortran/RoadSurfCompat.f90; reference: /root/reference/src/RoadSurf.f90:6-270).
 *
 * The reference's own time loop (examples/example1/src/Simulation.f90:57-115) drives ONE point through
 * fourteen module procedures per time index.  Behind this library a time index of a point is one fused
 * step kernel; the compatibility module keeps the reference's procedure signatures and state types and
 * maps them onto the device API one point and one index at a time:
 *
 *   Initialization        rs_compat_begin   plan (one point, cached per thread), per-point parameters,
 *                                           init_kernel from forcing index 1
 *   BalanceModelOneStep   rs_compat_step    the caller's forcing of index i -> a one-row window ->
 *                                           rs_hip_step / rs_hip_step_cpl with nsteps = 1 -> the state
 *                                           column back into the Fortran types
 *   CheckEndCoupling      rs_compat_replay  at the end of the coupling window: every replay of the window
 *                                           on the device (rs_hip_cpl_replay), rewritten output rows back
 *                                           into the caller's arrays
 *   (scope exit of SurfaceVariables: FINAL)  rs_compat_end
 *
 * A compatibility path, not a fast one: a kernel launch, an upload and a state download per time index
 * (INTEGRATION.md section 2 has the measured cost).  The boundaries to use are runsimulation_batch and
 * rs_driver_run.
 */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/roadsurf.h"
#include "rs_devutil.hpp"
#include "rs_state.h"

extern "C" void rs_host_set_error(const char *msg);

namespace {

constexpr int NF64 = 11; /* tair, tdew, vz, rhz, prec, sw, lw, sw_dir, lw_net, tsurfobs, depth */

struct Ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  RsPlan *plan = nullptr;
  RsConstants consts{};
  int64_t np_pad = 0;
  int32_t L = 0;
  bool coupled = false, sky = false, relax = false;
  RsCompatArrays arr{};
  /* device */
  double *d_row = nullptr;   /* one index: [NF64 doubles][PrecPhase, hour: 2 int32][3 write-back cells] */
  double *d_pp = nullptr;    /* per-point parameter rows, [k][np_pad] */
  double *d_hz = nullptr;    /* [360] */
  double *d_sun = nullptr;   /* [L][RS_SUN_COLS] */
  double *d_out = nullptr;   /* [6][rows_cap] output rows (one row per step; the coupling window in a replay) */
  double *d_win = nullptr;   /* replay window: [NF64][win_cap] + int32 [2][win_cap] */
  int32_t win_cap = 0, rows_cap = 0;
  /* pinned */
  double *h_row = nullptr;   /* staging of d_row */
  double *h_state = nullptr; /* [RS_NSTATE][np_pad] */
  double *h_win = nullptr;
  int32_t cs = 0, ce = 0;    /* coupling window of the point, 0 = none */
  int32_t failed_at = 0;
  double albedo_surr = 0.0;  /* InputParameters.Albedo_surroundings (sky view) */
  double last[RS_NSTATE] = {}; /* the state column as last downloaded */
  double last_out[6] = {};     /* the output row the last step wrote (Tsurf, Snow, Water, Ice, Deposit, Ice2) ... */
  int32_t last_out_i = 0;      /* ... and its index; 0: none (the point had failed before: no row) */
  double *h_out = nullptr;     /* pinned, [6] */
  /* the lock-step replay kernels step the index BEHIND the window too once a point's last replay is through
   * (the launch covers it: its forcing is what the rewind's CheckValues reads): its row, kept for the step
   * call of that index, which then finds the point already there */
  int32_t ahead_i = 0;
  double ahead_out[6] = {};
};

/* Contexts (plan, stream, buffers) are kept for the next point: a free list of the PROCESS (round 6; it was
 * thread-local and never released when its thread ended - a worker pool that starts and ends threads leaked a
 * plan and a stream per thread).  A handle is not the context's address but a number that is never
 * issued twice: a derived-type COPY of the Fortran handle's owner makes two finalizers release the same handle -
 * the second finds it gone, even when the context behind it has been handed to another point since.  Leaked heap
 * singletons: a FINAL procedure may run after static destruction. */
struct Registry {
  std::mutex m;
  std::vector<Ctx *> free_list;                      /* at most kKeep contexts */
  std::vector<std::pair<uint64_t, Ctx *>> live;      /* handles handed out and not yet given back */
  uint64_t next = 1;
};
constexpr size_t kKeep = 16;
Registry &reg() {
  static Registry *r = new Registry();
  return *r;
}
Ctx *ctx_of(RsCompat *h) {
  const uint64_t id = (uint64_t)reinterpret_cast<uintptr_t>(h);
  Registry &r = reg();
  std::lock_guard<std::mutex> lk(r.m);
  for (auto &e : r.live)
    if (e.first == id) return e.second;
  return nullptr;
}

/* the state slots module RoadSurf (RoadSurfCompat.f90) reads by number */
static_assert(RS_ST_TNW1 == 32 && RS_ST_TNW2 == 33 && RS_ST_TSURF == 34 && RS_ST_WAT == 35 && RS_ST_SNOW == 36 &&
              RS_ST_ICE == 37 && RS_ST_ICE2 == 38 && RS_ST_DEP == 39 && RS_ST_Q2MELT == 40 && RS_ST_T4MELT == 41 &&
              RS_ST_ALBEDO == 42 && RS_ST_VERYCOLD == 43 && RS_ST_FAILED == 44 && RS_ST_TAIR_END == 45 &&
              RS_ST_VZ_END == 46 && RS_ST_RH_END == 47 && RS_ST_CPL_ITER == 49 && RS_ST_CPL_FLAGS == 50 &&
              RS_ST_CPL_TABOVE == 51 && RS_ST_CPL_TBELOW == 52 && RS_ST_CPL_RADCOEFF == 53 &&
              RS_ST_CPL_RCABOVE == 54 && RS_ST_CPL_RCBELOW == 55 && RS_ST_CPL_RCPREV == 56 &&
              RS_ST_CPL_SWCOF == 57 && RS_ST_CPL_LWCOF == 58 && RS_ST_CPL_SWCORR == 59 &&
              RS_ST_CPL_LWCORR == 60 && RS_ST_CPL_TEND1 == 61 && RS_ST_CPL_LASTOBS == 62 && RS_NSTATE == 134 &&
              RS_NSTATE == RS_COMPAT_NSTATE,
              "roadsurf_amd/fortran/RoadSurfCompat.f90 reads the state column by these numbers");

int fail(const char *msg) {
  rs_host_set_error(msg);
  return -1;
}
#define COK(expr)                                                         \
  do {                                                                    \
    hipError_t e_ = (expr);                                               \
    if (e_ != hipSuccess) {                                               \
      char b_[256];                                                       \
      snprintf(b_, sizeof(b_), "rs_compat: %s: %s", #expr, hipGetErrorString(e_)); \
      rs_host_set_error(b_);                                              \
      return -10;                                                         \
    }                                                                     \
  } while (0)

void destroy(Ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->plan) rs_hip_plan_destroy(c->plan);
  for (double *p : {c->d_row, c->d_pp, c->d_hz, c->d_sun, c->d_out, c->d_win})
    if (p) (void)hipFree(p);
  for (double *p : {c->h_row, c->h_state, c->h_win, c->h_out})
    if (p) (void)hipHostFree(p);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

/* the state column of the point, [RS_NSTATE], from the device */
int pull_state(Ctx *c, double *state_out) {
  const size_t bytes = (size_t)RS_NSTATE * c->np_pad * sizeof(double);
  if (rs_hip_state_download(c->plan, c->h_state, bytes) != 0) return -1;
  for (int k = 0; k < RS_NSTATE; ++k) c->last[k] = state_out[k] = c->h_state[(size_t)k * c->np_pad];
  c->failed_at = (int32_t)state_out[RS_ST_FAILED];
  return 0;
}

/* per-point parameter rows in d_pp */
enum { PP_TBOT, PP_TAIRR, PP_VZR, PP_RHR, PP_CPLT, PP_SKY, PP_SINLAT, PP_COSLAT, PP_LON, PP_INT /* initlen, cpl index */, PP_ROWS };

void point_params(const Ctx *c, RsPointParams &pp) {
  std::memset(&pp, 0, sizeof(pp));
  const int64_t np = c->np_pad;
  pp.tbottom = c->d_pp + PP_TBOT * np;
  pp.initlen = reinterpret_cast<const int32_t *>(c->d_pp + PP_INT * np);
  if (c->relax) {
    pp.tair_relax = c->d_pp + PP_TAIRR * np;
    pp.vz_relax = c->d_pp + PP_VZR * np;
    pp.rh_relax = c->d_pp + PP_RHR * np;
  }
  if (c->coupled) {
    pp.coupling_index = reinterpret_cast<const int32_t *>(c->d_pp + PP_INT * np) + np;
    pp.coupling_tsurf = c->d_pp + PP_CPLT * np;
  }
  if (c->sky) {
    pp.sky_view = c->d_pp + PP_SKY * np;
    pp.sin_lat = c->d_pp + PP_SINLAT * np;
    pp.cos_lat = c->d_pp + PP_COSLAT * np;
    pp.lon_rad = c->d_pp + PP_LON * np;
    pp.horizons = c->d_hz;
    pp.horizons_by_point = 1;
  }
}

/* a window of `n` indices from index i0 (1-based) of the caller's series, in `dst` (device) through `h` (pinned) */
void fill_window(const Ctx *c, double *h, int32_t i0, int32_t n, int32_t cap) {
  const RsCompatArrays &a = c->arr;
  const double *src[NF64] = {a.tair, a.tdew, a.vz, a.rhz, a.prec, a.sw, a.lw, a.sw_dir, a.lw_net, a.tsurfobs, a.depth};
  for (int f = 0; f < NF64; ++f)
    for (int32_t k = 0; k < n; ++k) h[(size_t)f * cap + k] = src[f] ? src[f][i0 - 1 + k] : -9999.9;
  int32_t *hi = reinterpret_cast<int32_t *>(h + (size_t)NF64 * cap);
  for (int32_t k = 0; k < n; ++k) {
    hi[k] = a.precphase ? a.precphase[i0 - 1 + k] : -9999;
    hi[cap + k] = a.hour[i0 - 1 + k];
  }
}

void window_forcing(const Ctx *c, const double *d, int32_t cap, int32_t i0, bool with_depth, RsForcing &f) {
  std::memset(&f, 0, sizeof(f));
  f.tair = d;
  f.tdew = d + (size_t)1 * cap;
  f.vz = d + (size_t)2 * cap;
  f.rhz = d + (size_t)3 * cap;
  f.prec = d + (size_t)4 * cap;
  f.sw = d + (size_t)5 * cap;
  f.lw = d + (size_t)6 * cap;
  f.tsurfobs = d + (size_t)9 * cap;
  f.depth = with_depth ? d + (size_t)10 * cap : nullptr;
  const int32_t *di = reinterpret_cast<const int32_t *>(d + (size_t)NF64 * cap);
  f.precphase = di;
  f.hour = di + cap;
  f.t_stride = 1; /* one point: consecutive indices are consecutive elements */
  f.hour_pstride = 0;
  if (c->sky) {
    f.sw_dir = d + (size_t)7 * cap;
    f.lw_net = d + (size_t)8 * cap;
    f.sun = c->d_sun + (size_t)(i0 - 1) * RS_SUN_COLS;
  }
}

void outputs_at(const Ctx *c, int64_t row0, RsOutputs &o) {
  const size_t cap = (size_t)c->rows_cap;
  o.tsurf = c->d_out;
  o.snow = c->d_out + cap;
  o.water = c->d_out + 2 * cap;
  o.ice = c->d_out + 3 * cap;
  o.deposit = c->d_out + 4 * cap;
  o.ice2 = c->d_out + 5 * cap;
  o.t_stride = 1;
  o.decimate = 1;
  o.row0 = row0;
}

}  // namespace

extern "C" {

/* Initialization (src/Initialization.f90:65-147, device part).  consts: rs_build_constants of the settings
 * and parameters; tbottom: rs_bottom_temperature of the first date; arr: the caller's series (kept: the
 * steps read them index by index, so edits the caller makes between steps are honoured); sun: rs_sun_table
 * rows [SimLen][RS_SUN_COLS] and geo = {sin lat, cos lat, lon} where the point has a sky view, else NULL;
 * state_out [RS_NSTATE]: the point's state column as the initialization leaves it. */
RsCompat *rs_compat_begin(const RsConstants *consts, const LocalParameters *local, double tbottom,
                          const RsCompatArrays *arr, const double *sun, const double *geo,
                          double albedo_surroundings, double *state_out) {
  if (!consts || !local || !arr || !state_out || !arr->tair || !arr->hour) {
    fail("rs_compat_begin: bad arguments");
    return nullptr;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    fail("rs_compat_begin: no HIP device visible - this library has no CPU path");
    return nullptr;
  }
  /* the first entry of ROADSURF_HIP_DEVICES (the fan-out's list, rs_devices.hpp), or device 0 */
  const char *ed = getenv("ROADSURF_HIP_DEVICES");
  const int device = (ed && *ed >= '0' && *ed <= '9') ? atoi(ed) % ndev : 0;
  Ctx *c = nullptr;
  {
    Registry &r = reg();
    std::lock_guard<std::mutex> lk(r.m);
    for (size_t k = 0; k < r.free_list.size(); ++k)
      if (r.free_list[k]->device == device && std::memcmp(&r.free_list[k]->consts, consts, sizeof(RsConstants)) == 0) {
        c = r.free_list[k];
        r.free_list.erase(r.free_list.begin() + (long)k);
        break;
      }
  }
  auto bail = [&](const char *msg) -> RsCompat * {
    if (msg) fail(msg);
    destroy(c);
    return nullptr;
  };
  if (hipSetDevice(device) != hipSuccess) return bail("rs_compat_begin: hipSetDevice failed");
  const int32_t L = consts->SimLen;
  const int32_t wcap = consts->use_coupling ? consts->cplLenI + 3 : 1;
  if (!c) {
    c = new Ctx();
    c->device = device;
    c->consts = *consts;
    c->L = L;
    if (hipStreamCreate(&c->stream) != hipSuccess) return bail("rs_compat_begin: hipStreamCreate failed");
    c->plan = rs_hip_plan_create(device, 1, consts, c->stream);
    if (!c->plan) return bail(nullptr);
    c->np_pad = rs_hip_plan_npoints_padded(c->plan);
    c->win_cap = wcap;
    c->rows_cap = wcap;
    const size_t rowb = (size_t)(NF64 + 3) * sizeof(double) + 2 * sizeof(int32_t) + 16;
    const size_t winb = (size_t)NF64 * wcap * sizeof(double) + (size_t)2 * wcap * sizeof(int32_t) + 16;
    if (hipMalloc(&c->d_row, rowb) != hipSuccess || hipHostMalloc(&c->h_row, rowb) != hipSuccess ||
        hipMalloc(&c->d_pp, (size_t)(PP_ROWS + 1) * c->np_pad * sizeof(double)) != hipSuccess ||
        hipMalloc(&c->d_hz, 360 * sizeof(double)) != hipSuccess ||
        hipMalloc(&c->d_sun, (size_t)L * RS_SUN_COLS * sizeof(double)) != hipSuccess ||
        hipMalloc(&c->d_out, (size_t)6 * wcap * sizeof(double)) != hipSuccess ||
        hipMalloc(&c->d_win, winb) != hipSuccess || hipHostMalloc(&c->h_win, winb + (size_t)6 * wcap * sizeof(double)) != hipSuccess ||
        hipHostMalloc(&c->h_out, 6 * sizeof(double)) != hipSuccess ||
        hipHostMalloc(&c->h_state, (size_t)RS_NSTATE * c->np_pad * sizeof(double)) != hipSuccess)
      return bail("rs_compat_begin: out of device or page-locked memory");
  }
  c->arr = *arr;
  c->coupled = consts->use_coupling != 0;
  c->relax = consts->use_relaxation != 0;
  c->sky = sun && geo && local->sky_view < 1.0 && local->sky_view > (double)-0.01f;
  c->failed_at = 0;
  c->last_out_i = 0;
  c->ahead_i = 0;
  c->cs = c->ce = 0;
  if (c->coupled && !(local->couplingTsurf < -100) && local->couplingIndexI >= 1) { /* src/InputOutput.f90:34-36 */
    c->ce = local->couplingIndexI;
    c->cs = ((double)c->ce <= consts->cplLenR) ? 1 : c->ce - consts->cplLenI; /* initCouplingTimes, src/Coupling.f90:512-517 */
  }
  /* per-point parameters: rows of np_pad (only column 0 is a point) */
  {
    const int64_t np = c->np_pad;
    std::vector<double> h((size_t)(PP_ROWS + 1) * np, 0.0);
    h[PP_TBOT * np] = tbottom;
    h[PP_TAIRR * np] = local->tair_relax;
    h[PP_VZR * np] = local->VZ_relax;
    h[PP_RHR * np] = local->RH_relax;
    h[PP_CPLT * np] = local->couplingTsurf;
    for (int64_t k = 0; k < np; ++k) h[PP_SKY * np + k] = 1.0;
    h[PP_SKY * np] = local->sky_view;
    if (geo) {
      h[PP_SINLAT * np] = geo[0];
      h[PP_COSLAT * np] = geo[1];
      h[PP_LON * np] = geo[2];
    }
    int32_t *hi = reinterpret_cast<int32_t *>(&h[PP_INT * np]);
    hi[0] = local->InitLenI;
    hi[np] = local->couplingIndexI;
    if (hipMemcpyAsync(c->d_pp, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
      return bail("rs_compat_begin: upload of the point's parameters failed");
  }
  if (c->sky) {
    std::vector<double> hz(360, 0.0);
    if (arr->horizons) std::memcpy(hz.data(), arr->horizons, 360 * sizeof(double));
    if (hipMemcpyAsync(c->d_hz, hz.data(), 360 * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipMemcpyAsync(c->d_sun, sun, (size_t)L * RS_SUN_COLS * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
      return bail("rs_compat_begin: upload of the sky-view tables failed");
  }
  /* init_kernel from index 1 */
  fill_window(c, c->h_row, 1, 1, 1);
  RsForcing f;
  window_forcing(c, c->d_row, 1, 1, true, f);
  RsPointParams pp;
  point_params(c, pp);
  c->albedo_surr = albedo_surroundings;
  pp.albedo_surroundings = albedo_surroundings;
  const size_t rowb = (size_t)(NF64 + 1) * sizeof(double);
  if (hipMemcpyAsync(c->d_row, c->h_row, rowb, hipMemcpyHostToDevice, c->stream) != hipSuccess)
    return bail("rs_compat_begin: upload failed");
  if (rs_hip_init_state(c->plan, &f, &pp) != 0) return bail(nullptr);
  if (pull_state(c, state_out) != 0) return bail(nullptr);
  Registry &r = reg();
  std::lock_guard<std::mutex> lk(r.m);
  const uint64_t id = r.next++;
  r.live.emplace_back(id, c);
  return reinterpret_cast<RsCompat *>((uintptr_t)id);
}

/* One time index (BalanceModelOneStep and everything the fused kernel does with it: CheckValues,
 * SetCurrentValues, relaxation, precipitation, sky view, balance, wear, RoadCond, albedo; lastValues at
 * i = SimLen).  edits [3]: SW(i), SW_dir(i), LW(i) as the sky view leaves them (the reference edits the
 * caller's arrays in place, src/ModRadiation.f90:57-71) - written only where the point has a sky view and no
 * coupling.  Returns 0, or < 0 (rs_last_error). */
int rs_compat_step(RsCompat *h, int32_t i, double *state_out, double *edits) {
  Ctx *c = ctx_of(h);
  if (!c || !state_out || i < 1 || i > c->L) return fail("rs_compat_step: bad arguments");
  COK(hipSetDevice(c->device));
  if (c->coupled && (int32_t)c->last[RS_ST_CPL_RESUME] > i && c->failed_at == 0) {
    /* the point is past this index already (stepped inside the replay launch): nothing to run */
    std::memcpy(state_out, c->last, sizeof(c->last));
    c->last_out_i = 0;
    if (c->ahead_i == i) {
      std::memcpy(c->last_out, c->ahead_out, sizeof(c->last_out));
      c->last_out_i = i;
    }
    return 0;
  }
  fill_window(c, c->h_row, i, 1, 1);
  const bool wb = c->sky && !c->coupled && edits;
  /* write-back cells behind the row: pre-filled with the original values (rs_hip_set_writeback) */
  double *hw = c->h_row + NF64 + 1;
  if (wb) {
    hw[0] = c->h_row[5];
    hw[1] = c->h_row[7];
    hw[2] = c->h_row[6];
  }
  const size_t rowb = (size_t)(NF64 + 1 + 3) * sizeof(double);
  COK(hipMemcpyAsync(c->d_row, c->h_row, rowb, hipMemcpyHostToDevice, c->stream));
  RsForcing f;
  const bool with_depth = c->arr.depth && c->arr.depth[i - 1] >= 0.0; /* a missing stream reads -9999.9 */
  window_forcing(c, c->d_row, 1, i, with_depth, f);
  RsPointParams pp;
  point_params(c, pp);
  pp.albedo_surroundings = c->albedo_surr;
  RsOutputs o;
  outputs_at(c, (int64_t)i - 1, o);
  double *dw = c->d_row + NF64 + 1;
  if (rs_hip_set_writeback(c->plan, wb ? dw : nullptr, wb ? dw + 1 : nullptr, wb ? dw + 2 : nullptr, 1) != 0) return -1;
  const int rc = c->coupled ? rs_hip_step_cpl(c->plan, &f, &o, &pp, i, 1) : rs_hip_step(c->plan, &f, &o, &pp, i, 1);
  if (rc != 0) return -1;
  if (wb) COK(hipMemcpyAsync(hw, dw, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  /* the row SaveOutput stores: what the kernel wrote for this index - with coupling NOT the state's values at
   * the window end (Coupling_control's Kelvin round trip of TsurfAve comes behind SaveOutput in the
   * reference's loop, examples/example1/src/Simulation.f90:87-91) */
  const int32_t failed_before = c->failed_at;
  for (int k = 0; k < 6; ++k)
    COK(hipMemcpyAsync(c->h_out + k, c->d_out + (size_t)k * c->rows_cap, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (pull_state(c, state_out) != 0) return -1; /* (synchronises the stream) */
  c->last_out_i = failed_before > 0 ? 0 : i;
  for (int k = 0; k < 6; ++k) c->last_out[k] = c->h_out[k];
  if (wb) {
    edits[0] = hw[0];
    edits[1] = hw[1];
    edits[2] = hw[2];
  }
  return 0;
}

/* The end of the point's coupling window has been stepped (CheckEndCoupling at i = couplingEndI): every
 * replay the reference would run from here (src/Coupling.f90:61-78,292-481: up to 25 passes over the window)
 * on the device, the rewritten output rows straight into the caller's arrays.  rewritten[2]: first and last
 * index whose rows were rewritten, 0 0 if the point did not replay. */
int rs_compat_replay(RsCompat *h, int32_t i, double *state_out, int32_t *rewritten) {
  Ctx *c = ctx_of(h);
  if (!c || !state_out || !rewritten) return fail("rs_compat_replay: bad arguments");
  rewritten[0] = rewritten[1] = 0;
  if (!c->coupled || c->ce < 1 || i != c->ce) return pull_state(c, state_out);
  COK(hipSetDevice(c->device));
  const int32_t lo = c->cs, hi = c->ce + 1 <= c->L ? c->ce + 1 : c->L, n = hi - lo + 1;
  if (n > c->win_cap) return fail("rs_compat_replay: coupling window longer than the buffer");
  fill_window(c, c->h_win, lo, n, c->win_cap);
  const size_t winb = (size_t)NF64 * c->win_cap * sizeof(double) + (size_t)2 * c->win_cap * sizeof(int32_t);
  COK(hipMemcpyAsync(c->d_win, c->h_win, winb, hipMemcpyHostToDevice, c->stream));
  RsForcing f;
  window_forcing(c, c->d_win, c->win_cap, lo, true, f);
  RsPointParams pp;
  point_params(c, pp);
  pp.albedo_surroundings = c->albedo_surr;
  RsOutputs o;
  outputs_at(c, (int64_t)lo - 1, o);
  int32_t rounds = 0;
  if (rs_hip_set_writeback(c->plan, nullptr, nullptr, nullptr, 0) != 0) return -1;
  if (rs_hip_cpl_replay(c->plan, &f, &o, &pp, lo, n, &rounds) != 0) return -1;
  double *ho = c->h_win + ((size_t)NF64 * c->win_cap + (size_t)c->win_cap + 2); /* behind the window staging */
  if (rounds > 0) {
    COK(hipMemcpyAsync(ho, c->d_out, (size_t)6 * c->rows_cap * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  }
  if (pull_state(c, state_out) != 0) return -1;
  if (rounds > 0) {
    double *dst[6] = {c->arr.out[0], c->arr.out[1], c->arr.out[2], c->arr.out[3], c->arr.out[4], c->arr.out[5]};
    for (int k = 0; k < 6; ++k)
      if (dst[k])
        for (int32_t q = c->cs; q <= c->ce; ++q) dst[k][q - 1] = ho[(size_t)k * c->rows_cap + (q - lo)];
    rewritten[0] = c->cs;
    rewritten[1] = c->ce;
    if ((int32_t)c->last[RS_ST_CPL_RESUME] == c->ce + 2 && c->ce + 1 <= hi) {
      c->ahead_i = c->ce + 1;
      for (int k = 0; k < 6; ++k) c->ahead_out[k] = ho[(size_t)k * c->rows_cap + (c->ce + 1 - lo)];
    }
  }
  return 0;
}

/* the state column as the last step / replay left it (no device traffic) */
int rs_compat_last_state(const RsCompat *h, double *state_out) {
  const Ctx *c = ctx_of(const_cast<RsCompat *>(h));
  if (!c || !state_out) return fail("rs_compat_last_state: bad arguments");
  std::memcpy(state_out, c->last, sizeof(c->last));
  return 0;
}

/* the output row of index i as the step kernel wrote it: 0 and out[6] = Tsurf, Snow, Water, Ice, Deposit, Ice2,
 * or 1 if the last step was not index i (or wrote no row) */
int rs_compat_outputs(const RsCompat *h, int32_t i, double *out) {
  const Ctx *c = ctx_of(const_cast<RsCompat *>(h));
  if (!c || !out || c->last_out_i != i) return 1;
  std::memcpy(out, c->last_out, sizeof(c->last_out));
  return 0;
}

/* 0, or the 1-based index at which the device failed the point (CheckValues inside the fused step) */
int32_t rs_compat_failed_index(const RsCompat *h) {
  const Ctx *c = ctx_of(const_cast<RsCompat *>(h));
  return c ? c->failed_at : 0;
}

/* the point is finished: its plan, stream and buffers wait for the thread's next point */
void rs_compat_end(RsCompat *h) {
  const uint64_t id = (uint64_t)reinterpret_cast<uintptr_t>(h);
  Ctx *c = nullptr;
  bool keep = false;
  {
    Registry &r = reg();
    std::lock_guard<std::mutex> lk(r.m);
    size_t k = 0;
    while (k < r.live.size() && r.live[k].first != id) ++k;
    if (k == r.live.size()) return; /* released already (a copied handle's second finalizer), or never issued */
    c = r.live[k].second;
    r.live.erase(r.live.begin() + (long)k);
    keep = r.free_list.size() < kKeep;
    if (keep) r.free_list.push_back(c);
  }
  if (!keep) destroy(c);
}

} /* extern "C" */
