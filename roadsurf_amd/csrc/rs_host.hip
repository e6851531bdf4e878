/*
 * rs_host.hip — layer 2 of include/roadsurf.h: host-array batch entry used by
 * the Fortran `runsimulation_batch` (roadsurf_amd/fortran/RoadSurfHip.f90).
 *
 * The reference boundary hands over one contiguous [SimLen] array per point
 * and field (examples/example1/src/InputData.cpp:5-26, OutputData.cpp:5-13).
 * The kernels want points on the fastest axis.  So, per tile of points and per
 * chunk of time:
 *     host rows --memcpy--> pinned [field][point][t]  --H2D-->  device
 *     device: LDS-tiled transpose to [field][t][point]  -> step kernel ->
 *     transpose outputs back to [field][point][t]  --D2H-->  pinned -> rows
 * The carried state stays on the device between chunks.  Everything runs on
 * one stream per call; the rate of this path is PCIe/host-memcpy bound and is
 * quoted separately from the device-resident rate.
 */
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>
#include <omp.h>

#include "../../include/roadsurf.h"
#include "rs_kernels.h"
#include "rs_devutil.hpp"
#include "rs_devices.hpp"
#include "rs_state.h"

extern "C" void rs_host_set_error(const char *msg);
static int run_batch_on_device(int32_t n, OutputPointers *outPointers, const InputPointers *inPointers,
                               const RsConstants *consts, const LocalParameters *localParam,
                               const double *tbottom, const RsHostExtras *extras, int32_t device,
                               int nthreads);

namespace {

using rsu::Dev;
using rsu::Pinned;
using rsu::transpose;

int fail(const char *what, hipError_t e) {
  char buf[256];
  snprintf(buf, sizeof(buf), "rs_host_run_batch: %s: %s", what, hipGetErrorString(e));
  rs_host_set_error(buf);
  return -10;
}

#define HOK(expr)                                   \
  do {                                              \
    hipError_t e_ = (expr);                         \
    if (e_ != hipSuccess) return fail(#expr, e_);   \
  } while (0)

enum { F_TAIR, F_TDEW, F_VZ, F_RHZ, F_PREC, F_SW, F_LW, F_OBS, F_DEPTH, F_SWDIR, F_LWNET, NF64 };

inline const double *in_f64(const InputPointers &ip, int f) {
  switch (f) {
    case F_TAIR: return ip.c_tair;
    case F_TDEW: return ip.c_tdew;
    case F_VZ: return ip.c_VZ;
    case F_RHZ: return ip.c_Rhz;
    case F_PREC: return ip.c_prec;
    case F_SW: return ip.c_SW;
    case F_LW: return ip.c_LW;
    case F_OBS: return ip.c_TSurfObs;
    case F_SWDIR: return ip.c_SW_dir;
    case F_LWNET: return ip.c_LW_net;
    default: return ip.c_Depth;
  }
}
inline double *out_f64(const OutputPointers &op, int f) {
  switch (f) {
    case 0: return op.c_TsurfOut;
    case 1: return op.c_SnowOut;
    case 2: return op.c_WaterOut;
    case 3: return op.c_IceOut;
    case 4: return op.c_DepositOut;
    default: return op.c_Ice2Out;
  }
}

/* What a caller thread keeps between SMALL batches (round 5): the reference driver calls
 * runsimulation once per point from `-j` worker threads (examples/example1/src/roadrunner.cpp:454-497), and a
 * one-point call used to create a stream, two events, nine page-locked and a dozen device buffers and a plan -
 * and free them again, every hipFree waiting for the whole device: 16 callers got 25 points/s out of a GPU on
 * which one got 28.  Now the thread's stream, events and two blocks (device, page-locked) survive the call;
 * the buffers of a call are carved out of the blocks (rs_devutil.hpp: Arena), the plan's too. */
struct CallerCacheData {
  int device = -1;
  hipStream_t stream = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  rsu::Arena dev, pin;
};
/* caches of threads that have ended: adopted by the next thread that needs one (a thread's end - or the
 * process's, when the runtime may be gone already - is no place for HIP calls) */
/* (leaked heap singletons, like rs_coalesce.hip's runner: a thread that ends after static destruction - a detached
 * or OpenMP worker at process exit - must still find them) */
std::mutex &orphans_mutex() {
  static std::mutex *m = new std::mutex();
  return *m;
}
std::vector<CallerCacheData> &orphans() {
  static std::vector<CallerCacheData> *v = new std::vector<CallerCacheData>();
  return *v;
}

struct CallerCache : CallerCacheData {
  ~CallerCache() {
    if (device < 0) return;
    std::lock_guard<std::mutex> lk(orphans_mutex());
    orphans().push_back(static_cast<const CallerCacheData &>(*this));
  }
  void adopt(int d) {
    std::lock_guard<std::mutex> lk(orphans_mutex());
    std::vector<CallerCacheData> &o = orphans();
    for (size_t k = 0; k < o.size(); ++k)
      if (o[k].device == d) {
        static_cast<CallerCacheData &>(*this) = o[k];
        o.erase(o.begin() + (long)k);
        return;
      }
  }
  void drop() {
    if (device < 0) return;
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    if (dev.base) (void)hipFree(dev.base);
    if (pin.base) (void)hipHostFree(pin.base);
    for (hipEvent_t &e : ev)
      if (e) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
    static_cast<CallerCacheData &>(*this) = CallerCacheData();
  }
  /* blocks of at least these sizes on `d`; false: the call allocates the old way */
  bool reserve(int d, size_t dev_bytes, size_t pin_bytes) {
    if (device != d) drop();
    if (device < 0) adopt(d);
    if (device < 0) {
      if (hipStreamCreate(&stream) != hipSuccess) return false;
      if (hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) != hipSuccess) {
        device = d;
        drop();
        return false;
      }
      device = d;
    }
    if (dev.cap < dev_bytes) {
      (void)hipStreamSynchronize(stream);
      if (dev.base) (void)hipFree(dev.base);
      dev = rsu::Arena();
      void *q = nullptr;
      if (hipMalloc(&q, dev_bytes) != hipSuccess) return false;
      dev.base = static_cast<char *>(q);
      dev.cap = dev_bytes;
    }
    if (pin.cap < pin_bytes) {
      (void)hipStreamSynchronize(stream);
      if (pin.base) (void)hipHostFree(pin.base);
      pin = rsu::Arena();
      void *q = nullptr;
      if (hipHostMalloc(&q, pin_bytes, hipHostMallocDefault) != hipSuccess) return false;
      pin.base = static_cast<char *>(q);
      pin.cap = pin_bytes;
    }
    dev.rewind(0);
    pin.rewind(0);
    return true;
  }
};
thread_local CallerCache t_cache;
constexpr size_t kCacheDevMax = (size_t)768 << 20, kCachePinMax = (size_t)256 << 20;

}  // namespace

extern "C" {

/* -1: the device list of rs_devices.hpp (ROADSURF_HIP_DEVICES, default every visible device) */
int rs_host_default_device(void) { return -1; }
int rs_last_fanout(void) { return rsu::g_last_fanout; }

int rs_host_run_batch(int32_t n, OutputPointers *outPointers, const InputPointers *inPointers,
                      const RsConstants *consts, const LocalParameters *localParam,
                      const double *tbottom, const RsHostExtras *extras, int32_t device) {
  if (n < 1 || !outPointers || !inPointers || !consts || !localParam || !tbottom) {
    rs_host_set_error("rs_host_run_batch: bad arguments");
    return -1;
  }
  const int host_threads = rsu::host_threads(omp_get_num_procs());
  if (device >= 0) {
    rsu::g_last_fanout = 1;
    return run_batch_on_device(n, outPointers, inPointers, consts, localParam, tbottom, extras, device,
                               host_threads);
  }
  /* fan out: one contiguous block of points per device, one host thread + stream + plan each */
  const std::vector<rsu::Shard> shards = rsu::make_shards(n, rsu::device_list());
  return rsu::fan_out(shards, [&](const rsu::Shard &sh, int nshards) {
    RsHostExtras ex;
    const RsHostExtras *exp = nullptr;
    if (extras) {
      ex = *extras; /* sun is a shared axis; the per-point arrays move with the block */
      if (ex.sin_lat) ex.sin_lat += sh.off;
      if (ex.cos_lat) ex.cos_lat += sh.off;
      if (ex.lon_rad) ex.lon_rad += sh.off;
      if (ex.first_failed) ex.first_failed += sh.off;
      if (ex.diagnostics) ex.diagnostics += (size_t)sh.off * RS_DIAG_COLS;
      exp = &ex;
    }
    return run_batch_on_device((int32_t)sh.cnt, outPointers + sh.off, inPointers + sh.off, consts,
                               localParam + sh.off, tbottom + sh.off, exp, sh.device,
                               std::max(1, host_threads / nshards));
  });
}

} /* extern "C" */

static int run_batch_on_device(int32_t n, OutputPointers *outPointers, const InputPointers *inPointers,
                               const RsConstants *consts, const LocalParameters *localParam,
                               const double *tbottom, const RsHostExtras *extras, int32_t device,
                               int nthreads) {
  const int L = consts->SimLen;
  /* tile sizes: bounded pinned staging (~0.5 GB) whatever n and SimLen are */
  const char *ep = getenv("ROADSURF_HIP_TILE_POINTS"), *et = getenv("ROADSURF_HIP_CHUNK_STEPS");
  /* with coupling a point replays its window, so the whole series is one window
   * (rs_hip_step enforces it) and the point tile shrinks to keep staging bounded */
  const bool coupled = consts->use_coupling != 0;
  const bool skyview = extras && extras->sun;
  const int nf64 = skyview ? NF64 : NF64 - 2; /* SW_dir / LW_net travel only for sky view */
  const int P = std::min<int64_t>(n, ep ? std::max(1, atoi(ep)) : (coupled ? 4096 : 16384));
  /* time indices per pipeline item.  Measured on an MI355X box (16 CPUs, tools/exp_host.sh,
   * 32 768 points x 48 h over four blocks): 256 -> 3.3e8, 128 -> 4.0e8, 64 -> 4.3e8, 32 -> 4.5e8,
   * 16 -> 4.2e8, 8 -> 2.8e8 point-timesteps/s: short items keep the staging sets in the host's
   * caches and the five-stage pipeline full; below 16 the per-item launches take over, and a small
   * block (a few thousand points: a latency-bound step kernel) is served as well by 64 */
  /* A handful of points (the reference driver calling runsimulation per point, or the batches the
   * coalescer makes of such calls): the step kernel is a few wavefronts' dependency chain whatever the
   * item length, and every item costs a fixed round of copies, transposes and launches - so the items
   * grow until one holds ~256 K values per variable (one point: the whole series in ONE item instead
   * of 91: 41 -> ~36 ms per call, and far fewer runtime calls for concurrent callers to queue behind). */
  const int TC = coupled ? L
                         : std::min(L, et ? std::max(1, atoi(et))
                                          : (n >= 6144 ? 32 : std::max(64, (int)(262144 / std::max<int64_t>(n, 1)))));
  const int Ppad = (P + RS_BLOCK - 1) / RS_BLOCK * RS_BLOCK;

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    rs_host_set_error("rs_host_run_batch: no HIP device visible - this library has no CPU path");
    return -9;
  }
  HOK(hipSetDevice(device));
  const size_t in_elems = (size_t)P * TC, tp_elems = (size_t)Ppad * TC;
  /* a small batch lives in the calling thread's cached blocks (CallerCache): no allocation, no free */
  const bool writeback_dev = extras && extras->writeback && skyview;
  size_t need_pin = in_elems * ((size_t)2 * nf64 * 8 + 2 * 6 * 8 + 2 * 2 * 4) + (writeback_dev ? in_elems * 6 * 8 : 0) +
                    (skyview ? (size_t)P * 360 * 8 : 0) + ((size_t)64 << 10);
  size_t need_dev = in_elems * ((size_t)nf64 * 8 + 2 * 4 + 6 * 8 + (writeback_dev ? 3 * 8 : 0)) +
                    tp_elems * ((size_t)nf64 * 8 + 2 * 4 + 6 * 8 + (writeback_dev ? 3 * 8 : 0)) +
                    (size_t)Ppad * (9 * 8 + 2 * 4) + (skyview ? (size_t)L * RS_SUN_COLS * 8 + (size_t)P * 360 * 8 : 0) +
                    (size_t)Ppad * ((size_t)2 * RS_NSTATE * 8 + 64) + (size_t)2 * L * 8 + ((size_t)4 << 20);
  need_pin += need_pin / 8;
  need_dev += need_dev / 8;
  const bool cached = need_dev <= kCacheDevMax && need_pin <= kCachePinMax &&
                      n <= P && t_cache.reserve(device, need_dev, need_pin);
  struct ArenaScope { /* the thread's blocks serve Dev::alloc / Pinned::alloc / plan_malloc during this call */
    rsu::Arena *pd, *pp;
    bool on;
    explicit ArenaScope(bool use) : pd(rsu::tls_arena()), pp(rsu::tls_pinned_arena()), on(use) {
      if (on) {
        rsu::tls_arena() = &t_cache.dev;
        rsu::tls_pinned_arena() = &t_cache.pin;
      }
    }
    ~ArenaScope() {
      if (on) {
        rsu::tls_arena() = pd;
        rsu::tls_pinned_arena() = pp;
      }
    }
  };
  /* declared before every buffer, so destroyed after them: error returns below leave work in
   * flight, and the stream must outlive it (buffers are released by hipFree, which waits) */
  struct StreamGuard {
    hipStream_t s = nullptr;
    bool own = true;
    ~StreamGuard() {
      if (s) {
        (void)hipStreamSynchronize(s);
        if (own) (void)hipStreamDestroy(s);
      }
    }
  } stream_guard;
  if (cached) {
    stream_guard.s = t_cache.stream;
    stream_guard.own = false;
  } else {
    HOK(hipStreamCreate(&stream_guard.s));
  }
  hipStream_t stream = stream_guard.s;
  ArenaScope arena_scope(cached); /* (after the stream guard: the buffers go first, then the stream drains) */

  Pinned h_in, h_out, h_i32;
  Dev d_pt, d_tp, d_i32pt, d_i32tp, d_out_tp, d_out_pt, d_pp64, d_pp32;
  HOK(h_in.alloc(in_elems * nf64 * sizeof(double)));
  HOK(h_out.alloc(in_elems * 6 * sizeof(double)));
  HOK(h_i32.alloc(in_elems * 2 * sizeof(int32_t)));
  HOK(d_pt.alloc(in_elems * nf64 * sizeof(double)));
  HOK(d_tp.alloc(tp_elems * nf64 * sizeof(double)));
  HOK(d_i32pt.alloc(in_elems * 2 * sizeof(int32_t)));
  HOK(d_i32tp.alloc(tp_elems * 2 * sizeof(int32_t)));
  HOK(d_out_tp.alloc(tp_elems * 6 * sizeof(double)));
  HOK(d_out_pt.alloc(in_elems * 6 * sizeof(double)));
  constexpr int NPP64 = 9; /* tbottom, 3 relaxation targets, couplingTsurf, sky_view, sin/cos lat, lon */
  HOK(d_pp64.alloc((size_t)Ppad * NPP64 * sizeof(double)));
  HOK(d_pp32.alloc((size_t)Ppad * 2 * sizeof(int32_t)));
  std::vector<double> pp64((size_t)Ppad * NPP64);
  std::vector<int32_t> pp32((size_t)Ppad * 2);
  const bool writeback = extras && extras->writeback;
  int32_t *first_failed = extras ? extras->first_failed : nullptr;
  double *diagnostics = extras ? extras->diagnostics : nullptr; /* [n][RS_DIAG_COLS]: the plans run with rs_hip_set_diagnostics */
  /* sky view + writeback: SW, SW_dir, LW as the reference leaves them come back from the device
   * (rs_hip_set_writeback); [3][Ppad*TC] windows, their [3][P*TC] transposes, two pinned sets */
  const bool wb_dev = writeback && skyview;
  Dev d_wb_tp, d_wb_pt;
  Pinned h_wb, h_wb2;
  if (wb_dev) {
    HOK(d_wb_tp.alloc(tp_elems * 3 * sizeof(double)));
    HOK(d_wb_pt.alloc(in_elems * 3 * sizeof(double)));
    HOK(h_wb.alloc(in_elems * 3 * sizeof(double)));
    HOK(h_wb2.alloc(in_elems * 3 * sizeof(double)));
  }
  double *hwb_b[2] = {(double *)h_wb.p, (double *)h_wb2.p};
  std::vector<int32_t> ff_tile((size_t)P);
  Dev d_sun, d_hz_pt;
  Pinned h_hz;
  if (skyview) {
    HOK(d_sun.alloc((size_t)L * RS_SUN_COLS * sizeof(double)));
    HOK(hipMemcpyAsync(d_sun.p, extras->sun, (size_t)L * RS_SUN_COLS * sizeof(double), hipMemcpyHostToDevice,
                       stream));
    HOK(h_hz.alloc((size_t)P * 360 * sizeof(double)));
    HOK(d_hz_pt.alloc((size_t)P * 360 * sizeof(double)));
  }

  /* ---- work items: (tile of points) x (chunk of time), processed as a two-stage pipeline.
   * While the GPU works on item k (H2D, transposes, kernels, D2H on `stream`), the host
   * scatters the outputs of item k-1 to the caller's rows and gathers the inputs of item
   * k+1 into the other pinned staging set. */
  struct Item {
    int64_t p0;
    int m, t0, len;
    bool first, last; /* of its tile */
  };
  std::vector<Item> items;
  for (int64_t p0 = 0; p0 < n; p0 += P)
    for (int t0 = 1; t0 <= L; t0 += TC) {
      const int len = std::min(TC, L - t0 + 1);
      items.push_back(Item{p0, (int)std::min<int64_t>(P, n - p0), t0, len, t0 == 1, t0 + len > L});
    }
  Pinned h_in2, h_out2, h_i322;
  HOK(h_in2.alloc(in_elems * nf64 * sizeof(double)));
  HOK(h_out2.alloc(in_elems * 6 * sizeof(double)));
  HOK(h_i322.alloc(in_elems * 2 * sizeof(int32_t)));
  double *hin_b[2] = {(double *)h_in.p, (double *)h_in2.p};
  double *hout_b[2] = {(double *)h_out.p, (double *)h_out2.p};
  int32_t *hi_b[2] = {(int32_t *)h_i32.p, (int32_t *)h_i322.p};
  hipEvent_t done[2] = {nullptr, nullptr};
  struct EventGuard {
    hipEvent_t *e;
    bool own;
    ~EventGuard() {
      if (!own) return;
      if (e[0]) (void)hipEventDestroy(e[0]);
      if (e[1]) (void)hipEventDestroy(e[1]);
    }
  } event_guard{done, !cached};
  if (cached) {
    done[0] = t_cache.ev[0];
    done[1] = t_cache.ev[1];
  } else {
    HOK(hipEventCreateWithFlags(&done[0], hipEventDisableTiming));
    HOK(hipEventCreateWithFlags(&done[1], hipEventDisableTiming));
  }

  /* (a team of OpenMP threads only where the rows are worth it: a caller thread of the reference driver that
   * brings a few points would otherwise raise - and keep - a team of its own) */
  [[maybe_unused]] const bool omp_rows = n >= 8;
  [[maybe_unused]] const int row_threads = std::max(1, std::min(nthreads, (int)(n / 4)));
  /* an item none of whose points has an output depth (depth(i) >= 0) needs no depth stream: the kernels read a
   * missing stream as -9999.9, and without one the launch can take the two-wavefront flavour (rs_hip_step) */
  int item_depth[2] = {1, 1};
  auto gather = [&](const Item &it, int buf) {
    double *hin = hin_b[buf];
    int32_t *hi = hi_b[buf];
    const int m = it.m, len = it.len, t0 = it.t0;
    const int64_t p0 = it.p0;
    int any_depth = 0;
#pragma omp parallel for schedule(static) num_threads(row_threads) reduction(| : any_depth) if (omp_rows)
    for (int p = 0; p < m; ++p) {
      const InputPointers &ip = inPointers[p0 + p];
      for (int f = 0; f < nf64; ++f)
        std::memcpy(hin + ((size_t)f * m + p) * len, in_f64(ip, f) + (t0 - 1),
                    (size_t)len * sizeof(double));
      std::memcpy(hi + (size_t)p * len, ip.c_PrecPhase + (t0 - 1), (size_t)len * sizeof(int32_t));
      std::memcpy(hi + ((size_t)m + p) * len, ip.c_hour + (t0 - 1), (size_t)len * sizeof(int32_t));
      const double *dp = ip.c_Depth + (t0 - 1);
      int d = 0;
      for (int t = 0; t < len; ++t) d |= (dp[t] >= 0.0) ? 1 : 0;
      any_depth |= d;
    }
    item_depth[buf] = any_depth;
  };
  auto scatter = [&](const Item &it, int buf) {
    const double *hout = hout_b[buf];
    const int m = it.m, len = it.len, t0 = it.t0;
    const int64_t p0 = it.p0;
#pragma omp parallel for schedule(static) num_threads(row_threads) if (omp_rows)
    for (int p = 0; p < m; ++p) {
      for (int f = 0; f < 6; ++f)
        std::memcpy(out_f64(outPointers[p0 + p], f) + (t0 - 1), hout + ((size_t)f * m + p) * len,
                    (size_t)len * sizeof(double));
      if (wb_dev) { /* the reference edits these "const" inputs in place (src/ModRadiation.f90:57-71) */
        const InputPointers &ip = inPointers[p0 + p];
        const double *hw = hwb_b[buf];
        std::memcpy(const_cast<double *>(ip.c_SW) + (t0 - 1), hw + ((size_t)0 * m + p) * len, (size_t)len * sizeof(double));
        std::memcpy(const_cast<double *>(ip.c_SW_dir) + (t0 - 1), hw + ((size_t)1 * m + p) * len, (size_t)len * sizeof(double));
        std::memcpy(const_cast<double *>(ip.c_LW) + (t0 - 1), hw + ((size_t)2 * m + p) * len, (size_t)len * sizeof(double));
      }
    }
  };
  /* end of a tile: per-point failure index, and - without sky view - the one in-place edit the
   * reference makes to its inputs, on the host: CheckValues ran for every index up to the one
   * that failed the point (or SimLen-1) and clamped SW_dir there (src/InputOutput.f90:75-77) */
  auto finish_tile = [&](RsPlan *pl, int64_t p0, int m) -> int {
    if (diagnostics && rs_hip_diagnostics(pl, diagnostics + (size_t)p0 * RS_DIAG_COLS) != 0) return -15;
    if (!first_failed && !(writeback && !wb_dev)) return 0;
    if (rs_hip_first_failed_index(pl, ff_tile.data()) != 0) return -15;
    if (first_failed) std::memcpy(first_failed + p0, ff_tile.data(), (size_t)m * sizeof(int32_t));
    if (writeback && !wb_dev) {
#pragma omp parallel for schedule(static) num_threads(nthreads) if (omp_rows)
      for (int p = 0; p < m; ++p) {
        const InputPointers &ip = inPointers[p0 + p];
        if (!ip.c_SW_dir) continue;
        const int lim = ff_tile[p] > 0 ? std::min(ff_tile[p], L - 1) : L - 1;
        double *sd = const_cast<double *>(ip.c_SW_dir);
        for (int t = 0; t < lim; ++t)
          if (sd[t] > ip.c_SW[t]) sd[t] = ip.c_SW[t];
      }
    }
    return 0;
  };

  RsPlan *plan = nullptr, *retired = nullptr; /* retired: its last item is still in flight */
  int64_t retired_p0 = 0;
  int retired_m = 0;
  struct PlanGuard {
    RsPlan **a, **b;
    ~PlanGuard() {
      if (*a) rs_hip_plan_destroy(*a);
      if (*b) rs_hip_plan_destroy(*b);
    }
  } plan_guard{&plan, &retired};
  RsPointParams pp;
  std::memset(&pp, 0, sizeof(pp));
  int64_t mp = 0;
  int rc = 0;

  /* everything the GPU does for one item, enqueued on `stream` */
  auto enqueue = [&](const Item &it, int buf) -> int {
    const int m = it.m, len = it.len, t0 = it.t0;
    const int64_t p0 = it.p0;
    if (it.first) {
      plan = rs_hip_plan_create(device, m, consts, stream);
      if (!plan) return -11; /* rs_last_error() is set */
      mp = rs_hip_plan_npoints_padded(plan);
      for (int p = 0; p < m; ++p) {
        pp64[p] = tbottom[p0 + p];
        pp64[(size_t)Ppad + p] = localParam[p0 + p].tair_relax;
        pp64[(size_t)2 * Ppad + p] = localParam[p0 + p].VZ_relax;
        pp64[(size_t)3 * Ppad + p] = localParam[p0 + p].RH_relax;
        pp64[(size_t)4 * Ppad + p] = localParam[p0 + p].couplingTsurf;
        pp64[(size_t)5 * Ppad + p] = localParam[p0 + p].sky_view;
        if (skyview) {
          pp64[(size_t)6 * Ppad + p] = extras->sin_lat[p0 + p];
          pp64[(size_t)7 * Ppad + p] = extras->cos_lat[p0 + p];
          pp64[(size_t)8 * Ppad + p] = extras->lon_rad[p0 + p];
          std::memcpy((double *)h_hz.p + (size_t)p * 360, inPointers[p0 + p].c_local_horizons,
                      360 * sizeof(double));
        }
        pp32[p] = localParam[p0 + p].InitLenI;
        pp32[(size_t)Ppad + p] = localParam[p0 + p].couplingIndexI;
      }
      HOK(hipMemcpyAsync(d_pp64.p, pp64.data(), pp64.size() * sizeof(double),
                         hipMemcpyHostToDevice, stream));
      HOK(hipMemcpyAsync(d_pp32.p, pp32.data(), pp32.size() * sizeof(int32_t),
                         hipMemcpyHostToDevice, stream));
      pp.tbottom = (double *)d_pp64.p;
      pp.tair_relax = (double *)d_pp64.p + Ppad;
      pp.vz_relax = (double *)d_pp64.p + 2 * (size_t)Ppad;
      pp.rh_relax = (double *)d_pp64.p + 3 * (size_t)Ppad;
      pp.initlen = (int32_t *)d_pp32.p;
      pp.coupling_tsurf = coupled ? (double *)d_pp64.p + 4 * (size_t)Ppad : nullptr;
      pp.coupling_index = coupled ? (int32_t *)d_pp32.p + Ppad : nullptr;
      pp.sky_view = pp.sin_lat = pp.cos_lat = pp.lon_rad = pp.horizons = nullptr;
      pp.horizons_by_point = 0;
      pp.albedo_surroundings = 0.0;
      if (skyview) {
        pp.sky_view = (double *)d_pp64.p + 5 * (size_t)Ppad;
        pp.sin_lat = (double *)d_pp64.p + 6 * (size_t)Ppad;
        pp.cos_lat = (double *)d_pp64.p + 7 * (size_t)Ppad;
        pp.lon_rad = (double *)d_pp64.p + 8 * (size_t)Ppad;
        pp.albedo_surroundings = extras->albedo_surroundings;
        /* horizon table [point][360], as the rows come (RsPointParams::horizons_by_point) */
        HOK(hipMemcpyAsync(d_hz_pt.p, h_hz.p, (size_t)m * 360 * sizeof(double), hipMemcpyHostToDevice,
                           stream));
        pp.horizons = (double *)d_hz_pt.p;
        pp.horizons_by_point = 1;
        /* h_hz is reused by the next tile: its copy must have left the host first */
        HOK(hipStreamSynchronize(stream));
      }
    }
    HOK(hipMemcpyAsync(d_pt.p, hin_b[buf], (size_t)nf64 * m * len * sizeof(double),
                       hipMemcpyHostToDevice, stream));
    HOK(hipMemcpyAsync(d_i32pt.p, hi_b[buf], (size_t)2 * m * len * sizeof(int32_t),
                       hipMemcpyHostToDevice, stream));
    for (int f = 0; f < nf64; ++f)
      HOK(transpose((const double *)d_pt.p + (size_t)f * m * len,
                    (double *)d_tp.p + (size_t)f * mp * TC, m, len, len, mp, stream));
    for (int f = 0; f < 2; ++f)
      HOK(transpose((const int32_t *)d_i32pt.p + (size_t)f * m * len,
                    (int32_t *)d_i32tp.p + (size_t)f * mp * TC, m, len, len, mp, stream));
    RsForcing fo;
    double *b = (double *)d_tp.p;
    const size_t fs = (size_t)mp * TC;
    fo.tair = b + F_TAIR * fs; fo.tdew = b + F_TDEW * fs; fo.vz = b + F_VZ * fs;
    fo.rhz = b + F_RHZ * fs; fo.prec = b + F_PREC * fs; fo.sw = b + F_SW * fs;
    fo.lw = b + F_LW * fs; fo.tsurfobs = b + F_OBS * fs;
    fo.depth = item_depth[buf] ? b + F_DEPTH * fs : nullptr;
    fo.precphase = (int32_t *)d_i32tp.p;
    fo.hour = (int32_t *)d_i32tp.p + fs;
    fo.t_stride = mp;
    fo.hour_pstride = 1;
    fo.sw_dir = fo.lw_net = fo.sun = nullptr;
    if (skyview) {
      fo.sw_dir = b + F_SWDIR * fs;
      fo.lw_net = b + F_LWNET * fs;
      fo.sun = (double *)d_sun.p + (size_t)(t0 - 1) * RS_SUN_COLS;
      /* the time axis is shared (checked by the Fortran caller): hour as a shared axis */
      fo.hour = (int32_t *)d_i32pt.p + (size_t)m * len; /* point 0's row of the [p][t] copy */
      fo.hour_pstride = 0;
    }
    RsOutputs oo;
    double *ob = (double *)d_out_tp.p;
    oo.tsurf = ob; oo.snow = ob + fs; oo.water = ob + 2 * fs; oo.ice = ob + 3 * fs;
    oo.deposit = ob + 4 * fs; oo.ice2 = ob + 5 * fs;
    oo.t_stride = mp;
    oo.decimate = 1;
    oo.row0 = t0 - 1;
    if (wb_dev) { /* pre-filled with the caller's values: a point that fails keeps them from there on */
      double *w = (double *)d_wb_tp.p;
      HOK(hipMemcpyAsync(w, fo.sw, fs * sizeof(double), hipMemcpyDeviceToDevice, stream));
      HOK(hipMemcpyAsync(w + fs, fo.sw_dir, fs * sizeof(double), hipMemcpyDeviceToDevice, stream));
      HOK(hipMemcpyAsync(w + 2 * fs, fo.lw, fs * sizeof(double), hipMemcpyDeviceToDevice, stream));
      if (rs_hip_set_writeback(plan, w, w + fs, w + 2 * fs, mp) != 0) return -16;
    }
    /* (a cached plan: the switch is set for every tile, and switching it on zeroes the record) */
    if (t0 == 1 && rs_hip_set_diagnostics(plan, diagnostics ? 1 : 0) != 0) return -12;
    if (t0 == 1 && rs_hip_init_state(plan, &fo, &pp) != 0) return -12;
    if (rs_hip_step(plan, &fo, &oo, &pp, t0, len) != 0) return -13;
    for (int f = 0; f < 6; ++f)
      HOK(transpose((const double *)d_out_tp.p + (size_t)f * fs,
                    (double *)d_out_pt.p + (size_t)f * m * len, len, m, mp, len, stream));
    if (wb_dev) {
      for (int f = 0; f < 3; ++f)
        HOK(transpose((const double *)d_wb_tp.p + (size_t)f * fs,
                      (double *)d_wb_pt.p + (size_t)f * m * len, len, m, mp, len, stream));
      HOK(hipMemcpyAsync(hwb_b[buf], d_wb_pt.p, (size_t)3 * m * len * sizeof(double),
                         hipMemcpyDeviceToHost, stream));
    }
    HOK(hipMemcpyAsync(hout_b[buf], d_out_pt.p, (size_t)6 * m * len * sizeof(double),
                       hipMemcpyDeviceToHost, stream));
    HOK(hipEventRecord(done[buf], stream));
    return 0;
  };

  const int N = (int)items.size();
  gather(items[0], 0);
  for (int k = 0; k < N && rc == 0; ++k) {
    const int buf = k & 1;
    rc = enqueue(items[k], buf);
    if (rc != 0) break;
    if (k >= 1) { /* item k-1 used the other staging set */
      HOK(hipEventSynchronize(done[buf ^ 1]));
      scatter(items[k - 1], buf ^ 1);
      if (retired) {
        rc = finish_tile(retired, retired_p0, retired_m);
        rs_hip_plan_destroy(retired);
        retired = nullptr;
        if (rc != 0) break;
      }
    }
    if (items[k].last) { /* the next item starts a new tile with a new plan */
      retired = plan;
      retired_p0 = items[k].p0;
      retired_m = items[k].m;
      plan = nullptr;
    }
    if (k + 1 < N) gather(items[k + 1], buf ^ 1);
  }
  if (rc == 0) {
    HOK(hipEventSynchronize(done[(N - 1) & 1]));
    scatter(items[N - 1], (N - 1) & 1);
    if (retired) rc = finish_tile(retired, retired_p0, retired_m);
  }
  return rc;
}
