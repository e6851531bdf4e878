// Drives the product's host code from many threads, as the reference driver's worker pool does (/root/reference/
// examples/example1/src/WorkQueue.h:16-129, roadrunner.cpp:454-497), in the sanitizer builds (`make tsan`, `make
// asan`): the library is compiled host-only against sanitize/hip_stub.cpp - a device that never computes - so
// results mean nothing and every race / invalid access of the host side is reported.
//   phase 1  T threads x N points through `runsimulation`, two groups of settings (SimLen differs), the coalescer
//            limited to ROADSURF_HIP_COALESCE_MAX callers per batch (set by the test: 5 < threads)
//   phase 2  threads that come and go (thread-local caches adopted by later threads, rs_host.hip CallerCache)
//   phase 3  four concurrent runsimulation_batch calls of different sizes (arena / plan bookkeeping)
//   phase 4  rs_driver_run from two threads (its shards, segment table and per-block worker threads)
// usage: harness [threads=64] [points=640]      exit code 0 and "sanitize harness ok" when every call returned
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "roadsurf.h"

namespace {

struct PointData {
  std::vector<double> tair, tdew, vz, rhz, prec, sw, lw, sw_dir, lw_net, obs, depth, hz;
  std::vector<int32_t> phase, year, month, day, hour, minute, second;
  std::vector<double> out[6];
  explicit PointData(int L)
      : tair(L, -3.0), tdew(L, -5.0), vz(L, 2.0), rhz(L, 85.0), prec(L, 0.0), sw(L, 50.0), lw(L, 270.0), sw_dir(L, 30.0),
        lw_net(L, -40.0), obs(L, -9999.9), depth(L, -9999.9), hz(360, 0.0), phase(L, -9999), year(L, 2024), month(L, 1),
        day(L, 10), hour(L), minute(L), second(L) {
    for (auto &o : out) o.assign(L, -9999.0);
    for (int t = 0; t < L; ++t) {
      const int sec = t * 30;
      hour[t] = (sec / 3600) % 24;
      minute[t] = (sec / 60) % 60;
      second[t] = sec % 60;
    }
    obs[0] = -3.5;
  }
  void pointers(InputPointers &ip, OutputPointers &op, int L) {
    std::memset(&ip, 0, sizeof(ip));
    ip.inputLen = L;
    ip.c_tair = tair.data(); ip.c_tdew = tdew.data(); ip.c_VZ = vz.data(); ip.c_Rhz = rhz.data();
    ip.c_prec = prec.data(); ip.c_SW = sw.data(); ip.c_LW = lw.data(); ip.c_SW_dir = sw_dir.data();
    ip.c_LW_net = lw_net.data(); ip.c_TSurfObs = obs.data(); ip.c_PrecPhase = phase.data();
    ip.c_local_horizons = hz.data(); ip.c_Depth = depth.data();
    ip.c_year = year.data(); ip.c_month = month.data(); ip.c_day = day.data();
    ip.c_hour = hour.data(); ip.c_minute = minute.data(); ip.c_second = second.data();
    op.outputLen = L;
    op.c_TsurfOut = out[0].data(); op.c_SnowOut = out[1].data(); op.c_WaterOut = out[2].data();
    op.c_IceOut = out[3].data(); op.c_DepositOut = out[4].data(); op.c_Ice2Out = out[5].data();
  }
};

struct Group {
  int L;
  InputSettings s;
  InputParameters p;
  LocalParameters l;
  explicit Group(int L_) : L(L_) {
    rs_default_settings(&s, L);
    rs_default_parameters(&p, s.DTSecs);
    rs_default_local(&l);
    l.InitLenI = 1;
  }
};

std::atomic<int> g_errors{0};

void one_point_calls(const Group &g, std::atomic<int> &next, int N) {
  PointData d(g.L);
  for (;;) {
    const int pt = next.fetch_add(1);
    if (pt >= N) break;
    d.tair[1] = -3.0 + 0.001 * pt;
    InputPointers ip;
    OutputPointers op;
    d.pointers(ip, op, g.L);
    LocalParameters l = g.l;
    runsimulation(&op, &ip, &g.s, &g.p, &l);
    if (d.out[0][g.L - 1] < -9000.0) { /* runsimulation's own failure mark */
      fprintf(stderr, "runsimulation failed for point %d: %s\n", pt, rs_last_error());
      g_errors++;
    }
  }
}

void batch_call(const Group &g, int n) {
  std::vector<PointData> pts;
  pts.reserve(n);
  for (int k = 0; k < n; ++k) pts.emplace_back(g.L);
  std::vector<InputPointers> ip(n);
  std::vector<OutputPointers> op(n);
  std::vector<LocalParameters> l(n, g.l);
  for (int k = 0; k < n; ++k) pts[k].pointers(ip[k], op[k], g.L);
  int32_t st = 99;
  std::vector<int32_t> ff(n, -1);
  runsimulation_batch_ex(n, op.data(), ip.data(), &g.s, &g.p, l.data(), &st, ff.data());
  if (st != 0) {
    fprintf(stderr, "runsimulation_batch_ex(%d) -> %d: %s\n", n, st, rs_last_error());
    g_errors++;
  }
}

/* rs_driver_run: an hourly forecast on a shared axis + 10-minute observations on per-point axes (the shape of
 * examples/example1's two JSON files), `n` points, relaxation (and coupling) on: the host side cuts shards over
 * ROADSURF_HIP_DEVICES, builds the segment table of the raw times and runs one worker thread per block */
void driver_call(int n, int hours, bool coupling) {
  const int L = hours * 120 + 1, nf = hours + 1, no = 6 * 3 + 1; /* observations over the first three hours */
  const int64_t t0 = 1704844800; /* 2024-01-10 00:00 UTC */
  std::vector<int64_t> tf(nf), to((size_t)n * no);
  std::vector<int32_t> olen(n);
  for (int k = 0; k < nf; ++k) tf[k] = t0 + 3600 * k;
  auto series = [&](size_t w, double base) {
    std::vector<double> v((size_t)n * w);
    for (size_t q = 0; q < v.size(); ++q) v[q] = base + 0.01 * (double)(q % 37);
    return v;
  };
  std::vector<double> f_tair = series(nf, -4.0), f_rh = series(nf, 80.0), f_vz = series(nf, 2.0), f_prec = series(nf, 0.0),
                      f_sw = series(nf, 40.0), f_lw = series(nf, 260.0), f_swd = series(nf, 20.0), f_lwn = series(nf, -40.0);
  std::vector<double> o_tair = series(no, -3.5), o_rh = series(no, 82.0), o_vz = series(no, 1.5), o_ts = series(no, -2.0);
  for (int p = 0; p < n; ++p) {
    olen[p] = no - (p % 3); /* ragged: some stations stop reporting early */
    for (int k = 0; k < no; ++k) to[(size_t)p * no + k] = t0 + 600 * k;
  }
  RsRawSource src[2];
  std::memset(src, 0, sizeof(src));
  src[0].n_times = nf; src[0].times = tf.data();
  src[0].tair = f_tair.data(); src[0].rhz = f_rh.data(); src[0].vz = f_vz.data(); src[0].prec = f_prec.data();
  src[0].sw = f_sw.data(); src[0].lw = f_lw.data(); src[0].sw_dir = f_swd.data(); src[0].lw_net = f_lwn.data();
  src[1].n_times = no; src[1].is_observation = 1; src[1].times = to.data(); src[1].times_per_point = 1;
  src[1].lengths = olen.data();
  src[1].tair = o_tair.data(); src[1].rhz = o_rh.data(); src[1].vz = o_vz.data(); src[1].tsurfobs = o_ts.data();
  std::vector<int32_t> yy(L, 2024), mo(L, 1), dd(L, 10), hh(L), mi(L), ss(L);
  for (int t = 0; t < L; ++t) { hh[t] = (t / 120) % 24; mi[t] = (t / 2) % 60; ss[t] = (t % 2) * 30; }
  RsDriverInput in;
  std::memset(&in, 0, sizeof(in));
  in.n_points = n; in.n_sources = 2; in.sources = src; in.start_time = t0; in.forecast_time = t0 + 3 * 3600;
  in.year = yy.data(); in.month = mo.data(); in.day = dd.data(); in.hour = hh.data(); in.minute = mi.data(); in.second = ss.data();
  InputSettings s;
  InputParameters p;
  rs_default_settings(&s, L);
  rs_default_parameters(&p, s.DTSecs);
  s.use_relaxation = 1;
  s.use_coupling = coupling ? 1 : 0;
  std::vector<LocalParameters> local(n);
  for (auto &l : local) rs_default_local(&l);
  const int step = (int)((double)(s.outputStep * 60) / s.DTSecs), n_out = (L + step - 1) / step;
  std::vector<double> o[6];
  for (auto &v : o) v.assign((size_t)n * n_out, -9999.0);
  std::vector<int32_t> status(n, -1), missing(n, -2);
  RsDriverOutput out;
  out.n_out = n_out;
  out.tsurf = o[0].data(); out.snow = o[1].data(); out.water = o[2].data(); out.ice = o[3].data(); out.deposit = o[4].data();
  out.ice2 = o[5].data(); out.status = status.data(); out.missing_index = missing.data();
  const int rc = rs_driver_run(&in, &s, &p, local.data(), &out, -1);
  if (rc != 0) {
    fprintf(stderr, "rs_driver_run(%d points) -> %d: %s\n", n, rc, rs_last_error());
    g_errors++;
  }
}

}  // namespace

int main(int argc, char **argv) {
  const int T = argc > 1 ? std::max(2, atoi(argv[1])) : 64;
  const int N = argc > 2 ? std::max(T, atoi(argv[2])) : 640;
  const Group ga(241), gb(361); /* two hours / three hours: the coalescer keeps the groups apart */
  {
    std::atomic<int> na{0}, nb{0};
    std::vector<std::thread> th;
    for (int k = 0; k < T; ++k) th.emplace_back([&, k] { one_point_calls((k & 1) ? gb : ga, (k & 1) ? nb : na, N / 2); });
    for (auto &x : th) x.join();
  }
  int64_t batches = 0, points = 0;
  rs_coalesce_stats(&batches, &points);
  printf("phase 1: %d threads, %d points: %lld coalesced batches, %lld points in them\n", T, N, (long long)batches, (long long)points);
  for (int round = 0; round < 3; ++round) { /* threads that end: their caches are adopted by the next ones */
    std::atomic<int> na{0};
    std::vector<std::thread> th;
    for (int k = 0; k < 8; ++k) th.emplace_back([&] { one_point_calls(ga, na, 24); });
    for (auto &x : th) x.join();
  }
  printf("phase 2: short-lived caller threads done\n");
  {
    std::vector<std::thread> th;
    const int sizes[4] = {1, 7, 300, 1025};
    for (int k = 0; k < 4; ++k) th.emplace_back([&, k] { batch_call((k & 1) ? gb : ga, sizes[k]); });
    for (auto &x : th) x.join();
  }
  printf("phase 3: concurrent runsimulation_batch_ex calls done\n");
  {
    std::vector<std::thread> th;
    th.emplace_back([] { driver_call(9000, 6, false); });
    th.emplace_back([] { driver_call(700, 5, true); });
    for (auto &x : th) x.join();
  }
  printf("phase 4: concurrent rs_driver_run calls done\n");
  if (g_errors.load() != 0) {
    printf("sanitize harness: %d calls failed\n", g_errors.load());
    return 1;
  }
  printf("sanitize harness ok\n");
  return 0;
}
