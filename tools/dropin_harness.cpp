// The literal drop-in under a C++ caller shaped like the reference driver's worker pool
// (/root/reference/examples/example1/src/WorkQueue.h:16-129, roadrunner.cpp:454-497: `-j T` threads, each
// takes the next point off a queue, fills InputData/OutputData for it and calls `runsimulation`).
//
// Links against libroadsurf_hip.so only (include/roadsurf.h); makes its own forcing (a deterministic
// synthetic series per point: smooth diurnal cycles, some precipitation) - throughput, not parity, is what
// this tool measures (tests/test_hip_boundary.py holds the bits).
//
//   build: g++ -O2 -std=c++17 -pthread tools/dropin_harness.cpp -Iinclude -Lroadsurf_amd/lib -lroadsurf_hip
//              -Wl,-rpath,$PWD/roadsurf_amd/lib -o tools/bin/dropin_harness
//   usage: dropin_harness <threads> <points> [hours=48]
//   prints one JSON line: points/s, mean / max milliseconds per call
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "roadsurf.h"

namespace {

struct PointData {  // what InputData / OutputData hold for one point (InputData.cpp:5-26, OutputData.cpp:5-13)
  std::vector<double> tair, tdew, vz, rhz, prec, sw, lw, sw_dir, lw_net, obs, depth, hz;
  std::vector<int32_t> phase, year, month, day, hour, minute, second;
  std::vector<double> out[6];
  explicit PointData(int L)
      : tair(L), tdew(L), vz(L), rhz(L), prec(L), sw(L), lw(L), sw_dir(L), lw_net(L, -40.0), obs(L, -9999.9),
        depth(L, -9999.9), hz(360, 0.0), phase(L, -9999), year(L, 2024), month(L, 1), day(L), hour(L), minute(L),
        second(L) {
    for (auto &o : out) o.assign(L, -9999.0);
    for (int t = 0; t < L; ++t) {
      const int sec = t * 30;
      day[t] = 10 + sec / 86400;
      hour[t] = (sec / 3600) % 24;
      minute[t] = (sec / 60) % 60;
      second[t] = sec % 60;
    }
  }
  void fill(int point, int L) {  // a new point's forcing into the same buffers
    const double base = -12.0 + 0.017 * (point % 997), amp = 2.0 + 0.004 * (point % 499), ph = 0.001 * (point % 6283);
    for (int t = 0; t < L; ++t) {
      const double h = t / 120.0, d = 2.0 * M_PI * h / 24.0;
      tair[t] = base + amp * std::sin(d + ph);
      tdew[t] = tair[t] - 1.5;
      vz[t] = 0.4 + 3.0 * std::fabs(std::sin(2.0 * d + 0.5 * ph)) + 0.002 * (point % 61);
      rhz[t] = 85.0 + 8.0 * std::cos(d);
      prec[t] = ((point + (int)h) % 11 == 0) ? 0.6 : 0.0;
      sw[t] = std::fmax(0.0, 180.0 * std::sin(d - M_PI / 2.0));
      sw_dir[t] = 0.6 * sw[t];
      lw[t] = 270.0 + 20.0 * std::sin(d + 1.0);
    }
    obs[0] = tair[0] - 0.5;
    for (auto &o : out) std::fill(o.begin(), o.end(), -9999.0);
  }
};

}  // namespace

int main(int argc, char **argv) {
  const int T = argc > 1 ? std::max(1, atoi(argv[1])) : 16;
  const int N = argc > 2 ? std::max(1, atoi(argv[2])) : 1024;
  const int hours = argc > 3 ? std::max(1, atoi(argv[3])) : 48;
  const int L = hours * 120 + 1;
  InputSettings s;
  InputParameters p;
  LocalParameters l0;
  rs_default_settings(&s, L);
  rs_default_parameters(&p, s.DTSecs);
  rs_default_local(&l0);
  l0.InitLenI = 1;

  std::atomic<int> next{0};
  std::mutex m;
  double lat_sum = 0.0, lat_max = 0.0;
  double checksum = 0.0;
  auto worker = [&]() {
    PointData d(L);
    double ls = 0.0, lm = 0.0, cs = 0.0;
    for (;;) {
      const int pt = next.fetch_add(1);
      if (pt >= N) break;
      d.fill(pt, L);
      InputPointers ip;
      std::memset(&ip, 0, sizeof(ip));
      ip.inputLen = L;
      ip.c_tair = d.tair.data(); ip.c_tdew = d.tdew.data(); ip.c_VZ = d.vz.data(); ip.c_Rhz = d.rhz.data();
      ip.c_prec = d.prec.data(); ip.c_SW = d.sw.data(); ip.c_LW = d.lw.data(); ip.c_SW_dir = d.sw_dir.data();
      ip.c_LW_net = d.lw_net.data(); ip.c_TSurfObs = d.obs.data(); ip.c_PrecPhase = d.phase.data();
      ip.c_local_horizons = d.hz.data(); ip.c_Depth = d.depth.data();
      ip.c_year = d.year.data(); ip.c_month = d.month.data(); ip.c_day = d.day.data();
      ip.c_hour = d.hour.data(); ip.c_minute = d.minute.data(); ip.c_second = d.second.data();
      OutputPointers op;
      op.outputLen = L;
      op.c_TsurfOut = d.out[0].data(); op.c_SnowOut = d.out[1].data(); op.c_WaterOut = d.out[2].data();
      op.c_IceOut = d.out[3].data(); op.c_DepositOut = d.out[4].data(); op.c_Ice2Out = d.out[5].data();
      LocalParameters l = l0;
      const auto t0 = std::chrono::steady_clock::now();
      runsimulation(&op, &ip, &s, &p, &l);
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      ls += ms;
      lm = std::max(lm, ms);
      cs += d.out[0][L - 1];
      if (d.out[0][L - 1] < -9000.0) {
        fprintf(stderr, "point %d: no output at the last index (%s)\n", pt, rs_last_error());
        exit(2);
      }
    }
    std::lock_guard<std::mutex> lk(m);
    lat_sum += ls;
    lat_max = std::max(lat_max, lm);
    checksum += cs;
  };
  {  // first call: library, device and (per thread, later) context warm-up are not the steady state
    PointData d(L);
    (void)d;
  }
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int k = 0; k < T; ++k) th.emplace_back(worker);
  for (auto &x : th) x.join();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  int64_t batches = 0, pts = 0;
  rs_coalesce_stats(&batches, &pts);
  printf("{\"threads\": %d, \"points\": %d, \"simlen\": %d, \"seconds\": %.3f, \"points_per_s\": %.1f, "
         "\"ms_per_call_mean\": %.2f, \"ms_per_call_max\": %.2f, \"coalesced_batches\": %lld, \"coalesced_points\": %lld, "
         "\"checksum\": %.6f}\n",
         T, N, L, dt, N / dt, lat_sum / N, lat_max, (long long)batches, (long long)pts, checksum);
  return 0;
}
