/* rs_devices.hpp — multi-GPU fan-out of the host-array entry points (internal).
 *
 * Points are independent, so a batch is cut into contiguous blocks of points,
 * one block per HIP device, and every block runs the single-device path on its own host
 * thread with its own stream and plan(s).  No collective, nothing crosses xGMI.  This is the
 * in-process counterpart of the reference driver's worker pool
 * (examples/example1/src/roadrunner.cpp:423-501, WorkQueue.h:16-129): there a worker owns a
 * point at a time, here a worker owns a device and a block of points.
 *
 *   ROADSURF_HIP_DEVICES   comma-separated device indices ("0,1,2,3"; a device may be listed more
 *                          than once: that many concurrent plans on it), or "all" (default)
 *   ROADSURF_HIP_DEVICE    one device (kept from round 1; ROADSURF_HIP_DEVICES wins)
 *   LOCAL_RANK             with neither of the two set: a rank-local process (one process per GPU)
 *                          uses device LOCAL_RANK modulo the visible devices, not the whole node
 *   ROADSURF_HIP_PLANS_PER_DEVICE  when the list is not given explicitly, every device appears this
 *                          many times (default 4): four blocks per device run on four host threads
 *                          and four streams, so that one block's PCIe copies and host-side
 *                          gather/scatter overlap the other's kernels (measured on one MI355X,
 *                          rs_driver_run, 1 M points x 48 h: relaxation 5.5e9 -> 9.6e9, coupling 3.8e9 -> 6.3e9
 *                          point-timesteps/s)
 *   ROADSURF_HIP_MIN_SHARD a block is at least this many points (default 4096): small batches
 *                          use fewer devices, one-point calls pick a device round-robin
 */
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <cstdio>
#include <vector>

extern "C" const char *rs_last_error(void);
extern "C" void rs_host_set_error(const char *msg);

namespace rsu {

inline std::vector<int> device_list() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return {};
  std::vector<int> out;
  const char *e = getenv("ROADSURF_HIP_DEVICES");
  if (e && *e && strcmp(e, "all") != 0) {
    const char *p = e;
    while (*p) {
      char *end = nullptr;
      const long d = strtol(p, &end, 10);
      if (end == p) break;
      if (d >= 0 && d < ndev) out.push_back((int)d);
      if (*end != ',') break;
      p = end + 1;
    }
    if (!out.empty()) return out;
  }
  int lo = 0, hi = ndev;
  if (!(e && *e)) {
    if (const char *lr = getenv("LOCAL_RANK")) {
      /* one process per GPU (torch.distributed.run and friends): a rank that sees the whole node
       * keeps to its own device instead of spreading over - and oversubscribing - all of them.
       * Only where there ARE several local ranks: a lone process that a launcher happened to start
       * with LOCAL_RANK=0 (Slurm, torchrun --nproc-per-node 1, a notebook kernel) keeps the fan-out */
      const char *lws = getenv("LOCAL_WORLD_SIZE");
      const char *ws = lws ? lws : getenv("WORLD_SIZE");
      const int d = atoi(lr);
      if (d >= 0 && ws && atoi(ws) > 1) {
        lo = d % ndev;
        hi = lo + 1;
        static std::atomic<bool> said{false};
        if (ndev > 1 && !said.exchange(true))
          fprintf(stderr, "roadsurf_hip: LOCAL_RANK=%d of %s local ranks: this process keeps to device %d "
                          "(ROADSURF_HIP_DEVICES overrides)\n", d, ws, lo);
      }
    }
  }
  int per = 4;
  if (const char *k = getenv("ROADSURF_HIP_PLANS_PER_DEVICE"))
    if (atoi(k) >= 1 && atoi(k) <= 8) per = atoi(k);
  for (int r = 0; r < per; ++r) /* device-major: a batch too small for all entries still uses every device */
    for (int d = lo; d < hi; ++d) out.push_back(d);
  return out;
}

struct Shard {
  int device;
  int64_t off, cnt;
};
#ifndef RS_BLOCK_TAPER_PCT_DEFAULT
#define RS_BLOCK_TAPER_PCT_DEFAULT 0
#endif

/* contiguous block partition of n points over the listed devices, remainders to the first
 * blocks (roadsurf_amd/sharding.py strong_shard is the same rule) */
inline std::vector<Shard> make_shards(int64_t n, const std::vector<int> &devs, int default_taper = RS_BLOCK_TAPER_PCT_DEFAULT) {
  std::vector<Shard> s;
  if (n < 1 || devs.empty()) return s;
  int64_t min_shard = 4096;
  if (const char *e = getenv("ROADSURF_HIP_MIN_SHARD"))
    if (atoll(e) > 0) min_shard = atoll(e);
  int64_t k = (int64_t)devs.size();
  if (n / min_shard < k) k = n / min_shard;
  if (k < 1) k = 1;
  if (k == 1) {
    /* a batch too small to split: successive calls take the devices in turn, so a caller that
     * runs one point per worker thread (the reference driver) still spreads over the node */
    static std::atomic<unsigned> turn{0};
    const int d = devs[turn.fetch_add(1u) % devs.size()];
    s.push_back(Shard{d, 0, n});
    return s;
  }
  int64_t off = 0;
  /* Blocks that share ONE device take turns on the link, so they start staggered by an upload each and - of
   * equal size - end staggered too, the last ones alone on the GPU.  With a taper of t % the blocks shrink
   * linearly, the last one to (100 - t) % of the first (0: equal blocks; what the callers pass was measured level
   * with equal blocks, profiles/r05_ab_block_taper.txt - the environment knobs for it and for an undersized first
   * block are gone since round 6). */
  bool one_device = true;
  for (int64_t i = 1; i < k; ++i) one_device = one_device && devs[(size_t)i] == devs[0];
  const int taper = default_taper;
  if (one_device && k > 1 && taper > 0 && taper < 90 && n / k >= 4 * min_shard) {
    double wsum = 0.0;
    std::vector<double> w((size_t)k);
    for (int64_t i = 0; i < k; ++i) wsum += (w[(size_t)i] = 1.0 - (taper / 100.0) * (double)i / (double)(k - 1));
    for (int64_t i = 0; i < k; ++i) {
      int64_t cnt = (i + 1 == k) ? n - off : (int64_t)((double)n * w[(size_t)i] / wsum) / 256 * 256;
      if (cnt < min_shard) cnt = min_shard;
      if (cnt > n - off) cnt = n - off;
      s.push_back(Shard{devs[(size_t)i], off, cnt});
      off += cnt;
    }
    return s;
  }
  const int64_t base = n / k, rem = n % k;
  for (int64_t i = 0; i < k; ++i) {
    const int64_t cnt = base + (i < rem ? 1 : 0);
    s.push_back(Shard{devs[(size_t)i], off, cnt});
    off += cnt;
  }
  return s;
}

/* One block at a time uploads to a device.  The blocks of a fan-out start together; left alone
 * they all upload together - at 1/K of the link each - and then all compute together, so the GPU
 * idles through the whole upload and the link through the whole computation.  Taking turns, the
 * first block computes while the second uploads, and the phases stay staggered down to the
 * downloads (measured, rs_driver_run, 1 M points x 48 h: see DESIGN.md 6). */
inline std::mutex &copy_gate(int device) {
  static std::mutex gates[64];
  return gates[device & 63];
}

/* how many blocks the calling thread's last fan-out used (rs_last_fanout(), tests) */
inline thread_local int g_last_fanout = 0;

/* f(shard, nshards) -> rc on one host thread per shard; returns the first non-zero rc and
 * re-raises that worker's rs_last_error() text in the calling thread */
template <class F>
inline int fan_out(const std::vector<Shard> &shards, F f) {
  if (shards.empty()) {
    rs_host_set_error("no HIP device visible - this library has no CPU path");
    return -9;
  }
  g_last_fanout = (int)shards.size();
  if (shards.size() == 1) return f(shards[0], 1);
  std::vector<int> rc(shards.size(), 0);
  std::vector<std::string> msg(shards.size());
  std::vector<std::thread> th;
  for (size_t i = 0; i < shards.size(); ++i)
    th.emplace_back([&, i] {
      rc[i] = f(shards[i], (int)shards.size());
      if (rc[i] != 0) msg[i] = rs_last_error();
    });
  for (auto &t : th) t.join();
  for (size_t i = 0; i < shards.size(); ++i)
    if (rc[i] != 0) {
      const std::string m = "device " + std::to_string(shards[i].device) + ", points " +
                            std::to_string(shards[i].off) + "..: " + msg[i];
      rs_host_set_error(m.c_str());
      return rc[i];
    }
  return 0;
}

}  // namespace rsu
