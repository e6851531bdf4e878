/*
 * rs_coalesce.hip - the literal drop-in, made usable: `runsimulation` called once per point from the
 * reference driver's worker threads (examples/example1/src/roadrunner.cpp:454-497: `-j` threads, each
 * takes a point off the queue and calls runsimulation for it).
 *
 * One such call steps ONE lane of one wavefront through the whole series - the latency of 5 761
 * dependent time steps, tens of milliseconds, whatever else the GPU could be doing - and a process gets
 * four hardware queues, so 64 callers do not even run 64 such kernels at once (INTEGRATION.md section 1
 * has the measured table).  With ROADSURF_HIP_COALESCE_US = w > 0 concurrent callers are gathered
 * instead: a caller that finds nobody collecting becomes the collector, waits up to w microseconds (or
 * until ROADSURF_HIP_COALESCE_MAX callers, default 4096, are queued) and runs everything queued with
 * ITS settings and parameters (compared byte for byte) as one runsimulation_batch; the others sleep
 * until their point is done.  Same kernels, same bits: a batch is bit-identical to its points run alone
 * (tests/test_hip_boundary.py).  Callers with other settings are left in the queue and one of them
 * collects next; a batch starts when the one before it has finished (everybody who arrived meanwhile
 * is in it).
 */
#include <hip/hip_runtime.h>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/roadsurf.h"

extern "C" void runsimulation_batch(int32_t n, OutputPointers *out, const InputPointers *in,
                                    const InputSettings *settings, const InputParameters *params,
                                    const LocalParameters *local, int32_t *status);

namespace {

struct Request {
  OutputPointers *out;
  const InputPointers *in;
  const InputSettings *settings;
  const InputParameters *params;
  const LocalParameters *local;
  int32_t status = 0;
  bool taken = false, done = false;
};

std::mutex g_m;
std::condition_variable g_cv;
std::vector<Request *> g_queue;
bool g_collecting = false;
int g_inflight = 0; /* batches running: the next one is held back until the GPU is free again, so that
                       callers arriving meanwhile join it instead of trickling in as batches of a few */
long g_batches = 0, g_points = 0; /* diagnostics: rs_coalesce_stats */

int window_us() {
  static const int w = [] {
    const char *e = getenv("ROADSURF_HIP_COALESCE_US");
    return e ? atoi(e) : 0;
  }();
  return w;
}
size_t max_batch() {
  static const size_t m = [] {
    const char *e = getenv("ROADSURF_HIP_COALESCE_MAX");
    const long v = e ? atol(e) : 4096;
    return (size_t)(v >= 1 ? v : 1);
  }();
  return m;
}

}  // namespace

extern "C" {

/* batches run and points served through the coalescer so far (tests) */
void rs_coalesce_stats(int64_t *batches, int64_t *points) {
  std::lock_guard<std::mutex> lk(g_m);
  if (batches) *batches = g_batches;
  if (points) *points = g_points;
}

/* What the Fortran `runsimulation` calls: one point, coalesced with the other threads' points when
 * ROADSURF_HIP_COALESCE_US > 0, else a batch of one.  Returns the batch status (0 = ok). */
int32_t rs_coalesce_run(OutputPointers *out, const InputPointers *in, const InputSettings *settings,
                        const InputParameters *params, const LocalParameters *local) {
  const int w = window_us();
  if (w <= 0) {
    int32_t st = 0;
    runsimulation_batch(1, out, in, settings, params, local, &st);
    return st;
  }
  Request me;
  me.out = out; me.in = in; me.settings = settings; me.params = params; me.local = local;
  std::unique_lock<std::mutex> lk(g_m);
  g_queue.push_back(&me);
  g_cv.notify_all(); /* a collector waiting for its batch to fill looks again */
  for (;;) {
    if (me.done) return me.status;
    if (!me.taken && !g_collecting) {
      /* collect: wait for the window to close or the batch to fill */
      g_collecting = true;
      const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(w);
      g_cv.wait_until(lk, deadline, [&] { return g_queue.size() >= max_batch(); });
      /* a batch steps its points' whole series in a few wavefronts: a second batch beside it would be
       * as slow and hold fewer points.  Wait for the running one (callers keep arriving). */
      if (g_inflight > 0 && g_queue.size() < max_batch()) {
        g_cv.wait(lk, [&] { return g_inflight == 0 || g_queue.size() >= max_batch(); });
        /* the callers that batch has just released are on their way back with their next points:
         * one more window for them, so that the batches do not settle into two alternating halves */
        const auto again = std::chrono::steady_clock::now() + std::chrono::microseconds(w);
        g_cv.wait_until(lk, again, [&] { return g_queue.size() >= max_batch(); });
      }
      /* the collector's own point first (a queue longer than the maximum must not leave it
       * out of the batch it runs), then the others with its settings and parameters, in queue order */
      std::vector<Request *> batch, rest;
      batch.push_back(&me);
      for (Request *r : g_queue) {
        if (r == &me) continue;
        const bool same = batch.size() < max_batch() &&
                          std::memcmp(r->settings, me.settings, sizeof(InputSettings)) == 0 &&
                          std::memcmp(r->params, me.params, sizeof(InputParameters)) == 0;
        (same ? batch : rest).push_back(r);
      }
      g_queue.swap(rest);
      for (Request *r : batch) r->taken = true;
      g_collecting = false;
      g_inflight += 1;
      g_batches += 1;
      g_points += (long)batch.size();
      g_cv.notify_all(); /* whoever is left may collect the next batch while this one runs */
      lk.unlock();
      const int32_t n = (int32_t)batch.size();
      std::vector<OutputPointers> o(n);
      std::vector<InputPointers> i(n);
      std::vector<LocalParameters> l(n);
      for (int32_t k = 0; k < n; ++k) {
        o[k] = *batch[k]->out;
        i[k] = *batch[k]->in;
        l[k] = *batch[k]->local;
      }
      int32_t st = 0;
      runsimulation_batch(n, o.data(), i.data(), me.settings, me.params, l.data(), &st);
      /* a batch-level error is some ONE caller's (an array shorter than SimLen, ...): the header promises
       * every caller the bits and the status of a call of its own, so the members run again one by one */
      std::vector<int32_t> each((size_t)n, st);
      if (st != 0 && n > 1)
        for (int32_t k = 0; k < n; ++k) {
          each[k] = 0;
          runsimulation_batch(1, &o[k], &i[k], me.settings, me.params, &l[k], &each[k]);
        }
      lk.lock();
      g_inflight -= 1;
      for (int32_t k = 0; k < n; ++k) {
        batch[k]->status = each[k];
        batch[k]->done = true;
      }
      g_cv.notify_all();
      return me.status;
    }
    g_cv.wait(lk);
  }
}

} /* extern "C" */
