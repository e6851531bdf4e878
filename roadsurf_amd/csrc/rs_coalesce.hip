/*
 * rs_coalesce.hip - the literal drop-in, made usable: `runsimulation` called once per point from the
 * reference driver's worker threads (examples/example1/src/roadrunner.cpp:454-497: `-j` threads, each
 * takes a point off the queue and calls runsimulation for it).
 *
 * One such call steps ONE lane of one wavefront pair through the whole series - the latency of 5 761
 * dependent time steps, tens of milliseconds, whatever else the GPU could be doing.  So concurrent callers
 * are GATHERED: a caller that finds nobody collecting becomes the collector, waits a window (or until
 * ROADSURF_HIP_COALESCE_MAX callers, default 4096, are queued) and runs everything queued with ITS settings
 * and parameters (compared byte for byte) as one runsimulation_batch; the others sleep until their point is
 * done.  Same kernels, same bits: a batch is bit-identical to its points run alone
 * (tests/test_hip_boundary.py).  Callers with other settings are left in the queue and one of them collects
 * next; a batch starts when the one before it has finished (everybody who arrived meanwhile is in it).
 *
 * ROADSURF_HIP_COALESCE_US: unset (round 5: the default) - AUTOMATIC: a caller that finds no other call of
 * this process inside the library runs its point at once, as a batch of one (a single-threaded caller never
 * waits); one that finds another call in flight coalesces, with a window of 1/16 of the last batch's duration
 * (0.2 ... 3 ms).  w > 0: every caller coalesces with a window of w microseconds.  0: never (every call a
 * batch of one, on the calling thread's own stream: rs_host.hip CallerCache).
 */
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/roadsurf.h"

/* runsimulation_batch for the points of `runsimulation` callers (RoadSurfHip.f90): as that entry is the
 * reference's own, the reference's in-place edits of the input arrays are written back by default
 * (src/InputOutput.f90:75-77, src/ModRadiation.f90:57-71; ROADSURF_HIP_WRITEBACK=0 opts out) */
extern "C" void rs_runsimulation_gathered(int32_t n, OutputPointers *out, const InputPointers *in,
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

int g_active = 0;        /* calls of this process inside rs_coalesce_run (automatic mode) */
long g_last_batch_us = 16000; /* duration of the last batch: the automatic window is 1/16 of it */
size_t g_last_batch_n = 0;    /* callers in the last batch: the next window closes when as many are back */
/* automatic mode: when the last batch of several callers ended.  Its callers are on their way back with their
 * next points - the first of them must wait for the others rather than run alone (16 worker threads would
 * otherwise alternate between a batch of one and a batch of fifteen) */
std::chrono::steady_clock::time_point g_last_multi{};
bool g_seen_multi = false;

/* -1: automatic (unset), 0: off, w > 0: fixed window */
int window_us() {
  static const int w = [] {
    const char *e = getenv("ROADSURF_HIP_COALESCE_US");
    return e ? std::max(0, atoi(e)) : -1;
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

/* Batches of several callers run on ONE thread of the library's own (started with the first such batch): its
 * stream, its cached buffers (rs_host.hip CallerCache) and its OpenMP team for the row copies serve every
 * batch.  Run on whichever caller happened to collect, sixty-four worker threads each raised a team of their
 * own, whose idle members spin on the cores the next collector's team needs: 64 callers got 630 points/s where
 * they get 900 with the rows copied by a single thread. */
struct Runner {
  std::mutex m;
  std::condition_variable cv;
  bool started = false, has_job = false, job_done = false;
  int32_t n = 0;
  OutputPointers *o = nullptr;
  const InputPointers *i = nullptr;
  const InputSettings *s = nullptr;
  const InputParameters *p = nullptr;
  const LocalParameters *l = nullptr;
  int32_t status = 0;
  void loop() {
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv.wait(lk, [&] { return has_job; });
      lk.unlock();
      int32_t st = 0;
      rs_runsimulation_gathered(n, o, i, s, p, l, &st);
      lk.lock();
      status = st;
      has_job = false;
      job_done = true;
      cv.notify_all();
    }
  }
  /* one job at a time (callers queue on the mutex's condition) */
  int32_t run(int32_t n_, OutputPointers *o_, const InputPointers *i_, const InputSettings *s_,
              const InputParameters *p_, const LocalParameters *l_) {
    std::unique_lock<std::mutex> lk(m);
    if (!started) {
      started = true;
      std::thread([this] { loop(); }).detach();
    }
    cv.wait(lk, [&] { return !has_job && !job_done; });
    n = n_; o = o_; i = i_; s = s_; p = p_; l = l_;
    has_job = true;
    cv.notify_all();
    cv.wait(lk, [&] { return job_done; });
    const int32_t st = status;
    job_done = false;
    cv.notify_all();
    return st;
  }
};
Runner &runner() {
  static Runner *r = new Runner(); /* never destroyed: its thread outlives main() */
  return *r;
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
  int w = window_us();
  if (w == 0) {
    int32_t st = 0;
    rs_runsimulation_gathered(1, out, in, settings, params, local, &st);
    return st;
  }
  Request me;
  me.out = out; me.in = in; me.settings = settings; me.params = params; me.local = local;
  std::unique_lock<std::mutex> lk(g_m);
  struct Active { /* this call is inside the library (counted under g_m) */
    Active() { ++g_active; }
    ~Active() { --g_active; }
  };
  if (w < 0) {
    const bool gathering = g_seen_multi &&
        std::chrono::steady_clock::now() - g_last_multi < std::chrono::microseconds(std::max<long>(50000, 4 * g_last_batch_us));
    if (g_active == 0 && g_queue.empty() && !g_collecting && !gathering) {
      /* nobody else is here: the point runs at once.  It counts as a batch in flight, so that callers
       * arriving meanwhile gather behind it instead of starting batches of one beside it. */
      Active a;
      g_inflight += 1;
      lk.unlock();
      const auto t0 = std::chrono::steady_clock::now();
      int32_t st = 0;
      rs_runsimulation_gathered(1, out, in, settings, params, local, &st);
      const long us = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
      lk.lock();
      g_last_batch_us = us;
      g_inflight -= 1;
      g_cv.notify_all();
      return st;
    }
    w = (int)std::min<long>(3000, std::max<long>(200, g_last_batch_us / 16));
  }
  Active active; /* (destroyed before lk: under the lock) */
  g_queue.push_back(&me);
  g_cv.notify_all(); /* a collector waiting for its batch to fill looks again */
  for (;;) {
    if (me.done) return me.status;
    if (!me.taken && !g_collecting) {
      /* collect: wait for the window to close or the batch to fill */
      g_collecting = true;
      /* ... or until as many callers as the last batch had are back (a worker pool returns as one) */
      auto full = [&] { return g_queue.size() >= max_batch() || (g_last_batch_n > 1 && g_queue.size() >= g_last_batch_n); };
      const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(w);
      g_cv.wait_until(lk, deadline, full);
      /* a batch steps its points' whole series in a few wavefronts: a second batch beside it would be
       * as slow and hold fewer points.  Wait for the running one (callers keep arriving). */
      if (g_inflight > 0 && g_queue.size() < max_batch()) {
        g_cv.wait(lk, [&] { return g_inflight == 0 || g_queue.size() >= max_batch(); });
        /* the callers that batch has just released are on their way back with their next points:
         * one more window for them, so that the batches do not settle into two alternating halves */
        const auto again = std::chrono::steady_clock::now() + std::chrono::microseconds(w);
        g_cv.wait_until(lk, again, full);
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
      const auto tb0 = std::chrono::steady_clock::now();
      if (n > 1) st = runner().run(n, o.data(), i.data(), me.settings, me.params, l.data());
      else rs_runsimulation_gathered(n, o.data(), i.data(), me.settings, me.params, l.data(), &st);
      const long batch_us = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tb0).count();
      /* a batch-level error is some ONE caller's (an array shorter than SimLen, ...): the header promises
       * every caller the bits and the status of a call of its own, so the members run again one by one */
      std::vector<int32_t> each((size_t)n, st);
      if (st != 0 && n > 1)
        for (int32_t k = 0; k < n; ++k) {
          each[k] = 0;
          rs_runsimulation_gathered(1, &o[k], &i[k], me.settings, me.params, &l[k], &each[k]);
        }
      lk.lock();
      g_inflight -= 1;
      g_last_batch_us = batch_us;
      g_last_batch_n = (size_t)n;
      if (n > 1) {
        g_seen_multi = true;
        g_last_multi = std::chrono::steady_clock::now();
      }
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
