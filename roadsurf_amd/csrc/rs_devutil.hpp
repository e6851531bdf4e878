/* rs_devutil.hpp — small device/host helpers shared by the host-array layers (internal). */
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace rsu {

constexpr int TS = 32; /* transpose tile */

/* src[r][c] (rows x cols, leading dim ld_src) -> dst[c][r] (leading dim ld_dst) */
template <typename T>
__global__ void __launch_bounds__(TS * 8) transpose_kernel(const T *__restrict__ src,
                                                           T *__restrict__ dst, int rows, int cols,
                                                           int64_t ld_src, int64_t ld_dst) {
  __builtin_amdgcn_s_setprio(3); /* on the way in or out of a block: waited for, beside other blocks' step kernels */
  __shared__ T tile[TS][TS + 1];
  const int c0 = blockIdx.x * TS, r0 = blockIdx.y * TS;
  for (int j = threadIdx.y; j < TS; j += 8) {
    const int r = r0 + j, c = c0 + threadIdx.x;
    if (r < rows && c < cols) tile[j][threadIdx.x] = src[(int64_t)r * ld_src + c];
  }
  __syncthreads();
  for (int j = threadIdx.y; j < TS; j += 8) {
    const int c = c0 + j, r = r0 + threadIdx.x;
    if (r < rows && c < cols) dst[(int64_t)c * ld_dst + r] = tile[threadIdx.x][j];
  }
}

template <typename T>
inline hipError_t transpose(const T *src, T *dst, int rows, int cols, int64_t ld_src,
                            int64_t ld_dst, hipStream_t s) {
  if (rows < 1 || cols < 1) return hipSuccess;
  dim3 g((cols + TS - 1) / TS, (rows + TS - 1) / TS), b(TS, 8);
  hipLaunchKernelGGL(transpose_kernel<T>, g, b, 0, s, src, dst, rows, cols, ld_src, ld_dst);
  return hipGetLastError();
}

/* A bump allocator over one device block that a worker keeps across calls (rs_driver.hip): while a
 * thread has one installed (tls_arena), Dev::alloc carves its buffers out of it instead of calling
 * hipMalloc - two dozen allocations and, worse, as many hipFree per tile, each of which waits for the
 * whole device (every other worker's kernels included).  A request that does not fit falls back to
 * hipMalloc, so nothing depends on the size estimate.  The owner rewinds the arena when the buffers of
 * a tile are dead (after a stream synchronisation). */
struct Arena {
  char *base = nullptr;
  size_t cap = 0, off = 0;
  uint64_t epoch = 0; /* bumped by every rewind: what was carved out before it is dead */
  void rewind(size_t to) {
    off = to;
    ++epoch;
  }
  void *take(size_t n) {
    const size_t a = (off + 255) & ~(size_t)255;
    if (!base || a + n > cap) return nullptr;
    off = a + n;
    return base + a;
  }
};
inline Arena *&tls_arena() {
  static thread_local Arena *a = nullptr;
  return a;
}

struct Dev {
  void *p = nullptr;
  bool owned = true; /* false: a piece of the thread's arena */
  Dev() = default;
  Dev(const Dev &) = delete;
  Dev &operator=(const Dev &) = delete;
  ~Dev() { release(); }
  void release() {
    if (p && owned) (void)hipFree(p);
    p = nullptr;
    owned = true;
  }
  hipError_t alloc(size_t n) {
    release();
    if (Arena *a = tls_arena())
      if (void *q = a->take(n ? n : 8)) {
        p = q;
        owned = false;
        return hipSuccess;
      }
    return hipMalloc(&p, n ? n : 8);
  }
  template <typename T>
  T *as() const { return static_cast<T *>(p); }
};
/* the same for page-locked host memory (rs_host.hip: a caller thread that comes back with small batches -
 * the reference driver calling runsimulation point by point - keeps one block instead of nine
 * hipHostMalloc / hipHostFree per call, each of which takes milliseconds and serialises the callers) */
inline Arena *&tls_pinned_arena() {
  static thread_local Arena *a = nullptr;
  return a;
}
struct Pinned {
  void *p = nullptr;
  bool owned = true;
  Pinned() = default;
  Pinned(const Pinned &) = delete;
  Pinned &operator=(const Pinned &) = delete;
  ~Pinned() {
    if (p && owned) (void)hipHostFree(p);
  }
  hipError_t alloc(size_t n) {
    if (Arena *a = tls_pinned_arena())
      if (void *q = a->take(n ? n : 8)) {
        p = q;
        owned = false;
        return hipSuccess;
      }
    return hipHostMalloc(&p, n ? n : 8, hipHostMallocDefault);
  }
};

/* Host worker threads for row gather/scatter: the CPUs this process may actually use
 * (OpenMP's view AND the cgroup v2 cpu.max quota), not the machine's core count: a container
 * with a 16-CPU quota on a 256-thread host runs 10x slower with 256 OpenMP threads. */
inline int host_threads(int omp_procs) {
  const char *e = getenv("ROADSURF_HIP_HOST_THREADS");
  if (e && atoi(e) > 0) return atoi(e);
  int n = omp_procs;
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    long long period = 0;
    char q[32];
    if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
      const long long quota = atoll(q);
      const int c = (int)((quota + period - 1) / period);
      if (c > 0 && c < n) n = c;
    }
    fclose(f);
  }
  return n < 1 ? 1 : n;
}

}  // namespace rsu
