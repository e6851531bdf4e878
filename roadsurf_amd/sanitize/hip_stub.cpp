// A device layer that never computes: the HIP runtime entry points the product's HOST code calls, on host memory.
//
// `make tsan` / `make asan` (roadsurf_amd/Makefile; the reference has the same switches for its own host code,
// /root/reference/Makefile:38-48) compile the library's translation units host-only (no device code, no GPU),
// with the sanitizer, and link them against this file instead of libamdhip64: "device" allocations are calloc,
// copies are memcpy, a kernel launch does nothing, a stream is a counter.  What runs is every line of host code
// the product has - the coalescer of concurrent runsimulation callers (rs_coalesce.hip), the per-thread caches and
// arenas (rs_host.hip, rs_devutil.hpp), plan bookkeeping (rs_api.hip), the driver path's shards, segment scan and
// worker threads (rs_driver.hip), the Fortran entry points - against a device whose memory stays zero.  Results
// mean nothing; races, use-after-free and out-of-bounds accesses of the host side do
// (tests/test_host_sanitizers.py).  TEST INFRASTRUCTURE: nothing here is linked into libroadsurf_hip.so.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace {
std::atomic<long> g_launches{0};
struct StubStream { std::atomic<long> ops{0}; };
struct StubEvent { std::atomic<long> stamp{0}; };
/* a little latency where a real device would make the host wait, so that callers overlap as they do on a GPU */
void device_latency() {
  static const int us = [] { const char *e = getenv("RS_STUB_LATENCY_US"); return e ? atoi(e) : 200; }();
  if (us > 0) std::this_thread::sleep_for(std::chrono::microseconds(us));
}
}  // namespace

extern "C" {
long rs_stub_kernel_launches(void) { return g_launches.load(); }

hipError_t hipGetDeviceCount(int *count) { *count = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipPeekAtLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "stub device layer"; }
const char *hipGetErrorName(hipError_t) { return "hipStub"; }

hipError_t hipMalloc(void **p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned int) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = (size_t)64 << 30; *t = (size_t)64 << 30; return hipSuccess; }

hipError_t hipStreamCreate(hipStream_t *s) { *s = reinterpret_cast<hipStream_t>(new StubStream()); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned int) { return hipStreamCreate(s); }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned int, int) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) { delete reinterpret_cast<StubStream *>(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { device_latency(); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned int) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
int hipGetStreamDeviceId(hipStream_t) { return 0; }

hipError_t hipEventCreate(hipEvent_t *e) { *e = reinterpret_cast<hipEvent_t>(new StubEvent()); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned int) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<StubEvent *>(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { if (e) reinterpret_cast<StubEvent *>(e)->stamp++; return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { device_latency(); return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.1f; return hipSuccess; }

hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind) { if (n) memmove(dst, src, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(dst, src, n); return hipSuccess; }
hipError_t hipMemset(void *dst, int v, size_t n) { if (n) memset(dst, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t) { if (n) memset(dst, v, n); return hipSuccess; }
/* __device__ / __constant__ variables: in a host-only build the symbol is the host shadow of the variable */
hipError_t hipMemcpyToSymbolAsync(const void *sym, const void *src, size_t n, size_t off, hipMemcpyKind, hipStream_t) {
  if (n) memmove((char *)const_cast<void *>(sym) + off, src, n);
  return hipSuccess;
}
hipError_t hipMemcpyFromSymbolAsync(void *dst, const void *sym, size_t n, size_t off, hipMemcpyKind, hipStream_t) {
  if (n) memmove(dst, (const char *)sym + off, n);
  return hipSuccess;
}
hipError_t hipMemcpyToSymbol(const void *sym, const void *src, size_t n, size_t off, hipMemcpyKind k) {
  return hipMemcpyToSymbolAsync(sym, src, n, off, k, nullptr);
}
hipError_t hipMemcpyFromSymbol(void *dst, const void *sym, size_t n, size_t off, hipMemcpyKind k) {
  return hipMemcpyFromSymbolAsync(dst, sym, n, off, k, nullptr);
}

/* kernel launches: counted, never run */
hipError_t hipLaunchKernel(const void *, dim3, dim3, void **, size_t, hipStream_t) { g_launches++; return hipSuccess; }
hipError_t __hipPushCallConfiguration(dim3, dim3, size_t, hipStream_t) { return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3 *g, dim3 *b, size_t *sh, hipStream_t *s) {
  *g = dim3(1); *b = dim3(1); *sh = 0; *s = nullptr;
  return hipSuccess;
}
void **__hipRegisterFatBinary(const void *) { static void *h = nullptr; return &h; }
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned int, void *, void *, void *, void *, int *) {}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
void __hipRegisterManagedVar(void *, void **, void *, const char *, size_t, unsigned) {}

/* what rocPRIM / hipCUB's host side asks the runtime before a launch (coupling's stream compaction) */
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 64; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600 *p, int) {
  memset(p, 0, sizeof(*p));
  p->multiProcessorCount = 256;
  p->warpSize = 64;
  p->maxThreadsPerBlock = 1024;
  p->sharedMemPerBlock = 64 << 10;
  strcpy(p->gcnArchName, "gfx950");
  return hipSuccess;
}
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, const void *, int, size_t) { *n = 4; return hipSuccess; }
hipError_t hipFuncGetAttributes(hipFuncAttributes *a, const void *) { memset(a, 0, sizeof(*a)); a->maxThreadsPerBlock = 1024; return hipSuccess; }
}
