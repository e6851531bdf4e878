// How accurate is v_rcp_f64 / v_rsq_f64 on this GPU?  Max |1 - x*rcp(x)| over a dense sweep of mantissas
// (the residual is exact in the fma), and the same after one Newton step.  The bare division sequence of
// rs_math.hpp takes two steps, as the compiler's own expansion does; this says what a single one would leave.
// build: hipcc --offload-arch=gfx950 -O2 tools/rcp_accuracy.hip -o /tmp/rcp_accuracy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

__global__ void sweep(uint64_t n_per_thread, double *out) {
  const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  const uint64_t nthreads = gridDim.x * (uint64_t)blockDim.x;
  double m0 = 0, m1 = 0, m2 = 0, s0 = 0;
  uint64_t state = tid * 0x9E3779B97F4A7C15ull + 12345;
  for (uint64_t i = 0; i < n_per_thread; ++i) {
    // half of the samples on a regular grid of mantissas, half random
    uint64_t mant;
    if (i & 1) {
      state ^= state << 13; state ^= state >> 7; state ^= state << 17;
      mant = state & ((1ull << 52) - 1);
    } else {
      mant = ((i >> 1) * nthreads + tid) * 0x1000003ull & ((1ull << 52) - 1);
    }
    const double x = __longlong_as_double((long long)((1023ull << 52) | mant));
    const double r0 = __builtin_amdgcn_rcp(x);
    const double e0 = __builtin_fma(-x, r0, 1.0);
    const double r1 = __builtin_fma(r0, e0, r0);
    const double e1 = __builtin_fma(-x, r1, 1.0);
    const double r2 = __builtin_fma(r1, e1, r1);
    const double e2 = __builtin_fma(-x, r2, 1.0);
    m0 = fmax(m0, fabs(e0));
    m1 = fmax(m1, fabs(e1));
    m2 = fmax(m2, fabs(e2));
    const double y = __builtin_amdgcn_rsq(x);
    s0 = fmax(s0, fabs(__builtin_fma(-x * y, y, 1.0)));
  }
  out[4 * tid] = m0; out[4 * tid + 1] = m1; out[4 * tid + 2] = m2; out[4 * tid + 3] = s0;
}

int main() {
  const int blocks = 1024, threads = 256;
  const uint64_t per = 1 << 14;  // 4.3e9 samples
  double *d;
  if (hipMalloc(&d, (size_t)blocks * threads * 4 * 8) != hipSuccess) return 1;
  hipLaunchKernelGGL(sweep, dim3(blocks), dim3(threads), 0, 0, per, d);
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  double *h = new double[(size_t)blocks * threads * 4];
  (void)hipMemcpy(h, d, (size_t)blocks * threads * 4 * 8, hipMemcpyDeviceToHost);
  double m[4] = {0, 0, 0, 0};
  for (size_t i = 0; i < (size_t)blocks * threads; ++i)
    for (int k = 0; k < 4; ++k) m[k] = fmax(m[k], h[4 * i + k]);
  printf("samples %.3g in [1,2)\n", (double)blocks * threads * per);
  printf("v_rcp_f64        max |1 - x r| = %.3e = 2^%.2f\n", m[0], log2(m[0]));
  printf("one Newton step  max |1 - x r| = %.3e = 2^%.2f\n", m[1], log2(m[1]));
  printf("two Newton steps max |1 - x r| = %.3e = 2^%.2f\n", m[2], log2(m[2]));
  printf("v_rsq_f64        max |1 - x y^2| = %.3e = 2^%.2f\n", m[3], log2(m[3]));
  return 0;
}
