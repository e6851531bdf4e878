// Whole-chip fp64 FMA rate with HIP events (no shader-clock counter involved): what one SIMD issues per cycle.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void __launch_bounds__(256) k(double *out, int iters, double a, double b) {
  double x[CHAINS];
  for (int c = 0; c < CHAINS; ++c) x[c] = a + c + threadIdx.x;
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fma(x[c], b, a);
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
void run(int wgs, int iters) {
  double *out; (void)hipMalloc(&out, (size_t)wgs * 256 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<CHAINS>, dim3(wgs), dim3(256), 0, 0, out, iters, 0.5, 0.999);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<CHAINS>, dim3(wgs), dim3(256), 0, 0, out, iters, 0.5, 0.999);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double fma = (double)wgs * 256 * iters * 16 * CHAINS;
  printf("%d workgroups x 256, %d chain(s): %.2f ms, %.1f TFLOP/s fp64, %.2f wave64-FMA per SIMD per us (1024 SIMDs)\n", wgs, CHAINS, ms,
         2 * fma / ms / 1e9, fma / 64 / 1024 / (ms * 1e3));
  (void)hipFree(out);
}
int main() {
  run<1>(256 * 4, 40000);   // 4 waves per SIMD
  run<4>(256 * 4, 10000);
  run<4>(256 * 2, 10000);   // 2 waves per SIMD
  run<1>(256 * 1, 40000);   // 1 wave per SIMD
  return 0;
}
