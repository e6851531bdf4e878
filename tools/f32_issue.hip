// Whole-chip fp32 vector issue rates on MI355X with HIP events: what a SIMD issues per microsecond of
// v_fma_f32, v_pk_fma_f32 (two points per lane), v_rcp_f32, v_cndmask/v_cmp pairs and a mix with scalar
// instructions, at 1/2/4/8 wavefronts per SIMD and 1/4 independent chains per wavefront.  Decides how the
// fp32 flavour (rs_kernels_f32.hip) is organised: one point per lane or two (packed fp32).
//   hipcc --offload-arch=gfx950 -O3 tools/f32_issue.hip -o tools/bin/f32_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

enum { K_FMA32 = 0, K_PKFMA32 = 1, K_RCP32 = 2, K_FMA64 = 3, K_SELECT32 = 4, K_FMA32_SALU = 5, K_MIX32 = 6 };

template <int KIND, int CHAINS>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b) {
  float acc = 0.f;
  if (KIND == K_FMA32 || KIND == K_FMA32_SALU) {
    float x[CHAINS];
    for (int c = 0; c < CHAINS; ++c) x[c] = a + c + threadIdx.x;
    int sacc = iters;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fmaf(x[c], b, a);
        if (KIND == K_FMA32_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
      }
    }
    for (int c = 0; c < CHAINS; ++c) acc += x[c];
    acc += (float)sacc;
  } else if (KIND == K_PKFMA32) {
    float2v x[CHAINS];
    const float2v bb = {b, b}, aa = {a, a};
    for (int c = 0; c < CHAINS; ++c) x[c] = float2v{a + c + threadIdx.x, a - c};
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_elementwise_fma(x[c], bb, aa);
    for (int c = 0; c < CHAINS; ++c) acc += x[c].x + x[c].y;
  } else if (KIND == K_RCP32) {
    float x[CHAINS];
    for (int c = 0; c < CHAINS; ++c) x[c] = a + c + threadIdx.x;
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_amdgcn_rcpf(x[c]);
    for (int c = 0; c < CHAINS; ++c) acc += x[c];
  } else if (KIND == K_FMA64) {
    double x[CHAINS];
    for (int c = 0; c < CHAINS; ++c) x[c] = a + c + threadIdx.x;
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fma(x[c], (double)b, (double)a);
    for (int c = 0; c < CHAINS; ++c) acc += (float)x[c];
  } else if (KIND == K_SELECT32) { /* compare + select: two instructions per unit */
    float x[CHAINS];
    for (int c = 0; c < CHAINS; ++c) x[c] = a + c + threadIdx.x;
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
          float y;
          asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %3, vcc" : "=v"(y) : "v"(x[c]), "v"(b), "v"(a) : "vcc");
          x[c] = y;
        }
    for (int c = 0; c < CHAINS; ++c) acc += x[c];
  } else if (KIND == K_MIX32) { /* the layer loop's mix: 8 fma, 1 rcp, 1 cmp+select per unit */
    float x[CHAINS];
    for (int c = 0; c < CHAINS; ++c) x[c] = a + c + threadIdx.x;
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
          float t = x[c];
#pragma unroll
          for (int q = 0; q < 8; ++q) t = __builtin_fmaf(t, b, a);
          t = __builtin_amdgcn_rcpf(t);
          x[c] = t > b ? t : a;
        }
    for (int c = 0; c < CHAINS; ++c) acc += x[c];
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int KIND, int CHAINS>
void run(const char *name, int waves_per_simd, int iters, double units_per_iter) {
  const int wgs = 256 * waves_per_simd; /* 256 threads = 4 waves: one per SIMD of a CU */
  float *out;
  (void)hipMalloc(&out, (size_t)wgs * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, CHAINS>), dim3(wgs), dim3(256), 0, 0, out, iters, 0.5f, 0.999f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, CHAINS>), dim3(wgs), dim3(256), 0, 0, out, iters, 0.5f, 0.999f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double winstr = (double)waves_per_simd * iters * units_per_iter * CHAINS; /* wave-instructions per SIMD */
  printf("%-28s %d waves/SIMD %d chain(s): %8.2f ms  %7.1f wave-instr per SIMD per us  (%.2f cycles each at 2.4 GHz)\n", name,
         waves_per_simd, CHAINS, ms, winstr / (ms * 1e3), 2400.0 * ms * 1e3 / winstr);
  (void)hipFree(out);
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run<K_FMA32, 1>("v_fma_f32", w, 20000, 16);
    run<K_FMA32, 4>("v_fma_f32", w, 5000, 16);
    run<K_PKFMA32, 1>("v_pk_fma_f32", w, 20000, 16);
    run<K_PKFMA32, 4>("v_pk_fma_f32", w, 5000, 16);
    run<K_RCP32, 1>("v_rcp_f32", w, 10000, 16);
    run<K_RCP32, 4>("v_rcp_f32", w, 2500, 16);
    run<K_FMA64, 1>("v_fma_f64", w, 10000, 16);
    run<K_FMA64, 4>("v_fma_f64", w, 2500, 16);
    run<K_SELECT32, 4>("v_cmp+v_cndmask (2 instr)", w, 2500, 32);
    run<K_FMA32_SALU, 1>("v_fma_f32 + s_add (2 instr)", w, 10000, 32);
    run<K_FMA32_SALU, 4>("4 v_fma_f32 + s_add", w, 5000, 16 + 4);
    run<K_MIX32, 1>("8 fma + rcp + cmp/sel", w, 20000, 2 * 11);
    run<K_MIX32, 4>("8 fma + rcp + cmp/sel", w, 5000, 2 * 11);
  }
  return 0;
}
