// Where do the wavefronts of a small launch land?  Every wavefront records the XCD, shader engine,
// CU and SIMD it runs on (s_getreg HW_ID / XCC_ID), then spins so that the whole grid is resident at
// once, with the register and LDS footprint of the step kernels (128 VGPRs, 4.5 KB LDS per workgroup).
// usage: placement_probe <workgroups> <threads per workgroup> [vgprs=128] [streams=1]
// build: hipcc --offload-arch=gfx950 -O2 tools/placement_probe.hip -o /tmp/placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

template <int VGPRS>
__global__ void probe(uint32_t *out, uint32_t spin_ticks) {
  __shared__ double pad[576];
  pad[threadIdx.x % 576] = 0.0;
  if (VGPRS >= 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  if (VGPRS >= 168) asm volatile("v_mov_b32 v167, 0" ::: "v167");
  if (VGPRS >= 256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
  const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
  const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);  // HW_REG_XCC_ID
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
  if ((threadIdx.x & 63) == 0) {
    const uint32_t w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    out[2 * w] = hw;
    out[2 * w + 1] = xcc;
  }
}

int main(int argc, char **argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 489, threads = argc > 2 ? atoi(argv[2]) : 256;
  const int vg = argc > 3 ? atoi(argv[3]) : 128, nstreams = argc > 4 ? atoi(argv[4]) : 1;
  const int waves = wgs * (threads / 64);
  uint32_t *d;
  hipMalloc(&d, (size_t)waves * 2 * 4 * nstreams);
  hipMemset(d, 0xff, (size_t)waves * 2 * 4 * nstreams);
  std::vector<hipStream_t> st(nstreams);
  for (auto &s : st) hipStreamCreate(&s);
  for (int k = 0; k < nstreams; ++k) {
    uint32_t *o = d + (size_t)k * waves * 2;
    if (vg >= 256) hipLaunchKernelGGL(probe<256>, dim3(wgs), dim3(threads), 0, st[k], o, 30000u);
    else if (vg >= 168) hipLaunchKernelGGL(probe<168>, dim3(wgs), dim3(threads), 0, st[k], o, 30000u);
    else if (vg >= 128) hipLaunchKernelGGL(probe<128>, dim3(wgs), dim3(threads), 0, st[k], o, 30000u);
    else hipLaunchKernelGGL(probe<0>, dim3(wgs), dim3(threads), 0, st[k], o, 30000u);
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
  std::vector<uint32_t> h((size_t)waves * 2 * nstreams);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  std::map<uint32_t, int> per_simd, per_cu;
  for (int w = 0; w < waves * nstreams; ++w) {
    const uint32_t hw = h[2 * w], xcc = h[2 * w + 1] & 15;
    const uint32_t simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const uint32_t cukey = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    per_cu[cukey]++;
    per_simd[(cukey << 2) | simd]++;
  }
  std::map<int, int> hist_simd, hist_cu;
  for (auto &kv : per_simd) hist_simd[kv.second]++;
  for (auto &kv : per_cu) hist_cu[kv.second]++;
  printf("wgs %d x %d threads, %d VGPRs, %d stream(s): %d waves on %zu CUs / %zu SIMDs\n", wgs, threads, vg,
         nstreams, waves * nstreams, per_cu.size(), per_simd.size());
  printf("  waves per CU  :");
  for (auto &kv : hist_cu) printf("  %d waves: %d CUs;", kv.first, kv.second);
  printf("\n  waves per SIMD:");
  for (auto &kv : hist_simd) printf("  %d waves: %d SIMDs;", kv.first, kv.second);
  printf("\n");
  return 0;
}
