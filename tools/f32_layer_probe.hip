// What two points per lane buy the fp32 flavour's layer loop on MI355X (design probe for rs_kernels_f32.hip):
// the fused CalcHCapHCond + calcCapDZCondDZ + calcProfile update of 15 layers (src/BalanceModel.f90:90-251)
// per time step, profile in registers, in three organisations:
//   1  one point per lane, float                      (round 2-5's organisation)
//   2  two points per lane, two floats side by side    (the compiler may pair them)
//   3  two points per lane, ext_vector float2          (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32)
// Reported: nanoseconds of SIMD time per 64 point-steps (15 layers), at 2/4/8 wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast tools/f32_layer_probe.hip -o tools/bin/f32_layer_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

struct LayerK { float dryCap, WCont, DyC, condDZ; };
struct Args {
  LayerK lk[15];
  float hcw[8];
  float dt, tbot;
  int iters;
  float *out;
};

/* ---- organisation 1: float ---- */
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float rcp_(float a) { return __builtin_amdgcn_rcpf(a); }
__device__ __forceinline__ float sel_ge0(float t, float a, float b) { return t >= 0.f ? a : b; }
__device__ __forceinline__ float splat(float, float x) { return x; }

/* ---- organisation 3: packed ---- */
__device__ __forceinline__ f2 fma_(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 rcp_(f2 a) { return f2{__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)}; }
__device__ __forceinline__ f2 sel_ge0(f2 t, f2 a, f2 b) { return f2{t.x >= 0.f ? a.x : b.x, t.y >= 0.f ? a.y : b.y}; }
__device__ __forceinline__ f2 splat(f2, float x) { return f2{x, x}; }

/* ---- organisation 2: two floats ---- */
struct p2 { float a, b; };
__device__ __forceinline__ p2 operator+(p2 x, p2 y) { return p2{x.a + y.a, x.b + y.b}; }
__device__ __forceinline__ p2 operator-(p2 x, p2 y) { return p2{x.a - y.a, x.b - y.b}; }
__device__ __forceinline__ p2 operator*(p2 x, p2 y) { return p2{x.a * y.a, x.b * y.b}; }
__device__ __forceinline__ p2 operator-(p2 x) { return p2{-x.a, -x.b}; }
__device__ __forceinline__ p2 fma_(p2 a, p2 b, p2 c) { return p2{__builtin_fmaf(a.a, b.a, c.a), __builtin_fmaf(a.b, b.b, c.b)}; }
__device__ __forceinline__ p2 rcp_(p2 a) { return p2{__builtin_amdgcn_rcpf(a.a), __builtin_amdgcn_rcpf(a.b)}; }
__device__ __forceinline__ p2 sel_ge0(p2 t, p2 a, p2 b) { return p2{t.a >= 0.f ? a.a : b.a, t.b >= 0.f ? a.b : b.b}; }
__device__ __forceinline__ p2 splat(p2, float x) { return p2{x, x}; }

template <class V>
__device__ __forceinline__ V layer(const Args &a, int j, V tj, V tnext, V &Gprev) {
  const V z = tj;
#define K(x) splat(z, x)
  const V t2 = tj * tj;
  const V roow = fma_(K(a.hcw[0]), t2, fma_(K(a.hcw[1]), tj, K(a.hcw[2])));
  const V cw = fma_(K(a.hcw[3]), t2 * t2, fma_(K(-a.hcw[4]), t2 * tj, fma_(K(a.hcw[5]), t2, fma_(K(-a.hcw[6]), tj, K(a.hcw[7])))));
  const V chwt = sel_ge0(tj, roow * cw, K(920.0f * 2100.0f));
  const V vsh = fma_(K(a.lk[j].WCont), chwt, K(a.lk[j].dryCap));
  const V capDZ = -rcp_(K(a.lk[j].DyC) * vsh);
  const V G = K(a.lk[j].condDZ) * (tnext - tj);
  const V tn = fma_(K(a.dt), capDZ * (G - Gprev), tj);
  Gprev = G;
  return tn;
#undef K
}

template <class V, int WPS>
__global__ void __launch_bounds__(256, WPS) k(const Args a) {
  V T[15];
  const float seed = (float)(threadIdx.x & 63) * 0.01f;
  for (int j = 0; j < 15; ++j) T[j] = splat(T[0], -3.0f + 0.4f * j + seed);
  V acc = splat(T[0], 0.f);
  for (int it = 0; it < a.iters; ++it) {
    V Gprev = splat(T[0], 30.0f + (float)(it & 7));
#pragma unroll
    for (int j = 0; j < 15; ++j) {
      const V tnext = (j == 14) ? splat(T[0], a.tbot) : T[j + 1];
      T[j] = layer<V>(a, j, T[j], tnext, Gprev);
    }
    acc = acc + T[0] * T[1];
  }
  V s = acc;
  for (int j = 0; j < 15; ++j) s = s + T[j];
  float r;
  if constexpr (sizeof(V) == 4) r = *(float *)&s; else r = ((float *)&s)[0] + ((float *)&s)[1];
  a.out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <class V, int WPS>
void run(const char *name, Args a, int points_per_lane) {
  const int wgs = 256 * WPS;
  (void)hipMalloc(&a.out, (size_t)wgs * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<V, WPS>), dim3(wgs), dim3(256), 0, 0, a);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<V, WPS>), dim3(wgs), dim3(256), 0, 0, a);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  /* per SIMD: WPS waves x iters steps x points_per_lane x 64 points */
  const double wave_steps = (double)WPS * a.iters * points_per_lane;
  printf("%-34s %d waves/SIMD: %7.2f ms  %7.1f ns of SIMD time per 64 point-steps (15 layers) = %5.1f cycles per layer at 2.4 GHz\n",
         name, WPS, ms, ms * 1e6 / wave_steps, ms * 1e6 / wave_steps * 2.4 / 15);
  (void)hipFree(a.out);
}

int main() {
  Args a;
  for (int j = 0; j < 15; ++j) a.lk[j] = LayerK{1.6e6f + 1e4f * j, 0.1f + 0.01f * j, 0.03f * (1 + j), -1.0f / (0.03f * (1 + j))};
  const float h[8] = {-0.0050f, 0.0079f, 1000.0028f, 0.0000102f, 0.0017169f, 0.11516f, 3.4739f, 4217.2f};
  for (int q = 0; q < 8; ++q) a.hcw[q] = h[q];
  a.dt = 30.f;
  a.tbot = 4.f;
  a.iters = 20000;
  run<float, 2>("1 point/lane float", a, 1);
  run<float, 4>("1 point/lane float", a, 1);
  run<float, 8>("1 point/lane float", a, 1);
  run<p2, 2>("2 points/lane, two floats", a, 2);
  run<p2, 4>("2 points/lane, two floats", a, 2);
  run<p2, 8>("2 points/lane, two floats", a, 2);
  run<f2, 2>("2 points/lane, packed float2", a, 2);
  run<f2, 4>("2 points/lane, packed float2", a, 2);
  run<f2, 8>("2 points/lane, packed float2", a, 2);
  return 0;
}
