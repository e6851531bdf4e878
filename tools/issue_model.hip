// Issue model of one gfx950 SIMD for the instruction mix of the road model's time step: what W resident
// wavefronts, each an in-order stream of mostly DEPENDENT fp64 instructions with scalar work, scalar
// loads and branches in between, get out of their SIMD.  (the ceiling next to the
// small-shard number.)  Every instruction is a volatile asm, so the streams run as written.
//
//   mode 0  dependent v_add_f64 chain                       (the pure latency case)
//   mode 1  two independent chains interleaved in one wave  (ILP 2)
//   mode 2  chain + one independent s_add_u32 per 3 vector instructions  (the kernel's 1 374 : 408)
//   mode 3  chain + v_cmp / s_and_b64 pairs (exec-mask bookkeeping: scalar work that DEPENDS on the chain)
//   mode 4  mode 2 + one s_load_dwordx2 with an immediate wait per 20 vector instructions (69 per step)
//   mode 5  mode 4 + one v_rcp_f64 per 32 vector instructions (44 per step) + a taken branch per 20
//   mode 6  mode 5 with the scalar loads issued 20 instructions ahead of their wait
//   mode 7  four independent chains (ILP 4), no scalar work: what the SIMD can issue at all
// Launch shapes: W workgroups of 256 per CU in ONE launch (W waves per SIMD), and "2q": two launches of
// one workgroup per CU on two streams (two waves per SIMD that belong to different queues, as two plans'
// step kernels do).  Output: instructions a SIMD issues per microsecond (wall time of the launch) and per 4
// cycles at the clock given as argv[1].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

/* One asm statement per loop body: the compiler's hazard recogniser pads every inline-asm statement
 * that writes a VGPR with an s_nop (it cannot see inside), which would be an issued instruction of
 * its own.  Operands: %0 x, %1 y, %2 z, %3 w (chains), %4 s (scalar counter), %5 m (mask), %6 c
 * (loaded constant), %7 c2, %8 b, %9 constant block. */
#define A_(r) "v_add_f64 " r ", " r ", %8\n\t"
#define M_(r) "v_mul_f64 " r ", " r ", %8\n\t"
#define AX A_("%0")
#define MX M_("%0")
#define SA "s_add_u32 %4, %4, 1\n\t"
#define R3 AX MX AX SA          /* three dependent vector instructions and one scalar */
#define R3x6 R3 R3 R3 R3 R3 R3
#define CMP "v_cmp_gt_f64 vcc, %0, %8\n\ts_and_b64 %5, vcc, exec\n\t"
#define BR "s_cmp_lg_u32 %4, 0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:\n\t"
#define BR2 "s_cmp_lg_u32 %4, 0\n\ts_cbranch_scc1 2f\n\ts_nop 0\n2:\n\t"
#define BR3 "s_cmp_lg_u32 %4, 0\n\ts_cbranch_scc1 3f\n\ts_nop 0\n3:\n\t"
#define LDW "s_load_dwordx2 %6, %9, 0x0\n\ts_waitcnt lgkmcnt(0)\n\t"
#define USEC "v_add_f64 %0, %0, %6\n\t"
#define X10(a) a a a a a a a a a a
#define OPS : "+v"(x), "+v"(y), "+v"(z), "+v"(w), "+s"(s), "+s"(m), "=&s"(c), "=&s"(c2) : "v"(b), "s"(cp) : "vcc", "scc", "memory"

template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned long long *out, const double *cblock, int iters, double b) {
  double x = b + threadIdx.x, y = b * 2 + threadIdx.x, z = b * 3, w = b * 4, c = 0, c2 = 0;
  unsigned s = blockIdx.x + 1;
  const double *cp = cblock;
  unsigned long long m = 0;
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) asm volatile(X10(AX AX AX AX AX AX) OPS);
    else if (MODE == 1) asm volatile(X10(A_("%0") A_("%1") A_("%0") A_("%1") A_("%0") A_("%1")) OPS);
    else if (MODE == 7) asm volatile(X10(A_("%0") A_("%1") A_("%2") A_("%3")) A_("%0") A_("%1") A_("%2") A_("%3") A_("%0") A_("%1") A_("%2") A_("%3") A_("%0") A_("%1") A_("%2") A_("%3") A_("%0") A_("%1") A_("%2") A_("%3") A_("%0") A_("%1") A_("%2") A_("%3") OPS);
    else if (MODE == 2) asm volatile(X10(R3 R3) OPS);
    else if (MODE == 3) asm volatile(X10(AX MX CMP AX AX MX CMP AX) OPS);
    else if (MODE == 4) asm volatile(LDW USEC R3x6 AX LDW USEC R3x6 AX LDW USEC R3x6 AX OPS);
    else if (MODE == 5)
      asm volatile(LDW USEC R3x6 AX AX BR LDW USEC R3x6 AX "v_rcp_f64 %0, %0\n\t" BR2 LDW USEC R3x6 AX AX BR3 OPS);
    else if (MODE == 6) /* the load of group g+1 is issued at the top of group g, waited for a group later */
      asm volatile("s_load_dwordx2 %6, %9, 0x0\n\t"
                   "s_load_dwordx2 %7, %9, 0x8\n\ts_waitcnt lgkmcnt(1)\n\t" USEC R3x6 AX AX BR
                   "s_waitcnt lgkmcnt(0)\n\tv_add_f64 %0, %0, %7\n\ts_load_dwordx2 %6, %9, 0x10\n\t" R3x6 AX "v_rcp_f64 %0, %0\n\t" BR2
                   "s_waitcnt lgkmcnt(0)\n\t" USEC R3x6 AX AX BR3 OPS);
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
  if (x + y + z + w + c + c2 == 12345.678 && s == 77 && m == 5) out[0] = 1; /* keep everything alive */
}

static int instr_per_iter(int mode) { /* every issued instruction: s_waitcnt counts, a skipped s_nop does not */
  switch (mode) {
    case 0: case 1: return 60;
    case 7: return 64;
    case 2: return 80;
    case 3: return 100;
    case 4: return 3 * (2 + 1 + 24 + 1);
    case 5: case 6: return 3 * (2 + 1 + 24 + 1 + 1 + 2);
  }
  return 0;
}

template <int MODE>
static void run(int W, bool two_queues, int iters, double clock_hint_mhz) {
  const int cus = 256;
  const int wgs = cus * (two_queues ? 1 : W);
  unsigned long long *out[2];
  double *cb;
  (void)hipMalloc(&cb, 4096);
  (void)hipMemset(cb, 0, 4096);
  hipStream_t st[2];
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int nq = two_queues ? 2 : 1;
  for (int q = 0; q < nq; ++q) {
    (void)hipStreamCreate(&st[q]);
    (void)hipMalloc(&out[q], (size_t)wgs * 4 * 8);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, st[q], out[q], cb, 10, 1.0000001); /* warm */
  }
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, st[0]);
  for (int q = 0; q < nq; ++q)
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, st[q], out[q], cb, iters, 1.0000001);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e1, st[0]);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)wgs * 4);
  double cyc = 0;
  for (int q = 0; q < nq; ++q) {
    (void)hipMemcpy(h.data(), out[q], h.size() * 8, hipMemcpyDeviceToHost);
    for (auto v : h) cyc += (double)v;
  }
  cyc /= (double)(h.size() * nq);
  const double ninst = (double)instr_per_iter(MODE) * iters;
  const int waves_per_simd = two_queues ? 2 : W;
  /* From the wall time of the launch (HIP events): instructions a SIMD issues per microsecond, and per
   * four cycles at the clock given on the command line (default 2 300 MHz, what the chip holds under the
   * road model's load; a pure-FMA launch like mode 7 draws more power and clocks lower).  The wave's own
   * s_memtime delta is printed for reference only: it is not a clean shader-cycle count on this part. */
  const double per_us = ninst * waves_per_simd / (ms * 1e3);
  printf("mode %d  %s  waves/SIMD %d : %.2f ms  -> a SIMD issues %.0f instructions/us = %.2f per 4 cycles at %.0f MHz; "
         "a wave gets one every %.1f cycles   [s_memtime delta per wave %.0f]\n",
         MODE, two_queues ? "2 queues" : "1 launch", waves_per_simd, ms, per_us, per_us * 4.0 / clock_hint_mhz,
         clock_hint_mhz, clock_hint_mhz * (ms * 1e3) / ninst, cyc);
  for (int q = 0; q < nq; ++q) { (void)hipFree(out[q]); (void)hipStreamDestroy(st[q]); }
  (void)hipFree(cb);
}

template <int MODE>
static void sweep(int iters, double mhz) {
  for (int W = 1; W <= 4; ++W) run<MODE>(W, false, iters, mhz);
  run<MODE>(2, true, iters, mhz);
}

int main(int argc, char **argv) {
  const double mhz = argc > 1 ? atof(argv[1]) : 2300.0; /* the clock the chip holds under fp64 load */
  const int iters = 20000;
  sweep<0>(iters, mhz); sweep<1>(iters, mhz); sweep<7>(iters, mhz); sweep<2>(iters, mhz);
  sweep<3>(iters, mhz); sweep<4>(iters, mhz); sweep<5>(iters, mhz); sweep<6>(iters, mhz);
  return 0;
}
