/*
 * rs_cluster.hip — plan order (include/roadsurf.h, rs_hip_recluster): sort the slots of a plan
 * by the boundary-layer passes their points needed during the last launch and move the carried
 * state along.  Device-wide radix sort from hipCUB; everything else is streaming copies.
 */
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "rs_kernels.h"
#include "rs_state.h"

namespace {

constexpr int KEY_BITS = RS_SORT_KEY_BITS; /* rs_kernels.hip, bl_score_key: 19 bits of passes + regime + cover */

template <typename T> /* element type of the state block: double, or float for fp32 plans */
__global__ void __launch_bounds__(RS_BLOCK) keys_kernel(const T *__restrict__ state,
                                                        int64_t np_pad, int64_t npoints,
                                                        uint32_t *keys, uint32_t *slots) {
  const int64_t s = (int64_t)blockIdx.x * RS_BLOCK + threadIdx.x;
  if (s >= npoints) return;
  double v = (double)state[(int64_t)RS_ST_BLSCORE * np_pad + s];
  if (!(v >= 0.0)) v = 0.0;
  const double top = (double)((1u << KEY_BITS) - 1u);
  /* descending: the expensive points get the low slots, so their workgroups are dispatched
   * first and the cheap ones fill the end of the grid (longest job first) */
  keys[s] = (uint32_t)top - (uint32_t)(v > top ? top : v);
  slots[s] = (uint32_t)s;
}

/* Which state rows a re-sort has to move (rs_state.h).  blockIdx.y counts the rows in use:
 * the profile Tmp(1..NLayers), the scalars TmpNw(1)..BLSCORE, then - with coupling - the coupling
 * scalars ITER..RESUME, and while coupling windows are still open also the saved state of
 * saveDataForCoupling (6 scalars, TmpSave(1..NLayers)) and the stale TmpNw profile. */
struct RowMap {
  int32_t nlayers, nscal, ncpl, nsave; /* rows of each group; nsave counts the 6 scalars only */
  __host__ __device__ int32_t total() const {
    return nlayers + nscal + ncpl + nsave + (nsave ? 2 * nlayers : 0);
  }
  __device__ int64_t row(int32_t y) const {
    if (y < nlayers) return y;
    y -= nlayers;
    if (y < nscal) return RS_ST_TNW1 + y;
    y -= nscal;
    if (y < ncpl) return RS_ST_CPL_ITER + y;
    y -= ncpl;
    if (y < nsave) return RS_ST_CPL_SAVE_TSURF + y;
    y -= nsave;
    if (y < nlayers) return RS_ST_CPL_SAVE_TMP0 + y;
    return RS_ST_CPL_STALE_TMP0 + (y - nlayers);
  }
};

/* dst[row][s] = src[row][perm[s]] for the carried state, order_dst[s] = order_src[perm[s]];
 * slots beyond npoints (padding) stay where they are */
template <typename T>
__global__ void __launch_bounds__(RS_BLOCK) apply_kernel(const T *__restrict__ src,
                                                         T *__restrict__ dst,
                                                         const int32_t *__restrict__ order_src,
                                                         int32_t *__restrict__ order_dst,
                                                         const uint32_t *__restrict__ perm,
                                                         int64_t np_pad, int64_t npoints,
                                                         const RowMap rows) {
  __builtin_amdgcn_s_setprio(3); /* a link of the chain between two step launches of a plan: little work, all of it waited for */
  const int64_t s = (int64_t)blockIdx.x * RS_BLOCK + threadIdx.x;
  if (s >= np_pad) return;
  const int64_t from = (s < npoints) ? (int64_t)perm[s] : s;
  if (blockIdx.y == 0) order_dst[s] = order_src[from];
  const int64_t row = rows.row((int32_t)blockIdx.y);
  dst[row * np_pad + s] = src[row * np_pad + from];
}

__global__ void __launch_bounds__(RS_BLOCK) iota_kernel(int32_t *x, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * RS_BLOCK + threadIdx.x;
  if (i < n) x[i] = (int32_t)i;
}

inline dim3 grid1(int64_t n) { return dim3((unsigned)((n + RS_BLOCK - 1) / RS_BLOCK)); }

}  // namespace

/* The output rows of a launch, [row][slot] per stream, into POINT-MAJOR arrays [point][row] - what a consumer
 * that owns one series per point wants (SaveOutput writes a point's arrays, src/InputOutput.f90:151-165;
 * OutputData.cpp:5-13).  One wavefront per 64 slots and stream: a tile of up to 32 rows is read row by row
 * (coalesced, 512 B each) into LDS, then written point by point - a point's rows of the tile are contiguous in
 * its series, so every store instruction fills whole lines.  (Scattering single values into [row][point] arrays
 * instead would ask the memory for eight times the bytes.) */
struct ByPointArgs {
  const double *src[6];
  double *dst[6];
  const int32_t *order;
  int64_t npoints, src_stride, dst_rows, dst_row0;
  int32_t nrows;
};
__global__ void __launch_bounds__(64) outputs_by_point_kernel(const ByPointArgs a) {
  __shared__ double tile[32][65]; /* 16.6 KB: nine wavefronts to a CU keep enough loads in flight */
  __shared__ int32_t pt[64];
  const int lane = threadIdx.x;
  const int64_t s0 = (int64_t)blockIdx.x * 64;
  const double *src = a.src[blockIdx.y];
  double *dst = a.dst[blockIdx.y];
  const int64_t s = s0 + lane;
  pt[lane] = s < a.npoints ? a.order[s] : -1;
  const int rr = lane & 31, half = lane >> 5;
  for (int32_t r0 = 0; r0 < a.nrows; r0 += 32) {
    const int32_t nr = a.nrows - r0 < 32 ? a.nrows - r0 : 32;
    __syncthreads();
    if (s < a.npoints)
      for (int32_t r = 0; r < nr; ++r) tile[r][lane] = src[(int64_t)(r0 + r) * a.src_stride + s];
    __syncthreads();
    for (int jj = 0; jj < 32; ++jj) { /* two points per instruction: 32 rows = 256 B of each one's series */
      const int j = 2 * jj + half;
      const int32_t p = pt[j];
      if (p >= 0 && rr < nr) dst[(int64_t)p * a.dst_rows + a.dst_row0 + r0 + rr] = tile[rr][j];
    }
  }
}
hipError_t rs_cluster_outputs_by_point(const double *const src[6], double *const dst[6], const int32_t *order,
                                       int64_t npoints, int64_t src_stride, int32_t nrows, int64_t dst_rows,
                                       int64_t dst_row0, hipStream_t stream) {
  ByPointArgs a;
  for (int f = 0; f < 6; ++f) {
    a.src[f] = src[f];
    a.dst[f] = dst[f];
  }
  a.order = order;
  a.npoints = npoints;
  a.src_stride = src_stride;
  a.dst_rows = dst_rows;
  a.dst_row0 = dst_row0;
  a.nrows = nrows;
  /* (a variant without LDS - every lane carrying its slot's values into its point's series, the L2 merging eight
   * rows to a line - was slower: 438 against 388 ms per pass at 1 M points) */
  hipLaunchKernelGGL(outputs_by_point_kernel, dim3((unsigned)((npoints + 63) / 64), 6), dim3(64), 0, stream, a);
  return hipGetLastError();
}

hipError_t rs_cluster_identity(int32_t *order, int64_t np_pad, hipStream_t stream) {
  hipLaunchKernelGGL(iota_kernel, grid1(np_pad), dim3(RS_BLOCK), 0, stream, order, np_pad);
  return hipGetLastError();
}

size_t rs_cluster_scratch_bytes(int64_t npoints) {
  size_t bytes = 0;
  uint32_t *k = nullptr;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k, k, k, k, (int)npoints, 0, KEY_BITS);
  return bytes;
}

/* scratch: [4][np_pad] uint32 (keys in/out, slots in/out) + `tmp` for hipCUB.
 * Leaves the permutation (new slot -> old slot) in scratch + 3*np_pad. */
hipError_t rs_cluster_sort(const double *state, bool f32, int64_t np_pad, int64_t npoints,
                           uint32_t *scratch, void *tmp, size_t tmp_bytes, hipStream_t stream) {
  uint32_t *kin = scratch, *kout = scratch + np_pad, *sin = scratch + 2 * np_pad,
           *sout = scratch + 3 * np_pad;
  if (f32)
    hipLaunchKernelGGL(keys_kernel<float>, grid1(npoints), dim3(RS_BLOCK), 0, stream,
                       reinterpret_cast<const float *>(state), np_pad, npoints, kin, sin);
  else
    hipLaunchKernelGGL(keys_kernel<double>, grid1(npoints), dim3(RS_BLOCK), 0, stream, state, np_pad,
                       npoints, kin, sin);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  /* stable: points with equal scores keep their relative order */
  return hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, kin, kout, sin, sout, (int)npoints, 0,
                                            KEY_BITS, stream);
}

/* ---- the plan's own sort: one stable counting pass -----------------------------------------
 * The forecast key of the default field set is short - something on the road (1 bit), unstable
 * previews and table-path previews (2 bits each for the two or three previews of a window), predicted
 * extra passes saturating at 31 (5 bits), storage class (2 bits): 12 bits - so ONE counting pass
 * sorts it, in three small kernels instead of the 17 of the library's merge sort:
 *   cs_hist     a histogram per tile of CS_TILE keys (LDS atomics): H[tile][bin]
 *   cs_binscan  one lane per bin: exclusive prefix of its counts over the tiles, in place, and the
 *               bin's total (every access a coalesced row of H)
 *   cs_scatter  one wavefront per tile: scans the bins' totals (where every bin starts), start of
 *               (bin, tile) = the bin's start + its tile prefix;
 *               then it walks its keys in order, and the rank of a key among the equal keys of
 *               its 64 is a popcount of ballots - equal keys keep their order (stable, deterministic:
 *               no atomic decides a position, no workgroup waits for another)
 * perm_out[new slot] = old slot, as the library sort's value output. */
namespace {
constexpr int CS_TILE = 1024;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t u = (uint32_t)__shfl_up((int)v, off, 64);
    if (lane >= (uint32_t)off) v += u;
  }
  return v;
}

/* via: the permutation an earlier (less significant) pass left - this pass takes its keys in that order
 * (NULL: in slot order); shift: where this pass's digit starts in the key */
__global__ void __launch_bounds__(256) cs_hist_kernel(const uint32_t *__restrict__ keys,
                                                      const uint32_t *__restrict__ via, int64_t n, int shift,
                                                      int nbins, int ntiles, uint32_t *__restrict__ H) {
  __builtin_amdgcn_s_setprio(3); /* a link of the chain between two step launches of a plan: little work, all of it waited for */
  extern __shared__ uint32_t cs_h[];
  for (int b = threadIdx.x; b < nbins; b += 256) cs_h[b] = 0u;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * CS_TILE;
  for (int i = threadIdx.x; i < CS_TILE; i += 256) {
    const int64_t idx = base + i;
    if (idx < n) atomicAdd(&cs_h[(keys[via ? (int64_t)via[idx] : idx] >> shift) & (uint32_t)(nbins - 1)], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nbins; b += 256) H[(int64_t)blockIdx.x * nbins + b] = cs_h[b];
}

/* one lane per bin: walks the tiles in order (coalesced rows of H[tile][bin]), leaves the bin's
 * exclusive prefix over the tiles in place and its total in T[bin] */
__global__ void __launch_bounds__(64) cs_binscan_kernel(uint32_t *H, uint32_t *T, int nbins, int ntiles,
                                                        uint32_t *class_total, int class_shift) {
  __builtin_amdgcn_s_setprio(3); /* a link of the chain between two step launches of a plan: little work, all of it waited for */
  const int bin = (int)blockIdx.x * 64 + (int)threadIdx.x;
  uint32_t run = 0u;
  if (bin < nbins) {
#pragma unroll 8
    for (int t = 0; t < ntiles; ++t) {
      const uint32_t v = H[(int64_t)t * nbins + bin];
      H[(int64_t)t * nbins + bin] = run;
      run += v;
    }
    T[bin] = run;
  }
  /* for the wave table (cs_wave_table_kernel): the keys per CLASS, class = bin >> class_shift.  A
   * workgroup's 64 bins belong to one class (class_shift >= 6): one atomic per workgroup */
  if (class_total) {
    uint32_t sum = run;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += (uint32_t)__shfl_xor((int)sum, off, 64);
    if (threadIdx.x == 0 && (int)blockIdx.x * 64 < nbins) atomicAdd(&class_total[((int)blockIdx.x * 64) >> class_shift], sum);
  }
}

__global__ void __launch_bounds__(64) cs_scatter_kernel(const uint32_t *__restrict__ keys,
                                                        const uint32_t *__restrict__ via, int64_t n,
                                                        int shift, int nbits, int ntiles,
                                                        const uint32_t *__restrict__ H,
                                                        const uint32_t *__restrict__ T,
                                                        uint32_t *__restrict__ perm_out) {
  __builtin_amdgcn_s_setprio(3); /* a link of the chain between two step launches of a plan: little work, all of it waited for */
  extern __shared__ uint32_t cs_cur[];
  const int nbins = 1 << nbits;
  const uint32_t lane = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * CS_TILE;
  /* where every bin starts for this tile: the bin's start - the exclusive scan of the bins' totals,
   * which every wavefront forms for itself, a row of 64 bins at a time (64 independent coalesced loads,
   * a wavefront scan per row and the carry of the rows before: cheaper than a launch of its own, and no
   * workgroup waits for another) - plus its prefix over the tiles before this one (cs_binscan) */
  {
    /* the rows' loads are issued a batch at a time (round 3 issued them one row at a time, behind the
     * scan of the row before: 64 L2 round trips in a row, 80 of the kernel's 85 us on a 62 500-point
     * plan, where nothing else hides them) */
    constexpr int ROWS = 16;
    uint32_t carry = 0u;
    for (int r0 = 0; r0 < nbins; r0 += 64 * ROWS) {
      uint32_t tv[ROWS], hv[ROWS];
#pragma unroll
      for (int q = 0; q < ROWS; ++q) {
        const int b = r0 + q * 64 + (int)lane;
        tv[q] = b < nbins ? T[b] : 0u;
        hv[q] = b < nbins ? H[(int64_t)blockIdx.x * nbins + b] : 0u;
      }
#pragma unroll
      for (int q = 0; q < ROWS; ++q) {
        const int b = r0 + q * 64 + (int)lane;
        const uint32_t incl = wave_incl_scan(tv[q], lane);
        if (b < nbins) cs_cur[b] = carry + incl - tv[q] + hv[q];
        carry += (uint32_t)__shfl((int)incl, 63, 64);
      }
    }
  }
  __syncthreads();
  /* the keys of CS_BATCH rounds are fetched together (a load per round would put 64 memory round trips
   * behind one another), but no more: the kernel has to fit beside the step kernels' wavefronts, which
   * leave few vector registers free on a SIMD (64 keys in registers: 193 VGPRs, 0.5 ms per sort) */
  constexpr int CS_BATCH = 8;
  for (int r0 = 0; r0 < CS_TILE / 64; r0 += CS_BATCH) {
    if (base + (int64_t)r0 * 64 >= n) break; /* uniform: behind the last key */
    uint32_t kreg[CS_BATCH], sreg[CS_BATCH];
#pragma unroll
    for (int q = 0; q < CS_BATCH; ++q) {
      const int64_t idx = base + (int64_t)(r0 + q) * 64 + lane;
      sreg[q] = (idx < n && via) ? via[idx] : (uint32_t)idx;
    }
#pragma unroll
    for (int q = 0; q < CS_BATCH; ++q) {
      const int64_t idx = base + (int64_t)(r0 + q) * 64 + lane;
      kreg[q] = idx < n ? ((keys[sreg[q]] >> shift) & (uint32_t)(nbins - 1)) : 0u;
    }
#pragma unroll
  for (int q = 0; q < CS_BATCH; ++q) {
    const int64_t idx = base + (int64_t)(r0 + q) * 64 + lane;
    const bool valid = idx < n;
    const uint32_t key = kreg[q];
    unsigned long long same = __ballot(valid);
    for (int bit = 0; bit < nbits; ++bit) {
      const bool one = (key >> bit) & 1u;
      const unsigned long long b = __ballot(one);
      same &= one ? b : ~b;
    }
    const uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
    uint32_t start = 0u;
    if (valid) {
      start = cs_cur[key];
      perm_out[start + rank] = sreg[q];
    }
    /* every lane has read its cursor before the first of its group moves it: the workgroup is ONE
     * wavefront, whose LDS accesses execute in program order - a scheduling fence for the compiler is
     * all it takes (a __syncthreads() here also waits for the scattered store above: 64 memory round
     * trips per tile, measured 0.5 ms per sort) */
    __builtin_amdgcn_wave_barrier();
    if (valid && rank == 0u) cs_cur[key] = start + (uint32_t)__popcll(same);
    __builtin_amdgcn_wave_barrier();
  }
  }
}
}  // namespace

/* Wave table for the two-wavefront flavour (rs_kernels.hip, step_kernel_duo): wavefront w steps the slots
 * [start[w], start[w] + cnt[w]).  A CLASS = the slots whose sort key shares its class_bits most significant
 * bits (cover, unstable previews, table-path previews for the default key): every class starts a wavefront of
 * its own, so that no wavefront mixes two classes - the slowest wavefront of a small shard's launch is a MIXED
 * one (tools/bl_makespan.py), and the launch is as long as its slowest wavefront.  The slot space is
 * untouched (windows, state, order rows stay dense); only the kernel's wave -> slot mapping changes, at the
 * price of at most one partly filled wavefront per class.  One workgroup of 64 lanes: lane c takes the
 * keys of class c (counted by cs_binscan_kernel), two wavefront scans give every class its first slot and
 * first wavefront. */
__global__ void __launch_bounds__(64) cs_wave_table_kernel(uint32_t *__restrict__ class_total,
                                                           int class_bits, int32_t *__restrict__ wstart,
                                                           int32_t *__restrict__ wcnt, int32_t maxw) {
  __builtin_amdgcn_s_setprio(3); /* a link of the chain between two step launches of a plan: little work, all of it waited for */
  const uint32_t lane = threadIdx.x;
  const int nclasses = 1 << class_bits;
  uint32_t n = 0u;
  if ((int)lane < nclasses) {
    n = class_total[lane]; /* left by cs_binscan_kernel */
    class_total[lane] = 0u; /* ... and zero again for the next sort */
  }
  const uint32_t waves = (n + 63u) >> 6;
  const uint32_t sbase = wave_incl_scan(n, lane) - n, wbase = wave_incl_scan(waves, lane) - waves;
  uint32_t total = 0u;
  for (int c = 0; c < nclasses; ++c) {
    const uint32_t nc = (uint32_t)__shfl((int)n, c, 64), sb = (uint32_t)__shfl((int)sbase, c, 64),
                   wb = (uint32_t)__shfl((int)wbase, c, 64), wc = (nc + 63u) >> 6;
    for (uint32_t k = lane; k < wc; k += 64u) {
      const uint32_t idx = wb + k;
      if ((int32_t)idx < maxw) {
        wstart[idx] = (int32_t)(sb + 64u * k);
        const uint32_t left = nc - 64u * k;
        wcnt[idx] = (int32_t)(left < 64u ? left : 64u);
      }
    }
    total = wb + wc;
  }
  for (uint32_t idx = total + lane; (int32_t)idx < maxw; idx += 64u) { /* the launch's spare wavefronts */
    wstart[idx] = 0;
    wcnt[idx] = 0;
  }
}

/* class_total: the counters rs_cluster_count_sort filled (64 x uint32, zero before the sort; zero again after) */
hipError_t rs_cluster_wave_table(int class_bits, uint32_t *class_total, int32_t *wstart, int32_t *wcnt,
                                 int32_t maxw, hipStream_t stream) {
  if (class_bits < 1 || class_bits > 6 || !class_total) return hipErrorInvalidValue;
  hipLaunchKernelGGL(cs_wave_table_kernel, dim3(1), dim3(64), 0, stream, class_total, class_bits, wstart, wcnt, maxw);
  return hipGetLastError();
}

size_t rs_cluster_count_scratch_bytes(int64_t npoints, int nbits) {
  const int64_t ntiles = (npoints + CS_TILE - 1) / CS_TILE;
  return ((size_t)(1 << nbits) * (size_t)ntiles + (size_t)(1 << nbits)) * sizeof(uint32_t);
}

/* keys in scratch[0..npoints) (values < 2^(nbits + low_bits)), permutation out to scratch + 3*np_pad.
 * low_bits > 0: the key carries a second, less significant digit below its nbits - sorted first, by a pass
 * of its own whose permutation (scratch + np_pad) the main pass reads its keys through: two stable passes,
 * least significant digit first. */
hipError_t rs_cluster_count_sort(int64_t np_pad, int64_t npoints, int nbits, uint32_t *scratch, void *tmp,
                                 size_t tmp_bytes, hipStream_t stream, uint32_t *class_total, int class_bits,
                                 int low_bits) {
  if (nbits < 1 || nbits > 12 || low_bits < 0 || low_bits > 12 ||
      tmp_bytes < rs_cluster_count_scratch_bytes(npoints, nbits > low_bits ? nbits : low_bits))
    return hipErrorInvalidValue;
  const int ntiles = (int)((npoints + CS_TILE - 1) / CS_TILE);
  uint32_t *H = static_cast<uint32_t *>(tmp);
  const uint32_t *via = nullptr;
  if (low_bits > 0) {
    const int nb = 1 << low_bits;
    uint32_t *T = H + (size_t)nb * ntiles;
    hipLaunchKernelGGL(cs_hist_kernel, dim3(ntiles), dim3(256), nb * sizeof(uint32_t), stream, scratch, via,
                       npoints, 0, nb, ntiles, H);
    hipLaunchKernelGGL(cs_binscan_kernel, dim3((nb + 63) / 64), dim3(64), 0, stream, H, T, nb, ntiles,
                       (uint32_t *)nullptr, 0);
    hipLaunchKernelGGL(cs_scatter_kernel, dim3(ntiles), dim3(64), nb * sizeof(uint32_t), stream, scratch, via,
                       npoints, 0, low_bits, ntiles, H, T, scratch + np_pad);
    via = scratch + np_pad;
  }
  const int nbins = 1 << nbits;
  uint32_t *T = H + (size_t)nbins * ntiles;
  hipLaunchKernelGGL(cs_hist_kernel, dim3(ntiles), dim3(256), nbins * sizeof(uint32_t), stream, scratch, via,
                     npoints, low_bits, nbins, ntiles, H);
  /* classes of at least 64 bins only (a workgroup of the bin scan = one class) */
  const bool classes = class_total && class_bits >= 1 && nbits - class_bits >= 6;
  hipLaunchKernelGGL(cs_binscan_kernel, dim3((nbins + 63) / 64), dim3(64), 0, stream, H, T, nbins, ntiles,
                     classes ? class_total : nullptr, nbits - class_bits);
  hipLaunchKernelGGL(cs_scatter_kernel, dim3(ntiles), dim3(64), nbins * sizeof(uint32_t), stream, scratch, via,
                     npoints, low_bits, nbits, ntiles, H, T, scratch + 3 * np_pad);
  return hipGetLastError();
}

hipError_t rs_cluster_sort_keys(int64_t np_pad, int64_t npoints, uint32_t *scratch, void *tmp,
                                size_t tmp_bytes, hipStream_t stream) {
  uint32_t *kin = scratch, *kout = scratch + np_pad, *sin = scratch + 2 * np_pad,
           *sout = scratch + 3 * np_pad;
  return hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, kin, kout, sin, sout, (int)npoints, 0,
                                            KEY_BITS, stream);
}

namespace {
__global__ void __launch_bounds__(RS_BLOCK) again_flags_kernel(const double *__restrict__ state,
                                                               int64_t np_pad, int64_t npoints,
                                                               int32_t *flags, int32_t *iota) {
  const int64_t p = (int64_t)blockIdx.x * RS_BLOCK + threadIdx.x;
  if (p >= npoints) return;
  flags[p] = ((int32_t)state[(int64_t)RS_ST_CPL_FLAGS * np_pad + p]) & 1; /* start_coupling_again */
  iota[p] = (int32_t)p;
}
}  // namespace

size_t rs_cpl_select_scratch_bytes(int64_t npoints) {
  size_t bytes = 0;
  int32_t *x = nullptr;
  (void)hipcub::DeviceSelect::Flagged(nullptr, bytes, x, x, x, x, (int)npoints);
  return bytes;
}

/* flags: scratch int32[2*npoints] (flags, then the identity list to select from) */
hipError_t rs_cpl_select_again(const double *state, int64_t np_pad, int64_t npoints, int32_t *flags,
                               int32_t *list, int32_t *count_dev, void *tmp, size_t tmp_bytes,
                               hipStream_t stream) {
  int32_t *iota = flags + npoints;
  hipLaunchKernelGGL(again_flags_kernel, grid1(npoints), dim3(RS_BLOCK), 0, stream, state, np_pad,
                     npoints, flags, iota);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return hipcub::DeviceSelect::Flagged(tmp, tmp_bytes, iota, flags, list, count_dev, (int)npoints, stream);
}

/* nlayers: NLayers of the plan; cpl_rows: 0 no coupling block, 1 the coupling scalars only (every
 * coupling window is behind the plan: nothing reads the saved state any more), 2 everything */
hipError_t rs_cluster_apply(const double *state_src, double *state_dst, bool f32,
                            const int32_t *order_src, int32_t *order_dst, const uint32_t *perm,
                            int64_t np_pad, int64_t npoints, int nlayers, int cpl_rows,
                            hipStream_t stream) {
  dim3 g = grid1(np_pad);
  RowMap rows;
  rows.nlayers = nlayers;
  rows.nscal = RS_ST_BLSCORE - RS_ST_TNW1 + 1;
  rows.ncpl = cpl_rows >= 1 ? RS_ST_CPL_RESUME - RS_ST_CPL_ITER + 1 : 0;
  rows.nsave = cpl_rows >= 2 ? RS_ST_CPL_SAVE_ALBEDO - RS_ST_CPL_SAVE_TSURF + 1 : 0;
  g.y = (unsigned)rows.total();
  if (f32)
    hipLaunchKernelGGL(apply_kernel<float>, g, dim3(RS_BLOCK), 0, stream,
                       reinterpret_cast<const float *>(state_src), reinterpret_cast<float *>(state_dst),
                       order_src, order_dst, perm, np_pad, npoints, rows);
  else
    hipLaunchKernelGGL(apply_kernel<double>, g, dim3(RS_BLOCK), 0, stream, state_src, state_dst,
                       order_src, order_dst, perm, np_pad, npoints, rows);
  return hipGetLastError();
}
