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
