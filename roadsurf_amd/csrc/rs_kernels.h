/* rs_kernels.h — kernel argument blocks and host launchers (internal). */
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/roadsurf.h"
#include "rs_synth.h"
#include "rs_raw.hpp"

#define RS_BLOCK 256

enum { RS_VARIANT_AUTO = 0, RS_VARIANT_REG = 1, RS_VARIANT_LDS = 2, RS_VARIANT_DUO = 3, RS_VARIANT_HYBRID = 4 };
/* AUTO takes the two-wavefronts-per-64-points flavour for launches of at most this many points:
 * 1 024 wavefronts, a quarter of the chip's slots (measured, tools/r3_duo.sh: two plans of 62 500
 * points 1.13e10 against 1.06e10 point-timesteps/s with one point per lane; four such plans in
 * flight at once are better off with one point per lane - a caller that runs that many sets the
 * flavour itself, as bench.py does) */
#ifndef RS_DUO_MAX_POINTS
#define RS_DUO_MAX_POINTS 65536
#endif

namespace rs {

/* where the sky-view kernels leave the reference's in-place edits of the input arrays
 * (rs_hip_set_writeback): rows like the forcing window's, all NULL = not wanted */
struct Writeback {
  double *sw, *sw_dir, *lw;
  int64_t t_stride;
};

struct StepArgs {
  const void *consts; /* the plan's constants in HBM: RsConstants (fp64 kernels) or RsConstantsF
                         (fp32 kernels); read through the scalar cache (address space 4) */
  RsForcing f;
  RsOutputs o;
  RsPointParams pp;
  double *state;
  int64_t npoints, np_pad;
  int32_t t0, nsteps;
  Writeback wb;
  /* coupling rounds (step_kernel_coupled): thread g works on point cpl_list[g] (NULL: point g);
   * cpl_stop: a point parks right after the Coupling_control of its window end */
  const int32_t *cpl_list;
  int32_t cpl_nlist, cpl_stop;
  /* lock-step replay kernels: a listed point may run up to this many replays of its window in ONE
   * launch, until its Coupling_control stops asking (0 or 1: one replay, the round structure of
   * rs_hip_cpl_replay); cpl_prio: raise the wavefronts' issue priority (a sparse late round beside
   * another plan's full launches runs at the speed of its own dependency chain) */
  int32_t cpl_inner, cpl_prio;
  /* coupling kernels: the outputs of slot s go to column out_index[s] of the output window
   * (NULL: column s).  With the plan order as index the (decimated) outputs land in point order
   * whatever order the slots are in (rs_hip_set_output_by_point). */
  const int32_t *out_index;
  /* two-wavefront flavour: wavefront w steps the slots [wave_start[w], wave_start[w] + wave_cnt[w]) and the
   * launch has wave_n workgroups (rs_cluster_wave_table: no wavefront mixes two classes of the sort key);
   * NULL: wavefront w steps the slots 64 w ... */
  const int32_t *wave_start, *wave_cnt;
  int32_t wave_n;
  /* the launch's FULL feature set is one the two-wavefront flavour has (rs_hip_step: no sky view, no
   * coupling, no depth stream, no tsurfOutputDepth) */
  int32_t duo_full_ok;
  /* two-wavefront flavour: the surface wave runs at raised issue priority (rs_api.hip: set while all live
   * plans of the device together leave its SIMDs underfilled) */
  int32_t surface_prio;
  /* two-wavefront flavour on the synthetic workload (rs_hip_step_knots): no forcing window - the ground
   * wave makes the forcing of the next index from the hourly knots itself, with expand_kernel's arithmetic
   * (knots [knot - knot_k0][RS_KNOT_FIELDS][np_pad] in point order, column knot_gather[slot]; NULL knots:
   * the window `f`) */
  const double *knots;
  const int32_t *knot_gather;
  int32_t knot_k0, knot_n, spk, start_hour;
  double r_spk;
  /* two-wavefront flavour behind rs_driver_run (rs_step_raw): no forcing window either - the ground wave
   * makes the forcing of the next index from the RAW series (JsonSource::interpolate + the GetWeather
   * overlay, rs_raw.hpp); raw.nsrc = 0: not this launch */
  RawForcing raw;
  /* the plan's diagnostics block (rs_hip_set_diagnostics; rs_state.h RsDiagRow), or NULL */
  double *diag;
};

struct InitArgs {
  const void *consts;
  RsForcing f;
  RsPointParams pp;
  double *state;
  int64_t npoints, np_pad;
};

struct ForecastArgs {
  const void *consts; /* RsConstants (the predictor runs in fp64 for either flavour) */
  const void *state;
  int32_t f32;        /* the state block holds floats */
  int64_t npoints, np_pad;
  RsPreview pv;
  uint32_t *keys, *slots;
  int32_t compact; /* 1: the key is stored right-aligned in its own bits (counting sort), 0: left-aligned
                      in RS_SORT_KEY_BITS (library sort) */
  int32_t low_bits; /* compact keys: bits of the ground digit below the others (rs_forecast_key_low_bits) */
  int32_t extra_log; /* field 0 of the mode: 0 = the previews' extra passes summed, saturating at 7; 1 = the
                        LONGEST loop expected in the window (previews, the last index stepped, a passage through
                        the loop's slow band) in classes 5, 6, 7, 8, 9-12, 13-20, 21-30, 31+ */
};

struct KnotArgs {
  RsSynthSpec spec;
  double *knots;
  int64_t npoints, np_pad;
  int32_t k0, nknots;
};

struct ExpandArgs {
  RsForcing f;
  const double *knots;
  int64_t npoints, np_pad;
  int32_t k0, t0, spk, start_hour;
  int32_t kfirst, nsteps;
  double r_spk; /* RN(1 / spk): the interpolation's division has a uniform denominator (rs_div_u) */
  const int32_t *gather; /* NULL, or: the knots of window column p are knot column gather[p] */
};

}  // namespace rs

hipError_t rs_read_div_mismatch(unsigned long long *out /*[3]*/, hipStream_t stream);
hipError_t rs_read_div_samples(double *out /*[64][4]*/, hipStream_t stream);
hipError_t rs_read_bl_stats(unsigned long long *out /*[56]*/, hipStream_t stream);
hipError_t rs_launch_math_test(int fn, int64_t n, const double *x, double *y, hipStream_t stream);
/* raw-series Tdew<->RH completion (needs the math tables: create a plan first) */
hipError_t rs_launch_humidity_fill(const double *tair, double *tdew, double *rhz, int64_t n,
                                   hipStream_t stream);
/* exp/log tables of the device (same for every plan) */
hipError_t rs_upload_math_tables(hipStream_t stream);
hipError_t rs_launch_step(const rs::StepArgs &a, int NL, bool full, int variant, bool score,
                          hipStream_t stream);
/* the two-wavefront flavour with the forcing made from the knots in the kernel (StepArgs::knots) */
hipError_t rs_launch_step_duo_knots(const rs::StepArgs &a, bool score, hipStream_t stream);
/* ... and from the raw series of the driver path (StepArgs::raw); sky: per-point sky view on the ground wave */
hipError_t rs_launch_step_duo_raw(const rs::StepArgs &a, bool score, bool sky, bool cpl, hipStream_t stream);
/* the step of rs_driver_run's blocks (rs_api.hip): NLayers = 15, fp64, no output depth; pp in SLOT order, raw
 * series in point order behind raw.col.  A coupled plan (use_coupling, pp->coupling_index): a LOCK-STEP chunk as
 * rs_hip_step_cpl runs it - points park behind their coupling window until rs_hip_cpl_replay has run. */
hipError_t rs_launch_step_duo_raw_replay(const rs::StepArgs &a, hipStream_t stream);
struct RsPlan;
/* the replay rounds of a coupled plan whose lock-step chunks run through rs_step_raw: as rs_hip_cpl_replay, the
 * forcing of the block [t0, t0 + nsteps) from the raw series (no sky view; the block must end before SimLen) */
int rs_cpl_replay_raw(RsPlan *pl, const rs::RawForcing *raw, const RsOutputs *o, const RsPointParams *pp,
                      int32_t t0, int32_t nsteps, bool out_by_point, int32_t *rounds);
/* Elements per stream of a window that the kernels with 32-bit byte offsets can address (2^29 doubles = 4 GiB).
 * The launchers that need it refuse larger windows; rs_driver_run cuts its point tiles so that a tile's output
 * window stays below it.  ROADSURF_HIP_A32_LIMIT (elements) can only LOWER it: the tests reach the limit with
 * windows of megabytes instead of gigabytes. */
uint64_t rs_a32_limit(void);
bool rs_step_raw_ok(const RsPlan *pl); /* a plan whose settings rs_step_raw can run */
int rs_step_raw(RsPlan *pl, const rs::RawForcing *raw, const double *sun, const RsOutputs *o,
                const RsPointParams *pp, int32_t t0, int32_t nsteps, bool out_by_point);
hipError_t rs_launch_step_cpl_replay(const rs::StepArgs &a, int NL, hipStream_t stream);
hipError_t rs_launch_step_coupled(const rs::StepArgs &a, int NL, hipStream_t stream);
hipError_t rs_launch_step_cpl(const rs::StepArgs &a, int NL, hipStream_t stream);
/* out[2] (device): min couplingStartI / max couplingEndI over the points that ask for a replay */
hipError_t rs_launch_cpl_window_bounds(const rs::StepArgs &a, int32_t *out, hipStream_t stream);
/* list of the points whose coupling asks for another replay (start_coupling_again): list[0..*count) */
size_t rs_cpl_select_scratch_bytes(int64_t npoints);
hipError_t rs_cpl_select_again(const double *state, int64_t np_pad, int64_t npoints, int32_t *flags,
                               int32_t *list, int32_t *count_dev, void *tmp, size_t tmp_bytes,
                               hipStream_t stream);
hipError_t rs_launch_step_sky(const rs::StepArgs &a, int NL, bool score, hipStream_t stream);
hipError_t rs_launch_init(const rs::InitArgs &a, hipStream_t stream);
hipError_t rs_launch_knots(const rs::KnotArgs &a, int32_t nknots, hipStream_t stream);
hipError_t rs_launch_expand(const rs::ExpandArgs &a, int32_t nintervals, hipStream_t stream);
/* out[0] = shader-clock ticks, out[1] = 100 MHz ticks over the same ~spin_us microseconds (spin_us is
 * clamped to RS_CLOCK_PROBE_MAX_US; out[1] = 0 if the 100 MHz counter did not advance) */
#define RS_CLOCK_PROBE_MAX_US 10000u
hipError_t rs_launch_clock_probe(uint64_t *out, uint32_t spin_us, hipStream_t stream);
hipError_t rs_launch_count_failed(const double *st, int64_t np_pad, int64_t npoints,
                                  unsigned long long *out, hipStream_t stream);

hipError_t rs_launch_forecast_keys(const rs::ForecastArgs &a, hipStream_t stream);

/* plan order (rs_cluster.hip) */
#define RS_SORT_KEY_BITS 24
hipError_t rs_cluster_identity(int32_t *order, int64_t np_pad, hipStream_t stream);
hipError_t rs_cluster_outputs_by_point(const double *const src[6], double *const dst[6], const int32_t *order,
                                       int64_t npoints, int64_t src_stride, int32_t nrows, int64_t dst_rows,
                                       int64_t dst_row0, hipStream_t stream);
size_t rs_cluster_scratch_bytes(int64_t npoints);
hipError_t rs_cluster_sort(const double *state, bool f32, int64_t np_pad, int64_t npoints,
                           uint32_t *scratch, void *tmp, size_t tmp_bytes, hipStream_t stream);
/* same with keys/slots already in scratch[0..np_pad) / scratch[2*np_pad..) */
hipError_t rs_cluster_sort_keys(int64_t np_pad, int64_t npoints, uint32_t *scratch, void *tmp,
                                size_t tmp_bytes, hipStream_t stream);
/* the plan's own stable counting sort for keys of at most 12 bits (rs_cluster.hip) */
size_t rs_cluster_count_scratch_bytes(int64_t npoints, int nbits);
hipError_t rs_cluster_wave_table(int class_bits, uint32_t *class_total, int32_t *wstart, int32_t *wcnt,
                                 int32_t maxw, hipStream_t stream);
hipError_t rs_cluster_count_sort(int64_t np_pad, int64_t npoints, int nbits, uint32_t *scratch, void *tmp,
                                 size_t tmp_bytes, hipStream_t stream, uint32_t *class_total = nullptr,
                                 int class_bits = 0, int low_bits = 0);
/* significant bits of the forecast key for a field list (RsPreview::mode), without the ground digit ... */
int rs_forecast_key_bits(int32_t mode);
/* ... and the bits of the ground digit below them (field 9: 7, else 0) */
int rs_forecast_key_low_bits(int32_t mode);
hipError_t rs_cluster_apply(const double *state_src, double *state_dst, bool f32,
                            const int32_t *order_src, int32_t *order_dst, const uint32_t *perm,
                            int64_t np_pad, int64_t npoints, int nlayers, int cpl_rows,
                            hipStream_t stream);

/* fp32 flavour (rs_kernels_f32.hip) */
/* single-precision mirror of the constants: fills *dst (device, rs32_constants_bytes() bytes) */
size_t rs32_constants_bytes(void);
hipError_t rs32_upload_constants(void *dst, const RsConstants *c, hipStream_t stream);
hipError_t rs32_launch_step(const rs::StepArgs &a, int NL, int variant, bool score, bool full, bool sky, hipStream_t stream);
hipError_t rs32_launch_step_coupled(const rs::StepArgs &a, int NL, hipStream_t stream);
hipError_t rs32_launch_step_knots(const rs::StepArgs &a, bool score, bool full, hipStream_t stream);
hipError_t rs32_launch_init(const rs::InitArgs &a, hipStream_t stream);
hipError_t rs32_launch_expand(const rs::ExpandArgs &a, int32_t nintervals, hipStream_t stream);
