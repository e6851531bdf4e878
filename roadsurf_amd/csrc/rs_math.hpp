/*
 * rs_math.hpp — the two transcendental functions on the hot path.
 *
 * exp: CalcLE x2 per step (src/BoundaryLayer.f90:160-170), CalcPrecType <=1
 *      (src/Cond.f90:230), relaxation <=3 (src/Relaxation.f90:36-42).
 * log: once per BLCond iteration in unstable stratification
 *      (src/BoundaryLayer.f90:87).
 * The reference calls glibc's exp/log (< 1 ulp, not correctly rounded).  The
 * device versions below are OCML's (<= 1 ulp): results can differ from glibc's
 * in the last bit.  tests/test_hip_parity.py measures what that does to the
 * outputs (the tolerance of the parity gate, 1e-6 K, is ~9 orders of magnitude
 * above it).
 */
#pragma once
#include <hip/hip_runtime.h>

namespace rs {
__device__ __forceinline__ double rs_exp(double x) { return ::exp(x); }
__device__ __forceinline__ double rs_log(double x) { return ::log(x); }
}  // namespace rs
