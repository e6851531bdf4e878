/*
 * rs_physics.hpp — one RoadSurf time step for one point, as device code.
 *
 * One point per lane.  The ground temperature profile is reached through a
 * small accessor (registers for the NLayers-specialised kernel, an LDS column
 * for the generic one); every other piece of carried state is a scalar in
 * VGPRs.  Layer constants come from RsConstants (kernel argument -> SGPRs).
 *
 * Arithmetic contract: fp64, IEEE + - * / sqrt in the
 * reference's evaluation order, no FMA contraction (the translation unit is
 * compiled with -ffp-contract=off), every unsuffixed Fortran literal is REAL(4)
 * and enters as R4(x) = (double)x##f.  exp/log are the only operations that are
 * not bit-defined by IEEE; see rs_math.hpp.
 *
 * The reference splits this work over ~20 subroutines that communicate through
 * derived types; here it is fused into one pass with the dead stores removed:
 *   - GCond, HS(2:), GroundFlux, SnowIceRat, WetSnowFrozen, Rain/SnowIntensity,
 *     PrecType, Tdew, SensibleHeatFlux are never read on this path;
 *   - condDZ is constant in time and comes from RsConstants;
 *   - VSH/capDZ/flux/update of a layer are one loop body with a rolling flux
 *     instead of three loops over heap/array temporaries
 *     (src/BalanceModel.f90:109 allocates Gflux every call);
 *   - atm%BLCond need not be carried: it only seeds BLCond_Old at j = 1 and the
 *     loop cannot exit before j = 5 (src/BoundaryLayer.f90:67,92).
 * Each block cites the reference lines it restates.
 */
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/roadsurf.h"
#include "rs_math.hpp"
#include "rs_consts_dev.h"


#define RS_REAL double
#define RS_NS rs
/* the constants are read as constant memory (scalar loads): rs_kernels.hip, consts_of */
#define RS_CONSTS RsConstantsDev __attribute__((address_space(4)))
/* a / b with b uniform and its reciprocal in the constant block */
#define RS_DIVC(a, b, rb) rs_div_u(a, b, c.rb)
#define RS_BL_GUARD 1
#define RS_MELTDEN c.meltDen /* WatMHeat*WatDens, the same IEEE product formed once on the host */
#define RS_BARE_FAST(c) ((c).bareFastOk != 0) /* rs_consts_dev.h */
#define RS_PREC_FAST(c) ((c).precFastOk != 0)
#define RS_CHK(c, i, lit) ((c).chk[i])         /* rs_consts_dev.h: CheckValues' bounds */
#define R4(x) ((double)(x##f))
#ifndef RS_NO_FROZEN_TABLE
#define RS_FROZEN_TABLE 1 /* layer_step: capDZ of a layer that is frozen in all 64 points from RsConstantsDev::capDZF */
#endif
/* the five per-layer constants of layer_step next to one another (RsConstantsDev::lk): one scalar load per
 * layer instead of one per table (-DRS_NO_LAYER_ROWS: the tables of RsConstants) */
#ifndef RS_NO_LAYER_ROWS
#define RS_LK(c, j, name) ((c).lk[j].name)
#else
#define RS_LK(c, j, name) ((c).name[j])
#endif
#ifndef RS_NO_HCW_TABLE
#define RS_HCW_TABLE 1 /* layer_vsh: the water polynomials' coefficients from RsConstantsDev::hcw */
#endif
#include "rs_physics_body.inc"
#undef RS_FROZEN_TABLE
#undef RS_HCW_TABLE
#undef RS_LK
#undef RS_REAL
#undef RS_NS
#undef RS_CONSTS
#undef RS_BL_GUARD
#undef RS_MELTDEN
#undef RS_BARE_FAST
#undef RS_PREC_FAST
#undef RS_CHK

#include "rs_skyview.hpp" /* sky_view_radiation */
