/*
 * rs_skyview.hpp - sky view / local horizons for one point and one time index, shared by the fp64 flavours
 * (rs_physics.hpp) and the fp32 flavour (rs_kernels_f32.hip: the geometry and its decisions stay fp64 there too -
 * which degree of azimuth, sun above the horizon line or not -, only the radiation it hands the model is rounded).
 * Self-contained: the REAL(4) literals of the reference enter through RS_SKY_R4, not through the including
 * translation unit's R4.
 */
#pragma once
#include <hip/hip_runtime.h>
#include "rs_math.hpp"

#define RS_SKY_R4(x) ((double)(x##f))

namespace rs {

/* Sky view / local horizon: the per-point, per-step remainder of calcElevationAzimuth
 * (src/SunPosition.f90:123-193) and ModRadiationBySurroundings (src/ModRadiation.f90:7-73).
 * sun[RS_SUN_COLS] = {ra, stG, sin decl, cos decl, cos(stG - ra), sin(stG - ra)} comes from the host
 * (rs_sun_table, libm).  The
 * solar position only ever acts through discrete outcomes (sun above the horizon line or
 * not, which degree of azimuth, elevation > 0), so the device cos/acos need not reproduce
 * libm's last bit: a different outcome needs the elevation within ~1e-14 deg of the horizon
 * value or the azimuth within ~1e-13 deg of a half degree.  Returns false where the
 * reference would `stop` (|cos| >= 1.001: cannot happen for real inputs). */
__device__ __forceinline__ bool sky_view_radiation(const double *sun, double sin_lat,
                                                   double cos_lat, double lon_rad, double cos_lon,
                                                   double sin_lon, double sky_view,
                                                   double albedo_surr, const double *horizons,
                                                   int64_t hstride, double &sw, double &sw_dir,
                                                   double &lw, double lw_net) {
  const double pi = 3.141592653589793; /* 4*atan(1.0_8) */
  /* In the dark - global and direct short wave both +0.0 - the position of the sun cannot show:
   * above the horizon the reference forms SW_dir * shadow_fac = +0, SW_ref = a*0 + a*0, dif_SW =
   * sv*0 + (1-sv)*SW_ref and SW = dif_SW + SW_dir, which is +0.0 again for any finite albedo of the
   * surroundings (a sum of zeros of both signs is +0), below it nothing is touched; and SunPosition's
   * `stop` needs |cos| >= 1.001, out of reach of sines and cosines of finite angles.  So a lane in the
   * dark goes straight to the long-wave line: two cos, two acos and a division less per step for every
   * night-time index. */
  if (rs_is_pos_zero(sw) && rs_is_pos_zero(sw_dir) && __builtin_fabs(albedo_surr) < __builtin_inf()) {
    lw = sky_view * lw + (RS_SKY_R4(1.0) - sky_view) * (-(lw_net - lw));
    return true;
  }
  const double ra = sun[0], stG = sun[1], sin_decl = sun[2], cos_decl = sun[3];
  const double cos_dec_lat = cos_decl * cos_lat;
  const double sin_dec_lat = sin_decl * sin_lat;
  double hac = (stG + lon_rad - ra);
  /* cos of the hour angle by the addition theorem: cos((stG - ra) + lon) from the table's cos/sin of
   * stG - ra (host, once per time index) and the point's cos/sin of its longitude (once per launch) -
   * three instructions instead of a cosine per point-step.  Like every value of this block it reaches
   * the outputs only through decisions (elevation > 0, horizon > elevation, the rounded azimuth), see
   * the note above: a result that differs from cos(hac) in its last bits moves those by ~1e-14 deg. */
  const double cosah = sun[4] * cos_lon - sun[5] * sin_lon;
  const double cos_elev = sin_dec_lat + cos_dec_lat * cosah;
  double elevation, azimuth;
  if (cos_elev < -1e-9) {
    /* the sun is below the horizon by more than any rounding of acos: elevation <= 0 in the reference,
     * which then sets both to -9999.9 without looking at them again */
    azimuth = RS_SKY_R4(-9999.9);
    elevation = RS_SKY_R4(-9999.9);
  } else {
    double chi;
    if (cos_elev >= RS_SKY_R4(1.0) && cos_elev < RS_SKY_R4(1.001)) {
      chi = RS_SKY_R4(0.);
    } else if (cos_elev >= RS_SKY_R4(1.001)) {
      return false;
    } else if (cos_elev > RS_SKY_R4(-1.001) && cos_elev <= RS_SKY_R4(-1.0)) {
      chi = pi;
    } else {
      chi = ::acos(cos_elev);
    }
    elevation = RS_SKY_R4(90.0) - chi * (RS_SKY_R4(180.) / pi);
    if (hac < RS_SKY_R4(0.))
      hac = 2 * pi + hac;
    else if (hac > 2 * pi)
      hac = hac - 2 * pi;
    if (elevation > 0) {
      /* cos(pi/2 - chi) = sin(chi) = sqrt((1 - x)(1 + x)) for chi = acos(x) in [0, pi] */
      const double cosele = (cos_elev >= RS_SKY_R4(1.0)) ? 0.0 : rs_sqrt((1.0 - cos_elev) * (1.0 + cos_elev));
      if (cosele >= RS_SKY_R4(-0.0001) && cosele < RS_SKY_R4(0.0001)) {
        azimuth = RS_SKY_R4(-9999.9);
      } else {
        const double precos = rs_div(sin_decl * cos_lat - cos_decl * sin_lat * cosah, cosele);
        if (precos >= RS_SKY_R4(1.0) && precos < RS_SKY_R4(1.001))
          azimuth = RS_SKY_R4(0.0);
        else if (precos >= RS_SKY_R4(1.001))
          return false;
        else if (precos > RS_SKY_R4(-1.001) && precos <= RS_SKY_R4(-1.0))
          azimuth = pi;
        else
          azimuth = ::acos(precos);
      }
      if (hac < pi) azimuth = 2 * pi - azimuth;
      azimuth = azimuth * (RS_SKY_R4(180.) / pi);
    } else {
      azimuth = RS_SKY_R4(-9999.9);
      elevation = RS_SKY_R4(-9999.9);
    }
  }
  /* ModRadiationBySurroundings */
  double dif_sw = sw - sw_dir;
  const double lw_surroundings = lw_net - lw;
  int azim_idx = (int)__builtin_round(azimuth); /* NINT */
  if (azim_idx == 360) azim_idx = 0;
  double horizon = RS_SKY_R4(0.);
  if (horizons && azim_idx >= 0 && azim_idx < 360) horizon = horizons[(int64_t)azim_idx * hstride];
  const double shadow_fac = (horizon > elevation) ? RS_SKY_R4(0.0) : RS_SKY_R4(1.0);
  if (elevation > RS_SKY_R4(0.0)) {
    sw_dir = sw_dir * shadow_fac;
    const double sw_ref = albedo_surr * sw_dir + albedo_surr * dif_sw;
    dif_sw = sky_view * dif_sw + (RS_SKY_R4(1.0) - sky_view) * sw_ref;
    sw = dif_sw + sw_dir;
  }
  lw = sky_view * lw + (RS_SKY_R4(1.0) - sky_view) * (-lw_surroundings);
  return true;
}

}  // namespace rs

#undef RS_SKY_R4
