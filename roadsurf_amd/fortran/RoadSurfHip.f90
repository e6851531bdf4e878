!> RoadSurfHip — Fortran host orchestration of the MI355X RoadSurf hot path.
!!
!! This module is the host side of the device boundary: it keeps the
!! reference's own C-interop surface and calls the HIP kernels through an
!! ISO_C_BINDING shim (roadsurf_amd/csrc/rs_api.hip, rs_host.hip):
!!
!!  * the five Bind(C) types of the boundary, field for field
!!    (reference: src/InputPointers.f90.inc:4-27, src/OutputPointers.f90.inc:4-17,
!!     src/InputSettings.f90.inc:4-18, src/InputParameters.f90.inc:4-91,
!!     src/LocalParameters.f90.inc:4-15);
!!  * `runsimulation`, the reference's only BIND(C) procedure
!!    (examples/example1/src/Simulation.f90:4-6), same dummy arguments, now a
!!    one-point call into the batched GPU path;
!!  * `runsimulation_batch`, the many-point extension;
!!  * `rs_build_constants` / `rs_bottom_temperature`: everything the reference
!!    computes once per point in Initialization (src/Initialization.f90) that
!!    is in fact uniform over points — layer grid, conductivities, the four
!!    logarithms, REAL(4)-folded wear constants.  It is computed HERE, on the
!!    host, by the same compiler family that builds the reference, so default
!!    REAL literals, REAL(4) sub-expressions and `**` lower exactly as they do
!!    there (SURVEY.md Appendix C); the kernels only ever see the doubles.
!!
!! The time loop itself, and every per-step subroutine of the reference's
!! module RoadSurf, run on the device (roadsurf_amd/csrc/rs_physics.hpp).
module RoadSurfHip
   use, intrinsic :: iso_c_binding
   implicit none
   private

   integer, parameter, public :: RS_MAX_LAYERS = 32
   integer, parameter, public :: RS_DIAG_COLS = 13   ! include/roadsurf.h

   type, bind(C), public :: InputPointers
      integer(c_int) :: inputLen
      type(c_ptr) :: c_tair, c_tdew, c_VZ, c_Rhz, c_prec, c_SW, c_LW, c_SW_dir, c_LW_net
      type(c_ptr) :: c_TSurfObs, c_PrecPhase, c_local_horizons, c_Depth
      type(c_ptr) :: c_year, c_month, c_day, c_hour, c_minute, c_second
   end type InputPointers

   type, bind(C), public :: OutputPointers
      integer(c_int) :: outputLen
      type(c_ptr) :: c_TsurfOut, c_SnowOut, c_WaterOut, c_IceOut, c_DepositOut, c_Ice2Out
   end type OutputPointers

   type, bind(C), public :: InputSettings
      integer(c_int) :: SimLen, use_coupling, use_relaxation, force_tsurf
      real(c_double) :: DTSecs, tsurfOutputDepth
      integer(c_int) :: NLayers, coupling_minutes
      real(c_double) :: couplingEffectReduction
      integer(c_int) :: outputStep
   end type InputSettings

   type, bind(C), public :: InputParameters
      real(c_double) :: NightOn, NightOff, CalmLimDay, CalmLimNgt, TrfFricNgt, TrFfricDay
      real(c_double) :: Grav, SB_Const, VK_Const, LVap, LFus, WatDens, SnowDens, IceDens
      real(c_double) :: DepDens, WatMHeat, PorEvaF
      real(c_double) :: ZRefW, ZRefT, ZeroDisp, ZMom, ZHeat, Emiss, Albedo
      real(c_double) :: Albedo_surroundings, MaxPormms, TClimG, DampDpth, Omega, AZ, DampWearF
      real(c_double) :: AlbDry, AlbSnow, vsh1, vsh2, Poro1, Poro2, RhoB1, RhoB2, Silt1, Silt2
      real(c_double) :: freezing_limit_normal, snow_melting_limit_normal
      real(c_double) :: ice_melting_limit_normal, frost_melting_limit_normal
      real(c_double) :: frost_formation_limit_normal, T4Melt_normal
      real(c_double) :: TLimColdH, TLimColdL, WetSnowFormR, WetSnowMeltR
      real(c_double) :: PLimSnow, PLimRain, MaxSnowmms, MaxDepmms, MaxIcemms, MaxExtmms
      real(c_double) :: MissValI, MissValR
      real(c_double) :: Snow2IceFac
      real(c_double) :: MinPrecmm, MinWatmms, MinSnowmms
      real(c_double) :: MaxWatmms
      real(c_double) :: WDampLim, WWetLim
      real(c_double) :: WWearLim
      real(c_double) :: MinDepmms, MinIcemms
   end type InputParameters

   type, bind(C), public :: LocalParameters
      real(c_double) :: tair_relax, VZ_relax, RH_relax
      integer(c_int) :: couplingIndexI
      real(c_double) :: couplingTsurf, lat, lon, sky_view
      integer(c_int) :: InitLenI
   end type LocalParameters

   !> Mirror of `RsConstants` in include/roadsurf.h (same order, same types).
   type, bind(C), public :: RsConstants
      integer(c_int) :: NLayers, SimLen, use_relaxation, force_tsurf, use_coupling, cplLenI
      real(c_double) :: cplLenR, cplReduction
      real(c_double) :: DTSecs, Tph, tsurfOutputDepth, twoDT
      real(c_double) :: ZDpth(0:RS_MAX_LAYERS + 1)
      real(c_double) :: DyC(0:RS_MAX_LAYERS + 1)
      real(c_double) :: condDZ(0:RS_MAX_LAYERS + 1)
      real(c_double) :: WCont(0:RS_MAX_LAYERS + 1)
      real(c_double) :: dryCap(0:RS_MAX_LAYERS + 1)
      real(c_double) :: HSfac1
      real(c_double) :: logMom, logHeat, logCond, logUstar
      real(c_double) :: VK_Const, ZRefT, Grav, LVap, LFus
      real(c_double) :: Emiss, SB_Const, Albedo0
      real(c_double) :: NightOn, NightOff, CalmLimDay, CalmLimNgt, TrfFricNgt, TrFfricDay
      real(c_double) :: MaxPormms, MissValI, MinPrecmm, MinWatmms, MinSnowmms, MinDepmms
      real(c_double) :: MinIcemms, MaxSnowmms, MaxDepmms, MaxIcemms, MaxWatmms, AlbDry, AlbSnow
      real(c_double) :: WatDens, WatMHeat, PorEvaF, DampWearF, TLimFreeze, TLimMeltSnow
      real(c_double) :: TLimMeltIce, TLimMeltDep, TLimDew, TLimColdH, TLimColdL, WetSnowFormR
      real(c_double) :: WetSnowMeltR, PLimSnow, PLimRain, WWetLim, WWearLim, T4Melt0
      real(c_double) :: wSnowTran, wSnow2Ice, wIce, wIce2, wDep, wWat
   end type RsConstants

   !> Mirror of `RsHostExtras` in include/roadsurf.h.
   type, bind(C), public :: RsHostExtras
      type(c_ptr) :: sun, sin_lat, cos_lat, lon_rad
      real(c_double) :: albedo_surroundings
      type(c_ptr) :: first_failed
      integer(c_int) :: writeback
      type(c_ptr) :: diagnostics
   end type RsHostExtras

   interface
      subroutine rs_host_set_error(msg) bind(C, name='rs_host_set_error')
         import :: c_char
         character(kind=c_char), intent(in) :: msg(*)
      end subroutine rs_host_set_error
   end interface
   public :: rs_host_set_error

   public :: rs_build_constants, rs_bottom_temperature
   public :: rs_fortran_sizeof, rs_sun_table, rs_point_geometry

contains

   !> Size in bytes of the boundary types as this Fortran unit lays them out
   !! (cross-checked against the C header by tests/test_abi_layout.py).
   function rs_fortran_sizeof(which) bind(C, name='rs_fortran_sizeof') result(nbytes)
      integer(c_int), value :: which
      integer(c_int64_t) :: nbytes
      type(InputPointers) :: a0
      type(OutputPointers) :: a1
      type(InputSettings) :: a2
      type(InputParameters) :: a3
      type(LocalParameters) :: a4
      type(RsConstants) :: a5
      select case (which)
      case (0); nbytes = c_sizeof(a0)
      case (1); nbytes = c_sizeof(a1)
      case (2); nbytes = c_sizeof(a2)
      case (3); nbytes = c_sizeof(a3)
      case (4); nbytes = c_sizeof(a4)
      case (5); nbytes = c_sizeof(a5)
      case default; nbytes = -1
      end select
   end function rs_fortran_sizeof

   !> Day of year, 1 January = 1 (restates src/BalanceModel.f90:325-351).
   pure integer function day_of_year(year, month, day) result(doy)
      integer, intent(in) :: year, month, day
      integer, parameter :: before(12) = [0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334]
      integer :: leap
      leap = 1 - min(mod(year, 4), 1) + min(mod(year, 100), 1) - min(mod(year, 400), 1)
      doy = before(month) + day
      if (month > 2) doy = doy + leap
   end function day_of_year

   !> Uniform model constants (see include/roadsurf.h, RsConstants).
   subroutine rs_build_constants(inSettings, inputParam, c, status) bind(C, name='rs_build_constants')
      type(InputSettings), intent(in) :: inSettings
      type(InputParameters), intent(in) :: inputParam
      type(RsConstants), intent(out) :: c
      integer(c_int), intent(out) :: status

      integer :: n, k
      real(8) :: zadd, thick, cond
      real(8) :: a1, b1, c1, d1, a2, b2, c2, d2, efc

      status = 0
      n = inSettings%NLayers
      if (n < 5 .or. n > RS_MAX_LAYERS) then
         status = -1
         return
      end if
      if (inSettings%SimLen < 1 .or. .not. (inSettings%DTSecs > 0.0d0)) then
         status = -1
         return
      end if

      c%NLayers = n
      c%SimLen = inSettings%SimLen
      c%use_relaxation = merge(1, 0, inSettings%use_relaxation == 1)
      c%force_tsurf = merge(1, 0, inSettings%force_tsurf == 1)
      c%use_coupling = merge(1, 0, inSettings%use_coupling == 1)
      ! length of the coupling window in steps, as a real (for the comparison) and
      ! truncated (for the start index): src/Coupling.f90:512,516-517
      c%cplLenR = inSettings%coupling_minutes*60/inSettings%DTSecs
      c%cplLenI = int(inSettings%coupling_minutes*60/inSettings%DTSecs)
      c%cplReduction = inSettings%couplingEffectReduction
      c%DTSecs = inSettings%DTSecs
      c%Tph = inSettings%DTSecs/3600.0          ! src/Initialization.f90:92
      c%tsurfOutputDepth = inSettings%tsurfOutputDepth
      c%twoDT = 2.0*inSettings%DTSecs           ! src/BalanceModel.f90:241

      c%ZDpth = 0.0d0; c%DyC = 0.0d0; c%condDZ = 0.0d0; c%WCont = 0.0d0; c%dryCap = 0.0d0

      ! layer interfaces: thickness grows geometrically (src/Initialization.f90:217-235).
      ! The growth term is a default-REAL expression and must stay one.
      zadd = 0.02
      c%ZDpth(1) = 0.0
      do k = 1, n
         c%ZDpth(k + 1) = c%ZDpth(k) + 0.0103*1.4**(k - 1) + zadd
      end do

      ! node spacings (src/Initialization.f90:193-205)
      c%DyC(1) = (c%ZDpth(2) - c%ZDpth(1))/2.0
      do k = 2, n
         c%DyC(k) = (c%ZDpth(k + 1) - c%ZDpth(k - 1))/2.0
      end do
      c%HSfac1 = c%ZDpth(2) - c%ZDpth(1)

      ! water content and dry heat capacity per layer: two materials, the top
      ! two layers are asphalt (src/Initialization.f90:207-213, src/BalanceModel.f90:232-236)
      do k = 1, n
         if (k <= 2) then
            c%WCont(k) = 0.01
            c%dryCap(k) = (1.0 - inputParam%Poro1)*inputParam%vsh1
         else
            c%WCont(k) = 0.3
            c%dryCap(k) = (1.0 - inputParam%Poro2)*inputParam%vsh2
         end if
      end do

      ! conductivity (Campbell 1985) -> condDZ, constant in time
      ! (src/BalanceModel.f90:158-186, 254-279, 145-151)
      a1 = 0.65 - 0.78*inputParam%RhoB1 + 0.60*inputParam%RhoB1*inputParam%RhoB1
      b1 = 1.06*inputParam%RhoB1
      if (inputParam%Silt1 > 0.00001) then
         c1 = 1 + 2.6/sqrt(inputParam%Silt1)
      else
         c1 = 0.
      end if
      d1 = 0.03 + 0.1*inputParam%RhoB1*inputParam%RhoB1
      a2 = 0.65 - 0.78*inputParam%RhoB2 + 0.60*inputParam%RhoB2*inputParam%RhoB2
      b2 = 1.06*inputParam%RhoB2
      if (inputParam%Silt2 > 0.00001) then
         c2 = 1 + 2.6/sqrt(inputParam%Silt2)
      else
         c2 = 0.
      end if
      d2 = 0.03 + 0.1*inputParam%RhoB2*inputParam%RhoB2
      efc = 4
      do k = 1, n
         thick = c%ZDpth(k + 1) - c%ZDpth(k)
         if (k <= 2) then
            cond = a1 + b1*c%WCont(k) - (a1 - d1)*exp(-(c1*c%WCont(k))**efc)
         else
            cond = a2 + b2*c%WCont(k) - (a2 - d2)*exp(-(c2*c%WCont(k))**efc)
         end if
         c%condDZ(k) = -(cond/thick)
      end do

      ! boundary layer logarithms (src/Initialization.f90:330-337)
      c%logMom = log((inputParam%ZRefW + inputParam%ZMom)/inputParam%ZMom)
      c%logHeat = log((inputParam%ZRefW + inputParam%ZHeat)/inputParam%ZHeat)
      c%logCond = log((inputParam%ZRefW - inputParam%ZeroDisp + inputParam%ZHeat)/inputParam%ZHeat)
      c%logUstar = log((inputParam%ZRefW - inputParam%ZeroDisp + inputParam%ZMom)/inputParam%ZMom)
      c%VK_Const = inputParam%VK_Const
      c%ZRefT = inputParam%ZRefT
      c%Grav = inputParam%Grav
      c%LVap = inputParam%LVap
      c%LFus = inputParam%LFus
      c%Emiss = inputParam%Emiss
      c%SB_Const = inputParam%SB_Const
      c%Albedo0 = inputParam%Albedo

      c%NightOn = inputParam%NightOn
      c%NightOff = inputParam%NightOff
      c%CalmLimDay = inputParam%CalmLimDay
      c%CalmLimNgt = inputParam%CalmLimNgt
      c%TrfFricNgt = inputParam%TrfFricNgt
      c%TrFfricDay = inputParam%TrFfricDay

      ! storage thresholds (src/Initialization.f90:479-557)
      c%MaxPormms = inputParam%MaxPormms
      c%MissValI = inputParam%MissValI
      c%MinPrecmm = inputParam%MinPrecmm
      c%MinWatmms = inputParam%MinWatmms
      c%MinSnowmms = inputParam%MinSnowmms
      c%MinDepmms = inputParam%MinDepmms
      c%MinIcemms = inputParam%MinIcemms
      c%MaxSnowmms = inputParam%MaxSnowmms
      c%MaxDepmms = inputParam%MaxDepmms
      c%MaxIcemms = inputParam%MaxIcemms
      c%MaxWatmms = inputParam%MaxWatmms
      c%AlbDry = inputParam%AlbDry
      c%AlbSnow = inputParam%AlbSnow
      c%WatDens = inputParam%WatDens
      c%WatMHeat = inputParam%WatMHeat
      c%PorEvaF = inputParam%PorEvaF
      c%DampWearF = inputParam%DampWearF
      c%TLimFreeze = inputParam%freezing_limit_normal
      c%TLimMeltSnow = inputParam%snow_melting_limit_normal
      c%TLimMeltIce = inputParam%ice_melting_limit_normal
      c%TLimMeltDep = inputParam%frost_melting_limit_normal
      c%TLimDew = inputParam%frost_formation_limit_normal
      c%TLimColdH = inputParam%TLimColdH
      c%TLimColdL = inputParam%TLimColdL
      c%WetSnowFormR = inputParam%WetSnowFormR
      c%WetSnowMeltR = inputParam%WetSnowMeltR
      c%PLimSnow = inputParam%PLimSnow
      c%PLimRain = inputParam%PLimRain
      c%WWetLim = inputParam%WWetLim
      c%WWearLim = inputParam%WWearLim
      c%T4Melt0 = inputParam%T4Melt_normal

      ! traffic wear coefficients: default-REAL constant expressions
      ! (src/Cond.f90:78,86,89,92,96,100)
      c%wSnowTran = (0.2 + 0.25)
      c%wSnow2Ice = 0.25/(0.2 + 0.25)
      c%wIce = 1.1*2.0*0.145
      c%wIce2 = 1.1*2.0*(4.0*0.290)
      c%wDep = 0.5*2.0*(4.0*0.290)
      c%wWat = 0.145
   end subroutine rs_build_constants

   !> Climatological bottom-boundary temperature Tmp(NLayers+1) for a start date
   !! (src/Initialization.f90:266-268).
   function rs_bottom_temperature(inputParam, c, year, month, day) &
      bind(C, name='rs_bottom_temperature') result(tb)
      type(InputParameters), intent(in) :: inputParam
      type(RsConstants), intent(in) :: c
      integer(c_int), value :: year, month, day
      real(c_double) :: tb
      integer :: juld
      juld = day_of_year(int(year), int(month), int(day))
      tb = inputParam%TClimG + inputParam%AZ*sin(inputParam%Omega*juld + &
                                                  inputParam%Omega*(-170) - (c%ZDpth(c%NLayers + 1)/inputParam%DampDpth))
   end function rs_bottom_temperature

   !> Time-only solar quantities after Meeus, "Astronomical Algorithms" ch. 7, 12, 22, 25, as
   !! the reference evaluates them (src/SunPosition.f90:70-125, 241-258).  Default-REAL
   !! literals and REAL() conversions are deliberately left at default kind: the reference
   !! accumulates the day fraction in single precision and rounds its coefficients to REAL(4).
   subroutine rs_sun_table(n, year, month, day, hour, minute, second, table) bind(C, name='rs_sun_table')
      integer(c_int), value :: n
      integer(c_int), intent(in) :: year(n), month(n), day(n), hour(n), minute(n), second(n)
      real(c_double), intent(out) :: table(6, n)
      real(8), parameter :: pi = 4*atan(1.0_8)
      real(8) :: cy, yr, mo, dayf, ca, cb, jde, t, lmean, anom, centre, lapp, obl, ra, decl, sidereal, node
      integer :: k
      cy = 365.25
      do k = 1, n
         ! Julian Ephemeris Day
         if (month(k) <= 2) then
            yr = real(year(k) - 1)
            mo = real(month(k) + 12)
         else
            yr = real(year(k))
            mo = real(month(k))
         end if
         dayf = real(day(k)) + real(hour(k))/24. + real(minute(k))/(24.*60.) + real(second(k))/(24.*60.*60.)
         ca = dint(yr/100.)
         cb = 2. - ca + dint(ca/4.)
         jde = dint(cy*(yr + 4716)) + dint(30.6001*(mo + 1.)) + dayf + cb - 1.5245d3
         t = (jde - 2451545.0)/(cy*100.)
         ! geometric mean longitude and mean anomaly, reduced to [0, 360]
         lmean = 280.46645 + 36000.76983*t + 0.0003032*t*t
         if (lmean < 0.) lmean = lmean - 360.*(aint(lmean/360.) - 1.)
         if (lmean > 360.) lmean = lmean - 360.*aint(lmean/360.)
         anom = 357.52910 + 35999.05030*t - 0.0001559*t*t - 0.00000048*t*t*t
         if (anom < 0.) anom = anom - 360.*(aint(anom/360.) - 1.)
         if (anom > 360.) anom = anom - 360.*aint(anom/360.)
         ! equation of centre, apparent longitude, obliquity with nutation term
         centre = (1.913600 - 0.004817*t - 0.000014*t*t)*sin(anom*pi/180.) &
                  + (0.019993 - 0.000101*t)*sin(2.*anom*pi/180.) &
                  + 0.000290*sin(3.*anom*pi/180.)
         node = (125.04 - 1934.136*t)*pi/180.
         lapp = lmean + centre - 0.00569 - 0.00478*sin(node)
         lapp = lapp*pi/180.
         obl = 23.43929111 - 0.013004166*t - 0.001638888*t*t + 0.005036111*t*t*t
         obl = obl + 0.00256*cos(node)
         obl = obl*pi/180.
         ! right ascension in [0, 2 pi], declination
         ra = atan2(cos(obl)*sin(lapp), cos(lapp))
         if (ra < 0.) ra = ra - 2.*pi*(aint(ra/(2.*pi)) - 1.)
         if (ra > 2.*pi) ra = ra - 2.*pi*aint(ra/(2.*pi))
         decl = asin(sin(obl)*sin(lapp))
         ! mean sidereal time at Greenwich
         sidereal = 280.46061837 + 360.98564736629*(jde - 2451545.0) + 0.000387933*t*t - t*t*t/38710000.
         if (sidereal < 0.) sidereal = sidereal - 360.*(aint(sidereal/360.) - 1.)
         if (sidereal > 360.) sidereal = sidereal - 360.*aint(sidereal/360.)
         table(1, k) = ra
         table(2, k) = sidereal*pi/180.
         table(3, k) = sin(decl)
         table(4, k) = cos(decl)
         ! for the device: cos(hour angle) = cos((stG - ra) + lon) by the addition theorem
         table(5, k) = cos(sidereal*pi/180. - ra)
         table(6, k) = sin(sidereal*pi/180. - ra)
      end do
   end subroutine rs_sun_table

   !> sin/cos of the latitude and longitude in radians (src/SunPosition.f90:126-128,133).
   subroutine rs_point_geometry(n, localParam, sin_lat, cos_lat, lon_rad) bind(C, name='rs_point_geometry')
      integer(c_int), value :: n
      type(LocalParameters), intent(in) :: localParam(n)
      real(c_double), intent(out) :: sin_lat(n), cos_lat(n), lon_rad(n)
      real(8), parameter :: pi = 4*atan(1.0_8)
      real(8) :: latr
      integer :: p
      do p = 1, n
         latr = pi*localParam(p)%lat/180.
         sin_lat(p) = sin(latr)
         cos_lat(p) = cos(latr)
         lon_rad(p) = localParam(p)%lon*pi/180.
      end do
   end subroutine rs_point_geometry

end module RoadSurfHip

!> The entry points of the library: `runsimulation` (the reference's BIND(C) procedure, examples/example1/src/
!! Simulation.f90:4-6) and `runsimulation_batch[_ex]`.  A module of their own since round 5: a program unit
!! that DEFINES `runsimulation` itself - the reference's Simulation.f90 compiled against this library's
!! `module RoadSurf` (RoadSurfCompat.f90) - must not see a second definition of that global name through the
!! modules it uses.
module RoadSurfHipEntry
   use, intrinsic :: iso_c_binding
   use RoadSurfHip
   implicit none
   private

   interface
      !> C shim, roadsurf_amd/csrc/rs_host.hip
      function rs_host_run_batch(n, outPointers, inPointers, consts, localParam, tbottom, extras, device) &
         bind(C, name='rs_host_run_batch') result(rc)
         import :: c_int, c_double, OutputPointers, InputPointers, RsConstants, LocalParameters, RsHostExtras
         integer(c_int), value :: n
         type(OutputPointers), intent(inout) :: outPointers(*)
         type(InputPointers), intent(in) :: inPointers(*)
         type(RsConstants), intent(in) :: consts
         type(LocalParameters), intent(in) :: localParam(*)
         real(c_double), intent(in) :: tbottom(*)
         type(RsHostExtras), intent(in) :: extras
         integer(c_int), value :: device
         integer(c_int) :: rc
      end function rs_host_run_batch

      function rs_host_default_device() bind(C, name='rs_host_default_device') result(dev)
         import :: c_int
         integer(c_int) :: dev
      end function rs_host_default_device
   end interface

   interface
      !> roadsurf_amd/csrc/rs_coalesce.hip
      function rs_coalesce_run(outPointers, inPointers, inSettings, inputParam, localParam) &
         bind(C, name='rs_coalesce_run') result(rc)
         import :: c_int, OutputPointers, InputPointers, InputSettings, InputParameters, LocalParameters
         type(OutputPointers), intent(inout) :: outPointers
         type(InputPointers), intent(in) :: inPointers
         type(InputSettings), intent(in) :: inSettings
         type(InputParameters), intent(in) :: inputParam
         type(LocalParameters), intent(in) :: localParam
         integer(c_int) :: rc
      end function rs_coalesce_run
   end interface

   public :: runsimulation, runsimulation_batch, runsimulation_batch_ex, rs_runsimulation_gathered

contains

   subroutine fail(msg, status, code)
      character(len=*), intent(in) :: msg
      integer(c_int), intent(out) :: status
      integer, intent(in) :: code
      call rs_host_set_error(trim(msg)//c_null_char)
      status = code
   end subroutine fail

   !> n independent points, shared settings/parameters.  status = 0 on success.
   subroutine runsimulation_batch(n, outPointers, inPointers, inSettings, inputParam, localParam, status) &
      bind(C, name='runsimulation_batch')
      integer(c_int), value :: n
      type(OutputPointers), intent(inout) :: outPointers(n)
      type(InputPointers), intent(in) :: inPointers(n)
      type(InputSettings), intent(in) :: inSettings
      type(InputParameters), intent(in) :: inputParam
      type(LocalParameters), intent(in) :: localParam(n)
      integer(c_int), intent(out) :: status
      call runsimulation_batch_ex(n, outPointers, inPointers, inSettings, inputParam, localParam, status, &
                                  c_null_ptr)
   end subroutine runsimulation_batch

   !> The six calendar arrays of a point's time axis.
   subroutine axis_of(ip, nt, yy, mm, dd, hh, mi, ss)
      type(InputPointers), intent(in) :: ip
      integer, intent(in) :: nt
      integer(c_int), pointer, intent(out) :: yy(:), mm(:), dd(:), hh(:), mi(:), ss(:)
      call c_f_pointer(ip%c_year, yy, [nt]); call c_f_pointer(ip%c_month, mm, [nt])
      call c_f_pointer(ip%c_day, dd, [nt]); call c_f_pointer(ip%c_hour, hh, [nt])
      call c_f_pointer(ip%c_minute, mi, [nt]); call c_f_pointer(ip%c_second, ss, [nt])
   end subroutine axis_of

   !> Do two points carry the same time axis (same arrays, or equal values)?
   logical function same_axis(a, b, nt)
      type(InputPointers), intent(in) :: a, b
      integer, intent(in) :: nt
      integer(c_int), pointer :: y1(:), m1(:), d1(:), h1(:), i1(:), s1(:)
      integer(c_int), pointer :: y2(:), m2(:), d2(:), h2(:), i2(:), s2(:)
      same_axis = .true.
      if (c_associated(a%c_year, b%c_year) .and. c_associated(a%c_month, b%c_month) .and. &
          c_associated(a%c_day, b%c_day) .and. c_associated(a%c_hour, b%c_hour) .and. &
          c_associated(a%c_minute, b%c_minute) .and. c_associated(a%c_second, b%c_second)) return
      call axis_of(a, nt, y1, m1, d1, h1, i1, s1)
      call axis_of(b, nt, y2, m2, d2, h2, i2, s2)
      same_axis = all(y1 == y2) .and. all(m1 == m2) .and. all(d1 == d2) .and. all(h1 == h2) .and. &
                  all(i1 == i2) .and. all(s1 == s2)
   end function same_axis

   !> Same; first_failed (int32[n] or NULL) receives per point 0 or the 1-based time index at which
   !! CheckValues failed it.  The reference's in-place edits of the input arrays beyond VZ(1) (SW_dir
   !! clamp, src/InputOutput.f90:75-77; sky-view SW/SW_dir/LW, src/ModRadiation.f90:57-71) are written
   !! back to the caller with ROADSURF_HIP_WRITEBACK=1 only: for a batch they are three more arrays per
   !! point over PCIe that a batch caller rarely reads.  (The one-point entry `runsimulation` writes them
   !! back by default, as the reference does.)
   subroutine runsimulation_batch_ex(n, outPointers, inPointers, inSettings, inputParam, localParam, status, &
                                     first_failed_arg) bind(C, name='runsimulation_batch_ex')
      integer(c_int), value :: n
      type(OutputPointers), intent(inout) :: outPointers(n)
      type(InputPointers), intent(in) :: inPointers(n)
      type(InputSettings), intent(in) :: inSettings
      type(InputParameters), intent(in) :: inputParam
      type(LocalParameters), intent(in) :: localParam(n)
      integer(c_int), intent(out) :: status
      type(c_ptr), value :: first_failed_arg
      call rs_batch_core(n, outPointers, inPointers, inSettings, inputParam, localParam, status, &
                         first_failed_arg, 0_c_int)
   end subroutine runsimulation_batch_ex

   !> The points that concurrent callers of the one-point entry `runsimulation` brought (rs_coalesce.hip):
   !! a batch like any other, except that the reference's in-place input edits are written back by
   !! default - `runsimulation` is the reference's own entry and leaves the caller's arrays as the
   !! reference does (ROADSURF_HIP_WRITEBACK=0 opts out).
   subroutine rs_runsimulation_gathered(n, outPointers, inPointers, inSettings, inputParam, localParam, status) &
      bind(C, name='rs_runsimulation_gathered')
      integer(c_int), value :: n
      type(OutputPointers), intent(inout) :: outPointers(n)
      type(InputPointers), intent(in) :: inPointers(n)
      type(InputSettings), intent(in) :: inSettings
      type(InputParameters), intent(in) :: inputParam
      type(LocalParameters), intent(in) :: localParam(n)
      integer(c_int), intent(out) :: status
      call rs_batch_core(n, outPointers, inPointers, inSettings, inputParam, localParam, status, &
                         c_null_ptr, 1_c_int)
   end subroutine rs_runsimulation_gathered

   !> wb_default: whether the in-place input edits are written back when ROADSURF_HIP_WRITEBACK is unset.
   subroutine rs_batch_core(n, outPointers, inPointers, inSettings, inputParam, localParam, status, &
                            first_failed_arg, wb_default)
      integer(c_int), value :: n
      type(OutputPointers), intent(inout) :: outPointers(n)
      type(InputPointers), intent(in) :: inPointers(n)
      type(InputSettings), intent(in) :: inSettings
      type(InputParameters), intent(in) :: inputParam
      type(LocalParameters), intent(in) :: localParam(n)
      integer(c_int), intent(out) :: status
      type(c_ptr), value :: first_failed_arg
      integer(c_int), value :: wb_default
      type(c_ptr) :: first_failed
      integer(c_int), allocatable, target :: ffdiag(:)
      real(c_double), allocatable, target :: dgdiag(:, :), gdg(:, :)
      real(c_double), pointer :: dg_all(:, :)
      logical :: diag
      character(len=8) :: envv
      integer :: envl, envs

      type(RsConstants) :: consts
      type(RsHostExtras) :: extras
      real(c_double), allocatable, target :: tbottom(:), sun(:, :), slat(:), clat(:), lrad(:)
      integer(c_int), pointer :: yy(:), mm(:), dd(:), hh(:), mi(:), ss(:)
      logical :: any_sky
      integer :: k, nt, g, m, ngroups
      real(c_double), pointer :: vz(:)
      integer :: p
      integer(c_int) :: rc
      integer, allocatable :: group(:), rep(:), gidx(:)
      type(InputPointers), allocatable :: gi(:)
      type(OutputPointers), allocatable :: go(:)
      type(LocalParameters), allocatable :: gl(:)
      real(c_double), allocatable, target :: gtb(:), gslat(:), gclat(:), glrad(:)
      integer(c_int), allocatable, target :: gff(:)
      integer(c_int), pointer :: ff_all(:)

      status = 0
      any_sky = .false.
      ! ROADSURF_HIP_DIAGNOSTICS: print what the reference prints when CheckValues fails a point
      ! (src/InputOutput.f90:63-65,72,80-81) - needs the per-point failure index
      first_failed = first_failed_arg
      call get_environment_variable('ROADSURF_HIP_DIAGNOSTICS', envv, envl, envs)
      diag = (envs == 0 .and. envl > 0)
      if (diag .and. .not. c_associated(first_failed) .and. n >= 1) then
         allocate (ffdiag(n))
         ffdiag = 0
         first_failed = c_loc(ffdiag)
      end if
      call rs_host_set_error(c_null_char)   ! rs_last_error() is empty unless THIS call fails
      if (n < 1) return
      call rs_build_constants(inSettings, inputParam, consts, rc)
      if (rc /= 0) then
         call fail('runsimulation_batch: bad settings (NLayers in 5..32, SimLen >= 1, DTSecs > 0)', status, -1)
         return
      end if

      do p = 1, n
         if (localParam(p)%sky_view < 1.0 .and. localParam(p)%sky_view > -0.01) any_sky = .true.
         if (inPointers(p)%inputLen < inSettings%SimLen .or. outPointers(p)%outputLen < inSettings%SimLen) then
            call fail('runsimulation_batch: inputLen/outputLen shorter than SimLen', status, -4)
            return
         end if
      end do

      allocate (tbottom(n))
      do p = 1, n
         call c_f_pointer(inPointers(p)%c_year, yy, [1])
         call c_f_pointer(inPointers(p)%c_month, mm, [1])
         call c_f_pointer(inPointers(p)%c_day, dd, [1])
         tbottom(p) = rs_bottom_temperature(inputParam, consts, yy(1), mm(1), dd(1))
         ! the reference raises VZ(1) to 0.4 in the caller's array
         ! (src/Initialization.f90:121-123); keep that visible side effect
         call c_f_pointer(inPointers(p)%c_VZ, vz, [1])
         if (vz(1) < 0.4) vz(1) = 0.4
      end do

      extras%sun = c_null_ptr; extras%sin_lat = c_null_ptr; extras%cos_lat = c_null_ptr
      extras%lon_rad = c_null_ptr; extras%albedo_surroundings = inputParam%Albedo_surroundings
      extras%first_failed = first_failed
      extras%diagnostics = c_null_ptr
      if (diag) then
         ! ... and its other messages (boundary-layer loop, Coupling_control): rs_hip_diagnostics' record per point
         allocate (dgdiag(RS_DIAG_COLS, n))
         dgdiag = 0.0_c_double
         extras%diagnostics = c_loc(dgdiag)
      end if
      extras%writeback = wb_default
      call get_environment_variable('ROADSURF_HIP_WRITEBACK', envv, envl, envs)
      if (envs == 0 .and. envl >= 1) then
         extras%writeback = merge(0_c_int, 1_c_int, envv(1:1) == '0')
      end if
      if (any_sky) then
         ! The solar quantities that depend on time only (src/SunPosition.f90:196-260 and :70-125)
         ! are computed here, on the host, per DISTINCT time axis of the batch: the reference takes
         ! the Julian day from each point's own year(i)..second(i).  Points that share an axis - all
         ! of them in the reference driver's batches - form one group and one device call.
         nt = inSettings%SimLen
         allocate (group(n), rep(n))
         ngroups = 0
         do p = 1, n
            group(p) = 0
            do g = 1, ngroups
               if (same_axis(inPointers(p), inPointers(rep(g)), nt)) then
                  group(p) = g
                  exit
               end if
            end do
            if (group(p) == 0) then
               ngroups = ngroups + 1
               rep(ngroups) = p
               group(p) = ngroups
            end if
         end do
         allocate (sun(6, nt), slat(n), clat(n), lrad(n))
         call rs_point_geometry(n, localParam, slat, clat, lrad)
         if (ngroups == 1) then
            call axis_of(inPointers(1), nt, yy, mm, dd, hh, mi, ss)
            call rs_sun_table(int(nt, c_int), yy, mm, dd, hh, mi, ss, sun)
            extras%sun = c_loc(sun); extras%sin_lat = c_loc(slat); extras%cos_lat = c_loc(clat)
            extras%lon_rad = c_loc(lrad)
            rc = rs_host_run_batch(n, outPointers, inPointers, consts, localParam, tbottom, extras, &
                                   rs_host_default_device())
            if (rc /= 0) status = rc
         else
            if (c_associated(first_failed)) call c_f_pointer(first_failed, ff_all, [n])
            do g = 1, ngroups
               m = count(group == g)
               allocate (gi(m), go(m), gl(m), gtb(m), gslat(m), gclat(m), glrad(m), gff(m), gidx(m))
               k = 0
               do p = 1, n
                  if (group(p) /= g) cycle
                  k = k + 1
                  gidx(k) = p
                  gi(k) = inPointers(p); go(k) = outPointers(p); gl(k) = localParam(p)
                  gtb(k) = tbottom(p); gslat(k) = slat(p); gclat(k) = clat(p); glrad(k) = lrad(p)
               end do
               call axis_of(inPointers(rep(g)), nt, yy, mm, dd, hh, mi, ss)
               call rs_sun_table(int(nt, c_int), yy, mm, dd, hh, mi, ss, sun)
               extras%sun = c_loc(sun); extras%sin_lat = c_loc(gslat); extras%cos_lat = c_loc(gclat)
               extras%lon_rad = c_loc(glrad)
               extras%first_failed = c_null_ptr
               if (c_associated(first_failed)) extras%first_failed = c_loc(gff)
               extras%diagnostics = c_null_ptr
               if (diag) then
                  allocate (gdg(RS_DIAG_COLS, m))
                  gdg = 0.0_c_double
                  extras%diagnostics = c_loc(gdg)
               end if
               rc = rs_host_run_batch(int(m, c_int), go, gi, consts, gl, gtb, extras, rs_host_default_device())
               if (rc /= 0 .and. status == 0) status = rc
               if (c_associated(first_failed)) then
                  do k = 1, m
                     ff_all(gidx(k)) = gff(k)
                  end do
               end if
               if (diag) then
                  do k = 1, m
                     dgdiag(:, gidx(k)) = gdg(:, k)
                  end do
                  deallocate (gdg)
               end if
               deallocate (gi, go, gl, gtb, gslat, gclat, glrad, gff, gidx)
            end do
         end if
         deallocate (group, rep)
      else
         rc = rs_host_run_batch(n, outPointers, inPointers, consts, localParam, tbottom, extras, &
                                rs_host_default_device())
         if (rc /= 0) status = rc
      end if
      deallocate (tbottom)
      if (diag .and. status == 0) then
         call print_diagnostics(n, outPointers, inPointers, localParam, first_failed)
         call print_loop_diagnostics(n, dgdiag)
      end if
   end subroutine rs_batch_core

   !> The reference's messages from CalcBLCondAndLE and Coupling_control (src/BoundaryLayer.f90:69-74,98-101,
   !! src/Coupling.f90:400-401,451-452) from rs_hip_diagnostics' record: the text and format of the reference for
   !! the FIRST occurrence at a point; the reference repeats them at every pass / time index they occur at, here
   !! a line with the point, the time index and the count follows instead.
   subroutine print_loop_diagnostics(n, dg)
      integer(c_int), intent(in) :: n
      real(c_double), intent(in) :: dg(RS_DIAG_COLS, n)
      integer :: p, msg
      do p = 1, n
         if (dg(6, p) > 0.0) then
            Write (*, *) ' ERROR : UStar negative,vz ', dg(9, p)
            write (*, *) dg(8, p), dg(9, p), dg(10, p), dg(11, p), dg(12, p)
            write (*, '(a,i0,a,i0,a,i0,a)') ' (roadsurf_hip: point ', p, ', first at time index ', &
               int(dg(7, p)), '; passes with this message: ', int(dg(6, p)), ')'
         end if
         if (dg(1, p) > 0.0) then
            Write (*, "(' Max number of BLCond iterations (MaxIter,BLCond_Old,BLCond) :',I5,2F10.5)") &
               int(dg(3, p)), dg(4, p), dg(5, p)
            write (*, '(a,i0,a,i0,a,i0,a)') ' (roadsurf_hip: point ', p, ', first at time index ', &
               int(dg(2, p)), '; time indices with this message: ', int(dg(1, p)), ')'
         end if
         msg = int(dg(13, p))
         if (iand(msg, 8) /= 0) then
            write (*, *) "coupling coefficient too small, coupling failed"
            write (*, '(a,i0,a)') ' (roadsurf_hip: point ', p, ')'
         end if
         if (iand(msg, 16) /= 0) then
            write (*, *) "coupling coefficient too big, coupling failed"
            write (*, '(a,i0,a)') ' (roadsurf_hip: point ', p, ')'
         end if
      end do
      flush (6)
   end subroutine print_loop_diagnostics

   !> The reference's diagnostics for the points CheckValues failed (src/InputOutput.f90:55-82): the
   !! same three messages on standard output, from the caller's arrays at the failing index.  The
   !! surface temperature CheckValues saw at index i is the one SaveOutput wrote for index i-1.
   subroutine print_diagnostics(n, outPointers, inPointers, localParam, first_failed)
      integer(c_int), intent(in) :: n
      type(OutputPointers), intent(in) :: outPointers(n)
      type(InputPointers), intent(in) :: inPointers(n)
      type(LocalParameters), intent(in) :: localParam(n)
      type(c_ptr), intent(in) :: first_failed
      integer(c_int), pointer :: ff(:)
      real(c_double), pointer :: tair(:), tdew(:), rhz(:), vz(:), sw(:), lw(:), prec(:), swd(:), lwn(:), ts(:)
      integer :: p, i, nt
      if (.not. c_associated(first_failed)) return
      call c_f_pointer(first_failed, ff, [n])
      do p = 1, n
         i = ff(p)
         if (i < 1) cycle
         nt = inPointers(p)%inputLen
         if (i > nt) cycle
         call c_f_pointer(inPointers(p)%c_tair, tair, [nt]); call c_f_pointer(inPointers(p)%c_tdew, tdew, [nt])
         call c_f_pointer(inPointers(p)%c_Rhz, rhz, [nt]); call c_f_pointer(inPointers(p)%c_VZ, vz, [nt])
         call c_f_pointer(inPointers(p)%c_SW, sw, [nt]); call c_f_pointer(inPointers(p)%c_LW, lw, [nt])
         call c_f_pointer(inPointers(p)%c_prec, prec, [nt])
         if (tair(i) < -90.0 .or. tair(i) > 100.0 .or. tdew(i) < -90 .or. tdew(i) > 100.0 &
             .or. rhz(i) < -0.1 .or. rhz(i) > 120.0 .or. vz(i) < -1.0 .or. vz(i) > 100.0 &
             .or. sw(i) < -0.1 .or. sw(i) > 4000.0 .or. lw(i) < -0.1 .or. lw(i) > 1000.0 &
             .or. prec(i) < -0.1 .or. prec(i) > 500.0) then
            write (*, *) "BAD input value! ", tair(i), tdew(i), rhz(i), vz(i), sw(i), lw(i), prec(i)
         end if
         if (localParam(p)%sky_view < 1.0 .and. localParam(p)%sky_view > -0.01) then
            call c_f_pointer(inPointers(p)%c_SW_dir, swd, [nt]); call c_f_pointer(inPointers(p)%c_LW_net, lwn, [nt])
            if (swd(i) < -0.1 .or. swd(i) > 4000.0 .or. lwn(i) < -1000.0 .or. lwn(i) > 1000.0) then
               write (*, *) "BAD input value: SW_dir,LW_net", swd(i), lwn(i)
            end if
         end if
         if (i > 1) then
            call c_f_pointer(outPointers(p)%c_TsurfOut, ts, [outPointers(p)%outputLen])
            if (ts(i - 1) < -100.0 .or. ts(i - 1) > 100.0) then
               write (*, *) "Abnormal surface temperature", ts(i - 1), i, localParam(p)%lat, localParam(p)%lon
            end if
         end if
      end do
      flush (6)
   end subroutine print_diagnostics

   !> Drop-in for the reference's runsimulation (examples/example1/src/Simulation.f90:4-117).
   !! No status argument, as in the reference: on a device/runtime error the
   !! outputs are left at -9999.0 and the message goes to standard error.
   subroutine runsimulation(outPointers, inPointers, inSettings, inputParam, localParam) &
      bind(C, name='runsimulation')
      type(OutputPointers), intent(inout) :: outPointers
      type(InputPointers), intent(in) :: inPointers
      type(InputSettings), intent(in) :: inSettings
      type(InputParameters), intent(in) :: inputParam
      type(LocalParameters), intent(in) :: localParam
      integer(c_int) :: status
      real(c_double), pointer :: arr(:)
      integer :: k
      ! one point - gathered with the points other threads are calling for at this moment when
      ! ROADSURF_HIP_COALESCE_US > 0 (roadsurf_amd/csrc/rs_coalesce.hip), else a batch of one
      status = rs_coalesce_run(outPointers, inPointers, inSettings, inputParam, localParam)
      if (status /= 0) then
         write (0, *) 'runsimulation (HIP): failed with status ', status
         ! src/Initialization.f90:397-412: unwritten outputs read -9999.0
         call c_f_pointer(outPointers%c_TsurfOut, arr, [outPointers%outputLen]); arr = -9999.0
         call c_f_pointer(outPointers%c_SnowOut, arr, [outPointers%outputLen]); arr = -9999.0
         call c_f_pointer(outPointers%c_WaterOut, arr, [outPointers%outputLen]); arr = -9999.0
         call c_f_pointer(outPointers%c_IceOut, arr, [outPointers%outputLen]); arr = -9999.0
         call c_f_pointer(outPointers%c_DepositOut, arr, [outPointers%outputLen]); arr = -9999.0
         call c_f_pointer(outPointers%c_Ice2Out, arr, [outPointers%outputLen]); arr = -9999.0
      end if
      k = 0
   end subroutine runsimulation

end module RoadSurfHipEntry
