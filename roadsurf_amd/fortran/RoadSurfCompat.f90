!> The reference's Fortran module surface over this library: `module RoadSurfVariables` with its sixteen
!! derived types and `module RoadSurf` with its fourteen public procedures
!! (/root/reference/src/RoadSurfVariables.f90:13-28, src/*.f90.inc; src/RoadSurf.f90:6-270), plus the
!! external `lastValues` the reference's own time loop calls (src/InputOutput.f90:169-198) - enough for
!! /root/reference/examples/example1/src/Simulation.f90 to compile and link UNCHANGED against
!! libroadsurf_hip.so and to produce the bits of `runsimulation` (tests/test_abi_layout.py compiles it,
!! tests/test_hip_module_surface.py runs it).
!!
!! How the procedures map onto the device (roadsurf_amd/csrc/rs_compat.hip).  Behind this library one time
!! index of a point is ONE fused step kernel - CheckValues, SetCurrentValues, relaxation, precipitation,
!! sky view, heat balance, wear, RoadCond and albedo in one pass (rs_physics_body.inc) - so the fourteen
!! procedures cannot each be a piece of it.  Instead:
!!   * `Initialization` creates the point's device context (a one-point plan, cached per thread) and runs the
!!     device part of the initialization; `BalanceModelOneStep` IS the step: it uploads the caller's forcing
!!     of that index (whatever the caller has made of it), runs the fused kernel for one index and loads the
!!     point's state back into the derived types, as it stands at the END of the index;
!!   * what the reference does before it inside the same index (`SetCurrentValues`, `RelaxationOperations`,
!!     `PrecipitationToStorage`, `ModRadiationBySurroundings`) and after it (`WearFactors`, `RoadCond`,
!!     `CalcAlbedo`) is therefore already inside that step: those procedures keep the host-visible members
!!     they own up to date where that is a plain assignment and are otherwise no-ops;
!!   * `CheckValues` is the reference's test, on the host (it decides the caller's loop condition);
!!     `SaveOutput` is the reference's six assignments; `CheckEndCoupling` at the end of the coupling window
!!     runs EVERY replay of the window on the device at once and rewrites the caller's output rows - the
!!     caller's index never goes back (`CouplingOperations1` leaves it alone), the outputs are the ones the
!!     reference has after its last replay.
!! State visible to the caller between two procedures of one index is therefore not the reference's
!! intermediate state; at index boundaries the members listed as maintained below are exact.  The price:
!! a kernel launch, an upload and a state download per index (INTEGRATION.md section 2) - this is the path
!! for callers that own the time loop, not the fast one (`runsimulation_batch`, `rs_driver_run`).
module RoadSurfVariables
   use, intrinsic :: iso_c_binding
   use RoadSurfHip, only: InputPointers, OutputPointers, InputSettings, InputParameters, LocalParameters
   implicit none
   public
   private :: rs_surface_release, rs_compat_end

   !> the caller's series, aliased (src/InputArrays.f90.inc)
   type :: InputArrays
      ! (no default initialisation: the reference passes the connected arrays through the INTENT(OUT)
      ! dummies of Initialization and relies on their association surviving it)
      integer(c_int), pointer :: timeForFortran(:)
      real(c_double), pointer :: Tair(:), Tdew(:), VZ(:), Rhz(:), prec(:), SW(:), LW(:)
      real(c_double), pointer :: SW_dir(:), LW_net(:), TSurfObs(:)
      integer(c_int), pointer :: PrecPhase(:)
      real(c_double), pointer :: local_horizons(:), depth(:)
      integer(c_int), pointer :: year(:), month(:), day(:), hour(:), minute(:), second(:)
   end type InputArrays

   !> the caller's output arrays, aliased (src/OutputArrays.f90.inc)
   type :: OutputArrays
      real(c_double), pointer :: TsurfOut(:), SnowOut(:), WaterOut(:), IceOut(:), DepositOut(:), Ice2Out(:)
   end type OutputArrays

   !> src/PhysicalParameters.f90.inc.  Maintained: everything Initialization copies from InputParameters
   !! and the four logarithms; the heat-capacity fit coefficients (Afc1 ... Efc2) are not used on this path.
   type :: PhysicalParameters
      real(8) :: VK_Const, SB_const, ZRefW, ZRefT, ZeroDisp, ZMom, ZHeat
      real(8) :: logMom, logHeat, logCond, logUstar, Grav, Emiss
      real(8) :: Afc1 = 0, Bfc1 = 0, Cfc1 = 0, Dfc1 = 0, Efc1 = 0, Afc2 = 0, Bfc2 = 0, Cfc2 = 0, Dfc2 = 0, Efc2 = 0
      real(8) :: Poro1, Poro2, vsh1, vsh2, LVap, LFus, TClimG, MaxPormms, DampDpth, Omega, AZ
      real(8) :: Silt1, Silt2, RhoB1, RhoB2
   end type PhysicalParameters

   !> src/GroundVariables.f90.inc.  Maintained exactly at index boundaries: Albedo, Tmp(0:N+1), TmpNw(0:N+1);
   !! set once: ZDpth, DyC, DyK, condDZ, Wcont.  VSH, HS, capDZ, CC, GCond, HStor, GroundFlux keep the
   !! reference's initial fill: they are per-step temporaries that never leave the device.
   type :: GroundVariables
      real(8) :: Albedo, HStor
      real(8), allocatable :: condDZ(:), capDZ(:), Wcont(:), VSH(:), HS(:), CC(:)
      real(8), allocatable :: Tmp(:), TmpNw(:), DyC(:), DyK(:), ZDpth(:), GCond(:)
      real(8) :: GroundFlux
   end type GroundVariables

   !> src/SurfaceVariables.f90.inc.  Maintained exactly at index boundaries: TsurfAve, the five storages,
   !! Q2Melt, T4Melt, VeryCold; TsurfOBS by SetCurrentValues.  Carries the point's device context: released
   !! when the variable goes out of scope (or is the INTENT(OUT) argument of the next Initialization).
   type :: SurfaceVariables
      real(8) :: TsurfAve, SrfWatmms, SrfSnowmms, SrfIcemms, SrfIce2mms, SrfDepmms
      real(8) :: Q2Melt, T4Melt, TrfFric, EvapmmTS
      logical :: VeryCold, WearSurf
      real(8) :: TsurfOBS
      type(c_ptr), private :: rs_ctx = c_null_ptr
   contains
      final :: rs_surface_release
   end type SurfaceVariables

   !> src/AtmVariables.f90.inc.  Maintained: what SetCurrentValues / lastValues / RelaxationOperations assign
   !! (Tair, Tdew, VZ, RHz, PrecInTStep, the relaxation targets and anchors); the fluxes of a step (BLCond,
   !! RNet, LE_Flux, ...) never leave the device and keep the reference's initial fill.
   type :: AtmVariables
      real(8) :: Tair, VZ, Tdew, RHz, PrecInTStep, TairInitEnd, VZInitEnd, RhzInitEnd
      real(8) :: BLCond, RNet, LE_Flux, RainIntensity, SnowIntensity, TairR, VZR, RhzR, CalmLim
      real(8) :: SensibleHeatFlux, RainmmTS, SnowmmTS
      integer :: SnowType, PrecType
   end type AtmVariables

   !> src/CouplingVariables.f90.inc.  Maintained at index boundaries (from the device state): the iteration
   !! scalars (Coupling_iterations ... Tsurf_end_coup1, lastTsurfObs, the flags) and the first coupling window
   !! (couplingStartI/EndI(1)); the saved state of a window lives on the device.
   type :: CouplingVariables
      real(8), allocatable :: TmpSave(:)
      integer :: Coupling_iterations
      real(8) :: TsurfNearestAbove, TsurfNearestBelow, RadCoeff
      logical :: Down, start_coupling_again, Coupling_failed, inCouplingPhase, VeryColdSave
      real(8) :: RadCoefNearestAbove, RadCoefNearestBelow, RadCoeffPrevious
      real(8) :: SWRadCof, LWRadCof, SW_correction, LW_correction, Tsurf_end_coup1
      integer, dimension(48) :: couplingStartI, couplingEndI, couplingStartJ, couplingEndJ
      real(8) :: TSurfAveSave, SrfWatmmsSave, SrfIcemmsSave, SrfIce2mmsSave, SrfDepmmsSave, SrfSnowmmsSave
      real(8) :: AlbedoSave, lastTsurfObs
      integer :: saveDatai, saveDataj
      real(8), dimension(:), allocatable :: SWSave, SWDirSave, LWSave
      integer :: NObs, CoupPhaseN
      integer, dimension(48, 6) :: carObsTime
      integer, dimension(48) :: obsI
      real(8), dimension(48) :: obsTsurf
   end type CouplingVariables

   !> src/ModelSettings.f90.inc
   type :: ModelSettings
      integer :: InitLenI, SimLen
      logical :: use_coupling, use_relaxation, force_tsurf
      integer :: NLayers
      real(8) :: DTSecs, tsurfOutputDepth
      logical :: simulation_failed
      real(8) :: Tph, NightOn, NightOff, CalmLimDay, CalmLimNgt, TrfFricNgt, TrFfricDay
      integer :: coupling_minutes
      real(8) :: couplingEffectReduction
      integer :: outputStep
   end type ModelSettings

   !> src/InputRadiationCoefficient.f90.inc (not used by the model)
   type :: InputRadiationCoefficient
      real :: inputRadCofSW, inputRadCofLW
      integer :: inputRadCofI
      integer, dimension(6) :: timeToUseInputRadCof
   end type InputRadiationCoefficient

   !> src/RoadCondParameters.f90.inc: what Initialization copies from InputParameters
   type :: RoadCondParameters
      real(8) :: WDampLim, WWetLim, WWearLim, Snow2IceFac, SnowIceRat = 0, MissValI, MissValR
      real(8) :: MinPrecmm, MinWatmms, MinSnowmms, MinDepmms, MinIcemms
      real(8) :: MaxSnowmms, MaxDepmms, MaxIcemms, MaxExtmms, MaxWatmms, AlbDry, AlbSnow
      real(8) :: WatDens, SnowDens, IceDens, DepDens, WatMHeat, PorEvaF, DampWearF
      real(8) :: TLimFreeze, TLimMeltSnow, TLimMeltIce, TLimMeltDep, TLimDew, TLimColdH, TLimColdL
      real(8) :: WetSnowFormR, WetSnowMeltR, PLimSnow, PLimRain
      logical :: WetSnowFrozen
      real(8) :: freezing_limit_normal, snow_melting_limit_normal, ice_melting_limit_normal
      real(8) :: frost_melting_limit_normal, frost_formation_limit_normal, T4Melt_normal
      logical :: forceIceMelting, forceSnowMelting, CanMeltingChangeTemperature
   end type RoadCondParameters

   !> src/WearingFactors.f90.inc
   type :: WearingFactors
      real(8) :: SnowTran, SnowTran2, SnowTranDef, DepWear, IceWear, IceWear2
      real(8) :: IceWearNight, IceWear2Night, IceWearSW, IceWear2SW, WatWear
   end type WearingFactors

   interface
      subroutine rs_compat_end(ctx) bind(C, name='rs_compat_end')
         import :: c_ptr
         type(c_ptr), value :: ctx
      end subroutine rs_compat_end
   end interface

contains

   subroutine rs_surface_release(surf)
      type(SurfaceVariables), intent(inout) :: surf
      if (c_associated(surf%rs_ctx)) call rs_compat_end(surf%rs_ctx)
      surf%rs_ctx = c_null_ptr
   end subroutine rs_surface_release

   !> the point's device context (module RoadSurf's procedures; not part of the reference's surface)
   function rs_surface_context(surf) result(ctx)
      type(SurfaceVariables), intent(in) :: surf
      type(c_ptr) :: ctx
      ctx = surf%rs_ctx
   end function rs_surface_context

   subroutine rs_surface_set_context(surf, ctx)
      type(SurfaceVariables), intent(inout) :: surf
      type(c_ptr), intent(in) :: ctx
      if (c_associated(surf%rs_ctx)) call rs_compat_end(surf%rs_ctx)
      surf%rs_ctx = ctx
   end subroutine rs_surface_set_context

end module RoadSurfVariables

module RoadSurf
   use, intrinsic :: iso_c_binding
   implicit none
   private

   public :: ConnectFortran2Carrays, Initialization, CheckValues, CouplingOperations1
   public :: RelaxationOperations, SetCurrentValues, BalanceModelOneStep, SaveOutput
   public :: CheckEndCoupling, PrecipitationToStorage, ModRadiationBySurroundings
   public :: WearFactors, RoadCond, CalcAlbedo
   !> not in the reference: the many-point entries of this library under the same module
   public :: runsimulation_batch, runsimulation_batch_ex

   interface
      subroutine runsimulation_batch(n, outPointers, inPointers, inSettings, inputParam, localParam, status) &
         bind(C, name='runsimulation_batch')
         use RoadSurfVariables
         integer(c_int), value :: n
         type(OutputPointers), intent(inout) :: outPointers(*)
         type(InputPointers), intent(in) :: inPointers(*)
         type(InputSettings), intent(in) :: inSettings
         type(InputParameters), intent(in) :: inputParam
         type(LocalParameters), intent(in) :: localParam(*)
         integer(c_int), intent(out) :: status
      end subroutine runsimulation_batch
      subroutine runsimulation_batch_ex(n, outPointers, inPointers, inSettings, inputParam, localParam, status, &
                                        first_failed) bind(C, name='runsimulation_batch_ex')
         use RoadSurfVariables
         integer(c_int), value :: n
         type(OutputPointers), intent(inout) :: outPointers(*)
         type(InputPointers), intent(in) :: inPointers(*)
         type(InputSettings), intent(in) :: inSettings
         type(InputParameters), intent(in) :: inputParam
         type(LocalParameters), intent(in) :: localParam(*)
         integer(c_int), intent(out) :: status
         type(c_ptr), value :: first_failed
      end subroutine runsimulation_batch_ex
   end interface

   ! ---- roadsurf_amd/csrc/rs_compat.hip (include/roadsurf.h, layer 5) ----
   integer, parameter :: NSTATE = 134, MAXL = 32
   ! slots of the state column, 0-based as in roadsurf_amd/csrc/rs_state.h (rs_compat.hip asserts them)
   integer, parameter :: S_TNW1 = 32, S_TNW2 = 33, S_TSURF = 34, S_WAT = 35, S_SNOW = 36, S_ICE = 37
   integer, parameter :: S_ICE2 = 38, S_DEP = 39, S_Q2MELT = 40, S_T4MELT = 41, S_ALBEDO = 42
   integer, parameter :: S_VERYCOLD = 43, S_TAIR_END = 45, S_VZ_END = 46, S_RH_END = 47
   integer, parameter :: S_CPL_ITER = 49, S_CPL_FLAGS = 50, S_CPL_TABOVE = 51, S_CPL_TBELOW = 52
   integer, parameter :: S_CPL_RADCOEFF = 53, S_CPL_RCABOVE = 54, S_CPL_RCBELOW = 55, S_CPL_RCPREV = 56
   integer, parameter :: S_CPL_SWCOF = 57, S_CPL_LWCOF = 58, S_CPL_SWCORR = 59, S_CPL_LWCORR = 60
   integer, parameter :: S_CPL_TEND1 = 61, S_CPL_LASTOBS = 62

   type, bind(C) :: RsCompatArrays
      type(c_ptr) :: tair, tdew, vz, rhz, prec, sw, lw, sw_dir, lw_net, tsurfobs, depth
      type(c_ptr) :: precphase, hour
      type(c_ptr) :: horizons
      type(c_ptr) :: out(6)
   end type RsCompatArrays

   interface
      function rs_compat_begin(consts, local, tbottom, arrays, sun, geo, albedo_surroundings, state_out) &
         bind(C, name='rs_compat_begin') result(ctx)
         use RoadSurfHip, only: RsConstants, LocalParameters
         import :: c_ptr, c_double, RsCompatArrays
         type(RsConstants), intent(in) :: consts
         type(LocalParameters), intent(in) :: local
         real(c_double), value :: tbottom
         type(RsCompatArrays), intent(in) :: arrays
         type(c_ptr), value :: sun, geo
         real(c_double), value :: albedo_surroundings
         real(c_double), intent(out) :: state_out(*)
         type(c_ptr) :: ctx
      end function rs_compat_begin
      function rs_compat_step(ctx, i, state_out, edits) bind(C, name='rs_compat_step') result(rc)
         import :: c_ptr, c_int, c_double
         type(c_ptr), value :: ctx
         integer(c_int), value :: i
         real(c_double), intent(out) :: state_out(*)
         real(c_double), intent(inout) :: edits(3)
         integer(c_int) :: rc
      end function rs_compat_step
      function rs_compat_replay(ctx, i, state_out, rewritten) bind(C, name='rs_compat_replay') result(rc)
         import :: c_ptr, c_int, c_double
         type(c_ptr), value :: ctx
         integer(c_int), value :: i
         real(c_double), intent(out) :: state_out(*)
         integer(c_int), intent(out) :: rewritten(2)
         integer(c_int) :: rc
      end function rs_compat_replay
      function rs_compat_failed_index(ctx) bind(C, name='rs_compat_failed_index') result(idx)
         import :: c_ptr, c_int
         type(c_ptr), value :: ctx
         integer(c_int) :: idx
      end function rs_compat_failed_index
      function rs_compat_last_state(ctx, state_out) bind(C, name='rs_compat_last_state') result(rc)
         import :: c_ptr, c_int, c_double
         type(c_ptr), value :: ctx
         real(c_double), intent(out) :: state_out(*)
         integer(c_int) :: rc
      end function rs_compat_last_state
      function rs_compat_outputs(ctx, i, out6) bind(C, name='rs_compat_outputs') result(rc)
         import :: c_ptr, c_int, c_double
         type(c_ptr), value :: ctx
         integer(c_int), value :: i
         real(c_double), intent(out) :: out6(6)
         integer(c_int) :: rc
      end function rs_compat_outputs
      function rs_last_error() bind(C, name='rs_last_error') result(msg)
         import :: c_ptr
         type(c_ptr) :: msg
      end function rs_last_error
   end interface

contains

   !> src/ConnectFortran2Carrays.f90:9-21,38-84: the caller's C arrays under Fortran names
   subroutine ConnectFortran2Carrays(inPointers, modelInput, outPointers, modelOutput)
      use RoadSurfVariables
      type(InputPointers), intent(IN) :: inPointers
      type(OutputPointers), intent(INOUT) :: outPointers
      type(InputArrays), intent(OUT) :: modelInput
      type(OutputArrays), intent(OUT) :: modelOutput
      integer :: n
      n = inPointers%inputLen
      call c_f_pointer(inPointers%c_tair, modelInput%Tair, [n])
      call c_f_pointer(inPointers%c_tdew, modelInput%Tdew, [n])
      call c_f_pointer(inPointers%c_VZ, modelInput%VZ, [n])
      call c_f_pointer(inPointers%c_Rhz, modelInput%Rhz, [n])
      call c_f_pointer(inPointers%c_prec, modelInput%prec, [n])
      call c_f_pointer(inPointers%c_SW, modelInput%SW, [n])
      call c_f_pointer(inPointers%c_LW, modelInput%LW, [n])
      call c_f_pointer(inPointers%c_SW_dir, modelInput%SW_dir, [n])
      call c_f_pointer(inPointers%c_LW_net, modelInput%LW_net, [n])
      call c_f_pointer(inPointers%c_TSurfObs, modelInput%TSurfObs, [n])
      call c_f_pointer(inPointers%c_PrecPhase, modelInput%PrecPhase, [n])
      call c_f_pointer(inPointers%c_local_horizons, modelInput%local_horizons, [360])
      call c_f_pointer(inPointers%c_Depth, modelInput%depth, [n])
      call c_f_pointer(inPointers%c_year, modelInput%year, [n])
      call c_f_pointer(inPointers%c_month, modelInput%month, [n])
      call c_f_pointer(inPointers%c_day, modelInput%day, [n])
      call c_f_pointer(inPointers%c_hour, modelInput%hour, [n])
      call c_f_pointer(inPointers%c_minute, modelInput%minute, [n])
      call c_f_pointer(inPointers%c_second, modelInput%second, [n])
      n = outPointers%outputLen
      call c_f_pointer(outPointers%c_TsurfOut, modelOutput%TsurfOut, [n])
      call c_f_pointer(outPointers%c_SnowOut, modelOutput%SnowOut, [n])
      call c_f_pointer(outPointers%c_WaterOut, modelOutput%WaterOut, [n])
      call c_f_pointer(outPointers%c_IceOut, modelOutput%IceOut, [n])
      call c_f_pointer(outPointers%c_DepositOut, modelOutput%DepositOut, [n])
      call c_f_pointer(outPointers%c_Ice2Out, modelOutput%Ice2Out, [n])
   end subroutine ConnectFortran2Carrays

   !> the derived types from the point's state column (index boundaries)
   subroutine load_state(st, nl, surf, ground)
      use RoadSurfVariables
      real(c_double), intent(in) :: st(0:NSTATE - 1)
      integer, intent(in) :: nl
      type(SurfaceVariables), intent(inout) :: surf
      type(GroundVariables), intent(inout) :: ground
      integer :: j
      do j = 1, nl
         ground%Tmp(j) = st(j - 1)
         ground%TmpNw(j) = st(j - 1)
      end do
      ground%TmpNw(1) = st(S_TNW1)
      ground%TmpNw(2) = st(S_TNW2)
      ground%Albedo = st(S_ALBEDO)
      surf%TsurfAve = st(S_TSURF)
      surf%SrfWatmms = st(S_WAT)
      surf%SrfSnowmms = st(S_SNOW)
      surf%SrfIcemms = st(S_ICE)
      surf%SrfIce2mms = st(S_ICE2)
      surf%SrfDepmms = st(S_DEP)
      surf%Q2Melt = st(S_Q2MELT)
      surf%T4Melt = st(S_T4MELT)
      surf%VeryCold = st(S_VERYCOLD) /= 0.0d0
   end subroutine load_state

   subroutine load_coupling(st, coupling)
      use RoadSurfVariables
      real(c_double), intent(in) :: st(0:NSTATE - 1)
      type(CouplingVariables), intent(inout) :: coupling
      integer :: fl
      coupling%Coupling_iterations = int(st(S_CPL_ITER))
      fl = int(st(S_CPL_FLAGS))
      coupling%start_coupling_again = iand(fl, 1) /= 0
      coupling%Coupling_failed = iand(fl, 2) /= 0
      coupling%VeryColdSave = iand(fl, 4) /= 0
      coupling%TsurfNearestAbove = st(S_CPL_TABOVE)
      coupling%TsurfNearestBelow = st(S_CPL_TBELOW)
      coupling%RadCoeff = st(S_CPL_RADCOEFF)
      coupling%RadCoefNearestAbove = st(S_CPL_RCABOVE)
      coupling%RadCoefNearestBelow = st(S_CPL_RCBELOW)
      coupling%RadCoeffPrevious = st(S_CPL_RCPREV)
      coupling%SWRadCof = st(S_CPL_SWCOF)
      coupling%LWRadCof = st(S_CPL_LWCOF)
      coupling%SW_correction = st(S_CPL_SWCORR)
      coupling%LW_correction = st(S_CPL_LWCORR)
      coupling%Tsurf_end_coup1 = st(S_CPL_TEND1)
      coupling%lastTsurfObs = st(S_CPL_LASTOBS)
   end subroutine load_coupling

   subroutine stop_with_library_error(who)
      character(*), intent(in) :: who
      character(kind=c_char), pointer :: msg(:)
      type(c_ptr) :: p
      integer :: n
      p = rs_last_error()
      n = 0
      if (c_associated(p)) then
         call c_f_pointer(p, msg, [512])
         do while (n < 512)
            if (msg(n + 1) == c_null_char) exit
            n = n + 1
         end do
         write (*, *) who, ': ', msg(1:n)
      else
         write (*, *) who, ': the device step failed'
      end if
      error stop 'module RoadSurf over libroadsurf_hip: no CPU path to fall back to'
   end subroutine stop_with_library_error

   !> src/Initialization.f90:6-63 (initSettings, initOutputArrays, setInputParam,
   !! initVariablesAndParameters): the settings and parameter copies here on the host, the tables from
   !! rs_build_constants (the same expressions, RoadSurfHip), the profile and storages on the device.
   subroutine Initialization(modelInput, inSettings, settings, modelOutput, atm, surf, inputParam, localParam, &
                             coupling, phy, ground, condParam)
      use RoadSurfVariables
      use RoadSurfHip, only: RsConstants, rs_build_constants, rs_bottom_temperature, rs_sun_table, &
                             rs_point_geometry
      type(InputSettings), intent(IN) :: inSettings
      type(InputParameters), intent(IN) :: inputParam
      type(LocalParameters), intent(IN) :: localParam
      type(InputArrays), intent(OUT) :: modelInput
      type(OutputArrays), intent(OUT) :: modelOutput
      type(AtmVariables), intent(OUT) :: atm
      type(CouplingVariables), intent(OUT) :: coupling
      type(ModelSettings), intent(OUT) :: settings
      type(PhysicalParameters), intent(OUT) :: phy
      type(GroundVariables), intent(OUT) :: ground
      type(SurfaceVariables), intent(OUT) :: surf
      type(RoadCondParameters), intent(OUT) :: condParam
      type(RsConstants) :: consts
      type(RsCompatArrays) :: arr
      type(LocalParameters), target :: lp(1)
      integer(c_int) :: rc
      integer :: nl, j, n
      real(c_double) :: tbottom, st(0:NSTATE - 1)
      real(c_double), allocatable, target :: sun(:, :)
      real(c_double), target :: geo(3)
      type(c_ptr) :: ctx, psun, pgeo
      logical :: sky

      ! NOTE: modelInput / modelOutput are INTENT(OUT) in the reference too, although the caller has just
      ! connected them (ConnectFortran2Carrays): pointer components keep their association, as there.
      ! ---- settings (initSettings, :289-317) ----
      settings%SimLen = inSettings%SimLen
      settings%InitLenI = localParam%InitLenI
      settings%DTSecs = inSettings%DTSecs
      settings%tsurfOutputDepth = inSettings%tsurfOutputDepth
      settings%NLayers = inSettings%NLayers
      settings%NightOn = inputParam%NightOn
      settings%NightOff = inputParam%NightOff
      settings%CalmLimDay = inputParam%CalmLimDay
      settings%CalmLimNgt = inputParam%CalmLimNgt
      settings%TrfFricNgt = inputParam%TrfFricNgt
      settings%TrFfricDay = inputParam%TrFfricDay
      settings%use_coupling = inSettings%use_coupling == 1
      settings%use_relaxation = inSettings%use_relaxation == 1
      settings%force_tsurf = inSettings%force_tsurf == 1
      settings%coupling_minutes = inSettings%coupling_minutes
      settings%couplingEffectReduction = inSettings%couplingEffectReduction
      settings%outputStep = inSettings%outputStep
      settings%simulation_failed = .false.
      settings%Tph = settings%DTSecs/3600.0
      nl = settings%NLayers
      n = settings%SimLen

      call rs_build_constants(inSettings, inputParam, consts, rc)
      if (rc /= 0) then
         write (*, *) 'Initialization: settings outside what this library runs (NLayers in 5..32, SimLen >= 1, DTSecs > 0)'
         error stop 'module RoadSurf over libroadsurf_hip'
      end if

      ! ---- output arrays (initOutputArrays, :249-263) ----
      modelOutput%SnowOut(1:n) = -9999.0
      modelOutput%WaterOut(1:n) = -9999.0
      modelOutput%IceOut(1:n) = -9999.0
      modelOutput%Ice2Out(1:n) = -9999.0
      modelOutput%DepositOut(1:n) = -9999.0
      modelOutput%TsurfOut(1:n) = -9999.0

      ! ---- relaxation targets and coupling observation (setInputParam, src/InputOutput.f90:6-41) ----
      atm%TairR = real(localParam%tair_relax, 4)
      atm%VZR = real(localParam%VZ_relax, 4)
      atm%RhzR = real(localParam%RH_relax, 4)
      if (atm%TairR < -100.0 .or. atm%TairR > 100.0 .or. atm%VZR < 0.0 .or. atm%VZR > 100.0 .or. &
          atm%RhzR < 0.0 .or. atm%RhzR > 110) settings%use_relaxation = .false.
      coupling%carObsTime = -99
      coupling%obsI = -99
      coupling%obsTsurf = -99.0
      coupling%obsI(1) = localParam%couplingIndexI
      coupling%obsTsurf(1) = localParam%couplingTsurf
      coupling%lastTsurfObs = localParam%couplingTsurf
      coupling%NObs = 1
      if (localParam%couplingTsurf < -100 .or. coupling%obsI(1) < 1) settings%use_coupling = .false.

      ! ---- parameters (InitParam :168-205, condInit :319-371) ----
      phy%Grav = inputParam%Grav; phy%SB_Const = inputParam%SB_Const; phy%VK_Const = inputParam%VK_Const
      phy%ZRefW = inputParam%ZRefW; phy%ZRefT = inputParam%ZRefT; phy%ZeroDisp = inputParam%ZeroDisp
      phy%ZMom = inputParam%ZMom; phy%ZHeat = inputParam%ZHeat
      phy%logMom = consts%logMom; phy%logHeat = consts%logHeat
      phy%logCond = consts%logCond; phy%logUstar = consts%logUstar
      phy%Emiss = inputParam%Emiss; phy%MaxPormms = inputParam%MaxPormms; phy%TClimG = inputParam%TClimG
      phy%DampDpth = inputParam%DampDpth; phy%Omega = inputParam%Omega; phy%AZ = inputParam%AZ
      phy%LVap = inputParam%LVap; phy%LFus = inputParam%LFus; phy%vsh1 = inputParam%vsh1; phy%vsh2 = inputParam%vsh2
      phy%Poro1 = inputParam%Poro1; phy%Poro2 = inputParam%Poro2; phy%RhoB1 = inputParam%RhoB1
      phy%RhoB2 = inputParam%RhoB2; phy%Silt1 = inputParam%Silt1; phy%Silt2 = inputParam%Silt2
      condParam%WatDens = inputParam%WatDens; condParam%SnowDens = inputParam%SnowDens
      condParam%IceDens = inputParam%IceDens; condParam%DepDens = inputParam%DepDens
      condParam%WatMHeat = inputParam%WatMHeat; condParam%PorEvaF = inputParam%PorEvaF
      condParam%DampWearF = inputParam%DampWearF
      condParam%freezing_limit_normal = inputParam%freezing_limit_normal
      condParam%snow_melting_limit_normal = inputParam%snow_melting_limit_normal
      condParam%ice_melting_limit_normal = inputParam%ice_melting_limit_normal
      condParam%frost_melting_limit_normal = inputParam%frost_melting_limit_normal
      condParam%frost_formation_limit_normal = inputParam%frost_formation_limit_normal
      condParam%T4Melt_normal = inputParam%T4Melt_normal
      condParam%TLimFreeze = inputParam%freezing_limit_normal
      condParam%TLimMeltSnow = inputParam%snow_melting_limit_normal
      condParam%TLimMeltIce = inputParam%ice_melting_limit_normal
      condParam%TLimMeltDep = inputParam%frost_melting_limit_normal
      condParam%TLimDew = inputParam%frost_formation_limit_normal
      condParam%TLimColdH = inputParam%TLimColdH; condParam%TLimColdL = inputParam%TLimColdL
      condParam%WetSnowFormR = inputParam%WetSnowFormR; condParam%WetSnowMeltR = inputParam%WetSnowMeltR
      condParam%PLimSnow = inputParam%PLimSnow; condParam%PLimRain = inputParam%PLimRain
      condParam%MinPrecmm = inputParam%MinPrecmm; condParam%MinWatmms = inputParam%MinWatmms
      condParam%MinSnowmms = inputParam%MinSnowmms; condParam%MinDepmms = inputParam%MinDepmms
      condParam%MinIcemms = inputParam%MinIcemms; condParam%MaxSnowmms = inputParam%MaxSnowmms
      condParam%MaxDepmms = inputParam%MaxDepmms; condParam%MaxIcemms = inputParam%MaxIcemms
      condParam%MaxExtmms = inputParam%MaxExtmms; condParam%MaxWatmms = inputParam%MaxWatmms
      condParam%AlbDry = inputParam%AlbDry; condParam%AlbSnow = inputParam%AlbSnow
      condParam%MissValI = inputParam%MissValI; condParam%MissValR = inputParam%MissValR
      condParam%WDampLim = inputParam%WDampLim; condParam%WWetLim = inputParam%WWetLim
      condParam%WWearLim = inputParam%WWearLim; condParam%Snow2IceFac = inputParam%Snow2IceFac
      condParam%WetSnowFrozen = .false.; condParam%forceIceMelting = .false.
      condParam%forceSnowMelting = .false.; condParam%CanMeltingChangeTemperature = .true.

      ! ---- ground tables (allocator :96-114, initDepth / ground_prop_init through rs_build_constants) ----
      allocate (ground%condDZ(nl + 1), ground%capDZ(nl + 1), ground%Wcont(nl + 1), ground%VSH(nl + 1))
      allocate (ground%HS(nl + 1), ground%CC(nl + 1), ground%Tmp(0:nl + 1), ground%TmpNw(0:nl + 1))
      allocate (ground%DyC(nl + 1), ground%DyK(nl + 1), ground%ZDpth(nl + 1), ground%GCond(0:nl + 1))
      allocate (coupling%TmpSave(0:nl + 1))
      ground%condDZ = 0; ground%capDZ = 0; ground%Wcont = 0; ground%DyC = 0; ground%DyK = 0
      ground%VSH = -99.9; ground%HS = -99.9; ground%CC = -99.9; ground%GCond = -99.9
      ground%GroundFlux = -9999.9; ground%HStor = 0
      do j = 1, nl + 1
         ground%ZDpth(j) = consts%ZDpth(j)
      end do
      do j = 1, nl
         ground%DyC(j) = consts%DyC(j)
         ground%DyK(j) = ground%ZDpth(j + 1) - ground%ZDpth(j)
         ground%condDZ(j) = consts%condDZ(j)
         ground%Wcont(j) = consts%WCont(j)
      end do

      ! ---- surface, atmosphere, coupling scalars (initSurf :150-166, initVariables :207-236, initCoupling) ----
      surf%WearSurf = .true.
      surf%TrfFric = 5.0
      surf%EvapmmTS = 0.0
      surf%TSurfObs = -99.9
      atm%Tdew = -99.9; atm%PrecInTStep = -99.9; atm%BLCond = -99.9
      atm%RNet = 0; atm%LE_Flux = 0; atm%RainIntensity = 0; atm%SnowIntensity = 0
      atm%RainmmTS = 0.0; atm%SnowmmTS = 0.0; atm%SnowType = 0; atm%PrecType = 0
      atm%CalmLim = 0.4
      atm%SensibleHeatFlux = -9999.9
      coupling%Down = .false.; coupling%inCouplingPhase = .false.
      coupling%couplingStartI = -99; coupling%couplingEndI = -99
      coupling%couplingStartJ = -99; coupling%couplingEndJ = -99
      coupling%CoupPhaseN = 1
      coupling%TSurfAveSave = 0; coupling%SrfWatmmsSave = 0; coupling%SrfIcemmsSave = 0
      coupling%SrfIce2mmsSave = 0; coupling%SrfDepmmsSave = 0; coupling%SrfSnowmmsSave = 0
      coupling%AlbedoSave = 0; coupling%saveDatai = 0; coupling%saveDataj = 0
      if (settings%use_coupling) then  ! initCouplingTimes, src/Coupling.f90:512-517
         coupling%couplingEndI(1) = coupling%obsI(1)
         if (real(coupling%obsI(1), 8) <= consts%cplLenR) then
            coupling%couplingStartI(1) = 1
         else
            coupling%couplingStartI(1) = coupling%obsI(1) - consts%cplLenI
         end if
      end if

      ! the reference raises VZ(1) to 0.4 in the caller's array (:121-123)
      if (modelInput%VZ(1) < 0.4) modelInput%VZ(1) = 0.4
      atm%Tair = modelInput%Tair(1)
      atm%VZ = modelInput%VZ(1)
      atm%Rhz = modelInput%RHz(1)

      ! ---- the device context and the device part of the initialization ----
      tbottom = rs_bottom_temperature(inputParam, consts, modelInput%year(1), modelInput%month(1), modelInput%day(1))
      arr%tair = c_loc(modelInput%Tair(1)); arr%tdew = c_loc(modelInput%Tdew(1)); arr%vz = c_loc(modelInput%VZ(1))
      arr%rhz = c_loc(modelInput%Rhz(1)); arr%prec = c_loc(modelInput%prec(1)); arr%sw = c_loc(modelInput%SW(1))
      arr%lw = c_loc(modelInput%LW(1)); arr%sw_dir = c_loc(modelInput%SW_dir(1)); arr%lw_net = c_loc(modelInput%LW_net(1))
      arr%tsurfobs = c_loc(modelInput%TSurfObs(1)); arr%depth = c_loc(modelInput%depth(1))
      arr%precphase = c_loc(modelInput%PrecPhase(1)); arr%hour = c_loc(modelInput%hour(1))
      arr%horizons = c_loc(modelInput%local_horizons(1))
      arr%out(1) = c_loc(modelOutput%TsurfOut(1)); arr%out(2) = c_loc(modelOutput%SnowOut(1))
      arr%out(3) = c_loc(modelOutput%WaterOut(1)); arr%out(4) = c_loc(modelOutput%IceOut(1))
      arr%out(5) = c_loc(modelOutput%DepositOut(1)); arr%out(6) = c_loc(modelOutput%Ice2Out(1))
      sky = localParam%sky_view < 1.0 .and. localParam%sky_view > -0.01
      psun = c_null_ptr
      pgeo = c_null_ptr
      if (sky) then
         allocate (sun(6, n))
         call rs_sun_table(int(n, c_int), modelInput%year, modelInput%month, modelInput%day, modelInput%hour, &
                           modelInput%minute, modelInput%second, sun)
         lp(1) = localParam
         call rs_point_geometry(1_c_int, lp, geo(1:1), geo(2:2), geo(3:3))
         psun = c_loc(sun)
         pgeo = c_loc(geo)
      end if
      ctx = rs_compat_begin(consts, localParam, tbottom, arr, psun, pgeo, inputParam%Albedo_surroundings, st)
      if (.not. c_associated(ctx)) call stop_with_library_error('Initialization')
      call rs_surface_set_context(surf, ctx)
      ground%Tmp(0) = modelInput%Tair(1)
      ground%Tmp(nl + 1) = tbottom
      ground%TmpNw(0) = ground%Tmp(0)
      ground%TmpNw(nl + 1) = tbottom
      call load_state(st, nl, surf, ground)
      atm%TairInitEnd = st(S_TAIR_END)
      atm%VZInitEnd = st(S_VZ_END)
      atm%RhzInitEnd = st(S_RH_END)
      call load_coupling(st, coupling)
      coupling%lastTsurfObs = localParam%couplingTsurf
      if (coupling%lastTsurfObs < -100) coupling%Coupling_failed = .true.
   end subroutine Initialization

   !> src/InputOutput.f90:45-84, on the host: it owns the caller's loop condition.  (The fused device step
   !! of the index runs the same tests on the same values.)
   subroutine CheckValues(modelInput, i, settings, surf, localParam)
      use RoadSurfVariables
      type(InputArrays), intent(INOUT) :: modelInput
      integer, intent(IN) :: i
      type(SurfaceVariables), intent(IN) :: surf
      type(ModelSettings), intent(INOUT) :: settings
      type(LocalParameters), intent(IN) :: localParam
      logical :: bad
      bad = modelInput%Tair(i) < -90.0 .or. modelInput%Tair(i) > 100.0
      bad = bad .or. modelInput%Tdew(i) < -90 .or. modelInput%Tdew(i) > 100.0
      bad = bad .or. modelInput%RHz(i) < -0.1 .or. modelInput%RHz(i) > 120.0
      bad = bad .or. modelInput%VZ(i) < -1.0 .or. modelInput%VZ(i) > 100.0
      bad = bad .or. modelInput%SW(i) < -0.1 .or. modelInput%SW(i) > 4000.0
      bad = bad .or. modelInput%LW(i) < -0.1 .or. modelInput%LW(i) > 1000.0
      bad = bad .or. modelInput%prec(i) < -0.1 .or. modelInput%prec(i) > 500.0
      if (bad) then
         write (*, *) "BAD input value! ", modelInput%Tair(i), modelInput%Tdew(i), modelInput%RHz(i), &
            modelInput%VZ(i), modelInput%SW(i), modelInput%LW(i), modelInput%prec(i)
         settings%simulation_failed = .true.
      end if
      if (localParam%sky_view < 1.0 .and. localParam%sky_view > -0.01) then
         if (modelInput%SW_dir(i) < -0.1 .or. modelInput%SW_dir(i) > 4000.0 .or. &
             modelInput%LW_net(i) < -1000.0 .or. modelInput%LW_net(i) > 1000.0) then
            write (*, *) "BAD input value: SW_dir,LW_net", modelInput%SW_dir(i), modelInput%LW_net(i)
            settings%simulation_failed = .true.
         end if
      end if
      if (modelInput%SW_dir(i) > modelInput%SW(i)) modelInput%SW_dir(i) = modelInput%SW(i)
      if (surf%TsurfAve < -100.0 .or. surf%TsurfAve > 100.0) then
         write (*, *) "Abnormal surface temperature", surf%TsurfAve, i, localParam%lat, localParam%lon
         settings%simulation_failed = .true.
      end if
      ! a replay of the coupling window that failed on the device (rs_compat_replay) ends the loop here
      if (c_associated(rs_surface_context(surf))) then
         if (rs_compat_failed_index(rs_surface_context(surf)) > 0) settings%simulation_failed = .true.
      end if
   end subroutine CheckValues

   !> src/Coupling.f90:10-141.  The device keeps the coupling state machine (save at the window start, the
   !! coupling phase, the decaying corrections behind the window: step_kernel_cpl); the replays run at the
   !! window end (CheckEndCoupling below), so the caller's index is never taken back here.
   subroutine CouplingOperations1(i, coupling, surf, settings, ground, modelInput, CP, localParam)
      use RoadSurfVariables
      type(ModelSettings), intent(IN) :: settings
      type(InputArrays), intent(INOUT) :: modelInput
      type(RoadCondParameters), intent(IN) :: CP
      integer, intent(INOUT) :: i
      type(CouplingVariables), intent(INOUT) :: coupling
      type(SurfaceVariables), intent(INOUT) :: surf
      type(GroundVariables), intent(INOUT) :: ground
      type(LocalParameters), intent(IN) :: localParam
      if (.not. settings%use_coupling) return
      coupling%inCouplingPhase = i >= coupling%couplingStartI(1) .and. i <= coupling%couplingEndI(1)
   end subroutine CouplingOperations1

   !> src/Relaxation.f90:10-47: the host-visible air temperature, wind and humidity of the index (the fused
   !! step forms the same values from the same anchors; the anchors of the end of the initialization are set
   !! here as the reference sets them)
   subroutine RelaxationOperations(i, atm, settings, ground)
      use RoadSurfVariables
      integer, intent(IN) :: i
      type(ModelSettings), intent(IN) :: settings
      type(AtmVariables), intent(INOUT) :: atm
      type(GroundVariables), intent(INOUT) :: ground
      real(8) :: e
      if (i == settings%InitLenI) then
         atm%TairInitEnd = atm%Tair
         atm%VZInitEnd = atm%VZ
         atm%RhzInitEnd = atm%Rhz
      end if
      if (i > settings%InitLenI) then
         e = exp(-((settings%DTSecs*i) - (settings%DTSecs*settings%InitLenI))/(4.*3600.))
         atm%Tair = atm%Tair - (atm%TairR - atm%TairInitEnd)*e
         atm%VZ = atm%VZ - (atm%VZR - atm%VZInitEnd)*e
         atm%Rhz = atm%Rhz - (atm%RhzR - atm%RhzInitEnd)*e
         if (atm%Rhz > 100.) atm%Rhz = 100.0
         ground%Tmp(0) = atm%Tair
      end if
   end subroutine RelaxationOperations

   !> profile temperature at a depth (src/BalanceModel.f90:390-417), for the host-visible TsurfAve
   function temp_at_depth(ground, nl, depth) result(t)
      use RoadSurfVariables
      type(GroundVariables), intent(in) :: ground
      integer, intent(in) :: nl
      real(8), intent(in) :: depth
      real(8) :: t
      integer :: k
      t = 0.0
      if (abs(depth - 0.0) < 0.00001) then
         t = ground%Tmp(1)
      else if (depth > ground%ZDpth(nl + 1)) then
         t = ground%Tmp(nl + 1)
      else
         do k = 1, nl
            if (depth > ground%ZDpth(k) .and. depth <= ground%ZDpth(k + 1)) then
               t = ground%Tmp(k) + (depth - ground%ZDpth(k))*(ground%Tmp(k + 1) - ground%Tmp(k))/ &
                   (ground%ZDpth(k + 1) - ground%ZDpth(k))
               exit
            end if
         end do
      end if
   end function temp_at_depth

   !> src/InputOutput.f90:86-149: the host-visible forcing of the index and the observation forced on the two
   !! top layers during the initialization (the fused step does the same from the same arrays)
   subroutine SetCurrentValues(i, modelInput, atm, settings, surf, coupling, ground)
      use RoadSurfVariables
      integer, intent(IN) :: i
      type(ModelSettings), intent(IN) :: settings
      type(InputArrays), intent(IN) :: modelInput
      type(CouplingVariables), intent(IN) :: coupling
      type(AtmVariables), intent(INOUT) :: atm
      type(SurfaceVariables), intent(INOUT) :: surf
      type(GroundVariables), intent(INOUT) :: ground
      real(8) :: depth
      atm%Tair = modelInput%Tair(i)
      atm%Tdew = modelInput%Tdew(i)
      atm%VZ = modelInput%VZ(i)
      atm%Rhz = modelInput%RHz(i)
      atm%PrecInTStep = modelInput%prec(i)/3600*settings%DTSecs
      ground%Tmp(0) = atm%Tair
      if (i <= settings%InitLenI .or. settings%force_tsurf) then
         surf%TSurfObs = -9999.0
         if (modelInput%TsurfOBS(i) > -100.0) then
            if (.not. settings%use_coupling .or. i < coupling%couplingStartI(coupling%CoupPhaseN)) then
               surf%TSurfObs = modelInput%TsurfOBS(i)
               ground%Tmp(1) = surf%TSurfObs
               ground%Tmp(2) = surf%TSurfObs
               depth = modelInput%depth(i)
               if (settings%tsurfOutputDepth >= 0.0) depth = settings%tsurfOutputDepth
               if (depth >= 0) then
                  surf%TsurfAve = temp_at_depth(ground, settings%NLayers, depth)
               else
                  surf%TsurfAve = (ground%Tmp(1) + ground%Tmp(2))/2.0
               end if
            end if
         end if
      end if
   end subroutine SetCurrentValues

   !> src/BalanceModel.f90:7-86 and everything the fused kernel does around it: THE step of index inputIdx
   !! on the device, from the caller's arrays as they stand now.
   subroutine BalanceModelOneStep(SWi, LWi, phy, ground, surf, atm, settings, coupling, modelInput, inputIdx, condParam)
      use RoadSurfVariables
      real(8), intent(IN) :: SWi, LWi
      type(PhysicalParameters), intent(INOUT) :: phy
      type(CouplingVariables), intent(IN) :: coupling
      type(InputArrays), intent(IN) :: modelInput
      type(GroundVariables), intent(INOUT) :: ground
      type(SurfaceVariables), intent(INOUT) :: surf
      type(AtmVariables), intent(INOUT) :: atm
      type(ModelSettings), intent(INOUT) :: settings
      type(RoadCondParameters), intent(IN) :: condParam
      integer, intent(IN) :: inputIdx
      real(c_double) :: st(0:NSTATE - 1), edits(3)
      type(c_ptr) :: ctx
      ctx = rs_surface_context(surf)
      if (.not. c_associated(ctx)) then
         write (*, *) 'BalanceModelOneStep: the surface variables carry no device context (call Initialization first)'
         error stop 'module RoadSurf over libroadsurf_hip'
      end if
      edits = [modelInput%SW(inputIdx), modelInput%SW_dir(inputIdx), modelInput%LW(inputIdx)]
      if (rs_compat_step(ctx, int(inputIdx, c_int), st, edits) /= 0) call stop_with_library_error('BalanceModelOneStep')
      ! the sky view's edits of the caller's arrays (src/ModRadiation.f90:57-71); unchanged values otherwise
      modelInput%SW(inputIdx) = edits(1)
      modelInput%SW_dir(inputIdx) = edits(2)
      modelInput%LW(inputIdx) = edits(3)
      call load_state(st, settings%NLayers, surf, ground)
      atm%TairInitEnd = st(S_TAIR_END)
      atm%VZInitEnd = st(S_VZ_END)
      atm%RhzInitEnd = st(S_RH_END)
   end subroutine BalanceModelOneStep

   !> src/InputOutput.f90:151-165
   subroutine SaveOutput(modelOutput, i, surf)
      use RoadSurfVariables
      integer, intent(IN) :: i
      type(SurfaceVariables), intent(IN) :: surf
      type(OutputArrays), intent(INOUT) :: modelOutput
      integer :: fi
      real(c_double) :: row(6)
      if (c_associated(rs_surface_context(surf))) then
         ! an index behind the one at which the device failed the point (a replay of the coupling window) was
         ! not stepped: the reference never reaches its SaveOutput
         fi = rs_compat_failed_index(rs_surface_context(surf))
         if (fi > 0 .and. i > fi) return
         ! the row of this index as the fused step wrote it - what the reference's surface variables hold HERE,
         ! before CheckEndCoupling touches TsurfAve at the end of a coupling window (Simulation.f90:87-91)
         if (rs_compat_outputs(rs_surface_context(surf), int(i, c_int), row) == 0) then
            modelOutput%TsurfOut(i) = row(1)
            modelOutput%SnowOut(i) = row(2)
            modelOutput%WaterOut(i) = row(3)
            modelOutput%IceOut(i) = row(4)
            modelOutput%DepositOut(i) = row(5)
            modelOutput%Ice2Out(i) = row(6)
            return
         end if
      end if
      modelOutput%SnowOut(i) = surf%SrfSnowmms
      modelOutput%WaterOut(i) = surf%SrfWatmms
      modelOutput%IceOut(i) = surf%SrfIcemms
      modelOutput%Ice2Out(i) = surf%SrfIce2mms
      modelOutput%DepositOut(i) = surf%SrfDepmms
      modelOutput%TsurfOut(i) = surf%TsurfAve
   end subroutine SaveOutput

   !> src/Coupling.f90:98-141: at the end of the coupling window Coupling_control has decided on the device;
   !! every replay it asks for runs now (rs_compat_replay), the rows of the window in the caller's output
   !! arrays are rewritten, the derived types show the state behind the last replay.
   subroutine CheckEndCoupling(i, settings, coupling, surf)
      use RoadSurfVariables
      integer, intent(IN) :: i
      type(ModelSettings), intent(IN) :: settings
      type(SurfaceVariables), intent(INOUT) :: surf
      type(CouplingVariables), intent(INOUT) :: coupling
      real(c_double) :: st(0:NSTATE - 1)
      integer(c_int) :: rw(2)
      type(c_ptr) :: ctx
      if (.not. settings%use_coupling) return
      ctx = rs_surface_context(surf)
      if (.not. c_associated(ctx)) return
      if (i == coupling%couplingEndI(1)) then
         if (rs_compat_replay(ctx, int(i, c_int), st, rw) /= 0) call stop_with_library_error('CheckEndCoupling')
         surf%TsurfAve = st(S_TSURF)
         surf%SrfWatmms = st(S_WAT)
         surf%SrfSnowmms = st(S_SNOW)
         surf%SrfIcemms = st(S_ICE)
         surf%SrfIce2mms = st(S_ICE2)
         surf%SrfDepmms = st(S_DEP)
         surf%Q2Melt = st(S_Q2MELT)
         surf%T4Melt = st(S_T4MELT)
         surf%VeryCold = st(S_VERYCOLD) /= 0.0d0
         coupling%inCouplingPhase = .false.
      else
         if (rs_compat_last_state(ctx, st) /= 0) return
      end if
      call load_coupling(st, coupling)
   end subroutine CheckEndCoupling

   !> src/Storage.f90:9-29: inside the fused step of the index (BalanceModelOneStep)
   subroutine PrecipitationToStorage(settings, CP, PrecPhase, atm, surf)
      use RoadSurfVariables
      type(ModelSettings), intent(IN) :: settings
      type(RoadCondParameters), intent(IN) :: CP
      integer, intent(IN) :: PrecPhase
      type(AtmVariables), intent(INOUT) :: atm
      type(SurfaceVariables), intent(INOUT) :: surf
   end subroutine PrecipitationToStorage

   !> src/ModRadiation.f90:7-73: inside the fused step of the index; its edits of SW(i), SW_dir(i), LW(i)
   !! reach the caller's arrays when that step returns (BalanceModelOneStep)
   subroutine ModRadiationBySurroundings(modelInput, inputParam, localParam, i)
      use RoadSurfVariables
      type(InputArrays), intent(INOUT) :: modelInput
      type(InputParameters), intent(IN) :: inputParam
      type(LocalParameters), intent(IN) :: localParam
      integer, intent(IN) :: i
   end subroutine ModRadiationBySurroundings

   !> src/Cond.f90:69-103: the reference's rates from the storages as the derived type shows them - at this
   !! point of the index already the ones BEHIND RoadCond, which ran inside the fused step with rates of its
   !! own; Snow2IceFac as the reference overwrites it (:86)
   subroutine WearFactors(Snow2IceFac, Tph, surf, wearF)
      use RoadSurfVariables
      real(8), intent(IN) :: Tph
      type(SurfaceVariables), intent(IN) :: surf
      type(WearingFactors), intent(OUT) :: wearF
      real(8), intent(INOUT) :: Snow2IceFac
      ! (default-REAL literals and their folds as in the reference: REAL(4) products widened by the storage)
      wearF%SnowTran = max((0.2 + 0.25)*surf%SrfSnowmms, 0.01)
      if (surf%SrfSnowmms < 0.2) wearF%SnowTran = wearF%SnowTran*3 ! a thin snow layer wears faster
      Snow2IceFac = 0.25/(0.2 + 0.25)
      wearF%SnowTran = wearF%SnowTran*Tph
      wearF%IceWear = max(1.1*2.0*0.145*surf%SrfIcemms, 0.01)*Tph
      wearF%IceWear2 = max(1.1*2.0*(4.0*0.290)*surf%SrfIce2mms, 0.01)*Tph
      wearF%DepWear = max(0.5*2.0*(4.0*0.290)*surf%SrfDepmms, 0.01)*Tph
      wearF%WatWear = 10*max(0.145*surf%SrfWatmms, 0.06)*Tph
      ! (the reference leaves the other members of its INTENT(OUT) argument undefined)
      wearF%SnowTran2 = 0; wearF%SnowTranDef = 0; wearF%IceWearNight = 0; wearF%IceWear2Night = 0
      wearF%IceWearSW = 0; wearF%IceWear2SW = 0
   end subroutine WearFactors

   !> src/Cond.f90:9-65: inside the fused step of the index (BalanceModelOneStep)
   subroutine RoadCond(MaxPormms, surf, atm, settings, CP, wearF)
      use RoadSurfVariables
      real(8), intent(IN) :: MaxPormms
      type(ModelSettings), intent(IN) :: settings
      type(RoadCondParameters), intent(INOUT) :: CP
      type(SurfaceVariables), intent(INOUT) :: surf
      type(AtmVariables), intent(INOUT) :: atm
      type(WearingFactors), intent(IN) :: wearF
   end subroutine RoadCond

   !> src/Cond.f90:105-139: the albedo is part of the state the fused step leaves (ground%Albedo holds it)
   subroutine CalcAlbedo(albedo, surf, CP)
      use RoadSurfVariables
      type(SurfaceVariables), intent(IN) :: surf
      type(RoadCondParameters), intent(IN) :: CP
      real(8), intent(INOUT) :: Albedo
   end subroutine CalcAlbedo

end module RoadSurf

!> src/InputOutput.f90:169-198: external in the reference too (examples/example1/src/Simulation.f90:104
!! calls it without an interface).  The host-visible forcing of the last index; the fused step of that
!! index (BalanceModelOneStep with inputIdx = SimLen) takes the reference's final-step rules itself.
subroutine lastValues(modelInput, atm, settings, ground, surf)
   use RoadSurfVariables
   implicit none
   type(ModelSettings), intent(IN) :: settings
   type(InputArrays), intent(IN) :: modelInput
   type(AtmVariables), intent(INOUT) :: atm
   type(GroundVariables), intent(INOUT) :: ground
   type(SurfaceVariables), intent(INOUT) :: surf
   integer :: n
   n = settings%SimLen
   atm%Tair = modelInput%Tair(n)
   atm%Tdew = modelInput%Tdew(n)
   atm%VZ = modelInput%VZ(n)
   atm%Rhz = modelInput%RHz(n)
   atm%PrecInTStep = modelInput%prec(n)/3600*settings%DTSecs
   ground%Tmp(0) = atm%Tair
   if (.not. modelInput%depth(n) >= 0) surf%TsurfAve = (ground%Tmp(1) + ground%Tmp(2))/2.0
end subroutine lastValues
