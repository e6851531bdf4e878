!> oracle/ref_probe.f90 — TEST INFRASTRUCTURE (ours; compiled by oracle/build_ref.sh
!! against the REFERENCE's own modules, linked into oracle/_ref/libroadsurf_ref.so).
!!
!! Thin BIND(C) windows onto reference internals so that known-answer fixtures
!! can be captured from the reference itself (tests/golden/make_golden.py):
!!   ref_probe_init    -> products of the reference's Initialization
!!                        (src/Initialization.f90:9-147): layer grid, DyC, DyK, CC,
!!                        condDZ, initial Tmp(0:N+1), the four logarithms
!!   ref_probe_blcond  -> one CalcBLCondAndLE call (src/BoundaryLayer.f90:3-109)
!! Nothing here restates the algorithm; it only calls the reference.
subroutine ref_probe_init(inPointers, outPointers, inSettings, inputParam, localParam, &
                          zdpth, dyc, dyk, cc, conddz, tmp, logs, tsurfave) bind(C, name='ref_probe_init')
   use, intrinsic :: iso_c_binding
   use RoadSurfVariables
   use RoadSurf
   implicit none
   type(InputPointers), intent(in) :: inPointers
   type(OutputPointers), intent(inout) :: outPointers
   type(InputSettings), intent(in) :: inSettings
   type(InputParameters), intent(in) :: inputParam
   type(LocalParameters), intent(in) :: localParam
   real(c_double), intent(out) :: zdpth(*), dyc(*), dyk(*), cc(*), conddz(*), tmp(*), logs(4), tsurfave
   type(InputArrays) :: modelInput
   type(OutputArrays) :: modelOutput
   type(PhysicalParameters) :: phy
   type(GroundVariables) :: ground
   type(SurfaceVariables) :: surf
   type(AtmVariables) :: atm
   type(CouplingVariables) :: coupling
   type(ModelSettings) :: settings
   type(RoadCondParameters) :: condParam
   integer :: n, i
   call ConnectFortran2Carrays(inPointers, modelInput, outPointers, modelOutput)
   call Initialization(modelInput, inSettings, settings, modelOutput, atm, surf, inputParam, &
                       localParam, coupling, phy, ground, condParam)
   n = settings%NLayers
   do i = 1, n + 1
      zdpth(i) = ground%ZDpth(i)
   end do
   do i = 1, n
      dyc(i) = ground%DyC(i)
      dyk(i) = ground%DyK(i)
      cc(i) = ground%CC(i)
      conddz(i) = ground%condDZ(i)
   end do
   do i = 0, n + 1
      tmp(i + 1) = ground%Tmp(i)
   end do
   logs(1) = phy%logMom
   logs(2) = phy%logHeat
   logs(3) = phy%logCond
   logs(4) = phy%logUstar
   tsurfave = surf%TsurfAve
end subroutine ref_probe_init

subroutine ref_probe_blcond(inputParam, dtsecs, tsurfave, tair, vz, rhz, srfwat, blcond, le, evap) &
   bind(C, name='ref_probe_blcond')
   use, intrinsic :: iso_c_binding
   use RoadSurfVariables
   implicit none
   type(InputParameters), intent(in) :: inputParam
   real(c_double), value :: dtsecs, tsurfave, tair, vz, rhz, srfwat
   real(c_double), intent(out) :: blcond, le, evap
   type(PhysicalParameters) :: phy
   type(AtmVariables) :: atm
   real(8) :: albedo, ev
   external :: InitParam, CalcBLCondAndLE
   call InitParam(albedo, phy, inputParam)
   atm%Tair = tair
   atm%VZ = vz
   atm%Rhz = rhz
   atm%BLCond = -99.9
   atm%LE_Flux = 0.0
   ev = 0.0
   call CalcBLCondAndLE(tsurfave, ev, dtsecs, srfwat, phy, atm)
   blcond = atm%BLCond
   le = atm%LE_Flux
   evap = ev
end subroutine ref_probe_blcond

!> Coupling set-up as the reference's Initialization leaves it (src/Coupling.f90:486-534).
subroutine ref_probe_coupling_init(inPointers, outPointers, inSettings, inputParam, localParam, &
                                   ivals, rvals) bind(C, name='ref_probe_coupling_init')
   use, intrinsic :: iso_c_binding
   use RoadSurfVariables
   use RoadSurf
   implicit none
   type(InputPointers), intent(in) :: inPointers
   type(OutputPointers), intent(inout) :: outPointers
   type(InputSettings), intent(in) :: inSettings
   type(InputParameters), intent(in) :: inputParam
   type(LocalParameters), intent(in) :: localParam
   integer(c_int), intent(out) :: ivals(8)
   real(c_double), intent(out) :: rvals(4)
   type(InputArrays) :: modelInput
   type(OutputArrays) :: modelOutput
   type(PhysicalParameters) :: phy
   type(GroundVariables) :: ground
   type(SurfaceVariables) :: surf
   type(AtmVariables) :: atm
   type(CouplingVariables) :: coupling
   type(ModelSettings) :: settings
   type(RoadCondParameters) :: condParam
   call ConnectFortran2Carrays(inPointers, modelInput, outPointers, modelOutput)
   call Initialization(modelInput, inSettings, settings, modelOutput, atm, surf, inputParam, &
                       localParam, coupling, phy, ground, condParam)
   ivals(1) = merge(1, 0, settings%use_coupling)
   ivals(2) = coupling%obsI(1)
   ivals(3) = coupling%couplingStartI(1)
   ivals(4) = coupling%couplingEndI(1)
   ivals(5) = coupling%NObs
   ivals(6) = coupling%CoupPhaseN
   ivals(7) = merge(1, 0, coupling%Coupling_failed)
   ivals(8) = merge(1, 0, settings%use_relaxation)
   rvals(1) = coupling%lastTsurfObs
   rvals(2) = coupling%obsTsurf(1)
   rvals(3) = atm%TairR
   rvals(4) = coupling%RadCoeff
end subroutine ref_probe_coupling_init

!> Flush what the reference has written to standard output (unit 6) so far.  The tests silence
!! the reference's diagnostics by pointing file descriptor 1 elsewhere for the duration of a call
!! (tests/oracle_helpers.py, quiet_stdout); the Fortran runtime buffers unit 6, so the buffer has to
!! be emptied before the descriptor is restored.
subroutine ref_flush_stdout() bind(C, name='ref_flush_stdout')
   flush (6)
end subroutine ref_flush_stdout
