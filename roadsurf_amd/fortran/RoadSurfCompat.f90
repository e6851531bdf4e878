!> Source-level names of the reference for caller code written against it.
!!
!! The reference keeps its derived types in `module RoadSurfVariables`
!! (src/RoadSurfVariables.f90:13-28) and its procedures in `module RoadSurf`
!! (src/RoadSurf.f90:6-270).  Caller-side Fortran (a driver like the reference's own
!! examples/example1/src/Simulation.f90, or glue that fills the boundary structs) says
!! `use RoadSurfVariables` to get the five `Bind(C)` boundary types.  These two modules give that
!! code the same names over THIS library: the types are the ones of module RoadSurfHip (field for
!! field the reference's, src/InputPointers.f90.inc:4-27, src/OutputPointers.f90.inc:4-17,
!! src/InputSettings.f90.inc:4-18, src/InputParameters.f90.inc:4-91, src/LocalParameters.f90.inc:4-15),
!! and `module RoadSurf` exports the entry points that exist here: `runsimulation` (the reference's
!! BIND(C) procedure, one point) and `runsimulation_batch` (many points).
!!
!! Deliberately NOT here: the reference's eleven Fortran-only state types (GroundVariables, ...) and
!! its fourteen per-point, per-time-step procedures (Initialization, BalanceModelOneStep, ...).
!! They are a CPU API over one point's state in host memory; behind this library that state lives
!! in GPU registers for a whole window of time steps, and a host implementation of those
!! procedures would be a CPU path, which this library does not have.
module RoadSurfVariables
   use, intrinsic :: iso_c_binding
   use RoadSurfHip, only: InputPointers, OutputPointers, InputSettings, InputParameters, LocalParameters
   implicit none
   public
end module RoadSurfVariables

module RoadSurf
   use RoadSurfVariables
   use RoadSurfHip, only: runsimulation, runsimulation_batch
   implicit none
   public
end module RoadSurf
