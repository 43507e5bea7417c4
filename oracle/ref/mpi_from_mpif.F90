!=======================================================================
! TEST INFRASTRUCTURE (oracle/ref) -- the Fortran module `mpi` of the image's MPICH 3.3.2 in flang's
! module format.  The image ships /opt/conda/include/mpi.mod in gfortran's format, which amdflang cannot
! read, and /opt/conda/include/mpif.h, the same interface as an include file.  This file holds no
! declaration of its own: everything the reference's `use mpi` (comm/mpi/ice_communicate.F90:11 and the
! other comm/mpi modules) sees comes out of MPICH's header, and the executables link MPICH's libraries.
!=======================================================================
module mpi
  implicit none
  include 'mpif.h'
end module mpi
