#!/bin/bash
# TEST INFRASTRUCTURE -- builds the *unmodified* reference hot-path modules
# in place from /root/reference (nothing is copied into this repo) with
# amdflang, against the build-owned Icepack interface stub, and links them
# with the build-owned harnesses in this directory.  Outputs go ONLY to
# oracle/_ref/ (git-ignored; travels to the GPU box with the snapshot).
#
# Recipe = SURVEY.md Appendix A.1 (verified compile order).
#
#   usage: oracle/ref/build_ref.sh [strict|fast|mpi]   (default: all three)
#     strict : -O2 -ffp-contract=off  (bitwise-comparable with oracle/evp_oracle.c)
#     fast   : -O2 -fopenmp           (the reference's ordinary optimisation level; CPU baseline)
#     fma    : -O2 -march=native -ffp-contract=fast  (only on request)
#     mpi    : the same two, with the reference's comm/mpi modules instead of comm/serial
#              (mpistrict, mpifast: the reference's own MPI path, run under mpiexec).  MPI = the image's
#              MPICH 3.3.2 (/opt/conda).  Its mpi.mod is in gfortran's format, which flang cannot read;
#              mpi_from_mpif.F90 compiles MPICH's OWN Fortran header /opt/conda/include/mpif.h into a
#              module named mpi in flang's format -- every declaration in it is MPICH's, nothing is
#              restated here -- and the executables link the image's libmpifort / libmpi.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REPO="$(cd "$HERE/../.." && pwd)"
REF=${CICE_REFERENCE:-/root/reference}
OUT="$REPO/oracle/_ref"
FC=${FC:-/opt/rocm/bin/amdflang}

if [ ! -d "$REF/cicecore" ]; then
  echo "build_ref: $REF not present (GPU box?) -- using prebuilt oracle/_ref" >&2
  exit 0
fi

R=$REF/cicecore
MPIPREFIX=${MPIPREFIX:-/opt/conda}

set_sources () {       # $1 = serial | mpi
S=$R/cicedyn/infrastructure/comm/$1
SRCS=(
  $R/shared/ice_kinds_mod.F90 $R/shared/ice_constants.F90 $R/shared/ice_fileunits.F90
  $S/ice_exit.F90 $S/ice_communicate.F90
  $R/shared/ice_domain_size.F90
  $R/cicedyn/infrastructure/ice_blocks.F90 $R/cicedyn/infrastructure/ice_memusage.F90
  $R/shared/ice_spacecurve.F90 $R/shared/ice_distribution.F90
  $S/ice_broadcast.F90 $S/ice_reprosum.F90 $S/ice_gather_scatter.F90
  $S/ice_global_reductions.F90 $S/ice_boundary.F90
  $R/cicedyn/infrastructure/ice_domain.F90
  $S/ice_timers.F90
  $R/cicedyn/infrastructure/ice_read_write.F90
  $R/shared/ice_calendar.F90
  $R/cicedyn/infrastructure/ice_grid.F90
  $R/shared/ice_arrays_column.F90
  $R/cicedyn/general/ice_state.F90 $R/cicedyn/general/ice_flux_bgc.F90 $R/cicedyn/general/ice_flux.F90
  $R/cicedyn/dynamics/ice_dyn_shared.F90
  $R/cicedyn/infrastructure/ice_restoring.F90
  $R/cicedyn/dynamics/ice_dyn_core1d.F90
)
}
# ice_dyn_evp1d is either the reference's own (refonly) or a build-owned
# module of the same name (capture harness / HIP drop-in): see below.
EVP1D_REF=$R/cicedyn/dynamics/ice_dyn_evp1d.F90
EVP=$R/cicedyn/dynamics/ice_dyn_evp.F90
CSRCS=( $R/cicedyn/infrastructure/ice_shr_reprosum86.c $R/cicedyn/infrastructure/ice_memusage_gptl.c )

build_variant () {
  local variant=$1; shift
  local fflags="$*"
  local O="$OUT/obj_$variant"
  local LIBS=""
  mkdir -p "$O"
  case "$variant" in
    mpi*) set_sources mpi
          # only the four libraries MPICH needs that the system does not have, so that libgcc_s / libgomp /
          # libstdc++ keep coming from the system (the HIP runtime needs the newer ones)
          mkdir -p "$OUT/mpilib"
          for l in libmpi.so.12 libmpifort.so.12 libgfortran.so.4 libquadmath.so.0; do
            [ -e "$OUT/mpilib/$l" ] || cp -L "$MPIPREFIX/lib/$l" "$OUT/mpilib/$l"
          done
          LIBS="-L$OUT/mpilib -l:libmpifort.so.12 -l:libmpi.so.12 -Wl,-rpath,\$ORIGIN/mpilib" ;;
    *)    set_sources serial ;;
  esac
  ( cd "$O"
    case "$variant" in
      mpi*) [ -f mpi.o ] || $FC $fflags -cpp -I"$MPIPREFIX/include" -c "$HERE/mpi_from_mpif.F90" -o mpi.o ;;
    esac
    $FC $fflags -cpp -c "$HERE/icepack_intfc_stub.F90" -o icepack_intfc.o
    for f in "${SRCS[@]}"; do
      b=$(basename "$f" .F90)
      if [ ! -f "$b.o" ] || [ "$f" -nt "$b.o" ]; then
        $FC $fflags -cpp -c "$f" -o "$b.o"
      fi
    done
    for f in "${CSRCS[@]}"; do
      b=$(basename "$f" .c)
      [ -f "$b.o" ] || gcc -O2 -DFORTRANUNDERSCORE -c "$f" -o "$b.o"
    done
    COMMON=$(ls *.o | grep -v -E '^(ice_dyn_evp|ice_dyn_evp1d|evp_.*)\.o$' | tr '\n' ' ')

    # (1) fixture generator: capture module stands in for ice_dyn_evp1d so that
    #     evp() hands over every (otherwise private) subcycle input; the very
    #     next evp() call with evp_algorithm='standard_2d' gives the golden output.
    mkdir -p cap && cd cap
    $FC $fflags -cpp -I.. -c "$HERE/evp_dumpio.F90" -o evp_dumpio.o
    $FC $fflags -cpp -I.. -I. -c "$HERE/ice_dyn_evp1d_capture.F90" -o ice_dyn_evp1d.o
    $FC $fflags -cpp -I.. -I. -c "$EVP" -o ice_dyn_evp.o
    gcc -O2 -c "$HERE/evp_peek.c" -o evp_peek.o
    $FC $fflags -cpp -I.. -I. -c "$HERE/evp_ref_harness.F90" -o evp_ref_harness.o
    $FC $fflags evp_ref_harness.o evp_dumpio.o evp_peek.o ice_dyn_evp1d.o ice_dyn_evp.o \
        $(for o in $COMMON; do echo ../$o; done) $LIBS -o "$OUT/evp_ref_harness_$variant"
    cd ..

    # (1b) CPU baseline of the reference's own 1-d core (evp_algorithm='shared_mem_1d', its fastest CPU
    #      form per SURVEY 8 a11): the reference's ice_dyn_evp1d.F90, timing only, no capture
    if [ "$variant" = fast ]; then
      mkdir -p ref1d && cd ref1d
      $FC $fflags -cpp -I.. -c "$HERE/evp_dumpio.F90" -o evp_dumpio.o
      $FC $fflags -cpp -I.. -I. -c "$EVP1D_REF" -o ice_dyn_evp1d.o
      $FC $fflags -cpp -I.. -I. -c "$EVP" -o ice_dyn_evp.o
      gcc -O2 -c "$HERE/evp_peek.c" -o evp_peek.o
      $FC $fflags -cpp -DHARNESS_REF1D -I.. -I. -c "$HERE/evp_ref_harness.F90" -o evp_ref_harness.o
      $FC $fflags evp_ref_harness.o evp_dumpio.o evp_peek.o ice_dyn_evp1d.o ice_dyn_evp.o \
          $(for o in $COMMON; do echo ../$o; done) -o "$OUT/evp_ref_harness_fast1d"
      cd ..
      echo "built $OUT/evp_ref_harness_fast1d"
    fi

    # (2) drop-in demonstration: the reference's unmodified evp() driver linked with the
    #     build-owned `ice_dyn_evp1d` that forwards to the HIP core (cice_amd/fortran),
    #     i.e. Option B of INTEGRATION.md.  Needs cice_amd/libcice_evp_hip.so.
    if { [ "$variant" = strict ] || [ "$variant" = mpistrict ]; } && [ -f "$REPO/cice_amd/libcice_evp_hip.so" ]; then
      local dropin=evp_hip_dropin_harness
      [ "$variant" = mpistrict ] && dropin=evp_hip_dropin_harness_mpi
      mkdir -p hip && cd hip
      $FC $fflags -cpp -I.. -c "$HERE/evp_dumpio.F90" -o evp_dumpio.o
      $FC $fflags -cpp -I.. -I. -c "$REPO/cice_amd/fortran/ice_dyn_evp_hip.F90" -o ice_dyn_evp_hip.o
      $FC $fflags -cpp -I.. -I. -c "$REPO/cice_amd/fortran/ice_dyn_evp1d_hip.F90" -o ice_dyn_evp1d.o
      $FC $fflags -cpp -I.. -I. -c "$EVP" -o ice_dyn_evp.o
      gcc -O2 -c "$HERE/evp_peek.c" -o evp_peek.o
      $FC $fflags -cpp -DHARNESS_HIP_BODY -I.. -I. -c "$HERE/evp_ref_harness.F90" -o evp_ref_harness.o
      $FC $fflags evp_ref_harness.o evp_dumpio.o evp_peek.o ice_dyn_evp_hip.o ice_dyn_evp1d.o ice_dyn_evp.o \
          $(for o in $COMMON; do echo ../$o; done) \
          -L"$REPO/cice_amd" -lcice_evp_hip -Wl,-rpath,'$ORIGIN/../../cice_amd' $LIBS -o "$OUT/$dropin"
      cd ..
      echo "built $OUT/$dropin"
    fi
  )
  echo "built $OUT/evp_ref_harness_$variant"
}

want=${1:-both}
if [ "$want" = strict ] || [ "$want" = both ]; then build_variant strict -O2 -ffp-contract=off; fi
if [ "$want" = fast ]   || [ "$want" = both ]; then build_variant fast   -O2 -fopenmp; fi
if [ "$want" = mpi ]    || [ "$want" = both ]; then
  if [ -f "$MPIPREFIX/include/mpif.h" ] && [ -f "$MPIPREFIX/lib/libmpifort.so.12" ]; then
    build_variant mpistrict -O2 -ffp-contract=off
    build_variant mpifast   -O2
  else
    echo "build_ref: no MPICH under $MPIPREFIX -- the MPI variants are skipped" >&2
  fi
fi
# fma: contraction really happens (x86-64 baseline has no FMA, so "fast" == strict bit for bit);
#      used once to measure the reference's own build-to-build spread (tolerance)
if [ "$want" = fma ]; then build_variant fma -O2 -march=native -ffp-contract=fast; fi
