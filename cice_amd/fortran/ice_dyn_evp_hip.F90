!=======================================================================
! ice_dyn_evp_hip -- Fortran host side of the MI355X-native EVP core.
!
! ISO_C_BINDING interfaces of the C ABI in include/cice_evp_hip.h plus the
! three routines CICE's evp() needs, with the same call shape as the
! reference's alternative EVP core (module ice_dyn_evp1d, public
! dyn_evp1d_init / dyn_evp1d_run / dyn_evp1d_finalize,
! cicecore/cicedyn/dynamics/ice_dyn_evp1d.F90:25,73,121):
!
!    call dyn_evp_hip_init                       ! ice_dyn_evp.F90:153-155
!    call dyn_evp_hip_run(stressp_1, ..., iceUmask)   ! ice_dyn_evp.F90:846-856
!    call dyn_evp_hip_finalize
!
! CICE's module arrays (nx_block,ny_block,max_blocks) are passed straight to
! the library (sequence association, no copies).  Errors follow the
! reference's fail-stop convention: non-zero status -> abort_ice(msg,file,line)
! (comm/*/ice_exit.F90).  Unlike the reference's 1-d core the HIP core stays
! block-distributed: every MPI rank drives its own GPU and the velocity halo
! runs over RCCL; with the serial comm layer everything is on one GPU.
!
! This file is new code written for this repository (no reference code).
!=======================================================================
module ice_dyn_evp_hip

  use, intrinsic :: iso_c_binding
  use ice_kinds_mod
  use ice_exit, only: abort_ice

  implicit none
  private

  public :: dyn_evp_hip_init, dyn_evp_hip_run, dyn_evp_hip_finalize, dyn_evp_hip_evp_body, &
            dyn_evp_hip_fetch_stresses, dyn_evp_hip_invalidate_stresses, dyn_evp_hip_cgrid_run, &
            dyn_evp_hip_keep_stresses_resident, dyn_evp_hip_cgrid_deformations, dyn_evp_hip_cgrid_dyn_finish, &
            dyn_evp_hip_cgrid_evp_body, &
            dyn_evp_hip_cgrid_fetch_forcing

  ! mirror of cice_evp_hip_dims (include/cice_evp_hip.h)
  type, bind(C) :: cice_evp_hip_dims
     integer(c_int32_t) :: nx_block, ny_block, nblocks, max_blocks, nghost
     integer(c_int32_t) :: nx_global, ny_global
     integer(c_int32_t) :: ew_boundary_type, ns_boundary_type
     integer(c_int32_t) :: rank, nranks
     type(c_ptr) :: ilo, ihi, jlo, jhi, iglob0, jglob0
     integer(c_int32_t) :: nblocks_tot
     type(c_ptr) :: gi0, gj0, gnx, gny, gowner, glocal
  end type cice_evp_hip_dims

  ! mirror of cice_evp_hip_params
  type, bind(C) :: cice_evp_hip_params
     integer(c_int32_t) :: ndte, strict
     real(c_double) :: arlx1i, denom1, brlx, revp, e_factor, epp2i
     real(c_double) :: capping, Ktens, deltaminEVP, u0, cosw, sinw, rhow
  end type cice_evp_hip_params

  ! mirror of cice_evp_hip_prep_params
  type, bind(C) :: cice_evp_hip_prep_params
     real(c_double) :: dt, rhoi, rhos, gravit, dyn_area_min, dyn_mass_min
     integer(c_int32_t) :: ssh_stress_coupled
  end type cice_evp_hip_prep_params

  interface
     integer(c_int) function cice_evp_hip_init(dims, params, HTE, HTN, dxT, dyT, uarear, tarea) &
          bind(C, name='cice_evp_hip_init')
       import :: c_int, c_double, cice_evp_hip_dims, cice_evp_hip_params
       type(cice_evp_hip_dims), intent(in) :: dims
       type(cice_evp_hip_params), intent(in) :: params
       real(c_double), dimension(*), intent(in) :: HTE, HTN, dxT, dyT, uarear, tarea
     end function cice_evp_hip_init

     integer(c_int) function cice_evp_hip_run( &
          stressp_1, stressp_2, stressp_3, stressp_4, stressm_1, stressm_2, stressm_3, stressm_4, &
          stress12_1, stress12_2, stress12_3, stress12_4, strength, cdn_ocnU, aiU, uocnU, vocnU, &
          waterxU, wateryU, forcexU, forceyU, umassdti, fmU, strintxU, strintyU, TbU, taubxU, &
          taubyU, uvel, vvel, uvel_init, vvel_init, iceTmask, iceUmask, ndte) &
          bind(C, name='cice_evp_hip_run')
       import :: c_int, c_int32_t, c_double
       real(c_double), dimension(*), intent(inout) :: &
          stressp_1, stressp_2, stressp_3, stressp_4, stressm_1, stressm_2, stressm_3, stressm_4, &
          stress12_1, stress12_2, stress12_3, stress12_4, strintxU, strintyU, taubxU, taubyU, &
          uvel, vvel
       real(c_double), dimension(*), intent(in) :: &
          strength, cdn_ocnU, aiU, uocnU, vocnU, waterxU, wateryU, forcexU, forceyU, umassdti, &
          fmU, TbU, uvel_init, vvel_init
       integer(c_int32_t), dimension(*), intent(in) :: iceTmask, iceUmask   ! logical(4): non-zero = .true.
       integer(c_int32_t), value :: ndte
     end function cice_evp_hip_run

     integer(c_int) function cice_evp_hip_set_metrics(cxp, cyp, cxm, cym, dxhy, dyhx, DminTarea) &
          bind(C, name='cice_evp_hip_set_metrics')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: cxp, cyp, cxm, cym, DminTarea          ! c_null_ptr = keep derived
       real(c_double), dimension(*), intent(in) :: dxhy, dyhx
     end function cice_evp_hip_set_metrics

     integer(c_int) function cice_evp_hip_pin_host(ptr, bytes) bind(C, name='cice_evp_hip_pin_host')
       import :: c_int, c_int64_t, c_double
       real(c_double), dimension(*), intent(in) :: ptr
       integer(c_int64_t), value :: bytes
     end function cice_evp_hip_pin_host

     integer(c_int) function cice_evp_hip_finalize() bind(C, name='cice_evp_hip_finalize')
       import :: c_int
     end function cice_evp_hip_finalize

     integer(c_int) function cice_evp_hip_last_error(buf, buflen) bind(C, name='cice_evp_hip_last_error')
       import :: c_int, c_int32_t, c_char
       character(kind=c_char), dimension(*), intent(out) :: buf
       integer(c_int32_t), value :: buflen
     end function cice_evp_hip_last_error

     integer(c_int) function cice_evp_hip_comm_unique_id(id128) bind(C, name='cice_evp_hip_comm_unique_id')
       import :: c_int, c_int32_t
       integer(c_int32_t), dimension(32), intent(out) :: id128
     end function cice_evp_hip_comm_unique_id

     integer(c_int) function cice_evp_hip_comm_init(id128) bind(C, name='cice_evp_hip_comm_init')
       import :: c_int, c_int32_t
       integer(c_int32_t), dimension(32), intent(in) :: id128
     end function cice_evp_hip_comm_init

     integer(c_int) function cice_evp_hip_halo_export(blob) bind(C, name='cice_evp_hip_halo_export')
       import :: c_int, c_int32_t
       integer(c_int32_t), intent(out) :: blob(*)
     end function cice_evp_hip_halo_export

     integer(c_int) function cice_evp_hip_halo_import(blobs, nranks) bind(C, name='cice_evp_hip_halo_import')
       import :: c_int, c_int32_t
       integer(c_int32_t), intent(in) :: blobs(*)
       integer(c_int32_t), value :: nranks
     end function cice_evp_hip_halo_import

     ! ---- wider entry points (Option A: evp() patched to hand over its whole B-grid body) ----
     integer(c_int) function cice_evp_hip_set_prep_geometry(tmask, umask, hm, tarea, uarea, fcor_blk) &
          bind(C, name='cice_evp_hip_set_prep_geometry')
       import :: c_int, c_int32_t, c_double
       integer(c_int32_t), dimension(*), intent(in) :: tmask, umask       ! logical(4) storage
       real(c_double), dimension(*), intent(in) :: hm, tarea, uarea, fcor_blk
     end function cice_evp_hip_set_prep_geometry

     integer(c_int) function cice_evp_hip_prep(pp, tfields11, fields32, iceTmask, iceUmask, &
          strintxU, strintyU, strocnxU, strocnyU) bind(C, name='cice_evp_hip_prep')
       import :: c_int, c_int32_t, c_double, c_ptr, cice_evp_hip_prep_params
       type(cice_evp_hip_prep_params), intent(in) :: pp
       type(c_ptr), dimension(11), intent(in) :: tfields11
       type(c_ptr), dimension(32), intent(in) :: fields32
       integer(c_int32_t), dimension(*), intent(inout) :: iceTmask, iceUmask
       real(c_double), dimension(*), intent(inout) :: strintxU, strintyU, strocnxU, strocnyU
     end function cice_evp_hip_prep

     type(c_ptr) function cice_evp_hip_addr(array) bind(C, name='cice_evp_hip_addr')
       import :: c_ptr
       type(*), dimension(*), intent(in) :: array      ! any contiguous array; no TARGET needed
     end function cice_evp_hip_addr

     integer(c_int) function cice_evp_hip_set_strength(strength) bind(C, name='cice_evp_hip_set_strength')
       import :: c_int, c_double
       real(c_double), dimension(*), intent(in) :: strength
     end function cice_evp_hip_set_strength

     integer(c_int) function cice_evp_hip_set_tbu(TbU) bind(C, name='cice_evp_hip_set_tbu')
       import :: c_int, c_double
       real(c_double), dimension(*), intent(in) :: TbU
     end function cice_evp_hip_set_tbu

     integer(c_int) function cice_evp_hip_subcycle(ndte) bind(C, name='cice_evp_hip_subcycle')
       import :: c_int, c_int32_t
       integer(c_int32_t), value :: ndte
     end function cice_evp_hip_subcycle

     integer(c_int) function cice_evp_hip_stress_halo_available() bind(C, name='cice_evp_hip_stress_halo_available')
       import :: c_int
     end function cice_evp_hip_stress_halo_available
     integer(c_int) function cice_evp_hip_stress_halo() bind(C, name='cice_evp_hip_stress_halo')
       import :: c_int
     end function cice_evp_hip_stress_halo

     integer(c_int) function cice_evp_hip_halo_mask(halomask) bind(C, name='cice_evp_hip_halo_mask')
       import :: c_int, c_int32_t
       integer(c_int32_t), dimension(*), intent(in) :: halomask
     end function cice_evp_hip_halo_mask

     integer(c_int) function cice_evp_hip_seam_fin_plan(counts2, dst, a, b, coef) bind(C, name='cice_evp_hip_seam_fin_plan')
       import :: c_int, c_int32_t, c_ptr
       integer(c_int32_t), dimension(2), intent(out) :: counts2
       type(c_ptr), value :: dst, a, b, coef                          ! c_null_ptr: counts only
     end function cice_evp_hip_seam_fin_plan

     integer(c_int) function cice_evp_hip_set_option(key, val) bind(C, name='cice_evp_hip_set_option')
       import :: c_int, c_int32_t
       integer(c_int32_t), value :: key, val
     end function cice_evp_hip_set_option

     integer(c_int) function cice_evp_hip_fetch_stresses(sig12) bind(C, name='cice_evp_hip_fetch_stresses')
       import :: c_int, c_ptr
       type(c_ptr), dimension(12), intent(in) :: sig12
     end function cice_evp_hip_fetch_stresses

     integer(c_int) function cice_evp_hip_invalidate_stresses() bind(C, name='cice_evp_hip_invalidate_stresses')
       import :: c_int
     end function cice_evp_hip_invalidate_stresses

     integer(c_int) function cice_evp_hip_pin_ptr(ptr, bytes) bind(C, name='cice_evp_hip_pin_host')
       import :: c_int, c_int64_t, c_ptr
       type(c_ptr), value :: ptr
       integer(c_int64_t), value :: bytes
     end function cice_evp_hip_pin_ptr

     integer(c_int) function cice_evp_hip_cgrid_set_geometry(static23) bind(C, name='cice_evp_hip_cgrid_set_geometry')
       import :: c_int, c_ptr
       type(c_ptr), dimension(23), intent(in) :: static23
     end function cice_evp_hip_cgrid_set_geometry

     integer(c_int) function cice_evp_hip_cgrid_run(ndte, visc_method, fields19, inputs23, iceTmask, iceUmask, &
                                                    iceEmask, iceNmask) bind(C, name='cice_evp_hip_cgrid_run')
       import :: c_int, c_int32_t, c_ptr
       integer(c_int32_t), value :: ndte, visc_method
       type(c_ptr), dimension(19), intent(in) :: fields19
       type(c_ptr), dimension(23), intent(in) :: inputs23
       integer(c_int32_t), dimension(*), intent(in) :: iceTmask, iceUmask, iceEmask, iceNmask
     end function cice_evp_hip_cgrid_run

     integer(c_int) function cice_evp_hip_cgrid_deformations(tarear, divu, shear, vort, rdg_conv, rdg_shear) &
          bind(C, name='cice_evp_hip_cgrid_deformations')
       import :: c_int, c_double
       real(c_double), dimension(*), intent(in) :: tarear
       real(c_double), dimension(*), intent(inout) :: divu, shear, vort, rdg_conv, rdg_shear
     end function cice_evp_hip_cgrid_deformations
     integer(c_int) function cice_evp_hip_cgrid_dyn_finish(strocnxN, strocnyN, strocnxE, strocnyE) &
          bind(C, name='cice_evp_hip_cgrid_dyn_finish')
       import :: c_int, c_double
       real(c_double), intent(inout) :: strocnxN(*), strocnyN(*), strocnxE(*), strocnyE(*)
     end function cice_evp_hip_cgrid_dyn_finish

     integer(c_int) function cice_evp_hip_describe_path(buf, n) bind(C, name='cice_evp_hip_describe_path')
       import :: c_int, c_int32_t, c_char
       character(kind=c_char), dimension(*), intent(out) :: buf
       integer(c_int32_t), value :: n
     end function cice_evp_hip_describe_path

     integer(c_int) function cice_evp_hip_cgrid_set_prep_geometry(tmask, umaskCD, emask, nmask, fcor_blk, fcorE_blk, &
                                                                  fcorN_blk) bind(C, name='cice_evp_hip_cgrid_set_prep_geometry')
       import :: c_int, c_int32_t, c_double
       integer(c_int32_t), dimension(*), intent(in) :: tmask, umaskCD, emask, nmask
       real(c_double), dimension(*), intent(in) :: fcor_blk, fcorE_blk, fcorN_blk
     end function cice_evp_hip_cgrid_set_prep_geometry

     integer(c_int) function cice_evp_hip_cgrid_prep(pp, tfields11, state12, iceTmask, iceUmask, iceEmask, iceNmask) &
          bind(C, name='cice_evp_hip_cgrid_prep')
       import :: c_int, c_int32_t, c_ptr, cice_evp_hip_prep_params
       type(cice_evp_hip_prep_params), intent(in) :: pp
       type(c_ptr), dimension(11), intent(in) :: tfields11
       type(c_ptr), dimension(12), intent(in) :: state12
       integer(c_int32_t), dimension(*), intent(inout) :: iceTmask, iceUmask, iceEmask, iceNmask
     end function cice_evp_hip_cgrid_prep

     integer(c_int) function cice_evp_hip_cgrid_set_tb(TbE, TbN) bind(C, name='cice_evp_hip_cgrid_set_tb')
       import :: c_int, c_double
       real(c_double), dimension(*), intent(in) :: TbE, TbN
     end function cice_evp_hip_cgrid_set_tb

     integer(c_int) function cice_evp_hip_cgrid_prep_finish(strength, visc_method) bind(C, name='cice_evp_hip_cgrid_prep_finish')
       import :: c_int, c_int32_t, c_double
       real(c_double), dimension(*), intent(in) :: strength
       integer(c_int32_t), value :: visc_method
     end function cice_evp_hip_cgrid_prep_finish

     integer(c_int) function cice_evp_hip_cgrid_subcycle(ndte) bind(C, name='cice_evp_hip_cgrid_subcycle')
       import :: c_int, c_int32_t
       integer(c_int32_t), value :: ndte
     end function cice_evp_hip_cgrid_subcycle

     integer(c_int) function cice_evp_hip_cgrid_download(fields19) bind(C, name='cice_evp_hip_cgrid_download')
       import :: c_int, c_ptr
       type(c_ptr), dimension(19), intent(in) :: fields19
     end function cice_evp_hip_cgrid_download

     integer(c_int) function cice_evp_hip_cgrid_fetch(table, idx, dst) bind(C, name='cice_evp_hip_cgrid_fetch')
       import :: c_int, c_int32_t, c_double
       integer(c_int32_t), value :: table, idx
       real(c_double), dimension(*), intent(inout) :: dst
     end function cice_evp_hip_cgrid_fetch

     integer(c_int) function cice_evp_hip_download(fields32) bind(C, name='cice_evp_hip_download')
       import :: c_int, c_ptr
       type(c_ptr), dimension(32), intent(in) :: fields32
     end function cice_evp_hip_download
  end interface

  logical :: initialised = .false.
  ! this task holds no blocks (the reference's distributions allow it): it takes part in the bootstrap among the ranks and
  ! nothing else -- the library keeps it as a bystander, every routine below returns at once
  logical :: empty_rank = .false.
  logical :: pinned = .false.
  ! Default = dyn_evp1d_run's contract (ice_dyn_evp1d.F90:121-135): the 12 intent(inout) stress arrays are
  ! copied in and written back on every call, so an UNPATCHED host (restart write ice_restart_driver.F90:187-200,
  ! history sig1/sig2/sigP, evp()'s own ice_HaloUpdate_stress on tripole grids) always reads current values.
  ! Opt-in only: a host that has installed the two hooks (dyn_evp_hip_fetch_stresses before every reader,
  ! dyn_evp_hip_invalidate_stresses after every writer) may call dyn_evp_hip_keep_stresses_resident(.true.)
  ! -- or set CICE_EVP_HIP_STRESS_RESIDENT=1 -- and the stresses then stay on the device between calls.
  integer, parameter :: halo_blob_words = 256   ! CICE_EVP_HIP_HALO_BLOB (1024 bytes) as 4-byte words
  logical :: stress_resident = .false.
  logical :: stress_resident_requested = .false.
  logical :: body_sig_on_device = .false.   ! Option A with resident stresses: the device copy is current (dyn_evp_hip_evp_body)
  logical :: on_tripole = .false.
  logical :: cgrid_geometry_set = .false.
  logical :: cgrid_pinned = .false.
  logical :: cgrid_prep_geometry_set = .false.

contains

!-----------------------------------------------------------------------
  subroutine check(rc, subname, file, line)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: subname, file
    integer, intent(in) :: line
    character(kind=c_char), dimension(1024) :: cbuf
    character(len=1024) :: msg
    integer :: n, k
    if (rc == 0) return
    n = cice_evp_hip_last_error(cbuf, 1024_c_int32_t)
    msg = ' '
    do k = 1, min(n, 1023)
       msg(k:k) = cbuf(k)
    enddo
    call abort_ice(subname//' ERROR: '//trim(msg), file=file, line=line)
  end subroutine check

!-----------------------------------------------------------------------
! Replaces dyn_evp1d_init.  Everything it needs is public module data:
! block geometry (ice_blocks, ice_domain), static grid (ice_grid), EVP
! scalars (ice_dyn_shared, set by set_evp_parameters), rhow (Icepack).
  subroutine dyn_evp_hip_init

    use ice_blocks, only: nx_block, ny_block, nghost, block, get_block, nblocks_tot, &
        get_block_parameter
    use ice_communicate, only: my_task, master_task, get_num_procs
    use ice_broadcast, only: broadcast_array
    use ice_domain, only: nblocks, blocks_ice, distrb_info, ew_boundary_type, ns_boundary_type
    use ice_domain_size, only: max_blocks, nx_global, ny_global
    use ice_grid, only: HTE, HTN, dxT, dyT, uarear, tarea
    use ice_dyn_shared, only: ndte, arlx1i, denom1, brlx, revp, e_factor, epp2i, capping, &
        Ktens, deltaminEVP, u0, cosw, sinw, dxhy, dyhx
    use ice_fileunits, only: nu_diag
    use icepack_intfc, only: icepack_query_parameters, icepack_warnings_flush, &
        icepack_warnings_aborted

    type(cice_evp_hip_dims) :: d
    type(cice_evp_hip_params) :: p
    type(block) :: tb
    integer(c_int32_t), allocatable, target, save :: ilo(:), ihi(:), jlo(:), jhi(:), ig0(:), jg0(:), &
        gi0(:), gj0(:), gnx(:), gny(:), gown(:), gloc(:)
    integer(int_kind), pointer :: i_glob(:), j_glob(:)
    integer(int_kind) :: n, lo_i, hi_i, lo_j, hi_j, nprocs
    integer(c_int32_t) :: uid(32)
    integer(int_kind) :: blob(halo_blob_words), r
    integer(int_kind), allocatable :: blobs(:,:)
    real(dbl_kind) :: rhow
    character(len=16) :: envval
    integer :: envlen, envstat
    integer(c_int32_t) :: cnt2(2)
    character(len=*), parameter :: subname = '(dyn_evp_hip_init)'

    call icepack_query_parameters(rhow_out=rhow)
    call icepack_warnings_flush(nu_diag)
    if (icepack_warnings_aborted()) call abort_ice(error_message=subname, file=__FILE__, line=__LINE__)

    nprocs = get_num_procs()
    if (allocated(ilo)) deallocate(ilo, ihi, jlo, jhi, ig0, jg0, gi0, gj0, gnx, gny, gown, gloc)
    allocate(ilo(max(nblocks,1)), ihi(max(nblocks,1)), jlo(max(nblocks,1)), jhi(max(nblocks,1)), &
             ig0(max(nblocks,1)), jg0(max(nblocks,1)))      ! (a task without blocks still hands over valid addresses)
    do n = 1, nblocks
       tb = get_block(blocks_ice(n), n)
       ilo(n) = tb%ilo; ihi(n) = tb%ihi; jlo(n) = tb%jlo; jhi(n) = tb%jhi
       ig0(n) = tb%i_glob(tb%ilo); jg0(n) = tb%j_glob(tb%jlo)
    enddo
    ! global block table: who owns which rectangle of the global index space
    allocate(gi0(nblocks_tot), gj0(nblocks_tot), gnx(nblocks_tot), gny(nblocks_tot), &
             gown(nblocks_tot), gloc(nblocks_tot))
    do n = 1, nblocks_tot
       call get_block_parameter(n, ilo=lo_i, ihi=hi_i, jlo=lo_j, jhi=hi_j, i_glob=i_glob, j_glob=j_glob)
       gi0(n) = i_glob(lo_i); gj0(n) = j_glob(lo_j)
       gnx(n) = hi_i - lo_i + 1; gny(n) = hi_j - lo_j + 1
       gown(n) = distrb_info%blockLocation(n) - 1      ! 0 -> -1: eliminated land block
       gloc(n) = distrb_info%blockLocalID(n) - 1
    enddo

    d%nx_block = nx_block; d%ny_block = ny_block; d%nblocks = nblocks; d%max_blocks = max_blocks
    d%nghost = nghost; d%nx_global = nx_global; d%ny_global = ny_global
    d%ew_boundary_type = bnd_code(ew_boundary_type); d%ns_boundary_type = bnd_code(ns_boundary_type)
    d%rank = my_task; d%nranks = nprocs
    d%ilo = c_loc(ilo); d%ihi = c_loc(ihi); d%jlo = c_loc(jlo); d%jhi = c_loc(jhi)
    d%iglob0 = c_loc(ig0); d%jglob0 = c_loc(jg0)
    d%nblocks_tot = nblocks_tot
    d%gi0 = c_loc(gi0); d%gj0 = c_loc(gj0); d%gnx = c_loc(gnx); d%gny = c_loc(gny)
    d%gowner = c_loc(gown); d%glocal = c_loc(gloc)

    p%ndte = ndte; p%strict = 1
    p%arlx1i = arlx1i; p%denom1 = denom1; p%brlx = brlx; p%revp = revp
    p%e_factor = e_factor; p%epp2i = epp2i; p%capping = capping; p%Ktens = Ktens
    p%deltaminEVP = deltaminEVP; p%u0 = u0; p%cosw = cosw; p%sinw = sinw; p%rhow = rhow

    call check(cice_evp_hip_init(d, p, HTE, HTN, dxT, dyT, uarear, tarea), subname, __FILE__, __LINE__)
    empty_rank = nblocks == 0
    if (.not. empty_rank .and. (trim(ns_boundary_type) == 'tripole' .or. trim(ns_boundary_type) == 'tripoleT')) then
       ! the north ghost row of dxhy/dyhx is a mirrored interior value (halo update with
       ! sign, ice_dyn_shared.F90:412-417): hand over CICE's own arrays
       call check(cice_evp_hip_set_metrics(c_null_ptr, c_null_ptr, c_null_ptr, c_null_ptr, &
            dxhy, dyhx, c_null_ptr), subname, __FILE__, __LINE__)
    endif

    call get_environment_variable('CICE_EVP_HIP_STRESS_RESIDENT', envval, envlen, envstat)
    ! (every rank, also one without blocks: on tripoleT grids the verdict is a reduction over all tasks of the distribution)
    call settle_stress_residency(stress_resident_requested .or. (envstat == 0 .and. envlen > 0 .and. envval(1:1) == '1'), &
         .false.)

    if (nprocs > 1) then
       call get_environment_variable('CICE_EVP_HIP_BOOTSTRAP', envval, envlen, envstat)
       if (envstat == 0 .and. envlen >= 5 .and. envval(1:5) == 'blobs') then
          ! bootstrap without RCCL (ranks of one node; also several ranks sharing a GPU, which RCCL refuses): every
          ! rank exports its HIP-IPC handles, CICE's own broadcast_array ships them -- one broadcast per rank, an
          ! all-gather in rank order -- and every rank imports all of them (collective: runs the probe exchange)
          allocate(blobs(halo_blob_words, nprocs))
          do r = 0, nprocs - 1
             blob = 0
             if (my_task == r) call check(cice_evp_hip_halo_export(blob), subname, __FILE__, __LINE__)
             call broadcast_array(blob, r)
             blobs(:, r + 1) = blob
          enddo
          call check(cice_evp_hip_halo_import(blobs, int(nprocs, c_int32_t)), subname, __FILE__, __LINE__)
          deallocate(blobs)
       else
          ! RCCL bootstrap: the master's ncclUniqueId travels over CICE's own broadcast
          uid = 0
          if (my_task == master_task) &
             call check(cice_evp_hip_comm_unique_id(uid), subname, __FILE__, __LINE__)
          call broadcast_array(uid, master_task)
          call check(cice_evp_hip_comm_init(uid), subname, __FILE__, __LINE__)
       endif
    endif
    initialised = .true.

  end subroutine dyn_evp_hip_init

!-----------------------------------------------------------------------
  integer(c_int32_t) function bnd_code(bnd)
    character(len=*), intent(in) :: bnd
    select case (trim(bnd))
    case ('closed');  bnd_code = 0
    case ('open');    bnd_code = 1
    case ('cyclic');  bnd_code = 2
    case ('tripole'); bnd_code = 3
    case ('tripoleT'); bnd_code = 4          ! T-fold: the loop on any rank layout; preparation / symmetrisation with the top row on one rank
    case default
       bnd_code = -1
       call abort_ice('(dyn_evp_hip_init) ERROR: unsupported boundary type '//trim(bnd), &
            file=__FILE__, line=__LINE__)
    end select
  end function bnd_code

!-----------------------------------------------------------------------
! Option A (INTEGRATION.md): the B-grid body of evp() from its entry to the end of the subcycle
! loop -- preparation phase (ice_dyn_evp.F90:383-840), loop (:859-913) and, on a tripole grid, the
! stress symmetrisation (:1321-1389) -- on the device.  The caller supplies the ice strength
! through a callback that runs between the two device phases, because icepack_ice_strength
! needs the iceTmask the preparation produces (:541-552).  Ranks whose T-grid halo needs another
! rank keep evp()'s host preparation (the C side refuses them).
  subroutine dyn_evp_hip_evp_body(dt, compute_strength)

    use ice_blocks, only: nx_block, ny_block, block, get_block
    use ice_domain_size, only: max_blocks
    use ice_domain, only: ns_boundary_type, nblocks, blocks_ice
    use ice_grid, only: tmask, umask, hm, tarea, uarea
    use ice_state, only: aice, vice, vsno, uvel, vvel, aice_init, strength, aicen, vicen
    use ice_arrays_column, only: Cdn_ocn
    use ice_flux, only: uocn, vocn, ss_tltx, ss_tlty, strairxT, strairyT, &
         stressp_1, stressp_2, stressp_3, stressp_4, stressm_1, stressm_2, stressm_3, stressm_4, &
         stress12_1, stress12_2, stress12_3, stress12_4, strintxU, strintyU, strocnxU, strocnyU, &
         taubxU, taubyU, TbU, hwater
    use ice_dyn_shared, only: ndte, fcor_blk, iceTmask, iceUmask, dyn_area_min, dyn_mass_min, ssh_stress, &
         seabed_stress, seabed_stress_method, seabed_stress_factor_LKD, seabed_stress_factor_prob
    use icepack_intfc, only: icepack_query_parameters

    real(kind=dbl_kind), intent(in) :: dt
    interface
       subroutine compute_strength()    ! fills ice_flux's strength from iceTmask and halo-updates it
       end subroutine compute_strength
    end interface

    type(cice_evp_hip_prep_params) :: pp
    type(c_ptr) :: tf(11), f32(32), out32(32), s12(12)
    logical :: sig_travel
    integer(c_int32_t), pointer :: tmask_i(:), umask_i(:), itm(:), ium(:)
    integer :: nall, iblk, i, j, nT, nU
    integer(int_kind), allocatable :: ixT(:), jxT(:), ixU(:), jxU(:)
    type(block) :: tb
    logical, save :: geometry_set = .false.
    character(len=*), parameter :: subname = '(dyn_evp_hip_evp_body)'

    if (.not. initialised) call abort_ice(subname//' ERROR: dyn_evp_hip_init not called', &
         file=__FILE__, line=__LINE__)
    if (empty_rank) then
       call compute_strength()      ! (whatever the host does there among the tasks happens on this one too)
       return
    endif
    ! (tripoleT: the library takes the preparation and the symmetrisation where the top row lies on one rank and says
    ! so otherwise -- cice_evp_hip_set_prep_geometry / cice_evp_hip_stress_halo return an error that check() reports)
    nall = nx_block*ny_block*max_blocks
    if (.not. geometry_set) then
       ! logical(log_kind) is a 4-byte logical: the C side tests "non-zero"
       call c_f_pointer(cice_evp_hip_addr(tmask), tmask_i, [nall])
       call c_f_pointer(cice_evp_hip_addr(umask), umask_i, [nall])
       call check(cice_evp_hip_set_prep_geometry(tmask_i, umask_i, hm, tarea, uarea, fcor_blk), &
            subname, __FILE__, __LINE__)
       geometry_set = .true.
    endif
    pp%dt = dt
    call icepack_query_parameters(rhoi_out=pp%rhoi, rhos_out=pp%rhos, gravit_out=pp%gravit)
    pp%dyn_area_min = dyn_area_min
    pp%dyn_mass_min = dyn_mass_min
    pp%ssh_stress_coupled = merge(1_c_int32_t, 0_c_int32_t, trim(ssh_stress) == 'coupled')

    tf(1) = cice_evp_hip_addr(aice);      tf(2) = cice_evp_hip_addr(vice);     tf(3) = cice_evp_hip_addr(vsno)
    tf(4) = cice_evp_hip_addr(aice_init); tf(5) = cice_evp_hip_addr(Cdn_ocn)
    tf(6) = cice_evp_hip_addr(uocn);      tf(7) = cice_evp_hip_addr(vocn)
    tf(8) = cice_evp_hip_addr(ss_tltx);   tf(9) = cice_evp_hip_addr(ss_tlty)
    tf(10) = cice_evp_hip_addr(strairxT); tf(11) = cice_evp_hip_addr(strairyT)
    f32 = c_null_ptr
    ! the 12 stresses travel in and out at every call (CICE's arrays current after every evp()) -- unless the host opted in to
    ! device-resident stresses (dyn_evp_hip_keep_stresses_resident) and the device copy is current: then they stay where
    ! evp(), their only writer, left them (cice_evp_hip_prep with NULL stress entries), 24 array transfers less per call
    sig_travel = .not. (stress_resident .and. body_sig_on_device)
    s12(1) = cice_evp_hip_addr(stressp_1);  s12(2) = cice_evp_hip_addr(stressp_2)
    s12(3) = cice_evp_hip_addr(stressp_3);  s12(4) = cice_evp_hip_addr(stressp_4)
    s12(5) = cice_evp_hip_addr(stressm_1);  s12(6) = cice_evp_hip_addr(stressm_2)
    s12(7) = cice_evp_hip_addr(stressm_3);  s12(8) = cice_evp_hip_addr(stressm_4)
    s12(9) = cice_evp_hip_addr(stress12_1); s12(10) = cice_evp_hip_addr(stress12_2)
    s12(11) = cice_evp_hip_addr(stress12_3); s12(12) = cice_evp_hip_addr(stress12_4)
    if (sig_travel) f32(1:12) = s12
    ! TbU does not travel here: dyn_prep2 zeroes it and the seabed stress factor is computed from the
    ! NEW iceUmask afterwards (ice_dyn_evp.F90:770-826) -- below, once the preparation has returned
    f32(29) = cice_evp_hip_addr(uvel)
    f32(30) = cice_evp_hip_addr(vvel)
    call c_f_pointer(cice_evp_hip_addr(iceTmask), itm, [nall])
    call c_f_pointer(cice_evp_hip_addr(iceUmask), ium, [nall])
    call check(cice_evp_hip_prep(pp, tf, f32, itm, ium, strintxU, strintyU, strocnxU, strocnyU), &
         subname, __FILE__, __LINE__)

    call compute_strength()
    call check(cice_evp_hip_set_strength(strength), subname, __FILE__, __LINE__)

    if (seabed_stress) then
       ! the reference's own routines on the host (exp(): libm-exact), from the masks the device
       ! preparation returned; index lists rebuilt as dyn_prep2 builds them (ice_dyn_shared.F90:740-770)
       allocate(ixT(nx_block*ny_block), jxT(nx_block*ny_block), ixU(nx_block*ny_block), jxU(nx_block*ny_block))
       do iblk = 1, nblocks
          tb = get_block(blocks_ice(iblk), iblk)
          TbU(:,:,iblk) = 0.0_dbl_kind
          nT = 0; nU = 0
          do j = tb%jlo, tb%jhi+1
          do i = tb%ilo, tb%ihi+1
             if (iceTmask(i,j,iblk)) then
                nT = nT + 1; ixT(nT) = i; jxT(nT) = j
             endif
          enddo
          enddo
          do j = tb%jlo, tb%jhi
          do i = tb%ilo, tb%ihi
             if (iceUmask(i,j,iblk)) then
                nU = nU + 1; ixU(nU) = i; jxU(nU) = j
             endif
          enddo
          enddo
          if (trim(seabed_stress_method) == 'LKD') then
             call seabed_stress_factor_LKD(nx_block, ny_block, nU, ixU, jxU, vice(:,:,iblk), aice(:,:,iblk), &
                                           hwater(:,:,iblk), TbU(:,:,iblk))
          elseif (trim(seabed_stress_method) == 'probabilistic') then
             call seabed_stress_factor_prob(nx_block, ny_block, nT, ixT, jxT, nU, ixU, jxU, &
                                            aicen(:,:,:,iblk), vicen(:,:,:,iblk), hwater(:,:,iblk), TbU(:,:,iblk))
          endif
       enddo
       deallocate(ixT, jxT, ixU, jxU)
       call check(cice_evp_hip_set_tbu(TbU), subname, __FILE__, __LINE__)
    endif
    call check(cice_evp_hip_subcycle(int(ndte, c_int32_t)), subname, __FILE__, __LINE__)
    if (trim(ns_boundary_type) == 'tripole' .or. trim(ns_boundary_type) == 'tripoleT') &
       call check(cice_evp_hip_stress_halo(), subname, __FILE__, __LINE__)

    out32 = c_null_ptr
    if (.not. stress_resident) out32(1:12) = s12
    body_sig_on_device = stress_resident
    out32(24) = cice_evp_hip_addr(strintxU); out32(25) = cice_evp_hip_addr(strintyU)
    out32(27) = cice_evp_hip_addr(taubxU);   out32(28) = cice_evp_hip_addr(taubyU)
    out32(29) = f32(29);                     out32(30) = f32(30)
    call check(cice_evp_hip_download(out32), subname, __FILE__, __LINE__)

  end subroutine dyn_evp_hip_evp_body

!-----------------------------------------------------------------------
! Replaces dyn_evp1d_run: identical argument list (ice_dyn_evp1d.F90:121-153).
! uvel_init/vvel_init (read for revised EVP only) and ndte are module data.
  subroutine dyn_evp_hip_run(L_stressp_1 , L_stressp_2 , L_stressp_3 , L_stressp_4 , &
                             L_stressm_1 , L_stressm_2 , L_stressm_3 , L_stressm_4 , &
                             L_stress12_1, L_stress12_2, L_stress12_3, L_stress12_4, &
                             L_strength,                                             &
                             L_cdn_ocn   , L_aiu       , L_uocn      , L_vocn      , &
                             L_waterxU   , L_wateryU   , L_forcexU   , L_forceyU   , &
                             L_umassdti  , L_fmU       , L_strintxU  , L_strintyU  , &
                             L_Tbu       , L_taubxU    , L_taubyU    , L_uvel      , &
                             L_vvel      , L_icetmask  , L_iceUmask)

    use ice_dyn_shared, only: ndte, uvel_init, vvel_init
    use ice_timers, only: ice_timer_start, ice_timer_stop, timer_evp1dcore
    use ice_domain, only: maskhalo_dyn, halo_info
    use ice_boundary, only: ice_HaloUpdate
    use ice_constants, only: field_loc_center, field_type_scalar
    use ice_communicate, only: get_num_procs

    real(kind=dbl_kind), dimension(:,:,:), intent(inout), contiguous, target :: &
      L_stressp_1 , L_stressp_2 , L_stressp_3 , L_stressp_4 ,  &
      L_stressm_1 , L_stressm_2 , L_stressm_3 , L_stressm_4 ,  &
      L_stress12_1, L_stress12_2, L_stress12_3, L_stress12_4,  &
      L_strintxU  , L_strintyU  , L_uvel      , L_vvel      ,  &
      L_taubxU    , L_taubyU
    real(kind=dbl_kind), dimension(:,:,:), intent(in), contiguous, target :: &
      L_strength  ,                                            &
      L_cdn_ocn   , L_aiu       , L_uocn     , L_vocn   ,      &
      L_waterxU   , L_wateryU   , L_forcexU  , L_forceyU,      &
      L_umassdti  , L_fmU       , L_Tbu
    logical(kind=log_kind), dimension(:,:,:), intent(in), contiguous, target :: &
      L_iceUmask  , L_iceTmask

    integer(c_int32_t), pointer :: tmask_i(:), umask_i(:)
    integer(int_kind), allocatable, save :: halomask(:,:,:)
    character(len=*), parameter :: subname = '(dyn_evp_hip_run)'

    if (.not. initialised) call abort_ice(subname//' ERROR: dyn_evp_hip_init not called', &
         file=__FILE__, line=__LINE__)
    if (empty_rank) return

    ! logical(log_kind) is a 4-byte Fortran logical (Icepack kinds): hand its storage
    ! to the C side, which tests "non-zero"
    call c_f_pointer(c_loc(L_iceTmask), tmask_i, [size(L_iceTmask)])
    call c_f_pointer(c_loc(L_iceUmask), umask_i, [size(L_iceUmask)])

    if (.not. pinned) then
       ! CICE's module arrays live for the whole run: page-lock them once so that the
       ! per-call H2D/D2H copies are direct DMA
       call pin_all()
       pinned = .true.
    endif

    if (maskhalo_dyn .and. get_num_procs() > 1) then
       ! the masked halo evp() builds for its own loop (ice_dyn_evp.F90:739-770), rebuilt here because the
       ! reference keeps halo_info_mask private: 1 where iceUmask, ghost cells updated, handed to the core
       if (.not. allocated(halomask)) allocate(halomask(size(L_iceUmask,1), size(L_iceUmask,2), size(L_iceUmask,3)))
       halomask = 0
       where (L_iceUmask) halomask = 1
       call ice_HaloUpdate(halomask, halo_info, field_loc_center, field_type_scalar)
       call check(cice_evp_hip_halo_mask(halomask), subname, __FILE__, __LINE__)
    endif

    call ice_timer_start(timer_evp1dcore)
    call check(cice_evp_hip_run( &
         L_stressp_1, L_stressp_2, L_stressp_3, L_stressp_4, L_stressm_1, L_stressm_2, L_stressm_3, &
         L_stressm_4, L_stress12_1, L_stress12_2, L_stress12_3, L_stress12_4, L_strength, L_cdn_ocn, &
         L_aiu, L_uocn, L_vocn, L_waterxU, L_wateryU, L_forcexU, L_forceyU, L_umassdti, L_fmU, &
         L_strintxU, L_strintyU, L_Tbu, L_taubxU, L_taubyU, L_uvel, L_vvel, uvel_init, vvel_init, &
         tmask_i, umask_i, int(ndte, c_int32_t)), subname, __FILE__, __LINE__)
    ! stresses resident on a tripole grid: evp()'s 12 x ice_HaloUpdate_stress after the loop (ice_dyn_evp.F90:1321-1389)
    ! act on the stale host arrays; the device copy gets the same symmetrisation here
    if (stress_resident .and. on_tripole) &
       call check(cice_evp_hip_stress_halo(), subname, __FILE__, __LINE__)
    call ice_timer_stop(timer_evp1dcore)
    call report_path_once()

  contains

    subroutine pin_one(a)
      real(kind=dbl_kind), dimension(:,:,:), intent(in), contiguous :: a
      integer(c_int) :: rc
      rc = cice_evp_hip_pin_host(a, int(size(a), c_int64_t) * 8_c_int64_t)   ! best effort: a failure only costs speed
    end subroutine pin_one

    subroutine pin_all()
      call pin_one(L_stressp_1);  call pin_one(L_stressp_2);  call pin_one(L_stressp_3);  call pin_one(L_stressp_4)
      call pin_one(L_stressm_1);  call pin_one(L_stressm_2);  call pin_one(L_stressm_3);  call pin_one(L_stressm_4)
      call pin_one(L_stress12_1); call pin_one(L_stress12_2); call pin_one(L_stress12_3); call pin_one(L_stress12_4)
      call pin_one(L_strength);   call pin_one(L_cdn_ocn);    call pin_one(L_aiu);        call pin_one(L_uocn)
      call pin_one(L_vocn);       call pin_one(L_waterxU);    call pin_one(L_wateryU);    call pin_one(L_forcexU)
      call pin_one(L_forceyU);    call pin_one(L_umassdti);   call pin_one(L_fmU);        call pin_one(L_strintxU)
      call pin_one(L_strintyU);   call pin_one(L_Tbu);        call pin_one(L_taubxU);     call pin_one(L_taubyU)
      call pin_one(L_uvel);       call pin_one(L_vvel)
    end subroutine pin_all

  end subroutine dyn_evp_hip_run

!-----------------------------------------------------------------------
! C grid (grid_ice = 'C'): replaces the subcycle loop of evp(), ice_dyn_evp.F90:938-1099 -- everything between
! "do ksub = 1,ndte" and its "enddo" (strain_rates_U .. the last dyn_haloUpdate of uvel, vvel).  The loop's
! operands are private module arrays of ice_dyn_evp, so the call is made from inside evp() (INTEGRATION.md has the
! patch); ice_grid / ice_dyn_shared / ice_flux / ice_state data are taken from their modules here.
! The arguments are exactly the private arrays the loop touches.  evp() continues with deformationsC_T and its own
! halo update of strintxE / strintyN as before.
  subroutine dyn_evp_hip_cgrid_run(uocnE, vocnE, cdn_ocnE, waterxE, forcexE, aiE, rheofactE, emassdti, &
                                   uocnN, vocnN, cdn_ocnN, wateryN, forceyN, aiN, rheofactN, nmassdti, &
                                   ratiodxN, ratiodxNr, ratiodyE, ratiodyEr,                           &
                                   zetax2T, etax2T, etax2U, shearU, deltaU)

    use ice_dyn_shared, only: ndte, visc_method, DminTarea, uvelE_init, vvelN_init, &
                              iceTmask, iceUmask, iceEmask, iceNmask
    use ice_grid, only: dxT, dyT, dxU, dyU, dxE, dyE, dxN, dyN, uarea, tarea, earea, narea, earear, narear, &
                        epm, npm, uvm, hm
    use ice_state, only: uvel, vvel, uvelE, vvelE, uvelN, vvelN, strength
    use ice_flux, only: stresspT, stressmT, stress12T, stress12U, strintxE, strintyN, taubxE, taubyN, &
                        fmE, fmN, TbE, TbN
    use ice_timers, only: ice_timer_start, ice_timer_stop, timer_evp1dcore
    use ice_blocks, only: nx_block, ny_block, block, get_block
    use ice_domain, only: maskhalo_dyn, halo_info, nblocks, blocks_ice
    use ice_domain_size, only: max_blocks
    use ice_boundary, only: ice_HaloUpdate
    use ice_constants, only: field_loc_center, field_type_scalar
    use ice_communicate, only: get_num_procs

    real(kind=dbl_kind), dimension(:,:,:), intent(in), contiguous, target :: &
      uocnE, vocnE, cdn_ocnE, waterxE, forcexE, aiE, rheofactE, emassdti, &
      uocnN, vocnN, cdn_ocnN, wateryN, forceyN, aiN, rheofactN, nmassdti, &
      ratiodxN, ratiodxNr, ratiodyE, ratiodyEr
    real(kind=dbl_kind), dimension(:,:,:), intent(inout), contiguous, target :: &
      zetax2T, etax2T, etax2U, shearU, deltaU

    type(c_ptr) :: fl(19), inp(23)
    integer(c_int32_t), pointer :: mT(:), mU(:), mE(:), mN(:)
    integer(c_int32_t) :: vm
    integer(c_int) :: rc
    integer(c_int64_t) :: nbytes
    integer :: k
    character(len=*), parameter :: subname = '(dyn_evp_hip_cgrid_run)'

    if (.not. initialised) call abort_ice(subname//' ERROR: dyn_evp_hip_init not called', &
         file=__FILE__, line=__LINE__)
    if (empty_rank) return
    call cgrid_ensure_geometry(ratiodxN, ratiodxNr, ratiodyE, ratiodyEr)
    vm = cgrid_visc_method()
    fl = [cice_evp_hip_addr(uvelE), cice_evp_hip_addr(vvelE), cice_evp_hip_addr(uvelN), cice_evp_hip_addr(vvelN), &
          cice_evp_hip_addr(uvel), cice_evp_hip_addr(vvel), cice_evp_hip_addr(stresspT), cice_evp_hip_addr(stressmT), &
          cice_evp_hip_addr(stress12T), cice_evp_hip_addr(stress12U), cice_evp_hip_addr(strintxE), &
          cice_evp_hip_addr(strintyN), cice_evp_hip_addr(taubxE), cice_evp_hip_addr(taubyN), &
          c_loc(zetax2T), c_loc(etax2T), c_loc(etax2U), c_loc(shearU), c_loc(deltaU)]
    inp = [cice_evp_hip_addr(strength), c_loc(cdn_ocnE), c_loc(aiE), c_loc(uocnE), c_loc(vocnE), c_loc(waterxE), &
           c_loc(forcexE), c_loc(emassdti), cice_evp_hip_addr(fmE), cice_evp_hip_addr(uvelE_init), &
           cice_evp_hip_addr(TbE), c_loc(rheofactE), c_loc(cdn_ocnN), c_loc(aiN), c_loc(uocnN), c_loc(vocnN), &
           c_loc(wateryN), c_loc(forceyN), c_loc(nmassdti), cice_evp_hip_addr(fmN), cice_evp_hip_addr(vvelN_init), &
           cice_evp_hip_addr(TbN), c_loc(rheofactN)]
    call c_f_pointer(cice_evp_hip_addr(iceTmask), mT, [size(iceTmask)])
    call c_f_pointer(cice_evp_hip_addr(iceUmask), mU, [size(iceUmask)])
    call c_f_pointer(cice_evp_hip_addr(iceEmask), mE, [size(iceEmask)])
    call c_f_pointer(cice_evp_hip_addr(iceNmask), mN, [size(iceNmask)])
    if (.not. cgrid_pinned) then
       ! the module arrays live for the whole run: page-lock them once (best effort: a failure only costs speed),
       ! so that the per-call copies are one gather and one scatter launch over PCIe
       nbytes = int(size(uvelE), c_int64_t) * 8_c_int64_t
       do k = 1, 19
          rc = cice_evp_hip_pin_ptr(fl(k), nbytes)
       enddo
       do k = 1, 23
          rc = cice_evp_hip_pin_ptr(inp(k), nbytes)
       enddo
       cgrid_pinned = .true.
    endif
    call cgrid_halo_mask()
    call ice_timer_start(timer_evp1dcore)
    call check(cice_evp_hip_cgrid_run(int(ndte, c_int32_t), vm, fl, inp, mT, mU, mE, mN), subname, __FILE__, __LINE__)
    call ice_timer_stop(timer_evp1dcore)

  end subroutine dyn_evp_hip_cgrid_run

!-----------------------------------------------------------------------
! With CICE_EVP_HIP_VERBOSE set: one line per rank, after the first call, on which kernel and halo transport the library
! settled on (cice_evp_hip_describe_path) -- on the diagnostics unit of this rank's stdout
  subroutine report_path_once()
    use ice_communicate, only: my_task
    logical, save :: done = .false.
    character(kind=c_char), dimension(600) :: cbuf
    character(len=600) :: line
    character(len=8) :: envval
    integer :: envlen, envstat, k
    if (done) return
    done = .true.
    call get_environment_variable('CICE_EVP_HIP_VERBOSE', envval, envlen, envstat)
    if (envstat /= 0 .or. envlen < 1) return
    if (cice_evp_hip_describe_path(cbuf, 600_c_int32_t) /= 0) return
    line = ' '
    do k = 1, 600
       if (cbuf(k) == c_null_char) exit
       line(k:k) = cbuf(k)
    enddo
    write(*,'(a,i0,2a)') '(dyn_evp_hip) task ', my_task, ': ', trim(line)
  end subroutine report_path_once

!-----------------------------------------------------------------------
! C-grid helpers: static arrays handed over once; visc_method as the C ABI's code; evp()'s masked halo for the loop
  subroutine cgrid_ensure_geometry(ratiodxN, ratiodxNr, ratiodyE, ratiodyEr)
    use ice_dyn_shared, only: DminTarea
    use ice_grid, only: dxT, dyT, dxU, dyU, dxE, dyE, dxN, dyN, uarea, tarea, earea, narea, earear, narear, &
                        epm, npm, uvm, hm
    real(kind=dbl_kind), dimension(:,:,:), intent(in), contiguous, target :: ratiodxN, ratiodxNr, ratiodyE, ratiodyEr
    type(c_ptr) :: st(23)
    character(len=*), parameter :: subname = '(dyn_evp_hip cgrid_ensure_geometry)'
    if (cgrid_geometry_set) return
    st = [cice_evp_hip_addr(dxT), cice_evp_hip_addr(dyT), cice_evp_hip_addr(dxU), cice_evp_hip_addr(dyU), &
          cice_evp_hip_addr(dxE), cice_evp_hip_addr(dyE), cice_evp_hip_addr(dxN), cice_evp_hip_addr(dyN), &
          cice_evp_hip_addr(uarea), cice_evp_hip_addr(tarea), cice_evp_hip_addr(earea), cice_evp_hip_addr(narea), &
          cice_evp_hip_addr(earear), cice_evp_hip_addr(narear), cice_evp_hip_addr(epm), cice_evp_hip_addr(npm), &
          cice_evp_hip_addr(uvm), cice_evp_hip_addr(hm), cice_evp_hip_addr(DminTarea), &
          c_loc(ratiodxN), c_loc(ratiodxNr), c_loc(ratiodyE), c_loc(ratiodyEr)]
    call check(cice_evp_hip_cgrid_set_geometry(st), subname, __FILE__, __LINE__)
    cgrid_geometry_set = .true.
  end subroutine cgrid_ensure_geometry

  integer(c_int32_t) function cgrid_visc_method() result(vm)
    use ice_dyn_shared, only: visc_method
    character(len=*), parameter :: subname = '(dyn_evp_hip cgrid_visc_method)'
    vm = 0
    if (trim(visc_method) == 'avg_strength') then
       vm = 1
    elseif (trim(visc_method) /= 'avg_zeta') then
       call abort_ice(subname//' ERROR: unknown visc_method '//trim(visc_method), file=__FILE__, line=__LINE__)
    endif
  end function cgrid_visc_method

  subroutine cgrid_halo_mask()
    use ice_dyn_shared, only: iceTmask
    use ice_blocks, only: nx_block, ny_block, block, get_block
    use ice_domain, only: maskhalo_dyn, halo_info, nblocks, blocks_ice
    use ice_domain_size, only: max_blocks
    use ice_boundary, only: ice_HaloUpdate
    use ice_constants, only: field_loc_center, field_type_scalar
    use ice_communicate, only: get_num_procs
    integer(int_kind), allocatable, save :: halomask_c(:,:,:)
    type(block) :: tb
    integer :: i, j, iblk
    character(len=*), parameter :: subname = '(dyn_evp_hip cgrid_halo_mask)'
    if (.not. (maskhalo_dyn .and. get_num_procs() > 1)) return
    ! the masked halo evp() builds for the C-grid loop (ice_dyn_evp.F90:739-770: a cell and its four neighbours,
    ! where iceTmask; ghost cells updated), rebuilt here because halo_info_mask is private to ice_dyn_evp
    if (.not. allocated(halomask_c)) allocate(halomask_c(nx_block, ny_block, max_blocks))
    halomask_c = 0
    do iblk = 1, nblocks
       tb = get_block(blocks_ice(iblk), iblk)
       do j = tb%jlo, tb%jhi
       do i = tb%ilo, tb%ihi
          if (iceTmask(i,j,iblk) .or. iceTmask(i-1,j,iblk) .or. iceTmask(i+1,j,iblk) .or. &
              iceTmask(i,j-1,iblk) .or. iceTmask(i,j+1,iblk)) halomask_c(i,j,iblk) = 1
       enddo
       enddo
    enddo
    call ice_HaloUpdate(halomask_c, halo_info, field_loc_center, field_type_scalar)
    call check(cice_evp_hip_halo_mask(halomask_c), subname, __FILE__, __LINE__)
  end subroutine cgrid_halo_mask

!-----------------------------------------------------------------------
! C grid, wider (the C-grid counterpart of dyn_evp_hip_evp_body): replaces evp()'s preparation AND its loop for
! grid_ice = 'C' -- ice_dyn_evp.F90:372-735 (the zeroing of the deformation diagnostics, dyn_prep1, the T-grid halo
! updates, the T -> U / E / N averages, dyn_prep2 at U, N and E points, the velocity averages and exchanges), 770-840
! (seabed stress factors, by the reference's own routines on the host: libm exp()) and 938-1099 (the loop).  Public
! module data is taken from its modules; the arguments are the private arrays of ice_dyn_evp the loop hands back.
! compute_strength: fills ice_state's strength from the new iceTmask and halo-updates it (icepack_ice_strength,
! :596-608, 727-728).  Conditions: calc_strair = .true., ocean forcing on the T grid (grid_ocn_dynu = grid_ocn_dynv =
! 'T').  The host's own code after the loop that reads products of the preparation (dyn_finish at E / N points: aiX,
! fmX, uocnX, vocnX, cdn_ocnX and dyn_prep2's index lists) gets them through dyn_evp_hip_cgrid_fetch_forcing and from
! the masks.
  subroutine dyn_evp_hip_cgrid_evp_body(dt, compute_strength, ratiodxN, ratiodxNr, ratiodyE, ratiodyEr, &
                                        zetax2T, etax2T, etax2U, shearU, deltaU)

    use ice_blocks, only: nx_block, ny_block, block, get_block
    use ice_domain_size, only: max_blocks
    use ice_domain, only: nblocks, blocks_ice
    use ice_grid, only: tmask, umaskCD, emask, nmask, grid_ocn_dynu, grid_ocn_dynv
    use ice_state, only: aice, vice, vsno, aice_init, strength, aicen, vicen, uvel, vvel, uvelE, vvelE, uvelN, vvelN, &
                         divu, shear, vort
    use ice_arrays_column, only: Cdn_ocn
    use ice_flux, only: uocn, vocn, ss_tltx, ss_tlty, strairxT, strairyT, stresspT, stressmT, stress12T, stress12U, &
                        strintxE, strintyN, taubxE, taubyN, TbU, TbE, TbN, hwater, rdg_conv, rdg_shear
    use ice_dyn_shared, only: ndte, fcor_blk, fcorE_blk, fcorN_blk, iceTmask, iceUmask, iceEmask, iceNmask, &
                              dyn_area_min, dyn_mass_min, ssh_stress, seabed_stress, seabed_stress_method, &
                              seabed_stress_factor_LKD, seabed_stress_factor_prob
    use ice_timers, only: ice_timer_start, ice_timer_stop, timer_evp1dcore
    use icepack_intfc, only: icepack_query_parameters

    real(kind=dbl_kind), intent(in) :: dt
    interface
       subroutine compute_strength()
       end subroutine compute_strength
    end interface
    real(kind=dbl_kind), dimension(:,:,:), intent(in), contiguous, target :: ratiodxN, ratiodxNr, ratiodyE, ratiodyEr
    real(kind=dbl_kind), dimension(:,:,:), intent(inout), contiguous, target :: zetax2T, etax2T, etax2U, shearU, deltaU

    type(cice_evp_hip_prep_params) :: pp
    type(c_ptr) :: tf(11), s12(12), fl(19)
    integer(c_int32_t), pointer :: m1(:), m2(:), m3(:), m4(:), mT(:), mU(:), mE(:), mN(:)
    integer(int_kind), allocatable :: ixT(:), jxT(:), ixE(:), jxE(:), ixN(:), jxN(:), ixU(:), jxU(:)
    integer :: nall, iblk, i, j, nT, nE, nN, nU
    type(block) :: tb
    logical :: calc_strair
    character(len=*), parameter :: subname = '(dyn_evp_hip_cgrid_evp_body)'

    if (.not. initialised) call abort_ice(subname//' ERROR: dyn_evp_hip_init not called', &
         file=__FILE__, line=__LINE__)
    if (empty_rank) then
       call compute_strength()      ! (whatever the host does there among the tasks happens on this one too)
       return
    endif
    call icepack_query_parameters(calc_strair_out=calc_strair)
    if (.not. calc_strair .or. trim(grid_ocn_dynu) /= 'T' .or. trim(grid_ocn_dynv) /= 'T') &
       call abort_ice(subname//' ERROR: needs calc_strair = .true. and ocean forcing on the T grid', &
            file=__FILE__, line=__LINE__)
    nall = nx_block*ny_block*max_blocks
    call cgrid_ensure_geometry(ratiodxN, ratiodxNr, ratiodyE, ratiodyEr)
    if (.not. cgrid_prep_geometry_set) then
       call c_f_pointer(cice_evp_hip_addr(tmask), m1, [nall])
       call c_f_pointer(cice_evp_hip_addr(umaskCD), m2, [nall])
       call c_f_pointer(cice_evp_hip_addr(emask), m3, [nall])
       call c_f_pointer(cice_evp_hip_addr(nmask), m4, [nall])
       call check(cice_evp_hip_cgrid_set_prep_geometry(m1, m2, m3, m4, fcor_blk, fcorE_blk, fcorN_blk), &
            subname, __FILE__, __LINE__)
       cgrid_prep_geometry_set = .true.
    endif
    ! ice_dyn_evp.F90:372-382
    rdg_conv(:,:,1:nblocks) = 0.0_dbl_kind; rdg_shear(:,:,1:nblocks) = 0.0_dbl_kind
    divu(:,:,1:nblocks) = 0.0_dbl_kind; shear(:,:,1:nblocks) = 0.0_dbl_kind; vort(:,:,1:nblocks) = 0.0_dbl_kind

    pp%dt = dt
    call icepack_query_parameters(rhoi_out=pp%rhoi, rhos_out=pp%rhos, gravit_out=pp%gravit)
    pp%dyn_area_min = dyn_area_min
    pp%dyn_mass_min = dyn_mass_min
    pp%ssh_stress_coupled = merge(1_c_int32_t, 0_c_int32_t, trim(ssh_stress) == 'coupled')
    tf(1) = cice_evp_hip_addr(aice);      tf(2) = cice_evp_hip_addr(vice);     tf(3) = cice_evp_hip_addr(vsno)
    tf(4) = cice_evp_hip_addr(aice_init); tf(5) = cice_evp_hip_addr(Cdn_ocn)
    tf(6) = cice_evp_hip_addr(uocn);      tf(7) = cice_evp_hip_addr(vocn)
    tf(8) = cice_evp_hip_addr(ss_tltx);   tf(9) = cice_evp_hip_addr(ss_tlty)
    tf(10) = cice_evp_hip_addr(strairxT); tf(11) = cice_evp_hip_addr(strairyT)
    ! the state travels in at every call: the host's arrays stay the master copy (a host that never touches them
    ! between two evp() calls can pass null pointers to the C ABI instead and save the copies)
    s12 = [cice_evp_hip_addr(uvelE), cice_evp_hip_addr(vvelE), cice_evp_hip_addr(uvelN), cice_evp_hip_addr(vvelN), &
           cice_evp_hip_addr(uvel), cice_evp_hip_addr(vvel), cice_evp_hip_addr(stresspT), cice_evp_hip_addr(stressmT), &
           cice_evp_hip_addr(stress12T), cice_evp_hip_addr(stress12U), cice_evp_hip_addr(strintxE), &
           cice_evp_hip_addr(strintyN)]
    call c_f_pointer(cice_evp_hip_addr(iceTmask), mT, [nall])
    call c_f_pointer(cice_evp_hip_addr(iceUmask), mU, [nall])
    call c_f_pointer(cice_evp_hip_addr(iceEmask), mE, [nall])
    call c_f_pointer(cice_evp_hip_addr(iceNmask), mN, [nall])
    call check(cice_evp_hip_cgrid_prep(pp, tf, s12, mT, mU, mE, mN), subname, __FILE__, __LINE__)

    call compute_strength()
    call cgrid_halo_mask()

    if (seabed_stress) then
       ! the reference's own routines on the host, from the masks the device preparation returned (index lists as
       ! dyn_prep2 builds them, ice_dyn_shared.F90:740-770; call sites ice_dyn_evp.F90:803-827)
       allocate(ixT(nx_block*ny_block), jxT(nx_block*ny_block), ixE(nx_block*ny_block), jxE(nx_block*ny_block), &
                ixN(nx_block*ny_block), jxN(nx_block*ny_block), ixU(nx_block*ny_block), jxU(nx_block*ny_block))
       do iblk = 1, nblocks
          tb = get_block(blocks_ice(iblk), iblk)
          TbE(:,:,iblk) = 0.0_dbl_kind; TbN(:,:,iblk) = 0.0_dbl_kind
          nT = 0; nE = 0; nN = 0; nU = 0
          do j = tb%jlo, tb%jhi+1
          do i = tb%ilo, tb%ihi+1
             if (iceTmask(i,j,iblk)) then
                nT = nT + 1; ixT(nT) = i; jxT(nT) = j
             endif
          enddo
          enddo
          do j = tb%jlo, tb%jhi
          do i = tb%ilo, tb%ihi
             if (iceEmask(i,j,iblk)) then
                nE = nE + 1; ixE(nE) = i; jxE(nE) = j
             endif
             if (iceNmask(i,j,iblk)) then
                nN = nN + 1; ixN(nN) = i; jxN(nN) = j
             endif
             if (iceUmask(i,j,iblk)) then
                nU = nU + 1; ixU(nU) = i; jxU(nU) = j
             endif
          enddo
          enddo
          if (trim(seabed_stress_method) == 'LKD') then
             call seabed_stress_factor_LKD(nx_block, ny_block, nE, ixE, jxE, vice(:,:,iblk), aice(:,:,iblk), &
                                           hwater(:,:,iblk), TbE(:,:,iblk), grid_location='E')
             call seabed_stress_factor_LKD(nx_block, ny_block, nN, ixN, jxN, vice(:,:,iblk), aice(:,:,iblk), &
                                           hwater(:,:,iblk), TbN(:,:,iblk), grid_location='N')
          elseif (trim(seabed_stress_method) == 'probabilistic') then
             call seabed_stress_factor_prob(nx_block, ny_block, nT, ixT, jxT, nU, ixU, jxU, &
                                            aicen(:,:,:,iblk), vicen(:,:,:,iblk), hwater(:,:,iblk), TbU(:,:,iblk), &
                                            TbE(:,:,iblk), TbN(:,:,iblk), nE, ixE, jxE, nN, ixN, jxN)
          endif
       enddo
       deallocate(ixT, jxT, ixE, jxE, ixN, jxN, ixU, jxU)
       call check(cice_evp_hip_cgrid_set_tb(TbE, TbN), subname, __FILE__, __LINE__)
    endif
    call check(cice_evp_hip_cgrid_prep_finish(strength, cgrid_visc_method()), subname, __FILE__, __LINE__)

    call ice_timer_start(timer_evp1dcore)
    call check(cice_evp_hip_cgrid_subcycle(int(ndte, c_int32_t)), subname, __FILE__, __LINE__)
    fl(1:12) = s12
    fl(13) = cice_evp_hip_addr(taubxE); fl(14) = cice_evp_hip_addr(taubyN)
    fl(15) = c_loc(zetax2T); fl(16) = c_loc(etax2T); fl(17) = c_loc(etax2U); fl(18) = c_loc(shearU); fl(19) = c_loc(deltaU)
    call check(cice_evp_hip_cgrid_download(fl), subname, __FILE__, __LINE__)
    call ice_timer_stop(timer_evp1dcore)

  end subroutine dyn_evp_hip_cgrid_evp_body

!-----------------------------------------------------------------------
! Products of the device preparation that the host's own post-loop code reads at E and N points (dyn_finish,
! ice_dyn_evp.F90:1404-1424): private arrays of ice_dyn_evp, so the patched evp() passes them.  fmE / fmN are public.
  subroutine dyn_evp_hip_cgrid_fetch_forcing(cdn_ocnE, aiE, uocnE, vocnE, cdn_ocnN, aiN, uocnN, vocnN)
    use ice_flux, only: fmE, fmN
    use ice_dyn_shared, only: uvelE_init, vvelN_init
    real(kind=dbl_kind), dimension(:,:,:), intent(inout), contiguous :: cdn_ocnE, aiE, uocnE, vocnE, cdn_ocnN, aiN, uocnN, vocnN
    character(len=*), parameter :: subname = '(dyn_evp_hip_cgrid_fetch_forcing)'
    if (empty_rank) return
    ! indices into the loop's input table (include/cice_evp_hip.h: inputs23, 0-based)
    call check(cice_evp_hip_cgrid_fetch(1, 1, cdn_ocnE), subname, __FILE__, __LINE__)
    call check(cice_evp_hip_cgrid_fetch(1, 2, aiE), subname, __FILE__, __LINE__)
    call check(cice_evp_hip_cgrid_fetch(1, 3, uocnE), subname, __FILE__, __LINE__)
    call check(cice_evp_hip_cgrid_fetch(1, 4, vocnE), subname, __FILE__, __LINE__)
    call check(cice_evp_hip_cgrid_fetch(1, 8, fmE), subname, __FILE__, __LINE__)
    call check(cice_evp_hip_cgrid_fetch(1, 9, uvelE_init), subname, __FILE__, __LINE__)
    call check(cice_evp_hip_cgrid_fetch(1, 12, cdn_ocnN), subname, __FILE__, __LINE__)
    call check(cice_evp_hip_cgrid_fetch(1, 13, aiN), subname, __FILE__, __LINE__)
    call check(cice_evp_hip_cgrid_fetch(1, 14, uocnN), subname, __FILE__, __LINE__)
    call check(cice_evp_hip_cgrid_fetch(1, 15, vocnN), subname, __FILE__, __LINE__)
    call check(cice_evp_hip_cgrid_fetch(1, 19, fmN), subname, __FILE__, __LINE__)
    call check(cice_evp_hip_cgrid_fetch(1, 20, vvelN_init), subname, __FILE__, __LINE__)
  end subroutine dyn_evp_hip_cgrid_fetch_forcing

!-----------------------------------------------------------------------
! C grid: replaces the call of deformationsC_T that follows the loop in evp() (ice_dyn_evp.F90:1106-1119): divu, shear,
! vort, rdg_conv, rdg_shear from the final state dyn_evp_hip_cgrid_run left on the device.
  subroutine dyn_evp_hip_cgrid_deformations
    use ice_grid, only: tarear
    use ice_state, only: divu, shear, vort
    use ice_flux, only: rdg_conv, rdg_shear
    character(len=*), parameter :: subname = '(dyn_evp_hip_cgrid_deformations)'
    if (.not. initialised) call abort_ice(subname//' ERROR: dyn_evp_hip_init not called', file=__FILE__, line=__LINE__)
    if (empty_rank) return
    call check(cice_evp_hip_cgrid_deformations(tarear, divu, shear, vort, rdg_conv, rdg_shear), subname, __FILE__, __LINE__)
  end subroutine dyn_evp_hip_cgrid_deformations

  ! dyn_finish at N and E points (ice_dyn_evp.F90:1408-1436) on the device, from the state the C-grid loop left there:
  ! ice_flux's strocnxN / strocnyN / strocnxE / strocnyE are written on the cells of the N / E lists.
  subroutine dyn_evp_hip_cgrid_dyn_finish
    use ice_flux, only: strocnxN, strocnyN, strocnxE, strocnyE
    character(len=*), parameter :: subname = '(dyn_evp_hip_cgrid_dyn_finish)'
    if (.not. initialised) call abort_ice(subname//' ERROR: dyn_evp_hip_init not called', file=__FILE__, line=__LINE__)
    if (empty_rank) return
    call check(cice_evp_hip_cgrid_dyn_finish(strocnxN, strocnyN, strocnxE, strocnyE), subname, __FILE__, __LINE__)
  end subroutine dyn_evp_hip_cgrid_dyn_finish

!-----------------------------------------------------------------------
! The one place that decides whether the stresses stay on the device (dyn_evp_hip_init and
! dyn_evp_hip_keep_stresses_resident both come here).  Resident stresses on a tripole / tripoleT grid need the device to
! do what evp()'s twelve ice_HaloUpdate_stress calls do on the host arrays (ice_dyn_evp.F90:1321-1389), and every rank
! must reach the same verdict: tripole -- always possible; tripoleT -- only where cice_evp_hip_stress_halo_available()
! says so on EVERY rank (MIN over the ranks: one rank with its top row split falls back, so all do, and
! dyn_evp_hip_fetch_stresses / restart behaviour never depends on the rank).  explicit = the host asked for it after
! initialisation: a request that cannot be honoured aborts instead of silently leaving the device copy unsymmetrised.
  subroutine settle_stress_residency(want, explicit)
    use ice_domain, only: ns_boundary_type, distrb_info
    use ice_global_reductions, only: global_minval
    logical, intent(in) :: want, explicit
    integer (kind=int_kind) :: avail
    character(len=*), parameter :: subname = '(settle_stress_residency)'
    on_tripole = trim(ns_boundary_type) == 'tripole'
    stress_resident = want
    if (trim(ns_boundary_type) == 'tripoleT') then
       ! global_minval is an MPI_ALLREDUCE over every task of the distribution (ice_global_reductions enters it for all
       ! my_task < numProcs): a task without blocks takes part with the neutral value, or the others would wait for it for ever
       avail = 1
       if (.not. empty_rank) avail = cice_evp_hip_stress_halo_available()
       avail = global_minval(avail, distrb_info)
       if (avail == 1) then
          on_tripole = .true.
       else if (want .and. explicit) then
          call abort_ice(subname//' ERROR: device-resident stresses need the tripoleT fold row on one rank'// &
               ' (cice_evp_hip_stress_halo_available() == 0 on some rank)', file=__FILE__, line=__LINE__)
       else
          stress_resident = .false.
       endif
    endif
    if (.not. empty_rank) &
    call check(cice_evp_hip_set_option(1_c_int32_t, merge(1_c_int32_t, 0_c_int32_t, stress_resident)), &
         subname, __FILE__, __LINE__)
  end subroutine settle_stress_residency

! ice_flux's stress arrays <- the device copy.  Call before anything but evp() reads them (restart write,
! history).  No-op unless the stresses are resident.
  subroutine dyn_evp_hip_fetch_stresses
    use ice_flux, only: stressp_1, stressp_2, stressp_3, stressp_4, stressm_1, stressm_2, stressm_3, stressm_4, &
         stress12_1, stress12_2, stress12_3, stress12_4
    type(c_ptr) :: s12(12)
    character(len=*), parameter :: subname = '(dyn_evp_hip_fetch_stresses)'
    if (.not. (initialised .and. stress_resident) .or. empty_rank) return
    s12(1) = cice_evp_hip_addr(stressp_1);  s12(2) = cice_evp_hip_addr(stressp_2)
    s12(3) = cice_evp_hip_addr(stressp_3);  s12(4) = cice_evp_hip_addr(stressp_4)
    s12(5) = cice_evp_hip_addr(stressm_1);  s12(6) = cice_evp_hip_addr(stressm_2)
    s12(7) = cice_evp_hip_addr(stressm_3);  s12(8) = cice_evp_hip_addr(stressm_4)
    s12(9) = cice_evp_hip_addr(stress12_1); s12(10) = cice_evp_hip_addr(stress12_2)
    s12(11) = cice_evp_hip_addr(stress12_3); s12(12) = cice_evp_hip_addr(stress12_4)
    call check(cice_evp_hip_fetch_stresses(s12), subname, __FILE__, __LINE__)
  end subroutine dyn_evp_hip_fetch_stresses

! Opt in to (or out of) device-resident stresses.  ONLY for hosts that call dyn_evp_hip_fetch_stresses before
! every reader of ice_flux's stress arrays and dyn_evp_hip_invalidate_stresses after every writer; may be
! called before or after dyn_evp_hip_init.  Switching off brings the host arrays up to date first.
  subroutine dyn_evp_hip_keep_stresses_resident(flag)
    logical, intent(in) :: flag
    character(len=*), parameter :: subname = '(dyn_evp_hip_keep_stresses_resident)'
    stress_resident_requested = flag
    if (.not. initialised) return
    if (.not. flag .and. stress_resident) call dyn_evp_hip_fetch_stresses      ! (returns at once on a task without blocks)
    call settle_stress_residency(flag, .true.)                                  ! collective on tripoleT grids: every task
  end subroutine dyn_evp_hip_keep_stresses_resident

! The host changed ice_flux's stress arrays itself (restart read): the next evp() uploads them again.
  subroutine dyn_evp_hip_invalidate_stresses
    character(len=*), parameter :: subname = '(dyn_evp_hip_invalidate_stresses)'
    body_sig_on_device = .false.
    if (initialised .and. .not. empty_rank) call check(cice_evp_hip_invalidate_stresses(), subname, __FILE__, __LINE__)
  end subroutine dyn_evp_hip_invalidate_stresses

!-----------------------------------------------------------------------
  subroutine dyn_evp_hip_finalize
    character(len=*), parameter :: subname = '(dyn_evp_hip_finalize)'
    if (initialised) call check(cice_evp_hip_finalize(), subname, __FILE__, __LINE__)
    initialised = .false.
    empty_rank = .false.
    pinned = .false.
  end subroutine dyn_evp_hip_finalize

end module ice_dyn_evp_hip
