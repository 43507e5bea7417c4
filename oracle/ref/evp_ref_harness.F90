!=======================================================================
! TEST INFRASTRUCTURE (oracle/ref) -- drives the reference's own, unmodified
! evp() (/root/reference/cicecore/cicedyn/dynamics/ice_dyn_evp.F90:259) on a
! self-contained grid through the reference's serial comm layer, and dumps
!   * the static grid / metric arrays and EVP scalars,
!   * every input of the EVP subcycle, captured at the drop-in boundary by
!     the build-owned `ice_dyn_evp1d` capture module (evp_algorithm toggled to
!     'shared_mem_1d' for one call that computes nothing),
!   * the reference's outputs for exactly those inputs after nsub subcycles
!     (evp_algorithm='standard_2d', private 2-D `stress` + public `stepu` +
!     serial ice_HaloUpdate incl. tripole), for each nsub in nsub_list,
!   * its own timer for the subcycle loop (timer_evp), for the CPU baseline.
! Call order mirrors cice_init
! (/root/reference/cicecore/drivers/standalone/cice/CICE_InitMod.F90:97-141)
! minus everything Icepack/IO; see SURVEY.md Appendix A.3.
!
! Inputs: ./ice_in (only &domain_nml) and ./harness_in (&harness_nml).
! This program is written for this repo; it contains no reference code.
!=======================================================================
! C-grid capture (a module of its own: the program's internal procedures must not host-associate anything,
! or flang builds stack trampolines for the ice-strength callback of the drop-in variant)
module evp_cgrid_capture
  use ice_kinds_mod
  use ice_domain, only: nblocks
  use ice_domain_size, only: max_blocks
  use ice_blocks, only: nx_block, ny_block
  use ice_state, only: uvel, vvel, uvelE, vvelE, uvelN, vvelN, strength, divu, shear, vort, aice, vice, vsno, aice_init, &
                       aice0, aicen, vicen
  use ice_blocks, only: block, get_block
  use ice_domain, only: blocks_ice, halo_info
  use ice_boundary, only: ice_HaloUpdate
  use ice_constants, only: field_loc_center, field_type_scalar, c0, c1
  use ice_grid, only: tmask
  use ice_arrays_column, only: Cdn_ocn
  use ice_flux
  use ice_calendar, only: dt_dyn
  use ice_dyn_shared
  use ice_dyn_evp, only: evp, ratiodxN, ratiodxNr, ratiodyE, ratiodyEr
#ifdef HARNESS_HIP_BODY
  use ice_dyn_evp_hip, only: dyn_evp_hip_cgrid_run, dyn_evp_hip_cgrid_deformations, dyn_evp_hip_cgrid_evp_body, &
       dyn_evp_hip_cgrid_dyn_finish
#endif
  use evp_dumpio
  implicit none
  real(dbl_kind), allocatable, dimension(:,:,:), private :: c_uE, c_vN, c_uN, c_vE, c_spT, c_smT, c_s12T, c_s12U, c_u, c_v
  ! the state evp() is entered with (before its preparation), for the device-preparation variant
  real(dbl_kind), allocatable, dimension(:,:,:), private :: q_uE, q_vN, q_uN, q_vE, q_spT, q_smT, q_s12T, q_s12U, q_u, q_v, &
                                                            q_sxE, q_syN
  logical(log_kind), allocatable, dimension(:,:,:), private :: q_mU, q_mE, q_mN
contains

  ! the ice cover between two calls: some cells lose their ice, some open-water cells gain some (dyn_prep2's
  ! "new ice starts at the ocean velocity" / "no ice: zero" branches, ice_dyn_shared.F90:747-764)
  subroutine evolve_ice()
    integer(int_kind) :: ib, i, j, ig, jg
    type(block) :: tb
    do ib = 1, nblocks
       tb = get_block(blocks_ice(ib), ib)
       do j = 1, ny_block
       do i = 1, nx_block
          ig = tb%i_glob(i); jg = tb%j_glob(j)
          if (ig < 1 .or. jg < 1 .or. .not. tmask(i,j,ib)) cycle
          if (aice(i,j,ib) > c0 .and. (mod(ig + 3*jg, 7) == 0 .or. (ig >= 4 .and. ig <= 8 .and. jg >= 5 .and. jg <= 9))) then
             aice(i,j,ib) = c0; vice(i,j,ib) = c0; vsno(i,j,ib) = c0
          elseif (aice(i,j,ib) == c0 .and. mod(2*ig + jg, 3) /= 0) then
             aice(i,j,ib) = 0.6_dbl_kind; vice(i,j,ib) = 0.9_dbl_kind; vsno(i,j,ib) = 0.05_dbl_kind
          endif
          aice_init(i,j,ib) = aice(i,j,ib)
       enddo
       enddo
    enddo
    call ice_HaloUpdate(aice,      halo_info, field_loc_center, field_type_scalar)
    call ice_HaloUpdate(vice,      halo_info, field_loc_center, field_type_scalar)
    call ice_HaloUpdate(vsno,      halo_info, field_loc_center, field_type_scalar)
    call ice_HaloUpdate(aice_init, halo_info, field_loc_center, field_type_scalar)
    aice0(:,:,1:nblocks) = c1 - aice(:,:,1:nblocks)
    aicen(:,:,1,1:nblocks) = aice(:,:,1:nblocks)
    vicen(:,:,1,1:nblocks) = vice(:,:,1:nblocks)
  end subroutine evolve_ice

  ! icepack_ice_strength on the T-cells of the new iceTmask + its halo update (ice_dyn_evp.F90:596-608, 727-728):
  ! the callback of dyn_evp_hip_cgrid_evp_body
  subroutine cgrid_strength()
    use icepack_intfc, only: icepack_ice_strength
    integer :: ib, ii, jj
    type(block) :: bb
    do ib = 1, nblocks
       bb = get_block(blocks_ice(ib), ib)
       strength(:,:,ib) = c0
       do jj = bb%jlo, bb%jhi+1
       do ii = bb%ilo, bb%ihi+1
          if (iceTmask(ii,jj,ib)) &
             call icepack_ice_strength(aice=aice(ii,jj,ib), vice=vice(ii,jj,ib), aice0=aice0(ii,jj,ib), &
                                       aicen=aicen(ii,jj,:,ib), vicen=vicen(ii,jj,:,ib), strength=strength(ii,jj,ib))
       enddo
       enddo
    enddo
    call ice_HaloUpdate(strength, halo_info, field_loc_center, field_type_scalar)
  end subroutine cgrid_strength

  ! ---- C grid: the subcycle inputs are module-private; a preparation-only evp() (ndte = 0) leaves them in place,
  !      evp_peek.c reads them; the very next evp() calls from the same state give the reference's outputs ----

  subroutine cgrid_call(ic, nsub_list, nl, h_ndte, hipmode, evolve, hipprep)
    integer(int_kind), intent(in) :: ic, nsub_list(:), nl, h_ndte
    logical, intent(in) :: hipmode, evolve, hipprep
    integer(int_kind) :: kk, ns, ib, i, j, ig, jg
    type(block) :: tb
    character(len=16) :: tg
    if (.not. allocated(c_uE)) then
       allocate(c_uE(nx_block,ny_block,max_blocks), c_vN(nx_block,ny_block,max_blocks), &
                c_uN(nx_block,ny_block,max_blocks), c_vE(nx_block,ny_block,max_blocks), &
                c_spT(nx_block,ny_block,max_blocks), c_smT(nx_block,ny_block,max_blocks), &
                c_s12T(nx_block,ny_block,max_blocks), c_s12U(nx_block,ny_block,max_blocks), &
                c_u(nx_block,ny_block,max_blocks), c_v(nx_block,ny_block,max_blocks))
    endif
    if (evolve .and. ic == 2) call evolve_ice()
    ! what evp()'s preparation phase reads on the C grid (ice_dyn_evp.F90:383-735): the T-grid state and forcing, the
    ! velocities / stresses / ice masks the previous call left (SURVEY 8 f-2 for grid_ice = 'C')
    write(tg,'(a,i2.2)') 'cp', ic
    call dump_r8_3d(trim(tg)//'_aice', aice, nblocks);       call dump_r8_3d(trim(tg)//'_vice', vice, nblocks)
    call dump_r8_3d(trim(tg)//'_vsno', vsno, nblocks);       call dump_r8_3d(trim(tg)//'_aice_init', aice_init, nblocks)
    call dump_r8_3d(trim(tg)//'_cdn_ocn', Cdn_ocn, nblocks)
    call dump_r8_3d(trim(tg)//'_uocn', uocn, nblocks);       call dump_r8_3d(trim(tg)//'_vocn', vocn, nblocks)
    call dump_r8_3d(trim(tg)//'_ss_tltx', ss_tltx, nblocks); call dump_r8_3d(trim(tg)//'_ss_tlty', ss_tlty, nblocks)
    call dump_r8_3d(trim(tg)//'_strairxT', strairxT, nblocks); call dump_r8_3d(trim(tg)//'_strairyT', strairyT, nblocks)
    call dump_r8_3d(trim(tg)//'_uvelE', uvelE, nblocks);   call dump_r8_3d(trim(tg)//'_vvelE', vvelE, nblocks)
    call dump_r8_3d(trim(tg)//'_uvelN', uvelN, nblocks);   call dump_r8_3d(trim(tg)//'_vvelN', vvelN, nblocks)
    call dump_r8_3d(trim(tg)//'_uvel', uvel, nblocks);     call dump_r8_3d(trim(tg)//'_vvel', vvel, nblocks)
    call dump_r8_3d(trim(tg)//'_stresspT', stresspT, nblocks);   call dump_r8_3d(trim(tg)//'_stressmT', stressmT, nblocks)
    call dump_r8_3d(trim(tg)//'_stress12T', stress12T, nblocks); call dump_r8_3d(trim(tg)//'_stress12U', stress12U, nblocks)
    call dump_r8_3d(trim(tg)//'_strintxE', strintxE, nblocks);   call dump_r8_3d(trim(tg)//'_strintyN', strintyN, nblocks)
    call dump_l_3d (trim(tg)//'_iceUmask', iceUmask, nblocks)
    call dump_l_3d (trim(tg)//'_iceEmask', iceEmask, nblocks);   call dump_l_3d (trim(tg)//'_iceNmask', iceNmask, nblocks)
    if (.not. allocated(q_uE)) then
       allocate(q_uE(nx_block,ny_block,max_blocks), q_vN(nx_block,ny_block,max_blocks), q_uN(nx_block,ny_block,max_blocks), &
                q_vE(nx_block,ny_block,max_blocks), q_spT(nx_block,ny_block,max_blocks), q_smT(nx_block,ny_block,max_blocks), &
                q_s12T(nx_block,ny_block,max_blocks), q_s12U(nx_block,ny_block,max_blocks), q_u(nx_block,ny_block,max_blocks), &
                q_v(nx_block,ny_block,max_blocks), q_sxE(nx_block,ny_block,max_blocks), q_syN(nx_block,ny_block,max_blocks), &
                q_mU(nx_block,ny_block,max_blocks), q_mE(nx_block,ny_block,max_blocks), q_mN(nx_block,ny_block,max_blocks))
    endif
    q_uE = uvelE; q_vN = vvelN; q_uN = uvelN; q_vE = vvelE; q_u = uvel; q_v = vvel
    q_spT = stresspT; q_smT = stressmT; q_s12T = stress12T; q_s12U = stress12U; q_sxE = strintxE; q_syN = strintyN
    q_mU = iceUmask; q_mE = iceEmask; q_mN = iceNmask
    ndte = 0
    call evp(dt_dyn)                 ! preparation only
    ndte = h_ndte
    write(tg,'(a,i2.2)') 'in', ic
    ! state the loop starts from
    call dump_r8_3d(trim(tg)//'_uvelE', uvelE, nblocks);   call dump_r8_3d(trim(tg)//'_vvelE', vvelE, nblocks)
    call dump_r8_3d(trim(tg)//'_uvelN', uvelN, nblocks);   call dump_r8_3d(trim(tg)//'_vvelN', vvelN, nblocks)
    call dump_r8_3d(trim(tg)//'_uvel', uvel, nblocks);     call dump_r8_3d(trim(tg)//'_vvel', vvel, nblocks)
    call dump_r8_3d(trim(tg)//'_stresspT', stresspT, nblocks);   call dump_r8_3d(trim(tg)//'_stressmT', stressmT, nblocks)
    call dump_r8_3d(trim(tg)//'_stress12T', stress12T, nblocks); call dump_r8_3d(trim(tg)//'_stress12U', stress12U, nblocks)
    ! per-call inputs: public ...
    call dump_r8_3d(trim(tg)//'_strength', strength, nblocks)
    call dump_r8_3d(trim(tg)//'_fmE', fmE, nblocks);       call dump_r8_3d(trim(tg)//'_fmN', fmN, nblocks)
    call dump_r8_3d(trim(tg)//'_TbE', TbE, nblocks);       call dump_r8_3d(trim(tg)//'_TbN', TbN, nblocks)
    call dump_r8_3d(trim(tg)//'_uvelE_init', uvelE_init, nblocks); call dump_r8_3d(trim(tg)//'_vvelN_init', vvelN_init, nblocks)
    call dump_r8_3d(trim(tg)//'_strintxE', strintxE, nblocks);     call dump_r8_3d(trim(tg)//'_strintyN', strintyN, nblocks)
    call dump_r8_3d(trim(tg)//'_taubxE', taubxE, nblocks);         call dump_r8_3d(trim(tg)//'_taubyN', taubyN, nblocks)
    call dump_l_3d (trim(tg)//'_iceTmask', iceTmask, nblocks);     call dump_l_3d (trim(tg)//'_iceUmask', iceUmask, nblocks)
    call dump_l_3d (trim(tg)//'_iceEmask', iceEmask, nblocks);     call dump_l_3d (trim(tg)//'_iceNmask', iceNmask, nblocks)
    ! ... and module-private (evp_peek.c)
    call dump_peek(trim(tg)//'_uocnE', 1, nx_block, ny_block, max_blocks, nblocks);      call dump_peek(trim(tg)//'_vocnE', 2, nx_block, ny_block, max_blocks, nblocks);     call dump_peek(trim(tg)//'_cdn_ocnE', 3, nx_block, ny_block, max_blocks, nblocks)
    call dump_peek(trim(tg)//'_waterxE', 4, nx_block, ny_block, max_blocks, nblocks);    call dump_peek(trim(tg)//'_forcexE', 5, nx_block, ny_block, max_blocks, nblocks);   call dump_peek(trim(tg)//'_aiE', 6, nx_block, ny_block, max_blocks, nblocks)
    call dump_peek(trim(tg)//'_rheofactE', 7, nx_block, ny_block, max_blocks, nblocks);  call dump_peek(trim(tg)//'_emassdti', 8, nx_block, ny_block, max_blocks, nblocks)
    call dump_peek(trim(tg)//'_uocnN', 9, nx_block, ny_block, max_blocks, nblocks);      call dump_peek(trim(tg)//'_vocnN', 10, nx_block, ny_block, max_blocks, nblocks);    call dump_peek(trim(tg)//'_cdn_ocnN', 11, nx_block, ny_block, max_blocks, nblocks)
    call dump_peek(trim(tg)//'_wateryN', 12, nx_block, ny_block, max_blocks, nblocks);   call dump_peek(trim(tg)//'_forceyN', 13, nx_block, ny_block, max_blocks, nblocks);  call dump_peek(trim(tg)//'_aiN', 14, nx_block, ny_block, max_blocks, nblocks)
    call dump_peek(trim(tg)//'_rheofactN', 15, nx_block, ny_block, max_blocks, nblocks); call dump_peek(trim(tg)//'_nmassdti', 16, nx_block, ny_block, max_blocks, nblocks)
    c_uE = uvelE; c_vN = vvelN; c_uN = uvelN; c_vE = vvelE; c_u = uvel; c_v = vvel
    c_spT = stresspT; c_smT = stressmT; c_s12T = stress12T; c_s12U = stress12U
    ! (the device runs come first: the reference's last run below leaves the state the next call continues from)
#ifdef HARNESS_HIP_BODY
    if (hipmode) then
       ! the HIP loop through the Fortran entry a patched evp() would call (INTEGRATION.md, C grid): same start
       ! state, preparation by the reference (ndte = 0), then the loop on the device
       do kk = 1, nl
          ns = nsub_list(kk)
          uvelE = c_uE; vvelN = c_vN; uvelN = c_uN; vvelE = c_vE; uvel = c_u; vvel = c_v
          stresspT = c_spT; stressmT = c_smT; stress12T = c_s12T; stress12U = c_s12U
          if (hipprep) then
             ! the preparation on the device as well: from the state evp() was entered with (the C-grid counterpart of
             ! dyn_evp_hip_evp_body; the reference's evp() does not run at all for this output)
             uvelE = q_uE; vvelN = q_vN; uvelN = q_uN; vvelE = q_vE; uvel = q_u; vvel = q_v
             stresspT = q_spT; stressmT = q_smT; stress12T = q_s12T; stress12U = q_s12U; strintxE = q_sxE; strintyN = q_syN
             iceUmask = q_mU; iceEmask = q_mE; iceNmask = q_mN
             ndte = ns
             call dyn_evp_hip_cgrid_evp_body(dt_dyn, cgrid_strength, ratiodxN, ratiodxNr, ratiodyE, ratiodyEr, &
                                             pk3(17), pk3(18), pk3(19), pk3(20), pk3(21))
          else
          ndte = 0
          call evp(dt_dyn)
          ndte = ns
          call dyn_evp_hip_cgrid_run( &
               pk3(1), pk3(2), pk3(3), pk3(4), pk3(5), pk3(6), pk3(7), pk3(8), &
               pk3(9), pk3(10), pk3(11), pk3(12), pk3(13), pk3(14), pk3(15), pk3(16), &
               ratiodxN, ratiodxNr, ratiodyE, ratiodyEr, pk3(17), pk3(18), pk3(19), pk3(20), pk3(21))
          endif
          ndte = h_ndte
          write(tg,'(a,i2.2,a,i4.4)') 'h', ic, 'n', ns
          call dump_r8_3d(trim(tg)//'_uvelE', uvelE, nblocks);   call dump_r8_3d(trim(tg)//'_vvelE', vvelE, nblocks)
          call dump_r8_3d(trim(tg)//'_uvelN', uvelN, nblocks);   call dump_r8_3d(trim(tg)//'_vvelN', vvelN, nblocks)
          call dump_r8_3d(trim(tg)//'_uvel', uvel, nblocks);     call dump_r8_3d(trim(tg)//'_vvel', vvel, nblocks)
          call dump_r8_3d(trim(tg)//'_stresspT', stresspT, nblocks);   call dump_r8_3d(trim(tg)//'_stressmT', stressmT, nblocks)
          call dump_r8_3d(trim(tg)//'_stress12T', stress12T, nblocks); call dump_r8_3d(trim(tg)//'_stress12U', stress12U, nblocks)
          call dump_r8_3d(trim(tg)//'_strintxE', strintxE, nblocks);   call dump_r8_3d(trim(tg)//'_strintyN', strintyN, nblocks)
          call dump_r8_3d(trim(tg)//'_taubxE', taubxE, nblocks);       call dump_r8_3d(trim(tg)//'_taubyN', taubyN, nblocks)
          call dump_peek(trim(tg)//'_zetax2T', 17, nx_block, ny_block, max_blocks, nblocks)
          call dump_peek(trim(tg)//'_etax2T', 18, nx_block, ny_block, max_blocks, nblocks)
          call dump_peek(trim(tg)//'_etax2U', 19, nx_block, ny_block, max_blocks, nblocks)
          call dump_peek(trim(tg)//'_shearU', 20, nx_block, ny_block, max_blocks, nblocks)
          call dump_peek(trim(tg)//'_deltaU', 21, nx_block, ny_block, max_blocks, nblocks)
          ! deformationsC_T on the device, from the state the loop left there (evp(ndte = 0) above has already filled the
          ! five arrays from the INITIAL velocities: every list cell must be overwritten)
          call dyn_evp_hip_cgrid_deformations
          call dump_r8_3d(trim(tg)//'_divu', divu, nblocks);         call dump_r8_3d(trim(tg)//'_shear', shear, nblocks)
          call dump_r8_3d(trim(tg)//'_vort', vort, nblocks)
          call dump_r8_3d(trim(tg)//'_rdg_conv', rdg_conv, nblocks); call dump_r8_3d(trim(tg)//'_rdg_shear', rdg_shear, nblocks)
          ! dyn_finish at N and E points on the device (the arrays hold what evp(ndte = 0) computed from the INITIAL velocities:
          ! every list cell must be overwritten)
          call dyn_evp_hip_cgrid_dyn_finish
          call dump_r8_3d(trim(tg)//'_strocnxN', strocnxN, nblocks); call dump_r8_3d(trim(tg)//'_strocnyN', strocnyN, nblocks)
          call dump_r8_3d(trim(tg)//'_strocnxE', strocnxE, nblocks); call dump_r8_3d(trim(tg)//'_strocnyE', strocnyE, nblocks)
          write(*,'(a,i3,a,i5,3es24.16)') 'Hcall', ic, ' nsub', ns, &
               maxval(abs(uvelE(:,:,1:nblocks))), maxval(abs(vvelN(:,:,1:nblocks))), maxval(abs(stresspT(:,:,1:nblocks)))
       enddo
    endif
#endif
    do kk = 1, nl
       ns = nsub_list(kk)
       uvelE = c_uE; vvelN = c_vN; uvelN = c_uN; vvelE = c_vE; uvel = c_u; vvel = c_v
       stresspT = c_spT; stressmT = c_smT; stress12T = c_s12T; stress12U = c_s12U
       ndte = ns
       call evp(dt_dyn)
       ndte = h_ndte
       write(tg,'(a,i2.2,a,i4.4)') 'o', ic, 'n', ns
       call dump_r8_3d(trim(tg)//'_uvelE', uvelE, nblocks);   call dump_r8_3d(trim(tg)//'_vvelE', vvelE, nblocks)
       call dump_r8_3d(trim(tg)//'_uvelN', uvelN, nblocks);   call dump_r8_3d(trim(tg)//'_vvelN', vvelN, nblocks)
       call dump_r8_3d(trim(tg)//'_uvel', uvel, nblocks);     call dump_r8_3d(trim(tg)//'_vvel', vvel, nblocks)
       call dump_r8_3d(trim(tg)//'_stresspT', stresspT, nblocks);   call dump_r8_3d(trim(tg)//'_stressmT', stressmT, nblocks)
       call dump_r8_3d(trim(tg)//'_stress12T', stress12T, nblocks); call dump_r8_3d(trim(tg)//'_stress12U', stress12U, nblocks)
       call dump_r8_3d(trim(tg)//'_strintxE', strintxE, nblocks);   call dump_r8_3d(trim(tg)//'_strintyN', strintyN, nblocks)
       call dump_r8_3d(trim(tg)//'_taubxE', taubxE, nblocks);       call dump_r8_3d(trim(tg)//'_taubyN', taubyN, nblocks)
       call dump_peek(trim(tg)//'_zetax2T', 17, nx_block, ny_block, max_blocks, nblocks);  call dump_peek(trim(tg)//'_etax2T', 18, nx_block, ny_block, max_blocks, nblocks);  call dump_peek(trim(tg)//'_etax2U', 19, nx_block, ny_block, max_blocks, nblocks)
       call dump_peek(trim(tg)//'_shearU', 20, nx_block, ny_block, max_blocks, nblocks);   call dump_peek(trim(tg)//'_deltaU', 21, nx_block, ny_block, max_blocks, nblocks)
       ! deformationsC_T (ice_dyn_shared.F90:1968-2074), called by evp() right after the loop (:1106-1119)
       call dump_r8_3d(trim(tg)//'_divu', divu, nblocks);         call dump_r8_3d(trim(tg)//'_shear', shear, nblocks)
       call dump_r8_3d(trim(tg)//'_vort', vort, nblocks)
       call dump_r8_3d(trim(tg)//'_rdg_conv', rdg_conv, nblocks); call dump_r8_3d(trim(tg)//'_rdg_shear', rdg_shear, nblocks)
       ! dyn_finish at N and E points (ice_dyn_evp.F90:1408-1436)
       call dump_r8_3d(trim(tg)//'_strocnxN', strocnxN, nblocks); call dump_r8_3d(trim(tg)//'_strocnyN', strocnyN, nblocks)
       call dump_r8_3d(trim(tg)//'_strocnxE', strocnxE, nblocks); call dump_r8_3d(trim(tg)//'_strocnyE', strocnyE, nblocks)
       write(*,'(a,i3,a,i5,3es24.16)') 'Ccall', ic, ' nsub', ns, &
            maxval(abs(uvelE(:,:,1:nblocks))), maxval(abs(vvelN(:,:,1:nblocks))), maxval(abs(stresspT(:,:,1:nblocks)))
    enddo
  end subroutine cgrid_call

  function pk3(which) result(p)
    integer(int_kind), intent(in) :: which
    real(dbl_kind), pointer, contiguous :: p(:,:,:)
    p => peek_array(which, nx_block, ny_block, max_blocks)
  end function pk3
end module evp_cgrid_capture

program evp_ref_harness

  use ice_kinds_mod
  use ice_constants
  use ice_communicate, only: init_communicate, my_task, get_num_procs
  use ice_exit, only: end_run
  use ice_fileunits, only: init_fileunits, nu_diag, nml_filename
  use ice_domain, only: init_domain_blocks, nblocks, blocks_ice, halo_info, &
      ew_boundary_type, ns_boundary_type, maskhalo_dyn
  use ice_domain_size
  use ice_blocks, only: nx_block, ny_block, block, get_block, nghost
  use ice_boundary, only: ice_HaloUpdate
  use ice_grid
  use ice_state
  use ice_flux
  use ice_flux_bgc, only: alloc_flux_bgc
  use ice_arrays_column, only: alloc_arrays_column, Cdn_ocn
  use ice_timers, only: init_ice_timers, ice_timer_print_all, ice_timer_clear, &
      timer_evp
  use ice_calendar, only: dt, dt_dyn, ndtd
  use ice_dyn_shared
  use ice_dyn_evp, only: init_evp, evp, ratiodxN, ratiodxNr, ratiodyE, ratiodyEr
#ifdef HARNESS_REF1D
  ! timing-only build: the reference's OWN ice_dyn_evp1d (its 1-d "shared_mem_1d" EVP core,
  ! ice_dyn_evp1d.F90 + ice_dyn_core1d.F90) instead of the capture module
  use ice_dyn_evp1d, only: dyn_evp1d_init, dyn_evp1d_finalize
#else
  use ice_dyn_evp1d, only: capture_tag, dyn_evp1d_init, dyn_evp1d_finalize
#endif
#ifdef HARNESS_HIP_BODY
  use ice_dyn_evp_hip, only: dyn_evp_hip_evp_body, dyn_evp_hip_fetch_stresses, dyn_evp_hip_invalidate_stresses, &
      dyn_evp_hip_keep_stresses_resident
#endif
  use evp_dumpio
  use evp_cgrid_capture, only: cgrid_call, evolve_ice
  use icepack_intfc, only: icepack_query_parameters
#if defined (_OPENMP)
  use OMP_LIB
#endif

  implicit none

  ! ---- harness namelist ------------------------------------------------
  character(len=32)  :: grid_kind   = 'rect'      ! 'rect' | 'popfile' | 'tripolefile'
  character(len=32)  :: kmt_kind    = 'default'   ! rectgrid kmt_type, or 'file'
  character(len=32)  :: icecase     = 'full'      ! 'full' | 'caps' | 'patchy'
  character(len=256) :: dumpfile    = 'dump.bin'
  character(len=256) :: h_grid_file = 'grid.bin'
  character(len=256) :: h_kmt_file  = 'kmt.bin'
  real(dbl_kind)     :: h_dxrect    = 16.e5_dbl_kind   ! cm
  real(dbl_kind)     :: h_dyrect    = 16.e5_dbl_kind   ! cm
  real(dbl_kind)     :: h_dt        = 3600._dbl_kind
  integer(int_kind)  :: h_ndte      = 120
  integer(int_kind)  :: ncalls      = 1
  integer(int_kind)  :: nsub_list(8) = -1         ! subcycle counts to dump per call (last one advances state)
  logical            :: h_revised   = .false.
  real(dbl_kind)     :: h_arlx      = 300._dbl_kind
  real(dbl_kind)     :: h_brlx      = 300._dbl_kind
  real(dbl_kind)     :: h_capping   = 1._dbl_kind
  real(dbl_kind)     :: h_Ktens     = 0._dbl_kind
  real(dbl_kind)     :: h_e_yield   = 2._dbl_kind
  real(dbl_kind)     :: h_e_plast   = 2._dbl_kind
  real(dbl_kind)     :: h_elasticDamp = 0.36_dbl_kind
  character(len=32)  :: h_coriolis  = 'latitude'
  logical            :: h_seabed    = .false.
  character(len=16)  :: h_seabed_method = 'LKD'   ! 'LKD' | 'probabilistic' (seabed_stress_factor_prob, ice_dyn_shared.F90:1475-1683)
  logical            :: dump_arrays = .true.      ! .false. = timing-only run (no array dumps)
  integer(int_kind)  :: ntiming     = 0           ! extra evp() calls, timed, after the dumps
  logical            :: hipmode     = .false.     ! drop-in check: HIP core (via ice_dyn_evp1d) vs standard_2d
  logical            :: hipbody     = .false.     ! with hipmode: also Option A, preparation + loop on the device
  logical            :: h_evolve    = .false.     ! C grid: the ice cover changes before the second call (cells gain / lose ice)
  character(len=16)  :: h_ssh       = 'geostrophic' ! ssh_stress; 'coupled' also gives the sea surface a slope
  logical            :: hipresident = .false.     ! with hipmode: opt in to device-resident stresses and install the two hooks
                                                  ! (default: NO hook is called -- the unpatched-host contract of dyn_evp1d_run)
  logical            :: time_1d     = .false.     ! timing loop with evp_algorithm='shared_mem_1d' (HARNESS_REF1D build only)
  character(len=8)   :: h_grid_ice  = 'B'         ! 'B' | 'C': staggering of the dynamics (C: ice_dyn_evp.F90:936-1121)
  character(len=16)  :: h_visc_method = 'avg_zeta' ! C grid: 'avg_zeta' | 'avg_strength' (ice_dyn_evp.F90:992-996)

  namelist /harness_nml/ grid_kind, kmt_kind, icecase, dumpfile, h_grid_file, h_kmt_file, &
     h_dxrect, h_dyrect, h_dt, h_ndte, ncalls, nsub_list, h_revised, h_arlx, h_brlx, &
     h_capping, h_Ktens, h_e_yield, h_e_plast, h_elasticDamp, h_coriolis, h_seabed, h_seabed_method, &
     dump_arrays, ntiming, hipmode, hipbody, hipresident, time_1d, h_grid_ice, h_visc_method, h_evolve, h_ssh

  ! ---- locals ----------------------------------------------------------
  integer(int_kind) :: i, j, iblk, icall, k, nsub, nl, ios, nthreads
  type(block) :: tb
  real(dbl_kind) :: x, y, hi, taper, rhow_l
  real(dbl_kind), parameter :: twopi = 6.283185307179586_dbl_kind
  real(dbl_kind), allocatable, dimension(:,:,:) :: s_u, s_v, s_sp1, s_sp2, s_sp3, s_sp4, &
     s_sm1, s_sm2, s_sm3, s_sm4, s_s121, s_s122, s_s123, s_s124
  integer(int_kind), allocatable :: blkinfo(:)
  real(dbl_kind) :: scal(32)
  character(len=16) :: tag
  integer(kind=8) :: c0_clk, c1_clk, crate

  open(unit=55, file='harness_in', status='old', iostat=ios)
  if (ios /= 0) stop 'harness_in missing'
  read(55, nml=harness_nml)
  close(55)
#ifdef HARNESS_HIP_BODY
  if (hipresident) call dyn_evp_hip_keep_stresses_resident(.true.)
#endif

  call init_communicate
  ! several MPI tasks (the mpi* builds under mpiexec): every task dumps its own blocks; blkinfo says where they lie
  if (get_num_procs() > 1) then
     write(tag,'(i0)') my_task
     dumpfile = trim(dumpfile)//'.'//trim(tag)
  endif
  call init_fileunits
  nml_filename = 'ice_in'

  ncat=1; nfsd=1; nilyr=1; nslyr=1; nblyr=1
  n_iso=0; n_aero=0; n_zaero=0; n_algae=0; n_doc=0; n_dic=0; n_don=0; n_fed=0; n_fep=0
  nfreq=1

  grid_format='bin'; grid_ice=trim(h_grid_ice); grid_atm='A'; grid_ocn='A'
  grid_atm_thrm='T'; grid_atm_dynu='T'; grid_atm_dynv='T'
  grid_ocn_thrm='T'; grid_ocn_dynu='T'; grid_ocn_dynv='T'
  dxrect=h_dxrect; dyrect=h_dyrect; scale_dxdy=.false.
  lonrefrect=-156.5_dbl_kind; latrefrect=71.35_dbl_kind
  bathymetry_format='default'; use_bathymetry=.false.
  if (trim(grid_kind) == 'rect') then
     grid_type='rectangular'
     kmt_type=trim(kmt_kind)
  else
     if (trim(grid_kind) == 'tripolefile') then
        grid_type='tripole'
     else
        grid_type='displaced_pole'
     endif
     grid_file=trim(h_grid_file)
     kmt_file=trim(h_kmt_file)
     kmt_type='file'
  endif

#ifdef HARNESS_REF1D
  save_ghte_ghtn = .true.      ! the 1-d core gathers the global HTE/HTN (ice_init.F90:1464-1466 sets this for it)
#endif
  kdyn=1; ndte=h_ndte; revised_evp=h_revised; evp_algorithm='standard_2d'
  elasticDamp=h_elasticDamp
  e_yieldcurve=h_e_yield; e_plasticpot=h_e_plast; Ktens=h_Ktens
  deltaminEVP=1e-11_dbl_kind; capping=h_capping
  coriolis=trim(h_coriolis); ssh_stress=trim(h_ssh)
  seabed_stress=h_seabed; seabed_stress_method=trim(h_seabed_method)
  k1=7.5_dbl_kind; k2=15._dbl_kind; alphab=20._dbl_kind; threshold_hw=30._dbl_kind   ! ice_in defaults
  dyn_area_min=1e-11_dbl_kind; dyn_mass_min=1e-10_dbl_kind
  yield_curve='ellipse'; visc_method=trim(h_visc_method)
  arlx=h_arlx; brlx=h_brlx
  dt=h_dt; ndtd=1; dt_dyn=dt

  call init_domain_blocks
  call init_grid1
  call alloc_grid
  call alloc_arrays_column
  call alloc_state
  call alloc_flux_bgc
  call alloc_flux
  call init_ice_timers
  call init_grid2
  call init_evp
  call dyn_evp1d_init          ! no-op in the capture build; HIP core set-up in the drop-in build

  ! ---- fill the model state that evp() reads ----------------------------
  do iblk = 1, nblocks
     tb = get_block(blocks_ice(iblk), iblk)
     do j = 1, ny_block
     do i = 1, nx_block
        x = (real(tb%i_glob(i),dbl_kind) - p5)/real(nx_global,dbl_kind)
        y = (real(tb%j_glob(j),dbl_kind) - p5)/real(ny_global,dbl_kind)
        aice(i,j,iblk) = c0; vice(i,j,iblk) = c0; vsno(i,j,iblk) = c0
        if (tmask(i,j,iblk)) then
           hi = c2*(c1 + 0.1_dbl_kind*sin(c2*twopi*y)*cos(twopi*x))
           select case (trim(icecase))
           case ('full')
              aice(i,j,iblk) = 0.9_dbl_kind + 0.05_dbl_kind*sin(twopi*x)*cos(twopi*y)
           case ('caps')
              taper = (abs(y - p5) - p25)/0.05_dbl_kind
              taper = max(c0, min(c1, taper))
              aice(i,j,iblk) = 0.95_dbl_kind*taper
           case ('patchy')
              aice(i,j,iblk) = 0.85_dbl_kind + 0.1_dbl_kind*sin(twopi*x)*cos(twopi*y)
              if (sin(5.5_dbl_kind*twopi*x)*sin(3.5_dbl_kind*twopi*y) > 0.6_dbl_kind) aice(i,j,iblk) = c0
           case default
              stop 'unknown icecase'
           end select
           vice(i,j,iblk) = hi*aice(i,j,iblk)
           vsno(i,j,iblk) = 0.2_dbl_kind*aice(i,j,iblk)
        endif
        aice0(i,j,iblk)   = c1 - aice(i,j,iblk)
        aicen(i,j,1,iblk) = aice(i,j,iblk)
        vicen(i,j,1,iblk) = vice(i,j,iblk)
        aice_init(i,j,iblk) = aice(i,j,iblk)
        Cdn_ocn(i,j,iblk) = 0.00536_dbl_kind
        uocn(i,j,iblk) =  0.2_dbl_kind*y - 0.1_dbl_kind
        vocn(i,j,iblk) = -0.2_dbl_kind*x + 0.1_dbl_kind
        ss_tltx(i,j,iblk) = c0; ss_tlty(i,j,iblk) = c0
        if (trim(h_ssh) == 'coupled') then
           ss_tltx(i,j,iblk) = 2.e-6_dbl_kind*sin(twopi*x)*cos(c2*twopi*y)
           ss_tlty(i,j,iblk) = -1.5e-6_dbl_kind*cos(twopi*x)*sin(twopi*y)
        endif
        strairxT(i,j,iblk) = aice(i,j,iblk)*0.1_dbl_kind*sin(twopi*x)*sin(p5*twopi*y)
        strairyT(i,j,iblk) = aice(i,j,iblk)*0.1_dbl_kind*sin(p5*twopi*x)*sin(twopi*y)
        if (h_seabed) hwater(i,j,iblk) = 8._dbl_kind + 40._dbl_kind*y   ! shallow shelf in the south
     enddo
     enddo
  enddo
  ! consistent ghosts for every boundary type (cyclic, closed, tripole)
  call ice_HaloUpdate(aice,      halo_info, field_loc_center, field_type_scalar)
  call ice_HaloUpdate(vice,      halo_info, field_loc_center, field_type_scalar)
  call ice_HaloUpdate(vsno,      halo_info, field_loc_center, field_type_scalar)
  call ice_HaloUpdate(aice0,     halo_info, field_loc_center, field_type_scalar)
  call ice_HaloUpdate(aice_init, halo_info, field_loc_center, field_type_scalar)
  do iblk = 1, nblocks
     aicen(:,:,1,iblk) = aice(:,:,iblk)
     vicen(:,:,1,iblk) = vice(:,:,iblk)
  enddo

  call icepack_query_parameters(rhow_out=rhow_l)

  ! ---- static dump --------------------------------------------------------
  call dump_begin(trim(dumpfile))
  allocate(blkinfo(8*nblocks))
  do iblk = 1, nblocks
     tb = get_block(blocks_ice(iblk), iblk)
     blkinfo(8*(iblk-1)+1:8*iblk) = (/ tb%ilo, tb%ihi, tb%jlo, tb%jhi, tb%iblock, tb%jblock, &
                                       tb%i_glob(tb%ilo), tb%j_glob(tb%jlo) /)
  enddo
  nl = 0
  do k = 1, 8
     if (nsub_list(k) > 0) nl = nl + 1
  enddo
  if (nl == 0) then
     nl = 1; nsub_list(1) = h_ndte
  endif
  call dump_i4_1d('dims', (/ nx_block, ny_block, nblocks, nghost, nx_global, ny_global, &
                             h_ndte, ncalls, nl /))
  call dump_i4_1d('blkinfo', blkinfo)
  call dump_i4_1d('nsub_list', nsub_list(1:nl))
  scal = c0
  scal(1)=arlx1i; scal(2)=denom1; scal(3)=brlx; scal(4)=revp; scal(5)=e_factor; scal(6)=epp2i
  scal(7)=capping; scal(8)=Ktens; scal(9)=deltaminEVP; scal(10)=u0; scal(11)=cosw; scal(12)=sinw
  scal(13)=rhow_l; scal(14)=dt; scal(15)=arlx; scal(16)=dtei; scal(17)=ecci
  ! what the pre-subcycle preparation (dyn_prep1/2, seabed stress) reads besides the fields
  call icepack_query_parameters(rhoi_out=scal(18), rhos_out=scal(19), gravit_out=scal(20))
  scal(21)=dyn_area_min; scal(22)=dyn_mass_min
  scal(23)=c0; if (trim(ssh_stress) == 'coupled') scal(23)=c1
  scal(24)=c0; if (seabed_stress) scal(24)=c1
  scal(25)=k1; scal(26)=k2; scal(27)=alphab; scal(28)=threshold_hw; scal(29)=dt_dyn
  scal(30)=c0; if (trim(seabed_stress_method) == 'probabilistic') scal(30)=c1
  call icepack_query_parameters(pi_out=scal(31), puny_out=scal(32))
  call dump_r8_1d('scalars', scal)
  if (dump_arrays) then
     call dump_r8_3d('HTE', HTE, nblocks);       call dump_r8_3d('HTN', HTN, nblocks)
     call dump_r8_3d('dxT', dxT, nblocks);       call dump_r8_3d('dyT', dyT, nblocks)
     call dump_r8_3d('dxU', dxU, nblocks);       call dump_r8_3d('dyU', dyU, nblocks)
     call dump_r8_3d('tarea', tarea, nblocks);   call dump_r8_3d('uarear', uarear, nblocks)
     call dump_r8_3d('tarear', tarear, nblocks); call dump_r8_3d('hm', hm, nblocks)
     call dump_r8_3d('uvm', uvm, nblocks)
     call dump_l_3d ('tmask', tmask, nblocks);   call dump_l_3d ('umask', umask, nblocks)
     call dump_r8_3d('cxp', cxp, nblocks);       call dump_r8_3d('cyp', cyp, nblocks)
     call dump_r8_3d('cxm', cxm, nblocks);       call dump_r8_3d('cym', cym, nblocks)
     if (allocated(dxhy)) then      ! B grid only (ice_dyn_shared.F90:229)
        call dump_r8_3d('dxhy', dxhy, nblocks);     call dump_r8_3d('dyhx', dyhx, nblocks)
     endif
     call dump_r8_3d('DminTarea', DminTarea, nblocks)
     call dump_r8_3d('ULAT', ULAT, nblocks)
     call dump_r8_3d('uarea', uarea, nblocks);   call dump_r8_3d('fcor_blk', fcor_blk, nblocks)
     call dump_r8_3d('hwater', hwater, nblocks)
     if (trim(grid_ice) == 'C') then
        call dump_r8_3d('dxE', dxE, nblocks);       call dump_r8_3d('dyE', dyE, nblocks)
        call dump_r8_3d('dxN', dxN, nblocks);       call dump_r8_3d('dyN', dyN, nblocks)
        call dump_r8_3d('earea', earea, nblocks);   call dump_r8_3d('narea', narea, nblocks)
        call dump_r8_3d('earear', earear, nblocks); call dump_r8_3d('narear', narear, nblocks)
        call dump_r8_3d('epm', epm, nblocks);       call dump_r8_3d('npm', npm, nblocks)
        call dump_r8_3d('ratiodxN', ratiodxN, nblocks);   call dump_r8_3d('ratiodxNr', ratiodxNr, nblocks)
        call dump_r8_3d('ratiodyE', ratiodyE, nblocks);   call dump_r8_3d('ratiodyEr', ratiodyEr, nblocks)
        call dump_l_3d ('umaskCD', umaskCD, nblocks)
        call dump_l_3d ('emask', emask, nblocks);         call dump_l_3d ('nmask', nmask, nblocks)
        call dump_r8_3d('fcorE_blk', fcorE_blk, nblocks); call dump_r8_3d('fcorN_blk', fcorN_blk, nblocks)
     endif
  endif

  allocate(s_u(nx_block,ny_block,max_blocks), s_v(nx_block,ny_block,max_blocks))
  allocate(s_sp1(nx_block,ny_block,max_blocks), s_sp2(nx_block,ny_block,max_blocks), &
           s_sp3(nx_block,ny_block,max_blocks), s_sp4(nx_block,ny_block,max_blocks), &
           s_sm1(nx_block,ny_block,max_blocks), s_sm2(nx_block,ny_block,max_blocks), &
           s_sm3(nx_block,ny_block,max_blocks), s_sm4(nx_block,ny_block,max_blocks), &
           s_s121(nx_block,ny_block,max_blocks), s_s122(nx_block,ny_block,max_blocks), &
           s_s123(nx_block,ny_block,max_blocks), s_s124(nx_block,ny_block,max_blocks))

  ! ---- evp calls ------------------------------------------------------------
  do icall = 1, ncalls

     if (trim(grid_ice) == 'C') then
        call cgrid_call(icall, nsub_list, nl, h_ndte, hipmode, h_evolve, hipbody)
        cycle
     endif
     if (h_evolve .and. icall == 2) call evolve_ice()

     if (hipmode) then
        ! prep-only pass (ndte=0): applies dyn_prep2's one-off state changes (new-ice
        ! velocities, stresses zeroed off the ice) so that both cores start identically
        ndte = 0
        call evp(dt_dyn)
        ndte = h_ndte
     else
     ! (0) the model state evp() is entered with: inputs of its preparation phase (SURVEY 8 f-2)
     if (dump_arrays) then
        write(tag,'(a,i2.2)') 'pr', icall
        call dump_r8_3d(trim(tag)//'_aice', aice, nblocks);       call dump_r8_3d(trim(tag)//'_vice', vice, nblocks)
        call dump_r8_3d(trim(tag)//'_vsno', vsno, nblocks);       call dump_r8_3d(trim(tag)//'_aice_init', aice_init, nblocks)
        call dump_r8_3d(trim(tag)//'_cdn_ocn', Cdn_ocn, nblocks)
        call dump_r8_3d(trim(tag)//'_uocn', uocn, nblocks);       call dump_r8_3d(trim(tag)//'_vocn', vocn, nblocks)
        call dump_r8_3d(trim(tag)//'_ss_tltx', ss_tltx, nblocks); call dump_r8_3d(trim(tag)//'_ss_tlty', ss_tlty, nblocks)
        call dump_r8_3d(trim(tag)//'_strairxT', strairxT, nblocks); call dump_r8_3d(trim(tag)//'_strairyT', strairyT, nblocks)
        call dump_r8_3d(trim(tag)//'_uvel', uvel, nblocks);       call dump_r8_3d(trim(tag)//'_vvel', vvel, nblocks)
        call dump_l_3d (trim(tag)//'_iceUmask', iceUmask, nblocks)
        call dump_r8_3d(trim(tag)//'_stressp_1', stressp_1, nblocks); call dump_r8_3d(trim(tag)//'_stressp_2', stressp_2, nblocks)
        call dump_r8_3d(trim(tag)//'_stressp_3', stressp_3, nblocks); call dump_r8_3d(trim(tag)//'_stressp_4', stressp_4, nblocks)
        call dump_r8_3d(trim(tag)//'_stressm_1', stressm_1, nblocks); call dump_r8_3d(trim(tag)//'_stressm_2', stressm_2, nblocks)
        call dump_r8_3d(trim(tag)//'_stressm_3', stressm_3, nblocks); call dump_r8_3d(trim(tag)//'_stressm_4', stressm_4, nblocks)
        call dump_r8_3d(trim(tag)//'_stress12_1', stress12_1, nblocks); call dump_r8_3d(trim(tag)//'_stress12_2', stress12_2, nblocks)
        call dump_r8_3d(trim(tag)//'_stress12_3', stress12_3, nblocks); call dump_r8_3d(trim(tag)//'_stress12_4', stress12_4, nblocks)
     endif
#ifndef HARNESS_REF1D
     ! (1) capture the subcycle inputs at the boundary (computes nothing)
     write(tag,'(a,i2.2)') 'in', icall
     capture_tag = tag
     evp_algorithm = 'shared_mem_1d'
     if (.not. dump_arrays) call dump_end
     call evp(dt_dyn)
     evp_algorithm = 'standard_2d'
#endif
     endif

     ! other products of the preparation phase that the subcycle boundary does not carry
     if (dump_arrays .and. .not. hipmode) then
        write(tag,'(a,i2.2)') 'pq', icall
        call dump_r8_3d(trim(tag)//'_strtltxU', strtltxU, nblocks); call dump_r8_3d(trim(tag)//'_strtltyU', strtltyU, nblocks)
        call dump_r8_3d(trim(tag)//'_strairxU', strairxU, nblocks); call dump_r8_3d(trim(tag)//'_strairyU', strairyU, nblocks)
        call dump_r8_3d(trim(tag)//'_strocnxU', strocnxU, nblocks); call dump_r8_3d(trim(tag)//'_strocnyU', strocnyU, nblocks)
     endif
     ! post-prep state = the captured inputs
     s_u = uvel; s_v = vvel
     s_sp1 = stressp_1; s_sp2 = stressp_2; s_sp3 = stressp_3; s_sp4 = stressp_4
     s_sm1 = stressm_1; s_sm2 = stressm_2; s_sm3 = stressm_3; s_sm4 = stressm_4
     s_s121 = stress12_1; s_s122 = stress12_2; s_s123 = stress12_3; s_s124 = stress12_4

     ! (2) the reference's answer for those inputs after nsub subcycles
     do k = 1, nl
        nsub = nsub_list(k)
        uvel = s_u; vvel = s_v
        stressp_1 = s_sp1; stressp_2 = s_sp2; stressp_3 = s_sp3; stressp_4 = s_sp4
        stressm_1 = s_sm1; stressm_2 = s_sm2; stressm_3 = s_sm3; stressm_4 = s_sm4
        stress12_1 = s_s121; stress12_2 = s_s122; stress12_3 = s_s123; stress12_4 = s_s124
        ndte = nsub                 ! loop count only; EVP scalars were fixed by init_evp
        if (hipmode) then
           ! the reference's unmodified evp() driving the HIP core through the
           ! existing dyn_evp1d_run boundary (ice_dyn_evp.F90:846-856)
           evp_algorithm = 'shared_mem_1d'
#ifdef HARNESS_HIP_BODY
           ! only a host that opted in to resident stresses has hooks to call: the harness reset ice_flux's stresses
           ! behind the core's back.  Default: nothing -- the host arrays ARE the state, as with dyn_evp1d_run
           if (hipresident) call dyn_evp_hip_invalidate_stresses
#endif
           call evp(dt_dyn)
#ifdef HARNESS_HIP_BODY
           if (hipresident) call dyn_evp_hip_fetch_stresses   ! opted in: ice_flux's arrays are stale until fetched
#endif
           evp_algorithm = 'standard_2d'
           write(tag,'(a,i2.2,a,i4.4)') 'h', icall, 'n', nsub
           call dump_r8_3d(trim(tag)//'_uvel', uvel, nblocks)
           call dump_r8_3d(trim(tag)//'_vvel', vvel, nblocks)
           call dump_r8_3d(trim(tag)//'_stressp_1', stressp_1, nblocks)
           call dump_r8_3d(trim(tag)//'_stressp_2', stressp_2, nblocks)
           call dump_r8_3d(trim(tag)//'_stressp_3', stressp_3, nblocks)
           call dump_r8_3d(trim(tag)//'_stressp_4', stressp_4, nblocks)
           call dump_r8_3d(trim(tag)//'_stressm_1', stressm_1, nblocks)
           call dump_r8_3d(trim(tag)//'_stressm_2', stressm_2, nblocks)
           call dump_r8_3d(trim(tag)//'_stressm_3', stressm_3, nblocks)
           call dump_r8_3d(trim(tag)//'_stressm_4', stressm_4, nblocks)
           call dump_r8_3d(trim(tag)//'_stress12_1', stress12_1, nblocks)
           call dump_r8_3d(trim(tag)//'_stress12_2', stress12_2, nblocks)
           call dump_r8_3d(trim(tag)//'_stress12_3', stress12_3, nblocks)
           call dump_r8_3d(trim(tag)//'_stress12_4', stress12_4, nblocks)
           call dump_r8_3d(trim(tag)//'_strintxU', strintxU, nblocks)
           call dump_r8_3d(trim(tag)//'_strintyU', strintyU, nblocks)
           call dump_r8_3d(trim(tag)//'_taubxU', taubxU, nblocks)
           call dump_r8_3d(trim(tag)//'_taubyU', taubyU, nblocks)
           call dump_r8_3d(trim(tag)//'_divu', divu, nblocks)
           call dump_r8_3d(trim(tag)//'_shear', shear, nblocks)
           call dump_r8_3d(trim(tag)//'_strocnxU', strocnxU, nblocks)
#ifdef HARNESS_HIP_BODY
           if (hipbody) then
              ! Option A: evp()'s preparation phase + loop (+ tripole stress symmetrisation) through
              ! dyn_evp_hip_evp_body, ice strength by callback -- from the same state
              uvel = s_u; vvel = s_v
              stressp_1 = s_sp1; stressp_2 = s_sp2; stressp_3 = s_sp3; stressp_4 = s_sp4
              stressm_1 = s_sm1; stressm_2 = s_sm2; stressm_3 = s_sm3; stressm_4 = s_sm4
              stress12_1 = s_s121; stress12_2 = s_s122; stress12_3 = s_s123; stress12_4 = s_s124
              TbU = 777._dbl_kind       ! evp_body must not depend on what a previous evp() left here
              ! (a host that opted in to resident stresses tells the core that it rewrote them, and fetches them afterwards)
              if (hipresident) call dyn_evp_hip_invalidate_stresses
              call dyn_evp_hip_evp_body(dt_dyn, harness_strength)
              if (hipresident) call dyn_evp_hip_fetch_stresses
              write(tag,'(a,i2.2,a,i4.4)') 'b', icall, 'n', nsub
              call dump_r8_3d(trim(tag)//'_uvel', uvel, nblocks)
              call dump_r8_3d(trim(tag)//'_vvel', vvel, nblocks)
              call dump_r8_3d(trim(tag)//'_stressp_1', stressp_1, nblocks)
              call dump_r8_3d(trim(tag)//'_stressp_2', stressp_2, nblocks)
              call dump_r8_3d(trim(tag)//'_stressp_3', stressp_3, nblocks)
              call dump_r8_3d(trim(tag)//'_stressp_4', stressp_4, nblocks)
              call dump_r8_3d(trim(tag)//'_stressm_1', stressm_1, nblocks)
              call dump_r8_3d(trim(tag)//'_stressm_2', stressm_2, nblocks)
              call dump_r8_3d(trim(tag)//'_stressm_3', stressm_3, nblocks)
              call dump_r8_3d(trim(tag)//'_stressm_4', stressm_4, nblocks)
              call dump_r8_3d(trim(tag)//'_stress12_1', stress12_1, nblocks)
              call dump_r8_3d(trim(tag)//'_stress12_2', stress12_2, nblocks)
              call dump_r8_3d(trim(tag)//'_stress12_3', stress12_3, nblocks)
              call dump_r8_3d(trim(tag)//'_stress12_4', stress12_4, nblocks)
              call dump_r8_3d(trim(tag)//'_strintxU', strintxU, nblocks)
              call dump_r8_3d(trim(tag)//'_strintyU', strintyU, nblocks)
              call dump_r8_3d(trim(tag)//'_taubxU', taubxU, nblocks)
              call dump_r8_3d(trim(tag)//'_taubyU', taubyU, nblocks)
              if (hipresident) then
                 ! resident stresses, two evp() bodies in a row: the second one neither uploads nor downloads the 12 stresses
                 ! (they are where the first one left them); against the reference's evp() called twice from the same state
                 uvel = s_u; vvel = s_v
                 stressp_1 = s_sp1; stressp_2 = s_sp2; stressp_3 = s_sp3; stressp_4 = s_sp4
                 stressm_1 = s_sm1; stressm_2 = s_sm2; stressm_3 = s_sm3; stressm_4 = s_sm4
                 stress12_1 = s_s121; stress12_2 = s_s122; stress12_3 = s_s123; stress12_4 = s_s124
                 call dyn_evp_hip_invalidate_stresses
                 call dyn_evp_hip_evp_body(dt_dyn, harness_strength)
                 stressp_1 = -9.e9_dbl_kind; stress12_3 = -9.e9_dbl_kind    ! stale on purpose: nobody may read the host copies now
                 call dyn_evp_hip_evp_body(dt_dyn, harness_strength)
                 call dyn_evp_hip_fetch_stresses
                 write(tag,'(a,i2.2,a,i4.4)') 'c', icall, 'n', nsub
                 call dump_r8_3d(trim(tag)//'_uvel', uvel, nblocks)
                 call dump_r8_3d(trim(tag)//'_vvel', vvel, nblocks)
                 call dump_r8_3d(trim(tag)//'_stressp_1', stressp_1, nblocks)
                 call dump_r8_3d(trim(tag)//'_stressm_2', stressm_2, nblocks)
                 call dump_r8_3d(trim(tag)//'_stress12_3', stress12_3, nblocks)
                 call dump_r8_3d(trim(tag)//'_stress12_4', stress12_4, nblocks)
                 uvel = s_u; vvel = s_v
                 stressp_1 = s_sp1; stressp_2 = s_sp2; stressp_3 = s_sp3; stressp_4 = s_sp4
                 stressm_1 = s_sm1; stressm_2 = s_sm2; stressm_3 = s_sm3; stressm_4 = s_sm4
                 stress12_1 = s_s121; stress12_2 = s_s122; stress12_3 = s_s123; stress12_4 = s_s124
                 call evp(dt_dyn)
                 call evp(dt_dyn)
                 write(tag,'(a,i2.2,a,i4.4)') 'p', icall, 'n', nsub
                 call dump_r8_3d(trim(tag)//'_uvel', uvel, nblocks)
                 call dump_r8_3d(trim(tag)//'_vvel', vvel, nblocks)
                 call dump_r8_3d(trim(tag)//'_stressp_1', stressp_1, nblocks)
                 call dump_r8_3d(trim(tag)//'_stressm_2', stressm_2, nblocks)
                 call dump_r8_3d(trim(tag)//'_stress12_3', stress12_3, nblocks)
                 call dump_r8_3d(trim(tag)//'_stress12_4', stress12_4, nblocks)
              endif
           endif
#endif
           uvel = s_u; vvel = s_v
           stressp_1 = s_sp1; stressp_2 = s_sp2; stressp_3 = s_sp3; stressp_4 = s_sp4
           stressm_1 = s_sm1; stressm_2 = s_sm2; stressm_3 = s_sm3; stressm_4 = s_sm4
           stress12_1 = s_s121; stress12_2 = s_s122; stress12_3 = s_s123; stress12_4 = s_s124
        endif
        call evp(dt_dyn)
        ndte = h_ndte
        if (dump_arrays) then
           write(tag,'(a,i2.2,a,i4.4)') 'o', icall, 'n', nsub
           call dump_r8_3d(trim(tag)//'_uvel', uvel, nblocks)
           call dump_r8_3d(trim(tag)//'_vvel', vvel, nblocks)
           call dump_r8_3d(trim(tag)//'_stressp_1', stressp_1, nblocks)
           call dump_r8_3d(trim(tag)//'_stressp_2', stressp_2, nblocks)
           call dump_r8_3d(trim(tag)//'_stressp_3', stressp_3, nblocks)
           call dump_r8_3d(trim(tag)//'_stressp_4', stressp_4, nblocks)
           call dump_r8_3d(trim(tag)//'_stressm_1', stressm_1, nblocks)
           call dump_r8_3d(trim(tag)//'_stressm_2', stressm_2, nblocks)
           call dump_r8_3d(trim(tag)//'_stressm_3', stressm_3, nblocks)
           call dump_r8_3d(trim(tag)//'_stressm_4', stressm_4, nblocks)
           call dump_r8_3d(trim(tag)//'_stress12_1', stress12_1, nblocks)
           call dump_r8_3d(trim(tag)//'_stress12_2', stress12_2, nblocks)
           call dump_r8_3d(trim(tag)//'_stress12_3', stress12_3, nblocks)
           call dump_r8_3d(trim(tag)//'_stress12_4', stress12_4, nblocks)
           call dump_r8_3d(trim(tag)//'_strintxU', strintxU, nblocks)
           call dump_r8_3d(trim(tag)//'_strintyU', strintyU, nblocks)
           call dump_r8_3d(trim(tag)//'_taubxU', taubxU, nblocks)
           call dump_r8_3d(trim(tag)//'_taubyU', taubyU, nblocks)
           call dump_r8_3d(trim(tag)//'_divu', divu, nblocks)
           call dump_r8_3d(trim(tag)//'_shear', shear, nblocks)
           call dump_r8_3d(trim(tag)//'_vort', vort, nblocks)
           call dump_r8_3d(trim(tag)//'_rdg_conv', rdg_conv, nblocks)
           call dump_r8_3d(trim(tag)//'_rdg_shear', rdg_shear, nblocks)
           call dump_r8_3d(trim(tag)//'_strocnxU', strocnxU, nblocks)
           call dump_r8_3d(trim(tag)//'_strocnyU', strocnyU, nblocks)
        endif
        write(*,'(a,i3,a,i5,3es24.16)') 'call', icall, ' nsub', nsub, &
             maxval(abs(uvel(:,:,1:nblocks))), maxval(abs(vvel(:,:,1:nblocks))), &
             maxval(abs(stressp_1(:,:,1:nblocks)))
     enddo
  enddo
  call dump_end
  call dyn_evp1d_finalize

  ! ---- timing of the reference subcycle loop (its own timer_evp) -----------
  if (ntiming > 0) then
     nthreads = 1
#if defined (_OPENMP)
     nthreads = omp_get_max_threads()
#endif
#ifdef HARNESS_REF1D
     if (time_1d) then
        evp_algorithm = 'shared_mem_1d'
        call evp(dt_dyn)           ! first call of the 1-d core builds its index lists: not timed
     endif
#endif
     call ice_timer_clear(timer_evp)
     call system_clock(c0_clk, crate)
     do k = 1, ntiming
        call evp(dt_dyn)
     enddo
     call system_clock(c1_clk)
     write(*,'(a,i4,a,i6,a,i4,a,es14.6)') 'TIMING threads', nthreads, ' ndte', h_ndte, &
          ' calls', ntiming, ' wall_s_total_evp_calls', real(c1_clk-c0_clk,dbl_kind)/real(crate,dbl_kind)
     call ice_timer_print_all(stats=.false.)
  endif
  call end_run                 ! MPI_Finalize in the mpi* builds (comm/mpi/ice_exit.F90), nothing in the serial ones

contains

  ! what evp() does for the ice strength between its preparation phase and the loop
  ! (ice_dyn_evp.F90:541-552, 727-728): callback of dyn_evp_hip_evp_body
  subroutine harness_strength()
    use icepack_intfc, only: icepack_ice_strength
    integer :: ib, ii, jj
    type(block) :: bb
    do ib = 1, nblocks
       bb = get_block(blocks_ice(ib), ib)
       strength(:,:,ib) = c0
       do jj = bb%jlo, bb%jhi+1
       do ii = bb%ilo, bb%ihi+1
          if (iceTmask(ii,jj,ib)) &
             call icepack_ice_strength(aice=aice(ii,jj,ib), vice=vice(ii,jj,ib), aice0=aice0(ii,jj,ib), &
                                       aicen=aicen(ii,jj,:,ib), vicen=vicen(ii,jj,:,ib), &
                                       strength=strength(ii,jj,ib))
       enddo
       enddo
    enddo
    call ice_HaloUpdate(strength, halo_info, field_loc_center, field_type_scalar)
  end subroutine harness_strength

end program evp_ref_harness
