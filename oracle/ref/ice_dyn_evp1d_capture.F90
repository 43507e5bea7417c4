!=======================================================================
! TEST INFRASTRUCTURE (oracle/ref) -- fixture capture at the drop-in boundary.
!
! A build-owned module that takes the NAME of the reference's alternative
! EVP core (`ice_dyn_evp1d`, public dyn_evp1d_init/run/finalize,
! /root/reference/cicecore/cicedyn/dynamics/ice_dyn_evp1d.F90:25) so that the
! reference's own, unmodified evp() (ice_dyn_evp.F90:846-856) hands it every
! input of the EVP subcycle as arguments -- including waterxU, forcexU,
! umassdti, uocnU, cdn_ocnU ..., which are private to ice_dyn_evp
! (ice_dyn_evp.F90:106-118).  This version only writes them to a dump file;
! it computes nothing and leaves all inout arrays untouched.  The harness
! then re-runs evp() with evp_algorithm='standard_2d' to obtain the golden
! outputs for exactly these inputs (SURVEY.md Appendix A.3, route (b)).
!=======================================================================
module ice_dyn_evp1d

  use ice_kinds_mod
  use evp_dumpio
  implicit none
  private

  public :: dyn_evp1d_init, dyn_evp1d_run, dyn_evp1d_finalize

  character(len=16), public :: capture_tag = 'in'   ! prefix for record names

contains

  subroutine dyn_evp1d_init
  end subroutine dyn_evp1d_init

  subroutine dyn_evp1d_finalize
  end subroutine dyn_evp1d_finalize

  subroutine dyn_evp1d_run(L_stressp_1 , L_stressp_2 , L_stressp_3 , L_stressp_4 , &
                           L_stressm_1 , L_stressm_2 , L_stressm_3 , L_stressm_4 , &
                           L_stress12_1, L_stress12_2, L_stress12_3, L_stress12_4, &
                           L_strength,                                             &
                           L_cdn_ocn   , L_aiu       , L_uocn      , L_vocn      , &
                           L_waterxU   , L_wateryU   , L_forcexU   , L_forceyU   , &
                           L_umassdti  , L_fmU       , L_strintxU  , L_strintyU  , &
                           L_Tbu       , L_taubxU    , L_taubyU    , L_uvel      , &
                           L_vvel      , L_icetmask  , L_iceUmask)

    use ice_domain, only: nblocks
    use ice_dyn_shared, only: uvel_init, vvel_init

    real(kind=dbl_kind)   , dimension(:,:,:), intent(inout) :: &
      L_stressp_1 , L_stressp_2 , L_stressp_3 , L_stressp_4 ,  &
      L_stressm_1 , L_stressm_2 , L_stressm_3 , L_stressm_4 ,  &
      L_stress12_1, L_stress12_2, L_stress12_3, L_stress12_4,  &
      L_strintxU  , L_strintyU  , L_uvel      , L_vvel      ,  &
      L_taubxU    , L_taubyU
    real(kind=dbl_kind)   , dimension(:,:,:), intent(in) ::    &
      L_strength  ,                                            &
      L_cdn_ocn   , L_aiu       , L_uocn     , L_vocn   ,      &
      L_waterxU   , L_wateryU   , L_forcexU  , L_forceyU,      &
      L_umassdti  , L_fmU       , L_Tbu
    logical(kind=log_kind), dimension(:,:,:), intent(in) ::    &
      L_iceUmask  , L_iceTmask

    character(len=16) :: t
    integer(int_kind) :: nb

    if (.not. dump_open) return
    t  = capture_tag
    nb = nblocks

    call dump_r8_3d(trim(t)//'_stressp_1' , L_stressp_1 , nb)
    call dump_r8_3d(trim(t)//'_stressp_2' , L_stressp_2 , nb)
    call dump_r8_3d(trim(t)//'_stressp_3' , L_stressp_3 , nb)
    call dump_r8_3d(trim(t)//'_stressp_4' , L_stressp_4 , nb)
    call dump_r8_3d(trim(t)//'_stressm_1' , L_stressm_1 , nb)
    call dump_r8_3d(trim(t)//'_stressm_2' , L_stressm_2 , nb)
    call dump_r8_3d(trim(t)//'_stressm_3' , L_stressm_3 , nb)
    call dump_r8_3d(trim(t)//'_stressm_4' , L_stressm_4 , nb)
    call dump_r8_3d(trim(t)//'_stress12_1', L_stress12_1, nb)
    call dump_r8_3d(trim(t)//'_stress12_2', L_stress12_2, nb)
    call dump_r8_3d(trim(t)//'_stress12_3', L_stress12_3, nb)
    call dump_r8_3d(trim(t)//'_stress12_4', L_stress12_4, nb)
    call dump_r8_3d(trim(t)//'_strength'  , L_strength  , nb)
    call dump_r8_3d(trim(t)//'_cdn_ocnU'  , L_cdn_ocn   , nb)
    call dump_r8_3d(trim(t)//'_aiU'       , L_aiu       , nb)
    call dump_r8_3d(trim(t)//'_uocnU'     , L_uocn      , nb)
    call dump_r8_3d(trim(t)//'_vocnU'     , L_vocn      , nb)
    call dump_r8_3d(trim(t)//'_waterxU'   , L_waterxU   , nb)
    call dump_r8_3d(trim(t)//'_wateryU'   , L_wateryU   , nb)
    call dump_r8_3d(trim(t)//'_forcexU'   , L_forcexU   , nb)
    call dump_r8_3d(trim(t)//'_forceyU'   , L_forceyU   , nb)
    call dump_r8_3d(trim(t)//'_umassdti'  , L_umassdti  , nb)
    call dump_r8_3d(trim(t)//'_fmU'       , L_fmU       , nb)
    call dump_r8_3d(trim(t)//'_strintxU'  , L_strintxU  , nb)
    call dump_r8_3d(trim(t)//'_strintyU'  , L_strintyU  , nb)
    call dump_r8_3d(trim(t)//'_TbU'       , L_Tbu       , nb)
    call dump_r8_3d(trim(t)//'_taubxU'    , L_taubxU    , nb)
    call dump_r8_3d(trim(t)//'_taubyU'    , L_taubyU    , nb)
    call dump_r8_3d(trim(t)//'_uvel'      , L_uvel      , nb)
    call dump_r8_3d(trim(t)//'_vvel'      , L_vvel      , nb)
    call dump_r8_3d(trim(t)//'_uvel_init' , uvel_init   , nb)
    call dump_r8_3d(trim(t)//'_vvel_init' , vvel_init   , nb)
    call dump_l_3d (trim(t)//'_iceTmask'  , L_iceTmask  , nb)
    call dump_l_3d (trim(t)//'_iceUmask'  , L_iceUmask  , nb)

  end subroutine dyn_evp1d_run

end module ice_dyn_evp1d
