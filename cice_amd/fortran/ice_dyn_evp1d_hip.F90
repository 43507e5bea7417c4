!=======================================================================
! Zero-patch drop-in ("Option B" of INTEGRATION.md).
!
! CICE's evp() already dispatches to an alternative EVP core when
! evp_algorithm = 'shared_mem_1d':
!     call dyn_evp1d_init            ice_dyn_evp.F90:153-155
!     call dyn_evp1d_run(...)        ice_dyn_evp.F90:846-856
! This build-owned module takes the NAME of that core (`ice_dyn_evp1d`, public
! dyn_evp1d_init / dyn_evp1d_run / dyn_evp1d_finalize,
! cicecore/cicedyn/dynamics/ice_dyn_evp1d.F90:25) and forwards to the HIP core,
! so that linking it instead of the reference's ice_dyn_evp1d.F90 puts the
! MI355X kernels behind the unmodified ice_dyn_evp driver -- no edit to any
! reference source.  Limits inherited from the reference's own guards for that
! namelist value: B grid only (ice_init.F90:1537-1540), no tripole
! (ice_dyn_shared.F90:300-304); Option A (a new evp_algorithm value) lifts them.
!
! New code written for this repository.
!=======================================================================
module ice_dyn_evp1d

  use ice_kinds_mod
  use ice_dyn_evp_hip, only: dyn_evp_hip_init, dyn_evp_hip_run, dyn_evp_hip_finalize

  implicit none
  private

  public :: dyn_evp1d_init, dyn_evp1d_run, dyn_evp1d_finalize

  character(len=16), public :: capture_tag = 'hip'   ! keeps the test harness source common

contains

  subroutine dyn_evp1d_init
    call dyn_evp_hip_init
  end subroutine dyn_evp1d_init

  subroutine dyn_evp1d_finalize
    call dyn_evp_hip_finalize
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

    real(kind=dbl_kind)   , dimension(:,:,:), intent(inout), contiguous, target :: &
      L_stressp_1 , L_stressp_2 , L_stressp_3 , L_stressp_4 ,  &
      L_stressm_1 , L_stressm_2 , L_stressm_3 , L_stressm_4 ,  &
      L_stress12_1, L_stress12_2, L_stress12_3, L_stress12_4,  &
      L_strintxU  , L_strintyU  , L_uvel      , L_vvel      ,  &
      L_taubxU    , L_taubyU
    real(kind=dbl_kind)   , dimension(:,:,:), intent(in), contiguous, target ::    &
      L_strength  ,                                            &
      L_cdn_ocn   , L_aiu       , L_uocn     , L_vocn   ,      &
      L_waterxU   , L_wateryU   , L_forcexU  , L_forceyU,      &
      L_umassdti  , L_fmU       , L_Tbu
    logical(kind=log_kind), dimension(:,:,:), intent(in), contiguous, target ::    &
      L_iceUmask  , L_iceTmask

    call dyn_evp_hip_run(L_stressp_1 , L_stressp_2 , L_stressp_3 , L_stressp_4 , &
                         L_stressm_1 , L_stressm_2 , L_stressm_3 , L_stressm_4 , &
                         L_stress12_1, L_stress12_2, L_stress12_3, L_stress12_4, &
                         L_strength,                                             &
                         L_cdn_ocn   , L_aiu       , L_uocn      , L_vocn      , &
                         L_waterxU   , L_wateryU   , L_forcexU   , L_forceyU   , &
                         L_umassdti  , L_fmU       , L_strintxU  , L_strintyU  , &
                         L_Tbu       , L_taubxU    , L_taubyU    , L_uvel      , &
                         L_vvel      , L_icetmask  , L_iceUmask)

  end subroutine dyn_evp1d_run

end module ice_dyn_evp1d
