!=======================================================================
! TEST INFRASTRUCTURE (oracle/ref) -- never linked into the product.
!
! Minimal stand-in for the *interface module* of Icepack, the un-vendored
! git submodule of CICE (/root/reference/.gitmodules:1-3; CICE 6.6.3 pairs
! with Icepack 1.5.x; the pinned commit is not recorded in the tree).
!
! Nothing here is arithmetic of the EVP subcycle.  The hot path
! (stress / strain_rates / visc_replpress / stepu / halo) is 100 % in
! /root/reference and is compiled from there unmodified.  Icepack supplies
! to that path only
!   * Fortran kind parameters   (cicecore/shared/ice_kinds_mod.F90:13-20)
!   * the scalar rhow           (cicedyn/dynamics/ice_dyn_shared.F90:920)
!   * the warning-buffer hooks  (no-ops here)
! Everything else in this file exists so that the *callers* of the path
! (init_grid2, alloc_state, dyn_prep1/2, ...) link.  `icepack_ice_strength`
! is outside the replaced region: its result (`strength`) is an explicit,
! captured INPUT of every golden fixture, so its formula need not be
! faithful (Hibler 1979 form, kstrength=0 equivalent).
!
! Constants are the standard Icepack defaults (icepack_parameters.F90 of
! Icepack 1.5): rhow=1026, rhoi=917, rhos=330, gravit=9.80616, puny=1e-11.
!=======================================================================
      module icepack_intfc

      implicit none
      public

      integer, parameter :: icepack_char_len      = 80
      integer, parameter :: icepack_char_len_long = 256
      integer, parameter :: icepack_log_kind  = kind(.true.)
      integer, parameter :: icepack_int_kind  = selected_int_kind(6)
      integer, parameter :: icepack_int8_kind = selected_int_kind(13)
      integer, parameter :: icepack_real_kind = selected_real_kind(6)
      integer, parameter :: icepack_dbl_kind  = selected_real_kind(13)
      integer, parameter :: icepack_r16_kind  = selected_real_kind(13)   ! NO_R16 build

      integer, parameter :: icepack_max_iso    = 3
      integer, parameter :: icepack_max_nbtrcr = 1
      integer, parameter :: icepack_max_aero   = 1
      integer, parameter :: icepack_max_algae  = 1
      integer, parameter :: icepack_max_doc    = 1
      integer, parameter :: icepack_max_don    = 1
      integer, parameter :: icepack_max_dic    = 1
      integer, parameter :: icepack_max_fe     = 1
      integer, parameter :: icepack_nspint_3bd = 3

      integer, parameter, private :: dk = icepack_dbl_kind

      contains

!-----------------------------------------------------------------------
      subroutine icepack_warnings_flush(iounit)
      integer, intent(in) :: iounit
      end subroutine icepack_warnings_flush

      logical function icepack_warnings_aborted(instring)
      character(len=*), intent(in), optional :: instring
      icepack_warnings_aborted = .false.
      end function icepack_warnings_aborted

!-----------------------------------------------------------------------
      subroutine icepack_query_parameters(puny_out, pi_out, rad_to_deg_out, &
         secday_out, rhow_out, rhoi_out, rhos_out, gravit_out, bignum_out, &
         Tffresh_out, stefan_boltzmann_out, vonkar_out, zref_out, iceruf_out, &
         dragio_out, calc_strair_out, calc_dragio_out, formdrag_out, &
         skl_bgc_out, solve_zbgc_out, z_tracers_out)
      real(dk), intent(out), optional :: puny_out, pi_out, rad_to_deg_out, &
         secday_out, rhow_out, rhoi_out, rhos_out, gravit_out, bignum_out, &
         Tffresh_out, stefan_boltzmann_out, vonkar_out, zref_out, iceruf_out, &
         dragio_out
      logical, intent(out), optional :: calc_strair_out, calc_dragio_out, &
         formdrag_out, skl_bgc_out, solve_zbgc_out, z_tracers_out
      real(dk), parameter :: pi = 3.14159265358979323846_dk
      if (present(puny_out))       puny_out = 1.0e-11_dk
      if (present(pi_out))         pi_out = pi
      if (present(rad_to_deg_out)) rad_to_deg_out = 180._dk/pi
      if (present(secday_out))     secday_out = 86400._dk
      if (present(rhow_out))       rhow_out = 1026._dk
      if (present(rhoi_out))       rhoi_out = 917._dk
      if (present(rhos_out))       rhos_out = 330._dk
      if (present(gravit_out))     gravit_out = 9.80616_dk
      if (present(bignum_out))     bignum_out = 1.0e30_dk
      if (present(Tffresh_out))    Tffresh_out = 273.15_dk
      if (present(stefan_boltzmann_out)) stefan_boltzmann_out = 567.0e-10_dk
      if (present(vonkar_out))     vonkar_out = 0.4_dk
      if (present(zref_out))       zref_out = 10._dk
      if (present(iceruf_out))     iceruf_out = 0.0005_dk
      if (present(dragio_out))     dragio_out = 0.00536_dk
      if (present(calc_strair_out)) calc_strair_out = .true.
      if (present(calc_dragio_out)) calc_dragio_out = .false.
      if (present(formdrag_out))   formdrag_out = .false.
      if (present(skl_bgc_out))    skl_bgc_out = .false.
      if (present(solve_zbgc_out)) solve_zbgc_out = .false.
      if (present(z_tracers_out))  z_tracers_out = .false.
      end subroutine icepack_query_parameters

      subroutine icepack_init_parameters(thickness_ocn_layer1_in)
      real(dk), intent(in), optional :: thickness_ocn_layer1_in
      end subroutine icepack_init_parameters

!-----------------------------------------------------------------------
      subroutine icepack_query_tracer_sizes(ntrcr_out, max_nbtrcr_out, &
         max_algae_out, max_aero_out, nmodal1_out, nmodal2_out, max_don_out)
      integer, intent(out), optional :: ntrcr_out, max_nbtrcr_out, &
         max_algae_out, max_aero_out, nmodal1_out, nmodal2_out, max_don_out
      if (present(ntrcr_out))      ntrcr_out = 1
      if (present(max_nbtrcr_out)) max_nbtrcr_out = icepack_max_nbtrcr
      if (present(max_algae_out))  max_algae_out = icepack_max_algae
      if (present(max_aero_out))   max_aero_out = icepack_max_aero
      if (present(nmodal1_out))    nmodal1_out = 1
      if (present(nmodal2_out))    nmodal2_out = 1
      if (present(max_don_out))    max_don_out = icepack_max_don
      end subroutine icepack_query_tracer_sizes

      subroutine icepack_query_tracer_flags(tr_iage_out, tr_FY_out, tr_lvl_out, &
         tr_aero_out, tr_pond_out, tr_brine_out, tr_fsd_out, tr_snow_out, &
         tr_iso_out, tr_bgc_Nit_out, tr_bgc_N_out, tr_bgc_DON_out, tr_bgc_C_out, &
         tr_bgc_Am_out, tr_bgc_Sil_out, tr_bgc_DMS_out, tr_bgc_Fe_out, &
         tr_bgc_hum_out, tr_zaero_out)
      logical, intent(out), optional :: tr_iage_out, tr_FY_out, tr_lvl_out, &
         tr_aero_out, tr_pond_out, tr_brine_out, tr_fsd_out, tr_snow_out, &
         tr_iso_out, tr_bgc_Nit_out, tr_bgc_N_out, tr_bgc_DON_out, tr_bgc_C_out, &
         tr_bgc_Am_out, tr_bgc_Sil_out, tr_bgc_DMS_out, tr_bgc_Fe_out, &
         tr_bgc_hum_out, tr_zaero_out
      if (present(tr_iage_out))    tr_iage_out = .false.
      if (present(tr_FY_out))      tr_FY_out = .false.
      if (present(tr_lvl_out))     tr_lvl_out = .false.
      if (present(tr_aero_out))    tr_aero_out = .false.
      if (present(tr_pond_out))    tr_pond_out = .false.
      if (present(tr_brine_out))   tr_brine_out = .false.
      if (present(tr_fsd_out))     tr_fsd_out = .false.
      if (present(tr_snow_out))    tr_snow_out = .false.
      if (present(tr_iso_out))     tr_iso_out = .false.
      if (present(tr_bgc_Nit_out)) tr_bgc_Nit_out = .false.
      if (present(tr_bgc_N_out))   tr_bgc_N_out = .false.
      if (present(tr_bgc_DON_out)) tr_bgc_DON_out = .false.
      if (present(tr_bgc_C_out))   tr_bgc_C_out = .false.
      if (present(tr_bgc_Am_out))  tr_bgc_Am_out = .false.
      if (present(tr_bgc_Sil_out)) tr_bgc_Sil_out = .false.
      if (present(tr_bgc_DMS_out)) tr_bgc_DMS_out = .false.
      if (present(tr_bgc_Fe_out))  tr_bgc_Fe_out = .false.
      if (present(tr_bgc_hum_out)) tr_bgc_hum_out = .false.
      if (present(tr_zaero_out))   tr_zaero_out = .false.
      end subroutine icepack_query_tracer_flags

      subroutine icepack_query_tracer_indices(nt_Tsfc_out, nt_sice_out, &
         nt_qice_out, nt_qsno_out, nt_iage_out, nt_fy_out, nt_alvl_out, &
         nt_vlvl_out, nt_apnd_out, nt_hpnd_out, nt_ipnd_out, nt_fsd_out, &
         nt_aero_out, nt_smice_out, nt_smliq_out, nt_rhos_out, nt_rsnw_out, &
         nt_isosno_out, nt_isoice_out, nt_fbri_out, &
         nlt_bgc_N_out, nlt_bgc_C_out, nlt_bgc_DOC_out, nlt_bgc_DON_out, &
         nlt_bgc_DIC_out, nlt_bgc_Fed_out, nlt_bgc_Fep_out, nlt_zaero_out, &
         nlt_bgc_Nit_out, nlt_bgc_Am_out, nlt_bgc_Sil_out, nlt_bgc_DMSPd_out, &
         nlt_bgc_DMS_out, nlt_bgc_hum_out)
      integer, intent(out), optional :: nt_Tsfc_out, nt_sice_out, &
         nt_qice_out, nt_qsno_out, nt_iage_out, nt_fy_out, nt_alvl_out, &
         nt_vlvl_out, nt_apnd_out, nt_hpnd_out, nt_ipnd_out, nt_fsd_out, &
         nt_aero_out, nt_smice_out, nt_smliq_out, nt_rhos_out, nt_rsnw_out, &
         nt_isosno_out, nt_isoice_out, nt_fbri_out, &
         nlt_bgc_Nit_out, nlt_bgc_Am_out, nlt_bgc_Sil_out, nlt_bgc_DMSPd_out, &
         nlt_bgc_DMS_out, nlt_bgc_hum_out
      integer, intent(out), optional, dimension(:) :: &
         nlt_bgc_N_out, nlt_bgc_C_out, nlt_bgc_DOC_out, nlt_bgc_DON_out, &
         nlt_bgc_DIC_out, nlt_bgc_Fed_out, nlt_bgc_Fep_out, nlt_zaero_out
      if (present(nt_Tsfc_out))   nt_Tsfc_out = 1
      if (present(nt_sice_out))   nt_sice_out = 1
      if (present(nt_qice_out))   nt_qice_out = 1
      if (present(nt_qsno_out))   nt_qsno_out = 1
      if (present(nt_iage_out))   nt_iage_out = 1
      if (present(nt_fy_out))     nt_fy_out = 1
      if (present(nt_alvl_out))   nt_alvl_out = 1
      if (present(nt_vlvl_out))   nt_vlvl_out = 1
      if (present(nt_apnd_out))   nt_apnd_out = 1
      if (present(nt_hpnd_out))   nt_hpnd_out = 1
      if (present(nt_ipnd_out))   nt_ipnd_out = 1
      if (present(nt_fsd_out))    nt_fsd_out = 1
      if (present(nt_aero_out))   nt_aero_out = 1
      if (present(nt_smice_out))  nt_smice_out = 1
      if (present(nt_smliq_out))  nt_smliq_out = 1
      if (present(nt_rhos_out))   nt_rhos_out = 1
      if (present(nt_rsnw_out))   nt_rsnw_out = 1
      if (present(nt_isosno_out)) nt_isosno_out = 1
      if (present(nt_isoice_out)) nt_isoice_out = 1
      if (present(nt_fbri_out))   nt_fbri_out = 1
      if (present(nlt_bgc_N_out))   nlt_bgc_N_out = 1
      if (present(nlt_bgc_C_out))   nlt_bgc_C_out = 1
      if (present(nlt_bgc_DOC_out)) nlt_bgc_DOC_out = 1
      if (present(nlt_bgc_DON_out)) nlt_bgc_DON_out = 1
      if (present(nlt_bgc_DIC_out)) nlt_bgc_DIC_out = 1
      if (present(nlt_bgc_Fed_out)) nlt_bgc_Fed_out = 1
      if (present(nlt_bgc_Fep_out)) nlt_bgc_Fep_out = 1
      if (present(nlt_zaero_out))   nlt_zaero_out = 1
      if (present(nlt_bgc_Nit_out)) nlt_bgc_Nit_out = 1
      if (present(nlt_bgc_Am_out))  nlt_bgc_Am_out = 1
      if (present(nlt_bgc_Sil_out)) nlt_bgc_Sil_out = 1
      if (present(nlt_bgc_DMSPd_out)) nlt_bgc_DMSPd_out = 1
      if (present(nlt_bgc_DMS_out)) nlt_bgc_DMS_out = 1
      if (present(nlt_bgc_hum_out)) nlt_bgc_hum_out = 1
      end subroutine icepack_query_tracer_indices

!-----------------------------------------------------------------------
      subroutine icepack_init_trcr(Tair, Tf, Sprofile, Tprofile, Tsfc, qin, qsn)
      real(dk), intent(in) :: Tair, Tf
      real(dk), dimension(:), intent(in) :: Sprofile, Tprofile
      real(dk), intent(out) :: Tsfc
      real(dk), dimension(:), intent(out) :: qin, qsn
      Tsfc = Tf
      qin(:) = 0._dk
      qsn(:) = 0._dk
      end subroutine icepack_init_trcr

      real(dk) function icepack_liquidus_temperature(Sin)
      real(dk), intent(in) :: Sin
      icepack_liquidus_temperature = -0.054_dk*Sin
      end function icepack_liquidus_temperature

!-----------------------------------------------------------------------
! Outside the replaced region; `strength` is a captured fixture INPUT.
      subroutine icepack_ice_strength(aice, vice, aice0, aicen, vicen, strength)
      real(dk), intent(in) :: aice, vice, aice0
      real(dk), dimension(:), intent(in) :: aicen, vicen
      real(dk), intent(inout) :: strength
      strength = 2.75e4_dk*vice*exp(-20._dk*(1._dk-aice))
      end subroutine icepack_ice_strength

      end module icepack_intfc
