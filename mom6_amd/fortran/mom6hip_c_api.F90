!> ISO_C_BINDING interfaces to libmom6hip (include/mom6hip.h), in the style the reference uses for
!! libc (src/framework/posix.F90:52-229): interface blocks with bind(c, name=...), `value` scalars and
!! c_int returns.  This module has no dependence on any MOM6 module; the shims that present the
!! reference's procedure signatures (MOM_tracer_advect_hip.F90, ...) are built on top of it.
module mom6hip_c_api

use, intrinsic :: iso_c_binding, only : c_int, c_int32_t, c_int64_t, c_double, c_ptr, c_char, &
                                        c_null_ptr, c_null_char, c_loc, c_associated, c_f_pointer, c_funptr
implicit none ; public

integer(c_int32_t), parameter :: MOM6HIP_MEM_HOST = 0, MOM6HIP_MEM_DEVICE = 1
integer(c_int32_t), parameter :: MOM6HIP_ADV_PLM = 0, MOM6HIP_ADV_PPM_H3 = 1, MOM6HIP_ADV_PPM = 2
integer(c_int32_t), parameter :: MOM6HIP_POS_H = 0, MOM6HIP_POS_U = 1, MOM6HIP_POS_V = 2, MOM6HIP_POS_Q = 3
integer(c_int32_t), parameter :: MOM6HIP_PASS_SCALAR_PAIR = 8   !< ORed into the position of a u/v member of a SCALAR_PAIR in a group pass
integer(c_int32_t), parameter :: MOM6HIP_EOS_LINEAR = 1, MOM6HIP_EOS_UNESCO = 2, MOM6HIP_EOS_WRIGHT = 3, &
                                 MOM6HIP_EOS_WRIGHT_FULL = 4, MOM6HIP_EOS_WRIGHT_REDUCED = 5
!> REMAPPING_* of src/ALE/MOM_remapping.F90:51-59 and REGRIDDING_ZSTAR of regrid_consts.F90:14
integer(c_int32_t), parameter :: MOM6HIP_REMAP_PCM = 0, MOM6HIP_REMAP_PLM = 2, MOM6HIP_REMAP_PPM_H4 = 4, MOM6HIP_REMAP_PPM_IH4 = 5, &
                                 MOM6HIP_REMAP_PPM_CW = 10, MOM6HIP_REMAP_PLM_HYBGEN = 3, MOM6HIP_REMAP_PPM_HYBGEN = 6, &
                                 MOM6HIP_REMAP_WENO_HYBGEN = 7, MOM6HIP_REMAP_PQM_IH4IH3 = 8, MOM6HIP_REMAP_PQM_IH6IH5 = 9, &
                                 MOM6HIP_REGRIDDING_ZSTAR = 2

!> mom6hip_grid_t of include/mom6hip.h
type, bind(c) :: mom6hip_grid_t
  integer(c_int32_t) :: isc, iec, jsc, jec
  integer(c_int32_t) :: isd, ied, jsd, jed
  integer(c_int32_t) :: nk
  integer(c_int32_t) :: symmetric
  integer(c_int32_t) :: reentrant_x, reentrant_y
  integer(c_int32_t) :: first_direction
  integer(c_int32_t) :: tripolar_n   !< TRIPOLAR_N: the northern fold (one tile in x, REENTRANT_X)
  real(c_double) :: Angstrom_H, H_subroundoff, dZ_subroundoff, H_to_Z, Z_to_H, g_Earth, Rho0
  real(c_double) :: reserved1(8)
  type(c_ptr) :: mask2dT, areaT, IareaT, dxT, dyT, IdxT, IdyT, bathyT
  type(c_ptr) :: mask2dCu, dxCu, dyCu, dy_Cu, IdxCu, IdyCu, areaCu, IareaCu
  type(c_ptr) :: mask2dCv, dxCv, dyCv, dx_Cv, IdxCv, IdyCv, areaCv, IareaCv
  type(c_ptr) :: mask2dBu, dxBu, dyBu, areaBu, IareaBu, CoriolisBu, IdxBu, IdyBu
  type(c_ptr) :: reserved2(6)
end type mom6hip_grid_t

!> mom6hip_tracer_advect_cs_t
type, bind(c) :: mom6hip_tracer_advect_cs_t
  real(c_double) :: dt
  integer(c_int32_t) :: scheme
  integer(c_int32_t) :: use_huynh_stencil_bug
end type mom6hip_tracer_advect_cs_t

!> mom6hip_advect_stats_t
type, bind(c) :: mom6hip_advect_stats_t
  integer(c_int32_t) :: iterations, halo_updates, domore_remaining, reserved
end type mom6hip_advect_stats_t

!> mom6hip_remapping_cs_t (remapping_CS, src/ALE/MOM_remapping.F90:25)
type, bind(c) :: mom6hip_remapping_cs_t
  integer(c_int32_t) :: remapping_scheme, boundary_extrapolation, force_bounds_in_subcell, answer_date
end type mom6hip_remapping_cs_t

!> mom6hip_regridding_cs_t (regridding_CS, src/ALE/MOM_regridding.F90:50; z* only)
type, bind(c) :: mom6hip_regridding_cs_t
  integer(c_int32_t) :: regridding_scheme, nk
  real(c_double) :: min_thickness, old_grid_weight, depth_of_time_filter_shallow, depth_of_time_filter_deep, Z_ref
  type(c_ptr) :: coordinateResolution
end type mom6hip_regridding_cs_t

!> mom6hip_energy_sums_t: the global integrals of write_energy (src/diagnostics/MOM_sum_output.F90:490-760)
type, bind(c) :: mom6hip_energy_sums_t
  real(c_double) :: mass_tot, KE_tot, PE_tot, toten, Salt, Heat
  real(c_double) :: max_CFL(2)
  integer(c_int64_t) :: mass_EFP(6), salt_EFP(6), heat_EFP(6)
  integer(c_int64_t) :: npoints
end type mom6hip_energy_sums_t

!> mom6hip_coriolisadv_cs_t (CoriolisAdv_CS, src/core/MOM_CoriolisAdv.F90:30)
type, bind(c) :: mom6hip_coriolisadv_cs_t
  integer(c_int32_t) :: coriolis_scheme, ke_scheme, no_slip, bound_coriolis, coriolis_en_dis, pv_adv_scheme
  integer(c_int32_t) :: reserved(2)
  real(c_double) :: F_eff_max_blend, wt_lin_blend
end type mom6hip_coriolisadv_cs_t

!> mom6hip_continuity_cs_t (continuity_PPM_CS, src/core/MOM_continuity_PPM.F90:35)
type, bind(c) :: mom6hip_continuity_cs_t
  integer(c_int32_t) :: upwind_1st, monotonic, simple_2nd, aggress_adjust, vol_CFL, better_iter, use_visc_rem_max, &
                        marginal_faces
  real(c_double) :: tol_eta, tol_vel, CFL_limit_adjust
end type mom6hip_continuity_cs_t

!> mom6hip_bt_cont_t (BT_cont_type, src/core/MOM_variables.F90): c_loc of each member array
type, bind(c) :: mom6hip_bt_cont_t
  type(c_ptr) :: FA_u_W0, FA_u_WW, FA_u_E0, FA_u_EE, uBT_WW, uBT_EE
  type(c_ptr) :: FA_v_S0, FA_v_SS, FA_v_N0, FA_v_NN, vBT_SS, vBT_NN
  type(c_ptr) :: h_u, h_v
end type mom6hip_bt_cont_t

!> mom6hip_eos_t (EOS_type members of the provided forms, src/equation_of_state/MOM_EOS.F90)
type, bind(c) :: mom6hip_eos_t
  integer(c_int32_t) :: form, reserved
  real(c_double) :: Rho_T0_S0, dRho_dT, dRho_dS
end type mom6hip_eos_t

!> mom6hip_pressureforce_cs_t (PressureForce_FV_CS, src/core/MOM_PressureForce_FV.F90)
type, bind(c) :: mom6hip_pressureforce_cs_t
  real(c_double) :: Rho0, GFS_scale, Z_ref
  integer(c_int32_t) :: reconstruct, Recon_Scheme, boundary_extrap, useMassWghtInterp
  integer(c_int32_t) :: use_ALE = 1   !< associated(ALE_CSp)
  integer(c_int32_t) :: nkmb = 0      !< GV%nk_rho_varies
  real(c_double) :: P_Ref = 2.0d7     !< tv%P_Ref
  type(c_ptr) :: Rlay = c_null_ptr    !< GV%Rlay(1:nk): with nkmb > 0 or without an equation of state
  type(c_ptr) :: g_prime = c_null_ptr !< GV%g_prime(1:nk+1): without an equation of state
end type mom6hip_pressureforce_cs_t

!> mom6hip_barotropic_cs_t (barotropic_CS, src/core/MOM_barotropic.F90:104)
type, bind(c) :: mom6hip_barotropic_cs_t
  real(c_double) :: dtbt, dtbt_max, dtbt_fraction, bebt, dt_bt_filter, vel_underflow, G_extra, BT_Coriolis_scale, Z_ref
  real(c_double) :: maxCFL_BT_cont   !< MAXCFL_BT_CONT, with bound_BT_corr
  real(c_double) :: reserved0(6)
  integer(c_int32_t) :: Sadourny, linearized_BT_PV, strong_drag, visc_rem_u_uh0, adjust_BT_cont, use_wide_halos, &
                        hvel_scheme, nstep_last
  integer(c_int32_t) :: unsupported(12)
  integer(c_int32_t) :: bound_BT_corr   !< BOUND_BT_CORRECTION with BT_CONT_CORR_BOUNDS and a BT_cont argument
  integer(c_int32_t) :: BT_project_velocity   !< BT_PROJECT_VELOCITY
  integer(c_int32_t) :: Nonlinear_continuity = 0        !< NONLINEAR_BT_CONTINUITY
  integer(c_int32_t) :: Nonlin_cont_update_period = 1   !< NONLIN_BT_CONT_UPDATE_PERIOD
  type(c_ptr) :: frhatu, frhatv, eta_cor, IDatu, IDatv, ubtav, vbtav, q_D, D_u_Cor, D_v_Cor
  type(c_ptr) :: reserved2(6)
end type mom6hip_barotropic_cs_t

!> mom6hip_visc_hooks_t: the host's viscosity parameterisations, called at the reference's seams with device pointers
type, bind(c) :: mom6hip_visc_hooks_t
  type(c_ptr) :: user
  type(c_funptr) :: visc_remnant_pred, vertvisc, horizontal_viscosity
end type mom6hip_visc_hooks_t

!> mom6hip_vertvisc_cs_t (vertvisc_CS, src/parameterizations/vertical/MOM_vert_friction.F90:40)
type, bind(c) :: mom6hip_vertvisc_cs_t
  real(c_double) :: Hmix, Hmix_stress, Kvml_invZ2, Kv, Hbbl, Kv_extra_bbl, harm_BL_val, maxvel, CFL_trunc, vel_underflow, H_to_RZ
  real(c_double) :: vonKar = 0.41                 !< VON_KARMAN_CONST
  integer(c_int32_t) :: dynamic_viscous_ML = 0    !< DYNAMIC_VISCOUS_ML
  integer(c_int32_t) :: nkml = 0                  !< GV%nkml
  real(c_double) :: reserved0(3)
  integer(c_int32_t) :: bottomdraglaw, harmonic_visc, direct_stress, CFL_based_trunc, answer_date
  integer(c_int32_t) :: unsupported(7)
  integer(c_int64_t) :: ntrunc
  type(c_ptr) :: a_u, a_v, h_u, h_v
  type(c_ptr) :: reserved1(4)
end type mom6hip_vertvisc_cs_t

!> mom6hip_vertvisc_type_t (the members of vertvisc_type, src/core/MOM_variables.F90:218, that are read)
type, bind(c) :: mom6hip_vertvisc_type_t
  type(c_ptr) :: Kv_bbl_u, Kv_bbl_v, bbl_thick_u, bbl_thick_v, Ray_u, Ray_v, Kv_shear, Kv_shear_Bu
  type(c_ptr) :: nkml_visc_u = c_null_ptr, nkml_visc_v = c_null_ptr   !< visc%nkml_visc_u / _v (DYNAMIC_VISCOUS_ML)
  type(c_ptr) :: ustar = c_null_ptr                                   !< forces%ustar (find_ustar)
  type(c_ptr) :: reserved(1)
end type mom6hip_vertvisc_type_t

!> mom6hip_set_visc_cs_t (set_visc_CS, src/parameterizations/vertical/MOM_set_viscosity.F90:48)
type, bind(c) :: mom6hip_set_visc_cs_t
  real(c_double) :: cdrag, drag_bg_vel, Hbbl, dz_bbl, BBL_thick_min, Kv_BBL_min, BBL_thick_max, H_to_RZ
  real(c_double) :: omega = 7.2921d-5, omega_frac = 0.0d0, ustar_min = 0.0d0, TKE_decay = 0.0d0, bulk_Ri_ML = 0.0d0
  real(c_double) :: c_Smag = 0.15d0, Chan_drag_max_vol = -1.0d0
  real(c_double) :: Z_ref = 0.0d0      !< G%Z_ref
  integer(c_int32_t) :: bottomdraglaw, linear_drag, BBL_use_EOS, correct_BBL_bounds, body_force_drag, RiNo_mix, initialized
  integer(c_int32_t) :: unsupported(9)
  type(c_ptr) :: Rlay           !< c_loc of GV%Rlay (host), read without BBL_USE_EOS
  integer(c_int32_t) :: dynamic_viscous_ML = 0, nkml = 0, Channel_drag = 0, concave_trigonometric_L = 1
  type(c_ptr) :: reserved1(1)
end type mom6hip_set_visc_cs_t

!> mom6hip_tracer_hor_diff_cs_t (tracer_hor_diff_CS, src/tracer/MOM_tracer_hor_diff.F90:40)
type, bind(c) :: mom6hip_tracer_hor_diff_cs_t
  real(c_double) :: KhTr, max_diff_CFL
  real(c_double) :: KhTr_Slope_Cff = 0.0, KhTr_fac = 1.0, KhTr_min = 0.0, KhTr_max = 0.0, KhTr_passivity_coeff = 0.0, KhTr_passivity_min = 0.5
  integer(c_int32_t) :: check_diffusive_CFL, initialized
  integer(c_int32_t) :: unsupported(8)
  integer(c_int32_t) :: use_variable_mixing = 0, Resoln_scaled_KhTr = 0
  integer(c_int32_t) :: reserved1(4)
end type mom6hip_tracer_hor_diff_cs_t

!> mom6hip_hordiff_fields_t: the fields of MEKE and VarMix tracer_hordiff reads with variable mixing
type, bind(c) :: mom6hip_hordiff_fields_t
  type(c_ptr) :: MEKE_Kh = c_null_ptr, L2u = c_null_ptr, L2v = c_null_ptr, SN_u = c_null_ptr, SN_v = c_null_ptr
  type(c_ptr) :: Res_fn_h = c_null_ptr, Rd_dx_h = c_null_ptr
  type(c_ptr) :: h_ML = c_null_ptr      !< visc%h_ML (NDIFF_INTERIOR_ONLY)
  type(c_ptr) :: reserved(4) = c_null_ptr
end type mom6hip_hordiff_fields_t

!> mom6hip_neutral_diffusion_cs_t (neutral_diffusion_CS, src/tracer/MOM_neutral_diffusion.F90:38), the continuous branch
type, bind(c) :: mom6hip_neutral_diffusion_cs_t
  real(c_double) :: ref_pres = -1.0, H_to_RZ = 0.0
  real(c_double) :: reserved0(4) = 0.0
  integer(c_int32_t) :: ndiff_answer_date = 20240101, recalc_neutral_surf = 0, initialized = 0
  integer(c_int32_t) :: interior_only = 0
  integer(c_int32_t) :: unsupported(8) = 0
end type mom6hip_neutral_diffusion_cs_t

!> mom6hip_obc_segment_t / mom6hip_obc_t: what continuity_PPM reads of OBC_segment_type / ocean_OBC_type (src/core/MOM_open_boundary.F90:146, :266)
integer(c_int32_t), parameter :: MOM6HIP_OBC_TAN_RADIATION = 1, MOM6HIP_OBC_GRAD_RADIATION = 2, MOM6HIP_OBC_TAN_NUDGED = 4, &
                                 MOM6HIP_OBC_GRAD_NUDGED = 8, MOM6HIP_OBC_TAN_OBLIQUE = 16, MOM6HIP_OBC_GRAD_OBLIQUE = 32
integer(c_int32_t), parameter :: MOM6HIP_OBC_NONE = 0, MOM6HIP_OBC_DIRECTION_N = 100, MOM6HIP_OBC_DIRECTION_S = 200, &
                                 MOM6HIP_OBC_DIRECTION_E = 300, MOM6HIP_OBC_DIRECTION_W = 400
!> mom6hip_obc_segment_tracer_t: one registered tracer of a segment (segment%tr_Reg%Tr(m))
type, bind(c) :: mom6hip_obc_segment_tracer_t
  integer(c_int32_t) :: ntr_index = 0, reserved = 0
  type(c_ptr) :: tres = c_null_ptr
  real(c_double) :: OBC_inflow_conc = 0.0
  type(c_ptr) :: t = c_null_ptr
  real(c_double) :: resrv_lfac_in = 1.0, resrv_lfac_out = 1.0
end type mom6hip_obc_segment_tracer_t
type, bind(c) :: mom6hip_obc_segment_t
  integer(c_int32_t) :: direction = 0, open = 0, specified = 0, on_pe = 0, is_E_or_W = 0, is_N_or_S = 0
  integer(c_int32_t) :: IsdB = 0, IedB = 0, JsdB = 0, JedB = 0, isd = 0, ied = 0, jsd = 0, jed = 0
  integer(c_int32_t) :: radiation = 0, gradient = 0, nudged = 0, oblique = 0, radiation_tan_or_grad = 0
  integer(c_int32_t) :: Flather = 0
  type(c_ptr) :: normal_trans = c_null_ptr, normal_vel = c_null_ptr, tangential_vel = c_null_ptr, tangential_grad = c_null_ptr
  type(c_ptr) :: nudged_normal_vel = c_null_ptr, normal_vel_bt = c_null_ptr, SSH = c_null_ptr
  real(c_double) :: Velocity_nudging_timescale_in = 0.0, Velocity_nudging_timescale_out = 0.0
  type(c_ptr) :: tr_Reg = c_null_ptr
  integer(c_int32_t) :: ntseg = 0, reserved_i = 0
  real(c_double) :: Tr_InvLscale_in = 0.0, Tr_InvLscale_out = 0.0
  type(c_ptr) :: nudged_tangential_vel = c_null_ptr, nudged_tangential_grad = c_null_ptr
end type mom6hip_obc_segment_t
type, bind(c) :: mom6hip_obc_t
  integer(c_int32_t) :: number_of_segments = 0, OBC_pe = 0, open_u_BCs_exist_globally = 0, open_v_BCs_exist_globally = 0
  integer(c_int32_t) :: specified_u_BCs_exist_globally = 0, specified_v_BCs_exist_globally = 0
  integer(c_int32_t) :: Flather_u_BCs_exist_globally = 0, Flather_v_BCs_exist_globally = 0
  integer(c_int32_t) :: zero_vorticity = 0, freeslip_vorticity = 0, computed_vorticity = 0, specified_vorticity = 0
  integer(c_int32_t) :: zero_strain = 0, freeslip_strain = 0, computed_strain = 0, zero_biharmonic = 0
  type(c_ptr) :: segment = c_null_ptr, segnum_u = c_null_ptr, segnum_v = c_null_ptr
  type(c_ptr) :: rx_normal = c_null_ptr, ry_normal = c_null_ptr
  real(c_double) :: gamma_uv = 0.0, rx_max = 0.0
  type(c_ptr) :: rx_oblique_u = c_null_ptr, ry_oblique_u = c_null_ptr, cff_normal_u = c_null_ptr
  type(c_ptr) :: rx_oblique_v = c_null_ptr, ry_oblique_v = c_null_ptr, cff_normal_v = c_null_ptr
end type mom6hip_obc_t

!> mom6hip_epipycnal_cs_t (DIFFUSE_ML_TO_INTERIOR: tracer_epipycnal_ML_diff, src/tracer/MOM_tracer_hor_diff.F90:700)
type, bind(c) :: mom6hip_epipycnal_cs_t
  real(c_double) :: ML_KhTr_scale = 1.0, P_Ref = 0.0
  real(c_double) :: reserved0(4) = 0.0
  type(c_ptr) :: Rlay = c_null_ptr
  integer(c_int32_t) :: nkml = 0, nk_rho_varies = 0, answer_date = 20240101, limit_bug = 1
  integer(c_int32_t) :: reserved1(4) = 0
end type mom6hip_epipycnal_cs_t

!> mom6hip_hordiff_stats_t
type, bind(c) :: mom6hip_hordiff_stats_t
  integer(c_int32_t) :: num_itts, halo_updates
  real(c_double) :: max_CFL
end type mom6hip_hordiff_stats_t

!> mom6hip_hor_visc_cs_t (hor_visc_CS, src/parameterizations/lateral/MOM_hor_visc.F90:40)
type, bind(c) :: mom6hip_hor_visc_cs_t
  real(c_double) :: Kh, Kh_bg_min, Kh_vel_scale, Smag_Lap_const, Ah, Ah_vel_scale, Ah_time_scale, Smag_bi_const, bound_Cor_vel, bound_coef
  real(c_double) :: reserved0(6)
  integer(c_int32_t) :: Laplacian, biharmonic, Smagorinsky_Kh, Smagorinsky_Ah, bound_Kh, better_bound_Kh, bound_Ah, better_bound_Ah, &
                        bound_Coriolis, add_LES_viscosity, no_slip, use_land_mask, use_cont_thick, initialized
  integer(c_int32_t) :: unsupported(10)
  type(c_ptr) :: Kh_bg_xx, Kh_Max_xx, Ah_bg_xx, Ah_Max_xx, Laplac2_const_xx, Biharm_const_xx, Biharm_const2_xx, reduction_xx
  type(c_ptr) :: Kh_bg_xy, Kh_Max_xy, Ah_bg_xy, Ah_Max_xy, Laplac2_const_xy, Biharm_const_xy, Biharm_const2_xy, reduction_xy
  type(c_ptr) :: MEKE_Ku = c_null_ptr, MEKE_Au = c_null_ptr, MEKE_mom_src = c_null_ptr   !< MEKE%Ku, %Au (in), %mom_src (out) or null
  type(c_ptr) :: reserved1(1)
end type mom6hip_hor_visc_cs_t

!> mom6hip_thickness_diffuse_cs_t (thickness_diffuse_CS, src/parameterizations/lateral/MOM_thickness_diffuse.F90:40)
type, bind(c) :: mom6hip_thickness_diffuse_cs_t
  real(c_double) :: Khth = 0.0, Khth_Min = 0.0, Khth_Max = 0.0, max_Khth_CFL = 0.8, slope_max = 0.01, kappa_smooth = 1.0e-6
  real(c_double) :: KHTH_Slope_Cff = 0.0, KhTh_fac = 1.0, FGNV_scale = 1.0, N2_floor = 0.0
  real(c_double) :: reserved0(2) = 0.0
  integer(c_int32_t) :: thickness_diffuse = 0, use_GM_work_bug = 0, nkml = 0, initialized = 0, use_variable_mixing = 0
  integer(c_int32_t) :: use_FGNV_streamfn = 0
  integer(c_int32_t) :: unsupported(10) = 0
  type(c_ptr) :: MEKE_Kh = c_null_ptr, L2u = c_null_ptr, L2v = c_null_ptr, SN_u = c_null_ptr, SN_v = c_null_ptr
  type(c_ptr) :: Res_fn_u = c_null_ptr, Res_fn_v = c_null_ptr, slope_x = c_null_ptr, slope_y = c_null_ptr
  type(c_ptr) :: MEKE_GM_src = c_null_ptr, Rlay = c_null_ptr, cg1 = c_null_ptr, g_prime = c_null_ptr
  type(c_ptr) :: Depth_fn_u = c_null_ptr, Depth_fn_v = c_null_ptr
end type mom6hip_thickness_diffuse_cs_t

!> mom6hip_mixedlayer_restrat_cs_t (mixedlayer_restrat_CS, src/parameterizations/lateral/MOM_mixed_layer_restrat.F90:40)
type, bind(c) :: mom6hip_mixedlayer_restrat_cs_t
  real(c_double) :: ml_restrat_coef = 0.0, ml_restrat_coef2 = 0.0, front_length = 0.0, vonKar = 0.41
  real(c_double) :: MLE_MLD_decay_time = 0.0, MLE_MLD_decay_time2 = 0.0, MLE_density_diff = 0.03, MLE_tail_dh = 0.0
  real(c_double) :: MLE_MLD_stretch = 1.0, ustar_min = 0.0
  real(c_double) :: reserved0(4) = 0.0
  integer(c_int32_t) :: MLE_use_PBL_MLD = 0, nkml = 0, initialized = 0
  integer(c_int32_t) :: reserved_i(1) = 0
  integer(c_int32_t) :: unsupported(8) = 0
  type(c_ptr) :: MLD_filtered = c_null_ptr, MLD_filtered_slow = c_null_ptr, Rd_dx_h = c_null_ptr
  type(c_ptr) :: reserved1(3) = c_null_ptr
end type mom6hip_mixedlayer_restrat_cs_t

!> mom6hip_dyn_split_rk2_cs_t (MOM_dyn_split_RK2_CS, src/core/MOM_dynamics_split_RK2.F90:84); every array is a DEVICE
!! array obtained from mom6hip_malloc
type, bind(c) :: mom6hip_dyn_split_rk2_cs_t
  real(c_double) :: be, begw
  integer(c_int32_t) :: BT_use_layer_fluxes, store_CAu, CAu_pred_stored, split_bottom_stress
  integer(c_int32_t) :: reserved0(4)
  type(c_ptr) :: continuity_CSp, CoriolisAdv, PressureForce_CSp, eqn_of_state, barotropic_CSp, BT_cont, hooks
  type(c_ptr) :: vertvisc_CSp   !< c_loc of a mom6hip_vertvisc_cs_t, or c_null_ptr
  type(c_ptr) :: visc           !< c_loc of a mom6hip_vertvisc_type_t (the visc argument of the step)
  type(c_ptr) :: hor_visc       !< c_loc of a mom6hip_hor_visc_cs_t, or c_null_ptr
  type(c_ptr) :: CAu, CAv, CAu_pred, CAv_pred, PFu, PFv, diffu, diffv, visc_rem_u, visc_rem_v, u_accel_bt, v_accel_bt, &
                 u_av, v_av, h_av, pbce
  type(c_ptr) :: eta, eta_PF, uhbt, vhbt
  type(c_ptr) :: du_av_inst, dv_av_inst   !< SPLIT_RK2B only (MOM_dynamics_split_RK2b.F90:141-146)
  type(c_ptr) :: set_visc_CSp = c_null_ptr   !< c_loc of a mom6hip_set_visc_cs_t with dynamic_viscous_ML, or c_null_ptr
  type(c_ptr) :: OBC = c_null_ptr      !< CS%OBC: c_loc of a mom6hip_obc_t of DEVICE arrays, or c_null_ptr
  !> The surface pressures of the call in progress (DEVICE h-point arrays or c_null_ptr): the step's arguments p_surf_begin,
  !! p_surf_end and forces%p_surf (MOM_dynamics_split_RK2.F90:435-442, :495-503)
  type(c_ptr) :: p_surf_begin = c_null_ptr, p_surf_end = c_null_ptr, p_surf = c_null_ptr
end type mom6hip_dyn_split_rk2_cs_t

interface
  function mom6hip_init(device) bind(c, name="mom6hip_init") result(rc)
    import :: c_int
    integer(c_int), value :: device
    integer(c_int) :: rc
  end function mom6hip_init

  function mom6hip_last_error() bind(c, name="mom6hip_last_error") result(msg)
    import :: c_ptr
    type(c_ptr) :: msg
  end function mom6hip_last_error

  function mom6hip_grid_create(grid, stream, ctx) bind(c, name="mom6hip_grid_create") result(rc)
    import :: c_int, c_ptr, mom6hip_grid_t
    type(mom6hip_grid_t), intent(in) :: grid
    type(c_ptr), value :: stream
    type(c_ptr), intent(out) :: ctx
    integer(c_int) :: rc
  end function mom6hip_grid_create

  function mom6hip_grid_destroy(ctx) bind(c, name="mom6hip_grid_destroy") result(rc)
    import :: c_int, c_ptr
    type(c_ptr), value :: ctx
    integer(c_int) :: rc
  end function mom6hip_grid_destroy

  function mom6hip_sync(ctx) bind(c, name="mom6hip_sync") result(rc)
    import :: c_int, c_ptr
    type(c_ptr), value :: ctx
    integer(c_int) :: rc
  end function mom6hip_sync

  function mom6hip_advect_tracer(ctx, h_end, uhtr, vhtr, dt, cs, tr, conc_underflow, ntr, x_first_in, &
                                 vol_prev, max_iter_in, update_vol_prev, uhr_out, vhr_out, memspace, stats) &
                                 bind(c, name="mom6hip_advect_tracer") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_tracer_advect_cs_t, mom6hip_advect_stats_t
    type(c_ptr), value :: ctx
    type(c_ptr), value :: h_end, uhtr, vhtr
    real(c_double), value :: dt
    type(mom6hip_tracer_advect_cs_t), intent(in) :: cs
    type(c_ptr), intent(in) :: tr(*)          !< c_loc of each Reg%Tr(m)%t
    type(c_ptr), value :: conc_underflow      !< c_loc of a real(c_double) array, or c_null_ptr
    integer(c_int32_t), value :: ntr, x_first_in
    type(c_ptr), value :: vol_prev
    integer(c_int32_t), value :: max_iter_in, update_vol_prev
    type(c_ptr), value :: uhr_out, vhr_out
    integer(c_int32_t), value :: memspace
    type(mom6hip_advect_stats_t), intent(out) :: stats
    integer(c_int) :: rc
  end function mom6hip_advect_tracer

  function mom6hip_advect_tracer_obc(ctx, h_end, uhtr, vhtr, dt, cs, tr, conc_underflow, ntr, x_first_in, &
                                     vol_prev, max_iter_in, update_vol_prev, uhr_out, vhr_out, obc, memspace, stats) &
                                     bind(c, name="mom6hip_advect_tracer_obc") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_tracer_advect_cs_t, mom6hip_advect_stats_t
    type(c_ptr), value :: ctx
    type(c_ptr), value :: h_end, uhtr, vhtr
    real(c_double), value :: dt
    type(mom6hip_tracer_advect_cs_t), intent(in) :: cs
    type(c_ptr), intent(in) :: tr(*)          !< c_loc of each Reg%Tr(m)%t
    type(c_ptr), value :: conc_underflow      !< c_loc of a real(c_double) array, or c_null_ptr
    integer(c_int32_t), value :: ntr, x_first_in
    type(c_ptr), value :: vol_prev
    integer(c_int32_t), value :: max_iter_in, update_vol_prev
    type(c_ptr), value :: uhr_out, vhr_out
    type(c_ptr), value :: obc                 !< c_loc of a mom6hip_obc_t (with the tracer registries of its segments), or c_null_ptr
    integer(c_int32_t), value :: memspace
    type(mom6hip_advect_stats_t), intent(out) :: stats
    integer(c_int) :: rc
  end function mom6hip_advect_tracer_obc

  function mom6hip_update_segment_tracer_reservoirs(ctx, uhr, vhr, h, obc, dt, tr, ntr, memspace) &
                                                    bind(c, name="mom6hip_update_segment_tracer_reservoirs") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_obc_t
    type(c_ptr), value :: ctx, uhr, vhr, h
    type(mom6hip_obc_t), intent(in) :: obc
    real(c_double), value :: dt
    type(c_ptr), intent(in) :: tr(*)          !< c_loc of each Reg%Tr(m)%t
    integer(c_int32_t), value :: ntr, memspace
    integer(c_int) :: rc
  end function mom6hip_update_segment_tracer_reservoirs

  function mom6hip_malloc(dptr, bytes) bind(c, name="mom6hip_malloc") result(rc)
    import :: c_int, c_ptr, c_int64_t
    type(c_ptr), intent(out) :: dptr
    integer(c_int64_t), value :: bytes
    integer(c_int) :: rc
  end function mom6hip_malloc

  function mom6hip_free(dptr) bind(c, name="mom6hip_free") result(rc)
    import :: c_int, c_ptr
    type(c_ptr), value :: dptr
    integer(c_int) :: rc
  end function mom6hip_free

  function mom6hip_memset_zero(ctx, dptr, bytes) bind(c, name="mom6hip_memset_zero") result(rc)
    import :: c_int, c_ptr, c_int64_t
    type(c_ptr), value :: ctx, dptr
    integer(c_int64_t), value :: bytes
    integer(c_int) :: rc
  end function mom6hip_memset_zero

  function mom6hip_start_group_pass(ctx, fields, pos, nk_each, nfields) bind(c, name="mom6hip_start_group_pass") result(rc)
    import :: c_int, c_ptr, c_int32_t
    type(c_ptr), value :: ctx
    type(c_ptr), intent(in) :: fields(*)
    integer(c_int32_t), intent(in) :: pos(*), nk_each(*)
    integer(c_int32_t), value :: nfields
    integer(c_int) :: rc
  end function mom6hip_start_group_pass

  function mom6hip_complete_group_pass(ctx) bind(c, name="mom6hip_complete_group_pass") result(rc)
    import :: c_int, c_ptr
    type(c_ptr), value :: ctx
    integer(c_int) :: rc
  end function mom6hip_complete_group_pass

  function mom6hip_debug_poison_passes(ctx, enable) bind(c, name="mom6hip_debug_poison_passes") result(rc)
    import :: c_int, c_ptr, c_int32_t
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: enable
    integer(c_int) :: rc
  end function mom6hip_debug_poison_passes

  function mom6hip_overlap_stats(ctx, stats, reset) bind(c, name="mom6hip_overlap_stats") result(rc)
    import :: c_int, c_ptr, c_int64_t, c_int32_t
    type(c_ptr), value :: ctx
    integer(c_int64_t), intent(out) :: stats(4)
    integer(c_int32_t), value :: reset
    integer(c_int) :: rc
  end function mom6hip_overlap_stats

  function mom6hip_transfer_stats(ctx, stats, reset) bind(c, name="mom6hip_transfer_stats") result(rc)
    import :: c_int, c_ptr, c_int64_t, c_int32_t
    type(c_ptr), value :: ctx
    integer(c_int64_t), intent(out) :: stats(4)
    integer(c_int32_t), value :: reset
    integer(c_int) :: rc
  end function mom6hip_transfer_stats

  function mom6hip_sync_to_device(ctx, dptr, hptr, bytes) bind(c, name="mom6hip_sync_to_device") result(rc)
    import :: c_int, c_ptr, c_int64_t
    type(c_ptr), value :: ctx, dptr, hptr
    integer(c_int64_t), value :: bytes
    integer(c_int) :: rc
  end function mom6hip_sync_to_device

  function mom6hip_sync_to_host(ctx, hptr, dptr, bytes) bind(c, name="mom6hip_sync_to_host") result(rc)
    import :: c_int, c_ptr, c_int64_t
    type(c_ptr), value :: ctx, hptr, dptr
    integer(c_int64_t), value :: bytes
    integer(c_int) :: rc
  end function mom6hip_sync_to_host

  function mom6hip_halo_update(ctx, fields, pos, nk_each, nfields) bind(c, name="mom6hip_halo_update") result(rc)
    import :: c_int, c_int32_t, c_ptr
    type(c_ptr), value :: ctx
    type(c_ptr), intent(in) :: fields(*)
    integer(c_int32_t), intent(in) :: pos(*), nk_each(*)
    integer(c_int32_t), value :: nfields
    integer(c_int) :: rc
  end function mom6hip_halo_update

  !> halo_fn / sum_fn: bind(c) procedures of the host wrapping do_group_pass / sum_across_PEs on device buffers
  function mom6hip_set_domain_callbacks(ctx, halo_fn, sum_fn, user) bind(c, name="mom6hip_set_domain_callbacks") result(rc)
    import :: c_int, c_ptr, c_funptr
    type(c_ptr), value :: ctx, user
    type(c_funptr), value :: halo_fn, sum_fn
    integer(c_int) :: rc
  end function mom6hip_set_domain_callbacks

  function mom6hip_set_min_callback(ctx, min_fn, user) bind(c, name="mom6hip_set_min_callback") result(rc)
    import :: c_int, c_ptr, c_funptr
    type(c_ptr), value :: ctx, user
    type(c_funptr), value :: min_fn
    integer(c_int) :: rc
  end function mom6hip_set_min_callback

  function mom6hip_ale_remap_tracers(ctx, cs, h_old, h_new, tr, conc_underflow, ntr, memspace) &
                                     bind(c, name="mom6hip_ale_remap_tracers") result(rc)
    import :: c_int, c_int32_t, c_ptr, mom6hip_remapping_cs_t
    type(c_ptr), value :: ctx, h_old, h_new, conc_underflow
    type(mom6hip_remapping_cs_t), intent(in) :: cs
    type(c_ptr), intent(in) :: tr(*)
    integer(c_int32_t), value :: ntr, memspace
    integer(c_int) :: rc
  end function mom6hip_ale_remap_tracers

  function mom6hip_ale_regrid(ctx, cs, h, h_new, dzRegrid, memspace) bind(c, name="mom6hip_ale_regrid") result(rc)
    import :: c_int, c_int32_t, c_ptr, mom6hip_regridding_cs_t
    type(c_ptr), value :: ctx, h, h_new, dzRegrid
    type(mom6hip_regridding_cs_t), intent(in) :: cs
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_ale_regrid

  function mom6hip_ale_remap_set_h_vel(ctx, h_new, h_u, h_v, memspace) bind(c, name="mom6hip_ale_remap_set_h_vel") result(rc)
    import :: c_int, c_int32_t, c_ptr
    type(c_ptr), value :: ctx, h_new, h_u, h_v
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_ale_remap_set_h_vel

  function mom6hip_ale_remap_set_h_vel_via_dz(ctx, h_old, dzInterface, h_u, h_v, memspace) &
                                              bind(c, name="mom6hip_ale_remap_set_h_vel_via_dz") result(rc)
    import :: c_int, c_int32_t, c_ptr
    type(c_ptr), value :: ctx, h_old, dzInterface, h_u, h_v
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_ale_remap_set_h_vel_via_dz

  !> ALE_PLM_edge_values (MOM_ALE.F90:1520)
  function mom6hip_ale_plm_edge_values(ctx, h, Q, bdry_extrap, Q_t, Q_b, memspace) bind(c, name="mom6hip_ale_plm_edge_values") result(rc)
    import :: c_int, c_int32_t, c_ptr
    type(c_ptr), value :: ctx, h, Q, Q_t, Q_b
    integer(c_int32_t), value :: bdry_extrap, memspace
    integer(c_int) :: rc
  end function mom6hip_ale_plm_edge_values

  function mom6hip_ale_remap_velocities(ctx, cs, h_old_u, h_old_v, h_new_u, h_new_v, u, v, memspace) &
                                        bind(c, name="mom6hip_ale_remap_velocities") result(rc)
    import :: c_int, c_int32_t, c_ptr, mom6hip_remapping_cs_t
    type(c_ptr), value :: ctx, h_old_u, h_old_v, h_new_u, h_new_v, u, v
    type(mom6hip_remapping_cs_t), intent(in) :: cs
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_ale_remap_velocities

  !> tracer_hordiff (MOM_tracer_hor_diff.F90:119), along-layer diffusion with a constant KHTR; tr: array of ntr c_loc's
  function mom6hip_tracer_hordiff(ctx, cs, h, dt, tr, conc_underflow, ntr, memspace, stats) bind(c, name="mom6hip_tracer_hordiff") &
                                  result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_tracer_hor_diff_cs_t, mom6hip_hordiff_stats_t
    type(c_ptr), value :: ctx, h, conc_underflow
    type(mom6hip_tracer_hor_diff_cs_t), intent(in) :: cs
    real(c_double), value :: dt
    type(c_ptr), intent(in) :: tr(*)
    integer(c_int32_t), value :: ntr, memspace
    type(mom6hip_hordiff_stats_t), intent(out) :: stats
    integer(c_int) :: rc
  end function mom6hip_tracer_hordiff

  function mom6hip_tracer_hordiff_varmix(ctx, cs, fields, h, dt, tr, conc_underflow, ntr, memspace, stats) &
                                         bind(c, name="mom6hip_tracer_hordiff_varmix") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_tracer_hor_diff_cs_t, mom6hip_hordiff_stats_t, mom6hip_hordiff_fields_t
    type(c_ptr), value :: ctx, h, conc_underflow
    type(mom6hip_tracer_hor_diff_cs_t), intent(in) :: cs
    type(mom6hip_hordiff_fields_t), intent(in) :: fields
    real(c_double), value :: dt
    type(c_ptr), intent(in) :: tr(*)
    integer(c_int32_t), value :: ntr, memspace
    type(mom6hip_hordiff_stats_t), intent(out) :: stats
    integer(c_int) :: rc
  end function mom6hip_tracer_hordiff_varmix

  !> tracer_hordiff with CS%use_neutral_diffusion (cs%unsupported(1)); idx_T, idx_S: the 0-based places of tv%T, tv%S in tr
  function mom6hip_tracer_hordiff_neutral(ctx, cs, nd, fields, h, eos, p_surf, dt, tr, conc_underflow, ntr, idx_T, idx_S, memspace, &
                                          stats) bind(c, name="mom6hip_tracer_hordiff_neutral") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_tracer_hor_diff_cs_t, mom6hip_hordiff_stats_t, mom6hip_hordiff_fields_t, &
              mom6hip_neutral_diffusion_cs_t, mom6hip_eos_t
    type(c_ptr), value :: ctx, h, p_surf, conc_underflow
    type(mom6hip_tracer_hor_diff_cs_t), intent(in) :: cs
    type(mom6hip_neutral_diffusion_cs_t), intent(in) :: nd
    type(mom6hip_hordiff_fields_t), intent(in) :: fields
    type(mom6hip_eos_t), intent(in) :: eos
    real(c_double), value :: dt
    type(c_ptr), intent(in) :: tr(*)
    integer(c_int32_t), value :: ntr, idx_T, idx_S, memspace
    type(mom6hip_hordiff_stats_t), intent(out) :: stats
    integer(c_int) :: rc
  end function mom6hip_tracer_hordiff_neutral

  !> tracer_hordiff with CS%Diffuse_ML_interior (cs%unsupported(3)); idx_T, idx_S: the 0-based places of tv%T, tv%S in tr
  function mom6hip_tracer_hordiff_epipycnal(ctx, cs, epi, fields, h, eos, dt, tr, conc_underflow, ntr, idx_T, idx_S, memspace, &
                                            stats) bind(c, name="mom6hip_tracer_hordiff_epipycnal") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_tracer_hor_diff_cs_t, mom6hip_hordiff_stats_t, mom6hip_hordiff_fields_t, &
              mom6hip_epipycnal_cs_t, mom6hip_eos_t
    type(c_ptr), value :: ctx, h, conc_underflow
    type(mom6hip_tracer_hor_diff_cs_t), intent(in) :: cs
    type(mom6hip_epipycnal_cs_t), intent(in) :: epi
    type(mom6hip_hordiff_fields_t), intent(in) :: fields
    type(mom6hip_eos_t), intent(in) :: eos
    real(c_double), value :: dt
    type(c_ptr), intent(in) :: tr(*)
    integer(c_int32_t), value :: ntr, idx_T, idx_S, memspace
    type(mom6hip_hordiff_stats_t), intent(out) :: stats
    integer(c_int) :: rc
  end function mom6hip_tracer_hordiff_epipycnal

  !> subchk / subStats of MOM_checksums (MOM_checksums.F90:1387) on a device or host field of staggering pos
  function mom6hip_chksum(ctx, field, pos, nk, di, dj, symmetric, scale, bitcount, amin, amax, memspace) &
                          bind(c, name="mom6hip_chksum") result(rc)
    import :: c_int, c_int32_t, c_int64_t, c_double, c_ptr
    type(c_ptr), value :: ctx, field
    integer(c_int32_t), value :: pos, nk, di, dj, symmetric, memspace
    real(c_double), value :: scale
    integer(c_int64_t), intent(out) :: bitcount
    real(c_double), intent(out) :: amin, amax
    integer(c_int) :: rc
  end function mom6hip_chksum

  !> Restart / diagnostic staging: snapshot a device field and copy it to a (registered) host array on the copy stream
  function mom6hip_stage_to_host(ctx, hptr, dptr, bytes) bind(c, name="mom6hip_stage_to_host") result(rc)
    import :: c_int, c_int64_t, c_ptr
    type(c_ptr), value :: ctx, hptr, dptr
    integer(c_int64_t), value :: bytes
    integer(c_int) :: rc
  end function mom6hip_stage_to_host
  function mom6hip_stage_query(ctx, pending) bind(c, name="mom6hip_stage_query") result(rc)
    import :: c_int, c_int32_t, c_ptr
    type(c_ptr), value :: ctx
    integer(c_int32_t), intent(out) :: pending
    integer(c_int) :: rc
  end function mom6hip_stage_query
  function mom6hip_stage_wait(ctx) bind(c, name="mom6hip_stage_wait") result(rc)
    import :: c_int, c_ptr
    type(c_ptr), value :: ctx
    integer(c_int) :: rc
  end function mom6hip_stage_wait
  function mom6hip_host_register(hptr, bytes) bind(c, name="mom6hip_host_register") result(rc)
    import :: c_int, c_int64_t, c_ptr
    type(c_ptr), value :: hptr
    integer(c_int64_t), value :: bytes
    integer(c_int) :: rc
  end function mom6hip_host_register
  function mom6hip_host_unregister(hptr) bind(c, name="mom6hip_host_unregister") result(rc)
    import :: c_int, c_ptr
    type(c_ptr), value :: hptr
    integer(c_int) :: rc
  end function mom6hip_host_unregister

  !> The sums of write_energy for device or host fields; T, S, mass_lay, KE_lay may be c_null_ptr
  function mom6hip_write_energy_sums(ctx, u, v, h, T, S, dt, C_p, H_to_kg_m2, mass_lay, KE_lay, sums, memspace) &
                                     bind(c, name="mom6hip_write_energy_sums") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_energy_sums_t
    type(c_ptr), value :: ctx, u, v, h, T, S, mass_lay, KE_lay
    real(c_double), value :: dt, C_p, H_to_kg_m2
    type(mom6hip_energy_sums_t), intent(out) :: sums
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_write_energy_sums

  !> create_depth_list / depth_list_setup of MOM_sum_output for CALCULATE_APE; the tile's place in the global domain (0-based offsets)
  function mom6hip_depth_list_create(ctx, niglobal, njglobal, i_offset, j_offset, Z_ref, min_depth_inc, listsize) &
                                     bind(c, name="mom6hip_depth_list_create") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr
    type(c_ptr), value :: ctx
    integer(c_int32_t), value :: niglobal, njglobal, i_offset, j_offset
    real(c_double), value :: Z_ref, min_depth_inc
    integer(c_int32_t), intent(out) :: listsize
    integer(c_int) :: rc
  end function mom6hip_depth_list_create
  !> The available potential energy of write_energy: PE(nk+1), PE_tot, Z_0APE(nk+1); mass_lay from mom6hip_write_energy_sums
  function mom6hip_write_energy_ape(ctx, h, mass_lay, g_prime, Rho0, H_to_kg_m2, Z_ref, PE, PE_tot, Z_0APE, memspace) &
                                    bind(c, name="mom6hip_write_energy_ape") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr
    type(c_ptr), value :: ctx, h, mass_lay, g_prime, PE, Z_0APE
    real(c_double), value :: Rho0, H_to_kg_m2, Z_ref
    real(c_double), intent(out) :: PE_tot
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_write_energy_ape

  !> reproducing_sum of MOM_coms (MOM_coms.F90:318) over the h-point computational domain of a device or host field of
  !! staggering pos; lay_sums, efp_sum(6), efp_lay(6,nk), npoints and err are c_loc of the outputs or c_null_ptr
  function mom6hip_reproducing_sum(ctx, field, pos, nk, sum, lay_sums, efp_sum, efp_lay, npoints, err, memspace) &
                                   bind(c, name="mom6hip_reproducing_sum") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr
    type(c_ptr), value :: ctx, field, lay_sums, efp_sum, efp_lay, npoints, err
    integer(c_int32_t), value :: pos, nk, memspace
    real(c_double), intent(out) :: sum
    integer(c_int) :: rc
  end function mom6hip_reproducing_sum

  function mom6hip_coradcalc(ctx, cs, u, v, h, uh, vh, CAu, CAv, memspace) bind(c, name="mom6hip_coradcalc") result(rc)
    import :: c_int, c_int32_t, c_ptr, mom6hip_coriolisadv_cs_t
    type(c_ptr), value :: ctx, u, v, h, uh, vh, CAu, CAv
    type(mom6hip_coriolisadv_cs_t), intent(in) :: cs
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_coradcalc

  !> Optional Fortran arguments arrive as c_null_ptr; BT_cont is c_loc of a mom6hip_bt_cont_t or c_null_ptr
  function mom6hip_continuity(ctx, cs, u, v, hin, h, uh, vh, dt, uhbt, vhbt, visc_rem_u, visc_rem_v, u_cor, v_cor, &
                              BT_cont, du_cor, dv_cor, memspace) bind(c, name="mom6hip_continuity") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_continuity_cs_t
    type(c_ptr), value :: ctx, u, v, hin, h, uh, vh, uhbt, vhbt, visc_rem_u, visc_rem_v, u_cor, v_cor, BT_cont, du_cor, dv_cor
    type(mom6hip_continuity_cs_t), intent(in) :: cs
    real(c_double), value :: dt
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_continuity

  !> radiation_open_bdry_conds (the normal component: Orlanski, gradient, nudging), apply_normal_flow and the pass of u_new, v_new
  function mom6hip_radiation_open_bdry_conds(ctx, obc, gamma_uv, rx_max, rx_normal, ry_normal, u_new, u_old, v_new, v_old, dt, memspace) &
                                             bind(c, name="mom6hip_radiation_open_bdry_conds") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_obc_t
    type(c_ptr), value :: ctx, rx_normal, ry_normal, u_new, u_old, v_new, v_old
    type(mom6hip_obc_t), intent(in) :: obc
    real(c_double), value :: gamma_uv, rx_max, dt
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_radiation_open_bdry_conds

  function mom6hip_open_boundary_zero_normal_flow(ctx, obc, u, v, memspace) bind(c, name="mom6hip_open_boundary_zero_normal_flow") result(rc)
    import :: c_int, c_int32_t, c_ptr, mom6hip_obc_t
    type(c_ptr), value :: ctx, u, v
    type(mom6hip_obc_t), intent(in) :: obc
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_open_boundary_zero_normal_flow

  !> CorAdCalc with OBC associated
  function mom6hip_coradcalc_obc(ctx, cs, obc, u, v, h, uh, vh, CAu, CAv, memspace) bind(c, name="mom6hip_coradcalc_obc") result(rc)
    import :: c_int, c_int32_t, c_ptr, mom6hip_coriolisadv_cs_t, mom6hip_obc_t
    type(c_ptr), value :: ctx, u, v, h, uh, vh, CAu, CAv
    type(mom6hip_coriolisadv_cs_t), intent(in) :: cs
    type(mom6hip_obc_t), intent(in) :: obc
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_coradcalc_obc

  !> continuity_PPM with OBC associated: obc holds host pointers (its segment table and segnum arrays) whatever the memory space
  function mom6hip_continuity_obc(ctx, cs, obc, u, v, hin, h, uh, vh, dt, uhbt, vhbt, visc_rem_u, visc_rem_v, u_cor, v_cor, &
                                  BT_cont, du_cor, dv_cor, memspace) bind(c, name="mom6hip_continuity_obc") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_continuity_cs_t, mom6hip_obc_t
    type(c_ptr), value :: ctx, u, v, hin, h, uh, vh, uhbt, vhbt, visc_rem_u, visc_rem_v, u_cor, v_cor, BT_cont, du_cor, dv_cor
    type(mom6hip_continuity_cs_t), intent(in) :: cs
    type(mom6hip_obc_t), intent(in) :: obc
    real(c_double), value :: dt
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_continuity_obc

  !> eos: c_loc of a mom6hip_eos_t, or c_null_ptr without an equation of state (use_EOS = .false.; T and S may be null then)
  function mom6hip_pressureforce_fv_bouss(ctx, cs, eos, h, T, S, p_atm, PFu, PFv, pbce, eta, memspace) &
                                          bind(c, name="mom6hip_pressureforce_fv_bouss") result(rc)
    import :: c_int, c_int32_t, c_ptr, mom6hip_pressureforce_cs_t
    type(c_ptr), value :: ctx, eos, h, T, S, p_atm, PFu, PFv, pbce, eta
    type(mom6hip_pressureforce_cs_t), intent(in) :: cs
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_pressureforce_fv_bouss

  function mom6hip_pressureforce_fv_nonbouss(ctx, cs, eos, h, T, S, p_atm, H_to_RZ, PFu, PFv, pbce, eta, memspace) &
                                          bind(c, name="mom6hip_pressureforce_fv_nonbouss") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_pressureforce_cs_t, mom6hip_eos_t
    type(c_ptr), value :: ctx, h, T, S, p_atm, PFu, PFv, pbce, eta
    type(mom6hip_pressureforce_cs_t), intent(in) :: cs
    type(mom6hip_eos_t), intent(in) :: eos
    real(c_double), value :: H_to_RZ
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_pressureforce_fv_nonbouss

  function mom6hip_calculate_density(ctx, eos, T, S, pressure, rho, n, use_rho_ref, rho_ref, memspace) &
                                     bind(c, name="mom6hip_calculate_density") result(rc)
    import :: c_int, c_int32_t, c_int64_t, c_double, c_ptr, mom6hip_eos_t
    type(c_ptr), value :: ctx, T, S, pressure, rho
    type(mom6hip_eos_t), intent(in) :: eos
    integer(c_int64_t), value :: n
    integer(c_int32_t), value :: use_rho_ref, memspace
    real(c_double), value :: rho_ref
    integer(c_int) :: rc
  end function mom6hip_calculate_density

  function mom6hip_barotropic_init(ctx, cs, memspace) bind(c, name="mom6hip_barotropic_init") result(rc)
    import :: c_int, c_int32_t, c_ptr, mom6hip_barotropic_cs_t
    type(c_ptr), value :: ctx
    type(mom6hip_barotropic_cs_t), intent(inout) :: cs
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_barotropic_init

  function mom6hip_btcalc(ctx, cs, h, h_u, h_v, may_use_default, memspace) bind(c, name="mom6hip_btcalc") result(rc)
    import :: c_int, c_int32_t, c_ptr, mom6hip_barotropic_cs_t
    type(c_ptr), value :: ctx, h, h_u, h_v
    type(mom6hip_barotropic_cs_t), intent(inout) :: cs
    integer(c_int32_t), value :: may_use_default, memspace
    integer(c_int) :: rc
  end function mom6hip_btcalc

  !> btcalc with OBC associated
  function mom6hip_btcalc_obc(ctx, cs, h, h_u, h_v, may_use_default, obc, memspace) bind(c, name="mom6hip_btcalc_obc") result(rc)
    import :: c_int, c_int32_t, c_ptr, mom6hip_barotropic_cs_t, mom6hip_obc_t
    type(c_ptr), value :: ctx, h, h_u, h_v
    type(mom6hip_barotropic_cs_t), intent(inout) :: cs
    type(mom6hip_obc_t), intent(in) :: obc
    integer(c_int32_t), value :: may_use_default, memspace
    integer(c_int) :: rc
  end function mom6hip_btcalc_obc

  function mom6hip_bt_mass_source(ctx, cs, h, eta, set_cor, memspace) bind(c, name="mom6hip_bt_mass_source") result(rc)
    import :: c_int, c_int32_t, c_ptr, mom6hip_barotropic_cs_t
    type(c_ptr), value :: ctx, h, eta
    type(mom6hip_barotropic_cs_t), intent(inout) :: cs
    integer(c_int32_t), value :: set_cor, memspace
    integer(c_int) :: rc
  end function mom6hip_bt_mass_source

  function mom6hip_set_dtbt(ctx, cs, pbce, BT_cont, gtot_est, SSH_add, memspace) bind(c, name="mom6hip_set_dtbt") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_barotropic_cs_t
    type(c_ptr), value :: ctx, pbce, BT_cont
    type(mom6hip_barotropic_cs_t), intent(inout) :: cs
    real(c_double), value :: gtot_est, SSH_add
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_set_dtbt

  function mom6hip_set_dtbt_eta(ctx, cs, eta, pbce, BT_cont, gtot_est, SSH_add, memspace) bind(c, name="mom6hip_set_dtbt_eta") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_barotropic_cs_t
    type(c_ptr), value :: ctx, eta, pbce, BT_cont
    type(mom6hip_barotropic_cs_t), intent(inout) :: cs
    real(c_double), value :: gtot_est, SSH_add
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_set_dtbt_eta

  function mom6hip_btstep(ctx, cs, U_in, V_in, eta_in, dt, bc_accel_u, bc_accel_v, taux, tauy, RZ_to_H, pbce, eta_PF_in, &
                          U_Cor, V_Cor, accel_layer_u, accel_layer_v, eta_out, uhbtav, vhbtav, visc_rem_u, visc_rem_v, &
                          BT_cont, eta_PF_start, taux_bot, tauy_bot, uh0, vh0, u_uh0, v_vh0, etaav, memspace) &
                          bind(c, name="mom6hip_btstep") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_barotropic_cs_t
    type(c_ptr), value :: ctx, U_in, V_in, eta_in, bc_accel_u, bc_accel_v, taux, tauy, pbce, eta_PF_in, U_Cor, V_Cor, &
                          accel_layer_u, accel_layer_v, eta_out, uhbtav, vhbtav, visc_rem_u, visc_rem_v, BT_cont, &
                          eta_PF_start, taux_bot, tauy_bot, uh0, vh0, u_uh0, v_vh0, etaav
    type(mom6hip_barotropic_cs_t), intent(inout) :: cs
    real(c_double), value :: dt, RZ_to_H
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_btstep

  !> btstep with OBC associated (specified, Flather and gradient segments)
  function mom6hip_btstep_obc(ctx, cs, U_in, V_in, eta_in, dt, bc_accel_u, bc_accel_v, taux, tauy, RZ_to_H, pbce, eta_PF_in, &
                              U_Cor, V_Cor, accel_layer_u, accel_layer_v, eta_out, uhbtav, vhbtav, visc_rem_u, visc_rem_v, &
                              BT_cont, eta_PF_start, taux_bot, tauy_bot, uh0, vh0, u_uh0, v_vh0, etaav, obc, memspace) &
                              bind(c, name="mom6hip_btstep_obc") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_barotropic_cs_t, mom6hip_obc_t
    type(c_ptr), value :: ctx, U_in, V_in, eta_in, bc_accel_u, bc_accel_v, taux, tauy, pbce, eta_PF_in, U_Cor, V_Cor, &
                          accel_layer_u, accel_layer_v, eta_out, uhbtav, vhbtav, visc_rem_u, visc_rem_v, BT_cont, &
                          eta_PF_start, taux_bot, tauy_bot, uh0, vh0, u_uh0, v_vh0, etaav
    type(mom6hip_barotropic_cs_t), intent(inout) :: cs
    real(c_double), value :: dt, RZ_to_H
    type(mom6hip_obc_t), intent(in) :: obc
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_btstep_obc

  !> vertvisc_coef (MOM_vert_friction.F90:1168); dz = c_null_ptr stands for the Boussinesq thickness_to_dz
  function mom6hip_vertvisc_coef(ctx, cs, u, v, h, dz, visc, dt, memspace) bind(c, name="mom6hip_vertvisc_coef") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_vertvisc_cs_t, mom6hip_vertvisc_type_t
    type(c_ptr), value :: ctx, u, v, h, dz
    type(mom6hip_vertvisc_cs_t), intent(inout) :: cs
    type(mom6hip_vertvisc_type_t), intent(in) :: visc
    real(c_double), value :: dt
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_vertvisc_coef

  !> vertvisc (:526) with vertvisc_limit_vel; taux_bot / tauy_bot may be c_null_ptr
  function mom6hip_vertvisc(ctx, cs, u, v, h, taux, tauy, visc, dt, taux_bot, tauy_bot, memspace) &
                            bind(c, name="mom6hip_vertvisc") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_vertvisc_cs_t, mom6hip_vertvisc_type_t
    type(c_ptr), value :: ctx, u, v, h, taux, tauy, taux_bot, tauy_bot
    type(mom6hip_vertvisc_cs_t), intent(inout) :: cs
    type(mom6hip_vertvisc_type_t), intent(in) :: visc
    real(c_double), value :: dt
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_vertvisc

  !> vertvisc_coef and vertvisc with OBC associated (the projections at the segments' faces; the specified segments' velocities)
  function mom6hip_vertvisc_coef_obc(ctx, cs, u, v, h, dz, visc, dt, obc, memspace) bind(c, name="mom6hip_vertvisc_coef_obc") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_vertvisc_cs_t, mom6hip_vertvisc_type_t, mom6hip_obc_t
    type(c_ptr), value :: ctx, u, v, h, dz
    type(mom6hip_vertvisc_cs_t), intent(inout) :: cs
    type(mom6hip_vertvisc_type_t), intent(in) :: visc
    real(c_double), value :: dt
    type(mom6hip_obc_t), intent(in) :: obc
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_vertvisc_coef_obc
  function mom6hip_vertvisc_obc(ctx, cs, u, v, h, taux, tauy, visc, dt, taux_bot, tauy_bot, obc, memspace) &
                                bind(c, name="mom6hip_vertvisc_obc") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_vertvisc_cs_t, mom6hip_vertvisc_type_t, mom6hip_obc_t
    type(c_ptr), value :: ctx, u, v, h, taux, tauy, taux_bot, tauy_bot
    type(mom6hip_vertvisc_cs_t), intent(inout) :: cs
    type(mom6hip_vertvisc_type_t), intent(in) :: visc
    real(c_double), value :: dt
    type(mom6hip_obc_t), intent(in) :: obc
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_vertvisc_obc

  !> vertvisc followed by vertvisc_remnant with the same dt, in one pass
  function mom6hip_vertvisc_and_remnant(ctx, cs, u, v, h, taux, tauy, visc, dt, taux_bot, tauy_bot, visc_rem_u, visc_rem_v, &
                                        memspace) bind(c, name="mom6hip_vertvisc_and_remnant") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_vertvisc_cs_t, mom6hip_vertvisc_type_t
    type(c_ptr), value :: ctx, u, v, h, taux, tauy, taux_bot, tauy_bot, visc_rem_u, visc_rem_v
    type(mom6hip_vertvisc_cs_t), intent(inout) :: cs
    type(mom6hip_vertvisc_type_t), intent(in) :: visc
    real(c_double), value :: dt
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_vertvisc_and_remnant

  !> vertvisc_coef, then (update_velocities /= 0) vertvisc, then vertvisc_remnant with the same dt, one kernel per direction
  function mom6hip_vertvisc_step(ctx, cs, u, v, h, dz, taux, tauy, visc, dt, update_velocities, taux_bot, tauy_bot, &
                                 visc_rem_u, visc_rem_v, memspace) bind(c, name="mom6hip_vertvisc_step") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_vertvisc_cs_t, mom6hip_vertvisc_type_t
    type(c_ptr), value :: ctx, u, v, h, dz, taux, tauy, taux_bot, tauy_bot, visc_rem_u, visc_rem_v
    type(mom6hip_vertvisc_cs_t), intent(inout) :: cs
    type(mom6hip_vertvisc_type_t), intent(in) :: visc
    real(c_double), value :: dt
    integer(c_int32_t), value :: update_velocities, memspace
    integer(c_int) :: rc
  end function mom6hip_vertvisc_step

  !> CS%ntrunc: adds the truncations counted on the device since the last call (synchronises)
  function mom6hip_vertvisc_ntrunc(ctx, cs) bind(c, name="mom6hip_vertvisc_ntrunc") result(rc)
    import :: c_int, c_ptr, mom6hip_vertvisc_cs_t
    type(c_ptr), value :: ctx
    type(mom6hip_vertvisc_cs_t), intent(inout) :: cs
    integer(c_int) :: rc
  end function mom6hip_vertvisc_ntrunc

  !> vertvisc_remnant (:1064)
  function mom6hip_vertvisc_remnant(ctx, cs, visc, visc_rem_u, visc_rem_v, dt, memspace) &
                                    bind(c, name="mom6hip_vertvisc_remnant") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_vertvisc_cs_t, mom6hip_vertvisc_type_t
    type(c_ptr), value :: ctx, visc_rem_u, visc_rem_v
    type(mom6hip_vertvisc_cs_t), intent(in) :: cs
    type(mom6hip_vertvisc_type_t), intent(in) :: visc
    real(c_double), value :: dt
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_vertvisc_remnant

  !> set_viscous_BBL (MOM_set_viscosity.F90:134); tv%T, tv%S, tv%eqn_of_state as T, S, eos
  function mom6hip_set_viscous_bbl(ctx, cs, u, v, h, T, S, eos, visc, memspace) bind(c, name="mom6hip_set_viscous_bbl") result(rc)
    import :: c_int, c_int32_t, c_ptr, mom6hip_set_visc_cs_t, mom6hip_vertvisc_type_t
    type(c_ptr), value :: ctx, u, v, h, T, S, eos
    type(mom6hip_set_visc_cs_t), intent(in) :: cs
    type(mom6hip_vertvisc_type_t), intent(in) :: visc
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_set_viscous_bbl

  !> set_viscous_BBL with CS%OBC associated
  function mom6hip_set_viscous_bbl_obc(ctx, cs, u, v, h, T, S, eos, visc, obc, memspace) bind(c, name="mom6hip_set_viscous_bbl_obc") result(rc)
    import :: c_int, c_int32_t, c_ptr, mom6hip_set_visc_cs_t, mom6hip_vertvisc_type_t, mom6hip_obc_t
    type(c_ptr), value :: ctx, u, v, h, T, S, eos
    type(mom6hip_set_visc_cs_t), intent(in) :: cs
    type(mom6hip_vertvisc_type_t), intent(in) :: visc
    type(mom6hip_obc_t), intent(in) :: obc
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_set_viscous_bbl_obc

  !> set_viscous_ML (:1898): the early return (:2043), or with DYNAMIC_VISCOUS_ML the viscous mixed layer into visc%nkml_visc_u/v
  function mom6hip_set_viscous_ml(ctx, cs, u, v, h, T, S, eos, taux, tauy, visc, dt, memspace) &
                                  bind(c, name="mom6hip_set_viscous_ml") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_set_visc_cs_t, mom6hip_vertvisc_type_t
    type(c_ptr), value :: ctx, u, v, h, T, S, eos, taux, tauy
    type(mom6hip_set_visc_cs_t), intent(in) :: cs
    type(mom6hip_vertvisc_type_t), intent(in) :: visc
    real(c_double), value :: dt
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_set_viscous_ml

  !> thickness_diffuse (MOM_thickness_diffuse.F90:133); T, S, eos are c_null_ptr without an equation of state
  function mom6hip_thickness_diffuse(ctx, cs, h, uhtr, vhtr, T, S, eos, dt, uhGM, vhGM, memspace) &
      bind(c, name="mom6hip_thickness_diffuse") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_thickness_diffuse_cs_t
    type(c_ptr), value :: ctx
    type(mom6hip_thickness_diffuse_cs_t), intent(in) :: cs
    type(c_ptr), value :: h, uhtr, vhtr, T, S, eos
    real(c_double), value :: dt
    type(c_ptr), value :: uhGM, vhGM
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_thickness_diffuse

  !> mixedlayer_restrat (MOM_mixed_layer_restrat.F90:135); h_MLD, uhml, vhml may be c_null_ptr
  function mom6hip_mixedlayer_restrat(ctx, cs, h, uhtr, vhtr, T, S, eos, ustar, dt, h_MLD, uhml, vhml, memspace) &
      bind(c, name="mom6hip_mixedlayer_restrat") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_mixedlayer_restrat_cs_t
    type(c_ptr), value :: ctx
    type(mom6hip_mixedlayer_restrat_cs_t), intent(in) :: cs
    type(c_ptr), value :: h, uhtr, vhtr, T, S, eos, ustar
    real(c_double), value :: dt
    type(c_ptr), value :: h_MLD, uhml, vhml
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_mixedlayer_restrat
  !> mu(sigma, dh) (MOM_mixed_layer_restrat.F90:723)
  function mom6hip_mixedlayer_restrat_mu(sigma, dh) bind(c, name="mom6hip_mixedlayer_restrat_mu") result(mu)
    import :: c_double
    real(c_double), value :: sigma, dh
    real(c_double) :: mu
  end function mom6hip_mixedlayer_restrat_mu

  !> hor_visc_init (MOM_hor_visc.F90:1984): the static arrays of the control structure
  function mom6hip_hor_visc_init(ctx, cs, dt, memspace) bind(c, name="mom6hip_hor_visc_init") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_hor_visc_cs_t
    type(c_ptr), value :: ctx
    type(mom6hip_hor_visc_cs_t), intent(inout) :: cs
    real(c_double), value :: dt
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_hor_visc_init

  !> horizontal_viscosity (:245)
  function mom6hip_horizontal_viscosity(ctx, cs, u, v, h, diffu, diffv, dt, hu_cont, hv_cont, memspace) &
                                        bind(c, name="mom6hip_horizontal_viscosity") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_hor_visc_cs_t
    type(c_ptr), value :: ctx, u, v, h, diffu, diffv, hu_cont, hv_cont
    type(mom6hip_hor_visc_cs_t), intent(in) :: cs
    real(c_double), value :: dt
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_horizontal_viscosity

  !> horizontal_viscosity with OBC associated
  function mom6hip_horizontal_viscosity_obc(ctx, cs, u, v, h, diffu, diffv, dt, hu_cont, hv_cont, obc, memspace) &
                                            bind(c, name="mom6hip_horizontal_viscosity_obc") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_hor_visc_cs_t, mom6hip_obc_t
    type(c_ptr), value :: ctx, u, v, h, diffu, diffv, hu_cont, hv_cont
    type(mom6hip_hor_visc_cs_t), intent(in) :: cs
    real(c_double), value :: dt
    type(mom6hip_obc_t), intent(in) :: obc
    integer(c_int32_t), value :: memspace
    integer(c_int) :: rc
  end function mom6hip_horizontal_viscosity_obc

  function mom6hip_dyn_split_rk2_init(ctx, cs, u, v, h, uh, vh, dt) bind(c, name="mom6hip_dyn_split_rk2_init") result(rc)
    import :: c_int, c_double, c_ptr, mom6hip_dyn_split_rk2_cs_t
    type(c_ptr), value :: ctx, u, v, h, uh, vh
    type(mom6hip_dyn_split_rk2_cs_t), intent(inout) :: cs
    real(c_double), value :: dt
    integer(c_int) :: rc
  end function mom6hip_dyn_split_rk2_init

  function mom6hip_step_dyn_split_rk2(ctx, cs, u_inst, v_inst, h, T, S, dt, taux, tauy, RZ_to_H, uh, vh, uhtr, vhtr, &
                                      eta_av, calc_dtbt) bind(c, name="mom6hip_step_dyn_split_rk2") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_dyn_split_rk2_cs_t
    type(c_ptr), value :: ctx, u_inst, v_inst, h, T, S, taux, tauy, uh, vh, uhtr, vhtr, eta_av
    type(mom6hip_dyn_split_rk2_cs_t), intent(inout) :: cs
    real(c_double), value :: dt, RZ_to_H
    integer(c_int32_t), value :: calc_dtbt
    integer(c_int) :: rc
  end function mom6hip_step_dyn_split_rk2

  function mom6hip_dyn_split_rk2b_init(ctx, cs, h) bind(c, name="mom6hip_dyn_split_rk2b_init") result(rc)
    import :: c_int, c_ptr, mom6hip_dyn_split_rk2_cs_t
    type(c_ptr), value :: ctx, h
    type(mom6hip_dyn_split_rk2_cs_t), intent(inout) :: cs
    integer(c_int) :: rc
  end function mom6hip_dyn_split_rk2b_init

  function mom6hip_step_dyn_split_rk2b(ctx, cs, u_av, v_av, h, T, S, dt, taux, tauy, RZ_to_H, uh, vh, uhtr, vhtr, &
                                       eta_av, calc_dtbt) bind(c, name="mom6hip_step_dyn_split_rk2b") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_dyn_split_rk2_cs_t
    type(c_ptr), value :: ctx, u_av, v_av, h, T, S, taux, tauy, uh, vh, uhtr, vhtr, eta_av
    type(mom6hip_dyn_split_rk2_cs_t), intent(inout) :: cs
    real(c_double), value :: dt, RZ_to_H
    integer(c_int32_t), value :: calc_dtbt
    integer(c_int) :: rc
  end function mom6hip_step_dyn_split_rk2b
end interface

contains

!> The library's last error message as a Fortran string.
function mom6hip_error_string() result(str)
  character(len=:), allocatable :: str
  character(kind=c_char), pointer :: chars(:)
  type(c_ptr) :: p
  integer :: n
  p = mom6hip_last_error()
  str = ""
  if (.not. c_associated(p)) return
  call c_f_pointer(p, chars, [1024])
  n = 0
  do while (n < 1024)
    if (chars(n+1) == c_null_char) exit
    n = n + 1
  enddo
  allocate(character(len=n) :: str)
  if (n > 0) str = transfer(chars(1:n), str)
end function mom6hip_error_string

end module mom6hip_c_api
