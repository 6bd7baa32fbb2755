"""ctypes mirror of include/mom6hip.h (the C ABI of libmom6hip).

Field order and types must match the header exactly; tests/test_abi.py checks sizes/offsets
against the compiled library (mom6hip_abi_sizeof_*).
"""
import ctypes as C

MEM_HOST, MEM_DEVICE = 0, 1
ADV_PLM, ADV_PPM_H3, ADV_PPM = 0, 1, 2
POS_H, POS_U, POS_V, POS_Q = 0, 1, 2, 3

ADV_SCHEMES = {"PLM": ADV_PLM, "PPM:H3": ADV_PPM_H3, "PPM": ADV_PPM}

_dp = C.POINTER(C.c_double)

H_METRICS = ("mask2dT", "areaT", "IareaT", "dxT", "dyT", "IdxT", "IdyT", "bathyT")
U_METRICS = ("mask2dCu", "dxCu", "dyCu", "dy_Cu", "IdxCu", "IdyCu", "areaCu", "IareaCu")
V_METRICS = ("mask2dCv", "dxCv", "dyCv", "dx_Cv", "IdxCv", "IdyCv", "areaCv", "IareaCv")
Q_METRICS = ("mask2dBu", "dxBu", "dyBu", "areaBu", "IareaBu", "CoriolisBu", "IdxBu", "IdyBu")
ALL_METRICS = H_METRICS + U_METRICS + V_METRICS + Q_METRICS


PASS_SCALAR_PAIR = 8      # MOM6HIP_PASS_SCALAR_PAIR: ORed into the position of a u/v field that is one of a SCALAR_PAIR


class GridStruct(C.Structure):
    _fields_ = (
        [(n, C.c_int32) for n in ("isc", "iec", "jsc", "jec", "isd", "ied", "jsd", "jed", "nk",
                                  "symmetric", "reentrant_x", "reentrant_y", "first_direction",
                                  "tripolar_n")]
        + [(n, C.c_double) for n in ("Angstrom_H", "H_subroundoff", "dZ_subroundoff", "H_to_Z",
                                     "Z_to_H", "g_Earth", "Rho0")]
        + [("reserved1", C.c_double * 8)]
        + [(n, _dp) for n in ALL_METRICS]
        + [("reserved2", C.c_void_p * 6)]
    )


class DomainStruct(C.Structure):
    """mom6hip_domain_t (include/mom6hip.h)."""
    _fields_ = [(n, C.c_int32) for n in ("nranks", "rank", "nbr_w", "nbr_e", "nbr_s", "nbr_n")] + [("reserved", C.c_int32 * 2)]


class TracerAdvectCS(C.Structure):
    _fields_ = [("dt", C.c_double), ("scheme", C.c_int32), ("use_huynh_stencil_bug", C.c_int32)]


class AdvectStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("halo_updates", C.c_int32),
                ("domore_remaining", C.c_int32), ("reserved", C.c_int32)]


class AdvectTiming(C.Structure):
    _fields_ = [("ms_total", C.c_double), ("ms_setup", C.c_double), ("ms_x", C.c_double),
                ("ms_y", C.c_double), ("ms_halo", C.c_double), ("ms_x1", C.c_double),
                ("ms_y1", C.c_double), ("n_x", C.c_int32),
                ("n_y", C.c_int32)]


REMAP_SCHEMES = {"PCM": 0, "PLM": 2, "PLM_HYBGEN": 3, "PPM_H4": 4, "PPM_IH4": 5, "PPM_HYBGEN": 6, "WENO_HYBGEN": 7, "PQM_IH4IH3": 8, "PQM_IH6IH5": 9, "PPM_CW": 10}


class RemappingCS(C.Structure):
    _fields_ = [("remapping_scheme", C.c_int32), ("boundary_extrapolation", C.c_int32),
                ("force_bounds_in_subcell", C.c_int32), ("answer_date", C.c_int32)]

CORIOLIS_SCHEMES = {"SADOURNY75_ENERGY": 1, "ARAKAWA_HSU90": 2, "ROBUST_ENSTRO": 3, "SADOURNY75_ENSTRO": 4, "ARAKAWA_LAMB81": 5,
                    "ARAKAWA_LAMB_BLEND": 6}
PV_ADV_SCHEMES = {"PV_ADV_CENTERED": 21, "PV_ADV_UPWIND1": 22}
KE_SCHEMES = {"KE_ARAKAWA": 10, "KE_SIMPLE_GUDONOV": 11, "KE_GUDONOV": 12}


class CoriolisAdvCS(C.Structure):
    _fields_ = [("coriolis_scheme", C.c_int32), ("ke_scheme", C.c_int32), ("no_slip", C.c_int32),
                ("bound_coriolis", C.c_int32), ("coriolis_en_dis", C.c_int32), ("pv_adv_scheme", C.c_int32), ("reserved", C.c_int32 * 2),
                ("F_eff_max_blend", C.c_double), ("wt_lin_blend", C.c_double)]


class ContinuityCS(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("upwind_1st", "monotonic", "simple_2nd", "aggress_adjust", "vol_CFL",
                                         "better_iter", "use_visc_rem_max", "marginal_faces")] + \
               [("tol_eta", C.c_double), ("tol_vel", C.c_double), ("CFL_limit_adjust", C.c_double)]


BT_CONT_U = ("FA_u_W0", "FA_u_WW", "FA_u_E0", "FA_u_EE", "uBT_WW", "uBT_EE")
BT_CONT_V = ("FA_v_S0", "FA_v_SS", "FA_v_N0", "FA_v_NN", "vBT_SS", "vBT_NN")


class BTCont(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in BT_CONT_U + BT_CONT_V + ("h_u", "h_v")]


EOS_FORMS = {"LINEAR": 1, "UNESCO": 2, "JACKETT_MCD": 2, "WRIGHT": 3, "WRIGHT_FULL": 4, "WRIGHT_REDUCED": 5}


class EOS(C.Structure):
    _fields_ = [("form", C.c_int32), ("reserved", C.c_int32), ("Rho_T0_S0", C.c_double), ("dRho_dT", C.c_double),
                ("dRho_dS", C.c_double)]


class PressureForceCS(C.Structure):
    _fields_ = [("Rho0", C.c_double), ("GFS_scale", C.c_double), ("Z_ref", C.c_double), ("reconstruct", C.c_int32),
                ("Recon_Scheme", C.c_int32), ("boundary_extrap", C.c_int32), ("useMassWghtInterp", C.c_int32),
                ("use_ALE", C.c_int32), ("nkmb", C.c_int32), ("P_Ref", C.c_double), ("Rlay", C.c_void_p), ("g_prime", C.c_void_p)]


# ---- MOM_barotropic -----------------------------------------------------------------------------------
BT_THICK_SCHEMES = {"HARMONIC": 1, "ARITHMETIC": 2, "HYBRID": 3, "FROM_BT_CONT": 4}
BT_UNSUPPORTED = ("INTEGRAL_BT_CONTINUITY", "(free slot)", "(free slot 2)", "BOUND_BT_CORRECTION without BT_CONT_CORR_BOUNDS",
                  "GRADUAL_BT_ICS", "BT_NONLIN_STRESS", "DYNAMIC_SURFACE_PRESSURE", "BT_LINEAR_WAVE_DRAG",
                  "CLIP_BT_VELOCITY", "CALCULATE_SAL", "BT_USE_OLD_CORIOLIS_BRACKET_BUG", "BAROTROPIC_ANSWER_DATE<20190101")
BT_CS_ARRAYS = (("frhatu", POS_U, 3), ("frhatv", POS_V, 3), ("eta_cor", POS_H, 2), ("IDatu", POS_U, 2), ("IDatv", POS_V, 2),
                ("ubtav", POS_U, 2), ("vbtav", POS_V, 2), ("q_D", POS_Q, 2), ("D_u_Cor", POS_U, 2), ("D_v_Cor", POS_V, 2))


class BarotropicCS(C.Structure):
    """mom6hip_barotropic_cs_t (include/mom6hip.h)."""
    _fields_ = ([(n, C.c_double) for n in ("dtbt", "dtbt_max", "dtbt_fraction", "bebt", "dt_bt_filter", "vel_underflow",
                                            "G_extra", "BT_Coriolis_scale", "Z_ref", "maxCFL_BT_cont")]
                + [("reserved0", C.c_double * 6)]
                + [(n, C.c_int32) for n in ("Sadourny", "linearized_BT_PV", "strong_drag", "visc_rem_u_uh0", "adjust_BT_cont",
                                           "use_wide_halos", "hvel_scheme", "nstep_last")]
                + [("unsupported", C.c_int32 * 12), ("bound_BT_corr", C.c_int32), ("BT_project_velocity", C.c_int32), ("Nonlinear_continuity", C.c_int32),
                   ("Nonlin_cont_update_period", C.c_int32)]
                + [(n, C.c_void_p) for n, _, _ in BT_CS_ARRAYS]
                + [("reserved2", C.c_void_p * 6)])


# ---- MOM_set_viscosity ------------------------------------------------------------------------------------
SET_VISC_UNSUPPORTED = ("(free 0)", "BBL_use_tidal_bg", "(free)", "(free 2)", "non_Boussinesq", "p_surf", "OBC", "pbv", "ice_shelf")


class SetViscCS(C.Structure):
    """mom6hip_set_visc_cs_t (include/mom6hip.h)."""
    _fields_ = ([(n, C.c_double) for n in ("cdrag", "drag_bg_vel", "Hbbl", "dz_bbl", "BBL_thick_min", "Kv_BBL_min", "BBL_thick_max", "H_to_RZ")]
                + [(n, C.c_double) for n in ("omega", "omega_frac", "ustar_min", "TKE_decay", "bulk_Ri_ML", "c_Smag", "Chan_drag_max_vol")]
                + [("Z_ref", C.c_double)]
                + [(n, C.c_int32) for n in ("bottomdraglaw", "linear_drag", "BBL_use_EOS", "correct_BBL_bounds", "body_force_drag", "RiNo_mix",
                                           "initialized")]
                + [("unsupported", C.c_int32 * 9)]
                + [("Rlay", C.c_void_p)]
                + [(n, C.c_int32) for n in ("dynamic_viscous_ML", "nkml", "Channel_drag", "concave_trigonometric_L")]
                + [("reserved1", C.c_void_p * 1)])


# ---- MOM_hor_visc -----------------------------------------------------------------------------------------
HOR_VISC_UNSUPPORTED = ("Leith_Kh", "Leith_Ah", "use_Leithy", "MEKE_backscatter", "use_GME", "anisotropic", "Re_Ah", "Kh_sin_lat", "use_Kh_bg_2d",
                        "use_ZB2020")
HOR_VISC_ARRAYS_H = ("Kh_bg_xx", "Kh_Max_xx", "Ah_bg_xx", "Ah_Max_xx", "Laplac2_const_xx", "Biharm_const_xx", "Biharm_const2_xx", "reduction_xx")
HOR_VISC_ARRAYS_Q = ("Kh_bg_xy", "Kh_Max_xy", "Ah_bg_xy", "Ah_Max_xy", "Laplac2_const_xy", "Biharm_const_xy", "Biharm_const2_xy", "reduction_xy")


TRACER_HORDIFF_UNSUPPORTED = ("USE_NEUTRAL_DIFFUSION", "USE_HORIZONTAL_BOUNDARY_DIFFUSION", "DIFFUSE_ML_TO_INTERIOR", "VarMix", "MEKE",
                              "KHTR_USE_EBT_STRUCT", "offline", "flux_diagnostics")


class TracerHorDiffCS(C.Structure):
    """mom6hip_tracer_hor_diff_cs_t (include/mom6hip.h)."""
    _fields_ = [("KhTr", C.c_double), ("max_diff_CFL", C.c_double), ("KhTr_Slope_Cff", C.c_double), ("KhTr_fac", C.c_double), ("KhTr_min", C.c_double),
                ("KhTr_max", C.c_double), ("KhTr_passivity_coeff", C.c_double), ("KhTr_passivity_min", C.c_double), ("check_diffusive_CFL", C.c_int32),
                ("initialized", C.c_int32), ("unsupported", C.c_int32 * 8), ("use_variable_mixing", C.c_int32), ("Resoln_scaled_KhTr", C.c_int32),
                ("reserved1", C.c_int32 * 4)]


HORDIFF_FIELDS = ("MEKE_Kh", "L2u", "L2v", "SN_u", "SN_v", "Res_fn_h", "Rd_dx_h", "h_ML")


class HorDiffFields(C.Structure):
    """mom6hip_hordiff_fields_t (include/mom6hip.h)."""
    _fields_ = [(n, C.c_void_p) for n in HORDIFF_FIELDS] + [("reserved", C.c_void_p * 4)]


class NeutralDiffusionCS(C.Structure):
    """mom6hip_neutral_diffusion_cs_t (include/mom6hip.h)."""
    _fields_ = [("ref_pres", C.c_double), ("H_to_RZ", C.c_double), ("reserved0", C.c_double * 4), ("ndiff_answer_date", C.c_int32),
                ("recalc_neutral_surf", C.c_int32), ("initialized", C.c_int32), ("interior_only", C.c_int32), ("unsupported", C.c_int32 * 8)]


class EpipycnalCS(C.Structure):
    """mom6hip_epipycnal_cs_t (include/mom6hip.h)."""
    _fields_ = [("ML_KhTr_scale", C.c_double), ("P_Ref", C.c_double), ("reserved0", C.c_double * 4), ("Rlay", C.c_void_p), ("nkml", C.c_int32),
                ("nk_rho_varies", C.c_int32), ("answer_date", C.c_int32), ("limit_bug", C.c_int32), ("reserved1", C.c_int32 * 4)]


OBC_NONE, OBC_DIRECTION_N, OBC_DIRECTION_S, OBC_DIRECTION_E, OBC_DIRECTION_W = 0, 100, 200, 300, 400


OBC_TAN_RADIATION, OBC_GRAD_RADIATION, OBC_TAN_NUDGED, OBC_GRAD_NUDGED, OBC_TAN_OBLIQUE, OBC_GRAD_OBLIQUE = 1, 2, 4, 8, 16, 32


class ObcSegmentTracer(C.Structure):
    """mom6hip_obc_segment_tracer_t (include/mom6hip.h)."""
    _fields_ = [("ntr_index", C.c_int32), ("reserved", C.c_int32), ("tres", C.c_void_p), ("OBC_inflow_conc", C.c_double),
                ("t", C.c_void_p), ("resrv_lfac_in", C.c_double), ("resrv_lfac_out", C.c_double)]


class ObcSegment(C.Structure):
    """mom6hip_obc_segment_t (include/mom6hip.h)."""
    _fields_ = [(n, C.c_int32) for n in ("direction", "open", "specified", "on_pe", "is_E_or_W", "is_N_or_S", "IsdB", "IedB", "JsdB", "JedB",
                                         "isd", "ied", "jsd", "jed", "radiation", "gradient", "nudged", "oblique", "radiation_tan_or_grad")] + \
               [("Flather", C.c_int32), ("normal_trans", C.c_void_p), ("normal_vel", C.c_void_p), ("tangential_vel", C.c_void_p),
                ("tangential_grad", C.c_void_p), ("nudged_normal_vel", C.c_void_p), ("normal_vel_bt", C.c_void_p), ("SSH", C.c_void_p),
                ("Velocity_nudging_timescale_in", C.c_double), ("Velocity_nudging_timescale_out", C.c_double),
                ("tr_Reg", C.POINTER(ObcSegmentTracer)), ("ntseg", C.c_int32), ("reserved_i", C.c_int32),
                ("Tr_InvLscale_in", C.c_double), ("Tr_InvLscale_out", C.c_double),
                ("nudged_tangential_vel", C.c_void_p), ("nudged_tangential_grad", C.c_void_p)]


class Obc(C.Structure):
    """mom6hip_obc_t (include/mom6hip.h)."""
    _fields_ = [(n, C.c_int32) for n in ("number_of_segments", "OBC_pe", "open_u_BCs_exist_globally", "open_v_BCs_exist_globally",
                                         "specified_u_BCs_exist_globally", "specified_v_BCs_exist_globally",
                                         "Flather_u_BCs_exist_globally", "Flather_v_BCs_exist_globally", "zero_vorticity",
                                         "freeslip_vorticity", "computed_vorticity", "specified_vorticity", "zero_strain", "freeslip_strain",
                                         "computed_strain", "zero_biharmonic")] + \
               [("segment", C.POINTER(ObcSegment)), ("segnum_u", C.c_void_p), ("segnum_v", C.c_void_p),
                ("rx_normal", C.c_void_p), ("ry_normal", C.c_void_p), ("gamma_uv", C.c_double), ("rx_max", C.c_double)] + \
               [(n, C.c_void_p) for n in ("rx_oblique_u", "ry_oblique_u", "cff_normal_u", "rx_oblique_v", "ry_oblique_v", "cff_normal_v")]


class HorDiffStats(C.Structure):
    _fields_ = [("num_itts", C.c_int32), ("halo_updates", C.c_int32), ("max_CFL", C.c_double)]


class HorViscCS(C.Structure):
    """mom6hip_hor_visc_cs_t (include/mom6hip.h)."""
    _fields_ = ([(n, C.c_double) for n in ("Kh", "Kh_bg_min", "Kh_vel_scale", "Smag_Lap_const", "Ah", "Ah_vel_scale", "Ah_time_scale",
                                           "Smag_bi_const", "bound_Cor_vel", "bound_coef")]
                + [("reserved0", C.c_double * 6)]
                + [(n, C.c_int32) for n in ("Laplacian", "biharmonic", "Smagorinsky_Kh", "Smagorinsky_Ah", "bound_Kh", "better_bound_Kh",
                                           "bound_Ah", "better_bound_Ah", "bound_Coriolis", "add_LES_viscosity", "no_slip", "use_land_mask",
                                           "use_cont_thick", "initialized")]
                + [("unsupported", C.c_int32 * 10)]
                + [(n, C.c_void_p) for n in HOR_VISC_ARRAYS_H + HOR_VISC_ARRAYS_Q]
                + [(n, C.c_void_p) for n in ("MEKE_Ku", "MEKE_Au", "MEKE_mom_src")]
                + [("reserved1", C.c_void_p * 1)])


# ---- MOM_thickness_diffuse --------------------------------------------------------------------------------
THICKNESS_DIFFUSE_UNSUPPORTED = ("unused_0", "detangle_interfaces", "Kh_eta", "use_stanley_gm", "MEKE_GEOMETRIC", "GM_src_alt", "read_khth",
                                 "ebt_struct", "Use_KH_in_MEKE", "non_Boussinesq")
THICKNESS_DIFFUSE_FIELDS = ("MEKE_Kh", "L2u", "L2v", "SN_u", "SN_v", "Res_fn_u", "Res_fn_v", "slope_x", "slope_y", "MEKE_GM_src", "Rlay", "cg1", "g_prime", "Depth_fn_u", "Depth_fn_v")


class ThicknessDiffuseCS(C.Structure):
    """mom6hip_thickness_diffuse_cs_t (include/mom6hip.h)."""
    _fields_ = ([(n, C.c_double) for n in ("Khth", "Khth_Min", "Khth_Max", "max_Khth_CFL", "slope_max", "kappa_smooth", "KHTH_Slope_Cff", "KhTh_fac",
                                            "FGNV_scale", "N2_floor")]
                + [("reserved0", C.c_double * 2)]
                + [(n, C.c_int32) for n in ("thickness_diffuse", "use_GM_work_bug", "nkml", "initialized", "use_variable_mixing", "use_FGNV_streamfn")]
                + [("unsupported", C.c_int32 * 10)]
                + [(n, C.c_void_p) for n in THICKNESS_DIFFUSE_FIELDS])


MIXEDLAYER_RESTRAT_UNSUPPORTED = ("use_Bodner", "use_Stanley_ML", "non_Boussinesq")
MIXEDLAYER_RESTRAT_FIELDS = ("MLD_filtered", "MLD_filtered_slow", "Rd_dx_h")


class MixedlayerRestratCS(C.Structure):
    """mom6hip_mixedlayer_restrat_cs_t (include/mom6hip.h)."""
    _fields_ = ([(n, C.c_double) for n in ("ml_restrat_coef", "ml_restrat_coef2", "front_length", "vonKar", "MLE_MLD_decay_time", "MLE_MLD_decay_time2",
                                            "MLE_density_diff", "MLE_tail_dh", "MLE_MLD_stretch", "ustar_min")]
                + [("reserved0", C.c_double * 4)]
                + [(n, C.c_int32) for n in ("MLE_use_PBL_MLD", "nkml", "initialized")]
                + [("reserved_i", C.c_int32 * 1), ("unsupported", C.c_int32 * 8)]
                + [(n, C.c_void_p) for n in MIXEDLAYER_RESTRAT_FIELDS]
                + [("reserved1", C.c_void_p * 3)])


# ---- MOM_dynamics_split_RK2 -----------------------------------------------------------------------------
RK2_ARRAYS_3D = (("CAu", POS_U), ("CAv", POS_V), ("CAu_pred", POS_U), ("CAv_pred", POS_V), ("PFu", POS_U), ("PFv", POS_V),
                 ("diffu", POS_U), ("diffv", POS_V), ("visc_rem_u", POS_U), ("visc_rem_v", POS_V), ("u_accel_bt", POS_U),
                 ("v_accel_bt", POS_V), ("u_av", POS_U), ("v_av", POS_V), ("h_av", POS_H), ("pbce", POS_H))
RK2_ARRAYS_2D = (("eta", POS_H), ("eta_PF", POS_H), ("uhbt", POS_U), ("vhbt", POS_V), ("du_av_inst", POS_U), ("dv_av_inst", POS_V))


class DynSplitRK2CS(C.Structure):
    """mom6hip_dyn_split_rk2_cs_t (include/mom6hip.h)."""
    _fields_ = ([("be", C.c_double), ("begw", C.c_double), ("BT_use_layer_fluxes", C.c_int32), ("store_CAu", C.c_int32),
                 ("CAu_pred_stored", C.c_int32), ("split_bottom_stress", C.c_int32), ("reserved0", C.c_int32 * 4),
                 ("continuity_CSp", C.c_void_p), ("CoriolisAdv", C.c_void_p), ("PressureForce_CSp", C.c_void_p),
                 ("eqn_of_state", C.c_void_p), ("barotropic_CSp", C.c_void_p), ("BT_cont", C.c_void_p), ("hooks", C.c_void_p),
                 ("vertvisc_CSp", C.c_void_p), ("visc", C.c_void_p), ("hor_visc", C.c_void_p)]
                + [(n, C.c_void_p) for n, _ in RK2_ARRAYS_3D] + [(n, C.c_void_p) for n, _ in RK2_ARRAYS_2D]
                + [("set_visc_CSp", C.c_void_p), ("OBC", C.c_void_p), ("p_surf_begin", C.c_void_p), ("p_surf_end", C.c_void_p),
                   ("p_surf", C.c_void_p)])


# ---- MOM_vert_friction ------------------------------------------------------------------------------------
VERTVISC_UNSUPPORTED = ("(free)", "(free 2)", "fixed_LOTW_ML", "apply_LOTW_floor", "use_GL90_in_SSW", "StokesMixing",
                        "non_Boussinesq")
VERTVISC_CS_ARRAYS = (("a_u", POS_U, 1), ("a_v", POS_V, 1), ("h_u", POS_U, 0), ("h_v", POS_V, 0))      # (name, pos, extra interfaces)


class VertviscCS(C.Structure):
    """mom6hip_vertvisc_cs_t (include/mom6hip.h)."""
    _fields_ = ([(n, C.c_double) for n in ("Hmix", "Hmix_stress", "Kvml_invZ2", "Kv", "Hbbl", "Kv_extra_bbl", "harm_BL_val", "maxvel",
                                           "CFL_trunc", "vel_underflow", "H_to_RZ")]
                + [("vonKar", C.c_double), ("dynamic_viscous_ML", C.c_int32), ("nkml", C.c_int32), ("reserved0", C.c_double * 3)]
                + [(n, C.c_int32) for n in ("bottomdraglaw", "harmonic_visc", "direct_stress", "CFL_based_trunc", "answer_date")]
                + [("unsupported", C.c_int32 * 7), ("ntrunc", C.c_int64)]
                + [(n, C.c_void_p) for n in ("a_u", "a_v", "h_u", "h_v")] + [("reserved1", C.c_void_p * 4)])


class VertviscType(C.Structure):
    """mom6hip_vertvisc_type_t (include/mom6hip.h)."""
    _fields_ = ([(n, C.c_void_p) for n in ("Kv_bbl_u", "Kv_bbl_v", "bbl_thick_u", "bbl_thick_v", "Ray_u", "Ray_v", "Kv_shear",
                                           "Kv_shear_Bu", "nkml_visc_u", "nkml_visc_v", "ustar")] + [("reserved", C.c_void_p * 1)])


# ---- z* regridding ----------------------------------------------------------------------------------------
REGRIDDING_ZSTAR = 2


class RegriddingCS(C.Structure):
    """mom6hip_regridding_cs_t (include/mom6hip.h)."""
    _fields_ = [("regridding_scheme", C.c_int32), ("nk", C.c_int32), ("min_thickness", C.c_double), ("old_grid_weight", C.c_double),
                ("depth_of_time_filter_shallow", C.c_double), ("depth_of_time_filter_deep", C.c_double), ("Z_ref", C.c_double),
                ("coordinateResolution", C.c_void_p)]


class EnergySums(C.Structure):
    """mom6hip_energy_sums_t"""
    _fields_ = [("mass_tot", C.c_double), ("KE_tot", C.c_double), ("PE_tot", C.c_double), ("toten", C.c_double), ("Salt", C.c_double),
                ("Heat", C.c_double), ("max_CFL", C.c_double * 2), ("mass_EFP", C.c_int64 * 6), ("salt_EFP", C.c_int64 * 6),
                ("heat_EFP", C.c_int64 * 6), ("npoints", C.c_int64)]
