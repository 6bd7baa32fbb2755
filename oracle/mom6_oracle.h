/*
 * mom6_oracle.h -- CPU restatement of the MOM6 dynamical-core hot path (TEST INFRASTRUCTURE).
 *
 * This directory is the parity oracle, not the product: plain C99, scalar, one thread, written to
 * follow the reference Fortran loop for loop (each function cites the reference file:line it
 * restates).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (mom6_amd/, libmom6hip.so) never links or calls anything in here.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - ALE reconstruction pieces, EOS: pinned by the reference's own known-answer vectors
 *     (src/ALE/MOM_remapping.F90:1339-1780, src/equation_of_state/MOM_EOS.F90:1902-2032).
 *   - advect_tracer: PARITY UNPINNED -- the reference holds no known-answer vectors for it and
 *     src/tracer/MOM_tracer_advect.F90 cannot be built here without stand-ins for FMS-backed modules.
 *     It is checked through the invariants the reference's own test-suite relies on instead
 *     (conservation, x/y rotation equivalence, layout independence).
 *   - CorAdCalc, continuity_PPM, PressureForce_FV assembly, btstep: PARITY UNPINNED for the same reason
 *     (invariants in tests/: energy / enstrophy conservation, transport matching, hydrostatic consistency,
 *     barotropic mass budget).
 *
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction, so the evaluation is the source's explicit
 * parenthesisation in IEEE fp64, the same as the HIP build).
 */
#ifndef MOM6_ORACLE_H
#define MOM6_ORACLE_H

#include "../include/mom6hip.h"   /* the grid POD only; every pointer in it is a HOST pointer here */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- threading: the oracle is scalar code; loops whose iterations are independent (the ones the reference marks
 * !$OMP parallel do, e.g. src/core/MOM_continuity_PPM.F90:615-618, src/core/MOM_CoriolisAdv.F90:281-284,
 * src/core/MOM_barotropic.F90:1869-2422) carry ORC_PAR.  The default build ignores it (one thread);
 * libmom6oracle_omp.so (make omp: -fopenmp) runs them on all cores for bench.py's all-cores CPU baseline.  The
 * results are bit-identical either way (tests/test_oracle_omp.py): no reduction crosses a parallel loop. */
#define ORC_PAR _Pragma("omp parallel for schedule(static)")
#define ORC_PAR_DYN _Pragma("omp parallel for schedule(dynamic, 1)")

/* ---- index helpers: Fortran (i,j,k) with k 1-based, symmetric memory --------------------- */
#define ORC_NIH(G)  ((G)->ied - (G)->isd + 1)
#define ORC_NJH(G)  ((G)->jed - (G)->jsd + 1)
#define ORC_H2(G,i,j)   ((long)((i)-(G)->isd)     + (long)ORC_NIH(G)    *((j)-(G)->jsd))
#define ORC_U2(G,I,j)   ((long)((I)-(G)->isd+1)   + (long)(ORC_NIH(G)+1)*((j)-(G)->jsd))
#define ORC_V2(G,i,J)   ((long)((i)-(G)->isd)     + (long)ORC_NIH(G)    *((J)-(G)->jsd+1))
#define ORC_Q2(G,I,J)   ((long)((I)-(G)->isd+1)   + (long)(ORC_NIH(G)+1)*((J)-(G)->jsd+1))
#define ORC_H3(G,i,j,k) (ORC_H2(G,i,j) + (long)ORC_NIH(G)    *ORC_NJH(G)    *((k)-1))
#define ORC_U3(G,I,j,k) (ORC_U2(G,I,j) + (long)(ORC_NIH(G)+1)*ORC_NJH(G)    *((k)-1))
#define ORC_V3(G,i,J,k) (ORC_V2(G,i,J) + (long)ORC_NIH(G)    *(ORC_NJH(G)+1)*((k)-1))
#define ORC_Q3(G,I,J,k) (ORC_Q2(G,I,J) + (long)(ORC_NIH(G)+1)*(ORC_NJH(G)+1)*((k)-1))

/* ---- MOM_domains, one tile --------------------------------------------------------------- */
/* pass_var / pass_vector / do_group_pass on a single-PE domain
 * (config_src/infra/FMS2/MOM_domain_infra.F90:171,660,1141): re-entrant edges are filled from the
 * tile's own compute domain (corners included), closed edges are left untouched.
 * pos: MOM6HIP_POS_H/U/V/Q.  nk = 1 for a 2-D field. */
void orc_halo_update(const mom6hip_grid_t *G, double *f, int pos, int nk);
int orc_tracer_hordiff_varmix(const mom6hip_grid_t *G, const mom6hip_tracer_hor_diff_cs_t *CS, const mom6hip_hordiff_fields_t *F,
                              const double *h, double dt, double *const *tr, const double *conc_underflow, int ntr,
                              mom6hip_hordiff_stats_t *stats);
int orc_tracer_hordiff(const mom6hip_grid_t *G, const mom6hip_tracer_hor_diff_cs_t *CS, const double *h, double dt,
                       double *const *tr, const double *conc_underflow, int ntr, mom6hip_hordiff_stats_t *stats);
/* tracer_hordiff with CS%Diffuse_ML_interior (CS->unsupported[2]) and tracer_epipycnal_ML_diff */
int orc_tracer_hordiff_epipycnal(const mom6hip_grid_t *G, const mom6hip_tracer_hor_diff_cs_t *CS, const mom6hip_epipycnal_cs_t *EP,
                                 const mom6hip_hordiff_fields_t *F, const double *h, const mom6hip_eos_t *eos, double dt,
                                 double *const *tr, const double *conc_underflow, int ntr, int idx_T, int idx_S,
                                 mom6hip_hordiff_stats_t *stats);
/* tracer_hordiff with CS%use_neutral_diffusion (CS->unsupported[0]); tr[idx_T], tr[idx_S] are tv%T, tv%S */
int orc_tracer_hordiff_neutral(const mom6hip_grid_t *G, const mom6hip_tracer_hor_diff_cs_t *CS,
                               const mom6hip_neutral_diffusion_cs_t *ND, const mom6hip_hordiff_fields_t *F, const double *h,
                               const mom6hip_eos_t *eos, const double *p_surf, double dt, double *const *tr,
                               const double *conc_underflow, int ntr, int idx_T, int idx_S, mom6hip_hordiff_stats_t *stats);
/* MOM_neutral_diffusion column routines (oracle/neutral_diffusion.c) */
double orc_ndiff_fv_diff(double hkm1, double hk, double hkp1, double Skm1, double Sk, double Skp1);
double orc_ndiff_fvlsq_slope(double hkm1, double hk, double hkp1, double Skm1, double Sk, double Skp1);
void orc_ndiff_interface_scalar(int nk, const double *h, const double *S, double *Si, int i_method, double h_neglect);
double orc_ndiff_interpolate_for_nondim_position(double dRhoNeg, double Pneg, double dRhoPos, double Ppos);
void orc_ndiff_find_neutral_surface_positions_continuous(int nk, const double *Pl, const double *Tl, const double *Sl,
    const double *dRdTl, const double *dRdSl, const double *Pr, const double *Tr, const double *Sr, const double *dRdTr,
    const double *dRdSr, double *PoL, double *PoR, int *KoL, int *KoR, double *hEff);
void orc_ndiff_boundary_k_range_surface(int nk, const double *h, double hbl, int *k_bot, double *zeta_bot);
int orc_ndiff_neutral_surface_flux(int nk, const double *hl, const double *hr, const double *Tl, const double *Tr,
                                   const double *PiL, const double *PiR, const int *KoL, const int *KoR, const double *hEff,
                                   double *Flx, double h_neglect);
int orc_neutral_branch(const mom6hip_grid_t *G, const mom6hip_neutral_diffusion_cs_t *ND, const mom6hip_eos_t *eos,
                       const double *h, const double *p_surf, const double *h_ML, const double *khdt_x, const double *khdt_y, int num_itts,
                       double I_numitts, double *const *tr, const double *conc_underflow, int ntr, int idx_T, int idx_S,
                       int *halo_updates);

/* ---- MOM_tracer_advect -------------------------------------------------------------------- */
/* advect_tracer, src/tracer/MOM_tracer_advect.F90:52-324 (OBC not associated). */
int orc_advect_tracer(const mom6hip_grid_t *G, const double *h_end, const double *uhtr,
                      const double *vhtr, double dt, const mom6hip_tracer_advect_cs_t *cs,
                      double *const *tr, const double *conc_underflow, int ntr,
                      int x_first_in, double *vol_prev, int max_iter_in, int update_vol_prev,
                      double *uhr_out, double *vhr_out, mom6hip_advect_stats_t *stats);
/* advect_tracer with OBC associated (the segments' tracer registries: MOM_tracer_advect.F90:441-477, :580-627, :823-861, :965-1014) */
int orc_advect_tracer_obc(const mom6hip_grid_t *G, const double *h_end, const double *uhtr,
                          const double *vhtr, double dt, const mom6hip_tracer_advect_cs_t *cs,
                          double *const *tr, const double *conc_underflow, int ntr,
                          int x_first_in, double *vol_prev, int max_iter_in, int update_vol_prev,
                          double *uhr_out, double *vhr_out, mom6hip_advect_stats_t *stats, const mom6hip_obc_t *OBC);

/* ---- ALE reconstruction + remapping (oracle/remapping.c) ------------------------------------ */
/* REMAPPING_* and INTEGRATION_* of src/ALE/MOM_remapping.F90:50-64 */
#define ORC_REMAP_PCM     0
#define ORC_REMAP_PLM     2
#define ORC_REMAP_PLM_HYBGEN 3
#define ORC_REMAP_PPM_H4  4
#define ORC_REMAP_PPM_HYBGEN 6
#define ORC_REMAP_WENO_HYBGEN 7
#define ORC_REMAP_PPM_IH4 5
#define ORC_REMAP_PPM_CW  10
#define ORC_REMAP_PQM_IH4IH3 8
#define ORC_REMAP_PQM_IH6IH5 9
#define ORC_INT_PCM 0
#define ORC_INT_PLM 1
#define ORC_INT_PPM 3
#define ORC_INT_PQM 5
/* E(k,1:2) and coef(k,1:3) are Fortran-ordered: E[side*n + k], coef[d*n + k]. */
void orc_pcm_reconstruction(int n, const double *u, double *E, double *coef);
double orc_plm_slope_wa(double h_l, double h_c, double h_r, double h_neglect, double u_l, double u_c, double u_r);
double orc_plm_monotonized_slope(double u_l, double u_c, double u_r, double s_l, double s_c, double s_r);
double orc_plm_extrapolate_slope(double h_l, double h_c, double h_neglect, double u_l, double u_c);
void orc_plm_reconstruction(int n, const double *h, const double *u, double *E, double *coef, double h_neglect);
void orc_plm_boundary_extrapolation(int n, const double *h, const double *u, double *E, double *coef, double h_neglect);
void orc_hybgen_plm_coefs(int nk, const double *si, const double *dpi, double *slope, double thin);
void orc_hybgen_ppm_coefs(int nk, const double *s, const double *h_src, double *E, double thin);
void orc_hybgen_weno_coefs(int nk, const double *s, const double *h_src, double *E, double thin);
void orc_bound_edge_values(int n, const double *h, const double *u, double *E);
void orc_check_discontinuous_edge_values(int n, const double *u, double *E);
void orc_end_value_h4(const double dz[4], const double u[4], double Csys[4]);
void orc_edge_values_explicit_h4(int n, const double *h, const double *u, double *E, double h_neglect);
void orc_edge_values_implicit_h4(int n, const double *h, const double *u, double *E, double h_neglect);
void orc_edge_values_explicit_h4cw(int n, const double *h, const double *u, double *E, double h_neglect);
void orc_ppm_monotonicity(int n, const double *u, double *E);
void orc_solve_diag_dominant_tridiag(const double *Al, const double *Ac, const double *Au, const double *R, double *X, int n);
void orc_ppm_limiter_standard(int n, const double *h, const double *u, double *E);
void orc_ppm_reconstruction(int n, const double *h, const double *u, double *E, double *coef);
void orc_ppm_boundary_extrapolation(int n, const double *h, const double *u, double *E, double *coef, double h_neglect);
double orc_average_value_ppoly(int n, const double *u0, const double *E, const double *coef, int method,
                               int i0, double xa, double xb);
void orc_remap_via_sub_cells(int n0, const double *h0, const double *u0, const double *E, const double *coef,
                             int n1, const double *h1, int method, int force_bounds_in_subcell,
                             double *u1, double *uh_err);
void orc_edge_slopes_implicit_h3(int n, const double *h, const double *u, double *S, double h_neglect);
int orc_linear_solver6(double A[6][6], double R[6], double X[6]);
int orc_edge_slopes_implicit_h5(int n, const double *h, const double *u, double *S, double h_neglect);
int orc_edge_values_implicit_h6(int n, const double *h, const double *u, double *E, double h_neglect_edge);
void orc_pqm_limiter(int n, const double *h, const double *u, double *E, double *S, double h_neglect);
void orc_pqm_reconstruction(int n, const double *h, const double *u, double *E, double *S, double *coef, double h_neglect);
void orc_pqm_boundary_extrapolation_v1(int n, const double *h, const double *u, double *E, double *S, double *coef, double h_neglect);
int orc_build_reconstructions_1d(int scheme, int boundary_extrapolation, int n0, const double *h0, const double *u0,
                                 double *coef, double *E, double h_neglect, double h_neglect_edge);
int orc_remapping_core_h(int scheme, int boundary_extrapolation, int n0, const double *h0, const double *u0,
                         int n1, const double *h1, double *u1, double h_neglect, double h_neglect_edge);
int orc_remapping_core_w(int scheme, int boundary_extrapolation, int n0, const double *h0, const double *u0,
                         int n1, const double *dx, double *u1, double h_neglect, double h_neglect_edge);
void orc_dz_from_h1h2(int n1, const double *h1, int n2, const double *h2, double *dx);
/* ALE_remap_tracers, src/ALE/MOM_ALE.F90:737-867 (no PCM_cell, no tendency diagnostics) */
int orc_ale_remap_tracers(const mom6hip_grid_t *G, const mom6hip_remapping_cs_t *cs, const double *h_old,
                          const double *h_new, double *const *tr, const double *conc_underflow, int ntr);

/* ---- z* regridding and velocity remapping (oracle/regridding.c) ------------------------------------------ */
int orc_ale_regrid(const mom6hip_grid_t *G, const mom6hip_regridding_cs_t *CS, const double *h, double *h_new, double *dzRegrid);
int orc_ale_remap_set_h_vel(const mom6hip_grid_t *G, const double *h_new, double *h_u, double *h_v);
int orc_ale_remap_set_h_vel_via_dz(const mom6hip_grid_t *G, const double *h_old, const double *dzInterface, double *h_u, double *h_v);
int orc_ale_remap_velocities(const mom6hip_grid_t *G, const mom6hip_remapping_cs_t *cs, const double *h_old_u, const double *h_old_v,
                             const double *h_new_u, const double *h_new_v, double *u, double *v);

/* ---- MOM_CoriolisAdv (oracle/coriolis_adv.c) ------------------------------------------------ */
int orc_coradcalc(const mom6hip_grid_t *G, const mom6hip_coriolisadv_cs_t *CS, const double *u, const double *v,
                  const double *h, const double *uh, const double *vh, double *CAu, double *CAv);
/* radiation_open_bdry_conds (the normal component), open_boundary_apply_normal_flow and the halo update of u_new, v_new; open_boundary_zero_normal_flow */
int orc_radiation_open_bdry_conds(const mom6hip_grid_t *G, const mom6hip_obc_t *OBC, double gamma_uv, double rx_max, double *rx_normal,
                                  double *ry_normal, double *u_new, const double *u_old, double *v_new, const double *v_old, double dt);
int orc_open_boundary_zero_normal_flow(const mom6hip_grid_t *G, const mom6hip_obc_t *OBC, double *u, double *v);
int orc_update_segment_tracer_reservoirs(const mom6hip_grid_t *G, const double *uhr, const double *vhr, const double *h,
                                         const mom6hip_obc_t *OBC, double dt, const double *const *tr, int ntr);
/* CorAdCalc with OBC associated (OBC may be NULL: the call above) */
int orc_coradcalc_obc(const mom6hip_grid_t *G, const mom6hip_coriolisadv_cs_t *CS, const mom6hip_obc_t *OBC, const double *u,
                      const double *v, const double *h, const double *uh, const double *vh, double *CAu, double *CAv);

/* ---- MOM_continuity_PPM (oracle/continuity.c) ---------------------------------------------- */
int orc_continuity(const mom6hip_grid_t *G, const mom6hip_continuity_cs_t *CS, const double *u, const double *v,
                   const double *hin, double *h, double *uh, double *vh, double dt, const double *uhbt,
                   const double *vhbt, const double *visc_rem_u, const double *visc_rem_v, double *u_cor,
                   double *v_cor, const mom6hip_bt_cont_t *BT_cont, double *du_cor, double *dv_cor);
/* continuity_PPM with OBC associated (obc may be NULL: the call above) */
int orc_continuity_obc(const mom6hip_grid_t *G, const mom6hip_continuity_cs_t *CS, const mom6hip_obc_t *OBC, const double *u,
                       const double *v, const double *hin, double *h, double *uh, double *vh, double dt, const double *uhbt,
                       const double *vhbt, const double *visc_rem_u, const double *visc_rem_v, double *u_cor, double *v_cor,
                       const mom6hip_bt_cont_t *BT_cont, double *du_cor, double *dv_cor);

/* ---- MOM_EOS / MOM_PressureForce_FV (oracle/pressure_force.c) ------------------------------ */
double orc_eos_density(const mom6hip_eos_t *E, double T, double S, double p);
double orc_eos_density_anomaly(const mom6hip_eos_t *E, double T, double S, double p, double rho_ref);
void orc_eos_density_derivs(const mom6hip_eos_t *E, double T, double S, double p, double *dT, double *dS);
void orc_ale_plm_edge_values(const mom6hip_grid_t *G, const double *h, const double *Q, int bdry_extrap,
                             double *Q_t, double *Q_b);
int orc_pressureforce_fv_bouss(const mom6hip_grid_t *G, const mom6hip_pressureforce_cs_t *CS, const mom6hip_eos_t *EOS,
                               const double *h, const double *T, const double *S, const double *p_atm,
                               double *PFu, double *PFv, double *pbce, double *eta);
/* non-Boussinesq: calculate_spec_vol with spv_ref, and PressureForce_FV_nonBouss (MOM_PressureForce_FV.F90:89) + Set_pbce_nonBouss */
double orc_eos_spec_vol_anomaly(const mom6hip_eos_t *E, double T, double S, double p, double spv_ref);
int orc_pressureforce_fv_nonbouss(const mom6hip_grid_t *G, const mom6hip_pressureforce_cs_t *CS, const mom6hip_eos_t *EOS,
                                  const double *h, const double *T, const double *S, const double *p_atm, double H_to_RZ,
                                  double *PFu, double *PFv, double *pbce, double *eta);

/* ---- MOM_barotropic (oracle/barotropic.c) ----------------------------------------------------- */
int orc_thickness_diffuse(const mom6hip_grid_t *G, const mom6hip_thickness_diffuse_cs_t *CS, double *h, double *uhtr, double *vhtr,
                          const double *T, const double *S, const mom6hip_eos_t *EOS, double dt, double *uhGM, double *vhGM);
int orc_mixedlayer_restrat(const mom6hip_grid_t *G, const mom6hip_mixedlayer_restrat_cs_t *CS, double *h, double *uhtr, double *vhtr,
                           const double *T, const double *S, const mom6hip_eos_t *EOS, const double *ustar, double dt, const double *h_MLD,
                           double *uhml, double *vhml);
double orc_mle_mu(double sigma, double dh);
double orc_cr_exp(double t);             /* correctly rounded exp(t), t <= 0 (0 below -700) */
double orc_cr_pow(double x, double y);   /* correctly rounded x**y, 0 < x <= 1, 0 < y <= 1 */
double orc_cr_cos(double x);             /* correctly rounded cos(x), |x| <= pi (cr_trig.c) */
double orc_cr_acos(double x);            /* correctly rounded acos(x), |x| <= 1 (cr_trig.c) */
int orc_barotropic_init(const mom6hip_grid_t *G, mom6hip_barotropic_cs_t *CS);
int orc_btcalc(const mom6hip_grid_t *G, mom6hip_barotropic_cs_t *CS, const double *h, const double *h_u,
               const double *h_v, int may_use_default);
/* btcalc with OBC associated (:3610-3664) */
int orc_btcalc_obc(const mom6hip_grid_t *G, mom6hip_barotropic_cs_t *CS, const double *h, const double *h_u,
                   const double *h_v, int may_use_default, const mom6hip_obc_t *OBC);
int orc_bt_mass_source(const mom6hip_grid_t *G, mom6hip_barotropic_cs_t *CS, const double *h, const double *eta, int set_cor);
int orc_set_dtbt_eta(const mom6hip_grid_t *G, mom6hip_barotropic_cs_t *CS, const double *eta, const double *pbce,
                     const mom6hip_bt_cont_t *BT_cont, double gtot_est, double SSH_add);
int orc_set_dtbt(const mom6hip_grid_t *G, mom6hip_barotropic_cs_t *CS, const double *pbce, const mom6hip_bt_cont_t *BT_cont,
                 double gtot_est, double SSH_add);
int orc_btstep(const mom6hip_grid_t *G, mom6hip_barotropic_cs_t *CS, const double *U_in, const double *V_in,
               const double *eta_in, double dt, const double *bc_accel_u, const double *bc_accel_v, const double *taux,
               const double *tauy, double RZ_to_H, const double *pbce, const double *eta_PF_in, const double *U_Cor,
               const double *V_Cor, double *accel_layer_u, double *accel_layer_v, double *eta_out, double *uhbtav,
               double *vhbtav, const double *visc_rem_u, const double *visc_rem_v, const mom6hip_bt_cont_t *BT_cont,
               const double *eta_PF_start, const double *taux_bot, const double *tauy_bot, const double *uh0,
               const double *vh0, const double *u_uh0, const double *v_vh0, double *etaav);
/* btstep with OBC associated (Flather, gradient and specified segments; :1089-1110, set_up_BT_OBC :3172, apply_velocity_OBCs :2931, ...) */
int orc_btstep_obc(const mom6hip_grid_t *G, mom6hip_barotropic_cs_t *CS, const double *U_in, const double *V_in,
               const double *eta_in, double dt, const double *bc_accel_u, const double *bc_accel_v, const double *taux,
               const double *tauy, double RZ_to_H, const double *pbce, const double *eta_PF_in, const double *U_Cor,
               const double *V_Cor, double *accel_layer_u, double *accel_layer_v, double *eta_out, double *uhbtav,
               double *vhbtav, const double *visc_rem_u, const double *visc_rem_v, const mom6hip_bt_cont_t *BT_cont,
               const double *eta_PF_start, const double *taux_bot, const double *tauy_bot, const double *uh0,
               const double *vh0, const double *u_uh0, const double *v_vh0, double *etaav, const mom6hip_obc_t *OBC);

/* ---- MOM_dynamics_split_RK2 (oracle/dyn_split_rk2.c); every pointer in CS is a HOST pointer ------------------- */
/* ---- MOM_vert_friction (oracle/vert_friction.c) ---- */
int orc_vertvisc_coef(const mom6hip_grid_t *G, mom6hip_vertvisc_cs_t *CS, const double *u, const double *v, const double *h,
                      const double *dz, const mom6hip_vertvisc_type_t *visc, double dt);
int orc_vertvisc(const mom6hip_grid_t *G, mom6hip_vertvisc_cs_t *CS, double *u, double *v, const double *h, const double *taux,
                 const double *tauy, const mom6hip_vertvisc_type_t *visc, double dt, double *taux_bot, double *tauy_bot);
/* the two above with OBC associated (MOM_vert_friction.F90:1335-1355, :1546-1566, :1901-1925, :2061-2110; :988-1006) */
int orc_vertvisc_coef_obc(const mom6hip_grid_t *G, mom6hip_vertvisc_cs_t *CS, const double *u, const double *v, const double *h,
                          const double *dz, const mom6hip_vertvisc_type_t *visc, double dt, const mom6hip_obc_t *OBC);
int orc_vertvisc_obc(const mom6hip_grid_t *G, mom6hip_vertvisc_cs_t *CS, double *u, double *v, const double *h, const double *taux,
                     const double *tauy, const mom6hip_vertvisc_type_t *visc, double dt, double *taux_bot, double *tauy_bot,
                     const mom6hip_obc_t *OBC);
int orc_vertvisc_remnant(const mom6hip_grid_t *G, const mom6hip_vertvisc_cs_t *CS, const mom6hip_vertvisc_type_t *visc,
                         double *visc_rem_u, double *visc_rem_v, double dt);

/* ---- MOM_set_viscosity (oracle/set_viscosity.c) ---------------------------------------------------------------------- */
int orc_set_viscous_BBL(const mom6hip_grid_t *G, const mom6hip_set_visc_cs_t *CS, const double *u, const double *v, const double *h,
                        const double *T, const double *S, const mom6hip_eos_t *EOS, const mom6hip_vertvisc_type_t *visc);
/* set_viscous_BBL with CS%OBC associated (:374-413, :502-580, set_v_at_u / set_u_at_v :1829-1838, :1874-1883) */
int orc_set_viscous_BBL_obc(const mom6hip_grid_t *G, const mom6hip_set_visc_cs_t *CS, const double *u, const double *v, const double *h,
                            const double *T, const double *S, const mom6hip_eos_t *EOS, const mom6hip_vertvisc_type_t *visc,
                            const mom6hip_obc_t *OBC);
int orc_set_viscous_ML(const mom6hip_grid_t *G, const mom6hip_set_visc_cs_t *CS, const double *u, const double *v, const double *h,
                       const double *T, const double *S, const mom6hip_eos_t *EOS, const double *taux, const double *tauy,
                       const mom6hip_vertvisc_type_t *visc, double dt);

/* ---- MOM_hor_visc (oracle/hor_visc.c) ------------------------------------------------------------------------------- */
/* hor_visc_init (the static arrays, :2440-2760) and horizontal_viscosity (:245-1979); all arrays HOST arrays */
int orc_hor_visc_init(const mom6hip_grid_t *G, mom6hip_hor_visc_cs_t *CS, double dt);
int orc_horizontal_viscosity(const mom6hip_grid_t *G, const mom6hip_hor_visc_cs_t *CS, const double *u, const double *v,
                             const double *h, double *diffu, double *diffv, double dt, const double *hu_cont,
                             const double *hv_cont);
/* horizontal_viscosity with OBC associated (:733-849, :889-903, :1388-1409, :1751-1782) */
int orc_horizontal_viscosity_obc(const mom6hip_grid_t *G, const mom6hip_hor_visc_cs_t *CS, const double *u, const double *v,
                                 const double *h, double *diffu, double *diffv, double dt, const double *hu_cont,
                                 const double *hv_cont, const mom6hip_obc_t *OBC);

int orc_dyn_split_rk2_init(const mom6hip_grid_t *G, mom6hip_dyn_split_rk2_cs_t *CS, const double *u, const double *v,
                           const double *h, double *uh, double *vh, double dt);
int orc_step_dyn_split_rk2(const mom6hip_grid_t *G, mom6hip_dyn_split_rk2_cs_t *CS, double *u_inst, double *v_inst, double *h,
                           const double *T, const double *S, double dt, const double *taux, const double *tauy, double RZ_to_H,
                           double *uh, double *vh, double *uhtr, double *vhtr, double *eta_av, int calc_dtbt);
/* SPLIT_RK2B: MOM_dynamics_split_RK2b.F90 */
int orc_dyn_split_rk2b_init(const mom6hip_grid_t *G, mom6hip_dyn_split_rk2_cs_t *CS, const double *h);
int orc_step_dyn_split_rk2b(const mom6hip_grid_t *G, mom6hip_dyn_split_rk2_cs_t *CS, double *u_av, double *v_av, double *h,
                            const double *T, const double *S, double dt, const double *taux, const double *tauy, double RZ_to_H,
                            double *uh, double *vh, double *uhtr, double *vhtr, double *eta_av, int calc_dtbt);

/* ---- MOM_coms (oracle/coms.c) -------------------------------------------------------------------------------------- */
/* reproducing_sum_3d :318 on one PE over points i0..i1, rows j0..j1 (0-based) of a (ke, ncol, nrow) array; regularize_ints
 * :643 and ints_to_real :545 on six EFP integers */
int orc_reproducing_sum_3d(const double *a, int nrow, int ncol, int ke, int i0, int i1, int j0, int j1, double *sum,
                           double *lay_sums, int64_t *efp_sum, int64_t *efp_lay, int *err);
void orc_efp_regularize(int64_t *int_sum);
double orc_efp_to_real(const int64_t *ints);

/* ---- MOM_sum_output (oracle/sum_output.c): the global integrals of write_energy :490-760 */
int orc_write_energy_sums(const mom6hip_grid_t *G, const double *u, const double *v, const double *h, const double *T, const double *S,
                          double dt, double C_p, double H_to_kg_m2, double *mass_lay, double *KE_lay, mom6hip_energy_sums_t *out);

int orc_depth_list_create(int mls, const double *Dlist_in, const double *Area_in, double min_depth_inc, double **depth_out,
                          double **area_out, double **vol_below_out);
void orc_ape_reference_heights(int nz, int listsize, const double *DL_depth, const double *DL_area, const double *DL_vol_below,
                               const double *vol_lay, int *lH, double *Z_0APE);
int orc_write_energy_ape(const mom6hip_grid_t *G, const double *h, const double *mass_lay, const double *g_prime, double Rho0,
                         double H_to_kg_m2, double Z_ref, double min_depth_inc, int *lH_io, double *PE, double *PE_tot, double *Z_0APE);

#ifdef __cplusplus
}
#endif
#endif
