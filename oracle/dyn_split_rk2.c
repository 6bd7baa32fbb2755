/*
 * dyn_split_rk2.c -- CPU restatement of MOM_dynamics_split_RK2 (TEST INFRASTRUCTURE, see mom6_oracle.h).
 *
 * step_MOM_dyn_split_RK2, src/core/MOM_dynamics_split_RK2.F90:289-1176, and the state-setting part of
 * initialize_dyn_split_RK2 (:1521-1622), for the branch the library provides: no OBC, no p_surf, no Stokes PGF,
 * no FPMIX, BEGW = 0, and the parameterisations that SURVEY.md 8f lists as "next" (vertvisc*, set_viscous_ML,
 * horizontal_viscosity) absent: visc_rem = 1, diffu = diffv = 0 -- the reference's step with zero viscosities.
 * Every pointer in the control structure is a HOST pointer here.
 *
 * PARITY UNPINNED: assembled from the unpinned pieces (continuity, CorAdCalc, PressureForce, btstep); checked
 * through invariants in tests/test_dyn_split_rk2.py (volume conservation, rest state, eta consistency).
 */
#include <stdlib.h>
#include <string.h>

#include "mom6_oracle.h"

#define H2(i, j) ORC_H2(G, i, j)
#define U3(i, j, k) ORC_U3(G, i, j, k)
#define V3(i, j, k) ORC_V3(G, i, j, k)
#define H3(i, j, k) ORC_H3(G, i, j, k)

static long n_h3(const mom6hip_grid_t *G) { return (long)ORC_NIH(G) * ORC_NJH(G) * G->nk; }
static long n_u3(const mom6hip_grid_t *G) { return (long)(ORC_NIH(G) + 1) * ORC_NJH(G) * G->nk; }
static long n_v3(const mom6hip_grid_t *G) { return (long)ORC_NIH(G) * (ORC_NJH(G) + 1) * G->nk; }

static void pass3(const mom6hip_grid_t *G, double *f, int pos) { orc_halo_update(G, f, pos, G->nk); }

int orc_dyn_split_rk2_init(const mom6hip_grid_t *G, mom6hip_dyn_split_rk2_cs_t *CS, const double *u, const double *v,
                           const double *h, double *uh, double *vh, double dt) {
  const int is = G->isc, ie = G->iec, js = G->jsc, je = G->jec, nz = G->nk;
  if (CS->begw != 0.0 || CS->split_bottom_stress || CS->hooks || (CS->vertvisc_CSp && !CS->visc)) return 1;
  const mom6hip_obc_t *OBC = CS->OBC;      /* :1516-1519 */
  /* eta :1521-1535 */
  for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) CS->eta[H2(i, j)] = -G->Z_to_H * G->bathyT[H2(i, j)];
  for (int k = 1; k <= nz; k++) for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++)
    CS->eta[H2(i, j)] = CS->eta[H2(i, j)] + h[H3(i, j, k)];
  memset(CS->diffu, 0, sizeof(double) * n_u3(G)); memset(CS->diffv, 0, sizeof(double) * n_v3(G));
  if (CS->hor_visc) {   /* :1543-1550 */
    int rc = orc_horizontal_viscosity_obc(G, CS->hor_visc, u, v, h, CS->diffu, CS->diffv, dt, CS->BT_cont ? CS->BT_cont->h_u : NULL,
                                          CS->BT_cont ? CS->BT_cont->h_v : NULL, OBC);
    if (rc) return rc;
  }
  for (long n = 0; n < n_u3(G); n++) CS->visc_rem_u[n] = 1.0;
  for (long n = 0; n < n_v3(G); n++) CS->visc_rem_v[n] = 1.0;
  memcpy(CS->u_av, u, sizeof(double) * n_u3(G)); memcpy(CS->v_av, v, sizeof(double) * n_v3(G));   /* :1552-1558 */
  if (CS->store_CAu) {   /* :1560-1588 */
    double *h_tmp = (double *)malloc(sizeof(double) * n_h3(G));
    memcpy(h_tmp, h, sizeof(double) * n_h3(G));
    int rc = orc_continuity_obc(G, CS->continuity_CSp, OBC, CS->u_av, CS->v_av, h, h_tmp, uh, vh, dt, NULL, NULL, NULL, NULL, NULL, NULL,
                            NULL, NULL, NULL);
    if (rc) { free(h_tmp); return rc; }
    pass3(G, h_tmp, MOM6HIP_POS_H);
    for (long n = 0; n < n_h3(G); n++) CS->h_av[n] = 0.5 * (h[n] + h_tmp[n]);
    free(h_tmp);
    pass3(G, CS->u_av, MOM6HIP_POS_U); pass3(G, CS->v_av, MOM6HIP_POS_V); pass3(G, uh, MOM6HIP_POS_U); pass3(G, vh, MOM6HIP_POS_V);
    rc = orc_coradcalc_obc(G, CS->CoriolisAdv, OBC, CS->u_av, CS->v_av, CS->h_av, uh, vh, CS->CAu_pred, CS->CAv_pred);
    if (rc) return rc;
    CS->CAu_pred_stored = 1;
  } else {
    double *h_tmp = (double *)malloc(sizeof(double) * n_h3(G));
    memcpy(h_tmp, h, sizeof(double) * n_h3(G));
    int rc = orc_continuity_obc(G, CS->continuity_CSp, OBC, u, v, h, h_tmp, uh, vh, dt, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL);
    if (rc) { free(h_tmp); return rc; }
    pass3(G, h_tmp, MOM6HIP_POS_H);
    for (long n = 0; n < n_h3(G); n++) CS->h_av[n] = 0.5 * (h[n] + h_tmp[n]);
    free(h_tmp);
    CS->CAu_pred_stored = 0;
  }
  /* pass_av_h_uvh :1615-1622 */
  pass3(G, CS->u_av, MOM6HIP_POS_U); pass3(G, CS->v_av, MOM6HIP_POS_V);
  if (CS->CAu_pred_stored) { pass3(G, CS->CAu_pred, MOM6HIP_POS_U); pass3(G, CS->CAv_pred, MOM6HIP_POS_V); }
  else { pass3(G, CS->h_av, MOM6HIP_POS_H); pass3(G, uh, MOM6HIP_POS_U); pass3(G, vh, MOM6HIP_POS_V); }
  return 0;
}

int orc_step_dyn_split_rk2(const mom6hip_grid_t *G, mom6hip_dyn_split_rk2_cs_t *CS, double *u_inst, double *v_inst, double *h,
                           const double *T, const double *S, double dt, const double *taux, const double *tauy, double RZ_to_H,
                           double *uh, double *vh, double *uhtr, double *vhtr, double *eta_av, int calc_dtbt) {
  const int is = G->isc, ie = G->iec, js = G->jsc, je = G->jec, nz = G->nk;
  const int Isq = is - 1, Ieq = ie, Jsq = js - 1, Jeq = je;
  if (CS->begw != 0.0 || CS->split_bottom_stress || CS->hooks || (CS->vertvisc_CSp && !CS->visc)) return 1;
  mom6hip_barotropic_cs_t *BT = CS->barotropic_CSp;
  const mom6hip_bt_cont_t *BTC = CS->BT_cont;
  const int BT_cont_BT_thick = BTC && BTC->h_u && BTC->h_v;
  const long NU = n_u3(G), NV = n_v3(G), NH = n_h3(G);
  double *up = (double *)calloc(NU, sizeof(double)), *vp = (double *)calloc(NV, sizeof(double));       /* :419-421 */
  double *hp = (double *)malloc(sizeof(double) * NH);
  double *u_bc_accel = (double *)calloc(NU, sizeof(double)), *v_bc_accel = (double *)calloc(NV, sizeof(double));
  double *uh_in = (double *)calloc(NU, sizeof(double)), *vh_in = (double *)calloc(NV, sizeof(double));
  double *eta_pred = (double *)calloc((size_t)ORC_NIH(G) * ORC_NJH(G), sizeof(double));
  memcpy(hp, h, sizeof(double) * NH);                                                                   /* :422 */
  double *u_av = CS->u_av, *v_av = CS->v_av, *h_av = CS->h_av, *eta = CS->eta;
  int rc = 0;
#define CHECK(call) do { rc = (call); if (rc) goto done; } while (0)
  /* CS%OBC: the starting velocities of the radiation conditions (:444-456) */
  const mom6hip_obc_t *OBC = CS->OBC;
  /* the surface pressure of the step :435-442 (RK2b :422-429) */
  const int dyn_p_surf = CS->p_surf_begin && CS->p_surf_end;
  const double *p_surf = dyn_p_surf ? CS->p_surf_end : CS->p_surf;
  double *eta_PF_start = dyn_p_surf ? (double *)calloc((size_t)ORC_NIH(G) * ORC_NJH(G), sizeof(double)) : NULL;
  double *u_old_rad_OBC = NULL, *v_old_rad_OBC = NULL;
  if (OBC) {
    u_old_rad_OBC = (double *)malloc(sizeof(double) * NU); v_old_rad_OBC = (double *)malloc(sizeof(double) * NV);
    memcpy(u_old_rad_OBC, u_av, sizeof(double) * NU); memcpy(v_old_rad_OBC, v_av, sizeof(double) * NV);
  }

  /* PressureForce :495 */
  CHECK(orc_pressureforce_fv_bouss(G, CS->PressureForce_CSp, CS->eqn_of_state, h, T, S, p_surf, CS->PFu, CS->PFv, CS->pbce, CS->eta_PF));
  if (dyn_p_surf) {      /* :497-503 / RK2b :500-506 */
    const double pres_to_eta = 1.0 / (G->g_Earth * (G->Rho0 * G->H_to_Z));      /* 1 / (GV%g_Earth * GV%H_to_RZ), Boussinesq */
    for (int j = Jsq; j <= Jeq+1; j++) for (int i = Isq; i <= Ieq+1; i++)
      eta_PF_start[ORC_H2(G,i,j)] = CS->eta_PF[ORC_H2(G,i,j)] - pres_to_eta * (CS->p_surf_begin[ORC_H2(G,i,j)] - CS->p_surf_end[ORC_H2(G,i,j)]);
  }
  if (!CS->CAu_pred_stored)   /* :544-552 */
    CHECK(orc_coradcalc_obc(G, CS->CoriolisAdv, OBC, u_av, v_av, h_av, uh, vh, CS->CAu_pred, CS->CAv_pred));
  /* u_bc_accel :557-564 */
  ORC_PAR
  for (int k = 1; k <= nz; k++) {
    for (int j = js; j <= je; j++) for (int I = Isq; I <= Ieq; I++)
      u_bc_accel[U3(I, j, k)] = (CS->CAu_pred[U3(I, j, k)] + CS->PFu[U3(I, j, k)]) + CS->diffu[U3(I, j, k)];
    for (int J = Jsq; J <= Jeq; J++) for (int i = is; i <= ie; i++)
      v_bc_accel[V3(i, J, k)] = (CS->CAv_pred[V3(i, J, k)] + CS->PFv[V3(i, J, k)]) + CS->diffv[V3(i, J, k)];
  }
  if (OBC) CHECK(orc_open_boundary_zero_normal_flow(G, OBC, u_bc_accel, v_bc_accel));                  /* :565-567 */
  /* up :582-589 */
  ORC_PAR
  for (int k = 1; k <= nz; k++) {
    for (int j = js; j <= je; j++) for (int I = Isq; I <= Ieq; I++)
      up[U3(I, j, k)] = G->mask2dCu[ORC_U2(G, I, j)] * (u_inst[U3(I, j, k)] + dt * u_bc_accel[U3(I, j, k)]);
    for (int J = Jsq; J <= Jeq; J++) for (int i = is; i <= ie; i++)
      vp[V3(i, J, k)] = G->mask2dCv[ORC_V2(G, i, J)] * (v_inst[V3(i, J, k)] + dt * v_bc_accel[V3(i, J, k)]);
  }
  /* set_viscous_ML :592 (DYNAMIC_VISCOUS_ML); vertvisc_coef, vertvisc_remnant :598-600 (without them visc_rem stays 1) */
  if (CS->set_visc_CSp && CS->set_visc_CSp->dynamic_viscous_ML)
    CHECK(orc_set_viscous_ML(G, CS->set_visc_CSp, u_inst, v_inst, h, T, S, CS->eqn_of_state, taux, tauy, CS->visc, dt));
  if (CS->vertvisc_CSp) {
    CHECK(orc_vertvisc_coef_obc(G, CS->vertvisc_CSp, up, vp, h, NULL, CS->visc, dt, OBC));
    CHECK(orc_vertvisc_remnant(G, CS->vertvisc_CSp, CS->visc, CS->visc_rem_u, CS->visc_rem_v, dt));
  }
  /* pass_eta, pass_visc_rem :610-611 */
  orc_halo_update(G, eta, MOM6HIP_POS_H, 1);
  pass3(G, CS->visc_rem_u, MOM6HIP_POS_U | MOM6HIP_PASS_SCALAR_PAIR); pass3(G, CS->visc_rem_v, MOM6HIP_POS_V | MOM6HIP_PASS_SCALAR_PAIR);
  /* btcalc, bt_mass_source :627-630 */
  if (!BT_cont_BT_thick) CHECK(orc_btcalc_obc(G, BT, h, NULL, NULL, 0, OBC));
  orc_bt_mass_source(G, BT, h, eta, 1);
  /* continuity for BT_cont and the layer fluxes :634-644 */
  if (BTC || CS->BT_use_layer_fluxes) {
    CHECK(orc_continuity_obc(G, CS->continuity_CSp, OBC, u_inst, v_inst, h, hp, uh_in, vh_in, dt, NULL, NULL, CS->visc_rem_u, CS->visc_rem_v,
                         NULL, NULL, BTC, NULL, NULL));
    if (BT_cont_BT_thick) CHECK(orc_btcalc_obc(G, BT, h, BTC->h_u, BTC->h_v, 0, OBC));
  }
  if (calc_dtbt) orc_set_dtbt_eta(G, BT, eta, CS->pbce, NULL, 0.0, 0.0);                                        /* :651 */
  /* predictor btstep :655 */
  {
    const int lf = CS->BT_use_layer_fluxes;
    CHECK(orc_btstep_obc(G, BT, u_inst, v_inst, eta, dt, u_bc_accel, v_bc_accel, taux, tauy, RZ_to_H, CS->pbce, CS->eta_PF, u_av, v_av,
                     CS->u_accel_bt, CS->v_accel_bt, eta_pred, CS->uhbt, CS->vhbt, CS->visc_rem_u, CS->visc_rem_v, BTC, eta_PF_start, NULL,
                     NULL, lf ? uh_in : NULL, lf ? vh_in : NULL, lf ? u_inst : NULL, lf ? v_inst : NULL, NULL, OBC));
  }
  /* up = u + dt_pred*(u_bc_accel + u_accel_bt) :663-676 */
  const double dt_pred = dt * CS->be;
  ORC_PAR
  for (int k = 1; k <= nz; k++) {
    for (int J = Jsq; J <= Jeq; J++) for (int i = is; i <= ie; i++)
      vp[V3(i, J, k)] = G->mask2dCv[ORC_V2(G, i, J)] * (v_inst[V3(i, J, k)] + dt_pred * (v_bc_accel[V3(i, J, k)] + CS->v_accel_bt[V3(i, J, k)]));
    for (int j = js; j <= je; j++) for (int I = Isq; I <= Ieq; I++)
      up[U3(I, j, k)] = G->mask2dCu[ORC_U2(G, I, j)] * (u_inst[U3(I, j, k)] + dt_pred * (u_bc_accel[U3(I, j, k)] + CS->u_accel_bt[U3(I, j, k)]));
  }
  /* vertvisc_coef, vertvisc, vertvisc_remnant :717-744 */
  if (CS->vertvisc_CSp) {
    CHECK(orc_vertvisc_coef_obc(G, CS->vertvisc_CSp, up, vp, h, NULL, CS->visc, dt_pred, OBC));
    CHECK(orc_vertvisc_obc(G, CS->vertvisc_CSp, up, vp, h, taux, tauy, CS->visc, dt_pred, NULL, NULL, OBC));
    CHECK(orc_vertvisc_remnant(G, CS->vertvisc_CSp, CS->visc, CS->visc_rem_u, CS->visc_rem_v, dt_pred));
  }
  /* pass_visc_rem :747, pass_uvp :751 */
  pass3(G, CS->visc_rem_u, MOM6HIP_POS_U | MOM6HIP_PASS_SCALAR_PAIR); pass3(G, CS->visc_rem_v, MOM6HIP_POS_V | MOM6HIP_PASS_SCALAR_PAIR);
  pass3(G, up, MOM6HIP_POS_U); pass3(G, vp, MOM6HIP_POS_V);
  /* continuity :757 */
  CHECK(orc_continuity_obc(G, CS->continuity_CSp, OBC, up, vp, h, hp, uh, vh, dt, CS->uhbt, CS->vhbt, CS->visc_rem_u, CS->visc_rem_v, u_av, v_av,
                       BTC, NULL, NULL));
  /* pass_hp_uv :763 */
  pass3(G, hp, MOM6HIP_POS_H); pass3(G, u_av, MOM6HIP_POS_U); pass3(G, v_av, MOM6HIP_POS_V); pass3(G, uh, MOM6HIP_POS_U); pass3(G, vh, MOM6HIP_POS_V);
  if (OBC)                                                                                              /* :765-775 */
    CHECK(orc_radiation_open_bdry_conds(G, OBC, OBC->gamma_uv, OBC->rx_max, OBC->rx_normal, OBC->ry_normal, u_av, u_old_rad_OBC, v_av,
                                        v_old_rad_OBC, dt_pred));
  /* h_av :785-787 */
  ORC_PAR
  for (int k = 1; k <= nz; k++) for (int j = js - 2; j <= je + 2; j++) for (int i = is - 2; i <= ie + 2; i++)
    h_av[H3(i, j, k)] = 0.5 * (h[H3(i, j, k)] + hp[H3(i, j, k)]);
  orc_bt_mass_source(G, BT, hp, eta_pred, 0);                                                          /* :797 */
  if (BT_cont_BT_thick) CHECK(orc_btcalc_obc(G, BT, h, BTC->h_u, BTC->h_v, 0, OBC));                           /* :843 */
  /* horizontal_viscosity :860 (without it diffu stays 0) */
  if (CS->hor_visc)
    CHECK(orc_horizontal_viscosity_obc(G, CS->hor_visc, u_av, v_av, h_av, CS->diffu, CS->diffv, dt, BTC ? BTC->h_u : NULL, BTC ? BTC->h_v : NULL, OBC));
  /* CorAdCalc :869 */
  CHECK(orc_coradcalc_obc(G, CS->CoriolisAdv, OBC, u_av, v_av, h_av, uh, vh, CS->CAu, CS->CAv));
  /* u_bc_accel :879-886 */
  ORC_PAR
  for (int k = 1; k <= nz; k++) {
    for (int j = js; j <= je; j++) for (int I = Isq; I <= Ieq; I++)
      u_bc_accel[U3(I, j, k)] = (CS->CAu[U3(I, j, k)] + CS->PFu[U3(I, j, k)]) + CS->diffu[U3(I, j, k)];
    for (int J = Jsq; J <= Jeq; J++) for (int i = is; i <= ie; i++)
      v_bc_accel[V3(i, J, k)] = (CS->CAv[V3(i, J, k)] + CS->PFv[V3(i, J, k)]) + CS->diffv[V3(i, J, k)];
  }
  if (OBC) CHECK(orc_open_boundary_zero_normal_flow(G, OBC, u_bc_accel, v_bc_accel));                  /* :887-889 */
  /* corrector btstep :911 */
  {
    const int lf = CS->BT_use_layer_fluxes;
    CHECK(orc_btstep_obc(G, BT, u_inst, v_inst, eta, dt, u_bc_accel, v_bc_accel, taux, tauy, RZ_to_H, CS->pbce, CS->eta_PF, u_av, v_av,
                     CS->u_accel_bt, CS->v_accel_bt, eta_pred, CS->uhbt, CS->vhbt, CS->visc_rem_u, CS->visc_rem_v, BTC, eta_PF_start, NULL,
                     NULL, lf ? uh : NULL, lf ? vh : NULL, lf ? u_av : NULL, lf ? v_av : NULL, eta_av, OBC));
  }
  for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) eta[H2(i, j)] = eta_pred[H2(i, j)];   /* :918 */
  /* u = u + dt*(u_bc_accel + u_accel_bt) :928-939 */
  ORC_PAR
  for (int k = 1; k <= nz; k++) {
    for (int j = js; j <= je; j++) for (int I = Isq; I <= Ieq; I++)
      u_inst[U3(I, j, k)] = G->mask2dCu[ORC_U2(G, I, j)] * (u_inst[U3(I, j, k)] + dt * (u_bc_accel[U3(I, j, k)] + CS->u_accel_bt[U3(I, j, k)]));
    for (int J = Jsq; J <= Jeq; J++) for (int i = is; i <= ie; i++)
      v_inst[V3(i, J, k)] = G->mask2dCv[ORC_V2(G, i, J)] * (v_inst[V3(i, J, k)] + dt * (v_bc_accel[V3(i, J, k)] + CS->v_accel_bt[V3(i, J, k)]));
  }
  /* vertvisc_coef, vertvisc, vertvisc_remnant :974-994 */
  if (CS->vertvisc_CSp) {
    CHECK(orc_vertvisc_coef_obc(G, CS->vertvisc_CSp, u_inst, v_inst, h, NULL, CS->visc, dt, OBC));
    CHECK(orc_vertvisc_obc(G, CS->vertvisc_CSp, u_inst, v_inst, h, taux, tauy, CS->visc, dt, NULL, NULL, OBC));
    CHECK(orc_vertvisc_remnant(G, CS->vertvisc_CSp, CS->visc, CS->visc_rem_u, CS->visc_rem_v, dt));
  }
  /* h_av = h :1000-1002 */
  ORC_PAR
  for (int k = 1; k <= nz; k++) for (int j = js - 2; j <= je + 2; j++) for (int i = is - 2; i <= ie + 2; i++)
    h_av[H3(i, j, k)] = h[H3(i, j, k)];
  /* pass_visc_rem :1004, pass_uv :1008 */
  pass3(G, CS->visc_rem_u, MOM6HIP_POS_U | MOM6HIP_PASS_SCALAR_PAIR); pass3(G, CS->visc_rem_v, MOM6HIP_POS_V | MOM6HIP_PASS_SCALAR_PAIR);
  pass3(G, u_inst, MOM6HIP_POS_U); pass3(G, v_inst, MOM6HIP_POS_V);
  /* continuity :1015 */
  CHECK(orc_continuity_obc(G, CS->continuity_CSp, OBC, u_inst, v_inst, h, h, uh, vh, dt, CS->uhbt, CS->vhbt, CS->visc_rem_u, CS->visc_rem_v, u_av,
                       v_av, NULL, NULL, NULL));
  /* pass_h :1018, pass_av_uvh :1027 */
  pass3(G, h, MOM6HIP_POS_H);
  pass3(G, u_av, MOM6HIP_POS_U); pass3(G, v_av, MOM6HIP_POS_V); pass3(G, uh, MOM6HIP_POS_U); pass3(G, vh, MOM6HIP_POS_V);
  if (OBC)                                                                                              /* :1030-1034 */
    CHECK(orc_radiation_open_bdry_conds(G, OBC, OBC->gamma_uv, OBC->rx_max, OBC->rx_normal, OBC->ry_normal, u_inst, u_old_rad_OBC, v_inst,
                                        v_old_rad_OBC, dt));
  /* h_av :1038-1040 */
  ORC_PAR
  for (int k = 1; k <= nz; k++) for (int j = js - 2; j <= je + 2; j++) for (int i = is - 2; i <= ie + 2; i++)
    h_av[H3(i, j, k)] = 0.5 * (h_av[H3(i, j, k)] + h[H3(i, j, k)]);
  /* uhtr, vhtr :1046-1053 */
  ORC_PAR
  for (int k = 1; k <= nz; k++) {
    for (int j = js - 2; j <= je + 2; j++) for (int I = Isq - 2; I <= Ieq + 2; I++)
      uhtr[U3(I, j, k)] = uhtr[U3(I, j, k)] + uh[U3(I, j, k)] * dt;
    for (int J = Jsq - 2; J <= Jeq + 2; J++) for (int i = is - 2; i <= ie + 2; i++)
      vhtr[V3(i, J, k)] = vhtr[V3(i, J, k)] + vh[V3(i, J, k)] * dt;
  }
  if (CS->store_CAu) {   /* :1055-1069 */
    CHECK(orc_coradcalc_obc(G, CS->CoriolisAdv, OBC, u_av, v_av, h_av, uh, vh, CS->CAu_pred, CS->CAv_pred));
    CS->CAu_pred_stored = 1;
  } else {
    CS->CAu_pred_stored = 0;
  }
done:
  free(up); free(vp); free(hp); free(u_bc_accel); free(v_bc_accel); free(uh_in); free(vh_in); free(eta_pred);
  free(u_old_rad_OBC); free(v_old_rad_OBC); free(eta_PF_start);
  return rc;
}

/* ---- SPLIT_RK2B: step_MOM_dyn_split_RK2b, src/core/MOM_dynamics_split_RK2b.F90:274-1050, and the state-setting part of
 * initialize_dyn_split_RK2b (:1220-1420), under the same restrictions as above.  CS->u_av, CS->v_av, CS->h_av are the
 * step's local u_inst, v_inst, h_av (:338-341).  PARITY UNPINNED (assembled from the same pieces). */
int orc_dyn_split_rk2b_init(const mom6hip_grid_t *G, mom6hip_dyn_split_rk2_cs_t *CS, const double *h) {
  const int is = G->isc, ie = G->iec, js = G->jsc, je = G->jec, nz = G->nk;
  if (CS->begw != 0.0 || CS->split_bottom_stress || CS->hooks || !CS->du_av_inst || !CS->dv_av_inst) return 1;
  for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) CS->eta[H2(i, j)] = -G->Z_to_H * G->bathyT[H2(i, j)];   /* :1406-1420 */
  for (int k = 1; k <= nz; k++) for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++)
    CS->eta[H2(i, j)] = CS->eta[H2(i, j)] + h[H3(i, j, k)];
  memset(CS->diffu, 0, sizeof(double) * n_u3(G)); memset(CS->diffv, 0, sizeof(double) * n_v3(G));
  memset(CS->du_av_inst, 0, sizeof(double) * n_u3(G) / nz); memset(CS->dv_av_inst, 0, sizeof(double) * n_v3(G) / nz);
  for (long n = 0; n < n_u3(G); n++) CS->visc_rem_u[n] = 1.0;
  for (long n = 0; n < n_v3(G); n++) CS->visc_rem_v[n] = 1.0;
  return 0;
}

int orc_step_dyn_split_rk2b(const mom6hip_grid_t *G, mom6hip_dyn_split_rk2_cs_t *CS, double *u_av, double *v_av, double *h,
                            const double *T, const double *S, double dt, const double *taux, const double *tauy, double RZ_to_H,
                            double *uh, double *vh, double *uhtr, double *vhtr, double *eta_av, int calc_dtbt) {
  const int is = G->isc, ie = G->iec, js = G->jsc, je = G->jec, nz = G->nk;
  const int Isq = is - 1, Ieq = ie, Jsq = js - 1, Jeq = je;
  if (CS->begw != 0.0 || CS->split_bottom_stress || CS->hooks || (CS->vertvisc_CSp && !CS->visc) || !CS->du_av_inst || !CS->dv_av_inst)
    return 1;
  mom6hip_barotropic_cs_t *BT = CS->barotropic_CSp;
  const mom6hip_bt_cont_t *BTC = CS->BT_cont;
  const int BT_cont_BT_thick = BTC && BTC->h_u && BTC->h_v;
  const long NU = n_u3(G), NV = n_v3(G), NH = n_h3(G);
  double *up = (double *)calloc(NU, sizeof(double)), *vp = (double *)calloc(NV, sizeof(double));       /* :404 */
  double *hp = (double *)malloc(sizeof(double) * NH);
  double *u_bc_accel = (double *)calloc(NU, sizeof(double)), *v_bc_accel = (double *)calloc(NV, sizeof(double));
  double *uh_in = (double *)calloc(NU, sizeof(double)), *vh_in = (double *)calloc(NV, sizeof(double));
  double *eta_pred = (double *)calloc((size_t)ORC_NIH(G) * ORC_NJH(G), sizeof(double));
  double *u_inst = CS->u_av, *v_inst = CS->v_av, *h_av = CS->h_av, *eta = CS->eta;
  memcpy(hp, h, sizeof(double) * NH);                                                                   /* :403 */
  memset(u_inst, 0, sizeof(double) * NU); memset(v_inst, 0, sizeof(double) * NV);                      /* :404 */
  int rc = 0;
  /* CS%OBC: the starting velocities of the radiation conditions (:431-443) */
  const mom6hip_obc_t *OBC = CS->OBC;
  /* the surface pressure of the step :435-442 (RK2b :422-429) */
  const int dyn_p_surf = CS->p_surf_begin && CS->p_surf_end;
  const double *p_surf = dyn_p_surf ? CS->p_surf_end : CS->p_surf;
  double *eta_PF_start = dyn_p_surf ? (double *)calloc((size_t)ORC_NIH(G) * ORC_NJH(G), sizeof(double)) : NULL;
  double *u_old_rad_OBC = NULL, *v_old_rad_OBC = NULL;
  if (OBC) {
    u_old_rad_OBC = (double *)malloc(sizeof(double) * NU); v_old_rad_OBC = (double *)malloc(sizeof(double) * NV);
    memcpy(u_old_rad_OBC, u_av, sizeof(double) * NU); memcpy(v_old_rad_OBC, v_av, sizeof(double) * NV);
  }

  /* continuity with the filtered velocities :488 */
  CHECK(orc_continuity_obc(G, CS->continuity_CSp, OBC, u_av, v_av, h, hp, uh, vh, dt, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL));
  /* PressureForce :498 */
  CHECK(orc_pressureforce_fv_bouss(G, CS->PressureForce_CSp, CS->eqn_of_state, h, T, S, p_surf, CS->PFu, CS->PFv, CS->pbce, CS->eta_PF));
  if (dyn_p_surf) {      /* :497-503 / RK2b :500-506 */
    const double pres_to_eta = 1.0 / (G->g_Earth * (G->Rho0 * G->H_to_Z));      /* 1 / (GV%g_Earth * GV%H_to_RZ), Boussinesq */
    for (int j = Jsq; j <= Jeq+1; j++) for (int i = Isq; i <= Ieq+1; i++)
      eta_PF_start[ORC_H2(G,i,j)] = CS->eta_PF[ORC_H2(G,i,j)] - pres_to_eta * (CS->p_surf_begin[ORC_H2(G,i,j)] - CS->p_surf_end[ORC_H2(G,i,j)]);
  }
  /* pass_hp_uhvh :535 */
  pass3(G, hp, MOM6HIP_POS_H); pass3(G, uh, MOM6HIP_POS_U); pass3(G, vh, MOM6HIP_POS_V);
  /* h_av :540-542 */
  ORC_PAR
  for (int k = 1; k <= nz; k++) for (int j = js - 2; j <= je + 2; j++) for (int i = is - 2; i <= ie + 2; i++)
    h_av[H3(i, j, k)] = 0.5 * (h[H3(i, j, k)] + hp[H3(i, j, k)]);
  /* CorAdCalc :548, horizontal_viscosity :555 */
  CHECK(orc_coradcalc_obc(G, CS->CoriolisAdv, OBC, u_av, v_av, h_av, uh, vh, CS->CAu_pred, CS->CAv_pred));
  if (CS->hor_visc)
    CHECK(orc_horizontal_viscosity_obc(G, CS->hor_visc, u_av, v_av, h_av, CS->diffu, CS->diffv, dt, BTC ? BTC->h_u : NULL, BTC ? BTC->h_v : NULL, OBC));
  /* u_bc_accel :561-568 */
  ORC_PAR
  for (int k = 1; k <= nz; k++) {
    for (int j = js; j <= je; j++) for (int I = Isq; I <= Ieq; I++)
      u_bc_accel[U3(I, j, k)] = (CS->CAu_pred[U3(I, j, k)] + CS->PFu[U3(I, j, k)]) + CS->diffu[U3(I, j, k)];
    for (int J = Jsq; J <= Jeq; J++) for (int i = is; i <= ie; i++)
      v_bc_accel[V3(i, J, k)] = (CS->CAv_pred[V3(i, J, k)] + CS->PFv[V3(i, J, k)]) + CS->diffv[V3(i, J, k)];
  }
  if (OBC) CHECK(orc_open_boundary_zero_normal_flow(G, OBC, u_bc_accel, v_bc_accel));                  /* :571-573 */
  /* up :587-594 */
  ORC_PAR
  for (int k = 1; k <= nz; k++) {
    for (int j = js; j <= je; j++) for (int I = Isq; I <= Ieq; I++)
      up[U3(I, j, k)] = G->mask2dCu[ORC_U2(G, I, j)] * (u_av[U3(I, j, k)] + dt * u_bc_accel[U3(I, j, k)]);
    for (int J = Jsq; J <= Jeq; J++) for (int i = is; i <= ie; i++)
      vp[V3(i, J, k)] = G->mask2dCv[ORC_V2(G, i, J)] * (v_av[V3(i, J, k)] + dt * v_bc_accel[V3(i, J, k)]);
  }
  /* set_viscous_ML :598 (DYNAMIC_VISCOUS_ML); vertvisc_coef, vertvisc_remnant :605-606 */
  if (CS->set_visc_CSp && CS->set_visc_CSp->dynamic_viscous_ML)
    CHECK(orc_set_viscous_ML(G, CS->set_visc_CSp, u_av, v_av, h, T, S, CS->eqn_of_state, taux, tauy, CS->visc, dt));
  if (CS->vertvisc_CSp) {
    CHECK(orc_vertvisc_coef_obc(G, CS->vertvisc_CSp, up, vp, h, NULL, CS->visc, dt, OBC));
    CHECK(orc_vertvisc_remnant(G, CS->vertvisc_CSp, CS->visc, CS->visc_rem_u, CS->visc_rem_v, dt));
  }
  /* pass_eta, pass_visc_rem :616-617 */
  orc_halo_update(G, eta, MOM6HIP_POS_H, 1);
  pass3(G, CS->visc_rem_u, MOM6HIP_POS_U | MOM6HIP_PASS_SCALAR_PAIR); pass3(G, CS->visc_rem_v, MOM6HIP_POS_V | MOM6HIP_PASS_SCALAR_PAIR);
  /* btcalc, bt_mass_source :623-625 */
  if (!BT_cont_BT_thick) CHECK(orc_btcalc_obc(G, BT, h, NULL, NULL, 0, OBC));
  orc_bt_mass_source(G, BT, h, eta, 1);
  /* the instantaneous velocities :641-646 */
  ORC_PAR
  for (int k = 1; k <= nz; k++) {
    for (int j = js; j <= je; j++) for (int I = Isq; I <= Ieq; I++)
      u_inst[U3(I, j, k)] = u_av[U3(I, j, k)] - CS->du_av_inst[ORC_U2(G, I, j)] * CS->visc_rem_u[U3(I, j, k)];
    for (int J = Jsq; J <= Jeq; J++) for (int i = is; i <= ie; i++)
      v_inst[V3(i, J, k)] = v_av[V3(i, J, k)] - CS->dv_av_inst[ORC_V2(G, i, J)] * CS->visc_rem_v[V3(i, J, k)];
  }
  pass3(G, u_inst, MOM6HIP_POS_U); pass3(G, v_inst, MOM6HIP_POS_V);                                    /* :648 */
  /* continuity for BT_cont and the layer fluxes :652 */
  CHECK(orc_continuity_obc(G, CS->continuity_CSp, OBC, u_inst, v_inst, h, hp, uh_in, vh_in, dt, NULL, NULL, CS->visc_rem_u, CS->visc_rem_v,
                       NULL, NULL, BTC, NULL, NULL));
  if (BT_cont_BT_thick) CHECK(orc_btcalc_obc(G, BT, h, BTC->h_u, BTC->h_v, 0, OBC));                            /* :655-658 */
  if (calc_dtbt) orc_set_dtbt_eta(G, BT, eta, CS->pbce, NULL, 0.0, 0.0);                                        /* :664 */
  /* predictor btstep :668 */
  CHECK(orc_btstep_obc(G, BT, u_inst, v_inst, eta, dt, u_bc_accel, v_bc_accel, taux, tauy, RZ_to_H, CS->pbce, CS->eta_PF, u_av, v_av,
                   CS->u_accel_bt, CS->v_accel_bt, eta_pred, CS->uhbt, CS->vhbt, CS->visc_rem_u, CS->visc_rem_v, BTC, eta_PF_start, NULL,
                   NULL, uh_in, vh_in, u_inst, v_inst, NULL, OBC));
  /* up = u_inst + dt_pred*(u_bc_accel + u_accel_bt) :675-686 */
  const double dt_pred = dt * CS->be;
  ORC_PAR
  for (int k = 1; k <= nz; k++) {
    for (int J = Jsq; J <= Jeq; J++) for (int i = is; i <= ie; i++)
      vp[V3(i, J, k)] = G->mask2dCv[ORC_V2(G, i, J)] * (v_inst[V3(i, J, k)] + dt_pred * (v_bc_accel[V3(i, J, k)] + CS->v_accel_bt[V3(i, J, k)]));
    for (int j = js; j <= je; j++) for (int I = Isq; I <= Ieq; I++)
      up[U3(I, j, k)] = G->mask2dCu[ORC_U2(G, I, j)] * (u_inst[U3(I, j, k)] + dt_pred * (u_bc_accel[U3(I, j, k)] + CS->u_accel_bt[U3(I, j, k)]));
  }
  /* vertvisc_coef, vertvisc, vertvisc_remnant :724-745 */
  if (CS->vertvisc_CSp) {
    CHECK(orc_vertvisc_coef_obc(G, CS->vertvisc_CSp, up, vp, h, NULL, CS->visc, dt_pred, OBC));
    CHECK(orc_vertvisc_obc(G, CS->vertvisc_CSp, up, vp, h, taux, tauy, CS->visc, dt_pred, NULL, NULL, OBC));
    CHECK(orc_vertvisc_remnant(G, CS->vertvisc_CSp, CS->visc, CS->visc_rem_u, CS->visc_rem_v, dt_pred));
  }
  pass3(G, CS->visc_rem_u, MOM6HIP_POS_U | MOM6HIP_PASS_SCALAR_PAIR); pass3(G, CS->visc_rem_v, MOM6HIP_POS_V | MOM6HIP_PASS_SCALAR_PAIR);                    /* :748 */
  pass3(G, up, MOM6HIP_POS_U); pass3(G, vp, MOM6HIP_POS_V);                                            /* :752 */
  /* continuity :758 */
  CHECK(orc_continuity_obc(G, CS->continuity_CSp, OBC, up, vp, h, hp, uh, vh, dt, CS->uhbt, CS->vhbt, CS->visc_rem_u, CS->visc_rem_v, u_av, v_av,
                       BTC, NULL, NULL));
  /* pass_hp_uv :764 */
  pass3(G, hp, MOM6HIP_POS_H); pass3(G, u_av, MOM6HIP_POS_U); pass3(G, v_av, MOM6HIP_POS_V); pass3(G, uh, MOM6HIP_POS_U); pass3(G, vh, MOM6HIP_POS_V);
  if (OBC)                                                                                              /* :766-774 */
    CHECK(orc_radiation_open_bdry_conds(G, OBC, OBC->gamma_uv, OBC->rx_max, OBC->rx_normal, OBC->ry_normal, u_av, u_old_rad_OBC, v_av,
                                        v_old_rad_OBC, dt_pred));
  /* h_av :780-782 */
  ORC_PAR
  for (int k = 1; k <= nz; k++) for (int j = js - 2; j <= je + 2; j++) for (int i = is - 2; i <= ie + 2; i++)
    h_av[H3(i, j, k)] = 0.5 * (h[H3(i, j, k)] + hp[H3(i, j, k)]);
  orc_bt_mass_source(G, BT, hp, eta_pred, 0);                                                          /* :790 */
  if (BT_cont_BT_thick) CHECK(orc_btcalc_obc(G, BT, h, BTC->h_u, BTC->h_v, 0, OBC));                           /* :824-827 */
  /* horizontal_viscosity :841, CorAdCalc :848 */
  if (CS->hor_visc)
    CHECK(orc_horizontal_viscosity_obc(G, CS->hor_visc, u_av, v_av, h_av, CS->diffu, CS->diffv, dt, BTC ? BTC->h_u : NULL, BTC ? BTC->h_v : NULL, OBC));
  CHECK(orc_coradcalc_obc(G, CS->CoriolisAdv, OBC, u_av, v_av, h_av, uh, vh, CS->CAu, CS->CAv));
  /* u_bc_accel :854-861 */
  ORC_PAR
  for (int k = 1; k <= nz; k++) {
    for (int j = js; j <= je; j++) for (int I = Isq; I <= Ieq; I++)
      u_bc_accel[U3(I, j, k)] = (CS->CAu[U3(I, j, k)] + CS->PFu[U3(I, j, k)]) + CS->diffu[U3(I, j, k)];
    for (int J = Jsq; J <= Jeq; J++) for (int i = is; i <= ie; i++)
      v_bc_accel[V3(i, J, k)] = (CS->CAv[V3(i, J, k)] + CS->PFv[V3(i, J, k)]) + CS->diffv[V3(i, J, k)];
  }
  if (OBC) CHECK(orc_open_boundary_zero_normal_flow(G, OBC, u_bc_accel, v_bc_accel));                  /* :866-868 */
  /* corrector btstep :889 */
  CHECK(orc_btstep_obc(G, BT, u_inst, v_inst, eta, dt, u_bc_accel, v_bc_accel, taux, tauy, RZ_to_H, CS->pbce, CS->eta_PF, u_av, v_av,
                   CS->u_accel_bt, CS->v_accel_bt, eta_pred, CS->uhbt, CS->vhbt, CS->visc_rem_u, CS->visc_rem_v, BTC, eta_PF_start, NULL,
                   NULL, uh, vh, u_av, v_av, eta_av, OBC));
  for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) eta[H2(i, j)] = eta_pred[H2(i, j)];   /* :898 */
  /* u_inst = u_inst + dt*(u_bc_accel + u_accel_bt) :908-919 */
  ORC_PAR
  for (int k = 1; k <= nz; k++) {
    for (int j = js; j <= je; j++) for (int I = Isq; I <= Ieq; I++)
      u_inst[U3(I, j, k)] = G->mask2dCu[ORC_U2(G, I, j)] * (u_inst[U3(I, j, k)] + dt * (u_bc_accel[U3(I, j, k)] + CS->u_accel_bt[U3(I, j, k)]));
    for (int J = Jsq; J <= Jeq; J++) for (int i = is; i <= ie; i++)
      v_inst[V3(i, J, k)] = G->mask2dCv[ORC_V2(G, i, J)] * (v_inst[V3(i, J, k)] + dt * (v_bc_accel[V3(i, J, k)] + CS->v_accel_bt[V3(i, J, k)]));
  }
  /* vertvisc_coef, vertvisc, vertvisc_remnant :946-963 */
  if (CS->vertvisc_CSp) {
    CHECK(orc_vertvisc_coef_obc(G, CS->vertvisc_CSp, u_inst, v_inst, h, NULL, CS->visc, dt, OBC));
    CHECK(orc_vertvisc_obc(G, CS->vertvisc_CSp, u_inst, v_inst, h, taux, tauy, CS->visc, dt, NULL, NULL, OBC));
    CHECK(orc_vertvisc_remnant(G, CS->vertvisc_CSp, CS->visc, CS->visc_rem_u, CS->visc_rem_v, dt));
  }
  pass3(G, CS->visc_rem_u, MOM6HIP_POS_U | MOM6HIP_PASS_SCALAR_PAIR); pass3(G, CS->visc_rem_v, MOM6HIP_POS_V | MOM6HIP_PASS_SCALAR_PAIR);                    /* :967 */
  pass3(G, u_inst, MOM6HIP_POS_U); pass3(G, v_inst, MOM6HIP_POS_V);                                    /* :971 */
  /* continuity :979-981, returning the barotropic increments */
  CHECK(orc_continuity_obc(G, CS->continuity_CSp, OBC, u_inst, v_inst, h, h, uh, vh, dt, CS->uhbt, CS->vhbt, CS->visc_rem_u, CS->visc_rem_v, u_av,
                       v_av, NULL, CS->du_av_inst, CS->dv_av_inst));
  /* pass_h_uv :993 */
  pass3(G, h, MOM6HIP_POS_H);
  pass3(G, u_av, MOM6HIP_POS_U); pass3(G, v_av, MOM6HIP_POS_V); pass3(G, uh, MOM6HIP_POS_U); pass3(G, vh, MOM6HIP_POS_V);
  if (OBC)                                                                                              /* :1000-1002 */
    CHECK(orc_radiation_open_bdry_conds(G, OBC, OBC->gamma_uv, OBC->rx_max, OBC->rx_normal, OBC->ry_normal, u_av, u_old_rad_OBC, v_av,
                                        v_old_rad_OBC, dt));
  /* uhtr, vhtr :1004-1011 */
  ORC_PAR
  for (int k = 1; k <= nz; k++) {
    for (int j = js - 2; j <= je + 2; j++) for (int I = Isq - 2; I <= Ieq + 2; I++)
      uhtr[U3(I, j, k)] = uhtr[U3(I, j, k)] + uh[U3(I, j, k)] * dt;
    for (int J = Jsq - 2; J <= Jeq + 2; J++) for (int i = is - 2; i <= ie + 2; i++)
      vhtr[V3(i, J, k)] = vhtr[V3(i, J, k)] + vh[V3(i, J, k)] * dt;
  }
done:
  free(up); free(vp); free(hp); free(u_bc_accel); free(v_bc_accel); free(uh_in); free(vh_in); free(eta_pred);
  free(u_old_rad_OBC); free(v_old_rad_OBC); free(eta_PF_start);
  return rc;
}
