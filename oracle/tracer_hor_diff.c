/* oracle/tracer_hor_diff.c -- TEST INFRASTRUCTURE: a C restatement of the along-layer branch of tracer_hordiff
 * (src/tracer/MOM_tracer_hor_diff.F90:119-680) with KHTR or the VarMix / MEKE diffusivities, and the call of the neutral-diffusion branch
 * (neutral_diffusion.c) and tracer_epipycnal_ML_diff (:700-1621, DIFFUSE_ML_TO_INTERIOR); no boundary diffusion.  The reference holds no known-answer vectors for this routine: parity unpinned; the
 * tests hold it to exact conservation, preservation of constants and the maximum principle. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include "mom6_oracle.h"

static inline double min2(double a, double b) { return a < b ? a : b; }

static inline double max2(double a, double b) { return a > b ? a : b; }

int orc_tracer_hordiff(const mom6hip_grid_t *G, const mom6hip_tracer_hor_diff_cs_t *CS, const double *h, double dt,
                       double *const *tr, const double *conc_underflow, int ntr, mom6hip_hordiff_stats_t *stats)
{
  return orc_tracer_hordiff_varmix(G, CS, NULL, h, dt, tr, conc_underflow, ntr, stats);
}

int orc_tracer_hordiff_varmix(const mom6hip_grid_t *G, const mom6hip_tracer_hor_diff_cs_t *CS, const mom6hip_hordiff_fields_t *F,
                              const double *h, double dt, double *const *tr, const double *conc_underflow, int ntr,
                              mom6hip_hordiff_stats_t *stats)
{
  if (CS->unsupported[0]) return 2;
  return orc_tracer_hordiff_neutral(G, CS, NULL, F, h, NULL, NULL, dt, tr, conc_underflow, ntr, 0, 0, stats);
}

static int hordiff_core(const mom6hip_grid_t *G, const mom6hip_tracer_hor_diff_cs_t *CS, const mom6hip_neutral_diffusion_cs_t *ND,
                        const mom6hip_epipycnal_cs_t *EP, const mom6hip_hordiff_fields_t *F, const double *h, const mom6hip_eos_t *eos,
                        const double *p_surf, double dt, double *const *tr, const double *conc_underflow, int ntr, int idx_T, int idx_S,
                        mom6hip_hordiff_stats_t *stats);
static int epipycnal_ML_diff(const mom6hip_grid_t *G, const mom6hip_epipycnal_cs_t *EP, const mom6hip_eos_t *eos, const double *h,
                             double *const *tr, const double *conc_underflow, int ntr, int idx_T, int idx_S, const double *khdt_epi_x,
                             const double *khdt_epi_y, int num_itts, int *halo_updates);

int orc_tracer_hordiff_neutral(const mom6hip_grid_t *G, const mom6hip_tracer_hor_diff_cs_t *CS,
                               const mom6hip_neutral_diffusion_cs_t *ND, const mom6hip_hordiff_fields_t *F, const double *h,
                               const mom6hip_eos_t *eos, const double *p_surf, double dt, double *const *tr,
                               const double *conc_underflow, int ntr, int idx_T, int idx_S, mom6hip_hordiff_stats_t *stats)
{
  if (CS->unsupported[2]) return 2;
  return hordiff_core(G, CS, ND, NULL, F, h, eos, p_surf, dt, tr, conc_underflow, ntr, idx_T, idx_S, stats);
}

int orc_tracer_hordiff_epipycnal(const mom6hip_grid_t *G, const mom6hip_tracer_hor_diff_cs_t *CS, const mom6hip_epipycnal_cs_t *EP,
                                 const mom6hip_hordiff_fields_t *F, const double *h, const mom6hip_eos_t *eos, double dt,
                                 double *const *tr, const double *conc_underflow, int ntr, int idx_T, int idx_S,
                                 mom6hip_hordiff_stats_t *stats)
{
  if (CS->unsupported[0]) return 2;                                /* mutually exclusive :1732 */
  if (CS->unsupported[2]) {
    if (!EP || !eos || !EP->Rlay || idx_T < 0 || idx_T >= ntr || idx_S < 0 || idx_S >= ntr) return 3;
    if (EP->nk_rho_varies < 1 || EP->nk_rho_varies >= G->nk || EP->nkml < 0 || EP->nkml > EP->nk_rho_varies) return 3;
    if (G->isc - G->isd < 2 || G->jsc - G->jsd < 2) return 3;
  }
  return hordiff_core(G, CS, NULL, CS->unsupported[2] ? EP : NULL, F, h, eos, NULL, dt, tr, conc_underflow, ntr, idx_T, idx_S, stats);
}

static int hordiff_core(const mom6hip_grid_t *G, const mom6hip_tracer_hor_diff_cs_t *CS, const mom6hip_neutral_diffusion_cs_t *ND,
                        const mom6hip_epipycnal_cs_t *EP, const mom6hip_hordiff_fields_t *F, const double *h, const mom6hip_eos_t *eos,
                        const double *p_surf, double dt, double *const *tr, const double *conc_underflow, int ntr, int idx_T, int idx_S,
                        mom6hip_hordiff_stats_t *stats)
{
  for (int q = 1; q < 8; q++) if (q != 2 && CS->unsupported[q]) return 2;
  const int use_neutral = CS->unsupported[0] != 0;
  if (use_neutral) {
    if (!ND || !eos || idx_T < 0 || idx_T >= ntr || idx_S < 0 || idx_S >= ntr) return 3;
    for (int q = 0; q < 8; q++) if (ND->unsupported[q]) return 2;
    if (ND->interior_only && !(F && F->h_ML)) return 3;
  }
  if (stats) { stats->num_itts = 0; stats->halo_updates = 0; stats->max_CFL = 0.0; }
  const int use_VarMix = CS->use_variable_mixing != 0;
  if (ntr == 0 || (CS->KhTr <= 0.0 && !use_VarMix)) return 0;      /* :197 */
  const int use_Eady = use_VarMix && CS->KhTr_Slope_Cff > 0., Resoln_scaled = use_VarMix && CS->Resoln_scaled_KhTr;
  if (use_Eady && !(F && F->L2u && F->L2v && F->SN_u && F->SN_v)) return 3;
  if (Resoln_scaled && !(F && F->Res_fn_h)) return 3;
  if (use_VarMix && CS->KhTr_passivity_coeff > 0. && !(F && F->Rd_dx_h)) return 3;
  const int is = G->isc, ie = G->iec, js = G->jsc, je = G->jec, nz = G->nk;
  const int isd = G->isd, jsd = G->jsd;
  const int nih = G->ied - G->isd + 1, njh = G->jed - G->jsd + 1;
#define H2(i,j) ((size_t)((j)-jsd)*nih + ((i)-isd))
#define U2(I,j) ((size_t)((j)-jsd)*(nih+1) + ((I)-isd+1))
#define V2(i,J) ((size_t)((J)-jsd+1)*nih + ((i)-isd))
  const size_t hpl = (size_t)nih*njh;
  const double h_neglect = G->H_subroundoff;                       /* :208 */
  double *khdt_x = (double*)calloc((size_t)(nih+1)*njh, sizeof(double));
  double *khdt_y = (double*)calloc((size_t)nih*(njh+1), sizeof(double));
  double *Coef_x = (double*)calloc((size_t)(nih+1)*njh, sizeof(double));
  double *Coef_y = (double*)calloc((size_t)nih*(njh+1), sizeof(double));
  double *Ihdxdy = (double*)calloc(hpl, sizeof(double));
  double *dTr = (double*)calloc(hpl, sizeof(double));
  if (use_VarMix) {                                                /* :236-281 */
    for (int dir = 0; dir < 2; dir++)
      for (int j = (dir ? js-1 : js); j <= je; j++) for (int i = (dir ? is : is-1); i <= ie; i++) {
        const size_t c0 = H2(i,j), c1 = dir ? H2(i,j+1) : H2(i+1,j), f = dir ? V2(i,j) : U2(i,j);
        double Kh_loc = CS->KhTr, Kh;
        if (use_Eady) Kh_loc = Kh_loc + CS->KhTr_Slope_Cff*(dir ? F->L2v[f] : F->L2u[f])*(dir ? F->SN_v[f] : F->SN_u[f]);
        if (F && F->MEKE_Kh) Kh_loc = Kh_loc + CS->KhTr_fac*sqrt(F->MEKE_Kh[c0]*F->MEKE_Kh[c1]);
        if (CS->KhTr_max > 0.) Kh_loc = min2(Kh_loc, CS->KhTr_max);
        if (Resoln_scaled) Kh_loc = Kh_loc * 0.5*(F->Res_fn_h[c0] + F->Res_fn_h[c1]);
        Kh = max2(Kh_loc, CS->KhTr_min);
        if (CS->KhTr_passivity_coeff > 0.) {
          const double Rd_dx = 0.5*(F->Rd_dx_h[c0] + F->Rd_dx_h[c1]);
          Kh_loc = Kh*max2(CS->KhTr_passivity_min, CS->KhTr_passivity_coeff*Rd_dx);
          if (CS->KhTr_max > 0.) Kh_loc = min2(Kh_loc, CS->KhTr_max);
          Kh = max2(Kh_loc, CS->KhTr_min);
        }
        if (dir) khdt_y[f] = dt*(Kh*(G->dx_Cv[f]*G->IdyCv[f]));
        else khdt_x[f] = dt*(Kh*(G->dy_Cu[f]*G->IdxCu[f]));
      }
  } else {
  /* a simple constant diffusivity :305-328 */
  for (int j = js; j <= je; j++) for (int I = is-1; I <= ie; I++)
    khdt_x[U2(I,j)] = dt*(CS->KhTr*(G->dy_Cu[U2(I,j)]*G->IdxCu[U2(I,j)]));
  for (int J = js-1; J <= je; J++) for (int i = is; i <= ie; i++)
    khdt_y[V2(i,J)] = dt*(CS->KhTr*(G->dx_Cv[V2(i,J)]*G->IdyCv[V2(i,J)]));
  }
  if (CS->max_diff_CFL > 0.0) {                                    /* :368-398 */
    for (int j = js; j <= je; j++) for (int I = is-1; I <= ie; I++) {
      const int i = I;
      const double khdt_max = 0.125*CS->max_diff_CFL * min2(G->areaT[H2(i,j)], G->areaT[H2(i+1,j)]);
      khdt_x[U2(I,j)] = min2(khdt_x[U2(I,j)], khdt_max);
    }
    for (int J = js-1; J <= je; J++) for (int i = is; i <= ie; i++) {
      const int j = J;
      const double khdt_max = 0.125*CS->max_diff_CFL * min2(G->areaT[H2(i,j)], G->areaT[H2(i,j+1)]);
      khdt_y[V2(i,J)] = min2(khdt_y[V2(i,J)], khdt_max);
    }
  }
  int num_itts; double I_numitts, max_CFL = 0.0;
  if (CS->check_diffusive_CFL) {                                   /* :410-423 */
    for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
      const int I = i, J = j;
      const double CFL = 2.0*((khdt_x[U2(I-1,j)] + khdt_x[U2(I,j)]) + (khdt_y[V2(i,J-1)] + khdt_y[V2(i,J)])) * G->IareaT[H2(i,j)];
      if (max_CFL < CFL) max_CFL = CFL;
    }
    num_itts = (int)ceil(max_CFL - 4.0*DBL_EPSILON);              /* EPSILON(max_CFL): the machine epsilon of the kind */
    if (num_itts < 1) num_itts = 1;
    I_numitts = 1.0 / ((double)num_itts);
  } else if (CS->max_diff_CFL > 0.0) {
    num_itts = (int)ceil(CS->max_diff_CFL - 4.0*DBL_EPSILON);
    if (num_itts < 1) num_itts = 1;
    I_numitts = 1.0 / ((double)num_itts);
  } else { num_itts = 1; I_numitts = 1.0; }

  int halo_updates = 0, rc = 0;
  if (use_neutral)                                                 /* :474-534 */
    rc = orc_neutral_branch(G, ND, eos, h, p_surf, (ND->interior_only && F) ? F->h_ML : NULL, khdt_x, khdt_y, num_itts, I_numitts, tr, conc_underflow, ntr, idx_T, idx_S,
                            &halo_updates);
  else
  for (int itt = 1; itt <= num_itts; itt++) {                      /* :540-604 */
    for (int m = 0; m < ntr; m++) orc_halo_update(G, tr[m], MOM6HIP_POS_H, nz);
    halo_updates++;
    for (int k = 0; k < nz; k++) {
      double scale = I_numitts;
      if (EP) {                                                    /* CS%Diffuse_ML_interior :544-550 */
        if (k+1 <= EP->nkml) {
          if (EP->ML_KhTr_scale <= 0.0) continue;
          scale = I_numitts * EP->ML_KhTr_scale;
        }
        if ((k+1 > EP->nkml) && (k+1 <= EP->nk_rho_varies)) continue;
      }
      const double *hk = h + hpl*k;
      for (int J = js-1; J <= je; J++) for (int i = is; i <= ie; i++) {
        const int j = J;
        Coef_y[V2(i,J)] = ((scale * khdt_y[V2(i,J)])*2.0*(hk[H2(i,j)]*hk[H2(i,j+1)])) / (hk[H2(i,j)]+hk[H2(i,j+1)]+h_neglect);
      }
      for (int j = js; j <= je; j++) {
        for (int I = is-1; I <= ie; I++) {
          const int i = I;
          Coef_x[U2(I,j)] = ((scale * khdt_x[U2(I,j)])*2.0*(hk[H2(i,j)]*hk[H2(i+1,j)])) / (hk[H2(i,j)]+hk[H2(i+1,j)]+h_neglect);
        }
        for (int i = is; i <= ie; i++) Ihdxdy[H2(i,j)] = G->IareaT[H2(i,j)] / (hk[H2(i,j)]+h_neglect);
      }
      for (int m = 0; m < ntr; m++) {
        double *t = tr[m] + hpl*k;
        for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
          const int I = i, J = j;
          dTr[H2(i,j)] = Ihdxdy[H2(i,j)] *
            ((Coef_x[U2(I-1,j)] * (t[H2(i-1,j)] - t[H2(i,j)]) -
              Coef_x[U2(I,j)] * (t[H2(i,j)] - t[H2(i+1,j)])) +
             (Coef_y[V2(i,J-1)] * (t[H2(i,j-1)] - t[H2(i,j)]) -
              Coef_y[V2(i,J)] * (t[H2(i,j)] - t[H2(i,j+1)])));
        }
        for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) t[H2(i,j)] = t[H2(i,j)] + dTr[H2(i,j)];
      }
    }
    for (int m = 0; m < ntr; m++) if (conc_underflow && conc_underflow[m] > 0.0) {   /* :607-612 */
      for (int k = 0; k < nz; k++) for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
        double *t = tr[m] + hpl*k + H2(i,j);
        if (fabs(*t) < conc_underflow[m]) *t = 0.0;
      }
    }
  }
  if (EP && rc == 0)                                               /* :613-620 */
    rc = epipycnal_ML_diff(G, EP, eos, h, tr, conc_underflow, ntr, idx_T, idx_S, khdt_x, khdt_y, num_itts, &halo_updates);
  if (stats) { stats->num_itts = num_itts; stats->halo_updates = halo_updates; stats->max_CFL = max_CFL; }
#undef H2
#undef U2
#undef V2
  free(khdt_x); free(khdt_y); free(Coef_x); free(Coef_y); free(Ihdxdy); free(dTr);
  return rc;
}


/* ---- tracer_epipycnal_ML_diff :700-1621 ------------------------------------------------------------------------------------ */

/* the pairings of one face between the sorted columns L and R (:930-1081 at u faces, :1101-1251 at v faces: the same text with
 * (i+1,j) or (i,j+1) as the right column).  nL, nR: num_srt; k0 / rho / hs: the sorted lists (0-based positions, k0 1-based
 * layer numbers); the outputs are the face's lists of at most nL + nR pairs.  Returns nP. */
static int epi_pairings(int nL, const int *k0L, const double *rhoL, const double *hL,
                        int nR, const int *k0R, const double *rhoR, const double *hR, int nkmb, int nz,
                        int *k0b_L, int *k0a_L, int *k0b_R, int *k0a_R, double *deep_wt_L, double *deep_wt_R, double *hP_L, double *hP_R)
{
  double h_demand_L[nz+1], h_demand_R[nz+1], h_used_L[nz+1], h_used_R[nz+1], h_supply_frac_L[nz+1], h_supply_frac_R[nz+1];
  int kbs_Lp[2*nz+1], kbs_Rp[2*nz+1], left_set[2*nz+1], right_set[2*nz+1];
  int kL, kR, nP = 0;
  /* positions are 1-based as in the reference: list element kL is rhoL[kL-1] */
#define RL(k) rhoL[(k)-1]
#define RR(k) rhoR[(k)-1]
#define HL(k) hL[(k)-1]
#define HR(k) hR[(k)-1]
  for (int k = 1; k <= nL; k++) { h_demand_L[k] = 0.0; h_used_L[k] = 0.0; }
  for (int k = 1; k <= nR; k++) { h_demand_R[k] = 0.0; h_used_R[k] = 0.0; }
  /* layers lighter than the lightest of the other column are discarded :938-946 (an empty list has no first element: every
   * branch leaves the loop below at once) */
  const double rL1 = nL > 0 ? RL(1) : 0.0, rR1 = nR > 0 ? RR(1) : 0.0;
  if (rL1 < rR1) {
    kR = 1;
    for (kL = 2; kL <= nL; kL++) if (RL(kL) >= rR1) break;
  } else if (rR1 < rL1) {
    kL = 1;
    for (kR = 2; kR <= nR; kR++) if (RR(kR) >= rL1) break;
  } else { kL = 1; kR = 1; }
  for (;;) {                                                       /* :948-1010 */
    if ((kL > nL) || (kR > nR)) break;
    if (RL(kL) > RR(kR)) {                                         /* the right point is lighter and defines the density */
      nP = nP+1; const int k = nP - 1;
      const double rho_pair = RR(kR);
      k0b_L[k] = k0L[kL-1]; k0b_R[k] = k0R[kR-1];
      k0a_L[k] = k0L[kL-2]; k0a_R[k] = k0b_R[k];
      kbs_Lp[k] = kL; kbs_Rp[k] = kR;
      const double rho_a = RL(kL-1), rho_b = RL(kL);
      double wt_b = 1.0; if (fabs(rho_a - rho_b) > fabs(rho_pair - rho_a)) wt_b = (rho_pair - rho_a) / (rho_b - rho_a);
      deep_wt_L[k] = wt_b; deep_wt_R[k] = 1.0;
      h_demand_L[kL] = h_demand_L[kL] + 0.5*HR(kR) * wt_b;
      h_demand_L[kL-1] = h_demand_L[kL-1] + 0.5*HR(kR) * (1.0-wt_b);
      kR = kR+1; left_set[k] = 0; right_set[k] = 1;
    } else if (RL(kL) < RR(kR)) {                                  /* the left point is lighter */
      nP = nP+1; const int k = nP - 1;
      const double rho_pair = RL(kL);
      k0b_L[k] = k0L[kL-1]; k0b_R[k] = k0R[kR-1];
      k0a_L[k] = k0b_L[k]; k0a_R[k] = k0R[kR-2];
      kbs_Lp[k] = kL; kbs_Rp[k] = kR;
      const double rho_a = RR(kR-1), rho_b = RR(kR);
      double wt_b = 1.0; if (fabs(rho_a - rho_b) > fabs(rho_pair - rho_a)) wt_b = (rho_pair - rho_a) / (rho_b - rho_a);
      deep_wt_L[k] = 1.0; deep_wt_R[k] = wt_b;
      h_demand_R[kR] = h_demand_R[kR] + 0.5*HL(kL) * wt_b;
      h_demand_R[kR-1] = h_demand_R[kR-1] + 0.5*HL(kL) * (1.0-wt_b);
      kL = kL+1; left_set[k] = 1; right_set[k] = 0;
    } else if ((k0L[kL-1] <= nkmb) || (k0R[kR-1] <= nkmb)) {       /* equal densities, one layer above the interior */
      nP = nP+1; const int k = nP - 1;
      k0b_L[k] = k0L[kL-1]; k0b_R[k] = k0R[kR-1];
      k0a_L[k] = k0b_L[k]; k0a_R[k] = k0b_R[k];
      kbs_Lp[k] = kL; kbs_Rp[k] = kR;
      deep_wt_L[k] = 1.0; deep_wt_R[k] = 1.0;
      h_demand_L[kL] = h_demand_L[kL] + 0.5*HR(kR);
      h_demand_R[kR] = h_demand_R[kR] + 0.5*HL(kL);
      kL = kL+1; kR = kR+1; left_set[k] = 1; right_set[k] = 1;
    } else {                                                       /* equal densities in the interior: already mixed */
      h_demand_L[kL] = h_demand_L[kL] + 0.5*HR(kR);
      h_demand_R[kR] = h_demand_R[kR] + 0.5*HL(kL);
      kL = kL+1; kR = kR+1;
    }
  }
  /* the fraction of the demand that can be supplied :1013-1023 */
  for (int k = 1; k <= nR; k++) {
    h_supply_frac_R[k] = 1.0;
    if (h_demand_R[k] > 0.5*HR(k)) h_supply_frac_R[k] = 0.5*HR(k) / h_demand_R[k];
  }
  for (int k = 1; k <= nL; k++) {
    h_supply_frac_L[k] = 1.0;
    if (h_demand_L[k] > 0.5*HL(k)) h_supply_frac_L[k] = 0.5*HL(k) / h_demand_L[k];
  }
  /* the exported thicknesses :1026-1053 */
  for (int k = 0; k < nP; k++) {
    kL = kbs_Lp[k]; kR = kbs_Rp[k];
    hP_L[k] = 0.0; hP_R[k] = 0.0;
    if (left_set[k]) {
      if (deep_wt_R[k] < 1.0) {
        hP_R[k] = 0.5*HL(kL) * min2(h_supply_frac_R[kR], h_supply_frac_R[kR-1]);
        const double wt_b = deep_wt_R[k];
        h_used_R[kR-1] = h_used_R[kR-1] + (1.0 - wt_b)*hP_R[k];
        h_used_R[kR] = h_used_R[kR] + wt_b*hP_R[k];
      } else {
        hP_R[k] = 0.5*HL(kL) * h_supply_frac_R[kR];
        h_used_R[kR] = h_used_R[kR] + hP_R[k];
      }
    }
    if (right_set[k]) {
      if (deep_wt_L[k] < 1.0) {
        hP_L[k] = 0.5*HR(kR) * min2(h_supply_frac_L[kL], h_supply_frac_L[kL-1]);
        const double wt_b = deep_wt_L[k];
        h_used_L[kL-1] = h_used_L[kL-1] + (1.0 - wt_b)*hP_L[k];
        h_used_L[kL] = h_used_L[kL] + wt_b*hP_L[k];
      } else {
        hP_L[k] = 0.5*HR(kR) * h_supply_frac_L[kL];
        h_used_L[kL] = h_used_L[kL] + hP_L[k];
      }
    }
  }
  /* the left-over thickness goes to the importing columns :1057-1062 */
  for (int k = 0; k < nP; k++) {
    if (left_set[k]) hP_L[k] = hP_L[k] + (HL(kbs_Lp[k]) - h_used_L[kbs_Lp[k]]);
    if (right_set[k]) hP_R[k] = hP_R[k] + (HR(kbs_Rp[k]) - h_used_R[kbs_Rp[k]]);
  }
#undef RL
#undef RR
#undef HL
#undef HR
  return nP;
}

/* the vertical adjustment that keeps the two pieces of the left (exporting for Tr_flux > 0) side within the face's range
 * :1336-1352 (u) = :1487-1503 (v) */
static double epi_adj_left(double Tr_flux, double Tr_La, double Tr_Lb, double vol, double wt_a, double wt_b, double Tr_min_face,
                           double Tr_max_face)
{
  double Tr_adj_vert = 0.0;
  if (Tr_flux > 0.0) {
    if (Tr_La < Tr_Lb) { if (vol*(Tr_La-Tr_min_face) < Tr_flux)
      Tr_adj_vert = -wt_a * min2(Tr_flux - vol * (Tr_La-Tr_min_face), (vol*wt_b) * (Tr_Lb - Tr_La));
    } else { if (vol*(Tr_Lb-Tr_min_face) < Tr_flux)
      Tr_adj_vert = wt_b * min2(Tr_flux - vol * (Tr_Lb-Tr_min_face), (vol*wt_a) * (Tr_La - Tr_Lb));
    }
  } else if (Tr_flux < 0.0) {
    if (Tr_La > Tr_Lb) { if (vol * (Tr_max_face-Tr_La) < -Tr_flux)
      Tr_adj_vert = wt_a * min2(-Tr_flux - vol * (Tr_max_face-Tr_La), (vol*wt_b) * (Tr_La - Tr_Lb));
    } else { if (vol*(Tr_max_face-Tr_Lb) < -Tr_flux)
      Tr_adj_vert = -wt_b * min2(-Tr_flux - vol * (Tr_max_face-Tr_Lb), (vol*wt_a)*(Tr_Lb - Tr_La));
    }
  }
  return Tr_adj_vert;
}

/* the same for the right side :1383-1399 (u) = :1516-1532 (v) */
static double epi_adj_right(double Tr_flux, double Tr_Ra, double Tr_Rb, double vol, double wt_a, double wt_b, double Tr_min_face,
                            double Tr_max_face)
{
  double Tr_adj_vert = 0.0;
  if (Tr_flux < 0.0) {
    if (Tr_Ra < Tr_Rb) { if (vol * (Tr_Ra-Tr_min_face) < -Tr_flux)
      Tr_adj_vert = -wt_a * min2(-Tr_flux - vol * (Tr_Ra-Tr_min_face), (vol*wt_b) * (Tr_Rb - Tr_Ra));
    } else { if (vol*(Tr_Rb-Tr_min_face) < (-Tr_flux))
      Tr_adj_vert = wt_b * min2(-Tr_flux - vol * (Tr_Rb-Tr_min_face), (vol*wt_a) * (Tr_Ra - Tr_Rb));
    }
  } else if (Tr_flux > 0.0) {
    if (Tr_Ra > Tr_Rb) { if (vol * (Tr_max_face-Tr_Ra) < Tr_flux)
      Tr_adj_vert = wt_a * min2(Tr_flux - vol * (Tr_max_face-Tr_Ra), (vol*wt_b) * (Tr_Ra - Tr_Rb));
    } else { if (vol*(Tr_max_face-Tr_Rb) < Tr_flux)
      Tr_adj_vert = -wt_b * min2(Tr_flux - vol * (Tr_max_face-Tr_Rb), (vol*wt_a)*(Tr_Rb - Tr_Ra));
    }
  }
  return Tr_adj_vert;
}

static inline double min3(double a, double b, double c) { return min2(min2(a, b), c); }
static inline double max3(double a, double b, double c) { return max2(max2(a, b), c); }
static inline double min5(double a, double b, double c, double d, double e) { return min2(min2(min2(min2(a, b), c), d), e); }
static inline double max5(double a, double b, double c, double d, double e) { return max2(max2(max2(max2(a, b), c), d), e); }

static int epipycnal_ML_diff(const mom6hip_grid_t *G, const mom6hip_epipycnal_cs_t *EP, const mom6hip_eos_t *eos, const double *h,
                             double *const *Tr, const double *conc_underflow, int ntr, int idx_T, int idx_S, const double *khdt_epi_x,
                             const double *khdt_epi_y, int num_itts, int *halo_updates)
{
  const int is = G->isc, ie = G->iec, js = G->jsc, je = G->jec, nz = G->nk;
  const int isd = G->isd, jsd = G->jsd;
  const int nih = G->ied - G->isd + 1, njh = G->jed - G->jsd + 1;
  const int nkmb = EP->nk_rho_varies;
  const double *Rlay = EP->Rlay;                                   /* Rlay[k-1] = GV%Rlay(k) */
#define H2(i,j) ((size_t)((j)-jsd)*nih + ((i)-isd))
#define U2(I,j) ((size_t)((j)-jsd)*(nih+1) + ((I)-isd+1))
#define V2(i,J) ((size_t)((J)-jsd+1)*nih + ((i)-isd))
#define H3(i,j,k) (hpl*((k)-1) + H2(i,j))                          /* k 1-based */
  const size_t hpl = (size_t)nih*njh, upl = (size_t)(nih+1)*njh, vpl = (size_t)nih*(njh+1);
  int max_itt; double I_maxitt;
  if (num_itts <= 1) { max_itt = 1; I_maxitt = 1.0; }
  else { max_itt = num_itts; I_maxitt = 1.0 / ((double)max_itt); }

  double *Rml_max = (double*)calloc(hpl, sizeof(double));
  double *rho_coord = (double*)calloc(hpl*nkmb, sizeof(double));
  int *num_srt = (int*)calloc(hpl, sizeof(int)), *k_end_srt = (int*)calloc(hpl, sizeof(int)), *max_kRho = (int*)calloc(hpl, sizeof(int));
  double *rho_srt = (double*)calloc(hpl*nz, sizeof(double)), *h_srt = (double*)calloc(hpl*nz, sizeof(double));
  int *k0_srt = (int*)calloc(hpl*nz, sizeof(int));                 /* column c, position n (0-based): [c*nz + n] */
  const size_t np2 = 2*(size_t)nz;                                 /* pairs per face */
  int *nPu = (int*)calloc(upl, sizeof(int)), *nPv = (int*)calloc(vpl, sizeof(int));
  int *ku = (int*)calloc(upl*np2*4, sizeof(int)), *kv = (int*)calloc(vpl*np2*4, sizeof(int));        /* k0b_L, k0a_L, k0b_R, k0a_R */
  double *wu = (double*)calloc(upl*np2*4, sizeof(double)), *wv = (double*)calloc(vpl*np2*4, sizeof(double));  /* deep_wt_L, deep_wt_R, hP_L, hP_R */
#define PK(a,f,q) ((a) + ((size_t)(f)*4 + (q))*np2)
  double *tr_flux_conv = (double*)calloc(hpl*nz, sizeof(double));
  double *tr_flux_N = (double*)calloc(hpl*nz, sizeof(double)), *tr_flux_S = (double*)calloc(hpl*nz, sizeof(double));
  double *tr_flux_E = (double*)calloc(hpl*nz, sizeof(double)), *tr_flux_W = (double*)calloc(hpl*nz, sizeof(double));
  double *Tr_flux_3d = (double*)calloc(vpl*np2, sizeof(double)), *Tr_adj_vert_L = (double*)calloc(vpl*np2, sizeof(double));
  double *Tr_adj_vert_R = (double*)calloc(vpl*np2, sizeof(double));

  for (int m = 0; m < ntr; m++) orc_halo_update(G, Tr[m], MOM6HIP_POS_H, nz);      /* do_group_pass(CS%pass_t) :832 */
  (*halo_updates)++;
  /* the coordinate density of the variable-density layers :835-848 */
  for (int k = 1; k <= nkmb; k++) for (int j = js-2; j <= je+2; j++) for (int i = is-2; i <= ie+2; i++)
    rho_coord[hpl*(k-1) + H2(i,j)] = orc_eos_density(eos, Tr[idx_T][H3(i,j,k)], Tr[idx_S][H3(i,j,k)], EP->P_Ref);
  for (int j = js-2; j <= je+2; j++) for (int i = is-2; i <= ie+2; i++) {
    Rml_max[H2(i,j)] = rho_coord[H2(i,j)];
    num_srt[H2(i,j)] = 0; max_kRho[H2(i,j)] = 0;
  }
  for (int k = 2; k <= nkmb; k++) for (int j = js-2; j <= je+2; j++) for (int i = is-2; i <= ie+2; i++)
    if (Rml_max[H2(i,j)] < rho_coord[hpl*(k-1) + H2(i,j)]) Rml_max[H2(i,j)] = rho_coord[hpl*(k-1) + H2(i,j)];
  /* GV%Rlay(max_kRho-1) < Rml_max <= GV%Rlay(max_kRho) by bisection :854-869 */
#define RLAY(k) Rlay[(k)-1]
  for (int j = js-2; j <= je+2; j++) for (int i = is-2; i <= ie+2; i++) if (G->mask2dT[H2(i,j)] > 0.0) {
    const double R = Rml_max[H2(i,j)];
    if ((nkmb+1 > nz) || (R > RLAY(nz))) max_kRho[H2(i,j)] = nz+1;
    else if ((nkmb+2 > nz) || (R <= RLAY(nkmb+1))) max_kRho[H2(i,j)] = nkmb+1;
    else {
      int k_min = nkmb+2, k_max = nz;
      for (;;) {
        const int k_test = (k_min + k_max) / 2;
        if (R <= RLAY(k_test-1)) k_max = k_test-1;
        else if (RLAY(k_test) < R) k_min = k_test+1;
        else { max_kRho[H2(i,j)] = k_test; break; }
        if (k_min == k_max) { max_kRho[H2(i,j)] = k_max; break; }
      }
    }
  }
  int PEmax_kRho = 0;
  for (int j = js-1; j <= je+1; j++) for (int i = is-1; i <= ie+1; i++) {
    int m = max_kRho[H2(i,j)];
    if (m < max_kRho[H2(i-1,j)]) m = max_kRho[H2(i-1,j)];
    if (m < max_kRho[H2(i+1,j)]) m = max_kRho[H2(i+1,j)];
    if (m < max_kRho[H2(i,j-1)]) m = max_kRho[H2(i,j-1)];
    if (m < max_kRho[H2(i,j+1)]) m = max_kRho[H2(i,j+1)];
    k_end_srt[H2(i,j)] = m;
    if (PEmax_kRho < m) PEmax_kRho = m;
  }
  if (PEmax_kRho > nz) PEmax_kRho = nz;
  const double h_exclude = 10.0*(G->Angstrom_H + G->H_subroundoff);
  /* the lists of the layers that take part :880-899, sorted by density (straight insertion) :902-912 */
  for (int j = js-1; j <= je+1; j++) {
    for (int k = 1; k <= nkmb; k++) for (int i = is-1; i <= ie+1; i++) if (G->mask2dT[H2(i,j)] > 0.0) {
      if (h[H3(i,j,k)] > h_exclude) {
        const int ns = num_srt[H2(i,j)]++;
        k0_srt[H2(i,j)*nz + ns] = k;
        rho_srt[H2(i,j)*nz + ns] = rho_coord[hpl*(k-1) + H2(i,j)];
        h_srt[H2(i,j)*nz + ns] = h[H3(i,j,k)];
      }
    }
    for (int k = nkmb+1; k <= PEmax_kRho; k++) for (int i = is-1; i <= ie+1; i++) if (G->mask2dT[H2(i,j)] > 0.0) {
      if ((k <= k_end_srt[H2(i,j)]) && (h[H3(i,j,k)] > h_exclude)) {
        const int ns = num_srt[H2(i,j)]++;
        k0_srt[H2(i,j)*nz + ns] = k;
        rho_srt[H2(i,j)*nz + ns] = RLAY(k);
        h_srt[H2(i,j)*nz + ns] = h[H3(i,j,k)];
      }
    }
  }
  for (int j = js-1; j <= je+1; j++) for (int i = is-1; i <= ie+1; i++) {
    int *k0 = k0_srt + H2(i,j)*nz; double *rs = rho_srt + H2(i,j)*nz, *hs = h_srt + H2(i,j)*nz;
    for (int k = 2; k <= num_srt[H2(i,j)]; k++) if (rs[k-1] < rs[k-2]) {
      for (int k2 = k; k2 >= 2; k2--) { if (rs[k2-1] >= rs[k2-2]) break;
        const int itmp = k0[k2-2]; k0[k2-2] = k0[k2-1]; k0[k2-1] = itmp;
        double tmp = rs[k2-2]; rs[k2-2] = rs[k2-1]; rs[k2-1] = tmp;
        tmp = hs[k2-2]; hs[k2-2] = hs[k2-1]; hs[k2-1] = tmp;
      }
    }
  }
  /* the pairings of every face :930-1251 */
  for (int j = js; j <= je; j++) for (int I = is-1; I <= ie; I++) if (G->mask2dCu[U2(I,j)] > 0.0) {
    const int i = I; const size_t cL = H2(i,j), cR = H2(i+1,j), f = U2(I,j);
    nPu[f] = epi_pairings(num_srt[cL], k0_srt + cL*nz, rho_srt + cL*nz, h_srt + cL*nz, num_srt[cR], k0_srt + cR*nz, rho_srt + cR*nz, h_srt + cR*nz,
                          nkmb, nz, PK(ku,f,0), PK(ku,f,1), PK(ku,f,2), PK(ku,f,3), PK(wu,f,0), PK(wu,f,1), PK(wu,f,2), PK(wu,f,3));
  }
  for (int J = js-1; J <= je; J++) for (int i = is; i <= ie; i++) if (G->mask2dCv[V2(i,J)] > 0.0) {
    const int j = J; const size_t cL = H2(i,j), cR = H2(i,j+1), f = V2(i,J);
    nPv[f] = epi_pairings(num_srt[cL], k0_srt + cL*nz, rho_srt + cL*nz, h_srt + cL*nz, num_srt[cR], k0_srt + cR*nz, rho_srt + cR*nz, h_srt + cR*nz,
                          nkmb, nz, PK(kv,f,0), PK(kv,f,1), PK(kv,f,2), PK(kv,f,3), PK(wv,f,0), PK(wv,f,1), PK(wv,f,2), PK(wv,f,3));
  }

  /* the tracer-specific calculations :1255-1604 */
  const int old_answers = EP->answer_date <= 20240330;
  for (int itt = 1; itt <= max_itt; itt++) {
    if (itt > 1) { for (int m = 0; m < ntr; m++) orc_halo_update(G, Tr[m], MOM6HIP_POS_H, nz); (*halo_updates)++; }
    for (int m = 0; m < ntr; m++) {
      double *T = Tr[m];
      if (old_answers) memset(tr_flux_conv, 0, sizeof(double)*hpl*nz);
      else {
        memset(tr_flux_N, 0, sizeof(double)*hpl*nz); memset(tr_flux_S, 0, sizeof(double)*hpl*nz);
        memset(tr_flux_E, 0, sizeof(double)*hpl*nz); memset(tr_flux_W, 0, sizeof(double)*hpl*nz);
      }
      memset(Tr_flux_3d, 0, sizeof(double)*vpl*np2);
      memset(Tr_adj_vert_R, 0, sizeof(double)*vpl*np2); memset(Tr_adj_vert_L, 0, sizeof(double)*vpl*np2);
      for (int dir = 0; dir < 2; dir++)
      for (int jj = (dir ? js-1 : js); jj <= je; jj++) for (int ii = (dir ? is : is-1); ii <= ie; ii++) {
        const size_t f = dir ? V2(ii,jj) : U2(ii,jj);
        if (!((dir ? G->mask2dCv[f] : G->mask2dCu[f]) > 0.0)) continue;
        const int i = ii, j = jj, iR = dir ? ii : ii+1, jR = dir ? jj+1 : jj;      /* the left and right columns */
        const int nP = dir ? nPv[f] : nPu[f];
        const int *k0b_L = PK(dir ? kv : ku, f, 0), *k0a_L = PK(dir ? kv : ku, f, 1), *k0b_R = PK(dir ? kv : ku, f, 2), *k0a_R = PK(dir ? kv : ku, f, 3);
        const double *deep_wt_L = PK(dir ? wv : wu, f, 0), *deep_wt_R = PK(dir ? wv : wu, f, 1), *hP_L = PK(dir ? wv : wu, f, 2), *hP_R = PK(dir ? wv : wu, f, 3);
        double Tr_min_face = 0.0, Tr_max_face = 0.0, Tr_La = 0.0, Tr_Lb, Tr_Ra = 0.0, Tr_Rb;
        int kLa, kLb, kRa, kRb;
        /* the acceptable range of concentrations around this face :1275-1312 (u), :1427-1462 (v) */
        if (nP >= 1) {
          Tr_min_face = min2(T[H3(i,j,1)], T[H3(iR,jR,1)]);
          Tr_max_face = max2(T[H3(i,j,1)], T[H3(iR,jR,1)]);
          for (int k = 2; k <= nkmb; k++) {
            Tr_min_face = min3(Tr_min_face, T[H3(i,j,k)], T[H3(iR,jR,k)]);
            Tr_max_face = max3(Tr_max_face, T[H3(i,j,k)], T[H3(iR,jR,k)]);
          }
          /* the next two layers denser than the densest buffer layer */
          kLa = nkmb+1; if (max_kRho[H2(i,j)] < nz+1) kLa = max_kRho[H2(i,j)];
          kLb = kLa; if (max_kRho[H2(i,j)] < nz) kLb = max_kRho[H2(i,j)]+1;
          kRa = nkmb+1; if (max_kRho[H2(iR,jR)] < nz+1) kRa = max_kRho[H2(iR,jR)];
          kRb = kRa; if (max_kRho[H2(iR,jR)] < nz) kRb = max_kRho[H2(iR,jR)]+1;
          Tr_La = Tr_min_face; Tr_Lb = Tr_La; Tr_Ra = Tr_La; Tr_Rb = Tr_La;
          if (h[H3(i,j,kLa)] > h_exclude) Tr_La = T[H3(i,j,kLa)];
          if (old_answers && EP->limit_bug) {
            if (h[H3(i,j,kLb)] > h_exclude) Tr_La = T[H3(i,j,kLb)];
          } else {
            if (h[H3(i,j,kLb)] > h_exclude) Tr_Lb = T[H3(i,j,kLb)];
          }
          if (h[H3(iR,jR,kRa)] > h_exclude) Tr_Ra = T[H3(iR,jR,kRa)];
          if (h[H3(iR,jR,kRb)] > h_exclude) Tr_Rb = T[H3(iR,jR,kRb)];
          Tr_min_face = min5(Tr_min_face, Tr_La, Tr_Lb, Tr_Ra, Tr_Rb);
          Tr_max_face = max5(Tr_max_face, Tr_La, Tr_Lb, Tr_Ra, Tr_Rb);
          /* all points in diffusive pairings at this face */
          for (int k = 0; k < nP; k++) {
            Tr_Lb = T[H3(i,j,k0b_L[k])]; Tr_Rb = T[H3(iR,jR,k0b_R[k])];
            Tr_La = Tr_Lb; Tr_Ra = Tr_Rb;
            if (deep_wt_L[k] < 1.0) Tr_La = T[H3(i,j,k0a_L[k])];
            if (deep_wt_R[k] < 1.0) Tr_Ra = T[H3(iR,jR,k0a_R[k])];
            Tr_min_face = min5(Tr_min_face, Tr_La, Tr_Lb, Tr_Ra, Tr_Rb);
            Tr_max_face = max5(Tr_max_face, Tr_La, Tr_Lb, Tr_Ra, Tr_Rb);
          }
        }
        for (int k = 0; k < nP; k++) {                             /* :1314-1412 (u), :1464-1538 (v) */
          double Tr_av_L, Tr_av_R, wt_b, wt_a, Tr_flux, Tr_adj_vert, vol;
          kLb = k0b_L[k]; Tr_Lb = T[H3(i,j,kLb)]; Tr_av_L = Tr_Lb; kLa = kLb;
          if (deep_wt_L[k] < 1.0) {
            kLa = k0a_L[k]; Tr_La = T[H3(i,j,kLa)];
            wt_b = deep_wt_L[k];
            Tr_av_L = wt_b*Tr_Lb + (1.0-wt_b)*Tr_La;
          }
          kRb = k0b_R[k]; Tr_Rb = T[H3(iR,jR,kRb)]; Tr_av_R = Tr_Rb; kRa = kRb;
          if (deep_wt_R[k] < 1.0) {
            kRa = k0a_R[k]; Tr_Ra = T[H3(iR,jR,kRa)];
            wt_b = deep_wt_R[k];
            Tr_av_R = wt_b*Tr_Rb + (1.0-wt_b)*Tr_Ra;
          }
          const double h_L = hP_L[k], h_R = hP_R[k];
          if (!dir && old_answers)
            Tr_flux = I_maxitt * khdt_epi_x[f] * (Tr_av_L - Tr_av_R) * ((2.0 * h_L * h_R) / (h_L + h_R));
          else
            Tr_flux = I_maxitt * ((2.0 * h_L * h_R) / (h_L + h_R)) * (dir ? khdt_epi_y[f] : khdt_epi_x[f]) * (Tr_av_L - Tr_av_R);
          if (!dir) {
            if (deep_wt_L[k] >= 1.0) {
              if (old_answers) tr_flux_conv[H3(i,j,kLb)] = tr_flux_conv[H3(i,j,kLb)] - Tr_flux;
              else tr_flux_E[H3(i,j,kLb)] = tr_flux_E[H3(i,j,kLb)] + Tr_flux;
            } else {
              wt_b = deep_wt_L[k]; wt_a = 1.0 - wt_b;
              vol = hP_L[k] * G->areaT[H2(i,j)];
              Tr_adj_vert = epi_adj_left(Tr_flux, Tr_La, Tr_Lb, vol, wt_a, wt_b, Tr_min_face, Tr_max_face);
              if (old_answers) {
                tr_flux_conv[H3(i,j,kLa)] = tr_flux_conv[H3(i,j,kLa)] - (wt_a*Tr_flux + Tr_adj_vert);
                tr_flux_conv[H3(i,j,kLb)] = tr_flux_conv[H3(i,j,kLb)] - (wt_b*Tr_flux - Tr_adj_vert);
              } else {
                tr_flux_E[H3(i,j,kLa)] = tr_flux_E[H3(i,j,kLa)] + (wt_a*Tr_flux + Tr_adj_vert);
                tr_flux_E[H3(i,j,kLb)] = tr_flux_E[H3(i,j,kLb)] + (wt_b*Tr_flux - Tr_adj_vert);
              }
            }
            if (deep_wt_R[k] >= 1.0) {
              if (old_answers) tr_flux_conv[H3(iR,jR,kRb)] = tr_flux_conv[H3(iR,jR,kRb)] + Tr_flux;
              else tr_flux_W[H3(iR,jR,kRb)] = tr_flux_W[H3(iR,jR,kRb)] + Tr_flux;
            } else {
              wt_b = deep_wt_R[k]; wt_a = 1.0 - wt_b;
              vol = hP_R[k] * G->areaT[H2(iR,jR)];
              Tr_adj_vert = epi_adj_right(Tr_flux, Tr_Ra, Tr_Rb, vol, wt_a, wt_b, Tr_min_face, Tr_max_face);
              if (old_answers) {
                tr_flux_conv[H3(iR,jR,kRa)] = tr_flux_conv[H3(iR,jR,kRa)] + (wt_a*Tr_flux - Tr_adj_vert);
                tr_flux_conv[H3(iR,jR,kRb)] = tr_flux_conv[H3(iR,jR,kRb)] + (wt_b*Tr_flux + Tr_adj_vert);
              } else {
                tr_flux_W[H3(iR,jR,kRa)] = tr_flux_W[H3(iR,jR,kRa)] + (wt_a*Tr_flux - Tr_adj_vert);
                tr_flux_W[H3(iR,jR,kRb)] = tr_flux_W[H3(iR,jR,kRb)] + (wt_b*Tr_flux + Tr_adj_vert);
              }
            }
          } else {
            Tr_flux_3d[f*np2 + k] = Tr_flux;
            if (deep_wt_L[k] < 1.0) {
              wt_b = deep_wt_L[k]; wt_a = 1.0 - wt_b;
              vol = hP_L[k] * G->areaT[H2(i,j)];
              Tr_adj_vert_L[f*np2 + k] = epi_adj_left(Tr_flux, Tr_La, Tr_Lb, vol, wt_a, wt_b, Tr_min_face, Tr_max_face);
            }
            if (deep_wt_R[k] < 1.0) {
              wt_b = deep_wt_R[k]; wt_a = 1.0 - wt_b;
              vol = hP_R[k] * G->areaT[H2(iR,jR)];
              Tr_adj_vert_R[f*np2 + k] = epi_adj_right(Tr_flux, Tr_Ra, Tr_Rb, vol, wt_a, wt_b, Tr_min_face, Tr_max_face);
            }
          }
        }
      }
      /* the meridional fluxes into the cells :1541-1586 */
      for (int J = js-1; J <= je; J++) for (int i = is; i <= ie; i++) if (G->mask2dCv[V2(i,J)] > 0.0) {
        const int j = J; const size_t f = V2(i,J);
        const int *k0b_L = PK(kv,f,0), *k0a_L = PK(kv,f,1), *k0b_R = PK(kv,f,2), *k0a_R = PK(kv,f,3);
        const double *deep_wt_L = PK(wv,f,0), *deep_wt_R = PK(wv,f,1);
        for (int k = 0; k < nPv[f]; k++) {
          const int kLb = k0b_L[k], kRb = k0b_R[k];
          const double F3 = Tr_flux_3d[f*np2 + k];
          if (old_answers) {
            if (deep_wt_L[k] >= 1.0) tr_flux_conv[H3(i,j,kLb)] = tr_flux_conv[H3(i,j,kLb)] - F3;
            else {
              const int kLa = k0a_L[k]; const double wt_b = deep_wt_L[k], wt_a = 1.0 - wt_b;
              tr_flux_conv[H3(i,j,kLa)] = tr_flux_conv[H3(i,j,kLa)] - (wt_a*F3 + Tr_adj_vert_L[f*np2 + k]);
              tr_flux_conv[H3(i,j,kLb)] = tr_flux_conv[H3(i,j,kLb)] - (wt_b*F3 - Tr_adj_vert_L[f*np2 + k]);
            }
            if (deep_wt_R[k] >= 1.0) tr_flux_conv[H3(i,j+1,kRb)] = tr_flux_conv[H3(i,j+1,kRb)] + F3;
            else {
              const int kRa = k0a_R[k]; const double wt_b = deep_wt_R[k], wt_a = 1.0 - wt_b;
              tr_flux_conv[H3(i,j+1,kRa)] = tr_flux_conv[H3(i,j+1,kRa)] + (wt_a*F3 - Tr_adj_vert_R[f*np2 + k]);
              tr_flux_conv[H3(i,j+1,kRb)] = tr_flux_conv[H3(i,j+1,kRb)] + (wt_b*F3 + Tr_adj_vert_R[f*np2 + k]);
            }
          } else {
            if (deep_wt_L[k] >= 1.0) tr_flux_N[H3(i,j,kLb)] = tr_flux_N[H3(i,j,kLb)] + F3;
            else {
              const int kLa = k0a_L[k]; const double wt_b = deep_wt_L[k], wt_a = 1.0 - wt_b;
              tr_flux_N[H3(i,j,kLa)] = tr_flux_N[H3(i,j,kLa)] + (wt_a*F3 + Tr_adj_vert_L[f*np2 + k]);
              tr_flux_N[H3(i,j,kLb)] = tr_flux_N[H3(i,j,kLb)] + (wt_b*F3 - Tr_adj_vert_L[f*np2 + k]);
            }
            if (deep_wt_R[k] >= 1.0) tr_flux_S[H3(i,j+1,kRb)] = tr_flux_S[H3(i,j+1,kRb)] + F3;
            else {
              const int kRa = k0a_R[k]; const double wt_b = deep_wt_R[k], wt_a = 1.0 - wt_b;
              tr_flux_S[H3(i,j+1,kRa)] = tr_flux_S[H3(i,j+1,kRa)] + (wt_a*F3 - Tr_adj_vert_R[f*np2 + k]);
              tr_flux_S[H3(i,j+1,kRb)] = tr_flux_S[H3(i,j+1,kRb)] + (wt_b*F3 + Tr_adj_vert_R[f*np2 + k]);
            }
          }
        }
      }
      if (!old_answers)                                            /* :1588-1594 */
        for (int k = 1; k <= PEmax_kRho; k++) for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++)
          tr_flux_conv[H3(i,j,k)] = ((tr_flux_W[H3(i,j,k)] - tr_flux_E[H3(i,j,k)]) + (tr_flux_S[H3(i,j,k)] - tr_flux_N[H3(i,j,k)]));
      for (int k = 1; k <= PEmax_kRho; k++) for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++)      /* :1596-1601 */
        if ((G->mask2dT[H2(i,j)] > 0.0) && (h[H3(i,j,k)] > 0.0))
          T[H3(i,j,k)] = T[H3(i,j,k)] + tr_flux_conv[H3(i,j,k)] / (h[H3(i,j,k)]*G->areaT[H2(i,j)]);
      if (conc_underflow && conc_underflow[m] > 0.0)               /* :1604-1609 */
        for (int k = 1; k <= nz; k++) for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++)
          if (fabs(T[H3(i,j,k)]) < conc_underflow[m]) T[H3(i,j,k)] = 0.0;
    }
  }
  free(Rml_max); free(rho_coord); free(num_srt); free(k_end_srt); free(max_kRho); free(rho_srt); free(h_srt); free(k0_srt);
  free(nPu); free(nPv); free(ku); free(kv); free(wu); free(wv); free(tr_flux_conv); free(tr_flux_N); free(tr_flux_S);
  free(tr_flux_E); free(tr_flux_W); free(Tr_flux_3d); free(Tr_adj_vert_L); free(Tr_adj_vert_R);
#undef H2
#undef U2
#undef V2
#undef H3
#undef PK
#undef RLAY
  return 0;
}
