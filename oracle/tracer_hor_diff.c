/* oracle/tracer_hor_diff.c -- TEST INFRASTRUCTURE: a C restatement of the along-layer branch of tracer_hordiff
 * (src/tracer/MOM_tracer_hor_diff.F90:119-680) with KHTR or the VarMix / MEKE diffusivities, and the call of the neutral-diffusion branch
 * (neutral_diffusion.c); no boundary diffusion, no epipycnal mixed-layer diffusion.  The reference holds no known-answer vectors for this routine: parity unpinned; the
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

int orc_tracer_hordiff_neutral(const mom6hip_grid_t *G, const mom6hip_tracer_hor_diff_cs_t *CS,
                               const mom6hip_neutral_diffusion_cs_t *ND, const mom6hip_hordiff_fields_t *F, const double *h,
                               const mom6hip_eos_t *eos, const double *p_surf, double dt, double *const *tr,
                               const double *conc_underflow, int ntr, int idx_T, int idx_S, mom6hip_hordiff_stats_t *stats)
{
  for (int q = 1; q < 8; q++) if (CS->unsupported[q]) return 2;
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
      const double scale = I_numitts;
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
  if (stats) { stats->num_itts = num_itts; stats->halo_updates = halo_updates; stats->max_CFL = max_CFL; }
#undef H2
#undef U2
#undef V2
  free(khdt_x); free(khdt_y); free(Coef_x); free(Coef_y); free(Ihdxdy); free(dTr);
  return rc;
}
