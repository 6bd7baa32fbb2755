/*
 * thickness_diffuse.c -- CPU restatement of thickness_diffuse (TEST INFRASTRUCTURE, see mom6_oracle.h).
 *
 * Reference: src/parameterizations/lateral/MOM_thickness_diffuse.F90
 *   thickness_diffuse :133-629 (the diffusivities :182-262 / :340-420, the transports into uhtr / vhtr and the thickness
 *   tendency :607-620), thickness_diffuse_full :634-1670; vert_fill_TS src/core/MOM_isopycnal_slopes.F90:541-629;
 *   find_eta src/core/MOM_interface_heights.F90:48-112 (Boussinesq, no eta_bt).
 * Restated branch: Boussinesq; with an equation of state (calculate_density_derivs at the interface pressure) or without
 * (layer densities GV%Rlay); the slopes of this routine or stored ones (slope_x, slope_y); KHTH + MEKE%Kh + Visbeck term, resolution
 * function, KHTH_MIN / KHTH_MAX / KHTH_MAX_CFL; the work into MEKE%GM_src.  No FGNV streamfunction, no detangling, no
 * interface-height diffusivity, no Stanley terms, no MEKE_GEOMETRIC, no GM_src_alt, no ebt structure, no tv%p_surf.
 * PARITY UNPINNED: the reference holds no known-answer vectors for this module; invariants in tests/test_thickness_diffuse.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mom6_oracle.h"

static inline double min2(double a, double b) { return a < b ? a : b; }
static inline double max2(double a, double b) { return a > b ? a : b; }

#define H2(i,j) ORC_H2(G,i,j)
#define H3(i,j,k) ORC_H3(G,i,j,k)
#define U2(i,j) ORC_U2(G,i,j)
#define V2(i,j) ORC_V2(G,i,j)
#define U3(i,j,k) ORC_U3(G,i,j,k)
#define V3(i,j,k) ORC_V3(G,i,j,k)
/* interface arrays at h points: (nih, njh, nk+1) */
#define E3(i,j,K) (ORC_H2(G,i,j) + (long)ORC_NIH(G) * ORC_NJH(G) * ((K)-1))

/* vert_fill_TS :541-629 with larger_h_denom, for the cells is-halo..ie+halo */
static void vert_fill_TS(const mom6hip_grid_t *G, const double *h, const double *T_in, const double *S_in, double kappa_dt,
                         double *T_f, double *S_f, int halo) {
  const int nz = G->nk;
  const double h_neglect = G->H_subroundoff;
  const double kap_dt_x2 = (2.0 * kappa_dt) * (1.0 * G->Z_to_H);      /* US%Z_to_m*GV%m_to_H */
  const double h0 = 1.0e-16 * sqrt(0.5 * kap_dt_x2);
  const int is = G->isc - halo, ie = G->iec + halo, js = G->jsc - halo, je = G->jec + halo;
  if (kap_dt_x2 <= 0.0 || nz < 2) {
    for (int k = 1; k <= nz; k++) for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
      T_f[H3(i,j,k)] = T_in[H3(i,j,k)]; S_f[H3(i,j,k)] = S_in[H3(i,j,k)];
    }
    return;
  }
  double *ent = (double *)malloc(sizeof(double) * (nz + 2)), *c1 = (double *)malloc(sizeof(double) * (nz + 2));
  for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
    double b1, d1, h_tr;
    ent[2] = kap_dt_x2 / ((h[H3(i,j,1)] + h[H3(i,j,2)]) + h0);
    h_tr = h[H3(i,j,1)] + h_neglect;
    b1 = 1.0 / (h_tr + ent[2]);
    d1 = b1 * h_tr;
    T_f[H3(i,j,1)] = (b1 * h_tr) * T_in[H3(i,j,1)];
    S_f[H3(i,j,1)] = (b1 * h_tr) * S_in[H3(i,j,1)];
    for (int k = 2; k <= nz - 1; k++) {
      ent[k + 1] = kap_dt_x2 / ((h[H3(i,j,k)] + h[H3(i,j,k+1)]) + h0);
      h_tr = h[H3(i,j,k)] + h_neglect;
      c1[k] = ent[k] * b1;
      b1 = 1.0 / ((h_tr + d1 * ent[k]) + ent[k + 1]);
      d1 = b1 * (h_tr + d1 * ent[k]);
      T_f[H3(i,j,k)] = b1 * (h_tr * T_in[H3(i,j,k)] + ent[k] * T_f[H3(i,j,k-1)]);
      S_f[H3(i,j,k)] = b1 * (h_tr * S_in[H3(i,j,k)] + ent[k] * S_f[H3(i,j,k-1)]);
    }
    c1[nz] = ent[nz] * b1;
    h_tr = h[H3(i,j,nz)] + h_neglect;
    b1 = 1.0 / (h_tr + d1 * ent[nz]);
    T_f[H3(i,j,nz)] = b1 * (h_tr * T_in[H3(i,j,nz)] + ent[nz] * T_f[H3(i,j,nz-1)]);
    S_f[H3(i,j,nz)] = b1 * (h_tr * S_in[H3(i,j,nz)] + ent[nz] * S_f[H3(i,j,nz-1)]);
    for (int k = nz - 1; k >= 1; k--) {
      T_f[H3(i,j,k)] = T_f[H3(i,j,k)] + c1[k + 1] * T_f[H3(i,j,k+1)];
      S_f[H3(i,j,k)] = S_f[H3(i,j,k)] + c1[k + 1] * S_f[H3(i,j,k+1)];
    }
  }
  free(ent); free(c1);
}

typedef struct {
  const mom6hip_grid_t *G;
  const mom6hip_thickness_diffuse_cs_t *CS;
  const mom6hip_eos_t *EOS;
  const double *h, *e, *T, *S, *pres, *h_avail, *h_avail_rsum, *h_frac;
  double dt;
} tdf_t;

/* One velocity column of thickness_diffuse_full: dir 0 the u face (I,j) between the cells (i,j) and (i+1,j) (:812-1209), dir 1
 * the v face (i,J) between (i,j) and (i,j+1) (:1211-1515); then the top layer (:1517-1590).  KH: the diffusivity of the face (the
 * same at every interface in the provided branch).  Returns the work of the column. */
static double face_column(const tdf_t *A, int dir, int i, int j, double KH, const double *slope_st, double *hD) {
  const mom6hip_grid_t *G = A->G;
  const mom6hip_thickness_diffuse_cs_t *CS = A->CS;
  const int nz = G->nk;
  const int i2 = dir ? i : i + 1, j2 = dir ? j + 1 : j;      /* the cell on the other side of the face */
  const double *h = A->h, *e = A->e, *T = A->T, *S = A->S;
  const int use_EOS = A->EOS != NULL;
  const int find_work = CS->MEKE_GM_src != NULL;
  const int present_slope = slope_st != NULL;
  const int nk_linear = CS->nkml > 1 ? CS->nkml : 1;
  const double I_slope_max2 = 1.0 / (CS->slope_max * CS->slope_max);
  const double h_neglect = G->H_subroundoff, h_neglect2 = h_neglect * h_neglect;
  const double dz_neglect = G->dZ_subroundoff;
  const double IdL = dir ? G->IdyCv[V2(i,j)] : G->IdxCu[U2(i,j)];
  const double dLf = dir ? G->dx_Cv[V2(i,j)] : G->dy_Cu[U2(i,j)];
  const double OBCmask = dir ? G->mask2dCv[V2(i,j)] : G->mask2dCu[U2(i,j)];      /* G%OBCmaskCu = mask2dCu without OBCs */
  const double G_scale = G->g_Earth * G->H_to_Z;
  const long fpl = dir ? (long)ORC_NIH(G) * (ORC_NJH(G) + 1) : (long)(ORC_NIH(G) + 1) * ORC_NJH(G);
  const long f2 = dir ? V2(i,j) : U2(i,j);
  double htot = 0.0, Work = 0.0;
  double drdiA = 0.0, drdiB = 0.0, drdkL = 0.0, drdkR = 0.0;
  /* what the first sweep over the interfaces (:872-1101) hands the second (:1126-1209); with KHTH_USE_FGNV_STREAMFUNCTION the elliptic solve
   * of Ferrari et al. (2010) sits between them (:1105-1124) */
  const int use_FGNV = CS->use_FGNV_streamfn != 0;
  double *W = (double *)calloc((size_t)6 * (nz + 2), sizeof(double));
  double *Sfn_u = W, *s2R = W + (nz + 2), *dkDe = W + 2 * (nz + 2), *di_k = W + 3 * (nz + 2), *dzN2 = W + 4 * (nz + 2), *c2_dz = W + 5 * (nz + 2);
  const double G_rho0 = G->g_Earth / G->Rho0, N2_floor = CS->N2_floor, dz_neglect2 = dz_neglect * dz_neglect;
#define HL(k) h[H3(i,j,k)]
#define HR(k) h[H3(i2,j2,k)]
#define EL(K) e[E3(i,j,K)]
#define ER(K) e[E3(i2,j2,K)]
#define DZL(k) (G->H_to_Z * HL(k))      /* thickness_to_dz, Boussinesq (MOM_interface_heights.F90:780) */
#define DZR(k) (G->H_to_Z * HR(k))
  for (int K = nz; K >= 2; K--) {
    const int k = K;
    double Sfn_unlim, slope2_Ratio = 0.0, drdi_k = 0.0, drdkDe = 0.0;
    if (find_work && !use_EOS) {      /* :824-828 */
      drdiA = 0.0; drdiB = 0.0;
      drdkL = CS->Rlay[k - 1] - CS->Rlay[k - 2]; drdkR = drdkL;
    }
    const int calc_derivatives = use_EOS && (k >= nk_linear) && (find_work || !present_slope || use_FGNV);
    if (calc_derivatives) {      /* :833-842, :855-865 */
      const double pres_u = 0.5 * (A->pres[E3(i,j,K)] + A->pres[E3(i2,j2,K)]);
      const double T_u = 0.25 * ((T[H3(i,j,k)] + T[H3(i2,j2,k)]) + (T[H3(i,j,k-1)] + T[H3(i2,j2,k-1)]));
      const double S_u = 0.25 * ((S[H3(i,j,k)] + S[H3(i2,j2,k)]) + (S[H3(i,j,k-1)] + S[H3(i2,j2,k-1)]));
      double drho_dT, drho_dS;
      orc_eos_density_derivs(A->EOS, T_u, S_u, pres_u, &drho_dT, &drho_dS);
      drdiA = drho_dT * (T[H3(i2,j2,k-1)] - T[H3(i,j,k-1)]) + drho_dS * (S[H3(i2,j2,k-1)] - S[H3(i,j,k-1)]);
      drdiB = drho_dT * (T[H3(i2,j2,k)] - T[H3(i,j,k)]) + drho_dS * (S[H3(i2,j2,k)] - S[H3(i,j,k)]);
      drdkL = (drho_dT * (T[H3(i,j,k)] - T[H3(i,j,k-1)]) + drho_dS * (S[H3(i,j,k)] - S[H3(i,j,k-1)]));
      drdkR = (drho_dT * (T[H3(i2,j2,k)] - T[H3(i2,j2,k-1)]) + drho_dS * (S[H3(i2,j2,k)] - S[H3(i2,j2,k-1)]));
      drdkDe = drdkR * ER(K) - drdkL * EL(K);
    } else if (find_work) {
      drdkDe = drdkR * ER(K) - drdkL * EL(K);
    }
    if (find_work) drdi_k = drdiB;
    if (k > nk_linear) {
      if (use_EOS) {
        double hg2A = 0.0, hg2B = 0.0, haA = 0.0, haB = 0.0, drdz = 0.0, Slope;
        if (use_FGNV || find_work || !present_slope) {      /* :982-1024 (Boussinesq) */
          const double hg2L = HL(k-1) * HL(k) + h_neglect2;
          const double hg2R = HR(k-1) * HR(k) + h_neglect2;
          const double haL = 0.5 * (HL(k-1) + HL(k)) + h_neglect;
          const double haR = 0.5 * (HR(k-1) + HR(k)) + h_neglect;
          const double dzaL = haL * G->H_to_Z, dzaR = haR * G->H_to_Z;
          const double wtL = hg2L * (haR * dzaR), wtR = hg2R * (haL * dzaL);
          drdz = (wtL * drdkL + wtR * drdkR) / (dzaL * wtL + dzaR * wtR);
          hg2A = HL(k-1) * HR(k-1) + h_neglect2;
          hg2B = HL(k) * HR(k) + h_neglect2;
          haA = 0.5 * (HL(k-1) + HR(k-1)) + h_neglect;
          haB = 0.5 * (HL(k) + HR(k)) + h_neglect;
          const double N2_unlim = drdz * G_rho0;
          const double dzg2A = DZL(k-1) * DZR(k-1) + dz_neglect2;
          const double dzg2B = DZL(k) * DZR(k) + dz_neglect2;
          const double dzaA = 0.5 * (DZL(k-1) + DZR(k-1)) + dz_neglect;
          const double dzaB = 0.5 * (DZL(k) + DZR(k)) + dz_neglect;
          dzN2[K] = (0.5 * ( dzg2A / dzaA + dzg2B / dzaB )) * max2(N2_unlim, N2_floor);      /* :1021 */
        }
        if (present_slope) {      /* :1027-1029 */
          Slope = slope_st[f2 + fpl * (K - 1)];
          slope2_Ratio = (Slope * Slope) * I_slope_max2;
        } else {                  /* :1030-1044 */
          const double wtA = hg2A * haB, wtB = hg2B * haA;
          const double drdx = ((wtA * drdiA + wtB * drdiB) / (wtA + wtB) - drdz * (EL(K) - ER(K))) * IdL;
          const double mag_grad2 = (1.0 * drdx) * (1.0 * drdx) + drdz * drdz;      /* US%Z_to_L = 1 */
          if (mag_grad2 > 0.0) {
            Slope = drdx / sqrt(mag_grad2);
            slope2_Ratio = (Slope * Slope) * I_slope_max2;
          } else {
            Slope = 0.0;
            slope2_Ratio = 1.0e20;
          }
        }
        /* :1047-1051 with int_slope = 0 */
        Slope = (1.0 - 0.0) * Slope + 0.0 * ((ER(K) - EL(K)) * IdL);
        slope2_Ratio = (1.0 - 0.0) * slope2_Ratio;
        Sfn_unlim = -(KH * dLf) * Slope;      /* :1064 */
        if (Sfn_unlim > 0.0) {                 /* :1067-1084 */
          if (EL(K) < ER(nz + 1)) Sfn_unlim = 0.0;
          else if (ER(nz + 1) > EL(K + 1)) Sfn_unlim = Sfn_unlim * ((EL(K) - ER(nz + 1)) / ((EL(K) - EL(K + 1)) + dz_neglect));
        } else {
          if (ER(K) < EL(nz + 1)) Sfn_unlim = 0.0;
          else if (EL(nz + 1) > ER(K + 1)) Sfn_unlim = Sfn_unlim * ((ER(K) - EL(nz + 1)) / ((ER(K) - ER(K + 1)) + dz_neglect));
        }
      } else {      /* :1086-1095 */
        double Slope;
        if (present_slope) Slope = slope_st[f2 + fpl * (K - 1)];
        else Slope = ((EL(K) - ER(K)) * IdL) * OBCmask;
        Sfn_unlim = ((KH * dLf) * Slope);
        if (use_FGNV) dzN2[K] = CS->g_prime[K - 1];      /* GV%g_prime(K) */
      }
    } else {
      dzN2[K] = N2_floor * dz_neglect;      /* :1098 */
      Sfn_unlim = 0.;
    }
    Sfn_u[K] = Sfn_unlim; s2R[K] = slope2_Ratio; dkDe[K] = drdkDe; di_k[k] = drdi_k;
  }
  if (use_FGNV) {      /* :1105-1124: the streamfunction of Ferrari et al. (2010): streamfn_solver :1673-1707 */
    if (OBCmask > 0.) {
      const double *cg1 = CS->cg1;
      for (int k = 1; k <= nz; k++) {
        const double dz_harm = max2( dz_neglect, 2. * DZL(k) * DZR(k) / ( ( DZL(k) + DZR(k) ) + dz_neglect ) );
        const double cg = 0.5*( cg1[H2(i,j)] + cg1[H2(i2,j2)] );
        c2_dz[k] = CS->FGNV_scale * ( cg * cg ) / dz_harm;
      }
      for (int K = 2; K <= nz; K++) Sfn_u[K] = (1. + CS->FGNV_scale) * Sfn_u[K];
      double *c1 = (double *)calloc((size_t)nz + 2, sizeof(double));
      Sfn_u[1] = 0.;
      double b_denom = dzN2[2] + c2_dz[1];
      double beta = 1.0 / ( b_denom + c2_dz[2] );
      double d1 = beta * b_denom;
      Sfn_u[2] = ( beta * dzN2[2] )*Sfn_u[2];
      for (int K = 3; K <= nz; K++) {
        c1[K-1] = beta * c2_dz[K-1];
        b_denom = dzN2[K] + d1*c2_dz[K-1];
        beta = 1.0 / (b_denom + c2_dz[K]);
        d1 = beta * b_denom;
        Sfn_u[K] = beta * (dzN2[K]*Sfn_u[K] + c2_dz[K-1]*Sfn_u[K-1]);
      }
      c1[nz] = beta * c2_dz[nz];
      Sfn_u[nz+1] = 0.;
      for (int K = nz; K >= 2; K--) Sfn_u[K] = Sfn_u[K] + c1[K]*Sfn_u[K+1];
      free(c1);
    } else {
      for (int K = 2; K <= nz; K++) Sfn_u[K] = 0.;
    }
  }
  for (int K = nz; K >= 2; K--) {
    const int k = K;
    const double Sfn_unlim = Sfn_u[K], slope2_Ratio = s2R[K], drdkDe = dkDe[K], drdi_k = di_k[k];
    /* ---- the transport of layer k :1148-1209 */
    const double Z_to_H = G->Z_to_H;
    double hDk;
    if (k > nk_linear) {
      double Sfn_est;
      if (use_EOS) {
        double Sfn_safe;
        if (htot <= 0.0) Sfn_safe = htot * (1.0 - A->h_frac[H3(i,j,k)]);
        else Sfn_safe = htot * (1.0 - A->h_frac[H3(i2,j2,k)]);
        Sfn_est = (Z_to_H * Sfn_unlim + slope2_Ratio * Sfn_safe) / (1.0 + slope2_Ratio);
      } else {
        Sfn_est = Z_to_H * Sfn_unlim;
      }
      const double Sfn_in_H = min2(max2(Sfn_est, -A->h_avail_rsum[E3(i,j,K)]), A->h_avail_rsum[E3(i2,j2,K)]);
      hDk = max2(min2((Sfn_in_H - htot), A->h_avail[H3(i,j,k)]), -A->h_avail[H3(i2,j2,k)]);
    } else {
      if (htot <= 0.0) hDk = -htot * A->h_frac[H3(i,j,k)];
      else hDk = -htot * A->h_frac[H3(i2,j2,k)];
    }
    hD[f2 + fpl * (k - 1)] = hDk;
    htot = htot + hDk;
    if (find_work)      /* :1196-1209 */
      Work = Work + G_scale * (htot * drdkDe - (hDk * drdi_k) * 0.25 * ((EL(K) + EL(K + 1)) + (ER(K) + ER(K + 1))));
  }
  free(W);
  /* ---- the top layer :1517-1590 */
  const double hD1 = -htot;
  hD[f2] = hD1;
  if (find_work && use_EOS) {
    const double pres_u = 0.5 * (A->pres[E3(i,j,1)] + A->pres[E3(i2,j2,1)]);
    const double T_u = 0.5 * (T[H3(i,j,1)] + T[H3(i2,j2,1)]);
    const double S_u = 0.5 * (S[H3(i,j,1)] + S[H3(i2,j2,1)]);
    double drho_dT, drho_dS;
    orc_eos_density_derivs(A->EOS, T_u, S_u, pres_u, &drho_dT, &drho_dS);
    const double drdiB1 = drho_dT * (T[H3(i2,j2,1)] - T[H3(i,j,1)]) + drho_dS * (S[H3(i2,j2,1)] - S[H3(i,j,1)]);
    const double w = G_scale * ((hD1 * drdiB1) * 0.25 * ((EL(1) + EL(2)) + (ER(1) + ER(2))));
    if (!dir && CS->use_GM_work_bug) Work = Work + w;
    else Work = Work - w;
  }
#undef HL
#undef HR
#undef EL
#undef ER
#undef DZL
#undef DZR
  return Work;
}

static int unsupported(const mom6hip_thickness_diffuse_cs_t *CS) {
  for (int n = 0; n < 10; n++) if (CS->unsupported[n]) return 1;
  return 0;
}

int orc_thickness_diffuse(const mom6hip_grid_t *G, const mom6hip_thickness_diffuse_cs_t *CS, double *h, double *uhtr, double *vhtr,
                          const double *T_in, const double *S_in, const mom6hip_eos_t *EOS, double dt, double *uhGM, double *vhGM) {
  if (!CS->initialized) return 3;      /* "MOM_thickness_diffuse: Module must be initialized before it is used." */
  if (unsupported(CS)) return 1;
  if (CS->use_FGNV_streamfn && !CS->cg1) return 4;      /* "cg1 must be associated when using FGNV streamfunction." :860 */
  if (CS->use_FGNV_streamfn && !EOS && !CS->g_prime) return 4;
  /* use_VarMix .and. use_Visbeck :205-207, :242 */
  const int use_VarMix = CS->use_variable_mixing && (CS->KHTH_Slope_Cff > 0.) && CS->L2u && CS->L2v && CS->SN_u && CS->SN_v;
  if (!CS->thickness_diffuse || !(CS->Khth > 0.0 || CS->use_variable_mixing)) return 0;      /* :192-194 */
  if (!(CS->max_Khth_CFL > 0.0)) return 1;
  if (EOS && !(T_in && S_in)) return 2;
  if (CS->MEKE_GM_src && !EOS && !CS->Rlay) return 2;
  if ((CS->slope_x != NULL) != (CS->slope_y != NULL)) return 2;
  const int is = G->isc, ie = G->iec, js = G->jsc, je = G->jec, nz = G->nk;
  const long nH = (long)ORC_NIH(G) * ORC_NJH(G), nU = (long)(ORC_NIH(G) + 1) * ORC_NJH(G), nV = (long)ORC_NIH(G) * (ORC_NJH(G) + 1);
  if (CS->MEKE_GM_src) for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) CS->MEKE_GM_src[H2(i,j)] = 0.;      /* :197-199 */

  double *e = (double *)calloc(nH * (nz + 1), 8), *pres = (double *)calloc(nH * (nz + 1), 8), *rsum = (double *)calloc(nH * (nz + 1), 8);
  double *h_avail = (double *)calloc(nH * nz, 8), *h_frac = (double *)calloc(nH * nz, 8);
  double *T = NULL, *S = NULL;
  double *uhD = (double *)calloc(nU * nz, 8), *vhD = (double *)calloc(nV * nz, 8);
  double *Work_u = (double *)calloc(nU, 8), *Work_v = (double *)calloc(nV, 8);
  /* find_eta(h, tv, G, GV, US, e, halo_size=1) :226, Boussinesq */
  for (int j = js - 1; j <= je + 1; j++) for (int i = is - 1; i <= ie + 1; i++) {
    e[E3(i,j,nz+1)] = -(G->bathyT[H2(i,j)] + 0.0);
    for (int k = nz; k >= 1; k--) e[E3(i,j,k)] = e[E3(i,j,k+1)] + h[H3(i,j,k)] * G->H_to_Z;
  }
  if (EOS) {      /* :775-778 */
    T = (double *)calloc(nH * nz, 8); S = (double *)calloc(nH * nz, 8);
    vert_fill_TS(G, h, T_in, S_in, CS->kappa_smooth * dt, T, S, 1);
  }
  /* :786-806 */
  const double I4dt = 0.25 / dt;
  const double H_to_RZ = G->Rho0 * G->H_to_Z;
  for (int j = js - 1; j <= je + 1; j++) for (int i = is - 1; i <= ie + 1; i++) {
    rsum[E3(i,j,1)] = 0.0;
    pres[E3(i,j,1)] = 0.0;
    h_avail[H3(i,j,1)] = max2(I4dt * G->areaT[H2(i,j)] * (h[H3(i,j,1)] - G->Angstrom_H), 0.0);
    rsum[E3(i,j,2)] = h_avail[H3(i,j,1)];
    h_frac[H3(i,j,1)] = 1.0;
    pres[E3(i,j,2)] = pres[E3(i,j,1)] + (G->g_Earth * H_to_RZ) * h[H3(i,j,1)];
    for (int k = 2; k <= nz; k++) {
      h_avail[H3(i,j,k)] = max2(I4dt * G->areaT[H2(i,j)] * (h[H3(i,j,k)] - G->Angstrom_H), 0.0);
      rsum[E3(i,j,k+1)] = rsum[E3(i,j,k)] + h_avail[H3(i,j,k)];
      h_frac[H3(i,j,k)] = 0.0;
      if (h_avail[H3(i,j,k)] > 0.0) h_frac[H3(i,j,k)] = h_avail[H3(i,j,k)] / rsum[E3(i,j,k+1)];
      pres[E3(i,j,k+1)] = pres[E3(i,j,k)] + (G->g_Earth * H_to_RZ) * h[H3(i,j,k)];
    }
  }
  tdf_t A = { G, CS, EOS, h, e, T, S, pres, h_avail, rsum, h_frac, dt };
  /* the diffusivity of a face :204-262 / :340-400 */
  for (int j = js; j <= je; j++) for (int I = is - 1; I <= ie; I++) {
    const int i = I;
    const double KH_CFL = (0.25 * CS->max_Khth_CFL) / (dt * (G->IdxCu[U2(I,j)] * G->IdxCu[U2(I,j)] + G->IdyCu[U2(I,j)] * G->IdyCu[U2(I,j)]));
    double Kh = CS->Khth;
    if (use_VarMix) Kh = Kh + CS->KHTH_Slope_Cff * CS->L2u[U2(I,j)] * CS->SN_u[U2(I,j)];
    if (CS->MEKE_Kh) Kh = Kh + CS->KhTh_fac * sqrt(CS->MEKE_Kh[H2(i,j)] * CS->MEKE_Kh[H2(i+1,j)]);
    if (CS->Res_fn_u) Kh = Kh * CS->Res_fn_u[U2(I,j)];
    if (CS->Depth_fn_u) Kh = Kh * CS->Depth_fn_u[U2(I,j)];      /* DEPTH_SCALED_KHTH :284-289 */
    if (CS->Khth_Max > 0) Kh = max2(CS->Khth_Min, min2(Kh, CS->Khth_Max));
    else Kh = max2(CS->Khth_Min, Kh);
    const double KH = min2(KH_CFL, Kh);
    Work_u[U2(I,j)] = face_column(&A, 0, I, j, KH, CS->slope_x, uhD);
  }
  for (int J = js - 1; J <= je; J++) for (int i = is; i <= ie; i++) {
    const int j = J;
    const double KH_CFL = (0.25 * CS->max_Khth_CFL) / (dt * (G->IdxCv[V2(i,J)] * G->IdxCv[V2(i,J)] + G->IdyCv[V2(i,J)] * G->IdyCv[V2(i,J)]));
    double Kh = CS->Khth;
    if (use_VarMix) Kh = Kh + CS->KHTH_Slope_Cff * CS->L2v[V2(i,J)] * CS->SN_v[V2(i,J)];
    if (CS->MEKE_Kh) Kh = Kh + CS->KhTh_fac * sqrt(CS->MEKE_Kh[H2(i,j)] * CS->MEKE_Kh[H2(i,j+1)]);
    if (CS->Res_fn_v) Kh = Kh * CS->Res_fn_v[V2(i,J)];
    if (CS->Depth_fn_v) Kh = Kh * CS->Depth_fn_v[V2(i,J)];
    if (CS->Khth_Max > 0) Kh = max2(CS->Khth_Min, min2(Kh, CS->Khth_Max));
    else Kh = max2(CS->Khth_Min, Kh);
    const double KH = min2(KH_CFL, Kh);
    Work_v[V2(i,J)] = face_column(&A, 1, i, J, KH, CS->slope_y, vhD);
  }
  if (CS->MEKE_GM_src) {      /* :1552-1560 */
    for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
      const int I = i, J = j;
      const double Work_h = 0.5 * G->IareaT[H2(i,j)] * ((Work_u[U2(I-1,j)] + Work_u[U2(I,j)]) + (Work_v[V2(i,J-1)] + Work_v[V2(i,J)]));
      CS->MEKE_GM_src[H2(i,j)] = CS->MEKE_GM_src[H2(i,j)] + Work_h;
    }
  }
  /* :607-620 */
  for (int k = 1; k <= nz; k++) {
    for (int j = js; j <= je; j++) for (int I = is - 1; I <= ie; I++) {
      uhtr[U3(I,j,k)] = uhtr[U3(I,j,k)] + uhD[U3(I,j,k)] * dt;
      if (uhGM) uhGM[U3(I,j,k)] = uhD[U3(I,j,k)];
    }
    for (int J = js - 1; J <= je; J++) for (int i = is; i <= ie; i++) {
      vhtr[V3(i,J,k)] = vhtr[V3(i,J,k)] + vhD[V3(i,J,k)] * dt;
      if (vhGM) vhGM[V3(i,J,k)] = vhD[V3(i,J,k)];
    }
    for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
      const int I = i, J = j;
      h[H3(i,j,k)] = h[H3(i,j,k)] - dt * G->IareaT[H2(i,j)] *
                     ((uhD[U3(I,j,k)] - uhD[U3(I-1,j,k)]) + (vhD[V3(i,J,k)] - vhD[V3(i,J-1,k)]));
      if (h[H3(i,j,k)] < G->Angstrom_H) h[H3(i,j,k)] = G->Angstrom_H;
    }
  }
  free(e); free(pres); free(rsum); free(h_avail); free(h_frac); free(T); free(S); free(uhD); free(vhD); free(Work_u); free(Work_v);
  return 0;
}
