/*
 * continuity.c -- CPU restatement of continuity_PPM (TEST INFRASTRUCTURE, see mom6_oracle.h).
 *
 * Restates src/core/MOM_continuity_PPM.F90:
 *   continuity_PPM :86-194, continuity_{zonal,merdional}_convergence :348-417,
 *   {zonal,meridional}_edge_thickness :425-514, PPM_reconstruction_{x,y} :2310-2581,
 *   PPM_limit_pos :2583-2622, PPM_limit_CW84 :2625-2662, ratio_max :2665-2676,
 *   {zonal,meridional}_mass_flux :519-820 / :1413-1717, {zonal,merid}_flux_layer :896-972 / :1788-1869,
 *   {zonal,meridional}_flux_thickness :976-1090 / :1873-1986, {zonal,meridional}_flux_adjust :1094-1243 /
 *   :1990-2140, set_{zonal,merid}_BT_cont :1247-1410 / :2144-2307, set_continuity_loop_bounds :2772-2799
 * for porous barriers = 1, with the OBC branches of those routines (orc_continuity_obc; OBC == NULL: not associated).
 *
 * The meridional routines of the reference are index-for-index mirror images of the zonal ones (checked
 * line by line; the two places where they differ textually -- CFL_dt for I_dt under aggress_adjust at
 * :1566, and the do_I guard around the initialisation in set_merid_BT_cont :2227 -- give the same values
 * without OBCs), so both directions are restated once, over an "along/cross" index pair.
 *
 * PARITY UNPINNED: the reference holds no known-answer vectors for continuity_PPM.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "mom6_oracle.h"

static inline double max2(double a, double b) { return a > b ? a : b; }
static inline double min2(double a, double b) { return a < b ? a : b; }
static inline double max3(double a, double b, double c) { return max2(max2(a, b), c); }
static inline double min3(double a, double b, double c) { return min2(min2(a, b), c); }

typedef struct {
  const mom6hip_grid_t *G;
  int dir;                    /* 0: zonal (along = i), 1: meridional (along = j) */
  const double *dL_face;      /* dy_Cu | dx_Cv : open face length */
  const double *IdL_T;        /* IdxT  | IdyT */
  const double *dL_T;         /* dxT   | dyT */
  const double *dLC_face;     /* dxCu  | dyCv */
  const double *mask_face;    /* mask2dCu | mask2dCv */
  /* open boundaries (NULL: not associated): the flags of this direction, OBC%segnum_u | segnum_v, and the direction whose interior
   * cell is the face's minus-side cell (OBC_DIRECTION_E | _N; the other one, _W | _S, has it on the plus side) */
  const mom6hip_obc_t *OBC;
  int open_BC, specified_BC, Flather_BC, dir_plus, dir_minus;
  const int32_t *segnum;
} dirx_t;

/* the segments of this direction: the face index along the direction and the range across it (segment%HI) */
static inline int seg_in_dir(const dirx_t *D, const mom6hip_obc_segment_t *S) { return S->direction == D->dir_plus || S->direction == D->dir_minus; }
static inline int seg_face(const dirx_t *D, const mom6hip_obc_segment_t *S) { return D->dir ? S->JsdB : S->IsdB; }
static inline int seg_c0(const dirx_t *D, const mom6hip_obc_segment_t *S) { return D->dir ? S->isd : S->jsd; }
static inline int seg_c1(const dirx_t *D, const mom6hip_obc_segment_t *S) { return D->dir ? S->ied : S->jed; }
/* segment%normal_trans / normal_vel (IsdB:IedB, jsd:jed, nk) | (isd:ied, JsdB:JedB, nk) at face (A, c), layer k (1-based) */
static inline double seg_val(const dirx_t *D, const mom6hip_obc_segment_t *S, const double *f, int A, int c, int k) {
  if (D->dir) { const long ni = S->ied - S->isd + 1, nJ = S->JedB - S->JsdB + 1; return f[(c - S->isd) + ni*((A - S->JsdB) + nJ*(long)(k-1))]; }
  const long nI = S->IedB - S->IsdB + 1, nj = S->jed - S->jsd + 1;
  return f[(A - S->IsdB) + nI*((c - S->jsd) + nj*(long)(k-1))];
}
static inline const mom6hip_obc_segment_t *seg_at(const dirx_t *D, int A, int c) {
  if (!D->OBC) return NULL;
  const int l = D->segnum[D->dir ? ORC_V2(D->G, c, A) : ORC_U2(D->G, A, c)];
  return (l != MOM6HIP_OBC_NONE) ? &D->OBC->segment[l-1] : NULL;
}

static inline long H2d(const dirx_t *D, int a, int c) { return D->dir ? ORC_H2(D->G, c, a) : ORC_H2(D->G, a, c); }
static inline long F2d(const dirx_t *D, int A, int c) { return D->dir ? ORC_V2(D->G, c, A) : ORC_U2(D->G, A, c); }
static inline long H3d(const dirx_t *D, int a, int c, int k) { return D->dir ? ORC_H3(D->G, c, a, k) : ORC_H3(D->G, a, c, k); }
static inline long F3d(const dirx_t *D, int A, int c, int k) { return D->dir ? ORC_V3(D->G, c, A, k) : ORC_U3(D->G, A, c, k); }

/* ratio_max :2665 */
static double ratio_max(double a, double b, double maxrat) {
  if (fabs(a) > fabs(maxrat*b)) return maxrat;
  return a / b;
}

/* PPM_reconstruction_x/y + PPM_limit_pos / PPM_limit_CW84 for every layer; cells a in [a0-1, a1+1], rows c in [c0,c1]
 * (zonal_edge_thickness :425-466).  h_L = west/south, h_R = east/north edge values. */
static void edge_thickness(const dirx_t *D, const mom6hip_continuity_cs_t *CS, const double *h_in, double *h_L,
                           double *h_R, int a0, int a1, int c0, int c1)
{
  const mom6hip_grid_t *G = D->G;
  const int nz = G->nk;
  const double h_min = 2.0*G->Angstrom_H;
  const double oneSixth = 1./6.;
  const int al = a0-1, ah = a1+1;                 /* isl, iel */
  const long n2 = (long)ORC_NIH(G)*ORC_NJH(G);
  ORC_PAR
  for (int k = 1; k <= nz; k++) {
    double *slp = calloc(n2, sizeof(double));      /* (the reference's slp is a 2-D automatic array inside its k loop) */
    if (CS->upwind_1st) {
      for (int c = c0; c <= c1; c++) for (int a = a0-1; a <= a1+1; a++) {
        h_L[H3d(D,a,c,k)] = h_in[H3d(D,a,c,k)]; h_R[H3d(D,a,c,k)] = h_in[H3d(D,a,c,k)];
      }
      free(slp);
      continue;
    }
#define HI(a,c) h_in[H3d(D,a,c,k)]
#define MT(a,c) G->mask2dT[H2d(D,a,c)]
    if (CS->simple_2nd) {
      for (int c = c0; c <= c1; c++) for (int a = al; a <= ah; a++) {
        double h_m1 = MT(a-1,c) * HI(a-1,c) + (1.0-MT(a-1,c)) * HI(a,c);
        double h_p1 = MT(a+1,c) * HI(a+1,c) + (1.0-MT(a+1,c)) * HI(a,c);
        h_L[H3d(D,a,c,k)] = 0.5*( h_m1 + HI(a,c) );
        h_R[H3d(D,a,c,k)] = 0.5*( h_p1 + HI(a,c) );
      }
    } else {
      for (int c = c0; c <= c1; c++) for (int a = al-1; a <= ah+1; a++) {
        double s;
        if ((MT(a-1,c) * MT(a,c) * MT(a+1,c)) == 0.0) {
          s = 0.0;
        } else {
          s = 0.5 * (HI(a+1,c) - HI(a-1,c));
          double dMx = max3(HI(a+1,c), HI(a-1,c), HI(a,c)) - HI(a,c);
          double dMn = HI(a,c) - min3(HI(a+1,c), HI(a-1,c), HI(a,c));
          s = copysign(1., s) * min2(fabs(s), 2. * min2(dMx, dMn));
        }
        slp[H2d(D,a,c)] = s;
      }
      if (D->open_BC) {                     /* :2385-2398 / :2521-2534 */
        for (int n = 0; n < D->OBC->number_of_segments; n++) {
          const mom6hip_obc_segment_t *S = &D->OBC->segment[n];
          if (!S->on_pe || !seg_in_dir(D, S)) continue;
          const int A = seg_face(D, S);
          for (int c = seg_c0(D, S); c <= seg_c1(D, S); c++) { slp[H2d(D,A+1,c)] = 0.0; slp[H2d(D,A,c)] = 0.0; }
        }
      }
      for (int c = c0; c <= c1; c++) for (int a = al; a <= ah; a++) {
        double h_m1 = MT(a-1,c) * HI(a-1,c) + (1.0-MT(a-1,c)) * HI(a,c);
        double h_p1 = MT(a+1,c) * HI(a+1,c) + (1.0-MT(a+1,c)) * HI(a,c);
        h_L[H3d(D,a,c,k)] = 0.5*( h_m1 + HI(a,c) ) + oneSixth*( slp[H2d(D,a-1,c)] - slp[H2d(D,a,c)] );
        h_R[H3d(D,a,c,k)] = 0.5*( h_p1 + HI(a,c) ) + oneSixth*( slp[H2d(D,a,c)] - slp[H2d(D,a+1,c)] );
      }
    }
    if (D->open_BC) {                       /* :2411-2432 / :2547-2568 */
      for (int n = 0; n < D->OBC->number_of_segments; n++) {
        const mom6hip_obc_segment_t *S = &D->OBC->segment[n];
        if (!S->on_pe) continue;
        const int A = seg_face(D, S), a = A;
        if (S->direction == D->dir_plus) {
          for (int c = seg_c0(D, S); c <= seg_c1(D, S); c++) {
            h_L[H3d(D,a+1,c,k)] = HI(a,c); h_R[H3d(D,a+1,c,k)] = HI(a,c);
            h_L[H3d(D,a,c,k)] = HI(a,c); h_R[H3d(D,a,c,k)] = HI(a,c);
          }
        } else if (S->direction == D->dir_minus) {
          for (int c = seg_c0(D, S); c <= seg_c1(D, S); c++) {
            h_L[H3d(D,a,c,k)] = HI(a+1,c); h_R[H3d(D,a,c,k)] = HI(a+1,c);
            h_L[H3d(D,a+1,c,k)] = HI(a+1,c); h_R[H3d(D,a+1,c,k)] = HI(a+1,c);
          }
        }
      }
    }
    if (CS->monotonic) {                    /* PPM_limit_CW84 :2625 */
      for (int c = c0; c <= c1; c++) for (int a = al; a <= ah; a++) {
        double *L = &h_L[H3d(D,a,c,k)], *R = &h_R[H3d(D,a,c,k)];
        double h_i = HI(a,c);
        if ( ( *R - h_i ) * ( h_i - *L ) <= 0. ) {
          *L = h_i ; *R = h_i;
        } else {
          double RLdiff = *R - *L;
          double RLmean = 0.5 * ( *R + *L );
          double FunFac = 6. * RLdiff * ( h_i - RLmean );
          double RLdiff2 = RLdiff * RLdiff;
          if ( FunFac >  RLdiff2 ) *L = 3. * h_i - 2. * *R;
          if ( FunFac < -RLdiff2 ) *R = 3. * h_i - 2. * *L;
        }
      }
    } else {                                /* PPM_limit_pos :2583 */
      for (int c = c0; c <= c1; c++) for (int a = al; a <= ah; a++) {
        double *L = &h_L[H3d(D,a,c,k)], *R = &h_R[H3d(D,a,c,k)];
        double hi = HI(a,c);
        double curv = 3.0*(*L + *R - 2.0*hi);
        if (curv > 0.0) {
          double dh = *R - *L;
          if (fabs(dh) < curv) {
            if (hi <= h_min) {
              *L = hi ; *R = hi;
            } else if (12.0*curv*(hi - h_min) < (curv*curv + 3.0*(dh*dh))) {
              double scale = 12.0*curv*(hi - h_min) / (curv*curv + 3.0*(dh*dh));
              *L = hi + scale*(*L - hi);
              *R = hi + scale*(*R - hi);
            }
          }
        }
      }
    }
#undef HI
#undef MT
    free(slp);
  }
}

/* zonal_flux_layer :896-972 for one face: returns uh, sets *duhdu */
static double flux_layer(const dirx_t *D, const mom6hip_continuity_cs_t *CS, double u, const double *h, const double *h_L,
                         const double *h_R, double visc_rem, double dt, int A, int c, int k, double *duhdu)
{
  const mom6hip_grid_t *G = D->G;
  const double dLf = D->dL_face[F2d(D,A,c)];
  const int a = A;
  double CFL, curv_3, h_marg, uh;
  if (u > 0.0) {
    if (CS->vol_CFL) CFL = (u * dt) * (dLf * G->IareaT[H2d(D,a,c)]);
    else CFL = u * dt * D->IdL_T[H2d(D,a,c)];
    double hW = h_L[H3d(D,a,c,k)], hE = h_R[H3d(D,a,c,k)], hc = h[H3d(D,a,c,k)];
    curv_3 = hW + hE - 2.0*hc;
    uh = (dLf * 1.0) * u * (hE + CFL * (0.5*(hW - hE) + curv_3*(CFL - 1.5)));
    h_marg = hE + CFL * ((hW - hE) + 3.0*curv_3*(CFL - 1.0));
  } else if (u < 0.0) {
    if (CS->vol_CFL) CFL = (-u * dt) * (dLf * G->IareaT[H2d(D,a+1,c)]);
    else CFL = -u * dt * D->IdL_T[H2d(D,a+1,c)];
    double hW = h_L[H3d(D,a+1,c,k)], hE = h_R[H3d(D,a+1,c,k)], hc = h[H3d(D,a+1,c,k)];
    curv_3 = hW + hE - 2.0*hc;
    uh = (dLf * 1.0) * u * (hW + CFL * (0.5*(hE-hW) + curv_3*(CFL - 1.5)));
    h_marg = hW + CFL * ((hE-hW) + 3.0*curv_3*(CFL - 1.0));
  } else {
    uh = 0.0;
    h_marg = 0.5 * (h_L[H3d(D,a+1,c,k)] + h_R[H3d(D,a,c,k)]);
  }
  *duhdu = (dLf * 1.0) * h_marg * visc_rem;
  if (D->open_BC) {                         /* :956-971 / :1854-1870 */
    const mom6hip_obc_segment_t *S = seg_at(D, A, c);
    if (S && S->open) {
      const double hi = (S->direction == D->dir_plus) ? h[H3d(D,a,c,k)] : h[H3d(D,a+1,c,k)];
      uh = (dLf * 1.0) * u * hi;
      *duhdu = (dLf * 1.0) * hi * visc_rem;
    }
  }
  return uh;
}

/* zonal_flux_adjust :1094-1243 for ONE face column (the reference's row-wide loop acts on each face
 * independently: `domore` only decides when the row stops iterating).  uh3d may be NULL. */
static double flux_adjust(const dirx_t *D, const mom6hip_continuity_cs_t *CS, const double *u, const double *h_in,
                          const double *h_L, const double *h_R, double uhbt, double uh_tot_0, double duhdu_tot_0,
                          double du_max_CFL, double du_min_CFL, double dt, const double *visc_rem /* per k, or NULL */,
                          int A, int c, double *uh3d)
{
  const mom6hip_grid_t *G = D->G;
  const int nz = G->nk, max_itts = 20;
  double du = 0.0, du_max = du_max_CFL, du_min = du_min_CFL;
  double uh_err = uh_tot_0 - uhbt, duhdu_tot = duhdu_tot_0, uh_err_best = fabs(uh_err);
  int do_I = 1;
  double *uh_aux = calloc(nz+1, sizeof(double));
  if (uh3d) for (int k = 1; k <= nz; k++) uh_aux[k] = uh3d[F3d(D,A,c,k)];
  for (int itt = 1; itt <= max_itts; itt++) {
    double tol_eta;
    if (itt <= 1) tol_eta = 1e-6 * CS->tol_eta;
    else if (itt == 2) tol_eta = 1e-4 * CS->tol_eta;
    else if (itt == 3) tol_eta = 1e-2 * CS->tol_eta;
    else tol_eta = CS->tol_eta;
    const double tol_vel = CS->tol_vel;
    if (uh_err > 0.0) du_max = du;
    else if (uh_err < 0.0) du_min = du;
    else do_I = 0;
    int domore = 0;
    if (do_I) {
      if ((dt * min2(G->IareaT[H2d(D,A,c)], G->IareaT[H2d(D,A+1,c)])*fabs(uh_err) > tol_eta) ||
          (CS->better_iter && ((fabs(uh_err) > tol_vel * duhdu_tot) || (fabs(uh_err) > uh_err_best)) )) {
        double ddu = -uh_err / duhdu_tot;
        double du_prev = du;
        du = du + ddu;
        if (fabs(ddu) < 1.0e-15*fabs(du)) {
          do_I = 0;
        } else if (ddu > 0.0) {
          if (du >= du_max) {
            du = 0.5*(du_prev + du_max);
            if (du_max - du_prev < 1.0e-15*fabs(du)) do_I = 0;
          }
        } else {
          if (du <= du_min) {
            du = 0.5*(du_prev + du_min);
            if (du_prev - du_min < 1.0e-15*fabs(du)) do_I = 0;
          }
        }
        if (do_I) domore = 1;
      } else {
        do_I = 0;
      }
    }
    if (!domore) break;
    double dsum = 0.0, usum = -uhbt;
    if ((itt < max_itts) || uh3d) {
      for (int k = 1; k <= nz; k++) {
        double vr = visc_rem ? visc_rem[k] : 1.0;
        double u_new = u[F3d(D,A,c,k)] + du * vr, dd;
        uh_aux[k] = flux_layer(D, CS, u_new, h_in, h_L, h_R, vr, dt, A, c, k, &dd);
        if (itt < max_itts) { usum = usum + uh_aux[k]; dsum = dsum + dd; }
      }
    }
    if (itt < max_itts) {
      uh_err = usum; duhdu_tot = dsum;
      uh_err_best = min2(uh_err_best, fabs(uh_err));
    }
  }
  if (uh3d) for (int k = 1; k <= nz; k++) uh3d[F3d(D,A,c,k)] = uh_aux[k];
  free(uh_aux);
  return du;
}

typedef struct { double *FA_0m, *FA_mm, *FA_0p, *FA_pp, *uBT_mm, *uBT_pp, *h_face; } btc_dir_t;

/* zonal_mass_flux :519-820 / meridional_mass_flux :1413-1717 */
static void mass_flux(const dirx_t *D, const mom6hip_continuity_cs_t *CS, const double *u, const double *h_in,
                      const double *h_L, const double *h_R, double *uh, double dt, int a0, int a1, int c0, int c1,
                      const double *uhbt, const double *visc_rem_u, double *u_cor, const btc_dir_t *BT, double *du_cor)
{
  const mom6hip_grid_t *G = D->G;
  const int nz = G->nk;
  const int use_visc_rem = (visc_rem_u != NULL);
  const int set_BT_cont = (BT != NULL);
  double CFL_dt = CS->CFL_limit_adjust / dt;
  const double I_dt = 1.0 / dt;
  if (CS->aggress_adjust) CFL_dt = I_dt;
  if (du_cor) {
    long n = D->dir ? (long)ORC_NIH(G)*(ORC_NJH(G)+1) : (long)(ORC_NIH(G)+1)*ORC_NJH(G);
    memset(du_cor, 0, n*sizeof(double));
  }
  ORC_PAR_DYN
  for (int c = c0; c <= c1; c++) {
  double *vr = calloc(nz+1, sizeof(double)), *duhdu = calloc(nz+1, sizeof(double));
  for (int A = a0-1; A <= a1; A++) {
    const int a = A;
    for (int k = 1; k <= nz; k++) {
      vr[k] = use_visc_rem ? visc_rem_u[F3d(D,A,c,k)] : 1.0;
      uh[F3d(D,A,c,k)] = flux_layer(D, CS, u[F3d(D,A,c,k)], h_in, h_L, h_R, vr[k], dt, A, c, k, &duhdu[k]);
      if (D->specified_BC) {                /* :629-634 */
        const mom6hip_obc_segment_t *S = seg_at(D, A, c);
        if (S && S->specified) uh[F3d(D,A,c,k)] = seg_val(D, S, S->normal_trans, A, c, k);
      }
    }
    if (!(uhbt || set_BT_cont)) continue;
    /* transports are not reconciled where they are specified :722-734 */
    const mom6hip_obc_segment_t *S_simple = NULL;
    if (D->specified_BC || D->Flather_BC) { const mom6hip_obc_segment_t *S = seg_at(D, A, c); if (S && S->specified) S_simple = S; }
    const int do_I = (S_simple == NULL);
    double visc_rem_max;
    if (use_visc_rem && CS->use_visc_rem_max) {
      visc_rem_max = 0.0;
      for (int k = 1; k <= nz; k++) visc_rem_max = max2(visc_rem_max, vr[k]);
    } else visc_rem_max = 1.0;
    double I_vrm = 0.0;
    if (visc_rem_max > 0.0) I_vrm = 1.0 / visc_rem_max;
    double dx_W, dx_E;
    if (CS->vol_CFL) {
      dx_W = ratio_max(G->areaT[H2d(D,a,c)], D->dL_face[F2d(D,A,c)], 1000.0*D->dL_T[H2d(D,a,c)]);
      dx_E = ratio_max(G->areaT[H2d(D,a+1,c)], D->dL_face[F2d(D,A,c)], 1000.0*D->dL_T[H2d(D,a+1,c)]);
    } else { dx_W = D->dL_T[H2d(D,a,c)]; dx_E = D->dL_T[H2d(D,a+1,c)]; }
    double du_max_CFL = 2.0* (CFL_dt * dx_W) * I_vrm;
    double du_min_CFL = -2.0 * (CFL_dt * dx_E) * I_vrm;
    double uh_tot_0 = 0.0, duhdu_tot_0 = 0.0;
    for (int k = 1; k <= nz; k++) {
      duhdu_tot_0 = duhdu_tot_0 + duhdu[k];
      uh_tot_0 = uh_tot_0 + uh[F3d(D,A,c,k)];
    }
    for (int k = 1; k <= nz; k++) {
      const double uk = u[F3d(D,A,c,k)];
      if (use_visc_rem) {
        if (CS->aggress_adjust) {
          double du_lim = 0.499*((dx_W*I_dt - uk) + min2(0.0,u[F3d(D,A-1,c,k)]));
          if (du_max_CFL * vr[k] > du_lim) du_max_CFL = du_lim / vr[k];
          du_lim = 0.499*((-dx_E*I_dt - uk) + max2(0.0,u[F3d(D,A+1,c,k)]));
          if (du_min_CFL * vr[k] < du_lim) du_min_CFL = du_lim / vr[k];
        } else {
          if (du_max_CFL * vr[k] > dx_W*CFL_dt - uk*D->mask_face[F2d(D,A,c)])
            du_max_CFL = (dx_W*CFL_dt - uk) / vr[k];
          if (du_min_CFL * vr[k] < -dx_E*CFL_dt - uk*D->mask_face[F2d(D,A,c)])
            du_min_CFL = -(dx_E*CFL_dt + uk) / vr[k];
        }
      } else {
        if (CS->aggress_adjust) {
          du_max_CFL = min2(du_max_CFL, 0.499 * ((dx_W*I_dt - uk) + min2(0.0,u[F3d(D,A-1,c,k)])) );
          du_min_CFL = max2(du_min_CFL, 0.499 * ((-dx_E*I_dt - uk) + max2(0.0,u[F3d(D,A+1,c,k)])) );
        } else {
          du_max_CFL = min2(du_max_CFL, dx_W*CFL_dt - uk);
          du_min_CFL = max2(du_min_CFL, -(dx_E*CFL_dt + uk));
        }
      }
    }
    du_max_CFL = max2(du_max_CFL,0.0);
    du_min_CFL = min2(du_min_CFL,0.0);

    if (uhbt) {
      double du = 0.0;                      /* (a face left out of the adjustment keeps du = 0 and its transports) */
      if (do_I) du = flux_adjust(D, CS, u, h_in, h_L, h_R, uhbt[F2d(D,A,c)], uh_tot_0, duhdu_tot_0,
                                 du_max_CFL, du_min_CFL, dt, vr, A, c, uh);
      if (u_cor) for (int k = 1; k <= nz; k++) {
        u_cor[F3d(D,A,c,k)] = u[F3d(D,A,c,k)] + du * vr[k];
        if (S_simple) u_cor[F3d(D,A,c,k)] = seg_val(D, S_simple, S_simple->normal_vel, A, c, k);      /* :744-748 */
      }
      if (du_cor) du_cor[F2d(D,A,c)] = du;
    }

    if (set_BT_cont && !do_I) {             /* :1400-1404 (the face is not solved for), then :759-779 */
      double FAuI = G->H_subroundoff*D->dL_face[F2d(D,A,c)];
      for (int k = 1; k <= nz; k++) {
        const double nv = seg_val(D, S_simple, S_simple->normal_vel, A, c, k);
        if ((fabs(nv) > 0.0) && S_simple->specified) FAuI = FAuI + seg_val(D, S_simple, S_simple->normal_trans, A, c, k) / nv;
      }
      BT->FA_0m[F2d(D,A,c)] = FAuI; BT->FA_0p[F2d(D,A,c)] = FAuI; BT->FA_mm[F2d(D,A,c)] = FAuI; BT->FA_pp[F2d(D,A,c)] = FAuI;
      BT->uBT_mm[F2d(D,A,c)] = 0.0; BT->uBT_pp[F2d(D,A,c)] = 0.0;
    } else if (set_BT_cont) {      /* set_zonal_BT_cont :1247-1410: its calls of flux_adjust and flux_layer do not pass OBC */
      dirx_t Dn = *D; Dn.OBC = NULL; Dn.open_BC = 0; Dn.specified_BC = 0; Dn.Flather_BC = 0;
      const dirx_t *const D = &Dn;      /* (a name of this block: the rows of this loop run on several threads) */
      const double min_visc_rem = 0.1, CFL_min = 1e-6;
      double du0 = flux_adjust(D, CS, u, h_in, h_L, h_R, 0.0, uh_tot_0, duhdu_tot_0, du_max_CFL, du_min_CFL,
                               dt, vr, A, c, NULL);
      double du_CFL = (CFL_min * I_dt) * D->dLC_face[F2d(D,A,c)];
      double duR = min2(0.0,du0 - du_CFL);
      double duL = max2(0.0,du0 + du_CFL);
      double FAmt_L = 0.0, FAmt_R = 0.0, FAmt_0 = 0.0, uhtot_L = 0.0, uhtot_R = 0.0;
      for (int k = 1; k <= nz; k++) {
        double visc_rem_lim = max2(vr[k], min_visc_rem*visc_rem_max);
        const double uk = u[F3d(D,A,c,k)];
        if (visc_rem_lim > 0.0) {
          if (uk + duR*visc_rem_lim > -du_CFL*vr[k]) duR = -(uk + du_CFL*vr[k]) / visc_rem_lim;
          if (uk + duL*visc_rem_lim < du_CFL*vr[k]) duL = -(uk - du_CFL*vr[k]) / visc_rem_lim;
        }
      }
      for (int k = 1; k <= nz; k++) {
        const double uk = u[F3d(D,A,c,k)];
        double u_L = uk + duL * vr[k], u_R = uk + duR * vr[k], u_0 = uk + du0 * vr[k];
        double d0, dL, dR;
        (void)flux_layer(D, CS, u_0, h_in, h_L, h_R, vr[k], dt, A, c, k, &d0);
        double uh_L = flux_layer(D, CS, u_L, h_in, h_L, h_R, vr[k], dt, A, c, k, &dL);
        double uh_R = flux_layer(D, CS, u_R, h_in, h_L, h_R, vr[k], dt, A, c, k, &dR);
        FAmt_0 = FAmt_0 + d0; FAmt_L = FAmt_L + dL; FAmt_R = FAmt_R + dR;
        uhtot_L = uhtot_L + uh_L; uhtot_R = uhtot_R + uh_R;
      }
      double FA_0 = FAmt_0, FA_avg = FAmt_0;
      if ((duL - du0) != 0.0) FA_avg = uhtot_L / (duL - du0);
      if (FA_avg > max2(FA_0, FAmt_L)) FA_avg = max2(FA_0, FAmt_L);
      else if (FA_avg < min2(FA_0, FAmt_L)) FA_0 = FA_avg;
      BT->FA_0m[F2d(D,A,c)] = FA_0; BT->FA_mm[F2d(D,A,c)] = FAmt_L;
      if (fabs(FA_0-FAmt_L) <= 1e-12*FA_0) BT->uBT_mm[F2d(D,A,c)] = 0.0;
      else BT->uBT_mm[F2d(D,A,c)] = (1.5 * (duL - du0)) * ((FAmt_L - FA_avg) / (FAmt_L - FA_0));

      FA_0 = FAmt_0; FA_avg = FAmt_0;
      if ((duR - du0) != 0.0) FA_avg = uhtot_R / (duR - du0);
      if (FA_avg > max2(FA_0, FAmt_R)) FA_avg = max2(FA_0, FAmt_R);
      else if (FA_avg < min2(FA_0, FAmt_R)) FA_0 = FA_avg;
      BT->FA_0p[F2d(D,A,c)] = FA_0; BT->FA_pp[F2d(D,A,c)] = FAmt_R;
      if (fabs(FAmt_R - FA_0) <= 1e-12*FA_0) BT->uBT_pp[F2d(D,A,c)] = 0.0;
      else BT->uBT_pp[F2d(D,A,c)] = (1.5 * (duR - du0)) * ((FAmt_R - FA_avg) / (FAmt_R - FA_0));
    }
  }
  free(vr); free(duhdu);
  }

  /* the face areas of open segments :782-805 / :1672-1695 (after every row) */
  if (D->open_BC && set_BT_cont) {
    for (int n = 0; n < D->OBC->number_of_segments; n++) {
      const mom6hip_obc_segment_t *S = &D->OBC->segment[n];
      if (!(S->open && (D->dir ? S->is_N_or_S : S->is_E_or_W))) continue;
      const int A = seg_face(D, S), ai = (S->direction == D->dir_plus) ? A : A + 1;
      for (int c = seg_c0(D, S); c <= seg_c1(D, S); c++) {
        double FA_u = 0.0;
        for (int k = 1; k <= nz; k++) FA_u = FA_u + h_in[H3d(D,ai,c,k)]*(D->dL_face[F2d(D,A,c)]*1.0);
        BT->FA_0m[F2d(D,A,c)] = FA_u; BT->FA_0p[F2d(D,A,c)] = FA_u; BT->FA_mm[F2d(D,A,c)] = FA_u; BT->FA_pp[F2d(D,A,c)] = FA_u;
        BT->uBT_mm[F2d(D,A,c)] = 0.0; BT->uBT_pp[F2d(D,A,c)] = 0.0;
      }
    }
  }

  /* zonal_flux_thickness :976-1057 */
  if (set_BT_cont && BT->h_face) {
    const double *uu = u_cor ? u_cor : u;
    ORC_PAR
    for (int k = 1; k <= nz; k++) for (int c = c0; c <= c1; c++) for (int A = a0-1; A <= a1; A++) {
      const int a = A;
      const double uk = uu[F3d(D,A,c,k)];
      double CFL, curv_3, h_avg, h_marg;
      if (uk > 0.0) {
        if (CS->vol_CFL) CFL = (uk * dt) * (D->dL_face[F2d(D,A,c)] * G->IareaT[H2d(D,a,c)]);
        else CFL = uk * dt * D->IdL_T[H2d(D,a,c)];
        double hW = h_L[H3d(D,a,c,k)], hE = h_R[H3d(D,a,c,k)];
        curv_3 = hW + hE - 2.0*h_in[H3d(D,a,c,k)];
        h_avg = hE + CFL * (0.5*(hW - hE) + curv_3*(CFL - 1.5));
        h_marg = hE + CFL * ((hW - hE) + 3.0*curv_3*(CFL - 1.0));
      } else if (uk < 0.0) {
        if (CS->vol_CFL) CFL = (-uk*dt) * (D->dL_face[F2d(D,A,c)] * G->IareaT[H2d(D,a+1,c)]);
        else CFL = -uk * dt * D->IdL_T[H2d(D,a+1,c)];
        double hW = h_L[H3d(D,a+1,c,k)], hE = h_R[H3d(D,a+1,c,k)];
        curv_3 = hW + hE - 2.0*h_in[H3d(D,a+1,c,k)];
        h_avg = hW + CFL * (0.5*(hE-hW) + curv_3*(CFL - 1.5));
        h_marg = hW + CFL * ((hE-hW) + 3.0*curv_3*(CFL - 1.0));
      } else {
        h_avg = 0.5 * (h_L[H3d(D,a+1,c,k)] + h_R[H3d(D,a,c,k)]);
        h_marg = 0.5 * (h_L[H3d(D,a+1,c,k)] + h_R[H3d(D,a,c,k)]);
      }
      double hu = CS->marginal_faces ? h_marg : h_avg;
      if (visc_rem_u) hu = hu * (visc_rem_u[F3d(D,A,c,k)] * 1.0);
      else hu = hu * 1.0;
      BT->h_face[F3d(D,A,c,k)] = hu;
    }
    if (D->open_BC) {                       /* :1058-1088 / :1960-1990 */
      for (int n = 0; n < D->OBC->number_of_segments; n++) {
        const mom6hip_obc_segment_t *S = &D->OBC->segment[n];
        if (!(S->open && (D->dir ? S->is_N_or_S : S->is_E_or_W))) continue;
        const int A = seg_face(D, S), ai = (S->direction == D->dir_plus) ? A : A + 1;
        for (int k = 1; k <= nz; k++) for (int c = seg_c0(D, S); c <= seg_c1(D, S); c++) {
          if (visc_rem_u) BT->h_face[F3d(D,A,c,k)] = h_in[H3d(D,ai,c,k)] * (visc_rem_u[F3d(D,A,c,k)] * 1.0);
          else BT->h_face[F3d(D,A,c,k)] = h_in[H3d(D,ai,c,k)] * 1.0;
        }
      }
    }
  }
}

/* continuity_zonal_convergence :348-381 / continuity_merdional_convergence :384-417 */
static void convergence(const dirx_t *D, double *h, const double *uh, double dt, int a0, int a1, int c0, int c1,
                        const double *hin, double h_min)
{
  const mom6hip_grid_t *G = D->G;
  ORC_PAR
  for (int k = 1; k <= G->nk; k++) for (int c = c0; c <= c1; c++) for (int a = a0; a <= a1; a++) {
    const double *src = hin ? hin : h;
    h[H3d(D,a,c,k)] = max2( src[H3d(D,a,c,k)] - dt * G->IareaT[H2d(D,a,c)] *
                            (uh[F3d(D,a,c,k)] - uh[F3d(D,a-1,c,k)]), h_min );
  }
}

/* continuity_PPM :86-194 */
int orc_continuity(const mom6hip_grid_t *G, const mom6hip_continuity_cs_t *CS, const double *u, const double *v,
                   const double *hin, double *h, double *uh, double *vh, double dt, const double *uhbt,
                   const double *vhbt, const double *visc_rem_u, const double *visc_rem_v, double *u_cor,
                   double *v_cor, const mom6hip_bt_cont_t *BT_cont, double *du_cor, double *dv_cor)
{
  return orc_continuity_obc(G, CS, NULL, u, v, hin, h, uh, vh, dt, uhbt, vhbt, visc_rem_u, visc_rem_v, u_cor, v_cor, BT_cont, du_cor, dv_cor);
}

int orc_continuity_obc(const mom6hip_grid_t *G, const mom6hip_continuity_cs_t *CS, const mom6hip_obc_t *OBC, const double *u,
                       const double *v, const double *hin, double *h, double *uh, double *vh, double dt, const double *uhbt,
                       const double *vhbt, const double *visc_rem_u, const double *visc_rem_v, double *u_cor, double *v_cor,
                       const mom6hip_bt_cont_t *BT_cont, double *du_cor, double *dv_cor)
{
  if ((visc_rem_u != NULL) != (visc_rem_v != NULL)) return 2;
  const double h_min = G->Angstrom_H;
  const int x_first = ((G->first_direction % 2) == 0);
  int stencil = 3; if (CS->simple_2nd) stencil = 2; if (CS->upwind_1st) stencil = 1;
  const long n3 = (long)ORC_NIH(G)*ORC_NJH(G)*G->nk;
  double *h_L = calloc(n3, sizeof(double)), *h_R = calloc(n3, sizeof(double));
  dirx_t X = { G, 0, G->dy_Cu, G->IdxT, G->dxT, G->dxCu, G->mask2dCu, NULL, 0, 0, 0, MOM6HIP_OBC_DIRECTION_E, MOM6HIP_OBC_DIRECTION_W, NULL };
  dirx_t Y = { G, 1, G->dx_Cv, G->IdyT, G->dyT, G->dyCv, G->mask2dCv, NULL, 0, 0, 0, MOM6HIP_OBC_DIRECTION_N, MOM6HIP_OBC_DIRECTION_S, NULL };
  if (OBC) {
    if (OBC->number_of_segments > 0 && !(OBC->segment && OBC->segnum_u && OBC->segnum_v)) return 3;
    X.OBC = OBC; Y.OBC = OBC; X.segnum = OBC->segnum_u; Y.segnum = OBC->segnum_v;
    /* PPM_reconstruction_x/y, flux_layer and flux_thickness test OBC%open_*_BCs_exist_globally alone (:2341, :932, :1060); the mass
     * flux routines test OBC%OBC_pe first (:595-599) */
    X.open_BC = OBC->open_u_BCs_exist_globally != 0; Y.open_BC = OBC->open_v_BCs_exist_globally != 0;
    if (OBC->OBC_pe) {
      X.specified_BC = OBC->specified_u_BCs_exist_globally != 0; Y.specified_BC = OBC->specified_v_BCs_exist_globally != 0;
      X.Flather_BC = OBC->Flather_u_BCs_exist_globally != 0; Y.Flather_BC = OBC->Flather_v_BCs_exist_globally != 0;
    }
  }
  btc_dir_t bx, by, *pbx = NULL, *pby = NULL;
  if (BT_cont) {
    bx.FA_0m = BT_cont->FA_u_W0; bx.FA_mm = BT_cont->FA_u_WW; bx.FA_0p = BT_cont->FA_u_E0; bx.FA_pp = BT_cont->FA_u_EE;
    bx.uBT_mm = BT_cont->uBT_WW; bx.uBT_pp = BT_cont->uBT_EE; bx.h_face = BT_cont->h_u;
    by.FA_0m = BT_cont->FA_v_S0; by.FA_mm = BT_cont->FA_v_SS; by.FA_0p = BT_cont->FA_v_N0; by.FA_pp = BT_cont->FA_v_NN;
    by.uBT_mm = BT_cont->vBT_SS; by.uBT_pp = BT_cont->vBT_NN; by.h_face = BT_cont->h_v;
    pbx = &bx; pby = &by;
  }
  const int is = G->isc, ie = G->iec, js = G->jsc, je = G->jec;
  if (x_first) {
    /* zonal, with rows extended by the stencil for the meridional pass */
    edge_thickness(&X, CS, hin, h_L, h_R, is, ie, js-stencil, je+stencil);
    mass_flux(&X, CS, u, hin, h_L, h_R, uh, dt, is, ie, js-stencil, je+stencil, uhbt, visc_rem_u, u_cor, pbx, du_cor);
    convergence(&X, h, uh, dt, is, ie, js-stencil, je+stencil, hin, 0.0);
    edge_thickness(&Y, CS, h, h_L, h_R, js, je, is, ie);
    mass_flux(&Y, CS, v, h, h_L, h_R, vh, dt, js, je, is, ie, vhbt, visc_rem_v, v_cor, pby, dv_cor);
    convergence(&Y, h, vh, dt, js, je, is, ie, NULL, h_min);
  } else {
    edge_thickness(&Y, CS, hin, h_L, h_R, js, je, is-stencil, ie+stencil);
    mass_flux(&Y, CS, v, hin, h_L, h_R, vh, dt, js, je, is-stencil, ie+stencil, vhbt, visc_rem_v, v_cor, pby, dv_cor);
    convergence(&Y, h, vh, dt, js, je, is-stencil, ie+stencil, hin, 0.0);
    edge_thickness(&X, CS, h, h_L, h_R, is, ie, js, je);
    mass_flux(&X, CS, u, h, h_L, h_R, uh, dt, is, ie, js, je, uhbt, visc_rem_u, u_cor, pbx, du_cor);
    convergence(&X, h, uh, dt, is, ie, js, je, NULL, h_min);
  }
  free(h_L); free(h_R);
  return 0;
}
