/*
 * tracer_advect.c -- CPU restatement of MOM_tracer_advect (TEST INFRASTRUCTURE, see mom6_oracle.h).
 *
 * Restates, loop for loop and with the same parenthesisation:
 *   advect_tracer  src/tracer/MOM_tracer_advect.F90:52-324
 *   advect_x       src/tracer/MOM_tracer_advect.F90:329-701
 *   advect_y       src/tracer/MOM_tracer_advect.F90:705-1087
 * for the configuration of the hot path: OBC not associated, no diagnostic arrays (ad_x, ad_y,
 * advection_xy, ad2d_* are not associated in a benchmark run).
 *
 * PARITY UNPINNED: the reference has no known-answer vectors for this routine.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "mom6_oracle.h"

static inline double max2(double a, double b) { return a > b ? a : b; }
static inline double min2(double a, double b) { return a < b ? a : b; }
static inline double max3(double a, double b, double c) { return max2(max2(a, b), c); }
static inline double min3(double a, double b, double c) { return min2(min2(a, b), c); }
/* Fortran sign(a,b): |a| with the sign of b */
static inline double fsign(double a, double b) { return copysign(fabs(a), b); }

typedef struct {
  const mom6hip_grid_t *G;
  int ntr, usePPM, useHuynh;
  double *const *tr;
  const double *conc_underflow;
  double *hprev, *uhr, *vhr, *uh_neglect, *vh_neglect;
  unsigned char *domore_u;   /* (jsd:jed, nk) */
  unsigned char *domore_v;   /* (jsd-1:jed, nk) */
  const mom6hip_obc_t *OBC;  /* NULL, or the open boundaries: the tracer registries of the segments are read (:442-477, :580-627, ...) */
} adv_t;

/* the value a registered tracer takes outside a segment: its reservoir at the face (a, b) = (I, j) | (i, J) of layer k, or the inflow
 * concentration */
static inline double seg_tracer(const mom6hip_obc_segment_t *S, const mom6hip_obc_segment_tracer_t *T, int a, int b, int k) {
  if (!T->tres) return T->OBC_inflow_conc;
  const int a0 = S->is_N_or_S ? S->isd : S->IsdB, a1 = S->is_N_or_S ? S->ied : S->IedB;
  const int b0 = S->is_N_or_S ? S->JsdB : S->jsd, b1 = S->is_N_or_S ? S->JedB : S->jed;
  return T->tres[(a - a0) + (long)(a1 - a0 + 1) * ((b - b0) + (long)(b1 - b0 + 1) * (k - 1))];
}

#define DU(A,j,k) (A)->domore_u[((j)-G->jsd) + (long)ORC_NJH(G)*((k)-1)]
#define DV(A,J,k) (A)->domore_v[((J)-G->jsd+1) + (long)(ORC_NJH(G)+1)*((k)-1)]

/* limited slope, :427-431 / :809-813 */
static inline double plm_slope(double Tp, double Tc, double Tm, double mask)
{
  double dMx = max3(Tp, Tc, Tm) - Tc;
  double dMn = Tc - min3(Tp, Tc, Tm);
  return mask * fsign(min3(0.5*fabs(Tp-Tm), 2.0*dMx, 2.0*dMn), Tp-Tm);
}

/* PPM edge values + CW84 limiting + CFL-integrated flux, :526-556 / :911-941 */
static inline double ppm_flux(double Tp, double Tc, double Tm, double sm, double sc, double sp,
                              int useHuynh, double mask_prod, double uhh, double CFL)
{
  double aL, aR, dA, mA, a6;
  if (useHuynh) {
    aL = ( 5.*Tc + ( 2.*Tm - Tp ) )/6.;
    aL = max2( min2(Tc,Tm), aL) ; aL = min2( max2(Tc,Tm), aL);
    aR = ( 5.*Tc + ( 2.*Tp - Tm ) )/6.;
    aR = max2( min2(Tc,Tp), aR) ; aR = min2( max2(Tc,Tp), aR);
  } else {
    aL = 0.5 * ((Tm + Tc) + (sm - sc) / 3.);
    aR = 0.5 * ((Tc + Tp) + (sc - sp) / 3.);
  }
  dA = aR - aL ; mA = 0.5*( aR + aL );
  if (mask_prod*(Tp-Tc)*(Tc-Tm) <= 0.) {
    aL = Tc ; aR = Tc;
  } else if ( dA*(Tc-mA) > (dA*dA)/6. ) {
    aL = 3.*Tc - 2.*aR;
  } else if ( dA*(Tc-mA) < - (dA*dA)/6. ) {
    aR = 3.*Tc - 2.*aL;
  }
  a6 = 6.*Tc - 3. * (aR + aL);
  if (uhh >= 0.0)
    return uhh*( aR - 0.5 * CFL * ( ( aR - aL ) - a6 * ( 1. - 2./3. * CFL ) ) );
  else
    return uhh*( aL + 0.5 * CFL * ( ( aR - aL ) + a6 * ( 1. - 2./3. * CFL ) ) );
}

/* advect_x, :329-701 */
static void advect_x(adv_t *A, int is, int ie, int js, int je, int k)
{
  const mom6hip_grid_t *G = A->G;
  const int ntr = A->ntr, usePPM = A->usePPM, useHuynh = A->useHuynh;
  const int nih = ORC_NIH(G);
  const int usePLMslope = !(usePPM && useHuynh);
  int stencil = 1;
  if (usePPM && !useHuynh) stencil = 2;
  const double min_h = 0.1*G->Angstrom_H;
  const double tiny_h = DBL_MIN;           /* tiny(min_h) */
  const double h_neglect = G->H_subroundoff;

  /* row work arrays, indexed by i-isd+1 so that I = isd-1 .. ied are valid */
  const int nw = nih + 2;
#define W(a,i) a[(i)-G->isd+1]
  double *slope_x = calloc((size_t)nw*ntr, sizeof(double));
  double *flux_x  = calloc((size_t)nw*ntr, sizeof(double));
  double *T_tmp   = calloc((size_t)nw*ntr, sizeof(double));
  double *uhh = calloc(nw, sizeof(double)), *hlst = calloc(nw, sizeof(double));
  double *Ihnew = calloc(nw, sizeof(double)), *CFL = calloc(nw, sizeof(double));
  unsigned char *do_i = calloc(nw, 1);
#define WM(a,i,m) a[((i)-G->isd+1) + (long)nw*(m)]

  for (int I = is-1; I <= ie; I++) W(CFL,I) = 0.0;

  for (int j = js; j <= je; j++) if (DU(A,j,k)) {
    DU(A,j,k) = 0;

    if (usePLMslope) {
      for (int m = 0; m < ntr; m++) for (int i = is-stencil; i <= ie+stencil; i++) {
        const double *t = A->tr[m];
        double Tp = t[ORC_H3(G,i+1,j,k)], Tc = t[ORC_H3(G,i,j,k)], Tm = t[ORC_H3(G,i-1,j,k)];
        WM(slope_x,i,m) = plm_slope(Tp, Tc, Tm,
                             G->mask2dCu[ORC_U2(G,i,j)]*G->mask2dCu[ORC_U2(G,i-1,j)]);
      }
    }

    for (int m = 0; m < ntr; m++)
      for (int i = G->isd; i <= G->ied; i++) WM(T_tmp,i,m) = A->tr[m][ORC_H3(G,i,j,k)];
    /* :441-477: the registered tracers take their reservoir (or inflow) values in the cell outside a segment, and the slopes of the three
     * cells about its face are formed again */
    if (A->OBC && A->OBC->OBC_pe) for (int n = 0; n < A->OBC->number_of_segments; n++) {
      const mom6hip_obc_segment_t *S = &A->OBC->segment[n];
      if (!S->tr_Reg) continue;
      if (S->is_E_or_W && j >= S->jsd && j <= S->jed) {
        const int I = S->IsdB;
        for (int q = 0; q < S->ntseg; q++) {
          const int m = S->tr_Reg[q].ntr_index - 1;
          const double val = seg_tracer(S, &S->tr_Reg[q], I, j, k);
          if (S->direction == MOM6HIP_OBC_DIRECTION_W) WM(T_tmp,I,m) = val;
          else WM(T_tmp,I+1,m) = val;
        }
        for (int m = 0; m < ntr; m++) for (int i = I-1; i <= I+1; i++) {
          double Tp = WM(T_tmp,i+1,m), Tc = WM(T_tmp,i,m), Tm = WM(T_tmp,i-1,m);
          WM(slope_x,i,m) = plm_slope(Tp, Tc, Tm, G->mask2dCu[ORC_U2(G,i,j)]*G->mask2dCu[ORC_U2(G,i-1,j)]);   /* :472 (Fortran does not tell I from i: the masks are the loop cell's) */
        }
      }
    }

    /* :485-514 */
    for (int I = is-1; I <= ie; I++) {
      const int i = I;
      double uhrI = A->uhr[ORC_U3(G,I,j,k)];
      if ((uhrI == 0.0) ||
          ((uhrI < 0.0) && (A->hprev[ORC_H3(G,i+1,j,k)] <= tiny_h)) ||
          ((uhrI > 0.0) && (A->hprev[ORC_H3(G,i,j,k)] <= tiny_h)) ) {
        W(uhh,I) = 0.0;
        W(CFL,I) = 0.0;
      } else if (uhrI < 0.0) {
        double hup = A->hprev[ORC_H3(G,i+1,j,k)] - G->areaT[ORC_H2(G,i+1,j)]*min_h;
        double hlos = max2(0.0, A->uhr[ORC_U3(G,I+1,j,k)]);
        if ((((hup - hlos) + uhrI) < 0.0) &&
            ((0.5*hup + uhrI) < 0.0)) {
          W(uhh,I) = min3(-0.5*hup, -hup+hlos, 0.0);
          DU(A,j,k) = 1;
        } else {
          W(uhh,I) = uhrI;
        }
        W(CFL,I) = - W(uhh,I) / (A->hprev[ORC_H3(G,i+1,j,k)]);
      } else {
        double hup = A->hprev[ORC_H3(G,i,j,k)] - G->areaT[ORC_H2(G,i,j)]*min_h;
        double hlos = max2(0.0, -A->uhr[ORC_U3(G,I-1,j,k)]);
        if ((((hup - hlos) - uhrI) < 0.0) &&
            ((0.5*hup - uhrI) < 0.0)) {
          W(uhh,I) = max3(0.5*hup, hup-hlos, 0.0);
          DU(A,j,k) = 1;
        } else {
          W(uhh,I) = uhrI;
        }
        W(CFL,I) = W(uhh,I) / (A->hprev[ORC_H3(G,i,j,k)]);
      }
    }

    if (usePPM) {
      for (int m = 0; m < ntr; m++) for (int I = is-1; I <= ie; I++) {
        int i_up = (W(uhh,I) >= 0.0) ? I : I+1;
        double Tp = WM(T_tmp,i_up+1,m), Tc = WM(T_tmp,i_up,m), Tm = WM(T_tmp,i_up-1,m);
        double sm = 0., sc = 0., sp = 0.;
        if (!useHuynh) { sm = WM(slope_x,i_up-1,m); sc = WM(slope_x,i_up,m); sp = WM(slope_x,i_up+1,m); }
        WM(flux_x,I,m) = ppm_flux(Tp, Tc, Tm, sm, sc, sp, useHuynh,
                           G->mask2dCu[ORC_U2(G,i_up,j)]*G->mask2dCu[ORC_U2(G,i_up-1,j)],
                           W(uhh,I), W(CFL,I));
      }
    } else {
      for (int m = 0; m < ntr; m++) for (int I = is-1; I <= ie; I++) {
        const int i = I;
        if (W(uhh,I) >= 0.0) {
          double Tc = WM(T_tmp,i,m);
          WM(flux_x,I,m) = W(uhh,I)*( Tc + 0.5 * WM(slope_x,i,m) * ( 1. - W(CFL,I) ) );
        } else {
          double Tc = WM(T_tmp,i+1,m);
          WM(flux_x,I,m) = W(uhh,I)*( Tc - 0.5 * WM(slope_x,i+1,m) * ( 1. - W(CFL,I) ) );
        }
      }
    }

    if (A->OBC && A->OBC->OBC_pe) {      /* :580-627 */
      const mom6hip_obc_t *OBC = A->OBC;
      if (OBC->specified_u_BCs_exist_globally || OBC->open_u_BCs_exist_globally)
        for (int n = 0; n < OBC->number_of_segments; n++) {
          const mom6hip_obc_segment_t *S = &OBC->segment[n];
          if (!S->tr_Reg) continue;
          if (S->is_E_or_W && j >= S->jsd && j <= S->jed) {
            const int I = S->IsdB;
            const double u = A->uhr[ORC_U3(G,I,j,k)];
            if (((u > 0.0) && (S->direction == MOM6HIP_OBC_DIRECTION_W)) || ((u < 0.0) && (S->direction == MOM6HIP_OBC_DIRECTION_E))) {
              W(uhh,I) = u;
              for (int q = 0; q < S->ntseg; q++)
                WM(flux_x,I,S->tr_Reg[q].ntr_index - 1) = W(uhh,I) * seg_tracer(S, &S->tr_Reg[q], I, j, k);
            }
          }
        }
      if (OBC->open_u_BCs_exist_globally)
        for (int n = 0; n < OBC->number_of_segments; n++) {
          const mom6hip_obc_segment_t *S = &OBC->segment[n];
          const int I = S->IsdB;
          if (S->is_E_or_W && (j >= S->jsd && j <= S->jed)) {
            if (S->specified) continue;
            if (!S->tr_Reg) continue;
            const double u = A->uhr[ORC_U3(G,I,j,k)];
            if (((u > 0.0) && (G->mask2dT[ORC_H2(G,I,j)] < 0.5)) || ((u < 0.0) && (G->mask2dT[ORC_H2(G,I+1,j)] < 0.5))) {
              W(uhh,I) = u;
              for (int q = 0; q < S->ntseg; q++)
                WM(flux_x,I,S->tr_Reg[q].ntr_index - 1) = W(uhh,I) * seg_tracer(S, &S->tr_Reg[q], I, j, k);
            }
          }
        }
    }

    /* :632-649 */
    for (int I = is-1; I <= ie; I++) {
      double *u = &A->uhr[ORC_U3(G,I,j,k)];
      *u = *u - W(uhh,I);
      if (fabs(*u) < A->uh_neglect[ORC_U2(G,I,j)]) *u = 0.0;
    }
    for (int i = is; i <= ie; i++) {
      if ((W(uhh,i) != 0.0) || (W(uhh,i-1) != 0.0)) {
        double *hp = &A->hprev[ORC_H3(G,i,j,k)];
        double aT = G->areaT[ORC_H2(G,i,j)];
        W(do_i,i) = 1;
        W(hlst,i) = *hp;
        *hp = *hp - (W(uhh,i) - W(uhh,i-1));
        if (*hp <= 0.0) { W(do_i,i) = 0; }
        else if (*hp < h_neglect*aT) {
          W(hlst,i) = W(hlst,i) + (h_neglect*aT - *hp);
          W(Ihnew,i) = 1.0 / (h_neglect*aT);
        } else { W(Ihnew,i) = 1.0 / *hp; }
      } else {
        W(do_i,i) = 0;
      }
    }

    /* :652-662 */
    for (int m = 0; m < ntr; m++) {
      for (int i = is; i <= ie; i++) {
        if (W(do_i,i)) {
          if (W(Ihnew,i) > 0.0) {
            double *t = &A->tr[m][ORC_H3(G,i,j,k)];
            *t = (*t * W(hlst,i) - (WM(flux_x,i,m) - WM(flux_x,i-1,m))) * W(Ihnew,i);
          }
        }
      }
    }
  }

  /* :683-687 */
  for (int m = 0; m < ntr; m++) if (A->conc_underflow && A->conc_underflow[m] > 0.0) {
    for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
      double *t = &A->tr[m][ORC_H3(G,i,j,k)];
      if (fabs(*t) < A->conc_underflow[m]) *t = 0.0;
    }
  }

  free(slope_x); free(flux_x); free(T_tmp); free(uhh); free(hlst); free(Ihnew); free(CFL); free(do_i);
#undef W
#undef WM
}

/* advect_y, :705-1087 */
static void advect_y(adv_t *A, int is, int ie, int js, int je, int k)
{
  const mom6hip_grid_t *G = A->G;
  const int ntr = A->ntr, usePPM = A->usePPM, useHuynh = A->useHuynh;
  const int nih = ORC_NIH(G), njh = ORC_NJH(G);
  const int usePLMslope = !(usePPM && useHuynh);
  int stencil = 1;
  if (usePPM && !useHuynh) stencil = 2;
  const double min_h = 0.1*G->Angstrom_H;
  const double tiny_h = DBL_MIN;
  const double h_neglect = G->H_subroundoff;

  /* slope_y(i,m,j), flux_y(i,m,J), T_tmp(i,m,j), vhh(i,J): j index offset so J = jsd-1..jed valid */
  const long rowsz = (long)nih*ntr;
#define S3(a,i,m,j) a[((i)-G->isd) + (long)nih*(m) + rowsz*((j)-G->jsd+1)]
#define V2W(a,i,J)  a[((i)-G->isd) + (long)nih*((J)-G->jsd+1)]
  double *slope_y = calloc((size_t)rowsz*(njh+2), sizeof(double));
  double *flux_y  = calloc((size_t)rowsz*(njh+2), sizeof(double));
  double *T_tmp   = calloc((size_t)rowsz*(njh+2), sizeof(double));
  double *vhh = calloc((size_t)nih*(njh+2), sizeof(double));
  double *hlst = calloc(nih+2, sizeof(double)), *Ihnew = calloc(nih+2, sizeof(double));
  double *CFL = calloc(nih+2, sizeof(double));
  unsigned char *do_j_tr = calloc(njh+8, 1);   /* (jsd:jed), padded */
  unsigned char *do_i = calloc(nih+2, 1);
#define W(a,i) a[(i)-G->isd+1]
#define DJ(j) do_j_tr[(j)-G->jsd+4]

  /* :787-788 */
  for (int J = js-1; J <= je; J++) if (DV(A,J,k)) {
    for (int j2 = 1-stencil; j2 <= stencil; j2++) DJ(J+j2) = 1;
  }

  /* :793-815 */
  if (usePLMslope) {
    for (int j = js-stencil; j <= je+stencil; j++) if (DJ(j)) {
      for (int m = 0; m < ntr; m++) for (int i = is; i <= ie; i++) {
        const double *t = A->tr[m];
        double Tp = t[ORC_H3(G,i,j+1,k)], Tc = t[ORC_H3(G,i,j,k)], Tm = t[ORC_H3(G,i,j-1,k)];
        S3(slope_y,i,m,j) = plm_slope(Tp, Tc, Tm,
                               G->mask2dCv[ORC_V2(G,i,j)]*G->mask2dCv[ORC_V2(G,i,j-1)]);
      }
    }
  }

  /* :820-822 */
  for (int j = G->jsd; j <= G->jed; j++) for (int m = 0; m < ntr; m++)
    for (int i = G->isd; i <= G->ied; i++) S3(T_tmp,i,m,j) = A->tr[m][ORC_H3(G,i,j,k)];

  /* :823-861: the registered tracers outside the segments, and the slopes of the three cells about their faces */
  if (A->OBC && A->OBC->OBC_pe) for (int n = 0; n < A->OBC->number_of_segments; n++) {
    const mom6hip_obc_segment_t *S = &A->OBC->segment[n];
    if (!S->tr_Reg) continue;
    for (int i = is; i <= ie; i++) {
      if (S->is_N_or_S && i >= S->isd && i <= S->ied) {
        const int J = S->JsdB;
        for (int q = 0; q < S->ntseg; q++) {
          const int m = S->tr_Reg[q].ntr_index - 1;
          const double val = seg_tracer(S, &S->tr_Reg[q], i, J, k);
          if (S->direction == MOM6HIP_OBC_DIRECTION_S) S3(T_tmp,i,m,J) = val;
          else S3(T_tmp,i,m,J+1) = val;
        }
        for (int m = 0; m < ntr; m++) for (int j = J-1; j <= J+1; j++) {
          double Tp = S3(T_tmp,i,m,j+1), Tc = S3(T_tmp,i,m,j), Tm = S3(T_tmp,i,m,j-1);
          S3(slope_y,i,m,j) = plm_slope(Tp, Tc, Tm, G->mask2dCv[ORC_V2(G,i,j)]*G->mask2dCv[ORC_V2(G,i,j-1)]);   /* :854 (J is j in Fortran: the loop cell's masks) */
        }
      }
    }
  }

  /* :868-1019 */
  for (int J = js-1; J <= je; J++) {
    const int j = J;
    if (DV(A,J,k)) {
      DV(A,J,k) = 0;

      for (int i = is; i <= ie; i++) {
        double vhrJ = A->vhr[ORC_V3(G,i,J,k)];
        if ((vhrJ == 0.0) ||
            ((vhrJ < 0.0) && (A->hprev[ORC_H3(G,i,j+1,k)] <= tiny_h)) ||
            ((vhrJ > 0.0) && (A->hprev[ORC_H3(G,i,j,k)] <= tiny_h)) ) {
          V2W(vhh,i,J) = 0.0;
          W(CFL,i) = 0.0;
        } else if (vhrJ < 0.0) {
          double hup = A->hprev[ORC_H3(G,i,j+1,k)] - G->areaT[ORC_H2(G,i,j+1)]*min_h;
          double hlos = max2(0.0, A->vhr[ORC_V3(G,i,J+1,k)]);
          if ((((hup - hlos) + vhrJ) < 0.0) &&
              ((0.5*hup + vhrJ) < 0.0)) {
            V2W(vhh,i,J) = min3(-0.5*hup, -hup+hlos, 0.0);
            DV(A,J,k) = 1;
          } else {
            V2W(vhh,i,J) = vhrJ;
          }
          W(CFL,i) = - V2W(vhh,i,J) / A->hprev[ORC_H3(G,i,j+1,k)];
        } else {
          double hup = A->hprev[ORC_H3(G,i,j,k)] - G->areaT[ORC_H2(G,i,j)]*min_h;
          double hlos = max2(0.0, -A->vhr[ORC_V3(G,i,J-1,k)]);
          if ((((hup - hlos) - vhrJ) < 0.0) &&
              ((0.5*hup - vhrJ) < 0.0)) {
            V2W(vhh,i,J) = max3(0.5*hup, hup-hlos, 0.0);
            DV(A,J,k) = 1;
          } else {
            V2W(vhh,i,J) = vhrJ;
          }
          W(CFL,i) = V2W(vhh,i,J) / A->hprev[ORC_H3(G,i,j,k)];
        }
      }

      if (usePPM) {
        for (int m = 0; m < ntr; m++) for (int i = is; i <= ie; i++) {
          int j_up = (V2W(vhh,i,J) >= 0.0) ? j : j+1;
          double Tp = S3(T_tmp,i,m,j_up+1), Tc = S3(T_tmp,i,m,j_up), Tm = S3(T_tmp,i,m,j_up-1);
          double sm = 0., sc = 0., sp = 0.;
          if (!useHuynh) { sm = S3(slope_y,i,m,j_up-1); sc = S3(slope_y,i,m,j_up); sp = S3(slope_y,i,m,j_up+1); }
          S3(flux_y,i,m,J) = ppm_flux(Tp, Tc, Tm, sm, sc, sp, useHuynh,
                               G->mask2dCv[ORC_V2(G,i,j_up)]*G->mask2dCv[ORC_V2(G,i,j_up-1)],
                               V2W(vhh,i,J), W(CFL,i));
        }
      } else {
        for (int m = 0; m < ntr; m++) for (int i = is; i <= ie; i++) {
          if (V2W(vhh,i,J) >= 0.0) {
            double Tc = S3(T_tmp,i,m,j);
            S3(flux_y,i,m,J) = V2W(vhh,i,J)*( Tc + 0.5 * S3(slope_y,i,m,j) * ( 1. - W(CFL,i) ) );
          } else {
            double Tc = S3(T_tmp,i,m,j+1);
            S3(flux_y,i,m,J) = V2W(vhh,i,J)*( Tc - 0.5 * S3(slope_y,i,m,j+1) * ( 1. - W(CFL,i) ) );
          }
        }
      }
      if (A->OBC && A->OBC->OBC_pe) {      /* :965-1014 */
        const mom6hip_obc_t *OBC = A->OBC;
        if (OBC->specified_v_BCs_exist_globally || OBC->open_v_BCs_exist_globally)
          for (int n = 0; n < OBC->number_of_segments; n++) {
            const mom6hip_obc_segment_t *S = &OBC->segment[n];
            if (!S->specified) continue;
            if (!S->tr_Reg) continue;
            if (S->is_N_or_S && J >= S->JsdB && J <= S->JedB)
              for (int i = S->isd; i <= S->ied; i++) {
                const double v = A->vhr[ORC_V3(G,i,J,k)];
                if (((v > 0.0) && (S->direction == MOM6HIP_OBC_DIRECTION_S)) || ((v < 0.0) && (S->direction == MOM6HIP_OBC_DIRECTION_N))) {
                  V2W(vhh,i,J) = v;
                  for (int q = 0; q < S->ntseg; q++)
                    S3(flux_y,i,S->tr_Reg[q].ntr_index - 1,J) = V2W(vhh,i,J) * seg_tracer(S, &S->tr_Reg[q], i, J, k);
                }
              }
          }
        if (OBC->open_v_BCs_exist_globally)
          for (int n = 0; n < OBC->number_of_segments; n++) {
            const mom6hip_obc_segment_t *S = &OBC->segment[n];
            if (S->specified) continue;
            if (!S->tr_Reg) continue;
            if (S->is_N_or_S && (J >= S->JsdB && J <= S->JedB))
              for (int i = S->isd; i <= S->ied; i++) {
                const double v = A->vhr[ORC_V3(G,i,J,k)];
                if (((v > 0.0) && (G->mask2dT[ORC_H2(G,i,j)] < 0.5)) || ((v < 0.0) && (G->mask2dT[ORC_H2(G,i,j+1)] < 0.5))) {
                  V2W(vhh,i,J) = v;
                  for (int q = 0; q < S->ntseg; q++)
                    S3(flux_y,i,S->tr_Reg[q].ntr_index - 1,J) = V2W(vhh,i,J) * seg_tracer(S, &S->tr_Reg[q], i, J, k);
                }
              }
          }
      }
    } else {
      for (int i = is; i <= ie; i++) V2W(vhh,i,J) = 0.0;
      for (int m = 0; m < ntr; m++) for (int i = is; i <= ie; i++) S3(flux_y,i,m,J) = 0.0;
    }
  }

  /* :1021-1024 */
  for (int J = js-1; J <= je; J++) for (int i = is; i <= ie; i++) {
    double *v = &A->vhr[ORC_V3(G,i,J,k)];
    *v = *v - V2W(vhh,i,J);
    if (fabs(*v) < A->vh_neglect[ORC_V2(G,i,J)]) *v = 0.0;
  }

  /* :1028-1059 */
  for (int j = js; j <= je; j++) if (DJ(j)) {
    for (int i = is; i <= ie; i++) {
      if ((V2W(vhh,i,j) != 0.0) || (V2W(vhh,i,j-1) != 0.0)) {
        double *hp = &A->hprev[ORC_H3(G,i,j,k)];
        double aT = G->areaT[ORC_H2(G,i,j)];
        W(do_i,i) = 1;
        W(hlst,i) = *hp;
        *hp = max2(*hp - (V2W(vhh,i,j) - V2W(vhh,i,j-1)), 0.0);
        if (*hp <= 0.0) { W(do_i,i) = 0; }
        else if (*hp < h_neglect*aT) {
          W(hlst,i) = W(hlst,i) + (h_neglect*aT - *hp);
          W(Ihnew,i) = 1.0 / (h_neglect*aT);
        } else { W(Ihnew,i) = 1.0 / *hp; }
      } else { W(do_i,i) = 0; }
    }
    for (int m = 0; m < ntr; m++) {
      for (int i = is; i <= ie; i++) if (W(do_i,i)) {
        double *t = &A->tr[m][ORC_H3(G,i,j,k)];
        *t = (*t * W(hlst,i) - (S3(flux_y,i,m,j) - S3(flux_y,i,m,j-1))) * W(Ihnew,i);
      }
    }
  }

  /* :1062-1066 */
  for (int m = 0; m < ntr; m++) if (A->conc_underflow && A->conc_underflow[m] > 0.0) {
    for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
      double *t = &A->tr[m][ORC_H3(G,i,j,k)];
      if (fabs(*t) < A->conc_underflow[m]) *t = 0.0;
    }
  }

  free(slope_y); free(flux_y); free(T_tmp); free(vhh); free(hlst); free(Ihnew); free(CFL);
  free(do_j_tr); free(do_i);
#undef S3
#undef V2W
#undef W
#undef DJ
}

/* advect_tracer, :52-324 */
int orc_advect_tracer(const mom6hip_grid_t *G, const double *h_end, const double *uhtr,
                      const double *vhtr, double dt, const mom6hip_tracer_advect_cs_t *cs,
                      double *const *tr, const double *conc_underflow, int ntr,
                      int x_first_in, double *vol_prev, int max_iter_in, int update_vol_prev,
                      double *uhr_out, double *vhr_out, mom6hip_advect_stats_t *stats)
{
  return orc_advect_tracer_obc(G, h_end, uhtr, vhtr, dt, cs, tr, conc_underflow, ntr, x_first_in, vol_prev, max_iter_in, update_vol_prev,
                               uhr_out, vhr_out, stats, NULL);
}

/* advect_tracer with OBC associated: the tracer registries of the segments (reservoirs, inflow concentrations) */
int orc_advect_tracer_obc(const mom6hip_grid_t *G, const double *h_end, const double *uhtr,
                          const double *vhtr, double dt, const mom6hip_tracer_advect_cs_t *cs,
                          double *const *tr, const double *conc_underflow, int ntr,
                          int x_first_in, double *vol_prev, int max_iter_in, int update_vol_prev,
                          double *uhr_out, double *vhr_out, mom6hip_advect_stats_t *stats, const mom6hip_obc_t *OBC)
{
  if (OBC) for (int n = 0; n < OBC->number_of_segments; n++) {
    const mom6hip_obc_segment_t *S = &OBC->segment[n];
    if (S->tr_Reg) for (int q = 0; q < S->ntseg; q++) if (S->tr_Reg[q].ntr_index < 1 || S->tr_Reg[q].ntr_index > ntr) return 3;
  }
  const int is = G->isc, ie = G->iec, js = G->jsc, je = G->jec, nz = G->nk;
  const int isd = G->isd, ied = G->ied, jsd = G->jsd, jed = G->jed;
  const int IsdB = isd-1, IedB = ied, JsdB = jsd-1, JedB = jed;
  const int nih = ORC_NIH(G), njh = ORC_NJH(G);
  const size_t nh3 = (size_t)nih*njh*nz, nu3 = (size_t)(nih+1)*njh*nz, nv3 = (size_t)nih*(njh+1)*nz;
  int stencil = 2;

  if (stats) memset(stats, 0, sizeof(*stats));
  if (ntr == 0) return 0;
  int x_first = ((G->first_direction % 2) == 0);

  const int usePPM = (cs->scheme != MOM6HIP_ADV_PLM);
  const int useHuynh = (cs->scheme == MOM6HIP_ADV_PPM_H3);
  const int use_PPM_stencil = usePPM && !cs->use_huynh_stencil_bug;
  if (use_PPM_stencil) stencil = 3;

  int max_iter = 2*(int)ceil(dt/cs->dt) + 1;
  if (max_iter_in > 0) max_iter = max_iter_in;
  if (x_first_in >= 0) x_first = (x_first_in != 0);

  adv_t A;
  A.G = G; A.ntr = ntr; A.usePPM = usePPM; A.useHuynh = useHuynh; A.tr = tr;
  A.conc_underflow = conc_underflow; A.OBC = OBC;
  A.hprev = calloc(nh3, sizeof(double));
  A.uhr = calloc(nu3, sizeof(double));
  A.vhr = calloc(nv3, sizeof(double));
  A.uh_neglect = calloc((size_t)(nih+1)*njh, sizeof(double));
  A.vh_neglect = calloc((size_t)nih*(njh+1), sizeof(double));
  A.domore_u = calloc((size_t)njh*nz, 1);
  A.domore_v = calloc((size_t)(njh+1)*nz, 1);
  int *domore_k = calloc(nz+1, sizeof(int));
  (void)IsdB; (void)IedB; (void)JsdB; (void)JedB;

  /* :152-178 (uhr, vhr, hprev already zero from calloc) */
  ORC_PAR
  for (int k = 1; k <= nz; k++) {
    domore_k[k] = 1;
    for (int j = js; j <= je; j++) for (int I = is-1; I <= ie; I++)
      A.uhr[ORC_U3(G,I,j,k)] = uhtr[ORC_U3(G,I,j,k)];
    for (int J = js-1; J <= je; J++) for (int i = is; i <= ie; i++)
      A.vhr[ORC_V3(G,i,J,k)] = vhtr[ORC_V3(G,i,J,k)];
    if (!vol_prev) {
      for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
        double aT = G->areaT[ORC_H2(G,i,j)];
        double hp = max2(0.0, aT*h_end[ORC_H3(G,i,j,k)] +
             ((A.uhr[ORC_U3(G,i,j,k)] - A.uhr[ORC_U3(G,i-1,j,k)]) +
              (A.vhr[ORC_V3(G,i,j,k)] - A.vhr[ORC_V3(G,i,j-1,k)])));
        hp = hp + max2(0.0, 1.0e-13*hp - aT*h_end[ORC_H3(G,i,j,k)]);
        A.hprev[ORC_H3(G,i,j,k)] = hp;
      }
    } else {
      for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++)
        A.hprev[ORC_H3(G,i,j,k)] = vol_prev[ORC_H3(G,i,j,k)];
    }
  }

  /* :182-188 */
  for (int j = jsd; j <= jed; j++) for (int i = isd; i <= ied-1; i++)
    A.uh_neglect[ORC_U2(G,i,j)] = G->H_subroundoff * min2(G->areaT[ORC_H2(G,i,j)], G->areaT[ORC_H2(G,i+1,j)]);
  for (int j = jsd; j <= jed-1; j++) for (int i = isd; i <= ied; i++)
    A.vh_neglect[ORC_V2(G,i,j)] = G->H_subroundoff * min2(G->areaT[ORC_H2(G,i,j)], G->areaT[ORC_H2(G,i,j+1)]);

  int isv = is, iev = ie, jsv = js, jev = je;
  int itt, halo_updates = 0, remaining = 0;

  for (itt = 1; itt <= max_iter; itt++) {
    if (isv > is-stencil) {
      /* do_group_pass(CS%pass_uhr_vhr_t_hprev), :206 */
      orc_halo_update(G, A.uhr, MOM6HIP_POS_U, nz);
      orc_halo_update(G, A.vhr, MOM6HIP_POS_V, nz);
      orc_halo_update(G, A.hprev, MOM6HIP_POS_H, nz);
      for (int m = 0; m < ntr; m++) orc_halo_update(G, tr[m], MOM6HIP_POS_H, nz);
      halo_updates++;

      int mh = is-isd;
      if (ied-ie < mh) mh = ied-ie;
      if (js-jsd < mh) mh = js-jsd;
      if (jed-je < mh) mh = jed-je;
      int nsten_halo = mh/stencil;
      isv = is-nsten_halo*stencil ; jsv = js-nsten_halo*stencil;
      iev = ie+nsten_halo*stencil ; jev = je+nsten_halo*stencil;
      if ((nsten_halo > 1) || (itt == 1)) {
        ORC_PAR
        for (int k = 1; k <= nz; k++) if (domore_k[k] > 0) {
          for (int j = jsv; j <= jev; j++) if (!DU(&A,j,k)) {
            for (int i = isv+stencil-1; i <= iev-stencil; i++) if (A.uhr[ORC_U3(G,i,j,k)] != 0.0) {
              DU(&A,j,k) = 1; break;
            }
          }
          for (int J = jsv+stencil-1; J <= jev-stencil; J++) if (!DV(&A,J,k)) {
            for (int i = isv+stencil; i <= iev-stencil; i++) if (A.vhr[ORC_V3(G,i,J,k)] != 0.0) {
              DV(&A,J,k) = 1; break;
            }
          }
          domore_k[k] = 0;
          for (int j = jsv; j <= jev; j++) if (DU(&A,j,k)) domore_k[k] = 1;
          for (int J = jsv+stencil-1; J <= jev-stencil; J++) if (DV(&A,J,k)) domore_k[k] = 1;
        }
      }
    }

    isv = isv + stencil ; iev = iev - stencil;
    jsv = jsv + stencil ; jev = jev - stencil;

    if (x_first) {
      ORC_PAR_DYN      /* the layers are independent (the reference: !$OMP parallel do over k, :228-275) */
      for (int k = 1; k <= nz; k++) if (domore_k[k] > 0)
        advect_x(&A, isv, iev, jsv-stencil, jev+stencil, k);
      ORC_PAR_DYN
      for (int k = 1; k <= nz; k++) if (domore_k[k] > 0) {
        advect_y(&A, isv, iev, jsv, jev, k);
        domore_k[k] = 0;
        for (int j = jsv-stencil; j <= jev+stencil; j++) if (DU(&A,j,k)) domore_k[k] = 1;
        for (int J = jsv-1; J <= jev; J++) if (DV(&A,J,k)) domore_k[k] = 1;
      }
    } else {
      ORC_PAR_DYN
      for (int k = 1; k <= nz; k++) if (domore_k[k] > 0)
        advect_y(&A, isv-stencil, iev+stencil, jsv, jev, k);
      ORC_PAR_DYN
      for (int k = 1; k <= nz; k++) if (domore_k[k] > 0) {
        advect_x(&A, isv, iev, jsv, jev, k);
        domore_k[k] = 0;
        for (int j = jsv; j <= jev; j++) if (DU(&A,j,k)) domore_k[k] = 1;
        for (int J = jsv-1; J <= jev; J++) if (DV(&A,J,k)) domore_k[k] = 1;
      }
    }

    remaining = 0;
    for (int k = 1; k <= nz; k++) remaining += domore_k[k];

    if (itt >= max_iter) break;

    if (isv > is-stencil) {
      /* sum_across_PEs(domore_k) is the identity on one PE, :305 */
      if (remaining == 0) break;
    }
  }
  if (itt > max_iter) itt = max_iter;

  if (uhr_out) memcpy(uhr_out, A.uhr, nu3*sizeof(double));
  if (vhr_out) memcpy(vhr_out, A.vhr, nv3*sizeof(double));
  if (vol_prev && update_vol_prev) memcpy(vol_prev, A.hprev, nh3*sizeof(double));

  if (stats) {
    stats->iterations = itt; stats->halo_updates = halo_updates; stats->domore_remaining = remaining;
  }
  free(A.hprev); free(A.uhr); free(A.vhr); free(A.uh_neglect); free(A.vh_neglect);
  free(A.domore_u); free(A.domore_v); free(domore_k);
  return 0;
}
