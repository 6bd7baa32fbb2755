/*
 * coriolis_adv.c -- CPU restatement of CorAdCalc / gradKE (TEST INFRASTRUCTURE, see mom6_oracle.h).
 *
 * Restates src/core/MOM_CoriolisAdv.F90:125-965 (CorAdCalc) and :969-1051 (gradKE) for the configuration
 * of the hot path: OBC not associated, no Stokes vortex force, porous barriers = 1, no acceleration
 * diagnostics.  All six Coriolis schemes (SADOURNY75_ENERGY (default, with or without CORIOLIS_EN_DIS), ARAKAWA_HSU90,
 * ROBUST_ENSTRO with both PV_ADV_SCHEMEs, SADOURNY75_ENSTRO, ARAKAWA_LAMB81, ARAKAWA_LAMB_BLEND); KE schemes KE_ARAKAWA
 * (default), KE_SIMPLE_GUDONOV, KE_GUDONOV; NOSLIP and BOUND_CORIOLIS.
 *
 * PARITY UNPINNED: the reference holds no known-answer vectors for CorAdCalc.
 */
#include <math.h>
#include <stdlib.h>
#include "mom6_oracle.h"

static inline double max2(double a, double b) { return a > b ? a : b; }
static inline double min2(double a, double b) { return a < b ? a : b; }
static inline double max4(double a, double b, double c, double d) { return max2(max2(max2(a, b), c), d); }
static inline double min4(double a, double b, double c, double d) { return min2(min2(min2(a, b), c), d); }

int orc_coradcalc(const mom6hip_grid_t *G, const mom6hip_coriolisadv_cs_t *CS, const double *u, const double *v,
                  const double *h, const double *uh, const double *vh, double *CAu, double *CAv)
{
  return orc_coradcalc_obc(G, CS, NULL, u, v, h, uh, vh, CAu, CAv);
}

/* segment%tangential_vel / tangential_grad (IsdB:IedB, JsdB:JedB, nk) at the corner point (I, J), layer k (1-based) */
static inline double seg_q(const mom6hip_obc_segment_t *S, const double *f, int I, int J, int k) {
  const long nI = S->IedB - S->IsdB + 1, nJ = S->JedB - S->JsdB + 1;
  return f[(I - S->IsdB) + nI*((J - S->JsdB) + nJ*(long)(k-1))];
}
static inline int imax2(int a, int b) { return a > b ? a : b; }
static inline int imin2(int a, int b) { return a < b ? a : b; }

int orc_coradcalc_obc(const mom6hip_grid_t *G, const mom6hip_coriolisadv_cs_t *CS, const mom6hip_obc_t *OBC, const double *u,
                      const double *v, const double *h, const double *uh, const double *vh, double *CAu, double *CAv)
{
  const int nseg = OBC ? OBC->number_of_segments : 0;
  if (nseg > 0 && !OBC->segment) return 3;
  for (int n = 0; n < nseg; n++) {
    const mom6hip_obc_segment_t *S = &OBC->segment[n];
    if (S->on_pe && OBC->computed_vorticity && !S->tangential_vel) return 3;
    if (S->on_pe && OBC->specified_vorticity && !(S->tangential_grad && G->dxBu && G->dyBu)) return 3;
  }
  const int is = G->isc, ie = G->iec, js = G->jsc, je = G->jec, nz = G->nk;
  const int Isq = G->isc-1, Ieq = G->iec, Jsq = G->jsc-1, Jeq = G->jec;
  const double vol_neglect = G->H_subroundoff * (1e-4 * 1.0)*(1e-4 * 1.0);   /* :241, US%m_to_L = 1 */
  const double C1_12 = 1.0 / 12.0, C1_24 = 1.0 / 24.0;
  const double eps_vel = 1.0e-10 * 1.0, h_tiny = G->Angstrom_H;             /* :242-243, US%m_s_to_L_T = 1 */
  const int sch = CS->coriolis_scheme;
  const int al = (sch == MOM6HIP_ARAKAWA_LAMB81) || (sch == MOM6HIP_AL_BLEND);
  const int en_dis = CS->coriolis_en_dis != 0;
  const int upwind1 = CS->pv_adv_scheme == MOM6HIP_PV_ADV_UPWIND1;
  if (sch < MOM6HIP_SADOURNY75_ENERGY || sch > MOM6HIP_AL_BLEND) return 1;
  const long nH = (long)ORC_NIH(G)*ORC_NJH(G), nU = (long)(ORC_NIH(G)+1)*ORC_NJH(G);
  const long nV = (long)ORC_NIH(G)*(ORC_NJH(G)+1), nQ = (long)(ORC_NIH(G)+1)*(ORC_NJH(G)+1);
  double *Area_h = calloc(nH, 8), *Area_q = calloc(nQ, 8);
#define H2(i,j) ORC_H2(G,i,j)
#define U2(i,j) ORC_U2(G,i,j)
#define V2(i,j) ORC_V2(G,i,j)
#define Q2(i,j) ORC_Q2(G,i,j)
#define H3(i,j,k) ORC_H3(G,i,j,k)
#define U3(i,j,k) ORC_U3(G,i,j,k)
#define V3(i,j,k) ORC_V3(G,i,j,k)

  /* :246-248 */
  for (int j = Jsq-1; j <= Jeq+2; j++) for (int i = Isq-1; i <= Ieq+2; i++)
    Area_h[H2(i,j)] = G->mask2dT[H2(i,j)] * G->areaT[H2(i,j)];
  /* :249-269: the areas of the cells outside a segment are those of the cells inside it */
  for (int n = 0; n < nseg; n++) {
    const mom6hip_obc_segment_t *S = &OBC->segment[n];
    if (!S->on_pe) continue;
    const int I = S->IsdB, J = S->JsdB;
    if (S->is_N_or_S && (J >= Jsq-1) && (J <= Jeq+1)) {
      for (int i = imax2(Isq-1,S->isd); i <= imin2(Ieq+2,S->ied); i++) {
        const int j = J;
        if (S->direction == MOM6HIP_OBC_DIRECTION_N) Area_h[H2(i,j+1)] = Area_h[H2(i,j)];
        else Area_h[H2(i,j)] = Area_h[H2(i,j+1)];
      }
    } else if (S->is_E_or_W && (I >= Isq-1) && (I <= Ieq+1)) {
      for (int j = imax2(Jsq-1,S->jsd); j <= imin2(Jeq+2,S->jed); j++) {
        const int i = I;
        if (S->direction == MOM6HIP_OBC_DIRECTION_E) Area_h[H2(i+1,j)] = Area_h[H2(i,j)];
        else Area_h[H2(i,j)] = Area_h[H2(i+1,j)];
      }
    }
  }
  /* :271-274 */
  for (int J = Jsq-1; J <= Jeq+1; J++) for (int I = Isq-1; I <= Ieq+1; I++)
    Area_q[Q2(I,J)] = (Area_h[H2(I,J)] + Area_h[H2(I+1,J+1)]) + (Area_h[H2(I+1,J)] + Area_h[H2(I,J+1)]);

  /* the layers are independent (the reference: !$OMP parallel do over k, :281-284); the 2-D work arrays are per thread */
  _Pragma("omp parallel")
  {
  double *q = calloc(nQ, 8), *Ih_q = calloc(nQ, 8);
  double *abs_vort = calloc(nQ, 8), *dvdx = calloc(nQ, 8), *dudy = calloc(nQ, 8), *rel_vort = calloc(nQ, 8);
  double *hArea_u = calloc(nU, 8), *hArea_v = calloc(nV, 8), *KE = calloc(nH, 8), *KEx = calloc(nU, 8), *KEy = calloc(nV, 8);
  double *a = calloc(nU, 8), *b = calloc(nU, 8), *c = calloc(nU, 8), *d = calloc(nU, 8);
  double *ep_u = calloc(nH, 8), *ep_v = calloc(nH, 8);
  double *uh_min = calloc(nU, 8), *uh_max = calloc(nU, 8), *vh_min = calloc(nV, 8), *vh_max = calloc(nV, 8);
  double *uh_center = calloc(nU, 8), *vh_center = calloc(nV, 8);
  _Pragma("omp for schedule(static)")
  for (int k = 1; k <= nz; k++) {
    /* :314-324 */
    for (int J = Jsq-1; J <= Jeq+1; J++) for (int I = Isq-1; I <= Ieq+1; I++) {
      const int i = I, j = J;
      dvdx[Q2(I,J)] = (v[V3(i+1,J,k)]*G->dyCv[V2(i+1,J)] - v[V3(i,J,k)]*G->dyCv[V2(i,J)]);
      dudy[Q2(I,J)] = (u[U3(I,j+1,k)]*G->dxCu[U2(I,j+1)] - u[U3(I,j,k)]*G->dxCu[U2(I,j)]);
    }
    for (int J = Jsq-1; J <= Jeq+1; J++) for (int i = Isq-1; i <= Ieq+2; i++) {
      const int j = J;
      hArea_v[V2(i,J)] = 0.5*(Area_h[H2(i,j)] * h[H3(i,j,k)] + Area_h[H2(i,j+1)] * h[H3(i,j+1,k)]);
    }
    for (int j = Jsq-1; j <= Jeq+2; j++) for (int I = Isq-1; I <= Ieq+1; I++) {
      const int i = I;
      hArea_u[U2(I,j)] = 0.5*(Area_h[H2(i,j)] * h[H3(i,j,k)] + Area_h[H2(i+1,j)] * h[H3(i+1,j,k)]);
    }
    if (en_dis) {                                               /* :326-333 */
      for (int j = Jsq; j <= Jeq+1; j++) for (int I = is-1; I <= ie; I++) {
        const int i = I;
        uh_center[U2(I,j)] = 0.5 * ((G->dy_Cu[U2(I,j)]*1.0) * u[U3(I,j,k)]) * (h[H3(i,j,k)] + h[H3(i+1,j,k)]);
      }
      for (int J = js-1; J <= je; J++) for (int i = Isq; i <= Ieq+1; i++) {
        const int j = J;
        vh_center[V2(i,J)] = 0.5 * ((G->dx_Cv[V2(i,J)]*1.0) * v[V3(i,J,k)]) * (h[H3(i,j,k)] + h[H3(i,j+1,k)]);
      }
    }
    /* :337-420: the circulation and the thicknesses projected onto the velocity points of the open boundaries */
    for (int n = 0; n < nseg; n++) {
      const mom6hip_obc_segment_t *S = &OBC->segment[n];
      if (!S->on_pe) continue;
      if (S->is_N_or_S && (S->JsdB >= Jsq-1) && (S->JsdB <= Jeq+1)) {
        const int J = S->JsdB, j = J;
        if (OBC->zero_vorticity) for (int I = S->IsdB; I <= S->IedB; I++) { dvdx[Q2(I,J)] = 0.; dudy[Q2(I,J)] = 0.; }
        if (OBC->freeslip_vorticity) for (int I = S->IsdB; I <= S->IedB; I++) dudy[Q2(I,J)] = 0.;
        if (OBC->computed_vorticity) for (int I = S->IsdB; I <= S->IedB; I++) {
          if (S->direction == MOM6HIP_OBC_DIRECTION_N) dudy[Q2(I,J)] = 2.0*(seg_q(S, S->tangential_vel, I, J, k) - u[U3(I,j,k)])*G->dxCu[U2(I,j)];
          else dudy[Q2(I,J)] = 2.0*(u[U3(I,j+1,k)] - seg_q(S, S->tangential_vel, I, J, k))*G->dxCu[U2(I,j+1)];
        }
        if (OBC->specified_vorticity) for (int I = S->IsdB; I <= S->IedB; I++) {
          if (S->direction == MOM6HIP_OBC_DIRECTION_N) dudy[Q2(I,J)] = seg_q(S, S->tangential_grad, I, J, k)*G->dxCu[U2(I,j)]*G->dyBu[Q2(I,J)];
          else dudy[Q2(I,J)] = seg_q(S, S->tangential_grad, I, J, k)*G->dxCu[U2(I,j+1)]*G->dyBu[Q2(I,J)];
        }
        for (int i = imax2(Isq-1,S->isd); i <= imin2(Ieq+2,S->ied); i++) {
          if (S->direction == MOM6HIP_OBC_DIRECTION_N) hArea_v[V2(i,J)] = 0.5 * (Area_h[H2(i,j)] + Area_h[H2(i,j+1)]) * h[H3(i,j,k)];
          else hArea_v[V2(i,J)] = 0.5 * (Area_h[H2(i,j)] + Area_h[H2(i,j+1)]) * h[H3(i,j+1,k)];
        }
        if (en_dis) for (int i = imax2(Isq-1,S->isd); i <= imin2(Ieq+2,S->ied); i++) {
          if (S->direction == MOM6HIP_OBC_DIRECTION_N) vh_center[V2(i,J)] = (G->dx_Cv[V2(i,J)]*1.0) * v[V3(i,J,k)] * h[H3(i,j,k)];
          else vh_center[V2(i,J)] = (G->dx_Cv[V2(i,J)]*1.0) * v[V3(i,J,k)] * h[H3(i,j+1,k)];
        }
      } else if (S->is_E_or_W && (S->IsdB >= Isq-1) && (S->IsdB <= Ieq+1)) {
        const int I = S->IsdB, i = I;
        if (OBC->zero_vorticity) for (int J = S->JsdB; J <= S->JedB; J++) { dvdx[Q2(I,J)] = 0.; dudy[Q2(I,J)] = 0.; }
        if (OBC->freeslip_vorticity) for (int J = S->JsdB; J <= S->JedB; J++) dvdx[Q2(I,J)] = 0.;
        if (OBC->computed_vorticity) for (int J = S->JsdB; J <= S->JedB; J++) {
          if (S->direction == MOM6HIP_OBC_DIRECTION_E) dvdx[Q2(I,J)] = 2.0*(seg_q(S, S->tangential_vel, I, J, k) - v[V3(i,J,k)])*G->dyCv[V2(i,J)];
          else dvdx[Q2(I,J)] = 2.0*(v[V3(i+1,J,k)] - seg_q(S, S->tangential_vel, I, J, k))*G->dyCv[V2(i+1,J)];
        }
        if (OBC->specified_vorticity) for (int J = S->JsdB; J <= S->JedB; J++) {
          if (S->direction == MOM6HIP_OBC_DIRECTION_E) dvdx[Q2(I,J)] = seg_q(S, S->tangential_grad, I, J, k)*G->dyCv[V2(i,J)]*G->dxBu[Q2(I,J)];
          else dvdx[Q2(I,J)] = seg_q(S, S->tangential_grad, I, J, k)*G->dyCv[V2(i+1,J)]*G->dxBu[Q2(I,J)];
        }
        for (int j = imax2(Jsq-1,S->jsd); j <= imin2(Jeq+2,S->jed); j++) {
          if (S->direction == MOM6HIP_OBC_DIRECTION_E) hArea_u[U2(I,j)] = 0.5*(Area_h[H2(i,j)] + Area_h[H2(i+1,j)]) * h[H3(i,j,k)];
          else hArea_u[U2(I,j)] = 0.5*(Area_h[H2(i,j)] + Area_h[H2(i+1,j)]) * h[H3(i+1,j,k)];
        }
        if (en_dis) for (int j = imax2(Jsq-1,S->jsd); j <= imin2(Jeq+2,S->jed); j++) {
          if (S->direction == MOM6HIP_OBC_DIRECTION_E) uh_center[U2(I,j)] = (G->dy_Cu[U2(I,j)]*1.0) * u[U3(I,j,k)] * h[H3(i,j,k)];
          else uh_center[U2(I,j)] = (G->dy_Cu[U2(I,j)]*1.0) * u[U3(I,j,k)] * h[H3(i+1,j,k)];
        }
      }
    }
    /* :422-455: then onto the corner points of the open boundaries (in sequence: the two projections cannot be combined) */
    for (int n = 0; n < nseg; n++) {
      const mom6hip_obc_segment_t *S = &OBC->segment[n];
      if (!S->on_pe) continue;
      if (S->is_N_or_S && (S->JsdB >= Jsq-1) && (S->JsdB <= Jeq+1)) {
        const int J = S->JsdB, j = J;
        for (int I = imax2(Isq-1,S->IsdB); I <= imin2(Ieq+1,S->IedB); I++) {
          const int i = I;
          if (S->direction == MOM6HIP_OBC_DIRECTION_N) {
            if (Area_h[H2(i,j)] + Area_h[H2(i+1,j)] > 0.0)
              hArea_u[U2(I,j+1)] = hArea_u[U2(I,j)] * ((Area_h[H2(i,j+1)] + Area_h[H2(i+1,j+1)]) / (Area_h[H2(i,j)] + Area_h[H2(i+1,j)]));
            else hArea_u[U2(I,j+1)] = 0.0;
          } else {
            if (Area_h[H2(i,j+1)] + Area_h[H2(i+1,j+1)] > 0.0)
              hArea_u[U2(I,j)] = hArea_u[U2(I,j+1)] * ((Area_h[H2(i,j)] + Area_h[H2(i+1,j)]) / (Area_h[H2(i,j+1)] + Area_h[H2(i+1,j+1)]));
            else hArea_u[U2(I,j)] = 0.0;
          }
        }
      } else if (S->is_E_or_W && (S->IsdB >= Isq-1) && (S->IsdB <= Ieq+1)) {
        const int I = S->IsdB, i = I;
        for (int J = imax2(Jsq-1,S->JsdB); J <= imin2(Jeq+1,S->JedB); J++) {
          const int j = J;
          if (S->direction == MOM6HIP_OBC_DIRECTION_E) {
            if (Area_h[H2(i,j)] + Area_h[H2(i,j+1)] > 0.0)
              hArea_v[V2(i+1,J)] = hArea_v[V2(i,J)] * ((Area_h[H2(i+1,j)] + Area_h[H2(i+1,j+1)]) / (Area_h[H2(i,j)] + Area_h[H2(i,j+1)]));
            else hArea_v[V2(i+1,J)] = 0.0;
          } else {
            hArea_v[V2(i,J)] = 0.5 * (Area_h[H2(i,j)] + Area_h[H2(i,j+1)]) * h[H3(i,j+1,k)];      /* (:449, overwritten below) */
            if (Area_h[H2(i+1,j)] + Area_h[H2(i+1,j+1)] > 0.0)
              hArea_v[V2(i,J)] = hArea_v[V2(i+1,J)] * ((Area_h[H2(i,j)] + Area_h[H2(i,j+1)]) / (Area_h[H2(i+1,j)] + Area_h[H2(i+1,j+1)]));
            else hArea_v[V2(i,J)] = 0.0;
          }
        }
      }
    }
    /* :459-473 */
    for (int J = Jsq-1; J <= Jeq+1; J++) for (int I = Isq-1; I <= Ieq+1; I++) {
      if (CS->no_slip)
        rel_vort[Q2(I,J)] = (2.0 - G->mask2dBu[Q2(I,J)]) * (dvdx[Q2(I,J)] - dudy[Q2(I,J)]) * G->IareaBu[Q2(I,J)];
      else
        rel_vort[Q2(I,J)] = G->mask2dBu[Q2(I,J)] * (dvdx[Q2(I,J)] - dudy[Q2(I,J)]) * G->IareaBu[Q2(I,J)];
    }
    /* :483-491 */
    for (int J = Jsq-1; J <= Jeq+1; J++) for (int I = Isq-1; I <= Ieq+1; I++)
      abs_vort[Q2(I,J)] = G->CoriolisBu[Q2(I,J)] + rel_vort[Q2(I,J)];
    for (int J = Jsq-1; J <= Jeq+1; J++) for (int I = Isq-1; I <= Ieq+1; I++) {
      const int i = I, j = J;
      double hArea_q = (hArea_u[U2(I,j)] + hArea_u[U2(I,j+1)]) + (hArea_v[V2(i,J)] + hArea_v[V2(i+1,J)]);
      Ih_q[Q2(I,J)] = Area_q[Q2(I,J)] / (hArea_q + vol_neglect);
      q[Q2(I,J)] = abs_vort[Q2(I,J)] * Ih_q[Q2(I,J)];
    }
    /* :523-533 */
    if (sch == MOM6HIP_ARAKAWA_HSU90) {
      for (int j = Jsq; j <= Jeq+1; j++) {
        const int J = j;
        for (int I = is-1; I <= Ieq; I++) {
          a[U2(I,j)] = (q[Q2(I,J)] + (q[Q2(I+1,J)] + q[Q2(I,J-1)])) * C1_12;
          d[U2(I,j)] = ((q[Q2(I,J)] + q[Q2(I+1,J-1)]) + q[Q2(I,J-1)]) * C1_12;
        }
        for (int I = Isq; I <= Ieq; I++) {
          b[U2(I,j)] = (q[Q2(I,J)] + (q[Q2(I-1,J)] + q[Q2(I,J-1)])) * C1_12;
          c[U2(I,j)] = ((q[Q2(I,J)] + q[Q2(I-1,J-1)]) + q[Q2(I,J-1)]) * C1_12;
        }
      }
    }
    else if (sch == MOM6HIP_ARAKAWA_LAMB81) {                    /* :534-542 */
      for (int j = Jsq; j <= Jeq+1; j++) for (int I = Isq; I <= Ieq+1; I++) {
        const int J = j, i = I;
        a[U2(I-1,j)] = (2.0*(q[Q2(I,J)] + q[Q2(I-1,J-1)]) + (q[Q2(I-1,J)] + q[Q2(I,J-1)])) * C1_24;
        d[U2(I-1,j)] = ((q[Q2(I,J)] + q[Q2(I-1,J-1)]) + 2.0*(q[Q2(I-1,J)] + q[Q2(I,J-1)])) * C1_24;
        b[U2(I,j)] =   ((q[Q2(I,J)] + q[Q2(I-1,J-1)]) + 2.0*(q[Q2(I-1,J)] + q[Q2(I,J-1)])) * C1_24;
        c[U2(I,j)] =   (2.0*(q[Q2(I,J)] + q[Q2(I-1,J-1)]) + (q[Q2(I-1,J)] + q[Q2(I,J-1)])) * C1_24;
        ep_u[H2(i,j)] = ((q[Q2(I,J)] - q[Q2(I-1,J-1)]) + (q[Q2(I-1,J)] - q[Q2(I,J-1)])) * C1_24;
        ep_v[H2(i,j)] = (-(q[Q2(I,J)] - q[Q2(I-1,J-1)]) + (q[Q2(I-1,J)] - q[Q2(I,J-1)])) * C1_24;
      }
    } else if (sch == MOM6HIP_AL_BLEND) {                        /* :543-590 */
      double Fe_m2 = CS->F_eff_max_blend - 2.0;
      double rat_lin = 1.5 * Fe_m2 / max2(CS->wt_lin_blend, 1.0e-16);
      if (CS->F_eff_max_blend <= 2.0) { Fe_m2 = -1.; rat_lin = -1.0; }
      for (int j = Jsq; j <= Jeq+1; j++) for (int I = Isq; I <= Ieq+1; I++) {
        const int J = j, i = I;
        const double min_Ihq = min4(Ih_q[Q2(I-1,J-1)], Ih_q[Q2(I,J-1)], Ih_q[Q2(I-1,J)], Ih_q[Q2(I,J)]);
        const double max_Ihq = max4(Ih_q[Q2(I-1,J-1)], Ih_q[Q2(I,J-1)], Ih_q[Q2(I-1,J)], Ih_q[Q2(I,J)]);
        double rat_m1 = 1.0e15;
        if (max_Ihq < 1.0e15*min_Ihq) rat_m1 = max_Ihq / min_Ihq - 1.0;
        double AL_wt, Sad_wt;
        if (rat_m1 <= Fe_m2) AL_wt = 1.0;
        else if (rat_m1 < 1.5*Fe_m2) AL_wt = 3.0*Fe_m2 / rat_m1 - 2.0;
        else AL_wt = 0.0;
        if (rat_m1 <= 1.5*Fe_m2) Sad_wt = 0.0;
        else if (rat_m1 <= rat_lin) Sad_wt = 1.0 - (1.5*Fe_m2) / rat_m1;
        else if (rat_m1 < 2.0*rat_lin) Sad_wt = 1.0 - (CS->wt_lin_blend / rat_lin) * (rat_m1 - 2.0*rat_lin);
        else Sad_wt = 1.0;
        a[U2(I-1,j)] = Sad_wt * 0.25 * q[Q2(I-1,J)] + (1.0 - Sad_wt) *
                   ( ((2.0-AL_wt)* q[Q2(I-1,J)] + AL_wt*q[Q2(I,J-1)]) +
                      2.0 * (q[Q2(I,J)] + q[Q2(I-1,J-1)]) ) * C1_24;
        d[U2(I-1,j)] = Sad_wt * 0.25 * q[Q2(I-1,J-1)] + (1.0 - Sad_wt) *
                   ( ((2.0-AL_wt)* q[Q2(I-1,J-1)] + AL_wt*q[Q2(I,J)]) +
                      2.0 * (q[Q2(I-1,J)] + q[Q2(I,J-1)]) ) * C1_24;
        b[U2(I,j)] =   Sad_wt * 0.25 * q[Q2(I,J)] + (1.0 - Sad_wt) *
                   ( ((2.0-AL_wt)* q[Q2(I,J)] + AL_wt*q[Q2(I-1,J-1)]) +
                      2.0 * (q[Q2(I-1,J)] + q[Q2(I,J-1)]) ) * C1_24;
        c[U2(I,j)] =   Sad_wt * 0.25 * q[Q2(I,J-1)] + (1.0 - Sad_wt) *
                   ( ((2.0-AL_wt)* q[Q2(I,J-1)] + AL_wt*q[Q2(I-1,J)]) +
                      2.0 * (q[Q2(I,J)] + q[Q2(I-1,J-1)]) ) * C1_24;
        ep_u[H2(i,j)] = AL_wt  * ((q[Q2(I,J)] - q[Q2(I-1,J-1)]) + (q[Q2(I-1,J)] - q[Q2(I,J-1)])) * C1_24;
        ep_v[H2(i,j)] = AL_wt * (-(q[Q2(I,J)] - q[Q2(I-1,J-1)]) + (q[Q2(I-1,J)] - q[Q2(I,J-1)])) * C1_24;
      }
    }
    if (en_dis) {                                               /* :326-333, :594-642 */
      const double c1 = 1.0-1.5*0.5, c2 = 1.0-0.5, c3 = 2.0, slope = 0.5;
      for (int j = Jsq; j <= Jeq+1; j++) for (int I = is-1; I <= ie; I++) {
        const int i = I;
        (void)i;
        double uhc = uh_center[U2(I,j)];
        double uhm = uh[U3(I,j,k)];
        if (G->dy_Cu[U2(I,j)] == 0.0) uhc = uhm;
        if (fabs(uhc) < 0.1*fabs(uhm)) {
          uhm = 10.0*uhc;
        } else if (fabs(uhc) > c1*fabs(uhm)) {
          if (fabs(uhc) < c2*fabs(uhm)) uhc = (3.0*uhc+(1.0-c2*3.0)*uhm);
          else if (fabs(uhc) <= c3*fabs(uhm)) uhc = uhm;
          else uhc = slope*uhc+(1.0-c3*slope)*uhm;
        }
        if (uhc > uhm) { uh_min[U2(I,j)] = uhm; uh_max[U2(I,j)] = uhc; }
        else { uh_max[U2(I,j)] = uhm; uh_min[U2(I,j)] = uhc; }
      }
      for (int J = js-1; J <= je; J++) for (int i = Isq; i <= Ieq+1; i++) {
        const int j = J;
        (void)j;
        double vhc = vh_center[V2(i,J)];
        double vhm = vh[V3(i,J,k)];
        if (G->dx_Cv[V2(i,J)] == 0.0) vhc = vhm;
        if (fabs(vhc) < 0.1*fabs(vhm)) {
          vhm = 10.0*vhc;
        } else if (fabs(vhc) > c1*fabs(vhm)) {
          if (fabs(vhc) < c2*fabs(vhm)) vhc = (3.0*vhc+(1.0-c2*3.0)*vhm);
          else if (fabs(vhc) <= c3*fabs(vhm)) vhc = vhm;
          else vhc = slope*vhc+(1.0-c3*slope)*vhm;
        }
        if (vhc > vhm) { vh_min[V2(i,J)] = vhm; vh_max[V2(i,J)] = vhc; }
        else { vh_max[V2(i,J)] = vhm; vh_min[V2(i,J)] = vhc; }
      }
    }
    /* gradKE, :994-1035 */
    for (int j = Jsq; j <= Jeq+1; j++) for (int i = Isq; i <= Ieq+1; i++) {
      const int I = i, J = j;
      if (CS->ke_scheme == MOM6HIP_KE_ARAKAWA) {
        KE[H2(i,j)] = ( ( G->areaCu[U2(I,j)]*(u[U3(I,j,k)]*u[U3(I,j,k)]) +
                          G->areaCu[U2(I-1,j)]*(u[U3(I-1,j,k)]*u[U3(I-1,j,k)]) ) +
                        ( G->areaCv[V2(i,J)]*(v[V3(i,J,k)]*v[V3(i,J,k)]) +
                          G->areaCv[V2(i,J-1)]*(v[V3(i,J-1,k)]*v[V3(i,J-1,k)]) ) )*0.25*G->IareaT[H2(i,j)];
      } else {
        double up = 0.5*( u[U3(I-1,j,k)] + fabs( u[U3(I-1,j,k)] ) );
        double um = 0.5*( u[U3(I,j,k)] - fabs( u[U3(I,j,k)] ) );
        double vp = 0.5*( v[V3(i,J-1,k)] + fabs( v[V3(i,J-1,k)] ) );
        double vm = 0.5*( v[V3(i,J,k)] - fabs( v[V3(i,J,k)] ) );
        if (CS->ke_scheme == MOM6HIP_KE_SIMPLE_GUDONOV) {
          double up2 = up*up, um2 = um*um, vp2 = vp*vp, vm2 = vm*vm;
          KE[H2(i,j)] = ( max2(up2,um2) + max2(vp2,vm2) ) *0.5;
        } else {
          double up2a = up*up*G->areaCu[U2(I-1,j)], um2a = um*um*G->areaCu[U2(I,j)];
          double vp2a = vp*vp*G->areaCv[V2(i,J-1)], vm2a = vm*vm*G->areaCv[V2(i,J)];
          KE[H2(i,j)] = ( max2(um2a,up2a) + max2(vm2a,vp2a) )*0.5*G->IareaT[H2(i,j)];
        }
      }
    }
    for (int j = js; j <= je; j++) for (int I = Isq; I <= Ieq; I++)
      KEx[U2(I,j)] = (KE[H2(I+1,j)] - KE[H2(I,j)]) * G->IdxCu[U2(I,j)];
    for (int J = Jsq; J <= Jeq; J++) for (int i = is; i <= ie; i++)
      KEy[V2(i,J)] = (KE[H2(i,J+1)] - KE[H2(i,J)]) * G->IdyCv[V2(i,J)];
    for (int n = 0; n < nseg; n++) {                              /* gradKE :1037-1050 */
      const mom6hip_obc_segment_t *S = &OBC->segment[n];
      if (S->is_N_or_S) { for (int i = S->isd; i <= S->ied; i++) KEy[V2(i,S->JsdB)] = 0.; }
      else if (S->is_E_or_W) { for (int j = S->jsd; j <= S->jed; j++) KEx[U2(S->IsdB,j)] = 0.; }
    }

    /* CAu, :644-752 */
    for (int j = js; j <= je; j++) for (int I = Isq; I <= Ieq; I++) {
      const int i = I, J = j;
      double ca;
      if (sch == MOM6HIP_SADOURNY75_ENERGY && en_dis) {        /* :645-665; vh_max(i,j) is the face (i, J=j) */
        double temp1, temp2;
        if (q[Q2(I,J)]*u[U3(I,j,k)] == 0.0)
          temp1 = q[Q2(I,J)] * ( (vh_max[V2(i,j)]+vh_max[V2(i+1,j)])
                               + (vh_min[V2(i,j)]+vh_min[V2(i+1,j)]) )*0.5;
        else if (q[Q2(I,J)]*u[U3(I,j,k)] < 0.0)
          temp1 = q[Q2(I,J)] * (vh_max[V2(i,j)]+vh_max[V2(i+1,j)]);
        else
          temp1 = q[Q2(I,J)] * (vh_min[V2(i,j)]+vh_min[V2(i+1,j)]);
        if (q[Q2(I,J-1)]*u[U3(I,j,k)] == 0.0)
          temp2 = q[Q2(I,J-1)] * ( (vh_max[V2(i,j-1)]+vh_max[V2(i+1,j-1)])
                                 + (vh_min[V2(i,j-1)]+vh_min[V2(i+1,j-1)]) )*0.5;
        else if (q[Q2(I,J-1)]*u[U3(I,j,k)] < 0.0)
          temp2 = q[Q2(I,J-1)] * (vh_max[V2(i,j-1)]+vh_max[V2(i+1,j-1)]);
        else
          temp2 = q[Q2(I,J-1)] * (vh_min[V2(i,j-1)]+vh_min[V2(i+1,j-1)]);
        ca = 0.25 * G->IdxCu[U2(I,j)] * (temp1 + temp2);
      } else if (sch == MOM6HIP_SADOURNY75_ENERGY) {
        ca = 0.25 *
            (q[Q2(I,J)] * (vh[V3(i+1,J,k)] + vh[V3(i,J,k)]) +
             q[Q2(I,J-1)] * (vh[V3(i,J-1,k)] + vh[V3(i+1,J-1,k)])) * G->IdxCu[U2(I,j)];
      } else if (sch == MOM6HIP_SADOURNY75_ENSTRO) {
        ca = 0.125 * (G->IdxCu[U2(I,j)] * (q[Q2(I,J)] + q[Q2(I,J-1)])) *
                     ((vh[V3(i+1,J,k)] + vh[V3(i,J,k)]) + (vh[V3(i,J-1,k)] + vh[V3(i+1,J-1,k)]));
      } else if (sch == MOM6HIP_ROBUST_ENSTRO) {                 /* :687-714 */
        double Heff1 = fabs(vh[V3(i,J,k)] * G->IdxCv[V2(i,J)]) / (eps_vel+fabs(v[V3(i,J,k)]));
        Heff1 = max2(Heff1, min2(h[H3(i,j,k)],h[H3(i,j+1,k)]));
        Heff1 = min2(Heff1, max2(h[H3(i,j,k)],h[H3(i,j+1,k)]));
        double Heff2 = fabs(vh[V3(i,J-1,k)] * G->IdxCv[V2(i,J-1)]) / (eps_vel+fabs(v[V3(i,J-1,k)]));
        Heff2 = max2(Heff2, min2(h[H3(i,j-1,k)],h[H3(i,j,k)]));
        Heff2 = min2(Heff2, max2(h[H3(i,j-1,k)],h[H3(i,j,k)]));
        double Heff3 = fabs(vh[V3(i+1,J,k)] * G->IdxCv[V2(i+1,J)]) / (eps_vel+fabs(v[V3(i+1,J,k)]));
        Heff3 = max2(Heff3, min2(h[H3(i+1,j,k)],h[H3(i+1,j+1,k)]));
        Heff3 = min2(Heff3, max2(h[H3(i+1,j,k)],h[H3(i+1,j+1,k)]));
        double Heff4 = fabs(vh[V3(i+1,J-1,k)] * G->IdxCv[V2(i+1,J-1)]) / (eps_vel+fabs(v[V3(i+1,J-1,k)]));
        Heff4 = max2(Heff4, min2(h[H3(i+1,j-1,k)],h[H3(i+1,j,k)]));
        Heff4 = min2(Heff4, max2(h[H3(i+1,j-1,k)],h[H3(i+1,j,k)]));
        if (!upwind1) {
          ca = 0.5*(abs_vort[Q2(I,J)]+abs_vort[Q2(I,J-1)]) *
                       ((vh[V3(i,J,k)] + vh[V3(i+1,J-1,k)]) + (vh[V3(i,J-1,k)] + vh[V3(i+1,J,k)]) ) /
                       (h_tiny + ((Heff1+Heff4) + (Heff2+Heff3)) ) * G->IdxCu[U2(I,j)];
        } else {
          const double VHeff = ((vh[V3(i,J,k)] + vh[V3(i+1,J-1,k)]) + (vh[V3(i,J-1,k)] + vh[V3(i+1,J,k)]) );
          const double QVHeff = 0.5*( (abs_vort[Q2(I,J)]+abs_vort[Q2(I,J-1)])*VHeff
                                     -(abs_vort[Q2(I,J)]-abs_vort[Q2(I,J-1)])*fabs(VHeff) );
          ca = (QVHeff / ( h_tiny + ((Heff1+Heff4) + (Heff2+Heff3)) ) ) * G->IdxCu[U2(I,j)];
        }
      } else {
        ca = ((a[U2(I,j)] * vh[V3(i+1,J,k)] +  c[U2(I,j)] * vh[V3(i,J-1,k)])  +
              (b[U2(I,j)] * vh[V3(i,J,k)] +  d[U2(I,j)] * vh[V3(i+1,J-1,k)])) * G->IdxCu[U2(I,j)];
      }
      if (al)                                                    /* :717-721 */
        ca = ca + (ep_u[H2(i,j)]*uh[U3(I-1,j,k)] - ep_u[H2(i+1,j)]*uh[U3(I+1,j,k)]) * G->IdxCu[U2(I,j)];
      if (CS->bound_coriolis) {
        double fv1 = abs_vort[Q2(I,J)] * v[V3(i+1,J,k)];
        double fv2 = abs_vort[Q2(I,J)] * v[V3(i,J,k)];
        double fv3 = abs_vort[Q2(I,J-1)] * v[V3(i+1,J-1,k)];
        double fv4 = abs_vort[Q2(I,J-1)] * v[V3(i,J-1,k)];
        double max_fv = max4(fv1, fv2, fv3, fv4), min_fv = min4(fv1, fv2, fv3, fv4);
        ca = min2(ca, max_fv);
        ca = max2(ca, min_fv);
      }
      CAu[U3(I,j,k)] = ca - KEx[U2(I,j)];
    }
    /* CAv, :763-876 */
    for (int J = Jsq; J <= Jeq; J++) for (int i = is; i <= ie; i++) {
      const int I = i, j = J;
      double ca;
      if (sch == MOM6HIP_SADOURNY75_ENERGY && en_dis) {        /* :764-785; uh_max(i-1,j) is the face (I-1, j) */
        double temp1, temp2;
        if (q[Q2(I-1,J)]*v[V3(i,J,k)] == 0.0)
          temp1 = q[Q2(I-1,J)] * ( (uh_max[U2(i-1,j)]+uh_max[U2(i-1,j+1)])
                                 + (uh_min[U2(i-1,j)]+uh_min[U2(i-1,j+1)]) )*0.5;
        else if (q[Q2(I-1,J)]*v[V3(i,J,k)] > 0.0)
          temp1 = q[Q2(I-1,J)] * (uh_max[U2(i-1,j)]+uh_max[U2(i-1,j+1)]);
        else
          temp1 = q[Q2(I-1,J)] * (uh_min[U2(i-1,j)]+uh_min[U2(i-1,j+1)]);
        if (q[Q2(I,J)]*v[V3(i,J,k)] == 0.0)
          temp2 = q[Q2(I,J)] * ( (uh_max[U2(i,j)]+uh_max[U2(i,j+1)])
                               + (uh_min[U2(i,j)]+uh_min[U2(i,j+1)]) )*0.5;
        else if (q[Q2(I,J)]*v[V3(i,J,k)] > 0.0)
          temp2 = q[Q2(I,J)] * (uh_max[U2(i,j)]+uh_max[U2(i,j+1)]);
        else
          temp2 = q[Q2(I,J)] * (uh_min[U2(i,j)]+uh_min[U2(i,j+1)]);
        ca = -0.25 * G->IdyCv[V2(i,J)] * (temp1 + temp2);
      } else if (sch == MOM6HIP_SADOURNY75_ENERGY) {
        ca = - 0.25*
              (q[Q2(I-1,J)]*(uh[U3(I-1,j,k)] + uh[U3(I-1,j+1,k)]) +
               q[Q2(I,J)]*(uh[U3(I,j,k)] + uh[U3(I,j+1,k)])) * G->IdyCv[V2(i,J)];
      } else if (sch == MOM6HIP_SADOURNY75_ENSTRO) {
        ca = -0.125 * (G->IdyCv[V2(i,J)] * (q[Q2(I-1,J)] + q[Q2(I,J)])) *
                     ((uh[U3(I-1,j,k)] + uh[U3(I-1,j+1,k)]) + (uh[U3(I,j,k)] + uh[U3(I,j+1,k)]));
      } else if (sch == MOM6HIP_ROBUST_ENSTRO) {                 /* :808-838 */
        double Heff1 = fabs(uh[U3(I,j,k)] * G->IdyCu[U2(I,j)]) / (eps_vel+fabs(u[U3(I,j,k)]));
        Heff1 = max2(Heff1, min2(h[H3(i,j,k)],h[H3(i+1,j,k)]));
        Heff1 = min2(Heff1, max2(h[H3(i,j,k)],h[H3(i+1,j,k)]));
        double Heff2 = fabs(uh[U3(I-1,j,k)] * G->IdyCu[U2(I-1,j)]) / (eps_vel+fabs(u[U3(I-1,j,k)]));
        Heff2 = max2(Heff2, min2(h[H3(i-1,j,k)],h[H3(i,j,k)]));
        Heff2 = min2(Heff2, max2(h[H3(i-1,j,k)],h[H3(i,j,k)]));
        double Heff3 = fabs(uh[U3(I,j+1,k)] * G->IdyCu[U2(I,j+1)]) / (eps_vel+fabs(u[U3(I,j+1,k)]));
        Heff3 = max2(Heff3, min2(h[H3(i,j+1,k)],h[H3(i+1,j+1,k)]));
        Heff3 = min2(Heff3, max2(h[H3(i,j+1,k)],h[H3(i+1,j+1,k)]));
        double Heff4 = fabs(uh[U3(I-1,j+1,k)] * G->IdyCu[U2(I-1,j+1)]) / (eps_vel+fabs(u[U3(I-1,j+1,k)]));
        Heff4 = max2(Heff4, min2(h[H3(i-1,j+1,k)],h[H3(i,j+1,k)]));
        Heff4 = min2(Heff4, max2(h[H3(i-1,j+1,k)],h[H3(i,j+1,k)]));
        if (!upwind1) {
          ca = - 0.5*(abs_vort[Q2(I,J)]+abs_vort[Q2(I-1,J)]) *
                         ((uh[U3(I  ,j  ,k)]+uh[U3(I-1,j+1,k)]) +
                          (uh[U3(I-1,j  ,k)]+uh[U3(I  ,j+1,k)]) ) /
                      (h_tiny + ((Heff1+Heff4) +(Heff2+Heff3)) ) * G->IdyCv[V2(i,J)];
        } else {
          const double UHeff = ((uh[U3(I  ,j  ,k)]+uh[U3(I-1,j+1,k)]) +
                                (uh[U3(I-1,j  ,k)]+uh[U3(I  ,j+1,k)]) );
          const double QUHeff = 0.5*( (abs_vort[Q2(I,J)]+abs_vort[Q2(I-1,J)])*UHeff
                                     -(abs_vort[Q2(I,J)]-abs_vort[Q2(I-1,J)])*fabs(UHeff) );
          ca = - QUHeff /
                       (h_tiny + ((Heff1+Heff4) +(Heff2+Heff3)) ) * G->IdyCv[V2(i,J)];
        }
      } else {
        ca = - ((a[U2(I-1,j)]   * uh[U3(I-1,j,k)] +
                 c[U2(I,j+1)]   * uh[U3(I,j+1,k)])
              + (b[U2(I,j)]     * uh[U3(I,j,k)] +
                 d[U2(I-1,j+1)] * uh[U3(I-1,j+1,k)])) * G->IdyCv[V2(i,J)];
      }
      if (al)                                                    /* :841-845 */
        ca = ca + (ep_v[H2(i,j)]*vh[V3(i,J-1,k)] - ep_v[H2(i,j+1)]*vh[V3(i,J+1,k)]) * G->IdyCv[V2(i,J)];
      if (CS->bound_coriolis) {
        double fu1 = -abs_vort[Q2(I,J)] * u[U3(I,j+1,k)];
        double fu2 = -abs_vort[Q2(I,J)] * u[U3(I,j,k)];
        double fu3 = -abs_vort[Q2(I-1,J)] * u[U3(I-1,j+1,k)];
        double fu4 = -abs_vort[Q2(I-1,J)] * u[U3(I-1,j,k)];
        double max_fu = max4(fu1, fu2, fu3, fu4), min_fu = min4(fu1, fu2, fu3, fu4);
        ca = min2(ca, max_fu);
        ca = max2(ca, min_fu);
      }
      CAv[V3(i,J,k)] = ca - KEy[V2(i,J)];
    }
  }
  free(q); free(Ih_q); free(abs_vort); free(dvdx); free(dudy); free(rel_vort);
  free(hArea_u); free(hArea_v); free(KE); free(KEx); free(KEy); free(a); free(b); free(c); free(d);
  free(ep_u); free(ep_v); free(uh_center); free(vh_center); free(uh_min); free(uh_max); free(vh_min); free(vh_max);
  }
  free(Area_h); free(Area_q);
  return 0;
}
