/*
 * hor_visc.c -- CPU restatement of MOM_hor_visc (TEST INFRASTRUCTURE, see mom6_oracle.h).
 *
 * Reference: src/parameterizations/lateral/MOM_hor_visc.F90
 *   hor_visc_init, the computational part          :2440-2760
 *   horizontal_viscosity                            :245-1979
 * Restated branches: LAPLACIAN (background KH / KH_VEL_SCALE, SMAGORINSKY_KH, ADD_LES_VISCOSITY, BOUND_KH / BETTER_BOUND_KH),
 * BIHARMONIC (background AH / AH_VEL_SCALE / AH_TIME_SCALE, SMAGORINSKY_AH with BOUND_CORIOLIS_BIHARM, BOUND_AH /
 * BETTER_BOUND_AH), NOSLIP, USE_LAND_MASK_FOR_HVISC, USE_CONT_THICKNESS.  Everything else (Leith, Leith+E, MEKE, GME,
 * anisotropy, RE_AH, KH_SIN_LAT, KH_BG_2D, ZB2020, VarMix scaling, OBC, FrictWork) returns an error.
 * PARITY UNPINNED: the reference holds no known-answer vectors for this module; the invariants are in
 * tests/test_hor_visc.py (a rigid rotation / uniform flow feels no stress, momentum is conserved, energy is dissipated).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mom6_oracle.h"

static inline double max2(double a, double b) { return a > b ? a : b; }
static inline double min2(double a, double b) { return a < b ? a : b; }
static inline double min4(double a, double b, double c, double d) { return min2(min2(min2(a, b), c), d); }
static inline double max4(double a, double b, double c, double d) { return max2(max2(max2(a, b), c), d); }

#define H2(i,j) ORC_H2(G,i,j)
#define U2(i,j) ORC_U2(G,i,j)
#define V2(i,j) ORC_V2(G,i,j)
#define Q2(i,j) ORC_Q2(G,i,j)
#define H3(i,j,k) ORC_H3(G,i,j,k)
#define U3(i,j,k) ORC_U3(G,i,j,k)
#define V3(i,j,k) ORC_V3(G,i,j,k)

/* the products hor_visc_init stores in the control structure (:2446-2471, :2587-2594): evaluated where they are used */
#define dx2q(I,J) (G->dxBu[Q2(I,J)]*G->dxBu[Q2(I,J)])
#define dy2q(I,J) (G->dyBu[Q2(I,J)]*G->dyBu[Q2(I,J)])
#define DX_dyBu(I,J) (G->dxBu[Q2(I,J)]*G->IdyBu[Q2(I,J)])
#define DY_dxBu(I,J) (G->dyBu[Q2(I,J)]*G->IdxBu[Q2(I,J)])
#define dx2h(i,j) (G->dxT[H2(i,j)]*G->dxT[H2(i,j)])
#define dy2h(i,j) (G->dyT[H2(i,j)]*G->dyT[H2(i,j)])
#define DX_dyT(i,j) (G->dxT[H2(i,j)]*G->IdyT[H2(i,j)])
#define DY_dxT(i,j) (G->dyT[H2(i,j)]*G->IdxT[H2(i,j)])
#define Idx2dyCu(I,j) ((G->IdxCu[U2(I,j)]*G->IdxCu[U2(I,j)]) * G->IdyCu[U2(I,j)])
#define Idxdy2u(I,j) (G->IdxCu[U2(I,j)] * (G->IdyCu[U2(I,j)]*G->IdyCu[U2(I,j)]))
#define Idx2dyCv(i,J) ((G->IdxCv[V2(i,J)]*G->IdxCv[V2(i,J)]) * G->IdyCv[V2(i,J)])
#define Idxdy2v(i,J) (G->IdxCv[V2(i,J)] * (G->IdyCv[V2(i,J)]*G->IdyCv[V2(i,J)]))

static int unsupported(const mom6hip_hor_visc_cs_t *CS) {
  for (int n = 0; n < 10; n++) if (CS->unsupported[n]) return 1;
  if (CS->no_slip && CS->biharmonic) return 1;      /* "NOSLIP and BIHARMONIC cannot be defined at the same time" :2340 */
  return 0;
}

int orc_hor_visc_init(const mom6hip_grid_t *G, mom6hip_hor_visc_cs_t *CS, double dt) {
  if (unsupported(CS)) return 1;
  const int is = G->isc, ie = G->iec, js = G->jsc, je = G->jec;
  const int Isq = is - 1, Ieq = ie, Jsq = js - 1, Jeq = je;
  const long nH = (long)ORC_NIH(G) * ORC_NJH(G), nQ = (long)(ORC_NIH(G) + 1) * (ORC_NJH(G) + 1);
  double *hx[8] = {CS->Kh_bg_xx, CS->Kh_Max_xx, CS->Ah_bg_xx, CS->Ah_Max_xx, CS->Laplac2_const_xx, CS->Biharm_const_xx,
                   CS->Biharm_const2_xx, CS->reduction_xx};
  double *qx[8] = {CS->Kh_bg_xy, CS->Kh_Max_xy, CS->Ah_bg_xy, CS->Ah_Max_xy, CS->Laplac2_const_xy, CS->Biharm_const_xy,
                   CS->Biharm_const2_xy, CS->reduction_xy};
  for (int n = 0; n < 8; n++) { if (!hx[n] || !qx[n]) return 2; memset(hx[n], 0, sizeof(double) * nH); memset(qx[n], 0, sizeof(double) * nQ); }
  if (!(CS->Laplacian || CS->biharmonic)) { CS->initialized = 1; return 0; }
  const double Idt = 1.0 / dt;
  double Kh_Limit = 0.0, Ah_Limit = 0.0, BoundCorConst = 0.0;

  /* reduction_xx / reduction_xy :2473-2509 */
  for (int j = Jsq; j <= Jeq + 1; j++) for (int i = Isq; i <= Ieq + 1; i++) {
    const int I = i, J = j;
    double r = 1.0;
    if ((G->dy_Cu[U2(I,j)] > 0.0) && (G->dy_Cu[U2(I,j)] < G->dyCu[U2(I,j)]) && (G->dy_Cu[U2(I,j)] < G->dyCu[U2(I,j)] * r))
      r = G->dy_Cu[U2(I,j)] / (G->dyCu[U2(I,j)]);
    if ((G->dy_Cu[U2(I-1,j)] > 0.0) && (G->dy_Cu[U2(I-1,j)] < G->dyCu[U2(I-1,j)]) && (G->dy_Cu[U2(I-1,j)] < G->dyCu[U2(I-1,j)] * r))
      r = G->dy_Cu[U2(I-1,j)] / (G->dyCu[U2(I-1,j)]);
    if ((G->dx_Cv[V2(i,J)] > 0.0) && (G->dx_Cv[V2(i,J)] < G->dxCv[V2(i,J)]) && (G->dx_Cv[V2(i,J)] < G->dxCv[V2(i,J)] * r))
      r = G->dx_Cv[V2(i,J)] / (G->dxCv[V2(i,J)]);
    if ((G->dx_Cv[V2(i,J-1)] > 0.0) && (G->dx_Cv[V2(i,J-1)] < G->dxCv[V2(i,J-1)]) && (G->dx_Cv[V2(i,J-1)] < G->dxCv[V2(i,J-1)] * r))
      r = G->dx_Cv[V2(i,J-1)] / (G->dxCv[V2(i,J-1)]);
    CS->reduction_xx[H2(i,j)] = r;
  }
  for (int J = js - 1; J <= Jeq; J++) for (int I = is - 1; I <= Ieq; I++) {
    const int i = I, j = J;
    double r = 1.0;
    if ((G->dy_Cu[U2(I,j)] > 0.0) && (G->dy_Cu[U2(I,j)] < G->dyCu[U2(I,j)]) && (G->dy_Cu[U2(I,j)] < G->dyCu[U2(I,j)] * r))
      r = G->dy_Cu[U2(I,j)] / (G->dyCu[U2(I,j)]);
    if ((G->dy_Cu[U2(I,j+1)] > 0.0) && (G->dy_Cu[U2(I,j+1)] < G->dyCu[U2(I,j+1)]) && (G->dy_Cu[U2(I,j+1)] < G->dyCu[U2(I,j+1)] * r))
      r = G->dy_Cu[U2(I,j+1)] / (G->dyCu[U2(I,j+1)]);
    if ((G->dx_Cv[V2(i,J)] > 0.0) && (G->dx_Cv[V2(i,J)] < G->dxCv[V2(i,J)]) && (G->dx_Cv[V2(i,J)] < G->dxCv[V2(i,J)] * r))
      r = G->dx_Cv[V2(i,J)] / (G->dxCv[V2(i,J)]);
    if ((G->dx_Cv[V2(i+1,J)] > 0.0) && (G->dx_Cv[V2(i+1,J)] < G->dxCv[V2(i+1,J)]) && (G->dx_Cv[V2(i+1,J)] < G->dxCv[V2(i+1,J)] * r))
      r = G->dx_Cv[V2(i+1,J)] / (G->dxCv[V2(i+1,J)]);
    CS->reduction_xy[Q2(I,J)] = r;
  }

  if (CS->Laplacian) {      /* :2511-2568 */
    if (CS->bound_Kh || CS->bound_Ah) Kh_Limit = 0.3 / (dt * 4.0);
    for (int j = js - 1; j <= Jeq + 1; j++) for (int i = is - 1; i <= Ieq + 1; i++) {
      const double grid_sp_h2 = (2.0 * dx2h(i,j) * dy2h(i,j)) / (dx2h(i,j) + dy2h(i,j));
      if (CS->Smagorinsky_Kh) CS->Laplac2_const_xx[H2(i,j)] = CS->Smag_Lap_const * grid_sp_h2;
      CS->Kh_bg_xx[H2(i,j)] = max2(CS->Kh, CS->Kh_vel_scale * sqrt(grid_sp_h2));
      if (CS->bound_Kh && !CS->better_bound_Kh) {
        CS->Kh_Max_xx[H2(i,j)] = Kh_Limit * grid_sp_h2;
        CS->Kh_bg_xx[H2(i,j)] = min2(CS->Kh_bg_xx[H2(i,j)], CS->Kh_Max_xx[H2(i,j)]);
      }
    }
    for (int J = js - 1; J <= Jeq; J++) for (int I = is - 1; I <= Ieq; I++) {
      const double grid_sp_q2 = (2.0 * dx2q(I,J) * dy2q(I,J)) / (dx2q(I,J) + dy2q(I,J));
      if (CS->Smagorinsky_Kh) CS->Laplac2_const_xy[Q2(I,J)] = CS->Smag_Lap_const * grid_sp_q2;
      CS->Kh_bg_xy[Q2(I,J)] = max2(CS->Kh, CS->Kh_vel_scale * sqrt(grid_sp_q2));
      if (CS->bound_Kh && !CS->better_bound_Kh) {
        CS->Kh_Max_xy[Q2(I,J)] = Kh_Limit * grid_sp_q2;
        CS->Kh_bg_xy[Q2(I,J)] = min2(CS->Kh_bg_xy[Q2(I,J)], CS->Kh_Max_xy[Q2(I,J)]);
      }
    }
  }
  if (CS->biharmonic) {      /* :2570-2660 */
    if (CS->better_bound_Ah || CS->bound_Ah) Ah_Limit = 0.3 / (dt * 64.0);
    if (CS->Smagorinsky_Ah && CS->bound_Coriolis) BoundCorConst = 1.0 / (5.0 * (CS->bound_Cor_vel * CS->bound_Cor_vel));
    for (int j = js - 1; j <= Jeq + 1; j++) for (int i = is - 1; i <= Ieq + 1; i++) {
      const int I = i, J = j;
      const double grid_sp_h2 = (2.0 * dx2h(i,j) * dy2h(i,j)) / (dx2h(i,j) + dy2h(i,j));
      if (CS->Smagorinsky_Ah) {
        CS->Biharm_const_xx[H2(i,j)] = CS->Smag_bi_const * (grid_sp_h2 * grid_sp_h2);
        if (CS->bound_Coriolis) {
          const double fmax = max4(fabs(G->CoriolisBu[Q2(I-1,J-1)]), fabs(G->CoriolisBu[Q2(I,J-1)]),
                                   fabs(G->CoriolisBu[Q2(I-1,J)]), fabs(G->CoriolisBu[Q2(I,J)]));
          CS->Biharm_const2_xx[H2(i,j)] = (grid_sp_h2 * grid_sp_h2 * grid_sp_h2) * (fmax * BoundCorConst);
        }
      }
      CS->Ah_bg_xx[H2(i,j)] = max2(CS->Ah, CS->Ah_vel_scale * grid_sp_h2 * sqrt(grid_sp_h2));
      if (CS->Ah_time_scale > 0.) CS->Ah_bg_xx[H2(i,j)] = max2(CS->Ah_bg_xx[H2(i,j)], (grid_sp_h2 * grid_sp_h2) / CS->Ah_time_scale);
      if (CS->bound_Ah && !CS->better_bound_Ah) {
        CS->Ah_Max_xx[H2(i,j)] = Ah_Limit * (grid_sp_h2 * grid_sp_h2);
        CS->Ah_bg_xx[H2(i,j)] = min2(CS->Ah_bg_xx[H2(i,j)], CS->Ah_Max_xx[H2(i,j)]);
      }
    }
    for (int J = js - 1; J <= Jeq; J++) for (int I = is - 1; I <= Ieq; I++) {
      const double grid_sp_q2 = (2.0 * dx2q(I,J) * dy2q(I,J)) / (dx2q(I,J) + dy2q(I,J));
      if (CS->Smagorinsky_Ah) {
        CS->Biharm_const_xy[Q2(I,J)] = CS->Smag_bi_const * (grid_sp_q2 * grid_sp_q2);
        if (CS->bound_Coriolis)
          CS->Biharm_const2_xy[Q2(I,J)] = (grid_sp_q2 * grid_sp_q2 * grid_sp_q2) * (fabs(G->CoriolisBu[Q2(I,J)]) * BoundCorConst);
      }
      CS->Ah_bg_xy[Q2(I,J)] = max2(CS->Ah, CS->Ah_vel_scale * grid_sp_q2 * sqrt(grid_sp_q2));
      if (CS->Ah_time_scale > 0.) CS->Ah_bg_xy[Q2(I,J)] = max2(CS->Ah_bg_xy[Q2(I,J)], (grid_sp_q2 * grid_sp_q2) / CS->Ah_time_scale);
      if (CS->bound_Ah && !CS->better_bound_Ah) {
        CS->Ah_Max_xy[Q2(I,J)] = Ah_Limit * (grid_sp_q2 * grid_sp_q2);
        CS->Ah_bg_xy[Q2(I,J)] = min2(CS->Ah_bg_xy[Q2(I,J)], CS->Ah_Max_xy[Q2(I,J)]);
      }
    }
  }
  /* the stability bounds of the better_bound forms, :2664-2693 */
  if (CS->Laplacian && CS->better_bound_Kh) {
    for (int j = js - 1; j <= Jeq + 1; j++) for (int i = is - 1; i <= Ieq + 1; i++) {
      const int I = i, J = j;
      const double denom = max2(
          (dy2h(i,j) * DY_dxT(i,j) * (G->IdyCu[U2(I,j)] + G->IdyCu[U2(I-1,j)]) *
           max2(G->IdyCu[U2(I,j)] * G->IareaCu[U2(I,j)], G->IdyCu[U2(I-1,j)] * G->IareaCu[U2(I-1,j)])),
          (dx2h(i,j) * DX_dyT(i,j) * (G->IdxCv[V2(i,J)] + G->IdxCv[V2(i,J-1)]) *
           max2(G->IdxCv[V2(i,J)] * G->IareaCv[V2(i,J)], G->IdxCv[V2(i,J-1)] * G->IareaCv[V2(i,J-1)])));
      CS->Kh_Max_xx[H2(i,j)] = 0.0;
      if (denom > 0.0) CS->Kh_Max_xx[H2(i,j)] = CS->bound_coef * 0.25 * Idt / denom;
    }
    for (int J = js - 1; J <= Jeq; J++) for (int I = is - 1; I <= Ieq; I++) {
      const int i = I, j = J;
      const double denom = max2(
          (dx2q(I,J) * DX_dyBu(I,J) * (G->IdxCu[U2(I,j+1)] + G->IdxCu[U2(I,j)]) *
           max2(G->IdxCu[U2(I,j)] * G->IareaCu[U2(I,j)], G->IdxCu[U2(I,j+1)] * G->IareaCu[U2(I,j+1)])),
          (dy2q(I,J) * DY_dxBu(I,J) * (G->IdyCv[V2(i+1,J)] + G->IdyCv[V2(i,J)]) *
           max2(G->IdyCv[V2(i,J)] * G->IareaCv[V2(i,J)], G->IdyCv[V2(i+1,J)] * G->IareaCv[V2(i+1,J)])));
      CS->Kh_Max_xy[Q2(I,J)] = 0.0;
      if (denom > 0.0) CS->Kh_Max_xy[Q2(I,J)] = CS->bound_coef * 0.25 * Idt / denom;
    }
  }
  if (CS->biharmonic && CS->better_bound_Ah) {      /* :2695-2755 */
    const long nU = (long)(ORC_NIH(G) + 1) * ORC_NJH(G), nV = (long)ORC_NIH(G) * (ORC_NJH(G) + 1);
    double *u0u = calloc(nU, 8), *u0v = calloc(nU, 8), *v0u = calloc(nV, 8), *v0v = calloc(nV, 8);
    for (int j = js - 1; j <= Jeq + 1; j++) for (int I = is - 2; I <= Ieq + 1; I++) {
      const int i = I, J = j;
      u0u[U2(I,j)] = (Idxdy2u(I,j) * (dy2h(i+1,j) * DY_dxT(i+1,j) * (G->IdyCu[U2(I+1,j)] + G->IdyCu[U2(I,j)]) +
                                      dy2h(i,j) * DY_dxT(i,j) * (G->IdyCu[U2(I,j)] + G->IdyCu[U2(I-1,j)])) +
                      Idx2dyCu(I,j) * (dx2q(I,J) * DX_dyBu(I,J) * (G->IdxCu[U2(I,j+1)] + G->IdxCu[U2(I,j)]) +
                                       dx2q(I,J-1) * DX_dyBu(I,J-1) * (G->IdxCu[U2(I,j)] + G->IdxCu[U2(I,j-1)])));
      u0v[U2(I,j)] = (Idxdy2u(I,j) * (dy2h(i+1,j) * DX_dyT(i+1,j) * (G->IdxCv[V2(i+1,J)] + G->IdxCv[V2(i+1,J-1)]) +
                                      dy2h(i,j) * DX_dyT(i,j) * (G->IdxCv[V2(i,J)] + G->IdxCv[V2(i,J-1)])) +
                      Idx2dyCu(I,j) * (dx2q(I,J) * DY_dxBu(I,J) * (G->IdyCv[V2(i+1,J)] + G->IdyCv[V2(i,J)]) +
                                       dx2q(I,J-1) * DY_dxBu(I,J-1) * (G->IdyCv[V2(i+1,J-1)] + G->IdyCv[V2(i,J-1)])));
    }
    for (int J = js - 2; J <= Jeq + 1; J++) for (int i = is - 1; i <= Ieq + 1; i++) {
      const int I = i, j = J;
      v0u[V2(i,J)] = (Idxdy2v(i,J) * (dy2q(I,J) * DX_dyBu(I,J) * (G->IdxCu[U2(I,j+1)] + G->IdxCu[U2(I,j)]) +
                                      dy2q(I-1,J) * DX_dyBu(I-1,J) * (G->IdxCu[U2(I-1,j+1)] + G->IdxCu[U2(I-1,j)])) +
                      Idx2dyCv(i,J) * (dx2h(i,j+1) * DY_dxT(i,j+1) * (G->IdyCu[U2(I,j+1)] + G->IdyCu[U2(I-1,j+1)]) +
                                       dx2h(i,j) * DY_dxT(i,j) * (G->IdyCu[U2(I,j)] + G->IdyCu[U2(I-1,j)])));
      v0v[V2(i,J)] = (Idxdy2v(i,J) * (dy2q(I,J) * DY_dxBu(I,J) * (G->IdyCv[V2(i+1,J)] + G->IdyCv[V2(i,J)]) +
                                      dy2q(I-1,J) * DY_dxBu(I-1,J) * (G->IdyCv[V2(i,J)] + G->IdyCv[V2(i-1,J)])) +
                      Idx2dyCv(i,J) * (dx2h(i,j+1) * DX_dyT(i,j+1) * (G->IdxCv[V2(i,J+1)] + G->IdxCv[V2(i,J)]) +
                                       dx2h(i,j) * DX_dyT(i,j) * (G->IdxCv[V2(i,J)] + G->IdxCv[V2(i,J-1)])));
    }
    for (int j = js - 1; j <= Jeq + 1; j++) for (int i = is - 1; i <= Ieq + 1; i++) {
      const int I = i, J = j;
      const double denom = max2(
          (dy2h(i,j) *
           (DY_dxT(i,j) * (G->IdyCu[U2(I,j)] * u0u[U2(I,j)] + G->IdyCu[U2(I-1,j)] * u0u[U2(I-1,j)]) +
            DX_dyT(i,j) * (G->IdxCv[V2(i,J)] * v0u[V2(i,J)] + G->IdxCv[V2(i,J-1)] * v0u[V2(i,J-1)])) *
           max2(G->IdyCu[U2(I,j)] * G->IareaCu[U2(I,j)], G->IdyCu[U2(I-1,j)] * G->IareaCu[U2(I-1,j)])),
          (dx2h(i,j) *
           (DY_dxT(i,j) * (G->IdyCu[U2(I,j)] * u0v[U2(I,j)] + G->IdyCu[U2(I-1,j)] * u0v[U2(I-1,j)]) +
            DX_dyT(i,j) * (G->IdxCv[V2(i,J)] * v0v[V2(i,J)] + G->IdxCv[V2(i,J-1)] * v0v[V2(i,J-1)])) *
           max2(G->IdxCv[V2(i,J)] * G->IareaCv[V2(i,J)], G->IdxCv[V2(i,J-1)] * G->IareaCv[V2(i,J-1)])));
      CS->Ah_Max_xx[H2(i,j)] = 0.0;
      if (denom > 0.0) CS->Ah_Max_xx[H2(i,j)] = CS->bound_coef * 0.5 * Idt / denom;
    }
    for (int J = js - 1; J <= Jeq; J++) for (int I = is - 1; I <= Ieq; I++) {
      const int i = I, j = J;
      const double denom = max2(
          (dx2q(I,J) *
           (DX_dyBu(I,J) * (u0u[U2(I,j+1)] * G->IdxCu[U2(I,j+1)] + u0u[U2(I,j)] * G->IdxCu[U2(I,j)]) +
            DY_dxBu(I,J) * (v0u[V2(i+1,J)] * G->IdyCv[V2(i+1,J)] + v0u[V2(i,J)] * G->IdyCv[V2(i,J)])) *
           max2(G->IdxCu[U2(I,j)] * G->IareaCu[U2(I,j)], G->IdxCu[U2(I,j+1)] * G->IareaCu[U2(I,j+1)])),
          (dy2q(I,J) *
           (DX_dyBu(I,J) * (u0v[U2(I,j+1)] * G->IdxCu[U2(I,j+1)] + u0v[U2(I,j)] * G->IdxCu[U2(I,j)]) +
            DY_dxBu(I,J) * (v0v[V2(i+1,J)] * G->IdyCv[V2(i+1,J)] + v0v[V2(i,J)] * G->IdyCv[V2(i,J)])) *
           max2(G->IdyCv[V2(i,J)] * G->IareaCv[V2(i,J)], G->IdyCv[V2(i+1,J)] * G->IareaCv[V2(i+1,J)])));
      CS->Ah_Max_xy[Q2(I,J)] = 0.0;
      if (denom > 0.0) CS->Ah_Max_xy[Q2(I,J)] = CS->bound_coef * 0.5 * Idt / denom;
    }
    free(u0u); free(u0v); free(v0u); free(v0v);
  }
  CS->initialized = 1;
  return 0;
}

int orc_horizontal_viscosity(const mom6hip_grid_t *G, const mom6hip_hor_visc_cs_t *CS, const double *u, const double *v,
                             const double *h, double *diffu, double *diffv, double dt, const double *hu_cont,
                             const double *hv_cont) {
  return orc_horizontal_viscosity_obc(G, CS, u, v, h, diffu, diffv, dt, hu_cont, hv_cont, NULL);
}

#define MAX2I(a, b) ((a) > (b) ? (a) : (b))
#define MIN2I(a, b) ((a) < (b) ? (a) : (b))
/* a segment's tangential_vel on its own index ranges (IsdB:IedB, JsdB:JedB, nk), k 1-based */
static inline double seg_tang(const mom6hip_obc_segment_t *S, int I, int J, int k) {
  const long nI = S->IedB - S->IsdB + 1, nJ = S->JedB - S->JsdB + 1;
  return S->tangential_vel[(I - S->IsdB) + nI * ((J - S->JsdB) + nJ * (long)(k - 1))];
}

/* horizontal_viscosity with OBC associated (and OBC%OBC_pe): the strains at the corner points of the segments :733-790, the thicknesses at
 * and beside their faces :791-849, OBC_ZERO_BIHARMONIC :889-903, the gradient of the Laplacian :1388-1409, the accelerations of the
 * segments' own faces :1751-1780.  (A segment that is not on the PE carries no index ranges here: it is skipped.) */
int orc_horizontal_viscosity_obc(const mom6hip_grid_t *G, const mom6hip_hor_visc_cs_t *CS, const double *u, const double *v,
                                 const double *h, double *diffu, double *diffv, double dt, const double *hu_cont,
                                 const double *hv_cont, const mom6hip_obc_t *OBC) {
  (void)dt;
  const int apply_OBC = OBC && OBC->OBC_pe;      /* :449-452 */
  if (apply_OBC && OBC->number_of_segments > 0 && !OBC->segment) return 2;
  if (apply_OBC && OBC->computed_strain)
    for (int n = 0; n < OBC->number_of_segments; n++) if (OBC->segment[n].on_pe && !OBC->segment[n].tangential_vel) return 2;
  if (!CS->initialized) return 3;      /* "MOM_hor_visc: Module must be initialized before it is used." */
  if (unsupported(CS)) return 1;
  if (!(CS->Laplacian || CS->biharmonic)) return 0;      /* :451 */
  const int is = G->isc, ie = G->iec, js = G->jsc, je = G->jec, nz = G->nk;
  const int Isq = is - 1, Ieq = ie, Jsq = js - 1, Jeq = je;
  if (is - G->isd < 2 || js - G->jsd < 2) return 4;
  const double h_neglect = G->H_subroundoff;
  const double h_neglect3 = h_neglect * h_neglect * h_neglect;
  const int use_cont_huv = CS->use_cont_thick && hu_cont && hv_cont;
  const int is_Kh = Isq, ie_Kh = ie + 1, js_Kh = Jsq, je_Kh = je + 1;
  const int is_vort = is - 2, ie_vort = Ieq + 1, js_vort = js - 2, je_vort = Jeq + 1;
  const int legacy_bound = CS->Smagorinsky_Kh && (CS->bound_Kh && !CS->better_bound_Kh);
  const long nH = (long)ORC_NIH(G) * ORC_NJH(G), nU = (long)(ORC_NIH(G) + 1) * ORC_NJH(G);
  const long nV = (long)ORC_NIH(G) * (ORC_NJH(G) + 1), nQ = (long)(ORC_NIH(G) + 1) * (ORC_NJH(G) + 1);

  /* MEKE%mom_src: the frictional work of each layer, then its sum over the layers in order (:1833-1889 with backscatter_Ro_c = 0) */
  double *FrictWork = CS->MEKE_mom_src ? (double *)calloc((size_t)nH * nz, 8) : NULL;
  /* the layers are independent (the reference: !$OMP parallel do over k, :634-680); the 2-D work arrays are per thread */
  _Pragma("omp parallel")
  {
  double *dudx = calloc(nH, 8), *dvdy = calloc(nH, 8), *sh_xx = calloc(nH, 8), *str_xx = calloc(nH, 8);
  double *dvdx = calloc(nQ, 8), *dudy = calloc(nQ, 8), *sh_xy = calloc(nQ, 8), *str_xy = calloc(nQ, 8), *hq = calloc(nQ, 8);
  double *dDel2vdx = calloc(nQ, 8), *dDel2udy = calloc(nQ, 8);
  double *h_u = calloc(nU, 8), *Del2u = calloc(nU, 8), *h_v = calloc(nV, 8), *Del2v = calloc(nV, 8);
  /* Ah, Kh, Shear_mag, hrat_min, visc_bound_rem have q-point extents and are used at h and at q points */
  double *Ah = calloc(nQ, 8), *Kh = calloc(nQ, 8), *Shear_mag = calloc(nQ, 8), *hrat_min = calloc(nQ, 8), *visc_bound_rem = calloc(nQ, 8);
#define HQ(a,i,j) a[Q2((i)-1,(j)-1)]      /* an h-point value kept in a q-sized work array (the reference's own re-use) */
  _Pragma("omp for schedule(static)")
  for (int k = 1; k <= nz; k++) {
    /* horizontal tension :693-699 */
    for (int j = Jsq - 1; j <= Jeq + 2; j++) for (int i = Isq - 1; i <= Ieq + 2; i++) {
      const int I = i, J = j;
      dudx[H2(i,j)] = DY_dxT(i,j) * (G->IdyCu[U2(I,j)] * u[U3(I,j,k)] - G->IdyCu[U2(I-1,j)] * u[U3(I-1,j,k)]);
      dvdy[H2(i,j)] = DX_dyT(i,j) * (G->IdxCv[V2(i,J)] * v[V3(i,J,k)] - G->IdxCv[V2(i,J-1)] * v[V3(i,J-1,k)]);
      sh_xx[H2(i,j)] = dudx[H2(i,j)] - dvdy[H2(i,j)];
    }
    /* components of the shearing strain :702-705 */
    for (int J = js_vort; J <= je_vort; J++) for (int I = is_vort; I <= ie_vort; I++) {
      const int i = I, j = J;
      dvdx[Q2(I,J)] = DY_dxBu(I,J) * (v[V3(i+1,J,k)] * G->IdyCv[V2(i+1,J)] - v[V3(i,J,k)] * G->IdyCv[V2(i,J)]);
      dudy[Q2(I,J)] = DX_dyBu(I,J) * (u[U3(I,j+1,k)] * G->IdxCu[U2(I,j+1)] - u[U3(I,j,k)] * G->IdxCu[U2(I,j)]);
    }
    /* thicknesses at velocity points :740-765 */
    if (CS->use_land_mask) {
      for (int j = js - 2; j <= je + 2; j++) for (int I = is - 2; I <= Ieq + 1; I++) {
        const int i = I;
        h_u[U2(I,j)] = 0.5 * (G->mask2dT[H2(i,j)] * h[H3(i,j,k)] + G->mask2dT[H2(i+1,j)] * h[H3(i+1,j,k)]);
      }
      for (int J = js - 2; J <= Jeq + 1; J++) for (int i = is - 2; i <= ie + 2; i++) {
        const int j = J;
        h_v[V2(i,J)] = 0.5 * (G->mask2dT[H2(i,j)] * h[H3(i,j,k)] + G->mask2dT[H2(i,j+1)] * h[H3(i,j+1,k)]);
      }
    } else {
      for (int j = js - 2; j <= je + 2; j++) for (int I = is - 2; I <= Ieq + 1; I++) {
        const int i = I;
        h_u[U2(I,j)] = 0.5 * (h[H3(i,j,k)] + h[H3(i+1,j,k)]);
      }
      for (int J = js - 2; J <= Jeq + 1; J++) for (int i = is - 2; i <= ie + 2; i++) {
        const int j = J;
        h_v[V2(i,J)] = 0.5 * (h[H3(i,j,k)] + h[H3(i,j+1,k)]);
      }
    }
    if (use_cont_huv) {
      for (int j = js - 2; j <= je + 2; j++) for (int I = Isq - 1; I <= Ieq + 1; I++) h_u[U2(I,j)] = hu_cont[U3(I,j,k)];
      for (int J = Jsq - 1; J <= Jeq + 1; J++) for (int i = is - 2; i <= ie + 2; i++) h_v[V2(i,J)] = hv_cont[V3(i,J,k)];
    }
    if (apply_OBC) for (int n = 0; n < OBC->number_of_segments; n++) {      /* :733-819 */
      const mom6hip_obc_segment_t *S = &OBC->segment[n];
      if (!S->on_pe) continue;
      const int J = S->JsdB, I = S->IsdB;
      if (OBC->zero_strain || OBC->freeslip_strain || OBC->computed_strain) {
        if (S->is_N_or_S && (J >= js_vort) && (J <= je_vort)) {
          const int j = J;
          for (int Iq = MAX2I(S->IsdB, is_vort); Iq <= MIN2I(S->IedB, ie_vort); Iq++) {
            if (OBC->zero_strain) { dvdx[Q2(Iq,J)] = 0.; dudy[Q2(Iq,J)] = 0.; }
            else if (OBC->freeslip_strain) dudy[Q2(Iq,J)] = 0.;
            else if (OBC->computed_strain) {
              if (S->direction == MOM6HIP_OBC_DIRECTION_N)
                dudy[Q2(Iq,J)] = 2.0 * DX_dyBu(Iq,J) * (seg_tang(S, Iq, J, k) - u[U3(Iq,j,k)]) * G->IdxCu[U2(Iq,j)];
              else
                dudy[Q2(Iq,J)] = 2.0 * DX_dyBu(Iq,J) * (u[U3(Iq,j+1,k)] - seg_tang(S, Iq, J, k)) * G->IdxCu[U2(Iq,j+1)];
            }
          }
        } else if (S->is_E_or_W && (I >= is_vort) && (I <= ie_vort)) {
          const int i = I;
          for (int Jq = MAX2I(S->JsdB, js_vort); Jq <= MIN2I(S->JedB, je_vort); Jq++) {
            if (OBC->zero_strain) { dvdx[Q2(I,Jq)] = 0.; dudy[Q2(I,Jq)] = 0.; }
            else if (OBC->freeslip_strain) dvdx[Q2(I,Jq)] = 0.;
            else if (OBC->computed_strain) {
              if (S->direction == MOM6HIP_OBC_DIRECTION_E)
                dvdx[Q2(I,Jq)] = 2.0 * DY_dxBu(I,Jq) * (seg_tang(S, I, Jq, k) - v[V3(i,Jq,k)]) * G->IdyCv[V2(i,Jq)];
              else
                dvdx[Q2(I,Jq)] = 2.0 * DY_dxBu(I,Jq) * (v[V3(i+1,Jq,k)] - seg_tang(S, I, Jq, k)) * G->IdyCv[V2(i+1,Jq)];
            }
          }
        }
      }
      if (S->direction == MOM6HIP_OBC_DIRECTION_N) {      /* :791-819: the thickness of the cell inside at the segment's faces */
        if ((J >= js - 2) && (J <= Jeq + 1))
          for (int i = MAX2I(is - 2, S->isd); i <= MIN2I(ie + 2, S->ied); i++) h_v[V2(i,J)] = h[H3(i,J,k)];
      } else if (S->direction == MOM6HIP_OBC_DIRECTION_S) {
        if ((J >= js - 2) && (J <= Jeq + 1))
          for (int i = MAX2I(is - 2, S->isd); i <= MIN2I(ie + 2, S->ied); i++) h_v[V2(i,J)] = h[H3(i,J+1,k)];
      } else if (S->direction == MOM6HIP_OBC_DIRECTION_E) {
        if ((I >= is - 2) && (I <= Ieq + 1))
          for (int j = MAX2I(js - 2, S->jsd); j <= MIN2I(je + 2, S->jed); j++) h_u[U2(I,j)] = h[H3(I,j,k)];
      } else if (S->direction == MOM6HIP_OBC_DIRECTION_W) {
        if ((I >= is - 2) && (I <= Ieq + 1))
          for (int j = MAX2I(js - 2, S->jsd); j <= MIN2I(je + 2, S->jed); j++) h_u[U2(I,j)] = h[H3(I+1,j,k)];
      }
    }
    if (apply_OBC) for (int n = 0; n < OBC->number_of_segments; n++) {      /* :821-849: then across the corner points of the segments */
      const mom6hip_obc_segment_t *S = &OBC->segment[n];
      if (!S->on_pe) continue;
      const int J = S->JsdB, I = S->IsdB;
      if (S->direction == MOM6HIP_OBC_DIRECTION_N) {
        if ((J >= js - 2) && (J <= je))
          for (int Iq = MAX2I(is - 2, S->IsdB); Iq <= MIN2I(Ieq + 1, S->IedB); Iq++) h_u[U2(Iq,J+1)] = h_u[U2(Iq,J)];
      } else if (S->direction == MOM6HIP_OBC_DIRECTION_S) {
        if ((J >= js - 1) && (J <= je + 1))
          for (int Iq = MAX2I(is - 2, S->isd); Iq <= MIN2I(Ieq + 1, S->ied); Iq++) h_u[U2(Iq,J)] = h_u[U2(Iq,J+1)];
      } else if (S->direction == MOM6HIP_OBC_DIRECTION_E) {
        if ((I >= is - 2) && (I <= ie))
          for (int Jq = MAX2I(js - 2, S->jsd); Jq <= MIN2I(Jeq + 1, S->jed); Jq++) h_v[V2(I+1,Jq)] = h_v[V2(I,Jq)];
      } else if (S->direction == MOM6HIP_OBC_DIRECTION_W) {
        if ((I >= is - 1) && (I <= ie + 1))
          for (int Jq = MAX2I(js - 2, S->jsd); Jq <= MIN2I(Jeq + 1, S->jed); Jq++) h_v[V2(I,Jq)] = h_v[V2(I+1,Jq)];
      }
    }
    /* shearing strain :852-864 */
    for (int J = js - 2; J <= Jeq + 1; J++) for (int I = is - 2; I <= Ieq + 1; I++) {
      if (CS->no_slip) sh_xy[Q2(I,J)] = (2.0 - G->mask2dBu[Q2(I,J)]) * (dvdx[Q2(I,J)] + dudy[Q2(I,J)]);
      else sh_xy[Q2(I,J)] = G->mask2dBu[Q2(I,J)] * (dvdx[Q2(I,J)] + dudy[Q2(I,J)]);
    }
    /* Del2u, Del2v :882-891 */
    if (CS->biharmonic) {
      for (int j = js - 1; j <= Jeq + 1; j++) for (int I = Isq - 1; I <= Ieq + 1; I++) {
        const int i = I, J = j;
        Del2u[U2(I,j)] = Idxdy2u(I,j) * (dy2h(i+1,j) * sh_xx[H2(i+1,j)] - dy2h(i,j) * sh_xx[H2(i,j)]) +
                         Idx2dyCu(I,j) * (dx2q(I,J) * sh_xy[Q2(I,J)] - dx2q(I,J-1) * sh_xy[Q2(I,J-1)]);
      }
      for (int J = Jsq - 1; J <= Jeq + 1; J++) for (int i = is - 1; i <= Ieq + 1; i++) {
        const int I = i, j = J;
        Del2v[V2(i,J)] = Idxdy2v(i,J) * (dy2q(I,J) * sh_xy[Q2(I,J)] - dy2q(I-1,J) * sh_xy[Q2(I-1,J)]) -
                         Idx2dyCv(i,J) * (dx2h(i,j+1) * sh_xx[H2(i,j+1)] - dx2h(i,j) * sh_xx[H2(i,j)]);
      }
      if (apply_OBC && OBC->zero_biharmonic) for (int n = 0; n < OBC->number_of_segments; n++) {      /* :889-903 */
        const mom6hip_obc_segment_t *S = &OBC->segment[n];
        if (!S->on_pe) continue;
        const int I = S->IsdB, J = S->JsdB;
        if (S->is_N_or_S && (J >= Jsq - 1) && (J <= Jeq + 1)) {
          for (int i = S->isd; i <= S->ied; i++) Del2v[V2(i,J)] = 0.;
        } else if (S->is_E_or_W && (I >= Isq - 1) && (I <= Ieq + 1)) {
          for (int j = S->jsd; j <= S->jed; j++) Del2u[U2(I,j)] = 0.;
        }
      }
    }
    /* Smagorinsky shear magnitude at h points :1056-1063 */
    if (CS->Smagorinsky_Kh || CS->Smagorinsky_Ah) {
      for (int j = js_Kh; j <= je_Kh; j++) for (int i = is_Kh; i <= ie_Kh; i++) {
        const int I = i, J = j;
        const double sh_xx_sq = sh_xx[H2(i,j)] * sh_xx[H2(i,j)];
        const double sh_xy_sq = 0.25 * ((sh_xy[Q2(I-1,J-1)] * sh_xy[Q2(I-1,J-1)] + sh_xy[Q2(I,J)] * sh_xy[Q2(I,J)]) +
                                        (sh_xy[Q2(I-1,J)] * sh_xy[Q2(I-1,J)] + sh_xy[Q2(I,J-1)] * sh_xy[Q2(I,J-1)]));
        HQ(Shear_mag,i,j) = sqrt(sh_xx_sq + sh_xy_sq);
      }
    }
    if (CS->better_bound_Ah || CS->better_bound_Kh) {      /* :1065-1076 */
      for (int j = js_Kh; j <= je_Kh; j++) for (int i = is_Kh; i <= ie_Kh; i++) {
        const int I = i, J = j;
        const double h_min = min4(h_u[U2(I,j)], h_u[U2(I-1,j)], h_v[V2(i,J)], h_v[V2(i,J-1)]);
        HQ(hrat_min,i,j) = min2(1.0, h_min / (h[H3(i,j,k)] + h_neglect));
        if (CS->better_bound_Kh) HQ(visc_bound_rem,i,j) = 1.0;
      }
    }
    if (CS->Laplacian) {      /* Kh at h points and the Laplacian part of str_xx :1078-1214 */
      for (int j = js_Kh; j <= je_Kh; j++) for (int i = is_Kh; i <= ie_Kh; i++) {
        double K_ = CS->Kh_bg_xx[H2(i,j)];
        if (CS->add_LES_viscosity) {
          if (CS->Smagorinsky_Kh) K_ = K_ + CS->Laplac2_const_xx[H2(i,j)] * HQ(Shear_mag,i,j);
        } else {
          if (CS->Smagorinsky_Kh) K_ = max2(K_, CS->Laplac2_const_xx[H2(i,j)] * HQ(Shear_mag,i,j));
        }
        if (legacy_bound) K_ = min2(K_, CS->Kh_Max_xx[H2(i,j)]);
        K_ = max2(K_, CS->Kh_bg_min);
        if (CS->MEKE_Ku) K_ = K_ + CS->MEKE_Ku[H2(i,j)];      /* :1141-1151 */
        if (CS->better_bound_Kh) {
          if (K_ >= HQ(hrat_min,i,j) * CS->Kh_Max_xx[H2(i,j)]) {
            HQ(visc_bound_rem,i,j) = 0.0;
            K_ = HQ(hrat_min,i,j) * CS->Kh_Max_xx[H2(i,j)];
          } else {
            HQ(visc_bound_rem,i,j) = 1.0 - K_ / (HQ(hrat_min,i,j) * CS->Kh_Max_xx[H2(i,j)]);
          }
        }
        HQ(Kh,i,j) = K_;
      }
      for (int j = Jsq; j <= Jeq + 1; j++) for (int i = Isq; i <= Ieq + 1; i++) str_xx[H2(i,j)] = -HQ(Kh,i,j) * sh_xx[H2(i,j)];
    } else {
      for (int j = Jsq; j <= Jeq + 1; j++) for (int i = Isq; i <= Ieq + 1; i++) str_xx[H2(i,j)] = 0.0;
    }
    if (CS->biharmonic) {      /* Ah at h points and the biharmonic part of str_xx :1227-1380 */
      for (int j = js_Kh; j <= je_Kh; j++) for (int i = is_Kh; i <= ie_Kh; i++) {
        double A_ = CS->Ah_bg_xx[H2(i,j)];
        if (CS->Smagorinsky_Ah) {
          double AhSm;
          if (CS->bound_Coriolis)
            AhSm = HQ(Shear_mag,i,j) * (CS->Biharm_const_xx[H2(i,j)] + CS->Biharm_const2_xx[H2(i,j)] * HQ(Shear_mag,i,j));
          else
            AhSm = CS->Biharm_const_xx[H2(i,j)] * HQ(Shear_mag,i,j);
          A_ = max2(A_, AhSm);
          if (CS->bound_Ah && !CS->better_bound_Ah) A_ = min2(A_, CS->Ah_Max_xx[H2(i,j)]);
        }
        if (CS->MEKE_Au) A_ = A_ + CS->MEKE_Au[H2(i,j)];      /* :1318-1323 */
        if (CS->better_bound_Ah) {
          if (CS->better_bound_Kh) A_ = min2(A_, HQ(visc_bound_rem,i,j) * HQ(hrat_min,i,j) * CS->Ah_Max_xx[H2(i,j)]);
          else A_ = min2(A_, HQ(hrat_min,i,j) * CS->Ah_Max_xx[H2(i,j)]);
        }
        HQ(Ah,i,j) = A_;
      }
      for (int j = Jsq; j <= Jeq + 1; j++) for (int i = Isq; i <= Ieq + 1; i++) {
        const int I = i, J = j;
        const double d_del2u = G->IdyCu[U2(I,j)] * Del2u[U2(I,j)] - G->IdyCu[U2(I-1,j)] * Del2u[U2(I-1,j)];
        const double d_del2v = G->IdxCv[V2(i,J)] * Del2v[V2(i,J)] - G->IdxCv[V2(i,J-1)] * Del2v[V2(i,J-1)];
        const double d_str = HQ(Ah,i,j) * (DY_dxT(i,j) * d_del2u - DX_dyT(i,j) * d_del2v);
        str_xx[H2(i,j)] = str_xx[H2(i,j)] + d_str;
      }
      /* gradient of the Laplacian :1382-1387 */
      for (int J = js - 1; J <= Jeq; J++) for (int I = is - 1; I <= Ieq; I++) {
        const int i = I, j = J;
        dDel2vdx[Q2(I,J)] = DY_dxBu(I,J) * (Del2v[V2(i+1,J)] * G->IdyCv[V2(i+1,J)] - Del2v[V2(i,J)] * G->IdyCv[V2(i,J)]);
        dDel2udy[Q2(I,J)] = DX_dyBu(I,J) * (Del2u[U2(I,j+1)] * G->IdxCu[U2(I,j+1)] - Del2u[U2(I,j)] * G->IdxCu[U2(I,j)]);
      }
      if (apply_OBC && (OBC->zero_strain || OBC->freeslip_strain)) for (int n = 0; n < OBC->number_of_segments; n++) {      /* :1388-1409 */
        const mom6hip_obc_segment_t *S = &OBC->segment[n];
        if (!S->on_pe) continue;
        const int J = S->JsdB, I = S->IsdB;
        if (S->is_N_or_S && (J >= js - 1) && (J <= Jeq)) {
          for (int Iq = S->IsdB; Iq <= S->IedB; Iq++) {
            if (OBC->zero_strain) { dDel2vdx[Q2(Iq,J)] = 0.; dDel2udy[Q2(Iq,J)] = 0.; }
            else if (OBC->freeslip_strain) dDel2udy[Q2(Iq,J)] = 0.;
          }
        } else if (S->is_E_or_W && (I >= is - 1) && (I <= Ieq)) {
          for (int Jq = S->JsdB; Jq <= S->JedB; Jq++) {
            if (OBC->zero_strain) { dDel2vdx[Q2(I,Jq)] = 0.; dDel2udy[Q2(I,Jq)] = 0.; }
            else if (OBC->freeslip_strain) dDel2vdx[Q2(I,Jq)] = 0.;
          }
        }
      }
    }
    /* ---- q points ---- */
    if (CS->Smagorinsky_Kh || CS->Smagorinsky_Ah) {      /* :1414-1421 */
      for (int J = js - 1; J <= Jeq; J++) for (int I = is - 1; I <= Ieq; I++) {
        const int i = I, j = J;
        const double sh_xy_sq = sh_xy[Q2(I,J)] * sh_xy[Q2(I,J)];
        const double sh_xx_sq = 0.25 * ((sh_xx[H2(i,j)] * sh_xx[H2(i,j)] + sh_xx[H2(i+1,j+1)] * sh_xx[H2(i+1,j+1)]) +
                                        (sh_xx[H2(i,j+1)] * sh_xx[H2(i,j+1)] + sh_xx[H2(i+1,j)] * sh_xx[H2(i+1,j)]));
        Shear_mag[Q2(I,J)] = sqrt(sh_xy_sq + sh_xx_sq);
      }
    }
    for (int J = js - 1; J <= Jeq; J++) for (int I = is - 1; I <= Ieq; I++) {      /* :1423-1428 */
      const int i = I, j = J;
      const double h2uq = 4.0 * (h_u[U2(I,j)] * h_u[U2(I,j+1)]);
      const double h2vq = 4.0 * (h_v[V2(i,J)] * h_v[V2(i+1,J)]);
      hq[Q2(I,J)] = (2.0 * (h2uq * h2vq)) /
                    (h_neglect3 + (h2uq + h2vq) * ((h_u[U2(I,j)] + h_u[U2(I,j+1)]) + (h_v[V2(i,J)] + h_v[V2(i+1,J)])));
    }
    if (CS->better_bound_Ah || CS->better_bound_Kh) {      /* :1430-1441 */
      for (int J = js - 1; J <= Jeq; J++) for (int I = is - 1; I <= Ieq; I++) {
        const int i = I, j = J;
        const double h_min = min4(h_u[U2(I,j)], h_u[U2(I,j+1)], h_v[V2(i,J)], h_v[V2(i+1,J)]);
        hrat_min[Q2(I,J)] = min2(1.0, h_min / (hq[Q2(I,J)] + h_neglect));
        if (CS->better_bound_Kh) visc_bound_rem[Q2(I,J)] = 1.0;
      }
    }
    if (CS->no_slip) {      /* coastal vorticity points :1443-1466 */
      for (int J = js - 1; J <= Jeq; J++) for (int I = is - 1; I <= Ieq; I++) {
        const int i = I, j = J;
        if (G->mask2dBu[Q2(I,J)] < 0.5) {
          if ((G->mask2dCu[U2(I,j)] + G->mask2dCu[U2(I,j+1)]) + (G->mask2dCv[V2(i,J)] + G->mask2dCv[V2(i+1,J)]) > 0.0) {
            const double hu = G->mask2dCu[U2(I,j)] * h_u[U2(I,j)] + G->mask2dCu[U2(I,j+1)] * h_u[U2(I,j+1)];
            const double hv = G->mask2dCv[V2(i,J)] * h_v[V2(i,J)] + G->mask2dCv[V2(i+1,J)] * h_v[V2(i+1,J)];
            if ((G->mask2dCu[U2(I,j)] + G->mask2dCu[U2(I,j+1)]) * (G->mask2dCv[V2(i,J)] + G->mask2dCv[V2(i+1,J)]) == 0.0) {
              hq[Q2(I,J)] = hu + hv;
              hrat_min[Q2(I,J)] = 1.0;
            } else {
              hq[Q2(I,J)] = 2.0 * (hu * hv) / ((hu + hv) + h_neglect);
              hrat_min[Q2(I,J)] = min2(1.0, min2(hu, hv) / (hq[Q2(I,J)] + h_neglect));
            }
          }
        }
      }
    }
    if (CS->Laplacian) {      /* Kh at q points and the Laplacian part of str_xy :1473-1585 */
      for (int J = js - 1; J <= Jeq; J++) for (int I = is - 1; I <= Ieq; I++) {
        double K_ = CS->Kh_bg_xy[Q2(I,J)];
        if (CS->Smagorinsky_Kh) {
          if (CS->add_LES_viscosity) K_ = K_ + CS->Laplac2_const_xy[Q2(I,J)] * Shear_mag[Q2(I,J)];
          else K_ = max2(K_, CS->Laplac2_const_xy[Q2(I,J)] * Shear_mag[Q2(I,J)]);
        }
        if (legacy_bound) K_ = min2(K_, CS->Kh_Max_xy[Q2(I,J)]);
        K_ = max2(K_, CS->Kh_bg_min);
        if (CS->MEKE_Ku) {      /* :1537-1541 (meke_res_fn = 1) */
          const int i = I, j = J;
          K_ = K_ + 0.25 * ((CS->MEKE_Ku[H2(i,j)] + CS->MEKE_Ku[H2(i+1,j+1)]) + (CS->MEKE_Ku[H2(i+1,j)] + CS->MEKE_Ku[H2(i,j+1)])) * 1.0;
        }
        if (CS->better_bound_Kh) {
          if (K_ >= hrat_min[Q2(I,J)] * CS->Kh_Max_xy[Q2(I,J)]) {
            visc_bound_rem[Q2(I,J)] = 0.0;
            K_ = hrat_min[Q2(I,J)] * CS->Kh_Max_xy[Q2(I,J)];
          } else if (hrat_min[Q2(I,J)] * CS->Kh_Max_xy[Q2(I,J)] > 0.) {
            visc_bound_rem[Q2(I,J)] = 1.0 - K_ / (hrat_min[Q2(I,J)] * CS->Kh_Max_xy[Q2(I,J)]);
          }
        }
        Kh[Q2(I,J)] = K_;
        str_xy[Q2(I,J)] = -K_ * sh_xy[Q2(I,J)];
      }
    } else {
      for (int J = js - 1; J <= Jeq; J++) for (int I = is - 1; I <= Ieq; I++) str_xy[Q2(I,J)] = 0.;
    }
    if (CS->biharmonic) {      /* Ah at q points and the biharmonic part of str_xy :1598-1693 */
      for (int J = js - 1; J <= Jeq; J++) for (int I = is - 1; I <= Ieq; I++) {
        double A_ = CS->Ah_bg_xy[Q2(I,J)];
        if (CS->Smagorinsky_Ah) {
          double AhSm;
          if (CS->bound_Coriolis)
            AhSm = Shear_mag[Q2(I,J)] * (CS->Biharm_const_xy[Q2(I,J)] + CS->Biharm_const2_xy[Q2(I,J)] * Shear_mag[Q2(I,J)]);
          else
            AhSm = CS->Biharm_const_xy[Q2(I,J)] * Shear_mag[Q2(I,J)];
          A_ = max2(A_, AhSm);
          if (CS->bound_Ah && !CS->better_bound_Ah) A_ = min2(A_, CS->Ah_Max_xy[Q2(I,J)]);
        }
        if (CS->MEKE_Au) {      /* :1634-1639 */
          const int i = I, j = J;
          A_ = A_ + 0.25 * ((CS->MEKE_Au[H2(i,j)] + CS->MEKE_Au[H2(i+1,j+1)]) + (CS->MEKE_Au[H2(i+1,j)] + CS->MEKE_Au[H2(i,j+1)]));
        }
        if (CS->better_bound_Ah) {
          if (CS->better_bound_Kh) A_ = min2(A_, visc_bound_rem[Q2(I,J)] * hrat_min[Q2(I,J)] * CS->Ah_Max_xy[Q2(I,J)]);
          else A_ = min2(A_, hrat_min[Q2(I,J)] * CS->Ah_Max_xy[Q2(I,J)]);
        }
        const double d_str = A_ * (dDel2vdx[Q2(I,J)] + dDel2udy[Q2(I,J)]);
        str_xy[Q2(I,J)] = str_xy[Q2(I,J)] + d_str;
      }
    }
    /* to layer-integrated stresses :1726-1741 */
    for (int j = Jsq; j <= Jeq + 1; j++) for (int i = Isq; i <= Ieq + 1; i++)
      str_xx[H2(i,j)] = str_xx[H2(i,j)] * (h[H3(i,j,k)] * CS->reduction_xx[H2(i,j)]);
    for (int J = js - 1; J <= Jeq; J++) for (int I = is - 1; I <= Ieq; I++) {
      if (CS->no_slip) str_xy[Q2(I,J)] = str_xy[Q2(I,J)] * (hq[Q2(I,J)] * CS->reduction_xy[Q2(I,J)]);
      else str_xy[Q2(I,J)] = str_xy[Q2(I,J)] * (hq[Q2(I,J)] * G->mask2dBu[Q2(I,J)] * CS->reduction_xy[Q2(I,J)]);
    }
    /* the accelerations :1744-1770 */
    for (int j = js; j <= je; j++) for (int I = Isq; I <= Ieq; I++) {
      const int i = I, J = j;
      diffu[U3(I,j,k)] = ((G->IdyCu[U2(I,j)] * (dy2h(i,j) * str_xx[H2(i,j)] - dy2h(i+1,j) * str_xx[H2(i+1,j)]) +
                           G->IdxCu[U2(I,j)] * (dx2q(I,J-1) * str_xy[Q2(I,J-1)] - dx2q(I,J) * str_xy[Q2(I,J)])) *
                          G->IareaCu[U2(I,j)]) / (h_u[U2(I,j)] + h_neglect);
    }
    if (apply_OBC) for (int n = 0; n < OBC->number_of_segments; n++) {      /* :1751-1762 */
      const mom6hip_obc_segment_t *S = &OBC->segment[n];
      if (S->on_pe && S->is_E_or_W) for (int j = S->jsd; j <= S->jed; j++) diffu[U3(S->IsdB,j,k)] = 0.;
    }
    for (int J = Jsq; J <= Jeq; J++) for (int i = is; i <= ie; i++) {
      const int I = i, j = J;
      diffv[V3(i,J,k)] = ((G->IdyCv[V2(i,J)] * (dy2q(I-1,J) * str_xy[Q2(I-1,J)] - dy2q(I,J) * str_xy[Q2(I,J)]) -
                           G->IdxCv[V2(i,J)] * (dx2h(i,j) * str_xx[H2(i,j)] - dx2h(i,j+1) * str_xx[H2(i,j+1)])) *
                          G->IareaCv[V2(i,J)]) / (h_v[V2(i,J)] + h_neglect);
    }
    if (apply_OBC) for (int n = 0; n < OBC->number_of_segments; n++) {      /* :1771-1782 */
      const mom6hip_obc_segment_t *S = &OBC->segment[n];
      if (S->on_pe && S->is_N_or_S) for (int i = S->isd; i <= S->ied; i++) diffv[V3(i,S->JsdB,k)] = 0.;
    }
    if (FrictWork) {      /* :1783-1800 */
      for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
        const int I = i, J = j;
        FrictWork[H3(i,j,k)] = G->H_to_Z * G->Rho0 * (
                (str_xx[H2(i,j)] * (u[U3(I,j,k)] - u[U3(I-1,j,k)]) * G->IdxT[H2(i,j)]
               - str_xx[H2(i,j)] * (v[V3(i,J,k)] - v[V3(i,J-1,k)]) * G->IdyT[H2(i,j)])
            + 0.25 * ((str_xy[Q2(I,J)] *
                       ((u[U3(I,j+1,k)] - u[U3(I,j,k)]) * G->IdyBu[Q2(I,J)]
                      + (v[V3(i+1,J,k)] - v[V3(i,J,k)]) * G->IdxBu[Q2(I,J)])
                     + str_xy[Q2(I-1,J-1)] *
                       ((u[U3(I-1,j,k)] - u[U3(I-1,j-1,k)]) * G->IdyBu[Q2(I-1,J-1)]
                      + (v[V3(i,J-1,k)] - v[V3(i-1,J-1,k)]) * G->IdxBu[Q2(I-1,J-1)]))
                    + (str_xy[Q2(I-1,J)] *
                       ((u[U3(I-1,j+1,k)] - u[U3(I-1,j,k)]) * G->IdyBu[Q2(I-1,J)]
                      + (v[V3(i,J,k)] - v[V3(i-1,J,k)]) * G->IdxBu[Q2(I-1,J)])
                     + str_xy[Q2(I,J-1)] *
                       ((u[U3(I,j,k)] - u[U3(I,j-1,k)]) * G->IdyBu[Q2(I,J-1)]
                      + (v[V3(i+1,J-1,k)] - v[V3(i,J-1,k)]) * G->IdxBu[Q2(I,J-1)]))));
      }
    }
  }
  free(dudx); free(dvdy); free(sh_xx); free(str_xx); free(dvdx); free(dudy); free(sh_xy); free(str_xy); free(hq);
  free(dDel2vdx); free(dDel2udy); free(h_u); free(Del2u); free(h_v); free(Del2v);
  free(Ah); free(Kh); free(Shear_mag); free(hrat_min); free(visc_bound_rem);
  }
  if (FrictWork) {
    for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
      double src = 0.;
      for (int k = 1; k <= nz; k++) src = src + FrictWork[H3(i,j,k)];
      CS->MEKE_mom_src[H2(i,j)] = src;
    }
    free(FrictWork);
  }
  return 0;
}
