/*
 * set_viscosity.c -- CPU restatement of set_viscous_BBL (TEST INFRASTRUCTURE, see mom6_oracle.h).
 *
 * Reference: src/parameterizations/vertical/MOM_set_viscosity.F90
 *   set_viscous_BBL :134-1100, set_v_at_u :1804, set_u_at_v :1849, set_viscous_ML :1898 (the early return :2043, and the
 *   DYNAMIC_VISCOUS_ML search :2111-2230, :2400-2506; its exp through orc_cr_exp, correctly rounded)
 * Restated branch: BOTTOMDRAGLAW, quadratic or LINEAR_DRAG law, BBL_USE_EOS or GV%Rlay as the density variable, CHANNEL_DRAG
 * (:863-1002 with find_L_open_* :1104-1800), no tidal background velocity, Boussinesq (no tv%SpV_avg), no tv%p_surf, no OBC; DRAG_AS_BODY_FORCE; CORRECT_BBL_BOUNDS.
 * PARITY UNPINNED: the reference holds no known-answer vectors for this module; invariants in tests/test_set_viscosity.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mom6_oracle.h"

static inline double min2(double a, double b) { return a < b ? a : b; }
static inline double max2(double a, double b) { return a > b ? a : b; }

#define H3(i,j,k) ORC_H3(G,i,j,k)
#define U3(i,j,k) ORC_U3(G,i,j,k)
#define V3(i,j,k) ORC_V3(G,i,j,k)

/* the direction of the segment at a face (OBC%segment(OBC%segnum(...))%direction), 0 without one */
static int seg_dir(const mom6hip_obc_t *OBC, const int32_t *segnum, long f2) {
  if (!OBC || OBC->number_of_segments <= 0 || segnum[f2] == MOM6HIP_OBC_NONE) return 0;
  return OBC->segment[segnum[f2] - 1].direction;
}

/* set_v_at_u :1804-1846; mask2dCv: the mask the caller passes (set_viscous_BBL: its work array mask_v) */
static double set_v_at_u(const mom6hip_grid_t *G, const double *v, const double *h, int i, int j, int k, const double *mask2dCv,
                         const mom6hip_obc_t *OBC) {
  const int J = j;
  double hwt[2][2];      /* hwt(i0, j0), i0 = 0:1, j0 = -1:0 -> hwt[i0][j0+1] */
  for (int j0 = -1; j0 <= 0; j0++) for (int i0 = 0; i0 <= 1; i0++) {
    const int i1 = i + i0, J1 = J + j0;
    hwt[i0][j0 + 1] = (h[H3(i1, J1, k)] + h[H3(i1, J1 + 1, k)]) * mask2dCv[ORC_V2(G, i1, J1)];
    const int dir = seg_dir(OBC, OBC ? OBC->segnum_v : NULL, ORC_V2(G, i1, J1));      /* :1829-1838 */
    if (dir == MOM6HIP_OBC_DIRECTION_N) hwt[i0][j0 + 1] = 2.0 * h[H3(i1, J1, k)] * mask2dCv[ORC_V2(G, i1, J1)];
    else if (dir == MOM6HIP_OBC_DIRECTION_S) hwt[i0][j0 + 1] = 2.0 * h[H3(i1, J1 + 1, k)] * mask2dCv[ORC_V2(G, i1, J1)];
  }
  const double hwt_tot = (hwt[0][0] + hwt[1][1]) + (hwt[1][0] + hwt[0][1]);
  double r = 0.0;
  if (hwt_tot > 0.0)
    r = ((hwt[0][1] * v[V3(i, J, k)] + hwt[1][0] * v[V3(i + 1, J - 1, k)]) +
         (hwt[1][1] * v[V3(i + 1, J, k)] + hwt[0][0] * v[V3(i, J - 1, k)])) / hwt_tot;
  return r;
}

/* set_u_at_v :1849-1891 */
static double set_u_at_v(const mom6hip_grid_t *G, const double *u, const double *h, int i, int j, int k, const double *mask2dCu,
                         const mom6hip_obc_t *OBC) {
  const int I = i;
  double hwt[2][2];      /* hwt(i0, j0), i0 = -1:0, j0 = 0:1 -> hwt[i0+1][j0] */
  for (int j0 = 0; j0 <= 1; j0++) for (int i0 = -1; i0 <= 0; i0++) {
    const int I1 = I + i0, j1 = j + j0;
    hwt[i0 + 1][j0] = (h[H3(I1, j1, k)] + h[H3(I1 + 1, j1, k)]) * mask2dCu[ORC_U2(G, I1, j1)];
    const int dir = seg_dir(OBC, OBC ? OBC->segnum_u : NULL, ORC_U2(G, I1, j1));      /* :1874-1883 */
    if (dir == MOM6HIP_OBC_DIRECTION_E) hwt[i0 + 1][j0] = 2.0 * h[H3(I1, j1, k)] * mask2dCu[ORC_U2(G, I1, j1)];
    else if (dir == MOM6HIP_OBC_DIRECTION_W) hwt[i0 + 1][j0] = 2.0 * h[H3(I1 + 1, j1, k)] * mask2dCu[ORC_U2(G, I1, j1)];
  }
  const double hwt_tot = (hwt[0][0] + hwt[1][1]) + (hwt[1][0] + hwt[0][1]);
  double r = 0.0;
  if (hwt_tot > 0.0)
    r = ((hwt[1][0] * u[U3(I, j, k)] + hwt[0][1] * u[U3(I - 1, j + 1, k)]) +
         (hwt[0][0] * u[U3(I - 1, j, k)] + hwt[1][1] * u[U3(I, j + 1, k)])) / hwt_tot;
  return r;
}


/* ---- CHANNEL_DRAG: the normalized open width L(K) of a velocity cell at each interface (arrays indexed 1..nz+1) ------------------- */
#define C1_3 (1.0 / 3.0)
#define C1_6 (1.0 / 6.0)
#define C1_12 (1.0 / 12.0)
#define C2PI_3 (8.0 * 0.78539816339744830962 / 3.0)      /* 8.0*atan(1.0)/3.0 */

/* find_L_open_uniform_slope :1104-1140 */
static void find_L_open_uniform_slope(int nz, const double *vol_below, double Dp, double Dm, double *L) {
  const double slope = fabs(Dp - Dm);
  if (slope == 0.0) {
    for (int K = 1; K <= nz; K++) L[K] = 1.0;
    L[nz + 1] = 0.0;
  } else {
    const double Vol_open = 0.5 * slope, I_slope = 1.0 / slope;
    L[nz + 1] = 0.0;
    for (int K = nz; K >= 1; K--) {
      if (vol_below[K] >= Vol_open) L[K] = 1.0;
      else L[K] = sqrt(2.0 * vol_below[K] * I_slope);
    }
  }
}

/* find_L_open_concave_trigonometric :1144-1231 (cos and acos correctly rounded: cr_trig.c) */
static void find_L_open_concave_trigonometric(int nz, const double *vol_below, double D_vel, double Dp, double Dm, double *L) {
  const double crv_3 = (Dp + Dm - (2.0 * D_vel)), crv = 3.0 * crv_3;
  const double slope = Dp - Dm;
  double Vol_open, Vol_2_reg;
  if (slope >= crv) {
    Vol_open = D_vel - Dm; Vol_2_reg = Vol_open;
  } else {
    const double slope_crv = slope / crv;
    Vol_open = 0.25 * slope * slope_crv + C1_12 * crv;
    Vol_2_reg = 0.5 * (slope_crv * slope_crv) * (crv - C1_3 * slope);
  }
  const double C24_crv = 24.0 / crv, Iapb = 1.0 / (crv + slope);
  const double apb_4a = (slope + crv) / (4.0 * crv), a2x48_apb3 = (48.0 * (crv * crv)) * ((Iapb * Iapb) * Iapb);
  const double ax2_3apb = 2.0 * C1_3 * crv * Iapb;
  L[nz + 1] = 0.0;
  for (int K = nz; K >= 1; K--) {
    if (vol_below[K] >= Vol_open) {
      L[K] = 1.0;
    } else if (vol_below[K] < Vol_2_reg) {
      if (a2x48_apb3 * vol_below[K] < 1e-8) {
        const double L0 = sqrt(2.0 * vol_below[K] * Iapb);
        L[K] = L0 * (1.0 + (ax2_3apb * L0));
      } else {
        L[K] = apb_4a * (1.0 - 2.0 * orc_cr_cos(C1_3 * orc_cr_acos((a2x48_apb3 * vol_below[K]) - 1.0) - C2PI_3));
      }
    } else {
      double tmp_val_m1_to_p1 = 1.0 - C24_crv * (Vol_open - vol_below[K]);
      tmp_val_m1_to_p1 = max2(-1., min2(1., tmp_val_m1_to_p1));
      L[K] = 0.5 - orc_cr_cos(C1_3 * orc_cr_acos(tmp_val_m1_to_p1) - C2PI_3);
    }
  }
}

/* find_L_open_concave_iterative :1237-1556 */
static void find_L_open_concave_iterative(int nz, const double *vol_below, double D_vel, double Dp, double Dm, double *L) {
  const int max_itt = 10;
  const double crv_3 = (Dp + Dm - 2.0 * D_vel), crv = 3.0 * crv_3;
  const double slope = Dp - Dm;
  double Vol_open, Vol_2_reg, L_2_reg, L_inflect_1, vol_inflect_1, vol_inflect_2 = 0.0, slope_crv = 0.0, smc = 0.0, C3c_m_s = 0.0, I_3c_m_s = 0.0;
  double C4_crv = 0.0, slope2_4crv = 0.0, sxcms_c = 0.0, C3s_m_c = 0.0, I_3s_m_c = 0.0;
  if (slope >= crv) {
    Vol_open = D_vel - Dm; Vol_2_reg = Vol_open;
    L_2_reg = 1.0;
    if (crv + slope >= 4.0 * crv) {
      L_inflect_1 = 1.0; vol_inflect_1 = Vol_open;
    } else {
      slope_crv = slope / crv;
      L_inflect_1 = 0.25 + 0.25 * slope_crv;
      vol_inflect_1 = 0.25 * C1_12 * (((slope_crv + 1.0) * (slope_crv + 1.0)) * (slope + crv));
    }
    smc = slope - crv;
    C3c_m_s = 3.0 * crv - slope;
    if (C3c_m_s > 2.0 * smc) I_3c_m_s = 1.0 / C3c_m_s;
  } else {
    slope_crv = slope / crv;
    Vol_open = 0.25 * slope * slope_crv + C1_12 * crv;
    Vol_2_reg = 0.5 * (slope_crv * slope_crv) * (crv - C1_3 * slope);
    L_2_reg = slope_crv;
    vol_inflect_1 = 0.25 * C1_12 * (((slope_crv + 1.0) * (slope_crv + 1.0)) * (slope + crv));
    L_inflect_1 = 0.25 + 0.25 * slope_crv;
    vol_inflect_2 = 0.25 * slope * slope_crv + 0.125 * crv_3;
    C4_crv = 4.0 / crv;
    slope2_4crv = 0.25 * slope * slope_crv;
    sxcms_c = slope_crv * (crv - slope);
    C3s_m_c = 3.0 * slope - crv;
    if (C3s_m_c > 2.0 * sxcms_c) I_3s_m_c = 1.0 / C3s_m_c;
  }
  const double Icrvpslope = 1.0 / (crv + slope);
#define VERR1(Lk) (0.5 * ((Lk) * (Lk)) * (slope + crv * (1.0 - 4.0 * C1_3 * (Lk))) - vol_below[K])
#define VERR2(Lk) (crv_3 * (((Lk) * (Lk)) * (0.75 - 0.5 * (Lk))) + (slope2_4crv - vol_below[K]))
  L[nz + 1] = 0.0;
  for (int K = nz; K >= 1; K--) {
    double L_max, L_min, vol_err, dVol_dL, vol_err_max;
    if (vol_below[K] >= Vol_open) {
      L[K] = 1.0;
    } else if (vol_below[K] < Vol_2_reg) {
      L_max = min2(L_2_reg, 1.0);
      if (vol_below[K] <= vol_inflect_1) L_max = min2(L_max, L_inflect_1);
      L_min = L[K + 1];
      if (vol_below[K] >= vol_inflect_1) L_min = max2(L_min, L_inflect_1);
      if (2.0 * vol_below[K] * Icrvpslope > L_min * L_min) L_min = sqrt(2.0 * vol_below[K] * Icrvpslope);
      L[K] = L_min;
      if (vol_below[K] <= vol_inflect_1) {
        L[K] = L_min;
        vol_err = VERR1(L[K]);
        if (vol_err < 0.0) {
          dVol_dL = L[K] * (slope + crv * (1.0 - 2.0 * L[K]));
          if (L[K] * dVol_dL > vol_err + L_max * dVol_dL) L[K] = L_max;
          else L[K] = L[K] - (vol_err / dVol_dL);
          for (int itt = 1; itt <= max_itt; itt++) {
            vol_err = VERR1(L[K]);
            dVol_dL = L[K] * (slope + crv * (1.0 - 2.0 * L[K]));
            if (fabs(vol_err) < max2(1.0e-15 * L[K], 1.0e-25) * dVol_dL) break;
            L[K] = L[K] - (vol_err / dVol_dL);
          }
        }
      } else {
        L[K] = L_min;
        vol_err = VERR1(L[K]);
        if (vol_err < 0.0) {
          if (slope < crv) {
            if ((L_2_reg - L_min) * C3s_m_c > 2.0 * sxcms_c)
              L_max = (slope_crv * (2.0 * slope) - sqrt(sxcms_c * sxcms_c + 2.0 * C3s_m_c * (Vol_2_reg - vol_below[K]))) * I_3s_m_c;
            else
              L_max = slope_crv;
          } else {
            if ((1.0 - L_min) * C3c_m_s > 2.0 * smc)
              L_max = (2.0 * crv - sqrt(smc * smc + 2.0 * C3c_m_s * (Vol_open - vol_below[K]))) * I_3c_m_s;
            else
              L_max = 1.0;
          }
          vol_err_max = VERR1(L_max);
          if ((vol_err_max < fabs(vol_err)) && (L_max < 1.0)) {
            dVol_dL = L_max * (slope + crv * (1.0 - 2.0 * L_max));
            L[K] = max2(L_min, L_max - (vol_err_max / dVol_dL));
          }
          for (int itt = 1; itt <= max_itt; itt++) {
            vol_err = VERR1(L[K]);
            dVol_dL = L[K] * (slope + crv * (1.0 - 2.0 * L[K]));
            if (fabs(vol_err) < max2(1.0e-15 * L[K], 1.0e-25) * dVol_dL) break;
            L[K] = L[K] - (vol_err / dVol_dL);
          }
        }
      }
    } else {      /* two separate open regions */
      if (vol_below[K] <= vol_inflect_2) {
        L_min = max2(L[K + 1], L_2_reg);
        if ((4.0 * vol_below[K] - slope * slope_crv) > (crv + 2.0 * C1_3 * slope) * (L_min * L_min))
          L_min = max2(L_min, sqrt((4.0 * vol_below[K] - slope * slope_crv) / (crv + 2.0 * C1_3 * slope)));
        L_max = 0.5;
        L[K] = L_min;
        vol_err = crv_3 * (L[K] * L[K]) * (0.75 - 0.5 * L[K]) + (slope2_4crv - vol_below[K]);
        if (vol_err < 0.0) {
          dVol_dL = 0.5 * crv * (L[K] * (1.0 - L[K]));
          if (L[K] * dVol_dL >= vol_err + L_max * dVol_dL) L[K] = L_max;
          else L[K] = L[K] - (vol_err / dVol_dL);
          for (int itt = 1; itt <= max_itt; itt++) {
            vol_err = VERR2(L[K]);
            dVol_dL = 0.5 * crv * (L[K] * (1.0 - L[K]));
            if (fabs(vol_err) < max2(1.0e-15 * L[K], 1.0e-25) * dVol_dL) break;
            L[K] = L[K] - (vol_err / dVol_dL);
          }
        }
      } else {
        L_min = max2(L[K + 1], 0.5);
        L[K] = L_min;
        vol_err = VERR2(L[K]);
        if (vol_err < 0.0) {
          L_max = 1.0 - sqrt((Vol_open - vol_below[K]) * C4_crv);
          vol_err_max = VERR2(L_max);
          if ((vol_err_max < fabs(vol_err)) && (L_max < 1.0)) {
            dVol_dL = 0.5 * crv * (L_max * (1.0 - L_max));
            L[K] = max2(L_min, L_max - (vol_err_max / dVol_dL));
          }
          for (int itt = 1; itt <= max_itt; itt++) {
            vol_err = VERR2(L[K]);
            dVol_dL = 0.5 * crv * (L[K] * (1.0 - L[K]));
            if (fabs(vol_err) < max2(1.0e-15 * L[K], 1.0e-25) * dVol_dL) break;
            L[K] = L[K] - (vol_err / dVol_dL);
          }
        }
      }
    }
  }
#undef VERR1
#undef VERR2
}

/* find_L_open_convex :1640-1800, SET_VISC_ANSWER_DATE >= 20190101; the cube root through orc_cr_pow.  Angstrom_Z = GV%Angstrom_Z,
 * dZ_subroundoff = GV%dZ_subroundoff */
static void find_L_open_convex(int nz, const double *vol_below, double D_vel, double Dp, double Dm, double *L, double Angstrom_Z,
                               double dZ_subroundoff) {
  const int maxitt = 20;
  const double crv_3 = (Dp + Dm - 2.0 * D_vel), crv = 3.0 * crv_3;
  const double slope = Dp - Dm;
  const double Vol_open = D_vel - Dm;
  double Vol_direct, L_direct, C24_crv;
  if (slope >= -crv) {
    Vol_direct = 0.0; L_direct = 0.0; C24_crv = 0.0;
  } else {
    C24_crv = 24.0 / crv;
    L_direct = 1.0 + slope / crv;
    Vol_direct = -C1_6 * crv * ((L_direct * L_direct) * L_direct);
  }
  const double Ibma_2 = 2.0 / (slope - crv);
  double Vol_err = 0.0;
  L[nz + 1] = 0.0;
#define VERR(Lk) (0.5 * ((Lk) * (Lk)) * (slope + crv_3 * (3.0 - 4.0 * (Lk))) - vol_below[K])
  for (int K = nz; K >= 1; K--) {
    if (vol_below[K] >= Vol_open) {
      L[K] = 1.0;
    } else if (vol_below[K] <= Vol_direct) {
      const double x = -0.25 * C24_crv * vol_below[K];
      L[K] = (x > 0.0) ? orc_cr_pow(x, C1_3) : 0.0;
    } else {
      double L0, Vol_0;
      if (vol_below[K + 1] + Vol_err <= Vol_direct) { L0 = L_direct; Vol_0 = Vol_direct; }
      else { L0 = L[K + 1]; Vol_0 = vol_below[K + 1] + Vol_err; }
      const double dV_dL2 = 0.5 * (slope + crv) - crv * L0, dVol = (vol_below[K] - Vol_0);
      const int use_L0 = (dVol <= 0.);
      const double Vol_tol = max2(0.5 * Angstrom_Z + dZ_subroundoff, 1e-14 * vol_below[K]);
      const double Vol_quit = max2(0.9 * Angstrom_Z + dZ_subroundoff, 1e-14 * vol_below[K]);
      const double curv_tol = Vol_tol * (dV_dL2 * dV_dL2) * (dV_dL2 * Vol_tol - 2.0 * crv * L0 * dVol);
      const int do_one_L_iter = (crv * crv * ((dVol * dVol) * dVol)) < curv_tol;
      if (use_L0) {
        L[K] = L0;
        Vol_err = VERR(L[K]);
      } else if (do_one_L_iter) {
        L[K] = sqrt(L0 * L0 + dVol / dV_dL2);
        Vol_err = VERR(L[K]);
      } else {
        double L_max, L_min;
        if (dV_dL2 * (1.0 - L0 * L0) < dVol + dV_dL2 * (Vol_open - vol_below[K]) * Ibma_2)
          L_max = sqrt(1.0 - (Vol_open - vol_below[K]) * Ibma_2);
        else
          L_max = sqrt(L0 * L0 + dVol / dV_dL2);
        L_min = sqrt(L0 * L0 + dVol / (0.5 * (slope + crv) - crv * L_max));
        const double Vol_err_min = VERR(L_min), Vol_err_max = VERR(L_max);
        if (fabs(Vol_err_min) <= Vol_quit) {
          L[K] = L_min; Vol_err = Vol_err_min;
        } else {
          L[K] = sqrt(((L_min * L_min) * Vol_err_max - (L_max * L_max) * Vol_err_min) / (Vol_err_max - Vol_err_min));
          for (int itt = 1; itt <= maxitt; itt++) {
            Vol_err = VERR(L[K]);
            if (fabs(Vol_err) <= Vol_quit) break;
            L[K] = L[K] - Vol_err / (L[K] * (slope + crv - 2.0 * crv * L[K]));
          }
        }
      }
    }
  }
#undef VERR
}

static int unsupported(const mom6hip_set_visc_cs_t *CS) {
  for (int n = 0; n < 9; n++) if (CS->unsupported[n]) return 1;
  return 0;
}

/* set_viscous_ML :1898, the DYNAMIC_VISCOUS_ML part (:2111-2230 at u points, :2400-2506 at v points): one velocity column.
 * dir 0: the face (I,j) between the cells (i,j) and (i+1,j); dir 1: the face (i,J) between (i,j) and (i,j+1).  Boussinesq, no
 * tv%p_surf, no ice shelf.  The row-wide do_any / exit of the reference only skips work: a column's result does not depend on
 * its neighbours. */
static double viscous_ML_column(const mom6hip_grid_t *G, const mom6hip_set_visc_cs_t *CS, const double *u, const double *v, const double *h,
                                const double *T, const double *S, const mom6hip_eos_t *EOS, const double *taux, const double *tauy,
                                const double *ustar, double dt, int dir, int i, int j) {
  const int nz = G->nk, nkml = CS->nkml;
  const int use_EOS = (EOS != NULL);
  const double dt_Rho0 = dt / CS->H_to_RZ;
  const double h_neglect = G->H_subroundoff;
  const double h_tiny = 2.0 * G->Angstrom_H + h_neglect;
  const double g_H_Rho0 = (G->g_Earth * G->H_to_Z) / (G->Rho0);
  const int i2 = dir ? i : i + 1, j2 = dir ? j + 1 : j;      /* the second cell of the face */
  const double mask = dir ? G->mask2dCv[ORC_V2(G, i, j)] : G->mask2dCu[ORC_U2(G, i, j)];
  if (mask < 0.5) return (double)nkml;
  double htot = 0.0, Thtot = 0.0, Shtot = 0.0, Rhtot = 0.0, uhtot, vhtot, absf;
  int k_massive = nkml;
  if (!dir) {      /* :2128-2131 */
    uhtot = dt_Rho0 * taux[ORC_U2(G, i, j)];
    vhtot = 0.25 * dt_Rho0 * ((tauy[ORC_V2(G, i, j)] + tauy[ORC_V2(G, i + 1, j - 1)]) +
                              (tauy[ORC_V2(G, i, j - 1)] + tauy[ORC_V2(G, i + 1, j)]));
  } else {         /* :2411-2413 */
    vhtot = dt_Rho0 * tauy[ORC_V2(G, i, j)];
    uhtot = 0.25 * dt_Rho0 * ((taux[ORC_U2(G, i, j)] + taux[ORC_U2(G, i - 1, j + 1)]) +
                              (taux[ORC_U2(G, i - 1, j)] + taux[ORC_U2(G, i, j + 1)]));
  }
  if (CS->omega_frac >= 1.0) absf = 2.0 * CS->omega;      /* :2133-2140 */
  else {
    if (!dir) absf = 0.5 * (fabs(G->CoriolisBu[ORC_Q2(G, i, j)]) + fabs(G->CoriolisBu[ORC_Q2(G, i, j - 1)]));
    else absf = 0.5 * (fabs(G->CoriolisBu[ORC_Q2(G, i - 1, j)]) + fabs(G->CoriolisBu[ORC_Q2(G, i, j)]));
    if (CS->omega_frac > 0.0) absf = sqrt(CS->omega_frac * 4.0 * (CS->omega * CS->omega) + (1.0 - CS->omega_frac) * (absf * absf));
  }
  /* find_ustar(..., H_T_units=.true.), Boussinesq: U_star_2d = GV%Z_to_H * forces%ustar (MOM_forcing_type.F90:1272) */
  const double U_star = max2(CS->ustar_min, 0.5 * (G->Z_to_H * ustar[ORC_H2(G, i, j)] + G->Z_to_H * ustar[ORC_H2(G, i2, j2)]));
  const double Idecay_len_TKE = (absf / U_star) * CS->TKE_decay;
  double dR_dT = 0.0, dR_dS = 0.0, result = 0.0;
  int active = 1;
#define HA(k) h[ORC_H3(G, i, j, k)]
#define HB(k) h[ORC_H3(G, i2, j2, k)]
  for (int k = 1; k <= nz; k++) {
    /* the velocity of the other component at this point: v at u (:2168-2169) or u at v (:2449-2450) */
    if (k > nkml) {
      if (use_EOS && (k == nkml + 1)) {      /* :2147-2164 */
        const double press = (CS->H_to_RZ * G->g_Earth) * htot;
        const int k2 = nkml > 1 ? nkml : 1;
        const double I_2hlay = 1.0 / (HA(k2) + HB(k2) + h_neglect);
        const double T_EOS = (HA(k2) * T[ORC_H3(G, i, j, k2)] + HB(k2) * T[ORC_H3(G, i2, j2, k2)]) * I_2hlay;
        const double S_EOS = (HA(k2) * S[ORC_H3(G, i, j, k2)] + HB(k2) * S[ORC_H3(G, i2, j2, k2)]) * I_2hlay;
        orc_eos_density_derivs(EOS, T_EOS, S_EOS, press, &dR_dT, &dR_dS);
      }
      const double hlay = 0.5 * (HA(k) + HB(k));
      if (hlay > h_tiny) {
        const double I_2hlay = 1.0 / (HA(k) + HB(k));
        double Uh2;
        if (!dir) {
          const double v_at_u = 0.5 * (HA(k) * (v[ORC_V3(G, i, j, k)] + v[ORC_V3(G, i, j - 1, k)]) +
                                       HB(k) * (v[ORC_V3(G, i + 1, j, k)] + v[ORC_V3(G, i + 1, j - 1, k)])) * I_2hlay;
          const double du = uhtot - htot * u[ORC_U3(G, i, j, k)], dv = vhtot - htot * v_at_u;
          Uh2 = (du * du + dv * dv);
        } else {
          const double u_at_v = 0.5 * (HA(k) * (u[ORC_U3(G, i - 1, j, k)] + u[ORC_U3(G, i, j, k)]) +
                                       HB(k) * (u[ORC_U3(G, i - 1, j + 1, k)] + u[ORC_U3(G, i, j + 1, k)])) * I_2hlay;
          const double du = uhtot - htot * u_at_v, dv = vhtot - htot * v[ORC_V3(G, i, j, k)];
          Uh2 = (du * du + dv * dv);
        }
        double gHprime;
        if (use_EOS) {
          const double T_lay = (HA(k) * T[ORC_H3(G, i, j, k)] + HB(k) * T[ORC_H3(G, i2, j2, k)]) * I_2hlay;
          const double S_lay = (HA(k) * S[ORC_H3(G, i, j, k)] + HB(k) * S[ORC_H3(G, i2, j2, k)]) * I_2hlay;
          gHprime = g_H_Rho0 * (dR_dT * (T_lay * htot - Thtot) + dR_dS * (S_lay * htot - Shtot));
        } else {
          gHprime = g_H_Rho0 * (CS->Rlay[k - 1] * htot - Rhtot);
        }
        if (gHprime > 0.0) {
          const double RiBulk = CS->bulk_Ri_ML * orc_cr_exp(-htot * Idecay_len_TKE);
          if (RiBulk * Uh2 <= (htot * htot) * gHprime) {
            result = (double)k_massive;
            active = 0;
          } else if (RiBulk * Uh2 <= ((htot + hlay) * (htot + hlay)) * gHprime) {
            result = (double)(k - 1) + (sqrt(RiBulk * Uh2 / gHprime) - htot) / hlay;
            active = 0;
          }
        }
        k_massive = k;
      }
      if (!active) break;
    }
    /* :2195-2207 / :2476-2488 */
    htot = htot + 0.5 * (HA(k) + HB(k));
    if (!dir) {
      uhtot = uhtot + 0.5 * (HA(k) + HB(k)) * u[ORC_U3(G, i, j, k)];
      vhtot = vhtot + 0.25 * (HA(k) * (v[ORC_V3(G, i, j, k)] + v[ORC_V3(G, i, j - 1, k)]) +
                              HB(k) * (v[ORC_V3(G, i + 1, j, k)] + v[ORC_V3(G, i + 1, j - 1, k)]));
    } else {
      vhtot = vhtot + 0.5 * (HA(k) + HB(k)) * v[ORC_V3(G, i, j, k)];
      uhtot = uhtot + 0.25 * (HA(k) * (u[ORC_U3(G, i - 1, j, k)] + u[ORC_U3(G, i, j, k)]) +
                              HB(k) * (u[ORC_U3(G, i - 1, j + 1, k)] + u[ORC_U3(G, i, j + 1, k)]));
    }
    if (use_EOS) {
      Thtot = Thtot + 0.5 * (HA(k) * T[ORC_H3(G, i, j, k)] + HB(k) * T[ORC_H3(G, i2, j2, k)]);
      Shtot = Shtot + 0.5 * (HA(k) * S[ORC_H3(G, i, j, k)] + HB(k) * S[ORC_H3(G, i2, j2, k)]);
    } else {
      Rhtot = Rhtot + 0.5 * (HA(k) + HB(k)) * CS->Rlay[k - 1];
    }
  }
#undef HA
#undef HB
  if (active) result = (double)k_massive;      /* :2210-2212 */
  return result;
}

int orc_set_viscous_ML(const mom6hip_grid_t *G, const mom6hip_set_visc_cs_t *CS, const double *u, const double *v, const double *h,
                       const double *T, const double *S, const mom6hip_eos_t *EOS, const double *taux, const double *tauy,
                       const mom6hip_vertvisc_type_t *visc, double dt) {
  if (!CS->initialized) return 3;
  if (unsupported(CS)) return 1;
  if (!CS->dynamic_viscous_ML) return 0;      /* :2043-2044: nothing to do without DYNAMIC_VISCOUS_ML or an ice shelf */
  if (!(u && v && h && taux && tauy && visc && visc->ustar && visc->nkml_visc_u && visc->nkml_visc_v)) return 2;
  if (EOS ? !(T && S) : !CS->Rlay) return 2;
  double *nu = (double *)visc->nkml_visc_u, *nv = (double *)visc->nkml_visc_v;
  ORC_PAR
  for (int j = G->jsc; j <= G->jec; j++) for (int I = G->isc - 1; I <= G->iec; I++)
    nu[ORC_U2(G, I, j)] = viscous_ML_column(G, CS, u, v, h, T, S, EOS, taux, tauy, visc->ustar, dt, 0, I, j);
  ORC_PAR
  for (int J = G->jsc - 1; J <= G->jec; J++) for (int i = G->isc; i <= G->iec; i++)
    nv[ORC_V2(G, i, J)] = viscous_ML_column(G, CS, u, v, h, T, S, EOS, taux, tauy, visc->ustar, dt, 1, i, J);
  return 0;
}

int orc_set_viscous_BBL(const mom6hip_grid_t *G, const mom6hip_set_visc_cs_t *CS, const double *u, const double *v, const double *h,
                        const double *T, const double *S, const mom6hip_eos_t *EOS, const mom6hip_vertvisc_type_t *visc) {
  return orc_set_viscous_BBL_obc(G, CS, u, v, h, T, S, EOS, visc, NULL);
}

/* set_viscous_BBL with CS%OBC associated (set_visc_init :2903): the depths and masks of the faces at and beside the segments
 * :374-413, the zero-gradient projection of the thicknesses, T and S :502-580, the weights of set_v_at_u / set_u_at_v */
int orc_set_viscous_BBL_obc(const mom6hip_grid_t *G, const mom6hip_set_visc_cs_t *CS, const double *u, const double *v, const double *h,
                            const double *T, const double *S, const mom6hip_eos_t *EOS, const mom6hip_vertvisc_type_t *visc,
                            const mom6hip_obc_t *OBC) {
  if (!CS->initialized) return 3;      /* "MOM_set_viscosity(BBL): Module must be initialized before it is used." */
  if (unsupported(CS)) return 1;
  if (!CS->bottomdraglaw) return 0;      /* :321 */
  const int nz = G->nk;
  const int Isq = G->isc - 1, Ieq = G->iec, Jsq = G->jsc - 1, Jeq = G->jec;
  const double h_neglect = G->H_subroundoff, dz_neglect = G->dZ_subroundoff;
  const double Rho0x400_G = 400.0 * (CS->H_to_RZ / (1.0 * 1.0 * G->g_Earth));      /* US%L_to_Z = 1 */
  const int use_BBL_EOS = (EOS != NULL) && CS->BBL_use_EOS;
  if (use_BBL_EOS && !(T && S)) return 2;
  if (!use_BBL_EOS && !CS->Rlay) return 2;
  double *bbl_thick_u = (double *)visc->bbl_thick_u, *bbl_thick_v = (double *)visc->bbl_thick_v;
  double *Kv_bbl_u = (double *)visc->Kv_bbl_u, *Kv_bbl_v = (double *)visc->Kv_bbl_v;
  double *Ray_u = (double *)visc->Ray_u, *Ray_v = (double *)visc->Ray_v;
  if (!bbl_thick_u || !bbl_thick_v) return 2;
  if ((CS->body_force_drag || CS->Channel_drag) && !(Ray_u && Ray_v)) return 2;
  const double cdrag_sqrt = sqrt(CS->cdrag);
  const double cdrag_sqrt_H = cdrag_sqrt * 1.0 * G->Z_to_H;      /* US%L_to_m*GV%m_to_H */
  const double cdrag_L_to_H = CS->cdrag * 1.0 * G->Z_to_H;
  const double BBL_thick_max = CS->BBL_thick_max;
  const int K2 = 2;      /* max(nkmb+1, 2) with nkmb = 0 */
  const long upl = (long)(ORC_NIH(G) + 1) * ORC_NJH(G), vpl = (long)ORC_NIH(G) * (ORC_NJH(G) + 1);
  if (OBC && OBC->number_of_segments > 0 && !(OBC->segment && OBC->segnum_u && OBC->segnum_v)) return 2;
  /* :363-413: the depths and the masks of the faces, as work arrays (over the ranges the reference fills) */
  double *D_u = (double *)calloc((size_t)(2 * upl + 2 * vpl), sizeof(double)), *mask_u = D_u + upl, *D_v = mask_u + upl, *mask_v = D_v + vpl;
  {
    const int is = G->isc, ie = G->iec, js = G->jsc, je = G->jec;
    for (int J = js - 1; J <= je; J++) for (int i = is - 1; i <= ie + 1; i++) {
      D_v[ORC_V2(G, i, J)] = 0.5 * (G->bathyT[ORC_H2(G, i, J)] + G->bathyT[ORC_H2(G, i, J + 1)]) + CS->Z_ref;
      mask_v[ORC_V2(G, i, J)] = G->mask2dCv[ORC_V2(G, i, J)];
    }
    for (int j = js - 1; j <= je + 1; j++) for (int I = is - 1; I <= ie; I++) {
      D_u[ORC_U2(G, I, j)] = 0.5 * (G->bathyT[ORC_H2(G, I, j)] + G->bathyT[ORC_H2(G, I + 1, j)]) + CS->Z_ref;
      mask_u[ORC_U2(G, I, j)] = G->mask2dCu[ORC_U2(G, I, j)];
    }
#define MAX2I(a, b) ((a) > (b) ? (a) : (b))
#define MIN2I(a, b) ((a) < (b) ? (a) : (b))
    if (OBC) for (int n = 0; n < OBC->number_of_segments; n++) {      /* :374-389: a one-sided projection of the depths at the segments' faces */
      const mom6hip_obc_segment_t *Sg = &OBC->segment[n];
      if (!Sg->on_pe) continue;
      const int I = Sg->IsdB, J = Sg->JsdB;
      if (Sg->is_N_or_S && (J >= js - 1) && (J <= je)) {
        for (int i = MAX2I(is - 1, Sg->isd); i <= MIN2I(ie + 1, Sg->ied); i++) {
          if (Sg->direction == MOM6HIP_OBC_DIRECTION_N) D_v[ORC_V2(G, i, J)] = G->bathyT[ORC_H2(G, i, J)] + CS->Z_ref;
          if (Sg->direction == MOM6HIP_OBC_DIRECTION_S) D_v[ORC_V2(G, i, J)] = G->bathyT[ORC_H2(G, i, J + 1)] + CS->Z_ref;
        }
      } else if (Sg->is_E_or_W && (I >= is - 1) && (I <= ie)) {
        for (int j = MAX2I(js - 1, Sg->jsd); j <= MIN2I(je + 1, Sg->jed); j++) {
          if (Sg->direction == MOM6HIP_OBC_DIRECTION_E) D_u[ORC_U2(G, I, j)] = G->bathyT[ORC_H2(G, I, j)] + CS->Z_ref;
          if (Sg->direction == MOM6HIP_OBC_DIRECTION_W) D_u[ORC_U2(G, I, j)] = G->bathyT[ORC_H2(G, I + 1, j)] + CS->Z_ref;
        }
      }
    }
    if (OBC) for (int n = 0; n < OBC->number_of_segments; n++) {      /* :390-413: then across the corner points of the segments */
      const mom6hip_obc_segment_t *Sg = &OBC->segment[n];
      if (!Sg->on_pe) continue;
      const int I0 = Sg->IsdB, J0 = Sg->JsdB;
      if (Sg->is_N_or_S && (J0 >= js - 1) && (J0 <= je)) {
        const int j = J0;
        for (int I = MAX2I(is - 1, Sg->IsdB); I <= MIN2I(ie, Sg->IedB); I++) {
          if (Sg->direction == MOM6HIP_OBC_DIRECTION_N) { D_u[ORC_U2(G, I, j + 1)] = D_u[ORC_U2(G, I, j)]; mask_u[ORC_U2(G, I, j + 1)] = 0.0; }
          else if (Sg->direction == MOM6HIP_OBC_DIRECTION_S) { D_u[ORC_U2(G, I, j)] = D_u[ORC_U2(G, I, j + 1)]; mask_u[ORC_U2(G, I, j)] = 0.0; }
        }
      } else if (Sg->is_E_or_W && (I0 >= is - 1) && (I0 <= ie)) {
        const int i = I0;
        for (int J = MAX2I(js - 1, Sg->JsdB); J <= MIN2I(je, Sg->JedB); J++) {
          if (Sg->direction == MOM6HIP_OBC_DIRECTION_E) { D_v[ORC_V2(G, i + 1, J)] = D_v[ORC_V2(G, i, J)]; mask_v[ORC_V2(G, i + 1, J)] = 0.0; }
          else if (Sg->direction == MOM6HIP_OBC_DIRECTION_W) { D_v[ORC_V2(G, i, J)] = D_v[ORC_V2(G, i + 1, J)]; mask_v[ORC_V2(G, i, J)] = 0.0; }
        }
      }
    }
#undef MAX2I
#undef MIN2I
  }
  if (Ray_u) memset(Ray_u, 0, sizeof(double) * upl * nz);      /* :416-417 */
  if (Ray_v) memset(Ray_v, 0, sizeof(double) * vpl * nz);

  ORC_PAR
  for (int j = Jsq; j <= Jeq; j++) for (int m = 1; m <= 2; m++) {
    if (m == 1 && j < G->jsc) continue;
    const int is = (m == 1) ? Isq : G->isc, ie = (m == 1) ? Ieq : G->iec;
    double h_at_vel[nz + 1], dz_at_vel[nz + 1], h_vel[nz + 1], dz_vel[nz + 1], T_vel[nz + 1], S_vel[nz + 1];
    double vol_below[nz + 2], L[nz + 2];
    for (int i = is; i <= ie; i++) {
      const int I = i, J = j;
      const int ip = (m == 1) ? i + 1 : i, jp = (m == 1) ? j : j + 1;      /* the cell on the other side of the face */
      const int do_i = (m == 1) ? (G->mask2dCu[ORC_U2(G, I, j)] > 0.0) : (G->mask2dCv[ORC_V2(G, i, J)] > 0.0);
      if (!do_i) continue;      /* (every later block of the reference is inside `if (do_i(i))` for the outputs) */
      const double *vel = (m == 1) ? u : v;
#define VEL(k) vel[(m == 1) ? U3(I, j, k) : V3(i, J, k)]
      for (int k = 1; k <= nz; k++) {      /* :431-470 (dz = GV%H_to_Z*h: Boussinesq thickness_to_dz) */
        const double h0 = h[H3(i, j, k)], h1 = h[H3(ip, jp, k)];
        const double d0 = G->H_to_Z * h0, d1 = G->H_to_Z * h1;
        if (VEL(k) * (h1 - h0) >= 0) {
          h_at_vel[k] = 2.0 * h0 * h1 / (h0 + h1 + h_neglect);
          dz_at_vel[k] = 2.0 * d0 * d1 / (d0 + d1 + dz_neglect);
        } else {
          h_at_vel[k] = 0.5 * (h0 + h1);
          dz_at_vel[k] = 0.5 * (d0 + d1);
        }
        h_vel[k] = 0.5 * (h0 + h1);
        dz_vel[k] = 0.5 * (d0 + d1);
        if (use_BBL_EOS) { T_vel[k] = 0.5 * (T[H3(i, j, k)] + T[H3(ip, jp, k)]); S_vel[k] = 0.5 * (S[H3(i, j, k)] + S[H3(ip, jp, k)]); }
      }
      {      /* :502-580: a zero-gradient projection of the thicknesses, T and S across the faces of the segments */
        const int dir = (m == 1) ? seg_dir(OBC, OBC ? OBC->segnum_u : NULL, ORC_U2(G, I, j)) : seg_dir(OBC, OBC ? OBC->segnum_v : NULL, ORC_V2(G, i, J));
        const int first = (dir == MOM6HIP_OBC_DIRECTION_E || dir == MOM6HIP_OBC_DIRECTION_N);
        if (first || dir == MOM6HIP_OBC_DIRECTION_W || dir == MOM6HIP_OBC_DIRECTION_S) {
          const int ic = first ? i : ip, jc = first ? j : jp;
          if ((m == 1) == (dir == MOM6HIP_OBC_DIRECTION_E || dir == MOM6HIP_OBC_DIRECTION_W)) for (int k = 1; k <= nz; k++) {
            h_at_vel[k] = h[H3(ic, jc, k)]; h_vel[k] = h[H3(ic, jc, k)];
            dz_at_vel[k] = G->H_to_Z * h[H3(ic, jc, k)]; dz_vel[k] = G->H_to_Z * h[H3(ic, jc, k)];
            if (use_BBL_EOS) { T_vel[k] = T[H3(ic, jc, k)]; S_vel[k] = S[H3(ic, jc, k)]; }
          }
        }
      }
      /* the near-bottom velocity magnitude and ustar :565-660 */
      double ustar, umag_avg = 0.0, h_bbl_drag = 0.0, dz_bbl_drag = 0.0, T_EOS = 0.0, S_EOS = 0.0;
      if (use_BBL_EOS || CS->body_force_drag || !CS->linear_drag) {
        double htot_vel = 0.0, hwtot = 0.0, hutot = 0.0, dztot_vel = 0.0, dzwtot = 0.0, Thtot = 0.0, Shtot = 0.0;
        const double u2_bg = CS->drag_bg_vel * CS->drag_bg_vel;
        for (int k = nz; k >= 1; k--) {
          if (htot_vel >= CS->Hbbl) break;
          const double hweight = min2(CS->Hbbl - htot_vel, h_at_vel[k]);
          if (hweight < 1.5 * G->Angstrom_H + h_neglect) continue;
          const double dzweight = min2(CS->dz_bbl - dztot_vel, dz_at_vel[k]);
          htot_vel = htot_vel + h_at_vel[k];
          hwtot = hwtot + hweight;
          dztot_vel = dztot_vel + dz_at_vel[k];
          dzwtot = dzwtot + dzweight;
          if ((!CS->linear_drag) && (hweight >= 0.0)) {
            if (m == 1) {
              const double v_at_u = set_v_at_u(G, v, h, i, j, k, mask_v, OBC);
              hutot = hutot + hweight * sqrt(u[U3(I, j, k)] * u[U3(I, j, k)] + v_at_u * v_at_u + u2_bg);
            } else {
              const double u_at_v = set_u_at_v(G, u, h, i, j, k, mask_u, OBC);
              hutot = hutot + hweight * sqrt(v[V3(i, J, k)] * v[V3(i, J, k)] + u_at_v * u_at_v + u2_bg);
            }
          }
          if (use_BBL_EOS && (hweight >= 0.0)) {
            Thtot = Thtot + hweight * T_vel[k];
            Shtot = Shtot + hweight * S_vel[k];
          }
        }
        double I_hwtot = 0.0; if (hwtot > 0.0) I_hwtot = 1.0 / hwtot;
        if ((hwtot <= 0.0) || CS->linear_drag) ustar = cdrag_sqrt_H * CS->drag_bg_vel;
        else ustar = cdrag_sqrt_H * hutot / hwtot;
        umag_avg = hutot * I_hwtot;
        h_bbl_drag = hwtot;
        dz_bbl_drag = dzwtot;
        if (use_BBL_EOS) {
          if (hwtot > 0.0) { T_EOS = Thtot / hwtot; S_EOS = Shtot / hwtot; }
          else { T_EOS = 0.0; S_EOS = 0.0; }
        }
      } else {
        ustar = cdrag_sqrt_H * CS->drag_bg_vel;
      }
      double dR_dT = 0.0, dR_dS = 0.0;
      if (use_BBL_EOS) {      /* :662-680 */
        double press = 0.0;
        for (int k = 1; k <= nz; k++) press = press + (CS->H_to_RZ * G->g_Earth) * h_vel[k];
        orc_eos_density_derivs(EOS, T_EOS, S_EOS, press, &dR_dT, &dR_dS);
      }
      /* the thickness of the bottom boundary layer :682-790 */
      const double ustarsq = Rho0x400_G * (ustar * ustar);
      double htot = 0.0, dztot = 0.0;
      if (use_BBL_EOS) {
        double Thtot = 0.0, Shtot = 0.0, oldfn = 0.0;
        for (int k = nz; k >= 2; k--) {
          if (h_at_vel[k] <= 0.0) continue;
          oldfn = dR_dT * (Thtot - T_vel[k] * htot) + dR_dS * (Shtot - S_vel[k] * htot);
          if (oldfn >= ustarsq) break;
          const double Dfn = (dR_dT * (T_vel[k] - T_vel[k - 1]) + dR_dS * (S_vel[k] - S_vel[k - 1])) * (h_at_vel[k] + htot);
          double Dh, Ddz;
          if ((oldfn + Dfn) <= ustarsq) {
            Dh = h_at_vel[k];
            Ddz = dz_at_vel[k];
          } else {
            const double frac_used = sqrt((ustarsq - oldfn) / (Dfn));
            Dh = h_at_vel[k] * frac_used;
            Ddz = dz_at_vel[k] * frac_used;
          }
          htot = htot + Dh;
          dztot = dztot + Ddz;
          Thtot = Thtot + T_vel[k] * Dh; Shtot = Shtot + S_vel[k] * Dh;
        }
        if ((oldfn < ustarsq) && h_at_vel[1] > 0.0) {
          if (dR_dT * (Thtot - T_vel[1] * htot) + dR_dS * (Shtot - S_vel[1] * htot) < ustarsq) {
            htot = htot + h_at_vel[1];
            dztot = dztot + dz_at_vel[1];
          }
        }
      } else {
        double Rhtot = 0.0;
        for (int k = nz; k >= K2; k--) {
          const double oldfn = Rhtot - CS->Rlay[k - 1] * htot;
          const double Dfn = (CS->Rlay[k - 1] - CS->Rlay[k - 2]) * (h_at_vel[k] + htot);
          double Dh, Ddz;
          if (oldfn >= ustarsq) {
            continue;
          } else if ((oldfn + Dfn) <= ustarsq) {
            Dh = h_at_vel[k];
            Ddz = dz_at_vel[k];
          } else {
            const double frac_used = sqrt((ustarsq - oldfn) / (Dfn));
            Dh = h_at_vel[k] * frac_used;
            Ddz = dz_at_vel[k] * frac_used;
          }
          htot = htot + Dh;
          dztot = dztot + Ddz;
          Rhtot = Rhtot + CS->Rlay[k - 1] * Dh;
        }
        if (Rhtot - CS->Rlay[0] * htot < ustarsq) {
          htot = htot + h_at_vel[1];
          dztot = dztot + dz_at_vel[1];
        }
      }
      /* :792-830 */
      double C2f;
      if (m == 1) C2f = G->CoriolisBu[ORC_Q2(G, I, J - 1)] + G->CoriolisBu[ORC_Q2(G, I, J)];
      else C2f = G->CoriolisBu[ORC_Q2(G, I - 1, J)] + G->CoriolisBu[ORC_Q2(G, I, J)];
      const double u2_bg = CS->drag_bg_vel * CS->drag_bg_vel;
      double bbl_thick;
      if (CS->cdrag * u2_bg <= 0.0) {
        const double ustH = ustar, root = sqrt(0.25 * (ustH * ustH) + (htot * C2f) * (htot * C2f));
        if (dztot * ustH <= (CS->BBL_thick_min + dz_neglect) * (0.5 * ustH + root)) bbl_thick = CS->BBL_thick_min;
        else bbl_thick = (dztot * ustH) / (0.5 * ustH + root);
      } else {
        bbl_thick = dztot / (0.5 + sqrt(0.25 + htot * htot * C2f * C2f / (ustar * ustar)));
        if (bbl_thick < CS->BBL_thick_min) bbl_thick = CS->BBL_thick_min;
      }
      const double bbl_thick_before_caps = bbl_thick;
      if ((bbl_thick > 0.5 * CS->dz_bbl) && (CS->RiNo_mix)) bbl_thick = 0.5 * CS->dz_bbl;
      if (CS->body_force_drag) bbl_thick = dz_bbl_drag;
      double Vol_bbl_chan = bbl_thick_before_caps;      /* :849 (stored before the RiNo_mix and body-force overrides) */
      double kv_bbl;
      if (CS->Channel_drag) {      /* :863-1002 */
        vol_below[nz + 1] = 0.0;
        for (int K = nz; K >= 1; K--) vol_below[K] = vol_below[K + 1] + dz_vel[K];
        double D_vel, Dp, Dm, tmp;
#define D_U(I_, j_) D_u[ORC_U2(G, I_, j_)]
#define D_V(i_, J_) D_v[ORC_V2(G, i_, J_)]
        if (m == 1) {
          D_vel = D_U(I, j);
          tmp = mask_u[ORC_U2(G, I, j + 1)] * D_U(I, j + 1);
          Dp = 2.0 * D_vel * tmp / (D_vel + tmp);
          tmp = mask_u[ORC_U2(G, I, j - 1)] * D_U(I, j - 1);
          Dm = 2.0 * D_vel * tmp / (D_vel + tmp);
        } else {
          D_vel = D_V(i, J);
          tmp = mask_v[ORC_V2(G, i + 1, J)] * D_V(i + 1, J);
          Dp = 2.0 * D_vel * tmp / (D_vel + tmp);
          tmp = mask_v[ORC_V2(G, i - 1, J)] * D_V(i - 1, J);
          Dm = 2.0 * D_vel * tmp / (D_vel + tmp);
        }
#undef D_U
#undef D_V
        if (Dm > Dp) { tmp = Dp; Dp = Dm; Dm = tmp; }
        double crv = 3.0 * (Dp + Dm - 2.0 * D_vel);
        const double slope = Dp - Dm;
        if (fabs(crv) < 1e-2 * (slope + CS->BBL_thick_min)) crv = 0.0;
        if (crv == 0.0) find_L_open_uniform_slope(nz, vol_below, Dp, Dm, L);
        else if (crv > 0.0) {
          if (CS->concave_trigonometric_L) find_L_open_concave_trigonometric(nz, vol_below, D_vel, Dp, Dm, L);
          else find_L_open_concave_iterative(nz, vol_below, D_vel, Dp, Dm, L);
        } else find_L_open_convex(nz, vol_below, D_vel, Dp, Dm, L, G->Angstrom_H * G->H_to_Z, dz_neglect);
        if (CS->Chan_drag_max_vol >= 0.0) Vol_bbl_chan = min2(Vol_bbl_chan, CS->Chan_drag_max_vol);
        double BBL_visc_frac = 0.0;
        for (int K = nz; K >= 1; K--) {      /* (no porous barriers: por_layer_width = por_face_area = 1) */
          double Rayleigh;
          if (L[K] > L[K + 1]) {
            double BBL_frac;
            if (vol_below[K + 1] < Vol_bbl_chan) {
              const double q = (1.0 - vol_below[K + 1] / Vol_bbl_chan);
              BBL_frac = q * q;
              BBL_visc_frac = BBL_visc_frac + BBL_frac * (L[K] - L[K + 1]);
            } else {
              BBL_frac = 0.0;
            }
            const double cdrag_conv = cdrag_L_to_H;
            const double h_vel_pos = h_vel[K] + h_neglect;
            const double Cell_width = (m == 1) ? G->dy_Cu[ORC_U2(G, I, j)] : G->dx_Cv[ORC_V2(G, i, J)];
            const double gam = 1.0 - L[K + 1] / L[K];
            Rayleigh = cdrag_conv * (L[K] - L[K + 1]) * (1.0 - BBL_frac) *
                       (12.0 * CS->c_Smag * h_vel_pos) / (12.0 * CS->c_Smag * h_vel_pos +
                                                          cdrag_conv * gam * (1.0 - gam) * (1.0 - 1.5 * gam) * (L[K] * L[K]) * Cell_width);
          } else {
            Rayleigh = 0.0;
          }
          if (m == 1) {
            if (Rayleigh > 0.0) {
              const double v_at_u = set_v_at_u(G, v, h, i, j, K, mask_v, OBC);
              Ray_u[U3(I, j, K)] = Rayleigh * sqrt(u[U3(I, j, K)] * u[U3(I, j, K)] + v_at_u * v_at_u + u2_bg);
            } else Ray_u[U3(I, j, K)] = 0.0;
          } else {
            if (Rayleigh > 0.0) {
              const double u_at_v = set_u_at_v(G, u, h, i, j, K, mask_u, OBC);
              Ray_v[V3(i, J, K)] = Rayleigh * sqrt(v[V3(i, J, K)] * v[V3(i, J, K)] + u_at_v * u_at_v + u2_bg);
            } else Ray_v[V3(i, J, K)] = 0.0;
          }
        }
        if (CS->correct_BBL_bounds && cdrag_sqrt * ustar * bbl_thick * BBL_visc_frac <= CS->Kv_BBL_min) {
          kv_bbl = CS->Kv_BBL_min;
          if ((cdrag_sqrt * ustar) * BBL_visc_frac * BBL_thick_max > kv_bbl) bbl_thick = kv_bbl / ((cdrag_sqrt * ustar) * BBL_visc_frac);
          else bbl_thick = BBL_thick_max;
        } else {
          kv_bbl = (cdrag_sqrt * ustar) * bbl_thick * BBL_visc_frac;
        }
      } else
      /* not channel drag :1004-1022 */
      if (CS->correct_BBL_bounds && cdrag_sqrt * ustar * bbl_thick <= CS->Kv_BBL_min) {
        kv_bbl = CS->Kv_BBL_min;
        if ((cdrag_sqrt * ustar) * BBL_thick_max > kv_bbl) bbl_thick = kv_bbl / (cdrag_sqrt * ustar);
        else bbl_thick = BBL_thick_max;
      } else {
        kv_bbl = (cdrag_sqrt * ustar) * bbl_thick;
      }
      if (CS->body_force_drag) {      /* :1024-1046 */
        if (h_bbl_drag > 0.0) {
          double h_sum = 0.0;
          const double I_hwtot = 1.0 / h_bbl_drag;
          for (int k = nz; k >= 1; k--) {
            const double h_bbl_fr = min2(h_bbl_drag - h_sum, h_at_vel[k]) * I_hwtot;
            const double cdrag_conv = cdrag_L_to_H;
            if (m == 1) Ray_u[U3(I, j, k)] = Ray_u[U3(I, j, k)] + (cdrag_conv * umag_avg) * h_bbl_fr;
            else Ray_v[V3(i, J, k)] = Ray_v[V3(i, J, k)] + (cdrag_conv * umag_avg) * h_bbl_fr;
            h_sum = h_sum + h_at_vel[k];
            if (h_sum >= h_bbl_drag) break;
          }
          kv_bbl = CS->Kv_BBL_min;
        }
      }
      kv_bbl = max2(CS->Kv_BBL_min, kv_bbl);
      if (m == 1) {
        bbl_thick_u[ORC_U2(G, I, j)] = bbl_thick;
        if (Kv_bbl_u) Kv_bbl_u[ORC_U2(G, I, j)] = kv_bbl;
      } else {
        bbl_thick_v[ORC_V2(G, i, J)] = bbl_thick;
        if (Kv_bbl_v) Kv_bbl_v[ORC_V2(G, i, J)] = kv_bbl;
      }
#undef VEL
    }
  }
  free(D_u);
  return 0;
}
