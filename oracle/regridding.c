/*
 * regridding.c -- CPU restatement of the z* regridding and of the velocity remapping of MOM_ALE
 * (TEST INFRASTRUCTURE, see mom6_oracle.h).
 *   build_zstar_column        src/ALE/coord_zlike.F90:63-144   (no rigid top)
 *   filtered_grid_motion      src/ALE/MOM_regridding.F90:1022-1171
 *   build_zstar_grid          :1174-1284
 *   adjust_interface_motion   :1713-1774
 *   calc_h_new_by_dz          :925-958
 *   regridding_main           :763-889 (REGRIDDING_ZSTAR, Boussinesq)
 *   ALE_regrid                src/ALE/MOM_ALE.F90:484-520
 *   ALE_remap_set_h_vel       :870-908
 *   ALE_remap_set_h_vel_via_dz :912-960
 *   ALE_remap_velocities      :1061-1274
 * PARITY UNPINNED for the regridding (the reference's unit tests cover the remapping kernels, which the velocity
 * remap reuses); checked through invariants in tests/test_regridding.py.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mom6_oracle.h"

static inline double max2(double a, double b) { return a > b ? a : b; }
static inline double min2(double a, double b) { return a < b ? a : b; }
static inline double max3(double a, double b, double c) { return max2(max2(a, b), c); }

/* build_zstar_column (zInterface is 1-based: z[1..nk+1]) */
static void build_zstar_column(const mom6hip_regridding_cs_t *CS, double depth, double total_thickness, double *z, double z_scale) {
  const int nk = CS->nk;
  const double min_thickness = min2(CS->min_thickness, total_thickness / (double)nk);
  const double z0_top = 0.;
  const double eta = total_thickness - depth;
  const double stretching = total_thickness / (depth + z0_top);
  z[1] = eta;
  for (int k = 1; k <= nk; k++) {
    double dh = stretching * CS->coordinateResolution[k - 1] * z_scale;
    z[k + 1] = z[k] - dh;
  }
  z[nk + 1] = -depth;
  for (int k = nk; k >= 1; k--)
    if (z[k] < (z[k + 1] + min_thickness)) z[k] = z[k + 1] + min_thickness;
}

/* filtered_grid_motion (arrays 1-based) */
static int filtered_grid_motion(const mom6hip_regridding_cs_t *CS, int nk, const double *z_old, const double *z_new, double *dz_g) {
  const int cnk = CS->nk;
  double sgn;
  const double prod = (z_old[nk + 1] - z_old[1]) * (z_new[cnk + 1] - z_new[1]);
  if (prod < 0.0) return 1;
  else if (prod == 0.0) { for (int k = 1; k <= cnk + 1; k++) dz_g[k] = 0.0; return 0; }
  else if ((z_old[nk + 1] - z_old[1]) + (z_new[cnk + 1] - z_new[1]) > 0.0) sgn = 1.0;
  else sgn = -1.0;
  const double zs = CS->depth_of_time_filter_shallow, zd = CS->depth_of_time_filter_deep;
  const double wtd = 1.0 - CS->old_grid_weight, Iwtd = 1.0 / wtd;
  const double dzwt = (zd - zs);
  double Idzwt = 0.0; if (fabs(zd - zs) > 0.0) Idzwt = 1.0 / (zd - zs);
  const double dInt_zs_zd = 0.5 * (1.0 + Iwtd) * (zd - zs);
  const double Aq = 0.5 * (Iwtd - 1.0);
  dz_g[1] = 0.0;
  double z_old_k = z_old[1];
  for (int k = 2; k <= cnk + 1; k++) {
    if (k <= nk + 1) z_old_k = z_old[k];
    const double dz_tgt = sgn * (z_new[k] - z_old_k);
    const double zr1 = sgn * (z_old_k - z_old[1]);
    if ((zr1 > zd) && (zr1 + wtd * dz_tgt > zd)) {
      dz_g[k] = sgn * wtd * dz_tgt;
    } else if ((zr1 < zs) && (zr1 + dz_tgt < zs)) {
      dz_g[k] = sgn * dz_tgt;
    } else {
      double Int_zd, Int_zs;
      if (zr1 >= zd) { Int_zd = Iwtd * (zd - zr1); Int_zs = Int_zd - dInt_zs_zd; }
      else if (zr1 <= zs) { Int_zs = (zs - zr1); Int_zd = dInt_zs_zd + (zs - zr1); }
      else {
        Int_zd = (zd - zr1) * (Iwtd * (0.5 * (zd + zr1) - zs) + 0.5 * (zd - zr1)) * Idzwt;
        Int_zs = (zs - zr1) * (0.5 * Iwtd * ((zr1 - zs)) + (zd - 0.5 * (zr1 + zs))) * Idzwt;
      }
      if (dz_tgt >= Int_zd) dz_g[k] = sgn * ((zd - zr1) + wtd * (dz_tgt - Int_zd));
      else if (dz_tgt <= Int_zs) dz_g[k] = sgn * ((zs - zr1) + (dz_tgt - Int_zs));
      else {
        double dz0, z0, F0;
        if (zr1 <= zs) { dz0 = zs - zr1; z0 = zs; F0 = dz_tgt - Int_zs; }
        else if (zr1 >= zd) { dz0 = zd - zr1; z0 = zd; F0 = dz_tgt - Int_zd; }
        else { dz0 = 0.0; z0 = zr1; F0 = dz_tgt; }
        const double Bq = (dzwt + 2.0 * Aq * (z0 - zs));
        dz_g[k] = sgn * (dz0 + 2.0 * F0 * dzwt / (Bq + sqrt(Bq * Bq + 4.0 * Aq * F0 * dzwt)));
      }
    }
  }
  return 0;
}

/* adjust_interface_motion (arrays 1-based); returns nonzero where the reference raises FATAL */
static int adjust_interface_motion(const mom6hip_regridding_cs_t *CS, int nk, const double *h_old, double *dz_int) {
  const double eps = DBL_EPSILON;
  const int n = CS->nk < nk ? CS->nk : nk;
  double h_total = 0., h_err = 0.;
  for (int k = 1; k <= n; k++) {
    h_total = h_total + h_old[k];
    h_err = h_err + max3(h_old[k], fabs(dz_int[k]), fabs(dz_int[k + 1])) * eps;
    double h_new = h_old[k] + (dz_int[k] - dz_int[k + 1]);
    if (h_new < -3.0 * h_err) return 2;
  }
  (void)h_total;
  for (int k = n; k >= 2; k--) {
    double h_new = h_old[k] + (dz_int[k] - dz_int[k + 1]);
    if (h_new < CS->min_thickness) dz_int[k] = (dz_int[k + 1] - h_old[k]) + CS->min_thickness;
    h_new = h_old[k] + (dz_int[k] - dz_int[k + 1]);
    if (h_new < 0.) dz_int[k] = (1. - eps) * (dz_int[k + 1] - h_old[k]);
    h_new = h_old[k] + (dz_int[k] - dz_int[k + 1]);
    if (h_new < 0.) return 3;
  }
  return 0;
}

/* ALE_regrid -> regridding_main (z*) */
int orc_ale_regrid(const mom6hip_grid_t *G, const mom6hip_regridding_cs_t *CS, const double *h, double *h_new, double *dzRegrid) {
  const int nz = G->nk;
  if (CS->regridding_scheme != MOM6HIP_REGRIDDING_ZSTAR || CS->nk != nz) return 1;
  const long nh2 = (long)ORC_NIH(G) * ORC_NJH(G);
  const double Z_to_H = G->Z_to_H;
  memset(dzRegrid, 0, sizeof(double) * nh2 * (nz + 1));                         /* MOM_ALE.F90:508 */
  int rc = 0;
  ORC_PAR      /* the columns are independent */
  for (int j = G->jsc - 1; j <= G->jec + 1; j++) for (int i = G->isc - 1; i <= G->iec + 1; i++) {
    double zOld[nz + 2], zNew[nz + 2], dz[nz + 2], hc[nz + 2];
    const long n2 = ORC_H2(G, i, j);
#define DZ(k) dzRegrid[n2 + nh2 * ((k) - 1)]
    if (G->mask2dT[n2] == 0.) {
      for (int k = 1; k <= nz + 1; k++) DZ(k) = 0.;
    } else {
      const double nominalDepth = max2((G->bathyT[n2] + CS->Z_ref) * Z_to_H, 0.0);       /* :833 */
      double totalThickness = 0.0;
      for (int k = 1; k <= nz; k++) { hc[k] = h[ORC_H3(G, i, j, k)]; totalThickness = totalThickness + hc[k]; }
      zOld[nz + 1] = -nominalDepth;
      for (int k = nz; k >= 1; k--) zOld[k] = zOld[k + 1] + hc[k];
      build_zstar_column(CS, nominalDepth, totalThickness, zNew, Z_to_H);
      for (int k = 1; k <= nz + 1; k++) dz[k] = DZ(k);
      int rc1 = filtered_grid_motion(CS, nz, zOld, zNew, dz);
      if (!rc1) rc1 = adjust_interface_motion(CS, nz, hc, dz);
      if (rc1) { rc = rc1; continue; }
      for (int k = 1; k <= nz + 1; k++) DZ(k) = dz[k];
    }
    /* calc_h_new_by_dz :925 */
    if (G->mask2dT[n2] > 0.) {
      for (int k = 1; k <= nz; k++) h_new[ORC_H3(G, i, j, k)] = max2(0., h[ORC_H3(G, i, j, k)] + (DZ(k) - DZ(k + 1)));
    } else {
      for (int k = 1; k <= nz; k++) h_new[ORC_H3(G, i, j, k)] = h[ORC_H3(G, i, j, k)];
    }
#undef DZ
  }
  return rc;
}

/* ALE_remap_set_h_vel :870 */
int orc_ale_remap_set_h_vel(const mom6hip_grid_t *G, const double *h_new, double *h_u, double *h_v) {
  ORC_PAR
  for (int k = 1; k <= G->nk; k++) for (int j = G->jsc; j <= G->jec; j++) for (int I = G->isc - 1; I <= G->iec; I++)
    if (G->mask2dCu[ORC_U2(G, I, j)] > 0.) h_u[ORC_U3(G, I, j, k)] = 0.5 * (h_new[ORC_H3(G, I, j, k)] + h_new[ORC_H3(G, I + 1, j, k)]);
  ORC_PAR
  for (int k = 1; k <= G->nk; k++) for (int J = G->jsc - 1; J <= G->jec; J++) for (int i = G->isc; i <= G->iec; i++)
    if (G->mask2dCv[ORC_V2(G, i, J)] > 0.) h_v[ORC_V3(G, i, J, k)] = 0.5 * (h_new[ORC_H3(G, i, J, k)] + h_new[ORC_H3(G, i, J + 1, k)]);
  return 0;
}

/* ALE_remap_set_h_vel_via_dz :912-960 (REMAP_UV_USING_OLD_ALG = True, MOM.F90:1666-1667): the new velocity-point grid from the
 * old thicknesses and the interface movements dzInterface (nk+1 planes) */
int orc_ale_remap_set_h_vel_via_dz(const mom6hip_grid_t *G, const double *h_old, const double *dzInterface, double *h_u, double *h_v) {
  const long nH2 = (long)ORC_NIH(G) * ORC_NJH(G);
#define DZI(i, j, K) dzInterface[ORC_H2(G, i, j) + nH2 * ((K) - 1)]
  ORC_PAR
  for (int k = 1; k <= G->nk; k++) for (int j = G->jsc; j <= G->jec; j++) for (int I = G->isc - 1; I <= G->iec; I++)
    if (G->mask2dCu[ORC_U2(G, I, j)] > 0.) {
      const int i = I;
      const double v = 0.5 * (h_old[ORC_H3(G, i, j, k)] + h_old[ORC_H3(G, i + 1, j, k)]) +
                       0.5 * ((DZI(i, j, k) + DZI(i + 1, j, k)) - (DZI(i, j, k + 1) + DZI(i + 1, j, k + 1)));
      h_u[ORC_U3(G, I, j, k)] = 0. > v ? 0. : v;     /* max(0., v) */
    }
  ORC_PAR
  for (int k = 1; k <= G->nk; k++) for (int J = G->jsc - 1; J <= G->jec; J++) for (int i = G->isc; i <= G->iec; i++)
    if (G->mask2dCv[ORC_V2(G, i, J)] > 0.) {
      const int j = J;
      const double v = 0.5 * (h_old[ORC_H3(G, i, j, k)] + h_old[ORC_H3(G, i, j + 1, k)]) +
                       0.5 * ((DZI(i, j, k) + DZI(i, j + 1, k)) - (DZI(i, j, k + 1) + DZI(i, j + 1, k + 1)));
      h_v[ORC_V3(G, i, J, k)] = 0. > v ? 0. : v;
    }
#undef DZI
  return 0;
}

/* ALE_remap_velocities :1061 (conserve_ke false, no BBL masking, no diagnostics) */
int orc_ale_remap_velocities(const mom6hip_grid_t *G, const mom6hip_remapping_cs_t *cs, const double *h_old_u, const double *h_old_v,
                             const double *h_new_u, const double *h_new_v, double *u, double *v) {
  const int nz = G->nk;
  const double h_neglect = G->H_subroundoff, h_neglect_edge = G->H_subroundoff;
  int rc = 0;
  ORC_PAR
  for (int j = G->jsc; j <= G->jec; j++) for (int I = G->isc - 1; I <= G->iec; I++) {
    double h1[nz], h2[nz], src[nz], tgt[nz];
    if (!(G->mask2dCu[ORC_U2(G, I, j)] > 0.)) continue;
    for (int k = 1; k <= nz; k++) { h1[k - 1] = h_old_u[ORC_U3(G, I, j, k)]; h2[k - 1] = h_new_u[ORC_U3(G, I, j, k)]; src[k - 1] = u[ORC_U3(G, I, j, k)]; }
    const int rc1 = orc_remapping_core_h(cs->remapping_scheme, cs->boundary_extrapolation, nz, h1, src, nz, h2, tgt, h_neglect, h_neglect_edge);
    if (rc1) { rc = rc1; continue; }
    for (int k = 1; k <= nz; k++) u[ORC_U3(G, I, j, k)] = tgt[k - 1];
  }
  ORC_PAR
  for (int J = G->jsc - 1; J <= G->jec; J++) for (int i = G->isc; i <= G->iec; i++) {
    double h1[nz], h2[nz], src[nz], tgt[nz];
    if (!(G->mask2dCv[ORC_V2(G, i, J)] > 0.)) continue;
    for (int k = 1; k <= nz; k++) { h1[k - 1] = h_old_v[ORC_V3(G, i, J, k)]; h2[k - 1] = h_new_v[ORC_V3(G, i, J, k)]; src[k - 1] = v[ORC_V3(G, i, J, k)]; }
    const int rc1 = orc_remapping_core_h(cs->remapping_scheme, cs->boundary_extrapolation, nz, h1, src, nz, h2, tgt, h_neglect, h_neglect_edge);
    if (rc1) { rc = rc1; continue; }
    for (int k = 1; k <= nz; k++) v[ORC_V3(G, i, J, k)] = tgt[k - 1];
  }
  return rc;
}
