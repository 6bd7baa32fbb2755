/*
 * vert_friction.c -- CPU restatement (TEST INFRASTRUCTURE, see mom6_oracle.h) of
 * src/parameterizations/vertical/MOM_vert_friction.F90: vertvisc_coef (:1168-1763) with find_coupling_coef
 * (:1768-2254), vertvisc (:526-1059) with vertvisc_limit_vel (:2259-2462), vertvisc_remnant (:1064-1162), for the
 * branch libmom6hip provides (include/mom6hip.h, "MOM_vert_friction").  PARITY UNPINNED: the reference holds no
 * known-answer vectors for this module; tests/test_vert_friction.py checks it through what the scheme guarantees
 * (momentum budget of the implicit solve, bounds of visc_rem, the bottom-stress limit).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mom6_oracle.h"

static inline double max2(double a, double b) { return a > b ? a : b; }
static inline double min2(double a, double b) { return a < b ? a : b; }

static int unsupported(const mom6hip_vertvisc_cs_t *CS) {
  for (int n = 0; n < 7; n++) if (CS->unsupported[n]) return 1;
  return CS->answer_date < 20190101;
}
static int surface_bl(const mom6hip_vertvisc_cs_t *CS) { return CS->dynamic_viscous_ML || CS->nkml > 0; }

/* one face column of vertvisc_coef + find_coupling_coef.  c0 / c1: 2-D offsets of the two cells; f2: of the face;
 * hpl / fpl: plane strides of h-point and face arrays.  side: 0, or for a face of an open-boundary segment the cell the
 * thicknesses, the depth, Kv_shear and ustar are projected outward from (:1335-1355 / :1546-1566, :1901-1925, :2061-2110):
 * -1 the first cell (OBC_DIRECTION_E | N), +1 the second (OBC_DIRECTION_W | S). */
static void coef_column(const mom6hip_grid_t *G, const mom6hip_vertvisc_cs_t *CS, const mom6hip_vertvisc_type_t *visc,
                        const double *vel, const double *h, const double *dz, long c0, long c1, long f2, long hpl, long fpl,
                        const double *kv_bbl_2d, const double *bbl_thick_2d, double *a_out, double *h_out, long qA, long qB,
                        const double *nkml_visc_2d, int side) {
  const int nz = G->nk;
  const double h_neglect = G->H_subroundoff, dz_neglect = G->dZ_subroundoff;
  const double a_cpl_max = 1.0e37 * G->Z_to_H * 1.0;      /* 1.0e37 * GV%m_to_H * US%T_to_s :1283 */
  double I_Hbbl = 1.0 / (CS->Hbbl + dz_neglect);          /* :1284 */
  const double I_valBL = (CS->harm_BL_val > 0.0) ? 1.0 / CS->harm_BL_val : 0.0;   /* :1288 */
  double kv_bbl = 0.0, bbl_thick = 0.0;
  if (CS->bottomdraglaw) {      /* :1318-1322 */
    kv_bbl = kv_bbl_2d[f2];
    bbl_thick = bbl_thick_2d[f2] + dz_neglect;
    I_Hbbl = 1.0 / bbl_thick;
  }
  double *w = (double *)malloc(sizeof(double) * (size_t)(7 * nz + 3 * (nz + 1)));
  double *h_harm = w, *h_arith = w + nz, *h_delta = w + 2 * nz, *dz_harm = w + 3 * nz, *dz_arith = w + 4 * nz, *hvel = w + 5 * nz,
         *dz_vel = w + 6 * nz, *z_i = w + 7 * nz, *a_cpl = z_i + (nz + 1), *Kv_tot = a_cpl + (nz + 1);
#define DZ(c, k) (dz ? dz[(c) + hpl * (k)] : G->H_to_Z * h[(c) + hpl * (k)])      /* thickness_to_dz, Boussinesq */
  for (int k = 0; k < nz; k++) {      /* :1324-1330 */
    const double h0 = h[c0 + hpl * k], h1 = h[c1 + hpl * k];
    h_harm[k] = 2.0 * h0 * h1 / (h0 + h1 + h_neglect);
    h_arith[k] = 0.5 * (h1 + h0);
    h_delta[k] = h1 - h0;
    const double d0 = DZ(c0, k), d1 = DZ(c1, k);
    dz_harm[k] = 2.0 * d0 * d1 / (d0 + d1 + dz_neglect);
    dz_arith[k] = 0.5 * (d1 + d0);
    if (side) {      /* :1338-1341 / :1345-1348: a zero-gradient condition across the open boundary */
      const double hs = (side < 0) ? h0 : h1, ds = (side < 0) ? d0 : d1;
      h_harm[k] = hs; h_arith[k] = hs; h_delta[k] = 0.;
      dz_harm[k] = ds; dz_arith[k] = ds;
    }
  }
  double Dmin = min2(G->bathyT[c0], G->bathyT[c1]);      /* :1331 */
  if (side) Dmin = (side < 0) ? G->bathyT[c0] : G->bathyT[c1];      /* :1342, :1349; zi_dir = side */
  if (CS->harmonic_visc) {      /* :1363-1375 */
    z_i[nz] = 0.0;
    for (int k = nz - 1; k >= 0; k--) {
      hvel[k] = h_harm[k];
      dz_vel[k] = dz_harm[k];
      if (vel[f2 + fpl * k] * h_delta[k] < 0) {
        const double z2 = z_i[k + 1], botfn = 1.0 / (1.0 + 0.09 * z2 * z2 * z2 * z2 * z2 * z2);
        hvel[k] = (1.0 - botfn) * h_harm[k] + botfn * h_arith[k];
        dz_vel[k] = (1.0 - botfn) * dz_harm[k] + botfn * dz_arith[k];
      }
      z_i[k] = z_i[k + 1] + dz_harm[k] * I_Hbbl;
    }
  } else {      /* :1376-1408 */
    double zh = 0.0, zcol0 = -G->bathyT[c0], zcol1 = -G->bathyT[c1];
    z_i[nz] = 0.0;
    for (int k = nz - 1; k >= 0; k--) {
      zcol0 = zcol0 + DZ(c0, k); zcol1 = zcol1 + DZ(c1, k);
      zh = zh + dz_harm[k];
      double z_clear = max2(zcol0, zcol1) + Dmin;
      if (side < 0) z_clear = zcol0 + Dmin;      /* :1381-1382 */
      if (side > 0) z_clear = zcol1 + Dmin;
      z_i[k] = max2(zh, z_clear) * I_Hbbl;
      hvel[k] = h_arith[k];
      dz_vel[k] = dz_arith[k];
      if (vel[f2 + fpl * k] * h_delta[k] > 0) {
        if (zh * I_Hbbl < CS->harm_BL_val) {
          hvel[k] = h_harm[k];
          dz_vel[k] = dz_harm[k];
        } else {
          double z2_wt = 1.0;
          if (zh * I_Hbbl < 2.0 * CS->harm_BL_val) z2_wt = max2(0.0, min2(1.0, zh * I_Hbbl * I_valBL - 1.0));
          const double z2 = z2_wt * (max2(zh, z_clear) * I_Hbbl);
          const double botfn = 1.0 / (1.0 + 0.09 * z2 * z2 * z2 * z2 * z2 * z2);
          hvel[k] = (1.0 - botfn) * h_arith[k] + botfn * h_harm[k];
          dz_vel[k] = (1.0 - botfn) * dz_arith[k] + botfn * dz_harm[k];
        }
      }
    }
  }

  /* ---- find_coupling_coef(a_cpl, dz_vel, do_i, dz_harm, bbl_thick, kv_bbl, z_i, ...) :1768; its hvel is dz_vel, its
   * h_harm is dz_harm, its h_neglect is GV%dZ_subroundoff ---- */
  {
    const double hn = dz_neglect, I_amax = 0.0;      /* :1846, :1858 (answer_date >= 20190101) */
    for (int K = 0; K <= nz; K++) { a_cpl[K] = 0.0; Kv_tot[K] = 0.0; }
    Kv_tot[0] = 0.0;                                  /* :1868 */
    for (int K = 1; K <= nz; K++) Kv_tot[K] = CS->Kv; /* :1869-1871 */
    if (CS->Kvml_invZ2 > 0.0) {                       /* :1873-1886 */
      const double I_Hmix = 1.0 / (CS->Hmix + hn);
      double z_t = hn * I_Hmix;
      for (int K = 1; K < nz; K++) {
        z_t = z_t + dz_harm[K - 1] * I_Hmix;
        Kv_tot[K] = CS->Kv + CS->Kvml_invZ2 / ((z_t * z_t) * (1.0 + 0.09 * z_t * z_t * z_t * z_t * z_t * z_t));
      }
    }
    if (visc->Kv_shear) {                             /* :1888-1928 */
      for (int K = 1; K < nz; K++) {
        double Kv_add = 0.5 * (visc->Kv_shear[c0 + hpl * K] + visc->Kv_shear[c1 + hpl * K]);
        if (side) Kv_add = visc->Kv_shear[((side < 0) ? c0 : c1) + hpl * K];      /* :1901-1909, :1917-1925 */
        Kv_tot[K] = Kv_tot[K] + Kv_add;
      }
    }
    if (CS->bottomdraglaw) {                          /* :1948-1976 */
      double dhc = dz_vel[nz - 1] * 0.5;
      if (dhc < bbl_thick) a_cpl[nz] = kv_bbl / ((dhc + hn) + I_amax * kv_bbl);
      else a_cpl[nz] = kv_bbl / ((bbl_thick + hn) + I_amax * kv_bbl);
      for (int K = nz - 1; K >= 1; K--) {
        const double z2 = z_i[K], botfn = 1.0 / (1.0 + 0.09 * z2 * z2 * z2 * z2 * z2 * z2);
        Kv_tot[K] = Kv_tot[K] + (kv_bbl - CS->Kv) * botfn;
        dhc = 0.5 * (dz_vel[K] + dz_vel[K - 1]);
        double h_shear;
        if (dhc > bbl_thick) h_shear = ((1.0 - botfn) * dhc + botfn * bbl_thick) + hn;
        else h_shear = dhc + hn;
        a_cpl[K] = Kv_tot[K] / (h_shear + (I_amax * Kv_tot[K]));
      }
    } else if (fabs(CS->Kv_extra_bbl) > 0.0) {        /* :1977-1995 */
      a_cpl[nz] = (Kv_tot[nz] + CS->Kv_extra_bbl) / ((0.5 * dz_vel[nz - 1] + hn) + I_amax * (Kv_tot[nz] + CS->Kv_extra_bbl));
      for (int K = nz - 1; K >= 1; K--) {
        const double z2 = z_i[K], botfn = 1.0 / (1.0 + 0.09 * z2 * z2 * z2 * z2 * z2 * z2);
        Kv_tot[K] = Kv_tot[K] + CS->Kv_extra_bbl * botfn;
        const double h_shear = 0.5 * (dz_vel[K] + dz_vel[K - 1] + hn);
        a_cpl[K] = Kv_tot[K] / (h_shear + I_amax * Kv_tot[K]);
      }
    } else {                                          /* :1996-2007 */
      a_cpl[nz] = Kv_tot[nz] / ((0.5 * dz_vel[nz - 1] + hn) + I_amax * Kv_tot[nz]);
      for (int K = nz - 1; K >= 1; K--) {
        const double h_shear = 0.5 * (dz_vel[K] + dz_vel[K - 1] + hn);
        a_cpl[K] = Kv_tot[K] / (h_shear + I_amax * Kv_tot[K]);
      }
    }
    /* no shelf (:2012-2045 not taken).  The surface boundary layer :2047-2252, for DYNAMIC_VISCOUS_ML or a bulk mixed layer
     * (GV%nkml > 0), without FIXED_DEPTH_LOTW_ML / LOTW_VISCOUS_ML_FLOOR; Boussinesq; hvel is dz_vel here */
    if (surface_bl(CS)) {
      /* :2098-2116: u_star = the mean of the two cells' (find_ustar: forces%ustar), absf from the two corners */
      double u_star = 0.5 * (visc->ustar[c0] + visc->ustar[c1]);
      if (side) u_star = visc->ustar[(side < 0) ? c0 : c1];      /* :2095-2110 */
      const double absf = 0.5 * (fabs(G->CoriolisBu[qA]) + fabs(G->CoriolisBu[qB]));
      int nk_in_ml = 0;
      double h_ml = hn;                       /* :2123 / :2163 */
      if (CS->dynamic_viscous_ML) {           /* :2120-2150 */
        const double nkv = nkml_visc_2d[f2];
        nk_in_ml = (int)ceil(nkv);
        for (int k = 1; k <= nk_in_ml; k++) {
          if ((double)k <= nkv) h_ml = h_ml + dz_vel[k - 1];
          else if ((double)k < nkv + 1.0) h_ml = h_ml + ((nkv + 1.0) - (double)k) * dz_vel[k - 1];
        }
      } else {                                /* :2152-2166 */
        nk_in_ml = CS->nkml;
        for (int k = 1; k <= CS->nkml; k++) h_ml = h_ml + dz_vel[k - 1];
      }
      if (u_star <= 0.0) nk_in_ml = 0;        /* :2184 */
      double z_t = 0.0;
      for (int K = 2; K <= nk_in_ml; K++) {   /* :2231-2250 (a_cpl(K), K one-based: a_cpl[K-1]) */
        z_t = z_t + dz_vel[K - 2];
        const double temp1 = (z_t * h_ml - z_t * z_t);
        const double visc_ml = u_star * CS->vonKar * (G->Z_to_H * temp1 * u_star) / (absf * temp1 + (h_ml + hn) * u_star);
        const double a_ml = visc_ml / (0.25 * (dz_vel[K - 1] + dz_vel[K - 2] + hn) + 0.5 * I_amax * visc_ml);
        a_cpl[K - 1] = max2(a_cpl[K - 1], a_ml);
      }
    }
  }
  for (int K = 0; K <= nz; K++) a_out[f2 + fpl * K] = min2(a_cpl_max, a_cpl[K] + 0.0);      /* :1504-1506 (a_cpl_gl90 = 0) */
  for (int k = 0; k < nz; k++) h_out[f2 + fpl * k] = hvel[k] + h_neglect;                  /* :1510 */
#undef DZ
  free(w);
}

/* the cell an open-boundary face projects from: the segment of OBC%segnum_u | segnum_v at the face, by its direction */
static int obc_side(const mom6hip_obc_t *OBC, const int32_t *segnum, long f2) {
  if (!OBC || OBC->number_of_segments <= 0 || segnum[f2] == MOM6HIP_OBC_NONE) return 0;
  const int dir = OBC->segment[segnum[f2] - 1].direction;
  if (dir == MOM6HIP_OBC_DIRECTION_E || dir == MOM6HIP_OBC_DIRECTION_N) return -1;
  if (dir == MOM6HIP_OBC_DIRECTION_W || dir == MOM6HIP_OBC_DIRECTION_S) return 1;
  return 0;
}

int orc_vertvisc_coef(const mom6hip_grid_t *G, mom6hip_vertvisc_cs_t *CS, const double *u, const double *v, const double *h,
                      const double *dz, const mom6hip_vertvisc_type_t *visc, double dt) {
  return orc_vertvisc_coef_obc(G, CS, u, v, h, dz, visc, dt, NULL);
}

int orc_vertvisc_coef_obc(const mom6hip_grid_t *G, mom6hip_vertvisc_cs_t *CS, const double *u, const double *v, const double *h,
                          const double *dz, const mom6hip_vertvisc_type_t *visc, double dt, const mom6hip_obc_t *OBC) {
  (void)dt;
  if (OBC && OBC->number_of_segments > 0 && !(OBC->segment && OBC->segnum_u && OBC->segnum_v)) return 1;
  if (unsupported(CS) || visc->Kv_shear_Bu) return 1;
  if (CS->bottomdraglaw && !(visc->Kv_bbl_u && visc->Kv_bbl_v && visc->bbl_thick_u && visc->bbl_thick_v)) return 1;
  if (surface_bl(CS) && !visc->ustar) return 1;
  if (CS->dynamic_viscous_ML && !(visc->nkml_visc_u && visc->nkml_visc_v)) return 1;
  const int is = G->isc, ie = G->iec, js = G->jsc, je = G->jec;
  const long nih = ORC_NIH(G), njh = ORC_NJH(G), hpl = nih * njh, upl = (nih + 1) * njh, vpl = nih * (njh + 1);
  ORC_PAR
  for (int j = js; j <= je; j++) for (int I = is - 1; I <= ie; I++) {
    if (!(G->mask2dCu[ORC_U2(G, I, j)] > 0.0)) continue;
    coef_column(G, CS, visc, u, h, dz, ORC_H2(G, I, j), ORC_H2(G, I + 1, j), ORC_U2(G, I, j), hpl, upl, visc->Kv_bbl_u,
                visc->bbl_thick_u, CS->a_u, CS->h_u, ORC_Q2(G, I, j - 1), ORC_Q2(G, I, j), visc->nkml_visc_u,
                obc_side(OBC, OBC ? OBC->segnum_u : NULL, ORC_U2(G, I, j)));
  }
  ORC_PAR
  for (int J = js - 1; J <= je; J++) for (int i = is; i <= ie; i++) {
    if (!(G->mask2dCv[ORC_V2(G, i, J)] > 0.0)) continue;
    coef_column(G, CS, visc, v, h, dz, ORC_H2(G, i, J), ORC_H2(G, i, J + 1), ORC_V2(G, i, J), hpl, vpl, visc->Kv_bbl_v,
                visc->bbl_thick_v, CS->a_v, CS->h_v, ORC_Q2(G, i - 1, J), ORC_Q2(G, i, J), visc->nkml_visc_v,
                obc_side(OBC, OBC ? OBC->segnum_v : NULL, ORC_V2(G, i, J)));
  }
  return 0;
}

/* the implicit solve of one face column, :646-760 (u) and :862-960 (v): `x` is the velocity (vertvisc) or, with
 * remnant != 0, visc_rem (vertvisc_remnant :1105-1124).  c1 is work space of nz doubles. */
static void solve_column(int nz, double dt, const double *a, const double *hv, const double *Ray, long f2, long fpl, double *x,
                         double surface_stress, int remnant, double *c1) {
  double b_denom_1 = hv[f2] + dt * ((Ray ? Ray[f2] : 0.0) + a[f2]);
  double b1 = 1.0 / (b_denom_1 + dt * a[f2 + fpl]);
  double d1 = b_denom_1 * b1;
  if (remnant) x[f2] = b1 * hv[f2];
  else x[f2] = b1 * (hv[f2] * x[f2] + surface_stress);
  for (int k = 1; k < nz; k++) {
    const long n = f2 + fpl * k;
    c1[k] = dt * a[n] * b1;
    b_denom_1 = hv[n] + dt * ((Ray ? Ray[n] : 0.0) + a[n] * d1);
    b1 = 1.0 / (b_denom_1 + dt * a[n + fpl]);
    d1 = b_denom_1 * b1;
    if (remnant) x[n] = (hv[n] + dt * a[n] * x[n - fpl]) * b1;
    else x[n] = (hv[n] * x[n] + dt * a[n] * x[n - fpl]) * b1;
  }
  for (int k = nz - 2; k >= 0; k--) x[f2 + fpl * k] = x[f2 + fpl * k] + c1[k + 1] * x[f2 + fpl * (k + 1)];
}

int orc_vertvisc(const mom6hip_grid_t *G, mom6hip_vertvisc_cs_t *CS, double *u, double *v, const double *h, const double *taux,
                 const double *tauy, const mom6hip_vertvisc_type_t *visc, double dt, double *taux_bot, double *tauy_bot) {
  return orc_vertvisc_obc(G, CS, u, v, h, taux, tauy, visc, dt, taux_bot, tauy_bot, NULL);
}

int orc_vertvisc_obc(const mom6hip_grid_t *G, mom6hip_vertvisc_cs_t *CS, double *u, double *v, const double *h, const double *taux,
                     const double *tauy, const mom6hip_vertvisc_type_t *visc, double dt, double *taux_bot, double *tauy_bot,
                     const mom6hip_obc_t *OBC) {
  if (unsupported(CS)) return 1;
  const int is = G->isc, ie = G->iec, js = G->jsc, je = G->jec, nz = G->nk;
  const long nih = ORC_NIH(G), njh = ORC_NJH(G), hpl = nih * njh, upl = (nih + 1) * njh, vpl = nih * (njh + 1);
  const double dt_Rho0 = dt / CS->H_to_RZ, h_neglect = G->H_subroundoff;      /* :611-612 */
  double Hmix = 0.0, I_Hmix = 0.0;
  if (CS->direct_stress) { Hmix = CS->Hmix_stress; I_Hmix = 1.0 / Hmix; }       /* :607-610 */
  for (int dir = 0; dir < 2; dir++) {
    double *x = dir ? v : u;
    const double *tau = dir ? tauy : taux, *mask = dir ? G->mask2dCv : G->mask2dCu, *a = dir ? CS->a_v : CS->a_u,
                 *hv = dir ? CS->h_v : CS->h_u, *Ray = dir ? visc->Ray_v : visc->Ray_u;
    double *tbot = dir ? tauy_bot : taux_bot;
    const long fpl = dir ? vpl : upl;
    ORC_PAR      /* the columns are independent (the reference: !$OMP parallel do over j, :640) */
    for (int j = (dir ? js - 1 : js); j <= je; j++) for (int i = (dir ? is : is - 1); i <= ie; i++) {
      double c1[nz];
      const long f2 = dir ? ORC_V2(G, i, j) : ORC_U2(G, i, j);
      const long c0 = ORC_H2(G, i, j), cc1 = dir ? ORC_H2(G, i, j + 1) : ORC_H2(G, i + 1, j);
      const int do_i = mask[f2] > 0.0;
      double surface_stress;
      if (CS->direct_stress) {      /* :671-685 */
        surface_stress = 0.0;
        if (do_i) {
          double zDS = 0.0;
          const double stress = dt_Rho0 * tau[f2];
          for (int k = 0; k < nz; k++) {
            const double h_a = 0.5 * (h[c0 + hpl * k] + h[cc1 + hpl * k]) + h_neglect;
            double hfr = 1.0;
            if ((zDS + h_a) > Hmix) hfr = (Hmix - zDS) / h_a;
            x[f2 + fpl * k] = x[f2 + fpl * k] + I_Hmix * hfr * stress;
            zDS = zDS + h_a;
            if (zDS >= Hmix) break;
          }
        }
      } else {
        surface_stress = dt_Rho0 * (mask[f2] * tau[f2]);      /* :687 */
      }
      if (do_i) solve_column(nz, dt, a, hv, Ray, f2, fpl, x, surface_stress, 0, c1);
      if (tbot) {      /* :798-805 (every point of the row, masked or not) */
        tbot[f2] = CS->H_to_RZ * (x[f2 + fpl * (nz - 1)] * a[f2 + fpl * nz]);
        if (Ray) for (int k = 0; k < nz; k++) tbot[f2] = tbot[f2] + CS->H_to_RZ * (Ray[f2 + fpl * k] * x[f2 + fpl * k]);
      }
    }
  }
  /* vertvisc_limit_vel :2259-2462 (no U_TRUNC_FILE / V_TRUNC_FILE) */
  {
    const double maxvel = CS->maxvel, truncvel = 0.9 * maxvel, H_report = 6.0 * G->Angstrom_H;
    for (int dir = 0; dir < 2; dir++) {
      double *x = dir ? v : u;
      const double *dL = dir ? G->dx_Cv : G->dy_Cu;
      const long fpl = dir ? vpl : upl;
      long ntr = 0;
      _Pragma("omp parallel for schedule(static) reduction(+:ntr)")
      for (int k = 0; k < nz; k++) for (int j = (dir ? js - 1 : js); j <= je; j++) for (int i = (dir ? is : is - 1); i <= ie; i++) {
        const long f2 = dir ? ORC_V2(G, i, j) : ORC_U2(G, i, j), n = f2 + fpl * k;
        const long c0 = ORC_H2(G, i, j), cc1 = dir ? ORC_H2(G, i, j + 1) : ORC_H2(G, i + 1, j);
        const double hsum = h[c0 + hpl * k] + h[cc1 + hpl * k];
        if (CS->CFL_based_trunc) {
          if (fabs(x[n]) < CS->vel_underflow) { x[n] = 0.0; }
          else if ((x[n] * (dt * dL[f2])) * G->IareaT[cc1] < -CS->CFL_trunc) {
            x[n] = (-0.9 * CS->CFL_trunc) * (G->areaT[cc1] / (dt * dL[f2]));
            if (hsum > H_report) ntr = ntr + 1;
          } else if ((x[n] * (dt * dL[f2])) * G->IareaT[c0] > CS->CFL_trunc) {
            x[n] = (0.9 * CS->CFL_trunc) * (G->areaT[c0] / (dt * dL[f2]));
            if (hsum > H_report) ntr = ntr + 1;
          }
        } else {
          if (fabs(x[n]) < CS->vel_underflow) { x[n] = 0.0; }
          else if (fabs(x[n]) > maxvel) {
            x[n] = copysign(truncvel, x[n]);
            if (hsum > H_report) ntr = ntr + 1;
          }
        }
      }
      CS->ntrunc = CS->ntrunc + ntr;
    }
  }
  /* :988-1006: the velocities of the specified open-boundary segments */
  if (OBC) for (int n = 0; n < OBC->number_of_segments; n++) {
    const mom6hip_obc_segment_t *S = &OBC->segment[n];
    if (!S->specified || !S->on_pe) continue;
    if (!S->normal_vel) return 1;
    const long nA = S->is_N_or_S ? (S->ied - S->isd + 1) : (S->IedB - S->IsdB + 1);      /* the segment's own arrays: (i, j, k) */
    const long nB = S->is_N_or_S ? (S->JedB - S->JsdB + 1) : (S->jed - S->jsd + 1);
    if (S->is_N_or_S) {
      const int J = S->JsdB;
      for (int k = 0; k < nz; k++) for (int i = S->isd; i <= S->ied; i++)
        v[ORC_V2(G, i, J) + vpl * k] = S->normal_vel[(i - S->isd) + nA * ((J - S->JsdB) + nB * (long)k)];
    } else if (S->is_E_or_W) {
      const int I = S->IsdB;
      for (int k = 0; k < nz; k++) for (int j = S->jsd; j <= S->jed; j++)
        u[ORC_U2(G, I, j) + upl * k] = S->normal_vel[(I - S->IsdB) + nA * ((j - S->jsd) + nB * (long)k)];
    }
  }
  return 0;
}

int orc_vertvisc_remnant(const mom6hip_grid_t *G, const mom6hip_vertvisc_cs_t *CS, const mom6hip_vertvisc_type_t *visc,
                         double *visc_rem_u, double *visc_rem_v, double dt) {
  if (unsupported(CS)) return 1;
  const int is = G->isc, ie = G->iec, js = G->jsc, je = G->jec, nz = G->nk;
  const long nih = ORC_NIH(G), njh = ORC_NJH(G), upl = (nih + 1) * njh, vpl = nih * (njh + 1);
  ORC_PAR
  for (int j = js; j <= je; j++) for (int I = is - 1; I <= ie; I++) {
    double c1[nz];
    const long f2 = ORC_U2(G, I, j);
    if (G->mask2dCu[f2] > 0.0) solve_column(nz, dt, CS->a_u, CS->h_u, visc->Ray_u, f2, upl, visc_rem_u, 0.0, 1, c1);
  }
  ORC_PAR
  for (int J = js - 1; J <= je; J++) for (int i = is; i <= ie; i++) {
    double c1[nz];
    const long f2 = ORC_V2(G, i, J);
    if (G->mask2dCv[f2] > 0.0) solve_column(nz, dt, CS->a_v, CS->h_v, visc->Ray_v, f2, vpl, visc_rem_v, 0.0, 1, c1);
  }
  return 0;
}
