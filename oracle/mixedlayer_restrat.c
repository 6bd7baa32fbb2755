/*
 * mixedlayer_restrat.c -- CPU restatement of mixedlayer_restrat (TEST INFRASTRUCTURE, see mom6_oracle.h).
 *
 * Reference: src/parameterizations/lateral/MOM_mixed_layer_restrat.F90
 *   mixedlayer_restrat :135-172, mixedlayer_restrat_OM4 :175-720, mu :723-757, mixedlayer_restrat_BML :1209-1486;
 *   find_ustar src/core/MOM_forcing_type.F90:1236-1297 (forces%ustar, Boussinesq, H_T_units).
 * Restated branch: Boussinesq, no Stanley variance, no Bodner et al. (2023) form.
 * PINNED (the shape function): the known answers of mixedlayer_restrat_unit_tests :1847-1872 (tests/golden/mle_mu.json); the
 * transports themselves have no vectors in the reference -- invariants in tests/test_mixedlayer_restrat.py.
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
#define Q2(i,j) ORC_Q2(G,i,j)

/* mu :723-757.  x**(1+2 dh) of a correctly rounded power (the reference's is the Fortran intrinsic); 0**y = 0, x**1 = x */
double orc_mle_mu(double sigma, double dh) {
  const double s21 = 2. * sigma + 1.;
  double mu = max2(0., (1. - s21 * s21) * (1. + (5. / 21.) * (s21 * s21)));
  const double xp = max2(0., min2(1., (-sigma - 0.5) * 2. / (1. + 2. * dh)));
  const double base = max2(1. - (xp * xp) * (3. - 2. * xp), 0.);
  const double ex = 1. + 2. * dh;
  double dd;
  if (base == 0.0) dd = 0.0;
  else if (ex == 1.0 || base == 1.0) dd = base;
  else dd = orc_cr_pow(base, ex);
  const double bottop = 0.5 * (1. - copysign(1., sigma + 0.5));
  return max2(mu, dd * bottop);
}

/* the overturning timescale times the coefficient :523-529 (the same five lines at every face, fast and slow) */
static double timescale_of(double vonKar_x_pi2, double u_star, double absf, double h_vel, double h_neglect, double coef) {
  const double mom_mixrate = vonKar_x_pi2 * (u_star * u_star) / (absf * (h_vel * h_vel) + 4.0 * (h_vel + h_neglect) * u_star);
  double timescale = 0.0625 * (absf + 2.0 * mom_mixrate) / (absf * absf + mom_mixrate * mom_mixrate);
  timescale = timescale * coef;
  return timescale;
}

static int restrat_OM4(const mom6hip_grid_t *G, const mom6hip_mixedlayer_restrat_cs_t *CS, double *h, double *uhtr, double *vhtr,
                       const double *T, const double *S, const mom6hip_eos_t *EOS, const double *ustar, double dt, const double *h_MLD,
                       double *uhml, double *vhml) {
  const int nz = G->nk, is = G->isc, ie = G->iec, js = G->jsc, je = G->jec;
  const long n2 = (long)ORC_NIH(G) * ORC_NJH(G);
  const double h_min = 0.5 * G->Angstrom_H;
  const double vonKar_x_pi2 = CS->vonKar * 9.8696;
  double *MLD_fast = (double *)calloc(n2, 8), *MLD_slow = (double *)calloc(n2, 8), *htot_fast = (double *)calloc(n2, 8),
         *htot_slow = (double *)calloc(n2, 8), *Rml_av_fast = (double *)calloc(n2, 8), *Rml_av_slow = (double *)calloc(n2, 8),
         *U_star_2d = (double *)calloc(n2, 8), *h_avail = (double *)calloc(n2 * nz, 8);
  double *a = (double *)malloc(8 * (nz + 1)), *b = (double *)malloc(8 * (nz + 1));
  int rc = 0;
  for (int j = js - 1; j <= je + 1; j++) for (int i = is - 1; i <= ie + 1; i++) U_star_2d[H2(i,j)] = G->Z_to_H * ustar[H2(i,j)];

  if (CS->MLE_density_diff > 0.) {      /* :283-327 */
    for (int j = js - 1; j <= je + 1; j++) for (int i = is - 1; i <= ie + 1; i++) {
      double dK = 0.5 * h[H3(i,j,1)], dKm1;
      const double rhoSurf = orc_eos_density(EOS, T[H3(i,j,1)], S[H3(i,j,1)], 0.0);
      double deltaRhoAtK = 0., deltaRhoAtKm1, mld = 0.;
      for (int k = 2; k <= nz; k++) {
        dKm1 = dK;
        dK = dK + 0.5 * (h[H3(i,j,k)] + h[H3(i,j,k-1)]);
        deltaRhoAtKm1 = deltaRhoAtK;
        deltaRhoAtK = orc_eos_density(EOS, T[H3(i,j,k)], S[H3(i,j,k)], 0.0);
        deltaRhoAtK = deltaRhoAtK - rhoSurf;
        const double ddRho = deltaRhoAtK - deltaRhoAtKm1;
        if ((mld == 0.) && (ddRho > 0.) && (deltaRhoAtKm1 < CS->MLE_density_diff) && (deltaRhoAtK >= CS->MLE_density_diff)) {
          const double aFac = (CS->MLE_density_diff - deltaRhoAtKm1) / ddRho;
          mld = dK * aFac + dKm1 * (1. - aFac);
        }
      }
      mld = CS->MLE_MLD_stretch * mld;
      if ((mld == 0.) && (deltaRhoAtK < CS->MLE_density_diff)) mld = dK;
      MLD_fast[H2(i,j)] = mld;
    }
  } else if (CS->MLE_use_PBL_MLD) {
    for (int j = js - 1; j <= je + 1; j++) for (int i = is - 1; i <= ie + 1; i++) MLD_fast[H2(i,j)] = CS->MLE_MLD_stretch * h_MLD[H2(i,j)];
  } else { rc = 2; goto done; }      /* "No MLD to use for MLE parameterization." */

  if (CS->MLE_MLD_decay_time > 0.) {      /* :330-345 */
    const double aFac = CS->MLE_MLD_decay_time / (dt + CS->MLE_MLD_decay_time), bFac = dt / (dt + CS->MLE_MLD_decay_time);
    for (int j = js - 1; j <= je + 1; j++) for (int i = is - 1; i <= ie + 1; i++) {
      CS->MLD_filtered[H2(i,j)] = max2(MLD_fast[H2(i,j)], bFac * MLD_fast[H2(i,j)] + aFac * CS->MLD_filtered[H2(i,j)]);
      MLD_fast[H2(i,j)] = CS->MLD_filtered[H2(i,j)];
    }
  }
  if (CS->MLE_MLD_decay_time2 > 0.) {      /* :348-367 */
    const double aFac = CS->MLE_MLD_decay_time2 / (dt + CS->MLE_MLD_decay_time2), bFac = dt / (dt + CS->MLE_MLD_decay_time2);
    for (int j = js - 1; j <= je + 1; j++) for (int i = is - 1; i <= ie + 1; i++) {
      CS->MLD_filtered_slow[H2(i,j)] = max2(MLD_fast[H2(i,j)], bFac * MLD_fast[H2(i,j)] + aFac * CS->MLD_filtered_slow[H2(i,j)]);
      MLD_slow[H2(i,j)] = CS->MLD_filtered_slow[H2(i,j)];
    }
  } else {
    for (int j = js - 1; j <= je + 1; j++) for (int i = is - 1; i <= ie + 1; i++) MLD_slow[H2(i,j)] = MLD_fast[H2(i,j)];
  }

  const double I4dt = 0.25 / dt;
  const double g_Rho0 = G->H_to_Z * G->g_Earth / G->Rho0;
  const double h_neglect = G->H_subroundoff;
  const int res_upscale = CS->front_length > 0.;
  const double I_LFront = res_upscale ? 1. / CS->front_length : 0.;

  /* the mixed layer averages :392-428 (keep_going only skips rows where no column has anything left to add) */
  for (int j = js - 1; j <= je + 1; j++) for (int i = is - 1; i <= ie + 1; i++) {
    double hf = 0.0, hs = 0.0, Rf = 0.0, Rs = 0.0;
    const double mf = MLD_fast[H2(i,j)], ms = MLD_slow[H2(i,j)];
    for (int k = 1; k <= nz; k++) {
      h_avail[H3(i,j,k)] = max2(I4dt * G->areaT[H2(i,j)] * (h[H3(i,j,k)] - G->Angstrom_H), 0.0);
      if (hf < mf || hs < ms) {
        const double rho_ml = orc_eos_density(EOS, T[H3(i,j,k)], S[H3(i,j,k)], 0.0);
        if (hf < mf) { const double dh = min2(h[H3(i,j,k)], mf - hf); Rf = Rf + dh * rho_ml; hf = hf + dh; }
        if (hs < ms) { const double dh = min2(h[H3(i,j,k)], ms - hs); Rs = Rs + dh * rho_ml; hs = hs + dh; }
      }
    }
    htot_fast[H2(i,j)] = hf; htot_slow[H2(i,j)] = hs;
    Rml_av_fast[H2(i,j)] = -(g_Rho0 * Rf) / (hf + h_neglect);
    Rml_av_slow[H2(i,j)] = -(g_Rho0 * Rs) / (hs + h_neglect);
  }

  /* the faces :505-688: dir 0 the u faces I = is-1..ie of rows js..je, dir 1 the v faces J = js-1..je of columns is..ie */
  for (int dir = 0; dir < 2; dir++) {
    const int di = dir ? 0 : 1, dj = dir ? 1 : 0;
    for (int j = (dir ? js - 1 : js); j <= je; j++) for (int i = (dir ? is : is - 1); i <= ie; i++) {
      const long c0 = H2(i,j), c1 = H2(i+di,j+dj);
      const double u_star = max2(CS->ustar_min, 0.5 * (U_star_2d[c0] + U_star_2d[c1]));
      const double absf = dir ? 0.5 * (fabs(G->CoriolisBu[Q2(i-1,j)]) + fabs(G->CoriolisBu[Q2(i,j)]))
                              : 0.5 * (fabs(G->CoriolisBu[Q2(i,j-1)]) + fabs(G->CoriolisBu[Q2(i,j)]));
      const double dx = dir ? G->dxCv[V2(i,j)] : G->dxCu[U2(i,j)], dy = dir ? G->dyCv[V2(i,j)] : G->dyCu[U2(i,j)];
      double res_scaling_fac = 0.0;
      if (res_upscale) res_scaling_fac = (sqrt(0.5 * (dx * dx + dy * dy)) * I_LFront) * min2(1., 0.5 * (CS->Rd_dx_h[c0] + CS->Rd_dx_h[c1]));
      /* timescale * G%OBCmaskCu * G%dyCu * G%IdxCu * (...) * (h_vel**2) associates from the left (:523-524, :538-539, :1371-1372): the three
       * metric factors multiply the timescale one after the other, they are NOT a product of their own */
      const double gm = dir ? G->mask2dCv[V2(i,j)] : G->mask2dCu[U2(i,j)], gl = dir ? G->dxCv[V2(i,j)] : G->dyCu[U2(i,j)],
                   gi = dir ? G->IdyCv[V2(i,j)] : G->IdxCu[U2(i,j)];

      double h_vel = 0.5 * ((htot_fast[c0] + htot_fast[c1]) + h_neglect);
      double timescale = timescale_of(vonKar_x_pi2, u_star, absf, h_vel, h_neglect, CS->ml_restrat_coef);
      if (res_upscale) timescale = timescale * res_scaling_fac;
      double Dml = timescale * gm * gl * gi * (Rml_av_fast[c1] - Rml_av_fast[c0]) * (h_vel * h_vel);

      h_vel = 0.5 * ((htot_slow[c0] + htot_slow[c1]) + h_neglect);
      timescale = timescale_of(vonKar_x_pi2, u_star, absf, h_vel, h_neglect, CS->ml_restrat_coef2);
      if (res_upscale) timescale = timescale * res_scaling_fac;
      double Dml_slow = timescale * gm * gl * gi * (Rml_av_slow[c1] - Rml_av_slow[c0]) * (h_vel * h_vel);

      double *hml = dir ? vhml : uhml, *htr = dir ? vhtr : uhtr;
      if (Dml + Dml_slow == 0.) {
        for (int k = 1; k <= nz; k++) hml[dir ? V3(i,j,k) : U3(i,j,k)] = 0.0;
      } else {
        const double IhTot = 2.0 / ((htot_fast[c0] + htot_fast[c1]) + h_neglect);
        const double IhTot_slow = 2.0 / ((htot_slow[c0] + htot_slow[c1]) + h_neglect);
        double zpa = 0.0, zpb = 0.0;
        for (int k = 1; k <= nz; k++) {
          const double hAtVel = 0.5 * (h[H3(i,j,k)] + h[H3(i+di,j+dj,k)]);
          a[k] = orc_mle_mu(zpa, CS->MLE_tail_dh);
          zpa = zpa - (hAtVel * IhTot);
          a[k] = a[k] - orc_mle_mu(zpa, CS->MLE_tail_dh);
          if (a[k] * Dml > 0.0) {
            if (a[k] * Dml > h_avail[H3(i,j,k)]) Dml = h_avail[H3(i,j,k)] / a[k];
          } else if (a[k] * Dml < 0.0) {
            if (-a[k] * Dml > h_avail[H3(i+di,j+dj,k)]) Dml = -h_avail[H3(i+di,j+dj,k)] / a[k];
          }
        }
        for (int k = 1; k <= nz; k++) {
          const double hAtVel = 0.5 * (h[H3(i,j,k)] + h[H3(i+di,j+dj,k)]);
          b[k] = orc_mle_mu(zpb, CS->MLE_tail_dh);
          zpb = zpb - (hAtVel * IhTot_slow);
          b[k] = b[k] - orc_mle_mu(zpb, CS->MLE_tail_dh);
          if (b[k] * Dml_slow > 0.0) {
            if (b[k] * Dml_slow > h_avail[H3(i,j,k)] - a[k] * Dml) Dml_slow = max2(0., h_avail[H3(i,j,k)] - a[k] * Dml) / b[k];
          } else if (b[k] * Dml_slow < 0.0) {
            if (-b[k] * Dml_slow > h_avail[H3(i+di,j+dj,k)] + a[k] * Dml)
              Dml_slow = -max2(0., h_avail[H3(i+di,j+dj,k)] + a[k] * Dml) / b[k];
          }
        }
        for (int k = 1; k <= nz; k++) {
          const long n = dir ? V3(i,j,k) : U3(i,j,k);
          hml[n] = a[k] * Dml + b[k] * Dml_slow;
          htr[n] = htr[n] + hml[n] * dt;
        }
      }
    }
  }
  for (int j = js; j <= je; j++) for (int k = 1; k <= nz; k++) for (int i = is; i <= ie; i++) {      /* :690-696 */
    h[H3(i,j,k)] = h[H3(i,j,k)] - dt * G->IareaT[H2(i,j)] * ((uhml[U3(i,j,k)] - uhml[U3(i-1,j,k)]) + (vhml[V3(i,j,k)] - vhml[V3(i,j-1,k)]));
    if (h[H3(i,j,k)] < h_min) h[H3(i,j,k)] = h_min;
  }
done:
  free(MLD_fast); free(MLD_slow); free(htot_fast); free(htot_slow); free(Rml_av_fast); free(Rml_av_slow); free(U_star_2d); free(h_avail);
  free(a); free(b);
  return rc;
}

/* mixedlayer_restrat_BML :1209-1486 */
static int restrat_BML(const mom6hip_grid_t *G, const mom6hip_mixedlayer_restrat_cs_t *CS, double *h, double *uhtr, double *vhtr,
                       const double *T, const double *S, const mom6hip_eos_t *EOS, const double *ustar, double dt, double *uhml, double *vhml) {
  const int nz = G->nk, is = G->isc, ie = G->iec, js = G->jsc, je = G->jec, nkml = CS->nkml;
  const long n2 = (long)ORC_NIH(G) * ORC_NJH(G);
  if ((nkml < 2) || (CS->ml_restrat_coef <= 0.0)) return 0;
  const double h_min = 0.5 * G->Angstrom_H;
  const double I4dt = 0.25 / dt;
  const double g_Rho0 = G->H_to_Z * G->g_Earth / G->Rho0;
  const double vonKar_x_pi2 = CS->vonKar * 9.8696;
  const double h_neglect = G->H_subroundoff;
  double *htot = (double *)calloc(n2, 8), *Rml_av = (double *)calloc(n2, 8), *U_star_2d = (double *)calloc(n2, 8),
         *h_avail = (double *)calloc(n2 * nz, 8);
  double *a = (double *)malloc(8 * (nz + 1));
  for (int j = js - 1; j <= je + 1; j++) for (int i = is - 1; i <= ie + 1; i++) {
    U_star_2d[H2(i,j)] = G->Z_to_H * ustar[H2(i,j)];
    double ht = 0.0, rho_int = 0.0;
    for (int k = 1; k <= nkml; k++) {
      const double Rho_ml = orc_eos_density(EOS, T[H3(i,j,k)], S[H3(i,j,k)], 0.0);
      rho_int = rho_int + h[H3(i,j,k)] * Rho_ml;
      ht = ht + h[H3(i,j,k)];
      h_avail[H3(i,j,k)] = max2(I4dt * G->areaT[H2(i,j)] * (h[H3(i,j,k)] - G->Angstrom_H), 0.0);
    }
    htot[H2(i,j)] = ht;
    Rml_av[H2(i,j)] = (g_Rho0 * rho_int) / (ht + h_neglect);
  }
  for (int dir = 0; dir < 2; dir++) {
    const int di = dir ? 0 : 1, dj = dir ? 1 : 0;
    for (int j = (dir ? js - 1 : js); j <= je; j++) for (int i = (dir ? is : is - 1); i <= ie; i++) {
      const long c0 = H2(i,j), c1 = H2(i+di,j+dj);
      const double h_vel = 0.5 * (htot[c0] + htot[c1]);
      const double u_star = max2(CS->ustar_min, 0.5 * (U_star_2d[c0] + U_star_2d[c1]));
      const double absf = dir ? 0.5 * (fabs(G->CoriolisBu[Q2(i-1,j)]) + fabs(G->CoriolisBu[Q2(i,j)]))
                              : 0.5 * (fabs(G->CoriolisBu[Q2(i,j-1)]) + fabs(G->CoriolisBu[Q2(i,j)]));
      const double timescale = timescale_of(vonKar_x_pi2, u_star, absf, h_vel, h_neglect, CS->ml_restrat_coef);
      /* timescale * G%OBCmaskCu * G%dyCu * G%IdxCu * (...) * (h_vel**2) associates from the left (:523-524, :538-539, :1371-1372): the three
       * metric factors multiply the timescale one after the other, they are NOT a product of their own */
      const double gm = dir ? G->mask2dCv[V2(i,j)] : G->mask2dCu[U2(i,j)], gl = dir ? G->dxCv[V2(i,j)] : G->dyCu[U2(i,j)],
                   gi = dir ? G->IdyCv[V2(i,j)] : G->IdxCu[U2(i,j)];
      double Dml = timescale * gm * gl * gi * (Rml_av[c1] - Rml_av[c0]) * (h_vel * h_vel);
      double *hml = dir ? vhml : uhml, *htr = dir ? vhtr : uhtr;
      if (Dml == 0) {
        for (int k = 1; k <= nkml; k++) hml[dir ? V3(i,j,k) : U3(i,j,k)] = 0.0;
      } else {
        const double I2htot = 1.0 / (htot[c0] + htot[c1] + h_neglect);
        double z_topx2 = 0.0;
        for (int k = 1; k <= nkml; k++) {
          const double hx2 = (h[H3(i,j,k)] + h[H3(i+di,j+dj,k)] + h_neglect);
          a[k] = (hx2 * I2htot) * (2.0 - 4.0 * (z_topx2 + 0.5 * hx2) * I2htot);
          z_topx2 = z_topx2 + hx2;
          if (a[k] * Dml > 0.0) {
            if (a[k] * Dml > h_avail[H3(i,j,k)]) Dml = h_avail[H3(i,j,k)] / a[k];
          } else {
            if (-a[k] * Dml > h_avail[H3(i+di,j+dj,k)]) Dml = -h_avail[H3(i+di,j+dj,k)] / a[k];
          }
        }
        for (int k = 1; k <= nkml; k++) {
          const long n = dir ? V3(i,j,k) : U3(i,j,k);
          hml[n] = a[k] * Dml;
          htr[n] = htr[n] + hml[n] * dt;
        }
      }
    }
  }
  for (int j = js; j <= je; j++) for (int k = 1; k <= nkml; k++) for (int i = is; i <= ie; i++) {
    h[H3(i,j,k)] = h[H3(i,j,k)] - dt * G->IareaT[H2(i,j)] * ((uhml[U3(i,j,k)] - uhml[U3(i-1,j,k)]) + (vhml[V3(i,j,k)] - vhml[V3(i,j-1,k)]));
    if (h[H3(i,j,k)] < h_min) h[H3(i,j,k)] = h_min;
  }
  /* the layers below the mixed layer carry no transport (:1467-1470 zeroes them for the diagnostics) */
  for (int k = nkml + 1; k <= nz; k++) {
    for (int j = js; j <= je; j++) for (int i = is - 1; i <= ie; i++) uhml[U3(i,j,k)] = 0.0;
    for (int j = js - 1; j <= je; j++) for (int i = is; i <= ie; i++) vhml[V3(i,j,k)] = 0.0;
  }
  free(htot); free(Rml_av); free(U_star_2d); free(h_avail); free(a);
  return 0;
}

int orc_mixedlayer_restrat(const mom6hip_grid_t *G, const mom6hip_mixedlayer_restrat_cs_t *CS, double *h, double *uhtr, double *vhtr,
                           const double *T, const double *S, const mom6hip_eos_t *EOS, const double *ustar, double dt, const double *h_MLD,
                           double *uhml, double *vhml) {
  if (!CS->initialized) return 1;
  for (int q = 0; q < 8; q++) if (CS->unsupported[q]) return 3;
  if (!EOS || !T || !S) return 4;      /* "An equation of state must be used with this module." */
  const long n3u = (long)(ORC_NIH(G) + 1) * ORC_NJH(G) * G->nk, n3v = (long)ORC_NIH(G) * (ORC_NJH(G) + 1) * G->nk;
  double *um = uhml ? uhml : (double *)calloc(n3u, 8), *vm = vhml ? vhml : (double *)calloc(n3v, 8);
  int rc;
  if (CS->nkml > 0) rc = restrat_BML(G, CS, h, uhtr, vhtr, T, S, EOS, ustar, dt, um, vm);
  else {
    if (CS->front_length > 0. && !CS->Rd_dx_h) rc = 5;
    else if ((CS->MLE_MLD_decay_time > 0. && !CS->MLD_filtered) || (CS->MLE_MLD_decay_time2 > 0. && !CS->MLD_filtered_slow)) rc = 6;
    else if (!(CS->MLE_density_diff > 0.) && CS->MLE_use_PBL_MLD && !h_MLD) rc = 7;
    else rc = restrat_OM4(G, CS, h, uhtr, vhtr, T, S, EOS, ustar, dt, h_MLD, um, vm);
  }
  if (!uhml) free(um);
  if (!vhml) free(vm);
  return rc;
}
