/*
 * set_viscosity.c -- CPU restatement of set_viscous_BBL (TEST INFRASTRUCTURE, see mom6_oracle.h).
 *
 * Reference: src/parameterizations/vertical/MOM_set_viscosity.F90
 *   set_viscous_BBL :134-1100, set_v_at_u :1804, set_u_at_v :1849, set_viscous_ML :1898 (the early return :2043, and the
 *   DYNAMIC_VISCOUS_ML search :2111-2230, :2400-2506; its exp through orc_cr_exp, correctly rounded)
 * Restated branch: BOTTOMDRAGLAW, quadratic or LINEAR_DRAG law, BBL_USE_EOS or GV%Rlay as the density variable, no channel
 * drag, no tidal background velocity, Boussinesq (no tv%SpV_avg), no tv%p_surf, no OBC; DRAG_AS_BODY_FORCE; CORRECT_BBL_BOUNDS.
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

/* set_v_at_u :1804-1846 (no OBC) */
static double set_v_at_u(const mom6hip_grid_t *G, const double *v, const double *h, int i, int j, int k) {
  const int J = j;
  double hwt[2][2];      /* hwt(i0, j0), i0 = 0:1, j0 = -1:0 -> hwt[i0][j0+1] */
  for (int j0 = -1; j0 <= 0; j0++) for (int i0 = 0; i0 <= 1; i0++) {
    const int i1 = i + i0, J1 = J + j0;
    hwt[i0][j0 + 1] = (h[H3(i1, J1, k)] + h[H3(i1, J1 + 1, k)]) * G->mask2dCv[ORC_V2(G, i1, J1)];
  }
  const double hwt_tot = (hwt[0][0] + hwt[1][1]) + (hwt[1][0] + hwt[0][1]);
  double r = 0.0;
  if (hwt_tot > 0.0)
    r = ((hwt[0][1] * v[V3(i, J, k)] + hwt[1][0] * v[V3(i + 1, J - 1, k)]) +
         (hwt[1][1] * v[V3(i + 1, J, k)] + hwt[0][0] * v[V3(i, J - 1, k)])) / hwt_tot;
  return r;
}

/* set_u_at_v :1849-1891 (no OBC) */
static double set_u_at_v(const mom6hip_grid_t *G, const double *u, const double *h, int i, int j, int k) {
  const int I = i;
  double hwt[2][2];      /* hwt(i0, j0), i0 = -1:0, j0 = 0:1 -> hwt[i0+1][j0] */
  for (int j0 = 0; j0 <= 1; j0++) for (int i0 = -1; i0 <= 0; i0++) {
    const int I1 = I + i0, j1 = j + j0;
    hwt[i0 + 1][j0] = (h[H3(I1, j1, k)] + h[H3(I1 + 1, j1, k)]) * G->mask2dCu[ORC_U2(G, I1, j1)];
  }
  const double hwt_tot = (hwt[0][0] + hwt[1][1]) + (hwt[1][0] + hwt[0][1]);
  double r = 0.0;
  if (hwt_tot > 0.0)
    r = ((hwt[1][0] * u[U3(I, j, k)] + hwt[0][1] * u[U3(I - 1, j + 1, k)]) +
         (hwt[0][0] * u[U3(I - 1, j, k)] + hwt[1][1] * u[U3(I, j + 1, k)])) / hwt_tot;
  return r;
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
  if (CS->body_force_drag && !(Ray_u && Ray_v)) return 2;
  const double cdrag_sqrt = sqrt(CS->cdrag);
  const double cdrag_sqrt_H = cdrag_sqrt * 1.0 * G->Z_to_H;      /* US%L_to_m*GV%m_to_H */
  const double cdrag_L_to_H = CS->cdrag * 1.0 * G->Z_to_H;
  const double BBL_thick_max = CS->BBL_thick_max;
  const int K2 = 2;      /* max(nkmb+1, 2) with nkmb = 0 */
  const long upl = (long)(ORC_NIH(G) + 1) * ORC_NJH(G), vpl = (long)ORC_NIH(G) * (ORC_NJH(G) + 1);
  if (Ray_u) memset(Ray_u, 0, sizeof(double) * upl * nz);      /* :416-417 */
  if (Ray_v) memset(Ray_v, 0, sizeof(double) * vpl * nz);

  ORC_PAR
  for (int j = Jsq; j <= Jeq; j++) for (int m = 1; m <= 2; m++) {
    if (m == 1 && j < G->jsc) continue;
    const int is = (m == 1) ? Isq : G->isc, ie = (m == 1) ? Ieq : G->iec;
    double h_at_vel[nz + 1], dz_at_vel[nz + 1], h_vel[nz + 1], T_vel[nz + 1], S_vel[nz + 1];
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
        if (use_BBL_EOS) { T_vel[k] = 0.5 * (T[H3(i, j, k)] + T[H3(ip, jp, k)]); S_vel[k] = 0.5 * (S[H3(i, j, k)] + S[H3(ip, jp, k)]); }
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
              const double v_at_u = set_v_at_u(G, v, h, i, j, k);
              hutot = hutot + hweight * sqrt(u[U3(I, j, k)] * u[U3(I, j, k)] + v_at_u * v_at_u + u2_bg);
            } else {
              const double u_at_v = set_u_at_v(G, u, h, i, j, k);
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
      if ((bbl_thick > 0.5 * CS->dz_bbl) && (CS->RiNo_mix)) bbl_thick = 0.5 * CS->dz_bbl;
      if (CS->body_force_drag) bbl_thick = dz_bbl_drag;
      /* not channel drag :1010-1022 */
      double kv_bbl;
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
  return 0;
}
