/*
 * pressure_force.c -- CPU restatement of PressureForce_FV_Bouss and what it calls (TEST INFRASTRUCTURE).
 *
 * Restates, for the hot-path configuration (Boussinesq, ALE with PLM reconstruction of T and S for the
 * pressure gradient -- RECONSTRUCT_FOR_PRESSURE=True, PRESSURE_RECONSTRUCTION_SCHEME=1 --, an equation of
 * state, GFS_scale = 1, no tides / SAL, no Stanley SGS terms, use_inaccurate_pgf_rho_anom = False):
 *   PressureForce_FV_Bouss          src/core/MOM_PressureForce_FV.F90:462-919
 *   int_density_dz_generic_plm      src/core/MOM_density_integrals.F90:369-769
 *   TS_PLM_edge_values / ALE_PLM_edge_values   src/ALE/MOM_ALE.F90:1495-1579
 *   Set_pbce_Bouss (use_EOS branch) src/core/MOM_PressureForce_Montgomery.F90:649-748
 *   calculate_density_1d with rho_ref, calculate_density_derivs    src/equation_of_state/MOM_EOS.F90:299
 *   Wright (1997) "WRIGHT" form: density_elem / density_anomaly_elem / calculate_density_derivs_elem
 *                                   src/equation_of_state/MOM_EOS_Wright.F90:80-206
 *   linear EOS                      src/equation_of_state/MOM_EOS_linear.F90:59-130
 *
 * The EOS functions are PINNED by the reference's EOS_unit_tests check values
 * (src/equation_of_state/MOM_EOS.F90:1931,1944,1996; tests/golden/eos_check_values.json); the PLM slopes
 * are pinned through oracle/remapping.c.  The assembled pressure force has no known-answer vector in the
 * reference (parity unpinned for the assembly; checked through hydrostatic-consistency properties).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "mom6_oracle.h"

static inline double max2(double a, double b) { return a > b ? a : b; }
static inline double max3(double a, double b, double c) { return max2(max2(a, b), c); }

/* ---- Wright 1997, MOM_EOS_Wright.F90:23-38 ---- */
static const double a0 = 7.057924e-4, a1 = 3.480336e-7, a2 = -1.112733e-7;
static const double b0 = 5.790749e8, b1 = 3.516535e6, b2 = -4.002714e4, b3 = 2.084372e2, b4 = 5.944068e5, b5 = -9.643486e3;
static const double c0 = 1.704853e5, c1 = 7.904722e2, c2 = -7.984422, c3 = 5.140652e-2, c4 = -2.302158e2, c5 = -3.079464;

/* density_elem_buggy_Wright :80-95 */
static double wright_density(double T, double S, double pressure) {
  double al0 = (a0 + a1*T) +a2*S;
  double p0 = (b0 + b4*S) + T * (b1 + T*(b2 + b3*T) + b5*S);
  double lambda = (c0 +c4*S) + T * (c1 + T*(c2 + c3*T) + c5*S);
  return (pressure + p0) / (lambda + al0*(pressure + p0));
}

/* density_anomaly_elem_buggy_Wright :98-129 */
static double wright_density_anomaly(double T, double S, double pressure, double rho_ref) {
  double pa_000 = (b0*(1.0 - a0*rho_ref) - rho_ref*c0);
  double al_TS = a1*T +a2*S;
  double al0 = a0 + al_TS;
  double p_TSp = pressure + (b4*S + T * (b1 + (T*(b2 + b3*T) + b5*S)));
  double lam_TS = c4*S + T * (c1 + (T*(c2 + c3*T) + c5*S));
  return (pa_000 + (p_TSp - rho_ref*(p_TSp*al0 + (b0*al_TS + lam_TS)))) /
         ( (c0 + lam_TS) + al0*(b0 + p_TSp) );
}

/* calculate_density_derivs_elem_buggy_Wright :178-206 */
static void wright_density_derivs(double T, double S, double pressure, double *drho_dT, double *drho_dS) {
  double al0 = (a0 + a1*T) + a2*S;
  double p0 = (b0 + b4*S) + T * (b1 + T*((b2 + b3*T)) + b5*S);
  double lambda = (c0 +c4*S) + T * (c1 + T*((c2 + c3*T)) + c5*S);
  double I_denom2 = 1.0 / (lambda + al0*(pressure + p0));
  I_denom2 = I_denom2 *I_denom2;
  *drho_dT = I_denom2 *
    (lambda* (b1 + T*(2.0*b2 + 3.0*b3*T) + b5*S) -
     (pressure+p0) * ( (pressure+p0)*a1 +
      (c1 + T*(c2*2.0 + c3*3.0*T) + c5*S) ));
  *drho_dS = I_denom2 * (lambda* (b4 + b5*T) -
    (pressure+p0) * ( (pressure+p0)*a2 + (c4 + c5*T) ));
}

/* ---- UNESCO (Jackett & McDougall 1995), MOM_EOS_UNESCO.F90:14-66 (coefficients), :95-167 (density), :236-296 (derivatives).
 * The reference file compiles with nothing but MOM_EOS_base_type: these are checked bit for bit against the reference's own
 * code (oracle/_ref) and against its check value (MOM_EOS.F90:1918). */
static const double R00 = 999.842594, R01 = 6.793952e-2, R02 = -9.095290e-3, R03 = 1.001685e-4, R04 = -1.120083e-6, R05 = 6.536332e-9,
    R10 = 0.824493, R11 = -4.0899e-3, R12 = 7.6438e-5, R13 = -8.2467e-7, R14 = 5.3875e-9, R60 = -5.72466e-3, R61 = 1.0227e-4,
    R62 = -1.6546e-6, R20 = 4.8314e-4,
    S000 = 1.965933e4, S010 = 1.444304e2, S020 = -1.706103, S030 = 9.648704e-3, S040 = -4.190253e-5, S100 = 52.84855,
    S110 = -3.101089e-1, S120 = 6.283263e-3, S130 = -5.084188e-5, S600 = 3.886640e-1, S610 = 9.085835e-3, S620 = -4.619924e-4,
    S001 = 3.186519, S011 = 2.212276e-2, S021 = -2.984642e-4, S031 = 1.956415e-6, S101 = 6.704388e-3, S111 = -1.847318e-4,
    S121 = 2.059331e-7, S601 = 1.480266e-4,
    S002 = 2.102898e-4, S012 = -1.202016e-5, S022 = 1.394680e-7, S102 = -2.040237e-6, S112 = 6.128773e-8, S122 = 6.207323e-10;
#define MAX0(x) ((x) > 0.0 ? (x) : 0.0)
static double unesco_density(double T, double S, double pressure) {
  const double p1 = pressure*1.0e-5, t1 = T;
  const double s1 = MAX0(S), s12 = sqrt(s1);
  const double sig0 = ( t1*(R01 + t1*(R02 + t1*(R03 + t1*(R04 + t1*R05)))) +
           s1*((R10 + t1*(R11 + t1*(R12 + t1*(R13 + t1*R14)))) +
               (s12*(R60 + t1*(R61 + t1*R62)) + s1*R20)) );
  const double rho0 = R00 + sig0;
  const double ks = (S000 + ( t1*(S010 + t1*(S020 + t1*(S030 + t1*S040))) +
                 s1*((S100 + t1*(S110 + t1*(S120 + t1*S130))) + s12*(S600 + t1*(S610 + t1*S620))) )) +
       p1*( (S001 + ( t1*(S011 + t1*(S021 + t1*S031)) +
                      s1*((S101 + t1*(S111 + t1*S121)) + s12*S601) )) +
            p1*(S002 + ( t1*(S012 + t1*S022) + s1*(S102 + t1*(S112 + t1*S122)) )) );
  return rho0*ks / (ks - p1);
}
static double unesco_density_anomaly(double T, double S, double pressure, double rho_ref) {
  const double p1 = pressure*1.0e-5, t1 = T;
  const double s1 = MAX0(S), s12 = sqrt(s1);
  const double sig0 = ( t1*(R01 + t1*(R02 + t1*(R03 + t1*(R04 + t1*R05)))) +
           s1*((R10 + t1*(R11 + t1*(R12 + t1*(R13 + t1*R14)))) +
               (s12*(R60 + t1*(R61 + t1*R62)) + s1*R20)) );
  const double ks = (S000 + ( t1*(S010 + t1*(S020 + t1*(S030 + t1*S040))) +
                 s1*((S100 + t1*(S110 + t1*(S120 + t1*S130))) + s12*(S600 + t1*(S610 + t1*S620))) )) +
       p1*( (S001 + ( t1*(S011 + t1*(S021 + t1*S031)) +
                      s1*((S101 + t1*(S111 + t1*S121)) + s12*S601) )) +
            p1*(S002 + ( t1*(S012 + t1*S022) + s1*(S102 + t1*(S112 + t1*S122)) )) );
  return ((R00 - rho_ref)*ks + (sig0*ks + p1*rho_ref)) / (ks - p1);
}
static void unesco_density_derivs(double T, double S, double pressure, double *drho_dT, double *drho_dS) {
  const double p1 = pressure*1.0e-5, t1 = T;
  const double s1 = MAX0(S), s12 = sqrt(s1);
  const double rho0 = R00 + ( t1*(R01 + t1*(R02 + t1*(R03 + t1*(R04 + t1*R05)))) +
                 s1*((R10 + t1*(R11 + t1*(R12 + t1*(R13 + t1*R14)))) +
                     (s12*(R60 + t1*(R61 + t1*R62)) + s1*R20)) );
  const double drho0_dT = R01 + ( t1*(2.0*R02 + t1*(3.0*R03 + t1*(4.0*R04 + t1*(5.0*R05)))) +
                     s1*(R11 + (t1*(2.0*R12 + t1*(3.0*R13 + t1*(4.0*R14))) +
                                s12*(R61 + t1*(2.0*R62)) )) );
  const double drho0_dS = R10 + ( t1*(R11 + t1*(R12 + t1*(R13 + t1*R14))) +
                     (1.5*(s12*(R60 + t1*(R61 + t1*R62))) + s1*(2.0*R20)) );
  const double ks = ( S000 + (t1*(S010 + t1*(S020 + t1*(S030 + t1*S040))) +
                 s1*((S100 + t1*(S110 + t1*(S120 + t1*S130))) + s12*(S600 + t1*(S610 + t1*S620)))) ) +
       p1*( (S001 + ( t1*(S011 + t1*(S021 + t1*S031)) +
                      s1*((S101 + t1*(S111 + t1*S121)) + s12*S601) )) +
            p1*(S002 + ( t1*(S012 + t1*S022) + s1*(S102 + t1*(S112 + t1*S122)) )) );
  const double dks_dT = ( S010 + (t1*(2.0*S020 + t1*(3.0*S030 + t1*(4.0*S040))) +
                     s1*((S110 + t1*(2.0*S120 + t1*(3.0*S130))) + s12*(S610 + t1*(2.0*S620)))) ) +
           p1*(((S011 + t1*(2.0*S021 + t1*(3.0*S031))) + s1*(S111 + t1*(2.0*S121)) ) +
               p1*(S012 + t1*(2.0*S022) + s1*(S112 + t1*(2.0*S122))) );
  const double dks_dS = ( S100 + (t1*(S110 + t1*(S120 + t1*S130)) + 1.5*(s12*(S600 + t1*(S610 + t1*S620)))) ) +
           p1*((S101 + t1*(S111 + t1*S121) + s12*(1.5*S601)) +
               p1*(S102 + t1*(S112 + t1*S122)) );
  const double I_denom = 1.0 / (ks - p1);
  *drho_dT = (ks*drho0_dT - dks_dT*((rho0*p1)*I_denom)) * I_denom;
  *drho_dS = (ks*drho0_dS - dks_dS*((rho0*p1)*I_denom)) * I_denom;
}

/* ---- WRIGHT_FULL and WRIGHT_REDUCED (MOM_EOS_Wright_full.F90, MOM_EOS_Wright_red.F90: one code, two sets of coefficients
 * :21-36; density :73-89, anomaly :94-119, spec_vol_anomaly :150-170, derivatives :175-200); check values MOM_EOS.F90:1923-1932 */
typedef struct { double a0, a1, a2, b0, b1, b2, b3, b4, b5, c0, c1, c2, c3, c4, c5; } wright_coefs_t;
static const wright_coefs_t W_FULL = {7.133718e-4, 2.724670e-7, -1.646582e-7, 5.613770e8, 3.600337e6, -3.727194e4, 1.660557e2, 6.844158e5, -8.389457e3, 1.609893e5, 8.427815e2, -6.931554, 3.869318e-2, -1.664201e2, -2.765195};
static const wright_coefs_t W_RED = {7.057924e-4, 3.480336e-7, -1.112733e-7, 5.790749e8, 3.516535e6, -4.002714e4, 2.084372e2, 5.944068e5, -9.643486e3, 1.704853e5, 7.904722e2, -7.984422, 5.140652e-2, -2.302158e2, -3.079464};
static double wrightx_density(const wright_coefs_t W, double T, double S, double pressure) {
  const double al0 = W.a0 + (W.a1*T + W.a2*S);
  const double p0 = W.b0 + ( W.b4*S + T * (W.b1 + (T*(W.b2 + W.b3*T) + W.b5*S)) );
  const double lambda = W.c0 + ( W.c4*S + T * (W.c1 + (T*(W.c2 + W.c3*T) + W.c5*S)) );
  return (pressure + p0) / (lambda + al0*(pressure + p0));
}
static double wrightx_density_anomaly(const wright_coefs_t W, double T, double S, double pressure, double rho_ref) {
  const double pa_000 = W.b0*(1.0 - W.a0*rho_ref) - rho_ref*W.c0;
  const double al_TS = W.a1*T + W.a2*S;
  const double al0 = W.a0 + al_TS;
  const double p_TSp = pressure + (W.b4*S + T * (W.b1 + (T*(W.b2 + W.b3*T) + W.b5*S)));
  const double lam_TS = W.c4*S + T * (W.c1 + (T*(W.c2 + W.c3*T) + W.c5*S));
  return (pa_000 + (p_TSp - rho_ref*(p_TSp*al0 + (W.b0*al_TS + lam_TS)))) / ( (W.c0 + lam_TS) + al0*(W.b0 + p_TSp) );
}
static double wrightx_spv_anomaly(const wright_coefs_t W, double T, double S, double pressure, double spv_ref) {
  const double lam_000 = W.c0 + (W.a0 - spv_ref)*W.b0;
  const double al_TS = W.a1*T + W.a2*S;
  const double p_TSp = pressure + (W.b4*S + T * (W.b1 + (T*(W.b2 + W.b3*T) + W.b5*S)));
  const double lambda = lam_000 + ( W.c4*S + T * (W.c1 + (T*(W.c2 + W.c3*T) + W.c5*S)) );
  return al_TS + (lambda + (W.a0 - spv_ref)*p_TSp) / (W.b0 + p_TSp);
}
static void wrightx_density_derivs(const wright_coefs_t W, double T, double S, double pressure, double *pDT, double *pDS) {
  double DT, DS;
  const double al0 = W.a0 + (W.a1*T + W.a2*S);
  const double p0 = W.b0 + ( W.b4*S + T * (W.b1 + (T*(W.b2 + W.b3*T) + W.b5*S)) );
  const double lambda = W.c0 + ( W.c4*S + T * (W.c1 + (T*(W.c2 + W.c3*T) + W.c5*S)) );
  const double den = (lambda + al0*(pressure + p0));
  const double I_denom2 = 1.0 / (den*den);
  DT = I_denom2 * (lambda * (W.b1 + (T*(2.0*W.b2 + 3.0*W.b3*T) + W.b5*S)) -
     (pressure+p0) * ( (pressure+p0)*W.a1 + (W.c1 + (T*(W.c2*2.0 + W.c3*3.0*T) + W.c5*S)) ));
  DS = I_denom2 * (lambda * (W.b4 + W.b5*T) -
     (pressure+p0) * ( (pressure+p0)*W.a2 + (W.c4 + W.c5*T) ));
  *pDT = DT; *pDS = DS;
}

/* calculate_density (no rho_ref) / with rho_ref / derivs for the EOS forms provided */
double orc_eos_density(const mom6hip_eos_t *E, double T, double S, double p) {
  if (E->form == MOM6HIP_EOS_LINEAR) return E->Rho_T0_S0 + E->dRho_dT*T + E->dRho_dS*S;
  if (E->form == MOM6HIP_EOS_UNESCO) return unesco_density(T, S, p);
  if (E->form == MOM6HIP_EOS_WRIGHT_FULL) return wrightx_density(W_FULL, T, S, p);
  if (E->form == MOM6HIP_EOS_WRIGHT_REDUCED) return wrightx_density(W_RED, T, S, p);
  return wright_density(T, S, p);
}
double orc_eos_density_anomaly(const mom6hip_eos_t *E, double T, double S, double p, double rho_ref) {
  if (E->form == MOM6HIP_EOS_LINEAR) return (E->Rho_T0_S0 - rho_ref) + (E->dRho_dT*T + E->dRho_dS*S);
  if (E->form == MOM6HIP_EOS_UNESCO) return unesco_density_anomaly(T, S, p, rho_ref);
  if (E->form == MOM6HIP_EOS_WRIGHT_FULL) return wrightx_density_anomaly(W_FULL, T, S, p, rho_ref);
  if (E->form == MOM6HIP_EOS_WRIGHT_REDUCED) return wrightx_density_anomaly(W_RED, T, S, p, rho_ref);
  return wright_density_anomaly(T, S, p, rho_ref);
}
void orc_eos_density_derivs(const mom6hip_eos_t *E, double T, double S, double p, double *dT, double *dS) {
  if (E->form == MOM6HIP_EOS_LINEAR) { *dT = E->dRho_dT; *dS = E->dRho_dS; return; }
  if (E->form == MOM6HIP_EOS_UNESCO) { unesco_density_derivs(T, S, p, dT, dS); return; }
  if (E->form == MOM6HIP_EOS_WRIGHT_FULL) { wrightx_density_derivs(W_FULL, T, S, p, dT, dS); return; }
  if (E->form == MOM6HIP_EOS_WRIGHT_REDUCED) { wrightx_density_derivs(W_RED, T, S, p, dT, dS); return; }
  wright_density_derivs(T, S, p, dT, dS);
}

/* ALE_PLM_edge_values, MOM_ALE.F90:1520-1579 (answer_date >= 20190101) */
void orc_ale_plm_edge_values(const mom6hip_grid_t *G, const double *h, const double *Q, int bdry_extrap,
                             double *Q_t, double *Q_b)
{
  const int nz = G->nk;
  const double h_neglect = G->H_subroundoff;
#define HH(k) h[ORC_H3(G,i,j,k)]
#define QQ(k) Q[ORC_H3(G,i,j,k)]
  ORC_PAR
  for (int j = G->jsc-1; j <= G->jec+1; j++) { double *slp = calloc(nz+2, sizeof(double)); for (int i = G->isc-1; i <= G->iec+1; i++) {
    slp[1] = 0.;
    for (int k = 2; k <= nz-1; k++)
      slp[k] = orc_plm_slope_wa(HH(k-1), HH(k), HH(k+1), h_neglect, QQ(k-1), QQ(k), QQ(k+1));
    slp[nz] = 0.;
    for (int k = 2; k <= nz-1; k++) {
      double mslp = orc_plm_monotonized_slope(QQ(k-1), QQ(k), QQ(k+1), slp[k-1], slp[k], slp[k+1]);
      Q_t[ORC_H3(G,i,j,k)] = QQ(k) - 0.5 * mslp;
      Q_b[ORC_H3(G,i,j,k)] = QQ(k) + 0.5 * mslp;
    }
    if (bdry_extrap) {
      double mslp = - orc_plm_extrapolate_slope(HH(2), HH(1), h_neglect, QQ(2), QQ(1));
      Q_t[ORC_H3(G,i,j,1)] = QQ(1) - 0.5 * mslp;
      Q_b[ORC_H3(G,i,j,1)] = QQ(1) + 0.5 * mslp;
      mslp = orc_plm_extrapolate_slope(HH(nz-1), HH(nz), h_neglect, QQ(nz-1), QQ(nz));
      Q_t[ORC_H3(G,i,j,nz)] = QQ(nz) - 0.5 * mslp;
      Q_b[ORC_H3(G,i,j,nz)] = QQ(nz) + 0.5 * mslp;
    } else {
      Q_t[ORC_H3(G,i,j,1)] = QQ(1); Q_b[ORC_H3(G,i,j,1)] = QQ(1);
      Q_t[ORC_H3(G,i,j,nz)] = QQ(nz); Q_b[ORC_H3(G,i,j,nz)] = QQ(nz);
    }
  } free(slp); }
#undef HH
#undef QQ
}

/* e(i,j,K), K = 1..nz+1 */
#define E3(i,j,K) e[ORC_H2(G,i,j) + nH2*((K)-1)]

/* int_density_dz_generic_plm, MOM_density_integrals.F90:369-769 (use_rho_ref, no Stanley terms) */
static void int_density_dz_generic_plm(const mom6hip_grid_t *G, const mom6hip_eos_t *EOS, int k, const double *T_t,
                                       const double *T_b, const double *S_t, const double *S_b, const double *e,
                                       double rho_ref, double rho_0, double G_e, double dz_subroundoff,
                                       int useMassWghtInterp, double z0pres, double *dpa, double *intz_dpa,
                                       double *intx_dpa, double *inty_dpa)
{
  const long nH2 = (long)ORC_NIH(G)*ORC_NJH(G);
  const int Isq = G->isc-1, Ieq = G->iec, Jsq = G->jsc-1, Jeq = G->jec;
  const double C1_90 = 1.0/90.0;
  const double GxRho = G_e * rho_0;
  const double massWeightToggle = useMassWghtInterp ? 1. : 0.;
  const double *bathyT = G->bathyT;
  double wt_t[6], wt_b[6];
  for (int n = 1; n <= 5; n++) { wt_t[n] = 0.25 * (double)(5-n); wt_b[n] = 1.0 - wt_t[n]; }
#define TT(i,j) T_t[ORC_H3(G,i,j,k)]
#define TB(i,j) T_b[ORC_H3(G,i,j,k)]
#define ST(i,j) S_t[ORC_H3(G,i,j,k)]
#define SB(i,j) S_b[ORC_H3(G,i,j,k)]
  const int K = k;
  /* 1. vertical integrals */
  ORC_PAR
  for (int j = Jsq; j <= Jeq+1; j++) for (int i = Isq; i <= Ieq+1; i++) {
    double dz = E3(i,j,K) - E3(i,j,K+1);
    double r5[6];
    for (int n = 1; n <= 5; n++) {
      double p5 = -GxRho*((E3(i,j,K) - z0pres) - 0.25*(double)(n-1)*dz);
      double S5 = wt_t[n] * ST(i,j) + wt_b[n] * SB(i,j);
      double T5 = wt_t[n] * TT(i,j) + wt_b[n] * TB(i,j);
      r5[n] = orc_eos_density_anomaly(EOS, T5, S5, p5, rho_ref);
    }
    double rho_anom = C1_90*(7.0*(r5[1]+r5[5]) + 32.0*(r5[2]+r5[4]) + 12.0*r5[3]);
    dpa[ORC_H2(G,i,j)] = G_e*dz*rho_anom;
    intz_dpa[ORC_H2(G,i,j)] = 0.5*G_e*(dz*dz) *
            (rho_anom - C1_90*(16.0*(r5[4]-r5[2]) + 7.0*(r5[5]-r5[1])) );
  }
  /* 2./3. horizontal integrals; dir 0: x faces (I,j), dir 1: y faces (i,J) */
  for (int dir = 0; dir < 2; dir++) {
    const int j0 = dir ? Jsq : G->jsc, j1 = dir ? Jeq : G->jec;
    const int i0 = dir ? G->isc : Isq, i1 = dir ? G->iec : Ieq;
    ORC_PAR
    for (int j = j0; j <= j1; j++) for (int i = i0; i <= i1; i++) {
      const int ip = dir ? i : i+1, jp = dir ? j+1 : j;      /* the cell on the "right" of the face */
      double Ttl, Tbl, Ttr, Tbr, Stl, Sbl, Str, Sbr;
      double hWght = massWeightToggle *
              max3(0., -bathyT[ORC_H2(G,i,j)]-E3(ip,jp,K), -bathyT[ORC_H2(G,ip,jp)]-E3(i,j,K));
      if (hWght > 0.) {
        double hL = (E3(i,j,K) - E3(i,j,K+1)) + dz_subroundoff;
        double hR = (E3(ip,jp,K) - E3(ip,jp,K+1)) + dz_subroundoff;
        double rr = (hL-hR)/(hL+hR);
        hWght = hWght * ( rr*rr );
        double iDenom = 1./( hWght*(hR + hL) + hL*hR );
        Ttl = ( (hWght*hR)*TT(ip,jp) + (hWght*hL + hR*hL)*TT(i,j) ) * iDenom;
        Ttr = ( (hWght*hL)*TT(i,j) + (hWght*hR + hR*hL)*TT(ip,jp) ) * iDenom;
        Tbl = ( (hWght*hR)*TB(ip,jp) + (hWght*hL + hR*hL)*TB(i,j) ) * iDenom;
        Tbr = ( (hWght*hL)*TB(i,j) + (hWght*hR + hR*hL)*TB(ip,jp) ) * iDenom;
        Stl = ( (hWght*hR)*ST(ip,jp) + (hWght*hL + hR*hL)*ST(i,j) ) * iDenom;
        Str = ( (hWght*hL)*ST(i,j) + (hWght*hR + hR*hL)*ST(ip,jp) ) * iDenom;
        Sbl = ( (hWght*hR)*SB(ip,jp) + (hWght*hL + hR*hL)*SB(i,j) ) * iDenom;
        Sbr = ( (hWght*hL)*SB(i,j) + (hWght*hR + hR*hL)*SB(ip,jp) ) * iDenom;
      } else {
        Ttl = TT(i,j); Tbl = TB(i,j); Ttr = TT(ip,jp); Tbr = TB(ip,jp);
        Stl = ST(i,j); Sbl = SB(i,j); Str = ST(ip,jp); Sbr = SB(ip,jp);
      }
      double intz[6];
      intz[1] = dpa[ORC_H2(G,i,j)]; intz[5] = dpa[ORC_H2(G,ip,jp)];
      for (int m = 2; m <= 4; m++) {
        double w_left = wt_t[m], w_right = wt_b[m];
        double dz_x = w_left*(E3(i,j,K) - E3(i,j,K+1)) + w_right*(E3(ip,jp,K) - E3(ip,jp,K+1));
        double T15[6], S15[6], p15[6], r15[6];
        T15[1] = w_left*Ttl + w_right*Ttr;
        T15[5] = w_left*Tbl + w_right*Tbr;
        S15[1] = w_left*Stl + w_right*Str;
        S15[5] = w_left*Sbl + w_right*Sbr;
        p15[1] = -GxRho*((w_left*E3(i,j,K) + w_right*E3(ip,jp,K)) - z0pres);
        for (int n = 2; n <= 5; n++) p15[n] = p15[n-1] + GxRho*0.25*dz_x;
        for (int n = 2; n <= 4; n++) {
          S15[n] = wt_t[n] * S15[1] + wt_b[n] * S15[5];
          T15[n] = wt_t[n] * T15[1] + wt_b[n] * T15[5];
        }
        for (int n = 1; n <= 5; n++) r15[n] = orc_eos_density_anomaly(EOS, T15[n], S15[n], p15[n], rho_ref);
        intz[m] = G_e*dz_x*( C1_90*(7.0*(r15[1]+r15[5]) + 32.0*(r15[2]+r15[4]) + 12.0*r15[3]) );
      }
      double val = C1_90*(7.0*(intz[1]+intz[5]) + 32.0*(intz[2]+intz[4]) + 12.0*intz[3]);
      if (dir) inty_dpa[ORC_V2(G,i,j)] = val; else intx_dpa[ORC_U2(G,i,j)] = val;
    }
  }
#undef TT
#undef TB
#undef ST
#undef SB
}


/* ---- the non-PLM branch: int_density_dz (MOM_density_integrals.F90:41) -> analytic_int_density_dz (MOM_EOS.F90:1308) ----
 * Tk, Sk: the layer's T and S planes (tv_tmp); z_t = e(:,:,K), z_b = e(:,:,K+1).  rho_scale = pres_scale = 1 (no rescaling). */

/* int_density_dz_linear, MOM_EOS_linear.F90:259-424 */
static void int_density_dz_linear(const mom6hip_grid_t *G, const double *Tk, const double *Sk, const double *z_t, const double *z_b,
                                  double rho_ref, double G_e, double Rho_T0_S0, double dRho_dT, double dRho_dS,
                                  double dz_neglect, int useMassWghtInterp, double *dpa, double *intz_dpa, double *intx_dpa, double *inty_dpa)
{
  const int Isq = G->isc-1, Ieq = G->iec, Jsq = G->jsc-1, Jeq = G->jec;
  const double C1_6 = 1.0/6.0, C1_90 = 1.0/90.0;
  const double *bathyT = G->bathyT;
#define T2(i,j) Tk[ORC_H2(G,i,j)]
#define S2(i,j) Sk[ORC_H2(G,i,j)]
#define ZT(i,j) z_t[ORC_H2(G,i,j)]
#define ZB(i,j) z_b[ORC_H2(G,i,j)]
  for (int j = Jsq; j <= Jeq+1; j++) for (int i = Isq; i <= Ieq+1; i++) {
    double dz = ZT(i,j) - ZB(i,j);
    double rho_anom = (Rho_T0_S0 - rho_ref) + dRho_dT*T2(i,j) + dRho_dS*S2(i,j);
    dpa[ORC_H2(G,i,j)] = G_e*rho_anom*dz;
    intz_dpa[ORC_H2(G,i,j)] = 0.5*G_e*rho_anom*(dz*dz);
  }
  for (int dir = 0; dir < 2; dir++) {
    const int j0 = dir ? Jsq : G->jsc, j1 = dir ? Jeq : G->jec;
    const int i0 = dir ? G->isc : Isq, i1 = dir ? G->iec : Ieq;
    for (int j = j0; j <= j1; j++) for (int i = i0; i <= i1; i++) {
      const int ip = dir ? i : i+1, jp = dir ? j+1 : j;
      double hWght = 0.0, val;
      if (useMassWghtInterp) hWght = max3(0., -bathyT[ORC_H2(G,i,j)]-ZT(ip,jp), -bathyT[ORC_H2(G,ip,jp)]-ZT(i,j));
      if (hWght <= 0.0) {
        double dzL = ZT(i,j) - ZB(i,j), dzR = ZT(ip,jp) - ZB(ip,jp);
        double raL = (Rho_T0_S0 - rho_ref) + (dRho_dT*T2(i,j) + dRho_dS*S2(i,j));
        double raR = (Rho_T0_S0 - rho_ref) + (dRho_dT*T2(ip,jp) + dRho_dS*S2(ip,jp));
        val = G_e*C1_6 * (dzL*(2.0*raL + raR) + dzR*(2.0*raR + raL));
      } else {
        double hL = (ZT(i,j) - ZB(i,j)) + dz_neglect;
        double hR = (ZT(ip,jp) - ZB(ip,jp)) + dz_neglect;
        double rr = (hL-hR)/(hL+hR);
        hWght = hWght * ( rr*rr );
        double iDenom = 1.0 / ( hWght*(hR + hL) + hL*hR );
        double hWt_LL = (hWght*hL + hR*hL) * iDenom, hWt_LR = (hWght*hR) * iDenom;
        double hWt_RR = (hWght*hR + hR*hL) * iDenom, hWt_RL = (hWght*hL) * iDenom;
        double intz[6];
        intz[1] = dpa[ORC_H2(G,i,j)]; intz[5] = dpa[ORC_H2(G,ip,jp)];
        for (int m = 2; m <= 4; m++) {
          double wt_L = 0.25*(double)(5-m), wt_R = 1.0-wt_L;
          double wtT_L = wt_L*hWt_LL + wt_R*hWt_RL, wtT_R = wt_L*hWt_LR + wt_R*hWt_RR;
          double dz = wt_L*(ZT(i,j) - ZB(i,j)) + wt_R*(ZT(ip,jp) - ZB(ip,jp));
          double rho_anom = (Rho_T0_S0 - rho_ref) +
                     (dRho_dT * (wtT_L*T2(i,j) + wtT_R*T2(ip,jp)) +
                      dRho_dS * (wtT_L*S2(i,j) + wtT_R*S2(ip,jp)));
          intz[m] = G_e*rho_anom*dz;
        }
        val = C1_90*(7.0*(intz[1]+intz[5]) + 32.0*(intz[2]+intz[4]) + 12.0*intz[3]);
      }
      if (dir) inty_dpa[ORC_V2(G,i,j)] = val; else intx_dpa[ORC_U2(G,i,j)] = val;
    }
  }
}

/* int_density_dz_wright, MOM_EOS_Wright.F90:389-640 (the "WRIGHT" form; no rescaling arguments) */
static void int_density_dz_wright(const mom6hip_grid_t *G, const double *Tk, const double *Sk, const double *z_t, const double *z_b,
                                  double rho_ref, double rho_0, double G_e, double dz_neglect, int useMassWghtInterp, double Z_0p,
                                  double *dpa, double *intz_dpa, double *intx_dpa, double *inty_dpa)
{
  const int Isq = G->isc-1, Ieq = G->iec, Jsq = G->jsc-1, Jeq = G->jec;
  const double C1_3 = 1.0/3.0, C1_7 = 1.0/7.0, C1_9 = 1.0/9.0, C1_90 = 1.0/90.0;
  const double *bathyT = G->bathyT;
  const long nH2 = (long)ORC_NIH(G)*ORC_NJH(G);
  double *al0_2d = calloc(nH2, 8), *p0_2d = calloc(nH2, 8), *lambda_2d = calloc(nH2, 8);
  const double GxRho = G_e * rho_0, g_Earth = G_e, Pa_to_RL2_T2 = 1.0;
  const double rho_ref_mks = rho_ref, I_Rho = 1.0 / rho_0;
  const double z0pres = Z_0p;
  for (int j = Jsq; j <= Jeq+1; j++) for (int i = Isq; i <= Ieq+1; i++) {
    const long o = ORC_H2(G,i,j);
    al0_2d[o] = (a0 + a1*T2(i,j)) + a2*S2(i,j);
    p0_2d[o] = (b0 + b4*S2(i,j)) + T2(i,j) * (b1 + T2(i,j)*((b2 + b3*T2(i,j))) + b5*S2(i,j));
    lambda_2d[o] = (c0 +c4*S2(i,j)) + T2(i,j) * (c1 + T2(i,j)*((c2 + c3*T2(i,j))) + c5*S2(i,j));
    double al0 = al0_2d[o], p0 = p0_2d[o], lambda = lambda_2d[o];
    double dz = ZT(i,j) - ZB(i,j);
    double p_ave = -GxRho*(0.5*(ZT(i,j)+ZB(i,j)) - z0pres);
    double I_al0 = 1.0 / al0;
    double I_Lzz = 1.0 / (p0 + (lambda * I_al0) + p_ave);
    double eps = 0.5*GxRho*dz*I_Lzz, eps2 = eps*eps;
    double rho_anom = (p0 + p_ave)*(I_Lzz*I_al0) - rho_ref_mks;
    double rem = I_Rho * (lambda * (I_al0*I_al0)) * eps2 *
          (C1_3 + eps2*(0.2 + eps2*(C1_7 + C1_9*eps2)));
    dpa[o] = Pa_to_RL2_T2 * (g_Earth*rho_anom*dz - 2.0*eps*rem);
    intz_dpa[o] = Pa_to_RL2_T2 * (0.5*g_Earth*rho_anom*(dz*dz) - dz*(1.0+eps)*rem);
  }
  for (int dir = 0; dir < 2; dir++) {
    const int j0 = dir ? Jsq : G->jsc, j1 = dir ? Jeq : G->jec;
    const int i0 = dir ? G->isc : Isq, i1 = dir ? G->iec : Ieq;
    for (int j = j0; j <= j1; j++) for (int i = i0; i <= i1; i++) {
      const int ip = dir ? i : i+1, jp = dir ? j+1 : j;
      const long oL = ORC_H2(G,i,j), oR = ORC_H2(G,ip,jp);
      double hWght = 0.0, hWt_LL, hWt_LR, hWt_RR, hWt_RL;
      if (useMassWghtInterp) hWght = max3(0., -bathyT[oL]-ZT(ip,jp), -bathyT[oR]-ZT(i,j));
      if (hWght > 0.) {
        double hL = (ZT(i,j) - ZB(i,j)) + dz_neglect;
        double hR = (ZT(ip,jp) - ZB(ip,jp)) + dz_neglect;
        double rr = (hL-hR)/(hL+hR);
        hWght = hWght * ( rr*rr );
        double iDenom = 1.0 / ( hWght*(hR + hL) + hL*hR );
        hWt_LL = (hWght*hL + hR*hL) * iDenom ; hWt_LR = (hWght*hR) * iDenom;
        hWt_RR = (hWght*hR + hR*hL) * iDenom ; hWt_RL = (hWght*hL) * iDenom;
      } else {
        hWt_LL = 1.0 ; hWt_LR = 0.0 ; hWt_RR = 1.0 ; hWt_RL = 0.0;
      }
      double intz[6];
      intz[1] = dpa[oL]; intz[5] = dpa[oR];
      for (int m = 2; m <= 4; m++) {
        double wt_L = 0.25*(double)(5-m), wt_R = 1.0-wt_L;
        double wtT_L = wt_L*hWt_LL + wt_R*hWt_RL, wtT_R = wt_L*hWt_LR + wt_R*hWt_RR;
        double al0 = wtT_L*al0_2d[oL] + wtT_R*al0_2d[oR];
        double p0 = wtT_L*p0_2d[oL] + wtT_R*p0_2d[oR];
        double lambda = wtT_L*lambda_2d[oL] + wtT_R*lambda_2d[oR];
        double dz = wt_L*(ZT(i,j) - ZB(i,j)) + wt_R*(ZT(ip,jp) - ZB(ip,jp));
        double p_ave = -GxRho*(0.5*(wt_L*(ZT(i,j)+ZB(i,j)) + wt_R*(ZT(ip,jp)+ZB(ip,jp))) - z0pres);
        double I_al0 = 1.0 / al0;
        double I_Lzz = 1.0 / (p0 + (lambda * I_al0) + p_ave);
        double eps = 0.5*GxRho*dz*I_Lzz, eps2 = eps*eps;
        intz[m] = Pa_to_RL2_T2 * ( g_Earth*dz*((p0 + p_ave)*(I_Lzz*I_al0) - rho_ref_mks) - 2.0*eps *
                  I_Rho * (lambda * (I_al0*I_al0)) * eps2 * (C1_3 + eps2*(0.2 + eps2*(C1_7 + C1_9*eps2))) );
      }
      double val = C1_90*(7.0*(intz[1]+intz[5]) + 32.0*(intz[2]+intz[4]) + 12.0*intz[3]);
      if (dir) inty_dpa[ORC_V2(G,i,j)] = val; else intx_dpa[ORC_U2(G,i,j)] = val;
    }
  }
  free(al0_2d); free(p0_2d); free(lambda_2d);
#undef T2
#undef S2
#undef ZT
#undef ZB
}

/* PressureForce_FV_Bouss, MOM_PressureForce_FV.F90:462-919.  EOS == NULL: no equation of state (use_EOS false). */
int orc_pressureforce_fv_bouss(const mom6hip_grid_t *G, const mom6hip_pressureforce_cs_t *CS, const mom6hip_eos_t *EOS,
                               const double *h, const double *T, const double *S, const double *p_atm,
                               double *PFu, double *PFv, double *pbce, double *eta)
{
  const int is = G->isc, ie = G->iec, js = G->jsc, je = G->jec, nz = G->nk;
  const int Isq = G->isc-1, Ieq = G->iec, Jsq = G->jsc-1, Jeq = G->jec;
  const long nH2 = (long)ORC_NIH(G)*ORC_NJH(G), nH3 = nH2*nz;
  const int use_EOS = (EOS != NULL);
  const int nkmb = CS->nkmb;
  int use_ALE = 0;
  if (CS->use_ALE) use_ALE = CS->reconstruct && use_EOS;        /* :561-562 */
  if (CS->GFS_scale != 1.0) return 1;
  if (use_ALE && CS->Recon_Scheme != 1) return 1;
  if (use_EOS && !use_ALE && !(EOS->form == MOM6HIP_EOS_LINEAR || EOS->form == MOM6HIP_EOS_WRIGHT)) return 1;
  if ((nkmb > 0 || !use_EOS) && !CS->Rlay) return 1;
  if (!use_EOS && pbce && !CS->g_prime) return 1;
  const double h_neglect = G->H_subroundoff;
  const double dz_neglect = G->dZ_subroundoff;
  const double I_Rho0 = 1.0 / G->Rho0;
  const double G_Rho0 = G->g_Earth / G->Rho0;
  const double rho_ref = CS->Rho0;
  const double Z_ref = CS->Z_ref;
  double *e = calloc(nH2*(nz+1), 8);
  double *T_t = calloc(nH3, 8), *T_b = calloc(nH3, 8), *S_t = calloc(nH3, 8), *S_b = calloc(nH3, 8);
  double *T_tmp = NULL, *S_tmp = NULL;
  const double *Tt = T, *St = S;                  /* tv_tmp%T, tv_tmp%S */
  double *pa = calloc(nH2, 8), *dpa = calloc(nH2, 8), *intz_dpa = calloc(nH2, 8), *dz_geo = calloc(nH2, 8);
  double *intx_pa = calloc((size_t)(ORC_NIH(G)+1)*ORC_NJH(G), 8), *intx_dpa = calloc((size_t)(ORC_NIH(G)+1)*ORC_NJH(G), 8);
  double *inty_pa = calloc((size_t)ORC_NIH(G)*(ORC_NJH(G)+1), 8), *inty_dpa = calloc((size_t)ORC_NIH(G)*(ORC_NJH(G)+1), 8);

  /* :571-573 / :639-641 (no SAL, no tides) */
  for (int j = Jsq; j <= Jeq+1; j++) for (int i = Isq; i <= Ieq+1; i++)
    E3(i,j,nz+1) = -G->bathyT[ORC_H2(G,i,j)];
  /* :646-648 */
  ORC_PAR
  for (int j = Jsq; j <= Jeq+1; j++) for (int k = nz; k >= 1; k--) for (int i = Isq; i <= Ieq+1; i++)
    E3(i,j,k) = E3(i,j,k+1) + h[ORC_H3(G,i,j,k)]*G->H_to_Z;

  /* :650-680: with a bulk mixed layer, layers lighter than the buffer layer take its properties */
  if (use_EOS && nkmb > 0) {
    T_tmp = calloc(nH3, 8); S_tmp = calloc(nH3, 8);
    for (int j = Jsq; j <= Jeq+1; j++) {
      for (int k = 1; k <= nkmb; k++) for (int i = Isq; i <= Ieq+1; i++) {
        T_tmp[ORC_H3(G,i,j,k)] = T[ORC_H3(G,i,j,k)]; S_tmp[ORC_H3(G,i,j,k)] = S[ORC_H3(G,i,j,k)];
      }
      for (int i = Isq; i <= Ieq+1; i++) {
        const double Rho_cv_BL = orc_eos_density(EOS, T[ORC_H3(G,i,j,nkmb)], S[ORC_H3(G,i,j,nkmb)], CS->P_Ref);
        for (int k = nkmb+1; k <= nz; k++) {
          if (CS->Rlay[k-1] < Rho_cv_BL) {
            T_tmp[ORC_H3(G,i,j,k)] = T[ORC_H3(G,i,j,nkmb)]; S_tmp[ORC_H3(G,i,j,k)] = S[ORC_H3(G,i,j,nkmb)];
          } else {
            T_tmp[ORC_H3(G,i,j,k)] = T[ORC_H3(G,i,j,k)]; S_tmp[ORC_H3(G,i,j,k)] = S[ORC_H3(G,i,j,k)];
          }
        }
      }
    }
    Tt = T_tmp; St = S_tmp;
  }

  /* :712-718 */
  if (use_ALE) {
    orc_ale_plm_edge_values(G, h, S, CS->boundary_extrap, S_t, S_b);
    orc_ale_plm_edge_values(G, h, T, CS->boundary_extrap, T_t, T_b);
  }

  /* :723-741 */
  for (int j = Jsq; j <= Jeq+1; j++) for (int i = Isq; i <= Ieq+1; i++) {
    if (p_atm) pa[ORC_H2(G,i,j)] = (rho_ref*G->g_Earth)*(E3(i,j,1) - Z_ref) + p_atm[ORC_H2(G,i,j)];
    else       pa[ORC_H2(G,i,j)] = (rho_ref*G->g_Earth)*(E3(i,j,1) - Z_ref);
  }
  for (int j = js; j <= je; j++) for (int I = Isq; I <= Ieq; I++)
    intx_pa[ORC_U2(G,I,j)] = 0.5*(pa[ORC_H2(G,I,j)] + pa[ORC_H2(G,I+1,j)]);
  for (int J = Jsq; J <= Jeq; J++) for (int i = is; i <= ie; i++)
    inty_pa[ORC_V2(G,i,J)] = 0.5*(pa[ORC_H2(G,i,J)] + pa[ORC_H2(G,i,J+1)]);

  for (int k = 1; k <= nz; k++) {
#define HK(i,j) h[ORC_H3(G,i,j,k)]
    if (use_EOS) {
      if (use_ALE) {                 /* :753-758 */
        int_density_dz_generic_plm(G, EOS, k, T_t, T_b, S_t, S_b, e, rho_ref, CS->Rho0, G->g_Earth, dz_neglect,
                                   CS->useMassWghtInterp, Z_ref, dpa, intz_dpa, intx_dpa, inty_dpa);
      } else if (EOS->form == MOM6HIP_EOS_LINEAR) {       /* :765-768 -> MOM_EOS.F90:1364-1377 */
        int_density_dz_linear(G, Tt + nH2*(k-1), St + nH2*(k-1), e + nH2*(k-1), e + nH2*k, rho_ref, G->g_Earth,
                              EOS->Rho_T0_S0, EOS->dRho_dT, EOS->dRho_dS, dz_neglect, CS->useMassWghtInterp,
                              dpa, intz_dpa, intx_dpa, inty_dpa);
      } else {                                            /* MOM_EOS.F90:1378-1390 */
        int_density_dz_wright(G, Tt + nH2*(k-1), St + nH2*(k-1), e + nH2*(k-1), e + nH2*k, rho_ref, CS->Rho0, G->g_Earth,
                              dz_neglect, CS->useMassWghtInterp, Z_ref, dpa, intz_dpa, intx_dpa, inty_dpa);
      }
      for (int j = Jsq; j <= Jeq+1; j++) for (int i = Isq; i <= Ieq+1; i++)
        intz_dpa[ORC_H2(G,i,j)] = intz_dpa[ORC_H2(G,i,j)]*G->Z_to_H;
    } else {                         /* :775-789 */
      const double Rl = CS->Rlay[k-1];
      for (int j = Jsq; j <= Jeq+1; j++) for (int i = Isq; i <= Ieq+1; i++) {
        dz_geo[ORC_H2(G,i,j)] = G->g_Earth * G->H_to_Z*HK(i,j);
        dpa[ORC_H2(G,i,j)] = (Rl - rho_ref) * dz_geo[ORC_H2(G,i,j)];
        intz_dpa[ORC_H2(G,i,j)] = 0.5*(Rl - rho_ref) * dz_geo[ORC_H2(G,i,j)]*HK(i,j);
      }
      for (int j = js; j <= je; j++) for (int I = Isq; I <= Ieq; I++)
        intx_dpa[ORC_U2(G,I,j)] = 0.5*(Rl - rho_ref) * (dz_geo[ORC_H2(G,I,j)] + dz_geo[ORC_H2(G,I+1,j)]);
      for (int J = Jsq; J <= Jeq; J++) for (int i = is; i <= ie; i++)
        inty_dpa[ORC_V2(G,i,J)] = 0.5*(Rl - rho_ref) * (dz_geo[ORC_H2(G,i,J)] + dz_geo[ORC_H2(G,i,J+1)]);
    }
    /* :793-801 */
    ORC_PAR
    for (int j = js; j <= je; j++) for (int I = Isq; I <= Ieq; I++) {
      const int i = I;
      PFu[ORC_U3(G,I,j,k)] = (((pa[ORC_H2(G,i,j)]*HK(i,j) + intz_dpa[ORC_H2(G,i,j)]) -
                   (pa[ORC_H2(G,i+1,j)]*HK(i+1,j) + intz_dpa[ORC_H2(G,i+1,j)])) +
                   ((HK(i+1,j) - HK(i,j)) * intx_pa[ORC_U2(G,I,j)] -
                   (E3(i+1,j,k+1) - E3(i,j,k+1)) * intx_dpa[ORC_U2(G,I,j)] * G->Z_to_H)) *
                   ((2.0*I_Rho0*G->IdxCu[ORC_U2(G,I,j)]) /
                   ((HK(i,j) + HK(i+1,j)) + h_neglect));
      intx_pa[ORC_U2(G,I,j)] = intx_pa[ORC_U2(G,I,j)] + intx_dpa[ORC_U2(G,I,j)];
    }
    /* :804-812 */
    ORC_PAR
    for (int J = Jsq; J <= Jeq; J++) for (int i = is; i <= ie; i++) {
      const int j = J;
      PFv[ORC_V3(G,i,J,k)] = (((pa[ORC_H2(G,i,j)]*HK(i,j) + intz_dpa[ORC_H2(G,i,j)]) -
                   (pa[ORC_H2(G,i,j+1)]*HK(i,j+1) + intz_dpa[ORC_H2(G,i,j+1)])) +
                   ((HK(i,j+1) - HK(i,j)) * inty_pa[ORC_V2(G,i,J)] -
                   (E3(i,j+1,k+1) - E3(i,j,k+1)) * inty_dpa[ORC_V2(G,i,J)] * G->Z_to_H)) *
                   ((2.0*I_Rho0*G->IdyCv[ORC_V2(G,i,J)]) /
                   ((HK(i,j) + HK(i,j+1)) + h_neglect));
      inty_pa[ORC_V2(G,i,J)] = inty_pa[ORC_V2(G,i,J)] + inty_dpa[ORC_V2(G,i,J)];
    }
#undef HK
    for (int j = Jsq; j <= Jeq+1; j++) for (int i = Isq; i <= Ieq+1; i++)
      pa[ORC_H2(G,i,j)] = pa[ORC_H2(G,i,j)] + dpa[ORC_H2(G,i,j)];
  }

  /* Set_pbce_Bouss(e, tv_tmp, ...), MOM_PressureForce_Montgomery.F90:649-748 (no rho_star) */
  if (pbce && use_EOS) {
    const double Rho0xG = CS->Rho0 * G->g_Earth;
    ORC_PAR
    for (int j = Jsq; j <= Jeq+1; j++) for (int i = Isq; i <= Ieq+1; i++) {
      double Ihtot = G->H_to_Z / ((E3(i,j,1)-E3(i,j,nz+1)) + dz_neglect);
      double press = -Rho0xG*(E3(i,j,1) - Z_ref);
      double rho_in_situ = orc_eos_density(EOS, Tt[ORC_H3(G,i,j,1)], St[ORC_H3(G,i,j,1)], press);
      pbce[ORC_H3(G,i,j,1)] = G_Rho0*(CS->GFS_scale * rho_in_situ) * G->H_to_Z;
      for (int k = 2; k <= nz; k++) {
        press = -Rho0xG*(E3(i,j,k) - Z_ref);
        double T_int = 0.5*(Tt[ORC_H3(G,i,j,k-1)]+Tt[ORC_H3(G,i,j,k)]);
        double S_int = 0.5*(St[ORC_H3(G,i,j,k-1)]+St[ORC_H3(G,i,j,k)]);
        double dR_dT, dR_dS;
        orc_eos_density_derivs(EOS, T_int, S_int, press, &dR_dT, &dR_dS);
        pbce[ORC_H3(G,i,j,k)] = pbce[ORC_H3(G,i,j,k-1)] + G_Rho0 *
               ((E3(i,j,k) - E3(i,j,nz+1)) * Ihtot) *
               (dR_dT*(Tt[ORC_H3(G,i,j,k)]-Tt[ORC_H3(G,i,j,k-1)]) +
                dR_dS*(St[ORC_H3(G,i,j,k)]-St[ORC_H3(G,i,j,k-1)]));
      }
    }
  } else if (pbce) {                 /* :733-745 */
    for (int j = Jsq; j <= Jeq+1; j++) for (int i = Isq; i <= Ieq+1; i++) {
      double Ihtot = 1.0 / ((E3(i,j,1)-E3(i,j,nz+1)) + dz_neglect);
      pbce[ORC_H3(G,i,j,1)] = CS->g_prime[0] * G->H_to_Z;
      for (int k = 2; k <= nz; k++)
        pbce[ORC_H3(G,i,j,k)] = pbce[ORC_H3(G,i,j,k-1)] +
                      (CS->g_prime[k-1]*G->H_to_Z) * ((E3(i,j,k) - E3(i,j,nz+1)) * Ihtot);
    }
  }
  /* :839-843 */
  if (eta) for (int j = Jsq; j <= Jeq+1; j++) for (int i = Isq; i <= Ieq+1; i++)
    eta[ORC_H2(G,i,j)] = E3(i,j,1)*G->Z_to_H;

  free(e); free(T_t); free(T_b); free(S_t); free(S_b); free(pa); free(dpa); free(intz_dpa); free(dz_geo);
  free(intx_pa); free(intx_dpa); free(inty_pa); free(inty_dpa); free(T_tmp); free(S_tmp);
  return 0;
}

/* ======== non-Boussinesq: specific-volume forms and PressureForce_FV_nonBouss ================================================
 * spec_vol_anomaly_elem of MOM_EOS_Wright.F90:156-174, MOM_EOS_UNESCO.F90:208-240, MOM_EOS_linear.F90:102-113 (calculate_spec_vol
 * with spv_ref, MOM_EOS.F90; the Wright array form :910-933 calls the same function). */
static double wright_spv_anomaly(double T, double S, double pressure, double spv_ref) {
  const double al0 = (a0 + a1*T) + a2*S;
  const double p0 = (b0 + b4*S) + T * (b1 + T*((b2 + b3*T)) + b5*S);
  const double lambda = (c0 + c4*S) + T * (c1 + T*((c2 + c3*T)) + c5*S);
  return (lambda + (al0 - spv_ref)*(pressure + p0)) / (pressure + p0);
}
static double unesco_spv_anomaly(double T, double S, double pressure, double spv_ref) {
  const double p1 = pressure*1.0e-5, t1 = T;
  const double s1 = MAX0(S), s12 = sqrt(s1);
  const double rho0 = R00 + ( t1*(R01 + t1*(R02 + t1*(R03 + t1*(R04 + t1*R05)))) +
                 s1*((R10 + t1*(R11 + t1*(R12 + t1*(R13 + t1*R14)))) +
                     (s12*(R60 + t1*(R61 + t1*R62)) + s1*R20)) );
  const double ks = (S000 + ( t1*(S010 + t1*(S020 + t1*(S030 + t1*S040))) +
                 s1*((S100 + t1*(S110 + t1*(S120 + t1*S130))) + s12*(S600 + t1*(S610 + t1*S620))) )) +
       p1*( (S001 + ( t1*(S011 + t1*(S021 + t1*S031)) +
                      s1*((S101 + t1*(S111 + t1*S121)) + s12*S601) )) +
            p1*(S002 + ( t1*(S012 + t1*S022) + s1*(S102 + t1*(S112 + t1*S122)) )) );
  return (ks*(1.0 - (rho0*spv_ref)) - p1) / (rho0*ks);
}
double orc_eos_spec_vol_anomaly(const mom6hip_eos_t *E, double T, double S, double p, double spv_ref) {
  if (E->form == MOM6HIP_EOS_LINEAR)
    return ((1.0 - E->Rho_T0_S0*spv_ref) - spv_ref*(E->dRho_dT*T + E->dRho_dS*S)) / (E->Rho_T0_S0 + (E->dRho_dT*T + E->dRho_dS*S));
  if (E->form == MOM6HIP_EOS_UNESCO) return unesco_spv_anomaly(T, S, p, spv_ref);
  if (E->form == MOM6HIP_EOS_WRIGHT_FULL) return wrightx_spv_anomaly(W_FULL, T, S, p, spv_ref);
  if (E->form == MOM6HIP_EOS_WRIGHT_REDUCED) return wrightx_spv_anomaly(W_RED, T, S, p, spv_ref);
  return wright_spv_anomaly(T, S, p, spv_ref);
}

/* int_spec_vol_dp_generic_plm, MOM_density_integrals.F90:1479-1726, for layer k (1-based); p has nz+1 planes */
static void int_spec_vol_dp_generic_plm(const mom6hip_grid_t *G, const mom6hip_eos_t *EOS, int k, const double *T_t, const double *T_b,
                                        const double *S_t, const double *S_b, const double *p, double alpha_ref, double dP_neglect,
                                        int massw, double *dza, double *intp_dza, double *intx_dza, double *inty_dza) {
  const int nz = G->nk;
  const int Isq = G->isc-1, Ieq = G->iec, Jsq = G->jsc-1, Jeq = G->jec;
  const long nH2 = (long)ORC_NIH(G)*ORC_NJH(G);
  const double C1_90 = 1.0/90.0;
  double wt_t[6], wt_b[6];
  for (int n = 1; n <= 5; n++) { wt_t[n] = 0.25 * (double)(n-1); wt_b[n] = 1.0 - wt_t[n]; }      /* reversed from int_density_dz :1549 */
#define PT(i,j) p[ORC_H2(G,i,j) + nH2*(k-1)]
#define PB(i,j) p[ORC_H2(G,i,j) + nH2*k]
#define BP(i,j) p[ORC_H2(G,i,j) + nH2*nz]
#define Q3(a,i,j) a[ORC_H3(G,i,j,k)]
  ORC_PAR
  for (int j = Jsq; j <= Jeq+1; j++) for (int i = Isq; i <= Ieq+1; i++) {
    double a5[6];
    for (int n = 1; n <= 5; n++) {
      const double p5 = wt_t[n] * PT(i,j) + wt_b[n] * PB(i,j);
      const double S5 = wt_t[n] * Q3(S_t,i,j) + wt_b[n] * Q3(S_b,i,j);
      const double T5 = wt_t[n] * Q3(T_t,i,j) + wt_b[n] * Q3(T_b,i,j);
      a5[n] = orc_eos_spec_vol_anomaly(EOS, T5, S5, p5, alpha_ref);
    }
    const double dp = PB(i,j) - PT(i,j);
    const double alpha_anom = C1_90*((7.0*(a5[1]+a5[5]) + 32.0*(a5[2]+a5[4])) + 12.0*a5[3]);
    dza[ORC_H2(G,i,j)] = dp*alpha_anom;
    intp_dza[ORC_H2(G,i,j)] = 0.5*(dp*dp) * (alpha_anom - C1_90*(16.0*(a5[4]-a5[2]) + 7.0*(a5[5]-a5[1])));
  }
  for (int dir = 0; dir < 2; dir++) {
    const int j0 = dir ? Jsq : G->jsc, j1 = dir ? Jeq : G->jec, i0 = dir ? G->isc : Isq, i1 = dir ? G->iec : Ieq;
    ORC_PAR
    for (int j = j0; j <= j1; j++) for (int i = i0; i <= i1; i++) {
      const int ir = dir ? i : i+1, jr = dir ? j+1 : j;
      double hWght = 0.0, hWt_LL, hWt_LR, hWt_RR, hWt_RL;
      if (massw) hWght = max3(0., BP(i,j)-PT(ir,jr), BP(ir,jr)-PT(i,j));
      if (hWght > 0.) {
        const double hL = (PB(i,j) - PT(i,j)) + dP_neglect;
        const double hR = (PB(ir,jr) - PT(ir,jr)) + dP_neglect;
        const double rr = (hL-hR)/(hL+hR);
        hWght = hWght * (rr*rr);
        const double iDenom = 1.0 / ( hWght*(hR + hL) + hL*hR );
        hWt_LL = (hWght*hL + hR*hL) * iDenom ; hWt_LR = (hWght*hR) * iDenom;
        hWt_RR = (hWght*hR + hR*hL) * iDenom ; hWt_RL = (hWght*hL) * iDenom;
      } else {
        hWt_LL = 1.0 ; hWt_LR = 0.0 ; hWt_RR = 1.0 ; hWt_RL = 0.0;
      }
      double intp[6];
      intp[1] = dza[ORC_H2(G,i,j)]; intp[5] = dza[ORC_H2(G,ir,jr)];
      for (int m = 2; m <= 4; m++) {
        const double wt_L = 0.25*(double)(5-m), wt_R = 1.0-wt_L;
        const double wtT_L = wt_L*hWt_LL + wt_R*hWt_RL, wtT_R = wt_L*hWt_LR + wt_R*hWt_RR;
        const double P_top = wt_L*PT(i,j) + wt_R*PT(ir,jr);
        const double P_bot = wt_L*PB(i,j) + wt_R*PB(ir,jr);
        const double T_top = wtT_L*Q3(T_t,i,j) + wtT_R*Q3(T_t,ir,jr);
        const double T_bot = wtT_L*Q3(T_b,i,j) + wtT_R*Q3(T_b,ir,jr);
        const double S_top = wtT_L*Q3(S_t,i,j) + wtT_R*Q3(S_t,ir,jr);
        const double S_bot = wtT_L*Q3(S_b,i,j) + wtT_R*Q3(S_b,ir,jr);
        const double dp_90 = C1_90*(P_bot - P_top);
        double a15[6];
        for (int n = 1; n <= 5; n++)
          a15[n] = orc_eos_spec_vol_anomaly(EOS, wt_t[n] * T_top + wt_b[n] * T_bot, wt_t[n] * S_top + wt_b[n] * S_bot,
                                            wt_t[n] * P_top + wt_b[n] * P_bot, alpha_ref);
        intp[m] = dp_90*((7.0*(a15[1]+a15[5]) + 32.0*(a15[2]+a15[4])) + 12.0*a15[3]);
      }
      const double v = C1_90*((7.0*(intp[1]+intp[5]) + 32.0*(intp[2]+intp[4])) + 12.0*intp[3]);
      if (dir) inty_dza[ORC_V2(G,i,j)] = v; else intx_dza[ORC_U2(G,i,j)] = v;
    }
  }
#undef PT
#undef PB
#undef BP
#undef Q3
}

/* PressureForce_FV_nonBouss, MOM_PressureForce_FV.F90:89-452 (use_EOS, ALE PLM reconstruction, no tides / SAL, GFS_scale = 1,
 * nkmb = 0) with Set_pbce_nonBouss, MOM_PressureForce_Montgomery.F90:752-861 (the use_EOS branch without alpha_star).
 * H_to_RZ = GV%H_to_RZ (1 when thicknesses are in kg m-2).  PARITY UNPINNED as the Boussinesq assembly is; the specific-volume
 * form of UNESCO is checked against the reference's own (oracle/_ref). */
int orc_pressureforce_fv_nonbouss(const mom6hip_grid_t *G, const mom6hip_pressureforce_cs_t *CS, const mom6hip_eos_t *EOS,
                                  const double *h, const double *T, const double *S, const double *p_atm, double H_to_RZ,
                                  double *PFu, double *PFv, double *pbce, double *eta)
{
  const int is = G->isc, ie = G->iec, js = G->jsc, je = G->jec, nz = G->nk;
  const int Isq = G->isc-1, Ieq = G->iec, Jsq = G->jsc-1, Jeq = G->jec;
  const long nH2 = (long)ORC_NIH(G)*ORC_NJH(G), nH3 = nH2*nz;
  if (!(CS->reconstruct && CS->Recon_Scheme == 1) || CS->GFS_scale != 1.0) return 1;
  const double H_to_RL2_T2 = G->g_Earth*H_to_RZ;
  const double dp_neglect = G->g_Earth*H_to_RZ * G->H_subroundoff;
  const double alpha_ref = 1.0 / CS->Rho0;
  double *p = calloc(nH2*(nz+1), 8);
  double *T_t = calloc(nH3, 8), *T_b = calloc(nH3, 8), *S_t = calloc(nH3, 8), *S_b = calloc(nH3, 8);
  double *dza = calloc(nH3, 8), *intp_dza = calloc(nH3, 8);
  double *intx_dza = calloc((size_t)(ORC_NIH(G)+1)*ORC_NJH(G)*nz, 8), *inty_dza = calloc((size_t)ORC_NIH(G)*(ORC_NJH(G)+1)*nz, 8);
  double *za = calloc(nH2, 8), *dp = calloc(nH2, 8);
  double *intx_za = calloc((size_t)(ORC_NIH(G)+1)*ORC_NJH(G), 8), *inty_za = calloc((size_t)ORC_NIH(G)*(ORC_NJH(G)+1), 8);
  const long nU2 = (long)(ORC_NIH(G)+1)*ORC_NJH(G), nV2 = (long)ORC_NIH(G)*(ORC_NJH(G)+1);
#define P3(i,j,K) p[ORC_H2(G,i,j) + nH2*((K)-1)]
  /* :193-206 */
  for (int j = Jsq; j <= Jeq+1; j++) for (int i = Isq; i <= Ieq+1; i++) P3(i,j,1) = p_atm ? p_atm[ORC_H2(G,i,j)] : 0.0;
  for (int j = Jsq; j <= Jeq+1; j++) for (int K = 2; K <= nz+1; K++) for (int i = Isq; i <= Ieq+1; i++)
    P3(i,j,K) = P3(i,j,K-1) + H_to_RL2_T2 * h[ORC_H3(G,i,j,K-1)];
  /* :245 TS_PLM_edge_values */
  orc_ale_plm_edge_values(G, h, S, CS->boundary_extrap, S_t, S_b);
  orc_ale_plm_edge_values(G, h, T, CS->boundary_extrap, T_t, T_b);
  /* :255-262 */
  for (int k = 1; k <= nz; k++)
    int_spec_vol_dp_generic_plm(G, EOS, k, T_t, T_b, S_t, S_b, p, alpha_ref, dp_neglect, CS->useMassWghtInterp,
                                dza + nH2*(k-1), intp_dza + nH2*(k-1), intx_dza + nU2*(k-1), inty_dza + nV2*(k-1));
  /* :299-306: the geopotential anomaly at the sea surface, summed from the bottom */
  for (int j = Jsq; j <= Jeq+1; j++) {
    for (int i = Isq; i <= Ieq+1; i++) za[ORC_H2(G,i,j)] = alpha_ref*P3(i,j,nz+1) - G->g_Earth*G->bathyT[ORC_H2(G,i,j)];
    for (int k = nz; k >= 1; k--) for (int i = Isq; i <= Ieq+1; i++)
      za[ORC_H2(G,i,j)] = za[ORC_H2(G,i,j)] + dza[ORC_H2(G,i,j) + nH2*(k-1)];
  }
  /* :366-371 */
  for (int j = js; j <= je; j++) for (int I = Isq; I <= Ieq; I++) intx_za[ORC_U2(G,I,j)] = 0.5*(za[ORC_H2(G,I,j)] + za[ORC_H2(G,I+1,j)]);
  for (int J = Jsq; J <= Jeq; J++) for (int i = is; i <= ie; i++) inty_za[ORC_V2(G,i,J)] = 0.5*(za[ORC_H2(G,i,J)] + za[ORC_H2(G,i,J+1)]);
  /* :373-410 */
  for (int k = 1; k <= nz; k++) {
    for (int j = Jsq; j <= Jeq+1; j++) for (int i = Isq; i <= Ieq+1; i++) {
      dp[ORC_H2(G,i,j)] = H_to_RL2_T2 * h[ORC_H3(G,i,j,k)];
      za[ORC_H2(G,i,j)] = za[ORC_H2(G,i,j)] - dza[ORC_H2(G,i,j) + nH2*(k-1)];
    }
#define ZA(i,j) za[ORC_H2(G,i,j)]
#define DP(i,j) dp[ORC_H2(G,i,j)]
#define IPD(i,j) intp_dza[ORC_H2(G,i,j) + nH2*(k-1)]
    ORC_PAR
    for (int j = js; j <= je; j++) for (int I = Isq; I <= Ieq; I++) {
      const int i = I;
      intx_za[ORC_U2(G,I,j)] = intx_za[ORC_U2(G,I,j)] - intx_dza[ORC_U2(G,I,j) + nU2*(k-1)];
      PFu[ORC_U3(G,I,j,k)] = ( ((ZA(i,j)*DP(i,j) + IPD(i,j)) - (ZA(i+1,j)*DP(i+1,j) + IPD(i+1,j))) +
                               ((DP(i+1,j) - DP(i,j)) * intx_za[ORC_U2(G,I,j)] -
                                (P3(i+1,j,k) - P3(i,j,k)) * intx_dza[ORC_U2(G,I,j) + nU2*(k-1)]) ) *
                             (2.0*G->IdxCu[ORC_U2(G,I,j)] / ((DP(i,j) + DP(i+1,j)) + dp_neglect));
    }
    ORC_PAR
    for (int J = Jsq; J <= Jeq; J++) for (int i = is; i <= ie; i++) {
      const int j = J;
      inty_za[ORC_V2(G,i,J)] = inty_za[ORC_V2(G,i,J)] - inty_dza[ORC_V2(G,i,J) + nV2*(k-1)];
      PFv[ORC_V3(G,i,J,k)] = (((ZA(i,j)*DP(i,j) + IPD(i,j)) - (ZA(i,j+1)*DP(i,j+1) + IPD(i,j+1))) +
                              ((DP(i,j+1) - DP(i,j)) * inty_za[ORC_V2(G,i,J)] -
                               (P3(i,j+1,k) - P3(i,j,k)) * inty_dza[ORC_V2(G,i,J) + nV2*(k-1)])) *
                             (2.0*G->IdyCv[ORC_V2(G,i,J)] / ((DP(i,j) + DP(i,j+1)) + dp_neglect));
    }
#undef ZA
#undef DP
#undef IPD
  }
  /* Set_pbce_nonBouss :794-831 */
  if (pbce) {
    const double dP_dH = G->g_Earth * H_to_RZ;
    ORC_PAR
    for (int j = Jsq; j <= Jeq+1; j++) for (int i = Isq; i <= Ieq+1; i++) {
      const long n2 = ORC_H2(G,i,j);
      const double C_htot = dP_dH / ((P3(i,j,nz+1)-P3(i,j,1)) + dp_neglect);
      double pb = dP_dH / orc_eos_density(EOS, T[ORC_H3(G,i,j,nz)], S[ORC_H3(G,i,j,nz)], P3(i,j,nz+1));
      pbce[n2 + nH2*(nz-1)] = pb;
      for (int k = nz-1; k >= 1; k--) {
        const double Tk = T[ORC_H3(G,i,j,k)], Tk1 = T[ORC_H3(G,i,j,k+1)], Sk = S[ORC_H3(G,i,j,k)], Sk1 = S[ORC_H3(G,i,j,k+1)];
        const double T_int = 0.5*(Tk+Tk1), S_int = 0.5*(Sk+Sk1);
        const double rho = orc_eos_density(EOS, T_int, S_int, P3(i,j,k+1));
        double dR_dT, dR_dS;
        orc_eos_density_derivs(EOS, T_int, S_int, P3(i,j,k+1), &dR_dT, &dR_dS);
        pb = pb + ((P3(i,j,k+1)-P3(i,j,1))*C_htot) * ((dR_dT*(Tk1-Tk) + dR_dS*(Sk1-Sk)) / (rho*rho));
        pbce[n2 + nH2*(k-1)] = pb;
      }
    }
  }
  /* :417-428 */
  if (eta) {
    const double Pa_to_H = 1.0 / (G->g_Earth * H_to_RZ);
    for (int j = Jsq; j <= Jeq+1; j++) for (int i = Isq; i <= Ieq+1; i++)
      eta[ORC_H2(G,i,j)] = p_atm ? (P3(i,j,nz+1) - p_atm[ORC_H2(G,i,j)])*Pa_to_H : P3(i,j,nz+1)*Pa_to_H;
  }
#undef P3
  free(p); free(T_t); free(T_b); free(S_t); free(S_b); free(dza); free(intp_dza); free(intx_dza); free(inty_dza); free(za); free(dp);
  free(intx_za); free(inty_za);
  return 0;
}
