/*
 * remapping.c -- CPU restatement of the ALE column reconstruction + remapping (TEST INFRASTRUCTURE).
 *
 * Restates, with the same evaluation order, the latest-vintage (answer_date >= 20190101) branches of
 *   PCM_reconstruction              src/ALE/PCM_functions.F90:18-37
 *   PLM_slope_wa, PLM_monotonized_slope, PLM_extrapolate_slope, PLM_reconstruction,
 *   PLM_boundary_extrapolation      src/ALE/PLM_functions.F90:22-307
 *   bound_edge_values, check_discontinuous_edge_values, edge_values_explicit_h4, end_value_h4
 *                                   src/ALE/regrid_edge_values.F90:44-110,141-159,222-363,658-771
 *   PPM_reconstruction, PPM_limiter_standard, PPM_boundary_extrapolation
 *                                   src/ALE/PPM_functions.F90:28-316
 *   buildGridFromH, remapping_core_h, remapping_core_w, build_reconstructions_1d (PCM, PLM, PPM_H4),
 *   remap_via_sub_cells, average_value_ppoly, dzFromH1H2
 *                                   src/ALE/MOM_remapping.F90:145-386,463-852,998-1099,1235-1256
 *
 * PINNED by the reference's own known-answer vectors (remapping_unit_tests,
 * src/ALE/MOM_remapping.F90:1339-1569; fixtures in tests/golden/remapping_unit_tests.json) and, for the
 * PCM/PLM routines, bit-for-bit against the reference source compiled unmodified (oracle/_ref).
 *
 * Array convention: E(k,1:2) and coef(k,1:deg+1) are stored Fortran-style, E[(side)*n + k], k 0-based.
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
static inline double fsign(double a, double b) { return copysign(fabs(a), b); }

#define E_(a,k,s)  (a)[(size_t)(s)*n + (k)]     /* k 0-based, s = 0 (left/top) or 1 (right/bottom) */
#define C_(a,k,d)  (a)[(size_t)(d)*n + (k)]     /* polynomial coefficient d of cell k */

/* ---- PCM ---------------------------------------------------------------------------------- */
void orc_pcm_reconstruction(int n, const double *u, double *E, double *coef)
{
  for (int k = 0; k < n; k++) { C_(coef,k,0) = u[k]; E_(E,k,0) = u[k]; E_(E,k,1) = u[k]; }
}

/* ---- PLM ---------------------------------------------------------------------------------- */
/* PLM_slope_wa, PLM_functions.F90:22-65 */
double orc_plm_slope_wa(double h_l, double h_c, double h_r, double h_neglect, double u_l, double u_c, double u_r)
{
  double sigma_r = u_r - u_c;
  double sigma_l = u_c - u_l;
  double sigma_c = 2.0 * ( u_r - u_l ) * ( h_c / ( h_l + 2.0*h_c + h_r + h_neglect) );
  double u_min = min3( u_l, u_c, u_r );
  double u_max = max3( u_l, u_c, u_r );
  double slope;
  if ( (sigma_l * sigma_r) > 0.0 ) {
    slope = fsign( min2( fabs(sigma_c), 2.*min2( u_c - u_min, u_max - u_c ) ), sigma_c );
  } else {
    slope = 0.0;
  }
  if (u_c - 0.5*fabs(slope) < u_min || u_c + 0.5*fabs(slope) > u_max) {
    slope = slope * ( 1. - DBL_EPSILON );
  }
  if (fabs(slope) < 1.E-140) slope = 0.;
  return slope;
}

/* PLM_monotonized_slope, PLM_functions.F90:124-159 */
double orc_plm_monotonized_slope(double u_l, double u_c, double u_r, double s_l, double s_c, double s_r)
{
  double almost_two = 2. * ( 1. - DBL_EPSILON );
  double e_r = u_l + 0.5*s_l;
  double e_l = u_r - 0.5*s_r;
  double slp = fabs(s_c);
  double edge = u_c - 0.5 * s_c;
  if ( ( edge - e_r ) * ( u_c - edge ) < 0. ) {
    edge = 0.5 * ( edge + e_r );
    slp = min2( slp, fabs( edge - u_c ) * almost_two );
  }
  edge = u_c + 0.5 * s_c;
  if ( ( edge - u_c ) * ( e_l - edge ) < 0. ) {
    edge = 0.5 * ( edge + e_l );
    slp = min2( slp, fabs( edge - u_c ) * almost_two );
  }
  return fsign( slp, s_c );
}

/* PLM_extrapolate_slope, PLM_functions.F90:164-183 */
double orc_plm_extrapolate_slope(double h_l, double h_c, double h_neglect, double u_l, double u_c)
{
  double hl = h_l + h_neglect;
  double hc = h_c + h_neglect;
  double left_edge = (u_l*hc + u_c*hl) / (hl + hc);
  return 2.0 * ( u_c - left_edge );
}

/* PLM_reconstruction, PLM_functions.F90:190-260 */
void orc_plm_reconstruction(int n, const double *h, const double *u, double *E, double *coef, double h_neglect);
/* ncol columns of n layers, arrays [ncol][..]: the batch form tools/calibrate_ref.py times against the reference build */
void orc_plm_batch(int ncol, int n, const double *h, const double *u, double *E, double *coef, double h_neglect)
{
  for (int c = 0; c < ncol; c++)
    orc_plm_reconstruction(n, h + (size_t)c*n, u + (size_t)c*n, E + (size_t)c*2*n, coef + (size_t)c*3*n, h_neglect);
}

void orc_plm_reconstruction(int n, const double *h, const double *u, double *E, double *coef, double h_neglect)
{
  double almost_one = 1. - DBL_EPSILON;
  double *slp = calloc(n, sizeof(double)), *mslp = calloc(n, sizeof(double));
  for (int k = 1; k < n-1; k++)
    slp[k] = orc_plm_slope_wa(h[k-1], h[k], h[k+1], h_neglect, u[k-1], u[k], u[k+1]);
  slp[0] = 0.; slp[n-1] = 0.;
  for (int k = 1; k < n-1; k++)
    mslp[k] = orc_plm_monotonized_slope( u[k-1], u[k], u[k+1], slp[k-1], slp[k], slp[k+1] );
  mslp[0] = 0.; mslp[n-1] = 0.;

  E_(E,0,0) = u[0]; E_(E,0,1) = u[0]; C_(coef,0,0) = u[0]; C_(coef,0,1) = 0.;
  for (int k = 1; k < n-1; k++) {
    double slope = mslp[k];
    double u_l = u[k] - 0.5 * slope;
    double u_r = u[k] + 0.5 * slope;
    E_(E,k,0) = u_l; E_(E,k,1) = u_r;
    C_(coef,k,0) = u_l;
    C_(coef,k,1) = ( u_r - u_l );
    double edge = C_(coef,k,1) + C_(coef,k,0);
    double e_r = u[k+1] - 0.5 * fsign( mslp[k+1], slp[k+1] );
    if ( (edge-u[k])*(e_r-edge) < 0.) C_(coef,k,1) = C_(coef,k,1) * almost_one;
  }
  E_(E,n-1,0) = u[n-1]; E_(E,n-1,1) = u[n-1]; C_(coef,n-1,0) = u[n-1]; C_(coef,n-1,1) = 0.;
  free(slp); free(mslp);
}

/* PLM_boundary_extrapolation, PLM_functions.F90:272-307 */
void orc_plm_boundary_extrapolation(int n, const double *h, const double *u, double *E, double *coef, double h_neglect)
{
  double slope = - orc_plm_extrapolate_slope( h[1], h[0], h_neglect, u[1], u[0] );
  E_(E,0,0) = u[0] - 0.5 * slope;
  E_(E,0,1) = u[0] + 0.5 * slope;
  C_(coef,0,0) = E_(E,0,0);
  C_(coef,0,1) = E_(E,0,1) - E_(E,0,0);
  slope = orc_plm_extrapolate_slope( h[n-2], h[n-1], h_neglect, u[n-2], u[n-1] );
  E_(E,n-1,0) = u[n-1] - 0.5 * slope;
  E_(E,n-1,1) = u[n-1] + 0.5 * slope;
  C_(coef,n-1,0) = E_(E,n-1,0);
  C_(coef,n-1,1) = E_(E,n-1,1) - E_(E,n-1,0);
}

/* ---- edge values ---------------------------------------------------------------------------- */
static const double hMinFrac = 1.e-5;   /* regrid_edge_values.F90:30 */

/* bound_edge_values (answer_date >= 20190101), regrid_edge_values.F90:44-110 */
void orc_bound_edge_values(int n, const double *h, const double *u, double *E)
{
  for (int k = 0; k < n; k++) {
    int km1 = (k-1 > 0) ? k-1 : 0, kp1 = (k+1 < n-1) ? k+1 : n-1;
    double slope_x_h = 0.0;
    if ( ((h[km1] + h[kp1]) + 2.0*h[k]) > 0.0 ) {
      double sigma_l = ( u[k] - u[km1] );
      double sigma_c = ( u[kp1] - u[km1] ) * ( h[k] / ((h[km1] + h[kp1]) + 2.0*h[k]) );
      double sigma_r = ( u[kp1] - u[k] );
      if ( (sigma_l * sigma_r) > 0.0 )
        slope_x_h = fsign( min3(fabs(sigma_l),fabs(sigma_c),fabs(sigma_r)), sigma_c );
    }
    if ( (u[km1]-E_(E,k,0)) * (E_(E,k,0)-u[k]) < 0.0 )
      E_(E,k,0) = u[k] - fsign( min2( fabs(slope_x_h), fabs(E_(E,k,0)-u[k]) ), slope_x_h );
    if ( (u[kp1]-E_(E,k,1)) * (E_(E,k,1)-u[k]) < 0.0 )
      E_(E,k,1) = u[k] + fsign( min2( fabs(slope_x_h), fabs(E_(E,k,1)-u[k]) ), slope_x_h );
    E_(E,k,0) = max2( min2( E_(E,k,0), max2(u[km1], u[k]) ), min2(u[km1], u[k]) );
    E_(E,k,1) = max2( min2( E_(E,k,1), max2(u[kp1], u[k]) ), min2(u[kp1], u[k]) );
  }
}

/* check_discontinuous_edge_values, regrid_edge_values.F90:141-159 */
void orc_check_discontinuous_edge_values(int n, const double *u, double *E)
{
  for (int k = 0; k < n-1; k++) {
    if ( (E_(E,k+1,0) - E_(E,k,1)) * (u[k+1] - u[k]) < 0.0 ) {
      double u0_avg = 0.5 * ( E_(E,k,1) + E_(E,k+1,0) );
      u0_avg = max2( min2( u0_avg, max2(u[k], u[k+1]) ), min2(u[k], u[k+1]) );
      E_(E,k,1) = u0_avg;
      E_(E,k+1,0) = u0_avg;
    }
  }
}

/* end_value_h4, regrid_edge_values.F90:658-771 */
void orc_end_value_h4(const double dz[4], const double u[4], double Csys[4])
{
  const double min_frac = 1.0e-6;
  double Wt[3][4];   /* Wt[j-1][n-1] = Wt(j,n) */
  double h1 = dz[0], h2 = dz[1], h3 = dz[2], h4 = dz[3];
  if ((h2+h3) < min_frac*h1) h3 = min_frac*h1 - h2;
  if ((h3+h4) < min_frac*h1) h4 = min_frac*h1 - h3;
  double h12 = h1+h2, h23 = h2+h3, h34 = h3+h4;
  double h123 = h12 + h3, h234 = h2 + h34, h1234 = h12 + h34;
  double I_denB3 = 1.0 / (h123 * h12 * h23);
  double I_h12 = (h123 * h23) * I_denB3;
  double I_h23 = (h12 * h123) * I_denB3;
  double I_h123 = (h12 * h23) * I_denB3;
  double I_denom = 1.0 / ( h1234 * (h234 * h34) );
  double I_h234 = (h1234 * h34) * I_denom;
  double I_h1234 = (h234 * h34) * I_denom;

  Wt[0][0] = -h1 * (I_h1234 + I_h123 + I_h12);
  Wt[1][0] =  h1 * h12 * ( I_h234 * I_h1234 + I_h23 * (I_h234 + I_h123) );
  Wt[2][0] = -h1 * h12 * h123 * I_denom;

  Wt[0][1] =  2.0 * (I_h12*(1.0 + (h1+h12) * (I_h1234 + I_h123)) + h1 * I_h1234*I_h123);
  Wt[1][1] = -2.0 * ((h1 * h12 * I_h1234) *       (I_h23 * (I_h234 + I_h123)) +
                     (h1+h12) * ( I_h1234*I_h234 + I_h23 * (I_h234 + I_h123) ) );
  Wt[2][1] =  2.0 * ((h1+h12) * h123 + h1*h12 ) * I_denom;

  Wt[0][2] = -3.0 * I_h12 * I_h123* ( 1.0 + I_h1234 * ((h1+h12)+h123) );
  Wt[1][2] =  3.0 * I_h23 * ( I_h123 + I_h1234 * ((h1+h12)+h123) * (I_h123 + I_h234) );
  Wt[2][2] = -3.0 * ((h1+h12)+h123) * I_denom;

  Wt[0][3] =  4.0 * I_h1234 * I_h123 * I_h12;
  Wt[1][3] = -4.0 * I_h1234 * (I_h23 * (I_h123 + I_h234));
  Wt[2][3] =  4.0 * I_denom;

  Csys[0] = ((u[0] + Wt[0][0] * (u[1]-u[0])) + Wt[1][0] * (u[2]-u[1])) + Wt[2][0] * (u[3]-u[2]);
  Csys[1] = (Wt[0][1] * (u[1]-u[0]) + Wt[1][1] * (u[2]-u[1])) + Wt[2][1] * (u[3]-u[2]);
  Csys[2] = (Wt[0][2] * (u[1]-u[0]) + Wt[1][2] * (u[2]-u[1])) + Wt[2][2] * (u[3]-u[2]);
  Csys[3] = (Wt[0][3] * (u[1]-u[0]) + Wt[1][3] * (u[2]-u[1])) + Wt[2][3] * (u[3]-u[2]);
}

/* edge_values_explicit_h4 (answer_date >= 20190101), regrid_edge_values.F90:222-363.  n >= 4. */
void orc_edge_values_explicit_h4(int n, const double *h, const double *u, double *E, double h_neglect)
{
  const double hNeglect = h_neglect;
  for (int i = 2; i <= n-2; i++) {            /* Fortran i = 3..N-1 */
    double h0 = h[i-2], h1 = h[i-1], h2 = h[i], h3 = h[i+1];
    if (h0+h1==0.0 || h1+h2==0.0 || h2+h3==0.0) {
      double h_min = hMinFrac*max2( hNeglect, (h0+h1)+(h2+h3) );
      h0 = max2( h_min, h[i-2] );
      h1 = max2( h_min, h[i-1] );
      h2 = max2( h_min, h[i] );
      h3 = max2( h_min, h[i+1] );
    }
    double I_h12 = 1.0 / (h1+h2);
    double I_den_et2 = 1.0 / ( ((h0+h1)+h2)*(h0+h1) ); double I_h012 = (h0+h1) * I_den_et2;
    double I_den_et3 = 1.0 / ( (h1+(h2+h3))*(h2+h3) ); double I_h123 = (h2+h3) * I_den_et3;

    double et1 = ( 1.0 + (h1 * I_h012 + (h0+h1) * I_h123) ) * I_h12 * (h2*(h2+h3)) * u[i-1] +
                 ( 1.0 + (h2 * I_h123 + (h2+h3) * I_h012) ) * I_h12 * (h1*(h0+h1)) * u[i];
    double et2 = ( h1 * (h2*(h2+h3)) * I_den_et2 ) * (u[i-1]-u[i-2]);
    double et3 = ( h2 * (h1*(h0+h1)) * I_den_et3 ) * (u[i] - u[i+1]);
    E_(E,i,0) = (et1 + (et2 + et3)) / ((h0 + h1) + (h2 + h3));
    E_(E,i-1,1) = E_(E,i,0);
  }
  double dz[4], ut[4], C[4];
  for (int i = 0; i < 4; i++) { dz[i] = max2(hNeglect, h[i]); ut[i] = u[i]; }
  orc_end_value_h4(dz, ut, C);
  E_(E,0,0) = C[0];
  E_(E,0,1) = C[0] + dz[0]*(C[1] + dz[0]*(C[2] + dz[0]*C[3]));
  E_(E,1,0) = E_(E,0,1);
  for (int i = 0; i < 4; i++) { dz[i] = max2(hNeglect, h[n-1-i]); ut[i] = u[n-1-i]; }
  orc_end_value_h4(dz, ut, C);
  E_(E,n-1,1) = C[0];
  E_(E,n-1,0) = C[0] + dz[0]*(C[1] + dz[0]*(C[2] + dz[0]*C[3]));
  E_(E,n-2,1) = E_(E,n-1,0);
}

/* solve_diag_dominant_tridiag, regrid_solvers.F90:246-280 (Al, Ac, Au, R -> X; all of size n) */
void orc_solve_diag_dominant_tridiag(const double *Al, const double *Ac, const double *Au, const double *R, double *X, int n)
{
  double *c1 = (double*)malloc(sizeof(double)*n);
  double I_pivot = 1.0 / (Ac[0] + Au[0]);
  double d1 = Ac[0] * I_pivot;
  c1[0] = Au[0] * I_pivot;
  X[0] = R[0] * I_pivot;
  for (int k = 1; k < n-1; k++) {
    const double denom_t1 = Ac[k] + d1 * Al[k];
    I_pivot = 1.0 / (denom_t1 + Au[k]);
    d1 = denom_t1 * I_pivot;
    c1[k] = Au[k] * I_pivot;
    X[k] = (R[k] - Al[k] * X[k-1]) * I_pivot;
  }
  I_pivot = 1.0 / (Ac[n-1] + d1 * Al[n-1]);
  X[n-1] = (R[n-1] - Al[n-1] * X[n-2]) * I_pivot;
  for (int k = n-2; k >= 0; k--) X[k] = X[k] - c1[k] * X[k+1];
  free(c1);
}

/* edge_values_implicit_h4 (answer_date >= 20190101), regrid_edge_values.F90:491-654.  n >= 4. */
void orc_edge_values_implicit_h4(int n, const double *h, const double *u, double *E, double h_neglect)
{
  const double hNeglect = h_neglect;
  const int m = n + 1;
  double *tri_l = (double*)calloc(5*(size_t)m, sizeof(double));
  double *tri_c = tri_l + m, *tri_u = tri_c + m, *tri_b = tri_u + m, *tri_x = tri_b + m;
  for (int i = 0; i < n-1; i++) {            /* Fortran i = 1..N-1; row i+1 */
    double h0 = max2(h[i], hNeglect);
    double h1 = max2(h[i+1], hNeglect);
    if (fabs(h0) < 1.0e-12*fabs(h1)) h0 = 1.0e-12*h1;
    if (fabs(h1) < 1.0e-12*fabs(h0)) h1 = 1.0e-12*h0;
    const double I_h2 = 1.0 / ((h0 + h1)*(h0 + h1));
    const double alpha = (h1 * h1) * I_h2;
    const double beta = (h0 * h0) * I_h2;
    const double abmix = (h0 * h1) * I_h2;
    const double a = 2.0 * alpha * ( alpha + 2.0 * beta + 3.0 * abmix );
    const double b = 2.0 * beta * ( beta + 2.0 * alpha + 3.0 * abmix );
    tri_c[i+1] = 2.0*abmix;
    tri_l[i+1] = alpha;
    tri_u[i+1] = beta;
    tri_b[i+1] = a * u[i] + b * u[i+1];
  }
  double dz[4], ut[4], C[4];
  for (int i = 0; i < 4; i++) { dz[i] = max2(hNeglect, h[i]); ut[i] = u[i]; }
  orc_end_value_h4(dz, ut, C);
  tri_b[0] = C[0]; tri_c[0] = 1.0; tri_u[0] = 0.0;
  for (int i = 0; i < 4; i++) { dz[i] = max2(hNeglect, h[n-1-i]); ut[i] = u[n-1-i]; }
  orc_end_value_h4(dz, ut, C);
  tri_b[n] = C[0]; tri_c[n] = 1.0; tri_l[n] = 0.0;
  orc_solve_diag_dominant_tridiag(tri_l, tri_c, tri_u, tri_b, tri_x, m);
  E_(E,0,0) = tri_x[0];
  for (int i = 1; i < n; i++) { E_(E,i,0) = tri_x[i]; E_(E,i-1,1) = tri_x[i]; }
  E_(E,n-1,1) = tri_x[n];
  free(tri_l);
}

/* edge_values_explicit_h4cw, regrid_edge_values.F90:381-470 (Colella & Woodward 1984, after hybgen_ppm_coefs).  n >= 4. */
void orc_edge_values_explicit_h4cw(int n, const double *h, const double *u, double *E, double h_neglect)
{
  const double hNeglect = h_neglect;
  /* 1-based work arrays as in the reference: index k of the Fortran is [k] here */
  const size_t m = (size_t)n + 2;
  double *dp = (double*)calloc(10*m, sizeof(double));
  double *au = dp + m, *al = au + m, *ar = al + m, *h112 = ar + m, *h122 = h112 + m, *I_h12 = h122 + m, *h2_h123 = I_h12 + m,
         *I_h0123 = h2_h123 + m, *h01_h112 = I_h0123 + m;
  double *h23_h122 = (double*)calloc(m, sizeof(double));
#define U1(k) u[(k)-1]
  for (int k = 1; k <= n; k++) dp[k] = max2(h[k-1], hNeglect);
  for (int k = 2; k <= n; k++) {
    h112[k] = 2.*dp[k-1] + dp[k];
    h122[k] = dp[k-1] + 2.*dp[k];
    I_h12[k] = 1.0 / (dp[k-1] + dp[k]);
  }
  for (int k = 2; k <= n-1; k++) h2_h123[k] = dp[k] / (dp[k] + (dp[k-1]+dp[k+1]));
  for (int k = 3; k <= n-1; k++) {
    I_h0123[k] = 1.0 / ((dp[k-2] + dp[k-1]) + (dp[k] + dp[k+1]));
    h01_h112[k] = (dp[k-2] + dp[k-1]) / (2.0*dp[k-1] + dp[k]);
    h23_h122[k] = (dp[k] + dp[k+1])   / (dp[k-1] + 2.0*dp[k]);
  }
  au[1] = 0.;
  for (int k = 2; k <= n-1; k++) {
    const double slk = U1(k)-U1(k-1);
    const double srk = U1(k+1)-U1(k);
    if (slk*srk > 0.) {
      const double sck = h2_h123[k]*( h112[k]*srk*I_h12[k+1] + h122[k+1]*slk*I_h12[k] );
      au[k] = fsign(min3(fabs(2.0*slk), fabs(sck), fabs(2.0*srk)), sck);
    } else {
      au[k] = 0.;
    }
  }
  au[n] = 0.;
  al[1] = U1(1); ar[1] = U1(1); al[2] = U1(1);
  for (int k = 3; k <= n-1; k++) {
    al[k] = (dp[k]*U1(k-1) + dp[k-1]*U1(k)) * I_h12[k]
          + I_h0123[k]*( 2.*dp[k]*dp[k-1]*I_h12[k]*(U1(k)-U1(k-1)) *
                         ( h01_h112[k] - h23_h122[k] )
                  + (dp[k]*au[k-1]*h23_h122[k] - dp[k-1]*au[k]*h01_h112[k]) );
    ar[k-1] = al[k];
  }
  ar[n-1] = U1(n); al[n] = U1(n); ar[n] = U1(n);
  for (int k = 1; k <= n; k++) { E_(E,k-1,0) = al[k]; E_(E,k-1,1) = ar[k]; }
#undef U1
  free(dp); free(h23_h122);
}

/* PPM_monotonicity, PPM_functions.F90:132-158 */
void orc_ppm_monotonicity(int n, const double *u, double *E)
{
  for (int k = 1; k < n-1; k++) {
    if ((u[k+1]-u[k])*(u[k]-u[k-1]) <= 0.) {
      E_(E,k,0) = u[k];
      E_(E,k,1) = u[k];
    } else {
      const double da = E_(E,k,1)-E_(E,k,0);
      const double a6 = 6.0*u[k] - 3.0*(E_(E,k,0)+E_(E,k,1));
      if (da*a6 > da*da) E_(E,k,0) = 3.0*u[k] - 2.0*E_(E,k,1);
      else if (da*a6 < -da*da) E_(E,k,1) = 3.0*u[k] - 2.0*E_(E,k,0);
    }
  }
}

/* ---- PPM ---------------------------------------------------------------------------------- */
/* PPM_limiter_standard, PPM_functions.F90:62-128 */
void orc_ppm_limiter_standard(int n, const double *h, const double *u, double *E)
{
  orc_bound_edge_values(n, h, u, E);
  orc_check_discontinuous_edge_values(n, u, E);
  for (int k = 1; k < n-1; k++) {
    double u_l = u[k-1], u_c = u[k], u_r = u[k+1];
    double edge_l = E_(E,k,0), edge_r = E_(E,k,1);
    if ( (u_r - u_c)*(u_c - u_l) <= 0.0) {
      edge_l = u_c; edge_r = u_c;
    } else {
      double expr1 = 3.0 * (edge_r - edge_l) * ( (u_c - edge_l) + (u_c - edge_r));
      double expr2 = (edge_r - edge_l) * (edge_r - edge_l);
      if ( expr1 > expr2 ) {
        edge_l = u_c + 2.0 * ( u_c - edge_r );
        edge_l = max2( min2( edge_l, max2(u_l, u_c) ), min2(u_l, u_c) );
      } else if ( expr1 < -expr2 ) {
        edge_r = u_c + 2.0 * ( u_c - edge_l );
        edge_r = max2( min2( edge_r, max2(u_r, u_c) ), min2(u_r, u_c) );
      }
    }
    if ( fabs( edge_r - edge_l ) < max2(1.e-60, DBL_EPSILON*fabs(u_c)) ) {
      edge_l = u_c; edge_r = u_c;
    }
    E_(E,k,0) = edge_l; E_(E,k,1) = edge_r;
  }
  E_(E,0,0) = u[0]; E_(E,0,1) = u[0];
  E_(E,n-1,0) = u[n-1]; E_(E,n-1,1) = u[n-1];
}

/* PPM_reconstruction, PPM_functions.F90:28-57 */
void orc_ppm_reconstruction(int n, const double *h, const double *u, double *E, double *coef)
{
  orc_ppm_limiter_standard(n, h, u, E);
  for (int k = 0; k < n; k++) {
    double edge_l = E_(E,k,0), edge_r = E_(E,k,1);
    C_(coef,k,0) = edge_l;
    C_(coef,k,1) = 4.0 * ( u[k] - edge_l ) + 2.0 * ( u[k] - edge_r );
    C_(coef,k,2) = 3.0 * ( ( edge_r - u[k] ) + ( edge_l - u[k] ) );
  }
}

/* PPM_boundary_extrapolation, PPM_functions.F90:162-316 */
void orc_ppm_boundary_extrapolation(int n, const double *h, const double *u, double *E, double *coef, double h_neglect)
{
  const double hNeglect = h_neglect;
  /* left boundary */
  int i0 = 0, i1 = 1;
  double h0 = h[i0], h1 = h[i1], u0 = u[i0], u1 = u[i1];
  double b = C_(coef,i1,1);
  double u1_r = b *((h0+hNeglect)/(h1+hNeglect));
  double slope = 2.0 * ( u1 - u0 );
  if ( fabs(u1_r) > fabs(slope) ) u1_r = slope;
  double u0_r = E_(E,i1,0);
  double u0_l = 3.0 * u0 + 0.5 * u1_r - 2.0 * u0_r;
  double exp1 = (u0_r - u0_l) * (u0 - 0.5*(u0_l+u0_r));
  double exp2 = (u0_r - u0_l) * (u0_r - u0_l) / 6.0;
  if ( exp1 > exp2 ) u0_l = 3.0 * u0 - 2.0 * u0_r;
  if ( exp1 < -exp2 ) u0_r = 3.0 * u0 - 2.0 * u0_l;
  E_(E,i0,0) = u0_l; E_(E,i0,1) = u0_r;
  C_(coef,i0,0) = u0_l;
  C_(coef,i0,1) = 6.0 * u0 - 4.0 * u0_l - 2.0 * u0_r;
  C_(coef,i0,2) = 3.0 * ( u0_r + u0_l - 2.0 * u0 );

  /* right boundary */
  i0 = n-2; i1 = n-1;
  h0 = h[i0]; h1 = h[i1]; u0 = u[i0]; u1 = u[i1];
  b = C_(coef,i0,1);
  double c = C_(coef,i0,2);
  double u1_l = (b + 2*c);
  u1_l = u1_l * ((h1+hNeglect)/(h0+hNeglect));
  slope = 2.0 * ( u1 - u0 );
  if ( fabs(u1_l) > fabs(slope) ) u1_l = slope;
  u0_l = E_(E,i0,1);
  u0_r = 3.0 * u1 - 0.5 * u1_l - 2.0 * u0_l;
  exp1 = (u0_r - u0_l) * (u1 - 0.5*(u0_l+u0_r));
  exp2 = (u0_r - u0_l) * (u0_r - u0_l) / 6.0;
  if ( exp1 > exp2 ) u0_l = 3.0 * u1 - 2.0 * u0_r;
  if ( exp1 < -exp2 ) u0_r = 3.0 * u1 - 2.0 * u0_l;
  E_(E,i1,0) = u0_l; E_(E,i1,1) = u0_r;
  C_(coef,i1,0) = u0_l;
  C_(coef,i1,1) = 6.0 * u1 - 4.0 * u0_l - 2.0 * u0_r;
  C_(coef,i1,2) = 3.0 * ( u0_r + u0_l - 2.0 * u1 );
}

/* ---- PQM (piecewise quartic method) -------------------------------------------------------- */
/* edge_slopes_implicit_h3 (answer_date >= 20190101), regrid_edge_values.F90:803-972.  S(k,1:2) stored like E.  n >= 4. */
void orc_edge_slopes_implicit_h3(int n, const double *h, const double *u, double *S, double h_neglect)
{
  const double hNeglect = h_neglect;
  const int m = n + 1;
  double *tri_l = (double*)calloc(5*(size_t)m, sizeof(double));
  double *tri_c = tri_l + m, *tri_u = tri_c + m, *tri_b = tri_u + m, *tri_x = tri_b + m;
  for (int i = 0; i < n-1; i++) {            /* Fortran i = 1..N-1; row i+1 */
    double h0 = max2(h[i], hNeglect);
    double h1 = max2(h[i+1], hNeglect);
    const double I_h = 1.0 / (h0 + h1);
    h0 = h0 * I_h; h1 = h1 * I_h;
    const double h0h1 = h0 * h1, h0_2 = h0 * h0, h1_2 = h1 * h1;
    const double h0_3 = h0_2 * h0, h1_3 = h1_2 * h1;
    const double I_d = 1.0 / (4.0 * h0h1 * ( h0 + h1 ) + h1_3 + h0_3);
    tri_l[i+1] = (h1 * ((h0_2 + h0h1) - h1_2)) * I_d;
    tri_c[i+1] = 2.0 * ((h0_2 + h1_2) * (h0 + h1)) * I_d;
    tri_u[i+1] = (h0 * ((h1_2 + h0h1) - h0_2)) * I_d;
    tri_b[i+1] = 12.0 * (h0h1 * I_d) * ((u[i+1] - u[i]) * I_h);
  }
  double dz[4], ut[4], C[4];
  for (int i = 0; i < 4; i++) { dz[i] = max2(hNeglect, h[i]); ut[i] = u[i]; }
  orc_end_value_h4(dz, ut, C);
  tri_b[0] = C[1]; tri_c[0] = 1.0; tri_u[0] = 0.0;
  for (int i = 0; i < 4; i++) { dz[i] = max2(hNeglect, h[n-1-i]); ut[i] = u[n-1-i]; }
  orc_end_value_h4(dz, ut, C);
  tri_b[n] = -C[1]; tri_c[n] = 1.0; tri_l[n] = 0.0;
  orc_solve_diag_dominant_tridiag(tri_l, tri_c, tri_u, tri_b, tri_x, m);
  for (int i = 1; i < n; i++) { E_(S,i,0) = tri_x[i]; E_(S,i-1,1) = tri_x[i]; }
  E_(S,0,0) = tri_x[0];
  E_(S,n-1,1) = tri_x[n];
  free(tri_l);
}

/* The quartic of a cell from its edge values, edge slopes and mean (PQM_functions.F90:51-55, :150-154, :585-589, ...) */
static inline void pqm_quartic(double um, double hc, double u0_l, double u0_r, double u1_l, double u1_r,
                               double *a, double *b, double *c, double *d, double *e)
{
  *a = u0_l;
  *b = hc * u1_l;
  *c = 30.0 * um - 12.0*u0_r - 18.0*u0_l + 1.5*hc*(u1_r - 3.0*u1_l);
  *d = -60.0 * um + hc *(6.0*u1_l - 4.0*u1_r) + 28.0*u0_r + 32.0*u0_l;
  *e = 30.0 * um + 2.5*hc*(u1_r - u1_l) - 15.0*(u0_l + u0_r);
}
/* 4.0 * e * (x**3) + 3.0 * d * (x**2) + 2.0 * c * x + b, as the Fortran associates it */
static inline double pqm_gradient(double b, double c, double d, double e, double x)
{
  return 4.0 * e * ((x*x)*x) + 3.0 * d * (x*x) + 2.0 * c * x + b;
}

/* PQM_limiter, PQM_functions.F90:75-337 (answer_date >= 20190101 in bound_edge_values) */
void orc_pqm_limiter(int n, const double *h, const double *u, double *E, double *S, double h_neglect)
{
  const double hNeglect = h_neglect;
  orc_bound_edge_values(n, h, u, E);
  orc_check_discontinuous_edge_values(n, u, E);
  for (int k = 1; k < n-1; k++) {
    int inflexion_l = 0, inflexion_r = 0;
    double u0_l = E_(E,k,0), u0_r = E_(E,k,1), u1_l = E_(S,k,0), u1_r = E_(S,k,1);
    const double h_l = h[k-1], h_c = h[k], h_r = h[k+1];
    const double u_l = u[k-1], u_c = u[k], u_r = u[k+1];
    const double sigma_l = 2.0 * ( u_c - u_l ) / ( h_c + hNeglect );
    const double sigma_c = 2.0 * ( u_r - u_l ) / ( h_l + 2.0*h_c + h_r + hNeglect );
    const double sigma_r = 2.0 * ( u_r - u_c ) / ( h_c + hNeglect );
    double slope;
    if ( (sigma_l * sigma_r) > 0.0 ) slope = fsign( min3(fabs(sigma_l),fabs(sigma_c),fabs(sigma_r)), sigma_c );
    else slope = 0.0;
    if ( u1_l*slope <= 0.0 ) u1_l = slope;
    if ( u1_r*slope <= 0.0 ) u1_r = slope;
    if ( (u0_r - u_c) * (u_c - u0_l) <= 0.0) {      /* local extremum: flatten */
      u0_l = u_c; u0_r = u_c; u1_l = 0.0; u1_r = 0.0;
      inflexion_l = -1; inflexion_r = -1;
    }
    if ( (inflexion_l == 0) && (inflexion_r == 0) ) {
      double a, b, c, d, e;
      pqm_quartic(u[k], h_c, u0_l, u0_r, u1_l, u1_r, &a, &b, &c, &d, &e);
      const double alpha1 = 6*e, alpha2 = 3*d, alpha3 = c;
      const double rho = alpha2 * alpha2 - 4.0 * alpha1 * alpha3;
      int bad = 0;
      if (( alpha1 != 0.0 ) && ( rho >= 0.0 )) {
        const double sqrt_rho = sqrt( rho );
        const double x1 = 0.5 * ( - alpha2 - sqrt_rho ) / alpha1;
        const double x2 = 0.5 * ( - alpha2 + sqrt_rho ) / alpha1;
        const int in1 = (x1 >= 0.0) && (x1 <= 1.0), in2 = (x2 >= 0.0) && (x2 <= 1.0);
        if (in1 && in2) {
          const double gradient1 = pqm_gradient(b, c, d, e, x1), gradient2 = pqm_gradient(b, c, d, e, x2);
          if ( (gradient1 * slope < 0.0) || (gradient2 * slope < 0.0) ) bad = 1;
        } else if (in1) {
          if ( pqm_gradient(b, c, d, e, x1) * slope < 0.0 ) bad = 1;
        } else if (in2) {
          if ( pqm_gradient(b, c, d, e, x2) * slope < 0.0 ) bad = 1;
        }
        if (bad) { if ( fabs(sigma_l) < fabs(sigma_r) ) inflexion_l = 1; else inflexion_r = 1; }
      }
      if (( alpha1 == 0.0 ) && ( alpha2 != 0.0 )) {      /* the second derivative is a straight line */
        const double x1 = - alpha3 / alpha2;
        if ( (x1 >= 0.0) && (x1 <= 1.0) ) {
          if ( pqm_gradient(b, c, d, e, x1) * slope < 0.0 ) {
            if ( fabs(sigma_l) < fabs(sigma_r) ) inflexion_l = 1; else inflexion_r = 1;
          }
        }
      }
    }
    if ( inflexion_l == 1 ) {      /* both inflexion points collapse onto the left edge */
      u1_l = ( 10.0 * u_c - 2.0 * u0_r - 8.0 * u0_l ) / (3.0*h_c + hNeglect );
      u1_r = ( -10.0 * u_c + 6.0 * u0_r + 4.0 * u0_l ) / ( h_c + hNeglect );
      if ( u1_l * slope < 0.0 ) {
        u1_l = 0.0;
        u0_r = 5.0 * u_c - 4.0 * u0_l;
        u1_r = 20.0 * (u_c - u0_l) / ( h_c + hNeglect );
      } else if ( u1_r * slope < 0.0 ) {
        u1_r = 0.0;
        u0_l = (5.0*u_c - 3.0*u0_r) / 2.0;
        u1_l = 10.0 * (-u_c + u0_r) / (3.0 * h_c + hNeglect);
      }
    } else if ( inflexion_r == 1 ) {      /* onto the right edge */
      u1_r = ( -10.0 * u_c + 8.0 * u0_r + 2.0 * u0_l ) / (3.0 * h_c + hNeglect);
      u1_l = ( 10.0 * u_c - 4.0 * u0_r - 6.0 * u0_l ) / (h_c + hNeglect);
      if ( u1_l * slope < 0.0 ) {
        u1_l = 0.0;
        u0_r = ( 5.0 * u_c - 3.0 * u0_l ) / 2.0;
        u1_r = 10.0 * (u_c - u0_l) / (3.0 * h_c + hNeglect);
      } else if ( u1_r * slope < 0.0 ) {
        u1_r = 0.0;
        u0_l = 5.0 * u_c - 4.0 * u0_r;
        u1_l = 20.0 * ( -u_c + u0_r ) / (h_c + hNeglect);
      }
    }
    E_(E,k,0) = u0_l; E_(E,k,1) = u0_r; E_(S,k,0) = u1_l; E_(S,k,1) = u1_r;
  }
  E_(E,0,0) = u[0]; E_(E,0,1) = u[0]; E_(S,0,0) = 0.0; E_(S,0,1) = 0.0;
  E_(E,n-1,0) = u[n-1]; E_(E,n-1,1) = u[n-1]; E_(S,n-1,0) = 0.0; E_(S,n-1,1) = 0.0;
}

/* PQM_reconstruction, PQM_functions.F90:20-66.  coef holds 5*n doubles. */
void orc_pqm_reconstruction(int n, const double *h, const double *u, double *E, double *S, double *coef, double h_neglect)
{
  orc_pqm_limiter(n, h, u, E, S, h_neglect);
  for (int k = 0; k < n; k++)
    pqm_quartic(u[k], h[k], E_(E,k,0), E_(E,k,1), E_(S,k,0), E_(S,k,1),
                &C_(coef,k,0), &C_(coef,k,1), &C_(coef,k,2), &C_(coef,k,3), &C_(coef,k,4));
}

/* PQM_boundary_extrapolation_v1, PQM_functions.F90:502-831 */
void orc_pqm_boundary_extrapolation_v1(int n, const double *h, const double *u, double *E, double *S, double *coef, double h_neglect)
{
  const double hNeglect = h_neglect;
  double a, b, c, d, e;
  /* ----- left boundary (top) ----- */
  {
    const int i0 = 0, i1 = 1;
    const double h0 = h[i0], h1 = h[i1], u0 = u[i0], u1 = u[i1], um = u0;
    double slope = 2.0 * ( u1 - u0 ) / ( ( h0 + h1 ) + hNeglect );
    slope = slope * h0;
    a = C_(coef,i1,0); b = C_(coef,i1,1);
    double u0_r = a;
    double u1_r = b / (h1 + hNeglect);
    double beta;
    if (u1_r != 0.) beta = 2.0 * ( u0_r - um ) / ( (h0 + hNeglect)*u1_r) - 1.0;
    else beta = 0.;
    const double br = u0_r + beta*u0_r - um;
    const double ar = um + beta*um - br;
    double u0_l = ar, u1_l;
    const double u_plm = um - 0.5 * slope;
    if ( fabs(um-u0_l) < fabs(um-u_plm) ) {
      u1_l = 2.0 * ( br - ar*beta);
      u1_l = u1_l / (h0 + hNeglect);
    } else {
      u0_l = u_plm;
      u1_l = slope / (h0 + hNeglect);
    }
    int inflexion_l = 0;
    pqm_quartic(um, h0, u0_l, u0_r, u1_l, u1_r, &a, &b, &c, &d, &e);
    const double alpha1 = 6*e, alpha2 = 3*d, alpha3 = c;
    const double rho = alpha2 * alpha2 - 4.0 * alpha1 * alpha3;
    if (( alpha1 != 0.0 ) && ( rho >= 0.0 )) {
      const double sqrt_rho = sqrt( rho );
      const double x1 = 0.5 * ( - alpha2 - sqrt_rho ) / alpha1;
      if ( (x1 > 0.0) && (x1 < 1.0) ) { if ( pqm_gradient(b, c, d, e, x1) * slope < 0.0 ) inflexion_l = 1; }
      const double x2 = 0.5 * ( - alpha2 + sqrt_rho ) / alpha1;
      if ( (x2 > 0.0) && (x2 < 1.0) ) { if ( pqm_gradient(b, c, d, e, x2) * slope < 0.0 ) inflexion_l = 1; }
    }
    if (( alpha1 == 0.0 ) && ( alpha2 != 0.0 )) {
      const double x1 = - alpha3 / alpha2;
      if ( (x1 >= 0.0) && (x1 <= 1.0) ) {
        const double gradient1 = 3.0 * d * (x1*x1) + 2.0 * c * x1 + b;
        if ( gradient1 * slope < 0.0 ) inflexion_l = 1;
      }
    }
    if ( inflexion_l == 1 ) {
      u1_l = ( 10.0 * um - 2.0 * u0_r - 8.0 * u0_l ) / (3.0*h0 + hNeglect);
      u1_r = ( -10.0 * um + 6.0 * u0_r + 4.0 * u0_l ) / (h0 + hNeglect);
      if ( u1_l * slope < 0.0 ) {
        u1_l = 0.0;
        u0_r = 5.0 * um - 4.0 * u0_l;
        u1_r = 20.0 * (um - u0_l) / ( h0 + hNeglect );
      } else if ( u1_r * slope < 0.0 ) {
        u1_r = 0.0;
        u0_l = (5.0*um - 3.0*u0_r) / 2.0;
        u1_l = 10.0 * (-um + u0_r) / (3.0 * h0 + hNeglect );
      }
    }
    E_(E,i0,0) = u0_l; E_(E,i0,1) = u0_r; E_(S,i0,0) = u1_l; E_(S,i0,1) = u1_r;
    pqm_quartic(um, h0, u0_l, u0_r, u1_l, u1_r, &C_(coef,i0,0), &C_(coef,i0,1), &C_(coef,i0,2), &C_(coef,i0,3), &C_(coef,i0,4));
  }
  /* ----- right boundary (bottom) ----- */
  {
    const int i0 = n-2, i1 = n-1;
    const double h0 = h[i0], h1 = h[i1], u0 = u[i0], u1 = u[i1], um = u1;
    double slope = 2.0 * ( u1 - u0 ) / ( h0 + h1 );
    slope = slope * h1;
    a = C_(coef,i0,0); b = C_(coef,i0,1); c = C_(coef,i0,2); d = C_(coef,i0,3); e = C_(coef,i0,4);
    double u0_l = a + b + c + d + e;
    double u1_l = (b + 2*c + 3*d + 4*e) / h0;
    double beta;
    if (um-u0_l != 0.) beta = 0.5*h1*u1_l / (um-u0_l) - 1.0;
    else beta = 0.;
    const double br = beta*um + um - u0_l;
    const double ar = u0_l;
    double u0_r, u1_r;
    if (1+beta != 0.) u0_r = (ar + 2*br + beta*br ) / ((1+beta)*(1+beta));
    else u0_r = um + 0.5 * slope;
    const double u_plm = um + 0.5 * slope;
    if ( fabs(um-u0_r) < fabs(um-u_plm) ) {
      u1_r = 2.0 * ( br - ar*beta ) / ( (1+beta)*(1+beta)*(1+beta) );
      u1_r = u1_r / h1;
    } else {
      u0_r = u_plm;
      u1_r = slope / h1;
    }
    int inflexion_r = 0;
    pqm_quartic(um, h1, u0_l, u0_r, u1_l, u1_r, &a, &b, &c, &d, &e);
    const double alpha1 = 6*e, alpha2 = 3*d, alpha3 = c;
    const double rho = alpha2 * alpha2 - 4.0 * alpha1 * alpha3;
    if (( alpha1 != 0.0 ) && ( rho >= 0.0 )) {
      const double sqrt_rho = sqrt( rho );
      const double x1 = 0.5 * ( - alpha2 - sqrt_rho ) / alpha1;
      if ( (x1 > 0.0) && (x1 < 1.0) ) { if ( pqm_gradient(b, c, d, e, x1) * slope < 0.0 ) inflexion_r = 1; }
      const double x2 = 0.5 * ( - alpha2 + sqrt_rho ) / alpha1;
      if ( (x2 > 0.0) && (x2 < 1.0) ) { if ( pqm_gradient(b, c, d, e, x2) * slope < 0.0 ) inflexion_r = 1; }
    }
    if (( alpha1 == 0.0 ) && ( alpha2 != 0.0 )) {
      const double x1 = - alpha3 / alpha2;
      if ( (x1 >= 0.0) && (x1 <= 1.0) ) {
        const double gradient1 = 3.0 * d * (x1*x1) + 2.0 * c * x1 + b;
        if ( gradient1 * slope < 0.0 ) inflexion_r = 1;
      }
    }
    if ( inflexion_r == 1 ) {
      u1_r = ( -10.0 * um + 8.0 * u0_r + 2.0 * u0_l ) / (3.0 * h1);
      u1_l = ( 10.0 * um - 4.0 * u0_r - 6.0 * u0_l ) / h1;
      if ( u1_l * slope < 0.0 ) {
        u1_l = 0.0;
        u0_r = ( 5.0 * um - 3.0 * u0_l ) / 2.0;
        u1_r = 10.0 * (um - u0_l) / (3.0 * h1);
      } else if ( u1_r * slope < 0.0 ) {
        u1_r = 0.0;
        u0_l = 5.0 * um - 4.0 * u0_r;
        u1_l = 20.0 * ( -um + u0_r ) / h1;
      }
    }
    E_(E,i1,0) = u0_l; E_(E,i1,1) = u0_r; E_(S,i1,0) = u1_l; E_(S,i1,1) = u1_r;
    pqm_quartic(um, h1, u0_l, u0_r, u1_l, u1_r, &C_(coef,i1,0), &C_(coef,i1,1), &C_(coef,i1,2), &C_(coef,i1,3), &C_(coef,i1,4));
  }
}

/* ---- PQM_IH6IH5: the sixth-order edge values and fifth-order edge slopes ---------------------------------------- */
/* linear_solver, regrid_solvers.F90:115-176 (Gaussian elimination, the first nonzero pivot; A[row][col]).  Returns nonzero when the
 * system is singular (the reference stops with a FATAL error). */
int orc_linear_solver6(double A[6][6], double R[6], double X[6])
{
  const int N = 6;
  for (int i = 0; i < N-1; i++) {
    int k = i;
    while (k < N && !(fabs(A[k][i]) > 0.0)) k++;
    if (k >= N) return 1;
    if (k != i) {
      for (int j = i; j < N; j++) { const double swap = A[i][j]; A[i][j] = A[k][j]; A[k][j] = swap; }
      const double swap = R[i]; R[i] = R[k]; R[k] = swap;
    }
    const double I_pivot = 1.0 / A[i][i];
    A[i][i] = 1.0;
    for (int j = i+1; j < N; j++) A[i][j] = A[i][j] * I_pivot;
    R[i] = R[i] * I_pivot;
    for (int kk = i+1; kk < N; kk++) {
      const double factor = A[kk][i];
      for (int j = i+1; j < N; j++) A[kk][j] = A[kk][j] - factor * A[i][j];
      R[kk] = R[kk] - factor * R[i];
    }
  }
  if (A[N-1][N-1] == 0.0) return 2;
  X[N-1] = R[N-1] / A[N-1][N-1];
  for (int i = N-2; i >= 0; i--) {
    X[i] = R[i];
    for (int j = i+1; j < N; j++) X[i] = X[i] - A[i][j] * X[j];
  }
  return 0;
}

/* solve_tridiagonal_system without an answer_date (the 2008-2018 expressions), regrid_solvers.F90:183-217 */
void orc_solve_tridiagonal_system_2018(const double *Al, const double *Ad, const double *Au, const double *R, double *X, int N)
{
  double *pivot = (double*)calloc((size_t)N, sizeof(double));
  pivot[0] = Ad[0];
  X[0] = R[0];
  for (int k = 1; k < N; k++) {
    const double Al_piv = Al[k] / pivot[k-1];
    pivot[k] = Ad[k] - Al_piv * Au[k-1];
    X[k] = R[k] - Al_piv * X[k-1];
  }
  X[N-1] = R[N-1] / pivot[N-1];
  for (int k = N-2; k >= 0; k--) X[k] = ( X[k] - Au[k]*X[k+1] ) / pivot[k];
  free(pivot);
}

/* The polynomials of the cell widths that both routines' matrices share (Eq. 48 and 52 of White and Adcroft 2009):
 * P[m] = ((a+b)^(m+2) - a^(m+2)) / b for m = 0..4 as the reference writes them, with a the inner and b the outer cell */
static void ih_polys(double a, double b, double P[5])
{
  const double a_2 = a * a, a_3 = a_2 * a, a_4 = a_2 * a_2, a_5 = a_3 * a_2;
  P[0] = (2.0*a + b);
  P[1] = (3.0*a_2 + b*(3.0*a + b));
  P[2] = (4.0*a_3 + b*(6.0*a_2 + b*(4.0*a + b)));
  P[3] = (5.0*a_4 + b*(10.0*a_3 + b*(10.0*a_2 + b*(5.0*a + b))));
  P[4] = (6.0*a_5 + b*(15.0*a_4 + b*(20.0*a_3 + b*(15.0*a_2 + b*(6.0*a + b)))));
}
/* A row of the boundary systems (the cell average of 1, x, .., x^5 over a cell of width dx centred on xavg), :1169-1171, :1367-1369 */
static void ih_boundary_row(double xavg, double dx, double row[6])
{
  const double C1_12 = 1.0 / 12.0, C5_6 = 5.0 / 6.0;
  const double x2 = xavg*xavg, x4 = x2*x2, d2 = dx*dx, d4 = d2*d2;
  row[0] = 1.0; row[1] = xavg; row[2] = (x2 + C1_12*d2); row[3] = xavg * (x2 + 0.25*d2);
  row[4] = (x4 + 0.5*x2*d2 + 0.0125*d4);
  row[5] = xavg * (x4 + C5_6*x2*d2 + 0.0625*d4);
}

/* which: 0 the centred stencil, 1 the right-biased one of the second row, 2 the left-biased one of the second to last row */
static int ih5_coefs(double h0, double h1, double h2, double h3, int which, double C[6])
{
  double A[6][6], B[6], Pl[5], Pr[5];
  const double h1_2 = h1 * h1, h1_3 = h1_2 * h1, h1_4 = h1_2 * h1_2, h1_5 = h1_3 * h1_2;
  const double h2_2 = h2 * h2, h2_3 = h2_2 * h2, h2_4 = h2_2 * h2_2, h2_5 = h2_3 * h2_2;
  ih_polys(h1, h0, Pl); ih_polys(h2, h3, Pr);
  const double r3[6] = {1.0, Pl[0], Pl[1], -Pl[2], Pl[3], -Pl[4]};
  const double r4[6] = {1.0, h1, h1_2, -h1_3, h1_4, -h1_5};
  const double r5[6] = {1.0, -h2, h2_2, h2_3, h2_4, h2_5};
  const double r6[6] = {1.0, -Pr[0], Pr[1], Pr[2], Pr[3], Pr[4]};
  for (int n = 0; n < 6; n++) { A[n][2] = r3[n]; A[n][3] = r4[n]; A[n][4] = r5[n]; A[n][5] = r6[n]; }
  A[0][0] = 0.0; A[0][1] = 0.0; A[1][0] = 2.0; A[1][1] = 2.0;
  if (which == 0) {
    A[2][0] = 6.0*h1;     A[2][1] = -6.0* h2;
    A[3][0] = -12.0*h1_2; A[3][1] = -12.0*h2_2;
    A[4][0] = 20.0*h1_3;  A[4][1] = -20.0*h2_3;
    A[5][0] = -30.0*h1_4; A[5][1] = -30.0*h2_4;
    B[0] = 0.0; B[1] = -2.0; B[2] = 0.0; B[3] = 0.0; B[4] = 0.0; B[5] = 0.0;
  } else if (which == 1) {
    const double h01 = h0 + h1, h01_2 = h01 * h01;
    A[2][0] = 6.0*h01;             A[2][1] = 0.0;
    A[3][0] = -12.0*h01_2;         A[3][1] = 0.0;
    A[4][0] = 20.0*(h01*h01_2);    A[4][1] = 0.0;
    A[5][0] = -30.0*(h01_2*h01_2); A[5][1] = 0.0;
    B[0] = 0.0; B[1] = -2.0; B[2] = -6.0*h1; B[3] = 12.0*h1_2; B[4] = -20.0*h1_3; B[5] = 30.0*h1_4;
  } else {
    const double h23 = h2 + h3, h23_2 = h23 * h23;
    A[2][0] = 0.0; A[2][1] = -6.0*h23;
    A[3][0] = 0.0; A[3][1] = -12.0*h23_2;
    A[4][0] = 0.0; A[4][1] = -20.0*(h23*h23_2);
    A[5][0] = 0.0; A[5][1] = -30.0*(h23_2*h23_2);
    B[0] = 0.0; B[1] = -2.0; B[2] = 6.0*h2; B[3] = 12.0*h2_2; B[4] = 20.0*h2_3; B[5] = 30.0*h2_4;
  }
  return orc_linear_solver6(A, B, C);
}

/* edge_slopes_implicit_h5, regrid_edge_values.F90:977-1227.  n >= 6.  Returns nonzero if one of the 6x6 systems is singular. */
int orc_edge_slopes_implicit_h5(int n, const double *h, const double *u, double *S, double h_neglect)
{
  const double hNeglect = h_neglect, h_Min_Frac = 1.0e-4;
  const int m = n + 1;
  int rc = 0;
  double *tri_l = (double*)calloc(5*(size_t)m, sizeof(double));
  double *tri_d = tri_l + m, *tri_u = tri_d + m, *tri_b = tri_u + m, *tri_x = tri_b + m;
  double C[6];
  for (int k = 1; k <= n-3; k++) {           /* Fortran k = 2..N-2: cells k-1..k+2 are 0-based k-1..k+2; row k+1 is 0-based k+1 */
    const double hMin = max2(hNeglect, h_Min_Frac*((h[k-1] + h[k]) + (h[k+1] + h[k+2])));
    rc |= ih5_coefs(max2(h[k-1], hMin), max2(h[k], hMin), max2(h[k+1], hMin), max2(h[k+2], hMin), 0, C);
    tri_l[k+1] = C[0]; tri_d[k+1] = 1.0; tri_u[k+1] = C[1];
    tri_b[k+1] = C[2] * u[k-1] + C[3] * u[k] + C[4] * u[k+1] + C[5] * u[k+2];
  }
  {
    const double hMin = max2(hNeglect, h_Min_Frac*((h[0] + h[1]) + (h[2] + h[3])));
    rc |= ih5_coefs(max2(h[0], hMin), max2(h[1], hMin), max2(h[2], hMin), max2(h[3], hMin), 1, C);
    tri_l[1] = C[0]; tri_d[1] = 1.0; tri_u[1] = C[1];
    tri_b[1] = C[2] * u[0] + C[3] * u[1] + C[4] * u[2] + C[5] * u[3];
  }
  {
    double A[6][6], B[6], x = 0.0;
    for (int i = 0; i < 6; i++) {
      const double dx = h[i];
      const double xavg = x + 0.5 * dx;
      ih_boundary_row(xavg, dx, A[i]);
      B[i] = u[i];
      x = x + dx;
    }
    rc |= orc_linear_solver6(A, B, C);
    tri_d[0] = 1.0; tri_u[0] = 0.0; tri_b[0] = C[1];
  }
  {
    const double hMin = max2(hNeglect, h_Min_Frac*((h[n-4] + h[n-3]) + (h[n-2] + h[n-1])));
    rc |= ih5_coefs(max2(h[n-4], hMin), max2(h[n-3], hMin), max2(h[n-2], hMin), max2(h[n-1], hMin), 2, C);
    tri_l[n-1] = C[0]; tri_d[n-1] = 1.0; tri_u[n-1] = C[1];
    tri_b[n-1] = C[2] * u[n-4] + C[3] * u[n-3] + C[4] * u[n-2] + C[5] * u[n-1];
  }
  {
    double A[6][6], B[6], x = 0.0;
    for (int i = 0; i < 6; i++) {
      const double dx = h[n-1-i];
      const double xavg = x + 0.5*dx;
      ih_boundary_row(xavg, dx, A[i]);
      B[i] = u[n-1-i];
      x = x + dx;
    }
    rc |= orc_linear_solver6(A, B, C);
    tri_l[n] = 0.0; tri_d[n] = 1.0; tri_u[n] = 0.0; tri_b[n] = -C[1];
  }
  orc_solve_tridiagonal_system_2018(tri_l, tri_d, tri_u, tri_b, tri_x, m);
  for (int i = 1; i < n; i++) { E_(S,i,0) = tri_x[i]; E_(S,i-1,1) = tri_x[i]; }
  E_(S,0,0) = tri_x[0];
  E_(S,n-1,1) = tri_x[n];
  free(tri_l);
  return rc;
}

static int ih6_coefs(double h0, double h1, double h2, double h3, int which, double C[6])
{
  double A[6][6], B[6], Pl[5], Pr[5];
  const double h1_2 = h1 * h1, h1_3 = h1_2 * h1, h1_4 = h1_2 * h1_2, h1_5 = h1_3 * h1_2;
  const double h2_2 = h2 * h2, h2_3 = h2_2 * h2, h2_4 = h2_2 * h2_2, h2_5 = h2_3 * h2_2;
  ih_polys(h1, h0, Pl); ih_polys(h2, h3, Pr);
  const double r3[6] = {-1.0, Pl[0], -Pl[1], Pl[2], -Pl[3], Pl[4]};
  const double r4[6] = {-1.0, h1, -h1_2, h1_3, -h1_4, h1_5};
  const double r5[6] = {-1.0, -h2, -h2_2, -h2_3, -h2_4, -h2_5};
  const double r6[6] = {-1.0, -Pr[0], -Pr[1], -Pr[2], -Pr[3], -Pr[4]};
  for (int n = 0; n < 6; n++) { A[n][2] = r3[n]; A[n][3] = r4[n]; A[n][4] = r5[n]; A[n][5] = r6[n]; }
  A[0][0] = 1.0; A[0][1] = 1.0;
  if (which == 0) {
    A[1][0] = -2.0*h1;   A[1][1] = 2.0*h2;
    A[2][0] = 3.0*h1_2;  A[2][1] = 3.0*h2_2;
    A[3][0] = -4.0*h1_3; A[3][1] = 4.0*h2_3;
    A[4][0] = 5.0*h1_4;  A[4][1] = 5.0*h2_4;
    A[5][0] = -6.0*h1_5; A[5][1] = 6.0*h2_5;
    B[0] = -1.0; B[1] = 0.0; B[2] = 0.0; B[3] = 0.0; B[4] = 0.0; B[5] = 0.0;
  } else if (which == 1) {
    const double h01 = h0 + h1, h01_2 = h01 * h01, h01_3 = h01 * h01_2;
    A[1][0] = -2.0*h01;            A[1][1] = 0.0;
    A[2][0] = 3.0*h01_2;           A[2][1] = 0.0;
    A[3][0] = -4.0*h01_3;          A[3][1] = 0.0;
    A[4][0] = 5.0*(h01_2*h01_2);   A[4][1] = 0.0;
    A[5][0] = -6.0*(h01_3*h01_2);  A[5][1] = 0.0;
    B[0] = -1.0; B[1] = 2.0*h1; B[2] = -3.0*h1_2; B[3] = 4.0*h1_3; B[4] = -5.0*h1_4; B[5] = 6.0*h1_5;
  } else {
    const double h23 = h2 + h3, h23_2 = h23 * h23, h23_3 = h23 * h23_2;
    A[1][0] = 0.0; A[1][1] = 2.0*h23;
    A[2][0] = 0.0; A[2][1] = 3.0*h23_2;
    A[3][0] = 0.0; A[3][1] = 4.0*h23_3;
    A[4][0] = 0.0; A[4][1] = 5.0*(h23_2*h23_2);
    A[5][0] = 0.0; A[5][1] = 6.0*(h23_3*h23_2);
    B[0] = -1.0; B[1] = -2.0*h2; B[2] = -3.0*h2_2; B[3] = -4.0*h2_3; B[4] = -5.0*h2_4; B[5] = -6.0*h2_5;
  }
  return orc_linear_solver6(A, B, C);
}

/* edge_values_implicit_h6, regrid_edge_values.F90:1252-1454.  n >= 6. */
int orc_edge_values_implicit_h6(int n, const double *h, const double *u, double *E, double h_neglect_edge)
{
  const double hNeglect = h_neglect_edge;
  const int m = n + 1;
  int rc = 0;
  double *tri_l = (double*)calloc(5*(size_t)m, sizeof(double));
  double *tri_d = tri_l + m, *tri_u = tri_d + m, *tri_b = tri_u + m, *tri_x = tri_b + m;
  double C[6];
  for (int k = 1; k <= n-3; k++) {
    const double hMin = max2(hNeglect, hMinFrac*((h[k-1] + h[k]) + (h[k+1] + h[k+2])));
    rc |= ih6_coefs(max2(h[k-1], hMin), max2(h[k], hMin), max2(h[k+1], hMin), max2(h[k+2], hMin), 0, C);
    tri_l[k+1] = C[0]; tri_d[k+1] = 1.0; tri_u[k+1] = C[1];
    tri_b[k+1] = C[2] * u[k-1] + C[3] * u[k] + C[4] * u[k+1] + C[5] * u[k+2];
  }
  {
    const double hMin = max2(hNeglect, hMinFrac*((h[0] + h[1]) + (h[2] + h[3])));
    rc |= ih6_coefs(max2(h[0], hMin), max2(h[1], hMin), max2(h[2], hMin), max2(h[3], hMin), 1, C);
    tri_l[1] = C[0]; tri_d[1] = 1.0; tri_u[1] = C[1];
    tri_b[1] = C[2] * u[0] + C[3] * u[1] + C[4] * u[2] + C[5] * u[3];
  }
  {
    const double hMin = max2( hNeglect, hMinFrac*((h[0]+h[1]) + (h[4]+h[5]) + (h[2]+h[3])) );
    double A[6][6], B[6], x = 0.0;
    for (int i = 0; i < 6; i++) {
      const double dx = max2( hMin, h[i] );
      const double xavg = x + 0.5*dx;
      ih_boundary_row(xavg, dx, A[i]);
      B[i] = u[i];
      x = x + dx;
    }
    rc |= orc_linear_solver6(A, B, C);
    double f = 0.0;                            /* evaluation_polynomial( Csys, 6, x(1) = 0.0 ): 0.0**0 = 1.0 */
    f = f + C[0] * 1.0;
    for (int k = 1; k < 6; k++) f = f + C[k] * 0.0;
    tri_l[0] = 0.0; tri_d[0] = 1.0; tri_u[0] = 0.0; tri_b[0] = f;
  }
  {
    const double hMin = max2(hNeglect, hMinFrac*((h[n-4] + h[n-3]) + (h[n-2] + h[n-1])));
    rc |= ih6_coefs(max2(h[n-4], hMin), max2(h[n-3], hMin), max2(h[n-2], hMin), max2(h[n-1], hMin), 2, C);
    tri_l[n-1] = C[0]; tri_d[n-1] = 1.0; tri_u[n-1] = C[1];
    tri_b[n-1] = C[2] * u[n-4] + C[3] * u[n-3] + C[4] * u[n-2] + C[5] * u[n-1];
  }
  {
    /* (as the reference has it: hMinFrac multiplies the first pair only, :1436) */
    const double hMin = max2( hNeglect, hMinFrac*(h[n-4] + h[n-3]) + ((h[n-2] + h[n-1]) + (h[n-6] + h[n-5])) );
    double A[6][6], B[6], x = 0.0;
    for (int i = 0; i < 6; i++) {
      const double dx = max2( hMin, h[n-1-i] );
      const double xavg = x + 0.5 * dx;
      ih_boundary_row(xavg, dx, A[i]);
      B[i] = u[n-1-i];
      x = x + dx;
    }
    rc |= orc_linear_solver6(A, B, C);
    tri_l[n] = 0.0; tri_d[n] = 1.0; tri_u[n] = 0.0; tri_b[n] = C[0];
  }
  orc_solve_tridiagonal_system_2018(tri_l, tri_d, tri_u, tri_b, tri_x, m);
  for (int i = 1; i < n; i++) { E_(E,i,0) = tri_x[i]; E_(E,i-1,1) = tri_x[i]; }
  E_(E,0,0) = tri_x[0];
  E_(E,n-1,1) = tri_x[n];
  free(tri_l);
  return rc;
}

/* ---- remapping ---------------------------------------------------------------------------- */
/* average_value_ppoly, MOM_remapping.F90:998-1099 (method: ORC_INT_PCM/PLM/PPM/PQM; i0 0-based) */
double orc_average_value_ppoly(int n, const double *u0, const double *E, const double *coef, int method,
                               int i0, double xa, double xb)
{
  double u_ave = 0.0;
  if (xb > xa) {
    if (method == ORC_INT_PCM) {
      u_ave = u0[i0];
    } else if (method == ORC_INT_PLM) {
      u_ave = ( C_(coef,i0,0) + C_(coef,i0,1) * 0.5 * ( xb + xa ) );
    } else if (method == ORC_INT_PQM) {
      const double r_3 = 1.0/3.0;
      const double xa_2 = xa*xa, xb_2 = xb*xb;
      const double xa2pxb2 = xa_2 + xb_2, xapxb = xa + xb;
      u_ave = ( C_(coef,i0,0)
          + ( C_(coef,i0,1) * 0.5 * ( xapxb )
          + ( C_(coef,i0,2) * r_3 * ( xa2pxb2 + xa*xb )
          + ( C_(coef,i0,3) * 0.25* ( xa2pxb2 * xapxb )
          +   C_(coef,i0,4) * 0.2 * ( ( xb*xb_2 + xa*xa_2 ) * xapxb + xa_2*xb_2 ) ) ) ) );
    } else {
      double mx = 0.5 * ( xa + xb );
      double a_L = E_(E,i0,0), a_R = E_(E,i0,1), u_c = u0[i0];
      double a_c = 0.5 * ( ( u_c - a_L ) + ( u_c - a_R ) );
      if (mx < 0.5) {
        double xa2b2ab = (xa*xa+xb*xb)+xa*xb;
        u_ave = a_L + ( ( a_R - a_L ) * mx + a_c * ( 3. * ( xb + xa ) - 2.*xa2b2ab ) );
      } else {
        double Ya = 1. - xa, Yb = 1. - xb;
        double my = 0.5 * ( Ya + Yb );
        double Ya2b2ab = (Ya*Ya+Yb*Yb)+Ya*Yb;
        u_ave = a_R  + ( ( a_L - a_R ) * my + a_c * ( 3. * ( Yb + Ya ) - 2.*Ya2b2ab ) );
      }
    }
  } else {
    if (method == ORC_INT_PCM) {
      u_ave = C_(coef,i0,0);
    } else if (method == ORC_INT_PLM) {
      double a_L = E_(E,i0,0), a_R = E_(E,i0,1);
      double Ya = 1. - xa;
      if (xa < 0.5) u_ave = a_L + xa * ( a_R - a_L );
      else          u_ave = a_R + Ya * ( a_L - a_R );
    } else if (method == ORC_INT_PQM) {
      u_ave = C_(coef,i0,0) + xa * ( C_(coef,i0,1) + xa * ( C_(coef,i0,2) + xa * ( C_(coef,i0,3) + xa * C_(coef,i0,4) ) ) );
    } else {
      double a_L = E_(E,i0,0), a_R = E_(E,i0,1), u_c = u0[i0];
      double a_c = 3. * ( ( u_c - a_L ) + ( u_c - a_R ) );
      double Ya = 1. - xa;
      if (xa < 0.5) u_ave = a_L + xa * ( ( a_R - a_L ) + a_c * Ya );
      else          u_ave = a_R + Ya * ( ( a_L - a_R ) + a_c * xa );
    }
  }
  return u_ave;
}

/* remap_via_sub_cells, MOM_remapping.F90:463-852 (force_bounds_in_subcell as argument;
 * force_bounds_in_target and adjust_thickest_subcell are .true. parameters there). */
void orc_remap_via_sub_cells(int n0, const double *h0, const double *u0, const double *E, const double *coef,
                             int n1, const double *h1, int method, int force_bounds_in_subcell,
                             double *u1, double *uh_err_out)
{
  const int n = n0;   /* for the E_/C_ macros */
  const int ns = n0 + n1 + 1;
  /* 1-based work arrays to keep the reference's index arithmetic */
  double *h_sub = calloc(ns+2, sizeof(double)), *uh_sub = calloc(ns+2, sizeof(double)), *u_sub = calloc(ns+2, sizeof(double));
  int *isub_src = calloc(ns+2, sizeof(int));
  int *isrc_start = calloc(n0+2, sizeof(int)), *isrc_end = calloc(n0+2, sizeof(int)), *isrc_max = calloc(n0+2, sizeof(int));
  double *h0_eff = calloc(n0+2, sizeof(double)), *u0_min = calloc(n0+2, sizeof(double)), *u0_max = calloc(n0+2, sizeof(double));
  int *itgt_start = calloc(n1+2, sizeof(int)), *itgt_end = calloc(n1+2, sizeof(int));
#define H0(i) h0[(i)-1]
#define H1(i) h1[(i)-1]
#define U0(i) u0[(i)-1]
  int i0_last_thick_cell = 0;
  for (int i0 = 1; i0 <= n0; i0++) {
    u0_min[i0] = min2(E_(E,i0-1,0), E_(E,i0-1,1));
    u0_max[i0] = max2(E_(E,i0-1,0), E_(E,i0-1,1));
    if (H0(i0) > 0.) i0_last_thick_cell = i0;
  }
  double h0_supply = H0(1), h1_supply = H1(1);
  int src_has_volume = 1, tgt_has_volume = 1;
  int i0 = 1, i1 = 1, i_start0 = 1, i_start1 = 1, i_max = 1;
  double dh_max = 0., dh0_eff = 0., dh;
  h_sub[1] = 0.;
  isrc_start[1] = 1; isrc_end[1] = 1; isrc_max[1] = 1; isub_src[1] = 1;

  for (int i_sub = 2; i_sub <= ns; i_sub++) {
    dh = min2(h0_supply, h1_supply);
    dh0_eff = dh0_eff + min2(dh, h0_supply);
    isub_src[i_sub] = i0;
    h_sub[i_sub] = dh;
    if (dh >= dh_max) { i_max = i_sub; dh_max = dh; }
    if (h0_supply <= h1_supply && src_has_volume) {
      h1_supply = h1_supply - dh;
      isrc_start[i0] = i_start0; isrc_end[i0] = i_sub; i_start0 = i_sub + 1;
      isrc_max[i0] = i_max; i_max = i_sub + 1; dh_max = 0.;
      h0_eff[i0] = dh0_eff;
      if (i0 < n0) { i0 = i0 + 1; h0_supply = H0(i0); dh0_eff = 0.; }
      else { h0_supply = 0.; src_has_volume = 0; }
    } else if (h0_supply >= h1_supply && tgt_has_volume) {
      h0_supply = h0_supply - dh;
      itgt_start[i1] = i_start1; itgt_end[i1] = i_sub; i_start1 = i_sub + 1;
      if (i1 < n1) { i1 = i1 + 1; h1_supply = H1(i1); }
      else { h1_supply = 0.; tgt_has_volume = 0; }
    } else if (src_has_volume) {
      h_sub[i_sub] = h0_supply;
      isrc_start[i0] = i_start0; isrc_end[i0] = i_sub; i_start0 = i_sub + 1;
      isrc_max[i0] = i_max; i_max = i_sub + 1; dh_max = 0.;
      h0_eff[i0] = dh0_eff;
      if (i0 < n0) { i0 = i0 + 1; h0_supply = H0(i0); dh0_eff = 0.; }
      else { h0_supply = 0.; src_has_volume = 0; }
    } else if (tgt_has_volume) {
      h_sub[i_sub] = h1_supply;
      itgt_start[i1] = i_start1; itgt_end[i1] = i_sub; i_start1 = i_sub + 1;
      if (i1 < n1) { i1 = i1 + 1; h1_supply = H1(i1); }
      else { h1_supply = 0.; tgt_has_volume = 0; }
    } else {
      abort();   /* 'remap_via_sub_cells: THIS SHOULD NEVER HAPPEN!' */
    }
  }

  double xa = 0., xb;
  dh0_eff = 0.;
  uh_sub[1] = 0.;
  u_sub[1] = E_(E,0,0);
  double u02_err = 0.;
  for (int i_sub = 2; i_sub <= n0+n1; i_sub++) {
    dh = h_sub[i_sub];
    i0 = isub_src[i_sub];
    dh0_eff = dh0_eff + dh;
    if (h0_eff[i0] > 0.) {
      xb = dh0_eff / h0_eff[i0];
      xb = min2(1., xb);
      u_sub[i_sub] = orc_average_value_ppoly( n0, u0, E, coef, method, i0-1, xa, xb);
    } else {
      xb = 1.;
      u_sub[i_sub] = U0(i0);
    }
    if (force_bounds_in_subcell) {
      double u_orig = u_sub[i_sub];
      u_sub[i_sub] = max2( u_sub[i_sub], u0_min[i0] );
      u_sub[i_sub] = min2( u_sub[i_sub], u0_max[i0] );
      u02_err = u02_err + dh*fabs( u_sub[i_sub] - u_orig );
    }
    uh_sub[i_sub] = dh * u_sub[i_sub];
    if (isub_src[i_sub+1] != i0) { dh0_eff = 0.; xa = 0.; }
    else { xa = xb; }
  }
  u_sub[ns] = E_(E,n0-1,1);
  uh_sub[ns] = E_(E,n0-1,1) * h_sub[ns];

  /* adjust_thickest_subcell */
  for (i0 = 1; i0 <= i0_last_thick_cell; i0++) {
    i_max = isrc_max[i0];
    dh_max = h_sub[i_max];
    if (dh_max > 0.) {
      double duh = 0.;
      for (int i_sub = isrc_start[i0]; i_sub <= isrc_end[i0]; i_sub++)
        if (i_sub != i_max) duh = duh + uh_sub[i_sub];
      uh_sub[i_max] = U0(i0)*H0(i0) - duh;
      u02_err = u02_err + max3( fabs(uh_sub[i_max]), fabs(U0(i0)*H0(i0)), fabs(duh) );
    }
  }

  double uh_err = 0.;
  for (i1 = 1; i1 <= n1; i1++) {
    if (H1(i1) > 0.) {
      double duh = 0.; dh = 0.;
      int i_sub = itgt_start[i1];
      double u1min = u_sub[i_sub], u1max = u_sub[i_sub];
      for (i_sub = itgt_start[i1]; i_sub <= itgt_end[i1]; i_sub++) {
        u1min = min2(u1min, u_sub[i_sub]);
        u1max = max2(u1max, u_sub[i_sub]);
        dh = dh + h_sub[i_sub];
        duh = duh + uh_sub[i_sub];
        uh_err = uh_err + max2(fabs(duh),fabs(uh_sub[i_sub]))*DBL_EPSILON;
      }
      u1[i1-1] = duh / dh;
      uh_err = uh_err + fabs(duh)*DBL_EPSILON;
      double u_orig = u1[i1-1];
      u1[i1-1] = max2(u1min, min2(u1max, u1[i1-1]));
      uh_err = uh_err + dh*fabs( u1[i1-1]-u_orig );
    } else {
      u1[i1-1] = u_sub[itgt_start[i1]];
    }
  }
  uh_err = uh_err + u02_err;
  if (uh_err_out) *uh_err_out = uh_err;
#undef H0
#undef H1
#undef U0
  free(h_sub); free(uh_sub); free(u_sub); free(isub_src); free(isrc_start); free(isrc_end); free(isrc_max);
  free(h0_eff); free(u0_min); free(u0_max); free(itgt_start); free(itgt_end);
}

/* ---- MOM_hybgen_remap.F90: the HYCOM reconstructions behind PLM_HYBGEN, PPM_HYBGEN, WENO_HYBGEN (one scalar field, no PCM_lay).
 * The reference file compiles with no other module, so these three are checked bit for bit against the reference's own code
 * (oracle/_ref, tests/test_oracle_remapping.py). */

/* hybgen_plm_coefs :14-88: slope [n] (the PLM slope times the cell width) */
void orc_hybgen_plm_coefs(int nk, const double *si, const double *dpi, double *slope, double thin)
{
  slope[0] = 0.0; slope[nk-1] = 0.0;
  for (int k = 1; k < nk-1; k++) {
    if (dpi[k] <= thin) {
      slope[k] = 0.0;
    } else {
      const double qcen = dpi[k] / (dpi[k]+0.5*(dpi[k-1]+dpi[k+1]));
      const double ztop = 2.0*(si[k]-si[k-1]);
      const double zbot = 2.0*(si[k+1]-si[k]);
      const double zcen = qcen*(si[k+1]-si[k-1]);
      if (ztop*zbot > 0.0) slope[k] = copysign(min3(fabs(zcen),fabs(zbot),fabs(ztop)), zbot);
      else slope[k] = 0.0;
    }
  }
}

/* hybgen_ppm_coefs :91-222: E[0][k] the value at the interface above, E[1][k] below */
void orc_hybgen_ppm_coefs(int nk, const double *s, const double *h_src, double *E, double thin)
{
  double *dp = calloc(nk+2, 8), *as = calloc(nk+2, 8), *al = calloc(nk+2, 8), *ar = calloc(nk+2, 8);
  double *h112 = calloc(nk+3, 8), *h122 = calloc(nk+3, 8), *I_h12 = calloc(nk+3, 8), *h2_h123 = calloc(nk+2, 8);
  double *I_h0123 = calloc(nk+2, 8), *h01_h112 = calloc(nk+3, 8), *h23_h122 = calloc(nk+3, 8);
  int *PCM_layer = calloc(nk+2, sizeof(int));
  /* 1-based as in the reference */
#define S(k) s[(k)-1]
  for (int k = 1; k <= nk; k++) dp[k] = max2(h_src[k-1], thin);
  for (int k = 1; k <= nk; k++) PCM_layer[k] = (dp[k] <= thin);
  for (int k = 2; k <= nk; k++) {
    h112[k] = 2.*dp[k-1] + dp[k];
    h122[k] = dp[k-1] + 2.*dp[k];
    I_h12[k] = 1.0 / (dp[k-1] + dp[k]);
  }
  for (int k = 2; k <= nk-1; k++) h2_h123[k] = dp[k] / (dp[k] + (dp[k-1]+dp[k+1]));
  for (int k = 3; k <= nk-1; k++) {
    I_h0123[k] = 1.0 / ((dp[k-2] + dp[k-1]) + (dp[k] + dp[k+1]));
    h01_h112[k] = (dp[k-2] + dp[k-1]) / (2.0*dp[k-1] + dp[k]);
    h23_h122[k] = (dp[k] + dp[k+1])   / (dp[k-1] + 2.0*dp[k]);
  }
  as[1] = 0.;
  for (int k = 2; k <= nk-1; k++) {
    if (PCM_layer[k]) {
      as[k] = 0.0;
    } else {
      const double slk = S(k)-S(k-1);
      const double srk = S(k+1)-S(k);
      if (slk*srk > 0.) {
        const double sck = h2_h123[k]*( h112[k]*srk*I_h12[k+1] + h122[k+1]*slk*I_h12[k] );
        as[k] = copysign(min3(fabs(2.0*slk), fabs(sck), fabs(2.0*srk)), sck);
      } else {
        as[k] = 0.;
      }
    }
  }
  as[nk] = 0.;
  al[1] = S(1); ar[1] = S(1); al[2] = S(1);
  for (int k = 3; k <= nk-1; k++) {
    al[k] = (dp[k]*S(k-1) + dp[k-1]*S(k)) * I_h12[k]
          + I_h0123[k]*( 2.*dp[k]*dp[k-1]*I_h12[k]*(S(k)-S(k-1)) *
                         ( h01_h112[k] - h23_h122[k] )
                  + (dp[k]*as[k-1]*h23_h122[k] - dp[k-1]*as[k]*h01_h112[k]) );
    ar[k-1] = al[k];
  }
  ar[nk-1] = S(nk); al[nk] = S(nk); ar[nk] = S(nk);
  for (int k = 2; k <= nk-1; k++) {
    if (PCM_layer[k] || ((S(k+1)-S(k))*(S(k)-S(k-1)) <= 0.)) {
      al[k] = S(k); ar[k] = S(k);
    } else {
      const double da = ar[k]-al[k];
      const double a6 = 6.0*S(k) - 3.0*(al[k]+ar[k]);
      if (da*a6 > da*da) al[k] = 3.0*S(k) - 2.0*ar[k];
      else if (da*a6 < -da*da) ar[k] = 3.0*S(k) - 2.0*al[k];
    }
  }
  for (int k = 1; k <= nk; k++) { E[k-1] = al[k]; E[nk + k-1] = ar[k]; }
#undef S
  free(dp); free(as); free(al); free(ar); free(h112); free(h122); free(I_h12); free(h2_h123); free(I_h0123); free(h01_h112);
  free(h23_h122); free(PCM_layer);
}

/* hybgen_weno_coefs :226-386 */
void orc_hybgen_weno_coefs(int nk, const double *s, const double *h_src, double *E, double thin)
{
  const double min_ratio = 1.0e-8;
  double *dp = calloc(nk+2, 8), *qdpkm = calloc(nk+2, 8), *qdpkmkp = calloc(nk+2, 8), *dpkm2kp = calloc(nk+2, 8);
  double *zw1 = calloc(nk+2, 8), *zw2 = calloc(nk+2, 8), *slope_edge = calloc(nk+3, 8), *val_edge = calloc(nk+3, 8);
  double *e1 = calloc(nk+2, 8), *e2 = calloc(nk+2, 8);
  int *PCM_layer = calloc(nk+2, sizeof(int));
#define S(k) s[(k)-1]
  for (int k = 1; k <= nk; k++) dp[k] = max2(h_src[k-1], thin);
  for (int k = 1; k <= nk; k++) PCM_layer[k] = (dp[k] <= thin);
  for (int k = 2; k <= nk-1; k++) {
    qdpkm[k] = 1.0 / (dp[k-1] + dp[k]);
    qdpkmkp[k] = 1.0 / (dp[k-1] + dp[k] + dp[k+1]);
    dpkm2kp[k] = dp[k-1] + 2.0*dp[k] + dp[k+1];
  }
  qdpkm[nk] = 1.0 / (dp[nk-1] + dp[nk]);
  for (int k = 2; k <= nk; k++) slope_edge[k] = qdpkm[k] * (S(k)-S(k-1));
  e1[1] = S(1); e2[1] = S(1); zw1[1] = 0.0; zw2[1] = 0.0;
  for (int k = 2; k <= nk-1; k++) {
    if ((slope_edge[k]*slope_edge[k+1] < 0.0) || PCM_layer[k]) {
      e1[k] = S(k); e2[k] = S(k); zw1[k] = 0.0; zw2[k] = 0.0;
    } else {
      double seh1 = dp[k]*slope_edge[k+1];
      double seh2 = dp[k]*slope_edge[k];
      const double q01 = dpkm2kp[k]*slope_edge[k+1];
      const double q02 = dpkm2kp[k]*slope_edge[k];
      if (fabs(seh1) > fabs(q02)) seh1 = q02;
      if (fabs(seh2) > fabs(q01)) seh2 = q01;
      const double curv_cell = (seh1 - seh2) * qdpkmkp[k];
      const double q001 = seh1 - curv_cell*dp[k+1];
      const double q002 = seh2 + curv_cell*dp[k-1];
      e2[k] = S(k) + q001;
      e1[k] = S(k) - q002;
      zw1[k] = (2.0*q001 - q002)*(2.0*q001 - q002);
      zw2[k] = (2.0*q002 - q001)*(2.0*q002 - q001);
    }
  }
  e1[nk] = S(nk); e2[nk] = S(nk); zw1[nk] = 0.0; zw2[nk] = 0.0;
  for (int k = 2; k <= nk; k++) {
    double wt1;
    if (zw1[k] + zw2[k-1] <= 0.0) wt1 = 0.5;
    else if (zw1[k] <= min_ratio * (zw1[k] + zw2[k-1])) wt1 = min_ratio;
    else if (zw2[k-1] <= min_ratio * (zw1[k] + zw2[k-1])) wt1 = (1.0 - min_ratio);
    else wt1 = zw1[k] / (zw1[k] + zw2[k-1]);
    val_edge[k] = wt1*e2[k-1] + (1.0-wt1)*e1[k];
  }
  val_edge[1] = 2.0*S(1)-val_edge[2];
  val_edge[nk+1] = 2.0*S(nk)-val_edge[nk];
  for (int k = 2; k <= nk-1; k++) {
    if (!PCM_layer[k]) {
      double q01 = val_edge[k+1] - S(k);
      double q02 = S(k) - val_edge[k];
      if (q01*q02 < 0.0) { q01 = 0.0; q02 = 0.0; }
      else if (fabs(q01) > fabs(2.0*q02)) q01 = 2.0*q02;
      else if (fabs(q02) > fabs(2.0*q01)) q02 = 2.0*q01;
      e1[k] = S(k) - q02;
      e2[k] = S(k) + q01;
    }
  }
  for (int k = 1; k <= nk; k++) { E[k-1] = e1[k]; E[nk + k-1] = e2[k]; }
#undef S
  free(dp); free(qdpkm); free(qdpkmkp); free(dpkm2kp); free(zw1); free(zw2); free(slope_edge); free(val_edge); free(e1); free(e2);
  free(PCM_layer);
}

/* build_reconstructions_1d, MOM_remapping.F90:257-386 (schemes PCM, PLM, PLM_HYBGEN, PPM_H4, PPM_IH4, PPM_HYBGEN, WENO_HYBGEN,
 * PPM_CW, PQM_IH4IH3, PQM_IH6IH5; no PCM_cell).
 * E and coef must hold 2*n0 and 5*n0 doubles.  Returns the integration method. */
int orc_build_reconstructions_1d(int scheme, int boundary_extrapolation, int n0, const double *h0, const double *u0,
                                 double *coef, double *E, double h_neglect, double h_neglect_edge)
{
  const int n = n0;
  memset(E, 0, sizeof(double)*2*n0);
  memset(coef, 0, sizeof(double)*5*n0);
  int local = scheme;
  if (n0 <= 1) local = ORC_REMAP_PCM;
  else if (n0 <= 3) local = (local < ORC_REMAP_PLM) ? local : ORC_REMAP_PLM;
  else if (n0 <= 4 && local != ORC_REMAP_PPM_CW) local = (local < ORC_REMAP_PPM_H4) ? local : ORC_REMAP_PPM_H4;
  (void)n;
  switch (local) {
    case ORC_REMAP_PCM:
      orc_pcm_reconstruction(n0, u0, E, coef);
      return ORC_INT_PCM;
    case ORC_REMAP_PLM:
      orc_plm_reconstruction(n0, h0, u0, E, coef, h_neglect);
      if (boundary_extrapolation) orc_plm_boundary_extrapolation(n0, h0, u0, E, coef, h_neglect);
      return ORC_INT_PLM;
    case ORC_REMAP_PLM_HYBGEN:   /* :306-315 */
      orc_hybgen_plm_coefs(n0, u0, h0, coef + n0, h_neglect);
      for (int k = 0; k < n0; k++) {
        E[k] = u0[k] - 0.5 * coef[n0 + k];
        E[n0 + k] = u0[k] + 0.5 * coef[n0 + k];
        coef[k] = E[k];
      }
      if (boundary_extrapolation) orc_plm_boundary_extrapolation(n0, h0, u0, E, coef, h_neglect);
      return ORC_INT_PLM;
    case ORC_REMAP_PPM_HYBGEN:   /* :339-344 */
      orc_hybgen_ppm_coefs(n0, u0, h0, E, h_neglect);
      orc_ppm_reconstruction(n0, h0, u0, E, coef);
      if (boundary_extrapolation) orc_ppm_boundary_extrapolation(n0, h0, u0, E, coef, h_neglect);
      return ORC_INT_PPM;
    case ORC_REMAP_WENO_HYBGEN:  /* :345-350 */
      orc_hybgen_weno_coefs(n0, u0, h0, E, h_neglect);
      orc_ppm_reconstruction(n0, h0, u0, E, coef);
      if (boundary_extrapolation) orc_ppm_boundary_extrapolation(n0, h0, u0, E, coef, h_neglect);
      return ORC_INT_PPM;
    case ORC_REMAP_PPM_H4:
      orc_edge_values_explicit_h4(n0, h0, u0, E, h_neglect_edge);
      orc_ppm_reconstruction(n0, h0, u0, E, coef);
      if (boundary_extrapolation) orc_ppm_boundary_extrapolation(n0, h0, u0, E, coef, h_neglect);
      return ORC_INT_PPM;
    case ORC_REMAP_PPM_IH4:   /* :332-338 */
      orc_edge_values_implicit_h4(n0, h0, u0, E, h_neglect_edge);
      orc_ppm_reconstruction(n0, h0, u0, E, coef);
      if (boundary_extrapolation) orc_ppm_boundary_extrapolation(n0, h0, u0, E, coef, h_neglect);
      return ORC_INT_PPM;
    case ORC_REMAP_PPM_CW:    /* :316-324 */
      orc_edge_values_explicit_h4cw(n0, h0, u0, E, h_neglect_edge);
      orc_ppm_monotonicity(n0, u0, E);
      orc_ppm_reconstruction(n0, h0, u0, E, coef);
      if (boundary_extrapolation) orc_ppm_boundary_extrapolation(n0, h0, u0, E, coef, h_neglect);
      return ORC_INT_PPM;
    case ORC_REMAP_PQM_IH4IH3: {   /* :351-360 */
      double *S = (double*)calloc((size_t)2*n0, sizeof(double));
      orc_edge_values_implicit_h4(n0, h0, u0, E, h_neglect_edge);
      orc_edge_slopes_implicit_h3(n0, h0, u0, S, h_neglect);
      orc_pqm_reconstruction(n0, h0, u0, E, S, coef, h_neglect);
      if (boundary_extrapolation) orc_pqm_boundary_extrapolation_v1(n0, h0, u0, E, S, coef, h_neglect);
      free(S);
      return ORC_INT_PQM;
    }
    case ORC_REMAP_PQM_IH6IH5: {   /* :361-370 (six cells at least: the boundary systems of both routines read cells 1..6) */
      if (n0 < 6) return -998;
      double *S = (double*)calloc((size_t)2*n0, sizeof(double));
      int rc = orc_edge_values_implicit_h6(n0, h0, u0, E, h_neglect_edge);
      rc |= orc_edge_slopes_implicit_h5(n0, h0, u0, S, h_neglect);
      if (rc) { free(S); return -997; }      /* 'The linear system is singular !' */
      orc_pqm_reconstruction(n0, h0, u0, E, S, coef, h_neglect);
      if (boundary_extrapolation) orc_pqm_boundary_extrapolation_v1(n0, h0, u0, E, S, coef, h_neglect);
      free(S);
      return ORC_INT_PQM;
    }
    default:
      return -999;   /* 'The selected remapping method is invalid' */
  }
}

/* remapping_core_h, MOM_remapping.F90:160-201 */
int orc_remapping_core_h(int scheme, int boundary_extrapolation, int n0, const double *h0, const double *u0,
                         int n1, const double *h1, double *u1, double h_neglect, double h_neglect_edge)
{
  double *E = calloc((size_t)2*n0, sizeof(double)), *coef = calloc((size_t)5*n0, sizeof(double));
  int method = orc_build_reconstructions_1d(scheme, boundary_extrapolation, n0, h0, u0, coef, E, h_neglect, h_neglect_edge);
  if (method < 0) { free(E); free(coef); return 1; }
  double uh_err;
  orc_remap_via_sub_cells(n0, h0, u0, E, coef, n1, h1, method, 0, u1, &uh_err);
  free(E); free(coef);
  return 0;
}

/* dzFromH1H2, MOM_remapping.F90:1235-1256 */
void orc_dz_from_h1h2(int n1, const double *h1, int n2, const double *h2, double *dx)
{
  double x1 = 0.0, x2 = 0.0;
  int nmax = n1 > n2 ? n1 : n2;
  dx[0] = 0.0;
  for (int k = 1; k <= nmax; k++) {
    if (k <= n1) x1 = x1 + h1[k-1];
    if (k <= n2) { x2 = x2 + h2[k-1]; dx[k] = x2 - x1; }
  }
}

/* remapping_core_w, MOM_remapping.F90:205-254 */
int orc_remapping_core_w(int scheme, int boundary_extrapolation, int n0, const double *h0, const double *u0,
                         int n1, const double *dx, double *u1, double h_neglect, double h_neglect_edge)
{
  double *h1 = calloc(n1, sizeof(double));
  for (int k = 1; k <= n1; k++) {
    if (k <= n0) h1[k-1] = max2( 0., h0[k-1] + ( dx[k] - dx[k-1] ) );
    else         h1[k-1] = max2( 0., dx[k] - dx[k-1] );
  }
  int rc = orc_remapping_core_h(scheme, boundary_extrapolation, n0, h0, u0, n1, h1, u1, h_neglect, h_neglect_edge);
  free(h1);
  return rc;
}

/* ALE_remap_tracers, src/ALE/MOM_ALE.F90:737-867 */
int orc_ale_remap_tracers(const mom6hip_grid_t *G, const mom6hip_remapping_cs_t *cs, const double *h_old,
                          const double *h_new, double *const *tr, const double *conc_underflow, int ntr)
{
  const int nz = G->nk;
  /* :770-771, answer_date >= 20190101 */
  const double h_neglect = G->H_subroundoff, h_neglect_edge = G->H_subroundoff;
  int rc = 0;
  for (int m = 0; m < ntr && !rc; m++) {
    ORC_PAR      /* the columns are independent (the reference: !$OMP parallel do over j, MOM_ALE.F90:779) */
    for (int j = G->jsc; j <= G->jec; j++) {
      double h1[nz], h2[nz], col[nz], tr_column[nz];
      for (int i = G->isc; i <= G->iec; i++) {
        if (!(G->mask2dT[ORC_H2(G,i,j)] > 0.)) continue;
        for (int k = 1; k <= nz; k++) {
          h1[k-1] = h_old[ORC_H3(G,i,j,k)]; h2[k-1] = h_new[ORC_H3(G,i,j,k)]; col[k-1] = tr[m][ORC_H3(G,i,j,k)];
        }
        const int rc1 = orc_remapping_core_h(cs->remapping_scheme, cs->boundary_extrapolation, nz, h1, col, nz, h2, tr_column,
                                             h_neglect, h_neglect_edge);
        if (rc1) { rc = rc1; break; }      /* (an unknown scheme: the same for every column) */
        if (conc_underflow && conc_underflow[m] > 0.0)
          for (int k = 0; k < nz; k++) if (fabs(tr_column[k]) < conc_underflow[m]) tr_column[k] = 0.0;
        for (int k = 1; k <= nz; k++) tr[m][ORC_H3(G,i,j,k)] = tr_column[k-1];
      }
    }
  }
  return rc;
}
