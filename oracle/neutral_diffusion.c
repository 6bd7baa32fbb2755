/* oracle/neutral_diffusion.c -- TEST INFRASTRUCTURE: a C restatement of the continuous-reconstruction branch of
 * MOM_neutral_diffusion (src/tracer/MOM_neutral_diffusion.F90): neutral_diffusion_calc_coeffs :337-602, neutral_diffusion
 * :605-1019 and the column routines they call.  NDIFF_CONTINUOUS = True only (the default); no
 * NDIFF_TAPERING, no KHTR_USE_EBT_STRUCT, no NDIFF_USE_UNMASKED_TRANSPORT_BUG; NDIFF_INTERIOR_ONLY with boundary_k_range of
 * MOM_hor_bnd_diffusion.F90:609.
 * Pinned by the reference's own known answers: every case of ndiff_unit_tests_continuous (:2576-2835) is held in
 * tests/golden/neutral_diffusion.json and checked by tests/test_neutral_diffusion.py. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "mom6_oracle.h"

static inline double min2(double a, double b) { return a < b ? a : b; }
static inline double max2(double a, double b) { return a > b ? a : b; }
static inline double fsign(double a, double b) { return copysign(fabs(a), b); }
/* signum :1200 */
static inline double signum(double a, double x) { return (x == 0.) ? 0. : fsign(a, x); }

/* fv_diff :1282 */
double orc_ndiff_fv_diff(double hkm1, double hk, double hkp1, double Skm1, double Sk, double Skp1)
{
  double h_sum = (hkm1 + hkp1) + hk;
  if (h_sum != 0.) h_sum = 1. / h_sum;
  double hm = hkm1 + hk;
  if (hm != 0.) hm = 1. / hm;
  double hp = hkp1 + hk;
  if (hp != 0.) hp = 1. / hp;
  return (hk * h_sum) * ((2.*hkm1 + hk) * hp * (Skp1 - Sk) + (2.*hkp1 + hk) * hm * (Sk - Skm1));
}

/* fvlsq_slope :1313 */
double orc_ndiff_fvlsq_slope(double hkm1, double hk, double hkp1, double Skm1, double Sk, double Skp1)
{
  const double xkm1 = -0.5 * (hk + hkm1), xkp1 = 0.5 * (hk + hkp1);
  const double h_sum = (hkm1 + hkp1) + hk;
  const double hx_sum = hkm1*xkm1 + hkp1*xkp1;
  const double hxsq_sum = hkm1*(xkm1*xkm1) + hkp1*(xkp1*xkp1);
  const double hxy_sum = hkm1*xkm1*Skm1 + hkp1*xkp1*Skp1;
  const double hy_sum = (hkm1*Skm1 + hkp1*Skp1) + hk*Sk;
  const double det = h_sum * hxsq_sum - hx_sum*hx_sum;
  if (det != 0.) return (h_sum * hxy_sum - hx_sum*hy_sum) / det;
  return 0.;
}

/* PLM_diff :1211 (arrays 0-based: layer k of the reference is [k-1]) */
static void PLM_diff(int nk, const double *h, const double *S, int c_method, int b_method, double *diff)
{
  for (int k = 1; k < nk-1; k++) {
    const double hkm1 = h[k-1], hk = h[k], hkp1 = h[k+1];
    if ((hkp1 + hk) * (hkm1 + hk) > 0.) {
      const double Skm1 = S[k-1], Sk = S[k], Skp1 = S[k+1];
      double diff_c = 0.;
      if (c_method == 1) {
        if (hk + 0.5 * (hkm1 + hkp1) != 0.) diff_c = (Skp1 - Skm1) * (hk / (hk + 0.5 * (hkm1 + hkp1)));
        else diff_c = 0.;
      } else if (c_method == 2) {
        diff_c = orc_ndiff_fv_diff(hkm1, hk, hkp1, Skm1, Sk, Skp1);
      } else if (c_method == 3) {
        diff_c = hk * orc_ndiff_fvlsq_slope(hkm1, hk, hkp1, Skm1, Sk, Skp1);
      }
      const double diff_l = 2. * (Sk - Skm1), diff_r = 2. * (Skp1 - Sk);
      if (signum(1., diff_l) * signum(1., diff_r) <= 0.) diff[k] = 0.;
      else diff[k] = fsign(min2(min2(fabs(diff_l), fabs(diff_c)), fabs(diff_r)), diff_c);
    } else {
      diff[k] = 0.;
    }
  }
  if (b_method == 1) {
    diff[0] = 0.; diff[nk-1] = 0.;
  } else if (b_method == 2) {      /* as written in the reference (:1274-1275) */
    diff[0] = (S[1] - S[0]) * 2. * (h[0] / (h[0] + h[1]));
    diff[nk-1] = S[nk-1] - S[nk-2] * 2. * (h[nk-1] / (h[nk-2] + h[nk-1]));
  }
}

/* ppm_edge :1120 */
static double ppm_edge(double hkm1, double hk, double hkp1, double hkp2, double Ak, double Akp1, double Pk, double Pkp1,
                       double h_neglect)
{
  double R_hk_hkp1 = hk + hkp1, e;
  if (R_hk_hkp1 <= 0.) return 0.5 * (Ak + Akp1);
  R_hk_hkp1 = 1. / R_hk_hkp1;
  if (hk < hkp1) e = Ak + (hk * R_hk_hkp1) * (Akp1 - Ak);
  else e = Akp1 + (hkp1 * R_hk_hkp1) * (Ak - Akp1);
  const double R_2hk_hkp1 = 1. / ((2. * hk + hkp1) + h_neglect);
  const double R_hk_2hkp1 = 1. / ((hk + 2. * hkp1) + h_neglect);
  const double f1 = 1. / ((hk + hkp1) + (hkm1 + hkp2));
  const double f2 = 2. * (hkp1 * hk) * R_hk_hkp1 * ((hkm1 + hk) * R_2hk_hkp1 - (hkp2 + hkp1) * R_hk_2hkp1);
  const double f3 = hk * (hkm1 + hk) * R_2hk_hkp1;
  const double f4 = hkp1 * (hkp1 + hkp2) * R_hk_2hkp1;
  return e + f1 * (f2 * (Akp1 - Ak) - (f3 * Pkp1 - f4 * Pk));
}

/* interface_scalar :1078 ; Si has nk+1 entries */
void orc_ndiff_interface_scalar(int nk, const double *h, const double *S, double *Si, int i_method, double h_neglect)
{
  double *diff = (double*)calloc((size_t)nk, sizeof(double));
  PLM_diff(nk, h, S, 2, 1, diff);
  Si[0] = S[0] - 0.5 * diff[0];
  if (i_method == 1) {
    for (int k = 1; k < nk; k++) {
      const double Sa = S[k-1] + 0.5 * diff[k-1];
      const double Sb = S[k] - 0.5 * diff[k];
      Si[k] = 0.5 * (Sa + Sb);
    }
  } else if (i_method == 2) {
    for (int k = 1; k < nk; k++) {      /* the reference's k = this k + 1 */
      const int km2 = (k-2 > 0) ? k-2 : 0, kp1 = (k+1 < nk-1) ? k+1 : nk-1;
      Si[k] = ppm_edge(h[km2], h[k-1], h[k], h[kp1], S[k-1], S[k], diff[k-1], diff[k], h_neglect);
    }
  }
  Si[nk] = S[nk-1] + 0.5 * diff[nk-1];
  free(diff);
}

/* ppm_left_right_edge_values :2541 */
static void ppm_left_right_edge_values(int nk, const double *Tl, const double *Ti, double *aL, double *aR)
{
  for (int k = 0; k < nk; k++) {
    aL[k] = Ti[k]; aR[k] = Ti[k+1];
    if (signum(1., aR[k] - Tl[k]) * signum(1., Tl[k] - aL[k]) <= 0.0) {
      aL[k] = Tl[k]; aR[k] = Tl[k];
    } else if (fsign(3., aR[k] - aL[k]) * ((Tl[k] - aL[k]) + (Tl[k] - aR[k])) > fabs(aR[k] - aL[k])) {
      aL[k] = Tl[k] + 2.0 * (Tl[k] - aR[k]);
    } else if (fsign(3., aR[k] - aL[k]) * ((Tl[k] - aL[k]) + (Tl[k] - aR[k])) < -fabs(aR[k] - aL[k])) {
      aR[k] = Tl[k] + 2.0 * (Tl[k] - aL[k]);
    }
  }
}

/* ppm_ave :1166 ; *bad is set where the reference stops */
static double ppm_ave(double xL, double xR, double aL, double aR, double aMean, int *bad)
{
  const double dx = xR - xL;
  const double xave = 0.5 * (xR + xL);
  const double a6o3 = 2. * aMean - (aL + aR);
  const double a6 = 3. * a6o3;
  if (dx < 0. || dx > 1.) { *bad = 1; return 0.; }
  if (dx == 0.) return aL + (aR - aL) * xR + a6 * xR * (1. - xR);
  return (aL + xave * ((aR - aL) + a6)) - a6o3 * (xR*xR + xR * xL + xL*xL);
}

/* interpolate_for_nondim_position :1563 */
double orc_ndiff_interpolate_for_nondim_position(double dRhoNeg, double Pneg, double dRhoPos, double Ppos)
{
  if (Ppos <= Pneg) return 0.5;
  if (dRhoPos - dRhoNeg > 0.) return min2(1., max2(0., -dRhoNeg / (dRhoPos - dRhoNeg)));
  if (dRhoPos - dRhoNeg == 0) {
    if (dRhoNeg > 0.) return 0.;
    if (dRhoNeg < 0.) return 1.;
    return 0.5;
  }
  return 0.5;
}

/* absolute_position :2258 (k_surface 0-based, Karr holds the reference's 1-based layer index) */
static double absolute_position(const double *Pint, const int *Karr, const double *NParr, int k_surface)
{
  const int k = Karr[k_surface] - 1;
  return Pint[k] + NParr[k_surface] * (Pint[k+1] - Pint[k]);
}

/* find_neutral_surface_positions_continuous :1353 (without the optional boundary-layer limits).
 * Columns of nk+1 interface values; PoL, PoR, KoL, KoR hold 2*nk+2 entries, hEff 2*nk+1; KoL / KoR are 1-based as in the
 * reference. */
static void nsp_continuous(int nk, const double *Pl, const double *Tl, const double *Sl,
    const double *dRdTl, const double *dRdSl, const double *Pr, const double *Tr, const double *Sr, const double *dRdTr,
    const double *dRdSr, double *PoL, double *PoR, int *KoL, int *KoR, double *hEff, int interior_limit, int bl_kl, int bl_kr,
    double bl_zl, double bl_zr);

void orc_ndiff_find_neutral_surface_positions_continuous(int nk, const double *Pl, const double *Tl, const double *Sl,
    const double *dRdTl, const double *dRdSl, const double *Pr, const double *Tr, const double *Sr, const double *dRdTr,
    const double *dRdSr, double *PoL, double *PoR, int *KoL, int *KoR, double *hEff)
{
  nsp_continuous(nk, Pl, Tl, Sl, dRdTl, dRdSl, Pr, Tr, Sr, dRdTr, dRdSr, PoL, PoR, KoL, KoR, hEff, 0, 0, 0, 0., 0.);
}

/* boundary_k_range for the SURFACE boundary layer, src/tracer/MOM_hor_bnd_diffusion.F90:609-647 (k_bot 1-based) */
void orc_ndiff_boundary_k_range_surface(int nk, const double *h, double hbl, int *k_bot, double *zeta_bot)
{
  double htot = 0., hsum = 0.;
  *k_bot = 1; *zeta_bot = 0.;
  if (hbl == 0.) return;
  for (int k = 0; k < nk; k++) hsum = hsum + h[k];
  if (hbl >= hsum) { *k_bot = nk; *zeta_bot = 1.; return; }
  for (int k = 0; k < nk; k++) {
    htot = htot + h[k];
    if (htot >= hbl) { *k_bot = k+1; *zeta_bot = 1 - (htot - hbl)/h[k]; return; }
  }
}

/* with the optional boundary-layer limits bl_kl, bl_kr, bl_zl, bl_zr of the reference (:1508-1521) when interior_limit */
static void nsp_continuous(int nk, const double *Pl, const double *Tl, const double *Sl,
    const double *dRdTl, const double *dRdSl, const double *Pr, const double *Tr, const double *Sr, const double *dRdTr,
    const double *dRdSr, double *PoL, double *PoR, int *KoL, int *KoR, double *hEff, int interior_limit, int bl_kl, int bl_kr,
    double bl_zl, double bl_zr)
{
  const int ns = 2*nk + 2;
  int kr = 1, kl = 1, lastK_right = 1, lastK_left = 1;
  double lastP_right = 0., lastP_left = 0.;
  int reached_bottom = 0, searching_left_column = 0, searching_right_column = 0;
#define L(a,k) a[(k)-1]
  for (int ks = 0; ks < ns; ks++) {
    int klm1 = (kl-1 > 1) ? kl-1 : 1;
    int krm1 = (kr-1 > 1) ? kr-1 : 1;
    const double dRho = 0.5 * ((L(dRdTr,kr) + L(dRdTl,kl)) * (L(Tr,kr) - L(Tl,kl)) + (L(dRdSr,kr) + L(dRdSl,kl)) * (L(Sr,kr) - L(Sl,kl)));
    if (!reached_bottom) {
      if (dRho < 0.) { searching_left_column = 1; searching_right_column = 0; }
      else if (dRho > 0.) { searching_right_column = 1; searching_left_column = 0; }
      else {
        if (kl + kr == 2) { searching_left_column = 1; searching_right_column = 0; }
        else { searching_left_column = !searching_left_column; searching_right_column = !searching_right_column; }
      }
    }
    if (searching_left_column) {
      const double dRhoTop = 0.5 * ((L(dRdTl,klm1) + L(dRdTr,kr)) * (L(Tl,klm1) - L(Tr,kr)) + (L(dRdSl,klm1) + L(dRdSr,kr)) * (L(Sl,klm1) - L(Sr,kr)));
      const double dRhoBot = 0.5 * ((L(dRdTl,klm1+1) + L(dRdTr,kr)) * (L(Tl,klm1+1) - L(Tr,kr)) + (L(dRdSl,klm1+1) + L(dRdSr,kr)) * (L(Sl,klm1+1) - L(Sr,kr)));
      if (dRhoTop > 0. || kr+kl == 2) PoL[ks] = 0.;
      else if (dRhoTop >= dRhoBot) PoL[ks] = 1.;
      else PoL[ks] = orc_ndiff_interpolate_for_nondim_position(dRhoTop, L(Pl,klm1), dRhoBot, L(Pl,klm1+1));
      if (PoL[ks] >= 1. && klm1 < nk) { klm1 = klm1 + 1; PoL[ks] = PoL[ks] - 1.; }
      if ((double)(klm1-lastK_left) + (PoL[ks]-lastP_left) < 0.) { PoL[ks] = lastP_left; klm1 = lastK_left; }
      KoL[ks] = klm1;
      if (kr <= nk) { PoR[ks] = 0.; KoR[ks] = kr; } else { PoR[ks] = 1.; KoR[ks] = nk; }
      if (kr <= nk) kr = kr + 1;
      else { reached_bottom = 1; searching_right_column = 1; searching_left_column = 0; }
    } else {      /* searching_right_column */
      const double dRhoTop = 0.5 * ((L(dRdTr,krm1) + L(dRdTl,kl)) * (L(Tr,krm1) - L(Tl,kl)) + (L(dRdSr,krm1) + L(dRdSl,kl)) * (L(Sr,krm1) - L(Sl,kl)));
      const double dRhoBot = 0.5 * ((L(dRdTr,krm1+1) + L(dRdTl,kl)) * (L(Tr,krm1+1) - L(Tl,kl)) + (L(dRdSr,krm1+1) + L(dRdSl,kl)) * (L(Sr,krm1+1) - L(Sl,kl)));
      if (dRhoTop >= 0. || kr+kl == 2) PoR[ks] = 0.;
      else if (dRhoTop >= dRhoBot) PoR[ks] = 1.;
      else PoR[ks] = orc_ndiff_interpolate_for_nondim_position(dRhoTop, L(Pr,krm1), dRhoBot, L(Pr,krm1+1));
      if (PoR[ks] >= 1. && krm1 < nk) { krm1 = krm1 + 1; PoR[ks] = PoR[ks] - 1.; }
      if ((double)(krm1-lastK_right) + (PoR[ks]-lastP_right) < 0.) { PoR[ks] = lastP_right; krm1 = lastK_right; }
      KoR[ks] = krm1;
      if (kl <= nk) { PoL[ks] = 0.; KoL[ks] = kl; } else { PoL[ks] = 1.; KoL[ks] = nk; }
      if (kl <= nk) kl = kl + 1;
      else { reached_bottom = 1; searching_right_column = 0; searching_left_column = 1; }
    }
    if (interior_limit) {
      if (KoL[ks] <= bl_kl) { KoL[ks] = bl_kl; if (PoL[ks] < bl_zl) PoL[ks] = bl_zl; }
      if (KoR[ks] <= bl_kr) { KoR[ks] = bl_kr; if (PoR[ks] < bl_zr) PoR[ks] = bl_zr; }
    }
    lastK_left = KoL[ks]; lastP_left = PoL[ks];
    lastK_right = KoR[ks]; lastP_right = PoR[ks];
    if (ks > 0) {
      const double hL = absolute_position(Pl, KoL, PoL, ks) - absolute_position(Pl, KoL, PoL, ks-1);
      const double hR = absolute_position(Pr, KoR, PoR, ks) - absolute_position(Pr, KoR, PoR, ks-1);
      if (hL + hR > 0.) hEff[ks-1] = 2. * hL * hR / (hL + hR);
      else hEff[ks-1] = 0.;
    }
  }
#undef L
}

/* neutral_surface_flux :2297, continuous, without tapering.  Returns 1 where the reference's ppm_ave stops. */
int orc_ndiff_neutral_surface_flux(int nk, const double *hl, const double *hr, const double *Tl, const double *Tr,
                                   const double *PiL, const double *PiR, const int *KoL, const int *KoR, const double *hEff,
                                   double *Flx, double h_neglect)
{
  const int nsurf = 2*nk + 2;
  int bad = 0;
  double *w = (double*)calloc((size_t)6*nk + 2, sizeof(double));
  double *Til = w, *Tir = Til + (nk+1), *aL_l = Tir + (nk+1), *aR_l = aL_l + nk, *aL_r = aR_l + nk, *aR_r = aL_r + nk;
  orc_ndiff_interface_scalar(nk, hl, Tl, Til, 2, h_neglect);
  orc_ndiff_interface_scalar(nk, hr, Tr, Tir, 2, h_neglect);
  ppm_left_right_edge_values(nk, Tl, Til, aL_l, aR_l);
  ppm_left_right_edge_values(nk, Tr, Tir, aL_r, aR_r);
  for (int ks = 0; ks < nsurf-1; ks++) {
    if (hEff[ks] == 0.) { Flx[ks] = 0.; continue; }
    const int klb = KoL[ks+1] - 1, klt = KoL[ks] - 1, krb = KoR[ks+1] - 1, krt = KoR[ks] - 1;
    const double T_left_bottom = (1. - PiL[ks+1]) * Til[klb] + PiL[ks+1] * Til[klb+1];
    const double T_left_top = (1. - PiL[ks]) * Til[klt] + PiL[ks] * Til[klt+1];
    const double T_left_layer = ppm_ave(PiL[ks], PiL[ks+1] + (double)(klb-klt), aL_l[klt], aR_l[klt], Tl[klt], &bad);
    const double T_right_bottom = (1. - PiR[ks+1]) * Tir[krb] + PiR[ks+1] * Tir[krb+1];
    const double T_right_top = (1. - PiR[ks]) * Tir[krt] + PiR[ks] * Tir[krt+1];
    const double T_right_layer = ppm_ave(PiR[ks], PiR[ks+1] + (double)(krb-krt), aL_r[krt], aR_r[krt], Tr[krt], &bad);
    const double dT_top = T_right_top - T_left_top;
    const double dT_bottom = T_right_bottom - T_left_bottom;
    double dT_ave = 0.5 * (dT_top + dT_bottom);
    const double dT_layer = T_right_layer - T_left_layer;
    if (signum(1., dT_top) * signum(1., dT_bottom) <= 0. || signum(1., dT_ave) * signum(1., dT_layer) <= 0.) dT_ave = 0.;
    else dT_ave = dT_layer;
    Flx[ks] = dT_ave * hEff[ks] * 1.0;
  }
  free(w);
  return bad;
}

/* ---- the 3-D routines ------------------------------------------------------------------------------ */

typedef struct {
  int nsurf;
  double *Pint, *Tint, *Sint, *dRdT, *dRdS;      /* [(nk+1)][h points] */
  double *uPoL, *uPoR, *uhEff, *vPoL, *vPoR, *vhEff;   /* [nsurf][faces] */
  int *uKoL, *uKoR, *vKoL, *vKoR;
} nd_work_t;

static void gather(const double *a, size_t n2, size_t stride, int n, double *col) { for (int k = 0; k < n; k++) col[k] = a[n2 + stride*k]; }

/* neutral_diffusion_calc_coeffs :337 */
static void nd_calc_coeffs(const mom6hip_grid_t *G, const mom6hip_neutral_diffusion_cs_t *ND, const mom6hip_eos_t *eos,
                           const double *h, const double *T, const double *S, const double *p_surf, const double *h_ML, nd_work_t *W)
{
  const int is = G->isc, ie = G->iec, js = G->jsc, je = G->jec, nz = G->nk;
  const int isd = G->isd, jsd = G->jsd;
  const int nih = G->ied - G->isd + 1, njh = G->jed - G->jsd + 1;
  const size_t hpl = (size_t)nih*njh, upl = (size_t)(nih+1)*njh, vpl = (size_t)nih*(njh+1);
#define H2(i,j) ((size_t)((j)-jsd)*nih + ((i)-isd))
#define U2(I,j) ((size_t)((j)-jsd)*(nih+1) + ((I)-isd+1))
#define V2(i,J) ((size_t)((J)-jsd+1)*nih + ((i)-isd))
  const double h_neglect = G->H_subroundoff;
  const double pa_to_H = 1. / (ND->H_to_RZ * G->g_Earth);
  const int ns = W->nsurf;
  memset(W->dRdT, 0, sizeof(double)*hpl*(nz+1)); memset(W->dRdS, 0, sizeof(double)*hpl*(nz+1));
  /* :370-390: the boundary-layer depth, its halo, and the layer / fraction it ends in for every wet column */
  int *k_bot = (int*)calloc(hpl, sizeof(int));
  double *zeta_bot = (double*)calloc(hpl, sizeof(double)), *hbl = (double*)calloc(hpl, sizeof(double));
  for (size_t q = 0; q < hpl; q++) k_bot[q] = 1;
  if (ND->interior_only) {
    memcpy(hbl, h_ML, sizeof(double)*hpl);
    orc_halo_update(G, hbl, MOM6HIP_POS_H, 1);
    double *hcol = (double*)calloc((size_t)nz, sizeof(double));
    for (int j = js-1; j <= je+1; j++) for (int i = is-1; i <= ie+1; i++) if (G->mask2dT[H2(i,j)] > 0.0) {
      gather(h, H2(i,j), hpl, nz, hcol);
      orc_ndiff_boundary_k_range_surface(nz, hcol, hbl[H2(i,j)], &k_bot[H2(i,j)], &zeta_bot[H2(i,j)]);
    }
    free(hcol);
  }
  double *hc = (double*)calloc((size_t)8*(nz+1), sizeof(double));
  double *Tc = hc + (nz+1), *Sc = Tc + (nz+1), *Ti = Sc + (nz+1), *Si = Ti + (nz+1);
  for (int j = js-1; j <= je+1; j++) for (int i = is-1; i <= ie+1; i++) {
    const size_t n2 = H2(i,j);
    W->Pint[n2] = p_surf ? p_surf[n2] : 0.;
    for (int k = 0; k < nz; k++) W->Pint[n2 + hpl*(k+1)] = W->Pint[n2 + hpl*k] + h[n2 + hpl*k]*(G->g_Earth*ND->H_to_RZ);
    gather(h, n2, hpl, nz, hc); gather(T, n2, hpl, nz, Tc); gather(S, n2, hpl, nz, Sc);
    orc_ndiff_interface_scalar(nz, hc, Tc, Ti, 2, h_neglect);
    orc_ndiff_interface_scalar(nz, hc, Sc, Si, 2, h_neglect);
    for (int K = 0; K <= nz; K++) {
      W->Tint[n2 + hpl*K] = Ti[K]; W->Sint[n2 + hpl*K] = Si[K];
      const double pres = (ND->ref_pres >= 0.) ? ND->ref_pres : W->Pint[n2 + hpl*K];
      orc_eos_density_derivs(eos, Ti[K], Si[K], pres, &W->dRdT[n2 + hpl*K], &W->dRdS[n2 + hpl*K]);
    }
  }
  free(hc);
  for (size_t q = 0; q < upl*ns; q++) { W->uPoL[q] = 0.; W->uPoR[q] = 0.; W->uKoL[q] = 1; W->uKoR[q] = 1; }
  for (size_t q = 0; q < vpl*ns; q++) { W->vPoL[q] = 0.; W->vPoR[q] = 0.; W->vKoL[q] = 1; W->vKoR[q] = 1; }
  memset(W->uhEff, 0, sizeof(double)*upl*ns); memset(W->vhEff, 0, sizeof(double)*vpl*ns);
  double *c = (double*)calloc((size_t)10*(nz+1) + 3*(size_t)ns, sizeof(double));
  double *PoL = c + 10*(nz+1), *PoR = PoL + ns, *hE = PoR + ns;
  int *Ko = (int*)calloc((size_t)2*ns, sizeof(int));
  for (int dir = 0; dir < 2; dir++)
    for (int j = (dir ? js-1 : js); j <= je; j++) for (int i = (dir ? is : is-1); i <= ie; i++) {
      const size_t f = dir ? V2(i,j) : U2(i,j), pl = dir ? vpl : upl;
      if (!((dir ? G->mask2dCv[f] : G->mask2dCu[f]) > 0.0)) continue;
      const size_t cl = H2(i,j), cr = dir ? H2(i,j+1) : H2(i+1,j);
      const double *src[5] = {W->Pint, W->Tint, W->Sint, W->dRdT, W->dRdS};
      for (int q = 0; q < 5; q++) { gather(src[q], cl, hpl, nz+1, c + q*(nz+1)); gather(src[q], cr, hpl, nz+1, c + (5+q)*(nz+1)); }
      nsp_continuous(nz, c, c + (nz+1), c + 2*(nz+1), c + 3*(nz+1), c + 4*(nz+1),
          c + 5*(nz+1), c + 6*(nz+1), c + 7*(nz+1), c + 8*(nz+1), c + 9*(nz+1), PoL, PoR, Ko, Ko + ns, hE,
          ND->interior_only, k_bot[cl], k_bot[cr], zeta_bot[cl], zeta_bot[cr]);
      double *oPoL = dir ? W->vPoL : W->uPoL, *oPoR = dir ? W->vPoR : W->uPoR, *ohE = dir ? W->vhEff : W->uhEff;
      int *oKoL = dir ? W->vKoL : W->uKoL, *oKoR = dir ? W->vKoR : W->uKoR;
      for (int ks = 0; ks < ns; ks++) {
        oPoL[f + pl*ks] = PoL[ks]; oPoR[f + pl*ks] = PoR[ks]; oKoL[f + pl*ks] = Ko[ks]; oKoR[f + pl*ks] = Ko[ns + ks];
        if (ks < ns-1) ohE[f + pl*ks] = hE[ks] * pa_to_H;      /* :570-575 */
      }
    }
  free(c); free(Ko); free(k_bot); free(zeta_bot); free(hbl);
#undef H2
#undef U2
#undef V2
}

/* neutral_diffusion :605 for one tracer; Coef_x, Coef_y are the 2-D level 1 of the reference's arrays */
static int nd_apply(const mom6hip_grid_t *G, const mom6hip_neutral_diffusion_cs_t *ND, const double *h, const double *Coef_x,
                    const double *Coef_y, double *t, double conc_underflow, const nd_work_t *W)
{
  const int is = G->isc, ie = G->iec, js = G->jsc, je = G->jec, nz = G->nk;
  const int isd = G->isd, jsd = G->jsd;
  const int nih = G->ied - G->isd + 1, njh = G->jed - G->jsd + 1;
  const size_t hpl = (size_t)nih*njh, upl = (size_t)(nih+1)*njh, vpl = (size_t)nih*(njh+1);
#define H2(i,j) ((size_t)((j)-jsd)*nih + ((i)-isd))
#define U2(I,j) ((size_t)((j)-jsd)*(nih+1) + ((I)-isd+1))
#define V2(i,J) ((size_t)((J)-jsd+1)*nih + ((i)-isd))
  const double h_neglect = G->H_subroundoff;
  const int ns = W->nsurf;
  int bad = 0;
  double *uFlx = (double*)calloc(upl*(ns-1), sizeof(double)), *vFlx = (double*)calloc(vpl*(ns-1), sizeof(double));
  double *c = (double*)calloc((size_t)4*nz + 4*(size_t)ns, sizeof(double));
  double *hl = c, *hr = hl + nz, *Tl = hr + nz, *Tr = Tl + nz, *PiL = Tr + nz, *PiR = PiL + ns, *hE = PiR + ns, *Fl = hE + ns;
  int *Ko = (int*)calloc((size_t)2*ns, sizeof(int));
  for (int dir = 0; dir < 2; dir++)
    for (int j = (dir ? js-1 : js); j <= je; j++) for (int i = (dir ? is : is-1); i <= ie; i++) {
      const size_t f = dir ? V2(i,j) : U2(i,j), pl = dir ? vpl : upl;
      if (!((dir ? G->mask2dCv[f] : G->mask2dCu[f]) > 0.0)) continue;
      const size_t cl = H2(i,j), cr = dir ? H2(i,j+1) : H2(i+1,j);
      gather(h, cl, hpl, nz, hl); gather(h, cr, hpl, nz, hr); gather(t, cl, hpl, nz, Tl); gather(t, cr, hpl, nz, Tr);
      gather(dir ? W->vPoL : W->uPoL, f, pl, ns, PiL); gather(dir ? W->vPoR : W->uPoR, f, pl, ns, PiR);
      gather(dir ? W->vhEff : W->uhEff, f, pl, ns-1, hE);
      for (int ks = 0; ks < ns; ks++) { Ko[ks] = (dir ? W->vKoL : W->uKoL)[f + pl*ks]; Ko[ns+ks] = (dir ? W->vKoR : W->uKoR)[f + pl*ks]; }
      bad |= orc_ndiff_neutral_surface_flux(nz, hl, hr, Tl, Tr, PiL, PiR, Ko, Ko + ns, hE, Fl, h_neglect);
      for (int ks = 0; ks < ns-1; ks++) (dir ? vFlx : uFlx)[f + pl*ks] = Fl[ks];
    }
  double *dT = (double*)calloc((size_t)5*nz, sizeof(double));
  double *dN = dT + nz, *dS = dN + nz, *dE = dS + nz, *dW = dE + nz;
  for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
    if (!(G->mask2dT[H2(i,j)] > 0.)) continue;
    const size_t uE = U2(i,j), uW = U2(i-1,j), vN = V2(i,j), vS = V2(i,j-1);
    if (ND->ndiff_answer_date <= 20240330) {      /* :927-938 */
      for (int k = 0; k < nz; k++) dT[k] = 0.;
      for (int ks = 0; ks < ns-1; ks++) {
        int k = W->uKoL[uE + upl*ks] - 1;
        dT[k] = dT[k] + Coef_x[uE] * uFlx[uE + upl*ks];
        k = W->uKoR[uW + upl*ks] - 1;
        dT[k] = dT[k] - Coef_x[uW] * uFlx[uW + upl*ks];
        k = W->vKoL[vN + vpl*ks] - 1;
        dT[k] = dT[k] + Coef_y[vN] * vFlx[vN + vpl*ks];
        k = W->vKoR[vS + vpl*ks] - 1;
        dT[k] = dT[k] - Coef_y[vS] * vFlx[vS + vpl*ks];
      }
    } else {      /* :939-954 */
      for (int k = 0; k < nz; k++) { dN[k] = 0.; dS[k] = 0.; dE[k] = 0.; dW[k] = 0.; }
      for (int ks = 0; ks < ns-1; ks++) {
        int k = W->uKoL[uE + upl*ks] - 1;
        dE[k] = dE[k] + Coef_x[uE] * uFlx[uE + upl*ks];
        k = W->uKoR[uW + upl*ks] - 1;
        dW[k] = dW[k] - Coef_x[uW] * uFlx[uW + upl*ks];
        k = W->vKoL[vN + vpl*ks] - 1;
        dN[k] = dN[k] + Coef_y[vN] * vFlx[vN + vpl*ks];
        k = W->vKoR[vS + vpl*ks] - 1;
        dS[k] = dS[k] - Coef_y[vS] * vFlx[vS + vpl*ks];
      }
      for (int k = 0; k < nz; k++) dT[k] = (dN[k] + dS[k]) + (dE[k] + dW[k]);
    }
    for (int k = 0; k < nz; k++) {
      double *tp = t + H2(i,j) + hpl*k;
      *tp = *tp + dT[k] * (G->IareaT[H2(i,j)] / (h[H2(i,j) + hpl*k] + G->H_subroundoff));
      if (fabs(*tp) < conc_underflow) *tp = 0.0;
    }
  }
  free(dT); free(c); free(Ko); free(uFlx); free(vFlx);
#undef H2
#undef U2
#undef V2
  return bad;
}

/* the neutral branch of tracer_hordiff, src/tracer/MOM_tracer_hor_diff.F90:474-534: called by orc_tracer_hordiff_neutral
 * (tracer_hor_diff.c) with khdt_x, khdt_y and the iteration count it has formed.  Returns 0, or 4 where ppm_ave stops. */
int orc_neutral_branch(const mom6hip_grid_t *G, const mom6hip_neutral_diffusion_cs_t *ND, const mom6hip_eos_t *eos,
                       const double *h, const double *p_surf, const double *h_ML, const double *khdt_x, const double *khdt_y, int num_itts,
                       double I_numitts, double *const *tr, const double *conc_underflow, int ntr, int idx_T, int idx_S,
                       int *halo_updates)
{
  const int nz = G->nk, nih = G->ied - G->isd + 1, njh = G->jed - G->jsd + 1;
  const size_t hpl = (size_t)nih*njh, upl = (size_t)(nih+1)*njh, vpl = (size_t)nih*(njh+1);
  nd_work_t W;
  const int ns = W.nsurf = 2*nz + 2;
  W.Pint = (double*)calloc(5*hpl*(nz+1), sizeof(double));
  W.Tint = W.Pint + hpl*(nz+1); W.Sint = W.Tint + hpl*(nz+1); W.dRdT = W.Sint + hpl*(nz+1); W.dRdS = W.dRdT + hpl*(nz+1);
  W.uPoL = (double*)calloc(3*upl*ns, sizeof(double)); W.uPoR = W.uPoL + upl*ns; W.uhEff = W.uPoR + upl*ns;
  W.vPoL = (double*)calloc(3*vpl*ns, sizeof(double)); W.vPoR = W.vPoL + vpl*ns; W.vhEff = W.vPoR + vpl*ns;
  W.uKoL = (int*)calloc(2*upl*ns, sizeof(int)); W.uKoR = W.uKoL + upl*ns;
  W.vKoL = (int*)calloc(2*vpl*ns, sizeof(int)); W.vKoR = W.vKoL + vpl*ns;
  double *Coef_x = (double*)calloc(upl, sizeof(double)), *Coef_y = (double*)calloc(vpl, sizeof(double));
  int bad = 0;
  for (int m = 0; m < ntr; m++) orc_halo_update(G, tr[m], MOM6HIP_POS_H, nz);      /* do_group_pass(CS%pass_t) :478 */
  (*halo_updates)++;
  nd_calc_coeffs(G, ND, eos, h, tr[idx_T], tr[idx_S], p_surf, h_ML, &W);
  for (size_t q = 0; q < upl; q++) Coef_x[q] = I_numitts * khdt_x[q];           /* :489-503 */
  for (size_t q = 0; q < vpl; q++) Coef_y[q] = I_numitts * khdt_y[q];
  for (int itt = 1; itt <= num_itts; itt++) {
    if (itt > 1) {
      for (int m = 0; m < ntr; m++) orc_halo_update(G, tr[m], MOM6HIP_POS_H, nz);
      (*halo_updates)++;
      if (ND->recalc_neutral_surf) nd_calc_coeffs(G, ND, eos, h, tr[idx_T], tr[idx_S], p_surf, h_ML, &W);
    }
    for (int m = 0; m < ntr; m++)
      bad |= nd_apply(G, ND, h, Coef_x, Coef_y, tr[m], conc_underflow ? conc_underflow[m] : 0.0, &W);
  }
  free(W.Pint); free(W.uPoL); free(W.vPoL); free(W.uKoL); free(W.vKoL); free(Coef_x); free(Coef_y);
  return bad ? 4 : 0;
}
