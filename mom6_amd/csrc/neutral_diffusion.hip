// neutral_diffusion.hip -- the continuous-reconstruction branch of MOM_neutral_diffusion (src/tracer/MOM_neutral_diffusion.F90)
// as gfx950 kernels, called from tracer_hordiff's neutral branch (MOM_tracer_hor_diff.F90:474-534; tracer_hor_diff.hip).
//
//   nd_column_kernel      neutral_diffusion_calc_coeffs :337-470 for one h column (halo 1): interface pressures, interface T and S by
//                         interface_scalar (:1078, PLM_diff + ppm_edge), their density derivatives -> five [nk+1] columns
//   nd_surfaces_kernel    find_neutral_surface_positions_continuous (:1353) for one face: the merge walk down the 2nk+2 interfaces of
//                         the two columns; PoL, PoR, KoL, KoR, hEff (back in H units, :570-575) as [surface][face] planes
//   nd_tracer_cols_kernel interface values of a tracer (neutral_surface_flux :2373-2374) for one h column; the limited PPM edge values of a
//                         cell (ppm_left_right_edge_values :2541) are formed in the flux kernel from the two interface values it has
//                         read anyway and the cell mean, when a surface enters the cell
//   nd_flux_kernel        neutral_surface_flux (:2297) for one face -> Flx [surface][face]; the surfaces of the face are read once for a
//                         batch of up to 4 tracers (their fluxes are independent of each other)
//   nd_update_kernel      the tendencies of a cell from its four faces in the reference's order of the surfaces (:927-954), accumulated
//                         per layer in LDS (the layer index is data: KoL / KoR), and the update of the tracer (:955-959)
// A lane owns a column or a face and walks it: the walks are serial in the surface index and differ from lane to lane, the planes
// keep the lanes' accesses to one surface coalesced.  Traffic per call: ~30 (nk+1) doubles per column for the surfaces, and per tracer
// ~5 (2nk+2) doubles per face.
#include <cfloat>
#include <cmath>

#include "common.hpp"
#include "eos.hpp"

namespace {

using m6::max2;
using m6::min2;
using namespace m6::eos;

__device__ __forceinline__ double fsign(double a, double b) { return copysign(fabs(a), b); }
__device__ __forceinline__ double signum(double a, double x) { return (x == 0.) ? 0. : fsign(a, x); }      // :1200

// fv_diff :1282
__device__ __forceinline__ double fv_diff(double hkm1, double hk, double hkp1, double Skm1, double Sk, double Skp1) {
  double h_sum = (hkm1 + hkp1) + hk;
  if (h_sum != 0.) h_sum = 1. / h_sum;
  double hm = hkm1 + hk;
  if (hm != 0.) hm = 1. / hm;
  double hp = hkp1 + hk;
  if (hp != 0.) hp = 1. / hp;
  return (hk * h_sum) * ((2. * hkm1 + hk) * hp * (Skp1 - Sk) + (2. * hkp1 + hk) * hm * (Sk - Skm1));
}

// PLM_diff :1211 with c_method = 2, b_method = 1, for layer k (0-based) of the column at n2
__device__ __forceinline__ double plm_diff(const double *__restrict__ h, const double *__restrict__ S, long n2, long hpl, int k, int nk) {
  if (k <= 0 || k >= nk - 1) return 0.;
  const double hkm1 = h[n2 + hpl * (k - 1)], hk = h[n2 + hpl * k], hkp1 = h[n2 + hpl * (k + 1)];
  if (!((hkp1 + hk) * (hkm1 + hk) > 0.)) return 0.;
  const double Skm1 = S[n2 + hpl * (k - 1)], Sk = S[n2 + hpl * k], Skp1 = S[n2 + hpl * (k + 1)];
  const double diff_c = fv_diff(hkm1, hk, hkp1, Skm1, Sk, Skp1);
  const double diff_l = 2. * (Sk - Skm1), diff_r = 2. * (Skp1 - Sk);
  if (signum(1., diff_l) * signum(1., diff_r) <= 0.) return 0.;
  return fsign(min2(min2(fabs(diff_l), fabs(diff_c)), fabs(diff_r)), diff_c);
}

// ppm_edge :1120
__device__ __forceinline__ double ppm_edge(double hkm1, double hk, double hkp1, double hkp2, double Ak, double Akp1, double Pk, double Pkp1,
                                           double h_neglect) {
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

// interface_scalar :1078 with i_method = 2, walked down the column: call with K = 0 .. nk in order; d_prev carries diff(K-1)
__device__ __forceinline__ double interface_value(const double *__restrict__ h, const double *__restrict__ S, long n2, long hpl, int K, int nk,
                                                  double h_neglect, double &d_prev) {
  if (K == 0) {
    d_prev = plm_diff(h, S, n2, hpl, 0, nk);
    return S[n2] - 0.5 * d_prev;
  }
  if (K == nk) return S[n2 + hpl * (nk - 1)] + 0.5 * d_prev;
  const int km2 = (K - 2 > 0) ? K - 2 : 0, kp1 = (K + 1 < nk - 1) ? K + 1 : nk - 1;
  const double d = plm_diff(h, S, n2, hpl, K, nk);
  const double v = ppm_edge(h[n2 + hpl * km2], h[n2 + hpl * (K - 1)], h[n2 + hpl * K], h[n2 + hpl * kp1], S[n2 + hpl * (K - 1)], S[n2 + hpl * K],
                            d_prev, d, h_neglect);
  d_prev = d;
  return v;
}

// interpolate_for_nondim_position :1563
__device__ __forceinline__ double interp_nondim(double dRhoNeg, double Pneg, double dRhoPos, double Ppos) {
  if (Ppos <= Pneg) return 0.5;
  if (dRhoPos - dRhoNeg > 0.) return min2(1., max2(0., -dRhoNeg / (dRhoPos - dRhoNeg)));
  if (dRhoPos - dRhoNeg == 0) {
    if (dRhoNeg > 0.) return 0.;
    if (dRhoNeg < 0.) return 1.;
    return 0.5;
  }
  return 0.5;
}

// ppm_ave :1166
__device__ __forceinline__ double ppm_ave(double xL, double xR, double aL, double aR, double aMean, int &bad) {
  const double dx = xR - xL;
  const double xave = 0.5 * (xR + xL);
  const double a6o3 = 2. * aMean - (aL + aR);
  const double a6 = 3. * a6o3;
  if (dx < 0. || dx > 1.) { bad = 1; return 0.; }
  if (dx == 0.) return aL + (aR - aL) * xR + a6 * xR * (1. - xR);
  return (aL + xave * ((aR - aL) + a6)) - a6o3 * (xR * xR + xR * xL + xL * xL);
}

#define ND_BATCH 4
typedef unsigned char ko_t;      // a layer number 1 .. nk <= 128

// ppm_left_right_edge_values :2541 for one cell: the interface values above and below it and its mean
__device__ __forceinline__ void ppm_edges(double Ti0, double Ti1, double Tl, double &aL, double &aR) {
  aL = Ti0; aR = Ti1;
  if (signum(1., aR - Tl) * signum(1., Tl - aL) <= 0.0) { aL = Tl; aR = Tl; }
  else if (fsign(3., aR - aL) * ((Tl - aL) + (Tl - aR)) > fabs(aR - aL)) aL = Tl + 2.0 * (Tl - aR);
  else if (fsign(3., aR - aL) * ((Tl - aL) + (Tl - aR)) < -fabs(aR - aL)) aR = Tl + 2.0 * (Tl - aL);
}

struct NDArgs {
  m6::GridDev g;
  EosDev E;
  double ref_pres, gH;              // NDIFF_REF_PRES ; GV%g_Earth*GV%H_to_RZ
  double pa_to_H, h_neglect, scale; // scale: I_numitts
  int ns, symmetric;                // 2nk+2 ; ndiff_answer_date > 20240330
  const double *h, *T, *S, *p_surf;
  int interior;                     // NDIFF_INTERIOR_ONLY
  const double *hbl;                // the boundary-layer depth visc%h_ML with its halo (interior)
  int *kbot; double *zbot;          // boundary_k_range of every column: the layer and the fraction of it the boundary layer ends in
  double *Pint, *Tint, *Sint, *dRdT, *dRdS;   // [(nk+1)][h points]
  double *PoL[2], *PoR[2], *hEff[2], *Flx[2]; // [surface][faces of the direction]; Flx: + flx_stride per tracer of the batch
  ko_t *KoL[2], *KoR[2];
  const double *khdt[2];
  int nb;                           // tracers in this batch (<= ND_BATCH): the surfaces are read once for all of them
  double *t[ND_BATCH];              // the tracers being diffused
  double *Ti;                       // their interface values, [(nk+1)][h points], + col_stride per tracer
  long col_stride, flx_stride;
  double *stash;                    // [3][nk][h points], the symmetric form's N, S, E tendencies, + 3 nk hpl per tracer
  double cu[ND_BATCH];              // conc_underflow of the tracers
  int *bad;
};

__global__ __launch_bounds__(64) void nd_column_kernel(NDArgs A) {
  const m6::GridDev &g = A.g;
  const int i = g.isc - 1 + blockIdx.x * 64 + threadIdx.x, j = g.jsc - 1 + blockIdx.y, nk = g.nk;
  if (i > g.iec + 1) return;
  const long n2 = g.h2(i, j), hpl = (long)g.nih * g.njh;
  if (A.interior) {      // boundary_k_range(SURFACE, ...), src/tracer/MOM_hor_bnd_diffusion.F90:609-647, for wet columns (:381-386)
    int k_bot = 1;
    double zeta_bot = 0.;
    const double hbl = A.hbl[n2];
    if (g.mask2dT[n2] > 0.0 && hbl != 0.) {
      double hsum = 0., htot = 0.;
      for (int k = 0; k < nk; k++) hsum = hsum + A.h[n2 + hpl * k];
      if (hbl >= hsum) { k_bot = nk; zeta_bot = 1.; }
      else
        for (int k = 0; k < nk; k++) {
          const double hk = A.h[n2 + hpl * k];
          htot = htot + hk;
          if (htot >= hbl) { k_bot = k + 1; zeta_bot = 1 - (htot - hbl) / hk; break; }
        }
    }
    A.kbot[n2] = k_bot; A.zbot[n2] = zeta_bot;
  }
  double P = A.p_surf ? A.p_surf[n2] : 0., dT = 0., dS = 0.;
  for (int K = 0; K <= nk; K++) {
    if (K > 0) P = P + A.h[n2 + hpl * (K - 1)] * A.gH;
    const double Ti = interface_value(A.h, A.T, n2, hpl, K, nk, A.h_neglect, dT);
    const double Si = interface_value(A.h, A.S, n2, hpl, K, nk, A.h_neglect, dS);
    double rT, rS;
    eos_density_derivs(A.E, Ti, Si, (A.ref_pres >= 0.) ? A.ref_pres : P, rT, rS);
    A.Pint[n2 + hpl * K] = P; A.Tint[n2 + hpl * K] = Ti; A.Sint[n2 + hpl * K] = Si;
    A.dRdT[n2 + hpl * K] = rT; A.dRdS[n2 + hpl * K] = rS;
  }
}

// find_neutral_surface_positions_continuous :1353 (KoL / KoR keep the reference's 1-based layer numbers)
template <int DIR>
__global__ __launch_bounds__(64) void nd_surfaces_kernel(NDArgs A) {
  const m6::GridDev &g = A.g;
  const int i = (DIR ? g.isc : g.isc - 1) + blockIdx.x * 64 + threadIdx.x, j = (DIR ? g.jsc - 1 : g.jsc) + blockIdx.y, nk = g.nk;
  if (i > g.iec) return;
  const long f = DIR ? g.v2(i, j) : g.u2(i, j), pl = DIR ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh;
  const long hpl = (long)g.nih * g.njh, cl = g.h2(i, j), cr = DIR ? g.h2(i, j + 1) : g.h2(i + 1, j);
  double *__restrict__ PoL = A.PoL[DIR], *__restrict__ PoR = A.PoR[DIR], *__restrict__ hEff = A.hEff[DIR];
  ko_t *__restrict__ KoL = A.KoL[DIR], *__restrict__ KoR = A.KoR[DIR];
  const int ns = A.ns;
  if (!((DIR ? g.mask2dCv[f] : g.mask2dCu[f]) > 0.0)) {      // :472-481: what the arrays hold where no surfaces are searched for
    for (int ks = 0; ks < ns; ks++) {
      PoL[f + pl * ks] = 0.; PoR[f + pl * ks] = 0.; KoL[f + pl * ks] = 1; KoR[f + pl * ks] = 1;
      if (ks < ns - 1) hEff[f + pl * ks] = 0.;
    }
    return;
  }
  const double *__restrict__ Pint = A.Pint, *__restrict__ Tint = A.Tint, *__restrict__ Sint = A.Sint, *__restrict__ dRdT = A.dRdT,
                             *__restrict__ dRdS = A.dRdS;
#define LC(a, k) a[cl + hpl * ((k) - 1)]
#define RC(a, k) a[cr + hpl * ((k) - 1)]
  // The walk reads interface klm1 = max(kl-1, 1), klm1+1 and kl of the left column (and the same of the right): two interfaces a side
  // are kept in registers -- A at klm1, B at klm1+1; interface kl is A while kl = 1 and B afterwards -- and a step that moves kl or kr
  // down by one loads one new interface (the lanes of a wave stand at different depths, so every load here is a gather).
  struct Iface { double P, T, S, rT, rS; };
  auto loadL = [&](int k) { return Iface{LC(Pint, k), LC(Tint, k), LC(Sint, k), LC(dRdT, k), LC(dRdS, k)}; };
  auto loadR = [&](int k) { return Iface{RC(Pint, k), RC(Tint, k), RC(Sint, k), RC(dRdT, k), RC(dRdS, k)}; };
  Iface LA = loadL(1), LB = loadL(2), RA = loadR(1), RB = loadR(2);
  const int bl_kl = A.interior ? A.kbot[cl] : 0, bl_kr = A.interior ? A.kbot[cr] : 0;      // :1508-1521 (no layer number is <= 0)
  const double bl_zl = A.interior ? A.zbot[cl] : 0., bl_zr = A.interior ? A.zbot[cr] : 0.;
  int kr = 1, kl = 1, lastK_right = 1, lastK_left = 1;
  double lastP_right = 0., lastP_left = 0., absL_prev = 0., absR_prev = 0.;
  bool reached_bottom = false, searching_left = false, searching_right = false;
  for (int ks = 0; ks < ns; ks++) {
    int klm1 = (kl - 1 > 1) ? kl - 1 : 1;
    int krm1 = (kr - 1 > 1) ? kr - 1 : 1;
    const Iface Lk = (kl == 1) ? LA : LB, Rk = (kr == 1) ? RA : RB;
    const double Tr_kr = Rk.T, Sr_kr = Rk.S, aTr_kr = Rk.rT, aSr_kr = Rk.rS;
    const double Tl_kl = Lk.T, Sl_kl = Lk.S, aTl_kl = Lk.rT, aSl_kl = Lk.rS;
    const double dRho = 0.5 * ((aTr_kr + aTl_kl) * (Tr_kr - Tl_kl) + (aSr_kr + aSl_kl) * (Sr_kr - Sl_kl));
    if (!reached_bottom) {
      if (dRho < 0.) { searching_left = true; searching_right = false; }
      else if (dRho > 0.) { searching_right = true; searching_left = false; }
      else if (kl + kr == 2) { searching_left = true; searching_right = false; }
      else { searching_left = !searching_left; searching_right = !searching_right; }
    }
    double pL, pR;
    int oKL, oKR;
    if (searching_left) {
      const double dRhoTop = 0.5 * ((LA.rT + aTr_kr) * (LA.T - Tr_kr) + (LA.rS + aSr_kr) * (LA.S - Sr_kr));
      const double dRhoBot = 0.5 * ((LB.rT + aTr_kr) * (LB.T - Tr_kr) + (LB.rS + aSr_kr) * (LB.S - Sr_kr));
      if (dRhoTop > 0. || kr + kl == 2) pL = 0.;
      else if (dRhoTop >= dRhoBot) pL = 1.;
      else pL = interp_nondim(dRhoTop, LA.P, dRhoBot, LB.P);
      if (pL >= 1. && klm1 < nk) { klm1 = klm1 + 1; pL = pL - 1.; }
      if ((double)(klm1 - lastK_left) + (pL - lastP_left) < 0.) { pL = lastP_left; klm1 = lastK_left; }
      oKL = klm1;
      if (kr <= nk) { pR = 0.; oKR = kr; } else { pR = 1.; oKR = nk; }
      if (kr <= nk) {
        kr = kr + 1;
        if (kr > 2) { RA = RB; RB = loadR(kr); }
      } else { reached_bottom = true; searching_right = true; searching_left = false; }
    } else {
      const double dRhoTop = 0.5 * ((RA.rT + aTl_kl) * (RA.T - Tl_kl) + (RA.rS + aSl_kl) * (RA.S - Sl_kl));
      const double dRhoBot = 0.5 * ((RB.rT + aTl_kl) * (RB.T - Tl_kl) + (RB.rS + aSl_kl) * (RB.S - Sl_kl));
      if (dRhoTop >= 0. || kr + kl == 2) pR = 0.;
      else if (dRhoTop >= dRhoBot) pR = 1.;
      else pR = interp_nondim(dRhoTop, RA.P, dRhoBot, RB.P);
      if (pR >= 1. && krm1 < nk) { krm1 = krm1 + 1; pR = pR - 1.; }
      if ((double)(krm1 - lastK_right) + (pR - lastP_right) < 0.) { pR = lastP_right; krm1 = lastK_right; }
      oKR = krm1;
      if (kl <= nk) { pL = 0.; oKL = kl; } else { pL = 1.; oKL = nk; }
      if (kl <= nk) {
        kl = kl + 1;
        if (kl > 2) { LA = LB; LB = loadL(kl); }
      } else { reached_bottom = true; searching_right = false; searching_left = true; }
    }
    if (oKL <= bl_kl) { oKL = bl_kl; if (pL < bl_zl) pL = bl_zl; }
    if (oKR <= bl_kr) { oKR = bl_kr; if (pR < bl_zr) pR = bl_zr; }
    PoL[f + pl * ks] = pL; PoR[f + pl * ks] = pR; KoL[f + pl * ks] = (ko_t)oKL; KoR[f + pl * ks] = (ko_t)oKR;
    lastK_left = oKL; lastP_left = pL; lastK_right = oKR; lastP_right = pR;
    // absolute_position :2258 of this surface; that of the one above is the value formed a step ago from the same expression
    const double PL0 = LC(Pint, oKL), PR0 = RC(Pint, oKR);
    const double absL = PL0 + pL * (LC(Pint, oKL + 1) - PL0), absR = PR0 + pR * (RC(Pint, oKR + 1) - PR0);
    if (ks > 0) {
      const double hL = absL - absL_prev, hR = absR - absR_prev;
      double he = 0.;
      if (hL + hR > 0.) he = 2. * hL * hR / (hL + hR);
      hEff[f + pl * (ks - 1)] = he * A.pa_to_H;      // :570-575
    }
    absL_prev = absL; absR_prev = absR;
  }
#undef LC
#undef RC
}

__global__ __launch_bounds__(64) void nd_tracer_cols_kernel(NDArgs A) {
  const m6::GridDev &g = A.g;
  const int i = g.isc - 1 + blockIdx.x * 64 + threadIdx.x, j = g.jsc - 1 + blockIdx.y, nk = g.nk, z = blockIdx.z;
  if (i > g.iec + 1) return;
  const long n2 = g.h2(i, j), hpl = (long)g.nih * g.njh;
  const double *__restrict__ t = A.t[z];
  double *__restrict__ oTi = A.Ti + A.col_stride * z;
  double d = 0.;
  for (int K = 0; K <= nk; K++) oTi[n2 + hpl * K] = interface_value(A.h, t, n2, hpl, K, nk, A.h_neglect, d);
}

// neutral_surface_flux :2297 (continuous, no tapering: khtr_ave = 1)
template <int DIR>
__global__ __launch_bounds__(64) void nd_flux_kernel(NDArgs A) {
  const m6::GridDev &g = A.g;
  const int i = (DIR ? g.isc : g.isc - 1) + blockIdx.x * 64 + threadIdx.x, j = (DIR ? g.jsc - 1 : g.jsc) + blockIdx.y;
  if (i > g.iec) return;
  const long f = DIR ? g.v2(i, j) : g.u2(i, j), pl = DIR ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh;
  const long hpl = (long)g.nih * g.njh, cl = g.h2(i, j), cr = DIR ? g.h2(i, j + 1) : g.h2(i + 1, j);
  double *__restrict__ Flx = A.Flx[DIR];
  const int ns = A.ns, nb = A.nb;
  if (!((DIR ? g.mask2dCv[f] : g.mask2dCu[f]) > 0.0)) {
    for (int z = 0; z < nb; z++)
      for (int ks = 0; ks < ns - 1; ks++) Flx[A.flx_stride * z + f + pl * ks] = 0.;
    return;
  }
  const double *__restrict__ PiL = A.PoL[DIR], *__restrict__ PiR = A.PoR[DIR], *__restrict__ hEff = A.hEff[DIR];
  const ko_t *__restrict__ KoL = A.KoL[DIR], *__restrict__ KoR = A.KoR[DIR];
  int bad = 0;
  double pLt = PiL[f], pRt = PiR[f];
  int klt = KoL[f] - 1, krt = KoR[f] - 1;
  // per tracer: the value at the top surface (the reference's T_*_top of a layer is its T_*_bottom of the layer above: the same
  // expression of the same numbers) and the edge / mean values of the cells klt, krt, reloaded when the surface enters another cell
  double Ttop_l[ND_BATCH], Ttop_r[ND_BATCH], eL_l[ND_BATCH], eR_l[ND_BATCH], tm_l[ND_BATCH], eL_r[ND_BATCH], eR_r[ND_BATCH], tm_r[ND_BATCH];
#pragma unroll
  for (int z = 0; z < ND_BATCH; z++) {
    if (z < nb) {
      const double *__restrict__ Ti = A.Ti + A.col_stride * z;
      const double l0 = Ti[cl + hpl * klt], l1 = Ti[cl + hpl * (klt + 1)], r0 = Ti[cr + hpl * krt], r1 = Ti[cr + hpl * (krt + 1)];
      Ttop_l[z] = (1. - pLt) * l0 + pLt * l1;
      Ttop_r[z] = (1. - pRt) * r0 + pRt * r1;
      tm_l[z] = A.t[z][cl + hpl * klt]; tm_r[z] = A.t[z][cr + hpl * krt];
      ppm_edges(l0, l1, tm_l[z], eL_l[z], eR_l[z]);
      ppm_edges(r0, r1, tm_r[z], eL_r[z], eR_r[z]);
    }
  }
  for (int ks = 0; ks < ns - 1; ks++) {
    const double pLb = PiL[f + pl * (ks + 1)], pRb = PiR[f + pl * (ks + 1)];
    const int klb = KoL[f + pl * (ks + 1)] - 1, krb = KoR[f + pl * (ks + 1)] - 1;
    const double he = hEff[f + pl * ks];
#pragma unroll
    for (int z = 0; z < ND_BATCH; z++) {
      if (z < nb) {
        const double *__restrict__ Ti = A.Ti + A.col_stride * z;
        const double l0 = Ti[cl + hpl * klb], l1 = Ti[cl + hpl * (klb + 1)], r0 = Ti[cr + hpl * krb], r1 = Ti[cr + hpl * (krb + 1)];
        const double T_left_bottom = (1. - pLb) * l0 + pLb * l1;
        const double T_right_bottom = (1. - pRb) * r0 + pRb * r1;
        double flx = 0.;
        if (he != 0.) {
          const double T_left_top = Ttop_l[z], T_right_top = Ttop_r[z];
          const double T_left_layer = ppm_ave(pLt, pLb + (double)(klb - klt), eL_l[z], eR_l[z], tm_l[z], bad);
          const double T_right_layer = ppm_ave(pRt, pRb + (double)(krb - krt), eL_r[z], eR_r[z], tm_r[z], bad);
          const double dT_top = T_right_top - T_left_top;
          const double dT_bottom = T_right_bottom - T_left_bottom;
          double dT_ave = 0.5 * (dT_top + dT_bottom);
          const double dT_layer = T_right_layer - T_left_layer;
          if (signum(1., dT_top) * signum(1., dT_bottom) <= 0. || signum(1., dT_ave) * signum(1., dT_layer) <= 0.) dT_ave = 0.;
          else dT_ave = dT_layer;
          flx = dT_ave * he * 1.0;
        }
        Flx[A.flx_stride * z + f + pl * ks] = flx;
        Ttop_l[z] = T_left_bottom; Ttop_r[z] = T_right_bottom;
        // the surface enters another cell: its edge values (ppm_left_right_edge_values :2541) from the two interface values just read
        if (klb != klt) { tm_l[z] = A.t[z][cl + hpl * klb]; ppm_edges(l0, l1, tm_l[z], eL_l[z], eR_l[z]); }
        if (krb != krt) { tm_r[z] = A.t[z][cr + hpl * krb]; ppm_edges(r0, r1, tm_r[z], eL_r[z], eR_r[z]); }
      }
    }
    pLt = pLb; pRt = pRb; klt = klb; krt = krb;
  }
  if (bad) atomicOr(A.bad, 1);
}

// :921-960: the cell's tendency from the fluxes of its four faces, then the tracer
__global__ __launch_bounds__(64) void nd_update_kernel(NDArgs A) {
  extern __shared__ double acc[];      // [nk][64]
  const m6::GridDev &g = A.g;
  const int lane = threadIdx.x, i = g.isc + blockIdx.x * 64 + lane, j = g.jsc + blockIdx.y, nk = g.nk, z = blockIdx.z;
  if (i > g.iec) return;
  const long n2 = g.h2(i, j), hpl = (long)g.nih * g.njh;
  if (!(g.mask2dT[n2] > 0.)) return;
  double *__restrict__ t = A.t[z];
  const long upl = (long)(g.nih + 1) * g.njh, vpl = (long)g.nih * (g.njh + 1);
  const long uE = g.u2(i, j), uW = g.u2(i - 1, j), vN = g.v2(i, j), vS = g.v2(i, j - 1);
  const double cE = A.scale * A.khdt[0][uE], cW = A.scale * A.khdt[0][uW], cN = A.scale * A.khdt[1][vN], cS = A.scale * A.khdt[1][vS];
  const int ns = A.ns;
  const ko_t *__restrict__ uKoL = A.KoL[0], *__restrict__ uKoR = A.KoR[0], *__restrict__ vKoL = A.KoL[1], *__restrict__ vKoR = A.KoR[1];
  const double *__restrict__ uFlx = A.Flx[0] + A.flx_stride * z, *__restrict__ vFlx = A.Flx[1] + A.flx_stride * z;
#define ACC(k) acc[(k) * 64 + lane]
  // The surfaces are taken eight at a time: the loads of a group (layer numbers and fluxes of the four faces) are issued together, then
  // the group's contributions are added in the reference's order.  (With the LDS column a CU holds four waves; a global round trip per
  // surface was what the kernel waited for.)
  constexpr int NU = 8;
  const int nl = ns - 1;
  if (!A.symmetric) {      // :927-938
    for (int k = 0; k < nk; k++) ACC(k) = 0.;
    for (int ks0 = 0; ks0 < nl; ks0 += NU) {
      int kE[NU], kW[NU], kN[NU], kS[NU];
      double fE[NU], fW[NU], fN[NU], fS[NU];
#pragma unroll
      for (int q = 0; q < NU; q++) {
        const int ks = (ks0 + q < nl) ? ks0 + q : nl - 1;
        kE[q] = uKoL[uE + upl * ks] - 1; fE[q] = uFlx[uE + upl * ks];
        kW[q] = uKoR[uW + upl * ks] - 1; fW[q] = uFlx[uW + upl * ks];
        kN[q] = vKoL[vN + vpl * ks] - 1; fN[q] = vFlx[vN + vpl * ks];
        kS[q] = vKoR[vS + vpl * ks] - 1; fS[q] = vFlx[vS + vpl * ks];
      }
#pragma unroll
      for (int q = 0; q < NU; q++) {
        if (ks0 + q < nl) {
          ACC(kE[q]) = ACC(kE[q]) + cE * fE[q];
          ACC(kW[q]) = ACC(kW[q]) - cW * fW[q];
          ACC(kN[q]) = ACC(kN[q]) + cN * fN[q];
          ACC(kS[q]) = ACC(kS[q]) - cS * fS[q];
        }
      }
    }
  } else {                 // :939-954: one face at a time, the first three parked in the stash
    const long spl = hpl * nk;
    double *__restrict__ st = A.stash + 3 * spl * z;
    for (int face = 0; face < 4; face++) {      // N, S, E, W
      const ko_t *__restrict__ Kf = face == 0 ? vKoL + vN : (face == 1 ? vKoR + vS : (face == 2 ? uKoL + uE : uKoR + uW));
      const double *__restrict__ Ff = face == 0 ? vFlx + vN : (face == 1 ? vFlx + vS : (face == 2 ? uFlx + uE : uFlx + uW));
      const long fpl_ = face < 2 ? vpl : upl;
      const double cf = face == 0 ? cN : (face == 1 ? cS : (face == 2 ? cE : cW));
      const bool plus = (face == 0 || face == 2);
      for (int k = 0; k < nk; k++) ACC(k) = 0.;
      for (int ks0 = 0; ks0 < nl; ks0 += NU) {
        int kk[NU];
        double ff[NU];
#pragma unroll
        for (int q = 0; q < NU; q++) {
          const int ks = (ks0 + q < nl) ? ks0 + q : nl - 1;
          kk[q] = Kf[fpl_ * ks] - 1; ff[q] = Ff[fpl_ * ks];
        }
#pragma unroll
        for (int q = 0; q < NU; q++) {
          if (ks0 + q < nl) {
            if (plus) ACC(kk[q]) = ACC(kk[q]) + cf * ff[q];
            else ACC(kk[q]) = ACC(kk[q]) - cf * ff[q];
          }
        }
      }
      if (face < 3) for (int k = 0; k < nk; k++) st[spl * face + n2 + hpl * k] = ACC(k);
    }
    for (int k = 0; k < nk; k++)
      ACC(k) = (st[n2 + hpl * k] + st[spl + n2 + hpl * k]) + (st[2 * spl + n2 + hpl * k] + ACC(k));
  }
  const double IareaT = g.IareaT[n2];
  for (int k = 0; k < nk; k++) {
    double x = t[n2 + hpl * k] + ACC(k) * (IareaT / (A.h[n2 + hpl * k] + g.H_subroundoff));
    if (fabs(x) < A.cu[z]) x = 0.0;
    t[n2 + hpl * k] = x;
  }
#undef ACC
}

}  // namespace

namespace m6 {

// the neutral branch of tracer_hordiff (MOM_tracer_hor_diff.F90:474-534) on device arrays; khdt_x, khdt_y and the iteration count
// are those tracer_hordiff has formed
int neutral_branch(mom6hip_ctx_t *ctx, Stager &st, const mom6hip_neutral_diffusion_cs_t *nd, const mom6hip_eos_t *eos, const double *h,
                   const double *p_surf, const double *h_ML, const double *khdt_x, const double *khdt_y, int num_itts, double I_numitts,
                   const std::vector<double *> &d_tr, const std::vector<double> &cu, int idx_T, int idx_S, int *halo_updates) {
  static const char *names[8] = {"NDIFF_CONTINUOUS = False", "(free)", "NDIFF_TAPERING", "KHTR_USE_EBT_STRUCT",
                                 "NDIFF_USE_UNMASKED_TRANSPORT_BUG", "the neutral-diffusion diagnostics", "(free)", "(free)"};
  const int ntr = (int)d_tr.size();
  M6_REQUIRE(nd != nullptr && eos != nullptr, "tracer_hordiff: USE_NEUTRAL_DIFFUSION needs the neutral_diffusion control structure and tv%%eqn_of_state");
  M6_REQUIRE(nd->initialized, "neutral_diffusion: the control structure is not initialised");
  for (int q = 0; q < 8; q++) M6_REQUIRE(!nd->unsupported[q], "neutral_diffusion: %s is not provided by libmom6hip", names[q]);
  M6_REQUIRE(idx_T >= 0 && idx_T < ntr && idx_S >= 0 && idx_S < ntr, "tracer_hordiff: tv%%T and tv%%S must be among the tracers (idx_T, idx_S)");
  const m6::GridDev g = ctx->g;
  M6_REQUIRE(g.mask2dT && g.mask2dCu && g.mask2dCv && g.IareaT, "neutral_diffusion: mask2dT, mask2dCu, mask2dCv and IareaT are needed");
  M6_REQUIRE(g.nk >= 2 && g.nk <= 128, "neutral_diffusion: 2 to 128 layers are supported");
  hipStream_t s = ctx->stream;
  const int nk = g.nk, ns = 2 * nk + 2;
  const size_t hpl = (size_t)g.nih * g.njh, upl = (size_t)(g.nih + 1) * g.njh, vpl = (size_t)g.nih * (g.njh + 1);
  const size_t fpl = upl > vpl ? upl : vpl;
  NDArgs A;
  A.g = g; A.E = EosDev{eos->form, eos->Rho_T0_S0, eos->dRho_dT, eos->dRho_dS};
  A.ref_pres = nd->ref_pres; A.gH = g.g_Earth * nd->H_to_RZ; A.pa_to_H = 1. / (nd->H_to_RZ * g.g_Earth);
  A.h_neglect = g.H_subroundoff; A.scale = I_numitts; A.ns = ns; A.symmetric = nd->ndiff_answer_date > 20240330;
  A.h = h; A.p_surf = p_surf; A.khdt[0] = khdt_x; A.khdt[1] = khdt_y;
  double *cols = (double *)st.scratch(sizeof(double) * hpl * (nk + 1) * 5);
  const int nbmax = ntr < ND_BATCH ? ntr : ND_BATCH;
  double *faces = (double *)st.scratch(sizeof(double) * fpl * ns * (6 + 2 * (size_t)nbmax));
  ko_t *kos = (ko_t *)st.scratch(sizeof(ko_t) * fpl * ns * 4);
  double *tcols = (double *)st.scratch(sizeof(double) * hpl * ((size_t)nk + 1) * nbmax);
  double *stash = A.symmetric ? (double *)st.scratch(sizeof(double) * hpl * nk * 3 * nbmax) : nullptr;
  int *bad = (int *)st.scratch(64);
  A.interior = nd->interior_only != 0; A.hbl = nullptr; A.kbot = nullptr; A.zbot = nullptr;
  double *hbl = nullptr;
  if (A.interior) {      // :370-390: CS%hbl = visc%h_ML, pass_var(CS%hbl, halo=1)
    M6_REQUIRE(h_ML != nullptr, "hor_bnd_diffusion requires that visc%%h_ML is associated.");
    hbl = (double *)st.scratch(sizeof(double) * hpl);
    A.zbot = (double *)st.scratch(sizeof(double) * hpl); A.kbot = (int *)st.scratch(sizeof(int) * hpl);
    M6_REQUIRE(!st.failed() && hbl && A.zbot && A.kbot, "neutral_diffusion: out of device memory");
    A.hbl = hbl;
  }
  M6_REQUIRE(!st.failed() && cols && faces && kos && tcols && bad && (stash || !A.symmetric), "neutral_diffusion: out of device memory");
  A.Pint = cols; A.Tint = cols + hpl * (nk + 1); A.Sint = A.Tint + hpl * (nk + 1); A.dRdT = A.Sint + hpl * (nk + 1); A.dRdS = A.dRdT + hpl * (nk + 1);
  for (int d = 0; d < 2; d++) {
    A.PoL[d] = faces + fpl * ns * (3 * d); A.PoR[d] = A.PoL[d] + fpl * ns; A.hEff[d] = A.PoR[d] + fpl * ns;
    A.Flx[d] = faces + fpl * ns * (6 + d);      // the fluxes of tracer z of a batch: + 2 planes sets per tracer
    A.KoL[d] = kos + fpl * ns * (2 * d); A.KoR[d] = A.KoL[d] + fpl * ns;
  }
  A.flx_stride = (long)(fpl * ns * 2); A.col_stride = (long)(hpl * ((size_t)nk + 1));
  A.Ti = tcols; A.stash = stash; A.bad = bad;
  M6_HIP(hipMemsetAsync(bad, 0, sizeof(int), s));

  const int ni = g.iec - g.isc + 1, nj = g.jec - g.jsc + 1;
  std::vector<double *> pf(d_tr);
  std::vector<int32_t> ppos(ntr, MOM6HIP_POS_H), pnk(ntr, nk);
  auto calc_coeffs = [&]() -> int {      // neutral_diffusion_calc_coeffs :337
    A.T = d_tr[idx_T]; A.S = d_tr[idx_S];
    if (A.interior) {
      M6_HIP(hipMemcpyAsync(hbl, h_ML, sizeof(double) * hpl, hipMemcpyDeviceToDevice, s));
      double *f1[1] = {hbl}; int32_t p1[1] = {MOM6HIP_POS_H}, n1[1] = {1};
      if (int rc = m6::group_pass(ctx, f1, p1, n1, 1)) return rc;
    }
    hipLaunchKernelGGL(nd_column_kernel, dim3((ni + 2 + 63) / 64, nj + 2), dim3(64), 0, s, A);
    hipLaunchKernelGGL(nd_surfaces_kernel<0>, dim3((ni + 1 + 63) / 64, nj), dim3(64), 0, s, A);
    hipLaunchKernelGGL(nd_surfaces_kernel<1>, dim3((ni + 63) / 64, nj + 1), dim3(64), 0, s, A);
    return 0;
  };
  if (int rc = m6::group_pass(ctx, pf.data(), ppos.data(), pnk.data(), ntr)) return rc;      // do_group_pass(CS%pass_t) :478
  (*halo_updates)++;
  if (int rc = calc_coeffs()) return rc;
  for (int itt = 1; itt <= num_itts; itt++) {
    if (itt > 1) {
      if (int rc = m6::group_pass(ctx, pf.data(), ppos.data(), pnk.data(), ntr)) return rc;
      (*halo_updates)++;
      if (nd->recalc_neutral_surf) { if (int rc = calc_coeffs()) return rc; }
    }
    for (int m0 = 0; m0 < ntr; m0 += nbmax) {      // neutral_diffusion :605: the tracers are independent of each other, a batch at a time
      A.nb = (ntr - m0 < nbmax) ? ntr - m0 : nbmax;
      for (int z = 0; z < A.nb; z++) { A.t[z] = d_tr[m0 + z]; A.cu[z] = cu[m0 + z]; }
      hipLaunchKernelGGL(nd_tracer_cols_kernel, dim3((ni + 2 + 63) / 64, nj + 2, A.nb), dim3(64), 0, s, A);
      hipLaunchKernelGGL(nd_flux_kernel<0>, dim3((ni + 1 + 63) / 64, nj), dim3(64), 0, s, A);
      hipLaunchKernelGGL(nd_flux_kernel<1>, dim3((ni + 63) / 64, nj + 1), dim3(64), 0, s, A);
      hipLaunchKernelGGL(nd_update_kernel, dim3((ni + 63) / 64, nj, A.nb), dim3(64), sizeof(double) * 64 * nk, s, A);
    }
  }
  M6_HIP(hipGetLastError());
  int hbad = 0;
  M6_HIP(hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, s));
  M6_HIP(hipStreamSynchronize(s));
  M6_REQUIRE(!hbad, "ppm_ave: dx<0 or dx>1 should not happened! (a neutral layer spans more than one cell)");
  return 0;
}

}  // namespace m6
