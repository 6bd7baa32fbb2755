// neutral_diffusion.hip -- the continuous-reconstruction branch of MOM_neutral_diffusion (src/tracer/MOM_neutral_diffusion.F90)
// as gfx950 kernels, called from tracer_hordiff's neutral branch (MOM_tracer_hor_diff.F90:474-534; tracer_hor_diff.hip).
//
//   nd_column_kernel      neutral_diffusion_calc_coeffs :337-470 for one h column (halo 1): interface pressures, interface T and S by
//                         interface_scalar (:1078, PLM_diff + ppm_edge), their density derivatives -> one record of five numbers an interface
//   nd_surfaces_kernel    find_neutral_surface_positions_continuous (:1353) for one face: the merge walk down the 2nk+2 interfaces of
//                         the two columns; PoL, PoR, KoL, KoR, hEff (back in H units, :570-575) as [surface][face] planes
//   nd_tracer_cols_kernel interface values of a batch of up to 4 tracers (neutral_surface_flux :2373-2374) for one h column, with the means
//                         of the cells: one record of eight numbers an interface
//   nd_flux_kernel        neutral_surface_flux (:2297) for one face and the batch.  With the sums in the symmetric order
//                         (NDIFF_ANSWER_DATE > 20240330, :939-954: a face at a time) the kernel leaves the tendencies of the layers of the
//                         two cells of the face -- a surface never lies above the one before it, so a layer's sum is a run of consecutive
//                         surfaces and is kept in a register -- and nd_update_sym_kernel adds the four faces of a cell and updates the
//                         tracer (:955-959).  With the older order (:927-938: surface by surface round the four faces) it leaves the
//                         fluxes [surface][face] and nd_update_kernel accumulates them per layer in LDS.
// A lane owns a column or a face and walks it: the walks are serial in the surface index and differ from lane to lane.  What a walk
// reads at a depth of its own (the interfaces of a column) is stored as records, so that a gather is one or two sectors and not one a
// number, and kept in registers while the walk stays in the cell: a step down a column loads one record.  What all lanes read at the
// same step (the surfaces of a face) is stored as planes.  The blocks are dealt to the XCDs by their range of i (xcd_block).
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

// PLM_diff :1211 with c_method = 2, b_method = 1, for an interior layer from its own and its two neighbours' thicknesses and values
__device__ __forceinline__ double plm_diff(double hkm1, double hk, double hkp1, double Skm1, double Sk, double Skp1) {
  if (!((hkp1 + hk) * (hkm1 + hk) > 0.)) return 0.;
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

// interface_scalar :1078 with i_method = 2 for NZ fields of one column, walked down the interfaces K = 0 .. nk: out(K, v, m) gets the
// interface values v[z] and the means m[z] of the cell below the interface (0 at K = nk).  The walk keeps the four thicknesses and three
// values an interface reads in registers and loads one new layer a step (the reference's expressions on the same numbers).
template <int NZ, typename F>
__device__ __forceinline__ void walk_interfaces(const double *__restrict__ h, const double *const *S, int nz, long n2, long hpl, int nk,
                                                double h_neglect, F out) {
  const int k2 = (2 < nk - 1) ? 2 : nk - 1;
  double hB = h[n2], hC = h[n2 + hpl], hD = h[n2 + hpl * k2], hA = hB;
  double sB[NZ], sC[NZ], sD[NZ], dp[NZ], v[NZ];
#pragma unroll
  for (int z = 0; z < NZ; z++) {
    sB[z] = sC[z] = sD[z] = 0.; dp[z] = 0.;      // plm_diff of the top layer is 0
    if (z < nz) { sB[z] = S[z][n2]; sC[z] = S[z][n2 + hpl]; sD[z] = S[z][n2 + hpl * k2]; }
  }
#pragma unroll
  for (int z = 0; z < NZ; z++) v[z] = sB[z] - 0.5 * dp[z];
  out(0, v, sB);
  for (int K = 1; K < nk; K++) {      // hA, hB, hC, hD = h(max(K-2, 0)), h(K-1), h(K), h(min(K+1, nk-1)); sB, sC, sD = S(K-1), S(K), S(min(K+1, nk-1))
#pragma unroll
    for (int z = 0; z < NZ; z++) {
      const double d = (K < nk - 1) ? plm_diff(hB, hC, hD, sB[z], sC[z], sD[z]) : 0.;
      v[z] = ppm_edge(hA, hB, hC, hD, sB[z], sC[z], dp[z], d, h_neglect);
      dp[z] = d;
    }
    out(K, v, sC);
    const int kn = (K + 2 < nk - 1) ? K + 2 : nk - 1;
    hA = hB; hB = hC; hC = hD; hD = h[n2 + hpl * kn];
#pragma unroll
    for (int z = 0; z < NZ; z++) { sB[z] = sC[z]; sC[z] = sD[z]; if (z < nz) sD[z] = S[z][n2 + hpl * kn]; }
  }
  double zero[NZ];
#pragma unroll
  for (int z = 0; z < NZ; z++) { v[z] = sB[z] + 0.5 * dp[z]; zero[z] = 0.; }
  out(nk, v, zero);
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

constexpr int IREC = 5;      // an interface of a column: P, T, S, drho/dT, drho/dS
#ifndef ND_IREC_AOS
#define ND_IREC_AOS 0
#endif
// where field q of interface K (0-based) of the column c is: planes [field][K][h points] (the columns of a wave stand at nearly the same
// depth of their walks: a load of the wave is a few lines) or records [K][h points][field]
__device__ __forceinline__ long irec_at(long K, long c, int q, long hpl, int nk) {
  return ND_IREC_AOS ? (K * hpl + c) * IREC + q : ((long)q * (nk + 1) + K) * hpl + c;
}
constexpr int TREC = 2 * ND_BATCH;      // an interface of a column for a batch of tracers: their interface values, the means of the cell below

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
  double *irec;                     // [(nk+1)][h points][IREC]: the interfaces of the columns as records (a walk's gather reads one record)
  double *PoL[2], *PoR[2], *hEff[2]; // [surface][faces of the direction]
  double *Flx[2];                   // (the order of 2024 and before) [surface][faces], + flx_stride per tracer of the batch
  double *tend[2][2];               // (the symmetric order) [direction][side of the face: the cell it is the E | N face of, the W | S face of]
                                    // [layer][faces]: the tendency of the cell's layer from this face, + flx_stride per tracer
  ko_t *KoL[2], *KoR[2];
  const double *khdt[2];
  int nb;                           // tracers in this batch (<= ND_BATCH): the surfaces are read once for all of them
  double *t[ND_BATCH];              // the tracers being diffused
  double *trec;                     // [(nk+1)][h points][TREC]: interface values and cell means of the batch's tracers
  long flx_stride;
  double cu[ND_BATCH];              // conc_underflow of the tracers
  int *bad;
};

// Workgroups go round the eight XCDs by their index; a face reads the columns on its two sides, so the blocks of one range of i are given
// to one XCD, row after row: the column a row's faces read on their far side is in that XCD's L2 when the next row asks for it.
#ifndef ND_XCD
#define ND_XCD 0      // (measured: no gain on the benchmark grid, profiles/r04_experiments.txt section 9)
#endif
__device__ __forceinline__ bool xcd_block(int nbx, int nby, int &bx, int &by) {
  const int L = blockIdx.x, xcd = L & 7, q = L >> 3, per = (nbx + 7) >> 3;
  if (ND_XCD) { bx = xcd + 8 * (q % per); by = q / per; }
  else { const int nb8 = 8 * per; bx = L % nb8; by = L / nb8; }
  return bx < nbx && by < nby;
}
inline int xcd_grid(int nbx, int nby) { return 8 * ((nbx + 7) >> 3) * nby; }

__global__ __launch_bounds__(64) void nd_column_kernel(NDArgs A, int nbx, int nby) {
  const m6::GridDev &g = A.g;
  int bx, by;
  if (!xcd_block(nbx, nby, bx, by)) return;
  const int i = g.isc - 1 + bx * 64 + threadIdx.x, j = g.jsc - 1 + by, nk = g.nk;
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
  double P = A.p_surf ? A.p_surf[n2] : 0.;
  const double *TS[2] = {A.T, A.S};
  double h_above = 0.;
  walk_interfaces<2>(A.h, TS, 2, n2, hpl, nk, A.h_neglect, [&](int K, const double *v, const double *m) {
    (void)m;
    if (K > 0) P = P + h_above * A.gH;
    if (K < nk) h_above = A.h[n2 + hpl * K];
    double rT, rS;
    eos_density_derivs(A.E, v[0], v[1], (A.ref_pres >= 0.) ? A.ref_pres : P, rT, rS);
    double *r = A.irec;
    r[irec_at(K, n2, 0, hpl, nk)] = P; r[irec_at(K, n2, 1, hpl, nk)] = v[0]; r[irec_at(K, n2, 2, hpl, nk)] = v[1];
    r[irec_at(K, n2, 3, hpl, nk)] = rT; r[irec_at(K, n2, 4, hpl, nk)] = rS;
  });
}

// find_neutral_surface_positions_continuous :1353 (KoL / KoR keep the reference's 1-based layer numbers)
template <int DIR>
__global__ __launch_bounds__(64) void nd_surfaces_kernel(NDArgs A, int nbx, int nby) {
  const m6::GridDev &g = A.g;
  int bx, by;
  if (!xcd_block(nbx, nby, bx, by)) return;
  const int i = (DIR ? g.isc : g.isc - 1) + bx * 64 + threadIdx.x, j = (DIR ? g.jsc - 1 : g.jsc) + by, nk = g.nk;
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
  const double *__restrict__ irec = A.irec;
  // The walk reads interface klm1 = max(kl-1, 1), klm1+1 and kl of the left column (and the same of the right): two interfaces a side
  // are kept in registers -- A at klm1, B at klm1+1; interface kl is A while kl = 1 and B afterwards -- and a step that moves kl or kr
  // down by one loads one new interface: one record (the lanes of a wave stand at different depths, so every load here is a gather).
  struct Iface { double P, T, S, rT, rS; };
  auto load = [&](long c, int k) {
    return Iface{irec[irec_at(k - 1, c, 0, hpl, nk)], irec[irec_at(k - 1, c, 1, hpl, nk)], irec[irec_at(k - 1, c, 2, hpl, nk)],
                 irec[irec_at(k - 1, c, 3, hpl, nk)], irec[irec_at(k - 1, c, 4, hpl, nk)]};
  };
  auto loadL = [&](int k) { return load(cl, k); };
  auto loadR = [&](int k) { return load(cr, k); };
  auto PL = [&](int k) { return irec[irec_at(k - 1, cl, 0, hpl, nk)]; };
  auto PR = [&](int k) { return irec[irec_at(k - 1, cr, 0, hpl, nk)]; };
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
    const double PL0 = PL(oKL), PR0 = PR(oKR);
    const double absL = PL0 + pL * (PL(oKL + 1) - PL0), absR = PR0 + pR * (PR(oKR + 1) - PR0);
    if (ks > 0) {
      const double hL = absL - absL_prev, hR = absR - absR_prev;
      double he = 0.;
      if (hL + hR > 0.) he = 2. * hL * hR / (hL + hR);
      hEff[f + pl * (ks - 1)] = he * A.pa_to_H;      // :570-575
    }
    absL_prev = absL; absR_prev = absR;
  }
}

// the interface values of the batch's tracers (neutral_surface_flux :2373-2374) and the means of their cells, as records
__global__ __launch_bounds__(64) void nd_tracer_cols_kernel(NDArgs A, int nbx, int nby) {
  const m6::GridDev &g = A.g;
  int bx, by;
  if (!xcd_block(nbx, nby, bx, by)) return;
  const int i = g.isc - 1 + bx * 64 + threadIdx.x, j = g.jsc - 1 + by, nk = g.nk;
  if (i > g.iec + 1) return;
  const long n2 = g.h2(i, j), hpl = (long)g.nih * g.njh;
  const double *ts[ND_BATCH];
#pragma unroll
  for (int z = 0; z < ND_BATCH; z++) ts[z] = A.t[z < A.nb ? z : 0];
  walk_interfaces<ND_BATCH>(A.h, ts, A.nb, n2, hpl, nk, A.h_neglect, [&](int K, const double *v, const double *m) {
    double2 *r = (double2 *)(A.trec + ((long)K * hpl + n2) * TREC);
    r[0] = make_double2(v[0], v[1]); r[1] = make_double2(v[2], v[3]);
    r[2] = make_double2(m[0], m[1]); r[3] = make_double2(m[2], m[3]);
  });
}

// neutral_surface_flux :2297 (continuous, no tapering: khtr_ave = 1).  SYM: the tendencies of the two cells of the face from its fluxes,
// layer by layer (the order of the sums of neutral_diffusion :939-954: a face at a time, its surfaces top down); otherwise the fluxes
template <int DIR, bool SYM>
__global__ __launch_bounds__(64) void nd_flux_kernel(NDArgs A, int nbx, int nby) {
  const m6::GridDev &g = A.g;
  int bx, by;
  if (!xcd_block(nbx, nby, bx, by)) return;
  const int i = (DIR ? g.isc : g.isc - 1) + bx * 64 + threadIdx.x, j = (DIR ? g.jsc - 1 : g.jsc) + by;
  if (i > g.iec) return;
  const long f = DIR ? g.v2(i, j) : g.u2(i, j), pl = DIR ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh;
  const long hpl = (long)g.nih * g.njh, cl = g.h2(i, j), cr = DIR ? g.h2(i, j + 1) : g.h2(i + 1, j);
  double *__restrict__ Flx = A.Flx[DIR];
  double *__restrict__ tdL = A.tend[DIR][0], *__restrict__ tdR = A.tend[DIR][1];
  const int ns = A.ns, nb = A.nb, nk = g.nk;
  if (!((DIR ? g.mask2dCv[f] : g.mask2dCu[f]) > 0.0)) {
    for (int z = 0; z < nb; z++) {
      if (SYM) for (int k = 0; k < nk; k++) { tdL[A.flx_stride * z + f + pl * k] = 0.; tdR[A.flx_stride * z + f + pl * k] = 0.; }
      else for (int ks = 0; ks < ns - 1; ks++) Flx[A.flx_stride * z + f + pl * ks] = 0.;
    }
    return;
  }
  const double *__restrict__ PiL = A.PoL[DIR], *__restrict__ PiR = A.PoR[DIR], *__restrict__ hEff = A.hEff[DIR];
  const ko_t *__restrict__ KoL = A.KoL[DIR], *__restrict__ KoR = A.KoR[DIR];
  const double cf = A.scale * A.khdt[DIR][f];
  int bad = 0;
  double pLt = PiL[f], pRt = PiR[f];
  int klt = KoL[f] - 1, krt = KoR[f] - 1;
  // A side of the face: the cell the surface is in, with the records of the interfaces above and below it (the interface values i0, i1 of
  // the batch's tracers, the means m0 of the cell and m1 of the cell below), the limited edge values of the cell
  // (ppm_left_right_edge_values :2541) and the value at the surface above (the reference's T_*_top of a sublayer is its T_*_bottom of the
  // sublayer above: the same expression of the same numbers).  A surface that enters the next cell needs one new record.
  struct Side { double i0[ND_BATCH], i1[ND_BATCH], m0[ND_BATCH], m1[ND_BATCH], eL[ND_BATCH], eR[ND_BATCH], top[ND_BATCH]; };
  auto load = [&](long c, int K, double *iv, double *mv) {
    const double2 *r = (const double2 *)(A.trec + ((long)K * hpl + c) * TREC);
    const double2 a = r[0], b = r[1], c2 = r[2], d = r[3];
    iv[0] = a.x; iv[1] = a.y; iv[2] = b.x; iv[3] = b.y; mv[0] = c2.x; mv[1] = c2.y; mv[2] = d.x; mv[3] = d.y;
  };
  auto enter = [&](Side &s, long c, int k_old, int k_new) {      // the records of the cell k_new
    if (k_new == k_old + 1) {
#pragma unroll
      for (int z = 0; z < ND_BATCH; z++) { s.i0[z] = s.i1[z]; s.m0[z] = s.m1[z]; }
    } else {
      load(c, k_new, s.i0, s.m0);
    }
    load(c, k_new + 1, s.i1, s.m1);
  };
  Side sl, sr;
  load(cl, klt, sl.i0, sl.m0); load(cl, klt + 1, sl.i1, sl.m1);
  load(cr, krt, sr.i0, sr.m0); load(cr, krt + 1, sr.i1, sr.m1);
#pragma unroll
  for (int z = 0; z < ND_BATCH; z++) {
    sl.top[z] = (1. - pLt) * sl.i0[z] + pLt * sl.i1[z];
    sr.top[z] = (1. - pRt) * sr.i0[z] + pRt * sr.i1[z];
    ppm_edges(sl.i0[z], sl.i1[z], sl.m0[z], sl.eL[z], sl.eR[z]);
    ppm_edges(sr.i0[z], sr.i1[z], sr.m0[z], sr.eL[z], sr.eR[z]);
  }
  // SYM: the running sums of the layers klt, krt; every layer of the two columns is stored once (zero where no sublayer lies in it)
  double accL[ND_BATCH], accR[ND_BATCH];
#pragma unroll
  for (int z = 0; z < ND_BATCH; z++) { accL[z] = 0.; accR[z] = 0.; }
  if (SYM) {
    for (int z = 0; z < nb; z++) {
      for (int k = 0; k < klt; k++) tdL[A.flx_stride * z + f + pl * k] = 0.;
      for (int k = 0; k < krt; k++) tdR[A.flx_stride * z + f + pl * k] = 0.;
    }
  }
  for (int ks = 0; ks < ns - 1; ks++) {
    const double pLb = PiL[f + pl * (ks + 1)], pRb = PiR[f + pl * (ks + 1)];
    const int klb = KoL[f + pl * (ks + 1)] - 1, krb = KoR[f + pl * (ks + 1)] - 1;
    const double he = hEff[f + pl * ks];
    if (klb < klt || krb < krt) bad |= 2;      // (the positions of the surfaces never move up a column)
    double lay_l[ND_BATCH], lay_r[ND_BATCH];   // the means of the sublayer in the cells klt, krt
    if (he != 0.) {
#pragma unroll
      for (int z = 0; z < ND_BATCH; z++) {
        lay_l[z] = ppm_ave(pLt, pLb + (double)(klb - klt), sl.eL[z], sl.eR[z], sl.m0[z], bad);
        lay_r[z] = ppm_ave(pRt, pRb + (double)(krb - krt), sr.eL[z], sr.eR[z], sr.m0[z], bad);
      }
    }
    if (klb != klt) enter(sl, cl, klt, klb);
    if (krb != krt) enter(sr, cr, krt, krb);
#pragma unroll
    for (int z = 0; z < ND_BATCH; z++) {
      const double T_left_bottom = (1. - pLb) * sl.i0[z] + pLb * sl.i1[z];
      const double T_right_bottom = (1. - pRb) * sr.i0[z] + pRb * sr.i1[z];
      double flx = 0.;
      if (he != 0.) {
        const double T_left_top = sl.top[z], T_right_top = sr.top[z];
        const double T_left_layer = lay_l[z], T_right_layer = lay_r[z];
        const double dT_top = T_right_top - T_left_top;
        const double dT_bottom = T_right_bottom - T_left_bottom;
        double dT_ave = 0.5 * (dT_top + dT_bottom);
        const double dT_layer = T_right_layer - T_left_layer;
        if (signum(1., dT_top) * signum(1., dT_bottom) <= 0. || signum(1., dT_ave) * signum(1., dT_layer) <= 0.) dT_ave = 0.;
        else dT_ave = dT_layer;
        flx = dT_ave * he * 1.0;
      }
      if (SYM) { accL[z] = accL[z] + cf * flx; accR[z] = accR[z] - cf * flx; }
      else if (z < nb) Flx[A.flx_stride * z + f + pl * ks] = flx;
      sl.top[z] = T_left_bottom; sr.top[z] = T_right_bottom;
    }
    // the surface enters another cell: its edge values from the interface values and the mean just read; SYM: the layer it leaves is done
    if (klb != klt) {
#pragma unroll
      for (int z = 0; z < ND_BATCH; z++) {
        ppm_edges(sl.i0[z], sl.i1[z], sl.m0[z], sl.eL[z], sl.eR[z]);
        if (SYM) {
          if (z < nb) {
            tdL[A.flx_stride * z + f + pl * klt] = accL[z];
            for (int k = klt + 1; k < klb; k++) tdL[A.flx_stride * z + f + pl * k] = 0.;
          }
          accL[z] = 0.;
        }
      }
    }
    if (krb != krt) {
#pragma unroll
      for (int z = 0; z < ND_BATCH; z++) {
        ppm_edges(sr.i0[z], sr.i1[z], sr.m0[z], sr.eL[z], sr.eR[z]);
        if (SYM) {
          if (z < nb) {
            tdR[A.flx_stride * z + f + pl * krt] = accR[z];
            for (int k = krt + 1; k < krb; k++) tdR[A.flx_stride * z + f + pl * k] = 0.;
          }
          accR[z] = 0.;
        }
      }
    }
    pLt = pLb; pRt = pRb; klt = klb; krt = krb;
  }
  if (SYM) {
    for (int z = 0; z < nb; z++) {
      tdL[A.flx_stride * z + f + pl * klt] = accL[z];
      for (int k = klt + 1; k < nk; k++) tdL[A.flx_stride * z + f + pl * k] = 0.;
      tdR[A.flx_stride * z + f + pl * krt] = accR[z];
      for (int k = krt + 1; k < nk; k++) tdR[A.flx_stride * z + f + pl * k] = 0.;
    }
  }
  if (bad) atomicOr(A.bad, bad);
}

// :955-959 after the sums of :939-954 (the symmetric order): the four faces' tendencies of the layer, then the tracer
__global__ __launch_bounds__(256) void nd_update_sym_kernel(NDArgs A) {
  const m6::GridDev &g = A.g;
  const int i = g.isc + blockIdx.x * 256 + threadIdx.x, j = g.jsc + blockIdx.y, k = blockIdx.z % g.nk, z = blockIdx.z / g.nk;
  if (i > g.iec) return;
  const long n2 = g.h2(i, j), hpl = (long)g.nih * g.njh;
  if (!(g.mask2dT[n2] > 0.)) return;
  const long upl = (long)(g.nih + 1) * g.njh, vpl = (long)g.nih * (g.njh + 1);
  const double tN = A.tend[1][0][A.flx_stride * z + g.v2(i, j) + vpl * k], tS = A.tend[1][1][A.flx_stride * z + g.v2(i, j - 1) + vpl * k];
  const double tE = A.tend[0][0][A.flx_stride * z + g.u2(i, j) + upl * k], tW = A.tend[0][1][A.flx_stride * z + g.u2(i - 1, j) + upl * k];
  const double tendency = (tN + tS) + (tE + tW);
  double *__restrict__ t = A.t[z];
  double x = t[n2 + hpl * k] + tendency * (g.IareaT[n2] / (A.h[n2 + hpl * k] + g.H_subroundoff));
  if (fabs(x) < A.cu[z]) x = 0.0;
  t[n2 + hpl * k] = x;
}

// :921-938, :955-959 (the order of 2024 and before): the cell's tendency from the fluxes of its four faces, surface by surface round the
// faces E, W, N, S, accumulated per layer in LDS (the layer index is data: KoL / KoR), then the tracer.  (A thread per layer that walks
// the runs of its four faces -- no LDS, full occupancy -- reads the fluxes as gathers and is slower: profiles/r04_experiments.txt.)
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
  // the group's contributions are added in the reference's order.
  constexpr int NU = 8;
  const int nl = ns - 1;
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
  double *irec = (double *)st.scratch(sizeof(double) * hpl * (nk + 1) * IREC);
  const int nbmax = ntr < ND_BATCH ? ntr : ND_BATCH;
  // per direction PoL, PoR, hEff [surface][faces]; then per tracer of a batch either the fluxes [surface][faces] of the two directions or
  // the tendencies [layer][faces] of the two sides of the faces of the two directions
  const size_t per_tracer = A.symmetric ? fpl * nk * 4 : fpl * ns * 2;
  double *faces = (double *)st.scratch(sizeof(double) * (fpl * ns * 6 + per_tracer * (size_t)nbmax));
  ko_t *kos = (ko_t *)st.scratch(sizeof(ko_t) * fpl * ns * 4);
  double *trec = (double *)st.scratch(sizeof(double) * hpl * ((size_t)nk + 1) * TREC);
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
  M6_REQUIRE(!st.failed() && irec && faces && kos && trec && bad, "neutral_diffusion: out of device memory");
  A.irec = irec; A.trec = trec; A.bad = bad;
  double *per = faces + fpl * ns * 6;
  for (int d = 0; d < 2; d++) {
    A.PoL[d] = faces + fpl * ns * (3 * d); A.PoR[d] = A.PoL[d] + fpl * ns; A.hEff[d] = A.PoR[d] + fpl * ns;
    A.KoL[d] = kos + fpl * ns * (2 * d); A.KoR[d] = A.KoL[d] + fpl * ns;
    A.Flx[d] = A.symmetric ? nullptr : per + fpl * ns * d;
    A.tend[d][0] = A.symmetric ? per + fpl * nk * (2 * d) : nullptr; A.tend[d][1] = A.symmetric ? per + fpl * nk * (2 * d + 1) : nullptr;
  }
  A.flx_stride = (long)per_tracer;
  M6_HIP(hipMemsetAsync(bad, 0, sizeof(int), s));

  const int ni = g.iec - g.isc + 1, nj = g.jec - g.jsc + 1;
  const int nbc = (ni + 2 + 63) / 64, nbu = (ni + 1 + 63) / 64, nbv = (ni + 63) / 64;      // blocks along i: columns with the halo, u faces, v faces
  std::vector<double *> pf(d_tr);
  std::vector<int32_t> ppos(ntr, MOM6HIP_POS_H), pnk(ntr, nk);
  auto calc_coeffs = [&]() -> int {      // neutral_diffusion_calc_coeffs :337
    A.T = d_tr[idx_T]; A.S = d_tr[idx_S];
    if (A.interior) {
      M6_HIP(hipMemcpyAsync(hbl, h_ML, sizeof(double) * hpl, hipMemcpyDeviceToDevice, s));
      double *f1[1] = {hbl}; int32_t p1[1] = {MOM6HIP_POS_H}, n1[1] = {1};
      if (int rc = m6::group_pass(ctx, f1, p1, n1, 1)) return rc;
    }
    hipLaunchKernelGGL(nd_column_kernel, dim3(xcd_grid(nbc, nj + 2)), dim3(64), 0, s, A, nbc, nj + 2);
    hipLaunchKernelGGL(nd_surfaces_kernel<0>, dim3(xcd_grid(nbu, nj)), dim3(64), 0, s, A, nbu, nj);
    hipLaunchKernelGGL(nd_surfaces_kernel<1>, dim3(xcd_grid(nbv, nj + 1)), dim3(64), 0, s, A, nbv, nj + 1);
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
      for (int z = A.nb; z < ND_BATCH; z++) { A.t[z] = nullptr; A.cu[z] = 0.; }
      hipLaunchKernelGGL(nd_tracer_cols_kernel, dim3(xcd_grid(nbc, nj + 2)), dim3(64), 0, s, A, nbc, nj + 2);
      if (A.symmetric) {
        hipLaunchKernelGGL((nd_flux_kernel<0, true>), dim3(xcd_grid(nbu, nj)), dim3(64), 0, s, A, nbu, nj);
        hipLaunchKernelGGL((nd_flux_kernel<1, true>), dim3(xcd_grid(nbv, nj + 1)), dim3(64), 0, s, A, nbv, nj + 1);
        hipLaunchKernelGGL(nd_update_sym_kernel, dim3((ni + 255) / 256, nj, nk * A.nb), dim3(256), 0, s, A);
      } else {
        hipLaunchKernelGGL((nd_flux_kernel<0, false>), dim3(xcd_grid(nbu, nj)), dim3(64), 0, s, A, nbu, nj);
        hipLaunchKernelGGL((nd_flux_kernel<1, false>), dim3(xcd_grid(nbv, nj + 1)), dim3(64), 0, s, A, nbv, nj + 1);
        hipLaunchKernelGGL(nd_update_kernel, dim3((ni + 63) / 64, nj, A.nb), dim3(64), sizeof(double) * 64 * nk, s, A);
      }
    }
  }
  M6_HIP(hipGetLastError());
  int hbad = 0;
  M6_HIP(hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, s));
  M6_HIP(hipStreamSynchronize(s));
  M6_REQUIRE(!(hbad & 1), "ppm_ave: dx<0 or dx>1 should not happened! (a neutral layer spans more than one cell)");
  M6_REQUIRE(!(hbad & 2), "neutral_diffusion: a neutral surface lies above the one before it in a column");
  return 0;
}

}  // namespace m6
