// epipycnal_diff.hip -- tracer_epipycnal_ML_diff (src/tracer/MOM_tracer_hor_diff.F90:700-1621; DIFFUSE_ML_TO_INTERIOR, .testing/tc1):
// diffusion of every tracer along the coordinate density between the variable-density (mixed and buffer) layers and the interior of
// a layered run, as gfx950 kernels.
//
//   epi_column_kernel   thread per column, halo 2: the coordinate density of the nkmb variable-density layers at tv%P_Ref, its
//                       maximum, and the target-density layer it falls in by bisection (:835-869)
//   epi_sort_kernel     thread per column, halo 1: k_end_srt and its maximum over the tile (PEmax_kRho, an atomic max the later
//                       kernels read from memory -- the host never waits for it), the list of the layers thicker than h_exclude
//                       with their densities, sorted by straight insertion as the reference does (:871-912)
//   epi_pair_kernel     thread per face: the merge of the two sorted lists into pairings, the thickness demanded from every layer
//                       and the share that can be supplied, the thicknesses of both sides of every pairing (:930-1251)
//   epi_flux_kernel     thread per face and tracer: the range of concentrations around the face, and for every pairing the flux
//                       and the two vertical adjustments that keep the split sides within that range (:1265-1412, :1420-1538)
//   epi_update_kernel   thread per cell and tracer: the fluxes of the cell's four faces are summed INTO the cell in the order the
//                       reference's face loops visit them (west face, east face, south face, north face; the pairings of a face in
//                       their order) -- the reference scatters from the faces, which on a GPU would need atomics and lose the order --
//                       then the update, and the concentration underflow (:1541-1609)
// The per-column and per-face lists live in scratch as planes over the horizontal index (position in the list slowest), so that the
// lanes of a wave, which are neighbouring columns or faces at mostly the same position, read and write neighbouring addresses.
// This is a small-configuration path (layered runs): nothing here is tuned beyond that layout.
#include <cfloat>
#include <cmath>

#include "common.hpp"
#include "eos.hpp"

namespace {

using m6::max2;
using m6::min2;
using m6::eos::EosDev;
using m6::eos::eos_density;

struct EpiArgs {
  m6::GridDev g;
  EosDev E;
  double P_Ref, h_exclude, I_maxitt;
  int nkmb, old_answers, limit_bug;
  const double *Rlay;      // device, nk values: Rlay[k-1] = GV%Rlay(k)
  const double *h, *T, *S;
  const double *khdt[2];
  // per column (planes of hpl)
  double *rho_coord;       // nkmb planes
  int *max_kRho, *num_srt, *PEmax;      // PEmax: one int
  double *rho_srt, *h_srt; // nk planes
  int *k0_srt;             // nk planes
  // per face, [dir]: planes of upl / vpl; np2 = 2 nk pairings at most
  int *nP[2];
  int *kk[2];              // 4 np2 planes: k0b_L, k0a_L, k0b_R, k0a_R
  double *ww[2];           // 4 np2 planes: deep_wt_L, deep_wt_R, hP_L, hP_R
  double *work[2];         // 4 nk planes: h_demand_L / h_supply_frac_L, h_demand_R / ..., h_used_L, h_used_R
  int *kbs[2];             // 3 np2 planes: kbs_Lp, kbs_Rp, (left_set | right_set << 1)
  double *flx[2];          // 3 np2 planes: Tr_flux, Tr_adj_vert_L, Tr_adj_vert_R of the tracer in hand
  double *acc;             // per cell: nk planes (tr_flux_conv), or 4 nk planes (E, W, N, S)
  double *tr;              // the tracer in hand
  double cu;               // its conc_underflow
};

// thread (i, j) over (is-2 : ie+2, js-2 : je+2)
__global__ __launch_bounds__(256) void epi_column_kernel(EpiArgs A) {
  const m6::GridDev &g = A.g;
  const int i = g.isc - 2 + blockIdx.x * blockDim.x + threadIdx.x, j = g.jsc - 2 + blockIdx.y;
  if (i > g.iec + 2) return;
  const long hpl = (long)g.nih * g.njh, c = g.h2(i, j);
  const int nz = g.nk, nkmb = A.nkmb;
  double Rml_max = 0.0;
  for (int k = 1; k <= nkmb; k++) {
    const double r = eos_density(A.E, A.T[hpl * (k - 1) + c], A.S[hpl * (k - 1) + c], A.P_Ref);
    A.rho_coord[hpl * (k - 1) + c] = r;
    if (k == 1 || Rml_max < r) Rml_max = r;
  }
  int mk = 0;
  if (g.mask2dT[c] > 0.0) {      // GV%Rlay(max_kRho-1) < Rml_max <= GV%Rlay(max_kRho) :854-869
    const double *Rlay = A.Rlay - 1;      // 1-based
    if ((nkmb + 1 > nz) || (Rml_max > Rlay[nz])) mk = nz + 1;
    else if ((nkmb + 2 > nz) || (Rml_max <= Rlay[nkmb + 1])) mk = nkmb + 1;
    else {
      int k_min = nkmb + 2, k_max = nz;
      for (;;) {
        const int k_test = (k_min + k_max) / 2;
        if (Rml_max <= Rlay[k_test - 1]) k_max = k_test - 1;
        else if (Rlay[k_test] < Rml_max) k_min = k_test + 1;
        else { mk = k_test; break; }
        if (k_min == k_max) { mk = k_max; break; }
      }
    }
  }
  A.max_kRho[c] = mk;
  A.num_srt[c] = 0;
}

// thread (i, j) over (is-1 : ie+1, js-1 : je+1)
__global__ __launch_bounds__(256) void epi_sort_kernel(EpiArgs A) {
  const m6::GridDev &g = A.g;
  const int i = g.isc - 1 + blockIdx.x * blockDim.x + threadIdx.x, j = g.jsc - 1 + blockIdx.y;
  if (i > g.iec + 1) return;
  const long hpl = (long)g.nih * g.njh, c = g.h2(i, j);
  const int nz = g.nk, nkmb = A.nkmb;
  int k_end = A.max_kRho[c];
  k_end = max(k_end, A.max_kRho[g.h2(i - 1, j)]); k_end = max(k_end, A.max_kRho[g.h2(i + 1, j)]);
  k_end = max(k_end, A.max_kRho[g.h2(i, j - 1)]); k_end = max(k_end, A.max_kRho[g.h2(i, j + 1)]);
  atomicMax(A.PEmax, k_end < nz ? k_end : nz);      // PEmax_kRho :871-877
  int ns = 0;
  if (g.mask2dT[c] > 0.0) {
    for (int k = 1; k <= nkmb; k++) {
      const double hk = A.h[hpl * (k - 1) + c];
      if (hk > A.h_exclude) {
        A.k0_srt[hpl * ns + c] = k; A.rho_srt[hpl * ns + c] = A.rho_coord[hpl * (k - 1) + c]; A.h_srt[hpl * ns + c] = hk;
        ns++;
      }
    }
    const int kl = k_end < nz ? k_end : nz;      // k <= PEmax_kRho and k <= k_end_srt(i,j)
    for (int k = nkmb + 1; k <= kl; k++) {
      const double hk = A.h[hpl * (k - 1) + c];
      if (hk > A.h_exclude) {
        A.k0_srt[hpl * ns + c] = k; A.rho_srt[hpl * ns + c] = A.Rlay[k - 1]; A.h_srt[hpl * ns + c] = hk;
        ns++;
      }
    }
    // straight insertion :902-912 (positions 1-based in the comments of the reference: element k is plane k-1)
    for (int k = 2; k <= ns; k++) {
      if (A.rho_srt[hpl * (k - 1) + c] < A.rho_srt[hpl * (k - 2) + c]) {
        for (int k2 = k; k2 >= 2; k2--) {
          const long a = hpl * (k2 - 2) + c, b = hpl * (k2 - 1) + c;
          if (A.rho_srt[b] >= A.rho_srt[a]) break;
          const int it = A.k0_srt[a]; A.k0_srt[a] = A.k0_srt[b]; A.k0_srt[b] = it;
          double t = A.rho_srt[a]; A.rho_srt[a] = A.rho_srt[b]; A.rho_srt[b] = t;
          t = A.h_srt[a]; A.h_srt[a] = A.h_srt[b]; A.h_srt[b] = t;
        }
      }
    }
  }
  A.num_srt[c] = ns;
}

// thread per face: DIR 0 (I, j) over (is-1 : ie, js : je); DIR 1 (i, J) over (is : ie, js-1 : je)
template <int DIR>
__global__ __launch_bounds__(256) void epi_pair_kernel(EpiArgs A) {
  const m6::GridDev &g = A.g;
  const int i = (DIR ? g.isc : g.isc - 1) + blockIdx.x * blockDim.x + threadIdx.x, j = (DIR ? g.jsc - 1 : g.jsc) + blockIdx.y;
  if (i > g.iec) return;
  const long hpl = (long)g.nih * g.njh, fpl = DIR ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh;
  const long f = DIR ? g.v2(i, j) : g.u2(i, j), cL = g.h2(i, j), cR = DIR ? g.h2(i, j + 1) : g.h2(i + 1, j);
  const int np2 = 2 * g.nk, nkmb = A.nkmb;
  if (!((DIR ? g.mask2dCv[f] : g.mask2dCu[f]) > 0.0)) { A.nP[DIR][f] = 0; return; }
  const int nL = A.num_srt[cL], nR = A.num_srt[cR];
  // list element k (1-based) of the left and right columns
#define RL(k) A.rho_srt[hpl * ((k) - 1) + cL]
#define RR(k) A.rho_srt[hpl * ((k) - 1) + cR]
#define HL(k) A.h_srt[hpl * ((k) - 1) + cL]
#define HR(k) A.h_srt[hpl * ((k) - 1) + cR]
#define K0L(k) A.k0_srt[hpl * ((k) - 1) + cL]
#define K0R(k) A.k0_srt[hpl * ((k) - 1) + cR]
  // the face's lists: pairing k (0-based)
  int *k0b_L = A.kk[DIR] + f, *k0a_L = k0b_L + fpl * np2, *k0b_R = k0a_L + fpl * np2, *k0a_R = k0b_R + fpl * np2;
  double *deep_wt_L = A.ww[DIR] + f, *deep_wt_R = deep_wt_L + fpl * np2, *hP_L = deep_wt_R + fpl * np2, *hP_R = hP_L + fpl * np2;
  double *h_demand_L = A.work[DIR] + f - fpl, *h_demand_R = h_demand_L + fpl * g.nk, *h_used_L = h_demand_R + fpl * g.nk,
         *h_used_R = h_used_L + fpl * g.nk;      // 1-based: [fpl * k]
  int *kbs_Lp = A.kbs[DIR] + f, *kbs_Rp = kbs_Lp + fpl * np2, *sets = kbs_Rp + fpl * np2;
#define P(a, k) a[fpl * (k)]
  for (int k = 1; k <= nL; k++) { P(h_demand_L, k) = 0.0; P(h_used_L, k) = 0.0; }
  for (int k = 1; k <= nR; k++) { P(h_demand_R, k) = 0.0; P(h_used_R, k) = 0.0; }
  int kL, kR, nP = 0;
  const double rL1 = nL > 0 ? RL(1) : 0.0, rR1 = nR > 0 ? RR(1) : 0.0;
  if (rL1 < rR1) {      // :938-946
    kR = 1;
    for (kL = 2; kL <= nL; kL++) if (RL(kL) >= rR1) break;
  } else if (rR1 < rL1) {
    kL = 1;
    for (kR = 2; kR <= nR; kR++) if (RR(kR) >= rL1) break;
  } else { kL = 1; kR = 1; }
  for (;;) {      // :948-1010
    if ((kL > nL) || (kR > nR)) break;
    const double rl = RL(kL), rr = RR(kR);
    if (rl > rr) {
      const int k = nP++;
      const double rho_pair = rr;
      P(k0b_L, k) = K0L(kL); P(k0b_R, k) = K0R(kR);
      P(k0a_L, k) = K0L(kL - 1); P(k0a_R, k) = K0R(kR);
      P(kbs_Lp, k) = kL; P(kbs_Rp, k) = kR;
      const double rho_a = RL(kL - 1), rho_b = rl;
      double wt_b = 1.0; if (fabs(rho_a - rho_b) > fabs(rho_pair - rho_a)) wt_b = (rho_pair - rho_a) / (rho_b - rho_a);
      P(deep_wt_L, k) = wt_b; P(deep_wt_R, k) = 1.0;
      const double hr = HR(kR);
      P(h_demand_L, kL) = P(h_demand_L, kL) + 0.5 * hr * wt_b;
      P(h_demand_L, kL - 1) = P(h_demand_L, kL - 1) + 0.5 * hr * (1.0 - wt_b);
      kR = kR + 1; P(sets, k) = 2;
    } else if (rl < rr) {
      const int k = nP++;
      const double rho_pair = rl;
      P(k0b_L, k) = K0L(kL); P(k0b_R, k) = K0R(kR);
      P(k0a_L, k) = K0L(kL); P(k0a_R, k) = K0R(kR - 1);
      P(kbs_Lp, k) = kL; P(kbs_Rp, k) = kR;
      const double rho_a = RR(kR - 1), rho_b = rr;
      double wt_b = 1.0; if (fabs(rho_a - rho_b) > fabs(rho_pair - rho_a)) wt_b = (rho_pair - rho_a) / (rho_b - rho_a);
      P(deep_wt_L, k) = 1.0; P(deep_wt_R, k) = wt_b;
      const double hl = HL(kL);
      P(h_demand_R, kR) = P(h_demand_R, kR) + 0.5 * hl * wt_b;
      P(h_demand_R, kR - 1) = P(h_demand_R, kR - 1) + 0.5 * hl * (1.0 - wt_b);
      kL = kL + 1; P(sets, k) = 1;
    } else if ((K0L(kL) <= nkmb) || (K0R(kR) <= nkmb)) {
      const int k = nP++;
      P(k0b_L, k) = K0L(kL); P(k0b_R, k) = K0R(kR);
      P(k0a_L, k) = K0L(kL); P(k0a_R, k) = K0R(kR);
      P(kbs_Lp, k) = kL; P(kbs_Rp, k) = kR;
      P(deep_wt_L, k) = 1.0; P(deep_wt_R, k) = 1.0;
      P(h_demand_L, kL) = P(h_demand_L, kL) + 0.5 * HR(kR);
      P(h_demand_R, kR) = P(h_demand_R, kR) + 0.5 * HL(kL);
      kL = kL + 1; kR = kR + 1; P(sets, k) = 3;
    } else {
      P(h_demand_L, kL) = P(h_demand_L, kL) + 0.5 * HR(kR);
      P(h_demand_R, kR) = P(h_demand_R, kR) + 0.5 * HL(kL);
      kL = kL + 1; kR = kR + 1;
    }
  }
  A.nP[DIR][f] = nP;
  // h_supply_frac takes the place of h_demand :1013-1023
  for (int k = 1; k <= nR; k++) {
    const double d = P(h_demand_R, k), hh = 0.5 * HR(k);
    P(h_demand_R, k) = (d > hh) ? hh / d : 1.0;
  }
  for (int k = 1; k <= nL; k++) {
    const double d = P(h_demand_L, k), hh = 0.5 * HL(k);
    P(h_demand_L, k) = (d > hh) ? hh / d : 1.0;
  }
  double *h_supply_frac_L = h_demand_L, *h_supply_frac_R = h_demand_R;
  for (int k = 0; k < nP; k++) {      // :1026-1053
    kL = P(kbs_Lp, k); kR = P(kbs_Rp, k);
    const int st = P(sets, k);
    double hpl_ = 0.0, hpr_ = 0.0;
    if (st & 1) {      // left_set
      const double wt_b = P(deep_wt_R, k);
      if (wt_b < 1.0) {
        hpr_ = 0.5 * HL(kL) * min2(P(h_supply_frac_R, kR), P(h_supply_frac_R, kR - 1));
        P(h_used_R, kR - 1) = P(h_used_R, kR - 1) + (1.0 - wt_b) * hpr_;
        P(h_used_R, kR) = P(h_used_R, kR) + wt_b * hpr_;
      } else {
        hpr_ = 0.5 * HL(kL) * P(h_supply_frac_R, kR);
        P(h_used_R, kR) = P(h_used_R, kR) + hpr_;
      }
    }
    if (st & 2) {      // right_set
      const double wt_b = P(deep_wt_L, k);
      if (wt_b < 1.0) {
        hpl_ = 0.5 * HR(kR) * min2(P(h_supply_frac_L, kL), P(h_supply_frac_L, kL - 1));
        P(h_used_L, kL - 1) = P(h_used_L, kL - 1) + (1.0 - wt_b) * hpl_;
        P(h_used_L, kL) = P(h_used_L, kL) + wt_b * hpl_;
      } else {
        hpl_ = 0.5 * HR(kR) * P(h_supply_frac_L, kL);
        P(h_used_L, kL) = P(h_used_L, kL) + hpl_;
      }
    }
    P(hP_L, k) = hpl_; P(hP_R, k) = hpr_;
  }
  for (int k = 0; k < nP; k++) {      // :1057-1062
    const int st = P(sets, k);
    if (st & 1) { const int q = P(kbs_Lp, k); P(hP_L, k) = P(hP_L, k) + (HL(q) - P(h_used_L, q)); }
    if (st & 2) { const int q = P(kbs_Rp, k); P(hP_R, k) = P(hP_R, k) + (HR(q) - P(h_used_R, q)); }
  }
#undef RL
#undef RR
#undef HL
#undef HR
#undef K0L
#undef K0R
}

// :1336-1352 (u) = :1487-1503 (v)
__device__ __forceinline__ double epi_adj_left(double Tr_flux, double Tr_La, double Tr_Lb, double vol, double wt_a, double wt_b,
                                               double Tr_min_face, double Tr_max_face) {
  double Tr_adj_vert = 0.0;
  if (Tr_flux > 0.0) {
    if (Tr_La < Tr_Lb) { if (vol * (Tr_La - Tr_min_face) < Tr_flux)
      Tr_adj_vert = -wt_a * min2(Tr_flux - vol * (Tr_La - Tr_min_face), (vol * wt_b) * (Tr_Lb - Tr_La));
    } else { if (vol * (Tr_Lb - Tr_min_face) < Tr_flux)
      Tr_adj_vert = wt_b * min2(Tr_flux - vol * (Tr_Lb - Tr_min_face), (vol * wt_a) * (Tr_La - Tr_Lb));
    }
  } else if (Tr_flux < 0.0) {
    if (Tr_La > Tr_Lb) { if (vol * (Tr_max_face - Tr_La) < -Tr_flux)
      Tr_adj_vert = wt_a * min2(-Tr_flux - vol * (Tr_max_face - Tr_La), (vol * wt_b) * (Tr_La - Tr_Lb));
    } else { if (vol * (Tr_max_face - Tr_Lb) < -Tr_flux)
      Tr_adj_vert = -wt_b * min2(-Tr_flux - vol * (Tr_max_face - Tr_Lb), (vol * wt_a) * (Tr_Lb - Tr_La));
    }
  }
  return Tr_adj_vert;
}

// :1383-1399 (u) = :1516-1532 (v)
__device__ __forceinline__ double epi_adj_right(double Tr_flux, double Tr_Ra, double Tr_Rb, double vol, double wt_a, double wt_b,
                                                double Tr_min_face, double Tr_max_face) {
  double Tr_adj_vert = 0.0;
  if (Tr_flux < 0.0) {
    if (Tr_Ra < Tr_Rb) { if (vol * (Tr_Ra - Tr_min_face) < -Tr_flux)
      Tr_adj_vert = -wt_a * min2(-Tr_flux - vol * (Tr_Ra - Tr_min_face), (vol * wt_b) * (Tr_Rb - Tr_Ra));
    } else { if (vol * (Tr_Rb - Tr_min_face) < (-Tr_flux))
      Tr_adj_vert = wt_b * min2(-Tr_flux - vol * (Tr_Rb - Tr_min_face), (vol * wt_a) * (Tr_Ra - Tr_Rb));
    }
  } else if (Tr_flux > 0.0) {
    if (Tr_Ra > Tr_Rb) { if (vol * (Tr_max_face - Tr_Ra) < Tr_flux)
      Tr_adj_vert = wt_a * min2(Tr_flux - vol * (Tr_max_face - Tr_Ra), (vol * wt_b) * (Tr_Ra - Tr_Rb));
    } else { if (vol * (Tr_max_face - Tr_Rb) < Tr_flux)
      Tr_adj_vert = -wt_b * min2(Tr_flux - vol * (Tr_max_face - Tr_Rb), (vol * wt_a) * (Tr_Rb - Tr_Ra));
    }
  }
  return Tr_adj_vert;
}

__device__ __forceinline__ double min5(double a, double b, double c, double d, double e) { return min2(min2(min2(min2(a, b), c), d), e); }
__device__ __forceinline__ double max5(double a, double b, double c, double d, double e) { return max2(max2(max2(max2(a, b), c), d), e); }

// thread per face (as epi_pair_kernel): Tr_flux, Tr_adj_vert_L, Tr_adj_vert_R of every pairing of the tracer in hand
template <int DIR>
__global__ __launch_bounds__(256) void epi_flux_kernel(EpiArgs A) {
  const m6::GridDev &g = A.g;
  const int i = (DIR ? g.isc : g.isc - 1) + blockIdx.x * blockDim.x + threadIdx.x, j = (DIR ? g.jsc - 1 : g.jsc) + blockIdx.y;
  if (i > g.iec) return;
  const long hpl = (long)g.nih * g.njh, fpl = DIR ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh;
  const long f = DIR ? g.v2(i, j) : g.u2(i, j), cL = g.h2(i, j), cR = DIR ? g.h2(i, j + 1) : g.h2(i + 1, j);
  const int np2 = 2 * g.nk, nkmb = A.nkmb, nz = g.nk;
  const int nP = A.nP[DIR][f];
  if (nP < 1) return;
  const int *k0b_L = A.kk[DIR] + f, *k0a_L = k0b_L + fpl * np2, *k0b_R = k0a_L + fpl * np2, *k0a_R = k0b_R + fpl * np2;
  const double *deep_wt_L = A.ww[DIR] + f, *deep_wt_R = deep_wt_L + fpl * np2, *hP_L = deep_wt_R + fpl * np2, *hP_R = hP_L + fpl * np2;
  double *flux = A.flx[DIR] + f, *adj_L = flux + fpl * np2, *adj_R = adj_L + fpl * np2;
  const double *T = A.tr, *h = A.h;
#define TL(k) T[hpl * ((k) - 1) + cL]
#define TR(k) T[hpl * ((k) - 1) + cR]
  // the acceptable range of concentrations around the face :1275-1312
  double Tr_min_face = min2(TL(1), TR(1)), Tr_max_face = max2(TL(1), TR(1));
  for (int k = 2; k <= nkmb; k++) {
    const double a = TL(k), b = TR(k);
    Tr_min_face = min2(min2(Tr_min_face, a), b);
    Tr_max_face = max2(max2(Tr_max_face, a), b);
  }
  {
    const int mL = A.max_kRho[cL], mR = A.max_kRho[cR];
    int kLa = nkmb + 1; if (mL < nz + 1) kLa = mL;
    int kLb = kLa; if (mL < nz) kLb = mL + 1;
    int kRa = nkmb + 1; if (mR < nz + 1) kRa = mR;
    int kRb = kRa; if (mR < nz) kRb = mR + 1;
    double Tr_La = Tr_min_face, Tr_Lb = Tr_La, Tr_Ra = Tr_La, Tr_Rb = Tr_La;
    if (h[hpl * (kLa - 1) + cL] > A.h_exclude) Tr_La = TL(kLa);
    if (A.old_answers && A.limit_bug) {
      if (h[hpl * (kLb - 1) + cL] > A.h_exclude) Tr_La = TL(kLb);
    } else {
      if (h[hpl * (kLb - 1) + cL] > A.h_exclude) Tr_Lb = TL(kLb);
    }
    if (h[hpl * (kRa - 1) + cR] > A.h_exclude) Tr_Ra = TR(kRa);
    if (h[hpl * (kRb - 1) + cR] > A.h_exclude) Tr_Rb = TR(kRb);
    Tr_min_face = min5(Tr_min_face, Tr_La, Tr_Lb, Tr_Ra, Tr_Rb);
    Tr_max_face = max5(Tr_max_face, Tr_La, Tr_Lb, Tr_Ra, Tr_Rb);
  }
  for (int k = 0; k < nP; k++) {
    const double Tr_Lb = TL(P(k0b_L, k)), Tr_Rb = TR(P(k0b_R, k));
    double Tr_La = Tr_Lb, Tr_Ra = Tr_Rb;
    if (P(deep_wt_L, k) < 1.0) Tr_La = TL(P(k0a_L, k));
    if (P(deep_wt_R, k) < 1.0) Tr_Ra = TR(P(k0a_R, k));
    Tr_min_face = min5(Tr_min_face, Tr_La, Tr_Lb, Tr_Ra, Tr_Rb);
    Tr_max_face = max5(Tr_max_face, Tr_La, Tr_Lb, Tr_Ra, Tr_Rb);
  }
  const double khdt = A.khdt[DIR][f], areaL = g.areaT[cL], areaR = g.areaT[cR];
  for (int k = 0; k < nP; k++) {      // :1314-1412, :1464-1538
    const double wL = P(deep_wt_L, k), wR = P(deep_wt_R, k);
    const double Tr_Lb = TL(P(k0b_L, k)), Tr_Rb = TR(P(k0b_R, k));
    double Tr_av_L = Tr_Lb, Tr_av_R = Tr_Rb, Tr_La = Tr_Lb, Tr_Ra = Tr_Rb;
    if (wL < 1.0) { Tr_La = TL(P(k0a_L, k)); Tr_av_L = wL * Tr_Lb + (1.0 - wL) * Tr_La; }
    if (wR < 1.0) { Tr_Ra = TR(P(k0a_R, k)); Tr_av_R = wR * Tr_Rb + (1.0 - wR) * Tr_Ra; }
    const double h_L = P(hP_L, k), h_R = P(hP_R, k);
    double Tr_flux;
    if (!DIR && A.old_answers) Tr_flux = A.I_maxitt * khdt * (Tr_av_L - Tr_av_R) * ((2.0 * h_L * h_R) / (h_L + h_R));
    else Tr_flux = A.I_maxitt * ((2.0 * h_L * h_R) / (h_L + h_R)) * khdt * (Tr_av_L - Tr_av_R);
    P(flux, k) = Tr_flux;
    if (wL < 1.0) P(adj_L, k) = epi_adj_left(Tr_flux, Tr_La, Tr_Lb, h_L * areaL, 1.0 - wL, wL, Tr_min_face, Tr_max_face);
    if (wR < 1.0) P(adj_R, k) = epi_adj_right(Tr_flux, Tr_Ra, Tr_Rb, h_R * areaR, 1.0 - wR, wR, Tr_min_face, Tr_max_face);
  }
#undef TL
#undef TR
}

// what the pairings of face f put into column side SIDE (0: the column is the face's left one, 1: its right one); acc: the
// cell's accumulator planes; sgn: -1 / +1 for tr_flux_conv (old answers), +1 for the face sums (new answers)
template <int DIR, int SIDE>
__device__ __forceinline__ void epi_gather(const EpiArgs &A, long f, double *acc, long hpl, double sgn) {
  const m6::GridDev &g = A.g;
  const long fpl = DIR ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh;
  const int np2 = 2 * g.nk;
  const int nP = A.nP[DIR][f];
  const int *k0b = A.kk[DIR] + f + (SIDE ? 2 : 0) * fpl * np2, *k0a = k0b + fpl * np2;
  const double *deep_wt = A.ww[DIR] + f + (SIDE ? 1 : 0) * fpl * np2;
  const double *flux = A.flx[DIR] + f, *adj = flux + (SIDE ? 2 : 1) * fpl * np2;
  for (int k = 0; k < nP; k++) {
    const int kb = P(k0b, k);
    const double wt_b = P(deep_wt, k), F3 = P(flux, k);
    if (wt_b >= 1.0) {
      acc[hpl * (kb - 1)] = acc[hpl * (kb - 1)] + sgn * F3;
    } else {
      const int ka = P(k0a, k);
      const double wt_a = 1.0 - wt_b, av = P(adj, k);
      // left: a gets (wt_a F + adj), b gets (wt_b F - adj); right: a gets (wt_a F - adj), b gets (wt_b F + adj)
      const double ca = SIDE ? (wt_a * F3 - av) : (wt_a * F3 + av), cb = SIDE ? (wt_b * F3 + av) : (wt_b * F3 - av);
      acc[hpl * (ka - 1)] = acc[hpl * (ka - 1)] + sgn * ca;
      acc[hpl * (kb - 1)] = acc[hpl * (kb - 1)] + sgn * cb;
    }
  }
}

// thread (i, j) over the compute domain
__global__ __launch_bounds__(256) void epi_update_kernel(EpiArgs A) {
  const m6::GridDev &g = A.g;
  const int i = g.isc + blockIdx.x * blockDim.x + threadIdx.x, j = g.jsc + blockIdx.y;
  if (i > g.iec) return;
  const long hpl = (long)g.nih * g.njh, c = g.h2(i, j);
  const int nz = g.nk, kmax = *A.PEmax;
  double *T = A.tr + c;
  const long apl = hpl * nz;
  double *acc = A.acc + c;
  const int nacc = A.old_answers ? 1 : 4;
  for (int q = 0; q < nacc; q++) for (int k = 0; k < kmax; k++) acc[apl * q + hpl * k] = 0.0;
  if (A.old_answers) {
    // tr_flux_conv in the order of the reference's loops: the zonal faces (I ascending: the west face, where the cell is the right
    // column, before the east one), then the meridional faces the same way (:1541-1565)
    epi_gather<0, 1>(A, g.u2(i - 1, j), acc, hpl, 1.0);
    epi_gather<0, 0>(A, g.u2(i, j), acc, hpl, -1.0);
    epi_gather<1, 1>(A, g.v2(i, j - 1), acc, hpl, 1.0);
    epi_gather<1, 0>(A, g.v2(i, j), acc, hpl, -1.0);
  } else {      // tr_flux_E, _W, _N, _S and their symmetric sum :1566-1594
    double *E = acc, *W = acc + apl, *N = acc + 2 * apl, *S = acc + 3 * apl;
    epi_gather<0, 1>(A, g.u2(i - 1, j), W, hpl, 1.0);
    epi_gather<0, 0>(A, g.u2(i, j), E, hpl, 1.0);
    epi_gather<1, 1>(A, g.v2(i, j - 1), S, hpl, 1.0);
    epi_gather<1, 0>(A, g.v2(i, j), N, hpl, 1.0);
    for (int k = 0; k < kmax; k++) E[hpl * k] = ((W[hpl * k] - E[hpl * k]) + (S[hpl * k] - N[hpl * k]));
  }
  const double area = g.areaT[c];
  const bool wet = g.mask2dT[c] > 0.0;
  for (int k = 0; k < nz; k++) {
    double t = T[hpl * k];
    if (k < kmax) {
      const double hk = A.h[hpl * k + c];
      if (wet && (hk > 0.0)) t = t + acc[hpl * k] / (hk * area);
    }
    if (A.cu > 0.0 && fabs(t) < A.cu) t = 0.0;
    T[hpl * k] = t;
  }
}
#undef P

}  // namespace

namespace m6 {

// tracer_epipycnal_ML_diff on device arrays (the tail of tracer_hordiff with CS%Diffuse_ML_interior, :613-620); khdt_x, khdt_y and
// the iteration count are those tracer_hordiff has formed
int epipycnal_branch(mom6hip_ctx_t *ctx, Stager &st, const mom6hip_epipycnal_cs_t *epi, const mom6hip_eos_t *eos, const double *h,
                     const double *khdt_x, const double *khdt_y, int num_itts, const std::vector<double *> &d_tr,
                     const std::vector<double> &cu, int idx_T, int idx_S, int *halo_updates) {
  const int ntr = (int)d_tr.size();
  const m6::GridDev g = ctx->g;
  M6_REQUIRE(epi != nullptr && eos != nullptr && epi->Rlay != nullptr, "tracer_hordiff: DIFFUSE_ML_TO_INTERIOR needs its control structure with GV%%Rlay, and tv%%eqn_of_state");
  M6_REQUIRE(idx_T >= 0 && idx_T < ntr && idx_S >= 0 && idx_S < ntr, "tracer_hordiff: tv%%T and tv%%S must be among the tracers (idx_T, idx_S)");
  M6_REQUIRE(epi->nk_rho_varies >= 1 && epi->nk_rho_varies < g.nk && epi->nkml >= 0 && epi->nkml <= epi->nk_rho_varies,
             "tracer_epipycnal_ML_diff: 1 <= nk_rho_varies < nk and 0 <= nkml <= nk_rho_varies are needed (a layered run with a bulk mixed layer)");
  M6_REQUIRE(g.isc - g.isd >= 2 && g.jsc - g.jsd >= 2, "tracer_epipycnal_ML_diff: the halo must be at least 2 points wide");
  M6_REQUIRE(g.mask2dT && g.mask2dCu && g.mask2dCv && g.areaT, "tracer_epipycnal_ML_diff: mask2dT, mask2dCu, mask2dCv and areaT are needed");
  hipStream_t s = ctx->stream;
  const int nk = g.nk, np2 = 2 * nk;
  const size_t hpl = (size_t)g.nih * g.njh, upl = (size_t)(g.nih + 1) * g.njh, vpl = (size_t)g.nih * (g.njh + 1);
  EpiArgs A;
  A.g = g; A.E = EosDev{eos->form, eos->Rho_T0_S0, eos->dRho_dT, eos->dRho_dS};
  A.P_Ref = epi->P_Ref; A.h_exclude = 10.0 * (g.Angstrom_H + g.H_subroundoff);
  A.nkmb = epi->nk_rho_varies; A.old_answers = epi->answer_date <= 20240330; A.limit_bug = epi->limit_bug != 0;
  int max_itt = 1; A.I_maxitt = 1.0;
  if (num_itts > 1) { max_itt = num_itts; A.I_maxitt = 1.0 / ((double)max_itt); }
  A.h = h; A.T = d_tr[idx_T]; A.S = d_tr[idx_S]; A.khdt[0] = khdt_x; A.khdt[1] = khdt_y;
  double *d_Rlay = (double *)st.scratch(sizeof(double) * nk);
  A.rho_coord = (double *)st.scratch(sizeof(double) * hpl * A.nkmb);
  int *icol = (int *)st.scratch(sizeof(int) * (hpl * (2 + (size_t)nk) + 4));
  A.max_kRho = icol; A.num_srt = icol + hpl; A.k0_srt = icol + 2 * hpl; A.PEmax = icol + hpl * (2 + (size_t)nk);
  A.rho_srt = (double *)st.scratch(sizeof(double) * hpl * nk * 2); A.h_srt = A.rho_srt + hpl * nk;
  const size_t fpl[2] = {upl, vpl};
  for (int d = 0; d < 2; d++) {
    A.nP[d] = (int *)st.scratch(sizeof(int) * fpl[d] * (1 + 7 * (size_t)np2));
    A.kk[d] = A.nP[d] + fpl[d]; A.kbs[d] = A.kk[d] + fpl[d] * np2 * 4;
    A.ww[d] = (double *)st.scratch(sizeof(double) * fpl[d] * ((size_t)np2 * 7 + 4 * (size_t)nk));
    A.flx[d] = A.ww[d] + fpl[d] * np2 * 4; A.work[d] = A.flx[d] + fpl[d] * np2 * 3;
  }
  A.acc = (double *)st.scratch(sizeof(double) * hpl * nk * (A.old_answers ? 1 : 4));
  M6_REQUIRE(!st.failed() && d_Rlay && A.rho_coord && icol && A.rho_srt && A.nP[0] && A.nP[1] && A.ww[0] && A.ww[1] && A.acc,
             "tracer_epipycnal_ML_diff: staging failed");
  A.Rlay = d_Rlay;
  M6_HIP(hipMemcpyAsync(d_Rlay, epi->Rlay, sizeof(double) * nk, hipMemcpyHostToDevice, s));
  M6_HIP(hipMemsetAsync(icol, 0, sizeof(int) * (hpl * (2 + (size_t)nk) + 4), s));
  for (int d = 0; d < 2; d++) M6_HIP(hipMemsetAsync(A.nP[d], 0, sizeof(int) * fpl[d], s));
  M6_HIP(hipStreamSynchronize(s));      // epi->Rlay is the caller's

  std::vector<double *> pf(d_tr);
  std::vector<int32_t> ppos(ntr, MOM6HIP_POS_H), pnk(ntr, nk);
  if (int rc = m6::group_pass(ctx, pf.data(), ppos.data(), pnk.data(), ntr)) return rc;      // :832
  (*halo_updates)++;
  const int ni = g.iec - g.isc + 1, nj = g.jec - g.jsc + 1;
  hipLaunchKernelGGL(epi_column_kernel, dim3((ni + 4 + 255) / 256, nj + 4), dim3(256), 0, s, A);
  hipLaunchKernelGGL(epi_sort_kernel, dim3((ni + 2 + 255) / 256, nj + 2), dim3(256), 0, s, A);
  hipLaunchKernelGGL(epi_pair_kernel<0>, dim3((ni + 1 + 255) / 256, nj), dim3(256), 0, s, A);
  hipLaunchKernelGGL(epi_pair_kernel<1>, dim3((ni + 255) / 256, nj + 1), dim3(256), 0, s, A);
  M6_HIP(hipGetLastError());
  for (int itt = 1; itt <= max_itt; itt++) {      // :1255-1610
    if (itt > 1) {
      if (int rc = m6::group_pass(ctx, pf.data(), ppos.data(), pnk.data(), ntr)) return rc;
      (*halo_updates)++;
    }
    for (int m = 0; m < ntr; m++) {
      A.tr = d_tr[m]; A.cu = cu[m];
      hipLaunchKernelGGL(epi_flux_kernel<0>, dim3((ni + 1 + 255) / 256, nj), dim3(256), 0, s, A);
      hipLaunchKernelGGL(epi_flux_kernel<1>, dim3((ni + 255) / 256, nj + 1), dim3(256), 0, s, A);
      hipLaunchKernelGGL(epi_update_kernel, dim3((ni + 255) / 256, nj), dim3(256), 0, s, A);
    }
  }
  M6_HIP(hipGetLastError());
  return 0;
}

}  // namespace m6
