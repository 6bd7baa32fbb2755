// thickness_diffuse.hip -- thickness_diffuse (isopycnal height diffusion, Gent-McWilliams) of
// src/parameterizations/lateral/MOM_thickness_diffuse.F90 (:133-629, thickness_diffuse_full :634-1670) on gfx950, with vert_fill_TS
// (src/core/MOM_isopycnal_slopes.F90:541-629) and the Boussinesq find_eta (src/core/MOM_interface_heights.F90:48-112).
//
// Three kinds of kernels, all with one lane per column and lanes along i (every global access i-contiguous):
//   td_column_kernel   the cells is-1 .. ie+1: interface heights (bottom up), interface pressures and the running sum of the
//                      thickness available for the overturning (top down), T and S smoothed in the vertical by an implicit
//                      diffusion (a tridiagonal solve: forward sweep, c1 parked in global scratch, back substitution);
//   td_face_kernel<d>  the faces: ONE bottom-up sweep forms, interface by interface, the density derivatives at the face, the
//                      isoneutral slope, the unlimited streamfunction, its slope-limited and thickness-limited value and the
//                      layer transport.  The reference makes two sweeps (streamfunction, then transports: :812-1100 and
//                      :1126-1209) because the FGNV elliptic solve sits between them; without it the second needs of the first
//                      only the values of the same interface, so they are one loop here and nothing per interface is stored.
//                      The work against the stratification accumulates in the same sweep;
//   td_update_kernel   uhtr, vhtr, h (:607-620) and MEKE%GM_src (:1552-1560).
// Algorithmic traffic: read h, T, S; write/read T_f, S_f, e, pres, h_avail_rsum; write uhD, vhD; update h, uhtr, vhtr
// (~200 B per cell and call; the call runs once per dynamics step, MOM.F90:1165).
#include <cmath>

#include "common.hpp"
#include "eos.hpp"

namespace {

using m6::max2;
using m6::min2;
using namespace m6::eos;

struct TDArgs {
  m6::GridDev g;
  EosDev E;
  int use_EOS, find_work, nkml, use_GM_work_bug, use_Visbeck;
  double Khth, Khth_Min, Khth_Max, max_Khth_CFL, slope_max, kappa_dt, KHTH_Slope_Cff, KhTh_fac, dt;
  const double *h, *T_in, *S_in;
  double *e, *pres, *rsum, *T, *S, *c1;      // column scratch (e, pres, rsum: nk+1 interfaces)
  const double *MEKE_Kh, *L2, *SN, *Res_fn, *Depth_fn, *slope, *Rlay;      // of the direction of the launch
  double *hD, *Work;
  // KHTH_USE_FGNV_STREAMFUNCTION (td_face_fgnv_kernel): what the first sweep hands the second, nk+1 planes of faces each
  int use_FGNV;
  double FGNV_scale, N2_floor;
  const double *cg1, *g_prime;
  double *W_sfn, *W_s2r, *W_n2, *W_dkde, *W_dik;
};

__global__ __launch_bounds__(64) void td_column_kernel(TDArgs A) {
  const m6::GridDev &g = A.g;
  const int i = g.isc - 1 + blockIdx.x * 64 + threadIdx.x, j = g.jsc - 1 + blockIdx.y;
  if (i > g.iec + 1) return;
  const int nz = g.nk;
  const long hpl = (long)g.nih * g.njh, c = g.h2(i, j);
  // find_eta(h, tv, G, GV, US, e, halo_size=1) :226, Boussinesq
  double ek = -(g.bathyT[c] + 0.0);
  A.e[c + hpl * nz] = ek;
  for (int k = nz - 1; k >= 0; k--) { ek = ek + A.h[c + hpl * k] * g.H_to_Z; A.e[c + hpl * k] = ek; }
  // pres, h_avail_rsum :786-806
  const double I4dt = 0.25 / A.dt, aT = g.areaT[c], gH = g.g_Earth * (g.Rho0 * g.H_to_Z);
  double p = 0.0, rs = 0.0;
  A.pres[c] = p; A.rsum[c] = rs;
  for (int k = 0; k < nz; k++) {
    const double hk = A.h[c + hpl * k];
    const double h_avail = max2(I4dt * aT * (hk - g.Angstrom_H), 0.0);
    rs = (k == 0) ? h_avail : rs + h_avail;
    p = p + gH * hk;
    A.rsum[c + hpl * (k + 1)] = rs; A.pres[c + hpl * (k + 1)] = p;
  }
  if (!A.use_EOS) return;
  // vert_fill_TS :541-629 (larger_h_denom)
  const double h_neglect = g.H_subroundoff;
  const double kap_dt_x2 = (2.0 * A.kappa_dt) * (1.0 * g.Z_to_H);
  if (kap_dt_x2 <= 0.0 || nz < 2) {
    for (int k = 0; k < nz; k++) { A.T[c + hpl * k] = A.T_in[c + hpl * k]; A.S[c + hpl * k] = A.S_in[c + hpl * k]; }
    return;
  }
  const double h0 = 1.0e-16 * sqrt(0.5 * kap_dt_x2);
  double h_here = A.h[c], h_next = A.h[c + hpl];
  double ent_K = kap_dt_x2 / ((h_here + h_next) + h0);      // ent(2)
  double h_tr = h_here + h_neglect;
  double b1 = 1.0 / (h_tr + ent_K);
  double d1 = b1 * h_tr;
  double Tf = (b1 * h_tr) * A.T_in[c], Sf = (b1 * h_tr) * A.S_in[c];
  A.T[c] = Tf; A.S[c] = Sf;
  for (int k = 1; k < nz - 1; k++) {      // layers 2 .. nz-1
    h_here = h_next; h_next = A.h[c + hpl * (k + 1)];
    const double ent_Kp1 = kap_dt_x2 / ((h_here + h_next) + h0);
    h_tr = h_here + h_neglect;
    A.c1[c + hpl * k] = ent_K * b1;
    b1 = 1.0 / ((h_tr + d1 * ent_K) + ent_Kp1);
    d1 = b1 * (h_tr + d1 * ent_K);
    Tf = b1 * (h_tr * A.T_in[c + hpl * k] + ent_K * Tf);
    Sf = b1 * (h_tr * A.S_in[c + hpl * k] + ent_K * Sf);
    A.T[c + hpl * k] = Tf; A.S[c + hpl * k] = Sf;
    ent_K = ent_Kp1;
  }
  A.c1[c + hpl * (nz - 1)] = ent_K * b1;
  h_tr = h_next + h_neglect;
  b1 = 1.0 / (h_tr + d1 * ent_K);
  Tf = b1 * (h_tr * A.T_in[c + hpl * (nz - 1)] + ent_K * Tf);
  Sf = b1 * (h_tr * A.S_in[c + hpl * (nz - 1)] + ent_K * Sf);
  A.T[c + hpl * (nz - 1)] = Tf; A.S[c + hpl * (nz - 1)] = Sf;
  for (int k = nz - 2; k >= 0; k--) {
    const double c1 = A.c1[c + hpl * (k + 1)];
    Tf = A.T[c + hpl * k] + c1 * Tf;
    Sf = A.S[c + hpl * k] + c1 * Sf;
    A.T[c + hpl * k] = Tf; A.S[c + hpl * k] = Sf;
  }
}

// One velocity column of thickness_diffuse_full (:812-1209 / :1211-1515, the top layer :1517-1590)
#ifndef TD_FACE_OCC
#define TD_FACE_OCC 3      // waves per SIMD the register allocation aims at (tools/build_variant.sh for experiments)
#endif
template <int DIR>
__global__ __launch_bounds__(64, TD_FACE_OCC) void td_face_kernel(TDArgs A) {
  const m6::GridDev &g = A.g;
  const int i = (DIR ? g.isc : g.isc - 1) + blockIdx.x * 64 + threadIdx.x;
  const int j = (DIR ? g.jsc - 1 : g.jsc) + blockIdx.y;
  if (i > g.iec) return;
  const int nz = g.nk;
  const long hpl = (long)g.nih * g.njh, fpl = DIR ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh;
  const long cL = g.h2(i, j), cR = DIR ? g.h2(i, j + 1) : g.h2(i + 1, j);
  const long f2 = DIR ? g.v2(i, j) : g.u2(i, j);
  // ---- the diffusivity of the face :204-262 / :340-400
  const double IdxC = DIR ? g.IdxCv[f2] : g.IdxCu[f2], IdyC = DIR ? g.IdyCv[f2] : g.IdyCu[f2];
  const double KH_CFL = (0.25 * A.max_Khth_CFL) / (A.dt * (IdxC * IdxC + IdyC * IdyC));
  double Kh = A.Khth;
  if (A.use_Visbeck) Kh = Kh + A.KHTH_Slope_Cff * A.L2[f2] * A.SN[f2];
  if (A.MEKE_Kh) Kh = Kh + A.KhTh_fac * sqrt(A.MEKE_Kh[cL] * A.MEKE_Kh[cR]);
  if (A.Res_fn) Kh = Kh * A.Res_fn[f2];
  if (A.Depth_fn) Kh = Kh * A.Depth_fn[f2];      // DEPTH_SCALED_KHTH :284-289
  if (A.Khth_Max > 0) Kh = max2(A.Khth_Min, min2(Kh, A.Khth_Max));
  else Kh = max2(A.Khth_Min, Kh);
  const double KH = min2(KH_CFL, Kh);

  const bool use_EOS = A.use_EOS, find_work = A.find_work, present_slope = A.slope != nullptr;
  const int nk_linear = A.nkml > 1 ? A.nkml : 1;
  const double I_slope_max2 = 1.0 / (A.slope_max * A.slope_max);
  const double h_neglect = g.H_subroundoff, h_neglect2 = h_neglect * h_neglect, dz_neglect = g.dZ_subroundoff;
  const double IdL = DIR ? IdyC : IdxC;
  const double dLf = DIR ? g.dx_Cv[f2] : g.dy_Cu[f2];
  const double OBCmask = DIR ? g.mask2dCv[f2] : g.mask2dCu[f2];
  const double G_scale = g.g_Earth * g.H_to_Z;
  const double I4dt = 0.25 / A.dt, aL = g.areaT[cL], aR = g.areaT[cR];
  const double e_botL = A.e[cL + hpl * nz], e_botR = A.e[cR + hpl * nz];
  double htot = 0.0, Work = 0.0;
  double drdiA = 0.0, drdiB = 0.0, drdkL = 0.0, drdkR = 0.0, drdkDe = 0.0;
  // values of layer k and interface K+1 are handed down the sweep as those of layer k-1 / interface K are loaded
  double hL_k = A.h[cL + hpl * (nz - 1)], hR_k = A.h[cR + hpl * (nz - 1)];
  double eL_Kp1 = e_botL, eR_Kp1 = e_botR;
  double TL_k = 0.0, TR_k = 0.0, SL_k = 0.0, SR_k = 0.0;
  if (use_EOS) {
    TL_k = A.T[cL + hpl * (nz - 1)]; TR_k = A.T[cR + hpl * (nz - 1)]; SL_k = A.S[cL + hpl * (nz - 1)]; SR_k = A.S[cR + hpl * (nz - 1)];
  }
  for (int K = nz; K >= 2; K--) {
    const int k = K;      // the layer below interface K (one-based, as in the reference)
    const long oK = hpl * (K - 1), okm1 = hpl * (k - 2);
    const double hL_km1 = A.h[cL + okm1], hR_km1 = A.h[cR + okm1];
    const double eL_K = A.e[cL + oK], eR_K = A.e[cR + oK];
    double TL_km1 = 0.0, TR_km1 = 0.0, SL_km1 = 0.0, SR_km1 = 0.0;
    if (use_EOS) { TL_km1 = A.T[cL + okm1]; TR_km1 = A.T[cR + okm1]; SL_km1 = A.S[cL + okm1]; SR_km1 = A.S[cR + okm1]; }
    double Sfn_unlim, slope2_Ratio = 0.0, drdi_k = 0.0;
    if (find_work && !use_EOS) {      // :824-828
      drdiA = 0.0; drdiB = 0.0;
      drdkL = A.Rlay[k - 1] - A.Rlay[k - 2]; drdkR = drdkL;
    }
    const bool calc_derivatives = use_EOS && (k >= nk_linear) && (find_work || !present_slope);
    if (calc_derivatives) {      // :833-842, :855-865
      const double pres_u = 0.5 * (A.pres[cL + oK] + A.pres[cR + oK]);
      const double T_u = 0.25 * ((TL_k + TR_k) + (TL_km1 + TR_km1));
      const double S_u = 0.25 * ((SL_k + SR_k) + (SL_km1 + SR_km1));
      double drho_dT, drho_dS;
      eos_density_derivs(A.E, T_u, S_u, pres_u, drho_dT, drho_dS);
      drdiA = drho_dT * (TR_km1 - TL_km1) + drho_dS * (SR_km1 - SL_km1);
      drdiB = drho_dT * (TR_k - TL_k) + drho_dS * (SR_k - SL_k);
      drdkL = (drho_dT * (TL_k - TL_km1) + drho_dS * (SL_k - SL_km1));
      drdkR = (drho_dT * (TR_k - TR_km1) + drho_dS * (SR_k - SR_km1));
      drdkDe = drdkR * eR_K - drdkL * eL_K;
    } else if (find_work) {
      drdkDe = drdkR * eR_K - drdkL * eL_K;
    }
    if (find_work) drdi_k = drdiB;
    if (k > nk_linear) {
      if (use_EOS) {
        double hg2A = 0.0, hg2B = 0.0, haA = 0.0, haB = 0.0, drdz = 0.0, Slope;
        if (find_work || !present_slope) {      // :882-917 (Boussinesq)
          const double hg2L = hL_km1 * hL_k + h_neglect2;
          const double hg2R = hR_km1 * hR_k + h_neglect2;
          const double haL = 0.5 * (hL_km1 + hL_k) + h_neglect;
          const double haR = 0.5 * (hR_km1 + hR_k) + h_neglect;
          const double dzaL = haL * g.H_to_Z, dzaR = haR * g.H_to_Z;
          const double wtL = hg2L * (haR * dzaR), wtR = hg2R * (haL * dzaL);
          drdz = (wtL * drdkL + wtR * drdkR) / (dzaL * wtL + dzaR * wtR);
          hg2A = hL_km1 * hR_km1 + h_neglect2;
          hg2B = hL_k * hR_k + h_neglect2;
          haA = 0.5 * (hL_km1 + hR_km1) + h_neglect;
          haB = 0.5 * (hL_k + hR_k) + h_neglect;
        }
        if (present_slope) {      // :919-921
          Slope = A.slope[f2 + fpl * (K - 1)];
          slope2_Ratio = (Slope * Slope) * I_slope_max2;
        } else {                  // :922-936
          const double wtA = hg2A * haB, wtB = hg2B * haA;
          const double drdx = ((wtA * drdiA + wtB * drdiB) / (wtA + wtB) - drdz * (eL_K - eR_K)) * IdL;
          const double mag_grad2 = (1.0 * drdx) * (1.0 * drdx) + drdz * drdz;
          if (mag_grad2 > 0.0) {
            Slope = drdx / sqrt(mag_grad2);
            slope2_Ratio = (Slope * Slope) * I_slope_max2;
          } else {
            Slope = 0.0;
            slope2_Ratio = 1.0e20;
          }
        }
        // :939-943 with int_slope = 0
        Slope = (1.0 - 0.0) * Slope + 0.0 * ((eR_K - eL_K) * IdL);
        slope2_Ratio = (1.0 - 0.0) * slope2_Ratio;
        Sfn_unlim = -(KH * dLf) * Slope;      // :956
        if (Sfn_unlim > 0.0) {                 // :959-975
          if (eL_K < e_botR) Sfn_unlim = 0.0;
          else if (e_botR > eL_Kp1) Sfn_unlim = Sfn_unlim * ((eL_K - e_botR) / ((eL_K - eL_Kp1) + dz_neglect));
        } else {
          if (eR_K < e_botL) Sfn_unlim = 0.0;
          else if (e_botL > eR_Kp1) Sfn_unlim = Sfn_unlim * ((eR_K - e_botL) / ((eR_K - eR_Kp1) + dz_neglect));
        }
      } else {      // :979-988
        double Slope;
        if (present_slope) Slope = A.slope[f2 + fpl * (K - 1)];
        else Slope = ((eL_K - eR_K) * IdL) * OBCmask;
        Sfn_unlim = ((KH * dLf) * Slope);
      }
    } else {
      Sfn_unlim = 0.;
    }
    // ---- the transport of layer k :1148-1209
    const double availL = max2(I4dt * aL * (hL_k - g.Angstrom_H), 0.0), availR = max2(I4dt * aR * (hR_k - g.Angstrom_H), 0.0);
    const double rsumL_Kp1 = A.rsum[cL + hpl * K], rsumR_Kp1 = A.rsum[cR + hpl * K];
    double fracL = 0.0, fracR = 0.0;      // h_frac(k), k >= 2
    if (availL > 0.0) fracL = availL / rsumL_Kp1;
    if (availR > 0.0) fracR = availR / rsumR_Kp1;
    const double Z_to_H = g.Z_to_H;
    double hDk;
    if (k > nk_linear) {
      double Sfn_est;
      if (use_EOS) {
        double Sfn_safe;
        if (htot <= 0.0) Sfn_safe = htot * (1.0 - fracL);
        else Sfn_safe = htot * (1.0 - fracR);
        Sfn_est = (Z_to_H * Sfn_unlim + slope2_Ratio * Sfn_safe) / (1.0 + slope2_Ratio);
      } else {
        Sfn_est = Z_to_H * Sfn_unlim;
      }
      const double Sfn_in_H = min2(max2(Sfn_est, -A.rsum[cL + oK]), A.rsum[cR + oK]);
      hDk = max2(min2((Sfn_in_H - htot), availL), -availR);
    } else {
      if (htot <= 0.0) hDk = -htot * fracL;
      else hDk = -htot * fracR;
    }
    A.hD[f2 + fpl * (k - 1)] = hDk;
    htot = htot + hDk;
    if (find_work)      // :1196-1209
      Work = Work + G_scale * (htot * drdkDe - (hDk * drdi_k) * 0.25 * ((eL_K + eL_Kp1) + (eR_K + eR_Kp1)));
    hL_k = hL_km1; hR_k = hR_km1; eL_Kp1 = eL_K; eR_Kp1 = eR_K;
    TL_k = TL_km1; TR_k = TR_km1; SL_k = SL_km1; SR_k = SR_km1;
  }
  // ---- the top layer :1517-1590 (after the sweep the handed-down values are those of layer 1 and interface 2)
  const double hD1 = -htot;
  A.hD[f2] = hD1;
  if (find_work && use_EOS) {
    const double pres_u = 0.5 * (A.pres[cL] + A.pres[cR]);
    const double T_u = 0.5 * (TL_k + TR_k), S_u = 0.5 * (SL_k + SR_k);
    double drho_dT, drho_dS;
    eos_density_derivs(A.E, T_u, S_u, pres_u, drho_dT, drho_dS);
    const double drdiB1 = drho_dT * (TR_k - TL_k) + drho_dS * (SR_k - SL_k);
    const double eL_1 = A.e[cL], eR_1 = A.e[cR];
    const double w = G_scale * ((hD1 * drdiB1) * 0.25 * ((eL_1 + eL_Kp1) + (eR_1 + eR_Kp1)));
    if (DIR == 0 && A.use_GM_work_bug) Work = Work + w;
    else Work = Work - w;
  }
  if (A.Work) A.Work[f2] = Work;
}

// The same column with KHTH_USE_FGNV_STREAMFUNCTION: the two sweeps of the reference apart, the elliptic solve between them; what the first hands
// the second goes through planes of faces in global memory (td_face_kernel fuses the sweeps, which the solve forbids)
template <int DIR>
__global__ __launch_bounds__(64, TD_FACE_OCC) void td_face_fgnv_kernel(TDArgs A) {
  const m6::GridDev &g = A.g;
  const int i = (DIR ? g.isc : g.isc - 1) + blockIdx.x * 64 + threadIdx.x;
  const int j = (DIR ? g.jsc - 1 : g.jsc) + blockIdx.y;
  if (i > g.iec) return;
  const int nz = g.nk;
  const long hpl = (long)g.nih * g.njh, fpl = DIR ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh;
  const long cL = g.h2(i, j), cR = DIR ? g.h2(i, j + 1) : g.h2(i + 1, j);
  const long f2 = DIR ? g.v2(i, j) : g.u2(i, j);
  // ---- the diffusivity of the face :204-262 / :340-400
  const double IdxC = DIR ? g.IdxCv[f2] : g.IdxCu[f2], IdyC = DIR ? g.IdyCv[f2] : g.IdyCu[f2];
  const double KH_CFL = (0.25 * A.max_Khth_CFL) / (A.dt * (IdxC * IdxC + IdyC * IdyC));
  double Kh = A.Khth;
  if (A.use_Visbeck) Kh = Kh + A.KHTH_Slope_Cff * A.L2[f2] * A.SN[f2];
  if (A.MEKE_Kh) Kh = Kh + A.KhTh_fac * sqrt(A.MEKE_Kh[cL] * A.MEKE_Kh[cR]);
  if (A.Res_fn) Kh = Kh * A.Res_fn[f2];
  if (A.Depth_fn) Kh = Kh * A.Depth_fn[f2];      // DEPTH_SCALED_KHTH :284-289
  if (A.Khth_Max > 0) Kh = max2(A.Khth_Min, min2(Kh, A.Khth_Max));
  else Kh = max2(A.Khth_Min, Kh);
  const double KH = min2(KH_CFL, Kh);

  const bool use_EOS = A.use_EOS, find_work = A.find_work, present_slope = A.slope != nullptr;
  const int nk_linear = A.nkml > 1 ? A.nkml : 1;
  const double I_slope_max2 = 1.0 / (A.slope_max * A.slope_max);
  const double h_neglect = g.H_subroundoff, h_neglect2 = h_neglect * h_neglect, dz_neglect = g.dZ_subroundoff;
  const double IdL = DIR ? IdyC : IdxC;
  const double dLf = DIR ? g.dx_Cv[f2] : g.dy_Cu[f2];
  const double OBCmask = DIR ? g.mask2dCv[f2] : g.mask2dCu[f2];
  const double G_scale = g.g_Earth * g.H_to_Z;
  const double G_rho0 = g.g_Earth / g.Rho0, dz_neglect2 = dz_neglect * dz_neglect;
  const double I4dt = 0.25 / A.dt, aL = g.areaT[cL], aR = g.areaT[cR];
  const double e_botL = A.e[cL + hpl * nz], e_botR = A.e[cR + hpl * nz];
  double htot = 0.0, Work = 0.0;
  double drdiA = 0.0, drdiB = 0.0, drdkL = 0.0, drdkR = 0.0, drdkDe = 0.0;
  // values of layer k and interface K+1 are handed down the sweep as those of layer k-1 / interface K are loaded
  double hL_k = A.h[cL + hpl * (nz - 1)], hR_k = A.h[cR + hpl * (nz - 1)];
  double eL_Kp1 = e_botL, eR_Kp1 = e_botR;
  double TL_k = 0.0, TR_k = 0.0, SL_k = 0.0, SR_k = 0.0;
  if (use_EOS) {
    TL_k = A.T[cL + hpl * (nz - 1)]; TR_k = A.T[cR + hpl * (nz - 1)]; SL_k = A.S[cL + hpl * (nz - 1)]; SR_k = A.S[cR + hpl * (nz - 1)];
  }
  for (int K = nz; K >= 2; K--) {
    const int k = K;      // the layer below interface K (one-based, as in the reference)
    const long oK = hpl * (K - 1), okm1 = hpl * (k - 2);
    const double hL_km1 = A.h[cL + okm1], hR_km1 = A.h[cR + okm1];
    const double eL_K = A.e[cL + oK], eR_K = A.e[cR + oK];
    double TL_km1 = 0.0, TR_km1 = 0.0, SL_km1 = 0.0, SR_km1 = 0.0;
    if (use_EOS) { TL_km1 = A.T[cL + okm1]; TR_km1 = A.T[cR + okm1]; SL_km1 = A.S[cL + okm1]; SR_km1 = A.S[cR + okm1]; }
    double Sfn_unlim, slope2_Ratio = 0.0, drdi_k = 0.0, dzN2 = 0.0;
    drdkDe = 0.0;
    if (find_work && !use_EOS) {      // :824-828
      drdiA = 0.0; drdiB = 0.0;
      drdkL = A.Rlay[k - 1] - A.Rlay[k - 2]; drdkR = drdkL;
    }
    const bool calc_derivatives = use_EOS && (k >= nk_linear) && (find_work || !present_slope || true);      // (use_FGNV_streamfn, :926)
    if (calc_derivatives) {      // :833-842, :855-865
      const double pres_u = 0.5 * (A.pres[cL + oK] + A.pres[cR + oK]);
      const double T_u = 0.25 * ((TL_k + TR_k) + (TL_km1 + TR_km1));
      const double S_u = 0.25 * ((SL_k + SR_k) + (SL_km1 + SR_km1));
      double drho_dT, drho_dS;
      eos_density_derivs(A.E, T_u, S_u, pres_u, drho_dT, drho_dS);
      drdiA = drho_dT * (TR_km1 - TL_km1) + drho_dS * (SR_km1 - SL_km1);
      drdiB = drho_dT * (TR_k - TL_k) + drho_dS * (SR_k - SL_k);
      drdkL = (drho_dT * (TL_k - TL_km1) + drho_dS * (SL_k - SL_km1));
      drdkR = (drho_dT * (TR_k - TR_km1) + drho_dS * (SR_k - SR_km1));
      drdkDe = drdkR * eR_K - drdkL * eL_K;
    } else if (find_work) {
      drdkDe = drdkR * eR_K - drdkL * eL_K;
    }
    if (find_work) drdi_k = drdiB;
    if (k > nk_linear) {
      if (use_EOS) {
        double hg2A = 0.0, hg2B = 0.0, haA = 0.0, haB = 0.0, drdz = 0.0, Slope;
        {      // :982-1024 (use_FGNV_streamfn; Boussinesq)
          const double hg2L = hL_km1 * hL_k + h_neglect2;
          const double hg2R = hR_km1 * hR_k + h_neglect2;
          const double haL = 0.5 * (hL_km1 + hL_k) + h_neglect;
          const double haR = 0.5 * (hR_km1 + hR_k) + h_neglect;
          const double dzaL = haL * g.H_to_Z, dzaR = haR * g.H_to_Z;
          const double wtL = hg2L * (haR * dzaR), wtR = hg2R * (haL * dzaL);
          drdz = (wtL * drdkL + wtR * drdkR) / (dzaL * wtL + dzaR * wtR);
          hg2A = hL_km1 * hR_km1 + h_neglect2;
          hg2B = hL_k * hR_k + h_neglect2;
          haA = 0.5 * (hL_km1 + hR_km1) + h_neglect;
          haB = 0.5 * (hL_k + hR_k) + h_neglect;
          const double N2_unlim = drdz * G_rho0;
          const double dzLm = g.H_to_Z * hL_km1, dzRm = g.H_to_Z * hR_km1, dzLk = g.H_to_Z * hL_k, dzRk = g.H_to_Z * hR_k;      // thickness_to_dz
          const double dzg2A = dzLm * dzRm + dz_neglect2;
          const double dzg2B = dzLk * dzRk + dz_neglect2;
          const double dzaA = 0.5 * (dzLm + dzRm) + dz_neglect;
          const double dzaB = 0.5 * (dzLk + dzRk) + dz_neglect;
          dzN2 = (0.5 * (dzg2A / dzaA + dzg2B / dzaB)) * max2(N2_unlim, A.N2_floor);      // :1021
        }
        if (present_slope) {      // :919-921
          Slope = A.slope[f2 + fpl * (K - 1)];
          slope2_Ratio = (Slope * Slope) * I_slope_max2;
        } else {                  // :922-936
          const double wtA = hg2A * haB, wtB = hg2B * haA;
          const double drdx = ((wtA * drdiA + wtB * drdiB) / (wtA + wtB) - drdz * (eL_K - eR_K)) * IdL;
          const double mag_grad2 = (1.0 * drdx) * (1.0 * drdx) + drdz * drdz;
          if (mag_grad2 > 0.0) {
            Slope = drdx / sqrt(mag_grad2);
            slope2_Ratio = (Slope * Slope) * I_slope_max2;
          } else {
            Slope = 0.0;
            slope2_Ratio = 1.0e20;
          }
        }
        // :939-943 with int_slope = 0
        Slope = (1.0 - 0.0) * Slope + 0.0 * ((eR_K - eL_K) * IdL);
        slope2_Ratio = (1.0 - 0.0) * slope2_Ratio;
        Sfn_unlim = -(KH * dLf) * Slope;      // :956
        if (Sfn_unlim > 0.0) {                 // :959-975
          if (eL_K < e_botR) Sfn_unlim = 0.0;
          else if (e_botR > eL_Kp1) Sfn_unlim = Sfn_unlim * ((eL_K - e_botR) / ((eL_K - eL_Kp1) + dz_neglect));
        } else {
          if (eR_K < e_botL) Sfn_unlim = 0.0;
          else if (e_botL > eR_Kp1) Sfn_unlim = Sfn_unlim * ((eR_K - e_botL) / ((eR_K - eR_Kp1) + dz_neglect));
        }
      } else {      // :979-988
        double Slope;
        if (present_slope) Slope = A.slope[f2 + fpl * (K - 1)];
        else Slope = ((eL_K - eR_K) * IdL) * OBCmask;
        Sfn_unlim = ((KH * dLf) * Slope);
        dzN2 = A.g_prime[K - 1];      // GV%g_prime(K) :1095
      }
    } else {
      dzN2 = A.N2_floor * dz_neglect;      // :1098
      Sfn_unlim = 0.;
    }
    {
      const long w = f2 + fpl * (K - 1);
      A.W_sfn[w] = Sfn_unlim; A.W_s2r[w] = slope2_Ratio; A.W_n2[w] = dzN2;
      if (find_work) { A.W_dkde[w] = drdkDe; A.W_dik[w] = drdi_k; }
    }
    hL_k = hL_km1; hR_k = hR_km1; eL_Kp1 = eL_K; eR_Kp1 = eR_K;
    TL_k = TL_km1; TR_k = TR_km1; SL_k = SL_km1; SR_k = SR_km1;
  }
  // ---- the streamfunction of Ferrari et al. (2010) :1105-1124, streamfn_solver :1673-1707.  c2_dz of a layer is formed where it is used; the
  // solver's c1 takes the place of dzN2 in its plane (dzN2(K) is read at step K only, c1(K-1) is written there)
  if (OBCmask > 0.) {
    const double cg = 0.5 * (A.cg1[cL] + A.cg1[cR]);
    auto c2_dz = [&](int k) -> double {      // k one-based
      const double dzL = g.H_to_Z * A.h[cL + hpl * (k - 1)], dzR = g.H_to_Z * A.h[cR + hpl * (k - 1)];
      const double dz_harm = max2(dz_neglect, 2. * dzL * dzR / ((dzL + dzR) + dz_neglect));
      return A.FGNV_scale * (cg * cg) / dz_harm;
    };
    const double scale = (1. + A.FGNV_scale);
    double c2_km1 = c2_dz(1), c2_k = c2_dz(2);
    double hN2 = A.W_n2[f2 + fpl];      // dzN2(2)
    double b_denom = hN2 + c2_km1;
    double beta = 1.0 / (b_denom + c2_k);
    double d1 = beta * b_denom;
    double sfn_prev = (beta * hN2) * (scale * A.W_sfn[f2 + fpl]);
    A.W_sfn[f2 + fpl] = sfn_prev;
    for (int K = 3; K <= nz; K++) {
      c2_km1 = c2_k; c2_k = c2_dz(K);
      const long w = f2 + fpl * (K - 1);
      hN2 = A.W_n2[w];
      A.W_n2[w - fpl] = beta * c2_km1;      // c1(K-1)
      b_denom = hN2 + d1 * c2_km1;
      beta = 1.0 / (b_denom + c2_k);
      d1 = beta * b_denom;
      sfn_prev = beta * (hN2 * (scale * A.W_sfn[w]) + c2_km1 * sfn_prev);
      A.W_sfn[w] = sfn_prev;
    }
    A.W_n2[f2 + fpl * (nz - 1)] = beta * c2_k;      // c1(nz)
    double sfn_below = 0.;                          // sfn(nz+1)
    for (int K = nz; K >= 2; K--) {
      const long w = f2 + fpl * (K - 1);
      sfn_below = A.W_sfn[w] + A.W_n2[w] * sfn_below;
      A.W_sfn[w] = sfn_below;
    }
  } else {
    for (int K = 2; K <= nz; K++) A.W_sfn[f2 + fpl * (K - 1)] = 0.;
  }
  // ---- the second sweep :1126-1209
  hL_k = A.h[cL + hpl * (nz - 1)]; hR_k = A.h[cR + hpl * (nz - 1)];
  eL_Kp1 = e_botL; eR_Kp1 = e_botR;
  for (int K = nz; K >= 2; K--) {
    const int k = K;
    const long oK = hpl * (K - 1), okm1 = hpl * (k - 2);
    const double hL_km1 = A.h[cL + okm1], hR_km1 = A.h[cR + okm1];
    const double eL_K = A.e[cL + oK], eR_K = A.e[cR + oK];
    const long w = f2 + fpl * (K - 1);
    const double Sfn_unlim = A.W_sfn[w], slope2_Ratio = A.W_s2r[w];
    const double drdkDe_K = find_work ? A.W_dkde[w] : 0.0, drdi_k = find_work ? A.W_dik[w] : 0.0;
    // ---- the transport of layer k :1148-1209
    const double availL = max2(I4dt * aL * (hL_k - g.Angstrom_H), 0.0), availR = max2(I4dt * aR * (hR_k - g.Angstrom_H), 0.0);
    const double rsumL_Kp1 = A.rsum[cL + hpl * K], rsumR_Kp1 = A.rsum[cR + hpl * K];
    double fracL = 0.0, fracR = 0.0;      // h_frac(k), k >= 2
    if (availL > 0.0) fracL = availL / rsumL_Kp1;
    if (availR > 0.0) fracR = availR / rsumR_Kp1;
    const double Z_to_H = g.Z_to_H;
    double hDk;
    if (k > nk_linear) {
      double Sfn_est;
      if (use_EOS) {
        double Sfn_safe;
        if (htot <= 0.0) Sfn_safe = htot * (1.0 - fracL);
        else Sfn_safe = htot * (1.0 - fracR);
        Sfn_est = (Z_to_H * Sfn_unlim + slope2_Ratio * Sfn_safe) / (1.0 + slope2_Ratio);
      } else {
        Sfn_est = Z_to_H * Sfn_unlim;
      }
      const double Sfn_in_H = min2(max2(Sfn_est, -A.rsum[cL + oK]), A.rsum[cR + oK]);
      hDk = max2(min2((Sfn_in_H - htot), availL), -availR);
    } else {
      if (htot <= 0.0) hDk = -htot * fracL;
      else hDk = -htot * fracR;
    }
    A.hD[f2 + fpl * (k - 1)] = hDk;
    htot = htot + hDk;
    if (find_work)      // :1196-1209
      Work = Work + G_scale * (htot * drdkDe_K - (hDk * drdi_k) * 0.25 * ((eL_K + eL_Kp1) + (eR_K + eR_Kp1)));
    hL_k = hL_km1; hR_k = hR_km1; eL_Kp1 = eL_K; eR_Kp1 = eR_K;
  }
  // ---- the top layer :1517-1590 (after the sweep the handed-down values are those of layer 1 and interface 2)
  const double hD1 = -htot;
  A.hD[f2] = hD1;
  if (find_work && use_EOS) {
    TL_k = A.T[cL]; TR_k = A.T[cR]; SL_k = A.S[cL]; SR_k = A.S[cR];
    const double pres_u = 0.5 * (A.pres[cL] + A.pres[cR]);
    const double T_u = 0.5 * (TL_k + TR_k), S_u = 0.5 * (SL_k + SR_k);
    double drho_dT, drho_dS;
    eos_density_derivs(A.E, T_u, S_u, pres_u, drho_dT, drho_dS);
    const double drdiB1 = drho_dT * (TR_k - TL_k) + drho_dS * (SR_k - SL_k);
    const double eL_1 = A.e[cL], eR_1 = A.e[cR];
    const double w = G_scale * ((hD1 * drdiB1) * 0.25 * ((eL_1 + eL_Kp1) + (eR_1 + eR_Kp1)));
    if (DIR == 0 && A.use_GM_work_bug) Work = Work + w;
    else Work = Work - w;
  }
  if (A.Work) A.Work[f2] = Work;
}

struct UpdArgs {
  m6::GridDev g;
  const double *uhD, *vhD, *Work_u, *Work_v;
  double *h, *uhtr, *vhtr, *uhGM, *vhGM, *GM_src;
  double dt;
};

// :607-620 and :1552-1560: thread (i, j, k) owns the u face I = i, the v face J = j and the cell of the same indices
__global__ __launch_bounds__(256) void td_update_kernel(UpdArgs A) {
  const m6::GridDev &g = A.g;
  const int i = g.isc - 1 + blockIdx.x * 64 + threadIdx.x, j = g.jsc - 1 + blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
  if (i > g.iec || j > g.jec) return;
  const long hpl = (long)g.nih * g.njh, upl = (long)(g.nih + 1) * g.njh, vpl = (long)g.nih * (g.njh + 1);
  if (j >= g.jsc) {
    const long n = g.u2(i, j) + upl * k;
    A.uhtr[n] = A.uhtr[n] + A.uhD[n] * A.dt;
    if (A.uhGM) A.uhGM[n] = A.uhD[n];
  }
  if (i >= g.isc) {
    const long n = g.v2(i, j) + vpl * k;
    A.vhtr[n] = A.vhtr[n] + A.vhD[n] * A.dt;
    if (A.vhGM) A.vhGM[n] = A.vhD[n];
  }
  if (i >= g.isc && j >= g.jsc) {
    const long c = g.h2(i, j);
    double hn = A.h[c + hpl * k] - A.dt * g.IareaT[c] *
                ((A.uhD[g.u2(i, j) + upl * k] - A.uhD[g.u2(i - 1, j) + upl * k]) + (A.vhD[g.v2(i, j) + vpl * k] - A.vhD[g.v2(i, j - 1) + vpl * k]));
    if (hn < g.Angstrom_H) hn = g.Angstrom_H;
    A.h[c + hpl * k] = hn;
    if (k == 0 && A.GM_src) {
      const double Work_h = 0.5 * g.IareaT[c] * ((A.Work_u[g.u2(i - 1, j)] + A.Work_u[g.u2(i, j)]) + (A.Work_v[g.v2(i, j - 1)] + A.Work_v[g.v2(i, j)]));
      A.GM_src[c] = 0. + Work_h;      // (zeroed at :197-199, then incremented at :1560)
    }
  }
}

int check_cs(const mom6hip_thickness_diffuse_cs_t *cs) {
  static const char *names[10] = {"(unused)", "DETANGLE_INTERFACES", "KH_ETA_CONST / KH_ETA_VEL_SCALE", "USE_STANLEY_GM",
                                  "MEKE_GEOMETRIC", "MEKE_GM_SRC_ALT", "READ_KHTH", "KHTH_USE_EBT_STRUCT / QG Leith GM",
                                  "USE_KH_IN_MEKE", "non-Boussinesq mode / tv%p_surf / SKEB"};
  M6_REQUIRE(cs->initialized, "MOM_thickness_diffuse: Module must be initialized before it is used.");
  for (int n = 0; n < 10; n++) M6_REQUIRE(!cs->unsupported[n], "thickness_diffuse: %s is not provided by libmom6hip", names[n]);
  M6_REQUIRE(!cs->use_FGNV_streamfn || cs->cg1, "thickness_diffuse: cg1 must be associated when using FGNV streamfunction.");      // :860
  return 0;
}

}  // namespace

extern "C" int mom6hip_thickness_diffuse(mom6hip_ctx_t *ctx, const mom6hip_thickness_diffuse_cs_t *cs, double *h, double *uhtr, double *vhtr,
                                         const double *T, const double *S, const mom6hip_eos_t *eos, double dt, double *uhGM, double *vhGM,
                                         int32_t memspace) {
  M6_REQUIRE(ctx != nullptr, "MOM_thickness_diffuse: Module must be initialized before it is used.");
  M6_REQUIRE(cs && h && uhtr && vhtr, "thickness_diffuse: null argument");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "thickness_diffuse: bad memspace");
  if (check_cs(cs)) return 1;
  const bool use_Visbeck = cs->use_variable_mixing && (cs->KHTH_Slope_Cff > 0.) && cs->L2u && cs->L2v && cs->SN_u && cs->SN_v;
  if (!cs->thickness_diffuse || !(cs->Khth > 0.0 || cs->use_variable_mixing)) return 0;      // :192-194
  M6_REQUIRE(dt > 0.0, "thickness_diffuse: dt must be positive");
  M6_REQUIRE(cs->max_Khth_CFL > 0.0, "thickness_diffuse: KHTH_MAX_CFL <= 0 is not provided by libmom6hip");
  M6_REQUIRE(!eos || (T && S), "thickness_diffuse: an equation of state needs tv%%T and tv%%S");
  M6_REQUIRE(!cs->MEKE_GM_src || eos || cs->Rlay, "thickness_diffuse: the work without an equation of state needs GV%%Rlay");
  M6_REQUIRE((cs->slope_x != nullptr) == (cs->slope_y != nullptr), "thickness_diffuse: slope_x and slope_y come together");
  M6_REQUIRE((cs->Depth_fn_u != nullptr) == (cs->Depth_fn_v != nullptr), "thickness_diffuse: Depth_fn_u and Depth_fn_v come together");
  const m6::GridDev g = ctx->g;
  M6_REQUIRE(g.isc - g.isd >= 1 && g.jsc - g.jsd >= 1, "thickness_diffuse: the halo must be at least 1 point wide");
  M6_REQUIRE(g.areaT && g.IareaT && g.bathyT && g.IdxCu && g.IdyCu && g.IdxCv && g.IdyCv && g.dy_Cu && g.dx_Cv && g.mask2dCu && g.mask2dCv,
             "thickness_diffuse: a required grid metric is missing");
  const int nz = g.nk;
  const size_t nH = (size_t)g.nih * g.njh, nU = (size_t)(g.nih + 1) * g.njh, nV = (size_t)g.nih * (g.njh + 1);
  const size_t bH2 = 8 * nH, bU2 = 8 * nU, bV2 = 8 * nV, bH = bH2 * nz, bU = bU2 * nz, bV = bV2 * nz;
  hipStream_t s = ctx->stream;
  m6::Stager st(ctx, memspace);
  TDArgs A;
  A.g = g;
  A.E.form = eos ? eos->form : MOM6HIP_EOS_LINEAR; A.E.Rho_T0_S0 = eos ? eos->Rho_T0_S0 : 0.0;
  A.E.dRho_dT = eos ? eos->dRho_dT : 0.0; A.E.dRho_dS = eos ? eos->dRho_dS : 0.0;
  A.use_EOS = eos != nullptr; A.find_work = cs->MEKE_GM_src != nullptr; A.nkml = cs->nkml; A.use_GM_work_bug = cs->use_GM_work_bug;
  A.use_Visbeck = use_Visbeck;
  A.Khth = cs->Khth; A.Khth_Min = cs->Khth_Min; A.Khth_Max = cs->Khth_Max; A.max_Khth_CFL = cs->max_Khth_CFL; A.slope_max = cs->slope_max;
  A.kappa_dt = cs->kappa_smooth * dt; A.KHTH_Slope_Cff = cs->KHTH_Slope_Cff; A.KhTh_fac = cs->KhTh_fac; A.dt = dt;
  double *d_h = st.inout(h, bH), *d_uhtr = st.inout(uhtr, bU), *d_vhtr = st.inout(vhtr, bV);
  A.h = d_h; A.T_in = st.in(T, bH); A.S_in = st.in(S, bH);
  double *d_uhGM = st.inout(uhGM, bU), *d_vhGM = st.inout(vhGM, bV);      // (only the compute ranges are written)
  A.MEKE_Kh = st.in(cs->MEKE_Kh, bH2);
  const double *L2u = st.in(cs->L2u, bU2), *L2v = st.in(cs->L2v, bV2), *SNu = st.in(cs->SN_u, bU2), *SNv = st.in(cs->SN_v, bV2);
  const double *Ru = st.in(cs->Res_fn_u, bU2), *Rv = st.in(cs->Res_fn_v, bV2);
  const double *Du = st.in(cs->Depth_fn_u, bU2), *Dv = st.in(cs->Depth_fn_v, bV2);
  const double *sx = st.in(cs->slope_x, bU2 * (nz + 1)), *sy = st.in(cs->slope_y, bV2 * (nz + 1));
  double *d_src = cs->MEKE_GM_src ? st.inout(cs->MEKE_GM_src, bH2) : nullptr;
  A.e = (double *)st.scratch(bH2 * (nz + 1)); A.pres = (double *)st.scratch(bH2 * (nz + 1)); A.rsum = (double *)st.scratch(bH2 * (nz + 1));
  A.T = eos ? (double *)st.scratch(bH) : nullptr; A.S = eos ? (double *)st.scratch(bH) : nullptr; A.c1 = eos ? (double *)st.scratch(bH) : nullptr;
  double *uhD = (double *)st.scratch(bU), *vhD = (double *)st.scratch(bV);
  double *Work_u = A.find_work ? (double *)st.scratch(bU2) : nullptr, *Work_v = A.find_work ? (double *)st.scratch(bV2) : nullptr;
  A.use_FGNV = cs->use_FGNV_streamfn; A.FGNV_scale = cs->FGNV_scale; A.N2_floor = cs->N2_floor;
  A.cg1 = nullptr; A.g_prime = nullptr; A.W_sfn = A.W_s2r = A.W_n2 = A.W_dkde = A.W_dik = nullptr;
  if (A.use_FGNV) {
    M6_REQUIRE(eos || cs->g_prime, "thickness_diffuse: the FGNV streamfunction without an equation of state needs GV%%g_prime");
    A.cg1 = st.in(cs->cg1, bH2);
    const size_t bW = (bU2 > bV2 ? bU2 : bV2) * (nz + 1);      // (the planes serve the zonal launch, then the meridional one)
    A.W_sfn = (double *)st.scratch(bW); A.W_s2r = (double *)st.scratch(bW); A.W_n2 = (double *)st.scratch(bW);
    if (A.find_work) { A.W_dkde = (double *)st.scratch(bW); A.W_dik = (double *)st.scratch(bW); }
  }
  M6_REQUIRE(!st.failed(), "thickness_diffuse: staging failed");
  A.Rlay = (!eos && A.find_work) ? ctx->tables[m6::TABLE_TD_RLAY].get(cs->Rlay, (size_t)nz) : nullptr;
  M6_REQUIRE(eos || !A.find_work || A.Rlay, "thickness_diffuse: out of device memory");
  if (A.use_FGNV && !eos) {
    A.g_prime = ctx->tables[m6::TABLE_TD_GPRIME].get(cs->g_prime, (size_t)nz + 1);
    M6_REQUIRE(A.g_prime, "thickness_diffuse: out of device memory");
  }
  const int ni = g.iec - g.isc + 1, nj = g.jec - g.jsc + 1;
  hipLaunchKernelGGL(td_column_kernel, dim3((ni + 2 + 63) / 64, nj + 2), dim3(64), 0, s, A);
  A.L2 = L2u; A.SN = SNu; A.Res_fn = Ru; A.Depth_fn = Du; A.slope = sx; A.hD = uhD; A.Work = Work_u;
  if (A.use_FGNV) hipLaunchKernelGGL(td_face_fgnv_kernel<0>, dim3((ni + 1 + 63) / 64, nj), dim3(64), 0, s, A);
  else hipLaunchKernelGGL(td_face_kernel<0>, dim3((ni + 1 + 63) / 64, nj), dim3(64), 0, s, A);
  A.L2 = L2v; A.SN = SNv; A.Res_fn = Rv; A.Depth_fn = Dv; A.slope = sy; A.hD = vhD; A.Work = Work_v;
  if (A.use_FGNV) hipLaunchKernelGGL(td_face_fgnv_kernel<1>, dim3((ni + 63) / 64, nj + 1), dim3(64), 0, s, A);
  else hipLaunchKernelGGL(td_face_kernel<1>, dim3((ni + 63) / 64, nj + 1), dim3(64), 0, s, A);
  UpdArgs Up;
  Up.g = g; Up.uhD = uhD; Up.vhD = vhD; Up.Work_u = Work_u; Up.Work_v = Work_v; Up.h = d_h; Up.uhtr = d_uhtr; Up.vhtr = d_vhtr;
  Up.uhGM = d_uhGM; Up.vhGM = d_vhGM; Up.GM_src = d_src; Up.dt = dt;
  hipLaunchKernelGGL(td_update_kernel, dim3((ni + 1 + 63) / 64, (nj + 1 + 3) / 4, nz), dim3(64, 4), 0, s, Up);
  M6_HIP(hipGetLastError());
  return st.finish();
}
