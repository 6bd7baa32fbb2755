// pressure_force.hip -- PressureForce_FV_Bouss (src/core/MOM_PressureForce_FV.F90:462-919) with the analytic
// finite-volume pressure-gradient integrals of int_density_dz_generic_plm
// (src/core/MOM_density_integrals.F90:369-769), PLM edge values of T and S (ALE_PLM_edge_values,
// src/ALE/MOM_ALE.F90:1520-1579), Set_pbce_Bouss (src/core/MOM_PressureForce_Montgomery.F90:649-748) and the
// Wright / linear equations of state (src/equation_of_state/MOM_EOS_Wright.F90:80-206, MOM_EOS_linear.F90).
//
// This operator is fp64-VALU bound, not HBM bound: 35 equation-of-state evaluations per cell-layer (5 for
// the vertical Boole quadrature, 3x5 on each of the two faces), each ~45 flops and one IEEE division,
// against ~64 B of algorithmic traffic.  Two kernels, both one lane per column with the k loop inside the
// lane (the pressure anomaly pa and the interface heights e are running sums in k) and lanes along i:
//   pgf_column_kernel  per h-point column: e (bottom-up), then top-down the PLM edge values of T and S with
//                      a rolling k window, the vertical integrals dpa / intz_dpa (5 EOS evaluations), pbce
//                      and eta.  Writes e, T_t, T_b, S_t, S_b, dpa, intz_dpa.
//   pgf_face_kernel    per (I,j)/(i,J) face pair: the 15+15 EOS evaluations of intx_dpa / inty_dpa and the
//                      PFu / PFv formulas (:794-811), carrying pa, intx_pa, inty_pa down the column.
// Splitting at the column kernel keeps the EOS count at the reference's 35 per cell (a single fused
// kernel would have to recompute the neighbours' vertical integrals: 45).
#include <cmath>

#include "common.hpp"
#include "eos.hpp"

namespace {

using m6::max2;
using m6::min2;
using namespace m6::eos;
__device__ __forceinline__ double max3(double a, double b, double c) { return max2(max2(a, b), c); }
__device__ __forceinline__ double min3(double a, double b, double c) { return min2(min2(a, b), c); }
__device__ __forceinline__ double fsign(double a, double b) { return copysign(fabs(a), b); }

// ---- PLM_functions.F90 ---------------------------------------------------------------------------
__device__ double plm_slope_wa(double h_l, double h_c, double h_r, double h_neglect, double u_l, double u_c, double u_r) {
  const double sigma_r = u_r - u_c;
  const double sigma_l = u_c - u_l;
  const double sigma_c = 2.0 * (u_r - u_l) * (h_c / (h_l + 2.0 * h_c + h_r + h_neglect));
  const double u_min = min3(u_l, u_c, u_r);
  const double u_max = max3(u_l, u_c, u_r);
  double slope = 0.0;
  if ((sigma_l * sigma_r) > 0.0) slope = fsign(min2(fabs(sigma_c), 2. * min2(u_c - u_min, u_max - u_c)), sigma_c);
  if (u_c - 0.5 * fabs(slope) < u_min || u_c + 0.5 * fabs(slope) > u_max) slope = slope * (1. - 2.220446049250313e-16);
  if (fabs(slope) < 1.E-140) slope = 0.;
  return slope;
}
__device__ double plm_monotonized_slope(double u_l, double u_c, double u_r, double s_l, double s_c, double s_r) {
  const double almost_two = 2. * (1. - 2.220446049250313e-16);
  const double e_r = u_l + 0.5 * s_l;
  const double e_l = u_r - 0.5 * s_r;
  double slp = fabs(s_c);
  double edge = u_c - 0.5 * s_c;
  if ((edge - e_r) * (u_c - edge) < 0.) { edge = 0.5 * (edge + e_r); slp = min2(slp, fabs(edge - u_c) * almost_two); }
  edge = u_c + 0.5 * s_c;
  if ((edge - u_c) * (e_l - edge) < 0.) { edge = 0.5 * (edge + e_l); slp = min2(slp, fabs(edge - u_c) * almost_two); }
  return fsign(slp, s_c);
}
__device__ double plm_extrapolate_slope(double h_l, double h_c, double h_neglect, double u_l, double u_c) {
  const double hl = h_l + h_neglect, hc = h_c + h_neglect;
  const double left_edge = (u_l * hc + u_c * hl) / (hl + hc);
  return 2.0 * (u_c - left_edge);
}

struct PgfArgs {
  m6::GridDev g;
  EosDev eos;
  const double *h, *T, *S, *p_atm;
  double *e, *T_t, *T_b, *S_t, *S_b, *dpa, *intz_dpa;     // scratch (e has nk+1 planes)
  double *PFu, *PFv, *pbce, *eta;
  double rho_ref, Z_ref, GFS_scale;
  int boundary_extrap, massw;
  double *za0;        // non-Boussinesq: the geopotential anomaly at the sea surface (2-D scratch)
  double H_to_RZ;     // non-Boussinesq: GV%H_to_RZ
  // the int_density_dz / layered branches (pgf_pcm_*_kernel)
  const double *Rlay, *g_prime;     // device copies of GV%Rlay(1:nk), GV%g_prime(1:nk+1) (or null)
  int nkmb;                         // GV%nk_rho_varies
  double P_Ref;                     // tv%P_Ref
  // the RK2 step's u_bc_accel = (CAu + PFu) + diffu formed by pgf_face_kernel (mom6hip_ctx::BcAccelFuse; null: not asked for)
  const double *bc_CAu, *bc_CAv, *bc_diffu, *bc_diffv;
  double *bc_u, *bc_v;
  int bc_inviscid;
};

// ---- column kernel -------------------------------------------------------------------------------
#ifndef PGF_COL_OCC
#define PGF_COL_OCC 4      // waves per SIMD the register allocation aims at (tools/build_variant.sh for experiments)
#endif
#ifndef PGF_FACE_OCC
#define PGF_FACE_OCC 4      // 133 -> 126 VGPRs (20 B of scratch): 12.9 -> 12.3 ms for the pair (profiles/r03_occupancy_experiments.txt)
#endif
__global__ __launch_bounds__(64, PGF_COL_OCC) void pgf_column_kernel(PgfArgs p) {
  const m6::GridDev &g = p.g;
  const int i = g.isc - 1 + blockIdx.x * 64 + threadIdx.x;
  const int j = g.jsc - 1 + blockIdx.y;
  if (i > g.iec + 1) return;
  const int nz = g.nk;
  const long o2 = g.h2(i, j), pl = (long)g.nih * g.njh;
  const double h_neglect = g.H_subroundoff;
  // :572 / :646-648, bottom-up
  double ek = -g.bathyT[o2];
  p.e[o2 + pl * nz] = ek;
  for (int k = nz - 1; k >= 0; k--) {
    ek = ek + p.h[o2 + pl * k] * g.H_to_Z;
    p.e[o2 + pl * k] = ek;
  }
  const double e_top = ek, e_bot = -g.bathyT[o2];
  if (p.eta) p.eta[o2] = e_top * g.Z_to_H;      // :842

  const double G_e = g.g_Earth, rho_0 = p.rho_ref, rho_ref = p.rho_ref;
  const double GxRho = G_e * rho_0;
  const double C1_90 = 1.0 / 90.0;
  const double Rho0xG = p.rho_ref * g.g_Earth, G_Rho0 = g.g_Earth / g.Rho0;
  const double Ihtot = g.H_to_Z / ((e_top - e_bot) + g.dZ_subroundoff);

  // rolling window over k (0-based level kk): values at kk-1 (m), kk (c), kk+1 (p), kk+2 (q)
  auto ld = [&](const double *a, int kk) -> double { return (kk >= 0 && kk < nz) ? a[o2 + pl * kk] : 0.0; };
  double h_m = 0., h_c = ld(p.h, 0), h_p = ld(p.h, 1), h_q = ld(p.h, 2);
  double T_m = 0., T_c = ld(p.T, 0), T_p = ld(p.T, 1), T_q = ld(p.T, 2);
  double S_m = 0., S_c = ld(p.S, 0), S_p = ld(p.S, 1), S_q = ld(p.S, 2);
  // slp at kk-1, kk, kk+1 (Fortran slp(1) = slp(nz) = 0)
  double sT_m = 0., sT_c = 0., sT_p = 0., sS_m = 0., sS_c = 0., sS_p = 0.;
  if (nz >= 3) {   // slp of level 1 (Fortran k=2)
    sT_p = plm_slope_wa(h_c, h_p, h_q, h_neglect, T_c, T_p, T_q);
    sS_p = plm_slope_wa(h_c, h_p, h_q, h_neglect, S_c, S_p, S_q);
  }
  double e_K = e_top, pbce_prev = 0.0;
  for (int kk = 0; kk < nz; kk++) {
    const long o3 = o2 + pl * kk;
    // ---- ALE_PLM_edge_values :1549-1576 ----
    double Tt, Tb, St, Sb;
    if (kk >= 1 && kk <= nz - 2) {
      const double mT = plm_monotonized_slope(T_m, T_c, T_p, sT_m, sT_c, sT_p);
      Tt = T_c - 0.5 * mT; Tb = T_c + 0.5 * mT;
      const double mS = plm_monotonized_slope(S_m, S_c, S_p, sS_m, sS_c, sS_p);
      St = S_c - 0.5 * mS; Sb = S_c + 0.5 * mS;
    } else if (p.boundary_extrap) {
      if (kk == 0) {
        const double mT = -plm_extrapolate_slope(h_p, h_c, h_neglect, T_p, T_c);
        Tt = T_c - 0.5 * mT; Tb = T_c + 0.5 * mT;
        const double mS = -plm_extrapolate_slope(h_p, h_c, h_neglect, S_p, S_c);
        St = S_c - 0.5 * mS; Sb = S_c + 0.5 * mS;
      } else {
        const double mT = plm_extrapolate_slope(h_m, h_c, h_neglect, T_m, T_c);
        Tt = T_c - 0.5 * mT; Tb = T_c + 0.5 * mT;
        const double mS = plm_extrapolate_slope(h_m, h_c, h_neglect, S_m, S_c);
        St = S_c - 0.5 * mS; Sb = S_c + 0.5 * mS;
      }
    } else {
      Tt = T_c; Tb = T_c; St = S_c; Sb = S_c;
    }
    p.T_t[o3] = Tt; p.T_b[o3] = Tb; p.S_t[o3] = St; p.S_b[o3] = Sb;

    // ---- vertical integrals, MOM_density_integrals.F90:519-554 ----
    const double e_Kp1 = p.e[o3 + pl];
    const double dz = e_K - e_Kp1;
    double r5[5];
#pragma unroll
    for (int n = 1; n <= 5; n++) {
      const double wt_t = 0.25 * (double)(5 - n), wt_b = 1.0 - wt_t;
      const double p5 = -GxRho * ((e_K - p.Z_ref) - 0.25 * (double)(n - 1) * dz);
      const double S5 = wt_t * St + wt_b * Sb;
      const double T5 = wt_t * Tt + wt_b * Tb;
      r5[n - 1] = eos_density_anomaly(p.eos, T5, S5, p5, rho_ref);
    }
    const double rho_anom = C1_90 * (7.0 * (r5[0] + r5[4]) + 32.0 * (r5[1] + r5[3]) + 12.0 * r5[2]);
    p.dpa[o3] = G_e * dz * rho_anom;
    const double iz = 0.5 * G_e * (dz * dz) * (rho_anom - C1_90 * (16.0 * (r5[3] - r5[1]) + 7.0 * (r5[4] - r5[0])));
    p.intz_dpa[o3] = iz * g.Z_to_H;      // MOM_PressureForce_FV.F90:772

    // ---- Set_pbce_Bouss :702-729 ----
    if (p.pbce) {
      const double press = -Rho0xG * (e_K - p.Z_ref);
      double pb;
      if (kk == 0) {
        const double rho_in_situ = eos_density(p.eos, T_c, S_c, press);
        pb = G_Rho0 * (p.GFS_scale * rho_in_situ) * g.H_to_Z;
      } else {
        const double T_int = 0.5 * (T_m + T_c), S_int = 0.5 * (S_m + S_c);
        double dR_dT, dR_dS;
        eos_density_derivs(p.eos, T_int, S_int, press, dR_dT, dR_dS);
        pb = pbce_prev + G_Rho0 * ((e_K - e_bot) * Ihtot) * (dR_dT * (T_c - T_m) + dR_dS * (S_c - S_m));
      }
      p.pbce[o3] = pb;
      pbce_prev = pb;
    }

    // ---- advance the window ----
    e_K = e_Kp1;
    h_m = h_c; h_c = h_p; h_p = h_q; h_q = ld(p.h, kk + 3);
    T_m = T_c; T_c = T_p; T_p = T_q; T_q = ld(p.T, kk + 3);
    S_m = S_c; S_c = S_p; S_p = S_q; S_q = ld(p.S, kk + 3);
    sT_m = sT_c; sT_c = sT_p; sS_m = sS_c; sS_c = sS_p;
    // slope of the new level kk+2 (needs levels kk+1, kk+2, kk+3); zero at the bottom level and beyond
    if (kk + 2 <= nz - 2) {
      sT_p = plm_slope_wa(h_c, h_p, h_q, h_neglect, T_c, T_p, T_q);
      sS_p = plm_slope_wa(h_c, h_p, h_q, h_neglect, S_c, S_p, S_q);
    } else {
      sT_p = 0.; sS_p = 0.;
    }
  }
}

// ---- face kernel ---------------------------------------------------------------------------------
// 15 EOS evaluations across one face between the columns at offsets oL (left/south) and oR (right/north)
__device__ __forceinline__ double face_integral(const PgfArgs &p, long oL3, long oR3, long oL2, long oR2, double eL_K,
                                                double eL_Kp1, double eR_K, double eR_Kp1) {
  const m6::GridDev &g = p.g;
  const double G_e = g.g_Earth, GxRho = G_e * p.rho_ref, rho_ref = p.rho_ref;
  const double C1_90 = 1.0 / 90.0;
  double Ttl, Tbl, Ttr, Tbr, Stl, Sbl, Str, Sbr;
  const double TtL = p.T_t[oL3], TtR = p.T_t[oR3], TbL = p.T_b[oL3], TbR = p.T_b[oR3];
  const double StL = p.S_t[oL3], StR = p.S_t[oR3], SbL = p.S_b[oL3], SbR = p.S_b[oR3];
  double hWght = (p.massw ? 1. : 0.) * max3(0., -g.bathyT[oL2] - eR_K, -g.bathyT[oR2] - eL_K);
  if (hWght > 0.) {
    const double hL = (eL_K - eL_Kp1) + g.dZ_subroundoff;
    const double hR = (eR_K - eR_Kp1) + g.dZ_subroundoff;
    const double rr = (hL - hR) / (hL + hR);
    hWght = hWght * (rr * rr);
    const double iDenom = 1. / (hWght * (hR + hL) + hL * hR);
    Ttl = ((hWght * hR) * TtR + (hWght * hL + hR * hL) * TtL) * iDenom;
    Ttr = ((hWght * hL) * TtL + (hWght * hR + hR * hL) * TtR) * iDenom;
    Tbl = ((hWght * hR) * TbR + (hWght * hL + hR * hL) * TbL) * iDenom;
    Tbr = ((hWght * hL) * TbL + (hWght * hR + hR * hL) * TbR) * iDenom;
    Stl = ((hWght * hR) * StR + (hWght * hL + hR * hL) * StL) * iDenom;
    Str = ((hWght * hL) * StL + (hWght * hR + hR * hL) * StR) * iDenom;
    Sbl = ((hWght * hR) * SbR + (hWght * hL + hR * hL) * SbL) * iDenom;
    Sbr = ((hWght * hL) * SbL + (hWght * hR + hR * hL) * SbR) * iDenom;
  } else {
    Ttl = TtL; Tbl = TbL; Ttr = TtR; Tbr = TbR;
    Stl = StL; Sbl = SbL; Str = StR; Sbr = SbR;
  }
  double intz[5];
  intz[0] = p.dpa[oL3]; intz[4] = p.dpa[oR3];
#pragma unroll
  for (int m = 2; m <= 4; m++) {
    const double w_left = 0.25 * (double)(5 - m), w_right = 1.0 - w_left;
    const double dz_x = w_left * (eL_K - eL_Kp1) + w_right * (eR_K - eR_Kp1);
    const double T1 = w_left * Ttl + w_right * Ttr, T5 = w_left * Tbl + w_right * Tbr;
    const double S1 = w_left * Stl + w_right * Str, S5 = w_left * Sbl + w_right * Sbr;
    double pn = -GxRho * ((w_left * eL_K + w_right * eR_K) - p.Z_ref);
    double r[5];
#pragma unroll
    for (int n = 1; n <= 5; n++) {
      if (n > 1) pn = pn + GxRho * 0.25 * dz_x;
      double Tn, Sn;
      if (n == 1) { Tn = T1; Sn = S1; }
      else if (n == 5) { Tn = T5; Sn = S5; }
      else {
        const double wt_t = 0.25 * (double)(5 - n), wt_b = 1.0 - wt_t;
        Sn = wt_t * S1 + wt_b * S5;
        Tn = wt_t * T1 + wt_b * T5;
      }
      r[n - 1] = eos_density_anomaly(p.eos, Tn, Sn, pn, rho_ref);
    }
    intz[m - 1] = G_e * dz_x * (C1_90 * (7.0 * (r[0] + r[4]) + 32.0 * (r[1] + r[3]) + 12.0 * r[2]));
  }
  return C1_90 * (7.0 * (intz[0] + intz[4]) + 32.0 * (intz[1] + intz[3]) + 12.0 * intz[2]);
}

__global__ __launch_bounds__(64, PGF_FACE_OCC) void pgf_face_kernel(PgfArgs p) {
  const m6::GridDev &g = p.g;
  const int i = g.isc - 1 + blockIdx.x * 64 + threadIdx.x;      // I (x face) / i (y face)
  const int j = g.jsc - 1 + blockIdx.y;                         // j (x face) / J (y face)
  if (i > g.iec) return;
  const bool do_x = (j >= g.jsc), do_y = (i >= g.isc);          // j <= jec and i <= iec by the grid size
  if (!do_x && !do_y) return;
  const int nz = g.nk;
  const long pl = (long)g.nih * g.njh, plU = (long)(g.nih + 1) * g.njh, plV = (long)g.nih * (g.njh + 1);
  const long oc = g.h2(i, j), oe = oc + 1, on = oc + g.nih;
  const double h_neglect = g.H_subroundoff, I_Rho0 = 1.0 / g.Rho0;
  const double rg = p.rho_ref * g.g_Earth;
  auto pa0 = [&](long o2) -> double {
    double v = rg * (p.e[o2] - p.Z_ref);
    if (p.p_atm) v = v + p.p_atm[o2];
    return v;
  };
  double pa_c = pa0(oc), pa_e = do_x ? pa0(oe) : 0.0, pa_n = do_y ? pa0(on) : 0.0;
  double intx_pa = 0.5 * (pa_c + pa_e), inty_pa = 0.5 * (pa_c + pa_n);
  const double fx = do_x ? (2.0 * I_Rho0 * g.IdxCu[g.u2(i, j)]) : 0.0;
  const double fy = do_y ? (2.0 * I_Rho0 * g.IdyCv[g.v2(i, j)]) : 0.0;
  double ec_K = p.e[oc], ee_K = do_x ? p.e[oe] : 0.0, en_K = do_y ? p.e[on] : 0.0;
  for (int k = 0; k < nz; k++) {
    const long c3 = oc + pl * k;
    const double ec_Kp1 = p.e[c3 + pl];
    const double h_c = p.h[c3], dpa_c = p.dpa[c3], iz_c = p.intz_dpa[c3];
    if (do_x) {
      const long e3 = c3 + 1;
      const double ee_Kp1 = p.e[e3 + pl];
      const double h_e = p.h[e3];
      const double intx_dpa = face_integral(p, c3, e3, oc, oe, ec_K, ec_Kp1, ee_K, ee_Kp1);
      const double pfu = (((pa_c * h_c + iz_c) - (pa_e * h_e + p.intz_dpa[e3])) +
                          ((h_e - h_c) * intx_pa - (ee_Kp1 - ec_Kp1) * intx_dpa * g.Z_to_H)) *
                         (fx / ((h_c + h_e) + h_neglect));
      p.PFu[g.u2(i, j) + plU * k] = pfu;
      if (p.bc_u) {
        const long n = g.u2(i, j) + plU * k;
        double a = p.bc_CAu[n] + pfu;
        if (p.bc_inviscid) a = (a == 0.0) ? 0.0 : a; else a = a + p.bc_diffu[n];
        p.bc_u[n] = a;
      }
      intx_pa = intx_pa + intx_dpa;
      pa_e = pa_e + p.dpa[e3];
      ee_K = ee_Kp1;
    }
    if (do_y) {
      const long n3 = c3 + g.nih;
      const double en_Kp1 = p.e[n3 + pl];
      const double h_n = p.h[n3];
      const double inty_dpa = face_integral(p, c3, n3, oc, on, ec_K, ec_Kp1, en_K, en_Kp1);
      const double pfv = (((pa_c * h_c + iz_c) - (pa_n * h_n + p.intz_dpa[n3])) +
                          ((h_n - h_c) * inty_pa - (en_Kp1 - ec_Kp1) * inty_dpa * g.Z_to_H)) *
                         (fy / ((h_c + h_n) + h_neglect));
      p.PFv[g.v2(i, j) + plV * k] = pfv;
      if (p.bc_v) {
        const long n = g.v2(i, j) + plV * k;
        double a = p.bc_CAv[n] + pfv;
        if (p.bc_inviscid) a = (a == 0.0) ? 0.0 : a; else a = a + p.bc_diffv[n];
        p.bc_v[n] = a;
      }
      inty_pa = inty_pa + inty_dpa;
      pa_n = pa_n + p.dpa[n3];
      en_K = en_Kp1;
    }
    pa_c = pa_c + dpa_c;
    ec_K = ec_Kp1;
  }
}

// ---- ALE_PLM_edge_values (MOM_ALE.F90:1520-1579) of one 3-d scalar: the reconstruction above for a caller's own field ----
struct PlmEdgeArgs { m6::GridDev g; const double *h, *Q; double *Q_t, *Q_b; int bdry_extrap; };
__global__ __launch_bounds__(64) void plm_edge_kernel(PlmEdgeArgs p) {
  const m6::GridDev &g = p.g;
  const int i = g.isc - 1 + blockIdx.x * 64 + threadIdx.x;
  const int j = g.jsc - 1 + blockIdx.y;
  if (i > g.iec + 1) return;
  const int nz = g.nk;
  const long o2 = g.h2(i, j), pl = (long)g.nih * g.njh;
  const double h_neglect = g.H_subroundoff;
  auto ld = [&](const double *a, int kk) -> double { return (kk >= 0 && kk < nz) ? a[o2 + pl * kk] : 0.0; };
  double h_m = 0., h_c = ld(p.h, 0), h_p = ld(p.h, 1), h_q = ld(p.h, 2);
  double Q_m = 0., Q_c = ld(p.Q, 0), Q_p = ld(p.Q, 1), Q_q = ld(p.Q, 2);
  double s_m = 0., s_c = 0., s_p = 0.;
  if (nz >= 3) s_p = plm_slope_wa(h_c, h_p, h_q, h_neglect, Q_c, Q_p, Q_q);
  for (int kk = 0; kk < nz; kk++) {
    double Qt, Qb;
    if (kk >= 1 && kk <= nz - 2) {
      const double m = plm_monotonized_slope(Q_m, Q_c, Q_p, s_m, s_c, s_p);
      Qt = Q_c - 0.5 * m; Qb = Q_c + 0.5 * m;
    } else if (p.bdry_extrap) {
      const double m = kk == 0 ? -plm_extrapolate_slope(h_p, h_c, h_neglect, Q_p, Q_c) : plm_extrapolate_slope(h_m, h_c, h_neglect, Q_m, Q_c);
      Qt = Q_c - 0.5 * m; Qb = Q_c + 0.5 * m;
    } else {
      Qt = Q_c; Qb = Q_c;
    }
    p.Q_t[o2 + pl * kk] = Qt; p.Q_b[o2 + pl * kk] = Qb;
    h_m = h_c; h_c = h_p; h_p = h_q; h_q = ld(p.h, kk + 3);
    Q_m = Q_c; Q_c = Q_p; Q_p = Q_q; Q_q = ld(p.Q, kk + 3);
    s_m = s_c; s_c = s_p;
    s_p = (kk + 2 <= nz - 2) ? plm_slope_wa(h_c, h_p, h_q, h_neglect, Q_c, Q_p, Q_q) : 0.;
  }
}

// ======== the branches without a reconstruction (MOM_PressureForce_FV.F90:765-789) ========================================
// MODE 0: int_density_dz -> int_density_dz_linear (MOM_EOS_linear.F90:259-424); 1: int_density_dz_wright
// (MOM_EOS_Wright.F90:389-640); 2: no equation of state, the layered form with GV%Rlay (:775-789).
// The same split as the PLM pair: pgf_pcm_column_kernel forms e, dpa, intz_dpa (x Z_to_H), pbce and eta per column;
// pgf_pcm_face_kernel the face integrals and PFu / PFv.  With a bulk mixed layer (nkmb > 0) the T and S of a layer below it
// are the buffer layer's wherever GV%Rlay(k) is lighter than the buffer layer's coordinate density (tv_tmp, :650-670);
// the selection is re-derived per column in both kernels (one EOS evaluation) rather than stored as two 3-D arrays.
constexpr int PCM_LINEAR = 0, PCM_WRIGHT = 1, PCM_NOEOS = 2;

struct PcmCol {      // tv_tmp of one column: which layers take the buffer layer's T and S
  double T_bl, S_bl, Rho_cv_BL;
  int nkmb;
};
__device__ __forceinline__ PcmCol pcm_col(const PgfArgs &p, long o2, long pl) {
  PcmCol c; c.nkmb = p.nkmb; c.T_bl = 0.; c.S_bl = 0.; c.Rho_cv_BL = 0.;
  if (p.nkmb > 0) {
    c.T_bl = p.T[o2 + pl * (p.nkmb - 1)]; c.S_bl = p.S[o2 + pl * (p.nkmb - 1)];
    c.Rho_cv_BL = eos_density(p.eos, c.T_bl, c.S_bl, p.P_Ref);      // :662
  }
  return c;
}
__device__ __forceinline__ void pcm_TS(const PgfArgs &p, const PcmCol &c, long o3, int k, double &T, double &S) {      // k zero-based
  if (c.nkmb > 0 && k >= c.nkmb && p.Rlay[k] < c.Rho_cv_BL) { T = c.T_bl; S = c.S_bl; }
  else { T = p.T[o3]; S = p.S[o3]; }
}
struct WrightTerms { double al0, p0, lambda; };
__device__ __forceinline__ WrightTerms wright_terms(double T, double S) {      // MOM_EOS_Wright.F90:537-539
  WrightTerms w;
  w.al0 = (a0 + a1 * T) + a2 * S;
  w.p0 = (b0 + b4 * S) + T * (b1 + T * ((b2 + b3 * T)) + b5 * S);
  w.lambda = (c0 + c4 * S) + T * (c1 + T * ((c2 + c3 * T)) + c5 * S);
  return w;
}
// the layer integral of the Wright form at one point :541-557 / :587-595; returns dpa, and intz_dpa through iz when asked
template <bool WANT_IZ>
__device__ __forceinline__ double wright_layer(double al0, double p0, double lambda, double dz, double zsum_half, double GxRho, double g_Earth,
                                               double I_Rho, double rho_ref, double z0pres, double &iz) {
  const double C1_3 = 1.0 / 3.0, C1_7 = 1.0 / 7.0, C1_9 = 1.0 / 9.0;
  const double p_ave = -GxRho * (zsum_half - z0pres);
  const double I_al0 = 1.0 / al0;
  const double I_Lzz = 1.0 / (p0 + (lambda * I_al0) + p_ave);
  const double eps = 0.5 * GxRho * dz * I_Lzz, eps2 = eps * eps;
  if (WANT_IZ) {
    const double rho_anom = (p0 + p_ave) * (I_Lzz * I_al0) - rho_ref;
    const double rem = I_Rho * (lambda * (I_al0 * I_al0)) * eps2 * (C1_3 + eps2 * (0.2 + eps2 * (C1_7 + C1_9 * eps2)));
    iz = 1.0 * (0.5 * g_Earth * rho_anom * (dz * dz) - dz * (1.0 + eps) * rem);
    return 1.0 * (g_Earth * rho_anom * dz - 2.0 * eps * rem);
  }
  return 1.0 * (g_Earth * dz * ((p0 + p_ave) * (I_Lzz * I_al0) - rho_ref) -
                2.0 * eps * I_Rho * (lambda * (I_al0 * I_al0)) * eps2 * (C1_3 + eps2 * (0.2 + eps2 * (C1_7 + C1_9 * eps2))));
}

template <int MODE>
__global__ __launch_bounds__(64) void pgf_pcm_column_kernel(PgfArgs p) {
  const m6::GridDev &g = p.g;
  const int i = g.isc - 1 + blockIdx.x * 64 + threadIdx.x;
  const int j = g.jsc - 1 + blockIdx.y;
  if (i > g.iec + 1) return;
  const int nz = g.nk;
  const long o2 = g.h2(i, j), pl = (long)g.nih * g.njh;
  double ek = -g.bathyT[o2];
  p.e[o2 + pl * nz] = ek;
  for (int k = nz - 1; k >= 0; k--) {
    ek = ek + p.h[o2 + pl * k] * g.H_to_Z;
    p.e[o2 + pl * k] = ek;
  }
  const double e_top = ek, e_bot = -g.bathyT[o2];
  if (p.eta) p.eta[o2] = e_top * g.Z_to_H;
  const double G_e = g.g_Earth, rho_ref = p.rho_ref;
  const double GxRho = G_e * p.rho_ref, I_Rho = 1.0 / p.rho_ref;      // rho_0 = CS%Rho0 (:766)
  const double Rho0xG = p.rho_ref * g.g_Earth, G_Rho0 = g.g_Earth / g.Rho0;
  PcmCol col; col.nkmb = 0;
  if (MODE != PCM_NOEOS) col = pcm_col(p, o2, pl);
  double e_K = e_top, pbce_prev = 0.0, T_m = 0., S_m = 0.;
  const double Ihtot_eos = g.H_to_Z / ((e_top - e_bot) + g.dZ_subroundoff);
  const double Ihtot_lay = 1.0 / ((e_top - e_bot) + g.dZ_subroundoff);
  for (int k = 0; k < nz; k++) {
    const long o3 = o2 + pl * k;
    const double e_Kp1 = p.e[o3 + pl];
    double T_c = 0., S_c = 0.;
    if (MODE == PCM_NOEOS) {      // :777-781
      const double hk = p.h[o3];
      const double dz_geo = g.g_Earth * g.H_to_Z * hk;
      p.dpa[o3] = (p.Rlay[k] - rho_ref) * dz_geo;
      p.intz_dpa[o3] = 0.5 * (p.Rlay[k] - rho_ref) * dz_geo * hk;
    } else {
      pcm_TS(p, col, o3, k, T_c, S_c);
      const double dz = e_K - e_Kp1;
      if (MODE == PCM_LINEAR) {      // MOM_EOS_linear.F90:339-344
        const double rho_anom = (p.eos.Rho_T0_S0 - rho_ref) + p.eos.dRho_dT * T_c + p.eos.dRho_dS * S_c;
        p.dpa[o3] = G_e * rho_anom * dz;
        p.intz_dpa[o3] = (0.5 * G_e * rho_anom * (dz * dz)) * g.Z_to_H;
      } else {
        const WrightTerms w = wright_terms(T_c, S_c);
        double iz;
        p.dpa[o3] = wright_layer<true>(w.al0, w.p0, w.lambda, dz, 0.5 * (e_K + e_Kp1), GxRho, G_e, I_Rho, rho_ref, p.Z_ref, iz);
        p.intz_dpa[o3] = iz * g.Z_to_H;
      }
    }
    if (p.pbce) {      // Set_pbce_Bouss(e, tv_tmp, ...) MOM_PressureForce_Montgomery.F90:702-745
      double pb;
      if (MODE == PCM_NOEOS) {
        if (k == 0) pb = p.g_prime[0] * g.H_to_Z;
        else pb = pbce_prev + (p.g_prime[k] * g.H_to_Z) * ((e_K - e_bot) * Ihtot_lay);
      } else {
        const double press = -Rho0xG * (e_K - p.Z_ref);
        if (k == 0) {
          const double rho_in_situ = eos_density(p.eos, T_c, S_c, press);
          pb = G_Rho0 * (p.GFS_scale * rho_in_situ) * g.H_to_Z;
        } else {
          const double T_int = 0.5 * (T_m + T_c), S_int = 0.5 * (S_m + S_c);
          double dR_dT, dR_dS;
          eos_density_derivs(p.eos, T_int, S_int, press, dR_dT, dR_dS);
          pb = pbce_prev + G_Rho0 * ((e_K - e_bot) * Ihtot_eos) * (dR_dT * (T_c - T_m) + dR_dS * (S_c - S_m));
        }
      }
      p.pbce[o3] = pb;
      pbce_prev = pb;
    }
    e_K = e_Kp1; T_m = T_c; S_m = S_c;
  }
}

// one face between the columns L and R of layer k (zero-based): intx_dpa / inty_dpa
template <int MODE>
__device__ __forceinline__ double pcm_face_integral(const PgfArgs &p, const PcmCol &cL, const PcmCol &cR, long L3, long R3, long oL2, long oR2,
                                                    int k, double eL_K, double eL_Kp1, double eR_K, double eR_Kp1) {
  const m6::GridDev &g = p.g;
  const double G_e = g.g_Earth, rho_ref = p.rho_ref;
  const double C1_6 = 1.0 / 6.0, C1_90 = 1.0 / 90.0;
  if (MODE == PCM_NOEOS) {      // :782-789
    const double dzgL = g.g_Earth * g.H_to_Z * p.h[L3], dzgR = g.g_Earth * g.H_to_Z * p.h[R3];
    return 0.5 * (p.Rlay[k] - rho_ref) * (dzgL + dzgR);
  }
  double TL, SL, TR, SR;
  pcm_TS(p, cL, L3, k, TL, SL); pcm_TS(p, cR, R3, k, TR, SR);
  double hWght = 0.0;
  if (p.massw) hWght = max3(0., -g.bathyT[oL2] - eR_K, -g.bathyT[oR2] - eL_K);
  if (MODE == PCM_LINEAR) {      // MOM_EOS_linear.F90:346-383
    const double R0 = p.eos.Rho_T0_S0, dRdT = p.eos.dRho_dT, dRdS = p.eos.dRho_dS;
    if (hWght <= 0.0) {
      const double dzL = eL_K - eL_Kp1, dzR = eR_K - eR_Kp1;
      const double raL = (R0 - rho_ref) + (dRdT * TL + dRdS * SL);
      const double raR = (R0 - rho_ref) + (dRdT * TR + dRdS * SR);
      return G_e * C1_6 * (dzL * (2.0 * raL + raR) + dzR * (2.0 * raR + raL));
    }
    const double hL = (eL_K - eL_Kp1) + g.dZ_subroundoff;
    const double hR = (eR_K - eR_Kp1) + g.dZ_subroundoff;
    const double rr = (hL - hR) / (hL + hR);
    hWght = hWght * (rr * rr);
    const double iDenom = 1.0 / (hWght * (hR + hL) + hL * hR);
    const double hWt_LL = (hWght * hL + hR * hL) * iDenom, hWt_LR = (hWght * hR) * iDenom;
    const double hWt_RR = (hWght * hR + hR * hL) * iDenom, hWt_RL = (hWght * hL) * iDenom;
    double intz[5];
    intz[0] = p.dpa[L3]; intz[4] = p.dpa[R3];
#pragma unroll
    for (int m = 2; m <= 4; m++) {
      const double wt_L = 0.25 * (double)(5 - m), wt_R = 1.0 - wt_L;
      const double wtT_L = wt_L * hWt_LL + wt_R * hWt_RL, wtT_R = wt_L * hWt_LR + wt_R * hWt_RR;
      const double dz = wt_L * (eL_K - eL_Kp1) + wt_R * (eR_K - eR_Kp1);
      const double rho_anom = (R0 - rho_ref) + (dRdT * (wtT_L * TL + wtT_R * TR) + dRdS * (wtT_L * SL + wtT_R * SR));
      intz[m - 1] = G_e * rho_anom * dz;
    }
    return C1_90 * (7.0 * (intz[0] + intz[4]) + 32.0 * (intz[1] + intz[3]) + 12.0 * intz[2]);
  }
  // MOM_EOS_Wright.F90:560-599
  double hWt_LL, hWt_LR, hWt_RR, hWt_RL;
  if (hWght > 0.) {
    const double hL = (eL_K - eL_Kp1) + g.dZ_subroundoff;
    const double hR = (eR_K - eR_Kp1) + g.dZ_subroundoff;
    const double rr = (hL - hR) / (hL + hR);
    hWght = hWght * (rr * rr);
    const double iDenom = 1.0 / (hWght * (hR + hL) + hL * hR);
    hWt_LL = (hWght * hL + hR * hL) * iDenom; hWt_LR = (hWght * hR) * iDenom;
    hWt_RR = (hWght * hR + hR * hL) * iDenom; hWt_RL = (hWght * hL) * iDenom;
  } else {
    hWt_LL = 1.0; hWt_LR = 0.0; hWt_RR = 1.0; hWt_RL = 0.0;
  }
  const WrightTerms wL = wright_terms(TL, SL), wR = wright_terms(TR, SR);
  const double GxRho = G_e * p.rho_ref, I_Rho = 1.0 / p.rho_ref;
  double intz[5], unused;
  intz[0] = p.dpa[L3]; intz[4] = p.dpa[R3];
#pragma unroll
  for (int m = 2; m <= 4; m++) {
    const double wt_L = 0.25 * (double)(5 - m), wt_R = 1.0 - wt_L;
    const double wtT_L = wt_L * hWt_LL + wt_R * hWt_RL, wtT_R = wt_L * hWt_LR + wt_R * hWt_RR;
    const double al0 = wtT_L * wL.al0 + wtT_R * wR.al0;
    const double p0 = wtT_L * wL.p0 + wtT_R * wR.p0;
    const double lambda = wtT_L * wL.lambda + wtT_R * wR.lambda;
    const double dz = wt_L * (eL_K - eL_Kp1) + wt_R * (eR_K - eR_Kp1);
    intz[m - 1] = wright_layer<false>(al0, p0, lambda, dz, 0.5 * (wt_L * (eL_K + eL_Kp1) + wt_R * (eR_K + eR_Kp1)), GxRho, G_e, I_Rho, rho_ref,
                                      p.Z_ref, unused);
  }
  return C1_90 * (7.0 * (intz[0] + intz[4]) + 32.0 * (intz[1] + intz[3]) + 12.0 * intz[2]);
}

template <int MODE>
__global__ __launch_bounds__(64) void pgf_pcm_face_kernel(PgfArgs p) {
  const m6::GridDev &g = p.g;
  const int i = g.isc - 1 + blockIdx.x * 64 + threadIdx.x;
  const int j = g.jsc - 1 + blockIdx.y;
  if (i > g.iec) return;
  const bool do_x = (j >= g.jsc), do_y = (i >= g.isc);
  if (!do_x && !do_y) return;
  const int nz = g.nk;
  const long pl = (long)g.nih * g.njh, plU = (long)(g.nih + 1) * g.njh, plV = (long)g.nih * (g.njh + 1);
  const long oc = g.h2(i, j), oe = oc + 1, on = oc + g.nih;
  const double h_neglect = g.H_subroundoff, I_Rho0 = 1.0 / g.Rho0;
  const double rg = p.rho_ref * g.g_Earth;
  auto pa0 = [&](long o2) -> double {
    double v = rg * (p.e[o2] - p.Z_ref);
    if (p.p_atm) v = v + p.p_atm[o2];
    return v;
  };
  PcmCol cc, ce, cn; cc.nkmb = ce.nkmb = cn.nkmb = 0;
  if (MODE != PCM_NOEOS) {
    cc = pcm_col(p, oc, pl);
    if (do_x) ce = pcm_col(p, oe, pl);
    if (do_y) cn = pcm_col(p, on, pl);
  }
  double pa_c = pa0(oc), pa_e = do_x ? pa0(oe) : 0.0, pa_n = do_y ? pa0(on) : 0.0;
  double intx_pa = 0.5 * (pa_c + pa_e), inty_pa = 0.5 * (pa_c + pa_n);
  const double fx = do_x ? (2.0 * I_Rho0 * g.IdxCu[g.u2(i, j)]) : 0.0;
  const double fy = do_y ? (2.0 * I_Rho0 * g.IdyCv[g.v2(i, j)]) : 0.0;
  double ec_K = p.e[oc], ee_K = do_x ? p.e[oe] : 0.0, en_K = do_y ? p.e[on] : 0.0;
  for (int k = 0; k < nz; k++) {
    const long c3 = oc + pl * k;
    const double ec_Kp1 = p.e[c3 + pl];
    const double h_c = p.h[c3], dpa_c = p.dpa[c3], iz_c = p.intz_dpa[c3];
    if (do_x) {
      const long e3 = c3 + 1;
      const double ee_Kp1 = p.e[e3 + pl];
      const double h_e = p.h[e3];
      const double intx_dpa = pcm_face_integral<MODE>(p, cc, ce, c3, e3, oc, oe, k, ec_K, ec_Kp1, ee_K, ee_Kp1);
      p.PFu[g.u2(i, j) + plU * k] = (((pa_c * h_c + iz_c) - (pa_e * h_e + p.intz_dpa[e3])) +
                                     ((h_e - h_c) * intx_pa - (ee_Kp1 - ec_Kp1) * intx_dpa * g.Z_to_H)) *
                                    (fx / ((h_c + h_e) + h_neglect));
      intx_pa = intx_pa + intx_dpa;
      pa_e = pa_e + p.dpa[e3];
      ee_K = ee_Kp1;
    }
    if (do_y) {
      const long n3 = c3 + g.nih;
      const double en_Kp1 = p.e[n3 + pl];
      const double h_n = p.h[n3];
      const double inty_dpa = pcm_face_integral<MODE>(p, cc, cn, c3, n3, oc, on, k, ec_K, ec_Kp1, en_K, en_Kp1);
      p.PFv[g.v2(i, j) + plV * k] = (((pa_c * h_c + iz_c) - (pa_n * h_n + p.intz_dpa[n3])) +
                                     ((h_n - h_c) * inty_pa - (en_Kp1 - ec_Kp1) * inty_dpa * g.Z_to_H)) *
                                    (fy / ((h_c + h_n) + h_neglect));
      inty_pa = inty_pa + inty_dpa;
      pa_n = pa_n + p.dpa[n3];
      en_K = en_Kp1;
    }
    pa_c = pa_c + dpa_c;
    ec_K = ec_Kp1;
  }
}

// ======== PressureForce_FV_nonBouss (MOM_PressureForce_FV.F90:89-452) ==================================================
// The same two-kernel split: p.e holds the interface PRESSURES (nk+1 planes, top down), p.dpa the layers' dza, p.intz_dpa
// their intp_dza (int_spec_vol_dp_generic_plm, MOM_density_integrals.F90:1479-1726), p.za0 the surface geopotential anomaly.
__global__ __launch_bounds__(64) void pgfnb_column_kernel(PgfArgs p) {
  const m6::GridDev &g = p.g;
  const int i = g.isc - 1 + blockIdx.x * 64 + threadIdx.x;
  const int j = g.jsc - 1 + blockIdx.y;
  if (i > g.iec + 1) return;
  const int nz = g.nk;
  const long o2 = g.h2(i, j), pl = (long)g.nih * g.njh;
  const double h_neglect = g.H_subroundoff;
  const double H_to_RL2_T2 = g.g_Earth * p.H_to_RZ;
  const double dp_neglect = g.g_Earth * p.H_to_RZ * g.H_subroundoff;
  const double alpha_ref = 1.0 / p.rho_ref;
  const double C1_90 = 1.0 / 90.0;
  // :193-206
  double pk = p.p_atm ? p.p_atm[o2] : 0.0;
  const double p_top = pk;
  p.e[o2] = pk;
  for (int k = 0; k < nz; k++) {
    pk = pk + H_to_RL2_T2 * p.h[o2 + pl * k];
    p.e[o2 + pl * (k + 1)] = pk;
  }
  const double p_bot = pk;
  if (p.eta) {      // :417-428
    const double Pa_to_H = 1.0 / (g.g_Earth * p.H_to_RZ);
    p.eta[o2] = p.p_atm ? (p_bot - p_top) * Pa_to_H : p_bot * Pa_to_H;
  }
  auto ld = [&](const double *a, int kk) -> double { return (kk >= 0 && kk < nz) ? a[o2 + pl * kk] : 0.0; };
  double h_m = 0., h_c = ld(p.h, 0), h_p = ld(p.h, 1), h_q = ld(p.h, 2);
  double T_m = 0., T_c = ld(p.T, 0), T_p = ld(p.T, 1), T_q = ld(p.T, 2);
  double S_m = 0., S_c = ld(p.S, 0), S_p = ld(p.S, 1), S_q = ld(p.S, 2);
  double sT_m = 0., sT_c = 0., sT_p = 0., sS_m = 0., sS_c = 0., sS_p = 0.;
  if (nz >= 3) {
    sT_p = plm_slope_wa(h_c, h_p, h_q, h_neglect, T_c, T_p, T_q);
    sS_p = plm_slope_wa(h_c, h_p, h_q, h_neglect, S_c, S_p, S_q);
  }
  double p_K = p_top;
  for (int kk = 0; kk < nz; kk++) {
    const long o3 = o2 + pl * kk;
    // ---- TS_PLM_edge_values (ALE_PLM_edge_values, MOM_ALE.F90:1549-1576) ----
    double Tt, Tb, St, Sb;
    if (kk >= 1 && kk <= nz - 2) {
      const double mT = plm_monotonized_slope(T_m, T_c, T_p, sT_m, sT_c, sT_p);
      Tt = T_c - 0.5 * mT; Tb = T_c + 0.5 * mT;
      const double mS = plm_monotonized_slope(S_m, S_c, S_p, sS_m, sS_c, sS_p);
      St = S_c - 0.5 * mS; Sb = S_c + 0.5 * mS;
    } else if (p.boundary_extrap) {
      if (kk == 0) {
        const double mT = -plm_extrapolate_slope(h_p, h_c, h_neglect, T_p, T_c);
        Tt = T_c - 0.5 * mT; Tb = T_c + 0.5 * mT;
        const double mS = -plm_extrapolate_slope(h_p, h_c, h_neglect, S_p, S_c);
        St = S_c - 0.5 * mS; Sb = S_c + 0.5 * mS;
      } else {
        const double mT = plm_extrapolate_slope(h_m, h_c, h_neglect, T_m, T_c);
        Tt = T_c - 0.5 * mT; Tb = T_c + 0.5 * mT;
        const double mS = plm_extrapolate_slope(h_m, h_c, h_neglect, S_m, S_c);
        St = S_c - 0.5 * mS; Sb = S_c + 0.5 * mS;
      }
    } else {
      Tt = T_c; Tb = T_c; St = S_c; Sb = S_c;
    }
    p.T_t[o3] = Tt; p.T_b[o3] = Tb; p.S_t[o3] = St; p.S_b[o3] = Sb;
    // ---- the vertical integrals :1556-1574 (the weights run the other way than in int_density_dz) ----
    const double p_Kp1 = p.e[o3 + pl];
    double a5[5];
#pragma unroll
    for (int n = 1; n <= 5; n++) {
      const double wt_t = 0.25 * (double)(n - 1), wt_b = 1.0 - wt_t;
      const double p5 = wt_t * p_K + wt_b * p_Kp1;
      const double S5 = wt_t * St + wt_b * Sb;
      const double T5 = wt_t * Tt + wt_b * Tb;
      a5[n - 1] = eos_spec_vol_anomaly(p.eos, T5, S5, p5, alpha_ref);
    }
    const double dp = p_Kp1 - p_K;
    const double alpha_anom = C1_90 * ((7.0 * (a5[0] + a5[4]) + 32.0 * (a5[1] + a5[3])) + 12.0 * a5[2]);
    p.dpa[o3] = dp * alpha_anom;
    p.intz_dpa[o3] = 0.5 * (dp * dp) * (alpha_anom - C1_90 * (16.0 * (a5[3] - a5[1]) + 7.0 * (a5[4] - a5[0])));
    p_K = p_Kp1;
    h_m = h_c; h_c = h_p; h_p = h_q; h_q = ld(p.h, kk + 3);
    T_m = T_c; T_c = T_p; T_p = T_q; T_q = ld(p.T, kk + 3);
    S_m = S_c; S_c = S_p; S_p = S_q; S_q = ld(p.S, kk + 3);
    sT_m = sT_c; sT_c = sT_p; sS_m = sS_c; sS_c = sS_p;
    if (kk + 2 <= nz - 2) {
      sT_p = plm_slope_wa(h_c, h_p, h_q, h_neglect, T_c, T_p, T_q);
      sS_p = plm_slope_wa(h_c, h_p, h_q, h_neglect, S_c, S_p, S_q);
    } else {
      sT_p = 0.; sS_p = 0.;
    }
  }
  // :299-306: za at the sea surface, summed from the bottom
  double za = alpha_ref * p_bot - g.g_Earth * g.bathyT[o2];
  for (int k = nz - 1; k >= 0; k--) za = za + p.dpa[o2 + pl * k];
  p.za0[o2] = za;
  // Set_pbce_nonBouss, MOM_PressureForce_Montgomery.F90:809-830
  if (p.pbce) {
    const double dP_dH = g.g_Earth * p.H_to_RZ;
    const double C_htot = dP_dH / ((p_bot - p_top) + dp_neglect);
    double Tk1 = p.T[o2 + pl * (nz - 1)], Sk1 = p.S[o2 + pl * (nz - 1)];
    double pb = dP_dH / eos_density(p.eos, Tk1, Sk1, p_bot);
    p.pbce[o2 + pl * (nz - 1)] = pb;
    for (int k = nz - 2; k >= 0; k--) {
      const double Tk = p.T[o2 + pl * k], Sk = p.S[o2 + pl * k];
      const double T_int = 0.5 * (Tk + Tk1), S_int = 0.5 * (Sk + Sk1);
      const double pi = p.e[o2 + pl * (k + 1)];
      const double rho = eos_density(p.eos, T_int, S_int, pi);
      double dR_dT, dR_dS;
      eos_density_derivs(p.eos, T_int, S_int, pi, dR_dT, dR_dS);
      pb = pb + ((pi - p_top) * C_htot) * ((dR_dT * (Tk1 - Tk) + dR_dS * (Sk1 - Sk)) / (rho * rho));
      p.pbce[o2 + pl * k] = pb;
      Tk1 = Tk; Sk1 = Sk;
    }
  }
}

// the 15 specific-volume evaluations across one face :1576-1640
__device__ __forceinline__ double face_integral_nb(const PgfArgs &p, long L3, long R3, long oL2, long oR2, long pl, double dp_neglect,
                                                   double alpha_ref) {
  const int nz = p.g.nk;
  const double C1_90 = 1.0 / 90.0;
  const double PT_L = p.e[L3], PB_L = p.e[L3 + pl], PT_R = p.e[R3], PB_R = p.e[R3 + pl];
  double hWght = 0.0, hWt_LL, hWt_LR, hWt_RR, hWt_RL;
  if (p.massw) hWght = max3(0., p.e[oL2 + pl * nz] - PT_R, p.e[oR2 + pl * nz] - PT_L);
  if (hWght > 0.) {
    const double hL = (PB_L - PT_L) + dp_neglect;
    const double hR = (PB_R - PT_R) + dp_neglect;
    const double rr = (hL - hR) / (hL + hR);
    hWght = hWght * (rr * rr);
    const double iDenom = 1.0 / (hWght * (hR + hL) + hL * hR);
    hWt_LL = (hWght * hL + hR * hL) * iDenom; hWt_LR = (hWght * hR) * iDenom;
    hWt_RR = (hWght * hR + hR * hL) * iDenom; hWt_RL = (hWght * hL) * iDenom;
  } else {
    hWt_LL = 1.0; hWt_LR = 0.0; hWt_RR = 1.0; hWt_RL = 0.0;
  }
  const double TtL = p.T_t[L3], TtR = p.T_t[R3], TbL = p.T_b[L3], TbR = p.T_b[R3];
  const double StL = p.S_t[L3], StR = p.S_t[R3], SbL = p.S_b[L3], SbR = p.S_b[R3];
  double intp[5];
  intp[0] = p.dpa[L3]; intp[4] = p.dpa[R3];
#pragma unroll
  for (int m = 2; m <= 4; m++) {
    const double wt_L = 0.25 * (double)(5 - m), wt_R = 1.0 - wt_L;
    const double wtT_L = wt_L * hWt_LL + wt_R * hWt_RL, wtT_R = wt_L * hWt_LR + wt_R * hWt_RR;
    const double P_top = wt_L * PT_L + wt_R * PT_R;
    const double P_bot = wt_L * PB_L + wt_R * PB_R;
    const double T_top = wtT_L * TtL + wtT_R * TtR;
    const double T_bot = wtT_L * TbL + wtT_R * TbR;
    const double S_top = wtT_L * StL + wtT_R * StR;
    const double S_bot = wtT_L * SbL + wtT_R * SbR;
    const double dp_90 = C1_90 * (P_bot - P_top);
    double a[5];
#pragma unroll
    for (int n = 1; n <= 5; n++) {
      const double wt_t = 0.25 * (double)(n - 1), wt_b = 1.0 - wt_t;
      a[n - 1] = eos_spec_vol_anomaly(p.eos, wt_t * T_top + wt_b * T_bot, wt_t * S_top + wt_b * S_bot, wt_t * P_top + wt_b * P_bot, alpha_ref);
    }
    intp[m - 1] = dp_90 * ((7.0 * (a[0] + a[4]) + 32.0 * (a[1] + a[3])) + 12.0 * a[2]);
  }
  return C1_90 * ((7.0 * (intp[0] + intp[4]) + 32.0 * (intp[1] + intp[3])) + 12.0 * intp[2]);
}

__global__ __launch_bounds__(64) void pgfnb_face_kernel(PgfArgs p) {
  const m6::GridDev &g = p.g;
  const int i = g.isc - 1 + blockIdx.x * 64 + threadIdx.x;
  const int j = g.jsc - 1 + blockIdx.y;
  if (i > g.iec) return;
  const bool do_x = (j >= g.jsc), do_y = (i >= g.isc);
  if (!do_x && !do_y) return;
  const int nz = g.nk;
  const long pl = (long)g.nih * g.njh, plU = (long)(g.nih + 1) * g.njh, plV = (long)g.nih * (g.njh + 1);
  const long oc = g.h2(i, j), oe = oc + 1, on = oc + g.nih;
  const double H_to_RL2_T2 = g.g_Earth * p.H_to_RZ;
  const double dp_neglect = g.g_Earth * p.H_to_RZ * g.H_subroundoff;
  const double alpha_ref = 1.0 / p.rho_ref;
  double za_c = p.za0[oc], za_e = do_x ? p.za0[oe] : 0.0, za_n = do_y ? p.za0[on] : 0.0;
  double intx_za = 0.5 * (za_c + za_e), inty_za = 0.5 * (za_c + za_n);      // :366-371
  const double fx = do_x ? (2.0 * g.IdxCu[g.u2(i, j)]) : 0.0;
  const double fy = do_y ? (2.0 * g.IdyCv[g.v2(i, j)]) : 0.0;
  for (int k = 0; k < nz; k++) {      // :373-399
    const long c3 = oc + pl * k;
    const double dp_c = H_to_RL2_T2 * p.h[c3];
    za_c = za_c - p.dpa[c3];
    const double ipd_c = p.intz_dpa[c3], pc_K = p.e[c3];
    if (do_x) {
      const long e3 = c3 + 1;
      const double dp_e = H_to_RL2_T2 * p.h[e3];
      za_e = za_e - p.dpa[e3];
      const double intx_dza = face_integral_nb(p, c3, e3, oc, oe, pl, dp_neglect, alpha_ref);
      intx_za = intx_za - intx_dza;
      p.PFu[g.u2(i, j) + plU * k] = (((za_c * dp_c + ipd_c) - (za_e * dp_e + p.intz_dpa[e3])) +
                                     ((dp_e - dp_c) * intx_za - (p.e[e3] - pc_K) * intx_dza)) *
                                    (fx / ((dp_c + dp_e) + dp_neglect));
    }
    if (do_y) {
      const long n3 = c3 + g.nih;
      const double dp_n = H_to_RL2_T2 * p.h[n3];
      za_n = za_n - p.dpa[n3];
      const double inty_dza = face_integral_nb(p, c3, n3, oc, on, pl, dp_neglect, alpha_ref);
      inty_za = inty_za - inty_dza;
      p.PFv[g.v2(i, j) + plV * k] = (((za_c * dp_c + ipd_c) - (za_n * dp_n + p.intz_dpa[n3])) +
                                     ((dp_n - dp_c) * inty_za - (p.e[n3] - pc_K) * inty_dza)) *
                                    (fy / ((dp_c + dp_n) + dp_neglect));
    }
  }
}

__global__ void eos_density_kernel(EosDev E, const double *T, const double *S, const double *pr, double *rho, long n,
                                   int use_ref, double rho_ref) {
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x)
    rho[t] = use_ref ? eos_density_anomaly(E, T[t], S[t], pr[t], rho_ref) : eos_density(E, T[t], S[t], pr[t]);
}

int check_eos(const mom6hip_eos_t *eos) {
  M6_REQUIRE(eos != nullptr, "MOM_EOS: the equation of state must be initialized (EOS_init) before it is used");
  M6_REQUIRE(eos->form == MOM6HIP_EOS_WRIGHT || eos->form == MOM6HIP_EOS_LINEAR || eos->form == MOM6HIP_EOS_UNESCO ||
                 eos->form == MOM6HIP_EOS_WRIGHT_FULL || eos->form == MOM6HIP_EOS_WRIGHT_REDUCED,
             "MOM_EOS: EQN_OF_STATE form %d is not provided by libmom6hip (WRIGHT, WRIGHT_FULL, WRIGHT_REDUCED, UNESCO, LINEAR)", eos->form);
  return 0;
}

}  // namespace

extern "C" int mom6hip_calculate_density(mom6hip_ctx_t *ctx, const mom6hip_eos_t *eos, const double *T, const double *S,
                                         const double *pressure, double *rho, int64_t n, int32_t use_rho_ref,
                                         double rho_ref, int32_t memspace) {
  M6_REQUIRE(ctx && T && S && pressure && rho && n >= 0, "calculate_density: bad argument");
  if (check_eos(eos)) return 2;
  if (n == 0) return 0;
  m6::Stager st(ctx, memspace);
  const double *dT = st.in(T, n * 8), *dS = st.in(S, n * 8), *dp = st.in(pressure, n * 8);
  double *dr = st.out(rho, n * 8);
  if (st.failed()) return 1;
  EosDev E{eos->form, eos->Rho_T0_S0, eos->dRho_dT, eos->dRho_dS};
  int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(eos_density_kernel, dim3(blocks), dim3(256), 0, ctx->stream, E, dT, dS, dp, dr, (long)n,
                     (int)use_rho_ref, rho_ref);
  M6_HIP(hipGetLastError());
  return st.finish();
}

extern "C" int mom6hip_pressureforce_fv_bouss(mom6hip_ctx_t *ctx, const mom6hip_pressureforce_cs_t *cs,
                                              const mom6hip_eos_t *eos, const double *h, const double *T,
                                              const double *S, const double *p_atm, double *PFu, double *PFv,
                                              double *pbce, double *eta, int32_t memspace) {
  M6_REQUIRE(ctx != nullptr && cs != nullptr, "MOM_PressureForce_FV_Bouss: Module must be initialized before it is used.");
  M6_REQUIRE(h && PFu && PFv, "PressureForce_FV_Bouss: null argument");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "PressureForce_FV_Bouss: bad memspace");
  const bool use_EOS = (eos != nullptr);      // associated(tv%eqn_of_state)
  if (use_EOS) {
    if (check_eos(eos)) return 2;
    M6_REQUIRE(T && S, "PressureForce_FV_Bouss: tv%%T and tv%%S are needed with an equation of state");
  }
  const bool use_ALE = cs->use_ALE && cs->reconstruct && use_EOS;      // :561-562
  M6_REQUIRE(!use_ALE || cs->Recon_Scheme == 1,
             "PressureForce_FV_Bouss: with RECONSTRUCT_FOR_PRESSURE only PRESSURE_RECONSTRUCTION_SCHEME=1 (PLM) is provided");
  M6_REQUIRE(cs->GFS_scale == 1.0, "PressureForce_FV_Bouss: GFS_scale < 1 is not provided");
  M6_REQUIRE(cs->nkmb >= 0 && cs->nkmb < ctx->g.nk, "PressureForce_FV_Bouss: GV%%nk_rho_varies must be in 0 .. nk-1");
  M6_REQUIRE(!(use_ALE && cs->nkmb > 0), "PressureForce_FV_Bouss: a bulk mixed layer cannot be used with ALE");
  int mode = -1;
  if (use_EOS && !use_ALE) {
    M6_REQUIRE(eos->form == MOM6HIP_EOS_LINEAR || eos->form == MOM6HIP_EOS_WRIGHT,
               "PressureForce_FV_Bouss: without the PLM reconstruction (RECONSTRUCT_FOR_PRESSURE = False, or no ALE) int_density_dz is "
               "provided for EQN_OF_STATE = LINEAR and WRIGHT (their analytic integrals; EOS_QUADRATURE = False)");
    mode = eos->form == MOM6HIP_EOS_LINEAR ? PCM_LINEAR : PCM_WRIGHT;
  } else if (!use_EOS) {
    mode = PCM_NOEOS;
    M6_REQUIRE(cs->Rlay, "PressureForce_FV_Bouss: GV%%Rlay is needed without an equation of state");
    M6_REQUIRE(!pbce || cs->g_prime, "PressureForce_FV_Bouss: GV%%g_prime is needed for pbce without an equation of state");
  }
  M6_REQUIRE(cs->nkmb == 0 || cs->Rlay, "PressureForce_FV_Bouss: GV%%Rlay is needed with a bulk mixed layer");
  m6::GridDev &g = ctx->g;
  M6_REQUIRE(g.bathyT && g.IdxCu && g.IdyCv, "PressureForce_FV_Bouss: a required grid metric is missing");
  M6_REQUIRE(g.nk >= 2 || !use_ALE, "PressureForce_FV_Bouss: at least 2 layers are needed");
  M6_REQUIRE(g.isc - g.isd >= 1 && g.ied - g.iec >= 1 && g.jsc - g.jsd >= 1 && g.jed - g.jec >= 1,
             "PressureForce_FV_Bouss: needs a halo of at least 1");
  hipStream_t s = ctx->stream;
  const size_t bH = (size_t)g.nh3() * 8, bU = (size_t)g.nu3() * 8, bV = (size_t)g.nv3() * 8;
  const size_t bH2 = (size_t)g.nih * g.njh * 8;
  PgfArgs a;
  a.Rlay = a.g_prime = nullptr;
  a.bc_CAu = a.bc_CAv = a.bc_diffu = a.bc_diffv = nullptr; a.bc_u = a.bc_v = nullptr; a.bc_inviscid = 0;
  if (cs->Rlay && (mode == PCM_NOEOS || cs->nkmb > 0)) {
    a.Rlay = ctx->tables[m6::TABLE_PGF_RLAY].get(cs->Rlay, (size_t)g.nk);
    if (!a.Rlay) return 1;
  }
  if (mode == PCM_NOEOS && pbce) {
    a.g_prime = ctx->tables[m6::TABLE_PGF_GPRIME].get(cs->g_prime, (size_t)g.nk + 1);
    if (!a.g_prime) return 1;
  }
  m6::Stager st(ctx, memspace);
  a.g = g;
  a.eos = use_EOS ? EosDev{eos->form, eos->Rho_T0_S0, eos->dRho_dT, eos->dRho_dS} : EosDev{0, 0., 0., 0.};
  a.h = st.in(h, bH); a.T = use_EOS ? st.in(T, bH) : nullptr; a.S = use_EOS ? st.in(S, bH) : nullptr; a.p_atm = st.in(p_atm, bH2);
  a.PFu = st.inout(PFu, bU); a.PFv = st.inout(PFv, bV); a.pbce = st.inout(pbce, bH); a.eta = st.inout(eta, bH2);
  a.e = (double *)st.scratch(bH + bH2);
  a.T_t = a.T_b = a.S_t = a.S_b = nullptr;
  if (use_ALE) {
    a.T_t = (double *)st.scratch(bH); a.T_b = (double *)st.scratch(bH);
    a.S_t = (double *)st.scratch(bH); a.S_b = (double *)st.scratch(bH);
  }
  a.dpa = (double *)st.scratch(bH); a.intz_dpa = (double *)st.scratch(bH);
  if (st.failed()) return 1;
  a.rho_ref = cs->Rho0; a.Z_ref = cs->Z_ref; a.GFS_scale = cs->GFS_scale; a.za0 = nullptr; a.H_to_RZ = 0.0;
  a.boundary_extrap = cs->boundary_extrap; a.massw = cs->useMassWghtInterp;
  a.nkmb = use_EOS ? cs->nkmb : 0; a.P_Ref = cs->P_Ref;
  const int ncol_i = g.iec - g.isc + 3, ncol_j = g.jec - g.jsc + 3;
  const dim3 gc((ncol_i + 63) / 64, ncol_j), gf((ncol_i - 1 + 63) / 64, ncol_j - 1);
  if (use_ALE) {
    if (ctx->bc_fuse && memspace == MOM6HIP_MEM_DEVICE) {
      mom6hip_ctx::BcAccelFuse *f = ctx->bc_fuse;
      a.bc_CAu = f->au; a.bc_CAv = f->av; a.bc_diffu = f->diffu; a.bc_diffv = f->diffv; a.bc_u = f->u_bc; a.bc_v = f->v_bc;
      a.bc_inviscid = f->inviscid;
      f->done = true;
    }
    hipLaunchKernelGGL(pgf_column_kernel, gc, dim3(64), 0, s, a);
    { m6::KTimer kt(ctx, MOM6HIP_KT_PGF_FACE); hipLaunchKernelGGL(pgf_face_kernel, gf, dim3(64), 0, s, a); }
  } else if (mode == PCM_LINEAR) {
    hipLaunchKernelGGL(pgf_pcm_column_kernel<PCM_LINEAR>, gc, dim3(64), 0, s, a);
    hipLaunchKernelGGL(pgf_pcm_face_kernel<PCM_LINEAR>, gf, dim3(64), 0, s, a);
  } else if (mode == PCM_WRIGHT) {
    hipLaunchKernelGGL(pgf_pcm_column_kernel<PCM_WRIGHT>, gc, dim3(64), 0, s, a);
    hipLaunchKernelGGL(pgf_pcm_face_kernel<PCM_WRIGHT>, gf, dim3(64), 0, s, a);
  } else {
    hipLaunchKernelGGL(pgf_pcm_column_kernel<PCM_NOEOS>, gc, dim3(64), 0, s, a);
    hipLaunchKernelGGL(pgf_pcm_face_kernel<PCM_NOEOS>, gf, dim3(64), 0, s, a);
  }
  M6_HIP(hipGetLastError());
  return st.finish();
}

extern "C" int mom6hip_pressureforce_fv_nonbouss(mom6hip_ctx_t *ctx, const mom6hip_pressureforce_cs_t *cs,
                                                 const mom6hip_eos_t *eos, const double *h, const double *T,
                                                 const double *S, const double *p_atm, double H_to_RZ, double *PFu,
                                                 double *PFv, double *pbce, double *eta, int32_t memspace) {
  M6_REQUIRE(ctx != nullptr && cs != nullptr, "MOM_PressureForce_FV_nonBouss: Module must be initialized before it is used.");
  M6_REQUIRE(h && T && S && PFu && PFv, "PressureForce_FV_nonBouss: null argument");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "PressureForce_FV_nonBouss: bad memspace");
  if (check_eos(eos)) return 2;
  M6_REQUIRE(cs->use_ALE && cs->reconstruct && cs->Recon_Scheme == 1,
             "PressureForce_FV_nonBouss: only ALE with RECONSTRUCT_FOR_PRESSURE=True and PRESSURE_RECONSTRUCTION_SCHEME=1 is provided");
  M6_REQUIRE(cs->GFS_scale == 1.0, "PressureForce_FV_nonBouss: GFS_scale < 1 is not provided");
  M6_REQUIRE(H_to_RZ > 0.0, "PressureForce_FV_nonBouss: H_to_RZ must be positive");
  m6::GridDev &g = ctx->g;
  M6_REQUIRE(g.bathyT && g.IdxCu && g.IdyCv, "PressureForce_FV_nonBouss: a required grid metric is missing");
  M6_REQUIRE(g.nk >= 2, "PressureForce_FV_nonBouss: at least 2 layers are needed");
  M6_REQUIRE(g.isc - g.isd >= 1 && g.ied - g.iec >= 1 && g.jsc - g.jsd >= 1 && g.jed - g.jec >= 1,
             "PressureForce_FV_nonBouss: needs a halo of at least 1");
  hipStream_t s = ctx->stream;
  const size_t bH = (size_t)g.nh3() * 8, bU = (size_t)g.nu3() * 8, bV = (size_t)g.nv3() * 8;
  const size_t bH2 = (size_t)g.nih * g.njh * 8;
  m6::Stager st(ctx, memspace);
  PgfArgs a;
  a.bc_CAu = a.bc_CAv = a.bc_diffu = a.bc_diffv = nullptr; a.bc_u = a.bc_v = nullptr; a.bc_inviscid = 0;
  a.g = g;
  a.eos = EosDev{eos->form, eos->Rho_T0_S0, eos->dRho_dT, eos->dRho_dS};
  a.h = st.in(h, bH); a.T = st.in(T, bH); a.S = st.in(S, bH); a.p_atm = st.in(p_atm, bH2);
  a.PFu = st.inout(PFu, bU); a.PFv = st.inout(PFv, bV); a.pbce = st.inout(pbce, bH); a.eta = st.inout(eta, bH2);
  a.e = (double *)st.scratch(bH + bH2);
  a.T_t = (double *)st.scratch(bH); a.T_b = (double *)st.scratch(bH);
  a.S_t = (double *)st.scratch(bH); a.S_b = (double *)st.scratch(bH);
  a.dpa = (double *)st.scratch(bH); a.intz_dpa = (double *)st.scratch(bH);
  a.za0 = (double *)st.scratch(bH2);
  if (st.failed()) return 1;
  a.rho_ref = cs->Rho0; a.Z_ref = cs->Z_ref; a.GFS_scale = cs->GFS_scale; a.H_to_RZ = H_to_RZ;
  a.Rlay = a.g_prime = nullptr; a.nkmb = 0; a.P_Ref = 0.0;
  a.boundary_extrap = cs->boundary_extrap; a.massw = cs->useMassWghtInterp;
  const int ncol_i = g.iec - g.isc + 3, ncol_j = g.jec - g.jsc + 3;
  hipLaunchKernelGGL(pgfnb_column_kernel, dim3((ncol_i + 63) / 64, ncol_j), dim3(64), 0, s, a);
  hipLaunchKernelGGL(pgfnb_face_kernel, dim3((ncol_i - 1 + 63) / 64, ncol_j - 1), dim3(64), 0, s, a);
  M6_HIP(hipGetLastError());
  return st.finish();
}

// ALE_PLM_edge_values(CS, G, GV, h, Q, bdry_extrap, Q_t, Q_b)            src/ALE/MOM_ALE.F90:1520
extern "C" int mom6hip_ale_plm_edge_values(mom6hip_ctx_t *ctx, const double *h, const double *Q, int32_t bdry_extrap, double *Q_t,
                                           double *Q_b, int32_t memspace) {
  M6_REQUIRE(ctx != nullptr, "ALE_PLM_edge_values: the context is not initialised");
  M6_REQUIRE(h && Q && Q_t && Q_b, "ALE_PLM_edge_values: null argument");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "ALE_PLM_edge_values: bad memspace");
  const m6::GridDev g = ctx->g;
  M6_REQUIRE(g.nk >= 2, "ALE_PLM_edge_values: at least 2 layers are needed");
  M6_REQUIRE(g.isc - g.isd >= 1 && g.ied - g.iec >= 1 && g.jsc - g.jsd >= 1 && g.jed - g.jec >= 1, "ALE_PLM_edge_values: needs a halo of at least 1");
  const size_t bH = (size_t)g.nh3() * 8;
  m6::Stager st(ctx, memspace);
  PlmEdgeArgs a;
  a.g = g; a.h = st.in(h, bH); a.Q = st.in(Q, bH); a.Q_t = st.inout(Q_t, bH); a.Q_b = st.inout(Q_b, bH); a.bdry_extrap = bdry_extrap;
  M6_REQUIRE(!st.failed(), "ALE_PLM_edge_values: staging failed");
  const int ncol_i = g.iec - g.isc + 3, ncol_j = g.jec - g.jsc + 3;
  hipLaunchKernelGGL(plm_edge_kernel, dim3((ncol_i + 63) / 64, ncol_j), dim3(64), 0, ctx->stream, a);
  M6_HIP(hipGetLastError());
  return st.finish();
}
