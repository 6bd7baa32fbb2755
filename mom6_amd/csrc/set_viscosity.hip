// set_viscosity.hip -- set_viscous_BBL (and the early return of set_viscous_ML) of
// src/parameterizations/vertical/MOM_set_viscosity.F90 (:134-1100, :1898-2044) as a gfx950 kernel.
//
// The bottom-boundary-layer thickness and viscosity of a velocity column come from three sweeps through its layers: the
// pressure at the bottom (top down, for the density derivatives), the thickness-weighted speed within HBBL of the bottom
// (bottom up, usually a few layers: every term needs the transverse velocity averaged from four neighbours,
// set_v_at_u / set_u_at_v), and the height to which the turbulent energy 400 u*^2 can mix the stratification (bottom up,
// stops where it is used up).  One lane per face column, lanes along i; the reference's per-row work arrays h_at_vel,
// dz_at_vel, T_vel, S_vel are re-formed from h, T, S where they are read (no per-lane arrays).
// Algorithmic traffic: read u or v, h, T, S at both cells (mostly the bottom layers), write two 2-D fields.
#include <cmath>

#include "common.hpp"
#include "eos.hpp"
#include "cr_math.hpp"

namespace {

using m6::max2;
using m6::min2;
using namespace m6::eos;

struct BBLArgs {
  m6::GridDev g;
  EosDev E;
  double cdrag, drag_bg_vel, Hbbl, dz_bbl, BBL_thick_min, Kv_BBL_min, BBL_thick_max, H_to_RZ;
  double c_Smag, Chan_drag_max_vol, Z_ref;
  int linear_drag, use_BBL_EOS, correct_BBL_bounds, body_force_drag, RiNo_mix, Channel_drag, concave_trig;
  const double *u, *v, *h, *T, *S, *Rlay;      // Rlay: device copy of GV%Rlay
  double *bbl_thick, *Kv_bbl, *Ray;            // of this direction
  // open boundaries (the OBC instantiation only): per face the cell inside a segment (m6::obc_side_maps: -1 the first cell, +1 the
  // second, 0 no segment), and the reference's work arrays D_u, D_v, mask_u, mask_v after its two loops over the segments :374-413
  const int32_t *side_u, *side_v;
  const double *D_u, *D_v, *mask_u, *mask_v;
};

// set_v_at_u :1804-1846 / set_u_at_v :1849-1891: the transverse velocity at the face (i, j) of layer k
template <int DIR, bool OBC = false>
__device__ __forceinline__ double transverse_vel(const BBLArgs &A, int i, int j, long kH, long kU, long kV) {
  const m6::GridDev &g = A.g;
  if (DIR == 0) {      // v at the u point (I = i, j)
    const int J = j;
    double hwt[2][2];
#pragma unroll
    for (int j0 = -1; j0 <= 0; j0++)
#pragma unroll
      for (int i0 = 0; i0 <= 1; i0++) {
        const int i1 = i + i0, J1 = J + j0;
        const long fv = g.v2(i1, J1);
        const double mk = OBC ? A.mask_v[fv] : g.mask2dCv[fv];
        hwt[i0][j0 + 1] = (A.h[g.h2(i1, J1) + kH] + A.h[g.h2(i1, J1 + 1) + kH]) * mk;
        if (OBC) {      // :1829-1838
          const int sd = A.side_v[fv];
          if (sd < 0) hwt[i0][j0 + 1] = 2.0 * A.h[g.h2(i1, J1) + kH] * mk;
          else if (sd > 0) hwt[i0][j0 + 1] = 2.0 * A.h[g.h2(i1, J1 + 1) + kH] * mk;
        }
      }
    const double hwt_tot = (hwt[0][0] + hwt[1][1]) + (hwt[1][0] + hwt[0][1]);
    double r = 0.0;
    if (hwt_tot > 0.0)
      r = ((hwt[0][1] * A.v[g.v2(i, J) + kV] + hwt[1][0] * A.v[g.v2(i + 1, J - 1) + kV]) +
           (hwt[1][1] * A.v[g.v2(i + 1, J) + kV] + hwt[0][0] * A.v[g.v2(i, J - 1) + kV])) / hwt_tot;
    return r;
  } else {      // u at the v point (i, J = j)
    const int I = i;
    double hwt[2][2];
#pragma unroll
    for (int j0 = 0; j0 <= 1; j0++)
#pragma unroll
      for (int i0 = -1; i0 <= 0; i0++) {
        const int I1 = I + i0, j1 = j + j0;
        const long fu = g.u2(I1, j1);
        const double mk = OBC ? A.mask_u[fu] : g.mask2dCu[fu];
        hwt[i0 + 1][j0] = (A.h[g.h2(I1, j1) + kH] + A.h[g.h2(I1 + 1, j1) + kH]) * mk;
        if (OBC) {      // :1874-1883
          const int sd = A.side_u[fu];
          if (sd < 0) hwt[i0 + 1][j0] = 2.0 * A.h[g.h2(I1, j1) + kH] * mk;
          else if (sd > 0) hwt[i0 + 1][j0] = 2.0 * A.h[g.h2(I1 + 1, j1) + kH] * mk;
        }
      }
    const double hwt_tot = (hwt[0][0] + hwt[1][1]) + (hwt[1][0] + hwt[0][1]);
    double r = 0.0;
    if (hwt_tot > 0.0)
      r = ((hwt[1][0] * A.u[g.u2(I, j) + kU] + hwt[0][1] * A.u[g.u2(I - 1, j + 1) + kU]) +
           (hwt[0][0] * A.u[g.u2(I - 1, j) + kU] + hwt[1][1] * A.u[g.u2(I, j + 1) + kU])) / hwt_tot;
    return r;
  }
}


// ---- CHANNEL_DRAG: the normalized open width L(K) of a velocity cell at the interfaces, bottom up (find_L_open_uniform_slope
// :1104, find_L_open_concave_trigonometric :1144, find_L_open_concave_iterative :1237, find_L_open_convex :1640 with
// SET_VISC_ANSWER_DATE >= 20190101).  L(K) depends on the volume below the interface, on L(K+1) and, for convex bottoms, on the
// volume error carried up from the interface below: the sweep keeps these three scalars instead of the reference's arrays.
struct ChanGeom {
  static constexpr double C1_3 = 1.0 / 3.0, C1_6 = 1.0 / 6.0, C1_12 = 1.0 / 12.0, C2pi_3 = 8.0 * 0.78539816339744830962 / 3.0;
  int kind;      // 0 uniform slope, 1 concave (trigonometric), 2 concave (iterative), 3 convex
  double D_vel, Dp, Dm, slope, crv, crv_3;
  // uniform
  double Vol_open, I_slope;
  // concave
  double Vol_2_reg, C24_crv, Iapb, apb_4a, a2x48_apb3, ax2_3apb;
  double L_2_reg, L_inflect_1, vol_inflect_1, vol_inflect_2, slope_crv, smc, C3c_m_s, I_3c_m_s, C4_crv, slope2_4crv, sxcms_c, C3s_m_c, I_3s_m_c,
      Icrvpslope;
  // convex
  double Vol_direct, L_direct, Ibma_2, Vol_err, Angstrom_Z, dZ_subroundoff;

  __device__ void init(double D_vel_, double Dp_, double Dm_, double BBL_thick_min, int concave_trig, double Angstrom_Z_, double dZ_sub) {
    D_vel = D_vel_; Dp = Dp_; Dm = Dm_;
    crv = 3.0 * (Dp + Dm - 2.0 * D_vel);
    slope = Dp - Dm;
    if (fabs(crv) < 1e-2 * (slope + BBL_thick_min)) crv = 0.0;
    Vol_err = 0.0; Angstrom_Z = Angstrom_Z_; dZ_subroundoff = dZ_sub;
    I_3c_m_s = 0.0; I_3s_m_c = 0.0; slope_crv = 0.0; vol_inflect_2 = 0.0; C4_crv = 0.0; slope2_4crv = 0.0; sxcms_c = 0.0; C3s_m_c = 0.0;
    smc = 0.0; C3c_m_s = 0.0;
    if (crv == 0.0) {
      kind = 0;
      slope = fabs(Dp - Dm);
      Vol_open = 0.5 * slope; I_slope = (slope == 0.0) ? 0.0 : 1.0 / slope;
    } else if (crv > 0.0 && concave_trig) {
      kind = 1;
      crv_3 = (Dp + Dm - (2.0 * D_vel)); crv = 3.0 * crv_3;
      if (slope >= crv) {
        Vol_open = D_vel - Dm; Vol_2_reg = Vol_open;
      } else {
        slope_crv = slope / crv;
        Vol_open = 0.25 * slope * slope_crv + C1_12 * crv;
        Vol_2_reg = 0.5 * (slope_crv * slope_crv) * (crv - C1_3 * slope);
      }
      C24_crv = 24.0 / crv; Iapb = 1.0 / (crv + slope);
      apb_4a = (slope + crv) / (4.0 * crv); a2x48_apb3 = (48.0 * (crv * crv)) * ((Iapb * Iapb) * Iapb);
      ax2_3apb = 2.0 * C1_3 * crv * Iapb;
    } else if (crv > 0.0) {
      kind = 2;
      crv_3 = (Dp + Dm - 2.0 * D_vel); crv = 3.0 * crv_3;
      if (slope >= crv) {
        Vol_open = D_vel - Dm; Vol_2_reg = Vol_open;
        L_2_reg = 1.0;
        if (crv + slope >= 4.0 * crv) {
          L_inflect_1 = 1.0; vol_inflect_1 = Vol_open;
        } else {
          slope_crv = slope / crv;
          L_inflect_1 = 0.25 + 0.25 * slope_crv;
          vol_inflect_1 = 0.25 * C1_12 * (((slope_crv + 1.0) * (slope_crv + 1.0)) * (slope + crv));
        }
        smc = slope - crv;
        C3c_m_s = 3.0 * crv - slope;
        if (C3c_m_s > 2.0 * smc) I_3c_m_s = 1.0 / C3c_m_s;
      } else {
        slope_crv = slope / crv;
        Vol_open = 0.25 * slope * slope_crv + C1_12 * crv;
        Vol_2_reg = 0.5 * (slope_crv * slope_crv) * (crv - C1_3 * slope);
        L_2_reg = slope_crv;
        vol_inflect_1 = 0.25 * C1_12 * (((slope_crv + 1.0) * (slope_crv + 1.0)) * (slope + crv));
        L_inflect_1 = 0.25 + 0.25 * slope_crv;
        vol_inflect_2 = 0.25 * slope * slope_crv + 0.125 * crv_3;
        C4_crv = 4.0 / crv;
        slope2_4crv = 0.25 * slope * slope_crv;
        sxcms_c = slope_crv * (crv - slope);
        C3s_m_c = 3.0 * slope - crv;
        if (C3s_m_c > 2.0 * sxcms_c) I_3s_m_c = 1.0 / C3s_m_c;
      }
      Icrvpslope = 1.0 / (crv + slope);
    } else {
      kind = 3;
      crv_3 = (Dp + Dm - 2.0 * D_vel); crv = 3.0 * crv_3;
      Vol_open = D_vel - Dm;
      if (slope >= -crv) {
        Vol_direct = 0.0; L_direct = 0.0; C24_crv = 0.0;
      } else {
        C24_crv = 24.0 / crv;
        L_direct = 1.0 + slope / crv;
        Vol_direct = -C1_6 * crv * ((L_direct * L_direct) * L_direct);
      }
      Ibma_2 = 2.0 / (slope - crv);
    }
  }

  // L(K) from vol_below(K), vol_below(K+1) and L(K+1)
  __device__ double next_L(double volK, double volKp1, double LKp1) {
    if (kind == 0) {
      if (slope == 0.0) return 1.0;
      if (volK >= Vol_open) return 1.0;
      return sqrt(2.0 * volK * I_slope);
    }
    if (kind == 1) {
      if (volK >= Vol_open) return 1.0;
      if (volK < Vol_2_reg) {
        if (a2x48_apb3 * volK < 1e-8) {
          const double L0 = sqrt(2.0 * volK * Iapb);
          return L0 * (1.0 + (ax2_3apb * L0));
        }
        return apb_4a * (1.0 - 2.0 * m6::cr::cr_cos(C1_3 * m6::cr::cr_acos((a2x48_apb3 * volK) - 1.0) - C2pi_3));
      }
      double t = 1.0 - C24_crv * (Vol_open - volK);
      t = max2(-1., min2(1., t));
      return 0.5 - m6::cr::cr_cos(C1_3 * m6::cr::cr_acos(t) - C2pi_3);
    }
    if (kind == 2) {
      const int max_itt = 10;
      double L, L_max, L_min, vol_err, dVol_dL, vol_err_max;
      auto VERR1 = [&](double Lk) { return 0.5 * (Lk * Lk) * (slope + crv * (1.0 - 4.0 * C1_3 * Lk)) - volK; };
      auto VERR2 = [&](double Lk) { return crv_3 * ((Lk * Lk) * (0.75 - 0.5 * Lk)) + (slope2_4crv - volK); };
      if (volK >= Vol_open) return 1.0;
      if (volK < Vol_2_reg) {
        L_max = min2(L_2_reg, 1.0);
        if (volK <= vol_inflect_1) L_max = min2(L_max, L_inflect_1);
        L_min = LKp1;
        if (volK >= vol_inflect_1) L_min = max2(L_min, L_inflect_1);
        if (2.0 * volK * Icrvpslope > L_min * L_min) L_min = sqrt(2.0 * volK * Icrvpslope);
        L = L_min;
        if (volK <= vol_inflect_1) {
          vol_err = VERR1(L);
          if (vol_err < 0.0) {
            dVol_dL = L * (slope + crv * (1.0 - 2.0 * L));
            if (L * dVol_dL > vol_err + L_max * dVol_dL) L = L_max;
            else L = L - (vol_err / dVol_dL);
            for (int itt = 1; itt <= max_itt; itt++) {
              vol_err = VERR1(L);
              dVol_dL = L * (slope + crv * (1.0 - 2.0 * L));
              if (fabs(vol_err) < max2(1.0e-15 * L, 1.0e-25) * dVol_dL) break;
              L = L - (vol_err / dVol_dL);
            }
          }
        } else {
          vol_err = VERR1(L);
          if (vol_err < 0.0) {
            if (slope < crv) {
              if ((L_2_reg - L_min) * C3s_m_c > 2.0 * sxcms_c)
                L_max = (slope_crv * (2.0 * slope) - sqrt(sxcms_c * sxcms_c + 2.0 * C3s_m_c * (Vol_2_reg - volK))) * I_3s_m_c;
              else
                L_max = slope_crv;
            } else {
              if ((1.0 - L_min) * C3c_m_s > 2.0 * smc)
                L_max = (2.0 * crv - sqrt(smc * smc + 2.0 * C3c_m_s * (Vol_open - volK))) * I_3c_m_s;
              else
                L_max = 1.0;
            }
            vol_err_max = VERR1(L_max);
            if ((vol_err_max < fabs(vol_err)) && (L_max < 1.0)) {
              dVol_dL = L_max * (slope + crv * (1.0 - 2.0 * L_max));
              L = max2(L_min, L_max - (vol_err_max / dVol_dL));
            }
            for (int itt = 1; itt <= max_itt; itt++) {
              vol_err = VERR1(L);
              dVol_dL = L * (slope + crv * (1.0 - 2.0 * L));
              if (fabs(vol_err) < max2(1.0e-15 * L, 1.0e-25) * dVol_dL) break;
              L = L - (vol_err / dVol_dL);
            }
          }
        }
        return L;
      }
      if (volK <= vol_inflect_2) {
        L_min = max2(LKp1, L_2_reg);
        if ((4.0 * volK - slope * slope_crv) > (crv + 2.0 * C1_3 * slope) * (L_min * L_min))
          L_min = max2(L_min, sqrt((4.0 * volK - slope * slope_crv) / (crv + 2.0 * C1_3 * slope)));
        L_max = 0.5;
        L = L_min;
        vol_err = crv_3 * (L * L) * (0.75 - 0.5 * L) + (slope2_4crv - volK);
        if (vol_err < 0.0) {
          dVol_dL = 0.5 * crv * (L * (1.0 - L));
          if (L * dVol_dL >= vol_err + L_max * dVol_dL) L = L_max;
          else L = L - (vol_err / dVol_dL);
          for (int itt = 1; itt <= max_itt; itt++) {
            vol_err = VERR2(L);
            dVol_dL = 0.5 * crv * (L * (1.0 - L));
            if (fabs(vol_err) < max2(1.0e-15 * L, 1.0e-25) * dVol_dL) break;
            L = L - (vol_err / dVol_dL);
          }
        }
      } else {
        L_min = max2(LKp1, 0.5);
        L = L_min;
        vol_err = VERR2(L);
        if (vol_err < 0.0) {
          L_max = 1.0 - sqrt((Vol_open - volK) * C4_crv);
          vol_err_max = VERR2(L_max);
          if ((vol_err_max < fabs(vol_err)) && (L_max < 1.0)) {
            dVol_dL = 0.5 * crv * (L_max * (1.0 - L_max));
            L = max2(L_min, L_max - (vol_err_max / dVol_dL));
          }
          for (int itt = 1; itt <= max_itt; itt++) {
            vol_err = VERR2(L);
            dVol_dL = 0.5 * crv * (L * (1.0 - L));
            if (fabs(vol_err) < max2(1.0e-15 * L, 1.0e-25) * dVol_dL) break;
            L = L - (vol_err / dVol_dL);
          }
        }
      }
      return L;
    }
    // convex
    const int maxitt = 20;
    auto VERR = [&](double Lk) { return 0.5 * (Lk * Lk) * (slope + crv_3 * (3.0 - 4.0 * Lk)) - volK; };
    if (volK >= Vol_open) return 1.0;
    if (volK <= Vol_direct) {
      const double x = -0.25 * C24_crv * volK;
      return (x > 0.0) ? m6::cr::cr_pow(x, C1_3) : 0.0;
    }
    double L, L0, Vol_0;
    if (volKp1 + Vol_err <= Vol_direct) { L0 = L_direct; Vol_0 = Vol_direct; }
    else { L0 = LKp1; Vol_0 = volKp1 + Vol_err; }
    const double dV_dL2 = 0.5 * (slope + crv) - crv * L0, dVol = (volK - Vol_0);
    const bool use_L0 = (dVol <= 0.);
    const double Vol_tol = max2(0.5 * Angstrom_Z + dZ_subroundoff, 1e-14 * volK);
    const double Vol_quit = max2(0.9 * Angstrom_Z + dZ_subroundoff, 1e-14 * volK);
    const double curv_tol = Vol_tol * (dV_dL2 * dV_dL2) * (dV_dL2 * Vol_tol - 2.0 * crv * L0 * dVol);
    const bool do_one_L_iter = (crv * crv * ((dVol * dVol) * dVol)) < curv_tol;
    if (use_L0) {
      L = L0;
      Vol_err = VERR(L);
    } else if (do_one_L_iter) {
      L = sqrt(L0 * L0 + dVol / dV_dL2);
      Vol_err = VERR(L);
    } else {
      double L_max, L_min;
      if (dV_dL2 * (1.0 - L0 * L0) < dVol + dV_dL2 * (Vol_open - volK) * Ibma_2)
        L_max = sqrt(1.0 - (Vol_open - volK) * Ibma_2);
      else
        L_max = sqrt(L0 * L0 + dVol / dV_dL2);
      L_min = sqrt(L0 * L0 + dVol / (0.5 * (slope + crv) - crv * L_max));
      const double Vol_err_min = VERR(L_min), Vol_err_max = VERR(L_max);
      if (fabs(Vol_err_min) <= Vol_quit) {
        L = L_min; Vol_err = Vol_err_min;
      } else {
        L = sqrt(((L_min * L_min) * Vol_err_max - (L_max * L_max) * Vol_err_min) / (Vol_err_max - Vol_err_min));
        for (int itt = 1; itt <= maxitt; itt++) {
          Vol_err = VERR(L);
          if (fabs(Vol_err) <= Vol_quit) break;
          L = L - Vol_err / (L * (slope + crv - 2.0 * crv * L));
        }
      }
    }
    return L;
  }
};

template <int DIR, bool OBC>
__global__ __launch_bounds__(64) void set_viscous_bbl_kernel(BBLArgs A) {
  const m6::GridDev &g = A.g;
  const int i = (DIR ? g.isc : g.isc - 1) + blockIdx.x * 64 + threadIdx.x;
  const int j = (DIR ? g.jsc - 1 : g.jsc) + blockIdx.y;
  if (i > g.iec) return;
  const long f2 = DIR ? g.v2(i, j) : g.u2(i, j);
  if (!((DIR ? g.mask2dCv[f2] : g.mask2dCu[f2]) > 0.0)) return;      // do_i
  const int nz = g.nk;
  const long hpl = (long)g.nih * g.njh, upl = (long)(g.nih + 1) * g.njh, vpl = (long)g.nih * (g.njh + 1);
  const long fpl = DIR ? vpl : upl;
  long c0 = g.h2(i, j), c1 = DIR ? g.h2(i, j + 1) : g.h2(i + 1, j);
  const double *vel = DIR ? A.v : A.u;
  const double h_neglect = g.H_subroundoff, dz_neglect = g.dZ_subroundoff;
  // a face of an open-boundary segment: the thicknesses, T and S are those of the cell inside (:502-580); every two-cell mean below
  // is written so that it returns its argument when the two cells are the same one, except h_at_vel / dz_at_vel (at_vel)
  const int side = OBC ? (DIR ? A.side_v[f2] : A.side_u[f2]) : 0;
  if (OBC && side) { if (side < 0) c1 = c0; else c0 = c1; }
  const double Rho0x400_G = 400.0 * (A.H_to_RZ / (1.0 * 1.0 * g.g_Earth));
  const double cdrag_sqrt = sqrt(A.cdrag);
  const double cdrag_sqrt_H = cdrag_sqrt * 1.0 * g.Z_to_H;
  const double cdrag_L_to_H = A.cdrag * 1.0 * g.Z_to_H;
  const bool use_EOS = A.use_BBL_EOS;

  // h_at_vel(k), dz_at_vel(k) :431-470 for the zero-based layer k (dz = GV%H_to_Z*h: Boussinesq thickness_to_dz)
  auto at_vel = [&](int k, double &hat, double &dzat) {
    const double h0 = A.h[c0 + hpl * k], h1 = A.h[c1 + hpl * k];
    const double d0 = g.H_to_Z * h0, d1 = g.H_to_Z * h1;
    if (OBC && side) { hat = h0; dzat = d0; return; }
    if (vel[f2 + fpl * k] * (h1 - h0) >= 0) {
      hat = 2.0 * h0 * h1 / (h0 + h1 + h_neglect);
      dzat = 2.0 * d0 * d1 / (d0 + d1 + dz_neglect);
    } else {
      hat = 0.5 * (h0 + h1);
      dzat = 0.5 * (d0 + d1);
    }
  };
  auto T_vel = [&](int k) { return 0.5 * (A.T[c0 + hpl * k] + A.T[c1 + hpl * k]); };
  auto S_vel = [&](int k) { return 0.5 * (A.S[c0 + hpl * k] + A.S[c1 + hpl * k]); };

  // ---- the near-bottom velocity magnitude and ustar :565-660
  double ustar, umag_avg = 0.0, h_bbl_drag = 0.0, dz_bbl_drag = 0.0, T_EOS = 0.0, S_EOS = 0.0;
  if (use_EOS || A.body_force_drag || !A.linear_drag) {
    double htot_vel = 0.0, hwtot = 0.0, hutot = 0.0, dztot_vel = 0.0, dzwtot = 0.0, Thtot = 0.0, Shtot = 0.0;
    const double u2_bg = A.drag_bg_vel * A.drag_bg_vel;
    for (int k = nz - 1; k >= 0; k--) {
      if (htot_vel >= A.Hbbl) break;
      double hat, dzat;
      at_vel(k, hat, dzat);
      const double hweight = min2(A.Hbbl - htot_vel, hat);
      if (hweight < 1.5 * g.Angstrom_H + h_neglect) continue;
      const double dzweight = min2(A.dz_bbl - dztot_vel, dzat);
      htot_vel = htot_vel + hat;
      hwtot = hwtot + hweight;
      dztot_vel = dztot_vel + dzat;
      dzwtot = dzwtot + dzweight;
      if ((!A.linear_drag) && (hweight >= 0.0)) {
        const double vt = transverse_vel<DIR, OBC>(A, i, j, hpl * k, upl * k, vpl * k);
        const double vn = vel[f2 + fpl * k];
        hutot = hutot + hweight * sqrt(vn * vn + vt * vt + u2_bg);
      }
      if (use_EOS && (hweight >= 0.0)) {
        Thtot = Thtot + hweight * T_vel(k);
        Shtot = Shtot + hweight * S_vel(k);
      }
    }
    double I_hwtot = 0.0; if (hwtot > 0.0) I_hwtot = 1.0 / hwtot;
    if ((hwtot <= 0.0) || A.linear_drag) ustar = cdrag_sqrt_H * A.drag_bg_vel;
    else ustar = cdrag_sqrt_H * hutot / hwtot;
    umag_avg = hutot * I_hwtot;
    h_bbl_drag = hwtot;
    dz_bbl_drag = dzwtot;
    if (use_EOS) {
      if (hwtot > 0.0) { T_EOS = Thtot / hwtot; S_EOS = Shtot / hwtot; }
      else { T_EOS = 0.0; S_EOS = 0.0; }
    }
  } else {
    ustar = cdrag_sqrt_H * A.drag_bg_vel;
  }
  double dR_dT = 0.0, dR_dS = 0.0;
  if (use_EOS) {      // :662-680
    double press = 0.0;
    for (int k = 0; k < nz; k++) press = press + (A.H_to_RZ * g.g_Earth) * (0.5 * (A.h[c0 + hpl * k] + A.h[c1 + hpl * k]));
    eos_density_derivs(A.E, T_EOS, S_EOS, press, dR_dT, dR_dS);
  }
  // ---- the thickness of the bottom boundary layer :682-790
  const double ustarsq = Rho0x400_G * (ustar * ustar);
  double htot = 0.0, dztot = 0.0;
  if (use_EOS) {
    double Thtot = 0.0, Shtot = 0.0, oldfn = 0.0;
    for (int k = nz - 1; k >= 1; k--) {
      double hat, dzat;
      at_vel(k, hat, dzat);
      if (hat <= 0.0) continue;
      const double Tk = T_vel(k), Sk = S_vel(k);
      oldfn = dR_dT * (Thtot - Tk * htot) + dR_dS * (Shtot - Sk * htot);
      if (oldfn >= ustarsq) break;
      const double Dfn = (dR_dT * (Tk - T_vel(k - 1)) + dR_dS * (Sk - S_vel(k - 1))) * (hat + htot);
      double Dh, Ddz;
      if ((oldfn + Dfn) <= ustarsq) {
        Dh = hat;
        Ddz = dzat;
      } else {
        const double frac_used = sqrt((ustarsq - oldfn) / (Dfn));
        Dh = hat * frac_used;
        Ddz = dzat * frac_used;
      }
      htot = htot + Dh;
      dztot = dztot + Ddz;
      Thtot = Thtot + Tk * Dh; Shtot = Shtot + Sk * Dh;
    }
    double hat1, dzat1;
    at_vel(0, hat1, dzat1);
    if ((oldfn < ustarsq) && hat1 > 0.0) {
      if (dR_dT * (Thtot - T_vel(0) * htot) + dR_dS * (Shtot - S_vel(0) * htot) < ustarsq) {
        htot = htot + hat1;
        dztot = dztot + dzat1;
      }
    }
  } else {
    double Rhtot = 0.0;
    for (int k = nz - 1; k >= 1; k--) {      // k = nz .. K2 = 2 of the reference
      const double oldfn = Rhtot - A.Rlay[k] * htot;
      const double Dfn_r = (A.Rlay[k] - A.Rlay[k - 1]);
      if (oldfn >= ustarsq) continue;
      double hat, dzat;
      at_vel(k, hat, dzat);
      const double Dfn = Dfn_r * (hat + htot);
      double Dh, Ddz;
      if ((oldfn + Dfn) <= ustarsq) {
        Dh = hat;
        Ddz = dzat;
      } else {
        const double frac_used = sqrt((ustarsq - oldfn) / (Dfn));
        Dh = hat * frac_used;
        Ddz = dzat * frac_used;
      }
      htot = htot + Dh;
      dztot = dztot + Ddz;
      Rhtot = Rhtot + A.Rlay[k] * Dh;
    }
    if (Rhtot - A.Rlay[0] * htot < ustarsq) {
      double hat1, dzat1;
      at_vel(0, hat1, dzat1);
      htot = htot + hat1;
      dztot = dztot + dzat1;
    }
  }
  // :792-830
  double C2f;
  if (DIR == 0) C2f = g.CoriolisBu[g.q2(i, j - 1)] + g.CoriolisBu[g.q2(i, j)];
  else C2f = g.CoriolisBu[g.q2(i - 1, j)] + g.CoriolisBu[g.q2(i, j)];
  const double u2_bg = A.drag_bg_vel * A.drag_bg_vel;
  double bbl_thick;
  if (A.cdrag * u2_bg <= 0.0) {
    const double ustH = ustar, root = sqrt(0.25 * (ustH * ustH) + (htot * C2f) * (htot * C2f));
    if (dztot * ustH <= (A.BBL_thick_min + dz_neglect) * (0.5 * ustH + root)) bbl_thick = A.BBL_thick_min;
    else bbl_thick = (dztot * ustH) / (0.5 * ustH + root);
  } else {
    bbl_thick = dztot / (0.5 + sqrt(0.25 + htot * htot * C2f * C2f / (ustar * ustar)));
    if (bbl_thick < A.BBL_thick_min) bbl_thick = A.BBL_thick_min;
  }
  double Vol_bbl_chan = bbl_thick;      // :849
  if ((bbl_thick > 0.5 * A.dz_bbl) && (A.RiNo_mix)) bbl_thick = 0.5 * A.dz_bbl;
  if (A.body_force_drag) bbl_thick = dz_bbl_drag;
  double kv_bbl;
  if (A.Channel_drag) {      // :863-1002
    auto D_face = [&](int ii, int jj) {
      if (OBC) return DIR ? A.D_v[g.v2(ii, jj)] : A.D_u[g.u2(ii, jj)];
      return DIR ? (0.5 * (g.bathyT[g.h2(ii, jj)] + g.bathyT[g.h2(ii, jj + 1)]) + A.Z_ref)
                 : (0.5 * (g.bathyT[g.h2(ii, jj)] + g.bathyT[g.h2(ii + 1, jj)]) + A.Z_ref);
    };
    auto mask_face = [&](int ii, int jj) {
      if (OBC) return DIR ? A.mask_v[g.v2(ii, jj)] : A.mask_u[g.u2(ii, jj)];
      return DIR ? g.mask2dCv[g.v2(ii, jj)] : g.mask2dCu[g.u2(ii, jj)];
    };
    const double D_vel = D_face(i, j);
    double tmp = DIR ? mask_face(i + 1, j) * D_face(i + 1, j) : mask_face(i, j + 1) * D_face(i, j + 1);
    double Dp = 2.0 * D_vel * tmp / (D_vel + tmp);
    tmp = DIR ? mask_face(i - 1, j) * D_face(i - 1, j) : mask_face(i, j - 1) * D_face(i, j - 1);
    double Dm = 2.0 * D_vel * tmp / (D_vel + tmp);
    if (Dm > Dp) { tmp = Dp; Dp = Dm; Dm = tmp; }
    ChanGeom geo;
    geo.init(D_vel, Dp, Dm, A.BBL_thick_min, A.concave_trig, g.Angstrom_H * g.H_to_Z, dz_neglect);
    if (A.Chan_drag_max_vol >= 0.0) Vol_bbl_chan = min2(Vol_bbl_chan, A.Chan_drag_max_vol);
    const double Cell_width = DIR ? g.dx_Cv[f2] : g.dy_Cu[f2];
    double BBL_visc_frac = 0.0, vol_Kp1 = 0.0, L_Kp1 = 0.0;
    for (int k = nz - 1; k >= 0; k--) {      // (no porous barriers: por_layer_width = por_face_area = 1)
      const double h0 = A.h[c0 + hpl * k], h1 = A.h[c1 + hpl * k];
      const double dz_vel = 0.5 * (g.H_to_Z * h0 + g.H_to_Z * h1);
      const double vol_K = vol_Kp1 + dz_vel;
      const double L_K = geo.next_L(vol_K, vol_Kp1, L_Kp1);
      double Rayleigh;
      if (L_K > L_Kp1) {
        double BBL_frac;
        if (vol_Kp1 < Vol_bbl_chan) {
          const double q = (1.0 - vol_Kp1 / Vol_bbl_chan);
          BBL_frac = q * q;
          BBL_visc_frac = BBL_visc_frac + BBL_frac * (L_K - L_Kp1);
        } else {
          BBL_frac = 0.0;
        }
        const double cdrag_conv = cdrag_L_to_H;
        const double h_vel_pos = 0.5 * (h0 + h1) + h_neglect;
        const double gam = 1.0 - L_Kp1 / L_K;
        Rayleigh = cdrag_conv * (L_K - L_Kp1) * (1.0 - BBL_frac) *
                   (12.0 * A.c_Smag * h_vel_pos) / (12.0 * A.c_Smag * h_vel_pos +
                                                   cdrag_conv * gam * (1.0 - gam) * (1.0 - 1.5 * gam) * (L_K * L_K) * Cell_width);
      } else {
        Rayleigh = 0.0;
      }
      if (Rayleigh > 0.0) {
        const double vt = transverse_vel<DIR, OBC>(A, i, j, hpl * k, upl * k, vpl * k);
        const double vn = vel[f2 + fpl * k];
        A.Ray[f2 + fpl * k] = Rayleigh * sqrt(vn * vn + vt * vt + u2_bg);
      } else {
        A.Ray[f2 + fpl * k] = 0.0;
      }
      vol_Kp1 = vol_K; L_Kp1 = L_K;
    }
    if (A.correct_BBL_bounds && cdrag_sqrt * ustar * bbl_thick * BBL_visc_frac <= A.Kv_BBL_min) {
      kv_bbl = A.Kv_BBL_min;
      if ((cdrag_sqrt * ustar) * BBL_visc_frac * A.BBL_thick_max > kv_bbl) bbl_thick = kv_bbl / ((cdrag_sqrt * ustar) * BBL_visc_frac);
      else bbl_thick = A.BBL_thick_max;
    } else {
      kv_bbl = (cdrag_sqrt * ustar) * bbl_thick * BBL_visc_frac;
    }
  } else
  // not channel drag :1004-1022
  if (A.correct_BBL_bounds && cdrag_sqrt * ustar * bbl_thick <= A.Kv_BBL_min) {
    kv_bbl = A.Kv_BBL_min;
    if ((cdrag_sqrt * ustar) * A.BBL_thick_max > kv_bbl) bbl_thick = kv_bbl / (cdrag_sqrt * ustar);
    else bbl_thick = A.BBL_thick_max;
  } else {
    kv_bbl = (cdrag_sqrt * ustar) * bbl_thick;
  }
  if (A.body_force_drag) {      // :1024-1046
    if (h_bbl_drag > 0.0) {
      double h_sum = 0.0;
      const double I_hwtot = 1.0 / h_bbl_drag;
      for (int k = nz - 1; k >= 0; k--) {
        double hat, dzat;
        at_vel(k, hat, dzat);
        const double h_bbl_fr = min2(h_bbl_drag - h_sum, hat) * I_hwtot;
        const double cdrag_conv = cdrag_L_to_H;
        A.Ray[f2 + fpl * k] = A.Ray[f2 + fpl * k] + (cdrag_conv * umag_avg) * h_bbl_fr;
        h_sum = h_sum + hat;
        if (h_sum >= h_bbl_drag) break;
      }
      kv_bbl = A.Kv_BBL_min;
    }
  }
  kv_bbl = max2(A.Kv_BBL_min, kv_bbl);
  A.bbl_thick[f2] = bbl_thick;
  if (A.Kv_bbl) A.Kv_bbl[f2] = kv_bbl;
}

// ---- set_viscous_ML, the DYNAMIC_VISCOUS_ML search (:2111-2230 at u points, :2400-2506 at v points) ---------------------------
// One lane per velocity column, lanes along i: a top-down walk that stops where the bulk Richardson number of the water above
// reaches its critical value.  (The reference walks a row of columns together and leaves the loop when all are done; a column's
// result does not depend on its neighbours.)  exp() is the correctly rounded one of cr_math.hpp.
struct MLArgs {
  m6::GridDev g;
  m6::eos::EosDev E;
  const double *u, *v, *h, *T, *S, *taux, *tauy, *ustar, *Rlay;
  double *nkml_visc;
  double dt, H_to_RZ, omega, omega_frac, ustar_min, TKE_decay, bulk_Ri_ML;
  int nkml, use_EOS;
};

template <int DIR>
__global__ __launch_bounds__(64) void set_viscous_ml_kernel(MLArgs A) {
  const m6::GridDev &g = A.g;
  const int i = (DIR ? g.isc : g.isc - 1) + blockIdx.x * 64 + threadIdx.x;
  const int j = (DIR ? g.jsc - 1 : g.jsc) + blockIdx.y;
  if (i > g.iec) return;
  const int nz = g.nk, nkml = A.nkml;
  const long f2 = DIR ? g.v2(i, j) : g.u2(i, j);
  const double mask = DIR ? g.mask2dCv[f2] : g.mask2dCu[f2];
  if (mask < 0.5) { A.nkml_visc[f2] = (double)nkml; return; }
  const long hpl = (long)g.nih * g.njh, upl = (long)(g.nih + 1) * g.njh, vpl = (long)g.nih * (g.njh + 1);
  const long c0 = g.h2(i, j), c1 = DIR ? g.h2(i, j + 1) : g.h2(i + 1, j);
  const double dt_Rho0 = A.dt / A.H_to_RZ;
  const double h_neglect = g.H_subroundoff;
  const double h_tiny = 2.0 * g.Angstrom_H + h_neglect;
  const double g_H_Rho0 = (g.g_Earth * g.H_to_Z) / (g.Rho0);
  double htot = 0.0, Thtot = 0.0, Shtot = 0.0, Rhtot = 0.0, uhtot, vhtot, absf;
  int k_massive = nkml;
  if (DIR == 0) {
    uhtot = dt_Rho0 * A.taux[g.u2(i, j)];
    vhtot = 0.25 * dt_Rho0 * ((A.tauy[g.v2(i, j)] + A.tauy[g.v2(i + 1, j - 1)]) + (A.tauy[g.v2(i, j - 1)] + A.tauy[g.v2(i + 1, j)]));
  } else {
    vhtot = dt_Rho0 * A.tauy[g.v2(i, j)];
    uhtot = 0.25 * dt_Rho0 * ((A.taux[g.u2(i, j)] + A.taux[g.u2(i - 1, j + 1)]) + (A.taux[g.u2(i - 1, j)] + A.taux[g.u2(i, j + 1)]));
  }
  if (A.omega_frac >= 1.0) absf = 2.0 * A.omega;
  else {
    if (DIR == 0) absf = 0.5 * (fabs(g.CoriolisBu[g.q2(i, j)]) + fabs(g.CoriolisBu[g.q2(i, j - 1)]));
    else absf = 0.5 * (fabs(g.CoriolisBu[g.q2(i - 1, j)]) + fabs(g.CoriolisBu[g.q2(i, j)]));
    if (A.omega_frac > 0.0) absf = sqrt(A.omega_frac * 4.0 * (A.omega * A.omega) + (1.0 - A.omega_frac) * (absf * absf));
  }
  const double U_star = max2(A.ustar_min, 0.5 * (g.Z_to_H * A.ustar[c0] + g.Z_to_H * A.ustar[c1]));
  const double Idecay_len_TKE = (absf / U_star) * A.TKE_decay;
  double dR_dT = 0.0, dR_dS = 0.0, result = 0.0;
  bool active = true;
  // the other velocity component's four neighbours of this face
  const long oa = DIR ? g.u2(i - 1, j) : g.v2(i, j), ob = DIR ? g.u2(i, j) : g.v2(i, j - 1);
  const long oc = DIR ? g.u2(i - 1, j + 1) : g.v2(i + 1, j), od = DIR ? g.u2(i, j + 1) : g.v2(i + 1, j - 1);
  const double *own = DIR ? A.v : A.u, *oth = DIR ? A.u : A.v;
  const long ownpl = DIR ? vpl : upl, othpl = DIR ? upl : vpl;
  for (int k = 0; k < nz; k++) {      // k zero-based: layer k+1
    const double ha = A.h[c0 + hpl * k], hb = A.h[c1 + hpl * k];
    const double w_own = own[f2 + ownpl * k];
    const double sa = oth[oa + othpl * k] + oth[ob + othpl * k], sb = oth[oc + othpl * k] + oth[od + othpl * k];
    double Ta = 0., Tb = 0., Sa = 0., Sb = 0.;
    if (A.use_EOS) { Ta = A.T[c0 + hpl * k]; Tb = A.T[c1 + hpl * k]; Sa = A.S[c0 + hpl * k]; Sb = A.S[c1 + hpl * k]; }
    if (k + 1 > nkml) {
      if (A.use_EOS && (k + 1 == nkml + 1)) {
        const double press = (A.H_to_RZ * g.g_Earth) * htot;
        const int k2 = (nkml > 1 ? nkml : 1) - 1;
        const double h2a = A.h[c0 + hpl * k2], h2b = A.h[c1 + hpl * k2];
        const double I_2hlay = 1.0 / (h2a + h2b + h_neglect);
        const double T_EOS = (h2a * A.T[c0 + hpl * k2] + h2b * A.T[c1 + hpl * k2]) * I_2hlay;
        const double S_EOS = (h2a * A.S[c0 + hpl * k2] + h2b * A.S[c1 + hpl * k2]) * I_2hlay;
        m6::eos::eos_density_derivs(A.E, T_EOS, S_EOS, press, dR_dT, dR_dS);
      }
      const double hlay = 0.5 * (ha + hb);
      if (hlay > h_tiny) {
        const double I_2hlay = 1.0 / (ha + hb);
        const double other_at = 0.5 * (ha * sa + hb * sb) * I_2hlay;      // v_at_u :2168 / u_at_v :2449
        double du, dv;
        if (DIR == 0) { du = uhtot - htot * w_own; dv = vhtot - htot * other_at; }
        else { du = uhtot - htot * other_at; dv = vhtot - htot * w_own; }
        const double Uh2 = (du * du + dv * dv);
        double gHprime;
        if (A.use_EOS) {
          const double T_lay = (ha * Ta + hb * Tb) * I_2hlay;
          const double S_lay = (ha * Sa + hb * Sb) * I_2hlay;
          gHprime = g_H_Rho0 * (dR_dT * (T_lay * htot - Thtot) + dR_dS * (S_lay * htot - Shtot));
        } else {
          gHprime = g_H_Rho0 * (A.Rlay[k] * htot - Rhtot);
        }
        if (gHprime > 0.0) {
          const double RiBulk = A.bulk_Ri_ML * m6::cr::cr_exp(-htot * Idecay_len_TKE);
          if (RiBulk * Uh2 <= (htot * htot) * gHprime) {
            result = (double)k_massive;
            active = false;
          } else if (RiBulk * Uh2 <= ((htot + hlay) * (htot + hlay)) * gHprime) {
            result = (double)k + (sqrt(RiBulk * Uh2 / gHprime) - htot) / hlay;      // real(k-1), one-based k
            active = false;
          }
        }
        k_massive = k + 1;
      }
      if (!active) break;
    }
    htot = htot + 0.5 * (ha + hb);
    if (DIR == 0) {
      uhtot = uhtot + 0.5 * (ha + hb) * w_own;
      vhtot = vhtot + 0.25 * (ha * sa + hb * sb);
    } else {
      vhtot = vhtot + 0.5 * (ha + hb) * w_own;
      uhtot = uhtot + 0.25 * (ha * sa + hb * sb);
    }
    if (A.use_EOS) {
      Thtot = Thtot + 0.5 * (ha * Ta + hb * Tb);
      Shtot = Shtot + 0.5 * (ha * Sa + hb * Sb);
    } else {
      Rhtot = Rhtot + 0.5 * (ha + hb) * A.Rlay[k];
    }
  }
  if (active) result = (double)k_massive;
  A.nkml_visc[f2] = result;
}

int check_cs(const mom6hip_set_visc_cs_t *cs, const char *who) {
  static const char *names[9] = {"(free)", "BBL_USE_TIDAL_BG", "(free)", "(free)",
                                 "non-Boussinesq mode (tv%SpV_avg)", "tv%p_surf", "open boundary conditions", "porous barriers",
                                 "ice shelves"};
  M6_REQUIRE(cs->initialized, "%s: Module must be initialized before it is used.", who);
  for (int n = 0; n < 9; n++) M6_REQUIRE(!cs->unsupported[n], "%s: %s is not provided by libmom6hip", who, names[n]);
  return 0;
}

}  // namespace

// the launches on device arrays (also called by bench / the model drivers between steps); Rlay is a host array
namespace m6 {
int set_viscous_BBL_dev(mom6hip_ctx_t *ctx, const mom6hip_set_visc_cs_t *cs, const double *u, const double *v, const double *h,
                        const double *T, const double *S, const mom6hip_eos_t *eos, double *bbl_thick_u, double *bbl_thick_v,
                        double *Kv_bbl_u, double *Kv_bbl_v, double *Ray_u, double *Ray_v, const BBLObcDev *ob) {
  const m6::GridDev g = ctx->g;
  hipStream_t s = ctx->stream;
  const bool use_EOS = (eos != nullptr) && cs->BBL_use_EOS;
  BBLArgs A;
  A.g = g;
  A.E.form = eos ? eos->form : MOM6HIP_EOS_LINEAR; A.E.Rho_T0_S0 = eos ? eos->Rho_T0_S0 : 0.0;
  A.E.dRho_dT = eos ? eos->dRho_dT : 0.0; A.E.dRho_dS = eos ? eos->dRho_dS : 0.0;
  A.cdrag = cs->cdrag; A.drag_bg_vel = cs->drag_bg_vel; A.Hbbl = cs->Hbbl; A.dz_bbl = cs->dz_bbl; A.BBL_thick_min = cs->BBL_thick_min;
  A.Kv_BBL_min = cs->Kv_BBL_min; A.BBL_thick_max = cs->BBL_thick_max; A.H_to_RZ = cs->H_to_RZ;
  A.linear_drag = cs->linear_drag; A.use_BBL_EOS = use_EOS; A.correct_BBL_bounds = cs->correct_BBL_bounds;
  A.body_force_drag = cs->body_force_drag; A.RiNo_mix = cs->RiNo_mix;
  A.Channel_drag = cs->Channel_drag; A.concave_trig = cs->concave_trigonometric_L; A.c_Smag = cs->c_Smag;
  A.Chan_drag_max_vol = cs->Chan_drag_max_vol; A.Z_ref = cs->Z_ref;
  M6_REQUIRE(!(cs->Channel_drag || cs->body_force_drag) || (Ray_u && Ray_v),
             "set_viscous_BBL: CHANNEL_DRAG and DRAG_AS_BODY_FORCE need visc%%Ray_u and visc%%Ray_v");
  A.u = u; A.v = v; A.h = h; A.T = T; A.S = S; A.Rlay = nullptr;
  if (!use_EOS) {
    M6_REQUIRE(ctx->sv_rlay.reserve(sizeof(double) * (size_t)g.nk) == 0, "set_viscous_BBL: out of device memory");
    M6_HIP(hipMemcpyAsync(ctx->sv_rlay.p, cs->Rlay, sizeof(double) * (size_t)g.nk, hipMemcpyHostToDevice, s));
    A.Rlay = (const double *)ctx->sv_rlay.p;
  }
  if (Ray_u) M6_HIP(hipMemsetAsync(Ray_u, 0, sizeof(double) * (size_t)g.nu3(), s));      // :416-417
  if (Ray_v) M6_HIP(hipMemsetAsync(Ray_v, 0, sizeof(double) * (size_t)g.nv3(), s));
  const int ni = g.iec - g.isc + 1, nj = g.jec - g.jsc + 1;
  A.side_u = A.side_v = nullptr; A.D_u = A.D_v = A.mask_u = A.mask_v = nullptr;
  if (ob) { A.side_u = ob->side_u; A.side_v = ob->side_v; A.D_u = ob->D_u; A.D_v = ob->D_v; A.mask_u = ob->mask_u; A.mask_v = ob->mask_v; }
  A.bbl_thick = bbl_thick_u; A.Kv_bbl = Kv_bbl_u; A.Ray = Ray_u;
  if (ob) hipLaunchKernelGGL((set_viscous_bbl_kernel<0, true>), dim3((ni + 1 + 63) / 64, nj), dim3(64), 0, s, A);
  else hipLaunchKernelGGL((set_viscous_bbl_kernel<0, false>), dim3((ni + 1 + 63) / 64, nj), dim3(64), 0, s, A);
  A.bbl_thick = bbl_thick_v; A.Kv_bbl = Kv_bbl_v; A.Ray = Ray_v;
  if (ob) hipLaunchKernelGGL((set_viscous_bbl_kernel<1, true>), dim3((ni + 63) / 64, nj + 1), dim3(64), 0, s, A);
  else hipLaunchKernelGGL((set_viscous_bbl_kernel<1, false>), dim3((ni + 63) / 64, nj + 1), dim3(64), 0, s, A);
  M6_HIP(hipGetLastError());
  if (!use_EOS) M6_HIP(hipStreamSynchronize(s));      // (the host's Rlay may change after the call)
  return 0;
}
}  // namespace m6

// set_viscous_ML on device arrays (also called by the split RK2 step at :592)
namespace m6 {
int set_viscous_ML_dev(mom6hip_ctx_t *ctx, const mom6hip_set_visc_cs_t *cs, const double *u, const double *v, const double *h,
                       const double *T, const double *S, const mom6hip_eos_t *eos, const double *taux, const double *tauy,
                       const double *ustar, double *nkml_visc_u, double *nkml_visc_v, double dt) {
  const m6::GridDev g = ctx->g;
  M6_REQUIRE(g.mask2dCu && g.mask2dCv && g.CoriolisBu, "set_viscous_ML: mask2dCu, mask2dCv and CoriolisBu are needed");
  M6_REQUIRE(g.isc - g.isd >= 1 && g.jsc - g.jsd >= 1, "set_viscous_ML: needs a halo of at least 1");
  MLArgs A;
  A.g = g;
  A.use_EOS = eos != nullptr;
  A.E.form = eos ? eos->form : MOM6HIP_EOS_LINEAR; A.E.Rho_T0_S0 = eos ? eos->Rho_T0_S0 : 0.0;
  A.E.dRho_dT = eos ? eos->dRho_dT : 0.0; A.E.dRho_dS = eos ? eos->dRho_dS : 0.0;
  A.u = u; A.v = v; A.h = h; A.T = T; A.S = S; A.taux = taux; A.tauy = tauy; A.ustar = ustar; A.Rlay = nullptr;
  if (!A.use_EOS) {
    M6_REQUIRE(cs->Rlay, "set_viscous_ML: GV%%Rlay is needed without an equation of state");
    A.Rlay = ctx->tables[m6::TABLE_SVML_RLAY].get(cs->Rlay, (size_t)g.nk);
    if (!A.Rlay) return 1;
  }
  A.dt = dt; A.H_to_RZ = cs->H_to_RZ; A.omega = cs->omega; A.omega_frac = cs->omega_frac; A.ustar_min = cs->ustar_min;
  A.TKE_decay = cs->TKE_decay; A.bulk_Ri_ML = cs->bulk_Ri_ML; A.nkml = cs->nkml;
  const int ni = g.iec - g.isc + 1, nj = g.jec - g.jsc + 1;
  A.nkml_visc = nkml_visc_u;
  hipLaunchKernelGGL(set_viscous_ml_kernel<0>, dim3((ni + 1 + 63) / 64, nj), dim3(64), 0, ctx->stream, A);
  A.nkml_visc = nkml_visc_v;
  hipLaunchKernelGGL(set_viscous_ml_kernel<1>, dim3((ni + 63) / 64, nj + 1), dim3(64), 0, ctx->stream, A);
  M6_HIP(hipGetLastError());
  return 0;
}
}  // namespace m6

extern "C" int mom6hip_set_viscous_ml(mom6hip_ctx_t *ctx, const mom6hip_set_visc_cs_t *cs, const double *u, const double *v,
                                      const double *h, const double *T, const double *S, const mom6hip_eos_t *eos, const double *taux,
                                      const double *tauy, const mom6hip_vertvisc_type_t *visc, double dt, int32_t memspace) {
  M6_REQUIRE(ctx != nullptr && cs != nullptr, "MOM_set_viscosity(visc_ML): null argument");
  if (check_cs(cs, "MOM_set_viscosity(visc_ML)")) return 1;
  if (!cs->dynamic_viscous_ML) return 0;      // :2043-2044: nothing to do without DYNAMIC_VISCOUS_ML or an ice shelf
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "set_viscous_ML: bad memspace");
  M6_REQUIRE(u && v && h && taux && tauy && visc, "set_viscous_ML: null argument");
  M6_REQUIRE(visc->ustar && visc->nkml_visc_u && visc->nkml_visc_v,
             "set_viscous_ML: forces%%ustar (visc->ustar) and visc%%nkml_visc_u / nkml_visc_v must be allocated with DYNAMIC_VISCOUS_ML");
  M6_REQUIRE(cs->nkml >= 0 && cs->nkml < ctx->g.nk, "set_viscous_ML: GV%%nkml must be in 0 .. nk-1");
  M6_REQUIRE(!eos || (T && S), "set_viscous_ML: an equation of state needs tv%%T and tv%%S");
  const m6::GridDev g = ctx->g;
  const size_t bH = sizeof(double) * (size_t)g.nh3(), bU = sizeof(double) * (size_t)g.nu3(), bV = sizeof(double) * (size_t)g.nv3();
  const size_t bH2 = sizeof(double) * (size_t)g.nih * g.njh, bU2 = sizeof(double) * (size_t)(g.nih + 1) * g.njh,
               bV2 = sizeof(double) * (size_t)g.nih * (g.njh + 1);
  m6::Stager st(ctx, memspace);
  const double *du = st.in(u, bU), *dv = st.in(v, bV), *dh = st.in(h, bH), *dT = eos ? st.in(T, bH) : nullptr, *dS = eos ? st.in(S, bH) : nullptr;
  const double *dtx = st.in(taux, bU2), *dty = st.in(tauy, bV2), *dus = st.in(visc->ustar, bH2);
  double *nu = st.inout((double *)visc->nkml_visc_u, bU2), *nv = st.inout((double *)visc->nkml_visc_v, bV2);
  M6_REQUIRE(!st.failed(), "set_viscous_ML: staging failed");
  if (int rc = m6::set_viscous_ML_dev(ctx, cs, du, dv, dh, dT, dS, eos, dtx, dty, dus, nu, nv, dt)) return rc;
  return st.finish();
}

namespace {
// the reference's work arrays D_u, mask_u (DIR 0) or D_v, mask_v (DIR 1) before its loops over the segments :363-372, over the data domain
template <int DIR>
__global__ __launch_bounds__(256) void bbl_obc_fill_kernel(m6::GridDev g, double Z_ref, double *D, double *mask) {
  const int i = (DIR ? g.isd : g.isd - 1) + blockIdx.x * 256 + threadIdx.x, j = (DIR ? g.jsd - 1 : g.jsd) + blockIdx.y;
  if (i > g.ied) return;
  const long f2 = DIR ? g.v2(i, j) : g.u2(i, j);
  const bool inside = DIR ? (j >= g.jsd && j + 1 <= g.jed) : (i >= g.isd && i + 1 <= g.ied);
  D[f2] = inside ? 0.5 * (g.bathyT[g.h2(i, j)] + g.bathyT[DIR ? g.h2(i, j + 1) : g.h2(i + 1, j)]) + Z_ref : 0.0;
  mask[f2] = DIR ? g.mask2dCv[f2] : g.mask2dCu[f2];
}

struct BBLSeg { int ew, plus, A, c0, c1; };      // a segment: E/W or N/S, the cell inside is the first (E, N: plus) or the second, its face index, the range along it

// phase 0 :374-389: the depth of the segment's own faces from the cell inside; phase 1 :390-413: the depths and masks of the faces of
// the other direction just outside, across the segment's corner points
__global__ __launch_bounds__(64) void bbl_obc_segment_kernel(m6::GridDev g, BBLSeg S, int phase, double Z_ref, double *D_u, double *D_v, double *mask_u,
                                                             double *mask_v) {
  const int c = S.c0 + blockIdx.x * 64 + threadIdx.x;
  if (c > S.c1) return;
  if (phase == 0) {
    if (S.ew) D_u[g.u2(S.A, c)] = g.bathyT[g.h2(S.plus ? S.A : S.A + 1, c)] + Z_ref;
    else D_v[g.v2(c, S.A)] = g.bathyT[g.h2(c, S.plus ? S.A : S.A + 1)] + Z_ref;
  } else if (S.ew) {      // c = J: D_v(i+1,J) = D_v(i,J) (E) | D_v(i,J) = D_v(i+1,J) (W), i = the segment's I
    const long in = g.v2(S.plus ? S.A : S.A + 1, c), out = g.v2(S.plus ? S.A + 1 : S.A, c);
    D_v[out] = D_v[in]; mask_v[out] = 0.0;
  } else {                // c = I: D_u(I,j+1) = D_u(I,j) (N) | D_u(I,j) = D_u(I,j+1) (S), j = the segment's J
    const long in = g.u2(c, S.plus ? S.A : S.A + 1), out = g.u2(c, S.plus ? S.A + 1 : S.A);
    D_u[out] = D_u[in]; mask_u[out] = 0.0;
  }
}
}  // namespace

extern "C" int mom6hip_set_viscous_bbl(mom6hip_ctx_t *ctx, const mom6hip_set_visc_cs_t *cs, const double *u, const double *v,
                                       const double *h, const double *T, const double *S, const mom6hip_eos_t *eos,
                                       const mom6hip_vertvisc_type_t *visc, int32_t memspace) {
  return mom6hip_set_viscous_bbl_obc(ctx, cs, u, v, h, T, S, eos, visc, nullptr, memspace);
}

// set_viscous_BBL with CS%OBC associated: the depths and masks of the faces at and beside the segments :374-413, the zero-gradient
// projection of the thicknesses, T and S :502-580, the weights of set_v_at_u / set_u_at_v :1829-1838, :1874-1883
extern "C" int mom6hip_set_viscous_bbl_obc(mom6hip_ctx_t *ctx, const mom6hip_set_visc_cs_t *cs, const double *u, const double *v,
                                           const double *h, const double *T, const double *S, const mom6hip_eos_t *eos,
                                           const mom6hip_vertvisc_type_t *visc, const mom6hip_obc_t *obc, int32_t memspace) {
  M6_REQUIRE(ctx != nullptr, "MOM_set_viscosity(BBL): Module must be initialized before it is used.");
  M6_REQUIRE(cs && u && v && h && visc, "set_viscous_BBL: null argument");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "set_viscous_BBL: bad memspace");
  if (check_cs(cs, "MOM_set_viscosity(BBL)")) return 1;
  if (!cs->bottomdraglaw) return 0;      // :321
  const bool use_EOS = (eos != nullptr) && cs->BBL_use_EOS;
  M6_REQUIRE(!use_EOS || (T && S), "set_viscous_BBL: BBL_USE_EOS needs tv%%T and tv%%S");
  M6_REQUIRE(use_EOS || cs->Rlay, "set_viscous_BBL: GV%%Rlay is required without BBL_USE_EOS");
  M6_REQUIRE(!eos || eos->form == MOM6HIP_EOS_LINEAR || eos->form == MOM6HIP_EOS_WRIGHT || eos->form == MOM6HIP_EOS_UNESCO ||
                 eos->form == MOM6HIP_EOS_WRIGHT_FULL || eos->form == MOM6HIP_EOS_WRIGHT_REDUCED,
             "set_viscous_BBL: this equation of state is not provided");
  M6_REQUIRE(visc->bbl_thick_u && visc->bbl_thick_v, "set_viscous_BBL: visc%%bbl_thick_u and visc%%bbl_thick_v are required");
  M6_REQUIRE(!cs->body_force_drag || (visc->Ray_u && visc->Ray_v), "set_viscous_BBL: DRAG_AS_BODY_FORCE needs visc%%Ray_u and visc%%Ray_v");
  const m6::GridDev g = ctx->g;
  M6_REQUIRE(g.mask2dCu && g.mask2dCv && g.CoriolisBu, "set_viscous_BBL: mask2dCu, mask2dCv and CoriolisBu are required");
  M6_REQUIRE(g.isc - g.isd >= 1 && g.jsc - g.jsd >= 1, "set_viscous_BBL: the halo must be at least 1 point wide");
  const size_t bH = sizeof(double) * (size_t)g.nh3(), bU = sizeof(double) * (size_t)g.nu3(), bV = sizeof(double) * (size_t)g.nv3();
  const size_t bU2 = sizeof(double) * (size_t)(g.nih + 1) * g.njh, bV2 = sizeof(double) * (size_t)g.nih * (g.njh + 1);
  m6::Stager st(ctx, memspace);
  const double *du = st.in(u, bU), *dv = st.in(v, bV), *dh = st.in(h, bH), *dT = st.in(T, bH), *dS = st.in(S, bH);
  double *btu = st.inout((double *)visc->bbl_thick_u, bU2), *btv = st.inout((double *)visc->bbl_thick_v, bV2);
  double *kvu = st.inout((double *)visc->Kv_bbl_u, bU2), *kvv = st.inout((double *)visc->Kv_bbl_v, bV2);
  double *ru = st.inout((double *)visc->Ray_u, bU), *rv = st.inout((double *)visc->Ray_v, bV);
  M6_REQUIRE(!st.failed(), "set_viscous_BBL: staging failed");
  m6::BBLObcDev ob;
  const bool with_obc = obc != nullptr;      // (associated(OBC): the branches are taken whatever the number of segments)
  if (with_obc) {
    M6_REQUIRE(obc->number_of_segments == 0 || obc->segment, "set_viscous_BBL: OBC%%segment is required");
    if (m6::obc_side_maps(ctx, st, obc, &ob.side_u, &ob.side_v, "set_viscous_BBL")) return 1;
    const size_t nU2 = (size_t)(g.nih + 1) * g.njh, nV2 = (size_t)g.nih * (g.njh + 1);
    if (!ob.side_u) {      // no segments: maps of zeros
      int32_t *z = (int32_t *)st.scratch(4 * (nU2 + nV2));
      M6_REQUIRE(!st.failed() && z, "set_viscous_BBL: out of device memory");
      M6_HIP(hipMemsetAsync(z, 0, 4 * (nU2 + nV2), ctx->stream));
      ob.side_u = z; ob.side_v = z + nU2;
    }
    double *w = (double *)st.scratch(8 * 2 * (nU2 + nV2));
    M6_REQUIRE(!st.failed() && w, "set_viscous_BBL: out of device memory");
    double *D_u = w, *mask_u = w + nU2, *D_v = mask_u + nU2, *mask_v = D_v + nV2;
    hipLaunchKernelGGL(bbl_obc_fill_kernel<0>, dim3((g.nih + 1 + 255) / 256, g.njh), dim3(256), 0, ctx->stream, g, cs->Z_ref, D_u, mask_u);
    hipLaunchKernelGGL(bbl_obc_fill_kernel<1>, dim3((g.nih + 255) / 256, g.njh + 1), dim3(256), 0, ctx->stream, g, cs->Z_ref, D_v, mask_v);
    const int is = g.isc, ie = g.iec, js = g.jsc, je = g.jec;
    for (int phase = 0; phase < 2; phase++) for (int n = 0; n < obc->number_of_segments; n++) {
      const mom6hip_obc_segment_t &Sg = obc->segment[n];
      if (!Sg.on_pe) continue;
      BBLSeg d;
      d.plus = (Sg.direction == MOM6HIP_OBC_DIRECTION_E || Sg.direction == MOM6HIP_OBC_DIRECTION_N) ? 1 : 0;
      const bool known = d.plus || Sg.direction == MOM6HIP_OBC_DIRECTION_W || Sg.direction == MOM6HIP_OBC_DIRECTION_S;
      if (Sg.is_N_or_S && Sg.JsdB >= js - 1 && Sg.JsdB <= je) {
        d.ew = 0; d.A = Sg.JsdB;
        M6_REQUIRE(Sg.JsdB >= g.jsd && Sg.JsdB + 1 <= g.jed, "set_viscous_BBL: OBC segment %d lies outside the data domain", n + 1);
        if (phase == 0) { d.c0 = std::max(is - 1, Sg.isd); d.c1 = std::min(ie + 1, Sg.ied); }
        else { d.c0 = std::max(is - 1, Sg.IsdB); d.c1 = std::min(ie, Sg.IedB); }
      } else if (Sg.is_E_or_W && Sg.IsdB >= is - 1 && Sg.IsdB <= ie) {
        d.ew = 1; d.A = Sg.IsdB;
        M6_REQUIRE(Sg.IsdB >= g.isd && Sg.IsdB + 1 <= g.ied, "set_viscous_BBL: OBC segment %d lies outside the data domain", n + 1);
        if (phase == 0) { d.c0 = std::max(js - 1, Sg.jsd); d.c1 = std::min(je + 1, Sg.jed); }
        else { d.c0 = std::max(js - 1, Sg.JsdB); d.c1 = std::min(je, Sg.JedB); }
      } else continue;
      // (a segment whose direction does not go with its orientation changes nothing: the reference's tests on %direction)
      if (!known || (d.ew != (Sg.direction == MOM6HIP_OBC_DIRECTION_E || Sg.direction == MOM6HIP_OBC_DIRECTION_W))) continue;
      if (d.c1 < d.c0) continue;
      hipLaunchKernelGGL(bbl_obc_segment_kernel, dim3((d.c1 - d.c0 + 1 + 63) / 64), dim3(64), 0, ctx->stream, g, d, phase, cs->Z_ref, D_u, D_v, mask_u, mask_v);
    }
    M6_HIP(hipGetLastError());
    ob.D_u = D_u; ob.D_v = D_v; ob.mask_u = mask_u; ob.mask_v = mask_v;
  }
  if (m6::set_viscous_BBL_dev(ctx, cs, du, dv, dh, dT, dS, eos, btu, btv, kvu, kvv, ru, rv, with_obc ? &ob : nullptr)) return 1;
  return st.finish();
}
