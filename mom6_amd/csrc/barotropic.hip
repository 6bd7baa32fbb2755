// barotropic.hip -- MOM_barotropic on MI355X: btstep, btcalc, bt_mass_source, set_dtbt, barotropic_init.
//
// Reference: src/core/MOM_barotropic.F90 (btstep :423-2797, set_dtbt :2801, btcalc :3394, find_uhbt :3683,
// set_local_BT_cont_types :3949, adjust_local_BT_cont_types :4085, find_face_areas :4221, bt_mass_source :4318,
// barotropic_init :4376).  Provided branch: Boussinesq, no OBC, no SAL, BTHALO = 0, the reference defaults for the
// switches in mom6hip_barotropic_cs_t.unsupported.
//
// Layout of the work: the 3-D inputs are read exactly once by two "face-column" kernels (one lane per u- or v-point,
// marching k with i-contiguous wave loads; all the vertical sums of the reference's setup are accumulated in
// registers in its k order), the subcycle runs on 2-D arrays with four small kernels per barotropic step on the
// shrinking valid range of the reference's wide-halo march (one group pass every `num_cycles` steps), and the
// per-layer accelerations are written by one more face-column kernel.  The transport of the instantaneous velocity
// that the next step's predictor needs (find_uhbt(ubt)+uhbt0, :1884) is produced by the velocity kernel that already
// holds the BT_cont fit of that face in registers, and travels with eta/ubt/vbt in the group pass; it is the same
// pure function of (ubt, fit, uhbt0) that the reference evaluates after the pass.
#include "common.hpp"
#include "cr_math.hpp"

#include <algorithm>
#include <string>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <initializer_list>
#include <utility>

namespace m6 {

}

namespace {

constexpr double SUBROUNDOFF = 1e-30;   // MOM_barotropic.F90:413

using m6::cr::cr_pow;

// ---- launch helper: one thread per point of an inclusive index range, i fastest ------------------------------------
template <class F> __global__ void __launch_bounds__(256) range2d_kernel(int i0, int i1, int j0, int j1, F f) {
  const int i = i0 + blockIdx.x * 64 + threadIdx.x, j = j0 + blockIdx.y * 4 + threadIdx.y;
  if (i <= i1 && j <= j1) f(i, j);
}
template <class F> void launch2d(hipStream_t st, int i0, int i1, int j0, int j1, F f) {
  if (i1 < i0 || j1 < j0) return;
  dim3 grid((i1 - i0 + 64) / 64, (j1 - j0 + 4) / 4), block(64, 4);
  hipLaunchKernelGGL(range2d_kernel<F>, grid, block, 0, st, i0, i1, j0, j1, f);
}

// ---- the BT_cont fit at a face: local_BT_cont_u_type / _v_type as a structure of arrays (:335-372) -------------------
struct Btcl { double *FA_EE, *FA_E0, *FA_W0, *FA_WW, *uBT_WW, *uBT_EE, *uh_crvW, *uh_crvE, *uh_WW, *uh_EE; };

// find_uhbt :3683 / find_vhbt :3817
__device__ __forceinline__ double find_uhbt(double u, const Btcl &B, long n) {
  if (u == 0.0) return 0.0;
  const double ee = B.uBT_EE[n];
  if (u < ee) return (u - ee) * B.FA_EE[n] + B.uh_EE[n];
  if (u < 0.0) return u * (B.FA_E0[n] + B.uh_crvE[n] * (u * u));
  const double ww = B.uBT_WW[n];
  if (u <= ww) return u * (B.FA_W0[n] + B.uh_crvW[n] * (u * u));
  return (u - ww) * B.FA_WW[n] + B.uh_WW[n];
}

// every 2-D work array of btstep (device pointers into one scratch block)
struct Work {
  // u-points
  double *ubt, *bt_rem_u, *BT_force_u, *u_accel_bt, *uhbt, *uhbt0, *uhbtp, *azon, *bzon, *czon, *dzon, *Cor_ref_u,
      *DCor_u, *Datu, *ubt_Cor, *ubt0, *uhbtS, *amer, *bmer, *cmer, *dmer;
  // v-points
  double *vbt, *bt_rem_v, *BT_force_v, *v_accel_bt, *vhbt, *vhbt0, *vhbtp, *Cor_ref_v, *DCor_v, *Datv, *vbt_Cor, *vbt0,
      *vhbtS;
  // h-points
  double *eta, *eta_pred, *eta_sum, *eta_wtd, *eta_PF, *eta_PF_1, *d_eta_PF, *gtot_E, *gtot_W, *gtot_N, *gtot_S,
      *eta_src, *e_anom;
  double *q;
  Btcl BU, BV;
  // open boundaries (null without): the code of every face of a segment (OB_*, from OBC%segnum_u / segnum_v and the segments' flags), the
  // arrays of BT_OBC_type :71-101 that set_up_BT_OBC fills, the velocities before the time step and before the first one, and the
  // time-filtered velocity of the segments' faces
  const int32_t *obc_u, *obc_v;
  double *ob_Cg_u, *ob_dZ_u, *ob_uhbt, *ob_ubt_outer, *ob_SSH_u, *ubt_old, *ubt_first, *ubt_wtd;
  double *ob_Cg_v, *ob_dZ_v, *ob_vhbt, *ob_vbt_outer, *ob_SSH_v, *vbt_old, *vbt_first, *vbt_wtd;
};
enum { OB_SPEC = 1, OB_FLATHER = 2, OB_GRAD = 4, OB_PLUS = 8, OB_MINUS = 16 };      // PLUS: OBC_DIRECTION_E | N (the cell inside is the first one)

// uhbt_to_ubt :3733 / vhbt_to_vbt :3866
__device__ double uhbt_to_ubt(double uhbt, const Btcl &B, long n) {
  const double tol = 1.0e-10;
  const int max_itt = 20;
  double ubt, ubt_min, ubt_max, uherr_min, uherr_max;
  if (uhbt == 0.0) {
    ubt = 0.0;
  } else if (uhbt < B.uh_EE[n]) {
    ubt = B.uBT_EE[n] + (uhbt - B.uh_EE[n]) / B.FA_EE[n];
  } else if (uhbt < 0.0) {
    ubt_min = B.uBT_EE[n]; uherr_min = B.uh_EE[n] - uhbt;
    ubt_max = 0.0; uherr_max = -uhbt;
    ubt = B.uBT_EE[n] * (uhbt / B.uh_EE[n]);
    for (int itt = 1; itt <= max_itt; itt++) {
      const double uhbt_err = ubt * (B.FA_E0[n] + B.uh_crvE[n] * (ubt * ubt)) - uhbt;
      if (fabs(uhbt_err) < tol * fabs(uhbt)) break;
      if (uhbt_err > 0.0) { ubt_max = ubt; uherr_max = uhbt_err; }
      if (uhbt_err < 0.0) { ubt_min = ubt; uherr_min = uhbt_err; }
      const double derr_du = B.FA_E0[n] + 3.0 * B.uh_crvE[n] * (ubt * ubt);
      if ((uhbt_err >= derr_du * (ubt - ubt_min)) || (-uhbt_err >= derr_du * (ubt_max - ubt)) || (derr_du <= 0.0)) {
        ubt = ubt_max + (ubt_min - ubt_max) * (uherr_max / (uherr_max - uherr_min));
      } else {
        ubt = ubt - uhbt_err / derr_du;
        if (fabs(uhbt_err) < (0.01 * tol) * fabs(ubt_min * derr_du)) break;
      }
    }
  } else if (uhbt <= B.uh_WW[n]) {
    ubt_min = 0.0; uherr_min = -uhbt;
    ubt_max = B.uBT_WW[n]; uherr_max = B.uh_WW[n] - uhbt;
    ubt = B.uBT_WW[n] * (uhbt / B.uh_WW[n]);
    for (int itt = 1; itt <= max_itt; itt++) {
      const double uhbt_err = ubt * (B.FA_W0[n] + B.uh_crvW[n] * (ubt * ubt)) - uhbt;
      if (fabs(uhbt_err) < tol * fabs(uhbt)) break;
      if (uhbt_err > 0.0) { ubt_max = ubt; uherr_max = uhbt_err; }
      if (uhbt_err < 0.0) { ubt_min = ubt; uherr_min = uhbt_err; }
      const double derr_du = B.FA_W0[n] + 3.0 * B.uh_crvW[n] * (ubt * ubt);
      if ((uhbt_err >= derr_du * (ubt - ubt_min)) || (-uhbt_err >= derr_du * (ubt_max - ubt)) || (derr_du <= 0.0)) {
        ubt = ubt_min + (ubt_max - ubt_min) * (-uherr_min / (uherr_max - uherr_min));
      } else {
        ubt = ubt - uhbt_err / derr_du;
        if (fabs(uhbt_err) < (0.01 * tol) * (ubt_max * derr_du)) break;
      }
    }
  } else {
    ubt = B.uBT_WW[n] + (uhbt - B.uh_WW[n]) / B.FA_WW[n];
  }
  return ubt;
}

// scalars of one btstep call
struct Par {
  double dtbt, Instep, dgeo_de, vel_underflow, trans_wt1, trans_wt2, RZ_to_H;
  int project_velocity;      // BT_PROJECT_VELOCITY: the pressure force reads eta, not eta_pred (:1751); no predictor continuity (:1870)
  int nstep, use_BT_cont, interp_eta_PF, add_uh0, strong_drag, visc_rem_u_uh0, find_etaav, have_bot;
};

// ---- face-column kernel: all the vertical sums of the setup (:1035-1372, :1505-1541) --------------------------------
// DIR 0: u-points (I = is-1..ie, j = js..je); DIR 1: v-points (i = is..ie, J = js-1..je).
template <int DIR>
__global__ void __launch_bounds__(256)
bt_pre_face_kernel(m6::GridDev g, Work w, Par p, const double *__restrict__ frhat, const double *__restrict__ visc_rem,
                   const double *__restrict__ vel_Cor, const double *__restrict__ pbce, const double *__restrict__ uh0,
                   const double *__restrict__ u_uh0, const double *__restrict__ vel_in, const double *__restrict__ bc_accel,
                   const double *__restrict__ tau, const double *__restrict__ tau_bot, const double *__restrict__ IDat) {
  const int i = (DIR ? g.isc : g.isc - 1) + blockIdx.x * 64 + threadIdx.x;
  const int j = (DIR ? g.jsc - 1 : g.jsc) + blockIdx.y * 4 + threadIdx.y;
  if (i > g.iec || j > g.jec) return;
  const long f2 = DIR ? g.v2(i, j) : g.u2(i, j);
  const long fstr = DIR ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh;
  const long hm = g.h2(i, j), hp = DIR ? g.h2(i, j + 1) : g.h2(i + 1, j), hstr = (long)g.nih * g.njh;
  const double mask = DIR ? g.mask2dCv[f2] : g.mask2dCu[f2];

  double vel_cor = 0.0, gt_m = 0.0, gt_p = 0.0, uhS = 0.0, u0 = 0.0, ub = 0.0, av_rem = 0.0, force;
  if (mask > 0.0) {
    force = tau[f2] * p.RZ_to_H * IDat[f2] * visc_rem[f2];
    if (p.have_bot) force = force - tau_bot[f2] * p.RZ_to_H * IDat[f2];
  } else {
    force = 0.0;
  }
  for (int k = 0; k < g.nk; k++) {
    const long f3 = f2 + fstr * k;
    const double fr = frhat[f3], vr_in = visc_rem[f3];
    double vr = m6::min2(vr_in, 1.);
    vr = m6::max2(vr, 1. - 0.5 * p.Instep / (vr + SUBROUNDOFF));
    vr = m6::max2(vr, 0.);
    const double wt = fr * vr;
    vel_cor = vel_cor + wt * vel_Cor[f3];
    gt_m = gt_m + pbce[hm + hstr * k] * wt;
    gt_p = gt_p + pbce[hp + hstr * k] * wt;
    if (p.add_uh0) {
      uhS = uhS + uh0[f3];
      u0 = u0 + (p.visc_rem_u_uh0 ? wt : fr) * u_uh0[f3];
    }
    ub = ub + wt * vel_in[f3];
    force = force + wt * bc_accel[f3];
    av_rem = av_rem + fr * vr_in;
  }
  if (fabs(ub) < p.vel_underflow) ub = 0.0;
  double rem;
  if (p.strong_drag) {
    rem = mask * ((p.nstep * av_rem) / (1.0 + (p.nstep - 1) * av_rem));
  } else {
    rem = 0.0;
    if (mask * av_rem > 0.0) rem = mask * cr_pow(av_rem, p.Instep);
  }
  if (DIR == 0) {
    w.ubt_Cor[f2] = vel_cor; w.gtot_E[hm] = gt_m; w.gtot_W[hp] = gt_p; w.uhbtS[f2] = uhS; w.ubt0[f2] = u0;
    w.ubt[f2] = ub; w.BT_force_u[f2] = force; w.bt_rem_u[f2] = rem;
  } else {
    w.vbt_Cor[f2] = vel_cor; w.gtot_N[hm] = gt_m; w.gtot_S[hp] = gt_p; w.vhbtS[f2] = uhS; w.vbt0[f2] = u0;
    w.vbt[f2] = ub; w.BT_force_v[f2] = force; w.bt_rem_v[f2] = rem;
  }
}

// ---- btcalc with the face thicknesses given (:3665-3676): frhat = h_u * mask / (sum_k h_u + h_neglect) -----------------
// The lane keeps its column in registers between the sum and the scaling, so h_u is read once (the two-loop form reads it twice:
// 2.8 GB a launch at 1440x1080x75, this one 1.9).  NKMAX bounds the unrolled loops; the layer count is uniform.
template <int DIR, int NKMAX>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8)))
btcalc_given_kernel(m6::GridDev g, const double *__restrict__ hw, double *__restrict__ fr, double h_neglect) {
  const int i = (DIR ? g.isc : g.isc - 1) + blockIdx.x * 64 + threadIdx.x;
  const int j = (DIR ? g.jsc - 1 : g.jsc) + blockIdx.y * 4 + threadIdx.y;
  if (i > g.iec || j > g.jec) return;
  const long f2 = DIR ? g.v2(i, j) : g.u2(i, j);
  const long fstr = DIR ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh;
  const double mask = DIR ? g.mask2dCv[f2] : g.mask2dCu[f2];
  const int nz = g.nk;
  double v[NKMAX];
  const double *__restrict__ src = hw + f2;
#pragma unroll
  for (int k = 0; k < NKMAX; k++) { v[k] = (k < nz) ? *src : 0.0; src += fstr; }
  double tot = v[0];
#pragma unroll
  for (int k = 1; k < NKMAX; k++) if (k < nz) tot = tot + v[k];
  const double Ihat = mask / (tot + h_neglect);
  double *__restrict__ dst = fr + f2;
#pragma unroll
  for (int k = 0; k < NKMAX; k++) { if (k < nz) *dst = v[k] * Ihat; dst += fstr; }
}

// ---- per-layer accelerations (:2576-2589) ------------------------------------------------------------------------
template <int DIR>
__global__ void __launch_bounds__(256)
bt_accel_layer_kernel(m6::GridDev g, Work w, const double *__restrict__ pbce, double *__restrict__ accel, double accel_underflow) {
  const int i = (DIR ? g.isc : g.isc - 1) + blockIdx.x * 64 + threadIdx.x;
  const int j = (DIR ? g.jsc - 1 : g.jsc) + blockIdx.y * 4 + threadIdx.y;
  if (i > g.iec || j > g.jec) return;
  const long f2 = DIR ? g.v2(i, j) : g.u2(i, j);
  const long fstr = DIR ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh;
  const long hm = g.h2(i, j), hp = DIR ? g.h2(i, j + 1) : g.h2(i + 1, j), hstr = (long)g.nih * g.njh;
  const double abt = DIR ? w.v_accel_bt[f2] : w.u_accel_bt[f2];
  const double gp = DIR ? w.gtot_S[hp] : w.gtot_W[hp], gm = DIR ? w.gtot_N[hm] : w.gtot_E[hm];
  const double ep = w.e_anom[hp], em = w.e_anom[hm];
  const double Id = DIR ? g.IdyCv[f2] : g.IdxCu[f2];
  for (int k = 0; k < g.nk; k++) {
    double a = (abt - ((pbce[hp + hstr * k] - gp) * ep - (pbce[hm + hstr * k] - gm) * em) * Id);
    if (fabs(a) < accel_underflow) a = 0.0;
    accel[f2 + fstr * k] = a;
  }
}

// ---- the four kernels of one barotropic step ---------------------------------------------------------------------
// predictor continuity (:1882-1909) + eta_sum (:1921)
__global__ void __launch_bounds__(256)
bt_eta_pred_kernel(m6::GridDev g, Work w, Par p, int i0, int i1, int j0, int j1, double wt_accel2) {
  const int i = i0 + blockIdx.x * 64 + threadIdx.x, j = j0 + blockIdx.y * 4 + threadIdx.y;
  if (i > i1 || j > j1) return;
  const long n = g.h2(i, j);
  double ep;
  if (p.project_velocity) {
    ep = w.eta[n];
  } else {
    ep = (w.eta[n] + w.eta_src[n]) + (p.dtbt * g.IareaT[n]) *
        ((w.uhbtp[g.u2(i - 1, j)] - w.uhbtp[g.u2(i, j)]) + (w.vhbtp[g.v2(i, j - 1)] - w.vhbtp[g.v2(i, j)]));
    w.eta_pred[n] = ep;
  }
  if (p.find_etaav && i >= g.isc && i <= g.iec && j >= g.jsc && j <= g.jec) w.eta_sum[n] = w.eta_sum[n] + wt_accel2 * ep;
}

// corrector continuity (:2414-2421)
__global__ void __launch_bounds__(256)
bt_eta_kernel(m6::GridDev g, Work w, Par p, int i0, int i1, int j0, int j1, double wt_eta) {
  const int i = i0 + blockIdx.x * 64 + threadIdx.x, j = j0 + blockIdx.y * 4 + threadIdx.y;
  if (i > i1 || j > j1) return;
  const long n = g.h2(i, j);
  const double e = (w.eta[n] + w.eta_src[n]) + (p.dtbt * g.IareaT[n]) *
      ((w.uhbt[g.u2(i - 1, j)] - w.uhbt[g.u2(i, j)]) + (w.vhbt[g.v2(i, j - 1)] - w.vhbt[g.v2(i, j)]));
  w.eta[n] = e;
  w.eta_wtd[n] = w.eta_wtd[n] + e * wt_eta;
}

__device__ __forceinline__ double eta_pf_at(const Work &w, const Par &p, long n, double wt_end) {
  return p.interp_eta_PF ? (w.eta_PF_1[n] + wt_end * w.d_eta_PF[n]) : w.eta_PF[n];
}

// meridional velocity update (:1975-2044 when v goes first, :2217-2290 when second) + running sums (:2349-2354)
__global__ void __launch_bounds__(256)
bt_vbt_kernel(m6::GridDev g, Work w, Par p, int i0, int i1, int j0, int j1, double wt_accel, double wt_trans, double wt_end,
              double *__restrict__ vbt_sum, double *__restrict__ vhbt_sum) {
  const int i = i0 + blockIdx.x * 64 + threadIdx.x, J = j0 + blockIdx.y * 4 + threadIdx.y;
  if (i > i1 || J > j1) return;
  const int j = J;
  const long n = g.v2(i, J);
  if (w.obc_v && w.obc_v[n]) return;      // a face of a segment keeps its velocity and transport (:2043-2048, :2287-2292): bt_obc_kernel sets them
  const double Cor_v = -1.0 * ((w.amer[g.u2(i - 1, j)] * w.ubt[g.u2(i - 1, j)] + w.cmer[g.u2(i, j + 1)] * w.ubt[g.u2(i, j + 1)]) +
                               (w.bmer[g.u2(i, j)] * w.ubt[g.u2(i, j)] + w.dmer[g.u2(i - 1, j + 1)] * w.ubt[g.u2(i - 1, j + 1)])) -
                       w.Cor_ref_v[n];
  const long hs = g.h2(i, j), hn = g.h2(i, j + 1);
  const double *epb = p.project_velocity ? w.eta : w.eta_pred;
  const double PFv = ((epb[hs] - eta_pf_at(w, p, hs, wt_end)) * w.gtot_N[hs] -
                      (epb[hn] - eta_pf_at(w, p, hn, wt_end)) * w.gtot_S[hn]) * p.dgeo_de * g.IdyCv[n];
  const double vel_prev = w.vbt[n];
  double vb = w.bt_rem_v[n] * (vel_prev + p.dtbt * ((w.BT_force_v[n] + Cor_v) + PFv));
  if (fabs(vb) < p.vel_underflow) vb = 0.0;
  const double vtrans = p.trans_wt1 * vb + p.trans_wt2 * vel_prev;
  w.vbt[n] = vb;
  w.v_accel_bt[n] = w.v_accel_bt[n] + wt_accel * (Cor_v + PFv);
  const double vh0 = w.vhbt0[n];
  double vh, vhp;
  if (p.use_BT_cont) { vh = find_uhbt(vtrans, w.BV, n) + vh0; vhp = find_uhbt(vb, w.BV, n) + vh0; }
  else { const double D = w.Datv[n]; vh = D * vtrans + vh0; vhp = D * vb + vh0; }
  w.vhbt[n] = vh; w.vhbtp[n] = vhp;
  if (i >= g.isc && i <= g.iec && J >= g.jsc - 1 && J <= g.jec) {
    vbt_sum[n] = vbt_sum[n] + wt_trans * vtrans;
    vhbt_sum[n] = vhbt_sum[n] + wt_trans * vh;
  }
}

// zonal velocity update (:2047-2127 when second, :2130-2207 when first) + running sums (:2342-2347)
__global__ void __launch_bounds__(256)
bt_ubt_kernel(m6::GridDev g, Work w, Par p, int i0, int i1, int j0, int j1, double wt_accel, double wt_trans, double wt_end,
              double *__restrict__ ubt_sum, double *__restrict__ uhbt_sum) {
  const int I = i0 + blockIdx.x * 64 + threadIdx.x, j = j0 + blockIdx.y * 4 + threadIdx.y;
  if (I > i1 || j > j1) return;
  const int i = I;
  const long n = g.u2(I, j);
  if (w.obc_u && w.obc_u[n]) return;      // (:2121-2126, :2198-2203)
  const double Cor_u = ((w.azon[n] * w.vbt[g.v2(i + 1, j)] + w.czon[n] * w.vbt[g.v2(i, j - 1)]) +
                        (w.bzon[n] * w.vbt[g.v2(i, j)] + w.dzon[n] * w.vbt[g.v2(i + 1, j - 1)])) - w.Cor_ref_u[n];
  const long hw = g.h2(i, j), he = g.h2(i + 1, j);
  const double *epb = p.project_velocity ? w.eta : w.eta_pred;
  const double PFu = ((epb[hw] - eta_pf_at(w, p, hw, wt_end)) * w.gtot_E[hw] -
                      (epb[he] - eta_pf_at(w, p, he, wt_end)) * w.gtot_W[he]) * p.dgeo_de * g.IdxCu[n];
  const double vel_prev = w.ubt[n];
  double ub = w.bt_rem_u[n] * (vel_prev + p.dtbt * ((w.BT_force_u[n] + Cor_u) + PFu));
  if (fabs(ub) < p.vel_underflow) ub = 0.0;
  const double utrans = p.trans_wt1 * ub + p.trans_wt2 * vel_prev;
  w.ubt[n] = ub;
  w.u_accel_bt[n] = w.u_accel_bt[n] + wt_accel * (Cor_u + PFu);
  const double uh0 = w.uhbt0[n];
  double uh, uhp;
  if (p.use_BT_cont) { uh = find_uhbt(utrans, w.BU, n) + uh0; uhp = find_uhbt(ub, w.BU, n) + uh0; }
  else { const double D = w.Datu[n]; uh = D * utrans + uh0; uhp = D * ub + uh0; }
  w.uhbt[n] = uh; w.uhbtp[n] = uhp;
  if (I >= g.isc - 1 && I <= g.iec && j >= g.jsc && j <= g.jec) {
    ubt_sum[n] = ubt_sum[n] + wt_trans * utrans;
    uhbt_sum[n] = uhbt_sum[n] + wt_trans * uh;
  }
}

// ---- one barotropic step as ONE kernel with LDS-staged halo tiles (round 5) -----------------------------------------
// The four kernels above pass eta_pred, the new velocity of the first direction and the two corrector transports through HBM
// and read eta, eta_src and IareaT twice.  Here a block of 256 threads owns a tile of BTF_TX x BTF_TY cells of the step's
// cell space [isv-1, iev+1] x [jsv-1, jev+1] (the predictor's range) with the faces on their east and north sides, and works
// through the step with the intermediates in LDS:
//   1. eta_pred (:1882-1909) and D = eta_pred - eta_PF (the bracket of the pressure force, :1975 ...) on the tile + a rim of 1;
//   2. the velocity of the first direction (:1975-2044 | :2130-2207) on the tile's faces + a rim of 1 (the Coriolis term of
//      the second direction reads it at four points, the corrector reads its transport);
//   3. the velocity of the second direction (:2047-2127 | :2217-2290) on the tile's faces + the one column | row of faces on
//      its low side that the corrector of the tile's cells reads;
//   4. the corrector continuity (:2414-2421) on the tile's cells.
// Rim values are recomputed by the neighbouring tile from the same inputs (the same bits); only the owner stores and
// accumulates.  A tile reads the OLD eta, ubt, vbt, uhbtp, vhbtp of its rim while its neighbours write their new values, so
// these five fields alternate between two sets of arrays from step to step (in / out); every point a step reads was
// written by the step before it or refreshed by the group pass between them (the wide-halo march, :1842-1861), so nothing
// has to be copied through.  Not taken with open boundaries or NONLINEAR_BT_CONTINUITY (the four kernels stay).
#ifndef BTF_TY_DEF
#define BTF_TY_DEF 16
#endif
constexpr int BTF_TX = 64, BTF_TY = BTF_TY_DEF, BTF_W = BTF_TX + 2, BTF_H = BTF_TY + 2;
struct BtFused {
  const double *eta_in, *ubt_in, *vbt_in, *uhbtp_in, *vhbtp_in;
  double *eta_out, *ubt_out, *vbt_out, *uhbtp_out, *vhbtp_out;
  int isv, iev, jsv, jev;
  double wt_accel, wt_accel2, wt_trans, wt_eta, wt_end;
  double *ubt_sum, *uhbt_sum, *vbt_sum, *vhbt_sum;
};

template <bool VFIRST>
__global__ void __launch_bounds__(256) bt_step_fused_kernel(m6::GridDev g, Work w, Par p, BtFused a) {
  __shared__ double sD[BTF_H][BTF_W];       // eta_pred (or eta with BT_PROJECT_VELOCITY) minus eta_PF at the tile's cells + rim
  __shared__ double sV1[BTF_H][BTF_W];      // the new velocity of the first direction at its faces
  __shared__ double sH1[BTF_H][BTF_W];      // its corrector transport (uhbt | vhbt)
  __shared__ double sH2[BTF_H][BTF_W];      // the corrector transport of the second direction
  const int tid = threadIdx.y * 64 + threadIdx.x;
  const int ti0 = a.isv - 1 + blockIdx.x * BTF_TX, tj0 = a.jsv - 1 + blockIdx.y * BTF_TY;      // the tile's first cell
  const int ib = ti0 - 1, jb = tj0 - 1;     // cell / face (ib + li, jb + lj) sits at [lj][li]
  const int ti1 = min(ti0 + BTF_TX - 1, a.iev + 1), tj1 = min(tj0 + BTF_TY - 1, a.jev + 1);      // its last cell
  // ---- 1. the predictor continuity on the tile and a rim of one cell
  for (int idx = tid; idx < BTF_W * BTF_H; idx += 256) {
    const int li = idx % BTF_W, lj = idx / BTF_W, i = ib + li, j = jb + lj;
    double d = 0.0;
    if (i >= a.isv - 1 && i <= a.iev + 1 && j >= a.jsv - 1 && j <= a.jev + 1 && i <= ti1 + 1 && j <= tj1 + 1) {
      const long n = g.h2(i, j);
      double ep;
      if (p.project_velocity) {
        ep = a.eta_in[n];
      } else {
        ep = (a.eta_in[n] + w.eta_src[n]) + (p.dtbt * g.IareaT[n]) *
            ((a.uhbtp_in[g.u2(i - 1, j)] - a.uhbtp_in[g.u2(i, j)]) + (a.vhbtp_in[g.v2(i, j - 1)] - a.vhbtp_in[g.v2(i, j)]));
      }
      const bool own = li >= 1 && lj >= 1 && i <= ti1 && j <= tj1;
      if (own) {
        if (!p.project_velocity) w.eta_pred[n] = ep;
        if (p.find_etaav && i >= g.isc && i <= g.iec && j >= g.jsc && j <= g.jec) w.eta_sum[n] = w.eta_sum[n] + a.wt_accel2 * ep;
      }
      d = ep - eta_pf_at(w, p, n, a.wt_end);
    }
    sD[lj][li] = d;
  }
  __syncthreads();
  // ---- 2. + 3. the two velocity updates: pass 0 the first direction (old velocity of the other one from memory), pass 1 the second
  // (the new velocity of the first from LDS)
#pragma unroll
  for (int ps = 0; ps < 2; ps++) {
    const bool do_v = (ps == 0) ? VFIRST : !VFIRST;
    for (int idx = tid; idx < BTF_W * BTF_H; idx += 256) {
      const int li = idx % BTF_W, lj = idx / BTF_W, i = ib + li, j = jb + lj;
      if (do_v) {
        // faces (i, J = j): first direction i in [isv-1, iev+1], second i in [isv, iev]; J in [jsv-1, jev]
        const int i0 = VFIRST ? a.isv - 1 : a.isv, i1 = VFIRST ? a.iev + 1 : a.iev;
        const bool in_tile = VFIRST ? (i <= ti1 + 1 && j <= tj1) : (li >= 1 && i <= ti1 && j <= tj1);
        if (!(i >= i0 && i <= i1 && j >= a.jsv - 1 && j <= a.jev && in_tile)) continue;
        const long n = g.v2(i, j);
        double u00, u01, u10, u11;      // ubt at (I-1, j), (I, j), (I-1, j+1), (I, j+1)
        if (VFIRST) {
          u00 = a.ubt_in[g.u2(i - 1, j)]; u01 = a.ubt_in[g.u2(i, j)]; u10 = a.ubt_in[g.u2(i - 1, j + 1)]; u11 = a.ubt_in[g.u2(i, j + 1)];
        } else {
          u00 = sV1[lj][li - 1]; u01 = sV1[lj][li]; u10 = sV1[lj + 1][li - 1]; u11 = sV1[lj + 1][li];
        }
        const double Cor_v = -1.0 * ((w.amer[g.u2(i - 1, j)] * u00 + w.cmer[g.u2(i, j + 1)] * u11) +
                                     (w.bmer[g.u2(i, j)] * u01 + w.dmer[g.u2(i - 1, j + 1)] * u10)) - w.Cor_ref_v[n];
        const long hs = g.h2(i, j), hn = g.h2(i, j + 1);
        const double PFv = (sD[lj][li] * w.gtot_N[hs] - sD[lj + 1][li] * w.gtot_S[hn]) * p.dgeo_de * g.IdyCv[n];
        const double vel_prev = a.vbt_in[n];
        double vb = w.bt_rem_v[n] * (vel_prev + p.dtbt * ((w.BT_force_v[n] + Cor_v) + PFv));
        if (fabs(vb) < p.vel_underflow) vb = 0.0;
        const double vtrans = p.trans_wt1 * vb + p.trans_wt2 * vel_prev;
        const double vh0 = w.vhbt0[n];
        double vh, vhp;
        if (p.use_BT_cont) { vh = find_uhbt(vtrans, w.BV, n) + vh0; vhp = find_uhbt(vb, w.BV, n) + vh0; }
        else { const double Dv = w.Datv[n]; vh = Dv * vtrans + vh0; vhp = Dv * vb + vh0; }
        if (VFIRST) { sV1[lj][li] = vb; sH1[lj][li] = vh; } else { sH2[lj][li] = vh; }
        if (li >= 1 && lj >= 1 && i <= ti1 && j <= tj1) {      // the owner
          a.vbt_out[n] = vb;
          w.v_accel_bt[n] = w.v_accel_bt[n] + a.wt_accel * (Cor_v + PFv);
          w.vhbt[n] = vh; a.vhbtp_out[n] = vhp;
          if (i >= g.isc && i <= g.iec && j >= g.jsc - 1 && j <= g.jec) {
            a.vbt_sum[n] = a.vbt_sum[n] + a.wt_trans * vtrans;
            a.vhbt_sum[n] = a.vhbt_sum[n] + a.wt_trans * vh;
          }
        }
      } else {
        // faces (I = i, j): I in [isv-1, iev]; first direction j in [jsv-1, jev+1], second j in [jsv, jev]
        const int j0 = VFIRST ? a.jsv : a.jsv - 1, j1 = VFIRST ? a.jev : a.jev + 1;
        const bool in_tile = VFIRST ? (lj >= 1 && i <= ti1 && j <= tj1) : (i <= ti1 && j <= tj1 + 1);
        if (!(i >= a.isv - 1 && i <= a.iev && j >= j0 && j <= j1 && in_tile)) continue;
        const long n = g.u2(i, j);
        double v00, v01, v10, v11;      // vbt at (i, J-1), (i+1, J-1), (i, J), (i+1, J)
        if (VFIRST) {
          v00 = sV1[lj - 1][li]; v01 = sV1[lj - 1][li + 1]; v10 = sV1[lj][li]; v11 = sV1[lj][li + 1];
        } else {
          v00 = a.vbt_in[g.v2(i, j - 1)]; v01 = a.vbt_in[g.v2(i + 1, j - 1)]; v10 = a.vbt_in[g.v2(i, j)]; v11 = a.vbt_in[g.v2(i + 1, j)];
        }
        const double Cor_u = ((w.azon[n] * v11 + w.czon[n] * v00) + (w.bzon[n] * v10 + w.dzon[n] * v01)) - w.Cor_ref_u[n];
        const long hw = g.h2(i, j), he = g.h2(i + 1, j);
        const double PFu = (sD[lj][li] * w.gtot_E[hw] - sD[lj][li + 1] * w.gtot_W[he]) * p.dgeo_de * g.IdxCu[n];
        const double vel_prev = a.ubt_in[n];
        double ub = w.bt_rem_u[n] * (vel_prev + p.dtbt * ((w.BT_force_u[n] + Cor_u) + PFu));
        if (fabs(ub) < p.vel_underflow) ub = 0.0;
        const double utrans = p.trans_wt1 * ub + p.trans_wt2 * vel_prev;
        const double uh0 = w.uhbt0[n];
        double uh, uhp;
        if (p.use_BT_cont) { uh = find_uhbt(utrans, w.BU, n) + uh0; uhp = find_uhbt(ub, w.BU, n) + uh0; }
        else { const double Du = w.Datu[n]; uh = Du * utrans + uh0; uhp = Du * ub + uh0; }
        if (!VFIRST) { sV1[lj][li] = ub; sH1[lj][li] = uh; } else { sH2[lj][li] = uh; }
        if (li >= 1 && lj >= 1 && i <= ti1 && j <= tj1) {      // the owner
          a.ubt_out[n] = ub;
          w.u_accel_bt[n] = w.u_accel_bt[n] + a.wt_accel * (Cor_u + PFu);
          w.uhbt[n] = uh; a.uhbtp_out[n] = uhp;
          if (i >= g.isc - 1 && i <= g.iec && j >= g.jsc && j <= g.jec) {
            a.ubt_sum[n] = a.ubt_sum[n] + a.wt_trans * utrans;
            a.uhbt_sum[n] = a.uhbt_sum[n] + a.wt_trans * uh;
          }
        }
      }
    }
    __syncthreads();
  }
  // ---- 4. the corrector continuity on the tile's cells
  for (int idx = tid; idx < BTF_TX * BTF_TY; idx += 256) {
    const int li = 1 + idx % BTF_TX, lj = 1 + idx / BTF_TX, i = ib + li, j = jb + lj;
    if (!(i >= a.isv && i <= a.iev && j >= a.jsv && j <= a.jev)) continue;
    const long n = g.h2(i, j);
    const double (*sU)[BTF_W] = VFIRST ? sH2 : sH1;      // uhbt at [lj][I - ib]
    const double (*sV)[BTF_W] = VFIRST ? sH1 : sH2;      // vhbt at [J - jb][li]
    const double e = (a.eta_in[n] + w.eta_src[n]) + (p.dtbt * g.IareaT[n]) *
        ((sU[lj][li - 1] - sU[lj][li]) + (sV[lj - 1][li] - sV[lj][li]));
    a.eta_out[n] = e;
    w.eta_wtd[n] = w.eta_wtd[n] + e * a.wt_eta;
  }
}

// apply_velocity_OBCs :2931-3168 for the faces of the segments of one direction (halo = iev - ie: the window of this time step), with the
// running sums from the values the faces had before the step (:2367-2395) and the predictor transport of the next step (:1882-1893)
template <int DIR>
__global__ void __launch_bounds__(256)
bt_obc_kernel(m6::GridDev g, Work w, Par p, int i0, int i1, int j0, int j1, double bebt, double wt_trans, double wt_vel,
              double *__restrict__ xbt_sum, double *__restrict__ xhbt_sum) {
  const int i = i0 + blockIdx.x * 64 + threadIdx.x, j = j0 + blockIdx.y * 4 + threadIdx.y;
  if (i > i1 || j > j1) return;
  const long f = DIR ? g.v2(i, j) : g.u2(i, j);
  const int code = DIR ? w.obc_v[f] : w.obc_u[f];
  if (!code) return;
  double *xbt = DIR ? w.vbt : w.ubt, *xhbt = DIR ? w.vhbt : w.uhbt, *xhbtp = DIR ? w.vhbtp : w.uhbtp, *xwtd = DIR ? w.vbt_wtd : w.ubt_wtd;
  const double *xold = DIR ? w.vbt_old : w.ubt_old, *o_hbt = DIR ? w.ob_vhbt : w.ob_uhbt, *o_outer = DIR ? w.ob_vbt_outer : w.ob_ubt_outer,
               *o_dZ = DIR ? w.ob_dZ_v : w.ob_dZ_u, *o_Cg = DIR ? w.ob_Cg_v : w.ob_Cg_u, *o_SSH = DIR ? w.ob_SSH_v : w.ob_SSH_u;
  const double Idx = DIR ? g.IdyCv[f] : g.IdxCu[f], Dat = DIR ? w.Datv[f] : w.Datu[f], xhbt0 = DIR ? w.vhbt0[f] : w.uhbt0[f];
  const Btcl &B = DIR ? w.BV : w.BU;
  const long df = DIR ? g.nih : 1, dh = DIR ? g.nih : 1;      // one face | one cell along the direction
  const long c0 = g.h2(i, j);                                 // the first cell of the face
  double vel_trans = 0.0, vb = xbt[f];
  if (code & OB_SPEC) {
    xhbt[f] = o_hbt[f];
    vb = o_outer[f];
    vel_trans = vb;
  } else if (code & OB_PLUS) {
    if (code & OB_FLATHER) {
      const double cfl = p.dtbt * o_Cg[f] * Idx;
      const double u_inlet = cfl * xold[f - df] + (1.0 - cfl) * xold[f];
      const double ssh_in = g.H_to_Z * (w.eta[c0] + (0.5 - cfl) * (w.eta[c0] - w.eta[c0 - dh]));
      if (o_dZ[f] > 0.0) {
        const double vel_prev = vb;
        vb = 0.5 * ((u_inlet + o_outer[f]) + (o_Cg[f] / o_dZ[f]) * (ssh_in - o_SSH[f]));
        vel_trans = (1.0 - bebt) * vel_prev + bebt * vb;
      } else { vb = 0.0; vel_trans = 0.0; }
    } else {      // gradient
      vb = xbt[f - df];
      vel_trans = vb;
    }
  } else {
    if (code & OB_FLATHER) {
      const double cfl = p.dtbt * o_Cg[f] * Idx;
      const double u_inlet = cfl * xold[f + df] + (1.0 - cfl) * xold[f];
      const double ssh_in = g.H_to_Z * (w.eta[c0 + dh] + (0.5 - cfl) * (w.eta[c0 + dh] - w.eta[c0 + 2 * dh]));
      if (o_dZ[f] > 0.0) {
        const double vel_prev = vb;
        vb = 0.5 * ((u_inlet + o_outer[f]) + (o_Cg[f] / o_dZ[f]) * (o_SSH[f] - ssh_in));
        vel_trans = (1.0 - bebt) * vel_prev + bebt * vb;
      } else { vb = 0.0; vel_trans = 0.0; }
    } else {
      vb = xbt[f + df];
      vel_trans = vb;
    }
  }
  xbt[f] = vb;
  double xh = xhbt[f];
  if (!(code & OB_SPEC)) {
    xh = (p.use_BT_cont ? find_uhbt(vel_trans, B, f) : Dat * vel_trans) + xhbt0;
    xhbt[f] = xh;
  }
  xhbtp[f] = (p.use_BT_cont ? find_uhbt(vb, B, f) : Dat * vb) + xhbt0;
  const bool own = DIR ? (i >= g.isc && i <= g.iec && j >= g.jsc - 1 && j <= g.jec) : (i >= g.isc - 1 && i <= g.iec && j >= g.jsc && j <= g.jec);
  if (own) {
    xbt_sum[f] = xbt_sum[f] + wt_trans * vel_trans;
    xhbt_sum[f] = xhbt_sum[f] + wt_trans * xh;
    xwtd[f] = xwtd[f] + wt_vel * vb;
  }
}

// ---- set_dtbt: per-cell stability limit + min ---------------------------------------------------------------------
__global__ void __launch_bounds__(256)
bt_dtbt_kernel(m6::GridDev g, const double *__restrict__ pbce, const double *__restrict__ frhatu, const double *__restrict__ frhatv,
               const double *__restrict__ Datu, const double *__restrict__ Datv, double gtot_est, double bebt, double cor_scale,
               unsigned long long *__restrict__ result) {
  const int i = g.isc + blockIdx.x * 64 + threadIdx.x, j = g.jsc + blockIdx.y * 4 + threadIdx.y;
  double cand = 1.0e38;
  if (i <= g.iec && j <= g.jec) {
    const long n = g.h2(i, j);
    double gE = 0.0, gW = 0.0, gN = 0.0, gS = 0.0;
    if (pbce) {
      const long hstr = (long)g.nih * g.njh, ustr = (long)(g.nih + 1) * g.njh, vstr = (long)g.nih * (g.njh + 1);
      for (int k = 0; k < g.nk; k++) {
        const double pb = pbce[n + hstr * k];
        gE = gE + pb * frhatu[g.u2(i, j) + ustr * k];
        gW = gW + pb * frhatu[g.u2(i - 1, j) + ustr * k];
        gN = gN + pb * frhatv[g.v2(i, j) + vstr * k];
        gS = gS + pb * frhatv[g.v2(i, j - 1) + vstr * k];
      }
    } else {
      gE = gW = gN = gS = gtot_est;
    }
    const double f00 = g.CoriolisBu[g.q2(i, j)], f11 = g.CoriolisBu[g.q2(i - 1, j - 1)], f10 = g.CoriolisBu[g.q2(i - 1, j)],
                 f01 = g.CoriolisBu[g.q2(i, j - 1)];
    const double Idt_max2 = 0.5 * (1.0 + 2.0 * bebt) * (g.IareaT[n] *
        ((gE * Datu[g.u2(i, j)] * g.IdxCu[g.u2(i, j)] + gW * Datu[g.u2(i - 1, j)] * g.IdxCu[g.u2(i - 1, j)]) +
         (gN * Datv[g.v2(i, j)] * g.IdyCv[g.v2(i, j)] + gS * Datv[g.v2(i, j - 1)] * g.IdyCv[g.v2(i, j - 1)])) +
        ((f00 * f00 + f11 * f11) + (f10 * f10 + f01 * f01)) * (cor_scale * cor_scale));
    if (Idt_max2 * 1.0e38 > 1.0) cand = 1.0 / Idt_max2;
  }
  // positive doubles order like their bit patterns: atomicMin on the bits is the exact minimum
  __shared__ unsigned long long smin;
  if (threadIdx.x == 0 && threadIdx.y == 0) smin = 0x7FF0000000000000ull;
  __syncthreads();
  atomicMin(&smin, (unsigned long long)__double_as_longlong(cand));
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) atomicMin(result, smin);
}

int check_cs(const mom6hip_barotropic_cs_t *cs, const char *who) {
  M6_REQUIRE(cs != nullptr, "%s: null control structure", who);
  static const char *names[12] = {"INTEGRAL_BT_CONTINUITY", "(free slot)", "NONLINEAR_BT_CONTINUITY", "BOUND_BT_CORRECTION without BT_CONT_CORR_BOUNDS / USE_BT_CONT_TYPE",
                                  "GRADUAL_BT_ICS", "BT_NONLIN_STRESS", "DYNAMIC_SURFACE_PRESSURE", "BT_LINEAR_WAVE_DRAG",
                                  "CLIP_BT_VELOCITY", "CALCULATE_SAL", "BT_USE_OLD_CORIOLIS_BRACKET_BUG",
                                  "BAROTROPIC_ANSWER_DATE < 20190101"};
  for (int q = 0; q < 12; q++)
    M6_REQUIRE(cs->unsupported[q] == 0, "%s: %s is not provided by libmom6hip", who, names[q]);
  return 0;
}

struct Sizes { size_t h2, u2, v2, q2, h3, u3, v3; };
Sizes sizes_of(const m6::GridDev &g) {
  Sizes s;
  s.h2 = sizeof(double) * (size_t)g.nih * g.njh; s.u2 = sizeof(double) * (size_t)(g.nih + 1) * g.njh;
  s.v2 = sizeof(double) * (size_t)g.nih * (g.njh + 1); s.q2 = sizeof(double) * (size_t)(g.nih + 1) * (g.njh + 1);
  s.h3 = s.h2 * g.nk; s.u3 = s.u2 * g.nk; s.v3 = s.v2 * g.nk;
  return s;
}

// the state arrays of the control structure, staged like any other argument
struct CsDev { double *frhatu, *frhatv, *eta_cor, *IDatu, *IDatv, *ubtav, *vbtav, *q_D, *D_u_Cor, *D_v_Cor; };

dim3 grid2d(int i0, int i1, int j0, int j1) { return dim3((i1 - i0 + 64) / 64, (j1 - j0 + 4) / 4); }

}  // namespace

extern "C" {

int mom6hip_barotropic_init(mom6hip_ctx_t *ctx, mom6hip_barotropic_cs_t *cs, int32_t memspace) {
  M6_REQUIRE(ctx != nullptr, "barotropic_init: null context");
  if (int rc = check_cs(cs, "barotropic_init")) return rc;
  const m6::GridDev g = ctx->g;
  M6_REQUIRE(g.bathyT && g.areaT && g.mask2dT && g.mask2dCu && g.mask2dCv && g.CoriolisBu,
             "barotropic_init: bathyT, areaT, mask2dT, mask2dCu, mask2dCv and CoriolisBu are needed");
  M6_REQUIRE(cs->frhatu && cs->frhatv && cs->eta_cor && cs->IDatu && cs->IDatv && cs->ubtav && cs->vbtav,
             "barotropic_init: the control structure's state arrays must be allocated");
  M6_REQUIRE(!cs->linearized_BT_PV || (cs->q_D && cs->D_u_Cor && cs->D_v_Cor), "barotropic_init: q_D, D_u_Cor, D_v_Cor needed");
  const Sizes sz = sizes_of(g);
  m6::Stager st(ctx, memspace);
  CsDev d;
  d.frhatu = st.out(cs->frhatu, sz.u3); d.frhatv = st.out(cs->frhatv, sz.v3); d.eta_cor = st.out(cs->eta_cor, sz.h2);
  d.IDatu = st.out(cs->IDatu, sz.u2); d.IDatv = st.out(cs->IDatv, sz.v2); d.ubtav = st.out(cs->ubtav, sz.u2);
  d.vbtav = st.out(cs->vbtav, sz.v2);
  const bool lin = cs->linearized_BT_PV != 0;
  d.q_D = lin ? st.out(cs->q_D, sz.q2) : nullptr; d.D_u_Cor = lin ? st.out(cs->D_u_Cor, sz.u2) : nullptr;
  d.D_v_Cor = lin ? st.out(cs->D_v_Cor, sz.v2) : nullptr;
  M6_REQUIRE(!st.failed(), "barotropic_init: staging failed");
  hipStream_t s = ctx->stream;
  M6_HIP(hipMemsetAsync(d.frhatu, 0, sz.u3, s)); M6_HIP(hipMemsetAsync(d.frhatv, 0, sz.v3, s));
  M6_HIP(hipMemsetAsync(d.eta_cor, 0, sz.h2, s)); M6_HIP(hipMemsetAsync(d.IDatu, 0, sz.u2, s));
  M6_HIP(hipMemsetAsync(d.IDatv, 0, sz.v2, s)); M6_HIP(hipMemsetAsync(d.ubtav, 0, sz.u2, s));
  M6_HIP(hipMemsetAsync(d.vbtav, 0, sz.v2, s));
  const double Z_to_H = g.Z_to_H, Mean_SL = cs->Z_ref, scale = cs->BT_Coriolis_scale, hsub = g.H_subroundoff;
  if (cs->linearized_BT_PV) {   // :4826-4858
    M6_HIP(hipMemsetAsync(d.q_D, 0, sz.q2, s)); M6_HIP(hipMemsetAsync(d.D_u_Cor, 0, sz.u2, s));
    M6_HIP(hipMemsetAsync(d.D_v_Cor, 0, sz.v2, s));
    launch2d(s, g.isc - 1, g.iec, g.jsc, g.jec, [=] __device__(int I, int j) {
      d.D_u_Cor[g.u2(I, j)] = 0.5 * (m6::max2(Mean_SL + g.bathyT[g.h2(I + 1, j)], 0.0) + m6::max2(Mean_SL + g.bathyT[g.h2(I, j)], 0.0)) * Z_to_H;
    });
    launch2d(s, g.isc, g.iec, g.jsc - 1, g.jec, [=] __device__(int i, int J) {
      d.D_v_Cor[g.v2(i, J)] = 0.5 * (m6::max2(Mean_SL + g.bathyT[g.h2(i, J + 1)], 0.0) + m6::max2(Mean_SL + g.bathyT[g.h2(i, J)], 0.0)) * Z_to_H;
    });
    launch2d(s, g.isc - 1, g.iec, g.jsc - 1, g.jec, [=] __device__(int i, int j) {
      auto BT = [&](int a, int b) { return g.bathyT[g.h2(a, b)]; };
      auto AT = [&](int a, int b) { return g.areaT[g.h2(a, b)]; };
      auto MT = [&](int a, int b) { return g.mask2dT[g.h2(a, b)]; };
      double qv = 0.;
      if (MT(i, j) + MT(i, j + 1) + MT(i + 1, j) + MT(i + 1, j + 1) > 0.) {
        qv = 0.25 * (scale * g.CoriolisBu[g.q2(i, j)]) * ((AT(i, j) + AT(i + 1, j + 1)) + (AT(i + 1, j) + AT(i, j + 1))) /
             (Z_to_H * m6::max2(((AT(i, j) * m6::max2(Mean_SL + BT(i, j), 0.0) + AT(i + 1, j + 1) * m6::max2(Mean_SL + BT(i + 1, j + 1), 0.0)) +
                                 (AT(i + 1, j) * m6::max2(Mean_SL + BT(i + 1, j), 0.0) + AT(i, j + 1) * m6::max2(Mean_SL + BT(i, j + 1), 0.0))),
                                hsub));
      }
      d.q_D[g.q2(i, j)] = qv;
    });
    double *f[3] = {d.q_D, d.D_u_Cor, d.D_v_Cor};
    const int32_t pos[3] = {MOM6HIP_POS_Q, MOM6HIP_POS_U | MOM6HIP_PASS_SCALAR_PAIR, MOM6HIP_POS_V | MOM6HIP_PASS_SCALAR_PAIR},      // :4863-4865
                  nk[3] = {1, 1, 1};
    if (int rc = m6::group_pass(ctx, f, pos, nk, 3)) return rc;
  }
  launch2d(s, g.isc - 1, g.iec, g.jsc, g.jec, [=] __device__(int I, int j) {   // :5073-5079
    const double m = g.mask2dCu[g.u2(I, j)];
    d.IDatu[g.u2(I, j)] = (m > 0.) ? m * 2.0 / (Z_to_H * ((g.bathyT[g.h2(I + 1, j)] + g.bathyT[g.h2(I, j)]) + 2.0 * Mean_SL)) : 0.;
  });
  launch2d(s, g.isc, g.iec, g.jsc - 1, g.jec, [=] __device__(int i, int J) {   // :5080-5086
    const double m = g.mask2dCv[g.v2(i, J)];
    d.IDatv[g.v2(i, J)] = (m > 0.) ? m * 2.0 / (Z_to_H * ((g.bathyT[g.h2(i, J + 1)] + g.bathyT[g.h2(i, J)]) + 2.0 * Mean_SL)) : 0.;
  });
  M6_HIP(hipGetLastError());
  return st.finish();
}

int mom6hip_btcalc(mom6hip_ctx_t *ctx, mom6hip_barotropic_cs_t *cs, const double *h, const double *h_u, const double *h_v,
                   int32_t may_use_default, int32_t memspace) {
  return mom6hip_btcalc_obc(ctx, cs, h, h_u, h_v, may_use_default, nullptr, memspace);
}

// btcalc with OBC associated: at the faces of the open-boundary segments the weights are those of the cell inside (:3610-3664)
int mom6hip_btcalc_obc(mom6hip_ctx_t *ctx, mom6hip_barotropic_cs_t *cs, const double *h, const double *h_u, const double *h_v,
                       int32_t may_use_default, const mom6hip_obc_t *obc, int32_t memspace) {
  M6_REQUIRE(ctx != nullptr && cs != nullptr && h != nullptr, "btcalc: null argument");
  const m6::GridDev g = ctx->g;
  const int sch = cs->hvel_scheme;
  int use_default = 0;
  const bool given = h_u && h_v;
  if (!(given || sch == MOM6HIP_BT_HARMONIC || sch == MOM6HIP_BT_HYBRID || sch == MOM6HIP_BT_ARITHMETIC)) {
    M6_REQUIRE(may_use_default, "btcalc: Inconsistent settings of optional arguments and hvel_scheme.");
    use_default = 1;
  }
  const Sizes sz = sizes_of(g);
  m6::Stager st(ctx, memspace);
  const double *dh = st.in(h, sz.h3), *dhu = given ? st.in(h_u, sz.u3) : nullptr, *dhv = given ? st.in(h_v, sz.v3) : nullptr;
  double *fru = st.inout(cs->frhatu, sz.u3), *frv = st.inout(cs->frhatv, sz.v3);
  M6_REQUIRE(!st.failed(), "btcalc: staging failed");
  const double h_neglect = g.H_subroundoff, Z_to_H = g.Z_to_H;
  const int hybrid = (sch == MOM6HIP_BT_HYBRID || use_default), arith = (sch == MOM6HIP_BT_ARITHMETIC);
  // the segment at a face is the last one placed there (OBC%segnum_u / segnum_v): what the reference's loop over the segments leaves
  const int32_t *side_d[2] = {nullptr, nullptr};
  if (obc && obc->OBC_pe && m6::obc_side_maps(ctx, st, obc, &side_d[0], &side_d[1], "btcalc")) return 1;
  for (int dir = 0; dir < 2; dir++) {
    const double *hw = dir ? dhv : dhu;
    const int32_t *side = side_d[dir];
    double *fr = dir ? frv : fru;
    const long fstr = dir ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh, hstr = (long)g.nih * g.njh;
    if (hw && !side && g.nk <= 76) {      // the form of the RK2 step (BT_cont%h_u, %h_v given, no open boundaries)
      const dim3 grid = grid2d(dir ? g.isc : g.isc - 1, g.iec, dir ? g.jsc - 1 : g.jsc, g.jec), block(64, 4);
      if (g.nk <= 32) {
        if (dir) hipLaunchKernelGGL((btcalc_given_kernel<1, 32>), grid, block, 0, ctx->stream, g, hw, fr, h_neglect);
        else hipLaunchKernelGGL((btcalc_given_kernel<0, 32>), grid, block, 0, ctx->stream, g, hw, fr, h_neglect);
      } else {
        if (dir) hipLaunchKernelGGL((btcalc_given_kernel<1, 76>), grid, block, 0, ctx->stream, g, hw, fr, h_neglect);
        else hipLaunchKernelGGL((btcalc_given_kernel<0, 76>), grid, block, 0, ctx->stream, g, hw, fr, h_neglect);
      }
      continue;
    }
    launch2d(ctx->stream, dir ? g.isc : g.isc - 1, g.iec, dir ? g.jsc - 1 : g.jsc, g.jec, [=] __device__(int i, int j) {
      const long f2 = dir ? g.v2(i, j) : g.u2(i, j);
      const long hm = g.h2(i, j), hp = dir ? g.h2(i, j + 1) : g.h2(i + 1, j);
      const double mask = dir ? g.mask2dCv[f2] : g.mask2dCu[f2];
      const int nz = g.nk;
      double tot;
      if (side && side[f2]) {      // :3610-3664
        const long hc = side[f2] < 0 ? hm : hp;
        tot = dh[hc];
        for (int k = 1; k < nz; k++) tot = tot + dh[hc + hstr * k];
        const double Ihtot = mask / (tot + h_neglect);
        for (int k = 0; k < nz; k++) fr[f2 + fstr * k] = dh[hc + hstr * k] * Ihtot;
        return;
      }
      if (hw) {
        tot = hw[f2];
        for (int k = 1; k < nz; k++) tot = tot + hw[f2 + fstr * k];
        const double Ihat = mask / (tot + h_neglect);
        for (int k = 0; k < nz; k++) fr[f2 + fstr * k] = hw[f2 + fstr * k] * Ihat;
        return;
      }
      if (arith) {
        tot = 0.0;
        for (int k = 0; k < nz; k++) {
          const double v = 0.5 * (dh[hp + hstr * k] + dh[hm + hstr * k]);
          fr[f2 + fstr * k] = v;
          tot = (k == 0) ? v : tot + v;
        }
      } else if (hybrid) {
        const double bp = g.bathyT[hp], bm = g.bathyT[hm];
        double e_below = -0.5 * Z_to_H * (bp + bm);
        const double D_shallow = -Z_to_H * m6::min2(bp, bm);
        tot = 0.0;
        for (int k = nz - 1; k >= 0; k--) {
          const double a = dh[hp + hstr * k], b = dh[hm + hstr * k];
          const double e_here = e_below + 0.5 * (a + b);
          const double h_arith = 0.5 * (a + b);
          double v;
          if (e_below >= D_shallow) {
            v = h_arith;
          } else {
            const double h_harm = (a * b) / (h_arith + h_neglect);
            if (e_here <= D_shallow) {
              v = h_harm;
            } else {
              const double wt_arith = (e_here - D_shallow) / (h_arith + h_neglect);
              v = wt_arith * h_arith + (1.0 - wt_arith) * h_harm;
            }
          }
          fr[f2 + fstr * k] = v;
          tot = tot + v;
          e_below = e_here;
        }
      } else {   // HARMONIC
        tot = 0.0;
        for (int k = 0; k < nz; k++) {
          const double a = dh[hp + hstr * k], b = dh[hm + hstr * k];
          const double v = 2.0 * (a * b) / ((a + b) + h_neglect);
          fr[f2 + fstr * k] = v;
          tot = (k == 0) ? v : tot + v;
        }
      }
      const double Ihat = mask / (tot + h_neglect);
      for (int k = 0; k < nz; k++) fr[f2 + fstr * k] = fr[f2 + fstr * k] * Ihat;
    });
  }
  M6_HIP(hipGetLastError());
  return st.finish();
}

int mom6hip_bt_mass_source(mom6hip_ctx_t *ctx, mom6hip_barotropic_cs_t *cs, const double *h, const double *eta, int32_t set_cor,
                           int32_t memspace) {
  M6_REQUIRE(ctx != nullptr && cs != nullptr && h != nullptr && eta != nullptr, "bt_mass_source: null argument");
  const m6::GridDev g = ctx->g;
  const Sizes sz = sizes_of(g);
  m6::Stager st(ctx, memspace);
  const double *dh = st.in(h, sz.h3), *de = st.in(eta, sz.h2);
  double *cor = st.inout(cs->eta_cor, sz.h2);
  M6_REQUIRE(!st.failed(), "bt_mass_source: staging failed");
  const long hstr = (long)g.nih * g.njh;
  const double Z_to_H = g.Z_to_H;
  launch2d(ctx->stream, g.isc, g.iec, g.jsc, g.jec, [=] __device__(int i, int j) {
    const long n = g.h2(i, j);
    double eta_h = dh[n] - g.bathyT[n] * Z_to_H;
    for (int k = 1; k < g.nk; k++) eta_h = eta_h + dh[n + hstr * k];
    const double d_eta = eta_h - de[n];
    cor[n] = set_cor ? d_eta : cor[n] + d_eta;
  });
  M6_HIP(hipGetLastError());
  return st.finish();
}

int mom6hip_set_dtbt(mom6hip_ctx_t *ctx, mom6hip_barotropic_cs_t *cs, const double *pbce, const mom6hip_bt_cont_t *BT_cont,
                     double gtot_est, double SSH_add, int32_t memspace) {
  return mom6hip_set_dtbt_eta(ctx, cs, nullptr, pbce, BT_cont, gtot_est, SSH_add, memspace);
}

int mom6hip_set_dtbt_eta(mom6hip_ctx_t *ctx, mom6hip_barotropic_cs_t *cs, const double *eta, const double *pbce,
                         const mom6hip_bt_cont_t *BT_cont, double gtot_est, double SSH_add, int32_t memspace) {
  M6_REQUIRE(ctx != nullptr, "set_dtbt: null context");
  if (int rc = check_cs(cs, "set_dtbt")) return rc;
  const m6::GridDev g = ctx->g;
  const Sizes sz = sizes_of(g);
  m6::Stager st(ctx, memspace);
  const double *dp = pbce ? st.in(pbce, sz.h3) : nullptr;
  const double *deta = (eta && !BT_cont && cs->Nonlinear_continuity) ? st.in(eta, sz.h2) : nullptr;
  const double *fru = pbce ? st.in((const double *)cs->frhatu, sz.u3) : nullptr, *frv = pbce ? st.in((const double *)cs->frhatv, sz.v3) : nullptr;
  double *Datu = (double *)st.scratch(sz.u2), *Datv = (double *)st.scratch(sz.v2);
  unsigned long long *res = (unsigned long long *)st.scratch(sizeof(unsigned long long));
  const double *FA[8] = {};
  if (BT_cont) {
    const double *src[8] = {BT_cont->FA_u_EE, BT_cont->FA_u_E0, BT_cont->FA_u_W0, BT_cont->FA_u_WW,
                            BT_cont->FA_v_NN, BT_cont->FA_v_N0, BT_cont->FA_v_S0, BT_cont->FA_v_SS};
    for (int q = 0; q < 8; q++) { M6_REQUIRE(src[q], "set_dtbt: BT_cont is incomplete"); FA[q] = st.in(src[q], q < 4 ? sz.u2 : sz.v2); }
  }
  M6_REQUIRE(!st.failed() && Datu && Datv && res, "set_dtbt: staging failed");
  hipStream_t s = ctx->stream;
  M6_HIP(hipMemsetAsync(Datu, 0, sz.u2, s)); M6_HIP(hipMemsetAsync(Datv, 0, sz.v2, s));
  const unsigned long long inf_bits = 0x7FF0000000000000ull;
  M6_HIP(hipMemcpyAsync(res, &inf_bits, sizeof(inf_bits), hipMemcpyHostToDevice, s));
  if (BT_cont) {   // BT_cont_to_face_areas :4182, halo = 0
    const double *a0 = FA[0], *a1 = FA[1], *a2 = FA[2], *a3 = FA[3], *b0 = FA[4], *b1 = FA[5], *b2 = FA[6], *b3 = FA[7];
    launch2d(s, g.isc - 1, g.iec, g.jsc, g.jec, [=] __device__(int I, int j) {
      const long n = g.u2(I, j);
      Datu[n] = m6::max2(m6::max2(m6::max2(a0[n], a1[n]), a2[n]), a3[n]);
    });
    launch2d(s, g.isc, g.iec, g.jsc - 1, g.jec, [=] __device__(int i, int J) {
      const long n = g.v2(i, J);
      Datv[n] = m6::max2(m6::max2(m6::max2(b0[n], b1[n]), b2[n]), b3[n]);
    });
  } else if (deta) {   // NONLINEAR_BT_CONTINUITY with eta present :2871-2872: find_face_areas with eta :4246-4262, halo 0
    const double Z_to_H = g.Z_to_H;
    launch2d(s, g.isc - 1, g.iec, g.jsc, g.jec, [=] __device__(int I, int j) {
      const double H1 = g.bathyT[g.h2(I, j)] * Z_to_H + deta[g.h2(I, j)], H2 = g.bathyT[g.h2(I + 1, j)] * Z_to_H + deta[g.h2(I + 1, j)];
      double D = 0.0;
      if ((H1 > 0.0) && (H2 > 0.0)) D = g.dy_Cu[g.u2(I, j)] * (2.0 * H1 * H2) / (H1 + H2);
      Datu[g.u2(I, j)] = D;
    });
    launch2d(s, g.isc, g.iec, g.jsc - 1, g.jec, [=] __device__(int i, int J) {
      const double H1 = g.bathyT[g.h2(i, J)] * Z_to_H + deta[g.h2(i, J)], H2 = g.bathyT[g.h2(i, J + 1)] * Z_to_H + deta[g.h2(i, J + 1)];
      double D = 0.0;
      if ((H1 > 0.0) && (H2 > 0.0)) D = g.dx_Cv[g.v2(i, J)] * (2.0 * H1 * H2) / (H1 + H2);
      Datv[g.v2(i, J)] = D;
    });
  } else {   // find_face_areas with add_max :4283-4295, halo 0
    const double Z_to_H = g.Z_to_H, add = cs->Z_ref + SSH_add;
    launch2d(s, g.isc - 1, g.iec, g.jsc, g.jec, [=] __device__(int I, int j) {
      Datu[g.u2(I, j)] = g.dy_Cu[g.u2(I, j)] * Z_to_H * m6::max2(m6::max2(g.bathyT[g.h2(I + 1, j)], g.bathyT[g.h2(I, j)]) + add, 0.0);
    });
    launch2d(s, g.isc, g.iec, g.jsc - 1, g.jec, [=] __device__(int i, int J) {
      Datv[g.v2(i, J)] = g.dx_Cv[g.v2(i, J)] * Z_to_H * m6::max2(m6::max2(g.bathyT[g.h2(i, J + 1)], g.bathyT[g.h2(i, J)]) + add, 0.0);
    });
  }
  hipLaunchKernelGGL(bt_dtbt_kernel, grid2d(g.isc, g.iec, g.jsc, g.jec), dim3(64, 4), 0, s, g, dp, fru, frv, Datu, Datv, gtot_est,
                     cs->bebt, cs->BT_Coriolis_scale, res);
  M6_HIP(hipGetLastError());
  // min_across_PEs(dtbt_max) :2915 is taken of the tile's minimum where it lies, before the square root: dtbt_max = sqrt(min / dgeo_de) is
  // a non-decreasing function of it (the clamp to 1e38, the division and the correctly rounded square root all are), so the minimum over
  // the PEs of the tiles' dtbt_max is the dtbt_max of the minimum, bit for bit
  unsigned long long bits = 0;
  if (m6::multi_tile(ctx)) {
    if (int rc = m6::min_across_PEs_dev_u64(ctx, (unsigned long long *)res, 1, &bits)) return rc;
  } else {
    M6_HIP(hipMemcpyAsync(&bits, res, sizeof(bits), hipMemcpyDeviceToHost, s));
    M6_HIP(hipStreamSynchronize(s));
  }
  double min_max_dt2;
  memcpy(&min_max_dt2, &bits, sizeof(double));
  if (!(min_max_dt2 < 1.0e38)) min_max_dt2 = 1.0e38;
  const double dgeo_de = 1.0 + (cs->G_extra > 0.0 ? cs->G_extra : 0.0);
  const double dtbt_max = sqrt(min_max_dt2 / dgeo_de);
  cs->dtbt = cs->dtbt_fraction * dtbt_max;
  cs->dtbt_max = dtbt_max;
  return st.finish();
}

int mom6hip_btstep(mom6hip_ctx_t *ctx, mom6hip_barotropic_cs_t *cs, const double *U_in, const double *V_in, const double *eta_in,
                   double dt, const double *bc_accel_u, const double *bc_accel_v, const double *taux, const double *tauy,
                   double RZ_to_H, const double *pbce, const double *eta_PF_in, const double *U_Cor, const double *V_Cor,
                   double *accel_layer_u, double *accel_layer_v, double *eta_out, double *uhbtav, double *vhbtav,
                   const double *visc_rem_u, const double *visc_rem_v, const mom6hip_bt_cont_t *BT_cont, const double *eta_PF_start,
                   const double *taux_bot, const double *tauy_bot, const double *uh0, const double *vh0, const double *u_uh0,
                   const double *v_vh0, double *etaav, int32_t memspace) {
  return mom6hip_btstep_obc(ctx, cs, U_in, V_in, eta_in, dt, bc_accel_u, bc_accel_v, taux, tauy, RZ_to_H, pbce, eta_PF_in, U_Cor, V_Cor,
                            accel_layer_u, accel_layer_v, eta_out, uhbtav, vhbtav, visc_rem_u, visc_rem_v, BT_cont, eta_PF_start, taux_bot,
                            tauy_bot, uh0, vh0, u_uh0, v_vh0, etaav, nullptr, memspace);
}

// btstep with OBC associated: specified, Flather and gradient segments (:1089-1110, set_up_BT_OBC :3172, :1236-1250, apply_velocity_OBCs
// :2931 inside the time steps, :2490-2519, :2591-2606)
int mom6hip_btstep_obc(mom6hip_ctx_t *ctx, mom6hip_barotropic_cs_t *cs, const double *U_in, const double *V_in, const double *eta_in,
                   double dt, const double *bc_accel_u, const double *bc_accel_v, const double *taux, const double *tauy,
                   double RZ_to_H, const double *pbce, const double *eta_PF_in, const double *U_Cor, const double *V_Cor,
                   double *accel_layer_u, double *accel_layer_v, double *eta_out, double *uhbtav, double *vhbtav,
                   const double *visc_rem_u, const double *visc_rem_v, const mom6hip_bt_cont_t *BT_cont, const double *eta_PF_start,
                   const double *taux_bot, const double *tauy_bot, const double *uh0, const double *vh0, const double *u_uh0,
                   const double *v_vh0, double *etaav, const mom6hip_obc_t *obc, int32_t memspace) {
  M6_REQUIRE(ctx != nullptr, "btstep: null context");
  if (int rc = check_cs(cs, "btstep")) return rc;
  M6_REQUIRE(U_in && V_in && eta_in && bc_accel_u && bc_accel_v && taux && tauy && pbce && eta_PF_in && U_Cor && V_Cor &&
                 accel_layer_u && accel_layer_v && eta_out && uhbtav && vhbtav && visc_rem_u && visc_rem_v,
             "btstep: a required argument is null");
  M6_REQUIRE((uh0 != nullptr) == (vh0 != nullptr && u_uh0 != nullptr && v_vh0 != nullptr),
             "btstep: vh0, u_uh0, and v_vh0 must be associated if uh0 is used.");
  M6_REQUIRE(cs->dtbt > 0.0 && dt > 0.0, "btstep: dt and CS%%dtbt must be positive (call set_dtbt first)");
  const m6::GridDev g = ctx->g;
  M6_REQUIRE(g.bathyT && g.IareaT && g.areaT && g.IdxCu && g.IdyCv && g.dy_Cu && g.dx_Cv && g.mask2dT && g.mask2dCu &&
                 g.mask2dCv && g.CoriolisBu, "btstep: a metric array it needs was not given to grid_create");
  const int is = g.isc, ie = g.iec, js = g.jsc, je = g.jec;
  const Sizes sz = sizes_of(g);
  const bool use_BT_cont = BT_cont != nullptr, interp = eta_PF_start != nullptr, add_uh0 = uh0 != nullptr;
  const bool find_etaav = etaav != nullptr, have_bot = taux_bot && tauy_bot;
  // NONLINEAR_BT_CONTINUITY without a BT_cont: the face areas are refreshed from eta every Nonlin_cont_update_period steps, and
  // the wide-halo march gives up two points a step (:751-753)
  const bool nonlin_cont = !use_BT_cont && cs->Nonlinear_continuity;
  const bool nonlin_update = nonlin_cont && cs->Nonlin_cont_update_period > 0;
  const int stencil = nonlin_update ? 2 : 1;

  int num_cycles = 1;
  if (cs->use_wide_halos) num_cycles = std::min((is - g.isd) / stencil, (js - g.jsd) / stencil);
  M6_REQUIRE(num_cycles >= 1, "btstep: the grid has no halo");
  const int isvf = is - (num_cycles - 1) * stencil, ievf = ie + (num_cycles - 1) * stencil;
  const int jsvf = js - (num_cycles - 1) * stencil, jevf = je + (num_cycles - 1) * stencil;
  const int nstep = (int)ceil(dt / cs->dtbt - 0.0001);
  M6_REQUIRE(nstep >= 1, "btstep: nstep < 1");
  cs->nstep_last = nstep;

  Par p;
  p.Instep = 1.0 / (double)nstep; p.dtbt = dt * p.Instep; p.dgeo_de = 1.0 + cs->G_extra; p.vel_underflow = cs->vel_underflow;
  p.project_velocity = cs->BT_project_velocity ? 1 : 0;
  p.trans_wt1 = p.project_velocity ? (1.0 + cs->bebt) : cs->bebt; p.trans_wt2 = p.project_velocity ? -cs->bebt : (1.0 - cs->bebt);      // :804-808
  p.RZ_to_H = RZ_to_H; p.nstep = nstep; p.use_BT_cont = use_BT_cont;
  p.interp_eta_PF = interp; p.add_uh0 = add_uh0; p.strong_drag = cs->strong_drag; p.visc_rem_u_uh0 = cs->visc_rem_u_uh0;
  p.find_etaav = find_etaav; p.have_bot = have_bot;
  const double accel_underflow = cs->vel_underflow * (1.0 / dt);

  // filter weights :1753-1808 (host: a few dozen scalars, handed to the kernels by value)
  double dt_filt;
  if (cs->dt_bt_filter >= 0.0) dt_filt = 0.5 * std::max(0.0, std::min(cs->dt_bt_filter, 2.0 * dt));
  else dt_filt = 0.5 * std::max(0.0, dt * std::min(-cs->dt_bt_filter, 2.0));
  const int nfilter = (int)ceil(dt_filt / p.dtbt);
  const int nt = nstep + nfilter;
  M6_REQUIRE(nt > 0, "btstep: number of barotropic step (nstep+nfilter) is 0");
  std::vector<double> wt_vel(nt + 2, 0.0), wt_eta(nt + 2, 0.0), wt_trans(nt + 2, 0.0), wt_accel(nt + 2, 0.0), wt_accel2(nt + 2, 0.0);
  {
    double sum_wt_vel = 0.0, sum_wt_eta = 0.0, sum_wt_accel = 0.0, sum_wt_trans = 0.0;
    for (int n = 1; n <= nt; n++) {
      if ((n == nstep) || (dt_filt - abs(n - nstep) * p.dtbt >= 0.0)) { wt_vel[n] = 1.0; wt_eta[n] = 1.0; }
      else if (p.dtbt + dt_filt - abs(n - nstep) * p.dtbt > 0.0) { wt_vel[n] = 1.0 + (dt_filt / p.dtbt) - abs(n - nstep); wt_eta[n] = wt_vel[n]; }
      else { wt_vel[n] = 0.0; wt_eta[n] = 0.0; }
      sum_wt_vel = sum_wt_vel + wt_vel[n]; sum_wt_eta = sum_wt_eta + wt_eta[n];
    }
    for (int n = nt; n >= 1; n--) {
      wt_trans[n] = wt_trans[n + 1] + wt_eta[n];
      wt_accel[n] = wt_accel[n + 1] + wt_vel[n];
      sum_wt_accel = sum_wt_accel + wt_accel[n]; sum_wt_trans = sum_wt_trans + wt_trans[n];
    }
    const double Iv = 1.0 / sum_wt_vel, Ia = 1.0 / sum_wt_accel, Ie = 1.0 / sum_wt_eta, It = 1.0 / sum_wt_trans;
    for (int n = 1; n <= nt; n++) {
      wt_vel[n] = wt_vel[n] * Iv; wt_accel2[n] = wt_accel[n] * Ia; wt_trans[n] = wt_trans[n] * It;
      wt_accel[n] = wt_accel[n] * Ia; wt_eta[n] = wt_eta[n] * Ie;
    }
  }

  // ---- stage the arguments
  m6::Stager st(ctx, memspace);
  const double *dU = st.in(U_in, sz.u3), *dV = st.in(V_in, sz.v3), *deta_in = st.in(eta_in, sz.h2);
  const double *dbu = st.in(bc_accel_u, sz.u3), *dbv = st.in(bc_accel_v, sz.v3), *dtx = st.in(taux, sz.u2), *dty = st.in(tauy, sz.v2);
  const double *dpb = st.in(pbce, sz.h3), *depf = st.in(eta_PF_in, sz.h2), *dUc = st.in(U_Cor, sz.u3), *dVc = st.in(V_Cor, sz.v3);
  const double *dvru = st.in(visc_rem_u, sz.u3), *dvrv = st.in(visc_rem_v, sz.v3);
  const double *depfs = st.in(eta_PF_start, sz.h2), *dtbx = have_bot ? st.in(taux_bot, sz.u2) : nullptr,
               *dtby = have_bot ? st.in(tauy_bot, sz.v2) : nullptr;
  const double *duh0 = st.in(uh0, sz.u3), *dvh0 = st.in(vh0, sz.v3), *duu0 = st.in(u_uh0, sz.u3), *dvv0 = st.in(v_vh0, sz.v3);
  // outputs that are written on the compute domain only are staged in and out, so the rest keeps the caller's values
  double *dalu = st.inout(accel_layer_u, sz.u3), *dalv = st.inout(accel_layer_v, sz.v3);
  double *deta_out = st.inout(eta_out, sz.h2);
  double *duhbtav = st.inout(uhbtav, sz.u2), *dvhbtav = st.inout(vhbtav, sz.v2), *detaav = st.inout(etaav, sz.h2);
  CsDev c;
  c.frhatu = st.inout(cs->frhatu, sz.u3); c.frhatv = st.inout(cs->frhatv, sz.v3); c.eta_cor = st.inout(cs->eta_cor, sz.h2);
  c.IDatu = st.inout(cs->IDatu, sz.u2); c.IDatv = st.inout(cs->IDatv, sz.v2);
  c.ubtav = st.inout(cs->ubtav, sz.u2); c.vbtav = st.inout(cs->vbtav, sz.v2);
  c.q_D = cs->linearized_BT_PV ? st.inout(cs->q_D, sz.q2) : nullptr;
  c.D_u_Cor = cs->linearized_BT_PV ? st.inout(cs->D_u_Cor, sz.u2) : nullptr;
  c.D_v_Cor = cs->linearized_BT_PV ? st.inout(cs->D_v_Cor, sz.v2) : nullptr;
  M6_REQUIRE(!cs->linearized_BT_PV || (c.q_D && c.D_u_Cor && c.D_v_Cor), "btstep: q_D, D_u_Cor and D_v_Cor are needed (barotropic_init)");
  const double *bcU[6] = {}, *bcV[6] = {};
  if (use_BT_cont) {
    const double *su[6] = {BT_cont->FA_u_EE, BT_cont->FA_u_E0, BT_cont->FA_u_W0, BT_cont->FA_u_WW, BT_cont->uBT_EE, BT_cont->uBT_WW};
    const double *sv[6] = {BT_cont->FA_v_NN, BT_cont->FA_v_N0, BT_cont->FA_v_S0, BT_cont->FA_v_SS, BT_cont->vBT_NN, BT_cont->vBT_SS};
    for (int q = 0; q < 6; q++) {
      M6_REQUIRE(su[q] && sv[q], "btstep: BT_cont is incomplete");
      bcU[q] = st.in(su[q], sz.u2); bcV[q] = st.in(sv[q], sz.v2);
    }
  }
  // one zero-filled block for every 2-D work array
  const int NU_ARR = 21 + 10, NV_ARR = 13 + 10, NH_ARR = 13;
  const size_t total = NU_ARR * sz.u2 + NV_ARR * sz.v2 + NH_ARR * sz.h2 + sz.q2;
  char *blk = (char *)st.scratch(total);
  M6_REQUIRE(!st.failed() && blk, "btstep: staging failed");
  hipStream_t s = ctx->stream;
  M6_HIP(hipMemsetAsync(blk, 0, total, s));
  Work w;
  {
    char *q = blk;
    auto U = [&]() { double *r = (double *)q; q += sz.u2; return r; };
    auto V = [&]() { double *r = (double *)q; q += sz.v2; return r; };
    auto H = [&]() { double *r = (double *)q; q += sz.h2; return r; };
    w.ubt = U(); w.bt_rem_u = U(); w.BT_force_u = U(); w.u_accel_bt = U(); w.uhbt = U(); w.uhbt0 = U(); w.uhbtp = U();
    w.azon = U(); w.bzon = U(); w.czon = U(); w.dzon = U(); w.Cor_ref_u = U(); w.DCor_u = U(); w.Datu = U(); w.ubt_Cor = U();
    w.ubt0 = U(); w.uhbtS = U(); w.amer = U(); w.bmer = U(); w.cmer = U(); w.dmer = U();
    double **bu = (double **)&w.BU; for (int k = 0; k < 10; k++) bu[k] = U();
    w.vbt = V(); w.bt_rem_v = V(); w.BT_force_v = V(); w.v_accel_bt = V(); w.vhbt = V(); w.vhbt0 = V(); w.vhbtp = V();
    w.Cor_ref_v = V(); w.DCor_v = V(); w.Datv = V(); w.vbt_Cor = V(); w.vbt0 = V(); w.vhbtS = V();
    double **bv = (double **)&w.BV; for (int k = 0; k < 10; k++) bv[k] = V();
    w.eta = H(); w.eta_pred = H(); w.eta_sum = H(); w.eta_wtd = H(); w.eta_PF = H(); w.eta_PF_1 = H(); w.d_eta_PF = H();
    w.gtot_E = H(); w.gtot_W = H(); w.gtot_N = H(); w.gtot_S = H(); w.eta_src = H(); w.e_anom = H();
    w.q = (double *)q;
  }
  // the second set of the five fields that alternate between the steps of the fused kernel (bt_step_fused_kernel)
  double *alt_eta = nullptr, *alt_ubt = nullptr, *alt_vbt = nullptr, *alt_uhbtp = nullptr, *alt_vhbtp = nullptr;
  {
    char *q = (char *)st.scratch(2 * sz.u2 + 2 * sz.v2 + sz.h2);
    M6_REQUIRE(!st.failed() && q, "btstep: staging failed");
    M6_HIP(hipMemsetAsync(q, 0, 2 * sz.u2 + 2 * sz.v2 + sz.h2, s));
    alt_ubt = (double *)q; alt_uhbtp = (double *)(q + sz.u2); alt_vbt = (double *)(q + 2 * sz.u2); alt_vhbtp = (double *)(q + 2 * sz.u2 + sz.v2);
    alt_eta = (double *)(q + 2 * sz.u2 + 2 * sz.v2);
  }
  // ---- open boundaries :770-780
  w.obc_u = w.obc_v = nullptr;
  w.ob_Cg_u = w.ob_dZ_u = w.ob_uhbt = w.ob_ubt_outer = w.ob_SSH_u = w.ubt_old = w.ubt_first = w.ubt_wtd = nullptr;
  w.ob_Cg_v = w.ob_dZ_v = w.ob_vhbt = w.ob_vbt_outer = w.ob_SSH_v = w.vbt_old = w.vbt_first = w.vbt_wtd = nullptr;
  bool apply_OBCs = false, apply_u_OBCs = false, apply_v_OBCs = false;
  if (obc) {
    apply_u_OBCs = obc->open_u_BCs_exist_globally || obc->specified_u_BCs_exist_globally;
    apply_v_OBCs = obc->open_v_BCs_exist_globally || obc->specified_v_BCs_exist_globally;
    apply_OBCs = obc->specified_u_BCs_exist_globally || obc->specified_v_BCs_exist_globally || obc->Flather_u_BCs_exist_globally ||
                 obc->Flather_v_BCs_exist_globally || obc->open_u_BCs_exist_globally || obc->open_v_BCs_exist_globally;
  }
  if (apply_OBCs) {
    M6_REQUIRE(obc->number_of_segments == 0 || (obc->segment && obc->segnum_u && obc->segnum_v), "btstep: OBC%%segment, segnum_u and segnum_v are required");
    const size_t nU = sz.u2 / 8, nV = sz.v2 / 8;
    std::vector<int32_t> codes(nU + nV, 0);
    for (int d = 0; d < 2; d++) {
      if (!(d ? apply_v_OBCs : apply_u_OBCs)) continue;      // (the loops of the reference are inside BT_OBC%apply_u_OBCs | apply_v_OBCs)
      const int32_t *segnum = d ? obc->segnum_v : obc->segnum_u;
      int32_t *o = codes.data() + (d ? nU : 0);
      for (size_t n = 0; n < (d ? nV : nU); n++) {
        const int l = segnum[n];
        if (l == MOM6HIP_OBC_NONE) continue;
        M6_REQUIRE(l >= 1 && l <= obc->number_of_segments, "btstep: OBC%%segnum_%c holds %d, with %d segments", d ? 'v' : 'u', l, obc->number_of_segments);
        const mom6hip_obc_segment_t &S = obc->segment[l - 1];
        int c = 0;
        if (S.direction == (d ? MOM6HIP_OBC_DIRECTION_N : MOM6HIP_OBC_DIRECTION_E)) c = OB_PLUS;
        else if (S.direction == (d ? MOM6HIP_OBC_DIRECTION_S : MOM6HIP_OBC_DIRECTION_W)) c = OB_MINUS;
        if (S.specified) c |= OB_SPEC;
        else {
          M6_REQUIRE(c != 0, "btstep: OBC segment %d does not lie along the faces it is listed on", l);
          if (S.Flather) c |= OB_FLATHER;
          else if (S.gradient) c |= OB_GRAD;
          else M6_REQUIRE(false, "btstep: OBC segment %d is neither specified, Flather nor gradient: its barotropic velocity is not defined "
                                 "(apply_velocity_OBCs, MOM_barotropic.F90:3021-3092)", l);
        }
        o[n] = c;
      }
    }
    // (the codes change only with the OBC: uploaded when they differ from the copy the context keeps, never waited for)
    const int32_t *dcodes = (const int32_t *)m6::obc_table_content(ctx, m6::OBC_SITE_BT_CODES, m6::obc_fingerprint(ctx, obc), codes.data(), 4 * (nU + nV));
    char *ob = (char *)st.scratch(8 * sz.u2 + 8 * sz.v2);
    M6_REQUIRE(!st.failed() && dcodes && ob, "btstep: out of device memory for the open boundaries");
    M6_HIP(hipMemsetAsync(ob, 0, 8 * sz.u2 + 8 * sz.v2, s));
    w.obc_u = dcodes; w.obc_v = dcodes + nU;
    double **pu[8] = {&w.ob_Cg_u, &w.ob_dZ_u, &w.ob_uhbt, &w.ob_ubt_outer, &w.ob_SSH_u, &w.ubt_old, &w.ubt_first, &w.ubt_wtd};
    double **pv[8] = {&w.ob_Cg_v, &w.ob_dZ_v, &w.ob_vhbt, &w.ob_vbt_outer, &w.ob_SSH_v, &w.vbt_old, &w.vbt_first, &w.vbt_wtd};
    for (int q = 0; q < 8; q++) { *pu[q] = (double *)(ob + q * sz.u2); *pv[q] = (double *)(ob + 8 * sz.u2 + q * sz.v2); }
  }
  // the part of set_up_BT_OBC :3234-3262, :3296-3322 for the specified segments: the external transports summed over the layers, and
  // the barotropic velocities that carry them
  auto setup_specified = [&]() -> int {
    const int halo = ievf - ie, s_is = is - halo, s_ie = ie + halo, s_js = js - halo, s_je = je + halo;
    for (int d = 0; d < 2; d++) {
      if (!(d ? apply_v_OBCs : apply_u_OBCs)) continue;
      if (!(d ? obc->specified_v_BCs_exist_globally : obc->specified_u_BCs_exist_globally)) continue;
      double *o_hbt = d ? w.ob_vhbt : w.ob_uhbt, *o_outer = d ? w.ob_vbt_outer : w.ob_ubt_outer;
      const int32_t *code = d ? w.obc_v : w.obc_u;
      const Btcl B = d ? w.BV : w.BU;
      const double *Dat = d ? w.Datv : w.Datu;
      const int nk = g.nk;
      for (int n = 0; n < obc->number_of_segments; n++) {
        const mom6hip_obc_segment_t &S = obc->segment[n];
        if (!((d ? S.is_N_or_S : S.is_E_or_W) && S.specified)) continue;
        M6_REQUIRE(S.normal_trans, "btstep: segment %d is specified: normal_trans is required", n + 1);
        const int a0 = d ? S.isd : S.IsdB, a1 = d ? S.ied : S.IedB, b0 = d ? S.JsdB : S.jsd, b1 = d ? S.JedB : S.jed;
        const long na = a1 - a0 + 1, nb = b1 - b0 + 1;
        const double *nt_ = st.in(S.normal_trans, (size_t)na * nb * nk * 8);
        M6_REQUIRE(!st.failed() && nt_, "btstep: staging failed");
        launch2d(s, a0, a1, b0, b1, [=] __device__(int a, int b) {
          double sum = 0.;
          for (int k = 0; k < nk; k++) sum = sum + nt_[(a - a0) + na * ((b - b0) + nb * (long)k)];
          o_hbt[d ? g.v2(a, b) : g.u2(a, b)] = sum;
        });
      }
      launch2d(s, d ? s_is : s_is - 1, s_ie, d ? s_js - 1 : s_js, s_je, [=] __device__(int i, int j) {
        const long f = d ? g.v2(i, j) : g.u2(i, j);
        if (!(code[f] & OB_SPEC)) return;
        if (use_BT_cont) o_outer[f] = uhbt_to_ubt(o_hbt[f], B, f);
        else if (Dat[f] > 0.0) o_outer[f] = o_hbt[f] / Dat[f];
      });
    }
    return 0;
  };
  auto pass = [&](std::initializer_list<std::pair<double *, int>> fl) -> int {
    std::vector<double *> f; std::vector<int32_t> pos, nk;
    for (auto &e : fl) { f.push_back(e.first); pos.push_back(e.second); nk.push_back(1); }
    return m6::group_pass(ctx, f.data(), pos.data(), nk.data(), (int)f.size());
  };
  const int PH = MOM6HIP_POS_H, PU = MOM6HIP_POS_U, PV = MOM6HIP_POS_V, PQ = MOM6HIP_POS_Q;
  const int PUs = PU | MOM6HIP_PASS_SCALAR_PAIR, PVs = PV | MOM6HIP_PASS_SCALAR_PAIR;      // To_All+Scalar_Pair: no sign change across the fold

  // ---- q, DCor_u, DCor_v :884-945
  if (cs->linearized_BT_PV) {
    M6_HIP(hipMemcpyAsync(w.q, c.q_D, sz.q2, hipMemcpyDeviceToDevice, s));
    M6_HIP(hipMemcpyAsync(w.DCor_u, c.D_u_Cor, sz.u2, hipMemcpyDeviceToDevice, s));
    M6_HIP(hipMemcpyAsync(w.DCor_v, c.D_v_Cor, sz.v2, hipMemcpyDeviceToDevice, s));
  } else {
    const double Z_to_H = g.Z_to_H, scale = cs->BT_Coriolis_scale, h_neglect = g.H_subroundoff;
    launch2d(s, is - 1, ie, js, je, [=] __device__(int I, int j) {
      w.DCor_u[g.u2(I, j)] = 0.5 * (m6::max2(Z_to_H * g.bathyT[g.h2(I + 1, j)] + deta_in[g.h2(I + 1, j)], 0.0) +
                                    m6::max2(Z_to_H * g.bathyT[g.h2(I, j)] + deta_in[g.h2(I, j)], 0.0));
    });
    launch2d(s, is, ie, js - 1, je, [=] __device__(int i, int J) {   // eta_in(i+1,j): as written in the reference (:911)
      w.DCor_v[g.v2(i, J)] = 0.5 * (m6::max2(Z_to_H * g.bathyT[g.h2(i, J + 1)] + deta_in[g.h2(i + 1, J)], 0.0) +
                                    m6::max2(Z_to_H * g.bathyT[g.h2(i, J)] + deta_in[g.h2(i, J)], 0.0));
    });
    launch2d(s, is - 1, ie, js - 1, je, [=] __device__(int i, int j) {
      auto AT = [&](int a, int b) { return g.areaT[g.h2(a, b)]; };
      auto HT = [&](int a, int b) { return m6::max2(Z_to_H * g.bathyT[g.h2(a, b)] + deta_in[g.h2(a, b)], 0.0); };
      w.q[g.q2(i, j)] = 0.25 * (scale * g.CoriolisBu[g.q2(i, j)]) * ((AT(i, j) + AT(i + 1, j + 1)) + (AT(i + 1, j) + AT(i, j + 1))) /
                        (m6::max2((AT(i, j) * HT(i, j) + AT(i + 1, j + 1) * HT(i + 1, j + 1)) +
                                  (AT(i + 1, j) * HT(i + 1, j) + AT(i, j + 1) * HT(i, j + 1)), h_neglect));
    });
    if (int rc = pass({{w.q, PQ}, {w.DCor_u, PUs}, {w.DCor_v, PVs}})) return rc;      // :822-824
  }

  // ---- copies of the inputs on the data domain :1011-1033
  M6_HIP(hipMemcpyAsync(w.eta, deta_in, sz.h2, hipMemcpyDeviceToDevice, s));
  if (interp) {
    M6_HIP(hipMemcpyAsync(w.eta_PF_1, depfs, sz.h2, hipMemcpyDeviceToDevice, s));
    launch2d(s, g.isd, g.ied, g.jsd, g.jed, [=] __device__(int i, int j) { w.d_eta_PF[g.h2(i, j)] = depf[g.h2(i, j)] - depfs[g.h2(i, j)]; });
  } else {
    M6_HIP(hipMemcpyAsync(w.eta_PF, depf, sz.h2, hipMemcpyDeviceToDevice, s));
  }

  // ---- open face areas :1136-1148
  const int hs = 1 + ievf - ie;
  const bool early_btcl_pass = use_BT_cont && add_uh0 && cs->adjust_BT_cont;
  std::function<void(int)> btcl_derive;
  // find_face_areas with eta :4246-4262 (Boussinesq) on the stream `fs`: the harmonic mean of the two total depths
  auto face_areas_eta = [&](hipStream_t fs, int hs) {
    const double Z_to_H = g.Z_to_H;
    launch2d(fs, is - 1 - hs, ie + hs, js - hs, je + hs, [=] __device__(int I, int j) {
      const double H1 = g.bathyT[g.h2(I, j)] * Z_to_H + w.eta[g.h2(I, j)], H2 = g.bathyT[g.h2(I + 1, j)] * Z_to_H + w.eta[g.h2(I + 1, j)];
      double D = 0.0;
      if ((H1 > 0.0) && (H2 > 0.0)) D = g.dy_Cu[g.u2(I, j)] * (2.0 * H1 * H2) / (H1 + H2);
      w.Datu[g.u2(I, j)] = D;
    });
    launch2d(fs, is - hs, ie + hs, js - 1 - hs, je + hs, [=] __device__(int i, int J) {
      const double H1 = g.bathyT[g.h2(i, J)] * Z_to_H + w.eta[g.h2(i, J)], H2 = g.bathyT[g.h2(i, J + 1)] * Z_to_H + w.eta[g.h2(i, J + 1)];
      double D = 0.0;
      if ((H1 > 0.0) && (H2 > 0.0)) D = g.dx_Cv[g.v2(i, J)] * (2.0 * H1 * H2) / (H1 + H2);
      w.Datv[g.v2(i, J)] = D;
    });
  };
  if (use_BT_cont) {   // set_local_BT_cont_types :3949 (dt = 1)
    const Btcl BU = w.BU, BV = w.BV;
    const double *a0 = bcU[0], *a1 = bcU[1], *a2 = bcU[2], *a3 = bcU[3], *a4 = bcU[4], *a5 = bcU[5];
    const double *b0 = bcV[0], *b1 = bcV[1], *b2 = bcV[2], *b3 = bcV[3], *b4 = bcV[4], *b5 = bcV[5];
    launch2d(s, is - 1, ie, js, je, [=] __device__(int I, int j) {
      const long n = g.u2(I, j);
      BU.FA_EE[n] = a0[n]; BU.FA_E0[n] = a1[n]; BU.FA_W0[n] = a2[n]; BU.FA_WW[n] = a3[n]; BU.uBT_EE[n] = a4[n]; BU.uBT_WW[n] = a5[n];
    });
    launch2d(s, is, ie, js - 1, je, [=] __device__(int i, int J) {
      const long n = g.v2(i, J);
      BV.FA_EE[n] = b0[n]; BV.FA_E0[n] = b1[n]; BV.FA_W0[n] = b2[n]; BV.FA_WW[n] = b3[n]; BV.uBT_EE[n] = b4[n]; BV.uBT_WW[n] = b5[n];
    });
    // The halo pass of the six raw arrays per direction is merged with pass_gtot below (one message instead of two):
    // only the compute-range fits are needed until then (uhbt0), and the derived values are pure functions of the raw
    // ones, so they are evaluated on the compute range now and on the widened range after the pass.  With
    // ADJUST_BT_CONT the fits are modified on the widened range before uhbt0, so the reference's order is kept.
    btcl_derive = [=](int hs_) {
      for (int dir = 0; dir < 2; dir++) {
        const Btcl B = dir ? BV : BU;
        launch2d(s, (dir ? is : is - 1) - hs_, ie + hs_, (dir ? js - 1 : js) - hs_, je + hs_, [=] __device__(int i, int j) {
          const long n = dir ? g.v2(i, j) : g.u2(i, j);
          const double C1_3 = 1.0 / 3.0;
          if (g.tripolar_n && j > je) {      // reversed polarity in the tripolar halo regions :4036-4041, :4059-4064
            double t = B.FA_EE[n]; B.FA_EE[n] = B.FA_WW[n]; B.FA_WW[n] = t;
            t = B.FA_E0[n]; B.FA_E0[n] = B.FA_W0[n]; B.FA_W0[n] = t;
            t = B.uBT_EE[n]; B.uBT_EE[n] = B.uBT_WW[n]; B.uBT_WW[n] = t;
          }
          const double ee = 1.0 * B.uBT_EE[n], ww = 1.0 * B.uBT_WW[n];
          B.uBT_EE[n] = ee; B.uBT_WW[n] = ww;
          B.uh_EE[n] = ee * (C1_3 * (2.0 * B.FA_E0[n] + B.FA_EE[n]));
          B.uh_WW[n] = ww * (C1_3 * (2.0 * B.FA_W0[n] + B.FA_WW[n]));
          double cw = 0.0, ce = 0.0;
          if (fabs(ww) > 0.0) cw = (C1_3 * (B.FA_WW[n] - B.FA_W0[n])) / (ww * ww);
          if (fabs(ee) > 0.0) ce = (C1_3 * (B.FA_EE[n] - B.FA_E0[n])) / (ee * ee);
          B.uh_crvW[n] = cw; B.uh_crvE[n] = ce;
        });
      }
    };
    if (early_btcl_pass) {
      if (int rc = pass({{BU.uBT_EE, PU}, {BV.uBT_EE, PV}, {BU.uBT_WW, PU}, {BV.uBT_WW, PV}, {BU.FA_EE, PUs}, {BV.FA_EE, PVs},      // :4015-4022
                         {BU.FA_E0, PUs}, {BV.FA_E0, PVs}, {BU.FA_W0, PUs}, {BV.FA_W0, PVs}, {BU.FA_WW, PUs}, {BV.FA_WW, PVs}})) return rc;
      btcl_derive(hs);
    } else {
      btcl_derive(0);
    }
    // (with ADJUST_BT_CONT the external velocities of the specified faces are inverted from the fits as they are here, before the
    // adjustment: the place of set_up_BT_OBC in the reference)
    if (apply_OBCs && early_btcl_pass) { if (int rc = setup_specified()) return rc; }
  } else if (nonlin_cont) {   // :1137-1138
    face_areas_eta(s, 1);
  } else {   // find_face_areas :4297-4310, halo 1
    const double Z_to_H = g.Z_to_H, Zr = cs->Z_ref;
    launch2d(s, is - 2, ie + 1, js - 1, je + 1, [=] __device__(int I, int j) {
      const double H1 = (g.bathyT[g.h2(I, j)] + Zr) * Z_to_H, H2 = (g.bathyT[g.h2(I + 1, j)] + Zr) * Z_to_H;
      double D = 0.0;
      if ((H1 > 0.0) && (H2 > 0.0)) D = g.dy_Cu[g.u2(I, j)] * (2.0 * H1 * H2) / (H1 + H2);
      w.Datu[g.u2(I, j)] = D;
    });
    launch2d(s, is - 1, ie + 1, js - 2, je + 1, [=] __device__(int i, int J) {
      const double H1 = (g.bathyT[g.h2(i, J)] + Zr) * Z_to_H, H2 = (g.bathyT[g.h2(i, J + 1)] + Zr) * Z_to_H;
      double D = 0.0;
      if ((H1 > 0.0) && (H2 > 0.0)) D = g.dx_Cv[g.v2(i, J)] * (2.0 * H1 * H2) / (H1 + H2);
      w.Datv[g.v2(i, J)] = D;
    });
  }

  // ---- the vertical sums :1035-1372, :1505-1541
  hipLaunchKernelGGL(bt_pre_face_kernel<0>, grid2d(is - 1, ie, js, je), dim3(64, 4), 0, s, g, w, p, c.frhatu, dvru, dUc, dpb, duh0, duu0,
                     dU, dbu, dtx, dtbx, c.IDatu);
  hipLaunchKernelGGL(bt_pre_face_kernel<1>, grid2d(is, ie, js - 1, je), dim3(64, 4), 0, s, g, w, p, c.frhatv, dvrv, dVc, dpb, dvh0, dvv0,
                     dV, dbv, dty, dtby, c.IDatv);

  if (apply_OBCs) {
    // :1089-1110: the summed gravity of the cell inside a segment projected across it, segment by segment
    for (int n = 0; n < obc->number_of_segments; n++) {
      const mom6hip_obc_segment_t &S = obc->segment[n];
      if (!S.on_pe) continue;
      const int I = S.IsdB, J = S.JsdB, Isq = is - 1, Ieq = ie, Jsq = js - 1, Jeq = je;
      const int dirn = S.direction;
      if (S.is_N_or_S && (J >= Jsq - 1) && (J <= Jeq + 1)) {
        M6_REQUIRE(J >= g.jsd && J + 1 <= g.jed && S.isd >= g.isd && S.ied <= g.ied, "btstep: OBC segment %d lies outside the data domain", n + 1);
        launch2d(s, std::max(Isq - 1, S.isd), std::min(Ieq + 2, S.ied), J, J, [=] __device__(int i, int j) {
          if (dirn == MOM6HIP_OBC_DIRECTION_N) w.gtot_S[g.h2(i, j + 1)] = w.gtot_S[g.h2(i, j)];
          else w.gtot_N[g.h2(i, j)] = w.gtot_N[g.h2(i, j + 1)];
        });
      } else if (S.is_E_or_W && (I >= Isq - 1) && (I <= Ieq + 1)) {
        M6_REQUIRE(I >= g.isd && I + 1 <= g.ied && S.jsd >= g.jsd && S.jed <= g.jed, "btstep: OBC segment %d lies outside the data domain", n + 1);
        launch2d(s, I, I, std::max(Jsq - 1, S.jsd), std::min(Jeq + 2, S.jed), [=] __device__(int i, int j) {
          if (dirn == MOM6HIP_OBC_DIRECTION_E) w.gtot_W[g.h2(i + 1, j)] = w.gtot_W[g.h2(i, j)];
          else w.gtot_E[g.h2(i, j)] = w.gtot_E[g.h2(i + 1, j)];
        });
      }
    }
    // :1289-1291: the velocities before the first time step
    M6_HIP(hipMemcpyAsync(w.ubt_first, w.ubt, sz.u2, hipMemcpyDeviceToDevice, s));
    M6_HIP(hipMemcpyAsync(w.vbt_first, w.vbt, sz.v2, hipMemcpyDeviceToDevice, s));
  }

  // ---- uhbt0, vhbt0 :1165-1252
  if (add_uh0) {
    if (use_BT_cont && cs->adjust_BT_cont) {   // adjust_local_BT_cont_types :4085 (dt = 1)
      if (int rc = pass({{w.ubt0, PU}, {w.vbt0, PV}, {w.uhbtS, PU}, {w.vhbtS, PV}})) return rc;
      for (int dir = 0; dir < 2; dir++) {
        const Btcl B = dir ? w.BV : w.BU;
        const double *ubt = dir ? w.vbt0 : w.ubt0, *uhbt = dir ? w.vhbtS : w.uhbtS;
        launch2d(s, (dir ? is : is - 1) - hs, ie + hs, (dir ? js - 1 : js) - hs, je + hs, [=] __device__(int i, int j) {
          const long n = dir ? g.v2(i, j) : g.u2(i, j);
          const double ub = ubt[n], uh = uhbt[n], dtl = 1.0;
          if ((dtl * ub > B.uBT_WW[n]) && (dtl * uh > B.uh_WW[n])) {
            B.uBT_WW[n] = dtl * ub;
            if (3.0 * uh < 2.0 * ub * B.FA_W0[n]) {
              B.uh_crvW[n] = (uh - ub * B.FA_W0[n]) / ((dtl * dtl) * ((ub * ub) * ub));
            } else {
              B.FA_W0[n] = 1.5 * uh / ub;
              B.uh_crvW[n] = -0.5 * uh / ((dtl * dtl) * ((ub * ub) * ub));
            }
            B.uh_WW[n] = dtl * uh;
          } else if ((dtl * ub < B.uBT_EE[n]) && (dtl * uh < B.uh_EE[n])) {
            B.uBT_EE[n] = dtl * ub;
            if (3.0 * uh < 2.0 * ub * B.FA_E0[n]) {
              B.uh_crvE[n] = (uh - ub * B.FA_E0[n]) / ((dtl * dtl) * ((ub * ub) * ub));
            } else {
              B.FA_E0[n] = 1.5 * uh / ub;
              B.uh_crvE[n] = -0.5 * uh / ((dtl * dtl) * ((ub * ub) * ub));
            }
            B.uh_EE[n] = dtl * uh;
          }
        });
      }
    }
    launch2d(s, is - 1, ie, js, je, [=] __device__(int I, int j) {
      const long n = g.u2(I, j);
      w.uhbt0[n] = w.uhbtS[n] - (use_BT_cont ? find_uhbt(w.ubt0[n], w.BU, n) : w.Datu[n] * w.ubt0[n]);
    });
    launch2d(s, is, ie, js - 1, je, [=] __device__(int i, int J) {
      const long n = g.v2(i, J);
      w.vhbt0[n] = w.vhbtS[n] - (use_BT_cont ? find_uhbt(w.vbt0[n], w.BV, n) : w.Datv[n] * w.vbt0[n]);
    });
    if (apply_u_OBCs) launch2d(s, is - 1, ie, js, je, [=] __device__(int I, int j) { if (w.obc_u[g.u2(I, j)]) w.uhbt0[g.u2(I, j)] = 0.0; });      // :1236-1247
    if (apply_v_OBCs) launch2d(s, is, ie, js - 1, je, [=] __device__(int i, int J) { if (w.obc_v[g.v2(i, J)]) w.vhbt0[g.v2(i, J)] = 0.0; });
  }

  // ---- weighted Coriolis parameters :1421-1458
  {
    const int sad = cs->Sadourny;
    launch2d(s, isvf - 1, ievf + 1, jsvf - 1, jevf, [=] __device__(int i, int j) {
      const double q00 = w.q[g.q2(i, j)], qm0 = w.q[g.q2(i - 1, j)];
      if (sad) {
        w.amer[g.u2(i - 1, j)] = w.DCor_u[g.u2(i - 1, j)] * qm0;
        w.bmer[g.u2(i, j)] = w.DCor_u[g.u2(i, j)] * q00;
        w.cmer[g.u2(i, j + 1)] = w.DCor_u[g.u2(i, j + 1)] * q00;
        w.dmer[g.u2(i - 1, j + 1)] = w.DCor_u[g.u2(i - 1, j + 1)] * qm0;
      } else {
        const double qmm = w.q[g.q2(i - 1, j - 1)], q0m = w.q[g.q2(i, j - 1)], q0p = w.q[g.q2(i, j + 1)], qmp = w.q[g.q2(i - 1, j + 1)];
        w.amer[g.u2(i - 1, j)] = w.DCor_u[g.u2(i - 1, j)] * ((q00 + qmm) + qm0) / 3.0;
        w.bmer[g.u2(i, j)] = w.DCor_u[g.u2(i, j)] * (q00 + (qm0 + q0m)) / 3.0;
        w.cmer[g.u2(i, j + 1)] = w.DCor_u[g.u2(i, j + 1)] * (q00 + (qm0 + q0p)) / 3.0;
        w.dmer[g.u2(i - 1, j + 1)] = w.DCor_u[g.u2(i - 1, j + 1)] * ((q00 + qmp) + qm0) / 3.0;
      }
    });
    launch2d(s, isvf - 1, ievf, jsvf - 1, jevf + 1, [=] __device__(int i, int j) {
      const long n = g.u2(i, j);
      const double q00 = w.q[g.q2(i, j)], q0m = w.q[g.q2(i, j - 1)];
      if (sad) {
        w.azon[n] = w.DCor_v[g.v2(i + 1, j)] * q00;
        w.bzon[n] = w.DCor_v[g.v2(i, j)] * q00;
        w.czon[n] = w.DCor_v[g.v2(i, j - 1)] * q0m;
        w.dzon[n] = w.DCor_v[g.v2(i + 1, j - 1)] * q0m;
      } else {
        const double qp0 = w.q[g.q2(i + 1, j)], qm0 = w.q[g.q2(i - 1, j)], qmm = w.q[g.q2(i - 1, j - 1)], qpm = w.q[g.q2(i + 1, j - 1)];
        w.azon[n] = w.DCor_v[g.v2(i + 1, j)] * (q00 + (qp0 + q0m)) / 3.0;
        w.bzon[n] = w.DCor_v[g.v2(i, j)] * (q00 + (qm0 + q0m)) / 3.0;
        w.czon[n] = w.DCor_v[g.v2(i, j - 1)] * ((q00 + qmm) + q0m) / 3.0;
        w.dzon[n] = w.DCor_v[g.v2(i + 1, j - 1)] * ((q00 + qpm) + q0m) / 3.0;
      }
    });
  }

  // ---- pass_gtot, pass_ubt_Cor :1460-1476 ; Cor_ref :1478-1490
  if (use_BT_cont && !early_btcl_pass) {
    const Btcl BU = w.BU, BV = w.BV;
    if (int rc = pass({{w.gtot_E, PH}, {w.gtot_N, PH}, {w.gtot_W, PH}, {w.gtot_S, PH}, {w.ubt_Cor, PU}, {w.vbt_Cor, PV},
                       {BU.uBT_EE, PU}, {BV.uBT_EE, PV}, {BU.uBT_WW, PU}, {BV.uBT_WW, PV}, {BU.FA_EE, PUs}, {BV.FA_EE, PVs},
                       {BU.FA_E0, PUs}, {BV.FA_E0, PVs}, {BU.FA_W0, PUs}, {BV.FA_W0, PVs}, {BU.FA_WW, PUs}, {BV.FA_WW, PVs}})) return rc;
    btcl_derive(hs);
  } else {
    if (int rc = pass({{w.gtot_E, PH}, {w.gtot_N, PH}, {w.gtot_W, PH}, {w.gtot_S, PH}, {w.ubt_Cor, PU}, {w.vbt_Cor, PV}})) return rc;
  }
  if (g.tripolar_n) {      // ua_polarity / va_polarity < 0 in the halo rows beyond the fold :1471-1475
    const int jtop = (jevf + 1 < g.jed) ? jevf + 1 : g.jed;
    launch2d(s, isvf - 1, ievf + 1, je + 1, jtop, [=] __device__(int i, int j) {
      const long n = g.h2(i, j);
      double t = w.gtot_E[n]; w.gtot_E[n] = w.gtot_W[n]; w.gtot_W[n] = t;
      t = w.gtot_N[n]; w.gtot_N[n] = w.gtot_S[n]; w.gtot_S[n] = t;
    });
  }
  launch2d(s, is - 1, ie, js, je, [=] __device__(int i, int j) {
    const long n = g.u2(i, j);
    w.Cor_ref_u[n] = ((w.azon[n] * w.vbt_Cor[g.v2(i + 1, j)] + w.czon[n] * w.vbt_Cor[g.v2(i, j - 1)]) +
                      (w.bzon[n] * w.vbt_Cor[g.v2(i, j)] + w.dzon[n] * w.vbt_Cor[g.v2(i + 1, j - 1)]));
  });
  launch2d(s, is, ie, js - 1, je, [=] __device__(int i, int j) {
    w.Cor_ref_v[g.v2(i, j)] = -1.0 * ((w.amer[g.u2(i - 1, j)] * w.ubt_Cor[g.u2(i - 1, j)] + w.cmer[g.u2(i, j + 1)] * w.ubt_Cor[g.u2(i, j + 1)]) +
                                      (w.bmer[g.u2(i, j)] * w.ubt_Cor[g.u2(i, j)] + w.dmer[g.u2(i - 1, j + 1)] * w.ubt_Cor[g.u2(i - 1, j + 1)]));
  });

  // ---- eta_src :1626-1628 and the transport of the initial velocities (what :1884-1891 evaluates after the first pass)
  {
    const double Instep = p.Instep;
    double *cor = c.eta_cor;
    if (cs->bound_BT_corr) {      // BOUND_BT_CORRECTION with BT_CONT_CORR_BOUNDS :1587-1615 (the use_BT_cont branch): limits the state eta_cor
      M6_REQUIRE(use_BT_cont && cs->maxCFL_BT_cont > 0.0, "btstep: BOUND_BT_CORRECTION needs the BT_cont argument (BT_CONT_CORR_BOUNDS) and MAXCFL_BT_CONT > 0");
      const double cfl_Idt = cs->maxCFL_BT_cont * (1.0 / dt), Z_to_H = g.Z_to_H;
      launch2d(s, is, ie, js, je, [=] __device__(int i, int j) {
        const long n = g.h2(i, j);
        if (!(g.mask2dT[n] > 0.0)) return;
        const double ec = cor[n];
        if (ec > 0.0) {
          const double u_max_cor = g.dxT[n] * cfl_Idt, v_max_cor = g.dyT[n] * cfl_Idt;
          const double eta_cor_max = dt * (g.IareaT[n] *
                   (((find_uhbt(u_max_cor, w.BU, g.u2(i, j)) + w.uhbt0[g.u2(i, j)]) -
                     (find_uhbt(-u_max_cor, w.BU, g.u2(i - 1, j)) + w.uhbt0[g.u2(i - 1, j)])) +
                    ((find_uhbt(v_max_cor, w.BV, g.v2(i, j)) + w.vhbt0[g.v2(i, j)]) -
                     (find_uhbt(-v_max_cor, w.BV, g.v2(i, j - 1)) + w.vhbt0[g.v2(i, j - 1)]))));
          cor[n] = m6::min2(ec, m6::max2(0.0, eta_cor_max));
        } else {
          const double Htot = g.bathyT[n] * Z_to_H + w.eta[n];
          cor[n] = m6::max2(ec, -m6::max2(0.0, Htot));
        }
      });
    }
    launch2d(s, is, ie, js, je, [=] __device__(int i, int j) { w.eta_src[g.h2(i, j)] = g.mask2dT[g.h2(i, j)] * (Instep * cor[g.h2(i, j)]); });
    launch2d(s, is - 1, ie, js, je, [=] __device__(int I, int j) {
      const long n = g.u2(I, j);
      w.uhbtp[n] = (use_BT_cont ? find_uhbt(w.ubt[n], w.BU, n) : w.Datu[n] * w.ubt[n]) + w.uhbt0[n];
    });
    launch2d(s, is, ie, js - 1, je, [=] __device__(int i, int J) {
      const long n = g.v2(i, J);
      w.vhbtp[n] = (use_BT_cont ? find_uhbt(w.vbt[n], w.BV, n) : w.Datv[n] * w.vbt[n]) + w.vhbt0[n];
    });
  }

  // ---- pass_eta_bt_rem, pass_Dat_uv, pass_force_hbt0_Cor_ref :1672-1697
  {
    std::vector<double *> f; std::vector<int32_t> pos;
    auto add = [&](double *a, int ps) { f.push_back(a); pos.push_back(ps); };
    if (interp) { add(w.eta_PF_1, PH); add(w.d_eta_PF, PH); } else add(w.eta_PF, PH);
    add(w.eta_src, PH); add(w.bt_rem_u, PUs); add(w.bt_rem_v, PVs);      // :847, :859: scalar pairs
    if (!use_BT_cont) { add(w.Datu, PUs); add(w.Datv, PVs); }
    add(w.BT_force_u, PU); add(w.BT_force_v, PV);
    if (add_uh0) { add(w.uhbt0, PU); add(w.vhbt0, PV); }
    add(w.Cor_ref_u, PU); add(w.Cor_ref_v, PV);
    std::vector<int32_t> nk(f.size(), 1);
    if (int rc = m6::group_pass(ctx, f.data(), pos.data(), nk.data(), (int)f.size())) return rc;
  }

  // the filtered sums accumulate straight into the outputs: ubtav, vbtav (CS) and uhbtav, vhbtav
  launch2d(s, is - 1, ie, js, je, [=] __device__(int I, int j) { c.ubtav[g.u2(I, j)] = 0.0; duhbtav[g.u2(I, j)] = 0.0; });
  launch2d(s, is, ie, js - 1, je, [=] __device__(int i, int J) { c.vbtav[g.v2(i, J)] = 0.0; dvhbtav[g.v2(i, J)] = 0.0; });
  M6_HIP(hipGetLastError());

  // ---- set_up_BT_OBC :3172-3365 (Boussinesq, BTHALO = 0), here because the fits of BT_cont are complete on the widened range only now;
  // nothing before the time steps reads what it sets
  if (apply_OBCs) {
    if (!early_btcl_pass) { if (int rc = setup_specified()) return rc; }
    const int halo = ievf - ie, s_is = is - halo, s_ie = ie + halo, s_js = js - halo, s_je = je + halo;
    const double g_prime1 = g.g_Earth, H_to_Z = g.H_to_Z, Z_ref = cs->Z_ref;      // GV%g_prime(1): GFS, default the gravity of the Earth
    for (int d = 0; d < 2; d++) {
      if (!(d ? apply_v_OBCs : apply_u_OBCs)) continue;
      double *o_outer = d ? w.ob_vbt_outer : w.ob_ubt_outer, *o_dZ = d ? w.ob_dZ_v : w.ob_dZ_u,
             *o_Cg = d ? w.ob_Cg_v : w.ob_Cg_u, *o_SSH = d ? w.ob_SSH_v : w.ob_SSH_u;
      const int32_t *code = d ? w.obc_v : w.obc_u;
      launch2d(s, d ? s_is : s_is - 1, s_ie, d ? s_js - 1 : s_js, s_je, [=] __device__(int i, int j) {
        const long f = d ? g.v2(i, j) : g.u2(i, j);
        const int c = code[f];
        if (!c || (c & OB_SPEC)) return;      // (the specified faces: setup_specified)
        {
          const long cc = (c & OB_PLUS) ? g.h2(i, j) : (d ? g.h2(i, j + 1) : g.h2(i + 1, j));
          const double dZ = g.bathyT[cc] + H_to_Z * w.eta[cc];
          o_dZ[f] = dZ;
          o_Cg[f] = sqrt(1.0 * g_prime1 * dZ);
        }
      });
      if (d ? obc->Flather_v_BCs_exist_globally : obc->Flather_u_BCs_exist_globally)
        for (int n = 0; n < obc->number_of_segments; n++) {
          const mom6hip_obc_segment_t &S = obc->segment[n];
          if (!((d ? S.is_N_or_S : S.is_E_or_W) && S.Flather)) continue;
          M6_REQUIRE(S.normal_vel_bt && S.SSH, "btstep: segment %d is a Flather segment: normal_vel_bt and SSH are required", n + 1);
          const int a0 = d ? S.isd : S.IsdB, a1 = d ? S.ied : S.IedB, b0 = d ? S.JsdB : S.jsd, b1 = d ? S.JedB : S.jed;
          const long na = a1 - a0 + 1, nb = b1 - b0 + 1;
          const double *nv = st.in(S.normal_vel_bt, (size_t)na * nb * 8), *ssh = st.in(S.SSH, (size_t)na * nb * 8);
          M6_REQUIRE(!st.failed() && nv && ssh, "btstep: staging failed");
          launch2d(s, a0, a1, b0, b1, [=] __device__(int a, int b) {
            const long f = d ? g.v2(a, b) : g.u2(a, b), q = (a - a0) + na * (b - b0);
            o_outer[f] = nv[q];
            o_SSH[f] = ssh[q] + Z_ref;
          });
        }
    }
    // do_group_pass(BT_OBC%pass_uv | pass_uhvh | pass_eta_outer | pass_h | pass_cg) :3359-3363
    if (int rc = pass({{w.ob_ubt_outer, PU}, {w.ob_vbt_outer, PV}, {w.ob_uhbt, PU}, {w.ob_vhbt, PV}, {w.ob_SSH_u, PUs}, {w.ob_SSH_v, PVs},
                       {w.ob_dZ_u, PUs}, {w.ob_dZ_v, PVs}, {w.ob_Cg_u, PUs}, {w.ob_Cg_v, PVs}})) return rc;
  }

  // ---- the barotropic time steps :1812-2462
  // (replayed from hipGraphs: see graph_of below)
  // the valid range of step n and whether a group pass precedes it (:1842-1861): a function of n alone
  struct StepRange { int isv, iev, jsv, jev; bool pass_first; };
  std::vector<StepRange> rng(nt + 1);
  {
    int isv = is, iev = ie, jsv = js, jev = je;
    for (int n = 1; n <= nt; n++) {
      bool pf = false;
      if ((iev - stencil < ie) || (jev - stencil < je)) { pf = true; isv = isvf; iev = ievf; jsv = jsvf; jev = jevf; }
      else { isv += stencil; iev -= stencil; jsv += stencil; jev -= stencil; }
      rng[n] = {isv, iev, jsv, jev, pf};
    }
  }
  // MOM6HIP_BT_FUSED=1: one fused kernel a step (bt_step_fused_kernel; not with open boundaries or face areas that follow eta).  Bit-exact
  // and slower than the four kernels at every tile height measured (profiles/r05_experiments.txt section 2): not the default.  Its five
  // alternating fields: step n reads set A (the Work arrays) when n is odd, set B when it is even.  (Read at every call: tests switch it.)
  const char *fused_env = getenv("MOM6HIP_BT_FUSED");
  const bool fused = fused_env && atoi(fused_env) == 1 && !apply_OBCs && !nonlin_update;
  auto eta_at = [&](int n) { return (fused && n % 2 == 0) ? alt_eta : w.eta; };        // the set step n reads (and a pass before it updates)
  auto do_pass = [&](int n) -> int {
    if (fused && n % 2 == 0) return pass({{alt_eta, PH}, {alt_ubt, PU}, {alt_vbt, PV}, {alt_uhbtp, PU}, {alt_vhbtp, PV}});
    return pass({{w.eta, PH}, {w.ubt, PU}, {w.vbt, PV}, {w.uhbtp, PU}, {w.vhbtp, PV}});
  };
  (void)eta_at;
  // the kernels of steps n0 .. n1 (no group pass)
  auto run_steps = [&](hipStream_t st, int n0, int n1) {
    for (int n = n0; n <= n1; n++) {
      const int isv = rng[n].isv, iev = rng[n].iev, jsv = rng[n].jsv, jev = rng[n].jev;
      const double wt_end = n * p.Instep;
      if (fused) {
        BtFused a;
        const bool odd = (n % 2) == 1;
        a.eta_in = odd ? w.eta : alt_eta; a.ubt_in = odd ? w.ubt : alt_ubt; a.vbt_in = odd ? w.vbt : alt_vbt;
        a.uhbtp_in = odd ? w.uhbtp : alt_uhbtp; a.vhbtp_in = odd ? w.vhbtp : alt_vhbtp;
        a.eta_out = odd ? alt_eta : w.eta; a.ubt_out = odd ? alt_ubt : w.ubt; a.vbt_out = odd ? alt_vbt : w.vbt;
        a.uhbtp_out = odd ? alt_uhbtp : w.uhbtp; a.vhbtp_out = odd ? alt_vhbtp : w.vhbtp;
        a.isv = isv; a.iev = iev; a.jsv = jsv; a.jev = jev;
        a.wt_accel = wt_accel[n]; a.wt_accel2 = wt_accel2[n]; a.wt_trans = wt_trans[n]; a.wt_eta = wt_eta[n]; a.wt_end = wt_end;
        a.ubt_sum = c.ubtav; a.uhbt_sum = duhbtav; a.vbt_sum = c.vbtav; a.vhbt_sum = dvhbtav;
        const dim3 grid((iev - isv + 3 + BTF_TX - 1) / BTF_TX, (jev - jsv + 3 + BTF_TY - 1) / BTF_TY);
        const bool v_first = ((n + ctx->host.first_direction) % 2) == 1;
        if (v_first) hipLaunchKernelGGL(bt_step_fused_kernel<true>, grid, dim3(64, 4), 0, st, g, w, p, a);
        else hipLaunchKernelGGL(bt_step_fused_kernel<false>, grid, dim3(64, 4), 0, st, g, w, p, a);
        continue;
      }
      if (nonlin_update && (n > 1) && ((n - 1) % cs->Nonlin_cont_update_period == 0)) {      // :1852-1856
        face_areas_eta(st, 1 + iev - ie);
        // the predictor transports of this step were formed by the velocity kernels of the previous one, with the old face
        // areas: the reference forms them after the refresh (:1896-1903)
        if (!p.project_velocity) {
          launch2d(st, isv - 2, iev + 1, jsv - 1, jev + 1, [=] __device__(int I, int j) {
            const long q = g.u2(I, j);
            w.uhbtp[q] = w.Datu[q] * w.ubt[q] + w.uhbt0[q];
          });
          launch2d(st, isv - 1, iev + 1, jsv - 2, jev + 1, [=] __device__(int i, int J) {
            const long q = g.v2(i, J);
            w.vhbtp[q] = w.Datv[q] * w.vbt[q] + w.vhbt0[q];
          });
        }
      }
      hipLaunchKernelGGL(bt_eta_pred_kernel, grid2d(isv - 1, iev + 1, jsv - 1, jev + 1), dim3(64, 4), 0, st, g, w, p, isv - 1, iev + 1,
                         jsv - 1, jev + 1, wt_accel2[n]);
      const bool v_first = ((n + ctx->host.first_direction) % 2) == 1;
      if (apply_OBCs) {      // :1938-1947: the velocities before this time step (the faces inside a Flather face are read by bt_obc_kernel)
        (void)hipMemcpyAsync(w.ubt_old, w.ubt, sz.u2, hipMemcpyDeviceToDevice, st);
        (void)hipMemcpyAsync(w.vbt_old, w.vbt, sz.v2, hipMemcpyDeviceToDevice, st);
      }
      for (int ps = 0; ps < 2; ps++) {
        const bool do_v = (ps == 0) ? v_first : !v_first;
        if (do_v) {
          const int i0 = v_first ? isv - 1 : isv, i1 = v_first ? iev + 1 : iev;
          hipLaunchKernelGGL(bt_vbt_kernel, grid2d(i0, i1, jsv - 1, jev), dim3(64, 4), 0, st, g, w, p, i0, i1, jsv - 1, jev, wt_accel[n],
                             wt_trans[n], wt_end, c.vbtav, dvhbtav);
        } else {
          const int j0 = v_first ? jsv : jsv - 1, j1 = v_first ? jev : jev + 1;
          hipLaunchKernelGGL(bt_ubt_kernel, grid2d(isv - 1, iev, j0, j1), dim3(64, 4), 0, st, g, w, p, isv - 1, iev, j0, j1, wt_accel[n],
                             wt_trans[n], wt_end, c.ubtav, duhbtav);
        }
      }
      if (apply_OBCs) {      // apply_velocity_OBCs :2357-2395 (halo = iev - ie)
        const int halo = iev - ie;
        if (apply_u_OBCs)
          hipLaunchKernelGGL(bt_obc_kernel<0>, grid2d(is - halo - 1, ie + halo, js - halo, je + halo), dim3(64, 4), 0, st, g, w, p, is - halo - 1,
                             ie + halo, js - halo, je + halo, cs->bebt, wt_trans[n], wt_vel[n], c.ubtav, duhbtav);
        if (apply_v_OBCs)
          hipLaunchKernelGGL(bt_obc_kernel<1>, grid2d(is - halo, ie + halo, js - halo - 1, je + halo), dim3(64, 4), 0, st, g, w, p, is - halo,
                             ie + halo, js - halo - 1, je + halo, cs->bebt, wt_trans[n], wt_vel[n], c.vbtav, dvhbtav);
      }
      hipLaunchKernelGGL(bt_eta_kernel, grid2d(isv, iev, jsv, jev), dim3(64, 4), 0, st, g, w, p, isv, iev, jsv, jev, wt_eta[n]);
    }
  };
  // steps n0 .. n1 with their group passes, enqueued one by one on `st`
  auto run_loop = [&](hipStream_t st, int n0, int n1) -> int {
    for (int n = n0; n <= n1; n++) {
      if (rng[n].pass_first) { if (int rc = do_pass(n)) return rc; }
      run_steps(st, n, n);
    }
    return 0;
  };
  // A hipGraph of steps n0 .. n1, keyed on everything that is baked into its nodes (pointers, ranges, weights) and cached in
  // the context.  with_passes: the wrap kernels of a one-tile group pass are captured too.
  std::string base_key;
  {
    auto add = [&](const void *q, size_t nbytes) { base_key.append((const char *)q, nbytes); };
    add(&w, sizeof(w)); add(&p, sizeof(p)); add(&nt, sizeof(nt)); add(&ctx->host.first_direction, sizeof(int32_t));
    { const void *fk[2] = {fused ? (const void *)alt_eta : nullptr, fused ? (const void *)alt_ubt : nullptr}; add(fk, sizeof(fk)); }
    { const double ob_key[3] = {apply_u_OBCs ? 1.0 : 0.0, apply_v_OBCs ? 1.0 : 0.0, cs->bebt}; add(ob_key, sizeof(ob_key)); }
    { const int32_t nl[2] = {nonlin_update ? 1 : 0, cs->Nonlin_cont_update_period}; add(nl, sizeof(nl)); }
    add(&c.ubtav, sizeof(double *)); add(&c.vbtav, sizeof(double *)); add(&duhbtav, sizeof(double *)); add(&dvhbtav, sizeof(double *));
    add(wt_vel.data(), sizeof(double) * wt_vel.size()); add(wt_eta.data(), sizeof(double) * wt_eta.size());
    add(wt_trans.data(), sizeof(double) * wt_trans.size()); add(wt_accel.data(), sizeof(double) * wt_accel.size());
    add(wt_accel2.data(), sizeof(double) * wt_accel2.size());
  }
  auto graph_of = [&](int n0, int n1, bool with_passes, hipGraphExec_t *out) -> int {
    std::string key = base_key;
    key.append((const char *)&n0, sizeof(int)); key.append((const char *)&n1, sizeof(int)); key.push_back(with_passes ? 1 : 0);
    for (auto &e : ctx->bt_graphs) if (e.first == key) { *out = (hipGraphExec_t)e.second; return 0; }
    if (!ctx->cap_stream) M6_HIP(hipStreamCreateWithFlags(&ctx->cap_stream, hipStreamNonBlocking));
    hipStream_t saved = ctx->stream;
    ctx->stream = ctx->cap_stream;      // the wrap kernels of the group pass launch on the context's stream
    hipError_t e0 = hipStreamBeginCapture(ctx->cap_stream, hipStreamCaptureModeRelaxed);
    int rc = 1;
    if (e0 == hipSuccess) {
      if (with_passes) rc = run_loop(ctx->cap_stream, n0, n1);
      else { run_steps(ctx->cap_stream, n0, n1); rc = 0; }
    }
    hipGraph_t graph = nullptr;
    hipError_t e1 = (e0 == hipSuccess) ? hipStreamEndCapture(ctx->cap_stream, &graph) : e0;
    ctx->stream = saved;
    M6_REQUIRE(rc == 0 && e1 == hipSuccess && graph, "btstep: capturing the barotropic subcycle as a hipGraph failed (%s)",
               hipGetErrorString(e1));
    { size_t nn = 0; if (hipGraphGetNodes(graph, nullptr, &nn) == hipSuccess) ctx->bt_graph_nodes_last = (long)nn; }
    hipGraphExec_t exec = nullptr;
    hipError_t e2 = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    M6_REQUIRE(e2 == hipSuccess && exec, "btstep: hipGraphInstantiate failed (%s)", hipGetErrorString(e2));
    if (ctx->bt_graphs.size() >= 256) {      // drop the oldest
      (void)hipGraphExecDestroy((hipGraphExec_t)ctx->bt_graphs.front().second);
      ctx->bt_graphs.erase(ctx->bt_graphs.begin());
    }
    ctx->bt_graphs.push_back({key, (void *)exec});
    ctx->bt_graph_captures++;
    *out = exec;
    return 0;
  };
  // One tile: the whole subcycle (4 kernels per barotropic step + the wrap kernels of the group pass every num_cycles
  // steps; ~100-250 nodes) is ONE graph, replayed as a single launch.  With neighbours the group pass leaves the stream
  // (RCCL on the communication stream, or the host's callback), so the steps between two passes -- num_cycles of them,
  // 4 kernels each, a few microseconds of work per kernel on a tile of an 8-GPU run -- are one graph per segment and
  // the passes are enqueued between the graph launches: a launch per segment instead of one per kernel.
  static const bool graph_off = getenv("MOM6HIP_BT_GRAPH") && atoi(getenv("MOM6HIP_BT_GRAPH")) == 0;
  if (graph_off) {
    if (int rc = run_loop(s, 1, nt)) return rc;
  } else if (m6::multi_tile(ctx) && [&] {      // more segments than the graph cache keeps for the two btstep calls of a step
               int nseg = 0;                   // would recapture and re-instantiate every graph on every step: launch kernel by kernel
               for (int n = 1; n <= nt; n++) if (n == 1 || rng[n].pass_first) nseg++;
               return nseg > 100;
             }()) {
    if (int rc = run_loop(s, 1, nt)) return rc;
  } else if (m6::multi_tile(ctx)) {
    int n0 = 1;
    while (n0 <= nt) {
      int n1 = n0;
      while (n1 + 1 <= nt && !rng[n1 + 1].pass_first) n1++;
      if (rng[n0].pass_first) { if (int rc = do_pass(n0)) return rc; }
      hipGraphExec_t exec = nullptr;
      if (int rc = graph_of(n0, n1, false, &exec)) return rc;
      M6_HIP(hipGraphLaunch(exec, s));
      ctx->bt_graph_launches++;
      n0 = n1 + 1;
    }
  } else {
    hipGraphExec_t exec = nullptr;
    if (int rc = graph_of(1, nt, true, &exec)) return rc;
    m6::KTimer kt(ctx, MOM6HIP_KT_BT_SUBCYCLE);
    M6_HIP(hipGraphLaunch(exec, s));
    ctx->bt_graph_launches++;
  }
  M6_HIP(hipGetLastError());
  if (fused && nt % 2 == 1) {      // the last step wrote set B: the epilogue (and a caller that looks at the work arrays) reads set A
    M6_HIP(hipMemcpyAsync(w.eta, alt_eta, sz.h2, hipMemcpyDeviceToDevice, s));
    M6_HIP(hipMemcpyAsync(w.ubt, alt_ubt, sz.u2, hipMemcpyDeviceToDevice, s));
    M6_HIP(hipMemcpyAsync(w.vbt, alt_vbt, sz.v2, hipMemcpyDeviceToDevice, s));
    M6_HIP(hipMemcpyAsync(w.uhbtp, alt_uhbtp, sz.u2, hipMemcpyDeviceToDevice, s));
    M6_HIP(hipMemcpyAsync(w.vhbtp, alt_vhbtp, sz.v2, hipMemcpyDeviceToDevice, s));
  }

  // ---- epilogue :2467-2590
  {
    const double dgeo_de = p.dgeo_de;
    launch2d(s, is, ie, js, je, [=] __device__(int i, int j) {
      const long n = g.h2(i, j);
      double ea;
      if (interp) ea = dgeo_de * (0.5 * (w.eta[n] + deta_in[n]) - (w.eta_PF_1[n] + 0.5 * w.d_eta_PF[n]));
      else ea = dgeo_de * (0.5 * (w.eta[n] + deta_in[n]) - w.eta_PF[n]);
      w.e_anom[n] = ea;
      if (find_etaav) detaav[n] = w.eta_sum[n] * 1.0;
      deta_out[n] = w.eta_wtd[n] * 1.0;
    });
    for (int d = 0; d < 2 && apply_OBCs; d++) {      // :2490-2519: e_anom across the faces of the segments, the u faces first
      if (!(d ? apply_v_OBCs : apply_u_OBCs)) continue;
      const int32_t *code = d ? w.obc_v : w.obc_u;
      launch2d(s, d ? is : is - 1, ie, d ? js - 1 : js, je, [=] __device__(int i, int j) {
        const int cde = code[d ? g.v2(i, j) : g.u2(i, j)];
        const long c0 = g.h2(i, j), c1 = d ? g.h2(i, j + 1) : g.h2(i + 1, j);
        if (cde & OB_PLUS) w.e_anom[c1] = w.e_anom[c0];      // OBC_DIRECTION_E | N
        else if (cde & OB_MINUS) w.e_anom[c0] = w.e_anom[c1];
      });
    }
    // pass_etaav, pass_e_anom and pass_ubta_uhbta (:2527-2570) as one group
    if (find_etaav) { if (int rc = pass({{detaav, PH}, {w.e_anom, PH}, {c.ubtav, PU}, {c.vbtav, PV}, {duhbtav, PU}, {dvhbtav, PV}})) return rc; }
    else { if (int rc = pass({{w.e_anom, PH}, {c.ubtav, PU}, {c.vbtav, PV}, {duhbtav, PU}, {dvhbtav, PV}})) return rc; }
    hipLaunchKernelGGL(bt_accel_layer_kernel<0>, grid2d(is - 1, ie, js, je), dim3(64, 4), 0, s, g, w, dpb, dalu, accel_underflow);
    hipLaunchKernelGGL(bt_accel_layer_kernel<1>, grid2d(is, ie, js - 1, je), dim3(64, 4), 0, s, g, w, dpb, dalv, accel_underflow);
    for (int d = 0; d < 2 && apply_OBCs; d++) {      // :2591-2606: the accelerations of the segments' faces from their own velocities
      if (!(d ? apply_v_OBCs : apply_u_OBCs)) continue;
      const int32_t *code = d ? w.obc_v : w.obc_u;
      const double *wtd = d ? w.vbt_wtd : w.ubt_wtd, *first = d ? w.vbt_first : w.ubt_first;
      double *accel = d ? dalv : dalu;
      const long fstr = d ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh;
      const int nk = g.nk;
      launch2d(s, d ? is : is - 1, ie, d ? js - 1 : js, je, [=] __device__(int i, int j) {
        const long f = d ? g.v2(i, j) : g.u2(i, j);
        if (!code[f]) return;
        const double a = (wtd[f] - first[f]) / dt;
        for (int k = 0; k < nk; k++) accel[f + fstr * k] = a;
      });
    }
  }
  M6_HIP(hipGetLastError());
  return st.finish();
}

}  // extern "C"
