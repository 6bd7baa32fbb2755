// vert_friction.hip -- vertvisc_coef (+ find_coupling_coef), vertvisc (+ vertvisc_limit_vel) and vertvisc_remnant of
// src/parameterizations/vertical/MOM_vert_friction.F90 (:1168, :1768, :526, :2259, :1064) as gfx950 kernels.
//
// Everything here is a recurrence in k over one velocity column, independent between columns: one lane per face column,
// lanes along i, so every k-strided access is a contiguous row across the wave.
//   vv_coef_kernel<DIR>    one bottom-up sweep: the thickness at the velocity point (harmonic / arithmetic blend with the
//                          bottom function), the normalised height z_i, and the coupling coefficient of the interface
//                          between the layer just done and the one below it -- two layers of state in registers, no
//                          per-column arrays.  (KV_ML_INVZ2 > 0 adds a top-down sweep first, which leaves its
//                          viscosities in a_u.)
//   vv_solve_kernel<DIR>   the implicit solve: forward elimination writing the modified right-hand side in place and
//                          c1 to a work array, back substitution; the same kernel gives visc_rem (vertvisc_remnant).
//   vv_limit_kernel<DIR>   vertvisc_limit_vel, one thread per point; truncations are counted with an atomic.
// Algorithmic traffic per cell: coef 8 (h) + 4 (vel) + 16 (a, h_vel) [B, both directions: x2 on vel/a/h_vel];
// solve: read a, h_vel, x, write x, c1, read both again, write x.
#include <cmath>

#include "common.hpp"

namespace {

using m6::max2;
using m6::min2;

struct VVPar {      // the scalars of vertvisc_CS the kernels read
  double Hmix, Hmix_stress, Kvml_invZ2, Kv, Hbbl, Kv_extra_bbl, harm_BL_val, maxvel, CFL_trunc, vel_underflow, H_to_RZ, vonKar;
  int bottomdraglaw, harmonic_visc, direct_stress, CFL_based_trunc, dynamic_viscous_ML, nkml;
};

#ifndef VB_LAYERS
#define VB_LAYERS 2
#endif
constexpr int VB = VB_LAYERS;      // layers loaded at once by the column sweeps

struct CoefArgs {
  m6::GridDev g;
  VVPar p;
  const double *vel, *h, *dz, *kv_bbl, *bbl_thick, *Kv_shear;
  double *a, *hv;
  // the surface boundary layer of DYNAMIC_VISCOUS_ML / a bulk mixed layer (find_coupling_coef :2047-2252)
  const double *ustar, *nkml_visc;      // forces%ustar (h points), visc%nkml_visc_u or _v
  double *dzv;                          // scratch: dz_vel of every layer of the column (written bottom-up, read top-down)
  // the velocity as the increment the RK2 step applies just before the call (MOM_dynamics_split_RK2.F90:582-589, :667-676, :930-939):
  // vel = mask * (f_u0 + f_dt * (f_a1 [+ f_a2])) formed here instead of by a sweep of its own; f_u0 = null: vel is read as it is.
  // f_store (the solve's x): where the sweep leaves the velocity for the solve (null: the velocities are not updated)
  const double *f_u0, *f_a1, *f_a2;
  double f_dt;
  double *f_store;
  // open boundaries (the OBC instantiation only): per face, the cell the thicknesses, the depth, Kv_shear and ustar are projected
  // outward from, -1 the first (OBC_DIRECTION_E | N), +1 the second (W | S), 0 not on a segment (m6::obc_side_maps)
  const int32_t *side;
};

// the velocity of layer k of the face column at f2: read, or formed from the step's increment (the reference's expression)
__device__ __forceinline__ double coef_vel(const CoefArgs &A, long n, double mask) {
  if (!A.f_u0) return A.vel[n];
  const double acc = A.f_a2 ? (A.f_a1[n] + A.f_a2[n]) : A.f_a1[n];
  return mask * (A.f_u0[n] + A.f_dt * acc);
}

// vertvisc_coef + find_coupling_coef for the face column (i, j); the caller has checked do_i
template <int DIR, bool OBC = false>
__device__ __forceinline__ void coef_column(const CoefArgs &A, int i, int j) {
  const m6::GridDev &g = A.g;
  const VVPar &P = A.p;
  const long f2 = DIR ? g.v2(i, j) : g.u2(i, j);
  const long c0 = g.h2(i, j), c1 = DIR ? g.h2(i, j + 1) : g.h2(i + 1, j);
  const long hpl = (long)g.nih * g.njh, fpl = DIR ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh;
  const int nz = g.nk;
  const double h_neglect = g.H_subroundoff, dz_neglect = g.dZ_subroundoff;
  const double a_cpl_max = 1.0e37 * g.Z_to_H * 1.0;                 // :1283
  double I_Hbbl = 1.0 / (P.Hbbl + dz_neglect);                      // :1284
  const double I_valBL = (P.harm_BL_val > 0.0) ? 1.0 / P.harm_BL_val : 0.0;
  double kv_bbl = 0.0, bbl_thick = 0.0;
  if (P.bottomdraglaw) {                                            // :1318-1322
    kv_bbl = A.kv_bbl[f2];
    bbl_thick = A.bbl_thick[f2] + dz_neglect;
    I_Hbbl = 1.0 / bbl_thick;
  }
  const double hn = dz_neglect, I_amax = 0.0;                        // find_coupling_coef :1846, :1858
  const double fmask = A.f_u0 ? (DIR ? g.mask2dCv[f2] : g.mask2dCu[f2]) : 0.0;
  auto DZ = [&](long c, int k) { return A.dz ? A.dz[c + hpl * k] : g.H_to_Z * A.h[c + hpl * k]; };
  const int side = OBC ? A.side[f2] : 0;      // :1335-1355 / :1546-1566 (zi_dir)

  // ---- KV_ML_INVZ2: the top-down viscosity profile :1873-1886, parked in a(K) ----
  const bool kvml = P.Kvml_invZ2 > 0.0;
  if (kvml) {
    const double I_Hmix = 1.0 / (P.Hmix + hn);
    double z_t = hn * I_Hmix;
    for (int Kb = 1; Kb < nz; Kb += VB) {
      double b_d0[VB], b_d1[VB];
#pragma unroll
      for (int q = 0; q < VB; q++) {
        const int K = Kb + q;
        b_d0[q] = b_d1[q] = 0.0;
        if (K < nz) { b_d0[q] = DZ(c0, K - 1); b_d1[q] = DZ(c1, K - 1); }
      }
#pragma unroll
      for (int q = 0; q < VB; q++) {
        const int K = Kb + q;
        if (K >= nz) break;
        const double d0 = b_d0[q], d1 = b_d1[q];
        double dz_harm = 2.0 * d0 * d1 / (d0 + d1 + dz_neglect);
        if (OBC && side) dz_harm = (side < 0) ? d0 : d1;
        z_t = z_t + dz_harm * I_Hmix;
        A.a[f2 + fpl * K] = P.Kv + P.Kvml_invZ2 / ((z_t * z_t) * (1.0 + 0.09 * z_t * z_t * z_t * z_t * z_t * z_t));
      }
    }
  }

  // ---- the bottom-up sweep ----
  double Dmin = min2(g.bathyT[c0], g.bathyT[c1]);                    // :1331
  if (OBC && side) Dmin = (side < 0) ? g.bathyT[c0] : g.bathyT[c1];    // :1342, :1349
  double zh = 0.0, zcol0 = -g.bathyT[c0], zcol1 = -g.bathyT[c1];
  double z_i_below = 0.0;        // z_i(k+1)
  double dzv_below = 0.0;        // dz_vel(k+1)
  // The loop is a chain through k of short dependent steps: what a lane waits for is its loads.  VB layers are loaded at
  // once (before any store of the batch, so nothing orders them behind the stores), then worked through in order.
  for (int kb = nz - 1; kb >= 0; kb -= VB) {
    double b_h0[VB], b_h1[VB], b_d0[VB], b_d1[VB], b_vel[VB], b_kvml[VB], b_ksh[VB];
#pragma unroll
    for (int q = 0; q < VB; q++) {
      const int k = kb - q;
      b_h0[q] = b_h1[q] = b_d0[q] = b_d1[q] = b_vel[q] = b_kvml[q] = b_ksh[q] = 0.0;
      if (k >= 0) {
        b_h0[q] = A.h[c0 + hpl * k]; b_h1[q] = A.h[c1 + hpl * k];
        if (A.dz) { b_d0[q] = A.dz[c0 + hpl * k]; b_d1[q] = A.dz[c1 + hpl * k]; }
        b_vel[q] = coef_vel(A, f2 + fpl * k, fmask);
        if (k + 1 < nz) {
          if (kvml) b_kvml[q] = A.a[f2 + fpl * (k + 1)];
          if (A.Kv_shear) {
            b_ksh[q] = 0.5 * (A.Kv_shear[c0 + hpl * (k + 1)] + A.Kv_shear[c1 + hpl * (k + 1)]);
            if (OBC && side) b_ksh[q] = A.Kv_shear[((side < 0) ? c0 : c1) + hpl * (k + 1)];      // :1901-1909, :1917-1925
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < VB; q++) {
    const int k = kb - q;
    if (k < 0) break;
    const double h0 = b_h0[q], h1 = b_h1[q];
    double h_harm = 2.0 * h0 * h1 / (h0 + h1 + h_neglect);            // :1324-1330
    double h_arith = 0.5 * (h1 + h0);
    double h_delta = h1 - h0;
    const double d0 = A.dz ? b_d0[q] : g.H_to_Z * h0, d1 = A.dz ? b_d1[q] : g.H_to_Z * h1;
    double dz_harm = 2.0 * d0 * d1 / (d0 + d1 + dz_neglect);
    double dz_arith = 0.5 * (d1 + d0);
    if (OBC && side) {                                               // :1338-1341, :1345-1348
      const double hs = (side < 0) ? h0 : h1, ds = (side < 0) ? d0 : d1;
      h_harm = hs; h_arith = hs; h_delta = 0.;
      dz_harm = ds; dz_arith = ds;
    }
    const double vel = b_vel[q];
    double hvel, dz_vel, z_i;
    if (P.harmonic_visc) {                                           // :1363-1375
      hvel = h_harm; dz_vel = dz_harm;
      if (vel * h_delta < 0) {
        const double z2 = z_i_below, botfn = 1.0 / (1.0 + 0.09 * z2 * z2 * z2 * z2 * z2 * z2);
        hvel = (1.0 - botfn) * h_harm + botfn * h_arith;
        dz_vel = (1.0 - botfn) * dz_harm + botfn * dz_arith;
      }
      z_i = z_i_below + dz_harm * I_Hbbl;
    } else {                                                         // :1376-1408
      zcol0 = zcol0 + d0; zcol1 = zcol1 + d1;
      zh = zh + dz_harm;
      double z_clear = max2(zcol0, zcol1) + Dmin;
      if (OBC && side < 0) z_clear = zcol0 + Dmin;                   // :1381-1382
      if (OBC && side > 0) z_clear = zcol1 + Dmin;
      z_i = max2(zh, z_clear) * I_Hbbl;
      hvel = h_arith; dz_vel = dz_arith;
      if (vel * h_delta > 0) {
        if (zh * I_Hbbl < P.harm_BL_val) {
          hvel = h_harm; dz_vel = dz_harm;
        } else {
          double z2_wt = 1.0;
          if (zh * I_Hbbl < 2.0 * P.harm_BL_val) z2_wt = max2(0.0, min2(1.0, zh * I_Hbbl * I_valBL - 1.0));
          const double z2 = z2_wt * (max2(zh, z_clear) * I_Hbbl);
          const double botfn = 1.0 / (1.0 + 0.09 * z2 * z2 * z2 * z2 * z2 * z2);
          hvel = (1.0 - botfn) * h_arith + botfn * h_harm;
          dz_vel = (1.0 - botfn) * dz_arith + botfn * dz_harm;
        }
      }
    }
    A.hv[f2 + fpl * k] = hvel + h_neglect;                           // :1510
    if (A.f_store) A.f_store[f2 + fpl * k] = vel;                     // (the solve below reads it back: same lane)
    if (A.dzv) A.dzv[f2 + fpl * k] = dz_vel;

    // the interface below this layer, K = k+1 (find_coupling_coef :1948-2007 with hvel = dz_vel)
    const int K = k + 1;
    double a_cpl;
    if (K == nz) {
      const double Kv_tot = P.Kv;
      if (P.bottomdraglaw) {
        const double dhc = dz_vel * 0.5;
        if (dhc < bbl_thick) a_cpl = kv_bbl / ((dhc + hn) + I_amax * kv_bbl);
        else a_cpl = kv_bbl / ((bbl_thick + hn) + I_amax * kv_bbl);
      } else if (fabs(P.Kv_extra_bbl) > 0.0) {
        a_cpl = (Kv_tot + P.Kv_extra_bbl) / ((0.5 * dz_vel + hn) + I_amax * (Kv_tot + P.Kv_extra_bbl));
      } else {
        a_cpl = Kv_tot / ((0.5 * dz_vel + hn) + I_amax * Kv_tot);
      }
    } else {
      double Kv_tot = kvml ? b_kvml[q] : P.Kv;
      if (A.Kv_shear) {                                              // :1888-1928
        const double Kv_add = b_ksh[q];
        Kv_tot = Kv_tot + Kv_add;
      }
      if (P.bottomdraglaw) {
        const double z2 = z_i_below, botfn = 1.0 / (1.0 + 0.09 * z2 * z2 * z2 * z2 * z2 * z2);
        Kv_tot = Kv_tot + (kv_bbl - P.Kv) * botfn;
        const double dhc = 0.5 * (dzv_below + dz_vel);
        double h_shear;
        if (dhc > bbl_thick) h_shear = ((1.0 - botfn) * dhc + botfn * bbl_thick) + hn;
        else h_shear = dhc + hn;
        a_cpl = Kv_tot / (h_shear + (I_amax * Kv_tot));
      } else if (fabs(P.Kv_extra_bbl) > 0.0) {
        const double z2 = z_i_below, botfn = 1.0 / (1.0 + 0.09 * z2 * z2 * z2 * z2 * z2 * z2);
        Kv_tot = Kv_tot + P.Kv_extra_bbl * botfn;
        const double h_shear = 0.5 * (dzv_below + dz_vel + hn);
        a_cpl = Kv_tot / (h_shear + I_amax * Kv_tot);
      } else {
        const double h_shear = 0.5 * (dzv_below + dz_vel + hn);
        a_cpl = Kv_tot / (h_shear + I_amax * Kv_tot);
      }
    }
    A.a[f2 + fpl * K] = min2(a_cpl_max, a_cpl + 0.0);                 // :1504 (a_cpl_gl90 = 0)
    z_i_below = z_i; dzv_below = dz_vel;
    }
  }
  A.a[f2] = min2(a_cpl_max, 0.0 + 0.0);                               // a_cpl(:,1) = 0 (the boundary-layer scheme starts at K = 2)
  // ---- the surface boundary layer :2047-2252 (no LOTW floor; Boussinesq): top-down over the few layers in it ----
  if (A.dzv) {
    double u_star = 0.5 * (A.ustar[c0] + A.ustar[c1]);                // :2098-2112 (find_ustar: forces%ustar)
    if (OBC && side) u_star = A.ustar[(side < 0) ? c0 : c1];
    const double absf = DIR ? 0.5 * (fabs(g.CoriolisBu[g.q2(i - 1, j)]) + fabs(g.CoriolisBu[g.q2(i, j)]))
                            : 0.5 * (fabs(g.CoriolisBu[g.q2(i, j - 1)]) + fabs(g.CoriolisBu[g.q2(i, j)]));
    int nk_in_ml;
    double h_ml = hn;
    if (P.dynamic_viscous_ML) {                                      // :2120-2150
      const double nkv = A.nkml_visc[f2];
      nk_in_ml = (int)ceil(nkv);
      for (int k = 1; k <= nk_in_ml; k++) {
        const double dzk = A.dzv[f2 + fpl * (k - 1)];
        if ((double)k <= nkv) h_ml = h_ml + dzk;
        else if ((double)k < nkv + 1.0) h_ml = h_ml + ((nkv + 1.0) - (double)k) * dzk;
      }
    } else {                                                         // :2152-2166
      nk_in_ml = P.nkml;
      for (int k = 1; k <= P.nkml; k++) h_ml = h_ml + A.dzv[f2 + fpl * (k - 1)];
    }
    if (u_star <= 0.0) nk_in_ml = 0;                                  // :2184
    double z_t = 0.0;
    double dz_above = nk_in_ml >= 2 ? A.dzv[f2] : 0.0;
    for (int K = 2; K <= nk_in_ml; K++) {                             // :2231-2250
      const double dz_here = A.dzv[f2 + fpl * (K - 1)];
      z_t = z_t + dz_above;
      const double temp1 = (z_t * h_ml - z_t * z_t);
      const double visc_ml = u_star * P.vonKar * (g.Z_to_H * temp1 * u_star) / (absf * temp1 + (h_ml + hn) * u_star);
      const double a_ml = visc_ml / (0.25 * (dz_here + dz_above + hn) + 0.5 * I_amax * visc_ml);
      const long n = f2 + fpl * (K - 1);
      A.a[n] = min2(a_cpl_max, max2(A.a[n], a_ml) + 0.0);            // (a below a_cpl_max was stored unchanged)
      dz_above = dz_here;
    }
  }
}

template <int DIR, bool OBC>
__global__ __launch_bounds__(64) void vv_coef_kernel(CoefArgs A) {
  const m6::GridDev &g = A.g;
  const int i = (DIR ? g.isc : g.isc - 1) + blockIdx.x * 64 + threadIdx.x;
  const int j = (DIR ? g.jsc - 1 : g.jsc) + blockIdx.y;
  if (i > g.iec) return;
  const long f2 = DIR ? g.v2(i, j) : g.u2(i, j);
  if (!((DIR ? g.mask2dCv[f2] : g.mask2dCu[f2]) > 0.0)) return;      // do_i
  coef_column<DIR, OBC>(A, i, j);
}

struct SolveArgs {
  m6::GridDev g;
  VVPar p;
  const double *a, *hv, *Ray, *h, *tau;
  double *x, *xr, *c1, *tbot;      // x: velocity (or null), xr: visc_rem (or null)
  unsigned long long *ntrunc;      // non-null: vertvisc_limit_vel is applied as the final velocities are stored
  double dt;
};

// vertvisc_limit_vel :2346-2370 / :2431-2455 for one point (no truncation files); returns the value to store
template <int DIR>
__device__ __forceinline__ double limit_vel(const m6::GridDev &g, const VVPar &P, double xv, double dt, double dL, long c0, long c1,
                                            const double *h, long hoff, unsigned long long *ntrunc) {
  bool trunc = false;
  double out = xv;
  if (P.CFL_based_trunc) {
    if (fabs(xv) < P.vel_underflow) { out = 0.0; }
    else if ((xv * (dt * dL)) * g.IareaT[c1] < -P.CFL_trunc) {
      out = (-0.9 * P.CFL_trunc) * (g.areaT[c1] / (dt * dL));
      trunc = true;
    } else if ((xv * (dt * dL)) * g.IareaT[c0] > P.CFL_trunc) {
      out = (0.9 * P.CFL_trunc) * (g.areaT[c0] / (dt * dL));
      trunc = true;
    }
  } else {
    const double maxvel = P.maxvel, truncvel = 0.9 * maxvel;
    if (fabs(xv) < P.vel_underflow) { out = 0.0; }
    else if (fabs(xv) > maxvel) {
      out = copysign(truncvel, xv);
      trunc = true;
    }
  }
  if (trunc && (h[c0 + hoff] + h[c1 + hoff] > 6.0 * g.Angstrom_H)) atomicAdd(ntrunc, 1ull);
  return out;
}

// vertvisc :646-760 / :862-960 and vertvisc_remnant :1105-1155 for one face column.  The two solves share the matrix
// (b1, d1, c1 depend on a, h_vel, Ray and dt only), so when the caller wants both -- the RK2 step always calls them
// as a pair with the same dt -- they are done in one pass over a and h_vel.  The truncation of vertvisc_limit_vel acts on
// the finished velocities only: the back substitution carries the untruncated value in a register and stores the
// truncated one.
template <int DIR>
__device__ __forceinline__ void solve_column(const SolveArgs &A, int i, int j) {
  const m6::GridDev &g = A.g;
  const VVPar &P = A.p;
  const long f2 = DIR ? g.v2(i, j) : g.u2(i, j);
  const long fpl = DIR ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh, hpl = (long)g.nih * g.njh;
  const long c0 = g.h2(i, j), cc1 = DIR ? g.h2(i, j + 1) : g.h2(i + 1, j);
  const int nz = g.nk;
  const double dt = A.dt;
  const double mask = DIR ? g.mask2dCv[f2] : g.mask2dCu[f2];
  const bool do_i = mask > 0.0;
  double *x = A.x, *xr = A.xr;
  const bool lim = A.ntrunc != nullptr;
  const double dL = lim ? (DIR ? g.dx_Cv[f2] : g.dy_Cu[f2]) : 0.0;
  double x_bot = 0.0;      // the untruncated bottom velocity (for taux_bot)
  if (do_i) {
    double surface_stress = 0.0;
    if (x) {
      const double dt_Rho0 = dt / P.H_to_RZ;                          // :611
      if (P.direct_stress) {                                         // :671-685
        const double Hmix = P.Hmix_stress, I_Hmix = 1.0 / Hmix;
        double zDS = 0.0;
        const double stress = dt_Rho0 * A.tau[f2];
        for (int k = 0; k < nz; k++) {
          const double h_a = 0.5 * (A.h[c0 + hpl * k] + A.h[cc1 + hpl * k]) + g.H_subroundoff;
          double hfr = 1.0;
          if ((zDS + h_a) > Hmix) hfr = (Hmix - zDS) / h_a;
          x[f2 + fpl * k] = x[f2 + fpl * k] + I_Hmix * hfr * stress;
          zDS = zDS + h_a;
          if (zDS >= Hmix) break;
        }
      } else {
        surface_stress = dt_Rho0 * (mask * A.tau[f2]);               // :687
      }
    }
    double b_denom_1 = A.hv[f2] + dt * ((A.Ray ? A.Ray[f2] : 0.0) + A.a[f2]);
    double b1 = 1.0 / (b_denom_1 + dt * A.a[f2 + fpl]);
    double d1 = b_denom_1 * b1;
    double xk = 0.0, rk = 0.0;
    if (x) { xk = b1 * (A.hv[f2] * x[f2] + surface_stress); if (nz > 1 || !lim) x[f2] = xk; }
    if (xr) { rk = b1 * A.hv[f2]; xr[f2] = rk; }
    double a_next = A.a[f2 + fpl];
    for (int kb = 1; kb < nz; kb += VB) {      // VB layers loaded at once, as in the coefficient sweep
      double b_a[VB], b_hv[VB], b_x[VB], b_ray[VB];
#pragma unroll
      for (int q = 0; q < VB; q++) {
        const int k = kb + q;
        b_a[q] = b_hv[q] = b_x[q] = b_ray[q] = 0.0;
        if (k < nz) {
          const long n = f2 + fpl * k;
          b_a[q] = A.a[n + fpl]; b_hv[q] = A.hv[n];
          if (x) b_x[q] = x[n];
          if (A.Ray) b_ray[q] = A.Ray[n];
        }
      }
#pragma unroll
      for (int q = 0; q < VB; q++) {
        const int k = kb + q;
        if (k >= nz) break;
        const long n = f2 + fpl * k;
        const double ak = a_next, hvk = b_hv[q];
        a_next = b_a[q];
        A.c1[n] = dt * ak * b1;
        b_denom_1 = hvk + dt * ((A.Ray ? b_ray[q] : 0.0) + ak * d1);
        b1 = 1.0 / (b_denom_1 + dt * a_next);
        d1 = b_denom_1 * b1;
        if (x) { xk = (hvk * b_x[q] + dt * ak * xk) * b1; if (k < nz - 1 || !lim) x[n] = xk; }
        if (xr) { rk = (hvk + dt * ak * rk) * b1; xr[n] = rk; }
      }
    }
    x_bot = xk;
    if (x && lim) x[f2 + fpl * (nz - 1)] = limit_vel<DIR>(g, P, xk, dt, dL, c0, cc1, A.h, hpl * (nz - 1), A.ntrunc);
    for (int kb = nz - 2; kb >= 0; kb -= VB) {
      double b_c[VB], b_x[VB], b_r[VB];
#pragma unroll
      for (int q = 0; q < VB; q++) {
        const int k = kb - q;
        b_c[q] = b_x[q] = b_r[q] = 0.0;
        if (k >= 0) {
          const long n = f2 + fpl * k;
          b_c[q] = A.c1[n + fpl];
          if (x) b_x[q] = x[n];
          if (xr) b_r[q] = xr[n];
        }
      }
#pragma unroll
      for (int q = 0; q < VB; q++) {
        const int k = kb - q;
        if (k < 0) break;
        const long n = f2 + fpl * k;
        const double c = b_c[q];
        if (x) {
          xk = b_x[q] + c * xk;
          x[n] = lim ? limit_vel<DIR>(g, P, xk, dt, dL, c0, cc1, A.h, hpl * k, A.ntrunc) : xk;
        }
        if (xr) { rk = b_r[q] + c * rk; xr[n] = rk; }
      }
    }
  } else if (x) {
    x_bot = x[f2 + fpl * (nz - 1)];
    if (lim)      // vertvisc_limit_vel runs over every face of the compute rows, masked or not
      for (int k = 0; k < nz; k++) x[f2 + fpl * k] = limit_vel<DIR>(g, P, x[f2 + fpl * k], dt, dL, c0, cc1, A.h, hpl * k, A.ntrunc);
  }
  if (A.tbot) {                                                      // :798-805 (before the truncation)
    double tb = P.H_to_RZ * (x_bot * A.a[f2 + fpl * nz]);
    if (A.Ray) for (int k = 0; k < nz; k++) tb = tb + P.H_to_RZ * (A.Ray[f2 + fpl * k] * x[f2 + fpl * k]);
    A.tbot[f2] = tb;
  }
}

template <int DIR>
__global__ __launch_bounds__(64) void vv_solve_kernel(SolveArgs A) {
  const m6::GridDev &g = A.g;
  const int i = (DIR ? g.isc : g.isc - 1) + blockIdx.x * 64 + threadIdx.x;
  const int j = (DIR ? g.jsc - 1 : g.jsc) + blockIdx.y;
  if (i > g.iec) return;
  solve_column<DIR>(A, i, j);
}

// vertvisc_coef followed by the solve(s) of the same column: the coupling coefficients and thicknesses the bottom-up sweep
// has just stored are read back by the same lane while they are still in L2, instead of by a second kernel from HBM.
#ifndef VV_OCC
#define VV_OCC 4      // waves per SIMD the register allocation aims at (121 VGPRs: 4; tools/build_variant.sh for experiments)
#endif
template <int DIR>
__global__ __launch_bounds__(64, VV_OCC) void vv_coef_solve_kernel(CoefArgs C, SolveArgs A) {
  const m6::GridDev &g = A.g;
  const int i = (DIR ? g.isc : g.isc - 1) + blockIdx.x * 64 + threadIdx.x;
  const int j = (DIR ? g.jsc - 1 : g.jsc) + blockIdx.y;
  if (i > g.iec) return;
  const long f2 = DIR ? g.v2(i, j) : g.u2(i, j);
  const double mask = DIR ? g.mask2dCv[f2] : g.mask2dCu[f2];
  if (mask > 0.0) coef_column<DIR>(C, i, j);
  else if (C.f_u0 && C.f_store) {      // a masked column: the increment alone (0 * (...)), as the step's sweep would have left it
    const long fpl = DIR ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh;
    for (int k = 0; k < g.nk; k++) C.f_store[f2 + fpl * k] = coef_vel(C, f2 + fpl * k, mask);
  }
  solve_column<DIR>(A, i, j);
}

// the increment as a sweep of its own, for the forms of the step that go through the separate entries
template <int DIR>
__global__ __launch_bounds__(256) void vv_increment_kernel(m6::GridDev g, const double *u0, const double *a1, const double *a2, double dtv,
                                                           double *out) {
  const int i = (DIR ? g.isc : g.isc - 1) + blockIdx.x * 256 + threadIdx.x;
  const int j = (DIR ? g.jsc - 1 : g.jsc) + blockIdx.y;
  if (i > g.iec) return;
  const long f2 = DIR ? g.v2(i, j) : g.u2(i, j);
  const long n = f2 + (DIR ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh) * blockIdx.z;
  const double acc = a2 ? (a1[n] + a2[n]) : a1[n];
  out[n] = (DIR ? g.mask2dCv[f2] : g.mask2dCu[f2]) * (u0[n] + dtv * acc);
}

struct LimitArgs {
  m6::GridDev g;
  VVPar p;
  const double *h;
  double *x;
  unsigned long long *ntrunc;
  double dt;
};

// vertvisc_limit_vel :2346-2370 / :2431-2455 (no truncation files)
template <int DIR>
__global__ __launch_bounds__(256) void vv_limit_kernel(LimitArgs A) {
  const m6::GridDev &g = A.g;
  const VVPar &P = A.p;
  const int i = (DIR ? g.isc : g.isc - 1) + blockIdx.x * 256 + threadIdx.x;
  const int j = (DIR ? g.jsc - 1 : g.jsc) + blockIdx.y;
  const int k = blockIdx.z;
  if (i > g.iec) return;
  const long f2 = DIR ? g.v2(i, j) : g.u2(i, j);
  const long fpl = DIR ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh, hpl = (long)g.nih * g.njh;
  const long n = f2 + fpl * k;
  const long c0 = g.h2(i, j), c1 = DIR ? g.h2(i, j + 1) : g.h2(i + 1, j);
  const double dL = DIR ? g.dx_Cv[f2] : g.dy_Cu[f2];
  const double H_report = 6.0 * g.Angstrom_H;
  const double xv = A.x[n];
  const double dt = A.dt;
  bool trunc = false;
  if (P.CFL_based_trunc) {
    if (fabs(xv) < P.vel_underflow) { A.x[n] = 0.0; }
    else if ((xv * (dt * dL)) * g.IareaT[c1] < -P.CFL_trunc) {
      A.x[n] = (-0.9 * P.CFL_trunc) * (g.areaT[c1] / (dt * dL));
      trunc = true;
    } else if ((xv * (dt * dL)) * g.IareaT[c0] > P.CFL_trunc) {
      A.x[n] = (0.9 * P.CFL_trunc) * (g.areaT[c0] / (dt * dL));
      trunc = true;
    }
  } else {
    const double maxvel = P.maxvel, truncvel = 0.9 * maxvel;
    if (fabs(xv) < P.vel_underflow) { A.x[n] = 0.0; }
    else if (fabs(xv) > maxvel) {
      A.x[n] = copysign(truncvel, xv);
      trunc = true;
    }
  }
  if (trunc && (A.h[c0 + hpl * k] + A.h[c1 + hpl * k] > H_report)) atomicAdd(A.ntrunc, 1ull);
}

int check_cs(const mom6hip_vertvisc_cs_t *cs, const char *who) {
  static const char *names[7] = {"(free)", "(free)", "FIXED_DEPTH_LOTW_ML", "LOTW_VISCOUS_ML_FLOOR",
                                 "USE_GL90_IN_SSW", "STOKES_MIXING_COMBINED", "non-Boussinesq mode"};
  for (int n = 0; n < 7; n++) M6_REQUIRE(!cs->unsupported[n], "%s: %s is not provided by libmom6hip", who, names[n]);
  M6_REQUIRE(cs->answer_date >= 20190101, "%s: VERT_FRICTION_ANSWER_DATE < 20190101 is not provided", who);
  M6_REQUIRE(cs->a_u && cs->a_v && cs->h_u && cs->h_v, "%s: the a_u, a_v, h_u, h_v arrays of the control structure are required", who);
  return 0;
}

VVPar par_of(const mom6hip_vertvisc_cs_t *cs) {
  VVPar p;
  p.Hmix = cs->Hmix; p.Hmix_stress = cs->Hmix_stress; p.Kvml_invZ2 = cs->Kvml_invZ2; p.Kv = cs->Kv; p.Hbbl = cs->Hbbl;
  p.Kv_extra_bbl = cs->Kv_extra_bbl; p.harm_BL_val = cs->harm_BL_val; p.maxvel = cs->maxvel; p.CFL_trunc = cs->CFL_trunc;
  p.vel_underflow = cs->vel_underflow; p.H_to_RZ = cs->H_to_RZ; p.vonKar = cs->vonKar;
  p.dynamic_viscous_ML = cs->dynamic_viscous_ML; p.nkml = cs->nkml;
  p.bottomdraglaw = cs->bottomdraglaw; p.harmonic_visc = cs->harmonic_visc; p.direct_stress = cs->direct_stress;
  p.CFL_based_trunc = cs->CFL_based_trunc;
  return p;
}

struct Sz { size_t h2, u2, v2, h3, u3, v3, ui, vi, hi; };
Sz sizes(const m6::GridDev &g) {
  Sz s;
  s.h2 = sizeof(double) * (size_t)g.nih * g.njh; s.u2 = sizeof(double) * (size_t)(g.nih + 1) * g.njh;
  s.v2 = sizeof(double) * (size_t)g.nih * (g.njh + 1);
  s.h3 = s.h2 * g.nk; s.u3 = s.u2 * g.nk; s.v3 = s.v2 * g.nk;
  s.ui = s.u2 * (g.nk + 1); s.vi = s.v2 * (g.nk + 1); s.hi = s.h2 * (g.nk + 1);
  return s;
}

}  // namespace

extern "C" int mom6hip_vertvisc_coef(mom6hip_ctx_t *ctx, mom6hip_vertvisc_cs_t *cs, const double *u, const double *v,
                                     const double *h, const double *dz, const mom6hip_vertvisc_type_t *visc, double dt,
                                     int32_t memspace) {
  return mom6hip_vertvisc_coef_obc(ctx, cs, u, v, h, dz, visc, dt, nullptr, memspace);
}

// vertvisc_coef with OBC associated: at the faces of the open-boundary segments the thicknesses, the depth, Kv_shear and ustar are
// those of the cell inside (:1335-1355, :1546-1566, :1901-1925, :2061-2110)
extern "C" int mom6hip_vertvisc_coef_obc(mom6hip_ctx_t *ctx, mom6hip_vertvisc_cs_t *cs, const double *u, const double *v,
                                         const double *h, const double *dz, const mom6hip_vertvisc_type_t *visc, double dt,
                                         const mom6hip_obc_t *obc, int32_t memspace) {
  (void)dt;
  M6_REQUIRE(ctx != nullptr, "MOM_vert_friction(coef): Module must be initialized before it is used.");
  M6_REQUIRE(cs && u && v && h && visc, "vertvisc_coef: null argument");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "vertvisc_coef: bad memspace");
  if (check_cs(cs, "vertvisc_coef")) return 1;
  M6_REQUIRE(visc->Kv_shear_Bu == nullptr, "vertvisc_coef: visc%%Kv_shear_Bu is not provided by libmom6hip");
  M6_REQUIRE(!cs->bottomdraglaw || (visc->Kv_bbl_u && visc->Kv_bbl_v && visc->bbl_thick_u && visc->bbl_thick_v),
             "vertvisc_coef: BOTTOMDRAGLAW needs visc%%Kv_bbl_u/v and visc%%bbl_thick_u/v");
  M6_REQUIRE(!(cs->Kvml_invZ2 > 0.0) || cs->Hmix > 0.0, "vertvisc_coef: KV_ML_INVZ2 needs HMIX_FIXED");
  const m6::GridDev g = ctx->g;
  M6_REQUIRE(g.mask2dCu && g.mask2dCv && g.bathyT, "vertvisc_coef: mask2dCu, mask2dCv and bathyT are required");
  const bool surface_bl = cs->dynamic_viscous_ML || cs->nkml > 0;      // find_coupling_coef :2047
  if (surface_bl) {
    M6_REQUIRE(visc->ustar, "vertvisc_coef: DYNAMIC_VISCOUS_ML / a bulk mixed layer needs forces%%ustar (visc->ustar)");
    M6_REQUIRE(!cs->dynamic_viscous_ML || (visc->nkml_visc_u && visc->nkml_visc_v),
               "vertvisc_coef: DYNAMIC_VISCOUS_ML needs visc%%nkml_visc_u / nkml_visc_v (set_viscous_ML)");
    M6_REQUIRE(cs->nkml >= 0 && cs->nkml <= g.nk && g.CoriolisBu, "vertvisc_coef: GV%%nkml out of range, or CoriolisBu missing");
  }
  const Sz sz = sizes(g);
  m6::Stager st(ctx, memspace);
  CoefArgs A[2];
  for (int d = 0; d < 2; d++) {
    A[d].ustar = A[d].nkml_visc = nullptr; A[d].dzv = nullptr;
    A[d].f_u0 = A[d].f_a1 = A[d].f_a2 = nullptr; A[d].f_dt = 0.0; A[d].f_store = nullptr;
  }
  if (surface_bl) {
    const double *dus = st.in(visc->ustar, sz.h2);
    A[0].ustar = A[1].ustar = dus;
    if (cs->dynamic_viscous_ML) { A[0].nkml_visc = st.in(visc->nkml_visc_u, sz.u2); A[1].nkml_visc = st.in(visc->nkml_visc_v, sz.v2); }
    A[0].dzv = (double *)st.scratch(sz.u3); A[1].dzv = (double *)st.scratch(sz.v3);
  }
  A[0].vel = st.in(u, sz.u3); A[1].vel = st.in(v, sz.v3);
  const double *dh = st.in(h, sz.h3), *ddz = st.in(dz, sz.h3), *dks = st.in(visc->Kv_shear, sz.hi);
  A[0].kv_bbl = st.in(visc->Kv_bbl_u, sz.u2); A[1].kv_bbl = st.in(visc->Kv_bbl_v, sz.v2);
  A[0].bbl_thick = st.in(visc->bbl_thick_u, sz.u2); A[1].bbl_thick = st.in(visc->bbl_thick_v, sz.v2);
  A[0].a = st.inout(cs->a_u, sz.ui); A[1].a = st.inout(cs->a_v, sz.vi);
  A[0].hv = st.inout(cs->h_u, sz.u3); A[1].hv = st.inout(cs->h_v, sz.v3);
  M6_REQUIRE(!st.failed(), "vertvisc_coef: staging failed");
  A[0].side = A[1].side = nullptr;
  if (m6::obc_side_maps(ctx, st, obc, &A[0].side, &A[1].side, "vertvisc_coef")) return 1;
  for (int d = 0; d < 2; d++) {
    A[d].g = g; A[d].p = par_of(cs); A[d].h = dh; A[d].dz = ddz; A[d].Kv_shear = dks;
    const dim3 grid((g.iec - g.isc + 1 + (d ? 0 : 1) + 63) / 64, g.jec - g.jsc + 1 + (d ? 1 : 0));
    if (A[d].side) {
      if (d == 0) hipLaunchKernelGGL((vv_coef_kernel<0, true>), grid, dim3(64), 0, ctx->stream, A[d]);
      else hipLaunchKernelGGL((vv_coef_kernel<1, true>), grid, dim3(64), 0, ctx->stream, A[d]);
    } else {
      if (d == 0) hipLaunchKernelGGL((vv_coef_kernel<0, false>), grid, dim3(64), 0, ctx->stream, A[d]);
      else hipLaunchKernelGGL((vv_coef_kernel<1, false>), grid, dim3(64), 0, ctx->stream, A[d]);
    }
  }
  M6_HIP(hipGetLastError());
  return st.finish();
}

namespace {
// the solve of both directions: velocities x (with vertvisc_limit_vel fused when the Rayleigh drag is off), visc_rem xr,
// or both at once
int run_solve(mom6hip_ctx_t *ctx, m6::Stager &st, const mom6hip_vertvisc_cs_t *cs, const double *const a[2], const double *const hv[2],
              const double *const Ray[2], const double *dh, const double *const tau[2], double *const x[2], double *const xr[2],
              double *const tbot[2], double dt, bool want_limit) {
  const m6::GridDev g = ctx->g;
  const Sz sz = sizes(g);
  double *c1 = (double *)st.scratch(sz.u3 > sz.v3 ? sz.u3 : sz.v3);
  M6_REQUIRE(!st.failed() && c1, "vertvisc: out of device memory");
  unsigned long long *cnt = nullptr;
  if (want_limit) {
    if (ctx->vv_ntrunc.reserve(sizeof(unsigned long long)) || !ctx->vv_ntrunc.p) return 1;
    if (!ctx->vv_ntrunc_ready) {
      M6_HIP(hipMemsetAsync(ctx->vv_ntrunc.p, 0, sizeof(unsigned long long), ctx->stream));
      ctx->vv_ntrunc_ready = true;
    }
    cnt = (unsigned long long *)ctx->vv_ntrunc.p;
  }
  // with Rayleigh drag taux_bot sums Ray*u over the untruncated velocities in k order: the truncation stays a separate pass
  const bool fuse_limit = want_limit && !(Ray[0] || Ray[1]);
  for (int d = 0; d < 2; d++) {
    SolveArgs A;
    A.g = g; A.p = par_of(cs); A.a = a[d]; A.hv = hv[d]; A.Ray = Ray[d]; A.h = dh; A.tau = tau[d]; A.x = x[d]; A.xr = xr[d]; A.c1 = c1;
    A.tbot = tbot[d]; A.dt = dt; A.ntrunc = fuse_limit ? cnt : nullptr;
    const dim3 grid((g.iec - g.isc + 1 + (d ? 0 : 1) + 63) / 64, g.jec - g.jsc + 1 + (d ? 1 : 0));
    if (d == 0) hipLaunchKernelGGL(vv_solve_kernel<0>, grid, dim3(64), 0, ctx->stream, A);
    else hipLaunchKernelGGL(vv_solve_kernel<1>, grid, dim3(64), 0, ctx->stream, A);
  }
  if (want_limit && !fuse_limit) {
    for (int d = 0; d < 2; d++) {
      LimitArgs L;
      L.g = g; L.p = par_of(cs); L.h = dh; L.x = x[d]; L.ntrunc = cnt; L.dt = dt;
      const dim3 grid((g.iec - g.isc + 1 + (d ? 0 : 1) + 255) / 256, g.jec - g.jsc + 1 + (d ? 1 : 0), g.nk);
      if (d == 0) hipLaunchKernelGGL(vv_limit_kernel<0>, grid, dim3(256), 0, ctx->stream, L);
      else hipLaunchKernelGGL(vv_limit_kernel<1>, grid, dim3(256), 0, ctx->stream, L);
    }
  }
  M6_HIP(hipGetLastError());
  return 0;
}

int vertvisc_impl(mom6hip_ctx_t *ctx, mom6hip_vertvisc_cs_t *cs, double *u, double *v, const double *h, const double *taux,
                  const double *tauy, const mom6hip_vertvisc_type_t *visc, double dt, double *taux_bot, double *tauy_bot,
                  double *visc_rem_u, double *visc_rem_v, const mom6hip_obc_t *obc, int32_t memspace) {
  M6_REQUIRE(ctx != nullptr, "MOM_vert_friction(visc): Module must be initialized before it is used.");
  M6_REQUIRE(cs && u && v && h && taux && tauy && visc, "vertvisc: null argument");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "vertvisc: bad memspace");
  if (check_cs(cs, "vertvisc")) return 1;
  M6_REQUIRE(dt > 0.0 && cs->H_to_RZ > 0.0, "vertvisc: dt and GV%%H_to_RZ must be positive");
  M6_REQUIRE(!cs->direct_stress || cs->Hmix_stress > 0.0, "vertvisc_init: HMIX_STRESS must be set to a positive value if DIRECT_STRESS is true.");
  M6_REQUIRE((visc_rem_u != nullptr) == (visc_rem_v != nullptr), "vertvisc: visc_rem_u and visc_rem_v come as a pair");
  const m6::GridDev g = ctx->g;
  M6_REQUIRE(g.mask2dCu && g.mask2dCv && g.areaT && g.IareaT && g.dy_Cu && g.dx_Cv, "vertvisc: a required grid metric is missing");
  const Sz sz = sizes(g);
  m6::Stager st(ctx, memspace);
  double *x[2] = {st.inout(u, sz.u3), st.inout(v, sz.v3)};
  double *xr[2] = {st.inout(visc_rem_u, sz.u3), st.inout(visc_rem_v, sz.v3)};
  const double *dh = st.in(h, sz.h3);
  const double *tau[2] = {st.in(taux, sz.u2), st.in(tauy, sz.v2)};
  const double *a[2] = {st.in((const double *)cs->a_u, sz.ui), st.in((const double *)cs->a_v, sz.vi)};
  const double *hv[2] = {st.in((const double *)cs->h_u, sz.u3), st.in((const double *)cs->h_v, sz.v3)};
  const double *Ray[2] = {st.in(visc->Ray_u, sz.u3), st.in(visc->Ray_v, sz.v3)};
  double *tbot[2] = {st.inout(taux_bot, sz.u2), st.inout(tauy_bot, sz.v2)};      // (only the compute rows are written)
  M6_REQUIRE(!st.failed(), "vertvisc: staging failed");
  if (run_solve(ctx, st, cs, a, hv, Ray, dh, tau, x, xr, tbot, dt, true)) return 1;      // incl. vertvisc_limit_vel :986
  if (m6::obc_store_specified(ctx, st, obc, x[0], x[1], "vertvisc")) return 1;           // :988-1006
  const int rc = st.finish();
  if (rc == 0 && memspace == MOM6HIP_MEM_HOST) return mom6hip_vertvisc_ntrunc(ctx, cs);
  return rc;
}
}  // namespace

extern "C" int mom6hip_vertvisc(mom6hip_ctx_t *ctx, mom6hip_vertvisc_cs_t *cs, double *u, double *v, const double *h,
                                const double *taux, const double *tauy, const mom6hip_vertvisc_type_t *visc, double dt,
                                double *taux_bot, double *tauy_bot, int32_t memspace) {
  return vertvisc_impl(ctx, cs, u, v, h, taux, tauy, visc, dt, taux_bot, tauy_bot, nullptr, nullptr, nullptr, memspace);
}

// vertvisc with OBC associated: the velocities of the specified segments are stored over the result (:988-1006)
extern "C" int mom6hip_vertvisc_obc(mom6hip_ctx_t *ctx, mom6hip_vertvisc_cs_t *cs, double *u, double *v, const double *h,
                                    const double *taux, const double *tauy, const mom6hip_vertvisc_type_t *visc, double dt,
                                    double *taux_bot, double *tauy_bot, const mom6hip_obc_t *obc, int32_t memspace) {
  return vertvisc_impl(ctx, cs, u, v, h, taux, tauy, visc, dt, taux_bot, tauy_bot, nullptr, nullptr, obc, memspace);
}

// vertvisc followed by vertvisc_remnant with the same dt -- the pair step_MOM_dyn_split_RK2 calls at :731-744 and :985-994
// -- in one pass over the coupling coefficients.  Same results as the two calls.
extern "C" int mom6hip_vertvisc_and_remnant(mom6hip_ctx_t *ctx, mom6hip_vertvisc_cs_t *cs, double *u, double *v, const double *h,
                                            const double *taux, const double *tauy, const mom6hip_vertvisc_type_t *visc, double dt,
                                            double *taux_bot, double *tauy_bot, double *visc_rem_u, double *visc_rem_v,
                                            int32_t memspace) {
  M6_REQUIRE(visc_rem_u && visc_rem_v, "vertvisc_and_remnant: null argument");
  return vertvisc_impl(ctx, cs, u, v, h, taux, tauy, visc, dt, taux_bot, tauy_bot, visc_rem_u, visc_rem_v, nullptr, memspace);
}

// vertvisc_coef, then (update_velocities) vertvisc, then vertvisc_remnant, all with the same dt: the sequences of
// step_MOM_dyn_split_RK2 at :598-600 (velocities untouched), :717-744 and :974-994.  One kernel per direction: each lane
// runs the bottom-up coefficient sweep and then the solve of its own column.  Same results as the separate calls.
extern "C" int mom6hip_vertvisc_step(mom6hip_ctx_t *ctx, mom6hip_vertvisc_cs_t *cs, double *u, double *v, const double *h,
                                     const double *dz, const double *taux, const double *tauy, const mom6hip_vertvisc_type_t *visc,
                                     double dt, int32_t update_velocities, double *taux_bot, double *tauy_bot, double *visc_rem_u,
                                     double *visc_rem_v, int32_t memspace) {
  return m6::vertvisc_step_inc(ctx, cs, u, v, h, dz, taux, tauy, visc, dt, update_velocities, taux_bot, tauy_bot, visc_rem_u, visc_rem_v,
                               nullptr, memspace);
}

// mom6hip_vertvisc_step with the velocities given as the increment of the RK2 step (inc, device pointers; null: u and v as they
// are).  With update_velocities the incremented velocities are what u and v hold on entry of the reference's vertvisc_coef, so
// they are stored to u and v before the solve reads them; without it they are only used.  u, v may be the increment's own u0, v0.
int m6::vertvisc_step_inc(mom6hip_ctx_t *ctx, mom6hip_vertvisc_cs_t *cs, double *u, double *v, const double *h, const double *dz,
                          const double *taux, const double *tauy, const mom6hip_vertvisc_type_t *visc, double dt, int32_t update_velocities,
                          double *taux_bot, double *tauy_bot, double *visc_rem_u, double *visc_rem_v, const m6::VelIncrement *inc,
                          int32_t memspace) {
  M6_REQUIRE(ctx != nullptr, "MOM_vert_friction(visc): Module must be initialized before it is used.");
  M6_REQUIRE(cs && u && v && h && visc && visc_rem_u && visc_rem_v, "vertvisc_step: null argument");
  M6_REQUIRE(!update_velocities || (taux && tauy), "vertvisc_step: the wind stress is needed to update the velocities");
  M6_REQUIRE(!inc || memspace == MOM6HIP_MEM_DEVICE, "vertvisc_step: the increment form works on device arrays");
  // the Rayleigh-drag bottom stress needs the separate truncation pass, the surface boundary layer of DYNAMIC_VISCOUS_ML / a
  // bulk mixed layer its own scratch: the plain sequence
  if (visc->Ray_u || visc->Ray_v || cs->dynamic_viscous_ML || cs->nkml > 0) {
    if (inc) {      // the increment as its own sweep (into u and v: they are the step's work arrays when the velocities stay untouched)
      const m6::GridDev gg = ctx->g;
      const dim3 gu((gg.iec - gg.isc + 2 + 255) / 256, gg.jec - gg.jsc + 1, gg.nk), gv((gg.iec - gg.isc + 1 + 255) / 256, gg.jec - gg.jsc + 2, gg.nk);
      hipLaunchKernelGGL(vv_increment_kernel<0>, gu, dim3(256), 0, ctx->stream, gg, inc->u0, inc->a1u, inc->a2u, inc->dtv, u);
      hipLaunchKernelGGL(vv_increment_kernel<1>, gv, dim3(256), 0, ctx->stream, gg, inc->v0, inc->a1v, inc->a2v, inc->dtv, v);
      M6_HIP(hipGetLastError());
    }
    if (int rc = mom6hip_vertvisc_coef(ctx, cs, u, v, h, dz, visc, dt, memspace)) return rc;
    if (update_velocities)
      if (int rc = mom6hip_vertvisc(ctx, cs, u, v, h, taux, tauy, visc, dt, taux_bot, tauy_bot, memspace)) return rc;
    return mom6hip_vertvisc_remnant(ctx, cs, visc, visc_rem_u, visc_rem_v, dt, memspace);
  }
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "vertvisc_step: bad memspace");
  if (check_cs(cs, "vertvisc_step")) return 1;
  M6_REQUIRE(dt > 0.0 && cs->H_to_RZ > 0.0, "vertvisc: dt and GV%%H_to_RZ must be positive");
  M6_REQUIRE(visc->Kv_shear_Bu == nullptr, "vertvisc_coef: visc%%Kv_shear_Bu is not provided by libmom6hip");
  M6_REQUIRE(!cs->bottomdraglaw || (visc->Kv_bbl_u && visc->Kv_bbl_v && visc->bbl_thick_u && visc->bbl_thick_v),
             "vertvisc_coef: BOTTOMDRAGLAW needs visc%%Kv_bbl_u/v and visc%%bbl_thick_u/v");
  M6_REQUIRE(!(cs->Kvml_invZ2 > 0.0) || cs->Hmix > 0.0, "vertvisc_coef: KV_ML_INVZ2 needs HMIX_FIXED");
  M6_REQUIRE(!cs->direct_stress || cs->Hmix_stress > 0.0, "vertvisc_init: HMIX_STRESS must be set to a positive value if DIRECT_STRESS is true.");
  const m6::GridDev g = ctx->g;
  M6_REQUIRE(g.mask2dCu && g.mask2dCv && g.bathyT && g.areaT && g.IareaT && g.dy_Cu && g.dx_Cv, "vertvisc: a required grid metric is missing");
  const Sz sz = sizes(g);
  m6::Stager st(ctx, memspace);
  double *x[2] = {update_velocities ? st.inout(u, sz.u3) : (double *)st.in((const double *)u, sz.u3),
                  update_velocities ? st.inout(v, sz.v3) : (double *)st.in((const double *)v, sz.v3)};
  double *xr[2] = {st.inout(visc_rem_u, sz.u3), st.inout(visc_rem_v, sz.v3)};
  const double *dh = st.in(h, sz.h3), *ddz = st.in(dz, sz.h3), *dks = st.in(visc->Kv_shear, sz.hi);
  const double *tau[2] = {st.in(taux, sz.u2), st.in(tauy, sz.v2)};
  const double *kvb[2] = {st.in(visc->Kv_bbl_u, sz.u2), st.in(visc->Kv_bbl_v, sz.v2)};
  const double *bth[2] = {st.in(visc->bbl_thick_u, sz.u2), st.in(visc->bbl_thick_v, sz.v2)};
  double *a[2] = {st.inout(cs->a_u, sz.ui), st.inout(cs->a_v, sz.vi)};
  double *hv[2] = {st.inout(cs->h_u, sz.u3), st.inout(cs->h_v, sz.v3)};
  double *tbot[2] = {update_velocities ? st.inout(taux_bot, sz.u2) : nullptr, update_velocities ? st.inout(tauy_bot, sz.v2) : nullptr};
  double *c1 = (double *)st.scratch(sz.u3 > sz.v3 ? sz.u3 : sz.v3);
  M6_REQUIRE(!st.failed() && c1, "vertvisc_step: staging failed");
  unsigned long long *cnt = nullptr;
  if (update_velocities) {
    if (ctx->vv_ntrunc.reserve(sizeof(unsigned long long)) || !ctx->vv_ntrunc.p) return 1;
    if (!ctx->vv_ntrunc_ready) {
      M6_HIP(hipMemsetAsync(ctx->vv_ntrunc.p, 0, sizeof(unsigned long long), ctx->stream));
      ctx->vv_ntrunc_ready = true;
    }
    cnt = (unsigned long long *)ctx->vv_ntrunc.p;
  }
  for (int d = 0; d < 2; d++) {
    CoefArgs C;
    C.g = g; C.p = par_of(cs); C.vel = x[d]; C.h = dh; C.dz = ddz; C.kv_bbl = kvb[d]; C.bbl_thick = bth[d]; C.Kv_shear = dks;
    C.a = a[d]; C.hv = hv[d]; C.ustar = C.nkml_visc = nullptr; C.dzv = nullptr;
    C.f_u0 = C.f_a1 = C.f_a2 = nullptr; C.f_dt = 0.0; C.f_store = nullptr; C.side = nullptr;
    if (inc) {
      C.f_u0 = d ? inc->v0 : inc->u0; C.f_a1 = d ? inc->a1v : inc->a1u; C.f_a2 = d ? inc->a2v : inc->a2u; C.f_dt = inc->dtv;
      C.f_store = update_velocities ? x[d] : nullptr;
    }
    SolveArgs A;
    A.g = g; A.p = C.p; A.a = a[d]; A.hv = hv[d]; A.Ray = nullptr; A.h = dh; A.tau = tau[d]; A.x = update_velocities ? x[d] : nullptr;
    A.xr = xr[d]; A.c1 = c1; A.tbot = tbot[d]; A.dt = dt; A.ntrunc = cnt;
    const dim3 grid((g.iec - g.isc + 1 + (d ? 0 : 1) + 63) / 64, g.jec - g.jsc + 1 + (d ? 1 : 0));
    // MOM6HIP_VV_LDS_BYTES: unused dynamic LDS per block, an occupancy throttle for experiments (a column's intermediates stay in
    // the memory-side cache only while few enough waves are in flight); 0 = none
    static const int lds_throttle = [] { const char *e = getenv("MOM6HIP_VV_LDS_BYTES"); return e ? atoi(e) : 0; }();
    if (d == 0) hipLaunchKernelGGL(vv_coef_solve_kernel<0>, grid, dim3(64), lds_throttle, ctx->stream, C, A);
    else hipLaunchKernelGGL(vv_coef_solve_kernel<1>, grid, dim3(64), lds_throttle, ctx->stream, C, A);
  }
  M6_HIP(hipGetLastError());
  const int rc = st.finish();
  if (rc == 0 && update_velocities && memspace == MOM6HIP_MEM_HOST) return mom6hip_vertvisc_ntrunc(ctx, cs);
  return rc;
}

// Adds the truncations counted on the device since the last call to cs->ntrunc (synchronises the stream).
extern "C" int mom6hip_vertvisc_ntrunc(mom6hip_ctx_t *ctx, mom6hip_vertvisc_cs_t *cs) {
  M6_REQUIRE(ctx && cs, "vertvisc_ntrunc: null argument");
  if (!ctx->vv_ntrunc_ready) return 0;
  unsigned long long n = 0;
  M6_HIP(hipMemcpyAsync(&n, ctx->vv_ntrunc.p, sizeof(n), hipMemcpyDeviceToHost, ctx->stream));
  M6_HIP(hipMemsetAsync(ctx->vv_ntrunc.p, 0, sizeof(n), ctx->stream));
  M6_HIP(hipStreamSynchronize(ctx->stream));
  cs->ntrunc += (int64_t)n;
  return 0;
}

extern "C" int mom6hip_vertvisc_remnant(mom6hip_ctx_t *ctx, const mom6hip_vertvisc_cs_t *cs, const mom6hip_vertvisc_type_t *visc,
                                        double *visc_rem_u, double *visc_rem_v, double dt, int32_t memspace) {
  M6_REQUIRE(ctx != nullptr, "MOM_vert_friction(remant): Module must be initialized before it is used.");
  M6_REQUIRE(cs && visc && visc_rem_u && visc_rem_v, "vertvisc_remnant: null argument");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "vertvisc_remnant: bad memspace");
  if (check_cs(cs, "vertvisc_remnant")) return 1;
  const m6::GridDev g = ctx->g;
  const Sz sz = sizes(g);
  m6::Stager st(ctx, memspace);
  double *x[2] = {st.inout(visc_rem_u, sz.u3), st.inout(visc_rem_v, sz.v3)};
  const double *a[2] = {st.in((const double *)cs->a_u, sz.ui), st.in((const double *)cs->a_v, sz.vi)};
  const double *hv[2] = {st.in((const double *)cs->h_u, sz.u3), st.in((const double *)cs->h_v, sz.v3)};
  const double *Ray[2] = {st.in(visc->Ray_u, sz.u3), st.in(visc->Ray_v, sz.v3)};
  const double *tau[2] = {nullptr, nullptr};
  double *tbot[2] = {nullptr, nullptr}, *none[2] = {nullptr, nullptr};
  M6_REQUIRE(!st.failed(), "vertvisc_remnant: staging failed");
  if (run_solve(ctx, st, cs, a, hv, Ray, nullptr, tau, none, x, tbot, dt, false)) return 1;
  return st.finish();
}
