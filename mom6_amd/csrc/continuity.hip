// continuity.hip -- continuity_PPM (src/core/MOM_continuity_PPM.F90:86-194 and everything it calls) as
// gfx950 kernels.
//
// Per direction (zonal / meridional; the reference's two sets of routines are mirror images, so the
// kernels are templated on DIR and walk an "along/cross" index pair):
//   cont_edge_kernel<DIR>   PPM_reconstruction_x/y + PPM_limit_pos | PPM_limit_CW84   (:2310-2662)
//                           one thread per cell, lanes along i -> every access is i-contiguous
//   cont_flux_kernel<DIR>   zonal/meridional_mass_flux (:519-820): one lane per FACE COLUMN (all nk layers):
//                           the layer transports and marginal areas (flux_layer :896-972), the CFL
//                           brackets, the per-column Newton/bisection solve for the barotropic velocity
//                           correction (flux_adjust :1094-1243: the reference iterates a whole row until
//                           its slowest column is done; each lane here iterates only its own column --
//                           the arithmetic per column is identical), u_cor, and the BT_cont face-area fits
//                           (set_zonal_BT_cont :1247-1410) and h_u (flux_thickness :976-1057).
//                           Lanes run along i in both directions, the k loop is inside the lane, so the
//                           k-strided loads are 512-byte rows across the wave.  Layer transports are
//                           written straight to uh/vh on every re-evaluation (no per-lane k arrays).
//   cont_conv_kernel<DIR>   continuity_*_convergence (:348-417)
// Algorithmic traffic per call: read u, v, hin, visc_rem_u/v, write h, uh, vh, u_cor, v_cor = 96 B/cell
// (SURVEY.md section 8d); this first version also materialises the edge values like the reference does
// (+32 B/cell written and re-read per direction).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.hpp"

namespace {

using m6::max2;
using m6::min2;
__device__ __forceinline__ double max3(double a, double b, double c) { return max2(max2(a, b), c); }
__device__ __forceinline__ double min3(double a, double b, double c) { return min2(min2(a, b), c); }

struct ContOpts {
  int upwind_1st, monotonic, simple_2nd, aggress_adjust, vol_CFL, better_iter, use_visc_rem_max, marginal_faces;
  double tol_eta, tol_vel, CFL_limit_adjust;
};

// Index helper for one direction.  Cells are addressed by their real (i,j); `sa` is the h-point stride of
// the along direction (1 or nih); faces by (I,j) or (i,J).
template <int DIR>
struct Dir {
  const m6::GridDev &g;
  __device__ Dir(const m6::GridDev &g_) : g(g_) {}
  __device__ __forceinline__ long sa() const { return DIR ? (long)g.nih : 1L; }
  __device__ __forceinline__ long f2(int i, int j) const { return DIR ? g.v2(i, j) : g.u2(i, j); }
  __device__ __forceinline__ long fplane() const { return DIR ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh; }
  // stride of the along direction in the FACE arrays
  __device__ __forceinline__ long fsa() const { return DIR ? (long)g.nih : 1L; }
  __device__ __forceinline__ const double *dL_face() const { return DIR ? g.dx_Cv : g.dy_Cu; }
  __device__ __forceinline__ const double *IdL_T() const { return DIR ? g.IdyT : g.IdxT; }
  __device__ __forceinline__ const double *dL_T() const { return DIR ? g.dyT : g.dxT; }
  __device__ __forceinline__ const double *dLC_face() const { return DIR ? g.dyCv : g.dxCu; }
  __device__ __forceinline__ const double *mask_face() const { return DIR ? g.mask2dCv : g.mask2dCu; }
};

// ---- edge values ---------------------------------------------------------------------------------
struct EdgeArgs {
  m6::GridDev g;
  ContOpts o;
  const double *h_in;
  double *h_L, *h_R;
  int i0, i1, j0, j1;      // cells to reconstruct (already including the +-1 along the direction)
  // open boundaries (null: none): a code per cell for this direction, PPM_reconstruction_x/y :2385-2432 / :2521-2568 --
  // bit 0: the slope is zero (the two cells either side of a segment's faces); bits 1-2: the edge values are the thickness of
  // 1 the cell itself, 2 the cell before it, 3 the cell after it along the direction (set before the limiter)
  const int32_t *cell_code;
};

// slope of a cell from its neighbours along the direction, masked (:2371-2381)
__device__ __forceinline__ double ppm_slope(double hl, double hcc, double hr, double ml, double mcc, double mr) {
  if ((ml * mcc * mr) == 0.0) return 0.0;
  double sl = 0.5 * (hr - hl);
  const double dMx = max3(hr, hl, hcc) - hcc;
  const double dMn = hcc - min3(hr, hl, hcc);
  return copysign(1., sl) * min2(fabs(sl), 2. * min2(dMx, dMn));
}

// the edge values of a cell from its own and its neighbours' thicknesses, their masks and (PPM) the three slopes, then the
// limiter: the part of PPM_reconstruction_x/y (:2383-2420) + PPM_limit_pos / PPM_limit_CW84 after the slopes
__device__ __forceinline__ void edge_finish(const ContOpts &o, double Angstrom_H, double hm, double hc, double hp, double mm, double mp,
                                            double slp_m, double slp_c, double slp_p, double &L, double &R, int obc_src = 0) {
  const double h_m1 = mm * hm + (1.0 - mm) * hc;
  const double h_p1 = mp * hp + (1.0 - mp) * hc;
  if (o.simple_2nd) {
    L = 0.5 * (h_m1 + hc);
    R = 0.5 * (h_p1 + hc);
  } else {
    const double oneSixth = 1. / 6.;
    L = 0.5 * (h_m1 + hc) + oneSixth * (slp_m - slp_c);
    R = 0.5 * (h_p1 + hc) + oneSixth * (slp_c - slp_p);
  }
  if (obc_src) { const double hv = (obc_src == 1) ? hc : ((obc_src == 2) ? hm : hp); L = hv; R = hv; }      // :2411-2432
  if (o.monotonic) {          // PPM_limit_CW84 :2625
    if ((R - hc) * (hc - L) <= 0.) {
      L = hc; R = hc;
    } else {
      const double RLdiff = R - L;
      const double RLmean = 0.5 * (R + L);
      const double FunFac = 6. * RLdiff * (hc - RLmean);
      const double RLdiff2 = RLdiff * RLdiff;
      if (FunFac > RLdiff2) L = 3. * hc - 2. * R;
      if (FunFac < -RLdiff2) R = 3. * hc - 2. * L;
    }
  } else {                      // PPM_limit_pos :2583
    const double h_min = 2.0 * Angstrom_H;
    const double curv = 3.0 * (L + R - 2.0 * hc);
    if (curv > 0.0) {
      const double dh = R - L;
      if (fabs(dh) < curv) {
        if (hc <= h_min) {
          L = hc; R = hc;
        } else if (12.0 * curv * (hc - h_min) < (curv * curv + 3.0 * (dh * dh))) {
          const double scale = 12.0 * curv * (hc - h_min) / (curv * curv + 3.0 * (dh * dh));
          L = hc + scale * (L - hc);
          R = hc + scale * (R - hc);
        }
      }
    }
  }
}

// PPM_reconstruction + limiter for one cell from its five-point stencil along the direction (:2310-2662)
__device__ __forceinline__ void edge_values(const ContOpts &o, double Angstrom_H, double hmm, double hm, double hc, double hp,
                                            double hpp, double mmm, double mm, double mc, double mp, double mpp, double &L,
                                            double &R, int code_m = 0, int code_c = 0, int code_p = 0) {
  if (o.upwind_1st) { L = hc; R = hc; return; }
  double slp_m = 0.0, slp_c = 0.0, slp_p = 0.0;
  if (!o.simple_2nd) {      // slopes of cells a-1, a, a+1 (zero beside an open-boundary segment :2385-2398)
    slp_m = (code_m & 1) ? 0.0 : ppm_slope(hmm, hm, hc, mmm, mm, mc);
    slp_c = (code_c & 1) ? 0.0 : ppm_slope(hm, hc, hp, mm, mc, mp);
    slp_p = (code_p & 1) ? 0.0 : ppm_slope(hc, hp, hpp, mc, mp, mpp);
  }
  edge_finish(o, Angstrom_H, hm, hc, hp, mm, mp, slp_m, slp_c, slp_p, L, R, (code_c >> 1) & 3);
}

// the same for two neighbouring cells a, a+1 from the six thicknesses a-2 .. a+3: the slopes of a and a+1 serve both
__device__ __forceinline__ void edge_values2(const ContOpts &o, double Angstrom_H, const double hh[6], const double mk[6], double &La,
                                             double &Ra, double &Lb, double &Rb) {
  if (o.upwind_1st) { La = hh[2]; Ra = hh[2]; Lb = hh[3]; Rb = hh[3]; return; }
  double s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0;      // slopes of cells a-1, a, a+1, a+2
  if (!o.simple_2nd) {
    s1 = ppm_slope(hh[0], hh[1], hh[2], mk[0], mk[1], mk[2]);
    s2 = ppm_slope(hh[1], hh[2], hh[3], mk[1], mk[2], mk[3]);
    s3 = ppm_slope(hh[2], hh[3], hh[4], mk[2], mk[3], mk[4]);
    s4 = ppm_slope(hh[3], hh[4], hh[5], mk[3], mk[4], mk[5]);
  }
  edge_finish(o, Angstrom_H, hh[1], hh[2], hh[3], mk[1], mk[3], s1, s2, s3, La, Ra);
  edge_finish(o, Angstrom_H, hh[2], hh[3], hh[4], mk[2], mk[4], s2, s3, s4, Lb, Rb);
}

// DIR 0: one thread per cell (the stencil is along the lanes).  DIR 1: one thread per column of EDGE_RJ rows, marching
// in j with the five-row stencil in registers, so a row of h is fetched once per block instead of five times.
constexpr int EDGE_RJ = 16;

template <int DIR>
__global__ __launch_bounds__(256) void cont_edge_kernel(EdgeArgs p) {
  const m6::GridDev &g = p.g;
  const int i = p.i0 + blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.z;
  if (i > p.i1) return;
  const long hpl = (long)g.nih * g.njh;
  const bool wide = !(p.o.upwind_1st || p.o.simple_2nd);      // the 5-point stencil is only read by the PPM branch
  if (DIR == 0) {
    const int j = p.j0 + blockIdx.y;
    const long o2 = g.h2(i, j), o3 = o2 + hpl * k;
    const double *h = p.h_in + o3;
    const double *m = g.mask2dT + o2;
    double L, R;
    int cm = 0, cc = 0, cp = 0;
    if (p.cell_code) { const int32_t *c = p.cell_code + o2; cm = c[-1]; cc = c[0]; cp = c[1]; }
    if (p.o.upwind_1st) { L = h[0]; R = h[0]; }
    else edge_values(p.o, g.Angstrom_H, wide ? h[-2] : 0.0, h[-1], h[0], h[1], wide ? h[2] : 0.0, wide ? m[-2] : 0.0, m[-1], m[0],
                     m[1], wide ? m[2] : 0.0, L, R, cm, cc, cp);
    p.h_L[o3] = L; p.h_R[o3] = R;
  } else {
    const int ja = p.j0 + blockIdx.y * EDGE_RJ;
    const int jb = min(ja + EDGE_RJ - 1, p.j1);
    const long s = g.nih;
    long o2 = g.h2(i, ja), o3 = o2 + hpl * k;
    const double *h = p.h_in + o3;
    const double *m = g.mask2dT + o2;
    if (p.o.upwind_1st) {
      for (int j = ja; j <= jb; j++, h += s, o3 += s) { p.h_L[o3] = h[0]; p.h_R[o3] = h[0]; }
      return;
    }
    double hmm = wide ? h[-2 * s] : 0.0, hm = h[-s], hc = h[0], hp = h[s];
    double mmm = wide ? m[-2 * s] : 0.0, mm = m[-s], mc = m[0], mp = m[s];
    for (int j = ja; j <= jb; j++) {
      const double hpp = wide ? h[2 * s] : 0.0, mpp = wide ? m[2 * s] : 0.0;
      double L, R;
      int cm = 0, cc = 0, cp = 0;
      if (p.cell_code) { const int32_t *c = p.cell_code + g.h2(i, j); cm = c[-s]; cc = c[0]; cp = c[s]; }
      edge_values(p.o, g.Angstrom_H, hmm, hm, hc, hp, hpp, mmm, mm, mc, mp, mpp, L, R, cm, cc, cp);
      p.h_L[o3] = L; p.h_R[o3] = R;
      hmm = hm; hm = hc; hc = hp; hp = wide ? hpp : (j < jb ? h[2 * s] : 0.0);
      mmm = mm; mm = mc; mc = mp; mp = wide ? mpp : (j < jb ? m[2 * s] : 0.0);
      h += s; m += s; o3 += s;
    }
  }
}

// ---- mass fluxes ---------------------------------------------------------------------------------
constexpr int OBC_STRIP = 3;      // faces either side of an open segment that the lane kernel forms (continuity_PPM below)
struct FluxArgs {
  m6::GridDev g;
  ContOpts o;
  const double *u, *h_in, *h_L, *h_R, *uhbt, *visc_rem;
  double *uh, *u_cor, *du_cor;
  double *FA_0m, *FA_mm, *FA_0p, *FA_pp, *uBT_mm, *uBT_pp, *h_face;   // BT_cont members of this direction
  int set_BT_cont;
  double dt;
  int fi0, fi1, fj0, fj1;    // face index ranges: zonal (I = ish-1..ieh, j = jsh..jeh); meridional (i, J)
  // open boundaries (obc_on = 0: OBC not associated; the lane-per-column kernel only)
  int obc_on, obc_open, obc_simple;      // any; OBC%open_*_BCs_exist_globally; specified_* or Flather_*_BCs_exist_globally (with OBC%OBC_pe)
  int obc_specified;                     // OBC%specified_*_BCs_exist_globally (with OBC%OBC_pe)
  int obc_dir_plus;                      // OBC_DIRECTION_E | _N: the interior cell is the face's minus-side cell
  const int32_t *segnum;                 // OBC%segnum_u | segnum_v
  const int32_t *fa_code;                // faces of the open segments of this direction over their whole range (:782-805, :1058-1088):
                                         // 1 interior = minus-side cell, 2 interior = plus-side cell (the last segment wins)
  const struct SegDev *segs;             // the segments (device table)
  int xcd_w, nbx;                        // block-cooperative kernels, meridional: blocks dealt to the 8 XCDs in strips of xcd_w block columns
                                         // (0: the launch order), of nbx real block columns (xcd_block below)
  const int32_t *skip;                   // block-cooperative kernels: faces they leave alone (null: none) -- the strips around the open
                                         // segments, which the lane kernel forms with the OBC (continuity_PPM below)
};

// what the kernels read of a segment (mom6hip_obc_segment_t with device pointers)
struct SegDev {
  int direction, open, specified, pad;
  int IsdB, IedB, JsdB, JedB, isd, ied, jsd, jed;
  const double *normal_trans, *normal_vel;
};

// segment%normal_trans / normal_vel at face (fi, fj), layer k (0-based) :632, :746
template <int DIR>
__device__ __forceinline__ double seg_val(const SegDev &S, const double *f, int fi, int fj, int k) {
  if (DIR) { const long ni = S.ied - S.isd + 1, nJ = S.JedB - S.JsdB + 1; return f[(fi - S.isd) + ni * ((fj - S.JsdB) + nJ * (long)k)]; }
  const long nI = S.IedB - S.IsdB + 1, nj = S.jed - S.jsd + 1;
  return f[(fi - S.IsdB) + nI * ((fj - S.jsd) + nj * (long)k)];
}

// flux_layer :896-972 for one face; o = offset of the minus-side cell in the h-point arrays
template <int DIR>
__device__ __forceinline__ double flux_layer(const FluxArgs &p, const Dir<DIR> &D, double u, long o3, long o2, long f2,
                                             double visc_rem, double &duhdu, bool obc = false) {
  const m6::GridDev &g = p.g;
  const long s = D.sa();
  const double dLf = D.dL_face()[f2];
  double CFL, curv_3, h_marg, uh;
  if (u > 0.0) {
    if (p.o.vol_CFL) CFL = (u * p.dt) * (dLf * g.IareaT[o2]);
    else CFL = u * p.dt * D.IdL_T()[o2];
    const double hW = p.h_L[o3], hE = p.h_R[o3], hc = p.h_in[o3];
    curv_3 = hW + hE - 2.0 * hc;
    uh = (dLf * 1.0) * u * (hE + CFL * (0.5 * (hW - hE) + curv_3 * (CFL - 1.5)));
    h_marg = hE + CFL * ((hW - hE) + 3.0 * curv_3 * (CFL - 1.0));
  } else if (u < 0.0) {
    if (p.o.vol_CFL) CFL = (-u * p.dt) * (dLf * g.IareaT[o2 + s]);
    else CFL = -u * p.dt * D.IdL_T()[o2 + s];
    const double hW = p.h_L[o3 + s], hE = p.h_R[o3 + s], hc = p.h_in[o3 + s];
    curv_3 = hW + hE - 2.0 * hc;
    uh = (dLf * 1.0) * u * (hW + CFL * (0.5 * (hE - hW) + curv_3 * (CFL - 1.5)));
    h_marg = hW + CFL * ((hE - hW) + 3.0 * curv_3 * (CFL - 1.0));
  } else {
    uh = 0.0;
    h_marg = 0.5 * (p.h_L[o3 + s] + p.h_R[o3]);
  }
  duhdu = (dLf * 1.0) * h_marg * visc_rem;
  if (obc && p.obc_open) {      // an open face carries the thickness of the interior cell :956-971 / :1854-1870
    const int l = p.segnum[f2];
    if (l != MOM6HIP_OBC_NONE && p.segs[l - 1].open) {
      const double hi = (p.segs[l - 1].direction == p.obc_dir_plus) ? p.h_in[o3] : p.h_in[o3 + s];
      uh = (dLf * 1.0) * u * hi;
      duhdu = (dLf * 1.0) * hi * visc_rem;
    }
  }
  return uh;
}

// flux_thickness :976-1057 for one face and layer (uc: u, or u_cor if present :809-815)
template <int DIR>
__device__ __forceinline__ double flux_thickness_layer(const FluxArgs &p, const Dir<DIR> &D, double uc, long o3, long o2, long f2, double vr) {
  const m6::GridDev &g = p.g;
  const long s = D.sa();
  double CFL, curv_3, h_avg, h_marg;
  if (uc > 0.0) {
    if (p.o.vol_CFL) CFL = (uc * p.dt) * (D.dL_face()[f2] * g.IareaT[o2]);
    else CFL = uc * p.dt * D.IdL_T()[o2];
    const double hW = p.h_L[o3], hE = p.h_R[o3];
    curv_3 = hW + hE - 2.0 * p.h_in[o3];
    h_avg = hE + CFL * (0.5 * (hW - hE) + curv_3 * (CFL - 1.5));
    h_marg = hE + CFL * ((hW - hE) + 3.0 * curv_3 * (CFL - 1.0));
  } else if (uc < 0.0) {
    if (p.o.vol_CFL) CFL = (-uc * p.dt) * (D.dL_face()[f2] * g.IareaT[o2 + s]);
    else CFL = -uc * p.dt * D.IdL_T()[o2 + s];
    const double hW = p.h_L[o3 + s], hE = p.h_R[o3 + s];
    curv_3 = hW + hE - 2.0 * p.h_in[o3 + s];
    h_avg = hW + CFL * (0.5 * (hE - hW) + curv_3 * (CFL - 1.5));
    h_marg = hW + CFL * ((hE - hW) + 3.0 * curv_3 * (CFL - 1.0));
  } else {
    h_avg = 0.5 * (p.h_L[o3 + s] + p.h_R[o3]);
    h_marg = 0.5 * (p.h_L[o3 + s] + p.h_R[o3]);
  }
  double hu = p.o.marginal_faces ? h_marg : h_avg;
  if (p.visc_rem) hu = hu * (vr * 1.0); else hu = hu * 1.0;
  return hu;
}

__device__ __forceinline__ double ratio_max(double a, double b, double maxrat) {
  if (fabs(a) > fabs(maxrat * b)) return maxrat;
  return a / b;
}

// flux_adjust :1094-1243 for this lane's face column.  If write_uh, the re-evaluated layer transports
// are stored to p.uh (the reference's uh_3d argument).
template <int DIR>
__device__ double flux_adjust(const FluxArgs &p, const Dir<DIR> &D, long o3_0, long o2, long f3_0, long f2, double uhbt,
                              double uh_tot_0, double duhdu_tot_0, double du_max_CFL, double du_min_CFL,
                              bool write_uh, bool obc = false) {
  const m6::GridDev &g = p.g;
  const int nz = g.nk, max_itts = 20;
  const long hpl = (long)g.nih * g.njh, fpl = D.fplane();
  double du = 0.0, du_max = du_max_CFL, du_min = du_min_CFL;
  double uh_err = uh_tot_0 - uhbt, duhdu_tot = duhdu_tot_0, uh_err_best = fabs(uh_err);
  bool do_I = true;
  const double IaT = min2(g.IareaT[o2], g.IareaT[o2 + D.sa()]);
  for (int itt = 1; itt <= max_itts; itt++) {
    double tol_eta;
    if (itt <= 1) tol_eta = 1e-6 * p.o.tol_eta;
    else if (itt == 2) tol_eta = 1e-4 * p.o.tol_eta;
    else if (itt == 3) tol_eta = 1e-2 * p.o.tol_eta;
    else tol_eta = p.o.tol_eta;
    const double tol_vel = p.o.tol_vel;
    if (uh_err > 0.0) du_max = du;
    else if (uh_err < 0.0) du_min = du;
    else do_I = false;
    bool domore = false;
    if (do_I) {
      if ((p.dt * IaT * fabs(uh_err) > tol_eta) ||
          (p.o.better_iter && ((fabs(uh_err) > tol_vel * duhdu_tot) || (fabs(uh_err) > uh_err_best)))) {
        const double ddu = -uh_err / duhdu_tot;
        const double du_prev = du;
        du = du + ddu;
        if (fabs(ddu) < 1.0e-15 * fabs(du)) {
          do_I = false;
        } else if (ddu > 0.0) {
          if (du >= du_max) {
            du = 0.5 * (du_prev + du_max);
            if (du_max - du_prev < 1.0e-15 * fabs(du)) do_I = false;
          }
        } else {
          if (du <= du_min) {
            du = 0.5 * (du_prev + du_min);
            if (du_prev - du_min < 1.0e-15 * fabs(du)) do_I = false;
          }
        }
        if (do_I) domore = true;
      } else {
        do_I = false;
      }
    }
    if (!domore) break;
    if ((itt < max_itts) || write_uh) {
      double usum = -uhbt, dsum = 0.0;
      for (int k = 0; k < nz; k++) {
        const double vr = p.visc_rem ? p.visc_rem[f3_0 + k * fpl] : 1.0;
        const double u_new = p.u[f3_0 + k * fpl] + du * vr;
        double dd;
        const double uhk = flux_layer<DIR>(p, D, u_new, o3_0 + k * hpl, o2, f2, vr, dd, obc);
        if (write_uh) p.uh[f3_0 + k * fpl] = uhk;
        usum = usum + uhk; dsum = dsum + dd;
      }
      if (itt < max_itts) {
        uh_err = usum; duhdu_tot = dsum;
        uh_err_best = min2(uh_err_best, fabs(uh_err));
      }
    }
  }
  return du;
}

template <int DIR>
__global__ __launch_bounds__(64) void cont_flux_kernel(FluxArgs p) {
  const m6::GridDev &g = p.g;
  const Dir<DIR> D(g);
  const int fi = p.fi0 + blockIdx.x * 64 + threadIdx.x;     // I (zonal) or i (meridional)
  const int fj = p.fj0 + blockIdx.y;                        // j (zonal) or J (meridional)
  if (fi > p.fi1) return;
  const int nz = g.nk;
  const long hpl = (long)g.nih * g.njh, fpl = D.fplane();
  const long s = D.sa(), fs = D.fsa();
  // minus-side cell of the face is (fi, fj) in both directions
  const long o2 = g.h2(fi, fj), o3_0 = o2;
  const long f2 = D.f2(fi, fj), f3_0 = f2;

  // open boundaries: the segment of this face, whether its transports are the external ones (simple_OBC_pt :722-734)
  const bool obc = p.obc_on != 0;
  const SegDev *S_simple = nullptr, *S_spec = nullptr;
  if (obc && (p.obc_simple || p.obc_specified)) {
    const int l = p.segnum[f2];
    if (l != MOM6HIP_OBC_NONE && p.segs[l - 1].specified) { if (p.obc_simple) S_simple = &p.segs[l - 1]; if (p.obc_specified) S_spec = &p.segs[l - 1]; }
  }
  // layer transports and marginal areas, :622-635
  double uh_tot_0 = 0.0, duhdu_tot_0 = 0.0, visc_rem_max = 0.0;
  for (int k = 0; k < nz; k++) {
    const double vr = p.visc_rem ? p.visc_rem[f3_0 + k * fpl] : 1.0;
    double dd;
    double uhk = flux_layer<DIR>(p, D, p.u[f3_0 + k * fpl], o3_0 + k * hpl, o2, f2, vr, dd, obc);
    if (S_spec) uhk = seg_val<DIR>(*S_spec, S_spec->normal_trans, fi, fj, k);      // :629-634
    p.uh[f3_0 + k * fpl] = uhk;
    duhdu_tot_0 = duhdu_tot_0 + dd;
    uh_tot_0 = uh_tot_0 + uhk;
    visc_rem_max = max2(visc_rem_max, vr);
  }
  if (p.du_cor) p.du_cor[f2] = 0.0;
  if (!(p.uhbt || p.set_BT_cont)) return;

  if (!(p.visc_rem && p.o.use_visc_rem_max)) visc_rem_max = 1.0;
  double CFL_dt = p.o.CFL_limit_adjust / p.dt;
  const double I_dt = 1.0 / p.dt;
  if (p.o.aggress_adjust) CFL_dt = I_dt;
  double I_vrm = 0.0;
  if (visc_rem_max > 0.0) I_vrm = 1.0 / visc_rem_max;
  double dx_W, dx_E;
  if (p.o.vol_CFL) {
    dx_W = ratio_max(g.areaT[o2], D.dL_face()[f2], 1000.0 * D.dL_T()[o2]);
    dx_E = ratio_max(g.areaT[o2 + s], D.dL_face()[f2], 1000.0 * D.dL_T()[o2 + s]);
  } else { dx_W = D.dL_T()[o2]; dx_E = D.dL_T()[o2 + s]; }
  double du_max_CFL = 2.0 * (CFL_dt * dx_W) * I_vrm;
  double du_min_CFL = -2.0 * (CFL_dt * dx_E) * I_vrm;
  const double mface = D.mask_face()[f2];
  for (int k = 0; k < nz; k++) {      // :663-716
    const double uk = p.u[f3_0 + k * fpl];
    if (p.visc_rem) {
      const double vr = p.visc_rem[f3_0 + k * fpl];
      if (p.o.aggress_adjust) {
        double du_lim = 0.499 * ((dx_W * I_dt - uk) + min2(0.0, p.u[f3_0 + k * fpl - fs]));
        if (du_max_CFL * vr > du_lim) du_max_CFL = du_lim / vr;
        du_lim = 0.499 * ((-dx_E * I_dt - uk) + max2(0.0, p.u[f3_0 + k * fpl + fs]));
        if (du_min_CFL * vr < du_lim) du_min_CFL = du_lim / vr;
      } else {
        if (du_max_CFL * vr > dx_W * CFL_dt - uk * mface) du_max_CFL = (dx_W * CFL_dt - uk) / vr;
        if (du_min_CFL * vr < -dx_E * CFL_dt - uk * mface) du_min_CFL = -(dx_E * CFL_dt + uk) / vr;
      }
    } else {
      if (p.o.aggress_adjust) {
        du_max_CFL = min2(du_max_CFL, 0.499 * ((dx_W * I_dt - uk) + min2(0.0, p.u[f3_0 + k * fpl - fs])));
        du_min_CFL = max2(du_min_CFL, 0.499 * ((-dx_E * I_dt - uk) + max2(0.0, p.u[f3_0 + k * fpl + fs])));
      } else {
        du_max_CFL = min2(du_max_CFL, dx_W * CFL_dt - uk);
        du_min_CFL = max2(du_min_CFL, -(dx_E * CFL_dt + uk));
      }
    }
  }
  du_max_CFL = max2(du_max_CFL, 0.0);
  du_min_CFL = min2(du_min_CFL, 0.0);

  double du = 0.0;
  if (p.uhbt) {      // :737-754 (a face whose transports are specified is left out of the adjustment: du = 0)
    if (!S_simple) du = flux_adjust<DIR>(p, D, o3_0, o2, f3_0, f2, p.uhbt[f2], uh_tot_0, duhdu_tot_0, du_max_CFL, du_min_CFL, true, obc);
    if (p.u_cor && (!p.set_BT_cont || S_simple))      // with BT_cont, u_cor is written in the pass that evaluates the three fits below
      for (int k = 0; k < nz; k++) {
        const double vr = p.visc_rem ? p.visc_rem[f3_0 + k * fpl] : 1.0;
        p.u_cor[f3_0 + k * fpl] = S_simple ? seg_val<DIR>(*S_simple, S_simple->normal_vel, fi, fj, k) : p.u[f3_0 + k * fpl] + du * vr;      // :744-748
      }
    if (p.du_cor) p.du_cor[f2] = du;
  }

  if (p.set_BT_cont && S_simple) {      // :1400-1404 (not solved for), then :759-779
    double FAuI = g.H_subroundoff * D.dL_face()[f2];
    for (int k = 0; k < nz; k++) {
      const double nv = seg_val<DIR>(*S_simple, S_simple->normal_vel, fi, fj, k);
      if ((fabs(nv) > 0.0) && S_simple->specified) FAuI = FAuI + seg_val<DIR>(*S_simple, S_simple->normal_trans, fi, fj, k) / nv;
    }
    p.FA_0m[f2] = FAuI; p.FA_0p[f2] = FAuI; p.FA_mm[f2] = FAuI; p.FA_pp[f2] = FAuI; p.uBT_mm[f2] = 0.0; p.uBT_pp[f2] = 0.0;
    if (p.h_face) {      // flux_thickness :976-1057 with u_cor (= normal_vel) if present :809-815
      for (int k = 0; k < nz; k++) {
        const double vr = p.visc_rem ? p.visc_rem[f3_0 + k * fpl] : 1.0;
        const double uc = (p.uhbt && p.u_cor) ? p.u_cor[f3_0 + k * fpl] : p.u[f3_0 + k * fpl];
        p.h_face[f3_0 + k * fpl] = flux_thickness_layer<DIR>(p, D, uc, o3_0 + k * hpl, o2, f2, vr);
      }
    }
  } else if (p.set_BT_cont) {      // set_zonal_BT_cont :1247-1410 (its own calls of flux_adjust and flux_layer do not pass OBC)
    const double min_visc_rem = 0.1, CFL_min = 1e-6;
    const double du0 = flux_adjust<DIR>(p, D, o3_0, o2, f3_0, f2, 0.0, uh_tot_0, duhdu_tot_0, du_max_CFL, du_min_CFL, false);
    const double du_CFL = (CFL_min * I_dt) * D.dLC_face()[f2];
    double duR = min2(0.0, du0 - du_CFL);
    double duL = max2(0.0, du0 + du_CFL);
    for (int k = 0; k < nz; k++) {
      const double vr = p.visc_rem ? p.visc_rem[f3_0 + k * fpl] : 1.0;
      const double visc_rem_lim = max2(vr, min_visc_rem * visc_rem_max);
      const double uk = p.u[f3_0 + k * fpl];
      if (visc_rem_lim > 0.0) {
        if (uk + duR * visc_rem_lim > -du_CFL * vr) duR = -(uk + du_CFL * vr) / visc_rem_lim;
        if (uk + duL * visc_rem_lim < du_CFL * vr) duL = -(uk - du_CFL * vr) / visc_rem_lim;
      }
    }
    double FAmt_L = 0.0, FAmt_R = 0.0, FAmt_0 = 0.0, uhtot_L = 0.0, uhtot_R = 0.0;
    const bool cor = p.uhbt && p.u_cor;
    for (int k = 0; k < nz; k++) {
      const double vr = p.visc_rem ? p.visc_rem[f3_0 + k * fpl] : 1.0;
      const double uk = p.u[f3_0 + k * fpl];
      const double u_L = uk + duL * vr, u_R = uk + duR * vr, u_0 = uk + du0 * vr;
      double d0, dL, dR;
      (void)flux_layer<DIR>(p, D, u_0, o3_0 + k * hpl, o2, f2, vr, d0);
      const double uh_L = flux_layer<DIR>(p, D, u_L, o3_0 + k * hpl, o2, f2, vr, dL);
      const double uh_R = flux_layer<DIR>(p, D, u_R, o3_0 + k * hpl, o2, f2, vr, dR);
      FAmt_0 = FAmt_0 + d0; FAmt_L = FAmt_L + dL; FAmt_R = FAmt_R + dR;
      uhtot_L = uhtot_L + uh_L; uhtot_R = uhtot_R + uh_R;
      // u_cor (:744-748) and flux_thickness (:976-1057, with u_cor if present :809-815) ride on the same layer data
      double uc = uk;
      if (cor) { uc = uk + du * vr; p.u_cor[f3_0 + k * fpl] = uc; }
      if (p.h_face) p.h_face[f3_0 + k * fpl] = flux_thickness_layer<DIR>(p, D, uc, o3_0 + k * hpl, o2, f2, vr);
    }
    double FA_0 = FAmt_0, FA_avg = FAmt_0;
    if ((duL - du0) != 0.0) FA_avg = uhtot_L / (duL - du0);
    if (FA_avg > max2(FA_0, FAmt_L)) FA_avg = max2(FA_0, FAmt_L);
    else if (FA_avg < min2(FA_0, FAmt_L)) FA_0 = FA_avg;
    p.FA_0m[f2] = FA_0; p.FA_mm[f2] = FAmt_L;
    if (fabs(FA_0 - FAmt_L) <= 1e-12 * FA_0) p.uBT_mm[f2] = 0.0;
    else p.uBT_mm[f2] = (1.5 * (duL - du0)) * ((FAmt_L - FA_avg) / (FAmt_L - FA_0));

    FA_0 = FAmt_0; FA_avg = FAmt_0;
    if ((duR - du0) != 0.0) FA_avg = uhtot_R / (duR - du0);
    if (FA_avg > max2(FA_0, FAmt_R)) FA_avg = max2(FA_0, FAmt_R);
    else if (FA_avg < min2(FA_0, FAmt_R)) FA_0 = FA_avg;
    p.FA_0p[f2] = FA_0; p.FA_pp[f2] = FAmt_R;
    if (fabs(FAmt_R - FA_0) <= 1e-12 * FA_0) p.uBT_pp[f2] = 0.0;
    else p.uBT_pp[f2] = (1.5 * (duR - du0)) * ((FAmt_R - FA_avg) / (FAmt_R - FA_0));

  }
  // the faces of open segments: the face areas and thicknesses of the interior cell (:782-805 / :1058-1088, after every row in the
  // reference: the last word on these faces)
  if (obc && p.obc_open && p.set_BT_cont && p.fa_code[f2]) {
    const long oi = (p.fa_code[f2] == 1) ? o3_0 : o3_0 + s;
    const double dLf = D.dL_face()[f2];
    double FA_u = 0.0;
    for (int k = 0; k < nz; k++) FA_u = FA_u + p.h_in[oi + k * hpl] * (dLf * 1.0);
    p.FA_0m[f2] = FA_u; p.FA_0p[f2] = FA_u; p.FA_mm[f2] = FA_u; p.FA_pp[f2] = FA_u; p.uBT_mm[f2] = 0.0; p.uBT_pp[f2] = 0.0;
    if (p.h_face)
      for (int k = 0; k < nz; k++)
        p.h_face[f3_0 + k * fpl] = p.visc_rem ? p.h_in[oi + k * hpl] * (p.visc_rem[f3_0 + k * fpl] * 1.0) : p.h_in[oi + k * hpl] * 1.0;
  }
}

// The block a workgroup works on.  Workgroups are dealt round-robin to the 8 XCDs in launch order (x fastest), each XCD with an L2 of its
// own; a meridional block reads six rows of h of which the block of the next row reads five again.  With xcd_w > 0 the launch has 8 * xcd_w
// block columns and the workgroups of one XCD (launch index mod 8) walk a strip of xcd_w adjacent real columns row by row, so a row's
// re-reads hit that XCD's L2 instead of arriving once per XCD.  Placement only: any mapping gives the same results.  Returns false for
// the workgroups beyond the last real column (they leave before any barrier).
__device__ __forceinline__ bool xcd_block(int xcd_w, int nbx, int &bx, int &by) {
  bx = blockIdx.x; by = blockIdx.y;
  if (xcd_w <= 0) return true;
  const unsigned L = blockIdx.x + gridDim.x * blockIdx.y;
  const unsigned xcd = L & 7u, slot = L >> 3;
  bx = (int)(xcd * xcd_w + slot % xcd_w); by = (int)(slot / xcd_w);
  return bx < nbx;
}

// ---- mass fluxes, block-cooperative form -----------------------------------------------------------
// cont_flux_kernel re-reads five 3-D arrays on every pass over k (the first evaluation, every Newton iteration of
// flux_adjust, the BT_cont fits): 3.6x the algorithmic bytes on the 1/4-degree grid, and that traffic is its run time.
// Here a block of FC_NW = 4 waves owns 32 face columns and keeps them in REGISTERS for all passes: every half-wave (32
// lanes = the 32 faces) holds one slab of KS layers (u, visc_rem and the edge values / thickness of the cells on both
// sides: 8 doubles a layer), eight slabs per block, so every 3-D input is read once.  (254 VGPRs and 65 KB of LDS: two
// blocks per CU, whose load and arithmetic phases interleave.)  A pass evaluates flux_layer for the thread's own layers
// and leaves the layer values in LDS; the reference's k-ordered sums are then formed by ONE half-wave per sum, adding the
// nk values of each face in order (same rounding as the reference's loop), and handed to all threads through LDS.  The
// scalar Newton / bracket logic of each face is run redundantly by the eight half-waves (same inputs, same instructions,
// same result), which keeps the loop trip count uniform over the block without any further exchange.  The two loops
// whose state runs through k (the CFL brackets :663-716 and the duL/duR limits of set_*_BT_cont) are serial chains
// walked by one (half-)wave; the divisions of the bracket chain do not depend on the running state and are formed by all
// threads beforehand.
#ifndef FC_NW_DEF
#define FC_NW_DEF 4
#endif
constexpr int FC_NW = FC_NW_DEF;      // waves per block (experiments: -DFC_NW_DEF=2 -DFC_OCC=2: 20 layers a thread, one wave a SIMD)
constexpr int FC_KSMAX = 80 / (2 * FC_NW);      // layers per thread of the largest instance
constexpr int FC_FL = 32;     // face columns per block: a wave is two half-waves of 32 faces holding different layers
constexpr int FC_NS = 2 * FC_NW;      // layer slabs per block (one per half-wave)

struct FaceConst { double dLf, cm, cp, dt; };

// Marks a register value as redefined here, so that products of it are formed where the reference forms them (inside
// the pass) instead of being hoisted out of the Newton loop and held in registers for every layer.
__device__ __forceinline__ void pin(double &x) { asm volatile("" : "+v"(x)); }      // cm / cp: the CFL factor of the minus / plus side cell

// flux_layer :896-972 from the per-layer register set: upwind edge value, edge difference and curvature of the cell on
// either side (mE, mD = h_L - h_R, mC = (h_L + h_R) - 2 h of the minus-side cell; pW, pD = h_R - h_L, pC of the plus-side
// cell) -- the sub-expressions the reference forms first, so the arithmetic that follows is its own.  The upwind side
// is picked with selects (|u| dt = u dt or -u dt, bit for bit), the u = 0 branch by a final select.
__device__ __forceinline__ double flux_reg(const FaceConst &F, double u, double vr, double mE, double mD, double mC, double pW,
                                           double pD, double pC, double &duhdu) {
  const bool pos = u > 0.0;
  const double E = pos ? mE : pW, Dd = pos ? mD : pD, Cc = pos ? mC : pC, cf = pos ? F.cm : F.cp;
  const double CFL = (fabs(u) * F.dt) * cf;
  double uh = (F.dLf * 1.0) * u * (E + CFL * (0.5 * Dd + Cc * (CFL - 1.5)));
  double h_marg = E + CFL * (Dd + 3.0 * Cc * (CFL - 1.0));
  if (u == 0.0) { uh = 0.0; h_marg = 0.5 * (pW + mE); }
  duhdu = (F.dLf * 1.0) * h_marg * vr;
  return uh;
}

// The k-ordered sums of the layer values the waves left in the LDS planes 0 .. nsum-1, each started from its init value:
// half-wave q < nsum adds plane q; every thread gets all results.
__device__ __forceinline__ void ksums(double *fsm, int plane, int roff, int fl, int sb, int nz, int nsum, double i0, double i1,
                                      double i2, double &r0, double &r1, double &r2) {
  __syncthreads();
  if (sb < nsum) {
    double acc = (sb == 0) ? i0 : ((sb == 1) ? i1 : i2);
    const int base = sb * plane + fl;
    for (int k = 0; k < nz; k++) acc = acc + fsm[base + k * FC_FL];
    fsm[roff + sb * FC_FL + fl] = acc;
  }
  __syncthreads();
  r0 = fsm[roff + fl];
  r1 = (nsum > 1) ? fsm[roff + FC_FL + fl] : 0.;
  r2 = (nsum > 2) ? fsm[roff + 2 * FC_FL + fl] : 0.;
}

#ifdef FC_TRACE
// phase clock of the cooperative kernel (experiments only: tools/build_variant.sh ... -DFC_TRACE): thread 0 of every block adds
// the s_memtime ticks between marks; slot 15 counts blocks, slot 14 counts Newton passes; slots 16-39: faces by the number of
// evaluation passes they were alive for in a solve, slots 40-63: the solves of the blocks by the number of passes they ran
__device__ unsigned long long fc_trace[128];      // [direction][slot]
#define FC_MARK(n) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(&fc_trace[64 * DIR + (n)], t_ - fc_t0); fc_t0 = t_; } } while (0)
#else
#define FC_MARK(n) do { } while (0)
#endif

template <int DIR, int KS>
#ifndef FC_OCC
#define FC_OCC 2      // blocks per CU the register budget is set for (experiments: tools/build_variant.sh ... -DFC_OCC=1)
#endif
__global__ __launch_bounds__(64 * FC_NW, FC_OCC) void cont_flux_coop_kernel(FluxArgs p) {
#ifdef FC_TRACE
  unsigned long long fc_t0 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) atomicAdd(&fc_trace[64 * DIR + 15], 1ull);
#endif
  extern __shared__ double fsm[];      // three [KS*FC_NS][FC_FL] planes of layer values, then results [8][FC_FL], visc_rem max [FC_NS][FC_FL]
  const m6::GridDev &g = p.g;
  const Dir<DIR> D(g);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int fl = lane & (FC_FL - 1), sb = 2 * w + (lane >> 5);      // face within the block; layer slab of this half-wave
  const int nz = g.nk;
  constexpr int FPB = (DIR == 0) ? FC_FL - 1 : FC_FL;      // faces per block
  int bx, by;
  if (!xcd_block(p.xcd_w, p.nbx, bx, by)) return;
  const int fi_raw = p.fi0 + bx * FPB + fl;
  const bool valid = fl < FPB && fi_raw <= p.fi1 && !(p.skip && p.skip[D.f2(min(fi_raw, p.fi1), p.fj0 + by)]);
  // lanes past the row (and the last lane of a zonal half-wave) do everything but store, so barriers stay uniform; they sit
  // on the cell after the last face, whose reconstruction the last face needs, and never iterate (see `alive`)
  // (meridionally nothing is handed between lanes: idle lanes sit on the last face, which is always inside the row)
  const int fi_last = (DIR == 0) ? p.fi1 + 1 : p.fi1;
  const int fi = (fi_raw <= fi_last) ? fi_raw : fi_last;
  const int fj = p.fj0 + by;
  const long hpl = (long)g.nih * g.njh, fpl = D.fplane();
  const long s = D.sa(), fs = D.fsa();
  const long o2 = g.h2(fi, fj), f2 = D.f2(fi, fj);
  constexpr int PL = KS * FC_NS * FC_FL;          // one plane
  constexpr int RO = 3 * PL, VO = RO + 8 * FC_FL; // results, visc_rem max
  constexpr int CO = VO + FC_NS * FC_FL;          // per-face values parked between the phases (instead of held in registers)
  constexpr int PK_DXW = CO + 6 * FC_FL, PK_DXE = CO + 7 * FC_FL, PK_MF = CO + 8 * FC_FL, PK_IAT = CO + 9 * FC_FL,
                PK_UHBT = CO + 10 * FC_FL, PK_DLC = CO + 11 * FC_FL, PK_DLF = CO + 12 * FC_FL, PK_CM = CO + 13 * FC_FL,
                PK_CP = CO + 14 * FC_FL;
  // the bracket and best-error of the Newton iteration, one private copy per half-wave (every half-wave runs the scalar
  // logic of its faces with the same values): in LDS instead of registers that would be spilled to scratch memory
  constexpr int PV = CO + 15 * FC_FL;
  const int pv = PV + sb * FC_FL + fl;
  constexpr int PVS = FC_NS * FC_FL;      // stride between the private variables
  const int k0 = sb * KS;
  const int sl = k0 * FC_FL + fl;                 // this thread's slot of layer k0 in a plane

  // ---- the wave's layers into registers
  // The per-face metrics and masks are loaded without conditions (an address select instead of a branch where an option
  // decides whether a value is read at all) and used only after the layer loads below have been issued: a branch on an
  // option around a load is a memory round trip of its own in front of everything that follows.
  const bool wide = !(p.o.upwind_1st || p.o.simple_2nd);      // the 5-point stencil is only read by the PPM branch
  const long s2w = wide ? 2 * s : 0, s3w = (wide && DIR == 1) ? 3 * s : 0;
  double dLf_r = D.dL_face()[f2];
  double cm_r = (p.o.vol_CFL ? g.IareaT : D.IdL_T())[o2], cp_r = (p.o.vol_CFL ? g.IareaT : D.IdL_T())[o2 + s];
  double mk[6];                                               // mask2dT of cells -2 .. +3 along the direction
  {
    const double *mm = g.mask2dT + o2;
    mk[0] = mm[-s2w]; mk[1] = mm[-s]; mk[2] = mm[0]; mk[3] = mm[s]; mk[4] = mm[2 * s];
    mk[5] = mm[s3w];      // (only the meridional pair of reconstructions reads cell +3)
  }
  // what the later phases read of the 2-D arrays, fetched with everything else and parked in LDS (slots PK_*): a load where
  // it is used is a memory round trip on the critical path of the block, with nothing to overlap it
  double pk_aW = g.areaT[o2], pk_aE = g.areaT[o2 + s], pk_dW = D.dL_T()[o2], pk_dE = D.dL_T()[o2 + s], pk_mf = D.mask_face()[f2];
  double pk_iW = g.IareaT[o2], pk_iE = g.IareaT[o2 + s], pk_ub = p.uhbt ? p.uhbt[f2] : 0.0, pk_lc = D.dLC_face()[f2];
  double ru[KS], rvr[KS], mE[KS], mD[KS], mC[KS], pW[KS], pD[KS], pC[KS];
  // Every global load of the block is issued before anything is computed from one of them: the reconstruction below is
  // branchy code the compiler cannot move loads across, and a load -- wait -- compute sequence per layer is KS memory round
  // trips in a row with two waves a SIMD to hide them.  (Layers past nz read layer nz-1 and are zeroed afterwards, so the
  // load block has no branches.)
  constexpr int NH = (DIR == 0) ? 5 : 6;      // cells -2 .. +2 along the direction (+3 for the meridional pair)
  double hr[KS][NH];
  {
    // addresses as a uniform base (the array, displaced along the direction: scalar registers) plus one 32-bit byte offset
    // per layer: the `global_load v, v_offset, s[base]` form, one VGPR per layer instead of a 64-bit address per load
    // (launch_flux only takes this kernel when a 3-D array is below 4 GB)
    const char *ub = (const char *)p.u, *vb = p.visc_rem ? (const char *)p.visc_rem : ub;      // (no branch around a load)
    const char *hb[NH];
#pragma unroll
    for (int q = 0; q < NH; q++) {
      const long disp = (q == 0) ? -s2w : ((q == 5) ? s3w : (q - 2) * s);
      hb[q] = (const char *)(p.h_in + disp);
    }
    const unsigned o2b = (unsigned)(o2 * 8), f2b = (unsigned)(f2 * 8), hstep = (unsigned)(hpl * 8), fstep = (unsigned)(fpl * 8);
#pragma unroll
    for (int m = 0; m < KS; m++) {
      const unsigned k = (unsigned)((k0 + m < nz) ? k0 + m : nz - 1);
      const unsigned vh = o2b + k * hstep, vf = f2b + k * fstep;
      ru[m] = *(const double *)(ub + vf);
      rvr[m] = *(const double *)(vb + vf);
#pragma unroll
      for (int q = 0; q < NH; q++) hr[m][q] = *(const double *)(hb[q] + vh);
    }
  }
#pragma unroll
  for (int m = 0; m < KS; m++) {      // (a use of every loaded value here: none of the loads can sink into the code below)
    pin(ru[m]); pin(rvr[m]);
#pragma unroll
    for (int q = 0; q < NH; q++) pin(hr[m][q]);
  }
  pin(dLf_r); pin(cm_r); pin(cp_r);
  FC_MARK(0);
  if (!p.visc_rem) {
#pragma unroll
    for (int m = 0; m < KS; m++) rvr[m] = 1.0;
  }
#pragma unroll
  for (int q = 0; q < 6; q++) pin(mk[q]);
  if (!wide) { mk[0] = 0.0; mk[5] = 0.0; }
  if (DIR == 0) mk[5] = 0.0;
  FaceConst F0;
  F0.dLf = dLf_r; F0.dt = p.dt;
  F0.cm = p.o.vol_CFL ? (F0.dLf * cm_r) : cm_r;      // the CFL factor of the minus / plus side cell
  F0.cp = p.o.vol_CFL ? (F0.dLf * cp_r) : cp_r;
  pin(pk_aW); pin(pk_aE); pin(pk_dW); pin(pk_dE); pin(pk_mf); pin(pk_iW); pin(pk_iE); pin(pk_ub); pin(pk_lc);
  if (sb == 0) {      // (read after the barriers of the first k-ordered sums)
    double dxw, dxe;
    if (p.o.vol_CFL) {
      dxw = ratio_max(pk_aW, F0.dLf, 1000.0 * pk_dW);
      dxe = ratio_max(pk_aE, F0.dLf, 1000.0 * pk_dE);
    } else { dxw = pk_dW; dxe = pk_dE; }
    fsm[PK_DXW + fl] = dxw; fsm[PK_DXE + fl] = dxe; fsm[PK_MF + fl] = pk_mf; fsm[PK_IAT + fl] = min2(pk_iW, pk_iE);
    fsm[PK_UHBT + fl] = pk_ub; fsm[PK_DLC + fl] = pk_lc;
    fsm[PK_DLF + fl] = F0.dLf; fsm[PK_CM + fl] = F0.cm; fsm[PK_CP + fl] = F0.cp;
  }
  // ---- blocks whose 32 faces are all land: nothing to solve.  With dy_Cu (dx_Cv) = 0, the face masked, no velocity in any
  // layer and no barotropic transport asked of it, every pass of the reference multiplies by the zero face length: uh = +0 in
  // every layer (u = 0 takes flux_layer's own branch), the Newton iteration starts converged (uh_err = 0: du = 0), the
  // BT_cont sums are zero whatever the limits duL / duR are (FA = +0, uBT = 0: the branches below with all sums +0).  Only
  // h_u / h_v (flux_thickness has no mask) and u_cor = u + 0*visc_rem still need the layer data.  30 % of the 1/4-degree grid
  // is land in contiguous continents; a block-uniform vote keeps the barriers of the live blocks uniform.
  bool dead_lane = (pk_mf == 0.0) & (dLf_r == 0.0) & (pk_ub == 0.0);
#pragma unroll
  for (int m = 0; m < KS; m++) dead_lane = dead_lane & ((k0 + m >= nz) | (ru[m] == 0.0));
  const bool dead = __syncthreads_and(dead_lane ? 1 : 0) != 0;
  if (dead && !p.set_BT_cont) {
    const double du_dead = 0.0;
    if (valid) {
#pragma unroll
      for (int m = 0; m < KS; m++) {
        if (k0 + m < nz) {
          p.uh[f2 + (k0 + m) * fpl] = 0.0;
          if (p.uhbt && p.u_cor) p.u_cor[f2 + (k0 + m) * fpl] = ru[m] + du_dead * rvr[m];
        }
      }
      if (sb == 0 && p.du_cor) p.du_cor[f2] = p.uhbt ? du_dead : 0.0;
    }
    return;
  }
#pragma unroll
  for (int m = 0; m < KS; m++) {
    const int k = k0 + m;
    if (k < nz) {
      // the edge values of the two cells (cont_edge_kernel's arithmetic, PPM_reconstruction_x/y :2310-2662) from the
      // thicknesses along the direction: h_L / h_R never go through memory.  Zonal: the plus-side cell of a face is the
      // minus-side cell of the next lane's face, so each lane reconstructs one cell and hands it down one lane (the
      // last lane of a half-wave only serves its neighbour; blocks advance by FPB = 31 faces).  Meridional: both cells.
      const double hm1 = hr[m][1], hc0 = hr[m][2], hp1 = hr[m][3], hp2 = hr[m][4];
      double Lm, Rm, Lp, Rp;
      if (DIR == 0) {
        if (p.o.upwind_1st) { Lm = hc0; Rm = hc0; }
        else edge_values(p.o, g.Angstrom_H, hr[m][0], hm1, hc0, hp1, wide ? hp2 : 0.0, mk[0], mk[1], mk[2], mk[3],
                         wide ? mk[4] : 0.0, Lm, Rm);
        Lp = __shfl_down(Lm, 1); Rp = __shfl_down(Rm, 1);
      } else {      // both cells at once: the slopes of the two cells serve both reconstructions
        const double h6[6] = {hr[m][0], hm1, hc0, hp1, hp2, hr[m][NH - 1]};
        edge_values2(p.o, g.Angstrom_H, h6, mk, Lm, Rm, Lp, Rp);
      }
      mE[m] = Rm; mD[m] = Lm - Rm; mC[m] = Lm + Rm - 2.0 * hc0;
      pW[m] = Lp; pD[m] = Rp - Lp; pC[m] = Lp + Rp - 2.0 * hp1;
    } else {
      ru[m] = 0.; rvr[m] = 0.; mE[m] = 0.; mD[m] = 0.; mC[m] = 0.; pW[m] = 0.; pD[m] = 0.; pC[m] = 0.;
    }
  }

  if (dead) {      // (set_BT_cont: the layer data are needed for h_u / h_v and u_cor; every sum of the fits is +0)
    const double du_dead = 0.0;
    const bool cor_d = p.uhbt && p.u_cor;
#pragma unroll
    for (int m = 0; m < KS; m++) {
      if (k0 + m < nz) {
        const long f3 = f2 + (k0 + m) * fpl;
        const double vr = rvr[m], uk = ru[m];
        if (valid) p.uh[f3] = 0.0;
        double uc = uk;
        if (cor_d) { uc = uk + du_dead * vr; if (valid) p.u_cor[f3] = uc; }
        if (p.h_face) {      // flux_thickness :976-1057 at uc = 0 (or whatever u + 0*visc_rem is)
          const bool pos = uc > 0.0;
          const double E = pos ? mE[m] : pW[m], Dd = pos ? mD[m] : pD[m], Cc = pos ? mC[m] : pC[m], cf = pos ? F0.cm : F0.cp;
          const double CFL = (fabs(uc) * p.dt) * cf;
          double h_avg = E + CFL * (0.5 * Dd + Cc * (CFL - 1.5));
          double h_marg = E + CFL * (Dd + 3.0 * Cc * (CFL - 1.0));
          if (uc == 0.0) { h_avg = 0.5 * (pW[m] + mE[m]); h_marg = 0.5 * (pW[m] + mE[m]); }
          double hu = p.o.marginal_faces ? h_marg : h_avg;
          if (p.visc_rem) hu = hu * (vr * 1.0); else hu = hu * 1.0;
          if (valid) p.h_face[f3] = hu;
        }
      }
    }
    if (sb == 0 && valid) {
      if (p.du_cor) p.du_cor[f2] = p.uhbt ? du_dead : 0.0;
      p.FA_0m[f2] = 0.0; p.FA_mm[f2] = 0.0; p.uBT_mm[f2] = 0.0;
      p.FA_0p[f2] = 0.0; p.FA_pp[f2] = 0.0; p.uBT_pp[f2] = 0.0;
    }
    return;
  }

  FC_MARK(1);
  // ---- layer transports and marginal areas, :622-635
  double vmax_w = 0.0;
#pragma unroll
  for (int m = 0; m < KS; m++) {
    if (k0 + m < nz) {
      double dd;
      const double uhk = flux_reg(F0, ru[m], rvr[m], mE[m], mD[m], mC[m], pW[m], pD[m], pC[m], dd);
      if (valid && !p.uhbt) p.uh[f2 + (k0 + m) * fpl] = uhk;      // (with uhbt, uh is stored once, after the solve)
      fsm[sl + m * FC_FL] = uhk; fsm[PL + sl + m * FC_FL] = dd;
      vmax_w = max2(vmax_w, rvr[m]);
    }
  }
  fsm[VO + sb * FC_FL + fl] = vmax_w;
  double uh_tot_0, duhdu_tot_0, dummy;
  ksums(fsm, PL, RO, fl, sb, nz, 2, 0.0, 0.0, 0.0, uh_tot_0, duhdu_tot_0, dummy);
  double visc_rem_max = 0.0;
  for (int q = 0; q < FC_NS; q++) visc_rem_max = max2(visc_rem_max, fsm[VO + q * FC_FL + fl]);      // (max is order-free)
  if (!(p.visc_rem && p.o.use_visc_rem_max)) visc_rem_max = 1.0;

  FC_MARK(2);
  // ---- the CFL brackets of the velocity correction, :637-716: two chains through k over u and visc_rem staged in LDS
  double CFL_dt = p.o.CFL_limit_adjust / p.dt;
  const double I_dt = 1.0 / p.dt;
  if (p.o.aggress_adjust) CFL_dt = I_dt;
  double I_vrm = 0.0;
  if (visc_rem_max > 0.0) I_vrm = 1.0 / visc_rem_max;
  const double dx_W = fsm[PK_DXW + fl], dx_E = fsm[PK_DXE + fl], mface = fsm[PK_MF + fl];
  // The state of these loops runs through k, so one wave walks them; what it needs per layer -- the bound the bracket
  // is tested against and the value it takes when the test fails (a division) -- does not depend on that state and is
  // formed by all waves for their own layers first: planes 0 / 1 / 2 = visc_rem, bound, new bracket.
  // (no barrier needed before the planes are rewritten: every read of them lies between the two barriers of ksums)
#pragma unroll
  for (int m = 0; m < KS; m++) {
    if (k0 + m < nz) {
      const double uk = ru[m], vr = rvr[m];
      double bound, cand;
      if (p.o.aggress_adjust) {
        bound = 0.499 * ((dx_W * I_dt - uk) + min2(0.0, p.u[f2 + (k0 + m) * fpl - fs]));
        cand = p.visc_rem ? bound / vr : bound;
      } else if (p.visc_rem) {
        bound = dx_W * CFL_dt - uk * mface; cand = (dx_W * CFL_dt - uk) / vr;
      } else {
        bound = dx_W * CFL_dt - uk; cand = bound;
      }
      fsm[sl + m * FC_FL] = vr; fsm[PL + sl + m * FC_FL] = bound; fsm[2 * PL + sl + m * FC_FL] = cand;
    }
  }
  __syncthreads();
  if (sb == 0) {
    double du_max_CFL = 2.0 * (CFL_dt * dx_W) * I_vrm;
    if (p.visc_rem) {
#pragma unroll 8
      for (int k = 0; k < nz; k++) {      // (all three loads up front: no LDS round trip inside the dependent chain)
        const double vr = fsm[k * FC_FL + fl], bound = fsm[PL + k * FC_FL + fl], cand = fsm[2 * PL + k * FC_FL + fl];
        du_max_CFL = (du_max_CFL * vr > bound) ? cand : du_max_CFL;
      }
    } else {
#pragma unroll 8
      for (int k = 0; k < nz; k++) du_max_CFL = min2(du_max_CFL, fsm[2 * PL + k * FC_FL + fl]);
    }
    fsm[CO + fl] = max2(du_max_CFL, 0.0);
    // park the per-face values the later phases read once each (every half-wave holds the same values)
    fsm[CO + 2 * FC_FL + fl] = uh_tot_0; fsm[CO + 3 * FC_FL + fl] = duhdu_tot_0; fsm[CO + 4 * FC_FL + fl] = visc_rem_max;
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < KS; m++) {
    if (k0 + m < nz) {
      const double uk = ru[m], vr = rvr[m];
      double bound, cand;
      if (p.o.aggress_adjust) {
        bound = 0.499 * ((-dx_E * I_dt - uk) + max2(0.0, p.u[f2 + (k0 + m) * fpl + fs]));
        cand = p.visc_rem ? bound / vr : bound;
      } else if (p.visc_rem) {
        bound = -dx_E * CFL_dt - uk * mface; cand = -(dx_E * CFL_dt + uk) / vr;
      } else {
        bound = -(dx_E * CFL_dt + uk); cand = bound;
      }
      fsm[PL + sl + m * FC_FL] = bound; fsm[2 * PL + sl + m * FC_FL] = cand;
    }
  }
  __syncthreads();
  if (sb == 0) {
    double du_min_CFL = -2.0 * (CFL_dt * dx_E) * I_vrm;
    if (p.visc_rem) {
#pragma unroll 8
      for (int k = 0; k < nz; k++) {
        const double vr = fsm[k * FC_FL + fl], bound = fsm[PL + k * FC_FL + fl], cand = fsm[2 * PL + k * FC_FL + fl];
        du_min_CFL = (du_min_CFL * vr < bound) ? cand : du_min_CFL;
      }
    } else {
#pragma unroll 8
      for (int k = 0; k < nz; k++) du_min_CFL = max2(du_min_CFL, fsm[2 * PL + k * FC_FL + fl]);
    }
    fsm[CO + FC_FL + fl] = min2(du_min_CFL, 0.0);
  }
  __syncthreads();
  FC_MARK(3);

  // ---- flux_adjust :1094-1243 for the faces of the block: phase 0 matches uhbt (:737-754, storing the transports),
  // phase 1 finds the correction that gives no net transport for set_*_BT_cont (:1290-1292)
  double du_ph[2] = {0.0, 0.0};
#pragma unroll 1
  for (int phase = 0; phase < 2; phase++) {
    if (phase == 0 ? (p.uhbt == nullptr) : !p.set_BT_cont) continue;
    const bool write_uh = (phase == 0);
    const double uhbt = (phase == 0) ? fsm[PK_UHBT + fl] : 0.0;
    const int max_itts = 20;
    double du = 0.0;
    double uh_err = fsm[CO + 2 * FC_FL + fl] - uhbt, duhdu_tot = fsm[CO + 3 * FC_FL + fl];
    fsm[pv] = fsm[CO + fl]; fsm[pv + PVS] = fsm[CO + FC_FL + fl]; fsm[pv + 2 * PVS] = fabs(uh_err);      // du_max, du_min, uh_err_best
    fsm[pv + 3 * PVS] = 0.0;      // du_eval: du of this face's last re-evaluation of the transports (what uh_3d holds in the reference)
    bool do_I = true, alive = valid;
#ifdef FC_TRACE
    int fc_np_face = 0, fc_np_block = 0;
#endif
#pragma unroll 1
    for (int itt = 1; itt <= max_itts; itt++) {
      bool domore = false;
      if (alive) {
        double tol_eta;
        if (itt <= 1) tol_eta = 1e-6 * p.o.tol_eta;
        else if (itt == 2) tol_eta = 1e-4 * p.o.tol_eta;
        else if (itt == 3) tol_eta = 1e-2 * p.o.tol_eta;
        else tol_eta = p.o.tol_eta;
        const double tol_vel = p.o.tol_vel;
        double du_max = fsm[pv], du_min = fsm[pv + PVS];
        const double uh_err_best = fsm[pv + 2 * PVS], IaT = fsm[PK_IAT + fl];
        if (uh_err > 0.0) { du_max = du; fsm[pv] = du; }
        else if (uh_err < 0.0) { du_min = du; fsm[pv + PVS] = du; }
        else do_I = false;
        if (do_I) {
          if ((p.dt * IaT * fabs(uh_err) > tol_eta) ||
              (p.o.better_iter && ((fabs(uh_err) > tol_vel * duhdu_tot) || (fabs(uh_err) > uh_err_best)))) {
            const double ddu = -uh_err / duhdu_tot;
            const double du_prev = du;
            du = du + ddu;
            if (fabs(ddu) < 1.0e-15 * fabs(du)) {
              do_I = false;
            } else if (ddu > 0.0) {
              if (du >= du_max) {
                du = 0.5 * (du_prev + du_max);
                if (du_max - du_prev < 1.0e-15 * fabs(du)) do_I = false;
              }
            } else {
              if (du <= du_min) {
                du = 0.5 * (du_prev + du_min);
                if (du_prev - du_min < 1.0e-15 * fabs(du)) do_I = false;
              }
            }
            if (do_I) domore = true;
          } else {
            do_I = false;
          }
        }
        if (!domore) alive = false;
      }
      if (!__any(alive)) break;      // the same in every wave of the block: they hold the same values
      if ((itt < max_itts) || write_uh) {
#ifdef FC_TRACE
        if (threadIdx.x == 0) atomicAdd(&fc_trace[64 * DIR + 14], 1ull);
        fc_np_block++; if (alive) fc_np_face++;
#endif
        if (alive) fsm[pv + 3 * PVS] = du;
        FaceConst F; F.dLf = fsm[PK_DLF + fl]; F.cm = fsm[PK_CM + fl]; F.cp = fsm[PK_CP + fl]; F.dt = p.dt;
#pragma unroll
        for (int m = 0; m < KS; m++) {
          if (k0 + m < nz) {
            double dd;
            pin(mD[m]); pin(mC[m]); pin(pD[m]); pin(pC[m]);
            const double uhk = flux_reg(F, ru[m] + du * rvr[m], rvr[m], mE[m], mD[m], mC[m], pW[m], pD[m], pC[m], dd);
            fsm[sl + m * FC_FL] = uhk; fsm[PL + sl + m * FC_FL] = dd;
          }
        }
        double usum, dsum, d2;
        ksums(fsm, PL, RO, fl, sb, nz, 2, -uhbt, 0.0, 0.0, usum, dsum, d2);
        if (alive && itt < max_itts) {
          uh_err = usum; duhdu_tot = dsum;
          fsm[pv + 2 * PVS] = min2(fsm[pv + 2 * PVS], fabs(uh_err));
        }
      }
    }
    du_ph[phase] = du;
#ifdef FC_TRACE
    if (sb == 0 && valid) atomicAdd(&fc_trace[64 * DIR + 16 + min(fc_np_face, 23)], 1ull);
    if (threadIdx.x == 0) atomicAdd(&fc_trace[64 * DIR + 40 + min(fc_np_block, 23)], 1ull);
#endif
    FC_MARK(4 + phase);
    if (write_uh && valid) {
      // The reference stores the layer transports on every re-evaluation (uh_3d); what remains is the last one of each
      // face, or the first evaluation for a face that never iterated (du_eval = 0: u + 0*visc_rem is u).  Stored once here.
      const double du_eval = fsm[pv + 3 * PVS];
      FaceConst F; F.dLf = fsm[PK_DLF + fl]; F.cm = fsm[PK_CM + fl]; F.cp = fsm[PK_CP + fl]; F.dt = p.dt;
#pragma unroll
      for (int m = 0; m < KS; m++) {
        if (k0 + m < nz) {
          double dd;
          pin(mD[m]); pin(mC[m]); pin(pD[m]); pin(pC[m]);
          p.uh[f2 + (k0 + m) * fpl] = flux_reg(F, ru[m] + du_eval * rvr[m], rvr[m], mE[m], mD[m], mC[m], pW[m], pD[m], pC[m], dd);
        }
      }
    }
  }
  const double du = du_ph[0], du0 = du_ph[1];

  if (valid) {
    if (p.uhbt && p.u_cor && !p.set_BT_cont) {      // with BT_cont, u_cor is written in the pass of the three fits below
#pragma unroll
      for (int m = 0; m < KS; m++)
        if (k0 + m < nz) p.u_cor[f2 + (k0 + m) * fpl] = ru[m] + du * rvr[m];
    }
    if (sb == 0 && p.du_cor) p.du_cor[f2] = p.uhbt ? du : 0.0;
  }
  FC_MARK(6);
  if (!p.set_BT_cont) return;
  FaceConst F; F.dLf = fsm[PK_DLF + fl]; F.cm = fsm[PK_CM + fl]; F.cp = fsm[PK_CP + fl]; F.dt = p.dt;

  // ---- set_zonal_BT_cont :1247-1410
  const double min_visc_rem = 0.1, CFL_min = 1e-6;
  const double du_CFL = (CFL_min * I_dt) * fsm[PK_DLC + fl];
  // The duR / duL limits (:1321-1330) are chains through k as well.  The quotient a layer hands to the chain when its test
  // fires does not depend on the running limit, and some lane of the walking wave fires at nearly every layer, so a
  // division inside the chain is nk of them in a row on one wave while three wait: the quotients are formed by all
  // threads for their own layers first (planes 0 / 1 / 2 = u, visc_rem, quotient; one chain at a time, the planes
  // hold one quotient), and the chain is left with a multiply, an add, a compare and a select per layer.
  const double lim0 = min_visc_rem * fsm[CO + 4 * FC_FL + fl];
#pragma unroll 1
  for (int side = 0; side < 2; side++) {      // 0: duR (c = du_CFL, test ">"), 1: duL (c = -du_CFL, test "<")
    const double c = side ? -du_CFL : du_CFL;
#pragma unroll
    for (int m = 0; m < KS; m++) {
      if (k0 + m < nz) {
        const double vr = rvr[m], uk = ru[m];
        const double visc_rem_lim = max2(vr, lim0);
        if (side == 0) { fsm[sl + m * FC_FL] = uk; fsm[PL + sl + m * FC_FL] = vr; }
        fsm[2 * PL + sl + m * FC_FL] = (visc_rem_lim > 0.0) ? -(uk + c * vr) / visc_rem_lim : 0.0;
      }
    }
    __syncthreads();
    if (sb == 0) {
      double dlim = side ? max2(0.0, du0 - c) : min2(0.0, du0 - c);
      // branch-free steps (the tests combined with &, the sense of the comparison carried by a sign: x < y is -x > -y to the
      // bit), so that the LDS reads of eight layers are issued ahead of the dependent arithmetic
      const double mc = -c, sg = side ? -1.0 : 1.0;
#pragma unroll 8
      for (int k = 0; k < nz; k++) {
        const double uk = fsm[k * FC_FL + fl], vr = fsm[PL + k * FC_FL + fl], q = fsm[2 * PL + k * FC_FL + fl];
        const double visc_rem_lim = max2(vr, lim0);
        const double t = uk + dlim * visc_rem_lim, r = mc * vr;
        const bool fire = (visc_rem_lim > 0.0) & (sg * t > sg * r);
        dlim = fire ? q : dlim;
      }
      fsm[RO + side * FC_FL + fl] = dlim;
    }
    __syncthreads();
  }
  const double duR = fsm[RO + fl], duL = fsm[RO + FC_FL + fl];
  FC_MARK(7);
  const bool cor = p.uhbt && p.u_cor;
  // the three evaluations of every layer, each done once: the first round sums FAmt_0, FAmt_L and uhtot_L, the second
  // FAmt_R and uhtot_R (three planes)
#pragma unroll
  for (int m = 0; m < KS; m++) {
    if (k0 + m < nz) {
      const long f3 = f2 + (k0 + m) * fpl;
      const double vr = rvr[m], uk = ru[m];
      double d0, dL;
      pin(mD[m]); pin(mC[m]); pin(pD[m]); pin(pC[m]);
      (void)flux_reg(F, uk + du0 * vr, vr, mE[m], mD[m], mC[m], pW[m], pD[m], pC[m], d0);
      const double uh_L = flux_reg(F, uk + duL * vr, vr, mE[m], mD[m], mC[m], pW[m], pD[m], pC[m], dL);
      fsm[sl + m * FC_FL] = d0; fsm[PL + sl + m * FC_FL] = dL; fsm[2 * PL + sl + m * FC_FL] = uh_L;
      // u_cor (:744-748) and flux_thickness (:976-1057, with u_cor if present :809-815) ride on the same layer data
      double uc = uk;
      if (cor) { uc = uk + du * vr; if (valid) p.u_cor[f3] = uc; }
      if (p.h_face) {
        const bool pos = uc > 0.0;
        const double E = pos ? mE[m] : pW[m], Dd = pos ? mD[m] : pD[m], Cc = pos ? mC[m] : pC[m], cf = pos ? F.cm : F.cp;
        const double CFL = (fabs(uc) * p.dt) * cf;
        double h_avg = E + CFL * (0.5 * Dd + Cc * (CFL - 1.5));
        double h_marg = E + CFL * (Dd + 3.0 * Cc * (CFL - 1.0));
        if (uc == 0.0) { h_avg = 0.5 * (pW[m] + mE[m]); h_marg = 0.5 * (pW[m] + mE[m]); }
        double hu = p.o.marginal_faces ? h_marg : h_avg;
        if (p.visc_rem) hu = hu * (vr * 1.0); else hu = hu * 1.0;
        if (valid) p.h_face[f3] = hu;
      }
    }
  }
  double FAmt_0, FAmt_L, FAmt_R, uhtot_L, uhtot_R, d2;
  ksums(fsm, PL, RO, fl, sb, nz, 3, 0.0, 0.0, 0.0, FAmt_0, FAmt_L, uhtot_L);
#pragma unroll
  for (int m = 0; m < KS; m++) {
    if (k0 + m < nz) {
      double dR;
      pin(mD[m]); pin(mC[m]); pin(pD[m]); pin(pC[m]);
      const double uh_R = flux_reg(F, ru[m] + duR * rvr[m], rvr[m], mE[m], mD[m], mC[m], pW[m], pD[m], pC[m], dR);
      fsm[sl + m * FC_FL] = dR; fsm[PL + sl + m * FC_FL] = uh_R;
    }
  }
  ksums(fsm, PL, RO, fl, sb, nz, 2, 0.0, 0.0, 0.0, FAmt_R, uhtot_R, d2);
  FC_MARK(8);
  if (sb == 0 && valid) {
    double FA_0 = FAmt_0, FA_avg = FAmt_0;
    if ((duL - du0) != 0.0) FA_avg = uhtot_L / (duL - du0);
    if (FA_avg > max2(FA_0, FAmt_L)) FA_avg = max2(FA_0, FAmt_L);
    else if (FA_avg < min2(FA_0, FAmt_L)) FA_0 = FA_avg;
    p.FA_0m[f2] = FA_0; p.FA_mm[f2] = FAmt_L;
    if (fabs(FA_0 - FAmt_L) <= 1e-12 * FA_0) p.uBT_mm[f2] = 0.0;
    else p.uBT_mm[f2] = (1.5 * (duL - du0)) * ((FAmt_L - FA_avg) / (FAmt_L - FA_0));

    FA_0 = FAmt_0; FA_avg = FAmt_0;
    if ((duR - du0) != 0.0) FA_avg = uhtot_R / (duR - du0);
    if (FA_avg > max2(FA_0, FAmt_R)) FA_avg = max2(FA_0, FAmt_R);
    else if (FA_avg < min2(FA_0, FAmt_R)) FA_0 = FA_avg;
    p.FA_0p[f2] = FA_0; p.FA_pp[f2] = FAmt_R;
    if (fabs(FAmt_R - FA_0) <= 1e-12 * FA_0) p.uBT_pp[f2] = 0.0;
    else p.uBT_pp[f2] = (1.5 * (duR - du0)) * ((FAmt_R - FA_avg) / (FAmt_R - FA_0));
  }
}

// ---- mass fluxes, block-cooperative form at three or four waves a SIMD (round 5) -------------------------------------
// The same decomposition as cont_flux_coop_kernel (a half-wave = 32 face columns x one slab of KS layers, the slabs of a face
// spread over the NW waves of a block, k-ordered sums through LDS planes), with the per-layer register set cut so that three
// (6 waves x 7 layers) or four (8 waves x 5 layers) waves fit a SIMD without scratch:
//  * a cell is held as (h_L, h_R, curvature): the upwind edge E, the other edge O and the curvature C are picked by selects and
//    the edge difference of flux_layer is O - E in both of its branches (the reference's own subtraction);
//  * zonal: a lane holds ONE cell, the minus-side cell of its face; the plus-side cell is the next lane's, fetched in every
//    evaluation by a DPP wave shift (no LDS, no registers).  Meridional: both cells (the next row belongs to another block);
//  * visc_rem lives in an LDS plane for the whole kernel (the chain walkers read it there anyway), a thread reads its own slots;
//  * the Newton bracket and best error of a face are one LDS copy per block and phase (every half-wave computes the same values).
// Zonal 4 doubles a layer in registers, meridional 7 (cont_flux_coop_kernel: 8).
__device__ __forceinline__ double dpp_next_lane(double x) {      // the value of lane + 1 (0 in lane 63)
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);      // wave_shl:1
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// flux_layer :896-972 from the edge values and curvature of the two cells of a face
__device__ __forceinline__ double flux_lrc(const FaceConst &F, double u, double vr, double mL, double mR, double mC, double pL, double pR,
                                           double pC, double &duhdu) {
  const bool pos = u > 0.0;
  const double E = pos ? mR : pL, O = pos ? mL : pR, Cc = pos ? mC : pC, cf = pos ? F.cm : F.cp;
  const double Dd = O - E;
  const double CFL = (fabs(u) * F.dt) * cf;
  double uh = (F.dLf * 1.0) * u * (E + CFL * (0.5 * Dd + Cc * (CFL - 1.5)));
  double h_marg = E + CFL * (Dd + 3.0 * Cc * (CFL - 1.0));
  if (u == 0.0) { uh = 0.0; h_marg = 0.5 * (pL + mR); }
  duhdu = (F.dLf * 1.0) * h_marg * vr;
  return uh;
}

// flux_thickness :976-1057 likewise
__device__ __forceinline__ double thick_lrc(const FaceConst &F, int marginal, bool have_vr, double uc, double vr, double mL, double mR,
                                            double mC, double pL, double pR, double pC) {
  const bool pos = uc > 0.0;
  const double E = pos ? mR : pL, O = pos ? mL : pR, Cc = pos ? mC : pC, cf = pos ? F.cm : F.cp;
  const double Dd = O - E;
  const double CFL = (fabs(uc) * F.dt) * cf;
  double h_avg = E + CFL * (0.5 * Dd + Cc * (CFL - 1.5));
  double h_marg = E + CFL * (Dd + 3.0 * Cc * (CFL - 1.0));
  if (uc == 0.0) { h_avg = 0.5 * (pL + mR); h_marg = 0.5 * (pL + mR); }
  double hu = marginal ? h_marg : h_avg;
  if (have_vr) hu = hu * (vr * 1.0); else hu = hu * 1.0;
  return hu;
}

#ifndef FC3_GRP
#define FC3_GRP 2      // layers of a thread the scheduler may interleave in an evaluation pass (bounds the temporaries)
#endif
#define FC3_SCHED(m) do { if (((m) + 1) % FC3_GRP == 0) __builtin_amdgcn_sched_barrier(0); } while (0)
// FC3_UNCOND = 1: the evaluation passes run over all KS layers of a thread without a test on k (layers past nk hold zeros: u = 0 gives
// uh = +0 and a marginal thickness of 0; their rows of the planes exist and no sum reads them), one basic block the scheduler can
// interleave FC3_GRP layers in.  Measured (profiles/r05_experiments.txt): the test on k is the faster form (the slabs past nk do nothing).
#ifndef FC3_PF
#define FC3_PF 2      // layers of thicknesses a thread has in flight ahead of its reconstruction (>= KS: all of them at once)
#endif
#ifndef FC3_CHAIN_UNROLL
#define FC3_CHAIN_UNROLL 8      // LDS reads a chain walker issues ahead of its dependent arithmetic
#endif
#ifndef FC3_UNCOND
#define FC3_UNCOND 0
#endif
#define FC3_LIVE(m) (FC3_UNCOND || (k0 + (m) < nz))

__device__ __forceinline__ void ksums3(double *fsm, int plane, int roff, int fl, int sb, int nz, int nsum, double i0, double i1,
                                       double i2, double &r0, double &r1, double &r2) {
  __syncthreads();
  if (sb < nsum) {
    double acc = (sb == 0) ? i0 : ((sb == 1) ? i1 : i2);
    const int base = sb * plane + fl;
#pragma unroll 8
    for (int k = 0; k < nz; k++) acc = acc + fsm[base + k * FC_FL];
    fsm[roff + sb * FC_FL + fl] = acc;
  }
  __syncthreads();
  r0 = fsm[roff + fl];
  r1 = (nsum > 1) ? fsm[roff + FC_FL + fl] : 0.;
  r2 = (nsum > 2) ? fsm[roff + 2 * FC_FL + fl] : 0.;
}

template <int KS, int NW>
constexpr size_t fc3_lds_bytes() { return ((size_t)3 * KS * 2 * NW * FC_FL + 8 * FC_FL + 2 * NW * FC_FL + 15 * FC_FL + 6 * FC_FL) * sizeof(double); }

template <int DIR, int KS, int NW, int WPE>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void cont_flux_coop3_kernel(FluxArgs p) {
  // two work planes [KS*NS][FC_FL] of layer values and the plane of visc_rem, then results [8][FC_FL], visc_rem max [NS][FC_FL],
  // per-face values [15][FC_FL], the Newton bracket [2 phases][3][FC_FL]
#ifdef FC_TRACE
  unsigned long long fc_t0 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) atomicAdd(&fc_trace[64 * DIR + 15], 1ull);
#endif
  extern __shared__ double fsm[];
  constexpr int NS = 2 * NW;
  const m6::GridDev &g = p.g;
  const Dir<DIR> D(g);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int fl = lane & (FC_FL - 1), sb = 2 * w + (lane >> 5);      // face within the block; layer slab of this half-wave
  const int nz = g.nk;
  constexpr int FPB = (DIR == 0) ? FC_FL - 1 : FC_FL;      // faces per block (the last zonal lane only serves its neighbour)
  int bx, by;
  if (!xcd_block(p.xcd_w, p.nbx, bx, by)) return;
  const int fi_raw = p.fi0 + bx * FPB + fl;
  const bool valid = fl < FPB && fi_raw <= p.fi1 && !(p.skip && p.skip[D.f2(min(fi_raw, p.fi1), p.fj0 + by)]);
  const int fi_last = (DIR == 0) ? p.fi1 + 1 : p.fi1;
  const int fi = (fi_raw <= fi_last) ? fi_raw : fi_last;
  const int fj = p.fj0 + by;
  const long hpl = (long)g.nih * g.njh, fpl = D.fplane();
  const long s = D.sa(), fs = D.fsa();
  const long o2 = g.h2(fi, fj), f2 = D.f2(fi, fj);
  constexpr int PL = KS * NS * FC_FL;          // one plane
  constexpr int RO = 3 * PL, VO = RO + 8 * FC_FL; // results, visc_rem max
  constexpr int CO = VO + NS * FC_FL;          // per-face values parked between the phases
  constexpr int PK_DXW = CO + 6 * FC_FL, PK_DXE = CO + 7 * FC_FL, PK_MF = CO + 8 * FC_FL, PK_IAT = CO + 9 * FC_FL,
                PK_UHBT = CO + 10 * FC_FL, PK_DLC = CO + 11 * FC_FL, PK_DLF = CO + 12 * FC_FL, PK_CM = CO + 13 * FC_FL,
                PK_CP = CO + 14 * FC_FL;
  constexpr int PV = CO + 15 * FC_FL;
  const int k0 = sb * KS;
  const int sl = k0 * FC_FL + fl;                 // this thread's slot of layer k0 in a plane
  // the offset of layer m of this face, formed where it is used from an opaque copy of the 2-D offset: hoisted out of the passes, the
  // 64-bit offsets of every layer and output array are registers held for the whole kernel
  auto f3_at = [&](int m) -> long { long q = f2; asm volatile("" : "+v"(q)); return q + (long)(k0 + m) * fpl; };
#define FC3_F3(m) f3_at(m)
#define VR(m) fsm[2 * PL + sl + (m) * FC_FL]

  // ---- the thread's layers into registers (every global load issued before anything is computed from one of them)
  const bool wide = !(p.o.upwind_1st || p.o.simple_2nd);      // the 5-point stencil is only read by the PPM branch
  const long s2w = wide ? 2 * s : 0, s3w = (wide && DIR == 1) ? 3 * s : 0;
  double dLf_r = D.dL_face()[f2];
  double cm_r = (p.o.vol_CFL ? g.IareaT : D.IdL_T())[o2], cp_r = (p.o.vol_CFL ? g.IareaT : D.IdL_T())[o2 + s];
  double mk[6];                                               // mask2dT of cells -2 .. +3 along the direction
  {
    const double *mm = g.mask2dT + o2;
    mk[0] = mm[-s2w]; mk[1] = mm[-s]; mk[2] = mm[0]; mk[3] = mm[s]; mk[4] = mm[2 * s];
    mk[5] = mm[s3w];      // (only the meridional pair of reconstructions reads cell +3)
  }
  double pk_mf = D.mask_face()[f2], pk_ub = p.uhbt ? p.uhbt[f2] : 0.0;
  constexpr int KP = (DIR == 1) ? KS : 1;      // the plus-side cell is held only meridionally
  double ru[KS], sL[KS], sR[KS], sC[KS], tL[KP], tR[KP], tC[KP];
  constexpr int NH = (DIR == 0) ? 5 : 6;      // cells -2 .. +2 along the direction (+3 for the meridional pair)
  // The loads of a thread are a software pipeline over its layers (round 5): u, visc_rem, the 2-D values and the thicknesses of the
  // first FC3_PF layers are issued at once; the reconstruction of layer m then runs behind the loads of layer m + FC3_PF.  With every
  // load of the thread in flight at once (cont_flux_coop_kernel) the landing registers of KS x (NH + 2) values are the peak of the
  // kernel's register demand -- 200 VGPRs meridionally, the budget of four waves a SIMD is 128 -- and what does not fit goes to
  // scratch memory for the whole kernel.
  double hr[KS][NH];
  const char *hb[NH];
#pragma unroll
  for (int q = 0; q < NH; q++) {
    const long disp = (q == 0) ? -s2w : ((q == 5) ? s3w : (q - 2) * s);
    hb[q] = (const char *)(p.h_in + disp);
  }
  const unsigned o2b = (unsigned)(o2 * 8), hstep = (unsigned)(hpl * 8);
#define FC3_LOAD_H(m) do { \
    const unsigned k_ = (unsigned)((k0 + (m) < nz) ? k0 + (m) : nz - 1); \
    const unsigned vh_ = o2b + k_ * hstep; \
    _Pragma("unroll") for (int q = 0; q < NH; q++) hr[m][q] = *(const double *)(hb[q] + vh_); \
  } while (0)
  {
    double rvr[KS];
    {
      const char *ub = (const char *)p.u, *vb = p.visc_rem ? (const char *)p.visc_rem : ub;      // (no branch around a load)
      const unsigned f2b = (unsigned)(f2 * 8), fstep = (unsigned)(fpl * 8);
#pragma unroll
      for (int m = 0; m < KS; m++) {
        const unsigned k = (unsigned)((k0 + m < nz) ? k0 + m : nz - 1);
        const unsigned vf = f2b + k * fstep;
        ru[m] = *(const double *)(ub + vf);
        rvr[m] = *(const double *)(vb + vf);
      }
#pragma unroll
      for (int m = 0; m < KS; m++) if (m < FC3_PF) FC3_LOAD_H(m);
    }
#pragma unroll
    for (int m = 0; m < KS; m++) { pin(ru[m]); pin(rvr[m]); }
    pin(dLf_r); pin(cm_r); pin(cp_r);
    FC_MARK(0);
    double vmax_w = 0.0;
#pragma unroll
    for (int m = 0; m < KS; m++) {      // visc_rem to its plane (zero past nk, one without visc_rem); its maximum over the thread's layers
      double vr = p.visc_rem ? rvr[m] : 1.0;
      if (k0 + m >= nz) { vr = 0.0; ru[m] = 0.0; }
      VR(m) = vr;
      vmax_w = max2(vmax_w, vr);
    }
    fsm[VO + sb * FC_FL + fl] = vmax_w;
#pragma unroll
    for (int q = 0; q < 6; q++) pin(mk[q]);
    if (!wide) { mk[0] = 0.0; mk[5] = 0.0; }
    if (DIR == 0) mk[5] = 0.0;
  }
  // The face constants of flux_layer go to LDS at once (every half-wave stores the same values: no ordering between them is needed,
  // and none of them is held in registers through the reconstruction): every pass reads them back from there
  bool dead_lane;
  {
    const double dLf = dLf_r;
    const double cm = p.o.vol_CFL ? (dLf * cm_r) : cm_r;      // the CFL factor of the minus / plus side cell
    const double cp = p.o.vol_CFL ? (dLf * cp_r) : cp_r;
    pin(pk_mf); pin(pk_ub);
    fsm[PK_MF + fl] = pk_mf; fsm[PK_UHBT + fl] = pk_ub;
    fsm[PK_DLF + fl] = dLf; fsm[PK_CM + fl] = cm; fsm[PK_CP + fl] = cp;
    dead_lane = (pk_mf == 0.0) & (dLf_r == 0.0) & (pk_ub == 0.0);
  }
#define FC3_FACE(F) FaceConst F; F.dLf = fsm[PK_DLF + fl]; F.cm = fsm[PK_CM + fl]; F.cp = fsm[PK_CP + fl]; F.dt = p.dt
  // the plus-side cell of layer m: the next lane's cell (zonal; fetched outside every condition: a DPP fetch needs its source
  // lane active), or the thread's own registers (meridional)
#define FC3_PLUS(m) \
  const double pL_ = DIR ? tL[DIR ? (m) : 0] : dpp_next_lane(sL[m]), pR_ = DIR ? tR[DIR ? (m) : 0] : dpp_next_lane(sR[m]), \
               pC_ = DIR ? tC[DIR ? (m) : 0] : dpp_next_lane(sC[m])

  // ---- blocks whose 32 faces are all land: nothing to solve (see cont_flux_coop_kernel)
#pragma unroll
  for (int m = 0; m < KS; m++) dead_lane = dead_lane & (ru[m] == 0.0);      // (layers past nk hold zero)
  const bool dead = __syncthreads_and(dead_lane ? 1 : 0) != 0;
  if (dead && !p.set_BT_cont) {
    if (valid) {
#pragma unroll
      for (int m = 0; m < KS; m++) {
        if (k0 + m < nz) {
          p.uh[f2 + (k0 + m) * fpl] = 0.0;
          if (p.uhbt && p.u_cor) p.u_cor[f2 + (k0 + m) * fpl] = ru[m] + 0.0 * VR(m);
        }
      }
      if (sb == 0 && p.du_cor) p.du_cor[f2] = 0.0;
    }
    return;
  }
  // the edge values of the cells (cont_edge_kernel's arithmetic, PPM_reconstruction_x/y :2310-2662): h_L / h_R never go through memory
#pragma unroll
  for (int m = 0; m < KS; m++) {
    const int k = k0 + m;
    if (m + FC3_PF < KS) FC3_LOAD_H(m + FC3_PF);
#pragma unroll
    for (int q = 0; q < NH; q++) pin(hr[m][q]);      // (the wait for layer m leaves the later layers in flight)
    if (k < nz) {
      const double hm1 = hr[m][1], hc0 = hr[m][2], hp1 = hr[m][3], hp2 = hr[m][4];
      double Lm, Rm;
      if (DIR == 0) {
        if (p.o.upwind_1st) { Lm = hc0; Rm = hc0; }
        else edge_values(p.o, g.Angstrom_H, hr[m][0], hm1, hc0, hp1, wide ? hp2 : 0.0, mk[0], mk[1], mk[2], mk[3],
                         wide ? mk[4] : 0.0, Lm, Rm);
      } else {      // both cells at once: the slopes of the two cells serve both reconstructions
        const double h6[6] = {hr[m][0], hm1, hc0, hp1, hp2, hr[m][NH - 1]};
        double Lp, Rp;
        edge_values2(p.o, g.Angstrom_H, h6, mk, Lm, Rm, Lp, Rp);
        tL[DIR ? m : 0] = Lp; tR[DIR ? m : 0] = Rp; tC[DIR ? m : 0] = Lp + Rp - 2.0 * hp1;
      }
      sL[m] = Lm; sR[m] = Rm; sC[m] = Lm + Rm - 2.0 * hc0;
    } else {
      sL[m] = 0.; sR[m] = 0.; sC[m] = 0.;
      if (DIR) { tL[DIR ? m : 0] = 0.; tR[DIR ? m : 0] = 0.; tC[DIR ? m : 0] = 0.; }
    }
  }
  FC_MARK(1);
  if (dead) {      // (set_BT_cont: the layer data are needed for h_u / h_v and u_cor; every sum of the fits is +0)
    FC3_FACE(F0);
    const double du_dead = 0.0;
    const bool cor_d = p.uhbt && p.u_cor;
#pragma unroll
    for (int m = 0; m < KS; m++) {
      FC3_PLUS(m);
      if (k0 + m < nz) {
        const long f3 = f2 + (k0 + m) * fpl;
        const double vr = VR(m), uk = ru[m];
        if (valid) p.uh[f3] = 0.0;
        double uc = uk;
        if (cor_d) { uc = uk + du_dead * vr; if (valid) p.u_cor[f3] = uc; }
        if (p.h_face) {
          const double hu = thick_lrc(F0, p.o.marginal_faces, p.visc_rem != nullptr, uc, vr, sL[m], sR[m], sC[m], pL_, pR_, pC_);
          if (valid) p.h_face[f3] = hu;
        }
      }
    }
    if (sb == 0 && valid) {
      if (p.du_cor) p.du_cor[f2] = p.uhbt ? du_dead : 0.0;
      p.FA_0m[f2] = 0.0; p.FA_mm[f2] = 0.0; p.uBT_mm[f2] = 0.0;
      p.FA_0p[f2] = 0.0; p.FA_pp[f2] = 0.0; p.uBT_pp[f2] = 0.0;
    }
    return;
  }

  // What the phases after the first k-ordered sums read of the 2-D metrics is fetched now (the layer loads have landed and been
  // consumed: these seven values are not in flight beside them) and parked in LDS; the first evaluation pass covers their latency
  {
    double pk_aW = g.areaT[o2], pk_aE = g.areaT[o2 + s], pk_dW = D.dL_T()[o2], pk_dE = D.dL_T()[o2 + s];
    double pk_iW = g.IareaT[o2], pk_iE = g.IareaT[o2 + s], pk_lc = D.dLC_face()[f2];
    double dxw, dxe;
    if (p.o.vol_CFL) {
      const double dLf = fsm[PK_DLF + fl];
      dxw = ratio_max(pk_aW, dLf, 1000.0 * pk_dW);
      dxe = ratio_max(pk_aE, dLf, 1000.0 * pk_dE);
    } else { dxw = pk_dW; dxe = pk_dE; }
    fsm[PK_DXW + fl] = dxw; fsm[PK_DXE + fl] = dxe; fsm[PK_IAT + fl] = min2(pk_iW, pk_iE); fsm[PK_DLC + fl] = pk_lc;
  }
  // ---- layer transports and marginal areas, :622-635
  {
  FC3_FACE(F0);
#pragma unroll
  for (int m = 0; m < KS; m++) {
    FC3_PLUS(m);
    if (FC3_LIVE(m)) {
      double dd;
      const double uhk = flux_lrc(F0, ru[m], VR(m), sL[m], sR[m], sC[m], pL_, pR_, pC_, dd);
      if (valid && !p.uhbt && k0 + m < nz) p.uh[FC3_F3(m)] = uhk;      // (with uhbt, uh is stored once, after the solve)
      fsm[sl + m * FC_FL] = uhk; fsm[PL + sl + m * FC_FL] = dd;
    }
    FC3_SCHED(m);
  }
  }
  double uh_tot_0, duhdu_tot_0, dummy;
  ksums3(fsm, PL, RO, fl, sb, nz, 2, 0.0, 0.0, 0.0, uh_tot_0, duhdu_tot_0, dummy);
  double visc_rem_max = 0.0;
  for (int q = 0; q < NS; q++) visc_rem_max = max2(visc_rem_max, fsm[VO + q * FC_FL + fl]);      // (max is order-free)
  if (!(p.visc_rem && p.o.use_visc_rem_max)) visc_rem_max = 1.0;
  FC_MARK(2);

  // ---- the CFL brackets of the velocity correction, :637-716: two chains through k; what a layer hands to the chain -- the bound
  // the bracket is tested against and the value it takes when the test fails (a division) -- does not depend on the running state
  // and is formed by all threads for their own layers first (work planes 0 / 1 = bound, new bracket; visc_rem in its own plane)
  double CFL_dt = p.o.CFL_limit_adjust / p.dt;
  const double I_dt = 1.0 / p.dt;
  if (p.o.aggress_adjust) CFL_dt = I_dt;
  double I_vrm = 0.0;
  if (visc_rem_max > 0.0) I_vrm = 1.0 / visc_rem_max;
  {
    const double dx_W = fsm[PK_DXW + fl], dx_E = fsm[PK_DXE + fl], mface = fsm[PK_MF + fl];
    // (no barrier needed before the planes are rewritten: every read of them lies between the two barriers of ksums3)
#pragma unroll
    for (int m = 0; m < KS; m++) {
      if (k0 + m < nz) {
        const double uk = ru[m], vr = VR(m);
        double bound, cand;
        if (p.o.aggress_adjust) {
          bound = 0.499 * ((dx_W * I_dt - uk) + min2(0.0, p.u[f2 + (k0 + m) * fpl - fs]));
          cand = p.visc_rem ? bound / vr : bound;
        } else if (p.visc_rem) {
          bound = dx_W * CFL_dt - uk * mface; cand = (dx_W * CFL_dt - uk) / vr;
        } else {
          bound = dx_W * CFL_dt - uk; cand = bound;
        }
        fsm[sl + m * FC_FL] = bound; fsm[PL + sl + m * FC_FL] = cand;
      }
    }
    __syncthreads();
    if (sb == 0) {
      double du_max_CFL = 2.0 * (CFL_dt * dx_W) * I_vrm;
      if (p.visc_rem) {
#pragma unroll FC3_CHAIN_UNROLL
        for (int k = 0; k < nz; k++) {      // (all three loads up front: no LDS round trip inside the dependent chain)
          const double vr = fsm[2 * PL + k * FC_FL + fl], bound = fsm[k * FC_FL + fl], cand = fsm[PL + k * FC_FL + fl];
          du_max_CFL = (du_max_CFL * vr > bound) ? cand : du_max_CFL;
        }
      } else {
#pragma unroll FC3_CHAIN_UNROLL
        for (int k = 0; k < nz; k++) du_max_CFL = min2(du_max_CFL, fsm[PL + k * FC_FL + fl]);
      }
      fsm[CO + fl] = max2(du_max_CFL, 0.0);
      fsm[CO + 2 * FC_FL + fl] = uh_tot_0; fsm[CO + 3 * FC_FL + fl] = duhdu_tot_0; fsm[CO + 4 * FC_FL + fl] = visc_rem_max;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < KS; m++) {
      if (k0 + m < nz) {
        const double uk = ru[m], vr = VR(m);
        double bound, cand;
        if (p.o.aggress_adjust) {
          bound = 0.499 * ((-dx_E * I_dt - uk) + max2(0.0, p.u[f2 + (k0 + m) * fpl + fs]));
          cand = p.visc_rem ? bound / vr : bound;
        } else if (p.visc_rem) {
          bound = -dx_E * CFL_dt - uk * mface; cand = -(dx_E * CFL_dt + uk) / vr;
        } else {
          bound = -(dx_E * CFL_dt + uk); cand = bound;
        }
        fsm[sl + m * FC_FL] = bound; fsm[PL + sl + m * FC_FL] = cand;
      }
    }
    __syncthreads();
    if (sb == 0) {
      double du_min_CFL = -2.0 * (CFL_dt * dx_E) * I_vrm;
      if (p.visc_rem) {
#pragma unroll FC3_CHAIN_UNROLL
        for (int k = 0; k < nz; k++) {
          const double vr = fsm[2 * PL + k * FC_FL + fl], bound = fsm[k * FC_FL + fl], cand = fsm[PL + k * FC_FL + fl];
          du_min_CFL = (du_min_CFL * vr < bound) ? cand : du_min_CFL;
        }
      } else {
#pragma unroll FC3_CHAIN_UNROLL
        for (int k = 0; k < nz; k++) du_min_CFL = max2(du_min_CFL, fsm[PL + k * FC_FL + fl]);
      }
      fsm[CO + FC_FL + fl] = min2(du_min_CFL, 0.0);
    }
    __syncthreads();
  }
  FC_MARK(3);

  // ---- flux_adjust :1094-1243 for the faces of the block: phase 0 matches uhbt (:737-754, storing the transports),
  // phase 1 finds the correction that gives no net transport for set_*_BT_cont (:1290-1292)
  double du_ph[2] = {0.0, 0.0};
#pragma unroll 1
  for (int phase = 0; phase < 2; phase++) {
    if (phase == 0 ? (p.uhbt == nullptr) : !p.set_BT_cont) continue;
    const bool write_uh = (phase == 0);
    const double uhbt = (phase == 0) ? fsm[PK_UHBT + fl] : 0.0;
    const int max_itts = 20;
    double du = 0.0;
    double uh_err = fsm[CO + 2 * FC_FL + fl] - uhbt, duhdu_tot = fsm[CO + 3 * FC_FL + fl];
    // du_max, du_min, uh_err_best: one copy per phase (a half-wave that leaves the loop of phase 0 early must not see phase 1's)
    const int pv = PV + phase * 3 * FC_FL + fl;
    fsm[pv] = fsm[CO + fl]; fsm[pv + FC_FL] = fsm[CO + FC_FL + fl]; fsm[pv + 2 * FC_FL] = fabs(uh_err);
    double du_eval = 0.0;      // du of this face's last re-evaluation of the transports (what uh_3d holds in the reference)
    bool do_I = true, alive = valid;
#pragma unroll 1
    for (int itt = 1; itt <= max_itts; itt++) {
      bool domore = false;
      if (alive) {
        double tol_eta;
        if (itt <= 1) tol_eta = 1e-6 * p.o.tol_eta;
        else if (itt == 2) tol_eta = 1e-4 * p.o.tol_eta;
        else if (itt == 3) tol_eta = 1e-2 * p.o.tol_eta;
        else tol_eta = p.o.tol_eta;
        const double tol_vel = p.o.tol_vel;
        double du_max = fsm[pv], du_min = fsm[pv + FC_FL];
        const double uh_err_best = fsm[pv + 2 * FC_FL], IaT = fsm[PK_IAT + fl];
        if (uh_err > 0.0) { du_max = du; fsm[pv] = du; }
        else if (uh_err < 0.0) { du_min = du; fsm[pv + FC_FL] = du; }
        else do_I = false;
        if (do_I) {
          if ((p.dt * IaT * fabs(uh_err) > tol_eta) ||
              (p.o.better_iter && ((fabs(uh_err) > tol_vel * duhdu_tot) || (fabs(uh_err) > uh_err_best)))) {
            const double ddu = -uh_err / duhdu_tot;
            const double du_prev = du;
            du = du + ddu;
            if (fabs(ddu) < 1.0e-15 * fabs(du)) {
              do_I = false;
            } else if (ddu > 0.0) {
              if (du >= du_max) {
                du = 0.5 * (du_prev + du_max);
                if (du_max - du_prev < 1.0e-15 * fabs(du)) do_I = false;
              }
            } else {
              if (du <= du_min) {
                du = 0.5 * (du_prev + du_min);
                if (du_prev - du_min < 1.0e-15 * fabs(du)) do_I = false;
              }
            }
            if (do_I) domore = true;
          } else {
            do_I = false;
          }
        }
        if (!domore) alive = false;
      }
      if (!__any(alive)) break;      // the same in every wave of the block: they hold the same values
      if ((itt < max_itts) || write_uh) {
#ifdef FC_TRACE
        if (threadIdx.x == 0) atomicAdd(&fc_trace[64 * DIR + 14], 1ull);
#endif
        if (alive) du_eval = du;
        FC3_FACE(F);
#pragma unroll
        for (int m = 0; m < KS; m++) {
          pin(sC[m]);
          FC3_PLUS(m);
          if (FC3_LIVE(m)) {
            double dd;
            const double vr = VR(m);
            const double uhk = flux_lrc(F, ru[m] + du * vr, vr, sL[m], sR[m], sC[m], pL_, pR_, pC_, dd);
            fsm[sl + m * FC_FL] = uhk; fsm[PL + sl + m * FC_FL] = dd;
          }
          FC3_SCHED(m);
        }
        double usum, dsum, d2;
        ksums3(fsm, PL, RO, fl, sb, nz, 2, -uhbt, 0.0, 0.0, usum, dsum, d2);
        if (alive && itt < max_itts) {
          uh_err = usum; duhdu_tot = dsum;
          fsm[pv + 2 * FC_FL] = min2(fsm[pv + 2 * FC_FL], fabs(uh_err));
        }
      }
    }
    du_ph[phase] = du;
    FC_MARK(4 + phase);
    if (write_uh) {
      // The reference stores the layer transports on every re-evaluation (uh_3d); what remains is the last one of each
      // face, or the first evaluation for a face that never iterated (du_eval = 0: u + 0*visc_rem is u).  Stored once here.
      FC3_FACE(F);
#pragma unroll
      for (int m = 0; m < KS; m++) {
        pin(sC[m]);
        FC3_PLUS(m);
        if (FC3_LIVE(m)) {
          double dd;
          const double vr = VR(m);
          const double uhk = flux_lrc(F, ru[m] + du_eval * vr, vr, sL[m], sR[m], sC[m], pL_, pR_, pC_, dd);
          if (valid && k0 + m < nz) p.uh[FC3_F3(m)] = uhk;
        }
        FC3_SCHED(m);
      }
    }
  }
  const double du = du_ph[0], du0 = du_ph[1];

  if (valid) {
    if (p.uhbt && p.u_cor && !p.set_BT_cont) {      // with BT_cont, u_cor is written in the pass of the fits below
#pragma unroll
      for (int m = 0; m < KS; m++)
        if (k0 + m < nz) p.u_cor[FC3_F3(m)] = ru[m] + du * VR(m);
    }
    if (sb == 0 && p.du_cor) p.du_cor[f2] = p.uhbt ? du : 0.0;
  }
  FC_MARK(6);
  if (!p.set_BT_cont) return;
  FC3_FACE(F);

  // ---- set_zonal_BT_cont :1247-1410
  const double min_visc_rem = 0.1, CFL_min = 1e-6;
  const double du_CFL = (CFL_min * I_dt) * fsm[PK_DLC + fl];
  // The duR / duL limits (:1321-1330) are chains through k as well: the quotient a layer hands to the chain when its test fires
  // is formed by all threads for their own layers first (work planes 0 / 1 = u, quotient; one chain at a time)
  const double lim0 = min_visc_rem * fsm[CO + 4 * FC_FL + fl];
#pragma unroll 1
  for (int side = 0; side < 2; side++) {      // 0: duR (c = du_CFL, test ">"), 1: duL (c = -du_CFL, test "<")
    const double c = side ? -du_CFL : du_CFL;
#pragma unroll
    for (int m = 0; m < KS; m++) {
      if (k0 + m < nz) {
        const double vr = VR(m), uk = ru[m];
        const double visc_rem_lim = max2(vr, lim0);
        if (side == 0) fsm[sl + m * FC_FL] = uk;
        fsm[PL + sl + m * FC_FL] = (visc_rem_lim > 0.0) ? -(uk + c * vr) / visc_rem_lim : 0.0;
      }
    }
    __syncthreads();
    if (sb == 0) {
      double dlim = side ? max2(0.0, du0 - c) : min2(0.0, du0 - c);
      // branch-free steps (the tests combined with &, the sense of the comparison carried by a sign: x < y is -x > -y to the bit)
      const double mc = -c, sg = side ? -1.0 : 1.0;
#pragma unroll FC3_CHAIN_UNROLL
      for (int k = 0; k < nz; k++) {
        const double uk = fsm[k * FC_FL + fl], vr = fsm[2 * PL + k * FC_FL + fl], q = fsm[PL + k * FC_FL + fl];
        const double visc_rem_lim = max2(vr, lim0);
        const double t = uk + dlim * visc_rem_lim, r = mc * vr;
        const bool fire = (visc_rem_lim > 0.0) & (sg * t > sg * r);
        dlim = fire ? q : dlim;
      }
      fsm[RO + side * FC_FL + fl] = dlim;
    }
    __syncthreads();
  }
  const double duR = fsm[RO + fl], duL = fsm[RO + FC_FL + fl];
  FC_MARK(7);
  const bool cor = p.uhbt && p.u_cor;
  // The three evaluations of every layer, each done once.  First round: the one at duL (sums FAmt_L and uhtot_L), with u_cor
  // (:744-748) and flux_thickness (:976-1057, with u_cor if present :809-815) riding on the same layer data.  Second round: the
  // ones at duR and du0 (FAmt_R, uhtot_R, FAmt_0): it is the last use of visc_rem, whose plane takes the third set of values.
#pragma unroll
  for (int m = 0; m < KS; m++) {
    pin(sC[m]);
    FC3_PLUS(m);
    if (FC3_LIVE(m)) {
      const long f3 = FC3_F3(m);
      const double vr = VR(m), uk = ru[m];
      const bool st = valid && (k0 + m < nz);
      double dL;
      const double uh_L = flux_lrc(F, uk + duL * vr, vr, sL[m], sR[m], sC[m], pL_, pR_, pC_, dL);
      fsm[sl + m * FC_FL] = dL; fsm[PL + sl + m * FC_FL] = uh_L;
      double uc = uk;
      if (cor) { uc = uk + du * vr; if (st) p.u_cor[f3] = uc; }
      if (p.h_face) {
        const double hu = thick_lrc(F, p.o.marginal_faces, p.visc_rem != nullptr, uc, vr, sL[m], sR[m], sC[m], pL_, pR_, pC_);
        if (st) p.h_face[f3] = hu;
      }
    }
    FC3_SCHED(m);
  }
  double FAmt_0, FAmt_L, FAmt_R, uhtot_L, uhtot_R, d2;
  ksums3(fsm, PL, RO, fl, sb, nz, 2, 0.0, 0.0, 0.0, FAmt_L, uhtot_L, d2);
#pragma unroll
  for (int m = 0; m < KS; m++) {
    pin(sC[m]);
    FC3_PLUS(m);
    if (FC3_LIVE(m)) {
      double dR, d0;
      const double vr = VR(m), uk = ru[m];
      const double uh_R = flux_lrc(F, uk + duR * vr, vr, sL[m], sR[m], sC[m], pL_, pR_, pC_, dR);
      (void)flux_lrc(F, uk + du0 * vr, vr, sL[m], sR[m], sC[m], pL_, pR_, pC_, d0);
      fsm[sl + m * FC_FL] = dR; fsm[PL + sl + m * FC_FL] = uh_R; VR(m) = d0;
    }
    FC3_SCHED(m);
  }
  ksums3(fsm, PL, RO, fl, sb, nz, 3, 0.0, 0.0, 0.0, FAmt_R, uhtot_R, FAmt_0);
  FC_MARK(8);
  if (sb == 0 && valid) {
    double FA_0 = FAmt_0, FA_avg = FAmt_0;
    if ((duL - du0) != 0.0) FA_avg = uhtot_L / (duL - du0);
    if (FA_avg > max2(FA_0, FAmt_L)) FA_avg = max2(FA_0, FAmt_L);
    else if (FA_avg < min2(FA_0, FAmt_L)) FA_0 = FA_avg;
    p.FA_0m[f2] = FA_0; p.FA_mm[f2] = FAmt_L;
    if (fabs(FA_0 - FAmt_L) <= 1e-12 * FA_0) p.uBT_mm[f2] = 0.0;
    else p.uBT_mm[f2] = (1.5 * (duL - du0)) * ((FAmt_L - FA_avg) / (FAmt_L - FA_0));

    FA_0 = FAmt_0; FA_avg = FAmt_0;
    if ((duR - du0) != 0.0) FA_avg = uhtot_R / (duR - du0);
    if (FA_avg > max2(FA_0, FAmt_R)) FA_avg = max2(FA_0, FAmt_R);
    else if (FA_avg < min2(FA_0, FAmt_R)) FA_0 = FA_avg;
    p.FA_0p[f2] = FA_0; p.FA_pp[f2] = FAmt_R;
    if (fabs(FAmt_R - FA_0) <= 1e-12 * FA_0) p.uBT_pp[f2] = 0.0;
    else p.uBT_pp[f2] = (1.5 * (duR - du0)) * ((FAmt_R - FA_avg) / (FAmt_R - FA_0));
  }
#undef FC3_PLUS
#undef FC3_F3
#undef FC3_FACE
#undef FC3_LOAD_H
#undef VR
}

// MOM6HIP_CONT_FLUX_LANE=1 keeps the lane-per-column kernel for every call (comparison runs)
bool flux_lane_only() {
  static const int v = [] { const char *e = getenv("MOM6HIP_CONT_FLUX_LANE"); return (e && e[0] == '1') ? 1 : 0; }();
  return v != 0;
}

// Whether a flux launch takes the block-cooperative kernel (which forms the edge values itself): when there is a velocity
// correction or BT_cont to compute and the layers fit its registers; the single-pass lane-per-column kernel otherwise.
// The shape of the block-cooperative kernel for deep columns: MOM6HIP_CONT_COOP=4x10 keeps round 1's (4 waves x 10 layers a thread, two
// waves a SIMD); 6x7 and 8x5 are round 5's cont_flux_coop3_kernel at three and four waves a SIMD
// (defaults from profiles/r05_experiments.txt: zonally 8 x 5 at four waves a SIMD; meridionally the thread holds both cells of a
// face, 7 doubles a layer, and spills at every shape that fits three or four waves: round 1's kernel stays)
#ifndef FC3_DEFAULT_X
#define FC3_DEFAULT_X 85
#endif
#ifndef FC3_DEFAULT_Y
#define FC3_DEFAULT_Y 410
#endif
int flux_coop_shape(int dir) {      // 410: round 1's kernel; 67, 85: cont_flux_coop3_kernel; 853 (8x5w3), 163 (16x3), 124 (12x4), 104 (10x4): experiments of round 5
  auto parse = [](const char *e, int dflt) {
    if (!e) return dflt;
    if (strcmp(e, "4x10") == 0) return 410;
    if (strcmp(e, "8x5") == 0) return 85;
    if (strcmp(e, "8x5w3") == 0) return 853;
    if (strcmp(e, "16x3") == 0) return 163;
    if (strcmp(e, "12x4") == 0) return 124;
    if (strcmp(e, "10x4") == 0) return 104;
    return 67;
  };
  static const int v[2] = {
      [&] { const char *e = getenv("MOM6HIP_CONT_COOP_X"); if (!e) e = getenv("MOM6HIP_CONT_COOP"); return parse(e, FC3_DEFAULT_X); }(),
      [&] { const char *e = getenv("MOM6HIP_CONT_COOP_Y"); if (!e) e = getenv("MOM6HIP_CONT_COOP"); return parse(e, FC3_DEFAULT_Y); }()};
  return v[dir];
}
int flux_coop_nk_max() {
  auto cap = [](int shape) { return shape == 410 ? FC_KSMAX * FC_NS : (shape == 67 ? 84 : ((shape == 163 || shape == 124) ? 96 : 80)); };
  return std::min(cap(flux_coop_shape(0)), cap(flux_coop_shape(1)));
}

bool flux_is_coop(const FluxArgs &f) {
  // (and a 3-D array stays below 4 GB: the kernel addresses its layers with 32-bit byte offsets)
  const bool small = (size_t)(f.g.nih + 1) * (f.g.njh + 1) * f.g.nk * sizeof(double) < ((size_t)1 << 32);
  const int nk_max = flux_coop_nk_max();
  return (f.uhbt || f.set_BT_cont) && f.g.nk <= nk_max && small && !flux_lane_only() && !f.obc_on;      // (OBC: the lane kernel)
}

template <int DIR>
int launch_flux(mom6hip_ctx_t *ctx, const FluxArgs &f_in, int n_along, int n_rows) {
  const int nk = f_in.g.nk;
  dim3 grid((n_along + 63) / 64, n_rows);
  if (flux_is_coop(f_in)) {
    grid.x = (DIR == 0) ? (n_along + FC_FL - 2) / (FC_FL - 1) : (n_along + FC_FL - 1) / FC_FL;      // a zonal block yields 31 faces
    FluxArgs f = f_in;
    f.xcd_w = 0; f.nbx = (int)grid.x;
    static const int xcd_mode = getenv("MOM6HIP_CONT_XCD") ? atoi(getenv("MOM6HIP_CONT_XCD")) : 0;      // 0: launch order (the default: the strips measured slower, profiles/r05_experiments.txt section 5); 1: meridional; 3: both
    if (((DIR == 1 && xcd_mode >= 1) || (DIR == 0 && xcd_mode >= 3)) && grid.x >= 16 && n_rows >= 8) {      // (xcd_block)
      f.xcd_w = ((int)grid.x + 7) / 8;
      grid.x = 8 * f.xcd_w;
    }
    auto go = [&](auto kern, int KS) -> int {
      const size_t lds = ((size_t)3 * KS * FC_NS * FC_FL + 8 * FC_FL + FC_NS * FC_FL + 15 * FC_FL + 4 * FC_NS * FC_FL) * sizeof(double);
      std::vector<const void *> &configured = ctx->lds_configured;      // the attribute is per device: kept with the context
      if (std::find(configured.begin(), configured.end(), (const void *)kern) == configured.end()) {
        M6_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured.push_back((const void *)kern);
      }
      hipLaunchKernelGGL(kern, grid, dim3(64 * FC_NW), lds, ctx->stream, f);
      return 0;
    };
    if (nk <= FC_NS) return go(cont_flux_coop_kernel<DIR, 1>, 1);
    if (nk <= 4 * FC_NS) return go(cont_flux_coop_kernel<DIR, 4>, 4);
    auto go3 = [&](auto kern, int nw, size_t lds) -> int {
      std::vector<const void *> &configured = ctx->lds_configured;
      if (std::find(configured.begin(), configured.end(), (const void *)kern) == configured.end()) {
        M6_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured.push_back((const void *)kern);
      }
      hipLaunchKernelGGL(kern, grid, dim3(64 * nw), lds, ctx->stream, f);
      return 0;
    };
    const int shape = flux_coop_shape(DIR);
    if (shape == 67) return go3(cont_flux_coop3_kernel<DIR, 7, 6, 3>, 6, fc3_lds_bytes<7, 6>());
    if (shape == 85) return go3(cont_flux_coop3_kernel<DIR, 5, 8, 4>, 8, fc3_lds_bytes<5, 8>());
    if (shape == 853) return go3(cont_flux_coop3_kernel<DIR, 5, 8, 3>, 8, fc3_lds_bytes<5, 8>());
    if (shape == 163) return go3(cont_flux_coop3_kernel<DIR, 3, 16, 4>, 16, fc3_lds_bytes<3, 16>());
    if (shape == 124) return go3(cont_flux_coop3_kernel<DIR, 4, 12, 3>, 12, fc3_lds_bytes<4, 12>());
    if (shape == 104) return go3(cont_flux_coop3_kernel<DIR, 4, 10, 3>, 10, fc3_lds_bytes<4, 10>());
    return go(cont_flux_coop_kernel<DIR, FC_KSMAX>, FC_KSMAX);
  }
  hipLaunchKernelGGL(cont_flux_kernel<DIR>, grid, dim3(64), 0, ctx->stream, f_in);
  return 0;
}

// ---- convergence ---------------------------------------------------------------------------------
struct ConvArgs {
  m6::GridDev g;
  const double *hin, *uh;
  double *h;
  double dt, h_min;
  int i0, i1, j0, j1;
  double *h2 = nullptr;      // also stored here in the rows outside j2lo .. j2hi (the phased call: see mom6hip_continuity)
  int j2lo = 0, j2hi = -1;
};

template <int DIR>
__global__ __launch_bounds__(256) void cont_conv_kernel(ConvArgs p) {
  const m6::GridDev &g = p.g;
  const int i = p.i0 + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = p.j0 + blockIdx.y;
  const int k = blockIdx.z;
  if (i > p.i1) return;
  const long o2 = g.h2(i, j), o3 = o2 + (long)g.nih * g.njh * k;
  double uhp, uhm;
  if (DIR) {
    const long f = g.v2(i, j) + (long)g.nih * (g.njh + 1) * k;
    uhp = p.uh[f]; uhm = p.uh[f - g.nih];
  } else {
    const long f = g.u2(i, j) + (long)(g.nih + 1) * g.njh * k;
    uhp = p.uh[f]; uhm = p.uh[f - 1];
  }
  const double hn = max2(p.hin[o3] - p.dt * g.IareaT[o2] * (uhp - uhm), p.h_min);
  p.h[o3] = hn;
  if (p.h2 && (j < p.j2lo || j > p.j2hi)) p.h2[o3] = hn;
}

}  // namespace

#ifdef FC_TRACE
extern "C" int mom6hip_fc_trace(unsigned long long *out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(fc_trace), sizeof(unsigned long long) * 128) != hipSuccess) return 1;
  if (reset) { unsigned long long z[128] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(fc_trace), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#endif


extern "C" int mom6hip_continuity(mom6hip_ctx_t *ctx, const mom6hip_continuity_cs_t *cs, const double *u,
                                  const double *v, const double *hin, double *h, double *uh, double *vh, double dt,
                                  const double *uhbt, const double *vhbt, const double *visc_rem_u,
                                  const double *visc_rem_v, double *u_cor, double *v_cor,
                                  const mom6hip_bt_cont_t *BT_cont, double *du_cor, double *dv_cor, int32_t memspace) {
  return mom6hip_continuity_obc(ctx, cs, nullptr, u, v, hin, h, uh, vh, dt, uhbt, vhbt, visc_rem_u, visc_rem_v, u_cor, v_cor, BT_cont, du_cor,
                                dv_cor, memspace);
}

extern "C" int mom6hip_continuity_obc(mom6hip_ctx_t *ctx, const mom6hip_continuity_cs_t *cs, const mom6hip_obc_t *obc, const double *u,
                                      const double *v, const double *hin, double *h, double *uh, double *vh, double dt,
                                      const double *uhbt, const double *vhbt, const double *visc_rem_u,
                                      const double *visc_rem_v, double *u_cor, double *v_cor,
                                      const mom6hip_bt_cont_t *BT_cont, double *du_cor, double *dv_cor, int32_t memspace) {
  M6_REQUIRE(ctx != nullptr, "MOM_continuity_PPM: Module must be initialized before it is used.");
  M6_REQUIRE(cs && u && v && hin && h && uh && vh, "continuity_PPM: null argument");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "continuity_PPM: bad memspace");
  M6_REQUIRE((visc_rem_u != nullptr) == (visc_rem_v != nullptr),
             "MOM_continuity_PPM: Either both visc_rem_u and visc_rem_v or neither one must be present in call to "
             "continuity_PPM.");
  M6_REQUIRE(dt > 0.0, "continuity_PPM: dt must be positive");
  m6::GridDev &g = ctx->g;
  M6_REQUIRE(g.mask2dT && g.areaT && g.IareaT && g.dxT && g.dyT && g.IdxT && g.IdyT && g.dy_Cu && g.dx_Cv && g.dxCu &&
             g.dyCv && g.mask2dCu && g.mask2dCv, "continuity_PPM: a required grid metric is missing");
  int stencil = 3; if (cs->simple_2nd) stencil = 2; if (cs->upwind_1st) stencil = 1;
  const int rstencil = (cs->simple_2nd || cs->upwind_1st) ? 1 : 2;
  {
    const int need = stencil + 1 + rstencil;     // first pass: stencil rows + 1 cell + reconstruction stencil
    (void)need;
    const int halo = g.isc - g.isd;
    // PPM_reconstruction_x :2350-2361: "called with a x-halo that needs to be increased by ..."
    M6_REQUIRE(halo >= stencil && g.ied - g.iec >= stencil && g.jsc - g.jsd >= stencil && g.jed - g.jec >= stencil &&
               halo >= 1 + rstencil, "In MOM_continuity_PPM, PPM_reconstruction called with a halo that needs to be increased");
  }
  hipStream_t s = ctx->stream;
  const size_t bH = (size_t)g.nh3() * 8, bU = (size_t)g.nu3() * 8, bV = (size_t)g.nv3() * 8;
  const size_t bU2 = (size_t)(g.nih + 1) * g.njh * 8, bV2 = (size_t)g.nih * (g.njh + 1) * 8;

  m6::Stager st(ctx, memspace);
  const double *d_u = st.in(u, bU), *d_v = st.in(v, bV), *d_hin = st.in(hin, bH);
  // arrays that are only partly written keep the caller's other values: stage them in/out
  double *d_h = st.inout(h, bH);
  if (h == hin) d_hin = d_h;
  double *d_uh = st.inout(uh, bU), *d_vh = st.inout(vh, bV);
  const double *d_uhbt = st.in(uhbt, bU2), *d_vhbt = st.in(vhbt, bV2);
  const double *d_vru = st.in(visc_rem_u, bU), *d_vrv = st.in(visc_rem_v, bV);
  double *d_ucor = st.inout(u_cor, bU), *d_vcor = st.inout(v_cor, bV);
  double *d_ducor = st.out(du_cor, bU2), *d_dvcor = st.out(dv_cor, bV2);
  mom6hip_bt_cont_t bt = {};
  if (BT_cont) {
    double *const *src = &BT_cont->FA_u_W0;
    double **dst = &bt.FA_u_W0;
    for (int n = 0; n < 6; n++) { M6_REQUIRE(src[n], "continuity_PPM: BT_cont member %d is null", n); dst[n] = st.inout(src[n], bU2); }
    for (int n = 6; n < 12; n++) { M6_REQUIRE(src[n], "continuity_PPM: BT_cont member %d is null", n); dst[n] = st.inout(src[n], bV2); }
    bt.h_u = st.inout(BT_cont->h_u, bU); bt.h_v = st.inout(BT_cont->h_v, bV);
  }
  double *h_L = (double *)st.scratch(bH), *h_R = (double *)st.scratch(bH);
  if (st.failed()) return 1;
  // du_cor(:,:) = 0.0 over the whole array, :601
  if (d_ducor && ctx->cont_phase != 2) M6_HIP(hipMemsetAsync(d_ducor, 0, bU2, s));
  if (d_dvcor && ctx->cont_phase != 2) M6_HIP(hipMemsetAsync(d_dvcor, 0, bV2, s));

  ContOpts o;
  o.upwind_1st = cs->upwind_1st; o.monotonic = cs->monotonic; o.simple_2nd = cs->simple_2nd;
  o.aggress_adjust = cs->aggress_adjust; o.vol_CFL = cs->vol_CFL; o.better_iter = cs->better_iter;
  o.use_visc_rem_max = cs->use_visc_rem_max; o.marginal_faces = cs->marginal_faces;
  o.tol_eta = cs->tol_eta; o.tol_vel = cs->tol_vel; o.CFL_limit_adjust = cs->CFL_limit_adjust;

  const int is = g.isc, ie = g.iec, js = g.jsc, je = g.jec;
  const bool x_first = (ctx->host.first_direction % 2) == 0;
  const double h_min = g.Angstrom_H;

  // ---- open boundaries: what the kernels read of OBC, on the device (built once per OBC and kept with the context) ----
  struct ObcDir { int on = 0, open = 0, simple = 0, specified = 0; const int32_t *segnum = nullptr, *cell = nullptr, *fa = nullptr, *skip = nullptr; } ob[2];
  const SegDev *d_segs = nullptr;
  if (obc && obc->number_of_segments > 0) {
    M6_REQUIRE(ctx->cont_phase == 0, "continuity_PPM: open boundaries with a continuity call in two phases are not provided");
    M6_REQUIRE(obc->segment && obc->segnum_u && obc->segnum_v, "continuity_PPM: OBC%%segment, segnum_u and segnum_v are required");
    const int nseg = obc->number_of_segments;
    M6_REQUIRE(nseg <= 1024, "continuity_PPM: at most 1024 OBC segments");
    const size_t nH2 = (size_t)g.nih * g.njh, nU2 = (size_t)(g.nih + 1) * g.njh, nV2 = (size_t)g.nih * (g.njh + 1);
    // The validity of the segments and the device pointers of the specified ones' data (checked and staged at every call: the data may be
    // host arrays), then the two device tables, each built and uploaded once per OBC (m6::obc_table): the maps -- segnum_u, segnum_v, the
    // cells' reconstruction codes of either direction, the open faces' interior side -- and the segment table (keyed on the data pointers too)
    std::vector<SegDev> segs(nseg);
    const int open_d[2] = {obc->open_u_BCs_exist_globally != 0, obc->open_v_BCs_exist_globally != 0};
    uint64_t key = m6::obc_fingerprint(ctx, obc), key_segs = key;
    for (int n = 0; n < nseg; n++) {
      const mom6hip_obc_segment_t &S = obc->segment[n];
      SegDev &d = segs[n];
      d.direction = S.direction; d.open = S.open; d.specified = S.specified; d.pad = 0;
      d.IsdB = S.IsdB; d.IedB = S.IedB; d.JsdB = S.JsdB; d.JedB = S.JedB; d.isd = S.isd; d.ied = S.ied; d.jsd = S.jsd; d.jed = S.jed;
      d.normal_trans = nullptr; d.normal_vel = nullptr;
      const bool ew = S.direction == MOM6HIP_OBC_DIRECTION_E || S.direction == MOM6HIP_OBC_DIRECTION_W;
      const bool ns = S.direction == MOM6HIP_OBC_DIRECTION_N || S.direction == MOM6HIP_OBC_DIRECTION_S;
      if (!S.on_pe) continue;
      M6_REQUIRE(ew || ns, "continuity_PPM: OBC segment %d has no direction", n + 1);
      // (setup_u/v_point_obc leave a segment off the PE unless its faces lie two points inside the data domain :1448, :1588)
      M6_REQUIRE(ew ? (S.IsdB >= g.isd + 1 && S.IsdB <= g.ied - 2 && S.jsd >= g.jsd && S.jed <= g.jed)
                    : (S.JsdB >= g.jsd + 1 && S.JsdB <= g.jed - 2 && S.isd >= g.isd && S.ied <= g.ied),
                 "continuity_PPM: OBC segment %d lies outside the data domain", n + 1);
      if (S.specified) {
        M6_REQUIRE(S.normal_trans && S.normal_vel, "continuity_PPM: segment %d is specified: normal_trans and normal_vel are required", n + 1);
        const size_t cnt = ew ? (size_t)(S.IedB - S.IsdB + 1) * (S.jed - S.jsd + 1) * g.nk : (size_t)(S.ied - S.isd + 1) * (S.JedB - S.JsdB + 1) * g.nk;
        d.normal_trans = st.in(S.normal_trans, cnt * 8); d.normal_vel = st.in(S.normal_vel, cnt * 8);
      }
      key_segs = m6::obc_mix(m6::obc_mix(key_segs, (uint64_t)(uintptr_t)d.normal_trans), (uint64_t)(uintptr_t)d.normal_vel);
    }
    M6_REQUIRE(!st.failed(), "continuity_PPM: staging of the open boundaries failed");
    const size_t n_maps = 3 * nU2 + 3 * nV2 + 2 * nH2;
    const int32_t *maps = (const int32_t *)m6::obc_table(ctx, m6::OBC_SITE_CONT, key, 4 * n_maps, [&](void *host) -> int {
      int32_t *su = (int32_t *)host, *sv = su + nU2, *cx = sv + nV2, *cy = cx + nH2, *fx = cy + nH2, *fy = fx + nU2;
      int32_t *kx = fy + nV2, *ky = kx + nU2;
      memset(cx, 0, 4 * (2 * nH2 + 2 * nU2 + 2 * nV2));
      memcpy(su, obc->segnum_u, 4 * nU2); memcpy(sv, obc->segnum_v, 4 * nV2);
      int32_t *cell[2] = {cx, cy}, *fa[2] = {fx, fy}, *skip[2] = {kx, ky};
      for (int n = 0; n < nseg; n++) {
        const mom6hip_obc_segment_t &S = obc->segment[n];
        if (!S.on_pe) continue;
        const bool ew = S.direction == MOM6HIP_OBC_DIRECTION_E || S.direction == MOM6HIP_OBC_DIRECTION_W;
        const int dd = ew ? 0 : 1;
        const int A = ew ? S.IsdB : S.JsdB, c0 = ew ? S.jsd : S.isd, c1 = ew ? S.jed : S.ied;
        const bool plus = S.direction == MOM6HIP_OBC_DIRECTION_E || S.direction == MOM6HIP_OBC_DIRECTION_N;
        for (int c = c0; c <= c1; c++) {
          for (int a = A - OBC_STRIP; a <= A + OBC_STRIP; a++)      // the faces the lane kernel forms (OBC_STRIP above)
            if (a >= (ew ? g.isd - 1 : g.jsd - 1) && a <= (ew ? g.ied : g.jed)) skip[dd][ew ? g.u2(a, c) : g.v2(c, a)] = 1;
          const long ca = ew ? g.h2(A, c) : g.h2(c, A), cb = ew ? g.h2(A + 1, c) : g.h2(c, A + 1);
          if (open_d[dd]) {      // PPM_reconstruction_x/y :2385-2432: zero slopes, then the edge values (a later segment has the last word)
            cell[dd][ca] = 1 | ((plus ? 1 : 3) << 1);
            cell[dd][cb] = 1 | ((plus ? 2 : 1) << 1);
            if (S.open && (ew ? S.is_E_or_W : S.is_N_or_S)) fa[dd][ew ? g.u2(A, c) : g.v2(c, A)] = plus ? 1 : 2;      // :782-805, :1058-1088
          }
        }
      }
      return 0;
    });
    const SegDev *ds = (const SegDev *)m6::obc_table(ctx, m6::OBC_SITE_CONT, m6::obc_mix(key_segs, 0x5e95ull), sizeof(SegDev) * nseg,
                                                     [&](void *host) -> int { memcpy(host, segs.data(), sizeof(SegDev) * nseg); return 0; });
    if (!maps || !ds) return 1;
    const int32_t *dsu = maps, *dsv = dsu + nU2, *dcx = dsv + nV2, *dcy = dcx + nH2, *dfx = dcy + nH2, *dfy = dfx + nU2;
    ob[0].skip = dfy + nV2; ob[1].skip = ob[0].skip + nU2;
    d_segs = ds;
    const int pe = obc->OBC_pe != 0;
    ob[0].on = ob[1].on = 1;
    ob[0].open = open_d[0]; ob[1].open = open_d[1];
    ob[0].specified = pe && obc->specified_u_BCs_exist_globally; ob[1].specified = pe && obc->specified_v_BCs_exist_globally;
    ob[0].simple = pe && (obc->specified_u_BCs_exist_globally || obc->Flather_u_BCs_exist_globally);
    ob[1].simple = pe && (obc->specified_v_BCs_exist_globally || obc->Flather_v_BCs_exist_globally);
    ob[0].segnum = dsu; ob[1].segnum = dsv; ob[0].cell = dcx; ob[1].cell = dcy; ob[0].fa = dfx; ob[1].fa = dfy;
  }
  auto set_obc = [&](FluxArgs &f, int dd) {
    f.obc_on = ob[dd].on; f.obc_open = ob[dd].open; f.obc_simple = ob[dd].simple; f.obc_specified = ob[dd].specified;
    f.obc_dir_plus = dd ? MOM6HIP_OBC_DIRECTION_N : MOM6HIP_OBC_DIRECTION_E;
    f.segnum = ob[dd].segnum; f.fa_code = ob[dd].fa; f.segs = d_segs; f.skip = nullptr; f.xcd_w = 0; f.nbx = 0;
  };

  // With open boundaries the block-cooperative kernels run over the whole range as on a closed domain but leave the faces a segment can
  // reach alone (FluxArgs::skip); those are formed by the lane kernel with the OBC, on the side stream at the same time (a face's column
  // depends on its own inputs only, and both kernels leave the same bits where no segment reaches; the lane kernel's walk of a column
  // takes as long for a strip as for the grid, so it has to run beside the block kernel, not after it).  What a segment at face A of its direction reaches: the face itself (flux_layer :956-971, the
  // specified transports :629-634, the face areas :782-805) and, through the zeroed slopes and the copied edge values of the cells A
  // and A+1 (PPM_reconstruction :2385-2432) that enter the edge values of A-1 .. A+2, the faces A-2 .. A+2 -- along the segment's own
  // extent (its cell codes and face codes are set there only).  OBC_STRIP = 3 faces either side.
  static const bool strips_off = getenv("MOM6HIP_CONT_OBC_STRIPS") && atoi(getenv("MOM6HIP_CONT_OBC_STRIPS")) == 0;
  auto side_fork = [&]() -> int {      // the side streams wait for what the compute stream has been given so far
    if (!ctx->side_fork) {
      for (hipStream_t &st : ctx->side_stream) M6_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      M6_HIP(hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming));
      for (hipEvent_t &e : ctx->side_join) M6_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    M6_HIP(hipEventRecord(ctx->side_fork, s));
    for (hipStream_t st : ctx->side_stream) M6_HIP(hipStreamWaitEvent(st, ctx->side_fork, 0));
    return 0;
  };
  auto side_join = [&]() -> int {      // the compute stream waits for the side streams
    M6_HIP(hipGetLastError());
    for (int q = 0; q < mom6hip_ctx::NSIDE; q++) {
      M6_HIP(hipEventRecord(ctx->side_join[q], ctx->side_stream[q]));
      M6_HIP(hipStreamWaitEvent(s, ctx->side_join[q], 0));
    }
    return 0;
  };
  auto obc_strips = [&](const FluxArgs &f) -> bool {
    if (!f.obc_on || strips_off) return false;
    FluxArgs c = f; c.obc_on = 0;
    return flux_is_coop(c);
  };
  // hdst: where the thicknesses after this direction go (the output array, or the scratch of the phased call)
  auto zonal = [&](const double *hsrc, int jsh, int jeh, double hmin, double *hdst, double *also) -> int {
    if (jeh < jsh) return 0;
    FluxArgs f; f.g = g; f.o = o; f.u = d_u; f.h_in = hsrc; f.h_L = h_L; f.h_R = h_R; f.uhbt = d_uhbt; f.visc_rem = d_vru;
    f.uh = d_uh; f.u_cor = d_ucor; f.du_cor = d_ducor;
    f.FA_0m = bt.FA_u_W0; f.FA_mm = bt.FA_u_WW; f.FA_0p = bt.FA_u_E0; f.FA_pp = bt.FA_u_EE; f.uBT_mm = bt.uBT_WW;
    f.uBT_pp = bt.uBT_EE; f.h_face = bt.h_u; f.set_BT_cont = BT_cont != nullptr; f.dt = dt;
    f.fi0 = is - 1; f.fi1 = ie; f.fj0 = jsh; f.fj1 = jeh;
    set_obc(f, 0);
    const bool strips = obc_strips(f);
    if (strips) {      // the faces an E or W segment can reach: the lane kernel with the OBC, on the side stream beside the block kernel
      if (side_fork()) return 1;
      int nstrip = 0;
      for (int n = 0; n < obc->number_of_segments; n++) {
        const mom6hip_obc_segment_t &S = obc->segment[n];
        if (!S.on_pe || !(S.direction == MOM6HIP_OBC_DIRECTION_E || S.direction == MOM6HIP_OBC_DIRECTION_W)) continue;
        hipStream_t ss = ctx->side_stream[nstrip++ % mom6hip_ctx::NSIDE];      // (a strip's walk is latency: strips run beside each other)
        FluxArgs fl = f;
        fl.fi0 = std::max(is - 1, S.IsdB - OBC_STRIP); fl.fi1 = std::min(ie, S.IsdB + OBC_STRIP);
        fl.fj0 = std::max(jsh, S.jsd); fl.fj1 = std::min(jeh, S.jed);
        if (fl.fi1 < fl.fi0 || fl.fj1 < fl.fj0) continue;
        EdgeArgs e; e.g = g; e.o = o; e.h_in = hsrc; e.h_L = h_L; e.h_R = h_R; e.cell_code = ob[0].open ? ob[0].cell : nullptr;
        e.i0 = fl.fi0; e.i1 = fl.fi1 + 1; e.j0 = fl.fj0; e.j1 = fl.fj1;
        hipLaunchKernelGGL(cont_edge_kernel<0>, dim3((e.i1 - e.i0 + 256) / 256, e.j1 - e.j0 + 1, g.nk), dim3(256), 0, ss, e);
        hipLaunchKernelGGL(cont_flux_kernel<0>, dim3((fl.fi1 - fl.fi0 + 64) / 64, fl.fj1 - fl.fj0 + 1), dim3(64), 0, ss, fl);
      }
      f.obc_on = 0; f.skip = ob[0].skip;
    }
    if (!flux_is_coop(f)) {
      EdgeArgs e; e.g = g; e.o = o; e.h_in = hsrc; e.h_L = h_L; e.h_R = h_R; e.cell_code = ob[0].open ? ob[0].cell : nullptr;
      e.i0 = is - 1; e.i1 = ie + 1; e.j0 = jsh; e.j1 = jeh;
      hipLaunchKernelGGL(cont_edge_kernel<0>, dim3((e.i1 - e.i0 + 256) / 256, jeh - jsh + 1, g.nk), dim3(256), 0, s, e);
    }
    { m6::KTimer kt(ctx, MOM6HIP_KT_CONT_FLUX_X);
      if (launch_flux<0>(ctx, f, f.fi1 - f.fi0 + 1, jeh - jsh + 1)) return 1; }
    if (strips && side_join()) return 1;
    if (!x_first && ctx->cont_fluxes_only) return 0;      // (the second direction's thicknesses are not wanted: see the context)
    ConvArgs c; c.g = g; c.hin = hsrc; c.uh = d_uh; c.h = hdst; c.dt = dt; c.h_min = hmin;
    c.i0 = is; c.i1 = ie; c.j0 = jsh; c.j1 = jeh; c.h2 = also; c.j2lo = js; c.j2hi = je;
    hipLaunchKernelGGL(cont_conv_kernel<0>, dim3((ie - is + 256) / 256, jeh - jsh + 1, g.nk), dim3(256), 0, s, c);
    M6_HIP(hipGetLastError());
    return 0;
  };
  // faces fj0 .. fj1, then the cells cj0 .. cj1 (the whole call: js-1 .. je and js .. je)
  auto merid = [&](const double *hsrc, int ish, int ieh, double hmin, int fj0, int fj1, int cj0, int cj1) -> int {
    FluxArgs f; f.g = g; f.o = o; f.u = d_v; f.h_in = hsrc; f.h_L = h_L; f.h_R = h_R; f.uhbt = d_vhbt; f.visc_rem = d_vrv;
    f.uh = d_vh; f.u_cor = d_vcor; f.du_cor = d_dvcor;
    f.FA_0m = bt.FA_v_S0; f.FA_mm = bt.FA_v_SS; f.FA_0p = bt.FA_v_N0; f.FA_pp = bt.FA_v_NN; f.uBT_mm = bt.vBT_SS;
    f.uBT_pp = bt.vBT_NN; f.h_face = bt.h_v; f.set_BT_cont = BT_cont != nullptr; f.dt = dt;
    f.fi0 = ish; f.fi1 = ieh; f.fj0 = fj0; f.fj1 = fj1;
    set_obc(f, 1);
    if (fj1 >= fj0) {
      const bool strips = obc_strips(f);
      if (strips) {      // the faces a N or S segment can reach: the lane kernel with the OBC, on the side stream beside the block kernel
        if (side_fork()) return 1;
        int nstrip = 0;
        for (int n = 0; n < obc->number_of_segments; n++) {
          const mom6hip_obc_segment_t &S = obc->segment[n];
          if (!S.on_pe || !(S.direction == MOM6HIP_OBC_DIRECTION_N || S.direction == MOM6HIP_OBC_DIRECTION_S)) continue;
          hipStream_t ss = ctx->side_stream[nstrip++ % mom6hip_ctx::NSIDE];
          FluxArgs fl = f;
          fl.fi0 = std::max(ish, S.isd); fl.fi1 = std::min(ieh, S.ied);
          fl.fj0 = std::max(fj0, S.JsdB - OBC_STRIP); fl.fj1 = std::min(fj1, S.JsdB + OBC_STRIP);
          if (fl.fi1 < fl.fi0 || fl.fj1 < fl.fj0) continue;
          EdgeArgs e; e.g = g; e.o = o; e.h_in = hsrc; e.h_L = h_L; e.h_R = h_R; e.cell_code = ob[1].open ? ob[1].cell : nullptr;
          e.i0 = fl.fi0; e.i1 = fl.fi1; e.j0 = fl.fj0; e.j1 = fl.fj1 + 1;
          hipLaunchKernelGGL(cont_edge_kernel<1>, dim3((e.i1 - e.i0 + 256) / 256, (e.j1 - e.j0 + EDGE_RJ) / EDGE_RJ, g.nk), dim3(256), 0, ss, e);
          hipLaunchKernelGGL(cont_flux_kernel<1>, dim3((fl.fi1 - fl.fi0 + 64) / 64, fl.fj1 - fl.fj0 + 1), dim3(64), 0, ss, fl);
        }
        f.obc_on = 0; f.skip = ob[1].skip;
      }
      if (!flux_is_coop(f)) {
        EdgeArgs e; e.g = g; e.o = o; e.h_in = hsrc; e.h_L = h_L; e.h_R = h_R; e.cell_code = ob[1].open ? ob[1].cell : nullptr;
        e.i0 = ish; e.i1 = ieh; e.j0 = fj0; e.j1 = fj1 + 1;
        hipLaunchKernelGGL(cont_edge_kernel<1>, dim3((ieh - ish + 256) / 256, (e.j1 - e.j0 + EDGE_RJ) / EDGE_RJ, g.nk), dim3(256), 0, s, e);
      }
      { m6::KTimer kt(ctx, MOM6HIP_KT_CONT_FLUX_Y);
        if (launch_flux<1>(ctx, f, ieh - ish + 1, f.fj1 - f.fj0 + 1)) return 1; }
      if (strips && side_join()) return 1;
    }
    if (cj1 >= cj0 && !(x_first && ctx->cont_fluxes_only)) {
      ConvArgs c; c.g = g; c.hin = hsrc; c.uh = d_vh; c.h = d_h; c.dt = dt; c.h_min = hmin;
      c.i0 = ish; c.i1 = ieh; c.j0 = cj0; c.j1 = cj1;
      hipLaunchKernelGGL(cont_conv_kernel<1>, dim3((ieh - ish + 256) / 256, cj1 - cj0 + 1, g.nk), dim3(256), 0, s, c);
    }
    M6_HIP(hipGetLastError());
    return 0;
  };

  // ---- the phased call (ctx->cont_phase, set by the RK2 step around a group pass in flight; x first, device arrays) ----------
  // Phase 1, before the completion: the zonal pass of the tile's own rows (a row's fluxes read that row only), then the
  // meridional fluxes and thicknesses whose stencil stays inside those rows.  Phase 2, after it: the zonal pass of the halo rows,
  // then the faces and cells along the two edges.  The thicknesses between the two directions live in a scratch array of the
  // context, not in the output array as in the one-phase call, so no phase overwrites what the other still reads -- also when
  // the call works in place (h is hin); the halo rows of the output receive them as well, as the one-phase call leaves them.
  if (ctx->cont_phase) {
    M6_REQUIRE(x_first && memspace == MOM6HIP_MEM_DEVICE, "continuity_PPM: the phased call needs x first and device arrays");
    M6_REQUIRE(je - js + 1 >= 2 * stencil + 2, "continuity_PPM: the phased call needs a taller tile");
    M6_REQUIRE(ctx->cont_hmid.reserve(bH) == 0 && ctx->cont_hmid.p, "continuity_PPM: out of device memory");
    double *hm = (double *)ctx->cont_hmid.p;
    if (ctx->cont_phase == 1) {
      if (zonal(d_hin, js, je, 0.0, hm, nullptr)) return 1;
      if (merid(hm, is, ie, h_min, js + stencil - 1, je - stencil, js + stencil, je - stencil)) return 1;
    } else {
      if (zonal(d_hin, js - stencil, js - 1, 0.0, hm, d_h)) return 1;
      if (zonal(d_hin, je + 1, je + stencil, 0.0, hm, d_h)) return 1;
      if (merid(hm, is, ie, h_min, js - 1, js + stencil - 2, js, js + stencil - 1)) return 1;
      if (merid(hm, is, ie, h_min, je - stencil + 1, je, je - stencil + 1, je)) return 1;
    }
    return st.finish();
  }

  if (x_first) {
    if (zonal(d_hin, js - stencil, je + stencil, 0.0, d_h, nullptr)) return 1;
    if (merid(d_h, is, ie, h_min, js - 1, je, js, je)) return 1;
  } else {
    if (merid(d_hin, is - stencil, ie + stencil, 0.0, js - 1, je, js, je)) return 1;
    if (zonal(d_h, js, je, h_min, d_h, nullptr)) return 1;
  }
  return st.finish();
}
