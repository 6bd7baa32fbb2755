// ale_remap.hip -- the ALE block of step_MOM_thermo (src/core/MOM.F90:1647-1700) as gfx950 kernels: ALE_regrid for z*
// (MOM_regridding.F90:763-889), ALE_remap_set_h_vel (MOM_ALE.F90:870, :912) and the vertical remapping of tracers and velocities
// (ALE_remap_tracers :737, ALE_remap_velocities :1061 = remapping_core_h, src/ALE/MOM_remapping.F90:160-201: build_reconstructions_1d
// :257-386 + remap_via_sub_cells :463-852).
//
// Remapping has two forms, both with lanes along i so that every k-strided access is a contiguous row across the wave:
//   ale_remap_stream_kernel<NF>   PPM_H4 without boundary extrapolation on >= 6 layers (the OM4-class setting): one lane per column,
//                                 one top-down walk with a constant amount of state and no arrays (its header below);
//   ale_sub_cells_kernel + ale_remap_wave_kernel   every scheme (PCM, PLM, PPM_H4, PPM_IH4, PPM_CW, the three HYBGEN ones), with
//                                 or without extrapolation: the sub-cell structure per column, then one WAVE per column with the
//                                 column's arrays in LDS.
// Algorithmic traffic: read h_old, h_new once per launch + read/write each field = (16 / NF + 16) B per cell and field.
#include <algorithm>
#include <cfloat>
#include <cmath>

#include "common.hpp"

namespace {

constexpr int REMAP_PCM = 0, REMAP_PLM = 2, REMAP_PLM_HYBGEN = 3, REMAP_PPM_H4 = 4, REMAP_PPM_IH4 = 5, REMAP_PPM_HYBGEN = 6,
              REMAP_WENO_HYBGEN = 7, REMAP_PQM_IH4IH3 = 8, REMAP_PQM_IH6IH5 = 9, REMAP_PPM_CW = 10;   // MOM_remapping.F90:50-59
constexpr int INT_PCM = 0, INT_PLM = 1, INT_PPM = 3, INT_PQM = 5;   // :61-64

__device__ __forceinline__ double max3(double a, double b, double c) { return fmax(fmax(a, b), c); }
__device__ __forceinline__ double min3(double a, double b, double c) { return fmin(fmin(a, b), c); }
__device__ __forceinline__ double fsign(double a, double b) { return copysign(fabs(a), b); }

// ---- PLM_functions.F90 ----------------------------------------------------------------------------
// PLM_slope_wa :22-65
__device__ double plm_slope_wa(double h_l, double h_c, double h_r, double h_neglect, double u_l, double u_c, double u_r) {
  const double sigma_r = u_r - u_c;
  const double sigma_l = u_c - u_l;
  const double sigma_c = 2.0 * (u_r - u_l) * (h_c / (h_l + 2.0 * h_c + h_r + h_neglect));
  const double u_min = min3(u_l, u_c, u_r);
  const double u_max = max3(u_l, u_c, u_r);
  double slope = 0.0;
  if ((sigma_l * sigma_r) > 0.0) slope = fsign(fmin(fabs(sigma_c), 2. * fmin(u_c - u_min, u_max - u_c)), sigma_c);
  if (u_c - 0.5 * fabs(slope) < u_min || u_c + 0.5 * fabs(slope) > u_max) slope = slope * (1. - DBL_EPSILON);
  if (fabs(slope) < 1.E-140) slope = 0.;
  return slope;
}

// PLM_monotonized_slope :124-159
__device__ double plm_monotonized_slope(double u_l, double u_c, double u_r, double s_l, double s_c, double s_r) {
  const double almost_two = 2. * (1. - DBL_EPSILON);
  const double e_r = u_l + 0.5 * s_l;
  const double e_l = u_r - 0.5 * s_r;
  double slp = fabs(s_c);
  double edge = u_c - 0.5 * s_c;
  if ((edge - e_r) * (u_c - edge) < 0.) {
    edge = 0.5 * (edge + e_r);
    slp = fmin(slp, fabs(edge - u_c) * almost_two);
  }
  edge = u_c + 0.5 * s_c;
  if ((edge - u_c) * (e_l - edge) < 0.) {
    edge = 0.5 * (edge + e_l);
    slp = fmin(slp, fabs(edge - u_c) * almost_two);
  }
  return fsign(slp, s_c);
}

// PLM_extrapolate_slope :164-183
__device__ double plm_extrapolate_slope(double h_l, double h_c, double h_neglect, double u_l, double u_c) {
  const double hl = h_l + h_neglect;
  const double hc = h_c + h_neglect;
  const double left_edge = (u_l * hc + u_c * hl) / (hl + hc);
  return 2.0 * (u_c - left_edge);
}

// ---- regrid_edge_values.F90 -----------------------------------------------------------------------
// end_value_h4 :658-771
__device__ void end_value_h4(const double dz[4], const double u[4], double Csys[4]) {
  const double min_frac = 1.0e-6;
  double h1 = dz[0], h2 = dz[1], h3 = dz[2], h4 = dz[3];
  if ((h2 + h3) < min_frac * h1) h3 = min_frac * h1 - h2;
  if ((h3 + h4) < min_frac * h1) h4 = min_frac * h1 - h3;
  const double h12 = h1 + h2, h23 = h2 + h3, h34 = h3 + h4;
  const double h123 = h12 + h3, h234 = h2 + h34, h1234 = h12 + h34;
  const double I_denB3 = 1.0 / (h123 * h12 * h23);
  const double I_h12 = (h123 * h23) * I_denB3;
  const double I_h23 = (h12 * h123) * I_denB3;
  const double I_h123 = (h12 * h23) * I_denB3;
  const double I_denom = 1.0 / (h1234 * (h234 * h34));
  const double I_h234 = (h1234 * h34) * I_denom;
  const double I_h1234 = (h234 * h34) * I_denom;

  const double W11 = -h1 * (I_h1234 + I_h123 + I_h12);
  const double W21 = h1 * h12 * (I_h234 * I_h1234 + I_h23 * (I_h234 + I_h123));
  const double W31 = -h1 * h12 * h123 * I_denom;
  const double W12 = 2.0 * (I_h12 * (1.0 + (h1 + h12) * (I_h1234 + I_h123)) + h1 * I_h1234 * I_h123);
  const double W22 = -2.0 * ((h1 * h12 * I_h1234) * (I_h23 * (I_h234 + I_h123)) +
                             (h1 + h12) * (I_h1234 * I_h234 + I_h23 * (I_h234 + I_h123)));
  const double W32 = 2.0 * ((h1 + h12) * h123 + h1 * h12) * I_denom;
  const double W13 = -3.0 * I_h12 * I_h123 * (1.0 + I_h1234 * ((h1 + h12) + h123));
  const double W23 = 3.0 * I_h23 * (I_h123 + I_h1234 * ((h1 + h12) + h123) * (I_h123 + I_h234));
  const double W33 = -3.0 * ((h1 + h12) + h123) * I_denom;
  const double W14 = 4.0 * I_h1234 * I_h123 * I_h12;
  const double W24 = -4.0 * I_h1234 * (I_h23 * (I_h123 + I_h234));
  const double W34 = 4.0 * I_denom;

  Csys[0] = ((u[0] + W11 * (u[1] - u[0])) + W21 * (u[2] - u[1])) + W31 * (u[3] - u[2]);
  Csys[1] = (W12 * (u[1] - u[0]) + W22 * (u[2] - u[1])) + W32 * (u[3] - u[2]);
  Csys[2] = (W13 * (u[1] - u[0]) + W23 * (u[2] - u[1])) + W33 * (u[3] - u[2]);
  Csys[3] = (W14 * (u[1] - u[0]) + W24 * (u[2] - u[1])) + W34 * (u[3] - u[2]);
}

// ---- ALE_regrid, z* (MOM_regridding.F90:763-889, :1174-1284; coord_zlike.F90:63-144) ------------------------------
struct RegridArgs {
  m6::GridDev g;
  const double *h;
  double *h_new, *dz;
  const double *res;       // coordinateResolution on the device (nk values)
  double min_thickness, old_grid_weight, zs, zd, Z_ref;
};

template <int NK>
__global__ __launch_bounds__(64) void ale_regrid_zstar_kernel(RegridArgs a) {
  const m6::GridDev &g = a.g;
  const int i = g.isc - 1 + blockIdx.x * 64 + threadIdx.x;
  const int j = g.jsc - 1 + blockIdx.y;
  if (i > g.iec + 1) return;
  const long n2 = g.h2(i, j), hstr = (long)g.nih * g.njh;
  const int nz = g.nk;
  if (g.mask2dT[n2] == 0.) {        // :1213 ; calc_h_new_by_dz keeps the old thicknesses on land (:951)
    for (int k = 0; k <= nz; k++) a.dz[n2 + hstr * k] = 0.;
    for (int k = 0; k < nz; k++) a.h_new[n2 + hstr * k] = a.h[n2 + hstr * k];
    return;
  }
  // 1-based columns as in the reference
  double hc[NK + 2], zOld[NK + 2], zNew[NK + 2], dz[NK + 2];
  const double Z_to_H = g.Z_to_H;
  const double depth = m6::max2((g.bathyT[n2] + a.Z_ref) * Z_to_H, 0.0);                   // :833
  double total = 0.0;
  for (int k = 1; k <= nz; k++) { hc[k] = a.h[n2 + hstr * (k - 1)]; total = total + hc[k]; }
  zOld[nz + 1] = -depth;
  for (int k = nz; k >= 1; k--) zOld[k] = zOld[k + 1] + hc[k];
  // build_zstar_column, no rigid top
  {
    const double min_thickness = m6::min2(a.min_thickness, total / (double)nz);
    const double eta = total - depth;
    const double stretching = total / (depth + 0.);
    zNew[1] = eta;
    for (int k = 1; k <= nz; k++) {
      const double dh = stretching * a.res[k - 1] * Z_to_H;
      zNew[k + 1] = zNew[k] - dh;
    }
    zNew[nz + 1] = -depth;
    for (int k = nz; k >= 1; k--)
      if (zNew[k] < (zNew[k + 1] + min_thickness)) zNew[k] = zNew[k + 1] + min_thickness;
  }
  // filtered_grid_motion :1022
  {
    double sgn;
    const double prod = (zOld[nz + 1] - zOld[1]) * (zNew[nz + 1] - zNew[1]);
    bool done = false;
    if (prod == 0.0) { for (int k = 1; k <= nz + 1; k++) dz[k] = 0.0; done = true; }
    else if ((zOld[nz + 1] - zOld[1]) + (zNew[nz + 1] - zNew[1]) > 0.0) sgn = 1.0;
    else sgn = -1.0;
    if (!done) {
      const double zs = a.zs, zd = a.zd;
      const double wtd = 1.0 - a.old_grid_weight, Iwtd = 1.0 / wtd;
      const double dzwt = (zd - zs);
      double Idzwt = 0.0; if (fabs(zd - zs) > 0.0) Idzwt = 1.0 / (zd - zs);
      const double dInt_zs_zd = 0.5 * (1.0 + Iwtd) * (zd - zs);
      const double Aq = 0.5 * (Iwtd - 1.0);
      dz[1] = 0.0;
      for (int k = 2; k <= nz + 1; k++) {
        const double z_old_k = zOld[k];
        const double dz_tgt = sgn * (zNew[k] - z_old_k);
        const double zr1 = sgn * (z_old_k - zOld[1]);
        double r;
        if ((zr1 > zd) && (zr1 + wtd * dz_tgt > zd)) {
          r = sgn * wtd * dz_tgt;
        } else if ((zr1 < zs) && (zr1 + dz_tgt < zs)) {
          r = sgn * dz_tgt;
        } else {
          double Int_zd, Int_zs;
          if (zr1 >= zd) { Int_zd = Iwtd * (zd - zr1); Int_zs = Int_zd - dInt_zs_zd; }
          else if (zr1 <= zs) { Int_zs = (zs - zr1); Int_zd = dInt_zs_zd + (zs - zr1); }
          else {
            Int_zd = (zd - zr1) * (Iwtd * (0.5 * (zd + zr1) - zs) + 0.5 * (zd - zr1)) * Idzwt;
            Int_zs = (zs - zr1) * (0.5 * Iwtd * ((zr1 - zs)) + (zd - 0.5 * (zr1 + zs))) * Idzwt;
          }
          if (dz_tgt >= Int_zd) r = sgn * ((zd - zr1) + wtd * (dz_tgt - Int_zd));
          else if (dz_tgt <= Int_zs) r = sgn * ((zs - zr1) + (dz_tgt - Int_zs));
          else {
            double dz0, z0, F0;
            if (zr1 <= zs) { dz0 = zs - zr1; z0 = zs; F0 = dz_tgt - Int_zs; }
            else if (zr1 >= zd) { dz0 = zd - zr1; z0 = zd; F0 = dz_tgt - Int_zd; }
            else { dz0 = 0.0; z0 = zr1; F0 = dz_tgt; }
            const double Bq = (dzwt + 2.0 * Aq * (z0 - zs));
            r = sgn * (dz0 + 2.0 * F0 * dzwt / (Bq + sqrt(Bq * Bq + 4.0 * Aq * F0 * dzwt)));
          }
        }
        dz[k] = r;
      }
    }
  }
  // adjust_interface_motion :1754-1772 (the FATAL checks are not evaluated on the device)
  {
    const double eps = 2.220446049250313e-16;
    for (int k = nz; k >= 2; k--) {
      double hn = hc[k] + (dz[k] - dz[k + 1]);
      if (hn < a.min_thickness) dz[k] = (dz[k + 1] - hc[k]) + a.min_thickness;
      hn = hc[k] + (dz[k] - dz[k + 1]);
      if (hn < 0.) dz[k] = (1. - eps) * (dz[k + 1] - hc[k]);
    }
  }
  for (int k = 1; k <= nz + 1; k++) a.dz[n2 + hstr * (k - 1)] = dz[k];
  for (int k = 1; k <= nz; k++) a.h_new[n2 + hstr * (k - 1)] = m6::max2(0., hc[k] + (dz[k] - dz[k + 1]));   // calc_h_new_by_dz :943
}

// ALE_remap_set_h_vel :892-899: one thread per (I/i, j/J, k) of the union of the two face ranges
__global__ __launch_bounds__(64) void ale_set_h_vel_kernel(m6::GridDev g, const double *__restrict__ h, double *__restrict__ h_u,
                                                           double *__restrict__ h_v) {
  const int i = g.isc - 1 + blockIdx.x * 64 + threadIdx.x, j = g.jsc - 1 + blockIdx.y, k = blockIdx.z;
  if (i > g.iec) return;
  if (j >= g.jsc && g.mask2dCu[g.u2(i, j)] > 0.) h_u[g.u3(i, j, k)] = 0.5 * (h[g.h3(i, j, k)] + h[g.h3(i + 1, j, k)]);
  if (i >= g.isc && g.mask2dCv[g.v2(i, j)] > 0.) h_v[g.v3(i, j, k)] = 0.5 * (h[g.h3(i, j, k)] + h[g.h3(i, j + 1, k)]);
}

// ALE_remap_set_h_vel_via_dz :938-949 (REMAP_UV_USING_OLD_ALG): the same thread map; dzi has nk+1 planes
__global__ __launch_bounds__(64) void ale_set_h_vel_via_dz_kernel(m6::GridDev g, const double *__restrict__ h, const double *__restrict__ dzi,
                                                                  double *__restrict__ h_u, double *__restrict__ h_v) {
  const int i = g.isc - 1 + blockIdx.x * 64 + threadIdx.x, j = g.jsc - 1 + blockIdx.y, k = blockIdx.z;
  if (i > g.iec) return;
  const long pl = (long)g.nih * g.njh, c = g.h2(i, j) + pl * k;
  if (j >= g.jsc && g.mask2dCu[g.u2(i, j)] > 0.)
    h_u[g.u3(i, j, k)] = m6::max2(0., 0.5 * (h[c] + h[c + 1]) + 0.5 * ((dzi[c] + dzi[c + 1]) - (dzi[c + pl] + dzi[c + pl + 1])));
  if (i >= g.isc && g.mask2dCv[g.v2(i, j)] > 0.)
    h_v[g.v3(i, j, k)] = m6::max2(0., 0.5 * (h[c] + h[c + g.nih]) + 0.5 * ((dzi[c] + dzi[c + g.nih]) - (dzi[c + pl] + dzi[c + pl + g.nih])));
}

// ======================================================================================================================
// Wave-cooperative remapping: one 64-lane wave per column, every per-column array in LDS.
// The lane-per-column kernels above keep ~12 KB of per-lane arrays in scratch and are bound by that traffic (96 GB per
// 4-tracer call on the 1/4-degree grid).  Here the work is split by its shape:
//   ale_sub_cells_kernel   lane per column: the merge of the old and new grids into sub-cells (build_sub_cells) is a
//                          serial walk with no arrays of its own; it reads h_old / h_new and writes the structure
//                          (h_sub, h0_eff, the end / thickest sub-cell of every source cell, the end of every target
//                          cell) to a global buffer laid out in tiles of WR_NCOL columns.
//   ale_remap_wave_kernel  a block of WR_NCOL waves owns WR_NCOL consecutive columns: it loads their structure tile
//                          and layers with contiguous rows into LDS, then each wave works on its own column with the
//                          arithmetic of the functions above, unchanged -- the k loops of the reconstructions run across
//                          the lanes (they are stencils), and the three passes of remap_via_sub_cells that sum over
//                          sub-cells run one lane per source / target cell, each lane adding its few sub-cells in the
//                          reference's order.  Results are bit-identical to the serial form.
// ======================================================================================================================
constexpr int WR_NCOL = 8;

// Structure tile of WR_NCOL columns in global memory: rows of WR_NCOL entries.
//   doubles: h_sub[0 .. 2nz+1], h0_eff[0 .. nz]           shorts: isrc_end[0 .. nz], isrc_max[0 .. nz], itgt_end[0 .. nz], last_thick
// (1-based rows as in the reference; itgt_end(i1) is stored negated where h1(i1) <= 0.)
__host__ __device__ inline int sub_drows(int nz) { return (2 * nz + 2) + (nz + 1); }
__host__ __device__ inline int sub_srows(int nz) { return 3 * (nz + 1) + 1; }
__host__ __device__ inline size_t sub_tile_bytes(int nz) { return (size_t)sub_drows(nz) * WR_NCOL * 8 + (size_t)sub_srows(nz) * WR_NCOL * 2; }

struct SubArgs {
  m6::GridDev g;
  const double *h_old, *h_new;
  char *sub;        // structure tiles
  int pos, nbx;     // staggering; tiles per row
};

// build_sub_cells (MOM_remapping.F90:519-651), one lane per column, 1-based indices kept
__global__ __launch_bounds__(64) void ale_sub_cells_kernel(SubArgs a) {
  const m6::GridDev &g = a.g;
  const int nz = g.nk, n0 = nz, n1 = nz;
  const int xs = (a.pos == MOM6HIP_POS_U) ? 1 : 0, ys = (a.pos == MOM6HIP_POS_V) ? 1 : 0;
  const int ci = blockIdx.x * 64 + threadIdx.x;      // column within the row of the compute range
  const int i = g.isc - xs + ci, j = g.jsc - ys + blockIdx.y;
  if (i > g.iec) return;
  const long n2 = a.pos == MOM6HIP_POS_U ? g.u2(i, j) : (a.pos == MOM6HIP_POS_V ? g.v2(i, j) : g.h2(i, j));
  const double msk = a.pos == MOM6HIP_POS_U ? g.mask2dCu[n2] : (a.pos == MOM6HIP_POS_V ? g.mask2dCv[n2] : g.mask2dT[n2]);
  if (!(msk > 0.)) return;
  const long stride = (long)(g.nih + xs) * (g.njh + ys);
  const double *h0 = a.h_old + n2, *h1 = a.h_new + n2;
  char *tile = a.sub + sub_tile_bytes(nz) * ((size_t)blockIdx.y * a.nbx + ci / WR_NCOL);
  const int cc = ci % WR_NCOL;
  double *h_sub = (double *)tile + cc, *h0_eff = h_sub + (size_t)(2 * nz + 2) * WR_NCOL;
  short *isrc_end = (short *)((double *)tile + (size_t)sub_drows(nz) * WR_NCOL) + cc;
  short *isrc_max = isrc_end + (nz + 1) * WR_NCOL, *itgt_end = isrc_max + (nz + 1) * WR_NCOL, *meta = itgt_end + (nz + 1) * WR_NCOL;

  int i0_last_thick_cell = 0;
  for (int i0 = 1; i0 <= n0; i0++) if (h0[(i0 - 1) * stride] > 0.) i0_last_thick_cell = i0;
  meta[0] = (short)i0_last_thick_cell;
  double h0_supply = h0[0], h1_supply = h1[0];
  double h1_cur = h1_supply;      // h1 of the open target cell
  bool src_has_volume = true, tgt_has_volume = true;
  int i0 = 1, i1 = 1, i_max = 1;
  double dh_max = 0., dh0_eff = 0.;
  h_sub[1 * WR_NCOL] = 0.;
  const int ns = n0 + n1 + 1;
  for (int i_sub = 2; i_sub <= ns; i_sub++) {
    const double dh = fmin(h0_supply, h1_supply);
    dh0_eff = dh0_eff + fmin(dh, h0_supply);
    double hs = dh;
    if (dh >= dh_max) { i_max = i_sub; dh_max = dh; }
    bool end_src = false, end_tgt = false;
    if (h0_supply <= h1_supply && src_has_volume) {
      h1_supply = h1_supply - dh;
      end_src = true;
    } else if (h0_supply >= h1_supply && tgt_has_volume) {
      h0_supply = h0_supply - dh;
      end_tgt = true;
    } else if (src_has_volume) {
      hs = h0_supply;
      end_src = true;
    } else if (tgt_has_volume) {
      hs = h1_supply;
      end_tgt = true;
    }
    h_sub[(size_t)i_sub * WR_NCOL] = hs;
    if (end_src) {
      isrc_end[i0 * WR_NCOL] = (short)i_sub;
      isrc_max[i0 * WR_NCOL] = (short)i_max; i_max = i_sub + 1; dh_max = 0.;
      h0_eff[(size_t)i0 * WR_NCOL] = dh0_eff;
      if (i0 < n0) { i0 = i0 + 1; h0_supply = h0[(i0 - 1) * stride]; dh0_eff = 0.; }
      else { h0_supply = 0.; src_has_volume = false; }
    } else if (end_tgt) {
      itgt_end[i1 * WR_NCOL] = (short)((h1_cur > 0.) ? i_sub : -i_sub);
      if (i1 < n1) { i1 = i1 + 1; h1_supply = h1[(i1 - 1) * stride]; h1_cur = h1_supply; }
      else { h1_supply = 0.; tgt_has_volume = false; }
    }
  }
}

struct WCol {   // LDS arrays of one column (doubles first, then shorts); the first block mirrors the structure tile's rows
  double *h_sub, *h0_eff, *h0, *u0, *EL, *ER, *C1, *u_sub, *uh_sub;
  double *SR;                              // PQM only (xd = nz doubles more a column): the right edge slopes; the left ones live in C1
  short *isrc_end, *isrc_max, *itgt_end;   // (+ last_thick at itgt_end[nz + 1])
};
__host__ __device__ inline int wcol_extra(int scheme, int nz) { return (scheme == REMAP_PQM_IH4IH3 || scheme == REMAP_PQM_IH6IH5) ? nz : 0; }
__host__ __device__ inline size_t wcol_doubles(int nz, int xd) { return (size_t)sub_drows(nz) + 5 * nz + 2 * (2 * nz + 2) + xd; }
__host__ __device__ inline size_t wcol_bytes(int nz, int xd) { return (wcol_doubles(nz, xd) * 8 + (size_t)sub_srows(nz) * 2 + 7) / 8 * 8; }
__device__ inline WCol wcol_at(char *base, int nz, int xd) {
  WCol c;
  double *d = (double *)base;
  c.h_sub = d; d += 2 * nz + 2; c.h0_eff = d; d += nz + 1;
  c.h0 = d; d += nz; c.u0 = d; d += nz; c.EL = d; d += nz; c.ER = d; d += nz; c.C1 = d; d += nz;
  c.u_sub = d; d += 2 * nz + 2; c.uh_sub = d; d += 2 * nz + 2;
  c.SR = d; d += xd;
  short *sh = (short *)d;
  c.isrc_end = sh; sh += nz + 1; c.isrc_max = sh; sh += nz + 1; c.itgt_end = sh;
  return c;
}
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// The quartic of a cell from its mean, width, edge values and edge slopes (PQM_functions.F90:51-55); coefficient a is the left edge value
struct Quartic { double b, c, d, e; };
__device__ __forceinline__ Quartic pqm_quartic(double um, double hc, double u0_l, double u0_r, double u1_l, double u1_r) {
  Quartic q;
  q.b = hc * u1_l;
  q.c = 30.0 * um - 12.0 * u0_r - 18.0 * u0_l + 1.5 * hc * (u1_r - 3.0 * u1_l);
  q.d = -60.0 * um + hc * (6.0 * u1_l - 4.0 * u1_r) + 28.0 * u0_r + 32.0 * u0_l;
  q.e = 30.0 * um + 2.5 * hc * (u1_r - u1_l) - 15.0 * (u0_l + u0_r);
  return q;
}
// 4.0 * e * (x**3) + 3.0 * d * (x**2) + 2.0 * c * x + b with the Fortran's association
__device__ __forceinline__ double pqm_gradient(const Quartic &q, double x) {
  return 4.0 * q.e * ((x * x) * x) + 3.0 * q.d * (x * x) + 2.0 * q.c * x + q.b;
}
// Does the quartic have an inflexion point inside the cell where its gradient opposes `slope`?  PQM_limiter :161-237 tests the closed
// interval with a chain of cases (closed = true), PQM_boundary_extrapolation_v1 :597-632 the open one for each root on its own
__device__ bool pqm_bad_inflexion(const Quartic &q, double slope, bool closed) {
  const double alpha1 = 6 * q.e, alpha2 = 3 * q.d, alpha3 = q.c;
  const double rho = alpha2 * alpha2 - 4.0 * alpha1 * alpha3;
  bool bad = false;
  if ((alpha1 != 0.0) && (rho >= 0.0)) {
    const double sqrt_rho = sqrt(rho);
    const double x1 = 0.5 * (-alpha2 - sqrt_rho) / alpha1;
    const double x2 = 0.5 * (-alpha2 + sqrt_rho) / alpha1;
    const bool in1 = closed ? ((x1 >= 0.0) && (x1 <= 1.0)) : ((x1 > 0.0) && (x1 < 1.0));
    const bool in2 = closed ? ((x2 >= 0.0) && (x2 <= 1.0)) : ((x2 > 0.0) && (x2 < 1.0));
    if (in1 && (pqm_gradient(q, x1) * slope < 0.0)) bad = true;
    if (in2 && (pqm_gradient(q, x2) * slope < 0.0)) bad = true;
  }
  if ((alpha1 == 0.0) && (alpha2 != 0.0)) {
    const double x1 = -alpha3 / alpha2;
    if ((x1 >= 0.0) && (x1 <= 1.0)) {
      // (the limiter keeps the cubic term, 4 e x^3 with e = 0; the boundary routine drops it: e = alpha1 / 6 = 0 either way, and
      // 4.0 * 0.0 * x^3 + y = y for finite x)
      if (pqm_gradient(q, x1) * slope < 0.0) bad = true;
    }
  }
  return bad;
}

// ---- PQM_IH6IH5: the 6x6 systems behind edge_values_implicit_h6 and edge_slopes_implicit_h5 (regrid_edge_values.F90:977-1454)
// linear_solver (regrid_solvers.F90:115-176): Gaussian elimination with the first nonzero pivot, A[row][col].  A singular system (a
// FATAL error in the reference) leaves NaNs.
__device__ void linear_solver6(double A[6][6], double R[6], double X[6]) {
  const int N = 6;
  bool singular = false;
  for (int i = 0; i < N - 1; i++) {
    int k = i;
    while (k < N && !(fabs(A[k][i]) > 0.0)) k++;
    if (k >= N) { singular = true; break; }
    if (k != i) {
      for (int j = i; j < N; j++) { const double swap = A[i][j]; A[i][j] = A[k][j]; A[k][j] = swap; }
      const double swap = R[i]; R[i] = R[k]; R[k] = swap;
    }
    const double I_pivot = 1.0 / A[i][i];
    A[i][i] = 1.0;
    for (int j = i + 1; j < N; j++) A[i][j] = A[i][j] * I_pivot;
    R[i] = R[i] * I_pivot;
    for (int kk = i + 1; kk < N; kk++) {
      const double factor = A[kk][i];
      for (int j = i + 1; j < N; j++) A[kk][j] = A[kk][j] - factor * A[i][j];
      R[kk] = R[kk] - factor * R[i];
    }
  }
  if (singular || A[N - 1][N - 1] == 0.0) {
    for (int i = 0; i < N; i++) X[i] = __builtin_nan("");
    return;
  }
  X[N - 1] = R[N - 1] / A[N - 1][N - 1];
  for (int i = N - 2; i >= 0; i--) {
    X[i] = R[i];
    for (int j = i + 1; j < N; j++) X[i] = X[i] - A[i][j] * X[j];
  }
}
// ((a+b)^(m+2) - a^(m+2)) / b, m = 0..4, as the reference writes them (a the inner cell, b the outer one)
__device__ void ih_polys(double a, double b, double P[5]) {
  const double a_2 = a * a, a_3 = a_2 * a, a_4 = a_2 * a_2, a_5 = a_3 * a_2;
  P[0] = (2.0 * a + b);
  P[1] = (3.0 * a_2 + b * (3.0 * a + b));
  P[2] = (4.0 * a_3 + b * (6.0 * a_2 + b * (4.0 * a + b)));
  P[3] = (5.0 * a_4 + b * (10.0 * a_3 + b * (10.0 * a_2 + b * (5.0 * a + b))));
  P[4] = (6.0 * a_5 + b * (15.0 * a_4 + b * (20.0 * a_3 + b * (15.0 * a_2 + b * (6.0 * a + b)))));
}
// a row of the boundary systems :1169-1171, :1367-1369
__device__ void ih_boundary_row(double xavg, double dx, double row[6]) {
  const double C1_12 = 1.0 / 12.0, C5_6 = 5.0 / 6.0;
  const double x2 = xavg * xavg, x4 = x2 * x2, d2 = dx * dx, d4 = d2 * d2;
  row[0] = 1.0; row[1] = xavg; row[2] = (x2 + C1_12 * d2); row[3] = xavg * (x2 + 0.25 * d2);
  row[4] = (x4 + 0.5 * x2 * d2 + 0.0125 * d4);
  row[5] = xavg * (x4 + C5_6 * x2 * d2 + 0.0625 * d4);
}
// The coefficients of one row of the tridiagonal system (alpha, beta and the four weights of the cell means).  SLOPES: the system of
// edge_slopes_implicit_h5, else of edge_values_implicit_h6; which: 0 the centred stencil, 1 the right-biased one (second row), 2 the
// left-biased one (second to last row)
template <bool SLOPES>
__device__ void ih_coefs(double h0, double h1, double h2, double h3, int which, double C[6]) {
  double A[6][6], B[6], Pl[5], Pr[5];
  const double h1_2 = h1 * h1, h1_3 = h1_2 * h1, h1_4 = h1_2 * h1_2, h1_5 = h1_3 * h1_2;
  const double h2_2 = h2 * h2, h2_3 = h2_2 * h2, h2_4 = h2_2 * h2_2, h2_5 = h2_3 * h2_2;
  ih_polys(h1, h0, Pl); ih_polys(h2, h3, Pr);
  if (SLOPES) {
    const double r3[6] = {1.0, Pl[0], Pl[1], -Pl[2], Pl[3], -Pl[4]};
    const double r4[6] = {1.0, h1, h1_2, -h1_3, h1_4, -h1_5};
    const double r5[6] = {1.0, -h2, h2_2, h2_3, h2_4, h2_5};
    const double r6[6] = {1.0, -Pr[0], Pr[1], Pr[2], Pr[3], Pr[4]};
    for (int n = 0; n < 6; n++) { A[n][2] = r3[n]; A[n][3] = r4[n]; A[n][4] = r5[n]; A[n][5] = r6[n]; }
    A[0][0] = 0.0; A[0][1] = 0.0; A[1][0] = 2.0; A[1][1] = 2.0;
    if (which == 0) {
      A[2][0] = 6.0 * h1;      A[2][1] = -6.0 * h2;
      A[3][0] = -12.0 * h1_2;  A[3][1] = -12.0 * h2_2;
      A[4][0] = 20.0 * h1_3;   A[4][1] = -20.0 * h2_3;
      A[5][0] = -30.0 * h1_4;  A[5][1] = -30.0 * h2_4;
      B[0] = 0.0; B[1] = -2.0; B[2] = 0.0; B[3] = 0.0; B[4] = 0.0; B[5] = 0.0;
    } else if (which == 1) {
      const double h01 = h0 + h1, h01_2 = h01 * h01;
      A[2][0] = 6.0 * h01;                A[2][1] = 0.0;
      A[3][0] = -12.0 * h01_2;            A[3][1] = 0.0;
      A[4][0] = 20.0 * (h01 * h01_2);     A[4][1] = 0.0;
      A[5][0] = -30.0 * (h01_2 * h01_2);  A[5][1] = 0.0;
      B[0] = 0.0; B[1] = -2.0; B[2] = -6.0 * h1; B[3] = 12.0 * h1_2; B[4] = -20.0 * h1_3; B[5] = 30.0 * h1_4;
    } else {
      const double h23 = h2 + h3, h23_2 = h23 * h23;
      A[2][0] = 0.0; A[2][1] = -6.0 * h23;
      A[3][0] = 0.0; A[3][1] = -12.0 * h23_2;
      A[4][0] = 0.0; A[4][1] = -20.0 * (h23 * h23_2);
      A[5][0] = 0.0; A[5][1] = -30.0 * (h23_2 * h23_2);
      B[0] = 0.0; B[1] = -2.0; B[2] = 6.0 * h2; B[3] = 12.0 * h2_2; B[4] = 20.0 * h2_3; B[5] = 30.0 * h2_4;
    }
  } else {
    const double r3[6] = {-1.0, Pl[0], -Pl[1], Pl[2], -Pl[3], Pl[4]};
    const double r4[6] = {-1.0, h1, -h1_2, h1_3, -h1_4, h1_5};
    const double r5[6] = {-1.0, -h2, -h2_2, -h2_3, -h2_4, -h2_5};
    const double r6[6] = {-1.0, -Pr[0], -Pr[1], -Pr[2], -Pr[3], -Pr[4]};
    for (int n = 0; n < 6; n++) { A[n][2] = r3[n]; A[n][3] = r4[n]; A[n][4] = r5[n]; A[n][5] = r6[n]; }
    A[0][0] = 1.0; A[0][1] = 1.0;
    if (which == 0) {
      A[1][0] = -2.0 * h1;    A[1][1] = 2.0 * h2;
      A[2][0] = 3.0 * h1_2;   A[2][1] = 3.0 * h2_2;
      A[3][0] = -4.0 * h1_3;  A[3][1] = 4.0 * h2_3;
      A[4][0] = 5.0 * h1_4;   A[4][1] = 5.0 * h2_4;
      A[5][0] = -6.0 * h1_5;  A[5][1] = 6.0 * h2_5;
      B[0] = -1.0; B[1] = 0.0; B[2] = 0.0; B[3] = 0.0; B[4] = 0.0; B[5] = 0.0;
    } else if (which == 1) {
      const double h01 = h0 + h1, h01_2 = h01 * h01, h01_3 = h01 * h01_2;
      A[1][0] = -2.0 * h01;              A[1][1] = 0.0;
      A[2][0] = 3.0 * h01_2;             A[2][1] = 0.0;
      A[3][0] = -4.0 * h01_3;            A[3][1] = 0.0;
      A[4][0] = 5.0 * (h01_2 * h01_2);   A[4][1] = 0.0;
      A[5][0] = -6.0 * (h01_3 * h01_2);  A[5][1] = 0.0;
      B[0] = -1.0; B[1] = 2.0 * h1; B[2] = -3.0 * h1_2; B[3] = 4.0 * h1_3; B[4] = -5.0 * h1_4; B[5] = 6.0 * h1_5;
    } else {
      const double h23 = h2 + h3, h23_2 = h23 * h23, h23_3 = h23 * h23_2;
      A[1][0] = 0.0; A[1][1] = 2.0 * h23;
      A[2][0] = 0.0; A[2][1] = 3.0 * h23_2;
      A[3][0] = 0.0; A[3][1] = 4.0 * h23_3;
      A[4][0] = 0.0; A[4][1] = 5.0 * (h23_2 * h23_2);
      A[5][0] = 0.0; A[5][1] = 6.0 * (h23_3 * h23_2);
      B[0] = -1.0; B[1] = -2.0 * h2; B[2] = -3.0 * h2_2; B[3] = -4.0 * h2_3; B[4] = -5.0 * h2_4; B[5] = -6.0 * h2_5;
    }
  }
  linear_solver6(A, B, C);
}
// The tridiagonal system of edge_values_implicit_h6 (SLOPES = false) or edge_slopes_implicit_h5 (true) built across the lanes -- the
// interior rows a lane each, the four special rows on the last four lanes -- and solved by one lane with solve_tridiagonal_system's
// 2008-2018 expressions (both routines call it without an answer date, regrid_solvers.F90:199-217).  The solution is left in
// c.uh_sub[0 .. n] (the central diagonal is 1 in every row and is not stored).  n >= 6.
template <bool SLOPES>
__device__ void w_ih65_system(const WCol &c, int lane, int n, double hNeglect) {
  const double *h = c.h0, *u = c.u0;
  const double frac = SLOPES ? 1.0e-4 : 1.e-5;      // h_Min_Frac of :1042 / hMinFrac of :30
  double *tri_l = c.u_sub, *tri_u = c.u_sub + (n + 1), *tri_b = c.uh_sub, *pivot = c.uh_sub + (n + 1);
  double C[6];
  for (int k = lane + 1; k <= n - 3; k += 64) {      // Fortran k = 2 .. N-2, row k+1
    const double hMin = fmax(hNeglect, frac * ((h[k - 1] + h[k]) + (h[k + 1] + h[k + 2])));
    ih_coefs<SLOPES>(fmax(h[k - 1], hMin), fmax(h[k], hMin), fmax(h[k + 1], hMin), fmax(h[k + 2], hMin), 0, C);
    tri_l[k + 1] = C[0]; tri_u[k + 1] = C[1];
    tri_b[k + 1] = C[2] * u[k - 1] + C[3] * u[k] + C[4] * u[k + 1] + C[5] * u[k + 2];
  }
  if (lane == 60 || lane == 61) {      // the right-biased second row, the left-biased second to last row
    const int q = (lane == 60) ? 0 : n - 4;
    const double hMin = fmax(hNeglect, frac * ((h[q] + h[q + 1]) + (h[q + 2] + h[q + 3])));
    ih_coefs<SLOPES>(fmax(h[q], hMin), fmax(h[q + 1], hMin), fmax(h[q + 2], hMin), fmax(h[q + 3], hMin), (lane == 60) ? 1 : 2, C);
    const int row = (lane == 60) ? 1 : n - 1;
    tri_l[row] = C[0]; tri_u[row] = C[1];
    tri_b[row] = C[2] * u[q] + C[3] * u[q + 1] + C[4] * u[q + 2] + C[5] * u[q + 3];
  }
  if (lane >= 62) {      // the boundary conditions: a fifth-degree polynomial through the means of the six cells at the end
    const bool last = lane == 63;
    double hMin = 0.0;      // (the slopes' routine takes the widths as they are, :1166, :1212)
    if (!SLOPES) {
      if (!last) hMin = fmax(hNeglect, frac * ((h[0] + h[1]) + (h[4] + h[5]) + (h[2] + h[3])));
      else hMin = fmax(hNeglect, frac * (h[n - 4] + h[n - 3]) + ((h[n - 2] + h[n - 1]) + (h[n - 6] + h[n - 5])));      // (as :1436 has it)
    }
    double A[6][6], B[6], x = 0.0;
    for (int i = 0; i < 6; i++) {
      const int q = last ? n - 1 - i : i;
      const double dx = SLOPES ? h[q] : fmax(hMin, h[q]);
      const double xavg = x + 0.5 * dx;
      ih_boundary_row(xavg, dx, A[i]);
      B[i] = u[q];
      x = x + dx;
    }
    linear_solver6(A, B, C);
    double rhs;
    if (SLOPES) rhs = last ? -C[1] : C[1];
    else if (last) rhs = C[0];
    else {      // evaluation_polynomial( Csys, 6, 0.0 )
      rhs = 0.0;
      rhs = rhs + C[0] * 1.0;
      for (int k = 1; k < 6; k++) rhs = rhs + C[k] * 0.0;
    }
    const int row = last ? n : 0;
    tri_l[row] = 0.0; tri_u[row] = 0.0; tri_b[row] = rhs;
  }
  wsync();
  if (lane == 0) {
    const int N = n + 1;
    const double R_last = tri_b[N - 1];
    pivot[0] = 1.0;
    for (int k = 1; k < N; k++) {
      const double Al_piv = tri_l[k] / pivot[k - 1];
      pivot[k] = 1.0 - Al_piv * tri_u[k - 1];
      tri_b[k] = tri_b[k] - Al_piv * tri_b[k - 1];
    }
    tri_b[N - 1] = R_last / pivot[N - 1];
    for (int k = N - 2; k >= 0; k--) tri_b[k] = (tri_b[k] - tri_u[k] * tri_b[k + 1]) / pivot[k];
  }
  wsync();
}

// build_reconstructions_1d across the lanes; returns the integration method (uniform over the wave).
// Of ppoly_coef only the slope column is stored: coef(:,1) always equals the left edge value, and coef(:,3) is only read
// by the PPM boundary extrapolation, which recomputes it from the final edges with the expression that defined it.
__device__ int w_build_reconstructions(const WCol &c, int lane, int scheme, bool extrap, int n, double h_neglect, double h_neglect_edge) {
  const double *h = c.h0, *u = c.u0;
  double *EL = c.EL, *ER = c.ER;
  int local = scheme;
  if (n <= 1) local = REMAP_PCM;
  else if (n <= 3) local = (local < REMAP_PLM) ? local : REMAP_PLM;
  else if (n <= 4 && local != REMAP_PPM_CW) local = (local < REMAP_PPM_H4) ? local : REMAP_PPM_H4;      // :293
  if (local == REMAP_PCM) {
    for (int k = lane; k < n; k += 64) { EL[k] = u[k]; ER[k] = u[k]; c.C1[k] = 0.; }
    wsync();
    return INT_PCM;
  }
  if (local == REMAP_PLM) {      // PLM_reconstruction :190-260
    const double almost_one = 1. - DBL_EPSILON;
    double *slp = c.u_sub, *mslp = c.uh_sub;      // free until the sub-cell pass
    for (int k = lane; k < n; k += 64)
      slp[k] = (k >= 1 && k < n - 1) ? plm_slope_wa(h[k - 1], h[k], h[k + 1], h_neglect, u[k - 1], u[k], u[k + 1]) : 0.;
    wsync();
    for (int k = lane; k < n; k += 64)
      mslp[k] = (k >= 1 && k < n - 1) ? plm_monotonized_slope(u[k - 1], u[k], u[k + 1], slp[k - 1], slp[k], slp[k + 1]) : 0.;
    wsync();
    for (int k = lane; k < n; k += 64) {
      if (k >= 1 && k < n - 1) {
        const double slope = mslp[k];
        const double u_l = u[k] - 0.5 * slope;
        const double u_r = u[k] + 0.5 * slope;
        EL[k] = u_l; ER[k] = u_r;
        double c1 = (u_r - u_l);
        const double edge = c1 + u_l;
        const double e_r = u[k + 1] - 0.5 * fsign(mslp[k + 1], slp[k + 1]);
        if ((edge - u[k]) * (e_r - edge) < 0.) c1 = c1 * almost_one;
        c.C1[k] = c1;
      } else {
        EL[k] = u[k]; ER[k] = u[k]; c.C1[k] = 0.;
      }
    }
    wsync();
    if (extrap && lane == 0) {      // PLM_boundary_extrapolation :272-307
      double slope = -plm_extrapolate_slope(h[1], h[0], h_neglect, u[1], u[0]);
      EL[0] = u[0] - 0.5 * slope; ER[0] = u[0] + 0.5 * slope;
      c.C1[0] = ER[0] - EL[0];
      slope = plm_extrapolate_slope(h[n - 2], h[n - 1], h_neglect, u[n - 2], u[n - 1]);
      EL[n - 1] = u[n - 1] - 0.5 * slope; ER[n - 1] = u[n - 1] + 0.5 * slope;
      c.C1[n - 1] = ER[n - 1] - EL[n - 1];
    }
    wsync();
    return INT_PLM;
  }
  if (local == REMAP_PLM_HYBGEN) {      // hybgen_plm_coefs (MOM_hybgen_remap.F90:14-88), MOM_remapping.F90:306-315
    const double thin = h_neglect;
    for (int k = lane; k < n; k += 64) {
      double sl = 0.0;
      if (k >= 1 && k < n - 1 && !(h[k] <= thin)) {
        const double qcen = h[k] / (h[k] + 0.5 * (h[k - 1] + h[k + 1]));
        const double ztop = 2.0 * (u[k] - u[k - 1]);
        const double zbot = 2.0 * (u[k + 1] - u[k]);
        const double zcen = qcen * (u[k + 1] - u[k - 1]);
        if (ztop * zbot > 0.0) sl = fsign(min3(fabs(zcen), fabs(zbot), fabs(ztop)), zbot);
      }
      c.C1[k] = sl;
      EL[k] = u[k] - 0.5 * sl;
      ER[k] = u[k] + 0.5 * sl;
    }
    wsync();
    if (extrap && lane == 0) {      // PLM_boundary_extrapolation :272-307
      double slope = -plm_extrapolate_slope(h[1], h[0], h_neglect, u[1], u[0]);
      EL[0] = u[0] - 0.5 * slope; ER[0] = u[0] + 0.5 * slope;
      c.C1[0] = ER[0] - EL[0];
      slope = plm_extrapolate_slope(h[n - 2], h[n - 1], h_neglect, u[n - 2], u[n - 1]);
      EL[n - 1] = u[n - 1] - 0.5 * slope; ER[n - 1] = u[n - 1] + 0.5 * slope;
      c.C1[n - 1] = ER[n - 1] - EL[n - 1];
    }
    wsync();
    return INT_PLM;
  }
  if (local == REMAP_WENO_HYBGEN) {
    // ---- hybgen_weno_coefs (MOM_hybgen_remap.F90:226-386): the edge slopes, the two one-sided estimates of every cell with their
    // weights, the weighted edge values, the final limiter -- four passes across the lanes
    const double thin = h_neglect, min_ratio = 1.0e-8;
    auto dp = [&](int k) { return fmax(h[k], thin); };
    double *slope_edge = c.u_sub, *zw1 = c.uh_sub, *zw2 = c.uh_sub + n, *val_edge = c.C1;      // (free until the tail / the sub-cell pass)
    for (int K = lane + 1; K < n; K += 64) slope_edge[K] = (1.0 / (dp(K - 1) + dp(K))) * (u[K] - u[K - 1]);
    wsync();
    for (int k = lane; k < n; k += 64) {
      double e1 = u[k], e2 = u[k], w1 = 0.0, w2 = 0.0;
      if (k >= 1 && k < n - 1 && !((slope_edge[k] * slope_edge[k + 1] < 0.0) || (dp(k) <= thin))) {
        const double dpkm2kp = dp(k - 1) + 2.0 * dp(k) + dp(k + 1);
        const double qdpkmkp = 1.0 / (dp(k - 1) + dp(k) + dp(k + 1));
        double seh1 = dp(k) * slope_edge[k + 1];
        double seh2 = dp(k) * slope_edge[k];
        const double q01 = dpkm2kp * slope_edge[k + 1];
        const double q02 = dpkm2kp * slope_edge[k];
        if (fabs(seh1) > fabs(q02)) seh1 = q02;
        if (fabs(seh2) > fabs(q01)) seh2 = q01;
        const double curv_cell = (seh1 - seh2) * qdpkmkp;
        const double q001 = seh1 - curv_cell * dp(k + 1);
        const double q002 = seh2 + curv_cell * dp(k - 1);
        e2 = u[k] + q001;
        e1 = u[k] - q002;
        w1 = (2.0 * q001 - q002) * (2.0 * q001 - q002);
        w2 = (2.0 * q002 - q001) * (2.0 * q002 - q001);
      }
      EL[k] = e1; ER[k] = e2; zw1[k] = w1; zw2[k] = w2;
    }
    wsync();
    for (int K = lane + 1; K < n; K += 64) {
      double wt1;
      if (zw1[K] + zw2[K - 1] <= 0.0) wt1 = 0.5;
      else if (zw1[K] <= min_ratio * (zw1[K] + zw2[K - 1])) wt1 = min_ratio;
      else if (zw2[K - 1] <= min_ratio * (zw1[K] + zw2[K - 1])) wt1 = (1.0 - min_ratio);
      else wt1 = zw1[K] / (zw1[K] + zw2[K - 1]);
      val_edge[K] = wt1 * ER[K - 1] + (1.0 - wt1) * EL[K];
    }
    wsync();
    for (int k = lane + 1; k < n - 1; k += 64) {
      if (!(dp(k) <= thin)) {
        double q01 = val_edge[k + 1] - u[k];
        double q02 = u[k] - val_edge[k];
        if (q01 * q02 < 0.0) { q01 = 0.0; q02 = 0.0; }
        else if (fabs(q01) > fabs(2.0 * q02)) q01 = 2.0 * q02;
        else if (fabs(q02) > fabs(2.0 * q01)) q02 = 2.0 * q01;
        EL[k] = u[k] - q02;
        ER[k] = u[k] + q01;
      }
    }
    wsync();
  } else
  if (local == REMAP_PPM_IH4 || local == REMAP_PQM_IH4IH3) {
    // ---- PPM_IH4, PQM_IH4IH3: edge_values_implicit_h4 (regrid_edge_values.F90:491-654, answer_date >= 20190101).  The rows of the
    // tridiagonal system across the lanes, the two closing rows on the last two lanes, solve_diag_dominant_tridiag
    // (regrid_solvers.F90:246-280) as a serial walk by one lane (the solution overwrites the right-hand side).
    const double hNeglect = h_neglect_edge;
    double *tri_l = c.u_sub, *tri_c = c.u_sub + (n + 1), *tri_u = c.uh_sub, *tri_b = c.uh_sub + (n + 1), *c1 = c.C1;
    for (int i = lane; i < n - 1; i += 64) {
      double h0 = fmax(h[i], hNeglect);
      double h1 = fmax(h[i + 1], hNeglect);
      if (fabs(h0) < 1.0e-12 * fabs(h1)) h0 = 1.0e-12 * h1;
      if (fabs(h1) < 1.0e-12 * fabs(h0)) h1 = 1.0e-12 * h0;
      const double I_h2 = 1.0 / ((h0 + h1) * (h0 + h1));
      const double alpha = (h1 * h1) * I_h2;
      const double beta = (h0 * h0) * I_h2;
      const double abmix = (h0 * h1) * I_h2;
      const double a = 2.0 * alpha * (alpha + 2.0 * beta + 3.0 * abmix);
      const double b = 2.0 * beta * (beta + 2.0 * alpha + 3.0 * abmix);
      tri_c[i + 1] = 2.0 * abmix; tri_l[i + 1] = alpha; tri_u[i + 1] = beta;
      tri_b[i + 1] = a * u[i] + b * u[i + 1];
    }
    if (lane >= 62) {      // the first (lane 62) and the last (lane 63) boundary value
      const bool last = lane == 63;
      double dz[4], ut[4], Cs[4];
      for (int i = 0; i < 4; i++) { const int q = last ? n - 1 - i : i; dz[i] = fmax(hNeglect, h[q]); ut[i] = u[q]; }
      end_value_h4(dz, ut, Cs);
      const int row = last ? n : 0;
      tri_b[row] = Cs[0]; tri_c[row] = 1.0; tri_u[row] = 0.0; tri_l[row] = 0.0;
    }
    wsync();
    if (lane == 0) {
      const int N = n + 1;
      double I_pivot = 1.0 / (tri_c[0] + tri_u[0]);
      double d1 = tri_c[0] * I_pivot;
      c1[0] = tri_u[0] * I_pivot;
      tri_b[0] = tri_b[0] * I_pivot;
      for (int k = 1; k < N - 1; k++) {
        const double denom_t1 = tri_c[k] + d1 * tri_l[k];
        I_pivot = 1.0 / (denom_t1 + tri_u[k]);
        d1 = denom_t1 * I_pivot;
        c1[k] = tri_u[k] * I_pivot;
        tri_b[k] = (tri_b[k] - tri_l[k] * tri_b[k - 1]) * I_pivot;
      }
      I_pivot = 1.0 / (tri_c[N - 1] + d1 * tri_l[N - 1]);
      tri_b[N - 1] = (tri_b[N - 1] - tri_l[N - 1] * tri_b[N - 2]) * I_pivot;
      for (int k = N - 2; k >= 0; k--) tri_b[k] = tri_b[k] - c1[k] * tri_b[k + 1];
    }
    wsync();
    for (int k = lane; k < n; k += 64) { EL[k] = tri_b[k]; ER[k] = tri_b[k + 1]; }
    wsync();
  } else if (local == REMAP_PQM_IH6IH5) {
    w_ih65_system<false>(c, lane, n, h_neglect_edge);      // edge_values_implicit_h6
    for (int k = lane; k < n; k += 64) { EL[k] = c.uh_sub[k]; ER[k] = c.uh_sub[k + 1]; }
    wsync();
  } else if (local == REMAP_PPM_CW || local == REMAP_PPM_HYBGEN) {
    // ---- PPM_CW: edge_values_explicit_h4cw (regrid_edge_values.F90:381-463) and PPM_monotonicity (PPM_functions.F90:132).
    // ---- PPM_HYBGEN: hybgen_ppm_coefs (MOM_hybgen_remap.F90:91-222), the HYCOM routine the two above re-express: the same
    //      arithmetic with the minimum thickness `thin` = h_neglect and layers no thicker than it treated as PCM (:140-147)
    const bool hyb = local == REMAP_PPM_HYBGEN;
    const double hNeglect = hyb ? h_neglect : h_neglect_edge;
    double *au = c.u_sub;      // the limited slopes of Colella & Woodward eq. 1.8, 0-based cell index
    auto dp = [&](int k) { return fmax(h[k], hNeglect); };      // 0-based; the reference's dp(k+1)
    auto pcm = [&](int k) { return hyb && (dp(k) <= hNeglect); };
    for (int k = lane; k < n; k += 64) {
      double a = 0.;
      if (k >= 1 && k <= n - 2) {
        const double slk = u[k] - u[k - 1];
        const double srk = u[k + 1] - u[k];
        if (!pcm(k) && slk * srk > 0.) {
          // h2_h123(k), h112(K), I_h12(K+1), h122(K+1), I_h12(K) of the reference with K = k+1 (1-based)
          const double h2_h123 = dp(k) / (dp(k) + (dp(k - 1) + dp(k + 1)));
          const double h112 = 2. * dp(k - 1) + dp(k), h122p = dp(k) + 2. * dp(k + 1);
          const double I_h12 = 1.0 / (dp(k - 1) + dp(k)), I_h12p = 1.0 / (dp(k) + dp(k + 1));
          const double sck = h2_h123 * (h112 * srk * I_h12p + h122p * slk * I_h12);
          a = fsign(min3(fabs(2.0 * slk), fabs(sck), fabs(2.0 * srk)), sck);
        }
      }
      au[k] = a;
    }
    wsync();
    for (int k = lane; k < n; k += 64) {      // al(k): the edge between cells k-1 and k; also ar(k-1)
      if (k >= 2 && k <= n - 2) {
        const double I_h12 = 1.0 / (dp(k - 1) + dp(k));
        const double I_h0123 = 1.0 / ((dp(k - 2) + dp(k - 1)) + (dp(k) + dp(k + 1)));
        const double h01_h112 = (dp(k - 2) + dp(k - 1)) / (2.0 * dp(k - 1) + dp(k));
        const double h23_h122 = (dp(k) + dp(k + 1)) / (dp(k - 1) + 2.0 * dp(k));
        const double al = (dp(k) * u[k - 1] + dp(k - 1) * u[k]) * I_h12 +
                          I_h0123 * (2. * dp(k) * dp(k - 1) * I_h12 * (u[k] - u[k - 1]) * (h01_h112 - h23_h122) +
                                     (dp(k) * au[k - 1] * h23_h122 - dp(k - 1) * au[k] * h01_h112));
        EL[k] = al; ER[k - 1] = al;
      }
    }
    if (lane == 0) { EL[0] = u[0]; ER[0] = u[0]; EL[1] = u[0]; ER[n - 2] = u[n - 1]; EL[n - 1] = u[n - 1]; ER[n - 1] = u[n - 1]; }
    wsync();
    for (int k = lane + 1; k < n - 1; k += 64) {      // PPM_monotonicity
      if (pcm(k) || (u[k + 1] - u[k]) * (u[k] - u[k - 1]) <= 0.) {
        EL[k] = u[k]; ER[k] = u[k];
      } else {
        const double da = ER[k] - EL[k];
        const double a6 = 6.0 * u[k] - 3.0 * (EL[k] + ER[k]);
        if (da * a6 > da * da) EL[k] = 3.0 * u[k] - 2.0 * ER[k];
        else if (da * a6 < -da * da) ER[k] = 3.0 * u[k] - 2.0 * EL[k];
      }
    }
    wsync();
  } else
  // ---- PPM_H4: edge_values_explicit_h4 :222-363
  {
    const double hMinFrac = 1.e-5, hNeglect = h_neglect_edge;
    for (int i = lane + 2; i <= n - 2; i += 64) {
      double h0 = h[i - 2], h1 = h[i - 1], h2 = h[i], h3 = h[i + 1];
      if (h0 + h1 == 0.0 || h1 + h2 == 0.0 || h2 + h3 == 0.0) {
        const double h_min = hMinFrac * fmax(hNeglect, (h0 + h1) + (h2 + h3));
        h0 = fmax(h_min, h[i - 2]); h1 = fmax(h_min, h[i - 1]); h2 = fmax(h_min, h[i]); h3 = fmax(h_min, h[i + 1]);
      }
      const double I_h12 = 1.0 / (h1 + h2);
      const double I_den_et2 = 1.0 / (((h0 + h1) + h2) * (h0 + h1)); const double I_h012 = (h0 + h1) * I_den_et2;
      const double I_den_et3 = 1.0 / ((h1 + (h2 + h3)) * (h2 + h3)); const double I_h123 = (h2 + h3) * I_den_et3;
      const double et1 = (1.0 + (h1 * I_h012 + (h0 + h1) * I_h123)) * I_h12 * (h2 * (h2 + h3)) * u[i - 1] +
                         (1.0 + (h2 * I_h123 + (h2 + h3) * I_h012)) * I_h12 * (h1 * (h0 + h1)) * u[i];
      const double et2 = (h1 * (h2 * (h2 + h3)) * I_den_et2) * (u[i - 1] - u[i - 2]);
      const double et3 = (h2 * (h1 * (h0 + h1)) * I_den_et3) * (u[i] - u[i + 1]);
      const double ev = (et1 + (et2 + et3)) / ((h0 + h1) + (h2 + h3));
      EL[i] = ev; ER[i - 1] = ev;
    }
    if (lane >= 62) {      // the end values, on the two lanes that have no second interior edge to do: one code path for both
      const bool last = lane == 63;
      double dz[4], ut[4], Cs[4];
      for (int i = 0; i < 4; i++) { const int q = last ? n - 1 - i : i; dz[i] = fmax(hNeglect, h[q]); ut[i] = u[q]; }
      end_value_h4(dz, ut, Cs);
      const double inner = Cs[0] + dz[0] * (Cs[1] + dz[0] * (Cs[2] + dz[0] * Cs[3]));
      if (last) { ER[n - 1] = Cs[0]; EL[n - 1] = inner; ER[n - 2] = inner; }
      else { EL[0] = Cs[0]; ER[0] = inner; EL[1] = inner; }
    }
    wsync();
  }
  if (local == REMAP_PQM_IH4IH3) {
    // ---- edge_slopes_implicit_h3 (regrid_edge_values.F90:803-972, answer_date >= 20190101): the same shape of system as the edge
    // values', with h_neglect; the left slopes go to C1 (the solver's work array until then), the right ones to SR
    const double hNeglect = h_neglect;
    double *tri_l = c.u_sub, *tri_c = c.u_sub + (n + 1), *tri_u = c.uh_sub, *tri_b = c.uh_sub + (n + 1), *c1 = c.C1;
    for (int i = lane; i < n - 1; i += 64) {
      double h0 = fmax(h[i], hNeglect);
      double h1 = fmax(h[i + 1], hNeglect);
      const double I_h = 1.0 / (h0 + h1);
      h0 = h0 * I_h; h1 = h1 * I_h;
      const double h0h1 = h0 * h1, h0_2 = h0 * h0, h1_2 = h1 * h1;
      const double h0_3 = h0_2 * h0, h1_3 = h1_2 * h1;
      const double I_d = 1.0 / (4.0 * h0h1 * (h0 + h1) + h1_3 + h0_3);
      tri_l[i + 1] = (h1 * ((h0_2 + h0h1) - h1_2)) * I_d;
      tri_c[i + 1] = 2.0 * ((h0_2 + h1_2) * (h0 + h1)) * I_d;
      tri_u[i + 1] = (h0 * ((h1_2 + h0h1) - h0_2)) * I_d;
      tri_b[i + 1] = 12.0 * (h0h1 * I_d) * ((u[i + 1] - u[i]) * I_h);
    }
    if (lane >= 62) {      // the first (lane 62) and the last (lane 63) edge slope
      const bool last = lane == 63;
      double dz[4], ut[4], Cs[4];
      for (int i = 0; i < 4; i++) { const int q = last ? n - 1 - i : i; dz[i] = fmax(hNeglect, h[q]); ut[i] = u[q]; }
      end_value_h4(dz, ut, Cs);
      const int row = last ? n : 0;
      tri_b[row] = last ? -Cs[1] : Cs[1]; tri_c[row] = 1.0; tri_u[row] = 0.0; tri_l[row] = 0.0;
    }
    wsync();
    if (lane == 0) {      // solve_diag_dominant_tridiag
      const int N = n + 1;
      double I_pivot = 1.0 / (tri_c[0] + tri_u[0]);
      double d1 = tri_c[0] * I_pivot;
      c1[0] = tri_u[0] * I_pivot;
      tri_b[0] = tri_b[0] * I_pivot;
      for (int k = 1; k < N - 1; k++) {
        const double denom_t1 = tri_c[k] + d1 * tri_l[k];
        I_pivot = 1.0 / (denom_t1 + tri_u[k]);
        d1 = denom_t1 * I_pivot;
        c1[k] = tri_u[k] * I_pivot;
        tri_b[k] = (tri_b[k] - tri_l[k] * tri_b[k - 1]) * I_pivot;
      }
      I_pivot = 1.0 / (tri_c[N - 1] + d1 * tri_l[N - 1]);
      tri_b[N - 1] = (tri_b[N - 1] - tri_l[N - 1] * tri_b[N - 2]) * I_pivot;
      for (int k = N - 2; k >= 0; k--) tri_b[k] = tri_b[k] - c1[k] * tri_b[k + 1];
    }
    wsync();
    for (int k = lane; k < n; k += 64) { c.C1[k] = tri_b[k]; c.SR[k] = tri_b[k + 1]; }
    wsync();
  }
  if (local == REMAP_PQM_IH6IH5) {
    w_ih65_system<true>(c, lane, n, h_neglect);      // edge_slopes_implicit_h5
    for (int k = lane; k < n; k += 64) { c.C1[k] = c.uh_sub[k]; c.SR[k] = c.uh_sub[k + 1]; }
    wsync();
  }
  // ---- PPM_reconstruction / PQM_limiter: bound_edge_values, check_discontinuous_edge_values, then the scheme's limiter
  for (int k = lane; k < n; k += 64) {
    const int km1 = (k - 1 > 0) ? k - 1 : 0, kp1 = (k + 1 < n - 1) ? k + 1 : n - 1;
    double slope_x_h = 0.0;
    if (((h[km1] + h[kp1]) + 2.0 * h[k]) > 0.0) {
      const double sigma_l = (u[k] - u[km1]);
      const double sigma_c = (u[kp1] - u[km1]) * (h[k] / ((h[km1] + h[kp1]) + 2.0 * h[k]));
      const double sigma_r = (u[kp1] - u[k]);
      if ((sigma_l * sigma_r) > 0.0) slope_x_h = fsign(min3(fabs(sigma_l), fabs(sigma_c), fabs(sigma_r)), sigma_c);
    }
    double el = EL[k], er = ER[k];
    if ((u[km1] - el) * (el - u[k]) < 0.0) el = u[k] - fsign(fmin(fabs(slope_x_h), fabs(el - u[k])), slope_x_h);
    if ((u[kp1] - er) * (er - u[k]) < 0.0) er = u[k] + fsign(fmin(fabs(slope_x_h), fabs(er - u[k])), slope_x_h);
    el = fmax(fmin(el, fmax(u[km1], u[k])), fmin(u[km1], u[k]));
    er = fmax(fmin(er, fmax(u[kp1], u[k])), fmin(u[kp1], u[k]));
    EL[k] = el; ER[k] = er;
  }
  wsync();
  for (int k = lane; k < n - 1; k += 64) {
    if ((EL[k + 1] - ER[k]) * (u[k + 1] - u[k]) < 0.0) {
      double u0_avg = 0.5 * (ER[k] + EL[k + 1]);
      u0_avg = fmax(fmin(u0_avg, fmax(u[k], u[k + 1])), fmin(u[k], u[k + 1]));
      ER[k] = u0_avg; EL[k + 1] = u0_avg;
    }
  }
  wsync();
  if (local == REMAP_PQM_IH4IH3 || local == REMAP_PQM_IH6IH5) {
    // ---- PQM_limiter (PQM_functions.F90:103-337): a cell reads its own edge values and slopes and its neighbours' means and widths
    const double hNeglect = h_neglect;
    double *SL = c.C1, *SR = c.SR;
    for (int k = lane; k < n; k += 64) {
      double u0_l = u[k], u0_r = u[k], u1_l = 0.0, u1_r = 0.0;      // the boundary cells :331-335
      if (k >= 1 && k < n - 1) {
        u0_l = EL[k]; u0_r = ER[k]; u1_l = SL[k]; u1_r = SR[k];
        const double h_l = h[k - 1], h_c = h[k], h_r = h[k + 1];
        const double u_l = u[k - 1], u_c = u[k], u_r = u[k + 1];
        const double sigma_l = 2.0 * (u_c - u_l) / (h_c + hNeglect);
        const double sigma_c = 2.0 * (u_r - u_l) / (h_l + 2.0 * h_c + h_r + hNeglect);
        const double sigma_r = 2.0 * (u_r - u_c) / (h_c + hNeglect);
        double slope = 0.0;
        if ((sigma_l * sigma_r) > 0.0) slope = fsign(min3(fabs(sigma_l), fabs(sigma_c), fabs(sigma_r)), sigma_c);
        if (u1_l * slope <= 0.0) u1_l = slope;
        if (u1_r * slope <= 0.0) u1_r = slope;
        int inflexion = 0;      // 1: collapse the inflexion points onto the left edge, 2: onto the right edge
        if ((u0_r - u_c) * (u_c - u0_l) <= 0.0) {      // a local extremum: flat
          u0_l = u_c; u0_r = u_c; u1_l = 0.0; u1_r = 0.0;
        } else if (pqm_bad_inflexion(pqm_quartic(u_c, h_c, u0_l, u0_r, u1_l, u1_r), slope, true)) {
          inflexion = (fabs(sigma_l) < fabs(sigma_r)) ? 1 : 2;
        }
        if (inflexion == 1) {
          u1_l = (10.0 * u_c - 2.0 * u0_r - 8.0 * u0_l) / (3.0 * h_c + hNeglect);
          u1_r = (-10.0 * u_c + 6.0 * u0_r + 4.0 * u0_l) / (h_c + hNeglect);
          if (u1_l * slope < 0.0) {
            u1_l = 0.0;
            u0_r = 5.0 * u_c - 4.0 * u0_l;
            u1_r = 20.0 * (u_c - u0_l) / (h_c + hNeglect);
          } else if (u1_r * slope < 0.0) {
            u1_r = 0.0;
            u0_l = (5.0 * u_c - 3.0 * u0_r) / 2.0;
            u1_l = 10.0 * (-u_c + u0_r) / (3.0 * h_c + hNeglect);
          }
        } else if (inflexion == 2) {
          u1_r = (-10.0 * u_c + 8.0 * u0_r + 2.0 * u0_l) / (3.0 * h_c + hNeglect);
          u1_l = (10.0 * u_c - 4.0 * u0_r - 6.0 * u0_l) / (h_c + hNeglect);
          if (u1_l * slope < 0.0) {
            u1_l = 0.0;
            u0_r = (5.0 * u_c - 3.0 * u0_l) / 2.0;
            u1_r = 10.0 * (u_c - u0_l) / (3.0 * h_c + hNeglect);
          } else if (u1_r * slope < 0.0) {
            u1_r = 0.0;
            u0_l = 5.0 * u_c - 4.0 * u0_r;
            u1_l = 20.0 * (-u_c + u0_r) / (h_c + hNeglect);
          }
        }
      }
      EL[k] = u0_l; ER[k] = u0_r; SL[k] = u1_l; SR[k] = u1_r;
    }
    wsync();
    if (extrap && lane >= 62) {      // PQM_boundary_extrapolation_v1 (PQM_functions.F90:502-831): lane 62 the top cell, lane 63 the bottom cell
      const bool bottom = lane == 63;
      double u0_l, u0_r, u1_l, u1_r, slope, um;
      if (!bottom) {
        const double h0 = h[0], h1 = h[1], u0 = u[0], u1 = u[1];
        um = u0;
        slope = 2.0 * (u1 - u0) / ((h0 + h1) + hNeglect);
        slope = slope * h0;
        u0_r = EL[1];                                   // ppoly_coef(i1,1)
        u1_r = (h1 * SL[1]) / (h1 + hNeglect);          // ppoly_coef(i1,2) / (h1 + hNeglect)
        double beta = 0.;
        if (u1_r != 0.) beta = 2.0 * (u0_r - um) / ((h0 + hNeglect) * u1_r) - 1.0;
        const double br = u0_r + beta * u0_r - um;
        const double ar = um + beta * um - br;
        u0_l = ar;
        const double u_plm = um - 0.5 * slope;
        if (fabs(um - u0_l) < fabs(um - u_plm)) {
          u1_l = 2.0 * (br - ar * beta);
          u1_l = u1_l / (h0 + hNeglect);
        } else {
          u0_l = u_plm;
          u1_l = slope / (h0 + hNeglect);
        }
        if (pqm_bad_inflexion(pqm_quartic(um, h0, u0_l, u0_r, u1_l, u1_r), slope, false)) {
          u1_l = (10.0 * um - 2.0 * u0_r - 8.0 * u0_l) / (3.0 * h0 + hNeglect);
          u1_r = (-10.0 * um + 6.0 * u0_r + 4.0 * u0_l) / (h0 + hNeglect);
          if (u1_l * slope < 0.0) {
            u1_l = 0.0;
            u0_r = 5.0 * um - 4.0 * u0_l;
            u1_r = 20.0 * (um - u0_l) / (h0 + hNeglect);
          } else if (u1_r * slope < 0.0) {
            u1_r = 0.0;
            u0_l = (5.0 * um - 3.0 * u0_r) / 2.0;
            u1_l = 10.0 * (-um + u0_r) / (3.0 * h0 + hNeglect);
          }
        }
      } else {
        const int i0 = n - 2, i1 = n - 1;
        const double h0 = h[i0], h1 = h[i1], u0 = u[i0], u1 = u[i1];
        um = u1;
        slope = 2.0 * (u1 - u0) / (h0 + h1);
        slope = slope * h1;
        const Quartic q0 = pqm_quartic(u0, h0, EL[i0], ER[i0], SL[i0], SR[i0]);      // ppoly_coef(i0,:), a = EL[i0]
        u0_l = EL[i0] + q0.b + q0.c + q0.d + q0.e;
        u1_l = (q0.b + 2 * q0.c + 3 * q0.d + 4 * q0.e) / h0;
        double beta = 0.;
        if (um - u0_l != 0.) beta = 0.5 * h1 * u1_l / (um - u0_l) - 1.0;
        const double br = beta * um + um - u0_l;
        const double ar = u0_l;
        if (1 + beta != 0.) u0_r = (ar + 2 * br + beta * br) / ((1 + beta) * (1 + beta));
        else u0_r = um + 0.5 * slope;
        const double u_plm = um + 0.5 * slope;
        if (fabs(um - u0_r) < fabs(um - u_plm)) {
          u1_r = 2.0 * (br - ar * beta) / ((1 + beta) * (1 + beta) * (1 + beta));
          u1_r = u1_r / h1;
        } else {
          u0_r = u_plm;
          u1_r = slope / h1;
        }
        if (pqm_bad_inflexion(pqm_quartic(um, h1, u0_l, u0_r, u1_l, u1_r), slope, false)) {
          u1_r = (-10.0 * um + 8.0 * u0_r + 2.0 * u0_l) / (3.0 * h1);
          u1_l = (10.0 * um - 4.0 * u0_r - 6.0 * u0_l) / h1;
          if (u1_l * slope < 0.0) {
            u1_l = 0.0;
            u0_r = (5.0 * um - 3.0 * u0_l) / 2.0;
            u1_r = 10.0 * (um - u0_l) / (3.0 * h1);
          } else if (u1_r * slope < 0.0) {
            u1_r = 0.0;
            u0_l = 5.0 * um - 4.0 * u0_r;
            u1_l = 20.0 * (-um + u0_r) / h1;
          }
        }
      }
      const int kb = bottom ? n - 1 : 0;      // (the two lanes read cells 1 and n-2 and write cells 0 and n-1: n >= 5 here)
      EL[kb] = u0_l; ER[kb] = u0_r; SL[kb] = u1_l; SR[kb] = u1_r;
    }
    wsync();
    return INT_PQM;
  }
  for (int k = lane; k < n; k += 64) {
    double edge_l, edge_r;
    if (k >= 1 && k < n - 1) {
      const double u_l = u[k - 1], u_c = u[k], u_r = u[k + 1];
      edge_l = EL[k]; edge_r = ER[k];
      if ((u_r - u_c) * (u_c - u_l) <= 0.0) {
        edge_l = u_c; edge_r = u_c;
      } else {
        const double expr1 = 3.0 * (edge_r - edge_l) * ((u_c - edge_l) + (u_c - edge_r));
        const double expr2 = (edge_r - edge_l) * (edge_r - edge_l);
        if (expr1 > expr2) {
          edge_l = u_c + 2.0 * (u_c - edge_r);
          edge_l = fmax(fmin(edge_l, fmax(u_l, u_c)), fmin(u_l, u_c));
        } else if (expr1 < -expr2) {
          edge_r = u_c + 2.0 * (u_c - edge_l);
          edge_r = fmax(fmin(edge_r, fmax(u_r, u_c)), fmin(u_r, u_c));
        }
      }
      if (fabs(edge_r - edge_l) < fmax(1.e-60, DBL_EPSILON * fabs(u_c))) { edge_l = u_c; edge_r = u_c; }
    } else {
      edge_l = u[k]; edge_r = u[k];
    }
    EL[k] = edge_l; ER[k] = edge_r;
    c.C1[k] = 4.0 * (u[k] - edge_l) + 2.0 * (u[k] - edge_r);
  }
  wsync();
  if (extrap && lane == 0) {      // PPM_boundary_extrapolation, PPM_functions.F90:162-316
    int i0 = 0, i1 = 1;
    double h0 = h[i0], h1 = h[i1], u0 = u[i0], u1 = u[i1];
    double b = c.C1[i1];
    double u1_r = b * ((h0 + h_neglect) / (h1 + h_neglect));
    double slope = 2.0 * (u1 - u0);
    if (fabs(u1_r) > fabs(slope)) u1_r = slope;
    double u0_r = EL[i1];
    double u0_l = 3.0 * u0 + 0.5 * u1_r - 2.0 * u0_r;
    double exp1 = (u0_r - u0_l) * (u0 - 0.5 * (u0_l + u0_r));
    double exp2 = (u0_r - u0_l) * (u0_r - u0_l) / 6.0;
    if (exp1 > exp2) u0_l = 3.0 * u0 - 2.0 * u0_r;
    if (exp1 < -exp2) u0_r = 3.0 * u0 - 2.0 * u0_l;
    EL[i0] = u0_l; ER[i0] = u0_r;
    c.C1[i0] = 6.0 * u0 - 4.0 * u0_l - 2.0 * u0_r;
    i0 = n - 2; i1 = n - 1;
    h0 = h[i0]; h1 = h[i1]; u0 = u[i0]; u1 = u[i1];
    b = c.C1[i0];
    const double cc = 3.0 * ((ER[i0] - u[i0]) + (EL[i0] - u[i0]));      // ppoly_coef(i0,3)
    double u1_l = (b + 2 * cc);
    u1_l = u1_l * ((h1 + h_neglect) / (h0 + h_neglect));
    slope = 2.0 * (u1 - u0);
    if (fabs(u1_l) > fabs(slope)) u1_l = slope;
    u0_l = ER[i0];
    u0_r = 3.0 * u1 - 0.5 * u1_l - 2.0 * u0_l;
    exp1 = (u0_r - u0_l) * (u1 - 0.5 * (u0_l + u0_r));
    exp2 = (u0_r - u0_l) * (u0_r - u0_l) / 6.0;
    if (exp1 > exp2) u0_l = 3.0 * u1 - 2.0 * u0_r;
    if (exp1 < -exp2) u0_r = 3.0 * u1 - 2.0 * u0_l;
    EL[i1] = u0_l; ER[i1] = u0_r;
    c.C1[i1] = 6.0 * u1 - 4.0 * u0_l - 2.0 * u0_r;
  }
  wsync();
  return INT_PPM;
}

// average_value_ppoly :998-1099 on the LDS column
__device__ __forceinline__ double w_average_value_ppoly(const WCol &c, int method, int i0, double xa, double xb) {
  if (method == INT_PQM) {      // the quartic's coefficients from the cell's edge values and slopes, by the expressions that define them
    const double a = c.EL[i0];
    const Quartic q = pqm_quartic(c.u0[i0], c.h0[i0], a, c.ER[i0], c.C1[i0], c.SR[i0]);
    if (xb > xa) {
      const double r_3 = 1.0 / 3.0;
      const double xa_2 = xa * xa, xb_2 = xb * xb;
      const double xa2pxb2 = xa_2 + xb_2, xapxb = xa + xb;
      return (a + (q.b * 0.5 * (xapxb) + (q.c * r_3 * (xa2pxb2 + xa * xb) + (q.d * 0.25 * (xa2pxb2 * xapxb) +
                                                                             q.e * 0.2 * ((xb * xb_2 + xa * xa_2) * xapxb + xa_2 * xb_2)))));
    }
    return a + xa * (q.b + xa * (q.c + xa * (q.d + xa * q.e)));
  }
  if (xb > xa) {
    if (method == INT_PCM) return c.u0[i0];
    if (method == INT_PLM) return (c.EL[i0] + c.C1[i0] * 0.5 * (xb + xa));
    const double mx = 0.5 * (xa + xb);
    const double a_L = c.EL[i0], a_R = c.ER[i0], u_c = c.u0[i0];
    const double a_c = 0.5 * ((u_c - a_L) + (u_c - a_R));
    if (mx < 0.5) {
      const double xa2b2ab = (xa * xa + xb * xb) + xa * xb;
      return a_L + ((a_R - a_L) * mx + a_c * (3. * (xb + xa) - 2. * xa2b2ab));
    } else {
      const double Ya = 1. - xa, Yb = 1. - xb;
      const double my = 0.5 * (Ya + Yb);
      const double Ya2b2ab = (Ya * Ya + Yb * Yb) + Ya * Yb;
      return a_R + ((a_L - a_R) * my + a_c * (3. * (Yb + Ya) - 2. * Ya2b2ab));
    }
  } else {
    if (method == INT_PCM) return c.EL[i0];
    const double a_L = c.EL[i0], a_R = c.ER[i0];
    const double Ya = 1. - xa;
    if (method == INT_PLM) {
      if (xa < 0.5) return a_L + xa * (a_R - a_L);
      return a_R + Ya * (a_L - a_R);
    }
    const double u_c = c.u0[i0];
    const double a_c = 3. * ((u_c - a_L) + (u_c - a_R));
    if (xa < 0.5) return a_L + xa * ((a_R - a_L) + a_c * Ya);
    return a_R + Ya * ((a_L - a_R) + a_c * xa);
  }
}

// The field-dependent part of remap_via_sub_cells :653-766; the result replaces u0.
__device__ void w_integrate_sub_cells(const WCol &c, int lane, int n0, int n1, int method, double cu) {
  const int ns = n0 + n1 + 1;
  const int i0_last_thick_cell = c.itgt_end[n1 + 1];
  // u_sub / uh_sub of the sub-cells 2 .. n0+n1: the running (xa, dh0_eff) restart with every source cell, so one lane
  // walks the sub-cells of one source cell; the sub-cells after the source column has run out keep i0 = n0 and the
  // running values, so the lane of the last source cell walks on to n0+n1.
  for (int i0 = lane + 1; i0 <= n0; i0 += 64) {
    int first = (i0 > 1) ? c.isrc_end[i0 - 1] + 1 : 2;
    int last = (i0 == n0) ? n0 + n1 : c.isrc_end[i0];
    double xa = 0., xb, dh0_eff = 0.;
    const double h0e = c.h0_eff[i0];
    for (int i_sub = first; i_sub <= last; i_sub++) {
      const double dh = c.h_sub[i_sub];
      dh0_eff = dh0_eff + dh;
      double us;
      if (h0e > 0.) {
        xb = dh0_eff / h0e;
        xb = fmin(1., xb);
        us = w_average_value_ppoly(c, method, i0 - 1, xa, xb);
      } else {
        xb = 1.;
        us = c.u0[i0 - 1];
      }
      c.u_sub[i_sub] = us;
      c.uh_sub[i_sub] = dh * us;
      xa = xb;
    }
  }
  if (lane == 0) {
    c.uh_sub[1] = 0.; c.u_sub[1] = c.EL[0];
    c.u_sub[ns] = c.ER[n0 - 1];
    c.uh_sub[ns] = c.ER[n0 - 1] * c.h_sub[ns];
  }
  wsync();
  // the thickest sub-cell of every source cell with volume takes the remainder :700-718
  for (int i0 = lane + 1; i0 <= i0_last_thick_cell; i0 += 64) {
    const int i_max = c.isrc_max[i0];
    const double dh_max = c.h_sub[i_max];
    if (dh_max > 0.) {
      double duh = 0.;
      const int first = (i0 > 1) ? c.isrc_end[i0 - 1] + 1 : 1, last = c.isrc_end[i0];
      for (int i_sub = first; i_sub <= last; i_sub++)
        if (i_sub != i_max) duh = duh + c.uh_sub[i_sub];
      c.uh_sub[i_max] = c.u0[i0 - 1] * c.h0[i0 - 1] - duh;
    }
  }
  wsync();
  // target cells :720-766; u0 is dead after the pass above, so the result goes there
  for (int i1 = lane + 1; i1 <= n1; i1 += 64) {
    const int e_prev = (i1 > 1) ? c.itgt_end[i1 - 1] : 0, e = c.itgt_end[i1];
    const int first = ((e_prev < 0) ? -e_prev : e_prev) + 1;
    double r;
    if (e > 0) {      // h1(i1) > 0
      double duh = 0., dh = 0.;
      double u1min = c.u_sub[first], u1max = c.u_sub[first];
      for (int i_sub = first; i_sub <= e; i_sub++) {
        u1min = fmin(u1min, c.u_sub[i_sub]);
        u1max = fmax(u1max, c.u_sub[i_sub]);
        dh = dh + c.h_sub[i_sub];
        duh = duh + c.uh_sub[i_sub];
      }
      r = duh / dh;
      r = fmax(u1min, fmin(u1max, r));
    } else {
      r = c.u_sub[first];
    }
    if (cu > 0.0 && fabs(r) < cu) r = 0.0;      // ALE_remap_tracers: conc_underflow, MOM_ALE.F90:812-814
    c.u0[i1 - 1] = r;
  }
  wsync();
}

// ======== the streaming form (round 4): one lane per column, no arrays ================================================================
// remapping_core_h for REMAPPING_SCHEME = PPM_H4 without boundary extrapolation, as ONE top-down walk per lane with a constant
// amount of state, for NF fields that share the column's two grids:
//   * the outer loop runs over the SOURCE cells (the same index in every lane: the loads of h_old and of the fields are coalesced
//     rows, issued a cell ahead of their use); the inner loop creates the sub-cells of that source cell exactly as
//     remap_via_sub_cells :519-651 does (the four cases of the merge);
//   * the reconstruction of a source cell (edge_values_explicit_h4 with end_value_h4 at the two ends, bound_edge_values,
//     check_discontinuous_edge_values, PPM_limiter_standard: MOM_remapping.F90:257-386) needs the cells m-2 .. m+2 only: a register
//     window; the weights of the edge values depend on the thicknesses alone and are formed once for the NF fields;
//   * h0_eff of a source cell (:560-620: the sum of its sub-cells' dh, which average_value_ppoly's xb divides by) is needed at its
//     first sub-cell: a look-ahead repeats the merge arithmetic on a copy of the two supplies until the cell closes;
//   * "adjust_thickest_subcell" (:703-719) rewrites uh_sub of the LAST thickest sub-cell of a source cell with the cell's total minus
//     the others, summed in order: two running sums (all sub-cells so far; all but the current candidate) give that in any case.
//     A (source, target) pair meets in exactly one sub-cell, so the candidate's target cell has either just been closed by that very
//     sub-cell -- then it waits, as the ONE pending target a lane can have, until the candidate moves on or the source cell closes --
//     or is still open when the source cell closes, and the adjusted value goes into its sum in the reference's order;
//   * a target cell is written when it is final (:721-766: the bounded mean of its sub-cells, or the first sub-cell's value of a
//     vanished cell; conc_underflow of ALE_remap_tracers): stores at the lane's own layer.
// The fields are remapped IN PLACE.  A lane has read the layers 0 .. m+3 when it works on source cell m; a target cell below that
// (the new grid far ahead of the old one: thin target cells in a thick source cell) would overwrite values the lane has yet to read,
// so such a value goes to a side array instead, the layer is marked in a 128-bit register mask, and the lane copies its marked layers
// back when it is done.  With z* grids a few time steps apart this never happens; the walk is correct for any pair of grids.
// The same operations on the same numbers in the same order as the wave-per-column kernel and the oracle: bit-identical results
// (tests/test_ale_gpu.py runs both forms).  nk >= 6; other schemes, boundary extrapolation and fewer layers use the wave kernel.
struct SRemapArgs {
  m6::GridDev g;
  const double *h_old, *h_new;
  double *const *fld;      // device array of field pointers, or null with `single`
  double *single;
  const double *cu;        // underflow per field or null
  int f0, pos;             // first field of this launch; staggering
  double h_neglect, h_neglect_edge;
  double *side;            // NF arrays of the fields' size: values of target cells that lie below what the lane has read so far
  long side_stride;        // doubles between the side arrays of two fields
};

struct EdgeW { double A1, A2, B, Cc, Hs; };
// the thickness-only factors of edge_values_explicit_h4 (regrid_edge_values.F90:262-300) at the interface above cell i
__device__ __forceinline__ EdgeW edge_weights_h4(double h0, double h1, double h2, double h3, double hNeglect) {
  const double hMinFrac = 1.e-5;
  if (h0 + h1 == 0.0 || h1 + h2 == 0.0 || h2 + h3 == 0.0) {
    const double h_min = hMinFrac * fmax(hNeglect, (h0 + h1) + (h2 + h3));
    h0 = fmax(h_min, h0); h1 = fmax(h_min, h1); h2 = fmax(h_min, h2); h3 = fmax(h_min, h3);
  }
  const double I_h12 = 1.0 / (h1 + h2);
  const double I_den_et2 = 1.0 / (((h0 + h1) + h2) * (h0 + h1)); const double I_h012 = (h0 + h1) * I_den_et2;
  const double I_den_et3 = 1.0 / ((h1 + (h2 + h3)) * (h2 + h3)); const double I_h123 = (h2 + h3) * I_den_et3;
  EdgeW w;
  w.A1 = (1.0 + (h1 * I_h012 + (h0 + h1) * I_h123)) * I_h12 * (h2 * (h2 + h3));
  w.A2 = (1.0 + (h2 * I_h123 + (h2 + h3) * I_h012)) * I_h12 * (h1 * (h0 + h1));
  w.B = (h1 * (h2 * (h2 + h3)) * I_den_et2);
  w.Cc = (h2 * (h1 * (h0 + h1)) * I_den_et3);
  w.Hs = (h0 + h1) + (h2 + h3);
  return w;
}
__device__ __forceinline__ double edge_value_h4(const EdgeW &w, double um2, double um1, double u0, double up1) {
  const double et1 = w.A1 * um1 + w.A2 * u0;
  const double et2 = w.B * (um1 - um2);
  const double et3 = w.Cc * (u0 - up1);
  return (et1 + (et2 + et3)) / w.Hs;
}
// the slope of bound_edge_values (regrid_edge_values.F90:71-82) with hr = h(k) / ((h(km1) + h(kp1)) + 2 h(k)) or a negative hr for "no slope"
__device__ __forceinline__ double bound_slope(double ukm1, double uk, double ukp1, double hr) {
  double slope_x_h = 0.0;
  if (hr >= 0.0) {
    const double sigma_l = (uk - ukm1);
    const double sigma_c = (ukp1 - ukm1) * hr;
    const double sigma_r = (ukp1 - uk);
    if ((sigma_l * sigma_r) > 0.0) slope_x_h = fsign(min3(fabs(sigma_l), fabs(sigma_c), fabs(sigma_r)), sigma_c);
  }
  return slope_x_h;
}
__device__ __forceinline__ double bound_left(double EL, double ukm1, double uk, double slope) {
  if ((ukm1 - EL) * (EL - uk) < 0.0) EL = uk - fsign(fmin(fabs(slope), fabs(EL - uk)), slope);
  return fmax(fmin(EL, fmax(ukm1, uk)), fmin(ukm1, uk));
}
__device__ __forceinline__ double bound_right(double ER, double ukp1, double uk, double slope) {
  if ((ukp1 - ER) * (ER - uk) < 0.0) ER = uk + fsign(fmin(fabs(slope), fabs(ER - uk)), slope);
  return fmax(fmin(ER, fmax(ukp1, uk)), fmin(ukp1, uk));
}
// PPM_limiter_standard for an interior cell (PPM_functions.F90:84-121)
__device__ __forceinline__ void ppm_limit_cell(double u_l, double u_c, double u_r, double &edge_l, double &edge_r) {
  if ((u_r - u_c) * (u_c - u_l) <= 0.0) {
    edge_l = u_c; edge_r = u_c;
  } else {
    const double expr1 = 3.0 * (edge_r - edge_l) * ((u_c - edge_l) + (u_c - edge_r));
    const double expr2 = (edge_r - edge_l) * (edge_r - edge_l);
    if (expr1 > expr2) {
      edge_l = u_c + 2.0 * (u_c - edge_r);
      edge_l = fmax(fmin(edge_l, fmax(u_l, u_c)), fmin(u_l, u_c));
    } else if (expr1 < -expr2) {
      edge_r = u_c + 2.0 * (u_c - edge_l);
      edge_r = fmax(fmin(edge_r, fmax(u_r, u_c)), fmin(u_r, u_c));
    }
  }
  if (fabs(edge_r - edge_l) < fmax(1.e-60, DBL_EPSILON * fabs(u_c))) { edge_l = u_c; edge_r = u_c; }
}
// average_value_ppoly for INT_PPM (MOM_remapping.F90:998-1099)
__device__ __forceinline__ double average_ppm(double a_L, double a_R, double u_c, double xa, double xb) {
  if (xb > xa) {
    const double mx = 0.5 * (xa + xb);
    const double a_c = 0.5 * ((u_c - a_L) + (u_c - a_R));
    if (mx < 0.5) {
      const double xa2b2ab = (xa * xa + xb * xb) + xa * xb;
      return a_L + ((a_R - a_L) * mx + a_c * (3. * (xb + xa) - 2. * xa2b2ab));
    } else {
      const double Ya = 1. - xa, Yb = 1. - xb;
      const double my = 0.5 * (Ya + Yb);
      const double Ya2b2ab = (Ya * Ya + Yb * Yb) + Ya * Yb;
      return a_R + ((a_L - a_R) * my + a_c * (3. * (Yb + Ya) - 2. * Ya2b2ab));
    }
  }
  const double Ya = 1. - xa;
  const double a_c = 3. * ((u_c - a_L) + (u_c - a_R));
  if (xa < 0.5) return a_L + xa * ((a_R - a_L) + a_c * Ya);
  return a_R + Ya * ((a_L - a_R) + a_c * xa);
}

// waves per SIMD the register allocation aims at: one field fits 128 VGPRs (4 waves: 13.7 / 6.0 ms for 4 tracers / u and v at
// 1440x1080x75 against 15.2 / 6.6 with 3), two fields need 168 (3 waves: 10.0 ms for 4 tracers; 19 ms when squeezed into 128); four
// fields a launch spill 268 registers (31 ms) -- profiles/r04_experiments.txt
template <int NF>
__global__ __launch_bounds__(64, NF == 1 ? 4 : 3) void ale_remap_stream_kernel(SRemapArgs a) {
  const m6::GridDev &g = a.g;
  const int n = g.nk;      // n0 = n1 = n
  const int xs = (a.pos == MOM6HIP_POS_U) ? 1 : 0, ys = (a.pos == MOM6HIP_POS_V) ? 1 : 0;
  const int i = g.isc - xs + blockIdx.x * 64 + threadIdx.x, j = g.jsc - ys + blockIdx.y;
  if (i > g.iec) return;
  const long base = a.pos == MOM6HIP_POS_U ? g.u2(i, j) : (a.pos == MOM6HIP_POS_V ? g.v2(i, j) : g.h2(i, j));
  const double msk = a.pos == MOM6HIP_POS_U ? g.mask2dCu[base] : (a.pos == MOM6HIP_POS_V ? g.mask2dCv[base] : g.mask2dT[base]);
  if (!(msk > 0.)) return;
  const long stride = (long)(g.nih + xs) * (g.njh + ys);
  const double *h0p = a.h_old + base, *h1p = a.h_new + base;
  double *fp[NF];
  double cu[NF];
#pragma unroll
  for (int f = 0; f < NF; f++) {
    fp[f] = (a.fld ? a.fld[a.f0 + f] : a.single) + base;
    cu[f] = a.cu ? a.cu[a.f0 + f] : 0.0;
  }
  auto H0 = [&](int k) -> double { return (k >= 0 && k < n) ? h0p[k * stride] : 0.0; };
  auto H1 = [&](int k) -> double { return (k >= 0 && k < n) ? h1p[k * stride] : 0.0; };
  const double hNe = a.h_neglect_edge;
  const int ns = 2 * n + 1;

  // ---- the two ends of edge_values_explicit_h4 (:329-362): V[0], V[1] from the first four cells, V[n-1], V[n] from the last four ----
  double Vt0[NF], Vt1[NF], Vb1[NF], Vb0[NF];
  {
    double dz[4], ut[4], C[4];
#pragma unroll
    for (int q = 0; q < 4; q++) dz[q] = fmax(hNe, H0(q));
#pragma unroll
    for (int f = 0; f < NF; f++) {
#pragma unroll
      for (int q = 0; q < 4; q++) ut[q] = fp[f][q * stride];
      end_value_h4(dz, ut, C);
      Vt0[f] = C[0];
      Vt1[f] = C[0] + dz[0] * (C[1] + dz[0] * (C[2] + dz[0] * C[3]));
    }
#pragma unroll
    for (int q = 0; q < 4; q++) dz[q] = fmax(hNe, H0(n - 1 - q));
#pragma unroll
    for (int f = 0; f < NF; f++) {
#pragma unroll
      for (int q = 0; q < 4; q++) ut[q] = fp[f][(long)(n - 1 - q) * stride];
      end_value_h4(dz, ut, C);
      Vb0[f] = C[0];
      Vb1[f] = C[0] + dz[0] * (C[1] + dz[0] * (C[2] + dz[0] * C[3]));
    }
  }

  // ---- windows: thicknesses and values of the cells m-1 .. m+2 (and m+3 on its way) ----
  double hm1 = 0., hc = H0(0), hp1 = H0(1), hp2 = H0(2), hp3 = H0(3);
  double um1[NF], uc[NF], up1[NF], up2[NF], up3[NF];
#pragma unroll
  for (int f = 0; f < NF; f++) { um1[f] = 0.; uc[f] = fp[f][0]; up1[f] = fp[f][stride]; up2[f] = fp[f][2 * stride]; up3[f] = fp[f][3 * stride]; }
  double elC[NF], slope_c[NF];      // the left edge of cell m after bound + check; the bound slope of cell m
  {
    // cell 0: km1 = 0, kp1 = 1
    const double den = (hc + hp1) + 2.0 * hc;
    const double hr = den > 0.0 ? hc / den : -1.0;
#pragma unroll
    for (int f = 0; f < NF; f++) { slope_c[f] = bound_slope(uc[f], uc[f], up1[f], hr); elC[f] = bound_left(Vt0[f], uc[f], uc[f], slope_c[f]); }
  }

  // ---- the merge state (remap_via_sub_cells :519-560) ----
  int i0 = 1, i1 = 1, i_sub = 1;
  double h0s = hc, h1s = H1(0);
  double h1_open = h1s;                 // h1 of the open target cell
  bool src = true, tgt = true;
  double dh_max = 0.;
  // the second loop's state (:653-700)
  double xa = 0., dh0_eff_i = 0.;
  // the open target cell
  double dh_t = 0.;
  double duh_t[NF], umin_t[NF], umax_t[NF], ufirst_t[NF];
  // the sums of the open source cell: all sub-cells so far, and all but the candidate for the thickest
  double S_incl[NF], S_excl[NF], cand_uh[NF];
  double cand_hsub = 0.;
  // the pending target cell (closed by the candidate sub-cell of the open source cell)
  bool pend = false;
  int pend_i1 = 0;
  bool pend_thick = false;
  double pend_dh = 0.;
  double pend_pre[NF], pend_min[NF], pend_max[NF], pend_first[NF];
  double aL[NF], aR[NF];
#pragma unroll
  for (int f = 0; f < NF; f++) { duh_t[f] = 0.; S_incl[f] = 0.; S_excl[f] = 0.; cand_uh[f] = 0.; aL[f] = uc[f]; aR[f] = uc[f]; }

  int m_read = 3;      // the lane has read the layers 0 .. m_read
  unsigned long long ahead_lo = 0ull, ahead_hi = 0ull;      // layers whose new value waits in the side array
  auto store_target = [&](int t1, bool thick, double dh, const double *duh, const double *mn, const double *mx, const double *first) {
    const int kt = t1 - 1;
    const bool ahead = kt > m_read;
    if (ahead) { if (kt < 64) ahead_lo |= 1ull << kt; else ahead_hi |= 1ull << (kt - 64); }
#pragma unroll
    for (int f = 0; f < NF; f++) {
      double r;
      if (thick) { r = duh[f] / dh; r = fmax(mn[f], fmin(mx[f], r)); }
      else r = first[f];
      if (cu[f] > 0.0 && fabs(r) < cu[f]) r = 0.0;      // ALE_remap_tracers: conc_underflow, MOM_ALE.F90:812-814
      if (ahead) a.side[a.side_stride * f + base + (long)kt * stride] = r;
      else fp[f][(long)kt * stride] = r;
    }
  };
  auto finalize_pending = [&](const double *uh_last) {      // the pending target with the (adjusted or not) value of its last sub-cell
    double duh[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) duh[f] = pend_pre[f] + uh_last[f];
    store_target(pend_i1, pend_thick, pend_dh, duh, pend_min, pend_max, pend_first);
    pend = false;
  };

  // sub-cell 1 (:531-536, :655-656): no thickness, the left edge of the first cell
  // (cell 0 of PPM_H4 without extrapolation is piecewise constant: E(1,1) = u0(1))
#pragma unroll
  for (int f = 0; f < NF; f++) { umin_t[f] = uc[f]; umax_t[f] = uc[f]; ufirst_t[f] = uc[f]; }
  bool first_of_target = false;

  double h0_eff_cur = 0.;
  double hcell = hc;      // h0 of the open source cell
  double ucell[NF];
#pragma unroll
  for (int f = 0; f < NF; f++) ucell[f] = uc[f];

  // one sub-cell of the walk; returns true when it closed the open source cell
  auto sub_cell = [&]() -> bool {
    i_sub++;
    const double dh = fmin(h0s, h1s);
    const bool in_src = src, in_tgt = tgt;
    const bool is_cand = in_src && (dh >= dh_max);
    if (dh >= dh_max) dh_max = dh;
    double hsub = dh;
    bool end_src = false, end_tgt = false;
    if (h0s <= h1s && src) { h1s = h1s - dh; end_src = true; }
    else if (h0s >= h1s && tgt) { h0s = h0s - dh; end_tgt = true; }
    else if (src) { hsub = h0s; end_src = true; }
    else if (tgt) { hsub = h1s; end_tgt = true; }
    // the value of the sub-cell (:657-690; the last one :691-692)
    double usub[NF], uh[NF];
    double xb = 1.;
    if (i_sub < ns) {
      dh0_eff_i = dh0_eff_i + hsub;
      if (h0_eff_cur > 0.) {
        xb = dh0_eff_i / h0_eff_cur;
        xb = fmin(1., xb);
#pragma unroll
        for (int f = 0; f < NF; f++) usub[f] = average_ppm(aL[f], aR[f], ucell[f], xa, xb);
      } else {
#pragma unroll
        for (int f = 0; f < NF; f++) usub[f] = ucell[f];
      }
#pragma unroll
      for (int f = 0; f < NF; f++) uh[f] = hsub * usub[f];
    } else {
#pragma unroll
      for (int f = 0; f < NF; f++) { usub[f] = aR[f]; uh[f] = aR[f] * hsub; }
    }
    // the thickest sub-cell of the open source cell (:703-719)
    double uh_here[NF];      // what this sub-cell contributes to its target's sum
#pragma unroll
    for (int f = 0; f < NF; f++) uh_here[f] = uh[f];
    bool make_pending = false;
    if (in_src) {
      if (is_cand) {
        if (pend) finalize_pending(cand_uh);      // the previous candidate stays as it was
#pragma unroll
        for (int f = 0; f < NF; f++) { S_excl[f] = S_incl[f]; S_incl[f] = S_incl[f] + uh[f]; cand_uh[f] = uh[f]; }
        cand_hsub = hsub;
        if (end_src) {
          if (hsub > 0.) {
#pragma unroll
            for (int f = 0; f < NF; f++) uh_here[f] = ucell[f] * hcell - S_excl[f];
          }
        } else {
          make_pending = true;      // closes its target while the source cell goes on
        }
      } else {
#pragma unroll
        for (int f = 0; f < NF; f++) { S_excl[f] = S_excl[f] + uh[f]; S_incl[f] = S_incl[f] + uh[f]; }
        if (end_src && pend) {
          double adj[NF];
#pragma unroll
          for (int f = 0; f < NF; f++) adj[f] = (cand_hsub > 0.) ? (ucell[f] * hcell - S_excl[f]) : cand_uh[f];
          finalize_pending(adj);
        }
      }
    }
    // the open target cell (:721-766)
    if (in_tgt) {
      if (first_of_target) {
#pragma unroll
        for (int f = 0; f < NF; f++) { umin_t[f] = usub[f]; umax_t[f] = usub[f]; ufirst_t[f] = usub[f]; }
        first_of_target = false;
      }
      dh_t = dh_t + hsub;
#pragma unroll
      for (int f = 0; f < NF; f++) { umin_t[f] = fmin(umin_t[f], usub[f]); umax_t[f] = fmax(umax_t[f], usub[f]); }
      if (end_tgt) {
        const bool thick = h1_open > 0.;
        if (make_pending) {
          pend = true; pend_i1 = i1; pend_thick = thick; pend_dh = dh_t;
#pragma unroll
          for (int f = 0; f < NF; f++) { pend_pre[f] = duh_t[f]; pend_min[f] = umin_t[f]; pend_max[f] = umax_t[f]; pend_first[f] = ufirst_t[f]; }
        } else {
#pragma unroll
          for (int f = 0; f < NF; f++) duh_t[f] = duh_t[f] + uh_here[f];
          store_target(i1, thick, dh_t, duh_t, umin_t, umax_t, ufirst_t);
        }
        dh_t = 0.;
#pragma unroll
        for (int f = 0; f < NF; f++) duh_t[f] = 0.;
        first_of_target = true;
        if (i1 < n) { i1 = i1 + 1; h1s = H1(i1 - 1); h1_open = h1s; }
        else { h1s = 0.; tgt = false; }
      } else {
#pragma unroll
        for (int f = 0; f < NF; f++) duh_t[f] = duh_t[f] + uh_here[f];
      }
    }
    // the next sub-cell's place in its source cell (:693-699)
    if (end_src) {
      dh_max = 0.;
#pragma unroll
      for (int f = 0; f < NF; f++) { S_incl[f] = 0.; S_excl[f] = 0.; }
      if (i0 < n) { dh0_eff_i = 0.; xa = 0.; }
      else { xa = xb; h0s = 0.; src = false; }
    } else {
      xa = xb;
    }
    return end_src;
  };

  // sub-cell 1 belongs to source cell 1 and target cell 1 with no thickness and uh = 0 (:531-536, :655): S = 0 + 0, dh = 0 + 0
  for (int m = 0; m < n; m++) {
    // ---- the reconstruction of cell m (edge value at its lower interface, bounds, discontinuity check, limiter) ----
    {
      const int K = m + 1;      // the interface below cell m
      double Vn[NF];
      if (K >= 2 && K <= n - 2) {
        const EdgeW w = edge_weights_h4(hm1, hc, hp1, hp2, hNe);
#pragma unroll
        for (int f = 0; f < NF; f++) Vn[f] = edge_value_h4(w, um1[f], uc[f], up1[f], up2[f]);
      } else {
#pragma unroll
        for (int f = 0; f < NF; f++) Vn[f] = (K == 1) ? Vt1[f] : ((K == n - 1) ? Vb1[f] : Vb0[f]);
      }
      double hr_n = -1.0;
      if (m + 1 <= n - 1) {      // cell m+1: km1 = m, kp1 = min(m+2, n-1)
        const double hk = hp1, hkp = (m + 2 <= n - 1) ? hp2 : hp1;
        const double den = (hc + hkp) + 2.0 * hk;
        if (den > 0.0) hr_n = hk / den;
      }
#pragma unroll
      for (int f = 0; f < NF; f++) {
        const double ukp1 = (m + 1 <= n - 1) ? up1[f] : uc[f];
        double erB = bound_right(Vn[f], ukp1, uc[f], slope_c[f]);
        double el_next = 0., slope_n = 0.;
        if (m + 1 <= n - 1) {
          const double ukp2 = (m + 2 <= n - 1) ? up2[f] : up1[f];
          slope_n = bound_slope(uc[f], up1[f], ukp2, hr_n);
          el_next = bound_left(Vn[f], uc[f], up1[f], slope_n);
          if ((el_next - erB) * (up1[f] - uc[f]) < 0.0) {      // check_discontinuous_edge_values :141-159
            double u0_avg = 0.5 * (erB + el_next);
            u0_avg = fmax(fmin(u0_avg, fmax(uc[f], up1[f])), fmin(uc[f], up1[f]));
            erB = u0_avg; el_next = u0_avg;
          }
        }
        double eL = elC[f], eR = erB;
        if (m == 0 || m == n - 1) { eL = uc[f]; eR = uc[f]; }
        else ppm_limit_cell(um1[f], uc[f], up1[f], eL, eR);
#if defined(SR_EXP) && SR_EXP == 2
        eL = uc[f]; eR = uc[f];      // (experiment: no reconstruction)
#endif
        aL[f] = eL; aR[f] = eR; ucell[f] = uc[f];
        elC[f] = el_next; slope_c[f] = slope_n;
      }
      hcell = hc;
    }
    // ---- h0_eff of this source cell (:560-620) by a look-ahead on copies of the supplies ----
    {
      double eff = 0., s0 = h0s, s1 = h1s;
      bool t = tgt;
      int jj = i1;
      for (int it = 0; it < ns; it++) {
        const double d = fmin(s0, s1);
        eff = eff + fmin(d, s0);
        if (s0 <= s1) break;
        if (s0 >= s1 && t) {
          s0 = s0 - d;
          if (jj < n) { jj = jj + 1; s1 = H1(jj - 1); }      // (a register window of the next thicknesses was measured: 8 % slower)
          else { s1 = 0.; t = false; }
        } else break;
      }
      h0_eff_cur = eff;
#if defined(SR_EXP) && SR_EXP == 1
      h0_eff_cur = hcell;      // (experiment: no look-ahead)
#endif
    }
    // ---- the sub-cells of this source cell ----
    for (int it = 0; it < ns && i_sub < ns; it++)
      if (sub_cell()) break;
    // ---- advance the windows ----
    m_read = m + 4;
    if (m + 1 < n) { i0 = m + 2; h0s = hp1; }
    hm1 = hc; hc = hp1; hp1 = hp2; hp2 = hp3; hp3 = H0(m + 4);
#pragma unroll
    for (int f = 0; f < NF; f++) {
      um1[f] = uc[f]; uc[f] = up1[f]; up1[f] = up2[f]; up2[f] = up3[f];
      up3[f] = (m + 4 < n) ? fp[f][(long)(m + 4) * stride] : 0.0;
    }
  }
  // ---- what is left of the target grid below the last source cell (:621-640) ----
  for (int it = 0; it < ns && i_sub < ns; it++) sub_cell();
  if (pend) finalize_pending(cand_uh);
  // the layers that were written ahead of the reads
  while (ahead_lo | ahead_hi) {
    int kt;
    if (ahead_lo) { kt = __ffsll((long long)ahead_lo) - 1; ahead_lo &= ahead_lo - 1; }
    else { kt = 64 + __ffsll((long long)ahead_hi) - 1; ahead_hi &= ahead_hi - 1; }
#pragma unroll
    for (int f = 0; f < NF; f++) fp[f][(long)kt * stride] = a.side[a.side_stride * f + base + (long)kt * stride];
  }
}

struct WRemapArgs {
  m6::GridDev g;
  const double *h_old, *h_new;   // thicknesses at the points of the fields (h, u or v points)
  double *const *fld;            // device array of nfld pointers, or
  double *single;                // the one field (nfld = 1) when fld is null
  const double *cu;              // underflow per field or null
  const char *sub;               // structure tiles from ale_sub_cells_kernel
  int nfld, scheme, extrap, pos; // pos: MOM6HIP_POS_H / _U / _V
  double h_neglect, h_neglect_edge;
};

__global__ __launch_bounds__(64 * WR_NCOL) void ale_remap_wave_kernel(WRemapArgs a) {
  extern __shared__ char wsm[];
  const m6::GridDev &g = a.g;
  const int nz = g.nk;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int xs = (a.pos == MOM6HIP_POS_U) ? 1 : 0, ys = (a.pos == MOM6HIP_POS_V) ? 1 : 0;
  const int i_first = g.isc - xs + blockIdx.x * WR_NCOL, j = g.jsc - ys + blockIdx.y;
  const long stride = (long)(g.nih + xs) * (g.njh + ys);
  auto col2 = [&](int i) -> long { return a.pos == MOM6HIP_POS_U ? g.u2(i, j) : (a.pos == MOM6HIP_POS_V ? g.v2(i, j) : g.h2(i, j)); };
  auto active = [&](int i) -> bool {
    if (i > g.iec) return false;
    const long n2 = col2(i);
    const double m = a.pos == MOM6HIP_POS_U ? g.mask2dCu[n2] : (a.pos == MOM6HIP_POS_V ? g.mask2dCv[n2] : g.mask2dT[n2]);
    return m > 0.;
  };
  const int xd = wcol_extra(a.scheme, nz);
  const size_t cb = wcol_bytes(nz, xd);
  const WCol c = wcol_at(wsm + cb * wv, nz, xd);
  const bool mine = active(i_first + wv);
  const int nd = sub_drows(nz), nsr = sub_srows(nz);

  // the structure tile: contiguous rows of WR_NCOL entries; entry (row, column) goes to that column's LDS block, whose
  // leading arrays have the order of the tile's rows  (tiles of inactive columns hold garbage, never used)
  {
    const char *tile = a.sub + sub_tile_bytes(nz) * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
    const double *td = (const double *)tile;
    for (int t = threadIdx.x; t < nd * WR_NCOL; t += 64 * WR_NCOL)
      ((double *)(wsm + cb * (t % WR_NCOL)))[t / WR_NCOL] = td[t];
    const short *ts = (const short *)(td + (size_t)nd * WR_NCOL);
    const size_t soff = wcol_doubles(nz, xd) * 8;
    for (int t = threadIdx.x; t < nsr * WR_NCOL; t += 64 * WR_NCOL)
      ((short *)(wsm + cb * (t % WR_NCOL) + soff))[t / WR_NCOL] = ts[t];
  }
  // cooperative, row-contiguous load of a (k, column) tile: thread t handles column t % NCOL, layers t / NCOL, ...
  auto tile_load = [&](const double *src, int slot) {     // slot: 0 h0, 1 u0
    for (int t = threadIdx.x; t < nz * WR_NCOL; t += 64 * WR_NCOL) {
      const int cidx = t % WR_NCOL, k = t / WR_NCOL;
      const int i = i_first + cidx;
      if (i <= g.iec) ((double *)(wsm + cb * cidx))[nd + slot * nz + k] = src[col2(i) + stride * k];
    }
  };
  tile_load(a.h_old, 0);
  for (int m = 0; m < a.nfld; m++) {
    double *f = a.fld ? a.fld[m] : a.single;
    tile_load(f, 1);
    __syncthreads();
    if (mine) {
      const int method = w_build_reconstructions(c, lane, a.scheme, a.extrap != 0, nz, a.h_neglect, a.h_neglect_edge);
      w_integrate_sub_cells(c, lane, nz, nz, method, a.cu ? a.cu[m] : 0.0);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < nz * WR_NCOL; t += 64 * WR_NCOL) {
      const int cidx = t % WR_NCOL, k = t / WR_NCOL;
      const int i = i_first + cidx;
      if (active(i)) f[col2(i) + stride * k] = ((const double *)(wsm + cb * cidx))[nd + nz + k];
    }
    __syncthreads();
  }
}

bool scheme_provided(int scheme) {
  if (scheme == REMAP_PCM || scheme == REMAP_PLM || scheme == REMAP_PPM_H4) return true;
  return (scheme == REMAP_PPM_IH4 || scheme == REMAP_PPM_CW || scheme == REMAP_PLM_HYBGEN || scheme == REMAP_PPM_HYBGEN ||
          scheme == REMAP_WENO_HYBGEN || scheme == REMAP_PQM_IH4IH3 || scheme == REMAP_PQM_IH6IH5);
}

// MOM6HIP_ALE_STREAM: 0 = never the streaming kernel; 1 = one field a launch; 2 (default) = two fields a launch (read at every
// call: the tests switch between the forms)
int stream_fields() { const char *e = getenv("MOM6HIP_ALE_STREAM"); return e ? atoi(e) : 2; }

// the two launches of the wave-cooperative path over the compute range of the given staggering -- or, for PPM_H4 without boundary
// extrapolation on at least 6 layers, the streaming kernel (fields two at a time)
int launch_wave_remap(mom6hip_ctx_t *ctx, WRemapArgs a) {
  const m6::GridDev &g = a.g;
  const int xs = (a.pos == MOM6HIP_POS_U) ? 1 : 0, ys = (a.pos == MOM6HIP_POS_V) ? 1 : 0;
  const int nf_stream = stream_fields();
  if (nf_stream > 0 && a.scheme == REMAP_PPM_H4 && !a.extrap && g.nk >= 6) {
    const size_t fbytes = sizeof(double) * (size_t)(g.nih + 1) * (g.njh + 1) * g.nk;      // (covers every staggering)
    M6_REQUIRE(ctx->ale_side.reserve(2 * fbytes) == 0, "ALE remap: out of device memory");
    SRemapArgs sa;
    sa.g = g; sa.h_old = a.h_old; sa.h_new = a.h_new; sa.fld = a.fld; sa.single = a.single; sa.cu = a.cu; sa.pos = a.pos;
    sa.h_neglect = a.h_neglect; sa.h_neglect_edge = a.h_neglect_edge; sa.side = (double *)ctx->ale_side.p; sa.side_stride = (long)(fbytes / 8);
    const int ncol = g.iec - g.isc + 1 + xs, nrow = g.jec - g.jsc + 1 + ys;
    const dim3 grid((ncol + 63) / 64, nrow);
    int f = 0;
    while (f < a.nfld) {
      sa.f0 = f;
      if (nf_stream >= 2 && a.nfld - f >= 2 && a.fld) { hipLaunchKernelGGL(ale_remap_stream_kernel<2>, grid, dim3(64), 0, ctx->stream, sa); f += 2; }
      else { hipLaunchKernelGGL(ale_remap_stream_kernel<1>, grid, dim3(64), 0, ctx->stream, sa); f += 1; }
    }
    M6_HIP(hipGetLastError());
    return 0;
  }
  const size_t lds = wcol_bytes(g.nk, wcol_extra(a.scheme, g.nk)) * WR_NCOL;
  M6_REQUIRE(lds <= 160 * 1024, "ALE remap: too many layers for the LDS-resident kernel");
  std::vector<const void *> &configured = ctx->lds_configured;      // the attribute is per device: kept with the context
  if (std::find(configured.begin(), configured.end(), (const void *)ale_remap_wave_kernel) == configured.end()) {
    M6_HIP(hipFuncSetAttribute((const void *)ale_remap_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    configured.push_back((const void *)ale_remap_wave_kernel);
  }
  const int ncol = g.iec - g.isc + 1 + xs, nrow = g.jec - g.jsc + 1 + ys;
  const int nbx = (ncol + WR_NCOL - 1) / WR_NCOL;
  M6_REQUIRE(ctx->ale_sub.reserve(sub_tile_bytes(g.nk) * (size_t)nbx * nrow) == 0, "ALE remap: out of device memory");
  SubArgs sa;
  sa.g = g; sa.h_old = a.h_old; sa.h_new = a.h_new; sa.sub = (char *)ctx->ale_sub.p; sa.pos = a.pos; sa.nbx = nbx;
  hipLaunchKernelGGL(ale_sub_cells_kernel, dim3((ncol + 63) / 64, nrow), dim3(64), 0, ctx->stream, sa);
  a.sub = (const char *)ctx->ale_sub.p;
  hipLaunchKernelGGL(ale_remap_wave_kernel, dim3(nbx, nrow), dim3(64 * WR_NCOL), lds, ctx->stream, a);
  M6_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// ALE_remap_tracers(CS, G, GV, h_old, h_new, Reg, debug, dt, PCM_cell), src/ALE/MOM_ALE.F90:737.
extern "C" int mom6hip_ale_remap_tracers(mom6hip_ctx_t *ctx, const mom6hip_remapping_cs_t *cs, const double *h_old,
                                         const double *h_new, double *const *tr, const double *conc_underflow,
                                         int32_t ntr, int32_t memspace) {
  M6_REQUIRE(ctx && cs && h_old && h_new, "ALE_remap_tracers: null argument");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "ALE_remap_tracers: bad memspace");
  if (ntr <= 0) return 0;
  M6_REQUIRE(tr != nullptr && ntr <= 64, "ALE_remap_tracers: bad tracer list");
  M6_REQUIRE(scheme_provided(cs->remapping_scheme),
             "MOM_remapping, build_reconstructions_1d: The selected remapping method is invalid "
             "(libmom6hip provides PCM, PLM, PLM_HYBGEN, PPM_H4, PPM_IH4, PPM_HYBGEN, WENO_HYBGEN, PPM_CW, PQM_IH4IH3 and PQM_IH6IH5)");
  // (edge_values_implicit_h6 and edge_slopes_implicit_h5 read the six cells at either end of a column, whatever its length)
  M6_REQUIRE(cs->remapping_scheme != REMAP_PQM_IH6IH5 || ctx->g.nk >= 6 || ctx->g.nk <= 4, "PQM_IH6IH5 needs at least six layers (or at most four, where build_reconstructions_1d takes a lower scheme)");
  M6_REQUIRE(cs->answer_date >= 20190101, "ALE_remap_tracers: only REMAPPING_ANSWER_DATE >= 20190101 is provided");
  M6_REQUIRE(!cs->force_bounds_in_subcell, "ALE_remap_tracers: REMAP_BOUND_INTERMEDIATE_VALUES is not provided");
  M6_REQUIRE(ctx->g.mask2dT != nullptr, "ALE_remap_tracers: mask2dT is required");
  m6::GridDev &g = ctx->g;
  hipStream_t s = ctx->stream;
  M6_REQUIRE(g.nk <= 128, "ALE_remap_tracers: at most 128 layers are supported");
  const size_t bH = (size_t)g.nh3() * 8;

  const double *d_hold = h_old, *d_hnew = h_new;
  std::vector<double *> d_tr(ntr);
  if (memspace == MOM6HIP_MEM_HOST) {
    if (ctx->stage[0].reserve(bH) || ctx->stage[1].reserve(bH)) return 1;
    M6_HIP(hipMemcpyAsync(ctx->stage[0].p, h_old, bH, hipMemcpyHostToDevice, s));
    M6_HIP(hipMemcpyAsync(ctx->stage[1].p, h_new, bH, hipMemcpyHostToDevice, s));
    d_hold = (double *)ctx->stage[0].p; d_hnew = (double *)ctx->stage[1].p;
    for (int m = 0; m < ntr; m++) {
      M6_REQUIRE(tr[m] != nullptr, "ALE_remap_tracers: tracer %d is null", m);
      if (ctx->tr_stage[m].reserve(bH)) return 1;
      M6_HIP(hipMemcpyAsync(ctx->tr_stage[m].p, tr[m], bH, hipMemcpyHostToDevice, s));
      d_tr[m] = (double *)ctx->tr_stage[m].p;
    }
  } else {
    for (int m = 0; m < ntr; m++) { M6_REQUIRE(tr[m] != nullptr, "ALE_remap_tracers: tracer %d is null", m); d_tr[m] = tr[m]; }
  }
  // pointer + underflow tables on the device
  if (ctx->stage[15].reserve(64 * (sizeof(double *) + sizeof(double)))) return 1;
  double **d_ptrs = (double **)ctx->stage[15].p;
  double *d_cu = (double *)(d_ptrs + 64);
  M6_HIP(hipMemcpyAsync(d_ptrs, d_tr.data(), ntr * sizeof(double *), hipMemcpyHostToDevice, s));
  std::vector<double> cu(ntr, 0.0);
  if (conc_underflow) for (int m = 0; m < ntr; m++) cu[m] = conc_underflow[m];
  M6_HIP(hipMemcpyAsync(d_cu, cu.data(), ntr * sizeof(double), hipMemcpyHostToDevice, s));
  M6_HIP(hipStreamSynchronize(s));   // the tables above live on the host stack

  {
    WRemapArgs w;
    w.g = g; w.h_old = d_hold; w.h_new = d_hnew; w.fld = d_ptrs; w.single = nullptr; w.cu = d_cu; w.nfld = ntr;
    w.scheme = cs->remapping_scheme; w.extrap = cs->boundary_extrapolation; w.pos = MOM6HIP_POS_H;
    w.h_neglect = g.H_subroundoff; w.h_neglect_edge = g.H_subroundoff;      // MOM_ALE.F90:770-771 (answer_date >= 20190101)
    if (launch_wave_remap(ctx, w)) return 1;
  }
  M6_HIP(hipGetLastError());
  if (memspace == MOM6HIP_MEM_HOST) {
    for (int m = 0; m < ntr; m++) M6_HIP(hipMemcpyAsync(tr[m], d_tr[m], bH, hipMemcpyDeviceToHost, s));
    M6_HIP(hipStreamSynchronize(s));
  }
  return 0;
}

// ALE_regrid(G, GV, US, h, h_new, dzRegrid, tv, CS, frac_shelf_h, PCM_cell), src/ALE/MOM_ALE.F90:484 (z*).
extern "C" int mom6hip_ale_regrid(mom6hip_ctx_t *ctx, const mom6hip_regridding_cs_t *cs, const double *h, double *h_new,
                                  double *dzRegrid, int32_t memspace) {
  M6_REQUIRE(ctx && cs && h && h_new && dzRegrid, "ALE_regrid: null argument");
  M6_REQUIRE(cs->regridding_scheme == MOM6HIP_REGRIDDING_ZSTAR,
             "MOM_regridding, regridding_main: only the z* regridding scheme is provided by libmom6hip");
  const m6::GridDev g = ctx->g;
  M6_REQUIRE(cs->nk == g.nk && cs->coordinateResolution, "ALE_regrid: CS%%nk must equal GV%%ke and coordinateResolution must be set");
  M6_REQUIRE(g.nk <= 128 && g.mask2dT && g.bathyT, "ALE_regrid: at most 128 layers; mask2dT and bathyT are needed");
  M6_REQUIRE(g.isc - 1 >= g.isd && g.jsc - 1 >= g.jsd, "ALE_regrid: a halo of at least 1 is needed");
  hipStream_t s = ctx->stream;
  const size_t b2 = sizeof(double) * (size_t)g.nih * g.njh, b3 = b2 * g.nk, b3i = b2 * (g.nk + 1);
  m6::Stager st(ctx, memspace);
  RegridArgs a;
  a.g = g; a.h = st.in(h, b3); a.h_new = st.inout(h_new, b3); a.dz = st.out(dzRegrid, b3i);
  double *res = (double *)st.scratch(sizeof(double) * g.nk);
  M6_REQUIRE(!st.failed() && res, "ALE_regrid: staging failed");
  M6_HIP(hipMemcpyAsync(res, cs->coordinateResolution, sizeof(double) * g.nk, hipMemcpyHostToDevice, s));
  M6_HIP(hipMemsetAsync(a.dz, 0, b3i, s));                                       // dzRegrid(:,:,:) = 0.0  (:508)
  a.res = res; a.min_thickness = cs->min_thickness; a.old_grid_weight = cs->old_grid_weight;
  a.zs = cs->depth_of_time_filter_shallow; a.zd = cs->depth_of_time_filter_deep; a.Z_ref = cs->Z_ref;
  dim3 grid((g.iec - g.isc + 3 + 63) / 64, g.jec - g.jsc + 3);
  if (g.nk <= 8) hipLaunchKernelGGL(ale_regrid_zstar_kernel<8>, grid, dim3(64), 0, s, a);
  else if (g.nk <= 32) hipLaunchKernelGGL(ale_regrid_zstar_kernel<32>, grid, dim3(64), 0, s, a);
  else if (g.nk <= 80) hipLaunchKernelGGL(ale_regrid_zstar_kernel<80>, grid, dim3(64), 0, s, a);
  else hipLaunchKernelGGL(ale_regrid_zstar_kernel<128>, grid, dim3(64), 0, s, a);
  M6_HIP(hipGetLastError());
  M6_HIP(hipStreamSynchronize(s));      // coordinateResolution was copied from the caller's host array
  return st.finish();
}

// ALE_remap_set_h_vel(CS, G, GV, h_new, h_u, h_v, OBC, debug), src/ALE/MOM_ALE.F90:870.
extern "C" int mom6hip_ale_remap_set_h_vel(mom6hip_ctx_t *ctx, const double *h_new, double *h_u, double *h_v, int32_t memspace) {
  M6_REQUIRE(ctx && h_new && h_u && h_v, "ALE_remap_set_h_vel: null argument");
  const m6::GridDev g = ctx->g;
  M6_REQUIRE(g.mask2dCu && g.mask2dCv, "ALE_remap_set_h_vel: mask2dCu and mask2dCv are needed");
  const size_t b3 = sizeof(double) * (size_t)g.nh3(), bu = sizeof(double) * (size_t)g.nu3(), bv = sizeof(double) * (size_t)g.nv3();
  m6::Stager st(ctx, memspace);
  const double *dh = st.in(h_new, b3);
  double *du = st.inout(h_u, bu), *dv = st.inout(h_v, bv);
  M6_REQUIRE(!st.failed(), "ALE_remap_set_h_vel: staging failed");
  hipLaunchKernelGGL(ale_set_h_vel_kernel, dim3((g.iec - g.isc + 2 + 63) / 64, g.jec - g.jsc + 2, g.nk), dim3(64), 0, ctx->stream, g, dh, du, dv);
  M6_HIP(hipGetLastError());
  return st.finish();
}

// ALE_remap_set_h_vel_via_dz(CS, G, GV, h_new, h_u, h_v, OBC, h_old, dzInterface, debug), src/ALE/MOM_ALE.F90:912.
extern "C" int mom6hip_ale_remap_set_h_vel_via_dz(mom6hip_ctx_t *ctx, const double *h_old, const double *dzInterface, double *h_u, double *h_v,
                                                  int32_t memspace) {
  M6_REQUIRE(ctx && h_old && dzInterface && h_u && h_v, "ALE_remap_set_h_vel_via_dz: null argument");
  const m6::GridDev g = ctx->g;
  M6_REQUIRE(g.mask2dCu && g.mask2dCv, "ALE_remap_set_h_vel_via_dz: mask2dCu and mask2dCv are needed");
  const size_t b3 = sizeof(double) * (size_t)g.nh3(), bu = sizeof(double) * (size_t)g.nu3(), bv = sizeof(double) * (size_t)g.nv3();
  const size_t bi = b3 + sizeof(double) * (size_t)g.nih * g.njh;
  m6::Stager st(ctx, memspace);
  const double *dh = st.in(h_old, b3), *dz = st.in(dzInterface, bi);
  double *du = st.inout(h_u, bu), *dv = st.inout(h_v, bv);
  M6_REQUIRE(!st.failed(), "ALE_remap_set_h_vel_via_dz: staging failed");
  hipLaunchKernelGGL(ale_set_h_vel_via_dz_kernel, dim3((g.iec - g.isc + 2 + 63) / 64, g.jec - g.jsc + 2, g.nk), dim3(64), 0, ctx->stream, g, dh, dz,
                     du, dv);
  M6_HIP(hipGetLastError());
  return st.finish();
}

// ALE_remap_velocities(CS, G, GV, h_old_u, h_old_v, h_new_u, h_new_v, u, v, ...), src/ALE/MOM_ALE.F90:1061.
extern "C" int mom6hip_ale_remap_velocities(mom6hip_ctx_t *ctx, const mom6hip_remapping_cs_t *cs, const double *h_old_u,
                                            const double *h_old_v, const double *h_new_u, const double *h_new_v, double *u,
                                            double *v, int32_t memspace) {
  M6_REQUIRE(ctx && cs && h_old_u && h_old_v && h_new_u && h_new_v && u && v, "ALE_remap_velocities: null argument");
  M6_REQUIRE(scheme_provided(cs->remapping_scheme),
             "MOM_remapping, build_reconstructions_1d: The selected remapping method is invalid "
             "(libmom6hip provides PCM, PLM, PLM_HYBGEN, PPM_H4, PPM_IH4, PPM_HYBGEN, WENO_HYBGEN, PPM_CW, PQM_IH4IH3 and PQM_IH6IH5)");
  // (edge_values_implicit_h6 and edge_slopes_implicit_h5 read the six cells at either end of a column, whatever its length)
  M6_REQUIRE(cs->remapping_scheme != REMAP_PQM_IH6IH5 || ctx->g.nk >= 6 || ctx->g.nk <= 4, "PQM_IH6IH5 needs at least six layers (or at most four, where build_reconstructions_1d takes a lower scheme)");
  M6_REQUIRE(cs->answer_date >= 20190101 && !cs->force_bounds_in_subcell, "ALE_remap_velocities: unsupported remapping options");
  const m6::GridDev g = ctx->g;
  M6_REQUIRE(g.nk <= 128 && g.mask2dCu && g.mask2dCv, "ALE_remap_velocities: at most 128 layers; face masks are needed");
  const size_t bu = sizeof(double) * (size_t)g.nu3(), bv = sizeof(double) * (size_t)g.nv3();
  m6::Stager st(ctx, memspace);
  const double *ho[2] = {st.in(h_old_u, bu), st.in(h_old_v, bv)}, *hn[2] = {st.in(h_new_u, bu), st.in(h_new_v, bv)};
  double *vel[2] = {st.inout(u, bu), st.inout(v, bv)};
  M6_REQUIRE(!st.failed(), "ALE_remap_velocities: staging failed");
  for (int d = 0; d < 2; d++) {
    WRemapArgs w;
    w.g = g; w.h_old = ho[d]; w.h_new = hn[d]; w.fld = nullptr; w.single = vel[d]; w.cu = nullptr; w.nfld = 1;
    w.scheme = cs->remapping_scheme; w.extrap = cs->boundary_extrapolation; w.pos = d ? MOM6HIP_POS_V : MOM6HIP_POS_U;
    w.h_neglect = g.H_subroundoff; w.h_neglect_edge = g.H_subroundoff;      // :1118-1119
    if (launch_wave_remap(ctx, w)) return 1;
  }
  M6_HIP(hipGetLastError());
  return st.finish();
}
