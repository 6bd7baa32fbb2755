// tracer_advect.hip -- advect_tracer / advect_x / advect_y (src/tracer/MOM_tracer_advect.F90:52-1087)
// as gfx950 stencil kernels.
//
// Design (DESIGN.md "advect_tracer"):
//  * Everything is updated IN PLACE, like the reference, so an extra iteration or an inactive
//    row/layer costs no copy traffic.  In-place is made race-free by the thread mapping:
//      - advect_x: one 64-lane wavefront owns one (j,k) row and sweeps it west->east in 64-cell
//        chunks.  A wave-private LDS tile keeps the pre-update values of the chunk plus a 3-cell halo
//        on both sides (the left halo is carried over from the previous chunk, the right halo is read
//        before anything to its east is written), so every stencil operand is the reference's
//        "T_tmp"/old hprev/old uhr.  West-face fluxes come from the neighbouring lane by a wavefront
//        shuffle.  No __syncthreads anywhere.
//      - advect_y: one lane owns one i column of a layer and marches south->north with the j stencil
//        in a register ring (rows J-1..J+3); a row is written only after every row the stencil needs
//        has been read by the same lane.  All loads/stores are i-contiguous (512 B per wave).
//  * Algorithmic traffic per pass = read+write of Tr(ntr), hprev and the remaining transport:
//    (ntr+2)*16 B per cell; unchanged values (land, inactive faces) are not written back.
//  * domore_u/domore_v/domore_k stay on the device; only domore_k (nk ints) is read back per
//    iteration for the exit test (the reference's sum_across_PEs at :305).
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"

namespace m6 { int halo_update_field(mom6hip_ctx_t *ctx, double *f, int pos, int nk); }      // (group_pass etc.: common.hpp)

namespace {

constexpr int MAXG = 4;      // tracers processed per kernel instance (register budget)
constexpr int XHALO = 3;     // i-halo of the advect_x tile (Colella-Woodward PPM needs 3)
constexpr int XTW = 64 + 2 * XHALO;

enum { PLM = MOM6HIP_ADV_PLM, H3 = MOM6HIP_ADV_PPM_H3, CW = MOM6HIP_ADV_PPM };

struct AdvArgs {
  m6::GridDev g;
  double *tr[MAXG];
  double cu[MAXG];           // conc_underflow
  double *hprev, *uhr, *vhr;
  int *domore_u;             // (jsd:jed, nk)
  int *domore_v_in;          // (jsd-1:jed, nk), state on entry
  int *domore_v_out;         // state on exit
  const int *domore_k;       // (nk)
  int is, ie, js, je;        // the reference's advect_x/advect_y range arguments
  int write_mass;            // 1: this tracer group also updates hprev / uhr|vhr / domore flags
  int any_cu;                // any conc_underflow > 0 in this group
};

// Orders this wave's LDS traffic: the hardware executes one wave's LDS instructions in order, so
// all that is needed is to stop the compiler from moving loads/stores across this point.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// "changed" in the bit-pattern sense (so that -0.0 -> +0.0 is written back like the reference does)
__device__ __forceinline__ bool changed(double a, double b) {
  return __double_as_longlong(a) != __double_as_longlong(b);
}

__device__ __forceinline__ double max3(double a, double b, double c) { return fmax(fmax(a, b), c); }
__device__ __forceinline__ double min3(double a, double b, double c) { return fmin(fmin(a, b), c); }

// x / d for a constant d (6 or 3), bit-identical to the IEEE division the reference performs but
// without the ~13-slot v_div/v_rcp sequence: q = x*RN(1/d) is within 1 ulp, the FMA residual
// r = x - d*q is exact, and RN(q + r*RN(1/d)) is then the correctly rounded quotient (Markstein's
// correction step; checked against x/d on 4e8 random operands in tests/).  Zero keeps its sign and
// operands outside [2^-900, 2^900] (where the residual could under/overflow) take the real division.
template <int D>
__device__ __forceinline__ double div_const(double x) {
  constexpr double d = (double)D, c = 1.0 / (double)D;
  const double q = x * c;
  const double r = __fma_rn(-d, q, x);
  const double qq = __fma_rn(r, c, q);
  const double ax = fabs(x);
  const bool fast = (ax > 0x1p-900) && (ax < 0x1p900);
  double res = fast ? qq : x;
  if (__builtin_expect(!fast && ax != 0.0, 0)) res = x / d;
  return res;
}

// limited slope, MOM_tracer_advect.F90:427-431 / :809-813
__device__ __forceinline__ double plm_slope(double Tp, double Tc, double Tm, double mask) {
  double dMx = max3(Tp, Tc, Tm) - Tc;
  double dMn = Tc - min3(Tp, Tc, Tm);
  return mask * copysign(min3(0.5 * fabs(Tp - Tm), 2.0 * dMx, 2.0 * dMn), Tp - Tm);
}

// Flux limiting of the remaining transport through one face, :486-513 / :872-899.
//   r_c      remaining transport through this face (positive towards +i / +j)
//   r_m, r_p remaining transport through the faces on the minus / plus side
//   h_m, h_p volumes of the cells on the minus / plus side;  a_m, a_p their areas
__device__ __forceinline__ void face_transport(double r_c, double r_m, double r_p, double h_m, double h_p,
                                               double a_m, double a_p, double min_h, double &hh,
                                               double &CFL, bool &limited) {
  const double tiny_h = DBL_MIN;
  limited = false;
  const bool none = (r_c == 0.0) || ((r_c < 0.0) && (h_p <= tiny_h)) || ((r_c > 0.0) && (h_m <= tiny_h));
  const bool neg = r_c < 0.0;
  hh = r_c;
  if (neg) {
    const double hup = h_p - a_p * min_h;
    const double hlos = fmax(0.0, r_p);
    if ((((hup - hlos) + r_c) < 0.0) && ((0.5 * hup + r_c) < 0.0)) {
      hh = min3(-0.5 * hup, -hup + hlos, 0.0);
      limited = true;
    }
  } else {
    const double hup = h_m - a_m * min_h;
    const double hlos = fmax(0.0, -r_m);
    if ((((hup - hlos) - r_c) < 0.0) && ((0.5 * hup - r_c) < 0.0)) {
      hh = max3(0.5 * hup, hup - hlos, 0.0);
      limited = true;
    }
  }
  // CFL = -uhh/hprev(i+1) or uhh/hprev(i): one division, the negation is exact
  const double q = (neg ? -hh : hh) / (neg ? h_p : h_m);
  if (none) { hh = 0.0; CFL = 0.0; limited = false; } else { CFL = q; }
}

// PPM edge values, CW84 monotonicity and the CFL-integrated flux, :526-556 / :911-941
template <int SCHEME>
__device__ __forceinline__ double ppm_flux(double Tp, double Tc, double Tm, double sm, double sc, double sp,
                                           double mask_prod, double hh, double CFL) {
  double aL, aR;
  if (SCHEME == H3) {
    aL = div_const<6>(5. * Tc + (2. * Tm - Tp));
    aL = fmax(fmin(Tc, Tm), aL); aL = fmin(fmax(Tc, Tm), aL);
    aR = div_const<6>(5. * Tc + (2. * Tp - Tm));
    aR = fmax(fmin(Tc, Tp), aR); aR = fmin(fmax(Tc, Tp), aR);
  } else {
    aL = 0.5 * ((Tm + Tc) + div_const<3>(sm - sc));
    aR = 0.5 * ((Tc + Tp) + div_const<3>(sc - sp));
  }
  double dA = aR - aL, mA = 0.5 * (aR + aL);
  const double dA2_6 = div_const<6>(dA * dA);
  if (mask_prod * (Tp - Tc) * (Tc - Tm) <= 0.) {
    aL = Tc; aR = Tc;
  } else if (dA * (Tc - mA) > dA2_6) {
    aL = 3. * Tc - 2. * aR;
  } else if (dA * (Tc - mA) < -dA2_6) {
    aR = 3. * Tc - 2. * aL;
  }
  double a6 = 6. * Tc - 3. * (aR + aL);
  if (hh >= 0.0)
    return hh * (aR - 0.5 * CFL * ((aR - aL) - a6 * (1. - 2. / 3. * CFL)));
  else
    return hh * (aL + 0.5 * CFL * ((aR - aL) + a6 * (1. - 2. / 3. * CFL)));
}

// New cell volume and the factors of the tracer update, :637-648 (x) / :1030-1039 (y).
template <bool CLAMP0>
__device__ __forceinline__ bool cell_volume(double hh_p, double hh_m, double h_old, double areaT,
                                            double h_neglect, double &h_new, double &hlst, double &Ihnew) {
  // returns do_i
  if ((hh_p != 0.0) || (hh_m != 0.0)) {
    hlst = h_old;
    h_new = h_old - (hh_p - hh_m);
    if (CLAMP0) h_new = fmax(h_new, 0.0);
    const double hmin = h_neglect * areaT;
    const bool thin = h_new < hmin;
    const double inv = 1.0 / (thin ? hmin : h_new);
    if (h_new <= 0.0) { Ihnew = 0.0; return false; }
    if (thin) hlst = hlst + (hmin - h_new);
    Ihnew = inv;
    return true;
  }
  h_new = h_old; hlst = h_old; Ihnew = 0.0;
  return false;
}

// ---------------------------------------------------------------------------------------------
// setup: uhr, vhr, hprev   (:152-178)
__global__ __launch_bounds__(256) void adv_setup_kernel(m6::GridDev g, const double *__restrict__ h_end,
                                                        const double *__restrict__ uhtr,
                                                        const double *__restrict__ vhtr,
                                                        const double *__restrict__ vol_prev,
                                                        double *__restrict__ hprev, double *__restrict__ uhr,
                                                        double *__restrict__ vhr) {
  // thread <-> (I, J) over the q-point extent (isd-1:ied, jsd-1:jed), which covers every array
  const int I = g.isd - 1 + blockIdx.x * blockDim.x + threadIdx.x;
  const int J = g.jsd - 1 + blockIdx.y;
  const int k = blockIdx.z;
  if (I > g.ied) return;
  const bool in_i = (I >= g.isc && I <= g.iec), in_j = (J >= g.jsc && J <= g.jec);
  if (J >= g.jsd) {   // u-point (I, j=J)
    double v = 0.0;
    if (in_j && I >= g.isc - 1 && I <= g.iec) v = uhtr[g.u3(I, J, k)];
    uhr[g.u3(I, J, k)] = v;
  }
  if (I >= g.isd) {   // v-point (i=I, J)
    double v = 0.0;
    if (in_i && J >= g.jsc - 1 && J <= g.jec) v = vhtr[g.v3(I, J, k)];
    vhr[g.v3(I, J, k)] = v;
  }
  if (I >= g.isd && J >= g.jsd) {   // h-point
    double hp = 0.0;
    if (in_i && in_j) {
      if (vol_prev) {
        hp = vol_prev[g.h3(I, J, k)];
      } else {
        const double aT = g.areaT[g.h2(I, J)], he = h_end[g.h3(I, J, k)];
        hp = fmax(0.0, aT * he + ((uhtr[g.u3(I, J, k)] - uhtr[g.u3(I - 1, J, k)]) +
                                  (vhtr[g.v3(I, J, k)] - vhtr[g.v3(I, J - 1, k)])));
        hp = hp + fmax(0.0, 1.0e-13 * hp - aT * he);
      }
    }
    hprev[g.h3(I, J, k)] = hp;
  }
}

// domore_u / domore_v re-evaluation, :215-225.  One block per (row, k).
struct ScanArgs {
  m6::GridDev g;
  const double *uhr, *vhr;
  int *domore_u, *domore_v;
  const int *domore_k;
  int ju0, ju1, iu0, iu1;   // u rows j and the I range scanned
  int jv0, jv1, iv0, iv1;   // v rows J and the i range scanned
};

__global__ __launch_bounds__(256) void adv_scan_kernel(ScanArgs a) {
  const m6::GridDev &g = a.g;
  const int k = blockIdx.y;
  if (a.domore_k[k] <= 0) return;
  const int r = blockIdx.x;            // row index over (jsd-1 : jed)
  // u row j = jsd-1+r
  {
    const int j = g.jsd - 1 + r;
    if (j >= a.ju0 && j <= a.ju1 && j >= g.jsd) {
      int *flag = &a.domore_u[(j - g.jsd) + g.njh * k];
      if (!*flag) {
        for (int I0 = a.iu0; I0 <= a.iu1; I0 += blockDim.x) {
          const int I = I0 + threadIdx.x;
          int nz = (I <= a.iu1) && (a.uhr[g.u3(I, j, k)] != 0.0);
          if (__syncthreads_or(nz)) { if (threadIdx.x == 0) *flag = 1; break; }
        }
      }
    }
  }
  {
    const int J = g.jsd - 1 + r;
    if (J >= a.jv0 && J <= a.jv1) {
      int *flag = &a.domore_v[(J - g.jsd + 1) + (g.njh + 1) * k];
      if (!*flag) {
        for (int i0 = a.iv0; i0 <= a.iv1; i0 += blockDim.x) {
          const int i = i0 + threadIdx.x;
          int nz = (i <= a.iv1) && (a.vhr[g.v3(i, J, k)] != 0.0);
          if (__syncthreads_or(nz)) { if (threadIdx.x == 0) *flag = 1; break; }
        }
      }
    }
  }
}

// domore_k(k) = any(domore_u(ju0:ju1,k)) .or. any(domore_v(jv0:jv1,k)), only where domore_k(k) > 0
// (:229-231, :265-267, :287-289).  One wave per k.
__global__ void adv_domore_k_kernel(m6::GridDev g, const int *domore_u, const int *domore_v, int *domore_k,
                                    int ju0, int ju1, int jv0, int jv1) {
  const int k = blockIdx.x;
  if (domore_k[k] <= 0) return;
  int any = 0;
  for (int j = ju0 + threadIdx.x; j <= ju1; j += blockDim.x) any |= domore_u[(j - g.jsd) + g.njh * k];
  for (int J = jv0 + threadIdx.x; J <= jv1; J += blockDim.x) any |= domore_v[(J - g.jsd + 1) + (g.njh + 1) * k];
  any = __syncthreads_or(any);
  if (threadIdx.x == 0) domore_k[k] = any ? 1 : 0;
}

// domore_v_out = domore_v_in, except that rows advect_y is about to process are cleared (:869).
__global__ void adv_vflags_prep_kernel(m6::GridDev g, const int *in, int *out, const int *domore_k, int js, int je) {
  const int n = (g.njh + 1) * g.nk;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const int k = t / (g.njh + 1);
    const int J = g.jsd - 1 + (t - k * (g.njh + 1));
    int v = in[t];
    if (domore_k[k] > 0 && J >= js - 1 && J <= je) v = 0;
    out[t] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// advect_x: one wavefront per (j,k) row.
// FIRST tags the launches of iteration 1 (every row active); the code is identical, the separate
// symbol keeps profiler statistics of the dense pass apart from the sparse later iterations.
template <int NT, int SCHEME, bool FIRST>
__global__ __launch_bounds__(256) void adv_x_kernel(AdvArgs p) {
  const m6::GridDev &g = p.g;
  __shared__ double s_T[4][NT][XTW];
  __shared__ double s_h[4][XTW];
  __shared__ double s_u[4][XTW];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int k = blockIdx.y;
  const int j = p.js + blockIdx.x * 4 + wave;
  if (j > p.je) return;
  if (p.domore_k[k] <= 0) return;
  int *flag = &p.domore_u[(j - g.jsd) + g.njh * k];
  const bool active = (*flag != 0);
  if (!active && !p.any_cu) return;

  const double min_h = 0.1 * g.Angstrom_H;
  const double h_neglect = g.H_subroundoff;
  const long rowH = g.h3(g.isd, j, k);          // offset of (isd, j, k) in h-point arrays
  const long rowU = g.u3(g.isd, j, k);          // offset of face I = isd
  const long row2H = g.h2(g.isd, j), row2U = g.u2(g.isd, j);

  if (!active) {
    // only the user-controlled underflow applies to this row (:683-687)
    for (int i = p.is + lane; i <= p.ie; i += 64) {
#pragma unroll
      for (int m = 0; m < NT; m++) if (p.cu[m] > 0.0) {
        double t = p.tr[m][rowH + (i - g.isd)];
        if (fabs(t) < p.cu[m] && changed(t, 0.0)) p.tr[m][rowH + (i - g.isd)] = 0.0;
      }
    }
    return;
  }

  double(*sT)[XTW] = s_T[wave];
  double *sh = s_h[wave], *su = s_u[wave];

  // chunks are aligned to the start of the array row
  const int c0 = (p.is - 1 - g.isd) >> 6, c1 = (p.ie - g.isd) >> 6;
  auto ldH = [&](const double *a, int i) -> double { return (i >= g.isd && i <= g.ied) ? a[rowH + (i - g.isd)] : 0.0; };
  auto ldU = [&](const double *a, int I) -> double { return (I >= g.isd - 1 && I <= g.ied) ? a[rowU + (I - g.isd)] : 0.0; };

  // first tile: cells i0-XHALO .. i0+63+XHALO
  {
    const int i0 = g.isd + 64 * c0;
    for (int q = lane; q < XTW; q += 64) {
      const int i = i0 - XHALO + q;
#pragma unroll
      for (int m = 0; m < NT; m++) sT[m][q] = ldH(p.tr[m], i);
      sh[q] = ldH(p.hprev, i);
      su[q] = ldU(p.uhr, i);
    }
  }
  double hh_carry = 0.0, fl_carry[NT];
#pragma unroll
  for (int m = 0; m < NT; m++) fl_carry[m] = 0.0;
  bool any_limited = false;

  for (int c = c0; c <= c1; c++) {
    const int i0 = g.isd + 64 * c;
    const int i = i0 + lane;                 // this lane's cell, and its east face I = i
    const int P = XHALO + lane;
    // prefetch the part of the next tile that is not already in LDS: cells i0+64+XHALO+lane
    double nT[NT], nh = 0.0, nu = 0.0;
    const bool more = (c < c1);
    if (more) {
      const int in = i0 + 64 + XHALO + lane;
#pragma unroll
      for (int m = 0; m < NT; m++) nT[m] = ldH(p.tr[m], in);
      nh = ldH(p.hprev, in);
      nu = ldU(p.uhr, in);
    }
    wave_sync();

    const bool face_ok = (i >= p.is - 1 && i <= p.ie);
    const bool cell_ok = (i >= p.is && i <= p.ie);
    double hh = 0.0, CFL = 0.0, flux[NT];
#pragma unroll
    for (int m = 0; m < NT; m++) flux[m] = 0.0;
    const double u_c = su[P];
    const double h_c = sh[P];
    double aT_c = 0.0;
    if (face_ok) {
      aT_c = g.areaT[row2H + (i - g.isd)];
      const double aT_e = g.areaT[row2H + (i + 1 - g.isd)];
      bool lim;
      face_transport(u_c, su[P - 1], su[P + 1], h_c, sh[P + 1], aT_c, aT_e, min_h, hh, CFL, lim);
      any_limited |= lim;
      const int up = (hh >= 0.0) ? 0 : 1;        // i_up = i + up
      const int Pu = P + up;
      if (SCHEME == PLM) {
        const double mk = g.mask2dCu[row2U + (i + up - g.isd)] * g.mask2dCu[row2U + (i + up - 1 - g.isd)];
#pragma unroll
        for (int m = 0; m < NT; m++) {
          const double Tc = sT[m][Pu];
          const double sl = plm_slope(sT[m][Pu + 1], Tc, sT[m][Pu - 1], mk);
          if (hh >= 0.0) flux[m] = hh * (Tc + 0.5 * sl * (1. - CFL));
          else           flux[m] = hh * (Tc - 0.5 * sl * (1. - CFL));
        }
      } else {
        const double mk_c = g.mask2dCu[row2U + (i + up - g.isd)] * g.mask2dCu[row2U + (i + up - 1 - g.isd)];
        double mk_m = 0.0, mk_p = 0.0;
        if (SCHEME == CW) {
          mk_m = g.mask2dCu[row2U + (i + up - 1 - g.isd)] * g.mask2dCu[row2U + (i + up - 2 - g.isd)];
          mk_p = g.mask2dCu[row2U + (i + up + 1 - g.isd)] * g.mask2dCu[row2U + (i + up - g.isd)];
        }
#pragma unroll
        for (int m = 0; m < NT; m++) {
          const double Tp = sT[m][Pu + 1], Tc = sT[m][Pu], Tm = sT[m][Pu - 1];
          double sm = 0., sc = 0., sp = 0.;
          if (SCHEME == CW) {
            sm = plm_slope(Tc, Tm, sT[m][Pu - 2], mk_m);
            sc = plm_slope(Tp, Tc, Tm, mk_c);
            sp = plm_slope(sT[m][Pu + 2], Tp, Tc, mk_p);
          }
          flux[m] = ppm_flux<SCHEME>(Tp, Tc, Tm, sm, sc, sp, mk_c, hh, CFL);
        }
      }
    }
    // west-face values from the neighbouring lane (lane 0: carried from the previous chunk)
    double hh_w = __shfl_up(hh, 1);
    if (lane == 0) hh_w = hh_carry;
    double fl_w[NT];
#pragma unroll
    for (int m = 0; m < NT; m++) {
      fl_w[m] = __shfl_up(flux[m], 1);
      if (lane == 0) fl_w[m] = fl_carry[m];
    }
    hh_carry = __shfl(hh, 63);
#pragma unroll
    for (int m = 0; m < NT; m++) fl_carry[m] = __shfl(flux[m], 63);

    // remaining transport, :632-635
    if (face_ok && p.write_mass) {
      double u_new = u_c - hh;
      if (fabs(u_new) < g.uh_neglect[row2U + (i - g.isd)]) u_new = 0.0;
      if (changed(u_new, u_c)) p.uhr[rowU + (i - g.isd)] = u_new;
    }
    // cell volume and tracers, :636-662, and underflow :683-687
    if (cell_ok) {
      double h_new, hlst, Ihnew;
      const bool do_i = cell_volume<false>(hh, hh_w, h_c, aT_c, h_neglect, h_new, hlst, Ihnew);
      if (p.write_mass && changed(h_new, h_c)) p.hprev[rowH + (i - g.isd)] = h_new;
#pragma unroll
      for (int m = 0; m < NT; m++) {
        const double t_old = sT[m][P];
        double t_new = t_old;
        if (do_i && Ihnew > 0.0) t_new = (t_old * hlst - (flux[m] - fl_w[m])) * Ihnew;
        if (p.cu[m] > 0.0 && fabs(t_new) < p.cu[m]) t_new = 0.0;
        if (changed(t_new, t_old)) p.tr[m][rowH + (i - g.isd)] = t_new;
      }
    }
    wave_sync();
    // slide the tile east by 64 cells
    if (more) {
      double kT[NT], kh = 0.0, ku = 0.0;
      if (lane < 2 * XHALO) {
#pragma unroll
        for (int m = 0; m < NT; m++) kT[m] = sT[m][64 + lane];
        kh = sh[64 + lane]; ku = su[64 + lane];
      }
      wave_sync();
      if (lane < 2 * XHALO) {
#pragma unroll
        for (int m = 0; m < NT; m++) sT[m][lane] = kT[m];
        sh[lane] = kh; su[lane] = ku;
      }
#pragma unroll
      for (int m = 0; m < NT; m++) sT[m][2 * XHALO + lane] = nT[m];
      sh[2 * XHALO + lane] = nh; su[2 * XHALO + lane] = nu;
    }
  }
  if (p.write_mass) {
    const int lim = __any(any_limited ? 1 : 0);
    if (lane == 0) *flag = lim ? 1 : 0;
  }
}

// ---------------------------------------------------------------------------------------------
// advect_y: one lane per (i,k) column.  The J range [js-1, je] of a 64-column strip is cut into
// `nseg` segments, one per wavefront of the block (occupancy: a single marching wave per strip leaves
// most of the chip idle).  In-place safety across segments: before the block's only barrier every wave
// has (a) filled its register ring with the rows at and below its first face and (b) copied the rows
// just above its last face -- which the next wave will overwrite early in its own march -- into LDS;
// after the barrier a wave reads global memory only inside its own segment.  Each wave starts with one
// face-only step at J = Ja-1 (recomputing the flux its southern neighbour also computes) so that no
// flux has to cross a wave boundary.
constexpr int YSEG_MAX = 8;

template <int NT, int SCHEME, bool FIRST>
#ifndef ADVY_OCC
#define ADVY_OCC 2      // waves per SIMD the register allocation aims at (tools/build_variant.sh for experiments)
#endif
__global__ __launch_bounds__(64 * YSEG_MAX, ADVY_OCC) void adv_y_kernel(AdvArgs p, int seglen) {
  const m6::GridDev &g = p.g;
  constexpr int NHR = (SCHEME == CW) ? 3 : 2;        // foreign T rows needed above the segment's last face
  __shared__ double s_halo[YSEG_MAX][NHR * NT + 2][64];
  const int seg = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = p.is + blockIdx.x * 64 + lane;
  const int k = blockIdx.y;
  if (p.domore_k[k] <= 0) return;                    // block-uniform
  const bool lane_ok = (i <= p.ie);
  const int ii = lane_ok ? i : p.ie;                 // idle lanes shadow a valid column, never store
  const double min_h = 0.1 * g.Angstrom_H;
  const double h_neglect = g.H_subroundoff;
  const long colH = g.h3(ii, g.jsd, k), colV = g.v3(ii, g.jsd, k);
  const long col2H = g.h2(ii, g.jsd), col2V = g.v2(ii, g.jsd);
  const int sH = g.nih;                              // j stride of h- and v-point arrays
  auto gH = [&](const double *a, int j) -> double { return (j >= g.jsd && j <= g.jed) ? a[colH + (long)sH * (j - g.jsd)] : 0.0; };
  auto gV = [&](const double *a, int J) -> double { return (J >= g.jsd - 1 && J <= g.jed) ? a[colV + (long)sH * (J - g.jsd)] : 0.0; };
  auto ldA = [&](int j) -> double { return (j >= g.jsd && j <= g.jed) ? g.areaT[col2H + (long)sH * (j - g.jsd)] : 0.0; };
  auto ldM = [&](int J) -> double { return (J >= g.jsd - 1 && J <= g.jed) ? g.mask2dCv[col2V + (long)sH * (J - g.jsd)] : 0.0; };
  const int *dv_in = p.domore_v_in + (long)(g.njh + 1) * k - (g.jsd - 1);
  int *dv_out = p.domore_v_out + (long)(g.njh + 1) * k - (g.jsd - 1);
  // is face J one advect_y works on (:868)?  (uniform)
  const int J0 = p.js - 1, J1 = p.je;
  auto act_face = [&](int J) -> unsigned { return ((J >= J0 && J <= J1) && dv_in[J] != 0) ? 1u : 0u; };

  const int Ja = J0 + seg * seglen;                  // first face this wave owns
  const int Jb = (Ja + seglen - 1 < J1) ? Ja + seglen - 1 : J1;
  const bool have = (Ja <= J1);
  const int Js = (seg == 0) ? J0 : Ja - 1;           // first step (face-only when seg > 0)

  // Row-level sparsity for iterations >= 2: fb bit n = "face J-3+n is active" at step J (a wave-uniform
  // shift register, refilled one flag per step, two steps ahead of its first use).
  // T(r) feeds the fluxes of faces r-2..r+1 (CW: r-3..r+2); hprev(r) those of faces r-1, r; a cell is
  // updated only if one of its two faces is active, which the same masks cover.
  unsigned fb = 0;
  if (have) for (int n = 0; n < 16; n++) fb |= act_face(Js - 3 + n) << n;
  const unsigned all = p.any_cu ? 1u : 0u;           // the underflow test needs every cell
  auto need_T = [&](unsigned bits, int d) -> bool {  // row r = J + d
    const unsigned m = (SCHEME == CW) ? (0x3Fu << d) : (0xFu << (d + 1));
    return ((bits & m) | all) != 0;
  };
  auto need_h = [&](unsigned bits, int d) -> bool { return ((bits & (0x3u << (d + 4))) | all) != 0; };   // row r = J+2+d: faces r-1, r

  // ---- before the barrier: ring at step Js, and the rows above Jb into LDS ----
  double t0[NT], t1[NT], t2[NT], t3[NT], t4[NT], t5[NT];
  double h_c = 0., h_n = 0., v_s = 0., v_c = 0., v_n = 0., a_c = 0., a_n = 0.;
  double m_ss = 0., m_s = 0., m_c = 0., m_n = 0., m_nn = 0.;
#pragma unroll
  for (int m = 0; m < NT; m++) { t0[m] = t1[m] = t2[m] = t3[m] = t4[m] = t5[m] = 0.0; }
  if (have) {
#pragma unroll
    for (int m = 0; m < NT; m++) {
      if (SCHEME == CW) t0[m] = gH(p.tr[m], Js - 2);
      t1[m] = gH(p.tr[m], Js - 1);
      t2[m] = gH(p.tr[m], Js);
      t3[m] = gH(p.tr[m], Js + 1);
      t4[m] = gH(p.tr[m], Js + 2);
      if (SCHEME == CW) t5[m] = gH(p.tr[m], Js + 3);
    }
    h_c = gH(p.hprev, Js); h_n = gH(p.hprev, Js + 1);
    v_s = gV(p.vhr, Js - 1); v_c = gV(p.vhr, Js); v_n = gV(p.vhr, Js + 1);
    a_c = ldA(Js); a_n = ldA(Js + 1);
    if (SCHEME == CW) { m_ss = ldM(Js - 2); m_nn = ldM(Js + 2); }
    m_s = ldM(Js - 1); m_c = ldM(Js); m_n = ldM(Js + 1);
    double(*hl)[64] = s_halo[seg];
#pragma unroll
    for (int r = 0; r < NHR; r++) {
#pragma unroll
      for (int m = 0; m < NT; m++) hl[r * NT + m][lane] = gH(p.tr[m], Jb + 1 + r);
    }
    hl[NHR * NT][lane] = gH(p.hprev, Jb + 1);
    hl[NHR * NT + 1][lane] = gV(p.vhr, Jb + 1);
  }
  __syncthreads();
  if (!have) return;
  const double(*hl)[64] = s_halo[seg];

  double hh_prev = 0.0, fl_prev[NT];
#pragma unroll
  for (int m = 0; m < NT; m++) fl_prev[m] = 0.0;
  constexpr int DT_ = (SCHEME == CW) ? 4 : 3;        // the T row entering the ring is J + DT_
  unsigned pend = act_face(Js + 13);

  // One step of the march.  TAIL = the new rows may lie above Jb and then come from LDS.
  auto step = [&](const int J, auto tail_tag) __attribute__((always_inline)) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    const int j = J;
    const bool own = (J >= Ja);                  // false only on the face-only first step of seg > 0
    // issue the loads of the rows that enter the rings at the next step
    double nt[NT], nh, nv;
    const int rT = J + DT_;
    if (!TAIL) {
      const bool nT = need_T(fb, DT_);
#pragma unroll
      for (int m = 0; m < NT; m++) nt[m] = nT ? p.tr[m][colH + (long)sH * (rT - g.jsd)] : 0.0;
      nh = need_h(fb, 0) ? p.hprev[colH + (long)sH * (J + 2 - g.jsd)] : 0.0;
      nv = p.vhr[colV + (long)sH * (J + 2 - g.jsd)];
    } else {
#pragma unroll
      for (int m = 0; m < NT; m++) {
        if (rT <= Jb) nt[m] = p.tr[m][colH + (long)sH * (rT - g.jsd)];
        else nt[m] = (rT <= Jb + NHR) ? hl[(rT - Jb - 1) * NT + m][lane] : 0.0;
      }
      if (J + 2 <= Jb) { nh = p.hprev[colH + (long)sH * (J + 2 - g.jsd)]; nv = p.vhr[colV + (long)sH * (J + 2 - g.jsd)]; }
      else { nh = (J + 2 == Jb + 1) ? hl[NHR * NT][lane] : 0.0; nv = (J + 2 == Jb + 1) ? hl[NHR * NT + 1][lane] : 0.0; }
    }
    const double na = ldA(J + 2);
    const double nm = ldM((SCHEME == CW) ? J + 3 : J + 2);
    const unsigned pend_next = act_face(J + 14);

    const bool act = (fb >> 3) & 1u;
    double hh = 0.0, CFL = 0.0, flux[NT];
#pragma unroll
    for (int m = 0; m < NT; m++) flux[m] = 0.0;
    if (act) {
      bool lim;
      face_transport(v_c, v_s, v_n, h_c, h_n, a_c, a_n, min_h, hh, CFL, lim);
      if (own && p.write_mass && __any((lim && lane_ok) ? 1 : 0) && lane == 0) dv_out[J] = 1;
      const bool up = !(hh >= 0.0);              // j_up = j + up
      // mask2dCv(i,J_up)*mask2dCv(i,J_up-1)
      const double mk_c = up ? (m_n * m_c) : (m_c * m_s);
#pragma unroll
      for (int m = 0; m < NT; m++) {
        const double Tm = up ? t2[m] : t1[m], Tc = up ? t3[m] : t2[m], Tp = up ? t4[m] : t3[m];
        if (SCHEME == PLM) {
          const double sl = plm_slope(Tp, Tc, Tm, mk_c);
          if (hh >= 0.0) flux[m] = hh * (Tc + 0.5 * sl * (1. - CFL));
          else           flux[m] = hh * (Tc - 0.5 * sl * (1. - CFL));
        } else {
          double sm = 0., sc = 0., sp = 0.;
          if (SCHEME == CW) {
            const double Tmm = up ? t1[m] : t0[m], Tpp = up ? t5[m] : t4[m];
            const double mk_m = up ? (m_c * m_s) : (m_s * m_ss);
            const double mk_p = up ? (m_nn * m_n) : (m_n * m_c);
            sm = plm_slope(Tc, Tm, Tmm, mk_m);
            sc = plm_slope(Tp, Tc, Tm, mk_c);
            sp = plm_slope(Tpp, Tp, Tc, mk_p);
          }
          flux[m] = ppm_flux<SCHEME>(Tp, Tc, Tm, sm, sc, sp, mk_c, hh, CFL);
        }
      }
    }
    // remaining transport, :1021-1024 (every row J, active or not)
    if (own && p.write_mass && lane_ok) {
      double v_new = v_c - hh;
      if (fabs(v_new) < g.vh_neglect[col2V + (long)sH * (J - g.jsd)]) v_new = 0.0;
      if (changed(v_new, v_c)) p.vhr[colV + (long)sH * (J - g.jsd)] = v_new;
    }
    // cell j = J, :1028-1059, and underflow :1062-1066
    if (own && J >= p.js && lane_ok) {
      double h_new, hlst, Ihnew;
      const bool do_i = cell_volume<true>(hh, hh_prev, h_c, a_c, h_neglect, h_new, hlst, Ihnew);
      if (p.write_mass && changed(h_new, h_c)) p.hprev[colH + (long)sH * (j - g.jsd)] = h_new;
#pragma unroll
      for (int m = 0; m < NT; m++) {
        const double t_old = t2[m];
        double t_new = t_old;
        if (do_i) t_new = (t_old * hlst - (flux[m] - fl_prev[m])) * Ihnew;
        if (p.cu[m] > 0.0 && fabs(t_new) < p.cu[m]) t_new = 0.0;
        if (changed(t_new, t_old)) p.tr[m][colH + (long)sH * (j - g.jsd)] = t_new;
      }
    }
    // advance the rings
    hh_prev = hh;
#pragma unroll
    for (int m = 0; m < NT; m++) {
      fl_prev[m] = flux[m];
      t0[m] = t1[m]; t1[m] = t2[m]; t2[m] = t3[m]; t3[m] = t4[m];
      if (SCHEME == CW) { t4[m] = t5[m]; t5[m] = nt[m]; } else { t4[m] = nt[m]; }
    }
    h_c = h_n; h_n = nh;
    v_s = v_c; v_c = v_n; v_n = nv;
    a_c = a_n; a_n = na;
    m_ss = m_s; m_s = m_c; m_c = m_n;
    if (SCHEME == CW) { m_n = m_nn; m_nn = nm; } else { m_n = nm; }
    fb = (fb >> 1) | (pend << 15);
    pend = pend_next;
  };

  // main part: every row that enters the rings lies inside the segment (and inside the array)
  int J = Js;
  const int Jmain = ((Jb - DT_ < g.jed - DT_) ? Jb - DT_ : g.jed - DT_);
  for (; J <= Jmain; J++) step(J, std::false_type{});
  for (; J <= Jb; J++) step(J, std::true_type{});
}

template <int NT, bool FIRST>
void launch_x2(int scheme, dim3 grid, hipStream_t s, const AdvArgs &a) {
  switch (scheme) {
    case PLM: hipLaunchKernelGGL((adv_x_kernel<NT, PLM, FIRST>), grid, dim3(256), 0, s, a); break;
    case H3:  hipLaunchKernelGGL((adv_x_kernel<NT, H3, FIRST>), grid, dim3(256), 0, s, a); break;
    default:  hipLaunchKernelGGL((adv_x_kernel<NT, CW, FIRST>), grid, dim3(256), 0, s, a); break;
  }
}
template <int NT>
void launch_x(int scheme, bool first, dim3 grid, hipStream_t s, const AdvArgs &a) {
  if (first) launch_x2<NT, true>(scheme, grid, s, a); else launch_x2<NT, false>(scheme, grid, s, a);
}
template <int NT, bool FIRST>
void launch_y2(int scheme, dim3 grid, int nseg, int seglen, hipStream_t s, const AdvArgs &a) {
  const dim3 block(64 * nseg);
  switch (scheme) {
    case PLM: hipLaunchKernelGGL((adv_y_kernel<NT, PLM, FIRST>), grid, block, 0, s, a, seglen); break;
    case H3:  hipLaunchKernelGGL((adv_y_kernel<NT, H3, FIRST>), grid, block, 0, s, a, seglen); break;
    default:  hipLaunchKernelGGL((adv_y_kernel<NT, CW, FIRST>), grid, block, 0, s, a, seglen); break;
  }
}
template <int NT>
void launch_y(int scheme, bool first, dim3 grid, int nseg, int seglen, hipStream_t s, const AdvArgs &a) {
  if (first) launch_y2<NT, true>(scheme, grid, nseg, seglen, s, a);
  else launch_y2<NT, false>(scheme, grid, nseg, seglen, s, a);
}

// ---------------------------------------------------------------------------------------------
// The general form of advect_x / advect_y: two plain kernels a pass, a thread a face (the transport taken through the face and the fluxes
// of every tracer, :485-627 / :868-1014, into scratch) and a thread a cell (:632-687 / :1021-1066).  It is the path of an OBC whose
// segments carry tracer registries (segment%tr_Reg: a registered tracer takes its reservoir value or inflow concentration in the cell outside
// the segment :441-462 / :823-846, the slopes of the three cells about the segment's face are formed again, each with the masks of its own two
// faces (the loop index of :466 / :850 is the I / J of the mask expression: Fortran does not tell the cases apart) :463-473 / :847-856, an inflow carries the reservoir value with the whole remaining transport :580-627 / :965-1014); regional
// configurations are small, and the marching kernels above stay as they are.  MOM6HIP_ADV_GENERIC=1 takes this path without OBC (tests).
constexpr int GEN_MAXTR = 64;
struct GenTr { double *t[GEN_MAXTR]; double cu[GEN_MAXTR]; int ntr; };
struct GenSeg {      // a segment of the direction of the pass that has a tracer registry
  int plus;          // OBC_DIRECTION_E | N: the cell outside is the second cell of its faces
  int specified, A, c0, c1;      // the face index (IsdB | JsdB) and the range across it (jsd:jed | isd:ied)
  int a0, na, b0, nb;            // its own arrays: element (a, b, k) at (a - a0) + na * ((b - b0) + nb * k)
  int r0, ntseg;                 // its registry: regs[r0 .. r0 + ntseg)
};
struct GenReg { int m; const double *tres; double conc; };
struct GenArgs {
  m6::GridDev g;
  GenTr T;
  double *hprev, *xr;            // xr: uhr | vhr
  double *hh, *flux;             // scratch: the transport taken through every face in this pass; the fluxes, + fstride a tracer
  long fstride;
  int *dom_in, *dom_new;         // the rows' flags on entry; where a limited face flags its row for another pass
  const int *domore_k;
  const double *neglect;         // uh_neglect | vh_neglect
  int is, ie, js, je;
  int nseg; const GenSeg *segs; const GenReg *regs;
  int obc_any, obc_open;         // OBC%specified_?_BCs_exist_globally .or. OBC%open_?_BCs_exist_globally ; OBC%open_?_BCs_exist_globally
};

template <int DIR> __device__ __forceinline__ long gen_h2(const m6::GridDev &g, int n, int c) { return DIR ? g.h2(c, n) : g.h2(n, c); }
template <int DIR> __device__ __forceinline__ long gen_f2(const m6::GridDev &g, int n, int c) { return DIR ? g.v2(c, n) : g.u2(n, c); }

__device__ __forceinline__ double gen_seg_value(const GenSeg &s, const GenReg &r, int a, int b, int k) {
  if (!r.tres) return r.conc;
  return r.tres[(a - s.a0) + (long)s.na * ((b - s.b0) + (long)s.nb * k)];
}

template <int DIR, int SCHEME>
__global__ __launch_bounds__(256) void gen_flux_kernel(GenArgs p) {
  const m6::GridDev &g = p.g;
  const int tx = blockIdx.x * 256 + threadIdx.x, k = blockIdx.z;
  const int n = (DIR ? p.js - 1 + (int)blockIdx.y : p.is - 1 + tx);      // the face index along the direction
  const int c = (DIR ? p.is + tx : p.js + (int)blockIdx.y);             // the index across it
  if (DIR ? (c > p.ie) : (n > p.ie)) return;
  if (p.domore_k[k] <= 0) return;
  const long hpl = (long)g.nih * g.njh, fpl = DIR ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh;
  const long f2 = gen_f2<DIR>(g, n, c), f3 = f2 + fpl * k;
  const long fstep = DIR ? g.nih : 1;      // to the next face along the direction
  const int row = DIR ? (n - g.jsd + 1) + (g.njh + 1) * k : (c - g.jsd) + g.njh * k;
  const bool active = p.dom_in[row] != 0;
  if (!active) {
    if (DIR) { p.hh[f3] = 0.0; for (int m = 0; m < p.T.ntr; m++) p.flux[p.fstride * m + f3] = 0.0; }
    return;
  }
  const double min_h = 0.1 * g.Angstrom_H;
  const int nlo = DIR ? g.jsd : g.isd, nhi = DIR ? g.jed : g.ied;      // cells of the data domain along the direction
  const double *mask = DIR ? g.mask2dCv : g.mask2dCu;
  auto fmask = [&](int nn) { return mask[gen_f2<DIR>(g, nn, c)]; };
  // T_tmp of tracer m at the cell nn of this line
  auto Tt = [&](int m, int nn) -> double {
    double v = (nn >= nlo && nn <= nhi) ? p.T.t[m][gen_h2<DIR>(g, nn, c) + hpl * k] : 0.0;
    for (int q = 0; q < p.nseg; q++) {
      const GenSeg &s = p.segs[q];
      if (c < s.c0 || c > s.c1 || nn != (s.plus ? s.A + 1 : s.A)) continue;
      for (int r = s.r0; r < s.r0 + s.ntseg; r++)
        if (p.regs[r].m == m) v = DIR ? gen_seg_value(s, p.regs[r], c, s.A, k) : gen_seg_value(s, p.regs[r], s.A, c, k);
    }
    return v;
  };
  auto Traw = [&](int m, int nn) -> double { return (nn >= nlo && nn <= nhi) ? p.T.t[m][gen_h2<DIR>(g, nn, c) + hpl * k] : 0.0; };
  // the slopes are those of the tracer itself (:425-436), but for the three cells about a segment's face, formed again from T_tmp (:466-474)
  auto slope = [&](int m, int nn) -> double {
    for (int q = 0; q < p.nseg; q++) {
      const GenSeg &s = p.segs[q];
      if (c >= s.c0 && c <= s.c1 && nn >= s.A - 1 && nn <= s.A + 1)
        return plm_slope(Tt(m, nn + 1), Tt(m, nn), Tt(m, nn - 1), fmask(nn) * fmask(nn - 1));      // (:472, :854: Fortran's I is i -- the loop cell's masks)
    }
    return plm_slope(Traw(m, nn + 1), Traw(m, nn), Traw(m, nn - 1), fmask(nn) * fmask(nn - 1));
  };
  const long hm = gen_h2<DIR>(g, n, c), hp = gen_h2<DIR>(g, n + 1, c);
  double hh, CFL;
  bool lim;
  face_transport(p.xr[f3], p.xr[f3 - fstep], p.xr[f3 + fstep], p.hprev[hm + hpl * k], p.hprev[hp + hpl * k], g.areaT[hm], g.areaT[hp], min_h,
                 hh, CFL, lim);
  if (lim) atomicOr(&p.dom_new[row], 1);
  const int up = (hh >= 0.0) ? 0 : 1, cu = n + up;
  for (int m = 0; m < p.T.ntr; m++) {
    double fl;
    if (SCHEME == PLM) {
      const double Tc = Tt(m, cu), sl = slope(m, cu);
      if (hh >= 0.0) fl = hh * (Tc + 0.5 * sl * (1. - CFL));
      else           fl = hh * (Tc - 0.5 * sl * (1. - CFL));
    } else {
      const double Tp = Tt(m, cu + 1), Tc = Tt(m, cu), Tm = Tt(m, cu - 1);
      double sm = 0., sc = 0., sp = 0.;
      if (SCHEME == CW) { sm = slope(m, cu - 1); sc = slope(m, cu); sp = slope(m, cu + 1); }
      fl = ppm_flux<SCHEME>(Tp, Tc, Tm, sm, sc, sp, fmask(cu) * fmask(cu - 1), hh, CFL);
    }
    p.flux[p.fstride * m + f3] = fl;
  }
  // the inflows through the faces of the segments (two loops over the segments, as the reference)
  if (p.nseg > 0 && p.obc_any) {
    const double r = p.xr[f3];
    for (int pass = 0; pass < 2; pass++) {
      if (pass == 1 && !p.obc_open) break;
      for (int q = 0; q < p.nseg; q++) {
        const GenSeg &s = p.segs[q];
        if (n != s.A || c < s.c0 || c > s.c1) continue;
        bool inflow;
        if (pass == 0) {
          if (DIR && !s.specified) continue;      // (advect_y :969; advect_x takes every segment :583)
          inflow = ((r > 0.0) && !s.plus) || ((r < 0.0) && s.plus);
        } else {
          if (s.specified) continue;
          inflow = ((r > 0.0) && (g.mask2dT[hm] < 0.5)) || ((r < 0.0) && (g.mask2dT[hp] < 0.5));
        }
        if (!inflow) continue;
        hh = r;
        for (int t = s.r0; t < s.r0 + s.ntseg; t++)
          p.flux[p.fstride * p.regs[t].m + f3] = hh * (DIR ? gen_seg_value(s, p.regs[t], c, s.A, k) : gen_seg_value(s, p.regs[t], s.A, c, k));
      }
    }
  }
  p.hh[f3] = hh;
}

template <int DIR>
__global__ __launch_bounds__(256) void gen_update_kernel(GenArgs p) {
  const m6::GridDev &g = p.g;
  const int tx = blockIdx.x * 256 + threadIdx.x, k = blockIdx.z;
  const int n = (DIR ? p.js - 1 + (int)blockIdx.y : p.is - 1 + tx);
  const int c = (DIR ? p.is + tx : p.js + (int)blockIdx.y);
  if (DIR ? (c > p.ie) : (n > p.ie)) return;
  if (p.domore_k[k] <= 0) return;
  const long hpl = (long)g.nih * g.njh, fpl = DIR ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh;
  const long f2 = gen_f2<DIR>(g, n, c), f3 = f2 + fpl * k;
  const long fstep = DIR ? g.nih : 1;
  const bool active = DIR ? true : (p.dom_in[(c - g.jsd) + g.njh * k] != 0);
  const bool cell = n >= (DIR ? p.js : p.is);
  const long h2 = gen_h2<DIR>(g, n, c), h3 = h2 + hpl * k;
  if (active) {
    const double hh = p.hh[f3];
    double r = p.xr[f3] - hh;      // :632-635 / :1021-1024
    if (fabs(r) < p.neglect[f2]) r = 0.0;
    p.xr[f3] = r;
    if (cell) {
      const double hh_m = p.hh[f3 - fstep], h_old = p.hprev[h3];
      double h_new, hlst, Ihnew;
      const bool do_i = DIR ? cell_volume<true>(hh, hh_m, h_old, g.areaT[h2], g.H_subroundoff, h_new, hlst, Ihnew)
                            : cell_volume<false>(hh, hh_m, h_old, g.areaT[h2], g.H_subroundoff, h_new, hlst, Ihnew);
      if (changed(h_new, h_old)) p.hprev[h3] = h_new;
      if (do_i && Ihnew > 0.0)
        for (int m = 0; m < p.T.ntr; m++) {
          const double t_old = p.T.t[m][h3];
          p.T.t[m][h3] = (t_old * hlst - (p.flux[p.fstride * m + f3] - p.flux[p.fstride * m + f3 - fstep])) * Ihnew;
        }
    }
  }
  if (cell)      // :683-687 / :1062-1066
    for (int m = 0; m < p.T.ntr; m++) if (p.T.cu[m] > 0.0) {
      const double t = p.T.t[m][h3];
      if (fabs(t) < p.T.cu[m] && changed(t, 0.0)) p.T.t[m][h3] = 0.0;
    }
}

// advect_x: the flags of the rows that were processed take what the pass found (:413, :497, :507)
__global__ void gen_xflags_kernel(m6::GridDev g, int *dom, const int *dom_new, const int *domore_k, int js, int je) {
  const int n = g.njh * g.nk;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const int k = t / g.njh, j = g.jsd + (t - k * g.njh);
    if (domore_k[k] > 0 && j >= js && j <= je && dom[t]) dom[t] = dom_new[t];
  }
}

struct Timer {
  hipEvent_t a = nullptr, b = nullptr;
  bool on = false;
  hipStream_t s;
  Timer(bool enable, hipStream_t st) : on(enable), s(st) {
    if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); }
  }
  ~Timer() { if (on) { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } }
  void start() { if (on) (void)hipEventRecord(a, s); }
  // returns ms; synchronises (timing mode only)
  double stop() {
    if (!on) return 0.0;
    (void)hipEventRecord(b, s); (void)hipEventSynchronize(b);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, a, b);
    return ms;
  }
};

}  // namespace

extern "C" int mom6hip_advect_get_timing(mom6hip_ctx_t *ctx, mom6hip_advect_timing_t *t) {
  M6_REQUIRE(ctx && t, "mom6hip_advect_get_timing: null argument");
  *t = ctx->adv_timing;
  return 0;
}

extern "C" int mom6hip_advect_tracer(mom6hip_ctx_t *ctx, const double *h_end, const double *uhtr,
                                     const double *vhtr, double dt, const mom6hip_tracer_advect_cs_t *cs,
                                     double *const *tr, const double *conc_underflow, int32_t ntr,
                                     int32_t x_first_in, double *vol_prev, int32_t max_iter_in,
                                     int32_t update_vol_prev, double *uhr_out, double *vhr_out,
                                     int32_t memspace, mom6hip_advect_stats_t *stats) {
  return mom6hip_advect_tracer_obc(ctx, h_end, uhtr, vhtr, dt, cs, tr, conc_underflow, ntr, x_first_in, vol_prev, max_iter_in, update_vol_prev,
                                   uhr_out, vhr_out, nullptr, memspace, stats);
}

// advect_tracer with OBC associated: of the OBC, advect_x / advect_y read the tracer registries of the segments (segment%tr_Reg); without
// any the marching kernels run as for a closed domain, with one the general kernels (gen_flux_kernel, gen_update_kernel)
extern "C" int mom6hip_advect_tracer_obc(mom6hip_ctx_t *ctx, const double *h_end, const double *uhtr,
                                         const double *vhtr, double dt, const mom6hip_tracer_advect_cs_t *cs,
                                         double *const *tr, const double *conc_underflow, int32_t ntr,
                                         int32_t x_first_in, double *vol_prev, int32_t max_iter_in,
                                         int32_t update_vol_prev, double *uhr_out, double *vhr_out, const mom6hip_obc_t *obc,
                                         int32_t memspace, mom6hip_advect_stats_t *stats) {
  M6_REQUIRE(ctx != nullptr, "advect_tracer: null context (tracer_advect_init must be called before advect_tracer)");
  if (stats) memset(stats, 0, sizeof(*stats));
  M6_REQUIRE(ntr >= 0, "advect_tracer: ntr < 0");
  if (ntr == 0) return 0;   // :124
  M6_REQUIRE(cs != nullptr, "advect_tracer: tracer_advect_init must be called before advect_tracer");
  M6_REQUIRE(h_end && uhtr && vhtr && tr, "advect_tracer: null field pointer");
  M6_REQUIRE(cs->scheme == PLM || cs->scheme == H3 || cs->scheme == CW,
             "MOM_tracer_advect: Unknown TRACER_ADVECTION_SCHEME = %d", cs->scheme);
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "advect_tracer: bad memspace");
  M6_REQUIRE(ntr <= 64, "advect_tracer: at most 64 tracers are supported");
  M6_REQUIRE(dt > 0.0 && cs->dt > 0.0, "advect_tracer: dt must be positive");
  for (int m = 0; m < ntr; m++) M6_REQUIRE(tr[m] != nullptr, "advect_tracer: tracer %d is null", m);

  m6::GridDev &g = ctx->g;
  hipStream_t s = ctx->stream;
  const int is = g.isc, ie = g.iec, js = g.jsc, je = g.jec, nz = g.nk;
  const size_t bH = (size_t)g.nh3() * 8, bU = (size_t)g.nu3() * 8, bV = (size_t)g.nv3() * 8;

  int stencil = 2;
  const bool usePPM = cs->scheme != PLM;
  if (usePPM && !cs->use_huynh_stencil_bug) stencil = 3;
  M6_REQUIRE(is - g.isd >= stencil && g.ied - ie >= stencil && js - g.jsd >= stencil && g.jed - je >= stencil,
             "advect_tracer: halo (%d) narrower than the advection stencil (%d)", is - g.isd, stencil);
  bool x_first = (ctx->host.first_direction % 2) == 0;
  int max_iter = 2 * (int)ceil(dt / cs->dt) + 1;
  if (max_iter_in > 0) max_iter = max_iter_in;
  if (x_first_in >= 0) x_first = x_first_in != 0;

  Timer t_all(ctx->timing, s), t_k(ctx->timing, s);
  mom6hip_advect_timing_t tm = {};
  t_all.start();

  // ---- device views of the caller's arrays ----
  const double *d_hend = h_end, *d_uhtr = uhtr, *d_vhtr = vhtr;
  double *d_vol = vol_prev;
  std::vector<double *> d_tr(ntr);
  if (memspace == MOM6HIP_MEM_HOST) {
    if (ctx->stage[0].reserve(bH) || ctx->stage[1].reserve(bU) || ctx->stage[2].reserve(bV)) return 1;
    M6_HIP(hipMemcpyAsync(ctx->stage[0].p, h_end, bH, hipMemcpyHostToDevice, s));
    M6_HIP(hipMemcpyAsync(ctx->stage[1].p, uhtr, bU, hipMemcpyHostToDevice, s));
    M6_HIP(hipMemcpyAsync(ctx->stage[2].p, vhtr, bV, hipMemcpyHostToDevice, s));
    d_hend = (double *)ctx->stage[0].p; d_uhtr = (double *)ctx->stage[1].p; d_vhtr = (double *)ctx->stage[2].p;
    if (vol_prev) {
      if (ctx->stage[3].reserve(bH)) return 1;
      M6_HIP(hipMemcpyAsync(ctx->stage[3].p, vol_prev, bH, hipMemcpyHostToDevice, s));
      d_vol = (double *)ctx->stage[3].p;
    }
    for (int m = 0; m < ntr; m++) {
      if (ctx->tr_stage[m].reserve(bH)) return 1;
      M6_HIP(hipMemcpyAsync(ctx->tr_stage[m].p, tr[m], bH, hipMemcpyHostToDevice, s));
      d_tr[m] = (double *)ctx->tr_stage[m].p;
    }
  } else {
    for (int m = 0; m < ntr; m++) d_tr[m] = tr[m];
  }

  // ---- work space ----
  const size_t nfu = (size_t)g.njh * nz, nfv = (size_t)(g.njh + 1) * nz;
  if (ctx->hprev.reserve(bH) || ctx->uhr.reserve(bU) || ctx->vhr.reserve(bV) ||
      ctx->flags.reserve((nfu + 2 * nfv + nz) * sizeof(int)))
    return 1;
  double *hprev = (double *)ctx->hprev.p, *uhr = (double *)ctx->uhr.p, *vhr = (double *)ctx->vhr.p;
  int *domore_u = (int *)ctx->flags.p, *domore_v = domore_u + nfu, *domore_v2 = domore_v + nfv;
  int *domore_k = domore_v2 + nfv;
  M6_HIP(hipMemsetAsync(domore_u, 0, (nfu + 2 * nfv) * sizeof(int), s));
  for (int k = 0; k < nz; k++) ctx->h_domore_k[k] = 1;
  M6_HIP(hipMemcpyAsync(domore_k, ctx->h_domore_k, nz * sizeof(int), hipMemcpyHostToDevice, s));

  // ---- the general path: the tracer registries of the segments (OBC%OBC_pe), or on request ----
  bool generic = getenv("MOM6HIP_ADV_GENERIC") && atoi(getenv("MOM6HIP_ADV_GENERIC")) != 0;
  std::vector<GenSeg> gsegs[2];
  std::vector<GenReg> gregs;
  ctx->adv_obc_n = 0;      // (the staged reservoirs of this call: counted from zero whatever a refused call left behind)
  if (obc && obc->OBC_pe) {
    M6_REQUIRE(obc->number_of_segments == 0 || obc->segment, "advect_tracer: OBC%%segment is required");
    for (int n = 0; n < obc->number_of_segments; n++) {
      const mom6hip_obc_segment_t &S = obc->segment[n];
      if (!S.tr_Reg) continue;
      if (!(S.is_E_or_W || S.is_N_or_S)) continue;      // (not on this PE: read by neither advect_x nor advect_y)
      generic = true;
      const int d = S.is_N_or_S ? 1 : 0;
      GenSeg q;
      q.plus = (S.direction == MOM6HIP_OBC_DIRECTION_E || S.direction == MOM6HIP_OBC_DIRECTION_N) ? 1 : 0;
      q.specified = S.specified; q.A = d ? S.JsdB : S.IsdB; q.c0 = d ? S.isd : S.jsd; q.c1 = d ? S.ied : S.jed;
      q.a0 = d ? S.isd : S.IsdB; q.na = d ? (S.ied - S.isd + 1) : (S.IedB - S.IsdB + 1);
      q.b0 = d ? S.JsdB : S.jsd; q.nb = d ? (S.JedB - S.JsdB + 1) : (S.jed - S.jsd + 1);
      M6_REQUIRE(q.A >= (d ? g.jsd : g.isd) + 2 && q.A <= (d ? g.jed : g.ied) - 3 && q.c0 >= (d ? g.isd : g.jsd) && q.c1 <= (d ? g.ied : g.jed),
                 "advect_tracer: OBC segment %d lies outside the data domain", n + 1);
      q.r0 = (int)gregs.size(); q.ntseg = S.ntseg;
      for (int t = 0; t < S.ntseg; t++) {
        const mom6hip_obc_segment_tracer_t &R = S.tr_Reg[t];
        M6_REQUIRE(R.ntr_index >= 1 && R.ntr_index <= ntr, "advect_tracer: the registry of OBC segment %d names tracer %d of %d", n + 1, R.ntr_index, ntr);
        GenReg r; r.m = R.ntr_index - 1; r.conc = R.OBC_inflow_conc; r.tres = nullptr;
        if (R.tres) {
          const size_t bytes = (size_t)q.na * q.nb * nz * 8;
          if (memspace == MOM6HIP_MEM_HOST) {
            M6_REQUIRE(ctx->adv_obc_n < 64 && ctx->adv_obc[ctx->adv_obc_n].reserve(bytes) == 0, "advect_tracer: out of device memory for the tracer reservoirs");
            M6_HIP(hipMemcpyAsync(ctx->adv_obc[ctx->adv_obc_n].p, R.tres, bytes, hipMemcpyHostToDevice, s));
            r.tres = (const double *)ctx->adv_obc[ctx->adv_obc_n++].p;
          } else r.tres = R.tres;
        }
        gregs.push_back(r);
      }
      // the slopes about a segment are formed again with the tracer values of that moment: two segments with registries within three
      // cells of each other on a line would make the order of the reference's loop over the segments matter
      for (const GenSeg &o : gsegs[d])
        M6_REQUIRE(o.c1 < q.c0 || o.c0 > q.c1 || abs(o.A - q.A) > 3, "advect_tracer: OBC segments with tracer registries closer than four cells are not provided");
      gsegs[d].push_back(q);
    }
  }
  ctx->adv_obc_n = 0;
  M6_REQUIRE(!generic || ntr <= GEN_MAXTR, "advect_tracer: at most %d tracers on the general path", GEN_MAXTR);
  GenSeg *d_gsegs[2] = {nullptr, nullptr};
  GenReg *d_gregs = nullptr;
  double *gen_scratch = nullptr;
  int *gen_flags = nullptr;
  const size_t fmax_el = (size_t)(g.nu3() > g.nv3() ? g.nu3() : g.nv3());
  if (generic) {
    const size_t bs0 = sizeof(GenSeg) * (gsegs[0].size() + 1), bs1 = sizeof(GenSeg) * (gsegs[1].size() + 1), br = sizeof(GenReg) * (gregs.size() + 1);
    if (ctx->adv_gen.reserve(bs0 + bs1 + br + 64 + nfu * sizeof(int) + fmax_el * 8 * ((size_t)ntr + 1))) return 1;
    char *q = (char *)ctx->adv_gen.p;
    gen_scratch = (double *)q; q += fmax_el * 8 * ((size_t)ntr + 1);
    d_gsegs[0] = (GenSeg *)q; q += bs0; d_gsegs[1] = (GenSeg *)q; q += bs1; d_gregs = (GenReg *)q; q += br;
    q = (char *)(((uintptr_t)q + 15) & ~(uintptr_t)15);
    gen_flags = (int *)q;
    if (!gsegs[0].empty()) M6_HIP(hipMemcpyAsync(d_gsegs[0], gsegs[0].data(), sizeof(GenSeg) * gsegs[0].size(), hipMemcpyHostToDevice, s));
    if (!gsegs[1].empty()) M6_HIP(hipMemcpyAsync(d_gsegs[1], gsegs[1].data(), sizeof(GenSeg) * gsegs[1].size(), hipMemcpyHostToDevice, s));
    if (!gregs.empty()) M6_HIP(hipMemcpyAsync(d_gregs, gregs.data(), sizeof(GenReg) * gregs.size(), hipMemcpyHostToDevice, s));
    M6_HIP(hipStreamSynchronize(s));      // (the host vectors are read by the copies)
  }
  auto gen_args = [&](int d, GenArgs &a) {
    a.g = g; a.T.ntr = ntr;
    for (int m = 0; m < ntr; m++) { a.T.t[m] = d_tr[m]; a.T.cu[m] = conc_underflow ? conc_underflow[m] : 0.0; }
    a.hprev = hprev; a.xr = d ? vhr : uhr; a.hh = gen_scratch; a.flux = gen_scratch + fmax_el; a.fstride = (long)fmax_el;
    a.domore_k = domore_k; a.neglect = d ? g.vh_neglect : g.uh_neglect;
    a.nseg = (int)gsegs[d].size(); a.segs = d_gsegs[d]; a.regs = d_gregs;
    a.obc_any = obc ? (d ? (obc->specified_v_BCs_exist_globally || obc->open_v_BCs_exist_globally)
                         : (obc->specified_u_BCs_exist_globally || obc->open_u_BCs_exist_globally)) : 0;
    a.obc_open = obc ? (d ? obc->open_v_BCs_exist_globally : obc->open_u_BCs_exist_globally) : 0;
  };

  // ---- :152-178 ----
  t_k.start();
  {
    dim3 grid((g.nih + 1 + 255) / 256, g.njh + 1, nz);
    hipLaunchKernelGGL(adv_setup_kernel, grid, dim3(256), 0, s, g, d_hend, d_uhtr, d_vhtr, (const double *)d_vol,
                       hprev, uhr, vhr);
    M6_HIP(hipGetLastError());
  }
  tm.ms_setup += t_k.stop();

  int itt = 1;
  const int ngroups = (ntr + MAXG - 1) / MAXG;
  auto group_args = [&](int grp, AdvArgs &a) -> int {
    a.g = g; a.hprev = hprev; a.uhr = uhr; a.vhr = vhr;
    a.domore_u = domore_u; a.domore_k = domore_k;
    const int m0 = grp * MAXG;
    const int n = (ntr - m0 < MAXG) ? ntr - m0 : MAXG;
    a.any_cu = 0;
    for (int m = 0; m < MAXG; m++) {
      a.tr[m] = (m < n) ? d_tr[m0 + m] : nullptr;
      a.cu[m] = (m < n && conc_underflow) ? conc_underflow[m0 + m] : 0.0;
      if (a.cu[m] > 0.0) a.any_cu = 1;
    }
    a.write_mass = (grp == ngroups - 1);   // earlier groups must see the pre-pass hprev/uhr/vhr/flags
    return n;
  };
  auto run_x = [&](int xis, int xie, int xjs, int xje) -> int {
    t_k.start();
    if (generic) {
      GenArgs a; gen_args(0, a);
      a.is = xis; a.ie = xie; a.js = xjs; a.je = xje; a.dom_in = domore_u; a.dom_new = gen_flags;
      M6_HIP(hipMemsetAsync(gen_flags, 0, nfu * sizeof(int), s));
      const dim3 grid((xie - xis + 2 + 255) / 256, xje - xjs + 1, nz);
      if (cs->scheme == PLM) hipLaunchKernelGGL((gen_flux_kernel<0, PLM>), grid, dim3(256), 0, s, a);
      else if (cs->scheme == H3) hipLaunchKernelGGL((gen_flux_kernel<0, H3>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((gen_flux_kernel<0, CW>), grid, dim3(256), 0, s, a);
      hipLaunchKernelGGL(gen_update_kernel<0>, grid, dim3(256), 0, s, a);
      hipLaunchKernelGGL(gen_xflags_kernel, dim3(64), dim3(256), 0, s, g, domore_u, (const int *)gen_flags, (const int *)domore_k, xjs, xje);
      M6_HIP(hipGetLastError());
      tm.n_x++;
      { const double ms = t_k.stop(); tm.ms_x += ms; if (itt == 1) tm.ms_x1 += ms; }
      return 0;
    }
    for (int grp = 0; grp < ngroups; grp++) {
      AdvArgs a; const int n = group_args(grp, a);
      a.is = xis; a.ie = xie; a.js = xjs; a.je = xje;
      a.domore_v_in = domore_v; a.domore_v_out = domore_v;
      dim3 grid((xje - xjs + 1 + 3) / 4, nz);
      switch (n) {
        case 1: launch_x<1>(cs->scheme, itt == 1, grid, s, a); break;
        case 2: launch_x<2>(cs->scheme, itt == 1, grid, s, a); break;
        case 3: launch_x<3>(cs->scheme, itt == 1, grid, s, a); break;
        default: launch_x<4>(cs->scheme, itt == 1, grid, s, a); break;
      }
      M6_HIP(hipGetLastError());
      tm.n_x++;
    }
    { const double ms = t_k.stop(); tm.ms_x += ms; if (itt == 1) tm.ms_x1 += ms; }
    return 0;
  };
  auto run_y = [&](int yis, int yie, int yjs, int yje) -> int {
    t_k.start();
    hipLaunchKernelGGL(adv_vflags_prep_kernel, dim3(64), dim3(256), 0, s, g, (const int *)domore_v, domore_v2,
                       (const int *)domore_k, yjs, yje);
    if (generic) {
      GenArgs a; gen_args(1, a);
      a.is = yis; a.ie = yie; a.js = yjs; a.je = yje; a.dom_in = domore_v; a.dom_new = domore_v2;
      const dim3 grid((yie - yis + 1 + 255) / 256, yje - yjs + 2, nz);
      if (cs->scheme == PLM) hipLaunchKernelGGL((gen_flux_kernel<1, PLM>), grid, dim3(256), 0, s, a);
      else if (cs->scheme == H3) hipLaunchKernelGGL((gen_flux_kernel<1, H3>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((gen_flux_kernel<1, CW>), grid, dim3(256), 0, s, a);
      hipLaunchKernelGGL(gen_update_kernel<1>, grid, dim3(256), 0, s, a);
      M6_HIP(hipGetLastError());
      tm.n_y++;
      { int *t = domore_v; domore_v = domore_v2; domore_v2 = t; }
      { const double ms = t_k.stop(); tm.ms_y += ms; if (itt == 1) tm.ms_y1 += ms; }
      return 0;
    }
    for (int grp = 0; grp < ngroups; grp++) {
      AdvArgs a; const int n = group_args(grp, a);
      a.is = yis; a.ie = yie; a.js = yjs; a.je = yje;
      a.domore_v_in = domore_v; a.domore_v_out = domore_v2;
      dim3 grid((yie - yis + 1 + 63) / 64, nz);
      // segments of at least 8 faces, at most YSEG_MAX waves per column strip
      const int nfaces = yje - yjs + 2;
      int nseg_max = 4;   // measured best on MI355X (8: one block per CU; 1-2: too few waves)
      if (const char *e = getenv("MOM6HIP_YSEG")) { int v = atoi(e); if (v >= 1 && v <= YSEG_MAX) nseg_max = v; }
      int nseg = nfaces / 8; if (nseg < 1) nseg = 1; if (nseg > nseg_max) nseg = nseg_max;
      const int seglen = (nfaces + nseg - 1) / nseg;
      switch (n) {
        case 1: launch_y<1>(cs->scheme, itt == 1, grid, nseg, seglen, s, a); break;
        case 2: launch_y<2>(cs->scheme, itt == 1, grid, nseg, seglen, s, a); break;
        case 3: launch_y<3>(cs->scheme, itt == 1, grid, nseg, seglen, s, a); break;
        default: launch_y<4>(cs->scheme, itt == 1, grid, nseg, seglen, s, a); break;
      }
      M6_HIP(hipGetLastError());
      tm.n_y++;
    }
    { int *t = domore_v; domore_v = domore_v2; domore_v2 = t; }
    { const double ms = t_k.stop(); tm.ms_y += ms; if (itt == 1) tm.ms_y1 += ms; }
    return 0;
  };

  int isv = is, iev = ie, jsv = js, jev = je;
  int halo_updates = 0, remaining = nz;
  for (itt = 1; itt <= max_iter; itt++) {
    if (isv > is - stencil) {
      // do_group_pass(CS%pass_uhr_vhr_t_hprev), :206
      t_k.start();
      {      // (one tile: the local wrap kernels; several: the native exchange or the host's group pass)
        std::vector<double *> flds = {uhr, vhr, hprev};
        std::vector<int32_t> pos = {MOM6HIP_POS_U, MOM6HIP_POS_V, MOM6HIP_POS_H}, nks = {nz, nz, nz};
        for (int m = 0; m < ntr; m++) { flds.push_back(d_tr[m]); pos.push_back(MOM6HIP_POS_H); nks.push_back(nz); }
        if (int rc = m6::group_pass(ctx, flds.data(), pos.data(), nks.data(), (int)flds.size())) return rc;
      }
      halo_updates++;
      tm.ms_halo += t_k.stop();

      int mh = is - g.isd;
      if (g.ied - ie < mh) mh = g.ied - ie;
      if (js - g.jsd < mh) mh = js - g.jsd;
      if (g.jed - je < mh) mh = g.jed - je;
      const int nsten_halo = mh / stencil;
      isv = is - nsten_halo * stencil; jsv = js - nsten_halo * stencil;
      iev = ie + nsten_halo * stencil; jev = je + nsten_halo * stencil;
      if ((nsten_halo > 1) || (itt == 1)) {
        t_k.start();
        ScanArgs sa;
        sa.g = g; sa.uhr = uhr; sa.vhr = vhr; sa.domore_u = domore_u; sa.domore_v = domore_v; sa.domore_k = domore_k;
        sa.ju0 = jsv; sa.ju1 = jev; sa.iu0 = isv + stencil - 1; sa.iu1 = iev - stencil;
        sa.jv0 = jsv + stencil - 1; sa.jv1 = jev - stencil; sa.iv0 = isv + stencil; sa.iv1 = iev - stencil;
        hipLaunchKernelGGL(adv_scan_kernel, dim3(g.njh + 1, nz), dim3(256), 0, s, sa);
        hipLaunchKernelGGL(adv_domore_k_kernel, dim3(nz), dim3(64), 0, s, g, (const int *)domore_u,
                           (const int *)domore_v, domore_k, jsv, jev, jsv + stencil - 1, jev - stencil);
        M6_HIP(hipGetLastError());
        tm.ms_setup += t_k.stop();
      }
    }

    isv += stencil; iev -= stencil; jsv += stencil; jev -= stencil;

    if (x_first) {
      if (run_x(isv, iev, jsv - stencil, jev + stencil)) return 1;
      if (run_y(isv, iev, jsv, jev)) return 1;
      hipLaunchKernelGGL(adv_domore_k_kernel, dim3(nz), dim3(64), 0, s, g, (const int *)domore_u,
                         (const int *)domore_v, domore_k, jsv - stencil, jev + stencil, jsv - 1, jev);
    } else {
      if (run_y(isv - stencil, iev + stencil, jsv, jev)) return 1;
      if (run_x(isv, iev, jsv, jev)) return 1;
      hipLaunchKernelGGL(adv_domore_k_kernel, dim3(nz), dim3(64), 0, s, g, (const int *)domore_u,
                         (const int *)domore_v, domore_k, jsv, jev, jsv - 1, jev);
    }
    M6_HIP(hipGetLastError());

    if (itt >= max_iter) break;
    if (isv > is - stencil) {
      // sum_across_PEs(domore_k), :305 -- the one host read-back per iteration
      if (m6::multi_tile(ctx)) {
        // domore_k becomes the global count and is what the next iteration tests (:215, :252): reduced where it lies (an all-reduce on the
        // communication stream with the native domain), the host's copy read back behind it
        if (int rc = m6::sum_across_PEs_dev(ctx, domore_k, nz, ctx->h_domore_k)) return rc;
      } else {
        M6_HIP(hipMemcpyAsync(ctx->h_domore_k, domore_k, nz * sizeof(int), hipMemcpyDeviceToHost, s));
        M6_HIP(hipStreamSynchronize(s));
      }
      remaining = 0;
      for (int k = 0; k < nz; k++) remaining += ctx->h_domore_k[k];
      if (remaining == 0) break;
    }
  }
  if (itt > max_iter) itt = max_iter;

  // ---- outputs ----
  if (memspace == MOM6HIP_MEM_HOST) {
    for (int m = 0; m < ntr; m++) M6_HIP(hipMemcpyAsync(tr[m], d_tr[m], bH, hipMemcpyDeviceToHost, s));
    if (uhr_out) M6_HIP(hipMemcpyAsync(uhr_out, uhr, bU, hipMemcpyDeviceToHost, s));
    if (vhr_out) M6_HIP(hipMemcpyAsync(vhr_out, vhr, bV, hipMemcpyDeviceToHost, s));
    if (vol_prev && update_vol_prev) M6_HIP(hipMemcpyAsync(vol_prev, hprev, bH, hipMemcpyDeviceToHost, s));
  } else {
    if (uhr_out) M6_HIP(hipMemcpyAsync(uhr_out, uhr, bU, hipMemcpyDeviceToDevice, s));
    if (vhr_out) M6_HIP(hipMemcpyAsync(vhr_out, vhr, bV, hipMemcpyDeviceToDevice, s));
    if (vol_prev && update_vol_prev) M6_HIP(hipMemcpyAsync(vol_prev, hprev, bH, hipMemcpyDeviceToDevice, s));
  }
  if (stats || memspace == MOM6HIP_MEM_HOST) {
    M6_HIP(hipMemcpyAsync(ctx->h_domore_k, domore_k, nz * sizeof(int), hipMemcpyDeviceToHost, s));
    M6_HIP(hipStreamSynchronize(s));
    remaining = 0;
    for (int k = 0; k < nz; k++) remaining += ctx->h_domore_k[k];
  }
  if (stats) { stats->iterations = itt; stats->halo_updates = halo_updates; stats->domore_remaining = remaining; }
  if (ctx->timing) { tm.ms_total = t_all.stop(); ctx->adv_timing = tm; }
  return 0;
}
