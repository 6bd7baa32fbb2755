// coriolis_adv.hip -- CorAdCalc + gradKE (src/core/MOM_CoriolisAdv.F90:125-965, :969-1051) as one fused
// gfx950 stencil kernel.
//
// Mapping: a 64 x 8 thread block owns a 64 x 8 patch of (I,J) points of one layer.  The potential
// vorticity q and abs_vort on the 65 x 9 q-points and KE on the 65 x 9 h-points the patch needs are
// computed once into LDS (q costs a division per point, so sharing it 4 ways matters), then every thread
// writes CAu(I,j) and CAv(i,J).  All global accesses are i-contiguous.  Everything is per-layer, so the
// reference's 2-D scratch arrays (q, Ih_q, KE, a, b, c, d ...) never exist in HBM.
// Algorithmic traffic: read u, v, h, uh, vh, write CAu, CAv = 56 B per cell.
#include <cmath>

#include "common.hpp"

namespace {

constexpr int TI = 64, TJ = 8;
#ifndef COR_KCH_DEF
#define COR_KCH_DEF 15
#endif
constexpr int COR_KCH = COR_KCH_DEF;      // layers a block works through with the 2-D metrics of its points in registers

struct CorArgs {
  m6::GridDev g;
  const double *u, *v, *h, *uh, *vh;
  double *CAu, *CAv;
  int scheme, ke_scheme, no_slip, bound;
  double vol_neglect;
  // the schemes beyond the production three (coradcalc_kernel<true>)
  int en_dis, upwind1;
  double Fe_m2, rat_lin, wt_lin_blend, eps_vel, h_tiny;
  // open boundaries (coradcalc_kernel<true>; null: OBC not associated): the reference's 2-D work arrays of every layer after the
  // segments have been projected onto them (cor_obc_* below) -- the circulation at the corner points, the layer volumes at the velocity
  // points, the centred transports of CORIOLIS_EN_DIS -- Area_q with the areas carried across the segments, and the faces of segments,
  // where gradKE leaves no gradient
  const double *o_dvdx, *o_dudy, *o_hArea_u, *o_hArea_v, *o_uh_center, *o_vh_center, *o_Area_q;
  const int32_t *o_seg_u, *o_seg_v;
  // the RK2 step's u_bc_accel = (CAu + PFu) + diffu formed here (mom6hip_ctx::BcAccelFuse; null: not asked for)
  const double *bc_PFu, *bc_PFv, *bc_diffu, *bc_diffv;
  double *bc_u, *bc_v;
  int bc_inviscid;
};

using m6::max2;
using m6::min2;
__device__ __forceinline__ double max4(double a, double b, double c, double d) { return max2(max2(max2(a, b), c), d); }
__device__ __forceinline__ double min4(double a, double b, double c, double d) { return min2(min2(min2(a, b), c), d); }

// The a / b / c / d weights and the ep_u / ep_v of ARAKAWA_LAMB81 (:534-542) and ARAKAWA_LAMB_BLEND (:543-590) that the
// reference forms in one loop pass (I, j) from the four q around the h-point (i, j): a(I-1,j), d(I-1,j), b(I,j), c(I,j),
// ep_u(i,j), ep_v(i,j).  q11 = q(I,J), q00 = q(I-1,J-1), q01 = q(I-1,J), q10 = q(I,J-1); ih.. the Ih_q at the same points.
struct Quad { double am1, dm1, b, c, epu, epv; };
__device__ __forceinline__ Quad al_quad(const CorArgs &p, double q11, double q00, double q01, double q10, double ih11, double ih00,
                                        double ih01, double ih10) {
  const double C1_24 = 1.0 / 24.0;
  Quad r;
  if (p.scheme == MOM6HIP_ARAKAWA_LAMB81) {
    r.am1 = (2.0 * (q11 + q00) + (q01 + q10)) * C1_24;
    r.dm1 = ((q11 + q00) + 2.0 * (q01 + q10)) * C1_24;
    r.b = ((q11 + q00) + 2.0 * (q01 + q10)) * C1_24;
    r.c = (2.0 * (q11 + q00) + (q01 + q10)) * C1_24;
    r.epu = ((q11 - q00) + (q01 - q10)) * C1_24;
    r.epv = (-(q11 - q00) + (q01 - q10)) * C1_24;
    return r;
  }
  const double min_Ihq = min4(ih00, ih10, ih01, ih11), max_Ihq = max4(ih00, ih10, ih01, ih11);
  double rat_m1 = 1.0e15;
  if (max_Ihq < 1.0e15 * min_Ihq) rat_m1 = max_Ihq / min_Ihq - 1.0;
  double AL_wt, Sad_wt;
  if (rat_m1 <= p.Fe_m2) AL_wt = 1.0;
  else if (rat_m1 < 1.5 * p.Fe_m2) AL_wt = 3.0 * p.Fe_m2 / rat_m1 - 2.0;
  else AL_wt = 0.0;
  if (rat_m1 <= 1.5 * p.Fe_m2) Sad_wt = 0.0;
  else if (rat_m1 <= p.rat_lin) Sad_wt = 1.0 - (1.5 * p.Fe_m2) / rat_m1;
  else if (rat_m1 < 2.0 * p.rat_lin) Sad_wt = 1.0 - (p.wt_lin_blend / p.rat_lin) * (rat_m1 - 2.0 * p.rat_lin);
  else Sad_wt = 1.0;
  r.am1 = Sad_wt * 0.25 * q01 + (1.0 - Sad_wt) * (((2.0 - AL_wt) * q01 + AL_wt * q10) + 2.0 * (q11 + q00)) * C1_24;
  r.dm1 = Sad_wt * 0.25 * q00 + (1.0 - Sad_wt) * (((2.0 - AL_wt) * q00 + AL_wt * q11) + 2.0 * (q01 + q10)) * C1_24;
  r.b = Sad_wt * 0.25 * q11 + (1.0 - Sad_wt) * (((2.0 - AL_wt) * q11 + AL_wt * q00) + 2.0 * (q01 + q10)) * C1_24;
  r.c = Sad_wt * 0.25 * q10 + (1.0 - Sad_wt) * (((2.0 - AL_wt) * q10 + AL_wt * q01) + 2.0 * (q11 + q00)) * C1_24;
  r.epu = AL_wt * ((q11 - q00) + (q01 - q10)) * C1_24;
  r.epv = AL_wt * (-(q11 - q00) + (q01 - q10)) * C1_24;
  return r;
}

// uh_min / uh_max (or vh_min / vh_max) of CORIOLIS_EN_DIS at one face (:594-642): hc = the centred estimate of the transport
// (uh_center / vh_center :328, :331), hm = the transport of the continuity solver, dL = dy_Cu / dx_Cv
__device__ __forceinline__ void en_dis_range_hc(double dL, double hc, double hm, double &tmin, double &tmax);
__device__ __forceinline__ void en_dis_range(double dL, double vel, double hsum, double hm, double &tmin, double &tmax) {
  en_dis_range_hc(dL, 0.5 * ((dL * 1.0) * vel) * hsum, hm, tmin, tmax);
}
// the same from the centred transport itself (with open boundaries it is read from the projected array)
__device__ __forceinline__ void en_dis_range_hc(double dL, double hc, double hm, double &tmin, double &tmax) {
  const double c1 = 1.0 - 1.5 * 0.5, c2 = 1.0 - 0.5, c3 = 2.0, slope = 0.5;
  if (dL == 0.0) hc = hm;
  if (fabs(hc) < 0.1 * fabs(hm)) {
    hm = 10.0 * hc;
  } else if (fabs(hc) > c1 * fabs(hm)) {
    if (fabs(hc) < c2 * fabs(hm)) hc = (3.0 * hc + (1.0 - c2 * 3.0) * hm);
    else if (fabs(hc) <= c3 * fabs(hm)) hc = hm;
    else hc = slope * hc + (1.0 - c3 * slope) * hm;
  }
  if (hc > hm) { tmin = hm; tmax = hc; } else { tmax = hm; tmin = hc; }
}

// Heff of ROBUST_ENSTRO (:692-703): |transport * IdL| / (eps_vel + |vel|), held between the two thicknesses of the face
__device__ __forceinline__ double robust_heff(double tr, double IdL, double vel, double ha, double hb, double eps_vel) {
  double He = fabs(tr * IdL) / (eps_vel + fabs(vel));
  He = max2(He, min2(ha, hb));
  He = min2(He, max2(ha, hb));
  return He;
}

// EXT = false: the three schemes of production runs (SADOURNY75_ENERGY, SADOURNY75_ENSTRO, ARAKAWA_HSU90), unchanged;
// EXT = true: also ROBUST_ENSTRO, ARAKAWA_LAMB81, ARAKAWA_LAMB_BLEND and CORIOLIS_EN_DIS (Ih_q staged in LDS as well).
template <bool EXT>
__global__ __launch_bounds__(TI * TJ) void coradcalc_kernel(CorArgs p) {
  const m6::GridDev &g = p.g;
  // (two sets of tiles, alternating from layer to layer: a layer's tiles are written while the previous layer's are still read, and ONE
  // barrier a layer -- between writing a set and reading it -- orders everything: nobody writes set b again before the barrier of the layer
  // in between, which every thread passes only after its reads of set b)
  __shared__ double sb_ihq[2][EXT ? TJ + 2 : 1][EXT ? TI + 2 : 1];      // Ih_q at the q-points of s_q (ARAKAWA_LAMB_BLEND)
  __shared__ double sb_q[2][TJ + 2][TI + 2];      // q(I0-1 .. I0+64, J0-1 .. J0+8) (the +1 column/row: Arakawa-Hsu)
  __shared__ double sb_av[2][TJ + 2][TI + 2];     // abs_vort, same points
  __shared__ double sb_ke[2][TJ + 1][TI + 1];     // KE(i = I0 .. I0+64, j = J0 .. J0+8)
  // A block works through COR_KCH consecutive layers of its tile: what a point reads of the 2-D metrics (15 values for the vorticity
  // and the layer volume at a q point, 5 for the kinetic energy at an h point, 2 for the results) is loaded ONCE into registers and
  // serves every layer of the chunk -- per layer and point that was 22 loads from L2 beside the 9 of the layer's own fields, in a
  // kernel bound by instruction issue and latency.  The layer loop is unrolled in full: left as a loop it costs more than it saves
  // (OM4 1440x1080x75, ms per call: 3.45 before; chunks of 5 / 15 unrolled 3.01 / 2.88; chunks of 7 / 15 as a loop 4.6 / 4.4).
  // The chunk index is the fastest grid dimension.
  const int I0 = g.isc - 1 + blockIdx.y * TI, J0 = g.jsc - 1 + blockIdx.z * TJ;
  const int tid = threadIdx.y * TI + threadIdx.x;
  // Index arithmetic in 32 bits from one base per point (the 2-D offsets fit easily): the 64-bit h2 / u2 / v2 / q2 of every
  // access were most of this kernel's instructions.
  const int nih = g.nih, sU = g.nih + 1;      // row strides: h- and v-point arrays; u- and q-point arrays
  const int Iqmax = g.iec + 1, Jqmax = g.jec + 1;      // q is defined on (Isq-1:Ieq+1, Jsq-1:Jeq+1)
  constexpr int NSLOT = ((TI + 2) * (TJ + 2) + TI * TJ - 1) / (TI * TJ);      // points of the (TI+2) x (TJ+2) frame per thread
  struct QMet { double A00, A10, A01, A11, Area_q, dyCv1, dyCv0, dxCu1, dxCu0, fac, IareaBu, Cor; };
  struct KMet { double aCuE, aCuW, aCvN, aCvS, IareaT; };
  QMet qm[NSLOT]; KMet km[NSLOT];
  bool q_in[NSLOT], k_in[NSLOT];
#pragma unroll
  for (int sl = 0; sl < NSLOT; sl++) {
    const int t = tid + sl * TI * TJ;
    const int fy = t / (TI + 2), fx = t - fy * (TI + 2);
    q_in[sl] = false; k_in[sl] = false;
    qm[sl] = QMet{0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0.}; km[sl] = KMet{0., 0., 0., 0., 0.};
    if (t < (TI + 2) * (TJ + 2)) {
      const int I = I0 - 1 + fx, J = J0 - 1 + fy;
      if (I <= Iqmax && J <= Jqmax) {
        q_in[sl] = true;
        const int oh = (I - g.isd) + nih * (J - g.jsd), ou = (I - g.isd + 1) + sU * (J - g.jsd);
        const int ov = (I - g.isd) + nih * (J - g.jsd + 1), oq = (I - g.isd + 1) + sU * (J - g.jsd + 1);
        QMet &m = qm[sl];
        m.A00 = g.mask2dT[oh] * g.areaT[oh];
        m.A10 = g.mask2dT[oh + 1] * g.areaT[oh + 1];
        m.A01 = g.mask2dT[oh + nih] * g.areaT[oh + nih];
        m.A11 = g.mask2dT[oh + nih + 1] * g.areaT[oh + nih + 1];
        m.Area_q = (m.A00 + m.A11) + (m.A10 + m.A01);
        m.dyCv1 = g.dyCv[ov + 1]; m.dyCv0 = g.dyCv[ov]; m.dxCu1 = g.dxCu[ou + sU]; m.dxCu0 = g.dxCu[ou];
        const double mB = g.mask2dBu[oq];
        m.fac = p.no_slip ? (2.0 - mB) : mB;
        m.IareaBu = g.IareaBu[oq]; m.Cor = g.CoriolisBu[oq];
      }
      if (fx <= TI && fy <= TJ) {
        const int i = I0 + fx, j = J0 + fy;
        if (i <= g.iec + 1 && j <= g.jec + 1) {
          k_in[sl] = true;
          const int oh = (i - g.isd) + nih * (j - g.jsd), ou = (i - g.isd + 1) + sU * (j - g.jsd), ov = (i - g.isd) + nih * (j - g.jsd + 1);
          km[sl] = KMet{g.areaCu[ou], g.areaCu[ou - 1], g.areaCv[ov], g.areaCv[ov - nih], g.IareaT[oh]};
        }
      }
    }
  }
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int I = I0 + tx, J = J0 + ty;           // this thread's u-point (I, j=J) and v-point (i=I, J)
  const int ou = (I - g.isd + 1) + sU * (J - g.jsd), ov = (I - g.isd) + nih * (J - g.jsd + 1);      // u2(I, J), v2(I, J)
  const bool do_u = J >= g.jsc && J <= g.jec && I <= g.iec, do_v = I >= g.isc && I <= g.iec && J <= g.jec;
  const double IdxCu_r = do_u ? g.IdxCu[ou] : 0.0, IdyCv_r = do_v ? g.IdyCv[ov] : 0.0;

#pragma unroll
  for (int kk = 0; kk < COR_KCH; kk++) {
  const int k = blockIdx.x * COR_KCH + kk;
  if (k >= g.nk) break;      // (uniform over the block)
  double (*s_ihq)[EXT ? TI + 2 : 1] = sb_ihq[kk & 1];
  double (*s_q)[TI + 2] = sb_q[kk & 1];
  double (*s_av)[TI + 2] = sb_av[kk & 1];
  double (*s_ke)[TI + 1] = sb_ke[kk & 1];
  // uk / vk / hk ...: the layer's planes (wave-uniform pointers)
  const long kH = (long)g.nih * g.njh * k, kU = (long)(g.nih + 1) * g.njh * k, kV = (long)g.nih * (g.njh + 1) * k;
  const double *__restrict__ hk = p.h + kH, *__restrict__ uk = p.u + kU, *__restrict__ vk = p.v + kV;
  const double *__restrict__ uhk = p.uh + kU, *__restrict__ vhk = p.vh + kV;

#pragma unroll
  for (int sl = 0; sl < NSLOT; sl++) {
    const int t = tid + sl * TI * TJ;
    if (t >= (TI + 2) * (TJ + 2)) break;
    const int ty = t / (TI + 2), tx = t - ty * (TI + 2);
    // ---- q-point (I,J) = (I0-1+tx, J0-1+ty), :246-274, :314-324, :459-491 ----
    {
      const int I = I0 - 1 + tx, J = J0 - 1 + ty;
      double qv = 0.0, av = 0.0, ihq = 0.0;
      if (q_in[sl]) {
        const int oh = (I - g.isd) + nih * (J - g.jsd);               // h2(i, j)
        const int ou = (I - g.isd + 1) + sU * (J - g.jsd);            // u2(I, j)
        const int ov = (I - g.isd) + nih * (J - g.jsd + 1);           // v2(i, J)
        const QMet &m = qm[sl];
        const double A00 = m.A00, A10 = m.A10, A01 = m.A01, A11 = m.A11;
        double Area_q = m.Area_q;
        double dvdx, dudy, hArea_u0, hArea_u1, hArea_v0, hArea_v1;
        if (EXT && p.o_dvdx) {      // open boundaries: the work arrays with the segments projected onto them
          const int oq = ou + sU;                                     // q2(I, J)
          const long kQ = (long)sU * (g.njh + 1) * k;
          dvdx = p.o_dvdx[kQ + oq]; dudy = p.o_dudy[kQ + oq];
          hArea_u0 = p.o_hArea_u[kU + ou]; hArea_u1 = p.o_hArea_u[kU + ou + sU];
          hArea_v0 = p.o_hArea_v[kV + ov]; hArea_v1 = p.o_hArea_v[kV + ov + 1];
          Area_q = p.o_Area_q[oq];
        } else {
        dvdx = (vk[ov + 1] * m.dyCv1 - vk[ov] * m.dyCv0);
        dudy = (uk[ou + sU] * m.dxCu1 - uk[ou] * m.dxCu0);
        const double h00 = hk[oh], h10 = hk[oh + 1], h01 = hk[oh + nih], h11 = hk[oh + nih + 1];
        hArea_u0 = 0.5 * (A00 * h00 + A10 * h10);     // hArea_u(I,j)
        hArea_u1 = 0.5 * (A01 * h01 + A11 * h11);     // hArea_u(I,j+1)
        hArea_v0 = 0.5 * (A00 * h00 + A01 * h01);     // hArea_v(i,J)
        hArea_v1 = 0.5 * (A10 * h10 + A11 * h11);     // hArea_v(i+1,J)
        }
        const double rel_vort = m.fac * (dvdx - dudy) * m.IareaBu;
        av = m.Cor + rel_vort;
        const double hArea_q = (hArea_u0 + hArea_u1) + (hArea_v0 + hArea_v1);
        const double Ih_q = Area_q / (hArea_q + p.vol_neglect);
        qv = av * Ih_q; ihq = Ih_q;
      }
      s_q[ty][tx] = qv; s_av[ty][tx] = av;
      if (EXT) s_ihq[ty][tx] = ihq;
    }
    // ---- h-point (i,j) = (I0+tx, J0+ty): KE, :995-1025 ----
    if (tx <= TI && ty <= TJ) {
      const int i = I0 + tx, j = J0 + ty;
      double ke = 0.0;
      if (k_in[sl]) {
        const int ou = (i - g.isd + 1) + sU * (j - g.jsd), ov = (i - g.isd) + nih * (j - g.jsd + 1);
        const double uE = uk[ou], uW = uk[ou - 1], vN = vk[ov], vS = vk[ov - nih];
        const KMet &m = km[sl];
        if (p.ke_scheme == MOM6HIP_KE_ARAKAWA) {
          ke = ((m.aCuE * (uE * uE) + m.aCuW * (uW * uW)) +
                (m.aCvN * (vN * vN) + m.aCvS * (vS * vS))) * 0.25 * m.IareaT;
        } else {
          const double up = 0.5 * (uW + fabs(uW)), um = 0.5 * (uE - fabs(uE));
          const double vp = 0.5 * (vS + fabs(vS)), vm = 0.5 * (vN - fabs(vN));
          if (p.ke_scheme == MOM6HIP_KE_SIMPLE_GUDONOV) {
            ke = (max2(up * up, um * um) + max2(vp * vp, vm * vm)) * 0.5;
          } else {
            const double up2a = up * up * m.aCuW, um2a = um * um * m.aCuE;
            const double vp2a = vp * vp * m.aCvS, vm2a = vm * vm * m.aCvN;
            ke = (max2(um2a, up2a) + max2(vm2a, vp2a)) * 0.5 * m.IareaT;
          }
        }
      }
      s_ke[ty][tx] = ke;
    }
  }
  // the transports the second half reads, asked for before the barrier so that they travel while the block gathers at it
  double vh_ne = 0., vh_nw = 0., vh_sw = 0., vh_se = 0., uh_sw = 0., uh_nw = 0., uh_se = 0., uh_ne = 0.;
  if (do_u) { vh_ne = vhk[ov + 1]; vh_nw = vhk[ov]; vh_sw = vhk[ov - nih]; vh_se = vhk[ov - nih + 1]; }
  if (do_v) { uh_sw = uhk[ou - 1]; uh_nw = uhk[ou - 1 + sU]; uh_se = uhk[ou]; uh_ne = uhk[ou + sU]; }
  __syncthreads();

  // tile coordinates: q(I,J) = s_q[ty+1][tx+1]; KE(i,j) = s_ke[ty][tx]
  const double C1_12 = 1.0 / 12.0;
  // ---- CAu(I, j), j = J >= jsc, I <= iec : :644-752 ----
  if (do_u) {
    const double qN = s_q[ty + 1][tx + 1], qS = s_q[ty][tx + 1];           // q(I,J), q(I,J-1)
    const double IdxCu = IdxCu_r;
    double ca;
    const bool al = EXT && (p.scheme == MOM6HIP_ARAKAWA_LAMB81 || p.scheme == MOM6HIP_AL_BLEND);
    if (EXT && p.scheme == MOM6HIP_SADOURNY75_ENERGY && p.en_dis) {      // :645-665
      const int oh = (I - g.isd) + nih * (J - g.jsd);                     // h2(i, j)
      const double uI = uk[ou];
      // vh_min / vh_max at the faces (i, J), (i+1, J), (i, J-1), (i+1, J-1)
      double mnN0, mxN0, mnN1, mxN1, mnS0, mxS0, mnS1, mxS1;
      if (p.o_vh_center) {
        const double *vc = p.o_vh_center + kV;
        en_dis_range_hc(g.dx_Cv[ov], vc[ov], vh_nw, mnN0, mxN0);
        en_dis_range_hc(g.dx_Cv[ov + 1], vc[ov + 1], vh_ne, mnN1, mxN1);
        en_dis_range_hc(g.dx_Cv[ov - nih], vc[ov - nih], vh_sw, mnS0, mxS0);
        en_dis_range_hc(g.dx_Cv[ov - nih + 1], vc[ov - nih + 1], vh_se, mnS1, mxS1);
      } else {
      en_dis_range(g.dx_Cv[ov], vk[ov], hk[oh] + hk[oh + nih], vh_nw, mnN0, mxN0);
      en_dis_range(g.dx_Cv[ov + 1], vk[ov + 1], hk[oh + 1] + hk[oh + nih + 1], vh_ne, mnN1, mxN1);
      en_dis_range(g.dx_Cv[ov - nih], vk[ov - nih], hk[oh - nih] + hk[oh], vh_sw, mnS0, mxS0);
      en_dis_range(g.dx_Cv[ov - nih + 1], vk[ov - nih + 1], hk[oh - nih + 1] + hk[oh + 1], vh_se, mnS1, mxS1);
      }
      double temp1, temp2;
      if (qN * uI == 0.0) temp1 = qN * ((mxN0 + mxN1) + (mnN0 + mnN1)) * 0.5;
      else if (qN * uI < 0.0) temp1 = qN * (mxN0 + mxN1);
      else temp1 = qN * (mnN0 + mnN1);
      if (qS * uI == 0.0) temp2 = qS * ((mxS0 + mxS1) + (mnS0 + mnS1)) * 0.5;
      else if (qS * uI < 0.0) temp2 = qS * (mxS0 + mxS1);
      else temp2 = qS * (mnS0 + mnS1);
      ca = 0.25 * IdxCu * (temp1 + temp2);
    } else if (p.scheme == MOM6HIP_SADOURNY75_ENERGY) {
      ca = 0.25 * (qN * (vh_ne + vh_nw) + qS * (vh_sw + vh_se)) * IdxCu;
    } else if (p.scheme == MOM6HIP_SADOURNY75_ENSTRO) {
      ca = 0.125 * (IdxCu * (qN + qS)) * ((vh_ne + vh_nw) + (vh_sw + vh_se));
    } else if (EXT && p.scheme == MOM6HIP_ROBUST_ENSTRO) {      // :687-714
      const int oh = (I - g.isd) + nih * (J - g.jsd);
      const double Heff1 = robust_heff(vh_nw, g.IdxCv[ov], vk[ov], hk[oh], hk[oh + nih], p.eps_vel);
      const double Heff2 = robust_heff(vh_sw, g.IdxCv[ov - nih], vk[ov - nih], hk[oh - nih], hk[oh], p.eps_vel);
      const double Heff3 = robust_heff(vh_ne, g.IdxCv[ov + 1], vk[ov + 1], hk[oh + 1], hk[oh + nih + 1], p.eps_vel);
      const double Heff4 = robust_heff(vh_se, g.IdxCv[ov - nih + 1], vk[ov - nih + 1], hk[oh - nih + 1], hk[oh + 1], p.eps_vel);
      const double avN = s_av[ty + 1][tx + 1], avS = s_av[ty][tx + 1];
      if (!p.upwind1) {
        ca = 0.5 * (avN + avS) * ((vh_nw + vh_se) + (vh_sw + vh_ne)) / (p.h_tiny + ((Heff1 + Heff4) + (Heff2 + Heff3))) * IdxCu;
      } else {
        const double VHeff = ((vh_nw + vh_se) + (vh_sw + vh_ne));
        const double QVHeff = 0.5 * ((avN + avS) * VHeff - (avN - avS) * fabs(VHeff));
        ca = (QVHeff / (p.h_tiny + ((Heff1 + Heff4) + (Heff2 + Heff3)))) * IdxCu;
      }
    } else if (al) {      // ARAKAWA_LAMB81 / _BLEND: b, c, ep_u(i) of the point's own pass, a, d, ep_u(i+1) of the next one's
      const Quad q0 = al_quad(p, s_q[ty + 1][tx + 1], s_q[ty][tx], s_q[ty + 1][tx], s_q[ty][tx + 1], s_ihq[ty + 1][tx + 1], s_ihq[ty][tx],
                              s_ihq[ty + 1][tx], s_ihq[ty][tx + 1]);
      const Quad q1 = al_quad(p, s_q[ty + 1][tx + 2], s_q[ty][tx + 1], s_q[ty + 1][tx + 1], s_q[ty][tx + 2], s_ihq[ty + 1][tx + 2],
                              s_ihq[ty][tx + 1], s_ihq[ty + 1][tx + 1], s_ihq[ty][tx + 2]);
      ca = ((q1.am1 * vh_ne + q0.c * vh_sw) + (q0.b * vh_nw + q1.dm1 * vh_se)) * IdxCu;
      ca = ca + (q0.epu * uhk[ou - 1] - q1.epu * uhk[ou + 1]) * IdxCu;      // :717-721
    } else {   // ARAKAWA_HSU90, :526-531 and :684-685
      const double qE = s_q[ty + 1][tx + 2], qSE = s_q[ty][tx + 2], qW = s_q[ty + 1][tx], qSW = s_q[ty][tx];
      const double a = (qN + (qE + qS)) * C1_12;
      const double d = ((qN + qSE) + qS) * C1_12;
      const double b = (qN + (qW + qS)) * C1_12;
      const double c = ((qN + qSW) + qS) * C1_12;
      ca = ((a * vh_ne + c * vh_sw) + (b * vh_nw + d * vh_se)) * IdxCu;
    }
    if (p.bound) {
      const double avN = s_av[ty + 1][tx + 1], avS = s_av[ty][tx + 1];
      const double fv1 = avN * vk[ov + 1], fv2 = avN * vk[ov];
      const double fv3 = avS * vk[ov - nih + 1], fv4 = avS * vk[ov - nih];
      ca = min2(ca, max4(fv1, fv2, fv3, fv4));
      ca = max2(ca, min4(fv1, fv2, fv3, fv4));
    }
    double KEx = (s_ke[ty][tx + 1] - s_ke[ty][tx]) * IdxCu;
    if (EXT && p.o_seg_u && p.o_seg_u[ou]) KEx = 0.;      // gradKE :1037-1050
    const double cau = ca - KEx;
    p.CAu[kU + ou] = cau;
    if (p.bc_u) {
      double a = cau + p.bc_PFu[kU + ou];
      if (p.bc_inviscid) a = (a == 0.0) ? 0.0 : a; else a = a + p.bc_diffu[kU + ou];
      p.bc_u[kU + ou] = a;
    }
  }
  // ---- CAv(i, J), i = I >= isc, J <= jec : :763-876 ----
  if (do_v) {
    const double qE = s_q[ty + 1][tx + 1], qW = s_q[ty + 1][tx];           // q(I,J), q(I-1,J)
    const double IdyCv = IdyCv_r;
    double ca;
    const bool al = EXT && (p.scheme == MOM6HIP_ARAKAWA_LAMB81 || p.scheme == MOM6HIP_AL_BLEND);
    if (EXT && p.scheme == MOM6HIP_SADOURNY75_ENERGY && p.en_dis) {      // :764-785
      const int oh = (I - g.isd) + nih * (J - g.jsd);                     // h2(i, j)
      const double vJ = vk[ov];
      // uh_min / uh_max at the faces (I-1, j), (I-1, j+1), (I, j), (I, j+1)
      double mnW0, mxW0, mnW1, mxW1, mnE0, mxE0, mnE1, mxE1;
      if (p.o_uh_center) {
        const double *uc = p.o_uh_center + kU;
        en_dis_range_hc(g.dy_Cu[ou - 1], uc[ou - 1], uh_sw, mnW0, mxW0);
        en_dis_range_hc(g.dy_Cu[ou - 1 + sU], uc[ou - 1 + sU], uh_nw, mnW1, mxW1);
        en_dis_range_hc(g.dy_Cu[ou], uc[ou], uh_se, mnE0, mxE0);
        en_dis_range_hc(g.dy_Cu[ou + sU], uc[ou + sU], uh_ne, mnE1, mxE1);
      } else {
      en_dis_range(g.dy_Cu[ou - 1], uk[ou - 1], hk[oh - 1] + hk[oh], uh_sw, mnW0, mxW0);
      en_dis_range(g.dy_Cu[ou - 1 + sU], uk[ou - 1 + sU], hk[oh + nih - 1] + hk[oh + nih], uh_nw, mnW1, mxW1);
      en_dis_range(g.dy_Cu[ou], uk[ou], hk[oh] + hk[oh + 1], uh_se, mnE0, mxE0);
      en_dis_range(g.dy_Cu[ou + sU], uk[ou + sU], hk[oh + nih] + hk[oh + nih + 1], uh_ne, mnE1, mxE1);
      }
      double temp1, temp2;
      if (qW * vJ == 0.0) temp1 = qW * ((mxW0 + mxW1) + (mnW0 + mnW1)) * 0.5;
      else if (qW * vJ > 0.0) temp1 = qW * (mxW0 + mxW1);
      else temp1 = qW * (mnW0 + mnW1);
      if (qE * vJ == 0.0) temp2 = qE * ((mxE0 + mxE1) + (mnE0 + mnE1)) * 0.5;
      else if (qE * vJ > 0.0) temp2 = qE * (mxE0 + mxE1);
      else temp2 = qE * (mnE0 + mnE1);
      ca = -0.25 * IdyCv * (temp1 + temp2);
    } else if (p.scheme == MOM6HIP_SADOURNY75_ENERGY) {
      ca = -0.25 * (qW * (uh_sw + uh_nw) + qE * (uh_se + uh_ne)) * IdyCv;
    } else if (p.scheme == MOM6HIP_SADOURNY75_ENSTRO) {
      ca = -0.125 * (IdyCv * (qW + qE)) * ((uh_sw + uh_nw) + (uh_se + uh_ne));
    } else if (EXT && p.scheme == MOM6HIP_ROBUST_ENSTRO) {      // :808-838
      const int oh = (I - g.isd) + nih * (J - g.jsd);
      const double Heff1 = robust_heff(uh_se, g.IdyCu[ou], uk[ou], hk[oh], hk[oh + 1], p.eps_vel);
      const double Heff2 = robust_heff(uh_sw, g.IdyCu[ou - 1], uk[ou - 1], hk[oh - 1], hk[oh], p.eps_vel);
      const double Heff3 = robust_heff(uh_ne, g.IdyCu[ou + sU], uk[ou + sU], hk[oh + nih], hk[oh + nih + 1], p.eps_vel);
      const double Heff4 = robust_heff(uh_nw, g.IdyCu[ou - 1 + sU], uk[ou - 1 + sU], hk[oh + nih - 1], hk[oh + nih], p.eps_vel);
      const double avE = s_av[ty + 1][tx + 1], avW = s_av[ty + 1][tx];
      if (!p.upwind1) {
        ca = -0.5 * (avE + avW) * ((uh_se + uh_nw) + (uh_sw + uh_ne)) / (p.h_tiny + ((Heff1 + Heff4) + (Heff2 + Heff3))) * IdyCv;
      } else {
        const double UHeff = ((uh_se + uh_nw) + (uh_sw + uh_ne));
        const double QUHeff = 0.5 * ((avE + avW) * UHeff - (avE - avW) * fabs(UHeff));
        ca = -QUHeff / (p.h_tiny + ((Heff1 + Heff4) + (Heff2 + Heff3))) * IdyCv;
      }
    } else if (al) {      // a(I-1,j), b(I,j), ep_v(i,j) of the pass (I, j = J); c(I,j+1), d(I-1,j+1), ep_v(i,j+1) of (I, J+1)
      const Quad q0 = al_quad(p, s_q[ty + 1][tx + 1], s_q[ty][tx], s_q[ty + 1][tx], s_q[ty][tx + 1], s_ihq[ty + 1][tx + 1], s_ihq[ty][tx],
                              s_ihq[ty + 1][tx], s_ihq[ty][tx + 1]);
      const Quad q1 = al_quad(p, s_q[ty + 2][tx + 1], s_q[ty + 1][tx], s_q[ty + 2][tx], s_q[ty + 1][tx + 1], s_ihq[ty + 2][tx + 1],
                              s_ihq[ty + 1][tx], s_ihq[ty + 2][tx], s_ihq[ty + 1][tx + 1]);
      ca = -((q0.am1 * uh_sw + q1.c * uh_ne) + (q0.b * uh_se + q1.dm1 * uh_nw)) * IdyCv;
      ca = ca + (q0.epv * vhk[ov - nih] - q1.epv * vhk[ov + nih]) * IdyCv;      // :841-845
    } else {   // ARAKAWA_HSU90: a(I-1,j), c(I,j+1), b(I,j), d(I-1,j+1)
      // q(I,J) = s_q[ty+1][tx+1]
      auto Q = [&](int dI, int dJ) { return s_q[ty + 1 + dJ][tx + 1 + dI]; };
      const double a_ = (Q(-1, 0) + (Q(0, 0) + Q(-1, -1))) * C1_12;             // a(I-1,j): q(I-1,J)+(q(I,J)+q(I-1,J-1))
      const double c_ = ((Q(0, 1) + Q(-1, 0)) + Q(0, 0)) * C1_12;               // c(I,j+1): (q(I,J+1)+q(I-1,J))+q(I,J)
      const double b_ = (Q(0, 0) + (Q(-1, 0) + Q(0, -1))) * C1_12;              // b(I,j)
      const double d_ = ((Q(-1, 1) + Q(0, 0)) + Q(-1, 0)) * C1_12;              // d(I-1,j+1): (q(I-1,J+1)+q(I,J))+q(I-1,J)
      ca = -((a_ * uh_sw + c_ * uh_ne) + (b_ * uh_se + d_ * uh_nw)) * IdyCv;
    }
    if (p.bound) {
      const double avE = s_av[ty + 1][tx + 1], avW = s_av[ty + 1][tx];
      const double fu1 = -avE * uk[ou + sU], fu2 = -avE * uk[ou];
      const double fu3 = -avW * uk[ou - 1 + sU], fu4 = -avW * uk[ou - 1];
      ca = min2(ca, max4(fu1, fu2, fu3, fu4));
      ca = max2(ca, min4(fu1, fu2, fu3, fu4));
    }
    double KEy = (s_ke[ty + 1][tx] - s_ke[ty][tx]) * IdyCv;
    if (EXT && p.o_seg_v && p.o_seg_v[ov]) KEy = 0.;
    const double cav = ca - KEy;
    p.CAv[kV + ov] = cav;
    if (p.bc_v) {
      double a = cav + p.bc_PFv[kV + ov];
      if (p.bc_inviscid) a = (a == 0.0) ? 0.0 : a; else a = a + p.bc_diffv[kV + ov];
      p.bc_v[kV + ov] = a;
    }
  }
  }      // the layers of the chunk
}

// ---- open boundaries: the reference's 2-D work arrays of every layer, with the segments projected onto them in the reference's order ----
// (a small-configuration path: plain kernels, one thread a point; the segments are applied by one launch each, in sequence, because a
// later segment overwrites what an earlier one left)
struct CorObc {
  m6::GridDev g;
  const double *u, *v, *h;
  double *Area_h, *Area_q, *dvdx, *dudy, *hArea_u, *hArea_v, *uh_center, *vh_center;      // (uh_center, vh_center null without CORIOLIS_EN_DIS)
  int zero_vorticity, freeslip_vorticity, computed_vorticity, specified_vorticity;
};
struct CorSeg {      // one segment (mom6hip_obc_segment_t with device pointers)
  int direction, is_N_or_S, is_E_or_W, pad;
  int IsdB, IedB, JsdB, JedB, isd, ied, jsd, jed;
  const double *tangential_vel, *tangential_grad;
};
__device__ __forceinline__ double cor_seg_q(const CorSeg &S, const double *f, int I, int J, int k) {
  const long nI = S.IedB - S.IsdB + 1, nJ = S.JedB - S.JsdB + 1;
  return f[(I - S.IsdB) + nI * ((J - S.JsdB) + nJ * (long)k)];
}

// :246-248 (what = 0) and :271-274 (what = 1) over the ranges of the reference
__global__ __launch_bounds__(256) void cor_obc_area_kernel(CorObc a, int what) {
  const m6::GridDev &g = a.g;
  const int i = g.isc - 2 + blockIdx.x * 256 + threadIdx.x, j = g.jsc - 2 + blockIdx.y;
  if (what == 0) { if (i <= g.iec + 2) a.Area_h[g.h2(i, j)] = g.mask2dT[g.h2(i, j)] * g.areaT[g.h2(i, j)]; return; }
  if (i <= g.iec + 1 && j <= g.jec + 1)
    a.Area_q[g.q2(i, j)] = (a.Area_h[g.h2(i, j)] + a.Area_h[g.h2(i + 1, j + 1)]) + (a.Area_h[g.h2(i + 1, j)] + a.Area_h[g.h2(i, j + 1)]);
}

// the layer's work arrays :314-333; thread (i, j, k) over (Isq-1 : Ieq+2, Jsq-1 : Jeq+2)
__global__ __launch_bounds__(256) void cor_obc_fields_kernel(CorObc a) {
  const m6::GridDev &g = a.g;
  const int i = g.isc - 2 + blockIdx.x * 256 + threadIdx.x, j = g.jsc - 2 + blockIdx.y, k = blockIdx.z;
  if (i > g.iec + 2) return;
  const int I = i, J = j, Isq = g.isc - 1, Ieq = g.iec, Jsq = g.jsc - 1, Jeq = g.jec;
  const double *u = a.u + (long)(g.nih + 1) * g.njh * k, *v = a.v + (long)g.nih * (g.njh + 1) * k, *h = a.h + (long)g.nih * g.njh * k;
  const long kQ = (long)(g.nih + 1) * (g.njh + 1) * k, kU = (long)(g.nih + 1) * g.njh * k, kV = (long)g.nih * (g.njh + 1) * k;
  if (I <= Ieq + 1 && J <= Jeq + 1) {
    a.dvdx[kQ + g.q2(I, J)] = (v[g.v2(i + 1, J)] * g.dyCv[g.v2(i + 1, J)] - v[g.v2(i, J)] * g.dyCv[g.v2(i, J)]);
    a.dudy[kQ + g.q2(I, J)] = (u[g.u2(I, j + 1)] * g.dxCu[g.u2(I, j + 1)] - u[g.u2(I, j)] * g.dxCu[g.u2(I, j)]);
  }
  if (J <= Jeq + 1) a.hArea_v[kV + g.v2(i, J)] = 0.5 * (a.Area_h[g.h2(i, j)] * h[g.h2(i, j)] + a.Area_h[g.h2(i, j + 1)] * h[g.h2(i, j + 1)]);
  if (I <= Ieq + 1) a.hArea_u[kU + g.u2(I, j)] = 0.5 * (a.Area_h[g.h2(i, j)] * h[g.h2(i, j)] + a.Area_h[g.h2(i + 1, j)] * h[g.h2(i + 1, j)]);
  if (a.uh_center) {
    if (j >= Jsq && j <= Jeq + 1 && I >= g.isc - 1 && I <= g.iec)
      a.uh_center[kU + g.u2(I, j)] = 0.5 * ((g.dy_Cu[g.u2(I, j)] * 1.0) * u[g.u2(I, j)]) * (h[g.h2(i, j)] + h[g.h2(i + 1, j)]);
    if (J >= g.jsc - 1 && J <= g.jec && i >= Isq && i <= Ieq + 1)
      a.vh_center[kV + g.v2(i, J)] = 0.5 * ((g.dx_Cv[g.v2(i, J)] * 1.0) * v[g.v2(i, J)]) * (h[g.h2(i, j)] + h[g.h2(i, j + 1)]);
  }
}

// one segment; thread t along it (corner points IsdB : IedB or JsdB : JedB: they span its cells too), blockIdx.y = k (phases 1, 2).
// phase 0 :249-269 the areas; 1 :337-420 the velocity points; 2 :422-455 the corner points
__global__ __launch_bounds__(64) void cor_obc_segment_kernel(CorObc a, CorSeg S, int phase) {
  const m6::GridDev &g = a.g;
  const int Isq = g.isc - 1, Ieq = g.iec, Jsq = g.jsc - 1, Jeq = g.jec;
  const int k = blockIdx.y;
  const bool ns = S.is_N_or_S && (S.JsdB >= Jsq - 1) && (S.JsdB <= Jeq + 1);
  const bool ew = !ns && S.is_E_or_W && (S.IsdB >= Isq - 1) && (S.IsdB <= Ieq + 1);      // (the reference's elseif)
  if (S.is_N_or_S && !ns) return;
  if (!ns && !ew) return;
  const int t = (ns ? S.IsdB : S.JsdB) + blockIdx.x * 64 + threadIdx.x;
  if (t > (ns ? S.IedB : S.JedB)) return;
  const double *u = a.u + (long)(g.nih + 1) * g.njh * k, *v = a.v + (long)g.nih * (g.njh + 1) * k, *h = a.h + (long)g.nih * g.njh * k;
  const long kQ = (long)(g.nih + 1) * (g.njh + 1) * k, kU = (long)(g.nih + 1) * g.njh * k, kV = (long)g.nih * (g.njh + 1) * k;
  double *Ah = a.Area_h;
  if (ns) {
    const int J = S.JsdB, j = J, I = t, i = t;
    const bool N = S.direction == MOM6HIP_OBC_DIRECTION_N;
    if (phase == 0) {
      if (i >= max(Isq - 1, S.isd) && i <= min(Ieq + 2, S.ied)) { if (N) Ah[g.h2(i, j + 1)] = Ah[g.h2(i, j)]; else Ah[g.h2(i, j)] = Ah[g.h2(i, j + 1)]; }
    } else if (phase == 1) {
      double *dvdx = a.dvdx + kQ, *dudy = a.dudy + kQ;
      if (a.zero_vorticity) { dvdx[g.q2(I, J)] = 0.; dudy[g.q2(I, J)] = 0.; }
      if (a.freeslip_vorticity) dudy[g.q2(I, J)] = 0.;
      if (a.computed_vorticity) {
        if (N) dudy[g.q2(I, J)] = 2.0 * (cor_seg_q(S, S.tangential_vel, I, J, k) - u[g.u2(I, j)]) * g.dxCu[g.u2(I, j)];
        else dudy[g.q2(I, J)] = 2.0 * (u[g.u2(I, j + 1)] - cor_seg_q(S, S.tangential_vel, I, J, k)) * g.dxCu[g.u2(I, j + 1)];
      }
      if (a.specified_vorticity) {
        if (N) dudy[g.q2(I, J)] = cor_seg_q(S, S.tangential_grad, I, J, k) * g.dxCu[g.u2(I, j)] * g.dyBu[g.q2(I, J)];
        else dudy[g.q2(I, J)] = cor_seg_q(S, S.tangential_grad, I, J, k) * g.dxCu[g.u2(I, j + 1)] * g.dyBu[g.q2(I, J)];
      }
      if (i >= max(Isq - 1, S.isd) && i <= min(Ieq + 2, S.ied)) {
        const double hi = N ? h[g.h2(i, j)] : h[g.h2(i, j + 1)];
        a.hArea_v[kV + g.v2(i, J)] = 0.5 * (Ah[g.h2(i, j)] + Ah[g.h2(i, j + 1)]) * hi;
        if (a.vh_center) a.vh_center[kV + g.v2(i, J)] = (g.dx_Cv[g.v2(i, J)] * 1.0) * v[g.v2(i, J)] * hi;
      }
    } else {
      if (I >= max(Isq - 1, S.IsdB) && I <= min(Ieq + 1, S.IedB)) {
        double *hu = a.hArea_u + kU;
        if (N) {
          if (Ah[g.h2(i, j)] + Ah[g.h2(i + 1, j)] > 0.0)
            hu[g.u2(I, j + 1)] = hu[g.u2(I, j)] * ((Ah[g.h2(i, j + 1)] + Ah[g.h2(i + 1, j + 1)]) / (Ah[g.h2(i, j)] + Ah[g.h2(i + 1, j)]));
          else hu[g.u2(I, j + 1)] = 0.0;
        } else {
          if (Ah[g.h2(i, j + 1)] + Ah[g.h2(i + 1, j + 1)] > 0.0)
            hu[g.u2(I, j)] = hu[g.u2(I, j + 1)] * ((Ah[g.h2(i, j)] + Ah[g.h2(i + 1, j)]) / (Ah[g.h2(i, j + 1)] + Ah[g.h2(i + 1, j + 1)]));
          else hu[g.u2(I, j)] = 0.0;
        }
      }
    }
  } else {
    const int I = S.IsdB, i = I, J = t, j = t;
    const bool E = S.direction == MOM6HIP_OBC_DIRECTION_E;
    if (phase == 0) {
      if (j >= max(Jsq - 1, S.jsd) && j <= min(Jeq + 2, S.jed)) { if (E) Ah[g.h2(i + 1, j)] = Ah[g.h2(i, j)]; else Ah[g.h2(i, j)] = Ah[g.h2(i + 1, j)]; }
    } else if (phase == 1) {
      double *dvdx = a.dvdx + kQ, *dudy = a.dudy + kQ;
      if (a.zero_vorticity) { dvdx[g.q2(I, J)] = 0.; dudy[g.q2(I, J)] = 0.; }
      if (a.freeslip_vorticity) dvdx[g.q2(I, J)] = 0.;
      if (a.computed_vorticity) {
        if (E) dvdx[g.q2(I, J)] = 2.0 * (cor_seg_q(S, S.tangential_vel, I, J, k) - v[g.v2(i, J)]) * g.dyCv[g.v2(i, J)];
        else dvdx[g.q2(I, J)] = 2.0 * (v[g.v2(i + 1, J)] - cor_seg_q(S, S.tangential_vel, I, J, k)) * g.dyCv[g.v2(i + 1, J)];
      }
      if (a.specified_vorticity) {
        if (E) dvdx[g.q2(I, J)] = cor_seg_q(S, S.tangential_grad, I, J, k) * g.dyCv[g.v2(i, J)] * g.dxBu[g.q2(I, J)];
        else dvdx[g.q2(I, J)] = cor_seg_q(S, S.tangential_grad, I, J, k) * g.dyCv[g.v2(i + 1, J)] * g.dxBu[g.q2(I, J)];
      }
      if (j >= max(Jsq - 1, S.jsd) && j <= min(Jeq + 2, S.jed)) {
        const double hi = E ? h[g.h2(i, j)] : h[g.h2(i + 1, j)];
        a.hArea_u[kU + g.u2(I, j)] = 0.5 * (Ah[g.h2(i, j)] + Ah[g.h2(i + 1, j)]) * hi;
        if (a.uh_center) a.uh_center[kU + g.u2(I, j)] = (g.dy_Cu[g.u2(I, j)] * 1.0) * u[g.u2(I, j)] * hi;
      }
    } else {
      if (J >= max(Jsq - 1, S.JsdB) && J <= min(Jeq + 1, S.JedB)) {
        double *hv = a.hArea_v + kV;
        if (E) {
          if (Ah[g.h2(i, j)] + Ah[g.h2(i, j + 1)] > 0.0)
            hv[g.v2(i + 1, J)] = hv[g.v2(i, J)] * ((Ah[g.h2(i + 1, j)] + Ah[g.h2(i + 1, j + 1)]) / (Ah[g.h2(i, j)] + Ah[g.h2(i, j + 1)]));
          else hv[g.v2(i + 1, J)] = 0.0;
        } else {      // (:449 sets hArea_v(i,J) from h(i,j+1) first; both branches below overwrite it)
          if (Ah[g.h2(i + 1, j)] + Ah[g.h2(i + 1, j + 1)] > 0.0)
            hv[g.v2(i, J)] = hv[g.v2(i + 1, J)] * ((Ah[g.h2(i, j)] + Ah[g.h2(i, j + 1)]) / (Ah[g.h2(i + 1, j)] + Ah[g.h2(i + 1, j + 1)]));
          else hv[g.v2(i, J)] = 0.0;
        }
      }
    }
  }
}

}  // namespace

extern "C" int mom6hip_coradcalc(mom6hip_ctx_t *ctx, const mom6hip_coriolisadv_cs_t *cs, const double *u,
                                 const double *v, const double *h, const double *uh, const double *vh,
                                 double *CAu, double *CAv, int32_t memspace) {
  return mom6hip_coradcalc_obc(ctx, cs, nullptr, u, v, h, uh, vh, CAu, CAv, memspace);
}

extern "C" int mom6hip_coradcalc_obc(mom6hip_ctx_t *ctx, const mom6hip_coriolisadv_cs_t *cs, const mom6hip_obc_t *obc, const double *u,
                                     const double *v, const double *h, const double *uh, const double *vh,
                                     double *CAu, double *CAv, int32_t memspace) {
  M6_REQUIRE(ctx != nullptr, "MOM_CoriolisAdv: Module must be initialized before it is used.");
  M6_REQUIRE(cs && u && v && h && uh && vh && CAu && CAv, "CorAdCalc: null argument");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "CorAdCalc: bad memspace");
  M6_REQUIRE(cs->coriolis_scheme >= MOM6HIP_SADOURNY75_ENERGY && cs->coriolis_scheme <= MOM6HIP_AL_BLEND,
             "CoriolisAdv_init: Unrecognized setting #define CORIOLIS_SCHEME %d", cs->coriolis_scheme);
  M6_REQUIRE(cs->ke_scheme >= MOM6HIP_KE_ARAKAWA && cs->ke_scheme <= MOM6HIP_KE_GUDONOV, "CoriolisAdv_init: invalid KE_SCHEME");
  M6_REQUIRE(cs->pv_adv_scheme == 0 || cs->pv_adv_scheme == MOM6HIP_PV_ADV_CENTERED || cs->pv_adv_scheme == MOM6HIP_PV_ADV_UPWIND1,
             "CoriolisAdv_init: #DEFINE PV_ADV_SCHEME in input file is invalid.");
  const bool en_dis = cs->coriolis_en_dis && cs->coriolis_scheme == MOM6HIP_SADOURNY75_ENERGY;      // (read by that scheme only)
  const bool with_obc = obc && obc->number_of_segments > 0;
  const bool ext = en_dis || cs->coriolis_scheme == MOM6HIP_ROBUST_ENSTRO || cs->coriolis_scheme == MOM6HIP_ARAKAWA_LAMB81 ||
                   cs->coriolis_scheme == MOM6HIP_AL_BLEND || with_obc;
  m6::GridDev &g = ctx->g;
  M6_REQUIRE(g.mask2dT && g.areaT && g.IareaT && g.dxCu && g.IdxCu && g.areaCu && g.dyCv && g.IdyCv && g.areaCv &&
             g.mask2dBu && g.IareaBu && g.CoriolisBu, "CorAdCalc: a required grid metric is missing");
  M6_REQUIRE(cs->coriolis_scheme != MOM6HIP_ROBUST_ENSTRO || (g.IdyCu && g.IdxCv), "CorAdCalc: ROBUST_ENSTRO needs the metrics IdyCu, IdxCv");
  M6_REQUIRE(!en_dis || (g.dy_Cu && g.dx_Cv), "CorAdCalc: CORIOLIS_EN_DIS needs the metrics dy_Cu, dx_Cv");
  M6_REQUIRE(g.isc - g.isd >= 2 && g.ied - g.iec >= 2 && g.jsc - g.jsd >= 2 && g.jed - g.jec >= 2,
             "CorAdCalc: needs a halo of at least 2");
  hipStream_t s = ctx->stream;
  const size_t bH = (size_t)g.nh3() * 8, bU = (size_t)g.nu3() * 8, bV = (size_t)g.nv3() * 8;
  CorArgs a;
  a.g = g; a.u = u; a.v = v; a.h = h; a.uh = uh; a.vh = vh; a.CAu = CAu; a.CAv = CAv;
  if (memspace == MOM6HIP_MEM_HOST) {
    const size_t sz[7] = {bU, bV, bH, bU, bV, bU, bV};
    const void *src[7] = {u, v, h, uh, vh, CAu, CAv};
    for (int n = 0; n < 7; n++) {
      if (ctx->stage[n].reserve(sz[n])) return 1;
      M6_HIP(hipMemcpyAsync(ctx->stage[n].p, src[n], sz[n], hipMemcpyHostToDevice, s));
    }
    a.u = (double *)ctx->stage[0].p; a.v = (double *)ctx->stage[1].p; a.h = (double *)ctx->stage[2].p;
    a.uh = (double *)ctx->stage[3].p; a.vh = (double *)ctx->stage[4].p;
    a.CAu = (double *)ctx->stage[5].p; a.CAv = (double *)ctx->stage[6].p;
  }
  a.scheme = cs->coriolis_scheme; a.ke_scheme = cs->ke_scheme; a.no_slip = cs->no_slip; a.bound = cs->bound_coriolis;
  a.vol_neglect = g.H_subroundoff * (1e-4 * 1.0) * (1e-4 * 1.0);   // :241
  dim3 grid((g.nk + COR_KCH - 1) / COR_KCH, (g.iec - g.isc + 2 + TI - 1) / TI, (g.jec - g.jsc + 2 + TJ - 1) / TJ);
  a.en_dis = en_dis ? 1 : 0; a.upwind1 = cs->pv_adv_scheme == MOM6HIP_PV_ADV_UPWIND1 ? 1 : 0;
  a.eps_vel = 1.0e-10 * 1.0; a.h_tiny = g.Angstrom_H;      // :242-243
  a.wt_lin_blend = cs->wt_lin_blend;
  a.Fe_m2 = cs->F_eff_max_blend - 2.0;                     // :544-548
  a.rat_lin = 1.5 * a.Fe_m2 / fmax(cs->wt_lin_blend, 1.0e-16);
  if (cs->F_eff_max_blend <= 2.0) { a.Fe_m2 = -1.; a.rat_lin = -1.0; }
  a.o_dvdx = a.o_dudy = a.o_hArea_u = a.o_hArea_v = a.o_uh_center = a.o_vh_center = a.o_Area_q = nullptr;
  a.o_seg_u = a.o_seg_v = nullptr;
  m6::Stager st(ctx, memspace);      // (the segments' own arrays and the work arrays of the open-boundary path)
  if (with_obc) {
    M6_REQUIRE(obc->segment, "CorAdCalc: OBC%%segment is required");
    M6_REQUIRE(!obc->specified_vorticity || (g.dxBu && g.dyBu), "CorAdCalc: OBC_SPECIFIED_VORTICITY needs the metrics dxBu, dyBu");
    const int nseg = obc->number_of_segments;
    const size_t nH2 = (size_t)g.nih * g.njh, nU2 = (size_t)(g.nih + 1) * g.njh, nV2 = (size_t)g.nih * (g.njh + 1), nQ2 = (size_t)(g.nih + 1) * (g.njh + 1);
    CorObc o;
    o.g = g; o.u = a.u; o.v = a.v; o.h = a.h;
    o.Area_h = (double *)st.scratch(8 * nH2); o.Area_q = (double *)st.scratch(8 * nQ2);
    o.dvdx = (double *)st.scratch(8 * nQ2 * g.nk); o.dudy = (double *)st.scratch(8 * nQ2 * g.nk);
    o.hArea_u = (double *)st.scratch(bU); o.hArea_v = (double *)st.scratch(bV);
    o.uh_center = en_dis ? (double *)st.scratch(bU) : nullptr; o.vh_center = en_dis ? (double *)st.scratch(bV) : nullptr;
    o.zero_vorticity = obc->zero_vorticity; o.freeslip_vorticity = obc->freeslip_vorticity;
    o.computed_vorticity = obc->computed_vorticity; o.specified_vorticity = obc->specified_vorticity;
    M6_REQUIRE(!st.failed() && o.Area_h && o.Area_q && o.dvdx && o.dudy && o.hArea_u && o.hArea_v, "CorAdCalc: staging of the open boundaries failed");
    std::vector<CorSeg> segs(nseg);
    for (int n = 0; n < nseg; n++) {
      const mom6hip_obc_segment_t &S = obc->segment[n];
      CorSeg &d = segs[n];
      d.direction = S.direction; d.is_N_or_S = S.is_N_or_S && S.on_pe; d.is_E_or_W = S.is_E_or_W && S.on_pe; d.pad = 0;
      d.IsdB = S.IsdB; d.IedB = S.IedB; d.JsdB = S.JsdB; d.JedB = S.JedB; d.isd = S.isd; d.ied = S.ied; d.jsd = S.jsd; d.jed = S.jed;
      d.tangential_vel = nullptr; d.tangential_grad = nullptr;
      if (!S.on_pe) continue;
      const bool ew = S.is_E_or_W != 0, ns = S.is_N_or_S != 0;
      M6_REQUIRE(ew != ns, "CorAdCalc: OBC segment %d is neither E/W nor N/S", n + 1);
      M6_REQUIRE(S.IsdB >= g.isd - 1 && S.IedB <= g.ied && S.JsdB >= g.jsd - 1 && S.JedB <= g.jed && S.isd >= g.isd && S.ied <= g.ied &&
                 S.jsd >= g.jsd && S.jed <= g.jed && (ew ? (S.IsdB >= g.isd && S.IsdB < g.ied) : (S.JsdB >= g.jsd && S.JsdB < g.jed)),
                 "CorAdCalc: OBC segment %d lies outside the data domain", n + 1);
      const size_t cnt = (size_t)(S.IedB - S.IsdB + 1) * (S.JedB - S.JsdB + 1) * g.nk;
      if (obc->computed_vorticity) { M6_REQUIRE(S.tangential_vel, "CorAdCalc: OBC_COMPUTED_VORTICITY needs segment%%tangential_vel"); d.tangential_vel = st.in(S.tangential_vel, cnt * 8); }
      if (obc->specified_vorticity) { M6_REQUIRE(S.tangential_grad, "CorAdCalc: OBC_SPECIFIED_VORTICITY needs segment%%tangential_grad"); d.tangential_grad = st.in(S.tangential_grad, cnt * 8); }
    }
    M6_REQUIRE(!st.failed(), "CorAdCalc: staging of the open boundaries failed");
    // gradKE :1037-1050: the faces of every segment of the PE over its whole range (segnum holds them when it is given; the table is
    // rebuilt from the segments, which is what the reference loops over) -- once per OBC, kept with the context (m6::obc_table)
    const int32_t *d_su = (const int32_t *)m6::obc_table(ctx, m6::OBC_SITE_CORAD, m6::obc_fingerprint(ctx, obc), 4 * (nU2 + nV2), [&](void *host) -> int {
      int32_t *su = (int32_t *)host, *sv = su + nU2;
      for (int n = 0; n < nseg; n++) {
        const mom6hip_obc_segment_t &S = obc->segment[n];
        if (!S.on_pe) continue;
        if (S.is_N_or_S) for (int i = S.isd; i <= S.ied; i++) sv[g.v2(i, S.JsdB)] = 1;
        else for (int j = S.jsd; j <= S.jed; j++) su[g.u2(S.IsdB, j)] = 1;
      }
      return 0;
    });
    if (!d_su) return 1;
    const int32_t *d_sv = d_su + nU2;
    const int ni2 = g.iec - g.isc + 5, nj2 = g.jec - g.jsc + 5;      // (is-2 : ie+2, js-2 : je+2)
    hipLaunchKernelGGL(cor_obc_area_kernel, dim3((ni2 + 255) / 256, nj2), dim3(256), 0, s, o, 0);
    auto seg_launch = [&](int phase) {
      for (int n = 0; n < nseg; n++) {
        if (!obc->segment[n].on_pe) continue;
        const int len = (segs[n].is_N_or_S ? segs[n].IedB - segs[n].IsdB : segs[n].JedB - segs[n].JsdB) + 1;
        hipLaunchKernelGGL(cor_obc_segment_kernel, dim3((len + 63) / 64, phase == 0 ? 1 : g.nk), dim3(64), 0, s, o, segs[n], phase);
      }
    };
    seg_launch(0);
    hipLaunchKernelGGL(cor_obc_area_kernel, dim3((ni2 + 255) / 256, nj2), dim3(256), 0, s, o, 1);
    hipLaunchKernelGGL(cor_obc_fields_kernel, dim3((ni2 + 255) / 256, nj2, g.nk), dim3(256), 0, s, o);
    seg_launch(1);
    seg_launch(2);
    M6_HIP(hipGetLastError());
    a.o_dvdx = o.dvdx; a.o_dudy = o.dudy; a.o_hArea_u = o.hArea_u; a.o_hArea_v = o.hArea_v; a.o_uh_center = o.uh_center;
    a.o_vh_center = o.vh_center; a.o_Area_q = o.Area_q; a.o_seg_u = d_su; a.o_seg_v = d_sv;
  }
  a.bc_PFu = a.bc_PFv = a.bc_diffu = a.bc_diffv = nullptr; a.bc_u = a.bc_v = nullptr; a.bc_inviscid = 0;
  if (ctx->bc_fuse && memspace == MOM6HIP_MEM_DEVICE) {
    mom6hip_ctx::BcAccelFuse *f = ctx->bc_fuse;
    a.bc_PFu = f->au; a.bc_PFv = f->av; a.bc_diffu = f->diffu; a.bc_diffv = f->diffv; a.bc_u = f->u_bc; a.bc_v = f->v_bc;
    a.bc_inviscid = f->inviscid;
    f->done = true;
  }
  if (ext) hipLaunchKernelGGL(coradcalc_kernel<true>, grid, dim3(TI, TJ), 0, s, a);
  else hipLaunchKernelGGL(coradcalc_kernel<false>, grid, dim3(TI, TJ), 0, s, a);
  M6_HIP(hipGetLastError());
  if (memspace == MOM6HIP_MEM_HOST) {
    M6_HIP(hipMemcpyAsync(CAu, a.CAu, bU, hipMemcpyDeviceToHost, s));
    M6_HIP(hipMemcpyAsync(CAv, a.CAv, bV, hipMemcpyDeviceToHost, s));
    M6_HIP(hipStreamSynchronize(s));
  }
  return 0;
}
