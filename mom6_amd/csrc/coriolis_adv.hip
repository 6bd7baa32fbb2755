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

struct CorArgs {
  m6::GridDev g;
  const double *u, *v, *h, *uh, *vh;
  double *CAu, *CAv;
  int scheme, ke_scheme, no_slip, bound;
  double vol_neglect;
};

using m6::max2;
using m6::min2;
__device__ __forceinline__ double max4(double a, double b, double c, double d) { return max2(max2(max2(a, b), c), d); }
__device__ __forceinline__ double min4(double a, double b, double c, double d) { return min2(min2(min2(a, b), c), d); }

__global__ __launch_bounds__(TI * TJ) void coradcalc_kernel(CorArgs p) {
  const m6::GridDev &g = p.g;
  __shared__ double s_q[TJ + 2][TI + 2];      // q(I0-1 .. I0+64, J0-1 .. J0+8) (the +1 column/row: Arakawa-Hsu)
  __shared__ double s_av[TJ + 2][TI + 2];     // abs_vort, same points
  __shared__ double s_ke[TJ + 1][TI + 1];     // KE(i = I0 .. I0+64, j = J0 .. J0+8)
  // k is the fastest grid dimension: the blocks of one tile at k, k+8, ... land on the same XCD close in time, so
  // the 2-D metrics of the tile are served by that XCD's L2 instead of being fetched again for every layer
  const int k = blockIdx.x;
  const int I0 = g.isc - 1 + blockIdx.y * TI, J0 = g.jsc - 1 + blockIdx.z * TJ;
  const int tid = threadIdx.y * TI + threadIdx.x;
  // Index arithmetic in 32 bits from one base per point (the 2-D offsets fit easily): the 64-bit h2 / u2 / v2 / q2 of every
  // access were most of this kernel's instructions.  uk / vk / hk ...: the layer's planes (wave-uniform pointers).
  const long kH = (long)g.nih * g.njh * k, kU = (long)(g.nih + 1) * g.njh * k, kV = (long)g.nih * (g.njh + 1) * k;
  const double *__restrict__ hk = p.h + kH, *__restrict__ uk = p.u + kU, *__restrict__ vk = p.v + kV;
  const double *__restrict__ uhk = p.uh + kU, *__restrict__ vhk = p.vh + kV;
  const int nih = g.nih, sU = g.nih + 1;      // row strides: h- and v-point arrays; u- and q-point arrays
  const int Iqmax = g.iec + 1, Jqmax = g.jec + 1;      // q is defined on (Isq-1:Ieq+1, Jsq-1:Jeq+1)

  for (int t = tid; t < (TI + 2) * (TJ + 2); t += TI * TJ) {
    const int ty = t / (TI + 2), tx = t - ty * (TI + 2);
    // ---- q-point (I,J) = (I0-1+tx, J0-1+ty), :246-274, :314-324, :459-491 ----
    {
      const int I = I0 - 1 + tx, J = J0 - 1 + ty;
      double qv = 0.0, av = 0.0;
      if (I <= Iqmax && J <= Jqmax) {
        const int oh = (I - g.isd) + nih * (J - g.jsd);               // h2(i, j)
        const int ou = (I - g.isd + 1) + sU * (J - g.jsd);            // u2(I, j)
        const int ov = (I - g.isd) + nih * (J - g.jsd + 1);           // v2(i, J)
        const int oq = (I - g.isd + 1) + sU * (J - g.jsd + 1);        // q2(I, J)
        const double A00 = g.mask2dT[oh] * g.areaT[oh];
        const double A10 = g.mask2dT[oh + 1] * g.areaT[oh + 1];
        const double A01 = g.mask2dT[oh + nih] * g.areaT[oh + nih];
        const double A11 = g.mask2dT[oh + nih + 1] * g.areaT[oh + nih + 1];
        const double Area_q = (A00 + A11) + (A10 + A01);
        const double dvdx = (vk[ov + 1] * g.dyCv[ov + 1] - vk[ov] * g.dyCv[ov]);
        const double dudy = (uk[ou + sU] * g.dxCu[ou + sU] - uk[ou] * g.dxCu[ou]);
        const double h00 = hk[oh], h10 = hk[oh + 1], h01 = hk[oh + nih], h11 = hk[oh + nih + 1];
        const double hArea_u0 = 0.5 * (A00 * h00 + A10 * h10);     // hArea_u(I,j)
        const double hArea_u1 = 0.5 * (A01 * h01 + A11 * h11);     // hArea_u(I,j+1)
        const double hArea_v0 = 0.5 * (A00 * h00 + A01 * h01);     // hArea_v(i,J)
        const double hArea_v1 = 0.5 * (A10 * h10 + A11 * h11);     // hArea_v(i+1,J)
        const double mB = g.mask2dBu[oq];
        const double rel_vort = (p.no_slip ? (2.0 - mB) : mB) * (dvdx - dudy) * g.IareaBu[oq];
        av = g.CoriolisBu[oq] + rel_vort;
        const double hArea_q = (hArea_u0 + hArea_u1) + (hArea_v0 + hArea_v1);
        const double Ih_q = Area_q / (hArea_q + p.vol_neglect);
        qv = av * Ih_q;
      }
      s_q[ty][tx] = qv; s_av[ty][tx] = av;
    }
    // ---- h-point (i,j) = (I0+tx, J0+ty): KE, :995-1025 ----
    if (tx <= TI && ty <= TJ) {
      const int i = I0 + tx, j = J0 + ty;
      double ke = 0.0;
      if (i <= g.iec + 1 && j <= g.jec + 1) {
        const int oh = (i - g.isd) + nih * (j - g.jsd), ou = (i - g.isd + 1) + sU * (j - g.jsd), ov = (i - g.isd) + nih * (j - g.jsd + 1);
        const double uE = uk[ou], uW = uk[ou - 1], vN = vk[ov], vS = vk[ov - nih];
        if (p.ke_scheme == MOM6HIP_KE_ARAKAWA) {
          ke = ((g.areaCu[ou] * (uE * uE) + g.areaCu[ou - 1] * (uW * uW)) +
                (g.areaCv[ov] * (vN * vN) + g.areaCv[ov - nih] * (vS * vS))) * 0.25 * g.IareaT[oh];
        } else {
          const double up = 0.5 * (uW + fabs(uW)), um = 0.5 * (uE - fabs(uE));
          const double vp = 0.5 * (vS + fabs(vS)), vm = 0.5 * (vN - fabs(vN));
          if (p.ke_scheme == MOM6HIP_KE_SIMPLE_GUDONOV) {
            ke = (max2(up * up, um * um) + max2(vp * vp, vm * vm)) * 0.5;
          } else {
            const double up2a = up * up * g.areaCu[ou - 1], um2a = um * um * g.areaCu[ou];
            const double vp2a = vp * vp * g.areaCv[ov - nih], vm2a = vm * vm * g.areaCv[ov];
            ke = (max2(um2a, up2a) + max2(vm2a, vp2a)) * 0.5 * g.IareaT[oh];
          }
        }
      }
      s_ke[ty][tx] = ke;
    }
  }
  __syncthreads();

  const int tx = threadIdx.x, ty = threadIdx.y;
  const int I = I0 + tx, J = J0 + ty;           // this thread's u-point (I, j=J) and v-point (i=I, J)
  const int ou = (I - g.isd + 1) + sU * (J - g.jsd), ov = (I - g.isd) + nih * (J - g.jsd + 1);      // u2(I, J), v2(I, J)
  // tile coordinates: q(I,J) = s_q[ty+1][tx+1]; KE(i,j) = s_ke[ty][tx]
  const double C1_12 = 1.0 / 12.0;
  // ---- CAu(I, j), j = J >= jsc, I <= iec : :644-752 ----
  if (J >= g.jsc && J <= g.jec && I <= g.iec) {
    const double qN = s_q[ty + 1][tx + 1], qS = s_q[ty][tx + 1];           // q(I,J), q(I,J-1)
    const double vh_ne = vhk[ov + 1], vh_nw = vhk[ov], vh_sw = vhk[ov - nih], vh_se = vhk[ov - nih + 1];
    const double IdxCu = g.IdxCu[ou];
    double ca;
    if (p.scheme == MOM6HIP_SADOURNY75_ENERGY) {
      ca = 0.25 * (qN * (vh_ne + vh_nw) + qS * (vh_sw + vh_se)) * IdxCu;
    } else if (p.scheme == MOM6HIP_SADOURNY75_ENSTRO) {
      ca = 0.125 * (IdxCu * (qN + qS)) * ((vh_ne + vh_nw) + (vh_sw + vh_se));
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
    const double KEx = (s_ke[ty][tx + 1] - s_ke[ty][tx]) * IdxCu;
    p.CAu[kU + ou] = ca - KEx;
  }
  // ---- CAv(i, J), i = I >= isc, J <= jec : :763-876 ----
  if (I >= g.isc && I <= g.iec && J <= g.jec) {
    const double qE = s_q[ty + 1][tx + 1], qW = s_q[ty + 1][tx];           // q(I,J), q(I-1,J)
    const double uh_sw = uhk[ou - 1], uh_nw = uhk[ou - 1 + sU], uh_se = uhk[ou], uh_ne = uhk[ou + sU];
    const double IdyCv = g.IdyCv[ov];
    double ca;
    if (p.scheme == MOM6HIP_SADOURNY75_ENERGY) {
      ca = -0.25 * (qW * (uh_sw + uh_nw) + qE * (uh_se + uh_ne)) * IdyCv;
    } else if (p.scheme == MOM6HIP_SADOURNY75_ENSTRO) {
      ca = -0.125 * (IdyCv * (qW + qE)) * ((uh_sw + uh_nw) + (uh_se + uh_ne));
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
    const double KEy = (s_ke[ty + 1][tx] - s_ke[ty][tx]) * IdyCv;
    p.CAv[kV + ov] = ca - KEy;
  }
}

}  // namespace

extern "C" int mom6hip_coradcalc(mom6hip_ctx_t *ctx, const mom6hip_coriolisadv_cs_t *cs, const double *u,
                                 const double *v, const double *h, const double *uh, const double *vh,
                                 double *CAu, double *CAv, int32_t memspace) {
  M6_REQUIRE(ctx != nullptr, "MOM_CoriolisAdv: Module must be initialized before it is used.");
  M6_REQUIRE(cs && u && v && h && uh && vh && CAu && CAv, "CorAdCalc: null argument");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "CorAdCalc: bad memspace");
  M6_REQUIRE(cs->coriolis_scheme == MOM6HIP_SADOURNY75_ENERGY || cs->coriolis_scheme == MOM6HIP_SADOURNY75_ENSTRO ||
             cs->coriolis_scheme == MOM6HIP_ARAKAWA_HSU90,
             "CoriolisAdv_init: CORIOLIS_SCHEME %d is not provided (SADOURNY75_ENERGY, SADOURNY75_ENSTRO, ARAKAWA_HSU90)",
             cs->coriolis_scheme);
  M6_REQUIRE(cs->ke_scheme >= MOM6HIP_KE_ARAKAWA && cs->ke_scheme <= MOM6HIP_KE_GUDONOV, "CoriolisAdv_init: invalid KE_SCHEME");
  M6_REQUIRE(!cs->coriolis_en_dis, "CorAdCalc: CORIOLIS_EN_DIS is not provided");
  m6::GridDev &g = ctx->g;
  M6_REQUIRE(g.mask2dT && g.areaT && g.IareaT && g.dxCu && g.IdxCu && g.areaCu && g.dyCv && g.IdyCv && g.areaCv &&
             g.mask2dBu && g.IareaBu && g.CoriolisBu, "CorAdCalc: a required grid metric is missing");
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
  dim3 grid(g.nk, (g.iec - g.isc + 2 + TI - 1) / TI, (g.jec - g.jsc + 2 + TJ - 1) / TJ);
  hipLaunchKernelGGL(coradcalc_kernel, grid, dim3(TI, TJ), 0, s, a);
  M6_HIP(hipGetLastError());
  if (memspace == MOM6HIP_MEM_HOST) {
    M6_HIP(hipMemcpyAsync(CAu, a.CAu, bU, hipMemcpyDeviceToHost, s));
    M6_HIP(hipMemcpyAsync(CAv, a.CAv, bV, hipMemcpyDeviceToHost, s));
    M6_HIP(hipStreamSynchronize(s));
  }
  return 0;
}
