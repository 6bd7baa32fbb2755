// hor_visc.hip -- hor_visc_init (the static arrays) and horizontal_viscosity of
// src/parameterizations/lateral/MOM_hor_visc.F90 (:1984-2876, :245-1979) as gfx950 kernels.
//
// horizontal_viscosity is a chain of four 2-D stencils per layer (velocities -> strains -> Laplacian of the velocity ->
// stresses -> accelerations), every layer independent.  hv_fused_kernel does the chain for one tile of one layer with the
// intermediates (the reference's 2-D work arrays sh_xx, sh_xy, Del2u, Del2v, str_xx, str_xy, h_u, h_v) in LDS: see there.
// The products hor_visc_init keeps in the control structure (dx2h = dxT*dxT, DX_dyT = dxT*IdyT, Idx2dyCu ...) are single
// multiplications of grid metrics and are evaluated in place (the same bits as the stored arrays).
// Algorithmic traffic: read u, v, h, write diffu, diffv = 40 B per cell.
#include <cmath>

#include "common.hpp"

namespace {

using m6::max2;
using m6::min2;
__device__ __forceinline__ double max4(double a, double b, double c, double d) { return max2(max2(max2(a, b), c), d); }
__device__ __forceinline__ double min4(double a, double b, double c, double d) { return min2(min2(min2(a, b), c), d); }

struct HVOpt {
  double Kh, Kh_bg_min, Kh_vel_scale, Smag_Lap_const, Ah, Ah_vel_scale, Ah_time_scale, Smag_bi_const, bound_Cor_vel, bound_coef;
  int Laplacian, biharmonic, Smagorinsky_Kh, Smagorinsky_Ah, bound_Kh, better_bound_Kh, bound_Ah, better_bound_Ah, bound_Coriolis,
      add_LES_viscosity, no_slip, use_land_mask, use_cont_thick;
};

struct HVStatic {      // device pointers to the arrays of the control structure
  double *Kh_bg_xx, *Kh_Max_xx, *Ah_bg_xx, *Ah_Max_xx, *Laplac2_const_xx, *Biharm_const_xx, *Biharm_const2_xx, *reduction_xx;
  double *Kh_bg_xy, *Kh_Max_xy, *Ah_bg_xy, *Ah_Max_xy, *Laplac2_const_xy, *Biharm_const_xy, *Biharm_const2_xy, *reduction_xy;
};

#define H2(i, j) g.h2(i, j)
#define U2(i, j) g.u2(i, j)
#define V2(i, j) g.v2(i, j)
#define Q2(i, j) g.q2(i, j)
#define dx2q(I, J) (g.dxBu[Q2(I, J)] * g.dxBu[Q2(I, J)])
#define dy2q(I, J) (g.dyBu[Q2(I, J)] * g.dyBu[Q2(I, J)])
#define DX_dyBu(I, J) (g.dxBu[Q2(I, J)] * g.IdyBu[Q2(I, J)])
#define DY_dxBu(I, J) (g.dyBu[Q2(I, J)] * g.IdxBu[Q2(I, J)])
#define dx2h(i, j) (g.dxT[H2(i, j)] * g.dxT[H2(i, j)])
#define dy2h(i, j) (g.dyT[H2(i, j)] * g.dyT[H2(i, j)])
#define DX_dyT(i, j) (g.dxT[H2(i, j)] * g.IdyT[H2(i, j)])
#define DY_dxT(i, j) (g.dyT[H2(i, j)] * g.IdxT[H2(i, j)])
#define Idx2dyCu(I, j) ((g.IdxCu[U2(I, j)] * g.IdxCu[U2(I, j)]) * g.IdyCu[U2(I, j)])
#define Idxdy2u(I, j) (g.IdxCu[U2(I, j)] * (g.IdyCu[U2(I, j)] * g.IdyCu[U2(I, j)]))
#define Idx2dyCv(i, J) ((g.IdxCv[V2(i, J)] * g.IdxCv[V2(i, J)]) * g.IdyCv[V2(i, J)])
#define Idxdy2v(i, J) (g.IdxCv[V2(i, J)] * (g.IdyCv[V2(i, J)] * g.IdyCv[V2(i, J)]))

// ---- hor_visc_init ---------------------------------------------------------------------------------------------------
struct InitArgs {
  m6::GridDev g;
  HVOpt o;
  HVStatic s;
  double *u0u, *u0v, *v0u, *v0v;      // work arrays of the better_bound_Ah bound (:2695-2717)
  double dt;
};

// thread (i, j) over the data domain; each block of the reference's loops is guarded by its own index range
__global__ __launch_bounds__(256) void hv_init1_kernel(InitArgs A) {
  const m6::GridDev &g = A.g;
  const HVOpt &o = A.o;
  const int i = g.isd + blockIdx.x * blockDim.x + threadIdx.x, j = g.jsd + blockIdx.y;
  if (i > g.ied || j > g.jed) return;
  const int is = g.isc, ie = g.iec, js = g.jsc, je = g.jec;
  const int Isq = is - 1, Ieq = ie, Jsq = js - 1, Jeq = je;
  const int I = i, J = j;
  const double dt = A.dt;
  // reduction_xx :2473-2490
  if (j >= Jsq && j <= Jeq + 1 && i >= Isq && i <= Ieq + 1) {
    double r = 1.0;
    if ((g.dy_Cu[U2(I, j)] > 0.0) && (g.dy_Cu[U2(I, j)] < g.dyCu[U2(I, j)]) && (g.dy_Cu[U2(I, j)] < g.dyCu[U2(I, j)] * r))
      r = g.dy_Cu[U2(I, j)] / (g.dyCu[U2(I, j)]);
    if ((g.dy_Cu[U2(I - 1, j)] > 0.0) && (g.dy_Cu[U2(I - 1, j)] < g.dyCu[U2(I - 1, j)]) && (g.dy_Cu[U2(I - 1, j)] < g.dyCu[U2(I - 1, j)] * r))
      r = g.dy_Cu[U2(I - 1, j)] / (g.dyCu[U2(I - 1, j)]);
    if ((g.dx_Cv[V2(i, J)] > 0.0) && (g.dx_Cv[V2(i, J)] < g.dxCv[V2(i, J)]) && (g.dx_Cv[V2(i, J)] < g.dxCv[V2(i, J)] * r))
      r = g.dx_Cv[V2(i, J)] / (g.dxCv[V2(i, J)]);
    if ((g.dx_Cv[V2(i, J - 1)] > 0.0) && (g.dx_Cv[V2(i, J - 1)] < g.dxCv[V2(i, J - 1)]) && (g.dx_Cv[V2(i, J - 1)] < g.dxCv[V2(i, J - 1)] * r))
      r = g.dx_Cv[V2(i, J - 1)] / (g.dxCv[V2(i, J - 1)]);
    A.s.reduction_xx[H2(i, j)] = r;
  }
  // reduction_xy :2492-2509
  if (J >= js - 1 && J <= Jeq && I >= is - 1 && I <= Ieq) {
    double r = 1.0;
    if ((g.dy_Cu[U2(I, j)] > 0.0) && (g.dy_Cu[U2(I, j)] < g.dyCu[U2(I, j)]) && (g.dy_Cu[U2(I, j)] < g.dyCu[U2(I, j)] * r))
      r = g.dy_Cu[U2(I, j)] / (g.dyCu[U2(I, j)]);
    if ((g.dy_Cu[U2(I, j + 1)] > 0.0) && (g.dy_Cu[U2(I, j + 1)] < g.dyCu[U2(I, j + 1)]) && (g.dy_Cu[U2(I, j + 1)] < g.dyCu[U2(I, j + 1)] * r))
      r = g.dy_Cu[U2(I, j + 1)] / (g.dyCu[U2(I, j + 1)]);
    if ((g.dx_Cv[V2(i, J)] > 0.0) && (g.dx_Cv[V2(i, J)] < g.dxCv[V2(i, J)]) && (g.dx_Cv[V2(i, J)] < g.dxCv[V2(i, J)] * r))
      r = g.dx_Cv[V2(i, J)] / (g.dxCv[V2(i, J)]);
    if ((g.dx_Cv[V2(i + 1, J)] > 0.0) && (g.dx_Cv[V2(i + 1, J)] < g.dxCv[V2(i + 1, J)]) && (g.dx_Cv[V2(i + 1, J)] < g.dxCv[V2(i + 1, J)] * r))
      r = g.dx_Cv[V2(i + 1, J)] / (g.dxCv[V2(i + 1, J)]);
    A.s.reduction_xy[Q2(I, J)] = r;
  }
  const bool in_h = (j >= js - 1 && j <= Jeq + 1 && i >= is - 1 && i <= Ieq + 1);
  const bool in_q = (J >= js - 1 && J <= Jeq && I >= is - 1 && I <= Ieq);
  if (o.Laplacian) {      // :2511-2568
    double Kh_Limit = 0.0;
    if (o.bound_Kh || o.bound_Ah) Kh_Limit = 0.3 / (dt * 4.0);
    if (in_h) {
      const double grid_sp_h2 = (2.0 * dx2h(i, j) * dy2h(i, j)) / (dx2h(i, j) + dy2h(i, j));
      if (o.Smagorinsky_Kh) A.s.Laplac2_const_xx[H2(i, j)] = o.Smag_Lap_const * grid_sp_h2;
      double kb = max2(o.Kh, o.Kh_vel_scale * sqrt(grid_sp_h2));
      if (o.bound_Kh && !o.better_bound_Kh) {
        A.s.Kh_Max_xx[H2(i, j)] = Kh_Limit * grid_sp_h2;
        kb = min2(kb, Kh_Limit * grid_sp_h2);
      }
      A.s.Kh_bg_xx[H2(i, j)] = kb;
    }
    if (in_q) {
      const double grid_sp_q2 = (2.0 * dx2q(I, J) * dy2q(I, J)) / (dx2q(I, J) + dy2q(I, J));
      if (o.Smagorinsky_Kh) A.s.Laplac2_const_xy[Q2(I, J)] = o.Smag_Lap_const * grid_sp_q2;
      double kb = max2(o.Kh, o.Kh_vel_scale * sqrt(grid_sp_q2));
      if (o.bound_Kh && !o.better_bound_Kh) {
        A.s.Kh_Max_xy[Q2(I, J)] = Kh_Limit * grid_sp_q2;
        kb = min2(kb, Kh_Limit * grid_sp_q2);
      }
      A.s.Kh_bg_xy[Q2(I, J)] = kb;
    }
  }
  if (o.biharmonic) {      // :2570-2660
    double Ah_Limit = 0.0, BoundCorConst = 0.0;
    if (o.better_bound_Ah || o.bound_Ah) Ah_Limit = 0.3 / (dt * 64.0);
    if (o.Smagorinsky_Ah && o.bound_Coriolis) BoundCorConst = 1.0 / (5.0 * (o.bound_Cor_vel * o.bound_Cor_vel));
    if (in_h) {
      const double grid_sp_h2 = (2.0 * dx2h(i, j) * dy2h(i, j)) / (dx2h(i, j) + dy2h(i, j));
      if (o.Smagorinsky_Ah) {
        A.s.Biharm_const_xx[H2(i, j)] = o.Smag_bi_const * (grid_sp_h2 * grid_sp_h2);
        if (o.bound_Coriolis) {
          const double fmax = max4(fabs(g.CoriolisBu[Q2(I - 1, J - 1)]), fabs(g.CoriolisBu[Q2(I, J - 1)]),
                                   fabs(g.CoriolisBu[Q2(I - 1, J)]), fabs(g.CoriolisBu[Q2(I, J)]));
          A.s.Biharm_const2_xx[H2(i, j)] = (grid_sp_h2 * grid_sp_h2 * grid_sp_h2) * (fmax * BoundCorConst);
        }
      }
      double ab = max2(o.Ah, o.Ah_vel_scale * grid_sp_h2 * sqrt(grid_sp_h2));
      if (o.Ah_time_scale > 0.) ab = max2(ab, (grid_sp_h2 * grid_sp_h2) / o.Ah_time_scale);
      if (o.bound_Ah && !o.better_bound_Ah) {
        A.s.Ah_Max_xx[H2(i, j)] = Ah_Limit * (grid_sp_h2 * grid_sp_h2);
        ab = min2(ab, Ah_Limit * (grid_sp_h2 * grid_sp_h2));
      }
      A.s.Ah_bg_xx[H2(i, j)] = ab;
    }
    if (in_q) {
      const double grid_sp_q2 = (2.0 * dx2q(I, J) * dy2q(I, J)) / (dx2q(I, J) + dy2q(I, J));
      if (o.Smagorinsky_Ah) {
        A.s.Biharm_const_xy[Q2(I, J)] = o.Smag_bi_const * (grid_sp_q2 * grid_sp_q2);
        if (o.bound_Coriolis)
          A.s.Biharm_const2_xy[Q2(I, J)] = (grid_sp_q2 * grid_sp_q2 * grid_sp_q2) * (fabs(g.CoriolisBu[Q2(I, J)]) * BoundCorConst);
      }
      double ab = max2(o.Ah, o.Ah_vel_scale * grid_sp_q2 * sqrt(grid_sp_q2));
      if (o.Ah_time_scale > 0.) ab = max2(ab, (grid_sp_q2 * grid_sp_q2) / o.Ah_time_scale);
      if (o.bound_Ah && !o.better_bound_Ah) {
        A.s.Ah_Max_xy[Q2(I, J)] = Ah_Limit * (grid_sp_q2 * grid_sp_q2);
        ab = min2(ab, Ah_Limit * (grid_sp_q2 * grid_sp_q2));
      }
      A.s.Ah_bg_xy[Q2(I, J)] = ab;
    }
  }
  const double Idt = 1.0 / dt;
  if (o.Laplacian && o.better_bound_Kh) {      // :2664-2693
    if (in_h) {
      const double denom = max2(
          (dy2h(i, j) * DY_dxT(i, j) * (g.IdyCu[U2(I, j)] + g.IdyCu[U2(I - 1, j)]) *
           max2(g.IdyCu[U2(I, j)] * g.IareaCu[U2(I, j)], g.IdyCu[U2(I - 1, j)] * g.IareaCu[U2(I - 1, j)])),
          (dx2h(i, j) * DX_dyT(i, j) * (g.IdxCv[V2(i, J)] + g.IdxCv[V2(i, J - 1)]) *
           max2(g.IdxCv[V2(i, J)] * g.IareaCv[V2(i, J)], g.IdxCv[V2(i, J - 1)] * g.IareaCv[V2(i, J - 1)])));
      double km = 0.0;
      if (denom > 0.0) km = o.bound_coef * 0.25 * Idt / denom;
      A.s.Kh_Max_xx[H2(i, j)] = km;
    }
    if (in_q) {
      const double denom = max2(
          (dx2q(I, J) * DX_dyBu(I, J) * (g.IdxCu[U2(I, j + 1)] + g.IdxCu[U2(I, j)]) *
           max2(g.IdxCu[U2(I, j)] * g.IareaCu[U2(I, j)], g.IdxCu[U2(I, j + 1)] * g.IareaCu[U2(I, j + 1)])),
          (dy2q(I, J) * DY_dxBu(I, J) * (g.IdyCv[V2(i + 1, J)] + g.IdyCv[V2(i, J)]) *
           max2(g.IdyCv[V2(i, J)] * g.IareaCv[V2(i, J)], g.IdyCv[V2(i + 1, J)] * g.IareaCv[V2(i + 1, J)])));
      double km = 0.0;
      if (denom > 0.0) km = o.bound_coef * 0.25 * Idt / denom;
      A.s.Kh_Max_xy[Q2(I, J)] = km;
    }
  }
  if (o.biharmonic && o.better_bound_Ah) {      // the responses u0u, u0v, v0u, v0v :2695-2717
    if (j >= js - 1 && j <= Jeq + 1 && I >= is - 2 && I <= Ieq + 1) {
      A.u0u[U2(I, j)] = (Idxdy2u(I, j) * (dy2h(i + 1, j) * DY_dxT(i + 1, j) * (g.IdyCu[U2(I + 1, j)] + g.IdyCu[U2(I, j)]) +
                                         dy2h(i, j) * DY_dxT(i, j) * (g.IdyCu[U2(I, j)] + g.IdyCu[U2(I - 1, j)])) +
                         Idx2dyCu(I, j) * (dx2q(I, J) * DX_dyBu(I, J) * (g.IdxCu[U2(I, j + 1)] + g.IdxCu[U2(I, j)]) +
                                          dx2q(I, J - 1) * DX_dyBu(I, J - 1) * (g.IdxCu[U2(I, j)] + g.IdxCu[U2(I, j - 1)])));
      A.u0v[U2(I, j)] = (Idxdy2u(I, j) * (dy2h(i + 1, j) * DX_dyT(i + 1, j) * (g.IdxCv[V2(i + 1, J)] + g.IdxCv[V2(i + 1, J - 1)]) +
                                         dy2h(i, j) * DX_dyT(i, j) * (g.IdxCv[V2(i, J)] + g.IdxCv[V2(i, J - 1)])) +
                         Idx2dyCu(I, j) * (dx2q(I, J) * DY_dxBu(I, J) * (g.IdyCv[V2(i + 1, J)] + g.IdyCv[V2(i, J)]) +
                                          dx2q(I, J - 1) * DY_dxBu(I, J - 1) * (g.IdyCv[V2(i + 1, J - 1)] + g.IdyCv[V2(i, J - 1)])));
    }
    if (J >= js - 2 && J <= Jeq + 1 && i >= is - 1 && i <= Ieq + 1) {
      A.v0u[V2(i, J)] = (Idxdy2v(i, J) * (dy2q(I, J) * DX_dyBu(I, J) * (g.IdxCu[U2(I, j + 1)] + g.IdxCu[U2(I, j)]) +
                                         dy2q(I - 1, J) * DX_dyBu(I - 1, J) * (g.IdxCu[U2(I - 1, j + 1)] + g.IdxCu[U2(I - 1, j)])) +
                         Idx2dyCv(i, J) * (dx2h(i, j + 1) * DY_dxT(i, j + 1) * (g.IdyCu[U2(I, j + 1)] + g.IdyCu[U2(I - 1, j + 1)]) +
                                          dx2h(i, j) * DY_dxT(i, j) * (g.IdyCu[U2(I, j)] + g.IdyCu[U2(I - 1, j)])));
      A.v0v[V2(i, J)] = (Idxdy2v(i, J) * (dy2q(I, J) * DY_dxBu(I, J) * (g.IdyCv[V2(i + 1, J)] + g.IdyCv[V2(i, J)]) +
                                         dy2q(I - 1, J) * DY_dxBu(I - 1, J) * (g.IdyCv[V2(i, J)] + g.IdyCv[V2(i - 1, J)])) +
                         Idx2dyCv(i, J) * (dx2h(i, j + 1) * DX_dyT(i, j + 1) * (g.IdxCv[V2(i, J + 1)] + g.IdxCv[V2(i, J)]) +
                                          dx2h(i, j) * DX_dyT(i, j) * (g.IdxCv[V2(i, J)] + g.IdxCv[V2(i, J - 1)])));
    }
  }
}

// Ah_Max_xx / Ah_Max_xy from the responses :2719-2755
__global__ __launch_bounds__(256) void hv_init2_kernel(InitArgs A) {
  const m6::GridDev &g = A.g;
  const HVOpt &o = A.o;
  const int i = g.isd + blockIdx.x * blockDim.x + threadIdx.x, j = g.jsd + blockIdx.y;
  if (i > g.ied || j > g.jed) return;
  const int is = g.isc, ie = g.iec, js = g.jsc, je = g.jec;
  const int Ieq = ie, Jeq = je;
  const int I = i, J = j;
  const double Idt = 1.0 / A.dt;
  const double *u0u = A.u0u, *u0v = A.u0v, *v0u = A.v0u, *v0v = A.v0v;
  if (j >= js - 1 && j <= Jeq + 1 && i >= is - 1 && i <= Ieq + 1) {
    const double denom = max2(
        (dy2h(i, j) *
         (DY_dxT(i, j) * (g.IdyCu[U2(I, j)] * u0u[U2(I, j)] + g.IdyCu[U2(I - 1, j)] * u0u[U2(I - 1, j)]) +
          DX_dyT(i, j) * (g.IdxCv[V2(i, J)] * v0u[V2(i, J)] + g.IdxCv[V2(i, J - 1)] * v0u[V2(i, J - 1)])) *
         max2(g.IdyCu[U2(I, j)] * g.IareaCu[U2(I, j)], g.IdyCu[U2(I - 1, j)] * g.IareaCu[U2(I - 1, j)])),
        (dx2h(i, j) *
         (DY_dxT(i, j) * (g.IdyCu[U2(I, j)] * u0v[U2(I, j)] + g.IdyCu[U2(I - 1, j)] * u0v[U2(I - 1, j)]) +
          DX_dyT(i, j) * (g.IdxCv[V2(i, J)] * v0v[V2(i, J)] + g.IdxCv[V2(i, J - 1)] * v0v[V2(i, J - 1)])) *
         max2(g.IdxCv[V2(i, J)] * g.IareaCv[V2(i, J)], g.IdxCv[V2(i, J - 1)] * g.IareaCv[V2(i, J - 1)])));
    double am = 0.0;
    if (denom > 0.0) am = o.bound_coef * 0.5 * Idt / denom;
    A.s.Ah_Max_xx[H2(i, j)] = am;
  }
  if (J >= js - 1 && J <= Jeq && I >= is - 1 && I <= Ieq) {
    const double denom = max2(
        (dx2q(I, J) *
         (DX_dyBu(I, J) * (u0u[U2(I, j + 1)] * g.IdxCu[U2(I, j + 1)] + u0u[U2(I, j)] * g.IdxCu[U2(I, j)]) +
          DY_dxBu(I, J) * (v0u[V2(i + 1, J)] * g.IdyCv[V2(i + 1, J)] + v0u[V2(i, J)] * g.IdyCv[V2(i, J)])) *
         max2(g.IdxCu[U2(I, j)] * g.IareaCu[U2(I, j)], g.IdxCu[U2(I, j + 1)] * g.IareaCu[U2(I, j + 1)])),
        (dy2q(I, J) *
         (DX_dyBu(I, J) * (u0v[U2(I, j + 1)] * g.IdxCu[U2(I, j + 1)] + u0v[U2(I, j)] * g.IdxCu[U2(I, j)]) +
          DY_dxBu(I, J) * (v0v[V2(i + 1, J)] * g.IdyCv[V2(i + 1, J)] + v0v[V2(i, J)] * g.IdyCv[V2(i, J)])) *
         max2(g.IdyCv[V2(i, J)] * g.IareaCv[V2(i, J)], g.IdyCv[V2(i + 1, J)] * g.IareaCv[V2(i + 1, J)])));
    double am = 0.0;
    if (denom > 0.0) am = o.bound_coef * 0.5 * Idt / denom;
    A.s.Ah_Max_xy[Q2(I, J)] = am;
  }
}

// ---- horizontal_viscosity ----------------------------------------------------------------------------------------------
struct HVArgs {
  m6::GridDev g;
  HVOpt o;
  HVStatic s;
  const double *u, *v, *h, *hu_cont, *hv_cont;
  double *diffu, *diffv;
};

// thicknesses at velocity points :740-765 (land mask or not; hu_cont / hv_cont inside their ranges with USE_CONT_THICKNESS)
__device__ __forceinline__ double hu_at(const HVArgs &A, const double *hk, const double *huk, int I, int j) {
  const m6::GridDev &g = A.g;
  if (huk && I >= g.isc - 2) return huk[U2(I, j)];
  const int i = I;
  if (A.o.use_land_mask) return 0.5 * (g.mask2dT[H2(i, j)] * hk[H2(i, j)] + g.mask2dT[H2(i + 1, j)] * hk[H2(i + 1, j)]);
  return 0.5 * (hk[H2(i, j)] + hk[H2(i + 1, j)]);
}
__device__ __forceinline__ double hv_at(const HVArgs &A, const double *hk, const double *hvk, int i, int J) {
  const m6::GridDev &g = A.g;
  if (hvk && J >= g.jsc - 2) return hvk[V2(i, J)];
  const int j = J;
  if (A.o.use_land_mask) return 0.5 * (g.mask2dT[H2(i, j)] * hk[H2(i, j)] + g.mask2dT[H2(i, j + 1)] * hk[H2(i, j + 1)]);
  return 0.5 * (hk[H2(i, j)] + hk[H2(i, j + 1)]);
}

struct Planes {      // the layer's planes of every array (wave-uniform pointers)
  const double *u, *v, *h, *huc, *hvc;
  double *diffu, *diffv;
};
__device__ __forceinline__ Planes planes_of(const HVArgs &A, int k) {
  const m6::GridDev &g = A.g;
  const long kH = (long)g.nih * g.njh * k, kU = (long)(g.nih + 1) * g.njh * k, kV = (long)g.nih * (g.njh + 1) * k;
  Planes P;
  P.u = A.u + kU; P.v = A.v + kV; P.h = A.h + kH;
  const bool cont = A.o.use_cont_thick && A.hu_cont && A.hv_cont;
  P.huc = cont ? A.hu_cont + kU : nullptr; P.hvc = cont ? A.hv_cont + kV : nullptr;
  P.diffu = A.diffu + kU; P.diffv = A.diffv + kV;
  return P;
}

// ---- the fused kernel ------------------------------------------------------------------------------------------------
// One block = one tile of HV_TI x HV_TJ points of one layer.  The tile is treated as the reference treats a PE's compute
// domain (is_t : ie_t, js_t : je_t): every intermediate is formed over the reference's index ranges relative to the tile
// (sh_xx over is_t-2 : ie_t+2, Del2u over is_t-2 : ie_t+1, str_xx over is_t-1 : ie_t+1 ...), held in LDS in one local frame
// (li, lj) <-> (i0 - 2 + li, j0 - 2 + lj), and only the tile's own diffu / diffv go back to HBM -- every value is the
// result of the same operations on the same operands as in the layer-wide form.  The stresses overwrite the strains in
// LDS (the strains are dead by then).  512 threads sweep the 68 x 20 points of the frame in 3 passes per stage.
//
// Block order: the 30 two-dimensional metrics and coefficients a point needs are 6 times the 3-D traffic if they come from
// HBM for every layer.  blockIdx.x is decoded so that the blocks an XCD receives back to back (ids = xcd mod 8) are the nk
// layers of ONE tile: its metrics are read into that XCD's L2 once and hit there for the other layers.
constexpr int HV_TI = 64, HV_TJ = 16, HV_W = HV_TI + 4, HV_H = HV_TJ + 4, HV_NP = HV_W * HV_H, HV_NT = 512;
constexpr int HV_NIT = (HV_NP + HV_NT - 1) / HV_NT;

__global__ __launch_bounds__(HV_NT, 4) void hv_fused_kernel(HVArgs A, int ntx, int ntiles) {
  __shared__ double s_xx[HV_NP], s_xy[HV_NP], s_d2u[HV_NP], s_d2v[HV_NP], s_hu[HV_NP], s_hv[HV_NP];
  const m6::GridDev &g = A.g;
  const HVOpt &o = A.o;
  const int L = blockIdx.x, m = L >> 3;
  const int k = m % g.nk, tile = (m / g.nk) * 8 + (L & 7);
  if (tile >= ntiles) return;
  const int i0 = g.isc + (tile % ntx) * HV_TI, j0 = g.jsc + (tile / ntx) * HV_TJ;
  const int ib = i0 - 2, jb = j0 - 2;
  // the tile as a compute domain
  const int is = i0, ie = min(i0 + HV_TI - 1, g.iec), js = j0, je = min(j0 + HV_TJ - 1, g.jec);
  const int Isq = is - 1, Ieq = ie, Jsq = js - 1, Jeq = je;
  const Planes P = planes_of(A, k);
  const double h_neglect = g.H_subroundoff;
#define LX(a, i, j) a[((j) - jb) * HV_W + ((i) - ib)]
#define FOR_POINTS                                                   \
  _Pragma("unroll") for (int it = 0; it < HV_NIT; it++) {            \
    const int p = (int)threadIdx.x + it * HV_NT;                     \
    const int lj = p / HV_W, li = p - lj * HV_W;                     \
    const int i = ib + li, j = jb + lj, I = i, J = j;                \
    if (p < HV_NP)
#define END_POINTS }

  // ---- strains and the thicknesses at velocity points ----
  FOR_POINTS {
    if (j >= Jsq - 1 && j <= Jeq + 2 && i >= Isq - 1 && i <= Ieq + 2) {      // horizontal tension :693-699
      const double dudx = DY_dxT(i, j) * (g.IdyCu[U2(I, j)] * P.u[U2(I, j)] - g.IdyCu[U2(I - 1, j)] * P.u[U2(I - 1, j)]);
      const double dvdy = DX_dyT(i, j) * (g.IdxCv[V2(i, J)] * P.v[V2(i, J)] - g.IdxCv[V2(i, J - 1)] * P.v[V2(i, J - 1)]);
      LX(s_xx, i, j) = dudx - dvdy;
    }
    if (J >= js - 2 && J <= Jeq + 1 && I >= is - 2 && I <= Ieq + 1) {      // shearing strain :702-705, :852-864
      const double dvdx = DY_dxBu(I, J) * (P.v[V2(i + 1, J)] * g.IdyCv[V2(i + 1, J)] - P.v[V2(i, J)] * g.IdyCv[V2(i, J)]);
      const double dudy = DX_dyBu(I, J) * (P.u[U2(I, j + 1)] * g.IdxCu[U2(I, j + 1)] - P.u[U2(I, j)] * g.IdxCu[U2(I, j)]);
      if (o.no_slip) LX(s_xy, I, J) = (2.0 - g.mask2dBu[Q2(I, J)]) * (dvdx + dudy);
      else LX(s_xy, I, J) = g.mask2dBu[Q2(I, J)] * (dvdx + dudy);
    }
    if (j >= js - 1 && j <= je + 1 && I >= is - 2 && I <= ie + 1) LX(s_hu, I, j) = hu_at(A, P.h, P.huc, I, j);      // :740-765
    if (J >= js - 2 && J <= je + 1 && i >= is - 1 && i <= ie + 1) LX(s_hv, i, J) = hv_at(A, P.h, P.hvc, i, J);
  } END_POINTS
  __syncthreads();

  // ---- the Laplacian of the velocity :882-891 ----
  if (o.biharmonic) {
    FOR_POINTS {
      if (j >= js - 1 && j <= Jeq + 1 && I >= Isq - 1 && I <= Ieq + 1)
        LX(s_d2u, I, j) = Idxdy2u(I, j) * (dy2h(i + 1, j) * LX(s_xx, i + 1, j) - dy2h(i, j) * LX(s_xx, i, j)) +
                          Idx2dyCu(I, j) * (dx2q(I, J) * LX(s_xy, I, J) - dx2q(I, J - 1) * LX(s_xy, I, J - 1));
      if (J >= Jsq - 1 && J <= Jeq + 1 && i >= is - 1 && i <= Ieq + 1)
        LX(s_d2v, i, J) = Idxdy2v(i, J) * (dy2q(I, J) * LX(s_xy, I, J) - dy2q(I - 1, J) * LX(s_xy, I - 1, J)) -
                          Idx2dyCv(i, J) * (dx2h(i, j + 1) * LX(s_xx, i, j + 1) - dx2h(i, j) * LX(s_xx, i, j));
    } END_POINTS
    __syncthreads();
  }

  // ---- viscosities and layer-integrated stresses :1056-1741 (into registers: they replace the strains in LDS) ----
  const double h_neglect3 = h_neglect * h_neglect * h_neglect;
  const bool legacy_bound = o.Smagorinsky_Kh && (o.bound_Kh && !o.better_bound_Kh);
  const bool smag = o.Smagorinsky_Kh || o.Smagorinsky_Ah, bb = o.better_bound_Ah || o.better_bound_Kh;
  double r_xx[HV_NIT], r_xy[HV_NIT];
  FOR_POINTS {
    r_xx[it] = 0.0; r_xy[it] = 0.0;
    if (j >= Jsq && j <= Jeq + 1 && i >= Isq && i <= Ieq + 1) {      // ---- h point (is_Kh : ie_Kh, js_Kh : je_Kh) ----
      const double sxx = LX(s_xx, i, j);
      double Shear_mag = 0.0, hrat_min = 0.0, visc_bound_rem = 0.0;
      if (smag) {      // :1056-1063
        const double sh_xx_sq = sxx * sxx;
        const double sh_xy_sq = 0.25 * ((LX(s_xy, I - 1, J - 1) * LX(s_xy, I - 1, J - 1) + LX(s_xy, I, J) * LX(s_xy, I, J)) +
                                        (LX(s_xy, I - 1, J) * LX(s_xy, I - 1, J) + LX(s_xy, I, J - 1) * LX(s_xy, I, J - 1)));
        Shear_mag = sqrt(sh_xx_sq + sh_xy_sq);
      }
      if (bb) {      // :1065-1076
        const double h_min = min4(LX(s_hu, I, j), LX(s_hu, I - 1, j), LX(s_hv, i, J), LX(s_hv, i, J - 1));
        hrat_min = min2(1.0, h_min / (P.h[H2(i, j)] + h_neglect));
        if (o.better_bound_Kh) visc_bound_rem = 1.0;
      }
      double str = 0.0;
      if (o.Laplacian) {      // :1078-1214
        double K_ = A.s.Kh_bg_xx[H2(i, j)];
        if (o.add_LES_viscosity) {
          if (o.Smagorinsky_Kh) K_ = K_ + A.s.Laplac2_const_xx[H2(i, j)] * Shear_mag;
        } else {
          if (o.Smagorinsky_Kh) K_ = max2(K_, A.s.Laplac2_const_xx[H2(i, j)] * Shear_mag);
        }
        if (legacy_bound) K_ = min2(K_, A.s.Kh_Max_xx[H2(i, j)]);
        K_ = max2(K_, o.Kh_bg_min);
        if (o.better_bound_Kh) {
          if (K_ >= hrat_min * A.s.Kh_Max_xx[H2(i, j)]) {
            visc_bound_rem = 0.0;
            K_ = hrat_min * A.s.Kh_Max_xx[H2(i, j)];
          } else {
            visc_bound_rem = 1.0 - K_ / (hrat_min * A.s.Kh_Max_xx[H2(i, j)]);
          }
        }
        str = -K_ * sxx;
      }
      if (o.biharmonic) {      // :1227-1380
        double A_ = A.s.Ah_bg_xx[H2(i, j)];
        if (o.Smagorinsky_Ah) {
          double AhSm;
          if (o.bound_Coriolis) AhSm = Shear_mag * (A.s.Biharm_const_xx[H2(i, j)] + A.s.Biharm_const2_xx[H2(i, j)] * Shear_mag);
          else AhSm = A.s.Biharm_const_xx[H2(i, j)] * Shear_mag;
          A_ = max2(A_, AhSm);
          if (o.bound_Ah && !o.better_bound_Ah) A_ = min2(A_, A.s.Ah_Max_xx[H2(i, j)]);
        }
        if (o.better_bound_Ah) {
          if (o.better_bound_Kh) A_ = min2(A_, visc_bound_rem * hrat_min * A.s.Ah_Max_xx[H2(i, j)]);
          else A_ = min2(A_, hrat_min * A.s.Ah_Max_xx[H2(i, j)]);
        }
        const double d_del2u = g.IdyCu[U2(I, j)] * LX(s_d2u, I, j) - g.IdyCu[U2(I - 1, j)] * LX(s_d2u, I - 1, j);
        const double d_del2v = g.IdxCv[V2(i, J)] * LX(s_d2v, i, J) - g.IdxCv[V2(i, J - 1)] * LX(s_d2v, i, J - 1);
        const double d_str = A_ * (DY_dxT(i, j) * d_del2u - DX_dyT(i, j) * d_del2v);
        str = str + d_str;
      }
      r_xx[it] = str * (P.h[H2(i, j)] * A.s.reduction_xx[H2(i, j)]);      // :1728
    }

    if (J >= js - 1 && J <= Jeq && I >= is - 1 && I <= Ieq) {      // ---- q point ----
      const double sxy = LX(s_xy, I, J);
      double Shear_mag = 0.0;
      if (smag) {      // :1414-1421
        const double sh_xy_sq = sxy * sxy;
        const double sh_xx_sq = 0.25 * ((LX(s_xx, i, j) * LX(s_xx, i, j) + LX(s_xx, i + 1, j + 1) * LX(s_xx, i + 1, j + 1)) +
                                        (LX(s_xx, i, j + 1) * LX(s_xx, i, j + 1) + LX(s_xx, i + 1, j) * LX(s_xx, i + 1, j)));
        Shear_mag = sqrt(sh_xy_sq + sh_xx_sq);
      }
      const double hu0 = LX(s_hu, I, j), hu1 = LX(s_hu, I, j + 1);
      const double hv0 = LX(s_hv, i, J), hv1 = LX(s_hv, i + 1, J);
      const double h2uq = 4.0 * (hu0 * hu1);      // :1423-1428
      const double h2vq = 4.0 * (hv0 * hv1);
      double hq = (2.0 * (h2uq * h2vq)) / (h_neglect3 + (h2uq + h2vq) * ((hu0 + hu1) + (hv0 + hv1)));
      double hrat_min = 0.0, visc_bound_rem = 0.0;
      if (bb) {      // :1430-1441
        const double h_min = min4(hu0, hu1, hv0, hv1);
        hrat_min = min2(1.0, h_min / (hq + h_neglect));
        if (o.better_bound_Kh) visc_bound_rem = 1.0;
      }
      if (o.no_slip && (g.mask2dBu[Q2(I, J)] < 0.5)) {      // coastal vorticity points :1443-1466
        if ((g.mask2dCu[U2(I, j)] + g.mask2dCu[U2(I, j + 1)]) + (g.mask2dCv[V2(i, J)] + g.mask2dCv[V2(i + 1, J)]) > 0.0) {
          const double hu = g.mask2dCu[U2(I, j)] * hu0 + g.mask2dCu[U2(I, j + 1)] * hu1;
          const double hv = g.mask2dCv[V2(i, J)] * hv0 + g.mask2dCv[V2(i + 1, J)] * hv1;
          if ((g.mask2dCu[U2(I, j)] + g.mask2dCu[U2(I, j + 1)]) * (g.mask2dCv[V2(i, J)] + g.mask2dCv[V2(i + 1, J)]) == 0.0) {
            hq = hu + hv;
            hrat_min = 1.0;
          } else {
            hq = 2.0 * (hu * hv) / ((hu + hv) + h_neglect);
            hrat_min = min2(1.0, min2(hu, hv) / (hq + h_neglect));
          }
        }
      }
      double str = 0.0;
      if (o.Laplacian) {      // :1473-1585
        double K_ = A.s.Kh_bg_xy[Q2(I, J)];
        if (o.Smagorinsky_Kh) {
          if (o.add_LES_viscosity) K_ = K_ + A.s.Laplac2_const_xy[Q2(I, J)] * Shear_mag;
          else K_ = max2(K_, A.s.Laplac2_const_xy[Q2(I, J)] * Shear_mag);
        }
        if (legacy_bound) K_ = min2(K_, A.s.Kh_Max_xy[Q2(I, J)]);
        K_ = max2(K_, o.Kh_bg_min);
        if (o.better_bound_Kh) {
          if (K_ >= hrat_min * A.s.Kh_Max_xy[Q2(I, J)]) {
            visc_bound_rem = 0.0;
            K_ = hrat_min * A.s.Kh_Max_xy[Q2(I, J)];
          } else if (hrat_min * A.s.Kh_Max_xy[Q2(I, J)] > 0.) {
            visc_bound_rem = 1.0 - K_ / (hrat_min * A.s.Kh_Max_xy[Q2(I, J)]);
          }
        }
        str = -K_ * sxy;
      }
      if (o.biharmonic) {      // :1598-1693, with the gradient of the Laplacian :1382-1387
        double A_ = A.s.Ah_bg_xy[Q2(I, J)];
        if (o.Smagorinsky_Ah) {
          double AhSm;
          if (o.bound_Coriolis) AhSm = Shear_mag * (A.s.Biharm_const_xy[Q2(I, J)] + A.s.Biharm_const2_xy[Q2(I, J)] * Shear_mag);
          else AhSm = A.s.Biharm_const_xy[Q2(I, J)] * Shear_mag;
          A_ = max2(A_, AhSm);
          if (o.bound_Ah && !o.better_bound_Ah) A_ = min2(A_, A.s.Ah_Max_xy[Q2(I, J)]);
        }
        if (o.better_bound_Ah) {
          if (o.better_bound_Kh) A_ = min2(A_, visc_bound_rem * hrat_min * A.s.Ah_Max_xy[Q2(I, J)]);
          else A_ = min2(A_, hrat_min * A.s.Ah_Max_xy[Q2(I, J)]);
        }
        const double dDel2vdx = DY_dxBu(I, J) * (LX(s_d2v, i + 1, J) * g.IdyCv[V2(i + 1, J)] - LX(s_d2v, i, J) * g.IdyCv[V2(i, J)]);
        const double dDel2udy = DX_dyBu(I, J) * (LX(s_d2u, I, j + 1) * g.IdxCu[U2(I, j + 1)] - LX(s_d2u, I, j) * g.IdxCu[U2(I, j)]);
        const double d_str = A_ * (dDel2vdx + dDel2udy);
        str = str + d_str;
      }
      if (o.no_slip) r_xy[it] = str * (hq * A.s.reduction_xy[Q2(I, J)]);      // :1733-1740
      else r_xy[it] = str * (hq * g.mask2dBu[Q2(I, J)] * A.s.reduction_xy[Q2(I, J)]);
    }
  } END_POINTS
  __syncthreads();
  FOR_POINTS {
    LX(s_xx, i, j) = r_xx[it];
    LX(s_xy, I, J) = r_xy[it];
  } END_POINTS
  __syncthreads();

  // ---- diffu, diffv :1744-1770: the tile's own points (the western / southern edge points belong to the first tiles) ----
  const int Iw = (i0 == g.isc) ? is - 1 : is, Js = (j0 == g.jsc) ? js - 1 : js;
  FOR_POINTS {
    if (j >= js && j <= je && I >= Iw && I <= ie)
      P.diffu[U2(I, j)] = ((g.IdyCu[U2(I, j)] * (dy2h(i, j) * LX(s_xx, i, j) - dy2h(i + 1, j) * LX(s_xx, i + 1, j)) +
                            g.IdxCu[U2(I, j)] * (dx2q(I, J - 1) * LX(s_xy, I, J - 1) - dx2q(I, J) * LX(s_xy, I, J))) *
                           g.IareaCu[U2(I, j)]) / (LX(s_hu, I, j) + h_neglect);
    if (i >= is && i <= ie && J >= Js && J <= je)
      P.diffv[V2(i, J)] = ((g.IdyCv[V2(i, J)] * (dy2q(I - 1, J) * LX(s_xy, I - 1, J) - dy2q(I, J) * LX(s_xy, I, J)) -
                            g.IdxCv[V2(i, J)] * (dx2h(i, j) * LX(s_xx, i, j) - dx2h(i, j + 1) * LX(s_xx, i, j + 1))) *
                           g.IareaCv[V2(i, J)]) / (LX(s_hv, i, J) + h_neglect);
  } END_POINTS
#undef LX
#undef FOR_POINTS
#undef END_POINTS
}

int check_cs(const mom6hip_hor_visc_cs_t *cs, const char *who) {
  static const char *names[10] = {"LEITH_KH", "LEITH_AH", "USE_LEITHY", "USE_MEKE", "USE_GME", "ANISOTROPIC_VISCOSITY", "RE_AH", "KH_SIN_LAT",
                                  "USE_KH_BG_2D", "USE_ZB2020"};
  for (int n = 0; n < 10; n++) M6_REQUIRE(!cs->unsupported[n], "%s: %s is not provided by libmom6hip", who, names[n]);
  M6_REQUIRE(!(cs->no_slip && cs->biharmonic), "ERROR: NOSLIP and BIHARMONIC cannot be defined at the same time in MOM.");
  double *const *a = &cs->Kh_bg_xx;
  for (int n = 0; n < 16; n++) M6_REQUIRE(a[n] != nullptr, "%s: array %d of the control structure is missing", who, n);
  return 0;
}

HVOpt opt_of(const mom6hip_hor_visc_cs_t *cs) {
  HVOpt o;
  o.Kh = cs->Kh; o.Kh_bg_min = cs->Kh_bg_min; o.Kh_vel_scale = cs->Kh_vel_scale; o.Smag_Lap_const = cs->Smag_Lap_const;
  o.Ah = cs->Ah; o.Ah_vel_scale = cs->Ah_vel_scale; o.Ah_time_scale = cs->Ah_time_scale; o.Smag_bi_const = cs->Smag_bi_const;
  o.bound_Cor_vel = cs->bound_Cor_vel; o.bound_coef = cs->bound_coef;
  o.Laplacian = cs->Laplacian; o.biharmonic = cs->biharmonic; o.Smagorinsky_Kh = cs->Smagorinsky_Kh; o.Smagorinsky_Ah = cs->Smagorinsky_Ah;
  o.bound_Kh = cs->bound_Kh; o.better_bound_Kh = cs->better_bound_Kh; o.bound_Ah = cs->bound_Ah; o.better_bound_Ah = cs->better_bound_Ah;
  o.bound_Coriolis = cs->bound_Coriolis; o.add_LES_viscosity = cs->add_LES_viscosity; o.no_slip = cs->no_slip;
  o.use_land_mask = cs->use_land_mask; o.use_cont_thick = cs->use_cont_thick;
  return o;
}

}  // namespace

extern "C" int mom6hip_hor_visc_init(mom6hip_ctx_t *ctx, mom6hip_hor_visc_cs_t *cs, double dt, int32_t memspace) {
  M6_REQUIRE(ctx != nullptr && cs != nullptr, "hor_visc_init: null argument");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "hor_visc_init: bad memspace");
  if (check_cs(cs, "hor_visc_init")) return 1;
  M6_REQUIRE(dt > 0.0, "hor_visc_init: DT must be positive");
  const m6::GridDev g = ctx->g;
  M6_REQUIRE(g.dxT && g.dyT && g.IdxT && g.IdyT && g.dxCu && g.dyCu && g.dy_Cu && g.IdxCu && g.IdyCu && g.IareaCu && g.dxCv && g.dyCv &&
             g.dx_Cv && g.IdxCv && g.IdyCv && g.IareaCv && g.dxBu && g.dyBu && g.IdxBu && g.IdyBu && g.CoriolisBu && g.mask2dBu &&
             g.mask2dT && g.mask2dCu && g.mask2dCv, "hor_visc_init: a required grid metric is missing");
  M6_REQUIRE(g.isc - g.isd >= 2 && g.jsc - g.jsd >= 2, "hor_visc_init: the halo must be at least 2 points wide");
  const size_t bH = sizeof(double) * (size_t)g.nih * g.njh, bQ = sizeof(double) * (size_t)(g.nih + 1) * (g.njh + 1);
  const size_t bU = sizeof(double) * (size_t)(g.nih + 1) * g.njh, bV = sizeof(double) * (size_t)g.nih * (g.njh + 1);
  hipStream_t s = ctx->stream;
  m6::Stager st(ctx, memspace);
  InitArgs A;
  A.g = g; A.o = opt_of(cs); A.dt = dt;
  double *const *src = &cs->Kh_bg_xx;
  double **dst = &A.s.Kh_bg_xx;
  for (int n = 0; n < 16; n++) dst[n] = st.out(src[n], n < 8 ? bH : bQ);
  A.u0u = (double *)st.scratch(bU); A.u0v = (double *)st.scratch(bU); A.v0u = (double *)st.scratch(bV); A.v0v = (double *)st.scratch(bV);
  M6_REQUIRE(!st.failed(), "hor_visc_init: staging failed");
  for (int n = 0; n < 16; n++) M6_HIP(hipMemsetAsync(dst[n], 0, n < 8 ? bH : bQ, s));
  M6_HIP(hipMemsetAsync(A.u0u, 0, bU, s)); M6_HIP(hipMemsetAsync(A.u0v, 0, bU, s));
  M6_HIP(hipMemsetAsync(A.v0u, 0, bV, s)); M6_HIP(hipMemsetAsync(A.v0v, 0, bV, s));
  if (cs->Laplacian || cs->biharmonic) {
    const dim3 grid((g.nih + 255) / 256, g.njh);
    hipLaunchKernelGGL(hv_init1_kernel, grid, dim3(256), 0, s, A);
    if (cs->biharmonic && cs->better_bound_Ah) hipLaunchKernelGGL(hv_init2_kernel, grid, dim3(256), 0, s, A);
    M6_HIP(hipGetLastError());
  }
  cs->initialized = 1;
  return st.finish();
}

// the launches of horizontal_viscosity on device arrays (also called by the split RK2 step at :860 and :1543)
namespace m6 {
int horizontal_viscosity_dev(mom6hip_ctx_t *ctx, const mom6hip_hor_visc_cs_t *cs, const double *u, const double *v, const double *h,
                             double *diffu, double *diffv, const double *hu_cont, const double *hv_cont) {
  const m6::GridDev g = ctx->g;
  if (!(cs->Laplacian || cs->biharmonic)) return 0;      // :451
  HVArgs A;
  A.g = g; A.o = opt_of(cs);
  double *const *src = &cs->Kh_bg_xx;
  double **dst = &A.s.Kh_bg_xx;
  for (int n = 0; n < 16; n++) dst[n] = src[n];
  A.u = u; A.v = v; A.h = h; A.hu_cont = hu_cont; A.hv_cont = hv_cont; A.diffu = diffu; A.diffv = diffv;
  const int ni = g.iec - g.isc + 1, nj = g.jec - g.jsc + 1;
  const int ntx = (ni + HV_TI - 1) / HV_TI, nty = (nj + HV_TJ - 1) / HV_TJ, ntiles = ntx * nty;
  const long nblocks = (long)((ntiles + 7) / 8) * 8 * g.nk;
  M6_REQUIRE(nblocks < (1L << 31), "horizontal_viscosity: the grid is too large for one launch");
  hipLaunchKernelGGL(hv_fused_kernel, dim3((unsigned)nblocks), dim3(HV_NT), 0, ctx->stream, A, ntx, ntiles);
  M6_HIP(hipGetLastError());
  return 0;
}
}  // namespace m6

extern "C" int mom6hip_horizontal_viscosity(mom6hip_ctx_t *ctx, const mom6hip_hor_visc_cs_t *cs, const double *u, const double *v,
                                            const double *h, double *diffu, double *diffv, double dt, const double *hu_cont,
                                            const double *hv_cont, int32_t memspace) {
  (void)dt;
  M6_REQUIRE(ctx != nullptr, "MOM_hor_visc: Module must be initialized before it is used.");
  M6_REQUIRE(cs && u && v && h && diffu && diffv, "horizontal_viscosity: null argument");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "horizontal_viscosity: bad memspace");
  M6_REQUIRE(cs->initialized, "MOM_hor_visc: Module must be initialized before it is used.");
  if (check_cs(cs, "horizontal_viscosity")) return 1;
  const m6::GridDev g = ctx->g;
  M6_REQUIRE(g.isc - g.isd >= 2 && g.jsc - g.jsd >= 2, "horizontal_viscosity: the halo must be at least 2 points wide");
  const size_t bH2 = sizeof(double) * (size_t)g.nih * g.njh, bQ2 = sizeof(double) * (size_t)(g.nih + 1) * (g.njh + 1);
  const size_t bH = sizeof(double) * (size_t)g.nh3(), bU = sizeof(double) * (size_t)g.nu3(), bV = sizeof(double) * (size_t)g.nv3();
  m6::Stager st(ctx, memspace);
  mom6hip_hor_visc_cs_t dcs = *cs;
  double *const *src = &cs->Kh_bg_xx;
  double **dst = &dcs.Kh_bg_xx;
  for (int n = 0; n < 16; n++) dst[n] = (double *)st.in(src[n], n < 8 ? bH2 : bQ2);
  const double *du = st.in(u, bU), *dv = st.in(v, bV), *dh = st.in(h, bH);
  const double *dhu = st.in(hu_cont, bU), *dhv = st.in(hv_cont, bV);
  double *ddu = st.inout(diffu, bU), *ddv = st.inout(diffv, bV);      // (only the compute ranges are written)
  M6_REQUIRE(!st.failed(), "horizontal_viscosity: staging failed");
  if (m6::horizontal_viscosity_dev(ctx, &dcs, du, dv, dh, ddu, ddv, dhu, dhv)) return 1;
  return st.finish();
}
