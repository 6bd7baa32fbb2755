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
#include <vector>

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
// ---- the fused kernel ------------------------------------------------------------------------------------------------
// One block = one tile of HV_TI x HV_TJ points of one layer.  The tile is treated as the reference treats a PE's compute
// domain (is_t : ie_t, js_t : je_t): every intermediate is formed over the reference's index ranges relative to the tile
// (sh_xx over is_t-2 : ie_t+2, Del2u over is_t-2 : ie_t+1, str_xx over is_t-1 : ie_t+1 ...), held in LDS in one local frame
// (li, lj) <-> (i0 - 2 + li, j0 - 2 + lj), and only the tile's own diffu / diffv go back to HBM -- every value is the
// result of the same operations on the same operands as in the layer-wide form.  The stresses overwrite the strains in
// LDS (the strains are dead by then).  512 threads sweep the 68 x 20 points of the frame in 3 passes per stage.
//
// Block order: the two-dimensional metrics and coefficients a point needs are 6 times the 3-D traffic if they come from
// HBM for every layer.  blockIdx.x is decoded so that the blocks an XCD receives back to back (ids = xcd mod 8) are the nk
// layers of ONE tile: its metrics are read into that XCD's L2 once and hit there for the other layers (measured: 2.5 GB
// fetched per call at 1440x1080x75 for 2.8 GB of u, v, h).
//
// Addressing: the 22 grid metrics (and the products hor_visc_init keeps: dx2h, DX_dyT, Idx2dyCu ...) are gathered once
// per context into planes of ONE shape, (nih+1) x (njh+1) -- the shape of a q-point array, in which an h, u or v point
// (i, j) sits at the same place as the q point (I, J) -- so a point needs one 32-bit byte offset for all of them, every
// load is `global_load v, voffset, s[base] offset:imm`, and one base pointer replaces 40 (which the kernel's scalar
// registers could not hold: the first form spent a fifth of its instructions on v_readlane of spilled pointers).
enum HVPlane { P_DX2H, P_DY2H, P_DXDYT, P_DYDXT, P_MASKT, P_DX2Q, P_DY2Q, P_DXDYBU, P_DYDXBU, P_MASKBU,
               P_IDXCU, P_IDYCU, P_IAREACU, P_IDX2DYCU, P_IDXDY2U, P_MASKCU, P_IDXCV, P_IDYCV, P_IAREACV, P_IDX2DYCV, P_IDXDY2V,
               P_MASKCV, HV_NPLANES };

// thread (pi, pj) over the (nih+1) x (njh+1) frame: (i, j) = (isd - 1 + pi, jsd - 1 + pj)
__global__ __launch_bounds__(256) void hv_pack_kernel(m6::GridDev g, double *pk) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x, pj = blockIdx.y;
  if (pi > g.nih) return;
  const int i = g.isd - 1 + pi, j = g.jsd - 1 + pj, I = i, J = j;
  const size_t plane = (size_t)(g.nih + 1) * (g.njh + 1), n = (size_t)pj * (g.nih + 1) + pi;
  const bool inh = pi >= 1 && pj >= 1, inu = pj >= 1, inv = pi >= 1;
  auto put = [&](int m, double x) { pk[plane * m + n] = x; };
  put(P_DX2H, inh ? dx2h(i, j) : 0.0); put(P_DY2H, inh ? dy2h(i, j) : 0.0);
  put(P_DXDYT, inh ? DX_dyT(i, j) : 0.0); put(P_DYDXT, inh ? DY_dxT(i, j) : 0.0);
  put(P_MASKT, inh ? g.mask2dT[H2(i, j)] : 0.0);
  put(P_DX2Q, dx2q(I, J)); put(P_DY2Q, dy2q(I, J)); put(P_DXDYBU, DX_dyBu(I, J)); put(P_DYDXBU, DY_dxBu(I, J));
  put(P_MASKBU, g.mask2dBu[Q2(I, J)]);
  put(P_IDXCU, inu ? g.IdxCu[U2(I, j)] : 0.0); put(P_IDYCU, inu ? g.IdyCu[U2(I, j)] : 0.0);
  put(P_IAREACU, inu ? g.IareaCu[U2(I, j)] : 0.0); put(P_IDX2DYCU, inu ? Idx2dyCu(I, j) : 0.0);
  put(P_IDXDY2U, inu ? Idxdy2u(I, j) : 0.0); put(P_MASKCU, inu ? g.mask2dCu[U2(I, j)] : 0.0);
  put(P_IDXCV, inv ? g.IdxCv[V2(i, J)] : 0.0); put(P_IDYCV, inv ? g.IdyCv[V2(i, J)] : 0.0);
  put(P_IAREACV, inv ? g.IareaCv[V2(i, J)] : 0.0); put(P_IDX2DYCV, inv ? Idx2dyCv(i, J) : 0.0);
  put(P_IDXDY2V, inv ? Idxdy2v(i, J) : 0.0); put(P_MASKCV, inv ? g.mask2dCv[V2(i, J)] : 0.0);
}

// the options the stress stage reads, as bits of one scalar register (the struct of ints and doubles cost 40)
struct HVFlags {
  unsigned bits;
  __host__ __device__ bool operator[](int n) const { return (bits >> n) & 1u; }
};
enum { F_LAPLACIAN, F_BIHARMONIC, F_SMAG_KH, F_SMAG_AH, F_BOUND_KH, F_BETTER_BOUND_KH, F_BOUND_AH, F_BETTER_BOUND_AH, F_BOUND_CORIOLIS,
       F_ADD_LES, F_NO_SLIP, F_LAND_MASK, F_CONT_THICK };
struct HVFOpt {      // HVOpt as the fused kernel sees it
  bool Laplacian, biharmonic, Smagorinsky_Kh, Smagorinsky_Ah, bound_Kh, better_bound_Kh, bound_Ah, better_bound_Ah, bound_Coriolis,
      add_LES_viscosity, no_slip, use_land_mask, use_cont_thick;
  double Kh_bg_min;
};

struct HVFArgs {
  HVFlags flags;
  double Kh_bg_min;
  HVStatic s;
  const char *pk;                  // the packed metric planes
  unsigned plane_bytes;
  int isc, iec, jsc, jec, isd, jsd, nih, njh, nk;
  double h_neglect;
  const double *u, *v, *h, *hu_cont, *hv_cont;
  double *diffu, *diffv;
  // MEKE (generic configuration only): MEKE%Ku, MEKE%Au (h points, may be null); the layer-integrated stresses of every layer for the
  // frictional work (hv_frictwork_kernel), h- and q-shaped 3-D arrays, null unless MEKE%mom_src is wanted
  const double *Ku, *Au;
  double *str_xx_out, *str_xy_out;
  // open boundaries (generic configuration only; null without): what the reference's loops over the segments leave, as maps
  //   obc_q [q points]   bit 0: dvdx = 0, bit 1: dudy = 0 (:733-790), bit 2: dDel2vdx = 0, bit 3: dDel2udy = 0 (:1388-1409);
  //                      OBC_COMPUTED_STRAIN (:741-748, :766-773): bit 4 / 5: dudy from obc_tang_u at a northern / southern segment,
  //                      bit 6 / 7: dvdx from obc_tang_v at an eastern / western one
  //   obc_fu, obc_fv     bit 0: Del2u | Del2v = 0 (:889-903), bit 1: diffu | diffv = 0 (:1751-1782), bits 2-3: the thickness of the face is that
  //                      of its first (1) or second (2) cell (:791-819)
  //   obc_hu, obc_hv     the face whose thickness this face takes after the projections across the segments' corner points (:821-849), -1: its own
  const int32_t *obc_q, *obc_fu, *obc_fv, *obc_hu, *obc_hv;
  const double *obc_tang_u, *obc_tang_v;      // segment%tangential_vel at the q points, all layers
};


// CFG = HV_GENERIC: the options are read from the argument; otherwise they are the bits of CFG at compile time (the
// production set -- biharmonic Smagorinsky with the better bounds -- is instantiated: the branches and the static arrays it
// does not use are gone, which is what lets the remaining base pointers stay in scalar registers).
constexpr unsigned HV_GENERIC = ~0u;
constexpr unsigned HV_BIH_SMAG = (1u << F_BIHARMONIC) | (1u << F_SMAG_AH) | (1u << F_BOUND_KH) | (1u << F_BETTER_BOUND_KH) | (1u << F_BOUND_AH) |
                                 (1u << F_BETTER_BOUND_AH) | (1u << F_LAND_MASK);

template <unsigned CFG, int HV_TI, int HV_TJ, int HV_NT, int WPE>
__global__ __launch_bounds__(HV_NT, WPE) void hv_fused_kernel(HVFArgs A, int ntx, int ntiles) {
  constexpr int HV_W = HV_TI + 4, HV_H = HV_TJ + 4, HV_NP = HV_W * HV_H, HV_NIT = (HV_NP + HV_NT - 1) / HV_NT;
  __shared__ double s_xx[HV_NP], s_xy[HV_NP], s_d2u[HV_NP], s_d2v[HV_NP], s_hu[HV_NP], s_hv[HV_NP];
  const HVFlags F = {CFG == HV_GENERIC ? A.flags.bits : CFG};
  const HVFOpt o = {F[F_LAPLACIAN], F[F_BIHARMONIC], F[F_SMAG_KH], F[F_SMAG_AH], F[F_BOUND_KH], F[F_BETTER_BOUND_KH], F[F_BOUND_AH],
                    F[F_BETTER_BOUND_AH], F[F_BOUND_CORIOLIS], F[F_ADD_LES], F[F_NO_SLIP], F[F_LAND_MASK], F[F_CONT_THICK], A.Kh_bg_min};
  const int L = blockIdx.x, m = L >> 3;
  const int k = m % A.nk, tile = (m / A.nk) * 8 + (L & 7);
  if (tile >= ntiles) return;
  const int i0 = A.isc + (tile % ntx) * HV_TI, j0 = A.jsc + (tile / ntx) * HV_TJ;
  const int ib = i0 - 2, jb = j0 - 2;
  // the tile as a compute domain
  const int is = i0, ie = min(i0 + HV_TI - 1, A.iec), js = j0, je = min(j0 + HV_TJ - 1, A.jec);
  const int Isq = is - 1, Ieq = ie, Jsq = js - 1, Jeq = je;
  const double h_neglect = A.h_neglect;
  // row strides in bytes: q / u shaped rows, h / v shaped rows
  const unsigned RQ = (unsigned)(A.nih + 1) * 8u, RH = (unsigned)A.nih * 8u;
  // the layer's planes of the 3-D arrays (wave-uniform)
  const size_t kH = (size_t)A.nih * A.njh * k, kU = (size_t)(A.nih + 1) * A.njh * k, kV = (size_t)A.nih * (A.njh + 1) * k;
  const char *u_k = (const char *)(A.u + kU), *v_k = (const char *)(A.v + kV), *h_k = (const char *)(A.h + kH);
  const bool cont = o.use_cont_thick && A.hu_cont && A.hv_cont;
  const char *huc_k = cont ? (const char *)(A.hu_cont + kU) : nullptr, *hvc_k = cont ? (const char *)(A.hv_cont + kV) : nullptr;
  char *du_k = (char *)(A.diffu + kU), *dv_k = (char *)(A.diffv + kV);
  const char *pk = A.pk;
  const unsigned PB = A.plane_bytes;
  const bool obc = (CFG == HV_GENERIC) && A.obc_q != nullptr;
  // the thickness of the face fs of direction d (0: u, 1: v) with open boundaries: that of the face the projections across the segments'
  // corner points take it from, there the cell inside a segment or the plain interpolation (:740-849)
  auto h_face_obc = [&](int d, int fs) -> double {
    const int32_t *hs = d ? A.obc_hv : A.obc_hu, *fc = d ? A.obc_fv : A.obc_fu;
    const int f = hs[fs] >= 0 ? hs[fs] : fs;
    const int side = (fc[f] >> 2) & 3;
    const int nw = d ? A.nih : A.nih + 1, r = f / nw, c = f - r * nw;
    // the two cells of the face in the h-shaped arrays, and in the packed planes (q-shaped, offset by one point)
    const int h0 = d ? (r - 1) * A.nih + c : r * A.nih + c - 1, h1 = d ? r * A.nih + c : r * A.nih + c;
    const double *hk = (const double *)h_k;
    if (side == 1) return hk[h0];
    if (side == 2) return hk[h1];
    if (d == 0 && huc_k && (c - 1 + A.isd) >= A.isc - 2) return ((const double *)huc_k)[f];
    if (d == 1 && hvc_k && (r - 1 + A.jsd) >= A.jsc - 2) return ((const double *)hvc_k)[f];
    if (o.use_land_mask) {
      const double *mT = (const double *)(pk + (size_t)P_MASKT * PB);
      const int q0 = d ? r * (A.nih + 1) + c + 1 : (r + 1) * (A.nih + 1) + c, q1 = d ? (r + 1) * (A.nih + 1) + c + 1 : (r + 1) * (A.nih + 1) + c + 1;
      return 0.5 * (mT[q0] * hk[h0] + mT[q1] * hk[h1]);
    }
    return 0.5 * (hk[h0] + hk[h1]);
  };

#define LX(a, di, dj) a[lo + (dj) * HV_W + (di)]
// loads at the point (i + di, j + dj): packed metric plane m; an h-, u-, v- or q-shaped array with base pointer b
#define LDB(b, off, R, di, dj) (*(const double *)(((const char *)(b) + (di) * 8) + ((dj) == 0 ? (off) : (dj) > 0 ? (off) + (R) : (off) - (R))))
#define G(m, di, dj) LDB(pk + (size_t)(m) * PB, bq, RQ, di, dj)
#define AH(b, di, dj) LDB(b, bh, RH, di, dj)
#define AU(b, di, dj) LDB(b, bu, RQ, di, dj)
#define AV(b, di, dj) LDB(b, bv, RH, di, dj)
#define AQ(b, di, dj) LDB(b, bq, RQ, di, dj)
#define FOR_POINTS                                                                   \
  _Pragma("unroll") for (int it = 0; it < HV_NIT; it++) {                            \
    const int p = (int)threadIdx.x + it * HV_NT;                                     \
    const int lj = p / HV_W, li = p - lj * HV_W, lo = p;                             \
    const int i = ib + li, j = jb + lj, I = i, J = j;                                \
    const unsigned cj = (unsigned)(j - A.jsd), ci = (unsigned)(i - A.isd);           \
    const unsigned bq = ((cj + 1u) * (unsigned)(A.nih + 1) + ci + 1u) * 8u;          \
    const unsigned bh = (cj * (unsigned)A.nih + ci) * 8u;                            \
    const unsigned bu = (cj * (unsigned)(A.nih + 1) + ci + 1u) * 8u;                 \
    const unsigned bv = ((cj + 1u) * (unsigned)A.nih + ci) * 8u;                     \
    (void)I; (void)J; (void)bq; (void)bh; (void)bu; (void)bv; (void)lo;              \
    if (p < HV_NP)
#define END_POINTS }
// thicknesses at velocity points :740-765 (land mask or not; hu_cont / hv_cont inside their ranges with USE_CONT_THICKNESS)
#define HU_AT(di)                                                                                                        \
  ((huc_k && I + (di) >= A.isc - 2) ? AU(huc_k, di, 0)                                                                    \
   : o.use_land_mask ? 0.5 * (G(P_MASKT, di, 0) * AH(h_k, di, 0) + G(P_MASKT, (di) + 1, 0) * AH(h_k, (di) + 1, 0))          \
                     : 0.5 * (AH(h_k, di, 0) + AH(h_k, (di) + 1, 0)))

  // ---- strains and the thicknesses at velocity points ----
  FOR_POINTS {
    if (j >= Jsq - 1 && j <= Jeq + 2 && i >= Isq - 1 && i <= Ieq + 2) {      // horizontal tension :693-699
      const double dudx = G(P_DYDXT, 0, 0) * (G(P_IDYCU, 0, 0) * AU(u_k, 0, 0) - G(P_IDYCU, -1, 0) * AU(u_k, -1, 0));
      const double dvdy = G(P_DXDYT, 0, 0) * (G(P_IDXCV, 0, 0) * AV(v_k, 0, 0) - G(P_IDXCV, 0, -1) * AV(v_k, 0, -1));
      LX(s_xx, 0, 0) = dudx - dvdy;
    }
    if (J >= js - 2 && J <= Jeq + 1 && I >= is - 2 && I <= Ieq + 1) {      // shearing strain :702-705, :852-864
      double dvdx = G(P_DYDXBU, 0, 0) * (AV(v_k, 1, 0) * G(P_IDYCV, 1, 0) - AV(v_k, 0, 0) * G(P_IDYCV, 0, 0));
      double dudy = G(P_DXDYBU, 0, 0) * (AU(u_k, 0, 1) * G(P_IDXCU, 0, 1) - AU(u_k, 0, 0) * G(P_IDXCU, 0, 0));
      if (obc) {
        const int qc = A.obc_q[bq >> 3];
        if (qc & 1) dvdx = 0.;
        if (qc & 2) dudy = 0.;
        if (qc & 0xf0) {      // OBC_COMPUTED_STRAIN
          const size_t q3 = (size_t)(bq >> 3) + (size_t)(A.nih + 1) * (A.njh + 1) * k;
          if (qc & 16) dudy = 2.0 * G(P_DXDYBU, 0, 0) * (A.obc_tang_u[q3] - AU(u_k, 0, 0)) * G(P_IDXCU, 0, 0);
          if (qc & 32) dudy = 2.0 * G(P_DXDYBU, 0, 0) * (AU(u_k, 0, 1) - A.obc_tang_u[q3]) * G(P_IDXCU, 0, 1);
          if (qc & 64) dvdx = 2.0 * G(P_DYDXBU, 0, 0) * (A.obc_tang_v[q3] - AV(v_k, 0, 0)) * G(P_IDYCV, 0, 0);
          if (qc & 128) dvdx = 2.0 * G(P_DYDXBU, 0, 0) * (AV(v_k, 1, 0) - A.obc_tang_v[q3]) * G(P_IDYCV, 1, 0);
        }
      }
      if (o.no_slip) LX(s_xy, 0, 0) = (2.0 - G(P_MASKBU, 0, 0)) * (dvdx + dudy);
      else LX(s_xy, 0, 0) = G(P_MASKBU, 0, 0) * (dvdx + dudy);
    }
    if (j >= js - 1 && j <= je + 1 && I >= is - 2 && I <= ie + 1) {      // :740-765
      const int fs = obc ? (int)(bu >> 3) : 0;
      if (obc && (A.obc_hu[fs] >= 0 || (A.obc_fu[fs] & 12))) LX(s_hu, 0, 0) = h_face_obc(0, fs);
      else LX(s_hu, 0, 0) = HU_AT(0);
    }
    if (J >= js - 2 && J <= je + 1 && i >= is - 1 && i <= ie + 1) {
      const int fs = obc ? (int)(bv >> 3) : 0;
      if (obc && (A.obc_hv[fs] >= 0 || (A.obc_fv[fs] & 12))) LX(s_hv, 0, 0) = h_face_obc(1, fs);
      else if (hvc_k && J >= A.jsc - 2) LX(s_hv, 0, 0) = AV(hvc_k, 0, 0);
      else if (o.use_land_mask) LX(s_hv, 0, 0) = 0.5 * (G(P_MASKT, 0, 0) * AH(h_k, 0, 0) + G(P_MASKT, 0, 1) * AH(h_k, 0, 1));
      else LX(s_hv, 0, 0) = 0.5 * (AH(h_k, 0, 0) + AH(h_k, 0, 1));
    }
  } END_POINTS
  __syncthreads();

  // ---- the Laplacian of the velocity :882-891 ----
  if (o.biharmonic) {
    FOR_POINTS {
      if (j >= js - 1 && j <= Jeq + 1 && I >= Isq - 1 && I <= Ieq + 1) {
        LX(s_d2u, 0, 0) = G(P_IDXDY2U, 0, 0) * (G(P_DY2H, 1, 0) * LX(s_xx, 1, 0) - G(P_DY2H, 0, 0) * LX(s_xx, 0, 0)) +
                          G(P_IDX2DYCU, 0, 0) * (G(P_DX2Q, 0, 0) * LX(s_xy, 0, 0) - G(P_DX2Q, 0, -1) * LX(s_xy, 0, -1));
        if (obc && (A.obc_fu[bu >> 3] & 1)) LX(s_d2u, 0, 0) = 0.;
      }
      if (J >= Jsq - 1 && J <= Jeq + 1 && i >= is - 1 && i <= Ieq + 1) {
        LX(s_d2v, 0, 0) = G(P_IDXDY2V, 0, 0) * (G(P_DY2Q, 0, 0) * LX(s_xy, 0, 0) - G(P_DY2Q, -1, 0) * LX(s_xy, -1, 0)) -
                          G(P_IDX2DYCV, 0, 0) * (G(P_DX2H, 0, 1) * LX(s_xx, 0, 1) - G(P_DX2H, 0, 0) * LX(s_xx, 0, 0));
        if (obc && (A.obc_fv[bv >> 3] & 1)) LX(s_d2v, 0, 0) = 0.;
      }
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
      const double sxx = LX(s_xx, 0, 0);
      double Shear_mag = 0.0, hrat_min = 0.0, visc_bound_rem = 0.0;
      if (smag) {      // :1056-1063
        const double sh_xx_sq = sxx * sxx;
        const double sh_xy_sq = 0.25 * ((LX(s_xy, -1, -1) * LX(s_xy, -1, -1) + LX(s_xy, 0, 0) * LX(s_xy, 0, 0)) +
                                        (LX(s_xy, -1, 0) * LX(s_xy, -1, 0) + LX(s_xy, 0, -1) * LX(s_xy, 0, -1)));
        Shear_mag = sqrt(sh_xx_sq + sh_xy_sq);
      }
      const double h_here = AH(h_k, 0, 0);
      if (bb) {      // :1065-1076
        const double h_min = min4(LX(s_hu, 0, 0), LX(s_hu, -1, 0), LX(s_hv, 0, 0), LX(s_hv, 0, -1));
        hrat_min = min2(1.0, h_min / (h_here + h_neglect));
        if (o.better_bound_Kh) visc_bound_rem = 1.0;
      }
      double str = 0.0;
      if (o.Laplacian) {      // :1078-1214
        double K_ = AH(A.s.Kh_bg_xx, 0, 0);
        if (o.add_LES_viscosity) {
          if (o.Smagorinsky_Kh) K_ = K_ + AH(A.s.Laplac2_const_xx, 0, 0) * Shear_mag;
        } else {
          if (o.Smagorinsky_Kh) K_ = max2(K_, AH(A.s.Laplac2_const_xx, 0, 0) * Shear_mag);
        }
        if (legacy_bound) K_ = min2(K_, AH(A.s.Kh_Max_xx, 0, 0));
        K_ = max2(K_, o.Kh_bg_min);
        if (CFG == HV_GENERIC && A.Ku) K_ = K_ + AH(A.Ku, 0, 0);      // :1141-1151
        if (o.better_bound_Kh) {
          const double KhM = AH(A.s.Kh_Max_xx, 0, 0);
          if (K_ >= hrat_min * KhM) {
            visc_bound_rem = 0.0;
            K_ = hrat_min * KhM;
          } else {
            visc_bound_rem = 1.0 - K_ / (hrat_min * KhM);
          }
        }
        str = -K_ * sxx;
      }
      if (o.biharmonic) {      // :1227-1380
        double A_ = AH(A.s.Ah_bg_xx, 0, 0);
        if (o.Smagorinsky_Ah) {
          double AhSm;
          if (o.bound_Coriolis) AhSm = Shear_mag * (AH(A.s.Biharm_const_xx, 0, 0) + AH(A.s.Biharm_const2_xx, 0, 0) * Shear_mag);
          else AhSm = AH(A.s.Biharm_const_xx, 0, 0) * Shear_mag;
          A_ = max2(A_, AhSm);
          if (o.bound_Ah && !o.better_bound_Ah) A_ = min2(A_, AH(A.s.Ah_Max_xx, 0, 0));
        }
        if (CFG == HV_GENERIC && A.Au) A_ = A_ + AH(A.Au, 0, 0);      // :1318-1323
        if (o.better_bound_Ah) {
          if (o.better_bound_Kh) A_ = min2(A_, visc_bound_rem * hrat_min * AH(A.s.Ah_Max_xx, 0, 0));
          else A_ = min2(A_, hrat_min * AH(A.s.Ah_Max_xx, 0, 0));
        }
        const double d_del2u = G(P_IDYCU, 0, 0) * LX(s_d2u, 0, 0) - G(P_IDYCU, -1, 0) * LX(s_d2u, -1, 0);
        const double d_del2v = G(P_IDXCV, 0, 0) * LX(s_d2v, 0, 0) - G(P_IDXCV, 0, -1) * LX(s_d2v, 0, -1);
        const double d_str = A_ * (G(P_DYDXT, 0, 0) * d_del2u - G(P_DXDYT, 0, 0) * d_del2v);
        str = str + d_str;
      }
      r_xx[it] = str * (h_here * AH(A.s.reduction_xx, 0, 0));      // :1728
    }

    if (J >= js - 1 && J <= Jeq && I >= is - 1 && I <= Ieq) {      // ---- q point ----
      const double sxy = LX(s_xy, 0, 0);
      double Shear_mag = 0.0;
      if (smag) {      // :1414-1421
        const double sh_xy_sq = sxy * sxy;
        const double sh_xx_sq = 0.25 * ((LX(s_xx, 0, 0) * LX(s_xx, 0, 0) + LX(s_xx, 1, 1) * LX(s_xx, 1, 1)) +
                                        (LX(s_xx, 0, 1) * LX(s_xx, 0, 1) + LX(s_xx, 1, 0) * LX(s_xx, 1, 0)));
        Shear_mag = sqrt(sh_xy_sq + sh_xx_sq);
      }
      const double hu0 = LX(s_hu, 0, 0), hu1 = LX(s_hu, 0, 1);
      const double hv0 = LX(s_hv, 0, 0), hv1 = LX(s_hv, 1, 0);
      const double h2uq = 4.0 * (hu0 * hu1);      // :1423-1428
      const double h2vq = 4.0 * (hv0 * hv1);
      double hq = (2.0 * (h2uq * h2vq)) / (h_neglect3 + (h2uq + h2vq) * ((hu0 + hu1) + (hv0 + hv1)));
      double hrat_min = 0.0, visc_bound_rem = 0.0;
      if (bb) {      // :1430-1441
        const double h_min = min4(hu0, hu1, hv0, hv1);
        hrat_min = min2(1.0, h_min / (hq + h_neglect));
        if (o.better_bound_Kh) visc_bound_rem = 1.0;
      }
      if (o.no_slip && (G(P_MASKBU, 0, 0) < 0.5)) {      // coastal vorticity points :1443-1466
        const double mu0 = G(P_MASKCU, 0, 0), mu1 = G(P_MASKCU, 0, 1), mv0 = G(P_MASKCV, 0, 0), mv1 = G(P_MASKCV, 1, 0);
        if ((mu0 + mu1) + (mv0 + mv1) > 0.0) {
          const double hu = mu0 * hu0 + mu1 * hu1;
          const double hv = mv0 * hv0 + mv1 * hv1;
          if ((mu0 + mu1) * (mv0 + mv1) == 0.0) {
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
        double K_ = AQ(A.s.Kh_bg_xy, 0, 0);
        if (o.Smagorinsky_Kh) {
          if (o.add_LES_viscosity) K_ = K_ + AQ(A.s.Laplac2_const_xy, 0, 0) * Shear_mag;
          else K_ = max2(K_, AQ(A.s.Laplac2_const_xy, 0, 0) * Shear_mag);
        }
        if (legacy_bound) K_ = min2(K_, AQ(A.s.Kh_Max_xy, 0, 0));
        K_ = max2(K_, o.Kh_bg_min);
        if (CFG == HV_GENERIC && A.Ku)      // :1537-1541 (meke_res_fn = 1)
          K_ = K_ + 0.25 * ((AH(A.Ku, 0, 0) + AH(A.Ku, 1, 1)) + (AH(A.Ku, 1, 0) + AH(A.Ku, 0, 1))) * 1.0;
        if (o.better_bound_Kh) {
          const double KhM = AQ(A.s.Kh_Max_xy, 0, 0);
          if (K_ >= hrat_min * KhM) {
            visc_bound_rem = 0.0;
            K_ = hrat_min * KhM;
          } else if (hrat_min * KhM > 0.) {
            visc_bound_rem = 1.0 - K_ / (hrat_min * KhM);
          }
        }
        str = -K_ * sxy;
      }
      if (o.biharmonic) {      // :1598-1693, with the gradient of the Laplacian :1382-1387
        double A_ = AQ(A.s.Ah_bg_xy, 0, 0);
        if (o.Smagorinsky_Ah) {
          double AhSm;
          if (o.bound_Coriolis) AhSm = Shear_mag * (AQ(A.s.Biharm_const_xy, 0, 0) + AQ(A.s.Biharm_const2_xy, 0, 0) * Shear_mag);
          else AhSm = AQ(A.s.Biharm_const_xy, 0, 0) * Shear_mag;
          A_ = max2(A_, AhSm);
          if (o.bound_Ah && !o.better_bound_Ah) A_ = min2(A_, AQ(A.s.Ah_Max_xy, 0, 0));
        }
        if (CFG == HV_GENERIC && A.Au)      // :1634-1639
          A_ = A_ + 0.25 * ((AH(A.Au, 0, 0) + AH(A.Au, 1, 1)) + (AH(A.Au, 1, 0) + AH(A.Au, 0, 1)));
        if (o.better_bound_Ah) {
          if (o.better_bound_Kh) A_ = min2(A_, visc_bound_rem * hrat_min * AQ(A.s.Ah_Max_xy, 0, 0));
          else A_ = min2(A_, hrat_min * AQ(A.s.Ah_Max_xy, 0, 0));
        }
        double dDel2vdx = G(P_DYDXBU, 0, 0) * (LX(s_d2v, 1, 0) * G(P_IDYCV, 1, 0) - LX(s_d2v, 0, 0) * G(P_IDYCV, 0, 0));
        double dDel2udy = G(P_DXDYBU, 0, 0) * (LX(s_d2u, 0, 1) * G(P_IDXCU, 0, 1) - LX(s_d2u, 0, 0) * G(P_IDXCU, 0, 0));
        if (obc) {
          const int qc = A.obc_q[bq >> 3];
          if (qc & 4) dDel2vdx = 0.;
          if (qc & 8) dDel2udy = 0.;
        }
        const double d_str = A_ * (dDel2vdx + dDel2udy);
        str = str + d_str;
      }
      if (o.no_slip) r_xy[it] = str * (hq * AQ(A.s.reduction_xy, 0, 0));      // :1733-1740
      else r_xy[it] = str * (hq * G(P_MASKBU, 0, 0) * AQ(A.s.reduction_xy, 0, 0));
    }
  } END_POINTS
  __syncthreads();
  FOR_POINTS {
    LX(s_xx, 0, 0) = r_xx[it];
    LX(s_xy, 0, 0) = r_xy[it];
  } END_POINTS
  __syncthreads();

  // ---- diffu, diffv :1744-1770: the tile's own points (the western / southern edge points belong to the first tiles) ----
  const int Iw = (i0 == A.isc) ? is - 1 : is, Js = (j0 == A.jsc) ? js - 1 : js;
  if (CFG == HV_GENERIC && A.str_xx_out) {      // the stresses of this layer, for MEKE%mom_src
    char *sxx_k = (char *)(A.str_xx_out + kH), *sxy_k = (char *)(A.str_xy_out + (size_t)(A.nih + 1) * (A.njh + 1) * k);
    FOR_POINTS {
      if (j >= js && j <= je && i >= is && i <= ie) *(double *)(sxx_k + bh) = LX(s_xx, 0, 0);
      if (J >= Js && J <= je && I >= Iw && I <= ie) *(double *)(sxy_k + bq) = LX(s_xy, 0, 0);
    } END_POINTS
  }
  FOR_POINTS {
    if (j >= js && j <= je && I >= Iw && I <= ie)
      *(double *)(du_k + bu) = (obc && (A.obc_fu[bu >> 3] & 2)) ? 0. : ((G(P_IDYCU, 0, 0) * (G(P_DY2H, 0, 0) * LX(s_xx, 0, 0) - G(P_DY2H, 1, 0) * LX(s_xx, 1, 0)) +
                                 G(P_IDXCU, 0, 0) * (G(P_DX2Q, 0, -1) * LX(s_xy, 0, -1) - G(P_DX2Q, 0, 0) * LX(s_xy, 0, 0))) *
                                G(P_IAREACU, 0, 0)) / (LX(s_hu, 0, 0) + h_neglect);
    if (i >= is && i <= ie && J >= Js && J <= je)
      *(double *)(dv_k + bv) = (obc && (A.obc_fv[bv >> 3] & 2)) ? 0. : ((G(P_IDYCV, 0, 0) * (G(P_DY2Q, -1, 0) * LX(s_xy, -1, 0) - G(P_DY2Q, 0, 0) * LX(s_xy, 0, 0)) -
                                 G(P_IDXCV, 0, 0) * (G(P_DX2H, 0, 0) * LX(s_xx, 0, 0) - G(P_DX2H, 0, 1) * LX(s_xx, 0, 1))) *
                                G(P_IAREACV, 0, 0)) / (LX(s_hv, 0, 0) + h_neglect);
  } END_POINTS
#undef LX
#undef LDB
#undef G
#undef AH
#undef AU
#undef AV
#undef AQ
#undef HU_AT
#undef FOR_POINTS
#undef END_POINTS
}

// MEKE%mom_src (:1783-1800, :1833-1889 with MEKE%backscatter_Ro_c = 0): the frictional work of each layer from its layer-integrated
// stresses, summed over the layers in order.  One lane per column.
__global__ __launch_bounds__(256) void hv_frictwork_kernel(m6::GridDev g, const double *__restrict__ u, const double *__restrict__ v,
                                                           const double *__restrict__ sxx, const double *__restrict__ sxy,
                                                           double *__restrict__ mom_src) {
  const int i = g.isc + blockIdx.x * 256 + threadIdx.x, j = g.jsc + blockIdx.y;
  if (i > g.iec) return;
  const int I = i, J = j;
  const long hpl = (long)g.nih * g.njh, upl = (long)(g.nih + 1) * g.njh, vpl = (long)g.nih * (g.njh + 1), qpl = (long)(g.nih + 1) * (g.njh + 1);
  const double H_to_RZ = g.H_to_Z * g.Rho0;
  const double IdxT = g.IdxT[g.h2(i, j)], IdyT = g.IdyT[g.h2(i, j)];
  const double IdyB00 = g.IdyBu[g.q2(I, J)], IdxB00 = g.IdxBu[g.q2(I, J)], IdyBmm = g.IdyBu[g.q2(I - 1, J - 1)], IdxBmm = g.IdxBu[g.q2(I - 1, J - 1)];
  const double IdyBm0 = g.IdyBu[g.q2(I - 1, J)], IdxBm0 = g.IdxBu[g.q2(I - 1, J)], IdyB0m = g.IdyBu[g.q2(I, J - 1)], IdxB0m = g.IdxBu[g.q2(I, J - 1)];
  double src = 0.;
  for (int k = 0; k < g.nk; k++) {
    const double *uk = u + upl * k, *vk = v + vpl * k, *xy = sxy + qpl * k;
    const double str_xx = sxx[g.h2(i, j) + hpl * k];
    const double FrictWork = H_to_RZ * (
            (str_xx * (uk[g.u2(I, j)] - uk[g.u2(I - 1, j)]) * IdxT
           - str_xx * (vk[g.v2(i, J)] - vk[g.v2(i, J - 1)]) * IdyT)
        + 0.25 * ((xy[g.q2(I, J)] *
                   ((uk[g.u2(I, j + 1)] - uk[g.u2(I, j)]) * IdyB00
                  + (vk[g.v2(i + 1, J)] - vk[g.v2(i, J)]) * IdxB00)
                 + xy[g.q2(I - 1, J - 1)] *
                   ((uk[g.u2(I - 1, j)] - uk[g.u2(I - 1, j - 1)]) * IdyBmm
                  + (vk[g.v2(i, J - 1)] - vk[g.v2(i - 1, J - 1)]) * IdxBmm))
                + (xy[g.q2(I - 1, J)] *
                   ((uk[g.u2(I - 1, j + 1)] - uk[g.u2(I - 1, j)]) * IdyBm0
                  + (vk[g.v2(i, J)] - vk[g.v2(i - 1, J)]) * IdxBm0)
                 + xy[g.q2(I, J - 1)] *
                   ((uk[g.u2(I, j)] - uk[g.u2(I, j - 1)]) * IdyB0m
                  + (vk[g.v2(i + 1, J - 1)] - vk[g.v2(i, J - 1)]) * IdxB0m))));
    src = src + FrictWork;
  }
  mom_src[g.h2(i, j)] = src;
}

int check_cs(const mom6hip_hor_visc_cs_t *cs, const char *who) {
  static const char *names[10] = {"LEITH_KH", "LEITH_AH", "USE_LEITHY", "MEKE backscatter (MEKE_BACKSCAT_RO_C) / RES_SCALE_MEKE_VISC", "USE_GME", "ANISOTROPIC_VISCOSITY", "RE_AH", "KH_SIN_LAT",
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
                             double *diffu, double *diffv, const double *hu_cont, const double *hv_cont, const HVObcDev *ob) {
  const m6::GridDev g = ctx->g;
  if (!(cs->Laplacian || cs->biharmonic)) return 0;      // :451
  const size_t plane = sizeof(double) * (size_t)(g.nih + 1) * (g.njh + 1);
  M6_REQUIRE(plane < (1ull << 31) && sizeof(double) * (size_t)(g.nih + 1) * (g.njh + 1) < (1ull << 31),
             "horizontal_viscosity: a layer is too large for 32-bit byte offsets");
  if (!ctx->hv_pack_ready) {      // the metric planes of this grid, once per context
    M6_REQUIRE(ctx->hv_pack.reserve(plane * HV_NPLANES) == 0, "horizontal_viscosity: out of device memory");
    hipLaunchKernelGGL(hv_pack_kernel, dim3((g.nih + 1 + 255) / 256, g.njh + 1), dim3(256), 0, ctx->stream, g, (double *)ctx->hv_pack.p);
    M6_HIP(hipGetLastError());
    ctx->hv_pack_ready = true;
  }
  HVFArgs A;
  const HVOpt o = opt_of(cs);
  const int fl[13] = {o.Laplacian, o.biharmonic, o.Smagorinsky_Kh, o.Smagorinsky_Ah, o.bound_Kh, o.better_bound_Kh, o.bound_Ah, o.better_bound_Ah,
                      o.bound_Coriolis, o.add_LES_viscosity, o.no_slip, o.use_land_mask, o.use_cont_thick};
  A.flags.bits = 0;
  for (int n = 0; n < 13; n++) A.flags.bits |= (fl[n] ? 1u : 0u) << n;
  A.Kh_bg_min = o.Kh_bg_min;
  double *const *src = &cs->Kh_bg_xx;
  double **dst = &A.s.Kh_bg_xx;
  for (int n = 0; n < 16; n++) dst[n] = src[n];
  A.pk = (const char *)ctx->hv_pack.p; A.plane_bytes = (unsigned)plane;
  A.isc = g.isc; A.iec = g.iec; A.jsc = g.jsc; A.jec = g.jec; A.isd = g.isd; A.jsd = g.jsd; A.nih = g.nih; A.njh = g.njh; A.nk = g.nk;
  A.h_neglect = g.H_subroundoff;
  A.u = u; A.v = v; A.h = h; A.hu_cont = hu_cont; A.hv_cont = hv_cont; A.diffu = diffu; A.diffv = diffv;
  A.Ku = cs->Laplacian ? cs->MEKE_Ku : nullptr; A.Au = cs->biharmonic ? cs->MEKE_Au : nullptr;
  A.str_xx_out = nullptr; A.str_xy_out = nullptr;
  A.obc_q = A.obc_fu = A.obc_fv = A.obc_hu = A.obc_hv = nullptr;
  A.obc_tang_u = A.obc_tang_v = nullptr;
  if (ob) { A.obc_q = ob->q; A.obc_fu = ob->fu; A.obc_fv = ob->fv; A.obc_hu = ob->hu; A.obc_hv = ob->hv; A.obc_tang_u = ob->tang_u; A.obc_tang_v = ob->tang_v; }
  const bool meke = A.Ku || A.Au || cs->MEKE_mom_src;
  if (cs->MEKE_mom_src) {
    const size_t nH3 = (size_t)g.nih * g.njh * g.nk, nQ3 = (size_t)(g.nih + 1) * (g.njh + 1) * g.nk;
    M6_REQUIRE(ctx->hv_str.reserve(sizeof(double) * (nH3 + nQ3)) == 0, "horizontal_viscosity: out of device memory for MEKE%%mom_src");
    A.str_xx_out = (double *)ctx->hv_str.p; A.str_xy_out = A.str_xx_out + nH3;
  }
  const int ni = g.iec - g.isc + 1, nj = g.jec - g.jsc + 1;
  auto launch = [&](auto kern, int TI, int TJ, int NT) {
    const int ntx = (ni + TI - 1) / TI, nty = (nj + TJ - 1) / TJ, ntiles = ntx * nty;
    const long nblocks = (long)((ntiles + 7) / 8) * 8 * g.nk;
    if (nblocks >= (1L << 31)) return 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(NT), 0, ctx->stream, A, ntx, ntiles);
    return 0;
  };
  // (tiles of 64x8, 64x12, 32x16, 128x8 points and blocks of 256 / 1024 threads all ran within 5% of this one: the kernel is
  // bound by the L2 -> L1 traffic of the metric planes, which every block reads for its tile -- profiles/r02_hor_visc.txt)
  int rc;
  if (A.flags.bits == HV_BIH_SMAG && !meke && !ob) rc = launch(hv_fused_kernel<HV_BIH_SMAG, 64, 16, 512, 4>, 64, 16, 512);
  else rc = launch(hv_fused_kernel<HV_GENERIC, 64, 16, 512, 4>, 64, 16, 512);
  M6_REQUIRE(rc == 0, "horizontal_viscosity: the grid is too large for one launch");
  if (cs->MEKE_mom_src)
    hipLaunchKernelGGL(hv_frictwork_kernel, dim3((ni + 255) / 256, nj), dim3(256), 0, ctx->stream, g, u, v, A.str_xx_out, A.str_xy_out,
                       cs->MEKE_mom_src);
  M6_HIP(hipGetLastError());
  return 0;
}
}  // namespace m6

extern "C" int mom6hip_horizontal_viscosity(mom6hip_ctx_t *ctx, const mom6hip_hor_visc_cs_t *cs, const double *u, const double *v,
                                            const double *h, double *diffu, double *diffv, double dt, const double *hu_cont,
                                            const double *hv_cont, int32_t memspace) {
  return mom6hip_horizontal_viscosity_obc(ctx, cs, u, v, h, diffu, diffv, dt, hu_cont, hv_cont, nullptr, memspace);
}

namespace {
// What the reference's loops over the segments leave (MOM_hor_visc.F90:733-849, :889-903, :1388-1409, :1751-1782), as maps over the q points
// and the faces: the loops are run here on the indices alone, in the reference's order and with its ranges.
// OBC_COMPUTED_STRAIN: a segment's tangential_vel (IsdB:IedB, JsdB:JedB, nk) at its corner points a0..a1 of the line `fixed`, into the q-shaped 3-D array
__global__ __launch_bounds__(64) void hv_tang_scatter_kernel(m6::GridDev g, double *dst, const double *src, int along_i, int a0, int a1, int fixed,
                                                             int IsdB, int JsdB, int nI, int nJ) {
  const int a = a0 + blockIdx.x * 64 + threadIdx.x, k = blockIdx.y;
  if (a > a1) return;
  const int I = along_i ? a : fixed, J = along_i ? fixed : a;
  dst[g.q2(I, J) + (long)(g.nih + 1) * (g.njh + 1) * k] = src[(I - IsdB) + (long)nI * ((J - JsdB) + (long)nJ * k)];
}

int hv_obc_maps(mom6hip_ctx_t *ctx, m6::Stager &st, const mom6hip_obc_t *obc, m6::HVObcDev &ob) {
  const m6::GridDev g = ctx->g;
  M6_REQUIRE(obc->number_of_segments == 0 || obc->segment, "horizontal_viscosity: OBC%%segment is required");
  const int is = g.isc, ie = g.iec, js = g.jsc, je = g.jec, Isq = is - 1, Ieq = ie, Jsq = js - 1, Jeq = je;
  const int is_vort = is - 2, ie_vort = Ieq + 1, js_vort = js - 2, je_vort = Jeq + 1;
  const size_t nU = (size_t)(g.nih + 1) * g.njh, nV = (size_t)g.nih * (g.njh + 1), nQ = (size_t)(g.nih + 1) * (g.njh + 1);
  double *tang_u = nullptr, *tang_v = nullptr;
  std::vector<int32_t> m(nQ + 2 * nU + 2 * nV, 0);
  int32_t *q = m.data(), *fu = q + nQ, *fv = fu + nU, *hu = fv + nV, *hv = hu + nU;
  for (size_t n = 0; n < nU; n++) hu[n] = -1;
  for (size_t n = 0; n < nV; n++) hv[n] = -1;
  auto imax = [](int a, int b) { return a > b ? a : b; };
  auto imin = [](int a, int b) { return a < b ? a : b; };
  const int nseg = obc->number_of_segments;
  for (int n = 0; n < nseg; n++) {
    const mom6hip_obc_segment_t &S = obc->segment[n];
    if (!S.on_pe) continue;      // (a segment off the PE carries no index ranges)
    M6_REQUIRE(S.IsdB >= g.isd - 1 && S.IedB <= g.ied && S.JsdB >= g.jsd - 1 && S.JedB <= g.jed && S.isd >= g.isd && S.ied <= g.ied &&
                   S.jsd >= g.jsd && S.jed <= g.jed, "horizontal_viscosity: OBC segment %d lies outside the data domain", n + 1);
    const int J = S.JsdB, I = S.IsdB;
    if (obc->zero_strain || obc->freeslip_strain) {      // :735-790
      if (S.is_N_or_S && J >= js_vort && J <= je_vort) {
        for (int Iq = imax(S.IsdB, is_vort); Iq <= imin(S.IedB, ie_vort); Iq++) q[g.q2(Iq, J)] |= obc->zero_strain ? 3 : 2;
      } else if (S.is_E_or_W && I >= is_vort && I <= ie_vort) {
        for (int Jq = imax(S.JsdB, js_vort); Jq <= imin(S.JedB, je_vort); Jq++) q[g.q2(I, Jq)] |= obc->zero_strain ? 3 : 1;
      }
    } else if (obc->computed_strain) {      // :741-748, :766-773: from the segment's tangential_vel (a later segment has the last word)
      M6_REQUIRE(S.tangential_vel, "horizontal_viscosity: OBC_COMPUTED_STRAIN needs the tangential_vel of segment %d", n + 1);
      if (!tang_u) {      // the tangential velocities at the q points, for the segments' scatter kernels
        tang_u = (double *)st.scratch(2 * nQ * g.nk * 8); tang_v = tang_u ? tang_u + nQ * g.nk : nullptr;
        M6_REQUIRE(!st.failed() && tang_u, "horizontal_viscosity: staging of the open boundaries failed");
      }
      const size_t cnt = (size_t)(S.IedB - S.IsdB + 1) * (S.JedB - S.JsdB + 1) * g.nk * 8;
      const double *tv = st.in(S.tangential_vel, cnt);
      M6_REQUIRE(!st.failed() && tv, "horizontal_viscosity: staging of the open boundaries failed");
      if (S.is_N_or_S && J >= js_vort && J <= je_vort) {
        const int a0 = imax(S.IsdB, is_vort), a1 = imin(S.IedB, ie_vort);
        for (int Iq = a0; Iq <= a1; Iq++) { int32_t &c = q[g.q2(Iq, J)]; c = (c & ~48) | (S.direction == MOM6HIP_OBC_DIRECTION_N ? 16 : 32); }
        if (a1 >= a0) hipLaunchKernelGGL(hv_tang_scatter_kernel, dim3((a1 - a0 + 64) / 64, g.nk), dim3(64), 0, ctx->stream, g, tang_u, tv, 1, a0, a1, J,
                                         S.IsdB, S.JsdB, S.IedB - S.IsdB + 1, S.JedB - S.JsdB + 1);
      } else if (S.is_E_or_W && I >= is_vort && I <= ie_vort) {
        const int a0 = imax(S.JsdB, js_vort), a1 = imin(S.JedB, je_vort);
        for (int Jq = a0; Jq <= a1; Jq++) { int32_t &c = q[g.q2(I, Jq)]; c = (c & ~192) | (S.direction == MOM6HIP_OBC_DIRECTION_E ? 64 : 128); }
        if (a1 >= a0) hipLaunchKernelGGL(hv_tang_scatter_kernel, dim3((a1 - a0 + 64) / 64, g.nk), dim3(64), 0, ctx->stream, g, tang_v, tv, 0, a0, a1, I,
                                         S.IsdB, S.JsdB, S.IedB - S.IsdB + 1, S.JedB - S.JsdB + 1);
      }
    }
    // :791-819: the thickness of the cell inside at the segment's faces (a later segment has the last word)
    if (S.direction == MOM6HIP_OBC_DIRECTION_N || S.direction == MOM6HIP_OBC_DIRECTION_S) {
      if (J >= js - 2 && J <= Jeq + 1)
        for (int i = imax(is - 2, S.isd); i <= imin(ie + 2, S.ied); i++) {
          int32_t &c = fv[g.v2(i, J)];
          c = (c & ~12) | ((S.direction == MOM6HIP_OBC_DIRECTION_N ? 1 : 2) << 2);
        }
    } else if (S.direction == MOM6HIP_OBC_DIRECTION_E || S.direction == MOM6HIP_OBC_DIRECTION_W) {
      if (I >= is - 2 && I <= Ieq + 1)
        for (int j = imax(js - 2, S.jsd); j <= imin(je + 2, S.jed); j++) {
          int32_t &c = fu[g.u2(I, j)];
          c = (c & ~12) | ((S.direction == MOM6HIP_OBC_DIRECTION_E ? 1 : 2) << 2);
        }
    }
  }
  // :821-849: then across the corner points, segment by segment: a face takes the thickness another face holds at that moment
  auto copy = [](int32_t *src, long dst, long from) { src[dst] = src[from] >= 0 ? src[from] : (int32_t)from; };
  for (int n = 0; n < nseg; n++) {
    const mom6hip_obc_segment_t &S = obc->segment[n];
    if (!S.on_pe) continue;
    const int J = S.JsdB, I = S.IsdB;
    if (S.direction == MOM6HIP_OBC_DIRECTION_N) {
      if (J >= js - 2 && J <= je) for (int Iq = imax(is - 2, S.IsdB); Iq <= imin(Ieq + 1, S.IedB); Iq++) copy(hu, g.u2(Iq, J + 1), g.u2(Iq, J));
    } else if (S.direction == MOM6HIP_OBC_DIRECTION_S) {
      if (J >= js - 1 && J <= je + 1) for (int Iq = imax(is - 2, S.isd); Iq <= imin(Ieq + 1, S.ied); Iq++) copy(hu, g.u2(Iq, J), g.u2(Iq, J + 1));
    } else if (S.direction == MOM6HIP_OBC_DIRECTION_E) {
      if (I >= is - 2 && I <= ie) for (int Jq = imax(js - 2, S.jsd); Jq <= imin(Jeq + 1, S.jed); Jq++) copy(hv, g.v2(I + 1, Jq), g.v2(I, Jq));
    } else if (S.direction == MOM6HIP_OBC_DIRECTION_W) {
      if (I >= is - 1 && I <= ie + 1) for (int Jq = imax(js - 2, S.jsd); Jq <= imin(Jeq + 1, S.jed); Jq++) copy(hv, g.v2(I, Jq), g.v2(I + 1, Jq));
    }
  }
  for (int n = 0; n < nseg; n++) {
    const mom6hip_obc_segment_t &S = obc->segment[n];
    if (!S.on_pe) continue;
    const int J = S.JsdB, I = S.IsdB;
    if (obc->zero_biharmonic) {      // :889-903
      if (S.is_N_or_S && J >= Jsq - 1 && J <= Jeq + 1) { for (int i = S.isd; i <= S.ied; i++) fv[g.v2(i, J)] |= 1; }
      else if (S.is_E_or_W && I >= Isq - 1 && I <= Ieq + 1) { for (int j = S.jsd; j <= S.jed; j++) fu[g.u2(I, j)] |= 1; }
    }
    if (obc->zero_strain || obc->freeslip_strain) {      // :1388-1409
      if (S.is_N_or_S && J >= js - 1 && J <= Jeq) { for (int Iq = S.IsdB; Iq <= S.IedB; Iq++) q[g.q2(Iq, J)] |= obc->zero_strain ? 12 : 8; }
      else if (S.is_E_or_W && I >= is - 1 && I <= Ieq) { for (int Jq = S.JsdB; Jq <= S.JedB; Jq++) q[g.q2(I, Jq)] |= obc->zero_strain ? 12 : 4; }
    }
    if (S.is_E_or_W) for (int j = S.jsd; j <= S.jed; j++) fu[g.u2(I, j)] |= 2;      // :1751-1762
    if (S.is_N_or_S) for (int i = S.isd; i <= S.ied; i++) fv[g.v2(i, J)] |= 2;      // :1771-1782
  }
  // (the maps change only with the OBC: uploaded when their content differs from the copy the context keeps, never waited for)
  const int32_t *dm = (const int32_t *)m6::obc_table_content(ctx, m6::OBC_SITE_HORVISC, m6::obc_fingerprint(ctx, obc), m.data(), 4 * m.size());
  if (!dm) return 1;
  ob.q = dm; ob.fu = dm + nQ; ob.fv = ob.fu + nU; ob.hu = ob.fv + nV; ob.hv = ob.hu + nU;
  ob.tang_u = tang_u; ob.tang_v = tang_v;
  M6_HIP(hipGetLastError());
  return 0;
}
}  // namespace

// horizontal_viscosity with OBC associated; obc == NULL or not OBC%OBC_pe: mom6hip_horizontal_viscosity
extern "C" int mom6hip_horizontal_viscosity_obc(mom6hip_ctx_t *ctx, const mom6hip_hor_visc_cs_t *cs, const double *u, const double *v,
                                                const double *h, double *diffu, double *diffv, double dt, const double *hu_cont,
                                                const double *hv_cont, const mom6hip_obc_t *obc, int32_t memspace) {
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
  dcs.MEKE_Ku = st.in(cs->MEKE_Ku, bH2); dcs.MEKE_Au = st.in(cs->MEKE_Au, bH2);
  dcs.MEKE_mom_src = cs->MEKE_mom_src ? st.inout(cs->MEKE_mom_src, bH2) : nullptr;
  M6_REQUIRE(!st.failed(), "horizontal_viscosity: staging failed");
  m6::HVObcDev ob;
  const bool apply_OBC = obc && obc->OBC_pe;      // :449-452
  if (apply_OBC && hv_obc_maps(ctx, st, obc, ob)) return 1;
  if (m6::horizontal_viscosity_dev(ctx, &dcs, du, dv, dh, ddu, ddv, dhu, dhv, apply_OBC ? &ob : nullptr)) return 1;
  return st.finish();
}
