// open_boundary.hip -- the part of src/core/MOM_open_boundary.F90 the RK2 step calls with the settings of .testing/tc3, on the device:
// radiation_open_bdry_conds :2196 for the normal component (Orlanski radiation with its restart fields rx_normal / ry_normal, the
// gradient condition, nudging), open_boundary_apply_normal_flow :3337, the pass of u_new, v_new, and open_boundary_zero_normal_flow
// :3374.  One launch a segment (thread per point along the segment and layer): a configuration has a handful of segments, a point of
// a segment reads the two faces inside the boundary and writes its own face only, and a later segment has the last word where two
// overlap, as in the reference's loop over the segments.  Provided as well (round 4): oblique radiation (rad_segment_kernel's oblique branch
// with the rates kept in OBC%rx_oblique_u ... cff_normal_v), the tangential forms ORLANSKI_ / NUDGED_ / OBLIQUE_TAN and _GRAD
// (rad_tangential_kernel: segment%tangential_vel, tangential_grad at the corner points) and update_segment_tracer_reservoirs
// (obc_reservoir_kernel).
#include <cmath>
#include <vector>

#include "common.hpp"

namespace {

using m6::min2;

struct RadSeg {
  int ew, d1, A, c0, c1, radiation, gradient, nudged;      // d1 = -1 (E, N: inside towards smaller indices) or +1
  int oblique, gq0, gq1;   // segment%oblique; the corner points gradient_at_q_points computes segment%grad_normal at (zero beyond)
  double *rn_st, *rt_st, *cf_st;      // OBC%rx_oblique_u | ry_oblique_v (the rate along the normal), ry_oblique_u | rx_oblique_v, cff_normal_u | _v
  int a0, cs0;             // the first index of the segment's own arrays along / across (IsdB | JsdB, jsd | isd)
  long nA, nc;             // their extents
  double tau_in, tau_out;
  double *normal_vel;
  const double *nudged_normal_vel;
};

__device__ __forceinline__ long rad_idx(const RadSeg &S, int c, int k) {      // (A - a0 = 0: a segment is one face wide)
  return S.ew ? (long)0 + S.nA * ((c - S.cs0) + S.nc * (long)k) : (long)(c - S.cs0) + S.nc * ((long)0 + S.nA * (long)k);
}

// :2326-2400 (E), :2571 (W), :2816 (N), :3060 (S): the normal velocity of the segment's faces
__global__ __launch_bounds__(64) void rad_segment_kernel(m6::GridDev g, RadSeg S, double gamma_u, double rx_max, double dt, const double *xn,
                                                         const double *xo, double *r_normal) {
  const int c = S.c0 + blockIdx.x * 64 + threadIdx.x, k = blockIdx.y;
  if (c > S.c1) return;
  const long pl = S.ew ? (long)(g.nih + 1) * g.njh : (long)g.nih * (g.njh + 1);
  const long f0 = (S.ew ? g.u2(S.A, c) : g.v2(c, S.A)) + pl * k;
  const long step = S.ew ? 1 : g.nih;      // one face along the direction
  const long f1 = f0 + S.d1 * step, f2 = f0 + 2 * S.d1 * step;
  double dhdt = 0.0, dhdx = 0.0;
  double nv = S.normal_vel[rad_idx(S, c, k)];
  if (S.oblique && !S.radiation) {      // :2349-2383 (E), :2593-2628 (W), :2838-2872 (N), :3082-3117 (S)
    // segment%grad_normal(q, 1 | 2, k) of gradient_at_q_points (:3407): the difference of the normal component along the boundary at the corner
    // point q, one face inside (a = A + d1) | on the boundary (a = A)
    const long cstep = S.ew ? (g.nih + 1) : 1;      // one face along the boundary
    auto GN = [&](int q, long fa, int a) -> double {      // fa: the face (a, q) of this layer
      if (q < S.gq0 || q > S.gq1) return 0.0;
      return (xn[fa + cstep] - xn[fa]) * g.mask2dBu[S.ew ? g.q2(a, q) : g.q2(q, a)];
    };
    dhdt = (xo[f1] - xn[f1]);
    dhdx = (xn[f1] - xn[f2]);
    const double g1c = GN(c, f1, S.A + S.d1), g1m = GN(c - 1, f1 - cstep, S.A + S.d1);
    double dhdy;
    if (dhdt * (g1c + g1m) > 0.0) dhdy = g1m;
    else if (dhdt * (g1c + g1m) == 0.0) dhdy = 0.0;
    else dhdy = g1c;
    if (dhdt * dhdx < 0.0) dhdt = 0.0;
    const double cff_new = m6::max2(dhdx * dhdx + dhdy * dhdy, 1.0e-20);
    const double rn_new = min2(dhdt * dhdx, cff_new * rx_max);
    const double rt_new = min2(cff_new, m6::max2(dhdt * dhdy, -cff_new));
    double rn_avg, rt_avg, cff_avg;
    if (gamma_u < 1.0) {
      rn_avg = (1.0 - gamma_u) * S.rn_st[f0] + gamma_u * rn_new;
      rt_avg = (1.0 - gamma_u) * S.rt_st[f0] + gamma_u * rt_new;
      cff_avg = (1.0 - gamma_u) * S.cf_st[f0] + gamma_u * cff_new;
    } else { rn_avg = rn_new; rt_avg = rt_new; cff_avg = cff_new; }
    nv = ((cff_avg * xn[f0] + rn_avg * xn[f1]) - (m6::max2(rt_avg, 0.0) * GN(c - 1, f0 - cstep, S.A) + min2(rt_avg, 0.0) * GN(c, f0, S.A))) / (cff_avg + rn_avg);
    if (gamma_u < 1.0) { S.rn_st[f0] = rn_avg; S.rt_st[f0] = rt_avg; S.cf_st[f0] = cff_avg; }
  } else if (S.radiation) {
    dhdt = (xo[f1] - xn[f1]);
    dhdx = (xn[f1] - xn[f2]);
    double rx_new = 0.0, rx_avg;
    if (dhdt * dhdx > 0.0) rx_new = min2((dhdt / dhdx), rx_max);
    if (gamma_u < 1.0) rx_avg = (1.0 - gamma_u) * r_normal[f0] + gamma_u * rx_new;
    else rx_avg = rx_new;
    nv = (xn[f0] + rx_avg * xn[f1]) / (1.0 + rx_avg);
    if (gamma_u < 1.0) r_normal[f0] = rx_avg;
  } else if (S.gradient) {
    nv = xn[f1];
  }
  if ((S.radiation || S.oblique) && S.nudged) {
    const double tau = (dhdt * dhdx <= 0.0) ? S.tau_in : S.tau_out;
    const double gamma_2 = dt / (tau + dt);
    nv = (1.0 - gamma_2) * nv + gamma_2 * S.nudged_normal_vel[rad_idx(S, c, k)];
  }
  S.normal_vel[rad_idx(S, c, k)] = nv;
}

// the tangential forms :2403-2455 (E), :2648-2700 (W), :2892-2945 (N), :3137-3190 (S): the rate at a corner point of the segment, then
// segment%tangential_vel (ORLANSKI_TAN, NUDGED_TAN) and segment%tangential_grad (ORLANSKI_GRAD, NUDGED_GRAD); a thread a corner and layer
struct TanArgs {
  int bits, q0, q1;            // MOM6HIP_OBC_TAN_* ; the corner points along the segment (JsdB:JedB | IsdB:IedB)
  long nq;                     // their number: the segment's arrays (IsdB:IedB, JsdB:JedB, nk) are one point wide
  double *tangential_vel, *tangential_grad;
  const double *nudged_vel, *nudged_grad;
  const double *rn_st, *rt_st, *cf_st;      // OBLIQUE_TAN / _GRAD: what the segment's faces keep (the rate along the normal, along the boundary, cff)
};
__global__ __launch_bounds__(64) void rad_tangential_kernel(m6::GridDev g, RadSeg S, TanArgs a, double gamma_u, double rx_max, double dt,
                                                            const double *tn, const double *to, const double *r_normal) {
  const int q = a.q0 + blockIdx.x * 64 + threadIdx.x, k = blockIdx.y;
  if (q > a.q1) return;
  const bool plus = S.d1 < 0;      // E, N: the cells inside lie towards smaller indices
  const int t0 = plus ? S.A : S.A + 1, st = plus ? -1 : 1;      // the first row of cells inside, and the step further in
  // the tangential component at the corner q in the row / column t of cells: v(t, Q) for E / W, u(Q, t) for N / S
  const long tpl = S.ew ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh;
  auto T3 = [&](int t) -> long { return (S.ew ? g.v2(t, q) : g.u2(q, t)) + tpl * k; };
  const long npl = S.ew ? (long)(g.nih + 1) * g.njh : (long)g.nih * (g.njh + 1);
  auto F3 = [&](int c) -> long { return (S.ew ? g.u2(S.A, c) : g.v2(c, S.A)) + npl * k; };
  const long s3 = (long)(q - a.q0) + a.nq * (long)k;
  const int lo1 = plus ? t0 - 1 : t0, lo2 = plus ? t0 - 2 : t0 + 1;      // the low rows of the two differences across the boundary
  const double *Idm = S.ew ? g.IdxBu : g.IdyBu;
  auto QM = [&](int t, int qq) -> long { return S.ew ? g.q2(t, qq) : g.q2(qq, t); };
  const int g0 = (S.ew ? g.jsd : g.isd) + 1, g1 = (S.ew ? g.jed : g.ied) - 1;
  if (a.bits & (MOM6HIP_OBC_TAN_OBLIQUE | MOM6HIP_OBC_GRAD_OBLIQUE)) {      // :2456-2556 (E) and its three twins
    auto TQ = [&](int t, int qq) -> double { return tn[(S.ew ? g.v2(t, qq) : g.u2(qq, t)) + tpl * k]; };
    const int cd0 = S.ew ? g.jsd : g.isd, cd1 = S.ew ? g.jed : g.ied;
    const int gt0 = (S.c0 - 1 > cd0) ? S.c0 - 1 : cd0, gt1 = (S.c1 + 1 < cd1) ? S.c1 + 1 : cd1;
    const int gg0 = (S.c0 > cd0 + 1) ? S.c0 : cd0 + 1, gg1 = (S.c1 < cd1 - 1) ? S.c1 : cd1 - 1;
    // segment%grad_tan(c, 1 | 2) (the rows t0 + st | t0) and segment%grad_gradient(c, 2) of gradient_at_q_points, zero where it does not compute them
    auto GT = [&](int c, int t) -> double {
      if (c < gt0 || c > gt1) return 0.0;
      return (TQ(t, c) - TQ(t, c - 1)) * g.mask2dT[S.ew ? g.h2(t, c) : g.h2(c, t)];
    };
    auto GG2 = [&](int c) -> double {
      if (c < gg0 || c > gg1) return 0.0;
      const double *maskC = S.ew ? g.mask2dCu : g.mask2dCv;
      return (((TQ(lo1 + 1, c) - TQ(lo1, c)) * Idm[QM(lo1, c)]) - (TQ(lo1 + 1, c - 1) - TQ(lo1, c - 1)) * Idm[QM(lo1, c - 1)]) *
             maskC[S.ew ? g.u2(lo1, c) : g.v2(c, lo1)];
    };
    double rn, rt, cff;
    if (gamma_u < 1.0) {
      if (q == a.q0)      { rn = a.rn_st[F3(S.c0)]; rt = a.rt_st[F3(S.c0)]; cff = a.cf_st[F3(S.c0)]; }
      else if (q == a.q1) { rn = a.rn_st[F3(S.c1)]; rt = a.rt_st[F3(S.c1)]; cff = a.cf_st[F3(S.c1)]; }
      else { rn = 0.5 * (a.rn_st[F3(q)] + a.rn_st[F3(q + 1)]); rt = 0.5 * (a.rt_st[F3(q)] + a.rt_st[F3(q + 1)]); cff = 0.5 * (a.cf_st[F3(q)] + a.cf_st[F3(q + 1)]); }
    } else {
      double dhdt = to[T3(t0)] - tn[T3(t0)];
      const double dhdn = tn[T3(t0)] - tn[T3(t0 + st)];
      const double ga = GT(q, t0 + st), gb = GT(q + 1, t0 + st);
      double dhdl;
      if (dhdt * (ga + gb) > 0.0) dhdl = ga;
      else if (dhdt * (ga + gb) == 0.0) dhdl = 0.0;
      else dhdl = gb;
      if (dhdt * dhdn < 0.0) dhdt = 0.0;
      cff = m6::max2(dhdn * dhdn + dhdl * dhdl, 1.0e-20);
      rn = min2(dhdt * dhdn, cff * rx_max);
      rt = min2(cff, m6::max2(dhdt * dhdl, -cff));
    }
    const double tau = (rn <= 0.0) ? S.tau_in : S.tau_out;
    const double gamma_2 = dt / (tau + dt);
    if (a.bits & MOM6HIP_OBC_TAN_OBLIQUE)
      a.tangential_vel[s3] = ((cff * tn[T3(t0)] + rn * tn[T3(t0 + st)]) - (m6::max2(rt, 0.0) * GT(q, t0) + min2(rt, 0.0) * GT(q + 1, t0))) / (cff + rn);
    if (a.bits & MOM6HIP_OBC_TAN_NUDGED) a.tangential_vel[s3] = (1.0 - gamma_2) * a.tangential_vel[s3] + gamma_2 * a.nudged_vel[s3];
    if ((a.bits & MOM6HIP_OBC_GRAD_OBLIQUE) && q >= a.q0 + 1 && q <= a.q1 - 1)
      a.tangential_grad[s3] = ((cff * (tn[T3(lo1 + 1)] - tn[T3(lo1)]) * Idm[QM(lo1, q)] + rn * (tn[T3(lo2 + 1)] - tn[T3(lo2)]) * Idm[QM(lo2, q)]) -
                               (m6::max2(rt, 0.0) * GG2(q) + min2(rt, 0.0) * GG2(q + 1))) / (cff + rn);
    if (a.bits & MOM6HIP_OBC_GRAD_NUDGED) a.tangential_grad[s3] = (1.0 - gamma_2) * a.tangential_grad[s3] + gamma_2 * a.nudged_grad[s3];
    return;
  }
  double r_tang;
  if (gamma_u < 1.0) {      // segment%rx_norm_rad at the two faces about the corner (the ends: the one face there is)
    if (q == a.q0) r_tang = r_normal[F3(S.c0)];
    else if (q == a.q1) r_tang = r_normal[F3(S.c1)];
    else r_tang = 0.5 * (r_normal[F3(q)] + r_normal[F3(q + 1)]);
  } else {
    const int ta = (!S.ew && plus) ? t0 + st : t0;      // (the northern segment looks one row further in than the other three, :2904-2905)
    const double dhdt = to[T3(ta)] - tn[T3(ta)];
    const double dhdx = tn[T3(ta)] - tn[T3(ta + st)];
    r_tang = 0.0;
    if (dhdt * dhdx > 0.0) r_tang = min2((dhdt / dhdx), rx_max);
  }
  const double tau = (r_tang <= 0.0) ? S.tau_in : S.tau_out;
  const double gamma_2 = dt / (tau + dt);
  if (a.bits & MOM6HIP_OBC_TAN_RADIATION) a.tangential_vel[s3] = (tn[T3(t0)] + r_tang * tn[T3(t0 + st)]) / (1.0 + r_tang);
  if (a.bits & MOM6HIP_OBC_TAN_NUDGED) a.tangential_vel[s3] = (1.0 - gamma_2) * a.tangential_vel[s3] + gamma_2 * a.nudged_vel[s3];
  if ((a.bits & MOM6HIP_OBC_GRAD_RADIATION) && q >= g0 && q <= g1)      // differences towards larger indices, with the metric of the corner point between the rows
    a.tangential_grad[s3] = ((tn[T3(lo1 + 1)] - tn[T3(lo1)]) * Idm[QM(lo1, q)] + r_tang * (tn[T3(lo2 + 1)] - tn[T3(lo2)]) * Idm[QM(lo2, q)]) / (1.0 + r_tang);
  if (a.bits & MOM6HIP_OBC_GRAD_NUDGED) a.tangential_grad[s3] = (1.0 - gamma_2) * a.tangential_grad[s3] + gamma_2 * a.nudged_grad[s3];
}

// open_boundary_apply_normal_flow :3337 (from = the segment's normal_vel) / open_boundary_zero_normal_flow :3374 (from = null)
__global__ __launch_bounds__(64) void obc_face_store_kernel(m6::GridDev g, RadSeg S, double *x, int from_normal_vel) {
  const int c = S.c0 + blockIdx.x * 64 + threadIdx.x, k = blockIdx.y;
  if (c > S.c1) return;
  const long pl = S.ew ? (long)(g.nih + 1) * g.njh : (long)g.nih * (g.njh + 1);
  x[(S.ew ? g.u2(S.A, c) : g.v2(c, S.A)) + pl * k] = from_normal_vel ? S.normal_vel[rad_idx(S, c, k)] : 0.;
}

// update_segment_tracer_reservoirs :5439-5456 / :5483-5496: one registered tracer of a segment, a thread a face and layer
struct ResArgs { double *tres; const double *t, *tr; double InvL_in, InvL_out, lfac_in, lfac_out; };
__global__ __launch_bounds__(64) void obc_reservoir_kernel(m6::GridDev g, RadSeg S, ResArgs a, const double *xr3, const double *h) {
  const int c = S.c0 + blockIdx.x * 64 + threadIdx.x, k = blockIdx.y;
  if (c > S.c1) return;
  // d1 = +1 (W, S): the nearest interior tracer cell is A + 1 and the flow into the reservoir is the negative one; -1 (E, N): A, positive
  const int shift = S.d1 > 0 ? 1 : 0;
  const double dir = S.d1 > 0 ? -1.0 : 1.0;
  const long cell2 = S.ew ? g.h2(S.A + shift, c) : g.h2(c, S.A + shift);
  if (g.mask2dT[cell2] == 0.0) return;
  const long cell3 = cell2 + (long)g.nih * g.njh * k;
  const long f2 = S.ew ? g.u2(S.A, c) : g.v2(c, S.A);
  const long pl = S.ew ? (long)(g.nih + 1) * g.njh : (long)g.nih * (g.njh + 1);
  const double len = S.ew ? g.dyCu[f2] : g.dxCv[f2];
  const double b_in = (a.InvL_in == 0.0) ? 1.0 : 0.0, b_out = (a.InvL_out == 0.0) ? 1.0 : 0.0;
  const double xr = dir * xr3[f2 + pl * k];
  const double a_out = b_out * m6::max2(0.0, copysign(1.0, xr));
  const double a_in  = b_in  * m6::min2(0.0, copysign(1.0, xr));
  const double L_out = m6::max2(0.0, xr * a.InvL_out * a.lfac_out / ((h[cell3] + g.H_subroundoff) * len));
  const double L_in  = m6::min2(0.0, xr * a.InvL_in * a.lfac_in / ((h[cell3] + g.H_subroundoff) * len));
  const double fac1 = (1.0 - (a_out - a_in)) + ((L_out + a_out) - (L_in + a_in));
  const long s3 = rad_idx(S, c, k);
  a.tres[s3] = (1.0 / fac1) * ((1.0 - a_out + a_in) * a.tres[s3] + ((L_out + a_out) * a.tr[cell3] - (L_in + a_in) * a.t[s3]));
}

// the geometry of a segment for these kernels; false: not on the PE or neither E/W nor N/S
bool rad_segment(const m6::GridDev &g, const mom6hip_obc_segment_t &S, RadSeg &d) {
  if (!S.on_pe) return false;
  const bool ew = S.is_E_or_W != 0;
  if (!ew && !S.is_N_or_S) return false;
  d.ew = ew ? 1 : 0;
  d.d1 = (S.direction == MOM6HIP_OBC_DIRECTION_E || S.direction == MOM6HIP_OBC_DIRECTION_N) ? -1 : 1;
  d.A = ew ? S.IsdB : S.JsdB; d.c0 = ew ? S.jsd : S.isd; d.c1 = ew ? S.jed : S.ied;
  d.a0 = d.A; d.cs0 = d.c0;
  d.nA = ew ? (S.IedB - S.IsdB + 1) : (S.JedB - S.JsdB + 1); d.nc = d.c1 - d.c0 + 1;
  d.radiation = S.radiation; d.gradient = S.gradient; d.nudged = S.nudged; d.oblique = S.oblique;
  {
    const int q0 = ew ? S.JsdB : S.IsdB, q1 = ew ? S.JedB : S.IedB, lo = ew ? g.jsd : g.isd, hi = (ew ? g.jed : g.ied) - 1;
    d.gq0 = q0 > lo ? q0 : lo; d.gq1 = q1 < hi ? q1 : hi;
  }
  d.rn_st = d.rt_st = d.cf_st = nullptr;
  d.tau_in = S.Velocity_nudging_timescale_in; d.tau_out = S.Velocity_nudging_timescale_out;
  d.normal_vel = nullptr; d.nudged_normal_vel = nullptr;
  return true;
}

int check_segment_range(const m6::GridDev &g, const mom6hip_obc_segment_t &S, int n, const char *who) {
  const bool ew = S.is_E_or_W != 0;
  M6_REQUIRE(ew ? (S.IsdB == S.IedB && S.IsdB >= g.isd + 1 && S.IsdB <= g.ied - 2 && S.jsd >= g.jsd && S.jed <= g.jed)
                : (S.JsdB == S.JedB && S.JsdB >= g.jsd + 1 && S.JsdB <= g.jed - 2 && S.isd >= g.isd && S.ied <= g.ied),
             "%s: OBC segment %d lies outside the data domain", who, n + 1);
  return 0;
}

}  // namespace

// Per face of the u and v grids, the cell an open-boundary face takes its cell-centred fields from (the zero-gradient
// projections of vertvisc_coef, set_viscous_BBL, ...): -1 the first cell (OBC_DIRECTION_E | N), +1 the second (W | S), 0 elsewhere.
// From OBC%segnum_u / segnum_v and the segments' directions; device arrays that live as long as `st`.  null maps: no segments.
uint64_t m6::obc_fingerprint(const mom6hip_ctx_t *ctx, const mom6hip_obc_t *obc) {
  uint64_t h = 0xcbf29ce484222325ull;
  if (!obc) return h;
  const m6::GridDev &g = ctx->g;
  const int32_t *flags = &obc->number_of_segments;      // the int32 members up to zero_biharmonic
  for (int q = 0; q < 16; q++) h = m6::obc_mix(h, (uint64_t)(uint32_t)flags[q]);
  h = m6::obc_mix(h, (uint64_t)g.nih); h = m6::obc_mix(h, (uint64_t)g.njh);
  for (int n = 0; n < obc->number_of_segments && obc->segment; n++) {
    const int32_t *sp = &obc->segment[n].direction;      // direction ... Flather: 20 int32 members
    for (int q = 0; q < 20; q++) h = m6::obc_mix(h, (uint64_t)(uint32_t)sp[q]);
  }
  h = m6::obc_mix(h, (uint64_t)(uintptr_t)obc->segnum_u); h = m6::obc_mix(h, (uint64_t)(uintptr_t)obc->segnum_v);
  // a sample of the two maps (every 61st value): an array reused at the same address for another configuration gives another key
  const size_t nU2 = (size_t)(g.nih + 1) * g.njh, nV2 = (size_t)g.nih * (g.njh + 1);
  if (obc->segnum_u) for (size_t q = 0; q < nU2; q += 61) h = m6::obc_mix(h, (uint64_t)(uint32_t)obc->segnum_u[q]);
  if (obc->segnum_v) for (size_t q = 0; q < nV2; q += 61) h = m6::obc_mix(h, (uint64_t)(uint32_t)obc->segnum_v[q]);
  return h;
}

const void *m6::obc_table(mom6hip_ctx_t *ctx, int site, uint64_t key, size_t bytes, const std::function<int(void *)> &build) {
  ctx->obc_table_clock++;
  for (auto &e : ctx->obc_tables)
    if (e.site == site && e.key == key && e.bytes == bytes) { e.last_use = ctx->obc_table_clock; return e.buf.p; }
  std::vector<char> host(bytes ? bytes : 1, 0);
  if (build(host.data())) return nullptr;
  if (ctx->obc_tables.size() >= 64) {      // drop the entry that has not been used for the longest time
    size_t old = 0;
    for (size_t q = 1; q < ctx->obc_tables.size(); q++) if (ctx->obc_tables[q].last_use < ctx->obc_tables[old].last_use) old = q;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { m6::set_error("obc_table: hipStreamSynchronize failed"); return nullptr; }
    ctx->obc_tables[old].buf.release();
    ctx->obc_tables.erase(ctx->obc_tables.begin() + (long)old);
  }
  mom6hip_ctx::ObcTable e;
  e.site = site; e.key = key; e.bytes = bytes; e.last_use = ctx->obc_table_clock;
  if (e.buf.reserve(bytes ? bytes : 8) != 0 || !e.buf.p) { m6::set_error("obc_table: out of device memory"); return nullptr; }
  if (hipMemcpyAsync(e.buf.p, host.data(), bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) {      // (the host vector goes out of scope; once per OBC and site)
    m6::set_error("obc_table: the upload of an open-boundary table failed");
    e.buf.release();
    return nullptr;
  }
  ctx->obc_table_builds++;
  ctx->obc_tables.push_back(e);
  return ctx->obc_tables.back().buf.p;
}

const void *m6::obc_table_content(mom6hip_ctx_t *ctx, int site, uint64_t key, const void *host, size_t bytes) {
  ctx->obc_table_clock++;
  mom6hip_ctx::ObcTable *e = nullptr;
  for (auto &q : ctx->obc_tables) if (q.site == site && q.key == key) { e = &q; break; }
  if (e && e->bytes == bytes && e->host.size() == bytes && memcmp(e->host.data(), host, bytes) == 0) { e->last_use = ctx->obc_table_clock; return e->buf.p; }
  if (!e) {
    mom6hip_ctx::ObcTable n;
    n.site = site; n.key = key; n.bytes = 0; n.last_use = 0;
    ctx->obc_tables.push_back(n);
    e = &ctx->obc_tables.back();
  }
  if (e->buf.reserve(bytes ? bytes : 8) != 0 || !e->buf.p) { m6::set_error("obc_table: out of device memory"); return nullptr; }
  // (the entry's former contents are still being read by host-to-device copies only if one is in flight from them: wait for it)
  if (!e->host.empty() && hipStreamSynchronize(ctx->stream) != hipSuccess) { m6::set_error("obc_table: hipStreamSynchronize failed"); return nullptr; }
  e->host.assign((const char *)host, (const char *)host + bytes);
  e->bytes = bytes; e->last_use = ctx->obc_table_clock;
  if (hipMemcpyAsync(e->buf.p, e->host.data(), bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
    m6::set_error("obc_table: the upload of an open-boundary table failed");
    return nullptr;
  }
  ctx->obc_table_builds++;
  return e->buf.p;
}

int m6::obc_side_maps(mom6hip_ctx_t *ctx, m6::Stager &st, const mom6hip_obc_t *obc, const int32_t **side_u, const int32_t **side_v,
                      const char *who) {
  (void)st;
  *side_u = *side_v = nullptr;
  if (!obc || obc->number_of_segments <= 0) return 0;
  M6_REQUIRE(obc->segment && obc->segnum_u && obc->segnum_v, "%s: OBC%%segment, segnum_u and segnum_v are required", who);
  const m6::GridDev g = ctx->g;
  const size_t nU2 = (size_t)(g.nih + 1) * g.njh, nV2 = (size_t)g.nih * (g.njh + 1);
  const void *dm = m6::obc_table(ctx, m6::OBC_SITE_SIDE, m6::obc_fingerprint(ctx, obc), 4 * (nU2 + nV2), [&](void *host) -> int {
    int32_t *m = (int32_t *)host;
    for (int d = 0; d < 2; d++) {
      const int32_t *segnum = d ? obc->segnum_v : obc->segnum_u;
      int32_t *o = m + (d ? nU2 : 0);
      for (size_t n = 0; n < (d ? nV2 : nU2); n++) {
        const int l = segnum[n];
        if (l == MOM6HIP_OBC_NONE) continue;
        M6_REQUIRE(l >= 1 && l <= obc->number_of_segments, "%s: OBC%%segnum_%c holds %d, with %d segments", who, d ? 'v' : 'u', l, obc->number_of_segments);
        const int dir = obc->segment[l - 1].direction;
        if (dir == (d ? MOM6HIP_OBC_DIRECTION_N : MOM6HIP_OBC_DIRECTION_E)) o[n] = -1;
        else if (dir == (d ? MOM6HIP_OBC_DIRECTION_S : MOM6HIP_OBC_DIRECTION_W)) o[n] = 1;
      }
    }
    return 0;
  });
  if (!dm) return 1;
  *side_u = (const int32_t *)dm; *side_v = (const int32_t *)dm + nU2;
  return 0;
}

// u, v (device) = the normal_vel of the specified segments on their faces: the loop that ends vertvisc (MOM_vert_friction.F90:988-1006)
int m6::obc_store_specified(mom6hip_ctx_t *ctx, m6::Stager &st, const mom6hip_obc_t *obc, double *d_u, double *d_v, const char *who) {
  if (!obc) return 0;
  const m6::GridDev g = ctx->g;
  for (int n = 0; n < obc->number_of_segments; n++) {
    const mom6hip_obc_segment_t &S = obc->segment[n];
    RadSeg d;
    if (!S.specified || !rad_segment(g, S, d)) continue;
    if (check_segment_range(g, S, n, who)) return 1;
    M6_REQUIRE(S.normal_vel, "%s: segment %d is specified: normal_vel is required", who, n + 1);
    d.normal_vel = (double *)st.in(S.normal_vel, (size_t)d.nA * d.nc * g.nk * 8);
    M6_REQUIRE(!st.failed() && d.normal_vel, "%s: staging failed", who);
    hipLaunchKernelGGL(obc_face_store_kernel, dim3((d.nc + 63) / 64, g.nk), dim3(64), 0, ctx->stream, g, d, d.ew ? d_u : d_v, 1);
  }
  M6_HIP(hipGetLastError());
  return 0;
}

extern "C" int mom6hip_radiation_open_bdry_conds(mom6hip_ctx_t *ctx, const mom6hip_obc_t *obc, double gamma_uv, double rx_max, double *rx_normal,
                                                 double *ry_normal, double *u_new, const double *u_old, double *v_new, const double *v_old,
                                                 double dt, int32_t memspace) {
  M6_REQUIRE(ctx != nullptr, "radiation_open_bdry_conds: the context is required");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "radiation_open_bdry_conds: bad memspace");
  if (!obc) return 0;      // .not.associated(OBC) :2240
  if (!(obc->open_u_BCs_exist_globally || obc->open_v_BCs_exist_globally)) return 0;      // :2242
  M6_REQUIRE(u_new && u_old && v_new && v_old, "radiation_open_bdry_conds: null argument");
  M6_REQUIRE(obc->number_of_segments == 0 || obc->segment, "radiation_open_bdry_conds: OBC%%segment is required");
  const m6::GridDev g = ctx->g;
  hipStream_t s = ctx->stream;
  const size_t bU = (size_t)g.nu3() * 8, bV = (size_t)g.nv3() * 8;
  m6::Stager st(ctx, memspace);
  double *d_un = st.inout(u_new, bU), *d_vn = st.inout(v_new, bV);
  const double *d_uo = st.in(u_old, bU), *d_vo = st.in(v_old, bV);
  double *d_rx = st.inout(rx_normal, bU), *d_ry = st.inout(ry_normal, bV);
  std::vector<RadSeg> segs;
  std::vector<TanArgs> tans;
  double *d_ob[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  for (int n = 0; n < obc->number_of_segments; n++) {
    const mom6hip_obc_segment_t &S = obc->segment[n];
    RadSeg d;
    if (!rad_segment(g, S, d)) continue;
    M6_REQUIRE(!(S.oblique && S.radiation), "radiation_open_bdry_conds: segment %d: Orlanski and Oblique OBC options cannot be used together", n + 1);
    if (check_segment_range(g, S, n, "radiation_open_bdry_conds")) return 1;
    if (S.oblique && gamma_uv < 1.0) {      // what the oblique segments keep between steps, staged once for all of them
      if (!d_ob[0]) {
        double *src[6] = {obc->rx_oblique_u, obc->ry_oblique_u, obc->cff_normal_u, obc->rx_oblique_v, obc->ry_oblique_v, obc->cff_normal_v};
        for (int m = 0; m < 6; m++) {
          M6_REQUIRE(src[m], "radiation_open_bdry_conds: OBLIQUE with OBC_RAD_VEL_WT < 1 needs OBC%%rx_oblique_u, ry_oblique_u, cff_normal_u and their _v twins");
          d_ob[m] = st.inout(src[m], m < 3 ? bU : bV);
        }
      }
      d.rn_st = d.ew ? d_ob[0] : d_ob[4]; d.rt_st = d.ew ? d_ob[1] : d_ob[3]; d.cf_st = d.ew ? d_ob[2] : d_ob[5];
    }
    if (S.radiation || S.gradient || S.oblique) {
      M6_REQUIRE(S.normal_vel, "radiation_open_bdry_conds: segment %d needs normal_vel", n + 1);
      M6_REQUIRE(!(S.radiation && gamma_uv < 1.0) || (d.ew ? rx_normal : ry_normal), "radiation_open_bdry_conds: OBC_RAD_VEL_WT < 1 needs OBC%%rx_normal / ry_normal");
      M6_REQUIRE(!((S.radiation || S.oblique) && S.nudged) || S.nudged_normal_vel, "radiation_open_bdry_conds: segment %d is nudged: nudged_normal_vel is required", n + 1);
      const size_t cnt = (size_t)d.nA * d.nc * g.nk * 8;
      d.normal_vel = st.inout(S.normal_vel, cnt);
      d.nudged_normal_vel = ((S.radiation || S.oblique) && S.nudged) ? st.in(S.nudged_normal_vel, cnt) : nullptr;
    }
    TanArgs a;
    a.bits = S.radiation_tan_or_grad; a.q0 = d.ew ? S.JsdB : S.IsdB; a.q1 = d.ew ? S.JedB : S.IedB; a.nq = a.q1 - a.q0 + 1;
    a.tangential_vel = a.tangential_grad = nullptr; a.nudged_vel = a.nudged_grad = nullptr;
    a.rn_st = a.rt_st = a.cf_st = nullptr;
    // (the blocks of the tangential forms :2403 and, oblique, :2456)
    if (!(a.bits & (MOM6HIP_OBC_TAN_RADIATION | MOM6HIP_OBC_GRAD_RADIATION | MOM6HIP_OBC_TAN_OBLIQUE | MOM6HIP_OBC_GRAD_OBLIQUE))) a.bits = 0;
    if (a.bits & (MOM6HIP_OBC_TAN_OBLIQUE | MOM6HIP_OBC_GRAD_OBLIQUE)) {
      M6_REQUIRE(!(a.bits & (MOM6HIP_OBC_TAN_RADIATION | MOM6HIP_OBC_GRAD_RADIATION)), "radiation_open_bdry_conds: segment %d: Orlanski and Oblique OBC "
                 "options cannot be used together", n + 1);
      M6_REQUIRE(gamma_uv >= 1.0 || d.rn_st, "radiation_open_bdry_conds: segment %d: OBLIQUE_TAN / OBLIQUE_GRAD with OBC_RAD_VEL_WT < 1 belong to an OBLIQUE "
                 "segment (its stored rates)", n + 1);
      a.rn_st = d.rn_st; a.rt_st = d.rt_st; a.cf_st = d.cf_st;
    }
    if (a.bits) {
      const int plus = d.d1 < 0;
      M6_REQUIRE(a.q0 >= (d.ew ? g.jsd : g.isd) - 1 && a.q1 <= (d.ew ? g.jed : g.ied) && a.q1 >= a.q0,
                 "radiation_open_bdry_conds: the corner points of OBC segment %d lie outside the data domain", n + 1);
      M6_REQUIRE(plus ? (d.A - 2 >= (d.ew ? g.isd : g.jsd)) : (d.A + 3 <= (d.ew ? g.ied : g.jed)),
                 "radiation_open_bdry_conds: OBC segment %d: the tangential forms need three rows of cells inside the segment in the data domain", n + 1);
      M6_REQUIRE(gamma_uv >= 1.0 || !(a.bits & (MOM6HIP_OBC_TAN_RADIATION | MOM6HIP_OBC_GRAD_RADIATION)) || (d.ew ? rx_normal : ry_normal),
                 "radiation_open_bdry_conds: OBC_RAD_VEL_WT < 1 needs OBC%%rx_normal / ry_normal");
      const size_t cnt = (size_t)a.nq * g.nk * 8;
      if (a.bits & (MOM6HIP_OBC_TAN_RADIATION | MOM6HIP_OBC_TAN_NUDGED | MOM6HIP_OBC_TAN_OBLIQUE)) {
        M6_REQUIRE(S.tangential_vel, "radiation_open_bdry_conds: segment %d: tangential_vel is required", n + 1);
        a.tangential_vel = st.inout(S.tangential_vel, cnt);
      }
      if (a.bits & (MOM6HIP_OBC_GRAD_RADIATION | MOM6HIP_OBC_GRAD_NUDGED | MOM6HIP_OBC_GRAD_OBLIQUE)) {
        M6_REQUIRE(S.tangential_grad, "radiation_open_bdry_conds: segment %d: tangential_grad is required", n + 1);
        a.tangential_grad = st.inout(S.tangential_grad, cnt);
      }
      if (a.bits & MOM6HIP_OBC_TAN_NUDGED) {
        M6_REQUIRE(S.nudged_tangential_vel, "radiation_open_bdry_conds: segment %d: nudged_tangential_vel is required", n + 1);
        a.nudged_vel = st.in(S.nudged_tangential_vel, cnt);
      }
      if (a.bits & MOM6HIP_OBC_GRAD_NUDGED) {
        M6_REQUIRE(S.nudged_tangential_grad, "radiation_open_bdry_conds: segment %d: nudged_tangential_grad is required", n + 1);
        a.nudged_grad = st.in(S.nudged_tangential_grad, cnt);
      }
    }
    segs.push_back(d); tans.push_back(a);
  }
  M6_REQUIRE(!st.failed(), "radiation_open_bdry_conds: staging failed");
  for (size_t m = 0; m < segs.size(); m++) {
    const RadSeg &d = segs[m];
    // I < IscB (E), I > IecB (W), J < JscB (N), J > JecB (S): the segment is skipped :2329, :2573, :2818, :3062
    const int lo = d.ew ? g.isc - 1 : g.jsc - 1, hi = d.ew ? g.iec : g.jec;
    if (d.d1 < 0 ? (d.A < lo) : (d.A > hi)) continue;
    if (d.radiation || d.gradient || d.oblique)
      hipLaunchKernelGGL(rad_segment_kernel, dim3((d.nc + 63) / 64, g.nk), dim3(64), 0, s, g, d, gamma_uv, rx_max, dt, d.ew ? d_un : d_vn,
                         d.ew ? d_uo : d_vo, d.ew ? d_rx : d_ry);
    if (tans[m].bits)      // (after the rates of the segment's own faces; the tangential component of u_new, v_new is not changed by this routine)
      hipLaunchKernelGGL(rad_tangential_kernel, dim3((tans[m].nq + 63) / 64, g.nk), dim3(64), 0, s, g, d, tans[m], gamma_uv, rx_max, dt,
                         d.ew ? d_vn : d_un, d.ew ? d_vo : d_uo, d.ew ? d_rx : d_ry);
  }
  for (const RadSeg &d : segs) {      // open_boundary_apply_normal_flow :3337 (radiation, oblique or gradient segments)
    if (!(d.radiation || d.gradient || d.oblique)) continue;
    hipLaunchKernelGGL(obc_face_store_kernel, dim3((d.nc + 63) / 64, g.nk), dim3(64), 0, s, g, d, d.ew ? d_un : d_vn, 1);
  }
  M6_HIP(hipGetLastError());
  double *pf[2] = {d_un, d_vn};
  const int32_t ppos[2] = {MOM6HIP_POS_U, MOM6HIP_POS_V}, pnk[2] = {g.nk, g.nk};
  if (int rc = m6::group_pass(ctx, pf, ppos, pnk, 2)) return rc;      // pass_vector(u_new, v_new) :3309
  return st.finish();
}

extern "C" int mom6hip_open_boundary_zero_normal_flow(mom6hip_ctx_t *ctx, const mom6hip_obc_t *obc, double *u, double *v, int32_t memspace) {
  M6_REQUIRE(ctx != nullptr, "open_boundary_zero_normal_flow: the context is required");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "open_boundary_zero_normal_flow: bad memspace");
  if (!obc) return 0;
  M6_REQUIRE(u && v, "open_boundary_zero_normal_flow: null argument");
  M6_REQUIRE(obc->number_of_segments == 0 || obc->segment, "open_boundary_zero_normal_flow: OBC%%segment is required");
  const m6::GridDev g = ctx->g;
  m6::Stager st(ctx, memspace);
  double *d_u = st.inout(u, (size_t)g.nu3() * 8), *d_v = st.inout(v, (size_t)g.nv3() * 8);
  M6_REQUIRE(!st.failed(), "open_boundary_zero_normal_flow: staging failed");
  for (int n = 0; n < obc->number_of_segments; n++) {
    RadSeg d;
    if (!rad_segment(g, obc->segment[n], d)) continue;
    if (check_segment_range(g, obc->segment[n], n, "open_boundary_zero_normal_flow")) return 1;
    hipLaunchKernelGGL(obc_face_store_kernel, dim3((d.nc + 63) / 64, g.nk), dim3(64), 0, ctx->stream, g, d, d.ew ? d_u : d_v, 0);
  }
  M6_HIP(hipGetLastError());
  return st.finish();
}

// update_segment_tracer_reservoirs(G, GV, uhr, vhr, h, OBC, dt, Reg) :5373
extern "C" int mom6hip_update_segment_tracer_reservoirs(mom6hip_ctx_t *ctx, const double *uhr, const double *vhr, const double *h,
                                                        const mom6hip_obc_t *obc, double dt, const double *const *tr, int32_t ntr,
                                                        int32_t memspace) {
  (void)dt;
  M6_REQUIRE(ctx != nullptr, "update_segment_tracer_reservoirs: the context is required");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "update_segment_tracer_reservoirs: bad memspace");
  if (!obc || !obc->OBC_pe) return 0;      // :5410
  M6_REQUIRE(uhr && vhr && h && (tr || ntr == 0), "update_segment_tracer_reservoirs: null argument");
  M6_REQUIRE(obc->number_of_segments == 0 || obc->segment, "update_segment_tracer_reservoirs: OBC%%segment is required");
  const m6::GridDev g = ctx->g;
  m6::Stager st(ctx, memspace);
  const double *d_uhr = nullptr, *d_vhr = nullptr, *d_h = nullptr;
  std::vector<const double *> d_tr(ntr > 0 ? ntr : 0, nullptr);
  for (int n = 0; n < obc->number_of_segments; n++) {
    const mom6hip_obc_segment_t &S = obc->segment[n];
    if (!S.tr_Reg) continue;
    RadSeg d;
    if (!rad_segment(g, S, d)) continue;
    if (check_segment_range(g, S, n, "update_segment_tracer_reservoirs")) return 1;
    M6_REQUIRE(d.A + (d.d1 > 0 ? 1 : 0) >= (d.ew ? g.isd : g.jsd) && d.A + (d.d1 > 0 ? 1 : 0) <= (d.ew ? g.ied : g.jed),
               "update_segment_tracer_reservoirs: segment %d has no cell inside it in the data domain", n + 1);
    if (!d_h) {      // (staged at the first segment that needs them)
      d_uhr = st.in(uhr, (size_t)g.nu3() * 8); d_vhr = st.in(vhr, (size_t)g.nv3() * 8); d_h = st.in(h, (size_t)g.nh3() * 8);
    }
    for (int q = 0; q < S.ntseg; q++) {
      const mom6hip_obc_segment_tracer_t &T = S.tr_Reg[q];
      if (!T.tres) continue;      // .not.allocated(tres) :5439
      M6_REQUIRE(T.ntr_index >= 1 && T.ntr_index <= ntr, "update_segment_tracer_reservoirs: the registry of OBC segment %d names tracer %d of %d",
                 n + 1, T.ntr_index, ntr);
      M6_REQUIRE(T.t, "update_segment_tracer_reservoirs: segment %d: the external values tr_Reg%%Tr(%d)%%t are required", n + 1, q + 1);
      if (!d_tr[T.ntr_index - 1]) d_tr[T.ntr_index - 1] = st.in(tr[T.ntr_index - 1], (size_t)g.nh3() * 8);
      const size_t cnt = (size_t)d.nA * d.nc * g.nk * 8;
      ResArgs a;
      a.tres = st.inout((double *)T.tres, cnt); a.t = st.in(T.t, cnt); a.tr = d_tr[T.ntr_index - 1];
      a.InvL_in = S.Tr_InvLscale_in; a.InvL_out = S.Tr_InvLscale_out; a.lfac_in = T.resrv_lfac_in; a.lfac_out = T.resrv_lfac_out;
      M6_REQUIRE(!st.failed() && a.tres && a.t && a.tr && d_h, "update_segment_tracer_reservoirs: staging failed");
      hipLaunchKernelGGL(obc_reservoir_kernel, dim3((d.nc + 63) / 64, g.nk), dim3(64), 0, ctx->stream, g, d, a, d.ew ? d_uhr : d_vhr, d_h);
    }
  }
  M6_HIP(hipGetLastError());
  return st.finish();
}
