/* oracle/open_boundary.c -- TEST INFRASTRUCTURE: a C restatement of the part of src/core/MOM_open_boundary.F90 the hot path calls with
 * the settings of .testing/tc3: radiation_open_bdry_conds :2196-3336 for the normal component (Orlanski radiation, the gradient condition,
 * nudging), open_boundary_apply_normal_flow :3337-3370, open_boundary_zero_normal_flow :3374-3403.  The reference writes the four
 * directions out separately (E :2326, W :2571, N :2816, S :3060); they differ in the direction of "inside" only and are restated once.
 * Restated as well (round 4): oblique radiation (:2349-2383 and its three twins, with gradient_at_q_points :3407 evaluated in place and
 * the restart fields rx_oblique_u ... cff_normal_v), the tangential forms ORLANSKI_TAN / _GRAD, NUDGED_TAN / _GRAD, OBLIQUE_TAN / _GRAD
 * (:2403-2556 E and twins: segment%tangential_vel, tangential_grad), and update_segment_tracer_reservoirs :5373-5502.
 * PARITY UNPINNED: the reference holds no known-answer vectors for these routines. */
#include <math.h>
#include <stdlib.h>
#include "mom6_oracle.h"

static inline double min2(double a, double b) { return a < b ? a : b; }
static inline double max2(double a, double b) { return a > b ? a : b; }
static inline int max2i(int a, int b) { return a > b ? a : b; }
static inline int min2i(int a, int b) { return a < b ? a : b; }

/* segment arrays on the segment's own index ranges (IsdB:IedB, jsd:jed, nk) | (isd:ied, JsdB:JedB, nk); face (A along, c across), k 1-based */
static inline long seg_idx(const mom6hip_obc_segment_t *S, int ew, int A, int c, int k) {
  if (!ew) { const long ni = S->ied - S->isd + 1, nJ = S->JedB - S->JsdB + 1; return (c - S->isd) + ni*((A - S->JsdB) + nJ*(long)(k-1)); }
  const long nI = S->IedB - S->IsdB + 1, nj = S->jed - S->jsd + 1;
  return (A - S->IsdB) + nI*((c - S->jsd) + nj*(long)(k-1));
}

int orc_radiation_open_bdry_conds(const mom6hip_grid_t *G, const mom6hip_obc_t *OBC, double gamma_uv, double rx_max, double *rx_normal,
                                  double *ry_normal, double *u_new, const double *u_old, double *v_new, const double *v_old, double dt)
{
  if (!OBC) return 0;
  if (!(OBC->open_u_BCs_exist_globally || OBC->open_v_BCs_exist_globally)) return 0;      /* :2242 */
  const int nz = G->nk;
  const double gamma_u = gamma_uv;
  for (int n = 0; n < OBC->number_of_segments; n++) {
    const mom6hip_obc_segment_t *S = &OBC->segment[n];
    if (!S->on_pe) continue;
    if ((S->radiation || S->gradient || S->oblique) && !S->normal_vel) return 3;
    if (S->oblique && gamma_u < 1.0 && !(S->is_E_or_W ? (OBC->rx_oblique_u && OBC->ry_oblique_u && OBC->cff_normal_u)
                                                       : (OBC->rx_oblique_v && OBC->ry_oblique_v && OBC->cff_normal_v))) return 3;
    if ((S->oblique && S->nudged) && !S->nudged_normal_vel) return 3;
    if (S->radiation && gamma_u < 1.0 && !(S->is_E_or_W ? rx_normal : ry_normal)) return 3;
    if ((S->radiation && S->nudged) && !S->nudged_normal_vel) return 3;
    const int ew = S->direction == MOM6HIP_OBC_DIRECTION_E || S->direction == MOM6HIP_OBC_DIRECTION_W;
    const int plus = S->direction == MOM6HIP_OBC_DIRECTION_E || S->direction == MOM6HIP_OBC_DIRECTION_N;      /* inside = towards smaller indices */
    const int A = ew ? S->IsdB : S->JsdB, c0 = ew ? S->jsd : S->isd, c1 = ew ? S->jed : S->ied;
    if (ew) { if (plus ? (A < G->isc-1) : (A > G->iec)) continue; }      /* I < IscB, I > IecB :2329, :2573 */
    else    { if (plus ? (A < G->jsc-1) : (A > G->jec)) continue; }
    const int d1 = plus ? -1 : 1, d2 = plus ? -2 : 2;                   /* the two faces inside the boundary */
    double *xn = ew ? u_new : v_new, *r_normal = ew ? rx_normal : ry_normal;
    const double *xo = ew ? u_old : v_old;
#define F3(a,c,k) (ew ? ORC_U3(G,a,c,k) : ORC_V3(G,c,a,k))
    for (int k = 1; k <= nz; k++) for (int c = c0; c <= c1; c++) {
      double dhdt = 0.0, dhdx = 0.0;
      double *nv = &S->normal_vel[seg_idx(S, ew, A, c, k)];
      if (S->radiation) {
        dhdt = (xo[F3(A+d1,c,k)] - xn[F3(A+d1,c,k)]);
        dhdx = (xn[F3(A+d1,c,k)] - xn[F3(A+d2,c,k)]);
        double rx_new = 0.0, rx_avg;
        if (dhdt*dhdx > 0.0) rx_new = min2( (dhdt/dhdx), rx_max);
        if (gamma_u < 1.0) rx_avg = (1.0-gamma_u)*r_normal[F3(A,c,k)] + gamma_u*rx_new;      /* segment%rx_norm_rad = OBC%rx_normal :2250-2262 */
        else rx_avg = rx_new;
        *nv = (xn[F3(A,c,k)] + rx_avg*xn[F3(A+d1,c,k)]) / (1.0+rx_avg);
        if (gamma_u < 1.0) r_normal[F3(A,c,k)] = rx_avg;
      } else if (S->oblique) {      /* :2349-2383 (E), :2593-2628 (W), :2838-2872 (N), :3082-3117 (S) */
        /* segment%grad_normal(q, 1 | 2, k) of gradient_at_q_points (:3407): the difference of the normal component along the boundary at the
         * corner point q, one face inside | on the boundary; zero beyond the corner points it is computed at */
        const int q0 = (ew ? S->JsdB : S->IsdB), q1 = (ew ? S->JedB : S->IedB);
        const int gq0 = max2i(q0, (ew ? G->jsd : G->isd)), gq1 = min2i(q1, (ew ? G->jed : G->ied) - 1);      /* (G%HI%JsdB + 1 = jsd) */
#define GN(q,a) (((q) < gq0 || (q) > gq1) ? 0.0 : (xn[F3(a,(q)+1,k)] - xn[F3(a,(q),k)]) * (ew ? G->mask2dBu[ORC_Q2(G,a,q)] : G->mask2dBu[ORC_Q2(G,q,a)]))
        dhdt = (xo[F3(A+d1,c,k)] - xn[F3(A+d1,c,k)]);
        dhdx = (xn[F3(A+d1,c,k)] - xn[F3(A+d2,c,k)]);
        const double g1c = GN(c, A+d1), g1m = GN(c-1, A+d1);
        double dhdy;
        if (dhdt*(g1c + g1m) > 0.0) dhdy = g1m;
        else if (dhdt*(g1c + g1m) == 0.0) dhdy = 0.0;
        else dhdy = g1c;
        if (dhdt*dhdx < 0.0) dhdt = 0.0;
        const double cff_new = max2(dhdx*dhdx + dhdy*dhdy, 1.0e-20);      /* eps = 1.0e-20*US%m_s_to_L_T**2 :2245 */
        const double rn_new = min2(dhdt*dhdx, cff_new*rx_max);
        const double rt_new = min2(cff_new, max2(dhdt*dhdy, -cff_new));
        /* the rate along the normal is rx for E / W and ry for N / S; the stored fields are OBC%rx_oblique_u ... cff_normal_v */
        double *rn_st = ew ? OBC->rx_oblique_u : OBC->ry_oblique_v, *rt_st = ew ? OBC->ry_oblique_u : OBC->rx_oblique_v;
        double *cf_st = ew ? OBC->cff_normal_u : OBC->cff_normal_v;
        double rn_avg, rt_avg, cff_avg;
        if (gamma_u < 1.0) {
          rn_avg = (1.0-gamma_u)*rn_st[F3(A,c,k)] + gamma_u*rn_new;
          rt_avg = (1.0-gamma_u)*rt_st[F3(A,c,k)] + gamma_u*rt_new;
          cff_avg = (1.0-gamma_u)*cf_st[F3(A,c,k)] + gamma_u*cff_new;
        } else { rn_avg = rn_new; rt_avg = rt_new; cff_avg = cff_new; }
        *nv = ((cff_avg*xn[F3(A,c,k)] + rn_avg*xn[F3(A+d1,c,k)]) - (max2(rt_avg,0.0)*GN(c-1, A) + min2(rt_avg,0.0)*GN(c, A))) / (cff_avg + rn_avg);
        if (gamma_u < 1.0) { rn_st[F3(A,c,k)] = rn_avg; rt_st[F3(A,c,k)] = rt_avg; cf_st[F3(A,c,k)] = cff_avg; }
#undef GN
      } else if (S->gradient) {
        *nv = xn[F3(A+d1,c,k)];
      }
      if ((S->radiation || S->oblique) && S->nudged) {
        const double tau = (dhdt*dhdx <= 0.0) ? S->Velocity_nudging_timescale_in : S->Velocity_nudging_timescale_out;
        const double gamma_2 = dt / (tau + dt);
        *nv = (1.0 - gamma_2) * *nv + gamma_2 * S->nudged_normal_vel[seg_idx(S, ew, A, c, k)];
      }
    }
    /* the tangential forms :2403-2455 (E), :2648-2700 (W), :2892-2945 (N), :3137-3190 (S): the rate at the corner points of the segment, then
     * segment%tangential_vel (ORLANSKI_TAN, NUDGED_TAN) and segment%tangential_grad (ORLANSKI_GRAD, NUDGED_GRAD) */
    if (S->radiation_tan_or_grad & (MOM6HIP_OBC_TAN_RADIATION | MOM6HIP_OBC_GRAD_RADIATION)) {
      const int bits = S->radiation_tan_or_grad;
      const int q0 = ew ? S->JsdB : S->IsdB, q1 = ew ? S->JedB : S->IedB;      /* the corner points along the segment */
      /* the tangential component at the corner q in the row / column t of cells: v(t, Q) for E / W, u(Q, t) for N / S */
      double *tn = ew ? v_new : u_new;
      const double *to = ew ? v_old : u_old;
#define T3(t,q,k) (ew ? ORC_V3(G,t,q,k) : ORC_U3(G,q,t,k))
#define TIDX(q,k) (ew ? (long)(A - S->IsdB) + (long)(S->IedB - S->IsdB + 1)*(((q) - S->JsdB) + (long)(S->JedB - S->JsdB + 1)*((k)-1)) \
                      : (long)((q) - S->IsdB) + (long)(S->IedB - S->IsdB + 1)*((A - S->JsdB) + (long)(S->JedB - S->JsdB + 1)*((k)-1)))
      const int t0 = plus ? A : A + 1, st = plus ? -1 : 1;      /* the first row of cells inside, and the step further in */
      if ((bits & MOM6HIP_OBC_TAN_RADIATION) && !S->tangential_vel) return 3;
      if ((bits & MOM6HIP_OBC_GRAD_RADIATION) && !S->tangential_grad) return 3;
      if ((bits & MOM6HIP_OBC_TAN_NUDGED) && !(S->tangential_vel && S->nudged_tangential_vel)) return 3;
      if ((bits & MOM6HIP_OBC_GRAD_NUDGED) && !(S->tangential_grad && S->nudged_tangential_grad)) return 3;
      const double *Idm = ew ? G->IdxBu : G->IdyBu;
#define QM(t,q) (ew ? ORC_Q2(G,t,q) : ORC_Q2(G,q,t))      /* the corner point between the rows t and t+1 of cells */
      const int g0 = (ew ? G->jsd : G->isd) + 1, g1 = (ew ? G->jed : G->ied) - 1;
      for (int k = 1; k <= nz; k++) for (int q = q0; q <= q1; q++) {
        double r_tang;
        if (gamma_u < 1.0) {      /* segment%rx_norm_rad at the two faces about the corner (the ends: the one face there is) */
          if (q == q0) r_tang = r_normal[F3(A,c0,k)];
          else if (q == q1) r_tang = r_normal[F3(A,c1,k)];
          else r_tang = 0.5*(r_normal[F3(A,q,k)] + r_normal[F3(A,q+1,k)]);
        } else {
          /* (the northern segment looks one row further in than the other three, :2904-2905) */
          const int ta = (!ew && plus) ? t0 + st : t0;
          const double dhdt = to[T3(ta,q,k)] - tn[T3(ta,q,k)];
          const double dhdx = tn[T3(ta,q,k)] - tn[T3(ta+st,q,k)];
          r_tang = 0.0;
          if (dhdt*dhdx > 0.0) r_tang = min2( (dhdt/dhdx), rx_max);
        }
        const double tau = (r_tang <= 0.0) ? S->Velocity_nudging_timescale_in : S->Velocity_nudging_timescale_out;
        const double gamma_2 = dt / (tau + dt);
        if (bits & MOM6HIP_OBC_TAN_RADIATION)
          S->tangential_vel[TIDX(q,k)] = (tn[T3(t0,q,k)] + r_tang*tn[T3(t0+st,q,k)]) / (1.0+r_tang);
        if (bits & MOM6HIP_OBC_TAN_NUDGED)
          S->tangential_vel[TIDX(q,k)] = (1.0 - gamma_2) * S->tangential_vel[TIDX(q,k)] + gamma_2 * S->nudged_tangential_vel[TIDX(q,k)];
        if ((bits & MOM6HIP_OBC_GRAD_RADIATION) && q >= g0 && q <= g1 && q >= q0 && q <= q1) {
          /* differences towards larger indices, with the metric of the corner point between the two rows */
          const int lo1 = plus ? t0 - 1 : t0, lo2 = plus ? t0 - 2 : t0 + 1;
          S->tangential_grad[TIDX(q,k)] = ((tn[T3(lo1+1,q,k)] - tn[T3(lo1,q,k)])*Idm[QM(lo1,q)] +
                                           r_tang*(tn[T3(lo2+1,q,k)] - tn[T3(lo2,q,k)])*Idm[QM(lo2,q)]) / (1.0+r_tang);
        }
        if (bits & MOM6HIP_OBC_GRAD_NUDGED)
          S->tangential_grad[TIDX(q,k)] = (1.0 - gamma_2) * S->tangential_grad[TIDX(q,k)] + gamma_2 * S->nudged_tangential_grad[TIDX(q,k)];
      }
#undef T3
#undef TIDX
#undef QM
    }
    /* the tangential forms of the oblique radiation :2456-2556 (E), :2701-2801 (W), :2946-3046 (N), :3191-3291 (S), with segment%grad_tan and
     * segment%grad_gradient of gradient_at_q_points (:3426-3443 and its twins) evaluated in place */
    if (S->radiation_tan_or_grad & (MOM6HIP_OBC_TAN_OBLIQUE | MOM6HIP_OBC_GRAD_OBLIQUE)) {
      const int bits = S->radiation_tan_or_grad;
      const int q0 = ew ? S->JsdB : S->IsdB, q1 = ew ? S->JedB : S->IedB;
      double *tn = ew ? v_new : u_new;
      const double *to = ew ? v_old : u_old;
#define T3(t,q,k) (ew ? ORC_V3(G,t,q,k) : ORC_U3(G,q,t,k))
#define TIDX(q,k) (ew ? (long)(A - S->IsdB) + (long)(S->IedB - S->IsdB + 1)*(((q) - S->JsdB) + (long)(S->JedB - S->JsdB + 1)*((k)-1)) \
                      : (long)((q) - S->IsdB) + (long)(S->IedB - S->IsdB + 1)*((A - S->JsdB) + (long)(S->JedB - S->JsdB + 1)*((k)-1)))
      const int t0 = plus ? A : A + 1, st = plus ? -1 : 1;
      if ((bits & (MOM6HIP_OBC_TAN_OBLIQUE | MOM6HIP_OBC_TAN_NUDGED)) && !S->tangential_vel) return 3;
      if ((bits & (MOM6HIP_OBC_GRAD_OBLIQUE | MOM6HIP_OBC_GRAD_NUDGED)) && !S->tangential_grad) return 3;
      if ((bits & MOM6HIP_OBC_TAN_NUDGED) && !S->nudged_tangential_vel) return 3;
      if ((bits & MOM6HIP_OBC_GRAD_NUDGED) && !S->nudged_tangential_grad) return 3;
      const double *Idm = ew ? G->IdxBu : G->IdyBu;
      const double *maskC = ew ? G->mask2dCu : G->mask2dCv;
#define QM(t,q) (ew ? ORC_Q2(G,t,q) : ORC_Q2(G,q,t))
#define CELL2(t,c) (ew ? ORC_H2(G,t,c) : ORC_H2(G,c,t))
#define FACE2(t,c) (ew ? ORC_U2(G,t,c) : ORC_V2(G,c,t))
      const int cd0 = ew ? G->jsd : G->isd, cd1 = ew ? G->jed : G->ied;      /* the cells of the data domain along the boundary */
      const int gt0 = max2i(c0 - 1, cd0), gt1 = min2i(c1 + 1, cd1);          /* where grad_tan is computed (zero beyond) */
      const int gg0 = max2i(c0, cd0 + 1), gg1 = min2i(c1, cd1 - 1);          /* where grad_gradient is */
      /* grad_tan(c, 1 | 2): the difference of the tangential component along the boundary in the cell c of the row t0 + st | t0 */
#define GT(c,t) (((c) < gt0 || (c) > gt1) ? 0.0 : (tn[T3(t,c,k)] - tn[T3(t,(c)-1,k)]) * G->mask2dT[CELL2(t,c)])
      /* grad_gradient(c, 2): the difference along the boundary of the gradient across it, between the rows lo1 and lo1 + 1 */
      const int lo1 = plus ? t0 - 1 : t0, lo2 = plus ? t0 - 2 : t0 + 1;
#define GG2(c) (((c) < gg0 || (c) > gg1) ? 0.0 : (((tn[T3(lo1+1,c,k)] - tn[T3(lo1,c,k)])*Idm[QM(lo1,c)]) - \
                                                     (tn[T3(lo1+1,(c)-1,k)] - tn[T3(lo1,(c)-1,k)])*Idm[QM(lo1,(c)-1)]) * maskC[FACE2(lo1,c)])
      double *rn_st = ew ? OBC->rx_oblique_u : OBC->ry_oblique_v, *rt_st = ew ? OBC->ry_oblique_u : OBC->rx_oblique_v;
      double *cf_st = ew ? OBC->cff_normal_u : OBC->cff_normal_v;
      if (gamma_u < 1.0 && !(rn_st && rt_st && cf_st)) return 3;
      for (int k = 1; k <= nz; k++) for (int q = q0; q <= q1; q++) {
        double rn, rt, cff;
        if (gamma_u < 1.0) {      /* the stored fields of the two faces about the corner (the ends: of the one face there is) */
          if (q == q0)      { rn = rn_st[F3(A,c0,k)]; rt = rt_st[F3(A,c0,k)]; cff = cf_st[F3(A,c0,k)]; }
          else if (q == q1) { rn = rn_st[F3(A,c1,k)]; rt = rt_st[F3(A,c1,k)]; cff = cf_st[F3(A,c1,k)]; }
          else { rn = 0.5*(rn_st[F3(A,q,k)] + rn_st[F3(A,q+1,k)]); rt = 0.5*(rt_st[F3(A,q,k)] + rt_st[F3(A,q+1,k)]);
                 cff = 0.5*(cf_st[F3(A,q,k)] + cf_st[F3(A,q+1,k)]); }
        } else {
          double dhdt = to[T3(t0,q,k)] - tn[T3(t0,q,k)];
          const double dhdn = tn[T3(t0,q,k)] - tn[T3(t0+st,q,k)];
          const double ga = GT(q, t0+st), gb = GT(q+1, t0+st);
          double dhdl;
          if (dhdt*(ga + gb) > 0.0) dhdl = ga;
          else if (dhdt*(ga + gb) == 0.0) dhdl = 0.0;
          else dhdl = gb;
          if (dhdt*dhdn < 0.0) dhdt = 0.0;
          cff = max2(dhdn*dhdn + dhdl*dhdl, 1.0e-20);
          rn = min2(dhdt*dhdn, cff*rx_max);
          rt = min2(cff, max2(dhdt*dhdl, -cff));
        }
        const double tau = (rn <= 0.0) ? S->Velocity_nudging_timescale_in : S->Velocity_nudging_timescale_out;
        const double gamma_2 = dt / (tau + dt);
        if (bits & MOM6HIP_OBC_TAN_OBLIQUE)
          S->tangential_vel[TIDX(q,k)] = ((cff*tn[T3(t0,q,k)] + rn*tn[T3(t0+st,q,k)]) - (max2(rt,0.0)*GT(q, t0) + min2(rt,0.0)*GT(q+1, t0))) / (cff + rn);
        if (bits & MOM6HIP_OBC_TAN_NUDGED)
          S->tangential_vel[TIDX(q,k)] = (1.0 - gamma_2) * S->tangential_vel[TIDX(q,k)] + gamma_2 * S->nudged_tangential_vel[TIDX(q,k)];
        if ((bits & MOM6HIP_OBC_GRAD_OBLIQUE) && q >= q0 + 1 && q <= q1 - 1)
          S->tangential_grad[TIDX(q,k)] = ((cff*(tn[T3(lo1+1,q,k)] - tn[T3(lo1,q,k)])*Idm[QM(lo1,q)] +
                                            rn*(tn[T3(lo2+1,q,k)] - tn[T3(lo2,q,k)])*Idm[QM(lo2,q)]) -
                                           (max2(rt,0.0)*GG2(q) + min2(rt,0.0)*GG2(q+1))) / (cff + rn);
        if (bits & MOM6HIP_OBC_GRAD_NUDGED)
          S->tangential_grad[TIDX(q,k)] = (1.0 - gamma_2) * S->tangential_grad[TIDX(q,k)] + gamma_2 * S->nudged_tangential_grad[TIDX(q,k)];
      }
#undef T3
#undef TIDX
#undef QM
#undef CELL2
#undef FACE2
#undef GT
#undef GG2
    }
  }
  /* open_boundary_apply_normal_flow :3337 */
  for (int n = 0; n < OBC->number_of_segments; n++) {
    const mom6hip_obc_segment_t *S = &OBC->segment[n];
    if (!S->on_pe) continue;
    if (!(S->radiation || S->oblique || S->gradient)) continue;
    const int ew = S->is_E_or_W != 0;
    if (!ew && !S->is_N_or_S) continue;
    const int A = ew ? S->IsdB : S->JsdB, c0 = ew ? S->jsd : S->isd, c1 = ew ? S->jed : S->ied;
    double *xn = ew ? u_new : v_new;
    for (int k = 1; k <= nz; k++) for (int c = c0; c <= c1; c++) xn[F3(A,c,k)] = S->normal_vel[seg_idx(S, ew, A, c, k)];
  }
#undef F3
  orc_halo_update(G, u_new, MOM6HIP_POS_U, nz); orc_halo_update(G, v_new, MOM6HIP_POS_V, nz);      /* pass_vector(u_new, v_new) :3309 */
  return 0;
}

int orc_open_boundary_zero_normal_flow(const mom6hip_grid_t *G, const mom6hip_obc_t *OBC, double *u, double *v)
{
  if (!OBC) return 0;
  for (int n = 0; n < OBC->number_of_segments; n++) {
    const mom6hip_obc_segment_t *S = &OBC->segment[n];
    if (!S->on_pe) continue;
    if (S->is_E_or_W) { for (int k = 1; k <= G->nk; k++) for (int j = S->jsd; j <= S->jed; j++) u[ORC_U3(G,S->IsdB,j,k)] = 0.; }
    else if (S->is_N_or_S) { for (int k = 1; k <= G->nk; k++) for (int i = S->isd; i <= S->ied; i++) v[ORC_V3(G,i,S->JsdB,k)] = 0.; }
  }
  return 0;
}

/* update_segment_tracer_reservoirs :5373-5502 (OBC%tres_x / tres_y, the restart copies, are not kept here) */
int orc_update_segment_tracer_reservoirs(const mom6hip_grid_t *G, const double *uhr, const double *vhr, const double *h,
                                         const mom6hip_obc_t *OBC, double dt, const double *const *tr, int ntr)
{
  (void)dt;
  if (!OBC || !OBC->OBC_pe) return 0;
  const int nz = G->nk;
  for (int n = 0; n < OBC->number_of_segments; n++) {
    const mom6hip_obc_segment_t *S = &OBC->segment[n];
    if (!S->tr_Reg) continue;
    const double b_in = (S->Tr_InvLscale_in == 0.0) ? 1.0 : 0.0, b_out = (S->Tr_InvLscale_out == 0.0) ? 1.0 : 0.0;
    const int ew = S->is_E_or_W != 0;
    if (!ew && !S->is_N_or_S) continue;
    const int A = ew ? S->IsdB : S->JsdB, c0 = ew ? S->jsd : S->isd, c1 = ew ? S->jed : S->ied;
    /* shift + A: the nearest interior tracer cell; dir switches the sign of the flow so that positive is into the reservoir */
    const int minus = (S->direction == MOM6HIP_OBC_DIRECTION_W) || (S->direction == MOM6HIP_OBC_DIRECTION_S);
    const int shift = minus ? 1 : 0;
    const double dir = minus ? -1.0 : 1.0;
    for (int c = c0; c <= c1; c++) {
      const long cell2 = ew ? ORC_H2(G, A + shift, c) : ORC_H2(G, c, A + shift);
      if (G->mask2dT[cell2] == 0.0) continue;
      const double len = ew ? G->dyCu[ORC_U2(G, A, c)] : G->dxCv[ORC_V2(G, c, A)];
      for (int q = 0; q < S->ntseg; q++) {
        const mom6hip_obc_segment_tracer_t *T = &S->tr_Reg[q];
        if (!T->tres) continue;
        if (T->ntr_index < 1 || T->ntr_index > ntr || !T->t) return 4;
        double *tres = (double *)T->tres;
        for (int k = 1; k <= nz; k++) {
          const double xr = dir * (ew ? uhr[ORC_U3(G, A, c, k)] : vhr[ORC_V3(G, c, A, k)]);
          const double hc = ew ? h[ORC_H3(G, A + shift, c, k)] : h[ORC_H3(G, c, A + shift, k)];
          const double tc = ew ? tr[T->ntr_index - 1][ORC_H3(G, A + shift, c, k)] : tr[T->ntr_index - 1][ORC_H3(G, c, A + shift, k)];
          const double a_out = b_out * fmax(0.0, copysign(1.0, xr));
          const double a_in  = b_in  * fmin(0.0, copysign(1.0, xr));
          const double L_out = fmax(0.0, xr*S->Tr_InvLscale_out*T->resrv_lfac_out / ((hc + G->H_subroundoff)*len));
          const double L_in  = fmin(0.0, xr*S->Tr_InvLscale_in*T->resrv_lfac_in / ((hc + G->H_subroundoff)*len));
          const double fac1 = (1.0 - (a_out - a_in)) + ((L_out + a_out) - (L_in + a_in));
          const long s3 = seg_idx(S, ew, A, c, k);
          tres[s3] = (1.0/fac1) * ((1.0-a_out+a_in)*tres[s3] + ((L_out+a_out)*tc - (L_in+a_in)*T->t[s3]));
        }
      }
    }
  }
  return 0;
}
