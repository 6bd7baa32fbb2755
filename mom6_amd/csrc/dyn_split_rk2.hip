// dyn_split_rk2.hip -- step_MOM_dyn_split_RK2 on MI355X: the orchestration of the split RK2 step and its own
// streaming sweeps (src/core/MOM_dynamics_split_RK2.F90:289-1176; state initialisation :1521-1622).
//
// The whole step is enqueued on the context's stream in the reference's order: PressureForce, [CorAdCalc],
// continuity[BT_cont], btcalc, bt_mass_source, btstep (predictor), continuity (predictor), CorAdCalc, btstep
// (corrector), continuity (corrector), [CorAdCalc for the next predictor], with the group passes of the reference at
// the reference's seams.  The momentum sweeps (:557-564, :582-589, :667-676, :785-787, :879-886, :930-939,
// :1000-1002, :1038-1053) are plain i-contiguous streaming kernels; each is HBM-bound at 24-40 B/cell.
// All arrays are device arrays: nothing is staged, and without hooks / multi-tile passes / calc_dtbt nothing
// synchronises with the host.
#include "common.hpp"

#include <cstdlib>
#include <initializer_list>
#include <utility>

namespace m6 {

}

namespace {

template <class F> __global__ void __launch_bounds__(256) range3d_kernel(int i0, int i1, int j0, int j1, F f) {
  const int i = i0 + blockIdx.x * 64 + threadIdx.x, j = j0 + blockIdx.y * 4 + threadIdx.y;
  if (i <= i1 && j <= j1) f(i, j, (int)blockIdx.z);
}
// one thread per (i, j, k) of an inclusive horizontal range, all layers, i fastest
template <class F> void launch3d(hipStream_t st, int i0, int i1, int j0, int j1, int nk, F f) {
  if (i1 < i0 || j1 < j0) return;
  dim3 grid((i1 - i0 + 64) / 64, (j1 - j0 + 4) / 4, nk), block(64, 4);
  hipLaunchKernelGGL(range3d_kernel<F>, grid, block, 0, st, i0, i1, j0, j1, f);
}

struct Sz { size_t h2, h3, u3, v3; };
Sz sizes(const m6::GridDev &g) {
  Sz s;
  s.h2 = sizeof(double) * (size_t)g.nih * g.njh; s.h3 = s.h2 * g.nk;
  s.u3 = sizeof(double) * (size_t)(g.nih + 1) * g.njh * g.nk; s.v3 = sizeof(double) * (size_t)g.nih * (g.njh + 1) * g.nk;
  return s;
}

int pass(mom6hip_ctx_t *ctx, std::initializer_list<std::pair<double *, int>> fl, int nk3) {
  std::vector<double *> f; std::vector<int32_t> pos, nk;
  for (auto &e : fl) { f.push_back(e.first); pos.push_back(e.second & (3 | MOM6HIP_PASS_SCALAR_PAIR)); nk.push_back((e.second & 4) ? 1 : nk3); }
  return m6::group_pass(ctx, f.data(), pos.data(), nk.data(), (int)f.size());
}
// The non-blocking form (start_group_pass / complete_group_pass, MOM_domain_infra.F90:1141-1182): the exchange runs on the
// communication stream while the compute stream goes on with whatever does not read the halos in flight.
int pass_start(mom6hip_ctx_t *ctx, std::initializer_list<std::pair<double *, int>> fl, int nk3, int seam = 0) {
  std::vector<double *> f; std::vector<int32_t> pos, nk;
  for (auto &e : fl) { f.push_back(e.first); pos.push_back(e.second & (3 | MOM6HIP_PASS_SCALAR_PAIR)); nk.push_back((e.second & 4) ? 1 : nk3); }
  if (int rc = m6::start_group_pass(ctx, f.data(), pos.data(), nk.data(), (int)f.size())) return rc;
  // MOM6HIP_BLOCKING_PASSES=1: every pass completes where it starts (G%nonblocking_updates = False; comparison runs)
  // (=N > 1: a bit mask of the seams of the step, in order, that complete at once -- debugging)
  static const int blocking = [] { const char *e = getenv("MOM6HIP_BLOCKING_PASSES"); return e ? atoi(e) : 0; }();
  const bool now = blocking == 1 || (blocking > 1 && ((blocking >> 1) >> seam) & 1);
  return now ? m6::complete_group_pass(ctx) : 0;
}
// Whether the rows of the tile are worth splitting around a pass in flight: the pass must leave the x halos final at its start
// (the tile spans x) and the tile must be tall enough to have inner rows.
bool split_rows(mom6hip_ctx_t *ctx) {
  const int W = ctx->host.isc - ctx->host.isd, nj = ctx->host.jec - ctx->host.jsc + 1;
  return m6::pass_leaves_x_final(ctx) && nj >= 4 * W + 8 && (m6::multi_tile(ctx) || ctx->poison_passes || ctx->split_rows_always);
}
// `work` (kernels that are pure functions of their inputs row by row, taking their rows from ctx->g) around the completion of the
// pass in flight: the rows at least a halo width inside the tile before it, the two bands along the edges after it.
template <class F> int around_pass(mom6hip_ctx_t *ctx, F work) {
  if (!split_rows(ctx)) {
    ctx->overlap[2]++;
    if (int rc = m6::complete_group_pass(ctx)) return rc;
    return work();
  }
  ctx->overlap[0]++;
  const int js = ctx->host.jsc, je = ctx->host.jec, W = ctx->host.isc - ctx->host.isd;
  m6::row_window(ctx, js + W, je - W);
  int rc = work();
  m6::row_window_reset(ctx);
  // (an error between start and complete: the exchange in flight is completed all the same, so that the context is left without
  // a pending pass and the FIRST error is the one reported, not "a pass is already in flight" from the next call)
  if (rc) { m6::ErrorKeeper keep; m6::complete_group_pass(ctx); return rc; }
  if ((rc = m6::complete_group_pass(ctx))) return rc;
  m6::row_window(ctx, js, js + W - 1);
  rc = work();
  if (!rc) { m6::row_window(ctx, je - W + 1, je); rc = work(); }
  m6::row_window_reset(ctx);
  return rc;
}
// The continuity around the completion of the pass in flight: its zonal pass reads a row at a time, so the tile's own rows go
// first, with the meridional faces and cells whose stencil stays inside them (mom6hip_continuity, ctx->cont_phase = 1); the halo
// rows and the two edges follow the completion (cont_phase = 2).  x first, no fold on this tile, rows worth splitting; otherwise
// the pass completes and the continuity is the one call it always was.
template <class F> int continuity_around_pass(mom6hip_ctx_t *ctx, F call) {
  const bool phased = split_rows(ctx) && (ctx->host.first_direction % 2) == 0 && !ctx->host.tripolar_n;
  if (!phased) {
    ctx->overlap[2]++;
    if (int rc = m6::complete_group_pass(ctx)) return rc;
    return call();
  }
  ctx->overlap[1]++;
  ctx->cont_phase = 1;
  int rc = call();
  ctx->cont_phase = 0;
  if (rc) { m6::ErrorKeeper keep; m6::complete_group_pass(ctx); return rc; }
  if ((rc = m6::complete_group_pass(ctx))) return rc;
  ctx->cont_phase = 2;
  rc = call();
  ctx->cont_phase = 0;
  return rc;
}
constexpr int PH = MOM6HIP_POS_H, PU = MOM6HIP_POS_U, PV = MOM6HIP_POS_V, P2D = 4;
constexpr int PUs = PU | MOM6HIP_PASS_SCALAR_PAIR, PVs = PV | MOM6HIP_PASS_SCALAR_PAIR;      // To_All+SCALAR_PAIR (:462)

int check(const mom6hip_dyn_split_rk2_cs_t *cs, const char *who) {
  M6_REQUIRE(cs != nullptr, "%s: null control structure", who);
  M6_REQUIRE(cs->begw == 0.0, "%s: BEGW /= 0 is not provided", who);
  M6_REQUIRE(!cs->split_bottom_stress, "%s: SPLIT_BOTTOM_STRESS is not provided", who);
  // (eqn_of_state may be null: no equation of state, the layered PressureForce branch with GV%Rlay / GV%g_prime)
  M6_REQUIRE(cs->continuity_CSp && cs->CoriolisAdv && cs->PressureForce_CSp && cs->barotropic_CSp,
             "%s: a sub-module control structure is missing", who);
  M6_REQUIRE(cs->CAu && cs->CAv && cs->CAu_pred && cs->CAv_pred && cs->PFu && cs->PFv && cs->diffu && cs->diffv && cs->visc_rem_u &&
                 cs->visc_rem_v && cs->u_accel_bt && cs->v_accel_bt && cs->u_av && cs->v_av && cs->h_av && cs->pbce && cs->eta &&
                 cs->eta_PF && cs->uhbt && cs->vhbt, "%s: an array of the control structure is not allocated", who);
  return 0;
}

#define CALL(x) do { if (int rc_ = (x)) return rc_; } while (0)

// The surface pressure of a step (MOM_dynamics_split_RK2.F90:435-442): p_surf_end when both p_surf_begin and p_surf_end are given
// (dyn_p_surf), else forces%p_surf; and, after PressureForce, the eta that corresponds to the starting pressure (:497-503), which
// btstep takes as eta_PF_start (null without dyn_p_surf).
const double *step_p_surf(const mom6hip_dyn_split_rk2_cs_t *cs) {
  return (cs->p_surf_begin && cs->p_surf_end) ? cs->p_surf_end : cs->p_surf;
}
int step_eta_PF_start(mom6hip_ctx_t *ctx, const mom6hip_dyn_split_rk2_cs_t *cs, double **out) {
  *out = nullptr;
  if (!(cs->p_surf_begin && cs->p_surf_end)) return 0;
  const m6::GridDev g = ctx->g;
  const size_t bytes = (size_t)g.nih * g.njh * sizeof(double);
  M6_REQUIRE(ctx->rk2_eta_PF_start.reserve(bytes) == 0 && ctx->rk2_eta_PF_start.p, "step_MOM_dyn_split_RK2: out of device memory");
  double *eps = (double *)ctx->rk2_eta_PF_start.p;
  M6_HIP(hipMemsetAsync(eps, 0, bytes, ctx->stream));                                                // :439
  const double pres_to_eta = 1.0 / (g.g_Earth * (g.Rho0 * g.H_to_Z));                                // 1 / (GV%g_Earth * GV%H_to_RZ), Boussinesq
  const double *eta_PF = cs->eta_PF, *pb = cs->p_surf_begin, *pe = cs->p_surf_end;
  launch3d(ctx->stream, g.isc - 1, g.iec + 1, g.jsc - 1, g.jec + 1, 1, [=] __device__(int i, int j, int) {      // Isq .. Ieq+1, Jsq .. Jeq+1
    const long n = g.h2(i, j);
    eps[n] = eta_PF[n] - pres_to_eta * (pb[n] - pe[n]);
  });
  *out = eps;
  return 0;
}

// step_MOM_dyn_split_RK2 with CS%OBC associated: the reference's sequence of calls one after the other, every operator through its entry
// point with the OBC (regional grids are small: no fused sweeps, no work around the passes in flight), plus the step's own lines for
// the open boundaries: the starting velocities of the radiation (:444-456), open_boundary_zero_normal_flow on the accelerations
// (:565-567, :887-889), radiation_open_bdry_conds on u_av (:765-775) and u_inst (:1030-1034).
int step_with_obc(mom6hip_ctx_t *ctx, mom6hip_dyn_split_rk2_cs_t *cs, double *u_inst, double *v_inst, double *h, const double *T, const double *S,
                  double dt, const double *taux, const double *tauy, double RZ_to_H, double *uh, double *vh, double *uhtr, double *vhtr,
                  double *eta_av, int32_t calc_dtbt) {
  const mom6hip_obc_t *OBC = cs->OBC;
  const m6::GridDev g = ctx->g;
  const Sz sz = sizes(g);
  hipStream_t s = ctx->stream;
  const int D = MOM6HIP_MEM_DEVICE;
  const int is = g.isc, ie = g.iec, js = g.jsc, je = g.jec, nz = g.nk;
  const int Isq = is - 1, Ieq = ie, Jsq = js - 1, Jeq = je;
  mom6hip_barotropic_cs_t *BT = cs->barotropic_CSp;
  const mom6hip_bt_cont_t *BTC = cs->BT_cont;
  const bool BT_cont_BT_thick = BTC && BTC->h_u && BTC->h_v;
  mom6hip_vertvisc_cs_t *VV = cs->vertvisc_CSp;
  M6_REQUIRE(!cs->hooks, "step_MOM_dyn_split_RK2: host-side parameterisations (hooks) are not provided with an associated OBC");
  // (an associated OBC on several tiles: every tile holds the segments clipped to its data domain, as open_boundary_config leaves them on a PE;
  // tests/test_domains.py::test_rk2_step_with_open_boundaries_layout_independence.  MOM6HIP_OBC_ONE_TILE=1 brings the old refusal back)
  static const bool obc_one_tile = getenv("MOM6HIP_OBC_ONE_TILE") && atoi(getenv("MOM6HIP_OBC_ONE_TILE")) == 1;
  M6_REQUIRE(!obc_one_tile || !m6::multi_tile(ctx), "step_MOM_dyn_split_RK2: an associated OBC is provided on one tile (MOM6HIP_OBC_ONE_TILE)");

  // the step's automatic arrays (:336-369), with u_old_rad_OBC, v_old_rad_OBC (:360-363)
  const size_t blk_bytes = 4 * sz.u3 + 4 * sz.v3 + sz.h3 + sz.h2;
  const bool fresh = ctx->rk2_scratch.bytes < blk_bytes || ctx->rk2_scratch_layout != 3;
  M6_REQUIRE(ctx->rk2_scratch.reserve(blk_bytes) == 0, "step_MOM_dyn_split_RK2: out of device memory");
  ctx->rk2_scratch_layout = 3;
  char *blk = (char *)ctx->rk2_scratch.p;
  double *up = (double *)blk, *u_bc = (double *)(blk + sz.u3), *uh_in = (double *)(blk + 2 * sz.u3), *u_old = (double *)(blk + 3 * sz.u3);
  char *vb = blk + 4 * sz.u3;
  double *vp = (double *)vb, *v_bc = (double *)(vb + sz.v3), *vh_in = (double *)(vb + 2 * sz.v3), *v_old = (double *)(vb + 3 * sz.v3);
  double *hp = (double *)(vb + 4 * sz.v3), *eta_pred = (double *)(vb + 4 * sz.v3 + sz.h3);
  double *u_av = cs->u_av, *v_av = cs->v_av, *h_av = cs->h_av, *eta = cs->eta;
  if (fresh) M6_HIP(hipMemsetAsync(blk, 0, blk_bytes, s));                                           // :419-421
  M6_HIP(hipMemcpyAsync(hp, h, sz.h3, hipMemcpyDeviceToDevice, s));                                  // :422
  M6_HIP(hipMemcpyAsync(u_old, u_av, sz.u3, hipMemcpyDeviceToDevice, s));                            // :450-455
  M6_HIP(hipMemcpyAsync(v_old, v_av, sz.v3, hipMemcpyDeviceToDevice, s));

  CALL(mom6hip_pressureforce_fv_bouss(ctx, cs->PressureForce_CSp, cs->eqn_of_state, h, T, S, step_p_surf(cs), cs->PFu, cs->PFv, cs->pbce,   // :495
                                      cs->eta_PF, D));
  double *eta_PF_start = nullptr;
  CALL(step_eta_PF_start(ctx, cs, &eta_PF_start));                                                   // :497-503
  if (!cs->CAu_pred_stored)   // :544-552
    CALL(mom6hip_coradcalc_obc(ctx, cs->CoriolisAdv, OBC, u_av, v_av, h_av, uh, vh, cs->CAu_pred, cs->CAv_pred, D));
  auto bc_accel = [&](const double *CAu, const double *CAv) -> int {      // :557-567, :879-889
    const double *PFu = cs->PFu, *PFv = cs->PFv, *diffu = cs->diffu, *diffv = cs->diffv;
    launch3d(s, Isq, Ieq, js, je, nz, [=] __device__(int I, int j, int k) {
      const long n = g.u3(I, j, k);
      u_bc[n] = (CAu[n] + PFu[n]) + diffu[n];
    });
    launch3d(s, is, ie, Jsq, Jeq, nz, [=] __device__(int i, int J, int k) {
      const long n = g.v3(i, J, k);
      v_bc[n] = (CAv[n] + PFv[n]) + diffv[n];
    });
    return mom6hip_open_boundary_zero_normal_flow(ctx, OBC, u_bc, v_bc, D);
  };
  auto increment = [&](double *uo, double *vo, double dtx, bool with_bt) {      // :582-589, :667-676, :930-939
    const double *abu = cs->u_accel_bt, *abv = cs->v_accel_bt;
    launch3d(s, is, ie, Jsq, Jeq, nz, [=] __device__(int i, int J, int k) {
      const long n = g.v3(i, J, k);
      vo[n] = g.mask2dCv[g.v2(i, J)] * (v_inst[n] + dtx * (with_bt ? (v_bc[n] + abv[n]) : v_bc[n]));
    });
    launch3d(s, Isq, Ieq, js, je, nz, [=] __device__(int I, int j, int k) {
      const long n = g.u3(I, j, k);
      uo[n] = g.mask2dCu[g.u2(I, j)] * (u_inst[n] + dtx * (with_bt ? (u_bc[n] + abu[n]) : u_bc[n]));
    });
  };
  CALL(bc_accel(cs->CAu_pred, cs->CAv_pred));
  increment(up, vp, dt, false);
  if (VV && cs->set_visc_CSp && cs->set_visc_CSp->dynamic_viscous_ML) {      // set_viscous_ML :592 (the OBC acts there under ice shelves only)
    M6_REQUIRE(cs->visc && cs->visc->ustar && cs->visc->nkml_visc_u && cs->visc->nkml_visc_v,
               "step_MOM_dyn_split_RK2: DYNAMIC_VISCOUS_ML needs forces%%ustar (visc->ustar) and visc%%nkml_visc_u / nkml_visc_v");
    CALL(m6::set_viscous_ML_dev(ctx, cs->set_visc_CSp, u_inst, v_inst, h, T, S, cs->eqn_of_state, taux, tauy, cs->visc->ustar,
                                (double *)cs->visc->nkml_visc_u, (double *)cs->visc->nkml_visc_v, dt));
  }
  if (VV) {      // vertvisc_coef, vertvisc_remnant :598-600
    CALL(mom6hip_vertvisc_coef_obc(ctx, VV, up, vp, h, nullptr, cs->visc, dt, OBC, D));
    CALL(mom6hip_vertvisc_remnant(ctx, VV, cs->visc, cs->visc_rem_u, cs->visc_rem_v, dt, D));
  }
  CALL(pass(ctx, {{eta, PH | P2D}, {cs->visc_rem_u, PUs}, {cs->visc_rem_v, PVs}}, nz));            // :610-611
  if (!BT_cont_BT_thick) CALL(mom6hip_btcalc_obc(ctx, BT, h, nullptr, nullptr, 0, OBC, D));          // :618
  CALL(mom6hip_bt_mass_source(ctx, BT, h, eta, 1, D));
  if (BTC || cs->BT_use_layer_fluxes) {      // :634-644
    CALL(mom6hip_continuity_obc(ctx, cs->continuity_CSp, OBC, u_inst, v_inst, h, hp, uh_in, vh_in, dt, nullptr, nullptr, cs->visc_rem_u,
                                cs->visc_rem_v, nullptr, nullptr, BTC, nullptr, nullptr, D));
    if (BT_cont_BT_thick) CALL(mom6hip_btcalc_obc(ctx, BT, h, BTC->h_u, BTC->h_v, 0, OBC, D));
  }
  if (calc_dtbt) CALL(mom6hip_set_dtbt_eta(ctx, BT, eta, cs->pbce, nullptr, 0.0, 0.0, D));           // :651
  const bool lf = cs->BT_use_layer_fluxes != 0;
  CALL(mom6hip_btstep_obc(ctx, BT, u_inst, v_inst, eta, dt, u_bc, v_bc, taux, tauy, RZ_to_H, cs->pbce, cs->eta_PF, u_av, v_av,   // :655
                          cs->u_accel_bt, cs->v_accel_bt, eta_pred, cs->uhbt, cs->vhbt, cs->visc_rem_u, cs->visc_rem_v, BTC, eta_PF_start, nullptr,
                          nullptr, lf ? uh_in : nullptr, lf ? vh_in : nullptr, lf ? u_inst : nullptr, lf ? v_inst : nullptr, nullptr, OBC, D));
  const double dt_pred = dt * cs->be;
  increment(up, vp, dt_pred, true);                                                                    // :663-676
  if (VV) {      // :717-744
    CALL(mom6hip_vertvisc_coef_obc(ctx, VV, up, vp, h, nullptr, cs->visc, dt_pred, OBC, D));
    CALL(mom6hip_vertvisc_obc(ctx, VV, up, vp, h, taux, tauy, cs->visc, dt_pred, nullptr, nullptr, OBC, D));
    CALL(mom6hip_vertvisc_remnant(ctx, VV, cs->visc, cs->visc_rem_u, cs->visc_rem_v, dt_pred, D));
  }
  CALL(pass(ctx, {{cs->visc_rem_u, PUs}, {cs->visc_rem_v, PVs}, {up, PU}, {vp, PV}}, nz));          // :747-751
  CALL(mom6hip_continuity_obc(ctx, cs->continuity_CSp, OBC, up, vp, h, hp, uh, vh, dt, cs->uhbt, cs->vhbt, cs->visc_rem_u, cs->visc_rem_v,   // :757
                              u_av, v_av, BTC, nullptr, nullptr, D));
  CALL(pass(ctx, {{hp, PH}, {u_av, PU}, {v_av, PV}, {uh, PU}, {vh, PV}}, nz));                       // :763
  CALL(mom6hip_radiation_open_bdry_conds(ctx, OBC, OBC->gamma_uv, OBC->rx_max, OBC->rx_normal, OBC->ry_normal, u_av, u_old, v_av, v_old,   // :770
                                         dt_pred, D));
  launch3d(s, is - 2, ie + 2, js - 2, je + 2, nz, [=] __device__(int i, int j, int k) {              // :785-787
    const long n = g.h3(i, j, k);
    h_av[n] = 0.5 * (h[n] + hp[n]);
  });
  CALL(mom6hip_bt_mass_source(ctx, BT, hp, eta_pred, 0, D));                                           // :797
  if (BT_cont_BT_thick) CALL(mom6hip_btcalc_obc(ctx, BT, h, BTC->h_u, BTC->h_v, 0, OBC, D));         // :843
  if (cs->hor_visc)      // :860
    CALL(mom6hip_horizontal_viscosity_obc(ctx, cs->hor_visc, u_av, v_av, h_av, cs->diffu, cs->diffv, dt, BTC ? BTC->h_u : nullptr,
                                          BTC ? BTC->h_v : nullptr, OBC, D));
  CALL(mom6hip_coradcalc_obc(ctx, cs->CoriolisAdv, OBC, u_av, v_av, h_av, uh, vh, cs->CAu, cs->CAv, D));   // :869
  CALL(bc_accel(cs->CAu, cs->CAv));                                                                   // :879-889
  CALL(mom6hip_btstep_obc(ctx, BT, u_inst, v_inst, eta, dt, u_bc, v_bc, taux, tauy, RZ_to_H, cs->pbce, cs->eta_PF, u_av, v_av,   // :911
                          cs->u_accel_bt, cs->v_accel_bt, eta_pred, cs->uhbt, cs->vhbt, cs->visc_rem_u, cs->visc_rem_v, BTC, eta_PF_start, nullptr,
                          nullptr, lf ? uh : nullptr, lf ? vh : nullptr, lf ? u_av : nullptr, lf ? v_av : nullptr, eta_av, OBC, D));
  launch3d(s, is, ie, js, je, 1, [=] __device__(int i, int j, int) { eta[g.h2(i, j)] = eta_pred[g.h2(i, j)]; });   // :918
  increment(u_inst, v_inst, dt, true);                                                                 // :928-939 (in place: a point reads itself)
  if (VV) {      // :974-994
    CALL(mom6hip_vertvisc_coef_obc(ctx, VV, u_inst, v_inst, h, nullptr, cs->visc, dt, OBC, D));
    CALL(mom6hip_vertvisc_obc(ctx, VV, u_inst, v_inst, h, taux, tauy, cs->visc, dt, nullptr, nullptr, OBC, D));
    CALL(mom6hip_vertvisc_remnant(ctx, VV, cs->visc, cs->visc_rem_u, cs->visc_rem_v, dt, D));
  }
  launch3d(s, is - 2, ie + 2, js - 2, je + 2, nz, [=] __device__(int i, int j, int k) { h_av[g.h3(i, j, k)] = h[g.h3(i, j, k)]; });   // :1000
  CALL(pass(ctx, {{cs->visc_rem_u, PUs}, {cs->visc_rem_v, PVs}, {u_inst, PU}, {v_inst, PV}}, nz));  // :1004-1008
  CALL(mom6hip_continuity_obc(ctx, cs->continuity_CSp, OBC, u_inst, v_inst, h, h, uh, vh, dt, cs->uhbt, cs->vhbt, cs->visc_rem_u,   // :1015
                              cs->visc_rem_v, u_av, v_av, nullptr, nullptr, nullptr, D));
  CALL(pass(ctx, {{h, PH}, {u_av, PU}, {v_av, PV}, {uh, PU}, {vh, PV}}, nz));                        // :1018, :1027
  CALL(mom6hip_radiation_open_bdry_conds(ctx, OBC, OBC->gamma_uv, OBC->rx_max, OBC->rx_normal, OBC->ry_normal, u_inst, u_old, v_inst, v_old,   // :1033
                                         dt, D));
  launch3d(s, is - 2, ie + 2, js - 2, je + 2, nz, [=] __device__(int i, int j, int k) {              // :1038-1040
    const long n = g.h3(i, j, k);
    h_av[n] = 0.5 * (h_av[n] + h[n]);
  });
  launch3d(s, Isq - 2, Ieq + 2, js - 2, je + 2, nz, [=] __device__(int I, int j, int k) {            // :1046-1053
    const long n = g.u3(I, j, k);
    uhtr[n] = uhtr[n] + uh[n] * dt;
  });
  launch3d(s, is - 2, ie + 2, Jsq - 2, Jeq + 2, nz, [=] __device__(int i, int J, int k) {
    const long n = g.v3(i, J, k);
    vhtr[n] = vhtr[n] + vh[n] * dt;
  });
  if (cs->store_CAu) {      // :1055-1069
    CALL(mom6hip_coradcalc_obc(ctx, cs->CoriolisAdv, OBC, u_av, v_av, h_av, uh, vh, cs->CAu_pred, cs->CAv_pred, D));
    cs->CAu_pred_stored = 1;
  } else {
    cs->CAu_pred_stored = 0;
  }
  M6_HIP(hipGetLastError());
  return 0;
}

}  // namespace

extern "C" {

int mom6hip_dyn_split_rk2_init(mom6hip_ctx_t *ctx, mom6hip_dyn_split_rk2_cs_t *cs, const double *u, const double *v, const double *h,
                               double *uh, double *vh, double dt) {
  M6_REQUIRE(ctx && u && v && h && uh && vh, "dyn_split_rk2_init: null argument");
  CALL(check(cs, "dyn_split_rk2_init"));
  const m6::GridDev g = ctx->g;
  const Sz sz = sizes(g);
  hipStream_t s = ctx->stream;
  const int D = MOM6HIP_MEM_DEVICE;
  // eta :1521-1535
  {
    double *eta = cs->eta;
    const double Z_to_H = g.Z_to_H;
    const long hstr = (long)g.nih * g.njh;
    const int nz = g.nk;
    launch3d(s, g.isc, g.iec, g.jsc, g.jec, 1, [=] __device__(int i, int j, int) {
      const long n = g.h2(i, j);
      double e = -Z_to_H * g.bathyT[n];
      for (int k = 0; k < nz; k++) e = e + h[n + hstr * k];
      eta[n] = e;
    });
  }
  M6_HIP(hipMemsetAsync(cs->diffu, 0, sz.u3, s)); M6_HIP(hipMemsetAsync(cs->diffv, 0, sz.v3, s));
  if (cs->hor_visc) {   // :1543-1550
    const mom6hip_bt_cont_t *B = cs->BT_cont;
    M6_REQUIRE(cs->hor_visc->initialized, "MOM_hor_visc: Module must be initialized before it is used.");
    if (cs->OBC) CALL(mom6hip_horizontal_viscosity_obc(ctx, cs->hor_visc, u, v, h, cs->diffu, cs->diffv, dt, B ? B->h_u : nullptr, B ? B->h_v : nullptr, cs->OBC, D));
    else if (m6::horizontal_viscosity_dev(ctx, cs->hor_visc, u, v, h, cs->diffu, cs->diffv, B ? B->h_u : nullptr, B ? B->h_v : nullptr)) return 1;
  } else if (cs->hooks && cs->hooks->horizontal_viscosity) {
    M6_HIP(hipStreamSynchronize(s));
    M6_REQUIRE(cs->hooks->horizontal_viscosity(cs->hooks->user, u, v, h, cs->diffu, cs->diffv) == 0, "horizontal_viscosity hook failed");
  }
  {
    double *vru = cs->visc_rem_u, *vrv = cs->visc_rem_v;
    launch3d(s, g.isd - 1, g.ied, g.jsd, g.jed, g.nk, [=] __device__(int i, int j, int k) { vru[g.u3(i, j, k)] = 1.0; });
    launch3d(s, g.isd, g.ied, g.jsd - 1, g.jed, g.nk, [=] __device__(int i, int j, int k) { vrv[g.v3(i, j, k)] = 1.0; });
  }
  M6_HIP(hipMemcpyAsync(cs->u_av, u, sz.u3, hipMemcpyDeviceToDevice, s));     // :1552-1558
  M6_HIP(hipMemcpyAsync(cs->v_av, v, sz.v3, hipMemcpyDeviceToDevice, s));
  // :1560-1610: first transports, h_av and (store_CAu) the predictor's Coriolis terms
  {
    double *h_tmp = cs->CAu;   // free at this point; CAu is rewritten by the first step before it is read
    M6_REQUIRE(sz.u3 >= sz.h3, "dyn_split_rk2_init: internal scratch too small");
    M6_HIP(hipMemcpyAsync(h_tmp, h, sz.h3, hipMemcpyDeviceToDevice, s));
    const double *uu = cs->store_CAu ? cs->u_av : u, *vv = cs->store_CAu ? cs->v_av : v;
    CALL(mom6hip_continuity_obc(ctx, cs->continuity_CSp, cs->OBC, uu, vv, h, h_tmp, uh, vh, dt, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                nullptr, nullptr, nullptr, D));
    CALL(pass(ctx, {{h_tmp, PH}}, g.nk));
    double *h_av = cs->h_av;
    launch3d(s, g.isd, g.ied, g.jsd, g.jed, g.nk, [=] __device__(int i, int j, int k) {
      const long n = g.h3(i, j, k);
      h_av[n] = 0.5 * (h[n] + h_tmp[n]);
    });
  }
  if (cs->store_CAu) {
    CALL(pass(ctx, {{cs->u_av, PU}, {cs->v_av, PV}, {uh, PU}, {vh, PV}}, g.nk));
    CALL(mom6hip_coradcalc_obc(ctx, cs->CoriolisAdv, cs->OBC, cs->u_av, cs->v_av, cs->h_av, uh, vh, cs->CAu_pred, cs->CAv_pred, D));
    cs->CAu_pred_stored = 1;
    CALL(pass(ctx, {{cs->u_av, PU}, {cs->v_av, PV}, {cs->CAu_pred, PU}, {cs->CAv_pred, PV}}, g.nk));   // :1615-1622
  } else {
    cs->CAu_pred_stored = 0;
    CALL(pass(ctx, {{cs->u_av, PU}, {cs->v_av, PV}, {cs->h_av, PH}, {uh, PU}, {vh, PV}}, g.nk));
  }
  M6_HIP(hipGetLastError());
  return 0;
}

int mom6hip_step_dyn_split_rk2(mom6hip_ctx_t *ctx, mom6hip_dyn_split_rk2_cs_t *cs, double *u_inst, double *v_inst, double *h,
                               const double *T, const double *S, double dt, const double *taux, const double *tauy, double RZ_to_H,
                               double *uh, double *vh, double *uhtr, double *vhtr, double *eta_av, int32_t calc_dtbt) {
  M6_REQUIRE(ctx && u_inst && v_inst && h && taux && tauy && uh && vh && uhtr && vhtr && eta_av, "step_MOM_dyn_split_RK2: null argument");
  CALL(check(cs, "step_MOM_dyn_split_RK2"));
  M6_REQUIRE(!cs->eqn_of_state || (T && S), "step_MOM_dyn_split_RK2: an equation of state needs tv%%T and tv%%S");
  if (cs->OBC) return step_with_obc(ctx, cs, u_inst, v_inst, h, T, S, dt, taux, tauy, RZ_to_H, uh, vh, uhtr, vhtr, eta_av, calc_dtbt);
  const m6::GridDev g = ctx->g;
  const Sz sz = sizes(g);
  hipStream_t s = ctx->stream;
  const int D = MOM6HIP_MEM_DEVICE;
  const int is = g.isc, ie = g.iec, js = g.jsc, je = g.jec, nz = g.nk;
  const int Isq = is - 1, Ieq = ie, Jsq = js - 1, Jeq = je;
  mom6hip_barotropic_cs_t *BT = cs->barotropic_CSp;
  const mom6hip_bt_cont_t *BTC = cs->BT_cont;
  const bool BT_cont_BT_thick = BTC && BTC->h_u && BTC->h_v;
  const mom6hip_visc_hooks_t *hk = cs->hooks;
  mom6hip_vertvisc_cs_t *VV = cs->vertvisc_CSp;
  M6_REQUIRE(!VV || cs->visc, "step_MOM_dyn_split_RK2: vertvisc_CSp needs the visc argument (cs->visc)");

  // the step's automatic arrays (:336-369): one grow-only block in the context's pool
  // (its own buffer: the modules called below hand out the pool's buffers from the start in every call)
  const size_t blk_bytes = 3 * sz.u3 + 3 * sz.v3 + sz.h3 + sz.h2;
  const bool fresh = ctx->rk2_scratch.bytes < blk_bytes || ctx->rk2_scratch_layout != 1;
  M6_REQUIRE(ctx->rk2_scratch.reserve(blk_bytes) == 0, "step_MOM_dyn_split_RK2: out of device memory");
  ctx->rk2_scratch_layout = 1;
  char *blk = (char *)ctx->rk2_scratch.p;
  double *up = (double *)blk, *u_bc = (double *)(blk + sz.u3), *uh_in = (double *)(blk + 2 * sz.u3);
  double *vp = (double *)(blk + 3 * sz.u3), *v_bc = (double *)(blk + 3 * sz.u3 + sz.v3), *vh_in = (double *)(blk + 3 * sz.u3 + 2 * sz.v3);
  double *hp = (double *)(blk + 3 * sz.u3 + 3 * sz.v3), *eta_pred = (double *)(blk + 3 * sz.u3 + 3 * sz.v3 + sz.h3);
  double *u_av = cs->u_av, *v_av = cs->v_av, *h_av = cs->h_av, *eta = cs->eta;

  // up = vp = 0, hp = h (:419-422).  The zeros only matter where nothing writes afterwards: the halo faces beyond a
  // closed edge (every other point of up, vp is recomputed or refilled by pass_uvp each step; u_bc_accel, uh_in and
  // eta_pred are read only where they are written), so the block is zeroed when it is allocated, not every step.
  if (fresh) M6_HIP(hipMemsetAsync(blk, 0, blk_bytes, s));
  // hp = h (:422).  The continuity calls at :634 and :757 write every cell of the compute domain (continuity.hip: the convergence of
  // the last direction covers is..ie, js..je) and nothing reads hp before them, so only the frame outside the compute domain is copied:
  // what the calls and pass_hp_uv do not overwrite there keeps h, as in the reference.
  static const bool hp_whole = getenv("MOM6HIP_HP_COPY_WHOLE") && atoi(getenv("MOM6HIP_HP_COPY_WHOLE")) == 1;
  if (hp_whole) {
    M6_HIP(hipMemcpyAsync(hp, h, sz.h3, hipMemcpyDeviceToDevice, s));
  } else {
    auto copy_rect = [&](int i0, int i1, int j0, int j1) {
      launch3d(s, i0, i1, j0, j1, nz, [=] __device__(int i, int j, int k) { hp[g.h3(i, j, k)] = h[g.h3(i, j, k)]; });
    };
    copy_rect(g.isd, g.ied, g.jsd, js - 1); copy_rect(g.isd, g.ied, je + 1, g.jed);
    copy_rect(g.isd, is - 1, js, je); copy_rect(ie + 1, g.ied, js, je);
  }

  // u_bc_accel = (CAu_pred + PFu) + diffu (:557-564) is formed by the kernel that produces the later of its terms: pgf_face_kernel when
  // CAu_pred was stored by the step before, coradcalc_kernel otherwise (mom6hip_ctx::BcAccelFuse); the sweep below runs only when
  // neither took it (another form of PressureForce) or when the first up, vp are wanted from it (hooks)
  const bool inviscid = (cs->hooks == nullptr) && (cs->hor_visc == nullptr);      // diffu = diffv = 0
  const bool vv_fused = VV && !(hk && (hk->visc_remnant_pred || hk->vertvisc));
  const bool need_up1 = (!inviscid || VV) && !vv_fused;
  mom6hip_ctx::BcAccelFuse fuse1{nullptr, nullptr, cs->diffu, cs->diffv, u_bc, v_bc, inviscid ? 1 : 0, false};
  static const bool fuse_off = getenv("MOM6HIP_BC_FUSE") && atoi(getenv("MOM6HIP_BC_FUSE")) == 0;
  const bool try_fuse1 = !need_up1 && !fuse_off;
  // PressureForce :495
  if (try_fuse1 && cs->CAu_pred_stored) { fuse1.au = cs->CAu_pred; fuse1.av = cs->CAv_pred; ctx->bc_fuse = &fuse1; }
  const int rc_pf = mom6hip_pressureforce_fv_bouss(ctx, cs->PressureForce_CSp, cs->eqn_of_state, h, T, S, step_p_surf(cs), cs->PFu, cs->PFv,
                                                   cs->pbce, cs->eta_PF, D);
  ctx->bc_fuse = nullptr;
  if (rc_pf) return rc_pf;
  double *eta_PF_start = nullptr;
  CALL(step_eta_PF_start(ctx, cs, &eta_PF_start));                                                   // :497-503
  if (!cs->CAu_pred_stored) {   // :544-552
    if (try_fuse1) { fuse1.au = cs->PFu; fuse1.av = cs->PFv; ctx->bc_fuse = &fuse1; }
    const int rc_ca = mom6hip_coradcalc(ctx, cs->CoriolisAdv, u_av, v_av, h_av, uh, vh, cs->CAu_pred, cs->CAv_pred, D);
    ctx->bc_fuse = nullptr;
    if (rc_ca) return rc_ca;
  }

  // u_bc_accel = (CAu_pred + PFu) + diffu ; up = mask*(u + dt*u_bc_accel)   :557-564, :582-589
  // Without viscosity hooks diffu = diffv = +0.0 everywhere (set by dyn_split_rk2_init): (a + 0.0) is a, except that
  // -0.0 + 0.0 = +0.0, so the array need not be read; and the first up, vp (:582-589) are only read by vertvisc_coef.
  // (inviscid, vv_fused above) The library's own vertical viscosity forms the velocity increments of :582-589, :667-676 and :930-939
  // inside its coefficient sweep (m6::vertvisc_step_inc: the same expression on the same numbers), so the step's sweeps for them are
  // not launched.
  auto bc_accel = [&](const double *CAu, const double *CAv, bool first_up) {
    const double *PFu = cs->PFu, *PFv = cs->PFv, *diffu = cs->diffu, *diffv = cs->diffv;
    const bool need_up = first_up && (!inviscid || VV) && !vv_fused;
    launch3d(s, Isq, Ieq, js, je, nz, [=] __device__(int I, int j, int k) {
      const long n = g.u3(I, j, k);
      double a = (CAu[n] + PFu[n]);
      if (inviscid) a = (a == 0.0) ? 0.0 : a; else a = a + diffu[n];
      u_bc[n] = a;
      if (need_up) up[n] = g.mask2dCu[g.u2(I, j)] * (u_inst[n] + dt * a);
    });
    launch3d(s, is, ie, Jsq, Jeq, nz, [=] __device__(int i, int J, int k) {
      const long n = g.v3(i, J, k);
      double a = (CAv[n] + PFv[n]);
      if (inviscid) a = (a == 0.0) ? 0.0 : a; else a = a + diffv[n];
      v_bc[n] = a;
      if (need_up) vp[n] = g.mask2dCv[g.v2(i, J)] * (v_inst[n] + dt * a);
    });
  };
  if (!fuse1.done) bc_accel(cs->CAu_pred, cs->CAv_pred, true);
  if (hk && hk->visc_remnant_pred) {   // set_viscous_ML, vertvisc_coef, vertvisc_remnant :592-600
    M6_HIP(hipStreamSynchronize(s));
    M6_REQUIRE(hk->visc_remnant_pred(hk->user, up, vp, h, dt, cs->visc_rem_u, cs->visc_rem_v) == 0, "visc_remnant_pred hook failed");
  } else if (VV) {                     // set_viscous_ML :592 (DYNAMIC_VISCOUS_ML), vertvisc_coef, vertvisc_remnant :598-600
    if (cs->set_visc_CSp && cs->set_visc_CSp->dynamic_viscous_ML) {
      M6_REQUIRE(cs->visc->ustar && cs->visc->nkml_visc_u && cs->visc->nkml_visc_v,
                 "step_MOM_dyn_split_RK2: DYNAMIC_VISCOUS_ML needs forces%%ustar (visc->ustar) and visc%%nkml_visc_u / nkml_visc_v");
      CALL(m6::set_viscous_ML_dev(ctx, cs->set_visc_CSp, u_inst, v_inst, h, T, S, cs->eqn_of_state, taux, tauy, cs->visc->ustar,
                                  (double *)cs->visc->nkml_visc_u, (double *)cs->visc->nkml_visc_v, dt));
    }
    const m6::VelIncrement inc1{u_inst, v_inst, u_bc, v_bc, nullptr, nullptr, dt};      // up = mask * (u + dt * u_bc_accel) :582-589
    CALL(m6::vertvisc_step_inc(ctx, VV, up, vp, h, nullptr, nullptr, nullptr, cs->visc, dt, 0, nullptr, nullptr, cs->visc_rem_u, cs->visc_rem_v,
                               vv_fused ? &inc1 : nullptr, D));
  }
  // pass_eta, pass_visc_rem :541 / :607-611 / :631: in flight behind btcalc and bt_mass_source, which read no halo (the reference
  // completes pass_visc_rem at :631 for the same reason: the continuity below forms fluxes in the rows of visc_rem's halo)
  CALL(pass_start(ctx, {{eta, PH | P2D}, {cs->visc_rem_u, PUs}, {cs->visc_rem_v, PVs}}, nz, 0));

  // btcalc, bt_mass_source :627-630 ; continuity for BT_cont and the layer fluxes :634-644
  if (!BT_cont_BT_thick) CALL(mom6hip_btcalc(ctx, BT, h, nullptr, nullptr, 0, D));
  CALL(mom6hip_bt_mass_source(ctx, BT, h, eta, 1, D));
  if (BTC || cs->BT_use_layer_fluxes) {
    // (this call is made for uh_in, vh_in and BT_cont: the thicknesses it would leave in hp are rewritten by the call at :757 before
    // anything reads them, so the convergence of its second direction is not launched)
    ctx->cont_fluxes_only = true;
    const int rc_c = continuity_around_pass(ctx, [&]() -> int {
      return mom6hip_continuity(ctx, cs->continuity_CSp, u_inst, v_inst, h, hp, uh_in, vh_in, dt, nullptr, nullptr, cs->visc_rem_u,
                                cs->visc_rem_v, nullptr, nullptr, BTC, nullptr, nullptr, D);
    });
    ctx->cont_fluxes_only = false;
    if (rc_c) return rc_c;
    if (BT_cont_BT_thick) CALL(mom6hip_btcalc(ctx, BT, h, BTC->h_u, BTC->h_v, 0, D));
  } else {
    CALL(m6::complete_group_pass(ctx));
  }
  if (calc_dtbt) CALL(mom6hip_set_dtbt_eta(ctx, BT, eta, cs->pbce, nullptr, 0.0, 0.0, D));                       // :651
  const bool lf = cs->BT_use_layer_fluxes != 0;
  CALL(mom6hip_btstep(ctx, BT, u_inst, v_inst, eta, dt, u_bc, v_bc, taux, tauy, RZ_to_H, cs->pbce, cs->eta_PF, u_av, v_av,   // :655
                      cs->u_accel_bt, cs->v_accel_bt, eta_pred, cs->uhbt, cs->vhbt, cs->visc_rem_u, cs->visc_rem_v, BTC, eta_PF_start, nullptr,
                      nullptr, lf ? uh_in : nullptr, lf ? vh_in : nullptr, lf ? u_inst : nullptr, lf ? v_inst : nullptr, nullptr, D));

  // up = u + dt_pred*(u_bc_accel + u_accel_bt) :663-676
  const double dt_pred = dt * cs->be;
  if (!vv_fused) {
    const double *abu = cs->u_accel_bt, *abv = cs->v_accel_bt;
    launch3d(s, is, ie, Jsq, Jeq, nz, [=] __device__(int i, int J, int k) {
      const long n = g.v3(i, J, k);
      vp[n] = g.mask2dCv[g.v2(i, J)] * (v_inst[n] + dt_pred * (v_bc[n] + abv[n]));
    });
    launch3d(s, Isq, Ieq, js, je, nz, [=] __device__(int I, int j, int k) {
      const long n = g.u3(I, j, k);
      up[n] = g.mask2dCu[g.u2(I, j)] * (u_inst[n] + dt_pred * (u_bc[n] + abu[n]));
    });
  }
  if (hk && hk->vertvisc) {   // vertvisc_coef, vertvisc, vertvisc_remnant :717-744
    M6_HIP(hipStreamSynchronize(s));
    M6_REQUIRE(hk->vertvisc(hk->user, up, vp, h, dt_pred, cs->visc_rem_u, cs->visc_rem_v) == 0, "vertvisc hook failed");
  } else if (VV) {            // :717-744, with the increment of :667-676 formed in the coefficient sweep
    const m6::VelIncrement inc2{u_inst, v_inst, u_bc, v_bc, cs->u_accel_bt, cs->v_accel_bt, dt_pred};
    CALL(m6::vertvisc_step_inc(ctx, VV, up, vp, h, nullptr, taux, tauy, cs->visc, dt_pred, 1, nullptr, nullptr, cs->visc_rem_u, cs->visc_rem_v,
                               vv_fused ? &inc2 : nullptr, D));
  }
  // pass_visc_rem, pass_uvp :741-751 in flight behind the continuity's own rows (continuity_around_pass)
  CALL(pass_start(ctx, {{cs->visc_rem_u, PUs}, {cs->visc_rem_v, PVs}, {up, PU}, {vp, PV}}, nz, 1));
  CALL(continuity_around_pass(ctx, [&]() -> int {
    return mom6hip_continuity(ctx, cs->continuity_CSp, up, vp, h, hp, uh, vh, dt, cs->uhbt, cs->vhbt, cs->visc_rem_u, cs->visc_rem_v,   // :757
                              u_av, v_av, BTC, nullptr, nullptr, D);
  }));
  // pass_hp_uv :763 in flight behind bt_mass_source and btcalc (no halos) and behind the rows of h_av, horizontal_viscosity and
  // CorAdCalc that lie at least a halo width inside the tile; the rows along the two edges follow the completion
  const bool hv_hook = !cs->hor_visc && hk && hk->horizontal_viscosity;
  auto after_hp_uv = [&]() -> int {      // (pure functions of their inputs, row by row: a row computed twice gets the same bits)
    const m6::GridDev w = ctx->g;      // the window
    launch3d(s, is - 2, ie + 2, w.jsc - 2, w.jec + 2, nz, [=] __device__(int i, int j, int k) {     // :785-787
      const long n = g.h3(i, j, k);
      h_av[n] = 0.5 * (h[n] + hp[n]);
    });
    if (cs->hor_visc) {   // :860 (hu_cont, hv_cont = BT_cont%h_u, %h_v: read only with USE_CONT_THICKNESS)
      if (m6::horizontal_viscosity_dev(ctx, cs->hor_visc, u_av, v_av, h_av, cs->diffu, cs->diffv, BTC ? BTC->h_u : nullptr,
                                       BTC ? BTC->h_v : nullptr)) return 1;
    }
    if (!hv_hook) CALL(mom6hip_coradcalc(ctx, cs->CoriolisAdv, u_av, v_av, h_av, uh, vh, cs->CAu, cs->CAv, D));   // :869
    return 0;
  };
  CALL(pass_start(ctx, {{hp, PH}, {u_av, PU}, {v_av, PV}, {uh, PU}, {vh, PV}}, nz, 2));             // :763
  CALL(mom6hip_bt_mass_source(ctx, BT, hp, eta_pred, 0, D));                                           // :797
  if (BT_cont_BT_thick) CALL(mom6hip_btcalc(ctx, BT, h, BTC->h_u, BTC->h_v, 0, D));                    // :843
  // u_bc_accel = (CAu + PFu) + diffu (:879-886) formed by coradcalc_kernel, row window by row window (diffu of a row is complete
  // before CorAdCalc of that row is launched)
  mom6hip_ctx::BcAccelFuse fuse2{cs->PFu, cs->PFv, cs->diffu, cs->diffv, u_bc, v_bc, inviscid ? 1 : 0, false};
  if (!fuse_off) ctx->bc_fuse = &fuse2;
  int rc_ap = around_pass(ctx, after_hp_uv);
  if (rc_ap == 0 && hv_hook) {
    rc_ap = hipStreamSynchronize(s) == hipSuccess ? 0 : 1;
    if (rc_ap == 0 && hk->horizontal_viscosity(hk->user, u_av, v_av, h_av, cs->diffu, cs->diffv) != 0) {
      m6::set_error("horizontal_viscosity hook failed"); rc_ap = 1;
    }
    if (rc_ap == 0) rc_ap = mom6hip_coradcalc(ctx, cs->CoriolisAdv, u_av, v_av, h_av, uh, vh, cs->CAu, cs->CAv, D);   // :869
  }
  ctx->bc_fuse = nullptr;
  if (rc_ap) return rc_ap;
  if (!fuse2.done) bc_accel(cs->CAu, cs->CAv, false);                                                 // :879-886
  CALL(mom6hip_btstep(ctx, BT, u_inst, v_inst, eta, dt, u_bc, v_bc, taux, tauy, RZ_to_H, cs->pbce, cs->eta_PF, u_av, v_av,   // :911
                      cs->u_accel_bt, cs->v_accel_bt, eta_pred, cs->uhbt, cs->vhbt, cs->visc_rem_u, cs->visc_rem_v, BTC, eta_PF_start, nullptr,
                      nullptr, lf ? uh : nullptr, lf ? vh : nullptr, lf ? u_av : nullptr, lf ? v_av : nullptr, eta_av, D));
  launch3d(s, is, ie, js, je, 1, [=] __device__(int i, int j, int) { eta[g.h2(i, j)] = eta_pred[g.h2(i, j)]; });   // :918
  if (!vv_fused) {   // u = u + dt*(u_bc_accel + u_accel_bt) :928-939
    const double *abu = cs->u_accel_bt, *abv = cs->v_accel_bt;
    launch3d(s, Isq, Ieq, js, je, nz, [=] __device__(int I, int j, int k) {
      const long n = g.u3(I, j, k);
      u_inst[n] = g.mask2dCu[g.u2(I, j)] * (u_inst[n] + dt * (u_bc[n] + abu[n]));
    });
    launch3d(s, is, ie, Jsq, Jeq, nz, [=] __device__(int i, int J, int k) {
      const long n = g.v3(i, J, k);
      v_inst[n] = g.mask2dCv[g.v2(i, J)] * (v_inst[n] + dt * (v_bc[n] + abv[n]));
    });
  }
  if (hk && hk->vertvisc) {   // :974-994
    M6_HIP(hipStreamSynchronize(s));
    M6_REQUIRE(hk->vertvisc(hk->user, u_inst, v_inst, h, dt, cs->visc_rem_u, cs->visc_rem_v) == 0, "vertvisc hook failed");
  } else if (VV) {            // :974-994, with the increment of :930-939 formed in place in the coefficient sweep
    const m6::VelIncrement inc3{u_inst, v_inst, u_bc, v_bc, cs->u_accel_bt, cs->v_accel_bt, dt};
    CALL(m6::vertvisc_step_inc(ctx, VV, u_inst, v_inst, h, nullptr, taux, tauy, cs->visc, dt, 1, nullptr, nullptr, cs->visc_rem_u, cs->visc_rem_v,
                               vv_fused ? &inc3 : nullptr, D));
  }
  launch3d(s, is - 2, ie + 2, js - 2, je + 2, nz, [=] __device__(int i, int j, int k) { h_av[g.h3(i, j, k)] = h[g.h3(i, j, k)]; });   // :1000
  CALL(pass_start(ctx, {{cs->visc_rem_u, PUs}, {cs->visc_rem_v, PVs}, {u_inst, PU}, {v_inst, PV}}, nz, 3));     // :991-1008
  CALL(continuity_around_pass(ctx, [&]() -> int {
    return mom6hip_continuity(ctx, cs->continuity_CSp, u_inst, v_inst, h, h, uh, vh, dt, cs->uhbt, cs->vhbt, cs->visc_rem_u,      // :1015
                              cs->visc_rem_v, u_av, v_av, nullptr, nullptr, nullptr, D);
  }));
  // pass_h, pass_av_uvh :1018-1043 in flight behind the rows of the three accumulations and of CorAdCalc that need no halo row.
  // The accumulations work in place: their rows are split exactly (inner rows before the completion, the two edge bands after).
  auto accumulate = [&](int ja, int jb, int Ja, int Jb) {      // cell rows ja..jb, v-face rows Ja..Jb
    launch3d(s, is - 2, ie + 2, ja, jb, nz, [=] __device__(int i, int j, int k) {                      // :1038-1040
      const long n = g.h3(i, j, k);
      h_av[n] = 0.5 * (h_av[n] + h[n]);
    });
    launch3d(s, Isq - 2, Ieq + 2, ja, jb, nz, [=] __device__(int I, int j, int k) {                    // :1046-1053
      const long n = g.u3(I, j, k);
      uhtr[n] = uhtr[n] + uh[n] * dt;
    });
    launch3d(s, is - 2, ie + 2, Ja, Jb, nz, [=] __device__(int i, int J, int k) {
      const long n = g.v3(i, J, k);
      vhtr[n] = vhtr[n] + vh[n] * dt;
    });
  };
  auto next_CA = [&]() -> int {
    if (cs->store_CAu) CALL(mom6hip_coradcalc(ctx, cs->CoriolisAdv, u_av, v_av, h_av, uh, vh, cs->CAu_pred, cs->CAv_pred, D));   // :1055-1069
    return 0;
  };
  CALL(pass_start(ctx, {{h, PH}, {u_av, PU}, {v_av, PV}, {uh, PU}, {vh, PV}}, nz, 4));               // :1018, :1027
  if (split_rows(ctx)) {
    ctx->overlap[0]++;
    const int W = is - g.isd;      // the halo width: CorAdCalc's rows at least W inside the tile read rows of the compute domain only
    accumulate(js + 2, je - 2, Jsq + 2, Jeq - 2);      // (h_av's inner rows reach two rows beyond the window of next_CA below)
    m6::row_window(ctx, js + W + 2, je - W - 2);
    int rc = next_CA();
    m6::row_window_reset(ctx);
    if (rc) { m6::ErrorKeeper keep; m6::complete_group_pass(ctx); return rc; }
    CALL(m6::complete_group_pass(ctx));
    accumulate(js - 2, js + 1, Jsq - 2, Jsq + 1);
    accumulate(je - 1, je + 2, Jeq - 1, Jeq + 2);
    m6::row_window(ctx, js, js + W + 1);
    rc = next_CA();
    if (!rc) { m6::row_window(ctx, je - W - 1, je); rc = next_CA(); }
    m6::row_window_reset(ctx);
    if (rc) return rc;
  } else {
    ctx->overlap[2]++;
    CALL(m6::complete_group_pass(ctx));
    accumulate(js - 2, je + 2, Jsq - 2, Jeq + 2);
    CALL(next_CA());
  }
  cs->CAu_pred_stored = cs->store_CAu ? 1 : 0;
  M6_HIP(hipGetLastError());
  return 0;
}

// ---- SPLIT_RK2B: src/core/MOM_dynamics_split_RK2b.F90 ----------------------------------------------------------
// The same operators in another order: the step starts from the filtered velocities u_av, v_av (the model's prognostic
// velocities in this scheme), does a first continuity + CorAdCalc + horizontal_viscosity with them, rebuilds the
// instantaneous velocities from the stored barotropic increments (:641-646) and ends with the increments the final
// continuity returns (du_cor, dv_cor :979-981).  cs->u_av, cs->v_av, cs->h_av are the step's u_inst, v_inst, h_av.

int mom6hip_dyn_split_rk2b_init(mom6hip_ctx_t *ctx, mom6hip_dyn_split_rk2_cs_t *cs, const double *h) {
  M6_REQUIRE(ctx && h, "dyn_split_rk2b_init: null argument");
  CALL(check(cs, "dyn_split_rk2b_init"));
  M6_REQUIRE(cs->du_av_inst && cs->dv_av_inst, "dyn_split_rk2b_init: du_av_inst / dv_av_inst are not allocated");
  const m6::GridDev g = ctx->g;
  const Sz sz = sizes(g);
  hipStream_t s = ctx->stream;
  {   // eta :1406-1420
    double *eta = cs->eta;
    const double Z_to_H = g.Z_to_H;
    const long hstr = (long)g.nih * g.njh;
    const int nz = g.nk;
    launch3d(s, g.isc, g.iec, g.jsc, g.jec, 1, [=] __device__(int i, int j, int) {
      const long n = g.h2(i, j);
      double e = -Z_to_H * g.bathyT[n];
      for (int k = 0; k < nz; k++) e = e + h[n + hstr * k];
      eta[n] = e;
    });
  }
  M6_HIP(hipMemsetAsync(cs->diffu, 0, sz.u3, s)); M6_HIP(hipMemsetAsync(cs->diffv, 0, sz.v3, s));       // :1155-1156
  M6_HIP(hipMemsetAsync(cs->du_av_inst, 0, sz.u3 / g.nk, s)); M6_HIP(hipMemsetAsync(cs->dv_av_inst, 0, sz.v3 / g.nk, s));   // :1164-1165
  M6_HIP(hipMemsetAsync(cs->u_av, 0, sz.u3, s)); M6_HIP(hipMemsetAsync(cs->v_av, 0, sz.v3, s));         // u_inst = v_inst = 0 :404
  {
    double *vru = cs->visc_rem_u, *vrv = cs->visc_rem_v;
    launch3d(s, g.isd - 1, g.ied, g.jsd, g.jed, g.nk, [=] __device__(int i, int j, int k) { vru[g.u3(i, j, k)] = 1.0; });
    launch3d(s, g.isd, g.ied, g.jsd - 1, g.jed, g.nk, [=] __device__(int i, int j, int k) { vrv[g.v3(i, j, k)] = 1.0; });
  }
  M6_HIP(hipGetLastError());
  return 0;
}

int mom6hip_step_dyn_split_rk2b(mom6hip_ctx_t *ctx, mom6hip_dyn_split_rk2_cs_t *cs, double *u_av, double *v_av, double *h,
                                const double *T, const double *S, double dt, const double *taux, const double *tauy, double RZ_to_H,
                                double *uh, double *vh, double *uhtr, double *vhtr, double *eta_av, int32_t calc_dtbt) {
  M6_REQUIRE(ctx && u_av && v_av && h && taux && tauy && uh && vh && uhtr && vhtr && eta_av, "step_MOM_dyn_split_RK2b: null argument");
  CALL(check(cs, "step_MOM_dyn_split_RK2b"));
  M6_REQUIRE(!cs->eqn_of_state || (T && S), "step_MOM_dyn_split_RK2b: an equation of state needs tv%%T and tv%%S");
  M6_REQUIRE(cs->du_av_inst && cs->dv_av_inst, "step_MOM_dyn_split_RK2b: du_av_inst / dv_av_inst are not allocated");
  // CS%OBC: the OBC entry points of the operators, open_boundary_zero_normal_flow on the accelerations (:571-573, :866-868) and
  // radiation_open_bdry_conds on u_av (:766-774, :1000-1002), as in step_with_obc
  const mom6hip_obc_t *OBC = cs->OBC;
  M6_REQUIRE(!OBC || !cs->hooks, "step_MOM_dyn_split_RK2b: host-side parameterisations (hooks) are not provided with an associated OBC");
  static const bool obc_one_tile = getenv("MOM6HIP_OBC_ONE_TILE") && atoi(getenv("MOM6HIP_OBC_ONE_TILE")) == 1;
  M6_REQUIRE(!OBC || !obc_one_tile || !m6::multi_tile(ctx), "step_MOM_dyn_split_RK2b: an associated OBC is provided on one tile (MOM6HIP_OBC_ONE_TILE)");
  const m6::GridDev g = ctx->g;
  const Sz sz = sizes(g);
  hipStream_t s = ctx->stream;
  const int D = MOM6HIP_MEM_DEVICE;
  const int is = g.isc, ie = g.iec, js = g.jsc, je = g.jec, nz = g.nk;
  const int Isq = is - 1, Ieq = ie, Jsq = js - 1, Jeq = je;
  mom6hip_barotropic_cs_t *BT = cs->barotropic_CSp;
  const mom6hip_bt_cont_t *BTC = cs->BT_cont;
  const bool BT_cont_BT_thick = BTC && BTC->h_u && BTC->h_v;
  const mom6hip_visc_hooks_t *hk = cs->hooks;
  mom6hip_vertvisc_cs_t *VV = cs->vertvisc_CSp;
  M6_REQUIRE(!VV || cs->visc, "step_MOM_dyn_split_RK2b: vertvisc_CSp needs the visc argument (cs->visc)");

  const size_t blk_bytes = 3 * sz.u3 + 3 * sz.v3 + sz.h3 + sz.h2 + (OBC ? sz.u3 + sz.v3 : 0);
  // (the block is shared with step_MOM_dyn_split_RK2, whose layout differs: after the other stepper the faces that rely on
  // being zero hold other arrays' values, so a change of stepper zeroes the block like a new allocation)
  const bool fresh = ctx->rk2_scratch.bytes < blk_bytes || ctx->rk2_scratch_layout != 2;
  M6_REQUIRE(ctx->rk2_scratch.reserve(blk_bytes) == 0, "step_MOM_dyn_split_RK2b: out of device memory");
  ctx->rk2_scratch_layout = 2;      // (with an OBC: u_old_rad_OBC, v_old_rad_OBC behind the same layout)
  char *blk = (char *)ctx->rk2_scratch.p;
  double *up = (double *)blk, *u_bc = (double *)(blk + sz.u3), *uh_in = (double *)(blk + 2 * sz.u3);
  double *vp = (double *)(blk + 3 * sz.u3), *v_bc = (double *)(blk + 3 * sz.u3 + sz.v3), *vh_in = (double *)(blk + 3 * sz.u3 + 2 * sz.v3);
  double *hp = (double *)(blk + 3 * sz.u3 + 3 * sz.v3), *eta_pred = (double *)(blk + 3 * sz.u3 + 3 * sz.v3 + sz.h3);
  double *u_inst = cs->u_av, *v_inst = cs->v_av, *h_av = cs->h_av, *eta = cs->eta;
  // up = vp = u_inst = v_inst = 0 (:404) matters only at the halo faces beyond a closed edge, which nothing writes:
  // zeroed when allocated (the block) and by dyn_split_rk2b_init (u_inst, v_inst).
  if (fresh) M6_HIP(hipMemsetAsync(blk, 0, blk_bytes, s));
  M6_HIP(hipMemcpyAsync(hp, h, sz.h3, hipMemcpyDeviceToDevice, s));                                  // :403
  double *u_old = nullptr, *v_old = nullptr;
  if (OBC) {                                                                                          // :436-442
    u_old = (double *)(blk + 3 * sz.u3 + 3 * sz.v3 + sz.h3 + sz.h2); v_old = (double *)((char *)u_old + sz.u3);
    M6_HIP(hipMemcpyAsync(u_old, u_av, sz.u3, hipMemcpyDeviceToDevice, s));
    M6_HIP(hipMemcpyAsync(v_old, v_av, sz.v3, hipMemcpyDeviceToDevice, s));
  }
  // the operators, with the OBC where one is associated
  auto coradcalc = [&](double *CAu, double *CAv) -> int {
    return OBC ? mom6hip_coradcalc_obc(ctx, cs->CoriolisAdv, OBC, u_av, v_av, cs->h_av, uh, vh, CAu, CAv, D)
               : mom6hip_coradcalc(ctx, cs->CoriolisAdv, u_av, v_av, cs->h_av, uh, vh, CAu, CAv, D);
  };
  auto btcalc = [&](const double *hu, const double *hv) -> int {
    return OBC ? mom6hip_btcalc_obc(ctx, BT, h, hu, hv, 0, OBC, D) : mom6hip_btcalc(ctx, BT, h, hu, hv, 0, D);
  };
  auto vertvisc_step = [&](double *uu, double *vv, double dtx, int update) -> int {
    if (!OBC) return mom6hip_vertvisc_step(ctx, VV, uu, vv, h, nullptr, update ? taux : nullptr, update ? tauy : nullptr, cs->visc, dtx, update, nullptr,
                                           nullptr, cs->visc_rem_u, cs->visc_rem_v, D);
    CALL(mom6hip_vertvisc_coef_obc(ctx, VV, uu, vv, h, nullptr, cs->visc, dtx, OBC, D));
    if (update) CALL(mom6hip_vertvisc_obc(ctx, VV, uu, vv, h, taux, tauy, cs->visc, dtx, nullptr, nullptr, OBC, D));
    return mom6hip_vertvisc_remnant(ctx, VV, cs->visc, cs->visc_rem_u, cs->visc_rem_v, dtx, D);
  };

  // continuity with the filtered velocities :488, PressureForce :498, pass_hp_uhvh :535, h_av :540-542
  CALL(mom6hip_continuity_obc(ctx, cs->continuity_CSp, OBC, u_av, v_av, h, hp, uh, vh, dt, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                              nullptr, nullptr, nullptr, D));
  CALL(mom6hip_pressureforce_fv_bouss(ctx, cs->PressureForce_CSp, cs->eqn_of_state, h, T, S, step_p_surf(cs), cs->PFu, cs->PFv, cs->pbce,
                                      cs->eta_PF, D));
  double *eta_PF_start = nullptr;
  CALL(step_eta_PF_start(ctx, cs, &eta_PF_start));                                                   // :497-503
  CALL(pass(ctx, {{hp, PH}, {uh, PU}, {vh, PV}}, nz));
  auto set_h_av = [&]() {
    launch3d(s, is - 2, ie + 2, js - 2, je + 2, nz, [=] __device__(int i, int j, int k) {
      const long n = g.h3(i, j, k);
      h_av[n] = 0.5 * (h[n] + hp[n]);
    });
  };
  set_h_av();
  auto hor_visc = [&]() -> int {
    if (cs->hor_visc && OBC) {
      CALL(mom6hip_horizontal_viscosity_obc(ctx, cs->hor_visc, u_av, v_av, h_av, cs->diffu, cs->diffv, dt, BTC ? BTC->h_u : nullptr,
                                            BTC ? BTC->h_v : nullptr, OBC, D));
    } else if (cs->hor_visc) {
      if (m6::horizontal_viscosity_dev(ctx, cs->hor_visc, u_av, v_av, h_av, cs->diffu, cs->diffv, BTC ? BTC->h_u : nullptr,
                                       BTC ? BTC->h_v : nullptr)) return 1;
    } else if (hk && hk->horizontal_viscosity) {
      M6_HIP(hipStreamSynchronize(s));
      M6_REQUIRE(hk->horizontal_viscosity(hk->user, u_av, v_av, h_av, cs->diffu, cs->diffv) == 0, "horizontal_viscosity hook failed");
    }
    return 0;
  };
  CALL(coradcalc(cs->CAu_pred, cs->CAv_pred));                                                            // :548
  CALL(hor_visc());                                                                                          // :555

  // u_bc_accel :561-568 ; up = mask*(u_av + dt*u_bc_accel) :587-594 (read only by vertvisc_coef)
  auto bc_accel = [&](const double *CAu, const double *CAv, bool first_up) {
    const double *PFu = cs->PFu, *PFv = cs->PFv, *diffu = cs->diffu, *diffv = cs->diffv;
    const bool need_up = first_up && (VV || (hk && hk->visc_remnant_pred)) && !OBC;      // (with an OBC: after the accelerations are zeroed on the segments)
    launch3d(s, Isq, Ieq, js, je, nz, [=] __device__(int I, int j, int k) {
      const long n = g.u3(I, j, k);
      const double a = (CAu[n] + PFu[n]) + diffu[n];
      u_bc[n] = a;
      if (need_up) up[n] = g.mask2dCu[g.u2(I, j)] * (u_av[n] + dt * a);
    });
    launch3d(s, is, ie, Jsq, Jeq, nz, [=] __device__(int i, int J, int k) {
      const long n = g.v3(i, J, k);
      const double a = (CAv[n] + PFv[n]) + diffv[n];
      v_bc[n] = a;
      if (need_up) vp[n] = g.mask2dCv[g.v2(i, J)] * (v_av[n] + dt * a);
    });
  };
  bc_accel(cs->CAu_pred, cs->CAv_pred, true);
  if (OBC) {
    CALL(mom6hip_open_boundary_zero_normal_flow(ctx, OBC, u_bc, v_bc, D));                            // :571-573
    if (VV) {                                                                                         // :587-594
      launch3d(s, Isq, Ieq, js, je, nz, [=] __device__(int I, int j, int k) {
        const long n = g.u3(I, j, k);
        up[n] = g.mask2dCu[g.u2(I, j)] * (u_av[n] + dt * u_bc[n]);
      });
      launch3d(s, is, ie, Jsq, Jeq, nz, [=] __device__(int i, int J, int k) {
        const long n = g.v3(i, J, k);
        vp[n] = g.mask2dCv[g.v2(i, J)] * (v_av[n] + dt * v_bc[n]);
      });
    }
  }
  if (hk && hk->visc_remnant_pred) {   // set_viscous_ML, vertvisc_coef, vertvisc_remnant :598-606
    M6_HIP(hipStreamSynchronize(s));
    M6_REQUIRE(hk->visc_remnant_pred(hk->user, up, vp, h, dt, cs->visc_rem_u, cs->visc_rem_v) == 0, "visc_remnant_pred hook failed");
  } else if (VV) {
    if (cs->set_visc_CSp && cs->set_visc_CSp->dynamic_viscous_ML) {      // set_viscous_ML :598
      M6_REQUIRE(cs->visc->ustar && cs->visc->nkml_visc_u && cs->visc->nkml_visc_v,
                 "step_MOM_dyn_split_RK2b: DYNAMIC_VISCOUS_ML needs forces%%ustar (visc->ustar) and visc%%nkml_visc_u / nkml_visc_v");
      CALL(m6::set_viscous_ML_dev(ctx, cs->set_visc_CSp, u_av, v_av, h, T, S, cs->eqn_of_state, taux, tauy, cs->visc->ustar,
                                  (double *)cs->visc->nkml_visc_u, (double *)cs->visc->nkml_visc_v, dt));
    }
    CALL(vertvisc_step(up, vp, dt, 0));
  }
  CALL(pass(ctx, {{eta, PH | P2D}, {cs->visc_rem_u, PUs}, {cs->visc_rem_v, PVs}}, nz));                 // :616-617
  if (!BT_cont_BT_thick) CALL(btcalc(nullptr, nullptr));                                                 // :623-625
  CALL(mom6hip_bt_mass_source(ctx, BT, h, eta, 1, D));
  {   // the instantaneous velocities :641-646, pass_uv_inst :648
    const double *du = cs->du_av_inst, *dv = cs->dv_av_inst, *vru = cs->visc_rem_u, *vrv = cs->visc_rem_v;
    launch3d(s, Isq, Ieq, js, je, nz, [=] __device__(int I, int j, int k) {
      const long n = g.u3(I, j, k);
      u_inst[n] = u_av[n] - du[g.u2(I, j)] * vru[n];
    });
    launch3d(s, is, ie, Jsq, Jeq, nz, [=] __device__(int i, int J, int k) {
      const long n = g.v3(i, J, k);
      v_inst[n] = v_av[n] - dv[g.v2(i, J)] * vrv[n];
    });
  }
  CALL(pass(ctx, {{u_inst, PU}, {v_inst, PV}}, nz));
  CALL(mom6hip_continuity_obc(ctx, cs->continuity_CSp, OBC, u_inst, v_inst, h, hp, uh_in, vh_in, dt, nullptr, nullptr, cs->visc_rem_u,      // :652
                              cs->visc_rem_v, nullptr, nullptr, BTC, nullptr, nullptr, D));
  if (BT_cont_BT_thick) CALL(btcalc(BTC->h_u, BTC->h_v));                                                // :655-658
  if (calc_dtbt) CALL(mom6hip_set_dtbt_eta(ctx, BT, eta, cs->pbce, nullptr, 0.0, 0.0, D));                       // :664
  CALL(mom6hip_btstep_obc(ctx, BT, u_inst, v_inst, eta, dt, u_bc, v_bc, taux, tauy, RZ_to_H, cs->pbce, cs->eta_PF, u_av, v_av,   // :668
                          cs->u_accel_bt, cs->v_accel_bt, eta_pred, cs->uhbt, cs->vhbt, cs->visc_rem_u, cs->visc_rem_v, BTC, eta_PF_start, nullptr,
                          nullptr, uh_in, vh_in, u_inst, v_inst, nullptr, OBC, D));

  // up = u_inst + dt_pred*(u_bc_accel + u_accel_bt) :675-686
  const double dt_pred = dt * cs->be;
  {
    const double *abu = cs->u_accel_bt, *abv = cs->v_accel_bt;
    launch3d(s, is, ie, Jsq, Jeq, nz, [=] __device__(int i, int J, int k) {
      const long n = g.v3(i, J, k);
      vp[n] = g.mask2dCv[g.v2(i, J)] * (v_inst[n] + dt_pred * (v_bc[n] + abv[n]));
    });
    launch3d(s, Isq, Ieq, js, je, nz, [=] __device__(int I, int j, int k) {
      const long n = g.u3(I, j, k);
      up[n] = g.mask2dCu[g.u2(I, j)] * (u_inst[n] + dt_pred * (u_bc[n] + abu[n]));
    });
  }
  if (hk && hk->vertvisc) {   // vertvisc_coef, vertvisc, vertvisc_remnant :724-745
    M6_HIP(hipStreamSynchronize(s));
    M6_REQUIRE(hk->vertvisc(hk->user, up, vp, h, dt_pred, cs->visc_rem_u, cs->visc_rem_v) == 0, "vertvisc hook failed");
  } else if (VV) {
    CALL(vertvisc_step(up, vp, dt_pred, 1));
  }
  // pass_visc_rem, pass_uvp :748, :752 in flight behind the continuity's own rows, as in the RK2 stepping (continuity_around_pass)
  CALL(pass_start(ctx, {{cs->visc_rem_u, PUs}, {cs->visc_rem_v, PVs}, {up, PU}, {vp, PV}}, nz, 1));
  if (OBC) {
    CALL(m6::complete_group_pass(ctx));
    CALL(mom6hip_continuity_obc(ctx, cs->continuity_CSp, OBC, up, vp, h, hp, uh, vh, dt, cs->uhbt, cs->vhbt, cs->visc_rem_u, cs->visc_rem_v,
                                u_av, v_av, BTC, nullptr, nullptr, D));
  } else
  CALL(continuity_around_pass(ctx, [&]() -> int {
    return mom6hip_continuity(ctx, cs->continuity_CSp, up, vp, h, hp, uh, vh, dt, cs->uhbt, cs->vhbt, cs->visc_rem_u, cs->visc_rem_v,   // :758
                              u_av, v_av, BTC, nullptr, nullptr, D);
  }));
  CALL(pass(ctx, {{hp, PH}, {u_av, PU}, {v_av, PV}, {uh, PU}, {vh, PV}}, nz));                      // :764
  if (OBC) CALL(mom6hip_radiation_open_bdry_conds(ctx, OBC, OBC->gamma_uv, OBC->rx_max, OBC->rx_normal, OBC->ry_normal, u_av, u_old, v_av, v_old,   // :770
                                                  dt_pred, D));
  set_h_av();                                                                                         // :780-782
  CALL(mom6hip_bt_mass_source(ctx, BT, hp, eta_pred, 0, D));                                           // :790
  if (BT_cont_BT_thick) CALL(btcalc(BTC->h_u, BTC->h_v));                                                // :824-827
  CALL(hor_visc());                                                                                   // :841
  CALL(coradcalc(cs->CAu, cs->CAv));                                                                  // :848
  bc_accel(cs->CAu, cs->CAv, false);                                                                  // :854-861
  if (OBC) CALL(mom6hip_open_boundary_zero_normal_flow(ctx, OBC, u_bc, v_bc, D));                     // :866-868
  CALL(mom6hip_btstep_obc(ctx, BT, u_inst, v_inst, eta, dt, u_bc, v_bc, taux, tauy, RZ_to_H, cs->pbce, cs->eta_PF, u_av, v_av,   // :889
                          cs->u_accel_bt, cs->v_accel_bt, eta_pred, cs->uhbt, cs->vhbt, cs->visc_rem_u, cs->visc_rem_v, BTC, eta_PF_start, nullptr,
                          nullptr, uh, vh, u_av, v_av, eta_av, OBC, D));
  launch3d(s, is, ie, js, je, 1, [=] __device__(int i, int j, int) { eta[g.h2(i, j)] = eta_pred[g.h2(i, j)]; });   // :898
  {   // u_inst = u_inst + dt*(u_bc_accel + u_accel_bt) :908-919
    const double *abu = cs->u_accel_bt, *abv = cs->v_accel_bt;
    launch3d(s, Isq, Ieq, js, je, nz, [=] __device__(int I, int j, int k) {
      const long n = g.u3(I, j, k);
      u_inst[n] = g.mask2dCu[g.u2(I, j)] * (u_inst[n] + dt * (u_bc[n] + abu[n]));
    });
    launch3d(s, is, ie, Jsq, Jeq, nz, [=] __device__(int i, int J, int k) {
      const long n = g.v3(i, J, k);
      v_inst[n] = g.mask2dCv[g.v2(i, J)] * (v_inst[n] + dt * (v_bc[n] + abv[n]));
    });
  }
  if (hk && hk->vertvisc) {   // :946-963
    M6_HIP(hipStreamSynchronize(s));
    M6_REQUIRE(hk->vertvisc(hk->user, u_inst, v_inst, h, dt, cs->visc_rem_u, cs->visc_rem_v) == 0, "vertvisc hook failed");
  } else if (VV) {
    CALL(vertvisc_step(u_inst, v_inst, dt, 1));
  }
  CALL(pass_start(ctx, {{cs->visc_rem_u, PUs}, {cs->visc_rem_v, PVs}, {u_inst, PU}, {v_inst, PV}}, nz, 3));     // :967, :971
  if (OBC) {
    CALL(m6::complete_group_pass(ctx));
    CALL(mom6hip_continuity_obc(ctx, cs->continuity_CSp, OBC, u_inst, v_inst, h, h, uh, vh, dt, cs->uhbt, cs->vhbt, cs->visc_rem_u,
                                cs->visc_rem_v, u_av, v_av, nullptr, cs->du_av_inst, cs->dv_av_inst, D));
  } else
  CALL(continuity_around_pass(ctx, [&]() -> int {
    return mom6hip_continuity(ctx, cs->continuity_CSp, u_inst, v_inst, h, h, uh, vh, dt, cs->uhbt, cs->vhbt, cs->visc_rem_u,      // :979
                              cs->visc_rem_v, u_av, v_av, nullptr, cs->du_av_inst, cs->dv_av_inst, D);
  }));
  CALL(pass(ctx, {{h, PH}, {u_av, PU}, {v_av, PV}, {uh, PU}, {vh, PV}}, nz));                        // :993
  if (OBC) CALL(mom6hip_radiation_open_bdry_conds(ctx, OBC, OBC->gamma_uv, OBC->rx_max, OBC->rx_normal, OBC->ry_normal, u_av, u_old, v_av, v_old,   // :1001
                                                  dt, D));
  launch3d(s, Isq - 2, Ieq + 2, js - 2, je + 2, nz, [=] __device__(int I, int j, int k) {            // :1004-1011
    const long n = g.u3(I, j, k);
    uhtr[n] = uhtr[n] + uh[n] * dt;
  });
  launch3d(s, is - 2, ie + 2, Jsq - 2, Jeq + 2, nz, [=] __device__(int i, int J, int k) {
    const long n = g.v3(i, J, k);
    vhtr[n] = vhtr[n] + vh[n] * dt;
  });
  M6_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
