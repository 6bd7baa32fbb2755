// tracer_hor_diff.hip -- tracer_hordiff (src/tracer/MOM_tracer_hor_diff.F90:119-680) as gfx950 kernels: the along-layer diffusion
// with KHTR or the VarMix / MEKE diffusivities, the call of the neutral-diffusion branch (neutral_diffusion.hip), and with
// DIFFUSE_ML_TO_INTERIOR the scaled / skipped variable-density layers (:544-550) and the call of tracer_epipycnal_ML_diff
// (epipycnal_diff.hip).
//
//   hd_khdt_kernel    khdt_x, khdt_y (:340-351), their MAX_TR_DIFFUSION_CFL limit (:368-398) and, with CHECK_DIFFUSIVE_CFL,
//                     the largest diffusive CFL number (:410-416; an atomic max on the bit pattern of positive doubles)
//   hd_step_kernel    one iteration for every tracer of a layer point: the four face coefficients (harmonic-mean
//                     thicknesses, :552-561) and Ihdxdy are formed once per point and serve all tracers; the new values go
//                     to a work copy because the reference updates a layer only after all its dTr are formed (:568-594)
//   hd_commit_kernel  work copy -> tracer on the compute domain, with the concentration underflow (:607-612)
// Every layer and point is independent: thread per point, lanes along i, blockIdx.z = k.
// Algorithmic traffic per iteration: read h, read and write every tracer = 8 + 16 ntr B per cell (the work copy doubles the
// tracer part).
#include <cfloat>
#include <cmath>

#include "common.hpp"

namespace {

using m6::min2;

using m6::max2;

struct HDArgs {
  m6::GridDev g;
  double KhTr, max_diff_CFL, dt, scale;
  // with VarMix%use_variable_mixing (:236-281)
  int use_VarMix, use_Eady, Resoln_scaled;
  double KhTr_Slope_Cff, KhTr_fac, KhTr_min, KhTr_max, pass_coeff, pass_min;
  const double *MEKE_Kh, *L2u, *L2v, *SN_u, *SN_v, *Res_fn_h, *Rd_dx_h;
  const double *h;
  double *khdt_x, *khdt_y;
  double *const *tr;       // device table of ntr tracer pointers
  double *const *work;     // device table of ntr work copies
  const double *cu;        // conc_underflow per tracer (device) or null
  unsigned long long *cfl_bits;
  int ntr, check_cfl;
  int nkml, nkmb;          // CS%Diffuse_ML_interior: GV%nkml, GV%nk_rho_varies (nkmb = 0: not in use)
  double ML_KhTr_scale;
};

// the scale of layer k (0-based) :543-550; false: the layer is left out of the along-layer diffusion
__device__ __forceinline__ bool hd_layer_scale(const HDArgs &A, int k, double &scale) {
  scale = A.scale;
  if (A.nkmb > 0) {
    if (k + 1 <= A.nkml) {
      if (A.ML_KhTr_scale <= 0.0) return false;
      scale = A.scale * A.ML_KhTr_scale;
    }
    if ((k + 1 > A.nkml) && (k + 1 <= A.nkmb)) return false;
  }
  return true;
}

// the diffusivity of a face with variable mixing :236-281 (c0, c1: the cells either side; f: the face)
__device__ __forceinline__ double hd_Kh_face(const HDArgs &A, long c0, long c1, long f, bool dir) {
  double Kh_loc = A.KhTr;
  if (A.use_Eady) Kh_loc = Kh_loc + A.KhTr_Slope_Cff * (dir ? A.L2v[f] : A.L2u[f]) * (dir ? A.SN_v[f] : A.SN_u[f]);
  if (A.MEKE_Kh) Kh_loc = Kh_loc + A.KhTr_fac * sqrt(A.MEKE_Kh[c0] * A.MEKE_Kh[c1]);
  if (A.KhTr_max > 0.) Kh_loc = min2(Kh_loc, A.KhTr_max);
  if (A.Resoln_scaled) Kh_loc = Kh_loc * 0.5 * (A.Res_fn_h[c0] + A.Res_fn_h[c1]);
  double Kh = max2(Kh_loc, A.KhTr_min);
  if (A.pass_coeff > 0.) {
    const double Rd_dx = 0.5 * (A.Rd_dx_h[c0] + A.Rd_dx_h[c1]);
    Kh_loc = Kh * max2(A.pass_min, A.pass_coeff * Rd_dx);
    if (A.KhTr_max > 0.) Kh_loc = min2(Kh_loc, A.KhTr_max);
    Kh = max2(Kh_loc, A.KhTr_min);
  }
  return Kh;
}

// thread (i, j) over (is-1 : ie, js-1 : je)
__global__ __launch_bounds__(256) void hd_khdt_kernel(HDArgs A) {
  const m6::GridDev &g = A.g;
  const int i = g.isc - 1 + blockIdx.x * blockDim.x + threadIdx.x, j = g.jsc - 1 + blockIdx.y;
  if (i > g.iec) return;
  const int I = i, J = j;
  if (j >= g.jsc) {      // khdt_x(I,j), I = is-1 .. ie
    const double Kh = A.use_VarMix ? hd_Kh_face(A, g.h2(i, j), g.h2(i + 1, j), g.u2(I, j), false) : A.KhTr;
    double kx = A.dt * (Kh * (g.dy_Cu[g.u2(I, j)] * g.IdxCu[g.u2(I, j)]));
    if (A.max_diff_CFL > 0.0) kx = min2(kx, 0.125 * A.max_diff_CFL * min2(g.areaT[g.h2(i, j)], g.areaT[g.h2(i + 1, j)]));
    A.khdt_x[g.u2(I, j)] = kx;
  }
  if (i >= g.isc) {      // khdt_y(i,J), J = js-1 .. je
    const double Kh = A.use_VarMix ? hd_Kh_face(A, g.h2(i, j), g.h2(i, j + 1), g.v2(i, J), true) : A.KhTr;
    double ky = A.dt * (Kh * (g.dx_Cv[g.v2(i, J)] * g.IdyCv[g.v2(i, J)]));
    if (A.max_diff_CFL > 0.0) ky = min2(ky, 0.125 * A.max_diff_CFL * min2(g.areaT[g.h2(i, j)], g.areaT[g.h2(i, j + 1)]));
    A.khdt_y[g.v2(i, J)] = ky;
  }
}

// thread (i, j) over the compute domain: CFL(i,j) :412-414 and its maximum
__global__ __launch_bounds__(256) void hd_cfl_kernel(HDArgs A) {
  const m6::GridDev &g = A.g;
  const int i = g.isc + blockIdx.x * blockDim.x + threadIdx.x, j = g.jsc + blockIdx.y;
  double cfl = 0.0;
  if (i <= g.iec) {
    const int I = i, J = j;
    cfl = 2.0 * ((A.khdt_x[g.u2(I - 1, j)] + A.khdt_x[g.u2(I, j)]) + (A.khdt_y[g.v2(i, J - 1)] + A.khdt_y[g.v2(i, J)])) * g.IareaT[g.h2(i, j)];
  }
  unsigned long long b = (unsigned long long)__double_as_longlong(cfl > 0.0 ? cfl : 0.0);      // positive doubles order as integers
  for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_down(b, off); b = o > b ? o : b; }
  if ((threadIdx.x & 63) == 0) atomicMax(A.cfl_bits, b);
}

// thread (i, j, k) over the compute domain
__global__ __launch_bounds__(256) void hd_step_kernel(HDArgs A) {
  const m6::GridDev &g = A.g;
  const int i = g.isc + blockIdx.x * blockDim.x + threadIdx.x, j = g.jsc + blockIdx.y, k = blockIdx.z;
  if (i > g.iec) return;
  const int I = i, J = j;
  const long kH = (long)g.nih * g.njh * k;
  const double *h = A.h + kH;
  const double h_neglect = g.H_subroundoff;
  double scale;
  if (!hd_layer_scale(A, k, scale)) return;
  const double hc = h[g.h2(i, j)], hw = h[g.h2(i - 1, j)], he = h[g.h2(i + 1, j)], hs = h[g.h2(i, j - 1)], hn = h[g.h2(i, j + 1)];
  // Coef_x(I-1,j), Coef_x(I,j), Coef_y(i,J-1), Coef_y(i,J) :552-561
  const double CxW = ((scale * A.khdt_x[g.u2(I - 1, j)]) * 2.0 * (hw * hc)) / (hw + hc + h_neglect);
  const double CxE = ((scale * A.khdt_x[g.u2(I, j)]) * 2.0 * (hc * he)) / (hc + he + h_neglect);
  const double CyS = ((scale * A.khdt_y[g.v2(i, J - 1)]) * 2.0 * (hs * hc)) / (hs + hc + h_neglect);
  const double CyN = ((scale * A.khdt_y[g.v2(i, J)]) * 2.0 * (hc * hn)) / (hc + hn + h_neglect);
  const double Ihdxdy = g.IareaT[g.h2(i, j)] / (hc + h_neglect);
  for (int m = 0; m < A.ntr; m++) {
    const double *t = A.tr[m] + kH;
    const double tc = t[g.h2(i, j)];
    const double dTr = Ihdxdy * ((CxW * (t[g.h2(i - 1, j)] - tc) - CxE * (tc - t[g.h2(i + 1, j)])) +
                                 (CyS * (t[g.h2(i, j - 1)] - tc) - CyN * (tc - t[g.h2(i, j + 1)])));
    A.work[m][kH + g.h2(i, j)] = tc + dTr;
  }
}

__global__ __launch_bounds__(256) void hd_commit_kernel(HDArgs A) {
  const m6::GridDev &g = A.g;
  const int i = g.isc + blockIdx.x * blockDim.x + threadIdx.x, j = g.jsc + blockIdx.y, k = blockIdx.z;
  if (i > g.iec) return;
  const long n = (long)g.nih * g.njh * k + g.h2(i, j);
  double scale;
  const bool stepped = hd_layer_scale(A, k, scale);      // a layer that was left out has no work copy: only the underflow applies
  for (int m = 0; m < A.ntr; m++) {
    double x = stepped ? A.work[m][n] : A.tr[m][n];
    if (A.cu && A.cu[m] > 0.0 && fabs(x) < A.cu[m]) x = 0.0;
    A.tr[m][n] = x;
  }
}

}  // namespace

extern "C" int mom6hip_tracer_hordiff(mom6hip_ctx_t *ctx, const mom6hip_tracer_hor_diff_cs_t *cs, const double *h, double dt,
                                      double *const *tr, const double *conc_underflow, int32_t ntr, int32_t memspace,
                                      mom6hip_hordiff_stats_t *stats) {
  return mom6hip_tracer_hordiff_varmix(ctx, cs, nullptr, h, dt, tr, conc_underflow, ntr, memspace, stats);
}

extern "C" int mom6hip_tracer_hordiff_varmix(mom6hip_ctx_t *ctx, const mom6hip_tracer_hor_diff_cs_t *cs, const mom6hip_hordiff_fields_t *F,
                                             const double *h, double dt, double *const *tr, const double *conc_underflow, int32_t ntr,
                                             int32_t memspace, mom6hip_hordiff_stats_t *stats) {
  M6_REQUIRE(cs != nullptr, "tracer_hordiff: null argument");
  M6_REQUIRE(!cs->unsupported[0], "tracer_hordiff: USE_NEUTRAL_DIFFUSION is not provided by this entry point (mom6hip_tracer_hordiff_neutral takes it)");
  return mom6hip_tracer_hordiff_neutral(ctx, cs, nullptr, F, h, nullptr, nullptr, dt, tr, conc_underflow, ntr, 0, 0, memspace, stats);
}

namespace m6 {
int epipycnal_branch(mom6hip_ctx_t *ctx, Stager &st, const mom6hip_epipycnal_cs_t *epi, const mom6hip_eos_t *eos, const double *h,
                     const double *khdt_x, const double *khdt_y, int num_itts, const std::vector<double *> &d_tr,
                     const std::vector<double> &cu, int idx_T, int idx_S, int *halo_updates);
int neutral_branch(mom6hip_ctx_t *ctx, Stager &st, const mom6hip_neutral_diffusion_cs_t *nd, const mom6hip_eos_t *eos, const double *h,
                   const double *p_surf, const double *h_ML, const double *khdt_x, const double *khdt_y, int num_itts, double I_numitts,
                   const std::vector<double *> &d_tr, const std::vector<double> &cu, int idx_T, int idx_S, int *halo_updates);
}

static int hordiff_impl(mom6hip_ctx_t *ctx, const mom6hip_tracer_hor_diff_cs_t *cs, const mom6hip_neutral_diffusion_cs_t *nd,
                        const mom6hip_epipycnal_cs_t *epi, const mom6hip_hordiff_fields_t *F, const double *h, const mom6hip_eos_t *eos,
                        const double *p_surf, double dt, double *const *tr, const double *conc_underflow, int32_t ntr, int32_t idx_T,
                        int32_t idx_S, int32_t memspace, mom6hip_hordiff_stats_t *stats);

extern "C" int mom6hip_tracer_hordiff_neutral(mom6hip_ctx_t *ctx, const mom6hip_tracer_hor_diff_cs_t *cs, const mom6hip_neutral_diffusion_cs_t *nd,
                                              const mom6hip_hordiff_fields_t *F, const double *h, const mom6hip_eos_t *eos, const double *p_surf,
                                              double dt, double *const *tr, const double *conc_underflow, int32_t ntr, int32_t idx_T,
                                              int32_t idx_S, int32_t memspace, mom6hip_hordiff_stats_t *stats) {
  M6_REQUIRE(cs != nullptr, "tracer_hordiff: null argument");
  M6_REQUIRE(!cs->unsupported[2], "tracer_hordiff: DIFFUSE_ML_TO_INTERIOR is not provided by this entry point (mom6hip_tracer_hordiff_epipycnal takes it)");
  return hordiff_impl(ctx, cs, nd, nullptr, F, h, eos, p_surf, dt, tr, conc_underflow, ntr, idx_T, idx_S, memspace, stats);
}

extern "C" int mom6hip_tracer_hordiff_epipycnal(mom6hip_ctx_t *ctx, const mom6hip_tracer_hor_diff_cs_t *cs, const mom6hip_epipycnal_cs_t *epi,
                                                const mom6hip_hordiff_fields_t *F, const double *h, const mom6hip_eos_t *eos, double dt,
                                                double *const *tr, const double *conc_underflow, int32_t ntr, int32_t idx_T, int32_t idx_S,
                                                int32_t memspace, mom6hip_hordiff_stats_t *stats) {
  M6_REQUIRE(cs != nullptr, "tracer_hordiff: null argument");
  M6_REQUIRE(!cs->unsupported[0], "MOM_tracer_hor_diff: USE_NEUTRAL_DIFFUSION and DIFFUSE_ML_TO_INTERIOR are mutually exclusive!");      // :1732
  M6_REQUIRE(!cs->unsupported[2] || epi != nullptr, "tracer_hordiff: DIFFUSE_ML_TO_INTERIOR needs its control structure (mom6hip_epipycnal_cs_t)");
  return hordiff_impl(ctx, cs, nullptr, cs->unsupported[2] ? epi : nullptr, F, h, eos, nullptr, dt, tr, conc_underflow, ntr, idx_T, idx_S, memspace, stats);
}

static int hordiff_impl(mom6hip_ctx_t *ctx, const mom6hip_tracer_hor_diff_cs_t *cs, const mom6hip_neutral_diffusion_cs_t *nd,
                        const mom6hip_epipycnal_cs_t *epi, const mom6hip_hordiff_fields_t *F, const double *h, const mom6hip_eos_t *eos,
                        const double *p_surf, double dt, double *const *tr, const double *conc_underflow, int32_t ntr, int32_t idx_T,
                        int32_t idx_S, int32_t memspace, mom6hip_hordiff_stats_t *stats) {
  static const char *names[8] = {"USE_NEUTRAL_DIFFUSION", "USE_HORIZONTAL_BOUNDARY_DIFFUSION", "DIFFUSE_ML_TO_INTERIOR",
                                 "(free)", "(free)", "KHTR_USE_EBT_STRUCT", "offline khdt (do_online = false)",
                                 "the df_x / df_y flux diagnostics"};
  M6_REQUIRE(ctx != nullptr, "MOM_tracer_hor_diff: register_tracer must be called before tracer_hordiff.");
  M6_REQUIRE(cs != nullptr && h != nullptr, "tracer_hordiff: null argument");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "tracer_hordiff: bad memspace");
  for (int q = 1; q < 8; q++) M6_REQUIRE(q == 2 || !cs->unsupported[q], "tracer_hordiff: %s is not provided by libmom6hip", names[q]);
  M6_REQUIRE(!cs->unsupported[2] || epi != nullptr, "tracer_hordiff: %s is not provided by this entry point", names[2]);
  const bool use_neutral = cs->unsupported[0] != 0;      // CS%use_neutral_diffusion
  if (stats) { stats->num_itts = 0; stats->halo_updates = 0; stats->max_CFL = 0.0; }
  const bool use_VarMix = cs->use_variable_mixing != 0;
  if (ntr == 0 || (cs->KhTr <= 0.0 && !use_VarMix)) return 0;      // :197
  const bool use_Eady = use_VarMix && cs->KhTr_Slope_Cff > 0., Resoln_scaled = use_VarMix && cs->Resoln_scaled_KhTr;
  M6_REQUIRE(!use_Eady || (F && F->L2u && F->L2v && F->SN_u && F->SN_v), "tracer_hordiff: KHTR_SLOPE_CFF > 0 needs VarMix%%L2u, L2v, SN_u, SN_v");
  M6_REQUIRE(!Resoln_scaled || (F && F->Res_fn_h), "tracer_hordiff: RESOLN_SCALED_KHTR needs VarMix%%Res_fn_h");
  M6_REQUIRE(!(use_VarMix && cs->KhTr_passivity_coeff > 0.) || (F && F->Rd_dx_h), "tracer_hordiff: KHTR_PASSIVITY_COEFF > 0 needs VarMix%%Rd_dx_h");
  M6_REQUIRE(tr != nullptr && ntr > 0 && ntr <= 24, "tracer_hordiff: bad tracer list (at most 24 tracers)");
  M6_REQUIRE(dt > 0.0, "tracer_hordiff: dt must be positive");
  const m6::GridDev g = ctx->g;
  M6_REQUIRE(g.dy_Cu && g.IdxCu && g.dx_Cv && g.IdyCv && g.areaT && g.IareaT, "tracer_hordiff: dy_Cu, IdxCu, dx_Cv, IdyCv, areaT and IareaT are needed");
  M6_REQUIRE(g.isc - g.isd >= 1 && g.jsc - g.jsd >= 1, "tracer_hordiff: the halo must be at least 1 point wide");
  const size_t bH = sizeof(double) * (size_t)g.nh3();
  const size_t bU2 = sizeof(double) * (size_t)(g.nih + 1) * g.njh, bV2 = sizeof(double) * (size_t)g.nih * (g.njh + 1);
  hipStream_t s = ctx->stream;
  m6::Stager st(ctx, memspace);
  HDArgs A;
  A.g = g; A.KhTr = cs->KhTr; A.max_diff_CFL = cs->max_diff_CFL; A.dt = dt; A.ntr = ntr; A.check_cfl = cs->check_diffusive_CFL;
  A.h = st.in(h, bH);
  A.nkml = epi ? epi->nkml : 0; A.nkmb = epi ? epi->nk_rho_varies : 0; A.ML_KhTr_scale = epi ? epi->ML_KhTr_scale : 1.0;
  A.use_VarMix = use_VarMix; A.use_Eady = use_Eady; A.Resoln_scaled = Resoln_scaled;
  A.KhTr_Slope_Cff = cs->KhTr_Slope_Cff; A.KhTr_fac = cs->KhTr_fac; A.KhTr_min = cs->KhTr_min; A.KhTr_max = cs->KhTr_max;
  A.pass_coeff = use_VarMix ? cs->KhTr_passivity_coeff : 0.0; A.pass_min = cs->KhTr_passivity_min;
  {
    const size_t bH2 = sizeof(double) * (size_t)g.nih * g.njh;
    A.MEKE_Kh = (use_VarMix && F) ? st.in(F->MEKE_Kh, bH2) : nullptr;
    A.L2u = use_Eady ? st.in(F->L2u, bU2) : nullptr; A.L2v = use_Eady ? st.in(F->L2v, bV2) : nullptr;
    A.SN_u = use_Eady ? st.in(F->SN_u, bU2) : nullptr; A.SN_v = use_Eady ? st.in(F->SN_v, bV2) : nullptr;
    A.Res_fn_h = Resoln_scaled ? st.in(F->Res_fn_h, bH2) : nullptr;
    A.Rd_dx_h = (A.pass_coeff > 0.) ? st.in(F->Rd_dx_h, bH2) : nullptr;
  }
  std::vector<double *> d_tr(ntr), d_work(ntr);
  for (int m = 0; m < ntr; m++) {
    M6_REQUIRE(tr[m] != nullptr, "tracer_hordiff: tracer %d is null", m);
    d_tr[m] = st.inout(tr[m], bH);
    d_work[m] = use_neutral ? nullptr : (double *)st.scratch(bH);
  }
  A.khdt_x = (double *)st.scratch(bU2); A.khdt_y = (double *)st.scratch(bV2);
  char *tab = (char *)st.scratch(2 * 24 * sizeof(double *) + 24 * sizeof(double) + 16);
  M6_REQUIRE(!st.failed() && tab, "tracer_hordiff: staging failed");
  double **t_tr = (double **)tab, **t_work = t_tr + 24;
  double *t_cu = (double *)(t_work + 24);
  A.cfl_bits = (unsigned long long *)(t_cu + 24);
  M6_HIP(hipMemcpyAsync(t_tr, d_tr.data(), ntr * sizeof(double *), hipMemcpyHostToDevice, s));
  M6_HIP(hipMemcpyAsync(t_work, d_work.data(), ntr * sizeof(double *), hipMemcpyHostToDevice, s));
  std::vector<double> cu(ntr, 0.0);
  bool any_cu = false;
  if (conc_underflow) for (int m = 0; m < ntr; m++) { cu[m] = conc_underflow[m]; any_cu = any_cu || cu[m] > 0.0; }
  M6_HIP(hipMemcpyAsync(t_cu, cu.data(), ntr * sizeof(double), hipMemcpyHostToDevice, s));
  M6_HIP(hipMemsetAsync(A.cfl_bits, 0, sizeof(unsigned long long), s));
  M6_HIP(hipMemsetAsync(A.khdt_x, 0, bU2, s)); M6_HIP(hipMemsetAsync(A.khdt_y, 0, bV2, s));
  A.tr = t_tr; A.work = t_work; A.cu = any_cu ? t_cu : nullptr;
  M6_HIP(hipStreamSynchronize(s));      // the tables above live on the host stack

  const int ni = g.iec - g.isc + 1, nj = g.jec - g.jsc + 1;
  A.scale = 1.0;
  hipLaunchKernelGGL(hd_khdt_kernel, dim3((ni + 1 + 255) / 256, nj + 1), dim3(256), 0, s, A);
  int num_itts = 1;
  double max_CFL = 0.0;
  if (cs->check_diffusive_CFL) {      // :410-423
    hipLaunchKernelGGL(hd_cfl_kernel, dim3((ni + 255) / 256, nj), dim3(256), 0, s, A);
    unsigned long long bits = 0;
    M6_HIP(hipMemcpyAsync(&bits, A.cfl_bits, sizeof(bits), hipMemcpyDeviceToHost, s));
    M6_HIP(hipStreamSynchronize(s));
    memcpy(&max_CFL, &bits, 8);
    double neg = -max_CFL;      // max_across_PEs
    if (m6::multi_tile(ctx)) { if (int rc = m6::min_across_PEs(ctx, &neg, 1)) return rc; }
    max_CFL = -neg;
    num_itts = (int)ceil(max_CFL - 4.0 * DBL_EPSILON);
    if (num_itts < 1) num_itts = 1;
  } else if (cs->max_diff_CFL > 0.0) {
    num_itts = (int)ceil(cs->max_diff_CFL - 4.0 * DBL_EPSILON);
    if (num_itts < 1) num_itts = 1;
  }
  A.scale = 1.0 / ((double)num_itts);      // I_numitts
  M6_HIP(hipGetLastError());

  std::vector<double *> pf(d_tr);
  std::vector<int32_t> ppos(ntr, MOM6HIP_POS_H), pnk(ntr, g.nk);
  int halo_updates = 0;
  if (use_neutral) {      // :474-534
    const double *d_ps = p_surf ? st.in(p_surf, sizeof(double) * (size_t)g.nih * g.njh) : nullptr;
    const double *d_hml = (nd && nd->interior_only && F && F->h_ML) ? st.in(F->h_ML, sizeof(double) * (size_t)g.nih * g.njh) : nullptr;
    M6_REQUIRE(!st.failed(), "tracer_hordiff: staging failed");
    if (int rc = m6::neutral_branch(ctx, st, nd, eos, A.h, d_ps, d_hml, A.khdt_x, A.khdt_y, num_itts, A.scale, d_tr, cu, idx_T, idx_S, &halo_updates))
      return rc;
  } else
  for (int itt = 1; itt <= num_itts; itt++) {      // :540-614
    if (int rc = m6::group_pass(ctx, pf.data(), ppos.data(), pnk.data(), ntr)) return rc;
    halo_updates++;
    hipLaunchKernelGGL(hd_step_kernel, dim3((ni + 255) / 256, nj, g.nk), dim3(256), 0, s, A);
    hipLaunchKernelGGL(hd_commit_kernel, dim3((ni + 255) / 256, nj, g.nk), dim3(256), 0, s, A);
  }
  M6_HIP(hipGetLastError());
  if (epi) {      // :613-620
    if (int rc = m6::epipycnal_branch(ctx, st, epi, eos, A.h, A.khdt_x, A.khdt_y, num_itts, d_tr, cu, idx_T, idx_S, &halo_updates)) return rc;
  }
  if (stats) { stats->num_itts = num_itts; stats->halo_updates = halo_updates; stats->max_CFL = max_CFL; }
  return st.finish();
}

extern "C" uint64_t mom6hip_abi_sizeof_tracer_hor_diff_cs(void) { return sizeof(mom6hip_tracer_hor_diff_cs_t); }
extern "C" uint64_t mom6hip_abi_sizeof_epipycnal_cs(void) { return sizeof(mom6hip_epipycnal_cs_t); }
extern "C" uint64_t mom6hip_abi_sizeof_hordiff_stats(void) { return sizeof(mom6hip_hordiff_stats_t); }
