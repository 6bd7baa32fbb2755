// sum_output.hip -- the global integrals of write_energy (src/diagnostics/MOM_sum_output.F90:490-760) for fields that live on the
// GPU: total and by-layer mass (:503-510), kinetic energy (:683-689), salt and heat (:693-712) as order-invariant
// extended-fixed-point sums (MOM_coms reproducing_sum: the numbers of `ocean.stats`, which the reference's regression tests
// compare between runs and layouts), and the two maximum CFL numbers (:718-744).  The integrands are formed in the summing
// kernel (efp.hpp); no 3-D work array exists.  Boussinesq; the available-potential-energy part (CALCULATE_APE: a sorted depth
// list built at initialisation, :625-680) is not provided: PE_tot = 0, as with CALCULATE_APE = False.
#include "efp.hpp"

namespace {

__global__ __launch_bounds__(256) void max_cfl_kernel(m6::GridDev g, const double *__restrict__ u, const double *__restrict__ v, double dt,
                                                      unsigned long long *__restrict__ out) {
  // x = 0 .. ni: the u face I = isc-1+x of row j (x <= ni) and the v face (i = isc-1+x, J) (x >= 1); y = 0 .. nj likewise
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, k = blockIdx.z;
  const int ni = g.iec - g.isc + 1;
  double c1 = 0.0, c2 = 0.0;
  if (x <= ni) {
    const int I = g.isc - 1 + x, j = g.jsc - 1 + y;
    if (y >= 1) {      // u(I, j), I = Isq..Ieq, j = js..je :719-728
      const double uu = u[g.u3(I, j, k)];
      double CFL_Iarea = g.IareaT[g.h2(I, j)];
      if (uu < 0.0) CFL_Iarea = g.IareaT[g.h2(I + 1, j)];
      c1 = fabs(uu * dt) * (g.dy_Cu[g.u2(I, j)] * CFL_Iarea);
      c2 = fabs(uu * dt) * g.IdxCu[g.u2(I, j)];
    }
    if (x >= 1) {      // v(i, J), J = Jsq..Jeq, i = is..ie :729-738
      const int i = I, J = j;
      const double vv = v[g.v3(i, J, k)];
      double CFL_Iarea = g.IareaT[g.h2(i, J)];
      if (vv < 0.0) CFL_Iarea = g.IareaT[g.h2(i, J + 1)];
      c1 = m6::max2(c1, fabs(vv * dt) * (g.dx_Cv[g.v2(i, J)] * CFL_Iarea));
      c2 = m6::max2(c2, fabs(vv * dt) * g.IdyCv[g.v2(i, J)]);
    }
  }
  // (non-negative doubles order as their bit patterns; a NaN does not pass a `>` test in the reference's max either)
  unsigned long long b1 = c1 > 0.0 ? (unsigned long long)__double_as_longlong(c1) : 0ull;
  unsigned long long b2 = c2 > 0.0 ? (unsigned long long)__double_as_longlong(c2) : 0ull;
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o1 = __shfl_down(b1, off), o2 = __shfl_down(b2, off);
    b1 = o1 > b1 ? o1 : b1; b2 = o2 > b2 ? o2 : b2;
  }
  if ((threadIdx.x & 63) == 0) { if (b1) atomicMax(&out[0], b1); if (b2) atomicMax(&out[1], b2); }
}

}  // namespace

extern "C" int mom6hip_write_energy_sums(mom6hip_ctx_t *ctx, const double *u, const double *v, const double *h, const double *T,
                                         const double *S, double dt, double C_p, double H_to_kg_m2, double *mass_lay, double *KE_lay,
                                         mom6hip_energy_sums_t *out, int32_t memspace) {
  using namespace m6efp;
  M6_REQUIRE(ctx && u && v && h && out, "write_energy: null argument");
  M6_REQUIRE((T == nullptr) == (S == nullptr), "write_energy: give both T and S (use_temperature) or neither");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "write_energy: bad memspace");
  const m6::GridDev g = ctx->g;
  M6_REQUIRE(g.mask2dT && g.areaT && g.IareaT && g.dy_Cu && g.dx_Cv && g.IdxCu && g.IdyCv, "write_energy: a required grid metric is missing");
  const int nk = g.nk;
  m6::Stager st(ctx, memspace);
  const double *du = st.in(u, sizeof(double) * (size_t)g.nu3()), *dv = st.in(v, sizeof(double) * (size_t)g.nv3());
  const double *dh = st.in(h, sizeof(double) * (size_t)g.nh3());
  const double *dT = st.in(T, sizeof(double) * (size_t)g.nh3()), *dS = st.in(S, sizeof(double) * (size_t)g.nh3());
  unsigned long long *cfl = (unsigned long long *)st.scratch(2 * sizeof(unsigned long long));
  M6_REQUIRE(!st.failed() && du && dv && dh && cfl, "write_energy: staging failed");
  memset(out, 0, sizeof(*out));
  // h-point computational domain as array offsets; u(I-1), u(I) of cell i are offsets i, i+1 of a u row; v(J-1), v(J) rows j, j+1
  const int i0 = g.isc - g.isd, i1 = g.iec - g.isd, j0 = g.jsc - g.jsd, j1 = g.jec - g.jsd;
  const long long npts = (long long)(i1 - i0 + 1) * (j1 - j0 + 1);
  const int nih = g.nih, sU = g.nih + 1;
  const long hpl = (long)g.nih * g.njh, upl = (long)(g.nih + 1) * g.njh, vpl = (long)g.nih * (g.njh + 1);
  const double *mask = g.mask2dT, *area = g.areaT;
  const double HL2_to_kg = H_to_kg_m2 * (1.0 * 1.0);                      // GV%H_to_kg_m2*US%L_to_m**2 :490
  std::vector<unsigned long long> res;
  int64_t np = 0;
  // ---- mass :503-510 (Boussinesq)
  if (int rc = efp_reduce(ctx, [=] __device__(int i, int j, int k) {
        const long n = (long)j * nih + i;
        const double areaTm = mask[n] * area[n];
        return dh[hpl * k + n] * (HL2_to_kg * areaTm);
      }, i0, i1, j0, j1, nk, res)) return rc;
  std::vector<double> lay(nk);
  if (int rc = efp_finish(ctx, res, nk, npts, &out->mass_tot, mass_lay ? mass_lay : lay.data(), out->mass_EFP, nullptr, &np, nullptr)) return rc;
  // ---- kinetic energy :683-689
  const double KE_scale_factor = HL2_to_kg * (1.0 * 1.0);                 // HL2_to_kg*US%L_T_to_m_s**2
  if (int rc = efp_reduce(ctx, [=] __device__(int i, int j, int k) {
        const long n = (long)j * nih + i;
        const double areaTm = mask[n] * area[n];
        const double uW = du[upl * k + (long)j * sU + i], uE = du[upl * k + (long)j * sU + i + 1];
        const double vS = dv[vpl * k + (long)j * nih + i], vN = dv[vpl * k + (long)(j + 1) * nih + i];
        return (0.25 * KE_scale_factor * (areaTm * dh[hpl * k + n])) * ((uW * uW + uE * uE) + (vS * vS + vN * vN));
      }, i0, i1, j0, j1, nk, res)) return rc;
  if (int rc = efp_finish(ctx, res, nk, npts, &out->KE_tot, KE_lay ? KE_lay : lay.data(), nullptr, nullptr, nullptr, nullptr)) return rc;
  out->PE_tot = 0.0;
  out->toten = out->KE_tot + out->PE_tot;                                  // :691
  // ---- salt and heat :693-712: the column integrals in k order, then reproducing_sum_EFP(only_on_PE) + EFP_sum_across_PEs
  if (dT) {
    for (int which = 0; which < 2; which++) {
      const double *fld = which ? dT : dS;
      const double fac = which ? (1.0 * C_p) : 1.0;                       // US%Q_to_J_kg*tv%C_p ; US%S_to_ppt
      if (int rc = efp_reduce(ctx, [=] __device__(int i, int j, int) {
            const long n = (long)j * nih + i;
            const double w = HL2_to_kg * (mask[n] * area[n]);
            double acc = 0.0;
            for (int k = 0; k < nk; k++) acc = acc + (fac * fld[hpl * k + n]) * (dh[hpl * k + n] * w);
            return acc;
          }, i0, i1, j0, j1, 1, res)) return rc;
      if (int rc = efp_finish(ctx, res, 1, npts, which ? &out->Heat : &out->Salt, nullptr, which ? out->heat_EFP : out->salt_EFP, nullptr,
                              nullptr, nullptr, true)) return rc;
    }
  }
  // ---- the maximum CFL numbers :718-744
  M6_HIP(hipMemsetAsync(cfl, 0, 2 * sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(max_cfl_kernel, dim3((g.iec - g.isc + 2 + 255) / 256, g.jec - g.jsc + 2, nk), dim3(256), 0, ctx->stream, g, du, dv, dt, cfl);
  M6_HIP(hipGetLastError());
  unsigned long long hb[2];
  M6_HIP(hipMemcpyAsync(hb, cfl, sizeof(hb), hipMemcpyDeviceToHost, ctx->stream));
  M6_HIP(hipStreamSynchronize(ctx->stream));
  double mc[2]; memcpy(mc, hb, sizeof(mc));
  if (m6::multi_tile(ctx)) {      // max_across_PEs(max_CFL, 2) through the domain's min reduction
    double neg[2] = {-mc[0], -mc[1]};
    if (int rc = m6::min_across_PEs(ctx, neg, 2)) return rc;
    mc[0] = -neg[0]; mc[1] = -neg[1];
  }
  out->max_CFL[0] = mc[0]; out->max_CFL[1] = mc[1];
  out->npoints = np;
  return st.finish();
}

extern "C" uint64_t mom6hip_abi_sizeof_energy_sums(void) { return sizeof(mom6hip_energy_sums_t); }
