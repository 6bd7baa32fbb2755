// sum_output.hip -- the global integrals of write_energy (src/diagnostics/MOM_sum_output.F90:490-760) for fields that live on the
// GPU: total and by-layer mass (:503-510), kinetic energy (:683-689), salt and heat (:693-712) as order-invariant
// extended-fixed-point sums (MOM_coms reproducing_sum: the numbers of `ocean.stats`, which the reference's regression tests
// compare between runs and layouts), and the two maximum CFL numbers (:718-744).  The integrands are formed in the summing
// kernel (efp.hpp); no 3-D work array exists.  Boussinesq.  The available potential energy of CALCULATE_APE (the sorted depth list
// of create_depth_list :1109-1232, the reference heights :610-630, the integrand :633-645) is the second entry point below.
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
  // (the running maxima are read first: nearly every wave finds it has nothing to add and skips the contended atomic)
  if ((threadIdx.x & 63) == 0) {
    if (b1 > __hip_atomic_load(&out[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&out[0], b1);
    if (b2 > __hip_atomic_load(&out[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&out[1], b2);
  }
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

// ---- the available potential energy (CALCULATE_APE): depth list :1109-1232, reference heights :610-630, integrand :633-645 ----
namespace {

// PE_pt(i,j,K), K = nz .. 1, of one column per lane (the running thickness below runs through k); layer nz+1 stays zero
__global__ __launch_bounds__(64) void ape_integrand_kernel(m6::GridDev g, const double *__restrict__ h, const double *__restrict__ Z_0APE,
                                                           const double *__restrict__ g_prime, double Rho0, double Z_ref, double *__restrict__ PE_pt) {
  const int i = g.isc + blockIdx.x * 64 + threadIdx.x, j = g.jsc + blockIdx.y;
  if (i > g.iec) return;
  const long n = g.h2(i, j), hpl = (long)g.nih * g.njh;
  const double areaTm = g.mask2dT[n] * g.areaT[n];
  const double D = g.bathyT[n] + Z_ref;
  const double PE_scale_factor = 1.0;
  double hbelow = 0.0;
  for (int K = g.nk; K >= 1; K--) {
    hbelow = hbelow + h[n + hpl * (K - 1)] * g.H_to_Z;
    const double hint = Z_0APE[K - 1] + (hbelow - D);
    double hbot = Z_0APE[K - 1] - D;
    hbot = (hbot + fabs(hbot)) * 0.5;
    PE_pt[n + hpl * (K - 1)] = (0.5 * PE_scale_factor * areaTm) * (Rho0 * g_prime[K - 1]) * (hint * hint - hbot * hbot);
  }
  PE_pt[n + hpl * g.nk] = 0.0;
}

// sum_across_PEs of doubles of which exactly one PE holds a non-zero value: the bit pattern in four 16-bit pieces of the
// domain's 32-bit exchange (exact)
int sum_across_PEs_disjoint(mom6hip_ctx *ctx, std::vector<double> &v) {
  if (!m6::multi_tile(ctx)) return 0;
  std::vector<int32_t> w(v.size() * 4);
  for (size_t q = 0; q < v.size(); q++) {
    unsigned long long b; memcpy(&b, &v[q], 8);
    for (int p = 0; p < 4; p++) w[4 * q + p] = (int32_t)((b >> (16 * p)) & 0xffffull);
  }
  const size_t chunk = 1u << 22;      // (the exchange takes an int count)
  for (size_t o = 0; o < w.size(); o += chunk)
    if (int rc = m6::sum_across_PEs(ctx, w.data() + o, (int)std::min(chunk, w.size() - o))) return rc;
  for (size_t q = 0; q < v.size(); q++) {
    unsigned long long b = 0ull;
    for (int p = 0; p < 4; p++) b |= ((unsigned long long)(uint32_t)w[4 * q + p] & 0xffffull) << (16 * p);
    memcpy(&v[q], &b, 8);
  }
  return 0;
}

}  // namespace

extern "C" int mom6hip_depth_list_create(mom6hip_ctx_t *ctx, int32_t niglobal, int32_t njglobal, int32_t i_offset, int32_t j_offset, double Z_ref,
                                         double min_depth_inc, int32_t *listsize_out) {
  M6_REQUIRE(ctx != nullptr, "depth_list_setup: null context");
  const m6::GridDev g = ctx->g;
  const int ni = g.iec - g.isc + 1, nj = g.jec - g.jsc + 1;
  M6_REQUIRE(niglobal >= ni && njglobal >= nj && i_offset >= 0 && j_offset >= 0 && i_offset + ni <= niglobal && j_offset + nj <= njglobal,
             "depth_list_setup: the tile does not lie inside the global domain");
  M6_REQUIRE((long)niglobal * njglobal >= 2, "depth_list_setup: the global domain has fewer than two cells");
  const int mls = niglobal * njglobal;
  const size_t nH = (size_t)g.nih * g.njh;
  std::vector<double> bathy(nH), mask(nH), area(nH);
  M6_HIP(hipMemcpyAsync(bathy.data(), g.bathyT, nH * 8, hipMemcpyDeviceToHost, ctx->stream));
  M6_HIP(hipMemcpyAsync(mask.data(), g.mask2dT, nH * 8, hipMemcpyDeviceToHost, ctx->stream));
  M6_HIP(hipMemcpyAsync(area.data(), g.areaT, nH * 8, hipMemcpyDeviceToHost, ctx->stream));
  M6_HIP(hipStreamSynchronize(ctx->stream));
  // the global lists :1136-1148 (1-based positions; entry mls+1 exists and stays zero)
  std::vector<double> Dlist((size_t)mls + 2, 0.0), AreaList((size_t)mls + 2, 0.0);
  for (int j = g.jsc; j <= g.jec; j++) for (int i = g.isc; i <= g.iec; i++) {
    const int j_global = (j - g.jsc) + j_offset + 1, i_global = (i - g.isc) + i_offset + 1;
    const size_t list_pos = (size_t)(j_global - 1) * niglobal + i_global;
    const size_t n = (size_t)g.h2(i, j);
    Dlist[list_pos] = bathy[n] + Z_ref;
    AreaList[list_pos] = mask[n] * area[n];
  }
  if (int rc = sum_across_PEs_disjoint(ctx, Dlist)) return rc;
  if (int rc = sum_across_PEs_disjoint(ctx, AreaList)) return rc;
  std::vector<int> indx2((size_t)mls + 2);
  for (int j = 1; j <= mls + 1; j++) indx2[j] = j;
  {                                                                   // the heap sort :1150-1169
    int k = mls / 2 + 1, ir = mls;
    for (;;) {
      int indxt; double Dnow;
      if (k > 1) { k = k - 1; indxt = indx2[k]; Dnow = Dlist[indxt]; }
      else {
        indxt = indx2[ir]; Dnow = Dlist[indxt];
        indx2[ir] = indx2[1];
        ir = ir - 1;
        if (ir == 1) { indx2[1] = indxt; break; }
      }
      int i = k, j = k * 2;
      for (;;) {
        if (j > ir) break;
        if (j < ir && Dlist[indx2[j]] < Dlist[indx2[j + 1]]) j = j + 1;
        if (Dnow < Dlist[indx2[j]]) { indx2[i] = indx2[j]; i = j; j = j + i; }
        else j = ir + 1;
      }
      indx2[i] = indxt;
    }
  }
  double D_list_prev = Dlist[indx2[mls]];                            // :1177-1186
  int list_size = 2;
  for (int k = mls - 1; k >= 1; k--)
    if (Dlist[indx2[k]] < D_list_prev - min_depth_inc) { list_size = list_size + 1; D_list_prev = Dlist[indx2[k]]; }
  const int listsize = list_size + 1;
  std::vector<double> depth((size_t)listsize + 1), area_l((size_t)listsize + 1), vol_below((size_t)listsize + 1);
  double vol = 0.0, area_run = 0.0, Dprev = Dlist[indx2[mls]];
  D_list_prev = Dprev;
  int kl = 0;
  for (int k = mls; k >= 1; k--) {                                    // :1195-1214
    const int i = indx2[k];
    vol = vol + area_run * (Dprev - Dlist[i]);
    area_run = area_run + AreaList[i];
    bool add_to_list = false;
    if ((kl == 0) || (k == 1)) add_to_list = true;
    else if (Dlist[indx2[k - 1]] < D_list_prev - min_depth_inc) { add_to_list = true; D_list_prev = Dlist[indx2[k - 1]]; }
    if (add_to_list) { kl = kl + 1; depth[kl] = Dlist[i]; area_l[kl] = area_run; vol_below[kl] = vol; }
    Dprev = Dlist[i];
  }
  while (kl + 1 < listsize) {                                         // :1216-1222
    kl = kl + 1;
    vol_below[kl] = vol_below[kl - 1] * 1.000001; area_l[kl] = area_l[kl - 1]; depth[kl] = depth[kl - 1];
  }
  vol_below[listsize] = vol_below[listsize - 1] * 1000.0; area_l[listsize] = area_l[listsize - 1]; depth[listsize] = depth[listsize - 1];
  ctx->DL_depth.assign(depth.begin() + 1, depth.end());
  ctx->DL_area.assign(area_l.begin() + 1, area_l.end());
  ctx->DL_vol_below.assign(vol_below.begin() + 1, vol_below.end());
  ctx->DL_lH.assign((size_t)g.nk, listsize - 1);                    // depth_list_setup :1101-1103
  if (listsize_out) *listsize_out = listsize;
  return 0;
}

extern "C" int mom6hip_write_energy_ape(mom6hip_ctx_t *ctx, const double *h, const double *mass_lay, const double *g_prime, double Rho0,
                                        double H_to_kg_m2, double Z_ref, double *PE, double *PE_tot, double *Z_0APE_out, int32_t memspace) {
  using namespace m6efp;
  M6_REQUIRE(ctx && h && mass_lay && g_prime && PE_tot, "write_energy: null argument");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "write_energy: bad memspace");
  M6_REQUIRE(!ctx->DL_depth.empty(), "write_energy: CALCULATE_APE needs the depth list (mom6hip_depth_list_create) first");
  const m6::GridDev g = ctx->g;
  const int nz = g.nk, listsize = (int)ctx->DL_depth.size();
  // the reference heights :610-630 (host: a search of the depth list per layer)
  std::vector<double> Z0((size_t)nz + 1);
  {
    auto VB = [&](int l) { return ctx->DL_vol_below[(size_t)l - 1]; };
    int lbelow = 1, li = 1;
    double volbelow = 0.0;
    for (int k = nz; k >= 1; k--) {
      const double vol_lay = ((1.0 * 1.0) * g.H_to_Z / H_to_kg_m2) * mass_lay[k - 1];      // :510
      volbelow = volbelow + vol_lay;
      int &lH = ctx->DL_lH[(size_t)k - 1];
      if ((volbelow >= VB(lH)) && (volbelow < VB(lH + 1))) {
        li = lH;
      } else {
        int labove = listsize;
        li = (labove + lbelow) / 2;
        while (li > lbelow) {
          if (volbelow < VB(li)) labove = li; else lbelow = li;
          li = (labove + lbelow) / 2;
        }
        lH = li;
      }
      lbelow = li;
      Z0[(size_t)k - 1] = ctx->DL_depth[(size_t)li - 1] - (volbelow - VB(li)) / ctx->DL_area[(size_t)li - 1];
    }
    Z0[(size_t)nz] = ctx->DL_depth[1];
  }
  m6::Stager st(ctx, memspace);
  const double *dh = st.in(h, sizeof(double) * (size_t)g.nh3());
  const size_t hpl = (size_t)g.nih * g.njh;
  double *PE_pt = (double *)st.scratch(sizeof(double) * hpl * ((size_t)nz + 1));
  double *dZ = (double *)st.scratch(sizeof(double) * 2 * ((size_t)nz + 1));
  M6_REQUIRE(!st.failed() && dh && PE_pt && dZ, "write_energy: staging failed");
  double *dG = dZ + (nz + 1);
  M6_HIP(hipMemcpyAsync(dZ, Z0.data(), sizeof(double) * ((size_t)nz + 1), hipMemcpyHostToDevice, ctx->stream));
  M6_HIP(hipMemcpyAsync(dG, g_prime, sizeof(double) * ((size_t)nz + 1), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(ape_integrand_kernel, dim3((g.iec - g.isc + 64) / 64, g.jec - g.jsc + 1), dim3(64), 0, ctx->stream, g, dh, dZ, dG, Rho0, Z_ref,
                     PE_pt);
  M6_HIP(hipGetLastError());
  const int i0 = g.isc - g.isd, i1 = g.iec - g.isd, j0 = g.jsc - g.jsd, j1 = g.jec - g.jsd;
  const int nih = g.nih;
  std::vector<unsigned long long> res;
  if (int rc = efp_reduce(ctx, [=] __device__(int i, int j, int k) { return PE_pt[hpl * k + (size_t)j * nih + i]; }, i0, i1, j0, j1, nz + 1, res)) return rc;
  std::vector<double> lay((size_t)nz + 1);
  if (int rc = efp_finish(ctx, res, nz + 1, (long long)(i1 - i0 + 1) * (j1 - j0 + 1), PE_tot, PE ? PE : lay.data(), nullptr, nullptr, nullptr, nullptr)) return rc;
  if (Z_0APE_out) for (int k = 0; k <= nz; k++) Z_0APE_out[k] = Z0[(size_t)k];
  return st.finish();
}

extern "C" uint64_t mom6hip_abi_sizeof_energy_sums(void) { return sizeof(mom6hip_energy_sums_t); }
