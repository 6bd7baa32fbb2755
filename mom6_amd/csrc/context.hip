// context.hip -- library/context entry points of the C ABI and the single-tile halo update.
#include <cstdarg>
#include <cmath>

#include <cstring>

#include "common.hpp"

#include <vector>

namespace m6 {

static thread_local std::string g_err;

void set_error(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}
ErrorKeeper::ErrorKeeper() { snprintf(saved, sizeof(saved), "%s", g_err.c_str()); }
ErrorKeeper::~ErrorKeeper() { g_err = saved; }

int DevBuf::reserve(size_t n) {
  if (n <= bytes) return 0;
  if (p) {
    if (hipFree(p) != hipSuccess) { set_error("hipFree failed"); return 1; }
    p = nullptr; bytes = 0;
  }
  hipError_t e = hipMalloc(&p, n);
  if (e != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", n, hipGetErrorString(e)); p = nullptr; return 1; }
  bytes = n;
  return 0;
}

void DevBuf::release() {
  if (p) (void)hipFree(p);
  p = nullptr; bytes = 0;
}

const double *HostTable::get(const double *host, size_t n) {
  if (!host || n == 0) { set_error("HostTable: null host table"); return nullptr; }
  if (last.size() == n && buf.p && std::memcmp(last.data(), host, n * sizeof(double)) == 0) return (const double *)buf.p;
  // the values changed (or first use): kernels in flight may still read the old copy
  if (hipDeviceSynchronize() != hipSuccess) { set_error("HostTable: device synchronise failed"); return nullptr; }
  if (buf.reserve(n * sizeof(double))) return nullptr;
  if (hipMemcpy(buf.p, host, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { set_error("HostTable: upload failed"); return nullptr; }
  last.assign(host, host + n);
  return (const double *)buf.p;
}

void *Stager::scratch(size_t bytes) {
  if (next_ >= 64) { set_error("Stager: out of pool buffers"); fail_ = true; return nullptr; }
  DevBuf &b = ctx_->pool[next_++];
  if (b.reserve(bytes)) { fail_ = true; return nullptr; }
  return b.p;
}

void *Stager::map(void *p, size_t bytes, bool rd, bool wr) {
  if (!p) return nullptr;
  if (!host_) return p;
  for (auto &e : ents_) if (e.host == p) { e.wr = e.wr || wr; return e.dev; }
  void *d = scratch(bytes);
  if (!d) return nullptr;
  if (rd && hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, ctx_->stream) != hipSuccess) {
    set_error("Stager: host-to-device copy failed"); fail_ = true; return nullptr;
  }
  ents_.push_back({p, d, bytes, wr});
  return d;
}

int Stager::finish() {
  if (fail_) return 1;
  if (!host_) return 0;
  for (auto &e : ents_)
    if (e.wr && hipMemcpyAsync(e.host, e.dev, e.bytes, hipMemcpyDeviceToHost, ctx_->stream) != hipSuccess) {
      set_error("Stager: device-to-host copy failed"); return 1;
    }
  if (hipStreamSynchronize(ctx_->stream) != hipSuccess) { set_error("Stager: stream synchronise failed"); return 1; }
  return 0;
}

}  // namespace m6

using m6::set_error;

extern "C" {

const char *mom6hip_last_error(void) { return m6::g_err.c_str(); }

int mom6hip_init(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    set_error("mom6hip_init: no HIP device available (%s); libmom6hip has no CPU fallback",
              e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    return 1;
  }
  M6_REQUIRE(device >= 0 && device < n, "mom6hip_init: device %d out of range (0..%d)", device, n - 1);
  M6_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  M6_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("mom6hip_init: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    return 1;
  }
  return 0;
}

int mom6hip_malloc(void **dptr, uint64_t bytes) {
  M6_REQUIRE(dptr != nullptr, "mom6hip_malloc: null out pointer");
  M6_HIP(hipMalloc(dptr, bytes));
  return 0;
}

int mom6hip_free(void *dptr) {
  M6_HIP(hipFree(dptr));
  return 0;
}

int mom6hip_memset_zero(mom6hip_ctx_t *ctx, void *dptr, uint64_t bytes) {
  M6_REQUIRE(ctx != nullptr && dptr != nullptr, "mom6hip_memset_zero: null argument");
  M6_HIP(hipMemsetAsync(dptr, 0, bytes, ctx->stream));
  return 0;
}

int mom6hip_sync(mom6hip_ctx_t *ctx) {
  M6_REQUIRE(ctx != nullptr, "mom6hip_sync: null context");
  M6_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

int mom6hip_sync_to_device(mom6hip_ctx_t *ctx, void *dptr, const void *hptr, uint64_t bytes) {
  M6_REQUIRE(ctx != nullptr, "mom6hip_sync_to_device: null context");
  M6_HIP(hipMemcpyAsync(dptr, hptr, bytes, hipMemcpyHostToDevice, ctx->stream));
  M6_HIP(hipStreamSynchronize(ctx->stream));
  ctx->xfer[0]++; ctx->xfer[1] += bytes;
  return 0;
}

int mom6hip_transfer_stats(mom6hip_ctx_t *ctx, uint64_t *stats, int32_t reset) {
  M6_REQUIRE(ctx != nullptr && stats != nullptr, "mom6hip_transfer_stats: null argument");
  for (int q = 0; q < 4; q++) { stats[q] = ctx->xfer[q]; if (reset) ctx->xfer[q] = 0; }
  return 0;
}

int mom6hip_sync_to_host(mom6hip_ctx_t *ctx, void *hptr, const void *dptr, uint64_t bytes) {
  M6_REQUIRE(ctx != nullptr, "mom6hip_sync_to_host: null context");
  M6_HIP(hipMemcpyAsync(hptr, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
  M6_HIP(hipStreamSynchronize(ctx->stream));
  ctx->xfer[2]++; ctx->xfer[3] += bytes;
  return 0;
}

// ---- MOM_checksums on the device ------------------------------------------------------------------------------------
namespace {
// doubles as unsigned integers that sort the same way (for atomic min / max)
__device__ __forceinline__ unsigned long long sortable(double x) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__host__ inline double unsortable(unsigned long long s) {
  const unsigned long long b = (s >> 63) ? (s & 0x7fffffffffffffffull) : ~s;
  double x; memcpy(&x, &b, 8); return x;
}
// out[0] += popcount, out[1] = min, out[2] = max over (i0:i1, j0:j1, 0:nk-1) of an array with row length nrow
__global__ __launch_bounds__(256) void chksum_kernel(const double *__restrict__ a, long plane, int nrow, int ioff, int joff, int i0,
                                                     int i1, int j0, int j1, double scale, unsigned long long *out) {
  const int i = i0 + blockIdx.x * blockDim.x + threadIdx.x, j = j0 + blockIdx.y, k = blockIdx.z;
  unsigned long long bc = 0, mn = ~0ull, mx = 0ull;
  if (i <= i1) {
    const double x = a[plane * k + (long)(j - joff) * nrow + (i - ioff)];
    bc = __popcll((unsigned long long)__double_as_longlong(fabs(scale * x)));
    mn = mx = sortable(x);
  }
  for (int off = 32; off > 0; off >>= 1) {
    bc += __shfl_down(bc, off);
    const unsigned long long m2 = __shfl_down(mn, off), x2 = __shfl_down(mx, off);
    mn = m2 < mn ? m2 : mn; mx = x2 > mx ? x2 : mx;
  }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], bc); atomicMin(&out[1], mn); atomicMax(&out[2], mx); }
}
}  // namespace

int mom6hip_chksum(mom6hip_ctx_t *ctx, const double *field, int32_t pos, int32_t nk, int32_t di, int32_t dj, int32_t symmetric,
                   double scale, int64_t *bitcount, double *amin, double *amax, int32_t memspace) {
  M6_REQUIRE(ctx && field && bitcount, "mom6hip_chksum: null argument");
  M6_REQUIRE(pos >= MOM6HIP_POS_H && pos <= MOM6HIP_POS_Q && nk >= 1, "mom6hip_chksum: bad staggering or layer count");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "mom6hip_chksum: bad memspace");
  const m6::GridDev g = ctx->g;
  const int xs = (pos == MOM6HIP_POS_U || pos == MOM6HIP_POS_Q) ? 1 : 0, ys = (pos == MOM6HIP_POS_V || pos == MOM6HIP_POS_Q) ? 1 : 0;
  const int nrow = g.nih + xs, ncol = g.njh + ys, ioff = g.isd - xs, joff = g.jsd - ys;
  int i0 = g.isc + di, i1 = g.iec + di, j0 = g.jsc + dj, j1 = g.jec + dj;
  if (symmetric && xs) i0 -= 1;
  if (symmetric && ys) j0 -= 1;
  M6_REQUIRE(i0 >= ioff && i1 < ioff + nrow && j0 >= joff && j1 < joff + ncol, "mom6hip_chksum: the shifted range leaves the array");
  const size_t bytes = sizeof(double) * (size_t)nrow * ncol * nk;
  m6::Stager st(ctx, memspace);
  const double *d = st.in(field, bytes);
  unsigned long long *out = (unsigned long long *)st.scratch(3 * sizeof(unsigned long long));
  M6_REQUIRE(!st.failed() && d && out, "mom6hip_chksum: staging failed");
  const unsigned long long init[3] = {0ull, ~0ull, 0ull};
  M6_HIP(hipMemcpyAsync(out, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(chksum_kernel, dim3((i1 - i0 + 256) / 256, j1 - j0 + 1, nk), dim3(256), 0, ctx->stream, d, (long)nrow * ncol, nrow,
                     ioff, joff, i0, i1, j0, j1, scale, out);
  M6_HIP(hipGetLastError());
  unsigned long long res[3];
  M6_HIP(hipMemcpyAsync(res, out, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
  M6_HIP(hipStreamSynchronize(ctx->stream));
  // sum_across_PEs of the count, exactly: each PE's share is first reduced modulo 1e9 (< 2^30) and exchanged as two pieces of
  // 15 bits, so the 32-bit sums stay below 2^31 for up to 65536 PEs; the pieces are recombined in 64 bits before the last modulo
  const long long share = (long long)(res[0] % 1000000000ull);
  int32_t parts[2] = {(int32_t)(share >> 15), (int32_t)(share & 32767)};
  if (m6::multi_tile(ctx)) {
    if (int rc = m6::sum_across_PEs(ctx, parts, 2)) return rc;
  }
  *bitcount = (int64_t)((((long long)parts[0] << 15) + (long long)parts[1]) % 1000000000ll);
  double mm[2] = {unsortable(res[1]), -unsortable(res[2])};      // (min, -max): one min reduction serves both
  if (m6::multi_tile(ctx)) {
    if (int rc = m6::min_across_PEs(ctx, mm, 2)) return rc;
  }
  if (amin) *amin = mm[0];
  if (amax) *amax = -mm[1];
  return st.finish();
}

int mom6hip_set_domain_callbacks(mom6hip_ctx_t *ctx, mom6hip_halo_fn halo_fn, mom6hip_sum_fn sum_fn, void *user) {
  M6_REQUIRE(ctx != nullptr, "mom6hip_set_domain_callbacks: null context");
  M6_REQUIRE((halo_fn == nullptr) == (sum_fn == nullptr), "mom6hip_set_domain_callbacks: give both callbacks or neither");
  ctx->halo_cb = halo_fn; ctx->sum_cb = sum_fn; ctx->cb_user = user;
  return 0;
}

int mom6hip_bt_graph_stats(mom6hip_ctx_t *ctx, int64_t *captures, int64_t *launches) {
  M6_REQUIRE(ctx != nullptr, "mom6hip_bt_graph_stats: null context");
  if (captures) *captures = ctx->bt_graph_captures;
  if (launches) *launches = ctx->bt_graph_launches;
  return 0;
}

int mom6hip_bt_graph_nodes(mom6hip_ctx_t *ctx, int64_t *nodes) {
  M6_REQUIRE(ctx != nullptr && nodes != nullptr, "mom6hip_bt_graph_nodes: null argument");
  *nodes = ctx->bt_graph_nodes_last;
  return 0;
}

int mom6hip_set_callback_stream_ordered(mom6hip_ctx_t *ctx, int32_t stream_ordered) {
  M6_REQUIRE(ctx != nullptr, "mom6hip_set_callback_stream_ordered: null context");
  ctx->cb_stream_ordered = stream_ordered != 0;
  return 0;
}

int mom6hip_set_min_callback(mom6hip_ctx_t *ctx, mom6hip_min_fn min_fn, void *user) {
  M6_REQUIRE(ctx != nullptr, "mom6hip_set_min_callback: null context");
  ctx->min_cb = min_fn; ctx->min_user = user;
  return 0;
}

int mom6hip_kernel_timing(mom6hip_ctx_t *ctx, int32_t enable, double *ms_total, int64_t *launches) {
  M6_REQUIRE(ctx != nullptr, "mom6hip_kernel_timing: null context");
  M6_HIP(hipStreamSynchronize(ctx->stream));
  for (int s = 0; s < MOM6HIP_KT_SLOTS; s++) {
    double ms = 0.0;
    for (auto &e : ctx->kt_events[s]) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, e.first, e.second) == hipSuccess) ms += t;
      (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second);
    }
    if (ms_total) ms_total[s] = ms;
    if (launches) launches[s] = (int64_t)ctx->kt_events[s].size();
    ctx->kt_events[s].clear();
  }
  ctx->ktiming = enable != 0;
  return 0;
}

int mom6hip_set_timing(mom6hip_ctx_t *ctx, int32_t enable) {
  M6_REQUIRE(ctx != nullptr, "mom6hip_set_timing: null context");
  ctx->timing = enable != 0;
  return 0;
}

// ABI self-description, used by tests to check the ctypes mirror.
uint64_t mom6hip_abi_sizeof_grid(void) { return sizeof(mom6hip_grid_t); }
uint64_t mom6hip_abi_sizeof_tracer_advect_cs(void) { return sizeof(mom6hip_tracer_advect_cs_t); }
uint64_t mom6hip_abi_sizeof_advect_stats(void) { return sizeof(mom6hip_advect_stats_t); }
uint64_t mom6hip_abi_sizeof_advect_timing(void) { return sizeof(mom6hip_advect_timing_t); }
uint64_t mom6hip_abi_sizeof_remapping_cs(void) { return sizeof(mom6hip_remapping_cs_t); }
uint64_t mom6hip_abi_sizeof_regridding_cs(void) { return sizeof(mom6hip_regridding_cs_t); }
uint64_t mom6hip_abi_sizeof_coriolisadv_cs(void) { return sizeof(mom6hip_coriolisadv_cs_t); }
uint64_t mom6hip_abi_sizeof_continuity_cs(void) { return sizeof(mom6hip_continuity_cs_t); }
uint64_t mom6hip_abi_sizeof_bt_cont(void) { return sizeof(mom6hip_bt_cont_t); }
uint64_t mom6hip_abi_sizeof_eos(void) { return sizeof(mom6hip_eos_t); }
uint64_t mom6hip_abi_sizeof_pressureforce_cs(void) { return sizeof(mom6hip_pressureforce_cs_t); }
uint64_t mom6hip_abi_sizeof_barotropic_cs(void) { return sizeof(mom6hip_barotropic_cs_t); }
uint64_t mom6hip_abi_sizeof_dyn_split_rk2_cs(void) { return sizeof(mom6hip_dyn_split_rk2_cs_t); }
uint64_t mom6hip_abi_sizeof_vertvisc_cs(void) { return sizeof(mom6hip_vertvisc_cs_t); }
uint64_t mom6hip_abi_sizeof_vertvisc_type(void) { return sizeof(mom6hip_vertvisc_type_t); }
uint64_t mom6hip_abi_sizeof_hor_visc_cs(void) { return sizeof(mom6hip_hor_visc_cs_t); }
uint64_t mom6hip_abi_sizeof_set_visc_cs(void) { return sizeof(mom6hip_set_visc_cs_t); }
uint64_t mom6hip_abi_offsetof_grid_mask2dT(void) { return offsetof(mom6hip_grid_t, mask2dT); }

static int upload2d(mom6hip_ctx_t *ctx, const double *h, size_t n, double **d) {
  *d = nullptr;
  if (!h) return 0;
  void *p = nullptr;
  M6_HIP(hipMalloc(&p, n * sizeof(double)));
  ctx->metric_allocs.push_back(p);
  M6_HIP(hipMemcpyAsync(p, h, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  *d = (double *)p;
  return 0;
}

int mom6hip_grid_create(const mom6hip_grid_t *grid, void *stream, mom6hip_ctx_t **out) {
  M6_REQUIRE(grid && out, "mom6hip_grid_create: null argument");
  M6_REQUIRE(grid->symmetric == 1, "mom6hip_grid_create: only SYMMETRIC_MEMORY_ layouts are supported");
  M6_REQUIRE(!grid->tripolar_n || ((grid->iec - grid->isc + 1) % 2 == 0 && grid->jed - grid->jec <= grid->jec - grid->jsc + 1),
             "mom6hip_grid_create: TRIPOLAR_N needs an even NIGLOBAL (MOM_domains.F90:191), one tile in x, and at least as many rows as the halo is wide");
  M6_REQUIRE(grid->isd <= grid->isc && grid->isc <= grid->iec && grid->iec <= grid->ied &&
             grid->jsd <= grid->jsc && grid->jsc <= grid->jec && grid->jec <= grid->jed && grid->nk >= 1,
             "mom6hip_grid_create: inconsistent index ranges");
  M6_REQUIRE(grid->areaT != nullptr, "mom6hip_grid_create: areaT is required");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error("mom6hip_grid_create: no HIP device available; libmom6hip has no CPU fallback");
    return 1;
  }
  mom6hip_ctx_t *ctx = new mom6hip_ctx();
  ctx->host = *grid;
  ctx->stream = (hipStream_t)stream;
  const int nih = grid->ied - grid->isd + 1, njh = grid->jed - grid->jsd + 1;
  const size_t nH = (size_t)nih * njh, nU = (size_t)(nih + 1) * njh, nV = (size_t)nih * (njh + 1),
               nQ = (size_t)(nih + 1) * (njh + 1);
  // pointer members in declaration order: 8 h, 8 u, 8 v, 8 q
  const double *const *src = &grid->mask2dT;
  int rc = 0;
  for (int m = 0; m < 32 && rc == 0; m++) {
    size_t n = m < 8 ? nH : (m < 16 ? nU : (m < 24 ? nV : nQ));
    rc = upload2d(ctx, src[m], n, &ctx->d_metric[m]);
  }
  if (rc) { mom6hip_grid_destroy(ctx); return rc; }

  m6::GridDev &g = ctx->g;
  g.isc = grid->isc; g.iec = grid->iec; g.jsc = grid->jsc; g.jec = grid->jec;
  g.isd = grid->isd; g.ied = grid->ied; g.jsd = grid->jsd; g.jed = grid->jed; g.nk = grid->nk;
  g.nih = nih; g.njh = njh;
  g.tripolar_n = grid->tripolar_n ? 1 : 0;
  g.Angstrom_H = grid->Angstrom_H; g.H_subroundoff = grid->H_subroundoff;
  g.dZ_subroundoff = grid->dZ_subroundoff; g.H_to_Z = grid->H_to_Z; g.Z_to_H = grid->Z_to_H;
  g.g_Earth = grid->g_Earth; g.Rho0 = grid->Rho0;
  {
    // same declaration order in mom6hip_grid_t and GridDev
    const double **dst = &g.mask2dT;
    for (int m = 0; m < 32; m++) dst[m] = ctx->d_metric[m];
  }

  // uh_neglect / vh_neglect, src/tracer/MOM_tracer_advect.F90:182-188 (a property of the grid)
  {
    std::vector<double> un(nU, 0.0), vn(nV, 0.0);
    const double *aT = grid->areaT;
    auto H2 = [&](int i, int j) { return (size_t)(i - grid->isd) + (size_t)nih * (j - grid->jsd); };
    for (int j = grid->jsd; j <= grid->jed; j++)
      for (int i = grid->isd; i <= grid->ied - 1; i++)
        un[g.u2(i, j)] = grid->H_subroundoff * fmin(aT[H2(i, j)], aT[H2(i + 1, j)]);
    for (int j = grid->jsd; j <= grid->jed - 1; j++)
      for (int i = grid->isd; i <= grid->ied; i++)
        vn[g.v2(i, j)] = grid->H_subroundoff * fmin(aT[H2(i, j)], aT[H2(i, j + 1)]);
    double *d = nullptr;
    rc = upload2d(ctx, un.data(), nU, &d); g.uh_neglect = d;
    if (!rc) { rc = upload2d(ctx, vn.data(), nV, &d); g.vh_neglect = d; }
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) { set_error("grid upload failed"); rc = 1; }
    if (rc) { mom6hip_grid_destroy(ctx); return rc; }
  }
  if (hipHostMalloc((void **)&ctx->h_domore_k, sizeof(int) * (size_t)(grid->nk + 1), hipHostMallocDefault) != hipSuccess) {
    set_error("hipHostMalloc failed"); mom6hip_grid_destroy(ctx); return 1;
  }
  *out = ctx;
  return 0;
}

int mom6hip_grid_destroy(mom6hip_ctx_t *ctx) {
  if (!ctx) return 0;
  (void)hipStreamSynchronize(ctx->stream);
  m6::native_domain_destroy(ctx);
  for (void *p : ctx->metric_allocs) (void)hipFree(p);
  ctx->hprev.release(); ctx->uhr.release(); ctx->vhr.release(); ctx->flags.release();
  for (auto &b : ctx->stage) b.release();
  for (auto &b : ctx->tr_stage) b.release();
  for (auto &b : ctx->pool) b.release();
  ctx->rk2_scratch.release(); ctx->rk2_eta_PF_start.release(); ctx->cont_hmid.release();
  for (auto &e : ctx->obc_tables) e.buf.release();
  ctx->sv_rlay.release();
  for (auto &t : ctx->tables) t.buf.release();
  ctx->hv_pack.release(); ctx->hv_str.release();
  ctx->ale_sub.release();
  ctx->ale_side.release();
  ctx->vv_ntrunc.release();
  ctx->efp_acc.release();
  for (auto &e : ctx->bt_graphs) (void)hipGraphExecDestroy((hipGraphExec_t)e.second);
  if (ctx->cap_stream) (void)hipStreamDestroy(ctx->cap_stream);
  for (hipStream_t st : ctx->side_stream) if (st) (void)hipStreamDestroy(st);
  if (ctx->side_fork) (void)hipEventDestroy(ctx->side_fork);
  for (hipEvent_t e : ctx->side_join) if (e) (void)hipEventDestroy(e);
  m6::staging_destroy(ctx);
  if (ctx->h_domore_k) (void)hipHostFree(ctx->h_domore_k);
  delete ctx;
  return 0;
}

}  // extern "C"

// ---- halo update ---------------------------------------------------------------------------------
// One launch fills both re-entrant directions of one field: x wrap for the rows of the compute
// domain, then (a second launch, because it reads what the first wrote) the y wrap over full rows.
namespace {

struct HaloDesc {
  double *f;
  int ilo, jlo;        // first allocated index in i, j
  int nis, njs, nk;    // allocated extents
  int ics, ice, jcs, jce, ni, nj, ihi, jhi;
  int poison;          // debugging: write NaN where the update would write (bit 0: the low side, bit 1: the high side); 0: the update
};

__global__ void halo_x_kernel(HaloDesc d) {
  // threads over (halo column index h, j in compute rows, k)
  const int nwl = d.ics - d.ilo, nwr = d.ihi - d.ice;       // widths of the west / east halos
  const int nw = nwl + nwr;
  const long total = (long)nw * (d.jce - d.jcs + 1) * d.nk;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    int h = (int)(t % nw);
    long r = t / nw;
    int j = d.jcs + (int)(r % (d.jce - d.jcs + 1));
    int k = (int)(r / (d.jce - d.jcs + 1));
    int i, src;
    if (h < nwl) { i = d.ilo + h; src = i + d.ni; } else { i = d.ice + 1 + (h - nwl); src = i - d.ni; }
    long base = (long)d.nis * ((long)(j - d.jlo) + (long)d.njs * k);
    if (d.poison) { if (d.poison & (h < nwl ? 1 : 2)) d.f[base + (i - d.ilo)] = __builtin_nan(""); continue; }
    d.f[base + (i - d.ilo)] = d.f[base + (src - d.ilo)];
  }
}

__global__ void halo_y_kernel(HaloDesc d) {
  const int nws = d.jcs - d.jlo, nwn = d.jhi - d.jce;
  const int nw = nws + nwn;
  const long total = (long)d.nis * nw * d.nk;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    int ii = (int)(t % d.nis);
    long r = t / d.nis;
    int h = (int)(r % nw);
    int k = (int)(r / nw);
    int j, src;
    if (h < nws) { j = d.jlo + h; src = j + d.nj; } else { j = d.jce + 1 + (h - nws); src = j - d.nj; }
    long kb = (long)d.nis * d.njs * k;
    if (d.poison) { if (d.poison & (h < nws ? 1 : 2)) d.f[kb + (long)d.nis * (j - d.jlo) + ii] = __builtin_nan(""); continue; }
    d.f[kb + (long)d.nis * (j - d.jlo) + ii] = d.f[kb + (long)d.nis * (src - d.jlo) + ii];
  }
}

// TRIPOLAR_N: the rows beyond the northern edge are the tile's own northern rows turned by half a turn (oracle/domains.c):
// cell (i, nj+m) is cell (ni+1-i, nj+1-m).  isum = isc + iec (- 1 for east faces / corners); ys: north faces / corners.
__global__ void halo_fold_kernel(HaloDesc d, int isum, int ys, int negate) {
  const int nwn = d.jhi - d.jce;
  const long total = (long)d.nis * nwn * d.nk;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int ii = (int)(t % d.nis);
    long r = t / d.nis;
    const int m = 1 + (int)(r % nwn);
    const int k = (int)(r / nwn);
    const int si = isum - (d.ilo + ii);
    if (si < d.ilo || si > d.ihi) continue;
    const int j = d.jce + m, sj = ys ? d.jce - m : d.jce + 1 - m;
    const long kb = (long)d.nis * d.njs * k;
    const double v = d.poison ? __builtin_nan("") : d.f[kb + (long)d.nis * (sj - d.jlo) + (si - d.ilo)];
    d.f[kb + (long)d.nis * (j - d.jlo) + ii] = negate ? -v : v;
  }
}

}  // namespace

namespace m6 {

// The fold of one device field on `stream` (pos may carry MOM6HIP_PASS_SCALAR_PAIR); after the x wrap of the same field.
int halo_fold_north(mom6hip_ctx_t *ctx, double *f, int pos_flags, int nk, hipStream_t stream) {
  const mom6hip_grid_t &G = ctx->host;
  const int pos = pos_flags & 3;
  const int xs = (pos == MOM6HIP_POS_U || pos == MOM6HIP_POS_Q) ? 1 : 0;
  const int ys = (pos == MOM6HIP_POS_V || pos == MOM6HIP_POS_Q) ? 1 : 0;
  const int negate = ((pos == MOM6HIP_POS_U || pos == MOM6HIP_POS_V) && !(pos_flags & MOM6HIP_PASS_SCALAR_PAIR)) ? 1 : 0;
  HaloDesc d;
  d.f = f; d.ilo = G.isd - xs; d.jlo = G.jsd - ys; d.ihi = G.ied; d.jhi = G.jed;
  d.nis = d.ihi - d.ilo + 1; d.njs = d.jhi - d.jlo + 1; d.nk = nk;
  d.ics = G.isc - xs; d.ice = G.iec; d.jcs = G.jsc - ys; d.jce = G.jec;
  d.ni = G.iec - G.isc + 1; d.nj = G.jec - G.jsc + 1; d.poison = ctx->poison_now;
  const long total = (long)d.nis * (d.jhi - d.jce) * nk;
  if (total > 0) {
    int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(halo_fold_kernel, dim3(blocks), dim3(256), 0, stream, d, G.isc + G.iec - xs, ys, negate);
  }
  M6_HIP(hipGetLastError());
  return 0;
}

// The wrap of one re-entrant direction of one device field on `stream` (dir 0: x, the compute rows; dir 1: y, full rows).
int halo_wrap_dir(mom6hip_ctx_t *ctx, double *f, int pos_flags, int nk, int dir, hipStream_t stream) {
  const mom6hip_grid_t &G = ctx->host;
  const int pos = pos_flags & 3;
  const int xs = (pos == MOM6HIP_POS_U || pos == MOM6HIP_POS_Q) ? 1 : 0;
  const int ys = (pos == MOM6HIP_POS_V || pos == MOM6HIP_POS_Q) ? 1 : 0;
  HaloDesc d;
  d.f = f; d.ilo = G.isd - xs; d.jlo = G.jsd - ys; d.ihi = G.ied; d.jhi = G.jed;
  d.nis = d.ihi - d.ilo + 1; d.njs = d.jhi - d.jlo + 1; d.nk = nk;
  d.ics = G.isc - xs; d.ice = G.iec; d.jcs = G.jsc - ys; d.jce = G.jec;
  d.ni = G.iec - G.isc + 1; d.nj = G.jec - G.jsc + 1; d.poison = ctx->poison_now;
  if (dir == 0) {
    long total = (long)((d.ics - d.ilo) + (d.ihi - d.ice)) * (d.jce - d.jcs + 1) * nk;
    if (total > 0) {
      int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(halo_x_kernel, dim3(blocks), dim3(256), 0, stream, d);
    }
  } else {
    long total = (long)d.nis * ((d.jcs - d.jlo) + (d.jhi - d.jce)) * nk;
    if (total > 0) {
      int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(halo_y_kernel, dim3(blocks), dim3(256), 0, stream, d);
    }
  }
  M6_HIP(hipGetLastError());
  return 0;
}

// Enqueue the halo fill of one device field (see oracle/domains.c for the semantics).
int halo_update_field(mom6hip_ctx_t *ctx, double *f, int pos, int nk) {
  const mom6hip_grid_t &G = ctx->host;
  if (G.reentrant_x) if (int rc = halo_wrap_dir(ctx, f, pos, nk, 0, ctx->stream)) return rc;
  if (G.reentrant_y) if (int rc = halo_wrap_dir(ctx, f, pos, nk, 1, ctx->stream)) return rc;
  if (G.tripolar_n) if (int rc = halo_fold_north(ctx, f, pos, nk, ctx->stream)) return rc;
  return 0;
}

// start_group_pass / complete_group_pass / do_group_pass (MOM_domain_infra.F90:1141-1182)
namespace {
int start_now(mom6hip_ctx_t *ctx, double *const *fields, const int32_t *pos, const int32_t *nk, int n) {
  if (ctx->native) return native_start_group_pass(ctx, fields, pos, nk, n);
  if (ctx->halo_cb) {      // the host's collective: the complete update, the tripolar fold included (pos may carry
                           // MOM6HIP_PASS_SCALAR_PAIR: only a fold tells a scalar pair from a vector)
    if (!ctx->cb_stream_ordered) M6_HIP(hipStreamSynchronize(ctx->stream));
    M6_REQUIRE(ctx->halo_cb(ctx->cb_user, fields, pos, nk, n) == 0, "group pass: the domain halo callback failed");
    return 0;
  }
  for (int f = 0; f < n; f++)      // one tile: the local wrap kernels
    if (int rc = halo_update_field(ctx, fields[f], pos[f], nk[f])) return rc;
  return 0;
}
}  // namespace

int start_group_pass(mom6hip_ctx_t *ctx, double *const *fields, const int32_t *pos, const int32_t *nk, int n) {
  if (!ctx->poison_passes) return start_now(ctx, fields, pos, nk, n);
  // debugging (mom6hip_debug_poison_passes): NaNs into every halo this pass is going to update, the update itself at
  // complete_group_pass -- whatever reads one of these halos between the two calls shows up as a NaN, whatever the timing
  M6_REQUIRE(ctx->pending.f.empty(), "start_group_pass: a pass is already in flight");
  M6_REQUIRE(!ctx->halo_cb || ctx->native, "mom6hip_debug_poison_passes: not provided for the host's halo callback");
  const mom6hip_grid_t &G = ctx->host;
  int sides[2] = {0, G.reentrant_y ? 3 : 0};      // (one tile: the x wrap is done here, as the native pass of a tile that spans x does)
  if (ctx->native) native_pass_sides(ctx, sides);
  if (G.reentrant_x && (!ctx->native || native_x_is_local(ctx)))
    for (int f = 0; f < n; f++)
      if (int rc = halo_wrap_dir(ctx, fields[f], pos[f], nk[f], 0, ctx->stream)) return rc;
  for (int f = 0; f < n; f++) {
    for (int dir = 0; dir < 2; dir++) {
      if (!sides[dir]) continue;
      ctx->poison_now = sides[dir];
      const int rc = halo_wrap_dir(ctx, fields[f], pos[f], nk[f], dir, ctx->stream);
      ctx->poison_now = 0;
      if (rc) return rc;
    }
    if (G.tripolar_n) {
      ctx->poison_now = 3;
      const int rc = halo_fold_north(ctx, fields[f], pos[f], nk[f], ctx->stream);
      ctx->poison_now = 0;
      if (rc) return rc;
    }
  }
  ctx->pending.f.assign(fields, fields + n); ctx->pending.pos.assign(pos, pos + n); ctx->pending.nk.assign(nk, nk + n);
  return 0;
}

int complete_group_pass(mom6hip_ctx_t *ctx) {
  if (ctx->poison_passes && !ctx->pending.f.empty()) {
    auto &P = ctx->pending;
    const int rc = start_now(ctx, P.f.data(), P.pos.data(), P.nk.data(), (int)P.f.size());
    P.f.clear(); P.pos.clear(); P.nk.clear();
    if (rc) return rc;
  }
  if (ctx->native) return native_complete_group_pass(ctx);
  return 0;
}

int group_pass(mom6hip_ctx_t *ctx, double *const *fields, const int32_t *pos, const int32_t *nk, int n) {
  if (int rc = start_group_pass(ctx, fields, pos, nk, n)) return rc;
  return complete_group_pass(ctx);
}

bool pass_leaves_x_final(mom6hip_ctx_t *ctx) { return ctx->native ? native_x_is_local(ctx) : (ctx->halo_cb == nullptr); }
void row_window(mom6hip_ctx_t *ctx, int j0, int j1) { ctx->g.jsc = j0; ctx->g.jec = j1; }
void row_window_reset(mom6hip_ctx_t *ctx) { ctx->g.jsc = ctx->host.jsc; ctx->g.jec = ctx->host.jec; }

int sum_across_PEs(mom6hip_ctx_t *ctx, int32_t *values, int n) {
  if (ctx->native) return native_allreduce(ctx, values, n, true);
  if (ctx->sum_cb) M6_REQUIRE(ctx->sum_cb(ctx->cb_user, values, n) == 0, "the host's sum_across_PEs failed");
  return 0;
}

int sum_across_PEs_dev(mom6hip_ctx_t *ctx, int32_t *dvalues, int n, int32_t *host_copy) {
  if (ctx->native) return native_allreduce_dev(ctx, dvalues, n, 0, host_copy);
  std::vector<int32_t> tmp;
  int32_t *h = host_copy;
  if (!h) { tmp.resize((size_t)n); h = tmp.data(); }
  M6_HIP(hipMemcpyAsync(h, dvalues, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  M6_HIP(hipStreamSynchronize(ctx->stream));
  if (!ctx->sum_cb) return 0;
  M6_REQUIRE(ctx->sum_cb(ctx->cb_user, h, n) == 0, "the host's sum_across_PEs failed");
  M6_HIP(hipMemcpyAsync(dvalues, h, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  if (!host_copy) M6_HIP(hipStreamSynchronize(ctx->stream));      // (tmp goes out of scope)
  return 0;
}

int min_across_PEs_dev_u64(mom6hip_ctx_t *ctx, unsigned long long *dvalues, int n, unsigned long long *host_copy) {
  if (ctx->native) return native_allreduce_dev(ctx, dvalues, n, 2, host_copy);
  std::vector<double> h((size_t)n);      // the bit patterns of non-negative doubles order as the doubles do
  M6_HIP(hipMemcpyAsync(h.data(), dvalues, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  M6_HIP(hipStreamSynchronize(ctx->stream));
  for (double &v : h) if (v != v) v = HUGE_VAL;      // all ones (nothing found on this tile) is above every value: +inf for the callback
  if (ctx->min_cb) {
    M6_REQUIRE(ctx->min_cb(ctx->min_user, h.data(), n) == 0, "the host's min_across_PEs failed");
    M6_HIP(hipMemcpyAsync(dvalues, h.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    M6_HIP(hipStreamSynchronize(ctx->stream));
  }
  if (host_copy) memcpy(host_copy, h.data(), sizeof(double) * (size_t)n);
  return 0;
}

int min_across_PEs(mom6hip_ctx_t *ctx, double *values, int n) {
  if (ctx->native) return native_allreduce(ctx, values, n, false);
  if (ctx->min_cb) M6_REQUIRE(ctx->min_cb(ctx->min_user, values, n) == 0, "the host's min_across_PEs failed");
  return 0;
}

}  // namespace m6

// start_group_pass / complete_group_pass of MOM_domains (MOM_domain_infra.F90:1141-1182) on device fields: the update is in flight
// between the two calls (on the communication stream of the library's own domain; at once on one tile)
extern "C" int mom6hip_start_group_pass(mom6hip_ctx_t *ctx, double *const *fields, const int32_t *pos, const int32_t *nk_each,
                                        int32_t nfields) {
  M6_REQUIRE(ctx && fields && pos && nk_each && nfields >= 0, "mom6hip_start_group_pass: null argument");
  for (int f = 0; f < nfields; f++) M6_REQUIRE(fields[f] != nullptr, "mom6hip_start_group_pass: field %d is null", f);
  return m6::start_group_pass(ctx, fields, pos, nk_each, nfields);
}
extern "C" int mom6hip_complete_group_pass(mom6hip_ctx_t *ctx) {
  M6_REQUIRE(ctx != nullptr, "mom6hip_complete_group_pass: null context");
  return m6::complete_group_pass(ctx);
}

extern "C" int mom6hip_debug_poison_passes(mom6hip_ctx_t *ctx, int32_t enable) {
  M6_REQUIRE(ctx != nullptr, "mom6hip_debug_poison_passes: null context");
  M6_REQUIRE(ctx->pending.f.empty(), "mom6hip_debug_poison_passes: a pass is in flight");
  ctx->poison_passes = (enable & 1) != 0;
  ctx->split_rows_always = (enable & 2) != 0;      // (bit 1: the interior / edge-band launches without the poisoning)
  return 0;
}

extern "C" int mom6hip_overlap_stats(mom6hip_ctx_t *ctx, uint64_t *stats, int32_t reset) {
  M6_REQUIRE(ctx != nullptr && stats != nullptr, "mom6hip_overlap_stats: null argument");
  for (int n = 0; n < 4; n++) { stats[n] = ctx->overlap[n]; if (reset) ctx->overlap[n] = 0; }
  return 0;
}

extern "C" int mom6hip_halo_update(mom6hip_ctx_t *ctx, double *const *fields, const int32_t *pos,
                                   const int32_t *nk_each, int32_t nfields) {
  M6_REQUIRE(ctx && fields && pos && nk_each, "mom6hip_halo_update: null argument");
  if (ctx->native) return m6::group_pass(ctx, fields, pos, nk_each, nfields);      // the library's own multi-tile group pass
  for (int f = 0; f < nfields; f++) {
    M6_REQUIRE(fields[f] != nullptr, "mom6hip_halo_update: field %d is null", f);
    int rc = m6::halo_update_field(ctx, fields[f], pos[f], nk_each[f]);
    if (rc) return rc;
  }
  return 0;
}

// ---- packing for the multi-tile group pass -------------------------------------------------------------
namespace {
struct PackDesc {
  double *f[24];
  long off[25];          // prefix offsets into the buffer, in doubles
  int ni[24], nj[24], nk[24], a0[24], r0[24], r1[24];
  int n, dir, w, pack;
};

__global__ void __launch_bounds__(256) halo_pack_kernel(PackDesc d, double *__restrict__ buf) {
  const long total = d.off[d.n];
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    int q = 0;
    while (t >= d.off[q + 1]) q++;
    long r = t - d.off[q];
    long idx;
    if (d.dir == 0) {       // columns [a0, a0+w) of rows [r0, r1]
      const int i = (int)(r % d.w); r /= d.w;
      const int nr = d.r1[q] - d.r0[q] + 1;
      const int j = d.r0[q] + (int)(r % nr);
      const int k = (int)(r / nr);
      idx = ((long)k * d.nj[q] + j) * d.ni[q] + d.a0[q] + i;
    } else {                // rows [a0, a0+w), all columns
      const int i = (int)(r % d.ni[q]); r /= d.ni[q];
      const int j = d.a0[q] + (int)(r % d.w);
      const int k = (int)(r / d.w);
      idx = ((long)k * d.nj[q] + j) * d.ni[q] + i;
    }
    if (d.pack) buf[t] = d.f[q][idx]; else d.f[q][idx] = buf[t];
  }
}
}  // namespace

extern "C" int mom6hip_halo_pack(mom6hip_ctx_t *ctx, double *const *fields, const int32_t *pos, const int32_t *nk_each,
                                 const int32_t *a0, int32_t nfields, int32_t dir, int32_t width, double *buf, int32_t pack,
                                 int64_t *count) {
  M6_REQUIRE(ctx != nullptr, "mom6hip_halo_pack: null argument");
  return m6::halo_pack_on(ctx, fields, pos, nk_each, a0, nfields, dir, width, buf, pack, count, ctx->stream);
}

int m6::halo_pack_on(mom6hip_ctx_t *ctx, double *const *fields, const int32_t *pos, const int32_t *nk_each, const int32_t *a0,
                     int32_t nfields, int32_t dir, int32_t width, double *buf, int32_t pack, int64_t *count, hipStream_t stream) {
  M6_REQUIRE(ctx && fields && pos && nk_each && a0, "mom6hip_halo_pack: null argument");
  M6_REQUIRE(nfields >= 0 && nfields <= 24, "mom6hip_halo_pack: at most 24 fields per message");
  const mom6hip_grid_t &G = ctx->host;
  const int h = G.isc - G.isd, nih = G.ied - G.isd + 1, njh = G.jed - G.jsd + 1, nj = G.jec - G.jsc + 1;
  PackDesc d;
  d.n = nfields; d.dir = dir; d.w = width; d.pack = pack;
  d.off[0] = 0;
  for (int q = 0; q < nfields; q++) {
    const int pq = pos[q] & 3;      // (without MOM6HIP_PASS_SCALAR_PAIR)
    const int xs = (pq == MOM6HIP_POS_U || pq == MOM6HIP_POS_Q) ? 1 : 0;
    const int ys = (pq == MOM6HIP_POS_V || pq == MOM6HIP_POS_Q) ? 1 : 0;
    d.f[q] = fields[q]; d.ni[q] = nih + xs; d.nj[q] = njh + ys; d.nk[q] = nk_each[q]; d.a0[q] = a0[q];
    d.r0[q] = h; d.r1[q] = h + nj + ys - 1;
    const long cnt = dir == 0 ? (long)nk_each[q] * (d.r1[q] - d.r0[q] + 1) * width : (long)nk_each[q] * width * d.ni[q];
    M6_REQUIRE(a0[q] >= 0 && a0[q] + width <= (dir == 0 ? d.ni[q] : d.nj[q]), "mom6hip_halo_pack: slab out of range");
    d.off[q + 1] = d.off[q] + cnt;
  }
  if (count) *count = d.off[nfields];
  if (d.off[nfields] == 0 || buf == nullptr) return 0;
  int blocks = (int)((d.off[nfields] + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(halo_pack_kernel, dim3(blocks), dim3(256), 0, stream, d, buf);
  M6_HIP(hipGetLastError());
  return 0;
}
