// domain_rccl.hip -- the group pass of MOM_domains (config_src/infra/FMS2/MOM_domain_infra.F90: create_group_pass,
// do_group_pass, start_group_pass :1141-1182, complete_group_pass) and sum_across_PEs / min_across_PEs (MOM_coms) inside the
// library, over RCCL point-to-point on a second HIP stream.
//
// One process owns one tile and one GPU.  A group pass is, per direction, ONE packed message per neighbour: the send slabs of
// every field of the group are gathered into one buffer (halo_pack_kernel), ncclSend / ncclRecv move the buffers over xGMI
// inside one ncclGroup, the received buffers are scattered into the halos.  x before y, the y slabs span the full rows, so
// the corners need no message of their own (the reference's two-stage update).  Everything runs on the context's
// COMMUNICATION stream: start_group_pass makes that stream wait for the compute stream (the fields are final), enqueues
// pack -> exchange -> unpack for both directions and records an event; complete_group_pass makes the compute stream wait
// for the event.  Kernels enqueued between the two calls overlap with the exchange -- the reference's own
// start_/complete_group_pass seams (src/core/MOM_dynamics_split_RK2.F90:541/607, 608/631, 741/749, 780/841, 991/1006,
// 1025/1043) -- and nothing synchronises the host.  A direction with a single tile is the library's wrap kernel on the same
// stream.  RCCL is loaded at run time (dlopen) when a domain is attached: single-tile runs do not need it.
#include <dlfcn.h>

#include <vector>

#include "common.hpp"

// the part of the RCCL API that is used (rccl.h is not included so that the library builds and loads without RCCL)
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt32 = 2, ncclUint64 = 5, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
}

namespace {

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl &rccl() { static Rccl r; return r; }

int load_rccl() {
  Rccl &r = rccl();
  if (r.lib) return 0;
  // the copy a host (e.g. PyTorch) has already loaded is reused; otherwise the system's
  const char *names[] = {"librccl.so.1", "librccl.so"};
  for (const char *n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (r.lib) break; }
  if (!r.lib) for (const char *n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.lib) break; }
  M6_REQUIRE(r.lib != nullptr, "mom6hip: librccl.so could not be loaded: %s", dlerror());
#define SYM(field, name) \
  r.field = (decltype(r.field))dlsym(r.lib, name); \
  M6_REQUIRE(r.field != nullptr, "mom6hip: %s is missing from librccl", name)
  SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
  SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
  SYM(AllReduce, "ncclAllReduce"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  return 0;
}

#define M6_NCCL(call)                                                                                  \
  do {                                                                                                 \
    ncclResult_t e_ = (call);                                                                          \
    if (e_ != ncclSuccess) {                                                                           \
      m6::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, rccl().GetErrorString(e_));     \
      return 1;                                                                                        \
    }                                                                                                  \
  } while (0)

}  // namespace

// the native domain of a context
struct m6_native_domain {
  ncclComm_t comm = nullptr;
  hipStream_t cstream = nullptr;
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  hipEvent_t ev_red[2] = {nullptr, nullptr};   // the device-side reductions: compute stream -> communication stream and back
  int rank = 0, nranks = 1;
  int lo[2] = {-1, -1}, hi[2] = {-1, -1};      // neighbour ranks per direction (x, y); -1: none
  m6::DevBuf sbuf[2][2], rbuf[2][2];           // [direction][0: to / from hi, 1: to / from lo]
  m6::DevBuf red;                              // device buffer of the reductions
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> timed;      // (exchange enqueued, exchange done) on the communication stream
  long npasses = 0;
};

extern "C" int mom6hip_rccl_get_unique_id(void *id, int32_t nbytes) {
  M6_REQUIRE(id != nullptr && nbytes >= (int32_t)sizeof(ncclUniqueId), "mom6hip_rccl_get_unique_id: the buffer must hold 128 bytes");
  if (load_rccl()) return 1;
  ncclUniqueId u;
  M6_NCCL(rccl().GetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return 0;
}

extern "C" int mom6hip_domain_init_rccl(mom6hip_ctx_t *ctx, const mom6hip_domain_t *dom, const void *unique_id, int32_t nbytes) {
  M6_REQUIRE(ctx && dom && unique_id && nbytes >= (int32_t)sizeof(ncclUniqueId), "mom6hip_domain_init_rccl: bad argument");
  M6_REQUIRE(dom->nranks >= 1 && dom->rank >= 0 && dom->rank < dom->nranks, "mom6hip_domain_init_rccl: bad rank / nranks");
  const int nb[4] = {dom->nbr_w, dom->nbr_e, dom->nbr_s, dom->nbr_n};
  for (int q = 0; q < 4; q++) M6_REQUIRE(nb[q] >= -1 && nb[q] < dom->nranks, "mom6hip_domain_init_rccl: neighbour %d out of range", q);
  M6_REQUIRE((dom->nbr_w < 0) == (dom->nbr_e < 0) || !ctx->host.reentrant_x, "mom6hip_domain_init_rccl: inconsistent x neighbours");
  M6_REQUIRE(ctx->native == nullptr, "mom6hip_domain_init_rccl: the context already has a domain");
  if (load_rccl()) return 1;
  m6_native_domain *D = new m6_native_domain();
  D->rank = dom->rank; D->nranks = dom->nranks;
  D->lo[0] = dom->nbr_w; D->hi[0] = dom->nbr_e; D->lo[1] = dom->nbr_s; D->hi[1] = dom->nbr_n;
  ncclUniqueId u;
  memcpy(&u, unique_id, sizeof(u));
  M6_NCCL(rccl().CommInitRank(&D->comm, dom->nranks, u, dom->rank));
  M6_HIP(hipStreamCreateWithFlags(&D->cstream, hipStreamNonBlocking));
  M6_HIP(hipEventCreateWithFlags(&D->ev_ready, hipEventDisableTiming));
  M6_HIP(hipEventCreateWithFlags(&D->ev_done, hipEventDisableTiming));
  for (hipEvent_t &e : D->ev_red) M6_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  ctx->native = D;
  return 0;
}

namespace m6 {

void native_domain_destroy(mom6hip_ctx *ctx) {
  m6_native_domain *D = ctx->native;
  if (!D) return;
  (void)hipStreamSynchronize(D->cstream);
  if (D->comm) (void)rccl().CommDestroy(D->comm);
  for (auto &p : D->timed) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  for (int d = 0; d < 2; d++) for (int s = 0; s < 2; s++) { D->sbuf[d][s].release(); D->rbuf[d][s].release(); }
  D->red.release();
  (void)hipEventDestroy(D->ev_ready); (void)hipEventDestroy(D->ev_done);
  for (hipEvent_t e : D->ev_red) if (e) (void)hipEventDestroy(e);
  (void)hipStreamDestroy(D->cstream);
  delete D;
  ctx->native = nullptr;
}

// start_group_pass: everything of the pass is enqueued on the communication stream, behind what the compute stream has
// enqueued so far
int native_start_group_pass(mom6hip_ctx *ctx, double *const *fields, const int32_t *pos_flags, const int32_t *nk, int n) {
  m6_native_domain *D = ctx->native;
  std::vector<int32_t> pos_only(pos_flags, pos_flags + n);      // the staggering without MOM6HIP_PASS_SCALAR_PAIR
  for (auto &q : pos_only) q &= 3;
  const int32_t *pos = pos_only.data();
  const mom6hip_grid_t &G = ctx->host;
  hipStream_t cs = D->cstream;
  // the fold is turned from this tile's own rows: the tile must span x and end the domain to the north
  if (G.tripolar_n)
    M6_REQUIRE(D->hi[1] < 0 && (D->lo[0] < 0 || D->lo[0] == D->rank) && (D->hi[0] < 0 || D->hi[0] == D->rank),
               "group pass: a tile on the tripolar fold must span x (no x neighbour but itself) and cannot have a northern neighbour");
  // A tile that spans x wraps its own x halos: on the compute stream, now -- the pass is non-blocking in y only, which is what
  // lets the caller run the interior rows of the next kernels before complete_group_pass (m6::pass_leaves_x_final)
  const bool x_local = D->lo[0] < 0 && D->hi[0] < 0;
  if (x_local && G.reentrant_x)
    for (int f = 0; f < n; f++)
      if (int rc = halo_wrap_dir(ctx, fields[f], pos[f], nk[f], 0, ctx->stream)) return rc;
  M6_HIP(hipEventRecord(D->ev_ready, ctx->stream));
  M6_HIP(hipStreamWaitEvent(cs, D->ev_ready, 0));
  hipEvent_t t0 = nullptr, t1 = nullptr;
  if (D->timing) {
    if (hipEventCreate(&t0) != hipSuccess) t0 = nullptr;
    else if (hipEventCreate(&t1) != hipSuccess) { (void)hipEventDestroy(t0); t0 = t1 = nullptr; }
    else (void)hipEventRecord(t0, cs);
  }
  const int h = G.isc - G.isd, ni = G.iec - G.isc + 1, nj = G.jec - G.jsc + 1;
  const int w = h;
  for (int dir = 0; dir < 2; dir++) {
    const int lo = D->lo[dir], hi = D->hi[dir];
    if (lo < 0 && hi < 0) {      // one tile in this direction: the local wrap (or a closed edge); x was done above
      if (dir == 1 && G.reentrant_y)
        for (int f = 0; f < n; f++)
          if (int rc = halo_wrap_dir(ctx, fields[f], pos[f], nk[f], dir, cs)) return rc;
      continue;
    }
    std::vector<int32_t> a_to_hi(n), a_to_lo(n), a_lo_halo(n), a_hi_halo(n);
    const int nn = dir == 0 ? ni : nj;
    for (int f = 0; f < n; f++) {
      const int s = dir == 0 ? ((pos[f] == MOM6HIP_POS_U || pos[f] == MOM6HIP_POS_Q) ? 1 : 0)
                             : ((pos[f] == MOM6HIP_POS_V || pos[f] == MOM6HIP_POS_Q) ? 1 : 0);
      a_to_hi[f] = nn + h - w; a_to_lo[f] = h + s; a_lo_halo[f] = h - w; a_hi_halo[f] = h + nn + s;
    }
    int64_t cnt = 0;
    if (int rc = halo_pack_on(ctx, fields, pos, nk, a_to_hi.data(), n, dir, w, nullptr, 1, &cnt, cs)) return rc;
    const size_t bytes = sizeof(double) * (size_t)cnt;
    for (int s = 0; s < 2; s++)
      M6_REQUIRE(D->sbuf[dir][s].reserve(bytes) == 0 && D->rbuf[dir][s].reserve(bytes) == 0, "group pass: out of device memory");
    double *s_hi = (double *)D->sbuf[dir][0].p, *s_lo = (double *)D->sbuf[dir][1].p;
    double *r_hi = (double *)D->rbuf[dir][0].p, *r_lo = (double *)D->rbuf[dir][1].p;
    if (hi >= 0) if (int rc = halo_pack_on(ctx, fields, pos, nk, a_to_hi.data(), n, dir, w, s_hi, 1, nullptr, cs)) return rc;
    if (lo >= 0) if (int rc = halo_pack_on(ctx, fields, pos, nk, a_to_lo.data(), n, dir, w, s_lo, 1, nullptr, cs)) return rc;
    // Messages between one pair of ranks are matched in posting order.  When both neighbours are the same rank (two tiles
    // in a re-entrant direction, or this rank itself) the peer's first message is its "to-high" block, which lands in
    // my LOW halo: post send->hi, recv<-lo, send->lo, recv<-hi.
    // An error inside the group must not leave it open for every later call: remember the first one, always close the group.
    M6_NCCL(rccl().GroupStart());
    ncclResult_t ge = ncclSuccess, e1;
    const char *what = "";
    if (hi >= 0 && (e1 = rccl().Send(s_hi, (size_t)cnt, ncclFloat64, hi, D->comm, cs)) != ncclSuccess && ge == ncclSuccess) { ge = e1; what = "ncclSend to the high neighbour"; }
    if (lo >= 0 && (e1 = rccl().Recv(r_lo, (size_t)cnt, ncclFloat64, lo, D->comm, cs)) != ncclSuccess && ge == ncclSuccess) { ge = e1; what = "ncclRecv from the low neighbour"; }
    if (lo >= 0 && (e1 = rccl().Send(s_lo, (size_t)cnt, ncclFloat64, lo, D->comm, cs)) != ncclSuccess && ge == ncclSuccess) { ge = e1; what = "ncclSend to the low neighbour"; }
    if (hi >= 0 && (e1 = rccl().Recv(r_hi, (size_t)cnt, ncclFloat64, hi, D->comm, cs)) != ncclSuccess && ge == ncclSuccess) { ge = e1; what = "ncclRecv from the high neighbour"; }
    e1 = rccl().GroupEnd();
    if (ge != ncclSuccess || e1 != ncclSuccess) {
      if (t0) { (void)hipEventDestroy(t0); (void)hipEventDestroy(t1); }
      m6::set_error("group pass (direction %d): %s failed: %s", dir, ge != ncclSuccess ? what : "ncclGroupEnd",
                    rccl().GetErrorString(ge != ncclSuccess ? ge : e1));
      return 1;
    }
    if (lo >= 0) if (int rc = halo_pack_on(ctx, fields, pos, nk, a_lo_halo.data(), n, dir, w, r_lo, 0, nullptr, cs)) return rc;
    if (hi >= 0) if (int rc = halo_pack_on(ctx, fields, pos, nk, a_hi_halo.data(), n, dir, w, r_hi, 0, nullptr, cs)) return rc;
  }
  if (G.tripolar_n) {      // this tile's northern edge is the fold (the tiles span x): its own rows, turned
    for (int f = 0; f < n; f++)
      if (int rc = halo_fold_north(ctx, fields[f], pos_flags[f], nk[f], cs)) return rc;
  }
  if (t0 && t1) { (void)hipEventRecord(t1, cs); D->timed.push_back({t0, t1}); }
  M6_HIP(hipEventRecord(D->ev_done, cs));
  D->npasses++;
  return 0;
}

void native_pass_sides(mom6hip_ctx *ctx, int sides[2]) {
  const m6_native_domain *D = ctx->native;
  const mom6hip_grid_t &G = ctx->host;
  for (int dir = 0; dir < 2; dir++) sides[dir] = (D->lo[dir] >= 0 ? 1 : 0) | (D->hi[dir] >= 0 ? 2 : 0);
  if (D->lo[1] < 0 && D->hi[1] < 0 && G.reentrant_y) sides[1] = 3;      // (the local y wrap runs on the communication stream)
}

bool native_x_is_local(mom6hip_ctx *ctx) { return ctx->native->lo[0] < 0 && ctx->native->hi[0] < 0; }


int native_complete_group_pass(mom6hip_ctx *ctx) {
  M6_HIP(hipStreamWaitEvent(ctx->stream, ctx->native->ev_done, 0));
  return 0;
}

// sum_across_PEs (int32) / min_across_PEs (double) of n HOST values, in place (synchronises: the host needs the result)
int native_allreduce(mom6hip_ctx *ctx, void *values, int n, bool is_int_sum) {
  m6_native_domain *D = ctx->native;
  const size_t bytes = (is_int_sum ? sizeof(int32_t) : sizeof(double)) * (size_t)n;
  M6_REQUIRE(D->red.reserve(bytes) == 0, "allreduce: out of device memory");
  M6_HIP(hipMemcpyAsync(D->red.p, values, bytes, hipMemcpyHostToDevice, D->cstream));
  M6_NCCL(rccl().AllReduce(D->red.p, D->red.p, (size_t)n, is_int_sum ? ncclInt32 : ncclFloat64, is_int_sum ? ncclSum : ncclMin, D->comm,
                           D->cstream));
  M6_HIP(hipMemcpyAsync(values, D->red.p, bytes, hipMemcpyDeviceToHost, D->cstream));
  M6_HIP(hipStreamSynchronize(D->cstream));
  return 0;
}

// The same on n DEVICE values in place (sum of int32, or min of doubles / of the bit patterns of non-negative doubles as uint64): the
// all-reduce runs on the communication stream behind what the compute stream has been given so far, and the compute stream waits for
// it -- no host copy on the way.  host_copy (optional): the result for the host, read back behind the all-reduce; the call then returns
// when it has arrived (the one synchronisation of a loop that the host ends on the result, MOM_tracer_advect.F90:305).
int native_allreduce_dev(mom6hip_ctx *ctx, void *dvalues, int n, int kind, void *host_copy) {
  m6_native_domain *D = ctx->native;
  const ncclDataType_t ty = kind == 0 ? ncclInt32 : (kind == 1 ? ncclFloat64 : ncclUint64);
  const size_t bytes = (kind == 0 ? sizeof(int32_t) : sizeof(double)) * (size_t)n;
  M6_HIP(hipEventRecord(D->ev_red[0], ctx->stream));
  M6_HIP(hipStreamWaitEvent(D->cstream, D->ev_red[0], 0));
  M6_NCCL(rccl().AllReduce(dvalues, dvalues, (size_t)n, ty, kind == 0 ? ncclSum : ncclMin, D->comm, D->cstream));
  if (host_copy) M6_HIP(hipMemcpyAsync(host_copy, dvalues, bytes, hipMemcpyDeviceToHost, D->cstream));
  M6_HIP(hipEventRecord(D->ev_red[1], D->cstream));
  M6_HIP(hipStreamWaitEvent(ctx->stream, D->ev_red[1], 0));
  if (host_copy) M6_HIP(hipStreamSynchronize(D->cstream));
  return 0;
}

}  // namespace m6

extern "C" int mom6hip_domain_exchange_timing(mom6hip_ctx_t *ctx, int32_t enable, double *ms_total, int64_t *npasses) {
  M6_REQUIRE(ctx != nullptr, "mom6hip_domain_exchange_timing: null context");
  if (ms_total) *ms_total = 0.0;
  if (npasses) *npasses = 0;
  m6_native_domain *D = ctx->native;
  if (!D) return 0;
  M6_HIP(hipStreamSynchronize(D->cstream));
  double ms = 0.0;
  for (auto &p : D->timed) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, p.first, p.second) == hipSuccess) ms += t;
    (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second);
  }
  if (ms_total) *ms_total = ms;
  if (npasses) *npasses = (int64_t)D->timed.size();
  D->timed.clear();
  D->timing = enable != 0;
  return 0;
}
