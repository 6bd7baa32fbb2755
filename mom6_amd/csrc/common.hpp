// common.hpp -- context, error handling and index helpers shared by the libmom6hip sources.
// gfx950 (MI355X) only.  All arithmetic is fp64 and the library is built with -ffp-contract=off so
// that every expression is evaluated with the reference's explicit parenthesisation.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mom6hip.h"

struct mom6hip_ctx;
struct m6_native_domain;      // the RCCL domain of a context (domain_rccl.hip)

namespace m6 {

// ---- errors --------------------------------------------------------------------------------
void set_error(const char *fmt, ...);
// Keeps the message of the error being reported while clean-up calls that may fail themselves (and would overwrite it) run
struct ErrorKeeper {
  ErrorKeeper();
  ~ErrorKeeper();
  char saved[1024];
};

#define M6_HIP(call)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      m6::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return 1;                                                                             \
    }                                                                                       \
  } while (0)

#define M6_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      m6::set_error(__VA_ARGS__);        \
      return 2;                          \
    }                                    \
  } while (0)

// ---- device-side view of the grid -------------------------------------------------------------
// Index ranges + device pointers to the metrics; passed to kernels by value.
struct GridDev {
  int isc, iec, jsc, jec, isd, ied, jsd, jed, nk;
  int nih, njh;             // data-domain extents of h-point arrays
  int tripolar_n;           // this tile's northern edge is the tripolar fold (TRIPOLAR_N)
  double Angstrom_H, H_subroundoff, dZ_subroundoff, H_to_Z, Z_to_H, g_Earth, Rho0;
  // device copies of the metric arrays of mom6hip_grid_t (null if the caller did not provide them)
  const double *mask2dT, *areaT, *IareaT, *dxT, *dyT, *IdxT, *IdyT, *bathyT;
  const double *mask2dCu, *dxCu, *dyCu, *dy_Cu, *IdxCu, *IdyCu, *areaCu, *IareaCu;
  const double *mask2dCv, *dxCv, *dyCv, *dx_Cv, *IdxCv, *IdyCv, *areaCv, *IareaCv;
  const double *mask2dBu, *dxBu, *dyBu, *areaBu, *IareaBu, *CoriolisBu, *IdxBu, *IdyBu;
  const double *uh_neglect, *vh_neglect;   // derived: MOM_tracer_advect.F90:182-188
  // linear offsets, Fortran indices, k zero-based
  __host__ __device__ inline long h2(int i, int j) const { return (long)(i - isd) + (long)nih * (j - jsd); }
  __host__ __device__ inline long u2(int I, int j) const { return (long)(I - isd + 1) + (long)(nih + 1) * (j - jsd); }
  __host__ __device__ inline long v2(int i, int J) const { return (long)(i - isd) + (long)nih * (J - jsd + 1); }
  __host__ __device__ inline long q2(int I, int J) const { return (long)(I - isd + 1) + (long)(nih + 1) * (J - jsd + 1); }
  __host__ __device__ inline long h3(int i, int j, int k) const { return h2(i, j) + (long)nih * njh * k; }
  __host__ __device__ inline long u3(int I, int j, int k) const { return u2(I, j) + (long)(nih + 1) * njh * k; }
  __host__ __device__ inline long v3(int i, int J, int k) const { return v2(i, J) + (long)nih * (njh + 1) * k; }
  __host__ __device__ inline long nh3() const { return (long)nih * njh * nk; }
  __host__ __device__ inline long nu3() const { return (long)(nih + 1) * njh * nk; }
  __host__ __device__ inline long nv3() const { return (long)nih * (njh + 1) * nk; }
};

// min/max with the compare-and-select semantics of the oracle's (and a Fortran compiler's usual) MIN/MAX
// lowering: min2(a,b) = a < b ? a : b.  Unlike v_min_f64/v_max_f64 this is deterministic for (-0.0, +0.0)
// ties, which occur at every masked point; use it wherever exact zeros are common.
__device__ __forceinline__ double min2(double a, double b) { return a < b ? a : b; }
__device__ __forceinline__ double max2(double a, double b) { return a > b ? a : b; }

// grow-only device buffer
struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  int reserve(size_t n);
  void release();
};

// Device copy of a short host table of doubles (GV%Rlay, GV%g_prime ...): uploaded again only when the host's values have
// changed, so a call in the time loop neither copies nor synchronises (context.hip)
struct HostTable {
  DevBuf buf;
  std::vector<double> last;
  const double *get(const double *host, size_t n);      // nullptr on failure (error set)
};
enum { TABLE_PGF_RLAY = 0, TABLE_PGF_GPRIME, TABLE_SVML_RLAY, TABLE_TD_RLAY, TABLE_TD_GPRIME, TABLE_COUNT = 8 };

}  // namespace m6

// The opaque context of the C ABI.
struct mom6hip_ctx {
  mom6hip_grid_t host;          // copy of the caller's struct (host metric pointers not retained)
  m6::GridDev g;                // device view
  hipStream_t stream = nullptr;
  std::vector<void *> metric_allocs;
  double *d_metric[64] = {};    // device copies in the order of the struct's pointer members

  // advect_tracer work space (device)
  m6::DevBuf hprev, uhr, vhr, flags, stage[16], tr_stage[64];
  m6::DevBuf pool[64];          // staging / scratch buffers handed out by m6::Stager, in call order
  m6::DevBuf rk2_scratch;       // the automatic arrays of step_MOM_dyn_split_RK2
  m6::DevBuf rk2_eta_PF_start;  // eta_PF_start of a step with p_surf_begin and p_surf_end (MOM_dynamics_split_RK2.F90:438)
  int rk2_scratch_layout = 0;   // which stepper laid the block out last (1: RK2, 2: RK2B, 3: RK2 with an OBC); a change of layout zeroes it again
  m6::DevBuf hv_pack;           // the grid metrics of horizontal_viscosity gathered into planes of one shape (hor_visc.hip)
  bool hv_pack_ready = false;
  m6::DevBuf hv_str;            // the layer-integrated stresses of every layer, kept for MEKE%mom_src (hor_visc.hip)
  m6::DevBuf cont_hmid;         // continuity in two phases around a pass in flight: the thicknesses after the first direction
  uint64_t overlap[4] = {0, 0, 0, 0};      // mom6hip_overlap_stats
  int cont_phase = 0;           // 0: the whole continuity; 1: what needs no halo row (before the completion); 2: the rest (after it)
  // u_bc_accel = (CAu + PFu) + diffu of the RK2 step (MOM_dynamics_split_RK2.F90:557-564, :879-886) formed by the kernel that produces the
  // later of CAu and PFu (pgf_face_kernel or coradcalc_kernel) instead of a sweep of its own: set by the stepper around that call; the
  // module sets `done` when its kernel has taken it (other forms of the module leave it, and the stepper launches the sweep)
  struct BcAccelFuse { const double *au, *av, *diffu, *diffv; double *u_bc, *v_bc; int inviscid; bool done; };
  BcAccelFuse *bc_fuse = nullptr;
  bool cont_fluxes_only = false;   // the caller reads the transports and BT_cont of this continuity call but not its thicknesses (the RK2 step's
                                   // first call, MOM_dynamics_split_RK2.F90:634: hp is rewritten by :757 before anything reads it): the second
                                   // direction's convergence is not launched
  m6::DevBuf sv_rlay;           // device copy of GV%Rlay for set_viscous_BBL (set_viscosity.hip)
  m6::DevBuf adv_gen;           // the general path of advect_tracer: the transports and fluxes of a pass, the segments' tables (tracer_advect.hip)
  m6::DevBuf adv_obc[64];       // device copies of the segments' tracer reservoirs handed over as host arrays
  int adv_obc_n = 0;
  m6::HostTable tables[m6::TABLE_COUNT];      // cached device copies of short host tables (m6::HostTable)
  uint64_t xfer[4] = {0, 0, 0, 0};            // calls and bytes of mom6hip_sync_to_device, then of mom6hip_sync_to_host / stage_to_host
  m6::DevBuf ale_sub;           // sub-cell structure of the two grids, handed from ale_sub_cells_kernel to the remap kernel
  m6::DevBuf ale_side;          // the streaming remap kernel's side arrays (target cells below what a lane has read)
  std::vector<const void *> lds_configured;      // kernels whose dynamic-LDS limit has been raised on this context's device
  // The device tables the operators derive from an OBC (face / corner / side maps, segment tables): built and uploaded once per
  // (operator site, fingerprint of the OBC) and kept -- the segments do not move after initialisation (m6::obc_table, open_boundary.hip)
  struct ObcTable { int site; uint64_t key; m6::DevBuf buf; size_t bytes; uint64_t last_use; std::vector<char> host; };
  std::vector<ObcTable> obc_tables;
  uint64_t obc_table_clock = 0, obc_table_builds = 0;
  m6::DevBuf vv_ntrunc;         // device counter of vertvisc_limit_vel's truncations (vert_friction.hip)
  bool vv_ntrunc_ready = false;
  // hipGraphs of the barotropic subcycle, keyed on everything baked into their nodes (barotropic.hip)
  std::vector<std::pair<std::string, void *>> bt_graphs;
  hipStream_t cap_stream = nullptr;
  static constexpr int NSIDE = 4;
  hipStream_t side_stream[NSIDE] = {nullptr, nullptr, nullptr, nullptr};      // small launches that run beside a large one of the compute
                                                                              // stream and beside each other (continuity's OBC strips)
  hipEvent_t side_fork = nullptr, side_join[NSIDE] = {nullptr, nullptr, nullptr, nullptr};      // compute -> side streams, each side -> compute
  long bt_graph_captures = 0, bt_graph_launches = 0, bt_graph_nodes_last = 0;
  int *h_domore_k = nullptr;    // pinned host mirror of domore_k
  m6_native_domain *native = nullptr;      // set by mom6hip_domain_init_rccl: the group pass and the reductions are the library's own
  // multi-tile collectives provided by the host (null on a one-tile domain)
  mom6hip_halo_fn halo_cb = nullptr;
  mom6hip_sum_fn sum_cb = nullptr;
  bool cb_stream_ordered = false;
  mom6hip_min_fn min_cb = nullptr;
  void *min_user = nullptr;
  void *cb_user = nullptr;
  int num_PEs = 0;              // asked of the domain on first use (coms.hip); 0: not yet known
  // MOM6HIP debugging of the non-blocking passes (mom6hip_debug_poison_passes): start_group_pass fills the halos the pass is going to
  // update with NaNs and leaves the update itself to complete_group_pass, so that anything enqueued between the two that reads a
  // halo too early is found deterministically (a NaN in its result), whatever the streams' timing happens to be
  bool poison_passes = false;
  bool split_rows_always = false;      // (tests: the interior / edge-band launches on one tile, where nothing is in flight)
  int poison_now = 0;           // (transient: the next halo kernel writes NaNs on these sides instead of the update)
  struct PendingPass { std::vector<double *> f; std::vector<int32_t> pos, nk; } pending;
  m6::DevBuf efp_acc;           // the accumulators of the extended-fixed-point sums (efp.hpp)
  // the depth list of MOM_sum_output (create_depth_list) and the remembered list positions CS%lH (sum_output.hip)
  std::vector<double> DL_depth, DL_area, DL_vol_below;
  std::vector<int> DL_lH;
  // restart / diagnostic staging (staging.hip): snapshots on the compute stream, device-to-host copies on a stream of their own
  struct StageSlot { m6::DevBuf buf; hipEvent_t snap = nullptr, done = nullptr; };
  std::vector<StageSlot> stage_slots;
  int stage_next = 0;
  hipStream_t copy_stream = nullptr;
  // timing
  bool ktiming = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> kt_events[MOM6HIP_KT_SLOTS];
  bool timing = false;
  mom6hip_advect_timing_t adv_timing = {};
};

namespace m6 {

// Gives a kernel driver device views of the caller's arrays.  MOM6HIP_MEM_DEVICE: the pointers are used
// as they are.  MOM6HIP_MEM_HOST: inputs are copied to pooled device buffers, outputs are copied back by
// finish().  The same host pointer always maps to the same device buffer (h may alias hin, ...).
// event pair around a kernel launch when per-kernel timing is on (mom6hip_kernel_timing)
struct KTimer {
  mom6hip_ctx *c; int slot; hipEvent_t e0 = nullptr, e1 = nullptr;
  KTimer(mom6hip_ctx *ctx, int s) : c(ctx), slot(s) {
    if (!c->ktiming) return;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { e0 = e1 = nullptr; return; }
    (void)hipEventRecord(e0, c->stream);
  }
  ~KTimer() {
    if (!e0) return;
    (void)hipEventRecord(e1, c->stream);
    c->kt_events[slot].push_back({e0, e1});
  }
};

// MOM_domains inside the library (context.hip, domain_rccl.hip).  do_group_pass = start + complete; with a native (RCCL)
// domain the exchange runs on the communication stream between the two, otherwise start does the whole pass on the compute
// stream (the host's callback, or the local wrap kernels) and complete does nothing.
int group_pass(mom6hip_ctx *ctx, double *const *fields, const int32_t *pos, const int32_t *nk, int n);
int start_group_pass(mom6hip_ctx *ctx, double *const *fields, const int32_t *pos, const int32_t *nk, int n);
int complete_group_pass(mom6hip_ctx *ctx);
// Restricts the rows every module computes (the launches take their ranges from ctx->g) to [j0, j1] of the compute domain, for
// the interior / boundary-strip launches around a non-blocking pass (dyn_split_rk2.hip); row_window_reset restores the tile.
// whether start_group_pass leaves the x halos final (the tile spans x): then only rows near the tile's y edges wait for the pass
bool pass_leaves_x_final(mom6hip_ctx *ctx);
void row_window(mom6hip_ctx *ctx, int j0, int j1);
void row_window_reset(mom6hip_ctx *ctx);
int halo_wrap_dir(mom6hip_ctx *ctx, double *f, int pos, int nk, int dir, hipStream_t stream);
int halo_fold_north(mom6hip_ctx *ctx, double *f, int pos_flags, int nk, hipStream_t stream);      // TRIPOLAR_N
int halo_pack_on(mom6hip_ctx *ctx, double *const *fields, const int32_t *pos, const int32_t *nk_each, const int32_t *a0, int32_t nfields,
                 int32_t dir, int32_t width, double *buf, int32_t pack, int64_t *count, hipStream_t stream);
int native_start_group_pass(mom6hip_ctx *ctx, double *const *fields, const int32_t *pos, const int32_t *nk, int n);
int native_complete_group_pass(mom6hip_ctx *ctx);
bool native_x_is_local(mom6hip_ctx *ctx);
void native_pass_sides(mom6hip_ctx *ctx, int sides[2]);      // which halos a native pass fills: per direction, bit 0 low side, bit 1 high side
int native_allreduce(mom6hip_ctx *ctx, void *values, int n, bool is_int_sum);
int native_allreduce_dev(mom6hip_ctx *ctx, void *dvalues, int n, int kind, void *host_copy);      // kind 0: int32 sum, 1: f64 min, 2: u64 min
// sum_across_PEs of n DEVICE int32 / min_across_PEs of n DEVICE uint64 (bit patterns of non-negative doubles), in place, ordered behind the
// compute stream's work so far; the compute stream sees the result.  host_copy (pinned, optional) receives it too and the call returns when
// it is there.  With the native domain nothing passes through the host on the way; with the host's callbacks the values are read back,
// reduced by the callback and sent up again.
int sum_across_PEs_dev(mom6hip_ctx *ctx, int32_t *dvalues, int n, int32_t *host_copy);
int min_across_PEs_dev_u64(mom6hip_ctx *ctx, unsigned long long *dvalues, int n, unsigned long long *host_copy);
void native_domain_destroy(mom6hip_ctx *ctx);
// sum_across_PEs of n host int32 / min_across_PEs of n host doubles, in place (the native domain, the host's callback, or nothing)
int sum_across_PEs(mom6hip_ctx *ctx, int32_t *values, int n);
int min_across_PEs(mom6hip_ctx *ctx, double *values, int n);
inline bool multi_tile(const mom6hip_ctx *ctx);
void staging_destroy(mom6hip_ctx *ctx);      // staging.hip

// horizontal_viscosity on device arrays (hor_visc.hip); called by the split RK2 step at :860 and :1543
// ob: the maps of the open boundaries (hor_visc.hip, HVFArgs), null without
struct HVObcDev {
  const int32_t *q = nullptr, *fu = nullptr, *fv = nullptr, *hu = nullptr, *hv = nullptr;
  const double *tang_u = nullptr, *tang_v = nullptr;      // OBC_COMPUTED_STRAIN: segment%tangential_vel at the q points (3-D), of N / S and of E / W segments
};
int horizontal_viscosity_dev(mom6hip_ctx *ctx, const mom6hip_hor_visc_cs_t *cs, const double *u, const double *v, const double *h,
                             double *diffu, double *diffv, const double *hu_cont, const double *hv_cont, const HVObcDev *ob = nullptr);

// set_viscous_BBL on device arrays (set_viscosity.hip); ob: what the OBC branches read (device arrays), null without OBC
struct BBLObcDev { const int32_t *side_u = nullptr, *side_v = nullptr; const double *D_u = nullptr, *D_v = nullptr, *mask_u = nullptr, *mask_v = nullptr; };
int set_viscous_BBL_dev(mom6hip_ctx *ctx, const mom6hip_set_visc_cs_t *cs, const double *u, const double *v, const double *h,
                        const double *T, const double *S, const mom6hip_eos_t *eos, double *bbl_thick_u, double *bbl_thick_v,
                        double *Kv_bbl_u, double *Kv_bbl_v, double *Ray_u, double *Ray_v, const BBLObcDev *ob = nullptr);

// the velocities of vertvisc_coef / vertvisc as the increment the RK2 step applies just before them:
// u = mask2dCu * (u0 + dtv * (a1u [+ a2u])), v likewise (device pointers; a2u / a2v may be null)
struct VelIncrement { const double *u0, *v0, *a1u, *a1v, *a2u, *a2v; double dtv; };
int vertvisc_step_inc(mom6hip_ctx_t *ctx, mom6hip_vertvisc_cs_t *cs, double *u, double *v, const double *h, const double *dz,
                      const double *taux, const double *tauy, const mom6hip_vertvisc_type_t *visc, double dt, int32_t update_velocities,
                      double *taux_bot, double *tauy_bot, double *visc_rem_u, double *visc_rem_v, const VelIncrement *inc, int32_t memspace);
// open boundaries (open_boundary.hip): the side maps of the zero-gradient projections, and the store of the specified segments' velocities
class Stager;
int obc_side_maps(mom6hip_ctx_t *ctx, Stager &st, const mom6hip_obc_t *obc, const int32_t **side_u, const int32_t **side_v, const char *who);
// A fingerprint of what the maps of an OBC depend on: its flags, the segments' directions, flags and index ranges, and the identity of
// the segnum arrays (their addresses and a sample of their values) -- not the segments' data arrays
uint64_t obc_fingerprint(const mom6hip_ctx_t *ctx, const mom6hip_obc_t *obc);
inline uint64_t obc_mix(uint64_t h, uint64_t v) { h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2); return h; }
// The device copy of a host table that `build` fills (bytes long), cached per (site, key): `build` runs, and the table is uploaded (one
// synchronising copy), only when the pair has not been seen on this context.  nullptr on failure (error set).
enum { OBC_SITE_SIDE = 1, OBC_SITE_CONT, OBC_SITE_CORAD, OBC_SITE_HORVISC, OBC_SITE_BT_CODES, OBC_SITE_SETVISC, OBC_SITE_RK2 };
const void *obc_table(mom6hip_ctx_t *ctx, int site, uint64_t key, size_t bytes, const std::function<int(void *)> &build);
// The same for a table the caller has just formed on the host (its loop over the segments also launches kernels, so it runs at every
// call): uploaded only when its CONTENT differs from what the (site, key) entry holds; the entry keeps its own host copy, so the upload
// never has to be waited for.
const void *obc_table_content(mom6hip_ctx_t *ctx, int site, uint64_t key, const void *host, size_t bytes);
int obc_store_specified(mom6hip_ctx_t *ctx, Stager &st, const mom6hip_obc_t *obc, double *d_u, double *d_v, const char *who);
// set_viscous_ML on device arrays (set_viscosity.hip); called by the split RK2 step at :592
int set_viscous_ML_dev(mom6hip_ctx_t *ctx, const mom6hip_set_visc_cs_t *cs, const double *u, const double *v, const double *h,
                       const double *T, const double *S, const mom6hip_eos_t *eos, const double *taux, const double *tauy,
                       const double *ustar, double *nkml_visc_u, double *nkml_visc_v, double dt);

class Stager {
 public:
  Stager(mom6hip_ctx *ctx, int memspace) : ctx_(ctx), host_(memspace == MOM6HIP_MEM_HOST) {}
  // in: read by the kernels; out: written; inout: both.  A null pointer stays null.
  template <typename T> const T *in(const T *p, size_t bytes) { return (const T *)map((void *)p, bytes, true, false); }
  template <typename T> T *out(T *p, size_t bytes) { return (T *)map((void *)p, bytes, false, true); }
  template <typename T> T *inout(T *p, size_t bytes) { return (T *)map((void *)p, bytes, true, true); }
  // device scratch that is not visible to the caller
  void *scratch(size_t bytes);
  bool failed() const { return fail_; }
  int finish();   // copies the outputs back (HOST) and synchronises the stream in that case
 private:
  void *map(void *p, size_t bytes, bool rd, bool wr);
  struct Ent { void *host, *dev; size_t bytes; bool wr; };
  mom6hip_ctx *ctx_;
  bool host_, fail_ = false;
  int next_ = 0;
  std::vector<Ent> ents_;
};

}  // namespace m6

namespace m6 {
inline bool multi_tile(const mom6hip_ctx *ctx) { return ctx->native != nullptr || ctx->halo_cb != nullptr; }
}  // namespace m6
