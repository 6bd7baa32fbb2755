/*
 * mom6hip.h -- C ABI of libmom6hip, the MI355X (gfx950) implementation of the MOM6
 * split-RK2 dynamical-core hot path.
 *
 * The reference (mnlevy1981/MOM6) has no FFI: its drop-in boundary is the public Fortran module
 * procedure.  Every entry point below replaces one such procedure; the Fortran side binds to it
 * through ISO_C_BINDING interfaces (mom6_amd/fortran/mom6hip_c_api.F90) in the style the reference
 * itself uses for libc (src/framework/posix.F90:52-229).  Plain pointers and sizes only.
 *
 * Conventions
 *  - All reals are IEEE fp64 (the reference promotes `real` with -fdefault-real-8).
 *  - Arrays are whole Fortran allocations, column-major, i fastest, symmetric memory
 *    (config_src/memory/dynamic_symmetric/MOM_memory.h):
 *        h-point  (isd:ied,   jsd:jed  [,nk])
 *        u-point  (isd-1:ied, jsd:jed  [,nk])      "SZIB_"
 *        v-point  (isd:ied,   jsd-1:jed[,nk])      "SZJB_"
 *        q-point  (isd-1:ied, jsd-1:jed[,nk])
 *  - `memspace` says where the field pointers of a call live: MOM6HIP_MEM_HOST (the Fortran
 *    drop-in: arrays are staged to HBM, the kernels run, results are copied back to the same host
 *    arrays) or MOM6HIP_MEM_DEVICE (state already resident in HBM; no copies).  The 2-D metric
 *    arrays in mom6hip_grid_t are always HOST pointers; mom6hip_grid_create uploads them once.
 *  - Every function returns 0 on success, nonzero on error; mom6hip_last_error() gives the text.
 *    The Fortran shim turns nonzero into MOM_error(FATAL, ...) (src/framework/MOM_error_handler.F90:148).
 *  - One process <-> one tile <-> one GPU; calls come from one thread, never re-entrantly.
 */
#ifndef MOM6HIP_H
#define MOM6HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOM6HIP_MEM_HOST   0
#define MOM6HIP_MEM_DEVICE 1

/* TRACER_ADVECTION_SCHEME, src/tracer/MOM_tracer_advect.F90:1116-1134 */
#define MOM6HIP_ADV_PLM    0
#define MOM6HIP_ADV_PPM_H3 1   /* "PPM:H3": usePPM & useHuynh   */
#define MOM6HIP_ADV_PPM    2   /* "PPM"   : usePPM & .not.useHuynh (Colella-Woodward) */

/*
 * Horizontal grid, index ranges and the scalar constants the kernels read.  Mirrors the members of
 * ocean_grid_type (src/core/MOM_grid.F90), hor_index_type (src/framework/MOM_hor_index.F90) and
 * verticalGrid_type (src/core/MOM_verticalGrid.F90) that the hot path touches.
 * Index ranges are Fortran-style inclusive, in the tile-local numbering of hor_index_type.
 * Unused metric pointers may be NULL; a kernel that needs a NULL metric fails with an error.
 */
typedef struct mom6hip_grid {
  int32_t isc, iec, jsc, jec;       /* compute domain */
  int32_t isd, ied, jsd, jed;       /* data domain (compute + halo) */
  int32_t nk;                       /* GV%ke */
  int32_t symmetric;                /* must be 1: u/v/q arrays start at isd-1 / jsd-1 */
  int32_t reentrant_x, reentrant_y; /* REENTRANT_X / REENTRANT_Y of a single-tile domain */
  int32_t first_direction;          /* G%first_direction */
  int32_t tripolar_n;               /* TRIPOLAR_N (src/framework/MOM_domains.F90:189): the northern edge of the domain folds onto
                                     * itself, cell (i, nj+m) being cell (ni+1-i, nj+1-m); needs reentrant_x and one tile in x */
  double Angstrom_H;                /* GV%Angstrom_H */
  double H_subroundoff;             /* GV%H_subroundoff */
  double dZ_subroundoff;            /* GV%dZ_subroundoff */
  double H_to_Z, Z_to_H;            /* GV%H_to_Z, GV%Z_to_H */
  double g_Earth, Rho0;             /* GV%g_Earth, GV%Rho0 */
  double reserved1[8];
  /* h-points */
  const double *mask2dT, *areaT, *IareaT, *dxT, *dyT, *IdxT, *IdyT, *bathyT;
  /* u-points */
  const double *mask2dCu, *dxCu, *dyCu, *dy_Cu, *IdxCu, *IdyCu, *areaCu, *IareaCu;
  /* v-points */
  const double *mask2dCv, *dxCv, *dyCv, *dx_Cv, *IdxCv, *IdyCv, *areaCv, *IareaCv;
  /* q-points */
  const double *mask2dBu, *dxBu, *dyBu, *areaBu, *IareaBu, *CoriolisBu, *IdxBu, *IdyBu;
  const void *reserved2[6];
} mom6hip_grid_t;

/* Opaque handle: device copies of the metrics + scratch owned by the library. */
typedef struct mom6hip_ctx mom6hip_ctx_t;
struct mom6hip_obc;      /* the open boundaries: mom6hip_obc_t, below */

/* ---- library / context ------------------------------------------------------------------- */

/* Selects the HIP device (hipSetDevice).  Fails loudly if no gfx950 device is usable. */
int mom6hip_init(int device);

/* Text of the last error on this thread's context ("" if none). */
const char *mom6hip_last_error(void);

/* Uploads the metric arrays of `grid` (host pointers) and returns a context.  `stream` is the
 * hipStream_t (as void*) all work of this context is enqueued on; NULL = the default stream. */
int mom6hip_grid_create(const mom6hip_grid_t *grid, void *stream, mom6hip_ctx_t **ctx);
int mom6hip_grid_destroy(mom6hip_ctx_t *ctx);

/* Blocks until all work enqueued by this context is done (hipStreamSynchronize). */
int mom6hip_sync(mom6hip_ctx_t *ctx);

/* Plain device allocation helpers for hosts without another device allocator (Fortran driver). */
int mom6hip_malloc(void **dptr, uint64_t bytes);
int mom6hip_free(void *dptr);
/* zeroes `bytes` of device memory on the context's stream (what ALLOC_ / allocate(..., source=0.0) do for a device mirror) */
int mom6hip_memset_zero(mom6hip_ctx_t *ctx, void *dptr, uint64_t bytes);
int mom6hip_sync_to_device(mom6hip_ctx_t *ctx, void *dptr, const void *hptr, uint64_t bytes);
int mom6hip_sync_to_host(mom6hip_ctx_t *ctx, void *hptr, const void *dptr, uint64_t bytes);
/* stats[0..3] = calls and bytes of mom6hip_sync_to_device, calls and bytes of mom6hip_sync_to_host + mom6hip_stage_to_host on this
 * context since its creation or the last reset: what a device-resident host (MOM_dynamics_split_RK2_hip.F90) moved over PCIe. */
int mom6hip_transfer_stats(mom6hip_ctx_t *ctx, uint64_t *stats, int32_t reset);
/* start_group_pass / complete_group_pass (MOM_domain_infra.F90:1141-1182) on device fields (pos: MOM6HIP_POS_* | MOM6HIP_PASS_SCALAR_PAIR,
 * nk_each: layers of each field): between the two calls the halos are in flight -- nothing may write the fields or read their halos.
 * With the library's own domain the exchange runs on its communication stream; a tile that spans x has its x halos final when
 * start returns.  mom6hip_halo_update is the two calls back to back. */
int mom6hip_start_group_pass(mom6hip_ctx_t *ctx, double *const *fields, const int32_t *pos, const int32_t *nk_each, int32_t nfields);
int mom6hip_complete_group_pass(mom6hip_ctx_t *ctx);
/* Debugging of the non-blocking group passes (start_group_pass / complete_group_pass, MOM_domain_infra.F90:1141-1182): with
 * enable != 0 every start fills the halos the pass is going to update with NaNs and leaves the update to the matching complete, so
 * that a kernel enqueued between the two that reads one of those halos too early yields NaNs deterministically, whatever the
 * timing of the streams.  One tile and the library's own RCCL domain; not with the host's halo callback. */
int mom6hip_debug_poison_passes(mom6hip_ctx_t *ctx, int32_t enable);
/* How often the RK2 step has worked around a pass in flight since the context was made (or since the last reset): stats[0] the
 * row-split launches of h_av / horizontal_viscosity / CorAdCalc / the accumulations (inner rows before the completion, the edge bands
 * after it), stats[1] the continuity calls made in two phases, stats[2] the passes completed before anything else ran (one-tile
 * domains, tiles too short to split, G%first_direction = 1, the tripolar fold), stats[3] reserved. */
int mom6hip_overlap_stats(mom6hip_ctx_t *ctx, uint64_t *stats, int32_t reset);

/* ---- restart / diagnostic staging: fields to the host while the model keeps stepping ------ */

/* The host arrays MOM6 registers for restarts and diagnostics (register_restart_field, src/framework/MOM_restart.F90:
 * the u, v, h, T, S ... that save_restart writes; the arrays post_data averages, src/framework/MOM_diag_mediator.F90) are
 * filled from the device-resident fields without a stop of the compute stream:
 *   mom6hip_stage_to_host   snapshots `bytes` of the device field on the context's stream (a device-to-device copy; the
 *                           field may be overwritten by whatever is enqueued next) and queues the device-to-host copy of
 *                           the snapshot on the context's copy stream.  Returns at once.  Any number of fields per batch.
 *   mom6hip_stage_query     the number of staged copies of the current batch that have not landed yet (never blocks).
 *   mom6hip_stage_wait      blocks until every staged copy has landed and closes the batch: call it where the host reads
 *                           the arrays (before save_restart, before a diagnostic is posted).
 *   mom6hip_host_register / _unregister   page-lock a host array once (hipHostRegister), so that its copies run at the
 *                           full PCIe rate and truly asynchronously; staging works on pageable arrays too, more slowly. */
int mom6hip_host_register(void *hptr, uint64_t bytes);
int mom6hip_host_unregister(void *hptr);
int mom6hip_stage_to_host(mom6hip_ctx_t *ctx, void *hptr, const void *dptr, uint64_t bytes);
int mom6hip_stage_query(mom6hip_ctx_t *ctx, int32_t *pending);
int mom6hip_stage_wait(mom6hip_ctx_t *ctx);

/* The bandwidth the device delivers to plain streaming kernels, measured where the model runs (a = b and a = b + s*c over arrays
 * of bytes_per_array bytes, `reps` launches each, HIP events; GB/s counting 2 and 3 arrays of traffic): the measured companion of
 * the nominal HBM peak that the roofline fractions of bench.py are quoted against. */
int mom6hip_stream_bandwidth(mom6hip_ctx_t *ctx, uint64_t bytes_per_array, int32_t reps, double *copy_GBs, double *triad_GBs);

/* The launch floor of the barotropic subcycle, measured where the model runs: a hipGraph of `nodes` dependent kernel nodes, each
 * a one-pass read-modify-write of `points` doubles (the 2-D tile), replayed `reps` times on the context's stream between HIP events;
 * microseconds per node.  bench.py quotes `us per barotropic step` against kernels-per-step times this. */
int mom6hip_graph_node_floor(mom6hip_ctx_t *ctx, int32_t nodes, int64_t points, int32_t reps, double *us_per_node);

/* ---- MOM_checksums: the bit-count checksum of a field, on the device ----------------------- */

/* subchk of chksum_h_3d / chksum_u_3d / chksum_v_3d / chksum_B_3d (src/framework/MOM_checksums.F90:1387-1401, :1042-1059,
 * :1228-1245, :739-756) and the extrema of subStats (:1403-1425): over the points (isc+di : iec+di, jsc+dj : jec+dj) of the
 * field's own indexing (the reference deliberately uses the h-point computational domain for every staggering; symmetric
 * /= 0 extends it by the western / southern face row as its `symmetric` argument does) and all nk layers,
 *     *bitcount = mod( sum popcnt(transfer(abs(scale*x), 1_8)), 1000000000 )     (summed over PEs when a domain is attached)
 * -- the number debugging runs of MOM6 print and compare across builds, here without moving the field off the GPU.
 * (The reference accumulates in a default integer, which wraps above 2^31 set bits per PE; this sum is exact.)
 * amin / amax may be NULL.  `field` is a device or host array of staggering `pos` (MOM6HIP_POS_*), nk layers. */
int mom6hip_chksum(mom6hip_ctx_t *ctx, const double *field, int32_t pos, int32_t nk, int32_t di, int32_t dj, int32_t symmetric,
                   double scale, int64_t *bitcount, double *amin, double *amax, int32_t memspace);

/* ---- MOM_coms: the order-invariant (extended-fixed-point) sum of a field, on the device ---- */

/* reproducing_sum(array(isc:iec,jsc:jec,:), sums, EFP_sum, EFP_lay_sums, err) of src/framework/MOM_coms.F90:318 (and, with
 * nk = 1, reproducing_sum_2d :219) for a field that stays on the GPU: every value is split into the six 46-bit integer limbs
 * of real_to_ints (:508), the limbs are added as integers -- on the device, in no particular order, which integer addition
 * allows -- summed over PEs when a domain is attached, regularised (:643) and converted back (:545).  The result is the
 * reference's to the bit and does not depend on the decomposition (Hallberg & Adcroft 2014).  The sum runs over the
 * h-point computational domain in the field's own indexing for every staggering, as the means of MOM_checksums do
 * (MOM_checksums.F90:1079), and over all nk layers.
 *   sum       the total.  As in the reference it is the regularised EFP total converted to a real when neither lay_sums
 *             nor efp_lay is given, and the floating-point sum of the layer values (k ascending) when either is (:421-427).
 *   lay_sums  [nk] by-layer sums, or NULL.          efp_lay  [6*nk] their EFP integers, or NULL.
 *   efp_sum   [6] the EFP integers of the total, or NULL.
 *   npoints   the number of values summed, over all PEs (the divisor of subStats' means), or NULL.
 *   err       the reference's code: 0, +1 a term too large to represent, +2 overflow of the sum, +2 a NaN in the field
 *             (reproducing_sum_2d counts the NaN as +4 instead); when it is non-zero this PE contributes zeros.  With
 *             err == NULL those conditions are errors (the reference's FATAL). */
int mom6hip_reproducing_sum(mom6hip_ctx_t *ctx, const double *field, int32_t pos, int32_t nk, double *sum, double *lay_sums,
                            int64_t *efp_sum, int64_t *efp_lay, int64_t *npoints, int32_t *err, int32_t memspace);

/* ---- MOM_sum_output: the global integrals of write_energy (the numbers of ocean.stats) ---- */

typedef struct mom6hip_energy_sums {
  double mass_tot;        /* [kg]   reproducing_sum of h*(H_to_kg_m2*areaT*mask2dT), MOM_sum_output.F90:503-510 */
  double KE_tot;          /* [J]    :683-689 */
  double PE_tot;          /* 0 here: the available potential energy comes from mom6hip_write_energy_ape */
  double toten;           /* KE_tot + PE_tot :691 */
  double Salt, Heat;      /* [ppt kg], [J]: EFP_to_real of the summed column integrals :693-712 (0 without T and S) */
  double max_CFL[2];      /* the transport-based and the linear maximum CFL number :718-744 */
  int64_t mass_EFP[6], salt_EFP[6], heat_EFP[6];      /* the extended-fixed-point integers of the three totals */
  int64_t npoints;        /* cells summed, over all PEs */
} mom6hip_energy_sums_t;

/* The sums of write_energy(u, v, h, tv, day, n, G, GV, US, CS, tracer_CSp) (src/diagnostics/MOM_sum_output.F90:428) for
 * device-resident (or host) fields: every total is an order-invariant extended-fixed-point sum (MOM_coms reproducing_sum), so
 * the numbers are those the reference writes to ocean.stats, bit for bit and on any layout.  dt = CS%dt_in_T; C_p = tv%C_p;
 * T and S may both be NULL (use_temperature = False).  mass_lay, KE_lay: [nk] by-layer sums, or NULL.  Boussinesq. */
int mom6hip_write_energy_sums(mom6hip_ctx_t *ctx, const double *u, const double *v, const double *h, const double *T, const double *S,
                              double dt, double C_p, double H_to_kg_m2, double *mass_lay, double *KE_lay, mom6hip_energy_sums_t *out,
                              int32_t memspace);

/* The available potential energy of write_energy (CALCULATE_APE = True, the reference's default; :610-680):
 *   mom6hip_depth_list_create   create_depth_list (:1109-1232) + depth_list_setup (:1101-1103): the sorted list of bottom depths
 *                               with the ocean area at and the volume below each, kept in the context.  The tile's place in the
 *                               global domain (niglobal x njglobal cells, the tile's first cell at (i_offset, j_offset), 0-based)
 *                               orders the list as the reference's; the lists of the other PEs come through the domain's sum.
 *   mom6hip_write_energy_ape    the reference height of every interface (Z_0APE [nk+1], a search of the list from the layer
 *                               volumes vol_lay = H_to_Z/H_to_kg_m2 * mass_lay), the integrand PE_pt on the device, PE [nk+1] and
 *                               PE_tot as extended-fixed-point sums.  mass_lay from mom6hip_write_energy_sums; g_prime [nk+1] =
 *                               GV%g_prime.  toten of ocean.stats is KE_tot + PE_tot.  Boussinesq. */
int mom6hip_depth_list_create(mom6hip_ctx_t *ctx, int32_t niglobal, int32_t njglobal, int32_t i_offset, int32_t j_offset, double Z_ref,
                              double min_depth_inc, int32_t *listsize);
int mom6hip_write_energy_ape(mom6hip_ctx_t *ctx, const double *h, const double *mass_lay, const double *g_prime, double Rho0,
                             double H_to_kg_m2, double Z_ref, double *PE, double *PE_tot, double *Z_0APE, int32_t memspace);

/* ---- MOM_domains: single-tile halo update ------------------------------------------------- */

/* Staggering of a field, for halo updates (MOM_domains AGRID/CGRID_NE positions). */
#define MOM6HIP_POS_H 0
#define MOM6HIP_POS_U 1
#define MOM6HIP_POS_V 2
#define MOM6HIP_POS_Q 3
/* ORed into the position of a u- or v-point field in a group pass: the field is one of a SCALAR_PAIR (To_All+Scalar_Pair,
 * e.g. visc_rem_u / visc_rem_v, Datu / Datv) and does not change sign across the tripolar fold, as a vector component does */
#define MOM6HIP_PASS_SCALAR_PAIR 8

/* pass_var / pass_vector on a one-tile domain (config_src/infra/FMS2/MOM_domain_infra.F90:171,660):
 * fills the halo of `nfields` device arrays from the tile's own compute domain where the
 * domain is re-entrant; halos at closed edges are left untouched.  nk_each[f] is 1 for 2-D. */
int mom6hip_halo_update(mom6hip_ctx_t *ctx, double *const *fields, const int32_t *pos,
                        const int32_t *nk_each, int32_t nfields);

/* Multi-tile domains.  The library itself never talks to another process: when the tile has neighbours,
 * the host registers the two collective operations of MOM_domains / MOM_coms that happen INSIDE library
 * calls (the group pass and sum_across_PEs in advect_tracer, src/tracer/MOM_tracer_advect.F90:206,305).
 * In this repository they are torch.distributed point-to-point / all-reduce over RCCL (mom6_amd/domains.py);
 * in a MOM6 executable they would be MOM6's own do_group_pass / sum_across_PEs on the device buffers.
 *   halo_fn: fill the halos of `nfields` DEVICE arrays (positions `pos`, layer counts `nk`); the library has
 *            synchronised its stream before the call and continues on it after the call returns.  `pos` is the staggering
 *            (pos & 3), ORed with MOM6HIP_PASS_SCALAR_PAIR for the u/v members of a SCALAR_PAIR; halo_fn does the complete
 *            update, the tripolar fold included (a vector component changes sign across it, a scalar pair does not).
 *   sum_fn:  element-wise sum over all PEs of `n` HOST int32 values, in place.
 * Passing NULLs restores the one-tile behaviour. */
typedef int (*mom6hip_halo_fn)(void *user, double *const *fields, const int32_t *pos, const int32_t *nk, int32_t nfields);
typedef int (*mom6hip_sum_fn)(void *user, int32_t *values, int32_t n);
int mom6hip_set_domain_callbacks(mom6hip_ctx_t *ctx, mom6hip_halo_fn halo_fn, mom6hip_sum_fn sum_fn, void *user);
/* stream_ordered = 1: halo_fn enqueues its work on (or in order with) the context's stream, so the library does not
 * synchronise the stream before calling it (default 0: it does). */
int mom6hip_set_callback_stream_ordered(mom6hip_ctx_t *ctx, int32_t stream_ordered);
/* min_across_PEs (MOM_coms) of `n` HOST doubles, in place: used by set_dtbt (src/core/MOM_barotropic.F90:2915). */
typedef int (*mom6hip_min_fn)(void *user, double *values, int32_t n);
int mom6hip_set_min_callback(mom6hip_ctx_t *ctx, mom6hip_min_fn min_fn, void *user);

/* The native multi-tile domain: one process = one tile = one GPU, the group passes and reductions inside library calls go
 * over RCCL point-to-point on a second HIP stream (mom6_amd/csrc/domain_rccl.hip) -- no host callback, no host
 * synchronisation.  Rank r of `nranks` owns the tile whose neighbours in W / E / S / N are the given ranks (-1: a closed
 * edge, or a direction with one tile, where the grid's reentrant_x / reentrant_y select the library's own wrap).
 * Rank 0 obtains the 128-byte RCCL unique id and the host distributes it (MPI_Bcast in MOM6, torch.distributed here);
 * every rank then calls mom6hip_domain_init_rccl.  The callback path above stays available (an MPI host). */
typedef struct mom6hip_domain {
  int32_t nranks, rank;
  int32_t nbr_w, nbr_e, nbr_s, nbr_n;
  int32_t reserved[2];
} mom6hip_domain_t;
int mom6hip_rccl_get_unique_id(void *id, int32_t nbytes);
int mom6hip_domain_init_rccl(mom6hip_ctx_t *ctx, const mom6hip_domain_t *dom, const void *unique_id, int32_t nbytes);
/* Device time of the exchanges (pack + RCCL + unpack on the communication stream) since the previous call, and their
 * number; `enable` switches the recording for the calls that follow. */
int mom6hip_domain_exchange_timing(mom6hip_ctx_t *ctx, int32_t enable, double *ms_total, int64_t *npasses);

/* Packing for the multi-tile group pass: gathers (pack = 1) the slabs [a0[f], a0[f] + width) along direction `dir`
 * (0: i, rows restricted to the compute rows of each field; 1: j, full rows) of `nfields` DEVICE arrays into the
 * contiguous DEVICE buffer `buf`, field after field, or scatters them back (pack = 0).  a0 is 0-based in the
 * allocated array of each field.  Returns the number of doubles moved in *count (may be NULL).  This is what a
 * do_group_pass packs into one message per neighbour (mom6_amd/domains.py sends `buf` with torch.distributed). */
int mom6hip_halo_pack(mom6hip_ctx_t *ctx, double *const *fields, const int32_t *pos, const int32_t *nk_each,
                      const int32_t *a0, int32_t nfields, int32_t dir, int32_t width, double *buf, int32_t pack,
                      int64_t *count);

/* ---- MOM_tracer_advect -------------------------------------------------------------------- */

/* tracer_advect_CS, src/tracer/MOM_tracer_advect.F90:30-40 */
typedef struct mom6hip_tracer_advect_cs {
  double dt;                 /* CS%dt: baroclinic time step [T] */
  int32_t scheme;            /* MOM6HIP_ADV_* */
  int32_t use_huynh_stencil_bug; /* CS%useHuynhStencilBug (default 0) */
} mom6hip_tracer_advect_cs_t;

/* What advect_tracer reports back (not in the reference signature; used by tests and the bench). */
typedef struct mom6hip_advect_stats {
  int32_t iterations;        /* passes of the itt loop executed (:203) */
  int32_t halo_updates;      /* do_group_pass calls (:206) */
  int32_t domore_remaining;  /* sum(domore_k) when the loop ended */
  int32_t reserved;
} mom6hip_advect_stats_t;

/*
 * advect_tracer(h_end, uhtr, vhtr, OBC, dt, G, GV, US, CS, Reg, x_first_in, vol_prev, max_iter_in,
 *               update_vol_prev, uhr_out, vhr_out)            src/tracer/MOM_tracer_advect.F90:52
 *
 *  tr[m]            Reg%Tr(m)%t, h-point 3-D, updated in place, m = 0..ntr-1
 *  conc_underflow   Reg%Tr(m)%conc_underflow per tracer (NULL = all zero)
 *  x_first_in       <0 = absent (use mod(G%first_direction,2)==0), 0/1 = value
 *  vol_prev         NULL = absent (hprev is reconstructed from h_end and the fluxes, :160-172)
 *  max_iter_in      <=0 = absent
 *  update_vol_prev  0/1 (only meaningful with vol_prev)
 *  uhr_out/vhr_out  NULL = absent
 *  OBC must not be associated (open boundaries are out of scope; the shim FATALs).
 */
int mom6hip_advect_tracer(mom6hip_ctx_t *ctx, const double *h_end, const double *uhtr,
                          const double *vhtr, double dt, const mom6hip_tracer_advect_cs_t *cs,
                          double *const *tr, const double *conc_underflow, int32_t ntr,
                          int32_t x_first_in, double *vol_prev, int32_t max_iter_in,
                          int32_t update_vol_prev, double *uhr_out, double *vhr_out,
                          int32_t memspace, mom6hip_advect_stats_t *stats);
/* advect_tracer with OBC associated.  advect_x / advect_y read of the OBC the tracer registries of its segments (segment%tr_Reg,
 * MOM_tracer_advect.F90:441-477, :580-627, :823-861, :965-1014: the reservoir or inflow value of a registered tracer in the cell outside a
 * segment, the slopes about the segment's face, the inflow fluxes).  Without a registry on any segment the call is mom6hip_advect_tracer;
 * with one it takes the library's general kernels (a thread a face, a thread a cell).  obc == NULL: mom6hip_advect_tracer. */
int mom6hip_advect_tracer_obc(mom6hip_ctx_t *ctx, const double *h_end, const double *uhtr, const double *vhtr, double dt,
                              const mom6hip_tracer_advect_cs_t *cs, double *const *tr, const double *conc_underflow, int32_t ntr,
                              int32_t x_first_in, double *vol_prev, int32_t max_iter_in, int32_t update_vol_prev, double *uhr_out,
                              double *vhr_out, const struct mom6hip_obc *obc, int32_t memspace, mom6hip_advect_stats_t *stats);

/* Per-kernel device time of the last advect_tracer call, from HIP events on the context's stream:
 * ms_x / ms_y are the summed durations of the advect_x / advect_y kernels, n_x / n_y their launch
 * counts, ms_total the whole call.  Only filled when timing was enabled before the call. */
typedef struct mom6hip_advect_timing {
  double ms_total, ms_setup, ms_x, ms_y, ms_halo;
  double ms_x1, ms_y1;       /* the launches of the first iteration only (every row active) */
  int32_t n_x, n_y;
} mom6hip_advect_timing_t;
int mom6hip_set_timing(mom6hip_ctx_t *ctx, int32_t enable);

/* Per-kernel HIP-event timing on the context's stream for the roofline line of bench.py.  Slots:
 * 0 = cont_flux_kernel<0> (zonal), 1 = cont_flux_kernel<1> (meridional).  mom6hip_kernel_timing(enable = 1) starts
 * recording (an event pair around every launch of the slot's kernel); a later call returns the summed duration and the
 * number of launches since then (it synchronises the stream) and, with enable = 0, stops recording. */
#define MOM6HIP_KT_CONT_FLUX_X 0
#define MOM6HIP_KT_CONT_FLUX_Y 1
#define MOM6HIP_KT_BT_SUBCYCLE 2      /* the barotropic time steps of one btstep call (the replay of their hipGraph, one tile) */
#define MOM6HIP_KT_PGF_FACE 3         /* pgf_face_kernel of PressureForce_FV_Bouss */
#define MOM6HIP_KT_SLOTS 4
int mom6hip_kernel_timing(mom6hip_ctx_t *ctx, int32_t enable, double *ms_total, int64_t *launches);
int mom6hip_advect_get_timing(mom6hip_ctx_t *ctx, mom6hip_advect_timing_t *t);

/* ---- MOM_ALE / MOM_remapping ----------------------------------------------------------------- */

/* REMAPPING_SCHEME values, src/ALE/MOM_remapping.F90:50-59 (the ones libmom6hip provides) */
#define MOM6HIP_REMAP_PCM     0
#define MOM6HIP_REMAP_PLM     2
#define MOM6HIP_REMAP_PPM_H4  4
#define MOM6HIP_REMAP_PLM_HYBGEN  3
#define MOM6HIP_REMAP_PPM_IH4 5
#define MOM6HIP_REMAP_PPM_HYBGEN  6
#define MOM6HIP_REMAP_WENO_HYBGEN 7
#define MOM6HIP_REMAP_PQM_IH4IH3 8
#define MOM6HIP_REMAP_PQM_IH6IH5 9
#define MOM6HIP_REMAP_PPM_CW  10

/* remapping_CS, src/ALE/MOM_remapping.F90:25-41 */
typedef struct mom6hip_remapping_cs {
  int32_t remapping_scheme;         /* MOM6HIP_REMAP_* */
  int32_t boundary_extrapolation;   /* REMAP_BOUNDARY_EXTRAP */
  int32_t force_bounds_in_subcell;  /* REMAP_BOUND_INTERMEDIATE_VALUES (must be 0) */
  int32_t answer_date;              /* REMAPPING_ANSWER_DATE (must be >= 20190101) */
} mom6hip_remapping_cs_t;

/*
 * ALE_remap_tracers(CS, G, GV, h_old, h_new, Reg, debug, dt, PCM_cell)    src/ALE/MOM_ALE.F90:737
 * Remaps every tracer column of the compute domain (ocean points only) from the grid h_old to the
 * grid h_new with remapping_core_h (src/ALE/MOM_remapping.F90:160).  tr[m] is updated in place.
 * PCM_cell (hybgen only) and the tendency diagnostics (dt) are not provided.
 */
int mom6hip_ale_remap_tracers(mom6hip_ctx_t *ctx, const mom6hip_remapping_cs_t *cs, const double *h_old,
                              const double *h_new, double *const *tr, const double *conc_underflow,
                              int32_t ntr, int32_t memspace);

/* ---- ALE regridding (z*) and velocity remapping ------------------------------------------------ */

#define MOM6HIP_REGRIDDING_ZSTAR 2   /* regrid_consts.F90:14 */

/* regridding_CS, src/ALE/MOM_regridding.F90:50-150, the members the z* branch of regridding_main reads */
typedef struct mom6hip_regridding_cs {
  int32_t regridding_scheme;             /* must be MOM6HIP_REGRIDDING_ZSTAR */
  int32_t nk;                            /* CS%nk; must equal GV%ke */
  double min_thickness;                  /* MIN_THICKNESS [H] (default 1e-3 m) */
  double old_grid_weight;                /* REGRID_TIME_SCALE -> ALE_update_regrid_weights (default 0) */
  double depth_of_time_filter_shallow;   /* REGRID_FILTER_SHALLOW_DEPTH [H] (0) */
  double depth_of_time_filter_deep;      /* REGRID_FILTER_DEEP_DEPTH [H] (0) */
  double Z_ref;                          /* G%Z_ref */
  const double *coordinateResolution;    /* nk nominal thicknesses [Z]; always a HOST pointer */
} mom6hip_regridding_cs_t;

/*
 * ALE_regrid(G, GV, US, h, h_new, dzRegrid, tv, CS, frac_shelf_h, PCM_cell)      src/ALE/MOM_ALE.F90:484
 *   -> regridding_main (MOM_regridding.F90:763), REGRIDDING_ZSTAR: build_zstar_grid (:1174) with build_zstar_column
 *      (coord_zlike.F90:63), filtered_grid_motion (:1022), adjust_interface_motion (:1713), calc_h_new_by_dz (:925).
 * Boussinesq, no ice shelf.  h_new and dzRegrid (nk+1 levels) are written on (isc-1:iec+1, jsc-1:jec+1); dzRegrid is
 * zeroed everywhere first (:508).  The FATAL checks of the reference (negative thickness beyond roundoff) are not
 * evaluated on the device.
 */
int mom6hip_ale_regrid(mom6hip_ctx_t *ctx, const mom6hip_regridding_cs_t *cs, const double *h, double *h_new,
                       double *dzRegrid, int32_t memspace);

/* ALE_remap_set_h_vel(CS, G, GV, h_new, h_u, h_v, OBC, debug)                    src/ALE/MOM_ALE.F90:870
 * PARTIAL_CELL_VELOCITY_REMAP = False, OBC not associated.  Only open faces are written. */
int mom6hip_ale_remap_set_h_vel(mom6hip_ctx_t *ctx, const double *h_new, double *h_u, double *h_v, int32_t memspace);

/* ALE_remap_set_h_vel_via_dz(CS, G, GV, h_new, h_u, h_v, OBC, h_old, dzInterface, debug)   src/ALE/MOM_ALE.F90:912
 * What step_MOM_thermo calls for the NEW velocity-point grid when REMAP_UV_USING_OLD_ALG = True (src/core/MOM.F90:1666-1667;
 * .testing/tc2, tc4): h_u = max(0, mean(h_old) + half the difference of the mean interface movements); dzInterface has nk+1
 * levels (ALE_regrid's dzRegrid).  h_new is read by the reference only for OBC faces and is not an argument here. */
int mom6hip_ale_remap_set_h_vel_via_dz(mom6hip_ctx_t *ctx, const double *h_old, const double *dzInterface, double *h_u,
                                       double *h_v, int32_t memspace);

/* ALE_PLM_edge_values(CS, G, GV, h, Q, bdry_extrap, Q_t, Q_b)                      src/ALE/MOM_ALE.F90:1520
 * (TS_PLM_edge_values :1495 calls it for tv%S and tv%T.)  The top and bottom edge values of every layer of a 3-d scalar at h
 * points by the monotonised piecewise-linear reconstruction (PLM_slope_wa, PLM_monotonized_slope; the boundary cells piecewise
 * constant, or by PLM_extrapolate_slope with bdry_extrap), columns isc-1..iec+1 x jsc-1..jec+1, REMAPPING_ANSWER_DATE >= 20190101:
 * the edge values PressureForce_FV reconstructs T and S with. */
int mom6hip_ale_plm_edge_values(mom6hip_ctx_t *ctx, const double *h, const double *Q, int32_t bdry_extrap, double *Q_t, double *Q_b,
                                int32_t memspace);

/* ALE_remap_velocities(CS, G, GV, h_old_u, h_old_v, h_new_u, h_new_v, u, v, debug, dt, allow_preserve_variance)
 *                                                                                src/ALE/MOM_ALE.F90:1061
 * cs is CS%vel_remapCS; REMAP_VEL_CONSERVE_KE = False, REMAP_VEL_MASK_BBL_THICK <= 0, no KE diagnostics.
 * remap_dyn_split_RK2_aux_vars (MOM_dynamics_split_RK2.F90:1273) is this routine applied to (u_av, v_av),
 * (CAu_pred, CAv_pred) and (diffu, diffv). */
int mom6hip_ale_remap_velocities(mom6hip_ctx_t *ctx, const mom6hip_remapping_cs_t *cs, const double *h_old_u,
                                 const double *h_old_v, const double *h_new_u, const double *h_new_v, double *u,
                                 double *v, int32_t memspace);

/* ---- MOM_CoriolisAdv ----------------------------------------------------------------------- */

/* CORIOLIS_SCHEME / KE_SCHEME enumeration values, src/core/MOM_CoriolisAdv.F90:93-112 */
#define MOM6HIP_SADOURNY75_ENERGY 1
#define MOM6HIP_ARAKAWA_HSU90     2
#define MOM6HIP_ROBUST_ENSTRO     3
#define MOM6HIP_SADOURNY75_ENSTRO 4
#define MOM6HIP_ARAKAWA_LAMB81    5
#define MOM6HIP_AL_BLEND          6
#define MOM6HIP_PV_ADV_CENTERED   21
#define MOM6HIP_PV_ADV_UPWIND1    22
#define MOM6HIP_KE_ARAKAWA        10
#define MOM6HIP_KE_SIMPLE_GUDONOV 11
#define MOM6HIP_KE_GUDONOV        12

/* CoriolisAdv_CS, src/core/MOM_CoriolisAdv.F90:30-91 (the members the provided branches read) */
typedef struct mom6hip_coriolisadv_cs {
  int32_t coriolis_scheme;   /* CORIOLIS_SCHEME: SADOURNY75_ENERGY (default), ARAKAWA_HSU90, ROBUST_ENSTRO, SADOURNY75_ENSTRO,
                              * ARAKAWA_LAMB81, ARAKAWA_LAMB_BLEND (AL_BLEND): all six of the reference */
  int32_t ke_scheme;         /* KE_SCHEME: KE_ARAKAWA (default), KE_SIMPLE_GUDONOV, KE_GUDONOV */
  int32_t no_slip;           /* NOSLIP */
  int32_t bound_coriolis;    /* BOUND_CORIOLIS (CoriolisAdv_init sets it false with CORIOLIS_EN_DIS + SADOURNY75_ENERGY and
                              * with ROBUST_ENSTRO, :1155-1156; CorAdCalc applies what it is given) */
  int32_t coriolis_en_dis;   /* CORIOLIS_EN_DIS: the energy-dissipating bias of SADOURNY75_ENERGY (:326-333, :594-642) */
  int32_t pv_adv_scheme;     /* PV_ADV_SCHEME of ROBUST_ENSTRO: PV_ADV_CENTERED (default; 0 means the default), PV_ADV_UPWIND1 */
  int32_t reserved[2];
  double F_eff_max_blend;    /* CORIOLIS_BLEND_F_EFF_MAX (4.0), AL_BLEND only */
  double wt_lin_blend;       /* CORIOLIS_BLEND_WT_LIN (0.125, clipped to [1e-16, 1] by CoriolisAdv_init :1139), AL_BLEND only */
} mom6hip_coriolisadv_cs_t;

/*
 * CorAdCalc(u, v, h, uh, vh, CAu, CAv, OBC, AD, G, GV, US, CS, pbv, Waves)   src/core/MOM_CoriolisAdv.F90:125
 * OBC, Waves must not be associated, pbv must be all ones, AD diagnostics are not provided.
 * Needs the metrics mask2dT, areaT, IareaT, dxCu, IdxCu, areaCu, dyCv, IdyCv, areaCv, mask2dBu, IareaBu,
 * CoriolisBu (+ IdyCu, IdxCv with ROBUST_ENSTRO; dy_Cu, dx_Cv with CORIOLIS_EN_DIS).  CAu is written on (isc-1:iec, jsc:jec), CAv on (isc:iec, jsc-1:jec); other points are untouched.
 */
int mom6hip_coradcalc(mom6hip_ctx_t *ctx, const mom6hip_coriolisadv_cs_t *cs, const double *u, const double *v,
                      const double *h, const double *uh, const double *vh, double *CAu, double *CAv,
                      int32_t memspace);

/* radiation_open_bdry_conds(OBC, u_new, u_old, v_new, v_old, G, GV, US, dt)            src/core/MOM_open_boundary.F90:2196
 * The normal component: Orlanski radiation (segment%radiation: the phase speed from the two faces inside the boundary, capped by
 * OBC%rx_max = OBC_RADIATION_MAX, averaged in time with OBC%gamma_uv = OBC_RAD_VEL_WT into the restart fields OBC%rx_normal /
 * ry_normal), the gradient condition (segment%gradient), nudging towards segment%nudged_normal_vel; then
 * open_boundary_apply_normal_flow (:3337) and pass_vector(u_new, v_new).  segment%normal_vel is written.  The tangential forms
 * (segment%radiation_tan, %radiation_grad, %nudged_tan, %nudged_grad: :2403-2455 and its three twins) write segment%tangential_vel /
 * tangential_grad at the corner points of the segment.  Oblique radiation (segment%oblique: :2349-2383 and its three twins, with the
 * gradients along the boundary of gradient_at_q_points :3407 and the restart fields obc->rx_oblique_u ... cff_normal_v), with its
 * tangential forms OBLIQUE_TAN, OBLIQUE_GRAD (:2456-2556 and twins).
 * rx_normal (u points, 3-D) and ry_normal (v points, 3-D) may be NULL when gamma_uv >= 1. */
int mom6hip_radiation_open_bdry_conds(mom6hip_ctx_t *ctx, const struct mom6hip_obc *obc, double gamma_uv, double rx_max, double *rx_normal,
                                      double *ry_normal, double *u_new, const double *u_old, double *v_new, const double *v_old, double dt,
                                      int32_t memspace);
/* open_boundary_zero_normal_flow(OBC, G, GV, u, v) :3374 (the RK2 step applies it to the accelerations :566, :888) */
int mom6hip_open_boundary_zero_normal_flow(mom6hip_ctx_t *ctx, const struct mom6hip_obc *obc, double *u, double *v, int32_t memspace);

/* update_segment_tracer_reservoirs(G, GV, uhr, vhr, h, OBC, dt, Reg)                    src/core/MOM_open_boundary.F90:5373
 * (step_MOM_tracer_dyn calls it after advect_tracer and tracer_hordiff with the accumulated transports, MOM.F90:1447.)  The reservoirs
 * tres of the registered tracers of every segment with a registry, one backward-Euler step: outflow carries the tracer of the cell inside
 * (tr[ntr_index - 1]) into the reservoir, inflow restores it towards the external values t, with the inverse length scales
 * Tr_InvLscale_out / _in (zero: the value is taken over at once).  tr: the tracer arrays of the registry (read); tres and t of the
 * segments in the memory space of the call.  OBC%tres_x / tres_y (the restart copies, I_scale * tres) are the caller's to refresh. */
int mom6hip_update_segment_tracer_reservoirs(mom6hip_ctx_t *ctx, const double *uhr, const double *vhr, const double *h,
                                             const struct mom6hip_obc *obc, double dt, const double *const *tr, int32_t ntr,
                                             int32_t memspace);

/* CorAdCalc with OBC associated (:249-269 the areas across a segment, :337-420 the circulation and the thicknesses projected onto the
 * velocity points of a segment, :422-455 onto its corner points, gradKE :1037-1050); additionally needs the metrics dxBu, dyBu with
 * OBC%specified_vorticity.  obc == NULL: mom6hip_coradcalc. */
int mom6hip_coradcalc_obc(mom6hip_ctx_t *ctx, const mom6hip_coriolisadv_cs_t *cs, const struct mom6hip_obc *obc, const double *u,
                          const double *v, const double *h, const double *uh, const double *vh, double *CAu, double *CAv,
                          int32_t memspace);

/* ---- MOM_continuity_PPM --------------------------------------------------------------------- */

/* continuity_PPM_CS, src/core/MOM_continuity_PPM.F90:35-67; defaults of continuity_PPM_init :2679-2757 */
typedef struct mom6hip_continuity_cs {
  int32_t upwind_1st;        /* UPWIND_1ST_CONTINUITY      (False) */
  int32_t monotonic;         /* MONOTONIC_CONTINUITY       (False: positive-definite limiter) */
  int32_t simple_2nd;        /* SIMPLE_2ND_PPM_CONTINUITY  (False) */
  int32_t aggress_adjust;    /* CONT_PPM_AGGRESS_ADJUST    (False) */
  int32_t vol_CFL;           /* CONT_PPM_VOLUME_BASED_CFL  (= aggress_adjust) */
  int32_t better_iter;       /* CONT_PPM_BETTER_ITER       (True) */
  int32_t use_visc_rem_max;  /* CONT_PPM_USE_VISC_REM_MAX  (True) */
  int32_t marginal_faces;    /* CONT_PPM_MARGINAL_FACE_AREAS (True) */
  double tol_eta;            /* ETA_TOLERANCE [H]          (0.5*NK*Angstrom) */
  double tol_vel;            /* VELOCITY_TOLERANCE [L T-1] (3e8) */
  double CFL_limit_adjust;   /* CONTINUITY_CFL_LIMIT       (0.5) */
} mom6hip_continuity_cs_t;

/* BT_cont_type, src/core/MOM_variables.F90 (the members continuity_PPM sets).  2-D face arrays;
 * h_u / h_v are 3-D and may be NULL (they are only allocated for BT_THICK_SCHEME = FROM_BT_CONT). */
typedef struct mom6hip_bt_cont {
  double *FA_u_W0, *FA_u_WW, *FA_u_E0, *FA_u_EE, *uBT_WW, *uBT_EE;   /* u-points */
  double *FA_v_S0, *FA_v_SS, *FA_v_N0, *FA_v_NN, *vBT_SS, *vBT_NN;   /* v-points */
  double *h_u, *h_v;
} mom6hip_bt_cont_t;

/*
 * continuity_PPM(u, v, hin, h, uh, vh, dt, G, GV, US, CS, OBC, pbv, uhbt, vhbt, visc_rem_u, visc_rem_v,
 *                u_cor, v_cor, BT_cont, du_cor, dv_cor)                 src/core/MOM_continuity_PPM.F90:86
 * Optional arguments are NULL when absent.  h may be the same array as hin.  OBC must not be associated and
 * pbv must be all ones.  Metrics needed: mask2dT, areaT, IareaT, dxT, dyT, IdxT, IdyT, dy_Cu, dx_Cv, dxCu, dyCv,
 * mask2dCu, mask2dCv.
 */
int mom6hip_continuity(mom6hip_ctx_t *ctx, const mom6hip_continuity_cs_t *cs, const double *u, const double *v,
                       const double *hin, double *h, double *uh, double *vh, double dt, const double *uhbt,
                       const double *vhbt, const double *visc_rem_u, const double *visc_rem_v, double *u_cor,
                       double *v_cor, const mom6hip_bt_cont_t *BT_cont, double *du_cor, double *dv_cor,
                       int32_t memspace);

/*
 * Open boundaries (src/core/MOM_open_boundary.F90): what continuity_PPM reads of ocean_OBC_type (:266-386) and of its segments
 * (OBC_segment_type :146-263).  Round 4 provides the OBC branches of continuity_PPM (mom6hip_continuity_obc), of CorAdCalc
 * (mom6hip_coradcalc_obc), of vertvisc_coef / vertvisc (mom6hip_vertvisc_coef_obc, mom6hip_vertvisc_obc), of btcalc and btstep
 * (mom6hip_btcalc_obc, mom6hip_btstep_obc), of set_viscous_BBL (mom6hip_set_viscous_bbl_obc), of horizontal_viscosity
 * (mom6hip_horizontal_viscosity_obc), of advect_tracer (mom6hip_advect_tracer_obc: the tracer registries of the segments), and
 * radiation_open_bdry_conds / open_boundary_zero_normal_flow for the normal component, and of step_MOM_dyn_split_RK2 (cs->OBC of
 * mom6hip_dyn_split_rk2_cs_t); step_MOM_dyn_split_RK2b reads cs->OBC likewise (tracer_hordiff takes no OBC in the reference).  The segments' data
 * (update_OBC_segment_data) are the host's business: MOM_open_boundary stays the reference's.
 * Index ranges are in the local index space of the grid structure: isd, jsd and so on.
 */
#define MOM6HIP_OBC_NONE 0            /* OBC_NONE :79 */
#define MOM6HIP_OBC_DIRECTION_N 100   /* :80-83 */
#define MOM6HIP_OBC_DIRECTION_S 200
#define MOM6HIP_OBC_DIRECTION_E 300
#define MOM6HIP_OBC_DIRECTION_W 400
/* mom6hip_obc_segment_t.radiation_tan_or_grad: segment%radiation_tan (ORLANSKI_TAN), %radiation_grad (ORLANSKI_GRAD), %nudged_tan, %nudged_grad,
 * %oblique_tan, %oblique_grad */
#define MOM6HIP_OBC_TAN_RADIATION 1
#define MOM6HIP_OBC_GRAD_RADIATION 2
#define MOM6HIP_OBC_TAN_NUDGED 4
#define MOM6HIP_OBC_GRAD_NUDGED 8
#define MOM6HIP_OBC_TAN_OBLIQUE 16
#define MOM6HIP_OBC_GRAD_OBLIQUE 32

/* OBC_segment_tracer_type (:119-137) as advect_tracer reads it: one registered tracer of a segment's registry (segment%tr_Reg%Tr(m)) */
typedef struct mom6hip_obc_segment_tracer {
  int32_t ntr_index;           /* %ntr_index: which tracer of the registry (1-based: the order of the tracers handed to advect_tracer) */
  int32_t reserved;
  const double *tres;          /* %tres, the tracer reservoir on the segment's faces, (IsdB:IedB, jsd:jed, nk) for E / W, (isd:ied, JsdB:JedB, nk)
                                  for N / S, in the memory space of the call; NULL (not allocated): OBC_inflow_conc is used.  (Written by
                                  mom6hip_update_segment_tracer_reservoirs.) */
  double OBC_inflow_conc;      /* %OBC_inflow_conc */
  const double *t;             /* %t, the external tracer values on the segment's faces (the layout of tres): read by
                                  update_segment_tracer_reservoirs; may be NULL otherwise */
  double resrv_lfac_in, resrv_lfac_out;      /* segment%field(fd_index)%resrv_lfac_in / _out; 1.0 with fd_index = -1 (:5431-5437) */
} mom6hip_obc_segment_tracer_t;

typedef struct mom6hip_obc_segment {
  int32_t direction;       /* MOM6HIP_OBC_DIRECTION_* */
  int32_t open;            /* segment%open: open for the continuity solver */
  int32_t specified;       /* segment%specified: the normal velocity and transport are the external values */
  int32_t on_pe;           /* segment%on_pe */
  int32_t is_E_or_W, is_N_or_S;
  int32_t IsdB, IedB, JsdB, JedB;      /* segment%HI: the segment's face range on this PE's data domain */
  int32_t isd, ied, jsd, jed;          /* segment%HI: its cell range */
  int32_t radiation, gradient, nudged; /* segment%radiation (Orlanski), %gradient, %nudged: read by radiation_open_bdry_conds */
  int32_t oblique;                     /* segment%oblique: oblique radiation (:2349-2383), with obc->rx_oblique_u ... cff_normal_v */
  int32_t radiation_tan_or_grad;       /* the tangential forms, a bit each: MOM6HIP_OBC_TAN_* / _GRAD_* above */
  int32_t Flather;                     /* segment%Flather: read by btstep */
  /* segment%normal_trans, segment%normal_vel (IsdB:IedB, jsd:jed, nk) for E / W, (isd:ied, JsdB:JedB, nk) for N / S; read where
   * `specified`; in the memory space of the call; may be NULL otherwise */
  const double *normal_trans;
  double *normal_vel;                  /* (written by radiation_open_bdry_conds for radiation / gradient segments) */
  /* segment%tangential_vel, segment%tangential_grad (IsdB:IedB, JsdB:JedB, nk): read by CorAdCalc with OBC%computed_vorticity /
   * OBC%specified_vorticity; written by radiation_open_bdry_conds with the tangential forms; may be NULL otherwise */
  double *tangential_vel, *tangential_grad;
  const double *nudged_normal_vel;     /* segment%nudged_normal_vel, the layout of normal_vel: read where `nudged` */
  /* segment%normal_vel_bt, segment%SSH (IsdB:IedB, jsd:jed) for E / W, (isd:ied, JsdB:JedB) for N / S: the external barotropic velocity
   * and sea surface height of a Flather segment, read by btstep (set_up_BT_OBC); may be NULL otherwise */
  const double *normal_vel_bt, *SSH;
  double Velocity_nudging_timescale_in, Velocity_nudging_timescale_out;      /* [T] */
  /* segment%tr_Reg: NULL (not associated) or tr_Reg%ntseg entries, a HOST array; read by advect_tracer */
  const mom6hip_obc_segment_tracer_t *tr_Reg;
  int32_t ntseg, reserved_i;
  double Tr_InvLscale_in, Tr_InvLscale_out;      /* segment%Tr_InvLscale_in / _out [L-1]: read by update_segment_tracer_reservoirs */
  /* segment%nudged_tangential_vel, %nudged_tangential_grad (the layout of tangential_vel): read with MOM6HIP_OBC_TAN_NUDGED / _GRAD_NUDGED */
  const double *nudged_tangential_vel, *nudged_tangential_grad;
} mom6hip_obc_segment_t;

typedef struct mom6hip_obc {
  int32_t number_of_segments;
  int32_t OBC_pe;                                   /* OBC%OBC_pe: some segment touches this PE */
  int32_t open_u_BCs_exist_globally, open_v_BCs_exist_globally;
  int32_t specified_u_BCs_exist_globally, specified_v_BCs_exist_globally;
  int32_t Flather_u_BCs_exist_globally, Flather_v_BCs_exist_globally;
  int32_t zero_vorticity, freeslip_vorticity, computed_vorticity, specified_vorticity;      /* OBC_ZERO_VORTICITY ... (read by CorAdCalc) */
  int32_t zero_strain, freeslip_strain, computed_strain;      /* OBC_ZERO_STRAIN ... (read by horizontal_viscosity; OBC_SPECIFIED_STRAIN has
                                                                 no effect there: its branch hangs behind the other three, MOM_hor_visc.F90:735-750) */
  int32_t zero_biharmonic;                                    /* OBC_ZERO_BIHARMONIC */
  const mom6hip_obc_segment_t *segment;             /* number_of_segments entries (HOST array) */
  const int32_t *segnum_u, *segnum_v;               /* OBC%segnum_u(IsdB:IedB, jsd:jed), segnum_v(isd:ied, JsdB:JedB): the segment number
                                                       (1-based) of a face, MOM6HIP_OBC_NONE elsewhere; HOST arrays */
  /* OBC%rx_normal(IsdB:IedB, jsd:jed, nk), OBC%ry_normal(isd:ied, JsdB:JedB, nk): the radiation rates the Orlanski segments keep between
   * steps (restart fields), in the memory space of the call; OBC%gamma_uv, OBC%rx_max (OBC_RADIATION_MAX).  Read and written by the RK2
   * step (its calls of radiation_open_bdry_conds); mom6hip_radiation_open_bdry_conds takes them as arguments. */
  double *rx_normal, *ry_normal;
  double gamma_uv, rx_max;
  /* OBC%rx_oblique_u, ry_oblique_u, cff_normal_u (the layout of rx_normal), OBC%rx_oblique_v, ry_oblique_v, cff_normal_v (of ry_normal): what
   * the oblique segments keep between steps (restart fields), in the memory space of the call; read and written by
   * radiation_open_bdry_conds for OBLIQUE segments with gamma_uv < 1; may be NULL otherwise */
  double *rx_oblique_u, *ry_oblique_u, *cff_normal_u, *rx_oblique_v, *ry_oblique_v, *cff_normal_v;
} mom6hip_obc_t;

/* continuity_PPM with OBC associated (:86-194; the OBC branches of PPM_reconstruction_x/y :2385-2432 / :2521-2568, of
 * zonal/merid_flux_layer :956-971 / :1854-1870, of zonal/meridional_mass_flux :629-634, :722-734, :744-748, :759-779, :782-805 and
 * their twins, of zonal/meridional_flux_thickness :1058-1088 / :1960-1990).  obc == NULL: mom6hip_continuity. */
int mom6hip_continuity_obc(mom6hip_ctx_t *ctx, const mom6hip_continuity_cs_t *cs, const mom6hip_obc_t *obc, const double *u,
                           const double *v, const double *hin, double *h, double *uh, double *vh, double dt, const double *uhbt,
                           const double *vhbt, const double *visc_rem_u, const double *visc_rem_v, double *u_cor, double *v_cor,
                           const mom6hip_bt_cont_t *BT_cont, double *du_cor, double *dv_cor, int32_t memspace);

/* ---- MOM_EOS / MOM_PressureForce_FV --------------------------------------------------------- */

/* EQN_OF_STATE forms provided (src/equation_of_state/MOM_EOS.F90:145-173; default "WRIGHT") */
#define MOM6HIP_EOS_LINEAR 1   /* MOM_EOS_linear.F90 */
#define MOM6HIP_EOS_UNESCO 2   /* MOM_EOS_UNESCO.F90 (Jackett & McDougall 1995) */
#define MOM6HIP_EOS_WRIGHT 3   /* MOM_EOS_Wright.F90 (the "WRIGHT" form; its density agrees with WRIGHT_REDUCED to roundoff) */
#define MOM6HIP_EOS_WRIGHT_FULL 4      /* MOM_EOS_Wright_full.F90 */
#define MOM6HIP_EOS_WRIGHT_REDUCED 5   /* MOM_EOS_Wright_red.F90 */

typedef struct mom6hip_eos {
  int32_t form;                /* MOM6HIP_EOS_* */
  int32_t reserved;
  double Rho_T0_S0, dRho_dT, dRho_dS;   /* linear EOS: RHO_T0_S0, DRHO_DT, DRHO_DS */
} mom6hip_eos_t;

/* PressureForce_FV_CS, src/core/MOM_PressureForce_FV.F90 (the members the provided branches read), and what the routine takes
 * from its other arguments (ALE_CSp, GV, tv) to choose a branch */
typedef struct mom6hip_pressureforce_cs {
  double Rho0;                 /* RHO_PGF_REF (default GV%Rho0) */
  double GFS_scale;            /* must be 1.0 */
  double Z_ref;                /* G%Z_ref */
  int32_t reconstruct;         /* RECONSTRUCT_FOR_PRESSURE (default True; .testing/tc4 sets False) */
  int32_t Recon_Scheme;        /* PRESSURE_RECONSTRUCTION_SCHEME (1 = PLM when reconstruct) */
  int32_t boundary_extrap;     /* BOUNDARY_EXTRAPOLATION_PRESSURE (default True) */
  int32_t useMassWghtInterp;   /* MASS_WEIGHT_IN_PRESSURE_GRADIENT (default False) */
  int32_t use_ALE;             /* associated(ALE_CSp): USE_REGRIDDING; `use_ALE = CS%reconstruct .and. use_EOS` (:562) only then */
  int32_t nkmb;                /* GV%nk_rho_varies: the bulk mixed layer's variable-density layers (0 without one; layered mode) */
  double P_Ref;                /* tv%P_Ref, the reference pressure of the coordinate density (read with nkmb > 0) */
  const double *Rlay;          /* GV%Rlay(1:nk), HOST pointer: read with nkmb > 0 or without an equation of state; else NULL */
  const double *g_prime;       /* GV%g_prime(1:nk+1), HOST pointer: Set_pbce_Bouss without an equation of state; else NULL */
} mom6hip_pressureforce_cs_t;

/*
 * PressureForce_FV_Bouss(h, tv, PFu, PFv, G, GV, US, CS, ALE_CSp, p_atm, pbce, eta)
 *                                                            src/core/MOM_PressureForce_FV.F90:462
 * tv%T, tv%S are passed as T, S; p_atm (2-D), pbce (3-D), eta (2-D) may be NULL.  Branches (as the reference chooses them):
 *   eos != NULL, use_ALE and reconstruct   int_density_dz_generic_plm on PLM edge values of T and S (:753-758)
 *   eos != NULL otherwise                  int_density_dz (:765-768) -> the analytic integrals of the LINEAR and WRIGHT
 *                                          forms (MOM_EOS_linear.F90:259, MOM_EOS_Wright.F90:389; EOS_QUADRATURE = False);
 *                                          with nkmb > 0 the layers lighter than the buffer layer take its T and S (:650-670)
 *   eos == NULL (T, S may be NULL)         the layered form with GV%Rlay (:775-789), Set_pbce_Bouss with GV%g_prime
 * Tides, SAL, the Stanley SGS terms, GFS_scale < 1, the PPM reconstruction and the quadrature form of int_density_dz
 * (EOS_QUADRATURE, or an equation of state without analytic integrals outside the PLM branch) are not provided.
 * Metrics needed: bathyT, IdxCu, IdyCv.
 */
int mom6hip_pressureforce_fv_bouss(mom6hip_ctx_t *ctx, const mom6hip_pressureforce_cs_t *cs, const mom6hip_eos_t *eos,
                                   const double *h, const double *T, const double *S, const double *p_atm,
                                   double *PFu, double *PFv, double *pbce, double *eta, int32_t memspace);

/*
 * PressureForce_FV_nonBouss(h, tv, PFu, PFv, G, GV, US, CS, ALE_CSp, p_atm, pbce, eta)   src/core/MOM_PressureForce_FV.F90:89
 * The non-Boussinesq form (BOUSSINESQ = False): h in mass per unit area, H_to_RZ = GV%H_to_RZ (1 for kg m-2), the finite-volume
 * integrals in pressure of the specific-volume anomaly (int_spec_vol_dp_generic_plm, MOM_density_integrals.F90:1479) with the
 * same PLM reconstruction of T and S, Set_pbce_nonBouss (MOM_PressureForce_Montgomery.F90:752) for pbce, eta = the column mass.
 * Same restrictions as the Boussinesq entry (no tides / SAL, GFS_scale = 1, no bulk mixed layer).
 */
int mom6hip_pressureforce_fv_nonbouss(mom6hip_ctx_t *ctx, const mom6hip_pressureforce_cs_t *cs, const mom6hip_eos_t *eos,
                                      const double *h, const double *T, const double *S, const double *p_atm, double H_to_RZ,
                                      double *PFu, double *PFv, double *pbce, double *eta, int32_t memspace);

/* calculate_density(T, S, pressure, rho, EOS, dom, rho_ref) on n points (src/equation_of_state/MOM_EOS.F90:299);
 * use_rho_ref = 0: in-situ density. */
int mom6hip_calculate_density(mom6hip_ctx_t *ctx, const mom6hip_eos_t *eos, const double *T, const double *S,
                              const double *pressure, double *rho, int64_t n, int32_t use_rho_ref, double rho_ref,
                              int32_t memspace);

/* ---- MOM_barotropic ------------------------------------------------------------------------- */

/* BT_THICK_SCHEME (src/core/MOM_barotropic.F90:400-403; default FROM_BT_CONT when USE_BT_CONT_TYPE) */
#define MOM6HIP_BT_HARMONIC     1
#define MOM6HIP_BT_ARITHMETIC   2
#define MOM6HIP_BT_HYBRID       3
#define MOM6HIP_BT_FROM_BT_CONT 4

/*
 * barotropic_CS, src/core/MOM_barotropic.F90:104-332: the run-time parameters of barotropic_init (:4376) that
 * the provided branch reads, and the state the reference keeps in the control structure between calls.
 * The barotropic domain has the halo of the grid (BTHALO = 0, the default: clone_MOM_domain with min_halo 0,
 * :4741), so every 2-D array here has the shape of the grid's h/u/v/q arrays.
 * The switches in `unsupported` must all be 0 (their reference defaults); btstep fails with an error otherwise.
 * The array members are caller-owned, in the memory space of the call that uses them.
 */
typedef struct mom6hip_barotropic_cs {
  double dtbt;                 /* CS%dtbt: the barotropic time step [s] (set_dtbt or DTBT > 0) */
  double dtbt_max;             /* CS%dtbt_max (set_dtbt) */
  double dtbt_fraction;        /* -DTBT when DTBT < 0 (default 0.98) */
  double bebt;                 /* BEBT (0.1) */
  double dt_bt_filter;         /* DT_BT_FILTER (-0.25) */
  double vel_underflow;        /* VEL_UNDERFLOW (0) */
  double G_extra;              /* G_BT_EXTRA (0) */
  double BT_Coriolis_scale;    /* BT_CORIOLIS_SCALE (1) */
  double Z_ref;                /* G%Z_ref, REFERENCE_HEIGHT (0) */
  double maxCFL_BT_cont;       /* MAXCFL_BT_CONT (0.25): read with bound_BT_corr */
  double reserved0[6];
  int32_t Sadourny;            /* SADOURNY (1) */
  int32_t linearized_BT_PV;    /* LINEARIZED_BT_CORIOLIS (1) */
  int32_t strong_drag;         /* BT_STRONG_DRAG (0) */
  int32_t visc_rem_u_uh0;      /* BT_USE_VISC_REM_U_UH0 (0) */
  int32_t adjust_BT_cont;      /* ADJUST_BT_CONT (0) */
  int32_t use_wide_halos;      /* BT_USE_WIDE_HALOS (1) */
  int32_t hvel_scheme;         /* MOM6HIP_BT_* */
  int32_t nstep_last;          /* CS%nstep_last (out) */
  /* INTEGRAL_BT_CONTINUITY, (free), (free: NONLINEAR_BT_CONTINUITY is provided, see below), BOUND_BT_CORRECTION without BT_cont bounds, GRADUAL_BT_ICS,
   * BT_NONLIN_STRESS, DYNAMIC_SURFACE_PRESSURE, BT_LINEAR_WAVE_DRAG, CLIP_BT_VELOCITY, CALCULATE_SAL,
   * BT_USE_OLD_CORIOLIS_BRACKET_BUG, BAROTROPIC_ANSWER_DATE < 20190101 */
  int32_t unsupported[12];
  int32_t bound_BT_corr;       /* BOUND_BT_CORRECTION (0) with BT_CONT_CORR_BOUNDS (its default, True) and a BT_cont argument:
                                * the mass-source correction eta_cor is limited to what the open faces can carry at
                                * MAXCFL_BT_CONT, and to the water in the cell (MOM_barotropic.F90:1587-1615) */
  int32_t BT_project_velocity; /* BT_PROJECT_VELOCITY (0): the velocities are stepped first and projected for the transports
                                * (trans_wt = 1+bebt, -bebt; the pressure force from eta instead of eta_pred: :804-808, :1751, :1870) */
  int32_t Nonlinear_continuity;       /* NONLINEAR_BT_CONTINUITY (0): without a BT_cont argument the open face areas follow the free
                                       * surface (find_face_areas with eta, :4246-4262): in set_dtbt when eta is given (:2871), at the
                                       * start of btstep (:1137-1138) and, with a positive update period, again before every
                                       * Nonlin_cont_update_period-th barotropic step (:1852-1856; the stencil of the wide-halo march is 2
                                       * then, :752-753).  With a BT_cont argument (USE_BT_CONT_TYPE, .testing/tc1, tc2) btstep ignores it. */
  int32_t Nonlin_cont_update_period;  /* NONLIN_BT_CONT_UPDATE_PERIOD (1) */
  double *frhatu, *frhatv;     /* 3-D u / v: layer weights (btcalc) */
  double *eta_cor;             /* 2-D h: mass source over a baroclinic step (bt_mass_source) */
  double *IDatu, *IDatv;       /* 2-D u / v: inverse total depth at velocity points (barotropic_init :5070-5087) */
  double *ubtav, *vbtav;       /* 2-D u / v: time-filtered barotropic velocities (btstep) */
  double *q_D;                 /* 2-D q: f / D with the resting depth (linearized_BT_PV; barotropic_init :4838) */
  double *D_u_Cor, *D_v_Cor;   /* 2-D u / v: resting depths at velocity points */
  void *reserved2[6];
} mom6hip_barotropic_cs_t;

/* The time-invariant arrays of barotropic_init (:4826-4858, :5070-5087): IDatu, IDatv, q_D, D_u_Cor, D_v_Cor
 * (with their halos); zeroes frhatu, frhatv, eta_cor, ubtav, vbtav.  Metrics needed: bathyT, areaT, mask2dT,
 * mask2dCu, mask2dCv, CoriolisBu. */
int mom6hip_barotropic_init(mom6hip_ctx_t *ctx, mom6hip_barotropic_cs_t *cs, int32_t memspace);

/* btcalc(h, G, GV, CS, h_u, h_v, may_use_default, OBC)              src/core/MOM_barotropic.F90:3394
 * h_u / h_v may be NULL; without OBC (with: mom6hip_btcalc_obc). */
int mom6hip_btcalc(mom6hip_ctx_t *ctx, mom6hip_barotropic_cs_t *cs, const double *h, const double *h_u,
                   const double *h_v, int32_t may_use_default, int32_t memspace);
/* btcalc with OBC associated: the weights of the faces of the open-boundary segments are those of the cell inside (:3610-3664).
 * obc == NULL: mom6hip_btcalc. */
int mom6hip_btcalc_obc(mom6hip_ctx_t *ctx, mom6hip_barotropic_cs_t *cs, const double *h, const double *h_u, const double *h_v,
                       int32_t may_use_default, const struct mom6hip_obc *obc, int32_t memspace);

/* bt_mass_source(h, eta, set_cor, G, GV, CS)                         src/core/MOM_barotropic.F90:4318 */
int mom6hip_bt_mass_source(mom6hip_ctx_t *ctx, mom6hip_barotropic_cs_t *cs, const double *h, const double *eta,
                           int32_t set_cor, int32_t memspace);

/* set_dtbt(G, GV, US, CS, eta, pbce, BT_cont, gtot_est, SSH_add)     src/core/MOM_barotropic.F90:2801
 * Sets cs->dtbt_max to the maximum stable step and cs->dtbt = dtbt_fraction * dtbt_max; with more than one tile the
 * minimum over the tiles is taken through the registered min callback (min_across_PEs, :2915).  pbce or gtot_est is used
 * (pbce may be NULL); BT_cont may be NULL. */
int mom6hip_set_dtbt_eta(mom6hip_ctx_t *ctx, mom6hip_barotropic_cs_t *cs, const double *eta, const double *pbce,
                         const mom6hip_bt_cont_t *BT_cont, double gtot_est, double SSH_add, int32_t memspace);
/* the same without the eta argument (eta not present) */
int mom6hip_set_dtbt(mom6hip_ctx_t *ctx, mom6hip_barotropic_cs_t *cs, const double *pbce,
                     const mom6hip_bt_cont_t *BT_cont, double gtot_est, double SSH_add, int32_t memspace);

/*
 * btstep(U_in, V_in, eta_in, dt, bc_accel_u, bc_accel_v, forces, pbce, eta_PF_in, U_Cor, V_Cor, accel_layer_u,
 *        accel_layer_v, eta_out, uhbtav, vhbtav, G, GV, US, CS, visc_rem_u, visc_rem_v, SpV_avg, ADp, OBC, BT_cont,
 *        eta_PF_start, taux_bot, tauy_bot, uh0, vh0, u_uh0, v_vh0, etaav)    src/core/MOM_barotropic.F90:423
 * forces%taux / forces%tauy are passed as taux, tauy [R L Z T-2] together with RZ_to_H = GV%RZ_to_H.
 * Optional / pointer arguments are NULL when absent: BT_cont, eta_PF_start, taux_bot + tauy_bot,
 * uh0 + vh0 + u_uh0 + v_vh0, etaav.  OBC must not be associated (SpV_avg, ADp are not read).
 * eta_out may be the same array as eta_in.
 * On a one-tile domain the barotropic subcycle is captured as a hipGraph (cached in the context, keyed on pointers,
 * ranges and weights) and replayed with one launch; MOM6HIP_BT_GRAPH=0 in the environment disables that.
 * The one transcendental of the routine, bt_rem = av_rem ** (1/nstep) (:1529), is evaluated with a correctly
 * rounded pow; the reference's is the libm pow of its build (DESIGN.md "btstep").
 */
/* How often btstep captured its subcycle as a hipGraph and how often it replayed one (one-tile domains). */
int mom6hip_bt_graph_stats(mom6hip_ctx_t *ctx, int64_t *captures, int64_t *launches);
/* the number of nodes (kernels) of the subcycle graph captured last */
int mom6hip_bt_graph_nodes(mom6hip_ctx_t *ctx, int64_t *nodes);

int mom6hip_btstep(mom6hip_ctx_t *ctx, mom6hip_barotropic_cs_t *cs, const double *U_in, const double *V_in,
                   const double *eta_in, double dt, const double *bc_accel_u, const double *bc_accel_v,
                   const double *taux, const double *tauy, double RZ_to_H, const double *pbce,
                   const double *eta_PF_in, const double *U_Cor, const double *V_Cor, double *accel_layer_u,
                   double *accel_layer_v, double *eta_out, double *uhbtav, double *vhbtav,
                   const double *visc_rem_u, const double *visc_rem_v, const mom6hip_bt_cont_t *BT_cont,
                   const double *eta_PF_start, const double *taux_bot, const double *tauy_bot, const double *uh0,
                   const double *vh0, const double *u_uh0, const double *v_vh0, double *etaav, int32_t memspace);
/* btstep with OBC associated: specified, Flather and gradient segments -- the summed gravity projected across the segments (:1089-1110),
 * set_up_BT_OBC (:3172; the external values segment%normal_vel_bt, %SSH, %normal_trans), the velocities of the segments' faces kept
 * through the time step and set by apply_velocity_OBCs (:2931) with their running sums, e_anom across the segments (:2490-2519), the
 * accelerations of the segments' faces (:2591-2606).  A radiation-only segment has no barotropic velocity in the reference and is
 * refused.  obc == NULL: mom6hip_btstep. */
int mom6hip_btstep_obc(mom6hip_ctx_t *ctx, mom6hip_barotropic_cs_t *cs, const double *U_in, const double *V_in, const double *eta_in,
                       double dt, const double *bc_accel_u, const double *bc_accel_v, const double *taux, const double *tauy,
                       double RZ_to_H, const double *pbce, const double *eta_PF_in, const double *U_Cor, const double *V_Cor,
                       double *accel_layer_u, double *accel_layer_v, double *eta_out, double *uhbtav, double *vhbtav,
                       const double *visc_rem_u, const double *visc_rem_v, const mom6hip_bt_cont_t *BT_cont, const double *eta_PF_start,
                       const double *taux_bot, const double *tauy_bot, const double *uh0, const double *vh0, const double *u_uh0,
                       const double *v_vh0, double *etaav, const struct mom6hip_obc *obc, int32_t memspace);

/* ---- MOM_vert_friction ------------------------------------------------------------------------ */

/*
 * vertvisc_CS, src/parameterizations/vertical/MOM_vert_friction.F90:40-190: the parameters the provided branch reads
 * (defaults of vertvisc_init :2465-2961 in brackets) and the arrays the module keeps between its three entry points.
 * a_u / a_v: (nk+1) interfaces at u / v points [H T-1]; h_u / h_v: nk layers [H].
 * Provided: BOTTOMDRAGLAW or KV_EXTRA_BBL or neither, HARMONIC_VISC, HARMONIC_BL_SCALE, KV_ML_INVZ2 with HMIX_FIXED,
 * visc%Kv_shear, visc%Ray_u/v, DIRECT_STRESS, the CFL-based or MAXVEL velocity truncation, DYNAMIC_VISCOUS_ML and the
 * viscous mixed layer of a bulk mixed layer (nkml > 0).  Not provided (refused by name): FIXED_DEPTH_LOTW_ML,
 * LOTW_VISCOUS_ML_FLOOR, USE_GL90_IN_SSW,
 * STOKES_MIXING_COMBINED / FPMIX, ice shelves, OBC, visc%Kv_shear_Bu, VERT_FRICTION_ANSWER_DATE < 20190101,
 * non-Boussinesq, U_TRUNC_FILE / V_TRUNC_FILE.
 */
typedef struct mom6hip_vertvisc_cs {
  double Hmix;            /* HMIX_FIXED [Z] */
  double Hmix_stress;     /* HMIX_STRESS [H] (DIRECT_STRESS) */
  double Kvml_invZ2;      /* KV_ML_INVZ2 [H Z T-1] (0) */
  double Kv;              /* KV [H Z T-1] */
  double Hbbl;            /* HBBL [Z] */
  double Kv_extra_bbl;    /* KV_EXTRA_BBL [H Z T-1] (0), read if !bottomdraglaw */
  double harm_BL_val;     /* HARMONIC_BL_SCALE (0) */
  double maxvel;          /* MAXVEL [L T-1] (3e8) */
  double CFL_trunc;       /* CFL_TRUNCATE (0.5) */
  double vel_underflow;   /* VEL_UNDERFLOW [L T-1] (0) */
  double H_to_RZ;         /* GV%H_to_RZ */
  double vonKar;          /* VON_KARMAN_CONST (0.41): the surface boundary layer viscosity of dynamic_viscous_ML / nkml > 0 */
  int32_t dynamic_viscous_ML; /* DYNAMIC_VISCOUS_ML (0): find_coupling_coef adds the viscosity of a surface boundary layer whose
                                 fractional number of layers set_viscous_ML left in visc%nkml_visc_u/v (:2047-2252, the branch
                                 without LOTW_VISCOUS_ML_FLOOR); needs visc%ustar */
  int32_t nkml;           /* GV%nkml (0): with a bulk mixed layer and no dynamic_viscous_ML its nkml layers are the boundary layer */
  double reserved0[3];
  int32_t bottomdraglaw;  /* BOTTOMDRAGLAW (1) */
  int32_t harmonic_visc;  /* HARMONIC_VISC (0) */
  int32_t direct_stress;  /* DIRECT_STRESS (0) */
  int32_t CFL_based_trunc;/* CFL_BASED_TRUNCATIONS (1) */
  int32_t answer_date;    /* VERT_FRICTION_ANSWER_DATE (99991231); >= 20190101 */
  int32_t unsupported[7]; /* (free), (free), fixed_LOTW_ML, apply_LOTW_floor, use_GL90_in_SSW, StokesMixing,
                             non_Boussinesq: any nonzero is refused */
  int64_t ntrunc;         /* CS%ntrunc: velocity truncations so far (see mom6hip_vertvisc_ntrunc) */
  double *a_u, *a_v, *h_u, *h_v;
  void *reserved1[4];
} mom6hip_vertvisc_cs_t;

/* The members of vertvisc_type (src/core/MOM_variables.F90:218-283) the provided branch reads. */
typedef struct mom6hip_vertvisc_type {
  const double *Kv_bbl_u, *Kv_bbl_v;        /* 2-D, u / v points [H Z T-1]   (BOTTOMDRAGLAW) */
  const double *bbl_thick_u, *bbl_thick_v;  /* 2-D, u / v points [Z]         (BOTTOMDRAGLAW) */
  const double *Ray_u, *Ray_v;              /* 3-D, u / v points [H T-1], or NULL */
  const double *Kv_shear;                   /* (nk+1) interfaces at h points [H Z T-1], or NULL */
  const double *Kv_shear_Bu;                /* must be NULL */
  const double *nkml_visc_u, *nkml_visc_v;  /* 2-D, u / v points: the fractional number of layers in the viscous surface boundary
                                               layer; WRITTEN by set_viscous_ML, read by vertvisc_coef (DYNAMIC_VISCOUS_ML) */
  const double *ustar;                      /* 2-D, h points [Z T-1]: forces%ustar as find_ustar returns it (MOM_forcing_type.F90
                                               :1236, Boussinesq); read by set_viscous_ML and by vertvisc_coef with
                                               DYNAMIC_VISCOUS_ML or nkml > 0 (it is a member of `forces` in the reference) */
  const void *reserved[1];
} mom6hip_vertvisc_type_t;

/* vertvisc_coef(u, v, h, dz, forces, visc, tv, dt, G, GV, US, CS, OBC, VarMix)       MOM_vert_friction.F90:1168
 * (+ find_coupling_coef :1768).  dz = NULL stands for the Boussinesq thickness_to_dz, dz = GV%H_to_Z*h. */
int mom6hip_vertvisc_coef(mom6hip_ctx_t *ctx, mom6hip_vertvisc_cs_t *cs, const double *u, const double *v, const double *h,
                          const double *dz, const mom6hip_vertvisc_type_t *visc, double dt, int32_t memspace);

/* vertvisc(u, v, h, forces, visc, dt, OBC, ADp, CDp, G, GV, US, CS, taux_bot, tauy_bot, fpmix, Waves)        :526
 * including vertvisc_limit_vel (:2259).  forces%taux / %tauy are passed as taux, tauy; taux_bot / tauy_bot may be NULL. */
int mom6hip_vertvisc(mom6hip_ctx_t *ctx, mom6hip_vertvisc_cs_t *cs, double *u, double *v, const double *h, const double *taux,
                     const double *tauy, const mom6hip_vertvisc_type_t *visc, double dt, double *taux_bot, double *tauy_bot,
                     int32_t memspace);

/* vertvisc_coef and vertvisc with OBC associated.  vertvisc_coef: at the faces of the open-boundary segments the thicknesses, the
 * depth, visc%Kv_shear and ustar of the cell inside the boundary replace the two-cell means (:1335-1355, :1546-1566, :1901-1925,
 * :2061-2110).  vertvisc: the normal_vel of the specified segments is stored over the result (:988-1006).  obc == NULL: the entries
 * above.  (mom6hip_vertvisc_step and mom6hip_vertvisc_and_remnant have no OBC form: call the separate entries.) */
int mom6hip_vertvisc_coef_obc(mom6hip_ctx_t *ctx, mom6hip_vertvisc_cs_t *cs, const double *u, const double *v, const double *h,
                              const double *dz, const mom6hip_vertvisc_type_t *visc, double dt, const struct mom6hip_obc *obc,
                              int32_t memspace);
int mom6hip_vertvisc_obc(mom6hip_ctx_t *ctx, mom6hip_vertvisc_cs_t *cs, double *u, double *v, const double *h, const double *taux,
                         const double *tauy, const mom6hip_vertvisc_type_t *visc, double dt, double *taux_bot, double *tauy_bot,
                         const struct mom6hip_obc *obc, int32_t memspace);

/* vertvisc followed by vertvisc_remnant with the same dt -- the pair the split RK2 step calls at :731-744 and :985-994 --
 * in one pass over the coupling coefficients; the results are those of the two calls. */
int mom6hip_vertvisc_and_remnant(mom6hip_ctx_t *ctx, mom6hip_vertvisc_cs_t *cs, double *u, double *v, const double *h,
                                 const double *taux, const double *tauy, const mom6hip_vertvisc_type_t *visc, double dt,
                                 double *taux_bot, double *tauy_bot, double *visc_rem_u, double *visc_rem_v, int32_t memspace);

/* vertvisc_coef, then (update_velocities /= 0) vertvisc, then vertvisc_remnant with the same dt -- the sequences of the split
 * RK2 step at :598-600 (velocities untouched), :717-744 and :974-994 -- in one kernel per direction (each lane sweeps its
 * column for the coefficients and then solves it); the results are those of the separate calls.  taux / tauy / taux_bot /
 * tauy_bot are only used with update_velocities. */
int mom6hip_vertvisc_step(mom6hip_ctx_t *ctx, mom6hip_vertvisc_cs_t *cs, double *u, double *v, const double *h, const double *dz,
                          const double *taux, const double *tauy, const mom6hip_vertvisc_type_t *visc, double dt,
                          int32_t update_velocities, double *taux_bot, double *tauy_bot, double *visc_rem_u, double *visc_rem_v,
                          int32_t memspace);

/* CS%ntrunc: vertvisc counts truncations on the device; this adds the count since the last call to cs->ntrunc
 * (synchronises the stream).  With MOM6HIP_MEM_HOST vertvisc does it itself. */
int mom6hip_vertvisc_ntrunc(mom6hip_ctx_t *ctx, mom6hip_vertvisc_cs_t *cs);

/* vertvisc_remnant(visc, visc_rem_u, visc_rem_v, dt, G, GV, US, CS)                                            :1064 */
int mom6hip_vertvisc_remnant(mom6hip_ctx_t *ctx, const mom6hip_vertvisc_cs_t *cs, const mom6hip_vertvisc_type_t *visc,
                             double *visc_rem_u, double *visc_rem_v, double dt, int32_t memspace);

/* ---- MOM_set_viscosity ------------------------------------------------------------------------ */

/*
 * set_visc_CS, src/parameterizations/vertical/MOM_set_viscosity.F90:48-130, as set by set_visc_init (:2886-3190): the
 * parameters set_viscous_BBL's provided branch reads (defaults in brackets).
 * Provided: BOTTOMDRAGLAW with the quadratic or LINEAR_DRAG law, DRAG_BG_VEL, BBL_USE_EOS (Wright / linear equation of
 * state) or the layer target densities GV%Rlay, BBL_THICK_MIN, KV_BBL_MIN, CORRECT_BBL_BOUNDS, DRAG_AS_BODY_FORCE, the
 * kappa-shear cap of the layer thickness (RiNo_mix).  Not provided (refused by name): CHANNEL_DRAG, BBL_USE_TIDAL_BG,
 * a bulk mixed layer (nkml > 0), non-Boussinesq mode (tv%SpV_avg), tv%p_surf, OBC, porous barriers.
 * set_viscous_ML (:1898) does nothing unless DYNAMIC_VISCOUS_ML or an ice shelf is present (:2043-2044); DYNAMIC_VISCOUS_ML
 * is provided (.testing/tc1, tc2), ice shelves are not.
 */
typedef struct mom6hip_set_visc_cs {
  double cdrag;            /* CDRAG (0.003) */
  double drag_bg_vel;      /* DRAG_BG_VEL [L T-1] (0) */
  double Hbbl;             /* HBBL in thickness units [H] (:3127) */
  double dz_bbl;           /* HBBL [Z] */
  double BBL_thick_min;    /* BBL_THICK_MIN [Z] (0) */
  double Kv_BBL_min;       /* KV_BBL_MIN [H Z T-1] (KV) */
  double BBL_thick_max;    /* G%Rad_Earth_L*US%L_to_Z [Z] (6.378e6) */
  double H_to_RZ;          /* GV%H_to_RZ */
  double omega;            /* OMEGA (7.2921e-5) [T-1] */
  double omega_frac;       /* ML_OMEGA_FRAC (0) */
  double ustar_min;        /* 2e-4*omega*(GV%Angstrom_H + GV%H_subroundoff) (:2998) */
  double TKE_decay;        /* TKE_DECAY_VISC (= TKE_DECAY, 0) */
  double bulk_Ri_ML;       /* BULK_RI_ML_VISC (= BULK_RI_ML, 0) */
  double c_Smag;           /* SMAG_CONST_CHANNEL (SMAG_LAP_CONST or 0.15): CHANNEL_DRAG */
  double Chan_drag_max_vol;/* CHANNEL_DRAG_MAX_BBL_THICK [Z] (-1: no fixed limit) */
  double Z_ref;            /* G%Z_ref [Z] (0): the bottom depths at velocity points of CHANNEL_DRAG (:365-370) */
  int32_t bottomdraglaw;   /* BOTTOMDRAGLAW (1): without it set_viscous_BBL returns at once (:321) */
  int32_t linear_drag;     /* LINEAR_DRAG (0) */
  int32_t BBL_use_EOS;     /* BBL_USE_EOS (= USE_EOS) */
  int32_t correct_BBL_bounds; /* CORRECT_BBL_BOUNDS (0) */
  int32_t body_force_drag; /* DRAG_AS_BODY_FORCE (0): needs visc%Ray_u / %Ray_v */
  int32_t RiNo_mix;        /* kappa_shear_is_used (0) */
  int32_t initialized;
  int32_t unsupported[9];  /* (free), BBL_use_tidal_bg, (free), (free), non_Boussinesq, p_surf, OBC, pbv,
                              ice_shelf: any nonzero is refused */
  const double *Rlay;      /* GV%Rlay(1:nk) [R] (HOST array), read when BBL_use_EOS = 0 (and by set_viscous_ML without an EOS) */
  int32_t dynamic_viscous_ML;       /* DYNAMIC_VISCOUS_ML (0): set_viscous_ML finds the viscous mixed layer (:2111-2230, :2400-2506) */
  int32_t nkml;                     /* GV%nkml (0; 2 with the bulk mixed layer): the layers that are always in the mixed layer */
  int32_t Channel_drag;             /* CHANNEL_DRAG (0): Rayleigh drag on the layers that touch the sloping bottom, visc%Ray_u / %Ray_v,
                                       and the fraction of the bottom drag left to the viscous boundary layer (:863-1002) */
  int32_t concave_trigonometric_L;  /* TRIG_CHANNEL_DRAG_WIDTHS (1) */
  void *reserved1[1];
} mom6hip_set_visc_cs_t;

/* set_viscous_BBL(u, v, h, tv, visc, G, GV, US, CS, pbv)                                   MOM_set_viscosity.F90:134
 * tv%T, tv%S and tv%eqn_of_state are passed as T, S, eos (may be NULL when BBL_use_EOS = 0).  Sets visc%bbl_thick_u/v,
 * visc%Kv_bbl_u/v on the ocean faces of the compute domain (I = IscB..IecB, j = jsc..jec; i = isc..iec, J = JscB..JecB)
 * and, when they are present, zeroes visc%Ray_u/v (:416-417) and adds the body-force drag.  The members of `visc` this
 * call writes are declared const in mom6hip_vertvisc_type_t because vertvisc only reads them. */
int mom6hip_set_viscous_bbl(mom6hip_ctx_t *ctx, const mom6hip_set_visc_cs_t *cs, const double *u, const double *v,
                            const double *h, const double *T, const double *S, const mom6hip_eos_t *eos,
                            const mom6hip_vertvisc_type_t *visc, int32_t memspace);
/* set_viscous_BBL with CS%OBC associated (set_visc_init :2903): one-sided depths and masks of the faces at and beside the segments
 * (:374-413), the zero-gradient projection of the thicknesses, T and S across the segments' faces (:502-580), the weights of
 * set_v_at_u / set_u_at_v (:1829-1838, :1874-1883).  obc == NULL: mom6hip_set_viscous_bbl. */
int mom6hip_set_viscous_bbl_obc(mom6hip_ctx_t *ctx, const mom6hip_set_visc_cs_t *cs, const double *u, const double *v, const double *h,
                                const double *T, const double *S, const mom6hip_eos_t *eos, const mom6hip_vertvisc_type_t *visc,
                                const struct mom6hip_obc *obc, int32_t memspace);

/* set_viscous_ML(u, v, h, tv, forces, visc, dt, G, GV, US, CS)                                                :1898
 * Returns at once unless DYNAMIC_VISCOUS_ML (ice shelves are not provided) (:2043-2044).  With it: the bulk-Richardson-number
 * search down each velocity column for the base of the viscous surface boundary layer (:2111-2230 at u points, :2400-2506 at
 * v points), written to visc->nkml_visc_u / nkml_visc_v.  tv%T, tv%S, tv%eqn_of_state as T, S, eos (eos NULL: GV%Rlay from
 * cs->Rlay); forces%taux, %tauy as taux, tauy; forces%ustar as visc->ustar (tv%p_surf is not provided).  The reference's
 * exp(-htot*Idecay_len_TKE) (:2178) is evaluated correctly rounded (as is btstep's power, see mom6hip_btstep).
 * Metrics needed: mask2dCu, mask2dCv, CoriolisBu. */
int mom6hip_set_viscous_ml(mom6hip_ctx_t *ctx, const mom6hip_set_visc_cs_t *cs, const double *u, const double *v,
                           const double *h, const double *T, const double *S, const mom6hip_eos_t *eos, const double *taux,
                           const double *tauy, const mom6hip_vertvisc_type_t *visc, double dt, int32_t memspace);

/* ---- MOM_tracer_hor_diff ------------------------------------------------------------------------ */

/*
 * tracer_hor_diff_CS, src/tracer/MOM_tracer_hor_diff.F90:40-100, as set by tracer_hor_diff_init (:1625-1770).
 * Provided: the along-layer diffusion of tracer_hordiff (:119-680) -- KHTR, MAX_TR_DIFFUSION_CFL, CHECK_DIFFUSIVE_CFL (the
 * iteration count from the largest diffusive CFL number, max_across_PEs), conc_underflow; and, with VarMix%use_variable_mixing
 * (round 3, :236-281; .testing/tc1, tc2), the diffusivity of every face KHTR + KHTR_SLOPE_CFF * VarMix%L2u * VarMix%SN_u (the
 * Visbeck form) + MEKE%KhTr_fac * sqrt(MEKE%Kh(i) * MEKE%Kh(i+1)), limited by KHTR_MAX, times the mean of VarMix%Res_fn_h
 * (RESOLN_SCALED_KHTR), floored by KHTR_MIN, and the passivity factor max(KHTR_PASSIVITY_MIN, KHTR_PASSIVITY_COEFF * Rd/dx): the
 * fields come in mom6hip_hordiff_fields_t through mom6hip_tracer_hordiff_varmix.
 * USE_NEUTRAL_DIFFUSION (unsupported[0]) is taken by mom6hip_tracer_hordiff_neutral with its own control structure, and refused by
 * the two entry points that have none.
 * DIFFUSE_ML_TO_INTERIOR (unsupported[2]) is taken by mom6hip_tracer_hordiff_epipycnal with its own control structure (round 4),
 * and refused by the entry points that have none.
 * Not provided (refused by name, any nonzero `unsupported`): USE_HORIZONTAL_BOUNDARY_DIFFUSION,
 * KHTR_USE_EBT_STRUCT, offline khdt arrays, the df_x / df_y flux diagnostics.
 */
typedef struct mom6hip_tracer_hor_diff_cs {
  double KhTr;             /* KHTR [L2 T-1] (0: tracer_hordiff returns at once unless use_variable_mixing) */
  double max_diff_CFL;     /* MAX_TR_DIFFUSION_CFL (-1: no limit) */
  double KhTr_Slope_Cff;   /* KHTR_SLOPE_CFF (0): > 0 with use_variable_mixing needs L2u, L2v, SN_u, SN_v */
  double KhTr_fac;         /* MEKE%KhTr_fac (MEKE_KHTR_FAC): read with MEKE_Kh */
  double KhTr_min;         /* KHTR_MIN [L2 T-1] (0) */
  double KhTr_max;         /* KHTR_MAX [L2 T-1] (0: none) */
  double KhTr_passivity_coeff;  /* KHTR_PASSIVITY_COEFF (0): > 0 needs Rd_dx_h */
  double KhTr_passivity_min;    /* KHTR_PASSIVITY_MIN (0.5) */
  int32_t check_diffusive_CFL;  /* CHECK_DIFFUSIVE_CFL (0) */
  int32_t initialized;
  int32_t unsupported[8];  /* use_neutral_diffusion, use_hor_bnd_diffusion, Diffuse_ML_interior, (free), (free),
                              KhTr_use_ebt_struct, offline (do_online = false), flux diagnostics */
  int32_t use_variable_mixing;  /* VarMix%use_variable_mixing (0): the diffusivities of :236-281 */
  int32_t Resoln_scaled_KhTr;   /* VarMix%Resoln_scaled_KhTr (0): needs Res_fn_h */
  int32_t reserved1[4];
} mom6hip_tracer_hor_diff_cs_t;

/* the fields of MEKE and VarMix tracer_hordiff reads with use_variable_mixing, in the memory space of the call (NULL: not
 * allocated / not in use); h-point fields need a valid halo of 1 */
typedef struct mom6hip_hordiff_fields {
  const double *MEKE_Kh;            /* MEKE%Kh, h points 2-D */
  const double *L2u, *L2v, *SN_u, *SN_v;      /* VarMix%L2u ... (KHTR_SLOPE_CFF > 0), u / v points 2-D */
  const double *Res_fn_h;           /* VarMix%Res_fn_h (RESOLN_SCALED_KHTR), h points 2-D */
  const double *Rd_dx_h;            /* VarMix%Rd_dx_h (KHTR_PASSIVITY_COEFF > 0), h points 2-D */
  const double *h_ML;               /* visc%h_ML (NDIFF_INTERIOR_ONLY: the boundary-layer depth neutral diffusion stays below), h points 2-D */
  void *reserved[4];
} mom6hip_hordiff_fields_t;

typedef struct mom6hip_hordiff_stats {
  int32_t num_itts;        /* iterations of the diffusion (:424-434) */
  int32_t halo_updates;    /* do_group_pass(CS%pass_t) calls (:542) */
  double max_CFL;          /* the largest diffusive CFL number (CHECK_DIFFUSIVE_CFL; else 0) */
} mom6hip_hordiff_stats_t;

/* tracer_hordiff(h, dt, MEKE, VarMix, visc, G, GV, US, CS, Reg, tv, do_online_flag, read_khdt_x, read_khdt_y)       :119
 * tr: ntr tracer arrays (Reg%Tr(m)%t), conc_underflow: ntr values or NULL.  The tracers are updated in place on the compute
 * domain; their halos are refreshed by the group pass before every iteration (left as the last pass made them). */
int mom6hip_tracer_hordiff(mom6hip_ctx_t *ctx, const mom6hip_tracer_hor_diff_cs_t *cs, const double *h, double dt,
                           double *const *tr, const double *conc_underflow, int32_t ntr, int32_t memspace,
                           mom6hip_hordiff_stats_t *stats);
/* the same with the MEKE and VarMix arguments of the reference (fields may be NULL: the call above) */
int mom6hip_tracer_hordiff_varmix(mom6hip_ctx_t *ctx, const mom6hip_tracer_hor_diff_cs_t *cs, const mom6hip_hordiff_fields_t *fields,
                                  const double *h, double dt, double *const *tr, const double *conc_underflow, int32_t ntr,
                                  int32_t memspace, mom6hip_hordiff_stats_t *stats);

/*
 * neutral_diffusion_CS, src/tracer/MOM_neutral_diffusion.F90:38-120, as set by neutral_diffusion_init (:138-330).
 * Provided: NDIFF_CONTINUOUS = True (the default): interface values of T and S by the PPM edge formula (interface_scalar :1078),
 * their density derivatives at the interface pressure (or NDIFF_REF_PRES), find_neutral_surface_positions_continuous (:1353),
 * neutral_surface_flux (:2297) and the update of every tracer (:605-1019), for both values of NDIFF_ANSWER_DATE.
 * NDIFF_INTERIOR_ONLY (interior_only): the surfaces are kept below the surface boundary layer visc%h_ML (fields->h_ML) -- boundary_k_range
 * (src/tracer/MOM_hor_bnd_diffusion.F90:609) for every column and the limits of the walk (:1508-1521).
 * Not provided (refused by name, any nonzero `unsupported`): NDIFF_CONTINUOUS = False (the discontinuous reconstructions),
 * NDIFF_TAPERING, KHTR_USE_EBT_STRUCT, NDIFF_USE_UNMASKED_TRANSPORT_BUG, the flux / tendency diagnostics.
 */
typedef struct mom6hip_neutral_diffusion_cs {
  double ref_pres;             /* NDIFF_REF_PRES [R L2 T-2] (-1, the default: the pressure of the interface) */
  double H_to_RZ;              /* GV%H_to_RZ (with GV%g_Earth of the grid: interface pressures, and hEff back to H) */
  double reserved0[4];
  int32_t ndiff_answer_date;   /* NDIFF_ANSWER_DATE (20240101): > 20240330 sums the four faces' tendencies symmetrically */
  int32_t recalc_neutral_surf; /* tracer_hor_diff_CS%recalc_neutral_surf, RECALC_NEUTRAL_SURF (0) */
  int32_t initialized;
  int32_t interior_only;       /* NDIFF_INTERIOR_ONLY (0): needs fields->h_ML */
  int32_t unsupported[8];      /* .not.continuous_reconstruction, (free), tapering, KhTh_use_ebt_struct,
                                  use_unmasked_transport_bug, diagnostics, (free), (free) */
} mom6hip_neutral_diffusion_cs_t;

/* tracer_hordiff with cs->unsupported[0] (CS%use_neutral_diffusion) set: the branch :474-534.  tr[idx_T] and tr[idx_S] are tv%T
 * and tv%S (in the reference they are registered tracers, Reg%Tr(:)%t => tv%T, and are diffused with the others); eos is
 * tv%eqn_of_state, p_surf tv%p_surf (NULL: not associated).  nd may be NULL when cs->unsupported[0] is 0: the call above.
 * The reference's `stop` in ppm_ave (a neutral layer spanning more than one cell, :1186-1189) is returned as an error. */
int mom6hip_tracer_hordiff_neutral(mom6hip_ctx_t *ctx, const mom6hip_tracer_hor_diff_cs_t *cs,
                                   const mom6hip_neutral_diffusion_cs_t *nd, const mom6hip_hordiff_fields_t *fields,
                                   const double *h, const mom6hip_eos_t *eos, const double *p_surf, double dt, double *const *tr,
                                   const double *conc_underflow, int32_t ntr, int32_t idx_T, int32_t idx_S, int32_t memspace,
                                   mom6hip_hordiff_stats_t *stats);

/*
 * DIFFUSE_ML_TO_INTERIOR (cs->unsupported[2], CS%Diffuse_ML_interior; .testing/tc1): tracer_hordiff scales the along-layer
 * diffusion of the mixed layers (k <= GV%nkml) by ML_KHTR_SCALE, leaves the buffer layers out (:544-550), and then calls
 * tracer_epipycnal_ML_diff (:700-1621) -- the coordinate density of the nk_rho_varies variable-density layers at tv%P_Ref, the
 * density-sorted columns, the pairings of every face with their thicknesses, and for every tracer the limited fluxes between the
 * paired layers -- with both forms of HOR_DIFF_ANSWER_DATE and HOR_DIFF_LIMIT_BUG.  The df2d_x / df2d_y diagnostics are not provided.
 */
typedef struct mom6hip_epipycnal_cs {
  double ML_KhTr_scale;    /* ML_KHTR_SCALE (1.0) */
  double P_Ref;            /* tv%P_Ref [R L2 T-2] */
  double reserved0[4];
  const double *Rlay;      /* GV%Rlay(1:nk) [R] (HOST array) */
  int32_t nkml;            /* GV%nkml */
  int32_t nk_rho_varies;   /* GV%nk_rho_varies (nkmb) */
  int32_t answer_date;     /* HOR_DIFF_ANSWER_DATE (20240101): > 20240330 keeps the four faces' fluxes apart and sums them symmetrically */
  int32_t limit_bug;       /* HOR_DIFF_LIMIT_BUG (1; read with answer_date <= 20240330) */
  int32_t reserved1[4];
} mom6hip_epipycnal_cs_t;

/* tracer_hordiff with cs->unsupported[2] (CS%Diffuse_ML_interior) set.  tr[idx_T] and tr[idx_S] are tv%T and tv%S (registered
 * tracers, diffused with the others; the coordinate density is formed from them after the along-layer diffusion), eos is
 * tv%eqn_of_state.  The halo must be at least 2 points wide.  epi may be NULL when cs->unsupported[2] is 0: mom6hip_tracer_hordiff_varmix.
 * USE_NEUTRAL_DIFFUSION with DIFFUSE_ML_TO_INTERIOR is refused as the reference refuses it (:1732). */
int mom6hip_tracer_hordiff_epipycnal(mom6hip_ctx_t *ctx, const mom6hip_tracer_hor_diff_cs_t *cs, const mom6hip_epipycnal_cs_t *epi,
                                     const mom6hip_hordiff_fields_t *fields, const double *h, const mom6hip_eos_t *eos, double dt,
                                     double *const *tr, const double *conc_underflow, int32_t ntr, int32_t idx_T, int32_t idx_S,
                                     int32_t memspace, mom6hip_hordiff_stats_t *stats);

/* ---- MOM_hor_visc ----------------------------------------------------------------------------- */

/*
 * hor_visc_CS, src/parameterizations/lateral/MOM_hor_visc.F90:40-243, as set by hor_visc_init (:1984-2876): the
 * parameters the provided branch reads (defaults in brackets) and the static 2-D arrays hor_visc_init computes.
 * Provided: LAPLACIAN (KH, KH_VEL_SCALE, KH_BG_MIN, SMAGORINSKY_KH + SMAG_LAP_CONST, ADD_LES_VISCOSITY, BOUND_KH,
 * BETTER_BOUND_KH), BIHARMONIC (AH, AH_VEL_SCALE, AH_TIME_SCALE, SMAGORINSKY_AH + SMAG_BI_CONST, BOUND_CORIOLIS_BIHARM,
 * BOUND_AH, BETTER_BOUND_AH), NOSLIP, USE_LAND_MASK_FOR_HVISC, HORVISC_BOUND_COEF, USE_CONT_THICKNESS.
 * Not provided (refused by name): LEITH_KH / LEITH_AH / USE_LEITHY and their options, USE_MEKE, USE_GME,
 * ANISOTROPIC_VISCOSITY, RE_AH, KH_SIN_LAT, USE_KH_BG_2D, ZB2020, resolution-scaled viscosities (VarMix), OBC, the
 * FrictWork diagnostics, HOR_VISC_ANSWER_DATE < 20190101.
 * The arrays are caller-owned (h-point arrays *_xx, q-point arrays *_xy) and filled by mom6hip_hor_visc_init.
 */
typedef struct mom6hip_hor_visc_cs {
  double Kh;               /* KH [L2 T-1] (0) */
  double Kh_bg_min;        /* KH_BG_MIN (0) */
  double Kh_vel_scale;     /* KH_VEL_SCALE [L T-1] (0) */
  double Smag_Lap_const;   /* SMAG_LAP_CONST (0) */
  double Ah;               /* AH [L4 T-1] (0) */
  double Ah_vel_scale;     /* AH_VEL_SCALE [L T-1] (0) */
  double Ah_time_scale;    /* AH_TIME_SCALE [T] (0) */
  double Smag_bi_const;    /* SMAG_BI_CONST (0) */
  double bound_Cor_vel;    /* BOUND_CORIOLIS_VEL [L T-1] (MAXVEL) */
  double bound_coef;       /* HORVISC_BOUND_COEF (0.8) */
  double reserved0[6];
  int32_t Laplacian;       /* LAPLACIAN (0) */
  int32_t biharmonic;      /* BIHARMONIC (1) */
  int32_t Smagorinsky_Kh;  /* SMAGORINSKY_KH (0) */
  int32_t Smagorinsky_Ah;  /* SMAGORINSKY_AH (0) */
  int32_t bound_Kh;        /* BOUND_KH (1) */
  int32_t better_bound_Kh; /* BETTER_BOUND_KH (= BOUND_KH) */
  int32_t bound_Ah;        /* BOUND_AH (1) */
  int32_t better_bound_Ah; /* BETTER_BOUND_AH (= BOUND_AH) */
  int32_t bound_Coriolis;  /* BOUND_CORIOLIS_BIHARM (= BOUND_CORIOLIS) */
  int32_t add_LES_viscosity; /* ADD_LES_VISCOSITY (0) */
  int32_t no_slip;         /* NOSLIP (0) */
  int32_t use_land_mask;   /* USE_LAND_MASK_FOR_HVISC (1) */
  int32_t use_cont_thick;  /* USE_CONT_THICKNESS (0) */
  int32_t initialized;     /* set by mom6hip_hor_visc_init */
  int32_t unsupported[10]; /* Leith_Kh, Leith_Ah, use_Leithy, MEKE backscatter / RES_SCALE_MEKE_VISC, use_GME, anisotropic, Re_Ah,
                              Kh_sin_lat, use_Kh_bg_2d, use_ZB2020: any nonzero is refused */
  /* h points */
  double *Kh_bg_xx, *Kh_Max_xx, *Ah_bg_xx, *Ah_Max_xx, *Laplac2_const_xx, *Biharm_const_xx, *Biharm_const2_xx, *reduction_xx;
  /* q points */
  double *Kh_bg_xy, *Kh_Max_xy, *Ah_bg_xy, *Ah_Max_xy, *Laplac2_const_xy, *Biharm_const_xy, *Biharm_const2_xy, *reduction_xy;
  /* The MEKE argument of horizontal_viscosity (MOM_MEKE_types.F90), 2-D h-point arrays in the memory space of the call, NULL = not
   * allocated: MEKE%Ku is added to the Laplacian viscosity (:1141-1151 at h points, :1537-1541 at q points: it needs valid halos) and
   * MEKE%Au to the biharmonic one (:1318-1323, :1634-1639); MEKE%mom_src receives the vertical sum of the frictional work
   * (:1783-1800, :1833-1889 with MEKE%backscatter_Ro_c = 0), layers summed in order. */
  const double *MEKE_Ku, *MEKE_Au;
  double *MEKE_mom_src;
  void *reserved1[1];
} mom6hip_hor_visc_cs_t;

/* The computational part of hor_visc_init (:2440-2760): the static arrays of the control structure from the grid metrics
 * and dt (the baroclinic time step, for the stability bounds). */
int mom6hip_hor_visc_init(mom6hip_ctx_t *ctx, mom6hip_hor_visc_cs_t *cs, double dt, int32_t memspace);

/* horizontal_viscosity(u, v, h, diffu, diffv, MEKE, VarMix, G, GV, US, CS, tv, dt, OBC, BT, TD, ADp, hu_cont, hv_cont,
 *                      STOCH)                                     src/parameterizations/lateral/MOM_hor_visc.F90:245
 * MEKE: the members of the control structure above; VarMix, OBC, BT, TD, ADp, STOCH belong to branches that are not provided.  hu_cont / hv_cont may be NULL (they
 * are read only with USE_CONT_THICKNESS).  diffu is written for I = IscB..IecB, j = jsc..jec, diffv for i = isc..iec,
 * J = JscB..JecB; u, v need valid halos of width 2, h of width 2 (hor_visc_vel_stencil :2879). */
int mom6hip_horizontal_viscosity(mom6hip_ctx_t *ctx, const mom6hip_hor_visc_cs_t *cs, const double *u, const double *v,
                                 const double *h, double *diffu, double *diffv, double dt, const double *hu_cont,
                                 const double *hv_cont, int32_t memspace);
/* horizontal_viscosity with OBC associated (and OBC%OBC_pe): OBC_ZERO_STRAIN / OBC_FREESLIP_STRAIN at the corner points of the segments
 * (:733-790, :1388-1409), the thicknesses at and beside their faces (:791-849), OBC_ZERO_BIHARMONIC (:889-903), no viscous acceleration
 * of the segments' own faces (:1751-1782).  OBC_COMPUTED_STRAIN reads segment%tangential_vel (in the memory space of the call).  obc == NULL: mom6hip_horizontal_viscosity. */
int mom6hip_horizontal_viscosity_obc(mom6hip_ctx_t *ctx, const mom6hip_hor_visc_cs_t *cs, const double *u, const double *v,
                                     const double *h, double *diffu, double *diffv, double dt, const double *hu_cont,
                                     const double *hv_cont, const struct mom6hip_obc *obc, int32_t memspace);

/* ---- MOM_dynamics_split_RK2 ------------------------------------------------------------------ */

/*
 * The parameterisation calls inside the split RK2 step that this library does not provide:
 * set_viscous_ML + vertvisc_coef + vertvisc_remnant (src/core/MOM_dynamics_split_RK2.F90:592-600),
 * vertvisc_coef + vertvisc + vertvisc_remnant (:717-744 with dt_pred, :974-994 with dt) and
 * horizontal_viscosity (:860).  A host that has them registers them here; they are called at the reference's
 * seams with DEVICE pointers after the library has synchronised its stream.  With no hooks the step is the
 * inviscid one: visc_rem_u = visc_rem_v = 1, diffu = diffv = 0, velocities untouched by vertvisc.
 */
typedef struct mom6hip_visc_hooks {
  void *user;
  int (*visc_remnant_pred)(void *user, const double *up, const double *vp, const double *h, double dt,
                           double *visc_rem_u, double *visc_rem_v);
  int (*vertvisc)(void *user, double *u, double *v, const double *h, double dt, double *visc_rem_u, double *visc_rem_v);
  int (*horizontal_viscosity)(void *user, const double *u_av, const double *v_av, const double *h_av, double *diffu,
                              double *diffv);
} mom6hip_visc_hooks_t;

/* ---- MOM_thickness_diffuse (src/parameterizations/lateral/MOM_thickness_diffuse.F90) ---------------------------------
 * thickness_diffuse_CS (:40-128) as far as the provided branch reads it, and the fields the reference takes from MEKE / VarMix as
 * plain arrays.  Provided: isopycnal height diffusion (Gent-McWilliams) with the diffusivity KHTH, bounded by
 * KHTH_MIN / KHTH_MAX / KHTH_MAX_CFL, plus MEKE%KhTh_fac*sqrt(MEKE%Kh(i)*MEKE%Kh(i+1)), plus the Visbeck term
 * KHTH_SLOPE_CFF*VarMix%L2u*VarMix%SN_u, times VarMix%Res_fn_u (:182-262); thickness_diffuse_full (:634-1670) with an equation of
 * state (slopes from the density gradients, vert_fill_TS, the slope-limited "safe" streamfunction) or without one (layer
 * densities), with stored slopes (VarMix%slope_x/y) or its own, Boussinesq; the work done against the stratification into
 * MEKE%GM_src (:1196-1209, :1552-1622); the transports into uhtr / vhtr and the thickness tendency (:607-620).
 * Refused: KHTH_USE_FGNV_STREAMFUNCTION, DETANGLE_INTERFACES, KH_ETA_CONST / KH_ETA_VEL_SCALE, USE_STANLEY_GM, MEKE_GEOMETRIC,
 * MEKE_GM_SRC_ALT, READ_KHTH, KHTH_USE_EBT_STRUCT, the QG Leith GM coefficient, DEPTH_SCALED_KHTH, USE_KH_IN_MEKE, SKEB,
 * KHTH_MAX_CFL <= 0, non-Boussinesq, tv%p_surf. */
typedef struct mom6hip_thickness_diffuse_cs {
  double Khth;             /* KHTH [L2 T-1] (0) */
  double Khth_Min;         /* KHTH_MIN (0) */
  double Khth_Max;         /* KHTH_MAX (0: no maximum) */
  double max_Khth_CFL;     /* KHTH_MAX_CFL (0.8) */
  double slope_max;        /* KHTH_SLOPE_MAX [Z L-1] (0.01) */
  double kappa_smooth;     /* KD_SMOOTH [H Z T-1] (1e-6) */
  double KHTH_Slope_Cff;   /* KHTH_SLOPE_CFF (0): the Visbeck term needs L2u, L2v, SN_u, SN_v */
  double KhTh_fac;         /* MEKE%KhTh_fac (1): MEKE_KHTH_FAC, with MEKE_Kh */
  double FGNV_scale;       /* FGNV_FILTER_SCALE (1): the coefficient of the vertical smoothing term of the streamfunction equation (:58) */
  double N2_floor;         /* (FGNV_STRAT_FLOOR * OMEGA)**2 (:2337): the floor of N2 in the elliptic solve */
  double reserved0[2];
  int32_t thickness_diffuse;   /* THICKNESSDIFFUSE (0): without it (or with nothing to diffuse with, :192-194) the call returns at once */
  int32_t use_GM_work_bug;     /* USE_GM_WORK_BUG (0) */
  int32_t nkml;                /* GV%nkml (0): the streamfunction goes linearly to zero over max(nkml, 1) layers */
  int32_t initialized;
  int32_t use_variable_mixing; /* VarMix%use_variable_mixing (0): with it the call works even with KHTH = 0 (:192-194), and the
                                  Visbeck term is added when KHTH_SLOPE_CFF > 0 and L2u ... SN_v are given (:205-207, :242-249) */
  int32_t use_FGNV_streamfn;   /* KHTH_USE_FGNV_STREAMFUNCTION (0): the streamfunction of Ferrari et al. (2010) (:1105-1124, streamfn_solver :1673);
                                  needs cg1 (and g_prime without an equation of state) */
  int32_t unsupported[10];     /* (unused), detangle, Kh_eta, Stanley, MEKE_GEOMETRIC, GM_src_alt, read_khth, ebt_struct / QG Leith,
                                  Use_KH_in_MEKE, non-Boussinesq / p_surf / SKEB: any nonzero is refused */
  /* fields of MEKE and VarMix, in the memory space of the call; NULL = not allocated / not in use */
  const double *MEKE_Kh;       /* MEKE%Kh, h points 2-D (valid halo of 1) */
  const double *L2u, *L2v, *SN_u, *SN_v;      /* VarMix%L2u ... (use_Visbeck), u / v points 2-D */
  const double *Res_fn_u, *Res_fn_v;          /* VarMix%Res_fn_u / _v (RESOLN_SCALED_KHTH), 2-D */
  const double *slope_x, *slope_y;            /* VarMix%slope_x / _y (USE_STORED_SLOPES), u / v points, nk+1 interfaces */
  double *MEKE_GM_src;         /* MEKE%GM_src, h points 2-D: set to the work of this call (:197-199, :1560) */
  const double *Rlay;          /* GV%Rlay(1:nk) (HOST array): the work without an equation of state (:827, :1218) */
  const double *cg1;           /* VarMix%cg1, h points 2-D (valid halo of 1): the first baroclinic gravity wave speed, with use_FGNV_streamfn */
  const double *g_prime;       /* GV%g_prime(1:nk+1) (HOST array): dzN2 without an equation of state (:1095), with use_FGNV_streamfn */
  const double *Depth_fn_u, *Depth_fn_v;      /* VarMix%Depth_fn_u / _v (DEPTH_SCALED_KHTH, :284-289), u / v points 2-D */
} mom6hip_thickness_diffuse_cs_t;

/* thickness_diffuse(h, uhtr, vhtr, tv, dt, G, GV, US, MEKE, VarMix, CDp, CS, STOCH)                               :133
 * tv%T, tv%S, tv%eqn_of_state are passed as T, S, eos (all NULL without an equation of state).  h, uhtr, vhtr are updated on the
 * compute domain (h needs a valid halo of 1, T and S too); uhGM / vhGM (CDp%uhGM / vhGM) receive the transports when not NULL. */
int mom6hip_thickness_diffuse(mom6hip_ctx_t *ctx, const mom6hip_thickness_diffuse_cs_t *cs, double *h, double *uhtr, double *vhtr,
                              const double *T, const double *S, const mom6hip_eos_t *eos, double dt, double *uhGM, double *vhGM,
                              int32_t memspace);

/* ---- MOM_mixed_layer_restrat (src/parameterizations/lateral/MOM_mixed_layer_restrat.F90) ------------------------------
 * mixedlayer_restrat_CS (:40-126) as far as the provided branches read it.  Provided: mixedlayer_restrat (:135)
 * with its two Fox-Kemper et al. (2008) forms -- mixedlayer_restrat_OM4 (:175; general coordinates: the mixed layer depth from a
 * density difference MLE_DENSITY_DIFF or from the boundary-layer scheme (MLE_USE_PBL_MLD, h_MLD), MLE_MLD_STRETCH, the two running
 * means MLE_MLD_DECAY_TIME / _TIME2 kept in MLD_filtered / MLD_filtered_slow, FOX_KEMPER_ML_RESTRAT_COEF / _COEF2, the frontal
 * length scale MLE_FRONT_LENGTH with VarMix%Rd_dx_h, the shape function mu(sigma, MLE_TAIL_DH) :723) when nkml == 0, and
 * mixedlayer_restrat_BML (:1209; the bulk mixed layer of nkml layers) otherwise; Boussinesq.
 * Refused: USE_BODNER23, USE_STANLEY_ML, non-Boussinesq. */
typedef struct mom6hip_mixedlayer_restrat_cs {
  double ml_restrat_coef;      /* FOX_KEMPER_ML_RESTRAT_COEF (0) */
  double ml_restrat_coef2;     /* FOX_KEMPER_ML_RESTRAT_COEF2 (0) */
  double front_length;         /* MLE_FRONT_LENGTH [L] (0); positive needs Rd_dx_h */
  double vonKar;               /* VON_KARMAN_CONST (0.41) */
  double MLE_MLD_decay_time;   /* MLE_MLD_DECAY_TIME [T] (0); positive needs MLD_filtered */
  double MLE_MLD_decay_time2;  /* MLE_MLD_DECAY_TIME2 [T] (0); positive needs MLD_filtered_slow */
  double MLE_density_diff;     /* MLE_DENSITY_DIFF [R] (0.03); not positive: MLE_use_PBL_MLD must be set */
  double MLE_tail_dh;          /* MLE_TAIL_DH (0) */
  double MLE_MLD_stretch;      /* MLE_MLD_STRETCH (1) */
  double ustar_min;            /* RESTRAT_USTAR_MIN [H T-1] (2e-4 * OMEGA * (Angstrom_Z + dZ_subroundoff)) */
  double reserved0[4];
  int32_t MLE_use_PBL_MLD;     /* MLE_USE_PBL_MLD (0): the mixed layer depth is h_MLD of the call */
  int32_t nkml;                /* GV%nkml (0): > 0 selects mixedlayer_restrat_BML */
  int32_t initialized;
  int32_t reserved_i[1];
  int32_t unsupported[8];      /* Bodner, Stanley, non-Boussinesq: any nonzero is refused */
  double *MLD_filtered;        /* CS%MLD_filtered, h points 2-D (state; valid halo of 1; updated over the compute domain + 1) */
  double *MLD_filtered_slow;   /* CS%MLD_filtered_slow */
  const double *Rd_dx_h;       /* VarMix%Rd_dx_h, h points 2-D (valid halo of 1) */
  void *reserved1[3];
} mom6hip_mixedlayer_restrat_cs_t;

/* mixedlayer_restrat(h, uhtr, vhtr, tv, forces, dt, MLD, h_MLD, bflux, VarMix, G, GV, US, CS)                       :135
 * tv%T, tv%S, tv%eqn_of_state as T, S, eos (required: the module stops without an equation of state); forces%ustar [Z T-1] as
 * ustar (valid halo of 1, like h, T, S); h_MLD [H] may be NULL unless MLE_use_PBL_MLD.  h, uhtr, vhtr are updated on the compute
 * domain; uhml / vhml receive the restratifying transports when not NULL (the diagnostics uhml, vhml). */
int mom6hip_mixedlayer_restrat(mom6hip_ctx_t *ctx, const mom6hip_mixedlayer_restrat_cs_t *cs, double *h, double *uhtr, double *vhtr,
                               const double *T, const double *S, const mom6hip_eos_t *eos, const double *ustar, double dt,
                               const double *h_MLD, double *uhml, double *vhml, int32_t memspace);
/* mu(sigma, dh) :723, the shape of the streamfunction (host; the known answers of mixedlayer_restrat_unit_tests :1847 hold for it) */
double mom6hip_mixedlayer_restrat_mu(double sigma, double dh);

/*
 * MOM_dyn_split_RK2_CS, src/core/MOM_dynamics_split_RK2.F90:84-268: the parameters the provided branch reads, the
 * control structures of the modules the step calls, and the arrays the reference keeps in the control structure
 * between steps.  Every array is a DEVICE array owned by the caller (a MOM6 executable would keep these mirrors
 * for the whole run and copy out only what diagnostics and restarts need).
 */
typedef struct mom6hip_dyn_split_rk2_cs {
  double be;                     /* BE (0.6) */
  double begw;                   /* BEGW; must be 0 */
  int32_t BT_use_layer_fluxes;   /* BT_USE_LAYER_FLUXES (1) */
  int32_t store_CAu;             /* STORE_CORIOLIS_ACCEL (1) */
  int32_t CAu_pred_stored;       /* state: CAu_pred / CAv_pred hold the next predictor's Coriolis terms */
  int32_t split_bottom_stress;   /* SPLIT_BOTTOM_STRESS; must be 0 */
  int32_t reserved0[4];
  const mom6hip_continuity_cs_t *continuity_CSp;
  const mom6hip_coriolisadv_cs_t *CoriolisAdv;
  const mom6hip_pressureforce_cs_t *PressureForce_CSp;
  const mom6hip_eos_t *eqn_of_state;          /* tv%eqn_of_state */
  mom6hip_barotropic_cs_t *barotropic_CSp;    /* its arrays are DEVICE arrays too */
  const mom6hip_bt_cont_t *BT_cont;           /* NULL: USE_BT_CONT_TYPE = False */
  const mom6hip_visc_hooks_t *hooks;          /* NULL: no host-side parameterisations */
  mom6hip_vertvisc_cs_t *vertvisc_CSp;        /* NULL: no vertical viscosity from this library (its arrays are DEVICE arrays) */
  const mom6hip_vertvisc_type_t *visc;        /* the visc argument of the step (DEVICE arrays); needed with vertvisc_CSp */
  const mom6hip_hor_visc_cs_t *hor_visc;      /* NULL: no horizontal viscosity from this library (diffu = diffv = 0, or the hook) */
  /* 3-D */
  double *CAu, *CAv, *CAu_pred, *CAv_pred, *PFu, *PFv, *diffu, *diffv, *visc_rem_u, *visc_rem_v, *u_accel_bt,
      *v_accel_bt, *u_av, *v_av, *h_av, *pbce;
  /* 2-D */
  double *eta, *eta_PF, *uhbt, *vhbt;
  /* SPLIT_RK2B only (MOM_dynamics_split_RK2b.F90:141-146): the barotropic velocity increments between the filtered and
   * the instantaneous velocities, at u / v points; restart fields du_avg_inst / dv_avg_inst (:1181-1186) */
  double *du_av_inst, *dv_av_inst;
  const mom6hip_set_visc_cs_t *set_visc_CSp;  /* NULL, or with dynamic_viscous_ML: set_viscous_ML is called at :592 (visc->ustar,
                                                 visc->nkml_visc_u/v must be set) */
  const struct mom6hip_obc *OBC;              /* CS%OBC (:253): NULL, or the open boundaries (the arrays of its segments, rx_normal and
                                                 ry_normal DEVICE arrays): step_MOM_dyn_split_RK2 only, see mom6hip_step_dyn_split_rk2 */
  /* The surface pressures of the call in progress (arguments of step_MOM_dyn_split_RK2 :289 / _RK2b :274, set before the call; NULL =
   * not associated; DEVICE h-point arrays [R L2 T-2 ~> Pa]): with p_surf_begin and p_surf_end both given, PressureForce takes
   * p_surf_end as p_atm and btstep takes eta_PF_start = eta_PF - (p_surf_begin - p_surf_end) / (g_Earth H_to_RZ) (:435-442, :495-503);
   * otherwise PressureForce takes p_surf = forces%p_surf (or nothing). */
  const double *p_surf_begin, *p_surf_end, *p_surf;
} mom6hip_dyn_split_rk2_cs_t;

/* The part of initialize_dyn_split_RK2 (:1326) that sets state: eta from the layer thicknesses (:1521-1535),
 * u_av = u, v_av = v (:1552-1558), diffu/diffv (hook or zero), visc_rem = 1 without hooks, and for store_CAu the
 * first uh, vh, h_av, CAu_pred, CAv_pred (:1560-1588) with their halo passes (:1615-1622).
 * barotropic_init must have been called on cs->barotropic_CSp. */
int mom6hip_dyn_split_rk2_init(mom6hip_ctx_t *ctx, mom6hip_dyn_split_rk2_cs_t *cs, const double *u, const double *v,
                               const double *h, double *uh, double *vh, double dt);

/*
 * step_MOM_dyn_split_RK2(u_inst, v_inst, h, tv, visc, Time_local, dt, forces, p_surf_begin, p_surf_end, uh, vh, uhtr,
 *                        vhtr, eta_av, G, GV, US, CS, calc_dtbt, VarMix, MEKE, thickness_diffuse_CSp, pbv, STOCH, Waves)
 *                                                                  src/core/MOM_dynamics_split_RK2.F90:289
 * tv%T, tv%S are passed as T, S; forces%taux, %tauy as taux, tauy with RZ_to_H; p_surf_begin, p_surf_end and forces%p_surf
 * through cs->p_surf_begin, cs->p_surf_end, cs->p_surf; Waves, FPMIX and BEGW /= 0 are not provided.  All arrays are DEVICE arrays.  The whole step is enqueued on the context's
 * stream without host synchronisation, except for set_dtbt when calc_dtbt /= 0, the group passes of a multi-tile
 * domain and the hooks.
 * With cs->OBC (CS%OBC associated): the step's lines for the open boundaries -- u_old_rad_OBC = u_av :444-456,
 * open_boundary_zero_normal_flow on u_bc_accel :565-567 and :887-889, radiation_open_bdry_conds on u_av :765-775 and on u_inst
 * :1030-1034 (segment%normal_vel, OBC%rx_normal, OBC%ry_normal are updated in place) -- and the OBC in every operator that takes
 * one, as the plain sequence of the library's OBC entry points; update_OBC_data (:534-536, OBC%update_OBC) is the caller's business
 * before the call; no hooks.  On several tiles every tile passes the segments clipped to its own data domain, as open_boundary_config
 * leaves them on a PE (a segment that runs along a cut between tiles must keep more than a halo width from it: the reference's placement
 * rule, MOM_open_boundary.F90:1379, does not place it on the neighbouring tile).  mom6hip_dyn_split_rk2_init reads cs->OBC as well (:1543-1593).
 */
int mom6hip_step_dyn_split_rk2(mom6hip_ctx_t *ctx, mom6hip_dyn_split_rk2_cs_t *cs, double *u_inst, double *v_inst,
                               double *h, const double *T, const double *S, double dt, const double *taux,
                               const double *tauy, double RZ_to_H, double *uh, double *vh, double *uhtr, double *vhtr,
                               double *eta_av, int32_t calc_dtbt);

/*
 * SPLIT_RK2B = True (MOM.F90:2198, :1236): the alternate split time stepping, in which the filtered velocities u_av, v_av
 * are the prognostic velocities and the instantaneous ones are rebuilt from them each step,
 * u_inst = u_av - du_av_inst*visc_rem_u (:641-646).  Same control structure as step_MOM_dyn_split_RK2 with
 * cs->du_av_inst / dv_av_inst set; store_CAu, CAu_pred_stored and BT_use_layer_fluxes are not read (the scheme always
 * hands btstep the layer fluxes, :663-672), and cs->u_av, cs->v_av, cs->h_av are the step's work arrays u_inst, v_inst,
 * h_av (:338-341) -- they carry nothing from one step to the next.
 *
 * initialize_dyn_split_RK2b (:1220), the part that sets state: eta from the layer thicknesses (:1406-1420), du_av_inst =
 * dv_av_inst = 0 and diffu = diffv = 0 (:1155-1166), visc_rem = 1; a restart then overwrites sfc, du_avg_inst,
 * dv_avg_inst and the barotropic fields.
 */
int mom6hip_dyn_split_rk2b_init(mom6hip_ctx_t *ctx, mom6hip_dyn_split_rk2_cs_t *cs, const double *h);

/*
 * step_MOM_dyn_split_RK2b(u_av, v_av, h, tv, visc, Time_local, dt, forces, p_surf_begin, p_surf_end, uh, vh, uhtr, vhtr,
 *                         eta_av, G, GV, US, CS, calc_dtbt, VarMix, MEKE, thickness_diffuse_CSp, pbv, Waves)
 *                                                                  src/core/MOM_dynamics_split_RK2b.F90:274
 * Arguments as for mom6hip_step_dyn_split_rk2, with the filtered velocities in place of the instantaneous ones.
 */
int mom6hip_step_dyn_split_rk2b(mom6hip_ctx_t *ctx, mom6hip_dyn_split_rk2_cs_t *cs, double *u_av, double *v_av, double *h,
                                const double *T, const double *S, double dt, const double *taux, const double *tauy,
                                double RZ_to_H, double *uh, double *vh, double *uhtr, double *vhtr, double *eta_av,
                                int32_t calc_dtbt);

#ifdef __cplusplus
}
#endif
#endif /* MOM6HIP_H */
