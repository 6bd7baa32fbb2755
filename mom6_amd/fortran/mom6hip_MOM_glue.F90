!> What every drop-in module shim needs from the host model: a GPU context built from ocean_grid_type /
!! verticalGrid_type (all metrics the kernels read, SURVEY.md section 8b), the GPU chosen from the node-local rank, and
!! the collectives that happen INSIDE library calls routed to MOM6's own MOM_domains / MOM_coms:
!!   * the group passes (src/tracer/MOM_tracer_advect.F90:205-206, src/core/MOM_barotropic.F90:1842-1850,
!!     src/core/MOM_dynamics_split_RK2.F90:541 ...) -> pass_var on host copies of the device arrays;
!!   * sum_across_PEs(domore_k) (MOM_tracer_advect.F90:305) and min_across_PEs(dtbt_max) (MOM_barotropic.F90:2915).
!! The library's own wrap kernels are only used when the whole domain is one tile AND the shim knows the topology
!! (REENTRANT_X / REENTRANT_Y read from the parameter file); otherwise reentrant_x = reentrant_y = 0 and every halo update
!! goes through MOM6's pass_var, which is correct for any layout, mask table or periodicity.  TRIPOLAR_N is provided on ONE PE
!! (the library's own fold, whose fold-line convention btstep's polarity swaps assume); with several PEs it is refused: FMS's
!! pass_vector owns the sign and the fold-line row there, and that path has not run against FMS.
!!
!! Compiled inside a MOM6 source tree (it uses the real MOM_grid, MOM_domains, MOM_coms); tests/fortran/stubs holds
!! type-only stand-ins so that this repository can at least compile and drive it on one PE.
module mom6hip_MOM_glue

use, intrinsic :: iso_c_binding
use mom6hip_c_api
use MOM_coms,          only : num_PEs, PE_here, sum_across_PEs, min_across_PEs
use MOM_domains,       only : pass_var, CENTER, EAST_FACE, NORTH_FACE, CORNER
use MOM_error_handler, only : MOM_error, FATAL
use MOM_file_parser,   only : get_param, param_file_type
use MOM_grid,          only : ocean_grid_type
use MOM_open_boundary, only : ocean_OBC_type, OBC_segment_type, OBC_DIRECTION_W, OBC_DIRECTION_S
use MOM_string_functions, only : uppercase
use MOM_verticalGrid,  only : verticalGrid_type
implicit none ; private

public :: mom6hip_context_create, mom6hip_read_topology, mom6hip_read_eos, mom6hip_fatal_if, mom6hip_shared_context, mom6hip_shared_context_end
! the device mirrors of the host's arrays, shared by every module shim (GPU_RESIDENT_DYNAMICS)
public :: mom6hip_read_resident, mom6hip_resident, mom6hip_mirror, mom6hip_mirrors_stage, mom6hip_mirrors_to_host
public :: mom6hip_mirrors_host_was_modified, mom6hip_mirror_host_changed, mom6hip_mirror_zeroed, mom6hip_mirrors_end
public :: mom6hip_mirror_pass_var, mom6hip_mirror_require_host_current, mom6hip_mirror_forget
public :: mom6hip_obc_to_c, update_segment_tracer_reservoirs_hip

integer, parameter :: MAX_MIRRORS = 160
!> A host array of the caller and its copy in HBM
type :: dev_mirror
  type(c_ptr) :: h = c_null_ptr, d = c_null_ptr      !< host base address, device address
  integer(c_int64_t) :: bytes = 0
  logical :: dev_current = .false.    !< the device copy holds what the host copy holds, or something newer
  logical :: host_current = .true.    !< the host copy holds what the device copy holds, or something newer
end type dev_mirror
type(dev_mirror), save :: mir(MAX_MIRRORS)
integer, save :: nmir = 0
logical, save :: resident_mode = .false.     !< GPU_RESIDENT_DYNAMICS

!> The grid of the (single) ocean instance the callbacks act on
type(ocean_grid_type), pointer, save :: G_cb => NULL()
type(c_ptr), save :: ctx_cb = c_null_ptr   !< the context whose stream the staged copies use
integer, save :: nk_cb = 0
!> One context per process, shared by every module shim (the metrics are uploaded once): mom6hip_shared_context
type(c_ptr), save :: ctx_shared = c_null_ptr
logical, save :: topology_known = .false., reentrant_saved(2) = .false., tripolar_saved = .false.
logical, save :: on_fold_cb = .false.   !< this PE's tile ends at the tripolar fold (the halo callback negates vector components beyond it)

contains

!> Turn a nonzero return code of the library into MOM_error(FATAL) with the library's message (SURVEY.md 8b "Errors")
subroutine mom6hip_fatal_if(rc, who)
  integer(c_int),   intent(in) :: rc
  character(len=*), intent(in) :: who
  if (rc /= 0) call MOM_error(FATAL, trim(who)//" (HIP): "//mom6hip_error_string())
end subroutine mom6hip_fatal_if

!> REENTRANT_X / REENTRANT_Y / TRIPOLAR_N as MOM_domains reads them (src/framework/MOM_domains.F90:184-191).
!! Called from a module's *_init (which has the parameter file) so that the one-tile fast path knows the topology.
subroutine mom6hip_read_topology(param_file, reentrant)
  type(param_file_type), intent(in)  :: param_file
  logical,     optional, intent(out) :: reentrant(2)
  logical :: tripolar_N, re(2)
  call get_param(param_file, "mom6hip", "REENTRANT_X", re(1), default=.true., do_not_log=.true.)
  call get_param(param_file, "mom6hip", "REENTRANT_Y", re(2), default=.false., do_not_log=.true.)
  call get_param(param_file, "mom6hip", "TRIPOLAR_N", tripolar_N, default=.false., do_not_log=.true.)
  if (tripolar_N .and. .not.re(1)) call MOM_error(FATAL, "mom6hip: TRIPOLAR_N needs REENTRANT_X on the GPU path.")
  topology_known = .true. ; reentrant_saved(:) = re(:) ; tripolar_saved = tripolar_N
  if (present(reentrant)) reentrant(:) = re(:)
end subroutine mom6hip_read_topology

!> GPU_RESIDENT_DYNAMICS (a parameter of this port, default False), read by every shim that can work on the device mirrors:
!! False: a call uploads its inputs and copies its outputs back (correct inside an unmodified MOM6); True: the fields stay in HBM
!! between the calls of all shims, the host arrays are refreshed by mom6hip_mirrors_to_host (dyn_split_RK2_sync_to_host), and the
!! host announces what it has changed (mom6hip_mirror_host_changed / mom6hip_mirror_zeroed / dyn_split_RK2_host_was_modified).
subroutine mom6hip_read_resident(param_file, resident)
  type(param_file_type), intent(in)  :: param_file
  logical,     optional, intent(out) :: resident
  logical :: r
  call get_param(param_file, "mom6hip", "GPU_RESIDENT_DYNAMICS", r, &
                 "If true, the fields the GPU path works on stay in device memory between its calls.", default=.false.)
  resident_mode = r
  if (present(resident)) resident = r
end subroutine mom6hip_read_resident

logical function mom6hip_resident()
  mom6hip_resident = resident_mode
end function mom6hip_resident

!> The device mirror of the host array at hp (n doubles): created on first sight; uploaded when the host copy is the newer one (with
!! GPU_RESIDENT_DYNAMICS = False: at every call that reads it).  written: the call will write it (the device copy becomes the newer).
function mom6hip_mirror(ctx, hp, n, is_input, written) result(d)
  type(c_ptr),        intent(in) :: ctx, hp
  integer(c_int64_t), intent(in) :: n
  logical,            intent(in) :: is_input, written
  type(c_ptr) :: d
  integer :: m, q, rc
  q = 0
  do m = 1, nmir
    if (c_associated(mir(m)%h, hp)) then ; q = m ; exit ; endif
  enddo
  if (q > 0) then ; if (mir(q)%bytes /= 8_c_int64_t*n) then      ! the host reallocated something else at this address
    rc = mom6hip_free(mir(q)%d) ; call mom6hip_fatal_if(rc, "mom6hip_mirror")
    rc = mom6hip_malloc(mir(q)%d, 8_c_int64_t*n) ; call mom6hip_fatal_if(rc, "mom6hip_mirror")
    mir(q)%bytes = 8_c_int64_t*n ; mir(q)%dev_current = .false. ; mir(q)%host_current = .true.
  endif ; endif
  if (q == 0) then
    if (nmir >= MAX_MIRRORS) call MOM_error(FATAL, "mom6hip_mirror: too many host arrays to mirror.")
    nmir = nmir + 1 ; q = nmir
    mir(q)%h = hp ; mir(q)%bytes = 8_c_int64_t*n ; mir(q)%dev_current = .false. ; mir(q)%host_current = .true.
    rc = mom6hip_malloc(mir(q)%d, mir(q)%bytes) ; call mom6hip_fatal_if(rc, "mom6hip_mirror")
    if (.not.is_input) then
      rc = mom6hip_memset_zero(ctx, mir(q)%d, mir(q)%bytes) ; call mom6hip_fatal_if(rc, "mom6hip_mirror")
    endif
  endif
  if (is_input .and. (.not.mir(q)%dev_current .or. .not.resident_mode)) then
    if (mir(q)%host_current) then      ! (never overwrite a device copy that is newer than the host's)
      rc = mom6hip_sync_to_device(ctx, mir(q)%d, hp, mir(q)%bytes) ; call mom6hip_fatal_if(rc, "mom6hip_mirror (upload)")
    endif
    mir(q)%dev_current = .true.
  endif
  if (written) then ; mir(q)%dev_current = .true. ; mir(q)%host_current = .false. ; endif
  d = mir(q)%d
end function mom6hip_mirror

!> Start copying everything the device holds newer than the host back to the host arrays (snapshots on the compute stream, copies on
!! the copy stream); mom6hip_stage_wait completes them
subroutine mom6hip_mirrors_stage(ctx)
  type(c_ptr), intent(in) :: ctx
  integer :: m, rc
  do m = 1, nmir
    if (.not.mir(m)%host_current) then
      rc = mom6hip_stage_to_host(ctx, mir(m)%h, mir(m)%d, mir(m)%bytes) ; call mom6hip_fatal_if(rc, "mom6hip_mirrors_stage")
      mir(m)%host_current = .true.
    endif
  enddo
end subroutine mom6hip_mirrors_stage

!> The same, complete: call it where the host reads the fields
subroutine mom6hip_mirrors_to_host(ctx)
  type(c_ptr), intent(in) :: ctx
  integer :: rc
  call mom6hip_mirrors_stage(ctx)
  rc = mom6hip_stage_wait(ctx) ; call mom6hip_fatal_if(rc, "mom6hip_mirrors_to_host")
end subroutine mom6hip_mirrors_to_host

!> The host has changed fields (thermodynamics, ALE remapping, a new forcing ...): the next calls upload every input again
subroutine mom6hip_mirrors_host_was_modified()
  integer :: m
  do m = 1, nmir
    if (.not.mir(m)%host_current) call MOM_error(FATAL, "mom6hip_mirrors_host_was_modified: the device holds newer values of a "// &
        "field than the host; call mom6hip_mirrors_to_host (dyn_split_RK2_sync_to_host) before the host changes the fields.")
    mir(m)%dev_current = .false.
  enddo
end subroutine mom6hip_mirrors_host_was_modified

!> The host has changed this one array (a new wind stress, MEKE%Kh after the MEKE step ...): its next reader uploads it
subroutine mom6hip_mirror_host_changed(hp)
  type(c_ptr), intent(in) :: hp
  integer :: m
  do m = 1, nmir ; if (c_associated(mir(m)%h, hp)) then
    mir(m)%dev_current = .false. ; mir(m)%host_current = .true.
  endif ; enddo
end subroutine mom6hip_mirror_host_changed

!> The host has set this array to zero (uhtr, vhtr after the tracer advection, MOM.F90:1447): zero the device copy instead of uploading
subroutine mom6hip_mirror_zeroed(ctx, hp)
  type(c_ptr), intent(in) :: ctx, hp
  integer :: m, rc
  do m = 1, nmir ; if (c_associated(mir(m)%h, hp)) then
    rc = mom6hip_memset_zero(ctx, mir(m)%d, mir(m)%bytes) ; call mom6hip_fatal_if(rc, "mom6hip_mirror_zeroed")
    mir(m)%dev_current = .true. ; mir(m)%host_current = .true.
  endif ; enddo
end subroutine mom6hip_mirror_zeroed

!> pass_var between two calls of the GPU path (MOM.F90:1179, :1338 ...): if the device holds the newer copy of the array at hp, its halos
!! are updated THERE (the library's group pass: the wrap kernels of a one-tile domain, RCCL, or the host's pass_var behind the
!! callbacks) and done = .true.; otherwise done = .false. and the caller's own pass_var on the host array is the right one.
!! position: CENTER / EAST_FACE / NORTH_FACE / CORNER of MOM_domains; nk: the number of layers (1 for a 2-D field).
subroutine mom6hip_mirror_pass_var(ctx, hp, nk, position, done)
  type(c_ptr), intent(in)  :: ctx, hp
  integer,     intent(in)  :: nk, position
  logical,     intent(out) :: done
  type(c_ptr) :: f(1)
  integer(c_int32_t) :: pos(1), nks(1)
  integer :: m, rc
  done = .false.
  do m = 1, nmir ; if (c_associated(mir(m)%h, hp) .and. .not.mir(m)%host_current) then
    f(1) = mir(m)%d ; nks(1) = nk ; pos(1) = MOM6HIP_POS_H
    if (position == EAST_FACE) pos(1) = MOM6HIP_POS_U
    if (position == NORTH_FACE) pos(1) = MOM6HIP_POS_V
    if (position == CORNER) pos(1) = MOM6HIP_POS_Q
    rc = mom6hip_halo_update(ctx, f, pos, nks, 1_c_int32_t) ; call mom6hip_fatal_if(rc, "mom6hip_mirror_pass_var")
    done = .true.
  endif ; enddo
end subroutine mom6hip_mirror_pass_var

!> A shim that works on HOST arrays is about to read the array at hp: stop if the device holds a newer copy of it (the caller forgot
!! dyn_split_RK2_sync_to_host / mom6hip_mirrors_to_host), instead of computing from stale values
subroutine mom6hip_mirror_require_host_current(hp, who)
  type(c_ptr),      intent(in) :: hp
  character(len=*), intent(in) :: who
  integer :: m
  do m = 1, nmir ; if (c_associated(mir(m)%h, hp) .and. .not.mir(m)%host_current) then
    call MOM_error(FATAL, trim(who)//" (HIP): the device holds a newer copy of an array this call reads on the host; "// &
                   "call dyn_split_RK2_sync_to_host (mom6hip_mirrors_to_host) first (GPU_RESIDENT_DYNAMICS).")
  endif ; enddo
end subroutine mom6hip_mirror_require_host_current

!> The host has changed fields after such a call (ALE remapping ...): mom6hip_mirrors_host_was_modified / mom6hip_mirror_host_changed.

!> The host array at hp is about to go away (deallocate, the end of a *_end routine): drop its mirror, so that whatever the host
!! allocates at that address next is not taken for it.  A device copy newer than the host's is copied back first -- the host array
!! still exists at this point -- rather than thrown away.  Only PERSISTENT host arrays may be mirrored (module or control-structure
!! members, the model's state): a mirror of an automatic or temporary array outlives it, and the next array at that address with
!! the same size would inherit its device copy (INTEGRATION.md section 2c).
subroutine mom6hip_mirror_forget(hp)
  type(c_ptr), intent(in) :: hp
  type(c_ptr) :: ctx
  integer :: m, q, rc
  ctx = ctx_shared
  q = 0
  do m = 1, nmir ; if (c_associated(mir(m)%h, hp)) then ; q = m ; exit ; endif ; enddo
  if (q == 0) return
  if (.not.mir(q)%host_current) then
    if (.not.c_associated(ctx)) call MOM_error(FATAL, "mom6hip_mirror_forget: a device copy is newer than the host's, and the context has ended.")
    rc = mom6hip_sync_to_host(ctx, hp, mir(q)%d, mir(q)%bytes) ; call mom6hip_fatal_if(rc, "mom6hip_mirror_forget")
  endif
  rc = mom6hip_free(mir(q)%d) ; call mom6hip_fatal_if(rc, "mom6hip_mirror_forget")
  if (q < nmir) mir(q) = mir(nmir)
  mir(nmir)%h = c_null_ptr ; mir(nmir)%d = c_null_ptr
  nmir = nmir - 1
end subroutine mom6hip_mirror_forget

!> Free the mirrors (the last *_end of the run).  A device copy that is newer than the host's is an error of the calling sequence
!! (dyn_split_RK2_sync_to_host / mom6hip_mirrors_to_host was not called before the end): stop, as mom6hip_mirrors_host_was_modified
!! does, instead of discarding the newer values.
subroutine mom6hip_mirrors_end()
  integer :: m, rc
  do m = 1, nmir
    if (.not.mir(m)%host_current) call MOM_error(FATAL, "mom6hip_mirrors_end: the device holds newer values of a field than the "// &
        "host; call mom6hip_mirrors_to_host (dyn_split_RK2_sync_to_host) before the model ends.")
  enddo
  do m = 1, nmir ; rc = mom6hip_free(mir(m)%d) ; mir(m)%h = c_null_ptr ; mir(m)%d = c_null_ptr ; enddo
  nmir = 0
end subroutine mom6hip_mirrors_end

!> The equation of state as interpret_eos_selection reads it (MOM_EOS.F90:1474-1520): EOS_type is opaque in MOM6, so the shims
!! that need the equation of state (MOM_PressureForce_FV, MOM_thickness_diffuse) read its selection from the parameter file.
subroutine mom6hip_read_eos(param_file, eos, who)
  type(param_file_type), intent(in)  :: param_file
  type(mom6hip_eos_t),   intent(out) :: eos
  character(len=*),      intent(in)  :: who
  character(len=40) :: tmpstr
  call get_param(param_file, "MOM_EOS", "EQN_OF_STATE", tmpstr, &
                 "EQN_OF_STATE determines which ocean equation of state should be used.", default="WRIGHT")
  eos%reserved = 0 ; eos%Rho_T0_S0 = 1000.0 ; eos%dRho_dT = -0.2 ; eos%dRho_dS = 0.8
  select case (uppercase(tmpstr))
    case ("LINEAR")
      eos%form = MOM6HIP_EOS_LINEAR
      call get_param(param_file, "MOM_EOS", "RHO_T0_S0", eos%Rho_T0_S0, units="kg m-3", default=1000.0)
      call get_param(param_file, "MOM_EOS", "DRHO_DT", eos%dRho_dT, units="kg m-3 K-1", default=-0.2)
      call get_param(param_file, "MOM_EOS", "DRHO_DS", eos%dRho_dS, units="kg m-3 ppt-1", default=0.8)
    case ("WRIGHT")
      eos%form = MOM6HIP_EOS_WRIGHT
    case ("UNESCO", "JACKETT_MCD")
      eos%form = MOM6HIP_EOS_UNESCO
    case ("WRIGHT_FULL")
      eos%form = MOM6HIP_EOS_WRIGHT_FULL
    case ("WRIGHT_REDUCED")
      eos%form = MOM6HIP_EOS_WRIGHT_REDUCED
    case default
      call MOM_error(FATAL, who//" (HIP): EQN_OF_STATE "//trim(tmpstr)//" is not provided by the GPU path "// &
                            "(WRIGHT, WRIGHT_FULL, WRIGHT_REDUCED, UNESCO, LINEAR).")
  end select
end subroutine mom6hip_read_eos

!> The process-wide context (created on the first call, from whichever module gets there first).  The *_init of every
!! shim calls mom6hip_read_topology, so the one-tile fast path knows REENTRANT_X / REENTRANT_Y by then.
function mom6hip_shared_context(G, GV) result(ctx)
  type(ocean_grid_type), target, intent(in) :: G
  type(verticalGrid_type),       intent(in) :: GV
  type(c_ptr) :: ctx
  if (.not. c_associated(ctx_shared)) then
    if (topology_known) then ; call mom6hip_context_create(G, GV, ctx_shared, reentrant_saved)
    else ; call mom6hip_context_create(G, GV, ctx_shared) ; endif
  endif
  ctx = ctx_shared
end function mom6hip_shared_context

!> Release the shared context (the *_end of the last module; harmless when called again)
subroutine mom6hip_shared_context_end()
  integer :: rc
  if (c_associated(ctx_shared)) rc = mom6hip_grid_destroy(ctx_shared)
  ctx_shared = c_null_ptr
end subroutine mom6hip_shared_context_end

!> Upload the metrics of G once and return the context; registers the collectives when they are needed.
subroutine mom6hip_context_create(G, GV, ctx, reentrant)
  type(ocean_grid_type), target,   intent(in)    :: G
  type(verticalGrid_type),         intent(in)    :: GV
  type(c_ptr),                     intent(out)   :: ctx
  logical,               optional, intent(in)    :: reentrant(2) !< from mom6hip_read_topology: enables the one-tile fast path
  type(mom6hip_grid_t) :: cg
  logical :: reentrant_x, reentrant_y, know_topology
  integer :: gpus_per_node, rc
  character(len=16) :: env

  if (.not. G%symmetric) call MOM_error(FATAL, "mom6hip: SYMMETRIC_MEMORY_ is required.")
  cg%isc = G%isc ; cg%iec = G%iec ; cg%jsc = G%jsc ; cg%jec = G%jec
  cg%isd = G%isd ; cg%ied = G%ied ; cg%jsd = G%jsd ; cg%jed = G%jed
  cg%nk = GV%ke ; cg%symmetric = 1 ; cg%first_direction = G%first_direction
  know_topology = present(reentrant) ; reentrant_x = .false. ; reentrant_y = .false.
  if (know_topology) then ; reentrant_x = reentrant(1) ; reentrant_y = reentrant(2) ; endif
  cg%reentrant_x = 0 ; cg%reentrant_y = 0
  if (know_topology .and. (num_PEs() == 1)) then
    ! the whole domain is this tile: the library's wrap kernels are the halo update
    cg%reentrant_x = merge(1, 0, reentrant_x) ; cg%reentrant_y = merge(1, 0, reentrant_y)
  endif
  ! TRIPOLAR_N: a tile that ends at the fold (btstep swaps the directional fits in its halo rows beyond the fold,
  ! MOM_barotropic.F90:1471-1475, :4036-4064).  One PE: the library folds the halos itself.  Several PEs are refused: a scalar
  ! pass_var with the sign of vector components applied afterwards leaves the fold-line row (v and q at J = jec) to FMS's scalar
  ! update, while FMS's pass_vector would enforce v(i,nj) = -v(ni+1-i,nj) and the library's fold leaves the row as computed; the
  ! answers would depend on the layout.  (To lift this: carry the u / v partner through the callback and call pass_vector.)
  cg%tripolar_n = 0 ; on_fold_cb = .false.
  if (know_topology .and. tripolar_saved .and. (num_PEs() > 1)) &
    call MOM_error(FATAL, "mom6hip: TRIPOLAR_N with more than one PE is not provided by the GPU path (the fold-line "// &
                          "convention of FMS's pass_vector has not been verified against the library's).")
  if (know_topology .and. tripolar_saved) then
    if (G%jec + G%jdg_offset == G%Domain%njglobal) then ; cg%tripolar_n = 1 ; on_fold_cb = .true. ; endif
    if (G%iec - G%isc + 1 /= G%Domain%niglobal) &
      call MOM_error(FATAL, "mom6hip: with TRIPOLAR_N the GPU path needs tiles that span x (LAYOUT = 1,N).")
  endif
  cg%Angstrom_H = GV%Angstrom_H ; cg%H_subroundoff = GV%H_subroundoff
  cg%dZ_subroundoff = GV%dZ_subroundoff ; cg%H_to_Z = GV%H_to_Z ; cg%Z_to_H = GV%Z_to_H
  cg%g_Earth = GV%g_Earth ; cg%Rho0 = GV%Rho0
  cg%mask2dT = c_loc(G%mask2dT) ; cg%areaT = c_loc(G%areaT) ; cg%IareaT = c_loc(G%IareaT)
  cg%dxT = c_loc(G%dxT) ; cg%dyT = c_loc(G%dyT) ; cg%IdxT = c_loc(G%IdxT) ; cg%IdyT = c_loc(G%IdyT)
  cg%bathyT = c_loc(G%bathyT)
  cg%mask2dCu = c_loc(G%mask2dCu) ; cg%dxCu = c_loc(G%dxCu) ; cg%dyCu = c_loc(G%dyCu) ; cg%dy_Cu = c_loc(G%dy_Cu)
  cg%IdxCu = c_loc(G%IdxCu) ; cg%IdyCu = c_loc(G%IdyCu) ; cg%areaCu = c_loc(G%areaCu) ; cg%IareaCu = c_loc(G%IareaCu)
  cg%mask2dCv = c_loc(G%mask2dCv) ; cg%dxCv = c_loc(G%dxCv) ; cg%dyCv = c_loc(G%dyCv) ; cg%dx_Cv = c_loc(G%dx_Cv)
  cg%IdxCv = c_loc(G%IdxCv) ; cg%IdyCv = c_loc(G%IdyCv) ; cg%areaCv = c_loc(G%areaCv) ; cg%IareaCv = c_loc(G%IareaCv)
  cg%mask2dBu = c_loc(G%mask2dBu) ; cg%dxBu = c_loc(G%dxBu) ; cg%dyBu = c_loc(G%dyBu) ; cg%areaBu = c_loc(G%areaBu)
  cg%IareaBu = c_loc(G%IareaBu) ; cg%CoriolisBu = c_loc(G%CoriolisBu)
  cg%IdxBu = c_loc(G%IdxBu) ; cg%IdyBu = c_loc(G%IdyBu)

  ! one PE <-> one GPU: the node-local rank picks the device (MOM6HIP_GPUS_PER_NODE, default 8 on an MI355X node)
  gpus_per_node = 8
  call get_environment_variable("MOM6HIP_GPUS_PER_NODE", env, status=rc)
  if (rc == 0) read(env, *, iostat=rc) gpus_per_node
  if (gpus_per_node < 1) gpus_per_node = 1
  rc = mom6hip_init(int(mod(PE_here(), gpus_per_node), c_int))
  if (rc == 0) rc = mom6hip_grid_create(cg, c_null_ptr, ctx)
  call mom6hip_fatal_if(rc, "mom6hip_context_create")

  if ((.not.know_topology) .or. (num_PEs() > 1)) then
    ! every halo update inside a library call is MOM6's own pass_var; sums and minima are MOM_coms'
    ! (one PE with a known topology, closed basins included: the library's own updates, nothing leaves the device)
    G_cb => G ; ctx_cb = ctx ; nk_cb = GV%ke
    rc = mom6hip_set_domain_callbacks(ctx, c_funloc(halo_cb), c_funloc(sum_cb), c_null_ptr)
    if (rc == 0) rc = mom6hip_set_min_callback(ctx, c_funloc(min_cb), c_null_ptr)
    call mom6hip_fatal_if(rc, "mom6hip_context_create")
  endif
end subroutine mom6hip_context_create

!> mom6hip_halo_fn: the library hands over DEVICE arrays; they are copied to host scratch of the reference's shape for
!! their staggering, updated with pass_var on G%Domain, and copied back.
function halo_cb(user, fields, pos, nk, nfields) bind(c) result(rc)
  type(c_ptr),        value      :: user
  type(c_ptr),        intent(in) :: fields(*)
  integer(c_int32_t), intent(in) :: pos(*), nk(*)
  integer(c_int32_t), value      :: nfields
  integer(c_int) :: rc
  real(c_double), allocatable, target :: buf(:,:,:)
  integer :: f, i0, j0, position, p
  logical :: vector_cmpt
  integer(c_int64_t) :: bytes

  rc = 1
  if (.not. associated(G_cb)) return
  do f = 1, nfields
    if (.not. c_associated(fields(f))) cycle
    i0 = G_cb%isd ; j0 = G_cb%jsd ; position = CENTER
    p = iand(int(pos(f)), 3)
    ! a u- or v-point field that is not one of a SCALAR_PAIR is a vector component: it changes sign across the fold
    vector_cmpt = (p == MOM6HIP_POS_U .or. p == MOM6HIP_POS_V) .and. (iand(int(pos(f)), MOM6HIP_PASS_SCALAR_PAIR) == 0)
    select case (p)
      case (MOM6HIP_POS_U) ; i0 = G_cb%IsdB ; position = EAST_FACE
      case (MOM6HIP_POS_V) ; j0 = G_cb%JsdB ; position = NORTH_FACE
      case (MOM6HIP_POS_Q) ; i0 = G_cb%IsdB ; j0 = G_cb%JsdB ; position = CORNER
    end select
    allocate(buf(i0:G_cb%ied, j0:G_cb%jed, nk(f)))
    bytes = int(size(buf), c_int64_t) * 8_c_int64_t
    if (mom6hip_sync_to_host(ctx_cb, c_loc(buf), fields(f), bytes) /= 0) return
    call pass_var(buf, G_cb%Domain, position=position)
    ! (pass_var moves a scalar at this staggering: the image across the fold without the sign of a vector component)
    if (vector_cmpt .and. on_fold_cb) buf(:, G_cb%jec+1:G_cb%jed, :) = -buf(:, G_cb%jec+1:G_cb%jed, :)
    if (mom6hip_sync_to_device(ctx_cb, fields(f), c_loc(buf), bytes) /= 0) return
    deallocate(buf)
  enddo
  rc = 0
end function halo_cb

!> mom6hip_sum_fn: sum_across_PEs of n integers
function sum_cb(user, values, n) bind(c) result(rc)
  type(c_ptr),        value         :: user
  integer(c_int32_t), intent(inout) :: values(*)
  integer(c_int32_t), value         :: n
  integer(c_int) :: rc
  integer, allocatable :: v(:)
  allocate(v(n)) ; v(:) = values(1:n)
  call sum_across_PEs(v, n)
  values(1:n) = int(v(:), c_int32_t)
  rc = 0
end function sum_cb

!> mom6hip_min_fn: min_across_PEs of n reals
function min_cb(user, values, n) bind(c) result(rc)
  type(c_ptr),    value         :: user
  real(c_double), intent(inout) :: values(*)
  integer(c_int32_t), value     :: n
  integer(c_int) :: rc
  integer :: q
  real :: x
  do q = 1, n
    x = values(q) ; call min_across_PEs(x) ; values(q) = x
  enddo
  rc = 0
end function min_cb

!> What the library reads of ocean_OBC_type and its segments (MOM_open_boundary.F90:146-386) as a mom6hip_obc_t of host pointers:
!! the caller keeps csegs (and OBC) alive for the call.  Provided entry points: continuity_PPM and CorAdCalc (round 4).
subroutine mom6hip_obc_to_c(OBC, cobc, csegs, n_u2, n_v2, who, ctrs)
  type(ocean_OBC_type), target, intent(in)  :: OBC
  type(mom6hip_obc_t),          intent(out) :: cobc
  type(mom6hip_obc_segment_t), allocatable, target, intent(inout) :: csegs(:)
  integer,                      intent(in)  :: n_u2, n_v2   !< the sizes of a 2-D array at the u and v points of the data domain
  character(len=*),             intent(in)  :: who
  type(mom6hip_obc_segment_tracer_t), allocatable, target, optional, intent(inout) :: ctrs(:) !< with it, the tracer registries of the
                                                            !! segments (segment%tr_Reg) are handed over as well (advect_tracer)
  integer :: n, m, nt
  if (allocated(csegs)) deallocate(csegs)
  allocate(csegs(max(OBC%number_of_segments, 1)))
  do n=1,OBC%number_of_segments
    call segment_to_c(OBC%segment(n), csegs(n))
  enddo
  if (present(ctrs)) then
    if (allocated(ctrs)) deallocate(ctrs)
    nt = 0
    do n=1,OBC%number_of_segments ; if (associated(OBC%segment(n)%tr_Reg)) nt = nt + OBC%segment(n)%tr_Reg%ntseg ; enddo
    allocate(ctrs(nt+1))
    nt = 0
    do n=1,OBC%number_of_segments ; if (associated(OBC%segment(n)%tr_Reg)) then
      csegs(n)%tr_Reg = c_loc(ctrs(nt+1)) ; csegs(n)%ntseg = OBC%segment(n)%tr_Reg%ntseg
      do m=1,OBC%segment(n)%tr_Reg%ntseg
        nt = nt + 1
        ctrs(nt)%ntr_index = OBC%segment(n)%tr_Reg%Tr(m)%ntr_index
        ctrs(nt)%OBC_inflow_conc = OBC%segment(n)%tr_Reg%Tr(m)%OBC_inflow_conc
        if (allocated(OBC%segment(n)%tr_Reg%Tr(m)%tres)) ctrs(nt)%tres = c_loc(OBC%segment(n)%tr_Reg%Tr(m)%tres)
        if (allocated(OBC%segment(n)%tr_Reg%Tr(m)%t)) ctrs(nt)%t = c_loc(OBC%segment(n)%tr_Reg%Tr(m)%t)
        if (OBC%segment(n)%tr_Reg%Tr(m)%fd_index /= -1) then      ! (update_segment_tracer_reservoirs :5431-5437)
          ctrs(nt)%resrv_lfac_in = OBC%segment(n)%field(OBC%segment(n)%tr_Reg%Tr(m)%fd_index)%resrv_lfac_in
          ctrs(nt)%resrv_lfac_out = OBC%segment(n)%field(OBC%segment(n)%tr_Reg%Tr(m)%fd_index)%resrv_lfac_out
        endif
      enddo
      csegs(n)%Tr_InvLscale_in = OBC%segment(n)%Tr_InvLscale_in ; csegs(n)%Tr_InvLscale_out = OBC%segment(n)%Tr_InvLscale_out
    endif ; enddo
  endif
  cobc%number_of_segments = OBC%number_of_segments ; cobc%OBC_pe = merge(1, 0, OBC%OBC_pe)
  cobc%open_u_BCs_exist_globally = merge(1, 0, OBC%open_u_BCs_exist_globally)
  cobc%open_v_BCs_exist_globally = merge(1, 0, OBC%open_v_BCs_exist_globally)
  cobc%specified_u_BCs_exist_globally = merge(1, 0, OBC%specified_u_BCs_exist_globally)
  cobc%specified_v_BCs_exist_globally = merge(1, 0, OBC%specified_v_BCs_exist_globally)
  cobc%Flather_u_BCs_exist_globally = merge(1, 0, OBC%Flather_u_BCs_exist_globally)
  cobc%Flather_v_BCs_exist_globally = merge(1, 0, OBC%Flather_v_BCs_exist_globally)
  cobc%zero_vorticity = merge(1, 0, OBC%zero_vorticity) ; cobc%freeslip_vorticity = merge(1, 0, OBC%freeslip_vorticity)
  cobc%computed_vorticity = merge(1, 0, OBC%computed_vorticity) ; cobc%specified_vorticity = merge(1, 0, OBC%specified_vorticity)
  cobc%zero_strain = merge(1, 0, OBC%zero_strain) ; cobc%freeslip_strain = merge(1, 0, OBC%freeslip_strain)
  cobc%computed_strain = merge(1, 0, OBC%computed_strain) ; cobc%zero_biharmonic = merge(1, 0, OBC%zero_biharmonic)
  cobc%segment = c_loc(csegs)
  if (OBC%number_of_segments > 0) then
    if (.not.(allocated(OBC%segnum_u) .and. allocated(OBC%segnum_v))) call MOM_error(FATAL, &
      who//" (HIP): OBC%segnum_u and OBC%segnum_v must be allocated.")
    if (size(OBC%segnum_u) /= n_u2 .or. size(OBC%segnum_v) /= n_v2) call MOM_error(FATAL, &
      who//" (HIP): OBC%segnum_u / segnum_v do not have the shape of the u / v points of the data domain.")
    cobc%segnum_u = c_loc(OBC%segnum_u) ; cobc%segnum_v = c_loc(OBC%segnum_v)
  endif
contains
  subroutine segment_to_c(seg, c)
    type(OBC_segment_type), target, intent(in) :: seg
    type(mom6hip_obc_segment_t), intent(out) :: c
    c%direction = seg%direction ; c%open = merge(1, 0, seg%open) ; c%specified = merge(1, 0, seg%specified)
    c%on_pe = merge(1, 0, seg%on_pe) ; c%is_E_or_W = merge(1, 0, seg%is_E_or_W) ; c%is_N_or_S = merge(1, 0, seg%is_N_or_S)
    c%IsdB = seg%HI%IsdB ; c%IedB = seg%HI%IedB ; c%JsdB = seg%HI%JsdB ; c%JedB = seg%HI%JedB
    c%isd = seg%HI%isd ; c%ied = seg%HI%ied ; c%jsd = seg%HI%jsd ; c%jed = seg%HI%jed
    if (.not.seg%on_pe) return
    if (seg%specified) then
      if (.not.(allocated(seg%normal_trans) .and. allocated(seg%normal_vel))) call MOM_error(FATAL, &
        who//" (HIP): a specified segment needs normal_trans and normal_vel.")
      c%normal_trans = c_loc(seg%normal_trans) ; c%normal_vel = c_loc(seg%normal_vel)
    endif
    c%Flather = merge(1, 0, seg%Flather) ; c%radiation = merge(1, 0, seg%radiation) ; c%gradient = merge(1, 0, seg%gradient)
    c%nudged = merge(1, 0, seg%nudged) ; c%oblique = merge(1, 0, seg%oblique)
    if ((seg%gradient .or. seg%radiation .or. seg%oblique) .and. allocated(seg%normal_vel)) c%normal_vel = c_loc(seg%normal_vel)
    if (allocated(seg%normal_vel_bt)) c%normal_vel_bt = c_loc(seg%normal_vel_bt)
    if (allocated(seg%SSH)) c%SSH = c_loc(seg%SSH)
    if (allocated(seg%tangential_vel)) c%tangential_vel = c_loc(seg%tangential_vel)
    if (allocated(seg%tangential_grad)) c%tangential_grad = c_loc(seg%tangential_grad)
    if (allocated(seg%nudged_normal_vel)) c%nudged_normal_vel = c_loc(seg%nudged_normal_vel)
    c%radiation_tan_or_grad = merge(MOM6HIP_OBC_TAN_RADIATION, 0, seg%radiation_tan) + merge(MOM6HIP_OBC_GRAD_RADIATION, 0, seg%radiation_grad) + &
                              merge(MOM6HIP_OBC_TAN_NUDGED, 0, seg%nudged_tan) + merge(MOM6HIP_OBC_GRAD_NUDGED, 0, seg%nudged_grad) + &
                              merge(MOM6HIP_OBC_TAN_OBLIQUE, 0, seg%oblique_tan) + merge(MOM6HIP_OBC_GRAD_OBLIQUE, 0, seg%oblique_grad)
    if (allocated(seg%nudged_tangential_vel)) c%nudged_tangential_vel = c_loc(seg%nudged_tangential_vel)
    if (allocated(seg%nudged_tangential_grad)) c%nudged_tangential_grad = c_loc(seg%nudged_tangential_grad)
    c%Velocity_nudging_timescale_in = seg%Velocity_nudging_timescale_in
    c%Velocity_nudging_timescale_out = seg%Velocity_nudging_timescale_out
  end subroutine segment_to_c
end subroutine mom6hip_obc_to_c

!> update_segment_tracer_reservoirs(G, GV, uhr, vhr, h, OBC, dt, Reg) of MOM_open_boundary (:5373) on the GPU, with the reference's
!! argument list: MOM.F90:1447 calls this instead (MOM_open_boundary itself stays the reference's).  The reservoirs
!! segment%tr_Reg%Tr(m)%tres are updated in place, and OBC%tres_x / tres_y (the restart copies) after them on the host.
subroutine update_segment_tracer_reservoirs_hip(G, GV, uhr, vhr, h, OBC, dt, Reg)
  use MOM_tracer_registry, only : tracer_registry_type
  type(ocean_grid_type),      intent(inout) :: G
  type(verticalGrid_type),    intent(in)    :: GV
  real, dimension(G%IsdB:,G%jsd:,:), target, intent(in) :: uhr
  real, dimension(G%isd:,G%JsdB:,:), target, intent(in) :: vhr
  real, dimension(G%isd:,G%jsd:,:),  target, intent(in) :: h
  type(ocean_OBC_type),       pointer       :: OBC
  real,                       intent(in)    :: dt
  type(tracer_registry_type), pointer       :: Reg
  type(mom6hip_obc_t) :: cobc
  type(mom6hip_obc_segment_t), allocatable, target :: csegs(:)
  type(mom6hip_obc_segment_tracer_t), allocatable, target :: ctrs(:)
  type(c_ptr), allocatable :: tr(:)
  type(c_ptr) :: ctx
  integer :: n, m, i, j, k, rc, sh
  real :: I_scale
  logical :: any_reg
  if (.not.associated(OBC)) return
  if (.not.OBC%OBC_pe) return
  any_reg = .false.
  do n=1,OBC%number_of_segments ; if (associated(OBC%segment(n)%tr_Reg)) any_reg = .true. ; enddo
  if (.not.any_reg) return
  if (resident_mode) call MOM_error(FATAL, "update_segment_tracer_reservoirs (HIP): the tracer reservoirs of the open boundaries live "// &
                                    "on the host; GPU_RESIDENT_DYNAMICS is not provided for them.")
  ctx = mom6hip_shared_context(G, GV)
  call mom6hip_obc_to_c(OBC, cobc, csegs, size(uhr(:,:,1)), size(vhr(:,:,1)), "update_segment_tracer_reservoirs", ctrs)
  allocate(tr(Reg%ntr))
  do m=1,Reg%ntr ; tr(m) = c_loc(Reg%Tr(m)%t) ; enddo
  rc = mom6hip_update_segment_tracer_reservoirs(ctx, c_loc(uhr), c_loc(vhr), c_loc(h), cobc, real(dt, c_double), tr, &
                                                int(Reg%ntr, c_int32_t), MOM6HIP_MEM_HOST)
  call mom6hip_fatal_if(rc, "update_segment_tracer_reservoirs")
  ! :5455, :5495: the restart copies where the reservoirs were updated
  do n=1,OBC%number_of_segments ; if (associated(OBC%segment(n)%tr_Reg)) then
    do m=1,OBC%segment(n)%tr_Reg%ntseg ; if (allocated(OBC%segment(n)%tr_Reg%Tr(m)%tres)) then
      I_scale = 1.0 ; if (OBC%segment(n)%tr_Reg%Tr(m)%scale /= 0.0) I_scale = 1.0 / OBC%segment(n)%tr_Reg%Tr(m)%scale
      if (OBC%segment(n)%is_E_or_W .and. allocated(OBC%tres_x)) then
        i = OBC%segment(n)%HI%IsdB ; sh = 0 ; if (OBC%segment(n)%direction == OBC_DIRECTION_W) sh = 1
        do k=1,GV%ke ; do j=OBC%segment(n)%HI%jsd,OBC%segment(n)%HI%jed ; if (G%mask2dT(i+sh,j) /= 0.0) then
          OBC%tres_x(i,j,k,m) = I_scale * OBC%segment(n)%tr_Reg%Tr(m)%tres(i,j,k)
        endif ; enddo ; enddo
      elseif (OBC%segment(n)%is_N_or_S .and. allocated(OBC%tres_y)) then
        j = OBC%segment(n)%HI%JsdB ; sh = 0 ; if (OBC%segment(n)%direction == OBC_DIRECTION_S) sh = 1
        do k=1,GV%ke ; do i=OBC%segment(n)%HI%isd,OBC%segment(n)%HI%ied ; if (G%mask2dT(i,j+sh) /= 0.0) then
          OBC%tres_y(i,j,k,m) = I_scale * OBC%segment(n)%tr_Reg%Tr(m)%tres(i,j,k)
        endif ; enddo ; enddo
      endif
    endif ; enddo
  endif ; enddo
end subroutine update_segment_tracer_reservoirs_hip

end module mom6hip_MOM_glue
