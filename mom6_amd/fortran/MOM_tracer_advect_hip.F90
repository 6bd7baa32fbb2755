!> Drop-in replacement for module MOM_tracer_advect (src/tracer/MOM_tracer_advect.F90): the same
!! public procedures with the same dummy-argument lists, so src/core/MOM.F90 (:1438),
!! src/tracer/MOM_offline_main.F90 and the other callers compile unchanged; the work is done by
!! libmom6hip's HIP kernels through mom6hip_c_api.
!!
!! Build note: this file is compiled INSIDE a MOM6 source tree in place of
!! src/tracer/MOM_tracer_advect.F90 (it uses the real MOM_grid, MOM_tracer_registry, ... modules), together with
!! mom6hip_c_api.F90 and mom6hip_MOM_glue.F90, and the executable is linked with -lmom6hip.  In this repository it is
!! compiled against the type-only stand-ins of tests/fortran/stubs (tests/test_fortran_abi.py).
module MOM_tracer_advect

use, intrinsic :: iso_c_binding
use mom6hip_c_api
use mom6hip_MOM_glue,    only : mom6hip_context_create, mom6hip_read_topology
use mom6hip_MOM_glue,     only : mom6hip_read_resident, mom6hip_resident, mom6hip_mirror, mom6hip_obc_to_c
use MOM_cpu_clock,       only : cpu_clock_id, cpu_clock_begin, cpu_clock_end, CLOCK_MODULE
use MOM_diag_mediator,   only : diag_ctrl, time_type
use MOM_error_handler,   only : MOM_error, FATAL, WARNING
use MOM_file_parser,     only : get_param, log_version, param_file_type
use MOM_grid,            only : ocean_grid_type
use MOM_open_boundary,   only : ocean_OBC_type
use MOM_tracer_registry, only : tracer_registry_type
use MOM_unit_scaling,    only : unit_scale_type
use MOM_verticalGrid,    only : verticalGrid_type
implicit none ; private

#include <MOM_memory.h>

public advect_tracer
public tracer_advect_init
public tracer_advect_end

!> Control structure for this module (same role as the reference's tracer_advect_CS, :30-40)
type, public :: tracer_advect_CS ; private
  real    :: dt                    !< The baroclinic dynamics time step [T ~> s].
  type(diag_ctrl), pointer :: diag => NULL()
  logical :: debug
  logical :: usePPM
  logical :: useHuynh
  logical :: useHuynhStencilBug = .false.
  logical :: reentrant(2) = .false. !< REENTRANT_X, REENTRANT_Y (for the library's own wrap on a one-tile domain)
  type(c_ptr) :: ctx = c_null_ptr  !< mom6hip_ctx_t: metrics and work space resident on the GPU
end type tracer_advect_CS

integer :: id_clock_advect

contains

!> Same interface as the reference advect_tracer (src/tracer/MOM_tracer_advect.F90:52).
subroutine advect_tracer(h_end, uhtr, vhtr, OBC, dt, G, GV, US, CS, Reg, x_first_in, &
                         vol_prev, max_iter_in, update_vol_prev, uhr_out, vhr_out)
  type(ocean_grid_type),   intent(inout) :: G
  type(verticalGrid_type), intent(in)    :: GV
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  target, intent(in) :: h_end
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(in) :: uhtr
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(in) :: vhtr
  type(ocean_OBC_type),    pointer       :: OBC
  real,                    intent(in)    :: dt
  type(unit_scale_type),   intent(in)    :: US
  type(tracer_advect_CS),  pointer       :: CS
  type(tracer_registry_type), pointer    :: Reg
  logical,       optional, intent(in)    :: x_first_in
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  target, optional, intent(inout) :: vol_prev
  integer,       optional, intent(in)    :: max_iter_in
  logical,       optional, intent(in)    :: update_vol_prev
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, optional, intent(out) :: uhr_out
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, optional, intent(out) :: vhr_out

  type(c_ptr), allocatable :: tr(:)
  real(c_double), allocatable, target :: cu(:)
  type(mom6hip_tracer_advect_cs_t) :: ccs
  type(mom6hip_advect_stats_t) :: stats
  type(c_ptr) :: p_vol, p_uhr, p_vhr
  integer(c_int32_t) :: xf, mi, uv
  integer :: m, rc
  type(mom6hip_obc_t), target :: cobc
  type(mom6hip_obc_segment_t), allocatable, target :: csegs(:)
  type(mom6hip_obc_segment_tracer_t), allocatable, target :: ctrs(:)
  type(c_ptr) :: p_obc
  logical :: registries

  if (.not. associated(CS)) call MOM_error(FATAL, "MOM_tracer_advect: "// &
       "tracer_advect_init must be called before advect_tracer.")
  if (.not. associated(Reg)) call MOM_error(FATAL, "MOM_tracer_advect: "// &
       "register_tracer must be called before advect_tracer.")
  if (Reg%ntr==0) return
  ! advect_x / advect_y read of an associated OBC only the tracer registries of its segments (segment%tr_Reg: the reservoirs and inflow
  ! concentrations, :442-477, :580-627 and their twins): without one on any segment the advection is that of a closed domain
  p_obc = c_null_ptr ; registries = .false.
  if (associated(OBC)) then ; if (OBC%OBC_pe) then
    do m=1,OBC%number_of_segments
      if (associated(OBC%segment(m)%tr_Reg)) registries = .true.
    enddo
    if (registries) then
      if (mom6hip_resident()) call MOM_error(FATAL, "MOM_tracer_advect (HIP): open boundary segments with a tracer registry are not "// &
           "provided with GPU_RESIDENT_DYNAMICS (their reservoirs live on the host).")
      call mom6hip_obc_to_c(OBC, cobc, csegs, size(uhtr(:,:,1)), size(vhtr(:,:,1)), "MOM_tracer_advect", ctrs)
      p_obc = c_loc(cobc)
    endif
  endif ; endif
  call cpu_clock_begin(id_clock_advect)

  ! the context: metrics on the GPU, the device of this PE, and MOM6's pass_var / sum_across_PEs behind the group pass
  ! of every iteration (do_group_pass(CS%pass_uhr_vhr_t_hprev), :205-206) whenever the tile has neighbours
  if (.not. c_associated(CS%ctx)) call mom6hip_context_create(G, GV, CS%ctx, CS%reentrant)

  allocate(tr(Reg%ntr), cu(Reg%ntr))
  do m=1,Reg%ntr
    if (associated(Reg%Tr(m)%ad_x) .or. associated(Reg%Tr(m)%ad_y) .or. &
        associated(Reg%Tr(m)%advection_xy) .or. associated(Reg%Tr(m)%ad2d_x) .or. &
        associated(Reg%Tr(m)%ad2d_y)) call MOM_error(FATAL, "MOM_tracer_advect (HIP): "// &
        "advective flux diagnostics are not provided by the GPU tracer advection.")
    tr(m) = c_loc(Reg%Tr(m)%t)
    cu(m) = Reg%Tr(m)%conc_underflow
  enddo

  ccs%dt = CS%dt ; ccs%use_huynh_stencil_bug = merge(1, 0, CS%useHuynhStencilBug)
  ccs%scheme = MOM6HIP_ADV_PLM
  if (CS%usePPM .and. CS%useHuynh) ccs%scheme = MOM6HIP_ADV_PPM_H3
  if (CS%usePPM .and. .not.CS%useHuynh) ccs%scheme = MOM6HIP_ADV_PPM

  xf = -1 ; if (present(x_first_in)) xf = merge(1, 0, x_first_in)
  mi = 0 ; if (present(max_iter_in)) mi = max_iter_in
  uv = 0 ; if (present(update_vol_prev)) uv = merge(1, 0, update_vol_prev)
  p_vol = c_null_ptr ; if (present(vol_prev)) p_vol = c_loc(vol_prev)
  p_uhr = c_null_ptr ; if (present(uhr_out)) p_uhr = c_loc(uhr_out)
  p_vhr = c_null_ptr ; if (present(vhr_out)) p_vhr = c_loc(vhr_out)

  if (mom6hip_resident()) then      ! GPU_RESIDENT_DYNAMICS: the shared device mirrors of the host arrays
    do m=1,Reg%ntr ; tr(m) = mom6hip_mirror(CS%ctx, tr(m), int(size(h_end), c_int64_t), .true., .true.) ; enddo
    if (present(vol_prev)) p_vol = mom6hip_mirror(CS%ctx, p_vol, int(size(h_end), c_int64_t), .true., .true.)
    if (present(uhr_out)) p_uhr = mom6hip_mirror(CS%ctx, p_uhr, int(size(uhtr), c_int64_t), .false., .true.)
    if (present(vhr_out)) p_vhr = mom6hip_mirror(CS%ctx, p_vhr, int(size(vhtr), c_int64_t), .false., .true.)
    rc = mom6hip_advect_tracer(CS%ctx, mom6hip_mirror(CS%ctx, c_loc(h_end), int(size(h_end), c_int64_t), .true., .false.), &
                               mom6hip_mirror(CS%ctx, c_loc(uhtr), int(size(uhtr), c_int64_t), .true., .false.), &
                               mom6hip_mirror(CS%ctx, c_loc(vhtr), int(size(vhtr), c_int64_t), .true., .false.), dt, ccs, tr, c_loc(cu), &
                               int(Reg%ntr, c_int32_t), xf, p_vol, mi, uv, p_uhr, p_vhr, MOM6HIP_MEM_DEVICE, stats)
  else
    rc = mom6hip_advect_tracer_obc(CS%ctx, c_loc(h_end), c_loc(uhtr), c_loc(vhtr), dt, ccs, tr, c_loc(cu), &
                                   int(Reg%ntr, c_int32_t), xf, p_vol, mi, uv, p_uhr, p_vhr, p_obc, &
                                   MOM6HIP_MEM_HOST, stats)
  endif
  if (rc /= 0) call MOM_error(FATAL, "MOM_tracer_advect (HIP): "//mom6hip_error_string())

  call cpu_clock_end(id_clock_advect)
end subroutine advect_tracer

!> Same parameters as the reference tracer_advect_init (:1090-1150).
subroutine tracer_advect_init(Time, G, US, param_file, diag, CS)
  type(time_type), target, intent(in)    :: Time
  type(ocean_grid_type),   intent(in)    :: G
  type(unit_scale_type),   intent(in)    :: US
  type(param_file_type),   intent(in)    :: param_file
  type(diag_ctrl), target, intent(inout) :: diag
  type(tracer_advect_CS),  pointer       :: CS
# include "version_variable.h"
  character(len=40)  :: mdl = "MOM_tracer_advect"
  character(len=256) :: mesg

  if (associated(CS)) then
    call MOM_error(WARNING, "tracer_advect_init called with associated control structure.")
    return
  endif
  allocate(CS)
  CS%diag => diag
  call log_version(param_file, mdl, version, "")
  call get_param(param_file, mdl, "DT", CS%dt, fail_if_missing=.true., &
          desc="The (baroclinic) dynamics time step.", units="s", scale=US%s_to_T)
  call get_param(param_file, mdl, "DEBUG", CS%debug, default=.false.)
  call get_param(param_file, mdl, "TRACER_ADVECTION_SCHEME", mesg, &
          desc="The horizontal transport scheme for tracers:\n"//&
          "  PLM    - Piecewise Linear Method\n"//&
          "  PPM:H3 - Piecewise Parabolic Method (Huyhn 3rd order)\n"// &
          "  PPM    - Piecewise Parabolic Method (Colella-Woodward)" &
          , default='PLM')
  select case (trim(mesg))
    case ("PLM")
      CS%usePPM = .false. ; CS%useHuynh = .false.
    case ("PPM:H3")
      CS%usePPM = .true. ; CS%useHuynh = .true.
    case ("PPM")
      CS%usePPM = .true. ; CS%useHuynh = .false.
    case default
      call MOM_error(FATAL, "MOM_tracer_advect, tracer_advect_init: "//&
           "Unknown TRACER_ADVECTION_SCHEME = "//trim(mesg))
  end select
  if (CS%useHuynh) then
    call get_param(param_file, mdl, "USE_HUYNH_STENCIL_BUG", CS%useHuynhStencilBug, &
        desc="If true, use a stencil width of 2 in PPM:H3 tracer advection.", default=.false.)
  endif
  call mom6hip_read_topology(param_file, CS%reentrant)
  call mom6hip_read_resident(param_file)
  id_clock_advect = cpu_clock_id('(Ocean advect tracer)', grain=CLOCK_MODULE)
end subroutine tracer_advect_init

!> Close the tracer advection module and release the GPU context.
subroutine tracer_advect_end(CS)
  type(tracer_advect_CS), pointer :: CS
  integer :: rc
  if (associated(CS)) then
    if (c_associated(CS%ctx)) rc = mom6hip_grid_destroy(CS%ctx)
    deallocate(CS)
  endif
end subroutine tracer_advect_end

end module MOM_tracer_advect
