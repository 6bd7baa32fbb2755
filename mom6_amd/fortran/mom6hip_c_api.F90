!> ISO_C_BINDING interfaces to libmom6hip (include/mom6hip.h), in the style the reference uses for
!! libc (src/framework/posix.F90:52-229): interface blocks with bind(c, name=...), `value` scalars and
!! c_int returns.  This module has no dependence on any MOM6 module; the shims that present the
!! reference's procedure signatures (MOM_tracer_advect_hip.F90, ...) are built on top of it.
module mom6hip_c_api

use, intrinsic :: iso_c_binding, only : c_int, c_int32_t, c_int64_t, c_double, c_ptr, c_char, &
                                        c_null_ptr, c_null_char, c_loc, c_associated, c_f_pointer
implicit none ; public

integer(c_int32_t), parameter :: MOM6HIP_MEM_HOST = 0, MOM6HIP_MEM_DEVICE = 1
integer(c_int32_t), parameter :: MOM6HIP_ADV_PLM = 0, MOM6HIP_ADV_PPM_H3 = 1, MOM6HIP_ADV_PPM = 2
integer(c_int32_t), parameter :: MOM6HIP_POS_H = 0, MOM6HIP_POS_U = 1, MOM6HIP_POS_V = 2, MOM6HIP_POS_Q = 3

!> mom6hip_grid_t of include/mom6hip.h
type, bind(c) :: mom6hip_grid_t
  integer(c_int32_t) :: isc, iec, jsc, jec
  integer(c_int32_t) :: isd, ied, jsd, jed
  integer(c_int32_t) :: nk
  integer(c_int32_t) :: symmetric
  integer(c_int32_t) :: reentrant_x, reentrant_y
  integer(c_int32_t) :: first_direction
  integer(c_int32_t) :: reserved0
  real(c_double) :: Angstrom_H, H_subroundoff, dZ_subroundoff, H_to_Z, Z_to_H, g_Earth, Rho0
  real(c_double) :: reserved1(8)
  type(c_ptr) :: mask2dT, areaT, IareaT, dxT, dyT, IdxT, IdyT, bathyT
  type(c_ptr) :: mask2dCu, dxCu, dyCu, dy_Cu, IdxCu, IdyCu, areaCu, IareaCu
  type(c_ptr) :: mask2dCv, dxCv, dyCv, dx_Cv, IdxCv, IdyCv, areaCv, IareaCv
  type(c_ptr) :: mask2dBu, dxBu, dyBu, areaBu, IareaBu, CoriolisBu
  type(c_ptr) :: reserved2(8)
end type mom6hip_grid_t

!> mom6hip_tracer_advect_cs_t
type, bind(c) :: mom6hip_tracer_advect_cs_t
  real(c_double) :: dt
  integer(c_int32_t) :: scheme
  integer(c_int32_t) :: use_huynh_stencil_bug
end type mom6hip_tracer_advect_cs_t

!> mom6hip_advect_stats_t
type, bind(c) :: mom6hip_advect_stats_t
  integer(c_int32_t) :: iterations, halo_updates, domore_remaining, reserved
end type mom6hip_advect_stats_t

interface
  function mom6hip_init(device) bind(c, name="mom6hip_init") result(rc)
    import :: c_int
    integer(c_int), value :: device
    integer(c_int) :: rc
  end function mom6hip_init

  function mom6hip_last_error() bind(c, name="mom6hip_last_error") result(msg)
    import :: c_ptr
    type(c_ptr) :: msg
  end function mom6hip_last_error

  function mom6hip_grid_create(grid, stream, ctx) bind(c, name="mom6hip_grid_create") result(rc)
    import :: c_int, c_ptr, mom6hip_grid_t
    type(mom6hip_grid_t), intent(in) :: grid
    type(c_ptr), value :: stream
    type(c_ptr), intent(out) :: ctx
    integer(c_int) :: rc
  end function mom6hip_grid_create

  function mom6hip_grid_destroy(ctx) bind(c, name="mom6hip_grid_destroy") result(rc)
    import :: c_int, c_ptr
    type(c_ptr), value :: ctx
    integer(c_int) :: rc
  end function mom6hip_grid_destroy

  function mom6hip_sync(ctx) bind(c, name="mom6hip_sync") result(rc)
    import :: c_int, c_ptr
    type(c_ptr), value :: ctx
    integer(c_int) :: rc
  end function mom6hip_sync

  function mom6hip_advect_tracer(ctx, h_end, uhtr, vhtr, dt, cs, tr, conc_underflow, ntr, x_first_in, &
                                 vol_prev, max_iter_in, update_vol_prev, uhr_out, vhr_out, memspace, stats) &
                                 bind(c, name="mom6hip_advect_tracer") result(rc)
    import :: c_int, c_int32_t, c_double, c_ptr, mom6hip_tracer_advect_cs_t, mom6hip_advect_stats_t
    type(c_ptr), value :: ctx
    type(c_ptr), value :: h_end, uhtr, vhtr
    real(c_double), value :: dt
    type(mom6hip_tracer_advect_cs_t), intent(in) :: cs
    type(c_ptr), intent(in) :: tr(*)          !< c_loc of each Reg%Tr(m)%t
    type(c_ptr), value :: conc_underflow      !< c_loc of a real(c_double) array, or c_null_ptr
    integer(c_int32_t), value :: ntr, x_first_in
    type(c_ptr), value :: vol_prev
    integer(c_int32_t), value :: max_iter_in, update_vol_prev
    type(c_ptr), value :: uhr_out, vhr_out
    integer(c_int32_t), value :: memspace
    type(mom6hip_advect_stats_t), intent(out) :: stats
    integer(c_int) :: rc
  end function mom6hip_advect_tracer
end interface

contains

!> The library's last error message as a Fortran string.
function mom6hip_error_string() result(str)
  character(len=:), allocatable :: str
  character(kind=c_char), pointer :: chars(:)
  type(c_ptr) :: p
  integer :: n
  p = mom6hip_last_error()
  str = ""
  if (.not. c_associated(p)) return
  call c_f_pointer(p, chars, [1024])
  n = 0
  do while (n < 1024)
    if (chars(n+1) == c_null_char) exit
    n = n + 1
  enddo
  allocate(character(len=n) :: str)
  if (n > 0) str = transfer(chars(1:n), str)
end function mom6hip_error_string

end module mom6hip_c_api
