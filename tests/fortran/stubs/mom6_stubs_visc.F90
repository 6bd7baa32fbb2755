!> Stand-ins the reference's MOM_vert_friction.F90 and MOM_hor_visc.F90 need beyond tests/fortran/stubs/mom6_stubs.F90 when they are compiled in
!! place beside the oracle (tests/test_reference_kernels.py): modules of diagnostics and of parameterisations that the tests never switch on.
!! Declarations with the reference's argument lists; the procedures either do nothing (diagnostics) or stop (a parameterisation that is off).
!! Nothing of this is used by the library or its shims.
#include <MOM_memory.h>

module MOM_PointAccel      ! src/diagnostics/MOM_PointAccel.F90: the truncation reports
use MOM_diag_mediator, only : diag_ctrl
use MOM_file_parser,   only : param_file_type
use MOM_get_input,     only : directories
use MOM_grid,          only : ocean_grid_type
use MOM_time_manager,  only : time_type
use MOM_unit_scaling,  only : unit_scale_type
use MOM_variables,     only : ocean_internal_state, accel_diag_ptrs, cont_diag_ptrs
use MOM_verticalGrid,  only : verticalGrid_type
implicit none ; private
public :: write_u_accel, write_v_accel, PointAccel_init, PointAccel_CS
type :: PointAccel_CS
  integer :: unused = 0
end type PointAccel_CS
contains
subroutine write_u_accel(I, j, um, hin, ADp, CDp, dt, G, GV, US, CS, vel_rpt, str, a, hv)
  integer,                     intent(in) :: I, j
  type(ocean_grid_type),       intent(in) :: G
  type(verticalGrid_type),     intent(in) :: GV
  type(unit_scale_type),       intent(in) :: US
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in) :: um
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in) :: hin
  type(accel_diag_ptrs),       intent(in) :: ADp
  type(cont_diag_ptrs),        intent(in) :: CDp
  real,                        intent(in) :: dt
  type(PointAccel_CS),         pointer    :: CS
  real,                        intent(in) :: vel_rpt
  real, optional,              intent(in) :: str
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)+1), optional, intent(in) :: a
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)),   optional, intent(in) :: hv
end subroutine write_u_accel
subroutine write_v_accel(i, J, vm, hin, ADp, CDp, dt, G, GV, US, CS, vel_rpt, str, a, hv)
  integer,                     intent(in) :: i, J
  type(ocean_grid_type),       intent(in) :: G
  type(verticalGrid_type),     intent(in) :: GV
  type(unit_scale_type),       intent(in) :: US
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in) :: vm
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in) :: hin
  type(accel_diag_ptrs),       intent(in) :: ADp
  type(cont_diag_ptrs),        intent(in) :: CDp
  real,                        intent(in) :: dt
  type(PointAccel_CS),         pointer    :: CS
  real,                        intent(in) :: vel_rpt
  real, optional,              intent(in) :: str
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)+1), optional, intent(in) :: a
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)),   optional, intent(in) :: hv
end subroutine write_v_accel
subroutine PointAccel_init(MIS, Time, G, param_file, diag, dirs, CS)
  type(ocean_internal_state), target, intent(in) :: MIS
  type(time_type),      target, intent(in)    :: Time
  type(ocean_grid_type),        intent(in)    :: G
  type(param_file_type),        intent(in)    :: param_file
  type(diag_ctrl),      target, intent(inout) :: diag
  type(directories),            intent(in)    :: dirs
  type(PointAccel_CS),          pointer       :: CS
  if (.not.associated(CS)) allocate(CS)
end subroutine PointAccel_init
end module MOM_PointAccel

module CVMix_kpp      ! pkg/CVMix-src (an empty submodule in this checkout): the KPP shape function FPMIX uses
implicit none ; private
public :: cvmix_kpp_composite_Gshape
contains
subroutine cvmix_kpp_composite_Gshape(sigma, Gat1, Gsig, dGdsig)
  real, intent(in)  :: sigma, Gat1
  real, intent(out) :: Gsig, dGdsig
  Gsig = 0.0 ; dGdsig = 0.0
  error stop "cvmix_kpp_composite_Gshape stand-in: FPMIX is not provided"
end subroutine cvmix_kpp_composite_Gshape
end module CVMix_kpp

#ifndef REF_SET_VISC
module MOM_set_visc      ! the two interpolation functions MOM_vert_friction imports for FPMIX / the Stokes drift (never called by the tests)
use MOM_grid,          only : ocean_grid_type
use MOM_open_boundary, only : ocean_OBC_type
use MOM_verticalGrid,  only : verticalGrid_type
implicit none ; private
public :: set_v_at_u, set_u_at_v
contains
function set_v_at_u(v, h, G, GV, i, j, k, mask2dCv, OBC)
  type(ocean_grid_type),   intent(in) :: G
  type(verticalGrid_type), intent(in) :: GV
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in) :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in) :: h
  integer,                 intent(in) :: i, j, k
  real, dimension(SZI_(G),SZJB_(G)), intent(in) :: mask2dCv
  type(ocean_OBC_type),    pointer    :: OBC
  real :: set_v_at_u
  set_v_at_u = 0.0
  error stop "set_v_at_u stand-in: not provided"
end function set_v_at_u
function set_u_at_v(u, h, G, GV, i, j, k, mask2dCu, OBC)
  type(ocean_grid_type),   intent(in) :: G
  type(verticalGrid_type), intent(in) :: GV
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in) :: u
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in) :: h
  integer,                 intent(in) :: i, j, k
  real, dimension(SZIB_(G),SZJ_(G)), intent(in) :: mask2dCu
  type(ocean_OBC_type),    pointer    :: OBC
  real :: set_u_at_v
  set_u_at_v = 0.0
  error stop "set_u_at_v stand-in: not provided"
end function set_u_at_v
end module MOM_set_visc
#endif

module MOM_thickness_diffuse      ! the type and the one accessor MOM_hor_visc imports (GME only)
use MOM_grid,         only : ocean_grid_type
use MOM_verticalGrid, only : verticalGrid_type
implicit none ; private
public :: thickness_diffuse_CS, thickness_diffuse_get_KH
type :: thickness_diffuse_CS
  integer :: unused = 0
end type thickness_diffuse_CS
contains
subroutine thickness_diffuse_get_KH(CS, KH_u_GME, KH_v_GME, G, GV)
  type(thickness_diffuse_CS), intent(in) :: CS
  type(ocean_grid_type),      intent(in) :: G
  type(verticalGrid_type),    intent(in) :: GV
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)+1), intent(inout) :: KH_u_GME
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)+1), intent(inout) :: KH_v_GME
  error stop "thickness_diffuse_get_KH stand-in: GME is not provided"
end subroutine thickness_diffuse_get_KH
end module MOM_thickness_diffuse
