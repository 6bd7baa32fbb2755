!> Drop-in replacement for module MOM_continuity_PPM (src/core/MOM_continuity_PPM.F90): continuity_PPM (:86),
!! continuity_PPM_init (:2679), continuity_PPM_stencil (:2763), continuity_fluxes (:200, :239) and continuity_adjust_vel
!! (:276) with the reference's dummy-argument lists, so MOM_continuity.F90 and the callers of `continuity` (the split RK2
!! step, :634, :757, :1015; MOM.F90) compile unchanged.  The work is done by libmom6hip (mom6hip_continuity) on host
!! arrays (HOST memspace: staged through the GPU for every call -- the drop-in path; a resident step uses
!! mom6hip_step_dyn_split_rk2).  The component routines the reference also exports (zonal_mass_flux, ..., used by
!! MOM_dynamics_split_RK2b and the barotropic BT_cont set-up outside the split RK2 path) exist with their reference
!! argument lists and stop with a FATAL error: they are not provided by the GPU path.
!!
!! Compiled INSIDE a MOM6 source tree in place of src/core/MOM_continuity_PPM.F90, with mom6hip_c_api.F90 and
!! mom6hip_MOM_glue.F90; here against tests/fortran/stubs (tests/test_fortran_abi.py).
module MOM_continuity_PPM

use, intrinsic :: iso_c_binding
use mom6hip_c_api
use mom6hip_MOM_glue,  only : mom6hip_shared_context, mom6hip_read_topology, mom6hip_fatal_if, mom6hip_obc_to_c
use MOM_diag_mediator, only : time_type, diag_ctrl
use MOM_error_handler, only : MOM_error, FATAL
use MOM_file_parser,   only : get_param, log_version, param_file_type
use MOM_grid,          only : ocean_grid_type
use MOM_open_boundary, only : ocean_OBC_type
use MOM_unit_scaling,  only : unit_scale_type
use MOM_variables,     only : BT_cont_type, porous_barrier_type
use MOM_verticalGrid,  only : verticalGrid_type
implicit none ; private

#include <MOM_memory.h>

public continuity_PPM, continuity_PPM_init, continuity_PPM_stencil
public continuity_fluxes, continuity_adjust_vel
public zonal_mass_flux, meridional_mass_flux
public zonal_edge_thickness, meridional_edge_thickness
public continuity_zonal_convergence, continuity_merdional_convergence
public zonal_flux_thickness, meridional_flux_thickness
public zonal_BT_mass_flux, meridional_BT_mass_flux
public set_continuity_loop_bounds
public continuity_PPM_hip_struct      ! (GPU path only) the control structure as the library's struct, for MOM_dynamics_split_RK2

!> Control structure (the reference's members, :35-71)
type, public :: continuity_PPM_CS ; private
  logical :: initialized = .false.
  type(diag_ctrl), pointer :: diag => NULL()
  logical :: upwind_1st, monotonic, simple_2nd
  real :: tol_eta, tol_vel, CFL_limit_adjust
  logical :: aggress_adjust, vol_CFL, better_iter, use_visc_rem_max, marginal_faces
end type continuity_PPM_CS

!> A container for loop bounds (:74-78)
type, public :: cont_loop_bounds_type ; private
  integer :: ish, ieh, jsh, jeh
end type cont_loop_bounds_type

interface continuity_fluxes
  module procedure continuity_3d_fluxes, continuity_2d_fluxes
end interface continuity_fluxes

contains

!> The control structure as the library's struct
function c_struct(CS) result(ccs)
  type(continuity_PPM_CS), intent(in) :: CS
  type(mom6hip_continuity_cs_t) :: ccs
  ccs%upwind_1st = merge(1, 0, CS%upwind_1st) ; ccs%monotonic = merge(1, 0, CS%monotonic)
  ccs%simple_2nd = merge(1, 0, CS%simple_2nd) ; ccs%aggress_adjust = merge(1, 0, CS%aggress_adjust)
  ccs%vol_CFL = merge(1, 0, CS%vol_CFL) ; ccs%better_iter = merge(1, 0, CS%better_iter)
  ccs%use_visc_rem_max = merge(1, 0, CS%use_visc_rem_max) ; ccs%marginal_faces = merge(1, 0, CS%marginal_faces)
  ccs%tol_eta = CS%tol_eta ; ccs%tol_vel = CS%tol_vel ; ccs%CFL_limit_adjust = CS%CFL_limit_adjust
end function c_struct

!> (GPU path only) The control structure as the library's struct: what the device-resident step of MOM_dynamics_split_RK2 hands to
!! mom6hip_step_dyn_split_rk2
function continuity_PPM_hip_struct(CS) result(ccs)
  type(continuity_PPM_CS), intent(in) :: CS
  type(mom6hip_continuity_cs_t) :: ccs
  if (.not.CS%initialized) call MOM_error(FATAL, "MOM_continuity_PPM: Module must be initialized before it is used.")
  ccs = c_struct(CS)
end function continuity_PPM_hip_struct

!> What the GPU path does not do: open boundaries and porous barriers (face fractions other than 1)
subroutine refuse_unsupported(OBC, pbv, who, obc_ok)
  type(ocean_OBC_type),      pointer    :: OBC
  type(porous_barrier_type), intent(in) :: pbv
  character(len=*),          intent(in) :: who
  logical, optional, intent(in) :: obc_ok   !< continuity_PPM itself takes an associated OBC (round 4); its other entry points do not
  logical :: ok
  ok = .false. ; if (present(obc_ok)) ok = obc_ok
  if (associated(OBC) .and. .not.ok) call MOM_error(FATAL, who//" (HIP): open boundary conditions are not supported by the GPU path.")
  if (allocated(pbv%por_face_areaU)) then
    if (any(pbv%por_face_areaU /= 1.0)) call MOM_error(FATAL, who//" (HIP): porous barriers are not supported by the GPU path.")
  endif
  if (allocated(pbv%por_face_areaV)) then
    if (any(pbv%por_face_areaV /= 1.0)) call MOM_error(FATAL, who//" (HIP): porous barriers are not supported by the GPU path.")
  endif
end subroutine refuse_unsupported

!> Same interface as the reference continuity_PPM (:86).
subroutine continuity_PPM(u, v, hin, h, uh, vh, dt, G, GV, US, CS, OBC, pbv, uhbt, vhbt, &
                          visc_rem_u, visc_rem_v, u_cor, v_cor, BT_cont, du_cor, dv_cor)
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(in)    :: u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(in)    :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  target, intent(in)    :: hin
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  target, intent(inout) :: h
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(out)   :: uh
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(out)   :: vh
  real,                    intent(in)    :: dt
  type(unit_scale_type),   intent(in)    :: US
  type(continuity_PPM_CS), intent(in)    :: CS
  type(ocean_OBC_type),    pointer       :: OBC
  type(porous_barrier_type), intent(in)  :: pbv
  real, dimension(SZIB_(G),SZJ_(G)), target, optional, intent(in) :: uhbt
  real, dimension(SZI_(G),SZJB_(G)), target, optional, intent(in) :: vhbt
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, optional, intent(in)  :: visc_rem_u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, optional, intent(in)  :: visc_rem_v
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, optional, intent(out) :: u_cor
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, optional, intent(out) :: v_cor
  type(BT_cont_type), optional, pointer  :: BT_cont
  real, dimension(SZIB_(G),SZJ_(G)), target, optional, intent(out) :: du_cor
  real, dimension(SZI_(G),SZJB_(G)), target, optional, intent(out) :: dv_cor

  type(mom6hip_bt_cont_t), target :: cbt
  type(c_ptr) :: p_uhbt, p_vhbt, p_vru, p_vrv, p_ucor, p_vcor, p_bt, p_du, p_dv
  type(mom6hip_obc_t) :: cobc
  type(mom6hip_obc_segment_t), allocatable, target :: csegs(:)
  integer :: rc

  if (.not.CS%initialized) call MOM_error(FATAL, "MOM_continuity_PPM: Module must be initialized before it is used.")
  if (present(visc_rem_u) .neqv. present(visc_rem_v)) call MOM_error(FATAL, "MOM_continuity_PPM: "//&
        "Either both visc_rem_u and visc_rem_v or neither one must be present in call to continuity_PPM.")
  call refuse_unsupported(OBC, pbv, "MOM_continuity_PPM", obc_ok=.true.)

  p_uhbt = c_null_ptr ; if (present(uhbt)) p_uhbt = c_loc(uhbt)
  p_vhbt = c_null_ptr ; if (present(vhbt)) p_vhbt = c_loc(vhbt)
  p_vru = c_null_ptr ; if (present(visc_rem_u)) p_vru = c_loc(visc_rem_u)
  p_vrv = c_null_ptr ; if (present(visc_rem_v)) p_vrv = c_loc(visc_rem_v)
  p_ucor = c_null_ptr ; if (present(u_cor)) p_ucor = c_loc(u_cor)
  p_vcor = c_null_ptr ; if (present(v_cor)) p_vcor = c_loc(v_cor)
  p_du = c_null_ptr ; if (present(du_cor)) p_du = c_loc(du_cor)
  p_dv = c_null_ptr ; if (present(dv_cor)) p_dv = c_loc(dv_cor)
  p_bt = c_null_ptr
  if (present(BT_cont)) then ; if (associated(BT_cont)) then
    cbt%FA_u_W0 = c_loc(BT_cont%FA_u_W0) ; cbt%FA_u_WW = c_loc(BT_cont%FA_u_WW)
    cbt%FA_u_E0 = c_loc(BT_cont%FA_u_E0) ; cbt%FA_u_EE = c_loc(BT_cont%FA_u_EE)
    cbt%uBT_WW = c_loc(BT_cont%uBT_WW) ; cbt%uBT_EE = c_loc(BT_cont%uBT_EE)
    cbt%FA_v_S0 = c_loc(BT_cont%FA_v_S0) ; cbt%FA_v_SS = c_loc(BT_cont%FA_v_SS)
    cbt%FA_v_N0 = c_loc(BT_cont%FA_v_N0) ; cbt%FA_v_NN = c_loc(BT_cont%FA_v_NN)
    cbt%vBT_SS = c_loc(BT_cont%vBT_SS) ; cbt%vBT_NN = c_loc(BT_cont%vBT_NN)
    cbt%h_u = c_null_ptr ; if (allocated(BT_cont%h_u)) cbt%h_u = c_loc(BT_cont%h_u)
    cbt%h_v = c_null_ptr ; if (allocated(BT_cont%h_v)) cbt%h_v = c_loc(BT_cont%h_v)
    p_bt = c_loc(cbt)
  endif ; endif

  if (associated(OBC)) then      ! what continuity_PPM reads of ocean_OBC_type and its segments (MOM_open_boundary.F90:146-386)
    call mom6hip_obc_to_c(OBC, cobc, csegs, size(uh(:,:,1)), size(vh(:,:,1)), "MOM_continuity_PPM")
    rc = mom6hip_continuity_obc(mom6hip_shared_context(G, GV), c_struct(CS), cobc, c_loc(u), c_loc(v), c_loc(hin), c_loc(h), c_loc(uh), &
                                c_loc(vh), dt, p_uhbt, p_vhbt, p_vru, p_vrv, p_ucor, p_vcor, p_bt, p_du, p_dv, MOM6HIP_MEM_HOST)
  else
  rc = mom6hip_continuity(mom6hip_shared_context(G, GV), c_struct(CS), c_loc(u), c_loc(v), c_loc(hin), c_loc(h), c_loc(uh), &
                          c_loc(vh), dt, p_uhbt, p_vhbt, p_vru, p_vrv, p_ucor, p_vcor, p_bt, p_du, p_dv, MOM6HIP_MEM_HOST)
  endif
  call mom6hip_fatal_if(rc, "MOM_continuity_PPM")
end subroutine continuity_PPM

!> Same interface as the reference continuity_3d_fluxes (:200): the transports without the thickness update
subroutine continuity_3d_fluxes(u, v, h, uh, vh, dt, G, GV, US, CS, OBC, pbv)
  type(ocean_grid_type),   intent(inout) :: G
  type(verticalGrid_type), intent(in)    :: GV
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in)  :: u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in)  :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in)  :: h
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(out) :: uh
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(out) :: vh
  real,                    intent(in)    :: dt
  type(unit_scale_type),   intent(in)    :: US
  type(continuity_PPM_CS), intent(in)    :: CS
  type(ocean_OBC_type),    pointer       :: OBC
  type(porous_barrier_type), intent(in)  :: pbv
  call MOM_error(FATAL, "MOM_continuity_PPM (HIP): continuity_fluxes is not provided by the GPU path "//&
                        "(it evaluates both directions from the same thicknesses; continuity_PPM is directionally split).")
end subroutine continuity_3d_fluxes

!> Same interface as the reference continuity_2d_fluxes (:239)
subroutine continuity_2d_fluxes(u, v, h, uhbt, vhbt, dt, G, GV, US, CS, OBC, pbv)
  type(ocean_grid_type),   intent(inout) :: G
  type(verticalGrid_type), intent(in)    :: GV
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in)  :: u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in)  :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in)  :: h
  real, dimension(SZIB_(G),SZJ_(G)),          intent(out) :: uhbt
  real, dimension(SZI_(G),SZJB_(G)),          intent(out) :: vhbt
  real,                    intent(in)    :: dt
  type(unit_scale_type),   intent(in)    :: US
  type(continuity_PPM_CS), intent(in)    :: CS
  type(ocean_OBC_type),    pointer       :: OBC
  type(porous_barrier_type), intent(in)  :: pbv
  call MOM_error(FATAL, "MOM_continuity_PPM (HIP): continuity_fluxes is not provided by the GPU path.")
end subroutine continuity_2d_fluxes

!> Same interface as the reference continuity_adjust_vel (:276)
subroutine continuity_adjust_vel(u, v, h, dt, G, GV, US, CS, OBC, pbv, uhbt, vhbt, visc_rem_u, visc_rem_v)
  type(ocean_grid_type),   intent(inout) :: G
  type(verticalGrid_type), intent(in)    :: GV
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(inout) :: u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(inout) :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in)    :: h
  real,                    intent(in)    :: dt
  type(unit_scale_type),   intent(in)    :: US
  type(continuity_PPM_CS), intent(in)    :: CS
  type(ocean_OBC_type),    pointer       :: OBC
  type(porous_barrier_type), intent(in)  :: pbv
  real, dimension(SZIB_(G),SZJ_(G)),          intent(in) :: uhbt
  real, dimension(SZI_(G),SZJB_(G)),          intent(in) :: vhbt
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), optional, intent(in) :: visc_rem_u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), optional, intent(in) :: visc_rem_v
  call MOM_error(FATAL, "MOM_continuity_PPM (HIP): continuity_adjust_vel is not provided by the GPU path.")
end subroutine continuity_adjust_vel

!> Same interface as the reference continuity_PPM_init (:2679), same parameters and defaults (:2698-2757).
subroutine continuity_PPM_init(Time, G, GV, US, param_file, diag, CS)
  type(time_type), target, intent(in)    :: Time
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(unit_scale_type),   intent(in)    :: US
  type(param_file_type),   intent(in)    :: param_file
  type(diag_ctrl), target, intent(inout) :: diag
  type(continuity_PPM_CS), intent(inout) :: CS
# include "version_variable.h"
  character(len=40)  :: mdl = "MOM_continuity_PPM"

  CS%initialized = .true.
  call log_version(param_file, mdl, version, "")
  call get_param(param_file, mdl, "MONOTONIC_CONTINUITY", CS%monotonic, &
                 "If true, CONTINUITY_PPM uses the Colella and Woodward monotonic limiter.", default=.false.)
  call get_param(param_file, mdl, "SIMPLE_2ND_PPM_CONTINUITY", CS%simple_2nd, &
                 "If true, CONTINUITY_PPM uses a simple 2nd order interpolation of the edge values.", default=.false.)
  call get_param(param_file, mdl, "UPWIND_1ST_CONTINUITY", CS%upwind_1st, &
                 "If true, CONTINUITY_PPM becomes a 1st-order upwind continuity solver.", default=.false.)
  call get_param(param_file, mdl, "ETA_TOLERANCE", CS%tol_eta, &
                 "The tolerance for the differences between the barotropic and baroclinic estimates of the sea surface height.", &
                 units="m", default=0.5*GV%ke*GV%Angstrom_H*GV%H_to_m, scale=GV%m_to_H)
  call get_param(param_file, mdl, "VELOCITY_TOLERANCE", CS%tol_vel, &
                 "The tolerance for barotropic velocity discrepancies.", units="m s-1", default=3.0e8, scale=US%m_s_to_L_T)
  call get_param(param_file, mdl, "CONT_PPM_AGGRESS_ADJUST", CS%aggress_adjust, &
                 "If true, allow the adjusted velocities to have a relative CFL change up to 0.5.", default=.false.)
  CS%vol_CFL = CS%aggress_adjust
  call get_param(param_file, mdl, "CONT_PPM_VOLUME_BASED_CFL", CS%vol_CFL, &
                 "If true, use the ratio of the open face lengths to the tracer cell areas when estimating CFL numbers.", &
                 default=CS%aggress_adjust)
  call get_param(param_file, mdl, "CONTINUITY_CFL_LIMIT", CS%CFL_limit_adjust, &
                 "The maximum CFL of the adjusted velocities.", units="nondim", default=0.5)
  call get_param(param_file, mdl, "CONT_PPM_BETTER_ITER", CS%better_iter, &
                 "If true, stop corrective iterations using a velocity based criterion.", default=.true.)
  call get_param(param_file, mdl, "CONT_PPM_USE_VISC_REM_MAX", CS%use_visc_rem_max, &
                 "If true, use more appropriate limiting bounds for corrections in strongly viscous columns.", default=.true.)
  call get_param(param_file, mdl, "CONT_PPM_MARGINAL_FACE_AREAS", CS%marginal_faces, &
                 "If true, use the marginal face areas from the continuity solver as the weights in the barotropic solver.", &
                 default=.true.)
  CS%diag => diag
  call mom6hip_read_topology(param_file)
end subroutine continuity_PPM_init

!> continuity_PPM_stencil (:2763)
function continuity_PPM_stencil(CS) result(stencil)
  type(continuity_PPM_CS), intent(in) :: CS
  integer :: stencil
  stencil = 3 ; if (CS%simple_2nd) stencil = 2 ; if (CS%upwind_1st) stencil = 1
end function continuity_PPM_stencil

!> set_continuity_loop_bounds (:2772)
function set_continuity_loop_bounds(G, CS, i_stencil, j_stencil) result(LB)
  type(ocean_grid_type),   intent(in) :: G
  type(continuity_PPM_CS), intent(in) :: CS
  logical,       optional, intent(in) :: i_stencil, j_stencil
  type(cont_loop_bounds_type)         :: LB
  integer :: stencil
  logical :: add_i, add_j
  add_i = .false. ; if (present(i_stencil)) add_i = i_stencil
  add_j = .false. ; if (present(j_stencil)) add_j = j_stencil
  stencil = continuity_PPM_stencil(CS)
  LB%ish = G%isc ; LB%ieh = G%iec ; LB%jsh = G%jsc ; LB%jeh = G%jec
  if (add_i) then ; LB%ish = G%isc - stencil ; LB%ieh = G%iec + stencil ; endif
  if (add_j) then ; LB%jsh = G%jsc - stencil ; LB%jeh = G%jec + stencil ; endif
end function set_continuity_loop_bounds

! ---- the component routines of the reference's public list: present for the callers' `use` statements, not provided ----
subroutine not_provided(name)
  character(len=*), intent(in) :: name
  call MOM_error(FATAL, "MOM_continuity_PPM (HIP): "//name//" is not provided by the GPU path; the split RK2 step "//&
                        "only calls continuity_PPM.")
end subroutine not_provided

subroutine zonal_edge_thickness(h_in, h_W, h_E, G, GV, US, CS, OBC, LB_in)
  type(ocean_grid_type),   intent(in)  :: G
  type(verticalGrid_type), intent(in)  :: GV
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), intent(in)  :: h_in
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), intent(out) :: h_W, h_E
  type(unit_scale_type),   intent(in)  :: US
  type(continuity_PPM_CS), intent(in)  :: CS
  type(ocean_OBC_type),    pointer     :: OBC
  type(cont_loop_bounds_type), optional, intent(in) :: LB_in
  call not_provided("zonal_edge_thickness")
end subroutine zonal_edge_thickness

subroutine meridional_edge_thickness(h_in, h_S, h_N, G, GV, US, CS, OBC, LB_in)
  type(ocean_grid_type),   intent(in)  :: G
  type(verticalGrid_type), intent(in)  :: GV
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), intent(in)  :: h_in
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), intent(out) :: h_S, h_N
  type(unit_scale_type),   intent(in)  :: US
  type(continuity_PPM_CS), intent(in)  :: CS
  type(ocean_OBC_type),    pointer     :: OBC
  type(cont_loop_bounds_type), optional, intent(in) :: LB_in
  call not_provided("meridional_edge_thickness")
end subroutine meridional_edge_thickness

subroutine zonal_mass_flux(u, h_in, h_W, h_E, uh, dt, G, GV, US, CS, OBC, por_face_areaU, &
                           LB_in, uhbt, visc_rem_u, u_cor, BT_cont, du_cor)
  type(ocean_grid_type),   intent(in)  :: G
  type(verticalGrid_type), intent(in)  :: GV
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in)  :: u
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in)  :: h_in, h_W, h_E
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(out) :: uh
  real,                    intent(in)  :: dt
  type(unit_scale_type),   intent(in)  :: US
  type(continuity_PPM_CS), intent(in)  :: CS
  type(ocean_OBC_type),    pointer     :: OBC
  real, dimension(SZIB_(G),SZJ_(G),SZK_(G)),  intent(in)  :: por_face_areaU
  type(cont_loop_bounds_type), optional, intent(in) :: LB_in
  real, dimension(SZIB_(G),SZJ_(G)), optional, intent(in) :: uhbt
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), optional, intent(in)  :: visc_rem_u
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), optional, intent(out) :: u_cor
  type(BT_cont_type), optional, pointer :: BT_cont
  real, dimension(SZIB_(G),SZJ_(G)), optional, intent(out) :: du_cor
  call not_provided("zonal_mass_flux")
end subroutine zonal_mass_flux

subroutine meridional_mass_flux(v, h_in, h_S, h_N, vh, dt, G, GV, US, CS, OBC, por_face_areaV, &
                                LB_in, vhbt, visc_rem_v, v_cor, BT_cont, dv_cor)
  type(ocean_grid_type),   intent(in)  :: G
  type(verticalGrid_type), intent(in)  :: GV
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in)  :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in)  :: h_in, h_S, h_N
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(out) :: vh
  real,                    intent(in)  :: dt
  type(unit_scale_type),   intent(in)  :: US
  type(continuity_PPM_CS), intent(in)  :: CS
  type(ocean_OBC_type),    pointer     :: OBC
  real, dimension(SZI_(G),SZJB_(G),SZK_(G)),  intent(in)  :: por_face_areaV
  type(cont_loop_bounds_type), optional, intent(in) :: LB_in
  real, dimension(SZI_(G),SZJB_(G)), optional, intent(in) :: vhbt
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), optional, intent(in)  :: visc_rem_v
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), optional, intent(out) :: v_cor
  type(BT_cont_type), optional, pointer :: BT_cont
  real, dimension(SZI_(G),SZJB_(G)), optional, intent(out) :: dv_cor
  call not_provided("meridional_mass_flux")
end subroutine meridional_mass_flux

subroutine zonal_BT_mass_flux(u, h_in, h_W, h_E, uhbt, dt, G, GV, US, CS, OBC, por_face_areaU, LB_in)
  type(ocean_grid_type),   intent(in)  :: G
  type(verticalGrid_type), intent(in)  :: GV
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in)  :: u
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in)  :: h_in, h_W, h_E
  real, dimension(SZIB_(G),SZJ_(G)),          intent(out) :: uhbt
  real,                    intent(in)  :: dt
  type(unit_scale_type),   intent(in)  :: US
  type(continuity_PPM_CS), intent(in)  :: CS
  type(ocean_OBC_type),    pointer     :: OBC
  real, dimension(SZIB_(G),SZJ_(G),SZK_(G)),  intent(in)  :: por_face_areaU
  type(cont_loop_bounds_type), optional, intent(in) :: LB_in
  call not_provided("zonal_BT_mass_flux")
end subroutine zonal_BT_mass_flux

subroutine meridional_BT_mass_flux(v, h_in, h_S, h_N, vhbt, dt, G, GV, US, CS, OBC, por_face_areaV, LB_in)
  type(ocean_grid_type),   intent(in)  :: G
  type(verticalGrid_type), intent(in)  :: GV
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in)  :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in)  :: h_in, h_S, h_N
  real, dimension(SZI_(G),SZJB_(G)),          intent(out) :: vhbt
  real,                    intent(in)  :: dt
  type(unit_scale_type),   intent(in)  :: US
  type(continuity_PPM_CS), intent(in)  :: CS
  type(ocean_OBC_type),    pointer     :: OBC
  real, dimension(SZI_(G),SZJB_(G),SZK_(G)),  intent(in)  :: por_face_areaV
  type(cont_loop_bounds_type), optional, intent(in) :: LB_in
  call not_provided("meridional_BT_mass_flux")
end subroutine meridional_BT_mass_flux

subroutine zonal_flux_thickness(u, h, h_W, h_E, h_u, dt, G, GV, US, LB, vol_CFL, marginal, OBC, por_face_areaU, visc_rem_u)
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in)    :: u
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in)    :: h, h_W, h_E
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(inout) :: h_u
  real,                    intent(in)    :: dt
  type(unit_scale_type),   intent(in)    :: US
  type(cont_loop_bounds_type), intent(in) :: LB
  logical,                 intent(in)    :: vol_CFL, marginal
  type(ocean_OBC_type),    pointer       :: OBC
  real, dimension(SZIB_(G),SZJ_(G),SZK_(G)),  intent(in)    :: por_face_areaU
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), optional, intent(in) :: visc_rem_u
  call not_provided("zonal_flux_thickness")
end subroutine zonal_flux_thickness

subroutine meridional_flux_thickness(v, h, h_S, h_N, h_v, dt, G, GV, US, LB, vol_CFL, marginal, OBC, por_face_areaV, visc_rem_v)
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in)    :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in)    :: h, h_S, h_N
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(inout) :: h_v
  real,                    intent(in)    :: dt
  type(cont_loop_bounds_type), intent(in) :: LB
  type(unit_scale_type),   intent(in)    :: US
  logical,                 intent(in)    :: vol_CFL, marginal
  type(ocean_OBC_type),    pointer       :: OBC
  real, dimension(SZI_(G),SZJB_(G),SZK_(G)),  intent(in)    :: por_face_areaV
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), optional, intent(in) :: visc_rem_v
  call not_provided("meridional_flux_thickness")
end subroutine meridional_flux_thickness

subroutine continuity_zonal_convergence(h, uh, dt, G, GV, LB, hin, hmin)
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(inout) :: h
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in)    :: uh
  real,                    intent(in)    :: dt
  type(cont_loop_bounds_type), intent(in) :: LB
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), optional, intent(in) :: hin
  real,                      optional, intent(in) :: hmin
  call not_provided("continuity_zonal_convergence")
end subroutine continuity_zonal_convergence

subroutine continuity_merdional_convergence(h, vh, dt, G, GV, LB, hin, hmin)
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(inout) :: h
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in)    :: vh
  real,                    intent(in)    :: dt
  type(cont_loop_bounds_type), intent(in) :: LB
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), optional, intent(in) :: hin
  real,                      optional, intent(in) :: hmin
  call not_provided("continuity_merdional_convergence")
end subroutine continuity_merdional_convergence

end module MOM_continuity_PPM
