!> Drop-in replacement for module MOM_vert_friction (src/parameterizations/vertical/MOM_vert_friction.F90): vertvisc (:526),
!! vertvisc_remnant (:1064), vertvisc_coef (:1168), vertvisc_init (:2465), vertvisc_end, updateCFLtruncationValue with the
!! reference's dummy-argument lists, so the split RK2 step (:598-600, :717-744, :974-994) compiles unchanged.  The work is
!! done by libmom6hip (mom6hip_vertvisc_coef / _vertvisc / _vertvisc_remnant, HOST memspace).  Provided: BOTTOMDRAGLAW /
!! KV_EXTRA_BBL, HARMONIC_VISC, HARMONIC_BL_SCALE, KV_ML_INVZ2 + HMIX_FIXED, visc%Kv_shear, visc%Ray_u/v, DIRECT_STRESS,
!! CFL-based / MAXVEL truncation with its count.  DYNAMIC_VISCOUS_ML, a bulk mixed layer, FIXED_DEPTH_LOTW_ML,
!! LOTW_VISCOUS_ML_FLOOR, GL90, Stokes mixing / FPMIX, open boundaries, the truncation files, a CFL truncation ramp and the
!! acceleration diagnostics stop with a FATAL error.
!!
!! Compiled INSIDE a MOM6 source tree in place of the reference file; here against tests/fortran/stubs.
module MOM_vert_friction

use, intrinsic :: iso_c_binding
use mom6hip_c_api
use mom6hip_MOM_glue,          only : mom6hip_shared_context, mom6hip_read_topology, mom6hip_fatal_if, mom6hip_obc_to_c
use MOM_diag_mediator,         only : diag_ctrl, time_type
use MOM_error_handler,         only : MOM_error, FATAL, WARNING
use MOM_file_parser,           only : get_param, log_version, param_file_type
use MOM_forcing_type,          only : mech_forcing
use MOM_grid,                  only : ocean_grid_type
use MOM_get_input,             only : directories
use MOM_lateral_mixing_coeffs, only : VarMix_CS
use MOM_open_boundary,         only : ocean_OBC_type
use MOM_unit_scaling,          only : unit_scale_type
use MOM_variables,             only : thermo_var_ptrs, vertvisc_type, ocean_internal_state, accel_diag_ptrs, cont_diag_ptrs
use MOM_verticalGrid,          only : verticalGrid_type
use MOM_wave_interface,        only : wave_parameters_CS
implicit none ; private

#include <MOM_memory.h>

public vertvisc, vertvisc_remnant, vertvisc_coef
public vertvisc_limit_vel, vertvisc_init, vertvisc_end
public updateCFLtruncationValue
public vertFPmix
public vertvisc_hip_struct, vertvisc_hip_add_ntrunc      ! (GPU path only) for MOM_dynamics_split_RK2

!> The control structure: the library's struct and the arrays vertvisc_coef leaves for vertvisc / vertvisc_remnant
type, public :: vertvisc_CS ; private
  logical :: initialized = .false.
  type(mom6hip_vertvisc_cs_t) :: st
  real, allocatable, dimension(:,:,:) :: a_u, a_v   !< coupling coefficients across interfaces [H T-1 ~> m s-1]
  real, allocatable, dimension(:,:,:) :: h_u, h_v   !< thicknesses at velocity points [H ~> m]
  integer, pointer :: ntrunc => NULL()              !< the model's count of velocity truncations
  type(diag_ctrl), pointer :: diag => NULL()
end type vertvisc_CS

contains

subroutine bind_arrays(CS)
  type(vertvisc_CS), target, intent(inout) :: CS
  CS%st%a_u = c_loc(CS%a_u) ; CS%st%a_v = c_loc(CS%a_v) ; CS%st%h_u = c_loc(CS%h_u) ; CS%st%h_v = c_loc(CS%h_v)
  CS%st%reserved1(:) = c_null_ptr
end subroutine bind_arrays

!> (GPU path only) The library's struct of this control structure (pointers bound to the HOST arrays of CS)
function vertvisc_hip_struct(CS) result(st)
  type(vertvisc_CS), pointer :: CS
  type(mom6hip_vertvisc_cs_t) :: st
  if (.not.associated(CS)) call MOM_error(FATAL, "MOM_vert_friction(visc): Module must be initialized before it is used.")
  call bind_arrays(CS)
  st = CS%st
end function vertvisc_hip_struct

!> (GPU path only) Velocity truncations counted by a device-resident step
subroutine vertvisc_hip_add_ntrunc(CS, n)
  type(vertvisc_CS), pointer :: CS
  integer(c_int64_t), intent(in) :: n
  CS%st%ntrunc = CS%st%ntrunc + n
  if (associated(CS%ntrunc)) CS%ntrunc = CS%ntrunc + int(n)
end subroutine vertvisc_hip_add_ntrunc

!> vertvisc_type as the library's struct of pointers (members that are not allocated / associated travel as null)
subroutine visc_struct(visc, cv)
  type(vertvisc_type), target, intent(in)  :: visc
  type(mom6hip_vertvisc_type_t),  intent(out) :: cv
  cv%Kv_bbl_u = c_null_ptr ; if (allocated(visc%Kv_bbl_u)) cv%Kv_bbl_u = c_loc(visc%Kv_bbl_u)
  cv%Kv_bbl_v = c_null_ptr ; if (allocated(visc%Kv_bbl_v)) cv%Kv_bbl_v = c_loc(visc%Kv_bbl_v)
  cv%bbl_thick_u = c_null_ptr ; if (allocated(visc%bbl_thick_u)) cv%bbl_thick_u = c_loc(visc%bbl_thick_u)
  cv%bbl_thick_v = c_null_ptr ; if (allocated(visc%bbl_thick_v)) cv%bbl_thick_v = c_loc(visc%bbl_thick_v)
  cv%Ray_u = c_null_ptr ; if (allocated(visc%Ray_u)) cv%Ray_u = c_loc(visc%Ray_u)
  cv%Ray_v = c_null_ptr ; if (allocated(visc%Ray_v)) cv%Ray_v = c_loc(visc%Ray_v)
  cv%Kv_shear = c_null_ptr ; if (associated(visc%Kv_shear)) cv%Kv_shear = c_loc(visc%Kv_shear)
  cv%Kv_shear_Bu = c_null_ptr ; if (associated(visc%Kv_shear_Bu)) cv%Kv_shear_Bu = c_loc(visc%Kv_shear_Bu)
  cv%nkml_visc_u = c_null_ptr ; if (allocated(visc%nkml_visc_u)) cv%nkml_visc_u = c_loc(visc%nkml_visc_u)
  cv%nkml_visc_v = c_null_ptr ; if (allocated(visc%nkml_visc_v)) cv%nkml_visc_v = c_loc(visc%nkml_visc_v)
  cv%ustar = c_null_ptr      ! (forces%ustar: set by vertvisc_coef)
  cv%reserved(:) = c_null_ptr
end subroutine visc_struct

!> Same interface as the reference vertvisc (:526).
subroutine vertvisc(u, v, h, forces, visc, dt, OBC, ADp, CDp, G, GV, US, CS, taux_bot, tauy_bot, fpmix, Waves)
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(unit_scale_type),   intent(in)    :: US
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(inout) :: u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(inout) :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  target, intent(in)    :: h
  type(mech_forcing),    intent(in)      :: forces
  type(vertvisc_type),   intent(inout)   :: visc
  real,                  intent(in)      :: dt
  type(ocean_OBC_type),  pointer         :: OBC
  type(accel_diag_ptrs), intent(inout)   :: ADp
  type(cont_diag_ptrs),  intent(inout)   :: CDp
  type(vertvisc_CS),     pointer         :: CS
  real, dimension(SZIB_(G),SZJ_(G)), target, optional, intent(out) :: taux_bot
  real, dimension(SZI_(G),SZJB_(G)), target, optional, intent(out) :: tauy_bot
  logical,         optional, intent(in)  :: fpmix
  type(wave_parameters_CS), optional, pointer :: Waves
  type(mom6hip_vertvisc_type_t) :: cv
  type(mom6hip_obc_t) :: cobc
  type(mom6hip_obc_segment_t), allocatable, target :: csegs(:)
  type(c_ptr) :: p_tx, p_ty
  integer :: rc
  if (.not.associated(CS)) call MOM_error(FATAL, "MOM_vert_friction(visc): Module must be initialized before it is used.")
  if (.not.CS%initialized) call MOM_error(FATAL, "MOM_vert_friction(visc): Module must be initialized before it is used.")
  if (present(fpmix)) then ; if (fpmix) call MOM_error(FATAL, "vertvisc (HIP): FPMIX is not provided by the GPU path.") ; endif
  if (present(Waves)) then ; if (associated(Waves)) &
    call MOM_error(FATAL, "vertvisc (HIP): Stokes mixing (Waves) is not provided by the GPU path.") ; endif
  if (.not.(associated(forces%taux) .and. associated(forces%tauy))) &
    call MOM_error(FATAL, "vertvisc (HIP): forces%taux and forces%tauy must be associated.")
  call bind_arrays(CS) ; call visc_struct(visc, cv)
  p_tx = c_null_ptr ; if (present(taux_bot)) p_tx = c_loc(taux_bot)
  p_ty = c_null_ptr ; if (present(tauy_bot)) p_ty = c_loc(tauy_bot)
  if (associated(OBC)) then      ! the velocities of the specified segments end the call (:988-1006)
    call mom6hip_obc_to_c(OBC, cobc, csegs, size(u(:,:,1)), size(v(:,:,1)), "MOM_vert_friction")
    rc = mom6hip_vertvisc_obc(mom6hip_shared_context(G, GV), CS%st, c_loc(u), c_loc(v), c_loc(h), c_loc(forces%taux), &
                              c_loc(forces%tauy), cv, dt, p_tx, p_ty, cobc, MOM6HIP_MEM_HOST)
  else
    rc = mom6hip_vertvisc(mom6hip_shared_context(G, GV), CS%st, c_loc(u), c_loc(v), c_loc(h), c_loc(forces%taux), c_loc(forces%tauy), &
                          cv, dt, p_tx, p_ty, MOM6HIP_MEM_HOST)
  endif
  call mom6hip_fatal_if(rc, "vertvisc")
  if (associated(CS%ntrunc)) CS%ntrunc = int(CS%st%ntrunc)
end subroutine vertvisc

!> Same interface as the reference vertvisc_remnant (:1064).
subroutine vertvisc_remnant(visc, visc_rem_u, visc_rem_v, dt, G, GV, US, CS)
  type(ocean_grid_type), intent(in)   :: G
  type(verticalGrid_type), intent(in) :: GV
  type(vertvisc_type),   intent(in)   :: visc
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(inout) :: visc_rem_u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(inout) :: visc_rem_v
  real,                  intent(in)    :: dt
  type(unit_scale_type), intent(in)    :: US
  type(vertvisc_CS),     pointer       :: CS
  type(mom6hip_vertvisc_type_t) :: cv
  integer :: rc
  if (.not.associated(CS)) call MOM_error(FATAL, "MOM_vert_friction(remant): Module must be initialized before it is used.")
  if (.not.CS%initialized) call MOM_error(FATAL, "MOM_vert_friction(remnant): Module must be initialized before it is used.")
  call bind_arrays(CS) ; call visc_struct(visc, cv)
  rc = mom6hip_vertvisc_remnant(mom6hip_shared_context(G, GV), CS%st, cv, c_loc(visc_rem_u), c_loc(visc_rem_v), dt, MOM6HIP_MEM_HOST)
  call mom6hip_fatal_if(rc, "vertvisc_remnant")
end subroutine vertvisc_remnant

!> Same interface as the reference vertvisc_coef (:1168).
subroutine vertvisc_coef(u, v, h, dz, forces, visc, tv, dt, G, GV, US, CS, OBC, VarMix)
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(unit_scale_type),   intent(in)    :: US
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(in) :: u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(in) :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  target, intent(in) :: h
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  target, intent(in) :: dz
  type(mech_forcing),      intent(in)    :: forces
  type(vertvisc_type),     intent(in)    :: visc
  type(thermo_var_ptrs),   intent(in)    :: tv
  real,                    intent(in)    :: dt
  type(vertvisc_CS),       pointer       :: CS
  type(ocean_OBC_type),    pointer       :: OBC
  type(VarMix_CS),         intent(in)    :: VarMix
  type(mom6hip_vertvisc_type_t) :: cv
  type(mom6hip_obc_t) :: cobc
  type(mom6hip_obc_segment_t), allocatable, target :: csegs(:)
  integer :: rc
  if (.not.associated(CS)) call MOM_error(FATAL, "MOM_vert_friction(coef): Module must be initialized before it is used.")
  if (.not.CS%initialized) call MOM_error(FATAL, "MOM_vert_friction(coef): Module must be initialized before it is used.")
  call bind_arrays(CS) ; call visc_struct(visc, cv)
  if (CS%st%dynamic_viscous_ML /= 0 .or. CS%st%nkml > 0) then      ! find_ustar(forces, tv, Ustar_2d, ...) :1296, Boussinesq
    if (.not.associated(forces%ustar)) call MOM_error(FATAL, "vertvisc_coef (HIP): DYNAMIC_VISCOUS_ML / a bulk mixed layer needs "// &
         "forces%ustar (the GPU path is Boussinesq: find_ustar returns forces%ustar).")
    cv%ustar = c_loc(forces%ustar)
  endif
  if (associated(OBC)) then      ! the zero-gradient projections across the segments' faces (:1335-1355, :1546-1566, :1901-1925, :2061-2110)
    call mom6hip_obc_to_c(OBC, cobc, csegs, size(u(:,:,1)), size(v(:,:,1)), "MOM_vert_friction")
    rc = mom6hip_vertvisc_coef_obc(mom6hip_shared_context(G, GV), CS%st, c_loc(u), c_loc(v), c_loc(h), c_loc(dz), cv, dt, cobc, &
                                   MOM6HIP_MEM_HOST)
  else
    rc = mom6hip_vertvisc_coef(mom6hip_shared_context(G, GV), CS%st, c_loc(u), c_loc(v), c_loc(h), c_loc(dz), cv, dt, MOM6HIP_MEM_HOST)
  endif
  call mom6hip_fatal_if(rc, "vertvisc_coef")
end subroutine vertvisc_coef

!> Same interface as the reference vertvisc_limit_vel (:2259): the library applies it inside vertvisc
subroutine vertvisc_limit_vel(u, v, h, ADp, CDp, forces, visc, dt, G, GV, US, CS)
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(unit_scale_type),   intent(in)    :: US
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(inout) :: u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(inout) :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in)    :: h
  type(accel_diag_ptrs),   intent(in)    :: ADp
  type(cont_diag_ptrs),    intent(in)    :: CDp
  type(mech_forcing),      intent(in)    :: forces
  type(vertvisc_type),     intent(in)    :: visc
  real,                    intent(in)    :: dt
  type(vertvisc_CS),       pointer       :: CS
  call MOM_error(FATAL, "vertvisc_limit_vel (HIP): not provided as a separate call; vertvisc applies it on the GPU.")
end subroutine vertvisc_limit_vel

!> Same interface as the reference vertFPmix: not provided
subroutine vertFPmix(ui, vi, uold, vold, hbl_h, h, forces, dt, lpost, Cemp_NL, G, GV, US, CS, OBC, Waves)
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(inout) :: ui, uold
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(inout) :: vi, vold
  real, dimension(SZI_(G),SZJ_(G)),           intent(inout) :: hbl_h
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in)    :: h
  type(mech_forcing),      intent(in)    :: forces
  real,                    intent(in)    :: dt
  logical,                 intent(in)    :: lpost
  real,                    intent(in)    :: Cemp_NL
  type(unit_scale_type),   intent(in)    :: US
  type(vertvisc_CS),       pointer       :: CS
  type(ocean_OBC_type),    pointer       :: OBC
  type(wave_parameters_CS), optional, pointer :: Waves
  call MOM_error(FATAL, "vertFPmix (HIP): FPMIX is not provided by the GPU path.")
end subroutine vertFPmix

!> Same interface as the reference vertvisc_init (:2465), same parameters and defaults.
subroutine vertvisc_init(MIS, Time, G, GV, US, param_file, diag, ADp, dirs, ntrunc, CS, fpmix)
  type(ocean_internal_state), target, intent(in) :: MIS
  type(time_type), target, intent(in)    :: Time
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(unit_scale_type),   intent(in)    :: US
  type(param_file_type),   intent(in)    :: param_file
  type(diag_ctrl), target, intent(inout) :: diag
  type(accel_diag_ptrs),   intent(inout) :: ADp
  type(directories),       intent(in)    :: dirs
  integer, target,         intent(inout) :: ntrunc
  type(vertvisc_CS),       pointer       :: CS
  logical, optional,       intent(in)    :: fpmix
# include "version_variable.h"
  character(len=40)  :: mdl = "MOM_vert_friction"
  logical :: flag, bulkmixedlayer
  real :: Hmix_m, Hmix_stress_m, val
  integer :: isd, ied, jsd, jed, nz, nkml, default_answer_date, answer_date

  if (associated(CS)) then
    call MOM_error(WARNING, "vertvisc_init called with an associated control structure.")
    return
  endif
  allocate(CS)
  CS%initialized = .true. ; CS%diag => diag ; CS%ntrunc => ntrunc ; ntrunc = 0
  isd = G%isd ; ied = G%ied ; jsd = G%jsd ; jed = G%jed ; nz = GV%ke
  if (present(fpmix)) then ; if (fpmix) call refuse(.true., "FPMIX") ; endif
  if (.not.GV%Boussinesq) call refuse(.true., "a non-Boussinesq vertical grid")
  CS%st%unsupported(:) = 0 ; CS%st%reserved0(:) = 0.0 ; CS%st%ntrunc = 0
  call log_version(param_file, mdl, version, "")
  call get_param(param_file, mdl, "DEFAULT_ANSWER_DATE", default_answer_date, default=99991231)
  call get_param(param_file, mdl, "VERT_FRICTION_ANSWER_DATE", answer_date, default=default_answer_date)
  CS%st%answer_date = answer_date
  call get_param(param_file, mdl, "BOTTOMDRAGLAW", flag, &
                 "If true, the bottom stress is calculated with a drag law of the form c_drag*|u|*u.", default=.true.)
  CS%st%bottomdraglaw = merge(1, 0, flag)
  call get_param(param_file, mdl, "DIRECT_STRESS", flag, &
                 "If true, the wind stress is distributed over the topmost HMIX_STRESS of fluid.", default=.false.)
  CS%st%direct_stress = merge(1, 0, flag)
  call get_param(param_file, mdl, "DYNAMIC_VISCOUS_ML", flag, &
                 "If true, use a bulk Richardson number criterion to determine the mixed layer thickness for viscosity.", default=.false.)
  CS%st%dynamic_viscous_ML = merge(1, 0, flag)
  call get_param(param_file, mdl, "FIXED_DEPTH_LOTW_ML", flag, default=.false.) ; call refuse(flag, "FIXED_DEPTH_LOTW_ML")
  call get_param(param_file, mdl, "LOTW_VISCOUS_ML_FLOOR", flag, default=.false.) ; call refuse(flag, "LOTW_VISCOUS_ML_FLOOR")
  call get_param(param_file, mdl, "USE_GL90_IN_SSW", flag, default=.false.) ; call refuse(flag, "USE_GL90_IN_SSW")
  ! a bulk mixed layer: its GV%nkml layers are the viscous surface boundary layer (find_coupling_coef :2152-2166)
  ! (GV%nkml is zero unless BULKMIXEDLAYER, MOM.F90:2439-2445 / MOM_verticalGrid)
  nkml = GV%nkml
  CS%st%nkml = nkml
  call get_param(param_file, mdl, "VON_KARMAN_CONST", CS%st%vonKar, "The value the von Karman constant as used for mixed layer viscosity.", &
                 units="nondim", default=0.41)
  call get_param(param_file, mdl, "HARMONIC_VISC", flag, &
                 "If true, use the harmonic mean thicknesses for calculating the vertical viscosity.", default=.false.)
  CS%st%harmonic_visc = merge(1, 0, flag)
  call get_param(param_file, mdl, "HARMONIC_BL_SCALE", CS%st%harm_BL_val, &
                 "A scale to determine when water is in the boundary layers based solely on harmonic mean thicknesses.", &
                 units="nondim", default=0.0)
  Hmix_m = 0.0
  if (GV%nkml < 1) call get_param(param_file, mdl, "HMIX_FIXED", Hmix_m, &      ! (:2582-2587)
                 "The prescribed depth over which the near-surface viscosity and diffusivity are elevated when the bulk mixed layer "// &
                 "is not used.", units="m", scale=US%m_to_Z, fail_if_missing=.true.)
  CS%st%Hmix = Hmix_m
  call get_param(param_file, mdl, "HMIX_STRESS", Hmix_stress_m, &
                 "The depth over which the wind stress is applied if DIRECT_STRESS is true.", units="m", default=Hmix_m*US%Z_to_m, &
                 scale=US%m_to_Z)
  CS%st%Hmix_stress = Hmix_stress_m * GV%Z_to_H
  if (CS%st%direct_stress /= 0 .and. CS%st%Hmix_stress <= 0.0) call MOM_error(FATAL, "vertvisc_init: " // &
       "HMIX_STRESS must be set to a positive value if DIRECT_STRESS is true.")
  call get_param(param_file, mdl, "KV", CS%st%Kv, "The background kinematic viscosity in the interior.", units="m2 s-1", &
                 fail_if_missing=.true., scale=US%m_to_Z**2*US%T_to_s)
  CS%st%Kvml_invZ2 = 0.0
  if (GV%nkml < 1) call get_param(param_file, mdl, "KV_ML_INVZ2", CS%st%Kvml_invZ2, &      ! (:2672-2690)
                 "An extra kinematic viscosity in a mixed layer of thickness HMIX_FIXED, with the actual viscosity scaling as 1/(z*HMIX_FIXED)^2.", &
                 units="m2 s-1", default=0.0, scale=US%m_to_Z**2*US%T_to_s)
  call get_param(param_file, mdl, "KV_EXTRA_BBL", CS%st%Kv_extra_bbl, &
                 "An extra kinematic viscosity in the benthic boundary layer. KV_EXTRA_BBL is not used if BOTTOMDRAGLAW is true.", &
                 units="m2 s-1", default=0.0, scale=US%m_to_Z**2*US%T_to_s)
  call get_param(param_file, mdl, "HBBL", CS%st%Hbbl, "The thickness of a bottom boundary layer with a viscosity increased by KV_EXTRA_BBL.", &
                 units="m", fail_if_missing=.true., scale=US%m_to_Z)
  call get_param(param_file, mdl, "MAXVEL", CS%st%maxvel, "The maximum velocity allowed before the velocity components are truncated.", &
                 units="m s-1", default=3.0e8, scale=US%m_s_to_L_T)
  call get_param(param_file, mdl, "CFL_BASED_TRUNCATIONS", flag, &
                 "If true, base truncations on the CFL number, and not an absolute speed.", default=.true.)
  CS%st%CFL_based_trunc = merge(1, 0, flag)
  call get_param(param_file, mdl, "CFL_TRUNCATE", CS%st%CFL_trunc, "The value of the CFL number that will cause velocity components to be truncated.", &
                 units="nondim", default=0.5)
  call get_param(param_file, mdl, "CFL_TRUNCATE_RAMP_TIME", val, units="s", default=0.)
  call refuse(val > 0.0, "CFL_TRUNCATE_RAMP_TIME > 0")
  call get_param(param_file, mdl, "VEL_UNDERFLOW", CS%st%vel_underflow, &
                 "A negligibly small velocity magnitude below which velocity components are set to 0.", units="m s-1", default=0.0, &
                 scale=US%m_s_to_L_T)
  CS%st%H_to_RZ = GV%H_to_RZ

  allocate(CS%a_u(isd-1:ied,jsd:jed,nz+1), source=0.0) ; allocate(CS%a_v(isd:ied,jsd-1:jed,nz+1), source=0.0)
  allocate(CS%h_u(isd-1:ied,jsd:jed,nz), source=0.0) ; allocate(CS%h_v(isd:ied,jsd-1:jed,nz), source=0.0)
  call mom6hip_read_topology(param_file)
contains
  subroutine refuse(on, name)
    logical,          intent(in) :: on
    character(len=*), intent(in) :: name
    if (on) call MOM_error(FATAL, "vertvisc_init (HIP): "//name//" is not provided by the GPU path.")
  end subroutine refuse
end subroutine vertvisc_init

!> Same interface as the reference updateCFLtruncationValue (:2809): without a ramp the value never changes
subroutine updateCFLtruncationValue(Time, CS, US, activate)
  type(time_type), target, intent(in)    :: Time
  type(vertvisc_CS),       pointer       :: CS
  type(unit_scale_type),   intent(in)    :: US
  logical, optional,       intent(in)    :: activate
end subroutine updateCFLtruncationValue

!> Same interface as the reference vertvisc_end
subroutine vertvisc_end(CS)
  type(vertvisc_CS), intent(inout) :: CS
  if (allocated(CS%a_u)) deallocate(CS%a_u, CS%a_v, CS%h_u, CS%h_v)
  CS%initialized = .false.
end subroutine vertvisc_end

end module MOM_vert_friction
