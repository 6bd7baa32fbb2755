!> Drop-in replacement for module MOM_set_visc (src/parameterizations/vertical/MOM_set_viscosity.F90): set_viscous_BBL (:134),
!! set_viscous_ML (:1898), set_visc_init (:2886), set_visc_register_restarts (:2693), set_visc_end with the reference's
!! dummy-argument lists, so MOM.F90 (:1205) and the split RK2 step (:592) compile unchanged.  The work is done by libmom6hip
!! (mom6hip_set_viscous_bbl, HOST memspace).  Provided: BOTTOMDRAGLAW with LINEAR_DRAG or the quadratic law (CDRAG,
!! DRAG_BG_VEL), BBL_USE_EOS (WRIGHT / UNESCO / LINEAR, read from the parameter file: EOS_type is opaque) or GV%Rlay, HBBL,
!! BBL_THICK_MIN, KV_BBL_MIN, CORRECT_BBL_BOUNDS, DRAG_AS_BODY_FORCE.  CHANNEL_DRAG, BBL_USE_TIDAL_BG, DYNAMIC_VISCOUS_ML, a
!! bulk mixed layer, ice shelves, open boundaries and porous barriers stop with a FATAL error.
!!
!! Compiled INSIDE a MOM6 source tree in place of the reference file; here against tests/fortran/stubs.
module MOM_set_visc

use, intrinsic :: iso_c_binding
use mom6hip_c_api
use mom6hip_MOM_glue,     only : mom6hip_shared_context, mom6hip_read_topology, mom6hip_fatal_if
use mom6hip_MOM_glue,     only : mom6hip_read_resident, mom6hip_resident, mom6hip_mirror, mom6hip_obc_to_c
use MOM_ALE,              only : ALE_CS
use MOM_diag_mediator,    only : diag_ctrl, time_type
use MOM_error_handler,    only : MOM_error, FATAL, WARNING
use MOM_file_parser,      only : get_param, log_version, param_file_type
use MOM_forcing_type,     only : mech_forcing
use MOM_grid,             only : ocean_grid_type
use MOM_hor_index,        only : hor_index_type
use MOM_open_boundary,    only : ocean_OBC_type
use MOM_restart,          only : register_restart_field, MOM_restart_CS
use MOM_string_functions, only : uppercase
use MOM_unit_scaling,     only : unit_scale_type
use MOM_variables,        only : thermo_var_ptrs, vertvisc_type, porous_barrier_type
use MOM_verticalGrid,     only : verticalGrid_type
implicit none ; private

#include <MOM_memory.h>

public set_viscous_BBL, set_viscous_ML, set_visc_init, set_visc_end
public set_visc_register_restarts, set_u_at_v, set_v_at_u
public remap_vertvisc_aux_vars
public set_visc_hip_struct      ! (GPU path only) for MOM_dynamics_split_RK2

!> Control structure: the library's struct and the equation of state read at initialisation
type, public :: set_visc_CS ; private
  logical :: initialized = .false.
  type(mom6hip_set_visc_cs_t) :: st
  type(mom6hip_eos_t) :: eos
  real, allocatable :: Rlay(:)
  type(diag_ctrl), pointer :: diag => NULL()
  type(ocean_OBC_type), pointer :: OBC => NULL()      !< Open boundaries control structure (set_visc_init :2903)
end type set_visc_CS

contains

!> (GPU path only) The library's struct of this control structure (Rlay bound to the HOST copy kept by CS) and the equation of state
subroutine set_visc_hip_struct(CS, GV, st, eos)
  type(set_visc_CS), target, intent(inout) :: CS
  type(verticalGrid_type), intent(in) :: GV
  type(mom6hip_set_visc_cs_t), intent(out) :: st
  type(mom6hip_eos_t), intent(out) :: eos
  if (.not.CS%initialized) call MOM_error(FATAL, "MOM_set_viscosity(visc_ML): Module must be initialized before it is used.")
  CS%st%Rlay = c_null_ptr ; if (allocated(CS%Rlay)) CS%st%Rlay = c_loc(CS%Rlay)
  CS%st%nkml = GV%nkml
  st = CS%st ; eos = CS%eos
end subroutine set_visc_hip_struct

!> Same interface as the reference set_viscous_BBL (:134).
subroutine set_viscous_BBL(u, v, h, tv, visc, G, GV, US, CS, pbv)
  type(ocean_grid_type),    intent(inout) :: G
  type(verticalGrid_type),  intent(in)    :: GV
  type(unit_scale_type),    intent(in)    :: US
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(in) :: u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(in) :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  target, intent(in) :: h
  type(thermo_var_ptrs),    intent(in)    :: tv
  type(vertvisc_type), target, intent(inout) :: visc
  type(set_visc_CS),   target, intent(inout) :: CS
  type(porous_barrier_type),intent(in)    :: pbv
  type(mom6hip_vertvisc_type_t) :: cv
  type(mom6hip_eos_t), target :: eos
  type(mom6hip_obc_t) :: cobc
  type(mom6hip_obc_segment_t), allocatable, target :: csegs(:)
  type(c_ptr) :: p_T, p_S, ctx
  integer :: rc
  if (.not.CS%initialized) call MOM_error(FATAL, "MOM_set_viscosity(BBL): Module must be initialized before it is used.")
  if (CS%st%bottomdraglaw == 0) return      ! :307
  if (allocated(pbv%por_layer_widthU)) then
    if (any(pbv%por_layer_widthU /= 1.0) .or. any(pbv%por_layer_widthV /= 1.0)) &
      call MOM_error(FATAL, "set_viscous_BBL (HIP): porous barriers are not supported by the GPU path.")
  endif
  if (.not.(allocated(visc%bbl_thick_u) .and. allocated(visc%Kv_bbl_u))) call MOM_error(FATAL, "set_viscous_BBL (HIP): "// &
       "visc%bbl_thick_u/v and visc%Kv_bbl_u/v must be allocated (set_visc_init does it).")
  cv%Kv_bbl_u = c_loc(visc%Kv_bbl_u) ; cv%Kv_bbl_v = c_loc(visc%Kv_bbl_v)
  cv%bbl_thick_u = c_loc(visc%bbl_thick_u) ; cv%bbl_thick_v = c_loc(visc%bbl_thick_v)
  cv%Ray_u = c_null_ptr ; if (allocated(visc%Ray_u)) cv%Ray_u = c_loc(visc%Ray_u)
  cv%Ray_v = c_null_ptr ; if (allocated(visc%Ray_v)) cv%Ray_v = c_loc(visc%Ray_v)
  cv%Kv_shear = c_null_ptr ; cv%Kv_shear_Bu = c_null_ptr ; cv%reserved(:) = c_null_ptr
  cv%nkml_visc_u = c_null_ptr ; cv%nkml_visc_v = c_null_ptr ; cv%ustar = c_null_ptr
  p_T = c_null_ptr ; p_S = c_null_ptr
  if (CS%st%BBL_use_EOS /= 0) then
    if (.not.(associated(tv%T) .and. associated(tv%S))) &
      call MOM_error(FATAL, "set_viscous_BBL (HIP): BBL_USE_EOS needs tv%T and tv%S.")
    p_T = c_loc(tv%T) ; p_S = c_loc(tv%S)
  endif
  eos = CS%eos
  CS%st%Rlay = c_null_ptr ; if (allocated(CS%Rlay)) CS%st%Rlay = c_loc(CS%Rlay)
  ctx = mom6hip_shared_context(G, GV)
  ! the OBC branches :374-413, :502-580 and those of set_v_at_u / set_u_at_v read the segments' directions and ranges (host tables), none of
  ! their arrays
  if (associated(CS%OBC)) call mom6hip_obc_to_c(CS%OBC, cobc, csegs, size(u(:,:,1)), size(v(:,:,1)), "MOM_set_viscosity")
  if (mom6hip_resident()) then      ! GPU_RESIDENT_DYNAMICS: u, v, h, T, S where the step left them; visc%... where the step reads them
    call to_dev(cv%Kv_bbl_u, size(visc%Kv_bbl_u), .true.) ; call to_dev(cv%Kv_bbl_v, size(visc%Kv_bbl_v), .true.)
    call to_dev(cv%bbl_thick_u, size(visc%bbl_thick_u), .true.) ; call to_dev(cv%bbl_thick_v, size(visc%bbl_thick_v), .true.)
    if (allocated(visc%Ray_u)) then
      call to_dev(cv%Ray_u, size(visc%Ray_u), .true.) ; call to_dev(cv%Ray_v, size(visc%Ray_v), .true.)
    endif
    call to_dev(p_T, size(h), .false.) ; call to_dev(p_S, size(h), .false.)
    if (associated(CS%OBC)) then
      rc = mom6hip_set_viscous_bbl_obc(ctx, CS%st, mom6hip_mirror(ctx, c_loc(u), int(size(u), c_int64_t), .true., .false.), &
                                       mom6hip_mirror(ctx, c_loc(v), int(size(v), c_int64_t), .true., .false.), &
                                       mom6hip_mirror(ctx, c_loc(h), int(size(h), c_int64_t), .true., .false.), p_T, p_S, c_loc(eos), cv, &
                                       cobc, MOM6HIP_MEM_DEVICE)
    else
      rc = mom6hip_set_viscous_bbl(ctx, CS%st, mom6hip_mirror(ctx, c_loc(u), int(size(u), c_int64_t), .true., .false.), &
                                   mom6hip_mirror(ctx, c_loc(v), int(size(v), c_int64_t), .true., .false.), &
                                   mom6hip_mirror(ctx, c_loc(h), int(size(h), c_int64_t), .true., .false.), p_T, p_S, c_loc(eos), cv, &
                                   MOM6HIP_MEM_DEVICE)
    endif
  elseif (associated(CS%OBC)) then
    rc = mom6hip_set_viscous_bbl_obc(ctx, CS%st, c_loc(u), c_loc(v), c_loc(h), p_T, p_S, c_loc(eos), cv, cobc, MOM6HIP_MEM_HOST)
  else
    rc = mom6hip_set_viscous_bbl(ctx, CS%st, c_loc(u), c_loc(v), c_loc(h), p_T, p_S, c_loc(eos), cv, MOM6HIP_MEM_HOST)
  endif
  call mom6hip_fatal_if(rc, "set_viscous_BBL")
contains
  !> a host pointer -> its device mirror (an input, or an in/out array the call writes)
  subroutine to_dev(p, n, written)
    type(c_ptr), intent(inout) :: p
    integer,     intent(in)    :: n
    logical,     intent(in)    :: written
    if (c_associated(p)) p = mom6hip_mirror(ctx, p, int(n, c_int64_t), .true., written)
  end subroutine to_dev
end subroutine set_viscous_BBL

!> Same interface as the reference set_viscous_ML (:1898): returns as the reference does without DYNAMIC_VISCOUS_ML (:2043); with it,
!! the viscous surface boundary layer goes to visc%nkml_visc_u / visc%nkml_visc_v (forces%ustar as find_ustar hands it over in
!! Boussinesq mode, MOM_forcing_type.F90:1265-1274).
subroutine set_viscous_ML(u, v, h, tv, forces, visc, dt, G, GV, US, CS)
  type(ocean_grid_type),   intent(inout) :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(unit_scale_type),   intent(in)    :: US
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(in) :: u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(in) :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  target, intent(in) :: h
  type(thermo_var_ptrs),   intent(in)    :: tv
  type(mech_forcing),      intent(in)    :: forces
  type(vertvisc_type), target, intent(inout) :: visc
  real,                    intent(in)    :: dt
  type(set_visc_CS),   target, intent(inout) :: CS
  type(mom6hip_vertvisc_type_t) :: cv
  type(c_ptr) :: p_T, p_S, p_eos
  integer :: rc
  if (.not.CS%initialized) call MOM_error(FATAL, "MOM_set_viscosity(visc_ML): Module must be initialized before it is used.")
  if (CS%st%dynamic_viscous_ML == 0) return      ! :2043-2044 (ice shelves are refused at initialisation)
  ! (an associated CS%OBC: the masks of :2099-2117 are read by set_v_at_u / set_u_at_v under ice shelves only (:2242-2391, :2519-2670),
  ! which set_visc_init has refused: the open boundaries leave set_viscous_ML as it is)
  if (.not.(associated(forces%taux) .and. associated(forces%tauy))) call MOM_error(FATAL, "set_viscous_ML (HIP): "// &
       "forces%taux and forces%tauy must be associated.")
  if (.not.associated(forces%ustar)) call MOM_error(FATAL, "set_viscous_ML (HIP): forces%ustar must be associated (the GPU "// &
       "path is Boussinesq: find_ustar returns forces%ustar).")
  if (associated(tv%p_surf)) call MOM_error(FATAL, "set_viscous_ML (HIP): tv%p_surf is not provided by the GPU path.")
  if (.not.(allocated(visc%nkml_visc_u) .and. allocated(visc%nkml_visc_v))) call MOM_error(FATAL, "set_viscous_ML (HIP): "// &
       "visc%nkml_visc_u/v must be allocated (set_visc_init does it).")
  cv%Kv_bbl_u = c_null_ptr ; cv%Kv_bbl_v = c_null_ptr ; cv%bbl_thick_u = c_null_ptr ; cv%bbl_thick_v = c_null_ptr
  cv%Ray_u = c_null_ptr ; cv%Ray_v = c_null_ptr ; cv%Kv_shear = c_null_ptr ; cv%Kv_shear_Bu = c_null_ptr ; cv%reserved(:) = c_null_ptr
  cv%nkml_visc_u = c_loc(visc%nkml_visc_u) ; cv%nkml_visc_v = c_loc(visc%nkml_visc_v) ; cv%ustar = c_loc(forces%ustar)
  p_T = c_null_ptr ; p_S = c_null_ptr ; p_eos = c_null_ptr
  if (associated(tv%eqn_of_state)) then
    if (.not.(associated(tv%T) .and. associated(tv%S))) call MOM_error(FATAL, "set_viscous_ML (HIP): an equation of state needs tv%T, tv%S.")
    p_T = c_loc(tv%T) ; p_S = c_loc(tv%S) ; p_eos = c_loc(CS%eos)
  endif
  CS%st%Rlay = c_null_ptr ; if (allocated(CS%Rlay)) CS%st%Rlay = c_loc(CS%Rlay)
  CS%st%nkml = GV%nkml
  rc = mom6hip_set_viscous_ml(mom6hip_shared_context(G, GV), CS%st, c_loc(u), c_loc(v), c_loc(h), p_T, p_S, p_eos, &
                              c_loc(forces%taux), c_loc(forces%tauy), cv, real(dt, c_double), MOM6HIP_MEM_HOST)
  call mom6hip_fatal_if(rc, "set_viscous_ML")
end subroutine set_viscous_ML

!> Same interface as the reference set_visc_register_restarts (:2693): nothing of the provided branch is in the restart file
subroutine set_visc_register_restarts(HI, G, GV, US, param_file, visc, restart_CS, use_ice_shelf)
  type(hor_index_type),    intent(in)    :: HI
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(unit_scale_type),   intent(in)    :: US
  type(param_file_type),   intent(in)    :: param_file
  type(vertvisc_type),     intent(inout) :: visc
  type(MOM_restart_CS),    intent(inout) :: restart_CS
  logical,                 intent(in)    :: use_ice_shelf
  if (use_ice_shelf) call MOM_error(FATAL, "set_visc_register_restarts (HIP): ice shelves are not provided by the GPU path.")
end subroutine set_visc_register_restarts

!> Same interface as the reference set_visc_init (:2886), same parameters and defaults.
subroutine set_visc_init(Time, G, GV, US, param_file, diag, visc, CS, restart_CS, OBC)
  type(time_type), target, intent(in)    :: Time
  type(ocean_grid_type),   intent(inout) :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(unit_scale_type),   intent(in)    :: US
  type(param_file_type),   intent(in)    :: param_file
  type(diag_ctrl), target, intent(inout) :: diag
  type(vertvisc_type),     intent(inout) :: visc
  type(set_visc_CS),       intent(inout) :: CS
  type(MOM_restart_CS),    intent(inout) :: restart_CS
  type(ocean_OBC_type),    pointer       :: OBC
# include "version_variable.h"
  character(len=40)  :: mdl = "MOM_set_visc"
  character(len=40)  :: tmpstr
  logical :: flag, flag2, use_EOS, use_temperature, use_regridding
  real :: Kv_background, val
  integer :: isd, ied, jsd, jed, default_answer_date, answer_date

  CS%initialized = .true. ; CS%diag => diag
  isd = G%isd ; ied = G%ied ; jsd = G%jsd ; jed = G%jed
  CS%OBC => OBC      ! :2903
  if (.not.GV%Boussinesq) call refuse(.true., "a non-Boussinesq vertical grid")
  CS%st%unsupported(:) = 0 ; CS%st%reserved1(:) = c_null_ptr ; CS%st%Rlay = c_null_ptr
  CS%st%nkml = GV%nkml
  call log_version(param_file, mdl, version, "")
  call get_param(param_file, mdl, "BOTTOMDRAGLAW", flag, &
                 "If true, the bottom stress is calculated with a drag law of the form c_drag*|u|*u.", default=.true.)
  CS%st%bottomdraglaw = merge(1, 0, flag)
  call get_param(param_file, mdl, "DRAG_AS_BODY_FORCE", flag, &
                 "If true, the bottom stress is imposed as an explicit body force applied over a fixed distance from the bottom.", &
                 default=.false.)
  CS%st%body_force_drag = merge(1, 0, flag)
  call get_param(param_file, mdl, "CHANNEL_DRAG", flag, &
                 "If true, the bottom drag is exerted directly on each layer proportional to the fraction of the bottom it overlies.", &
                 default=.false.)
  CS%st%Channel_drag = merge(1, 0, flag)
  call get_param(param_file, mdl, "LINEAR_DRAG", flag, "If LINEAR_DRAG and BOTTOMDRAGLAW are defined the drag law is cdrag*DRAG_BG_VEL*u.", &
                 default=.false.)
  CS%st%linear_drag = merge(1, 0, flag)
  call get_param(param_file, mdl, "USE_JACKSON_PARAM", flag, default=.false., do_not_log=.true.)
  CS%st%RiNo_mix = merge(1, 0, flag)
  ! DYNAMIC_VISCOUS_ML and its parameters (:2962-2999)
  call get_param(param_file, mdl, "DYNAMIC_VISCOUS_ML", flag, &
                 "If true, use a bulk Richardson number criterion to determine the mixed layer thickness for viscosity.", default=.false.)
  CS%st%dynamic_viscous_ML = merge(1, 0, flag)
  CS%st%bulk_Ri_ML = 0.0 ; CS%st%TKE_decay = 0.0 ; CS%st%omega_frac = 0.0 ; CS%st%ustar_min = 0.0
  if (flag) then
    call get_param(param_file, mdl, "BULK_RI_ML", val, units="nondim", default=0.0)
    call get_param(param_file, mdl, "BULK_RI_ML_VISC", CS%st%bulk_Ri_ML, &
                 "The efficiency with which mean kinetic energy released by mechanically forced entrainment of the mixed layer "// &
                 "is converted to turbulent kinetic energy.  By default, BULK_RI_ML_VISC = BULK_RI_ML or 0.", units="nondim", default=val)
    call get_param(param_file, mdl, "TKE_DECAY", val, units="nondim", default=0.0)
    call get_param(param_file, mdl, "TKE_DECAY_VISC", CS%st%TKE_decay, &
                 "TKE_DECAY_VISC relates the vertical rate of decay of the TKE available for mechanical entrainment to the natural "// &
                 "Ekman depth for use in calculating the dynamic mixed layer viscosity.  By default, TKE_DECAY_VISC = TKE_DECAY or 0.", &
                 units="nondim", default=val)
    call get_param(param_file, mdl, "ML_USE_OMEGA", flag2, default=.false., do_not_log=.true.)
    val = 0.0 ; if (flag2) val = 1.0
    call get_param(param_file, mdl, "ML_OMEGA_FRAC", CS%st%omega_frac, &
                 "When setting the decay scale for turbulence, use this fraction of the absolute rotation rate blended with the "// &
                 "local value of f, as sqrt((1-of)*f^2 + of*4*omega^2).", units="nondim", default=val)
  endif
  call get_param(param_file, mdl, "OMEGA", CS%st%omega, "The rotation rate of the earth.", units="s-1", default=7.2921e-5, scale=US%T_to_s)
  if (CS%st%dynamic_viscous_ML /= 0) CS%st%ustar_min = 2e-4*CS%st%omega*(GV%Angstrom_H + GV%H_subroundoff)      ! :2998
  call get_param(param_file, mdl, "HBBL", CS%st%dz_bbl, "The thickness of a bottom boundary layer.", units="m", &
                 fail_if_missing=.true., scale=US%m_to_Z)
  CS%st%Hbbl = CS%st%dz_bbl * GV%Z_to_H      ! :3127
  call get_param(param_file, mdl, "CDRAG", CS%st%cdrag, "The drag coefficient relating the magnitude of the velocity field to the bottom stress.", &
                 units="nondim", default=0.003)
  call get_param(param_file, mdl, "BBL_USE_TIDAL_BG", flag, default=.false.) ; call refuse(flag, "BBL_USE_TIDAL_BG")
  call get_param(param_file, mdl, "DRAG_BG_VEL", CS%st%drag_bg_vel, &
                 "The assumed bottom velocity magnitude used with LINEAR_DRAG, or an unresolved velocity within the bottom boundary layer.", &
                 units="m s-1", default=0.0, scale=US%m_s_to_L_T)
  call get_param(param_file, "MOM", "USE_REGRIDDING", use_regridding, default=.false., do_not_log=.true.)
  call get_param(param_file, "MOM", "ENABLE_THERMODYNAMICS", use_temperature, default=.true., do_not_log=.true.)
  use_EOS = .false.
  if (use_temperature) call get_param(param_file, "MOM", "USE_EOS", use_EOS, default=.true., do_not_log=.true.)
  call get_param(param_file, mdl, "BBL_USE_EOS", flag, &
                 "If true, use the equation of state in determining the properties of the bottom boundary layer.", &
                 default=use_EOS, do_not_log=.not.use_temperature)
  CS%st%BBL_use_EOS = merge(1, 0, flag)
  call get_param(param_file, mdl, "BBL_THICK_MIN", CS%st%BBL_thick_min, "The minimum bottom boundary layer thickness.", units="m", &
                 default=0.0, scale=US%m_to_Z)
  call get_param(param_file, mdl, "KV", Kv_background, "The background kinematic viscosity in the interior.", units="m2 s-1", &
                 fail_if_missing=.true., scale=US%m_to_Z**2*US%T_to_s)
  call get_param(param_file, mdl, "KV_BBL_MIN", CS%st%Kv_BBL_min, "The minimum viscosities in the bottom boundary layer.", &
                 units="m2 s-1", default=US%Z_to_m**2*US%s_to_T*Kv_background, scale=US%m_to_Z**2*US%T_to_s)
  call get_param(param_file, mdl, "CORRECT_BBL_BOUNDS", flag, &
                 "If true, uses the correct bounds on the BBL thickness and viscosity so that the bottom layer feels the intended drag.", &
                 default=.false.)
  CS%st%correct_BBL_bounds = merge(1, 0, flag)
  ! CHANNEL_DRAG (:3092-3122)
  CS%st%c_Smag = 0.15 ; CS%st%concave_trigonometric_L = 1 ; CS%st%Z_ref = G%Z_ref
  if (CS%st%Channel_drag /= 0) then
    call get_param(param_file, mdl, "SMAG_LAP_CONST", val, units="nondim", default=-1.0)
    if (val < 0.0) val = 0.15
    call get_param(param_file, mdl, "SMAG_CONST_CHANNEL", CS%st%c_Smag, &
                 "The nondimensional Laplacian Smagorinsky constant used in calculating the channel drag if it is enabled.  The default "// &
                 "is to use the same value as SMAG_LAP_CONST if it is defined, or 0.15 if it is not.", units="nondim", default=val)
    if (CS%st%c_Smag < 0.0) CS%st%c_Smag = 0.15
    call get_param(param_file, mdl, "TRIG_CHANNEL_DRAG_WIDTHS", flag, &
                 "If true, use trigonometric expressions to determine the fractional open interface lengths for concave topography.", &
                 default=.true.)
    CS%st%concave_trigonometric_L = merge(1, 0, flag)
    call get_param(param_file, mdl, "DEFAULT_ANSWER_DATE", default_answer_date, default=99991231)
    call get_param(param_file, mdl, "SET_VISC_ANSWER_DATE", answer_date, default=default_answer_date)
    call refuse(answer_date < 20190101, "CHANNEL_DRAG with SET_VISC_ANSWER_DATE < 20190101")
  endif
  val = -1.0
  if (CS%st%RiNo_mix /= 0) val = 0.5*CS%st%dz_bbl
  if (CS%st%body_force_drag /= 0) val = CS%st%dz_bbl
  call get_param(param_file, mdl, "CHANNEL_DRAG_MAX_BBL_THICK", CS%st%Chan_drag_max_vol, &
                 "The maximum bottom boundary layer thickness over which the channel drag is exerted, or a negative value for no fixed "// &
                 "limit.  The default is proportional to HBBL if USE_JACKSON_PARAM or DRAG_AS_BODY_FORCE is true.", &
                 units="m", default=US%Z_to_m*val, scale=US%m_to_Z, do_not_log=(CS%st%Channel_drag == 0))
  CS%st%BBL_thick_max = 6.378e6      ! G%Rad_Earth_L * US%L_to_Z (:3128)
  CS%st%H_to_RZ = GV%H_to_RZ
  CS%st%initialized = 1
  if (allocated(GV%Rlay)) then ; allocate(CS%Rlay(GV%ke)) ; CS%Rlay(:) = GV%Rlay(1:GV%ke) ; endif
  ! the equation of state, as interpret_eos_selection reads it (MOM_EOS.F90:1474-1520)
  CS%eos%reserved = 0 ; CS%eos%Rho_T0_S0 = 1000.0 ; CS%eos%dRho_dT = -0.2 ; CS%eos%dRho_dS = 0.8 ; CS%eos%form = MOM6HIP_EOS_WRIGHT
  if (CS%st%BBL_use_EOS /= 0) then
    call get_param(param_file, "MOM_EOS", "EQN_OF_STATE", tmpstr, default="WRIGHT")
    select case (uppercase(tmpstr))
      case ("LINEAR")
        CS%eos%form = MOM6HIP_EOS_LINEAR
        call get_param(param_file, "MOM_EOS", "RHO_T0_S0", CS%eos%Rho_T0_S0, units="kg m-3", default=1000.0)
        call get_param(param_file, "MOM_EOS", "DRHO_DT", CS%eos%dRho_dT, units="kg m-3 K-1", default=-0.2)
        call get_param(param_file, "MOM_EOS", "DRHO_DS", CS%eos%dRho_dS, units="kg m-3 ppt-1", default=0.8)
      case ("WRIGHT")
        CS%eos%form = MOM6HIP_EOS_WRIGHT
      case ("UNESCO", "JACKETT_MCD")
        CS%eos%form = MOM6HIP_EOS_UNESCO
      case ("WRIGHT_FULL")
        CS%eos%form = MOM6HIP_EOS_WRIGHT_FULL
      case ("WRIGHT_REDUCED")
        CS%eos%form = MOM6HIP_EOS_WRIGHT_REDUCED
      case default
        call refuse(.true., "EQN_OF_STATE "//trim(tmpstr))
    end select
  endif
  ! the arrays set_viscous_BBL fills (:3133-3140)
  if (CS%st%bottomdraglaw /= 0) then
    if (.not.allocated(visc%bbl_thick_u)) allocate(visc%bbl_thick_u(isd-1:ied,jsd:jed), source=0.0)
    if (.not.allocated(visc%bbl_thick_v)) allocate(visc%bbl_thick_v(isd:ied,jsd-1:jed), source=0.0)
    if (.not.allocated(visc%Kv_bbl_u)) allocate(visc%Kv_bbl_u(isd-1:ied,jsd:jed), source=0.0)
    if (.not.allocated(visc%Kv_bbl_v)) allocate(visc%Kv_bbl_v(isd:ied,jsd-1:jed), source=0.0)
  endif
  if ((CS%st%Channel_drag /= 0) .or. (CS%st%body_force_drag /= 0)) then      ! :3168-3170
    if (.not.allocated(visc%Ray_u)) allocate(visc%Ray_u(isd-1:ied,jsd:jed,GV%ke), source=0.0)
    if (.not.allocated(visc%Ray_v)) allocate(visc%Ray_v(isd:ied,jsd-1:jed,GV%ke), source=0.0)
  endif
  if (CS%st%dynamic_viscous_ML /= 0) then      ! :3178-3180
    if (.not.allocated(visc%nkml_visc_u)) allocate(visc%nkml_visc_u(isd-1:ied,jsd:jed), source=0.0)
    if (.not.allocated(visc%nkml_visc_v)) allocate(visc%nkml_visc_v(isd:ied,jsd-1:jed), source=0.0)
  endif
  call mom6hip_read_resident(param_file)
  call mom6hip_read_topology(param_file)
contains
  subroutine refuse(on, name)
    logical,          intent(in) :: on
    character(len=*), intent(in) :: name
    if (on) call MOM_error(FATAL, "set_visc_init (HIP): "//name//" is not provided by the GPU path.")
  end subroutine refuse
end subroutine set_visc_init

!> Same interface as the reference set_visc_end
subroutine set_visc_end(visc, CS)
  type(vertvisc_type), intent(inout) :: visc
  type(set_visc_CS),   intent(inout) :: CS
  if (allocated(visc%bbl_thick_u)) deallocate(visc%bbl_thick_u, visc%bbl_thick_v, visc%Kv_bbl_u, visc%Kv_bbl_v)
  if (allocated(visc%Ray_u)) deallocate(visc%Ray_u, visc%Ray_v)
  if (allocated(CS%Rlay)) deallocate(CS%Rlay)
  CS%initialized = .false.
end subroutine set_visc_end

!> Same interface as the reference remap_vertvisc_aux_vars: the auxiliary viscosities it remaps are not provided
subroutine remap_vertvisc_aux_vars(G, GV, visc, h_old, h_new, ALE_CSp, OBC)
  type(ocean_grid_type),   intent(inout) :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(vertvisc_type),     intent(inout) :: visc
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), intent(in) :: h_old, h_new
  type(ALE_CS),            pointer       :: ALE_CSp
  type(ocean_OBC_type),    pointer       :: OBC
  if (associated(visc%Kv_shear) .or. associated(visc%Kv_shear_Bu)) &
    call MOM_error(FATAL, "remap_vertvisc_aux_vars (HIP): remapping of Kv_shear is not provided by the GPU path.")
end subroutine remap_vertvisc_aux_vars

!> Same interface as the reference set_u_at_v: a host helper of set_viscous_BBL, which runs on the GPU here
function set_u_at_v(u, h, G, GV, i, j, k, mask2dCu, OBC)
  type(ocean_grid_type),   intent(in) :: G
  type(verticalGrid_type), intent(in) :: GV
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in) :: u
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in) :: h
  integer,                 intent(in) :: i, j, k
  real, dimension(SZIB_(G),SZJ_(G)), intent(in) :: mask2dCu
  type(ocean_OBC_type),    pointer    :: OBC
  real                                :: set_u_at_v
  set_u_at_v = 0.0
  call MOM_error(FATAL, "set_u_at_v (HIP): not provided as a host routine; set_viscous_BBL forms it on the GPU.")
end function set_u_at_v

!> Same interface as the reference set_v_at_u
function set_v_at_u(v, h, G, GV, i, j, k, mask2dCv, OBC)
  type(ocean_grid_type),   intent(in) :: G
  type(verticalGrid_type), intent(in) :: GV
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in) :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in) :: h
  integer,                 intent(in) :: i, j, k
  real, dimension(SZI_(G),SZJB_(G)), intent(in) :: mask2dCv
  type(ocean_OBC_type),    pointer    :: OBC
  real                                :: set_v_at_u
  set_v_at_u = 0.0
  call MOM_error(FATAL, "set_v_at_u (HIP): not provided as a host routine; set_viscous_BBL forms it on the GPU.")
end function set_v_at_u

end module MOM_set_visc
