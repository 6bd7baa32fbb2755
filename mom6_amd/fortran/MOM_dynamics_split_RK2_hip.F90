!> Drop-in replacement for module MOM_dynamics_split_RK2 (src/core/MOM_dynamics_split_RK2.F90): step_MOM_dyn_split_RK2 (:289),
!! register_restarts_dyn_split_RK2 (:1181), remap_dyn_split_RK2_aux_vars (:1273), initialize_dyn_split_RK2 (:1317) and
!! end_dyn_split_RK2 (:1829) with the reference's dummy-argument lists, parameter names and defaults, so that MOM.F90 compiles
!! unchanged.  The whole baroclinic step is ONE library call (mom6hip_step_dyn_split_rk2) on DEVICE-RESIDENT state:
!!
!!   * every array the reference keeps in MOM_dyn_split_RK2_CS (:84-268), in barotropic_CS, vertvisc_CS, hor_visc_CS and BT_cont_type
!!     lives in HBM for the whole run (allocated at initialisation, the static ones uploaded once);
!!   * the arguments of the step (u, v, h, tv%T, tv%S, uh, vh, uhtr, vhtr, eta_av, forces%taux / tauy / ustar, the members of visc)
!!     are HOST arrays of the caller; each gets a device mirror keyed on its host address.  A mirror is uploaded when the host copy is
!!     newer and copied back when the host asks for it:
!!       - GPU_RESIDENT_DYNAMICS = False (default): every step uploads its inputs and copies its outputs back -- correct inside an
!!         unchanged MOM.F90 whatever runs between two steps, at the price of the PCIe transfers;
!!       - GPU_RESIDENT_DYNAMICS = True: inputs are uploaded only when they are new or after dyn_split_RK2_host_was_modified, and
!!         outputs stay on the device until dyn_split_RK2_sync_to_host (staged copies on the copy stream) -- N steps then cost one
!!         upload and one download.  The host calls the two routines where it reads or writes the fields on its side (before the
!!         thermodynamics / diagnostics / save_restart, and after them).
!!   * the restart fields of register_restarts_dyn_split_RK2 (sfc, u2, v2, h2, CAu, CAv, diffu, diffv, and ubtav / vbtav / DTBT of the
!!     barotropic module) are host arrays of this control structure that dyn_split_RK2_sync_to_host refreshes; fields found in a
!!     restart file are uploaded at initialisation (query_initialized).
!!
!! Provided: what mom6hip_step_dyn_split_rk2 provides (include/mom6hip.h); refused by name with a FATAL error: BEGW /= 0,
!! SPLIT_BOTTOM_STRESS, FPMIX, TIDES / CALCULATE_SAL, porous barriers, Stokes PGF, STOCH, and everything the sub-modules' shims refuse.
!!
!! Compiled INSIDE a MOM6 source tree in place of src/core/MOM_dynamics_split_RK2.F90 (together with the other *_hip.F90 shims);
!! here against tests/fortran/stubs.
module MOM_dynamics_split_RK2

use, intrinsic :: iso_c_binding
use mom6hip_c_api
use mom6hip_MOM_glue,          only : mom6hip_shared_context, mom6hip_fatal_if, mom6hip_read_resident, mom6hip_mirror
use mom6hip_MOM_glue,          only : mom6hip_mirrors_stage, mom6hip_mirrors_host_was_modified, mom6hip_mirrors_end, mom6hip_obc_to_c
use MOM_variables,             only : vertvisc_type, thermo_var_ptrs, porous_barrier_type
use MOM_variables,             only : BT_cont_type, alloc_BT_cont_type
use MOM_variables,             only : accel_diag_ptrs, ocean_internal_state, cont_diag_ptrs
use MOM_forcing_type,          only : mech_forcing
use MOM_diag_mediator,         only : diag_ctrl
use MOM_error_handler,         only : MOM_error, FATAL, WARNING
use MOM_file_parser,           only : get_param, log_version, param_file_type
use MOM_get_input,             only : directories
use MOM_restart,               only : register_restart_field, query_initialized, MOM_restart_CS, is_new_run
use MOM_time_manager,          only : time_type
use MOM_ALE,                   only : ALE_CS, ALE_remap_velocities
use MOM_domains,               only : pass_vector
use MOM_barotropic,            only : barotropic_init, register_barotropic_restarts, barotropic_CS, barotropic_end
use MOM_barotropic,            only : barotropic_hip_struct, barotropic_hip_update
use MOM_boundary_update,       only : update_OBC_CS, update_OBC_data
use MOM_diabatic_driver,        only : diabatic_CS
use MOM_continuity_PPM,        only : continuity_PPM_CS, continuity_PPM_init, continuity_PPM_stencil, continuity_PPM_hip_struct
use MOM_CoriolisAdv,           only : CoriolisAdv_CS, CoriolisAdv_init, CoriolisAdv_end, CoriolisAdv_hip_struct
use MOM_grid,                  only : ocean_grid_type
use MOM_hor_index,             only : hor_index_type
use MOM_hor_visc,              only : hor_visc_CS, hor_visc_init, hor_visc_end, hor_visc_hip_struct
use MOM_lateral_mixing_coeffs, only : VarMix_CS
use MOM_MEKE_types,            only : MEKE_type
use MOM_open_boundary,         only : ocean_OBC_type, OBC_segment_type, update_OBC_ramp
use MOM_PressureForce_FV,      only : PressureForce_FV_CS, PressureForce_FV_init, PressureForce_FV_hip_struct
use MOM_set_visc,              only : set_visc_CS, set_visc_hip_struct
use MOM_stochastics,           only : stochastic_CS
use MOM_thickness_diffuse,     only : thickness_diffuse_CS
use MOM_unit_scaling,          only : unit_scale_type
use MOM_vert_friction,         only : vertvisc_init, vertvisc_end, vertvisc_CS, vertvisc_hip_struct, vertvisc_hip_add_ntrunc
use MOM_verticalGrid,          only : verticalGrid_type
use MOM_wave_interface,        only : wave_parameters_CS

implicit none ; private

#include <MOM_memory.h>

public step_MOM_dyn_split_RK2, register_restarts_dyn_split_RK2, initialize_dyn_split_RK2
public remap_dyn_split_RK2_aux_vars, end_dyn_split_RK2, init_dyn_split_RK2_diabatic
public dyn_split_RK2_sync_to_host, dyn_split_RK2_host_was_modified      ! the two calls of GPU_RESIDENT_DYNAMICS = True

!> MOM_dynamics_split_RK2 module control structure (the reference's :84-268: what outlives a call)
type, public :: MOM_dyn_split_RK2_CS ; private
  ! the restart fields on the host, refreshed by dyn_split_RK2_sync_to_host (register_restarts_dyn_split_RK2 :1210-1269)
  real, allocatable, dimension(:,:,:) :: CAu_pred, CAv_pred, u_av, v_av, h_av, diffu, diffv
  real, allocatable, dimension(:,:)   :: eta
  real, pointer, dimension(:,:,:) :: uh_rst => NULL(), vh_rst => NULL()      !< the uh, vh handed to register_restarts (restart fields)
  real :: be = 0.6, begw = 0.0
  logical :: BT_use_layer_fluxes = .true., store_CAu = .true., split_bottom_stress = .false., remap_aux = .false.
  logical :: resident = .false.                !< GPU_RESIDENT_DYNAMICS
  logical :: module_is_initialized = .false.
  ! the control structures of the modules the step calls (the reference's pointers :228-262)
  type(continuity_PPM_CS)   :: continuity_CSp
  type(CoriolisAdv_CS)      :: CoriolisAdv
  type(PressureForce_FV_CS) :: PressureForce_CSp
  type(hor_visc_CS)         :: hor_visc
  type(vertvisc_CS), pointer :: vertvisc_CSp => NULL()
  type(barotropic_CS)       :: barotropic_CSp
  type(set_visc_CS), pointer :: set_visc_CSp => NULL()
  type(BT_cont_type), pointer :: BT_cont => NULL()
  type(ALE_CS), pointer :: ALE_CSp => NULL()
  type(diag_ctrl), pointer :: diag => NULL()
  ! ---- the device side ----
  type(c_ptr) :: ctx = c_null_ptr
  type(mom6hip_continuity_cs_t)    :: c_cont
  type(mom6hip_coriolisadv_cs_t)   :: c_cor
  type(mom6hip_pressureforce_cs_t) :: c_pf
  type(mom6hip_eos_t)              :: c_eos, c_sv_eos
  type(mom6hip_barotropic_cs_t)    :: c_bt, h_bt      !< (h_bt: the host arrays of barotropic_CS, for ubtav / vbtav)
  type(mom6hip_bt_cont_t)          :: c_btc
  type(mom6hip_vertvisc_cs_t)      :: c_vv
  type(mom6hip_vertvisc_type_t)    :: c_visc
  type(mom6hip_hor_visc_cs_t)      :: c_hv
  type(mom6hip_set_visc_cs_t)      :: c_sv
  type(mom6hip_dyn_split_rk2_cs_t) :: c_rk2
  type(ocean_OBC_type), pointer :: OBC => NULL()     !< CS%OBC (:253)
  type(update_OBC_CS), pointer :: update_OBC_CSp => NULL()      !< (:258)
  type(mom6hip_obc_t) :: c_obc                       !< the library's view of it: device mirrors of the segments' arrays
  type(mom6hip_obc_segment_t), allocatable :: c_obc_segs(:)
  logical :: use_EOS = .true., use_BT_cont = .true.
  integer(c_int64_t) :: nh2 = 0, nu2 = 0, nv2 = 0, nq2 = 0, nh3 = 0, nu3 = 0, nv3 = 0      !< array sizes in doubles
  type(c_ptr) :: d_owned(96) = c_null_ptr      !< what end_dyn_split_RK2 frees
  integer :: n_owned = 0
end type MOM_dyn_split_RK2_CS

contains

! ------------------------------------------------------------------------------------------------------------------------
! device memory and mirrors
! ------------------------------------------------------------------------------------------------------------------------

!> n zeroed doubles in HBM, owned by the control structure
function dalloc(CS, n) result(p)
  type(MOM_dyn_split_RK2_CS), intent(inout) :: CS
  integer(c_int64_t),         intent(in)    :: n
  type(c_ptr) :: p
  integer :: rc
  rc = mom6hip_malloc(p, 8_c_int64_t*max(n, 1_c_int64_t))
  if (rc == 0) rc = mom6hip_memset_zero(CS%ctx, p, 8_c_int64_t*max(n, 1_c_int64_t))
  call mom6hip_fatal_if(rc, "MOM_dynamics_split_RK2 (device allocation)")
  if (CS%n_owned >= size(CS%d_owned)) call MOM_error(FATAL, "MOM_dynamics_split_RK2 (HIP): too many device arrays.")
  CS%n_owned = CS%n_owned + 1 ; CS%d_owned(CS%n_owned) = p
end function dalloc

!> a device copy of n host doubles, owned by the control structure
function dput(CS, hp, n) result(p)
  type(MOM_dyn_split_RK2_CS), intent(inout) :: CS
  type(c_ptr),                intent(in)    :: hp
  integer(c_int64_t),         intent(in)    :: n
  type(c_ptr) :: p
  integer :: rc
  p = dalloc(CS, n)
  rc = mom6hip_sync_to_device(CS%ctx, p, hp, 8_c_int64_t*n)
  call mom6hip_fatal_if(rc, "MOM_dynamics_split_RK2 (upload)")
end function dput

!> The device mirror of the host array at hp: the registry all shims share (mom6hip_MOM_glue)
function mirror(CS, hp, n, is_input, written) result(d)
  type(MOM_dyn_split_RK2_CS), intent(inout) :: CS
  type(c_ptr),                intent(in)    :: hp
  integer(c_int64_t),         intent(in)    :: n
  logical,                    intent(in)    :: is_input, written
  type(c_ptr) :: d
  d = mom6hip_mirror(CS%ctx, hp, n, is_input, written)
end function mirror

!> (GPU path only) Copy everything the device holds newer than the host back to the host arrays: the step's outputs (u, v, h, uh, vh,
!! uhtr, vhtr, eta_av, visc%nkml_visc_u/v) and the restart fields of this control structure and of the barotropic one.  Staged:
!! snapshots on the compute stream, copies on the copy stream, one wait.  Call it where the host reads these fields.
subroutine dyn_split_RK2_sync_to_host(CS)
  type(MOM_dyn_split_RK2_CS), pointer :: CS
  integer :: m, rc
  if (.not.associated(CS)) return
  if (.not.CS%module_is_initialized) return
  call mom6hip_mirrors_stage(CS%ctx)
  call restart_fields(CS, to_host=.true.)
  rc = mom6hip_stage_wait(CS%ctx) ; call mom6hip_fatal_if(rc, "dyn_split_RK2_sync_to_host")
end subroutine dyn_split_RK2_sync_to_host

!> (GPU path only) The host has changed the prognostic fields (thermodynamics, ALE remapping, a new forcing ...): the next step uploads
!! every input again.  With GPU_RESIDENT_DYNAMICS = False this is what every step assumes anyway.
subroutine dyn_split_RK2_host_was_modified(CS)
  type(MOM_dyn_split_RK2_CS), pointer :: CS
  integer :: m
  if (.not.associated(CS)) return
  call mom6hip_mirrors_host_was_modified()
end subroutine dyn_split_RK2_host_was_modified

!> The restart fields between their host arrays in CS and the device arrays of the library's structs
subroutine restart_fields(CS, to_host)
  type(MOM_dyn_split_RK2_CS), target, intent(inout) :: CS
  logical, intent(in) :: to_host
  call one(c_loc(CS%eta), CS%c_rk2%eta, CS%nh2)
  call one(c_loc(CS%u_av), CS%c_rk2%u_av, CS%nu3) ; call one(c_loc(CS%v_av), CS%c_rk2%v_av, CS%nv3)
  call one(c_loc(CS%h_av), CS%c_rk2%h_av, CS%nh3)
  call one(c_loc(CS%CAu_pred), CS%c_rk2%CAu_pred, CS%nu3) ; call one(c_loc(CS%CAv_pred), CS%c_rk2%CAv_pred, CS%nv3)
  call one(c_loc(CS%diffu), CS%c_rk2%diffu, CS%nu3) ; call one(c_loc(CS%diffv), CS%c_rk2%diffv, CS%nv3)
  call one(CS%h_bt%ubtav, CS%c_bt%ubtav, CS%nu2) ; call one(CS%h_bt%vbtav, CS%c_bt%vbtav, CS%nv2)      ! barotropic_CS's restart fields
contains
  subroutine one(hp, dp, n)
    type(c_ptr), intent(in) :: hp, dp
    integer(c_int64_t), intent(in) :: n
    integer :: rc
    if (to_host) then ; rc = mom6hip_stage_to_host(CS%ctx, hp, dp, 8_c_int64_t*n)
    else ; rc = mom6hip_sync_to_device(CS%ctx, dp, hp, 8_c_int64_t*n) ; endif
    call mom6hip_fatal_if(rc, "MOM_dynamics_split_RK2 (restart fields)")
  end subroutine one
end subroutine restart_fields

! ------------------------------------------------------------------------------------------------------------------------
! the reference's public procedures
! ------------------------------------------------------------------------------------------------------------------------

!> Same interface as the reference step_MOM_dyn_split_RK2 (:289): one library call on the device mirrors.
subroutine step_MOM_dyn_split_RK2(u_inst, v_inst, h, tv, visc, Time_local, dt, forces, p_surf_begin, p_surf_end, &
                                  uh, vh, uhtr, vhtr, eta_av, G, GV, US, CS, calc_dtbt, VarMix, &
                                  MEKE, thickness_diffuse_CSp, pbv, STOCH, Waves)
  type(ocean_grid_type),             intent(inout) :: G
  type(verticalGrid_type),           intent(in)    :: GV
  type(unit_scale_type),             intent(in)    :: US
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(inout) :: u_inst
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(inout) :: v_inst
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  target, intent(inout) :: h
  type(thermo_var_ptrs),             intent(in)    :: tv
  type(vertvisc_type), target,       intent(inout) :: visc
  type(time_type),                   intent(in)    :: Time_local
  real,                              intent(in)    :: dt
  type(mech_forcing),                intent(in)    :: forces
  real, dimension(:,:),              pointer       :: p_surf_begin
  real, dimension(:,:),              pointer       :: p_surf_end
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(inout) :: uh
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(inout) :: vh
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(inout) :: uhtr
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(inout) :: vhtr
  real, dimension(SZI_(G),SZJ_(G)),  target, intent(out)   :: eta_av
  type(MOM_dyn_split_RK2_CS),        pointer       :: CS
  logical,                           intent(in)    :: calc_dtbt
  type(VarMix_CS),                   intent(inout) :: VarMix
  type(MEKE_type), target,           intent(inout) :: MEKE
  type(thickness_diffuse_CS),        intent(inout) :: thickness_diffuse_CSp
  type(porous_barrier_type),         intent(in)    :: pbv
  type(stochastic_CS),               intent(inout) :: STOCH      ! not optional in the reference either (:331; MOM.F90:1242-1245)
  type(wave_parameters_CS), optional, pointer      :: Waves

  type(c_ptr) :: d_u, d_v, d_h, d_T, d_S, d_uh, d_vh, d_uhtr, d_vhtr, d_eta_av, d_tx, d_ty
  integer(c_int64_t) :: ntr
  integer :: rc

  if (.not.associated(CS)) call MOM_error(FATAL, "step_MOM_dyn_split_RK2: Module must be initialized before it is used.")
  if (.not.CS%module_is_initialized) call MOM_error(FATAL, "step_MOM_dyn_split_RK2: Module must be initialized before it is used.")
  ! (the surface pressures -- p_surf_end when both pointers are associated, with eta_PF interpolated between the two, else forces%p_surf
  ! :435-442 -- go to the library with the other arguments below)
  ! Waves: the Stokes pressure force (Waves%Stokes_PGF, :505-520) and the Stokes terms of CorAdCalc / vertFPmix
  if (present(Waves)) then ; if (associated(Waves)) then
    if (Waves%Stokes_PGF .or. Waves%Stokes_VF) call refuse("surface waves (Waves%Stokes_PGF / Waves%Stokes_VF)")
  endif ; endif
  if (allocated(pbv%por_face_areaU)) then
    if (any(pbv%por_face_areaU /= 1.0) .or. any(pbv%por_face_areaV /= 1.0)) call refuse("porous barriers")
  endif
  ! VarMix: the resolution function scales the Laplacian viscosity only (MOM_hor_visc.F90:474-476, :1123, :1525)
  if (VarMix%use_variable_mixing .and. VarMix%Resoln_scaled_Kh .and. (CS%c_hv%Laplacian /= 0)) call refuse("RESOLN_SCALED_KH with LAPLACIAN")
  ! STOCH reaches horizontal_viscosity only (:863), which reads it for the SKEB amplitude (MOM_hor_visc.F90: STOCH%skeb_use_frict)
  if (STOCH%do_skeb .and. STOCH%skeb_use_frict) call refuse("the stochastic kinetic-energy backscatter (DO_SKEB with SKEB_USE_FRICT)")
  if (.not.(associated(forces%taux) .and. associated(forces%tauy))) call MOM_error(FATAL, "step_MOM_dyn_split_RK2 (HIP): "// &
      "forces%taux and forces%tauy must be associated.")
  if (CS%use_EOS .and. .not.(associated(tv%T) .and. associated(tv%S))) call MOM_error(FATAL, "step_MOM_dyn_split_RK2 (HIP): "// &
      "an equation of state needs tv%T and tv%S.")

  ! ---- the device mirrors of the arguments
  d_u = mirror(CS, c_loc(u_inst), CS%nu3, .true., .true.) ; d_v = mirror(CS, c_loc(v_inst), CS%nv3, .true., .true.)
  d_h = mirror(CS, c_loc(h), CS%nh3, .true., .true.)
  d_T = c_null_ptr ; d_S = c_null_ptr
  if (CS%use_EOS) then
    d_T = mirror(CS, c_loc(tv%T), CS%nh3, .true., .false.) ; d_S = mirror(CS, c_loc(tv%S), CS%nh3, .true., .false.)
  endif
  d_uh = mirror(CS, c_loc(uh), CS%nu3, .true., .true.) ; d_vh = mirror(CS, c_loc(vh), CS%nv3, .true., .true.)
  d_uhtr = mirror(CS, c_loc(uhtr), CS%nu3, .true., .true.) ; d_vhtr = mirror(CS, c_loc(vhtr), CS%nv3, .true., .true.)
  d_eta_av = mirror(CS, c_loc(eta_av), CS%nh2, .false., .true.)
  d_tx = mirror(CS, c_loc(forces%taux), CS%nu2, .true., .false.) ; d_ty = mirror(CS, c_loc(forces%tauy), CS%nv2, .true., .false.)
  call visc_mirrors(CS, visc, forces)
  ! the surface pressures :435-442 (the library takes p_surf_end and interpolates eta_PF when both are there, else forces%p_surf)
  CS%c_rk2%p_surf_begin = c_null_ptr ; CS%c_rk2%p_surf_end = c_null_ptr ; CS%c_rk2%p_surf = c_null_ptr
  if (associated(p_surf_begin)) CS%c_rk2%p_surf_begin = mirror(CS, c_loc(p_surf_begin), CS%nh2, .true., .false.)
  if (associated(p_surf_end)) CS%c_rk2%p_surf_end = mirror(CS, c_loc(p_surf_end), CS%nh2, .true., .false.)
  if (associated(forces%p_surf)) CS%c_rk2%p_surf = mirror(CS, c_loc(forces%p_surf), CS%nh2, .true., .false.)
  ! the MEKE argument of horizontal_viscosity (MOM_hor_visc.F90:1141, :1318, :1537, :1634; :1833-1889)
  CS%c_hv%MEKE_Ku = c_null_ptr ; CS%c_hv%MEKE_Au = c_null_ptr ; CS%c_hv%MEKE_mom_src = c_null_ptr
  if (allocated(MEKE%Ku)) CS%c_hv%MEKE_Ku = mirror(CS, c_loc(MEKE%Ku), CS%nh2, .true., .false.)
  if (allocated(MEKE%Au)) CS%c_hv%MEKE_Au = mirror(CS, c_loc(MEKE%Au), CS%nh2, .true., .false.)
  if (allocated(MEKE%mom_src)) then
    if (MEKE%backscatter_Ro_c /= 0.0) call refuse("MEKE_BACKSCAT_RO_C /= 0")
    CS%c_hv%MEKE_mom_src = mirror(CS, c_loc(MEKE%mom_src), CS%nh2, .true., .true.)
    if (allocated(MEKE%GME_snk)) MEKE%GME_snk(:,:) = 0.0
  endif

  if (associated(CS%OBC)) then
    call update_OBC_ramp(Time_local, CS%OBC, US)      ! :448
    ! :534-536: the external data of the segments, by the reference's own MOM_boundary_update on the host.  It reads h (and tv), which the
    ! step does not change before its line :534, so calling it ahead of the step is the reference's order; staged, the segments' arrays
    ! are uploaded with every call; resident, the host's h may be stale and the mirrors of the segments would not see the new data.
    if (CS%OBC%update_OBC) then
      if (CS%resident) call refuse("OBC%update_OBC (update_OBC_data inside the step) with GPU_RESIDENT_DYNAMICS")
      call update_OBC_data(CS%OBC, G, GV, US, tv, h, CS%update_OBC_CSp, Time_local)
    endif
  endif
  call obc_mirrors(CS)
  rc = mom6hip_step_dyn_split_rk2(CS%ctx, CS%c_rk2, d_u, d_v, d_h, d_T, d_S, real(dt, c_double), d_tx, d_ty, &
                                  real(GV%Z_to_H / GV%Rho0, c_double), d_uh, d_vh, d_uhtr, d_vhtr, d_eta_av, &
                                  merge(1_c_int32_t, 0_c_int32_t, calc_dtbt))
  call mom6hip_fatal_if(rc, "step_MOM_dyn_split_RK2")
  call barotropic_hip_update(CS%barotropic_CSp, CS%c_bt)
  if (associated(CS%vertvisc_CSp)) then      ! the truncations the device counted
    ntr = CS%c_vv%ntrunc
    rc = mom6hip_vertvisc_ntrunc(CS%ctx, CS%c_vv) ; call mom6hip_fatal_if(rc, "step_MOM_dyn_split_RK2 (ntrunc)")
    call vertvisc_hip_add_ntrunc(CS%vertvisc_CSp, CS%c_vv%ntrunc - ntr)
  endif
  if (.not.CS%resident) call dyn_split_RK2_sync_to_host(CS)
contains
  subroutine refuse(what)
    character(len=*), intent(in) :: what
    call MOM_error(FATAL, "step_MOM_dyn_split_RK2 (HIP): "//what//" is not provided by the GPU path.")
  end subroutine refuse
end subroutine step_MOM_dyn_split_RK2

!> CS%OBC as the library's step reads it: the flags, segnum_u / segnum_v and the segments' ranges as mom6hip_obc_to_c states them (host
!! tables), the segments' own arrays and OBC%rx_normal / ry_normal as device mirrors (the step writes segment%normal_vel of the radiating
!! segments and the two rates).  A host routine that changes a segment's data between steps says so with mom6hip_mirror_host_changed.
subroutine obc_mirrors(CS)
  type(MOM_dyn_split_RK2_CS), target, intent(inout) :: CS
  type(OBC_segment_type), pointer :: seg
  integer :: n
  logical :: rad
  CS%c_rk2%OBC = c_null_ptr
  if (.not.associated(CS%OBC)) return
  call mom6hip_obc_to_c(CS%OBC, CS%c_obc, CS%c_obc_segs, int(CS%nu2), int(CS%nv2), "MOM_dynamics_split_RK2")
  do n = 1, CS%OBC%number_of_segments
    seg => CS%OBC%segment(n)
    if (.not.seg%on_pe) cycle
    rad = seg%radiation .or. seg%gradient .or. seg%nudged .or. seg%oblique
    CS%c_obc_segs(n)%normal_trans = c_null_ptr ; CS%c_obc_segs(n)%normal_vel = c_null_ptr ; CS%c_obc_segs(n)%normal_vel_bt = c_null_ptr
    CS%c_obc_segs(n)%SSH = c_null_ptr ; CS%c_obc_segs(n)%tangential_vel = c_null_ptr ; CS%c_obc_segs(n)%tangential_grad = c_null_ptr
    CS%c_obc_segs(n)%nudged_normal_vel = c_null_ptr ; CS%c_obc_segs(n)%nudged_tangential_vel = c_null_ptr
    CS%c_obc_segs(n)%nudged_tangential_grad = c_null_ptr
    if (allocated(seg%normal_trans)) CS%c_obc_segs(n)%normal_trans = &
        mirror(CS, c_loc(seg%normal_trans), int(size(seg%normal_trans), c_int64_t), .true., .false.)
    if (allocated(seg%normal_vel)) CS%c_obc_segs(n)%normal_vel = &
        mirror(CS, c_loc(seg%normal_vel), int(size(seg%normal_vel), c_int64_t), .true., rad)
    if (allocated(seg%normal_vel_bt)) CS%c_obc_segs(n)%normal_vel_bt = &
        mirror(CS, c_loc(seg%normal_vel_bt), int(size(seg%normal_vel_bt), c_int64_t), .true., .false.)
    if (allocated(seg%SSH)) CS%c_obc_segs(n)%SSH = mirror(CS, c_loc(seg%SSH), int(size(seg%SSH), c_int64_t), .true., .false.)
    ! (the tangential forms of the radiation write segment%tangential_vel / tangential_grad)
    if (allocated(seg%tangential_vel)) CS%c_obc_segs(n)%tangential_vel = &
        mirror(CS, c_loc(seg%tangential_vel), int(size(seg%tangential_vel), c_int64_t), .true., seg%radiation_tan .or. seg%nudged_tan .or. seg%oblique_tan)
    if (allocated(seg%tangential_grad)) CS%c_obc_segs(n)%tangential_grad = &
        mirror(CS, c_loc(seg%tangential_grad), int(size(seg%tangential_grad), c_int64_t), .true., seg%radiation_grad .or. seg%nudged_grad .or. seg%oblique_grad)
    if (allocated(seg%nudged_tangential_vel)) CS%c_obc_segs(n)%nudged_tangential_vel = &
        mirror(CS, c_loc(seg%nudged_tangential_vel), int(size(seg%nudged_tangential_vel), c_int64_t), .true., .false.)
    if (allocated(seg%nudged_tangential_grad)) CS%c_obc_segs(n)%nudged_tangential_grad = &
        mirror(CS, c_loc(seg%nudged_tangential_grad), int(size(seg%nudged_tangential_grad), c_int64_t), .true., .false.)
    if (allocated(seg%nudged_normal_vel)) CS%c_obc_segs(n)%nudged_normal_vel = &
        mirror(CS, c_loc(seg%nudged_normal_vel), int(size(seg%nudged_normal_vel), c_int64_t), .true., .false.)
  enddo
  CS%c_obc%gamma_uv = CS%OBC%gamma_uv ; CS%c_obc%rx_max = CS%OBC%rx_max
  CS%c_obc%rx_normal = c_null_ptr ; CS%c_obc%ry_normal = c_null_ptr
  if (allocated(CS%OBC%rx_normal)) CS%c_obc%rx_normal = mirror(CS, c_loc(CS%OBC%rx_normal), CS%nu3, .true., .true.)
  if (allocated(CS%OBC%ry_normal)) CS%c_obc%ry_normal = mirror(CS, c_loc(CS%OBC%ry_normal), CS%nv3, .true., .true.)
  ! what the oblique segments keep between steps
  CS%c_obc%rx_oblique_u = c_null_ptr ; CS%c_obc%ry_oblique_u = c_null_ptr ; CS%c_obc%cff_normal_u = c_null_ptr
  CS%c_obc%rx_oblique_v = c_null_ptr ; CS%c_obc%ry_oblique_v = c_null_ptr ; CS%c_obc%cff_normal_v = c_null_ptr
  if (allocated(CS%OBC%rx_oblique_u)) CS%c_obc%rx_oblique_u = mirror(CS, c_loc(CS%OBC%rx_oblique_u), CS%nu3, .true., .true.)
  if (allocated(CS%OBC%ry_oblique_u)) CS%c_obc%ry_oblique_u = mirror(CS, c_loc(CS%OBC%ry_oblique_u), CS%nu3, .true., .true.)
  if (allocated(CS%OBC%cff_normal_u)) CS%c_obc%cff_normal_u = mirror(CS, c_loc(CS%OBC%cff_normal_u), CS%nu3, .true., .true.)
  if (allocated(CS%OBC%rx_oblique_v)) CS%c_obc%rx_oblique_v = mirror(CS, c_loc(CS%OBC%rx_oblique_v), CS%nv3, .true., .true.)
  if (allocated(CS%OBC%ry_oblique_v)) CS%c_obc%ry_oblique_v = mirror(CS, c_loc(CS%OBC%ry_oblique_v), CS%nv3, .true., .true.)
  if (allocated(CS%OBC%cff_normal_v)) CS%c_obc%cff_normal_v = mirror(CS, c_loc(CS%OBC%cff_normal_v), CS%nv3, .true., .true.)
  CS%c_rk2%OBC = c_loc(CS%c_obc)
end subroutine obc_mirrors

!> The members of visc and forces%ustar that the step reads (or, for nkml_visc_u/v, writes), as device mirrors in the library's struct
subroutine visc_mirrors(CS, visc, forces)
  type(MOM_dyn_split_RK2_CS), target, intent(inout) :: CS
  type(vertvisc_type), target, intent(inout) :: visc
  type(mech_forcing),          intent(in)    :: forces
  logical :: dyn_ml
  CS%c_visc%Kv_bbl_u = c_null_ptr ; CS%c_visc%Kv_bbl_v = c_null_ptr ; CS%c_visc%bbl_thick_u = c_null_ptr ; CS%c_visc%bbl_thick_v = c_null_ptr
  CS%c_visc%Ray_u = c_null_ptr ; CS%c_visc%Ray_v = c_null_ptr ; CS%c_visc%Kv_shear = c_null_ptr ; CS%c_visc%Kv_shear_Bu = c_null_ptr
  CS%c_visc%nkml_visc_u = c_null_ptr ; CS%c_visc%nkml_visc_v = c_null_ptr ; CS%c_visc%ustar = c_null_ptr ; CS%c_visc%reserved(:) = c_null_ptr
  if (.not.associated(CS%vertvisc_CSp)) return
  if (allocated(visc%Kv_bbl_u)) CS%c_visc%Kv_bbl_u = mirror(CS, c_loc(visc%Kv_bbl_u), CS%nu2, .true., .false.)
  if (allocated(visc%Kv_bbl_v)) CS%c_visc%Kv_bbl_v = mirror(CS, c_loc(visc%Kv_bbl_v), CS%nv2, .true., .false.)
  if (allocated(visc%bbl_thick_u)) CS%c_visc%bbl_thick_u = mirror(CS, c_loc(visc%bbl_thick_u), CS%nu2, .true., .false.)
  if (allocated(visc%bbl_thick_v)) CS%c_visc%bbl_thick_v = mirror(CS, c_loc(visc%bbl_thick_v), CS%nv2, .true., .false.)
  if (allocated(visc%Ray_u)) CS%c_visc%Ray_u = mirror(CS, c_loc(visc%Ray_u), CS%nu3, .true., .false.)
  if (allocated(visc%Ray_v)) CS%c_visc%Ray_v = mirror(CS, c_loc(visc%Ray_v), CS%nv3, .true., .false.)
  if (associated(visc%Kv_shear)) CS%c_visc%Kv_shear = mirror(CS, c_loc(visc%Kv_shear), CS%nh3 + CS%nh2, .true., .false.)
  if (associated(visc%Kv_shear_Bu)) call MOM_error(FATAL, "step_MOM_dyn_split_RK2 (HIP): visc%Kv_shear_Bu is not provided by the GPU path.")
  dyn_ml = (CS%c_vv%dynamic_viscous_ML /= 0)
  if (dyn_ml .or. CS%c_vv%nkml > 0) then
    if (.not.associated(forces%ustar)) call MOM_error(FATAL, "step_MOM_dyn_split_RK2 (HIP): DYNAMIC_VISCOUS_ML / a bulk mixed "// &
        "layer needs forces%ustar (the GPU path is Boussinesq: find_ustar returns forces%ustar).")
    CS%c_visc%ustar = mirror(CS, c_loc(forces%ustar), CS%nh2, .true., .false.)
  endif
  if (dyn_ml) then
    if (.not.(allocated(visc%nkml_visc_u) .and. allocated(visc%nkml_visc_v))) call MOM_error(FATAL, "step_MOM_dyn_split_RK2 (HIP): "// &
        "visc%nkml_visc_u/v must be allocated with DYNAMIC_VISCOUS_ML (set_visc_init does it).")
    CS%c_visc%nkml_visc_u = mirror(CS, c_loc(visc%nkml_visc_u), CS%nu2, .true., .true.)
    CS%c_visc%nkml_visc_v = mirror(CS, c_loc(visc%nkml_visc_v), CS%nv2, .true., .true.)
  endif
end subroutine visc_mirrors

!> Same interface as the reference register_restarts_dyn_split_RK2 (:1181): allocates the control structure and its restart
!! fields on the host and registers them (sfc, u2, v2, h2, uh, vh, diffu, diffv, CAu, CAv; then the barotropic module's).
subroutine register_restarts_dyn_split_RK2(HI, GV, US, param_file, CS, restart_CS, uh, vh)
  type(hor_index_type),          intent(in)    :: HI
  type(verticalGrid_type),       intent(in)    :: GV
  type(unit_scale_type),         intent(in)    :: US
  type(param_file_type),         intent(in)    :: param_file
  type(MOM_dyn_split_RK2_CS),    pointer       :: CS
  type(MOM_restart_CS),          intent(inout) :: restart_CS
  real, dimension(SZIB_(HI),SZJ_(HI),SZK_(GV)), target, intent(inout) :: uh
  real, dimension(SZI_(HI),SZJB_(HI),SZK_(GV)), target, intent(inout) :: vh
  integer :: isd, ied, jsd, jed, nz, IsdB, IedB, JsdB, JedB

  isd = HI%isd ; ied = HI%ied ; jsd = HI%jsd ; jed = HI%jed ; nz = GV%ke
  IsdB = HI%IsdB ; IedB = HI%IedB ; JsdB = HI%JsdB ; JedB = HI%JedB
  if (associated(CS)) then
    call MOM_error(WARNING, "register_restarts_dyn_split_RK2 called with an associated control structure.")
    return
  endif
  allocate(CS)
  allocate(CS%diffu(IsdB:IedB,jsd:jed,nz), source=0.0) ; allocate(CS%diffv(isd:ied,JsdB:JedB,nz), source=0.0)
  allocate(CS%CAu_pred(IsdB:IedB,jsd:jed,nz), source=0.0) ; allocate(CS%CAv_pred(isd:ied,JsdB:JedB,nz), source=0.0)
  allocate(CS%eta(isd:ied,jsd:jed), source=0.0)
  allocate(CS%u_av(IsdB:IedB,jsd:jed,nz), source=0.0) ; allocate(CS%v_av(isd:ied,JsdB:JedB,nz), source=0.0)
  allocate(CS%h_av(isd:ied,jsd:jed,nz), source=0.0)
  CS%uh_rst => uh ; CS%vh_rst => vh

  call register_restart_field(CS%eta, "sfc", .false., restart_CS, longname="Free surface Height", units="m")
  call register_restart_field(CS%u_av, "u2", .false., restart_CS, longname="Auxiliary Zonal velocity", units="m s-1", hor_grid="u")
  call register_restart_field(CS%v_av, "v2", .false., restart_CS, longname="Auxiliary Meridional velocity", units="m s-1", hor_grid="v")
  call register_restart_field(CS%h_av, "h2", .false., restart_CS, longname="Auxiliary Layer Thickness", units="m")
  call register_restart_field(uh, "uh", .false., restart_CS, longname="Zonal thickness flux", units="m3 s-1", hor_grid="u")
  call register_restart_field(vh, "vh", .false., restart_CS, longname="Meridional thickness flux", units="m3 s-1", hor_grid="v")
  call register_restart_field(CS%diffu, "diffu", .false., restart_CS, longname="Zonal horizontal viscous acceleration", units="m s-2", &
                              hor_grid="u")
  call register_restart_field(CS%diffv, "diffv", .false., restart_CS, longname="Meridional horizontal viscous acceleration", &
                              units="m s-2", hor_grid="v")
  call get_param(param_file, "MOM_dynamics_split_RK2", "STORE_CORIOLIS_ACCEL", CS%store_CAu, &
                 "If true, calculate the Coriolis accelerations at the end of each timestep for use in the predictor step of the "// &
                 "next split RK2 timestep.", default=.true., do_not_log=.true.)
  if (CS%store_CAu) then
    call register_restart_field(CS%CAu_pred, "CAu", .false., restart_CS, longname="Zonal Coriolis and advactive acceleration", &
                                units="m s-2", hor_grid="u")
    call register_restart_field(CS%CAv_pred, "CAv", .false., restart_CS, longname="Meridional Coriolis and advactive acceleration", &
                                units="m s-2", hor_grid="v")
  endif
  call register_barotropic_restarts(HI, GV, US, param_file, CS%barotropic_CSp, restart_CS)
end subroutine register_restarts_dyn_split_RK2

!> Same interface as the reference remap_dyn_split_RK2_aux_vars (:1273) (REMAP_AUXILIARY_VARS = True): the restart fields u_av, v_av, CAu_pred,
!! CAv_pred (with STORE_CORIOLIS_ACCEL) and diffu, diffv are brought to the host, remapped by ALE_remap_velocities (the MOM_ALE shim: on the GPU,
!! HOST memspace) with the reference's two passes, and handed back to the device arrays of the library's control structure.
subroutine remap_dyn_split_RK2_aux_vars(G, GV, CS, h_old_u, h_old_v, h_new_u, h_new_v, ALE_CSp)
  type(ocean_grid_type),            intent(inout) :: G
  type(verticalGrid_type),          intent(in)    :: GV
  type(MOM_dyn_split_RK2_CS),       pointer       :: CS
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in) :: h_old_u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in) :: h_old_v
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in) :: h_new_u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in) :: h_new_v
  type(ALE_CS),                     pointer       :: ALE_CSp
  integer :: rc
  if (.not.CS%remap_aux) return
  call restart_fields(CS, to_host=.true.)
  rc = mom6hip_stage_wait(CS%ctx) ; call mom6hip_fatal_if(rc, "remap_dyn_split_RK2_aux_vars")
  if (CS%store_CAu) then
    call ALE_remap_velocities(ALE_CSp, G, GV, h_old_u, h_old_v, h_new_u, h_new_v, CS%u_av, CS%v_av)
    call pass_vector(CS%u_av, CS%v_av, G%Domain, complete=.false.)
    call ALE_remap_velocities(ALE_CSp, G, GV, h_old_u, h_old_v, h_new_u, h_new_v, CS%CAu_pred, CS%CAv_pred)
    call pass_vector(CS%CAu_pred, CS%CAv_pred, G%Domain, complete=.true.)
  endif
  call ALE_remap_velocities(ALE_CSp, G, GV, h_old_u, h_old_v, h_new_u, h_new_v, CS%diffu, CS%diffv)
  call restart_fields(CS, to_host=.false.)
end subroutine remap_dyn_split_RK2_aux_vars

!> Same interface as the reference initialize_dyn_split_RK2 (:1317), same parameters and defaults: the control structures of the
!! modules the step calls, the device-resident state, and the first values initialize_dyn_split_RK2 computes (eta from the layer
!! thicknesses :1521-1535, u_av / v_av / h_av :1552-1609, diffu / diffv :1543, the stored Coriolis terms :1560-1588) or takes
!! from the restart file.
subroutine initialize_dyn_split_RK2(u, v, h, tv, uh, vh, eta, Time, G, GV, US, param_file, &
                      diag, CS, restart_CS, dt, Accel_diag, Cont_diag, MIS, &
                      VarMix, MEKE, thickness_diffuse_CSp,                  &
                      OBC, update_OBC_CSp, ALE_CSp, set_visc, &
                      visc, dirs, ntrunc, pbv, calc_dtbt, cont_stencil)
  type(ocean_grid_type),            intent(inout) :: G
  type(verticalGrid_type), target,  intent(in)    :: GV
  type(unit_scale_type),            intent(in)    :: US
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(inout) :: u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(inout) :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  target, intent(inout) :: h
  type(thermo_var_ptrs),            intent(in)    :: tv
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(inout) :: uh
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(inout) :: vh
  real, dimension(SZI_(G),SZJ_(G)), intent(inout) :: eta
  type(time_type),          target, intent(in)    :: Time
  type(param_file_type),            intent(in)    :: param_file
  type(diag_ctrl),          target, intent(inout) :: diag
  type(MOM_dyn_split_RK2_CS),       pointer       :: CS
  type(MOM_restart_CS),             intent(inout) :: restart_CS
  real,                             intent(in)    :: dt
  type(accel_diag_ptrs),    target, intent(inout) :: Accel_diag
  type(cont_diag_ptrs),     target, intent(inout) :: Cont_diag
  type(ocean_internal_state),       intent(inout) :: MIS
  type(VarMix_CS),                  intent(inout) :: VarMix
  type(MEKE_type), target,          intent(inout) :: MEKE
  type(thickness_diffuse_CS),       intent(inout) :: thickness_diffuse_CSp
  type(ocean_OBC_type),             pointer       :: OBC
  type(update_OBC_CS),              pointer       :: update_OBC_CSp
  type(ALE_CS),                     pointer       :: ALE_CSp
  type(set_visc_CS),        target, intent(in)    :: set_visc
  type(vertvisc_type),              intent(inout) :: visc
  type(directories),                intent(in)    :: dirs
  integer, target,                  intent(inout) :: ntrunc
  type(porous_barrier_type),        intent(in)    :: pbv
  logical,                          intent(out)   :: calc_dtbt
  integer,                          intent(out)   :: cont_stencil
# include "version_variable.h"
  character(len=40) :: mdl = "MOM_dynamics_split_RK2"
  logical :: flag, use_tides
  integer :: i, j, k, is, ie, js, je, isd, ied, jsd, jed, nz, rc
  type(c_ptr) :: d_u, d_v, d_h, d_uh, d_vh
  type(mom6hip_barotropic_cs_t) :: hb
  type(mom6hip_vertvisc_cs_t) :: hvv
  type(mom6hip_hor_visc_cs_t) :: hhv
  type(set_visc_CS), pointer :: sv_p

  is = G%isc ; ie = G%iec ; js = G%jsc ; je = G%jec ; nz = GV%ke
  isd = G%isd ; ied = G%ied ; jsd = G%jsd ; jed = G%jed
  if (.not.associated(CS)) call MOM_error(FATAL, "initialize_dyn_split_RK2 called with an unassociated control structure.")
  if (CS%module_is_initialized) then
    call MOM_error(WARNING, "initialize_dyn_split_RK2 called with a control structure that has already been initialized.")
    return
  endif
  CS%module_is_initialized = .true.
  CS%diag => diag
  if (associated(OBC)) then      ! :1516-1519.  The segments' data are the host's business (MOM_open_boundary, MOM_boundary_update)
    CS%OBC => OBC
    if (associated(update_OBC_CSp)) CS%update_OBC_CSp => update_OBC_CSp      ! :1520
    ! (OBC%update_OBC: update_OBC_data is the reference's own host routine, called before the step: see step_MOM_dyn_split_RK2)
    ! (OBC%ramp_value scales the external data in update_OBC_segment_data, a host routine of the reference's MOM_open_boundary)
    if (OBC%ramp) call update_OBC_ramp(Time, CS%OBC, US, activate=is_new_run(restart_CS))      ! :1518
  endif
  if (.not.GV%Boussinesq) call refuse(.true., "a non-Boussinesq vertical grid")

  call log_version(param_file, mdl, version, "")
  call get_param(param_file, mdl, "TIDES", use_tides, "If true, apply tidal momentum forcing.", default=.false.)
  call refuse(use_tides, "TIDES")
  call get_param(param_file, mdl, "CALCULATE_SAL", flag, "If true, calculate self-attraction and loading.", default=use_tides)
  call refuse(flag, "CALCULATE_SAL")
  call get_param(param_file, mdl, "BE", CS%be, &
                 "If SPLIT is true, BE determines the relative weighting of a 2nd-order Runga-Kutta baroclinic time stepping scheme "// &
                 "(0.5) and a backward Euler scheme (1) that is used for the Coriolis and inertial terms.", units="nondim", default=0.6)
  call get_param(param_file, mdl, "BEGW", CS%begw, &
                 "If SPLIT is true, BEGW is a number from 0 to 1 that controls the extent to which the treatment of gravity waves is "// &
                 "forward-backward (0) or simulated backward Euler (1).", units="nondim", default=0.0)
  call refuse(CS%begw /= 0.0, "BEGW /= 0")
  call get_param(param_file, mdl, "SPLIT_BOTTOM_STRESS", CS%split_bottom_stress, &
                 "If true, provide the bottom stress calculated by the vertical viscosity to the barotropic solver.", default=.false.)
  call refuse(CS%split_bottom_stress, "SPLIT_BOTTOM_STRESS")
  call get_param(param_file, mdl, "REMAP_AUXILIARY_VARS", CS%remap_aux, &
                 "If true, apply ALE remapping to all of the auxiliary 3-dimensional variables that are needed to reproduce across "// &
                 "restarts, similarly to what is already being done with the primary state variables.", default=.false., do_not_log=.true.)
  call get_param(param_file, mdl, "BT_USE_LAYER_FLUXES", CS%BT_use_layer_fluxes, &
                 "If true, use the summed layered fluxes plus an adjustment due to the change in the barotropic velocity in the "// &
                 "barotropic continuity equation.", default=.true.)
  call get_param(param_file, mdl, "STORE_CORIOLIS_ACCEL", CS%store_CAu, &
                 "If true, calculate the Coriolis accelerations at the end of each timestep for use in the predictor step of the "// &
                 "next split RK2 timestep.", default=.true.)
  if (CS%remap_aux .and. .not.CS%store_CAu) call MOM_error(FATAL, "REMAP_AUXILIARY_VARS requires that STORE_CORIOLIS_ACCEL = True.")      ! :1439
  call get_param(param_file, mdl, "FPMIX", flag, "If true, add non-local momentum flux increments and diffuse down the Eulerian gradient.", &
                 default=.false.)
  call refuse(flag, "FPMIX")
  call get_param(param_file, mdl, "VISC_REM_BUG", flag, default=.false., do_not_log=.true.)
  call refuse(flag, "VISC_REM_BUG")
  ! DEBUG_OBC (:1444): open_boundary_test_extern_h on h (:445) and open_boundary_zero_normal_flow on PFu, PFv (:537) in the step
  call get_param(param_file, mdl, "DEBUG_OBC", flag, default=.false., do_not_log=.true.)
  call refuse(flag .and. associated(OBC), "DEBUG_OBC with open boundaries")
  ! GPU_RESIDENT_DYNAMICS (GPU path): if true, the fields the split RK2 step (and the other shims) work on stay on the GPU between
  ! calls: the host uploads them only after dyn_split_RK2_host_was_modified and sees them only after dyn_split_RK2_sync_to_host.  If
  ! false, every step uploads its inputs and copies its outputs back.
  call mom6hip_read_resident(param_file, CS%resident)

  ! ---- the modules the step calls (:1490-1519)
  call continuity_PPM_init(Time, G, GV, US, param_file, diag, CS%continuity_CSp)
  cont_stencil = continuity_PPM_stencil(CS%continuity_CSp)
  call CoriolisAdv_init(Time, G, GV, US, param_file, diag, Accel_diag, CS%CoriolisAdv)
  call PressureForce_FV_init(Time, G, GV, US, param_file, diag, CS%PressureForce_CSp)
  call hor_visc_init(Time, G, GV, US, param_file, diag, CS%hor_visc, ADp=Accel_diag)
  call vertvisc_init(MIS, Time, G, GV, US, param_file, diag, Accel_diag, dirs, ntrunc, CS%vertvisc_CSp)
  sv_p => set_visc ; CS%set_visc_CSp => sv_p
  CS%ALE_CSp => ALE_CSp

  ! eta from the layer thicknesses unless the restart file had it (:1521-1535)
  if (.not.query_initialized(CS%eta, "sfc", restart_CS)) then
    do j=js,je ; do i=is,ie ; CS%eta(i,j) = -GV%Z_to_H * G%bathyT(i,j) ; enddo ; enddo
    do k=1,nz ; do j=js,je ; do i=is,ie ; CS%eta(i,j) = CS%eta(i,j) + h(i,j,k) ; enddo ; enddo ; enddo
  endif
  do j=js,je ; do i=is,ie ; eta(i,j) = CS%eta(i,j) ; enddo ; enddo

  call get_param(param_file, "MOM_barotropic", "USE_BT_CONT_TYPE", CS%use_BT_cont, default=.true., do_not_log=.true.)
  if (CS%use_BT_cont) call alloc_BT_cont_type(CS%BT_cont, isd, ied, jsd, jed, nz, alloc_faces=.true.)
  call barotropic_init(u, v, h, CS%eta, Time, G, GV, US, param_file, diag, CS%barotropic_CSp, restart_CS, calc_dtbt, CS%BT_cont)

  ! ---- the device side -------------------------------------------------------------------------------------------------------
  CS%ctx = mom6hip_shared_context(G, GV)
  CS%nh2 = int(size(G%bathyT), c_int64_t) ; CS%nu2 = int((ied-isd+2), c_int64_t) * int((jed-jsd+1), c_int64_t)
  CS%nv2 = int((ied-isd+1), c_int64_t) * int((jed-jsd+2), c_int64_t) ; CS%nq2 = int((ied-isd+2), c_int64_t) * int((jed-jsd+2), c_int64_t)
  CS%nh3 = CS%nh2*nz ; CS%nu3 = CS%nu2*nz ; CS%nv3 = CS%nv2*nz

  CS%c_cont = continuity_PPM_hip_struct(CS%continuity_CSp)
  CS%c_cor = CoriolisAdv_hip_struct(CS%CoriolisAdv)
  call PressureForce_FV_hip_struct(CS%PressureForce_CSp, G, GV, tv, ALE_CSp, CS%c_pf, CS%c_eos, CS%use_EOS)
  call set_visc_hip_struct(CS%set_visc_CSp, GV, CS%c_sv, CS%c_sv_eos)

  ! barotropic_CS: the scalars of the host structure, device arrays with the values barotropic_init left on the host
  hb = barotropic_hip_struct(CS%barotropic_CSp)
  CS%c_bt = hb ; CS%h_bt = hb
  CS%c_bt%frhatu = dput(CS, hb%frhatu, CS%nu3) ; CS%c_bt%frhatv = dput(CS, hb%frhatv, CS%nv3)
  CS%c_bt%eta_cor = dput(CS, hb%eta_cor, CS%nh2) ; CS%c_bt%IDatu = dput(CS, hb%IDatu, CS%nu2) ; CS%c_bt%IDatv = dput(CS, hb%IDatv, CS%nv2)
  CS%c_bt%ubtav = dput(CS, hb%ubtav, CS%nu2) ; CS%c_bt%vbtav = dput(CS, hb%vbtav, CS%nv2) ; CS%c_bt%q_D = dput(CS, hb%q_D, CS%nq2)
  CS%c_bt%D_u_Cor = dput(CS, hb%D_u_Cor, CS%nu2) ; CS%c_bt%D_v_Cor = dput(CS, hb%D_v_Cor, CS%nv2)
  CS%c_bt%reserved2(:) = c_null_ptr
  if (CS%use_BT_cont) then
    CS%c_btc%FA_u_W0 = dalloc(CS, CS%nu2) ; CS%c_btc%FA_u_WW = dalloc(CS, CS%nu2) ; CS%c_btc%FA_u_E0 = dalloc(CS, CS%nu2)
    CS%c_btc%FA_u_EE = dalloc(CS, CS%nu2) ; CS%c_btc%uBT_WW = dalloc(CS, CS%nu2) ; CS%c_btc%uBT_EE = dalloc(CS, CS%nu2)
    CS%c_btc%FA_v_S0 = dalloc(CS, CS%nv2) ; CS%c_btc%FA_v_SS = dalloc(CS, CS%nv2) ; CS%c_btc%FA_v_N0 = dalloc(CS, CS%nv2)
    CS%c_btc%FA_v_NN = dalloc(CS, CS%nv2) ; CS%c_btc%vBT_SS = dalloc(CS, CS%nv2) ; CS%c_btc%vBT_NN = dalloc(CS, CS%nv2)
    CS%c_btc%h_u = dalloc(CS, CS%nu3) ; CS%c_btc%h_v = dalloc(CS, CS%nv3)
  endif
  ! vertvisc_CS: device arrays a_u, a_v, h_u, h_v
  hvv = vertvisc_hip_struct(CS%vertvisc_CSp)
  CS%c_vv = hvv
  CS%c_vv%a_u = dalloc(CS, CS%nu3 + CS%nu2) ; CS%c_vv%a_v = dalloc(CS, CS%nv3 + CS%nv2)
  CS%c_vv%h_u = dalloc(CS, CS%nu3) ; CS%c_vv%h_v = dalloc(CS, CS%nv3) ; CS%c_vv%reserved1(:) = c_null_ptr ; CS%c_vv%ntrunc = 0
  ! hor_visc_CS: the 16 static arrays hor_visc_init has filled on the host
  hhv = hor_visc_hip_struct(CS%hor_visc)
  CS%c_hv = hhv
  CS%c_hv%Kh_bg_xx = dput(CS, hhv%Kh_bg_xx, CS%nh2) ; CS%c_hv%Kh_Max_xx = dput(CS, hhv%Kh_Max_xx, CS%nh2)
  CS%c_hv%Ah_bg_xx = dput(CS, hhv%Ah_bg_xx, CS%nh2) ; CS%c_hv%Ah_Max_xx = dput(CS, hhv%Ah_Max_xx, CS%nh2)
  CS%c_hv%Laplac2_const_xx = dput(CS, hhv%Laplac2_const_xx, CS%nh2) ; CS%c_hv%Biharm_const_xx = dput(CS, hhv%Biharm_const_xx, CS%nh2)
  CS%c_hv%Biharm_const2_xx = dput(CS, hhv%Biharm_const2_xx, CS%nh2) ; CS%c_hv%reduction_xx = dput(CS, hhv%reduction_xx, CS%nh2)
  CS%c_hv%Kh_bg_xy = dput(CS, hhv%Kh_bg_xy, CS%nq2) ; CS%c_hv%Kh_Max_xy = dput(CS, hhv%Kh_Max_xy, CS%nq2)
  CS%c_hv%Ah_bg_xy = dput(CS, hhv%Ah_bg_xy, CS%nq2) ; CS%c_hv%Ah_Max_xy = dput(CS, hhv%Ah_Max_xy, CS%nq2)
  CS%c_hv%Laplac2_const_xy = dput(CS, hhv%Laplac2_const_xy, CS%nq2) ; CS%c_hv%Biharm_const_xy = dput(CS, hhv%Biharm_const_xy, CS%nq2)
  CS%c_hv%Biharm_const2_xy = dput(CS, hhv%Biharm_const2_xy, CS%nq2) ; CS%c_hv%reduction_xy = dput(CS, hhv%reduction_xy, CS%nq2)
  CS%c_hv%reserved1(:) = c_null_ptr

  ! MOM_dyn_split_RK2_CS
  CS%c_rk2%be = CS%be ; CS%c_rk2%begw = CS%begw
  CS%c_rk2%BT_use_layer_fluxes = merge(1, 0, CS%BT_use_layer_fluxes) ; CS%c_rk2%store_CAu = merge(1, 0, CS%store_CAu)
  CS%c_rk2%CAu_pred_stored = 0 ; CS%c_rk2%split_bottom_stress = 0 ; CS%c_rk2%reserved0(:) = 0
  CS%c_rk2%hooks = c_null_ptr ; CS%c_rk2%OBC = c_null_ptr
  CS%c_rk2%CAu = dalloc(CS, CS%nu3) ; CS%c_rk2%CAv = dalloc(CS, CS%nv3) ; CS%c_rk2%CAu_pred = dalloc(CS, CS%nu3)
  CS%c_rk2%CAv_pred = dalloc(CS, CS%nv3) ; CS%c_rk2%PFu = dalloc(CS, CS%nu3) ; CS%c_rk2%PFv = dalloc(CS, CS%nv3)
  CS%c_rk2%diffu = dalloc(CS, CS%nu3) ; CS%c_rk2%diffv = dalloc(CS, CS%nv3) ; CS%c_rk2%visc_rem_u = dalloc(CS, CS%nu3)
  CS%c_rk2%visc_rem_v = dalloc(CS, CS%nv3) ; CS%c_rk2%u_accel_bt = dalloc(CS, CS%nu3) ; CS%c_rk2%v_accel_bt = dalloc(CS, CS%nv3)
  CS%c_rk2%u_av = dalloc(CS, CS%nu3) ; CS%c_rk2%v_av = dalloc(CS, CS%nv3) ; CS%c_rk2%h_av = dalloc(CS, CS%nh3)
  CS%c_rk2%pbce = dalloc(CS, CS%nh3) ; CS%c_rk2%eta = dalloc(CS, CS%nh2) ; CS%c_rk2%eta_PF = dalloc(CS, CS%nh2)
  CS%c_rk2%uhbt = dalloc(CS, CS%nu2) ; CS%c_rk2%vhbt = dalloc(CS, CS%nv2)
  CS%c_rk2%du_av_inst = c_null_ptr ; CS%c_rk2%dv_av_inst = c_null_ptr
  call bind_structs(CS)

  ! the state the first step needs (mom6hip_dyn_split_rk2_init: :1521-1622 for a cold start), then what the restart file had
  d_u = mirror(CS, c_loc(u), CS%nu3, .true., .false.) ; d_v = mirror(CS, c_loc(v), CS%nv3, .true., .false.)
  d_h = mirror(CS, c_loc(h), CS%nh3, .true., .false.)
  d_uh = mirror(CS, c_loc(uh), CS%nu3, .true., .true.) ; d_vh = mirror(CS, c_loc(vh), CS%nv3, .true., .true.)
  ! (horizontal_viscosity at :1543 gets MEKE: its viscosities count, its momentum source is the first step's business)
  if (allocated(MEKE%Ku)) CS%c_hv%MEKE_Ku = mirror(CS, c_loc(MEKE%Ku), CS%nh2, .true., .false.)
  if (allocated(MEKE%Au)) CS%c_hv%MEKE_Au = mirror(CS, c_loc(MEKE%Au), CS%nh2, .true., .false.)
  if (VarMix%use_variable_mixing .and. VarMix%Resoln_scaled_Kh .and. (CS%c_hv%Laplacian /= 0)) call refuse(.true., "RESOLN_SCALED_KH with LAPLACIAN")
  call obc_mirrors(CS)
  rc = mom6hip_dyn_split_rk2_init(CS%ctx, CS%c_rk2, d_u, d_v, d_h, d_uh, d_vh, real(dt, c_double))
  call mom6hip_fatal_if(rc, "initialize_dyn_split_RK2")
  call from_restart(c_loc(CS%eta), CS%c_rk2%eta, CS%nh2, query_initialized(CS%eta, "sfc", restart_CS))
  call from_restart(c_loc(CS%u_av), CS%c_rk2%u_av, CS%nu3, query_initialized(CS%u_av, "u2", restart_CS))
  call from_restart(c_loc(CS%v_av), CS%c_rk2%v_av, CS%nv3, query_initialized(CS%v_av, "v2", restart_CS))
  call from_restart(c_loc(CS%h_av), CS%c_rk2%h_av, CS%nh3, query_initialized(CS%h_av, "h2", restart_CS))
  call from_restart(c_loc(CS%diffu), CS%c_rk2%diffu, CS%nu3, query_initialized(CS%diffu, "diffu", restart_CS))
  call from_restart(c_loc(CS%diffv), CS%c_rk2%diffv, CS%nv3, query_initialized(CS%diffv, "diffv", restart_CS))
  if (CS%store_CAu) then
    flag = query_initialized(CS%CAu_pred, "CAu", restart_CS) .and. query_initialized(CS%CAv_pred, "CAv", restart_CS)
    call from_restart(c_loc(CS%CAu_pred), CS%c_rk2%CAu_pred, CS%nu3, flag)
    call from_restart(c_loc(CS%CAv_pred), CS%c_rk2%CAv_pred, CS%nv3, flag)
    if (flag) CS%c_rk2%CAu_pred_stored = 1      ! :1561-1563
  endif
  if (query_initialized(uh, "uh", restart_CS) .and. query_initialized(vh, "vh", restart_CS)) then      ! the restart's transports win
    rc = mom6hip_sync_to_device(CS%ctx, d_uh, c_loc(uh), 8_c_int64_t*CS%nu3) ; call mom6hip_fatal_if(rc, "initialize_dyn_split_RK2")
    rc = mom6hip_sync_to_device(CS%ctx, d_vh, c_loc(vh), 8_c_int64_t*CS%nv3) ; call mom6hip_fatal_if(rc, "initialize_dyn_split_RK2")
  endif
  call dyn_split_RK2_sync_to_host(CS)      ! uh, vh and the restart fields as the initialisation left them
contains
  subroutine refuse(on, name)
    logical,          intent(in) :: on
    character(len=*), intent(in) :: name
    if (on) call MOM_error(FATAL, "initialize_dyn_split_RK2 (HIP): "//name//" is not provided by the GPU path.")
  end subroutine refuse
  subroutine from_restart(hp, dp, n, found)
    type(c_ptr), intent(in) :: hp, dp
    integer(c_int64_t), intent(in) :: n
    logical, intent(in) :: found
    integer :: rc2
    if (.not.found) return
    rc2 = mom6hip_sync_to_device(CS%ctx, dp, hp, 8_c_int64_t*n) ; call mom6hip_fatal_if(rc2, "initialize_dyn_split_RK2 (restart)")
  end subroutine from_restart
end subroutine initialize_dyn_split_RK2

!> The pointers of the library's step structure to the other structures of this control structure (which may have moved)
subroutine bind_structs(CS)
  type(MOM_dyn_split_RK2_CS), target, intent(inout) :: CS
  CS%c_rk2%continuity_CSp = c_loc(CS%c_cont) ; CS%c_rk2%CoriolisAdv = c_loc(CS%c_cor) ; CS%c_rk2%PressureForce_CSp = c_loc(CS%c_pf)
  CS%c_rk2%eqn_of_state = c_null_ptr ; if (CS%use_EOS) CS%c_rk2%eqn_of_state = c_loc(CS%c_eos)
  CS%c_rk2%barotropic_CSp = c_loc(CS%c_bt)
  CS%c_rk2%BT_cont = c_null_ptr ; if (CS%use_BT_cont) CS%c_rk2%BT_cont = c_loc(CS%c_btc)
  CS%c_rk2%vertvisc_CSp = c_loc(CS%c_vv) ; CS%c_rk2%visc = c_loc(CS%c_visc) ; CS%c_rk2%hor_visc = c_loc(CS%c_hv)
  CS%c_rk2%set_visc_CSp = c_null_ptr ; if (CS%c_sv%dynamic_viscous_ML /= 0) CS%c_rk2%set_visc_CSp = c_loc(CS%c_sv)
end subroutine bind_structs

!> Same interface as the reference init_dyn_split_RK2_diabatic (:1306).  MOM.F90:3350-3352 calls it only with FPMIX = True (the
!! boundary-layer depths of KPP / ePBL for vertFPmix), which initialize_dyn_split_RK2 of this file has already refused by name.
subroutine init_dyn_split_RK2_diabatic(diabatic_CSp, CS)
  type(diabatic_CS),                intent(in) :: diabatic_CSp
  type(MOM_dyn_split_RK2_CS),       pointer    :: CS
  call MOM_error(FATAL, "init_dyn_split_RK2_diabatic (HIP): FPMIX (the non-local momentum mixing of vertFPmix with the boundary-layer "// &
                        "depths of KPP or ePBL) is not provided by the GPU path.")
end subroutine init_dyn_split_RK2_diabatic

!> Same interface as the reference end_dyn_split_RK2 (:1829)
subroutine end_dyn_split_RK2(CS)
  type(MOM_dyn_split_RK2_CS), pointer :: CS
  integer :: m, rc
  if (.not.associated(CS)) return
  call mom6hip_mirrors_end()
  do m = 1, CS%n_owned ; rc = mom6hip_free(CS%d_owned(m)) ; enddo
  call barotropic_end(CS%barotropic_CSp)
  if (associated(CS%vertvisc_CSp)) then ; call vertvisc_end(CS%vertvisc_CSp) ; deallocate(CS%vertvisc_CSp) ; endif
  call hor_visc_end(CS%hor_visc)
  call CoriolisAdv_end(CS%CoriolisAdv)
  if (allocated(CS%diffu)) deallocate(CS%diffu, CS%diffv, CS%CAu_pred, CS%CAv_pred, CS%eta, CS%u_av, CS%v_av, CS%h_av)
  deallocate(CS)
end subroutine end_dyn_split_RK2

end module MOM_dynamics_split_RK2
