!> Drop-in replacement for module MOM_barotropic (src/core/MOM_barotropic.F90): btstep (:423), btcalc (:3394),
!! bt_mass_source (:4318), set_dtbt (:2801), barotropic_init (:4376), barotropic_get_tav, barotropic_end and
!! register_barotropic_restarts (:5165) with the reference's dummy-argument lists, so MOM_dynamics_split_RK2.F90 compiles
!! unchanged.  The work is done by libmom6hip on host arrays (HOST memspace); the barotropic subcycle of one btstep call
!! is a single hipGraph launch on the GPU.  Options outside the library's scope (INTEGRAL_BT_CONTINUITY,
!! NONLINEAR_BT_CONTINUITY, BOUND_BT_CORRECTION without its BT_cont bounds, GRADUAL_BT_ICS, BT_NONLIN_STRESS,
!! DYNAMIC_SURFACE_PRESSURE, BT_LINEAR_WAVE_DRAG, CLIP_BT_VELOCITY, CALCULATE_SAL / TIDES, the old bracket bug, answer
!! dates before 2019, open boundaries, a non-Boussinesq vertical grid) stop in barotropic_init with a FATAL error.
!!
!! Compiled INSIDE a MOM6 source tree in place of src/core/MOM_barotropic.F90; here against tests/fortran/stubs.
module MOM_barotropic

use, intrinsic :: iso_c_binding
use mom6hip_c_api
use mom6hip_MOM_glue,   only : mom6hip_shared_context, mom6hip_read_topology, mom6hip_fatal_if, mom6hip_obc_to_c
use MOM_diag_mediator,  only : diag_ctrl, time_type
use MOM_error_handler,  only : MOM_error, MOM_mesg, FATAL, WARNING
use MOM_file_parser,    only : get_param, log_version, param_file_type
use MOM_forcing_type,   only : mech_forcing
use MOM_grid,           only : ocean_grid_type
use MOM_hor_index,      only : hor_index_type
use MOM_open_boundary,  only : ocean_OBC_type
use MOM_restart,        only : register_restart_field, query_initialized, MOM_restart_CS
use MOM_self_attr_load, only : SAL_CS
use MOM_unit_scaling,   only : unit_scale_type
use MOM_variables,      only : BT_cont_type, accel_diag_ptrs
use MOM_verticalGrid,   only : verticalGrid_type
implicit none ; private

#include <MOM_memory.h>

public btcalc, bt_mass_source, btstep, barotropic_init, barotropic_end
public register_barotropic_restarts, set_dtbt, barotropic_get_tav
public barotropic_hip_struct, barotropic_hip_update      ! (GPU path only) for MOM_dynamics_split_RK2

!> Control structure: the library's struct (run-time parameters, dtbt) and the arrays the reference keeps in barotropic_CS
!! (:104-332) that outlive a call
type, public :: barotropic_CS ; private
  logical :: module_is_initialized = .false.
  logical :: split = .true.
  type(mom6hip_barotropic_cs_t) :: st
  real, allocatable, dimension(:,:,:) :: frhatu, frhatv   !< fraction of the column in each layer at velocity points
  real, allocatable, dimension(:,:)   :: eta_cor, IDatu, IDatv, ubtav, vbtav, q_D, D_u_Cor, D_v_Cor
  real :: dtbt                                             !< restart variable DTBT (mirrors st%dtbt)
  type(diag_ctrl), pointer :: diag => NULL()
end type barotropic_CS

integer, parameter :: HARMONIC = 1, ARITHMETIC = 2, HYBRID = 3, FROM_BT_CONT = 4      ! the reference's (:334-338)

contains

!> Point the library's struct at the arrays of this control structure (they may have moved if CS was copied)
subroutine bind_arrays(CS)
  type(barotropic_CS), target, intent(inout) :: CS
  CS%st%frhatu = c_loc(CS%frhatu) ; CS%st%frhatv = c_loc(CS%frhatv) ; CS%st%eta_cor = c_loc(CS%eta_cor)
  CS%st%IDatu = c_loc(CS%IDatu) ; CS%st%IDatv = c_loc(CS%IDatv) ; CS%st%ubtav = c_loc(CS%ubtav) ; CS%st%vbtav = c_loc(CS%vbtav)
  CS%st%q_D = c_loc(CS%q_D) ; CS%st%D_u_Cor = c_loc(CS%D_u_Cor) ; CS%st%D_v_Cor = c_loc(CS%D_v_Cor)
  CS%st%reserved2(:) = c_null_ptr
end subroutine bind_arrays

!> (GPU path only) The library's struct of this control structure, its pointers bound to the HOST arrays of CS: the
!! device-resident step of MOM_dynamics_split_RK2 copies the scalars and gives the copy device arrays of its own.
function barotropic_hip_struct(CS) result(st)
  type(barotropic_CS), target, intent(inout) :: CS
  type(mom6hip_barotropic_cs_t) :: st
  if (.not.CS%module_is_initialized) call MOM_error(FATAL, "btstep: Module MOM_barotropic must be initialized before it is used.")
  call bind_arrays(CS)
  st = CS%st
end function barotropic_hip_struct

!> (GPU path only) What a device-resident step changed in the control structure: the barotropic time step (set_dtbt) and the
!! number of barotropic steps of the last call
subroutine barotropic_hip_update(CS, st)
  type(barotropic_CS), intent(inout) :: CS
  type(mom6hip_barotropic_cs_t), intent(in) :: st
  CS%st%dtbt = st%dtbt ; CS%st%dtbt_max = st%dtbt_max ; CS%st%nstep_last = st%nstep_last
  CS%dtbt = st%dtbt
end subroutine barotropic_hip_update

!> BT_cont_type as the library's struct of pointers
subroutine bt_cont_struct(BT_cont, cbt)
  type(BT_cont_type), target, intent(in)  :: BT_cont
  type(mom6hip_bt_cont_t),    intent(out) :: cbt
  cbt%FA_u_W0 = c_loc(BT_cont%FA_u_W0) ; cbt%FA_u_WW = c_loc(BT_cont%FA_u_WW)
  cbt%FA_u_E0 = c_loc(BT_cont%FA_u_E0) ; cbt%FA_u_EE = c_loc(BT_cont%FA_u_EE)
  cbt%uBT_WW = c_loc(BT_cont%uBT_WW) ; cbt%uBT_EE = c_loc(BT_cont%uBT_EE)
  cbt%FA_v_S0 = c_loc(BT_cont%FA_v_S0) ; cbt%FA_v_SS = c_loc(BT_cont%FA_v_SS)
  cbt%FA_v_N0 = c_loc(BT_cont%FA_v_N0) ; cbt%FA_v_NN = c_loc(BT_cont%FA_v_NN)
  cbt%vBT_SS = c_loc(BT_cont%vBT_SS) ; cbt%vBT_NN = c_loc(BT_cont%vBT_NN)
  cbt%h_u = c_null_ptr ; if (allocated(BT_cont%h_u)) cbt%h_u = c_loc(BT_cont%h_u)
  cbt%h_v = c_null_ptr ; if (allocated(BT_cont%h_v)) cbt%h_v = c_loc(BT_cont%h_v)
end subroutine bt_cont_struct

!> Same interface as the reference btstep (:423).
subroutine btstep(U_in, V_in, eta_in, dt, bc_accel_u, bc_accel_v, forces, pbce, &
                  eta_PF_in, U_Cor, V_Cor, accel_layer_u, accel_layer_v, &
                  eta_out, uhbtav, vhbtav, G, GV, US, CS, &
                  visc_rem_u, visc_rem_v, SpV_avg, ADp, OBC, BT_cont, eta_PF_start, &
                  taux_bot, tauy_bot, uh0, vh0, u_uh0, v_vh0, etaav)
  type(ocean_grid_type),   intent(inout) :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(unit_scale_type),   intent(in)    :: US
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(in)  :: U_in
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(in)  :: V_in
  real, dimension(SZI_(G),SZJ_(G)),           target, intent(in)  :: eta_in
  real,                                               intent(in)  :: dt
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(in)  :: bc_accel_u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(in)  :: bc_accel_v
  type(mech_forcing),                                 intent(in)  :: forces
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  target, intent(in)  :: pbce
  real, dimension(SZI_(G),SZJ_(G)),           target, intent(in)  :: eta_PF_in
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(in)  :: U_Cor
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(in)  :: V_Cor
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(out) :: accel_layer_u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(out) :: accel_layer_v
  real, dimension(SZI_(G),SZJ_(G)),           target, intent(out) :: eta_out
  real, dimension(SZIB_(G),SZJ_(G)),          target, intent(out) :: uhbtav
  real, dimension(SZI_(G),SZJB_(G)),          target, intent(out) :: vhbtav
  type(barotropic_CS),                        target, intent(inout) :: CS
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(in)  :: visc_rem_u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(in)  :: visc_rem_v
  real, dimension(SZI_(G),SZJ_(G)),                   intent(in)  :: SpV_avg
  type(accel_diag_ptrs),                      pointer    :: ADp
  type(ocean_OBC_type),                       pointer    :: OBC
  type(BT_cont_type),                         pointer    :: BT_cont
  real, dimension(:,:),                       pointer    :: eta_PF_start
  real, dimension(:,:),                       pointer    :: taux_bot
  real, dimension(:,:),                       pointer    :: tauy_bot
  real, dimension(:,:,:),                     pointer    :: uh0
  real, dimension(:,:,:),                     pointer    :: u_uh0
  real, dimension(:,:,:),                     pointer    :: vh0
  real, dimension(:,:,:),                     pointer    :: v_vh0
  real, dimension(SZI_(G),SZJ_(G)), target, optional, intent(out) :: etaav

  type(mom6hip_bt_cont_t), target :: cbt
  type(c_ptr) :: p_bt, p_pfs, p_txb, p_tyb, p_uh0, p_vh0, p_uuh0, p_vvh0, p_etaav
  type(mom6hip_obc_t) :: cobc
  type(mom6hip_obc_segment_t), allocatable, target :: csegs(:)
  integer :: rc

  if (.not.CS%module_is_initialized) call MOM_error(FATAL, "btstep: Module MOM_barotropic must be initialized before it is used.")
  if (.not.CS%split) return
  if (.not.(associated(forces%taux) .and. associated(forces%tauy))) &
    call MOM_error(FATAL, "btstep (HIP): forces%taux and forces%tauy must be associated.")
  call bind_arrays(CS)
  p_bt = c_null_ptr
  if (associated(BT_cont)) then ; call bt_cont_struct(BT_cont, cbt) ; p_bt = c_loc(cbt) ; endif
  p_pfs = c_null_ptr ; if (associated(eta_PF_start)) p_pfs = c_loc(eta_PF_start)
  p_txb = c_null_ptr ; if (associated(taux_bot)) p_txb = c_loc(taux_bot)
  p_tyb = c_null_ptr ; if (associated(tauy_bot)) p_tyb = c_loc(tauy_bot)
  p_uh0 = c_null_ptr ; if (associated(uh0)) p_uh0 = c_loc(uh0)
  p_vh0 = c_null_ptr ; if (associated(vh0)) p_vh0 = c_loc(vh0)
  p_uuh0 = c_null_ptr ; if (associated(u_uh0)) p_uuh0 = c_loc(u_uh0)
  p_vvh0 = c_null_ptr ; if (associated(v_vh0)) p_vvh0 = c_loc(v_vh0)
  p_etaav = c_null_ptr ; if (present(etaav)) p_etaav = c_loc(etaav)

  if (associated(OBC)) then      ! specified, Flather and gradient segments (set_up_BT_OBC :3172, apply_velocity_OBCs :2931, ...)
    call mom6hip_obc_to_c(OBC, cobc, csegs, size(uhbtav), size(vhbtav), "MOM_barotropic")
    rc = mom6hip_btstep_obc(mom6hip_shared_context(G, GV), CS%st, c_loc(U_in), c_loc(V_in), c_loc(eta_in), dt, c_loc(bc_accel_u), &
                      c_loc(bc_accel_v), c_loc(forces%taux), c_loc(forces%tauy), GV%RZ_to_H, c_loc(pbce), c_loc(eta_PF_in), &
                      c_loc(U_Cor), c_loc(V_Cor), c_loc(accel_layer_u), c_loc(accel_layer_v), c_loc(eta_out), c_loc(uhbtav), &
                      c_loc(vhbtav), c_loc(visc_rem_u), c_loc(visc_rem_v), p_bt, p_pfs, p_txb, p_tyb, p_uh0, p_vh0, p_uuh0, &
                      p_vvh0, p_etaav, cobc, MOM6HIP_MEM_HOST)
  else
    rc = mom6hip_btstep(mom6hip_shared_context(G, GV), CS%st, c_loc(U_in), c_loc(V_in), c_loc(eta_in), dt, c_loc(bc_accel_u), &
                      c_loc(bc_accel_v), c_loc(forces%taux), c_loc(forces%tauy), GV%RZ_to_H, c_loc(pbce), c_loc(eta_PF_in), &
                      c_loc(U_Cor), c_loc(V_Cor), c_loc(accel_layer_u), c_loc(accel_layer_v), c_loc(eta_out), c_loc(uhbtav), &
                      c_loc(vhbtav), c_loc(visc_rem_u), c_loc(visc_rem_v), p_bt, p_pfs, p_txb, p_tyb, p_uh0, p_vh0, p_uuh0, &
                      p_vvh0, p_etaav, MOM6HIP_MEM_HOST)
  endif
  call mom6hip_fatal_if(rc, "btstep")
  CS%dtbt = CS%st%dtbt
end subroutine btstep

!> Same interface as the reference set_dtbt (:2801).
subroutine set_dtbt(G, GV, US, CS, eta, pbce, BT_cont, gtot_est, SSH_add)
  type(ocean_grid_type),        intent(inout) :: G
  type(verticalGrid_type),      intent(in)    :: GV
  type(unit_scale_type),        intent(in)    :: US
  type(barotropic_CS), target,  intent(inout) :: CS
  real, dimension(SZI_(G),SZJ_(G)),          target, optional, intent(in) :: eta
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), target, optional, intent(in) :: pbce
  type(BT_cont_type), optional, pointer       :: BT_cont
  real,               optional, intent(in)    :: gtot_est
  real,               optional, intent(in)    :: SSH_add

  type(mom6hip_bt_cont_t), target :: cbt
  type(c_ptr) :: p_bt, p_pbce, p_eta
  real :: gt, ssh
  integer :: rc

  if (.not.CS%module_is_initialized) call MOM_error(FATAL, "set_dtbt: Module MOM_barotropic must be initialized before it is used.")
  if (.not.(present(pbce) .or. present(gtot_est))) call MOM_error(FATAL, "set_dtbt: Either pbce or gtot_est must be present.")
  call bind_arrays(CS)
  p_bt = c_null_ptr
  if (present(BT_cont)) then ; if (associated(BT_cont)) then ; call bt_cont_struct(BT_cont, cbt) ; p_bt = c_loc(cbt) ; endif ; endif
  p_pbce = c_null_ptr ; if (present(pbce)) p_pbce = c_loc(pbce)
  gt = 0.0 ; if (present(gtot_est)) gt = gtot_est
  ssh = 0.0 ; if (present(SSH_add)) ssh = SSH_add
  p_eta = c_null_ptr ; if (present(eta)) p_eta = c_loc(eta)      ! read with NONLINEAR_BT_CONTINUITY and no BT_cont (:2871)
  rc = mom6hip_set_dtbt_eta(mom6hip_shared_context(G, GV), CS%st, p_eta, p_pbce, p_bt, gt, ssh, MOM6HIP_MEM_HOST)
  call mom6hip_fatal_if(rc, "set_dtbt")
  CS%dtbt = CS%st%dtbt
end subroutine set_dtbt

!> Same interface as the reference btcalc (:3394).
subroutine btcalc(h, G, GV, CS, h_u, h_v, may_use_default, OBC)
  type(ocean_grid_type),   intent(inout) :: G
  type(verticalGrid_type), intent(in)    :: GV
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), target, intent(in) :: h
  type(barotropic_CS), target, intent(inout) :: CS
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, optional, intent(in) :: h_u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, optional, intent(in) :: h_v
  logical,       optional, intent(in)    :: may_use_default
  type(ocean_OBC_type), optional, pointer :: OBC
  type(mom6hip_obc_t) :: cobc
  type(mom6hip_obc_segment_t), allocatable, target :: csegs(:)
  type(c_ptr) :: p_hu, p_hv
  integer :: rc, mud
  logical :: with_OBC
  if (.not.CS%module_is_initialized) call MOM_error(FATAL, "btcalc: Module MOM_barotropic must be initialized before it is used.")
  if (.not.CS%split) return
  with_OBC = .false. ; if (present(OBC)) with_OBC = associated(OBC)
  call bind_arrays(CS)
  p_hu = c_null_ptr ; if (present(h_u)) p_hu = c_loc(h_u)
  p_hv = c_null_ptr ; if (present(h_v)) p_hv = c_loc(h_v)
  mud = 0 ; if (present(may_use_default)) mud = merge(1, 0, may_use_default)
  if (with_OBC) then      ! the weights at the faces of the segments are those of the cell inside (:3610-3664)
    call mom6hip_obc_to_c(OBC, cobc, csegs, size(CS%frhatu(:,:,1)), size(CS%frhatv(:,:,1)), "MOM_barotropic")
    rc = mom6hip_btcalc_obc(mom6hip_shared_context(G, GV), CS%st, c_loc(h), p_hu, p_hv, int(mud, c_int32_t), cobc, MOM6HIP_MEM_HOST)
  else
    rc = mom6hip_btcalc(mom6hip_shared_context(G, GV), CS%st, c_loc(h), p_hu, p_hv, int(mud, c_int32_t), MOM6HIP_MEM_HOST)
  endif
  call mom6hip_fatal_if(rc, "btcalc")
end subroutine btcalc

!> Same interface as the reference bt_mass_source (:4318).
subroutine bt_mass_source(h, eta, set_cor, G, GV, CS)
  type(ocean_grid_type),   intent(in) :: G
  type(verticalGrid_type), intent(in) :: GV
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), target, intent(in) :: h
  real, dimension(SZI_(G),SZJ_(G)),          target, intent(in) :: eta
  logical,                 intent(in) :: set_cor
  type(barotropic_CS), target, intent(inout) :: CS
  integer :: rc
  if (.not.CS%module_is_initialized) call MOM_error(FATAL, "bt_mass_source: Module MOM_barotropic must be initialized before it is used.")
  if (.not.CS%split) return
  call bind_arrays(CS)
  rc = mom6hip_bt_mass_source(mom6hip_shared_context(G, GV), CS%st, c_loc(h), c_loc(eta), merge(1_c_int32_t, 0_c_int32_t, set_cor), &
                              MOM6HIP_MEM_HOST)
  call mom6hip_fatal_if(rc, "bt_mass_source")
end subroutine bt_mass_source

!> Same interface as the reference barotropic_init (:4376): parameters by the reference's names and defaults
!! (:4465-4740), the time-invariant arrays, the first DTBT from an estimate of the total reduced gravity (:4899-4912), and
!! ubtav / vbtav from the initial velocities on a cold start (:5050-5062).
subroutine barotropic_init(u, v, h, eta, Time, G, GV, US, param_file, diag, CS, restart_CS, calc_dtbt, BT_cont, SAL_CSp)
  type(ocean_grid_type),   intent(inout) :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(unit_scale_type),   intent(in)    :: US
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), intent(in) :: u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), intent(in) :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  intent(in) :: h
  real, dimension(SZI_(G),SZJ_(G)),           intent(in) :: eta
  type(time_type), target, intent(in)    :: Time
  type(param_file_type),   intent(in)    :: param_file
  type(diag_ctrl), target, intent(inout) :: diag
  type(barotropic_CS), target, intent(inout) :: CS
  type(MOM_restart_CS),    intent(in)    :: restart_CS
  logical,                 intent(out)   :: calc_dtbt
  type(BT_cont_type),      pointer       :: BT_cont
  type(SAL_CS), target, optional :: SAL_CSp
# include "version_variable.h"
  character(len=40)  :: mdl = "MOM_barotropic"
  character(len=32)  :: hvel_str
  logical :: use_BT_cont_type, flag, flag2, flag3, use_tides, bug
  integer :: default_answer_date, answer_date, isd, ied, jsd, jed, nz, i, j, k, rc
  real :: dtbt_input, dtbt_tmp, gtot_estimate, SSH_extra, bt_cont_bounds

  if (CS%module_is_initialized) then
    call MOM_error(WARNING, "barotropic_init called with a control structure that has already been initialized.")
    return
  endif
  CS%module_is_initialized = .true.
  CS%diag => diag
  isd = G%isd ; ied = G%ied ; jsd = G%jsd ; jed = G%jed ; nz = GV%ke

  call get_param(param_file, mdl, "SPLIT", CS%split, "Use the split time stepping if true.", default=.true.)
  call log_version(param_file, mdl, version, "")
  if (.not.CS%split) return
  if (.not.GV%Boussinesq) call MOM_error(FATAL, "barotropic_init (HIP): a non-Boussinesq vertical grid is not supported by the GPU path.")

  CS%st%unsupported(:) = 0 ; CS%st%reserved0(:) = 0.0
  call get_param(param_file, mdl, "USE_BT_CONT_TYPE", use_BT_cont_type, &
                 "If true, use a structure with elements that describe effective face areas from the summed continuity solver.", &
                 default=.true.)
  ! the options the library does not provide are read with the reference's defaults and refused when switched on
  call get_param(param_file, mdl, "INTEGRAL_BT_CONTINUITY", flag, default=.false.) ; call refuse(flag, "INTEGRAL_BT_CONTINUITY")
  ! BOUND_BT_CORRECTION (:4485): provided with BT_CONT_CORR_BOUNDS (:4490, its default) and USE_BT_CONT_TYPE
  call get_param(param_file, mdl, "BOUND_BT_CORRECTION", flag, default=.false.)
  CS%st%bound_BT_corr = merge(1, 0, flag) ; CS%st%maxCFL_BT_cont = 0.25
  if (flag) then
    call get_param(param_file, mdl, "BT_CONT_CORR_BOUNDS", flag2, default=.true.)
    call get_param(param_file, mdl, "USE_BT_CONT_TYPE", flag3, default=.true.)
    call refuse(.not.(flag2 .and. flag3), "BOUND_BT_CORRECTION without BT_CONT_CORR_BOUNDS and USE_BT_CONT_TYPE")
    call get_param(param_file, mdl, "MAXCFL_BT_CONT", CS%st%maxCFL_BT_cont, units="nondim", default=0.25)
  endif
  call get_param(param_file, mdl, "GRADUAL_BT_ICS", flag, default=.false.) ; call refuse(flag, "GRADUAL_BT_ICS")
  call get_param(param_file, mdl, "NONLINEAR_BT_CONTINUITY", flag, &
                 "If true, use nonlinear transports in the barotropic continuity equation.", default=.false.)
  CS%st%Nonlinear_continuity = merge(1, 0, flag)
  call get_param(param_file, mdl, "NONLIN_BT_CONT_UPDATE_PERIOD", CS%st%Nonlin_cont_update_period, &
                 "If NONLINEAR_BT_CONTINUITY is true, this is the number of barotropic time steps between updates to the face areas, "// &
                 "or 0 to update only before the barotropic stepping.", units="nondim", default=1, do_not_log=.not.flag)
  call get_param(param_file, mdl, "BT_PROJECT_VELOCITY", flag, default=.false.) ; CS%st%BT_project_velocity = merge(1, 0, flag)
  call get_param(param_file, mdl, "BT_NONLIN_STRESS", flag, default=.false.) ; call refuse(flag, "BT_NONLIN_STRESS")
  call get_param(param_file, mdl, "DYNAMIC_SURFACE_PRESSURE", flag, default=.false.) ; call refuse(flag, "DYNAMIC_SURFACE_PRESSURE")
  call get_param(param_file, mdl, "BT_LINEAR_WAVE_DRAG", flag, default=.false.) ; call refuse(flag, "BT_LINEAR_WAVE_DRAG")
  call get_param(param_file, mdl, "CLIP_BT_VELOCITY", flag, default=.false.) ; call refuse(flag, "CLIP_BT_VELOCITY")
  call get_param(param_file, mdl, "TIDES", use_tides, default=.false.)
  call get_param(param_file, mdl, "CALCULATE_SAL", flag, default=use_tides) ; call refuse(flag, "CALCULATE_SAL")
  call get_param(param_file, mdl, "BT_USE_OLD_CORIOLIS_BRACKET_BUG", bug, default=.false.)
  call refuse(bug, "BT_USE_OLD_CORIOLIS_BRACKET_BUG")
  call get_param(param_file, mdl, "DEFAULT_ANSWER_DATE", default_answer_date, default=99991231)
  call get_param(param_file, mdl, "BAROTROPIC_ANSWER_DATE", answer_date, default=default_answer_date)
  call refuse(answer_date < 20190101, "BAROTROPIC_ANSWER_DATE < 20190101")
  call get_param(param_file, mdl, "BT_CONT_CORR_BOUNDS", flag, default=.true., do_not_log=.true.)

  call get_param(param_file, mdl, "ADJUST_BT_CONT", flag, &
                 "If true, adjust the curve fit to the BT_cont type that is used by the barotropic solver.", default=.false.)
  CS%st%adjust_BT_cont = merge(1, 0, flag)
  call get_param(param_file, mdl, "BT_USE_VISC_REM_U_UH0", flag, &
                 "If true, use the viscous remnants when estimating the barotropic velocities that were used to calculate uh0.", &
                 default=.false.)
  CS%st%visc_rem_u_uh0 = merge(1, 0, flag)
  call get_param(param_file, mdl, "BT_USE_WIDE_HALOS", flag, &
                 "If true, use wide halos and march in during the barotropic time stepping for efficiency.", default=.true.)
  CS%st%use_wide_halos = merge(1, 0, flag)
  call get_param(param_file, mdl, "BT_CORIOLIS_SCALE", CS%st%BT_Coriolis_scale, &
                 "A factor by which the barotropic Coriolis anomaly terms are scaled.", units="nondim", default=1.0)
  call get_param(param_file, mdl, "SADOURNY", flag, &
                 "If true, the Coriolis terms are discretized with the Sadourny (1975) energy conserving scheme.", default=.true.)
  CS%st%Sadourny = merge(1, 0, flag)
  call get_param(param_file, mdl, "BT_THICK_SCHEME", hvel_str, &
                 "A string describing the scheme that is used to set the open face areas used for barotropic transport.", &
                 default="FROM_BT_CONT")
  select case (trim(hvel_str))
    case ("HYBRID") ; CS%st%hvel_scheme = HYBRID
    case ("HARMONIC") ; CS%st%hvel_scheme = HARMONIC
    case ("ARITHMETIC") ; CS%st%hvel_scheme = ARITHMETIC
    case ("FROM_BT_CONT") ; CS%st%hvel_scheme = FROM_BT_CONT
    case default
      call MOM_mesg('barotropic_init: BT_THICK_SCHEME ="'//trim(hvel_str)//'"', 0)
      call MOM_error(FATAL, "barotropic_init: Unrecognized setting #define BT_THICK_SCHEME "//trim(hvel_str)//" found in input file.")
  end select
  if ((CS%st%hvel_scheme == FROM_BT_CONT) .and. .not.use_BT_cont_type) &
    call MOM_error(FATAL, "barotropic_init: BT_THICK_SCHEME FROM_BT_CONT can only be used if USE_BT_CONT_TYPE is defined.")
  call get_param(param_file, mdl, "BT_STRONG_DRAG", flag, &
                 "If true, use a stronger estimate of the retarding effects of strong bottom drag.", default=.false.)
  CS%st%strong_drag = merge(1, 0, flag)
  call get_param(param_file, mdl, "VEL_UNDERFLOW", CS%st%vel_underflow, &
                 "A negligibly small velocity magnitude below which velocity components are set to 0.", units="m s-1", &
                 default=0.0, scale=US%m_s_to_L_T)
  call get_param(param_file, mdl, "DT_BT_FILTER", CS%st%dt_bt_filter, &
                 "A time-scale over which the barotropic mode solutions are filtered.", units="sec or nondim", default=-0.25)
  if (CS%st%dt_bt_filter > 0.0) CS%st%dt_bt_filter = US%s_to_T*CS%st%dt_bt_filter
  call get_param(param_file, mdl, "G_BT_EXTRA", CS%st%G_extra, &
                 "A nondimensional factor by which gtot is enhanced.", units="nondim", default=0.0)
  call get_param(param_file, mdl, "SSH_EXTRA", SSH_extra, &
                 "An estimate of how much higher SSH might get, for use in calculating the safe external wave speed.", &
                 units="m", default=min(10.0, 0.05*G%max_depth*US%Z_to_m), scale=US%m_to_Z)
  call get_param(param_file, mdl, "LINEARIZED_BT_CORIOLIS", flag, &
                 "If true use the bottom depth instead of the total water column thickness in the barotropic Coriolis term.", &
                 default=.true.)
  CS%st%linearized_BT_PV = merge(1, 0, flag)
  call get_param(param_file, mdl, "BEBT", CS%st%bebt, &
                 "BEBT determines whether the barotropic time stepping uses the forward-backward time-stepping scheme or a "//&
                 "backward Euler scheme.", units="nondim", default=0.1)
  call get_param(param_file, mdl, "DTBT", dtbt_input, &
                 "The barotropic time step, in s; negative: the fraction of the stable maximum.", units="s or nondim", &
                 default=-0.98)
  CS%st%Z_ref = G%Z_ref ; CS%st%nstep_last = 0 ; CS%st%dtbt_max = 0.0

  ! the arrays of the control structure (ubtav, vbtav: register_barotropic_restarts)
  allocate(CS%frhatu(isd-1:ied,jsd:jed,nz), source=0.0) ; allocate(CS%frhatv(isd:ied,jsd-1:jed,nz), source=0.0)
  allocate(CS%eta_cor(isd:ied,jsd:jed), source=0.0)
  allocate(CS%IDatu(isd-1:ied,jsd:jed), source=0.0) ; allocate(CS%IDatv(isd:ied,jsd-1:jed), source=0.0)
  allocate(CS%q_D(isd-1:ied,jsd-1:jed), source=0.0)
  allocate(CS%D_u_Cor(isd-1:ied,jsd:jed), source=0.0) ; allocate(CS%D_v_Cor(isd:ied,jsd-1:jed), source=0.0)
  if (.not.allocated(CS%ubtav)) allocate(CS%ubtav(isd-1:ied,jsd:jed), source=0.0)
  if (.not.allocated(CS%vbtav)) allocate(CS%vbtav(isd:ied,jsd-1:jed), source=0.0)
  call mom6hip_read_topology(param_file)
  call bind_arrays(CS)
  dtbt_tmp = -1.0
  if (query_initialized(CS%dtbt, "DTBT", restart_CS)) dtbt_tmp = CS%dtbt
  CS%st%dtbt_fraction = 0.98 ; if (dtbt_input < 0.0) CS%st%dtbt_fraction = -dtbt_input
  CS%st%dtbt = 0.0
  rc = mom6hip_barotropic_init(mom6hip_shared_context(G, GV), CS%st, MOM6HIP_MEM_HOST)
  call mom6hip_fatal_if(rc, "barotropic_init")

  ! the first estimate of the stable barotropic time step (:4899-4920)
  calc_dtbt = .true.
  gtot_estimate = 0.0
  do k=1,GV%ke ; gtot_estimate = gtot_estimate + GV%H_to_Z*GV%g_prime(K) ; enddo
  call set_dtbt(G, GV, US, CS, gtot_est=gtot_estimate, SSH_add=SSH_extra)
  if (dtbt_input > 0.0) then
    CS%st%dtbt = US%s_to_T * dtbt_input
  elseif (dtbt_tmp > 0.0) then
    CS%st%dtbt = dtbt_tmp
  endif
  if ((dtbt_tmp > 0.0) .and. (dtbt_input > 0.0)) calc_dtbt = .false.
  CS%dtbt = CS%st%dtbt

  ! a cold start: the time-mean barotropic velocities from the initial state (:5050-5062)
  if (.not.query_initialized(CS%ubtav, "ubtav", restart_CS) .or. .not.query_initialized(CS%vbtav, "vbtav", restart_CS)) then
    call btcalc(h, G, GV, CS, may_use_default=.true.)
    CS%ubtav(:,:) = 0.0 ; CS%vbtav(:,:) = 0.0
    do k=1,nz ; do j=G%jsc,G%jec ; do I=G%isc-1,G%iec
      CS%ubtav(I,j) = CS%ubtav(I,j) + CS%frhatu(I,j,k) * u(I,j,k)
    enddo ; enddo ; enddo
    do k=1,nz ; do J=G%jsc-1,G%jec ; do i=G%isc,G%iec
      CS%vbtav(i,J) = CS%vbtav(i,J) + CS%frhatv(i,J,k) * v(i,J,k)
    enddo ; enddo ; enddo
  endif

contains
  subroutine refuse(on, name)
    logical,          intent(in) :: on
    character(len=*), intent(in) :: name
    if (on) call MOM_error(FATAL, "barotropic_init (HIP): "//name//" is not provided by the GPU path.")
  end subroutine refuse
end subroutine barotropic_init

!> Same interface as the reference barotropic_get_tav
subroutine barotropic_get_tav(CS, ubtav, vbtav, G, US)
  type(barotropic_CS),               intent(in)    :: CS
  type(ocean_grid_type),             intent(in)    :: G
  real, dimension(SZIB_(G),SZJ_(G)), intent(inout) :: ubtav
  real, dimension(SZI_(G),SZJB_(G)), intent(inout) :: vbtav
  type(unit_scale_type),             intent(in)    :: US
  integer :: i, j
  do j=G%jsc,G%jec ; do I=G%isc-1,G%iec ; ubtav(I,j) = CS%ubtav(I,j) ; enddo ; enddo
  do J=G%jsc-1,G%jec ; do i=G%isc,G%iec ; vbtav(i,J) = CS%vbtav(i,J) ; enddo ; enddo
end subroutine barotropic_get_tav

!> Same interface as the reference barotropic_end
subroutine barotropic_end(CS)
  type(barotropic_CS), intent(inout) :: CS
  if (allocated(CS%frhatu)) deallocate(CS%frhatu, CS%frhatv, CS%eta_cor, CS%IDatu, CS%IDatv, CS%q_D, CS%D_u_Cor, CS%D_v_Cor)
  if (allocated(CS%ubtav)) deallocate(CS%ubtav, CS%vbtav)
  CS%module_is_initialized = .false.
end subroutine barotropic_end

!> Same interface as the reference register_barotropic_restarts (:5165): ubtav, vbtav and DTBT
subroutine register_barotropic_restarts(HI, GV, US, param_file, CS, restart_CS)
  type(hor_index_type),    intent(in) :: HI
  type(verticalGrid_type), intent(in) :: GV
  type(unit_scale_type),   intent(in) :: US
  type(param_file_type),   intent(in) :: param_file
  type(barotropic_CS),     intent(inout) :: CS
  type(MOM_restart_CS),    intent(inout) :: restart_CS
  allocate(CS%ubtav(HI%IsdB:HI%IedB,HI%jsd:HI%jed), source=0.0)
  allocate(CS%vbtav(HI%isd:HI%ied,HI%JsdB:HI%JedB), source=0.0)
  call register_restart_field(CS%ubtav, "ubtav", .false., restart_CS, longname="Time mean barotropic zonal velocity", &
                              units="m s-1", conversion=US%L_T_to_m_s, hor_grid='u', z_grid='1')
  call register_restart_field(CS%vbtav, "vbtav", .false., restart_CS, longname="Time mean barotropic meridional velocity", &
                              units="m s-1", conversion=US%L_T_to_m_s, hor_grid='v', z_grid='1')
  call register_restart_field(CS%dtbt, "DTBT", .false., restart_CS, longname="Barotropic timestep", units="seconds", &
                              conversion=US%T_to_s)
end subroutine register_barotropic_restarts

end module MOM_barotropic
