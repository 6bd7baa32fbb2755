!> Drop-in replacement for module MOM_CoriolisAdv (src/core/MOM_CoriolisAdv.F90): CorAdCalc (:125), CoriolisAdv_init
!! (:1054) and CoriolisAdv_end (:1322) with the reference's dummy-argument lists, so the split RK2 step (:566, :869, :1058)
!! compiles unchanged.  The work is done by libmom6hip (mom6hip_coradcalc) on host arrays (HOST memspace).
!! Provided: CORIOLIS_SCHEME = SADOURNY75_ENERGY, ARAKAWA_HSU90, SADOURNY75_ENSTRO; KE_SCHEME = KE_ARAKAWA,
!! KE_SIMPLE_GUDONOV, KE_GUDONOV; NOSLIP; BOUND_CORIOLIS; PV_ADV_SCHEME = PV_ADV_CENTERED.  Anything else, open boundaries,
!! porous barriers, the Stokes vortex force and the acceleration diagnostics of accel_diag_ptrs stop with a FATAL error.
!!
!! Compiled INSIDE a MOM6 source tree in place of src/core/MOM_CoriolisAdv.F90; here against tests/fortran/stubs.
module MOM_CoriolisAdv

use, intrinsic :: iso_c_binding
use mom6hip_c_api
use mom6hip_MOM_glue,     only : mom6hip_shared_context, mom6hip_read_topology, mom6hip_fatal_if, mom6hip_obc_to_c
use MOM_diag_mediator,    only : diag_ctrl, time_type
use MOM_error_handler,    only : MOM_error, MOM_mesg, FATAL, WARNING
use MOM_file_parser,      only : get_param, log_version, param_file_type
use MOM_grid,             only : ocean_grid_type
use MOM_open_boundary,    only : ocean_OBC_type
use MOM_string_functions, only : uppercase
use MOM_unit_scaling,     only : unit_scale_type
use MOM_variables,        only : accel_diag_ptrs, porous_barrier_type
use MOM_verticalGrid,     only : verticalGrid_type
use MOM_wave_interface,   only : Wave_parameters_CS
implicit none ; private

#include <MOM_memory.h>

public CorAdCalc, CoriolisAdv_init, CoriolisAdv_end
public CoriolisAdv_hip_struct      ! (GPU path only) for MOM_dynamics_split_RK2

!> Control structure (the members of the reference's CoriolisAdv_CS, :30-88, that the provided options need)
type, public :: CoriolisAdv_CS ; private
  logical :: initialized = .false.
  integer :: Coriolis_Scheme, KE_Scheme, PV_Adv_Scheme
  logical :: no_slip, bound_Coriolis, Coriolis_En_Dis
  real :: F_eff_max_blend = 4.0, wt_lin_blend = 0.125
  type(diag_ctrl), pointer :: diag => NULL()
  type(time_type), pointer :: Time => NULL()
end type CoriolisAdv_CS

! the reference's enumerations (:91-117) -- the values the library's struct takes
integer, parameter :: SADOURNY75_ENERGY = 1, ARAKAWA_HSU90 = 2, ROBUST_ENSTRO = 3, SADOURNY75_ENSTRO = 4, &
                      ARAKAWA_LAMB81 = 5, AL_BLEND = 6
integer, parameter :: KE_ARAKAWA = 10, KE_SIMPLE_GUDONOV = 11, KE_GUDONOV = 12
integer, parameter :: PV_ADV_CENTERED = 21, PV_ADV_UPWIND1 = 22

contains

!> (GPU path only) The control structure as the library's struct
function CoriolisAdv_hip_struct(CS) result(ccs)
  type(CoriolisAdv_CS), intent(in) :: CS
  type(mom6hip_coriolisadv_cs_t) :: ccs
  if (.not.CS%initialized) call MOM_error(FATAL, "MOM_CoriolisAdv: Module must be initialized before it is used.")
  ccs%coriolis_scheme = CS%Coriolis_Scheme ; ccs%ke_scheme = CS%KE_Scheme
  ccs%no_slip = merge(1, 0, CS%no_slip) ; ccs%bound_coriolis = merge(1, 0, CS%bound_Coriolis)
  ccs%coriolis_en_dis = merge(1, 0, CS%Coriolis_En_Dis) ; ccs%pv_adv_scheme = CS%PV_Adv_Scheme ; ccs%reserved(:) = 0
  ccs%F_eff_max_blend = CS%F_eff_max_blend ; ccs%wt_lin_blend = CS%wt_lin_blend
end function CoriolisAdv_hip_struct

!> Same interface as the reference CorAdCalc (:125).
subroutine CorAdCalc(u, v, h, uh, vh, CAu, CAv, OBC, AD, G, GV, US, CS, pbv, Waves)
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(in)  :: u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(in)  :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  target, intent(in)  :: h
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(in)  :: uh
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(in)  :: vh
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(out) :: CAu
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(out) :: CAv
  type(ocean_OBC_type),    pointer       :: OBC
  type(accel_diag_ptrs),   intent(inout) :: AD
  type(unit_scale_type),   intent(in)    :: US
  type(CoriolisAdv_CS),    intent(in)    :: CS
  type(porous_barrier_type), intent(in)  :: pbv
  type(Wave_parameters_CS), optional, pointer :: Waves

  type(mom6hip_coriolisadv_cs_t) :: ccs
  type(mom6hip_obc_t) :: cobc
  type(mom6hip_obc_segment_t), allocatable, target :: csegs(:)
  integer :: rc

  if (.not.CS%initialized) call MOM_error(FATAL, "MOM_CoriolisAdv: Module must be initialized before it is used.")
  if (allocated(pbv%por_face_areaU)) then
    if (any(pbv%por_face_areaU /= 1.0) .or. any(pbv%por_face_areaV /= 1.0)) &
      call MOM_error(FATAL, "MOM_CoriolisAdv (HIP): porous barriers are not supported by the GPU path.")
  endif
  if (present(Waves)) then ; if (associated(Waves)) then
    if (Waves%Stokes_VF) call MOM_error(FATAL, "MOM_CoriolisAdv (HIP): the Stokes vortex force is not supported by the GPU path.")
  endif ; endif
  if (associated(AD%gradKEu) .or. associated(AD%gradKEv) .or. associated(AD%rv_x_u) .or. associated(AD%rv_x_v)) &
    call MOM_error(FATAL, "MOM_CoriolisAdv (HIP): the acceleration diagnostics (gradKEu, rv_x_u ...) are not provided "//&
                          "by the GPU path.")

  ccs%coriolis_scheme = CS%Coriolis_Scheme ; ccs%ke_scheme = CS%KE_Scheme
  ccs%no_slip = merge(1, 0, CS%no_slip) ; ccs%bound_coriolis = merge(1, 0, CS%bound_Coriolis)
  ccs%coriolis_en_dis = merge(1, 0, CS%Coriolis_En_Dis) ; ccs%pv_adv_scheme = CS%PV_Adv_Scheme ; ccs%reserved(:) = 0
  ccs%F_eff_max_blend = CS%F_eff_max_blend ; ccs%wt_lin_blend = CS%wt_lin_blend
  if (associated(OBC)) then      ! the OBC branches of CorAdCalc (:249-269, :337-455, gradKE :1037-1050)
    call mom6hip_obc_to_c(OBC, cobc, csegs, size(CAu(:,:,1)), size(CAv(:,:,1)), "MOM_CoriolisAdv")
    rc = mom6hip_coradcalc_obc(mom6hip_shared_context(G, GV), ccs, cobc, c_loc(u), c_loc(v), c_loc(h), c_loc(uh), c_loc(vh), c_loc(CAu), &
                               c_loc(CAv), MOM6HIP_MEM_HOST)
  else
  rc = mom6hip_coradcalc(mom6hip_shared_context(G, GV), ccs, c_loc(u), c_loc(v), c_loc(h), c_loc(uh), c_loc(vh), c_loc(CAu), &
                         c_loc(CAv), MOM6HIP_MEM_HOST)
  endif
  call mom6hip_fatal_if(rc, "MOM_CoriolisAdv")
end subroutine CorAdCalc

!> Same interface as the reference CoriolisAdv_init (:1054), same parameters and defaults (:1079-1192).
subroutine CoriolisAdv_init(Time, G, GV, US, param_file, diag, AD, CS)
  type(time_type), target, intent(in)    :: Time
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(unit_scale_type),   intent(in)    :: US
  type(param_file_type),   intent(in)    :: param_file
  type(diag_ctrl), target, intent(inout) :: diag
  type(accel_diag_ptrs), target, intent(inout) :: AD
  type(CoriolisAdv_CS),    intent(inout) :: CS
#include "version_variable.h"
  character(len=40)  :: mdl = "MOM_CoriolisAdv"
  character(len=20)  :: tmpstr

  CS%initialized = .true.
  CS%diag => diag ; CS%Time => Time
  call log_version(param_file, mdl, version, "")
  call get_param(param_file, mdl, "NOSLIP", CS%no_slip, &
                 "If true, no slip boundary conditions are used; otherwise free slip boundary conditions are assumed.", &
                 default=.false.)
  call get_param(param_file, mdl, "CORIOLIS_EN_DIS", CS%Coriolis_En_Dis, &
                 "If true, two estimates of the thickness fluxes are used to estimate the Coriolis term, and the one that "//&
                 "dissipates energy relative to the other one is used.", default=.false.)
  call get_param(param_file, mdl, "CORIOLIS_SCHEME", tmpstr, &
                 "CORIOLIS_SCHEME selects the discretization for the Coriolis terms.", default="SADOURNY75_ENERGY")
  select case (uppercase(tmpstr))
    case ("SADOURNY75_ENERGY") ; CS%Coriolis_Scheme = SADOURNY75_ENERGY
    case ("ARAKAWA_HSU90") ; CS%Coriolis_Scheme = ARAKAWA_HSU90
    case ("SADOURNY75_ENSTRO") ; CS%Coriolis_Scheme = SADOURNY75_ENSTRO
    case ("ROBUST_ENSTRO") ; CS%Coriolis_Scheme = ROBUST_ENSTRO
    case ("ARAKAWA_LAMB81") ; CS%Coriolis_Scheme = ARAKAWA_LAMB81
    case ("ARAKAWA_LAMB_BLEND") ; CS%Coriolis_Scheme = AL_BLEND
    case default
      call MOM_mesg('CoriolisAdv_init: Coriolis_Scheme ="'//trim(tmpstr)//'"', 0)
      call MOM_error(FATAL, "CoriolisAdv_init: Unrecognized setting #define CORIOLIS_SCHEME "//trim(tmpstr)//" found in input file.")
  end select
  if (CS%Coriolis_Scheme == AL_BLEND) then      ! :1125-1142
    call get_param(param_file, mdl, "CORIOLIS_BLEND_WT_LIN", CS%wt_lin_blend, &
                 "A weighting value for the ratio of inverse thicknesses, beyond which the blending between Sadourny Energy "//&
                 "and Arakawa & Hsu goes linearly to 0 when CORIOLIS_SCHEME is ARAWAKA_LAMB_BLEND.", units="nondim", default=0.125)
    call get_param(param_file, mdl, "CORIOLIS_BLEND_F_EFF_MAX", CS%F_eff_max_blend, &
                 "The factor by which the maximum effective Coriolis acceleration from any point can be increased when "//&
                 "blending different discretizations with the ARAKAWA_LAMB_BLEND Coriolis scheme.", units="nondim", default=4.0)
    CS%wt_lin_blend = min(1.0, max(CS%wt_lin_blend,1e-16))
    if (CS%F_eff_max_blend < 2.0) call MOM_error(WARNING, "CoriolisAdv_init: CORIOLIS_BLEND_F_EFF_MAX should be at least 2.")
  endif
  call get_param(param_file, mdl, "BOUND_CORIOLIS", CS%bound_Coriolis, &
                 "If true, the Coriolis terms at u-points are bounded by the four estimates of (f+rv)v from the four "//&
                 "neighboring v-points, and similarly at v-points.", default=.false.)
  if ((CS%Coriolis_En_Dis .and. (CS%Coriolis_Scheme == SADOURNY75_ENERGY)) .or. &      ! :1155-1156
      (CS%Coriolis_Scheme == ROBUST_ENSTRO)) CS%bound_Coriolis = .false.
  call get_param(param_file, mdl, "KE_SCHEME", tmpstr, &
                 "KE_SCHEME selects the discretization for acceleration due to the kinetic energy gradient.", default="KE_ARAKAWA")
  select case (uppercase(tmpstr))
    case ("KE_ARAKAWA") ; CS%KE_Scheme = KE_ARAKAWA
    case ("KE_SIMPLE_GUDONOV") ; CS%KE_Scheme = KE_SIMPLE_GUDONOV
    case ("KE_GUDONOV") ; CS%KE_Scheme = KE_GUDONOV
    case default
      call MOM_mesg('CoriolisAdv_init: KE_Scheme ="'//trim(tmpstr)//'"', 0)
      call MOM_error(FATAL, "CoriolisAdv_init: #define KE_SCHEME "//trim(tmpstr)//" in input file is invalid.")
  end select
  call get_param(param_file, mdl, "PV_ADV_SCHEME", tmpstr, &
                 "PV_ADV_SCHEME selects the discretization for PV advection.", default="PV_ADV_CENTERED")
  select case (uppercase(tmpstr))
    case ("PV_ADV_CENTERED") ; CS%PV_Adv_Scheme = PV_ADV_CENTERED
    case ("PV_ADV_UPWIND1") ; CS%PV_Adv_Scheme = PV_ADV_UPWIND1
    case default
      call MOM_mesg('CoriolisAdv_init: PV_Adv_Scheme ="'//trim(tmpstr)//'"', 0)
      call MOM_error(FATAL, "CoriolisAdv_init: #DEFINE PV_ADV_SCHEME in input file is invalid.")
  end select
  call mom6hip_read_topology(param_file)
end subroutine CoriolisAdv_init

!> Same interface as the reference CoriolisAdv_end (:1322)
subroutine CoriolisAdv_end(CS)
  type(CoriolisAdv_CS), intent(inout) :: CS
  CS%initialized = .false.
end subroutine CoriolisAdv_end

end module MOM_CoriolisAdv
