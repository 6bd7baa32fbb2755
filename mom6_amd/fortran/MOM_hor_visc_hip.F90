!> Drop-in replacement for module MOM_hor_visc (src/parameterizations/lateral/MOM_hor_visc.F90): horizontal_viscosity (:245),
!! hor_visc_init (:1984), hor_visc_end, hor_visc_vel_stencil (:2879) with the reference's dummy-argument lists, so the split
!! RK2 step (:860) and its initialisation (:1543) compile unchanged.  The work is done by libmom6hip (mom6hip_hor_visc_init,
!! mom6hip_horizontal_viscosity, HOST memspace).  Provided: LAPLACIAN (KH, KH_VEL_SCALE, KH_BG_MIN, SMAGORINSKY_KH,
!! ADD_LES_VISCOSITY, BOUND_KH, BETTER_BOUND_KH), BIHARMONIC (AH, AH_VEL_SCALE, AH_TIME_SCALE, SMAGORINSKY_AH,
!! BOUND_CORIOLIS_BIHARM, BOUND_AH, BETTER_BOUND_AH), NOSLIP, USE_LAND_MASK_FOR_HVISC, HORVISC_BOUND_COEF,
!! USE_CONT_THICKNESS.  Leith / Leithy, MEKE, GME, anisotropic viscosity, RE_AH, KH_SIN_LAT, USE_KH_BG_2D, ZB2020,
!! resolution-scaled viscosities, open boundaries and the FrictWork diagnostics stop with a FATAL error.
!!
!! Compiled INSIDE a MOM6 source tree in place of the reference file; here against tests/fortran/stubs.
module MOM_hor_visc

use, intrinsic :: iso_c_binding
use mom6hip_c_api
use mom6hip_MOM_glue,          only : mom6hip_shared_context, mom6hip_read_topology, mom6hip_fatal_if, mom6hip_obc_to_c
use MOM_barotropic,            only : barotropic_CS
use MOM_diag_mediator,         only : diag_ctrl, time_type
use MOM_error_handler,         only : MOM_error, FATAL, WARNING
use MOM_file_parser,           only : get_param, log_version, param_file_type
use MOM_grid,                  only : ocean_grid_type
use MOM_lateral_mixing_coeffs, only : VarMix_CS
use MOM_MEKE_types,            only : MEKE_type
use MOM_open_boundary,         only : ocean_OBC_type
use MOM_stochastics,           only : stochastic_CS
use MOM_thickness_diffuse,     only : thickness_diffuse_CS
use MOM_unit_scaling,          only : unit_scale_type
use MOM_variables,             only : accel_diag_ptrs, thermo_var_ptrs
use MOM_verticalGrid,          only : verticalGrid_type
implicit none ; private

#include <MOM_memory.h>

public horizontal_viscosity, hor_visc_init, hor_visc_end, hor_visc_vel_stencil
public hor_visc_hip_struct, hor_visc_hip_MEKE      ! (GPU path only) for MOM_dynamics_split_RK2

!> Control structure: the library's struct and the static arrays hor_visc_init computes (h-point *_xx, q-point *_xy)
type, public :: hor_visc_CS ; private
  logical :: initialized = .false.
  type(mom6hip_hor_visc_cs_t) :: st
  real, allocatable, dimension(:,:) :: Kh_bg_xx, Kh_Max_xx, Ah_bg_xx, Ah_Max_xx, Laplac2_const_xx, Biharm_const_xx, Biharm_const2_xx, &
                                       reduction_xx
  real, allocatable, dimension(:,:) :: Kh_bg_xy, Kh_Max_xy, Ah_bg_xy, Ah_Max_xy, Laplac2_const_xy, Biharm_const_xy, Biharm_const2_xy, &
                                       reduction_xy
  type(diag_ctrl), pointer :: diag => NULL()
end type hor_visc_CS

contains

subroutine bind_arrays(CS)
  type(hor_visc_CS), target, intent(inout) :: CS
  CS%st%Kh_bg_xx = c_loc(CS%Kh_bg_xx) ; CS%st%Kh_Max_xx = c_loc(CS%Kh_Max_xx) ; CS%st%Ah_bg_xx = c_loc(CS%Ah_bg_xx)
  CS%st%Ah_Max_xx = c_loc(CS%Ah_Max_xx) ; CS%st%Laplac2_const_xx = c_loc(CS%Laplac2_const_xx)
  CS%st%Biharm_const_xx = c_loc(CS%Biharm_const_xx) ; CS%st%Biharm_const2_xx = c_loc(CS%Biharm_const2_xx)
  CS%st%reduction_xx = c_loc(CS%reduction_xx)
  CS%st%Kh_bg_xy = c_loc(CS%Kh_bg_xy) ; CS%st%Kh_Max_xy = c_loc(CS%Kh_Max_xy) ; CS%st%Ah_bg_xy = c_loc(CS%Ah_bg_xy)
  CS%st%Ah_Max_xy = c_loc(CS%Ah_Max_xy) ; CS%st%Laplac2_const_xy = c_loc(CS%Laplac2_const_xy)
  CS%st%Biharm_const_xy = c_loc(CS%Biharm_const_xy) ; CS%st%Biharm_const2_xy = c_loc(CS%Biharm_const2_xy)
  CS%st%reduction_xy = c_loc(CS%reduction_xy)
  CS%st%reserved1(:) = c_null_ptr
end subroutine bind_arrays

!> (GPU path only) The library's struct of this control structure (pointers bound to the HOST arrays of CS, which hor_visc_init
!! has filled)
function hor_visc_hip_struct(CS) result(st)
  type(hor_visc_CS), target, intent(inout) :: CS
  type(mom6hip_hor_visc_cs_t) :: st
  if (.not.CS%initialized) call MOM_error(FATAL, "MOM_hor_visc: Module must be initialized before it is used.")
  call bind_arrays(CS)
  st = CS%st
end function hor_visc_hip_struct

!> (GPU path only) The MEKE argument of horizontal_viscosity in the library's struct: MEKE%Ku, MEKE%Au (added to the viscosities,
!! :1141, :1318, :1537, :1634) and MEKE%mom_src (the vertically summed frictional work, :1833-1889) as HOST pointers, null when not
!! allocated.  Refused: the Rossby-number dependent backscatter (MEKE_BACKSCAT_RO_C /= 0).
subroutine hor_visc_hip_MEKE(st, MEKE)
  type(mom6hip_hor_visc_cs_t), intent(inout) :: st
  type(MEKE_type), target,     intent(inout) :: MEKE
  st%MEKE_Ku = c_null_ptr ; st%MEKE_Au = c_null_ptr ; st%MEKE_mom_src = c_null_ptr
  if (allocated(MEKE%Ku)) st%MEKE_Ku = c_loc(MEKE%Ku)
  if (allocated(MEKE%Au)) st%MEKE_Au = c_loc(MEKE%Au)
  if (allocated(MEKE%mom_src)) then
    if (MEKE%backscatter_Ro_c /= 0.0) call MOM_error(FATAL, "horizontal_viscosity (HIP): MEKE_BACKSCAT_RO_C /= 0 is not provided "// &
        "by the GPU path.")
    st%MEKE_mom_src = c_loc(MEKE%mom_src)
    if (allocated(MEKE%GME_snk)) MEKE%GME_snk(:,:) = 0.0      ! :1838-1842 (USE_GME is refused)
  endif
end subroutine hor_visc_hip_MEKE

!> Same interface as the reference horizontal_viscosity (:245).
subroutine horizontal_viscosity(u, v, h, diffu, diffv, MEKE, VarMix, G, GV, US, &
                                CS, tv, dt, OBC, BT, TD, ADp, hu_cont, hv_cont, STOCH)
  type(ocean_grid_type),         intent(in)  :: G
  type(verticalGrid_type),       intent(in)  :: GV
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(in)    :: u
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(in)    :: v
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  target, intent(inout) :: h
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(out)   :: diffu
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(out)   :: diffv
  type(MEKE_type),               intent(inout) :: MEKE
  type(VarMix_CS),               intent(inout) :: VarMix
  type(unit_scale_type),         intent(in)    :: US
  type(hor_visc_CS), target,     intent(inout) :: CS
  type(thermo_var_ptrs),         intent(in)    :: tv
  real,                          intent(in)    :: dt
  type(ocean_OBC_type), optional, pointer      :: OBC
  type(barotropic_CS), optional, intent(in)    :: BT
  type(thickness_diffuse_CS), optional, intent(in) :: TD
  type(accel_diag_ptrs), optional, intent(in)  :: ADp
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, optional, intent(in) :: hu_cont
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, optional, intent(in) :: hv_cont
  type(stochastic_CS), intent(inout), optional :: STOCH
  type(c_ptr) :: p_hu, p_hv
  type(mom6hip_obc_t) :: cobc
  type(mom6hip_obc_segment_t), allocatable, target :: csegs(:)
  logical :: apply_OBC
  integer :: rc
  if (.not.CS%initialized) call MOM_error(FATAL, "MOM_hor_visc: Module must be initialized before it is used.")
  if (.not.(CS%st%Laplacian /= 0 .or. CS%st%biharmonic /= 0)) return      ! :451
  apply_OBC = .false.      ! :449-452
  if (present(OBC)) then ; if (associated(OBC)) then ; if (OBC%OBC_pe) apply_OBC = .true. ; endif ; endif
  ! VarMix: the resolution function scales the Laplacian viscosity only (rescale_Kh, :474-476, :1123, :1525)
  if (VarMix%use_variable_mixing .and. VarMix%Resoln_scaled_Kh .and. (CS%st%Laplacian /= 0)) &
    call MOM_error(FATAL, "horizontal_viscosity (HIP): RESOLN_SCALED_KH with LAPLACIAN is not provided by the GPU path.")
  call bind_arrays(CS)
  call hor_visc_hip_MEKE(CS%st, MEKE)
  p_hu = c_null_ptr ; if (present(hu_cont)) p_hu = c_loc(hu_cont)
  p_hv = c_null_ptr ; if (present(hv_cont)) p_hv = c_loc(hv_cont)
  diffu(:,:,:) = 0.0 ; diffv(:,:,:) = 0.0      ! (intent(out): the library writes the computational ranges only)
  if (apply_OBC) then      ! the strains, thicknesses and Laplacians at the segments (:733-849, :889-903, :1388-1409, :1751-1782)
    call mom6hip_obc_to_c(OBC, cobc, csegs, size(u(:,:,1)), size(v(:,:,1)), "MOM_hor_visc")
    rc = mom6hip_horizontal_viscosity_obc(mom6hip_shared_context(G, GV), CS%st, c_loc(u), c_loc(v), c_loc(h), c_loc(diffu), &
                                          c_loc(diffv), dt, p_hu, p_hv, cobc, MOM6HIP_MEM_HOST)
  else
    rc = mom6hip_horizontal_viscosity(mom6hip_shared_context(G, GV), CS%st, c_loc(u), c_loc(v), c_loc(h), c_loc(diffu), c_loc(diffv), dt, &
                                      p_hu, p_hv, MOM6HIP_MEM_HOST)
  endif
  call mom6hip_fatal_if(rc, "horizontal_viscosity")
end subroutine horizontal_viscosity

!> Same interface as the reference hor_visc_init (:1984), same parameters and defaults (:2062-2340).
subroutine hor_visc_init(Time, G, GV, US, param_file, diag, CS, ADp)
  type(time_type),         intent(in)    :: Time
  type(ocean_grid_type),   intent(inout) :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(unit_scale_type),   intent(in)    :: US
  type(param_file_type),   intent(in)    :: param_file
  type(diag_ctrl), target, intent(inout) :: diag
  type(hor_visc_CS), target, intent(inout) :: CS
  type(accel_diag_ptrs), intent(in), optional :: ADp
# include "version_variable.h"
  character(len=40)  :: mdl = "MOM_hor_visc"
  logical :: flag, flag2, bound_Cor_def
  real :: val, maxvel, dt
  integer :: isd, ied, jsd, jed, default_answer_date, answer_date, rc

  CS%initialized = .true. ; CS%diag => diag
  isd = G%isd ; ied = G%ied ; jsd = G%jsd ; jed = G%jed
  CS%st%unsupported(:) = 0 ; CS%st%reserved0(:) = 0.0 ; CS%st%initialized = 0
  call log_version(param_file, mdl, version, "")
  call get_param(param_file, mdl, "DEFAULT_ANSWER_DATE", default_answer_date, default=99991231)
  call get_param(param_file, mdl, "HOR_VISC_ANSWER_DATE", answer_date, default=default_answer_date)
  call refuse(answer_date < 20190101, "HOR_VISC_ANSWER_DATE < 20190101")
  call get_param(param_file, mdl, "USE_CONT_THICKNESS", flag, &
                 "If true, use thickness at velocity points from continuity solver.", default=.false.)
  CS%st%use_cont_thick = merge(1, 0, flag)
  call get_param(param_file, mdl, "LAPLACIAN", flag, "If true, use a Laplacian horizontal viscosity.", default=.false.)
  CS%st%Laplacian = merge(1, 0, flag)
  call get_param(param_file, mdl, "KH", CS%st%Kh, "The background Laplacian horizontal viscosity.", units="m2 s-1", default=0.0, &
                 scale=US%m_to_L**2*US%T_to_s)
  call get_param(param_file, mdl, "KH_BG_MIN", CS%st%Kh_bg_min, "The minimum value allowed for Laplacian horizontal viscosity, KH.", &
                 units="m2 s-1", default=0.0, scale=US%m_to_L**2*US%T_to_s)
  call get_param(param_file, mdl, "KH_VEL_SCALE", CS%st%Kh_vel_scale, &
                 "The velocity scale which is multiplied by the grid spacing to calculate the Laplacian viscosity.", &
                 units="m s-1", default=0.0, scale=US%m_s_to_L_T)
  call get_param(param_file, mdl, "KH_SIN_LAT", val, units="m2 s-1", default=0.0) ; call refuse(val /= 0.0, "KH_SIN_LAT")
  call get_param(param_file, mdl, "SMAGORINSKY_KH", flag, "If true, use a Smagorinsky nonlinear eddy viscosity.", default=.false.)
  CS%st%Smagorinsky_Kh = merge(1, 0, flag)
  call get_param(param_file, mdl, "SMAG_LAP_CONST", CS%st%Smag_Lap_const, &
                 "The nondimensional Laplacian Smagorinsky constant, often 0.15.", units="nondim", default=0.0, &
                 fail_if_missing=(CS%st%Smagorinsky_Kh /= 0))
  call get_param(param_file, mdl, "LEITH_KH", flag, default=.false.) ; call refuse(flag, "LEITH_KH")
  ! USE_MEKE only decides what is logged (:2122-2128): MEKE acts through the MEKE argument of horizontal_viscosity
  call get_param(param_file, mdl, "USE_MEKE", flag, default=.false., do_not_log=.true.)
  call get_param(param_file, mdl, "RES_SCALE_MEKE_VISC", flag2, &
                 "If true, the viscosity contribution from MEKE is scaled by the resolution function.", default=.false., &
                 do_not_log=.not.((CS%st%Laplacian /= 0) .and. flag))
  call refuse(flag2 .and. flag .and. (CS%st%Laplacian /= 0), "RES_SCALE_MEKE_VISC")
  call get_param(param_file, mdl, "BOUND_KH", flag, &
                 "If true, the Laplacian coefficient is locally limited to be stable.", default=.true.)
  CS%st%bound_Kh = merge(1, 0, flag)
  call get_param(param_file, mdl, "BETTER_BOUND_KH", flag, &
                 "If true, the Laplacian coefficient is locally limited to be stable with a better bounding than just BOUND_KH.", &
                 default=(CS%st%bound_Kh /= 0))
  CS%st%better_bound_Kh = merge(1, 0, flag)
  call get_param(param_file, mdl, "ANISOTROPIC_VISCOSITY", flag, default=.false.) ; call refuse(flag, "ANISOTROPIC_VISCOSITY")
  call get_param(param_file, mdl, "ADD_LES_VISCOSITY", flag, &
                 "If true, adds the viscosity from Smagorinsky and Leith to the background viscosity instead of taking the maximum.", &
                 default=.false.)
  CS%st%add_LES_viscosity = merge(1, 0, flag)
  call get_param(param_file, mdl, "BIHARMONIC", flag, "If true, use a biharmonic horizontal viscosity.", default=.true.)
  CS%st%biharmonic = merge(1, 0, flag)
  call get_param(param_file, mdl, "AH", CS%st%Ah, "The background biharmonic horizontal viscosity.", units="m4 s-1", default=0.0, &
                 scale=US%m_to_L**4*US%T_to_s)
  call get_param(param_file, mdl, "AH_VEL_SCALE", CS%st%Ah_vel_scale, &
                 "The velocity scale which is multiplied by the cube of the grid spacing to calculate the biharmonic viscosity.", &
                 units="m s-1", default=0.0, scale=US%m_s_to_L_T)
  call get_param(param_file, mdl, "AH_TIME_SCALE", CS%st%Ah_time_scale, &
                 "A time scale whose inverse is multiplied by the fourth power of the grid spacing to calculate biharmonic viscosity.", &
                 units="s", default=0.0, scale=US%s_to_T)
  call get_param(param_file, mdl, "SMAGORINSKY_AH", flag, "If true, use a biharmonic Smagorinsky nonlinear eddy viscosity.", &
                 default=.false.)
  CS%st%Smagorinsky_Ah = merge(1, 0, flag)
  call get_param(param_file, mdl, "LEITH_AH", flag, default=.false.) ; call refuse(flag, "LEITH_AH")
  call get_param(param_file, mdl, "USE_LEITHY", flag, default=.false.) ; call refuse(flag, "USE_LEITHY")
  call get_param(param_file, mdl, "BOUND_AH", flag, "If true, the biharmonic coefficient is locally limited to be stable.", &
                 default=.true.)
  CS%st%bound_Ah = merge(1, 0, flag)
  call get_param(param_file, mdl, "BETTER_BOUND_AH", flag, &
                 "If true, the biharmonic coefficient is locally limited to be stable with a better bounding than just BOUND_AH.", &
                 default=(CS%st%bound_Ah /= 0))
  CS%st%better_bound_Ah = merge(1, 0, flag)
  call get_param(param_file, mdl, "RE_AH", val, units="nondim", default=0.0) ; call refuse(val /= 0.0, "RE_AH")
  call get_param(param_file, mdl, "SMAG_BI_CONST", CS%st%Smag_bi_const, &
                 "The nondimensional biharmonic Smagorinsky constant, typically 0.015 - 0.06.", units="nondim", default=0.0, &
                 fail_if_missing=(CS%st%Smagorinsky_Ah /= 0))
  call get_param(param_file, mdl, "BOUND_CORIOLIS", bound_Cor_def, default=.false.)
  call get_param(param_file, mdl, "BOUND_CORIOLIS_BIHARM", flag, &
                 "If true use a viscosity that increases with the square of the velocity shears.", default=bound_Cor_def)
  if (CS%st%Smagorinsky_Ah == 0) flag = .false.      ! :2256
  CS%st%bound_Coriolis = merge(1, 0, flag)
  call get_param(param_file, mdl, "MAXVEL", maxvel, units="m s-1", default=3.0e8)
  call get_param(param_file, mdl, "BOUND_CORIOLIS_VEL", CS%st%bound_Cor_vel, &
                 "The velocity scale at which BOUND_CORIOLIS_BIHARM causes the biharmonic drag to have comparable magnitude to the Coriolis acceleration.", &
                 units="m s-1", default=maxvel, scale=US%m_s_to_L_T)
  call get_param(param_file, mdl, "USE_LAND_MASK_FOR_HVISC", flag, &
                 "If true, use the land mask for the computation of thicknesses at velocity locations.", default=.true.)
  CS%st%use_land_mask = merge(1, 0, flag)
  call get_param(param_file, mdl, "HORVISC_BOUND_COEF", CS%st%bound_coef, &
                 "The nondimensional coefficient of the ratio of the viscosity bounds to the theoretical maximum for stability.", &
                 units="nondim", default=0.8)
  call get_param(param_file, mdl, "NOSLIP", flag, "If true, no slip boundary conditions are used.", default=.false.)
  CS%st%no_slip = merge(1, 0, flag)
  call get_param(param_file, mdl, "USE_KH_BG_2D", flag, default=.false.) ; call refuse(flag, "USE_KH_BG_2D")
  call get_param(param_file, mdl, "USE_GME", flag, default=.false.) ; call refuse(flag, "USE_GME")
  call get_param(param_file, mdl, "USE_ZB2020", flag, default=.false.) ; call refuse(flag, "USE_ZB2020")
  if (CS%st%no_slip /= 0 .and. CS%st%biharmonic /= 0) &
    call MOM_error(FATAL, "ERROR: NOSLIP and BIHARMONIC cannot be defined at the same time in MOM.")
  call get_param(param_file, mdl, "DT", dt, "The (baroclinic) dynamics time step.", units="s", scale=US%s_to_T, fail_if_missing=.true.)

  allocate(CS%Kh_bg_xx(isd:ied,jsd:jed), CS%Kh_Max_xx(isd:ied,jsd:jed), CS%Ah_bg_xx(isd:ied,jsd:jed), CS%Ah_Max_xx(isd:ied,jsd:jed), &
           CS%Laplac2_const_xx(isd:ied,jsd:jed), CS%Biharm_const_xx(isd:ied,jsd:jed), CS%Biharm_const2_xx(isd:ied,jsd:jed), &
           CS%reduction_xx(isd:ied,jsd:jed), source=0.0)
  allocate(CS%Kh_bg_xy(isd-1:ied,jsd-1:jed), CS%Kh_Max_xy(isd-1:ied,jsd-1:jed), CS%Ah_bg_xy(isd-1:ied,jsd-1:jed), &
           CS%Ah_Max_xy(isd-1:ied,jsd-1:jed), CS%Laplac2_const_xy(isd-1:ied,jsd-1:jed), CS%Biharm_const_xy(isd-1:ied,jsd-1:jed), &
           CS%Biharm_const2_xy(isd-1:ied,jsd-1:jed), CS%reduction_xy(isd-1:ied,jsd-1:jed), source=0.0)
  call mom6hip_read_topology(param_file)
  call bind_arrays(CS)
  rc = mom6hip_hor_visc_init(mom6hip_shared_context(G, GV), CS%st, dt, MOM6HIP_MEM_HOST)
  call mom6hip_fatal_if(rc, "hor_visc_init")
contains
  subroutine refuse(on, name)
    logical,          intent(in) :: on
    character(len=*), intent(in) :: name
    if (on) call MOM_error(FATAL, "hor_visc_init (HIP): "//name//" is not provided by the GPU path.")
  end subroutine refuse
end subroutine hor_visc_init

!> hor_visc_vel_stencil (:2879)
function hor_visc_vel_stencil(CS) result(stencil)
  type(hor_visc_CS), intent(in) :: CS
  integer ::  stencil
  stencil = 2
end function hor_visc_vel_stencil

!> Same interface as the reference hor_visc_end
subroutine hor_visc_end(CS)
  type(hor_visc_CS), intent(inout) :: CS
  if (allocated(CS%Kh_bg_xx)) deallocate(CS%Kh_bg_xx, CS%Kh_Max_xx, CS%Ah_bg_xx, CS%Ah_Max_xx, CS%Laplac2_const_xx, CS%Biharm_const_xx, &
                                         CS%Biharm_const2_xx, CS%reduction_xx, CS%Kh_bg_xy, CS%Kh_Max_xy, CS%Ah_bg_xy, CS%Ah_Max_xy, &
                                         CS%Laplac2_const_xy, CS%Biharm_const_xy, CS%Biharm_const2_xy, CS%reduction_xy)
  CS%initialized = .false.
end subroutine hor_visc_end

end module MOM_hor_visc
