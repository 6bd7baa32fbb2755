!> Drop-in replacement for module MOM_thickness_diffuse (src/parameterizations/lateral/MOM_thickness_diffuse.F90):
!! thickness_diffuse (:133), thickness_diffuse_init (:2169), thickness_diffuse_get_KH (:2470) and thickness_diffuse_end (:2492)
!! with the reference's dummy-argument lists, so step_MOM_thermo / step_MOM (src/core/MOM.F90:1149-1165) compile unchanged.
!! Provided: isopycnal height diffusion with KHTH, KHTH_MIN / KHTH_MAX / KHTH_MAX_CFL, the MEKE%Kh and Visbeck contributions,
!! VarMix%Res_fn_u/v, stored slopes or slopes from the density gradients (with or without an equation of state), the work into
!! MEKE%GM_src, CDp%uhGM / vhGM -- on the GPU through libmom6hip (mom6hip_thickness_diffuse, HOST memspace).  The FGNV
!! streamfunction, DETANGLE_INTERFACES, KH_ETA_*, USE_STANLEY_GM, MEKE_GEOMETRIC, MEKE_GM_SRC_ALT, READ_KHTH, the EBT structure,
!! QG Leith GM, depth scaling, USE_KH_IN_MEKE, USE_GME, SKEB and non-Boussinesq mode stop with a FATAL error; the diagnostics
!! other than uhGM / vhGM are not registered.
!!
!! Compiled INSIDE a MOM6 source tree in place of src/parameterizations/lateral/MOM_thickness_diffuse.F90; here against
!! tests/fortran/stubs.
module MOM_thickness_diffuse

use, intrinsic :: iso_c_binding
use mom6hip_c_api
use mom6hip_MOM_glue,          only : mom6hip_shared_context, mom6hip_read_topology, mom6hip_read_eos, mom6hip_fatal_if
use mom6hip_MOM_glue,          only : mom6hip_read_resident, mom6hip_resident, mom6hip_mirror, mom6hip_mirror_forget
use MOM_diag_mediator,         only : diag_ctrl, time_type
use MOM_error_handler,         only : MOM_error, FATAL
use MOM_file_parser,           only : get_param, log_version, param_file_type
use MOM_grid,                  only : ocean_grid_type
use MOM_lateral_mixing_coeffs, only : VarMix_CS
use MOM_MEKE_types,            only : MEKE_type
use MOM_stochastics,           only : stochastic_CS
use MOM_unit_scaling,          only : unit_scale_type
use MOM_variables,             only : thermo_var_ptrs, cont_diag_ptrs
use MOM_verticalGrid,          only : verticalGrid_type
implicit none ; private

#include <MOM_memory.h>

public thickness_diffuse, thickness_diffuse_init, thickness_diffuse_end
public thickness_diffuse_get_KH

!> Control structure (the members of the reference's thickness_diffuse_CS, :40-128, that the provided branch reads)
type, public :: thickness_diffuse_CS ; private
  logical :: initialized = .false.       !< True if this control structure has been initialized.
  logical :: thickness_diffuse = .false. !< If true, interfaces heights are diffused.
  real    :: Khth = 0.0                  !< Background isopycnal depth diffusivity [L2 T-1 ~> m2 s-1]
  real    :: Khth_Slope_Cff = 0.0        !< Slope dependence coefficient of Khth [nondim]
  real    :: max_Khth_CFL = 0.8          !< Maximum value of the diffusive CFL for isopycnal height diffusion [nondim]
  real    :: Khth_Min = 0.0              !< Minimum value of Khth [L2 T-1 ~> m2 s-1]
  real    :: Khth_Max = 0.0              !< Maximum value of Khth [L2 T-1 ~> m2 s-1], or 0 for no max
  real    :: slope_max = 0.01            !< Slopes steeper than slope_max are limited in some way [Z L-1 ~> nondim]
  real    :: kappa_smooth = 1.0e-6       !< Vertical diffusivity used to interpolate more sensible values of T & S
                                         !! into thin layers [H Z T-1 ~> m2 s-1 or kg m-1 s-1]
  logical :: use_GM_work_bug = .false.   !< If true, use the incorrect sign for the top-level work tendency on the top layer.
  logical :: use_GME_thickness_diffuse = .false.
  type(mom6hip_eos_t) :: eos             !< the equation of state, as read from the parameter file
  type(diag_ctrl), pointer :: diag => NULL()
  logical :: use_FGNV_streamfn = .false. !< KHTH_USE_FGNV_STREAMFUNCTION: the streamfunction of Ferrari et al. (2010)
  real :: FGNV_scale = 1.0, N2_floor = 0.0 !< FGNV_FILTER_SCALE and (FGNV_STRAT_FLOOR * OMEGA)**2
end type thickness_diffuse_CS

contains

!> Same interface as the reference thickness_diffuse (:133).
subroutine thickness_diffuse(h, uhtr, vhtr, tv, dt, G, GV, US, MEKE, VarMix, CDp, CS, STOCH)
  type(ocean_grid_type),                      intent(in)    :: G
  type(verticalGrid_type),                    intent(in)    :: GV
  type(unit_scale_type),                      intent(in)    :: US
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  target, intent(inout) :: h
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(inout) :: uhtr
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(inout) :: vhtr
  type(thermo_var_ptrs),                      intent(in)    :: tv
  real,                                       intent(in)    :: dt
  type(MEKE_type), target,                    intent(inout) :: MEKE
  type(VarMix_CS), target,                    intent(in)    :: VarMix
  type(cont_diag_ptrs),                       intent(inout) :: CDp
  type(thickness_diffuse_CS), target,         intent(inout) :: CS
  type(stochastic_CS),                        intent(inout) :: STOCH

  type(mom6hip_thickness_diffuse_cs_t) :: ccs
  type(c_ptr) :: p_T, p_S, p_eos, p_uhGM, p_vhGM, ctx
  integer :: n2
  real(c_double), allocatable, target :: Rlay(:), g_prime(:)
  integer :: i, j, rc

  if (.not. CS%initialized) call MOM_error(FATAL, "MOM_thickness_diffuse: "//&
         "Module must be initialized before it is used.")
  if ((.not.CS%thickness_diffuse) .or. .not. (CS%Khth > 0.0 .or. VarMix%use_variable_mixing)) return
  if (STOCH%skeb_use_gm) call MOM_error(FATAL, "thickness_diffuse (HIP): SKEB_USE_GM is not provided by the GPU path.")
  if (.not.GV%Boussinesq) call MOM_error(FATAL, "thickness_diffuse (HIP): non-Boussinesq mode is not provided by the GPU path.")
  if (associated(tv%p_surf)) call MOM_error(FATAL, "thickness_diffuse (HIP): tv%p_surf is not provided by the GPU path.")

  ccs%Khth = CS%Khth ; ccs%Khth_Min = CS%Khth_Min ; ccs%Khth_Max = CS%Khth_Max ; ccs%max_Khth_CFL = CS%max_Khth_CFL
  ccs%slope_max = CS%slope_max ; ccs%kappa_smooth = CS%kappa_smooth ; ccs%KHTH_Slope_Cff = CS%Khth_Slope_Cff
  ccs%thickness_diffuse = 1 ; ccs%use_GM_work_bug = merge(1, 0, CS%use_GM_work_bug) ; ccs%nkml = GV%nkml ; ccs%initialized = 1
  if (VarMix%use_variable_mixing) then
    ccs%use_variable_mixing = 1
    if (VarMix%khth_use_ebt_struct .or. VarMix%use_QG_Leith_GM) call MOM_error(FATAL, &
        "thickness_diffuse (HIP): KHTH_USE_EBT_STRUCT and USE_QG_LEITH_GM are not provided by the GPU path.")
    if (VarMix%Depth_scaled_KhTh) then      ! DEPTH_SCALED_KHTH :284-289
      ccs%Depth_fn_u = c_loc(VarMix%Depth_fn_u) ; ccs%Depth_fn_v = c_loc(VarMix%Depth_fn_v)
    endif
    if (VarMix%use_Visbeck .and. (CS%Khth_Slope_Cff > 0.)) then
      ccs%L2u = c_loc(VarMix%L2u) ; ccs%L2v = c_loc(VarMix%L2v) ; ccs%SN_u = c_loc(VarMix%SN_u) ; ccs%SN_v = c_loc(VarMix%SN_v)
    endif
    if (VarMix%Resoln_scaled_KhTh) then
      ccs%Res_fn_u = c_loc(VarMix%Res_fn_u) ; ccs%Res_fn_v = c_loc(VarMix%Res_fn_v)
    endif
    if (VarMix%use_stored_slopes) then
      ccs%slope_x = c_loc(VarMix%slope_x) ; ccs%slope_y = c_loc(VarMix%slope_y)
    endif
  endif
  if (allocated(MEKE%Kh)) then ; ccs%MEKE_Kh = c_loc(MEKE%Kh) ; ccs%KhTh_fac = MEKE%KhTh_fac ; endif
  if (allocated(MEKE%GM_src)) ccs%MEKE_GM_src = c_loc(MEKE%GM_src)
  if (allocated(GV%Rlay)) then ; allocate(Rlay(GV%ke)) ; Rlay(:) = GV%Rlay(1:GV%ke) ; ccs%Rlay = c_loc(Rlay) ; endif
  if (CS%use_FGNV_streamfn) then      ! :217, :860, :1095: VarMix%cg1, and GV%g_prime for the stratification without an equation of state
    ccs%use_FGNV_streamfn = 1 ; ccs%FGNV_scale = CS%FGNV_scale ; ccs%N2_floor = CS%N2_floor
    if (.not.(VarMix%use_variable_mixing .and. allocated(VarMix%cg1))) call MOM_error(FATAL, "cg1 must be associated when using FGNV streamfunction.")
    ccs%cg1 = c_loc(VarMix%cg1)
    if (allocated(GV%g_prime)) then ; allocate(g_prime(GV%ke+1)) ; g_prime(:) = GV%g_prime(1:GV%ke+1) ; ccs%g_prime = c_loc(g_prime) ; endif
  endif

  p_T = c_null_ptr ; p_S = c_null_ptr ; p_eos = c_null_ptr ; p_uhGM = c_null_ptr ; p_vhGM = c_null_ptr
  if (associated(tv%eqn_of_state)) then
    if (.not.(associated(tv%T) .and. associated(tv%S))) call MOM_error(FATAL, "thickness_diffuse (HIP): "// &
        "an equation of state needs tv%T and tv%S.")
    p_T = c_loc(tv%T) ; p_S = c_loc(tv%S) ; p_eos = c_loc(CS%eos)
  endif
  if (associated(CDp%uhGM)) p_uhGM = c_loc(CDp%uhGM)
  if (associated(CDp%vhGM)) p_vhGM = c_loc(CDp%vhGM)

  ctx = mom6hip_shared_context(G, GV)
  if (mom6hip_resident()) then      ! GPU_RESIDENT_DYNAMICS: the shared device mirrors of the host arrays (GV%Rlay stays a host table)
    n2 = size(h(:,:,1))
    call to_dev(p_T, size(h), .false.) ; call to_dev(p_S, size(h), .false.)
    call to_dev(p_uhGM, size(uhtr), .true.) ; call to_dev(p_vhGM, size(vhtr), .true.)
    call to_dev(ccs%MEKE_Kh, n2, .false.) ; call to_dev(ccs%MEKE_GM_src, n2, .true.) ; call to_dev(ccs%cg1, n2, .false.)
    call to_dev(ccs%Res_fn_u, size(uhtr(:,:,1)), .false.) ; call to_dev(ccs%Res_fn_v, size(vhtr(:,:,1)), .false.)
    call to_dev(ccs%Depth_fn_u, size(uhtr(:,:,1)), .false.) ; call to_dev(ccs%Depth_fn_v, size(vhtr(:,:,1)), .false.)
    call to_dev(ccs%L2u, size(uhtr(:,:,1)), .false.) ; call to_dev(ccs%SN_u, size(uhtr(:,:,1)), .false.)
    call to_dev(ccs%L2v, size(vhtr(:,:,1)), .false.) ; call to_dev(ccs%SN_v, size(vhtr(:,:,1)), .false.)
    call to_dev(ccs%slope_x, size(uhtr(:,:,1))*(GV%ke+1), .false.) ; call to_dev(ccs%slope_y, size(vhtr(:,:,1))*(GV%ke+1), .false.)
    rc = mom6hip_thickness_diffuse(ctx, ccs, mom6hip_mirror(ctx, c_loc(h), int(size(h), c_int64_t), .true., .true.), &
                                   mom6hip_mirror(ctx, c_loc(uhtr), int(size(uhtr), c_int64_t), .true., .true.), &
                                   mom6hip_mirror(ctx, c_loc(vhtr), int(size(vhtr), c_int64_t), .true., .true.), p_T, p_S, p_eos, dt, &
                                   p_uhGM, p_vhGM, MOM6HIP_MEM_DEVICE)
  else
    rc = mom6hip_thickness_diffuse(ctx, ccs, c_loc(h), c_loc(uhtr), c_loc(vhtr), p_T, p_S, p_eos, dt, &
                                   p_uhGM, p_vhGM, MOM6HIP_MEM_HOST)
  endif
  call mom6hip_fatal_if(rc, "thickness_diffuse")

  if (VarMix%use_variable_mixing) then
    if (allocated(MEKE%Rd_dx_h) .and. allocated(VarMix%Rd_dx_h)) then
      do j=G%jsc,G%jec ; do i=G%isc,G%iec
        MEKE%Rd_dx_h(i,j) = VarMix%Rd_dx_h(i,j)
      enddo ; enddo
    endif
  endif
contains
  !> a host pointer -> its device mirror (an input, or an in/out array the call writes)
  subroutine to_dev(p, n, written)
    type(c_ptr), intent(inout) :: p
    integer,     intent(in)    :: n
    logical,     intent(in)    :: written
    if (c_associated(p)) p = mom6hip_mirror(ctx, p, int(n, c_int64_t), .true., written)
  end subroutine to_dev
end subroutine thickness_diffuse

!> Same interface as the reference thickness_diffuse_init (:2169), same parameters and defaults (:2203-2400).
subroutine thickness_diffuse_init(Time, G, GV, US, param_file, diag, CDp, CS)
  type(time_type),         intent(in) :: Time
  type(ocean_grid_type),   intent(in) :: G
  type(verticalGrid_type), intent(in) :: GV
  type(unit_scale_type),   intent(in) :: US
  type(param_file_type),   intent(in) :: param_file
  type(diag_ctrl), target, intent(inout) :: diag
  type(cont_diag_ptrs),    intent(inout) :: CDp
  type(thickness_diffuse_CS), intent(inout) :: CS
# include "version_variable.h"
  character(len=40)  :: mdl = "MOM_thickness_diffuse"
  logical :: flag
  real :: val, val2, strat_floor, omega

  CS%initialized = .true.
  CS%diag => diag
  call log_version(param_file, mdl, version, "")
  call get_param(param_file, mdl, "THICKNESSDIFFUSE", CS%thickness_diffuse, &
                 "If true, interface heights are diffused with a coefficient of KHTH.", default=.false.)
  call get_param(param_file, mdl, "KHTH", CS%Khth, "The background horizontal thickness diffusivity.", &
                 default=0.0, units="m2 s-1", scale=US%m_to_L**2*US%T_to_s)
  call get_param(param_file, mdl, "READ_KHTH", flag, default=.false.) ; call refuse(flag, "READ_KHTH")
  call get_param(param_file, mdl, "KHTH_SLOPE_CFF", CS%KHTH_Slope_Cff, &
                 "The nondimensional coefficient in the Visbeck formula for the interface depth diffusivity", &
                 units="nondim", default=0.0)
  call get_param(param_file, mdl, "KHTH_MIN", CS%KHTH_Min, "The minimum horizontal thickness diffusivity.", &
                 default=0.0, units="m2 s-1", scale=US%m_to_L**2*US%T_to_s)
  call get_param(param_file, mdl, "KHTH_USE_EBT_STRUCT", flag, default=.false., do_not_log=.true.)
  call refuse(flag, "KHTH_USE_EBT_STRUCT")
  call get_param(param_file, mdl, "KHTH_MAX", CS%KHTH_Max, "The maximum horizontal thickness diffusivity.", &
                 default=0.0, units="m2 s-1", scale=US%m_to_L**2*US%T_to_s)
  call get_param(param_file, mdl, "KHTH_MAX_CFL", CS%max_Khth_CFL, &
                 "The maximum value of the local diffusive CFL ratio that is permitted for the thickness diffusivity.", &
                 units="nondimensional", default=0.8)
  call get_param(param_file, mdl, "KH_ETA_CONST", val, default=0.0, units="m2 s-1", scale=US%m_to_L**2*US%T_to_s)
  call get_param(param_file, mdl, "KH_ETA_VEL_SCALE", val2, default=0.0, units="m s-1", scale=US%m_to_L*US%T_to_s)
  call refuse((val > 0.0) .or. (val2 > 0.0), "KH_ETA_CONST / KH_ETA_VEL_SCALE")
  if (CS%max_Khth_CFL < 0.0) CS%max_Khth_CFL = 0.0
  if (CS%thickness_diffuse) call refuse(CS%max_Khth_CFL <= 0.0, "KHTH_MAX_CFL <= 0")
  call get_param(param_file, mdl, "DETANGLE_INTERFACES", flag, default=.false.) ; call refuse(flag, "DETANGLE_INTERFACES")
  call get_param(param_file, mdl, "KHTH_SLOPE_MAX", CS%slope_max, &
                 "A slope beyond which the calculated isopycnal slope is not reliable and is scaled away.", &
                 units="nondim", default=0.01, scale=US%L_to_Z)
  call get_param(param_file, mdl, "KD_SMOOTH", CS%kappa_smooth, &
                 "A diapycnal diffusivity that is used to interpolate more sensible values of T & S into thin layers.", &
                 units="m2 s-1", default=1.0e-6, scale=GV%m2_s_to_HZ_T)
  call get_param(param_file, mdl, "KHTH_USE_FGNV_STREAMFUNCTION", CS%use_FGNV_streamfn, &
                 "If true, use the streamfunction formulation of Ferrari et al., 2010, which effectively emphasizes graver vertical modes "// &
                 "by smoothing in the vertical.", default=.false.)
  call get_param(param_file, mdl, "FGNV_FILTER_SCALE", CS%FGNV_scale, &
                 "A coefficient scaling the vertical smoothing term in the Ferrari et al., 2010, streamfunction formulation.", &
                 units="nondim", default=1., do_not_log=.not.CS%use_FGNV_streamfn)
  call get_param(param_file, mdl, "FGNV_STRAT_FLOOR", strat_floor, &
                 "A floor for Brunt-Vasaila frequency in the Ferrari et al., 2010, streamfunction formulation, in units of the Coriolis "// &
                 "frequency.", default=1.e-15, units="nondim", do_not_log=.not.CS%use_FGNV_streamfn)
  call get_param(param_file, mdl, "OMEGA", omega, "The rotation rate of the earth.", default=7.2921e-5, units="s-1", scale=US%T_to_s, &
                 do_not_log=.not.CS%use_FGNV_streamfn)
  if (CS%use_FGNV_streamfn) CS%N2_floor = (strat_floor*omega)**2      ! :2337
  call get_param(param_file, mdl, "USE_STANLEY_GM", flag, default=.false.) ; call refuse(flag, "USE_STANLEY_GM")
  call get_param(param_file, mdl, "MEKE_GM_SRC_ALT", flag, default=.false.) ; call refuse(flag, "MEKE_GM_SRC_ALT")
  call get_param(param_file, mdl, "MEKE_GEOMETRIC", flag, default=.false.) ; call refuse(flag, "MEKE_GEOMETRIC")
  call get_param(param_file, mdl, "USE_KH_IN_MEKE", flag, default=.false.) ; call refuse(flag, "USE_KH_IN_MEKE")
  call get_param(param_file, mdl, "USE_GME", CS%use_GME_thickness_diffuse, default=.false.)
  call refuse(CS%use_GME_thickness_diffuse, "USE_GME")
  call get_param(param_file, mdl, "USE_GM_WORK_BUG", CS%use_GM_work_bug, &
                 "If true, compute the top-layer work tendency on the u-grid with the incorrect sign, for legacy reproducibility.", &
                 default=.false.)
  call mom6hip_read_eos(param_file, CS%eos, "thickness_diffuse_init")
  call mom6hip_read_topology(param_file)
  call mom6hip_read_resident(param_file)
contains
  subroutine refuse(on, name)
    logical,          intent(in) :: on
    character(len=*), intent(in) :: name
    if (on) call MOM_error(FATAL, "thickness_diffuse_init (HIP): "//name//" is not provided by the GPU path.")
  end subroutine refuse
end subroutine thickness_diffuse_init

!> Same interface as the reference thickness_diffuse_get_KH (:2470); its arrays exist only with USE_GME, which is refused.
subroutine thickness_diffuse_get_KH(CS, KH_u_GME, KH_v_GME, G, GV)
  type(thickness_diffuse_CS),          intent(in)  :: CS
  type(ocean_grid_type),               intent(in)  :: G
  type(verticalGrid_type),             intent(in)  :: GV
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)+1), intent(inout) :: KH_u_GME
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)+1), intent(inout) :: KH_v_GME
  call MOM_error(FATAL, "thickness_diffuse_get_KH (HIP): USE_GME is not provided by the GPU path.")
end subroutine thickness_diffuse_get_KH

!> Same interface as the reference thickness_diffuse_end (:2492)
subroutine thickness_diffuse_end(CS, CDp)
  type(thickness_diffuse_CS), intent(inout) :: CS
  type(cont_diag_ptrs),       intent(inout) :: CDp
  ! (the mirrors of arrays that go away are dropped, so that nothing allocated at their addresses later inherits a device copy)
  if (associated(CDp%uhGM)) then ; call mom6hip_mirror_forget(c_loc(CDp%uhGM)) ; deallocate(CDp%uhGM) ; endif
  if (associated(CDp%vhGM)) then ; call mom6hip_mirror_forget(c_loc(CDp%vhGM)) ; deallocate(CDp%vhGM) ; endif
  CS%initialized = .false.
end subroutine thickness_diffuse_end

end module MOM_thickness_diffuse
