!> Drop-in replacement for module MOM_tracer_hor_diff (src/tracer/MOM_tracer_hor_diff.F90): tracer_hordiff (:119),
!! tracer_hor_diff_init (:1625) and tracer_hor_diff_end (:1772) with the reference's dummy-argument lists, so
!! step_MOM_tracer_dyn (src/core/MOM.F90:1441) compiles unchanged.  Provided: the along-layer diffusion with a constant
!! KHTR (MAX_TR_DIFFUSION_CFL, CHECK_DIFFUSIVE_CFL, the tracers' conc_underflow) or, with VarMix%use_variable_mixing, the face
!! diffusivities of :236-281 (KHTR_SLOPE_CFF with VarMix%L2u / SN_u, MEKE%KhTr_fac with MEKE%Kh, KHTR_MIN / KHTR_MAX,
!! RESOLN_SCALED_KHTR with VarMix%Res_fn_h, KHTR_PASSIVITY_COEFF / _MIN with VarMix%Rd_dx_h) on the GPU through libmom6hip
!! (mom6hip_tracer_hordiff_varmix, HOST memspace), and with USE_NEUTRAL_DIFFUSION the continuous branch of MOM_neutral_diffusion
!! (neutral_diffusion_init :138, neutral_diffusion_calc_coeffs :337, neutral_diffusion :605: NDIFF_REF_PRES, NDIFF_ANSWER_DATE,
!! RECALC_NEUTRAL_SURF, NDIFF_INTERIOR_ONLY with visc%h_ML; mom6hip_tracer_hordiff_neutral), and with DIFFUSE_ML_TO_INTERIOR the
!! epipycnal diffusion between the variable-density layers and the interior of a layered run (tracer_epipycnal_ML_diff :700,
!! ML_KHTR_SCALE, HOR_DIFF_ANSWER_DATE, HOR_DIFF_LIMIT_BUG; mom6hip_tracer_hordiff_epipycnal).  NDIFF_CONTINUOUS = False,
!! NDIFF_TAPERING, horizontal boundary diffusion, KHTR_USE_EBT_STRUCT, offline khdt arrays and the df_x / df_y flux diagnostics stop
!! with a FATAL error.
!!
!! Compiled INSIDE a MOM6 source tree in place of src/tracer/MOM_tracer_hor_diff.F90; here against tests/fortran/stubs.
module MOM_tracer_hor_diff

use, intrinsic :: iso_c_binding
use mom6hip_c_api
use mom6hip_MOM_glue,          only : mom6hip_shared_context, mom6hip_read_topology, mom6hip_fatal_if
use mom6hip_MOM_glue,          only : mom6hip_read_resident, mom6hip_resident, mom6hip_mirror, mom6hip_read_eos
use MOM_cpu_clock,             only : cpu_clock_id, cpu_clock_begin, cpu_clock_end, CLOCK_MODULE
use MOM_diabatic_driver,       only : diabatic_CS
use MOM_diag_mediator,         only : diag_ctrl, time_type
use MOM_EOS,                   only : EOS_type
use MOM_error_handler,         only : MOM_error, FATAL, WARNING
use MOM_file_parser,           only : get_param, log_version, param_file_type
use MOM_grid,                  only : ocean_grid_type
use MOM_lateral_mixing_coeffs, only : VarMix_CS
use MOM_MEKE_types,            only : MEKE_type
use MOM_tracer_registry,       only : tracer_registry_type
use MOM_unit_scaling,          only : unit_scale_type
use MOM_variables,             only : thermo_var_ptrs, vertvisc_type
use MOM_verticalGrid,          only : verticalGrid_type
implicit none ; private

#include <MOM_memory.h>

public tracer_hordiff, tracer_hor_diff_init, tracer_hor_diff_end

!> Control structure (the members of the reference's tracer_hor_diff_CS, :40-100, that the provided branch reads)
type, public :: tracer_hor_diff_CS ; private
  real    :: KhTr                 !< The along-isopycnal tracer diffusivity [L2 T-1 ~> m2 s-1].
  real    :: KhTr_Slope_Cff       !< The non-dimensional coefficient in KhTr formula [nondim]
  real    :: KhTr_min             !< Minimum along-isopycnal tracer diffusivity [L2 T-1 ~> m2 s-1].
  real    :: KhTr_max             !< Maximum along-isopycnal tracer diffusivity [L2 T-1 ~> m2 s-1].
  real    :: KhTr_passivity_coeff !< Passivity coefficient that scales Rd/dx [nondim]
  real    :: KhTr_passivity_min   !< Passivity minimum [nondim]
  real    :: max_diff_CFL         !< If positive, locally limit the diffusivity to this diffusive CFL [nondim].
  logical :: check_diffusive_CFL  !< If true, use enough iterations that the diffusive equations are stable.
  logical :: Diffuse_ML_interior, use_neutral_diffusion, use_hor_bnd_diffusion
  logical :: first_call = .true.
  logical :: recalc_neutral_surf  !< If true, recalculate the neutral surfaces if CFL has been exceeded
  type(mom6hip_neutral_diffusion_cs_t) :: nd   !< neutral_diffusion_CS as the library reads it
  type(mom6hip_epipycnal_cs_t) :: epi          !< the parameters of tracer_epipycnal_ML_diff as the library reads them
  type(mom6hip_eos_t) :: eos                   !< the equation of state tracer_hor_diff_init was given
  type(diag_ctrl), pointer :: diag => NULL()
end type tracer_hor_diff_CS

integer :: id_clock_diffuse

contains

!> Same interface as the reference tracer_hordiff (:119).
subroutine tracer_hordiff(h, dt, MEKE, VarMix, visc, G, GV, US, CS, Reg, tv, do_online_flag, read_khdt_x, read_khdt_y)
  type(ocean_grid_type),      intent(inout) :: G
  type(verticalGrid_type),    intent(in)    :: GV
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), target, intent(in) :: h
  real,                       intent(in)    :: dt
  type(MEKE_type), target,    intent(in)    :: MEKE
  type(VarMix_CS), target,    intent(in)    :: VarMix
  type(vertvisc_type),        intent(in)    :: visc
  type(unit_scale_type),      intent(in)    :: US
  type(tracer_hor_diff_CS),   pointer       :: CS
  type(tracer_registry_type), pointer       :: Reg
  type(thermo_var_ptrs),      intent(in)    :: tv
  logical,          optional, intent(in)    :: do_online_flag
  real, dimension(SZIB_(G),SZJ_(G)), optional, intent(in) :: read_khdt_x
  real, dimension(SZI_(G),SZJB_(G)), optional, intent(in) :: read_khdt_y

  type(mom6hip_tracer_hor_diff_cs_t) :: ccs
  type(mom6hip_hordiff_fields_t) :: fld
  type(c_ptr) :: ctx
  type(mom6hip_hordiff_stats_t) :: stats
  type(c_ptr), allocatable :: tr(:)
  real(c_double), allocatable, target :: cu(:)
  type(c_ptr) :: p_surf
  real(c_double), allocatable, target :: Rlay(:)
  integer :: m, rc, idx_T, idx_S

  if (.not. associated(CS)) call MOM_error(FATAL, "MOM_tracer_hor_diff: "// &
       "register_tracer must be called before tracer_hordiff.")
  if (.not. associated(Reg)) call MOM_error(FATAL, "MOM_tracer_hor_diff: "// &
       "register_tracer must be called before tracer_hordiff.")
  if (Reg%ntr == 0 .or. (CS%KhTr <= 0.0 .and. .not. VarMix%use_variable_mixing)) return
  if (present(do_online_flag)) then ; if (.not.do_online_flag) &
    call MOM_error(FATAL, "tracer_hordiff (HIP): offline tracer diffusion (read_khdt_x/y) is not provided by the GPU path.")
  endif
  call cpu_clock_begin(id_clock_diffuse)
  CS%first_call = .false.

  allocate(tr(Reg%ntr), cu(Reg%ntr))
  do m=1,Reg%ntr
    if (associated(Reg%Tr(m)%df_x) .or. associated(Reg%Tr(m)%df_y) .or. associated(Reg%Tr(m)%df2d_x) .or. &
        associated(Reg%Tr(m)%df2d_y)) call MOM_error(FATAL, "tracer_hordiff (HIP): the diffusive flux diagnostics of tracer "// &
        trim(Reg%Tr(m)%name)//" are not provided by the GPU path.")
    tr(m) = c_loc(Reg%Tr(m)%t)
    cu(m) = Reg%Tr(m)%conc_underflow
  enddo
  ccs%KhTr = CS%KhTr ; ccs%max_diff_CFL = CS%max_diff_CFL
  ccs%KhTr_Slope_Cff = CS%KhTr_Slope_Cff ; ccs%KhTr_min = CS%KhTr_min ; ccs%KhTr_max = CS%KhTr_max
  ccs%KhTr_passivity_coeff = CS%KhTr_passivity_coeff ; ccs%KhTr_passivity_min = CS%KhTr_passivity_min
  ccs%check_diffusive_CFL = merge(1, 0, CS%check_diffusive_CFL) ; ccs%initialized = 1
  ccs%unsupported(:) = 0 ; ccs%reserved1(:) = 0
  if (VarMix%use_variable_mixing) then      ! :219-224, :236-281
    ccs%use_variable_mixing = 1
    if (CS%KhTr_Slope_Cff > 0.) then
      fld%L2u = c_loc(VarMix%L2u) ; fld%L2v = c_loc(VarMix%L2v) ; fld%SN_u = c_loc(VarMix%SN_u) ; fld%SN_v = c_loc(VarMix%SN_v)
    endif
    if (VarMix%Resoln_scaled_KhTr) then ; ccs%Resoln_scaled_KhTr = 1 ; fld%Res_fn_h = c_loc(VarMix%Res_fn_h) ; endif
    if (CS%KhTr_passivity_coeff > 0.) fld%Rd_dx_h = c_loc(VarMix%Rd_dx_h)
    if (allocated(MEKE%Kh)) then ; fld%MEKE_Kh = c_loc(MEKE%Kh) ; ccs%KhTr_fac = MEKE%KhTr_fac ; endif
  endif
  idx_T = -1 ; idx_S = -1 ; p_surf = c_null_ptr
  if (CS%use_neutral_diffusion) then      ! :474-534: tv%T and tv%S are registered tracers, found by association
    ccs%unsupported(1) = 1
    if (.not.(associated(tv%T) .and. associated(tv%S))) call MOM_error(FATAL, &
      "tracer_hordiff (HIP): USE_NEUTRAL_DIFFUSION needs tv%T and tv%S.")
    do m=1,Reg%ntr
      if (associated(Reg%Tr(m)%t, tv%T)) idx_T = m-1
      if (associated(Reg%Tr(m)%t, tv%S)) idx_S = m-1
    enddo
    if (idx_T < 0 .or. idx_S < 0) call MOM_error(FATAL, "tracer_hordiff (HIP): tv%T and tv%S must be registered tracers.")
    if (associated(tv%p_surf)) p_surf = c_loc(tv%p_surf)
    if (CS%nd%interior_only /= 0) then      ! MOM_neutral_diffusion.F90:372-378
      if (.not.associated(visc%h_ML)) call MOM_error(FATAL, "hor_bnd_diffusion requires that visc%h_ML is associated.")
      fld%h_ML = c_loc(visc%h_ML)
    endif
    CS%nd%H_to_RZ = GV%H_to_RZ ; CS%nd%recalc_neutral_surf = merge(1, 0, CS%recalc_neutral_surf)
  endif
  if (CS%Diffuse_ML_interior) then      ! :544-550, :613-620: tv%T and tv%S are registered tracers, found by association
    ccs%unsupported(3) = 1
    if (.not.(associated(tv%T) .and. associated(tv%S))) call MOM_error(FATAL, &
      "tracer_hordiff (HIP): DIFFUSE_ML_TO_INTERIOR needs tv%T and tv%S.")
    do m=1,Reg%ntr
      if (associated(Reg%Tr(m)%t, tv%T)) idx_T = m-1
      if (associated(Reg%Tr(m)%t, tv%S)) idx_S = m-1
    enddo
    if (idx_T < 0 .or. idx_S < 0) call MOM_error(FATAL, "tracer_hordiff (HIP): tv%T and tv%S must be registered tracers.")
    allocate(Rlay(GV%ke)) ; Rlay(:) = GV%Rlay(1:GV%ke)
    CS%epi%Rlay = c_loc(Rlay) ; CS%epi%nkml = GV%nkml ; CS%epi%nk_rho_varies = GV%nk_rho_varies ; CS%epi%P_Ref = tv%P_Ref
  endif
  if (mom6hip_resident()) then      ! GPU_RESIDENT_DYNAMICS: the shared device mirrors of the host arrays
    ctx = mom6hip_shared_context(G, GV)
    do m=1,Reg%ntr ; tr(m) = mom6hip_mirror(ctx, tr(m), int(size(h), c_int64_t), .true., .true.) ; enddo
    call to_dev(p_surf, size(h(:,:,1))) ; call to_dev(fld%h_ML, size(h(:,:,1)))
    call to_dev(fld%MEKE_Kh, size(h(:,:,1))) ; call to_dev(fld%Res_fn_h, size(h(:,:,1))) ; call to_dev(fld%Rd_dx_h, size(h(:,:,1)))
    if (c_associated(fld%L2u)) then
      call to_dev(fld%L2u, size(VarMix%L2u)) ; call to_dev(fld%SN_u, size(VarMix%SN_u))
      call to_dev(fld%L2v, size(VarMix%L2v)) ; call to_dev(fld%SN_v, size(VarMix%SN_v))
    endif
    if (CS%Diffuse_ML_interior) then
      rc = mom6hip_tracer_hordiff_epipycnal(ctx, ccs, CS%epi, fld, mom6hip_mirror(ctx, c_loc(h), int(size(h), c_int64_t), .true., .false.), &
                                            CS%eos, dt, tr, c_loc(cu), int(Reg%ntr, c_int32_t), int(idx_T, c_int32_t), &
                                            int(idx_S, c_int32_t), MOM6HIP_MEM_DEVICE, stats)
    else
    rc = mom6hip_tracer_hordiff_neutral(ctx, ccs, CS%nd, fld, mom6hip_mirror(ctx, c_loc(h), int(size(h), c_int64_t), .true., .false.), &
                                        CS%eos, p_surf, dt, tr, c_loc(cu), int(Reg%ntr, c_int32_t), int(idx_T, c_int32_t), &
                                        int(idx_S, c_int32_t), MOM6HIP_MEM_DEVICE, stats)
    endif
  elseif (CS%Diffuse_ML_interior) then
    rc = mom6hip_tracer_hordiff_epipycnal(mom6hip_shared_context(G, GV), ccs, CS%epi, fld, c_loc(h), CS%eos, dt, tr, c_loc(cu), &
                                          int(Reg%ntr, c_int32_t), int(idx_T, c_int32_t), int(idx_S, c_int32_t), MOM6HIP_MEM_HOST, stats)
  else
    rc = mom6hip_tracer_hordiff_neutral(mom6hip_shared_context(G, GV), ccs, CS%nd, fld, c_loc(h), CS%eos, p_surf, dt, tr, c_loc(cu), &
                                        int(Reg%ntr, c_int32_t), int(idx_T, c_int32_t), int(idx_S, c_int32_t), MOM6HIP_MEM_HOST, stats)
  endif
  CS%epi%Rlay = c_null_ptr
  call mom6hip_fatal_if(rc, "tracer_hordiff")
  call cpu_clock_end(id_clock_diffuse)
contains
  !> a host pointer of the struct -> its device mirror (an input)
  subroutine to_dev(p, n)
    type(c_ptr), intent(inout) :: p
    integer,     intent(in)    :: n
    if (c_associated(p)) p = mom6hip_mirror(ctx, p, int(n, c_int64_t), .true., .false.)
  end subroutine to_dev
end subroutine tracer_hordiff

!> Same interface as the reference tracer_hor_diff_init (:1625), same parameters and defaults (:1652-1700).
subroutine tracer_hor_diff_init(Time, G, GV, US, param_file, diag, EOS, diabatic_CSp, CS)
  type(time_type), target,    intent(in)    :: Time
  type(ocean_grid_type),      intent(in)    :: G
  type(verticalGrid_type),    intent(in)    :: GV
  type(unit_scale_type),      intent(in)    :: US
  type(diag_ctrl), target,    intent(inout) :: diag
  type(EOS_type),  target,    intent(in)    :: EOS
  type(diabatic_CS), pointer, intent(in)    :: diabatic_CSp
  type(param_file_type),      intent(in)    :: param_file
  type(tracer_hor_diff_CS),   pointer       :: CS
# include "version_variable.h"
  character(len=40)  :: mdl = "MOM_tracer_hor_diff"
  logical :: flag

  if (associated(CS)) then
    call MOM_error(WARNING, "tracer_hor_diff_init called with associated control structure.")
    return
  endif
  allocate(CS)
  CS%diag => diag
  call log_version(param_file, mdl, version, "")
  call get_param(param_file, mdl, "KHTR", CS%KhTr, "The background along-isopycnal tracer diffusivity.", &
                 units="m2 s-1", default=0.0, scale=US%m_to_L**2*US%T_to_s)
  call get_param(param_file, mdl, "KHTR_USE_EBT_STRUCT", flag, default=.false.) ; call refuse(flag, "KHTR_USE_EBT_STRUCT")
  call get_param(param_file, mdl, "KHTR_SLOPE_CFF", CS%KhTr_Slope_Cff, &
                 "The scaling coefficient for along-isopycnal tracer diffusivity using a shear-based (Visbeck-like) "//&
                 "parameterization.  A non-zero value enables this param.", units="nondim", default=0.0)
  call get_param(param_file, mdl, "KHTR_MIN", CS%KhTr_Min, "The minimum along-isopycnal tracer diffusivity.", &
                 units="m2 s-1", default=0.0, scale=US%m_to_L**2*US%T_to_s)
  call get_param(param_file, mdl, "KHTR_MAX", CS%KhTr_Max, "The maximum along-isopycnal tracer diffusivity.", &
                 units="m2 s-1", default=0.0, scale=US%m_to_L**2*US%T_to_s)
  call get_param(param_file, mdl, "KHTR_PASSIVITY_COEFF", CS%KhTr_passivity_coeff, &
                 "The coefficient that scales deformation radius over grid-spacing in passivity.", units="nondim", default=0.0)
  call get_param(param_file, mdl, "KHTR_PASSIVITY_MIN", CS%KhTr_passivity_min, &
                 "The minimum passivity which is the ratio between along isopycnal mixing of tracers to thickness mixing.", &
                 units="nondim", default=0.5)
  call get_param(param_file, mdl, "DIFFUSE_ML_TO_INTERIOR", CS%Diffuse_ML_interior, &
                 "If true, enable epipycnal mixing between the surface boundary layer and the interior.", default=.false.)
  call get_param(param_file, mdl, "CHECK_DIFFUSIVE_CFL", CS%check_diffusive_CFL, &
                 "If true, use enough iterations the diffusion to ensure that the diffusive equations are stable.", &
                 default=.false.)
  call get_param(param_file, mdl, "MAX_TR_DIFFUSION_CFL", CS%max_diff_CFL, &
                 "If positive, locally limit the along-isopycnal tracer diffusivity to keep the diffusive CFL below this.", &
                 units="nondim", default=-1.0)
  call get_param(param_file, mdl, "RECALC_NEUTRAL_SURF", CS%recalc_neutral_surf, &
                 "If true, then recalculate the neutral surfaces if the CFL has been exceeded", default=.false.)
  call get_param(param_file, mdl, "HOR_DIFF_ANSWER_DATE", CS%epi%answer_date, &
                 "The vintage of the order of arithmetic to use for the tracer diffusion.", default=20240101, &
                 do_not_log=.not.CS%Diffuse_ML_interior)
  call get_param(param_file, mdl, "HOR_DIFF_LIMIT_BUG", flag, &
                 "If true and the answer date is 20240330 or below, use a rotational symmetry breaking bug when limiting the "//&
                 "tracer properties in tracer_epipycnal_ML_diff.", default=.true., &
                 do_not_log=((.not.CS%Diffuse_ML_interior).or.(CS%epi%answer_date>=20240331)))
  CS%epi%limit_bug = merge(1, 0, flag)
  CS%epi%ML_KhTr_scale = 1.0
  if (CS%Diffuse_ML_interior) then
    call get_param(param_file, mdl, "ML_KHTR_SCALE", CS%epi%ML_KhTr_scale, &
                 "With Diffuse_ML_interior, the ratio of the truly horizontal diffusivity in the mixed layer to the "//&
                 "epipycnal diffusivity.  The valid range is 0 to 1.", units="nondim", default=1.0)
    if (GV%nk_rho_varies < 1 .or. GV%nk_rho_varies >= GV%ke) call MOM_error(FATAL, "tracer_hor_diff_init (HIP): "// &
      "DIFFUSE_ML_TO_INTERIOR is provided for layered runs with variable-density layers above an interior (0 < nk_rho_varies < nk).")
    call mom6hip_read_eos(param_file, CS%eos, "tracer_hor_diff_init")
  endif
  ! neutral_diffusion_init :138-330
  call get_param(param_file, "MOM_neutral_diffusion", "USE_NEUTRAL_DIFFUSION", CS%use_neutral_diffusion, &
                 "If true, enables the neutral diffusion module.", default=.false.)
  if (CS%use_neutral_diffusion) then
    call get_param(param_file, "MOM_neutral_diffusion", "NDIFF_CONTINUOUS", flag, &
                   "If true, uses a continuous reconstruction of T and S when finding neutral surfaces.", default=.true.)
    call refuse(.not.flag, "NDIFF_CONTINUOUS = False")
    call get_param(param_file, "MOM_neutral_diffusion", "NDIFF_REF_PRES", CS%nd%ref_pres, &
                   "The reference pressure (Pa) used for the derivatives of the equation of state. If negative (default), "//&
                   "local pressure is used.", units="Pa", default=-1., scale=US%Pa_to_RL2_T2)
    call get_param(param_file, "MOM_neutral_diffusion", "NDIFF_INTERIOR_ONLY", flag, &
                   "If true, only applies neutral diffusion in the ocean interior. That is, the algorithm will exclude the "//&
                   "surface and bottom boundary layers.", default=.false.)
    CS%nd%interior_only = merge(1, 0, flag)
    if (flag) then
      call get_param(param_file, "MOM_neutral_diffusion", "NDIFF_TAPERING", flag, default=.false.)
      call refuse(flag, "NDIFF_TAPERING")
    endif
    call get_param(param_file, "MOM_neutral_diffusion", "NDIFF_USE_UNMASKED_TRANSPORT_BUG", flag, default=.false.)
    call refuse(flag, "NDIFF_USE_UNMASKED_TRANSPORT_BUG")
    call get_param(param_file, "MOM_neutral_diffusion", "NDIFF_ANSWER_DATE", CS%nd%ndiff_answer_date, &
                   "The vintage of the order of arithmetic to use for the neutral diffusion.", default=20240101)
    call mom6hip_read_eos(param_file, CS%eos, "neutral_diffusion_init")
    CS%nd%initialized = 1
  endif
  if (CS%use_neutral_diffusion .and. CS%Diffuse_ML_interior) call MOM_error(FATAL, "MOM_tracer_hor_diff: "// &
       "USE_NEUTRAL_DIFFUSION and DIFFUSE_ML_TO_INTERIOR are mutually exclusive!")
  call get_param(param_file, mdl, "USE_HORIZONTAL_BOUNDARY_DIFFUSION", CS%use_hor_bnd_diffusion, default=.false.)
  call refuse(CS%use_hor_bnd_diffusion, "USE_HORIZONTAL_BOUNDARY_DIFFUSION")
  call mom6hip_read_topology(param_file)
  call mom6hip_read_resident(param_file)
  id_clock_diffuse = cpu_clock_id('(Ocean diffuse tracer)', grain=CLOCK_MODULE)
contains
  subroutine refuse(on, name)
    logical,          intent(in) :: on
    character(len=*), intent(in) :: name
    if (on) call MOM_error(FATAL, "tracer_hor_diff_init (HIP): "//name//" is not provided by the GPU path.")
  end subroutine refuse
end subroutine tracer_hor_diff_init

!> Same interface as the reference tracer_hor_diff_end (:1772)
subroutine tracer_hor_diff_end(CS)
  type(tracer_hor_diff_CS), pointer :: CS
  if (associated(CS)) deallocate(CS)
end subroutine tracer_hor_diff_end

end module MOM_tracer_hor_diff
