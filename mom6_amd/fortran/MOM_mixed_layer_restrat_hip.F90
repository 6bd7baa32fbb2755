!> Drop-in replacement for module MOM_mixed_layer_restrat (src/parameterizations/lateral/MOM_mixed_layer_restrat.F90):
!! mixedlayer_restrat (:135), mixedlayer_restrat_init (:1532) and mixedlayer_restrat_register_restarts (:1794) with the
!! reference's dummy-argument lists, so step_MOM_thermo / initialize_MOM (src/core/MOM.F90:1335, :2854, :3312) compile unchanged.
!! Provided: the Fox-Kemper et al. (2008) restratification in general coordinates (mixedlayer_restrat_OM4: MLE_DENSITY_DIFF or
!! MLE_USE_PBL_MLD, MLE_MLD_STRETCH, MLE_MLD_DECAY_TIME / _TIME2 with their restart fields, FOX_KEMPER_ML_RESTRAT_COEF / _COEF2,
!! MLE_FRONT_LENGTH, MLE_TAIL_DH) and with a bulk mixed layer (mixedlayer_restrat_BML), Boussinesq -- on the GPU through
!! libmom6hip (mom6hip_mixedlayer_restrat, HOST memspace).  USE_BODNER23, USE_STANLEY_ML and non-Boussinesq mode stop with a FATAL
!! error; the diagnostics are not registered.
!!
!! Compiled INSIDE a MOM6 source tree in place of src/parameterizations/lateral/MOM_mixed_layer_restrat.F90; here against
!! tests/fortran/stubs.
module MOM_mixed_layer_restrat

use, intrinsic :: iso_c_binding
use mom6hip_c_api
use mom6hip_MOM_glue,          only : mom6hip_shared_context, mom6hip_read_topology, mom6hip_read_eos, mom6hip_fatal_if
use mom6hip_MOM_glue,          only : mom6hip_read_resident, mom6hip_resident, mom6hip_mirror
use MOM_diag_mediator,         only : diag_ctrl, time_type
use MOM_domains,               only : pass_var
use MOM_error_handler,         only : MOM_error, FATAL
use MOM_file_parser,           only : get_param, log_version, param_file_type
use MOM_file_parser,           only : openParameterBlock, closeParameterBlock
use MOM_forcing_type,          only : mech_forcing
use MOM_grid,                  only : ocean_grid_type
use MOM_hor_index,             only : hor_index_type
use MOM_lateral_mixing_coeffs, only : VarMix_CS
use MOM_restart,               only : register_restart_field, query_initialized, MOM_restart_CS
use MOM_unit_scaling,          only : unit_scale_type
use MOM_variables,             only : thermo_var_ptrs
use MOM_verticalGrid,          only : verticalGrid_type
implicit none ; private

#include <MOM_memory.h>

public mixedlayer_restrat
public mixedlayer_restrat_init
public mixedlayer_restrat_register_restarts
public mixedlayer_restrat_unit_tests

!> Control structure (the members of the reference's mixedlayer_restrat_CS, :40-126, that the provided branches read)
type, public :: mixedlayer_restrat_CS ; private
  logical :: initialized = .false. !< True if this control structure has been initialized.
  real    :: ml_restrat_coef = 0.0, ml_restrat_coef2 = 0.0, front_length = 0.0, vonKar = 0.41
  logical :: MLE_use_PBL_MLD = .false.
  real    :: MLE_MLD_decay_time = 0.0, MLE_MLD_decay_time2 = 0.0, MLE_density_diff = 0.03, MLE_tail_dh = 0.0, MLE_MLD_stretch = 1.0
  logical :: use_Bodner = .false., use_Stanley_ML = .false.
  real    :: ustar_min = 0.0       !< A minimum value of ustar in thickness units [H T-1 ~> m s-1 or kg m-2 s-1]
  real    :: Kv_restrat = 0.0      !< read and logged; the reference's live code does not use it (growth_time is commented out)
  type(mom6hip_eos_t) :: eos       !< the equation of state, as read from the parameter file
  type(diag_ctrl), pointer :: diag => NULL()
  real, dimension(:,:), allocatable :: &
         MLD_filtered, &           !< Time-filtered MLD [H ~> m or kg m-2]
         MLD_filtered_slow         !< Slower time-filtered MLD [H ~> m or kg m-2]
end type mixedlayer_restrat_CS

character(len=40)  :: mdl = "MOM_mixed_layer_restrat" !< This module's name.

contains

!> Same interface as the reference mixedlayer_restrat (:135).
subroutine mixedlayer_restrat(h, uhtr, vhtr, tv, forces, dt, MLD, h_MLD, bflux, VarMix, G, GV, US, CS)
  type(ocean_grid_type),                      intent(inout) :: G
  type(verticalGrid_type),                    intent(in)    :: GV
  type(unit_scale_type),                      intent(in)    :: US
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  target, intent(inout) :: h
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(inout) :: uhtr
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(inout) :: vhtr
  type(thermo_var_ptrs),                      intent(in)    :: tv
  type(mech_forcing),                         intent(in)    :: forces
  real,                                       intent(in)    :: dt
  real, dimension(:,:),                       pointer       :: MLD
  real, dimension(:,:),                       pointer       :: h_MLD
  real, dimension(:,:),                       pointer       :: bflux
  type(VarMix_CS), target,                    intent(in)    :: VarMix
  type(mixedlayer_restrat_CS), target,        intent(inout) :: CS

  type(mom6hip_mixedlayer_restrat_cs_t) :: ccs
  type(c_ptr) :: p_hMLD, ctx
  integer :: n2
  integer :: rc

  if (.not. CS%initialized) call MOM_error(FATAL, "mixedlayer_restrat: "// &
         "Module must be initialized before it is used.")
  if (.not.GV%Boussinesq) call MOM_error(FATAL, "mixedlayer_restrat (HIP): non-Boussinesq mode is not provided by the GPU path.")
  if (.not.associated(tv%eqn_of_state)) call MOM_error(FATAL, "mixedlayer_restrat: "// &
         "An equation of state must be used with this module.")
  if (.not.(associated(tv%T) .and. associated(tv%S))) call MOM_error(FATAL, "mixedlayer_restrat (HIP): tv%T and tv%S are needed.")
  if (.not.associated(forces%ustar)) call MOM_error(FATAL, "mixedlayer_restrat (HIP): "// &
         "forces%ustar must be associated (find_ustar from tau_mag is not provided by the GPU path).")

  ccs%ml_restrat_coef = CS%ml_restrat_coef ; ccs%ml_restrat_coef2 = CS%ml_restrat_coef2 ; ccs%front_length = CS%front_length
  ccs%vonKar = CS%vonKar ; ccs%MLE_MLD_decay_time = CS%MLE_MLD_decay_time ; ccs%MLE_MLD_decay_time2 = CS%MLE_MLD_decay_time2
  ccs%MLE_density_diff = CS%MLE_density_diff ; ccs%MLE_tail_dh = CS%MLE_tail_dh ; ccs%MLE_MLD_stretch = CS%MLE_MLD_stretch
  ccs%ustar_min = CS%ustar_min ; ccs%MLE_use_PBL_MLD = merge(1, 0, CS%MLE_use_PBL_MLD) ; ccs%nkml = GV%nkml ; ccs%initialized = 1
  if (GV%nkml == 0) then
    if (allocated(CS%MLD_filtered)) ccs%MLD_filtered = c_loc(CS%MLD_filtered)
    if (allocated(CS%MLD_filtered_slow)) ccs%MLD_filtered_slow = c_loc(CS%MLD_filtered_slow)
    if (CS%front_length > 0.) then
      if (.not. allocated(VarMix%Rd_dx_h)) call MOM_error(FATAL, "mixedlayer_restrat_OM4: "// &
           "The resolution argument, Rd/dx, was not associated.")
      ccs%Rd_dx_h = c_loc(VarMix%Rd_dx_h)
    endif
  endif
  p_hMLD = c_null_ptr
  if (associated(h_MLD)) then
    if (.not.is_contiguous(h_MLD)) call MOM_error(FATAL, "mixedlayer_restrat (HIP): h_MLD must be contiguous.")
    p_hMLD = c_loc(h_MLD)
  endif

  ctx = mom6hip_shared_context(G, GV)
  if (mom6hip_resident()) then      ! GPU_RESIDENT_DYNAMICS: the shared device mirrors of the host arrays; the running means of the
    n2 = size(h(:,:,1))             ! control structure too (restart fields: current on the host after mom6hip_mirrors_to_host)
    call to_dev(ccs%MLD_filtered, n2, .true.) ; call to_dev(ccs%MLD_filtered_slow, n2, .true.) ; call to_dev(ccs%Rd_dx_h, n2, .false.)
    call to_dev(p_hMLD, n2, .false.)
    rc = mom6hip_mixedlayer_restrat(ctx, ccs, mom6hip_mirror(ctx, c_loc(h), int(size(h), c_int64_t), .true., .true.), &
                                    mom6hip_mirror(ctx, c_loc(uhtr), int(size(uhtr), c_int64_t), .true., .true.), &
                                    mom6hip_mirror(ctx, c_loc(vhtr), int(size(vhtr), c_int64_t), .true., .true.), &
                                    mom6hip_mirror(ctx, c_loc(tv%T), int(size(h), c_int64_t), .true., .false.), &
                                    mom6hip_mirror(ctx, c_loc(tv%S), int(size(h), c_int64_t), .true., .false.), c_loc(CS%eos), &
                                    mom6hip_mirror(ctx, c_loc(forces%ustar), int(n2, c_int64_t), .true., .false.), dt, p_hMLD, &
                                    c_null_ptr, c_null_ptr, MOM6HIP_MEM_DEVICE)
  else
    rc = mom6hip_mixedlayer_restrat(ctx, ccs, c_loc(h), c_loc(uhtr), c_loc(vhtr), c_loc(tv%T), c_loc(tv%S), &
                                    c_loc(CS%eos), c_loc(forces%ustar), dt, p_hMLD, c_null_ptr, c_null_ptr, MOM6HIP_MEM_HOST)
  endif
  call mom6hip_fatal_if(rc, "mixedlayer_restrat")
contains
  !> a host pointer -> its device mirror (an input, or an in/out array the call writes)
  subroutine to_dev(p, n, written)
    type(c_ptr), intent(inout) :: p
    integer,     intent(in)    :: n
    logical,     intent(in)    :: written
    if (c_associated(p)) p = mom6hip_mirror(ctx, p, int(n, c_int64_t), .true., written)
  end subroutine to_dev
end subroutine mixedlayer_restrat

!> Same interface as the reference mixedlayer_restrat_init (:1532), same parameters and defaults (:1554-1735).
logical function mixedlayer_restrat_init(Time, G, GV, US, param_file, diag, CS, restart_CS)
  type(time_type),             intent(in)    :: Time
  type(ocean_grid_type),       intent(inout) :: G
  type(verticalGrid_type),     intent(in)    :: GV
  type(unit_scale_type),       intent(in)    :: US
  type(param_file_type),       intent(in)    :: param_file
  type(diag_ctrl), target,     intent(inout) :: diag
  type(mixedlayer_restrat_CS), intent(inout) :: CS
  type(MOM_restart_CS),        intent(in)    :: restart_CS
# include "version_variable.h"
  real :: omega, ustar_min_dflt

  call get_param(param_file, mdl, "MIXEDLAYER_RESTRAT", mixedlayer_restrat_init, default=.false., do_not_log=.true.)
  call log_version(param_file, mdl, version, "", all_default=.not.mixedlayer_restrat_init)
  call get_param(param_file, mdl, "MIXEDLAYER_RESTRAT", mixedlayer_restrat_init, &
             "If true, a density-gradient dependent re-stratifying flow is imposed in the mixed layer.", default=.false.)
  if (.not. mixedlayer_restrat_init) return

  CS%initialized = .true.
  CS%MLE_MLD_decay_time = -9.e9*US%s_to_T
  CS%MLE_density_diff = -9.e9*US%kg_m3_to_R
  CS%MLE_tail_dh = -9.e9
  CS%MLE_use_PBL_MLD = .false.
  CS%MLE_MLD_stretch = -9.e9
  CS%use_Stanley_ML = .false.
  CS%use_Bodner = .false.

  call openParameterBlock(param_file, 'MLE')
  if (GV%nkml==0) then
    call get_param(param_file, mdl, "USE_BODNER23", CS%use_Bodner, &
             "If true, use the Bodner et al., 2023, formulation of the re-stratifying mixed-layer restratification "//&
             "parameterization.", default=.false.)
  endif
  call closeParameterBlock(param_file)
  if (CS%use_Bodner) call MOM_error(FATAL, "mixedlayer_restrat_init (HIP): MLE%USE_BODNER23 is not provided by the GPU path.")

  call get_param(param_file, mdl, "FOX_KEMPER_ML_RESTRAT_COEF", CS%ml_restrat_coef, &
             "A nondimensional coefficient that is proportional to the ratio of the deformation radius to the "//&
             "dominant lengthscale of the submesoscale mixed layer instabilities, times the minimum of the ratio of "//&
             "the mesoscale eddy kinetic energy to the large-scale geostrophic kinetic energy or 1 plus the square of "//&
             "the grid spacing over the deformation radius, as detailed by Fox-Kemper et al. (2010)", units="nondim", default=0.0)
  call get_param(param_file, mdl, "USE_STANLEY_ML", CS%use_Stanley_ML, default=.false.)
  if (CS%use_Stanley_ML) call MOM_error(FATAL, "mixedlayer_restrat_init (HIP): USE_STANLEY_ML is not provided by the GPU path.")
  call get_param(param_file, mdl, 'VON_KARMAN_CONST', CS%vonKar, 'The value the von Karman constant as used for mixed layer viscosity.', &
                 units='nondim', default=0.41)
  if (GV%nkml==0) then
    call get_param(param_file, mdl, "FOX_KEMPER_ML_RESTRAT_COEF2", CS%ml_restrat_coef2, &
             "As for FOX_KEMPER_ML_RESTRAT_COEF but used in a second application of the MLE restratification parameterization.", &
             units="nondim", default=0.0)
    call get_param(param_file, mdl, "MLE_FRONT_LENGTH", CS%front_length, &
             "If non-zero, is the frontal-length scale used to calculate the upscaling of buoyancy gradients.", &
             units="m", default=0.0, scale=US%m_to_L)
    call get_param(param_file, mdl, "MLE_USE_PBL_MLD", CS%MLE_use_PBL_MLD, &
             "If true, the MLE parameterization will use the mixed-layer depth provided by the active PBL parameterization.", &
             default=.false.)
    call get_param(param_file, mdl, "MLE_MLD_DECAY_TIME", CS%MLE_MLD_decay_time, &
             "The time-scale for a running-mean filter applied to the mixed-layer depth used in the MLE restratification "//&
             "parameterization.", units="s", default=0., scale=US%s_to_T)
    call get_param(param_file, mdl, "MLE_MLD_DECAY_TIME2", CS%MLE_MLD_decay_time2, &
             "The time-scale for a running-mean filter applied to the filtered mixed-layer depth used in a second MLE "//&
             "restratification parameterization.", units="s", default=0., scale=US%s_to_T)
    if (.not. CS%MLE_use_PBL_MLD) then
      call get_param(param_file, mdl, "MLE_DENSITY_DIFF", CS%MLE_density_diff, &
             "Density difference used to detect the mixed-layer depth used for the mixed-layer eddy parameterization "//&
             "by Fox-Kemper et al. (2010)", units="kg/m3", default=0.03, scale=US%kg_m3_to_R)
    endif
    call get_param(param_file, mdl, "MLE_TAIL_DH", CS%MLE_tail_dh, &
             "Fraction by which to extend the mixed-layer restratification depth used for a smoother stream function at "//&
             "the base of the mixed-layer.", units="nondim", default=0.0)
    call get_param(param_file, mdl, "MLE_MLD_STRETCH", CS%MLE_MLD_stretch, &
             "A scaling coefficient for stretching/shrinking the MLD used in the MLE scheme.", units="nondim", default=1.0)
  endif
  call get_param(param_file, mdl, "KV_RESTRAT", CS%Kv_restrat, &
                 "A small viscosity that sets a floor on the momentum mixing rate during restratification.", &
                 units="m2 s-1", default=0.0, scale=GV%m2_s_to_HZ_T*(US%Z_to_m*GV%m_to_H))
  call get_param(param_file, mdl, "OMEGA", omega, "The rotation rate of the earth.", units="s-1", default=7.2921e-5, scale=US%T_to_s)
  ustar_min_dflt = 2.0e-4 * omega * (GV%Angstrom_Z + GV%dZ_subroundoff)
  call get_param(param_file, mdl, "RESTRAT_USTAR_MIN", CS%ustar_min, &
                 "The minimum value of ustar that will be used by the mixed layer restratification module.", &
                 units="m s-1", default=US%Z_to_m*US%s_to_T*ustar_min_dflt, scale=GV%m_to_H*US%T_to_s)

  CS%diag => diag
  call mom6hip_read_eos(param_file, CS%eos, "mixedlayer_restrat_init")
  call mom6hip_read_topology(param_file)
  call mom6hip_read_resident(param_file)

  ! If MLD_filtered is being used, we need to update halo regions after a restart
  if (allocated(CS%MLD_filtered)) call pass_var(CS%MLD_filtered, G%domain)
  if (allocated(CS%MLD_filtered_slow)) call pass_var(CS%MLD_filtered_slow, G%domain)
end function mixedlayer_restrat_init

!> Same interface as the reference mixedlayer_restrat_register_restarts (:1794); the Bodner fields are not registered (refused).
subroutine mixedlayer_restrat_register_restarts(HI, GV, US, param_file, CS, restart_CS)
  type(hor_index_type),        intent(in)    :: HI
  type(verticalGrid_type),     intent(in)    :: GV
  type(unit_scale_type),       intent(in)    :: US
  type(param_file_type),       intent(in)    :: param_file
  type(mixedlayer_restrat_CS), intent(inout) :: CS
  type(MOM_restart_CS),        intent(inout) :: restart_CS
  logical :: mixedlayer_restrat_init

  call get_param(param_file, mdl, "MIXEDLAYER_RESTRAT", mixedlayer_restrat_init, default=.false., do_not_log=.true.)
  if (.not. mixedlayer_restrat_init) return
  call get_param(param_file, mdl, "MLE_MLD_DECAY_TIME", CS%MLE_MLD_decay_time, units="s", default=0., scale=US%s_to_T, do_not_log=.true.)
  call get_param(param_file, mdl, "MLE_MLD_DECAY_TIME2", CS%MLE_MLD_decay_time2, units="s", default=0., scale=US%s_to_T, do_not_log=.true.)
  if (CS%MLE_MLD_decay_time>0. .or. CS%MLE_MLD_decay_time2>0.) then
    allocate(CS%MLD_filtered(HI%isd:HI%ied,HI%jsd:HI%jed), source=0.)
    call register_restart_field(CS%MLD_filtered, "MLD_MLE_filtered", .false., restart_CS, &
                                longname="Time-filtered MLD for use in MLE", units="m")
  endif
  if (CS%MLE_MLD_decay_time2>0.) then
    allocate(CS%MLD_filtered_slow(HI%isd:HI%ied,HI%jsd:HI%jed), source=0.)
    call register_restart_field(CS%MLD_filtered_slow, "MLD_MLE_filtered_slow", .false., restart_CS, &
                                longname="Slower time-filtered MLD for use in MLE", units="m")
  endif
end subroutine mixedlayer_restrat_register_restarts

!> Same interface as the reference mixedlayer_restrat_unit_tests (:1846; imported by MOM_unit_tests.F90 and the unit-test driver):
!! the ten known answers of the shape function mu(sigma, dh) (:1855-1874), evaluated by the library's kernel on the GPU
!! (mom6hip_mixedlayer_restrat_mu).  The four answers of rmean2ts (:1878-1885) belong to the Bodner et al. (2023) form, which this
!! shim refuses (MLE%USE_BODNER23), and are not evaluated.  Returns true if a test fails.
logical function mixedlayer_restrat_unit_tests(verbose)
  logical, intent(in) :: verbose
  real, parameter :: sig(10)  = (/ 3., 0., -0.25, -0.5, -0.75, -1., -3., -0.5, -1., -1.5 /)
  real, parameter :: dh(10)   = (/ 0., 0., 0., 0., 0., 0., 0., 0.5, 0.5, 0.5 /)
  real, parameter :: want(10) = (/ 0., 0., 0.7946428571428572, 1., 0.7946428571428572, 0., 0., 1., 0.25, 0. /)
  real :: got, tol
  integer :: n
  print *,'===== mixedlayer_restrat: mixedlayer_restrat_unit_tests =================='
  mixedlayer_restrat_unit_tests = .false.
  do n=1,10
    got = mom6hip_mixedlayer_restrat_mu(real(sig(n), c_double), real(dh(n), c_double))
    tol = 0. ; if (n == 3 .or. n == 5) tol = epsilon(1.)
    if (abs(got - want(n)) > tol) then
      mixedlayer_restrat_unit_tests = .true.
      write(0,'(a,2f8.3,a,es23.15,a,es23.15)') "mixedlayer_restrat_unit_tests: mu(", sig(n), dh(n), ") = ", got, " but expected ", want(n)
    elseif (verbose) then
      write(*,'(a,2f8.3,a,es23.15)') "  mu(", sig(n), dh(n), ") = ", got
    endif
  enddo
  if (.not. mixedlayer_restrat_unit_tests) print '(a)','  Passed tests of mu(z)'
end function mixedlayer_restrat_unit_tests

end module MOM_mixed_layer_restrat
