!> Stand-ins the reference's MOM_set_viscosity.F90 needs beyond tests/fortran/stubs/mom6_stubs.F90 when it is compiled in place beside the oracle
!! (tests/test_reference_kernels.py): the "is this parameterisation on" queries of the shear-mixing and CVMix modules (set_visc_register_restarts
!! and set_visc_init ask them: USE_JACKSON_PARAM decides, for one, the default of CHANNEL_DRAG_MAX_BBL_THICK and whether visc%Kv_shear exists).
!! They read the same parameters under the same names as the reference's functions (MOM_kappa_shear.F90:2009-2041, MOM_CVMix_conv.F90:303,
!! MOM_CVMix_ddiff.F90:276, MOM_CVMix_shear.F90:346); the modules themselves -- the mixing schemes -- are not part of the build.
module MOM_cvmix_conv
use MOM_file_parser, only : param_file_type, get_param
implicit none ; private
public :: cvmix_conv_is_used
contains
logical function cvmix_conv_is_used(param_file)
  type(param_file_type), intent(in) :: param_file
  call get_param(param_file, "MOM_CVMix_conv", "USE_CVMix_CONVECTION", cvmix_conv_is_used, default=.false., do_not_log=.true.)
end function cvmix_conv_is_used
end module MOM_cvmix_conv

module MOM_CVMix_ddiff
use MOM_file_parser, only : param_file_type, get_param
implicit none ; private
public :: CVMix_ddiff_is_used
contains
logical function CVMix_ddiff_is_used(param_file)
  type(param_file_type), intent(in) :: param_file
  call get_param(param_file, "MOM_CVMix_ddiff", "USE_CVMIX_DDIFF", CVMix_ddiff_is_used, default=.false., do_not_log=.true.)
end function CVMix_ddiff_is_used
end module MOM_CVMix_ddiff

module MOM_cvmix_shear
use MOM_file_parser, only : param_file_type, get_param
implicit none ; private
public :: cvmix_shear_is_used
contains
logical function cvmix_shear_is_used(param_file)
  type(param_file_type), intent(in) :: param_file
  logical :: LMD94, PP81
  call get_param(param_file, "MOM_CVMix_shear", "USE_LMD94", LMD94, default=.false., do_not_log=.true.)
  call get_param(param_file, "MOM_CVMix_shear", "USE_PP81", PP81, default=.false., do_not_log=.true.)
  cvmix_shear_is_used = (LMD94 .or. PP81)
end function cvmix_shear_is_used
end module MOM_cvmix_shear

module MOM_kappa_shear
use MOM_file_parser, only : param_file_type, get_param
implicit none ; private
public :: kappa_shear_is_used, kappa_shear_at_vertex
contains
logical function kappa_shear_is_used(param_file)
  type(param_file_type), intent(in) :: param_file
  call get_param(param_file, "MOM_kappa_shear", "USE_JACKSON_PARAM", kappa_shear_is_used, default=.false., do_not_log=.true.)
end function kappa_shear_is_used
logical function kappa_shear_at_vertex(param_file)
  type(param_file_type), intent(in) :: param_file
  logical :: on
  kappa_shear_at_vertex = .false.
  call get_param(param_file, "MOM_kappa_shear", "USE_JACKSON_PARAM", on, default=.false., do_not_log=.true.)
  if (on) call get_param(param_file, "MOM_kappa_shear", "VERTEX_SHEAR", kappa_shear_at_vertex, default=.false., do_not_log=.true.)
end function kappa_shear_at_vertex
end module MOM_kappa_shear
