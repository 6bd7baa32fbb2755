!> Stand-ins the reference's MOM_set_viscosity.F90 needs beyond tests/fortran/stubs/mom6_stubs.F90 when it is compiled in place beside the oracle
!! (tests/test_reference_kernels.py): the "is this parameterisation on" queries of the shear-mixing and CVMix modules (set_visc_register_restarts
!! asks them which restart fields to register).  All answer .false.: none of those modules is part of the build.
module MOM_cvmix_conv
use MOM_file_parser, only : param_file_type
implicit none ; private
public :: cvmix_conv_is_used
contains
logical function cvmix_conv_is_used(param_file)
  type(param_file_type), intent(in) :: param_file
  cvmix_conv_is_used = .false.
end function cvmix_conv_is_used
end module MOM_cvmix_conv

module MOM_CVMix_ddiff
use MOM_file_parser, only : param_file_type
implicit none ; private
public :: CVMix_ddiff_is_used
contains
logical function CVMix_ddiff_is_used(param_file)
  type(param_file_type), intent(in) :: param_file
  CVMix_ddiff_is_used = .false.
end function CVMix_ddiff_is_used
end module MOM_CVMix_ddiff

module MOM_cvmix_shear
use MOM_file_parser, only : param_file_type
implicit none ; private
public :: cvmix_shear_is_used
contains
logical function cvmix_shear_is_used(param_file)
  type(param_file_type), intent(in) :: param_file
  cvmix_shear_is_used = .false.
end function cvmix_shear_is_used
end module MOM_cvmix_shear

module MOM_kappa_shear
use MOM_file_parser, only : param_file_type
implicit none ; private
public :: kappa_shear_is_used, kappa_shear_at_vertex
contains
logical function kappa_shear_is_used(param_file)
  type(param_file_type), intent(in) :: param_file
  kappa_shear_is_used = .false.
end function kappa_shear_is_used
logical function kappa_shear_at_vertex(param_file)
  type(param_file_type), intent(in) :: param_file
  kappa_shear_at_vertex = .false.
end function kappa_shear_at_vertex
end module MOM_kappa_shear
