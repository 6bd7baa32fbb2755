!> Stand-ins the reference's MOM_open_boundary.F90 needs beyond tests/fortran/stubs/mom6_stubs.F90 when it is compiled in place beside the
!! oracle (-DREF_OBC, tests/test_reference_kernels.py): the grid type of the initialisation, external-field interpolation and the
!! obsolete-parameter checks, none of which a test drives.  Declarations with the reference's argument lists; the procedures do nothing or stop.
!! Nothing of this is used by the library or its shims.
#include <MOM_memory.h>

module MOM_dyn_horgrid      ! src/framework/MOM_dyn_horgrid.F90: the grid as the initialisation sees it (open_boundary_config, the land-mask routines)
use MOM_domains,   only : MOM_domain_type
use MOM_hor_index, only : hor_index_type
implicit none ; private
public :: dyn_horgrid_type
type :: dyn_horgrid_type
  type(MOM_domain_type), pointer :: Domain => NULL()
  type(hor_index_type) :: HI
  integer :: isc, iec, jsc, jec, isd, ied, jsd, jed, IscB, IecB, JscB, JecB, IsdB, IedB, JsdB, JedB
  integer :: isg, ieg, jsg, jeg, IsgB, IegB, JsgB, JegB
  integer :: idg_offset = 0, jdg_offset = 0
  logical :: symmetric = .true.
  real :: max_depth = 0.0, Z_ref = 0.0
  real, allocatable, dimension(:,:) :: areaT, mask2dT, mask2dCu, mask2dCv, mask2dBu, bathyT, areaCu, areaCv, dxCu, dyCu, dxCv, dyCv, dy_Cu, dx_Cv
  real, allocatable, dimension(:,:) :: OBCmaskCu, OBCmaskCv, IareaCu, IareaCv, geoLonT, geoLatT, geoLonCu, geoLatCu, geoLonCv, geoLatCv, geoLonBu, geoLatBu
end type dyn_horgrid_type
end module MOM_dyn_horgrid

module MOM_interpolate      ! src/framework/MOM_interpolate.F90: segment data from files (never read by the tests: the segments' data are set by the driver)
use MOM_domains,      only : MOM_domain_type
use MOM_time_manager, only : time_type
implicit none ; private
public :: external_field, init_external_field, time_interp_external, time_interp_external_init
type :: external_field
  integer :: id = -1
end type external_field
interface time_interp_external
  module procedure time_interp_external_0d, time_interp_external_2d, time_interp_external_3d
end interface
contains
subroutine time_interp_external_init()
end subroutine time_interp_external_init
function init_external_field(file, fieldname, MOM_domain, domain, verbose, threading, ierr, ignore_axis_atts, correct_leap_year_inconsistency)
  character(len=*),                intent(in)  :: file, fieldname
  type(MOM_domain_type), optional, intent(in)  :: MOM_domain
  integer,               optional, intent(in)  :: domain, threading
  logical,               optional, intent(in)  :: verbose, ignore_axis_atts, correct_leap_year_inconsistency
  integer,               optional, intent(out) :: ierr
  type(external_field) :: init_external_field
  init_external_field%id = -1
  error stop "init_external_field stand-in: segment data from files are not provided"
end function init_external_field
subroutine time_interp_external_0d(field, time, data_in, verbose, scale)
  type(external_field), intent(in)    :: field
  type(time_type),      intent(in)    :: time
  real,                 intent(inout) :: data_in
  logical,    optional, intent(in)    :: verbose
  real,       optional, intent(in)    :: scale
  error stop "time_interp_external stand-in: not provided"
end subroutine time_interp_external_0d
subroutine time_interp_external_2d(field, time, data_in, interp, verbose, horz_interp, mask_out, turns, scale)
  type(external_field), intent(in)    :: field
  type(time_type),      intent(in)    :: time
  real, dimension(:,:), intent(inout) :: data_in
  integer,    optional, intent(in)    :: interp, horz_interp, turns
  logical,    optional, intent(in)    :: verbose
  logical, dimension(:,:), optional, intent(out) :: mask_out
  real,       optional, intent(in)    :: scale
  error stop "time_interp_external stand-in: not provided"
end subroutine time_interp_external_2d
subroutine time_interp_external_3d(field, time, data_in, interp, verbose, horz_interp, mask_out, turns, scale)
  type(external_field), intent(in)    :: field
  type(time_type),      intent(in)    :: time
  real, dimension(:,:,:), intent(inout) :: data_in
  integer,    optional, intent(in)    :: interp, horz_interp, turns
  logical,    optional, intent(in)    :: verbose
  logical, dimension(:,:,:), optional, intent(out) :: mask_out
  real,       optional, intent(in)    :: scale
  error stop "time_interp_external stand-in: not provided"
end subroutine time_interp_external_3d
end module MOM_interpolate

module MOM_obsolete_params      ! src/diagnostics/MOM_obsolete_params.F90: the checks for retired parameter names (no test sets one)
use MOM_file_parser, only : param_file_type
implicit none ; private
public :: obsolete_logical, obsolete_int, obsolete_real, obsolete_char
contains
subroutine obsolete_logical(param_file, varname, warning_val, hint)
  type(param_file_type),      intent(in) :: param_file
  character(len=*),           intent(in) :: varname
  logical,          optional, intent(in) :: warning_val
  character(len=*), optional, intent(in) :: hint
end subroutine obsolete_logical
subroutine obsolete_char(param_file, varname, warning_val, hint)
  type(param_file_type),      intent(in) :: param_file
  character(len=*),           intent(in) :: varname
  character(len=*), optional, intent(in) :: warning_val, hint
end subroutine obsolete_char
subroutine obsolete_real(param_file, varname, warning_val, hint, only_warn)
  type(param_file_type),      intent(in) :: param_file
  character(len=*),           intent(in) :: varname
  real,             optional, intent(in) :: warning_val
  character(len=*), optional, intent(in) :: hint
  logical,          optional, intent(in) :: only_warn
end subroutine obsolete_real
subroutine obsolete_int(param_file, varname, warning_val, hint)
  type(param_file_type),      intent(in) :: param_file
  character(len=*),           intent(in) :: varname
  integer,          optional, intent(in) :: warning_val
  character(len=*), optional, intent(in) :: hint
end subroutine obsolete_int
end module MOM_obsolete_params
