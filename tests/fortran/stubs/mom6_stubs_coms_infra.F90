!> One-PE stand-ins for the two modules the reference's src/framework/MOM_coms.F90 imports (MOM_coms_infra, which ends in FMS, and
!! MOM_error_handler), so that MOM_coms.F90 -- reproducing_sum and the extended-fixed-point arithmetic behind it -- can be compiled in place and
!! run beside oracle/coms.c (tests/test_reference_kernels.py).  Sums across PEs leave their argument as it is.  Test infrastructure only.
module MOM_error_handler
implicit none ; private
public :: MOM_error, MOM_mesg, FATAL, WARNING, NOTE
integer, parameter :: NOTE = 0, WARNING = 1, FATAL = 2
contains
subroutine MOM_error(level, message, all_print)
  integer,           intent(in) :: level
  character(len=*),  intent(in) :: message
  logical, optional, intent(in) :: all_print
  write(0,'(a)') trim(message)
  if (level == FATAL) error stop 1
end subroutine MOM_error
subroutine MOM_mesg(message, verb, all_print)
  character(len=*),  intent(in) :: message
  integer, optional, intent(in) :: verb
  logical, optional, intent(in) :: all_print
end subroutine MOM_mesg
end module MOM_error_handler

module MOM_coms_infra
use, intrinsic :: iso_fortran_env, only : int32, int64
implicit none ; private
public :: PE_here, root_PE, num_PEs, set_rootPE, Set_PElist, Get_PElist, broadcast, field_chksum, MOM_infra_init, MOM_infra_end
public :: sum_across_PEs, max_across_PEs, min_across_PEs, all_across_PEs, any_across_PEs, sync_PEs
interface sum_across_PEs
  module procedure sum_int64_1d, sum_int64_2d, sum_real_1d, sum_real_0d, sum_int_0d
end interface
interface max_across_PEs ; module procedure max_real_0d, max_int_0d ; end interface
interface min_across_PEs ; module procedure min_real_0d, min_int_0d ; end interface
interface broadcast ; module procedure bc_char, bc_int_0d, bc_real_0d, bc_int64_0d ; end interface
interface field_chksum ; module procedure chk_real_2d ; end interface
contains
integer function PE_here() ; PE_here = 0 ; end function PE_here
integer function root_PE() ; root_PE = 0 ; end function root_PE
integer function num_PEs() ; num_PEs = 1 ; end function num_PEs
subroutine set_rootPE(pe) ; integer, intent(in) :: pe ; end subroutine set_rootPE
subroutine Set_PElist(pelist, no_sync)
  integer, optional, intent(in) :: pelist(:)
  logical, optional, intent(in) :: no_sync
end subroutine Set_PElist
subroutine Get_PElist(pelist, name, commID)
  integer,                    intent(out) :: pelist(:)
  character(len=*), optional, intent(out) :: name
  integer,          optional, intent(out) :: commID
  pelist(:) = 0
  if (present(name)) name = ""
  if (present(commID)) commID = 0
end subroutine Get_PElist
subroutine MOM_infra_init(localcomm) ; integer, optional, intent(in) :: localcomm ; end subroutine MOM_infra_init
subroutine MOM_infra_end() ; end subroutine MOM_infra_end
subroutine sync_PEs(pelist) ; integer, optional, intent(in) :: pelist(:) ; end subroutine sync_PEs
subroutine sum_int64_1d(field, length, pelist)
  integer(kind=int64), dimension(:), intent(inout) :: field ; integer, intent(in) :: length ; integer, optional, intent(in) :: pelist(:)
end subroutine sum_int64_1d
subroutine sum_int64_2d(field, length, pelist)
  integer(kind=int64), dimension(:,:), intent(inout) :: field ; integer, intent(in) :: length ; integer, optional, intent(in) :: pelist(:)
end subroutine sum_int64_2d
subroutine sum_real_1d(field, length, pelist)
  real, dimension(:), intent(inout) :: field ; integer, intent(in) :: length ; integer, optional, intent(in) :: pelist(:)
end subroutine sum_real_1d
subroutine sum_real_0d(field, pelist)
  real, intent(inout) :: field ; integer, optional, intent(in) :: pelist(:)
end subroutine sum_real_0d
subroutine sum_int_0d(field, pelist)
  integer, intent(inout) :: field ; integer, optional, intent(in) :: pelist(:)
end subroutine sum_int_0d
subroutine max_real_0d(field, pelist) ; real, intent(inout) :: field ; integer, optional, intent(in) :: pelist(:) ; end subroutine max_real_0d
subroutine max_int_0d(field, pelist) ; integer, intent(inout) :: field ; integer, optional, intent(in) :: pelist(:) ; end subroutine max_int_0d
subroutine min_real_0d(field, pelist) ; real, intent(inout) :: field ; integer, optional, intent(in) :: pelist(:) ; end subroutine min_real_0d
subroutine min_int_0d(field, pelist) ; integer, intent(inout) :: field ; integer, optional, intent(in) :: pelist(:) ; end subroutine min_int_0d
logical function all_across_PEs(field, pelist)
  logical, intent(in) :: field ; integer, optional, intent(in) :: pelist(:)
  all_across_PEs = field
end function all_across_PEs
logical function any_across_PEs(field, pelist)
  logical, intent(in) :: field ; integer, optional, intent(in) :: pelist(:)
  any_across_PEs = field
end function any_across_PEs
subroutine bc_char(dat, length, from_PE, PElist, blocking)
  character(len=*), intent(inout) :: dat(:) ; integer, intent(in) :: length
  integer, optional, intent(in) :: from_PE, PElist(:) ; logical, optional, intent(in) :: blocking
end subroutine bc_char
subroutine bc_int_0d(dat, from_PE, PElist, blocking)
  integer, intent(inout) :: dat ; integer, optional, intent(in) :: from_PE, PElist(:) ; logical, optional, intent(in) :: blocking
end subroutine bc_int_0d
subroutine bc_int64_0d(dat, from_PE, PElist, blocking)
  integer(kind=int64), intent(inout) :: dat ; integer, optional, intent(in) :: from_PE, PElist(:) ; logical, optional, intent(in) :: blocking
end subroutine bc_int64_0d
subroutine bc_real_0d(dat, from_PE, PElist, blocking)
  real, intent(inout) :: dat ; integer, optional, intent(in) :: from_PE, PElist(:) ; logical, optional, intent(in) :: blocking
end subroutine bc_real_0d
function chk_real_2d(field, pelist, mask_val) result(chksum)
  real, dimension(:,:), intent(in) :: field ; integer, optional, intent(in) :: pelist(:) ; real, optional, intent(in) :: mask_val
  integer(kind=int64) :: chksum
  chksum = 0
end function chk_real_2d
end module MOM_coms_infra
