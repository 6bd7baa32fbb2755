!> TYPE-ONLY stand-ins for the MOM6 framework modules the shims of mom6_amd/fortran `use`, so that this repository can
!! compile and drive the shims on one PE without a MOM6 source tree.  They carry only the members and procedures the
!! shims touch -- and, since round 4, the members, constants and procedure interfaces (names, dummy names, optional arguments)
!! that the reference's own callers of the shims touch, so that /root/reference/src/core/MOM_dynamics_split_RK2.F90,
!! MOM_continuity.F90 and MOM_PressureForce.F90 compile UNMODIFIED, where they lie, against the shims' .mod files
!! (tests/test_reference_callers.py).  They do no model work, pin nothing about the reference and are never used as an
!! oracle.  Inside a MOM6 tree the real modules take their place.

module MOM_error_handler
implicit none ; private
public :: MOM_error, MOM_mesg, FATAL, WARNING, NOTE, is_root_pe, assert
public :: MOM_set_verbosity, callTree_showQuery, callTree_enter, callTree_leave, callTree_waypoint
integer, parameter :: NOTE = 0, WARNING = 1, FATAL = 2
contains
subroutine MOM_set_verbosity(verb)
  integer, intent(in) :: verb
end subroutine MOM_set_verbosity
subroutine assert(logical_arg, msg)
  logical, intent(in) :: logical_arg
  character(len=*), intent(in) :: msg
  if (.not.logical_arg) call MOM_error(FATAL, msg)
end subroutine assert
logical function callTree_showQuery()
  callTree_showQuery = .false.
end function callTree_showQuery
subroutine callTree_enter(mesg, n)
  character(len=*),  intent(in) :: mesg
  integer, optional, intent(in) :: n
end subroutine callTree_enter
subroutine callTree_leave(mesg)
  character(len=*),  intent(in) :: mesg
end subroutine callTree_leave
subroutine callTree_waypoint(mesg, n)
  character(len=*),  intent(in) :: mesg
  integer, optional, intent(in) :: n
end subroutine callTree_waypoint
subroutine MOM_error(level, message, all_print)
  integer,           intent(in) :: level
  character(len=*),  intent(in) :: message
  logical, optional, intent(in) :: all_print
  if (level == FATAL) then
    write(0,'(a)') "FATAL: "//trim(message) ; error stop 1
  elseif (level == WARNING) then
    write(0,'(a)') "WARNING: "//trim(message)
  else
    write(0,'(a)') "NOTE: "//trim(message)
  endif
end subroutine MOM_error
subroutine MOM_mesg(message, verb, all_print)
  character(len=*),  intent(in) :: message
  integer, optional, intent(in) :: verb
  logical, optional, intent(in) :: all_print
  write(0,'(a)') trim(message)
end subroutine MOM_mesg
logical function is_root_pe()
  is_root_pe = .true.
end function is_root_pe
end module MOM_error_handler

#ifdef REF_ALE
! the reference's own MOM_string_functions.F90 (no dependencies), compiled where it lies: MOM_regridding parses coordinate definitions with it
#include "MOM_string_functions.F90"
#else
module MOM_string_functions
implicit none ; private
public :: uppercase
contains
function uppercase(input_string)
  character(len=*), intent(in) :: input_string
  character(len=len(input_string)) :: uppercase
  integer :: k, c
  uppercase = input_string
  do k = 1, len_trim(input_string)
    c = iachar(input_string(k:k))
    if (c >= iachar('a') .and. c <= iachar('z')) uppercase(k:k) = achar(c - 32)
  enddo
end function uppercase
end module MOM_string_functions
#endif

module MOM_coms
implicit none ; private
public :: num_PEs, PE_here, sum_across_PEs, min_across_PEs, max_across_PEs, Set_PElist, Get_PElist
interface sum_across_PEs
  module procedure sum_int_1d, sum_int_0d
end interface
interface min_across_PEs
  module procedure min_real_0d
end interface
interface max_across_PEs
  module procedure max_real_0d
end interface
contains
integer function num_PEs() ; num_PEs = 1 ; end function num_PEs
integer function PE_here() ; PE_here = 0 ; end function PE_here
subroutine Set_PElist(pelist, no_sync)      ! (one PE: nothing to set)
  integer, optional, intent(in) :: pelist(:)
  logical, optional, intent(in) :: no_sync
end subroutine Set_PElist
subroutine Get_PElist(pelist, name, commID)
  integer,                    intent(out) :: pelist(:)
  character(len=*), optional, intent(out) :: name
  integer,          optional, intent(out) :: commID
  pelist(:) = 0
  if (present(name)) name = "one PE"
  if (present(commID)) commID = 0
end subroutine Get_PElist
subroutine sum_int_0d(field)      ! (one PE)
  integer, intent(inout) :: field
end subroutine sum_int_0d
subroutine sum_int_1d(field, length)
  integer, intent(inout) :: field(:)
  integer, intent(in)    :: length
end subroutine sum_int_1d
subroutine min_real_0d(field)
  real, intent(inout) :: field
end subroutine min_real_0d
subroutine max_real_0d(field)
  real, intent(inout) :: field
end subroutine max_real_0d
end module MOM_coms

module MOM_domains
use MOM_coms, only : sum_across_PEs, min_across_PEs, max_across_PEs
implicit none ; private
public :: sum_across_PEs, min_across_PEs, max_across_PEs
public :: MOM_domain_type, pass_var, pass_vector, CENTER, EAST_FACE, NORTH_FACE, CORNER, group_pass_type
public :: To_East, To_West, To_North, To_South, To_All, Omit_Corners, AGRID, BGRID_NE, CGRID_NE, SCALAR_PAIR
public :: create_group_pass, do_group_pass, start_group_pass, complete_group_pass, clone_MOM_domain, deallocate_MOM_domain
integer, parameter :: CENTER = 0, EAST_FACE = 1, NORTH_FACE = 2, CORNER = 3
integer, parameter :: To_East = 1, To_West = 2, To_North = 4, To_South = 8, To_All = 15, Omit_Corners = 16
integer, parameter :: AGRID = 0, BGRID_NE = 1, CGRID_NE = 2, SCALAR_PAIR = 32
type :: MOM_domain_type
  logical :: reentrant(2) = .false.   !< this stand-in's one-PE topology: its pass_var wraps re-entrant directions
  integer :: nihalo = 0, njhalo = 0, niglobal = 0, njglobal = 0
  logical :: symmetric = .true.
end type MOM_domain_type
!> the fields of a group pass (one PE: do_group_pass wraps each of them in the re-entrant directions, as pass_var does).  As in FMS, a call of
!! create_group_pass after the group has been used replaces the fields one after the other, in the order of the first round of calls
!! (mpp_reset_group_update_field): the reference creates its groups anew in every call of btstep / step_MOM_dyn_split_RK2 on that call's arrays.
type :: group_entry
  real, pointer :: a2(:,:) => NULL(), a3(:,:,:) => NULL()
  integer :: pos = 0
end type group_entry
type :: group_pass_type
  integer :: n = 0, ridx = 0
  logical :: used = .false.
  type(group_entry) :: e(24)
end type group_pass_type
interface pass_var
  module procedure pass_var_3d, pass_var_2d
end interface
interface pass_vector
  module procedure pass_vector_3d, pass_vector_2d
end interface
interface create_group_pass
  module procedure create_var_group_pass_2d, create_var_group_pass_3d, create_vector_group_pass_2d, create_vector_group_pass_3d
end interface create_group_pass
contains
!> A copy of a domain with halos of at least min_halo (which returns the halos taken): one PE, the same topology
subroutine clone_MOM_domain(MD_in, MOM_dom, min_halo, halo_size, symmetric, domain_name, turns, refine, extra_halo)
  type(MOM_domain_type),           intent(in)    :: MD_in
  type(MOM_domain_type),           pointer       :: MOM_dom
  integer, dimension(2), optional, intent(inout) :: min_halo
  integer,               optional, intent(in)    :: halo_size, turns, refine, extra_halo
  logical,               optional, intent(in)    :: symmetric
  character(len=*),      optional, intent(in)    :: domain_name
  if (.not.associated(MOM_dom)) allocate(MOM_dom)
  MOM_dom = MD_in
  if (present(min_halo)) then
    MOM_dom%nihalo = max(MOM_dom%nihalo, min_halo(1)) ; min_halo(1) = MOM_dom%nihalo
    MOM_dom%njhalo = max(MOM_dom%njhalo, min_halo(2)) ; min_halo(2) = MOM_dom%njhalo
  endif
  if (present(symmetric)) MOM_dom%symmetric = symmetric
end subroutine clone_MOM_domain
subroutine deallocate_MOM_domain(MOM_domain, cursory)
  type(MOM_domain_type), pointer :: MOM_domain
  logical,     optional, intent(in) :: cursory
  if (associated(MOM_domain)) deallocate(MOM_domain)
end subroutine deallocate_MOM_domain
! The group passes of the stand-in remember nothing: the reference's callers are compiled against these interfaces, never run.
subroutine group_slot(group, k)
  type(group_pass_type), intent(inout) :: group
  integer,               intent(out)   :: k
  if (group%used) then      ! a later round of create_group_pass calls: the fields are replaced in order
    group%ridx = group%ridx + 1 ; if (group%ridx > group%n) group%ridx = 1
    k = group%ridx
  else
    group%n = group%n + 1 ; k = group%n
    if (k > size(group%e)) error stop "group_pass stand-in: too many fields in a group"
  endif
  group%e(k)%a2 => NULL() ; group%e(k)%a3 => NULL()
end subroutine group_slot
subroutine vector_positions(stagger, pu, pv)
  integer, optional, intent(in)  :: stagger
  integer,           intent(out) :: pu, pv
  pu = EAST_FACE ; pv = NORTH_FACE
  if (present(stagger)) then
    if (stagger == AGRID) then ; pu = CENTER ; pv = CENTER ; endif
    if (stagger == BGRID_NE) then ; pu = CORNER ; pv = CORNER ; endif
  endif
end subroutine vector_positions
subroutine create_var_group_pass_2d(group, array, MOM_dom, sideflag, position, halo, clock)
  type(group_pass_type),  intent(inout) :: group
  real, dimension(:,:), target, intent(inout) :: array
  type(MOM_domain_type),  intent(inout) :: MOM_dom
  integer,      optional, intent(in)    :: sideflag, position, halo, clock
  integer :: k
  call group_slot(group, k)
  group%e(k)%a2 => array ; group%e(k)%pos = CENTER ; if (present(position)) group%e(k)%pos = position
end subroutine create_var_group_pass_2d
subroutine create_var_group_pass_3d(group, array, MOM_dom, sideflag, position, halo, clock)
  type(group_pass_type),  intent(inout) :: group
  real, dimension(:,:,:), target, intent(inout) :: array
  type(MOM_domain_type),  intent(inout) :: MOM_dom
  integer,      optional, intent(in)    :: sideflag, position, halo, clock
  integer :: k
  call group_slot(group, k)
  group%e(k)%a3 => array ; group%e(k)%pos = CENTER ; if (present(position)) group%e(k)%pos = position
end subroutine create_var_group_pass_3d
subroutine create_vector_group_pass_2d(group, u_cmpt, v_cmpt, MOM_dom, direction, stagger, halo, clock)
  type(group_pass_type),  intent(inout) :: group
  real, dimension(:,:), target, intent(inout) :: u_cmpt, v_cmpt
  type(MOM_domain_type),  intent(inout) :: MOM_dom
  integer,      optional, intent(in)    :: direction, stagger, halo, clock
  integer :: k, pu, pv
  call vector_positions(stagger, pu, pv)
  call group_slot(group, k) ; group%e(k)%a2 => u_cmpt ; group%e(k)%pos = pu
  call group_slot(group, k) ; group%e(k)%a2 => v_cmpt ; group%e(k)%pos = pv
end subroutine create_vector_group_pass_2d
subroutine create_vector_group_pass_3d(group, u_cmpt, v_cmpt, MOM_dom, direction, stagger, halo, clock)
  type(group_pass_type),  intent(inout) :: group
  real, dimension(:,:,:), target, intent(inout) :: u_cmpt, v_cmpt
  type(MOM_domain_type),  intent(inout) :: MOM_dom
  integer,      optional, intent(in)    :: direction, stagger, halo, clock
  integer :: k, pu, pv
  call vector_positions(stagger, pu, pv)
  call group_slot(group, k) ; group%e(k)%a3 => u_cmpt ; group%e(k)%pos = pu
  call group_slot(group, k) ; group%e(k)%a3 => v_cmpt ; group%e(k)%pos = pv
end subroutine create_vector_group_pass_3d
subroutine do_group_pass(group, MOM_dom, clock)
  type(group_pass_type), intent(inout) :: group
  type(MOM_domain_type), intent(inout) :: MOM_dom
  integer,     optional, intent(in)    :: clock
  integer :: k
  group%used = .true. ; group%ridx = 0
  if (.not.(MOM_dom%reentrant(1) .or. MOM_dom%reentrant(2))) return
  do k = 1, group%n
    if (associated(group%e(k)%a2)) call pass_var_2d(group%e(k)%a2, MOM_dom, position=group%e(k)%pos)
    if (associated(group%e(k)%a3)) call pass_var_3d(group%e(k)%a3, MOM_dom, position=group%e(k)%pos)
  enddo
end subroutine do_group_pass
subroutine start_group_pass(group, MOM_dom, clock)
  type(group_pass_type), intent(inout) :: group
  type(MOM_domain_type), intent(inout) :: MOM_dom
  integer,     optional, intent(in)    :: clock
  call do_group_pass(group, MOM_dom)
end subroutine start_group_pass
subroutine complete_group_pass(group, MOM_dom, clock)
  type(group_pass_type), intent(inout) :: group
  type(MOM_domain_type), intent(inout) :: MOM_dom
  integer,     optional, intent(in)    :: clock
end subroutine complete_group_pass
subroutine pass_var_3d(array, MOM_dom, sideflag, complete, position, halo, clock)
  real, dimension(:,:,:), intent(inout) :: array
  type(MOM_domain_type),  intent(inout) :: MOM_dom
  integer, optional,      intent(in)    :: sideflag, position, halo, clock
  logical, optional,      intent(in)    :: complete
  integer :: k
  do k = 1, size(array, 3) ; call pass_var_2d(array(:,:,k), MOM_dom, position=position) ; enddo
end subroutine pass_var_3d
subroutine pass_var_2d(array, MOM_dom, sideflag, complete, position, halo, inner_halo, clock)
  real, dimension(:,:),  intent(inout) :: array
  type(MOM_domain_type), intent(inout) :: MOM_dom
  integer, optional,     intent(in)    :: sideflag, position, halo, inner_halo, clock
  logical, optional,     intent(in)    :: complete
  integer :: pos, hx, hy, ni, nj, xs, ys, i, j
  pos = CENTER ; if (present(position)) pos = position
  xs = 0 ; if (pos == EAST_FACE .or. pos == CORNER) xs = 1
  ys = 0 ; if (pos == NORTH_FACE .or. pos == CORNER) ys = 1
  hx = MOM_dom%nihalo ; hy = MOM_dom%njhalo ; ni = MOM_dom%niglobal ; nj = MOM_dom%njglobal
  ! local indices: cell i of the compute domain sits at hx + i (+ xs for the face east of it ...); symmetric memory
  if (MOM_dom%reentrant(1)) then
    do j = 1, size(array, 2)
      do i = 1, hx + xs ; array(i, j) = array(i + ni, j) ; enddo
      do i = hx + xs + ni + 1, size(array, 1) ; array(i, j) = array(i - ni, j) ; enddo
    enddo
  endif
  if (MOM_dom%reentrant(2)) then
    do j = 1, hy + ys ; array(:, j) = array(:, j + nj) ; enddo
    do j = hy + ys + nj + 1, size(array, 2) ; array(:, j) = array(:, j - nj) ; enddo
  endif
end subroutine pass_var_2d
subroutine pass_vector_3d(u_cmpt, v_cmpt, MOM_dom, direction, stagger, complete, halo, clock)
  real, dimension(:,:,:), intent(inout) :: u_cmpt, v_cmpt
  type(MOM_domain_type),  intent(inout) :: MOM_dom
  integer, optional,      intent(in)    :: direction, stagger, halo, clock
  logical, optional,      intent(in)    :: complete
  call pass_var_3d(u_cmpt, MOM_dom, position=EAST_FACE)
  call pass_var_3d(v_cmpt, MOM_dom, position=NORTH_FACE)
end subroutine pass_vector_3d
subroutine pass_vector_2d(u_cmpt, v_cmpt, MOM_dom, direction, stagger, complete, halo, clock)
  real, dimension(:,:),   intent(inout) :: u_cmpt, v_cmpt
  type(MOM_domain_type),  intent(inout) :: MOM_dom
  integer, optional,      intent(in)    :: direction, stagger, halo, clock
  logical, optional,      intent(in)    :: complete
  call pass_var_2d(u_cmpt, MOM_dom, position=EAST_FACE)
  call pass_var_2d(v_cmpt, MOM_dom, position=NORTH_FACE)
end subroutine pass_vector_2d
end module MOM_domains

module MOM_hor_index
implicit none ; private
public :: hor_index_type
type :: hor_index_type
  integer :: isc, iec, jsc, jec, isd, ied, jsd, jed, IscB, IecB, JscB, JecB, IsdB, IedB, JsdB, JedB
  integer :: isg = 0, ieg = 0, jsg = 0, jeg = 0, IsgB = 0, IegB = 0, JsgB = 0, JegB = 0
  integer :: idg_offset = 0, jdg_offset = 0, turns = 0
  logical :: symmetric = .true.
end type hor_index_type
end module MOM_hor_index

module MOM_unit_scaling
implicit none ; private
public :: unit_scale_type
type :: unit_scale_type
  real :: m_to_Z = 1.0, Z_to_m = 1.0, m_to_L = 1.0, L_to_m = 1.0, s_to_T = 1.0, T_to_s = 1.0, m_s_to_L_T = 1.0, L_T_to_m_s = 1.0, &
          R_to_kg_m3 = 1.0, kg_m3_to_R = 1.0, L_to_Z = 1.0, Z_to_L = 1.0, Pa_to_RL2_T2 = 1.0, L_T2_to_m_s2 = 1.0, &
          m_s_to_Z_T = 1.0, Z_T_to_m_s = 1.0, RL2_T2_to_Pa = 1.0, RZ_to_kg_m2 = 1.0, Z2_T_to_m2_s = 1.0, m2_s_to_Z2_T = 1.0, &
          L_T_to_Z_T = 1.0, RZ_T_to_kg_m2s = 1.0, RLZ_T2_to_Pa = 1.0, Pa_to_RLZ_T2 = 1.0, T_to_sec = 1.0, Q_to_J_kg = 1.0, &
          J_kg_to_Q = 1.0, C_to_degC = 1.0, degC_to_C = 1.0, S_to_ppt = 1.0, ppt_to_S = 1.0, RZ3_T3_to_W_m2 = 1.0, W_m2_to_RZ3_T3 = 1.0
end type unit_scale_type
end module MOM_unit_scaling

module MOM_grid
use MOM_domains, only : MOM_domain_type
use MOM_hor_index, only : hor_index_type
use MOM_unit_scaling, only : unit_scale_type
implicit none ; private
public :: ocean_grid_type, hor_index_type
type :: ocean_grid_type
  type(MOM_domain_type), pointer :: Domain => NULL()
  integer :: isc, iec, jsc, jec, isd, ied, jsd, jed, IscB, IecB, JscB, JecB, IsdB, IedB, JsdB, JedB, ke
  integer :: first_direction = 0
  integer :: idg_offset = 0, jdg_offset = 0
  logical :: symmetric = .true.
  logical :: nonblocking_updates = .false.
  type(hor_index_type) :: HI
  type(unit_scale_type), pointer :: US => NULL()
  real :: max_depth = 0.0, Z_ref = 0.0, Rad_Earth_L = 6.378e6
  real, allocatable, dimension(:,:) :: mask2dT, areaT, IareaT, dxT, dyT, IdxT, IdyT, bathyT
  real, allocatable, dimension(:,:) :: mask2dCu, dxCu, dyCu, dy_Cu, IdxCu, IdyCu, areaCu, IareaCu, OBCmaskCu, OBCmaskCv
  real, allocatable, dimension(:,:) :: mask2dCv, dxCv, dyCv, dx_Cv, IdxCv, IdyCv, areaCv, IareaCv
  real, allocatable, dimension(:,:) :: mask2dBu, dxBu, dyBu, areaBu, IareaBu, CoriolisBu, IdxBu, IdyBu
  real, allocatable, dimension(:,:) :: geoLonT, geoLatT, geoLonBu, geoLatBu
  real, allocatable, dimension(:,:) :: dF_dx, dF_dy      !< (the Leith viscosities only)
  integer :: isd_global = 0, jsd_global = 0
end type ocean_grid_type
end module MOM_grid

module MOM_verticalGrid
implicit none ; private
public :: verticalGrid_type, get_thickness_units, get_flux_units, get_tr_flux_units
type :: verticalGrid_type
  integer :: ke
  real :: Angstrom_m = 1.0e-10, Angstrom_Z = 1.0e-10, Angstrom_H = 1.0e-10, H_subroundoff = 1.0e-30, dZ_subroundoff = 1.0e-30, H_to_Z = 1.0, Z_to_H = 1.0, g_Earth = 9.8, &
          Rho0 = 1035.0, m_to_H = 1.0, H_to_m = 1.0, RZ_to_H = 1.0/1035.0, H_to_RZ = 1035.0, m2_s_to_HZ_T = 1.0, H_to_MKS = 1.0, &
          H_to_kg_m2 = 1035.0, kg_m2_to_H = 1.0/1035.0, HZ_T_to_m2_s = 1.0, HZ_T_to_MKS = 1.0, H_to_Pa = 9.8*1035.0
  integer :: nk_rho_varies = 0, nkml = 0
  logical :: Boussinesq = .true., semi_Boussinesq = .false.
  real, allocatable :: Rlay(:), g_prime(:)
  character(len=40) :: zAxisUnits = "", zAxisLongName = ""
  real, allocatable, dimension(:) :: sLayer, sInterface
  integer :: direction = 1
end type verticalGrid_type
contains
function get_thickness_units(GV)
  character(len=48)                   :: get_thickness_units
  type(verticalGrid_type), intent(in) :: GV
  get_thickness_units = "m"
end function get_thickness_units
function get_flux_units(GV)
  character(len=48)                   :: get_flux_units
  type(verticalGrid_type), intent(in) :: GV
  get_flux_units = "m3 s-1"
end function get_flux_units
function get_tr_flux_units(GV, tr_units, tr_vol_conc_units, tr_mass_conc_units)
  character(len=48)                      :: get_tr_flux_units
  type(verticalGrid_type),    intent(in) :: GV
  character(len=*), optional, intent(in) :: tr_units, tr_vol_conc_units, tr_mass_conc_units
  get_tr_flux_units = "m3 s-1"
end function get_tr_flux_units
end module MOM_verticalGrid


module MOM_time_manager
implicit none ; private
public :: time_type, time_type_to_real, real_to_time, set_date, operator(+), operator(-), operator(*), operator(/), operator(>), operator(<)
type :: time_type
  integer :: seconds = 0, days = 0
end type time_type
interface operator(+) ; module procedure time_plus ; end interface
interface operator(-) ; module procedure time_minus ; end interface
interface operator(*) ; module procedure time_scalar_mult, scalar_time_mult ; end interface
interface operator(/) ; module procedure time_scalar_divide ; end interface
interface operator(>) ; module procedure time_gt ; end interface
interface operator(<) ; module procedure time_lt ; end interface
contains
real function time_type_to_real(time)
  type(time_type), intent(in) :: time
  time_type_to_real = 86400.0*time%days + time%seconds
end function time_type_to_real
type(time_type) function real_to_time(x, err_msg)
  real, intent(in) :: x
  character(len=*), optional, intent(out) :: err_msg
  real_to_time%days = int(x / 86400.0) ; real_to_time%seconds = int(x - 86400.0*real_to_time%days)
end function real_to_time
type(time_type) function set_date(year, month, day, hour, minute, second, err_msg)      ! (tidal reference dates only: no calendar in the stand-in)
  integer,                    intent(in)  :: year, month, day
  integer,          optional, intent(in)  :: hour, minute, second
  character(len=*), optional, intent(out) :: err_msg
  set_date%days = 0 ; set_date%seconds = 0
  error stop "set_date stand-in: no calendar"
end function set_date
type(time_type) function time_plus(a, b)
  type(time_type), intent(in) :: a, b
  time_plus%days = a%days + b%days ; time_plus%seconds = a%seconds + b%seconds
end function time_plus
type(time_type) function time_minus(a, b)
  type(time_type), intent(in) :: a, b
  time_minus%days = a%days - b%days ; time_minus%seconds = a%seconds - b%seconds
end function time_minus
type(time_type) function time_scalar_mult(a, n)
  type(time_type), intent(in) :: a
  integer,         intent(in) :: n
  time_scalar_mult%days = a%days * n ; time_scalar_mult%seconds = a%seconds * n
end function time_scalar_mult
type(time_type) function scalar_time_mult(n, a)
  integer,         intent(in) :: n
  type(time_type), intent(in) :: a
  scalar_time_mult = time_scalar_mult(a, n)
end function scalar_time_mult
type(time_type) function time_scalar_divide(a, n)
  type(time_type), intent(in) :: a
  integer,         intent(in) :: n
  time_scalar_divide%days = a%days / n ; time_scalar_divide%seconds = a%seconds / n
end function time_scalar_divide
logical function time_gt(a, b)
  type(time_type), intent(in) :: a, b
  time_gt = (86400.0*a%days + a%seconds) > (86400.0*b%days + b%seconds)
end function time_gt
logical function time_lt(a, b)
  type(time_type), intent(in) :: a, b
  time_lt = (86400.0*a%days + a%seconds) < (86400.0*b%days + b%seconds)
end function time_lt
end module MOM_time_manager

module MOM_cpu_clock
implicit none ; private
public :: cpu_clock_id, cpu_clock_begin, cpu_clock_end, CLOCK_MODULE, CLOCK_ROUTINE
public :: CLOCK_COMPONENT, CLOCK_SUBCOMPONENT, CLOCK_MODULE_DRIVER, CLOCK_LOOP, CLOCK_INFRA
integer, parameter :: CLOCK_COMPONENT = 1, CLOCK_SUBCOMPONENT = 11, CLOCK_MODULE_DRIVER = 21, CLOCK_MODULE = 31, CLOCK_ROUTINE = 41, &
                      CLOCK_LOOP = 51, CLOCK_INFRA = 61
contains
integer function cpu_clock_id(name, sync, grain)
  character(len=*),  intent(in) :: name
  logical, optional, intent(in) :: sync
  integer, optional, intent(in) :: grain
  cpu_clock_id = 0
end function cpu_clock_id
subroutine cpu_clock_begin(id) ; integer, intent(in) :: id ; end subroutine cpu_clock_begin
subroutine cpu_clock_end(id) ; integer, intent(in) :: id ; end subroutine cpu_clock_end
end module MOM_cpu_clock

!> A parameter "file" held in memory: param_set(...) before the *_init calls, get_param with the reference's keywords
module MOM_file_parser
use MOM_error_handler, only : MOM_error, FATAL
implicit none ; private
public :: param_file_type, get_param, log_version, param_set, openParameterBlock, closeParameterBlock, log_param
character(len=64), save :: block_prefix = ''      !< "NAME%" inside openParameterBlock(NAME) ... closeParameterBlock
type :: param_file_type
  integer :: n = 0
  character(len=64)  :: names(256)
  character(len=128) :: values(256)
end type param_file_type
interface get_param
  module procedure get_param_logical, get_param_real, get_param_int, get_param_char, get_param_real_array, get_param_int_array
end interface
interface log_param
  module procedure log_param_logical, log_param_real, log_param_int, log_param_char, log_param_real_array
end interface
contains
subroutine log_param_real_array(CS, modulename, varname, value, desc, units, default, debuggingParam, like_default, unscale)
  type(param_file_type), intent(in) :: CS
  character(len=*),      intent(in) :: modulename, varname
  real, dimension(:),    intent(in) :: value
  character(len=*), optional, intent(in) :: desc
  character(len=*),           intent(in) :: units
  real,             optional, intent(in) :: default, unscale
  logical,          optional, intent(in) :: debuggingParam, like_default
end subroutine log_param_real_array
subroutine log_param_logical(CS, modulename, varname, value, desc, units, default, layoutParam, debuggingParam, like_default)
  type(param_file_type), intent(in) :: CS
  character(len=*),      intent(in) :: modulename, varname
  logical,               intent(in) :: value
  character(len=*), optional, intent(in) :: desc, units
  logical,          optional, intent(in) :: default, layoutParam, debuggingParam, like_default
end subroutine log_param_logical
subroutine log_param_real(CS, modulename, varname, value, desc, units, default, debuggingParam, like_default, unscale)
  type(param_file_type), intent(in) :: CS
  character(len=*),      intent(in) :: modulename, varname
  real,                  intent(in) :: value
  character(len=*), optional, intent(in) :: desc, units
  real,             optional, intent(in) :: default, unscale
  logical,          optional, intent(in) :: debuggingParam, like_default
end subroutine log_param_real
subroutine log_param_int(CS, modulename, varname, value, desc, units, default, layoutParam, debuggingParam, like_default)
  type(param_file_type), intent(in) :: CS
  character(len=*),      intent(in) :: modulename, varname
  integer,               intent(in) :: value
  character(len=*), optional, intent(in) :: desc, units
  integer,          optional, intent(in) :: default
  logical,          optional, intent(in) :: layoutParam, debuggingParam, like_default
end subroutine log_param_int
subroutine log_param_char(CS, modulename, varname, value, desc, units, default, layoutParam, debuggingParam, like_default)
  type(param_file_type), intent(in) :: CS
  character(len=*),      intent(in) :: modulename, varname, value
  character(len=*), optional, intent(in) :: desc, units, default
  logical,          optional, intent(in) :: layoutParam, debuggingParam, like_default
end subroutine log_param_char
subroutine param_set(CS, name, value)
  type(param_file_type), intent(inout) :: CS
  character(len=*),      intent(in)    :: name, value
  CS%n = CS%n + 1 ; CS%names(CS%n) = name ; CS%values(CS%n) = value
end subroutine param_set
function lookup(CS, name, found) result(val)
  type(param_file_type), intent(in)  :: CS
  character(len=*),      intent(in)  :: name
  logical,               intent(out) :: found
  character(len=128) :: val
  integer :: m
  found = .false. ; val = ''
  do m = 1, CS%n
    if (trim(CS%names(m)) == trim(block_prefix)//trim(name)) then ; val = CS%values(m) ; found = .true. ; endif
  enddo
end function lookup
subroutine openParameterBlock(CS, blockName, desc, do_not_log)
  type(param_file_type), intent(in) :: CS
  character(len=*),      intent(in) :: blockName
  character(len=*), optional, intent(in) :: desc
  logical,          optional, intent(in) :: do_not_log
  block_prefix = trim(block_prefix)//trim(blockName)//'%'
end subroutine openParameterBlock
subroutine closeParameterBlock(CS)
  type(param_file_type), intent(in) :: CS
  integer :: m
  m = index(block_prefix(1:max(len_trim(block_prefix)-1,0)), '%', back=.true.)
  block_prefix = block_prefix(1:m)
end subroutine closeParameterBlock
subroutine missing(varname, fail_if_missing)
  character(len=*),  intent(in) :: varname
  logical, optional, intent(in) :: fail_if_missing
  if (present(fail_if_missing)) then
    if (fail_if_missing) call MOM_error(FATAL, "get_param: "//trim(varname)//" is required but missing.")
  endif
end subroutine missing
subroutine log_version(CS, modulename, version, desc, log_to_all, all_default, layout, debugging)
  type(param_file_type), intent(in) :: CS
  character(len=*),      intent(in) :: modulename, version
  character(len=*), optional, intent(in) :: desc
  logical, optional,     intent(in) :: log_to_all, all_default, layout, debugging
end subroutine log_version
subroutine get_param_logical(CS, modulename, varname, value, desc, units, default, fail_if_missing, do_not_read, do_not_log, &
                             layoutParam, debuggingParam)
  type(param_file_type), intent(in)    :: CS
  character(len=*),      intent(in)    :: modulename, varname
  logical,               intent(inout) :: value
  character(len=*), optional, intent(in) :: desc, units
  logical, optional,     intent(in)    :: default, fail_if_missing, do_not_read, do_not_log, layoutParam, debuggingParam
  character(len=128) :: v ; logical :: found
  v = lookup(CS, varname, found)
  if (found) then
    value = (index(v, 'T') > 0 .or. index(v, 't') > 0)
  else
    if (present(default)) value = default
    call missing(varname, fail_if_missing)
  endif
end subroutine get_param_logical
subroutine get_param_real(CS, modulename, varname, value, desc, units, default, fail_if_missing, do_not_read, do_not_log, &
                          debuggingParam, scale, unscaled)
  type(param_file_type), intent(in)    :: CS
  character(len=*),      intent(in)    :: modulename, varname
  real,                  intent(inout) :: value
  character(len=*), optional, intent(in) :: desc, units
  real, optional,        intent(in)    :: default, scale
  logical, optional,     intent(in)    :: fail_if_missing, do_not_read, do_not_log, debuggingParam
  real, optional,        intent(out)   :: unscaled
  character(len=128) :: v ; logical :: found
  v = lookup(CS, varname, found)
  if (found) then
    read(v, *) value
  else
    if (present(default)) value = default
    call missing(varname, fail_if_missing)
  endif
  if (present(unscaled)) unscaled = value
  if (present(scale)) value = scale * value
end subroutine get_param_real
subroutine get_param_real_array(CS, modulename, varname, value, desc, units, default, fail_if_missing, do_not_read, do_not_log, &
                                debuggingParam, scale)
  type(param_file_type), intent(in)    :: CS
  character(len=*),      intent(in)    :: modulename, varname
  real, dimension(:),    intent(inout) :: value
  character(len=*), optional, intent(in) :: desc, units
  real, optional,        intent(in)    :: default, scale
  logical, optional,     intent(in)    :: fail_if_missing, do_not_read, do_not_log, debuggingParam
  character(len=128) :: v ; logical :: found
  v = lookup(CS, varname, found)
  if (found) then
    read(v, *) value
  else
    if (present(default)) value(:) = default
    call missing(varname, fail_if_missing)
  endif
  if (present(scale)) value(:) = scale * value(:)
end subroutine get_param_real_array
subroutine get_param_int_array(CS, modulename, varname, value, desc, units, default, fail_if_missing, do_not_read, do_not_log, &
                               layoutParam, debuggingParam)
  type(param_file_type), intent(in)    :: CS
  character(len=*),      intent(in)    :: modulename, varname
  integer, dimension(:), intent(inout) :: value
  character(len=*), optional, intent(in) :: desc, units
  integer, optional,     intent(in)    :: default
  logical, optional,     intent(in)    :: fail_if_missing, do_not_read, do_not_log, layoutParam, debuggingParam
  character(len=128) :: v ; logical :: found
  v = lookup(CS, varname, found)
  if (found) then
    read(v, *) value
  else
    if (present(default)) value(:) = default
    call missing(varname, fail_if_missing)
  endif
end subroutine get_param_int_array
subroutine get_param_int(CS, modulename, varname, value, desc, units, default, fail_if_missing, do_not_read, do_not_log, &
                         layoutParam, debuggingParam)
  type(param_file_type), intent(in)    :: CS
  character(len=*),      intent(in)    :: modulename, varname
  integer,               intent(inout) :: value
  character(len=*), optional, intent(in) :: desc, units
  integer, optional,     intent(in)    :: default
  logical, optional,     intent(in)    :: fail_if_missing, do_not_read, do_not_log, layoutParam, debuggingParam
  character(len=128) :: v ; logical :: found
  v = lookup(CS, varname, found)
  if (found) then
    read(v, *) value
  else
    if (present(default)) value = default
    call missing(varname, fail_if_missing)
  endif
end subroutine get_param_int
subroutine get_param_char(CS, modulename, varname, value, desc, units, default, fail_if_missing, do_not_read, do_not_log, &
                          layoutParam, debuggingParam)
  type(param_file_type), intent(in)    :: CS
  character(len=*),      intent(in)    :: modulename, varname
  character(len=*),      intent(inout) :: value
  character(len=*), optional, intent(in) :: desc, units, default
  logical, optional,     intent(in)    :: fail_if_missing, do_not_read, do_not_log, layoutParam, debuggingParam
  character(len=128) :: v ; logical :: found
  v = lookup(CS, varname, found)
  if (found) then
    value = trim(v)
  else
    if (present(default)) value = default
    call missing(varname, fail_if_missing)
  endif
end subroutine get_param_char
end module MOM_file_parser

module MOM_diag_mediator
use MOM_time_manager, only : time_type
use MOM_grid, only : ocean_grid_type
use MOM_verticalGrid, only : verticalGrid_type
use MOM_unit_scaling, only : unit_scale_type
implicit none ; private
public :: diag_ctrl, time_type, axes_grp
public :: diag_mediator_init, enable_averages, enable_averaging, disable_averaging, post_data, safe_alloc_ptr, safe_alloc_alloc
public :: post_product_u, post_product_sum_u, post_product_v, post_product_sum_v, query_averaging_enabled
public :: register_diag_field, register_static_field, set_diag_mediator_grid, diag_update_remap_grids
type :: axes_grp
  integer :: id = 0
end type axes_grp
type :: diag_ctrl
  integer :: unused = 0
  type(axes_grp) :: axesBL, axesTL, axesCuL, axesCvL, axesBi, axesTi, axesCui, axesCvi, axesB1, axesT1, axesCu1, axesCv1
  type(axes_grp) :: axesZi, axesZL, axesNull
end type diag_ctrl
interface post_data
  module procedure post_data_3d, post_data_2d, post_data_0d
end interface post_data
interface safe_alloc_ptr
  module procedure safe_alloc_ptr_3d_2arg, safe_alloc_ptr_3d_3arg, safe_alloc_ptr_3d_6arg, safe_alloc_ptr_2d_2arg, safe_alloc_ptr_2d
end interface safe_alloc_ptr
interface safe_alloc_alloc
  module procedure safe_alloc_allocatable_3d, safe_alloc_allocatable_2d
end interface safe_alloc_alloc
contains
! Nothing is written anywhere: these are the interfaces the reference's callers are compiled against.
subroutine post_data_0d(diag_field_id, field, diag_cs, is_static)
  integer,           intent(in) :: diag_field_id
  real,              intent(in) :: field
  type(diag_ctrl), target, intent(in) :: diag_CS
  logical, optional, intent(in) :: is_static
end subroutine post_data_0d
subroutine post_data_2d(diag_field_id, field, diag_cs, is_static, mask)
  integer,           intent(in) :: diag_field_id
  real,              intent(in) :: field(:,:)
  type(diag_ctrl), target, intent(in) :: diag_CS
  logical, optional, intent(in) :: is_static
  real,    optional, intent(in) :: mask(:,:)
end subroutine post_data_2d
subroutine post_data_3d(diag_field_id, field, diag_cs, is_static, mask, alt_h)
  integer,           intent(in) :: diag_field_id
  real,              intent(in) :: field(:,:,:)
  type(diag_ctrl), target, intent(in) :: diag_CS
  logical, optional, intent(in) :: is_static
  real,    optional, intent(in) :: mask(:,:,:)
  real, dimension(:,:,:), target, optional, intent(in) :: alt_h
end subroutine post_data_3d
subroutine post_product_u(id, u_a, u_b, G, nz, diag, mask, alt_h)
  integer,                  intent(in) :: id
  type(ocean_grid_type),    intent(in) :: G
  integer,                  intent(in) :: nz
  real, dimension(G%IsdB:G%IedB, G%jsd:G%jed, nz), intent(in) :: u_a, u_b
  type(diag_ctrl),          intent(in) :: diag
  real,           optional, intent(in) :: mask(:,:,:)
  real, target,   optional, intent(in) :: alt_h(:,:,:)
end subroutine post_product_u
subroutine post_product_sum_u(id, u_a, u_b, G, nz, diag)
  integer,                  intent(in) :: id
  type(ocean_grid_type),    intent(in) :: G
  integer,                  intent(in) :: nz
  real, dimension(G%IsdB:G%IedB, G%jsd:G%jed, nz), intent(in) :: u_a, u_b
  type(diag_ctrl),          intent(in) :: diag
end subroutine post_product_sum_u
subroutine post_product_v(id, v_a, v_b, G, nz, diag, mask, alt_h)
  integer,                  intent(in) :: id
  type(ocean_grid_type),    intent(in) :: G
  integer,                  intent(in) :: nz
  real, dimension(G%isd:G%ied, G%JsdB:G%JedB, nz), intent(in) :: v_a, v_b
  type(diag_ctrl),          intent(in) :: diag
  real,           optional, intent(in) :: mask(:,:,:)
  real, target,   optional, intent(in) :: alt_h(:,:,:)
end subroutine post_product_v
subroutine post_product_sum_v(id, v_a, v_b, G, nz, diag)
  integer,                  intent(in) :: id
  type(ocean_grid_type),    intent(in) :: G
  integer,                  intent(in) :: nz
  real, dimension(G%isd:G%ied, G%JsdB:G%JedB, nz), intent(in) :: v_a, v_b
  type(diag_ctrl),          intent(in) :: diag
end subroutine post_product_sum_v
subroutine enable_averaging(time_int_in, time_end_in, diag_cs)
  real,            intent(in)    :: time_int_in
  type(time_type), intent(in)    :: time_end_in
  type(diag_ctrl), intent(inout) :: diag_CS
end subroutine enable_averaging
subroutine enable_averages(time_int, time_end, diag_CS, T_to_s)
  real,            intent(in)    :: time_int
  type(time_type), intent(in)    :: time_end
  type(diag_ctrl), intent(inout) :: diag_CS
  real,  optional, intent(in)    :: T_to_s
end subroutine enable_averages
subroutine disable_averaging(diag_cs)
  type(diag_ctrl), intent(inout) :: diag_CS
end subroutine disable_averaging
logical function query_averaging_enabled(diag_cs, time_int, time_end)
  type(diag_ctrl),           intent(in)  :: diag_CS
  real,            optional, intent(out) :: time_int
  type(time_type), optional, intent(out) :: time_end
  query_averaging_enabled = .false.
  if (present(time_int)) time_int = 0.0
end function query_averaging_enabled
integer function register_diag_field(module_name, field_name, axes_in, init_time, &
            long_name, units, missing_value, range, mask_variant, standard_name,      &
            verbose, do_not_log, err_msg, interp_method, tile_count, cmor_field_name, &
            cmor_long_name, cmor_units, cmor_standard_name, cell_methods, &
            x_cell_method, y_cell_method, v_cell_method, conversion, v_extensive)
  character(len=*),           intent(in) :: module_name, field_name
  type(axes_grp),     target, intent(in) :: axes_in
  type(time_type),            intent(in) :: init_time
  character(len=*), optional, intent(in) :: long_name, units, standard_name
  real,             optional, intent(in) :: missing_value, range(2)
  logical,          optional, intent(in) :: mask_variant, verbose, do_not_log
  character(len=*), optional, intent(out):: err_msg
  character(len=*), optional, intent(in) :: interp_method
  integer,          optional, intent(in) :: tile_count
  character(len=*), optional, intent(in) :: cmor_field_name, cmor_long_name, cmor_units, cmor_standard_name, cell_methods
  character(len=*), optional, intent(in) :: x_cell_method, y_cell_method, v_cell_method
  real,             optional, intent(in) :: conversion
  logical,          optional, intent(in) :: v_extensive
  register_diag_field = -1
end function register_diag_field
integer function register_static_field(module_name, field_name, axes, &
            long_name, units, missing_value, range, mask_variant, standard_name, &
            do_not_log, interp_method, tile_count, &
            cmor_field_name, cmor_long_name, cmor_units, cmor_standard_name, area, &
            x_cell_method, y_cell_method, area_cell_method, conversion)
  character(len=*),           intent(in) :: module_name, field_name
  type(axes_grp),     target, intent(in) :: axes
  character(len=*), optional, intent(in) :: long_name, units, standard_name
  real,             optional, intent(in) :: missing_value, range(2)
  logical,          optional, intent(in) :: mask_variant, do_not_log
  character(len=*), optional, intent(in) :: interp_method
  integer,          optional, intent(in) :: tile_count, area
  character(len=*), optional, intent(in) :: cmor_field_name, cmor_long_name, cmor_units, cmor_standard_name
  character(len=*), optional, intent(in) :: x_cell_method, y_cell_method, area_cell_method
  real,             optional, intent(in) :: conversion
  register_static_field = -1
end function register_static_field
subroutine set_diag_mediator_grid(G, diag_cs)
  type(ocean_grid_type), intent(inout) :: G
  type(diag_ctrl),       intent(inout) :: diag_CS
end subroutine set_diag_mediator_grid
subroutine diag_update_remap_grids(diag_cs, alt_h, alt_T, alt_S, update_intensive, update_extensive)
  type(diag_ctrl),        intent(inout) :: diag_cs
  real, target, optional, intent(in   ) :: alt_h(:,:,:), alt_T(:,:,:), alt_S(:,:,:)
  logical, optional,      intent(in   ) :: update_intensive, update_extensive
end subroutine diag_update_remap_grids
subroutine diag_mediator_init(G, GV, US, nz, param_file, diag_cs, doc_file_dir)
  use MOM_file_parser, only : param_file_type
  type(ocean_grid_type), target, intent(inout) :: G
  type(verticalGrid_type), target, intent(in)  :: GV
  type(unit_scale_type),   target, intent(in)  :: US
  integer,                    intent(in)    :: nz
  type(param_file_type),      intent(in)    :: param_file
  type(diag_ctrl),            intent(inout) :: diag_cs
  character(len=*), optional, intent(in)    :: doc_file_dir
end subroutine diag_mediator_init
subroutine safe_alloc_ptr_3d_2arg(ptr, ni, nj, nk)
  real, dimension(:,:,:), pointer :: ptr
  integer, intent(in) :: ni, nj, nk
  if (.not.associated(ptr)) then ; allocate(ptr(ni,nj,nk), source=0.0) ; endif
end subroutine safe_alloc_ptr_3d_2arg
subroutine safe_alloc_ptr_3d_3arg(ptr, is, ie, js, je, nk)
  real, dimension(:,:,:), pointer :: ptr
  integer, intent(in) :: is, ie, js, je, nk
  if (.not.associated(ptr)) then ; allocate(ptr(is:ie,js:je,nk), source=0.0) ; endif
end subroutine safe_alloc_ptr_3d_3arg
subroutine safe_alloc_ptr_3d_6arg(ptr, is, ie, js, je, ks, ke)
  real, dimension(:,:,:), pointer :: ptr
  integer, intent(in) :: is, ie, js, je, ks, ke
  if (.not.associated(ptr)) then ; allocate(ptr(is:ie,js:je,ks:ke), source=0.0) ; endif
end subroutine safe_alloc_ptr_3d_6arg
subroutine safe_alloc_ptr_2d_2arg(ptr, ni, nj)
  real, dimension(:,:), pointer :: ptr
  integer, intent(in) :: ni, nj
  if (.not.associated(ptr)) then ; allocate(ptr(ni,nj), source=0.0) ; endif
end subroutine safe_alloc_ptr_2d_2arg
subroutine safe_alloc_ptr_2d(ptr, is, ie, js, je)
  real, dimension(:,:), pointer :: ptr
  integer, intent(in) :: is, ie, js, je
  if (.not.associated(ptr)) then ; allocate(ptr(is:ie,js:je), source=0.0) ; endif
end subroutine safe_alloc_ptr_2d
subroutine safe_alloc_allocatable_3d(ptr, is, ie, js, je, nk)
  real, dimension(:,:,:), allocatable :: ptr
  integer, intent(in) :: is, ie, js, je, nk
  if (.not.allocated(ptr)) then ; allocate(ptr(is:ie,js:je,nk), source=0.0) ; endif
end subroutine safe_alloc_allocatable_3d
subroutine safe_alloc_allocatable_2d(ptr, is, ie, js, je)
  real, dimension(:,:), allocatable :: ptr
  integer, intent(in) :: is, ie, js, je
  if (.not.allocated(ptr)) then ; allocate(ptr(is:ie,js:je), source=0.0) ; endif
end subroutine safe_alloc_allocatable_2d
end module MOM_diag_mediator

module MOM_safe_alloc      ! (the reference's MOM_diag_mediator re-exports MOM_safe_alloc's generics: here it is the other way round)
use MOM_diag_mediator, only : safe_alloc_ptr, safe_alloc_alloc
implicit none ; private
public :: safe_alloc_ptr, safe_alloc_alloc
end module MOM_safe_alloc

module MOM_io
use MOM_domains, only : CENTER, CORNER, EAST_FACE, NORTH_FACE
implicit none ; private
public :: vardesc, var_desc, query_vardesc, CENTER, CORNER, EAST_FACE, NORTH_FACE, stdout, stderr, MOM_read_data, slasher
public :: file_exists, field_exists, field_size, SINGLE_FILE, MULTIPLE, create_MOM_file, MOM_write_field, MOM_file, MOM_infra_file, MOM_netCDF_file, MOM_field
public :: verify_variable_units
interface MOM_read_data
  module procedure MOM_read_data_2d, MOM_read_data_1d
end interface MOM_read_data
integer, parameter :: stdout = 6, stderr = 0
integer, parameter :: SINGLE_FILE = 1, MULTIPLE = 2
!> the file handles and field descriptors the coordinate-file writers of MOM_regridding / MOM_hybgen_regrid declare (nothing is ever written)
type :: MOM_field
  character(len=64) :: label = ""
end type MOM_field
type :: MOM_file
  logical :: is_open = .false.
contains
  procedure :: close => close_MOM_file
end type MOM_file
type, extends(MOM_file) :: MOM_infra_file
end type MOM_infra_file
type, extends(MOM_file) :: MOM_netCDF_file
end type MOM_netCDF_file
type :: vardesc
  character(len=64)  :: name = ""
  character(len=48)  :: units = ""
  character(len=240) :: longname = ""
  character(len=8)   :: hor_grid = "h", z_grid = "L", t_grid = "s"
  real :: conversion = 1.0
  integer :: position = -1
end type vardesc
contains
subroutine close_MOM_file(handle)
  class(MOM_file), intent(inout) :: handle
  handle%is_open = .false.
end subroutine close_MOM_file
logical function file_exists(filename, MOM_Domain)
  use MOM_domains, only : MOM_domain_type
  character(len=*), intent(in) :: filename
  type(MOM_domain_type), optional, intent(in) :: MOM_Domain
  inquire(file=trim(filename), exist=file_exists)
end function file_exists
logical function field_exists(filename, field_name, MOM_domain)
  use MOM_domains, only : MOM_domain_type
  character(len=*), intent(in) :: filename, field_name
  type(MOM_domain_type), target, optional, intent(in) :: MOM_domain
  field_exists = .false.
end function field_exists
subroutine field_size(filename, fieldname, sizes, field_found, no_domain, ndims, ncid_in)
  character(len=*),      intent(in)    :: filename, fieldname
  integer, dimension(:), intent(inout) :: sizes
  logical,     optional, intent(out)   :: field_found
  logical,     optional, intent(in)    :: no_domain
  integer,     optional, intent(out)   :: ndims
  integer,     optional, intent(in)    :: ncid_in
  sizes(:) = 0
  if (present(field_found)) field_found = .false.
  if (present(ndims)) ndims = 0
end subroutine field_size
subroutine create_MOM_file(IO_handle, filename, vars, novars, fields, threading, timeunit, G, dG, GV, checksums, extra_axes, global_atts)
  class(MOM_file),       intent(inout) :: IO_handle
  character(len=*),      intent(in)    :: filename
  type(vardesc),         intent(in)    :: vars(:)
  integer,               intent(in)    :: novars
  type(MOM_field),       intent(inout) :: fields(:)
  integer,     optional, intent(in)    :: threading
  real,        optional, intent(in)    :: timeunit
  class(*),    optional, intent(in)    :: G, dG, GV, extra_axes(:), global_atts(:)
  integer(kind=8), optional, intent(in) :: checksums(:,:)
  IO_handle%is_open = .true.
end subroutine create_MOM_file
subroutine MOM_write_field(IO_handle, field_md, field, tstamp, scale)
  class(MOM_file),    intent(inout) :: IO_handle
  type(MOM_field),    intent(in)    :: field_md
  real, dimension(:), intent(in)    :: field
  real,     optional, intent(in)    :: tstamp, scale
end subroutine MOM_write_field
function var_desc(name, units, longname, hor_grid, z_grid, t_grid, cmor_field_name, &
                  cmor_units, cmor_longname, conversion, caller, position, dim_names, &
                  extra_axes, fixed) result(vd)
  character(len=*),           intent(in) :: name
  character(len=*), optional, intent(in) :: units, longname, hor_grid, z_grid, t_grid, cmor_field_name, cmor_units, cmor_longname
  real            , optional, intent(in) :: conversion
  character(len=*), optional, intent(in) :: caller
  integer,          optional, intent(in) :: position
  character(len=*), dimension(:), optional, intent(in) :: dim_names
  integer,          dimension(:), optional, intent(in) :: extra_axes
  logical,          optional, intent(in) :: fixed
  type(vardesc) :: vd
  vd%name = name
  if (present(units)) vd%units = units
  if (present(longname)) vd%longname = longname
  if (present(hor_grid)) vd%hor_grid = hor_grid
  if (present(z_grid)) vd%z_grid = z_grid
  if (present(conversion)) vd%conversion = conversion
end function var_desc
subroutine query_vardesc(vd, name, units, longname, hor_grid, z_grid, t_grid, cmor_field_name, cmor_units, cmor_longname, conversion, caller, &
                         position, dim_names)
  type(vardesc),              intent(in)  :: vd
  character(len=*), optional, intent(out) :: name, units, longname, hor_grid, z_grid, t_grid, cmor_field_name, cmor_units, cmor_longname
  real            , optional, intent(out) :: conversion
  character(len=*), optional, intent(in)  :: caller
  integer,          optional, intent(out) :: position
  character(len=*), dimension(:), optional, intent(out) :: dim_names
  if (present(name)) name = vd%name
  if (present(units)) units = vd%units
  if (present(longname)) longname = vd%longname
  if (present(hor_grid)) hor_grid = vd%hor_grid
  if (present(z_grid)) z_grid = vd%z_grid
  if (present(t_grid)) t_grid = vd%t_grid
  if (present(cmor_field_name)) cmor_field_name = ""
  if (present(cmor_units)) cmor_units = ""
  if (present(cmor_longname)) cmor_longname = ""
  if (present(conversion)) conversion = vd%conversion
  if (present(position)) position = vd%position
  if (present(dim_names)) dim_names(:) = ""
end subroutine query_vardesc
!> No files in the stand-in: a read is an error
subroutine MOM_read_data_2d(filename, fieldname, data, MOM_Domain, timelevel, position, scale, global_file, file_may_be_4d)
  use MOM_domains, only : MOM_domain_type
  use MOM_error_handler, only : MOM_error, FATAL
  character(len=*),       intent(in)    :: filename, fieldname
  real, dimension(:,:),   intent(inout) :: data
  type(MOM_domain_type),  intent(in)    :: MOM_Domain
  integer,      optional, intent(in)    :: timelevel, position
  real,         optional, intent(in)    :: scale
  logical,      optional, intent(in)    :: global_file, file_may_be_4d
  call MOM_error(FATAL, "MOM_read_data (stand-in): no files, asked for "//trim(fieldname)//" of "//trim(filename))
end subroutine MOM_read_data_2d
subroutine MOM_read_data_1d(filename, fieldname, data, timelevel, scale, MOM_Domain)
  use MOM_domains, only : MOM_domain_type
  use MOM_error_handler, only : MOM_error, FATAL
  character(len=*),       intent(in)    :: filename, fieldname
  real, dimension(:),     intent(inout) :: data
  integer,      optional, intent(in)    :: timelevel
  real,         optional, intent(in)    :: scale
  type(MOM_domain_type), optional, intent(in) :: MOM_Domain
  call MOM_error(FATAL, "MOM_read_data (stand-in): no files, asked for "//trim(fieldname)//" of "//trim(filename))
end subroutine MOM_read_data_1d
subroutine verify_variable_units(filename, varname, expected_units, msg, ierr, alt_units)
  character(len=*),           intent(in)    :: filename, varname, expected_units
  character(len=*),           intent(inout) :: msg
  logical,                    intent(out)   :: ierr
  character(len=*), optional, intent(in)    :: alt_units
  ierr = .true.
end subroutine verify_variable_units
function slasher(dir)
  character(len=*), intent(in) :: dir
  character(len=len(dir)+1) :: slasher
  slasher = trim(dir)
  if (len_trim(dir) > 0) then ; if (dir(len_trim(dir):len_trim(dir)) /= "/") slasher = trim(dir)//"/" ; endif
end function slasher
end module MOM_io

module MOM_get_input
implicit none ; private
public :: directories
type :: directories
  character(len=240) :: restart_input_dir = ".", restart_output_dir = ".", output_directory = ".", input_filename = "n"
end type directories
end module MOM_get_input

!> Restart registration that remembers nothing: every field reads as "not initialized" (a cold start)
module MOM_restart
use MOM_io, only : vardesc
use MOM_grid, only : ocean_grid_type
use MOM_verticalGrid, only : verticalGrid_type
use MOM_time_manager, only : time_type
use MOM_file_parser, only : param_file_type
implicit none ; private
public :: MOM_restart_CS, register_restart_field, query_initialized, register_restart_field_as_obsolete
public :: register_restart_pair, set_initialized, save_restart, only_read_from_restarts, restart_init, is_new_run
type :: MOM_restart_CS
  integer :: nfields = 0
end type MOM_restart_CS
interface register_restart_field
  module procedure register_3d, register_2d, register_0d, register_vd_3d, register_vd_2d
end interface
interface register_restart_pair
  module procedure register_pair_3d, register_pair_2d
end interface
interface query_initialized
  module procedure query_3d, query_2d, query_0d, query_name
end interface
interface set_initialized
  module procedure set_initialized_name, set_initialized_3d_name, set_initialized_2d_name, set_initialized_0d_name
end interface
interface only_read_from_restarts
  module procedure only_read_restart_field_3d, only_read_restart_field_2d, only_read_restart_pair_3d
end interface
contains
subroutine register_restart_field_as_obsolete(field_name, replacement_name, CS)
  character(*), intent(in) :: field_name, replacement_name
  type(MOM_restart_CS), intent(inout) :: CS
end subroutine register_restart_field_as_obsolete
subroutine register_vd_3d(f_ptr, var_desc, mandatory, CS, conversion)
  real, dimension(:,:,:), target, intent(in) :: f_ptr
  type(vardesc),        intent(in)    :: var_desc
  logical,              intent(in)    :: mandatory
  type(MOM_restart_CS), intent(inout) :: CS
  real,       optional, intent(in)    :: conversion
  CS%nfields = CS%nfields + 1
end subroutine register_vd_3d
subroutine register_vd_2d(f_ptr, var_desc, mandatory, CS, conversion)
  real, dimension(:,:), target, intent(in) :: f_ptr
  type(vardesc),        intent(in)    :: var_desc
  logical,              intent(in)    :: mandatory
  type(MOM_restart_CS), intent(inout) :: CS
  real,       optional, intent(in)    :: conversion
  CS%nfields = CS%nfields + 1
end subroutine register_vd_2d
subroutine register_pair_3d(a_ptr, b_ptr, a_desc, b_desc, mandatory, CS, conversion)
  real, dimension(:,:,:), target, intent(in) :: a_ptr, b_ptr
  type(vardesc),        intent(in)    :: a_desc, b_desc
  logical,              intent(in)    :: mandatory
  type(MOM_restart_CS), intent(inout) :: CS
  real,       optional, intent(in)    :: conversion
  CS%nfields = CS%nfields + 2
end subroutine register_pair_3d
subroutine register_pair_2d(a_ptr, b_ptr, a_desc, b_desc, mandatory, CS, conversion)
  real, dimension(:,:), target, intent(in) :: a_ptr, b_ptr
  type(vardesc),        intent(in)    :: a_desc, b_desc
  logical,              intent(in)    :: mandatory
  type(MOM_restart_CS), intent(inout) :: CS
  real,       optional, intent(in)    :: conversion
  CS%nfields = CS%nfields + 2
end subroutine register_pair_2d
logical function query_name(name, CS)
  character(len=*),     intent(in) :: name
  type(MOM_restart_CS), intent(in) :: CS
  query_name = .false.
end function query_name
subroutine set_initialized_name(name, CS)
  character(len=*),     intent(in)    :: name
  type(MOM_restart_CS), intent(inout) :: CS
end subroutine set_initialized_name
subroutine set_initialized_3d_name(f_ptr, name, CS)
  real, dimension(:,:,:), intent(in)  :: f_ptr
  character(len=*),     intent(in)    :: name
  type(MOM_restart_CS), intent(inout) :: CS
end subroutine set_initialized_3d_name
subroutine set_initialized_2d_name(f_ptr, name, CS)
  real, dimension(:,:), intent(in)    :: f_ptr
  character(len=*),     intent(in)    :: name
  type(MOM_restart_CS), intent(inout) :: CS
end subroutine set_initialized_2d_name
subroutine set_initialized_0d_name(f_ptr, name, CS)
  real,                 intent(in)    :: f_ptr
  character(len=*),     intent(in)    :: name
  type(MOM_restart_CS), intent(inout) :: CS
end subroutine set_initialized_0d_name
logical function is_new_run(CS)
  type(MOM_restart_CS), intent(in) :: CS
  is_new_run = .true.
end function is_new_run
subroutine save_restart(directory, time, G, CS, time_stamped, filename, GV, num_rest_files, write_IC)
  character(len=*),        intent(in)    :: directory
  type(time_type),         intent(in)    :: time
  type(ocean_grid_type),   intent(inout) :: G
  type(MOM_restart_CS),    intent(inout) :: CS
  logical,       optional, intent(in)    :: time_stamped
  character(len=*), optional, intent(in) :: filename
  type(verticalGrid_type), optional, intent(in) :: GV
  integer,       optional, intent(out)   :: num_rest_files
  logical,       optional, intent(in)    :: write_IC
end subroutine save_restart
subroutine restart_init(param_file, CS, restart_root)
  type(param_file_type), intent(in) :: param_file
  type(MOM_restart_CS),  pointer    :: CS
  character(len=*), optional, intent(in) :: restart_root
  if (.not.associated(CS)) allocate(CS)
end subroutine restart_init
subroutine only_read_restart_field_3d(varname, f_ptr, G, CS, position, filename, directory, success, scale)
  character(len=*),                intent(in)    :: varname
  real, dimension(:,:,:),          intent(inout) :: f_ptr
  type(ocean_grid_type),           intent(in)    :: G
  type(MOM_restart_CS),            intent(in)    :: CS
  integer,               optional, intent(in)    :: position
  character(len=*),      optional, intent(in)    :: filename, directory
  logical,               optional, intent(out)   :: success
  real,                  optional, intent(in)    :: scale
  if (present(success)) success = .false.
end subroutine only_read_restart_field_3d
subroutine only_read_restart_field_2d(varname, f_ptr, G, CS, position, filename, directory, success, scale)
  character(len=*),                intent(in)    :: varname
  real, dimension(:,:),            intent(inout) :: f_ptr
  type(ocean_grid_type),           intent(in)    :: G
  type(MOM_restart_CS),            intent(in)    :: CS
  integer,               optional, intent(in)    :: position
  character(len=*),      optional, intent(in)    :: filename, directory
  logical,               optional, intent(out)   :: success
  real,                  optional, intent(in)    :: scale
  if (present(success)) success = .false.
end subroutine only_read_restart_field_2d
subroutine only_read_restart_pair_3d(a_ptr, b_ptr, a_name, b_name, G, CS, stagger, filename, directory, success, scale)
  real, dimension(:,:,:),          intent(inout) :: a_ptr, b_ptr
  character(len=*),                intent(in)    :: a_name, b_name
  type(ocean_grid_type),           intent(in)    :: G
  type(MOM_restart_CS),            intent(in)    :: CS
  integer,               optional, intent(in)    :: stagger
  character(len=*),      optional, intent(in)    :: filename, directory
  logical,               optional, intent(out)   :: success
  real,                  optional, intent(in)    :: scale
  if (present(success)) success = .false.
end subroutine only_read_restart_pair_3d
subroutine register_3d(f_ptr, name, mandatory, CS, longname, units, conversion, hor_grid, z_grid, t_grid)
  real, dimension(:,:,:), target, intent(in) :: f_ptr
  character(len=*),     intent(in)    :: name
  logical,              intent(in)    :: mandatory
  type(MOM_restart_CS), intent(inout) :: CS
  character(len=*), optional, intent(in) :: longname, units, hor_grid, z_grid, t_grid
  real,             optional, intent(in) :: conversion
  CS%nfields = CS%nfields + 1
end subroutine register_3d
logical function query_3d(f_ptr, name, CS)
  real, dimension(:,:,:), intent(in) :: f_ptr
  character(len=*),     intent(in) :: name
  type(MOM_restart_CS), intent(in) :: CS
  query_3d = .false.
end function query_3d
subroutine register_2d(f_ptr, name, mandatory, CS, longname, units, conversion, hor_grid, z_grid, t_grid)
  real, dimension(:,:), target, intent(in) :: f_ptr
  character(len=*),     intent(in)    :: name
  logical,              intent(in)    :: mandatory
  type(MOM_restart_CS), intent(inout) :: CS
  character(len=*), optional, intent(in) :: longname, units, hor_grid, z_grid, t_grid
  real,             optional, intent(in) :: conversion
  CS%nfields = CS%nfields + 1
end subroutine register_2d
subroutine register_0d(f_ptr, name, mandatory, CS, longname, units, conversion, t_grid)
  real, target,         intent(in)    :: f_ptr
  character(len=*),     intent(in)    :: name
  logical,              intent(in)    :: mandatory
  type(MOM_restart_CS), intent(inout) :: CS
  character(len=*), optional, intent(in) :: longname, units, t_grid
  real,             optional, intent(in) :: conversion
  CS%nfields = CS%nfields + 1
end subroutine register_0d
logical function query_2d(f_ptr, name, CS)
  real, dimension(:,:), intent(in) :: f_ptr
  character(len=*),     intent(in) :: name
  type(MOM_restart_CS), intent(in) :: CS
  query_2d = .false.
end function query_2d
logical function query_0d(f_ptr, name, CS)
  real,                 intent(in) :: f_ptr
  character(len=*),     intent(in) :: name
  type(MOM_restart_CS), intent(in) :: CS
  query_0d = .false.
end function query_0d
end module MOM_restart

#ifdef REF_EOS
#include "mom6_stubs_eos.inc"
#else
module MOM_EOS
implicit none ; private
public :: EOS_type
type :: EOS_type
  integer :: form_of_EOS = 0
end type EOS_type
end module MOM_EOS
#endif

module MOM_variables
use MOM_domains, only : group_pass_type
use MOM_grid, only : ocean_grid_type
use MOM_EOS, only : EOS_type
implicit none ; private
public :: BT_cont_type, porous_barrier_type, accel_diag_ptrs, cont_diag_ptrs, thermo_var_ptrs, vertvisc_type, &
          ocean_internal_state, alloc_BT_cont_type, dealloc_BT_cont_type, ocean_grid_type
interface alloc_BT_cont_type
  module procedure alloc_BT_cont_type_ranges, alloc_BT_cont_type_grid
end interface
type :: BT_cont_type
  real, allocatable :: FA_u_EE(:,:), FA_u_E0(:,:), FA_u_W0(:,:), FA_u_WW(:,:), uBT_WW(:,:), uBT_EE(:,:)
  real, allocatable :: FA_v_NN(:,:), FA_v_N0(:,:), FA_v_S0(:,:), FA_v_SS(:,:), vBT_SS(:,:), vBT_NN(:,:)
  real, allocatable :: h_u(:,:,:), h_v(:,:,:)
  type(group_pass_type) :: pass_polarity_BT, pass_FA_uv
end type BT_cont_type
type :: porous_barrier_type
  real, allocatable :: por_face_areaU(:,:,:), por_face_areaV(:,:,:), por_layer_widthU(:,:,:), por_layer_widthV(:,:,:)
end type porous_barrier_type
type :: accel_diag_ptrs
  real, pointer, dimension(:,:,:) :: diffu => NULL(), diffv => NULL(), CAu => NULL(), CAv => NULL(), PFu => NULL(), PFv => NULL()
  real, pointer, dimension(:,:,:) :: du_dt_visc => NULL(), dv_dt_visc => NULL(), du_dt_visc_gl90 => NULL(), dv_dt_visc_gl90 => NULL()
  real, pointer, dimension(:,:,:) :: du_dt_str => NULL(), dv_dt_str => NULL(), du_dt_dia => NULL(), dv_dt_dia => NULL()
  real, pointer, dimension(:,:,:) :: u_accel_bt => NULL(), v_accel_bt => NULL(), du_other => NULL(), dv_other => NULL()
  real, pointer, dimension(:,:,:) :: gradKEu => NULL(), gradKEv => NULL(), rv_x_u => NULL(), rv_x_v => NULL()
  real, pointer, dimension(:,:,:) :: diag_hfrac_u => NULL(), diag_hfrac_v => NULL(), diag_hu => NULL(), diag_hv => NULL()
  real, pointer, dimension(:,:,:) :: visc_rem_u => NULL(), visc_rem_v => NULL()
end type accel_diag_ptrs
type :: cont_diag_ptrs
  real, pointer, dimension(:,:,:) :: uh => NULL(), vh => NULL(), uhGM => NULL(), vhGM => NULL()
end type cont_diag_ptrs
type :: thermo_var_ptrs
  real, pointer, dimension(:,:,:) :: T => NULL(), S => NULL()
  real, pointer, dimension(:,:) :: p_surf => NULL()
  type(EOS_type), pointer :: eqn_of_state => NULL()      !< associated = an equation of state is used (use_EOS)
  real :: P_Ref = 2.0e7
  real, allocatable, dimension(:,:,:) :: SpV_avg      !< (non-Boussinesq runs only)
  real, pointer, dimension(:,:,:) :: varT => NULL(), varS => NULL(), covarTS => NULL()      !< (the Stanley parameterisations only)
  integer :: valid_SpV_halo = -1
end type thermo_var_ptrs
type :: vertvisc_type
  real :: Prandtl_turb = 1.0
  real, allocatable, dimension(:,:) :: Kv_bbl_u, Kv_bbl_v, bbl_thick_u, bbl_thick_v
  real, allocatable, dimension(:,:) :: nkml_visc_u, nkml_visc_v
  real, allocatable, dimension(:,:,:) :: Ray_u, Ray_v
  real, pointer, dimension(:,:,:) :: Kv_shear => NULL(), Kv_shear_Bu => NULL()
  real, pointer, dimension(:,:) :: h_ML => NULL()
  ! members the reference's own MOM_vert_friction / MOM_set_viscosity name (ice shelves, the slow viscosity, the shear-mixing outputs); never
  ! allocated by the tests
  real, allocatable, dimension(:,:) :: taux_shelf, tauy_shelf, tbl_thick_shelf_u, tbl_thick_shelf_v, kv_tbl_shelf_u, kv_tbl_shelf_v
  real, allocatable, dimension(:,:) :: ustar_BBL, TKE_BBL
  real, pointer, dimension(:,:) :: sfc_buoy_flx => NULL(), MLD => NULL()
  real, pointer, dimension(:,:,:) :: Kv_slow => NULL(), Kd_shear => NULL(), TKE_turb => NULL()
end type vertvisc_type
type :: ocean_internal_state
  real, pointer, dimension(:,:,:) :: T => NULL(), S => NULL(), u => NULL(), v => NULL(), h => NULL(), uh => NULL(), vh => NULL()
  real, pointer, dimension(:,:,:) :: CAu => NULL(), CAv => NULL(), PFu => NULL(), PFv => NULL(), diffu => NULL(), diffv => NULL()
  real, pointer, dimension(:,:,:) :: pbce => NULL(), u_accel_bt => NULL(), v_accel_bt => NULL()
  real, pointer, dimension(:,:,:) :: u_av => NULL(), v_av => NULL(), u_prev => NULL(), v_prev => NULL()
end type ocean_internal_state
contains
subroutine alloc_BT_cont_type_ranges(BT_cont, isd, ied, jsd, jed, nz, alloc_faces)
  type(BT_cont_type), pointer :: BT_cont
  integer, intent(in) :: isd, ied, jsd, jed, nz
  logical, optional, intent(in) :: alloc_faces
  allocate(BT_cont)
  allocate(BT_cont%FA_u_WW(isd-1:ied,jsd:jed), source=0.0) ; allocate(BT_cont%FA_u_W0(isd-1:ied,jsd:jed), source=0.0)
  allocate(BT_cont%FA_u_E0(isd-1:ied,jsd:jed), source=0.0) ; allocate(BT_cont%FA_u_EE(isd-1:ied,jsd:jed), source=0.0)
  allocate(BT_cont%uBT_WW(isd-1:ied,jsd:jed), source=0.0)  ; allocate(BT_cont%uBT_EE(isd-1:ied,jsd:jed), source=0.0)
  allocate(BT_cont%FA_v_SS(isd:ied,jsd-1:jed), source=0.0) ; allocate(BT_cont%FA_v_S0(isd:ied,jsd-1:jed), source=0.0)
  allocate(BT_cont%FA_v_N0(isd:ied,jsd-1:jed), source=0.0) ; allocate(BT_cont%FA_v_NN(isd:ied,jsd-1:jed), source=0.0)
  allocate(BT_cont%vBT_SS(isd:ied,jsd-1:jed), source=0.0)  ; allocate(BT_cont%vBT_NN(isd:ied,jsd-1:jed), source=0.0)
  if (present(alloc_faces)) then ; if (alloc_faces) then
    allocate(BT_cont%h_u(isd-1:ied,jsd:jed,nz), source=0.0) ; allocate(BT_cont%h_v(isd:ied,jsd-1:jed,nz), source=0.0)
  endif ; endif
end subroutine alloc_BT_cont_type_ranges
!> The reference's form (MOM_variables.F90: alloc_BT_cont_type(BT_cont, G, GV, alloc_faces))
subroutine alloc_BT_cont_type_grid(BT_cont, G, GV, alloc_faces)
  use MOM_grid, only : ocean_grid_type
  use MOM_verticalGrid, only : verticalGrid_type
  type(BT_cont_type),      pointer    :: BT_cont
  type(ocean_grid_type),   intent(in) :: G
  type(verticalGrid_type), intent(in) :: GV
  logical,       optional, intent(in) :: alloc_faces
  if (associated(BT_cont)) return
  call alloc_BT_cont_type_ranges(BT_cont, G%isd, G%ied, G%jsd, G%jed, GV%ke, alloc_faces)
end subroutine alloc_BT_cont_type_grid
subroutine dealloc_BT_cont_type(BT_cont)
  type(BT_cont_type), pointer :: BT_cont
  if (associated(BT_cont)) deallocate(BT_cont)
end subroutine dealloc_BT_cont_type
end module MOM_variables

module MOM_forcing_type
use MOM_grid, only : ocean_grid_type
use MOM_verticalGrid, only : verticalGrid_type
use MOM_unit_scaling, only : unit_scale_type
use MOM_variables, only : thermo_var_ptrs
implicit none ; private
public :: mech_forcing, forcing, find_ustar
interface find_ustar
  module procedure find_ustar_fluxes, find_ustar_mech_forcing
end interface find_ustar
type :: forcing
  real, pointer, dimension(:,:) :: ustar => NULL(), tau_mag => NULL(), buoy => NULL()
end type forcing
type :: mech_forcing
  real, pointer, dimension(:,:) :: taux => NULL(), tauy => NULL()
  real, pointer, dimension(:,:) :: ustar => NULL()      !< the surface friction velocity [Z T-1]
  real, pointer, dimension(:,:) :: p_surf => NULL(), p_surf_full => NULL(), net_mass_src => NULL()
  real, pointer, dimension(:,:) :: rigidity_ice_u => NULL(), rigidity_ice_v => NULL(), frac_shelf_u => NULL(), frac_shelf_v => NULL()
  real, pointer, dimension(:,:) :: tau_mag => NULL(), omega_w2x => NULL()
end type mech_forcing
contains
!> find_ustar in Boussinesq mode with forces%ustar associated (MOM_forcing_type.F90:1236-1297: a copy over the halo asked for)
subroutine find_ustar_mech_forcing(forces, tv, U_star, G, GV, US, halo, H_T_units)
  type(ocean_grid_type),   intent(in)  :: G
  type(verticalGrid_type), intent(in)  :: GV
  type(unit_scale_type),   intent(in)  :: US
  type(mech_forcing),      intent(in)  :: forces
  type(thermo_var_ptrs),   intent(in)  :: tv
  real, dimension(G%isd:G%ied,G%jsd:G%jed), intent(out) :: U_star
  integer,       optional, intent(in)  :: halo
  logical,       optional, intent(in)  :: H_T_units
  integer :: i, j, hs
  real :: fac
  if (.not.associated(forces%ustar)) error stop "find_ustar stand-in: forces%ustar is needed"
  if (.not.GV%Boussinesq) error stop "find_ustar stand-in: Boussinesq mode only"
  hs = 0 ; if (present(halo)) hs = max(halo, 0)
  fac = 1.0 ; if (present(H_T_units)) then ; if (H_T_units) fac = GV%Z_to_H ; endif      ! (in thickness units over time, :1271)
  if (fac == 1.0) then
    do j=G%jsc-hs,G%jec+hs ; do i=G%isc-hs,G%iec+hs ; U_star(i,j) = forces%ustar(i,j) ; enddo ; enddo
  else
    do j=G%jsc-hs,G%jec+hs ; do i=G%isc-hs,G%iec+hs ; U_star(i,j) = fac * forces%ustar(i,j) ; enddo ; enddo
  endif
end subroutine find_ustar_mech_forcing
subroutine find_ustar_fluxes(fluxes, tv, U_star, G, GV, US, halo, H_T_units)
  type(ocean_grid_type),   intent(in)  :: G
  type(verticalGrid_type), intent(in)  :: GV
  type(unit_scale_type),   intent(in)  :: US
  type(forcing),           intent(in)  :: fluxes
  type(thermo_var_ptrs),   intent(in)  :: tv
  real, dimension(G%isd:G%ied,G%jsd:G%jed), intent(out) :: U_star
  integer,       optional, intent(in)  :: halo
  logical,       optional, intent(in)  :: H_T_units
  error stop "find_ustar(fluxes) stand-in: not provided"
end subroutine find_ustar_fluxes
end module MOM_forcing_type

#ifndef REF_ALE
module MOM_regridding
implicit none ; private
public :: regridding_CS
type :: regridding_CS
  integer :: nk = 0
end type regridding_CS
end module MOM_regridding

module MOM_remapping
implicit none ; private
public :: remapping_CS
type :: remapping_CS
  integer :: remapping_scheme = 0
end type remapping_CS
end module MOM_remapping
#endif

#ifdef REF_PF
! The reference's PLM_functions.F90, compiled where it lies (cpp #include by -I/root/reference/src/ALE), for the stand-in below.
#include "PLM_functions.F90"
#endif
#if !defined(MOM6HIP_WITH_ALE_SHIM) && !defined(REF_ALE)
module MOM_ALE
use MOM_grid, only : ocean_grid_type
use MOM_verticalGrid, only : verticalGrid_type
#ifdef REF_PF
use MOM_variables, only : thermo_var_ptrs
use PLM_functions, only : PLM_slope_wa, PLM_monotonized_slope, PLM_extrapolate_slope
#endif
implicit none ; private
public :: ALE_CS
type :: ALE_CS
  integer :: unused = 0
end type ALE_CS
public :: ALE_remap_velocities, ALE_remap_interface_vals, ALE_remap_vertex_vals
#ifdef REF_PF
! What MOM_PressureForce_FV imports from MOM_ALE.  MOM_ALE.F90 itself stands on the regridding and tracer-registry modules and is not part of
! this build: this stand-in forms the edge values of T and S of a column from the reference's OWN slope functions (PLM_functions.F90 above), the
! way ALE_PLM_edge_values does for REMAPPING_ANSWER_DATE >= 20190101 (MOM_ALE.F90:1520-1577).  The glue is ours; the arithmetic is theirs.
public :: TS_PLM_edge_values, TS_PPM_edge_values
#endif
contains
! What MOM_set_viscosity and MOM_dynamics_split_RK2 import from MOM_ALE for remap_vertvisc_aux_vars / remap_dyn_split_RK2_aux_vars
! (REMAP_AUXILIARY_VARS: reached only in builds with the MOM_ALE shim or the reference's own MOM_ALE, which take this module's place)
subroutine ALE_remap_velocities(CS, G, GV, h_old_u, h_old_v, h_new_u, h_new_v, u, v, debug, dt, allow_preserve_variance)
  type(ALE_CS),            intent(in)    :: CS
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  real, dimension(G%IsdB:G%IedB,G%jsd:G%jed,GV%ke), intent(in)    :: h_old_u, h_new_u
  real, dimension(G%isd:G%ied,G%JsdB:G%JedB,GV%ke), intent(in)    :: h_old_v, h_new_v
  real, dimension(G%IsdB:G%IedB,G%jsd:G%jed,GV%ke), intent(inout) :: u
  real, dimension(G%isd:G%ied,G%JsdB:G%JedB,GV%ke), intent(inout) :: v
  logical, optional, intent(in) :: debug, allow_preserve_variance
  real,    optional, intent(in) :: dt
  error stop "ALE_remap_velocities stand-in: not provided"
end subroutine ALE_remap_velocities
subroutine ALE_remap_interface_vals(CS, G, GV, h_old, h_new, int_val)
  type(ALE_CS),            intent(in)    :: CS
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke),   intent(in)    :: h_old, h_new
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke+1), intent(inout) :: int_val
  error stop "ALE_remap_interface_vals stand-in: not provided"
end subroutine ALE_remap_interface_vals
subroutine ALE_remap_vertex_vals(CS, G, GV, h_old, h_new, vert_val)
  type(ALE_CS),            intent(in)    :: CS
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke),       intent(in)    :: h_old, h_new
  real, dimension(G%IsdB:G%IedB,G%JsdB:G%JedB,GV%ke+1), intent(inout) :: vert_val
  error stop "ALE_remap_vertex_vals stand-in: not provided"
end subroutine ALE_remap_vertex_vals
#ifdef REF_PF
subroutine plm_edges(G, GV, h, Q, bdry_extrap, Q_t, Q_b)
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke), intent(in)    :: h, Q
  logical,                 intent(in)    :: bdry_extrap
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke), intent(inout) :: Q_t, Q_b
  real :: s(GV%ke), ms, hn
  integer :: i, j, k, nz
  nz = GV%ke ; hn = GV%H_subroundoff
  do j=G%jsc-1,G%jec+1 ; do i=G%isc-1,G%iec+1
    s(1) = 0. ; s(nz) = 0.
    do k=2,nz-1 ; s(k) = PLM_slope_wa(h(i,j,k-1), h(i,j,k), h(i,j,k+1), hn, Q(i,j,k-1), Q(i,j,k), Q(i,j,k+1)) ; enddo
    do k=2,nz-1
      ms = PLM_monotonized_slope(Q(i,j,k-1), Q(i,j,k), Q(i,j,k+1), s(k-1), s(k), s(k+1))
      Q_t(i,j,k) = Q(i,j,k) - 0.5 * ms ; Q_b(i,j,k) = Q(i,j,k) + 0.5 * ms
    enddo
    if (bdry_extrap) then
      ms = - PLM_extrapolate_slope(h(i,j,2), h(i,j,1), hn, Q(i,j,2), Q(i,j,1))
      Q_t(i,j,1) = Q(i,j,1) - 0.5 * ms ; Q_b(i,j,1) = Q(i,j,1) + 0.5 * ms
      ms = PLM_extrapolate_slope(h(i,j,nz-1), h(i,j,nz), hn, Q(i,j,nz-1), Q(i,j,nz))
      Q_t(i,j,nz) = Q(i,j,nz) - 0.5 * ms ; Q_b(i,j,nz) = Q(i,j,nz) + 0.5 * ms
    else
      Q_t(i,j,1) = Q(i,j,1) ; Q_b(i,j,1) = Q(i,j,1) ; Q_t(i,j,nz) = Q(i,j,nz) ; Q_b(i,j,nz) = Q(i,j,nz)
    endif
  enddo ; enddo
end subroutine plm_edges
subroutine TS_PLM_edge_values(CS, S_t, S_b, T_t, T_b, G, GV, tv, h, bdry_extrap)
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(ALE_CS),            intent(inout) :: CS
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke), intent(inout) :: S_t, S_b, T_t, T_b
  type(thermo_var_ptrs),   intent(in)    :: tv
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke), intent(in)    :: h
  logical,                 intent(in)    :: bdry_extrap
  call plm_edges(G, GV, h, tv%S, bdry_extrap, S_t, S_b)
  call plm_edges(G, GV, h, tv%T, bdry_extrap, T_t, T_b)
end subroutine TS_PLM_edge_values
subroutine TS_PPM_edge_values(CS, S_t, S_b, T_t, T_b, G, GV, tv, h, bdry_extrap)
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(ALE_CS),            intent(inout) :: CS
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke), intent(inout) :: S_t, S_b, T_t, T_b
  type(thermo_var_ptrs),   intent(in)    :: tv
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke), intent(in)    :: h
  logical,                 intent(in)    :: bdry_extrap
  error stop "TS_PPM_edge_values stand-in: PRESSURE_RECONSTRUCTION_SCHEME = 2 is not provided"
end subroutine TS_PPM_edge_values
#endif
end module MOM_ALE
#endif

module MOM_barotropic_types_for_hor_visc
end module MOM_barotropic_types_for_hor_visc

module MOM_stochastics
implicit none ; private
public :: stochastic_CS
type :: stochastic_CS
  logical :: do_sppt = .false., do_skeb = .false., skeb_use_gm = .false., skeb_use_frict = .false., pert_epbl = .false.
  real, allocatable, dimension(:,:,:) :: skeb_diss
  real :: skeb_frict_coef = 0.0, skeb_gm_coef = 0.0
end type stochastic_CS
end module MOM_stochastics

module MOM_MEKE_types
implicit none ; private
public :: MEKE_type
type :: MEKE_type
  real, allocatable :: Kh(:,:), Ku(:,:), Au(:,:), mom_src(:,:), GME_snk(:,:), GM_src(:,:), MEKE(:,:), Rd_dx_h(:,:), Kh_diff(:,:)
  real :: KhTr_fac = 1.0, KhTh_fac = 1.0, backscatter_Ro_c = 0.0, backscatter_Ro_pow = 0.0
end type MEKE_type
end module MOM_MEKE_types


module MOM_tracer_registry
use MOM_grid, only : ocean_grid_type
use MOM_verticalGrid, only : verticalGrid_type
implicit none ; private
public :: tracer_registry_type, tracer_type, MOM_tracer_chkinv, MOM_tracer_chksum, tracer_name_lookup
interface MOM_tracer_chksum
  module procedure tracer_array_chksum, tracer_Reg_chksum
end interface MOM_tracer_chksum
interface MOM_tracer_chkinv
  module procedure tracer_array_chkinv, tracer_Reg_chkinv
end interface MOM_tracer_chkinv
type :: tracer_type
  real, dimension(:,:,:), pointer :: t => NULL()
  real :: conc_underflow = 0.0
  real, dimension(:,:,:), pointer :: ad_x => NULL(), ad_y => NULL(), advection_xy => NULL()
  real, dimension(:,:),   pointer :: ad2d_x => NULL(), ad2d_y => NULL()
  real, dimension(:,:,:), pointer :: df_x => NULL(), df_y => NULL()
  real, dimension(:,:),   pointer :: df2d_x => NULL(), df2d_y => NULL()
  character(len=32) :: name = ""
  integer :: id_remap_conc = -1, id_remap_cont = -1, id_remap_cont_2d = -1      !< (diagnostics of ALE_remap_tracers: never registered)
  integer :: id_dfxy_cont = -1, id_dfxy_cont_2d = -1, id_dfxy_conc = -1, id_dfx_2d = -1, id_dfy_2d = -1      !< (of the neutral diffusion)
  integer :: id_hbdxy_cont = -1, id_hbdxy_cont_2d = -1, id_hbdxy_conc = -1, id_hbd_dfx = -1, id_hbd_dfy = -1, id_hbd_dfx_2d = -1, id_hbd_dfy_2d = -1
  real :: conc_scale = 1.0
end type tracer_type
type :: tracer_registry_type
  integer :: ntr = 0
  type(tracer_type) :: Tr(16)
end type tracer_registry_type
contains
subroutine tracer_name_lookup(Reg, n, tr_ptr, name)      ! MOM_tracer_registry.F90:912 (names compared as they are: the stand-in has no lowercase here)
  type(tracer_registry_type), pointer    :: Reg
  type(tracer_type), pointer             :: tr_ptr
  character(len=32), intent(in)          :: name
  integer, intent(out)                   :: n
  do n=1,Reg%ntr
    if (trim(Reg%Tr(n)%name) == trim(name)) then
      tr_ptr => Reg%Tr(n)
      return
    endif
  enddo
  error stop "MOM cannot find registered tracer"
end subroutine tracer_name_lookup
!> the debugging inventories of MOM_tracer_registry.F90 (printed with DEBUG only): nothing is printed here
subroutine tracer_array_chksum(mesg, Tr, ntr, G)      ! (debugging checksums: nothing is printed)
  character(len=*),      intent(in) :: mesg
  type(tracer_type),     intent(in) :: Tr(:)
  integer,               intent(in) :: ntr
  type(ocean_grid_type), intent(in) :: G
end subroutine tracer_array_chksum
subroutine tracer_Reg_chksum(mesg, Reg, G)
  character(len=*),           intent(in) :: mesg
  type(tracer_registry_type), pointer    :: Reg
  type(ocean_grid_type),      intent(in) :: G
end subroutine tracer_Reg_chksum
subroutine tracer_array_chkinv(mesg, G, GV, h, Tr, ntr)
  character(len=*),         intent(in) :: mesg
  type(ocean_grid_type),    intent(in) :: G
  type(verticalGrid_type),  intent(in) :: GV
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke), intent(in) :: h
  type(tracer_type), dimension(:), intent(in) :: Tr
  integer,                  intent(in) :: ntr
end subroutine tracer_array_chkinv
subroutine tracer_Reg_chkinv(mesg, G, GV, h, Reg)
  character(len=*),           intent(in) :: mesg
  type(ocean_grid_type),      intent(in) :: G
  type(verticalGrid_type),    intent(in) :: GV
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke), intent(in) :: h
  type(tracer_registry_type), pointer    :: Reg
end subroutine tracer_Reg_chkinv
end module MOM_tracer_registry

module MOM_self_attr_load
use MOM_grid, only : ocean_grid_type
use MOM_unit_scaling, only : unit_scale_type
use MOM_file_parser, only : param_file_type
implicit none ; private
public :: SAL_CS, SAL_init, SAL_end, scalar_SAL_sensitivity, calc_SAL
type :: SAL_CS
  integer :: unused = 0
end type SAL_CS
contains
subroutine calc_SAL(eta, eta_sal, G, CS, tmp_scale)
  type(ocean_grid_type), intent(in)  :: G
  real, dimension(G%isd:G%ied,G%jsd:G%jed), intent(in)  :: eta
  real, dimension(G%isd:G%ied,G%jsd:G%jed), intent(out) :: eta_sal
  type(SAL_CS), intent(inout) :: CS
  real, optional, intent(in) :: tmp_scale
  eta_sal(:,:) = 0.0
  error stop "calc_SAL stand-in: self-attraction and loading is not provided"
end subroutine calc_SAL
subroutine scalar_SAL_sensitivity(CS, deta_geo_dpbot)
  type(SAL_CS), intent(in)  :: CS
  real,         intent(out) :: deta_geo_dpbot
  deta_geo_dpbot = 0.0
end subroutine scalar_SAL_sensitivity
subroutine SAL_init(G, US, param_file, CS)
  type(ocean_grid_type),  intent(inout) :: G
  type(unit_scale_type),  intent(in)    :: US
  type(param_file_type),  intent(in)    :: param_file
  type(SAL_CS), intent(inout) :: CS
end subroutine SAL_init
subroutine SAL_end(CS)
  type(SAL_CS), intent(inout) :: CS
end subroutine SAL_end
end module MOM_self_attr_load

module MOM_tidal_forcing
use MOM_grid, only : ocean_grid_type
use MOM_unit_scaling, only : unit_scale_type
use MOM_file_parser, only : param_file_type
use MOM_time_manager, only : time_type
implicit none ; private
public :: tidal_forcing_CS, tidal_forcing_init, tidal_forcing_end, calc_tidal_forcing, calc_tidal_forcing_legacy
public :: astro_longitudes, astro_longitudes_init, eq_phase, nodal_fu, tidal_frequency
type :: astro_longitudes
  real :: s = 0.0, h = 0.0, p = 0.0, N = 0.0
end type astro_longitudes
type :: tidal_forcing_CS
  integer :: unused = 0
end type tidal_forcing_CS
contains
subroutine astro_longitudes_init(time_ref, longitudes)      ! (tidal segment data: not provided)
  type(time_type), intent(in) :: time_ref
  type(astro_longitudes), intent(out) :: longitudes
  error stop "astro_longitudes_init stand-in: tides are not provided"
end subroutine astro_longitudes_init
function eq_phase(constit, longitudes)
  character (len=2), intent(in) :: constit
  type(astro_longitudes), intent(in) :: longitudes
  real :: eq_phase
  eq_phase = 0.0
  error stop "eq_phase stand-in: tides are not provided"
end function eq_phase
function tidal_frequency(constit)
  character (len=2), intent(in) :: constit
  real :: tidal_frequency
  tidal_frequency = 0.0
  error stop "tidal_frequency stand-in: tides are not provided"
end function tidal_frequency
subroutine nodal_fu(constit, nodelon, fn, un)
  character (len=2), intent(in)  :: constit
  real,              intent(in)  :: nodelon
  real,              intent(out) :: fn, un
  fn = 1.0 ; un = 0.0
  error stop "nodal_fu stand-in: tides are not provided"
end subroutine nodal_fu
subroutine calc_tidal_forcing(Time, e_tide_eq, e_tide_sal, G, US, CS)
  type(ocean_grid_type),            intent(in)  :: G
  type(time_type),                  intent(in)  :: Time
  real, dimension(G%isd:G%ied,G%jsd:G%jed), intent(out) :: e_tide_eq, e_tide_sal
  type(unit_scale_type),            intent(in)  :: US
  type(tidal_forcing_CS),           intent(in)  :: CS
  e_tide_eq(:,:) = 0.0 ; e_tide_sal(:,:) = 0.0
  error stop "calc_tidal_forcing stand-in: tides are not provided"
end subroutine calc_tidal_forcing
subroutine calc_tidal_forcing_legacy(Time, e_sal, e_sal_tide, e_tide_eq, e_tide_sal, G, US, CS)
  type(ocean_grid_type),            intent(in)  :: G
  type(time_type),                  intent(in)  :: Time
  real, dimension(G%isd:G%ied,G%jsd:G%jed), intent(in)  :: e_sal
  real, dimension(G%isd:G%ied,G%jsd:G%jed), intent(out) :: e_sal_tide, e_tide_eq, e_tide_sal
  type(unit_scale_type),            intent(in)  :: US
  type(tidal_forcing_CS),           intent(in)  :: CS
  e_sal_tide(:,:) = 0.0 ; e_tide_eq(:,:) = 0.0 ; e_tide_sal(:,:) = 0.0
  error stop "calc_tidal_forcing_legacy stand-in: tides are not provided"
end subroutine calc_tidal_forcing_legacy
subroutine tidal_forcing_init(Time, G, US, param_file, CS)
  type(time_type),        intent(in)    :: Time
  type(ocean_grid_type),  intent(inout) :: G
  type(unit_scale_type),  intent(in)    :: US
  type(param_file_type),  intent(in)    :: param_file
  type(tidal_forcing_CS), intent(inout) :: CS
end subroutine tidal_forcing_init
subroutine tidal_forcing_end(CS)
  type(tidal_forcing_CS), intent(inout) :: CS
end subroutine tidal_forcing_end
end module MOM_tidal_forcing

module MOM_CVMix_KPP
use MOM_grid, only : ocean_grid_type
use MOM_unit_scaling, only : unit_scale_type
implicit none ; private
public :: KPP_CS, KPP_get_BLD
type :: KPP_CS
  integer :: unused = 0
end type KPP_CS
contains
subroutine KPP_get_BLD(CS, BLD, G, US, m_to_BLD_units)
  type(KPP_CS),                     pointer     :: CS
  type(ocean_grid_type),            intent(in)  :: G
  type(unit_scale_type),            intent(in)  :: US
  real, dimension(G%isd:G%ied,G%jsd:G%jed), intent(inout) :: BLD
  real,                   optional, intent(in)  :: m_to_BLD_units
end subroutine KPP_get_BLD
end module MOM_CVMix_KPP

module MOM_energetic_PBL
use MOM_grid, only : ocean_grid_type
use MOM_unit_scaling, only : unit_scale_type
implicit none ; private
public :: energetic_PBL_CS, energetic_PBL_get_MLD
type :: energetic_PBL_CS
  integer :: unused = 0
end type energetic_PBL_CS
contains
subroutine energetic_PBL_get_MLD(CS, MLD, G, US, m_to_MLD_units)
  type(energetic_PBL_CS),           intent(in)  :: CS
  type(ocean_grid_type),            intent(in)  :: G
  type(unit_scale_type),            intent(in)  :: US
  real, dimension(G%isd:G%ied,G%jsd:G%jed), intent(out) :: MLD
  real,                   optional, intent(in)  :: m_to_MLD_units
  MLD(:,:) = 0.0
end subroutine energetic_PBL_get_MLD
end module MOM_energetic_PBL

module MOM_diabatic_driver
use MOM_CVMix_KPP, only : KPP_CS
use MOM_energetic_PBL, only : energetic_PBL_CS
implicit none ; private
public :: diabatic_CS, extract_diabatic_member
type :: diabatic_CS
  integer :: unused = 0
  type(energetic_PBL_CS), pointer :: ePBL => NULL()      ! (the reference's diabatic_CS%ePBL: associated when ePBL is the boundary layer scheme)
end type diabatic_CS
contains
subroutine extract_diabatic_member(CS, evap_CFL_limit, minimum_forcing_depth, KPP_CSp, energetic_PBL_CSp, diabatic_halo, use_KPP)
  type(diabatic_CS), target, intent(in)      :: CS
  type(KPP_CS),      optional, pointer       :: KPP_CSp
  type(energetic_PBL_CS), optional, pointer  :: energetic_PBL_CSp
  real,              optional, intent(  out) :: evap_CFL_limit, minimum_forcing_depth
  integer,           optional, intent(  out) :: diabatic_halo
  logical,           optional, intent(  out) :: use_KPP
  if (present(energetic_PBL_CSp)) energetic_PBL_CSp => CS%ePBL
end subroutine extract_diabatic_member
end module MOM_diabatic_driver

#ifndef REF_OBC
module MOM_open_boundary
use MOM_hor_index, only : hor_index_type
use MOM_grid, only : ocean_grid_type
use MOM_verticalGrid, only : verticalGrid_type
use MOM_unit_scaling, only : unit_scale_type
use MOM_time_manager, only : time_type
implicit none ; private
public :: ocean_OBC_type, radiation_open_bdry_conds, open_boundary_zero_normal_flow, open_boundary_query
public :: open_boundary_test_extern_h, update_OBC_ramp
public :: OBC_segment_tracer_type, segment_tracer_registry_type, OBC_segment_data_type
public :: OBC_segment_type, OBC_NONE, OBC_DIRECTION_N, OBC_DIRECTION_S, OBC_DIRECTION_E, OBC_DIRECTION_W
integer, parameter :: OBC_NONE = 0, OBC_DIRECTION_N = 100, OBC_DIRECTION_S = 200, OBC_DIRECTION_E = 300, OBC_DIRECTION_W = 400
type :: OBC_segment_data_type
  real :: resrv_lfac_in = 1., resrv_lfac_out = 1.
end type OBC_segment_data_type
type :: OBC_segment_tracer_type
  real :: OBC_inflow_conc = 0.0
  real, allocatable :: t(:,:,:), tres(:,:,:)
  real :: scale = 1.0
  integer :: ntr_index = -1, fd_index = -1
end type OBC_segment_tracer_type
type :: segment_tracer_registry_type
  integer :: ntseg = 0
  type(OBC_segment_tracer_type) :: Tr(50)
end type segment_tracer_registry_type
type :: OBC_segment_type
  type(segment_tracer_registry_type), pointer :: tr_Reg => NULL()
  logical :: Flather = .false., radiation = .false., oblique = .false., nudged = .false., specified = .false., open = .false.
  logical :: gradient = .false., on_pe = .false., is_N_or_S = .false., is_E_or_W = .false.
  logical :: radiation_tan = .false., radiation_grad = .false., nudged_tan = .false., nudged_grad = .false.
  logical :: oblique_tan = .false., oblique_grad = .false.
  integer :: direction = 0
  type(hor_index_type) :: HI
  real, allocatable :: normal_vel(:,:,:), normal_trans(:,:,:), normal_vel_bt(:,:), tangential_vel(:,:,:), tangential_grad(:,:,:)
  real, allocatable :: SSH(:,:), nudged_normal_vel(:,:,:), nudged_tangential_vel(:,:,:), nudged_tangential_grad(:,:,:)
  real :: Velocity_nudging_timescale_in = 0.0, Velocity_nudging_timescale_out = 0.0
  real :: Tr_InvLscale_in = 0.0, Tr_InvLscale_out = 0.0
  type(OBC_segment_data_type), pointer :: field(:) => NULL()
end type OBC_segment_type
type :: ocean_OBC_type
  logical :: OBC_pe = .false.
  logical :: open_u_BCs_exist_globally = .false., open_v_BCs_exist_globally = .false.
  logical :: Flather_u_BCs_exist_globally = .false., Flather_v_BCs_exist_globally = .false.
  logical :: specified_u_BCs_exist_globally = .false., specified_v_BCs_exist_globally = .false.
  type(OBC_segment_type), allocatable :: segment(:)
  integer, allocatable :: segnum_u(:,:), segnum_v(:,:)
  integer :: number_of_segments = 0
  logical :: update_OBC = .false., oblique_BCs_exist_globally = .false., radiation_BCs_exist_globally = .false.
  logical :: ramp = .false., zero_vorticity = .false., freeslip_vorticity = .false., computed_vorticity = .false.
  logical :: specified_vorticity = .false., zero_strain = .false., freeslip_strain = .false., computed_strain = .false.
  logical :: specified_strain = .false., zero_biharmonic = .false.
  real :: gamma_uv = 0.3, rx_max = 1.0
  real, allocatable :: rx_normal(:,:,:), ry_normal(:,:,:)
  real, allocatable :: tres_x(:,:,:,:), tres_y(:,:,:,:)
  real, allocatable :: rx_oblique_u(:,:,:), ry_oblique_u(:,:,:), cff_normal_u(:,:,:), rx_oblique_v(:,:,:), ry_oblique_v(:,:,:), cff_normal_v(:,:,:)
end type ocean_OBC_type
contains
logical function open_boundary_query(OBC, apply_open_OBC, apply_specified_OBC, apply_Flather_OBC, apply_nudged_OBC, needs_ext_seg_data)
  type(ocean_OBC_type), pointer    :: OBC
  logical, optional,    intent(in) :: apply_open_OBC, apply_specified_OBC, apply_Flather_OBC, apply_nudged_OBC, needs_ext_seg_data
  open_boundary_query = .false.
end function open_boundary_query
subroutine open_boundary_zero_normal_flow(OBC, G, GV, u, v)
  type(ocean_OBC_type),                       pointer       :: OBC
  type(ocean_grid_type),                      intent(inout) :: G
  type(verticalGrid_type),                    intent(in)    :: GV
  real, dimension(G%IsdB:G%IedB,G%jsd:G%jed,GV%ke), intent(inout) :: u
  real, dimension(G%isd:G%ied,G%JsdB:G%JedB,GV%ke), intent(inout) :: v
end subroutine open_boundary_zero_normal_flow
subroutine open_boundary_test_extern_h(G, GV, OBC, h)
  type(ocean_grid_type),                     intent(in)    :: G
  type(verticalGrid_type),                   intent(in)    :: GV
  type(ocean_OBC_type),                      pointer       :: OBC
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke),intent(inout) :: h
end subroutine open_boundary_test_extern_h
subroutine update_OBC_ramp(Time, OBC, US, activate)
  type(time_type), target, intent(in)    :: Time
  type(ocean_OBC_type),    intent(inout) :: OBC
  type(unit_scale_type),   intent(in)    :: US
  logical, optional,       intent(in)    :: activate
end subroutine update_OBC_ramp
subroutine radiation_open_bdry_conds(OBC, u_new, u_old, v_new, v_old, G, GV, US, dt)
  type(ocean_grid_type),                      intent(inout) :: G
  type(verticalGrid_type),                    intent(in)    :: GV
  type(ocean_OBC_type),                       pointer       :: OBC
  real, dimension(G%IsdB:G%IedB,G%jsd:G%jed,GV%ke), intent(inout) :: u_new
  real, dimension(G%IsdB:G%IedB,G%jsd:G%jed,GV%ke), intent(in)    :: u_old
  real, dimension(G%isd:G%ied,G%JsdB:G%JedB,GV%ke), intent(inout) :: v_new
  real, dimension(G%isd:G%ied,G%JsdB:G%JedB,GV%ke), intent(in)    :: v_old
  type(unit_scale_type),                      intent(in)    :: US
  real,                                       intent(in)    :: dt
end subroutine radiation_open_bdry_conds
end module MOM_open_boundary

#include "mom6_stubs_after_obc.inc"
#endif


module MOM_wave_interface
use MOM_grid, only : ocean_grid_type
use MOM_verticalGrid, only : verticalGrid_type
use MOM_unit_scaling, only : unit_scale_type
implicit none ; private
public :: Wave_parameters_CS, Stokes_PGF
type :: Wave_parameters_CS
  logical :: Stokes_VF = .false., Stokes_PGF = .false., Passive_Stokes_PGF = .false., Passive_Stokes_VF = .false., Stokes_DDT = .false.
  real, allocatable, dimension(:,:,:) :: Us_x, Us_y
  real, allocatable, dimension(:,:,:) :: Ustk_Hb, Vstk_Hb      !< (the surface-band Stokes drift: FPMIX / Stokes mixing only)
  integer :: NumBands = 0
  real, allocatable, dimension(:) :: WaveNum_Cen
end type Wave_parameters_CS
contains
subroutine Stokes_PGF(G, GV, US, dz, u, v, PFu_Stokes, PFv_Stokes, CS)
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(unit_scale_type),   intent(in)    :: US
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke),   intent(in)    :: dz
  real, dimension(G%IsdB:G%IedB,G%jsd:G%jed,GV%ke), intent(in)    :: u
  real, dimension(G%isd:G%ied,G%JsdB:G%JedB,GV%ke), intent(in)    :: v
  real, dimension(G%IsdB:G%IedB,G%jsd:G%jed,GV%ke), intent(out)   :: PFu_Stokes
  real, dimension(G%isd:G%ied,G%JsdB:G%JedB,GV%ke), intent(out)   :: PFv_Stokes
  type(Wave_parameters_CS), pointer      :: CS
  PFu_Stokes(:,:,:) = 0.0 ; PFv_Stokes(:,:,:) = 0.0
end subroutine Stokes_PGF
end module MOM_wave_interface

#ifndef REF_INTERFACE_HEIGHTS
module MOM_interface_heights
use MOM_grid, only : ocean_grid_type
use MOM_verticalGrid, only : verticalGrid_type
use MOM_unit_scaling, only : unit_scale_type
use MOM_variables, only : thermo_var_ptrs
implicit none ; private
public :: thickness_to_dz, find_col_avg_SpV
interface thickness_to_dz
  module procedure thickness_to_dz_3d, thickness_to_dz_jslice
end interface thickness_to_dz
contains
subroutine thickness_to_dz_3d(h, tv, dz, G, GV, US, halo_size)
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke), intent(in)    :: h
  type(thermo_var_ptrs),   intent(in)    :: tv
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke), intent(inout) :: dz
  type(unit_scale_type),   intent(in)    :: US
  integer,       optional, intent(in)    :: halo_size
  dz(:,:,:) = GV%H_to_Z * h(:,:,:)
end subroutine thickness_to_dz_3d
subroutine thickness_to_dz_jslice(h, tv, dz, j, G, GV, halo_size)
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke), intent(in)    :: h
  type(thermo_var_ptrs),   intent(in)    :: tv
  real, dimension(G%isd:G%ied,GV%ke), intent(inout) :: dz
  integer,                 intent(in)    :: j
  integer,       optional, intent(in)    :: halo_size
  dz(:,:) = GV%H_to_Z * h(:,j,:)
end subroutine thickness_to_dz_jslice
subroutine find_col_avg_SpV(h, SpV_avg, tv, G, GV, US, halo_size)
  type(ocean_grid_type),    intent(in)    :: G
  type(verticalGrid_type),  intent(in)    :: GV
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke), intent(in)    :: h
  real, dimension(G%isd:G%ied,G%jsd:G%jed),       intent(inout) :: SpV_avg
  type(thermo_var_ptrs),    intent(in)    :: tv
  type(unit_scale_type),    intent(in)    :: US
  integer,        optional, intent(in)    :: halo_size
end subroutine find_col_avg_SpV
end module MOM_interface_heights
#endif


module MOM_debugging
use MOM_grid, only : ocean_grid_type
use MOM_hor_index, only : hor_index_type
implicit none ; private
public :: hchksum, uvchksum, Bchksum, check_redundant, check_column_integrals
interface hchksum
  module procedure chksum_h_3d, chksum_h_2d
end interface
interface Bchksum
  module procedure chksum_B_3d, chksum_B_2d
end interface
interface uvchksum
  module procedure chksum_uv_3d, chksum_uv_2d
end interface
interface check_redundant
  module procedure check_redundant_vC3d, check_redundant_vC2d
end interface
contains
!> (ALE_offline_inputs only, which no test reaches)
logical function check_column_integrals(nk_1, field_1, nk_2, field_2, missing_value)
  integer,               intent(in) :: nk_1, nk_2
  real, dimension(nk_1), intent(in) :: field_1
  real, dimension(nk_2), intent(in) :: field_2
  real, optional,        intent(in) :: missing_value
  check_column_integrals = .false.
  error stop "check_column_integrals stand-in: not provided"
end function check_column_integrals
subroutine chksum_h_3d(array_m, mesg, HI_m, haloshift, omit_corners, scale, logunit)
  type(hor_index_type),    target,   intent(in) :: HI_m
  real, dimension(HI_m%isd:,HI_m%jsd:,:), target, intent(in) :: array_m
  character(len=*),                  intent(in) :: mesg
  integer,                 optional, intent(in) :: haloshift, logunit
  logical,                 optional, intent(in) :: omit_corners
  real,                    optional, intent(in) :: scale
end subroutine chksum_h_3d
subroutine chksum_h_2d(array_m, mesg, HI_m, haloshift, omit_corners, scale, logunit)
  type(hor_index_type),    target,   intent(in) :: HI_m
  real, dimension(HI_m%isd:,HI_m%jsd:), target, intent(in) :: array_m
  character(len=*),                  intent(in) :: mesg
  integer,                 optional, intent(in) :: haloshift, logunit
  logical,                 optional, intent(in) :: omit_corners
  real,                    optional, intent(in) :: scale
end subroutine chksum_h_2d
subroutine chksum_B_3d(array_m, mesg, HI_m, haloshift, symmetric, omit_corners, scale, logunit)
  type(hor_index_type),    target,   intent(in) :: HI_m
  real, dimension(HI_m%IsdB:,HI_m%JsdB:,:), target, intent(in) :: array_m
  character(len=*),                  intent(in) :: mesg
  integer,                 optional, intent(in) :: haloshift, logunit
  logical,                 optional, intent(in) :: symmetric, omit_corners
  real,                    optional, intent(in) :: scale
end subroutine chksum_B_3d
subroutine chksum_B_2d(array_m, mesg, HI_m, haloshift, symmetric, omit_corners, scale, logunit)
  type(hor_index_type),    target,   intent(in) :: HI_m
  real, dimension(HI_m%IsdB:,HI_m%JsdB:), target, intent(in) :: array_m
  character(len=*),                  intent(in) :: mesg
  integer,                 optional, intent(in) :: haloshift, logunit
  logical,                 optional, intent(in) :: symmetric, omit_corners
  real,                    optional, intent(in) :: scale
end subroutine chksum_B_2d
subroutine chksum_uv_3d(mesg, arrayU, arrayV, HI, haloshift, symmetric, omit_corners, scale, logunit, scalar_pair)
  character(len=*),                    intent(in) :: mesg
  type(hor_index_type),      target,   intent(in) :: HI
  real, dimension(HI%IsdB:,HI%jsd:,:), target, intent(in) :: arrayU
  real, dimension(HI%isd:,HI%JsdB:,:), target, intent(in) :: arrayV
  integer,                   optional, intent(in) :: haloshift, logunit
  logical,                   optional, intent(in) :: symmetric, omit_corners, scalar_pair
  real,                      optional, intent(in) :: scale
end subroutine chksum_uv_3d
subroutine chksum_uv_2d(mesg, arrayU, arrayV, HI, haloshift, symmetric, omit_corners, scale, logunit, scalar_pair)
  character(len=*),                    intent(in) :: mesg
  type(hor_index_type),      target,   intent(in) :: HI
  real, dimension(HI%IsdB:,HI%jsd:), target, intent(in) :: arrayU
  real, dimension(HI%isd:,HI%JsdB:), target, intent(in) :: arrayV
  integer,                   optional, intent(in) :: haloshift, logunit
  logical,                   optional, intent(in) :: symmetric, omit_corners, scalar_pair
  real,                      optional, intent(in) :: scale
end subroutine chksum_uv_2d
subroutine check_redundant_vC3d(mesg, u_comp, v_comp, G, is, ie, js, je, direction, unscale)
  character(len=*),                    intent(in)    :: mesg
  type(ocean_grid_type),               intent(inout) :: G
  real, dimension(G%IsdB:,G%jsd:,:),   intent(in)    :: u_comp
  real, dimension(G%isd:,G%JsdB:,:),   intent(in)    :: v_comp
  integer,                   optional, intent(in)    :: is, ie, js, je, direction
  real,                      optional, intent(in)    :: unscale
end subroutine check_redundant_vC3d
subroutine check_redundant_vC2d(mesg, u_comp, v_comp, G, is, ie, js, je, direction, unscale)
  character(len=*),                    intent(in)    :: mesg
  type(ocean_grid_type),               intent(inout) :: G
  real, dimension(G%IsdB:,G%jsd:),     intent(in)    :: u_comp
  real, dimension(G%isd:,G%JsdB:),     intent(in)    :: v_comp
  integer,                   optional, intent(in)    :: is, ie, js, je, direction
  real,                      optional, intent(in)    :: unscale
end subroutine check_redundant_vC2d
end module MOM_debugging

module MOM_checksums
use MOM_debugging, only : hchksum, Bchksum, uvchksum
implicit none ; private
public :: chksum0, hchksum, Bchksum, uvchksum
contains
subroutine chksum0(scalar, mesg, scale, logunit, unscale)
  real,              intent(in) :: scalar
  character(len=*),  intent(in) :: mesg
  real,    optional, intent(in) :: scale, unscale
  integer, optional, intent(in) :: logunit
end subroutine chksum0
end module MOM_checksums


module MOM_checksum_packages
use MOM_grid, only : ocean_grid_type
use MOM_verticalGrid, only : verticalGrid_type
use MOM_unit_scaling, only : unit_scale_type
use MOM_variables, only : thermo_var_ptrs
implicit none ; private
public :: MOM_state_chksum, MOM_thermo_chksum, MOM_accel_chksum
interface MOM_state_chksum
  module procedure MOM_state_chksum_5arg, MOM_state_chksum_3arg
end interface
contains
subroutine MOM_state_chksum_5arg(mesg, u, v, h, uh, vh, G, GV, US, haloshift, symmetric, omit_corners, vel_scale)
  character(len=*),        intent(in) :: mesg
  type(ocean_grid_type),   intent(in) :: G
  type(verticalGrid_type), intent(in) :: GV
  real, dimension(G%IsdB:G%IedB,G%jsd:G%jed,GV%ke), intent(in) :: u, uh
  real, dimension(G%isd:G%ied,G%JsdB:G%JedB,GV%ke), intent(in) :: v, vh
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke),   intent(in) :: h
  type(unit_scale_type),   intent(in) :: US
  integer,       optional, intent(in) :: haloshift
  logical,       optional, intent(in) :: symmetric, omit_corners
  real,          optional, intent(in) :: vel_scale
end subroutine MOM_state_chksum_5arg
subroutine MOM_state_chksum_3arg(mesg, u, v, h, G, GV, US, haloshift, symmetric, omit_corners)
  character(len=*),        intent(in) :: mesg
  type(ocean_grid_type),   intent(in) :: G
  type(verticalGrid_type), intent(in) :: GV
  real, dimension(G%IsdB:G%IedB,G%jsd:G%jed,GV%ke), intent(in) :: u
  real, dimension(G%isd:G%ied,G%JsdB:G%JedB,GV%ke), intent(in) :: v
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke),   intent(in) :: h
  type(unit_scale_type),   intent(in) :: US
  integer,       optional, intent(in) :: haloshift
  logical,       optional, intent(in) :: symmetric, omit_corners
end subroutine MOM_state_chksum_3arg
subroutine MOM_thermo_chksum(mesg, tv, G, US, haloshift, omit_corners)
  character(len=*),         intent(in) :: mesg
  type(thermo_var_ptrs),    intent(in) :: tv
  type(ocean_grid_type),    intent(in) :: G
  type(unit_scale_type),    intent(in) :: US
  integer,        optional, intent(in) :: haloshift
  logical,        optional, intent(in) :: omit_corners
end subroutine MOM_thermo_chksum
subroutine MOM_accel_chksum(mesg, CAu, CAv, PFu, PFv, diffu, diffv, G, GV, US, pbce, u_accel_bt, v_accel_bt, symmetric)
  character(len=*),         intent(in) :: mesg
  type(ocean_grid_type),    intent(in) :: G
  type(verticalGrid_type),  intent(in) :: GV
  real, dimension(G%IsdB:G%IedB,G%jsd:G%jed,GV%ke), intent(in) :: CAu, PFu, diffu
  real, dimension(G%isd:G%ied,G%JsdB:G%JedB,GV%ke), intent(in) :: CAv, PFv, diffv
  type(unit_scale_type),    intent(in) :: US
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke),   optional, intent(in) :: pbce
  real, dimension(G%IsdB:G%IedB,G%jsd:G%jed,GV%ke), optional, intent(in) :: u_accel_bt
  real, dimension(G%isd:G%ied,G%JsdB:G%JedB,GV%ke), optional, intent(in) :: v_accel_bt
  logical,        optional, intent(in) :: symmetric
end subroutine MOM_accel_chksum
end module MOM_checksum_packages

!> The Montgomery-potential pressure force stays the reference's own module in a MOM6 tree (it is not on the path of
!! ANALYTIC_FV_PGF = True); this is its interface as MOM_PressureForce.F90 uses it.
#ifndef REF_PF_MONT
module MOM_PressureForce_Mont
use MOM_grid, only : ocean_grid_type
use MOM_verticalGrid, only : verticalGrid_type
use MOM_unit_scaling, only : unit_scale_type
use MOM_variables, only : thermo_var_ptrs
use MOM_time_manager, only : time_type
use MOM_file_parser, only : param_file_type
use MOM_diag_mediator, only : diag_ctrl
use MOM_self_attr_load, only : SAL_CS
use MOM_tidal_forcing, only : tidal_forcing_CS
use MOM_error_handler, only : MOM_error, FATAL
implicit none ; private
public :: PressureForce_Mont_CS, PressureForce_Mont_Bouss, PressureForce_Mont_nonBouss, PressureForce_Mont_init
type :: PressureForce_Mont_CS
  logical :: initialized = .false.
end type PressureForce_Mont_CS
contains
subroutine PressureForce_Mont_Bouss(h, tv, PFu, PFv, G, GV, US, CS, p_atm, pbce, eta)
  type(ocean_grid_type),                      intent(in)  :: G
  type(verticalGrid_type),                    intent(in)  :: GV
  type(unit_scale_type),                      intent(in)  :: US
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke),   intent(in)  :: h
  type(thermo_var_ptrs),                      intent(in)  :: tv
  real, dimension(G%IsdB:G%IedB,G%jsd:G%jed,GV%ke), intent(out) :: PFu
  real, dimension(G%isd:G%ied,G%JsdB:G%JedB,GV%ke), intent(out) :: PFv
  type(PressureForce_Mont_CS),                intent(inout) :: CS
  real, dimension(:,:),                       pointer     :: p_atm
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke), optional, intent(out) :: pbce
  real, dimension(G%isd:G%ied,G%jsd:G%jed),       optional, intent(out) :: eta
  call MOM_error(FATAL, "PressureForce_Mont_Bouss: a stand-in; ANALYTIC_FV_PGF = False is the reference's own module.")
end subroutine PressureForce_Mont_Bouss
subroutine PressureForce_Mont_nonBouss(h, tv, PFu, PFv, G, GV, US, CS, p_atm, pbce, eta)
  type(ocean_grid_type),                      intent(in)  :: G
  type(verticalGrid_type),                    intent(in)  :: GV
  type(unit_scale_type),                      intent(in)  :: US
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke),   intent(in)  :: h
  type(thermo_var_ptrs),                      intent(in)  :: tv
  real, dimension(G%IsdB:G%IedB,G%jsd:G%jed,GV%ke), intent(out) :: PFu
  real, dimension(G%isd:G%ied,G%JsdB:G%JedB,GV%ke), intent(out) :: PFv
  type(PressureForce_Mont_CS),                intent(inout) :: CS
  real, dimension(:,:),                       pointer     :: p_atm
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke), optional, intent(out) :: pbce
  real, dimension(G%isd:G%ied,G%jsd:G%jed),       optional, intent(out) :: eta
  call MOM_error(FATAL, "PressureForce_Mont_nonBouss: a stand-in; ANALYTIC_FV_PGF = False is the reference's own module.")
end subroutine PressureForce_Mont_nonBouss
subroutine PressureForce_Mont_init(Time, G, GV, US, param_file, diag, CS, SAL_CSp, tides_CSp)
  type(time_type), target, intent(in)    :: Time
  type(ocean_grid_type),   intent(in)    :: G
  type(verticalGrid_type), intent(in)    :: GV
  type(unit_scale_type),   intent(in)    :: US
  type(param_file_type),   intent(in)    :: param_file
  type(diag_ctrl), target, intent(inout) :: diag
  type(PressureForce_Mont_CS), intent(inout) :: CS
  type(SAL_CS), intent(in), target, optional :: SAL_CSp
  type(tidal_forcing_CS), intent(in), target, optional :: tides_CSp
  CS%initialized = .true.
end subroutine PressureForce_Mont_init
end module MOM_PressureForce_Mont
#endif


module MOM_spatial_means      ! src/diagnostics/MOM_spatial_means.F90: the global integral MOM_hor_bnd_diffusion prints in its debugging branch
use MOM_grid, only : ocean_grid_type
use MOM_verticalGrid, only : verticalGrid_type
implicit none ; private
public :: global_mass_integral
contains
function global_mass_integral(h, G, GV, var, on_PE_only, scale, tmp_scale)
  type(ocean_grid_type),   intent(in)  :: G
  type(verticalGrid_type), intent(in)  :: GV
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke), intent(in) :: h
  real, dimension(G%isd:G%ied,G%jsd:G%jed,GV%ke), optional, intent(in) :: var
  logical,       optional, intent(in)  :: on_PE_only
  real,          optional, intent(in)  :: scale, tmp_scale
  real :: global_mass_integral
  global_mass_integral = 0.0
end function global_mass_integral
end module MOM_spatial_means
