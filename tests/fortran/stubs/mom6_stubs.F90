!> TYPE-ONLY stand-ins for the MOM6 framework modules the shims of mom6_amd/fortran `use`, so that this repository can
!! compile and drive the shims on one PE without a MOM6 source tree.  They carry only the members and procedures the
!! shims touch, do no model work, pin nothing about the reference and are never used as an oracle.  Inside a MOM6 tree
!! the real modules take their place.

module MOM_error_handler
implicit none ; private
public :: MOM_error, MOM_mesg, FATAL, WARNING, NOTE, is_root_pe
integer, parameter :: NOTE = 0, WARNING = 1, FATAL = 2
contains
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

module MOM_coms
implicit none ; private
public :: num_PEs, PE_here, sum_across_PEs, min_across_PEs, max_across_PEs
interface sum_across_PEs
  module procedure sum_int_1d
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
implicit none ; private
public :: MOM_domain_type, pass_var, pass_vector, CENTER, EAST_FACE, NORTH_FACE, CORNER, group_pass_type
integer, parameter :: CENTER = 0, EAST_FACE = 1, NORTH_FACE = 2, CORNER = 3
type :: MOM_domain_type
  logical :: reentrant(2) = .false.   !< this stand-in's one-PE topology: its pass_var wraps re-entrant directions
  integer :: nihalo = 0, njhalo = 0, niglobal = 0, njglobal = 0
end type MOM_domain_type
type :: group_pass_type
  integer :: unused = 0
end type group_pass_type
interface pass_var
  module procedure pass_var_3d, pass_var_2d
end interface
contains
!> One PE: wrap the re-entrant directions (whole allocation with the halo; position gives the staggering)
subroutine pass_var_3d(array, MOM_dom, sideflag, complete, position, halo)
  real, dimension(:,:,:), intent(inout) :: array
  type(MOM_domain_type),  intent(inout) :: MOM_dom
  integer, optional,      intent(in)    :: sideflag, position, halo
  logical, optional,      intent(in)    :: complete
  integer :: k
  do k = 1, size(array, 3) ; call pass_var_2d(array(:,:,k), MOM_dom, position=position) ; enddo
end subroutine pass_var_3d
subroutine pass_var_2d(array, MOM_dom, sideflag, complete, position, halo)
  real, dimension(:,:),  intent(inout) :: array
  type(MOM_domain_type), intent(inout) :: MOM_dom
  integer, optional,     intent(in)    :: sideflag, position, halo
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
subroutine pass_vector(u_cmpt, v_cmpt, MOM_dom, direction, stagger, complete, halo)
  real, dimension(:,:,:), intent(inout) :: u_cmpt, v_cmpt
  type(MOM_domain_type),  intent(inout) :: MOM_dom
  integer, optional,      intent(in)    :: direction, stagger, halo
  logical, optional,      intent(in)    :: complete
  call pass_var_3d(u_cmpt, MOM_dom, position=EAST_FACE)
  call pass_var_3d(v_cmpt, MOM_dom, position=NORTH_FACE)
end subroutine pass_vector
end module MOM_domains

module MOM_grid
use MOM_domains, only : MOM_domain_type
implicit none ; private
public :: ocean_grid_type
type :: ocean_grid_type
  type(MOM_domain_type), pointer :: Domain => NULL()
  integer :: isc, iec, jsc, jec, isd, ied, jsd, jed, IscB, IecB, JscB, JecB, IsdB, IedB, JsdB, JedB, ke
  integer :: first_direction = 0
  integer :: idg_offset = 0, jdg_offset = 0
  logical :: symmetric = .true.
  real :: max_depth = 0.0, Z_ref = 0.0
  real, allocatable, dimension(:,:) :: mask2dT, areaT, IareaT, dxT, dyT, IdxT, IdyT, bathyT
  real, allocatable, dimension(:,:) :: mask2dCu, dxCu, dyCu, dy_Cu, IdxCu, IdyCu, areaCu, IareaCu
  real, allocatable, dimension(:,:) :: mask2dCv, dxCv, dyCv, dx_Cv, IdxCv, IdyCv, areaCv, IareaCv
  real, allocatable, dimension(:,:) :: mask2dBu, dxBu, dyBu, areaBu, IareaBu, CoriolisBu, IdxBu, IdyBu
end type ocean_grid_type
end module MOM_grid

module MOM_verticalGrid
implicit none ; private
public :: verticalGrid_type
type :: verticalGrid_type
  integer :: ke
  real :: Angstrom_Z = 1.0e-10, Angstrom_H = 1.0e-10, H_subroundoff = 1.0e-30, dZ_subroundoff = 1.0e-30, H_to_Z = 1.0, Z_to_H = 1.0, g_Earth = 9.8, &
          Rho0 = 1035.0, m_to_H = 1.0, H_to_m = 1.0, RZ_to_H = 1.0/1035.0, H_to_RZ = 1035.0, m2_s_to_HZ_T = 1.0
  integer :: nk_rho_varies = 0, nkml = 0
  logical :: Boussinesq = .true.
  real, allocatable :: Rlay(:), g_prime(:)
end type verticalGrid_type
end module MOM_verticalGrid

module MOM_unit_scaling
implicit none ; private
public :: unit_scale_type
type :: unit_scale_type
  real :: m_to_Z = 1.0, Z_to_m = 1.0, m_to_L = 1.0, L_to_m = 1.0, s_to_T = 1.0, T_to_s = 1.0, m_s_to_L_T = 1.0, L_T_to_m_s = 1.0, &
          R_to_kg_m3 = 1.0, kg_m3_to_R = 1.0, L_to_Z = 1.0, Z_to_L = 1.0, Pa_to_RL2_T2 = 1.0
end type unit_scale_type
end module MOM_unit_scaling

module MOM_time_manager
implicit none ; private
public :: time_type
type :: time_type
  integer :: seconds = 0, days = 0
end type time_type
end module MOM_time_manager

module MOM_diag_mediator
use MOM_time_manager, only : time_type
implicit none ; private
public :: diag_ctrl, time_type
type :: diag_ctrl
  integer :: unused = 0
end type diag_ctrl
end module MOM_diag_mediator

module MOM_cpu_clock
implicit none ; private
public :: cpu_clock_id, cpu_clock_begin, cpu_clock_end, CLOCK_MODULE, CLOCK_ROUTINE
integer, parameter :: CLOCK_MODULE = 1, CLOCK_ROUTINE = 2
contains
integer function cpu_clock_id(name, grain)
  character(len=*),  intent(in) :: name
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
public :: param_file_type, get_param, log_version, param_set, openParameterBlock, closeParameterBlock
character(len=64), save :: block_prefix = ''      !< "NAME%" inside openParameterBlock(NAME) ... closeParameterBlock
type :: param_file_type
  integer :: n = 0
  character(len=64)  :: names(256)
  character(len=128) :: values(256)
end type param_file_type
interface get_param
  module procedure get_param_logical, get_param_real, get_param_int, get_param_char, get_param_real_array
end interface
contains
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

module MOM_hor_index
implicit none ; private
public :: hor_index_type
type :: hor_index_type
  integer :: isc, iec, jsc, jec, isd, ied, jsd, jed, IscB, IecB, JscB, JecB, IsdB, IedB, JsdB, JedB
end type hor_index_type
end module MOM_hor_index

!> Restart registration that remembers nothing: every field reads as "not initialized" (a cold start)
module MOM_restart
implicit none ; private
public :: MOM_restart_CS, register_restart_field, query_initialized
type :: MOM_restart_CS
  integer :: nfields = 0
end type MOM_restart_CS
interface register_restart_field
  module procedure register_3d, register_2d, register_0d
end interface
interface query_initialized
  module procedure query_3d, query_2d, query_0d
end interface
contains
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

module MOM_forcing_type
implicit none ; private
public :: mech_forcing
type :: mech_forcing
  real, pointer, dimension(:,:) :: taux => NULL(), tauy => NULL()
  real, pointer, dimension(:,:) :: ustar => NULL()      !< the surface friction velocity [Z T-1]
end type mech_forcing
end module MOM_forcing_type

module MOM_self_attr_load
implicit none ; private
public :: SAL_CS
type :: SAL_CS
  integer :: unused = 0
end type SAL_CS
end module MOM_self_attr_load

#ifndef MOM6HIP_WITH_ALE_SHIM
module MOM_ALE
implicit none ; private
public :: ALE_CS
type :: ALE_CS
  integer :: unused = 0
end type ALE_CS
end module MOM_ALE
#endif

module MOM_tidal_forcing
implicit none ; private
public :: tidal_forcing_CS
type :: tidal_forcing_CS
  integer :: unused = 0
end type tidal_forcing_CS
end module MOM_tidal_forcing

module MOM_io
implicit none ; private
public :: directories
type :: directories
  character(len=240) :: output_directory = "."
end type directories
end module MOM_io

module MOM_barotropic_types_for_hor_visc
end module MOM_barotropic_types_for_hor_visc

module MOM_stochastics
implicit none ; private
public :: stochastic_CS
type :: stochastic_CS
  logical :: skeb_use_gm = .false.
end type stochastic_CS
end module MOM_stochastics

module MOM_EOS
implicit none ; private
public :: EOS_type
type :: EOS_type
  integer :: form_of_EOS = 0
end type EOS_type
end module MOM_EOS

module MOM_diabatic_driver
implicit none ; private
public :: diabatic_CS
type :: diabatic_CS
  integer :: unused = 0
end type diabatic_CS
end module MOM_diabatic_driver

module MOM_MEKE_types
implicit none ; private
public :: MEKE_type
type :: MEKE_type
  real, allocatable :: Kh(:,:), Ku(:,:), Au(:,:), mom_src(:,:), GME_snk(:,:), GM_src(:,:), MEKE(:,:), Rd_dx_h(:,:), Kh_diff(:,:)
  real :: KhTr_fac = 1.0, KhTh_fac = 1.0, backscatter_Ro_c = 0.0, backscatter_Ro_pow = 0.0
end type MEKE_type
end module MOM_MEKE_types

module MOM_lateral_mixing_coeffs
implicit none ; private
public :: VarMix_CS
type :: VarMix_CS
  logical :: use_variable_mixing = .false., Resoln_scaled_Kh = .false., Resoln_scaled_KhTr = .false., Resoln_scaled_KhTh = .false.
  logical :: Depth_scaled_KhTh = .false., use_stored_slopes = .false., khth_use_ebt_struct = .false., use_Visbeck = .false.
  logical :: use_QG_Leith_GM = .false.
  real, allocatable, dimension(:,:) :: L2u, L2v, SN_u, SN_v, Res_fn_u, Res_fn_v, Res_fn_h, Rd_dx_h, cg1
  real, allocatable, dimension(:,:,:) :: slope_x, slope_y
end type VarMix_CS
end module MOM_lateral_mixing_coeffs

module MOM_open_boundary
implicit none ; private
public :: ocean_OBC_type
type :: ocean_OBC_type
  integer :: number_of_segments = 0
end type ocean_OBC_type
end module MOM_open_boundary

module MOM_boundary_update
implicit none ; private
public :: update_OBC_CS
type :: update_OBC_CS
  integer :: unused = 0
end type update_OBC_CS
end module MOM_boundary_update

module MOM_wave_interface
implicit none ; private
public :: Wave_parameters_CS
type :: Wave_parameters_CS
  logical :: Stokes_VF = .false.
end type Wave_parameters_CS
end module MOM_wave_interface

module MOM_tracer_registry
implicit none ; private
public :: tracer_registry_type, tracer_type
type :: tracer_type
  real, dimension(:,:,:), pointer :: t => NULL()
  real :: conc_underflow = 0.0
  real, dimension(:,:,:), pointer :: ad_x => NULL(), ad_y => NULL(), advection_xy => NULL()
  real, dimension(:,:),   pointer :: ad2d_x => NULL(), ad2d_y => NULL()
  real, dimension(:,:,:), pointer :: df_x => NULL(), df_y => NULL()
  real, dimension(:,:),   pointer :: df2d_x => NULL(), df2d_y => NULL()
  character(len=32) :: name = ""
end type tracer_type
type :: tracer_registry_type
  integer :: ntr = 0
  type(tracer_type) :: Tr(16)
end type tracer_registry_type
end module MOM_tracer_registry

module MOM_variables
use MOM_domains, only : group_pass_type
use MOM_EOS, only : EOS_type
implicit none ; private
public :: BT_cont_type, porous_barrier_type, accel_diag_ptrs, cont_diag_ptrs, thermo_var_ptrs, vertvisc_type, &
          ocean_internal_state, alloc_BT_cont_type
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
  real, pointer, dimension(:,:,:) :: gradKEu => NULL(), gradKEv => NULL(), rv_x_u => NULL(), rv_x_v => NULL()
end type accel_diag_ptrs
type :: cont_diag_ptrs
  real, pointer, dimension(:,:,:) :: uh => NULL(), vh => NULL(), uhGM => NULL(), vhGM => NULL()
end type cont_diag_ptrs
type :: thermo_var_ptrs
  real, pointer, dimension(:,:,:) :: T => NULL(), S => NULL()
  real, pointer, dimension(:,:) :: p_surf => NULL()
  type(EOS_type), pointer :: eqn_of_state => NULL()      !< associated = an equation of state is used (use_EOS)
  real :: P_Ref = 2.0e7
end type thermo_var_ptrs
type :: vertvisc_type
  real :: Prandtl_turb = 1.0
  real, allocatable, dimension(:,:) :: Kv_bbl_u, Kv_bbl_v, bbl_thick_u, bbl_thick_v
  real, allocatable, dimension(:,:) :: nkml_visc_u, nkml_visc_v
  real, allocatable, dimension(:,:,:) :: Ray_u, Ray_v
  real, pointer, dimension(:,:,:) :: Kv_shear => NULL(), Kv_shear_Bu => NULL()
  real, pointer, dimension(:,:) :: h_ML => NULL()
end type vertvisc_type
type :: ocean_internal_state
  integer :: unused = 0
end type ocean_internal_state
contains
subroutine alloc_BT_cont_type(BT_cont, isd, ied, jsd, jed, nz, alloc_faces)
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
end subroutine alloc_BT_cont_type
end module MOM_variables
