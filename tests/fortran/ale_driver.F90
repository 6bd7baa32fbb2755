!> Drives the MOM_ALE shim the way step_MOM_thermo does (src/core/MOM.F90:1647-1700): ALE_init (parameters by name),
!! ALE_update_regrid_weights, ALE_regrid, ALE_remap_tracers on a two-tracer registry, ALE_remap_set_h_vel for the old and the
!! new grid, ALE_remap_velocities -- on plain host arrays.  tests/test_fortran_abi.py writes the input file and compares the
!! output with the oracle bit for bit.   Usage: ale_driver <input file> <output file> [NAME=VALUE ...]
program ale_driver
use, intrinsic :: iso_c_binding
use MOM_ALE,             only : ALE_CS, ALE_init, ALE_end, ALE_regrid, ALE_remap_tracers, ALE_remap_set_h_vel, ALE_remap_velocities, &
                                ALE_update_regrid_weights, ALE_set_extrap_boundaries
use MOM_domains,         only : MOM_domain_type
use MOM_file_parser,     only : param_file_type, param_set
use MOM_grid,            only : ocean_grid_type
use MOM_open_boundary,   only : ocean_OBC_type
use MOM_tracer_registry, only : tracer_registry_type
use MOM_unit_scaling,    only : unit_scale_type
use MOM_variables,       only : thermo_var_ptrs
use MOM_verticalGrid,    only : verticalGrid_type
#ifndef REFERENCE_KERNELS
use mom6hip_MOM_glue,    only : mom6hip_shared_context_end
#endif
implicit none

type(ocean_grid_type), target :: G
type(verticalGrid_type) :: GV
type(unit_scale_type) :: US
type(param_file_type) :: pf
type(ALE_CS), pointer :: CS => NULL()
type(tracer_registry_type), pointer :: Reg => NULL()
type(thermo_var_ptrs) :: tv
type(ocean_OBC_type), pointer :: OBC => NULL()
integer(c_int32_t) :: hdr(8)
integer :: ni, nj, nk, halo, u_in, u_out, isd, ied, jsd, jed
real :: scal(7), dt, max_depth
real, allocatable, dimension(:,:,:) :: u, v, h, h_new, dzRegrid, hu0, hv0, hu1, hv1
real, allocatable, dimension(:,:,:), target :: T, S
character(len=512) :: f_in, f_out, f_arg
character(len=32) :: str
integer :: m

call get_command_argument(1, f_in) ; call get_command_argument(2, f_out)
open(newunit=u_in, file=trim(f_in), access="stream", form="unformatted", status="old")
read(u_in) hdr
ni = hdr(1) ; nj = hdr(2) ; nk = hdr(3) ; halo = hdr(4)
isd = 1 ; ied = ni + 2*halo ; jsd = 1 ; jed = nj + 2*halo
G%isd = isd ; G%ied = ied ; G%jsd = jsd ; G%jed = jed ; G%IsdB = isd-1 ; G%IedB = ied ; G%JsdB = jsd-1 ; G%JedB = jed
G%isc = isd+halo ; G%iec = ied-halo ; G%jsc = jsd+halo ; G%jec = jed-halo
G%IscB = G%isc-1 ; G%IecB = G%iec ; G%JscB = G%jsc-1 ; G%JecB = G%jec ; G%ke = nk ; GV%ke = nk
G%first_direction = hdr(7)
allocate(G%Domain)
G%Domain%reentrant(1) = (hdr(5) /= 0) ; G%Domain%reentrant(2) = (hdr(6) /= 0)
G%Domain%nihalo = halo ; G%Domain%njhalo = halo ; G%Domain%niglobal = ni ; G%Domain%njglobal = nj
read(u_in) scal, dt
GV%Angstrom_H = scal(1) ; GV%H_subroundoff = scal(2) ; GV%dZ_subroundoff = scal(3) ; GV%H_to_Z = scal(4) ; GV%Z_to_H = scal(5)
GV%g_Earth = scal(6) ; GV%Rho0 = scal(7) ; GV%RZ_to_H = GV%Z_to_H / GV%Rho0 ; GV%H_to_RZ = GV%H_to_Z * GV%Rho0
allocate(G%mask2dT(isd:ied,jsd:jed), G%areaT(isd:ied,jsd:jed), G%IareaT(isd:ied,jsd:jed), G%dxT(isd:ied,jsd:jed), &
         G%dyT(isd:ied,jsd:jed), G%IdxT(isd:ied,jsd:jed), G%IdyT(isd:ied,jsd:jed), G%bathyT(isd:ied,jsd:jed))
allocate(G%mask2dCu(isd-1:ied,jsd:jed), G%dxCu(isd-1:ied,jsd:jed), G%dyCu(isd-1:ied,jsd:jed), G%dy_Cu(isd-1:ied,jsd:jed), &
         G%IdxCu(isd-1:ied,jsd:jed), G%IdyCu(isd-1:ied,jsd:jed), G%areaCu(isd-1:ied,jsd:jed), G%IareaCu(isd-1:ied,jsd:jed))
allocate(G%mask2dCv(isd:ied,jsd-1:jed), G%dxCv(isd:ied,jsd-1:jed), G%dyCv(isd:ied,jsd-1:jed), G%dx_Cv(isd:ied,jsd-1:jed), &
         G%IdxCv(isd:ied,jsd-1:jed), G%IdyCv(isd:ied,jsd-1:jed), G%areaCv(isd:ied,jsd-1:jed), G%IareaCv(isd:ied,jsd-1:jed))
allocate(G%mask2dBu(isd-1:ied,jsd-1:jed), G%dxBu(isd-1:ied,jsd-1:jed), G%dyBu(isd-1:ied,jsd-1:jed), G%areaBu(isd-1:ied,jsd-1:jed), &
         G%IareaBu(isd-1:ied,jsd-1:jed), G%CoriolisBu(isd-1:ied,jsd-1:jed), G%IdxBu(isd-1:ied,jsd-1:jed), G%IdyBu(isd-1:ied,jsd-1:jed))
read(u_in) G%mask2dT, G%areaT, G%IareaT, G%dxT, G%dyT, G%IdxT, G%IdyT, G%bathyT
read(u_in) G%mask2dCu, G%dxCu, G%dyCu, G%dy_Cu, G%IdxCu, G%IdyCu, G%areaCu, G%IareaCu
read(u_in) G%mask2dCv, G%dxCv, G%dyCv, G%dx_Cv, G%IdxCv, G%IdyCv, G%areaCv, G%IareaCv
read(u_in) G%mask2dBu, G%dxBu, G%dyBu, G%areaBu, G%IareaBu, G%CoriolisBu, G%IdxBu, G%IdyBu
allocate(u(isd-1:ied,jsd:jed,nk), v(isd:ied,jsd-1:jed,nk), h(isd:ied,jsd:jed,nk), T(isd:ied,jsd:jed,nk), S(isd:ied,jsd:jed,nk))
read(u_in) u, v, h, T, S
read(u_in) max_depth
close(u_in)
allocate(h_new(isd:ied,jsd:jed,nk), dzRegrid(isd:ied,jsd:jed,nk+1), hu0(isd-1:ied,jsd:jed,nk), hv0(isd:ied,jsd-1:jed,nk), &
         hu1(isd-1:ied,jsd:jed,nk), hv1(isd:ied,jsd-1:jed,nk))
h_new = 0.0 ; dzRegrid = 0.0 ; hu0 = 0.0 ; hv0 = 0.0 ; hu1 = 0.0 ; hv1 = 0.0

call param_set(pf, "REENTRANT_X", merge("True ", "False", hdr(5) /= 0))
call param_set(pf, "REENTRANT_Y", merge("True ", "False", hdr(6) /= 0))
call param_set(pf, "REGRIDDING_COORDINATE_MODE", "Z*")
call param_set(pf, "REMAPPING_SCHEME", "PPM_H4") ; call param_set(pf, "VELOCITY_REMAPPING_SCHEME", "PLM")
call param_set(pf, "REGRID_TIME_SCALE", "3600.0") ; call param_set(pf, "REGRID_FILTER_DEEP_DEPTH", "500.0")
call param_set(pf, "REMAP_BOUNDARY_EXTRAP", "True") ; call param_set(pf, "INIT_BOUNDARY_EXTRAP", "False")

do m = 3, command_argument_count()      ! further NAME=VALUE pairs of the parameter file (they replace the ones above)
  call get_command_argument(m, f_arg)
  if (index(f_arg, "=") > 1) call param_set(pf, f_arg(1:index(f_arg, "=")-1), trim(f_arg(index(f_arg, "=")+1:)))
enddo
#ifdef REFERENCE_KERNELS
! (built with -DREFERENCE_KERNELS -DREF_ALE: the reference's OWN MOM_ALE.F90, MOM_regridding.F90, MOM_remapping.F90 and the 22 files under them)
G%HI%isd = isd ; G%HI%ied = ied ; G%HI%jsd = jsd ; G%HI%jed = jed ; G%HI%IsdB = isd-1 ; G%HI%IedB = ied ; G%HI%JsdB = jsd-1 ; G%HI%JedB = jed
G%HI%isc = G%isc ; G%HI%iec = G%iec ; G%HI%jsc = G%jsc ; G%HI%jec = G%jec
G%HI%IscB = G%IscB ; G%HI%IecB = G%IecB ; G%HI%JscB = G%JscB ; G%HI%JecB = G%JecB
allocate(G%US) ; G%max_depth = max_depth
allocate(GV%Rlay(nk), GV%g_prime(nk+1), GV%sInterface(nk+1), GV%sLayer(nk))      ! (initialize_regridding reads GV%Rlay for the density range of a uniform
GV%g_prime(:) = 0.0 ; GV%g_prime(1) = GV%g_Earth                                  !  coordinate whatever the coordinate is, :350)
do m=1,nk ; GV%Rlay(m) = 1025.0 + 0.5*real(m-1) ; GV%sLayer(m) = real(m) ; enddo
do m=1,nk+1 ; GV%sInterface(m) = real(m) - 0.5 ; enddo
#endif
call ALE_init(pf, GV, US, max_depth, CS)
call ALE_set_extrap_boundaries(pf, CS)       ! MOM.F90:3136: the run switches to REMAP_BOUNDARY_EXTRAP after initialisation
allocate(Reg)
Reg%ntr = 2 ; Reg%Tr(1)%t => T ; Reg%Tr(2)%t => S ; Reg%Tr(1)%name = "T" ; Reg%Tr(2)%name = "S"
Reg%Tr(2)%conc_underflow = 1.0e-30

call ALE_update_regrid_weights(dt, CS)
call ALE_regrid(G, GV, US, h, h_new, dzRegrid, tv, CS)
call ALE_remap_tracers(CS, G, GV, h, h_new, Reg)
call ALE_remap_set_h_vel(CS, G, GV, h, hu0, hv0, OBC)
call ALE_remap_set_h_vel(CS, G, GV, h_new, hu1, hv1, OBC)
call ALE_remap_velocities(CS, G, GV, hu0, hv0, hu1, hv1, u, v)

open(newunit=u_out, file=trim(f_out), access="stream", form="unformatted", status="replace")
write(u_out) h_new, dzRegrid, T, S, hu1, hv1, u, v
close(u_out)
call ALE_end(CS)
#ifndef REFERENCE_KERNELS
call mom6hip_shared_context_end()
#endif
write(*,'(a)') "ale_driver ok"
end program ale_driver
