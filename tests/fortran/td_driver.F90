!> Drives the MOM_thickness_diffuse shim the way step_MOM does (src/core/MOM.F90:1149-1165): an ocean_grid_type filled from a file
!! written by tests/test_thickness_diffuse.py, thickness_diffuse_init from a parameter list (KEY = VALUE lines), then
!!   thickness_diffuse(h, uhtr, vhtr, tv, dt, G, GV, US, MEKE, VarMix, CDp, CS, STOCH)
!! on plain host arrays; h, uhtr, vhtr, CDp%uhGM, CDp%vhGM (and MEKE%GM_src) go to the output file, which the test compares with
!! the oracle bit for bit.   Usage: td_driver <input file> <output file> <parameter file>
program td_driver
use, intrinsic :: iso_c_binding
use MOM_thickness_diffuse, only : thickness_diffuse, thickness_diffuse_init, thickness_diffuse_end, thickness_diffuse_CS
use MOM_diag_mediator,  only : diag_ctrl, time_type
use MOM_domains,        only : MOM_domain_type
use MOM_file_parser,    only : param_file_type, param_set
use MOM_grid,           only : ocean_grid_type
use MOM_lateral_mixing_coeffs, only : VarMix_CS
use MOM_MEKE_types,     only : MEKE_type
use MOM_stochastics,    only : stochastic_CS
use MOM_unit_scaling,   only : unit_scale_type
use MOM_variables,      only : cont_diag_ptrs, thermo_var_ptrs
use MOM_verticalGrid,   only : verticalGrid_type
#ifdef REFERENCE_KERNELS
use MOM_EOS,            only : EOS_init      ! (built with -DREFERENCE_KERNELS -DREF_EOS -DREF_INTERFACE_HEIGHTS: the reference's OWN MOM_thickness_diffuse.F90,
                                             ! MOM_isopycnal_slopes.F90, MOM_interface_heights.F90, MOM_density_integrals.F90 and MOM_EOS)
#else
use mom6hip_MOM_glue,   only : mom6hip_shared_context_end, mom6hip_shared_context, mom6hip_mirrors_to_host, mom6hip_mirrors_end
#endif
implicit none

type(ocean_grid_type), target :: G
type(verticalGrid_type) :: GV
type(unit_scale_type) :: US
type(param_file_type) :: pf
type(time_type), target :: Time
type(diag_ctrl), target :: diag
type(thickness_diffuse_CS) :: CS
type(thermo_var_ptrs) :: tv
type(MEKE_type) :: MEKE
type(VarMix_CS) :: VarMix
type(cont_diag_ptrs) :: CDp
type(stochastic_CS) :: STOCH
integer(c_int32_t) :: hdr(8), opt(8)
integer :: ni, nj, nk, halo, u_in, u_out, u_par, isd, ied, jsd, jed, ios, eq, m
real :: scal(7), dt
real, allocatable, dimension(:,:,:) :: h, uhtr, vhtr
character(len=512) :: f_in, f_out, f_par, line

call get_command_argument(1, f_in) ; call get_command_argument(2, f_out) ; call get_command_argument(3, f_par)
open(newunit=u_in, file=trim(f_in), access="stream", form="unformatted", status="old")
read(u_in) hdr
ni = hdr(1) ; nj = hdr(2) ; nk = hdr(3) ; halo = hdr(4)
isd = 1 ; ied = ni + 2*halo ; jsd = 1 ; jed = nj + 2*halo
G%isd = isd ; G%ied = ied ; G%jsd = jsd ; G%jed = jed ; G%IsdB = isd-1 ; G%IedB = ied ; G%JsdB = jsd-1 ; G%JedB = jed
G%isc = isd+halo ; G%iec = ied-halo ; G%jsc = jsd+halo ; G%jec = jed-halo
G%IscB = G%isc-1 ; G%IecB = G%iec ; G%JscB = G%jsc-1 ; G%JecB = G%jec ; G%ke = nk ; GV%ke = nk
G%first_direction = hdr(7) ; G%symmetric = .true.
allocate(G%Domain)
G%Domain%reentrant(1) = (hdr(5) /= 0) ; G%Domain%reentrant(2) = (hdr(6) /= 0)
G%Domain%nihalo = halo ; G%Domain%njhalo = halo ; G%Domain%niglobal = ni ; G%Domain%njglobal = nj
read(u_in) scal, dt
GV%Angstrom_H = scal(1) ; GV%H_subroundoff = scal(2) ; GV%dZ_subroundoff = scal(3) ; GV%H_to_Z = scal(4) ; GV%Z_to_H = scal(5)
GV%g_Earth = scal(6) ; GV%Rho0 = scal(7)
! opt = [use_EOS, MEKE%Kh, Visbeck, Resoln_scaled_KhTh, use_stored_slopes, MEKE%GM_src, GV%nkml, use_variable_mixing]
read(u_in) opt
GV%nkml = opt(7)

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

allocate(h(isd:ied,jsd:jed,nk), uhtr(isd-1:ied,jsd:jed,nk), vhtr(isd:ied,jsd-1:jed,nk))
allocate(tv%T(isd:ied,jsd:jed,nk), tv%S(isd:ied,jsd:jed,nk), GV%Rlay(nk))
read(u_in) h, tv%T, tv%S, GV%Rlay
uhtr = 0.0 ; vhtr = 0.0
if (opt(1) /= 0) allocate(tv%eqn_of_state)
allocate(MEKE%Kh(isd:ied,jsd:jed), VarMix%L2u(isd-1:ied,jsd:jed), VarMix%L2v(isd:ied,jsd-1:jed), VarMix%SN_u(isd-1:ied,jsd:jed), &
         VarMix%SN_v(isd:ied,jsd-1:jed), VarMix%Res_fn_u(isd-1:ied,jsd:jed), VarMix%Res_fn_v(isd:ied,jsd-1:jed), &
         VarMix%slope_x(isd-1:ied,jsd:jed,nk+1), VarMix%slope_y(isd:ied,jsd-1:jed,nk+1))
read(u_in) MEKE%Kh, VarMix%L2u, VarMix%L2v, VarMix%SN_u, VarMix%SN_v, VarMix%Res_fn_u, VarMix%Res_fn_v, VarMix%slope_x, VarMix%slope_y
if (iand(hdr(8), 1) /= 0) then      ! VarMix%cg1 (KHTH_USE_FGNV_STREAMFUNCTION)
  allocate(VarMix%cg1(isd:ied,jsd:jed)) ; read(u_in) VarMix%cg1
endif
if (iand(hdr(8), 2) /= 0) then      ! VarMix%Depth_fn_u / _v (DEPTH_SCALED_KHTH)
  allocate(VarMix%Depth_fn_u(isd-1:ied,jsd:jed), VarMix%Depth_fn_v(isd:ied,jsd-1:jed)) ; read(u_in) VarMix%Depth_fn_u, VarMix%Depth_fn_v
  VarMix%Depth_scaled_KhTh = .true.
endif
close(u_in)
if (opt(2) == 0) deallocate(MEKE%Kh)
VarMix%use_variable_mixing = (opt(8) /= 0) ; VarMix%use_Visbeck = (opt(3) /= 0) ; VarMix%Resoln_scaled_KhTh = (opt(4) /= 0)
VarMix%use_stored_slopes = (opt(5) /= 0)
if (opt(6) /= 0) then ; allocate(MEKE%GM_src(isd:ied,jsd:jed)) ; MEKE%GM_src(:,:) = 7.0 ; endif
allocate(CDp%uhGM(isd-1:ied,jsd:jed,nk), CDp%vhGM(isd:ied,jsd-1:jed,nk)) ; CDp%uhGM = 0.0 ; CDp%vhGM = 0.0

call param_set(pf, "REENTRANT_X", merge("True ", "False", hdr(5) /= 0))
call param_set(pf, "REENTRANT_Y", merge("True ", "False", hdr(6) /= 0))
open(newunit=u_par, file=trim(f_par), status="old", action="read")
do
  read(u_par, '(a)', iostat=ios) line
  if (ios /= 0) exit
  eq = index(line, "=")
  if (eq > 1 .and. line(1:1) /= "!") call param_set(pf, trim(adjustl(line(1:eq-1))), trim(adjustl(line(eq+1:))))
enddo
close(u_par)

#ifdef REFERENCE_KERNELS
G%HI%isd = isd ; G%HI%ied = ied ; G%HI%jsd = jsd ; G%HI%jed = jed ; G%HI%IsdB = isd-1 ; G%HI%IedB = ied ; G%HI%JsdB = jsd-1 ; G%HI%JedB = jed
G%HI%isc = G%isc ; G%HI%iec = G%iec ; G%HI%jsc = G%jsc ; G%HI%jec = G%jec
G%HI%IscB = G%IscB ; G%HI%IecB = G%IecB ; G%HI%JscB = G%JscB ; G%HI%JecB = G%JecB
allocate(G%OBCmaskCu(isd-1:ied,jsd:jed), G%OBCmaskCv(isd:ied,jsd-1:jed))      ! no open boundaries: the masks of the faces (MOM_grid.F90)
G%OBCmaskCu(:,:) = G%mask2dCu(:,:) ; G%OBCmaskCv(:,:) = G%mask2dCv(:,:)
GV%RZ_to_H = GV%Z_to_H / GV%Rho0 ; GV%H_to_RZ = GV%H_to_Z * GV%Rho0
if (associated(tv%eqn_of_state)) call EOS_init(pf, tv%eqn_of_state, US)
#endif
allocate(GV%g_prime(nk+1))      ! (read without an equation of state for the stratification of the FGNV solve, :1095; from Rlay as MOM_coord_initialization sets it)
GV%g_prime(:) = 0.0 ; GV%g_prime(1) = GV%g_Earth
do m=2,nk ; GV%g_prime(m) = (GV%g_Earth/GV%Rho0) * (GV%Rlay(m) - GV%Rlay(m-1)) ; enddo
call thickness_diffuse_init(Time, G, GV, US, pf, diag, CDp, CS)
call thickness_diffuse(h, uhtr, vhtr, tv, dt, G, GV, US, MEKE, VarMix, CDp, CS, STOCH)

#ifndef REFERENCE_KERNELS
! with GPU_RESIDENT_DYNAMICS the results are on the device until the host asks for them
call mom6hip_mirrors_to_host(mom6hip_shared_context(G, GV))
#endif
open(newunit=u_out, file=trim(f_out), access="stream", form="unformatted", status="replace")
write(u_out) h, uhtr, vhtr, CDp%uhGM, CDp%vhGM
if (opt(6) /= 0) write(u_out) MEKE%GM_src
close(u_out)
call thickness_diffuse_end(CS, CDp)
#ifndef REFERENCE_KERNELS
call mom6hip_mirrors_end()
call mom6hip_shared_context_end()
#endif
write(*,'(a)') "td_driver ok"
end program td_driver
