!> Drives the MOM_mixed_layer_restrat shim the way MOM.F90 does (:2854, :3312, :1335): an ocean_grid_type filled from a file written
!! by tests/test_mixedlayer_restrat.py, mixedlayer_restrat_register_restarts and mixedlayer_restrat_init from a parameter list
!! (KEY = VALUE lines), then ncalls times
!!   mixedlayer_restrat(h, uhtr, vhtr, tv, forces, dt, MLD, h_MLD, bflux, VarMix, G, GV, US, CS)
!! on plain host arrays (the running means of the mixed layer depth live in the control structure between the calls); h, uhtr, vhtr
!! go to the output file, which the test compares with the oracle bit for bit.
!! Usage: mle_driver <input file> <output file> <parameter file>
program mle_driver
use, intrinsic :: iso_c_binding
use MOM_mixed_layer_restrat, only : mixedlayer_restrat, mixedlayer_restrat_init, mixedlayer_restrat_register_restarts, mixedlayer_restrat_CS
use MOM_forcing_type,   only : mech_forcing
use MOM_hor_index,      only : hor_index_type
use MOM_restart,        only : MOM_restart_CS
use MOM_diag_mediator,  only : diag_ctrl, time_type
use MOM_domains,        only : MOM_domain_type
use MOM_file_parser,    only : param_file_type, param_set
use MOM_grid,           only : ocean_grid_type
use MOM_lateral_mixing_coeffs, only : VarMix_CS
use MOM_unit_scaling,   only : unit_scale_type
use MOM_variables,      only : thermo_var_ptrs
use MOM_verticalGrid,   only : verticalGrid_type
#ifdef REFERENCE_KERNELS
use MOM_EOS,            only : EOS_init      ! (built with -DREFERENCE_KERNELS -DREF_EOS: the reference's OWN MOM_mixed_layer_restrat.F90 and MOM_EOS)
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
type(mixedlayer_restrat_CS) :: CS
type(mech_forcing) :: forces
type(hor_index_type) :: HI
type(MOM_restart_CS) :: restart_CS
real, dimension(:,:), pointer :: MLD => NULL(), h_MLD => NULL(), bflux => NULL()
logical :: on
integer :: n
type(thermo_var_ptrs) :: tv
type(VarMix_CS) :: VarMix
integer(c_int32_t) :: hdr(8), opt(8)
integer :: ni, nj, nk, halo, u_in, u_out, u_par, isd, ied, jsd, jed, ios, eq
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
! opt = [GV%nkml, h_MLD associated, VarMix%Rd_dx_h allocated, ncalls, ...]
read(u_in) opt
GV%nkml = opt(1)

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
allocate(tv%T(isd:ied,jsd:jed,nk), tv%S(isd:ied,jsd:jed,nk), forces%ustar(isd:ied,jsd:jed), h_MLD(isd:ied,jsd:jed), &
         VarMix%Rd_dx_h(isd:ied,jsd:jed))
read(u_in) h, tv%T, tv%S, forces%ustar, h_MLD, VarMix%Rd_dx_h
close(u_in)
uhtr = 0.0 ; vhtr = 0.0
allocate(tv%eqn_of_state)
if (opt(2) == 0) then ; deallocate(h_MLD) ; h_MLD => NULL() ; endif
if (opt(3) == 0) deallocate(VarMix%Rd_dx_h)

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

HI%isc = G%isc ; HI%iec = G%iec ; HI%jsc = G%jsc ; HI%jec = G%jec ; HI%isd = isd ; HI%ied = ied ; HI%jsd = jsd ; HI%jed = jed
#ifdef REFERENCE_KERNELS
HI%IsdB = isd-1 ; HI%IedB = ied ; HI%JsdB = jsd-1 ; HI%JedB = jed ; HI%IscB = G%IscB ; HI%IecB = G%IecB ; HI%JscB = G%JscB ; HI%JecB = G%JecB
G%HI = HI
allocate(G%OBCmaskCu(isd-1:ied,jsd:jed), G%OBCmaskCv(isd:ied,jsd-1:jed))      ! no open boundaries: the masks of the faces (MOM_grid.F90)
G%OBCmaskCu(:,:) = G%mask2dCu(:,:) ; G%OBCmaskCv(:,:) = G%mask2dCv(:,:)
GV%RZ_to_H = GV%Z_to_H / GV%Rho0 ; GV%H_to_RZ = GV%H_to_Z * GV%Rho0
call EOS_init(pf, tv%eqn_of_state, US)
#endif
call mixedlayer_restrat_register_restarts(HI, GV, US, pf, CS, restart_CS)
on = mixedlayer_restrat_init(Time, G, GV, US, pf, diag, CS, restart_CS)
if (.not.on) error stop "mle_driver: MIXEDLAYER_RESTRAT is not set"
do n=1,opt(4)
  call mixedlayer_restrat(h, uhtr, vhtr, tv, forces, dt, MLD, h_MLD, bflux, VarMix, G, GV, US, CS)
enddo

#ifndef REFERENCE_KERNELS
! with GPU_RESIDENT_DYNAMICS the results are on the device until the host asks for them
call mom6hip_mirrors_to_host(mom6hip_shared_context(G, GV))
#endif
open(newunit=u_out, file=trim(f_out), access="stream", form="unformatted", status="replace")
write(u_out) h, uhtr, vhtr
close(u_out)
#ifndef REFERENCE_KERNELS
call mom6hip_mirrors_end()
call mom6hip_shared_context_end()
#endif
write(*,'(a,i0)') "mle_driver ok restart_fields=", restart_CS%nfields
end program mle_driver
