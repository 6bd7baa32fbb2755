!> Drives the two tracer module shims the way step_MOM_tracer_dyn does (src/core/MOM.F90:1437-1443): an ocean_grid_type filled from a
!! file written by tests/test_fortran_abi.py, tracer_advect_init and tracer_hor_diff_init from a parameter list (KEY = VALUE lines),
!! a tracer registry of ntr arrays, then
!!   advect_tracer(h_end, uhtr, vhtr, OBC, dt, G, GV, US, CS, Reg)
!!   tracer_hordiff(h_end, dt, MEKE, VarMix, visc, G, GV, US, CS, Reg, tv)
!! on plain host arrays; the tracers go to the output file, which the test compares with the oracle bit for bit.
!! Usage: tracer_driver <input file> <output file> <parameter file>
program tracer_driver
use, intrinsic :: iso_c_binding
use MOM_tracer_advect,   only : advect_tracer, tracer_advect_init, tracer_advect_end, tracer_advect_CS
use MOM_tracer_hor_diff, only : tracer_hordiff, tracer_hor_diff_init, tracer_hor_diff_end, tracer_hor_diff_CS
use MOM_tracer_registry, only : tracer_registry_type
use MOM_MEKE_types,      only : MEKE_type
use MOM_EOS,             only : EOS_type
use MOM_diabatic_driver, only : diabatic_CS
use MOM_open_boundary,   only : ocean_OBC_type
use MOM_variables,       only : vertvisc_type
use MOM_diag_mediator,  only : diag_ctrl, time_type
use MOM_domains,        only : MOM_domain_type
use MOM_file_parser,    only : param_file_type, param_set
use MOM_grid,           only : ocean_grid_type
use MOM_lateral_mixing_coeffs, only : VarMix_CS
use MOM_unit_scaling,   only : unit_scale_type
use MOM_variables,      only : thermo_var_ptrs
use MOM_verticalGrid,   only : verticalGrid_type
#ifdef REFERENCE_KERNELS
use MOM_EOS,            only : EOS_init      ! (built with -DREFERENCE_KERNELS -DREF_EOS -DREF_INTERFACE_HEIGHTS -DREF_ALE: the reference's OWN MOM_tracer_advect.F90,
                                             ! MOM_tracer_hor_diff.F90, MOM_neutral_diffusion.F90, MOM_hor_bnd_diffusion.F90, the ALE remapping stack and MOM_EOS)
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
type(tracer_advect_CS), pointer :: ACS => NULL()
type(tracer_hor_diff_CS), pointer :: DCS => NULL()
type(tracer_registry_type), pointer :: Reg => NULL()
type(MEKE_type) :: MEKE
type(EOS_type), target :: EOS
type(diabatic_CS), pointer :: diabatic_CSp => NULL()
type(ocean_OBC_type), pointer :: OBC => NULL()
type(vertvisc_type) :: visc
real, allocatable, target, dimension(:,:,:,:) :: trs
real :: dt_therm
integer :: m, ntr
type(thermo_var_ptrs) :: tv
type(VarMix_CS) :: VarMix
integer(c_int32_t) :: hdr(8), opt(8)
integer :: ni, nj, nk, halo, u_in, u_out, u_par, isd, ied, jsd, jed, ios, eq
real :: scal(7), dt
real, allocatable, target, dimension(:,:,:) :: h, uhtr, vhtr
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
! opt = [ntr, VarMix%use_variable_mixing, VarMix%Resoln_scaled_KhTr, MEKE%Kh allocated, visc%h_ML follows, GV%nk_rho_varies, GV%nkml, -]
read(u_in) opt
ntr = opt(1)

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

allocate(h(isd:ied,jsd:jed,nk), uhtr(isd-1:ied,jsd:jed,nk), vhtr(isd:ied,jsd-1:jed,nk), trs(isd:ied,jsd:jed,nk,ntr))
allocate(MEKE%Kh(isd:ied,jsd:jed), VarMix%L2u(isd-1:ied,jsd:jed), VarMix%L2v(isd:ied,jsd-1:jed), VarMix%SN_u(isd-1:ied,jsd:jed), &
         VarMix%SN_v(isd:ied,jsd-1:jed), VarMix%Res_fn_h(isd:ied,jsd:jed), VarMix%Rd_dx_h(isd:ied,jsd:jed))
read(u_in) dt_therm, MEKE%KhTr_fac
read(u_in) h, uhtr, vhtr, trs
read(u_in) MEKE%Kh, VarMix%L2u, VarMix%L2v, VarMix%SN_u, VarMix%SN_v, VarMix%Res_fn_h, VarMix%Rd_dx_h
if (opt(5) /= 0) then      ! visc%h_ML (NDIFF_INTERIOR_ONLY)
  allocate(visc%h_ML(isd:ied,jsd:jed)) ; read(u_in) visc%h_ML
endif
if (opt(6) /= 0) then      ! a layered run with opt(6) variable-density layers, opt(7) of them mixed layers (DIFFUSE_ML_TO_INTERIOR)
  GV%nk_rho_varies = opt(6) ; GV%nkml = opt(7)
  if (allocated(GV%Rlay)) deallocate(GV%Rlay)
  allocate(GV%Rlay(nk)) ; read(u_in) GV%Rlay, tv%P_Ref
endif
close(u_in)
VarMix%use_variable_mixing = (opt(2) /= 0) ; VarMix%Resoln_scaled_KhTr = (opt(3) /= 0)
if (opt(4) == 0) deallocate(MEKE%Kh)
allocate(Reg) ; Reg%ntr = ntr
do m=1,ntr ; Reg%Tr(m)%t => trs(:,:,:,m) ; enddo
if (ntr >= 2) then ; tv%T => trs(:,:,:,1) ; tv%S => trs(:,:,:,2) ; tv%eqn_of_state => EOS ; endif      ! (read with USE_NEUTRAL_DIFFUSION)

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
GV%RZ_to_H = GV%Z_to_H / GV%Rho0 ; GV%H_to_RZ = GV%H_to_Z * GV%Rho0 ; GV%H_to_Pa = GV%g_Earth * GV%H_to_RZ
call EOS_init(pf, EOS, US)
if (opt(5) /= 0) then      ! NDIFF_INTERIOR_ONLY asks for a boundary layer scheme to be there (MOM_neutral_diffusion.F90:281-285); visc%h_ML is what it reads
  allocate(diabatic_CSp) ; allocate(diabatic_CSp%ePBL)
endif
#endif
call tracer_advect_init(Time, G, US, pf, diag, ACS)
call tracer_hor_diff_init(Time, G, GV, US, pf, diag, EOS, diabatic_CSp, DCS)
call advect_tracer(h, uhtr, vhtr, OBC, dt_therm, G, GV, US, ACS, Reg)
call tracer_hordiff(h, dt_therm, MEKE, VarMix, visc, G, GV, US, DCS, Reg, tv)

#ifndef REFERENCE_KERNELS
! with GPU_RESIDENT_DYNAMICS the results are on the device until the host asks for them
call mom6hip_mirrors_to_host(mom6hip_shared_context(G, GV))
#endif
open(newunit=u_out, file=trim(f_out), access="stream", form="unformatted", status="replace")
write(u_out) trs
close(u_out)
call tracer_hor_diff_end(DCS)
call tracer_advect_end(ACS)
#ifndef REFERENCE_KERNELS
call mom6hip_mirrors_end()
call mom6hip_shared_context_end()
#endif
write(*,'(a)') "tracer_driver ok"
end program tracer_driver
