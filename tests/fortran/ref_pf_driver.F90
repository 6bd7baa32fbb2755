!> Drives the reference's OWN PressureForce_FV_Bouss (src/core/MOM_PressureForce_FV.F90 with MOM_density_integrals.F90,
!! MOM_PressureForce_Montgomery.F90 for Set_pbce_Bouss, and the whole equation-of-state stack src/equation_of_state/*.F90), compiled in place
!! against the stand-ins of tests/fortran/stubs (-DREF_EOS -DREF_PF -DREF_PF_MONT), on a file written by tests/test_reference_kernels.py:
!! EOS_init and PressureForce_FV_init from a parameter list, then PressureForce_FV_Bouss(h, tv, PFu, PFv, G, GV, US, CS, ALE_CSp, p_atm, pbce, eta)
!! as the split RK2 step calls it (:495).  The outputs go to a file that the test compares with the oracle bit for bit.
!! Usage: ref_pf_driver <input file> <output file> [NAME=VALUE ...]
program ref_pf_driver
use, intrinsic :: iso_c_binding
use MOM_PressureForce_FV, only : PressureForce_FV_Bouss, PressureForce_FV_nonBouss, PressureForce_FV_init, PressureForce_FV_CS
use MOM_ALE,            only : ALE_CS
use MOM_EOS,            only : EOS_init
use MOM_diag_mediator,  only : diag_ctrl, time_type
use MOM_file_parser,    only : param_file_type, param_set
use MOM_grid,           only : ocean_grid_type
use MOM_unit_scaling,   only : unit_scale_type
use MOM_variables,      only : thermo_var_ptrs
use MOM_verticalGrid,   only : verticalGrid_type
implicit none

type(ocean_grid_type), target :: G
type(verticalGrid_type) :: GV
type(unit_scale_type) :: US
type(param_file_type) :: pf
type(time_type), target :: Time
type(diag_ctrl), target :: diag
type(PressureForce_FV_CS) :: PCS
type(thermo_var_ptrs) :: tv
type(ALE_CS), pointer :: ALE_CSp => NULL()
real, pointer, dimension(:,:) :: p_atm => NULL()
real, allocatable, dimension(:,:,:) :: h, PFu, PFv, pbce
real, allocatable, dimension(:,:) :: eta
integer(c_int32_t) :: hdr(8)
integer :: ni, nj, nk, halo, u_in, u_out, isd, ied, jsd, jed, m, i0
real :: scal(7), dt
character(len=512) :: f_in, f_out, f_arg

call get_command_argument(1, f_in) ; call get_command_argument(2, f_out)
open(newunit=u_in, file=trim(f_in), access="stream", form="unformatted", status="old")
read(u_in) hdr
ni = hdr(1) ; nj = hdr(2) ; nk = hdr(3) ; halo = hdr(4)
isd = 1 ; ied = ni + 2*halo ; jsd = 1 ; jed = nj + 2*halo
G%isd = isd ; G%ied = ied ; G%jsd = jsd ; G%jed = jed ; G%IsdB = isd-1 ; G%IedB = ied ; G%JsdB = jsd-1 ; G%JedB = jed
G%isc = isd+halo ; G%iec = ied-halo ; G%jsc = jsd+halo ; G%jec = jed-halo
G%IscB = G%isc-1 ; G%IecB = G%iec ; G%JscB = G%jsc-1 ; G%JecB = G%jec ; G%ke = nk ; GV%ke = nk
G%HI%isd = isd ; G%HI%ied = ied ; G%HI%jsd = jsd ; G%HI%jed = jed ; G%HI%IsdB = isd-1 ; G%HI%IedB = ied ; G%HI%JsdB = jsd-1 ; G%HI%JedB = jed
G%HI%isc = G%isc ; G%HI%iec = G%iec ; G%HI%jsc = G%jsc ; G%HI%jec = G%jec
G%HI%IscB = G%IscB ; G%HI%IecB = G%IecB ; G%HI%JscB = G%JscB ; G%HI%JecB = G%JecB
allocate(G%Domain)
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
allocate(h(isd:ied,jsd:jed,nk), tv%T(isd:ied,jsd:jed,nk), tv%S(isd:ied,jsd:jed,nk))
read(u_in) h, tv%T, tv%S
if (hdr(8) /= 0) then      ! a surface pressure
  allocate(p_atm(isd:ied,jsd:jed)) ; read(u_in) p_atm
endif
close(u_in)
allocate(PFu(isd-1:ied,jsd:jed,nk), PFv(isd:ied,jsd-1:jed,nk), pbce(isd:ied,jsd:jed,nk), eta(isd:ied,jsd:jed))
PFu = 0.0 ; PFv = 0.0 ; pbce = 0.0 ; eta = 0.0
allocate(GV%g_prime(nk+1)) ; GV%g_prime(:) = 0.0 ; GV%g_prime(1) = GV%g_Earth

call param_set(pf, "USE_REGRIDDING", "True") ; call param_set(pf, "EQN_OF_STATE", "WRIGHT")
do m = 3, command_argument_count()      ! further NAME=VALUE pairs of the parameter file (they replace the ones above)
  call get_command_argument(m, f_arg)
  i0 = index(f_arg, "=")
  if (i0 > 1) call param_set(pf, f_arg(1:i0-1), trim(f_arg(i0+1:)))
  if (trim(f_arg) == "BOUSSINESQ=False") then      ! thicknesses in kg m-2 (H_TO_KG_M2 = 1): verticalGridInit's non-Boussinesq factors
    GV%Boussinesq = .false. ; GV%H_to_RZ = 1.0 ; GV%RZ_to_H = 1.0 ; GV%H_to_Z = 1.0 / GV%Rho0 ; GV%Z_to_H = GV%Rho0
  endif
enddo
allocate(tv%eqn_of_state)
call EOS_init(pf, tv%eqn_of_state, US)
call PressureForce_FV_init(Time, G, GV, US, pf, diag, PCS)
allocate(ALE_CSp)
if (GV%Boussinesq) then
  call PressureForce_FV_Bouss(h, tv, PFu, PFv, G, GV, US, PCS, ALE_CSp, p_atm, pbce, eta)
else
  call PressureForce_FV_nonBouss(h, tv, PFu, PFv, G, GV, US, PCS, ALE_CSp, p_atm, pbce, eta)
endif

open(newunit=u_out, file=trim(f_out), access="stream", form="unformatted", status="replace")
write(u_out) PFu, PFv, pbce, eta
close(u_out)
write(*,'(a)') "ref_pf_driver ok"
end program ref_pf_driver
