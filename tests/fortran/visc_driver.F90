!> Drives the viscosity shims (MOM_set_visc, MOM_vert_friction, MOM_hor_visc) the way MOM.F90 and the split RK2 step call the
!! reference modules: set_visc_init / set_viscous_BBL (MOM.F90:1205), vertvisc_init / vertvisc_coef / vertvisc /
!! vertvisc_remnant (RK2 :717-744), hor_visc_init / horizontal_viscosity (RK2 :860), on plain host arrays, parameters by name.
!! tests/test_fortran_abi.py writes the input file and compares the output with the oracle bit for bit.
!! Usage: visc_driver <input file> <output file>
!! Built with -DREFERENCE_KERNELS (tests/test_reference_kernels.py) the same program drives the reference's OWN MOM_vert_friction.F90 and
!! MOM_hor_visc.F90, compiled in place against the stand-ins of tests/fortran/stubs: MOM_set_viscosity.F90 is not part of that build (its
!! imports reach the shear-mixing and CVMix modules), so the bottom boundary layer's viscosities and thicknesses are read from the input
!! file (the oracle's) and set_viscous_BBL / set_viscous_ML are not called -- unless -DREF_SET_VISC (and -DREF_EOS) is given as well: then
!! the reference's MOM_set_viscosity.F90 is part of the build (with further stand-ins, mom6_stubs_setvisc.F90) and the whole sequence runs.
program visc_driver
use, intrinsic :: iso_c_binding
#if !defined(REFERENCE_KERNELS) || defined(REF_SET_VISC)
use MOM_set_visc,      only : set_visc_CS, set_visc_init, set_viscous_BBL, set_viscous_ML, set_visc_end
#endif
#ifdef REF_EOS
use MOM_EOS,           only : EOS_init
#endif
use MOM_vert_friction, only : vertvisc_CS, vertvisc_init, vertvisc_coef, vertvisc, vertvisc_remnant, vertvisc_end
use MOM_hor_visc,      only : hor_visc_CS, hor_visc_init, horizontal_viscosity, hor_visc_end, hor_visc_vel_stencil
use MOM_diag_mediator, only : diag_ctrl, time_type
use MOM_domains,       only : MOM_domain_type
use MOM_file_parser,   only : param_file_type, param_set
use MOM_forcing_type,  only : mech_forcing
use MOM_grid,          only : ocean_grid_type
use MOM_get_input,     only : directories
use MOM_lateral_mixing_coeffs, only : VarMix_CS
use MOM_MEKE_types,    only : MEKE_type
use MOM_open_boundary, only : ocean_OBC_type
use MOM_restart,       only : MOM_restart_CS
use MOM_unit_scaling,  only : unit_scale_type
use MOM_variables,     only : vertvisc_type, thermo_var_ptrs, porous_barrier_type, accel_diag_ptrs, cont_diag_ptrs, ocean_internal_state
use MOM_verticalGrid,  only : verticalGrid_type
#ifndef REFERENCE_KERNELS
use mom6hip_MOM_glue,  only : mom6hip_shared_context_end
#endif
implicit none

type(ocean_grid_type), target :: G
type(verticalGrid_type) :: GV
type(unit_scale_type) :: US
type(param_file_type) :: pf
type(time_type), target :: Time
type(diag_ctrl), target :: diag
type(MOM_restart_CS) :: restart_CS
type(ocean_internal_state), target :: MIS
type(directories) :: dirs
#if !defined(REFERENCE_KERNELS) || defined(REF_SET_VISC)
type(set_visc_CS) :: SV
#endif
type(vertvisc_CS), pointer :: VV => NULL()
type(hor_visc_CS) :: HV
type(vertvisc_type) :: visc
type(thermo_var_ptrs) :: tv
type(mech_forcing) :: forces
type(porous_barrier_type) :: pbv
type(accel_diag_ptrs) :: ADp
type(cont_diag_ptrs) :: CDp
type(MEKE_type) :: MEKE
type(VarMix_CS) :: VarMix
type(ocean_OBC_type), pointer :: OBC => NULL()
integer(c_int32_t) :: hdr(8)
integer, target :: ntrunc
integer :: ni, nj, nk, halo, u_in, u_out, isd, ied, jsd, jed
real :: scal(7), dt
real, allocatable, dimension(:,:,:) :: u, v, h, dz, u1, v1, vru, vrv, diffu, diffv
real, allocatable, dimension(:,:) :: tbx, tby
character(len=512) :: f_in, f_out, f_arg
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
allocate(u(isd-1:ied,jsd:jed,nk), v(isd:ied,jsd-1:jed,nk), h(isd:ied,jsd:jed,nk), tv%T(isd:ied,jsd:jed,nk), tv%S(isd:ied,jsd:jed,nk), &
         forces%taux(isd-1:ied,jsd:jed), forces%tauy(isd:ied,jsd-1:jed))
read(u_in) u, v, h, tv%T, tv%S, forces%taux, forces%tauy
#ifdef REFERENCE_KERNELS
#ifndef REF_SET_VISC
allocate(visc%bbl_thick_u(isd-1:ied,jsd:jed), visc%bbl_thick_v(isd:ied,jsd-1:jed), visc%Kv_bbl_u(isd-1:ied,jsd:jed), &
         visc%Kv_bbl_v(isd:ied,jsd-1:jed))
read(u_in) visc%bbl_thick_u, visc%bbl_thick_v, visc%Kv_bbl_u, visc%Kv_bbl_v
#endif
allocate(forces%ustar(isd:ied,jsd:jed), source=0.0)      ! vertvisc_coef asks find_ustar for it whether or not a surface boundary layer reads it (:1307)
G%HI%isd = isd ; G%HI%ied = ied ; G%HI%jsd = jsd ; G%HI%jed = jed ; G%HI%IsdB = isd-1 ; G%HI%IedB = ied ; G%HI%JsdB = jsd-1 ; G%HI%JedB = jed
G%HI%isc = G%isc ; G%HI%iec = G%iec ; G%HI%jsc = G%jsc ; G%HI%jec = G%jec
G%HI%IscB = G%IscB ; G%HI%IecB = G%IecB ; G%HI%JscB = G%JscB ; G%HI%JecB = G%JecB
#endif
close(u_in)
allocate(dz(isd:ied,jsd:jed,nk), u1(isd-1:ied,jsd:jed,nk), v1(isd:ied,jsd-1:jed,nk), vru(isd-1:ied,jsd:jed,nk), vrv(isd:ied,jsd-1:jed,nk), &
         diffu(isd-1:ied,jsd:jed,nk), diffv(isd:ied,jsd-1:jed,nk), tbx(isd-1:ied,jsd:jed), tby(isd:ied,jsd-1:jed))
dz(:,:,:) = GV%H_to_Z * h(:,:,:)      ! thickness_to_dz in Boussinesq mode
u1 = u ; v1 = v ; vru = 0.0 ; vrv = 0.0 ; tbx = 0.0 ; tby = 0.0

call param_set(pf, "REENTRANT_X", merge("True ", "False", hdr(5) /= 0))
call param_set(pf, "REENTRANT_Y", merge("True ", "False", hdr(6) /= 0))
call param_set(pf, "HBBL", "10.0") ; call param_set(pf, "KV", "1.0e-4") ; call param_set(pf, "DT", "900.0")
call param_set(pf, "HMIX_FIXED", "20.0") ; call param_set(pf, "KV_ML_INVZ2", "1.0e-2")
call param_set(pf, "SMAGORINSKY_AH", "True") ; call param_set(pf, "SMAG_BI_CONST", "0.06") ; call param_set(pf, "AH_VEL_SCALE", "0.01")
do m = 3, command_argument_count()      ! further NAME=VALUE pairs of the parameter file (they replace the ones above)
  call get_command_argument(m, f_arg)
  if (index(f_arg, "=") > 1) call param_set(pf, f_arg(1:index(f_arg, "=")-1), trim(f_arg(index(f_arg, "=")+1:)))
enddo

#if !defined(REFERENCE_KERNELS) || defined(REF_SET_VISC)
#ifdef REF_EOS
if (hdr(8) /= 0) then      ! an equation of state for BBL_USE_EOS (the reference's own MOM_EOS; EQN_OF_STATE from the parameter list)
  allocate(tv%eqn_of_state) ; call EOS_init(pf, tv%eqn_of_state, US)
endif
#endif
call set_visc_init(Time, G, GV, US, pf, diag, visc, SV, restart_CS, OBC)
call set_viscous_BBL(u, v, h, tv, visc, G, GV, US, SV, pbv)
call set_viscous_ML(u, v, h, tv, forces, visc, dt, G, GV, US, SV)
#endif
call vertvisc_init(MIS, Time, G, GV, US, pf, diag, ADp, dirs, ntrunc, VV)
call vertvisc_coef(u1, v1, h, dz, forces, visc, tv, dt, G, GV, US, VV, OBC, VarMix)
call vertvisc(u1, v1, h, forces, visc, dt, OBC, ADp, CDp, G, GV, US, VV, taux_bot=tbx, tauy_bot=tby)
call vertvisc_remnant(visc, vru, vrv, dt, G, GV, US, VV)
call hor_visc_init(Time, G, GV, US, pf, diag, HV)
if (hor_visc_vel_stencil(HV) /= 2) error stop "visc_driver: unexpected stencil"
call horizontal_viscosity(u, v, h, diffu, diffv, MEKE, VarMix, G, GV, US, HV, tv, dt)

open(newunit=u_out, file=trim(f_out), access="stream", form="unformatted", status="replace")
write(u_out) visc%bbl_thick_u, visc%bbl_thick_v, visc%Kv_bbl_u, visc%Kv_bbl_v, u1, v1, vru, vrv, tbx, tby, diffu, diffv
close(u_out)
call hor_visc_end(HV) ; call vertvisc_end(VV)
#if !defined(REFERENCE_KERNELS) || defined(REF_SET_VISC)
call set_visc_end(visc, SV)
#endif
#ifndef REFERENCE_KERNELS
call mom6hip_shared_context_end()
#endif
write(*,'(a,i0)') "visc_driver ok ntrunc=", ntrunc
end program visc_driver
