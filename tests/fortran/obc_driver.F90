!> Drives continuity_PPM of the module shim with an associated OBC the way step_MOM_dyn_split_RK2 calls it
!! (MOM_dynamics_split_RK2.F90:757): an ocean_grid_type and an ocean_OBC_type -- its segments with their hor_index ranges and
!! flags, segnum_u / segnum_v, the external transports and velocities of the specified segments -- filled from a file written by
!! tests/test_continuity_obc.py, then
!!   continuity(u, v, h, hp, uh, vh, dt, G, GV, US, CS, OBC, pbv, uhbt, vhbt, visc_rem_u, visc_rem_v, u_cor, v_cor, BT_cont)
!!   CorAdCalc(u, v, h, uh, vh, CAu, CAv, OBC, AD, G, GV, US, CS, pbv)                         (MOM_dynamics_split_RK2.F90:869)
!!   set_visc_init(..., OBC) ; set_viscous_BBL(u, v, h, tv, visc, G, GV, US, CS, pbv)     (MOM.F90:1205; layer mode, as .testing/tc3)
!!   hor_visc_init ; horizontal_viscosity(u, v, h, diffu, diffv, MEKE, VarMix, G, GV, US, CS, tv, dt, OBC=OBC)          (:860; tc3's viscosities)
!!   vertvisc_coef(u, v, h, dz, forces, visc, tv, dt, G, GV, US, CS, OBC, VarMix) ; vertvisc(u, v, h, forces, visc, dt, OBC, ...)   (:717-731)
!! on plain host arrays; the results go to the output file, which the test compares with the oracle bit for bit.
!! Usage: obc_driver <input file> <output file>
program obc_driver
use, intrinsic :: iso_c_binding
use MOM_continuity_PPM, only : continuity_PPM, continuity_PPM_init, continuity_PPM_CS
use MOM_CoriolisAdv,    only : CorAdCalc, CoriolisAdv_init, CoriolisAdv_end, CoriolisAdv_CS
use MOM_set_visc,       only : set_visc_CS, set_visc_init, set_viscous_BBL, set_visc_end
use MOM_restart,        only : MOM_restart_CS
use MOM_hor_visc,       only : hor_visc_CS, hor_visc_init, horizontal_viscosity, hor_visc_end
use MOM_MEKE_types,     only : MEKE_type
use MOM_vert_friction,  only : vertvisc_CS, vertvisc_init, vertvisc_coef, vertvisc, vertvisc_end
use MOM_variables,      only : accel_diag_ptrs, cont_diag_ptrs, vertvisc_type, thermo_var_ptrs, ocean_internal_state
use MOM_forcing_type,   only : mech_forcing
use MOM_get_input,      only : directories
use MOM_lateral_mixing_coeffs, only : VarMix_CS
use MOM_diag_mediator,  only : diag_ctrl, time_type
use MOM_domains,        only : MOM_domain_type
use MOM_file_parser,    only : param_file_type, param_set
use MOM_grid,           only : ocean_grid_type
use MOM_open_boundary,  only : ocean_OBC_type, OBC_segment_type
use MOM_tracer_advect,  only : tracer_advect_CS, tracer_advect_init, advect_tracer, tracer_advect_end
use MOM_tracer_registry, only : tracer_registry_type
use MOM_unit_scaling,   only : unit_scale_type
use MOM_variables,      only : BT_cont_type, porous_barrier_type, alloc_BT_cont_type
use MOM_verticalGrid,   only : verticalGrid_type
use mom6hip_MOM_glue,   only : mom6hip_shared_context_end, update_segment_tracer_reservoirs_hip
implicit none

type(ocean_grid_type), target :: G
type(verticalGrid_type) :: GV
type(unit_scale_type) :: US
type(param_file_type) :: pf
type(time_type), target :: Time
type(diag_ctrl), target :: diag
type(continuity_PPM_CS) :: CS
type(CoriolisAdv_CS) :: CCS
type(accel_diag_ptrs), target :: AD
integer(c_int32_t) :: vflags(4)
real, allocatable, dimension(:,:,:) :: CAu, CAv
type(ocean_OBC_type), pointer :: OBC => NULL()
type(porous_barrier_type) :: pbv
type(BT_cont_type), pointer :: BT => NULL()
integer(c_int32_t) :: hdr(8), oflags(8), sflags(14)
integer(c_int32_t), allocatable :: su(:,:), sv(:,:)
integer :: ni, nj, nk, halo, u_in, u_out, isd, ied, jsd, jed, n, nseg
real :: scal(7), dt
real, allocatable, dimension(:,:,:) :: u, v, h, hp, uh, vh, vru, vrv, u_cor, v_cor
real, allocatable, dimension(:,:) :: uhbt, vhbt
character(len=512) :: f_in, f_out
type(ocean_internal_state), target :: MIS
type(directories) :: dirs
type(vertvisc_CS), pointer :: VV => NULL()
type(vertvisc_type) :: visc
type(set_visc_CS) :: SVC
type(hor_visc_CS) :: HV
type(MEKE_type) :: MEKE
integer(c_int32_t) :: stflags(4)
real, allocatable, dimension(:,:,:) :: diffu, diffv
type(MOM_restart_CS) :: restart_CS
type(thermo_var_ptrs) :: tv
type(mech_forcing) :: forces
type(cont_diag_ptrs) :: CDp
type(VarMix_CS) :: VarMix
integer, target :: ntrunc
real, allocatable, dimension(:,:,:) :: dz, u1, v1
type(tracer_advect_CS), pointer :: TA => NULL()
type(tracer_registry_type), pointer :: Reg => NULL()
real, allocatable, dimension(:,:,:) :: uhtr, vhtr
integer :: i, j, k, i0, i1, j0, j1

call get_command_argument(1, f_in) ; call get_command_argument(2, f_out)
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

allocate(u(isd-1:ied,jsd:jed,nk), v(isd:ied,jsd-1:jed,nk), h(isd:ied,jsd:jed,nk), uhbt(isd-1:ied,jsd:jed), vhbt(isd:ied,jsd-1:jed), &
         vru(isd-1:ied,jsd:jed,nk), vrv(isd:ied,jsd-1:jed,nk))
read(u_in) u, v, h, uhbt, vhbt, vru, vrv

! ocean_OBC_type as open_boundary_config leaves it: [number_of_segments, OBC_pe, open_u, open_v, specified_u, specified_v, Flather_u, Flather_v],
! per segment [direction, open, specified, on_pe, is_E_or_W, is_N_or_S, IsdB, IedB, JsdB, JedB, isd, ied, jsd, jed], segnum_u, segnum_v,
! and for the specified segments on the PE normal_trans, normal_vel on the segment's own index ranges
allocate(OBC)
read(u_in) oflags
nseg = oflags(1)
OBC%number_of_segments = nseg ; OBC%OBC_pe = (oflags(2) /= 0)
OBC%open_u_BCs_exist_globally = (oflags(3) /= 0) ; OBC%open_v_BCs_exist_globally = (oflags(4) /= 0)
OBC%specified_u_BCs_exist_globally = (oflags(5) /= 0) ; OBC%specified_v_BCs_exist_globally = (oflags(6) /= 0)
OBC%Flather_u_BCs_exist_globally = (oflags(7) /= 0) ; OBC%Flather_v_BCs_exist_globally = (oflags(8) /= 0)
allocate(OBC%segment(nseg))
do n=1,nseg
  read(u_in) sflags
  OBC%segment(n)%direction = sflags(1) ; OBC%segment(n)%open = (sflags(2) /= 0) ; OBC%segment(n)%specified = (sflags(3) /= 0)
  OBC%segment(n)%on_pe = (sflags(4) /= 0) ; OBC%segment(n)%is_E_or_W = (sflags(5) /= 0) ; OBC%segment(n)%is_N_or_S = (sflags(6) /= 0)
  OBC%segment(n)%HI%IsdB = sflags(7) ; OBC%segment(n)%HI%IedB = sflags(8) ; OBC%segment(n)%HI%JsdB = sflags(9) ; OBC%segment(n)%HI%JedB = sflags(10)
  OBC%segment(n)%HI%isd = sflags(11) ; OBC%segment(n)%HI%ied = sflags(12) ; OBC%segment(n)%HI%jsd = sflags(13) ; OBC%segment(n)%HI%jed = sflags(14)
enddo
allocate(su(isd-1:ied,jsd:jed), sv(isd:ied,jsd-1:jed), OBC%segnum_u(isd-1:ied,jsd:jed), OBC%segnum_v(isd:ied,jsd-1:jed))
read(u_in) su, sv
OBC%segnum_u(:,:) = su(:,:) ; OBC%segnum_v(:,:) = sv(:,:)
do n=1,nseg ; if (OBC%segment(n)%specified .and. OBC%segment(n)%on_pe) then
  if (OBC%segment(n)%is_E_or_W) then
    allocate(OBC%segment(n)%normal_trans(OBC%segment(n)%HI%IsdB:OBC%segment(n)%HI%IedB, OBC%segment(n)%HI%jsd:OBC%segment(n)%HI%jed, nk))
    allocate(OBC%segment(n)%normal_vel(OBC%segment(n)%HI%IsdB:OBC%segment(n)%HI%IedB, OBC%segment(n)%HI%jsd:OBC%segment(n)%HI%jed, nk))
  else
    allocate(OBC%segment(n)%normal_trans(OBC%segment(n)%HI%isd:OBC%segment(n)%HI%ied, OBC%segment(n)%HI%JsdB:OBC%segment(n)%HI%JedB, nk))
    allocate(OBC%segment(n)%normal_vel(OBC%segment(n)%HI%isd:OBC%segment(n)%HI%ied, OBC%segment(n)%HI%JsdB:OBC%segment(n)%HI%JedB, nk))
  endif
  read(u_in) OBC%segment(n)%normal_trans, OBC%segment(n)%normal_vel
endif ; enddo
! OBC_ZERO_VORTICITY, OBC_FREESLIP_VORTICITY, OBC_COMPUTED_VORTICITY, OBC_SPECIFIED_VORTICITY, then tangential_vel and tangential_grad of every
! segment on the PE (IsdB:IedB, JsdB:JedB, nk)
read(u_in) vflags
OBC%zero_vorticity = (vflags(1) /= 0) ; OBC%freeslip_vorticity = (vflags(2) /= 0)
OBC%computed_vorticity = (vflags(3) /= 0) ; OBC%specified_vorticity = (vflags(4) /= 0)
do n=1,nseg ; if (OBC%segment(n)%on_pe) then
  allocate(OBC%segment(n)%tangential_vel(OBC%segment(n)%HI%IsdB:OBC%segment(n)%HI%IedB, OBC%segment(n)%HI%JsdB:OBC%segment(n)%HI%JedB, nk))
  allocate(OBC%segment(n)%tangential_grad(OBC%segment(n)%HI%IsdB:OBC%segment(n)%HI%IedB, OBC%segment(n)%HI%JsdB:OBC%segment(n)%HI%JedB, nk))
  read(u_in) OBC%segment(n)%tangential_vel, OBC%segment(n)%tangential_grad
endif ; enddo
! OBC_ZERO_STRAIN, OBC_FREESLIP_STRAIN, OBC_COMPUTED_STRAIN, OBC_ZERO_BIHARMONIC
read(u_in) stflags
OBC%zero_strain = (stflags(1) /= 0) ; OBC%freeslip_strain = (stflags(2) /= 0)
OBC%computed_strain = (stflags(3) /= 0) ; OBC%zero_biharmonic = (stflags(4) /= 0)
close(u_in)

allocate(hp(isd:ied,jsd:jed,nk), uh(isd-1:ied,jsd:jed,nk), vh(isd:ied,jsd-1:jed,nk), u_cor(isd-1:ied,jsd:jed,nk), v_cor(isd:ied,jsd-1:jed,nk))
hp = h ; uh = 0.0 ; vh = 0.0 ; u_cor = 0.0 ; v_cor = 0.0
call param_set(pf, "REENTRANT_X", merge("True ", "False", hdr(5) /= 0))
call param_set(pf, "REENTRANT_Y", merge("True ", "False", hdr(6) /= 0))
call continuity_PPM_init(Time, G, GV, US, pf, diag, CS)
call alloc_BT_cont_type(BT, isd, ied, jsd, jed, nk, alloc_faces=.true.)

call continuity_PPM(u, v, h, hp, uh, vh, dt, G, GV, US, CS, OBC, pbv, uhbt, vhbt, vru, vrv, u_cor, v_cor, BT_cont=BT)
call param_set(pf, "BOUND_CORIOLIS", "True")
call CoriolisAdv_init(Time, G, GV, US, pf, diag, AD, CCS)
allocate(CAu(isd-1:ied,jsd:jed,nk), CAv(isd:ied,jsd-1:jed,nk)) ; CAu = 0.0 ; CAv = 0.0
call CorAdCalc(u, v, h, uh, vh, CAu, CAv, OBC, AD, G, GV, US, CCS, pbv)

! the bottom boundary layer in layer mode (ENABLE_THERMODYNAMICS = False: GV%Rlay), then the vertical viscosity of the predictor
! (:717-731) with the wind stress as a plain function of the grid
allocate(GV%Rlay(nk))
do n=1,nk ; GV%Rlay(n) = 1025.0 + 0.5*real(n-1) ; enddo
call param_set(pf, "ENABLE_THERMODYNAMICS", "False") ; call param_set(pf, "HBBL", "10.0") ; call param_set(pf, "KV", "1.0e-4")
call param_set(pf, "DRAG_BG_VEL", "0.05") ; call param_set(pf, "BBL_THICK_MIN", "0.1") ; call param_set(pf, "CDRAG", "0.002")
call set_visc_init(Time, G, GV, US, pf, diag, visc, SVC, restart_CS, OBC)
call set_viscous_BBL(u, v, h, tv, visc, G, GV, US, SVC, pbv)
allocate(forces%taux(isd-1:ied,jsd:jed), forces%tauy(isd:ied,jsd-1:jed))
forces%taux(:,:) = 0.05*G%mask2dCu(:,:) ; forces%tauy(:,:) = -0.02*G%mask2dCv(:,:)
allocate(dz(isd:ied,jsd:jed,nk), u1(isd-1:ied,jsd:jed,nk), v1(isd:ied,jsd-1:jed,nk))
dz(:,:,:) = GV%H_to_Z * h(:,:,:) ; u1 = u ; v1 = v
call param_set(pf, "DT", "900.0") ; call param_set(pf, "HMIX_FIXED", "20.0")
call vertvisc_init(MIS, Time, G, GV, US, pf, diag, AD, dirs, ntrunc, VV)
call vertvisc_coef(u1, v1, h, dz, forces, visc, tv, dt, G, GV, US, VV, OBC, VarMix)
call vertvisc(u1, v1, h, forces, visc, dt, OBC, AD, CDp, G, GV, US, VV)

! the horizontal viscosity of the corrector (:860) with tc3's coefficients
call param_set(pf, "LAPLACIAN", "True") ; call param_set(pf, "KH", "25.0") ; call param_set(pf, "KH_VEL_SCALE", "0.003")
call param_set(pf, "SMAGORINSKY_KH", "True") ; call param_set(pf, "SMAG_LAP_CONST", "0.15") ; call param_set(pf, "AH_VEL_SCALE", "0.003")
call param_set(pf, "SMAGORINSKY_AH", "True") ; call param_set(pf, "SMAG_BI_CONST", "0.06")
call hor_visc_init(Time, G, GV, US, pf, diag, HV)
allocate(diffu(isd-1:ied,jsd:jed,nk), diffv(isd:ied,jsd-1:jed,nk))
call horizontal_viscosity(u, v, h, diffu, diffv, MEKE, VarMix, G, GV, US, HV, tv, dt, OBC=OBC)

open(newunit=u_out, file=trim(f_out), access="stream", form="unformatted", status="replace")
write(u_out) hp, uh, vh, u_cor, v_cor
write(u_out) BT%FA_u_W0, BT%FA_u_WW, BT%FA_u_E0, BT%FA_u_EE, BT%uBT_WW, BT%uBT_EE
write(u_out) BT%FA_v_S0, BT%FA_v_SS, BT%FA_v_N0, BT%FA_v_NN, BT%vBT_SS, BT%vBT_NN, BT%h_u, BT%h_v
write(u_out) CAu, CAv
write(u_out) visc%bbl_thick_u, visc%bbl_thick_v, visc%Kv_bbl_u, visc%Kv_bbl_v, u1, v1, diffu, diffv

! advect_tracer (PPM:H3) of two tracers with the transports of the continuity step, every segment on the PE with a registry: tracer 1 with a
! reservoir, tracer 2 with an inflow concentration (the values: exact quotients of small integers, the same in the test)
allocate(Reg) ; Reg%ntr = 2
allocate(Reg%Tr(1)%t(isd:ied,jsd:jed,nk), Reg%Tr(2)%t(isd:ied,jsd:jed,nk), uhtr(isd-1:ied,jsd:jed,nk), vhtr(isd:ied,jsd-1:jed,nk))
do k=1,nk ; do j=jsd,jed ; do i=isd,ied
  Reg%Tr(1)%t(i,j,k) = 1.0 + real(mod(3*i + 5*j + 7*k, 11)) / 11.0
  Reg%Tr(2)%t(i,j,k) = real(mod(2*i + 3*j + k, 7)) / 7.0
enddo ; enddo ; enddo
uhtr(:,:,:) = dt * uh(:,:,:) ; vhtr(:,:,:) = dt * vh(:,:,:)
do n=1,nseg ; if (OBC%segment(n)%on_pe) then
  allocate(OBC%segment(n)%tr_Reg) ; OBC%segment(n)%tr_Reg%ntseg = 2
  if (OBC%segment(n)%is_E_or_W) then
    i0 = OBC%segment(n)%HI%IsdB ; i1 = OBC%segment(n)%HI%IedB ; j0 = OBC%segment(n)%HI%jsd ; j1 = OBC%segment(n)%HI%jed
  else
    i0 = OBC%segment(n)%HI%isd ; i1 = OBC%segment(n)%HI%ied ; j0 = OBC%segment(n)%HI%JsdB ; j1 = OBC%segment(n)%HI%JedB
  endif
  allocate(OBC%segment(n)%tr_Reg%Tr(1)%tres(i0:i1,j0:j1,nk))
  do k=1,nk ; do j=j0,j1 ; do i=i0,i1
    OBC%segment(n)%tr_Reg%Tr(1)%tres(i,j,k) = 5.0 + real(mod(i + 2*j + 3*k, 13)) / 13.0
  enddo ; enddo ; enddo
  OBC%segment(n)%tr_Reg%Tr(1)%ntr_index = 1
  OBC%segment(n)%tr_Reg%Tr(2)%ntr_index = 2 ; OBC%segment(n)%tr_Reg%Tr(2)%OBC_inflow_conc = 0.25 + 0.125 * n
endif ; enddo
call param_set(pf, "TRACER_ADVECTION_SCHEME", "PPM:H3")
call tracer_advect_init(Time, G, US, pf, diag, TA)
call advect_tracer(hp, uhtr, vhtr, OBC, dt, G, GV, US, TA, Reg)
write(u_out) Reg%Tr(1)%t, Reg%Tr(2)%t
call tracer_advect_end(TA)

! update_segment_tracer_reservoirs (MOM.F90:1447) with the same transports: the reservoir of tracer 1 on every segment, external values
! and inverse length scales as the test states them
do n=1,nseg ; if (OBC%segment(n)%on_pe) then
  OBC%segment(n)%Tr_InvLscale_in = 1.0e-4 ; OBC%segment(n)%Tr_InvLscale_out = merge(0.0, 3.0e-5, mod(n, 2) == 0)
  i0 = lbound(OBC%segment(n)%tr_Reg%Tr(1)%tres, 1) ; i1 = ubound(OBC%segment(n)%tr_Reg%Tr(1)%tres, 1)
  j0 = lbound(OBC%segment(n)%tr_Reg%Tr(1)%tres, 2) ; j1 = ubound(OBC%segment(n)%tr_Reg%Tr(1)%tres, 2)
  allocate(OBC%segment(n)%tr_Reg%Tr(1)%t(i0:i1,j0:j1,nk))
  do k=1,nk ; do j=j0,j1 ; do i=i0,i1
    OBC%segment(n)%tr_Reg%Tr(1)%t(i,j,k) = 7.0 + real(mod(2*i + j + k, 5)) / 5.0
  enddo ; enddo ; enddo
endif ; enddo
call update_segment_tracer_reservoirs_hip(G, GV, uhtr, vhtr, hp, OBC, dt, Reg)
do n=1,nseg ; if (OBC%segment(n)%on_pe) write(u_out) OBC%segment(n)%tr_Reg%Tr(1)%tres ; enddo
close(u_out)
call hor_visc_end(HV) ; call vertvisc_end(VV) ; call set_visc_end(visc, SVC)
call CoriolisAdv_end(CCS)
call mom6hip_shared_context_end()
write(*,'(a)') "obc_driver ok"
end program obc_driver
