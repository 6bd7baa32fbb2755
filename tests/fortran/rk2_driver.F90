!> The device-resident integration path of INTEGRATION.md section 2(b), from Fortran and with nothing but
!! mom6hip_c_api: a context from a mom6hip_grid_t, every prognostic and control-structure array allocated in HBM with
!! mom6hip_malloc, the control structures of the reference (continuity_PPM_CS, CoriolisAdv_CS, PressureForce_FV_CS, EOS,
!! barotropic_CS, BT_cont_type, MOM_dyn_split_RK2_CS) as bind(c) types, initialize_dyn_split_RK2's state
!! (mom6hip_dyn_split_rk2_init) and two calls of step_MOM_dyn_split_RK2 (mom6hip_step_dyn_split_rk2, one library call per
!! baroclinic step).  The state is copied to the host only at the end; tests/test_fortran_abi.py compares it with the
!! oracle bit for bit.   Usage: rk2_driver <input file> <output file>
program rk2_driver
use, intrinsic :: iso_c_binding
use mom6hip_c_api
implicit none

type(mom6hip_grid_t) :: cg
type(mom6hip_continuity_cs_t), target :: ccs
type(mom6hip_coriolisadv_cs_t), target :: cor
type(mom6hip_pressureforce_cs_t), target :: pcs
type(mom6hip_eos_t), target :: eos
type(mom6hip_barotropic_cs_t), target :: bcs
type(mom6hip_bt_cont_t), target :: btc
type(mom6hip_dyn_split_rk2_cs_t) :: cs
type(c_ptr) :: ctx, d_u, d_v, d_h, d_T, d_S, d_taux, d_tauy, d_uh, d_vh, d_uhtr, d_vhtr, d_eta_av
integer(c_int32_t) :: hdr(8)
integer :: ni, nj, nk, halo, isd, ied, jsd, jed, u_in, u_out, rc, n
integer(c_int64_t) :: nh2, nu2, nv2, nq2, nh3, nu3, nv3
real(c_double) :: scal(7), dt, maxdepth
real(c_double), allocatable, target :: mT(:,:,:), mU(:,:,:), mV(:,:,:), mQ(:,:,:)      ! the 8 metrics of each staggering
real(c_double), allocatable, target :: u(:,:,:), v(:,:,:), h(:,:,:), T(:,:,:), S(:,:,:), taux(:,:), tauy(:,:), eta_av(:,:), &
                                       uhtr(:,:,:), zeros(:)
character(len=512) :: f_in, f_out
integer(c_int64_t), target :: efp_h(6), efp_u(6), npts
integer(c_int64_t) :: bc_h, bc_u
real(c_double) :: sum_h, sum_u, amin, amax

call get_command_argument(1, f_in) ; call get_command_argument(2, f_out)
open(newunit=u_in, file=trim(f_in), access="stream", form="unformatted", status="old")
read(u_in) hdr
ni = hdr(1) ; nj = hdr(2) ; nk = hdr(3) ; halo = hdr(4)
isd = 1 ; ied = ni + 2*halo ; jsd = 1 ; jed = nj + 2*halo
read(u_in) scal, dt
allocate(mT(isd:ied,jsd:jed,8), mU(isd-1:ied,jsd:jed,8), mV(isd:ied,jsd-1:jed,8), mQ(isd-1:ied,jsd-1:jed,8))
read(u_in) mT, mU, mV, mQ
allocate(u(isd-1:ied,jsd:jed,nk), v(isd:ied,jsd-1:jed,nk), h(isd:ied,jsd:jed,nk), T(isd:ied,jsd:jed,nk), S(isd:ied,jsd:jed,nk), &
         taux(isd-1:ied,jsd:jed), tauy(isd:ied,jsd-1:jed), eta_av(isd:ied,jsd:jed), uhtr(isd-1:ied,jsd:jed,nk))
read(u_in) u, v, h, T, S, taux, tauy
close(u_in)
nh2 = size(mT(:,:,1)) ; nu2 = size(mU(:,:,1)) ; nv2 = size(mV(:,:,1)) ; nq2 = size(mQ(:,:,1))
nh3 = nh2*nk ; nu3 = nu2*nk ; nv3 = nv2*nk
allocate(zeros(max(nu3, nv3) + nq2)) ; zeros(:) = 0.0d0
maxdepth = maxval(mT(:,:,8))

! ---- the grid (mom6hip_grid_t: index ranges, vertical-grid scalars, c_loc of every metric in the order of the header)
cg%isc = isd+halo ; cg%iec = ied-halo ; cg%jsc = jsd+halo ; cg%jec = jed-halo
cg%isd = isd ; cg%ied = ied ; cg%jsd = jsd ; cg%jed = jed ; cg%nk = nk ; cg%symmetric = 1
cg%reentrant_x = hdr(5) ; cg%reentrant_y = hdr(6) ; cg%first_direction = hdr(7) ; cg%tripolar_n = 0
cg%Angstrom_H = scal(1) ; cg%H_subroundoff = scal(2) ; cg%dZ_subroundoff = scal(3) ; cg%H_to_Z = scal(4) ; cg%Z_to_H = scal(5)
cg%g_Earth = scal(6) ; cg%Rho0 = scal(7) ; cg%reserved1(:) = 0.0d0 ; cg%reserved2(:) = c_null_ptr
cg%mask2dT = c_loc(mT(isd,jsd,1)) ; cg%areaT = c_loc(mT(isd,jsd,2)) ; cg%IareaT = c_loc(mT(isd,jsd,3)) ; cg%dxT = c_loc(mT(isd,jsd,4))
cg%dyT = c_loc(mT(isd,jsd,5)) ; cg%IdxT = c_loc(mT(isd,jsd,6)) ; cg%IdyT = c_loc(mT(isd,jsd,7)) ; cg%bathyT = c_loc(mT(isd,jsd,8))
cg%mask2dCu = c_loc(mU(isd-1,jsd,1)) ; cg%dxCu = c_loc(mU(isd-1,jsd,2)) ; cg%dyCu = c_loc(mU(isd-1,jsd,3))
cg%dy_Cu = c_loc(mU(isd-1,jsd,4)) ; cg%IdxCu = c_loc(mU(isd-1,jsd,5)) ; cg%IdyCu = c_loc(mU(isd-1,jsd,6))
cg%areaCu = c_loc(mU(isd-1,jsd,7)) ; cg%IareaCu = c_loc(mU(isd-1,jsd,8))
cg%mask2dCv = c_loc(mV(isd,jsd-1,1)) ; cg%dxCv = c_loc(mV(isd,jsd-1,2)) ; cg%dyCv = c_loc(mV(isd,jsd-1,3))
cg%dx_Cv = c_loc(mV(isd,jsd-1,4)) ; cg%IdxCv = c_loc(mV(isd,jsd-1,5)) ; cg%IdyCv = c_loc(mV(isd,jsd-1,6))
cg%areaCv = c_loc(mV(isd,jsd-1,7)) ; cg%IareaCv = c_loc(mV(isd,jsd-1,8))
cg%mask2dBu = c_loc(mQ(isd-1,jsd-1,1)) ; cg%dxBu = c_loc(mQ(isd-1,jsd-1,2)) ; cg%dyBu = c_loc(mQ(isd-1,jsd-1,3))
cg%areaBu = c_loc(mQ(isd-1,jsd-1,4)) ; cg%IareaBu = c_loc(mQ(isd-1,jsd-1,5)) ; cg%CoriolisBu = c_loc(mQ(isd-1,jsd-1,6))
cg%IdxBu = c_loc(mQ(isd-1,jsd-1,7)) ; cg%IdyBu = c_loc(mQ(isd-1,jsd-1,8))
rc = mom6hip_init(0) ; call check("mom6hip_init")
rc = mom6hip_grid_create(cg, c_null_ptr, ctx) ; call check("mom6hip_grid_create")

! ---- the control structures of the modules the step calls, with the reference's defaults
ccs%upwind_1st = 0 ; ccs%monotonic = 0 ; ccs%simple_2nd = 0 ; ccs%aggress_adjust = 0 ; ccs%vol_CFL = 0 ; ccs%better_iter = 1
ccs%use_visc_rem_max = 1 ; ccs%marginal_faces = 1 ; ccs%tol_eta = 0.5d0*nk*cg%Angstrom_H ; ccs%tol_vel = 3.0d8
ccs%CFL_limit_adjust = 0.5d0
cor%coriolis_scheme = 1 ; cor%ke_scheme = 10 ; cor%no_slip = 0 ; cor%bound_coriolis = 1 ; cor%coriolis_en_dis = 0 ; cor%reserved(:) = 0
pcs%Rho0 = cg%Rho0 ; pcs%GFS_scale = 1.0d0 ; pcs%Z_ref = 0.0d0 ; pcs%reconstruct = 1 ; pcs%Recon_Scheme = 1
pcs%boundary_extrap = 1 ; pcs%useMassWghtInterp = 0
eos%form = MOM6HIP_EOS_WRIGHT ; eos%reserved = 0 ; eos%Rho_T0_S0 = 1000.0d0 ; eos%dRho_dT = -0.2d0 ; eos%dRho_dS = 0.8d0
bcs%dtbt = 0.0d0 ; bcs%dtbt_max = 0.0d0 ; bcs%dtbt_fraction = 0.98d0 ; bcs%bebt = 0.1d0 ; bcs%dt_bt_filter = -0.25d0
bcs%vel_underflow = 0.0d0 ; bcs%G_extra = 0.0d0 ; bcs%BT_Coriolis_scale = 1.0d0 ; bcs%Z_ref = 0.0d0 ; bcs%maxCFL_BT_cont = 0.25d0 ; bcs%reserved0(:) = 0.0d0
bcs%Sadourny = 1 ; bcs%linearized_BT_PV = 1 ; bcs%strong_drag = 0 ; bcs%visc_rem_u_uh0 = 0 ; bcs%adjust_BT_cont = 0
bcs%use_wide_halos = 1 ; bcs%hvel_scheme = 4 ; bcs%nstep_last = 0 ; bcs%unsupported(:) = 0 ; bcs%bound_BT_corr = 0 ; bcs%BT_project_velocity = 0 ; bcs%Nonlinear_continuity = 0 ; bcs%Nonlin_cont_update_period = 1
bcs%frhatu = dalloc(nu3) ; bcs%frhatv = dalloc(nv3) ; bcs%eta_cor = dalloc(nh2) ; bcs%IDatu = dalloc(nu2) ; bcs%IDatv = dalloc(nv2)
bcs%ubtav = dalloc(nu2) ; bcs%vbtav = dalloc(nv2) ; bcs%q_D = dalloc(nq2) ; bcs%D_u_Cor = dalloc(nu2) ; bcs%D_v_Cor = dalloc(nv2)
bcs%reserved2(:) = c_null_ptr
rc = mom6hip_barotropic_init(ctx, bcs, MOM6HIP_MEM_DEVICE) ; call check("mom6hip_barotropic_init")
rc = mom6hip_set_dtbt(ctx, bcs, c_null_ptr, c_null_ptr, cg%H_to_Z*cg%g_Earth, min(10.0d0, 0.05d0*maxdepth), MOM6HIP_MEM_DEVICE)
call check("mom6hip_set_dtbt")
btc%FA_u_W0 = dalloc(nu2) ; btc%FA_u_WW = dalloc(nu2) ; btc%FA_u_E0 = dalloc(nu2) ; btc%FA_u_EE = dalloc(nu2)
btc%uBT_WW = dalloc(nu2) ; btc%uBT_EE = dalloc(nu2)
btc%FA_v_S0 = dalloc(nv2) ; btc%FA_v_SS = dalloc(nv2) ; btc%FA_v_N0 = dalloc(nv2) ; btc%FA_v_NN = dalloc(nv2)
btc%vBT_SS = dalloc(nv2) ; btc%vBT_NN = dalloc(nv2) ; btc%h_u = dalloc(nu3) ; btc%h_v = dalloc(nv3)

! ---- MOM_dyn_split_RK2_CS: parameters, the sub-module structures, the arrays the reference allocates in its CS
cs%be = 0.6d0 ; cs%begw = 0.0d0 ; cs%BT_use_layer_fluxes = 1 ; cs%store_CAu = 1 ; cs%CAu_pred_stored = 0 ; cs%split_bottom_stress = 0
cs%reserved0(:) = 0 ; cs%set_visc_CSp = c_null_ptr ; cs%OBC = c_null_ptr
cs%continuity_CSp = c_loc(ccs) ; cs%CoriolisAdv = c_loc(cor) ; cs%PressureForce_CSp = c_loc(pcs) ; cs%eqn_of_state = c_loc(eos)
cs%barotropic_CSp = c_loc(bcs) ; cs%BT_cont = c_loc(btc) ; cs%hooks = c_null_ptr
cs%vertvisc_CSp = c_null_ptr ; cs%visc = c_null_ptr ; cs%hor_visc = c_null_ptr
cs%CAu = dalloc(nu3) ; cs%CAv = dalloc(nv3) ; cs%CAu_pred = dalloc(nu3) ; cs%CAv_pred = dalloc(nv3)
cs%PFu = dalloc(nu3) ; cs%PFv = dalloc(nv3) ; cs%diffu = dalloc(nu3) ; cs%diffv = dalloc(nv3)
cs%visc_rem_u = dalloc(nu3) ; cs%visc_rem_v = dalloc(nv3) ; cs%u_accel_bt = dalloc(nu3) ; cs%v_accel_bt = dalloc(nv3)
cs%u_av = dalloc(nu3) ; cs%v_av = dalloc(nv3) ; cs%h_av = dalloc(nh3) ; cs%pbce = dalloc(nh3)
cs%eta = dalloc(nh2) ; cs%eta_PF = dalloc(nh2) ; cs%uhbt = dalloc(nu2) ; cs%vhbt = dalloc(nv2)

! ---- the prognostic state and the step's other arguments, resident on the device
d_u = dput(c_loc(u), nu3) ; d_v = dput(c_loc(v), nv3) ; d_h = dput(c_loc(h), nh3) ; d_T = dput(c_loc(T), nh3) ; d_S = dput(c_loc(S), nh3)
d_taux = dput(c_loc(taux), nu2) ; d_tauy = dput(c_loc(tauy), nv2)
d_uh = dalloc(nu3) ; d_vh = dalloc(nv3) ; d_uhtr = dalloc(nu3) ; d_vhtr = dalloc(nv3) ; d_eta_av = dalloc(nh2)

rc = mom6hip_dyn_split_rk2_init(ctx, cs, d_u, d_v, d_h, d_uh, d_vh, dt) ; call check("mom6hip_dyn_split_rk2_init")
do n = 1, 2
  rc = mom6hip_step_dyn_split_rk2(ctx, cs, d_u, d_v, d_h, d_T, d_S, dt, d_taux, d_tauy, cg%Z_to_H/cg%Rho0, d_uh, d_vh, d_uhtr, &
                                  d_vhtr, d_eta_av, merge(1_c_int32_t, 0_c_int32_t, n == 1))
  call check("mom6hip_step_dyn_split_rk2")
enddo

! ---- the debugging statistics of MOM_checksums / MOM_coms, formed where the fields live (no copy to the host)
rc = mom6hip_chksum(ctx, d_h, MOM6HIP_POS_H, int(nk, c_int32_t), 0_c_int32_t, 0_c_int32_t, 0_c_int32_t, 1.0_c_double, bc_h, amin, amax, &
                    MOM6HIP_MEM_DEVICE) ; call check("mom6hip_chksum h")
rc = mom6hip_chksum(ctx, d_u, MOM6HIP_POS_U, int(nk, c_int32_t), 0_c_int32_t, 0_c_int32_t, 1_c_int32_t, 1.0_c_double, bc_u, amin, amax, &
                    MOM6HIP_MEM_DEVICE) ; call check("mom6hip_chksum u")
rc = mom6hip_reproducing_sum(ctx, d_h, MOM6HIP_POS_H, int(nk, c_int32_t), sum_h, c_null_ptr, c_loc(efp_h), c_null_ptr, c_loc(npts), &
                             c_null_ptr, MOM6HIP_MEM_DEVICE) ; call check("mom6hip_reproducing_sum h")
rc = mom6hip_reproducing_sum(ctx, d_u, MOM6HIP_POS_U, int(nk, c_int32_t), sum_u, c_null_ptr, c_loc(efp_u), c_null_ptr, c_null_ptr, &
                             c_null_ptr, MOM6HIP_MEM_DEVICE) ; call check("mom6hip_reproducing_sum u")
write(*,'(a,2(1x,i0),1x,i0,12(1x,i0))') "rk2_driver stats", bc_h, bc_u, npts, efp_h, efp_u

! ---- the state to the host the way restarts and diagnostics leave the device: staged (snapshots on the compute stream, the
! copies on the copy stream), one wait where the host reads the arrays
rc = mom6hip_host_register(c_loc(u), 8_c_int64_t*nu3) ; call check("host_register u")
rc = mom6hip_stage_to_host(ctx, c_loc(u), d_u, 8_c_int64_t*nu3) ; call check("stage u")
rc = mom6hip_stage_to_host(ctx, c_loc(v), d_v, 8_c_int64_t*nv3) ; call check("stage v")
rc = mom6hip_stage_to_host(ctx, c_loc(h), d_h, 8_c_int64_t*nh3) ; call check("stage h")
rc = mom6hip_stage_to_host(ctx, c_loc(eta_av), d_eta_av, 8_c_int64_t*nh2) ; call check("stage eta_av")
rc = mom6hip_stage_to_host(ctx, c_loc(uhtr), d_uhtr, 8_c_int64_t*nu3) ; call check("stage uhtr")
rc = mom6hip_stage_wait(ctx) ; call check("stage_wait")
rc = mom6hip_host_unregister(c_loc(u)) ; call check("host_unregister u")
open(newunit=u_out, file=trim(f_out), access="stream", form="unformatted", status="replace")
write(u_out) u, v, h, eta_av, uhtr
close(u_out)
rc = mom6hip_grid_destroy(ctx)
write(*,'(a,i0,a,es23.16)') "rk2_driver ok nstep=", bcs%nstep_last, " dtbt=", bcs%dtbt

contains

subroutine check(who)
  character(len=*), intent(in) :: who
  if (rc /= 0) then
    write(0,'(a)') "rk2_driver: "//who//": "//mom6hip_error_string() ; stop 2
  endif
end subroutine check

!> n zeroed doubles in HBM
function dalloc(n) result(p)
  integer(c_int64_t), intent(in) :: n
  type(c_ptr) :: p
  integer(c_int) :: rc2
  rc2 = mom6hip_malloc(p, 8_c_int64_t*n)
  if (rc2 == 0) rc2 = mom6hip_sync_to_device(ctx, p, c_loc(zeros), 8_c_int64_t*n)
  if (rc2 /= 0) then
    write(0,'(a)') "rk2_driver: device allocation failed: "//mom6hip_error_string() ; stop 3
  endif
end function dalloc

!> a device copy of n doubles of the host
function dput(hp, n) result(p)
  type(c_ptr),        intent(in) :: hp
  integer(c_int64_t), intent(in) :: n
  type(c_ptr) :: p
  integer(c_int) :: rc2
  rc2 = mom6hip_malloc(p, 8_c_int64_t*n)
  if (rc2 == 0) rc2 = mom6hip_sync_to_device(ctx, p, hp, 8_c_int64_t*n)
  if (rc2 /= 0) then
    write(0,'(a)') "rk2_driver: upload failed: "//mom6hip_error_string() ; stop 3
  endif
end function dput

end program rk2_driver
