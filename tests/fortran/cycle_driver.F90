!> Drives ONE THERMODYNAMIC CYCLE of step_MOM's hot sequence through the module shims, reference-named procedures only (MOM.F90:1149-1165,
!! :1205-1260, :1335-1338, :1437-1447), ncycles times:
!!   thickness_diffuse -> pass_var(h) -> set_viscous_BBL -> nsteps x step_MOM_dyn_split_RK2 -> mixedlayer_restrat -> pass_var(h) -> advect_tracer (T, S)
!!   -> tracer_hordiff (T, S) -> uhtr = vhtr = 0
!! With GPU_RESIDENT_DYNAMICS = True the fields stay in HBM through all of it: the pass_var between the calls runs on the device copies
!! (mom6hip_mirror_pass_var), the zeroing of uhtr / vhtr is announced (mom6hip_mirror_zeroed), and the host sees the fields after
!! dyn_split_RK2_sync_to_host at the end.  The test compares with the oracle bit for bit and bounds what crossed PCIe.
!! Usage: cycle_driver <input file> <output file> <parameter file>   (the input file of tests/fortran/dyn_driver.F90)
program cycle_driver
use, intrinsic :: iso_c_binding
use MOM_dynamics_split_RK2, only : MOM_dyn_split_RK2_CS, register_restarts_dyn_split_RK2, initialize_dyn_split_RK2
use MOM_dynamics_split_RK2, only : step_MOM_dyn_split_RK2, end_dyn_split_RK2, dyn_split_RK2_sync_to_host, dyn_split_RK2_host_was_modified
use MOM_dynamics_split_RK2, only : remap_dyn_split_RK2_aux_vars
use MOM_set_visc,      only : set_visc_CS, set_visc_init, set_viscous_BBL, set_visc_end
use MOM_ALE,           only : ALE_CS, ALE_init, ALE_set_extrap_boundaries, ALE_update_regrid_weights, ALE_regrid, ALE_remap_tracers
use MOM_ALE,           only : ALE_remap_set_h_vel, ALE_remap_velocities, ALE_end
use MOM_boundary_update, only : update_OBC_CS
use MOM_diag_mediator, only : diag_ctrl
use MOM_time_manager,  only : time_type
use MOM_wave_interface, only : wave_parameters_CS
use MOM_EOS,           only : EOS_type
use MOM_file_parser,   only : param_file_type, param_set, get_param
use MOM_forcing_type,  only : mech_forcing
use MOM_grid,          only : ocean_grid_type
use MOM_hor_index,     only : hor_index_type
use MOM_get_input,     only : directories
use MOM_lateral_mixing_coeffs, only : VarMix_CS
use MOM_MEKE_types,    only : MEKE_type
use MOM_open_boundary, only : ocean_OBC_type
use MOM_restart,       only : MOM_restart_CS
use MOM_thickness_diffuse, only : thickness_diffuse_CS, thickness_diffuse_init, thickness_diffuse
use MOM_mixed_layer_restrat, only : mixedlayer_restrat_CS, mixedlayer_restrat_init, mixedlayer_restrat_register_restarts, mixedlayer_restrat
use MOM_tracer_advect,   only : advect_tracer, tracer_advect_init, tracer_advect_end, tracer_advect_CS
use MOM_tracer_hor_diff, only : tracer_hordiff, tracer_hor_diff_init, tracer_hor_diff_end, tracer_hor_diff_CS
use MOM_tracer_registry, only : tracer_registry_type
use MOM_stochastics,     only : stochastic_CS
use MOM_diabatic_driver, only : diabatic_CS
use MOM_domains,         only : pass_var, CENTER, EAST_FACE, NORTH_FACE
use MOM_unit_scaling,  only : unit_scale_type
use MOM_variables,     only : vertvisc_type, thermo_var_ptrs, porous_barrier_type, accel_diag_ptrs, cont_diag_ptrs, ocean_internal_state
use MOM_verticalGrid,  only : verticalGrid_type
use mom6hip_c_api,     only : mom6hip_transfer_stats
use mom6hip_MOM_glue,  only : mom6hip_shared_context, mom6hip_shared_context_end, mom6hip_mirror_pass_var, mom6hip_mirror_zeroed
use mom6hip_MOM_glue,  only : mom6hip_mirrors_to_host
implicit none

type(ocean_grid_type), target :: G
type(hor_index_type) :: HI
type(verticalGrid_type), target :: GV
type(unit_scale_type) :: US
type(param_file_type) :: pf
type(time_type), target :: Time
type(diag_ctrl), target :: diag
type(MOM_restart_CS) :: restart_CS
type(ocean_internal_state) :: MIS
type(directories) :: dirs
type(set_visc_CS), target :: SV
type(MOM_dyn_split_RK2_CS), pointer :: CS => NULL()
type(vertvisc_type), target :: visc
type(thermo_var_ptrs) :: tv
type(mech_forcing) :: forces
type(porous_barrier_type) :: pbv
type(accel_diag_ptrs), target :: ADp
type(cont_diag_ptrs), target :: CDp
type(MEKE_type), target :: MEKE
type(VarMix_CS) :: VarMix
type(thickness_diffuse_CS) :: TD
type(mixedlayer_restrat_CS) :: MLE
logical :: mle_on, td_on, done, lflag
type(tracer_advect_CS), pointer :: ACS => NULL()
type(tracer_hor_diff_CS), pointer :: DCS => NULL()
type(tracer_registry_type), pointer :: Reg => NULL()
type(stochastic_CS) :: STOCH
type(diabatic_CS), pointer :: diabatic_CSp => NULL()
type(EOS_type), target :: EOS
real, dimension(:,:), pointer :: MLD => NULL(), h_MLD => NULL(), bflux => NULL()
integer :: nc, ncycles, i, j, k
logical :: test_ALE
real, allocatable, target, dimension(:,:,:) :: h_new, dzRegrid, hu0, hv0, hu1, hv1
type(ocean_OBC_type), pointer :: OBC => NULL()
type(update_OBC_CS), pointer :: update_OBC_CSp => NULL()
type(ALE_CS), pointer :: ALE_CSp => NULL()
real, dimension(:,:), pointer :: p_surf_begin => NULL(), p_surf_end => NULL()
type(wave_parameters_CS), pointer :: Waves => NULL()
integer(c_int32_t) :: hdr(8), hdr2(8)
integer(c_int64_t) :: xfer(4)
integer, target :: ntrunc
integer :: ni, nj, nk, halo, u_in, u_out, u_par, isd, ied, jsd, jed, n, nsteps, ios, eq, cont_stencil, rc
logical :: resident, calc_dtbt, calc_dtbt_init, bbl_each_step
real :: scal(7), dt, dtbt_in, dtbt_reset_period, dt_therm
real, allocatable, target, dimension(:,:,:) :: u, v, h, uh, vh, uhtr, vhtr
real, allocatable, target, dimension(:,:) :: eta, eta_av
real, allocatable, dimension(:,:) :: nk_u, nk_v
character(len=512) :: f_in, f_out, f_par, line

call get_command_argument(1, f_in) ; call get_command_argument(2, f_out) ; call get_command_argument(3, f_par)
open(newunit=u_in, file=trim(f_in), access="stream", form="unformatted", status="old")
read(u_in) hdr, hdr2
ni = hdr(1) ; nj = hdr(2) ; nk = hdr(3) ; halo = hdr(4)
nsteps = hdr2(1) ; resident = (hdr2(2) /= 0) ; bbl_each_step = (hdr2(7) == 1)
isd = 1 ; ied = ni + 2*halo ; jsd = 1 ; jed = nj + 2*halo
G%isd = isd ; G%ied = ied ; G%jsd = jsd ; G%jed = jed ; G%IsdB = isd-1 ; G%IedB = ied ; G%JsdB = jsd-1 ; G%JedB = jed
G%isc = isd+halo ; G%iec = ied-halo ; G%jsc = jsd+halo ; G%jec = jed-halo
G%IscB = G%isc-1 ; G%IecB = G%iec ; G%JscB = G%jsc-1 ; G%JecB = G%jec ; G%ke = nk ; GV%ke = nk
HI%isd = isd ; HI%ied = ied ; HI%jsd = jsd ; HI%jed = jed ; HI%IsdB = isd-1 ; HI%IedB = ied ; HI%JsdB = jsd-1 ; HI%JedB = jed
HI%isc = G%isc ; HI%iec = G%iec ; HI%jsc = G%jsc ; HI%jec = G%jec ; HI%IscB = G%IscB ; HI%IecB = G%IecB ; HI%JscB = G%JscB ; HI%JecB = G%JecB
G%first_direction = hdr(7)
allocate(G%Domain)
G%Domain%reentrant(1) = (hdr(5) /= 0) ; G%Domain%reentrant(2) = (hdr(6) /= 0)
G%Domain%nihalo = halo ; G%Domain%njhalo = halo ; G%Domain%niglobal = ni ; G%Domain%njglobal = nj
read(u_in) scal, dt
GV%Angstrom_H = scal(1) ; GV%H_subroundoff = scal(2) ; GV%dZ_subroundoff = scal(3) ; GV%H_to_Z = scal(4) ; GV%Z_to_H = scal(5)
GV%g_Earth = scal(6) ; GV%Rho0 = scal(7) ; GV%RZ_to_H = GV%Z_to_H / GV%Rho0 ; GV%H_to_RZ = GV%H_to_Z * GV%Rho0
GV%nk_rho_varies = hdr2(5) ; GV%nkml = hdr2(6)
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
G%max_depth = maxval(G%bathyT)
allocate(u(isd-1:ied,jsd:jed,nk), v(isd:ied,jsd-1:jed,nk), h(isd:ied,jsd:jed,nk), tv%T(isd:ied,jsd:jed,nk), tv%S(isd:ied,jsd:jed,nk), &
         forces%taux(isd-1:ied,jsd:jed), forces%tauy(isd:ied,jsd-1:jed), forces%ustar(isd:ied,jsd:jed))
allocate(GV%Rlay(nk), GV%g_prime(nk+1))
read(u_in) u, v, h, tv%T, tv%S, forces%taux, forces%tauy, forces%ustar, GV%Rlay, GV%g_prime
! the solo driver allocates forces%p_surf and leaves it at zero (MOM_surface_forcing.F90:260); MOM.F90:772 points p_surf_end at it
allocate(forces%p_surf(isd:ied,jsd:jed), source=0.0) ; p_surf_end => forces%p_surf
if (hdr2(3) /= 0) allocate(tv%eqn_of_state)      ! an equation of state is in use
allocate(uh(isd-1:ied,jsd:jed,nk), vh(isd:ied,jsd-1:jed,nk), uhtr(isd-1:ied,jsd:jed,nk), vhtr(isd:ied,jsd-1:jed,nk), &
         eta(isd:ied,jsd:jed), eta_av(isd:ied,jsd:jed), nk_u(isd-1:ied,jsd:jed), nk_v(isd:ied,jsd-1:jed))
uh = 0.0 ; vh = 0.0 ; uhtr = 0.0 ; vhtr = 0.0 ; eta = 0.0 ; eta_av = 0.0 ; nk_u = 0.0 ; nk_v = 0.0

! ---- the parameter file
call param_set(pf, "REENTRANT_X", merge("True ", "False", hdr(5) /= 0))
call param_set(pf, "REENTRANT_Y", merge("True ", "False", hdr(6) /= 0))
call param_set(pf, "GPU_RESIDENT_DYNAMICS", merge("True ", "False", resident))
open(newunit=u_par, file=trim(f_par), status="old", action="read")
do
  read(u_par, '(a)', iostat=ios) line
  if (ios /= 0) exit
  eq = index(line, "=")
  if (eq > 1 .and. line(1:1) /= "!") call param_set(pf, trim(adjustl(line(1:eq-1))), trim(adjustl(line(eq+1:))))
enddo
close(u_par)

if (hdr2(4) /= 0) then                           ! USE_REGRIDDING: ALE_init (MOM.F90:2772), REMAP_BOUNDARY_EXTRAP after the initialisation (:3136)
  call ALE_init(pf, GV, US, G%max_depth, ALE_CSp) ; call ALE_set_extrap_boundaries(pf, ALE_CSp)
endif
call get_param(pf, "MOM", "TEST_ALE", test_ALE, default=.false.)
if (test_ALE) then
  allocate(h_new(isd:ied,jsd:jed,nk), dzRegrid(isd:ied,jsd:jed,nk+1), hu0(isd-1:ied,jsd:jed,nk), hv0(isd:ied,jsd-1:jed,nk), &
           hu1(isd-1:ied,jsd:jed,nk), hv1(isd:ied,jsd-1:jed,nk))
  h_new = 0.0 ; dzRegrid = 0.0 ; hu0 = 0.0 ; hv0 = 0.0 ; hu1 = 0.0 ; hv1 = 0.0
endif
! ---- MOM.F90's initialisation order for these modules
call set_visc_init(Time, G, GV, US, pf, diag, visc, SV, restart_CS, OBC)
! the two lateral parameterisations beside the step accept the same parameter file (MOM.F90:2854, :3305-3313)
call mixedlayer_restrat_register_restarts(HI, GV, US, pf, MLE, restart_CS)
call thickness_diffuse_init(Time, G, GV, US, pf, diag, CDp, TD)
mle_on = mixedlayer_restrat_init(Time, G, GV, US, pf, diag, MLE, restart_CS)
call get_param(pf, "MOM", "THICKNESSDIFFUSE", td_on, default=.false.)
call register_restarts_dyn_split_RK2(HI, GV, US, pf, CS, restart_CS, uh, vh)
if (hdr2(7) == 0) then      ! the bottom boundary layer as given (set_viscous_BBL belongs to another test)
  read(u_in) visc%Kv_bbl_u, visc%Kv_bbl_v, visc%bbl_thick_u, visc%bbl_thick_v
else
  call set_viscous_BBL(u, v, h, tv, visc, G, GV, US, SV, pbv)
endif
if (hdr2(8) /= 0) then      ! USE_MEKE with MEKE_VISCOSITY_COEFF_KU: MEKE%Ku as the MEKE module left it, MEKE%mom_src for it to read
  allocate(MEKE%Ku(isd:ied,jsd:jed), MEKE%mom_src(isd:ied,jsd:jed), MEKE%GME_snk(isd:ied,jsd:jed))
  read(u_in) MEKE%Ku
  MEKE%mom_src(:,:) = 0.0 ; MEKE%GME_snk(:,:) = 1.0
endif
if (hdr(8) /= 0) then      ! the fields the modules beside the hot path hand to it (MOM_MEKE, MOM_lateral_mixing_coeffs: .testing/tc2, tc1)
  allocate(MEKE%Kh(isd:ied,jsd:jed), VarMix%L2u(isd-1:ied,jsd:jed), VarMix%L2v(isd:ied,jsd-1:jed), VarMix%SN_u(isd-1:ied,jsd:jed), &
           VarMix%SN_v(isd:ied,jsd-1:jed), VarMix%Res_fn_u(isd-1:ied,jsd:jed), VarMix%Res_fn_v(isd:ied,jsd-1:jed), &
           VarMix%Res_fn_h(isd:ied,jsd:jed), VarMix%Rd_dx_h(isd:ied,jsd:jed), &
           VarMix%slope_x(isd-1:ied,jsd:jed,nk+1), VarMix%slope_y(isd:ied,jsd-1:jed,nk+1))
  read(u_in) MEKE%Kh, VarMix%L2u, VarMix%L2v, VarMix%SN_u, VarMix%SN_v, VarMix%Res_fn_u, VarMix%Res_fn_v, VarMix%Res_fn_h, VarMix%Rd_dx_h, &
             VarMix%slope_x, VarMix%slope_y
  call get_param(pf, "MOM", "USE_MEKE", lflag, default=.false.)
  if (.not.lflag) deallocate(MEKE%Kh)
  call get_param(pf, "MOM", "MEKE_KHTH_FAC", MEKE%KhTh_fac, default=0.0)      ! MOM_MEKE.F90: the factors default to zero
  call get_param(pf, "MOM", "MEKE_KHTR_FAC", MEKE%KhTr_fac, default=0.0)
  ! VarMix_init (MOM_lateral_mixing_coeffs.F90:1168-1300)
  call get_param(pf, "MOM", "USE_VARIABLE_MIXING", VarMix%use_variable_mixing, default=.false.)
  call get_param(pf, "MOM", "USE_VISBECK", VarMix%use_Visbeck, default=.false.)
  call get_param(pf, "MOM", "RESOLN_SCALED_KH", VarMix%Resoln_scaled_Kh, default=.false.)
  call get_param(pf, "MOM", "RESOLN_SCALED_KHTH", VarMix%Resoln_scaled_KhTh, default=.false.)
  call get_param(pf, "MOM", "RESOLN_SCALED_KHTR", VarMix%Resoln_scaled_KhTr, default=.false.)
  call get_param(pf, "MOM", "USE_STORED_SLOPES", VarMix%use_stored_slopes, default=.false.)
endif
close(u_in)
call initialize_dyn_split_RK2(u, v, h, tv, uh, vh, eta, Time, G, GV, US, pf, diag, CS, restart_CS, dt, ADp, CDp, MIS, VarMix, MEKE, TD, &
                              OBC, update_OBC_CSp, ALE_CSp, SV, visc, dirs, ntrunc, pbv, calc_dtbt_init, cont_stencil)
! when the barotropic time step is recalculated (MOM.F90:2380-2389, :1227-1234)
call get_param(pf, "MOM", "DT_THERM", dt_therm, default=dt)
call get_param(pf, "MOM", "DTBT", dtbt_in, default=-0.98)
dtbt_reset_period = -1.0
if (dtbt_in <= 0.0) call get_param(pf, "MOM", "DTBT_RESET_PERIOD", dtbt_reset_period, default=dt_therm)
rc = mom6hip_transfer_stats(mom6hip_shared_context(G, GV), xfer, 1_c_int32_t)

call tracer_advect_init(Time, G, US, pf, diag, ACS)
call tracer_hor_diff_init(Time, G, GV, US, pf, diag, EOS, diabatic_CSp, DCS)
allocate(Reg) ; Reg%ntr = 2 ; Reg%Tr(1)%t => tv%T ; Reg%Tr(2)%t => tv%S
call get_param(pf, "MOM", "TEST_NCYCLES", ncycles, default=1)
do nc = 1, ncycles
  ! THICKNESSDIFFUSE_FIRST (MOM.F90:1149-1181)
  call thickness_diffuse(h, uhtr, vhtr, tv, dt_therm, G, GV, US, MEKE, VarMix, CDp, TD, STOCH)
  call mom6hip_mirror_pass_var(mom6hip_shared_context(G, GV), c_loc(h), nk, CENTER, done)
  if (.not.done) call pass_var(h, G%Domain)
  call set_viscous_BBL(u, v, h, tv, visc, G, GV, US, SV, pbv)      ! the first dynamic step of a thermodynamic cycle (MOM.F90:1200-1207)
  do n = 1, nsteps
    calc_dtbt = (dtbt_reset_period == 0.0) .or. ((dtbt_reset_period > 0.0) .and. (n == 1) .and. (nc == 1) .and. calc_dtbt_init)
    call step_MOM_dyn_split_RK2(u, v, h, tv, visc, Time, dt, forces, p_surf_begin, p_surf_end, uh, vh, uhtr, vhtr, eta_av, G, GV, US, CS, &
                                calc_dtbt, VarMix, MEKE, TD, pbv, STOCH, Waves)      ! as MOM.F90:1242-1245 calls it
  enddo
  ! MOM.F90:1335-1338
  call mixedlayer_restrat(h, uhtr, vhtr, tv, forces, dt_therm, MLD, h_MLD, bflux, VarMix, G, GV, US, MLE)
  call mom6hip_mirror_pass_var(mom6hip_shared_context(G, GV), c_loc(h), nk, CENTER, done)
  if (.not.done) call pass_var(h, G%Domain)
  ! step_MOM_tracer_dyn (MOM.F90:1437-1447)
  call advect_tracer(h, uhtr, vhtr, OBC, dt_therm, G, GV, US, ACS, Reg)
  call tracer_hordiff(h, dt_therm, MEKE, VarMix, visc, G, GV, US, DCS, Reg, tv)
  uhtr(:,:,:) = 0.0 ; vhtr(:,:,:) = 0.0
  call mom6hip_mirror_zeroed(mom6hip_shared_context(G, GV), c_loc(uhtr)) ; call mom6hip_mirror_zeroed(mom6hip_shared_context(G, GV), c_loc(vhtr))
  if (test_ALE) then
    ! step_MOM_thermo's ALE block (MOM.F90:1647-1700) works on the HOST arrays (MOM_ALE_hip.F90 stages them per call): the host asks for
    ! the fields first and announces afterwards that it has changed them -- what a diabatic step on the host needs as well
    call dyn_split_RK2_sync_to_host(CS) ; call mom6hip_mirrors_to_host(mom6hip_shared_context(G, GV))
    call ALE_update_regrid_weights(dt_therm, ALE_CSp)
    call ALE_regrid(G, GV, US, h, h_new, dzRegrid, tv, ALE_CSp)
    call ALE_remap_tracers(ALE_CSp, G, GV, h, h_new, Reg)
    call ALE_remap_set_h_vel(ALE_CSp, G, GV, h, hu0, hv0, OBC)
    call ALE_remap_set_h_vel(ALE_CSp, G, GV, h_new, hu1, hv1, OBC)
    call ALE_remap_velocities(ALE_CSp, G, GV, hu0, hv0, hu1, hv1, u, v)
    call remap_dyn_split_RK2_aux_vars(G, GV, CS, hu0, hv0, hu1, hv1, ALE_CSp)      ! (REMAP_AUXILIARY_VARS, MOM.F90:1678-1683; returns without it)
    do k=1,nk ; do j=G%jsc-1,G%jec+1 ; do i=G%isc-1,G%iec+1 ; h(i,j,k) = h_new(i,j,k) ; enddo ; enddo ; enddo      ! :1694-1698
    call dyn_split_RK2_host_was_modified(CS)
    call pass_var(u, G%Domain, position=EAST_FACE) ; call pass_var(v, G%Domain, position=NORTH_FACE)      ! pass_uv_T_S_h :1713-1719
    call pass_var(tv%T, G%Domain) ; call pass_var(tv%S, G%Domain) ; call pass_var(h, G%Domain)
  else
    ! pass_uv_T_S_h (MOM.F90:1713-1719): of its five fields only T and S have stale halos here (no diabatic step, no ALE in this cycle)
    call mom6hip_mirror_pass_var(mom6hip_shared_context(G, GV), c_loc(tv%T), nk, CENTER, done) ; if (.not.done) call pass_var(tv%T, G%Domain)
    call mom6hip_mirror_pass_var(mom6hip_shared_context(G, GV), c_loc(tv%S), nk, CENTER, done) ; if (.not.done) call pass_var(tv%S, G%Domain)
  endif
enddo
call dyn_split_RK2_sync_to_host(CS)
call mom6hip_mirrors_to_host(mom6hip_shared_context(G, GV))
rc = mom6hip_transfer_stats(mom6hip_shared_context(G, GV), xfer, 0_c_int32_t)
if (allocated(visc%nkml_visc_u)) then ; nk_u = visc%nkml_visc_u ; nk_v = visc%nkml_visc_v ; endif

open(newunit=u_out, file=trim(f_out), access="stream", form="unformatted", status="replace")
write(u_out) u, v, h, uh, vh, uhtr, vhtr, eta_av, nk_u, nk_v
write(u_out) tv%T, tv%S
if (allocated(MEKE%mom_src)) write(u_out) MEKE%mom_src
close(u_out)
call end_dyn_split_RK2(CS)
call set_visc_end(visc, SV)
call mom6hip_shared_context_end()
write(*,'(a,i0,a,i0,a,i0,a,i0,a,i0,a,i0,a,i0,a,i0)') "cycle_driver ok cont_stencil=", cont_stencil, " ntrunc=", ntrunc, " h2d_calls=", xfer(1), &
    " h2d_bytes=", xfer(2), " d2h_calls=", xfer(3), " d2h_bytes=", xfer(4), " thickness_diffuse=", merge(1, 0, td_on), &
    " mixedlayer_restrat=", merge(1, 0, mle_on)
end program cycle_driver
