!> Drives the GPU path the way MOM.F90 drives the reference: ONLY the reference-named procedures on plain host arrays --
!! set_visc_init (MOM.F90:3130), register_restarts_dyn_split_RK2 (:2834), initialize_dyn_split_RK2 (:3325), then per step
!! set_viscous_BBL (:1205) and step_MOM_dyn_split_RK2 (:1258), end_dyn_split_RK2 -- with the parameter file handed over as
!! KEY = VALUE lines (the transcribed sets of .testing/tc4, tc2, tc1 in tests/test_testing_configs.py), read by every module's
!! get_param under the reference's names.  With GPU_RESIDENT_DYNAMICS = True the fields stay in HBM between the steps; the
!! driver then asks for them once, at the end (dyn_split_RK2_sync_to_host), and prints what crossed PCIe.
!! Usage: dyn_driver <input file> <output file> <parameter file>
!! Built with -DREFERENCE_KERNELS (tests/test_reference_kernels.py) the same program drives the reference's OWN dynamical core on the CPU:
!! MOM_dynamics_split_RK2.F90 with MOM_continuity(_PPM), MOM_CoriolisAdv, MOM_PressureForce(_FV, _Montgomery), MOM_density_integrals, the
!! equation-of-state stack, MOM_barotropic, MOM_vert_friction, MOM_set_viscosity, MOM_hor_visc -- every one compiled where it lies against
!! the stand-ins of tests/fortran/stubs -- and the test compares its fields with the oracle's, bit for bit.  The two lateral
!! parameterisations beside the step and the transfer statistics belong to the shims' build only.
#ifdef REF_RK2B
! (with -DREFERENCE_KERNELS -DREF_RK2B: the reference's own MOM_dynamics_split_RK2b.F90 -- SPLIT_RK2B -- under the same driver)
#define MOM_dynamics_split_RK2 MOM_dynamics_split_RK2b
#define MOM_dyn_split_RK2_CS MOM_dyn_split_RK2b_CS
#define register_restarts_dyn_split_RK2 register_restarts_dyn_split_RK2b
#define initialize_dyn_split_RK2 initialize_dyn_split_RK2b
#define step_MOM_dyn_split_RK2 step_MOM_dyn_split_RK2b
#define end_dyn_split_RK2 end_dyn_split_RK2b
#endif
program dyn_driver
use, intrinsic :: iso_c_binding
use MOM_dynamics_split_RK2, only : MOM_dyn_split_RK2_CS, register_restarts_dyn_split_RK2, initialize_dyn_split_RK2
#ifdef REFERENCE_KERNELS
use MOM_dynamics_split_RK2, only : step_MOM_dyn_split_RK2, end_dyn_split_RK2
#else
use MOM_dynamics_split_RK2, only : step_MOM_dyn_split_RK2, end_dyn_split_RK2, dyn_split_RK2_sync_to_host
#endif
use MOM_set_visc,      only : set_visc_CS, set_visc_init, set_viscous_BBL, set_visc_end
use MOM_ALE,           only : ALE_CS
use MOM_boundary_update, only : update_OBC_CS
use MOM_diag_mediator, only : diag_ctrl
use MOM_time_manager,  only : time_type
use MOM_wave_interface, only : wave_parameters_CS
use MOM_stochastics,   only : stochastic_CS
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
#ifdef REFERENCE_KERNELS
use MOM_thickness_diffuse, only : thickness_diffuse_CS
use MOM_EOS,           only : EOS_init
#else
use MOM_thickness_diffuse, only : thickness_diffuse_CS, thickness_diffuse_init
use MOM_mixed_layer_restrat, only : mixedlayer_restrat_CS, mixedlayer_restrat_init, mixedlayer_restrat_register_restarts
#endif
use MOM_unit_scaling,  only : unit_scale_type
use MOM_variables,     only : vertvisc_type, thermo_var_ptrs, porous_barrier_type, accel_diag_ptrs, cont_diag_ptrs, ocean_internal_state
use MOM_verticalGrid,  only : verticalGrid_type
#ifndef REFERENCE_KERNELS
use mom6hip_c_api,     only : mom6hip_transfer_stats
use mom6hip_MOM_glue,  only : mom6hip_shared_context, mom6hip_shared_context_end, mom6hip_mirror_host_changed
#endif
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
#ifndef REFERENCE_KERNELS
type(mixedlayer_restrat_CS) :: MLE
#endif
logical :: mle_on, td_on
type(ocean_OBC_type), pointer :: OBC => NULL()
type(update_OBC_CS), pointer :: update_OBC_CSp => NULL()
type(ALE_CS), pointer :: ALE_CSp => NULL()
real, dimension(:,:), pointer :: p_surf_begin => NULL(), p_surf_end => NULL()
type(wave_parameters_CS), pointer :: Waves => NULL()
type(stochastic_CS) :: STOCH
integer(c_int32_t) :: hdr(8), hdr2(8)
integer(c_int64_t) :: xfer(4)
integer, target :: ntrunc
integer :: ni, nj, nk, halo, u_in, u_out, u_par, isd, ied, jsd, jed, n, nsteps, ios, eq, cont_stencil, rc, psurf_mode, i, j
logical :: resident, calc_dtbt, calc_dtbt_init, bbl_each_step
real :: scal(7), dt, dtbt_in, dtbt_reset_period, dt_therm
real, allocatable, target, dimension(:,:,:) :: u, v, h, uh, vh, uhtr, vhtr
real, allocatable, target, dimension(:,:) :: eta, eta_av
real, allocatable, dimension(:,:) :: nk_u, nk_v
character(len=512) :: f_in, f_out, f_par, f_obc, line
integer(c_int32_t) :: oflags(8), sflags(20), gflags(8)
integer(c_int32_t), allocatable :: seg_u(:,:), seg_v(:,:)
integer :: u_obc, nseg, m, i0, i1, j0, j1
real :: oscal(2)

call get_command_argument(1, f_in) ; call get_command_argument(2, f_out) ; call get_command_argument(3, f_par)
f_obc = "" ; if (command_argument_count() >= 4) call get_command_argument(4, f_obc)
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
if (hdr2(4) /= 0) allocate(ALE_CSp)              ! USE_REGRIDDING
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

! ---- the open boundaries as open_boundary_config / open_boundary_init and the first update_OBC_segment_data leave them (a fourth argument):
! [number_of_segments, OBC_pe, open_u, open_v, specified_u, specified_v, Flather_u, Flather_v], [zero_vorticity, freeslip_vorticity,
! computed_vorticity, specified_vorticity, zero_strain, freeslip_strain, computed_strain, zero_biharmonic], gamma_uv, rx_max, per segment
! [direction, open, specified, on_pe, is_E_or_W, is_N_or_S, IsdB, IedB, JsdB, JedB, isd, ied, jsd, jed, Flather, radiation, gradient,
! nudged, 0, 0], segnum_u, segnum_v, and for every segment on the PE normal_vel, normal_trans, normal_vel_bt, SSH
if (len_trim(f_obc) > 0) then
  open(newunit=u_obc, file=trim(f_obc), access="stream", form="unformatted", status="old")
  allocate(OBC)
  read(u_obc) oflags, gflags, oscal
  nseg = oflags(1)
  OBC%number_of_segments = nseg ; OBC%OBC_pe = (oflags(2) /= 0)
  OBC%open_u_BCs_exist_globally = (oflags(3) /= 0) ; OBC%open_v_BCs_exist_globally = (oflags(4) /= 0)
  OBC%specified_u_BCs_exist_globally = (oflags(5) /= 0) ; OBC%specified_v_BCs_exist_globally = (oflags(6) /= 0)
  OBC%Flather_u_BCs_exist_globally = (oflags(7) /= 0) ; OBC%Flather_v_BCs_exist_globally = (oflags(8) /= 0)
  OBC%zero_vorticity = (gflags(1) /= 0) ; OBC%freeslip_vorticity = (gflags(2) /= 0) ; OBC%computed_vorticity = (gflags(3) /= 0)
  OBC%specified_vorticity = (gflags(4) /= 0) ; OBC%zero_strain = (gflags(5) /= 0) ; OBC%freeslip_strain = (gflags(6) /= 0)
  OBC%computed_strain = (gflags(7) /= 0) ; OBC%zero_biharmonic = (gflags(8) /= 0)
  OBC%gamma_uv = oscal(1) ; OBC%rx_max = oscal(2)
  allocate(OBC%segment(nseg))
  do m=1,nseg
    read(u_obc) sflags
    OBC%segment(m)%direction = sflags(1) ; OBC%segment(m)%open = (sflags(2) /= 0) ; OBC%segment(m)%specified = (sflags(3) /= 0)
    OBC%segment(m)%on_pe = (sflags(4) /= 0) ; OBC%segment(m)%is_E_or_W = (sflags(5) /= 0) ; OBC%segment(m)%is_N_or_S = (sflags(6) /= 0)
    OBC%segment(m)%HI%IsdB = sflags(7) ; OBC%segment(m)%HI%IedB = sflags(8) ; OBC%segment(m)%HI%JsdB = sflags(9) ; OBC%segment(m)%HI%JedB = sflags(10)
    OBC%segment(m)%HI%isd = sflags(11) ; OBC%segment(m)%HI%ied = sflags(12) ; OBC%segment(m)%HI%jsd = sflags(13) ; OBC%segment(m)%HI%jed = sflags(14)
    OBC%segment(m)%Flather = (sflags(15) /= 0) ; OBC%segment(m)%radiation = (sflags(16) /= 0) ; OBC%segment(m)%gradient = (sflags(17) /= 0)
    OBC%segment(m)%nudged = (sflags(18) /= 0)
  enddo
  allocate(seg_u(isd-1:ied,jsd:jed), seg_v(isd:ied,jsd-1:jed), OBC%segnum_u(isd-1:ied,jsd:jed), OBC%segnum_v(isd:ied,jsd-1:jed))
  read(u_obc) seg_u, seg_v
  OBC%segnum_u(:,:) = seg_u(:,:) ; OBC%segnum_v(:,:) = seg_v(:,:)
  do m=1,nseg ; if (OBC%segment(m)%on_pe) then
    if (OBC%segment(m)%is_E_or_W) then
      i0 = OBC%segment(m)%HI%IsdB ; i1 = OBC%segment(m)%HI%IedB ; j0 = OBC%segment(m)%HI%jsd ; j1 = OBC%segment(m)%HI%jed
    else
      i0 = OBC%segment(m)%HI%isd ; i1 = OBC%segment(m)%HI%ied ; j0 = OBC%segment(m)%HI%JsdB ; j1 = OBC%segment(m)%HI%JedB
    endif
    allocate(OBC%segment(m)%normal_vel(i0:i1,j0:j1,nk), OBC%segment(m)%normal_trans(i0:i1,j0:j1,nk), OBC%segment(m)%normal_vel_bt(i0:i1,j0:j1), &
             OBC%segment(m)%SSH(i0:i1,j0:j1))
    read(u_obc) OBC%segment(m)%normal_vel, OBC%segment(m)%normal_trans, OBC%segment(m)%normal_vel_bt, OBC%segment(m)%SSH
#ifdef REF_OBC
    ! (the reference's own MOM_open_boundary.F90: what allocate_OBC_segment_data gives a radiating segment, :3642, :3678)
    if (OBC%segment(m)%radiation .and. OBC%segment(m)%is_E_or_W) allocate(OBC%segment(m)%rx_norm_rad(i0:i1,j0:j1,nk), source=0.0)
    if (OBC%segment(m)%radiation .and. OBC%segment(m)%is_N_or_S) allocate(OBC%segment(m)%ry_norm_rad(i0:i1,j0:j1,nk), source=0.0)
    ! (:3649-3658: the velocities and gradients along the segment that the computed / specified vorticity and strain read; zero, as value-type
    ! segment data leave them)
    if (OBC%computed_vorticity .or. OBC%computed_strain) &
      allocate(OBC%segment(m)%tangential_vel(OBC%segment(m)%HI%IsdB:OBC%segment(m)%HI%IedB,OBC%segment(m)%HI%JsdB:OBC%segment(m)%HI%JedB,nk), source=0.0)
    if (OBC%specified_vorticity .or. OBC%specified_strain) &
      allocate(OBC%segment(m)%tangential_grad(OBC%segment(m)%HI%IsdB:OBC%segment(m)%HI%IedB,OBC%segment(m)%HI%JsdB:OBC%segment(m)%HI%JedB,nk), source=0.0)
#endif
  endif ; enddo
  allocate(OBC%rx_normal(isd-1:ied,jsd:jed,nk), source=0.0) ; allocate(OBC%ry_normal(isd:ied,jsd-1:jed,nk), source=0.0)
  close(u_obc)
endif

! ---- MOM.F90's initialisation order for these modules
#ifdef REFERENCE_KERNELS
G%HI = HI ; G%Domain%symmetric = .true.
! the porous-barrier weights as MOM.F90 allocates them without the parameterisation: 1 everywhere (the shims do not read them)
allocate(pbv%por_face_areaU(isd-1:ied,jsd:jed,nk), source=1.0) ; allocate(pbv%por_face_areaV(isd:ied,jsd-1:jed,nk), source=1.0)
allocate(pbv%por_layer_widthU(isd-1:ied,jsd:jed,nk+1), source=1.0) ; allocate(pbv%por_layer_widthV(isd:ied,jsd-1:jed,nk+1), source=1.0)
if (associated(tv%eqn_of_state)) call EOS_init(pf, tv%eqn_of_state, US)      ! the reference's own MOM_EOS (MOM.F90:2746)
mle_on = .false. ; td_on = .false.
call set_visc_init(Time, G, GV, US, pf, diag, visc, SV, restart_CS, OBC)
#else
call set_visc_init(Time, G, GV, US, pf, diag, visc, SV, restart_CS, OBC)
! the two lateral parameterisations beside the step accept the same parameter file (MOM.F90:2854, :3305-3313)
call mixedlayer_restrat_register_restarts(HI, GV, US, pf, MLE, restart_CS)
call thickness_diffuse_init(Time, G, GV, US, pf, diag, CDp, TD)
mle_on = mixedlayer_restrat_init(Time, G, GV, US, pf, diag, MLE, restart_CS)
call get_param(pf, "MOM", "THICKNESSDIFFUSE", td_on, default=.false.)
#endif
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
close(u_in)
call initialize_dyn_split_RK2(u, v, h, tv, uh, vh, eta, Time, G, GV, US, pf, diag, CS, restart_CS, dt, ADp, CDp, MIS, VarMix, MEKE, TD, &
                              OBC, update_OBC_CSp, ALE_CSp, SV, visc, dirs, ntrunc, pbv, calc_dtbt_init, cont_stencil)
! when the barotropic time step is recalculated (MOM.F90:2380-2389, :1227-1234)
call get_param(pf, "MOM", "DT_THERM", dt_therm, default=dt)
call get_param(pf, "MOM", "DTBT", dtbt_in, default=-0.98)
dtbt_reset_period = -1.0
if (dtbt_in <= 0.0) call get_param(pf, "MOM", "DTBT_RESET_PERIOD", dtbt_reset_period, default=dt_therm)
#ifndef REFERENCE_KERNELS
rc = mom6hip_transfer_stats(mom6hip_shared_context(G, GV), xfer, 1_c_int32_t)
#else
xfer(:) = 0
#endif

! DRIVER_P_SURF (read by this driver only): 1 a surface pressure in forces%p_surf, with p_surf_end pointing at it as MOM.F90:772 has it
! (p_surf_begin is not associated: PressureForce takes forces%p_surf); 2 both p_surf_begin and p_surf_end (the pressure force takes
! p_surf_end, btstep the eta_PF interpolated between the two, MOM_dynamics_split_RK2.F90:435-442, :497-503).  The pressures are made of
! integers, so that the test forms the same bits: 1e5 + 8 mod(7i + 13j, 97) + 16 n at the end of step n, 4 mod(3i + 5j, 31) less at its start.
call get_param(pf, "dyn_driver", "DRIVER_P_SURF", psurf_mode, default=0)
if (psurf_mode == 2) allocate(p_surf_begin(isd:ied,jsd:jed), source=0.0)

do n = 1, nsteps
  if (bbl_each_step .and. (n > 1)) then      ! (resident: on the device mirrors of u, v, h, T, S)
    call set_viscous_BBL(u, v, h, tv, visc, G, GV, US, SV, pbv)
  endif
  if (psurf_mode > 0) then
    do j=jsd,jed ; do i=isd,ied
      forces%p_surf(i,j) = 1.0e5 + 8.0*real(mod(7*i + 13*j, 97)) + 16.0*real(n)
      if (psurf_mode == 2) p_surf_begin(i,j) = forces%p_surf(i,j) - 4.0*real(mod(3*i + 5*j, 31))
    enddo ; enddo
#ifndef REFERENCE_KERNELS
    call mom6hip_mirror_host_changed(c_loc(forces%p_surf))      ! (a new forcing field: its next reader uploads it)
    if (psurf_mode == 2) call mom6hip_mirror_host_changed(c_loc(p_surf_begin))
#endif
  endif
  calc_dtbt = (dtbt_reset_period == 0.0) .or. ((dtbt_reset_period > 0.0) .and. (n == 1) .and. calc_dtbt_init)
#ifdef REF_RK2B
  call step_MOM_dyn_split_RK2(u, v, h, tv, visc, Time, dt, forces, p_surf_begin, p_surf_end, uh, vh, uhtr, vhtr, eta_av, G, GV, US, CS, &
                              calc_dtbt, VarMix, MEKE, TD, pbv, Waves)             ! as MOM.F90:1250-1253 calls it
#else
  call step_MOM_dyn_split_RK2(u, v, h, tv, visc, Time, dt, forces, p_surf_begin, p_surf_end, uh, vh, uhtr, vhtr, eta_av, G, GV, US, CS, &
                              calc_dtbt, VarMix, MEKE, TD, pbv, STOCH, Waves)      ! as MOM.F90:1242-1245 calls it
#endif
enddo
#ifndef REFERENCE_KERNELS
call dyn_split_RK2_sync_to_host(CS)
rc = mom6hip_transfer_stats(mom6hip_shared_context(G, GV), xfer, 0_c_int32_t)
#endif
if (allocated(visc%nkml_visc_u)) then ; nk_u = visc%nkml_visc_u ; nk_v = visc%nkml_visc_v ; endif

open(newunit=u_out, file=trim(f_out), access="stream", form="unformatted", status="replace")
write(u_out) u, v, h, uh, vh, uhtr, vhtr, eta_av, nk_u, nk_v
if (allocated(MEKE%mom_src)) write(u_out) MEKE%mom_src
if (associated(OBC)) then
  write(u_out) OBC%rx_normal, OBC%ry_normal
  do m=1,OBC%number_of_segments ; if (OBC%segment(m)%on_pe) write(u_out) OBC%segment(m)%normal_vel ; enddo
endif
close(u_out)
call end_dyn_split_RK2(CS)
call set_visc_end(visc, SV)
#ifndef REFERENCE_KERNELS
call mom6hip_shared_context_end()
#endif
write(*,'(a,i0,a,i0,a,i0,a,i0,a,i0,a,i0,a,i0,a,i0)') "dyn_driver ok cont_stencil=", cont_stencil, " ntrunc=", ntrunc, " h2d_calls=", xfer(1), &
    " h2d_bytes=", xfer(2), " d2h_calls=", xfer(3), " d2h_bytes=", xfer(4), " thickness_diffuse=", merge(1, 0, td_on), &
    " mixedlayer_restrat=", merge(1, 0, mle_on)
end program dyn_driver
