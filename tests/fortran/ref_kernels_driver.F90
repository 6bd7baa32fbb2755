!> Runs the REFERENCE's own kernels -- /root/reference/src/core/MOM_continuity_PPM.F90, MOM_CoriolisAdv.F90 and
!! src/tracer/MOM_tracer_advect.F90, compiled unmodified where they lie -- against the stand-ins of tests/fortran/stubs, on the
!! inputs tests/test_reference_kernels.py writes (the file format of shim_driver.F90 followed by an advection section), and
!! writes their results; the test compares them with the oracle bit for bit, tools/calibrate_ref_kernels.py times them beside the
!! oracle's port.  Build container only (the reference is not on the GPU box); supplementary evidence: a build against
!! stand-ins pins nothing (DESIGN.md section 5).
!! Usage: ref_kernels_driver <input file> <output file> [repetitions for the timing lines [OBC file]]
!! The OBC file (tests/test_reference_kernels.py: write_obc): the ocean_OBC_type as open_boundary_config / open_boundary_init leave it --
!! dyn_driver.F90's format, then the segments' tangential_vel and tangential_grad, then their tracer registries.
program ref_kernels_driver
use, intrinsic :: iso_c_binding
use MOM_continuity_PPM, only : continuity_PPM, continuity_PPM_init, continuity_PPM_CS
use MOM_CoriolisAdv,    only : CorAdCalc, CoriolisAdv_init, CoriolisAdv_end, CoriolisAdv_CS
use MOM_tracer_advect,  only : advect_tracer, tracer_advect_init, tracer_advect_CS
use MOM_tracer_registry, only : tracer_registry_type
use MOM_diag_mediator,  only : diag_ctrl, time_type
use MOM_domains,        only : MOM_domain_type, pass_var, EAST_FACE, NORTH_FACE
use MOM_file_parser,    only : param_file_type, param_set
use MOM_grid,           only : ocean_grid_type
use MOM_open_boundary,  only : ocean_OBC_type, segment_tracer_registry_type
use MOM_unit_scaling,   only : unit_scale_type
use MOM_variables,      only : BT_cont_type, porous_barrier_type, accel_diag_ptrs, alloc_BT_cont_type
use MOM_verticalGrid,   only : verticalGrid_type
implicit none

type(ocean_grid_type), target :: G
type(verticalGrid_type) :: GV
type(unit_scale_type) :: US
type(param_file_type) :: pf
type(time_type), target :: Time
type(diag_ctrl), target :: diag
type(accel_diag_ptrs), target :: AD
type(continuity_PPM_CS) :: CS
type(CoriolisAdv_CS) :: CCS
type(tracer_advect_CS), pointer :: ACS => NULL()
type(tracer_registry_type), pointer :: Reg => NULL()
type(ocean_OBC_type), pointer :: OBC => NULL()
type(porous_barrier_type) :: pbv
type(BT_cont_type), pointer :: BT => NULL()
integer(c_int32_t) :: hdr(8), ahdr(4), oflags(8), sflags(20), gflags(8), rflags(2)
integer(c_int32_t), allocatable :: seg_u(:,:), seg_v(:,:)
integer :: u_obc, nseg, i0, i1, j0, j1, q, nreg
real :: oscal(2), conc
character(len=512) :: f_obc
integer :: ni, nj, nk, halo, u_in, u_out, isd, ied, jsd, jed, nrep, n, m, ntr
integer(kind=8) :: c0, c1, crate
real :: scal(7), dt, dt_adv
real, allocatable, dimension(:,:,:) :: u, v, h, hp, uh, vh, hp2, uh2, vh2, vru, vrv, u_cor, v_cor, CAu, CAv, T, S
real, allocatable, dimension(:,:,:) :: h_end, uhtr, vhtr
real, allocatable, target, dimension(:,:,:,:) :: tr, tr0
real, allocatable, dimension(:,:) :: uhbt, vhbt
character(len=512) :: f_in, f_out, arg
character(len=16) :: scheme

call get_command_argument(1, f_in) ; call get_command_argument(2, f_out)
nrep = 0
if (command_argument_count() >= 3) then ; call get_command_argument(3, arg) ; read(arg, *) nrep ; endif
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
         vru(isd-1:ied,jsd:jed,nk), vrv(isd:ied,jsd-1:jed,nk), T(isd:ied,jsd:jed,nk), S(isd:ied,jsd:jed,nk))
read(u_in) u, v, h, uhbt, vhbt, vru, vrv
read(u_in) T, S
allocate(hp(isd:ied,jsd:jed,nk), hp2(isd:ied,jsd:jed,nk), uh(isd-1:ied,jsd:jed,nk), uh2(isd-1:ied,jsd:jed,nk), &
         vh(isd:ied,jsd-1:jed,nk), vh2(isd:ied,jsd-1:jed,nk), u_cor(isd-1:ied,jsd:jed,nk), v_cor(isd:ied,jsd-1:jed,nk), &
         CAu(isd-1:ied,jsd:jed,nk), CAv(isd:ied,jsd-1:jed,nk))
hp = h ; hp2 = h ; uh = 0.0 ; vh = 0.0 ; uh2 = 0.0 ; vh2 = 0.0 ; u_cor = 0.0 ; v_cor = 0.0 ; CAu = 0.0 ; CAv = 0.0
! no porous barriers: the whole face is open in every layer (MOM_porous_barriers.F90, the default)
allocate(pbv%por_face_areaU(isd-1:ied,jsd:jed,nk), pbv%por_face_areaV(isd:ied,jsd-1:jed,nk))
allocate(pbv%por_layer_widthU(isd-1:ied,jsd:jed,nk+1), pbv%por_layer_widthV(isd:ied,jsd-1:jed,nk+1))
pbv%por_face_areaU = 1.0 ; pbv%por_face_areaV = 1.0 ; pbv%por_layer_widthU = 1.0 ; pbv%por_layer_widthV = 1.0

! ---- the open boundaries (optional fourth argument)
f_obc = ""
if (command_argument_count() >= 4) call get_command_argument(4, f_obc)
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
  endif ; enddo
  ! segment%tangential_vel, tangential_grad at the corner points of the segment (IsdB:IedB, JsdB:JedB, nk)
  do m=1,nseg ; if (OBC%segment(m)%on_pe) then
    i0 = OBC%segment(m)%HI%IsdB ; i1 = OBC%segment(m)%HI%IedB ; j0 = OBC%segment(m)%HI%JsdB ; j1 = OBC%segment(m)%HI%JedB
    allocate(OBC%segment(m)%tangential_vel(i0:i1,j0:j1,nk), OBC%segment(m)%tangential_grad(i0:i1,j0:j1,nk))
    read(u_obc) OBC%segment(m)%tangential_vel, OBC%segment(m)%tangential_grad
  endif ; enddo
  ! the segments' tracer registries: [ntseg], then per entry [ntr_index, has a reservoir], OBC_inflow_conc, [tres in the layout of normal_vel]
  do m=1,nseg ; if (OBC%segment(m)%on_pe) then
    read(u_obc) rflags(1)
    nreg = rflags(1)
    if (nreg > 0) then
      allocate(OBC%segment(m)%tr_Reg)
      OBC%segment(m)%tr_Reg%ntseg = nreg
      do q=1,nreg
        read(u_obc) rflags
        read(u_obc) conc
        OBC%segment(m)%tr_Reg%Tr(q)%ntr_index = rflags(1) ; OBC%segment(m)%tr_Reg%Tr(q)%OBC_inflow_conc = conc
        if (rflags(2) /= 0) then
          allocate(OBC%segment(m)%tr_Reg%Tr(q)%tres(lbound(OBC%segment(m)%normal_vel,1):ubound(OBC%segment(m)%normal_vel,1), &
                   lbound(OBC%segment(m)%normal_vel,2):ubound(OBC%segment(m)%normal_vel,2), nk))
          read(u_obc) OBC%segment(m)%tr_Reg%Tr(q)%tres
        endif
      enddo
    endif
  endif ; enddo
  close(u_obc)
endif

call param_set(pf, "REENTRANT_X", merge("True ", "False", hdr(5) /= 0))
call param_set(pf, "REENTRANT_Y", merge("True ", "False", hdr(6) /= 0))
call param_set(pf, "BOUND_CORIOLIS", "True")
call continuity_PPM_init(Time, G, GV, US, pf, diag, CS)
call CoriolisAdv_init(Time, G, GV, US, pf, diag, AD, CCS)
call alloc_BT_cont_type(BT, isd, ied, jsd, jed, nk, alloc_faces=.true.)

call continuity_PPM(u, v, h, hp, uh, vh, dt, G, GV, US, CS, OBC, pbv, visc_rem_u=vru, visc_rem_v=vrv, BT_cont=BT)
call continuity_PPM(u, v, h, hp2, uh2, vh2, dt, G, GV, US, CS, OBC, pbv, uhbt, vhbt, vru, vrv, u_cor, v_cor, BT_cont=BT)
call pass_var(uh2, G%Domain, position=EAST_FACE) ; call pass_var(vh2, G%Domain, position=NORTH_FACE)
call CorAdCalc(u, v, h, uh2, vh2, CAu, CAv, OBC, AD, G, GV, US, CCS, pbv)

! ---- the advection section: ntr, x_first (-1: the default), max_iter (0: the default), scheme code; dt; h_end, uhtr, vhtr, tracers
read(u_in) ahdr
ntr = ahdr(1)
read(u_in) dt_adv
allocate(h_end(isd:ied,jsd:jed,nk), uhtr(isd-1:ied,jsd:jed,nk), vhtr(isd:ied,jsd-1:jed,nk), tr(isd:ied,jsd:jed,nk,ntr), &
         tr0(isd:ied,jsd:jed,nk,ntr))
read(u_in) h_end, uhtr, vhtr, tr
close(u_in)
tr0 = tr
scheme = "PLM" ; if (ahdr(4) == 1) scheme = "PPM:H3" ; if (ahdr(4) == 2) scheme = "PPM"
call param_set(pf, "TRACER_ADVECTION_SCHEME", trim(scheme))
call param_set(pf, "DT", "900.0")
call tracer_advect_init(Time, G, US, pf, diag, ACS)
allocate(Reg) ; Reg%ntr = ntr
do m = 1, ntr ; Reg%Tr(m)%t => tr(:,:,:,m) ; enddo
if (ahdr(2) < 0) then
  call advect_tracer(h_end, uhtr, vhtr, OBC, dt_adv, G, GV, US, ACS, Reg)
else
  call advect_tracer(h_end, uhtr, vhtr, OBC, dt_adv, G, GV, US, ACS, Reg, x_first_in=(ahdr(2) /= 0))
endif

open(newunit=u_out, file=trim(f_out), access="stream", form="unformatted", status="replace")
write(u_out) hp, uh, vh, hp2, uh2, vh2, u_cor, v_cor, CAu, CAv
write(u_out) BT%FA_u_W0, BT%FA_u_WW, BT%FA_u_E0, BT%FA_u_EE, BT%uBT_WW, BT%uBT_EE
write(u_out) BT%FA_v_S0, BT%FA_v_SS, BT%FA_v_N0, BT%FA_v_NN, BT%vBT_SS, BT%vBT_NN, BT%h_u, BT%h_v
write(u_out) tr
close(u_out)

! ---- timing lines (seconds per call, wall clock), for tools/calibrate_ref_kernels.py
if (nrep > 0) then
  call system_clock(count_rate=crate)
  call system_clock(c0)
  do n = 1, nrep
    call continuity_PPM(u, v, h, hp2, uh2, vh2, dt, G, GV, US, CS, OBC, pbv, uhbt, vhbt, vru, vrv, u_cor, v_cor, BT_cont=BT)
  enddo
  call system_clock(c1)
  write(*,'(a,es14.6)') "time continuity_PPM ", real(c1 - c0) / real(crate) / real(nrep)
  call system_clock(c0)
  do n = 1, nrep
    call CorAdCalc(u, v, h, uh2, vh2, CAu, CAv, OBC, AD, G, GV, US, CCS, pbv)
  enddo
  call system_clock(c1)
  write(*,'(a,es14.6)') "time CorAdCalc ", real(c1 - c0) / real(crate) / real(nrep)
  call system_clock(c0)
  do n = 1, nrep
    tr = tr0
    call advect_tracer(h_end, uhtr, vhtr, OBC, dt_adv, G, GV, US, ACS, Reg)
  enddo
  call system_clock(c1)
  write(*,'(a,es14.6)') "time advect_tracer ", real(c1 - c0) / real(crate) / real(nrep)
endif
call CoriolisAdv_end(CCS)
write(*,'(a)') "ref_kernels_driver ok"
end program ref_kernels_driver
