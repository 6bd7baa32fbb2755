!> Drives the reference's OWN radiation_open_bdry_conds (src/core/MOM_open_boundary.F90:2196-3334, compiled in place with -DREF_OBC against
!! the stand-ins of tests/fortran/stubs) on a file written by tests/test_reference_kernels.py: every form of the routine -- Orlanski and
!! oblique radiation of the normal component, the gradient condition, nudging, the tangential velocities and their gradients in their
!! radiating, oblique and nudged forms -- `ncall` calls in a row, so that what the routine keeps between calls (OBC%rx_normal, ry_normal,
!! the oblique arrays) is carried.  The outputs are compared with the oracle bit for bit.  Build container only; nothing of the library is used.
!! With a third argument "reservoirs" it drives the reference's update_segment_tracer_reservoirs (:5373-5502) instead, on its own input layout.
!! Usage: ref_rad_driver <input file> <output file> [reservoirs]
program ref_rad_driver
use, intrinsic :: iso_c_binding
use MOM_domains,       only : MOM_domain_type
use MOM_grid,          only : ocean_grid_type
use MOM_open_boundary, only : ocean_OBC_type, radiation_open_bdry_conds, update_segment_tracer_reservoirs
use MOM_tracer_registry, only : tracer_registry_type
use MOM_unit_scaling,  only : unit_scale_type
use MOM_verticalGrid,  only : verticalGrid_type
implicit none

type(ocean_grid_type), target :: G
type(verticalGrid_type) :: GV
type(unit_scale_type) :: US
type(ocean_OBC_type), pointer :: OBC => NULL()
integer(c_int32_t) :: hdr(8), sflags(24)
integer(c_int32_t), allocatable :: seg_u(:,:), seg_v(:,:)
integer :: ni, nj, nk, halo, u_in, u_out, isd, ied, jsd, jed, nseg, ncall, m, n, i0, i1, j0, j1
real :: scal(3), tscale(2), dt
real, allocatable, dimension(:,:,:) :: u_new, u_old, v_new, v_old
character(len=512) :: f_in, f_out, f_mode
logical :: tan_any, grad_any

call get_command_argument(1, f_in) ; call get_command_argument(2, f_out)
f_mode = "" ; if (command_argument_count() >= 3) call get_command_argument(3, f_mode)
if (trim(f_mode) == "reservoirs") then
  call reservoirs(trim(f_in), trim(f_out))
  write(*,'(a)') "ref_rad_driver ok"
  stop
endif
open(newunit=u_in, file=trim(f_in), access="stream", form="unformatted", status="old")
! hdr = [ni, nj, nk, halo, number of segments, oblique segments exist, calls, -]
read(u_in) hdr
ni = hdr(1) ; nj = hdr(2) ; nk = hdr(3) ; halo = hdr(4) ; nseg = hdr(5) ; ncall = hdr(7)
isd = 1 ; ied = ni + 2*halo ; jsd = 1 ; jed = nj + 2*halo
G%isd = isd ; G%ied = ied ; G%jsd = jsd ; G%jed = jed ; G%IsdB = isd-1 ; G%IedB = ied ; G%JsdB = jsd-1 ; G%JedB = jed
G%isc = isd+halo ; G%iec = ied-halo ; G%jsc = jsd+halo ; G%jec = jed-halo
G%IscB = G%isc-1 ; G%IecB = G%iec ; G%JscB = G%jsc-1 ; G%JecB = G%jec ; G%ke = nk ; GV%ke = nk
G%HI%isd = isd ; G%HI%ied = ied ; G%HI%jsd = jsd ; G%HI%jed = jed ; G%HI%IsdB = isd-1 ; G%HI%IedB = ied ; G%HI%JsdB = jsd-1 ; G%HI%JedB = jed
G%HI%isc = G%isc ; G%HI%iec = G%iec ; G%HI%jsc = G%jsc ; G%HI%jec = G%jec
G%HI%IscB = G%IscB ; G%HI%IecB = G%IecB ; G%HI%JscB = G%JscB ; G%HI%JecB = G%JecB
allocate(G%Domain)
G%Domain%reentrant(1) = .false. ; G%Domain%reentrant(2) = .false.
G%Domain%nihalo = halo ; G%Domain%njhalo = halo ; G%Domain%niglobal = ni ; G%Domain%njglobal = nj
read(u_in) scal      ! gamma_uv, rx_max, dt
dt = scal(3)
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
allocate(u_new(isd-1:ied,jsd:jed,nk), u_old(isd-1:ied,jsd:jed,nk), v_new(isd:ied,jsd-1:jed,nk), v_old(isd:ied,jsd-1:jed,nk))
read(u_in) u_new, u_old, v_new, v_old

allocate(OBC)
OBC%number_of_segments = nseg ; OBC%OBC_pe = .true. ; OBC%ke = nk
OBC%gamma_uv = scal(1) ; OBC%rx_max = scal(2)
allocate(OBC%rx_normal(isd-1:ied,jsd:jed,nk), OBC%ry_normal(isd:ied,jsd-1:jed,nk))
read(u_in) OBC%rx_normal, OBC%ry_normal
OBC%oblique_BCs_exist_globally = (hdr(6) /= 0)
if (OBC%oblique_BCs_exist_globally) then      ! (open_boundary_register_restarts allocates them)
  allocate(OBC%rx_oblique_u(isd-1:ied,jsd:jed,nk), OBC%ry_oblique_u(isd-1:ied,jsd:jed,nk), OBC%cff_normal_u(isd-1:ied,jsd:jed,nk), &
           OBC%rx_oblique_v(isd:ied,jsd-1:jed,nk), OBC%ry_oblique_v(isd:ied,jsd-1:jed,nk), OBC%cff_normal_v(isd:ied,jsd-1:jed,nk))
  read(u_in) OBC%rx_oblique_u, OBC%ry_oblique_u, OBC%cff_normal_u, OBC%rx_oblique_v, OBC%ry_oblique_v, OBC%cff_normal_v
endif
allocate(seg_u(isd-1:ied,jsd:jed), seg_v(isd:ied,jsd-1:jed), OBC%segnum_u(isd-1:ied,jsd:jed), OBC%segnum_v(isd:ied,jsd-1:jed))
read(u_in) seg_u, seg_v
OBC%segnum_u(:,:) = seg_u(:,:) ; OBC%segnum_v(:,:) = seg_v(:,:)
allocate(OBC%segment(nseg))
! per segment: [direction, open, specified, on_pe, is_E_or_W, is_N_or_S, IsdB, IedB, JsdB, JedB, isd, ied, jsd, jed, Flather, radiation, gradient, nudged,
! oblique, radiation_tan, radiation_grad, oblique_tan, oblique_grad, nudged_tan + 2 nudged_grad], the two nudging timescales, and on the PE
! normal_vel, nudged_normal_vel (nudged), tangential_vel, tangential_grad (a _tan or _grad form), nudged_tangential_vel (nudged_tan),
! nudged_tangential_grad (nudged_grad); the other arrays as allocate_OBC_segment_data gives them (:3618-3707)
do m=1,nseg
  read(u_in) sflags, tscale
  OBC%segment(m)%direction = sflags(1) ; OBC%segment(m)%open = (sflags(2) /= 0) ; OBC%segment(m)%specified = (sflags(3) /= 0)
  OBC%segment(m)%on_pe = (sflags(4) /= 0) ; OBC%segment(m)%is_E_or_W = (sflags(5) /= 0) ; OBC%segment(m)%is_N_or_S = (sflags(6) /= 0)
  OBC%segment(m)%HI%IsdB = sflags(7) ; OBC%segment(m)%HI%IedB = sflags(8) ; OBC%segment(m)%HI%JsdB = sflags(9) ; OBC%segment(m)%HI%JedB = sflags(10)
  OBC%segment(m)%HI%isd = sflags(11) ; OBC%segment(m)%HI%ied = sflags(12) ; OBC%segment(m)%HI%jsd = sflags(13) ; OBC%segment(m)%HI%jed = sflags(14)
  OBC%segment(m)%Flather = (sflags(15) /= 0) ; OBC%segment(m)%radiation = (sflags(16) /= 0) ; OBC%segment(m)%gradient = (sflags(17) /= 0)
  OBC%segment(m)%nudged = (sflags(18) /= 0) ; OBC%segment(m)%oblique = (sflags(19) /= 0)
  OBC%segment(m)%radiation_tan = (sflags(20) /= 0) ; OBC%segment(m)%radiation_grad = (sflags(21) /= 0)
  OBC%segment(m)%oblique_tan = (sflags(22) /= 0) ; OBC%segment(m)%oblique_grad = (sflags(23) /= 0)
  OBC%segment(m)%nudged_tan = (iand(sflags(24), 1) /= 0) ; OBC%segment(m)%nudged_grad = (iand(sflags(24), 2) /= 0)
  OBC%segment(m)%Velocity_nudging_timescale_in = tscale(1) ; OBC%segment(m)%Velocity_nudging_timescale_out = tscale(2)
  if (OBC%segment(m)%open .and. OBC%segment(m)%is_E_or_W) OBC%open_u_BCs_exist_globally = .true.
  if (OBC%segment(m)%open .and. OBC%segment(m)%is_N_or_S) OBC%open_v_BCs_exist_globally = .true.
  if (OBC%segment(m)%radiation) OBC%radiation_BCs_exist_globally = .true.
  if (.not.OBC%segment(m)%on_pe) cycle
  if (OBC%segment(m)%is_E_or_W) then
    i0 = OBC%segment(m)%HI%IsdB ; i1 = OBC%segment(m)%HI%IedB ; j0 = OBC%segment(m)%HI%jsd ; j1 = OBC%segment(m)%HI%jed
  else
    i0 = OBC%segment(m)%HI%isd ; i1 = OBC%segment(m)%HI%ied ; j0 = OBC%segment(m)%HI%JsdB ; j1 = OBC%segment(m)%HI%JedB
  endif
  tan_any = OBC%segment(m)%radiation_tan .or. OBC%segment(m)%nudged_tan .or. OBC%segment(m)%oblique_tan
  grad_any = OBC%segment(m)%radiation_grad .or. OBC%segment(m)%nudged_grad .or. OBC%segment(m)%oblique_grad
  allocate(OBC%segment(m)%normal_vel(i0:i1,j0:j1,nk), OBC%segment(m)%normal_trans(i0:i1,j0:j1,nk), source=0.0)
  allocate(OBC%segment(m)%normal_vel_bt(i0:i1,j0:j1), OBC%segment(m)%SSH(i0:i1,j0:j1), source=0.0)
  read(u_in) OBC%segment(m)%normal_vel
  if (OBC%segment(m)%nudged) then
    allocate(OBC%segment(m)%nudged_normal_vel(i0:i1,j0:j1,nk)) ; read(u_in) OBC%segment(m)%nudged_normal_vel
  endif
  if (OBC%segment(m)%radiation .and. OBC%segment(m)%is_E_or_W) allocate(OBC%segment(m)%rx_norm_rad(i0:i1,j0:j1,nk), source=0.0)
  if (OBC%segment(m)%radiation .and. OBC%segment(m)%is_N_or_S) allocate(OBC%segment(m)%ry_norm_rad(i0:i1,j0:j1,nk), source=0.0)
  if (OBC%segment(m)%oblique) then
    allocate(OBC%segment(m)%rx_norm_obl(i0:i1,j0:j1,nk), OBC%segment(m)%ry_norm_obl(i0:i1,j0:j1,nk), OBC%segment(m)%cff_normal(i0:i1,j0:j1,nk), source=0.0)
    if (OBC%segment(m)%is_E_or_W) then
      allocate(OBC%segment(m)%grad_normal(OBC%segment(m)%HI%JsdB:OBC%segment(m)%HI%JedB,2,nk), source=0.0)
      if (OBC%segment(m)%oblique_tan) allocate(OBC%segment(m)%grad_tan(j0-1:j1+1,2,nk), source=0.0)
      if (OBC%segment(m)%oblique_grad) allocate(OBC%segment(m)%grad_gradient(j0:j1,2,nk), source=0.0)
    else
      allocate(OBC%segment(m)%grad_normal(OBC%segment(m)%HI%IsdB:OBC%segment(m)%HI%IedB,2,nk), source=0.0)
      if (OBC%segment(m)%oblique_tan) allocate(OBC%segment(m)%grad_tan(i0-1:i1+1,2,nk), source=0.0)
      if (OBC%segment(m)%oblique_grad) allocate(OBC%segment(m)%grad_gradient(i0:i1,2,nk), source=0.0)
    endif
  endif
  i0 = OBC%segment(m)%HI%IsdB ; i1 = OBC%segment(m)%HI%IedB ; j0 = OBC%segment(m)%HI%JsdB ; j1 = OBC%segment(m)%HI%JedB
  if (tan_any .or. grad_any) then
    allocate(OBC%segment(m)%tangential_vel(i0:i1,j0:j1,nk), OBC%segment(m)%tangential_grad(i0:i1,j0:j1,nk))
    read(u_in) OBC%segment(m)%tangential_vel, OBC%segment(m)%tangential_grad
  endif
  if (OBC%segment(m)%nudged_tan) then
    allocate(OBC%segment(m)%nudged_tangential_vel(i0:i1,j0:j1,nk)) ; read(u_in) OBC%segment(m)%nudged_tangential_vel
  endif
  if (OBC%segment(m)%nudged_grad) then
    allocate(OBC%segment(m)%nudged_tangential_grad(i0:i1,j0:j1,nk)) ; read(u_in) OBC%segment(m)%nudged_tangential_grad
  endif
enddo
close(u_in)

do n=1,ncall
  call radiation_open_bdry_conds(OBC, u_new, u_old, v_new, v_old, G, GV, US, dt)
  if (n < ncall) then      ! the next call sees this one's result as the old velocities and another increment on top of it
    u_old(:,:,:) = u_new(:,:,:) ; v_old(:,:,:) = v_new(:,:,:)
    u_new(:,:,:) = 0.5*u_new(:,:,:) + 0.01 ; v_new(:,:,:) = 0.5*v_new(:,:,:) - 0.01
  endif
enddo

open(newunit=u_out, file=trim(f_out), access="stream", form="unformatted", status="replace")
write(u_out) u_new, v_new, OBC%rx_normal, OBC%ry_normal
if (OBC%oblique_BCs_exist_globally) &
  write(u_out) OBC%rx_oblique_u, OBC%ry_oblique_u, OBC%cff_normal_u, OBC%rx_oblique_v, OBC%ry_oblique_v, OBC%cff_normal_v
do m=1,nseg ; if (OBC%segment(m)%on_pe) then
  write(u_out) OBC%segment(m)%normal_vel
  if (allocated(OBC%segment(m)%tangential_vel)) write(u_out) OBC%segment(m)%tangential_vel, OBC%segment(m)%tangential_grad
endif ; enddo
close(u_out)
write(*,'(a)') "ref_rad_driver ok"

contains

!> update_segment_tracer_reservoirs(G, GV, uhr, vhr, h, OBC, dt, Reg) on a file: [ni, nj, nk, halo, segments, tracers, -, -], [H_subroundoff, dt],
!! mask2dT, dyCu, dxCv, uhr, vhr, h, the tracers, then per segment [direction, on_pe, is_E_or_W, is_N_or_S, IsdB, IedB, JsdB, JedB, isd, ied, jsd,
!! jed], [Tr_InvLscale_in, Tr_InvLscale_out] and on the PE the number of registry entries, per entry [ntr_index, has a reservoir, has length-scale
!! factors], [resrv_lfac_in, resrv_lfac_out], and tres, t in the layout of normal_vel.  The reservoirs go to the output file.
subroutine reservoirs(fi, fo)
  character(len=*), intent(in) :: fi, fo
  type(ocean_grid_type), target :: G
  type(verticalGrid_type) :: GV
  type(ocean_OBC_type), pointer :: OBC => NULL()
  type(tracer_registry_type), pointer :: Reg => NULL()
  integer(c_int32_t) :: hdr(8), sf(12), ef(3)
  integer(c_int32_t) :: nreg(1)
  integer :: ni, nj, nk, halo, ui, uo, isd, ied, jsd, jed, nseg, ntr, m, q, i0, i1, j0, j1
  real :: sc(2), ls(2), lf(2)
  real, allocatable, dimension(:,:,:) :: uhr, vhr, h
  real, allocatable, target, dimension(:,:,:,:) :: tr
  open(newunit=ui, file=fi, access="stream", form="unformatted", status="old")
  read(ui) hdr
  ni = hdr(1) ; nj = hdr(2) ; nk = hdr(3) ; halo = hdr(4) ; nseg = hdr(5) ; ntr = hdr(6)
  isd = 1 ; ied = ni + 2*halo ; jsd = 1 ; jed = nj + 2*halo
  G%isd = isd ; G%ied = ied ; G%jsd = jsd ; G%jed = jed ; G%IsdB = isd-1 ; G%IedB = ied ; G%JsdB = jsd-1 ; G%JedB = jed
  G%isc = isd+halo ; G%iec = ied-halo ; G%jsc = jsd+halo ; G%jec = jed-halo ; G%ke = nk ; GV%ke = nk
  read(ui) sc
  GV%H_subroundoff = sc(1)
  allocate(G%mask2dT(isd:ied,jsd:jed), G%dyCu(isd-1:ied,jsd:jed), G%dxCv(isd:ied,jsd-1:jed))
  read(ui) G%mask2dT, G%dyCu, G%dxCv
  allocate(uhr(isd-1:ied,jsd:jed,nk), vhr(isd:ied,jsd-1:jed,nk), h(isd:ied,jsd:jed,nk), tr(isd:ied,jsd:jed,nk,ntr))
  read(ui) uhr, vhr, h, tr
  allocate(Reg) ; Reg%ntr = ntr
  do m=1,ntr ; Reg%Tr(m)%t => tr(:,:,:,m) ; enddo
  allocate(OBC) ; OBC%number_of_segments = nseg ; OBC%OBC_pe = .true. ; OBC%ke = nk ; OBC%ntr = ntr
  allocate(OBC%segment(nseg))
  do m=1,nseg
    read(ui) sf, ls
    OBC%segment(m)%direction = sf(1) ; OBC%segment(m)%on_pe = (sf(2) /= 0) ; OBC%segment(m)%is_E_or_W = (sf(3) /= 0) ; OBC%segment(m)%is_N_or_S = (sf(4) /= 0)
    OBC%segment(m)%HI%IsdB = sf(5) ; OBC%segment(m)%HI%IedB = sf(6) ; OBC%segment(m)%HI%JsdB = sf(7) ; OBC%segment(m)%HI%JedB = sf(8)
    OBC%segment(m)%HI%isd = sf(9) ; OBC%segment(m)%HI%ied = sf(10) ; OBC%segment(m)%HI%jsd = sf(11) ; OBC%segment(m)%HI%jed = sf(12)
    OBC%segment(m)%Tr_InvLscale_in = ls(1) ; OBC%segment(m)%Tr_InvLscale_out = ls(2)
    if (.not.OBC%segment(m)%on_pe) cycle
    if (OBC%segment(m)%is_E_or_W) then
      i0 = sf(5) ; i1 = sf(6) ; j0 = sf(11) ; j1 = sf(12)
    else
      i0 = sf(9) ; i1 = sf(10) ; j0 = sf(7) ; j1 = sf(8)
    endif
    read(ui) nreg
    if (nreg(1) <= 0) cycle
    allocate(OBC%segment(m)%tr_Reg) ; OBC%segment(m)%tr_Reg%ntseg = nreg(1)
    allocate(OBC%segment(m)%field(nreg(1)))
    do q=1,nreg(1)
      read(ui) ef, lf
      OBC%segment(m)%tr_Reg%Tr(q)%ntr_index = ef(1)
      OBC%segment(m)%tr_Reg%Tr(q)%fd_index = -1
      if (ef(3) /= 0) then      ! the field's factors on the reservoir length scales (segment%field(fd_index)%resrv_lfac_in | _out)
        OBC%segment(m)%tr_Reg%Tr(q)%fd_index = q
        OBC%segment(m)%field(q)%resrv_lfac_in = lf(1) ; OBC%segment(m)%field(q)%resrv_lfac_out = lf(2)
      endif
      if (ef(2) /= 0) then
        allocate(OBC%segment(m)%tr_Reg%Tr(q)%tres(i0:i1,j0:j1,nk), OBC%segment(m)%tr_Reg%Tr(q)%t(i0:i1,j0:j1,nk))
        read(ui) OBC%segment(m)%tr_Reg%Tr(q)%tres, OBC%segment(m)%tr_Reg%Tr(q)%t
      endif
    enddo
  enddo
  close(ui)
  call update_segment_tracer_reservoirs(G, GV, uhr, vhr, h, OBC, sc(2), Reg)
  open(newunit=uo, file=fo, access="stream", form="unformatted", status="replace")
  do m=1,nseg ; if (OBC%segment(m)%on_pe) then ; if (associated(OBC%segment(m)%tr_Reg)) then
    do q=1,OBC%segment(m)%tr_Reg%ntseg
      if (allocated(OBC%segment(m)%tr_Reg%Tr(q)%tres)) write(uo) OBC%segment(m)%tr_Reg%Tr(q)%tres
    enddo
  endif ; endif ; enddo
  close(uo)
end subroutine reservoirs
end program ref_rad_driver
