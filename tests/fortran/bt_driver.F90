!> Drives the MOM_barotropic shim the way the split RK2 step calls the reference module: barotropic_init from a parameter
!! list (DTBT > 0, so the time step is the file's), btcalc with the BT_cont face thicknesses, bt_mass_source, btstep with
!! every optional pointer of the RK2 call (:655) on plain host arrays.  tests/test_fortran_abi.py writes the input file and
!! compares the output with the oracle bit for bit.   Usage: bt_driver <input file> <output file>
!! With -DREFERENCE_KERNELS the same program drives the reference's own MOM_barotropic.F90 compiled against the stand-ins
!! (tests/test_reference_kernels.py; the library's glue is not linked).
program bt_driver
use, intrinsic :: iso_c_binding
use MOM_barotropic,    only : barotropic_CS, barotropic_init, btcalc, bt_mass_source, btstep, barotropic_end, &
                              register_barotropic_restarts, barotropic_get_tav
use MOM_diag_mediator, only : diag_ctrl, time_type
use MOM_domains,       only : MOM_domain_type
use MOM_file_parser,   only : param_file_type, param_set
use MOM_forcing_type,  only : mech_forcing
use MOM_grid,          only : ocean_grid_type
use MOM_hor_index,     only : hor_index_type
use MOM_open_boundary, only : ocean_OBC_type
use MOM_restart,       only : MOM_restart_CS
use MOM_unit_scaling,  only : unit_scale_type
use MOM_variables,     only : BT_cont_type, accel_diag_ptrs, alloc_BT_cont_type
use MOM_verticalGrid,  only : verticalGrid_type
#ifndef REFERENCE_KERNELS
use mom6hip_MOM_glue,  only : mom6hip_shared_context_end
#endif
implicit none

type(ocean_grid_type), target :: G
type(verticalGrid_type) :: GV
type(unit_scale_type) :: US
type(hor_index_type) :: HI
type(param_file_type) :: pf
type(time_type), target :: Time
type(diag_ctrl), target :: diag
type(MOM_restart_CS) :: restart_CS
type(barotropic_CS) :: CS
type(mech_forcing) :: forces
type(accel_diag_ptrs), pointer :: ADp => NULL()
type(ocean_OBC_type), pointer :: OBC => NULL()
type(BT_cont_type), pointer :: BT => NULL()
integer(c_int32_t) :: hdr(8)
integer :: ni, nj, nk, halo, u_in, u_out, isd, ied, jsd, jed
real :: scal(7), dt, dtbt
logical :: calc_dtbt
character(len=32) :: str
real, allocatable, dimension(:,:,:) :: u, v, h, bcu, bcv, pbce, vru, vrv, alu, alv
real, pointer, dimension(:,:,:) :: uh0 => NULL(), vh0 => NULL(), u_uh0 => NULL(), v_vh0 => NULL()
real, pointer, dimension(:,:) :: eta_PF_start => NULL(), taux_bot => NULL(), tauy_bot => NULL()
real, allocatable, dimension(:,:) :: eta, eta_PF, eta_out, uhbtav, vhbtav, etaav, SpV, ubtav, vbtav
character(len=512) :: f_in, f_out, f_obc
integer(c_int32_t) :: oflags(8), sflags(20), gflags(8)
integer(c_int32_t), allocatable :: seg_u(:,:), seg_v(:,:)
integer :: u_obc, nseg, m, i0, i1, j0, j1
real :: oscal(2)

call get_command_argument(1, f_in) ; call get_command_argument(2, f_out)
f_obc = "" ; if (command_argument_count() >= 3) call get_command_argument(3, f_obc)
open(newunit=u_in, file=trim(f_in), access="stream", form="unformatted", status="old")
read(u_in) hdr
ni = hdr(1) ; nj = hdr(2) ; nk = hdr(3) ; halo = hdr(4)
isd = 1 ; ied = ni + 2*halo ; jsd = 1 ; jed = nj + 2*halo
G%isd = isd ; G%ied = ied ; G%jsd = jsd ; G%jed = jed ; G%IsdB = isd-1 ; G%IedB = ied ; G%JsdB = jsd-1 ; G%JedB = jed
G%isc = isd+halo ; G%iec = ied-halo ; G%jsc = jsd+halo ; G%jec = jed-halo
G%IscB = G%isc-1 ; G%IecB = G%iec ; G%JscB = G%jsc-1 ; G%JecB = G%jec ; G%ke = nk ; GV%ke = nk
G%first_direction = hdr(7)
HI%isd = isd ; HI%ied = ied ; HI%jsd = jsd ; HI%jed = jed ; HI%IsdB = isd-1 ; HI%IedB = ied ; HI%JsdB = jsd-1 ; HI%JedB = jed
HI%isc = G%isc ; HI%iec = G%iec ; HI%jsc = G%jsc ; HI%jec = G%jec
HI%IscB = G%IscB ; HI%IecB = G%IecB ; HI%JscB = G%JscB ; HI%JecB = G%JecB
allocate(G%Domain)
G%Domain%reentrant(1) = (hdr(5) /= 0) ; G%Domain%reentrant(2) = (hdr(6) /= 0)
G%Domain%nihalo = halo ; G%Domain%njhalo = halo ; G%Domain%niglobal = ni ; G%Domain%njglobal = nj
read(u_in) scal, dt, dtbt
GV%Angstrom_H = scal(1) ; GV%H_subroundoff = scal(2) ; GV%dZ_subroundoff = scal(3) ; GV%H_to_Z = scal(4) ; GV%Z_to_H = scal(5)
GV%g_Earth = scal(6) ; GV%Rho0 = scal(7) ; GV%RZ_to_H = GV%Z_to_H / GV%Rho0 ; GV%H_to_RZ = GV%H_to_Z * GV%Rho0
allocate(GV%g_prime(nk+1)) ; GV%g_prime(:) = 0.0 ; GV%g_prime(1) = GV%g_Earth

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

allocate(u(isd-1:ied,jsd:jed,nk), v(isd:ied,jsd-1:jed,nk), h(isd:ied,jsd:jed,nk), eta(isd:ied,jsd:jed), bcu(isd-1:ied,jsd:jed,nk), &
         bcv(isd:ied,jsd-1:jed,nk), pbce(isd:ied,jsd:jed,nk), eta_PF(isd:ied,jsd:jed), vru(isd-1:ied,jsd:jed,nk), &
         vrv(isd:ied,jsd-1:jed,nk), uh0(isd-1:ied,jsd:jed,nk), vh0(isd:ied,jsd-1:jed,nk))
allocate(forces%taux(isd-1:ied,jsd:jed), forces%tauy(isd:ied,jsd-1:jed))
call alloc_BT_cont_type(BT, isd, ied, jsd, jed, nk, alloc_faces=.true.)
read(u_in) u, v, h, eta, bcu, bcv, forces%taux, forces%tauy, pbce, eta_PF, vru, vrv, uh0, vh0
read(u_in) BT%FA_u_W0, BT%FA_u_WW, BT%FA_u_E0, BT%FA_u_EE, BT%uBT_WW, BT%uBT_EE
read(u_in) BT%FA_v_S0, BT%FA_v_SS, BT%FA_v_N0, BT%FA_v_NN, BT%vBT_SS, BT%vBT_NN, BT%h_u, BT%h_v
close(u_in)
allocate(u_uh0(isd-1:ied,jsd:jed,nk), v_vh0(isd:ied,jsd-1:jed,nk)) ; u_uh0 = u ; v_vh0 = v
allocate(alu(isd-1:ied,jsd:jed,nk), alv(isd:ied,jsd-1:jed,nk), eta_out(isd:ied,jsd:jed), uhbtav(isd-1:ied,jsd:jed), &
         vhbtav(isd:ied,jsd-1:jed), etaav(isd:ied,jsd:jed), SpV(isd:ied,jsd:jed), ubtav(isd-1:ied,jsd:jed), vbtav(isd:ied,jsd-1:jed))
alu = 0.0 ; alv = 0.0 ; eta_out = 0.0 ; uhbtav = 0.0 ; vhbtav = 0.0 ; etaav = 0.0 ; SpV = 1.0 ; ubtav = 0.0 ; vbtav = 0.0

! the open boundaries (a third argument): the file of dyn_driver.F90
if (len_trim(f_obc) > 0) then
  open(newunit=u_obc, file=trim(f_obc), access="stream", form="unformatted", status="old")
  allocate(OBC)
  read(u_obc) oflags, gflags, oscal
  nseg = oflags(1)
  OBC%number_of_segments = nseg ; OBC%OBC_pe = (oflags(2) /= 0)
  OBC%open_u_BCs_exist_globally = (oflags(3) /= 0) ; OBC%open_v_BCs_exist_globally = (oflags(4) /= 0)
  OBC%specified_u_BCs_exist_globally = (oflags(5) /= 0) ; OBC%specified_v_BCs_exist_globally = (oflags(6) /= 0)
  OBC%Flather_u_BCs_exist_globally = (oflags(7) /= 0) ; OBC%Flather_v_BCs_exist_globally = (oflags(8) /= 0)
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
  close(u_obc)
endif

allocate(ADp)      ! (MOM.F90 always hands btstep an allocated accel_diag_ptrs; its members stay unassociated)
call param_set(pf, "REENTRANT_X", merge("True ", "False", hdr(5) /= 0))
call param_set(pf, "REENTRANT_Y", merge("True ", "False", hdr(6) /= 0))
write(str, '(es24.16)') dtbt ; call param_set(pf, "DTBT", str)
do m = 4, command_argument_count()      ! further NAME=VALUE pairs of the parameter file
  call get_command_argument(m, f_obc)
  i0 = index(f_obc, "=")
  if (i0 > 1) call param_set(pf, f_obc(1:i0-1), trim(f_obc(i0+1:)))
enddo
call register_barotropic_restarts(HI, GV, US, pf, CS, restart_CS)
call barotropic_init(u, v, h, eta, Time, G, GV, US, pf, diag, CS, restart_CS, calc_dtbt, BT)
call barotropic_get_tav(CS, ubtav, vbtav, G, US)
call btcalc(h, G, GV, CS, BT%h_u, BT%h_v, OBC=OBC)
call bt_mass_source(h, eta, .true., G, GV, CS)
call btstep(u, v, eta, dt, bcu, bcv, forces, pbce, eta_PF, u, v, alu, alv, eta_out, uhbtav, vhbtav, G, GV, US, CS, &
            vru, vrv, SpV, ADp, OBC, BT, eta_PF_start, taux_bot, tauy_bot, uh0, vh0, u_uh0, v_vh0, etaav=etaav)

open(newunit=u_out, file=trim(f_out), access="stream", form="unformatted", status="replace")
write(u_out) alu, alv, eta_out, uhbtav, vhbtav, etaav, ubtav, vbtav
close(u_out)
call barotropic_end(CS)
#ifndef REFERENCE_KERNELS
call mom6hip_shared_context_end()
#endif
write(*,'(a,l2)') "bt_driver ok calc_dtbt=", calc_dtbt
end program bt_driver
