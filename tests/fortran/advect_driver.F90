!> Fortran host driver for the C ABI: builds a small analytic problem in plain Fortran arrays (whole
!! allocations, symmetric-memory index ranges exactly as a MOM6 caller has them), calls
!! mom6hip_advect_tracer through mom6hip_c_api with HOST pointers, and writes the updated tracers as
!! raw little-endian fp64 to the file named on the command line (and its inputs to <file>.in).
!! tests/test_fortran_abi.py runs the oracle on those inputs and compares bit for bit.
program advect_driver
use, intrinsic :: iso_c_binding
use mom6hip_c_api
implicit none

integer, parameter :: ni = 20, nj = 12, nk = 3, halo = 4, ntr = 2
integer, parameter :: isd = 1, ied = ni + 2*halo, jsd = 1, jed = nj + 2*halo
integer, parameter :: isc = 1 + halo, iec = halo + ni, jsc = 1 + halo, jec = halo + nj
real(c_double), target :: areaT(isd:ied,jsd:jed), IareaT(isd:ied,jsd:jed), mask2dT(isd:ied,jsd:jed)
real(c_double), target :: mask2dCu(isd-1:ied,jsd:jed), mask2dCv(isd:ied,jsd-1:jed)
real(c_double), target :: h_end(isd:ied,jsd:jed,nk), uhtr(isd-1:ied,jsd:jed,nk), vhtr(isd:ied,jsd-1:jed,nk)
real(c_double), target :: t1(isd:ied,jsd:jed,nk), t2(isd:ied,jsd:jed,nk), vol0(isd:ied,jsd:jed,nk)
type(mom6hip_grid_t) :: cg
type(mom6hip_tracer_advect_cs_t) :: ccs
type(mom6hip_advect_stats_t) :: stats
type(c_ptr) :: ctx, tr(ntr)
integer :: i, j, k, rc, ig, u
character(len=512) :: fname
real(c_double) :: x, y, c

call get_command_argument(1, fname)

! metrics: doubly periodic box, unit masks, mildly varying areas
do j=jsd,jed ; do i=isd,ied
  areaT(i,j) = 1.0d6 * (1.0d0 + 0.1d0*real(modulo(j-jsc, nj), c_double)/real(nj, c_double))
  IareaT(i,j) = 1.0d0 / areaT(i,j)
  mask2dT(i,j) = 1.0d0
enddo ; enddo
mask2dCu(:,:) = 1.0d0 ; mask2dCv(:,:) = 1.0d0

h_end(:,:,:) = 0.0d0 ; uhtr(:,:,:) = 0.0d0 ; vhtr(:,:,:) = 0.0d0 ; t1(:,:,:) = 0.0d0 ; t2(:,:,:) = 0.0d0
vol0(:,:,:) = 0.0d0
do k=1,nk ; do j=jsc,jec ; do i=isc,iec
  vol0(i,j,k) = areaT(i,j) * (10.0d0 + real(k, c_double))
enddo ; enddo ; enddo
! transports: fractions of the cell volume, periodic in both directions
do k=1,nk ; do j=jsc,jec ; do i=isc-1,iec
  ig = modulo(i-isc, ni)
  x = real(ig, c_double)/real(ni, c_double) ; y = real(j-jsc, c_double)/real(nj, c_double)
  c = 0.125d0*(1.0d0 - 2.0d0*x)*(0.5d0 + y) * (1.0d0 + 0.25d0*real(k, c_double))
  uhtr(i,j,k) = c * 1.0d6 * (10.0d0 + real(k, c_double))
enddo ; enddo ; enddo
do k=1,nk ; do j=jsc-1,jec ; do i=isc,iec
  x = real(i-isc, c_double)/real(ni, c_double) ; y = real(modulo(j-jsc, nj), c_double)/real(nj, c_double)
  c = 0.0625d0*(x - 0.5d0)*(1.0d0 - y)
  vhtr(i,j,k) = c * 1.0d6 * (10.0d0 + real(k, c_double))
enddo ; enddo ; enddo
do k=1,nk ; do j=jsc,jec ; do i=isc,iec
  h_end(i,j,k) = (vol0(i,j,k) - ((uhtr(i,j,k)-uhtr(i-1,j,k)) + (vhtr(i,j,k)-vhtr(i,j-1,k)))) * IareaT(i,j)
  x = real(i-isc, c_double)/real(ni, c_double) ; y = real(j-jsc, c_double)/real(nj, c_double)
  t1(i,j,k) = 10.0d0 + 4.0d0*x*(1.0d0-x) + y + 0.5d0*real(k, c_double)
  t2(i,j,k) = 0.0d0
  if (x > 0.25d0 .and. x < 0.6d0 .and. y > 0.2d0 .and. y < 0.7d0) t2(i,j,k) = 1.0d0
enddo ; enddo ; enddo

cg%isc = isc ; cg%iec = iec ; cg%jsc = jsc ; cg%jec = jec
cg%isd = isd ; cg%ied = ied ; cg%jsd = jsd ; cg%jed = jed
cg%nk = nk ; cg%symmetric = 1 ; cg%reentrant_x = 1 ; cg%reentrant_y = 1 ; cg%first_direction = 0
cg%tripolar_n = 0
cg%Angstrom_H = 1.0d-10 ; cg%H_subroundoff = 1.0d-30 ; cg%dZ_subroundoff = 1.0d-30
cg%H_to_Z = 1.0d0 ; cg%Z_to_H = 1.0d0 ; cg%g_Earth = 9.8d0 ; cg%Rho0 = 1035.0d0
cg%reserved1(:) = 0.0d0
cg%mask2dT = c_loc(mask2dT) ; cg%areaT = c_loc(areaT) ; cg%IareaT = c_loc(IareaT)
cg%dxT = c_null_ptr ; cg%dyT = c_null_ptr ; cg%IdxT = c_null_ptr ; cg%IdyT = c_null_ptr ; cg%bathyT = c_null_ptr
cg%mask2dCu = c_loc(mask2dCu) ; cg%dxCu = c_null_ptr ; cg%dyCu = c_null_ptr ; cg%dy_Cu = c_null_ptr
cg%IdxCu = c_null_ptr ; cg%IdyCu = c_null_ptr ; cg%areaCu = c_null_ptr ; cg%IareaCu = c_null_ptr
cg%mask2dCv = c_loc(mask2dCv) ; cg%dxCv = c_null_ptr ; cg%dyCv = c_null_ptr ; cg%dx_Cv = c_null_ptr
cg%IdxCv = c_null_ptr ; cg%IdyCv = c_null_ptr ; cg%areaCv = c_null_ptr ; cg%IareaCv = c_null_ptr
cg%mask2dBu = c_null_ptr ; cg%dxBu = c_null_ptr ; cg%dyBu = c_null_ptr ; cg%areaBu = c_null_ptr
cg%IareaBu = c_null_ptr ; cg%CoriolisBu = c_null_ptr ; cg%IdxBu = c_null_ptr ; cg%IdyBu = c_null_ptr
cg%reserved2(:) = c_null_ptr

! inputs, for the checker
open(newunit=u, file=trim(fname)//".in", access="stream", form="unformatted", status="replace")
write(u) areaT, h_end, uhtr, vhtr, t1, t2
close(u)

rc = mom6hip_init(0)
if (rc == 0) rc = mom6hip_grid_create(cg, c_null_ptr, ctx)
if (rc /= 0) then
  write(0,'(a)') "advect_driver: "//mom6hip_error_string() ; stop 2
endif

ccs%dt = 900.0d0 ; ccs%scheme = MOM6HIP_ADV_PPM_H3 ; ccs%use_huynh_stencil_bug = 0
tr(1) = c_loc(t1) ; tr(2) = c_loc(t2)
rc = mom6hip_advect_tracer(ctx, c_loc(h_end), c_loc(uhtr), c_loc(vhtr), 3600.0d0, ccs, tr, c_null_ptr, &
                           int(ntr, c_int32_t), -1_c_int32_t, c_null_ptr, 0_c_int32_t, 0_c_int32_t, &
                           c_null_ptr, c_null_ptr, MOM6HIP_MEM_HOST, stats)
if (rc /= 0) then
  write(0,'(a)') "advect_driver: "//mom6hip_error_string() ; stop 3
endif
rc = mom6hip_grid_destroy(ctx)

open(newunit=u, file=trim(fname), access="stream", form="unformatted", status="replace")
write(u) t1, t2
close(u)
write(*,'(a,i0,a,i0)') "advect_driver ok iterations=", stats%iterations, " halo_updates=", stats%halo_updates
end program advect_driver
