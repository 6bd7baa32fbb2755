!> Drives the reference's OWN reproducing_sum (src/framework/MOM_coms.F90:219, :318, compiled in place against the one-PE stand-in of its
!! infrastructure module, tests/fortran/stubs/mom6_stubs_coms_infra.F90) on a field written by tests/test_reference_kernels.py: the total and
!! the sums by layer, as reals and through EFP_to_real of the extended-fixed-point results, go to the output file.  Build container only.
!! Usage: ref_coms_driver <input file> <output file>
program ref_coms_driver
use, intrinsic :: iso_c_binding
use MOM_coms, only : reproducing_sum, EFP_type, EFP_to_real, EFP_real_diff
implicit none
integer(c_int32_t) :: hdr(8)
integer :: ni, nj, nk, ui, uo, k
real, allocatable :: a(:,:,:), sums(:), lay_r(:)
type(EFP_type) :: tot
type(EFP_type), allocatable :: lay(:)
real :: s3, s2, tot_r, d
character(len=512) :: f_in, f_out
call get_command_argument(1, f_in) ; call get_command_argument(2, f_out)
open(newunit=ui, file=trim(f_in), access="stream", form="unformatted", status="old")
read(ui) hdr      ! [array columns, array rows, layers, isr, ier, jsr, jer, -]
ni = hdr(1) ; nj = hdr(2) ; nk = hdr(3)
allocate(a(ni,nj,nk), sums(nk), lay(nk), lay_r(nk))
read(ui) a
close(ui)
s3 = reproducing_sum(a, hdr(4), hdr(5), hdr(6), hdr(7), sums=sums, EFP_sum=tot, EFP_lay_sums=lay)
s2 = reproducing_sum(a(:,:,1), hdr(4), hdr(5), hdr(6), hdr(7))
tot_r = EFP_to_real(tot)
do k=1,nk ; lay_r(k) = EFP_to_real(lay(k)) ; enddo
d = EFP_real_diff(tot, lay(1))      ! the whole minus the first layer, formed in extended fixed point
open(newunit=uo, file=trim(f_out), access="stream", form="unformatted", status="replace")
write(uo) s3, s2, tot_r, d, sums, lay_r
close(uo)
write(*,'(a)') "ref_coms_driver ok"
end program ref_coms_driver
