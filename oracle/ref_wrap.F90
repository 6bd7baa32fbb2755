!> bind(C) entry points around the reference routines that compile from their own source files with
!! no stand-ins (src/ALE/PLM_functions.F90, PCM_functions.F90: no `use` of any other module).
!! This file is ours (a caller of the reference, not a copy of it); the reference sources are
!! compiled where they lie under /root/reference by oracle/build_ref.sh into oracle/_ref/.
module mom6_ref_wrap
use, intrinsic :: iso_c_binding
use PLM_functions, only : PLM_reconstruction, PLM_boundary_extrapolation, PLM_slope_wa, &
                          PLM_monotonized_slope, PLM_extrapolate_slope
use PCM_functions, only : PCM_reconstruction
implicit none
contains

subroutine ref_plm_reconstruction(n, h, u, E, coef, h_neglect, extrapolate) bind(c, name="ref_plm_reconstruction")
  integer(c_int), value :: n
  real(c_double), intent(in) :: h(n), u(n)
  real(c_double), intent(inout) :: E(n,2), coef(n,2)
  real(c_double), value :: h_neglect
  integer(c_int), value :: extrapolate
  call PLM_reconstruction(n, h, u, E, coef, h_neglect)
  if (extrapolate /= 0) call PLM_boundary_extrapolation(n, h, u, E, coef, h_neglect)
end subroutine ref_plm_reconstruction

subroutine ref_pcm_reconstruction(n, u, E, coef) bind(c, name="ref_pcm_reconstruction")
  integer(c_int), value :: n
  real(c_double), intent(in) :: u(n)
  real(c_double), intent(inout) :: E(n,2), coef(n,1)
  call PCM_reconstruction(n, u, E, coef)
end subroutine ref_pcm_reconstruction

function ref_plm_slope_wa(h_l, h_c, h_r, h_neglect, u_l, u_c, u_r) bind(c, name="ref_plm_slope_wa") result(s)
  real(c_double), value :: h_l, h_c, h_r, h_neglect, u_l, u_c, u_r
  real(c_double) :: s
  s = PLM_slope_wa(h_l, h_c, h_r, h_neglect, u_l, u_c, u_r)
end function ref_plm_slope_wa

function ref_plm_monotonized_slope(u_l, u_c, u_r, s_l, s_c, s_r) bind(c, name="ref_plm_monotonized_slope") result(s)
  real(c_double), value :: u_l, u_c, u_r, s_l, s_c, s_r
  real(c_double) :: s
  s = PLM_monotonized_slope(u_l, u_c, u_r, s_l, s_c, s_r)
end function ref_plm_monotonized_slope

function ref_plm_extrapolate_slope(h_l, h_c, h_neglect, u_l, u_c) bind(c, name="ref_plm_extrapolate_slope") result(s)
  real(c_double), value :: h_l, h_c, h_neglect, u_l, u_c
  real(c_double) :: s
  s = PLM_extrapolate_slope(h_l, h_c, h_neglect, u_l, u_c)
end function ref_plm_extrapolate_slope

end module mom6_ref_wrap
