!> bind(C) entry points around the reference routines that compile from their own source files with
!! no stand-ins (src/ALE/PLM_functions.F90, PCM_functions.F90, MOM_hybgen_remap.F90, src/framework/MOM_array_transform.F90: no
!! `use` of any other module; src/equation_of_state/MOM_EOS_UNESCO.F90 with MOM_EOS_base_type.F90, which it alone uses).
!! This file is ours (a caller of the reference, not a copy of it); the reference sources are
!! compiled where they lie under /root/reference by oracle/build_ref.sh into oracle/_ref/.
module mom6_ref_wrap
use, intrinsic :: iso_c_binding
use PLM_functions, only : PLM_reconstruction, PLM_boundary_extrapolation, PLM_slope_wa, &
                          PLM_monotonized_slope, PLM_extrapolate_slope
use PCM_functions, only : PCM_reconstruction
use MOM_array_transform, only : rotate_array, rotate_vector
use MOM_hybgen_remap, only : hybgen_plm_coefs, hybgen_ppm_coefs, hybgen_weno_coefs
use MOM_EOS_UNESCO, only : UNESCO_EOS
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

!> hybgen_plm_coefs / hybgen_ppm_coefs / hybgen_weno_coefs (src/ALE/MOM_hybgen_remap.F90:14, :91, :226) for one scalar field
subroutine ref_hybgen_plm(n, s, dp, slope, thin) bind(c, name="ref_hybgen_plm")
  integer(c_int), value :: n
  real(c_double), intent(in) :: s(n,1), dp(n)
  real(c_double), intent(inout) :: slope(n,1)
  real(c_double), value :: thin
  call hybgen_plm_coefs(s, dp, slope, n, 1, thin)
end subroutine ref_hybgen_plm

subroutine ref_hybgen_ppm(n, s, h, edges, thin) bind(c, name="ref_hybgen_ppm")
  integer(c_int), value :: n
  real(c_double), intent(in) :: s(n,1), h(n)
  real(c_double), intent(inout) :: edges(n,2,1)
  real(c_double), value :: thin
  call hybgen_ppm_coefs(s, h, edges, n, 1, thin)
end subroutine ref_hybgen_ppm

subroutine ref_hybgen_weno(n, s, h, edges, thin) bind(c, name="ref_hybgen_weno")
  integer(c_int), value :: n
  real(c_double), intent(in) :: s(n,1), h(n)
  real(c_double), intent(inout) :: edges(n,2,1)
  real(c_double), value :: thin
  call hybgen_weno_coefs(s, h, edges, n, 1, thin)
end subroutine ref_hybgen_weno

!> calculate_density_array (MOM_EOS_base_type.F90:229; with rho_ref when use_ref /= 0) and calculate_density_derivs_array (:331)
!! of the reference's UNESCO_EOS
subroutine ref_unesco(n, T, S, p, rho_ref, use_ref, rho, drho_dT, drho_dS) bind(c, name="ref_unesco")
  integer(c_int), value :: n, use_ref
  real(c_double), intent(in) :: T(n), S(n), p(n)
  real(c_double), value :: rho_ref
  real(c_double), intent(inout) :: rho(n), drho_dT(n), drho_dS(n)
  type(UNESCO_EOS) :: E
  if (use_ref /= 0) then
    call E%calculate_density_array(T, S, p, rho, 1, n, rho_ref=rho_ref)
  else
    call E%calculate_density_array(T, S, p, rho, 1, n)
  endif
  call E%calculate_density_derivs_array(T, S, p, drho_dT, drho_dS, 1, n)
end subroutine ref_unesco

!> calculate_spec_vol_array with spv_ref (MOM_EOS_base_type.F90:291) of the reference's UNESCO_EOS
subroutine ref_unesco_spv(n, T, S, p, spv_ref, spv) bind(c, name="ref_unesco_spv")
  integer(c_int), value :: n
  real(c_double), intent(in) :: T(n), S(n), p(n)
  real(c_double), value :: spv_ref
  real(c_double), intent(inout) :: spv(n)
  type(UNESCO_EOS) :: E
  call E%calculate_spec_vol_array(T, S, p, spv, 1, n, spv_ref=spv_ref)
end subroutine ref_unesco_spv

!> PLM_reconstruction of ncol columns of n layers (timing the reference's code against the restatement: tools/calibrate_ref.py)
subroutine ref_plm_batch(ncol, n, h, u, E, coef, h_neglect) bind(c, name="ref_plm_batch")
  integer(c_int), value :: ncol, n
  real(c_double), intent(in) :: h(n,ncol), u(n,ncol)
  real(c_double), intent(inout) :: E(n,2,ncol), coef(n,2,ncol)
  real(c_double), value :: h_neglect
  integer :: c
  do c = 1, ncol
    call PLM_reconstruction(n, h(:,c), u(:,c), E(:,:,c), coef(:,:,c), h_neglect)
  enddo
end subroutine ref_plm_batch

!> rotate_array (src/framework/MOM_array_transform.F90:26) of a 3-D field: A_in(m,n,nk) -> A (n,m,nk for odd turns)
subroutine ref_rotate_array(m, n, nk, A_in, turns, A) bind(c, name="ref_rotate_array")
  integer(c_int), value :: m, n, nk, turns
  real(c_double), intent(in) :: A_in(m,n,nk)
  real(c_double), intent(inout) :: A(*)
  real(c_double), allocatable :: R(:,:,:)
  if (modulo(turns, 2) /= 0) then ; allocate(R(n,m,nk)) ; else ; allocate(R(m,n,nk)) ; endif
  call rotate_array(A_in, turns, R)
  A(1:size(R)) = reshape(R, (/ size(R) /))
end subroutine ref_rotate_array

!> rotate_vector (:54) of a C-grid vector field on symmetric memory: u(mu,nu,nk), v(mv,nv,nk) -> the turned pair
subroutine ref_rotate_vector(mu, nu, mv, nv, nk, u_in, v_in, turns, u, v) bind(c, name="ref_rotate_vector")
  integer(c_int), value :: mu, nu, mv, nv, nk, turns
  real(c_double), intent(in) :: u_in(mu,nu,nk), v_in(mv,nv,nk)
  real(c_double), intent(inout) :: u(*), v(*)
  real(c_double), allocatable :: Ru(:,:,:), Rv(:,:,:)
  if (modulo(turns, 2) /= 0) then
    allocate(Ru(nv,mv,nk)) ; allocate(Rv(nu,mu,nk))      ! the turned u points are the v points and the other way round
  else
    allocate(Ru(mu,nu,nk)) ; allocate(Rv(mv,nv,nk))
  endif
  call rotate_vector(u_in, v_in, turns, Ru, Rv)
  u(1:size(Ru)) = reshape(Ru, (/ size(Ru) /))
  v(1:size(Rv)) = reshape(Rv, (/ size(Rv) /))
end subroutine ref_rotate_vector

end module mom6_ref_wrap
