!> Drop-in replacement for module MOM_PressureForce_FV (src/core/MOM_PressureForce_FV.F90): PressureForce_FV_Bouss (:462)
!! and PressureForce_FV_init (:921) with the reference's dummy-argument lists, so MOM_PressureForce.F90 (`PressureForce`,
!! called by the split RK2 step at :548 and :795) compiles unchanged.  Provided: the analytic finite-volume pressure force
!! in Boussinesq mode with PLM reconstruction of T and S (RECONSTRUCT_FOR_PRESSURE, PRESSURE_RECONSTRUCTION_SCHEME = 1,
!! BOUNDARY_EXTRAPOLATION_PRESSURE, MASS_WEIGHT_IN_PRESSURE_GRADIENT, RHO_PGF_REF), the WRIGHT, UNESCO and LINEAR equations of state,
!! p_atm, pbce, eta, on the GPU through libmom6hip (mom6hip_pressureforce_fv_bouss, HOST memspace).  The equation of state
!! is opaque in MOM6 (EOS_type is private), so its selection is read from the parameter file the way
!! interpret_eos_selection does (MOM_EOS.F90:1474-1520).  Tides / SAL, the Stanley correction, PPM reconstruction,
!! USE_INACCURATE_PGF_RHO_ANOM stops with a FATAL error; PressureForce_FV_nonBouss (mom6hip_pressureforce_fv_nonbouss) serves a
!! non-Boussinesq vertical grid.
!!
!! Compiled INSIDE a MOM6 source tree in place of src/core/MOM_PressureForce_FV.F90; here against tests/fortran/stubs.
module MOM_PressureForce_FV

use, intrinsic :: iso_c_binding
use mom6hip_c_api
use mom6hip_MOM_glue,     only : mom6hip_shared_context, mom6hip_read_topology, mom6hip_read_eos, mom6hip_fatal_if
use MOM_ALE,              only : ALE_CS
use MOM_diag_mediator,    only : diag_ctrl, time_type
use MOM_error_handler,    only : MOM_error, MOM_mesg, FATAL
use MOM_file_parser,      only : get_param, log_version, param_file_type
use MOM_grid,             only : ocean_grid_type
use MOM_self_attr_load,   only : SAL_CS
use MOM_tidal_forcing,    only : tidal_forcing_CS
use MOM_unit_scaling,     only : unit_scale_type
use MOM_variables,        only : thermo_var_ptrs
use MOM_verticalGrid,     only : verticalGrid_type
implicit none ; private

#include <MOM_memory.h>

public PressureForce_FV_init
public PressureForce_FV_Bouss, PressureForce_FV_nonBouss
public PressureForce_FV_hip_struct      ! (GPU path only) for MOM_dynamics_split_RK2

!> Finite volume pressure gradient control structure (the members of the reference's, :36-80, that the provided form reads)
type, public :: PressureForce_FV_CS ; private
  logical :: initialized = .false.
  real    :: Rho0, GFS_scale
  logical :: useMassWghtInterp, reconstruct, boundary_extrap
  integer :: Recon_Scheme
  type(mom6hip_eos_t) :: eos       !< the equation of state, as read from the parameter file
  type(time_type), pointer :: Time => NULL()
  type(diag_ctrl), pointer :: diag => NULL()
end type PressureForce_FV_CS

contains

!> (GPU path only) The control structure, and what PressureForce_FV_Bouss takes from its other arguments to choose its branch, as
!! the library's structs.  use_EOS says whether tv%eqn_of_state is associated (eos is then the equation of state read at
!! initialisation).  GV must stay in place while ccs is in use (its Rlay and g_prime are referenced, not copied).
subroutine PressureForce_FV_hip_struct(CS, G, GV, tv, ALE_CSp, ccs, eos, use_EOS)
  type(PressureForce_FV_CS), intent(in) :: CS
  type(ocean_grid_type),   intent(in)  :: G
  type(verticalGrid_type), target, intent(in) :: GV
  type(thermo_var_ptrs),   intent(in)  :: tv
  type(ALE_CS),            pointer     :: ALE_CSp
  type(mom6hip_pressureforce_cs_t), intent(out) :: ccs
  type(mom6hip_eos_t),     intent(out) :: eos
  logical,                 intent(out) :: use_EOS
  if (.not.CS%initialized) call MOM_error(FATAL, "MOM_PressureForce_FV_Bouss: Module must be initialized before it is used.")
  use_EOS = associated(tv%eqn_of_state)
  ccs%Rho0 = CS%Rho0 ; ccs%GFS_scale = CS%GFS_scale ; ccs%Z_ref = G%Z_ref
  ccs%reconstruct = merge(1, 0, CS%reconstruct) ; ccs%Recon_Scheme = CS%Recon_Scheme
  ccs%boundary_extrap = merge(1, 0, CS%boundary_extrap) ; ccs%useMassWghtInterp = merge(1, 0, CS%useMassWghtInterp)
  ccs%use_ALE = merge(1, 0, associated(ALE_CSp)) ; ccs%nkmb = GV%nk_rho_varies ; ccs%P_Ref = tv%P_Ref
  ccs%Rlay = c_null_ptr ; if (allocated(GV%Rlay)) ccs%Rlay = c_loc(GV%Rlay)
  ccs%g_prime = c_null_ptr ; if (allocated(GV%g_prime)) ccs%g_prime = c_loc(GV%g_prime)
  eos = CS%eos
end subroutine PressureForce_FV_hip_struct

!> Same interface as the reference PressureForce_FV_nonBouss (:89).
subroutine PressureForce_FV_nonBouss(h, tv, PFu, PFv, G, GV, US, CS, ALE_CSp, p_atm, pbce, eta)
  type(ocean_grid_type),   intent(in)  :: G
  type(verticalGrid_type), intent(in)  :: GV
  type(unit_scale_type),   intent(in)  :: US
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  target, intent(in)  :: h
  type(thermo_var_ptrs),   intent(in)  :: tv
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(out) :: PFu
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(out) :: PFv
  type(PressureForce_FV_CS), intent(in) :: CS
  type(ALE_CS),            pointer     :: ALE_CSp
  real, dimension(:,:),    pointer     :: p_atm
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), target, optional, intent(out) :: pbce
  real, dimension(SZI_(G),SZJ_(G)),          target, optional, intent(out) :: eta

  type(mom6hip_pressureforce_cs_t) :: ccs
  type(c_ptr) :: p_patm, p_pbce, p_eta
  integer :: rc

  if (.not.CS%initialized) call MOM_error(FATAL, "MOM_PressureForce_FV_nonBouss: Module must be initialized before it is used.")
  if (.not.(associated(tv%T) .and. associated(tv%S))) call MOM_error(FATAL, "MOM_PressureForce_FV_nonBouss (HIP): "// &
       "the GPU path needs temperature and salinity (USE_EOS); the layered mode with GV%Rlay is not provided.")
  if (GV%nk_rho_varies > 0) call MOM_error(FATAL, "MOM_PressureForce_FV_nonBouss (HIP): a bulk mixed layer is not provided.")
  ccs%Rho0 = CS%Rho0 ; ccs%GFS_scale = CS%GFS_scale ; ccs%Z_ref = G%Z_ref
  ccs%reconstruct = merge(1, 0, CS%reconstruct) ; ccs%Recon_Scheme = CS%Recon_Scheme
  ccs%boundary_extrap = merge(1, 0, CS%boundary_extrap) ; ccs%useMassWghtInterp = merge(1, 0, CS%useMassWghtInterp)
  ccs%use_ALE = merge(1, 0, associated(ALE_CSp)) ; ccs%nkmb = 0 ; ccs%P_Ref = tv%P_Ref
  p_patm = c_null_ptr ; if (associated(p_atm)) p_patm = c_loc(p_atm)
  p_pbce = c_null_ptr ; if (present(pbce)) p_pbce = c_loc(pbce)
  p_eta = c_null_ptr ; if (present(eta)) p_eta = c_loc(eta)
  rc = mom6hip_pressureforce_fv_nonbouss(mom6hip_shared_context(G, GV), ccs, CS%eos, c_loc(h), c_loc(tv%T), c_loc(tv%S), p_patm, &
                                         real(GV%H_to_RZ, c_double), c_loc(PFu), c_loc(PFv), p_pbce, p_eta, MOM6HIP_MEM_HOST)
  call mom6hip_fatal_if(rc, "MOM_PressureForce_FV_nonBouss")
end subroutine PressureForce_FV_nonBouss

!> Same interface as the reference PressureForce_FV_Bouss (:462).  The branch is chosen as the reference chooses it (:559-562,
!! :746-789): with an equation of state, ALE and RECONSTRUCT_FOR_PRESSURE the PLM form; with an equation of state otherwise
!! int_density_dz (the analytic LINEAR / WRIGHT integrals), with the bulk mixed layer's replacement of light layers when
!! GV%nk_rho_varies > 0; without an equation of state the layered form with GV%Rlay and GV%g_prime.
subroutine PressureForce_FV_Bouss(h, tv, PFu, PFv, G, GV, US, CS, ALE_CSp, p_atm, pbce, eta)
  type(ocean_grid_type),   intent(in)  :: G
  type(verticalGrid_type), target, intent(in)  :: GV
  type(unit_scale_type),   intent(in)  :: US
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)),  target, intent(in)  :: h
  type(thermo_var_ptrs),   intent(in)  :: tv
  real, dimension(SZIB_(G),SZJ_(G),SZK_(GV)), target, intent(out) :: PFu
  real, dimension(SZI_(G),SZJB_(G),SZK_(GV)), target, intent(out) :: PFv
  type(PressureForce_FV_CS), target, intent(in) :: CS
  type(ALE_CS),            pointer     :: ALE_CSp
  real, dimension(:,:),    pointer     :: p_atm
  real, dimension(SZI_(G),SZJ_(G),SZK_(GV)), target, optional, intent(out) :: pbce
  real, dimension(SZI_(G),SZJ_(G)),          target, optional, intent(out) :: eta

  type(mom6hip_pressureforce_cs_t) :: ccs
  type(c_ptr) :: p_patm, p_pbce, p_eta, p_eos, p_T, p_S
  logical :: use_EOS
  integer :: rc

  if (.not.CS%initialized) call MOM_error(FATAL, "MOM_PressureForce_FV_Bouss: Module must be initialized before it is used.")
  use_EOS = associated(tv%eqn_of_state)
  if (use_EOS .and. .not.(associated(tv%T) .and. associated(tv%S))) call MOM_error(FATAL, "MOM_PressureForce_FV_Bouss (HIP): "// &
       "an equation of state needs tv%T and tv%S.")
  ccs%Rho0 = CS%Rho0 ; ccs%GFS_scale = CS%GFS_scale ; ccs%Z_ref = G%Z_ref
  ccs%reconstruct = merge(1, 0, CS%reconstruct) ; ccs%Recon_Scheme = CS%Recon_Scheme
  ccs%boundary_extrap = merge(1, 0, CS%boundary_extrap) ; ccs%useMassWghtInterp = merge(1, 0, CS%useMassWghtInterp)
  ccs%use_ALE = merge(1, 0, associated(ALE_CSp)) ; ccs%nkmb = GV%nk_rho_varies ; ccs%P_Ref = tv%P_Ref
  ccs%Rlay = c_null_ptr ; if (allocated(GV%Rlay)) ccs%Rlay = c_loc(GV%Rlay)
  ccs%g_prime = c_null_ptr ; if (allocated(GV%g_prime)) ccs%g_prime = c_loc(GV%g_prime)
  p_eos = c_null_ptr ; p_T = c_null_ptr ; p_S = c_null_ptr
  if (use_EOS) then ; p_eos = c_loc(CS%eos) ; p_T = c_loc(tv%T) ; p_S = c_loc(tv%S) ; endif
  p_patm = c_null_ptr ; if (associated(p_atm)) p_patm = c_loc(p_atm)
  p_pbce = c_null_ptr ; if (present(pbce)) p_pbce = c_loc(pbce)
  p_eta = c_null_ptr ; if (present(eta)) p_eta = c_loc(eta)
  rc = mom6hip_pressureforce_fv_bouss(mom6hip_shared_context(G, GV), ccs, p_eos, c_loc(h), p_T, p_S, p_patm, &
                                      c_loc(PFu), c_loc(PFv), p_pbce, p_eta, MOM6HIP_MEM_HOST)
  call mom6hip_fatal_if(rc, "MOM_PressureForce_FV_Bouss")
end subroutine PressureForce_FV_Bouss

!> Same interface as the reference PressureForce_FV_init (:921), same parameters and defaults.
subroutine PressureForce_FV_init(Time, G, GV, US, param_file, diag, CS, SAL_CSp, tides_CSp)
  type(time_type), target,    intent(in)    :: Time
  type(ocean_grid_type),      intent(in)    :: G
  type(verticalGrid_type),    intent(in)    :: GV
  type(unit_scale_type),      intent(in)    :: US
  type(param_file_type),      intent(in)    :: param_file
  type(diag_ctrl), target,    intent(inout) :: diag
  type(PressureForce_FV_CS),  intent(inout) :: CS
  type(SAL_CS),           intent(in), target, optional :: SAL_CSp
  type(tidal_forcing_CS), intent(in), target, optional :: tides_CSp
# include "version_variable.h"
  character(len=40)  :: mdl
  logical :: use_ALE, flag

  CS%initialized = .true.
  CS%diag => diag ; CS%Time => Time
  mdl = "MOM_PressureForce_FV"
  call log_version(param_file, mdl, version, "")
  call get_param(param_file, mdl, "RHO_PGF_REF", CS%Rho0, &
                 "The reference density that is subtracted off when calculating pressure gradient forces.", &
                 units="kg m-3", default=GV%Rho0*US%R_to_kg_m3, scale=US%kg_m3_to_R)
  call get_param(param_file, mdl, "TIDES", flag, "If true, apply tidal momentum forcing.", default=.false.)
  call refuse(flag, "TIDES")
  call get_param(param_file, mdl, "CALCULATE_SAL", flag, "If true, calculate self-attraction and loading.", default=.false.)
  call refuse(flag, "CALCULATE_SAL")
  call get_param(param_file, "MOM", "USE_REGRIDDING", use_ALE, &
                 "If True, use the ALE algorithm (regridding/remapping).", default=.false., do_not_log=.true.)
  call get_param(param_file, mdl, "MASS_WEIGHT_IN_PRESSURE_GRADIENT", CS%useMassWghtInterp, &
                 "If true, use mass weighting when interpolating T/S for integrals near the bathymetry.", default=.false.)
  call get_param(param_file, mdl, "USE_INACCURATE_PGF_RHO_ANOM", flag, default=.false.)
  call refuse(flag, "USE_INACCURATE_PGF_RHO_ANOM")
  call get_param(param_file, mdl, "RECONSTRUCT_FOR_PRESSURE", CS%reconstruct, &
                 "If True, use vertical reconstruction of T & S within the integrals of the FV pressure gradient calculation.", &
                 default=use_ALE)
  call get_param(param_file, mdl, "PRESSURE_RECONSTRUCTION_SCHEME", CS%Recon_Scheme, &
                 "Order of vertical reconstruction of T/S to use in the integrals within the FV pressure gradient calculation: "// &
                 "0: PCM, 1: PLM, 2: PPM.", default=1)
  if (CS%reconstruct .and. (CS%Recon_Scheme /= 1)) call refuse(.true., "PRESSURE_RECONSTRUCTION_SCHEME /= 1 (PLM)")
  call get_param(param_file, mdl, "BOUNDARY_EXTRAPOLATION_PRESSURE", CS%boundary_extrap, &
                 "If true, the reconstruction of T & S for pressure in boundary cells is extrapolated.", default=.true.)
  call get_param(param_file, mdl, "USE_STANLEY_PGF", flag, default=.false.)
  call refuse(flag, "USE_STANLEY_PGF")
  CS%GFS_scale = 1.0
  if (GV%g_prime(1) /= GV%g_Earth) CS%GFS_scale = GV%g_prime(1) / GV%g_Earth

  call mom6hip_read_eos(param_file, CS%eos, "PressureForce_FV_init")
  call mom6hip_read_topology(param_file)
contains
  subroutine refuse(on, name)
    logical,          intent(in) :: on
    character(len=*), intent(in) :: name
    if (on) call MOM_error(FATAL, "PressureForce_FV_init (HIP): "//name//" is not provided by the GPU path.")
  end subroutine refuse
end subroutine PressureForce_FV_init

end module MOM_PressureForce_FV
