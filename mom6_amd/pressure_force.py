"""Host-side mirror of MOM_PressureForce / MOM_PressureForce_FV and the part of MOM_EOS the pressure force
uses (reference: src/core/MOM_PressureForce.F90:41, src/core/MOM_PressureForce_FV.F90:462,
src/equation_of_state/MOM_EOS.F90)."""
from __future__ import annotations

import ctypes as C

from . import _abi
from ._lib import Mom6HipError, check, lib
from .tracer_advect import DeviceGrid, _ptr_space


def EOS_init(form="WRIGHT", Rho_T0_S0=1000.0, dRho_dT=-0.2, dRho_dS=0.8):
    """EOS_init (MOM_EOS.F90): EQN_OF_STATE and, for LINEAR, RHO_T0_S0 / DRHO_DT / DRHO_DS."""
    if form not in _abi.EOS_FORMS:
        raise Mom6HipError("interpret_eos_selection: EQN_OF_STATE " + str(form) + " is not provided by libmom6hip "
                           "(WRIGHT, WRIGHT_FULL, WRIGHT_REDUCED, UNESCO, LINEAR)")
    return _abi.EOS(_abi.EOS_FORMS[form], 0, float(Rho_T0_S0), float(dRho_dT), float(dRho_dS))


def calculate_density(T, S, pressure, rho, EOS, G: DeviceGrid, rho_ref=None):
    """calculate_density(T, S, pressure, rho, EOS, dom, rho_ref) -- MOM_EOS.F90:299, on the points of four equally shaped arrays
    (numpy: staged; torch.cuda: in place); with rho_ref the density anomaly from it."""
    spaces = set()
    n = None
    ptrs = []
    for a in (T, S, pressure, rho):
        p, sp = _ptr_space(a)
        spaces.add(sp); ptrs.append(C.c_void_p(p))
        m = int(a.numel()) if hasattr(a, "numel") else int(a.size)
        if n is not None and m != n:
            raise Mom6HipError("calculate_density: T, S, pressure and rho must have the same number of points")
        n = m
    if len(spaces) != 1:
        raise Mom6HipError("calculate_density: all fields must be in the same memory space")
    L = lib()
    L.mom6hip_calculate_density.argtypes = ([C.c_void_p, C.POINTER(_abi.EOS)] + [C.c_void_p] * 4 + [C.c_int64, C.c_int32, C.c_double, C.c_int32])
    check(L.mom6hip_calculate_density(G.handle, C.byref(EOS), *ptrs, n, 0 if rho_ref is None else 1, 0.0 if rho_ref is None else float(rho_ref),
                                      spaces.pop()), "calculate_density")


def PressureForce_init(grid, Rho0=None, boundary_extrap=True, useMassWghtInterp=False, Z_ref=0.0, reconstruct=True, use_ALE=True,
                       nk_rho_varies=0, P_Ref=2.0e7, Rlay=None, g_prime=None):
    """PressureForce_FV_init (MOM_PressureForce_FV.F90:921): RHO_PGF_REF, RECONSTRUCT_FOR_PRESSURE,
    BOUNDARY_EXTRAPOLATION_PRESSURE, MASS_WEIGHT_IN_PRESSURE_GRADIENT, and what PressureForce_FV_Bouss takes from its other
    arguments to choose its branch: use_ALE = associated(ALE_CSp), GV%nk_rho_varies, tv%P_Ref, GV%Rlay, GV%g_prime."""
    import numpy as np
    cs = _abi.PressureForceCS(float(grid.Rho0 if Rho0 is None else Rho0), 1.0, float(Z_ref), int(bool(reconstruct)), 1,
                              int(bool(boundary_extrap)), int(bool(useMassWghtInterp)), int(bool(use_ALE)), int(nk_rho_varies),
                              float(P_Ref), None, None)
    cs._keep = []
    for name, a in (("Rlay", Rlay), ("g_prime", g_prime)):
        if a is not None:
            a = np.ascontiguousarray(a, dtype=np.float64); cs._keep.append(a)
            setattr(cs, name, a.ctypes.data)
    return cs


def PressureForce(h, tv, PFu, PFv, G: DeviceGrid, CS, ALE_CSp=None, p_atm=None, pbce=None, eta=None, Boussinesq=True, H_to_RZ=1.0):
    """PressureForce(h, tv, PFu, PFv, G, GV, US, CS, ALE_CSp, p_atm, pbce, eta) -- MOM_PressureForce.F90:41.
    `tv` is (T, S, EOS); EOS None = no equation of state (layer densities GV%Rlay; T and S may be None).  Boussinesq=False (GV%Boussinesq, MOM_PressureForce_FV.F90:89): PressureForce_FV_nonBouss with h in
    mass per unit area and H_to_RZ = GV%H_to_RZ."""
    if CS is None:
        raise Mom6HipError("MOM_PressureForce_FV_Bouss: Module must be initialized before it is used.")
    T, S, EOS = tv
    spaces = set()

    def P(a):
        if a is None:
            return None
        p, s = _ptr_space(a)
        spaces.add(s)
        return C.c_void_p(p)

    args = [P(x) for x in (h, T, S, p_atm, PFu, PFv, pbce, eta)]
    if len(spaces) != 1:
        raise Mom6HipError("PressureForce: all fields must be in the same memory space")
    if not Boussinesq:
        check(lib().mom6hip_pressureforce_fv_nonbouss(G.handle, C.byref(CS), C.byref(EOS), *args[:4], float(H_to_RZ), *args[4:], spaces.pop()),
              "PressureForce_FV_nonBouss")
        return
    check(lib().mom6hip_pressureforce_fv_bouss(G.handle, C.byref(CS), C.byref(EOS) if EOS is not None else None, *args, spaces.pop()),
          "PressureForce_FV_Bouss")
