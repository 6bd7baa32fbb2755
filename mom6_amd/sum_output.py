"""Host-side mirror of MOM_sum_output's write_energy (reference: src/diagnostics/MOM_sum_output.F90:428): the global integrals
behind `ocean.stats` -- mass, kinetic energy, salt, heat as order-invariant extended-fixed-point sums, the maximum CFL numbers --
from fields that live on the GPU (mom6hip_write_energy_sums)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi
from ._lib import Mom6HipError, check, lib
from .tracer_advect import DeviceGrid, _ptr_space


def write_energy(u, v, h, tv, G: DeviceGrid, dt, C_p=3991.86795711963, H_to_kg_m2=1035.0):
    """write_energy(u, v, h, tv, day, n, G, GV, US, CS, tracer_CSp): the sums it forms, as a dict (the reference prints them to
    ocean.stats).  tv = (T, S) or None (use_temperature = False).  CALCULATE_APE is not provided (PE_tot = 0)."""
    L = lib()
    if not getattr(L, "_energy_ready", False):
        L.mom6hip_write_energy_sums.argtypes = [C.c_void_p] * 6 + [C.c_double] * 3 + [C.c_void_p, C.c_void_p, C.POINTER(_abi.EnergySums), C.c_int32]
        L._energy_ready = True
    T, S = tv if tv is not None else (None, None)
    ptrs, spaces = [], set()
    for a in (u, v, h, T, S):
        if a is None:
            ptrs.append(None); continue
        p, s = _ptr_space(a)
        ptrs.append(C.c_void_p(p)); spaces.add(s)
    if len(spaces) != 1:
        raise Mom6HipError("write_energy: all fields must be in the same memory space")
    nk = G.grid.nk
    ml, kl = np.zeros(nk), np.zeros(nk)
    out = _abi.EnergySums()
    check(L.mom6hip_write_energy_sums(G.handle, *ptrs, float(dt), float(C_p), float(H_to_kg_m2), C.c_void_p(ml.ctypes.data),
                                      C.c_void_p(kl.ctypes.data), C.byref(out), spaces.pop()), "write_energy")
    return dict(mass_tot=out.mass_tot, KE_tot=out.KE_tot, PE_tot=out.PE_tot, toten=out.toten, Salt=out.Salt, Heat=out.Heat,
                max_CFL=[out.max_CFL[0], out.max_CFL[1]], mass_EFP=list(out.mass_EFP), salt_EFP=list(out.salt_EFP),
                heat_EFP=list(out.heat_EFP), npoints=int(out.npoints), mass_lay=[float(x) for x in ml], KE_lay=[float(x) for x in kl])
