"""Host-side mirror of MOM_sum_output's write_energy (reference: src/diagnostics/MOM_sum_output.F90:428): the global integrals
behind `ocean.stats` -- mass, kinetic energy, salt, heat as order-invariant extended-fixed-point sums, the maximum CFL numbers --
from fields that live on the GPU (mom6hip_write_energy_sums)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi
from ._lib import Mom6HipError, check, lib
from .tracer_advect import DeviceGrid, _ptr_space


def depth_list_setup(G: DeviceGrid, Z_ref=0.0, min_depth_inc=1.0e-10, domain=None):
    """depth_list_setup / create_depth_list (MOM_sum_output.F90:1067, :1109): the sorted depth list CALCULATE_APE needs, kept with
    the grid's context.  `domain`: the mom6_amd.domains.Domain of a multi-tile run (global size and the tile's offset)."""
    L = lib()
    L.mom6hip_depth_list_create.argtypes = [C.c_void_p] + [C.c_int32] * 4 + [C.c_double, C.c_double, C.POINTER(C.c_int32)]
    g = G.grid
    nig, njg, io, jo = (g.ni, g.nj, 0, 0) if domain is None else (domain.NI, domain.NJ, domain.i0, domain.j0)
    n = C.c_int32(0)
    check(L.mom6hip_depth_list_create(G.handle, int(nig), int(njg), int(io), int(jo), float(Z_ref), float(min_depth_inc), C.byref(n)),
          "depth_list_setup")
    return int(n.value)


def write_energy(u, v, h, tv, G: DeviceGrid, dt, C_p=3991.86795711963, H_to_kg_m2=1035.0, g_prime=None, Rho0=1035.0, Z_ref=0.0):
    """write_energy(u, v, h, tv, day, n, G, GV, US, CS, tracer_CSp): the sums it forms, as a dict (the reference prints them to
    ocean.stats).  tv = (T, S) or None (use_temperature = False).  With g_prime (GV%g_prime, nk+1 values) and a depth list
    (depth_list_setup) the available potential energy of CALCULATE_APE is included: PE, PE_tot, Z_0APE, toten = KE_tot + PE_tot."""
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
    r = dict(mass_tot=out.mass_tot, KE_tot=out.KE_tot, PE_tot=out.PE_tot, toten=out.toten, Salt=out.Salt, Heat=out.Heat,
             max_CFL=[out.max_CFL[0], out.max_CFL[1]], mass_EFP=list(out.mass_EFP), salt_EFP=list(out.salt_EFP),
             heat_EFP=list(out.heat_EFP), npoints=int(out.npoints), mass_lay=[float(x) for x in ml], KE_lay=[float(x) for x in kl])
    if g_prime is not None:      # CALCULATE_APE :610-680
        gp = np.ascontiguousarray(g_prime, dtype=np.float64)
        if gp.shape != (nk + 1,):
            raise Mom6HipError("write_energy: g_prime needs nk+1 values (GV%g_prime)")
        L.mom6hip_write_energy_ape.argtypes = [C.c_void_p] * 4 + [C.c_double] * 3 + [C.c_void_p] * 3 + [C.c_int32]
        PE, Z0, tot = np.zeros(nk + 1), np.zeros(nk + 1), C.c_double(0.0)
        ph, sp = _ptr_space(h)
        check(L.mom6hip_write_energy_ape(G.handle, C.c_void_p(ph), C.c_void_p(ml.ctypes.data), C.c_void_p(gp.ctypes.data), float(Rho0),
                                         float(H_to_kg_m2), float(Z_ref), C.c_void_p(PE.ctypes.data), C.byref(tot), C.c_void_p(Z0.ctypes.data), sp),
              "write_energy")
        r["PE"] = [float(x) for x in PE]; r["Z_0APE"] = [float(x) for x in Z0]; r["PE_tot"] = float(tot.value)
        r["toten"] = r["KE_tot"] + r["PE_tot"]      # :691
    return r
