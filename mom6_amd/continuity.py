"""Host-side mirror of MOM_continuity_PPM (reference: src/core/MOM_continuity_PPM.F90): continuity_PPM_init /
continuity_PPM; the alias module MOM_continuity.F90:6 exports the latter as `continuity`."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi
from ._lib import Mom6HipError, check, lib
from .tracer_advect import DeviceGrid, _ptr_space


def continuity_PPM_init(G: DeviceGrid, **kw):
    """continuity_PPM_init (:2679-2757): the parameters and their defaults."""
    g = G.grid if isinstance(G, DeviceGrid) else G
    d = dict(upwind_1st=False, monotonic=False, simple_2nd=False, aggress_adjust=False, vol_CFL=None, better_iter=True,
             use_visc_rem_max=True, marginal_faces=True, tol_eta=0.5 * g.nk * g.Angstrom_H, tol_vel=3.0e8,
             CFL_limit_adjust=0.5)
    for k in kw:
        if k not in d:
            raise Mom6HipError(f"continuity_PPM_init: unknown parameter {k}")
    d.update(kw)
    if d["vol_CFL"] is None:
        d["vol_CFL"] = d["aggress_adjust"]
    return _abi.ContinuityCS(*[int(bool(d[n])) for n in ("upwind_1st", "monotonic", "simple_2nd", "aggress_adjust",
                                                         "vol_CFL", "better_iter", "use_visc_rem_max", "marginal_faces")],
                             float(d["tol_eta"]), float(d["tol_vel"]), float(d["CFL_limit_adjust"]))


def continuity_stencil(CS):
    """continuity_PPM_stencil (:2763)."""
    return 1 if CS.upwind_1st else (2 if CS.simple_2nd else 3)


class BT_cont_type:
    """The members of BT_cont_type (src/core/MOM_variables.F90) that continuity_PPM sets; arrays are numpy
    (host) or torch CUDA tensors (device), allocated by the caller."""

    def __init__(self, **arrays):
        self.arrays = arrays

    def struct(self, spaces):
        st = _abi.BTCont()
        for n in _abi.BT_CONT_U + _abi.BT_CONT_V + ("h_u", "h_v"):
            a = self.arrays.get(n)
            if a is not None:
                p, s = _ptr_space(a)
                spaces.add(s)
                setattr(st, n, p)
        return st


def continuity(u, v, hin, h, uh, vh, dt, G: DeviceGrid, CS, OBC=None, pbv=None, uhbt=None, vhbt=None, visc_rem_u=None,
               visc_rem_v=None, u_cor=None, v_cor=None, BT_cont: BT_cont_type | None = None, du_cor=None, dv_cor=None):
    """continuity_PPM(u, v, hin, h, uh, vh, dt, G, GV, US, CS, OBC, pbv, uhbt, vhbt, visc_rem_u, visc_rem_v,
    u_cor, v_cor, BT_cont, du_cor, dv_cor) -- MOM_continuity_PPM.F90:86."""
    if CS is None:
        raise Mom6HipError("MOM_continuity_PPM: Module must be initialized before it is used.")
    if pbv is not None:
        raise Mom6HipError("MOM_continuity_PPM (HIP): porous barriers are not supported")
    spaces = set()

    def P(a):
        if a is None:
            return None
        p, s = _ptr_space(a)
        spaces.add(s)
        return C.c_void_p(p)

    args = [P(x) for x in (u, v, hin, h, uh, vh)]
    opt = [P(x) for x in (uhbt, vhbt, visc_rem_u, visc_rem_v, u_cor, v_cor)]
    bt = None if BT_cont is None else BT_cont.struct(spaces)
    tail = [P(du_cor), P(dv_cor)]
    if len(spaces) != 1:
        raise Mom6HipError("continuity_PPM: all fields must be in the same memory space")
    space = spaces.pop()
    if OBC is not None:      # an ocean_OBC_type (mom6_amd/open_boundary.py): the OBC branches of continuity_PPM
        def to_ptr(a):      # the external values of a segment, in the memory space of the call
            if space == _abi.MEM_DEVICE:
                import torch
                t = a if hasattr(a, "data_ptr") else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
                return t.data_ptr(), t
            a = np.ascontiguousarray(a, dtype=np.float64)
            return a.ctypes.data, a
        obc = OBC.struct(to_ptr)
        L = lib()
        L.mom6hip_continuity_obc.argtypes = [C.c_void_p, C.POINTER(_abi.ContinuityCS), C.POINTER(_abi.Obc)] + [C.c_void_p] * 6 + [C.c_double] + \
            [C.c_void_p] * 6 + [C.POINTER(_abi.BTCont)] + [C.c_void_p] * 2 + [C.c_int32]
        check(L.mom6hip_continuity_obc(G.handle, C.byref(CS), C.byref(obc), *args, float(dt), *opt, None if bt is None else C.byref(bt), *tail, space),
              "continuity_PPM")
        return
    check(lib().mom6hip_continuity(G.handle, C.byref(CS), *args, float(dt), *opt,
                                   None if bt is None else C.byref(bt), *tail, space), "continuity_PPM")
