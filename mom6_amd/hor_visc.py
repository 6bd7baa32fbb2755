"""Host-side mirror of MOM_hor_visc (reference: src/parameterizations/lateral/MOM_hor_visc.F90): hor_visc_init (:1984) and
horizontal_viscosity (:245).  The work is done by libmom6hip (mom6_amd/csrc/hor_visc.hip).  The static arrays of the
control structure live where `device_arrays` says (torch CUDA tensors, or numpy for the staged path)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi
from ._lib import Mom6HipError, check, lib
from .tracer_advect import DeviceGrid, _ptr_space

# parameter name of the reference -> member of the control structure
_PARAMS = {"LAPLACIAN": "Laplacian", "KH": "Kh", "KH_BG_MIN": "Kh_bg_min", "KH_VEL_SCALE": "Kh_vel_scale", "SMAGORINSKY_KH": "Smagorinsky_Kh",
           "SMAG_LAP_CONST": "Smag_Lap_const", "BOUND_KH": "bound_Kh", "BETTER_BOUND_KH": "better_bound_Kh",
           "ADD_LES_VISCOSITY": "add_LES_viscosity", "BIHARMONIC": "biharmonic", "AH": "Ah", "AH_VEL_SCALE": "Ah_vel_scale",
           "AH_TIME_SCALE": "Ah_time_scale", "SMAGORINSKY_AH": "Smagorinsky_Ah", "SMAG_BI_CONST": "Smag_bi_const", "BOUND_AH": "bound_Ah",
           "BETTER_BOUND_AH": "better_bound_Ah", "BOUND_CORIOLIS_BIHARM": "bound_Coriolis", "BOUND_CORIOLIS_VEL": "bound_Cor_vel",
           "HORVISC_BOUND_COEF": "bound_coef", "NOSLIP": "no_slip", "USE_LAND_MASK_FOR_HVISC": "use_land_mask",
           "USE_CONT_THICKNESS": "use_cont_thick"}
_UNSUPPORTED = {"LEITH_KH": "Leith_Kh", "LEITH_AH": "Leith_Ah", "USE_LEITHY": "use_Leithy", "RES_SCALE_MEKE_VISC": "MEKE_backscatter", "USE_GME": "use_GME",
                "ANISOTROPIC_VISCOSITY": "anisotropic", "RE_AH": "Re_Ah", "KH_SIN_LAT": "Kh_sin_lat", "USE_KH_BG_2D": "use_Kh_bg_2d",
                "USE_ZB2020": "use_ZB2020"}


def _setup():
    L = lib()
    if not getattr(L, "_hv_ready", False):
        cs = C.POINTER(_abi.HorViscCS)
        L.mom6hip_hor_visc_init.argtypes = [C.c_void_p, cs, C.c_double, C.c_int32]
        L.mom6hip_horizontal_viscosity.argtypes = [C.c_void_p, cs] + [C.c_void_p] * 5 + [C.c_double, C.c_void_p, C.c_void_p, C.c_int32]
        L.mom6hip_horizontal_viscosity_obc.argtypes = ([C.c_void_p, cs] + [C.c_void_p] * 5 + [C.c_double, C.c_void_p, C.c_void_p,
                                                                                              C.POINTER(_abi.Obc), C.c_int32])
        L._hv_ready = True
    return L


class hor_visc_CS:
    """hor_visc_CS (:40-243) as set by hor_visc_init: parameters by their reference names (defaults :2062-2300)."""

    def __init__(self, G: DeviceGrid, DT, device_arrays=True, MAXVEL=3.0e8, **params):
        g = G.grid
        d = dict(Kh=0.0, Kh_bg_min=0.0, Kh_vel_scale=0.0, Smag_Lap_const=0.0, Ah=0.0, Ah_vel_scale=0.0, Ah_time_scale=0.0, Smag_bi_const=0.0,
                 bound_Cor_vel=None, bound_coef=0.8, Laplacian=False, biharmonic=True, Smagorinsky_Kh=False, Smagorinsky_Ah=False,
                 bound_Kh=True, better_bound_Kh=None, bound_Ah=True, better_bound_Ah=None, bound_Coriolis=False, add_LES_viscosity=False,
                 no_slip=False, use_land_mask=True, use_cont_thick=False)
        st = self.st = _abi.HorViscCS()
        params.pop("USE_MEKE", None)      # (only decides what hor_visc_init logs, :2122-2128; MEKE acts through the MEKE argument)
        for k, v in params.items():
            if k in _PARAMS:
                d[_PARAMS[k]] = v
            elif k in _UNSUPPORTED:
                st.unsupported[_abi.HOR_VISC_UNSUPPORTED.index(_UNSUPPORTED[k])] = int(bool(v))
            else:
                raise Mom6HipError(f"hor_visc_init: unknown parameter {k}")
        if d["better_bound_Kh"] is None:
            d["better_bound_Kh"] = d["bound_Kh"]
        if d["better_bound_Ah"] is None:
            d["better_bound_Ah"] = d["bound_Ah"]
        if d["bound_Cor_vel"] is None:
            d["bound_Cor_vel"] = MAXVEL
        if not d["Smagorinsky_Ah"]:
            d["bound_Coriolis"] = False      # :2256
        for k, v in d.items():
            setattr(st, k, float(v) if isinstance(getattr(st, k), float) else int(bool(v)))
        self.arrays = {}
        for names, pos in ((_abi.HOR_VISC_ARRAYS_H, _abi.POS_H), (_abi.HOR_VISC_ARRAYS_Q, _abi.POS_Q)):
            for n in names:
                shp = g.shape2(pos)
                if device_arrays:
                    import torch
                    a = torch.zeros(shp, dtype=torch.float64, device="cuda")
                else:
                    a = np.zeros(shp)
                self.arrays[n] = a
                setattr(st, n, _ptr_space(a)[0])
        self.space = _abi.MEM_DEVICE if device_arrays else _abi.MEM_HOST
        self.dt = float(DT)
        check(_setup().mom6hip_hor_visc_init(G.handle, C.byref(st), self.dt, self.space), "hor_visc_init")

    def __getattr__(self, n):
        a = self.__dict__.get("arrays", {})
        if n in a:
            return a[n]
        raise AttributeError(n)


def hor_visc_init(G: DeviceGrid, DT, **params) -> hor_visc_CS:
    """hor_visc_init(Time, G, GV, US, param_file, diag, CS, ADp) -- :1984; DT is the baroclinic time step (the stability bounds)."""
    return hor_visc_CS(G, DT, **params)


def hor_visc_vel_stencil(CS):
    """hor_visc_vel_stencil (:2879)."""
    return 2


def horizontal_viscosity(u, v, h, diffu, diffv, MEKE, VarMix, G: DeviceGrid, CS: hor_visc_CS, tv=None, dt=None, OBC=None, BT=None, TD=None,
                         ADp=None, hu_cont=None, hv_cont=None, STOCH=None):
    """horizontal_viscosity(u, v, h, diffu, diffv, MEKE, VarMix, G, GV, US, CS, tv, dt, OBC, BT, TD, ADp, hu_cont, hv_cont, STOCH)
    -- :245.  MEKE: None, or a dict with any of Ku, Au (h-point 2-D arrays added to the Laplacian / biharmonic viscosity) and mom_src
    (receives the vertically summed frictional work); OBC: None or an ocean_OBC_type (mom6_amd/open_boundary.py); VarMix, BT, TD, ADp,
    STOCH belong to branches this build does not provide and must be None."""
    if CS is None or not CS.st.initialized:
        raise Mom6HipError("MOM_hor_visc: Module must be initialized before it is used.")
    if any(x is not None for x in (VarMix, BT, TD, ADp, STOCH)):
        raise Mom6HipError("horizontal_viscosity (HIP): VarMix, GME (BT, TD), ADp and STOCH are not supported on this path")
    MEKE = MEKE or {}
    if set(MEKE) - {"Ku", "Au", "mom_src"}:
        raise Mom6HipError("horizontal_viscosity (HIP): of MEKE only Ku, Au and mom_src are provided (no GME_snk, no backscatter)")
    spaces = {CS.space}
    for n in ("Ku", "Au", "mom_src"):
        a = MEKE.get(n)
        if a is None:
            setattr(CS.st, "MEKE_" + n, None)
        else:
            p, s = _ptr_space(a)
            spaces.add(s); setattr(CS.st, "MEKE_" + n, p)
    ptrs = []
    for a in (u, v, h, diffu, diffv, hu_cont, hv_cont):
        if a is None:
            ptrs.append(None)
            continue
        p, s = _ptr_space(a)
        spaces.add(s); ptrs.append(C.c_void_p(p))
    if len(spaces) != 1:
        raise Mom6HipError("horizontal_viscosity: the fields and the arrays of the control structure must be in the same memory space")
    if OBC is not None:      # the branches of an associated OBC on the PE (:449-452)
        from .open_boundary import _seg_to_ptr
        obc = OBC.struct(_seg_to_ptr(CS.space))      # (segment%tangential_vel is read with OBC_COMPUTED_STRAIN, in the memory space of the fields)
        check(_setup().mom6hip_horizontal_viscosity_obc(G.handle, C.byref(CS.st), ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4],
                                                        float(CS.dt if dt is None else dt), ptrs[5], ptrs[6], C.byref(obc), CS.space),
              "horizontal_viscosity")
        return
    check(_setup().mom6hip_horizontal_viscosity(G.handle, C.byref(CS.st), ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4],
                                                float(CS.dt if dt is None else dt), ptrs[5], ptrs[6], CS.space), "horizontal_viscosity")
