"""Host-side mirror of MOM_thickness_diffuse (reference: src/parameterizations/lateral/MOM_thickness_diffuse.F90): thickness_diffuse_init
(:2169) and thickness_diffuse (:133).  The work is done by libmom6hip (mom6_amd/csrc/thickness_diffuse.hip)."""
from __future__ import annotations

import ctypes as C

from . import _abi
from ._lib import Mom6HipError, check, lib
from .tracer_advect import DeviceGrid, _ptr_space

# parameter of the reference -> member of _abi.THICKNESS_DIFFUSE_UNSUPPORTED
_UNSUPPORTED = {"DETANGLE_INTERFACES": "detangle_interfaces", "USE_STANLEY_GM": "use_stanley_gm",
                "MEKE_GEOMETRIC": "MEKE_GEOMETRIC", "MEKE_GM_SRC_ALT": "GM_src_alt", "READ_KHTH": "read_khth", "KHTH_USE_EBT_STRUCT": "ebt_struct",
                "USE_KH_IN_MEKE": "Use_KH_in_MEKE"}


def _setup():
    L = lib()
    if not getattr(L, "_td_ready", False):
        L.mom6hip_thickness_diffuse.argtypes = ([C.c_void_p, C.POINTER(_abi.ThicknessDiffuseCS)] + [C.c_void_p] * 5
                                                + [C.POINTER(_abi.EOS), C.c_double, C.c_void_p, C.c_void_p, C.c_int32])
        L._td_ready = True
    return L


class thickness_diffuse_CS:
    """thickness_diffuse_CS (:40-128) as set by thickness_diffuse_init: parameters by their reference names (defaults :2203-2400)."""

    def __init__(self, G: DeviceGrid, THICKNESSDIFFUSE=False, KHTH=0.0, KHTH_MIN=0.0, KHTH_MAX=0.0, KHTH_MAX_CFL=0.8, KHTH_SLOPE_MAX=0.01,
                 KD_SMOOTH=1.0e-6, KHTH_SLOPE_CFF=0.0, MEKE_KHTH_FAC=1.0, USE_GM_WORK_BUG=False, NKML=0, KH_ETA_CONST=0.0, KH_ETA_VEL_SCALE=0.0,
                 KHTH_USE_FGNV_STREAMFUNCTION=False, FGNV_FILTER_SCALE=1.0, FGNV_STRAT_FLOOR=1.0e-15, OMEGA=7.2921e-5, **unsupported):
        st = self.st = _abi.ThicknessDiffuseCS()
        st.thickness_diffuse = int(bool(THICKNESSDIFFUSE))
        st.Khth, st.Khth_Min, st.Khth_Max, st.max_Khth_CFL, st.slope_max = float(KHTH), float(KHTH_MIN), float(KHTH_MAX), float(KHTH_MAX_CFL), float(KHTH_SLOPE_MAX)
        st.kappa_smooth, st.KHTH_Slope_Cff, st.KhTh_fac = float(KD_SMOOTH), float(KHTH_SLOPE_CFF), float(MEKE_KHTH_FAC)
        st.use_GM_work_bug, st.nkml = int(bool(USE_GM_WORK_BUG)), int(NKML)
        # KHTH_USE_FGNV_STREAMFUNCTION (:2305-2337): the streamfunction of Ferrari et al. (2010); N2_floor = (FGNV_STRAT_FLOOR * OMEGA)**2
        st.use_FGNV_streamfn, st.FGNV_scale = int(bool(KHTH_USE_FGNV_STREAMFUNCTION)), float(FGNV_FILTER_SCALE)
        st.N2_floor = (float(FGNV_STRAT_FLOOR) * float(OMEGA)) ** 2 if KHTH_USE_FGNV_STREAMFUNCTION else 0.0
        if KH_ETA_CONST > 0.0 or KH_ETA_VEL_SCALE > 0.0:
            st.unsupported[_abi.THICKNESS_DIFFUSE_UNSUPPORTED.index("Kh_eta")] = 1
        for k, v in unsupported.items():
            if k not in _UNSUPPORTED:
                raise Mom6HipError(f"thickness_diffuse_init: unknown parameter {k}")
            st.unsupported[_abi.THICKNESS_DIFFUSE_UNSUPPORTED.index(_UNSUPPORTED[k])] = int(bool(v))
        st.initialized = 1
        self._keep = {}


def thickness_diffuse_init(G: DeviceGrid, **params) -> thickness_diffuse_CS:
    """thickness_diffuse_init(Time, G, GV, US, param_file, diag, CDp, CS) -- :2169."""
    return thickness_diffuse_CS(G, **params)


def thickness_diffuse(h, uhtr, vhtr, tv, dt, G: DeviceGrid, MEKE, VarMix, CDp, CS: thickness_diffuse_CS, STOCH=None):
    """thickness_diffuse(h, uhtr, vhtr, tv, dt, G, GV, US, MEKE, VarMix, CDp, CS, STOCH) -- :133.  tv = (T, S, EOS) or None (no equation
    of state; the work then needs VarMix-independent GV%Rlay in MEKE["Rlay"], the FGNV streamfunction GV%g_prime in MEKE["g_prime"]); MEKE: None or a dict with Kh (MEKE%Kh), GM_src (MEKE%GM_src,
    output), Rlay; VarMix: None or a dict with any of L2u, L2v, SN_u, SN_v (use_Visbeck), Res_fn_u, Res_fn_v (Resoln_scaled_KhTh), Depth_fn_u, Depth_fn_v (Depth_scaled_KhTh),
    slope_x, slope_y (use_stored_slopes), cg1 (with KHTH_USE_FGNV_STREAMFUNCTION) -- its presence is VarMix%use_variable_mixing; CDp: None or a dict with uhGM, vhGM (outputs)."""
    if CS is None or not CS.st.initialized:
        raise Mom6HipError("MOM_thickness_diffuse: Module must be initialized before it is used.")
    if STOCH is not None:
        raise Mom6HipError("thickness_diffuse (HIP): stochastic parameterizations are not supported on this path")
    T, S, EOS = tv if tv is not None else (None, None, None)
    MEKE, VarMix, CDp = MEKE or {}, VarMix, CDp or {}
    st = CS.st
    st.use_variable_mixing = int(VarMix is not None)
    fields = {"MEKE_Kh": MEKE.get("Kh"), "MEKE_GM_src": MEKE.get("GM_src")}
    for n in ("L2u", "L2v", "SN_u", "SN_v", "Res_fn_u", "Res_fn_v", "slope_x", "slope_y", "cg1", "Depth_fn_u", "Depth_fn_v"):
        fields[n] = (VarMix or {}).get(n)
    if set(VarMix or {}) - set(fields):
        raise Mom6HipError("thickness_diffuse (HIP): of VarMix only L2u/v, SN_u/v, Res_fn_u/v, Depth_fn_u/v, slope_x/y and cg1 are provided")
    spaces = set()
    for n, a in fields.items():
        if a is None:
            setattr(st, n, None)
        else:
            p, sp = _ptr_space(a)
            spaces.add(sp); setattr(st, n, p)
    rl = MEKE.get("Rlay")
    if rl is not None:
        import numpy as np
        CS._keep["Rlay"] = np.ascontiguousarray(rl, dtype=np.float64)
        st.Rlay = CS._keep["Rlay"].ctypes.data
    gp = MEKE.get("g_prime")      # GV%g_prime(1:nk+1), host: the stratification of the FGNV solve without an equation of state
    if gp is not None:
        import numpy as np
        CS._keep["g_prime"] = np.ascontiguousarray(gp, dtype=np.float64)
        st.g_prime = CS._keep["g_prime"].ctypes.data
    ptrs = []
    for a in (h, uhtr, vhtr, T, S, CDp.get("uhGM"), CDp.get("vhGM")):
        if a is None:
            ptrs.append(None)
            continue
        p, sp = _ptr_space(a)
        spaces.add(sp); ptrs.append(C.c_void_p(p))
    if len(spaces) != 1:
        raise Mom6HipError("thickness_diffuse: all fields must be in the same memory space")
    check(_setup().mom6hip_thickness_diffuse(G.handle, C.byref(st), *ptrs[:5], None if EOS is None else C.byref(EOS), float(dt), ptrs[5], ptrs[6],
                                             spaces.pop()), "thickness_diffuse")
