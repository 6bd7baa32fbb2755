"""Host-side mirror of MOM_tracer_hor_diff (reference: src/tracer/MOM_tracer_hor_diff.F90): tracer_hor_diff_init (:1625) and
tracer_hordiff (:119) -- the along-layer diffusion with a constant KHTR or the VarMix / MEKE diffusivities of :236-281, and with
USE_NEUTRAL_DIFFUSION the continuous branch of MOM_neutral_diffusion (:474-534; mom6_amd/csrc/neutral_diffusion.hip), with
DIFFUSE_ML_TO_INTERIOR tracer_epipycnal_ML_diff (:700; mom6_amd/csrc/epipycnal_diff.hip).  The work is done by libmom6hip
(mom6_amd/csrc/tracer_hor_diff.hip)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi
from ._lib import Mom6HipError, check, lib
from .tracer_advect import DeviceGrid, _ptr_space

_PARAMS = {"KHTR": "KhTr", "MAX_TR_DIFFUSION_CFL": "max_diff_CFL", "CHECK_DIFFUSIVE_CFL": "check_diffusive_CFL", "KHTR_SLOPE_CFF": "KhTr_Slope_Cff",
           "KHTR_MIN": "KhTr_min", "KHTR_MAX": "KhTr_max", "KHTR_PASSIVITY_COEFF": "KhTr_passivity_coeff", "KHTR_PASSIVITY_MIN": "KhTr_passivity_min"}
# parameters of the reference whose branches this build does not provide: accepted at their defaults, refused otherwise
_REFUSED = {"USE_HORIZONTAL_BOUNDARY_DIFFUSION": 1, "KHTR_USE_EBT_STRUCT": 5}
# DIFFUSE_ML_TO_INTERIOR and its parameters (:1687-1727)
_EPI_PARAMS = {"ML_KHTR_SCALE": ("ML_KhTr_scale", float), "HOR_DIFF_ANSWER_DATE": ("answer_date", int), "HOR_DIFF_LIMIT_BUG": ("limit_bug", bool)}
# neutral_diffusion_init (src/tracer/MOM_neutral_diffusion.F90:138): the parameters of the continuous branch, and those refused
_ND_PARAMS = {"NDIFF_REF_PRES": ("ref_pres", float), "NDIFF_ANSWER_DATE": ("ndiff_answer_date", int), "RECALC_NEUTRAL_SURF": ("recalc_neutral_surf", bool),
              "NDIFF_INTERIOR_ONLY": ("interior_only", bool)}
_ND_REFUSED = {"NDIFF_TAPERING": 2, "NDIFF_USE_UNMASKED_TRANSPORT_BUG": 4}


class tracer_hor_diff_CS:
    """tracer_hor_diff_CS (:40-100); parameters by their reference names (defaults :1652-1700)."""

    def __init__(self, **params):
        st = self.st = _abi.TracerHorDiffCS()
        st.KhTr, st.max_diff_CFL, st.check_diffusive_CFL, st.KhTr_passivity_min = 0.0, -1.0, 0, 0.5
        nd = self.neutral_diffusion_CSp = _abi.NeutralDiffusionCS()
        nd.ref_pres, nd.ndiff_answer_date, nd.H_to_RZ = -1.0, 20240101, 0.0      # H_to_RZ: GV%H_to_RZ, taken from the grid at the call
        ep = self.epipycnal = _abi.EpipycnalCS()
        ep.ML_KhTr_scale, ep.answer_date, ep.limit_bug = 1.0, 20240101, 1
        for k, v in params.items():
            if k == "DIFFUSE_ML_TO_INTERIOR":
                st.unsupported[2] = int(bool(v))      # CS%Diffuse_ML_interior (taken by mom6hip_tracer_hordiff_epipycnal)
            elif k in _EPI_PARAMS:
                setattr(ep, _EPI_PARAMS[k][0], _EPI_PARAMS[k][1](v))
            elif k == "USE_NEUTRAL_DIFFUSION":
                st.unsupported[0] = int(bool(v))      # CS%use_neutral_diffusion (taken by mom6hip_tracer_hordiff_neutral)
            elif k in _ND_PARAMS:
                setattr(nd, _ND_PARAMS[k][0], _ND_PARAMS[k][1](v))
            elif k == "NDIFF_CONTINUOUS":
                nd.unsupported[0] = int(not v)
            elif k in _ND_REFUSED:
                nd.unsupported[_ND_REFUSED[k]] = int(bool(v))
            elif k == "GV_H_to_RZ":
                nd.H_to_RZ = float(v)
            elif k in _PARAMS:
                a = _PARAMS[k]
                setattr(st, a, int(bool(v)) if a == "check_diffusive_CFL" else float(v))
            elif k in _REFUSED:
                if v:
                    st.unsupported[_REFUSED[k]] = 1
            else:
                raise Mom6HipError(f"tracer_hor_diff_init: unknown parameter {k}")
        if st.unsupported[0] and st.unsupported[2]:
            raise Mom6HipError("MOM_tracer_hor_diff: USE_NEUTRAL_DIFFUSION and DIFFUSE_ML_TO_INTERIOR are mutually exclusive!")      # :1732
        st.initialized = 1; nd.initialized = 1
        self.last = None


def tracer_hor_diff_init(Time=None, G=None, GV=None, US=None, param_file=None, diag=None, EOS=None, diabatic_CSp=None, **params):
    """tracer_hor_diff_init(Time, G, GV, US, param_file, diag, EOS, diabatic_CSp, CS) -- :1625 (parameters as keywords)."""
    return tracer_hor_diff_CS(**params)


def tracer_hordiff(h, dt, MEKE, VarMix, visc, G: DeviceGrid, CS: tracer_hor_diff_CS, Reg, tv=None, do_online_flag=None, read_khdt_x=None,
                   read_khdt_y=None, conc_underflow=None, GV=None):
    """tracer_hordiff(h, dt, MEKE, VarMix, visc, G, GV, US, CS, Reg, tv, do_online_flag, read_khdt_x, read_khdt_y) -- :119.
    Reg: the list of tracer arrays (Reg%Tr(m)%t), updated in place.  VarMix: None, or a dict (its presence is
    VarMix%use_variable_mixing) with any of L2u, L2v, SN_u, SN_v (read with KHTR_SLOPE_CFF > 0), Res_fn_h (its presence is
    VarMix%Resoln_scaled_KhTr), Rd_dx_h (KHTR_PASSIVITY_COEFF > 0); MEKE: None, or a dict with Kh (MEKE%Kh) and KhTr_fac
    (MEKE%KhTr_fac); MEKE%Kh is read with variable mixing only, as in the reference.  Returns the iteration statistics."""
    if CS is None or Reg is None:
        raise Mom6HipError("MOM_tracer_hor_diff: register_tracer must be called before tracer_hordiff.")
    if do_online_flag is False or read_khdt_x is not None or read_khdt_y is not None:
        raise Mom6HipError("tracer_hordiff (HIP): offline khdt arrays are not supported on this path")
    if CS.st.unsupported[0]:
        return _tracer_hordiff_neutral(h, dt, MEKE, VarMix, visc, G, CS, Reg, tv, conc_underflow)
    if CS.st.unsupported[2]:
        return _tracer_hordiff_epipycnal(h, dt, MEKE, VarMix, G, GV, CS, Reg, tv, conc_underflow)
    L = lib()
    L.mom6hip_tracer_hordiff_varmix.argtypes = [C.c_void_p, C.POINTER(_abi.TracerHorDiffCS), C.POINTER(_abi.HorDiffFields), C.c_void_p, C.c_double,
                                                C.POINTER(C.c_void_p), C.c_void_p, C.c_int32, C.c_int32, C.POINTER(_abi.HorDiffStats)]
    tr = list(Reg)
    spaces = set()
    hp, s0 = _ptr_space(h); spaces.add(s0)
    F = _abi.HorDiffFields()
    st = CS.st
    st.use_variable_mixing = int(VarMix is not None)
    st.Resoln_scaled_KhTr = int(VarMix is not None and VarMix.get("Res_fn_h") is not None)
    if set(VarMix or {}) - set(_abi.HORDIFF_FIELDS):
        raise Mom6HipError("tracer_hordiff (HIP): of VarMix only L2u/v, SN_u/v, Res_fn_h and Rd_dx_h are read")
    for n, a in list((VarMix or {}).items()) + ([("MEKE_Kh", MEKE.get("Kh"))] if MEKE else []):
        if a is not None:
            p, s = _ptr_space(a); spaces.add(s); setattr(F, n, p)
    if MEKE:
        st.KhTr_fac = float(MEKE.get("KhTr_fac", 1.0))
    ptrs = (C.c_void_p * max(len(tr), 1))()
    for m, t in enumerate(tr):
        p, s = _ptr_space(t); ptrs[m] = p; spaces.add(s)
    if len(spaces) != 1:
        raise Mom6HipError("tracer_hordiff: h and every tracer must be in the same memory space")
    cu = None if conc_underflow is None else np.ascontiguousarray(conc_underflow, dtype=np.float64)
    stats = _abi.HorDiffStats()
    check(L.mom6hip_tracer_hordiff_varmix(G.handle, C.byref(CS.st), C.byref(F), C.c_void_p(hp), float(dt), ptrs, None if cu is None else cu.ctypes.data,
                                          len(tr), spaces.pop(), C.byref(stats)), "tracer_hordiff")
    CS.last = stats
    return stats


def _same(a, b):
    return a is b or (hasattr(a, "data_ptr") and hasattr(b, "data_ptr") and a.data_ptr() == b.data_ptr())


def _tracer_hordiff_neutral(h, dt, MEKE, VarMix, visc, G, CS, Reg, tv, conc_underflow):
    """the USE_NEUTRAL_DIFFUSION branch (:474-534): tv has T, S (two of the arrays of Reg, as in the reference where the registry
    points at tv%T and tv%S), eqn_of_state (an _abi.EOS) and optionally p_surf -- a dict or an object; with NDIFF_INTERIOR_ONLY visc
    has h_ML (visc%h_ML, the boundary-layer depth)."""
    get = (lambda n: tv.get(n)) if isinstance(tv, dict) else (lambda n: getattr(tv, n, None))
    if tv is None or get("T") is None or get("S") is None or get("eqn_of_state") is None:
        raise Mom6HipError("tracer_hordiff: USE_NEUTRAL_DIFFUSION needs tv%T, tv%S and tv%eqn_of_state")
    tr = list(Reg)
    idx = [next((m for m, t in enumerate(tr) if _same(t, get(n))), -1) for n in ("T", "S")]
    if min(idx) < 0:
        raise Mom6HipError("tracer_hordiff: tv%T and tv%S must be registered tracers (entries of Reg)")
    if VarMix is not None and set(VarMix) - set(_abi.HORDIFF_FIELDS):
        raise Mom6HipError("tracer_hordiff (HIP): of VarMix only L2u/v, SN_u/v, Res_fn_h and Rd_dx_h are read")
    L = lib()
    L.mom6hip_tracer_hordiff_neutral.argtypes = [C.c_void_p, C.POINTER(_abi.TracerHorDiffCS), C.POINTER(_abi.NeutralDiffusionCS),
                                                 C.POINTER(_abi.HorDiffFields), C.c_void_p, C.POINTER(_abi.EOS), C.c_void_p, C.c_double,
                                                 C.POINTER(C.c_void_p), C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                 C.POINTER(_abi.HorDiffStats)]
    spaces = set()
    hp, s0 = _ptr_space(h); spaces.add(s0)
    F = _abi.HorDiffFields()
    st = CS.st
    st.use_variable_mixing = int(VarMix is not None)
    st.Resoln_scaled_KhTr = int(VarMix is not None and VarMix.get("Res_fn_h") is not None)
    for n, a in list((VarMix or {}).items()) + ([("MEKE_Kh", MEKE.get("Kh"))] if MEKE else []):
        if a is not None:
            p, s = _ptr_space(a); spaces.add(s); setattr(F, n, p)
    if MEKE:
        st.KhTr_fac = float(MEKE.get("KhTr_fac", 1.0))
    ptrs = (C.c_void_p * max(len(tr), 1))()
    for m, t in enumerate(tr):
        p, s = _ptr_space(t); ptrs[m] = p; spaces.add(s)
    if CS.neutral_diffusion_CSp.interior_only:
        hml = None if visc is None else (visc.get("h_ML") if isinstance(visc, dict) else getattr(visc, "h_ML", None))
        if hml is None:
            raise Mom6HipError("hor_bnd_diffusion requires that visc%h_ML is associated.")
        p, s = _ptr_space(hml); spaces.add(s); F.h_ML = p
    ps = None
    if get("p_surf") is not None:
        ps, s = _ptr_space(get("p_surf")); spaces.add(s)
    if len(spaces) != 1:
        raise Mom6HipError("tracer_hordiff: h, p_surf and every tracer must be in the same memory space")
    nd = CS.neutral_diffusion_CSp
    if nd.H_to_RZ == 0.0:
        nd.H_to_RZ = float(G.grid.Rho0 * G.grid.H_to_Z)      # GV%H_to_RZ, Boussinesq
    cu = None if conc_underflow is None else np.ascontiguousarray(conc_underflow, dtype=np.float64)
    stats = _abi.HorDiffStats()
    check(L.mom6hip_tracer_hordiff_neutral(G.handle, C.byref(st), C.byref(nd), C.byref(F), C.c_void_p(hp), C.byref(get("eqn_of_state")),
                                           None if ps is None else C.c_void_p(ps), float(dt), ptrs, None if cu is None else cu.ctypes.data,
                                           len(tr), idx[0], idx[1], spaces.pop(), C.byref(stats)), "tracer_hordiff")
    CS.last = stats
    return stats


def _tracer_hordiff_epipycnal(h, dt, MEKE, VarMix, G, GV, CS, Reg, tv, conc_underflow):
    """the DIFFUSE_ML_TO_INTERIOR branches (:544-550, :613-620 and tracer_epipycnal_ML_diff :700): tv has T, S (two of the arrays of
    Reg), eqn_of_state (an _abi.EOS) and P_Ref; GV -- a dict or an object -- has Rlay (nk values), nkml and nk_rho_varies."""
    get = (lambda n: tv.get(n)) if isinstance(tv, dict) else (lambda n: getattr(tv, n, None))
    gv = (lambda n: GV.get(n)) if isinstance(GV, dict) else (lambda n: getattr(GV, n, None))
    if tv is None or get("T") is None or get("S") is None or get("eqn_of_state") is None or get("P_Ref") is None:
        raise Mom6HipError("tracer_hordiff: DIFFUSE_ML_TO_INTERIOR needs tv%T, tv%S, tv%eqn_of_state and tv%P_Ref")
    if GV is None or gv("Rlay") is None or gv("nkml") is None or gv("nk_rho_varies") is None:
        raise Mom6HipError("tracer_hordiff: DIFFUSE_ML_TO_INTERIOR needs GV%Rlay, GV%nkml and GV%nk_rho_varies")
    tr = list(Reg)
    idx = [next((m for m, t in enumerate(tr) if _same(t, get(n))), -1) for n in ("T", "S")]
    if min(idx) < 0:
        raise Mom6HipError("tracer_hordiff: tv%T and tv%S must be registered tracers (entries of Reg)")
    if VarMix is not None and set(VarMix) - set(_abi.HORDIFF_FIELDS):
        raise Mom6HipError("tracer_hordiff (HIP): of VarMix only L2u/v, SN_u/v, Res_fn_h and Rd_dx_h are read")
    L = lib()
    L.mom6hip_tracer_hordiff_epipycnal.argtypes = [C.c_void_p, C.POINTER(_abi.TracerHorDiffCS), C.POINTER(_abi.EpipycnalCS),
                                                   C.POINTER(_abi.HorDiffFields), C.c_void_p, C.POINTER(_abi.EOS), C.c_double,
                                                   C.POINTER(C.c_void_p), C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                   C.POINTER(_abi.HorDiffStats)]
    spaces = set()
    hp, s0 = _ptr_space(h); spaces.add(s0)
    F = _abi.HorDiffFields()
    st = CS.st
    st.use_variable_mixing = int(VarMix is not None)
    st.Resoln_scaled_KhTr = int(VarMix is not None and VarMix.get("Res_fn_h") is not None)
    for n, a in list((VarMix or {}).items()) + ([("MEKE_Kh", MEKE.get("Kh"))] if MEKE else []):
        if a is not None:
            p, s = _ptr_space(a); spaces.add(s); setattr(F, n, p)
    if MEKE:
        st.KhTr_fac = float(MEKE.get("KhTr_fac", 1.0))
    ptrs = (C.c_void_p * max(len(tr), 1))()
    for m, t in enumerate(tr):
        p, s = _ptr_space(t); ptrs[m] = p; spaces.add(s)
    if len(spaces) != 1:
        raise Mom6HipError("tracer_hordiff: h and every tracer must be in the same memory space")
    ep = CS.epipycnal
    Rlay = np.ascontiguousarray(gv("Rlay"), dtype=np.float64)
    if Rlay.size != G.grid.nk:
        raise Mom6HipError("tracer_hordiff: GV%Rlay must have nk values")
    ep.Rlay = Rlay.ctypes.data; ep.nkml, ep.nk_rho_varies, ep.P_Ref = int(gv("nkml")), int(gv("nk_rho_varies")), float(get("P_Ref"))
    cu = None if conc_underflow is None else np.ascontiguousarray(conc_underflow, dtype=np.float64)
    stats = _abi.HorDiffStats()
    check(L.mom6hip_tracer_hordiff_epipycnal(G.handle, C.byref(st), C.byref(ep), C.byref(F), C.c_void_p(hp), C.byref(get("eqn_of_state")),
                                             float(dt), ptrs, None if cu is None else cu.ctypes.data, len(tr), idx[0], idx[1],
                                             spaces.pop(), C.byref(stats)), "tracer_hordiff")
    CS.last = stats
    return stats
