"""Host-side mirror of MOM_vert_friction (reference: src/parameterizations/vertical/MOM_vert_friction.F90):
vertvisc_init (:2465), vertvisc_coef (:1168), vertvisc (:526), vertvisc_remnant (:1064).  The work is done by
libmom6hip (mom6_amd/csrc/vert_friction.hip).  Arrays are numpy (host: staged) or torch CUDA tensors (device-resident);
the arrays of the control structure (a_u, a_v, h_u, h_v) live in the same memory space as the fields."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi
from ._lib import Mom6HipError, check, lib
from .open_boundary import _seg_to_ptr
from .tracer_advect import DeviceGrid, _ptr_space


def _setup():
    L = lib()
    if not getattr(L, "_vv_ready", False):
        cs, vt = C.POINTER(_abi.VertviscCS), C.POINTER(_abi.VertviscType)
        L.mom6hip_vertvisc_coef.argtypes = [C.c_void_p, cs] + [C.c_void_p] * 4 + [vt, C.c_double, C.c_int32]
        L.mom6hip_vertvisc.argtypes = [C.c_void_p, cs] + [C.c_void_p] * 5 + [vt, C.c_double, C.c_void_p, C.c_void_p, C.c_int32]
        L.mom6hip_vertvisc_coef_obc.argtypes = [C.c_void_p, cs] + [C.c_void_p] * 4 + [vt, C.c_double, C.POINTER(_abi.Obc), C.c_int32]
        L.mom6hip_vertvisc_obc.argtypes = ([C.c_void_p, cs] + [C.c_void_p] * 5 + [vt, C.c_double, C.c_void_p, C.c_void_p, C.POINTER(_abi.Obc),
                                            C.c_int32])
        L.mom6hip_vertvisc_remnant.argtypes = [C.c_void_p, cs, vt, C.c_void_p, C.c_void_p, C.c_double, C.c_int32]
        L.mom6hip_vertvisc_ntrunc.argtypes = [C.c_void_p, cs]
        L.mom6hip_vertvisc_and_remnant.argtypes = [C.c_void_p, cs] + [C.c_void_p] * 5 + [vt, C.c_double] + [C.c_void_p] * 4 + [C.c_int32]
        L.mom6hip_vertvisc_step.argtypes = ([C.c_void_p, cs] + [C.c_void_p] * 6 + [vt, C.c_double, C.c_int32] + [C.c_void_p] * 4 + [C.c_int32])
        L._vv_ready = True
    return L


class vertvisc_type:
    """The members of vertvisc_type (src/core/MOM_variables.F90:218-283) the provided branch reads."""
    FIELDS = ("Kv_bbl_u", "Kv_bbl_v", "bbl_thick_u", "bbl_thick_v", "Ray_u", "Ray_v", "Kv_shear", "Kv_shear_Bu",
              "nkml_visc_u", "nkml_visc_v",      # written by set_viscous_ML (DYNAMIC_VISCOUS_ML), read by vertvisc_coef
              "ustar")                           # forces%ustar, as find_ustar returns it (Z T-1)

    def __init__(self, **arrays):
        for n in arrays:
            if n not in self.FIELDS:
                raise Mom6HipError(f"vertvisc_type: unknown member {n}")
        self.arrays = {n: arrays.get(n) for n in self.FIELDS}
        self.st = _abi.VertviscType()
        self.space = None
        spaces = set()
        for n, a in self.arrays.items():
            if a is not None:
                p, s = _ptr_space(a)
                spaces.add(s)
                setattr(self.st, n, p)
        if len(spaces) > 1:
            raise Mom6HipError("vertvisc_type: all members must be in the same memory space")
        self.space = spaces.pop() if spaces else None


class vertvisc_CS:
    """vertvisc_CS (MOM_vert_friction.F90:40-190) as set by vertvisc_init (:2465): parameters by their reference names."""

    def __init__(self, G: DeviceGrid, KV, HBBL, HMIX_FIXED=0.0, BOTTOMDRAGLAW=True, HARMONIC_VISC=False, HARMONIC_BL_SCALE=0.0,
                 DIRECT_STRESS=False, HMIX_STRESS=None, KV_ML_INVZ2=0.0, KV_EXTRA_BBL=0.0, MAXVEL=3.0e8, CFL_BASED_TRUNCATIONS=True,
                 CFL_TRUNCATE=0.5, VEL_UNDERFLOW=0.0, VERT_FRICTION_ANSWER_DATE=99991231, device_arrays=True, DYNAMIC_VISCOUS_ML=False,
                 NKML=0, VON_KARMAN_CONST=0.41, **unsupported):
        g = G.grid
        st = self.st = _abi.VertviscCS()
        # DYNAMIC_VISCOUS_ML (:2543), GV%nkml (a bulk mixed layer), VON_KARMAN_CONST (:2590)
        st.dynamic_viscous_ML, st.nkml, st.vonKar = int(bool(DYNAMIC_VISCOUS_ML)), int(NKML), float(VON_KARMAN_CONST)
        for n, val in unsupported.items():
            if n not in _abi.VERTVISC_UNSUPPORTED:
                raise Mom6HipError(f"vertvisc_init: unknown parameter {n}")
            st.unsupported[_abi.VERTVISC_UNSUPPORTED.index(n)] = int(val)
        st.Kv, st.Hbbl, st.Hmix = float(KV), float(HBBL), float(HMIX_FIXED)
        st.bottomdraglaw, st.harmonic_visc, st.direct_stress = int(bool(BOTTOMDRAGLAW)), int(bool(HARMONIC_VISC)), int(bool(DIRECT_STRESS))
        st.harm_BL_val = float(HARMONIC_BL_SCALE)
        # HMIX_STRESS defaults to HMIX_FIXED (:2591-2593), in thickness units
        st.Hmix_stress = float(HMIX_STRESS if HMIX_STRESS is not None else HMIX_FIXED) * g.Z_to_H
        st.Kvml_invZ2, st.Kv_extra_bbl = float(KV_ML_INVZ2), float(KV_EXTRA_BBL)
        st.maxvel, st.CFL_based_trunc, st.CFL_trunc = float(MAXVEL), int(bool(CFL_BASED_TRUNCATIONS)), float(CFL_TRUNCATE)
        st.vel_underflow, st.answer_date = float(VEL_UNDERFLOW), int(VERT_FRICTION_ANSWER_DATE)
        st.H_to_RZ = g.Rho0 * g.H_to_Z      # GV%H_to_RZ (Boussinesq, unscaled units)
        self.arrays = {}
        for n, pos, extra in _abi.VERTVISC_CS_ARRAYS:
            shp = g.shape3(pos)
            shp = (shp[0] + extra,) + tuple(shp[1:])
            if device_arrays:
                import torch
                a = torch.zeros(shp, dtype=torch.float64, device="cuda")
            else:
                a = np.zeros(shp)
            self.arrays[n] = a
            setattr(st, n, _ptr_space(a)[0])
        self.space = _abi.MEM_DEVICE if device_arrays else _abi.MEM_HOST

    def __getattr__(self, n):
        a = self.__dict__.get("arrays", {})
        if n in a:
            return a[n]
        raise AttributeError(n)

    @property
    def ntrunc(self):
        return int(self.st.ntrunc)


def vertvisc_init(G: DeviceGrid, **params) -> vertvisc_CS:
    """vertvisc_init (:2465)."""
    return vertvisc_CS(G, **params)


def _space_of(CS, arrs, who):
    spaces = {CS.space}
    ptrs = []
    for a in arrs:
        if a is None:
            ptrs.append(None)
            continue
        p, s = _ptr_space(a)
        spaces.add(s)
        ptrs.append(C.c_void_p(p))
    if len(spaces) != 1:
        raise Mom6HipError(f"{who}: the fields and the arrays of the control structure must be in the same memory space")
    return ptrs, CS.space


def vertvisc_coef(u, v, h, dz, forces, visc: vertvisc_type, tv, dt, G: DeviceGrid, CS: vertvisc_CS, OBC=None, VarMix=None):
    """vertvisc_coef(u, v, h, dz, forces, visc, tv, dt, G, GV, US, CS, OBC, VarMix) -- :1168.  dz=None: Boussinesq
    thickness_to_dz.  forces and tv are only read by branches this build does not provide."""
    if CS is None:
        raise Mom6HipError("MOM_vert_friction(coef): Module must be initialized before it is used.")
    (pu, pv, ph, pdz), space = _space_of(CS, (u, v, h, dz), "vertvisc_coef")
    if visc.space not in (None, space):
        raise Mom6HipError("vertvisc_coef: visc must be in the same memory space as the fields")
    if OBC is not None:      # an ocean_OBC_type (mom6_amd/open_boundary.py): the zero-gradient projections at the segments' faces
        obc = OBC.struct(_seg_to_ptr(space))
        check(_setup().mom6hip_vertvisc_coef_obc(G.handle, C.byref(CS.st), pu, pv, ph, pdz, C.byref(visc.st), float(dt), C.byref(obc), space),
              "vertvisc_coef")
        return
    check(_setup().mom6hip_vertvisc_coef(G.handle, C.byref(CS.st), pu, pv, ph, pdz, C.byref(visc.st), float(dt), space), "vertvisc_coef")


def vertvisc(u, v, h, forces, visc: vertvisc_type, dt, OBC, ADp, CDp, G: DeviceGrid, CS: vertvisc_CS, taux_bot=None, tauy_bot=None,
             fpmix=None, Waves=None):
    """vertvisc(u, v, h, forces, visc, dt, OBC, ADp, CDp, G, GV, US, CS, taux_bot, tauy_bot, fpmix, Waves) -- :526.
    forces = (taux, tauy); u, v are updated in place."""
    if CS is None:
        raise Mom6HipError("MOM_vert_friction(visc): Module must be initialized before it is used.")
    if Waves is not None or fpmix:
        raise Mom6HipError("vertvisc (HIP): Waves and FPMIX are not supported on this path")
    taux, tauy = forces
    (pu, pv, ph, ptx, pty, pbx, pby), space = _space_of(CS, (u, v, h, taux, tauy, taux_bot, tauy_bot), "vertvisc")
    if visc.space not in (None, space):
        raise Mom6HipError("vertvisc: visc must be in the same memory space as the fields")
    if OBC is not None:      # the velocities of the specified segments are stored over the result (:988-1006)
        obc = OBC.struct(_seg_to_ptr(space))
        check(_setup().mom6hip_vertvisc_obc(G.handle, C.byref(CS.st), pu, pv, ph, ptx, pty, C.byref(visc.st), float(dt), pbx, pby,
                                            C.byref(obc), space), "vertvisc")
        return
    check(_setup().mom6hip_vertvisc(G.handle, C.byref(CS.st), pu, pv, ph, ptx, pty, C.byref(visc.st), float(dt), pbx, pby, space),
          "vertvisc")


def vertvisc_and_remnant(u, v, h, forces, visc: vertvisc_type, dt, G: DeviceGrid, CS: vertvisc_CS, visc_rem_u, visc_rem_v, taux_bot=None,
                         tauy_bot=None):
    """vertvisc followed by vertvisc_remnant with the same dt (the pair of MOM_dynamics_split_RK2.F90:731-744, :985-994) in one
    pass; the results are those of the two calls."""
    taux, tauy = forces
    (pu, pv, ph, ptx, pty, pbx, pby, pru, prv), space = _space_of(CS, (u, v, h, taux, tauy, taux_bot, tauy_bot, visc_rem_u, visc_rem_v),
                                                                  "vertvisc_and_remnant")
    check(_setup().mom6hip_vertvisc_and_remnant(G.handle, C.byref(CS.st), pu, pv, ph, ptx, pty, C.byref(visc.st), float(dt), pbx, pby,
                                                pru, prv, space), "vertvisc_and_remnant")


def vertvisc_step(u, v, h, dz, forces, visc: vertvisc_type, dt, G: DeviceGrid, CS: vertvisc_CS, visc_rem_u, visc_rem_v, update_velocities=True,
                  taux_bot=None, tauy_bot=None):
    """vertvisc_coef, then (update_velocities) vertvisc, then vertvisc_remnant with the same dt -- the sequences of
    MOM_dynamics_split_RK2.F90:598-600 / :717-744 / :974-994 -- in one kernel per direction; same results as the calls."""
    taux, tauy = forces if forces is not None else (None, None)
    (pu, pv, ph, pdz, ptx, pty, pbx, pby, pru, prv), space = _space_of(
        CS, (u, v, h, dz, taux, tauy, taux_bot, tauy_bot, visc_rem_u, visc_rem_v), "vertvisc_step")
    check(_setup().mom6hip_vertvisc_step(G.handle, C.byref(CS.st), pu, pv, ph, pdz, ptx, pty, C.byref(visc.st), float(dt),
                                         int(bool(update_velocities)), pbx, pby, pru, prv, space), "vertvisc_step")


def vertvisc_ntrunc(G: DeviceGrid, CS: vertvisc_CS) -> int:
    """CS%ntrunc after adding what the device counted since the last call (synchronises)."""
    check(_setup().mom6hip_vertvisc_ntrunc(G.handle, C.byref(CS.st)), "vertvisc_ntrunc")
    return CS.ntrunc


def vertvisc_remnant(visc: vertvisc_type, visc_rem_u, visc_rem_v, dt, G: DeviceGrid, CS: vertvisc_CS):
    """vertvisc_remnant(visc, visc_rem_u, visc_rem_v, dt, G, GV, US, CS) -- :1064."""
    if CS is None:
        raise Mom6HipError("MOM_vert_friction(remant): Module must be initialized before it is used.")
    (pu, pv), space = _space_of(CS, (visc_rem_u, visc_rem_v), "vertvisc_remnant")
    check(_setup().mom6hip_vertvisc_remnant(G.handle, C.byref(CS.st), C.byref(visc.st), pu, pv, float(dt), space), "vertvisc_remnant")
