"""Host-side mirror of MOM_dynamics_split_RK2 (reference: src/core/MOM_dynamics_split_RK2.F90):
initialize_dyn_split_RK2 (:1326) and step_MOM_dyn_split_RK2 (:289).  The step itself is one call into the library,
which enqueues every kernel of the step on the GPU in the reference's order; this module only owns the control
structures and the device arrays the reference keeps in MOM_dyn_split_RK2_CS."""
from __future__ import annotations

import ctypes as C

import torch

from . import _abi
from ._lib import Mom6HipError, check, lib
from .barotropic import barotropic_CS, barotropic_init
from .continuity import BT_cont_type, continuity_PPM_init
from .coriolis_adv import CoriolisAdv_init
from .pressure_force import EOS_init, PressureForce_init
from .tracer_advect import DeviceGrid


def _setup():
    L = lib()
    if not getattr(L, "_rk2_ready", False):
        cs = C.POINTER(_abi.DynSplitRK2CS)
        L.mom6hip_dyn_split_rk2_init.argtypes = [C.c_void_p, cs] + [C.c_void_p] * 5 + [C.c_double]
        L.mom6hip_step_dyn_split_rk2.argtypes = ([C.c_void_p, cs] + [C.c_void_p] * 5 + [C.c_double] + [C.c_void_p] * 2 + [C.c_double]
                                                 + [C.c_void_p] * 5 + [C.c_int32])
        L.mom6hip_dyn_split_rk2b_init.argtypes = [C.c_void_p, cs, C.c_void_p]
        L.mom6hip_step_dyn_split_rk2b.argtypes = L.mom6hip_step_dyn_split_rk2.argtypes
        L._rk2_ready = True
    return L


class MOM_dyn_split_RK2_CS:
    """MOM_dyn_split_RK2_CS (:84-268)."""

    def __init__(self, G: DeviceGrid, BE=0.6, BEGW=0.0, BT_USE_LAYER_FLUXES=True, STORE_CORIOLIS_ACCEL=True, USE_BT_CONT_TYPE=True,
                 EQN_OF_STATE="WRIGHT", continuity=None, coriolis=None, pressure_force=None, barotropic=None, vertvisc=None, hor_visc=None,
                 DT=None, set_visc=None, eos=None, OBC=None):
        g = G.grid
        dev = "cuda"
        self.G = G
        self.continuity_CSp = continuity_PPM_init(G, **(continuity or {}))
        self.CoriolisAdv = CoriolisAdv_init(**(coriolis or {}))
        self.PressureForce_CSp = PressureForce_init(g, **(pressure_force or {}))
        # eos=dict(Rho_T0_S0=..., dRho_dT=..., dRho_dS=...) for LINEAR; EQN_OF_STATE = None: no equation of state (tv%eqn_of_state not
        # associated): the layered PressureForce branch with GV%Rlay / GV%g_prime (pressure_force=dict(Rlay=..., g_prime=...))
        self.eqn_of_state = None if EQN_OF_STATE is None else EOS_init(EQN_OF_STATE, **(eos or {}))
        Z3 = lambda pos: torch.zeros(g.shape3(pos), dtype=torch.float64, device=dev)
        Z2 = lambda pos: torch.zeros(g.shape2(pos), dtype=torch.float64, device=dev)
        self.BT_cont = None
        if USE_BT_CONT_TYPE:      # alloc_BT_cont_type, with h_u / h_v for BT_THICK_SCHEME = FROM_BT_CONT
            self.BT_cont = BT_cont_type(**{n: Z2(_abi.POS_U) for n in _abi.BT_CONT_U}, **{n: Z2(_abi.POS_V) for n in _abi.BT_CONT_V},
                                        h_u=Z3(_abi.POS_U), h_v=Z3(_abi.POS_V))
        bkw = dict(barotropic or {})
        bkw.setdefault("USE_BT_CONT_TYPE", USE_BT_CONT_TYPE)
        self.barotropic_CSp: barotropic_CS = barotropic_init(G, **bkw)
        st = self.st = _abi.DynSplitRK2CS()
        st.be, st.begw = float(BE), float(BEGW)
        st.BT_use_layer_fluxes, st.store_CAu = int(bool(BT_USE_LAYER_FLUXES)), int(bool(STORE_CORIOLIS_ACCEL))
        self._cor_struct = self.CoriolisAdv.struct()
        st.continuity_CSp = C.addressof(self.continuity_CSp); st.CoriolisAdv = C.addressof(self._cor_struct)
        st.PressureForce_CSp = C.addressof(self.PressureForce_CSp)
        st.eqn_of_state = None if self.eqn_of_state is None else C.addressof(self.eqn_of_state)
        st.barotropic_CSp = C.addressof(self.barotropic_CSp.st)
        if self.BT_cont is not None:
            self._bt_struct = self.BT_cont.struct(set())
            st.BT_cont = C.addressof(self._bt_struct)
        self.arrays = {}
        for n, pos in _abi.RK2_ARRAYS_3D:
            self.arrays[n] = Z3(pos); setattr(st, n, self.arrays[n].data_ptr())
        for n, pos in _abi.RK2_ARRAYS_2D:
            self.arrays[n] = Z2(pos); setattr(st, n, self.arrays[n].data_ptr())
        # vertvisc_init (:1500): vertvisc=dict(KV=..., HBBL=..., ...) switches the library's vertical viscosity on
        self.vertvisc_CSp = None
        if vertvisc is not None:
            from .vert_friction import vertvisc_init
            self.vertvisc_CSp = vertvisc_init(G, **vertvisc)
            st.vertvisc_CSp = C.addressof(self.vertvisc_CSp.st)
        # hor_visc_init (:1497): hor_visc=dict(BIHARMONIC=True, AH_VEL_SCALE=..., ...) switches the library's horizontal viscosity on
        self.hor_visc = None
        if hor_visc is not None:
            from .hor_visc import hor_visc_init
            if DT is None:
                raise Mom6HipError("initialize_dyn_split_RK2: hor_visc needs DT (the baroclinic time step)")
            self.hor_visc = hor_visc_init(G, DT, **hor_visc)
            st.hor_visc = C.addressof(self.hor_visc.st)
        # set_visc_CSp (:1500): set_visc=dict(HBBL=..., KV=..., DYNAMIC_VISCOUS_ML=True, ...) makes the step call set_viscous_ML (:592)
        self.set_visc_CSp = None
        if set_visc is not None:
            from .set_viscosity import set_visc_init
            self.set_visc_CSp = set_visc_init(G, **set_visc)
            st.set_visc_CSp = C.addressof(self.set_visc_CSp.st)
        # CS%OBC => OBC (:1516-1519): an ocean_OBC_type whose arrays (the segments' own, rx_normal, ry_normal) are CUDA tensors
        # (ocean_OBC_type.cuda()); the step updates segment%normal_vel and OBC%rx_normal / ry_normal in place
        self.OBC = OBC
        if OBC is not None:
            from .open_boundary import _seg_to_ptr
            for a in [OBC.rx_normal, OBC.ry_normal] + [getattr(sg, k) for sg in OBC.segment if sg.on_pe
                                                        for k in ("normal_vel", "normal_trans", "normal_vel_bt", "SSH", "tangential_vel", "tangential_grad", "nudged_normal_vel")]:
                if a is not None and not hasattr(a, "data_ptr"):
                    raise Mom6HipError("initialize_dyn_split_RK2: the arrays of the OBC must be CUDA tensors (ocean_OBC_type.cuda())")
            self._obc = OBC.struct(_seg_to_ptr(_abi.MEM_DEVICE))
            st.OBC = C.addressof(self._obc)
        self.module_is_initialized = False

    def __getattr__(self, n):
        a = self.__dict__.get("arrays", {})
        if n in a:
            return a[n]
        raise AttributeError(n)


# the restart fields of the split scheme (register_restarts_dyn_split_RK2 :1181-1269, register_barotropic_restarts
# MOM_barotropic.F90:5180-5220): restart name -> where it lives
_RESTART_CS = {"sfc": "eta", "u2": "u_av", "v2": "v_av", "CAu": "CAu_pred", "CAv": "CAv_pred", "h2": "h_av", "diffu": "diffu", "diffv": "diffv"}
_RESTART_BT = {"ubtav": "ubtav", "vbtav": "vbtav"}


def save_restart_dyn_split_RK2(CS: MOM_dyn_split_RK2_CS, uh, vh) -> dict:
    """What register_restarts_dyn_split_RK2 (:1181) and register_barotropic_restarts (MOM_barotropic.F90:5180) put into a
    restart file, as copies: sfc, u2, v2, CAu, CAv (STORE_CORIOLIS_ACCEL), h2, uh, vh, diffu, diffv, ubtav, vbtav, DTBT."""
    r = {n: getattr(CS, a).clone() for n, a in _RESTART_CS.items() if CS.st.store_CAu or n not in ("CAu", "CAv")}
    r.update({n: getattr(CS.barotropic_CSp, a).clone() for n, a in _RESTART_BT.items()})
    r["uh"], r["vh"] = uh.clone(), vh.clone()
    r["DTBT"] = float(CS.barotropic_CSp.st.dtbt)
    r["DTBT_max"] = float(CS.barotropic_CSp.st.dtbt_max)
    return r


def initialize_dyn_split_RK2(u, v, h, uh, vh, dt, G: DeviceGrid, restart=None, **params) -> MOM_dyn_split_RK2_CS:
    """initialize_dyn_split_RK2 (:1326): control structures of the step and of the modules it calls (parameters by
    their reference names, e.g. BE=0.6, barotropic=dict(BEBT=0.1, DTBT=-0.98)), then the state the first step needs.
    restart = the fields of save_restart_dyn_split_RK2: every field present there counts as query_initialized (:1523-1610,
    MOM_barotropic.F90:4895, :5050) and replaces what the cold start computes, uh / vh included."""
    params.setdefault("DT", dt)
    CS = MOM_dyn_split_RK2_CS(G, **params)
    check(_setup().mom6hip_dyn_split_rk2_init(G.handle, C.byref(CS.st), u.data_ptr(), v.data_ptr(), h.data_ptr(), uh.data_ptr(),
                                              vh.data_ptr(), float(dt)), "initialize_dyn_split_RK2")
    if restart is not None:
        for n, a in _RESTART_CS.items():
            if n in restart:
                getattr(CS, a).copy_(restart[n])
        if "CAu" in restart and "CAv" in restart and CS.st.store_CAu:
            CS.st.CAu_pred_stored = 1      # :1561-1563
        for n, a in _RESTART_BT.items():
            if n in restart:
                getattr(CS.barotropic_CSp, a).copy_(restart[n])
        if "uh" in restart:
            uh.copy_(restart["uh"]); vh.copy_(restart["vh"])
        if "DTBT" in restart:      # MOM_barotropic.F90:4895: the restart's DTBT when DTBT_RESET_PERIOD does not force a reset
            CS.barotropic_CSp.st.dtbt = float(restart["DTBT"])
            CS.barotropic_CSp.st.dtbt_max = float(restart.get("DTBT_max", CS.barotropic_CSp.st.dtbt_max))
    CS.module_is_initialized = True
    return CS


# ---- SPLIT_RK2B (src/core/MOM_dynamics_split_RK2b.F90) ------------------------------------------------------------
# restart name -> where it lives (register_restarts_dyn_split_RK2b :1139-1190)
_RESTART_CS_B = {"sfc": "eta", "du_avg_inst": "du_av_inst", "dv_avg_inst": "dv_av_inst"}


def save_restart_dyn_split_RK2b(CS: MOM_dyn_split_RK2_CS) -> dict:
    """What register_restarts_dyn_split_RK2b (:1139) and register_barotropic_restarts put into a restart file: sfc,
    du_avg_inst, dv_avg_inst, ubtav, vbtav, DTBT."""
    r = {n: getattr(CS, a).clone() for n, a in _RESTART_CS_B.items()}
    r.update({n: getattr(CS.barotropic_CSp, a).clone() for n, a in _RESTART_BT.items()})
    r["DTBT"] = float(CS.barotropic_CSp.st.dtbt)
    r["DTBT_max"] = float(CS.barotropic_CSp.st.dtbt_max)
    return r


def initialize_dyn_split_RK2b(u, v, h, uh, vh, dt, G: DeviceGrid, restart=None, **params) -> MOM_dyn_split_RK2_CS:
    """initialize_dyn_split_RK2b (:1220), SPLIT_RK2B = True: the same control structure as the RK2 scheme (the parameters
    STORE_CORIOLIS_ACCEL and BT_USE_LAYER_FLUXES do not exist in this scheme and are not read); eta from h, zero
    barotropic increments, or the fields of save_restart_dyn_split_RK2b."""
    params.setdefault("DT", dt)
    CS = MOM_dyn_split_RK2_CS(G, **params)
    CS.split_RK2b = True
    check(_setup().mom6hip_dyn_split_rk2b_init(G.handle, C.byref(CS.st), h.data_ptr()), "initialize_dyn_split_RK2b")
    if restart is not None:
        for n, a in _RESTART_CS_B.items():
            if n in restart:
                getattr(CS, a).copy_(restart[n])
        for n, a in _RESTART_BT.items():
            if n in restart:
                getattr(CS.barotropic_CSp, a).copy_(restart[n])
        if "DTBT" in restart:
            CS.barotropic_CSp.st.dtbt = float(restart["DTBT"])
            CS.barotropic_CSp.st.dtbt_max = float(restart.get("DTBT_max", CS.barotropic_CSp.st.dtbt_max))
    CS.module_is_initialized = True
    return CS


def step_MOM_dyn_split_RK2b(u_av, v_av, h, tv, visc, Time_local, dt, forces, p_surf_begin, p_surf_end, uh, vh, uhtr, vhtr, eta_av,
                            G: DeviceGrid, CS: MOM_dyn_split_RK2_CS, calc_dtbt=False, VarMix=None, MEKE=None,
                            thickness_diffuse_CSp=None, pbv=None, Waves=None):
    """step_MOM_dyn_split_RK2b(u_av, v_av, h, tv, visc, Time_local, dt, forces, p_surf_begin, p_surf_end, uh, vh, uhtr, vhtr,
    eta_av, G, GV, US, CS, calc_dtbt, VarMix, MEKE, thickness_diffuse_CSp, pbv, Waves) -- MOM_dynamics_split_RK2b.F90:274."""
    if CS is None or not CS.module_is_initialized or not getattr(CS, "split_RK2b", False):
        raise Mom6HipError("step_MOM_dyn_split_RK2b: Module must be initialized before it is used.")
    step_MOM_dyn_split_RK2(u_av, v_av, h, tv, visc, Time_local, dt, forces, p_surf_begin, p_surf_end, uh, vh, uhtr, vhtr, eta_av,
                           G, CS, calc_dtbt=calc_dtbt, VarMix=VarMix, MEKE=MEKE, thickness_diffuse_CSp=thickness_diffuse_CSp,
                           pbv=pbv, Waves=Waves, _entry="mom6hip_step_dyn_split_rk2b")


def step_MOM_dyn_split_RK2(u_inst, v_inst, h, tv, visc, Time_local, dt, forces, p_surf_begin, p_surf_end, uh, vh, uhtr, vhtr, eta_av,
                           G: DeviceGrid, CS: MOM_dyn_split_RK2_CS, calc_dtbt=False, VarMix=None, MEKE=None,
                           thickness_diffuse_CSp=None, pbv=None, STOCH=None, Waves=None, _entry="mom6hip_step_dyn_split_rk2"):
    """step_MOM_dyn_split_RK2(u_inst, v_inst, h, tv, visc, Time_local, dt, forces, p_surf_begin, p_surf_end, uh, vh, uhtr,
    vhtr, eta_av, G, GV, US, CS, calc_dtbt, VarMix, MEKE, thickness_diffuse_CSp, pbv, STOCH, Waves) -- :289.
    tv = (T, S); forces = (taux, tauy) or (taux, tauy, p_surf) with forces%p_surf; p_surf_begin, p_surf_end: the surface pressures at
    the start and the end of the step (device h-point arrays or None: with both, PressureForce takes p_surf_end and btstep the
    interpolated eta_PF :435-442, :497-503); visc is a vert_friction.vertvisc_type (device arrays) when the control structure
    was made with vertvisc=..., else None.  VarMix / MEKE / ... belong to parameterisations this build does not provide
    and must be None."""
    if CS is None or not CS.module_is_initialized:
        raise Mom6HipError("step_MOM_dyn_split_RK2: Module must be initialized before it is used.")
    if Waves is not None or pbv is not None:
        raise Mom6HipError("step_MOM_dyn_split_RK2 (HIP): waves and porous barriers are not supported")
    g = G.grid
    T, S = (tv[0], tv[1]) if tv is not None else (None, None)      # tv None: no equation of state
    if (T is None or S is None) and CS.eqn_of_state is not None:
        raise Mom6HipError("step_MOM_dyn_split_RK2: an equation of state needs tv = (T, S)")
    taux, tauy = forces[0], forces[1]
    p_surf = forces[2] if len(forces) > 2 else None
    for a in (p_surf_begin, p_surf_end, p_surf):
        if a is not None and not (a.is_cuda and a.is_contiguous() and a.dtype == torch.float64 and tuple(a.shape) == tuple(g.shape2(_abi.POS_H))):
            raise Mom6HipError("step_MOM_dyn_split_RK2: a surface pressure must be a contiguous float64 CUDA tensor on the h points")
    CS._p_surf = (p_surf_begin, p_surf_end, p_surf)      # (kept alive for the call)
    CS.st.p_surf_begin = None if p_surf_begin is None else p_surf_begin.data_ptr()
    CS.st.p_surf_end = None if p_surf_end is None else p_surf_end.data_ptr()
    CS.st.p_surf = None if p_surf is None else p_surf.data_ptr()
    if CS.vertvisc_CSp is not None:
        if visc is None:
            raise Mom6HipError("step_MOM_dyn_split_RK2: the control structure has vertical viscosity on, visc is required")
        CS._visc = visc      # keep the arrays and the struct alive
        CS.st.visc = C.addressof(visc.st)
    elif visc is not None:
        raise Mom6HipError("step_MOM_dyn_split_RK2: visc given but the control structure was made without vertvisc=...")
    for a in (u_inst, v_inst, h, T, S, taux, tauy, uh, vh, uhtr, vhtr, eta_av):
        if a is None:
            continue
        if not (a.is_cuda and a.is_contiguous() and a.dtype == torch.float64):
            raise Mom6HipError("step_MOM_dyn_split_RK2: all fields must be contiguous float64 CUDA tensors")
    P = lambda a: None if a is None else C.c_void_p(a.data_ptr())
    if getattr(CS, "split_RK2b", False) != (_entry == "mom6hip_step_dyn_split_rk2b"):
        raise Mom6HipError("step_MOM_dyn_split_RK2: the control structure was initialized for the other split scheme (SPLIT_RK2B)")
    check(getattr(_setup(), _entry)(G.handle, C.byref(CS.st), P(u_inst), P(v_inst), P(h), P(T), P(S), float(dt), P(taux),
                                    P(tauy), g.Z_to_H / g.Rho0, P(uh), P(vh), P(uhtr), P(vhtr), P(eta_av),
                                    int(bool(calc_dtbt))), "step_MOM_dyn_split_RK2")
