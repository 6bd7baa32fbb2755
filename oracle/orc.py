"""ctypes front-end of the CPU oracle (oracle/libmom6oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by mom6_amd."""
import ctypes as C
import os
import subprocess

import numpy as np

from mom6_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_dp = C.POINTER(C.c_double)


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


_THREADS = 1
_LIBS = {}


def set_threads(n):
    """n = 1: the scalar oracle (libmom6oracle.so).  n > 1 (or 0 = all cores): libmom6oracle_omp.so, the same sources
    built with -fopenmp -- the loops the reference marks !$OMP run on n threads, results bit-identical
    (tests/test_oracle_omp.py).  Used by bench.py's all-cores CPU baseline."""
    global _THREADS, _LIB
    n = int(n)
    if n == 0:
        n = len(os.sched_getaffinity(0))
    _THREADS = n
    _LIB = None
    L = lib()
    if n > 1:
        L._gomp.omp_set_num_threads(n)
    return n


def lib():
    global _LIB
    if _LIB is None:
        omp = _THREADS > 1
        if omp in _LIBS:
            _LIB = _LIBS[omp]
            return _LIB
        path = os.path.join(_HERE, "libmom6oracle_omp.so" if omp else "libmom6oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        if omp:
            L._gomp = C.CDLL("libgomp.so.1")
        _LIBS[omp] = L
        L.orc_halo_update.argtypes = [C.POINTER(_abi.GridStruct), _dp, C.c_int, C.c_int]
        L.orc_halo_update.restype = None
        L.orc_advect_tracer.argtypes = [
            C.POINTER(_abi.GridStruct), _dp, _dp, _dp, C.c_double, C.POINTER(_abi.TracerAdvectCS),
            C.POINTER(_dp), _dp, C.c_int, C.c_int, _dp, C.c_int, C.c_int, _dp, _dp,
            C.POINTER(_abi.AdvectStats)]
        L.orc_advect_tracer.restype = C.c_int
        _LIB = L
    return _LIB


def _p(a):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(_dp)


def halo_update(grid, f, pos):
    nk = 1 if f.ndim == 2 else f.shape[0]
    lib().orc_halo_update(C.byref(grid.struct()), _p(f), pos, nk)


def advect_tracer(grid, h_end, uhtr, vhtr, dt, cs_dt, scheme, tr, conc_underflow=None,
                  x_first=None, vol_prev=None, max_iter=None, update_vol_prev=False,
                  uhr_out=None, vhr_out=None, use_huynh_stencil_bug=False, OBC=None):
    """advect_tracer on numpy arrays (tracers updated in place).  Returns AdvectStats."""
    L = lib()
    L.orc_advect_tracer_obc.argtypes = [C.POINTER(_abi.GridStruct), _dp, _dp, _dp, C.c_double, C.POINTER(_abi.TracerAdvectCS), C.POINTER(_dp), _dp,
                                        C.c_int, C.c_int, _dp, C.c_int, C.c_int, _dp, _dp, C.POINTER(_abi.AdvectStats), C.POINTER(_abi.Obc)]
    obc = None if OBC is None else OBC.struct()
    cs = _abi.TracerAdvectCS(float(cs_dt), _abi.ADV_SCHEMES[scheme], int(use_huynh_stencil_bug))
    ntr = len(tr)
    trp = (_dp * ntr)(*[_p(t) for t in tr])
    cu = None if conc_underflow is None else np.ascontiguousarray(conc_underflow, dtype=np.float64)
    st = _abi.AdvectStats()
    rc = L.orc_advect_tracer_obc(
        C.byref(grid.struct()), _p(h_end), _p(uhtr), _p(vhtr), float(dt), C.byref(cs), trp, _p(cu),
        ntr, -1 if x_first is None else int(bool(x_first)), _p(vol_prev),
        0 if max_iter is None else int(max_iter), int(bool(update_vol_prev)), _p(uhr_out),
        _p(vhr_out), C.byref(st), None if obc is None else C.byref(obc))
    if rc != 0:
        raise RuntimeError(f"orc_advect_tracer failed rc={rc}")
    return st


def update_segment_tracer_reservoirs(grid, uhr, vhr, h, OBC, dt, tr):
    """update_segment_tracer_reservoirs on numpy arrays: the reservoirs tr_Reg[m]["tres"] of the segments are updated in place"""
    L = lib()
    L.orc_update_segment_tracer_reservoirs.argtypes = [C.POINTER(_abi.GridStruct), _dp, _dp, _dp, C.POINTER(_abi.Obc), C.c_double, C.POINTER(_dp), C.c_int]
    obc = OBC.struct()
    trp = (_dp * len(tr))(*[_p(t) for t in tr])
    rc = L.orc_update_segment_tracer_reservoirs(C.byref(grid.struct()), _p(uhr), _p(vhr), _p(h), C.byref(obc), float(dt), trp, len(tr))
    if rc != 0:
        raise RuntimeError(f"orc_update_segment_tracer_reservoirs failed rc={rc}")


# ---- ALE reconstruction + remapping ---------------------------------------------------------------
REMAP_SCHEMES = {"PCM": 0, "PLM": 2, "PLM_HYBGEN": 3, "PPM_H4": 4, "PPM_IH4": 5, "PPM_HYBGEN": 6, "WENO_HYBGEN": 7, "PQM_IH4IH3": 8, "PQM_IH6IH5": 9, "PPM_CW": 10}
INT_PCM, INT_PLM, INT_PPM = 0, 1, 3
_ip = C.POINTER(C.c_int)


def _remap_lib():
    L = lib()
    if not getattr(L, "_remap_ready", False):
        d, i = C.c_double, C.c_int
        L.orc_pcm_reconstruction.argtypes = [i, _dp, _dp, _dp]
        L.orc_plm_slope_wa.argtypes = [d] * 7; L.orc_plm_slope_wa.restype = d
        L.orc_plm_monotonized_slope.argtypes = [d] * 6; L.orc_plm_monotonized_slope.restype = d
        L.orc_plm_extrapolate_slope.argtypes = [d] * 5; L.orc_plm_extrapolate_slope.restype = d
        L.orc_plm_reconstruction.argtypes = [i, _dp, _dp, _dp, _dp, d]
        L.orc_plm_boundary_extrapolation.argtypes = [i, _dp, _dp, _dp, _dp, d]
        L.orc_edge_values_explicit_h4.argtypes = [i, _dp, _dp, _dp, d]
        L.orc_ppm_reconstruction.argtypes = [i, _dp, _dp, _dp, _dp]
        L.orc_ppm_boundary_extrapolation.argtypes = [i, _dp, _dp, _dp, _dp, d]
        L.orc_remap_via_sub_cells.argtypes = [i, _dp, _dp, _dp, _dp, i, _dp, i, i, _dp, _dp]
        L.orc_build_reconstructions_1d.argtypes = [i, i, i, _dp, _dp, _dp, _dp, d, d]
        L.orc_build_reconstructions_1d.restype = i
        L.orc_remapping_core_h.argtypes = [i, i, i, _dp, _dp, i, _dp, _dp, d, d]
        L.orc_remapping_core_w.argtypes = [i, i, i, _dp, _dp, i, _dp, _dp, d, d]
        L.orc_dz_from_h1h2.argtypes = [i, _dp, i, _dp, _dp]
        for n in ("orc_pcm_reconstruction", "orc_plm_reconstruction", "orc_plm_boundary_extrapolation",
                  "orc_edge_values_explicit_h4", "orc_ppm_reconstruction", "orc_ppm_boundary_extrapolation",
                  "orc_remap_via_sub_cells", "orc_dz_from_h1h2"):
            getattr(L, n).restype = None
        L._remap_ready = True
    return L


def _a(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def pcm_reconstruction(u):
    u = _a(u); n = len(u); E = np.zeros((2, n)); co = np.zeros((3, n))
    _remap_lib().orc_pcm_reconstruction(n, _p(u), _p(E), _p(co))
    return E, co


def plm_reconstruction(h, u, h_neglect=1e-30, extrapolate=False):
    h, u = _a(h), _a(u); n = len(u); E = np.zeros((2, n)); co = np.zeros((3, n))
    L = _remap_lib()
    L.orc_plm_reconstruction(n, _p(h), _p(u), _p(E), _p(co), h_neglect)
    if extrapolate:
        L.orc_plm_boundary_extrapolation(n, _p(h), _p(u), _p(E), _p(co), h_neglect)
    return E, co


def hybgen_coefs(which, s, h, thin=1e-30):
    """hybgen_plm_coefs (slope [n]) / hybgen_ppm_coefs / hybgen_weno_coefs (edges [2, n]) of MOM_hybgen_remap.F90"""
    s, h = _a(s), _a(h); n = len(s)
    L = _remap_lib()
    for f in (L.orc_hybgen_plm_coefs, L.orc_hybgen_ppm_coefs, L.orc_hybgen_weno_coefs):
        f.argtypes = [C.c_int, _dp, _dp, _dp, C.c_double]; f.restype = None
    out = np.zeros(n) if which == "plm" else np.zeros((2, n))
    {"plm": L.orc_hybgen_plm_coefs, "ppm": L.orc_hybgen_ppm_coefs, "weno": L.orc_hybgen_weno_coefs}[which](n, _p(s), _p(h), _p(out), float(thin))
    return out


def edge_values_explicit_h4(h, u, h_neglect=1e-30):
    h, u = _a(h), _a(u); n = len(u); E = np.zeros((2, n))
    _remap_lib().orc_edge_values_explicit_h4(n, _p(h), _p(u), _p(E), h_neglect)
    return E


def ppm_reconstruction(h, u, E, h_neglect=1e-30, extrapolate=False):
    h, u = _a(h), _a(u); n = len(u); E = _a(E).copy(); co = np.zeros((3, n))
    L = _remap_lib()
    L.orc_ppm_reconstruction(n, _p(h), _p(u), _p(E), _p(co))
    if extrapolate:
        L.orc_ppm_boundary_extrapolation(n, _p(h), _p(u), _p(E), _p(co), h_neglect)
    return E, co


def remap_via_sub_cells(h0, u0, E, coef, h1, method, force_bounds_in_subcell=False):
    h0, u0, h1, E, coef = _a(h0), _a(u0), _a(h1), _a(E), _a(coef)
    u1 = np.zeros(len(h1)); err = C.c_double()
    _remap_lib().orc_remap_via_sub_cells(len(h0), _p(h0), _p(u0), _p(E), _p(coef), len(h1), _p(h1), method,
                                         int(force_bounds_in_subcell), _p(u1), C.cast(C.byref(err), _dp))
    return u1, err.value


def edge_values(kind, h, u, h_neglect=1e-10):
    """edge_values_implicit_h4 ("ih4") / edge_values_explicit_h4cw ("h4cw") -> (left, right) edge values"""
    h, u = _a(h), _a(u); n = len(h); E = np.zeros(2 * n)
    L = _remap_lib()
    f = {"ih4": L.orc_edge_values_implicit_h4, "h4cw": L.orc_edge_values_explicit_h4cw, "h4": L.orc_edge_values_explicit_h4}[kind]
    f.argtypes = [C.c_int, _dp, _dp, _dp, C.c_double]; f.restype = None
    f(n, _p(h), _p(u), _p(E), h_neglect)
    return E[:n].copy(), E[n:].copy()


def remapping_core_h(scheme, h0, u0, h1, h_neglect=1e-30, h_neglect_edge=1e-10, boundary_extrapolation=True):
    h0, u0, h1 = _a(h0), _a(u0), _a(h1); u1 = np.zeros(len(h1))
    rc = _remap_lib().orc_remapping_core_h(REMAP_SCHEMES[scheme], int(boundary_extrapolation), len(h0), _p(h0),
                                          _p(u0), len(h1), _p(h1), _p(u1), h_neglect, h_neglect_edge)
    if rc:
        raise RuntimeError("MOM_remapping, build_reconstructions_1d: The selected remapping method is invalid")
    return u1


def remapping_core_w(scheme, h0, u0, dx, h_neglect=1e-30, h_neglect_edge=1e-10, boundary_extrapolation=True):
    h0, u0, dx = _a(h0), _a(u0), _a(dx); n1 = len(dx) - 1; u1 = np.zeros(n1)
    rc = _remap_lib().orc_remapping_core_w(REMAP_SCHEMES[scheme], int(boundary_extrapolation), len(h0), _p(h0),
                                          _p(u0), n1, _p(dx), _p(u1), h_neglect, h_neglect_edge)
    if rc:
        raise RuntimeError("MOM_remapping, build_reconstructions_1d: The selected remapping method is invalid")
    return u1


def dz_from_h1h2(h1, h2):
    h1, h2 = _a(h1), _a(h2); dx = np.zeros(len(h2) + 1)
    _remap_lib().orc_dz_from_h1h2(len(h1), _p(h1), len(h2), _p(h2), _p(dx))
    return dx


# ---- oracle/_ref: reference sources compiled unmodified (only where they were built) ---------------
def ref_lib():
    path = os.path.join(_HERE, "_ref", "libmom6ref.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    d, i = C.c_double, C.c_int
    L.ref_plm_reconstruction.argtypes = [i, _dp, _dp, _dp, _dp, d, i]; L.ref_plm_reconstruction.restype = None
    L.ref_pcm_reconstruction.argtypes = [i, _dp, _dp, _dp]; L.ref_pcm_reconstruction.restype = None
    L.ref_plm_slope_wa.argtypes = [d] * 7; L.ref_plm_slope_wa.restype = d
    L.ref_plm_monotonized_slope.argtypes = [d] * 6; L.ref_plm_monotonized_slope.restype = d
    L.ref_plm_extrapolate_slope.argtypes = [d] * 5; L.ref_plm_extrapolate_slope.restype = d
    if hasattr(L, "ref_hybgen_plm"):
        L.ref_hybgen_plm.argtypes = [i, _dp, _dp, _dp, d]; L.ref_hybgen_plm.restype = None
        L.ref_hybgen_ppm.argtypes = [i, _dp, _dp, _dp, d]; L.ref_hybgen_ppm.restype = None
        L.ref_hybgen_weno.argtypes = [i, _dp, _dp, _dp, d]; L.ref_hybgen_weno.restype = None
    if hasattr(L, "ref_unesco"):
        L.ref_unesco.argtypes = [i, _dp, _dp, _dp, d, i, _dp, _dp, _dp]; L.ref_unesco.restype = None
    if hasattr(L, "ref_unesco_spv"):
        L.ref_unesco_spv.argtypes = [i, _dp, _dp, _dp, d, _dp]; L.ref_unesco_spv.restype = None
    if hasattr(L, "ref_rotate_array"):
        L.ref_rotate_array.argtypes = [i, i, i, _dp, i, _dp]; L.ref_rotate_array.restype = None
        L.ref_rotate_vector.argtypes = [i] * 5 + [_dp, _dp, i, _dp, _dp]; L.ref_rotate_vector.restype = None
    return L


def ale_remap_tracers(grid, scheme, h_old, h_new, tr, conc_underflow=None, boundary_extrapolation=False):
    L = lib()
    L.orc_ale_remap_tracers.argtypes = [C.POINTER(_abi.GridStruct), C.POINTER(_abi.RemappingCS), _dp, _dp,
                                        C.POINTER(_dp), _dp, C.c_int]
    cs = _abi.RemappingCS(_abi.REMAP_SCHEMES[scheme], int(boundary_extrapolation), 0, 99991231)
    ntr = len(tr)
    trp = (_dp * ntr)(*[_p(t) for t in tr])
    cu = None if conc_underflow is None else np.ascontiguousarray(conc_underflow, dtype=np.float64)
    rc = L.orc_ale_remap_tracers(C.byref(grid.struct()), C.byref(cs), _p(h_old), _p(h_new), trp, _p(cu), ntr)
    if rc:
        raise RuntimeError("orc_ale_remap_tracers failed")


def coradcalc(grid, u, v, h, uh, vh, coriolis_scheme="SADOURNY75_ENERGY", ke_scheme="KE_ARAKAWA", no_slip=False,
              bound_coriolis=False, coriolis_en_dis=False, pv_adv_scheme="PV_ADV_CENTERED", coriolis_blend_wt_lin=0.125,
              coriolis_blend_f_eff_max=4.0, OBC=None):
    """CorAdCalc on numpy arrays; returns (CAu, CAv) (zero outside the computed ranges).  OBC: None or an ocean_OBC_type"""
    L = lib()
    L.orc_coradcalc_obc.argtypes = [C.POINTER(_abi.GridStruct), C.POINTER(_abi.CoriolisAdvCS), C.POINTER(_abi.Obc)] + [_dp] * 7
    if (coriolis_en_dis and coriolis_scheme == "SADOURNY75_ENERGY") or coriolis_scheme == "ROBUST_ENSTRO":
        bound_coriolis = False      # CoriolisAdv_init :1155-1156
    cs = _abi.CoriolisAdvCS(_abi.CORIOLIS_SCHEMES[coriolis_scheme], _abi.KE_SCHEMES[ke_scheme], int(no_slip),
                            int(bound_coriolis), int(bool(coriolis_en_dis)), _abi.PV_ADV_SCHEMES[pv_adv_scheme])
    cs.F_eff_max_blend = float(coriolis_blend_f_eff_max); cs.wt_lin_blend = min(1.0, max(float(coriolis_blend_wt_lin), 1e-16))
    CAu = np.zeros_like(u); CAv = np.zeros_like(v)
    obc = None if OBC is None else OBC.struct()
    rc = L.orc_coradcalc_obc(C.byref(grid.struct()), C.byref(cs), None if obc is None else C.byref(obc), _p(u), _p(v), _p(h), _p(uh), _p(vh),
                             _p(CAu), _p(CAv))
    if rc:
        raise RuntimeError("orc_coradcalc failed")
    return CAu, CAv


def continuity_cs(nk, Angstrom=1e-10, **kw):
    """continuity_PPM_CS with the defaults of continuity_PPM_init (MOM_continuity_PPM.F90:2679-2757)."""
    d = dict(upwind_1st=0, monotonic=0, simple_2nd=0, aggress_adjust=0, vol_CFL=None, better_iter=1,
             use_visc_rem_max=1, marginal_faces=1, tol_eta=0.5 * nk * Angstrom, tol_vel=3.0e8, CFL_limit_adjust=0.5)
    d.update(kw)
    if d["vol_CFL"] is None:
        d["vol_CFL"] = d["aggress_adjust"]
    return _abi.ContinuityCS(*[int(d[n]) for n in ("upwind_1st", "monotonic", "simple_2nd", "aggress_adjust", "vol_CFL",
                                                   "better_iter", "use_visc_rem_max", "marginal_faces")],
                             float(d["tol_eta"]), float(d["tol_vel"]), float(d["CFL_limit_adjust"]))


def make_bt_cont(grid, with_h=False):
    """Zeroed BT_cont arrays (numpy) and the struct pointing at them."""
    arrs = {n: grid.zeros2(_abi.POS_U) for n in _abi.BT_CONT_U}
    arrs.update({n: grid.zeros2(_abi.POS_V) for n in _abi.BT_CONT_V})
    if with_h:
        arrs["h_u"] = grid.zeros3(_abi.POS_U); arrs["h_v"] = grid.zeros3(_abi.POS_V)
    st = _abi.BTCont()
    for n, a in arrs.items():
        setattr(st, n, a.ctypes.data)
    return arrs, st


def continuity(grid, cs, u, v, hin, h, uh, vh, dt, uhbt=None, vhbt=None, visc_rem_u=None, visc_rem_v=None,
               u_cor=None, v_cor=None, bt_cont=None, du_cor=None, dv_cor=None, OBC=None):
    """continuity_PPM; OBC: None (not associated) or a mom6_amd.open_boundary.ocean_OBC_type"""
    L = lib()
    L.orc_continuity_obc.argtypes = ([C.POINTER(_abi.GridStruct), C.POINTER(_abi.ContinuityCS), C.POINTER(_abi.Obc)] + [_dp] * 6 + [C.c_double]
                                     + [_dp] * 6 + [C.POINTER(_abi.BTCont)] + [_dp] * 2)
    obc = None if OBC is None else OBC.struct()
    rc = L.orc_continuity_obc(C.byref(grid.struct()), C.byref(cs), None if obc is None else C.byref(obc), _p(u), _p(v), _p(hin), _p(h), _p(uh),
                              _p(vh), float(dt), _p(uhbt), _p(vhbt), _p(visc_rem_u), _p(visc_rem_v), _p(u_cor), _p(v_cor),
                              None if bt_cont is None else C.byref(bt_cont), _p(du_cor), _p(dv_cor))
    if rc == 3:
        raise RuntimeError("orc_continuity_obc: bad OBC structure")
    if rc:
        raise RuntimeError("MOM_continuity_PPM: Either both visc_rem_u and visc_rem_v or neither one must be present "
                           "in call to continuity_PPM.")


def radiation_open_bdry_conds(grid, OBC, u_new, u_old, v_new, v_old, dt, gamma_uv=0.3, rx_max=1.0, rx_normal=None, ry_normal=None):
    """radiation_open_bdry_conds (the normal component) + apply_normal_flow + the halo update; u_new, v_new, the segments' normal_vel and
    rx_normal / ry_normal are updated in place (OBC_RAD_VEL_WT = 0.3 and OBC_RADIATION_MAX = 1.0 are the reference's defaults)"""
    L = lib()
    L.orc_radiation_open_bdry_conds.argtypes = [C.POINTER(_abi.GridStruct), C.POINTER(_abi.Obc), C.c_double, C.c_double] + [_dp] * 6 + [C.c_double]
    obc = OBC.struct()
    rc = L.orc_radiation_open_bdry_conds(C.byref(grid.struct()), C.byref(obc), float(gamma_uv), float(rx_max), _p(rx_normal), _p(ry_normal), _p(u_new),
                                         _p(u_old), _p(v_new), _p(v_old), float(dt))
    if rc:
        raise RuntimeError(f"orc_radiation_open_bdry_conds rc={rc}")


def open_boundary_zero_normal_flow(grid, OBC, u, v):
    L = lib()
    L.orc_open_boundary_zero_normal_flow.argtypes = [C.POINTER(_abi.GridStruct), C.POINTER(_abi.Obc), _dp, _dp]
    obc = OBC.struct()
    L.orc_open_boundary_zero_normal_flow(C.byref(grid.struct()), C.byref(obc), _p(u), _p(v))


def eos(form="WRIGHT", Rho_T0_S0=1000.0, dRho_dT=-0.2, dRho_dS=0.8):
    return _abi.EOS(_abi.EOS_FORMS[form], 0, Rho_T0_S0, dRho_dT, dRho_dS)


def eos_density(E, T, S, p, rho_ref=None):
    L = lib()
    L.orc_eos_density.argtypes = [C.POINTER(_abi.EOS)] + [C.c_double] * 3; L.orc_eos_density.restype = C.c_double
    L.orc_eos_density_anomaly.argtypes = [C.POINTER(_abi.EOS)] + [C.c_double] * 4
    L.orc_eos_density_anomaly.restype = C.c_double
    if rho_ref is None:
        return L.orc_eos_density(C.byref(E), T, S, p)
    return L.orc_eos_density_anomaly(C.byref(E), T, S, p, rho_ref)


def eos_density_derivs(E, T, S, p):
    L = lib()
    L.orc_eos_density_derivs.argtypes = [C.POINTER(_abi.EOS)] + [C.c_double] * 3 + [_dp, _dp]
    L.orc_eos_density_derivs.restype = None
    a, b = C.c_double(), C.c_double()
    L.orc_eos_density_derivs(C.byref(E), T, S, p, C.cast(C.byref(a), _dp), C.cast(C.byref(b), _dp))
    return a.value, b.value


def pressureforce_cs(grid, Rho0=None, boundary_extrap=True, useMassWghtInterp=False, reconstruct=True, use_ALE=True, nkmb=0,
                     P_Ref=2.0e7, Rlay=None, g_prime=None):
    """PressureForce_FV_CS + the branch selectors of PressureForce_FV_Bouss (use_ALE = associated(ALE_CSp), nkmb =
    GV%nk_rho_varies, tv%P_Ref, GV%Rlay, GV%g_prime)"""
    cs = _abi.PressureForceCS(grid.Rho0 if Rho0 is None else Rho0, 1.0, 0.0, int(reconstruct), 1, int(boundary_extrap),
                              int(useMassWghtInterp), int(use_ALE), int(nkmb), float(P_Ref), None, None)
    cs._keep = []
    for name, a in (("Rlay", Rlay), ("g_prime", g_prime)):
        if a is not None:
            a = np.ascontiguousarray(a, dtype=np.float64); cs._keep.append(a)
            setattr(cs, name, a.ctypes.data)
    return cs


def pressureforce(grid, cs, E, h, T, S, p_atm=None, want_pbce=True, want_eta=True):
    """E None: no equation of state (T, S unused)"""
    L = lib()
    L.orc_pressureforce_fv_bouss.argtypes = [C.POINTER(_abi.GridStruct), C.POINTER(_abi.PressureForceCS),
                                             C.POINTER(_abi.EOS)] + [_dp] * 8
    PFu, PFv = grid.zeros3(_abi.POS_U), grid.zeros3(_abi.POS_V)
    pbce = grid.zeros3(_abi.POS_H) if want_pbce else None
    eta = grid.zeros2(_abi.POS_H) if want_eta else None
    rc = L.orc_pressureforce_fv_bouss(C.byref(grid.struct()), C.byref(cs), C.byref(E) if E is not None else None, _p(h), _p(T), _p(S),
                                      _p(p_atm), _p(PFu), _p(PFv), _p(pbce), _p(eta))
    if rc:
        raise RuntimeError("orc_pressureforce_fv_bouss: unsupported configuration")
    return PFu, PFv, pbce, eta


def ale_plm_edge_values(grid, h, Q, bdry_extrap=False):
    """ALE_PLM_edge_values (MOM_ALE.F90:1520): (Q_t, Q_b) on the columns isc-1..iec+1 x jsc-1..jec+1"""
    L = lib()
    L.orc_ale_plm_edge_values.argtypes = [C.POINTER(_abi.GridStruct), _dp, _dp, C.c_int, _dp, _dp]
    L.orc_ale_plm_edge_values.restype = None
    Qt, Qb = grid.zeros3(_abi.POS_H), grid.zeros3(_abi.POS_H)
    L.orc_ale_plm_edge_values(C.byref(grid.struct()), _p(h), _p(Q), 1 if bdry_extrap else 0, _p(Qt), _p(Qb))
    return Qt, Qb


def pressureforce_nonbouss(grid, cs, E, h, T, S, p_atm=None, H_to_RZ=1.0, want_pbce=True, want_eta=True):
    """PressureForce_FV_nonBouss (h in kg m-2 when H_to_RZ = 1)"""
    L = lib()
    L.orc_pressureforce_fv_nonbouss.argtypes = ([C.POINTER(_abi.GridStruct), C.POINTER(_abi.PressureForceCS), C.POINTER(_abi.EOS)] + [_dp] * 4
                                                + [C.c_double] + [_dp] * 4)
    PFu, PFv = grid.zeros3(_abi.POS_U), grid.zeros3(_abi.POS_V)
    pbce = grid.zeros3(_abi.POS_H) if want_pbce else None
    eta = grid.zeros2(_abi.POS_H) if want_eta else None
    rc = L.orc_pressureforce_fv_nonbouss(C.byref(grid.struct()), C.byref(cs), C.byref(E), _p(h), _p(T), _p(S), _p(p_atm), float(H_to_RZ),
                                         _p(PFu), _p(PFv), _p(pbce), _p(eta))
    if rc:
        raise RuntimeError("orc_pressureforce_fv_nonbouss: unsupported configuration")
    return PFu, PFv, pbce, eta


def eos_spec_vol_anomaly(E, T, S, p, spv_ref):
    L = lib()
    L.orc_eos_spec_vol_anomaly.argtypes = [C.POINTER(_abi.EOS)] + [C.c_double] * 4; L.orc_eos_spec_vol_anomaly.restype = C.c_double
    return L.orc_eos_spec_vol_anomaly(C.byref(E), T, S, p, spv_ref)


# ---- MOM_barotropic -------------------------------------------------------------------------------------
def cr_pow(x, y):
    L = lib(); L.orc_cr_pow.argtypes = [C.c_double, C.c_double]; L.orc_cr_pow.restype = C.c_double
    return L.orc_cr_pow(float(x), float(y))


def barotropic_cs(grid, dtbt=0.0, hvel_scheme="FROM_BT_CONT", **kw):
    """barotropic_CS with the defaults of barotropic_init (MOM_barotropic.F90:4376) and numpy state arrays.
    Returns (struct, arrays); keep `arrays` alive as long as the struct is used."""
    d = dict(dtbt_max=0.0, dtbt_fraction=0.98, bebt=0.1, dt_bt_filter=-0.25, vel_underflow=0.0, G_extra=0.0,
             BT_Coriolis_scale=1.0, Z_ref=0.0, Sadourny=1, linearized_BT_PV=1, strong_drag=0, visc_rem_u_uh0=0,
             adjust_BT_cont=0, use_wide_halos=1, Nonlinear_continuity=0, Nonlin_cont_update_period=1)
    d.update(kw)
    cs = _abi.BarotropicCS()
    cs.dtbt = float(dtbt)
    for k, v in d.items():
        setattr(cs, k, v)
    cs.hvel_scheme = _abi.BT_THICK_SCHEMES[hvel_scheme]
    arrs = {}
    for n, pos, nd in _abi.BT_CS_ARRAYS:
        arrs[n] = grid.zeros3(pos) if nd == 3 else grid.zeros2(pos)
        setattr(cs, n, arrs[n].ctypes.data)
    return cs, arrs


def barotropic_init(grid, cs):
    L = lib(); L.orc_barotropic_init.argtypes = [C.POINTER(_abi.GridStruct), C.POINTER(_abi.BarotropicCS)]
    if L.orc_barotropic_init(C.byref(grid.struct()), C.byref(cs)):
        raise RuntimeError("orc_barotropic_init: unsupported configuration")


def btcalc(grid, cs, h, h_u=None, h_v=None, may_use_default=False, OBC=None):
    L = lib(); L.orc_btcalc_obc.argtypes = [C.POINTER(_abi.GridStruct), C.POINTER(_abi.BarotropicCS), _dp, _dp, _dp, C.c_int, C.POINTER(_abi.Obc)]
    obc = None if OBC is None else OBC.struct()
    if L.orc_btcalc_obc(C.byref(grid.struct()), C.byref(cs), _p(h), _p(h_u), _p(h_v), int(may_use_default), None if obc is None else C.byref(obc)):
        raise RuntimeError("btcalc: Inconsistent settings of optional arguments and hvel_scheme.")


def bt_mass_source(grid, cs, h, eta, set_cor):
    L = lib(); L.orc_bt_mass_source.argtypes = [C.POINTER(_abi.GridStruct), C.POINTER(_abi.BarotropicCS), _dp, _dp, C.c_int]
    L.orc_bt_mass_source(C.byref(grid.struct()), C.byref(cs), _p(h), _p(eta), int(set_cor))


def set_dtbt(grid, cs, pbce=None, bt_cont=None, gtot_est=0.0, SSH_add=0.0, eta=None):
    L = lib()
    L.orc_set_dtbt_eta.argtypes = [C.POINTER(_abi.GridStruct), C.POINTER(_abi.BarotropicCS), _dp, _dp, C.POINTER(_abi.BTCont),
                                   C.c_double, C.c_double]
    L.orc_set_dtbt_eta(C.byref(grid.struct()), C.byref(cs), _p(eta), _p(pbce), None if bt_cont is None else C.byref(bt_cont),
                       float(gtot_est), float(SSH_add))
    return cs.dtbt_max


def btstep(grid, cs, U_in, V_in, eta_in, dt, bc_accel_u, bc_accel_v, taux, tauy, pbce, eta_PF_in, U_Cor, V_Cor,
           visc_rem_u, visc_rem_v, RZ_to_H=None, bt_cont=None, eta_PF_start=None, taux_bot=None, tauy_bot=None, uh0=None,
           vh0=None, u_uh0=None, v_vh0=None, want_etaav=False, OBC=None):
    """btstep on numpy arrays.  Returns dict(accel_layer_u, accel_layer_v, eta_out, uhbtav, vhbtav[, etaav])."""
    L = lib()
    L.orc_btstep_obc.argtypes = ([C.POINTER(_abi.GridStruct), C.POINTER(_abi.BarotropicCS)] + [_dp] * 3 + [C.c_double] + [_dp] * 4
                                 + [C.c_double] + [_dp] * 11 + [C.POINTER(_abi.BTCont)] + [_dp] * 8 + [C.POINTER(_abi.Obc)])
    obc = None if OBC is None else OBC.struct()
    out = dict(accel_layer_u=grid.zeros3(_abi.POS_U), accel_layer_v=grid.zeros3(_abi.POS_V), eta_out=grid.zeros2(_abi.POS_H),
               uhbtav=grid.zeros2(_abi.POS_U), vhbtav=grid.zeros2(_abi.POS_V))
    if want_etaav:
        out["etaav"] = grid.zeros2(_abi.POS_H)
    rz = grid.Z_to_H / grid.Rho0 if RZ_to_H is None else RZ_to_H
    rc = L.orc_btstep_obc(C.byref(grid.struct()), C.byref(cs), _p(U_in), _p(V_in), _p(eta_in), float(dt), _p(bc_accel_u),
                      _p(bc_accel_v), _p(taux), _p(tauy), float(rz), _p(pbce), _p(eta_PF_in), _p(U_Cor), _p(V_Cor),
                      _p(out["accel_layer_u"]), _p(out["accel_layer_v"]), _p(out["eta_out"]), _p(out["uhbtav"]),
                      _p(out["vhbtav"]), _p(visc_rem_u), _p(visc_rem_v), None if bt_cont is None else C.byref(bt_cont),
                      _p(eta_PF_start), _p(taux_bot), _p(tauy_bot), _p(uh0), _p(vh0), _p(u_uh0), _p(v_vh0),
                      _p(out.get("etaav")), None if obc is None else C.byref(obc))
    if rc:
        raise RuntimeError(f"orc_btstep failed rc={rc}")
    return out


def tracer_hordiff(grid, h, dt, tr, KhTr, max_diff_CFL=-1.0, check_diffusive_CFL=False, conc_underflow=None, VarMix=None, MEKE=None, KhTr_Slope_Cff=0.0,
                   KhTr_min=0.0, KhTr_max=0.0, KhTr_passivity_coeff=0.0, KhTr_passivity_min=0.5, neutral=None, epipycnal=None):
    """tracer_hordiff (along-layer; constant KHTR, or with VarMix / MEKE the face diffusivities of :236-281) on numpy arrays; tr updated
    in place.  VarMix: None or a dict with any of L2u, L2v, SN_u, SN_v, Res_fn_h (its presence is Resoln_scaled_KhTr), Rd_dx_h; MEKE: None
    or a dict with Kh and KhTr_fac.  Returns the stats struct."""
    L = lib()
    L.orc_tracer_hordiff_varmix.argtypes = [C.POINTER(_abi.GridStruct), C.POINTER(_abi.TracerHorDiffCS), C.POINTER(_abi.HorDiffFields), _dp, C.c_double,
                                            C.POINTER(_dp), _dp, C.c_int, C.POINTER(_abi.HorDiffStats)]
    cs = _abi.TracerHorDiffCS(); cs.KhTr = float(KhTr); cs.max_diff_CFL = float(max_diff_CFL); cs.check_diffusive_CFL = int(bool(check_diffusive_CFL))
    cs.KhTr_Slope_Cff, cs.KhTr_min, cs.KhTr_max, cs.KhTr_passivity_coeff, cs.KhTr_passivity_min = KhTr_Slope_Cff, KhTr_min, KhTr_max, KhTr_passivity_coeff, KhTr_passivity_min
    F = _abi.HorDiffFields(); keep = []
    if VarMix is not None:
        cs.use_variable_mixing = 1
        cs.Resoln_scaled_KhTr = int("Res_fn_h" in VarMix)
        for n, a in VarMix.items():
            keep.append(np.ascontiguousarray(a, dtype=np.float64)); setattr(F, n, keep[-1].ctypes.data)
    if MEKE is not None:
        cs.KhTr_fac = float(MEKE.get("KhTr_fac", 1.0))
        keep.append(np.ascontiguousarray(MEKE["Kh"], dtype=np.float64)); F.MEKE_Kh = keep[-1].ctypes.data
    ntr = len(tr)
    trp = (_dp * max(ntr, 1))(*[_p(t) for t in tr])
    cu = None if conc_underflow is None else np.ascontiguousarray(conc_underflow, dtype=np.float64)
    st = _abi.HorDiffStats()
    if neutral is not None:      # USE_NEUTRAL_DIFFUSION: dict(eos=, idx_T=, idx_S=, [ref_pres, ndiff_answer_date, recalc_neutral_surf, p_surf, H_to_RZ])
        L.orc_tracer_hordiff_neutral.argtypes = [C.POINTER(_abi.GridStruct), C.POINTER(_abi.TracerHorDiffCS), C.POINTER(_abi.NeutralDiffusionCS),
                                                 C.POINTER(_abi.HorDiffFields), _dp, C.POINTER(_abi.EOS), _dp, C.c_double, C.POINTER(_dp), _dp,
                                                 C.c_int, C.c_int, C.c_int, C.POINTER(_abi.HorDiffStats)]
        cs.unsupported[0] = 1
        nd = neutral_diffusion_cs(grid, **{k: v for k, v in neutral.items() if k not in ("eos", "idx_T", "idx_S", "p_surf", "h_ML")})
        if neutral.get("h_ML") is not None:      # NDIFF_INTERIOR_ONLY: visc%h_ML
            keep.append(np.ascontiguousarray(neutral["h_ML"], dtype=np.float64)); F.h_ML = keep[-1].ctypes.data; nd.interior_only = 1
        ps = neutral.get("p_surf")
        ps = None if ps is None else np.ascontiguousarray(ps, dtype=np.float64)
        rc = L.orc_tracer_hordiff_neutral(C.byref(grid.struct()), C.byref(cs), C.byref(nd), C.byref(F), _p(h), C.byref(neutral["eos"]), _p(ps),
                                          float(dt), trp, _p(cu), ntr, int(neutral.get("idx_T", 0)), int(neutral.get("idx_S", 1)), C.byref(st))
    elif epipycnal is not None:      # DIFFUSE_ML_TO_INTERIOR: dict(eos=, Rlay=, nkml=, nk_rho_varies=, [idx_T, idx_S, ML_KhTr_scale, P_Ref, answer_date, limit_bug])
        L.orc_tracer_hordiff_epipycnal.argtypes = [C.POINTER(_abi.GridStruct), C.POINTER(_abi.TracerHorDiffCS), C.POINTER(_abi.EpipycnalCS),
                                                   C.POINTER(_abi.HorDiffFields), _dp, C.POINTER(_abi.EOS), C.c_double, C.POINTER(_dp), _dp,
                                                   C.c_int, C.c_int, C.c_int, C.POINTER(_abi.HorDiffStats)]
        cs.unsupported[2] = 1
        ep = epipycnal_cs(**{k: v for k, v in epipycnal.items() if k not in ("eos", "idx_T", "idx_S")})
        rc = L.orc_tracer_hordiff_epipycnal(C.byref(grid.struct()), C.byref(cs), C.byref(ep), C.byref(F), _p(h), C.byref(epipycnal["eos"]),
                                            float(dt), trp, _p(cu), ntr, int(epipycnal.get("idx_T", 0)), int(epipycnal.get("idx_S", 1)), C.byref(st))
    else:
        rc = L.orc_tracer_hordiff_varmix(C.byref(grid.struct()), C.byref(cs), C.byref(F), _p(h), float(dt), trp, _p(cu), ntr, C.byref(st))
    if rc:
        raise RuntimeError(f"orc_tracer_hordiff rc={rc}")
    return st


def epipycnal_cs(Rlay, nkml, nk_rho_varies, ML_KhTr_scale=1.0, P_Ref=2.0e7, answer_date=20240101, limit_bug=True):
    """mom6hip_epipycnal_cs_t as tracer_hor_diff_init (:1687-1727) and the vertical grid leave it; Rlay is kept alive on the struct"""
    ep = _abi.EpipycnalCS()
    ep._Rlay = np.ascontiguousarray(Rlay, dtype=np.float64)
    ep.Rlay = ep._Rlay.ctypes.data
    ep.nkml, ep.nk_rho_varies, ep.ML_KhTr_scale, ep.P_Ref = int(nkml), int(nk_rho_varies), float(ML_KhTr_scale), float(P_Ref)
    ep.answer_date, ep.limit_bug = int(answer_date), int(bool(limit_bug))
    return ep


def neutral_diffusion_cs(grid, ref_pres=-1.0, ndiff_answer_date=20240101, recalc_neutral_surf=False, H_to_RZ=None):
    """neutral_diffusion_CS as neutral_diffusion_init leaves it (NDIFF_CONTINUOUS = True)"""
    nd = _abi.NeutralDiffusionCS()
    nd.ref_pres = float(ref_pres); nd.ndiff_answer_date = int(ndiff_answer_date); nd.recalc_neutral_surf = int(bool(recalc_neutral_surf))
    nd.H_to_RZ = float(grid.Rho0 * grid.H_to_Z if H_to_RZ is None else H_to_RZ); nd.initialized = 1
    return nd


def ndiff_fv_diff(*a):
    L = lib(); L.orc_ndiff_fv_diff.argtypes = [C.c_double] * 6; L.orc_ndiff_fv_diff.restype = C.c_double
    return L.orc_ndiff_fv_diff(*[float(x) for x in a])


def ndiff_fvlsq_slope(*a):
    L = lib(); L.orc_ndiff_fvlsq_slope.argtypes = [C.c_double] * 6; L.orc_ndiff_fvlsq_slope.restype = C.c_double
    return L.orc_ndiff_fvlsq_slope(*[float(x) for x in a])


def ndiff_ifndp(*a):
    L = lib(); f = L.orc_ndiff_interpolate_for_nondim_position; f.argtypes = [C.c_double] * 4; f.restype = C.c_double
    return f(*[float(x) for x in a])


def ndiff_interface_scalar(h, S, i_method, h_neglect):
    L = lib(); L.orc_ndiff_interface_scalar.argtypes = [C.c_int, _dp, _dp, _dp, C.c_int, C.c_double]; L.orc_ndiff_interface_scalar.restype = None
    h = np.ascontiguousarray(h, dtype=np.float64); S = np.ascontiguousarray(S, dtype=np.float64); Si = np.zeros(len(h) + 1)
    L.orc_ndiff_interface_scalar(len(h), _p(h), _p(S), _p(Si), int(i_method), float(h_neglect))
    return Si


def ndiff_find_neutral_surface_positions_continuous(Pl, Tl, Sl, dRdTl, dRdSl, Pr, Tr, Sr, dRdTr, dRdSr):
    """-> PoL, PoR, KoL, KoR (1-based, as in the reference), hEff"""
    L = lib(); f = L.orc_ndiff_find_neutral_surface_positions_continuous
    _ip = C.POINTER(C.c_int)
    f.argtypes = [C.c_int] + [_dp] * 10 + [_dp, _dp, _ip, _ip, _dp]; f.restype = None
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (Pl, Tl, Sl, dRdTl, dRdSl, Pr, Tr, Sr, dRdTr, dRdSr)]
    nk = len(a[0]) - 1; ns = 2 * nk + 2
    PoL, PoR, hEff = np.zeros(ns), np.zeros(ns), np.zeros(ns - 1)
    KoL, KoR = np.zeros(ns, dtype=np.int32), np.zeros(ns, dtype=np.int32)
    f(nk, *[_p(x) for x in a], _p(PoL), _p(PoR), KoL.ctypes.data_as(_ip), KoR.ctypes.data_as(_ip), _p(hEff))
    return PoL, PoR, KoL, KoR, hEff


def ndiff_neutral_surface_flux(hl, hr, Tl, Tr, PiL, PiR, KoL, KoR, hEff, h_neglect):
    L = lib(); f = L.orc_ndiff_neutral_surface_flux
    _ip = C.POINTER(C.c_int)
    f.argtypes = [C.c_int] + [_dp] * 6 + [_ip, _ip, _dp, _dp, C.c_double]
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (hl, hr, Tl, Tr, PiL, PiR)]
    KoL = np.ascontiguousarray(KoL, dtype=np.int32); KoR = np.ascontiguousarray(KoR, dtype=np.int32)
    hEff = np.ascontiguousarray(hEff, dtype=np.float64)
    Flx = np.zeros(len(hEff))
    rc = f(len(a[0]), *[_p(x) for x in a], KoL.ctypes.data_as(_ip), KoR.ctypes.data_as(_ip), _p(hEff), _p(Flx), float(h_neglect))
    if rc:
        raise RuntimeError("ppm_ave: dx<0 or dx>1 should not happened!")
    return Flx


# ---- MOM_coms ---------------------------------------------------------------------------------------------
def reproducing_sum(grid, a, pos, by_layer=False, return_err=False):
    """reproducing_sum_3d (MOM_coms.F90:318) over the h-point computational domain of a numpy field of staggering `pos`:
    dict(sum, efp, [sums, efp_lay], err)."""
    L = lib()
    _ip = C.POINTER(C.c_int64)
    L.orc_reproducing_sum_3d.argtypes = [_dp] + [C.c_int] * 7 + [_dp, _dp, _ip, _ip, C.POINTER(C.c_int)]
    a = np.ascontiguousarray(a, dtype=np.float64)
    a3 = a[None] if a.ndim == 2 else a
    ke, ncol, nrow = a3.shape
    xs = 1 if pos in (_abi.POS_U, _abi.POS_Q) else 0
    ys = 1 if pos in (_abi.POS_V, _abi.POS_Q) else 0
    i0, j0 = grid.halo + xs, grid.halo + ys
    s, e = C.c_double(0.0), C.c_int(0)
    tot = (C.c_int64 * 6)()
    lay = (C.c_double * ke)() if by_layer else None
    elay = (C.c_int64 * (6 * ke))() if by_layer else None
    rc = L.orc_reproducing_sum_3d(_p(a3), nrow, ncol, ke, i0, i0 + grid.ni - 1, j0, j0 + grid.nj - 1, C.byref(s), lay, tot, elay,
                                  C.byref(e) if return_err else None)
    if rc:
        raise RuntimeError("orc_reproducing_sum_3d: a term too large, an overflow or a NaN")
    out = dict(sum=float(s.value), efp=[int(x) for x in tot], err=int(e.value))
    if by_layer:
        out["sums"] = [float(x) for x in lay]
        out["efp_lay"] = [[int(elay[6 * k + i]) for i in range(6)] for k in range(ke)]
    return out


# ---- MOM_sum_output ---------------------------------------------------------------------------------------
def write_energy_sums(grid, u, v, h, T, S, dt, C_p=3991.86795711963, H_to_kg_m2=1035.0):
    """the global integrals of write_energy (MOM_sum_output.F90:490-760): dict of totals, by-layer sums and EFP integers"""
    L = lib()
    L.orc_write_energy_sums.argtypes = [C.POINTER(_abi.GridStruct)] + [_dp] * 5 + [C.c_double] * 3 + [_dp, _dp, C.POINTER(_abi.EnergySums)]
    out = _abi.EnergySums()
    ml, kl = np.zeros(grid.nk), np.zeros(grid.nk)
    rc = L.orc_write_energy_sums(C.byref(grid.struct()), _p(u), _p(v), _p(h), _p(T), _p(S), float(dt), float(C_p), float(H_to_kg_m2),
                                 _p(ml), _p(kl), C.byref(out))
    if rc:
        raise RuntimeError("orc_write_energy_sums failed")
    return energy_dict(out, ml, kl)


def write_energy_ape(grid, h, mass_lay, g_prime, Rho0=1035.0, H_to_kg_m2=1035.0, Z_ref=0.0, min_depth_inc=1.0e-10, lH=None):
    """the CALCULATE_APE part of write_energy (:610-680) with its depth list (:1109-1232): dict(PE, PE_tot, Z_0APE)"""
    L = lib()
    _ipt = C.POINTER(C.c_int)
    L.orc_write_energy_ape.argtypes = [C.POINTER(_abi.GridStruct), _dp, _dp, _dp] + [C.c_double] * 4 + [_ipt, _dp, _dp, _dp]
    nz = grid.nk
    PE, Z0, tot = np.zeros(nz + 1), np.zeros(nz + 1), np.zeros(1)
    ml = np.ascontiguousarray(mass_lay, dtype=np.float64); gp = np.ascontiguousarray(g_prime, dtype=np.float64)
    lh = None if lH is None else lH.ctypes.data_as(_ipt)
    rc = L.orc_write_energy_ape(C.byref(grid.struct()), _p(h), _p(ml), _p(gp), float(Rho0), float(H_to_kg_m2), float(Z_ref), float(min_depth_inc),
                                lh, _p(PE), _p(tot), _p(Z0))
    if rc:
        raise RuntimeError("orc_write_energy_ape failed")
    return dict(PE=[float(x) for x in PE], PE_tot=float(tot[0]), Z_0APE=[float(x) for x in Z0])


def depth_list(Dlist, AreaList, min_depth_inc=1.0e-10):
    """create_depth_list :1109-1232 on global lists: (depth, area, vol_below)"""
    L = lib()
    pp = C.POINTER(_dp)
    L.orc_depth_list_create.argtypes = [C.c_int, _dp, _dp, C.c_double, pp, pp, pp]
    L.orc_depth_list_create.restype = C.c_int
    d, a = np.ascontiguousarray(Dlist, dtype=np.float64), np.ascontiguousarray(AreaList, dtype=np.float64)
    o = [_dp(), _dp(), _dp()]
    n = L.orc_depth_list_create(d.size, _p(d), _p(a), float(min_depth_inc), C.byref(o[0]), C.byref(o[1]), C.byref(o[2]))
    out = tuple(np.ctypeslib.as_array(q, shape=(n,)).copy() for q in o)
    libc = C.CDLL(None)
    for q in o:
        libc.free(q)
    return out


def energy_dict(out, mass_lay, KE_lay):
    return dict(mass_tot=out.mass_tot, KE_tot=out.KE_tot, PE_tot=out.PE_tot, toten=out.toten, Salt=out.Salt, Heat=out.Heat,
                max_CFL=[out.max_CFL[0], out.max_CFL[1]], mass_EFP=list(out.mass_EFP), salt_EFP=list(out.salt_EFP),
                heat_EFP=list(out.heat_EFP), npoints=int(out.npoints), mass_lay=[float(x) for x in mass_lay], KE_lay=[float(x) for x in KE_lay])


# ---- MOM_dynamics_split_RK2 -------------------------------------------------------------------------------
class DynState:
    """Everything one oracle run of the split RK2 step owns: sub-module control structures, the control structure of
    the step with numpy arrays behind its pointers, and the prognostic state."""

    def __init__(self, grid, u, v, h, T, S, dt, use_bt_cont=True, be=0.6, BT_use_layer_fluxes=True, store_CAu=True,
                 bound_coriolis=True, dtbt=None, vertvisc=None, visc=None, eos_form="WRIGHT", hor_visc=None, rk2b=False, set_visc=None,
                 pressureforce=None, continuity=None, coriolis=None, OBC=None, **bt_kw):
        """rk2b: SPLIT_RK2B (MOM_dynamics_split_RK2b.F90); u, v are then the filtered velocities.  OBC: CS%OBC (an ocean_OBC_type of
        mom6_amd/open_boundary.py with numpy arrays: its segments' normal_vel and its rx_normal / ry_normal are updated in place)."""
        g = self.grid = grid
        self.rk2b = bool(rk2b)
        self.u, self.v, self.h, self.T, self.S = (np.ascontiguousarray(a).copy() for a in (u, v, h, T, S))
        self.ccs = continuity_cs(g.nk, g.Angstrom_H, **(continuity or {}))      # e.g. tol_eta (ETA_TOLERANCE), tol_vel
        self.cor = _abi.CoriolisAdvCS(_abi.CORIOLIS_SCHEMES["SADOURNY75_ENERGY"], _abi.KE_SCHEMES["KE_ARAKAWA"], 0, int(bound_coriolis), 0)
        self.cor.F_eff_max_blend, self.cor.wt_lin_blend = 4.0, 0.125
        for k, v in (coriolis or {}).items():      # members of mom6hip_coriolisadv_cs_t, e.g. coriolis_en_dis
            setattr(self.cor, k, v)
        self.pcs = pressureforce_cs(g, **(pressureforce or {}))      # e.g. reconstruct=False (RECONSTRUCT_FOR_PRESSURE)
        # eos_form None: no equation of state (the layered PressureForce branch: pressureforce=dict(use_ALE=False, Rlay=..., g_prime=...))
        self.eos = None if eos_form is None else (eos(eos_form) if not isinstance(eos_form, _abi.EOS) else eos_form)
        self.bt_arrs, self.bt = make_bt_cont(g, with_h=True) if use_bt_cont else ({}, None)
        self.bcs, self.bcs_arrs = barotropic_cs(g, hvel_scheme="FROM_BT_CONT" if use_bt_cont else "HARMONIC", **bt_kw)
        barotropic_init(g, self.bcs)
        cs = self.cs = _abi.DynSplitRK2CS()
        cs.be, cs.begw, cs.BT_use_layer_fluxes, cs.store_CAu = be, 0.0, int(BT_use_layer_fluxes), int(store_CAu)
        cs.continuity_CSp = C.addressof(self.ccs); cs.CoriolisAdv = C.addressof(self.cor)
        cs.PressureForce_CSp = C.addressof(self.pcs); cs.eqn_of_state = None if self.eos is None else C.addressof(self.eos)
        cs.barotropic_CSp = C.addressof(self.bcs); cs.BT_cont = C.addressof(self.bt) if use_bt_cont else None
        if vertvisc is not None:      # (vertvisc_cs(...) struct, vertvisc_type(...) struct) of this module
            self.vvcs, self.visc = vertvisc, visc
            cs.vertvisc_CSp = C.addressof(vertvisc); cs.visc = C.addressof(visc)
        if hor_visc is not None:      # a hor_visc_cs(...) struct of this module
            self.hvcs = hor_visc
            cs.hor_visc = C.addressof(hor_visc)
        if set_visc is not None:      # a set_visc_cs(...) struct with dynamic_viscous_ML: set_viscous_ML at :592
            self.svcs = set_visc
            cs.set_visc_CSp = C.addressof(set_visc)
        if OBC is not None:
            self.OBC = OBC
            self.obc = OBC.struct()
            cs.OBC = C.addressof(self.obc)
        self.arrs = {}
        for n, pos in _abi.RK2_ARRAYS_3D:
            self.arrs[n] = grid.zeros3(pos); setattr(cs, n, self.arrs[n].ctypes.data)
        for n, pos in _abi.RK2_ARRAYS_2D:
            self.arrs[n] = grid.zeros2(pos); setattr(cs, n, self.arrs[n].ctypes.data)
        self.uh, self.vh = grid.zeros3(_abi.POS_U), grid.zeros3(_abi.POS_V)
        self.uhtr, self.vhtr = grid.zeros3(_abi.POS_U), grid.zeros3(_abi.POS_V)
        self.eta_av = grid.zeros2(_abi.POS_H)
        self.dt = float(dt)
        L = lib()
        L.orc_dyn_split_rk2_init.argtypes = [C.POINTER(_abi.GridStruct), C.POINTER(_abi.DynSplitRK2CS)] + [_dp] * 5 + [C.c_double]
        L.orc_step_dyn_split_rk2.argtypes = ([C.POINTER(_abi.GridStruct), C.POINTER(_abi.DynSplitRK2CS)] + [_dp] * 5 + [C.c_double]
                                             + [_dp] * 2 + [C.c_double] + [_dp] * 5 + [C.c_int])
        L.orc_dyn_split_rk2b_init.argtypes = [C.POINTER(_abi.GridStruct), C.POINTER(_abi.DynSplitRK2CS), _dp]
        L.orc_step_dyn_split_rk2b.argtypes = L.orc_step_dyn_split_rk2.argtypes
        if self.rk2b:
            rc = L.orc_dyn_split_rk2b_init(C.byref(g.struct()), C.byref(cs), _p(self.h))
        else:
            rc = L.orc_dyn_split_rk2_init(C.byref(g.struct()), C.byref(cs), _p(self.u), _p(self.v), _p(self.h), _p(self.uh), _p(self.vh), self.dt)
        if rc:
            raise RuntimeError(f"orc_dyn_split_rk2_init rc={rc}")
        if dtbt is not None:
            self.bcs.dtbt = float(dtbt)
        self.nsteps = 0

    def step(self, taux, tauy, calc_dtbt=False, p_surf_begin=None, p_surf_end=None, p_surf=None):
        """p_surf_begin, p_surf_end: the step's pointer arguments; p_surf: forces%p_surf (h-point arrays or None) :435-442"""
        g = self.grid
        self._p_surf = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (p_surf_begin, p_surf_end, p_surf)]
        for name, a in zip(("p_surf_begin", "p_surf_end", "p_surf"), self._p_surf):
            setattr(self.cs, name, None if a is None else a.ctypes.data)
        f = lib().orc_step_dyn_split_rk2b if self.rk2b else lib().orc_step_dyn_split_rk2
        rc = f(C.byref(g.struct()), C.byref(self.cs), _p(self.u), _p(self.v), _p(self.h), _p(self.T),
               _p(self.S), self.dt, _p(taux), _p(tauy), g.Z_to_H / g.Rho0, _p(self.uh), _p(self.vh),
               _p(self.uhtr), _p(self.vhtr), _p(self.eta_av), int(calc_dtbt))
        if rc:
            raise RuntimeError(f"orc_step_dyn_split_rk2 rc={rc}")
        self.nsteps += 1


# ---- MOM_vert_friction -------------------------------------------------------------------------------------------
def vertvisc_cs(grid, Kv, Hbbl, Hmix=0.0, bottomdraglaw=True, harmonic_visc=False, harm_BL_val=0.0, direct_stress=False,
                Hmix_stress=None, Kvml_invZ2=0.0, Kv_extra_bbl=0.0, maxvel=3.0e8, CFL_based_trunc=True, CFL_trunc=0.5,
                vel_underflow=0.0, answer_date=99991231, dynamic_viscous_ML=False, nkml=0, vonKar=0.41):
    """mom6hip_vertvisc_cs_t with numpy arrays behind a_u, a_v, h_u, h_v (kept on the struct as ._arrs)."""
    cs = _abi.VertviscCS()
    cs.dynamic_viscous_ML, cs.nkml, cs.vonKar = int(dynamic_viscous_ML), int(nkml), float(vonKar)
    cs.Kv, cs.Hbbl, cs.Hmix = Kv, Hbbl, Hmix
    cs.bottomdraglaw, cs.harmonic_visc, cs.direct_stress = int(bottomdraglaw), int(harmonic_visc), int(direct_stress)
    cs.harm_BL_val = harm_BL_val
    cs.Hmix_stress = (Hmix_stress if Hmix_stress is not None else Hmix) * grid.Z_to_H
    cs.Kvml_invZ2, cs.Kv_extra_bbl = Kvml_invZ2, Kv_extra_bbl
    cs.maxvel, cs.CFL_based_trunc, cs.CFL_trunc, cs.vel_underflow, cs.answer_date = maxvel, int(CFL_based_trunc), CFL_trunc, vel_underflow, answer_date
    cs.H_to_RZ = grid.Rho0 * grid.H_to_Z
    cs._arrs = {}
    for n, pos, extra in _abi.VERTVISC_CS_ARRAYS:
        shp = grid.shape3(pos)
        cs._arrs[n] = np.zeros((shp[0] + extra,) + tuple(shp[1:]))
        setattr(cs, n, cs._arrs[n].ctypes.data)
    return cs


def vertvisc_type(**arrays):
    vt = _abi.VertviscType()
    vt._keep = {n: np.ascontiguousarray(a) for n, a in arrays.items() if a is not None}
    for n, a in vt._keep.items():
        setattr(vt, n, a.ctypes.data)
    return vt


def vertvisc_coef(grid, cs, u, v, h, visc, dt, dz=None, OBC=None):
    L = lib(); L.orc_vertvisc_coef_obc.argtypes = [C.POINTER(_abi.GridStruct), C.POINTER(_abi.VertviscCS)] + [_dp] * 4 + [C.POINTER(_abi.VertviscType), C.c_double,
                                                                                                                    C.POINTER(_abi.Obc)]
    obc = None if OBC is None else OBC.struct()
    rc = L.orc_vertvisc_coef_obc(C.byref(grid.struct()), C.byref(cs), _p(u), _p(v), _p(h), None if dz is None else _p(dz), C.byref(visc), dt,
                                 None if obc is None else C.byref(obc))
    if rc:
        raise RuntimeError(f"orc_vertvisc_coef rc={rc}")


def vertvisc(grid, cs, u, v, h, taux, tauy, visc, dt, taux_bot=None, tauy_bot=None, OBC=None):
    L = lib(); L.orc_vertvisc_obc.argtypes = ([C.POINTER(_abi.GridStruct), C.POINTER(_abi.VertviscCS)] + [_dp] * 5
                                             + [C.POINTER(_abi.VertviscType), C.c_double, _dp, _dp, C.POINTER(_abi.Obc)])
    obc = None if OBC is None else OBC.struct()
    rc = L.orc_vertvisc_obc(C.byref(grid.struct()), C.byref(cs), _p(u), _p(v), _p(h), _p(taux), _p(tauy), C.byref(visc), dt,
                            None if taux_bot is None else _p(taux_bot), None if tauy_bot is None else _p(tauy_bot),
                            None if obc is None else C.byref(obc))
    if rc:
        raise RuntimeError(f"orc_vertvisc rc={rc}")


def vertvisc_remnant(grid, cs, visc, visc_rem_u, visc_rem_v, dt):
    L = lib(); L.orc_vertvisc_remnant.argtypes = [C.POINTER(_abi.GridStruct), C.POINTER(_abi.VertviscCS), C.POINTER(_abi.VertviscType), _dp, _dp, C.c_double]
    rc = L.orc_vertvisc_remnant(C.byref(grid.struct()), C.byref(cs), C.byref(visc), _p(visc_rem_u), _p(visc_rem_v), dt)
    if rc:
        raise RuntimeError(f"orc_vertvisc_remnant rc={rc}")


# ---- MOM_set_viscosity -------------------------------------------------------------------------------------------
def set_visc_cs(grid, Hbbl, Kv, cdrag=0.003, drag_bg_vel=0.0, BBL_thick_min=0.0, Kv_BBL_min=None, bottomdraglaw=True, linear_drag=False,
                BBL_use_EOS=True, correct_BBL_bounds=False, body_force_drag=False, RiNo_mix=False, Rlay=None, dynamic_viscous_ML=False,
                nkml=0, bulk_Ri_ML=0.0, TKE_decay=0.0, omega_frac=0.0, omega=7.2921e-5, Channel_drag=False, c_Smag=0.15, Chan_drag_max_vol=None,
                concave_trigonometric_L=True, Z_ref=0.0, **unsupported):
    """mom6hip_set_visc_cs_t with the defaults of set_visc_init (MOM_set_viscosity.F90:2886-3190)."""
    cs = _abi.SetViscCS()
    cs.dynamic_viscous_ML, cs.nkml, cs.bulk_Ri_ML, cs.TKE_decay, cs.omega_frac, cs.omega = (int(dynamic_viscous_ML), int(nkml), bulk_Ri_ML,
                                                                                          TKE_decay, omega_frac, omega)
    cs.ustar_min = 2e-4 * omega * (grid.Angstrom_H + grid.H_subroundoff)      # :2998
    # CHANNEL_DRAG (:3092-3122): CHANNEL_DRAG_MAX_BBL_THICK defaults to HBBL/2 with kappa shear, HBBL with DRAG_AS_BODY_FORCE, else none
    if Chan_drag_max_vol is None:
        Chan_drag_max_vol = Hbbl if body_force_drag else (0.5 * Hbbl if RiNo_mix else -1.0)
    cs.Channel_drag, cs.c_Smag, cs.Chan_drag_max_vol, cs.concave_trigonometric_L, cs.Z_ref = (int(Channel_drag), c_Smag, Chan_drag_max_vol,
                                                                                            int(concave_trigonometric_L), Z_ref)
    cs.cdrag, cs.drag_bg_vel, cs.dz_bbl, cs.Hbbl = cdrag, drag_bg_vel, Hbbl, Hbbl * grid.Z_to_H
    cs.BBL_thick_min, cs.Kv_BBL_min, cs.BBL_thick_max = BBL_thick_min, (Kv if Kv_BBL_min is None else Kv_BBL_min), 6.378e6
    cs.H_to_RZ = grid.Rho0 * grid.H_to_Z
    cs.bottomdraglaw, cs.linear_drag, cs.BBL_use_EOS = int(bottomdraglaw), int(linear_drag), int(BBL_use_EOS)
    cs.correct_BBL_bounds, cs.body_force_drag, cs.RiNo_mix, cs.initialized = int(correct_BBL_bounds), int(body_force_drag), int(RiNo_mix), 1
    for k, v in unsupported.items():
        cs.unsupported[_abi.SET_VISC_UNSUPPORTED.index(k)] = int(bool(v))
    if Rlay is not None:
        cs._rlay = np.ascontiguousarray(Rlay, dtype=np.float64)
        cs.Rlay = cs._rlay.ctypes.data
    return cs


def set_viscous_BBL(grid, cs, u, v, h, T, S, E, visc, OBC=None):
    L = lib(); L.orc_set_viscous_BBL_obc.argtypes = ([C.POINTER(_abi.GridStruct), C.POINTER(_abi.SetViscCS)] + [_dp] * 5
                                                    + [C.POINTER(_abi.EOS), C.POINTER(_abi.VertviscType), C.POINTER(_abi.Obc)])
    obc = None if OBC is None else OBC.struct()
    rc = L.orc_set_viscous_BBL_obc(C.byref(grid.struct()), C.byref(cs), _p(u), _p(v), _p(h), None if T is None else _p(T),
                                   None if S is None else _p(S), None if E is None else C.byref(E), C.byref(visc),
                                   None if obc is None else C.byref(obc))
    if rc:
        raise RuntimeError(f"orc_set_viscous_BBL rc={rc}")


def set_viscous_ML(grid, cs, u, v, h, T, S, E, taux, tauy, visc, dt):
    """set_viscous_ML (:1898): visc must carry ustar and the (written) nkml_visc_u / nkml_visc_v"""
    L = lib(); L.orc_set_viscous_ML.argtypes = ([C.POINTER(_abi.GridStruct), C.POINTER(_abi.SetViscCS)] + [_dp] * 5 + [C.POINTER(_abi.EOS)]
                                               + [_dp, _dp, C.POINTER(_abi.VertviscType), C.c_double])
    rc = L.orc_set_viscous_ML(C.byref(grid.struct()), C.byref(cs), _p(u), _p(v), _p(h), _p(T), _p(S), None if E is None else C.byref(E),
                              _p(taux), _p(tauy), C.byref(visc), float(dt))
    if rc:
        raise RuntimeError(f"orc_set_viscous_ML rc={rc}")


def cr_exp(t):
    L = lib(); L.orc_cr_exp.argtypes = [C.c_double]; L.orc_cr_exp.restype = C.c_double
    return L.orc_cr_exp(float(t))


def cr_cos(x):
    L = lib(); L.orc_cr_cos.argtypes = [C.c_double]; L.orc_cr_cos.restype = C.c_double
    return L.orc_cr_cos(float(x))


def cr_acos(x):
    L = lib(); L.orc_cr_acos.argtypes = [C.c_double]; L.orc_cr_acos.restype = C.c_double
    return L.orc_cr_acos(float(x))


# ---- MOM_hor_visc ---------------------------------------------------------------------------------------------------
def hor_visc_cs(grid, dt, **kw):
    """mom6hip_hor_visc_cs_t with the defaults of hor_visc_init (MOM_hor_visc.F90:2062-2300), numpy arrays behind its
    pointers (kept on the struct as ._arrs), initialised by orc_hor_visc_init."""
    d = dict(Kh=0.0, Kh_bg_min=0.0, Kh_vel_scale=0.0, Smag_Lap_const=0.0, Ah=0.0, Ah_vel_scale=0.0, Ah_time_scale=0.0, Smag_bi_const=0.0,
             bound_Cor_vel=3.0e8, bound_coef=0.8, Laplacian=0, biharmonic=1, Smagorinsky_Kh=0, Smagorinsky_Ah=0, bound_Kh=1,
             better_bound_Kh=None, bound_Ah=1, better_bound_Ah=None, bound_Coriolis=0, add_LES_viscosity=0, no_slip=0, use_land_mask=1,
             use_cont_thick=0)
    unsupported = {k: kw.pop(k) for k in list(kw) if k in _abi.HOR_VISC_UNSUPPORTED}
    d.update(kw)
    if d["better_bound_Kh"] is None:
        d["better_bound_Kh"] = d["bound_Kh"]
    if d["better_bound_Ah"] is None:
        d["better_bound_Ah"] = d["bound_Ah"]
    if not d["Smagorinsky_Ah"]:
        d["bound_Coriolis"] = 0
    cs = _abi.HorViscCS()
    for k, v in d.items():
        setattr(cs, k, float(v) if isinstance(getattr(cs, k), float) else int(bool(v)))
    for k, v in unsupported.items():
        cs.unsupported[_abi.HOR_VISC_UNSUPPORTED.index(k)] = int(bool(v))
    cs._arrs = {}
    for n in _abi.HOR_VISC_ARRAYS_H:
        cs._arrs[n] = grid.zeros2(_abi.POS_H); setattr(cs, n, cs._arrs[n].ctypes.data)
    for n in _abi.HOR_VISC_ARRAYS_Q:
        cs._arrs[n] = grid.zeros2(_abi.POS_Q); setattr(cs, n, cs._arrs[n].ctypes.data)
    L = lib(); L.orc_hor_visc_init.argtypes = [C.POINTER(_abi.GridStruct), C.POINTER(_abi.HorViscCS), C.c_double]
    rc = L.orc_hor_visc_init(C.byref(grid.struct()), C.byref(cs), float(dt))
    if rc:
        raise RuntimeError(f"orc_hor_visc_init rc={rc}: an option of MOM_hor_visc that is not provided")
    return cs


def hor_visc_set_meke(cs, Ku=None, Au=None, mom_src=None):
    """the MEKE argument of horizontal_viscosity: MEKE%Ku, MEKE%Au, MEKE%mom_src (numpy h-point 2-D arrays or None)"""
    cs._meke = [None if a is None else np.ascontiguousarray(a) for a in (Ku, Au, mom_src)]
    for n, a in zip(("MEKE_Ku", "MEKE_Au", "MEKE_mom_src"), cs._meke):
        setattr(cs, n, None if a is None else a.ctypes.data)
    return cs._meke[2]


def horizontal_viscosity(grid, cs, u, v, h, dt, hu_cont=None, hv_cont=None, diffu=None, diffv=None, OBC=None):
    L = lib(); L.orc_horizontal_viscosity_obc.argtypes = ([C.POINTER(_abi.GridStruct), C.POINTER(_abi.HorViscCS)] + [_dp] * 5
                                                         + [C.c_double, _dp, _dp, C.POINTER(_abi.Obc)])
    diffu = grid.zeros3(_abi.POS_U) if diffu is None else diffu
    diffv = grid.zeros3(_abi.POS_V) if diffv is None else diffv
    obc = None if OBC is None else OBC.struct()
    rc = L.orc_horizontal_viscosity_obc(C.byref(grid.struct()), C.byref(cs), _p(u), _p(v), _p(h), _p(diffu), _p(diffv), float(dt),
                                        _p(hu_cont), _p(hv_cont), None if obc is None else C.byref(obc))
    if rc:
        raise RuntimeError(f"orc_horizontal_viscosity rc={rc}")
    return diffu, diffv


# ---- z* regridding + velocity remapping ---------------------------------------------------------------------
def regridding_cs(res, min_thickness=1.0e-3, old_grid_weight=0.0, zs=0.0, zd=0.0, Z_ref=0.0):
    res = np.ascontiguousarray(res, dtype=np.float64)
    cs = _abi.RegriddingCS(_abi.REGRIDDING_ZSTAR, int(res.size), min_thickness, old_grid_weight, zs, zd, Z_ref, res.ctypes.data)
    cs._keep = res
    return cs


def ale_regrid(grid, cs, h):
    L = lib(); L.orc_ale_regrid.argtypes = [C.POINTER(_abi.GridStruct), C.POINTER(_abi.RegriddingCS), _dp, _dp, _dp]
    h_new = grid.zeros3(_abi.POS_H); dz = np.zeros((grid.nk + 1,) + grid.shape2(_abi.POS_H))
    rc = L.orc_ale_regrid(C.byref(grid.struct()), C.byref(cs), _p(h), _p(h_new), _p(dz))
    if rc:
        raise RuntimeError(f"orc_ale_regrid rc={rc}")
    return h_new, dz


def ale_remap_set_h_vel(grid, h_new, h_u=None, h_v=None):
    L = lib(); L.orc_ale_remap_set_h_vel.argtypes = [C.POINTER(_abi.GridStruct), _dp, _dp, _dp]
    h_u = grid.zeros3(_abi.POS_U) if h_u is None else h_u
    h_v = grid.zeros3(_abi.POS_V) if h_v is None else h_v
    L.orc_ale_remap_set_h_vel(C.byref(grid.struct()), _p(h_new), _p(h_u), _p(h_v))
    return h_u, h_v


def ale_remap_set_h_vel_via_dz(grid, h_old, dzInterface, h_u=None, h_v=None):
    """ALE_remap_set_h_vel_via_dz, MOM_ALE.F90:912 (REMAP_UV_USING_OLD_ALG)"""
    L = lib(); L.orc_ale_remap_set_h_vel_via_dz.argtypes = [C.POINTER(_abi.GridStruct), _dp, _dp, _dp, _dp]
    h_u = grid.zeros3(_abi.POS_U) if h_u is None else h_u
    h_v = grid.zeros3(_abi.POS_V) if h_v is None else h_v
    L.orc_ale_remap_set_h_vel_via_dz(C.byref(grid.struct()), _p(h_old), _p(dzInterface), _p(h_u), _p(h_v))
    return h_u, h_v


def ale_remap_velocities(grid, scheme, h_old_u, h_old_v, h_new_u, h_new_v, u, v, boundary_extrapolation=False):
    L = lib(); L.orc_ale_remap_velocities.argtypes = [C.POINTER(_abi.GridStruct), C.POINTER(_abi.RemappingCS)] + [_dp] * 6
    cs = _abi.RemappingCS(REMAP_SCHEMES[scheme], int(boundary_extrapolation), 0, 99991231)
    rc = L.orc_ale_remap_velocities(C.byref(grid.struct()), C.byref(cs), _p(h_old_u), _p(h_old_v), _p(h_new_u), _p(h_new_v), _p(u), _p(v))
    if rc:
        raise RuntimeError(f"orc_ale_remap_velocities rc={rc}")


# ---- MOM_thickness_diffuse ----------------------------------------------------------------------------------------
def thickness_diffuse_cs(grid, Khth=0.0, Khth_Min=0.0, Khth_Max=0.0, max_Khth_CFL=0.8, slope_max=0.01, kappa_smooth=1.0e-6, KHTH_Slope_Cff=0.0,
                         KhTh_fac=1.0, thickness_diffuse=True, use_GM_work_bug=False, nkml=0, use_variable_mixing=False, use_FGNV_streamfn=False,
                         FGNV_scale=1.0, FGNV_strat_floor=1.0e-15, omega=7.2921e-5, **fields):
    """mom6hip_thickness_diffuse_cs_t with the defaults of thickness_diffuse_init (MOM_thickness_diffuse.F90:2169-2400); fields: the
    arrays of MEKE / VarMix by the struct's member names (MEKE_Kh, L2u, ..., slope_x, slope_y, MEKE_GM_src, Rlay) or the names of
    _abi.THICKNESS_DIFFUSE_UNSUPPORTED set to True"""
    cs = _abi.ThicknessDiffuseCS()
    cs.Khth, cs.Khth_Min, cs.Khth_Max, cs.max_Khth_CFL, cs.slope_max = Khth, Khth_Min, Khth_Max, max_Khth_CFL, slope_max
    cs.kappa_smooth, cs.KHTH_Slope_Cff, cs.KhTh_fac = kappa_smooth, KHTH_Slope_Cff, KhTh_fac
    cs.thickness_diffuse, cs.use_GM_work_bug, cs.nkml, cs.use_variable_mixing = int(thickness_diffuse), int(use_GM_work_bug), int(nkml), int(use_variable_mixing)
    cs.use_FGNV_streamfn, cs.FGNV_scale = int(use_FGNV_streamfn), float(FGNV_scale)
    cs.N2_floor = (FGNV_strat_floor * omega) ** 2 if use_FGNV_streamfn else 0.0      # :2337
    cs.initialized = 1
    cs._keep = {}
    for n, a in fields.items():
        if n in _abi.THICKNESS_DIFFUSE_UNSUPPORTED:
            cs.unsupported[_abi.THICKNESS_DIFFUSE_UNSUPPORTED.index(n)] = int(bool(a))
        elif n in _abi.THICKNESS_DIFFUSE_FIELDS:
            if a is not None:
                cs._keep[n] = np.ascontiguousarray(a, dtype=np.float64)
                setattr(cs, n, cs._keep[n].ctypes.data)
        else:
            raise ValueError(n)
    return cs


def thickness_diffuse(grid, cs, h, uhtr, vhtr, T, S, E, dt, uhGM=None, vhGM=None):
    """thickness_diffuse (h, uhtr, vhtr updated in place); E None: no equation of state"""
    L = lib()
    L.orc_thickness_diffuse.argtypes = [C.POINTER(_abi.GridStruct), C.POINTER(_abi.ThicknessDiffuseCS)] + [_dp] * 5 + [C.POINTER(_abi.EOS), C.c_double, _dp, _dp]
    rc = L.orc_thickness_diffuse(C.byref(grid.struct()), C.byref(cs), _p(h), _p(uhtr), _p(vhtr), _p(T), _p(S), None if E is None else C.byref(E),
                                 float(dt), _p(uhGM), _p(vhGM))
    if rc:
        raise RuntimeError(f"orc_thickness_diffuse rc={rc}")


# ---- MOM_mixed_layer_restrat --------------------------------------------------------------------------------------
def mle_mu(sigma, dh):
    """mu(sigma, dh), MOM_mixed_layer_restrat.F90:723"""
    L = lib()
    L.orc_mle_mu.restype = C.c_double
    L.orc_mle_mu.argtypes = [C.c_double, C.c_double]
    return L.orc_mle_mu(float(sigma), float(dh))


def mixedlayer_restrat_cs(grid, ml_restrat_coef=0.0, ml_restrat_coef2=0.0, front_length=0.0, vonKar=0.41, MLE_MLD_decay_time=0.0, MLE_MLD_decay_time2=0.0,
                          MLE_density_diff=0.03, MLE_tail_dh=0.0, MLE_MLD_stretch=1.0, ustar_min=None, MLE_use_PBL_MLD=False, nkml=0, omega=7.2921e-5,
                          **fields):
    """mom6hip_mixedlayer_restrat_cs_t with the defaults of mixedlayer_restrat_init (MOM_mixed_layer_restrat.F90:1532-1735); fields:
    MLD_filtered, MLD_filtered_slow, Rd_dx_h (arrays, kept and updated in place) or the names of _abi.MIXEDLAYER_RESTRAT_UNSUPPORTED"""
    cs = _abi.MixedlayerRestratCS()
    cs.ml_restrat_coef, cs.ml_restrat_coef2, cs.front_length, cs.vonKar = ml_restrat_coef, ml_restrat_coef2, front_length, vonKar
    cs.MLE_MLD_decay_time, cs.MLE_MLD_decay_time2, cs.MLE_tail_dh, cs.MLE_MLD_stretch = MLE_MLD_decay_time, MLE_MLD_decay_time2, MLE_tail_dh, MLE_MLD_stretch
    cs.MLE_density_diff = -9.0e9 if MLE_use_PBL_MLD else MLE_density_diff      # :1568: not read with MLE_USE_PBL_MLD
    # RESTRAT_USTAR_MIN :1728-1733: 2e-4 * OMEGA * (GV%Angstrom_Z + GV%dZ_subroundoff) [Z T-1] -> [H T-1]
    cs.ustar_min = (2.0e-4 * omega * (grid.Angstrom_H * grid.H_to_Z + grid.dZ_subroundoff) if ustar_min is None else ustar_min) * grid.Z_to_H
    cs.MLE_use_PBL_MLD, cs.nkml, cs.initialized = int(MLE_use_PBL_MLD), int(nkml), 1
    cs._keep = {}
    for n, a in fields.items():
        if n in _abi.MIXEDLAYER_RESTRAT_UNSUPPORTED:
            cs.unsupported[_abi.MIXEDLAYER_RESTRAT_UNSUPPORTED.index(n)] = int(bool(a))
        elif n in _abi.MIXEDLAYER_RESTRAT_FIELDS:
            if a is not None:
                assert a.dtype == np.float64 and a.flags.c_contiguous
                cs._keep[n] = a
                setattr(cs, n, a.ctypes.data)
        else:
            raise ValueError(n)
    return cs


def mixedlayer_restrat(grid, cs, h, uhtr, vhtr, T, S, E, ustar, dt, h_MLD=None, uhml=None, vhml=None):
    """mixedlayer_restrat (h, uhtr, vhtr and the filtered depths of cs updated in place)"""
    L = lib()
    L.orc_mixedlayer_restrat.argtypes = [C.POINTER(_abi.GridStruct), C.POINTER(_abi.MixedlayerRestratCS)] + [_dp] * 5 + [C.POINTER(_abi.EOS), _dp, C.c_double, _dp, _dp, _dp]
    rc = L.orc_mixedlayer_restrat(C.byref(grid.struct()), C.byref(cs), _p(h), _p(uhtr), _p(vhtr), _p(T), _p(S), None if E is None else C.byref(E),
                                  _p(ustar), float(dt), _p(h_MLD), _p(uhml), _p(vhml))
    if rc:
        raise RuntimeError(f"orc_mixedlayer_restrat rc={rc}")
