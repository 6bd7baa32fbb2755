"""The hot-path parameter sets of the reference's own test configurations (.testing/tc4, tc2, tc1: MOM_input, transcribed below as
KEY = VALUE pairs -- data, not source) through the Fortran shims' get_param, from a driver that calls only reference-named procedures
(tests/fortran/dyn_driver.F90: set_visc_init, register_restarts_dyn_split_RK2, initialize_dyn_split_RK2, set_viscous_BBL,
step_MOM_dyn_split_RK2) on the configurations' grid sizes (14 x 10 x 2, 10 x 8 x 8), against the oracle configured by hand from the same
pairs with the reference's defaults (the function `oracle_for` below cites where each default comes from)."""
import os
import subprocess

import numpy as np
import pytest

from mom6_amd import _abi, synth
from helpers import bits_equal, interior
from test_fortran_abi import FC, ROOT, FDIR, STUBS, SHIMS

# ---- transcribed from .testing/tc*/MOM_input: every pair a hot-path module reads, plus the switches of neighbouring modules that
# decide what the hot path is handed (USE_REGRIDDING, THICKNESSDIFFUSE, USE_MEKE ...) ------------------------------------------------------
TC_INPUT = {
    "tc4": dict(shape=(14, 10, 2), pairs="""
        USE_REGRIDDING = True
        DT = 1200.0
        DT_THERM = 3600.0
        USE_PSURF_IN_EOS = False
        REENTRANT_X = False
        EQN_OF_STATE = "LINEAR"
        DRHO_DS = 0.0
        REMAP_UV_USING_OLD_ALG = True
        REGRIDDING_COORDINATE_MODE = "Z*"
        REMAPPING_SCHEME = "PPM_IH4"
        LINEAR_DRAG = True
        HBBL = 10.0
        CDRAG = 0.002
        DRAG_BG_VEL = 0.05
        BBL_USE_EOS = True
        BBL_THICK_MIN = 0.1
        KV = 1.0E-04
        KHTH = 500.0
        USE_GM_WORK_BUG = True
        BE = 0.7
        ETA_TOLERANCE = 1.0E-12
        CORIOLIS_EN_DIS = True
        BOUND_CORIOLIS = True
        RECONSTRUCT_FOR_PRESSURE = False
        SMAGORINSKY_AH = True
        SMAG_BI_CONST = 0.03
        USE_LAND_MASK_FOR_HVISC = False
        DIRECT_STRESS = True
        HMIX_FIXED = 20.0
        KV_ML_INVZ2 = 0.01
        MAXVEL = 10.0
        BOUND_BT_CORRECTION = True
        SSH_EXTRA = 10.0
        BEBT = 0.2
        DTBT = 10.0
        DEBUG = True
        """),
    "tc2": dict(shape=(10, 8, 8), pairs="""
        USE_REGRIDDING = True
        THICKNESSDIFFUSE = True
        THICKNESSDIFFUSE_FIRST = True
        MIXEDLAYER_RESTRAT = True
        MEKE_KHTH_FAC = 0.5
        USE_STORED_SLOPES = True
        KHTH = 1.0
        KHTH_MAX = 900.0
        FOX_KEMPER_ML_RESTRAT_COEF = 5.0
        USE_GM_WORK_BUG = False
        DT = 3600.0
        DT_THERM = 7200.0
        DTBT_RESET_PERIOD = -.98
        REGRIDDING_COORDINATE_MODE = "Z*"
        REMAPPING_SCHEME = "PPM_IH4"
        USE_MEKE = True
        MEKE_VISCOSITY_COEFF_KU = 1.0
        USE_VARIABLE_MIXING = True
        RESOLN_SCALED_KH = False
        ETA_TOLERANCE = 1.0E-06
        VELOCITY_TOLERANCE = 0.001
        BOUND_CORIOLIS = True
        LAPLACIAN = True
        KH_VEL_SCALE = 0.05
        SMAGORINSKY_KH = True
        SMAG_LAP_CONST = 0.06
        AH_VEL_SCALE = 0.05
        SMAGORINSKY_AH = True
        SMAG_BI_CONST = 0.06
        DYNAMIC_VISCOUS_ML = True
        KV = 1.0E-04
        HMIX_FIXED = 0.5
        CHANNEL_DRAG = True
        HBBL = 10.0
        MAXVEL = 10.0
        USE_JACKSON_PARAM = True
        ML_OMEGA_FRAC = 1.0
        DRAG_BG_VEL = 0.1
        BBL_THICK_MIN = 0.1
        BOUND_BT_CORRECTION = True
        NONLINEAR_BT_CONTINUITY = True
        BT_PROJECT_VELOCITY = True
        BT_THICK_SCHEME = "FROM_BT_CONT"
        BEBT = 0.2
        DTBT = -0.95
        BULK_RI_ML = 0.05
        TKE_DECAY = 10.0
        DEBUG = True
        USE_PSURF_IN_EOS = False
        REMAP_UV_USING_OLD_ALG = True
        USE_LAND_MASK_FOR_HVISC = False
        """),
    "tc1": dict(shape=(10, 8, 8), pairs="""
        THICKNESSDIFFUSE = True
        THICKNESSDIFFUSE_FIRST = True
        MIXEDLAYER_RESTRAT = True
        KHTH_SLOPE_CFF = 0.1
        KHTH = 1.0
        KHTH_MAX = 900.0
        FOX_KEMPER_ML_RESTRAT_COEF = 5.0
        USE_GM_WORK_BUG = True
        DT = 900.0
        DT_THERM = 3600.0
        DTBT_RESET_PERIOD = 0.0
        USE_VARIABLE_MIXING = True
        USE_VISBECK = True
        RESOLN_SCALED_KH = True
        RESOLN_SCALED_KHTH = True
        RESOLN_SCALED_KHTR = True
        ETA_TOLERANCE = 1.0E-06
        VELOCITY_TOLERANCE = 0.001
        BOUND_CORIOLIS = True
        AH_VEL_SCALE = 0.05
        SMAGORINSKY_AH = True
        SMAG_BI_CONST = 0.06
        PRANDTL_TURB = 0.0
        DYNAMIC_VISCOUS_ML = True
        KV = 1.0E-04
        HBBL = 10.0
        MAXVEL = 10.0
        USE_JACKSON_PARAM = True
        ML_OMEGA_FRAC = 1.0
        DRAG_BG_VEL = 0.1
        BBL_THICK_MIN = 0.1
        BOUND_BT_CORRECTION = True
        NONLINEAR_BT_CONTINUITY = True
        BT_PROJECT_VELOCITY = True
        BT_THICK_SCHEME = "FROM_BT_CONT"
        BEBT = 0.2
        DTBT = -0.95
        BULK_RI_ML = 0.05
        TKE_DECAY = 10.0
        DEBUG = True
        USE_PSURF_IN_EOS = False
        USE_LAND_MASK_FOR_HVISC = False
                DIFFUSE_ML_TO_INTERIOR = True
        ML_KHTR_SCALE = 0.0
        """),
}


def pairs_of(name):
    out = {}
    for line in TC_INPUT[name]["pairs"].strip().splitlines():
        if "=" not in line:
            continue
        k, v = (s.strip() for s in line.split("=", 1))
        out[k] = v.strip('"')
    return out


def _b(p, k, default=False):
    return (p[k].strip().lower().startswith("t")) if k in p else default


def _f(p, k, default):
    return float(p[k]) if k in p else default


def build_driver(tmp):
    flags = ["-cpp", "-fdefault-real-8", "-O0", "-ffp-contract=off", f"-I{STUBS}", f"-I{tmp}", "-J", str(tmp)]
    objs = []
    srcs = [os.path.join(STUBS, "mom6_stubs.F90")] + [os.path.join(FDIR, s) for s in SHIMS if "tracer" not in s] + \
           [os.path.join(FDIR, "MOM_dynamics_split_RK2_hip.F90"), os.path.join(ROOT, "tests", "fortran", "dyn_driver.F90")]
    for src in srcs:
        o = str(tmp / (os.path.basename(src)[:-4] + ".o"))
        subprocess.run([FC, *flags, "-c", src, "-o", o], check=True)
        objs.append(o)
    libdir = os.path.join(ROOT, "mom6_amd")
    exe = str(tmp / "dyn_driver")
    subprocess.run([FC, *objs, f"-L{libdir}", "-lmom6hip", f"-Wl,-rpath,{libdir}", "-o", exe], check=True)
    return exe


def case_state(name, seed=21):
    """a synthetic state on the configuration's grid size (its files -- topography, initial conditions -- are not part of the hot path)"""
    ni, nj, nk = TC_INPUT[name]["shape"]
    p = pairs_of(name)
    g = synth.make_grid(ni, nj, nk, land_frac=0.2, seed=seed + 300, reentrant_x=_b(p, "REENTRANT_X", True), reentrant_y=False)
    d = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=seed, umax=0.1, eta_amp=0.2).items()}
    yy = np.linspace(0.0, np.pi, g.shape2(_abi.POS_U)[0])
    taux = np.ascontiguousarray(0.1 * np.cos(2 * yy)[:, None] * g.mask2dCu)
    tauy = np.ascontiguousarray(0.0 * g.mask2dCv)
    rng = np.random.default_rng(17)
    ustar = np.ascontiguousarray(0.004 + 0.008 * rng.random(g.shape2(_abi.POS_H)))
    rng = np.random.default_rng(9)
    su, sv = g.shape2(_abi.POS_U), g.shape2(_abi.POS_V)
    bbl = dict(Kv_bbl_u=1.0e-3 * (0.5 + rng.random(su)), Kv_bbl_v=1.0e-3 * (0.5 + rng.random(sv)),
               bbl_thick_u=2.0 + 8.0 * rng.random(su), bbl_thick_v=2.0 + 8.0 * rng.random(sv))
    # the vertical grid of a layered run: target densities and reduced gravities (GV%Rlay, GV%g_prime)
    Rlay = np.linspace(1024.0, 1028.0, nk)
    g_prime = np.zeros(nk + 1); g_prime[0] = g.g_Earth
    g_prime[1:nk] = g.g_Earth * np.diff(Rlay) / g.Rho0
    return g, d, taux, tauy, ustar, bbl, Rlay, g_prime


def config_state(name):
    """The configuration's own analytic initial state on its grid (tests/config_ics.py): tc1 -- `benchmark` topography, thicknesses and
    temperatures on the `ts_range` coordinate (.testing/tc1/MOM_input: TOPO_CONFIG, THICKNESS_CONFIG, TS_CONFIG = "benchmark", COORD_CONFIG =
    "ts_range", SOUTHLAT = -41, LENLAT = 41, LENLON = 90, MAXIMUM_DEPTH = 5500, MINIMUM_DEPTH = 1); tc3 -- a flat 600 m box of 100 km with the
    disc of `circle_obcs` (DISK_RADIUS = 24) in ten layers of `layer_ref` densities; the ocean at rest (VELOCITY_CONFIG = "zero").  The
    horizontal metrics stay the synthetic grid's (they are inputs of both sides); wind and boundary-layer inputs as in case_state."""
    import config_ics as ci
    from oracle import orc
    ni, nj, nk = TC_INPUT[name]["shape"]
    p = pairs_of(name)
    g = synth.make_grid(ni, nj, nk, land_frac=0.0, seed=321, reentrant_x=_b(p, "REENTRANT_X", True), reentrant_y=False)
    if name == "tc1":
        west, len_lon, south, len_lat, max_depth, min_depth = 0.0, 90.0, -41.0, 41.0, 5500.0, 1.0
        lon, lat = ci.cell_coordinates(g, west, len_lon, south, len_lat)
        D = ci.benchmark_topography(lon, lat, west, len_lon, south, len_lat, max_depth, min_depth)
        ci.set_bathymetry(g, D, min_depth)
        E = orc.eos("WRIGHT"); P_Ref = 2.0e7; nkmb = 4      # NKML + NKBL (bulk mixed layer: MOM.F90:2439-2444)
        Rlay, g_prime = ci.coord_from_TS_range(nk, E, P_Ref, 25.0, 3.0, 5.0, g.g_Earth, g.Rho0, nk_rho_varies=nkmb)      # TS_RANGE_*
        h_c = ci.benchmark_thickness(D, lat, south, len_lat, max_depth, Rlay, E, P_Ref, g.Angstrom_H * g.H_to_Z, nk_rho_varies=nkmb) * g.Z_to_H
        h_c = np.where(D[None] > min_depth, h_c, g.Angstrom_H)
        T_c, S_c = ci.benchmark_temperature_salinity(lat, south, len_lat, Rlay, E, P_Ref, nk_rho_varies=nkmb)
    elif name == "tc3":
        west, len_lon, south, len_lat, max_depth, min_depth = 0.0, 100.0, 0.0, 100.0, 600.0, 1.0
        lon, lat = ci.cell_coordinates(g, west, len_lon, south, len_lat)
        D = np.full((nj, ni), max_depth)
        ci.set_bathymetry(g, D, min_depth)
        Rlay = 1030.0 + 2.0 * np.arange(nk) / float(nk - 1)      # COORD_CONFIG = "layer_ref": LIGHTEST_DENSITY = 1030, DENSITY_RANGE = 2 (default)
        g_prime = np.zeros(nk + 1); g_prime[0] = g.g_Earth; g_prime[1:nk] = (g.g_Earth / g.Rho0) * np.diff(Rlay)
        h_c = ci.circle_obcs_thickness(D, lon, lat, west, len_lon, south, len_lat, max_depth, nk, g.Angstrom_H * g.H_to_Z, 24.0) * g.Z_to_H
        T_c = np.full((nk, nj, ni), 10.0); S_c = np.full((nk, nj, ni), 35.0)
    else:
        raise ValueError(name)
    d = dict(u=g.zeros3(_abi.POS_U), v=g.zeros3(_abi.POS_V), h=ci.embed3(g, h_c, fill=g.Angstrom_H), T=ci.embed3(g, T_c, fill=10.0),
             S=ci.embed3(g, S_c, fill=35.0))
    yy = np.linspace(0.0, np.pi, g.shape2(_abi.POS_U)[0])
    taux = np.ascontiguousarray(0.1 * np.cos(2 * yy)[:, None] * g.mask2dCu)
    tauy = np.ascontiguousarray(0.0 * g.mask2dCv)
    rng = np.random.default_rng(17)
    ustar = np.ascontiguousarray(0.004 + 0.008 * rng.random(g.shape2(_abi.POS_H)))
    rng = np.random.default_rng(9)
    su, sv = g.shape2(_abi.POS_U), g.shape2(_abi.POS_V)
    bbl = dict(Kv_bbl_u=1.0e-3 * (0.5 + rng.random(su)), Kv_bbl_v=1.0e-3 * (0.5 + rng.random(sv)),
               bbl_thick_u=2.0 + 8.0 * rng.random(su), bbl_thick_v=2.0 + 8.0 * rng.random(sv))
    return g, d, taux, tauy, ustar, bbl, Rlay, g_prime


def meke_of(name, g):
    """USE_MEKE with a nonzero MEKE_VISCOSITY_COEFF_KU: the MEKE module (not part of the hot path) hands horizontal_viscosity MEKE%Ku and
    takes MEKE%mom_src back (MOM_MEKE.F90:1365, MOM_hor_visc.F90:469, :1833).  A synthetic Ku with valid halos."""
    from oracle import orc
    p = pairs_of(name)
    if not (_b(p, "USE_MEKE") and _f(p, "MEKE_VISCOSITY_COEFF_KU", 0.0) != 0.0):
        return None
    rng = np.random.default_rng(31)
    Ku = np.ascontiguousarray(150.0 * rng.random(g.shape2(_abi.POS_H)) * g.mask2dT)
    orc.halo_update(g, Ku, _abi.POS_H)
    return Ku


def lateral_fields(name, g):
    """USE_MEKE / USE_VARIABLE_MIXING: what MOM_MEKE and MOM_lateral_mixing_coeffs (beside the hot path) hand to thickness_diffuse,
    mixedlayer_restrat and tracer_hordiff through the reference's arguments -- MEKE%Kh, VarMix%L2u/v, SN_u/v, Res_fn_u/v/h, Rd_dx_h and the
    stored slopes.  Synthetic, with valid halos.  None: the set uses neither."""
    from oracle import orc
    p = pairs_of(name)
    if not (_b(p, "USE_MEKE") or _b(p, "USE_VARIABLE_MIXING")):
        return None
    rng = np.random.default_rng(77)
    H, U, V = _abi.POS_H, _abi.POS_U, _abi.POS_V
    f = dict(MEKE_Kh=300.0 * rng.random(g.shape2(H)) * g.mask2dT,
             L2u=4.0e8 * (0.2 + rng.random(g.shape2(U))), L2v=4.0e8 * (0.2 + rng.random(g.shape2(V))),
             SN_u=1.0e-6 * rng.random(g.shape2(U)) * g.mask2dCu, SN_v=1.0e-6 * rng.random(g.shape2(V)) * g.mask2dCv,
             Res_fn_u=rng.random(g.shape2(U)), Res_fn_v=rng.random(g.shape2(V)), Res_fn_h=rng.random(g.shape2(H)),
             Rd_dx_h=0.2 + 2.0 * rng.random(g.shape2(H)),
             slope_x=2.0e-3 * (rng.random((g.nk + 1,) + g.shape2(U)) - 0.5) * g.mask2dCu[None],
             slope_y=2.0e-3 * (rng.random((g.nk + 1,) + g.shape2(V)) - 0.5) * g.mask2dCv[None])
    f = {k: np.ascontiguousarray(v) for k, v in f.items()}
    for k, pos in (("MEKE_Kh", H), ("L2u", U), ("L2v", V), ("SN_u", U), ("SN_v", V), ("Res_fn_u", U), ("Res_fn_v", V), ("Res_fn_h", H), ("Rd_dx_h", H)):
        orc.halo_update(g, f[k], pos)
    return f


LATERAL_ORDER = ("MEKE_Kh", "L2u", "L2v", "SN_u", "SN_v", "Res_fn_u", "Res_fn_v", "Res_fn_h", "Rd_dx_h", "slope_x", "slope_y")


def oracle_for(name, g, d, ustar, bbl, Rlay, g_prime, OBC=None):
    """The oracle's control structures as the reference's *_init routines would set them from the pairs (defaults cited)."""
    from oracle import orc
    p = pairs_of(name)
    nk = g.nk
    dt = float(p["DT"])
    use_ALE = _b(p, "USE_REGRIDDING")                                   # MOM.F90:2271
    thermo = _b(p, "ENABLE_THERMODYNAMICS", True)                      # MOM.F90:2253 (False: no temperature, no equation of state)
    bulkml = (not use_ALE) and thermo                                  # BULKMIXEDLAYER default (MOM.F90:2296-2303, with temperature)
    nkml, nkbl = (2, 2) if bulkml else (0, 0)                          # NKML, NKBL defaults (MOM.F90:2439-2444)
    eosn = p.get("EQN_OF_STATE", "WRIGHT")                             # MOM_EOS.F90 EQN_OF_STATE default
    E = orc.eos("LINEAR", 1000.0, -0.2, _f(p, "DRHO_DS", 0.8)) if eosn == "LINEAR" else orc.eos(eosn)      # RHO_T0_S0, DRHO_DT, DRHO_DS
    if not thermo:
        E = None
    pf = dict(reconstruct=_b(p, "RECONSTRUCT_FOR_PRESSURE", use_ALE), use_ALE=use_ALE)      # MOM_PressureForce_FV.F90:1078
    if bulkml or not thermo:
        pf.update(nkmb=nkml + nkbl, Rlay=Rlay, g_prime=g_prime)
    cont = dict(tol_eta=_f(p, "ETA_TOLERANCE", 0.5 * nk * g.Angstrom_H))                    # MOM_continuity_PPM.F90:2717
    if "VELOCITY_TOLERANCE" in p:
        cont["tol_vel"] = float(p["VELOCITY_TOLERANCE"])
    en_dis = _b(p, "CORIOLIS_EN_DIS")
    cor = dict(coriolis_en_dis=int(en_dis), bound_coriolis=int(_b(p, "BOUND_CORIOLIS") and not en_dis))      # MOM_CoriolisAdv.F90:1150-1158
    bt = dict(bebt=_f(p, "BEBT", 0.1), bound_BT_corr=int(_b(p, "BOUND_BT_CORRECTION")), maxCFL_BT_cont=0.25,
              Nonlinear_continuity=int(_b(p, "NONLINEAR_BT_CONTINUITY")), BT_project_velocity=int(_b(p, "BT_PROJECT_VELOCITY")))
    dtbt_in = _f(p, "DTBT", -0.98)                                     # MOM_barotropic.F90:4700
    bt["dtbt_fraction"] = -dtbt_in if dtbt_in < 0 else 0.98
    dyn_ml = _b(p, "DYNAMIC_VISCOUS_ML")
    vv = dict(Kv=float(p["KV"]), Hbbl=float(p["HBBL"]), Hmix=_f(p, "HMIX_FIXED", 0.0) if nkml < 1 else 0.0,
              Kvml_invZ2=_f(p, "KV_ML_INVZ2", 0.0) if nkml < 1 else 0.0, direct_stress=_b(p, "DIRECT_STRESS"),
              maxvel=_f(p, "MAXVEL", 3.0e8), CFL_based_trunc=True, CFL_trunc=0.5, dynamic_viscous_ML=dyn_ml, nkml=nkml,
              harmonic_visc=_b(p, "HARMONIC_VISC"))
    hv = dict(Kh=_f(p, "KH", 0.0), Laplacian=int(_b(p, "LAPLACIAN")), biharmonic=int(_b(p, "BIHARMONIC", True)), Smagorinsky_Kh=int(_b(p, "SMAGORINSKY_KH")),
              Smag_Lap_const=_f(p, "SMAG_LAP_CONST", 0.0), Kh_vel_scale=_f(p, "KH_VEL_SCALE", 0.0), Smagorinsky_Ah=int(_b(p, "SMAGORINSKY_AH")),
              Smag_bi_const=_f(p, "SMAG_BI_CONST", 0.0), Ah_vel_scale=_f(p, "AH_VEL_SCALE", 0.0),
              use_land_mask=int(_b(p, "USE_LAND_MASK_FOR_HVISC", True)),
              # BOUND_CORIOLIS_BIHARM defaults to BOUND_CORIOLIS, BOUND_CORIOLIS_VEL to MAXVEL (MOM_hor_visc.F90:2247-2264)
              bound_Coriolis=int(_b(p, "BOUND_CORIOLIS") and _b(p, "SMAGORINSKY_AH")), bound_Cor_vel=_f(p, "MAXVEL", 3.0e8))
    arrs = {k: v.copy() for k, v in bbl.items()}
    if dyn_ml or nkml > 0:
        arrs["ustar"] = ustar
    # set_visc_CS (MOM_set_viscosity.F90:2920-3130): one structure for set_viscous_BBL and set_viscous_ML
    chan = _b(p, "CHANNEL_DRAG"); rino = _b(p, "USE_JACKSON_PARAM")
    c_smag = _f(p, "SMAG_CONST_CHANNEL", _f(p, "SMAG_LAP_CONST", 0.15))      # :3093-3105
    bulk_Ri = _f(p, "BULK_RI_ML_VISC", _f(p, "BULK_RI_ML", 0.0)); decay = _f(p, "TKE_DECAY_VISC", _f(p, "TKE_DECAY", 0.0))
    sv = orc.set_visc_cs(g, float(p["HBBL"]), float(p["KV"]), cdrag=_f(p, "CDRAG", 0.003), drag_bg_vel=_f(p, "DRAG_BG_VEL", 0.0),
                         BBL_thick_min=_f(p, "BBL_THICK_MIN", 0.0), linear_drag=_b(p, "LINEAR_DRAG"), BBL_use_EOS=_b(p, "BBL_USE_EOS", thermo),
                         RiNo_mix=rino, Channel_drag=chan, c_Smag=c_smag if c_smag >= 0.0 else 0.15,
                         dynamic_viscous_ML=dyn_ml, nkml=nkml, bulk_Ri_ML=bulk_Ri if dyn_ml else 0.0, TKE_decay=decay if dyn_ml else 0.0,
                         omega_frac=_f(p, "ML_OMEGA_FRAC", 0.0) if dyn_ml else 0.0, Rlay=Rlay if (bulkml or not thermo) else None)
    if chan:
        arrs.update(Ray_u=g.zeros3(_abi.POS_U), Ray_v=g.zeros3(_abi.POS_V))
    if dyn_ml:
        arrs.update(nkml_visc_u=g.zeros2(_abi.POS_U), nkml_visc_v=g.zeros2(_abi.POS_V))
    hvcs = orc.hor_visc_cs(g, dt, **hv)
    Ku = meke_of(name, g)
    mom_src = orc.hor_visc_set_meke(hvcs, Ku=Ku, mom_src=g.zeros2(_abi.POS_H)) if Ku is not None else None
    st = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], dt, be=_f(p, "BE", 0.6), eos_form=E, pressureforce=pf,
                      vertvisc=orc.vertvisc_cs(g, **vv), visc=orc.vertvisc_type(**arrs), hor_visc=hvcs, set_visc=sv if dyn_ml else None,
                      continuity=cont, coriolis=cor, OBC=OBC, **bt)
    # the barotropic time step as barotropic_init leaves it (MOM_barotropic.F90:4899-4914) and MOM.F90's DTBT_RESET_PERIOD (:1227-1234)
    SSH_extra = _f(p, "SSH_EXTRA", min(10.0, 0.05 * float(g.bathyT.max())))
    orc.set_dtbt(g, st.bcs, gtot_est=float(sum(g.H_to_Z * g_prime[k] for k in range(nk))), SSH_add=SSH_extra)
    if dtbt_in > 0:
        st.bcs.dtbt = dtbt_in
    reset = _f(p, "DTBT_RESET_PERIOD", -1.0)
    calc = lambda n: (reset == 0.0)
    st.mom_src = mom_src
    st.E = E
    st.bbl = lambda: orc.set_viscous_BBL(g, sv, st.u, st.v, st.h, st.T if thermo else None, st.S if thermo else None, E, st.visc,
                                         **({} if OBC is None else dict(OBC=OBC)))      # MOM.F90:1205
    return st, calc, dict(use_eos=int(thermo), use_ale=int(use_ALE), nk_rho_varies=nkml + nkbl, nkml=nkml)


def write_case(tmp, name, nsteps, resident, state, bbl_mode=0, extra_pairs=None):
    g, d, taux, tauy, ustar, bbl, Rlay, g_prime = state
    p = dict(pairs_of(name), **(extra_pairs or {}))
    use_ALE = _b(p, "USE_REGRIDDING")
    nkml, nkbl = (0, 0) if (use_ALE or not _b(p, "ENABLE_THERMODYNAMICS", True)) else (2, 2)
    with open(tmp / "in.bin", "wb") as f:
        lat = lateral_fields(name, g)
        np.array([g.ni, g.nj, g.nk, g.halo, int(_b(p, "REENTRANT_X", True)), 0, g.first_direction, int(lat is not None)], dtype="<i4").tofile(f)
        Ku = meke_of(name, g)
        np.array([nsteps, int(resident), int(_b(p, "ENABLE_THERMODYNAMICS", True)), int(use_ALE), nkml + nkbl, nkml, bbl_mode, int(Ku is not None)],
                 dtype="<i4").tofile(f)
        np.array([g.Angstrom_H, g.H_subroundoff, g.dZ_subroundoff, g.H_to_Z, g.Z_to_H, g.g_Earth, g.Rho0, float(p["DT"])], dtype="<f8").tofile(f)
        for n in _abi.ALL_METRICS:
            np.ascontiguousarray(g.metrics[n], dtype="<f8").tofile(f)
        for a in (d["u"], d["v"], d["h"], d["T"], d["S"], taux, tauy, ustar, Rlay, g_prime):
            np.ascontiguousarray(a, dtype="<f8").tofile(f)
        if bbl_mode == 0:
            for n in ("Kv_bbl_u", "Kv_bbl_v", "bbl_thick_u", "bbl_thick_v"):
                np.ascontiguousarray(bbl[n], dtype="<f8").tofile(f)
        if Ku is not None:
            Ku.tofile(f)
        if lat is not None:      # (read by cycle_driver.F90 only)
            for k in LATERAL_ORDER:
                np.ascontiguousarray(lat[k], dtype="<f8").tofile(f)
    with open(tmp / "params.txt", "w") as f:
        for k, v in p.items():
            f.write(f"{k} = {v}\n")


OUT = [("u", _abi.POS_U, 3), ("v", _abi.POS_V, 3), ("h", _abi.POS_H, 3), ("uh", _abi.POS_U, 3), ("vh", _abi.POS_V, 3), ("uhtr", _abi.POS_U, 3),
       ("vhtr", _abi.POS_V, 3), ("eta_av", _abi.POS_H, 2), ("nkml_visc_u", _abi.POS_U, 2), ("nkml_visc_v", _abi.POS_V, 2)]


def read_out(path, g, meke=False, OBC=None):
    raw = np.fromfile(path, dtype="<f8")
    out = OUT + ([("mom_src", _abi.POS_H, 2)] if meke else [])
    shapes = [g.shape3(pos) if nd == 3 else g.shape2(pos) for _, pos, nd in out]
    if OBC is not None:      # OBC%rx_normal, OBC%ry_normal, then segment%normal_vel of every segment on the PE
        out = out + [("rx_normal", _abi.POS_U, 3), ("ry_normal", _abi.POS_V, 3)] + [(f"normal_vel_{n + 1}", None, 3) for n, s in enumerate(OBC.segment) if s.on_pe]
        shapes = shapes + [g.shape3(_abi.POS_U), g.shape3(_abi.POS_V)] + [s.normal_vel.shape for s in OBC.segment if s.on_pe]
    sizes = [int(np.prod(s)) for s in shapes]
    assert raw.size == sum(sizes)
    return {n: a.reshape(s) for (n, _, _), a, s in zip(out, np.split(raw, np.cumsum(sizes)[:-1]), shapes)}


@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")
def test_the_split_RK2_module_shim_compiles_and_fails_loudly_without_gpu(tmp_path):
    """MOM_dynamics_split_RK2_hip.F90 (the reference's module name and dummy-argument lists) and the driver that uses nothing else
    compile against the type stand-ins; without a GPU the first library call is a FATAL error with the library's message"""
    import torch
    exe = build_driver(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tests")
    write_case(tmp_path, "tc4", 1, False, case_state("tc4"))
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "params.txt")], capture_output=True, text=True)
    assert r.returncode != 0
    assert "FATAL" in r.stderr and "no HIP device" in r.stderr


@pytest.mark.parametrize("name", list(TC_INPUT))
def test_oracle_runs_the_transcribed_sets(name):
    """the oracle configured from the pairs: two steps, finite, positive thicknesses"""
    state = case_state(name)
    g, d, taux, tauy, ustar, bbl, Rlay, g_prime = state
    st, calc, _ = oracle_for(name, g, d, ustar, bbl, Rlay, g_prime)
    for n in range(2):
        st.bbl()
        st.step(taux, tauy, calc_dtbt=calc(n))
    if _b(pairs_of(name), "CHANNEL_DRAG"):
        assert st.visc._keep["Ray_u"].max() > 0.0
    assert np.all(np.isfinite(st.u)) and np.all(np.isfinite(st.h)) and st.h.min() > 0 and st.bcs.nstep_last >= 1


def driver_p_surf(g, n, mode):
    """the pressures tests/fortran/dyn_driver.F90 makes for step n (1-based) with DRIVER_P_SURF = mode: integers, the same bits on both sides"""
    sh = g.shape2(_abi.POS_H)
    jj, ii = np.meshgrid(np.arange(1, sh[0] + 1), np.arange(1, sh[1] + 1), indexing="ij")      # the driver's isd = jsd = 1
    p_end = 1.0e5 + 8.0 * np.mod(7 * ii + 13 * jj, 97) + 16.0 * n
    p_begin = p_end - 4.0 * np.mod(3 * ii + 5 * jj, 31)
    return (np.ascontiguousarray(p_begin) if mode == 2 else None), np.ascontiguousarray(p_end)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")
@pytest.mark.parametrize("resident", [False, True])
@pytest.mark.parametrize("mode", [1, 2])
def test_reference_named_driver_with_surface_pressures_matches_oracle_bitwise(tmp_path, mode, resident):
    """step_MOM_dyn_split_RK2 through the module shim with a non-zero forces%p_surf (p_surf_end => forces%p_surf, MOM.F90:772), and with
    p_surf_begin as well (dyn_p_surf: PressureForce with p_surf_end, btstep with eta_PF_start; MOM_dynamics_split_RK2.F90:435-442,
    :497-503): the coupled models' calling form, which the shim refused until round 5"""
    name, nsteps = "tc4", 3
    state = case_state(name)
    g, d, taux, tauy, ustar, bbl, Rlay, g_prime = state
    exe = build_driver(tmp_path)
    write_case(tmp_path, name, nsteps, resident, state, bbl_mode=1, extra_pairs={"DRIVER_P_SURF": str(mode)})
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "params.txt")], capture_output=True, text=True)
    assert r.returncode == 0 and "dyn_driver ok" in r.stdout, r.stderr[-2000:]
    st, calc, _ = oracle_for(name, g, d, ustar, bbl, Rlay, g_prime)
    plain, _, _ = oracle_for(name, g, d, ustar, bbl, Rlay, g_prime)
    for n in range(nsteps):
        st.bbl(); plain.bbl()
        pb, pe = driver_p_surf(g, n + 1, mode)
        # mode 1: p_surf_begin is not associated, so forces%p_surf decides (:441); p_surf_end (=> forces%p_surf) alone changes nothing
        st.step(taux, tauy, calc_dtbt=calc(n), p_surf_begin=pb, p_surf_end=pe, p_surf=pe)
        plain.step(taux, tauy, calc_dtbt=calc(n))
    got = read_out(str(tmp_path / "out.bin"), g, meke=st.mom_src is not None)
    want = dict(u=st.u, v=st.v, h=st.h, uh=st.uh, vh=st.vh, uhtr=st.uhtr, vhtr=st.vhtr, eta_av=st.eta_av)
    for n, pos, nd in OUT:
        if n in want:
            assert bits_equal(interior(g, got[n], pos), interior(g, want[n], pos)), (mode, n, float(np.abs(got[n] - want[n]).max()))
    assert not bits_equal(st.u, plain.u)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")
@pytest.mark.parametrize("resident", [False, True])
@pytest.mark.parametrize("name", list(TC_INPUT))
def test_reference_named_driver_with_the_testing_sets_matches_oracle_bitwise(tmp_path, name, resident):
    """every shim's *_init reads the configuration's pairs without refusing one; three steps of step_MOM_dyn_split_RK2 on host arrays
    are the oracle's bits; with GPU_RESIDENT_DYNAMICS the steps move nothing over PCIe between the first upload and the final
    dyn_split_RK2_sync_to_host"""
    nsteps = 3
    state = case_state(name)
    g, d, taux, tauy, ustar, bbl, Rlay, g_prime = state
    exe = build_driver(tmp_path)
    write_case(tmp_path, name, nsteps, resident, state, bbl_mode=1)
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "params.txt")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "dyn_driver ok" in r.stdout
    st, calc, _ = oracle_for(name, g, d, ustar, bbl, Rlay, g_prime)
    for n in range(nsteps):
        st.bbl()      # set_viscous_BBL before every step (resident: on the device mirrors of u, v, h, T, S, its results straight into the step's)
        st.step(taux, tauy, calc_dtbt=calc(n))
    got = read_out(str(tmp_path / "out.bin"), g, meke=st.mom_src is not None)
    want = dict(u=st.u, v=st.v, h=st.h, uh=st.uh, vh=st.vh, uhtr=st.uhtr, vhtr=st.vhtr, eta_av=st.eta_av)
    if st.mom_src is not None:
        assert bits_equal(interior(g, got["mom_src"], _abi.POS_H), interior(g, st.mom_src, _abi.POS_H)) and st.mom_src.min() < 0.0
    if "nkml_visc_u" in st.visc._keep:
        want.update(nkml_visc_u=st.visc._keep["nkml_visc_u"], nkml_visc_v=st.visc._keep["nkml_visc_v"])
    for n, pos, nd in OUT:
        if n in want:
            assert bits_equal(interior(g, got[n], pos), interior(g, want[n], pos)), (name, n, float(np.abs(got[n] - want[n]).max()))
    # what crossed PCIe after initialisation
    words = r.stdout.split()
    stats = {w.split("=")[0]: int(w.split("=")[1]) for w in words if "=" in w}
    # thickness_diffuse_init and mixedlayer_restrat_init (+ its restart registration) took the same parameter file
    p = pairs_of(name)
    assert stats["thickness_diffuse"] == int(p.get("THICKNESSDIFFUSE", "False") == "True")
    assert stats["mixedlayer_restrat"] == int(p.get("MIXEDLAYER_RESTRAT", "False") == "True")
    n3 = int(np.prod(g.shape3(_abi.POS_H)))
    if resident:
        # one upload of each input (u, v, h are already there from the initialisation), one download of each output and restart field
        # (the six arrays set_viscous_BBL writes -- Kv_bbl_u/v, bbl_thick_u/v, with CHANNEL_DRAG Ray_u/v -- are device-side now: no upload,
        # one download each at the end)
        assert stats["h2d_calls"] <= 16 and stats["d2h_calls"] <= 30, stats
        # (independent of the number of steps: T, S, uhtr, vhtr, with CHANNEL_DRAG visc%Ray_u/v, and the 2-D forcing / visc fields
        # go up once; 7 fields + 8 restart fields and a few 2-D ones come down once)
        n2 = n3 // g.nk
        assert stats["h2d_bytes"] <= 8 * (6 * n3 + 16 * n2) * 1.25 and stats["d2h_bytes"] <= 8 * (16 * n3 + 8 * n2) * 1.25, stats
    else:
        assert stats["h2d_calls"] >= nsteps * 8


# ---- one thermodynamic cycle of step_MOM's hot sequence through the shims, staged and resident -------------------------------------------
CYCLE_PAIRS = """
        THICKNESSDIFFUSE = True
        MIXEDLAYER_RESTRAT = True
        FOX_KEMPER_ML_RESTRAT_COEF = 5.0
        KHTR = 100.0
        TRACER_ADVECTION_SCHEME = "PPM:H3"
        TEST_NCYCLES = 2
        """


CYCLE_ALE_PAIRS = """
        TEST_ALE = True
        REMAPPING_SCHEME = "PPM_H4"
        VELOCITY_REMAPPING_SCHEME = "PLM"
        REMAP_UV_USING_OLD_ALG = False
        REGRID_TIME_SCALE = 3600.0
        REGRID_FILTER_DEEP_DEPTH = 500.0
        REMAP_BOUNDARY_EXTRAP = True
        INIT_BOUNDARY_EXTRAP = False
        """


def build_cycle_driver(tmp):
    # (-DMOM6HIP_WITH_ALE_SHIM: the MOM_ALE shim takes the place of the stand-in module of the same name)
    flags = ["-cpp", "-DMOM6HIP_WITH_ALE_SHIM", "-fdefault-real-8", "-O0", "-ffp-contract=off", f"-I{STUBS}", f"-I{tmp}", "-J", str(tmp)]
    objs = []
    # (MOM_ALE right after the glue: its ALE_CS replaces the type-only stand-in before any module that takes one is compiled)
    shims = SHIMS[:2] + ["MOM_ALE_hip.F90"] + SHIMS[2:]
    srcs = [os.path.join(STUBS, "mom6_stubs.F90")] + [os.path.join(FDIR, s) for s in shims] + \
           [os.path.join(FDIR, "MOM_dynamics_split_RK2_hip.F90"), os.path.join(ROOT, "tests", "fortran", "cycle_driver.F90")]
    for src in srcs:
        o = str(tmp / (os.path.basename(src)[:-4] + ".o"))
        subprocess.run([FC, *flags, "-c", src, "-o", o], check=True)
        objs.append(o)
    libdir = os.path.join(ROOT, "mom6_amd")
    exe = str(tmp / "cycle_driver")
    subprocess.run([FC, *objs, f"-L{libdir}", "-lmom6hip", f"-Wl,-rpath,{libdir}", "-o", exe], check=True)
    return exe


def cycle_oracle(name, state, ncycles, nsteps):
    """the same sequence on the oracle: thickness_diffuse, pass h, set_viscous_BBL, nsteps RK2 steps, mixedlayer_restrat, pass h, advect_tracer and
    tracer_hordiff of T and S, uhtr = vhtr = 0, pass T and S"""
    from oracle import orc
    g, d, taux, tauy, ustar, bbl, Rlay, g_prime = state
    p = pairs_of(name)
    dt, dt_therm = float(p["DT"]), float(p["DT_THERM"])
    st, calc, meta = oracle_for(name, g, d, ustar, bbl, Rlay, g_prime)
    H = _abi.POS_H
    lat = lateral_fields(name, g)
    varmix = _b(p, "USE_VARIABLE_MIXING")
    tdf = {}
    if lat is not None:      # what MOM_thickness_diffuse_hip.F90 hands the library from VarMix and MEKE (thickness_diffuse :133-330)
        if _b(p, "USE_MEKE"):
            tdf["MEKE_Kh"] = lat["MEKE_Kh"]
        if varmix and _b(p, "USE_VISBECK") and _f(p, "KHTH_SLOPE_CFF", 0.0) > 0.0:
            tdf.update({k: lat[k] for k in ("L2u", "L2v", "SN_u", "SN_v")})
        if varmix and _b(p, "RESOLN_SCALED_KHTH"):
            tdf.update(Res_fn_u=lat["Res_fn_u"], Res_fn_v=lat["Res_fn_v"])
        if varmix and _b(p, "USE_STORED_SLOPES"):
            tdf.update(slope_x=lat["slope_x"], slope_y=lat["slope_y"])
    if meta["nkml"] > 0:
        tdf["Rlay"] = Rlay
    tdcs = orc.thickness_diffuse_cs(g, Khth=float(p["KHTH"]), Khth_Max=_f(p, "KHTH_MAX", 0.0), KHTH_Slope_Cff=_f(p, "KHTH_SLOPE_CFF", 0.0),
                                    KhTh_fac=_f(p, "MEKE_KHTH_FAC", 0.0) if _b(p, "USE_MEKE") else 1.0, use_GM_work_bug=_b(p, "USE_GM_WORK_BUG", False),
                                    nkml=meta["nkml"], use_variable_mixing=varmix, **tdf)
    mlecs = orc.mixedlayer_restrat_cs(g, ml_restrat_coef=5.0, nkml=meta["nkml"])
    hd_varmix = None
    if varmix:      # tracer_hordiff :219-281: KHTR_SLOPE_CFF is not set in these sets; RESOLN_SCALED_KHTR scales with Res_fn_h
        hd_varmix = dict(Res_fn_h=lat["Res_fn_h"]) if _b(p, "RESOLN_SCALED_KHTR") else {}
    hd_meke = dict(Kh=lat["MEKE_Kh"], KhTr_fac=_f(p, "MEKE_KHTR_FAC", 0.0)) if (varmix and _b(p, "USE_MEKE")) else None
    for nc in range(ncycles):
        orc.thickness_diffuse(g, tdcs, st.h, st.uhtr, st.vhtr, st.T, st.S, st.E, dt_therm)
        orc.halo_update(g, st.h, H)
        st.bbl()
        for n in range(nsteps):
            st.step(taux, tauy, calc_dtbt=calc(n))
        orc.mixedlayer_restrat(g, mlecs, st.h, st.uhtr, st.vhtr, st.T, st.S, st.E, ustar, dt_therm)
        orc.halo_update(g, st.h, H)
        orc.advect_tracer(g, st.h, st.uhtr, st.vhtr, dt_therm, dt, "PPM:H3", [st.T, st.S])
        orc.tracer_hordiff(g, st.h, dt_therm, [st.T, st.S], 100.0, VarMix=hd_varmix, MEKE=hd_meke,
                           neutral=dict(eos=st.E, idx_T=0, idx_S=1) if _b(p, "USE_NEUTRAL_DIFFUSION") else None,
                           epipycnal=dict(eos=st.E, Rlay=Rlay, nkml=meta["nkml"], nk_rho_varies=meta["nk_rho_varies"], idx_T=0, idx_S=1,
                                          ML_KhTr_scale=_f(p, "ML_KHTR_SCALE", 1.0)) if (_b(p, "DIFFUSE_ML_TO_INTERIOR") and meta["nkml"] > 0) else None)
        st.uhtr[:] = 0.0; st.vhtr[:] = 0.0
        if _b(p, "TEST_ALE"):      # the ALE block of step_MOM_thermo (MOM.F90:1647-1700) with the parameters of CYCLE_ALE_PAIRS
            ts = float(p["REGRID_TIME_SCALE"])
            rcs = orc.regridding_cs(np.full(g.nk, float(g.bathyT.max()) / g.nk), min_thickness=1.0e-3, old_grid_weight=ts / (ts + dt_therm), zs=0.0,
                                    zd=float(p["REGRID_FILTER_DEEP_DEPTH"]))
            h_new, _ = orc.ale_regrid(g, rcs, st.h)
            # (ALE_set_extrap_boundaries switches the tracers' remapping structure to REMAP_BOUNDARY_EXTRAP, MOM_ALE.F90:336; the velocities'
            # keeps INIT_BOUNDARY_EXTRAP, :256-261 -- as the reference's own MOM_ALE shows in tests/test_reference_kernels.py)
            orc.ale_remap_tracers(g, p["REMAPPING_SCHEME"], st.h, h_new, [st.T, st.S], boundary_extrapolation=_b(p, "REMAP_BOUNDARY_EXTRAP"))
            hu0, hv0 = orc.ale_remap_set_h_vel(g, st.h)
            hu1, hv1 = orc.ale_remap_set_h_vel(g, h_new)
            orc.ale_remap_velocities(g, p["VELOCITY_REMAPPING_SCHEME"], hu0, hv0, hu1, hv1, st.u, st.v, boundary_extrapolation=_b(p, "INIT_BOUNDARY_EXTRAP"))
            if _b(p, "REMAP_AUXILIARY_VARS"):      # remap_dyn_split_RK2_aux_vars (MOM_dynamics_split_RK2.F90:1273-1301; STORE_CORIOLIS_ACCEL is the default)
                A = st.arrs
                for a, b in (("u_av", "v_av"), ("CAu_pred", "CAv_pred"), ("diffu", "diffv")):
                    orc.ale_remap_velocities(g, p["VELOCITY_REMAPPING_SCHEME"], hu0, hv0, hu1, hv1, A[a], A[b], boundary_extrapolation=_b(p, "INIT_BOUNDARY_EXTRAP"))
                    if a != "diffu":      # the two pass_vector calls of the routine
                        orc.halo_update(g, A[a], _abi.POS_U); orc.halo_update(g, A[b], _abi.POS_V)
            sj, si = g.csl(H)
            sj, si = slice(sj.start - 1, sj.stop + 1), slice(si.start - 1, si.stop + 1)
            st.h[:, sj, si] = h_new[:, sj, si]
            orc.halo_update(g, st.u, _abi.POS_U); orc.halo_update(g, st.v, _abi.POS_V); orc.halo_update(g, st.h, H)
        orc.halo_update(g, st.T, H); orc.halo_update(g, st.S, H)
    return st


@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")
def test_the_cycle_driver_compiles(tmp_path):
    """every shim in one program (the split RK2 module, the two lateral parameterisations, the two tracer modules) compiles and links"""
    build_cycle_driver(tmp_path)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")
@pytest.mark.parametrize("resident", [False, True])
@pytest.mark.parametrize("with_ALE", [False, True, "neutral", "remap_aux"], ids=["no_ALE", "ALE", "neutral_diffusion", "ALE_remap_aux"])
def test_one_thermodynamic_cycle_of_step_MOM_from_fortran_matches_oracle(tmp_path, resident, with_ALE):
    """thickness_diffuse -> set_viscous_BBL -> step_MOM_dyn_split_RK2 x (DT_THERM / DT) -> mixedlayer_restrat -> advect_tracer -> tracer_hordiff, twice, from a
    Fortran program that calls reference-named procedures only, with the .testing/tc4 parameter set: u, v, h, T, S and the transports equal
    the oracle's bit for bit; with GPU_RESIDENT_DYNAMICS the fields cross PCIe once in each direction, whatever the number of cycles"""
    name = "tc4"
    neutral = with_ALE == "neutral"      # tracer_hordiff with .testing/tc2's USE_NEUTRAL_DIFFUSION = True (six layers, no ALE block)
    remap_aux = with_ALE == "remap_aux"      # REMAP_AUXILIARY_VARS = True: remap_dyn_split_RK2_aux_vars in the ALE block (MOM.F90:1678-1683)
    with_ALE = with_ALE is True or remap_aux
    # with_ALE: six layers, and after the tracers the ALE block on the host arrays between dyn_split_RK2_sync_to_host and
    # dyn_split_RK2_host_was_modified (z* regrid with a time scale, PPM_H4 / PLM remapping of T, S, u, v)
    TC_INPUT["tc4c"] = dict(shape=(14, 10, 6) if with_ALE or neutral else TC_INPUT[name]["shape"],
                            pairs=TC_INPUT[name]["pairs"] + CYCLE_PAIRS + (CYCLE_ALE_PAIRS if with_ALE else "") +
                            ("\n        USE_NEUTRAL_DIFFUSION = True\n" if neutral else "") +
                            ("\n        REMAP_AUXILIARY_VARS = True\n" if remap_aux else ""))
    exe = build_cycle_driver(tmp_path)
    state = case_state("tc4c")
    g = state[0]
    p = pairs_of("tc4c")
    nsteps = int(round(float(p["DT_THERM"]) / float(p["DT"])))
    want = cycle_oracle("tc4c", state, 2, nsteps)
    write_case(tmp_path, "tc4c", nsteps, resident, state, bbl_mode=1)
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "params.txt")], capture_output=True, text=True)
    assert r.returncode == 0 and "cycle_driver ok" in r.stdout, r.stderr[-800:]
    raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8")
    n3 = int(np.prod(g.shape3(_abi.POS_H)))
    T_, S_ = raw[-2 * n3:-n3].reshape(g.shape3(_abi.POS_H)), raw[-n3:].reshape(g.shape3(_abi.POS_H))
    shapes = [g.shape3(pos) if nd == 3 else g.shape2(pos) for _, pos, nd in OUT]
    got = {n: a.reshape(s) for (n, _, _), a, s in zip(OUT, np.split(raw[:-2 * n3], np.cumsum([int(np.prod(s)) for s in shapes])[:-1]), shapes)}
    for n, pos in (("u", _abi.POS_U), ("v", _abi.POS_V), ("h", _abi.POS_H), ("uh", _abi.POS_U), ("vh", _abi.POS_V)):
        assert bits_equal(interior(g, got[n], pos), interior(g, getattr(want, n), pos)), (n, float(np.abs(got[n] - getattr(want, n)).max()))
    assert bits_equal(interior(g, T_), interior(g, want.T)) and bits_equal(interior(g, S_), interior(g, want.S))
    assert np.all(got["uhtr"] == 0.0) and np.all(got["vhtr"] == 0.0)
    stats = {w.split("=")[0]: int(w.split("=")[1]) for w in r.stdout.split() if "=" in w}
    if resident and not with_ALE:
        n2 = n3 // g.nk
        assert stats["h2d_bytes"] <= 8 * (10 * n3 + 30 * n2) * 1.25 and stats["d2h_bytes"] <= 8 * (20 * n3 + 20 * n2) * 1.25, stats


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")
@pytest.mark.parametrize("resident", [False, True])
@pytest.mark.parametrize("name", ["tc2", "tc1"])
def test_thermodynamic_cycle_with_the_tc2_and_tc1_sets(tmp_path, name, resident):
    """The same composed cycle with the other two parameter sets: tc2 (MEKE%Kh with MEKE_KHTH_FAC, KHTH_MAX, the stored slopes of
    USE_STORED_SLOPES in thickness_diffuse; the OM4 form of mixedlayer_restrat; DYNAMIC_VISCOUS_ML, CHANNEL_DRAG, MEKE viscosity in the steps)
    and tc1 (no ALE: the bulk mixed layer -- mixedlayer_restrat_BML, the layered pressure force; the Visbeck term KHTH_SLOPE_CFF with VarMix%L2u /
    SN_u and RESOLN_SCALED_KHTH / _KHTR with the resolution functions), the fields of MOM_MEKE / MOM_lateral_mixing_coeffs synthetic
    (lateral_fields), and tc1's DIFFUSE_ML_TO_INTERIOR with ML_KHTR_SCALE = 0: tracer_epipycnal_ML_diff between the four variable-density layers
    and the interior."""
    cname = name + "c"
    TC_INPUT[cname] = dict(shape=TC_INPUT[name]["shape"], pairs=TC_INPUT[name]["pairs"] + CYCLE_PAIRS)
    exe = build_cycle_driver(tmp_path)
    state = case_state(cname)
    g = state[0]
    p = pairs_of(cname)
    nsteps = int(round(float(p["DT_THERM"]) / float(p["DT"])))
    want = cycle_oracle(cname, state, 2, nsteps)
    write_case(tmp_path, cname, nsteps, resident, state, bbl_mode=1)
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "params.txt")], capture_output=True, text=True)
    assert r.returncode == 0 and "cycle_driver ok" in r.stdout, r.stderr[-1500:]
    raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8")
    n3 = int(np.prod(g.shape3(_abi.POS_H)))
    n2 = n3 // g.nk
    if want.mom_src is not None:      # (written last: MEKE%mom_src of the last step)
        mom_src, raw = raw[-n2:].reshape(g.shape2(_abi.POS_H)), raw[:-n2]
        assert bits_equal(interior(g, mom_src, _abi.POS_H), interior(g, want.mom_src, _abi.POS_H))
    T_, S_ = raw[-2 * n3:-n3].reshape(g.shape3(_abi.POS_H)), raw[-n3:].reshape(g.shape3(_abi.POS_H))
    shapes = [g.shape3(pos) if nd == 3 else g.shape2(pos) for _, pos, nd in OUT]
    got = {n: a.reshape(s) for (n, _, _), a, s in zip(OUT, np.split(raw[:-2 * n3], np.cumsum([int(np.prod(s)) for s in shapes])[:-1]), shapes)}
    for n, pos in (("u", _abi.POS_U), ("v", _abi.POS_V), ("h", _abi.POS_H), ("uh", _abi.POS_U), ("vh", _abi.POS_V)):
        assert bits_equal(interior(g, got[n], pos), interior(g, getattr(want, n), pos)), (name, n, float(np.abs(got[n] - getattr(want, n)).max()))
    assert bits_equal(interior(g, T_), interior(g, want.T)) and bits_equal(interior(g, S_), interior(g, want.S)), name
    assert "thickness_diffuse=1" in r.stdout and "mixedlayer_restrat=1" in r.stdout


# ---- .testing/tc3: the open boundaries (four FLATHER,ORLANSKI segments with zero external data) under the RK2 module shim -----------------
TC3_PAIRS = """
        REENTRANT_X = False
        ENABLE_THERMODYNAMICS = False
        DT = 120.0
        DTBT_RESET_PERIOD = -1.0
        CDRAG = 0.002
        BOUND_CORIOLIS = True
        LAPLACIAN = True
        KH = 25.0
        KH_VEL_SCALE = 0.003
        SMAGORINSKY_KH = True
        SMAG_LAP_CONST = 0.15
        AH_VEL_SCALE = 0.003
        SMAGORINSKY_AH = True
        SMAG_BI_CONST = 0.06
        DIRECT_STRESS = True
        HARMONIC_VISC = True
        HMIX_FIXED = 20.0
        KV = 1.0E-04
        KV_ML_INVZ2 = 0.01
        HBBL = 10.0
        MAXVEL = 10.0
        USE_JACKSON_PARAM = True
        DRAG_BG_VEL = 0.05
        BBL_THICK_MIN = 0.1
        BOUND_BT_CORRECTION = True
        NONLINEAR_BT_CONTINUITY = True
        BT_PROJECT_VELOCITY = True
        BT_THICK_SCHEME = "FROM_BT_CONT"
        BT_STRONG_DRAG = False
        BEBT = 0.2
        DTBT = -0.95
        DEBUG = True
        USE_LAND_MASK_FOR_HVISC = False
        """
TC3_SEGMENTS = ["J=N,I=N:0,FLATHER,ORLANSKI", "J=0,I=0:N,FLATHER,ORLANSKI", "I=N,J=0:N,FLATHER,ORLANSKI", "I=0,J=N:0,FLATHER,ORLANSKI"]
TC3_OBC = dict(freeslip_vorticity=True, freeslip_strain=True, zero_biharmonic=True, gamma_uv=0.3, rx_max=10.0)      # OBC_RADIATION_MAX = 10.0


def write_obc_file(path, g, OBC):
    """ocean_OBC_type as open_boundary_config / open_boundary_init and the first update_OBC_segment_data leave it (dyn_driver.F90)"""
    with open(path, "wb") as f:
        np.array([OBC.number_of_segments, OBC.OBC_pe, OBC.open_u_BCs_exist_globally, OBC.open_v_BCs_exist_globally, OBC.specified_u_BCs_exist_globally,
                  OBC.specified_v_BCs_exist_globally, OBC.Flather_u_BCs_exist_globally, OBC.Flather_v_BCs_exist_globally], dtype="<i4").tofile(f)
        np.array([OBC.zero_vorticity, OBC.freeslip_vorticity, OBC.computed_vorticity, OBC.specified_vorticity, OBC.zero_strain, OBC.freeslip_strain,
                  OBC.computed_strain, OBC.zero_biharmonic], dtype="<i4").tofile(f)
        np.array([OBC.gamma_uv, OBC.rx_max], dtype="<f8").tofile(f)
        for s in OBC.segment:
            np.array([s.direction, s.open, s.specified, s.on_pe, s.is_E_or_W, s.is_N_or_S] +
                     [s.HI.get(k, 0) for k in ("IsdB", "IedB", "JsdB", "JedB", "isd", "ied", "jsd", "jed")] +
                     [s.Flather, s.radiation, s.gradient, s.nudged, 0, 0], dtype="<i4").tofile(f)
        OBC.segnum_u.astype("<i4").tofile(f); OBC.segnum_v.astype("<i4").tofile(f)
        for s in OBC.segment:
            if s.on_pe:
                for a in (s.normal_vel, s.normal_trans, s.normal_vel_bt, s.SSH):
                    np.ascontiguousarray(a, dtype="<f8").tofile(f)


def tc3_case(ic="synthetic"):
    from mom6_amd.open_boundary import ocean_OBC_type
    from test_continuity_obc import open_faces
    TC_INPUT["tc3"] = dict(shape=(10, 8, 10), pairs=TC3_PAIRS)
    g, d, taux, tauy, ustar, bbl, Rlay, g_prime = case_state("tc3") if ic == "synthetic" else config_state("tc3")
    OBC = ocean_OBC_type(g, TC3_SEGMENTS, **TC3_OBC)
    open_faces(g, OBC)
    OBC.rx_normal, OBC.ry_normal = g.zeros3(_abi.POS_U), g.zeros3(_abi.POS_V)
    for s in OBC.segment:      # the value-type data of the segments (U = V = SSH = 0) as the first update_OBC_segment_data leaves them
        s.normal_vel_bt = np.zeros(s.normal_vel.shape[1:]); s.SSH = np.zeros(s.normal_vel.shape[1:])
    taux, tauy = 0.0 * taux, 0.0 * tauy      # WIND_CONFIG = "zero"
    return (g, d, taux, tauy, ustar, bbl, Rlay, g_prime), OBC


def test_oracle_runs_the_tc3_set_with_its_open_boundaries():
    try:
        state, OBC = tc3_case()
        g, d, taux, tauy, ustar, bbl, Rlay, g_prime = state
        st, calc, info = oracle_for("tc3", g, d, ustar, bbl, Rlay, g_prime, OBC=OBC)
        assert info["use_eos"] == 0 and info["nkml"] == 0
        for n in range(3):
            st.bbl(); st.step(taux, tauy, calc_dtbt=calc(n))
        assert np.all(np.isfinite(st.u)) and np.all(np.isfinite(st.h)) and st.h.min() > 0
        assert np.abs(st.u[:, OBC.segnum_u != 0]).max() > 0 and np.abs(OBC.rx_normal).max() > 0
    finally:
        TC_INPUT.pop("tc3", None)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")
@pytest.mark.parametrize("resident", [False, True])
def test_tc3_with_its_open_boundaries_from_fortran_matches_oracle_bitwise(tmp_path, resident):
    """the transcribed parameter set of .testing/tc3 (ENABLE_THERMODYNAMICS = False: the layered pressure force; HARMONIC_VISC; its four
    FLATHER,ORLANSKI segments with OBC_FREESLIP_VORTICITY, OBC_FREESLIP_STRAIN, OBC_ZERO_BIHARMONIC, OBC_RADIATION_MAX = 10) through the
    reference-named driver: set_visc_init(OBC), initialize_dyn_split_RK2(OBC), then set_viscous_BBL and step_MOM_dyn_split_RK2 three
    times; the prognostic fields, OBC%rx_normal / ry_normal and segment%normal_vel are the oracle's bits"""
    import copy
    nsteps = 3
    try:
        state, OBC = tc3_case()
        g, d, taux, tauy, ustar, bbl, Rlay, g_prime = state
        exe = build_driver(tmp_path)
        write_case(tmp_path, "tc3", nsteps, resident, state, bbl_mode=1)
        write_obc_file(str(tmp_path / "obc.bin"), g, OBC)
        r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "params.txt"), str(tmp_path / "obc.bin")],
                           capture_output=True, text=True)
        assert r.returncode == 0 and "dyn_driver ok" in r.stdout, r.stderr[-2000:]
        st, calc, _ = oracle_for("tc3", g, d, ustar, bbl, Rlay, g_prime, OBC=OBC)
        for n in range(nsteps):
            st.bbl(); st.step(taux, tauy, calc_dtbt=calc(n))
        got = read_out(str(tmp_path / "out.bin"), g, OBC=OBC)
        want = dict(u=st.u, v=st.v, h=st.h, uh=st.uh, vh=st.vh, uhtr=st.uhtr, vhtr=st.vhtr, eta_av=st.eta_av)
        for n, pos, nd in OUT:
            if n in want:
                assert bits_equal(interior(g, got[n], pos), interior(g, want[n], pos)), (n, float(np.abs(got[n] - want[n]).max()))
        assert np.abs(st.u[:, OBC.segnum_u != 0]).max() > 0      # (the boundaries are open)
        assert bits_equal(interior(g, got["rx_normal"], _abi.POS_U), interior(g, OBC.rx_normal, _abi.POS_U)) and np.abs(OBC.rx_normal).max() > 0
        assert bits_equal(interior(g, got["ry_normal"], _abi.POS_V), interior(g, OBC.ry_normal, _abi.POS_V))
        for n, s in enumerate(OBC.segment):
            assert bits_equal(got[f"normal_vel_{n + 1}"], s.normal_vel), n
    finally:
        TC_INPUT.pop("tc3", None)


# ---- the configurations' own initial conditions (tests/config_ics.py) ------------------------------------------------------------------------
def test_oracle_runs_tc1_from_its_benchmark_state():
    """`benchmark` thicknesses over the `benchmark` basin: vanished buffer layers, a mixed layer of 50 m, interfaces that outcrop polewards"""
    g, d, taux, tauy, ustar, bbl, Rlay, g_prime = config_state("tc1")
    wet = interior(g, np.asarray(g.mask2dT)) > 0
    col = interior(g, d["h"]).sum(0)
    assert np.allclose(col[wet] * g.H_to_Z, interior(g, np.asarray(g.bathyT))[wet], rtol=1e-9) and 0.05 < 1.0 - wet.mean() < 0.5
    assert np.all(np.diff(Rlay[4:]) > 0) and interior(g, d["h"])[0][wet].max() == pytest.approx(50.0 * g.Z_to_H)
    st, calc, _ = oracle_for("tc1", g, d, ustar, bbl, Rlay, g_prime)
    for n in range(3):
        st.bbl(); st.step(taux, tauy, calc_dtbt=calc(n))
    assert np.all(np.isfinite(st.u)) and st.h.min() > 0 and np.abs(st.u).max() < 1.0


def test_oracle_runs_tc3_from_its_disc_for_its_whole_run():
    """DAYMAX = 0.25 (6 h) at DT = 120 s: 180 steps of the disc of circle_obcs spreading through the four open boundaries"""
    try:
        state, OBC = tc3_case(ic="config")
        g, d, taux, tauy, ustar, bbl, Rlay, g_prime = state
        st, calc, _ = oracle_for("tc3", g, d, ustar, bbl, Rlay, g_prime, OBC=OBC)
        eta0 = interior(g, st.h.sum(0))
        for n in range(180):
            st.bbl(); st.step(taux, tauy, calc_dtbt=calc(n))
        assert np.all(np.isfinite(st.u)) and st.h.min() > 0 and np.abs(st.u).max() < 2.0
        eta1 = interior(g, st.h.sum(0))
        assert np.ptp(eta1) < np.ptp(eta0)      # the depression of the column height has spread
    finally:
        TC_INPUT.pop("tc3", None)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")
@pytest.mark.parametrize("resident", [False, True])
def test_tc1_from_its_benchmark_state_from_fortran_matches_oracle_bitwise(tmp_path, resident):
    nsteps = 3
    state = config_state("tc1")
    g, d, taux, tauy, ustar, bbl, Rlay, g_prime = state
    exe = build_driver(tmp_path)
    write_case(tmp_path, "tc1", nsteps, resident, state, bbl_mode=1)
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "params.txt")], capture_output=True, text=True)
    assert r.returncode == 0 and "dyn_driver ok" in r.stdout, r.stderr[-2000:]
    st, calc, _ = oracle_for("tc1", g, d, ustar, bbl, Rlay, g_prime)
    for n in range(nsteps):
        st.bbl(); st.step(taux, tauy, calc_dtbt=calc(n))
    got = read_out(str(tmp_path / "out.bin"), g, meke=st.mom_src is not None)
    want = dict(u=st.u, v=st.v, h=st.h, uh=st.uh, vh=st.vh, uhtr=st.uhtr, vhtr=st.vhtr, eta_av=st.eta_av)
    for n, pos, nd in OUT:
        if n in want:
            assert bits_equal(interior(g, got[n], pos), interior(g, want[n], pos)), (n, float(np.abs(got[n] - want[n]).max()))


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")
@pytest.mark.parametrize("resident", [False, True])
def test_tc3_from_its_disc_for_its_whole_run_from_fortran_matches_oracle_bitwise(tmp_path, resident):
    """.testing/tc3 as it runs: the disc of circle_obcs, four FLATHER,ORLANSKI segments, DAYMAX = 6 h = 180 steps, through the RK2 module shim"""
    nsteps = 180
    try:
        state, OBC = tc3_case(ic="config")
        g, d, taux, tauy, ustar, bbl, Rlay, g_prime = state
        exe = build_driver(tmp_path)
        write_case(tmp_path, "tc3", nsteps, resident, state, bbl_mode=1)
        write_obc_file(str(tmp_path / "obc.bin"), g, OBC)
        r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "params.txt"), str(tmp_path / "obc.bin")],
                           capture_output=True, text=True)
        assert r.returncode == 0 and "dyn_driver ok" in r.stdout, r.stderr[-2000:]
        st, calc, _ = oracle_for("tc3", g, d, ustar, bbl, Rlay, g_prime, OBC=OBC)
        for n in range(nsteps):
            st.bbl(); st.step(taux, tauy, calc_dtbt=calc(n))
        got = read_out(str(tmp_path / "out.bin"), g, OBC=OBC)
        want = dict(u=st.u, v=st.v, h=st.h, uh=st.uh, vh=st.vh, uhtr=st.uhtr, vhtr=st.vhtr, eta_av=st.eta_av)
        for n, pos, nd in OUT:
            if n in want:
                assert bits_equal(interior(g, got[n], pos), interior(g, want[n], pos)), (n, float(np.abs(got[n] - want[n]).max()))
    finally:
        TC_INPUT.pop("tc3", None)
