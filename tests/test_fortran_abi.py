"""The Fortran side of the boundary: mom6_amd/fortran/mom6hip_c_api.F90 (ISO_C_BINDING interfaces)
compiles with amdflang, and a Fortran host program that owns plain Fortran arrays drives the HIP
advect_tracer through it (HOST memspace: the drop-in path) with bit-identical results to the oracle."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from mom6_amd import _abi
from mom6_amd.grid import Grid
from helpers import bits_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FC = shutil.which("amdflang") or "/opt/rocm/bin/amdflang"
API = os.path.join(ROOT, "mom6_amd", "fortran", "mom6hip_c_api.F90")
DRV = os.path.join(ROOT, "tests", "fortran", "advect_driver.F90")


def _build(tmp):
    flags = ["-O0", "-ffp-contract=off"]
    subprocess.run([FC, *flags, "-c", API, "-o", str(tmp / "api.o"), "-J", str(tmp)], check=True)
    subprocess.run([FC, *flags, f"-I{tmp}", "-c", DRV, "-o", str(tmp / "drv.o"), "-J", str(tmp)], check=True)
    libdir = os.path.join(ROOT, "mom6_amd")
    subprocess.run([FC, str(tmp / "drv.o"), str(tmp / "api.o"), f"-L{libdir}", "-lmom6hip",
                    f"-Wl,-rpath,{libdir}", "-o", str(tmp / "advect_driver")], check=True)
    return str(tmp / "advect_driver")


@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")
def test_fortran_binding_compiles_and_fails_loudly_without_gpu(tmp_path):
    exe = _build(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([exe, str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode != 0
    assert "no HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")
def test_fortran_driver_matches_oracle(tmp_path, oracle):
    exe = _build(tmp_path)
    out = tmp_path / "out.bin"
    r = subprocess.run([exe, str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "advect_driver ok" in r.stdout
    ni, nj, nk, halo = 20, 12, 3, 4
    g = Grid(ni=ni, nj=nj, nk=nk, halo=halo, reentrant_x=True, reentrant_y=True)
    g.H_subroundoff = 1.0e-30
    nih, njh = g.nih, g.njh
    raw = np.fromfile(str(out) + ".in", dtype="<f8")
    sizes = [njh * nih, nk * njh * nih, nk * njh * (nih + 1), nk * (njh + 1) * nih, nk * njh * nih, nk * njh * nih]
    assert raw.size == sum(sizes)
    parts = np.split(raw, np.cumsum(sizes)[:-1])
    areaT = parts[0].reshape(njh, nih)
    h_end = parts[1].reshape(nk, njh, nih); uhtr = parts[2].reshape(nk, njh, nih + 1)
    vhtr = parts[3].reshape(nk, njh + 1, nih)
    t1 = parts[4].reshape(nk, njh, nih).copy(); t2 = parts[5].reshape(nk, njh, nih).copy()
    g.set_metric("areaT", areaT); g.set_metric("IareaT", 1.0 / areaT)
    g.set_metric("mask2dT", np.ones((njh, nih)))
    g.set_metric("mask2dCu", np.ones((njh, nih + 1))); g.set_metric("mask2dCv", np.ones((njh + 1, nih)))
    st = oracle.advect_tracer(g, np.ascontiguousarray(h_end), np.ascontiguousarray(uhtr),
                              np.ascontiguousarray(vhtr), 3600.0, 900.0, "PPM:H3", [t1, t2])
    res = np.fromfile(str(out), dtype="<f8")
    r1, r2 = np.split(res, 2)
    assert bits_equal(t1.ravel(), r1) and bits_equal(t2.ravel(), r2)
    assert f"iterations={st.iterations}" in r.stdout


# ---- the module shims (MOM_continuity_PPM, MOM_CoriolisAdv, MOM_tracer_advect) against the type-only stand-ins --------
FDIR = os.path.join(ROOT, "mom6_amd", "fortran")
STUBS = os.path.join(ROOT, "tests", "fortran", "stubs")
SHIMS = ["mom6hip_c_api.F90", "mom6hip_MOM_glue.F90", "MOM_continuity_PPM_hip.F90", "MOM_CoriolisAdv_hip.F90", "MOM_barotropic_hip.F90",
         "MOM_PressureForce_FV_hip.F90", "MOM_tracer_advect_hip.F90", "MOM_tracer_hor_diff_hip.F90", "MOM_set_viscosity_hip.F90",
         "MOM_vert_friction_hip.F90", "MOM_thickness_diffuse_hip.F90", "MOM_mixed_layer_restrat_hip.F90", "MOM_hor_visc_hip.F90"]


def _build_shims(tmp, driver="shim_driver"):
    """amdflang with MOM6's conventions: preprocessed .F90, default real = 8 bytes"""
    flags = ["-cpp", "-fdefault-real-8", "-O0", "-ffp-contract=off", f"-I{STUBS}", f"-I{tmp}", "-J", str(tmp)]
    objs = []
    for src in [os.path.join(STUBS, "mom6_stubs.F90")] + [os.path.join(FDIR, s) for s in SHIMS] + \
               [os.path.join(ROOT, "tests", "fortran", driver + ".F90")]:
        o = str(tmp / (os.path.basename(src)[:-4] + ".o"))
        subprocess.run([FC, *flags, "-c", src, "-o", o], check=True)
        objs.append(o)
    libdir = os.path.join(ROOT, "mom6_amd")
    exe = str(tmp / driver)
    subprocess.run([FC, *objs, f"-L{libdir}", "-lmom6hip", f"-Wl,-rpath,{libdir}", "-o", exe], check=True)
    return exe


def _shim_case(path, reentrant=(True, True)):
    """the input file of tests/fortran/shim_driver.F90 and the same inputs for the oracle"""
    from mom6_amd import synth
    from oracle import orc
    ni, nj, nk, halo = 30, 14, 4, 4
    g = synth.make_grid(ni, nj, nk, halo=halo, land_frac=0.2, seed=77, reentrant_x=reentrant[0], reentrant_y=reentrant[1])
    d = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=5, umax=0.3, eta_amp=0.2).items()}
    kk = (np.arange(nk) + 0.5) / nk
    vru = np.ascontiguousarray(np.clip(1.0 - 0.8 * kk[:, None, None] ** 2 + 0 * d["u"], 0.0, 1.0) * (g.mask2dCu[None] > 0))
    vrv = np.ascontiguousarray(np.clip(1.0 - 0.8 * kk[:, None, None] ** 2 + 0 * d["v"], 0.0, 1.0) * (g.mask2dCv[None] > 0))
    dt = 900.0
    ccs = orc.continuity_cs(nk, g.Angstrom_H)
    # call 1 (also gives the barotropic transports the second call has to match)
    hp = d["h"].copy(); uh = np.zeros_like(d["u"]); vh = np.zeros_like(d["v"])
    arrs, bt = orc.make_bt_cont(g, with_h=True)
    orc.continuity(g, ccs, d["u"], d["v"], d["h"], hp, uh, vh, dt, visc_rem_u=vru, visc_rem_v=vrv, bt_cont=bt)
    uhbt = np.ascontiguousarray(uh.sum(0) * 1.02); vhbt = np.ascontiguousarray(vh.sum(0) * 0.98)
    with open(path, "wb") as f:
        np.array([ni, nj, nk, halo, int(reentrant[0]), int(reentrant[1]), g.first_direction, 0], dtype="<i4").tofile(f)
        np.array([g.Angstrom_H, g.H_subroundoff, g.dZ_subroundoff, g.H_to_Z, g.Z_to_H, g.g_Earth, g.Rho0, dt], dtype="<f8").tofile(f)
        for n in _abi.ALL_METRICS:
            np.ascontiguousarray(g.metrics[n], dtype="<f8").tofile(f)
        for a in (d["u"], d["v"], d["h"], uhbt, vhbt, vru, vrv, d["T"], d["S"]):
            np.ascontiguousarray(a, dtype="<f8").tofile(f)
    # call 2 and CorAdCalc, as the driver does them
    hp2 = d["h"].copy(); uh2 = np.zeros_like(d["u"]); vh2 = np.zeros_like(d["v"]); ucor = np.zeros_like(d["u"]); vcor = np.zeros_like(d["v"])
    orc.continuity(g, ccs, d["u"], d["v"], d["h"], hp2, uh2, vh2, dt, uhbt=uhbt, vhbt=vhbt, visc_rem_u=vru, visc_rem_v=vrv, u_cor=ucor,
                   v_cor=vcor, bt_cont=bt)
    orc.halo_update(g, uh2, _abi.POS_U); orc.halo_update(g, vh2, _abi.POS_V)
    CAu, CAv = orc.coradcalc(g, d["u"], d["v"], d["h"], uh2, vh2, bound_coriolis=True)
    PFu, PFv, pbce, eta = orc.pressureforce(g, orc.pressureforce_cs(g), orc.eos("WRIGHT"), d["h"], d["T"], d["S"])[:4]
    want = [hp, uh, vh, hp2, uh2, vh2, ucor, vcor, CAu, CAv] + [arrs[n] for n in ("FA_u_W0", "FA_u_WW", "FA_u_E0", "FA_u_EE", "uBT_WW", "uBT_EE",
                                                                                     "FA_v_S0", "FA_v_SS", "FA_v_N0", "FA_v_NN", "vBT_SS", "vBT_NN",
                                                                                     "h_u", "h_v")] + [PFu, PFv, pbce, eta]
    names = ["hp", "uh", "vh", "hp2", "uh2", "vh2", "u_cor", "v_cor", "CAu", "CAv", "FA_u_W0", "FA_u_WW", "FA_u_E0", "FA_u_EE", "uBT_WW", "uBT_EE",
             "FA_v_S0", "FA_v_SS", "FA_v_N0", "FA_v_NN", "vBT_SS", "vBT_NN", "h_u", "h_v", "PFu", "PFv", "pbce", "eta"]
    return g, names, want


@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")
def test_module_shims_compile_against_type_stubs_and_fail_loudly_without_gpu(tmp_path):
    """mom6_amd/fortran/*.F90 with the reference's module names and dummy-argument lists compile (amdflang, real*8) against
    tests/fortran/stubs; without a GPU the first library call stops with MOM_error(FATAL) carrying the library's message"""
    exe = _build_shims(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    _shim_case(str(tmp_path / "in.bin"))
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode != 0
    assert "FATAL" in r.stderr and "no HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")
@pytest.mark.parametrize("reentrant", [(True, True), (True, False)])
def test_module_shims_match_oracle(tmp_path, reentrant):
    """continuity (both call forms of the RK2 step, with BT_cont) and CorAdCalc called through MOM_continuity_PPM /
    MOM_CoriolisAdv with the reference's argument lists on Fortran host arrays: bit-identical with the oracle.
    (True, False): a closed y direction -- the one-tile context wraps x only, as REENTRANT_Y = False says."""
    exe = _build_shims(tmp_path)
    g, names, want = _shim_case(str(tmp_path / "in.bin"), reentrant)
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "shim_driver ok" in r.stdout
    raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8")
    sizes = [w.size for w in want]
    assert raw.size == sum(sizes)
    got = np.split(raw, np.cumsum(sizes)[:-1])
    from helpers import interior
    for n, a, w in zip(names, got, want):
        a = a.reshape(w.shape)
        pos = _abi.POS_U if n in ("uh", "uh2", "u_cor", "CAu", "h_u", "PFu") or n.startswith(("FA_u", "uBT")) else \
            (_abi.POS_V if n in ("vh", "vh2", "v_cor", "CAv", "h_v", "PFv") or n.startswith(("FA_v", "vBT")) else _abi.POS_H)
        ia, iw = interior(g, a, pos), interior(g, w, pos)
        if n == "PFu":      # PressureForce computes the faces Isq..Ieq between computed columns (is-1..ie here needs the halo columns)
            ia, iw = ia[..., 1:-1], iw[..., 1:-1]
        if n == "PFv":
            ia, iw = ia[..., 1:-1, :], iw[..., 1:-1, :]
        assert bits_equal(ia, iw), n


def _bt_case(path, reentrant=(True, True), **kw):
    """the input file of tests/fortran/bt_driver.F90 (the btstep inputs of helpers.barotropic_case) and the oracle's results"""
    from helpers import barotropic_case
    from oracle import orc
    g, cs, case, keep = barotropic_case(orc, **dict(dict(ni=26, nj=14, nk=4, reentrant_x=reentrant[0], reentrant_y=reentrant[1], dt=900.0), **kw))
    h = keep["h"]; bt = keep["bt_arrs"]
    with open(path, "wb") as f:
        np.array([g.ni, g.nj, g.nk, g.halo, int(reentrant[0]), int(reentrant[1]), g.first_direction, 0], dtype="<i4").tofile(f)
        np.array([g.Angstrom_H, g.H_subroundoff, g.dZ_subroundoff, g.H_to_Z, g.Z_to_H, g.g_Earth, g.Rho0, case["dt"], cs.dtbt], dtype="<f8").tofile(f)
        for n in _abi.ALL_METRICS:
            np.ascontiguousarray(g.metrics[n], dtype="<f8").tofile(f)
        for a in (case["U_in"], case["V_in"], h, case["eta_in"], case["bc_accel_u"], case["bc_accel_v"], case["taux"], case["tauy"], case["pbce"],
                  case["eta_PF_in"], case["visc_rem_u"], case["visc_rem_v"], case["uh0"], case["vh0"]):
            np.ascontiguousarray(a, dtype="<f8").tofile(f)
        for n in ("FA_u_W0", "FA_u_WW", "FA_u_E0", "FA_u_EE", "uBT_WW", "uBT_EE", "FA_v_S0", "FA_v_SS", "FA_v_N0", "FA_v_NN", "vBT_SS", "vBT_NN", "h_u", "h_v"):
            np.ascontiguousarray(bt[n], dtype="<f8").tofile(f)
    # ubtav / vbtav of a cold start (MOM_barotropic.F90:5050-5062): btcalc with the default thicknesses, then the k-ordered sums
    cs0, arrs0 = orc.barotropic_cs(g, hvel_scheme="FROM_BT_CONT")
    orc.barotropic_init(g, cs0)
    orc.btcalc(g, cs0, h, may_use_default=True)
    ub = g.zeros2(_abi.POS_U); vb = g.zeros2(_abi.POS_V)
    for k in range(g.nk):
        ub = ub + arrs0["frhatu"][k] * case["U_in"][k]; vb = vb + arrs0["frhatv"][k] * case["V_in"][k]
    out = orc.btstep(g, cs, want_etaav=True, **{k: v for k, v in case.items()})
    want = [out["accel_layer_u"], out["accel_layer_v"], out["eta_out"], out["uhbtav"], out["vhbtav"], out["etaav"], ub, vb]
    names = ["accel_layer_u", "accel_layer_v", "eta_out", "uhbtav", "vhbtav", "etaav", "ubtav", "vbtav"]
    return g, names, want, int(cs.nstep_last)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")
def test_barotropic_shim_matches_oracle(tmp_path):
    """barotropic_init (parameters by name, DTBT > 0), btcalc, bt_mass_source and btstep with the argument list of the RK2
    step's call (MOM_dynamics_split_RK2.F90:655) through the MOM_barotropic shim on Fortran host arrays: bit-identical."""
    exe = _build_shims(tmp_path, "bt_driver")
    g, names, want, nstep = _bt_case(str(tmp_path / "in.bin"))
    assert nstep >= 10
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "bt_driver ok calc_dtbt= T" in r.stdout
    raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8")
    sizes = [w.size for w in want]
    assert raw.size == sum(sizes)
    from helpers import interior
    for n, a, w in zip(names, np.split(raw, np.cumsum(sizes)[:-1]), want):
        a = a.reshape(w.shape)
        pos = _abi.POS_U if n in ("accel_layer_u", "uhbtav", "ubtav") else (_abi.POS_V if n in ("accel_layer_v", "vhbtav", "vbtav") else _abi.POS_H)
        assert bits_equal(interior(g, a, pos), interior(g, w, pos)), n


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")
def test_barotropic_shim_with_open_boundaries_matches_oracle(tmp_path):
    """btcalc and btstep of the MOM_barotropic shim with the reference's ocean_OBC_type (Flather, gradient and specified segments, two of
    them inside the domain; segment%normal_vel_bt, %SSH, %normal_trans on host arrays): the oracle's bits"""
    from helpers import interior
    from oracle import orc
    from test_barotropic_obc import BT_SEGS, btstep_obc_case
    from test_testing_configs import write_obc_file
    exe = _build_shims(tmp_path, "bt_driver")
    g, cs, case, keep, OBC = btstep_obc_case(BT_SEGS, ni=26, nj=14)
    h, bt = keep["h"], keep["bt_arrs"]
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([g.ni, g.nj, g.nk, g.halo, 0, 0, g.first_direction, 0], dtype="<i4").tofile(f)
        np.array([g.Angstrom_H, g.H_subroundoff, g.dZ_subroundoff, g.H_to_Z, g.Z_to_H, g.g_Earth, g.Rho0, case["dt"], cs.dtbt], dtype="<f8").tofile(f)
        for n in _abi.ALL_METRICS:
            np.ascontiguousarray(g.metrics[n], dtype="<f8").tofile(f)
        for a in (case["U_in"], case["V_in"], h, case["eta_in"], case["bc_accel_u"], case["bc_accel_v"], case["taux"], case["tauy"], case["pbce"],
                  case["eta_PF_in"], case["visc_rem_u"], case["visc_rem_v"], case["uh0"], case["vh0"]):
            np.ascontiguousarray(a, dtype="<f8").tofile(f)
        for n in ("FA_u_W0", "FA_u_WW", "FA_u_E0", "FA_u_EE", "uBT_WW", "uBT_EE", "FA_v_S0", "FA_v_SS", "FA_v_N0", "FA_v_NN", "vBT_SS", "vBT_NN", "h_u", "h_v"):
            np.ascontiguousarray(bt[n], dtype="<f8").tofile(f)
    write_obc_file(str(tmp_path / "obc.bin"), g, OBC)
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "obc.bin")], capture_output=True, text=True)
    assert r.returncode == 0 and "bt_driver ok" in r.stdout, r.stderr[-1500:]
    out = orc.btstep(g, cs, want_etaav=True, OBC=OBC, **case)
    closed = orc.btstep(g, cs, want_etaav=True, **case)
    assert not bits_equal(out["eta_out"], closed["eta_out"])
    names = ["accel_layer_u", "accel_layer_v", "eta_out", "uhbtav", "vhbtav", "etaav"]
    raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8")
    sizes = [out[n].size for n in names]
    for n, a in zip(names, np.split(raw[:sum(sizes)], np.cumsum(sizes)[:-1])):
        a = a.reshape(out[n].shape)
        pos = _abi.POS_U if n in ("accel_layer_u", "uhbtav") else (_abi.POS_V if n in ("accel_layer_v", "vhbtav") else _abi.POS_H)
        assert bits_equal(interior(g, a, pos), interior(g, out[n], pos)), (n, np.argwhere(interior(g, a, pos) != interior(g, out[n], pos))[:4].tolist())


# ---- the device-resident step, from Fortran through mom6hip_c_api only ---------------------------------------------
def _build_rk2_driver(tmp):
    flags = ["-O0", "-ffp-contract=off"]
    subprocess.run([FC, *flags, "-c", API, "-o", str(tmp / "api.o"), "-J", str(tmp)], check=True)
    subprocess.run([FC, *flags, f"-I{tmp}", "-c", os.path.join(ROOT, "tests", "fortran", "rk2_driver.F90"), "-o", str(tmp / "rk2.o"), "-J", str(tmp)],
                   check=True)
    libdir = os.path.join(ROOT, "mom6_amd")
    subprocess.run([FC, str(tmp / "rk2.o"), str(tmp / "api.o"), f"-L{libdir}", "-lmom6hip", f"-Wl,-rpath,{libdir}", "-o", str(tmp / "rk2_driver")],
                   check=True)
    return str(tmp / "rk2_driver")


@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")
def test_rk2_driver_compiles():
    import tempfile, pathlib
    with tempfile.TemporaryDirectory() as d:
        assert os.path.exists(_build_rk2_driver(pathlib.Path(d)))


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")
def test_fortran_device_resident_rk2_steps_match_oracle(tmp_path):
    """INTEGRATION.md 2(b) end to end from Fortran: bind(c) control structures, every array allocated in HBM through the
    binding, initialize + two step_MOM_dyn_split_RK2 calls, state copied back once: bit-identical with the oracle"""
    from mom6_amd import synth
    from oracle import orc
    from helpers import interior
    exe = _build_rk2_driver(tmp_path)
    ni, nj, nk, halo = 36, 20, 4, 4
    g = synth.make_grid(ni, nj, nk, halo=halo, land_frac=0.2, seed=91, reentrant_x=True, reentrant_y=False)
    d = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=6, umax=0.1, eta_amp=0.2).items()}
    yy = np.linspace(0.0, np.pi, g.shape2(_abi.POS_U)[0])
    taux = np.ascontiguousarray(0.1 * np.cos(2 * yy)[:, None] * g.mask2dCu); tauy = np.ascontiguousarray(0.02 * g.mask2dCv)
    dt = 900.0
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([ni, nj, nk, halo, 1, 0, g.first_direction, 0], dtype="<i4").tofile(f)
        np.array([g.Angstrom_H, g.H_subroundoff, g.dZ_subroundoff, g.H_to_Z, g.Z_to_H, g.g_Earth, g.Rho0, dt], dtype="<f8").tofile(f)
        for n in _abi.ALL_METRICS:
            np.ascontiguousarray(g.metrics[n], dtype="<f8").tofile(f)
        for a in (d["u"], d["v"], d["h"], d["T"], d["S"], taux, tauy):
            np.ascontiguousarray(a, dtype="<f8").tofile(f)
    ref = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], dt, bound_coriolis=True)
    for n in range(2):
        ref.step(taux, tauy, calc_dtbt=(n == 0))
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert f"rk2_driver ok nstep={int(ref.bcs.nstep_last)} " in r.stdout
    # the debugging statistics the driver formed on the device: MOM_checksums' bit counts, MOM_coms' order-invariant sums
    from test_checksums import ref_chksum
    stats = [int(x) for x in next(l for l in r.stdout.splitlines() if l.startswith("rk2_driver stats")).split()[2:]]
    assert stats[0] == ref_chksum(g, ref.h, _abi.POS_H)[0] and stats[1] == ref_chksum(g, ref.u, _abi.POS_U, symmetric=True)[0]
    assert stats[2] == ni * nj * nk
    assert stats[3:9] == orc.reproducing_sum(g, ref.h, _abi.POS_H)["efp"] and stats[9:15] == orc.reproducing_sum(g, ref.u, _abi.POS_U)["efp"]
    raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8")
    want = [ref.u, ref.v, ref.h, ref.eta_av, ref.uhtr]
    sizes = [w.size for w in want]
    assert raw.size == sum(sizes)
    for name, a, w, pos in zip(("u", "v", "h", "eta_av", "uhtr"), np.split(raw, np.cumsum(sizes)[:-1]), want,
                               (_abi.POS_U, _abi.POS_V, _abi.POS_H, _abi.POS_H, _abi.POS_U)):
        assert bits_equal(interior(g, a.reshape(w.shape), pos), interior(g, w, pos)), name


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")
def test_viscosity_shims_match_oracle(tmp_path):
    """set_visc_init / set_viscous_BBL, vertvisc_init / vertvisc_coef / vertvisc / vertvisc_remnant, hor_visc_init /
    horizontal_viscosity through the MOM_set_visc, MOM_vert_friction and MOM_hor_visc shims (parameters by name, the
    reference's argument lists) on Fortran host arrays: bit-identical with the oracle"""
    from mom6_amd import synth
    from oracle import orc
    from helpers import interior
    exe = _build_shims(tmp_path, "visc_driver")
    ni, nj, nk, halo = 34, 18, 5, 4
    g = synth.make_grid(ni, nj, nk, halo=halo, land_frac=0.2, seed=55, reentrant_x=True, reentrant_y=False)
    d = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=8, umax=0.3, eta_amp=0.2).items()}
    yy = np.linspace(0.0, np.pi, g.shape2(_abi.POS_U)[0])
    taux = np.ascontiguousarray(0.1 * np.cos(2 * yy)[:, None] * g.mask2dCu); tauy = np.ascontiguousarray(0.02 * g.mask2dCv)
    dt = 900.0
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([ni, nj, nk, halo, 1, 0, g.first_direction, 0], dtype="<i4").tofile(f)
        np.array([g.Angstrom_H, g.H_subroundoff, g.dZ_subroundoff, g.H_to_Z, g.Z_to_H, g.g_Earth, g.Rho0, dt], dtype="<f8").tofile(f)
        for n in _abi.ALL_METRICS:
            np.ascontiguousarray(g.metrics[n], dtype="<f8").tofile(f)
        for a in (d["u"], d["v"], d["h"], d["T"], d["S"], taux, tauy):
            np.ascontiguousarray(a, dtype="<f8").tofile(f)
    # the oracle, with the parameters the driver sets by name
    U, V = _abi.POS_U, _abi.POS_V
    bb = dict(Kv_bbl_u=g.zeros2(U), Kv_bbl_v=g.zeros2(V), bbl_thick_u=g.zeros2(U), bbl_thick_v=g.zeros2(V))
    visc = orc.vertvisc_type(**bb)
    orc.set_viscous_BBL(g, orc.set_visc_cs(g, 10.0, 1.0e-4), d["u"], d["v"], d["h"], d["T"], d["S"], orc.eos("WRIGHT"), visc)
    bb = visc._keep
    vcs = orc.vertvisc_cs(g, Kv=1.0e-4, Hbbl=10.0, Hmix=20.0, Kvml_invZ2=1.0e-2)
    u1, v1 = d["u"].copy(), d["v"].copy()
    orc.vertvisc_coef(g, vcs, u1, v1, d["h"], visc, dt)
    tbx, tby = g.zeros2(U), g.zeros2(V)
    orc.vertvisc(g, vcs, u1, v1, d["h"], taux, tauy, visc, dt, taux_bot=tbx, tauy_bot=tby)
    vru, vrv = g.zeros3(U), g.zeros3(V)
    orc.vertvisc_remnant(g, vcs, visc, vru, vrv, dt)
    diffu, diffv = orc.horizontal_viscosity(g, orc.hor_visc_cs(g, dt, Smagorinsky_Ah=1, Smag_bi_const=0.06, Ah_vel_scale=0.01), d["u"], d["v"],
                                            d["h"], dt)
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "visc_driver ok ntrunc=" in r.stdout
    want = [bb["bbl_thick_u"], bb["bbl_thick_v"], bb["Kv_bbl_u"], bb["Kv_bbl_v"], u1, v1, vru, vrv, tbx, tby, diffu, diffv]
    names = ["bbl_thick_u", "bbl_thick_v", "Kv_bbl_u", "Kv_bbl_v", "u", "v", "visc_rem_u", "visc_rem_v", "taux_bot", "tauy_bot", "diffu", "diffv"]
    raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8")
    sizes = [w.size for w in want]
    assert raw.size == sum(sizes)
    for n, a, w in zip(names, np.split(raw, np.cumsum(sizes)[:-1]), want):
        pos = V if n.endswith("_v") or n in ("v", "tauy_bot", "diffv") else U
        assert bits_equal(interior(g, a.reshape(w.shape), pos), interior(g, w, pos)), n


def _build_ale_shim(tmp):
    """the MOM_ALE shim takes the place of the type-only stand-in of the same module (-DMOM6HIP_WITH_ALE_SHIM)"""
    flags = ["-cpp", "-DMOM6HIP_WITH_ALE_SHIM", "-fdefault-real-8", "-O0", "-ffp-contract=off", f"-I{STUBS}", f"-I{tmp}", "-J", str(tmp)]
    objs = []
    srcs = [os.path.join(STUBS, "mom6_stubs.F90")] + [os.path.join(FDIR, s) for s in ("mom6hip_c_api.F90", "mom6hip_MOM_glue.F90", "MOM_ALE_hip.F90")] + \
           [os.path.join(ROOT, "tests", "fortran", "ale_driver.F90")]
    for src in srcs:
        o = str(tmp / (os.path.basename(src)[:-4] + ".o"))
        subprocess.run([FC, *flags, "-c", src, "-o", o], check=True)
        objs.append(o)
    libdir = os.path.join(ROOT, "mom6_amd")
    exe = str(tmp / "ale_driver")
    subprocess.run([FC, *objs, f"-L{libdir}", "-lmom6hip", f"-Wl,-rpath,{libdir}", "-o", exe], check=True)
    return exe


@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")
def test_ale_shim_compiles():
    import tempfile, pathlib
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    with tempfile.TemporaryDirectory() as d:
        assert os.path.exists(_build_ale_shim(pathlib.Path(d)))


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")
@pytest.mark.parametrize("scheme,vel_scheme,nk", [("PPM_H4", "PLM", 6), ("PQM_IH4IH3", "PQM_IH6IH5", 12)])
def test_ale_shim_matches_oracle(tmp_path, scheme, vel_scheme, nk):
    """ALE_init (Z*, UNIFORM, PPM_H4 / PLM, REGRID_TIME_SCALE with a deep filter, REMAP_BOUNDARY_EXTRAP through
    ALE_set_extrap_boundaries), ALE_update_regrid_weights, ALE_regrid, ALE_remap_tracers, ALE_remap_set_h_vel x2,
    ALE_remap_velocities through the MOM_ALE shim on Fortran host arrays, the sequence of MOM.F90:1647-1700: the oracle's bits"""
    from mom6_amd import synth
    from oracle import orc
    from helpers import interior
    exe = _build_ale_shim(tmp_path)
    ni, nj, halo = 34, 18, 4
    g = synth.make_grid(ni, nj, nk, halo=halo, land_frac=0.2, seed=21, reentrant_x=True, reentrant_y=False)
    d = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=9, umax=0.3, eta_amp=0.5).items()}
    dt = 1800.0
    max_depth = float(g.bathyT.max())
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([ni, nj, nk, halo, 1, 0, g.first_direction, 0], dtype="<i4").tofile(f)
        np.array([g.Angstrom_H, g.H_subroundoff, g.dZ_subroundoff, g.H_to_Z, g.Z_to_H, g.g_Earth, g.Rho0, dt], dtype="<f8").tofile(f)
        for n in _abi.ALL_METRICS:
            np.ascontiguousarray(g.metrics[n], dtype="<f8").tofile(f)
        for a in (d["u"], d["v"], d["h"], d["T"], d["S"]):
            np.ascontiguousarray(a, dtype="<f8").tofile(f)
        np.array([max_depth], dtype="<f8").tofile(f)
    # the oracle with what the driver's parameters mean
    res = np.full(nk, max_depth / nk)                                   # ALE_COORDINATE_CONFIG = UNIFORM
    w = 3600.0 / (3600.0 + dt)                                          # ALE_update_regrid_weights, REGRID_TIME_SCALE = 3600
    rcs = orc.regridding_cs(res, min_thickness=1.0e-3, old_grid_weight=w, zs=0.0, zd=500.0)
    h_new, dz = orc.ale_regrid(g, rcs, d["h"])
    T, S = d["T"].copy(), d["S"].copy()
    orc.ale_remap_tracers(g, scheme, d["h"], h_new, [T, S], conc_underflow=np.array([0.0, 1.0e-30]), boundary_extrapolation=True)
    hu0, hv0 = orc.ale_remap_set_h_vel(g, d["h"])
    hu1, hv1 = orc.ale_remap_set_h_vel(g, h_new)
    u, v = d["u"].copy(), d["v"].copy()
    # (the velocities' remapping structure keeps INIT_BOUNDARY_EXTRAP = False: ALE_set_extrap_boundaries sets the tracers' only, MOM_ALE.F90:336)
    orc.ale_remap_velocities(g, vel_scheme, hu0, hv0, hu1, hv1, u, v, boundary_extrapolation=False)
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), f"REMAPPING_SCHEME={scheme}", f"VELOCITY_REMAPPING_SCHEME={vel_scheme}"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "ale_driver ok" in r.stdout
    want = [h_new, dz, T, S, hu1, hv1, u, v]
    names = ["h_new", "dzRegrid", "T", "S", "h_new_u", "h_new_v", "u", "v"]
    raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8")
    sizes = [x.size for x in want]
    assert raw.size == sum(sizes)
    U, V, H = _abi.POS_U, _abi.POS_V, _abi.POS_H
    for n, a, x in zip(names, np.split(raw, np.cumsum(sizes)[:-1]), want):
        pos = U if n in ("h_new_u", "u") else (V if n in ("h_new_v", "v") else H)
        assert bits_equal(interior(g, a.reshape(x.shape), pos), interior(g, x, pos)), n
    assert not np.array_equal(interior(g, h_new), interior(g, d["h"]))
