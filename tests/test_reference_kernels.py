"""The reference's own kernels run beside the oracle (build container only: needs /root/reference and amdflang).

The unmodified /root/reference/src/core/MOM_continuity_PPM.F90, MOM_CoriolisAdv.F90 and src/tracer/MOM_tracer_advect.F90 are compiled
where they lie (amdflang -O0 -ffp-contract=off, the reference's own memory headers by -I) against the stand-ins of tests/fortran/stubs
(parameter table, one-PE pass_var, empty diagnostics) and driven by tests/fortran/ref_kernels_driver.F90 on the inputs the oracle gets:
continuity_PPM in the two call forms of the RK2 step (BT_cont; uhbt + u_cor + BT_cont), CorAdCalc, advect_tracer.  Their outputs have to
equal oracle/*.c bit for bit.

What this is and is not: SUPPLEMENTARY evidence.  A reference build that stands on hand-written stand-ins for
MOM_grid / MOM_domains / MOM_file_parser ... pins nothing by this tier's rule, and the parity grade of these operators stays "unpinned".
It does catch a misreading shared by oracle and kernel (ADVICE r04's `I` / `i` in advect_x would have failed here), and it is the
per-kernel CPU calibration of bench.py's port (tools/calibrate_ref_kernels.py).  Nothing of it travels to the GPU box."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import bits_equal, interior
from mom6_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
FC = shutil.which("amdflang") or "/opt/rocm/bin/amdflang"
STUBS = os.path.join(ROOT, "tests", "fortran", "stubs")
REF_SOURCES = ("src/core/MOM_continuity_PPM.F90", "src/core/MOM_CoriolisAdv.F90", "src/tracer/MOM_tracer_advect.F90")
SCHEMES = {"PLM": 0, "PPM:H3": 1, "PPM": 2}

def _unlimited_stack():      # (the reference's automatic arrays)
    import resource
    resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))


pytestmark = [pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="the reference is not mounted (GPU box)"),
              pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")]

@pytest.fixture(scope="module")
def ref_builds(tmp_path_factory):
    """every reference executable of this file, built side by side (each build is a chain of modules, the builds are independent of each other):
    the fixtures below wait for their own"""
    from concurrent.futures import ThreadPoolExecutor
    jobs = {
        "visc_exe": (build_ref_visc_driver, "ref_visc", {}), "visc_full_exe": (build_ref_visc_driver_full, "ref_visc_full", {}),
        "pf_exe": (build_ref_pf_driver, "ref_pf", {}), "dyn_exe": (build_ref_dyn_driver, "ref_dyn", {}), "ref_exe": (build_ref_kernels, "ref_kernels", {}),
        "ref_bt_exe": (lambda tmp: build_ref_module_driver(tmp, ["src/core/MOM_barotropic.F90"], "bt_driver"), "ref_bt", {}),
        "mle_exe": (build_ref_mle_driver, "ref_mle", {}), "td_exe": (build_ref_td_driver, "ref_td", {}), "ale_exe": (build_ref_ale_driver, "ref_ale", {}),
        "dyn_rk2b_exe": (build_ref_dyn_driver, "ref_dyn_rk2b", dict(rk2b=True)), "tracer_exe": (build_ref_tracer_driver, "ref_tracer", {}),
        "dyn_obc_exe": (build_ref_dyn_obc_driver, "ref_dyn_obc", {}), "rad_exe": (build_ref_rad_driver, "ref_rad", {}),
        "dyn_obc_rk2b_exe": (build_ref_dyn_obc_driver, "ref_dyn_obc_rk2b", dict(rk2b=True)), "coms_exe": (build_ref_coms_driver, "ref_coms", {}),
    }
    pool = ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1))
    futures = {name: pool.submit(f, tmp_path_factory.mktemp(d), **kw) for name, (f, d, kw) in jobs.items()}
    yield futures
    pool.shutdown(wait=True, cancel_futures=True)


@pytest.fixture(scope="module")
def visc_exe(ref_builds):
    return ref_builds["visc_exe"].result()


@pytest.fixture(scope="module")
def visc_full_exe(ref_builds):
    return ref_builds["visc_full_exe"].result()


@pytest.fixture(scope="module")
def pf_exe(ref_builds):
    return ref_builds["pf_exe"].result()


@pytest.fixture(scope="module")
def dyn_exe(ref_builds):
    return ref_builds["dyn_exe"].result()



def build_ref_kernels(tmp, opt="-O0", openmp=False):
    """the stand-ins, the three reference files (in place) and the driver -> an executable in tmp"""
    flags = ["-cpp", "-fdefault-real-8", opt, "-ffp-contract=off", f"-I{REF}/config_src/memory/dynamic_symmetric", f"-I{REF}/src/framework",
             f"-I{STUBS}", f"-I{tmp}", "-J", str(tmp)] + (["-fopenmp"] if openmp else [])
    objs = []
    for src in [os.path.join(STUBS, "mom6_stubs.F90")] + [os.path.join(REF, r) for r in REF_SOURCES] + \
               [os.path.join(ROOT, "tests", "fortran", "ref_kernels_driver.F90")]:
        o = os.path.join(str(tmp), os.path.basename(src)[:-4] + ".o")
        r = subprocess.run([FC, *flags, "-c", src, "-o", o], capture_output=True, text=True)
        assert r.returncode == 0, f"{src}:\n" + r.stderr[-3000:]
        objs.append(o)
    exe = os.path.join(str(tmp), "ref_kernels_driver")
    r = subprocess.run([FC, *objs, "-o", exe] + (["-fopenmp"] if openmp else []), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def write_case(path, ni=30, nj=14, nk=4, scheme="PPM:H3", x_first=-1, seed=77, ntr=3, land_frac=0.2):
    """the input file of ref_kernels_driver.F90; returns the grid, the oracle's inputs and what the oracle makes of them.
    Closed in x and y: the stand-in's group passes (advect_tracer's) do nothing, which is what one closed tile needs."""
    from mom6_amd import synth
    from oracle import orc
    halo = 4
    g = synth.make_grid(ni, nj, nk, halo=halo, land_frac=land_frac, seed=seed, reentrant_x=False, reentrant_y=False)
    d = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=5, umax=0.3, eta_amp=0.2).items()}
    kk = (np.arange(nk) + 0.5) / nk
    vru = np.ascontiguousarray(np.clip(1.0 - 0.8 * kk[:, None, None] ** 2 + 0 * d["u"], 0.0, 1.0) * (g.mask2dCu[None] > 0))
    vrv = np.ascontiguousarray(np.clip(1.0 - 0.8 * kk[:, None, None] ** 2 + 0 * d["v"], 0.0, 1.0) * (g.mask2dCv[None] > 0))
    dt = 900.0
    ccs = orc.continuity_cs(nk, g.Angstrom_H)
    hp = d["h"].copy(); uh = np.zeros_like(d["u"]); vh = np.zeros_like(d["v"])
    arrs, bt = orc.make_bt_cont(g, with_h=True)
    orc.continuity(g, ccs, d["u"], d["v"], d["h"], hp, uh, vh, dt, visc_rem_u=vru, visc_rem_v=vrv, bt_cont=bt)
    first = {n: arrs[n].copy() for n in arrs}
    uhbt = np.ascontiguousarray(uh.sum(0) * 1.02); vhbt = np.ascontiguousarray(vh.sum(0) * 0.98)
    adv = synth.make_advection_state(g, ntr=ntr, seed=seed + 1, hot_frac=0.01, vanish_frac=0.05)
    adv = {k: (v.numpy() if k != "tr" else [t.numpy() for t in v]) for k, v in adv.items()}
    dt_adv = 3600.0
    with open(path, "wb") as f:
        np.array([ni, nj, nk, halo, 0, 0, g.first_direction, 0], dtype="<i4").tofile(f)
        np.array([g.Angstrom_H, g.H_subroundoff, g.dZ_subroundoff, g.H_to_Z, g.Z_to_H, g.g_Earth, g.Rho0, dt], dtype="<f8").tofile(f)
        for n in _abi.ALL_METRICS:
            np.ascontiguousarray(g.metrics[n], dtype="<f8").tofile(f)
        for a in (d["u"], d["v"], d["h"], uhbt, vhbt, vru, vrv, d["T"], d["S"]):
            np.ascontiguousarray(a, dtype="<f8").tofile(f)
        np.array([ntr, x_first, 0, SCHEMES[scheme]], dtype="<i4").tofile(f)
        np.array([dt_adv], dtype="<f8").tofile(f)
        for a in [adv["h_end"], adv["uhtr"], adv["vhtr"]] + adv["tr"]:
            np.ascontiguousarray(a, dtype="<f8").tofile(f)
    hp2 = d["h"].copy(); uh2 = np.zeros_like(d["u"]); vh2 = np.zeros_like(d["v"]); ucor = np.zeros_like(d["u"]); vcor = np.zeros_like(d["v"])
    orc.continuity(g, ccs, d["u"], d["v"], d["h"], hp2, uh2, vh2, dt, uhbt=uhbt, vhbt=vhbt, visc_rem_u=vru, visc_rem_v=vrv, u_cor=ucor,
                   v_cor=vcor, bt_cont=bt)
    CAu, CAv = orc.coradcalc(g, d["u"], d["v"], d["h"], uh2, vh2, bound_coriolis=True)
    tr = [t.copy() for t in adv["tr"]]
    orc.advect_tracer(g, adv["h_end"], adv["uhtr"], adv["vhtr"], dt_adv, 900.0, scheme, tr, x_first=None if x_first < 0 else bool(x_first))
    names = ["hp", "uh", "vh", "hp2", "uh2", "vh2", "u_cor", "v_cor", "CAu", "CAv", "FA_u_W0", "FA_u_WW", "FA_u_E0", "FA_u_EE", "uBT_WW", "uBT_EE",
             "FA_v_S0", "FA_v_SS", "FA_v_N0", "FA_v_NN", "vBT_SS", "vBT_NN", "h_u", "h_v"] + [f"tr{m + 1}" for m in range(ntr)]
    want = [hp, uh, vh, hp2, uh2, vh2, ucor, vcor, CAu, CAv] + [arrs[n] for n in names[10:24]] + tr
    return g, names, want, dict(d=d, adv=adv, vru=vru, vrv=vrv, uhbt=uhbt, vhbt=vhbt, dt=dt, dt_adv=dt_adv, first_bt=first)


def position_of(n):
    if n in ("uh", "uh2", "u_cor", "CAu", "h_u") or n.startswith(("FA_u", "uBT")):
        return _abi.POS_U
    if n in ("vh", "vh2", "v_cor", "CAv", "h_v") or n.startswith(("FA_v", "vBT")):
        return _abi.POS_V
    return _abi.POS_H


@pytest.fixture(scope="module")
def ref_exe(ref_builds):
    return ref_builds["ref_exe"].result()


@pytest.mark.parametrize("scheme,x_first,shape", [("PPM:H3", -1, (30, 14, 4)), ("PLM", 0, (22, 17, 3)), ("PPM", 1, (26, 12, 5))])
def test_the_reference_kernels_equal_the_oracle_bit_for_bit(ref_exe, tmp_path, scheme, x_first, shape):
    g, names, want, _ = write_case(str(tmp_path / "in.bin"), *shape, scheme=scheme, x_first=x_first, seed=70 + shape[0])
    r = subprocess.run([ref_exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0 and "ref_kernels_driver ok" in r.stdout, r.stderr[-2000:]
    raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8")
    sizes = [w.size for w in want]
    assert raw.size == sum(sizes)
    for n, a, w in zip(names, np.split(raw, np.cumsum(sizes)[:-1]), want):
        a = a.reshape(w.shape)
        pos = position_of(n)
        assert bits_equal(interior(g, a, pos), interior(g, w, pos)), (n, np.argwhere(interior(g, a, pos) != interior(g, w, pos))[:4])
    assert not bits_equal(want[-1], np.zeros_like(want[-1]))


# ---- with open boundaries: the OBC branches of the three reference kernels ---------------------------------------------------------------
def write_obc(path, g, OBC):
    """the OBC file of ref_kernels_driver.F90: test_testing_configs.write_obc_file's format (what dyn_driver.F90 reads), then the segments'
    tangential_vel / tangential_grad, then their tracer registries"""
    from test_testing_configs import write_obc_file
    write_obc_file(path, g, OBC)
    with open(path, "ab") as f:
        for s in OBC.segment:
            if s.on_pe:
                for a in (s.tangential_vel, s.tangential_grad):
                    np.ascontiguousarray(a, dtype="<f8").tofile(f)
        for s in OBC.segment:
            if s.on_pe:
                regs = s.tr_Reg or []
                np.array([len(regs)], dtype="<i4").tofile(f)
                for r in regs:
                    tres = r.get("tres")
                    np.array([r["ntr_index"], int(tres is not None)], dtype="<i4").tofile(f)
                    np.array([r.get("OBC_inflow_conc", 0.0)], dtype="<f8").tofile(f)
                    if tres is not None:
                        np.ascontiguousarray(tres, dtype="<f8").tofile(f)


OBC_SETS = [
    # the four sides of tc3 (open, Flather + Orlanski) with the free-slip options of tc3
    (["J=N,I=N:0,FLATHER,ORLANSKI", "J=0,I=0:N,FLATHER,ORLANSKI", "I=N,J=0:N,FLATHER,ORLANSKI", "I=0,J=N:0,FLATHER,ORLANSKI"],
     dict(freeslip_vorticity=True), "PLM", -1),
    # specified (SIMPLE) segments inside the domain and on two sides, computed vorticity from the segments' tangential velocities
    (["J=N,I=N:0,FLATHER,ORLANSKI", "I=N,J=0:N,SIMPLE", "I=9,J=0:N,SIMPLE", "J=7,I=N:0,SIMPLE"], dict(computed_vorticity=True), "PPM:H3", 1),
    # gradient and Orlanski segments, specified vorticity (segment%tangential_grad), zero vorticity in a second run of the same set
    (["J=0,I=0:N,GRADIENT", "I=0,J=N:0,ORLANSKI", "I=N,J=0:N,FLATHER,ORLANSKI", "J=11,I=0:N,SIMPLE"], dict(specified_vorticity=True), "PPM", 0),
    (["J=0,I=0:N,GRADIENT", "I=0,J=N:0,ORLANSKI", "I=N,J=0:N,FLATHER,ORLANSKI", "J=11,I=0:N,SIMPLE"], dict(zero_vorticity=True), "PLM", 1),
]


@pytest.mark.parametrize("case", range(len(OBC_SETS)))
def test_the_reference_kernels_with_open_boundaries_equal_the_oracle_bit_for_bit(ref_exe, tmp_path, case):
    """continuity_PPM (both call forms), CorAdCalc and advect_tracer (with segment tracer registries) of the reference, OBC associated"""
    from mom6_amd import synth
    from mom6_amd.open_boundary import ocean_OBC_type
    from oracle import orc
    from test_continuity_obc import open_faces
    segs, flags, scheme, x_first = OBC_SETS[case]
    ni, nj, nk, halo, ntr = 22, 16, 4, 4, 3
    g = synth.make_grid(ni, nj, nk, halo=halo, land_frac=0.1, seed=40 + case, reentrant_x=False, reentrant_y=False)
    OBC = ocean_OBC_type(g, segs, **flags)
    open_faces(g, OBC)
    rng = np.random.default_rng(case)
    d = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=5 + case, umax=0.3, eta_amp=0.2).items()}
    d["u"] = np.ascontiguousarray(d["u"] + 0.05 * rng.standard_normal(d["u"].shape) * (OBC.segnum_u != 0)[None])
    d["v"] = np.ascontiguousarray(d["v"] + 0.05 * rng.standard_normal(d["v"].shape) * (OBC.segnum_v != 0)[None])
    for n, s in enumerate(OBC.segment):
        if not s.on_pe:
            continue
        if s.specified:
            s.normal_vel[:] = 0.1 * rng.standard_normal(s.normal_vel.shape)
            s.normal_vel[rng.random(s.normal_vel.shape) < 0.2] = 0.0
            s.normal_trans[:] = s.normal_vel * (3.0e4 * (5.0 + 50.0 * rng.random(s.normal_vel.shape)))
        s.normal_vel_bt = np.zeros(s.normal_vel.shape[1:]); s.SSH = np.zeros(s.normal_vel.shape[1:])
        s.tangential_vel[:] = 0.1 * rng.standard_normal(s.tangential_vel.shape)
        s.tangential_grad[:] = 1.0e-6 * rng.standard_normal(s.tangential_grad.shape)
        s.tr_Reg = [dict(ntr_index=1, tres=np.ascontiguousarray(5.0 + rng.random(s.normal_vel.shape))), dict(ntr_index=3, OBC_inflow_conc=0.25 + 0.1 * n)]
    kk = (np.arange(nk) + 0.5) / nk
    vru = np.ascontiguousarray(np.clip(1.0 - 0.8 * kk[:, None, None] ** 2 + 0 * d["u"], 0.0, 1.0) * (g.mask2dCu[None] > 0))
    vrv = np.ascontiguousarray(np.clip(1.0 - 0.8 * kk[:, None, None] ** 2 + 0 * d["v"], 0.0, 1.0) * (g.mask2dCv[None] > 0))
    dt, dt_adv = 900.0, 3600.0
    ccs = orc.continuity_cs(nk, g.Angstrom_H)
    hp = d["h"].copy(); uh = np.zeros_like(d["u"]); vh = np.zeros_like(d["v"])
    arrs, bt = orc.make_bt_cont(g, with_h=True)
    orc.continuity(g, ccs, d["u"], d["v"], d["h"], hp, uh, vh, dt, visc_rem_u=vru, visc_rem_v=vrv, bt_cont=bt, OBC=OBC)
    uhbt = np.ascontiguousarray(uh.sum(0) * 1.02); vhbt = np.ascontiguousarray(vh.sum(0) * 0.98)
    adv = synth.make_advection_state(g, ntr=ntr, seed=90 + case, hot_frac=0.004, vanish_frac=0.05, cfl=0.15)
    adv = {k: (v.numpy() if k != "tr" else [t.numpy() for t in v]) for k, v in adv.items()}
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([ni, nj, nk, halo, 0, 0, g.first_direction, 0], dtype="<i4").tofile(f)
        np.array([g.Angstrom_H, g.H_subroundoff, g.dZ_subroundoff, g.H_to_Z, g.Z_to_H, g.g_Earth, g.Rho0, dt], dtype="<f8").tofile(f)
        for n in _abi.ALL_METRICS:
            np.ascontiguousarray(g.metrics[n], dtype="<f8").tofile(f)
        for a in (d["u"], d["v"], d["h"], uhbt, vhbt, vru, vrv, d["T"], d["S"]):
            np.ascontiguousarray(a, dtype="<f8").tofile(f)
        np.array([ntr, x_first, 0, SCHEMES[scheme]], dtype="<i4").tofile(f)
        np.array([dt_adv], dtype="<f8").tofile(f)
        for a in [adv["h_end"], adv["uhtr"], adv["vhtr"]] + adv["tr"]:
            np.ascontiguousarray(a, dtype="<f8").tofile(f)
    write_obc(str(tmp_path / "obc.bin"), g, OBC)
    hp2 = d["h"].copy(); uh2 = np.zeros_like(d["u"]); vh2 = np.zeros_like(d["v"]); ucor = np.zeros_like(d["u"]); vcor = np.zeros_like(d["v"])
    orc.continuity(g, ccs, d["u"], d["v"], d["h"], hp2, uh2, vh2, dt, uhbt=uhbt, vhbt=vhbt, visc_rem_u=vru, visc_rem_v=vrv, u_cor=ucor,
                   v_cor=vcor, bt_cont=bt, OBC=OBC)
    CAu, CAv = orc.coradcalc(g, d["u"], d["v"], d["h"], uh2, vh2, bound_coriolis=True, OBC=OBC)
    tr = [t.copy() for t in adv["tr"]]
    orc.advect_tracer(g, adv["h_end"], adv["uhtr"], adv["vhtr"], dt_adv, 900.0, scheme, tr, x_first=None if x_first < 0 else bool(x_first), OBC=OBC)
    names = ["hp", "uh", "vh", "hp2", "uh2", "vh2", "u_cor", "v_cor", "CAu", "CAv", "FA_u_W0", "FA_u_WW", "FA_u_E0", "FA_u_EE", "uBT_WW", "uBT_EE",
             "FA_v_S0", "FA_v_SS", "FA_v_N0", "FA_v_NN", "vBT_SS", "vBT_NN", "h_u", "h_v"] + [f"tr{m + 1}" for m in range(ntr)]
    want = [hp, uh, vh, hp2, uh2, vh2, ucor, vcor, CAu, CAv] + [arrs[n] for n in names[10:24]] + tr
    r = subprocess.run([ref_exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), "0", str(tmp_path / "obc.bin")], capture_output=True, text=True)
    assert r.returncode == 0 and "ref_kernels_driver ok" in r.stdout, r.stderr[-2000:]
    raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8")
    sizes = [w.size for w in want]
    assert raw.size == sum(sizes)
    for n, a, w in zip(names, np.split(raw, np.cumsum(sizes)[:-1]), want):
        a = a.reshape(w.shape)
        pos = position_of(n)
        assert bits_equal(interior(g, a, pos), interior(g, w, pos)), (case, n, np.argwhere(interior(g, a, pos) != interior(g, w, pos))[:4])


# ---- MOM_barotropic: btstep --------------------------------------------------------------------------------------------------------------
def build_ref_module_driver(tmp, ref_sources, driver, opt="-O0"):
    """the stand-ins, reference files (in place) and one of the repository's module drivers compiled with -DREFERENCE_KERNELS (the same
    program that drives the module shim on the GPU, without the library's glue)"""
    flags = ["-cpp", "-DREFERENCE_KERNELS", "-fdefault-real-8", opt, "-ffp-contract=off", f"-I{REF}/config_src/memory/dynamic_symmetric",
             f"-I{REF}/src/framework", f"-I{STUBS}", f"-I{tmp}", "-J", str(tmp)]
    objs = []
    for src in [os.path.join(STUBS, "mom6_stubs.F90")] + [os.path.join(REF, r) for r in ref_sources] + \
               [os.path.join(ROOT, "tests", "fortran", driver + ".F90")]:
        o = os.path.join(str(tmp), os.path.basename(src)[:-4] + ".o")
        r = subprocess.run([FC, *flags, "-c", src, "-o", o], capture_output=True, text=True)
        assert r.returncode == 0, f"{src}:\n" + r.stderr[-3000:]
        objs.append(o)
    exe = os.path.join(str(tmp), driver + "_ref")
    r = subprocess.run([FC, *objs, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.fixture(scope="module")
def ref_bt_exe(ref_builds):
    return ref_builds["ref_bt_exe"].result()


@pytest.mark.parametrize("kw,params,exact", [(dict(strong_drag=1), ["BT_STRONG_DRAG=True"], True), (dict(strong_drag=1, ni=40, nj=30, seed=11), ["BT_STRONG_DRAG=True"], True),
                                             (dict(), [], False), (dict(ni=40, nj=30, seed=11), [], False)])
def test_the_reference_btstep_equals_the_oracle(ref_bt_exe, tmp_path, kw, params, exact):
    """barotropic_init, btcalc, bt_mass_source and btstep of the reference's MOM_barotropic.F90 with the argument list of the RK2 step's
    call (BT_cont, layer fluxes, eta_av; 13 barotropic steps) on a closed basin (the re-entrant form of btstep, with the exchanges of its wide-halo
    march, runs inside test_reference_dynamical_core_equals_the_oracle[*-reentrant_x], where its inputs carry the halos a step gives them).  With BT_STRONG_DRAG (no real power in bt_rem, :1525): every output bit for bit.  With the default, bt_rem =
    av_rem ** (1/nstep) (:1529) is the libm power of the build, where oracle and library take a correctly rounded one (DESIGN.md section 3:
    one ulp apart in ~0.1 % of arguments): the outputs agree in every bit except downstream of the one or two faces where the two powers
    differ, and there within a few ulps -- the documented deviation, seen here against the reference itself."""
    from test_fortran_abi import _bt_case
    g, names, want, nstep = _bt_case(str(tmp_path / "in.bin"), reentrant=(False, False), **kw)
    assert nstep >= 10
    r = subprocess.run([ref_bt_exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), ""] + params, capture_output=True, text=True)
    assert r.returncode == 0 and "bt_driver ok" in r.stdout, r.stderr[-2000:]
    raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8")
    sizes = [w.size for w in want]
    assert raw.size == sum(sizes)
    ndiff = 0
    for n, a, w in zip(names, np.split(raw, np.cumsum(sizes)[:-1]), want):
        a = a.reshape(w.shape)
        pos = _abi.POS_U if n in ("accel_layer_u", "uhbtav", "ubtav") else (_abi.POS_V if n in ("accel_layer_v", "vhbtav", "vbtav") else _abi.POS_H)
        ia, iw = interior(g, a, pos), interior(g, w, pos)
        if exact:
            assert bits_equal(ia, iw), (n, np.argwhere(ia != iw)[:4])
        else:
            d = ia != iw
            ndiff += int(d.sum())
            assert d.mean() <= 0.01 and np.all(np.abs(ia - iw)[d] <= 1.0e-14 * np.abs(iw)[d] + 1.0e-30), (n, int(d.sum()))
    if not exact:
        assert ndiff > 0      # (if this ever fails the libm power has become correctly rounded on these arguments: tighten the test)


# ---- the reference's own MOM_vert_friction.F90 and MOM_hor_visc.F90 beside the oracle ----------------------------------------------------
VISC_SOURCES = ("src/parameterizations/vertical/MOM_vert_friction.F90", "src/parameterizations/lateral/MOM_Zanna_Bolton.F90",
                "src/core/MOM_barotropic.F90", "src/parameterizations/lateral/MOM_hor_visc.F90")


def build_ref_visc_driver(tmp):
    """the stand-ins (+ tests/fortran/stubs/mom6_stubs_visc.F90), the reference's vert_friction and hor_visc with the two modules hor_visc
    imports types from (in place), and tests/fortran/visc_driver.F90 built with -DREFERENCE_KERNELS"""
    flags = ["-cpp", "-fdefault-real-8", "-O0", "-ffp-contract=off", "-DREFERENCE_KERNELS", f"-I{REF}/config_src/memory/dynamic_symmetric",
             f"-I{REF}/src/framework", f"-I{STUBS}", f"-I{tmp}", "-J", str(tmp)]
    objs = []
    for src in [os.path.join(STUBS, "mom6_stubs.F90"), os.path.join(STUBS, "mom6_stubs_visc.F90")] + \
               [os.path.join(REF, r) for r in VISC_SOURCES] + [os.path.join(ROOT, "tests", "fortran", "visc_driver.F90")]:
        o = os.path.join(str(tmp), os.path.basename(src)[:-4] + ".o")
        r = subprocess.run([FC, *flags, "-c", src, "-o", o], capture_output=True, text=True)
        assert r.returncode == 0, f"{src}:\n" + r.stderr[-3000:]
        objs.append(o)
    exe = os.path.join(str(tmp), "visc_ref_driver")
    r = subprocess.run([FC, *objs, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


# option sets: (NAME=VALUE arguments of the driver, the oracle's vertvisc_cs keywords, its hor_visc_cs keywords)
VISC_SETS = {
    "default": ([], {}, dict(Smagorinsky_Ah=1, Smag_bi_const=0.06, Ah_vel_scale=0.01)),
    "harmonic_laplacian": (["HARMONIC_VISC=True", "HARMONIC_BL_SCALE=0.5", "KV_EXTRA_BBL=2.0e-4", "LAPLACIAN=True", "KH=20.0", "KH_VEL_SCALE=0.005",
                            "SMAGORINSKY_KH=True", "SMAG_LAP_CONST=0.15", "SMAGORINSKY_AH=False", "AH=1.0e9"],
                           dict(harmonic_visc=True, harm_BL_val=0.5, Kv_extra_bbl=2.0e-4),
                           dict(Laplacian=1, Kh=20.0, Kh_vel_scale=0.005, Smagorinsky_Kh=1, Smag_Lap_const=0.15, Smagorinsky_Ah=0, Ah=1.0e9,
                                Ah_vel_scale=0.01)),
    "laplacian_noslip": (["LAPLACIAN=True", "BIHARMONIC=False", "KH_VEL_SCALE=0.01", "SMAGORINSKY_KH=True", "SMAG_LAP_CONST=0.15", "NOSLIP=True",
                          "SMAGORINSKY_AH=False", "BETTER_BOUND_KH=False"], {},
                         dict(Laplacian=1, biharmonic=0, Kh_vel_scale=0.01, Smagorinsky_Kh=1, Smag_Lap_const=0.15, no_slip=1, better_bound_Kh=0)),
    "direct_stress_maxvel": (["DIRECT_STRESS=True", "HMIX_STRESS=15.0", "CFL_BASED_TRUNCATIONS=False", "MAXVEL=0.25", "VEL_UNDERFLOW=1.0e-30",
                              "BOUND_CORIOLIS=True", "BETTER_BOUND_AH=False"],
                             dict(direct_stress=True, Hmix_stress=15.0, CFL_based_trunc=False, maxvel=0.25, vel_underflow=1.0e-30),
                             dict(Smagorinsky_Ah=1, Smag_bi_const=0.06, Ah_vel_scale=0.01, bound_Coriolis=1, better_bound_Ah=0, bound_Cor_vel=0.25)),
}


@pytest.mark.parametrize("ni,nj,nk,seed,opts", [(34, 18, 5, 55, "default"), (21, 26, 9, 7, "default"), (34, 18, 5, 56, "harmonic_laplacian"),
                                                (30, 22, 6, 57, "direct_stress_maxvel"), (26, 20, 4, 58, "laplacian_noslip")])
def test_reference_vert_friction_and_hor_visc_equal_the_oracle(tmp_path, visc_exe, ni, nj, nk, seed, opts):
    """vertvisc_init / vertvisc_coef / vertvisc / vertvisc_remnant and hor_visc_init / horizontal_viscosity of the reference's own modules, with
    the parameters tests/fortran/visc_driver.F90 sets by name (the set tests/test_fortran_abi.py runs through the shims on the GPU) and the
    bottom boundary layer the oracle's set_viscous_BBL leaves: velocities, visc_rem, the bottom stresses and the viscous accelerations
    equal the oracle's bit for bit"""
    from mom6_amd import synth
    from oracle import orc
    exe = visc_exe
    halo = 4
    g = synth.make_grid(ni, nj, nk, halo=halo, land_frac=0.2, seed=seed, reentrant_x=False, reentrant_y=False)
    d = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=8, umax=0.3, eta_amp=0.2).items()}
    U, V = _abi.POS_U, _abi.POS_V
    yy = np.linspace(0.0, np.pi, g.shape2(U)[0])
    taux = np.ascontiguousarray(0.1 * np.cos(2 * yy)[:, None] * g.mask2dCu); tauy = np.ascontiguousarray(0.02 * g.mask2dCv)
    dt = 900.0
    bb = dict(Kv_bbl_u=g.zeros2(U), Kv_bbl_v=g.zeros2(V), bbl_thick_u=g.zeros2(U), bbl_thick_v=g.zeros2(V))
    visc = orc.vertvisc_type(**bb)
    orc.set_viscous_BBL(g, orc.set_visc_cs(g, 10.0, 1.0e-4), d["u"], d["v"], d["h"], d["T"], d["S"], orc.eos("WRIGHT"), visc)
    bb = visc._keep
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([ni, nj, nk, halo, 0, 0, g.first_direction, 0], dtype="<i4").tofile(f)
        np.array([g.Angstrom_H, g.H_subroundoff, g.dZ_subroundoff, g.H_to_Z, g.Z_to_H, g.g_Earth, g.Rho0, dt], dtype="<f8").tofile(f)
        for n in _abi.ALL_METRICS:
            np.ascontiguousarray(g.metrics[n], dtype="<f8").tofile(f)
        for a in (d["u"], d["v"], d["h"], d["T"], d["S"], taux, tauy, bb["bbl_thick_u"], bb["bbl_thick_v"], bb["Kv_bbl_u"], bb["Kv_bbl_v"]):
            np.ascontiguousarray(a, dtype="<f8").tofile(f)
    args, vv_kw, hv_kw = VISC_SETS[opts]
    vcs = orc.vertvisc_cs(g, **dict(dict(Kv=1.0e-4, Hbbl=10.0, Hmix=20.0, Kvml_invZ2=1.0e-2), **vv_kw))
    u1, v1 = d["u"].copy(), d["v"].copy()
    orc.vertvisc_coef(g, vcs, u1, v1, d["h"], visc, dt)
    tbx, tby = g.zeros2(U), g.zeros2(V)
    orc.vertvisc(g, vcs, u1, v1, d["h"], taux, tauy, visc, dt, taux_bot=tbx, tauy_bot=tby)
    vru, vrv = g.zeros3(U), g.zeros3(V)
    orc.vertvisc_remnant(g, vcs, visc, vru, vrv, dt)
    diffu, diffv = orc.horizontal_viscosity(g, orc.hor_visc_cs(g, dt, **hv_kw), d["u"], d["v"], d["h"], dt)
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")] + args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    want = [bb["bbl_thick_u"], bb["bbl_thick_v"], bb["Kv_bbl_u"], bb["Kv_bbl_v"], u1, v1, vru, vrv, tbx, tby, diffu, diffv]
    names = ["bbl_thick_u", "bbl_thick_v", "Kv_bbl_u", "Kv_bbl_v", "u", "v", "visc_rem_u", "visc_rem_v", "taux_bot", "tauy_bot", "diffu", "diffv"]
    raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8")
    sizes = [w.size for w in want]
    assert raw.size == sum(sizes)
    bad = []
    for n, a, w in zip(names, np.split(raw, np.cumsum(sizes)[:-1]), want):
        pos = V if n.endswith("_v") or n in ("v", "tauy_bot", "diffv") else U
        if not bits_equal(interior(g, a.reshape(w.shape), pos), interior(g, w, pos)):
            bad.append(n)
    assert not bad, bad


# ---- the reference's own PressureForce_FV_Bouss (with its density integrals and equation-of-state stack) beside the oracle ------------------
PF_SOURCES = ("src/core/MOM_density_integrals.F90", "src/core/MOM_PressureForce_Montgomery.F90", "src/core/MOM_PressureForce_FV.F90")


def build_ref_pf_driver(tmp):
    """the stand-ins with the reference's whole src/equation_of_state and src/ALE/PLM_functions.F90 #included in place (-DREF_EOS -DREF_PF), the
    reference's density integrals, Montgomery module (Set_pbce_Bouss) and PressureForce_FV, and tests/fortran/ref_pf_driver.F90"""
    flags = ["-cpp", "-fdefault-real-8", "-O0", "-ffp-contract=off", "-DREF_EOS", "-DREF_PF", "-DREF_PF_MONT",
             f"-I{REF}/config_src/memory/dynamic_symmetric", f"-I{REF}/src/framework", f"-I{REF}/src/equation_of_state", f"-I{REF}/src/ALE",
             f"-I{STUBS}", f"-I{tmp}", "-J", str(tmp)]
    objs = []
    for src in [os.path.join(STUBS, "mom6_stubs.F90")] + [os.path.join(REF, r) for r in PF_SOURCES] + \
               [os.path.join(ROOT, "tests", "fortran", "ref_pf_driver.F90")]:
        o = os.path.join(str(tmp), os.path.basename(src)[:-4] + ".o")
        r = subprocess.run([FC, *flags, "-c", src, "-o", o], capture_output=True, text=True)
        assert r.returncode == 0, f"{src}:\n" + r.stderr[-3000:]
        objs.append(o)
    exe = os.path.join(str(tmp), "ref_pf_driver")
    r = subprocess.run([FC, *objs, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


# (NAME=VALUE arguments of the driver, the oracle's equation of state, its pressureforce_cs keywords, a surface pressure?)
PF_SETS = {
    "wright_plm": ([], "WRIGHT", {}, False),
    "wright_plm_massw_psurf": (["MASS_WEIGHT_IN_PRESSURE_GRADIENT=True", "BOUNDARY_EXTRAPOLATION_PRESSURE=False"], "WRIGHT",
                               dict(useMassWghtInterp=True, boundary_extrap=False), True),
    "wright_pcm": (["RECONSTRUCT_FOR_PRESSURE=False"], "WRIGHT", dict(reconstruct=False), False),
    "linear_plm": (["EQN_OF_STATE=LINEAR", "RHO_T0_S0=1000.0", "DRHO_DT=-0.2", "DRHO_DS=0.8"], "LINEAR", {}, False),
    "wright_full_plm": (["EQN_OF_STATE=WRIGHT_FULL"], "WRIGHT_FULL", {}, True),
    "unesco_plm": (["EQN_OF_STATE=UNESCO"], "UNESCO", {}, False),
    # BOUSSINESQ = False: PressureForce_FV_nonBouss (int_spec_vol_dp_generic_plm, Set_pbce_nonBouss), thicknesses in kg m-2
    "nonbouss_wright": (["BOUSSINESQ=False"], "WRIGHT", {}, True),
    "nonbouss_unesco_massw": (["BOUSSINESQ=False", "EQN_OF_STATE=UNESCO", "MASS_WEIGHT_IN_PRESSURE_GRADIENT=True", "BOUNDARY_EXTRAPOLATION_PRESSURE=False"],
                              "UNESCO", dict(useMassWghtInterp=True, boundary_extrap=False), False),
    "nonbouss_linear": (["BOUSSINESQ=False", "EQN_OF_STATE=LINEAR", "RHO_T0_S0=1000.0", "DRHO_DT=-0.2", "DRHO_DS=0.8"], "LINEAR", {}, False),
}


@pytest.mark.parametrize("ni,nj,nk,seed,opts", [(30, 14, 6, 61, "wright_plm"), (22, 25, 9, 62, "wright_plm_massw_psurf"), (30, 14, 6, 63, "wright_pcm"),
                                                (26, 18, 5, 64, "linear_plm"), (26, 18, 5, 65, "wright_full_plm"), (24, 16, 5, 66, "unesco_plm"),
                                                (28, 16, 6, 67, "nonbouss_wright"), (22, 20, 5, 68, "nonbouss_unesco_massw"), (24, 14, 5, 69, "nonbouss_linear")])
def test_reference_pressureforce_equals_the_oracle(tmp_path, pf_exe, ni, nj, nk, seed, opts):
    """PressureForce_FV_Bouss of the reference -- int_density_dz_generic_plm / int_density_dz, the PLM edge values of T and S from its own slope
    functions, its equation-of-state modules, Set_pbce_Bouss -- on the oracle's inputs: PFu, PFv, pbce and eta equal the oracle's bit for bit"""
    from mom6_amd import synth
    from oracle import orc
    exe = pf_exe
    args, eos_form, cs_kw, with_p = PF_SETS[opts]
    halo = 4
    g = synth.make_grid(ni, nj, nk, halo=halo, land_frac=0.2, seed=seed, reentrant_x=False, reentrant_y=False)
    d = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=seed + 1, umax=0.3, eta_amp=0.3).items()}
    H = _abi.POS_H
    rng = np.random.default_rng(seed)
    p_atm = np.ascontiguousarray(1.0e5 + 2.0e3 * rng.standard_normal(g.shape2(H))) if with_p else None
    nonbouss = "BOUSSINESQ=False" in args
    if nonbouss:
        d["h"] = np.ascontiguousarray(d["h"] * g.Rho0)      # layer masses [kg m-2]
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([ni, nj, nk, halo, 0, 0, g.first_direction, 1 if with_p else 0], dtype="<i4").tofile(f)
        np.array([g.Angstrom_H, g.H_subroundoff, g.dZ_subroundoff, g.H_to_Z, g.Z_to_H, g.g_Earth, g.Rho0, 900.0], dtype="<f8").tofile(f)
        for n in _abi.ALL_METRICS:
            np.ascontiguousarray(g.metrics[n], dtype="<f8").tofile(f)
        for a in (d["h"], d["T"], d["S"]) + ((p_atm,) if with_p else ()):
            np.ascontiguousarray(a, dtype="<f8").tofile(f)
    if nonbouss:
        PFu, PFv, pbce, eta = orc.pressureforce_nonbouss(g, orc.pressureforce_cs(g, **cs_kw), orc.eos(eos_form), d["h"], d["T"], d["S"], p_atm, H_to_RZ=1.0)
    else:
        PFu, PFv, pbce, eta = orc.pressureforce(g, orc.pressureforce_cs(g, **cs_kw), orc.eos(eos_form), d["h"], d["T"], d["S"], p_atm=p_atm)
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")] + args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8")
    want = [PFu, PFv, pbce, eta]
    sizes = [w.size for w in want]
    assert raw.size == sum(sizes)
    bad = []
    for n, a, w, pos in zip(("PFu", "PFv", "pbce", "eta"), np.split(raw, np.cumsum(sizes)[:-1]), want, (_abi.POS_U, _abi.POS_V, H, H)):
        if not bits_equal(interior(g, a.reshape(w.shape), pos), interior(g, w, pos)):
            bad.append(n)
    assert not bad, bad


# ---- the same with the reference's own MOM_set_viscosity.F90 (set_viscous_BBL with BBL_USE_EOS through its own MOM_EOS) in front ------------
def build_ref_visc_driver_full(tmp):
    """as build_ref_visc_driver, with src/parameterizations/vertical/MOM_set_viscosity.F90, src/framework/MOM_intrinsic_functions.F90 and the
    reference's equation-of-state stack in place (-DREF_EOS -DREF_SET_VISC; tests/fortran/stubs/mom6_stubs_setvisc.F90)"""
    flags = ["-cpp", "-fdefault-real-8", "-O0", "-ffp-contract=off", "-DREFERENCE_KERNELS", "-DREF_EOS", "-DREF_SET_VISC",
             f"-I{REF}/config_src/memory/dynamic_symmetric", f"-I{REF}/src/framework", f"-I{REF}/src/equation_of_state", f"-I{REF}/src/ALE",
             f"-I{STUBS}", f"-I{tmp}", "-J", str(tmp)]
    srcs = [os.path.join(STUBS, "mom6_stubs.F90"), os.path.join(STUBS, "mom6_stubs_setvisc.F90"),
            os.path.join(REF, "src/framework/MOM_intrinsic_functions.F90"), os.path.join(REF, "src/parameterizations/vertical/MOM_set_viscosity.F90"),
            os.path.join(STUBS, "mom6_stubs_visc.F90")] + [os.path.join(REF, r) for r in VISC_SOURCES] + \
           [os.path.join(ROOT, "tests", "fortran", "visc_driver.F90")]
    objs = []
    for src in srcs:
        o = os.path.join(str(tmp), os.path.basename(src)[:-4] + ".o")
        r = subprocess.run([FC, *flags, "-c", src, "-o", o], capture_output=True, text=True)
        assert r.returncode == 0, f"{src}:\n" + r.stderr[-3000:]
        objs.append(o)
    exe = os.path.join(str(tmp), "visc_ref_driver_full")
    r = subprocess.run([FC, *objs, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


BBL_SETS = {
    "bbl_default": ([], {}),
    "bbl_linear_correct": (["LINEAR_DRAG=True", "CORRECT_BBL_BOUNDS=True", "DRAG_BG_VEL=0.05"],
                           dict(linear_drag=True, correct_BBL_bounds=True, drag_bg_vel=0.05)),
    "bbl_thin_min": (["BBL_THICK_MIN=0.5", "CDRAG=0.002", "KV_BBL_MIN=2.0e-4"], dict(BBL_thick_min=0.5, cdrag=0.002, Kv_BBL_min=2.0e-4)),
}


@pytest.mark.parametrize("ni,nj,nk,seed,opts", [(34, 18, 5, 71, "bbl_default"), (22, 24, 8, 72, "bbl_linear_correct"), (28, 16, 6, 73, "bbl_thin_min")])
def test_reference_set_viscous_bbl_and_the_viscosities_equal_the_oracle(tmp_path, visc_full_exe, ni, nj, nk, seed, opts):
    """set_visc_init / set_viscous_BBL (BBL_USE_EOS, Wright through the reference's MOM_EOS) / set_viscous_ML, then the vertical and horizontal
    viscosities on what it leaves -- all the reference's own modules: the bottom boundary layer's thicknesses and viscosities and everything
    downstream equal the oracle's bit for bit"""
    from mom6_amd import synth
    from oracle import orc
    exe = visc_full_exe
    args, sv_kw = BBL_SETS[opts]
    halo = 4
    g = synth.make_grid(ni, nj, nk, halo=halo, land_frac=0.2, seed=seed, reentrant_x=False, reentrant_y=False)
    d = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=seed + 1, umax=0.3, eta_amp=0.2).items()}
    U, V = _abi.POS_U, _abi.POS_V
    yy = np.linspace(0.0, np.pi, g.shape2(U)[0])
    taux = np.ascontiguousarray(0.1 * np.cos(2 * yy)[:, None] * g.mask2dCu); tauy = np.ascontiguousarray(0.02 * g.mask2dCv)
    dt = 900.0
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([ni, nj, nk, halo, 0, 0, g.first_direction, 1], dtype="<i4").tofile(f)
        np.array([g.Angstrom_H, g.H_subroundoff, g.dZ_subroundoff, g.H_to_Z, g.Z_to_H, g.g_Earth, g.Rho0, dt], dtype="<f8").tofile(f)
        for n in _abi.ALL_METRICS:
            np.ascontiguousarray(g.metrics[n], dtype="<f8").tofile(f)
        for a in (d["u"], d["v"], d["h"], d["T"], d["S"], taux, tauy):
            np.ascontiguousarray(a, dtype="<f8").tofile(f)
    bb = dict(Kv_bbl_u=g.zeros2(U), Kv_bbl_v=g.zeros2(V), bbl_thick_u=g.zeros2(U), bbl_thick_v=g.zeros2(V))
    visc = orc.vertvisc_type(**bb)
    orc.set_viscous_BBL(g, orc.set_visc_cs(g, 10.0, 1.0e-4, **sv_kw), d["u"], d["v"], d["h"], d["T"], d["S"], orc.eos("WRIGHT"), visc)
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
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), "EQN_OF_STATE=WRIGHT"] + args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    want = [bb["bbl_thick_u"], bb["bbl_thick_v"], bb["Kv_bbl_u"], bb["Kv_bbl_v"], u1, v1, vru, vrv, tbx, tby, diffu, diffv]
    names = ["bbl_thick_u", "bbl_thick_v", "Kv_bbl_u", "Kv_bbl_v", "u", "v", "visc_rem_u", "visc_rem_v", "taux_bot", "tauy_bot", "diffu", "diffv"]
    raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8")
    sizes = [w.size for w in want]
    assert raw.size == sum(sizes)
    bad = []
    for n, a, w in zip(names, np.split(raw, np.cumsum(sizes)[:-1]), want):
        pos = V if n.endswith("_v") or n in ("v", "tauy_bot", "diffv") else U
        if not bits_equal(interior(g, a.reshape(w.shape), pos), interior(g, w, pos)):
            bad.append(n)
    assert not bad, bad


# ---- the reference's whole split RK2 dynamical core beside the oracle -----------------------------------------------------------------------
CORE_SOURCES = ("src/core/MOM_density_integrals.F90", "src/core/MOM_PressureForce_Montgomery.F90", "src/core/MOM_PressureForce_FV.F90",
                "src/core/MOM_PressureForce.F90", "src/core/MOM_continuity_PPM.F90", "src/core/MOM_continuity.F90", "src/core/MOM_CoriolisAdv.F90",
                "src/core/MOM_dynamics_split_RK2.F90")


def build_ref_dyn_driver(tmp, opt="-O0", rk2b=False):
    """tests/fortran/dyn_driver.F90 (-DREFERENCE_KERNELS) on the reference's OWN MOM_dynamics_split_RK2.F90 and everything it steps through:
    set_viscosity, vert_friction, hor_visc, barotropic, the pressure force with its density integrals and equation-of-state stack,
    continuity, CoriolisAdv -- each compiled where it lies against the stand-ins"""
    flags = ["-cpp", "-fdefault-real-8", opt, "-ffp-contract=off", "-DREFERENCE_KERNELS", "-DREF_EOS", "-DREF_PF", "-DREF_PF_MONT", "-DREF_SET_VISC"] + \
            (["-DREF_RK2B"] if rk2b else []) + [
             f"-I{REF}/config_src/memory/dynamic_symmetric", f"-I{REF}/src/framework", f"-I{REF}/src/equation_of_state", f"-I{REF}/src/ALE",
             f"-I{STUBS}", f"-I{tmp}", "-J", str(tmp)]
    srcs = [os.path.join(STUBS, "mom6_stubs.F90"), os.path.join(STUBS, "mom6_stubs_setvisc.F90"),
            os.path.join(REF, "src/framework/MOM_intrinsic_functions.F90"), os.path.join(REF, "src/parameterizations/vertical/MOM_set_viscosity.F90"),
            os.path.join(STUBS, "mom6_stubs_visc.F90")] + [os.path.join(REF, r) for r in VISC_SOURCES + CORE_SOURCES] + \
           [os.path.join(ROOT, "tests", "fortran", "dyn_driver.F90")]
    if rk2b:
        srcs.insert(-1, os.path.join(REF, "src/core/MOM_dynamics_split_RK2b.F90"))
    objs = []
    for src in srcs:
        o = os.path.join(str(tmp), os.path.basename(src)[:-4] + ".o")
        r = subprocess.run([FC, *flags, "-c", src, "-o", o], capture_output=True, text=True)
        assert r.returncode == 0, f"{src}:\n" + r.stderr[-3000:]
        objs.append(o)
    exe = os.path.join(str(tmp), "dyn_ref_driver")
    r = subprocess.run([FC, *objs, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


# the settings of bench.py's step as a parameter file: Wright, PLM pressure reconstruction, BT_cont from the layer
# continuity, Sadourny energy with BOUND_CORIOLIS, biharmonic Smagorinsky with the better bounds, BOTTOMDRAGLAW with BBL_USE_EOS, KV_ML_INVZ2
BENCH_LIKE = dict(shape=(16, 12, 6), pairs="""
        USE_REGRIDDING = True
        DT = 900.0
        BOUND_CORIOLIS = True
        SMAGORINSKY_AH = True
        SMAG_BI_CONST = 0.06
        AH_VEL_SCALE = 0.01
        KV = 1.0E-04
        HMIX_FIXED = 20.0
        KV_ML_INVZ2 = 0.01
        HBBL = 10.0
        CDRAG = 0.003
        """)


@pytest.mark.parametrize("reentrant", [False, True], ids=["closed", "reentrant_x"])
@pytest.mark.parametrize("name", ["tc4", "tc2", "tc1", "bench_like"])
def test_reference_dynamical_core_equals_the_oracle(tmp_path, dyn_exe, name, reentrant, monkeypatch):
    """three steps of the reference's step_MOM_dyn_split_RK2 (after its own set_viscous_BBL each), every module of the dynamical core the
    reference's own, with the transcribed parameter sets of .testing/tc4, tc2 and tc1 and with the settings of bench.py's step, on a closed
    basin and on a zonally re-entrant one (the stand-in's pass_var and group passes wrap the re-entrant direction on the one PE: every pass of
    the step and of btstep's wide-halo march is exercised): u, v, h, uh, vh, uhtr, vhtr, eta_av (and MEKE%mom_src, visc%nkml_visc_u/v where the
    set has them) equal the oracle's DynState.step bit for bit"""
    import test_testing_configs as tc
    from oracle import orc
    exe = dyn_exe
    nsteps = 3
    base = BENCH_LIKE if name == "bench_like" else tc.TC_INPUT[name]
    monkeypatch.setitem(tc.TC_INPUT, name, dict(base, pairs=base["pairs"] + f"\n        REENTRANT_X = {reentrant}\n"))
    state = tc.case_state(name)
    g, d, taux, tauy, ustar, bbl, Rlay, g_prime = state
    assert g.reentrant_x == reentrant and not g.reentrant_y
    if reentrant:      # the forcing's halos as a run has them (the synthetic friction velocity is drawn halo and all: not periodic)
        ustar = np.ascontiguousarray(ustar); orc.halo_update(g, ustar, _abi.POS_H)
        state = (g, d, taux, tauy, ustar, bbl, Rlay, g_prime)
    tc.write_case(tmp_path, name, nsteps, False, state, bbl_mode=1)
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "params.txt")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:] + r.stdout[-2000:]
    assert "dyn_driver ok" in r.stdout
    st, calc, _ = tc.oracle_for(name, g, d, ustar, bbl, Rlay, g_prime)
    for n in range(nsteps):
        st.bbl()
        st.step(taux, tauy, calc_dtbt=calc(n))
    got = tc.read_out(str(tmp_path / "out.bin"), g, meke=st.mom_src is not None)
    want = dict(u=st.u, v=st.v, h=st.h, uh=st.uh, vh=st.vh, uhtr=st.uhtr, vhtr=st.vhtr, eta_av=st.eta_av)
    if "nkml_visc_u" in st.visc._keep:
        want.update(nkml_visc_u=st.visc._keep["nkml_visc_u"], nkml_visc_v=st.visc._keep["nkml_visc_v"])
    bad = [(n, float(np.abs(got[n] - want[n]).max())) for n, pos, nd in tc.OUT
           if n in want and not bits_equal(interior(g, got[n], pos), interior(g, want[n], pos))]
    if st.mom_src is not None and not bits_equal(interior(g, got["mom_src"], _abi.POS_H), interior(g, st.mom_src, _abi.POS_H)):
        bad.append(("mom_src", float(np.abs(got["mom_src"] - st.mom_src).max())))
    assert not bad, bad


@pytest.mark.parametrize("reentrant", [False, True], ids=["closed", "reentrant_x"])
def test_reference_dynamical_core_at_a_size_where_the_two_powers_differ(tmp_path, dyn_exe, reentrant, monkeypatch):
    """On a larger basin btstep's one real power, bt_rem = av_rem ** (1/nstep) (MOM_barotropic.F90:1529), meets arguments where the host's
    libm pow (the reference build's) and the correctly rounded power of oracle and library are an ulp apart.  With the
    oracle told to take libm's (ORC_BT_LIBM_POW) two steps of the reference's dynamical core at 120x80x20 (closed, and zonally re-entrant as the
    bench's domain is) equal it bit for bit; with its own, the two
    agree except downstream of those faces, there within 1e-10 relative."""
    import test_testing_configs as tc
    name = "bench_like"
    monkeypatch.setitem(tc.TC_INPUT, name, dict(shape=(120, 80, 20), pairs=BENCH_LIKE["pairs"] + f"\n        REENTRANT_X = {reentrant}\n"))
    nsteps = 2
    state = tc.case_state(name)
    g, d, taux, tauy, ustar, bbl, Rlay, g_prime = state
    if reentrant:      # (the forcing's halos as a run has them)
        from oracle import orc
        ustar = np.ascontiguousarray(ustar); orc.halo_update(g, ustar, _abi.POS_H)
        state = (g, d, taux, tauy, ustar, bbl, Rlay, g_prime)
    tc.write_case(tmp_path, name, nsteps, False, state, bbl_mode=1)
    import resource
    unlimited = lambda: resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))      # (the reference's automatic arrays)
    r = subprocess.run([dyn_exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "params.txt")], capture_output=True, text=True,
                       preexec_fn=unlimited)
    assert r.returncode == 0, r.stderr[-3000:]
    got = tc.read_out(str(tmp_path / "out.bin"), g, meke=False)

    def oracle_fields():
        st, calc, _ = tc.oracle_for(name, g, d, ustar, bbl, Rlay, g_prime)
        for n in range(nsteps):
            st.bbl(); st.step(taux, tauy, calc_dtbt=calc(n))
        return dict(u=st.u, v=st.v, h=st.h, uh=st.uh, vh=st.vh, uhtr=st.uhtr, vhtr=st.vhtr, eta_av=st.eta_av)
    monkeypatch.setenv("ORC_BT_LIBM_POW", "1")
    want = oracle_fields()
    bad = [n for n, pos, nd in tc.OUT if n in want and not bits_equal(interior(g, got[n], pos), interior(g, want[n], pos))]
    assert not bad, bad
    monkeypatch.delenv("ORC_BT_LIBM_POW")
    own = oracle_fields()
    ndiff = 0
    for n, pos, nd in tc.OUT:
        if n in own:
            a, b = interior(g, got[n], pos), interior(g, own[n], pos)
            dd = a != b
            assert dd.mean() <= 0.05 and np.all(np.abs(a - b)[dd] <= 1.0e-10 * np.abs(b)[dd] + 1.0e-30), (n, int(dd.sum()))
            ndiff += int(dd.sum())
    assert ndiff > 0      # (the two powers did differ somewhere: otherwise this test shows nothing)


# ---- the reference's own MOM_mixed_layer_restrat.F90 beside the oracle ------------------------------------------------------------------------
def build_ref_mle_driver(tmp):
    flags = ["-cpp", "-fdefault-real-8", "-O0", "-ffp-contract=off", "-DREFERENCE_KERNELS", "-DREF_EOS", f"-I{REF}/config_src/memory/dynamic_symmetric",
             f"-I{REF}/src/framework", f"-I{REF}/src/equation_of_state", f"-I{REF}/src/ALE", f"-I{STUBS}", f"-I{tmp}", "-J", str(tmp)]
    objs = []
    for src in [os.path.join(STUBS, "mom6_stubs.F90"), os.path.join(REF, "src/framework/MOM_intrinsic_functions.F90"),
                os.path.join(REF, "src/parameterizations/lateral/MOM_mixed_layer_restrat.F90"), os.path.join(ROOT, "tests", "fortran", "mle_driver.F90")]:
        o = os.path.join(str(tmp), os.path.basename(src)[:-4] + ".o")
        r = subprocess.run([FC, *flags, "-c", src, "-o", o], capture_output=True, text=True)
        assert r.returncode == 0, f"{src}:\n" + r.stderr[-3000:]
        objs.append(o)
    exe = os.path.join(str(tmp), "mle_ref_driver")
    r = subprocess.run([FC, *objs, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.fixture(scope="module")
def mle_exe(ref_builds):
    return ref_builds["mle_exe"].result()


@pytest.mark.parametrize("topo", [(False, False), (True, False)], ids=["closed", "reentrant_x"])
def test_reference_mixedlayer_restrat_equals_the_oracle(tmp_path, mle_exe, topo):
    """mixedlayer_restrat_register_restarts, mixedlayer_restrat_init and two calls of mixedlayer_restrat of the reference's own module (the OM4
    form and the bulk-mixed-layer form; the running means of the mixed layer depth; its own MOM_EOS), every variant of
    tests/test_mixedlayer_restrat.py on a closed basin: h, uhtr, vhtr equal the oracle's bit for bit"""
    import test_mixedlayer_restrat as tm
    g, d = tm.case(36, 22, 6, reentrant_x=topo[0], reentrant_y=topo[1])
    bad = []
    for name in tm.VARIANTS:
        ref, nrest = tm._write_mle_case(tmp_path, g, d, name)
        r = subprocess.run([mle_exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "params.txt")], capture_output=True, text=True)
        assert r.returncode == 0 and "mle_driver ok" in r.stdout, (name, r.stdout[-200:], r.stderr[-1500:])
        raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8")
        want = [ref["h"], ref["uhtr"], ref["vhtr"]]
        assert raw.size == sum(w.size for w in want), name
        for n, a, w, pos in zip(("h", "uhtr", "vhtr"), np.split(raw, np.cumsum([w.size for w in want])[:-1]), want, (_abi.POS_H, _abi.POS_U, _abi.POS_V)):
            if not bits_equal(interior(g, a.reshape(w.shape), pos), interior(g, w, pos)):
                bad.append((name, n, float(np.abs(a.reshape(w.shape) - w).max())))
    assert not bad, bad


# ---- the reference's own MOM_thickness_diffuse.F90 beside the oracle --------------------------------------------------------------------------
def build_ref_td_driver(tmp):
    flags = ["-cpp", "-fdefault-real-8", "-O0", "-ffp-contract=off", "-DREFERENCE_KERNELS", "-DREF_EOS", "-DREF_INTERFACE_HEIGHTS",
             f"-I{REF}/config_src/memory/dynamic_symmetric", f"-I{REF}/src/framework", f"-I{REF}/src/equation_of_state", f"-I{REF}/src/ALE",
             f"-I{STUBS}", f"-I{tmp}", "-J", str(tmp)]
    objs = []
    for src in [os.path.join(STUBS, "mom6_stubs.F90"), os.path.join(REF, "src/core/MOM_density_integrals.F90"),
                os.path.join(REF, "src/core/MOM_interface_heights.F90"), os.path.join(REF, "src/core/MOM_isopycnal_slopes.F90"),
                os.path.join(REF, "src/parameterizations/lateral/MOM_thickness_diffuse.F90"), os.path.join(ROOT, "tests", "fortran", "td_driver.F90")]:
        o = os.path.join(str(tmp), os.path.basename(src)[:-4] + ".o")
        r = subprocess.run([FC, *flags, "-c", src, "-o", o], capture_output=True, text=True)
        assert r.returncode == 0, f"{src}:\n" + r.stderr[-3000:]
        objs.append(o)
    exe = os.path.join(str(tmp), "td_ref_driver")
    r = subprocess.run([FC, *objs, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.fixture(scope="module")
def td_exe(ref_builds):
    return ref_builds["td_exe"].result()


@pytest.mark.parametrize("topo", [(False, False), (True, False)], ids=["closed", "reentrant_x"])
def test_reference_thickness_diffuse_equals_the_oracle(tmp_path, td_exe, topo):
    """thickness_diffuse_init and thickness_diffuse of the reference's own module -- with its MOM_isopycnal_slopes (vert_fill_TS),
    MOM_interface_heights (find_eta), density integrals and equation of state -- for every variant of tests/test_thickness_diffuse.py on a
    closed basin: h, uhtr, vhtr, CDp%uhGM, CDp%vhGM and MEKE%GM_src equal the oracle's bit for bit"""
    import test_thickness_diffuse as tt
    g, d = tt.case(36, 22, 6, reentrant_x=topo[0], reentrant_y=topo[1])
    bad = []
    for name in tt.VARIANTS:
        ref, opt = tt._write_td_case(tmp_path, g, d, name)
        r = subprocess.run([td_exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "params.txt")], capture_output=True, text=True)
        assert r.returncode == 0 and "td_driver ok" in r.stdout, (name, r.stdout[-200:], r.stderr[-1500:])
        raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8")
        want = [ref["h"], ref["uhtr"], ref["vhtr"], ref["uhGM"], ref["vhGM"]] + ([ref["GM_src"]] if opt[5] else [])
        assert raw.size == sum(w.size for w in want), name
        poss = (_abi.POS_H, _abi.POS_U, _abi.POS_V, _abi.POS_U, _abi.POS_V, _abi.POS_H)
        for n, a, w, pos in zip(("h", "uhtr", "vhtr", "uhGM", "vhGM", "GM_src"), np.split(raw, np.cumsum([w.size for w in want])[:-1]), want, poss):
            if not bits_equal(interior(g, a.reshape(w.shape), pos), interior(g, w, pos)):
                dd = interior(g, a.reshape(w.shape), pos) != interior(g, w, pos)
                bad.append((name, n, int(dd.sum()), float(np.abs(a.reshape(w.shape) - w).max())))
    assert not bad, bad


# ---- the reference's own MOM_ALE.F90 (z* regridding, remapping of tracers and velocities) beside the oracle --------------------------------------
ALE_SOURCES = ("src/core/MOM_density_integrals.F90", "src/core/MOM_interface_heights.F90", "src/ALE/regrid_consts.F90", "src/ALE/regrid_solvers.F90",
               "src/ALE/polynomial_functions.F90", "src/ALE/regrid_edge_values.F90", "src/ALE/PCM_functions.F90", "src/ALE/PLM_functions.F90",
               "src/ALE/PPM_functions.F90", "src/ALE/PQM_functions.F90", "src/ALE/P1M_functions.F90", "src/ALE/P3M_functions.F90",
               "src/ALE/regrid_interp.F90", "src/ALE/MOM_hybgen_remap.F90", "src/ALE/remapping_attic.F90", "src/ALE/MOM_remapping.F90",
               "src/ALE/coord_zlike.F90", "src/ALE/coord_sigma.F90", "src/ALE/coord_rho.F90", "src/ALE/coord_hycom.F90", "src/ALE/coord_adapt.F90",
               "src/ALE/MOM_hybgen_regrid.F90", "src/ALE/MOM_hybgen_unmix.F90", "src/ALE/MOM_regridding.F90", "src/ALE/MOM_ALE.F90")


def build_ref_ale_driver(tmp):
    """tests/fortran/ale_driver.F90 (-DREFERENCE_KERNELS) on the reference's OWN MOM_ALE.F90 with all of src/ALE under it (25 files in place; the
    stand-ins #include the reference's MOM_string_functions.F90 and equation-of-state stack as well: -DREF_ALE -DREF_EOS)"""
    flags = ["-cpp", "-fdefault-real-8", "-O0", "-ffp-contract=off", "-DREFERENCE_KERNELS", "-DREF_EOS", "-DREF_INTERFACE_HEIGHTS", "-DREF_ALE",
             f"-I{REF}/config_src/memory/dynamic_symmetric", f"-I{REF}/src/framework", f"-I{REF}/src/equation_of_state", f"-I{REF}/src/ALE",
             f"-I{STUBS}", f"-I{tmp}", "-J", str(tmp)]
    objs = []
    for src in [os.path.join(STUBS, "mom6_stubs.F90")] + [os.path.join(REF, r) for r in ALE_SOURCES] + [os.path.join(ROOT, "tests", "fortran", "ale_driver.F90")]:
        o = os.path.join(str(tmp), os.path.basename(src)[:-4] + ".o")
        r = subprocess.run([FC, *flags, "-c", src, "-o", o], capture_output=True, text=True)
        assert r.returncode == 0, f"{src}:\n" + r.stderr[-3000:]
        objs.append(o)
    exe = os.path.join(str(tmp), "ale_ref_driver")
    r = subprocess.run([FC, *objs, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.fixture(scope="module")
def ale_exe(ref_builds):
    return ref_builds["ale_exe"].result()


@pytest.mark.parametrize("scheme,vel_scheme,extrap", [("PPM_H4", "PLM", True), ("PPM_H4", "PPM_H4", False), ("PLM", "PLM", False), ("PPM_IH4", "PLM", True),
                                                      ("PCM", "PCM", False), ("PPM_CW", "PPM_CW", False), ("PPM_HYBGEN", "PLM_HYBGEN", False),
                                                      ("WENO_HYBGEN", "PLM", False), ("PQM_IH4IH3", "PQM_IH4IH3", True), ("PQM_IH4IH3", "PLM", False),
                                                      ("PQM_IH6IH5", "PQM_IH6IH5", True), ("PQM_IH6IH5", "PLM", False)])
def test_reference_ale_regrid_and_remap_equal_the_oracle(tmp_path, ale_exe, scheme, vel_scheme, extrap):
    """ALE_init (Z*, UNIFORM resolution, REGRID_TIME_SCALE with the deep filter), ALE_update_regrid_weights, ALE_regrid, ALE_remap_tracers,
    ALE_remap_set_h_vel x2, ALE_remap_velocities of the reference's own MOM_ALE / MOM_regridding / coord_zlike / MOM_remapping, the sequence of
    MOM.F90:1647-1700: the new grid, the interface movement, the remapped T, S, u, v and the face thicknesses equal the oracle's bit for bit,
    for the ten remapping schemes the library provides"""
    from mom6_amd import synth
    from oracle import orc
    ni, nj, nk, halo = 34, 18, (14 if scheme.startswith("PQM") else 6), 4
    g = synth.make_grid(ni, nj, nk, halo=halo, land_frac=0.2, seed=21, reentrant_x=False, reentrant_y=False)
    d = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=9, umax=0.3, eta_amp=0.5).items()}
    if scheme.startswith("PQM"):      # rough columns beside smooth ones: the limiter's inflexion branches and the boundary cells' rational functions
        rng = np.random.default_rng(5)
        for n, amp in (("T", 3.0), ("S", 1.0), ("u", 0.2), ("v", 0.2)):
            col = rng.choice([0.0, 0.02, 1.0], size=d[n].shape[1:])[None]
            d[n] = np.ascontiguousarray(d[n] + amp * col * rng.standard_normal(d[n].shape) * (d[n] != 0.0))
    dt = 1800.0
    max_depth = float(g.bathyT.max())
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([ni, nj, nk, halo, 0, 0, g.first_direction, 0], dtype="<i4").tofile(f)
        np.array([g.Angstrom_H, g.H_subroundoff, g.dZ_subroundoff, g.H_to_Z, g.Z_to_H, g.g_Earth, g.Rho0, dt], dtype="<f8").tofile(f)
        for n in _abi.ALL_METRICS:
            np.ascontiguousarray(g.metrics[n], dtype="<f8").tofile(f)
        for a in (d["u"], d["v"], d["h"], d["T"], d["S"]):
            np.ascontiguousarray(a, dtype="<f8").tofile(f)
        np.array([max_depth], dtype="<f8").tofile(f)
    res = np.full(nk, max_depth / nk)
    w = 3600.0 / (3600.0 + dt)
    rcs = orc.regridding_cs(res, min_thickness=1.0e-3, old_grid_weight=w, zs=0.0, zd=500.0)
    h_new, dz = orc.ale_regrid(g, rcs, d["h"])
    T, S = d["T"].copy(), d["S"].copy()
    orc.ale_remap_tracers(g, scheme, d["h"], h_new, [T, S], conc_underflow=np.array([0.0, 1.0e-30]), boundary_extrapolation=extrap)
    hu0, hv0 = orc.ale_remap_set_h_vel(g, d["h"])
    hu1, hv1 = orc.ale_remap_set_h_vel(g, h_new)
    u, v = d["u"].copy(), d["v"].copy()
    # (the velocities' remapping structure keeps INIT_BOUNDARY_EXTRAP = False: ALE_set_extrap_boundaries sets the tracers' only, MOM_ALE.F90:336)
    orc.ale_remap_velocities(g, vel_scheme, hu0, hv0, hu1, hv1, u, v, boundary_extrapolation=False)
    args = [f"REMAPPING_SCHEME={scheme}", f"VELOCITY_REMAPPING_SCHEME={vel_scheme}", f"REMAP_BOUNDARY_EXTRAP={extrap}"]
    r = subprocess.run([ale_exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")] + args, capture_output=True, text=True)
    assert r.returncode == 0 and "ale_driver ok" in r.stdout, r.stderr[-2000:]
    want = [h_new, dz, T, S, hu1, hv1, u, v]
    names = ["h_new", "dzRegrid", "T", "S", "h_new_u", "h_new_v", "u", "v"]
    raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8")
    sizes = [x.size for x in want]
    assert raw.size == sum(sizes)
    U, V, H = _abi.POS_U, _abi.POS_V, _abi.POS_H
    bad = []
    for n, a, x in zip(names, np.split(raw, np.cumsum(sizes)[:-1]), want):
        pos = U if n in ("h_new_u", "u") else (V if n in ("h_new_v", "v") else H)
        if not bits_equal(interior(g, a.reshape(x.shape), pos), interior(g, x, pos)):
            dd = interior(g, a.reshape(x.shape), pos) != interior(g, x, pos)
            bad.append((n, int(dd.sum()), float(np.abs(a.reshape(x.shape) - x).max())))
    assert not bad, bad
    assert not np.array_equal(interior(g, h_new), interior(g, d["h"]))


@pytest.fixture(scope="module")
def dyn_rk2b_exe(ref_builds):
    return ref_builds["dyn_rk2b_exe"].result()


@pytest.mark.parametrize("reentrant", [False, True], ids=["closed", "reentrant_x"])
@pytest.mark.parametrize("name", ["tc4", "bench_like"])
def test_reference_rk2b_core_equals_the_oracle(tmp_path, dyn_rk2b_exe, name, reentrant, monkeypatch):
    """SPLIT_RK2B: three steps of the reference's own step_MOM_dyn_split_RK2b (MOM_dynamics_split_RK2b.F90 in place, every module under it the
    reference's) with the tc4 set and the bench's settings on a closed basin: the filtered velocities, h, the transports and eta_av equal the
    oracle's DynState(rk2b=True).step bit for bit"""
    import functools
    import test_testing_configs as tc
    from oracle import orc
    base = BENCH_LIKE if name == "bench_like" else tc.TC_INPUT[name]
    monkeypatch.setitem(tc.TC_INPUT, name, dict(base, pairs=base["pairs"] + f"\n        REENTRANT_X = {reentrant}\n"))
    monkeypatch.setattr(orc, "DynState", functools.partial(orc.DynState, rk2b=True))
    nsteps = 3
    state = tc.case_state(name)
    g, d, taux, tauy, ustar, bbl, Rlay, g_prime = state
    if reentrant:
        ustar = np.ascontiguousarray(ustar); orc.halo_update(g, ustar, _abi.POS_H)
        state = (g, d, taux, tauy, ustar, bbl, Rlay, g_prime)
    tc.write_case(tmp_path, name, nsteps, False, state, bbl_mode=1)
    r = subprocess.run([dyn_rk2b_exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "params.txt")], capture_output=True, text=True)
    assert r.returncode == 0 and "dyn_driver ok" in r.stdout, r.stderr[-3000:] + r.stdout[-2000:]
    st, calc, _ = tc.oracle_for(name, g, d, ustar, bbl, Rlay, g_prime)
    assert st.rk2b
    for n in range(nsteps):
        st.bbl()
        st.step(taux, tauy, calc_dtbt=calc(n))
    got = tc.read_out(str(tmp_path / "out.bin"), g, meke=False)
    want = dict(u=st.u, v=st.v, h=st.h, uh=st.uh, vh=st.vh, uhtr=st.uhtr, vhtr=st.vhtr, eta_av=st.eta_av)
    bad = [(n, float(np.abs(got[n] - want[n]).max())) for n, pos, nd in tc.OUT
           if n in want and not bits_equal(interior(g, got[n], pos), interior(g, want[n], pos))]
    assert not bad, bad


# ---- the reference's own MOM_tracer_advect.F90 and MOM_tracer_hor_diff.F90 (with neutral diffusion and the epipycnal mixed-layer exchange) ------------
TRACER_SOURCES = tuple(s for s in ALE_SOURCES if s != "src/ALE/MOM_ALE.F90" and "hybgen_unmix" not in s) + (
    "src/tracer/MOM_tracer_advect.F90", "src/tracer/MOM_hor_bnd_diffusion.F90", "src/tracer/MOM_neutral_diffusion.F90", "src/tracer/MOM_tracer_hor_diff.F90")


def build_ref_tracer_driver(tmp):
    """tests/fortran/tracer_driver.F90 (-DREFERENCE_KERNELS) on the reference's OWN MOM_tracer_advect.F90 and MOM_tracer_hor_diff.F90, with its
    MOM_neutral_diffusion.F90, MOM_hor_bnd_diffusion.F90, remapping stack and equation of state in place under them"""
    flags = ["-cpp", "-fdefault-real-8", "-O0", "-ffp-contract=off", "-DREFERENCE_KERNELS", "-DREF_EOS", "-DREF_INTERFACE_HEIGHTS", "-DREF_ALE",
             f"-I{REF}/config_src/memory/dynamic_symmetric", f"-I{REF}/src/framework", f"-I{REF}/src/equation_of_state", f"-I{REF}/src/ALE",
             f"-I{STUBS}", f"-I{tmp}", "-J", str(tmp)]
    objs = []
    for src in [os.path.join(STUBS, "mom6_stubs.F90")] + [os.path.join(REF, r) for r in TRACER_SOURCES] + [os.path.join(ROOT, "tests", "fortran", "tracer_driver.F90")]:
        o = os.path.join(str(tmp), os.path.basename(src)[:-4] + ".o")
        r = subprocess.run([FC, *flags, "-c", src, "-o", o], capture_output=True, text=True)
        assert r.returncode == 0, f"{src}:\n" + r.stderr[-3000:]
        objs.append(o)
    exe = os.path.join(str(tmp), "tracer_ref_driver")
    r = subprocess.run([FC, *objs, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.fixture(scope="module")
def tracer_exe(ref_builds):
    return ref_builds["tracer_exe"].result()


@pytest.mark.parametrize("topo", [(False, False), (True, False)], ids=["closed", "reentrant_x"])
def test_reference_tracer_advect_and_hordiff_equal_the_oracle(tmp_path, tracer_exe, topo):
    """advect_tracer then tracer_hordiff of the reference's own modules on a closed basin, for constant KHTR, every variable-mixing set of
    tests/test_tracer_hor_diff.py and neutral diffusion (whole column, and below visc%h_ML): the tracers equal the oracle's bit for bit"""
    import test_tracer_hor_diff as th
    g, h, tr = th.case(36, 22, 4, reentrant=topo)
    bad = []
    for name in [None, "neutral", "neutral_interior"] + list(th.VM):
        ref = th._write_tracer_case(tmp_path, g, h, tr, name)
        r = subprocess.run([tracer_exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "params.txt")], capture_output=True, text=True)
        assert r.returncode == 0 and "tracer_driver ok" in r.stdout, (name, r.stdout[-200:], r.stderr[-1500:])
        raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8").reshape((len(tr),) + tr[0].shape)
        for m, w in enumerate(ref):
            if not bits_equal(interior(g, raw[m]), interior(g, w)):
                bad.append((name, m, int((interior(g, raw[m]) != interior(g, w)).sum()), float(np.abs(interior(g, raw[m]) - interior(g, w)).max())))
    assert not bad, bad


@pytest.mark.parametrize("topo", [(False, False), (True, False)], ids=["closed", "reentrant_x"])
def test_reference_epipycnal_diffusion_equals_the_oracle(tmp_path, tracer_exe, topo):
    """DIFFUSE_ML_TO_INTERIOR of the reference's own tracer_epipycnal_ML_diff on a layered state at rest in a closed basin, as .testing/tc1 sets it
    and with the later answer date: the tracers equal the oracle's bit for bit"""
    import test_epipycnal as te
    from oracle import orc
    g, h, tr, eos, Rlay = te.layered_case(ni=30, nj=16, nk=8, reentrant=topo)
    bad = []
    for params in [dict(KHTR=800.0, ML_KHTR_SCALE=0.0), dict(KHTR=800.0), dict(KHTR=2.0e5, CHECK_DIFFUSIVE_CFL=True, HOR_DIFF_ANSWER_DATE=20240401)]:
        dt = te._write_layered_tracer_case(tmp_path, g, h, tr, "WRIGHT", Rlay, params, False)
        ref = [t.copy() for t in tr]
        orc.advect_tracer(g, h, np.zeros(g.shape3(_abi.POS_U)), np.zeros(g.shape3(_abi.POS_V)), dt, 900.0, "PPM:H3", ref)
        for t in ref:
            orc.halo_update(g, t, _abi.POS_H)
        orc.tracer_hordiff(g, h, dt, ref, params["KHTR"], check_diffusive_CFL=params.get("CHECK_DIFFUSIVE_CFL", False),
                           epipycnal=te.epi(eos, Rlay, ML_KhTr_scale=params.get("ML_KHTR_SCALE", 1.0), answer_date=params.get("HOR_DIFF_ANSWER_DATE", 20240101)))
        r = subprocess.run([tracer_exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "params.txt")], capture_output=True, text=True)
        assert r.returncode == 0 and "tracer_driver ok" in r.stdout, (params, r.stdout[-200:], r.stderr[-1500:])
        raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8").reshape((len(tr),) + tr[0].shape)
        for m, w in enumerate(ref):
            if not bits_equal(interior(g, raw[m]), interior(g, w)):
                bad.append((params, m, int((interior(g, raw[m]) != interior(g, w)).sum()), float(np.abs(interior(g, raw[m]) - interior(g, w)).max())))
        assert not bits_equal(interior(g, raw[2]), interior(g, tr[2]))
    assert not bad, bad


# ---- the reference's whole dynamical core WITH ITS OWN MOM_open_boundary.F90 beside the oracle --------------------------------------------------------
def build_ref_dyn_obc_driver(tmp, rk2b=False):
    """tests/fortran/dyn_driver.F90 (-DREFERENCE_KERNELS -DREF_OBC) on the reference's own MOM_open_boundary.F90 (6116 lines, in place; under it the
    reference's MOM_interface_heights, remapping / regridding stack and MOM_array_transform; stand-ins for the grid type of the initialisation, file
    interpolation, tides and the obsolete-parameter checks) and, compiled against IT, the reference's MOM_dynamics_split_RK2.F90 with every module
    it steps through (continuity, CoriolisAdv, PressureForce, barotropic, set_viscosity, vert_friction, hor_visc)"""
    flags = ["-cpp", "-fdefault-real-8", "-O0", "-ffp-contract=off", "-DREFERENCE_KERNELS", "-DREF_EOS", "-DREF_PF_MONT", "-DREF_SET_VISC",
             "-DREF_INTERFACE_HEIGHTS", "-DREF_ALE", "-DREF_OBC"] + (["-DREF_RK2B"] if rk2b else []) + [
             f"-I{REF}/config_src/memory/dynamic_symmetric", f"-I{REF}/src/framework", f"-I{REF}/src/equation_of_state", f"-I{REF}/src/ALE",
             f"-I{STUBS}", f"-I{tmp}", "-J", str(tmp)]
    S, R = (lambda n: os.path.join(STUBS, n)), (lambda n: os.path.join(REF, n))
    srcs = [S("mom6_stubs.F90")] + [R(r) for r in ALE_SOURCES[:-1]] + [R("src/framework/MOM_array_transform.F90"), S("mom6_stubs_obc.F90"),
            R("src/core/MOM_open_boundary.F90"), R(ALE_SOURCES[-1]), S("mom6_stubs_after_obc.F90"), S("mom6_stubs_setvisc.F90"), R("src/framework/MOM_intrinsic_functions.F90"),
            R("src/parameterizations/vertical/MOM_set_viscosity.F90"), S("mom6_stubs_visc.F90")] + \
           [R(r) for r in VISC_SOURCES + CORE_SOURCES if r != "src/core/MOM_density_integrals.F90"] + \
           ([R("src/core/MOM_dynamics_split_RK2b.F90")] if rk2b else []) + [os.path.join(ROOT, "tests", "fortran", "dyn_driver.F90")]
    objs = []
    for src in srcs:
        o = os.path.join(str(tmp), os.path.basename(src)[:-4] + ".o")
        r = subprocess.run([FC, *flags, "-c", src, "-o", o], capture_output=True, text=True)
        assert r.returncode == 0, f"{src}:\n" + r.stderr[-3000:]
        objs.append(o)
    exe = os.path.join(str(tmp), "dyn_obc_ref_driver")
    r = subprocess.run([FC, *objs, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.fixture(scope="module")
def dyn_obc_exe(ref_builds):
    return ref_builds["dyn_obc_exe"].result()


TC3 = ["J=N,I=N:0,FLATHER,ORLANSKI", "J=0,I=0:N,FLATHER,ORLANSKI", "I=N,J=0:N,FLATHER,ORLANSKI", "I=0,J=N:0,FLATHER,ORLANSKI"]
OBC_DYN_CASES = {
    "tc3": (TC3, "synthetic", 3),                                              # the transcribed parameter set of .testing/tc3 on a random state
    "tc3_as_it_runs": (TC3, "config", 180),                                    # the disc of circle_obcs for DAYMAX = 6 h
    "mixed": (TC3[:2] + ["I=N,J=0:N,SIMPLE", "I=0,J=N:0,FLATHER"], "synthetic", 3),      # a specified and a Flather-only segment with external data
    "gradient": (["J=N,I=N:0,GRADIENT", "J=0,I=0:N,FLATHER,ORLANSKI", "I=N,J=0:N,FLATHER,GRADIENT", "I=0,J=N:0,SIMPLE"], "synthetic", 3),
    "inner": (["I=N,J=0:N,FLATHER,ORLANSKI", "J=5,I=N:0,SIMPLE"], "synthetic", 3),       # a segment inside the domain
    # the other vorticity / strain options of CorAdCalc and horizontal_viscosity at the segments (tc3 has the free-slip pair and OBC_ZERO_BIHARMONIC)
    "zero_vort_strain": (TC3, "synthetic", 3, dict(zero_vorticity=True, zero_strain=True, gamma_uv=0.3, rx_max=10.0)),
    "computed_vort_strain": (TC3, "synthetic", 3, dict(computed_vorticity=True, computed_strain=True, gamma_uv=0.3, rx_max=10.0)),
    "specified_vort": (TC3[:2] + ["I=N,J=0:N,SIMPLE", "I=0,J=N:0,FLATHER"], "synthetic", 3, dict(specified_vorticity=True, freeslip_strain=True, zero_biharmonic=True)),
}


@pytest.mark.parametrize("case", list(OBC_DYN_CASES))
def test_reference_dynamical_core_with_its_own_open_boundaries_equals_the_oracle(tmp_path, dyn_obc_exe, case, monkeypatch):
    """set_visc_init(OBC), initialize_dyn_split_RK2(OBC), then set_viscous_BBL and step_MOM_dyn_split_RK2 of the reference with its own
    MOM_open_boundary.F90 (radiation_open_bdry_conds, open_boundary_zero_normal_flow, the OBC branches of every operator, btstep's
    apply_velocity_OBCs / set_up_BT_OBC): the prognostic fields, the transports, eta_av, OBC%rx_normal / ry_normal and every segment's normal_vel
    equal the oracle's bit for bit -- for .testing/tc3 as it runs (its own initial condition, all 180 steps) and for sets with specified,
    Flather-only, gradient and interior segments carrying external data"""
    import test_testing_configs as tc
    segs, ic, nsteps = OBC_DYN_CASES[case][:3]
    monkeypatch.setattr(tc, "TC3_SEGMENTS", segs)
    if len(OBC_DYN_CASES[case]) > 3:
        monkeypatch.setattr(tc, "TC3_OBC", OBC_DYN_CASES[case][3])
    try:
        state, OBC = tc.tc3_case(ic=ic)
        g, d, taux, tauy, ustar, bbl, Rlay, g_prime = state
        rng = np.random.default_rng(11)
        for s in OBC.segment:      # external values of the specified and Flather segments (zero in tc3 itself)
            if case.startswith("tc3") or not s.on_pe:
                continue
            if s.specified:
                s.normal_vel[:] = 0.05 * rng.standard_normal(s.normal_vel.shape)
                s.normal_trans[:] = s.normal_vel * (3.0e4 * (5.0 + 50.0 * rng.random(s.normal_vel.shape)))
            if s.Flather:
                s.normal_vel_bt[:] = 0.02 * rng.standard_normal(s.normal_vel_bt.shape); s.SSH[:] = 0.05 * rng.standard_normal(s.SSH.shape)
        tc.write_case(tmp_path, "tc3", nsteps, False, state, bbl_mode=1)
        tc.write_obc_file(str(tmp_path / "obc.bin"), g, OBC)
        r = subprocess.run([dyn_obc_exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "params.txt"), str(tmp_path / "obc.bin")],
                           capture_output=True, text=True, preexec_fn=_unlimited_stack)
        assert r.returncode == 0 and "dyn_driver ok" in r.stdout, r.stderr[-3000:]
        st, calc, _ = tc.oracle_for("tc3", g, d, ustar, bbl, Rlay, g_prime, OBC=OBC)
        for n in range(nsteps):
            st.bbl(); st.step(taux, tauy, calc_dtbt=calc(n))
        got = tc.read_out(str(tmp_path / "out.bin"), g, OBC=OBC)
        want = dict(u=st.u, v=st.v, h=st.h, uh=st.uh, vh=st.vh, uhtr=st.uhtr, vhtr=st.vhtr, eta_av=st.eta_av)
        bad = [(n, float(np.abs(got[n] - want[n]).max())) for n, pos, nd in tc.OUT
               if n in want and not bits_equal(interior(g, got[n], pos), interior(g, want[n], pos))]
        if not bits_equal(interior(g, got["rx_normal"], _abi.POS_U), interior(g, OBC.rx_normal, _abi.POS_U)):
            bad.append("rx_normal")
        if not bits_equal(interior(g, got["ry_normal"], _abi.POS_V), interior(g, OBC.ry_normal, _abi.POS_V)):
            bad.append("ry_normal")
        bad += [f"normal_vel_{n + 1}" for n, s in enumerate(OBC.segment) if s.on_pe and not bits_equal(got[f"normal_vel_{n + 1}"], s.normal_vel)]
        assert not bad, bad
        assert np.abs(st.u[:, OBC.segnum_u != 0]).max() > 0      # (the boundaries are open)
        if any("ORLANSKI" in s for s in segs):
            assert max(np.abs(OBC.rx_normal).max(), np.abs(OBC.ry_normal).max()) > 0
    finally:
        tc.TC_INPUT.pop("tc3", None)


# ---- the reference's own radiation_open_bdry_conds, every form, beside the oracle --------------------------------------------------------------------
def build_ref_rad_driver(tmp):
    """tests/fortran/ref_rad_driver.F90 on the reference's own MOM_open_boundary.F90 (-DREF_OBC; the modules under it in place)"""
    flags = ["-cpp", "-fdefault-real-8", "-O0", "-ffp-contract=off", "-DREFERENCE_KERNELS", "-DREF_EOS", "-DREF_INTERFACE_HEIGHTS", "-DREF_ALE", "-DREF_OBC",
             f"-I{REF}/config_src/memory/dynamic_symmetric", f"-I{REF}/src/framework", f"-I{REF}/src/equation_of_state", f"-I{REF}/src/ALE",
             f"-I{STUBS}", f"-I{tmp}", "-J", str(tmp)]
    srcs = [os.path.join(STUBS, "mom6_stubs.F90")] + [os.path.join(REF, r) for r in ALE_SOURCES[:-1]] + \
           [os.path.join(REF, "src/framework/MOM_array_transform.F90"), os.path.join(STUBS, "mom6_stubs_obc.F90"),
            os.path.join(REF, "src/core/MOM_open_boundary.F90"), os.path.join(ROOT, "tests", "fortran", "ref_rad_driver.F90")]
    objs = []
    for src in srcs:
        o = os.path.join(str(tmp), os.path.basename(src)[:-4] + ".o")
        r = subprocess.run([FC, *flags, "-c", src, "-o", o], capture_output=True, text=True)
        assert r.returncode == 0, f"{src}:\n" + r.stderr[-3000:]
        objs.append(o)
    exe = os.path.join(str(tmp), "ref_rad_driver")
    r = subprocess.run([FC, *objs, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.fixture(scope="module")
def dyn_obc_rk2b_exe(ref_builds):
    return ref_builds["dyn_obc_rk2b_exe"].result()


@pytest.mark.parametrize("case", ["tc3", "mixed"])
def test_reference_rk2b_core_with_its_own_open_boundaries_equals_the_oracle(tmp_path, dyn_obc_rk2b_exe, case, monkeypatch):
    """SPLIT_RK2B with open boundaries: the reference's step_MOM_dyn_split_RK2b on its own MOM_open_boundary.F90 equals DynState(rk2b=True).step
    bit for bit (tc3's segments; a specified and a Flather-only segment with external data)"""
    import functools
    from oracle import orc
    monkeypatch.setattr(orc, "DynState", functools.partial(orc.DynState, rk2b=True))
    test_reference_dynamical_core_with_its_own_open_boundaries_equals_the_oracle(tmp_path, dyn_obc_rk2b_exe, case, monkeypatch)


@pytest.fixture(scope="module")
def rad_exe(ref_builds):
    return ref_builds["rad_exe"].result()


def _write_rad_case(path, g, d, OBC, gamma_uv, rx_max, dt, ncall, oblique):
    with open(path, "wb") as f:
        np.array([g.ni, g.nj, g.nk, g.halo, OBC.number_of_segments, int(oblique), ncall, 0], dtype="<i4").tofile(f)
        np.array([gamma_uv, rx_max, dt], dtype="<f8").tofile(f)
        for n in _abi.ALL_METRICS:
            np.ascontiguousarray(g.metrics[n], dtype="<f8").tofile(f)
        for k in ("u_new", "u_old", "v_new", "v_old", "rx", "ry"):
            np.ascontiguousarray(d[k], dtype="<f8").tofile(f)
        if oblique:
            for k in ("rx_oblique_u", "ry_oblique_u", "cff_normal_u", "rx_oblique_v", "ry_oblique_v", "cff_normal_v"):
                np.ascontiguousarray(getattr(OBC, k), dtype="<f8").tofile(f)
        OBC.segnum_u.astype("<i4").tofile(f); OBC.segnum_v.astype("<i4").tofile(f)
        for s in OBC.segment:
            np.array([s.direction, s.open, s.specified, s.on_pe, s.is_E_or_W, s.is_N_or_S] +
                     [s.HI.get(k, 0) for k in ("IsdB", "IedB", "JsdB", "JedB", "isd", "ied", "jsd", "jed")] +
                     [s.Flather, s.radiation, s.gradient, s.nudged, s.oblique, s.radiation_tan, s.radiation_grad, s.oblique_tan, s.oblique_grad,
                      int(s.nudged_tan) + 2 * int(s.nudged_grad)], dtype="<i4").tofile(f)
            np.array([s.Velocity_nudging_timescale_in, s.Velocity_nudging_timescale_out], dtype="<f8").tofile(f)
            if not s.on_pe:
                continue
            np.ascontiguousarray(s.normal_vel, dtype="<f8").tofile(f)
            if s.nudged:
                np.ascontiguousarray(s.nudged_normal_vel, dtype="<f8").tofile(f)
            if s.radiation_tan or s.nudged_tan or s.oblique_tan or s.radiation_grad or s.nudged_grad or s.oblique_grad:
                np.ascontiguousarray(s.tangential_vel, dtype="<f8").tofile(f); np.ascontiguousarray(s.tangential_grad, dtype="<f8").tofile(f)
            if s.nudged_tan:
                np.ascontiguousarray(s.nudged_tangential_vel, dtype="<f8").tofile(f)
            if s.nudged_grad:
                np.ascontiguousarray(s.nudged_tangential_grad, dtype="<f8").tofile(f)


@pytest.mark.parametrize("gamma_uv", [0.3, 1.0])
@pytest.mark.parametrize("form", ["normal", "tangential", "oblique", "oblique_tangential"])
def test_reference_radiation_open_bdry_conds_equals_the_oracle(tmp_path, rad_exe, form, gamma_uv, monkeypatch):
    """radiation_open_bdry_conds of the reference's own MOM_open_boundary.F90, two calls in a row (what it keeps between calls is carried): Orlanski
    radiation, the gradient condition and nudging of the normal component; ORLANSKI_TAN / _GRAD and NUDGED_TAN / _GRAD; OBLIQUE with nudging;
    OBLIQUE_TAN / _GRAD -- on segments of all four directions at the edges and inside the domain.  u_new, v_new, OBC%rx_normal / ry_normal, the
    oblique arrays, every segment's normal_vel, tangential_vel and tangential_grad equal the oracle's bit for bit.  (Round 4 had checked the
    tangential and oblique forms against the reference's four blocks written out as a table of their rows; this is the routine itself.)"""
    import copy
    import test_open_boundary as ob
    from oracle import orc
    if form == "oblique_tangential" and gamma_uv == 1.0:
        # with OBC_RAD_VEL_WT = 1 the reference forms the tangential rates from segment%grad_tan (:2487-2506), which allocate_OBC_segment_data
        # gives only to OBLIQUE_TAN segments (:3665): an OBLIQUE_GRAD segment without OBLIQUE_TAN reads an unallocated array there.  The
        # library and the oracle evaluate the gradients where they are used; here every such segment has both
        monkeypatch.setattr(ob, "OBLT_SEGS", [s if "OBLIQUE_TAN" in s else s.replace("OBLIQUE,", "OBLIQUE,OBLIQUE_TAN,") for s in ob.OBLT_SEGS])
    g, d, OBC = {"normal": ob.rad_case, "tangential": ob.tan_case, "oblique": ob.obl_case, "oblique_tangential": ob.oblt_case}[form]()
    oblique = form.startswith("oblique")
    rx_max, dt, ncall = 1.0, 900.0, 2
    OBC.gamma_uv, OBC.rx_max = gamma_uv, rx_max
    _write_rad_case(tmp_path / "in.bin", g, d, OBC, gamma_uv, rx_max, dt, ncall, oblique)
    r = subprocess.run([rad_exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, preexec_fn=_unlimited_stack)
    assert r.returncode == 0 and "ref_rad_driver ok" in r.stdout, r.stderr[-3000:]
    ref = copy.deepcopy(OBC)
    o = {k: v.copy() for k, v in d.items()}
    for n in range(ncall):
        orc.radiation_open_bdry_conds(g, ref, o["u_new"], o["u_old"], o["v_new"], o["v_old"], dt, gamma_uv=gamma_uv, rx_max=rx_max,
                                      rx_normal=o["rx"], ry_normal=o["ry"])
        if n < ncall - 1:
            o["u_old"][:] = o["u_new"]; o["v_old"][:] = o["v_new"]
            o["u_new"][:] = 0.5 * o["u_new"] + 0.01; o["v_new"][:] = 0.5 * o["v_new"] - 0.01
    want = [("u_new", o["u_new"]), ("v_new", o["v_new"]), ("rx_normal", o["rx"]), ("ry_normal", o["ry"])]
    if oblique:
        want += [(k, getattr(ref, k)) for k in ob.OBL_FIELDS]
    for n, s in enumerate(ref.segment):
        if s.on_pe:
            want.append((f"normal_vel_{n + 1}", s.normal_vel))
            if s.radiation_tan or s.nudged_tan or s.oblique_tan or s.radiation_grad or s.nudged_grad or s.oblique_grad:
                want += [(f"tangential_vel_{n + 1}", s.tangential_vel), (f"tangential_grad_{n + 1}", s.tangential_grad)]
    raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8")
    assert raw.size == sum(w.size for _, w in want), (raw.size, sum(w.size for _, w in want))
    bad = []
    for (name, w), a in zip(want, np.split(raw, np.cumsum([w.size for _, w in want])[:-1])):
        a = a.reshape(w.shape)
        if not bits_equal(a, w):
            bad.append((name, int((a != w).sum()), float(np.nanmax(np.abs(a - w)))))
    assert not bad, bad
    assert not bits_equal(o["u_new"], d["u_new"])


@pytest.mark.parametrize("InvL", [(0.0, 0.0), (1.0e-4, 3.0e-5), (0.0, 3.0e-5)])
def test_reference_update_segment_tracer_reservoirs_equals_the_oracle(tmp_path, rad_exe, InvL):
    """update_segment_tracer_reservoirs of the reference's own MOM_open_boundary.F90 (the backward-Euler blend of the reservoir, the value inside and
    the external value; zero and non-zero reservoir length scales, per-field factors on them, registry entries without a reservoir, land inside a
    segment): every reservoir equals the oracle's bit for bit"""
    import copy
    import test_advect_obc as ao
    from oracle import orc
    g, case, OBC = ao.reservoir_case(InvL)
    ref = copy.deepcopy(OBC)
    dt = 3600.0
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([g.ni, g.nj, g.nk, g.halo, OBC.number_of_segments, len(case["tr"]), 0, 0], dtype="<i4").tofile(f)
        np.array([g.H_subroundoff, dt], dtype="<f8").tofile(f)
        for n in ("mask2dT", "dyCu", "dxCv"):
            np.ascontiguousarray(g.metrics[n], dtype="<f8").tofile(f)
        for a in [case["uhtr"], case["vhtr"], case["h_end"]] + list(case["tr"]):
            np.ascontiguousarray(a, dtype="<f8").tofile(f)
        for s in OBC.segment:
            np.array([s.direction, s.on_pe, s.is_E_or_W, s.is_N_or_S] + [s.HI.get(k, 0) for k in ("IsdB", "IedB", "JsdB", "JedB", "isd", "ied", "jsd", "jed")],
                     dtype="<i4").tofile(f)
            np.array([s.Tr_InvLscale_in, s.Tr_InvLscale_out], dtype="<f8").tofile(f)
            if not s.on_pe:
                continue
            regs = s.tr_Reg or []
            np.array([len(regs)], dtype="<i4").tofile(f)
            for t in regs:
                has_l = "resrv_lfac_in" in t or "resrv_lfac_out" in t
                np.array([t["ntr_index"], int(t.get("tres") is not None), int(has_l)], dtype="<i4").tofile(f)
                np.array([t.get("resrv_lfac_in", 1.0), t.get("resrv_lfac_out", 1.0)], dtype="<f8").tofile(f)
                if t.get("tres") is not None:
                    np.ascontiguousarray(t["tres"], dtype="<f8").tofile(f); np.ascontiguousarray(t["t"], dtype="<f8").tofile(f)
    r = subprocess.run([rad_exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), "reservoirs"], capture_output=True, text=True, preexec_fn=_unlimited_stack)
    assert r.returncode == 0 and "ref_rad_driver ok" in r.stdout, r.stderr[-3000:]
    orc.update_segment_tracer_reservoirs(g, case["uhtr"], case["vhtr"], case["h_end"], ref, dt, case["tr"])
    want = [t["tres"] for s in ref.segment if s.on_pe for t in (s.tr_Reg or []) if t.get("tres") is not None]
    before = [t["tres"] for s in OBC.segment if s.on_pe for t in (s.tr_Reg or []) if t.get("tres") is not None]
    raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8")
    assert raw.size == sum(w.size for w in want)
    bad = [(n, int((a.reshape(w.shape) != w).sum()), float(np.abs(a.reshape(w.shape) - w).max()))
           for n, (w, a) in enumerate(zip(want, np.split(raw, np.cumsum([w.size for w in want])[:-1]))) if not bits_equal(a.reshape(w.shape), w)]
    assert not bad, bad
    assert any(not bits_equal(w, b) for w, b in zip(want, before))


# ---- the reference's own reproducing_sum (MOM_coms.F90) beside the oracle ---------------------------------------------------------------------------
def build_ref_coms_driver(tmp):
    """tests/fortran/ref_coms_driver.F90 on the reference's own src/framework/MOM_coms.F90, compiled in place against a one-PE stand-in of
    MOM_coms_infra (tests/fortran/stubs/mom6_stubs_coms_infra.F90; the rest of the stand-ins are not in this build)"""
    flags = ["-cpp", "-fdefault-real-8", "-O0", "-ffp-contract=off", f"-I{REF}/config_src/memory/dynamic_symmetric", f"-I{REF}/src/framework",
             f"-I{tmp}", "-J", str(tmp)]
    objs = []
    for src in [os.path.join(STUBS, "mom6_stubs_coms_infra.F90"), os.path.join(REF, "src/framework/MOM_coms.F90"),
                os.path.join(ROOT, "tests", "fortran", "ref_coms_driver.F90")]:
        o = os.path.join(str(tmp), os.path.basename(src)[:-4] + ".o")
        r = subprocess.run([FC, *flags, "-c", src, "-o", o], capture_output=True, text=True)
        assert r.returncode == 0, f"{src}:\n" + r.stderr[-3000:]
        objs.append(o)
    exe = os.path.join(str(tmp), "ref_coms_driver")
    r = subprocess.run([FC, *objs, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.fixture(scope="module")
def coms_exe(ref_builds):
    return ref_builds["coms_exe"].result()


@pytest.mark.parametrize("kind", ["unit", "wide", "cancelling", "tiny"])
def test_reference_reproducing_sum_equals_the_oracle(tmp_path, coms_exe, kind):
    """reproducing_sum_3d / _2d of the reference's own MOM_coms.F90 (the extended-fixed-point sum of Hallberg & Adcroft 2014) over the h-point
    computational domain: the total, the sums by layer and EFP_to_real of the extended-fixed-point results equal the oracle's bit for bit -- on fields of order one, over twenty decades, with near-total cancellation, and near the smallest representable increment"""
    from mom6_amd import synth
    from oracle import orc
    g = synth.make_grid(37, 23, 5, halo=4, land_frac=0.2, seed=3)
    rng = np.random.default_rng({"unit": 1, "wide": 2, "cancelling": 3, "tiny": 4}[kind])
    shp = g.shape3(_abi.POS_H)
    a = rng.standard_normal(shp)
    if kind == "wide":
        a = a * 10.0 ** rng.uniform(-8, 12, shp)
    elif kind == "cancelling":
        a = a * 1.0e9; a[:, :, 1::2] = -a[:, :, 0:-1:2][:, :, :a[:, :, 1::2].shape[2]] * (1.0 + 1.0e-13)
    elif kind == "tiny":
        a = a * 1.0e-38
    a = np.ascontiguousarray(a)
    i0 = g.halo
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([shp[2], shp[1], shp[0], i0 + 1, i0 + g.ni, i0 + 1, i0 + g.nj, 0], dtype="<i4").tofile(f)
        a.astype("<f8").tofile(f)
    r = subprocess.run([coms_exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0 and "ref_coms_driver ok" in r.stdout, r.stderr[-2000:]
    raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8")
    nk = shp[0]
    s3, s2, tot_r, diff = raw[:4]; sums, lay_r = raw[4:4 + nk], raw[4 + nk:4 + 2 * nk]
    o3 = orc.reproducing_sum(g, a, _abi.POS_H, by_layer=True)
    o2 = orc.reproducing_sum(g, a[0], _abi.POS_H)
    bits = lambda x: np.float64(x).view(np.uint64)
    assert bits(s3) == bits(o3["sum"]) and bits(tot_r) == bits(o3["sum"]), (s3, o3["sum"])
    assert bits(s2) == bits(o2["sum"]), (s2, o2["sum"])
    assert all(bits(x) == bits(y) for x, y in zip(sums, o3["sums"])) and all(bits(x) == bits(y) for x, y in zip(lay_r, o3["sums"]))
    # EFP_real_diff(total, first layer): the reference's own extended-fixed-point subtraction; the oracle has no such entry, so this one value is
    # checked to rounding against the sum of the other layers (not bitwise)
    assert np.isclose(diff, sum(o3["sums"][1:]), rtol=1e-12, atol=abs(s3) * 1e-15 + 1e-300)
    assert s3 != 0.0
