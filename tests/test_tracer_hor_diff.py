"""tracer_hordiff (src/tracer/MOM_tracer_hor_diff.F90:119), the along-layer diffusion with a constant KHTR: the oracle against
what the scheme guarantees on the CPU (the reference holds no known-answer vectors for it), the library against the oracle
on the GPU, bit for bit."""
import numpy as np
import pytest

from helpers import bits_equal, interior
from mom6_amd import _abi, synth
from oracle import orc


def case(ni=30, nj=22, nk=4, seed=2, reentrant=(True, False), land_frac=0.2, ntr=3):
    g = synth.make_grid(ni, nj, nk, land_frac=land_frac, seed=seed + 500, reentrant_x=reentrant[0], reentrant_y=reentrant[1])
    d = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=seed, umax=0.1, eta_amp=0.2).items()}
    rng = np.random.default_rng(seed)
    mT = g.mask2dT[None]
    tr = [np.ascontiguousarray(d["T"]), np.ascontiguousarray(d["S"]),
          np.ascontiguousarray((rng.random(d["h"].shape) > 0.7) * 1.0 * mT)][:ntr]
    for t in tr:
        orc.halo_update(g, t, _abi.POS_H)
    return g, d["h"], tr


def inventory(g, h, t):
    return float((interior(g, h) * interior(g, g.areaT)[None] * interior(g, t)).sum())


@pytest.mark.parametrize("KhTr,check", [(50.0, False), (5.0e7, True)])
def test_oracle_conserves_preserves_constants_and_bounds(KhTr, check):
    g, h, tr = case()
    const = np.full_like(tr[0], 7.25)
    tr = [t.copy() for t in tr] + [const]
    before = [t.copy() for t in tr]
    st = orc.tracer_hordiff(g, h, 3600.0, tr, KhTr, check_diffusive_CFL=check)
    if check:
        assert st.num_itts > 1 and st.halo_updates == st.num_itts and st.max_CFL > 1.0     # the big diffusivity needs iterations
    else:
        assert st.num_itts == 1
    hh = h + g.H_subroundoff
    for t0, t1 in zip(before, tr):
        # flux form over (h + h_neglect) * area: the inventory is conserved to roundoff (land cells have zero face coefficients
        # only through vanishing h, so the sum runs over every cell)
        a, b = inventory(g, hh, t0), inventory(g, hh, t1)
        assert abs(a - b) <= 1e-11 * max(1.0, abs(a)), (a, b)
        # the maximum principle under the CFL limit the iteration count enforces
        assert interior(g, t1).max() <= interior(g, t0).max() + 1e-12 and interior(g, t1).min() >= interior(g, t0).min() - 1e-12
    assert np.array_equal(interior(g, tr[-1]), interior(g, before[-1]))      # a constant stays, to the bit
    assert not np.array_equal(interior(g, tr[0]), interior(g, before[0]))


def test_oracle_limit_and_early_return():
    g, h, tr = case(ntr=1)
    t0 = tr[0].copy()
    st = orc.tracer_hordiff(g, h, 3600.0, tr, 0.0)
    assert st.num_itts == 0 and np.array_equal(tr[0], t0)                      # KHTR <= 0: returns at once (:199)
    a = [t0.copy()]; b = [t0.copy()]
    sa = orc.tracer_hordiff(g, h, 3600.0, a, 1.0e9, max_diff_CFL=0.5)
    sb = orc.tracer_hordiff(g, h, 3600.0, b, 1.0e9, max_diff_CFL=2.5)
    assert sa.num_itts == 1 and sb.num_itts == 3                              # ceiling(MAX_TR_DIFFUSION_CFL) iterations (:424-426)
    assert np.isfinite(a[0]).all() and np.isfinite(b[0]).all()
    c = [t0.copy()]
    orc.tracer_hordiff(g, h, 3600.0, c, 50.0, conc_underflow=[1.0e30])
    assert np.all(interior(g, c[0]) == 0.0)                                    # everything below the underflow is zeroed (:607-612)


HD_CASES = [dict(KhTr=50.0), dict(KhTr=5.0e7, check_diffusive_CFL=True), dict(KhTr=1.0e9, max_diff_CFL=2.5),
            dict(KhTr=300.0, conc_underflow=[0.0, 1.0e-3, 0.5]), dict(KhTr=50.0, reentrant=(True, True)),
            dict(KhTr=50.0, reentrant=(False, False), ni=70, nj=9, nk=2)]


@pytest.mark.gpu
@pytest.mark.parametrize("kw", HD_CASES, ids=[",".join(f"{k}={v}" for k, v in c.items()) for c in HD_CASES])
@pytest.mark.parametrize("space", ["device", "host"])
def test_tracer_hordiff_matches_oracle_bitwise(kw, space):
    import torch
    from mom6_amd.tracer_advect import DeviceGrid
    from mom6_amd.tracer_hor_diff import tracer_hor_diff_init, tracer_hordiff
    kw = dict(kw)
    gk = {k: kw.pop(k) for k in ("reentrant", "ni", "nj", "nk") if k in kw}
    cu = kw.pop("conc_underflow", None)
    g, h, tr = case(**gk)
    ref = [t.copy() for t in tr]
    rs = orc.tracer_hordiff(g, h, 3600.0, ref, kw["KhTr"], max_diff_CFL=kw.get("max_diff_CFL", -1.0),
                            check_diffusive_CFL=kw.get("check_diffusive_CFL", False), conc_underflow=cu)
    dg = DeviceGrid(g)
    put = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if space == "device" else (lambda a: np.ascontiguousarray(a).copy())
    dtr = [put(t) for t in tr]
    CS = tracer_hor_diff_init(KHTR=kw["KhTr"], MAX_TR_DIFFUSION_CFL=kw.get("max_diff_CFL", -1.0),
                              CHECK_DIFFUSIVE_CFL=kw.get("check_diffusive_CFL", False))
    st = tracer_hordiff(put(h), 3600.0, None, None, None, dg, CS, dtr, conc_underflow=cu)
    dg.sync()
    assert (st.num_itts, st.halo_updates) == (rs.num_itts, rs.halo_updates) and st.max_CFL == rs.max_CFL
    for m, (a, b) in enumerate(zip(dtr, ref)):
        an = a.cpu().numpy() if space == "device" else a
        assert bits_equal(interior(g, an), interior(g, b)), m
    dg.close()


@pytest.mark.gpu
def test_tracer_hordiff_refuses_what_it_does_not_provide():
    import torch
    from mom6_amd._lib import Mom6HipError
    from mom6_amd.tracer_advect import DeviceGrid
    from mom6_amd.tracer_hor_diff import tracer_hor_diff_init, tracer_hordiff
    g, h, tr = case(ntr=1)
    dg = DeviceGrid(g)
    dh = torch.from_numpy(h).cuda(); dt_ = [torch.from_numpy(tr[0]).cuda()]
    with pytest.raises(Mom6HipError, match="USE_NEUTRAL_DIFFUSION"):
        tracer_hordiff(dh, 3600.0, None, None, None, dg, tracer_hor_diff_init(KHTR=50.0, USE_NEUTRAL_DIFFUSION=True), dt_)
    with pytest.raises(Mom6HipError, match="L2u"):      # KHTR_SLOPE_CFF > 0 without the Visbeck fields
        tracer_hordiff(dh, 3600.0, None, {}, None, dg, tracer_hor_diff_init(KHTR=50.0, KHTR_SLOPE_CFF=0.1), dt_)
    with pytest.raises(Mom6HipError, match="only"):
        tracer_hordiff(dh, 3600.0, None, dict(ebt_struct=dh), None, dg, tracer_hor_diff_init(KHTR=50.0), dt_)
    dg.close()


# ---- the VarMix / MEKE diffusivities (:236-281; .testing/tc1, tc2) ----------------------------------------------------------------
def varmix_fields(g, seed=4):
    rng = np.random.default_rng(seed)
    sh, su, sv = g.shape2(_abi.POS_H), g.shape2(_abi.POS_U), g.shape2(_abi.POS_V)
    f = dict(L2u=1.0e9 * rng.random(su), L2v=1.0e9 * rng.random(sv), SN_u=1.0e-6 * rng.random(su), SN_v=1.0e-6 * rng.random(sv))
    for n in f:      # (one value per physical face: the western and the eastern edge of a re-entrant domain are the same faces,
        f[n] = np.ascontiguousarray(f[n])      # both inside the symmetric compute range, so no halo update makes them agree)
        if n.endswith("u") and g.reentrant_x:
            f[n][:, g.halo] = f[n][:, g.halo + g.ni]
        if n.endswith("v") and g.reentrant_y:
            f[n][g.halo, :] = f[n][g.halo + g.nj, :]
        orc.halo_update(g, f[n], _abi.POS_U if n.endswith("u") else _abi.POS_V)
    for n, a in (("Res_fn_h", rng.random(sh)), ("Rd_dx_h", 2.0 * rng.random(sh)), ("Kh", 900.0 * rng.random(sh) * g.mask2dT)):
        a = np.ascontiguousarray(a); orc.halo_update(g, a, _abi.POS_H); f[n] = a
    return f


VM = {
    "tc1": dict(KhTr=1.0, KhTr_Slope_Cff=0.1, use=("L2u", "L2v", "SN_u", "SN_v", "Res_fn_h")),            # Visbeck + RESOLN_SCALED_KHTR
    "tc2": dict(KhTr=1.0, meke=0.5, use=()),                                                              # MEKE_KHTR_FAC = 0.5
    "bounds": dict(KhTr=10.0, KhTr_Slope_Cff=0.3, KhTr_min=40.0, KhTr_max=600.0, meke=1.0, use=("L2u", "L2v", "SN_u", "SN_v"), check=True),
    "passivity": dict(KhTr=100.0, KhTr_passivity_coeff=3.0, KhTr_passivity_min=0.5, KhTr_max=800.0, use=("Rd_dx_h", "Res_fn_h"), max_diff_CFL=1.5),
    "varmix_only": dict(KhTr=0.0, KhTr_Slope_Cff=0.2, use=("L2u", "L2v", "SN_u", "SN_v")),                 # KHTR = 0 still diffuses with VarMix (:197)
}


def run_vm(g, h, tr, name, dt=3600.0):
    kw = dict(VM[name]); use = kw.pop("use"); meke = kw.pop("meke", None); check = kw.pop("check", False)
    f = varmix_fields(g)
    tr = [t.copy() for t in tr]
    st = orc.tracer_hordiff(g, h, dt, tr, kw.pop("KhTr"), check_diffusive_CFL=check, VarMix={n: f[n] for n in use},
                            MEKE=None if meke is None else dict(Kh=f["Kh"], KhTr_fac=meke), **kw)
    return tr, st, (f, use, meke, check)


@pytest.mark.parametrize("name", list(VM))
def test_oracle_varmix_diffusion_conserves_and_is_bounded(name):
    g, h, tr = case(40, 26, 4)
    out, st, _ = run_vm(g, h, tr + [np.full_like(tr[0], -3.5)], name)
    hh = h + g.H_subroundoff
    for t0, t1 in zip(tr, out):
        a, b = inventory(g, hh, t0), inventory(g, hh, t1)
        assert abs(a - b) <= 1e-11 * max(1.0, abs(a)), (a, b)
    assert np.array_equal(interior(g, out[-1]), np.full_like(interior(g, out[-1]), -3.5))
    assert not np.array_equal(interior(g, out[0]), interior(g, tr[0]))


def test_oracle_varmix_with_nothing_to_add_is_the_constant_diffusivity():
    """use_variable_mixing with no slope coefficient, no MEKE, no scaling: Kh = max(KHTR, KHTR_MIN) at every face -- the bits of the
    constant-KHTR branch (:305-328); with Res_fn_h = 1/2 everywhere the bits of half the diffusivity"""
    g, h, tr = case()
    a = [t.copy() for t in tr]; b = [t.copy() for t in tr]; c = [t.copy() for t in tr]; d = [t.copy() for t in tr]
    orc.tracer_hordiff(g, h, 3600.0, a, 300.0)
    orc.tracer_hordiff(g, h, 3600.0, b, 300.0, VarMix={})
    for x, y in zip(a, b):
        assert bits_equal(x, y)
    orc.tracer_hordiff(g, h, 3600.0, c, 150.0)
    orc.tracer_hordiff(g, h, 3600.0, d, 300.0, VarMix=dict(Res_fn_h=np.full(g.shape2(_abi.POS_H), 0.5)))
    for x, y in zip(c, d):
        assert bits_equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(VM))
def test_varmix_diffusivities_match_oracle(name):
    """tracer_hordiff with VarMix / MEKE: library == oracle bit for bit, device and staged host arrays, three grids"""
    import torch
    from mom6_amd.tracer_advect import DeviceGrid
    from mom6_amd.tracer_hor_diff import tracer_hor_diff_init, tracer_hordiff
    REF = dict(KhTr="KHTR", KhTr_Slope_Cff="KHTR_SLOPE_CFF", KhTr_min="KHTR_MIN", KhTr_max="KHTR_MAX", KhTr_passivity_coeff="KHTR_PASSIVITY_COEFF",
               KhTr_passivity_min="KHTR_PASSIVITY_MIN", max_diff_CFL="MAX_TR_DIFFUSION_CFL")
    for (ni, nj, nk, topo) in [(40, 26, 4, (True, False)), (33, 20, 3, (True, True)), (130, 9, 12, (False, False))]:
        g, h, tr = case(ni, nj, nk, reentrant=topo)
        ref, rst, (f, use, meke, check) = run_vm(g, h, tr, name)
        dg = DeviceGrid(g)
        for resident in (True, False):
            X = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if resident else (lambda a: np.ascontiguousarray(a).copy())
            N = (lambda a: a.cpu().numpy()) if resident else (lambda a: a)
            kw = {REF[k]: v for k, v in VM[name].items() if k in REF}
            CS = tracer_hor_diff_init(CHECK_DIFFUSIVE_CFL=check, **kw)
            dtr = [X(t) for t in tr]
            st = tracer_hordiff(X(h), 3600.0, None if meke is None else dict(Kh=X(f["Kh"]), KhTr_fac=meke), {n: X(f[n]) for n in use}, None, dg, CS, dtr)
            dg.sync()
            assert st.num_itts == rst.num_itts and st.max_CFL == rst.max_CFL
            for m, (a, b) in enumerate(zip(dtr, ref)):
                assert bits_equal(interior(g, N(a)), interior(g, b)), (name, (ni, nj, nk), resident, m)
        dg.close()


# ---- the two tracer module shims (MOM_tracer_advect_hip.F90, MOM_tracer_hor_diff_hip.F90) with the reference's argument lists ---------
def _write_tracer_case(tmp, g, h, tr, name, dt=3600.0, scheme="PPM:H3", resident=False):
    """the input and parameter files of tests/fortran/tracer_driver.F90 for one of VM (or None: constant KHTR), and the oracle's tracers
    after advect_tracer + tracer_hordiff"""
    from mom6_amd import synth as sy
    ad = {k: (v if isinstance(v, list) else v.numpy()) for k, v in sy.make_advection_state(g, ntr=1, seed=9).items()}
    f = varmix_fields(g)
    neutral = name in ("neutral", "neutral_interior")      # USE_NEUTRAL_DIFFUSION with the first two tracers as tv%T, tv%S
    h_ML = None
    if name == "neutral_interior":      # NDIFF_INTERIOR_ONLY below visc%h_ML
        h_ML = np.ascontiguousarray(np.clip(0.8 * np.random.default_rng(3).random(g.shape2(_abi.POS_H)) - 0.1, 0.0, 1.0) * ad["h_end"].sum(0))
    kw = dict(VM[name]) if name and not neutral else dict(KhTr=300.0, use=None)
    use = kw.pop("use"); meke = kw.pop("meke", None); check = kw.pop("check", False)
    ref = [t.copy() for t in tr]
    orc.advect_tracer(g, ad["h_end"], ad["uhtr"], ad["vhtr"], dt, 900.0, scheme, ref)
    for t in ref:
        orc.halo_update(g, t, _abi.POS_H)
    KhTr = kw.pop("KhTr")
    orc.tracer_hordiff(g, ad["h_end"], dt, ref, KhTr, check_diffusive_CFL=check, VarMix=None if use is None else {n: f[n] for n in use},
                       MEKE=None if meke is None else dict(Kh=f["Kh"], KhTr_fac=meke),
                       neutral=dict(eos=orc.eos("WRIGHT"), idx_T=0, idx_S=1, ndiff_answer_date=20240401, H_to_RZ=1035.0, h_ML=h_ML) if neutral else None, **kw)
    opt = [len(tr), int(use is not None), int(use is not None and "Res_fn_h" in use), int(meke is not None), int(h_ML is not None), 0, 0, 0]
    with open(tmp / "in.bin", "wb") as fh:
        np.array([g.ni, g.nj, g.nk, g.halo, int(g.reentrant_x), int(g.reentrant_y), g.first_direction, 0], dtype="<i4").tofile(fh)
        np.array([g.Angstrom_H, g.H_subroundoff, g.dZ_subroundoff, g.H_to_Z, g.Z_to_H, g.g_Earth, g.Rho0, 900.0], dtype="<f8").tofile(fh)
        np.array(opt, dtype="<i4").tofile(fh)
        for n in _abi.ALL_METRICS:
            np.ascontiguousarray(g.metrics[n], dtype="<f8").tofile(fh)
        np.array([dt, 1.0 if meke is None else meke], dtype="<f8").tofile(fh)
        for a in [ad["h_end"], ad["uhtr"], ad["vhtr"]] + tr + [f[n] for n in ("Kh", "L2u", "L2v", "SN_u", "SN_v", "Res_fn_h", "Rd_dx_h")] + \
                ([h_ML] if h_ML is not None else []):
            np.ascontiguousarray(a, dtype="<f8").tofile(fh)
    REF = dict(KhTr_Slope_Cff="KHTR_SLOPE_CFF", KhTr_min="KHTR_MIN", KhTr_max="KHTR_MAX", KhTr_passivity_coeff="KHTR_PASSIVITY_COEFF",
               KhTr_passivity_min="KHTR_PASSIVITY_MIN", max_diff_CFL="MAX_TR_DIFFUSION_CFL")
    with open(tmp / "params.txt", "w") as fh:
        fh.write(f"TRACER_ADVECTION_SCHEME = {scheme}\nDT = 900.0\nKHTR = {KhTr!r}\nCHECK_DIFFUSIVE_CFL = {check}\nGPU_RESIDENT_DYNAMICS = {resident}\n")
        for k, v in kw.items():
            fh.write(f"{REF[k]} = {float(v)!r}\n")
        if neutral:
            fh.write("USE_NEUTRAL_DIFFUSION = True\nNDIFF_ANSWER_DATE = 20240401\nEQN_OF_STATE = WRIGHT\n")
            fh.write(f"NDIFF_INTERIOR_ONLY = {h_ML is not None}\n")
    return ref


def test_tracer_module_shims_compile_and_fail_loudly_without_gpu(tmp_path):
    import os, subprocess, torch
    from test_fortran_abi import FC, _build_shims
    if not os.path.exists(FC):
        pytest.skip("amdflang not present")
    exe = _build_shims(tmp_path, driver="tracer_driver")
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    g, h, tr = case(24, 16, 3)
    _write_tracer_case(tmp_path, g, h, tr, "tc1")
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "params.txt")], capture_output=True, text=True)
    assert r.returncode != 0 and "FATAL" in r.stderr and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_tracer_module_shims_match_oracle(tmp_path):
    """tracer_advect_init / advect_tracer and tracer_hor_diff_init / tracer_hordiff called from Fortran with the reference's argument lists
    (a tracer registry, MEKE, VarMix) on host arrays, constant KHTR and every variable-mixing set: the oracle's bits"""
    import os, subprocess
    from test_fortran_abi import FC, _build_shims
    if not os.path.exists(FC):
        pytest.skip("amdflang not present")
    exe = _build_shims(tmp_path, driver="tracer_driver")
    g, h, tr = case(36, 22, 4)
    for name, resident in [(n, r) for n in [None, "neutral", "neutral_interior"] + list(VM) for r in (False, True)]:      # staged host arrays, or the shared device mirrors
        ref = _write_tracer_case(tmp_path, g, h, tr, name, resident=resident)
        r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "params.txt")], capture_output=True, text=True)
        assert r.returncode == 0 and "tracer_driver ok" in r.stdout, (name, r.stderr[-600:])
        raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8").reshape((len(tr),) + tr[0].shape)
        for m, w in enumerate(ref):
            assert bits_equal(interior(g, raw[m]), interior(g, w)), (name, m)
