"""step_MOM_dyn_split_RK2: the oracle's step (oracle/dyn_split_rk2.c, the reference's order of calls with zero
viscosities) against invariants on the CPU, and the library's step against the oracle on the GPU, bit for bit,
over several steps."""
import numpy as np
import pytest

from helpers import bits_equal, interior
from mom6_amd import _abi, synth
from oracle import orc


def make_case(ni=20, nj=16, nk=3, seed=4, reentrant_x=True, reentrant_y=False, land_frac=0.2, umax=0.1, rest=False):
    g = synth.make_grid(ni, nj, nk, land_frac=land_frac, seed=seed + 300, reentrant_x=reentrant_x, reentrant_y=reentrant_y)
    d = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=seed, umax=0.0 if rest else umax, eta_amp=0.2).items()}
    if rest:
        tot = d["h"].sum(0)
        d["h"] = np.ascontiguousarray(d["h"] * np.where(tot > 0, g.bathyT * g.Z_to_H / np.maximum(tot, 1e-30), 1.0)[None])
        d["T"][:] = 10.0; d["S"][:] = 35.0
    yy = np.linspace(0.0, np.pi, g.shape2(_abi.POS_U)[0])
    taux = np.ascontiguousarray((0.0 if rest else 0.1) * np.cos(2 * yy)[:, None] * g.mask2dCu)
    tauy = np.ascontiguousarray(0.0 * g.mask2dCv)
    return g, d, taux, tauy


def volume(g, h):
    return float((interior(g, h) * interior(g, g.areaT)[None] * interior(g, g.mask2dT)[None]).sum())


@pytest.mark.parametrize("use_bt_cont", [True, False])
def test_oracle_step_conserves_volume_and_keeps_eta_consistent(use_bt_cont):
    g, d, taux, tauy = make_case()
    dt = 1800.0
    st = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], dt, use_bt_cont=use_bt_cont)
    st.bcs.dtbt = dt / 12.6
    v0 = volume(g, st.h)
    for n in range(4):
        st.step(taux, tauy)
        assert np.all(np.isfinite(st.u)) and np.all(np.isfinite(st.h)) and st.h.min() > 0
    # continuity is flux-form: the ocean's volume is conserved to roundoff
    assert abs(volume(g, st.h) - v0) <= 1e-12 * v0
    # the barotropic free surface tracks the layer sum (bt_mass_source feeds the difference back)
    eta_h = interior(g, st.h.sum(0) - g.bathyT * g.Z_to_H)
    err = np.abs(interior(g, st.arrs["eta"]) - eta_h)[interior(g, g.mask2dT) > 0]
    assert err.max() < 1e-6
    # accumulated transports are the time integral of uh
    assert np.abs(st.uhtr).max() > 0 and st.cs.CAu_pred_stored == 1
    assert np.abs(st.u).max() < 3.0


def test_oracle_ocean_at_rest_stays_at_rest():
    g, d, taux, tauy = make_case(rest=True, land_frac=0.0)
    st = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], 1800.0)
    st.bcs.dtbt = 1800.0 / 10.6
    for n in range(3):
        st.step(taux, tauy)
    # a homogeneous (T, S) ocean under terrain-following layers: the compressible EOS leaves a truncation-level
    # pressure force (analytically zero); nothing else may move
    assert np.abs(st.u).max() < 1e-3 and np.abs(st.v).max() < 1e-3
    assert np.abs(interior(g, st.arrs["eta"])).max() < 1e-3


# ---- the surface pressures of the step (MOM_dynamics_split_RK2.F90:435-442, :495-503) ----
def _pressures(g, seed=12):
    """an atmospheric / ice load: a smooth high and noise, [Pa]"""
    rng = np.random.default_rng(seed)
    sh = g.shape2(_abi.POS_H)
    yy, xx = np.meshgrid(np.linspace(0, np.pi, sh[0]), np.linspace(0, 2 * np.pi, sh[1]), indexing="ij")
    p_end = np.ascontiguousarray(1.0e5 + 800.0 * np.sin(xx) * np.sin(yy) + 20.0 * rng.standard_normal(sh))
    p_begin = np.ascontiguousarray(p_end - 300.0 * np.cos(xx) * np.sin(yy))
    for a in (p_begin, p_end):
        orc.halo_update(g, a, _abi.POS_H)
    return p_begin, p_end


def test_oracle_step_with_a_uniform_surface_pressure_is_the_step_without_it_and_a_load_pushes_the_water_away():
    g, d, taux, tauy = make_case(land_frac=0.0)
    dt = 1800.0
    def run(**kw):
        st = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], dt)
        st.bcs.dtbt = dt / 9.6
        for n in range(2):
            st.step(taux, tauy, **kw)
        return st
    a = run()
    # forces%p_surf alone is PressureForce's p_atm (:441); the same field as p_surf_begin = p_surf_end gives the same pressure force and an
    # eta_PF_start equal to eta_PF (:501: the difference of the two pressures is zero), i.e. d_eta_PF = 0 in btstep: the same bits
    p_begin, p_end = _pressures(g)
    b = run(p_surf=p_end)
    c = run(p_surf_begin=p_end, p_surf_end=p_end)
    assert bits_equal(b.u, c.u) and bits_equal(b.h, c.h) and bits_equal(b.eta_av, c.eta_av)
    assert not bits_equal(a.u, b.u)
    # a pressure that changes over the step: the barotropic solver sees the force grow from eta_PF_start to eta_PF
    e = run(p_surf_begin=p_begin, p_surf_end=p_end)
    assert not bits_equal(e.u, b.u) and np.all(np.isfinite(e.u)) and np.abs(e.u).max() < 3.0
    # p_surf_begin without p_surf_end is not dyn_p_surf (:435): forces%p_surf (absent here) decides
    f = run(p_surf_begin=p_begin)
    assert bits_equal(f.u, a.u)
    # the inverse barometer: the water moves from under the high (divergent transport where the load sits)
    st = orc.DynState(g, 0 * d["u"], 0 * d["v"], d["h"], d["T"], d["S"], dt)
    st.bcs.dtbt = dt / 9.6
    blob = np.zeros(g.shape2(_abi.POS_H)); jj, ii = blob.shape[0] // 2, blob.shape[1] // 2
    blob[jj - 1:jj + 2, ii - 1:ii + 2] = 2.0e3
    orc.halo_update(g, blob, _abi.POS_H)
    st0 = orc.DynState(g, 0 * d["u"], 0 * d["v"], d["h"], d["T"], d["S"], dt); st0.bcs.dtbt = dt / 9.6
    st.step(0 * taux, tauy, p_surf=blob); st0.step(0 * taux, tauy)
    assert (st.arrs["eta"] - st0.arrs["eta"])[jj, ii] < -1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["forces", "dyn", "dyn_rk2b", "forces_vertvisc"])
def test_step_with_surface_pressures_matches_oracle_bitwise(mode):
    import torch
    from mom6_amd.dynamics_split_rk2 import (initialize_dyn_split_RK2, initialize_dyn_split_RK2b, step_MOM_dyn_split_RK2,
                                             step_MOM_dyn_split_RK2b)
    from mom6_amd.tracer_advect import DeviceGrid
    from mom6_amd.vert_friction import vertvisc_type
    g, d, taux, tauy = make_case(ni=26, nj=18, nk=4, seed=6)
    dt = 1800.0
    rk2b = mode == "dyn_rk2b"
    vv = mode == "forces_vertvisc"
    va = _visc_arrays(g)
    ref = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], dt, rk2b=rk2b,
                       **(dict(vertvisc=orc.vertvisc_cs(g, Kv=1.0e-3, Hbbl=10.0), visc=orc.vertvisc_type(**va)) if vv else {}))
    ref.bcs.dtbt = dt / 9.6
    dg = DeviceGrid(g)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    u, v, h, Tt, Ss = (T(d[k]) for k in ("u", "v", "h", "T", "S"))
    Z = lambda pos, k3=True: torch.zeros(g.shape3(pos) if k3 else g.shape2(pos), dtype=torch.float64, device="cuda")
    uh, vh, uhtr, vhtr, eta_av = Z(_abi.POS_U), Z(_abi.POS_V), Z(_abi.POS_U), Z(_abi.POS_V), Z(_abi.POS_H, False)
    CS = (initialize_dyn_split_RK2b if rk2b else initialize_dyn_split_RK2)(
        u, v, h, uh, vh, dt, dg, coriolis=dict(bound_coriolis=True), **(dict(vertvisc=dict(KV=1.0e-3, HBBL=10.0)) if vv else {}))
    CS.barotropic_CSp.st.dtbt = ref.bcs.dtbt
    visc = vertvisc_type(**{n: T(a) for n, a in va.items()}) if vv else None
    p_begin, p_end = _pressures(g)
    tx, ty = T(taux), T(tauy)
    step = step_MOM_dyn_split_RK2b if rk2b else step_MOM_dyn_split_RK2
    for n in range(3):
        pb = p_begin + 10.0 * n; pe = p_end + 10.0 * n
        if mode.startswith("forces"):
            ref.step(taux, tauy, p_surf=pe)
            step(u, v, h, (Tt, Ss), visc, None, dt, (tx, ty, T(pe)), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS)
        else:
            ref.step(taux, tauy, p_surf_begin=pb, p_surf_end=pe)
            step(u, v, h, (Tt, Ss), visc, None, dt, (tx, ty), T(pb), T(pe), uh, vh, uhtr, vhtr, eta_av, dg, CS)
        dg.sync()
        for name, a, b in (("u", u, ref.u), ("v", v, ref.v), ("h", h, ref.h), ("uh", uh, ref.uh), ("eta_av", eta_av, ref.eta_av),
                           ("eta", CS.eta, ref.arrs["eta"])):
            an = a.cpu().numpy()
            assert bits_equal(an, b), (mode, n, name, float(np.abs(an - b).max()))
    dg.close()


RK2_CASES = [dict(), dict(use_bt_cont=False), dict(reentrant_x=False), dict(reentrant_y=True), dict(store_CAu=False),
             dict(BT_use_layer_fluxes=False), dict(ni=70, nj=10, nk=2, seed=8)]


@pytest.mark.gpu
@pytest.mark.parametrize("kw", RK2_CASES, ids=[",".join(f"{k}={v}" for k, v in c.items()) or "default" for c in RK2_CASES])
def test_step_matches_oracle_bitwise(kw):
    import torch
    from mom6_amd.dynamics_split_rk2 import initialize_dyn_split_RK2, step_MOM_dyn_split_RK2
    from mom6_amd.tracer_advect import DeviceGrid
    kw = dict(kw)
    opts = {k: kw.pop(k) for k in ("use_bt_cont", "store_CAu", "BT_use_layer_fluxes") if k in kw}
    g, d, taux, tauy = make_case(**kw)
    dt = 1800.0
    use_bt = opts.get("use_bt_cont", True)
    ref = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], dt, use_bt_cont=use_bt, store_CAu=opts.get("store_CAu", True),
                       BT_use_layer_fluxes=opts.get("BT_use_layer_fluxes", True))
    ref.bcs.dtbt = dt / 9.6
    dg = DeviceGrid(g)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    u, v, h, Tt, Ss = (T(d[k]) for k in ("u", "v", "h", "T", "S"))
    Z = lambda pos, k3=True: torch.zeros(g.shape3(pos) if k3 else g.shape2(pos), dtype=torch.float64, device="cuda")
    uh, vh, uhtr, vhtr, eta_av = Z(_abi.POS_U), Z(_abi.POS_V), Z(_abi.POS_U), Z(_abi.POS_V), Z(_abi.POS_H, False)
    CS = initialize_dyn_split_RK2(u, v, h, uh, vh, dt, dg, USE_BT_CONT_TYPE=use_bt, STORE_CORIOLIS_ACCEL=opts.get("store_CAu", True),
                                  BT_USE_LAYER_FLUXES=opts.get("BT_use_layer_fluxes", True), coriolis=dict(bound_coriolis=True),
                                  barotropic=dict(BT_THICK_SCHEME="FROM_BT_CONT" if use_bt else "HARMONIC"))
    CS.barotropic_CSp.st.dtbt = ref.bcs.dtbt
    for n in ("eta", "h_av", "CAu_pred", "u_av"):
        assert bits_equal(CS.arrays[n].cpu().numpy(), ref.arrs[n]), ("init", n)
    tx, ty = T(taux), T(tauy)
    for n in range(3):
        ref.step(taux, tauy, calc_dtbt=(n == 1))
        step_MOM_dyn_split_RK2(u, v, h, (Tt, Ss), None, None, dt, (tx, ty), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS,
                               calc_dtbt=(n == 1))
        dg.sync()
        assert CS.barotropic_CSp.st.dtbt == ref.bcs.dtbt
        for name, a, b in (("u", u, ref.u), ("v", v, ref.v), ("h", h, ref.h), ("uh", uh, ref.uh), ("vh", vh, ref.vh),
                           ("uhtr", uhtr, ref.uhtr), ("eta_av", eta_av, ref.eta_av), ("eta", CS.eta, ref.arrs["eta"]),
                           ("u_av", CS.u_av, ref.arrs["u_av"]), ("h_av", CS.h_av, ref.arrs["h_av"]),
                           ("CAu_pred", CS.CAu_pred, ref.arrs["CAu_pred"])):
            an = a.cpu().numpy()
            assert bits_equal(an, b), (n, name, float(np.abs(an - b).max()))
    cap, lau = dg.bt_graph_stats()      # one-tile domain: every btstep replays a hipGraph of its subcycle
    assert lau == 6 and 1 <= cap <= 6, (cap, lau)
    dg.close()


@pytest.mark.gpu
def test_step_with_the_fused_barotropic_kernel_matches_oracle_bitwise(monkeypatch):
    """the subcycle replayed from its hipGraph with the fused step kernel (MOM6HIP_BT_FUSED=1), three RK2 steps"""
    monkeypatch.setenv("MOM6HIP_BT_FUSED", "1")
    test_step_matches_oracle_bitwise(dict(ni=70, nj=10, nk=2, seed=8))
    test_step_matches_oracle_bitwise(dict(reentrant_y=True))


def _visc_arrays(g, seed=9):
    rng = np.random.default_rng(seed)
    su, sv = g.shape2(_abi.POS_U), g.shape2(_abi.POS_V)
    return dict(Kv_bbl_u=1.0e-3 * (0.5 + rng.random(su)), Kv_bbl_v=1.0e-3 * (0.5 + rng.random(sv)),
                bbl_thick_u=2.0 + 8.0 * rng.random(su), bbl_thick_v=2.0 + 8.0 * rng.random(sv))


def test_oracle_step_with_vertvisc_conserves_volume_and_damps():
    """the step with the library's vertical viscosity: volume is still conserved, visc_rem < 1 feeds the
    barotropic solver, and the bottom drag takes kinetic energy out compared with the inviscid step"""
    g, d, taux, tauy = make_case()
    dt = 1800.0
    arrs = _visc_arrays(g)
    inv = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], dt)
    vis = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], dt, vertvisc=orc.vertvisc_cs(g, Kv=1.0e-2, Hbbl=10.0),
                       visc=orc.vertvisc_type(**arrs))
    v0 = volume(g, vis.h)
    for st in (inv, vis):
        st.bcs.dtbt = dt / 12.6
        for n in range(4):
            st.step(0.0 * taux, tauy)
            assert np.all(np.isfinite(st.u)) and np.all(np.isfinite(st.h)) and st.h.min() > 0
    # (flux-form continuity; only the Angstrom floor of nearly vanished layers adds volume)
    assert abs(volume(g, vis.h) - v0) <= 1e-9 * v0
    vr = vis.arrs["visc_rem_u"]
    assert vr.max() <= 1.0 + 1e-12 and vr[:, np.asarray(g.mask2dCu) > 0].min() < 1.0
    ke = lambda s: float((s.u ** 2).sum() + (s.v ** 2).sum())
    assert ke(vis) < ke(inv)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(ni=44, nj=40, nk=2, reentrant_y=True), dict(use_bt_cont=False, nk=5)],
                         ids=["default", "44x40x2", "no_bt_cont"])
def test_step_with_vertvisc_matches_oracle_bitwise(kw):
    import torch
    from mom6_amd.dynamics_split_rk2 import initialize_dyn_split_RK2, step_MOM_dyn_split_RK2
    from mom6_amd.tracer_advect import DeviceGrid
    from mom6_amd.vert_friction import vertvisc_type
    kw = dict(kw)
    use_bt = kw.pop("use_bt_cont", True)
    g, d, taux, tauy = make_case(**kw)
    dt = 1800.0
    arrs = _visc_arrays(g)
    ref = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], dt, use_bt_cont=use_bt,
                       vertvisc=orc.vertvisc_cs(g, Kv=1.0e-3, Hbbl=10.0, Hmix=20.0, Kvml_invZ2=1.0e-3), visc=orc.vertvisc_type(**arrs))
    ref.bcs.dtbt = dt / 9.6
    dg = DeviceGrid(g)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    u, v, h, Tt, Ss = (T(d[k]) for k in ("u", "v", "h", "T", "S"))
    Z = lambda pos, k3=True: torch.zeros(g.shape3(pos) if k3 else g.shape2(pos), dtype=torch.float64, device="cuda")
    uh, vh, uhtr, vhtr, eta_av = Z(_abi.POS_U), Z(_abi.POS_V), Z(_abi.POS_U), Z(_abi.POS_V), Z(_abi.POS_H, False)
    CS = initialize_dyn_split_RK2(u, v, h, uh, vh, dt, dg, USE_BT_CONT_TYPE=use_bt, coriolis=dict(bound_coriolis=True),
                                  barotropic=dict(BT_THICK_SCHEME="FROM_BT_CONT" if use_bt else "HARMONIC"),
                                  vertvisc=dict(KV=1.0e-3, HBBL=10.0, HMIX_FIXED=20.0, KV_ML_INVZ2=1.0e-3))
    CS.barotropic_CSp.st.dtbt = ref.bcs.dtbt
    visc = vertvisc_type(**{n: T(a) for n, a in arrs.items()})
    tx, ty = T(taux), T(tauy)
    for n in range(3):
        ref.step(taux, tauy)
        step_MOM_dyn_split_RK2(u, v, h, (Tt, Ss), visc, None, dt, (tx, ty), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS)
        dg.sync()
        for name, a, b in (("u", u, ref.u), ("v", v, ref.v), ("h", h, ref.h), ("uh", uh, ref.uh), ("vh", vh, ref.vh),
                           ("eta_av", eta_av, ref.eta_av), ("visc_rem_u", CS.visc_rem_u, ref.arrs["visc_rem_u"]),
                           ("visc_rem_v", CS.visc_rem_v, ref.arrs["visc_rem_v"]), ("a_u", CS.vertvisc_CSp.a_u, ref.vvcs._arrs["a_u"]),
                           ("u_av", CS.u_av, ref.arrs["u_av"])):
            an = a.cpu().numpy()
            assert bits_equal(an, b), (n, name, float(np.abs(an - b).max()))
    dg.close()


def test_oracle_bound_bt_correction_limits_the_mass_source():
    """BOUND_BT_CORRECTION (MOM_barotropic.F90:1587-1615): with a tiny MAXCFL_BT_CONT the positive corrections are cut to what the
    open faces can carry; with the default 0.25 nothing is cut in a quiet ocean, and the option changes no answers"""
    g, d, taux, tauy = make_case()
    runs = {}
    for name, kw in (("off", {}), ("default", dict(bound_BT_corr=1, maxCFL_BT_cont=0.25)), ("tight", dict(bound_BT_corr=1, maxCFL_BT_cont=1.0e-18))):
        st = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], 1800.0, **kw)
        st.bcs.dtbt = 1800.0 / 9.6
        for n in range(3):
            st.step(taux, tauy)
        runs[name] = st
    assert bits_equal(runs["off"].h, runs["default"].h) and bits_equal(runs["off"].bcs_arrs["eta_cor"], runs["default"].bcs_arrs["eta_cor"])
    # (the bound of a source is what the faces carry at MAXCFL_BT_CONT plus the divergence of uhbt0, so it is not simply small)
    a, b = runs["off"].bcs_arrs["eta_cor"], runs["tight"].bcs_arrs["eta_cor"]
    assert a.max() > 0 and not bits_equal(a, b)
    assert not bits_equal(runs["off"].h, runs["tight"].h) and np.abs(runs["off"].h - runs["tight"].h).max() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("maxcfl", [0.25, 1.0e-18])
def test_step_with_bound_bt_correction_matches_oracle_bitwise(maxcfl):
    import torch
    from mom6_amd.dynamics_split_rk2 import initialize_dyn_split_RK2, step_MOM_dyn_split_RK2
    from mom6_amd._lib import Mom6HipError
    from mom6_amd.tracer_advect import DeviceGrid
    g, d, taux, tauy = make_case(ni=70, nj=21, nk=3, seed=6)
    dt = 1800.0
    ref = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], dt, bound_BT_corr=1, maxCFL_BT_cont=maxcfl)
    ref.bcs.dtbt = dt / 9.6
    dg = DeviceGrid(g)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    u, v, h, Tt, Ss = (T(d[k]) for k in ("u", "v", "h", "T", "S"))
    Z = lambda pos, k3=True: torch.zeros(g.shape3(pos) if k3 else g.shape2(pos), dtype=torch.float64, device="cuda")
    uh, vh, uhtr, vhtr, eta_av = Z(_abi.POS_U), Z(_abi.POS_V), Z(_abi.POS_U), Z(_abi.POS_V), Z(_abi.POS_H, False)
    with pytest.raises(Mom6HipError, match="BOUND_BT_CORRECTION"):      # without BT_cont the option is refused by name
        initialize_dyn_split_RK2(u, v, h, uh, vh, dt, dg, USE_BT_CONT_TYPE=False,
                                 barotropic=dict(BOUND_BT_CORRECTION=True, BT_THICK_SCHEME="HARMONIC"))
        raise Mom6HipError("BOUND_BT_CORRECTION accepted without BT_cont")
    CS = initialize_dyn_split_RK2(u, v, h, uh, vh, dt, dg, coriolis=dict(bound_coriolis=True),
                                  barotropic=dict(BOUND_BT_CORRECTION=True, MAXCFL_BT_CONT=maxcfl))
    CS.barotropic_CSp.st.dtbt = ref.bcs.dtbt
    tx, ty = T(taux), T(tauy)
    for n in range(3):
        ref.step(taux, tauy)
        step_MOM_dyn_split_RK2(u, v, h, (Tt, Ss), None, None, dt, (tx, ty), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS)
        dg.sync()
        for name, a, b in (("u", u, ref.u), ("v", v, ref.v), ("h", h, ref.h), ("eta", CS.eta, ref.arrs["eta"]),
                           ("eta_cor", CS.barotropic_CSp.eta_cor, ref.bcs_arrs["eta_cor"])):
            assert bits_equal(a.cpu().numpy(), b), (n, name)
    dg.close()


def test_oracle_bt_project_velocity_is_a_consistent_alternative():
    """BT_PROJECT_VELOCITY (MOM_barotropic.F90:804-808, :1751, :1870): another second-order treatment of the barotropic transports;
    it conserves volume and stays close to the default"""
    g, d, taux, tauy = make_case()
    runs = []
    for kw in ({}, dict(BT_project_velocity=1)):
        st = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], 1800.0, **kw)
        st.bcs.dtbt = 1800.0 / 9.6
        v0 = volume(g, st.h)
        for n in range(4):
            st.step(taux, tauy)
        assert abs(volume(g, st.h) - v0) <= 1e-12 * v0 and np.all(np.isfinite(st.u))
        runs.append(st)
    dh = np.abs(interior(g, runs[0].h) - interior(g, runs[1].h)).max()
    assert 0 < dh < 0.05 * np.abs(interior(g, runs[0].h) - interior(g, d["h"])).max() + 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(use_bt_cont=False), dict(om4=True)], ids=["default", "no_bt_cont", "with_bound_bt_correction"])
def test_step_with_bt_project_velocity_matches_oracle_bitwise(kw):
    """BT_PROJECT_VELOCITY, alone and together with BOUND_BT_CORRECTION (the barotropic settings of OM4)"""
    import torch
    from mom6_amd.dynamics_split_rk2 import initialize_dyn_split_RK2, step_MOM_dyn_split_RK2
    from mom6_amd.tracer_advect import DeviceGrid
    use_bt, om4 = kw.get("use_bt_cont", True), kw.get("om4", False)
    g, d, taux, tauy = make_case(ni=70, nj=21, nk=3, seed=7)
    dt = 1800.0
    okw = dict(bound_BT_corr=1, maxCFL_BT_cont=0.25) if om4 else {}
    ref = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], dt, use_bt_cont=use_bt, BT_project_velocity=1, **okw)
    ref.bcs.dtbt = dt / 9.6
    dg = DeviceGrid(g)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    u, v, h, Tt, Ss = (T(d[k]) for k in ("u", "v", "h", "T", "S"))
    Z = lambda pos, k3=True: torch.zeros(g.shape3(pos) if k3 else g.shape2(pos), dtype=torch.float64, device="cuda")
    uh, vh, uhtr, vhtr, eta_av = Z(_abi.POS_U), Z(_abi.POS_V), Z(_abi.POS_U), Z(_abi.POS_V), Z(_abi.POS_H, False)
    bkw = dict(BT_PROJECT_VELOCITY=True, BT_THICK_SCHEME="FROM_BT_CONT" if use_bt else "HARMONIC")
    if om4:
        bkw.update(BOUND_BT_CORRECTION=True)
    CS = initialize_dyn_split_RK2(u, v, h, uh, vh, dt, dg, USE_BT_CONT_TYPE=use_bt, coriolis=dict(bound_coriolis=True), barotropic=bkw)
    CS.barotropic_CSp.st.dtbt = ref.bcs.dtbt
    tx, ty = T(taux), T(tauy)
    for n in range(3):
        ref.step(taux, tauy)
        step_MOM_dyn_split_RK2(u, v, h, (Tt, Ss), None, None, dt, (tx, ty), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS)
        dg.sync()
        for name, a, b in (("u", u, ref.u), ("v", v, ref.v), ("h", h, ref.h), ("eta", CS.eta, ref.arrs["eta"]), ("eta_av", eta_av, ref.eta_av),
                           ("u_av", CS.u_av, ref.arrs["u_av"])):
            assert bits_equal(a.cpu().numpy(), b), (n, name)
    dg.close()


# ---- the hot-path switches of the reference's .testing configurations, in the whole step ---------------------------------------
# (values transcribed from .testing/tc4/MOM_input, tc2/MOM_input and tc1/MOM_input; the Fortran shims read the same sets by name in
# tests/test_testing_configs.py)
TC_SETS = {
    # tc4: 14 x 10 x 2, ALE z*, EQN_OF_STATE = LINEAR with DRHO_DS = 0, RECONSTRUCT_FOR_PRESSURE = False, BE = 0.7, BEBT = 0.2,
    # BOUND_BT_CORRECTION, CORIOLIS_EN_DIS, DIRECT_STRESS with HMIX_FIXED = 20, KV_ML_INVZ2 = 0.01, KV = 1e-4, HBBL = 10,
    # SMAGORINSKY_AH with SMAG_BI_CONST = 0.03, ETA_TOLERANCE = 1e-12, DT = 1200
    "tc4": dict(shape=(14, 10, 2), reentrant_x=False, dt=1200.0, be=0.7, eos=("LINEAR", 1000.0, -0.2, 0.0), pressureforce=dict(reconstruct=False),
                bt=dict(bebt=0.2, bound_BT_corr=1, maxCFL_BT_cont=0.25),
                cor=dict(coriolis_en_dis=1, bound_coriolis=0),      # (CORIOLIS_EN_DIS switches BOUND_CORIOLIS off, MOM_CoriolisAdv.F90:1155)
                vv=dict(Kv=1.0e-4, Hbbl=10.0, Hmix=20.0, Kvml_invZ2=0.01, direct_stress=True, maxvel=10.0, CFL_based_trunc=False),
                hv=dict(biharmonic=1, Smagorinsky_Ah=1, Smag_bi_const=0.03, use_land_mask=0), tol_eta=1.0e-12, ml=None),
    # tc2: 10 x 8 x 8, ALE z*, WRIGHT, DT = 3600, BEBT = 0.2, BOUND_BT_CORRECTION, NONLINEAR_BT_CONTINUITY, BT_PROJECT_VELOCITY,
    # DYNAMIC_VISCOUS_ML with BULK_RI_ML = 0.05, TKE_DECAY = 10, ML_OMEGA_FRAC = 1, HMIX_FIXED = 0.5, KV = 1e-4, HBBL = 10, MAXVEL = 10,
    # LAPLACIAN + SMAGORINSKY_KH (0.06, KH_VEL_SCALE = 0.05) and SMAGORINSKY_AH (0.06, AH_VEL_SCALE = 0.05), ETA_TOLERANCE = 1e-6,
    # VELOCITY_TOLERANCE = 1e-3  (CHANNEL_DRAG belongs to set_viscous_BBL, outside the step)
    "tc2": dict(shape=(10, 8, 8), reentrant_x=False, dt=3600.0, be=0.6, eos=("WRIGHT",), pressureforce=dict(),
                bt=dict(bebt=0.2, bound_BT_corr=1, maxCFL_BT_cont=0.25, Nonlinear_continuity=1, BT_project_velocity=1), cor=dict(),
                vv=dict(Kv=1.0e-4, Hbbl=10.0, Hmix=0.5, maxvel=10.0, CFL_based_trunc=False, dynamic_viscous_ML=True),
                hv=dict(Laplacian=1, biharmonic=1, Smagorinsky_Kh=1, Smag_Lap_const=0.06, Kh_vel_scale=0.05, Smagorinsky_Ah=1, Smag_bi_const=0.06,
                        Ah_vel_scale=0.05, use_land_mask=0), tol_eta=1.0e-6, tol_vel=1.0e-3,
                ml=dict(bulk_Ri_ML=0.05, TKE_decay=10.0, omega_frac=1.0)),
    # tc1: 10 x 8 x 8, the same dynamics switches as tc2 with DT = 900 and biharmonic Smagorinsky only (AH_VEL_SCALE = 0.05, 0.06); its
    # layered (bulk mixed layer) thermodynamics is outside the step: here the step runs with nkml = 2 layers always in the viscous mixed
    # layer and PressureForce without ALE (int_density_dz, nk_rho_varies = 4 with GV%Rlay)
    "tc1": dict(shape=(10, 8, 8), reentrant_x=False, dt=900.0, be=0.6, eos=("WRIGHT",),
                pressureforce=dict(use_ALE=False, nkmb=4, Rlay=np.linspace(1024.0, 1028.0, 8)),
                bt=dict(bebt=0.2, bound_BT_corr=1, maxCFL_BT_cont=0.25, Nonlinear_continuity=1, BT_project_velocity=1), cor=dict(),
                vv=dict(Kv=1.0e-4, Hbbl=10.0, maxvel=10.0, CFL_based_trunc=False, dynamic_viscous_ML=True, nkml=2),
                hv=dict(biharmonic=1, Smagorinsky_Ah=1, Smag_bi_const=0.06, Ah_vel_scale=0.05, use_land_mask=0), tol_eta=1.0e-6, tol_vel=1.0e-3,
                ml=dict(bulk_Ri_ML=0.05, TKE_decay=10.0, omega_frac=1.0, nkml=2)),
}


def tc_oracle_state(name):
    c = TC_SETS[name]
    ni, nj, nk = c["shape"]
    g, d, taux, tauy = make_case(ni=ni, nj=nj, nk=nk, seed=21, reentrant_x=c["reentrant_x"])
    rng = np.random.default_rng(17)
    arrs = _visc_arrays(g)
    if c["ml"]:
        arrs.update(ustar=np.ascontiguousarray(0.004 + 0.008 * rng.random(g.shape2(_abi.POS_H))),
                    nkml_visc_u=g.zeros2(_abi.POS_U), nkml_visc_v=g.zeros2(_abi.POS_V))
    E = orc.eos(*c["eos"])
    ccs_kw = dict(tol_eta=c["tol_eta"])
    st = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], c["dt"], be=c["be"], eos_form=E, pressureforce=c["pressureforce"],
                      vertvisc=orc.vertvisc_cs(g, **c["vv"]), visc=orc.vertvisc_type(**arrs), hor_visc=orc.hor_visc_cs(g, c["dt"], **c["hv"]),
                      set_visc=orc.set_visc_cs(g, 10.0, 1.0e-4, dynamic_viscous_ML=True, **c["ml"]) if c["ml"] else None,
                      continuity={k: c[k] for k in ("tol_eta", "tol_vel") if k in c}, coriolis=c["cor"], **c["bt"])
    return g, d, taux, tauy, arrs, st


@pytest.mark.parametrize("name", list(TC_SETS))
def test_oracle_step_with_the_testing_switch_sets(name):
    """two steps with each configuration's hot-path switches: finite, volume conserving, and not the default step"""
    g, d, taux, tauy, arrs, st = tc_oracle_state(name)
    v0 = volume(g, st.h)
    for n in range(2):
        st.step(taux, tauy, calc_dtbt=(n == 0))
    assert np.all(np.isfinite(st.u)) and np.all(np.isfinite(st.h)) and st.h.min() > 0
    assert abs(volume(g, st.h) - v0) <= 1e-9 * v0
    if TC_SETS[name]["ml"]:
        nkv = st.visc._keep["nkml_visc_u"]
        assert nkv.max() >= 1.0 and nkv.max() <= g.nk


# the same sets by the reference's parameter names, for the library's host mirror
TC_PARAMS = {
    "tc4": dict(BE=0.7, EQN_OF_STATE="LINEAR", eos=dict(Rho_T0_S0=1000.0, dRho_dT=-0.2, dRho_dS=0.0), pressure_force=dict(reconstruct=False),
                barotropic=dict(BEBT=0.2, BOUND_BT_CORRECTION=True), coriolis=dict(bound_coriolis=True, coriolis_en_dis=True),
                continuity=dict(tol_eta=1.0e-12),
                vertvisc=dict(KV=1.0e-4, HBBL=10.0, HMIX_FIXED=20.0, KV_ML_INVZ2=0.01, DIRECT_STRESS=True, MAXVEL=10.0, CFL_BASED_TRUNCATIONS=False),
                hor_visc=dict(BIHARMONIC=True, SMAGORINSKY_AH=True, SMAG_BI_CONST=0.03, USE_LAND_MASK_FOR_HVISC=False)),
    "tc2": dict(EQN_OF_STATE="WRIGHT",
                barotropic=dict(BEBT=0.2, BOUND_BT_CORRECTION=True, NONLINEAR_BT_CONTINUITY=True, BT_PROJECT_VELOCITY=True),
                coriolis=dict(bound_coriolis=True), continuity=dict(tol_eta=1.0e-6, tol_vel=1.0e-3),
                vertvisc=dict(KV=1.0e-4, HBBL=10.0, HMIX_FIXED=0.5, MAXVEL=10.0, CFL_BASED_TRUNCATIONS=False, DYNAMIC_VISCOUS_ML=True),
                hor_visc=dict(LAPLACIAN=True, BIHARMONIC=True, SMAGORINSKY_KH=True, SMAG_LAP_CONST=0.06, KH_VEL_SCALE=0.05, SMAGORINSKY_AH=True,
                              SMAG_BI_CONST=0.06, AH_VEL_SCALE=0.05, USE_LAND_MASK_FOR_HVISC=False),
                set_visc=dict(HBBL=10.0, KV=1.0e-4, DYNAMIC_VISCOUS_ML=True, BULK_RI_ML=0.05, TKE_DECAY=10.0, ML_OMEGA_FRAC=1.0)),
    "tc1": dict(EQN_OF_STATE="WRIGHT", pressure_force=dict(use_ALE=False, nk_rho_varies=4, Rlay=np.linspace(1024.0, 1028.0, 8)),
                barotropic=dict(BEBT=0.2, BOUND_BT_CORRECTION=True, NONLINEAR_BT_CONTINUITY=True, BT_PROJECT_VELOCITY=True),
                coriolis=dict(bound_coriolis=True), continuity=dict(tol_eta=1.0e-6, tol_vel=1.0e-3),
                vertvisc=dict(KV=1.0e-4, HBBL=10.0, MAXVEL=10.0, CFL_BASED_TRUNCATIONS=False, DYNAMIC_VISCOUS_ML=True, NKML=2),
                hor_visc=dict(BIHARMONIC=True, SMAGORINSKY_AH=True, SMAG_BI_CONST=0.06, AH_VEL_SCALE=0.05, USE_LAND_MASK_FOR_HVISC=False),
                set_visc=dict(HBBL=10.0, KV=1.0e-4, DYNAMIC_VISCOUS_ML=True, BULK_RI_ML=0.05, TKE_DECAY=10.0, ML_OMEGA_FRAC=1.0, NKML=2)),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(TC_SETS))
def test_step_with_the_testing_switch_sets_matches_oracle_bitwise(name):
    """the hot-path switch sets of .testing/tc4, tc2 and tc1 in step_MOM_dyn_split_RK2 (three steps, set_dtbt in the first): the
    library's bits are the oracle's"""
    import torch
    from mom6_amd.dynamics_split_rk2 import initialize_dyn_split_RK2, step_MOM_dyn_split_RK2
    from mom6_amd.tracer_advect import DeviceGrid
    from mom6_amd.vert_friction import vertvisc_type
    g, d, taux, tauy, arrs, ref = tc_oracle_state(name)
    c = TC_SETS[name]
    dt = c["dt"]
    dg = DeviceGrid(g)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    u, v, h, Tt, Ss = (T(d[k]) for k in ("u", "v", "h", "T", "S"))
    Z = lambda pos, k3=True: torch.zeros(g.shape3(pos) if k3 else g.shape2(pos), dtype=torch.float64, device="cuda")
    uh, vh, uhtr, vhtr, eta_av = Z(_abi.POS_U), Z(_abi.POS_V), Z(_abi.POS_U), Z(_abi.POS_V), Z(_abi.POS_H, False)
    CS = initialize_dyn_split_RK2(u, v, h, uh, vh, dt, dg, **TC_PARAMS[name])
    for n in ("eta", "h_av", "CAu_pred", "u_av", "diffu"):
        assert bits_equal(CS.arrays[n].cpu().numpy(), ref.arrs[n]), ("init", n)
    va = {n: T(a) for n, a in arrs.items()}
    visc = vertvisc_type(**va)
    tx, ty = T(taux), T(tauy)
    for n in range(3):
        ref.step(taux, tauy, calc_dtbt=(n == 0))
        step_MOM_dyn_split_RK2(u, v, h, (Tt, Ss), visc, None, dt, (tx, ty), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS, calc_dtbt=(n == 0))
        dg.sync()
        assert CS.barotropic_CSp.st.dtbt == ref.bcs.dtbt and CS.barotropic_CSp.st.nstep_last == ref.bcs.nstep_last
        checks = [("u", u, ref.u), ("v", v, ref.v), ("h", h, ref.h), ("uh", uh, ref.uh), ("vh", vh, ref.vh), ("eta_av", eta_av, ref.eta_av),
                  ("eta", CS.eta, ref.arrs["eta"]), ("PFu", CS.PFu, ref.arrs["PFu"]), ("pbce", CS.pbce, ref.arrs["pbce"]),
                  ("diffu", CS.diffu, ref.arrs["diffu"]), ("visc_rem_u", CS.visc_rem_u, ref.arrs["visc_rem_u"]),
                  ("a_v", CS.vertvisc_CSp.a_v, ref.vvcs._arrs["a_v"])]
        if c["ml"]:
            checks += [("nkml_visc_u", va["nkml_visc_u"], ref.visc._keep["nkml_visc_u"]), ("nkml_visc_v", va["nkml_visc_v"], ref.visc._keep["nkml_visc_v"])]
        for nm, a, b in checks:
            an = a.cpu().numpy()
            assert bits_equal(an, b), (name, n, nm, float(np.abs(an - b).max()))
    dg.close()
