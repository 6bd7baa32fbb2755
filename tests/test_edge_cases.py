"""Edge cases of the hot path on the GPU, bit for bit against the oracle: a single layer (every k recurrence degenerates),
no land and almost only land (mask branches), domains narrower than one wave / one block tile, and the whole RK2 step
(with vertical viscosity) on each of them."""
import numpy as np
import pytest

from mom6_amd import _abi, synth
from helpers import bits_equal
from oracle import orc

H, U, V = _abi.POS_H, _abi.POS_U, _abi.POS_V
CASES = [
    dict(ni=12, nj=9, nk=1, land_frac=0.25),          # one layer
    dict(ni=20, nj=14, nk=3, land_frac=0.0),          # no land at all
    dict(ni=26, nj=18, nk=4, land_frac=0.85),         # a few wet cells
    dict(ni=5, nj=5, nk=2, land_frac=0.0, reentrant_x=True, reentrant_y=True),      # the tile is barely wider than the halo
    dict(ni=70, nj=5, nk=2, land_frac=0.2, reentrant_x=False),                      # more than one block tile in x, closed
]
IDS = ["nk1", "all_ocean", "mostly_land", "5x5_doubly_periodic", "70x5_closed"]


def _grid(c):
    kw = {k: v for k, v in c.items() if k not in ("ni", "nj", "nk")}
    return synth.make_grid(c["ni"], c["nj"], c["nk"], seed=77, **kw)


def _T(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).copy()).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES, ids=IDS)
def test_operators_on_edge_grids(oracle, c):
    import torch
    from mom6_amd.continuity import BT_cont_type, continuity
    from mom6_amd.coriolis_adv import CorAdCalc, CoriolisAdv_init
    from mom6_amd.tracer_advect import DeviceGrid
    from mom6_amd.vert_friction import vertvisc_init, vertvisc_step, vertvisc_type
    g = _grid(c)
    st = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=3, umax=0.2).items()}
    dg = DeviceGrid(g)
    dt = 900.0
    # ---- continuity with uhbt and BT_cont
    cs = oracle.continuity_cs(g.nk, g.Angstrom_H)
    h1, uh, vh = st["h"].copy(), np.zeros_like(st["u"]), np.zeros_like(st["v"])
    oracle.continuity(g, cs, st["u"], st["v"], st["h"].copy(), h1, uh, vh, dt)
    uhbt = np.ascontiguousarray(uh.sum(0) * 1.05); vhbt = np.ascontiguousarray(vh.sum(0) * 0.95)
    arrs, btst = oracle.make_bt_cont(g, with_h=True)
    rh, ruh, rvh, ruc, rvc = st["h"].copy(), np.zeros_like(st["u"]), np.zeros_like(st["v"]), np.zeros_like(st["u"]), np.zeros_like(st["v"])
    oracle.continuity(g, cs, st["u"], st["v"], st["h"].copy(), rh, ruh, rvh, dt, uhbt=uhbt, vhbt=vhbt, u_cor=ruc, v_cor=rvc, bt_cont=btst)
    arrs2, _ = oracle.make_bt_cont(g, with_h=True)
    bt = BT_cont_type(**{n: _T(a) for n, a in arrs2.items()})
    dh, duh, dvh, duc, dvc = _T(st["h"]), _T(np.zeros_like(st["u"])), _T(np.zeros_like(st["v"])), _T(np.zeros_like(st["u"])), _T(np.zeros_like(st["v"]))
    continuity(_T(st["u"]), _T(st["v"]), _T(st["h"]), dh, duh, dvh, dt, dg, cs, uhbt=_T(uhbt), vhbt=_T(vhbt), u_cor=duc, v_cor=dvc, BT_cont=bt)
    dg.sync()
    for name, a, b in (("h", rh, dh), ("uh", ruh, duh), ("vh", rvh, dvh), ("u_cor", ruc, duc), ("v_cor", rvc, dvc)):
        assert bits_equal(a, b.cpu().numpy()), ("continuity", name)
    for n, a in arrs.items():
        assert bits_equal(a, bt.arrays[n].cpu().numpy()), ("BT_cont", n)
    # ---- CorAdCalc
    rCAu, rCAv = oracle.coradcalc(g, st["u"], st["v"], st["h"], ruh, rvh, bound_coriolis=True)
    dCAu, dCAv = _T(np.zeros_like(st["u"])), _T(np.zeros_like(st["v"]))
    CorAdCalc(_T(st["u"]), _T(st["v"]), _T(st["h"]), _T(ruh), _T(rvh), dCAu, dCAv, None, dg, CoriolisAdv_init(bound_coriolis=True))
    dg.sync()
    assert bits_equal(rCAu, dCAu.cpu().numpy()) and bits_equal(rCAv, dCAv.cpu().numpy()), "CorAdCalc"
    # ---- vertical viscosity (coefficients + solve + remnant)
    rng = np.random.default_rng(1)
    va = dict(Kv_bbl_u=1e-3 * (0.5 + rng.random(g.shape2(U))), Kv_bbl_v=1e-3 * (0.5 + rng.random(g.shape2(V))),
              bbl_thick_u=2.0 + 8.0 * rng.random(g.shape2(U)), bbl_thick_v=2.0 + 8.0 * rng.random(g.shape2(V)))
    taux = np.ascontiguousarray(0.1 * np.asarray(g.mask2dCu)); tauy = np.ascontiguousarray(-0.05 * np.asarray(g.mask2dCv))
    rcs = oracle.vertvisc_cs(g, Kv=1e-4, Hbbl=10.0, Hmix=20.0, Kvml_invZ2=1e-2); rv = oracle.vertvisc_type(**va)
    ru, rvv = st["u"].copy(), st["v"].copy(); rru, rrv = g.zeros3(U), g.zeros3(V)
    oracle.vertvisc_coef(g, rcs, ru, rvv, st["h"], rv, dt)
    oracle.vertvisc(g, rcs, ru, rvv, st["h"], taux, tauy, rv, dt)
    oracle.vertvisc_remnant(g, rcs, rv, rru, rrv, dt)
    CS = vertvisc_init(dg, KV=1e-4, HBBL=10.0, HMIX_FIXED=20.0, KV_ML_INVZ2=1e-2)
    visc = vertvisc_type(**{n: _T(a) for n, a in va.items()})
    du, dv, dru, drv = _T(st["u"]), _T(st["v"]), _T(g.zeros3(U)), _T(g.zeros3(V))
    vertvisc_step(du, dv, _T(st["h"]), None, (_T(taux), _T(tauy)), visc, dt, dg, CS, dru, drv, True)
    dg.sync()
    for name, a, b in (("u", ru, du), ("v", rvv, dv), ("visc_rem_u", rru, dru), ("visc_rem_v", rrv, drv)):
        assert bits_equal(a, b.cpu().numpy()), ("vertvisc", name)
    dg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES, ids=IDS)
def test_thermodynamic_block_on_edge_grids(oracle, c):
    from mom6_amd.ale import ALE_regrid, ALE_remap_set_h_vel, ALE_remap_tracers, ALE_remap_velocities, initialize_regridding, initialize_remapping
    from mom6_amd.tracer_advect import DeviceGrid, advect_tracer, tracer_advect_init
    g = _grid(c)
    dg = DeviceGrid(g)
    ad = synth.make_advection_state(g, ntr=2, seed=4)
    ref = [t.numpy().copy() for t in ad["tr"]]
    rstats = oracle.advect_tracer(g, ad["h_end"].numpy(), ad["uhtr"].numpy(), ad["vhtr"].numpy(), 3600.0, 900.0, "PPM:H3", ref)
    tr = [t.cuda() for t in ad["tr"]]
    stats = advect_tracer(ad["h_end"].cuda(), ad["uhtr"].cuda(), ad["vhtr"].cuda(), None, 3600.0, dg, tracer_advect_init(900.0, "PPM:H3"), tr)
    dg.sync()
    assert stats.iterations == rstats.iterations
    for a, b in zip(ref, tr):
        assert bits_equal(a, b.cpu().numpy()), "advect_tracer"
    # regrid + remap of tracers and velocities
    st = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=3, umax=0.2).items()}
    kn = (np.arange(g.nk) + 0.5) / g.nk
    res = (2.0 + 300.0 * kn ** 2); res = res * (5500.0 / res.sum())
    rcs = oracle.regridding_cs(res, old_grid_weight=0.5)
    h_new, dzr = oracle.ale_regrid(g, rcs, st["h"])
    rT, rS = st["T"].copy(), st["S"].copy()
    oracle.ale_remap_tracers(g, "PPM_H4", st["h"], h_new, [rT, rS])
    hou, hov = oracle.ale_remap_set_h_vel(g, st["h"]); hnu, hnv = oracle.ale_remap_set_h_vel(g, h_new)
    ru, rv = st["u"].copy(), st["v"].copy()
    oracle.ale_remap_velocities(g, "PPM_H4", hou, hov, hnu, hnv, ru, rv)
    import torch
    dh = _T(st["h"]); dhn = torch.zeros_like(dh); ddz = torch.zeros((g.nk + 1,) + g.shape2(H), dtype=torch.float64, device="cuda")
    ALE_regrid(dg, dh, dhn, ddz, None, initialize_regridding(dg, coordinateResolution=res, old_grid_weight=0.5))
    dT, dS = _T(st["T"]), _T(st["S"])
    R = initialize_remapping("PPM_H4")
    ALE_remap_tracers(R, dg, dh, dhn, [dT, dS])
    Z = lambda p: torch.zeros(g.shape3(p), dtype=torch.float64, device="cuda")
    a, b, cc, d = Z(U), Z(V), Z(U), Z(V)
    ALE_remap_set_h_vel(None, dg, dh, a, b); ALE_remap_set_h_vel(None, dg, dhn, cc, d)
    du, dv = _T(st["u"]), _T(st["v"])
    ALE_remap_velocities(R, dg, a, b, cc, d, du, dv)
    dg.sync()
    for name, x, y in (("h_new", h_new, dhn), ("T", rT, dT), ("S", rS, dS), ("u", ru, du), ("v", rv, dv)):
        assert bits_equal(x, y.cpu().numpy()), ("ALE", name)
    dg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES, ids=IDS)
def test_rk2_step_on_edge_grids(oracle, c):
    import torch
    from mom6_amd.dynamics_split_rk2 import initialize_dyn_split_RK2, step_MOM_dyn_split_RK2
    from mom6_amd.tracer_advect import DeviceGrid
    from mom6_amd.vert_friction import vertvisc_type
    g = _grid(c)
    d = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=5, umax=0.05, eta_amp=0.1).items()}
    dt = 900.0
    rng = np.random.default_rng(2)
    va = dict(Kv_bbl_u=1e-3 * (0.5 + rng.random(g.shape2(U))), Kv_bbl_v=1e-3 * (0.5 + rng.random(g.shape2(V))),
              bbl_thick_u=2.0 + 8.0 * rng.random(g.shape2(U)), bbl_thick_v=2.0 + 8.0 * rng.random(g.shape2(V)))
    taux = np.ascontiguousarray(0.05 * np.asarray(g.mask2dCu)); tauy = g.zeros2(V)
    ref = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], dt, vertvisc=orc.vertvisc_cs(g, Kv=1e-3, Hbbl=10.0), visc=orc.vertvisc_type(**va)) \
        if g.nk >= 2 else None
    if ref is not None:
        ref.bcs.dtbt = dt / 6.3
    dg = DeviceGrid(g)
    u, v, h, Tt, Ss = (_T(d[k]) for k in ("u", "v", "h", "T", "S"))
    Zf = lambda pos, k3=True: torch.zeros(g.shape3(pos) if k3 else g.shape2(pos), dtype=torch.float64, device="cuda")
    uh, vh, uhtr, vhtr, eta_av = Zf(U), Zf(V), Zf(U), Zf(V), Zf(H, False)
    CS = initialize_dyn_split_RK2(u, v, h, uh, vh, dt, dg, coriolis=dict(bound_coriolis=True), vertvisc=dict(KV=1e-3, HBBL=10.0))
    if g.nk < 2:      # PressureForce_FV_Bouss with PLM reconstruction is refused for a single layer
        from mom6_amd._lib import Mom6HipError
        with pytest.raises(Mom6HipError, match="at least 2 layers"):
            step_MOM_dyn_split_RK2(u, v, h, (Tt, Ss), vertvisc_type(**{n: _T(a) for n, a in va.items()}), None, dt, (_T(taux), _T(tauy)),
                                   None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS)
        dg.close()
        return
    CS.barotropic_CSp.st.dtbt = ref.bcs.dtbt
    visc = vertvisc_type(**{n: _T(a) for n, a in va.items()})
    tx, ty = _T(taux), _T(tauy)
    for n in range(2):
        ref.step(taux, tauy)
        step_MOM_dyn_split_RK2(u, v, h, (Tt, Ss), visc, None, dt, (tx, ty), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS)
    dg.sync()
    for name, a, b in (("u", u, ref.u), ("v", v, ref.v), ("h", h, ref.h), ("uh", uh, ref.uh), ("eta_av", eta_av, ref.eta_av)):
        assert bits_equal(a.cpu().numpy(), b), ("rk2", name)
    dg.close()
