"""TRIPOLAR_N: the fold of the halo update and the polarity swaps of btstep, oracle and library, against the unfolded domain
(tests/tripolar.py)."""
import numpy as np
import pytest

import tripolar as tp
from helpers import bits_equal, interior
from mom6_amd import _abi
from oracle import orc

H, U, V, Q = _abi.POS_H, _abi.POS_U, _abi.POS_V, _abi.POS_Q


def test_fold_of_the_halo_update():
    """every staggering, vector and scalar pair: the halo beyond the fold == the rows of the unfolded field"""
    g, g2 = tp.grids(ni=12, nj=6, nk=2)
    rng = np.random.default_rng(1)
    for pos in (H, U, V, Q):
        for vec in ((False, True) if pos in (U, V) else (False,)):
            a2 = tp.symmetrize(g2, rng.standard_normal(g2.shape3(pos)), pos, vec)
            a = tp.folded(g, a2, pos)
            want = a.copy()
            sj, si = g.csl(pos)
            b = np.full_like(a, np.nan); b[:, sj, si] = a[:, sj, si]
            b[:, : g.halo] = a[:, : g.halo]                  # the closed southern halo is not touched
            orc.halo_update(g, b, pos | (0 if (vec or pos in (H, Q)) else _abi.PASS_SCALAR_PAIR))
            assert np.array_equal(b, want), (pos, vec, np.argwhere(b != want)[:4])


def run_pair(nsteps, viscous, rk2b=False, ni=20, nj=8, nk=3, use_bt_cont=True):
    g, g2 = tp.grids(ni=ni, nj=nj, nk=nk)
    d, d2, tau, tau2 = tp.states(g, g2)
    out = []
    for gg, dd, tt in ((g, d, tau), (g2, d2, tau2)):
        kw = dict(rk2b=rk2b, use_bt_cont=use_bt_cont)
        if viscous:
            visc = orc.vertvisc_type(Kv_bbl_u=1.0e-3 * gg.mask2dCu, Kv_bbl_v=1.0e-3 * gg.mask2dCv, bbl_thick_u=5.0 * gg.mask2dCu,
                                     bbl_thick_v=5.0 * gg.mask2dCv)
            kw.update(vertvisc=orc.vertvisc_cs(gg, Kv=1.0e-3, Hbbl=10.0, Hmix=20.0, Kvml_invZ2=1.0e-3), visc=visc,
                      hor_visc=orc.hor_visc_cs(gg, 900.0, biharmonic=True, Smagorinsky_Ah=True, Smag_bi_const=0.06, Ah_vel_scale=0.01))
        st = orc.DynState(gg, dd["u"], dd["v"], dd["h"], dd["T"], dd["S"], 900.0, **kw)
        st.bcs.dtbt = 900.0 / 6.6      # (the synthetic cells are hundreds of km wide: DTBT would exceed DT)
        for n in range(nsteps):
            st.step(tt[0], tt[1])
        out.append(st)
    return g, g2, out[0], out[1]


@pytest.mark.parametrize("case", [dict(viscous=False), dict(viscous=True), dict(viscous=False, rk2b=True), dict(viscous=True, use_bt_cont=False)],
                         ids=["inviscid", "viscous", "rk2b", "viscous-no_bt_cont"])
def test_oracle_folded_run_is_the_unfolded_run(case):
    g, g2, a, b = run_pair(3, **case)
    assert a.bcs.nstep_last == b.bcs.nstep_last and a.bcs.dtbt == b.bcs.dtbt and a.bcs.nstep_last > 3
    assert np.abs(a.u).max() > 1e-3 and np.abs(a.u).max() < 3.0
    for n, pos in (("h", H), ("u", U), ("v", V), ("uh", U), ("vh", V), ("uhtr", U), ("vhtr", V), ("eta_av", H)):
        x = interior(g, getattr(a, n), pos); y = interior(g, tp.folded(g, getattr(b, n), pos), pos)
        assert bits_equal(np.ascontiguousarray(x), np.ascontiguousarray(y)), (n, float(np.abs(x - y).max()), np.argwhere(x != y)[:3])
    # the flow crosses the fold: the meridional transport on the fold line is not zero
    sj, si = g.csl(V)
    assert np.abs(a.vh[:, sj, si][:, -1]).max() > 0


def _thermo_pair(g, g2, d, d2, gg_list):
    """advect_tracer (PPM:H3) + tracer_hordiff on both domains from fold-symmetric transports"""
    out = []
    for gg, dd in gg_list:
        dt = 3600.0
        uhtr = np.ascontiguousarray(dd["uh"] * dt); vhtr = np.ascontiguousarray(dd["vh"] * dt)
        tr = [dd["T"].copy(), dd["S"].copy()]
        orc.advect_tracer(gg, dd["h"], uhtr, vhtr, dt, dt, "PPM:H3", tr)
        orc.tracer_hordiff(gg, dd["h"], dt, tr, 500.0)
        out.append(tr)
    return out


def test_oracle_tracer_advection_and_diffusion_across_the_fold():
    g, g2 = tp.grids(ni=20, nj=8, nk=3)
    d, d2, _, _ = tp.states(g, g2, umax=0.05)
    a, b = _thermo_pair(g, g2, d, d2, [(g, d), (g2, d2)])
    for m in range(2):
        x = interior(g, a[m]); y = interior(g, tp.folded(g, b[m], H))
        assert bits_equal(np.ascontiguousarray(x), np.ascontiguousarray(y)), (m, float(np.abs(x - y).max()))
        assert not bits_equal(np.ascontiguousarray(x), np.ascontiguousarray(interior(g, (d["T"], d["S"])[m])))


@pytest.mark.gpu
def test_gpu_fold_of_the_halo_update():
    import torch
    from mom6_amd.tracer_advect import DeviceGrid
    g, g2 = tp.grids(ni=70, nj=10, nk=3)
    dg = DeviceGrid(g)
    rng = np.random.default_rng(2)
    SP = _abi.PASS_SCALAR_PAIR
    fields, poss, want = [], [], []
    for pf in (H, U, V, Q, U | SP, V | SP):
        for three_d in (True, False):
            a = rng.standard_normal(g.shape3(pf & 3) if three_d else g.shape2(pf & 3))
            w = a.copy(); orc.halo_update(g, w, pf)
            fields.append(torch.from_numpy(a).cuda()); poss.append(pf); want.append(w)
    dg.halo_update(fields, poss)
    dg.sync()
    for f, w, pf in zip(fields, want, poss):
        assert bits_equal(f.cpu().numpy(), w), pf
    dg.close()


TRI_CASES = [dict(), dict(viscous=True), dict(rk2b=True, viscous=True), dict(use_bt_cont=False), dict(ni=130, nj=12, nk=5, viscous=True)]


@pytest.mark.gpu
@pytest.mark.parametrize("kw", TRI_CASES, ids=[",".join(f"{k}={v}" for k, v in c.items()) or "default" for c in TRI_CASES])
def test_gpu_step_on_a_tripolar_grid_matches_oracle(kw):
    """3 steps of the RK2 (or RK2B) step on a TRIPOLAR_N grid with flow across the fold: library == oracle, bit for bit (the oracle
    itself == the unfolded domain, above)"""
    import torch
    from mom6_amd import dynamics_split_rk2 as M
    from mom6_amd.tracer_advect import DeviceGrid
    from mom6_amd.vert_friction import vertvisc_type
    kw = dict(kw)
    viscous, rk2b, use_bt = kw.pop("viscous", False), kw.pop("rk2b", False), kw.pop("use_bt_cont", True)
    g, g2 = tp.grids(**{**dict(ni=70, nj=10, nk=3), **kw})
    d, _, (taux, tauy), _ = tp.states(g, g2)
    dt = 900.0
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    okw, hkw, visc = dict(rk2b=rk2b, use_bt_cont=use_bt), {}, None
    if viscous:
        va = dict(Kv_bbl_u=1.0e-3 * g.mask2dCu, Kv_bbl_v=1.0e-3 * g.mask2dCv, bbl_thick_u=5.0 * g.mask2dCu, bbl_thick_v=5.0 * g.mask2dCv)
        okw.update(vertvisc=orc.vertvisc_cs(g, Kv=1.0e-3, Hbbl=10.0, Hmix=20.0, Kvml_invZ2=1.0e-3), visc=orc.vertvisc_type(**va),
                   hor_visc=orc.hor_visc_cs(g, dt, biharmonic=True, Smagorinsky_Ah=True, Smag_bi_const=0.06, Ah_vel_scale=0.01))
        hkw = dict(vertvisc=dict(KV=1.0e-3, HBBL=10.0, HMIX_FIXED=20.0, KV_ML_INVZ2=1.0e-3),
                   hor_visc=dict(BIHARMONIC=True, SMAGORINSKY_AH=True, SMAG_BI_CONST=0.06, AH_VEL_SCALE=0.01))
        visc = vertvisc_type(**{n: T(a) for n, a in va.items()})
    ref = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], dt, **okw)
    ref.bcs.dtbt = dt / 6.6
    dg = DeviceGrid(g)
    u, v, h, Tt, Ss = (T(d[k]) for k in ("u", "v", "h", "T", "S"))
    Z = lambda pos, k3=True: torch.zeros(g.shape3(pos) if k3 else g.shape2(pos), dtype=torch.float64, device="cuda")
    uh, vh, uhtr, vhtr, eta_av = Z(U), Z(V), Z(U), Z(V), Z(H, False)
    init, step = (M.initialize_dyn_split_RK2b, M.step_MOM_dyn_split_RK2b) if rk2b else (M.initialize_dyn_split_RK2, M.step_MOM_dyn_split_RK2)
    CS = init(u, v, h, uh, vh, dt, dg, USE_BT_CONT_TYPE=use_bt, coriolis=dict(bound_coriolis=True),
              barotropic=dict(BT_THICK_SCHEME="FROM_BT_CONT" if use_bt else "HARMONIC"), **hkw)
    CS.barotropic_CSp.st.dtbt = ref.bcs.dtbt
    tx, ty = T(taux), T(tauy)
    for n in range(3):
        ref.step(taux, tauy)
        step(u, v, h, (Tt, Ss), visc, None, dt, (tx, ty), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS)
        dg.sync()
        for name, a, b in (("u", u, ref.u), ("v", v, ref.v), ("h", h, ref.h), ("uh", uh, ref.uh), ("vh", vh, ref.vh), ("uhtr", uhtr, ref.uhtr),
                           ("eta_av", eta_av, ref.eta_av), ("eta", CS.eta, ref.arrs["eta"]), ("visc_rem_u", CS.visc_rem_u, ref.arrs["visc_rem_u"])):
            an = a.cpu().numpy()
            assert bits_equal(an, b), (n, name, float(np.abs(an - b).max()), np.argwhere(an != b)[:3])
    sj, si = g.csl(V)
    assert np.abs(ref.vh[:, sj, si][:, -1]).max() > 0
    dg.close()


@pytest.mark.gpu
def test_gpu_tracers_across_the_fold_match_oracle():
    import torch
    from mom6_amd.tracer_advect import DeviceGrid, advect_tracer, tracer_advect_init
    from mom6_amd.tracer_hor_diff import tracer_hor_diff_init, tracer_hordiff
    g, g2 = tp.grids(ni=70, nj=10, nk=4)
    d, _, _, _ = tp.states(g, g2, umax=0.05)
    dt = 3600.0
    uhtr = np.ascontiguousarray(d["uh"] * dt); vhtr = np.ascontiguousarray(d["vh"] * dt)
    ref = [d["T"].copy(), d["S"].copy()]
    orc.advect_tracer(g, d["h"], uhtr, vhtr, dt, dt, "PPM:H3", ref)
    orc.tracer_hordiff(g, d["h"], dt, ref, 500.0)
    dg = DeviceGrid(g)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    tr = [T(d["T"]), T(d["S"])]
    hh = T(d["h"])
    advect_tracer(hh, T(uhtr), T(vhtr), None, dt, dg, tracer_advect_init(dt, "PPM:H3"), tr)
    tracer_hordiff(hh, dt, None, None, None, dg, tracer_hor_diff_init(KHTR=500.0), tr)
    dg.sync()
    for m in range(2):
        assert bits_equal(tr[m].cpu().numpy(), ref[m]), m
    dg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("rk2b", [False, True], ids=["RK2", "RK2B"])
def test_gpu_folded_run_is_the_unfolded_run_at_benchmark_size(rk2b):
    """BASELINE configs[2]'s shape, 360 x 180 x 75, as the UNFOLDED image of a 360 x 90 x 75 tripolar grid: two viscous steps of the
    library on both (multi-block launches, the hipGraph of the barotropic subcycle, ~20 barotropic steps) agree bit for bit on
    the folded half -- the geometry of the fold at a size the oracle is not needed for."""
    import torch
    from mom6_amd import dynamics_split_rk2 as M
    from mom6_amd.tracer_advect import DeviceGrid
    from mom6_amd.vert_friction import vertvisc_type
    g, g2 = tp.grids(ni=360, nj=90, nk=75)
    d, d2, tau, tau2 = tp.states(g, g2)
    dt = 3600.0
    init, step = (M.initialize_dyn_split_RK2b, M.step_MOM_dyn_split_RK2b) if rk2b else (M.initialize_dyn_split_RK2, M.step_MOM_dyn_split_RK2)
    res = []
    for gg, dd, tt in ((g, d, tau), (g2, d2, tau2)):
        dg = DeviceGrid(gg)
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        u, v, h, Tt, Ss = (T(dd[k]) for k in ("u", "v", "h", "T", "S"))
        Z = lambda pos, k3=True: torch.zeros(gg.shape3(pos) if k3 else gg.shape2(pos), dtype=torch.float64, device="cuda")
        uh, vh, uhtr, vhtr, eta_av = Z(U), Z(V), Z(U), Z(V), Z(H, False)
        visc = vertvisc_type(Kv_bbl_u=T(1.0e-3 * gg.mask2dCu), Kv_bbl_v=T(1.0e-3 * gg.mask2dCv), bbl_thick_u=T(5.0 * gg.mask2dCu),
                             bbl_thick_v=T(5.0 * gg.mask2dCv))
        CS = init(u, v, h, uh, vh, dt, dg, coriolis=dict(bound_coriolis=True), vertvisc=dict(KV=1.0e-3, HBBL=10.0),
                  hor_visc=dict(BIHARMONIC=True, SMAGORINSKY_AH=True, SMAG_BI_CONST=0.06, AH_VEL_SCALE=0.01))
        tx, ty = T(tt[0]), T(tt[1])
        for n in range(2):
            step(u, v, h, (Tt, Ss), visc, None, dt, (tx, ty), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS, calc_dtbt=(n == 0))
        dg.sync()
        res.append(dict(u=u.cpu().numpy(), v=v.cpu().numpy(), h=h.cpu().numpy(), uhtr=uhtr.cpu().numpy(), eta=CS.eta.cpu().numpy(),
                        nstep=int(CS.barotropic_CSp.st.nstep_last), dtbt=float(CS.barotropic_CSp.st.dtbt)))
        dg.close()
        del u, v, h, Tt, Ss, uh, vh, uhtr, vhtr, CS, visc
        torch.cuda.empty_cache()
    a, b = res
    assert a["nstep"] == b["nstep"] and a["dtbt"] == b["dtbt"] and a["nstep"] >= 10, (a["nstep"], b["nstep"])
    for n, pos in (("h", H), ("u", U), ("v", V), ("uhtr", U), ("eta", H)):
        x = np.ascontiguousarray(interior(g, a[n], pos)); y = np.ascontiguousarray(interior(g, tp.folded(g, b[n], pos), pos))
        assert bits_equal(x, y), (n, float(np.abs(x - y).max()), np.argwhere(x != y)[:3])
    assert np.isfinite(a["u"]).all() and np.abs(a["u"]).max() > 1e-3
