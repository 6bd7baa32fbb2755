"""The library's own group pass over RCCL (mom6_amd/csrc/domain_rccl.hip), rehearsed on ONE GPU: a one-tile domain whose
re-entrant directions go through ncclSend / ncclRecv with the rank as its own neighbour (Domain(self_exchange=True)) instead
of the local wrap kernels.  Packing, the message order, the communication stream and its events, the reductions -- all
of the native path except a second device -- are exercised, and the results must equal the oracle's bit for bit.  (The
neighbour logic itself is covered by the two-rank gloo tests of tests/test_domains.py.)"""
import numpy as np
import pytest

import exact_synth as xs
from helpers import bits_equal
from mom6_amd import _abi
from oracle import orc

H, U, V = _abi.POS_H, _abi.POS_U, _abi.POS_V


def native_grid(g):
    import torch  # noqa: F401
    from mom6_amd.domains import Domain
    from mom6_amd.tracer_advect import DeviceGrid
    dom = Domain(g.ni, g.nj, (1, 1), 0, g.halo, g.reentrant_x, g.reentrant_y, self_exchange=True, tripolar_n=g.tripolar_n)
    tg = dom.tile_grid(g)
    assert not tg.reentrant_x and not tg.reentrant_y and tg.tripolar_n == g.tripolar_n
    dg = DeviceGrid(tg)
    dom.attach_native(dg)
    return dom, dg


@pytest.mark.gpu
@pytest.mark.parametrize("topo", [(True, False), (True, True), (False, True)])
def test_halo_update_through_rccl_self_exchange(topo):
    import torch
    g = xs.make_grid(37, 23, 3, reentrant_x=topo[0], reentrant_y=topo[1])
    dom, dg = native_grid(g)
    rng = np.random.default_rng(3)
    fields, poss, expect = [], [], []
    for pos in (H, U, V, _abi.POS_Q):
        for three_d in (True, False):
            shp = g.shape3(pos) if three_d else g.shape2(pos)
            a = rng.standard_normal(shp)
            sj, si = g.csl(pos)
            if topo[0] and pos in (U, _abi.POS_Q):
                a[..., sj, g.halo] = a[..., sj, g.halo + g.ni]
            if topo[1] and pos in (V, _abi.POS_Q):
                a[..., g.halo, si] = a[..., g.halo + g.nj, si]
            e = a.copy(); orc.halo_update(g, e, pos)
            fields.append(torch.from_numpy(a).cuda()); poss.append(pos); expect.append(e)
    dom.exchange_timing(True)
    dg.halo_update(fields, poss)
    dg.sync()
    ms, n = dom.exchange_timing(False)
    assert n == 1 and ms > 0.0
    for f, e, pos in zip(fields, expect, poss):
        assert bits_equal(f.cpu().numpy(), e), (topo, pos, f.dim())
    dg.close()


@pytest.mark.gpu
def test_model_steps_through_the_native_exchange():
    """initialize + two RK2 steps (both viscosities) + advect_tracer on a re-entrant grid, every group pass and both
    reductions going through RCCL: the oracle's bits"""
    import torch
    from mom6_amd.dynamics_split_rk2 import initialize_dyn_split_RK2, step_MOM_dyn_split_RK2
    from mom6_amd.tracer_advect import advect_tracer, tracer_advect_init
    from mom6_amd.vert_friction import vertvisc_type
    g = xs.make_grid(44, 40, 4, reentrant_x=True, reentrant_y=True, land_frac=0.2)
    d = xs.make_state(g, umax=0.1, terrain_following=True)
    taux, tauy = xs.wind_stress(g); bbl = xs.bbl_arrays(g)
    dt = 1800.0
    hv = dict(Ah_vel_scale=0.05, Smagorinsky_Ah=1, Smag_bi_const=0.06)
    ref = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], dt, vertvisc=orc.vertvisc_cs(g, Kv=1.0e-3, Hbbl=10.0),
                       visc=orc.vertvisc_type(**bbl), hor_visc=orc.hor_visc_cs(g, dt, **hv))
    dom, dg = native_grid(g)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    u, v, h, Tt, Ss = (T(d[k]) for k in ("u", "v", "h", "T", "S"))
    Z = lambda pos, k3=True: torch.zeros(g.shape3(pos) if k3 else g.shape2(pos), dtype=torch.float64, device="cuda")
    uh, vh, uhtr, vhtr, eta_av = Z(U), Z(V), Z(U), Z(V), Z(H, False)
    CS = initialize_dyn_split_RK2(u, v, h, uh, vh, dt, dg, coriolis=dict(bound_coriolis=True), vertvisc=dict(KV=1.0e-3, HBBL=10.0),
                                  hor_visc=dict(AH_VEL_SCALE=0.05, SMAGORINSKY_AH=True, SMAG_BI_CONST=0.06))
    visc = vertvisc_type(**{n: T(a) for n, a in bbl.items()})
    tx, ty = T(taux), T(tauy)
    for n in range(2):
        ref.step(taux, tauy, calc_dtbt=(n == 0))
        step_MOM_dyn_split_RK2(u, v, h, (Tt, Ss), visc, None, dt, (tx, ty), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS, calc_dtbt=(n == 0))
        dg.sync()
        assert CS.barotropic_CSp.st.dtbt == ref.bcs.dtbt
        for name, a, b in (("u", u, ref.u), ("v", v, ref.v), ("h", h, ref.h), ("uhtr", uhtr, ref.uhtr), ("eta_av", eta_av, ref.eta_av)):
            assert bits_equal(a.cpu().numpy(), b), (n, name)
    tr = [Tt.clone(), Ss.clone()]
    rtr = [ref.T.copy(), ref.S.copy()]
    st = advect_tracer(h, uhtr, vhtr, None, 2 * dt, dg, tracer_advect_init(dt, "PPM:H3"), tr)
    rst = orc.advect_tracer(g, ref.h, ref.uhtr, ref.vhtr, 2 * dt, dt, "PPM:H3", rtr)
    dg.sync()
    assert st.iterations == rst.iterations
    for a, b in zip(tr, rtr):
        assert bits_equal(a.cpu().numpy(), b)
    dg.close()


@pytest.mark.gpu
def test_tripolar_fold_in_the_native_exchange():
    """TRIPOLAR_N in the library's own group pass: x through ncclSend / ncclRecv (self-exchange), then the fold on the
    communication stream; halo updates (vector, scalar pair, every staggering) and three viscous RK2 steps == oracle"""
    import torch
    import tripolar as tp
    from mom6_amd.dynamics_split_rk2 import initialize_dyn_split_RK2, step_MOM_dyn_split_RK2
    g, g2 = tp.grids(ni=70, nj=10, nk=3)
    dom, dg = native_grid(g)
    rng = np.random.default_rng(4)
    SP = _abi.PASS_SCALAR_PAIR
    fields, poss, want = [], [], []
    for pf in (H, U, V, _abi.POS_Q, U | SP, V | SP):
        a = rng.standard_normal(g.shape3(pf & 3))
        w = a.copy(); orc.halo_update(g, w, pf)
        fields.append(torch.from_numpy(a).cuda()); poss.append(pf); want.append(w)
    dg.halo_update(fields, poss)
    dg.sync()
    for f, w, pf in zip(fields, want, poss):
        assert bits_equal(f.cpu().numpy(), w), pf
    d, _, (taux, tauy), _ = tp.states(g, g2)
    dt = 900.0
    ref = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], dt)
    ref.bcs.dtbt = dt / 6.6
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    u, v, h, Tt, Ss = (T(d[k]) for k in ("u", "v", "h", "T", "S"))
    Z = lambda pos, k3=True: torch.zeros(g.shape3(pos) if k3 else g.shape2(pos), dtype=torch.float64, device="cuda")
    uh, vh, uhtr, vhtr, eta_av = Z(U), Z(V), Z(U), Z(V), Z(H, False)
    CS = initialize_dyn_split_RK2(u, v, h, uh, vh, dt, dg, coriolis=dict(bound_coriolis=True))
    CS.barotropic_CSp.st.dtbt = ref.bcs.dtbt
    tx, ty = T(taux), T(tauy)
    for n in range(3):
        ref.step(taux, tauy)
        step_MOM_dyn_split_RK2(u, v, h, (Tt, Ss), None, None, dt, (tx, ty), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS)
    dg.sync()
    for name, a, b in (("u", u, ref.u), ("v", v, ref.v), ("h", h, ref.h), ("uhtr", uhtr, ref.uhtr), ("eta", CS.eta, ref.arrs["eta"])):
        assert bits_equal(a.cpu().numpy(), b), name
    dg.close()


def _three_viscous_steps(g, dg, poison=None, continuity=None):
    """three steps of step_MOM_dyn_split_RK2 with vertical and horizontal viscosity on the device grid dg against the oracle on g"""
    import torch
    from mom6_amd.dynamics_split_rk2 import initialize_dyn_split_RK2, step_MOM_dyn_split_RK2
    from mom6_amd.vert_friction import vertvisc_type
    d = xs.make_state(g, umax=0.1, terrain_following=True)
    taux, tauy = xs.wind_stress(g); bbl = xs.bbl_arrays(g)
    dt = 1800.0
    hv = dict(Laplacian=1, Kh_vel_scale=0.01, Ah_vel_scale=0.05, Smagorinsky_Ah=1, Smag_bi_const=0.06)
    hvn = dict(LAPLACIAN=1, KH_VEL_SCALE=0.01, AH_VEL_SCALE=0.05, SMAGORINSKY_AH=1, SMAG_BI_CONST=0.06)
    ref = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], dt, vertvisc=orc.vertvisc_cs(g, Kv=1.0e-3, Hbbl=10.0),
                       visc=orc.vertvisc_type(**bbl), hor_visc=orc.hor_visc_cs(g, dt, **hv), continuity=continuity)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    u, v, h, Tt, Ss = (T(d[k]) for k in ("u", "v", "h", "T", "S"))
    Z = lambda pos, k3=True: torch.zeros(g.shape3(pos) if k3 else g.shape2(pos), dtype=torch.float64, device="cuda")
    uh, vh, uhtr, vhtr, eta_av = Z(U), Z(V), Z(U), Z(V), Z(H, False)
    CS = initialize_dyn_split_RK2(u, v, h, uh, vh, dt, dg, coriolis=dict(bound_coriolis=True), vertvisc=dict(KV=1.0e-3, HBBL=10.0), hor_visc=hvn,
                                  continuity=continuity)
    if poison is not None:
        dg.debug_poison_passes(**poison)
    visc = vertvisc_type(**{n: T(a) for n, a in bbl.items()})
    tx, ty = T(taux), T(tauy)
    for n in range(3):
        ref.step(taux, tauy, calc_dtbt=(n == 0))
        step_MOM_dyn_split_RK2(u, v, h, (Tt, Ss), visc, None, dt, (tx, ty), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS, calc_dtbt=(n == 0))
        dg.sync()
        for name, a, b in (("u", u, ref.u), ("v", v, ref.v), ("h", h, ref.h), ("uh", uh, ref.uh), ("vh", vh, ref.vh), ("uhtr", uhtr, ref.uhtr),
                           ("vhtr", vhtr, ref.vhtr), ("eta_av", eta_av, ref.eta_av), ("h_av", CS.h_av, ref.arrs["h_av"]),
                           ("CAu_pred", CS.CAu_pred, ref.arrs["CAu_pred"]), ("diffu", CS.diffu, ref.arrs["diffu"])):
            an = a.cpu().numpy()
            assert not np.isnan(an).any(), (n, name, "a halo was read while its pass was in flight")
            assert bits_equal(an, b), (n, name, float(np.abs(an - b).max()))


@pytest.mark.gpu
def test_poisoned_passes_hold_nans_until_they_complete():
    """the debugging aid itself: between start_group_pass and complete_group_pass every halo the pass fills (and nothing else) is NaN;
    the completed pass equals the plain update"""
    import torch
    from mom6_amd.tracer_advect import DeviceGrid
    g = xs.make_grid(20, 14, 2, reentrant_x=True, reentrant_y=True)
    dg = DeviceGrid(g)
    dg.debug_poison_passes(True)
    rng = np.random.default_rng(8)
    for pos in (H, U, V):
        a = rng.standard_normal(g.shape3(pos))
        w = a.copy(); orc.halo_update(g, w, pos)
        f = torch.from_numpy(a).cuda()
        dg.start_group_pass([f], [pos]); dg.sync()
        mid = f.cpu().numpy()
        sj, si = g.csl(pos)
        assert not np.isnan(mid[:, sj, :]).any()              # x is final when start returns (the tile spans x)
        assert np.isnan(mid[:, :sj.start, :]).all() and np.isnan(mid[:, sj.stop:, :]).all()
        dg.complete_group_pass(); dg.sync()
        assert bits_equal(f.cpu().numpy(), w), pos
    dg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["poison", "poison+split", "split"])
def test_nothing_reads_a_halo_while_its_pass_is_in_flight(mode):
    """The step starts its group passes where the reference does (MOM_dynamics_split_RK2.F90:541, :608, :741, :763, :991, :1018) and
    completes them as late as the next reader of a halo allows, with the rows a halo width inside the tile of h_av, horizontal
    viscosity, CorAdCalc and the accumulations of uhtr / vhtr / h_av, and the continuity's zonal pass of the tile's own rows with the
    meridional faces and cells whose stencil stays inside them (the three continuity calls in two phases), enqueued BEFORE the completion.  Proof that none of this reads a
    halo too early, independent of any timing: with mom6hip_debug_poison_passes every halo a pass will fill is NaN from its start to
    its completion -- three viscous steps on a doubly re-entrant tile are still the oracle's bits, without a NaN"""
    from mom6_amd.tracer_advect import DeviceGrid
    # (no land: with land along the tile's edges nothing depends on the halos -- a test of nothing)
    g = xs.make_grid(44, 40, 4, land_frac=0.0, reentrant_x=True, reentrant_y=True)
    dg = DeviceGrid(g)
    _three_viscous_steps(g, dg, poison=dict(poison="poison" in mode, split_rows="split" in mode))
    # per step: two row-split groups (after pass_hp_uv, after pass_h + pass_av_uvh) and the three continuity calls in two phases
    assert dg.overlap_stats() == (3 * 2, 3 * 3, 0, 0)
    dg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("opts", [dict(simple_2nd=True), dict(upwind_1st=True), dict(monotonic=True)], ids=["simple_2nd", "upwind_1st", "monotonic"])
def test_two_phase_continuity_with_the_other_stencils(opts):
    """the two phases follow the continuity's stencil (3 rows for PPM, 2 for SIMPLE_2ND_PPM_CONTINUITY, 1 for UPWIND_1ST_CONTINUITY): with
    every halo in flight poisoned, three viscous steps are the oracle's bits for each of them"""
    from mom6_amd.tracer_advect import DeviceGrid
    g = xs.make_grid(44, 40, 4, land_frac=0.0, reentrant_x=True, reentrant_y=True)
    dg = DeviceGrid(g)
    _three_viscous_steps(g, dg, poison=dict(poison=True, split_rows=True), continuity=opts)
    assert dg.overlap_stats() == (3 * 2, 3 * 3, 0, 0)
    dg.close()


@pytest.mark.gpu
def test_y_first_continuity_waits_for_its_pass():
    """G%first_direction = 1: the meridional pass comes first and needs the halo rows at once, so the continuity is not split
    around its pass (continuity_around_pass falls back to the completed pass); everything else still is"""
    from mom6_amd.tracer_advect import DeviceGrid
    g = xs.make_grid(44, 40, 4, land_frac=0.0, reentrant_x=True, reentrant_y=True, first_direction=1)
    dg = DeviceGrid(g)
    _three_viscous_steps(g, dg, poison=dict(poison=True, split_rows=True))
    assert dg.overlap_stats() == (3 * 2, 0, 3 * 3, 0)
    dg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("poison", [False, True])
def test_row_split_around_the_native_exchange(poison):
    """the same through the library's own exchange: x is the tile's wrap, y goes through ncclSend / ncclRecv on the communication
    stream (this rank as its own northern and southern neighbour: a 1 x N layout seen from one tile), inner rows before the
    completion, edge bands after it"""
    import torch  # noqa: F401
    from mom6_amd.domains import Domain
    from mom6_amd.tracer_advect import DeviceGrid
    g = xs.make_grid(44, 40, 4, land_frac=0.0, reentrant_x=True, reentrant_y=True)
    dom = Domain(g.ni, g.nj, (1, 1), 0, g.halo, True, True, self_exchange="y")
    tg = dom.tile_grid(g)
    assert tg.reentrant_x and not tg.reentrant_y
    dg = DeviceGrid(tg)
    dom.attach_native(dg)
    _three_viscous_steps(g, dg, poison=dict(poison=poison, split_rows=False))
    assert dg.overlap_stats() == (3 * 2, 3 * 3, 0, 0)
    dg.close()


@pytest.mark.gpu
def test_rk2b_continuity_in_two_phases_under_poisoned_halos():
    """SPLIT_RK2B: the two continuity calls that follow pass_visc_rem + pass_uvp / pass_uv (MOM_dynamics_split_RK2b.F90:748-758, :967-979) run
    in two phases around their passes as in the RK2 stepping (the last one with du_cor / dv_cor): with every halo in flight poisoned three
    viscous steps are the oracle's bits"""
    import torch
    from mom6_amd.dynamics_split_rk2 import initialize_dyn_split_RK2b, step_MOM_dyn_split_RK2b
    from mom6_amd.tracer_advect import DeviceGrid
    from mom6_amd.vert_friction import vertvisc_type
    g = xs.make_grid(44, 40, 4, land_frac=0.0, reentrant_x=True, reentrant_y=True)
    d = xs.make_state(g, umax=0.1, terrain_following=True)
    taux, tauy = xs.wind_stress(g); bbl = xs.bbl_arrays(g)
    dt = 1800.0
    ref = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], dt, rk2b=True, vertvisc=orc.vertvisc_cs(g, Kv=1.0e-3, Hbbl=10.0),
                       visc=orc.vertvisc_type(**bbl), hor_visc=orc.hor_visc_cs(g, dt, biharmonic=True, Smagorinsky_Ah=True, Smag_bi_const=0.06, Ah_vel_scale=0.01))
    dg = DeviceGrid(g)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    u, v, h, Tt, Ss = (T(d[k]) for k in ("u", "v", "h", "T", "S"))
    Z = lambda pos, k3=True: torch.zeros(g.shape3(pos) if k3 else g.shape2(pos), dtype=torch.float64, device="cuda")
    uh, vh, uhtr, vhtr, eta_av = Z(U), Z(V), Z(U), Z(V), Z(H, False)
    CS = initialize_dyn_split_RK2b(u, v, h, uh, vh, dt, dg, coriolis=dict(bound_coriolis=True), vertvisc=dict(KV=1.0e-3, HBBL=10.0),
                                   hor_visc=dict(BIHARMONIC=True, SMAGORINSKY_AH=True, SMAG_BI_CONST=0.06, AH_VEL_SCALE=0.01))
    dg.debug_poison_passes(poison=True, split_rows=True)
    visc = vertvisc_type(**{n: T(a) for n, a in bbl.items()})
    tx, ty = T(taux), T(tauy)
    for n in range(3):
        ref.step(taux, tauy, calc_dtbt=(n == 0))
        step_MOM_dyn_split_RK2b(u, v, h, (Tt, Ss), visc, None, dt, (tx, ty), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS, calc_dtbt=(n == 0))
        dg.sync()
        for name, a, b in (("u_av", u, ref.u), ("v_av", v, ref.v), ("h", h, ref.h), ("uh", uh, ref.uh), ("vh", vh, ref.vh), ("uhtr", uhtr, ref.uhtr),
                           ("du_av_inst", CS.du_av_inst, ref.arrs["du_av_inst"]), ("h_av", CS.h_av, ref.arrs["h_av"])):
            an = a.cpu().numpy()
            assert not np.isnan(an).any(), (n, name, "a halo was read while its pass was in flight")
            assert bits_equal(an, b), (n, name, float(np.abs(an - b).max()))
    assert dg.overlap_stats()[1] == 3 * 2
    dg.close()
