"""Worker functions for the multi-process (torch.distributed) tests; importable by spawned processes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def halo_worker(rank, world, port, layout, rx, ry, out_dir, dims=(23, 17), tripolar=False):
    """Every rank fills the halos of its tile of random global fields through Domain.pass_var; the result must
    equal the same window of the one-tile halo update (oracle/domains.c), bit for bit."""
    import numpy as np
    import torch
    from mom6_amd import _abi, synth
    from mom6_amd.domains import Domain
    from oracle import orc
    dist = _init(rank, world, port)
    try:
        (NI, NJ), NK, halo = dims, 3, 4
        gg = synth.make_grid(NI, NJ, NK, halo=halo, reentrant_x=rx, reentrant_y=ry)
        gg.tripolar_n = bool(tripolar); gg._struct = None
        dom = Domain(NI, NJ, layout, rank, halo, rx, ry, tripolar_n=tripolar)
        rng = np.random.default_rng(5)
        ok = True
        fields, poss, expect = [], [], []
        SP = _abi.PASS_SCALAR_PAIR
        for pos_flags in (_abi.POS_H, _abi.POS_U, _abi.POS_V, _abi.POS_Q) + ((_abi.POS_U | SP, _abi.POS_V | SP) if tripolar else ()):
            pos = pos_flags & 3
            for three_d in (True, False):
                shp = gg.shape3(pos) if three_d else gg.shape2(pos)
                G = np.full(shp, -777.0)
                sj, si = gg.csl(pos)
                G[..., sj, si] = rng.standard_normal(G[..., sj, si].shape)
                if rx:      # the duplicated west/south face of a re-entrant one-tile domain holds the same value
                    if pos in (_abi.POS_U, _abi.POS_Q):
                        G[..., sj, halo] = G[..., sj, halo + NI]
                if ry and pos in (_abi.POS_V, _abi.POS_Q):
                    G[..., halo, si] = G[..., halo + NJ, si]
                after = G.copy()
                orc.halo_update(gg, after, pos_flags)
                tile = dom.cut(G, pos).copy()
                # poison the tile's halos
                h = halo
                xs = 1 if pos in (_abi.POS_U, _abi.POS_Q) else 0
                ys = 1 if pos in (_abi.POS_V, _abi.POS_Q) else 0
                keep = tile[..., h:h + dom.nj + ys, h:h + dom.ni + xs].copy()
                tile[:] = -777.0
                tile[..., h:h + dom.nj + ys, h:h + dom.ni + xs] = keep
                fields.append(torch.from_numpy(tile)); poss.append(pos_flags); expect.append(dom.cut(after, pos))
        dom.pass_var(fields, poss)
        for f, e, pos in zip(fields, expect, poss):
            a = f.numpy()
            # corners beyond two closed edges etc. stay poisoned on both sides; compare everything
            if not np.array_equal(a.view(np.uint64), np.ascontiguousarray(e).view(np.uint64)):
                ok = False
                bad = np.argwhere(a != e)
                print(f"rank {rank} layout {layout} pos {pos} ndim {a.ndim}: {len(bad)} mismatches, first {bad[:3]}", flush=True)
        t = torch.tensor([7 * (rank + 1), 1], dtype=torch.int32)
        dom.sum_across_PEs(t)
        ok = ok and int(t[0]) == 7 * sum(range(1, world + 1)) and int(t[1]) == world
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("1" if ok else "0")
    finally:
        dist.destroy_process_group()


def advect_layout_worker(rank, world, port, layout, scheme, out_dir):
    """test.layout for advect_tracer on the GPU: the tiles of an (npi x npj) layout, exchanging halos through the
    domain callbacks, must reproduce the one-tile run bit for bit."""
    import numpy as np
    import torch
    from mom6_amd import _abi, synth
    from mom6_amd.domains import Domain
    from mom6_amd.tracer_advect import DeviceGrid, advect_tracer, tracer_advect_init
    dist = _init(rank, world, port)
    try:
        NI, NJ, NK, halo = 70, 40, 3, 4
        gg = synth.make_grid(NI, NJ, NK, halo=halo, reentrant_x=True, reentrant_y=False, seed=77)
        st = synth.make_advection_state(gg, ntr=3, seed=9, hot_frac=0.02, cfl=0.1)
        CS = tracer_advect_init(900.0, scheme)
        dom = Domain(NI, NJ, layout, rank, halo, True, False)
        tg = dom.tile_grid(gg)
        dg = DeviceGrid(tg)
        dg.set_domain(dom)
        cut = lambda a, pos: dom.cut(a, pos).cuda()
        tr = [cut(t, _abi.POS_H) for t in st["tr"]]
        stats = advect_tracer(cut(st["h_end"], _abi.POS_H), cut(st["uhtr"], _abi.POS_U), cut(st["vhtr"], _abi.POS_V), None,
                              3600.0, dg, CS, tr)
        dg.sync()
        h = halo
        res = [t.cpu().numpy()[:, h:h + dom.nj, h:h + dom.ni] for t in tr]
        np.savez(os.path.join(out_dir, f"tile{rank}.npz"), *res, ij=np.array([dom.i0, dom.j0, dom.ni, dom.nj, stats.iterations]))
        dg.close()
        if rank == 0:      # the one-tile answer
            dg1 = DeviceGrid(gg)
            tr1 = [t.clone().cuda() for t in st["tr"]]
            s1 = advect_tracer(st["h_end"].cuda(), st["uhtr"].cuda(), st["vhtr"].cuda(), None, 3600.0, dg1, CS, tr1)
            dg1.sync()
            np.savez(os.path.join(out_dir, "global.npz"), *[t.cpu().numpy()[:, h:h + NJ, h:h + NI] for t in tr1],
                     it=np.array([s1.iterations]))
            dg1.close()
    finally:
        dist.destroy_process_group()


def hordiff_layout_worker(rank, world, port, layout, out_dir, neutral=False):
    """test.layout for tracer_hordiff on the GPU (several iterations: a group pass and the max across PEs of the diffusive CFL
    number inside the call), along layers or with USE_NEUTRAL_DIFFUSION: the tiles must reproduce the one-tile run bit for bit."""
    import numpy as np
    import torch
    from mom6_amd import _abi, synth
    from mom6_amd.domains import Domain
    from mom6_amd.tracer_advect import DeviceGrid
    from mom6_amd.pressure_force import EOS_init
    from mom6_amd.tracer_hor_diff import tracer_hor_diff_init, tracer_hordiff
    dist = _init(rank, world, port)
    try:
        NI, NJ, NK, halo = 70, 40, 3, 4
        eos = EOS_init("WRIGHT")
        tvof = lambda t: dict(T=t[0], S=t[1], eqn_of_state=eos) if neutral else None
        gg = synth.make_grid(NI, NJ, NK, halo=halo, reentrant_x=True, reentrant_y=False, seed=78)
        d = synth.make_dynamics_state(gg, seed=3, umax=0.1, eta_amp=0.2)
        trs = [d["T"], d["S"]]
        CS = tracer_hor_diff_init(KHTR=3.0e7, CHECK_DIFFUSIVE_CFL=True, USE_NEUTRAL_DIFFUSION=neutral)
        dom = Domain(NI, NJ, layout, rank, halo, True, False)
        dg = DeviceGrid(dom.tile_grid(gg))
        dg.set_domain(dom)
        cut = lambda a, pos: dom.cut(a, pos).cuda()
        tr = [cut(t, _abi.POS_H) for t in trs]
        st = tracer_hordiff(cut(d["h"], _abi.POS_H), 3600.0, None, None, None, dg, CS, tr, tv=tvof(tr))
        dg.sync()
        h = halo
        res = [t.cpu().numpy()[:, h:h + dom.nj, h:h + dom.ni] for t in tr]
        np.savez(os.path.join(out_dir, f"tile{rank}.npz"), *res, ij=np.array([dom.i0, dom.j0, dom.ni, dom.nj, st.num_itts]), cfl=np.array([st.max_CFL]))
        dg.close()
        if rank == 0:      # the one-tile answer
            dg1 = DeviceGrid(gg)
            tr1 = [t.clone().cuda() for t in trs]
            s1 = tracer_hordiff(d["h"].cuda(), 3600.0, None, None, None, dg1,
                                tracer_hor_diff_init(KHTR=3.0e7, CHECK_DIFFUSIVE_CFL=True, USE_NEUTRAL_DIFFUSION=neutral), tr1, tv=tvof(tr1))
            dg1.sync()
            np.savez(os.path.join(out_dir, "global.npz"), *[t.cpu().numpy()[:, h:h + NJ, h:h + NI] for t in tr1],
                     it=np.array([s1.num_itts]), cfl=np.array([s1.max_CFL]))
            dg1.close()
    finally:
        dist.destroy_process_group()


def lateral_layout_worker(rank, world, port, layout, out_dir):
    """test.layout for thickness_diffuse and mixedlayer_restrat on the GPU (they read one halo point of h, T, S and of the 2-D fields
    and exchange nothing themselves): the tiles must reproduce the one-tile run bit for bit."""
    import numpy as np
    import torch
    from mom6_amd import _abi, synth
    from mom6_amd.domains import Domain
    from mom6_amd.mixedlayer_restrat import mixedlayer_restrat, mixedlayer_restrat_init
    from mom6_amd.pressure_force import EOS_init
    from mom6_amd.thickness_diffuse import thickness_diffuse, thickness_diffuse_init
    from mom6_amd.tracer_advect import DeviceGrid
    dist = _init(rank, world, port)
    try:
        NI, NJ, NK, halo = 60, 36, 6, 4
        H, U, V = _abi.POS_H, _abi.POS_U, _abi.POS_V
        gg = synth.make_grid(NI, NJ, NK, halo=halo, reentrant_x=True, reentrant_y=False, seed=31)
        d = synth.make_dynamics_state(gg, seed=4, umax=0.1, eta_amp=0.2)
        rng = np.random.default_rng(9)
        two = {n: torch.from_numpy(np.ascontiguousarray(a * gg.mask2dT))
               for n, a in (("ustar", 2e-3 + 0.02 * rng.random(gg.shape2(H))), ("h_MLD", 15.0 + 100.0 * rng.random(gg.shape2(H))),
                            ("Kh", 800.0 * rng.random(gg.shape2(H))))}
        for a in two.values():      # valid halos, as the operators expect of their inputs
            a[:, :halo] = a[:, NI:NI + halo]; a[:, NI + halo:] = a[:, halo:2 * halo]
        eos = EOS_init("WRIGHT")

        def run(dg, cut):
            h, T, S = cut(d["h"], H), cut(d["T"], H), cut(d["S"], H)
            uhtr, vhtr = torch.zeros_like(cut(d["u"], U)), torch.zeros_like(cut(d["v"], V))
            td = thickness_diffuse_init(dg, THICKNESSDIFFUSE=True, KHTH=1.0, KHTH_MAX=900.0)
            thickness_diffuse(h, uhtr, vhtr, (T, S, eos), 3600.0, dg, dict(Kh=cut(two["Kh"], H)), None, None, td)
            dg.start_group_pass([h], [H]); dg.complete_group_pass()      # call pass_var(h, G%Domain), MOM.F90:1179
            mle = mixedlayer_restrat_init(dg, FOX_KEMPER_ML_RESTRAT_COEF=20.0, MLE_USE_PBL_MLD=True, MLE_MLD_DECAY_TIME=86400.0,
                                          MLD_filtered=torch.zeros_like(cut(two["ustar"], H)))
            mixedlayer_restrat(h, uhtr, vhtr, (T, S, eos), dict(ustar=cut(two["ustar"], H)), 3600.0, None, cut(two["h_MLD"], H), None, None, dg, mle)
            dg.sync()
            return h.cpu().numpy(), uhtr.cpu().numpy(), vhtr.cpu().numpy()
        dom = Domain(NI, NJ, layout, rank, halo, True, False)
        dg = DeviceGrid(dom.tile_grid(gg))
        dg.set_domain(dom)
        h, uh, vh = run(dg, lambda a, pos: dom.cut(a, pos).cuda())
        q = halo
        np.savez(os.path.join(out_dir, f"tile{rank}.npz"), h=h[:, q:q + dom.nj, q:q + dom.ni], uh=uh[:, q:q + dom.nj, q:q + dom.ni + 1],
                 vh=vh[:, q:q + dom.nj + 1, q:q + dom.ni], ij=np.array([dom.i0, dom.j0, dom.ni, dom.nj]))
        dg.close()
        if rank == 0:      # the one-tile answer
            dg1 = DeviceGrid(gg)
            h, uh, vh = run(dg1, lambda a, pos: a.clone().cuda())
            np.savez(os.path.join(out_dir, "global.npz"), h=h[:, q:q + NJ, q:q + NI], uh=uh[:, q:q + NJ, q:q + NI + 1], vh=vh[:, q:q + NJ + 1, q:q + NI])
            dg1.close()
    finally:
        dist.destroy_process_group()


def btstep_layout_worker(rank, world, port, layout, topo, out_dir):
    """test.layout for btstep on the GPU: each tile runs barotropic_init / btcalc / bt_mass_source / btstep with the
    group passes going through the domain callbacks; the compute-domain results must equal the one-tile oracle run."""
    import numpy as np
    import torch
    from helpers import barotropic_case
    from mom6_amd import _abi
    from mom6_amd.barotropic import barotropic_init, bt_mass_source, btcalc, btstep, set_dtbt
    from mom6_amd.continuity import BT_cont_type
    from mom6_amd.domains import Domain
    from mom6_amd.tracer_advect import DeviceGrid
    from oracle import orc
    dist = _init(rank, world, port)
    try:
        gg, cs_o, case, keep = barotropic_case(orc, ni=36, nj=24, nk=3, seed=21, reentrant_x=topo[0], reentrant_y=topo[1])
        dom = Domain(gg.ni, gg.nj, layout, rank, gg.halo, topo[0], topo[1])
        tg = dom.tile_grid(gg)
        dg = DeviceGrid(tg)
        dg.set_domain(dom)
        H, U, V = _abi.POS_H, _abi.POS_U, _abi.POS_V
        pos = dict(U_in=U, V_in=V, eta_in=H, bc_accel_u=U, bc_accel_v=V, taux=U, tauy=V, pbce=H, eta_PF_in=H, U_Cor=U, V_Cor=V,
                   visc_rem_u=U, visc_rem_v=V, uh0=U, vh0=V, u_uh0=U, v_vh0=V)
        c = {k: torch.from_numpy(dom.cut(case[k], p)).cuda() for k, p in pos.items()}
        bt_arrs = {}
        for n, a in keep["bt_arrs"].items():
            p = U if (n in _abi.BT_CONT_U or n == "h_u") else V
            bt_arrs[n] = torch.from_numpy(dom.cut(a, p)).cuda()
        BT = BT_cont_type(**bt_arrs)
        CS = barotropic_init(dg, BT_THICK_SCHEME="FROM_BT_CONT")
        h = torch.from_numpy(dom.cut(keep["h"], H)).cuda()
        btcalc(h, dg, CS, bt_arrs["h_u"], bt_arrs["h_v"])
        bt_mass_source(h, c["eta_in"], True, dg, CS)
        dtbt_max = set_dtbt(dg, CS, pbce=c["pbce"], BT_cont=BT)          # min_across_PEs inside
        CS.st.dtbt = cs_o.dtbt
        Z = lambda p, k3: torch.zeros(tg.shape3(p) if k3 else tg.shape2(p), dtype=torch.float64, device="cuda")
        out = dict(accel_layer_u=Z(U, True), accel_layer_v=Z(V, True), eta_out=Z(H, False), uhbtav=Z(U, False), vhbtav=Z(V, False),
                   etaav=Z(H, False))
        btstep(c["U_in"], c["V_in"], c["eta_in"], case["dt"], c["bc_accel_u"], c["bc_accel_v"], (c["taux"], c["tauy"]), c["pbce"],
               c["eta_PF_in"], c["U_Cor"], c["V_Cor"], out["accel_layer_u"], out["accel_layer_v"], out["eta_out"], out["uhbtav"],
               out["vhbtav"], dg, CS, c["visc_rem_u"], c["visc_rem_v"], BT_cont=BT, uh0=c["uh0"], vh0=c["vh0"], u_uh0=c["u_uh0"],
               v_vh0=c["v_vh0"], etaav=out["etaav"])
        dg.sync()
        np.savez(os.path.join(out_dir, f"bt_tile{rank}.npz"), ij=np.array([dom.i0, dom.j0, dom.ni, dom.nj]), dtbt_max=dtbt_max,
                 **{k: v.cpu().numpy() for k, v in out.items()})
        dg.close()
        if rank == 0:
            orc.set_dtbt(gg, cs_o, pbce=case["pbce"], bt_cont=case["bt_cont"])
            dm = cs_o.dtbt_max
            cs_o.dtbt = CS.st.dtbt
            ref = orc.btstep(gg, cs_o, **{k: case[k] for k in pos}, dt=case["dt"], bt_cont=case["bt_cont"], want_etaav=True)
            np.savez(os.path.join(out_dir, "bt_global.npz"), dtbt_max=dm, **ref)
    finally:
        dist.destroy_process_group()


def rk2_layout_worker(rank, world, port, layout, topo, out_dir):
    """test.layout for the whole split RK2 step on the GPU: two tiles, group passes through the packed exchange."""
    import numpy as np
    import torch
    from mom6_amd import _abi
    from mom6_amd.domains import Domain
    from mom6_amd.dynamics_split_rk2 import initialize_dyn_split_RK2, step_MOM_dyn_split_RK2
    from mom6_amd.tracer_advect import DeviceGrid
    from oracle import orc
    from test_dyn_split_rk2 import make_case
    dist = _init(rank, world, port)
    try:
        tripolar = len(topo) > 2 and topo[2]
        if tripolar:      # TRIPOLAR_N: the tile on the fold is its own northern neighbour
            import tripolar as tp
            gg, _ = tp.grids(ni=32, nj=24, nk=3, seed=13)
            d, _, (taux, tauy), _ = tp.states(gg, _)
        else:
            gg, d, taux, tauy = make_case(ni=32, nj=24, nk=3, seed=13, reentrant_x=topo[0], reentrant_y=topo[1])
        dt = 1800.0
        dom = Domain(gg.ni, gg.nj, layout, rank, gg.halo, topo[0], topo[1], tripolar_n=tripolar)
        tg = dom.tile_grid(gg)
        dg = DeviceGrid(tg)
        dg.set_domain(dom)
        H, U, V = _abi.POS_H, _abi.POS_U, _abi.POS_V
        T = lambda a, p: torch.from_numpy(dom.cut(a, p)).cuda()
        u, v, h, Tt, Ss = T(d["u"], U), T(d["v"], V), T(d["h"], H), T(d["T"], H), T(d["S"], H)
        Z = lambda p, k3=True: torch.zeros(tg.shape3(p) if k3 else tg.shape2(p), dtype=torch.float64, device="cuda")
        uh, vh, uhtr, vhtr, eta_av = Z(U), Z(V), Z(U), Z(V), Z(H, False)
        # the step runs with the library's vertical viscosity: visc%Kv_bbl / bbl_thick are cut per tile like any field
        from mom6_amd.vert_friction import vertvisc_type
        from test_dyn_split_rk2 import _visc_arrays
        va = _visc_arrays(gg)
        pos_of = dict(Kv_bbl_u=U, Kv_bbl_v=V, bbl_thick_u=U, bbl_thick_v=V)
        visc = vertvisc_type(**{n: T(a, pos_of[n]) for n, a in va.items()})
        CS = initialize_dyn_split_RK2(u, v, h, uh, vh, dt, dg, coriolis=dict(bound_coriolis=True), vertvisc=dict(KV=1.0e-3, HBBL=10.0))
        tx, ty = T(taux, U), T(tauy, V)
        for n in range(2):
            step_MOM_dyn_split_RK2(u, v, h, (Tt, Ss), visc, None, dt, (tx, ty), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS,
                                   calc_dtbt=(n == 0))
        dg.sync()
        np.savez(os.path.join(out_dir, f"rk2_tile{rank}.npz"), ij=np.array([dom.i0, dom.j0, dom.ni, dom.nj]), u=u.cpu().numpy(),
                 v=v.cpu().numpy(), h=h.cpu().numpy(), eta=CS.eta.cpu().numpy(), uhtr=uhtr.cpu().numpy(), dtbt=CS.barotropic_CSp.st.dtbt)
        dg.close()
        if rank == 0:
            ref = orc.DynState(gg, d["u"], d["v"], d["h"], d["T"], d["S"], dt, vertvisc=orc.vertvisc_cs(gg, Kv=1.0e-3, Hbbl=10.0),
                               visc=orc.vertvisc_type(**va))
            for n in range(2):
                ref.step(taux, tauy, calc_dtbt=(n == 0))
            np.savez(os.path.join(out_dir, "rk2_global.npz"), u=ref.u, v=ref.v, h=ref.h, eta=ref.arrs["eta"], uhtr=ref.uhtr, dtbt=ref.bcs.dtbt)
    finally:
        dist.destroy_process_group()


def _obc_data_from_fields(OBC, fu, fv, f2u, f2v):
    """external data of the specified and Flather segments sampled from fields on the grid of the OBC (global or a tile's cut of the same fields):
    the same numbers whatever the layout"""
    for s in OBC.segment:
        if not s.on_pe:
            continue
        hi = s.HI
        if s.is_E_or_W:
            sl3 = (slice(None), slice(hi["jsd"] - 1, hi["jed"]), slice(hi["IsdB"], hi["IsdB"] + 1)); f3, f2 = fu, f2u
        else:
            sl3 = (slice(None), slice(hi["JsdB"], hi["JsdB"] + 1), slice(hi["isd"] - 1, hi["ied"])); f3, f2 = fv, f2v
        if s.specified:
            s.normal_vel[:] = 0.05 * f3[sl3]; s.normal_trans[:] = s.normal_vel * 9.0e5
        if s.Flather:
            s.normal_vel_bt[:] = 0.02 * f2[sl3[1:]]; s.SSH[:] = 0.05 * f2[sl3[1:]] ** 2


def rk2_obc_layout_worker(rank, world, port, layout, segs, viscous, rk2b, out_dir):
    """the split RK2 / RK2B step with an associated OBC on two tiles: every tile places the segments with its own offsets, as open_boundary_config
    does on a PE, and holds its part of the segments' data"""
    import numpy as np
    import torch
    from mom6_amd import _abi
    from mom6_amd.domains import Domain
    from mom6_amd.dynamics_split_rk2 import initialize_dyn_split_RK2, step_MOM_dyn_split_RK2
    if rk2b:
        from mom6_amd.dynamics_split_rk2 import initialize_dyn_split_RK2b as initialize_dyn_split_RK2, step_MOM_dyn_split_RK2b as step_MOM_dyn_split_RK2
    from mom6_amd.open_boundary import ocean_OBC_type
    from mom6_amd.tracer_advect import DeviceGrid
    from mom6_amd.vert_friction import vertvisc_type
    from test_dyn_split_rk2_obc import DT, HV, TC3_FLAGS, oracle_state, rk2_obc_case, visc_arrays
    from test_hor_visc import REF_NAMES
    dist = _init(rank, world, port)
    try:
        gg, d, taux, tauy, OBCg = rk2_obc_case(segs, ni=32, nj=24, nk=3, seed=6)
        H, U, V = _abi.POS_H, _abi.POS_U, _abi.POS_V
        rng = np.random.default_rng(77)
        fu, fv = rng.standard_normal(gg.shape3(U)), rng.standard_normal(gg.shape3(V))
        f2u, f2v = rng.standard_normal(gg.shape2(U)), rng.standard_normal(gg.shape2(V))
        _obc_data_from_fields(OBCg, fu, fv, f2u, f2v)
        dom = Domain(gg.ni, gg.nj, layout, rank, gg.halo, False, False)
        tg = dom.tile_grid(gg)
        dg = DeviceGrid(tg)
        dg.set_domain(dom)
        OBC = ocean_OBC_type(tg, segs, idg_offset=dom.i0 - gg.halo, jdg_offset=dom.j0 - gg.halo, ni_global=gg.ni, nj_global=gg.nj, gamma_uv=0.3,
                             rx_max=10.0, **TC3_FLAGS)
        OBC.rx_normal, OBC.ry_normal = tg.zeros3(U), tg.zeros3(V)
        if any(sg.oblique for sg in OBC.segment):
            OBC.rx_oblique_u, OBC.ry_oblique_u, OBC.cff_normal_u = tg.zeros3(U), tg.zeros3(U), tg.zeros3(U)
            OBC.rx_oblique_v, OBC.ry_oblique_v, OBC.cff_normal_v = tg.zeros3(V), tg.zeros3(V), tg.zeros3(V)
        _obc_data_from_fields(OBC, dom.cut(fu, U), dom.cut(fv, V), dom.cut(f2u, U), dom.cut(f2v, V))
        T = lambda a, p: torch.from_numpy(dom.cut(a, p)).cuda()
        u, v, h, Tt, Ss = T(d["u"], U), T(d["v"], V), T(d["h"], H), T(d["T"], H), T(d["S"], H)
        Z = lambda p, k3=True: torch.zeros(tg.shape3(p) if k3 else tg.shape2(p), dtype=torch.float64, device="cuda")
        uh, vh, uhtr, vhtr, eta_av = Z(U), Z(V), Z(U), Z(V), Z(H, False)
        bbl = visc_arrays(gg)
        kw = dict(vertvisc=dict(KV=1.0e-3, HBBL=10.0), hor_visc={REF_NAMES[k]: x for k, x in HV.items()}) if viscous else {}
        CS = initialize_dyn_split_RK2(u, v, h, uh, vh, DT, dg, coriolis=dict(bound_coriolis=True), OBC=OBC.cuda(), **kw)
        CS.barotropic_CSp.st.dtbt = DT / 12.6
        pos_of = dict(Kv_bbl_u=U, Kv_bbl_v=V, bbl_thick_u=U, bbl_thick_v=V)
        visc = vertvisc_type(**{n: T(a, pos_of[n]) for n, a in bbl.items()}) if viscous else None
        tx, ty = T(taux, U), T(tauy, V)
        nsteps = 3
        for n in range(nsteps):
            step_MOM_dyn_split_RK2(u, v, h, (Tt, Ss), visc, None, DT, (tx, ty), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS)
        dg.sync()
        np.savez(os.path.join(out_dir, f"obc_tile{rank}.npz"), ij=np.array([dom.i0, dom.j0, dom.ni, dom.nj]), u=u.cpu().numpy(),
                 v=v.cpu().numpy(), h=h.cpu().numpy(), eta=CS.eta.cpu().numpy(), uhtr=uhtr.cpu().numpy(), rx=OBC.rx_normal.cpu().numpy(),
                 ry=OBC.ry_normal.cpu().numpy())
        dg.close()
        if rank == 0:
            ref = oracle_state(gg, d, OBCg, viscous, bbl=bbl, rk2b=rk2b)
            for n in range(nsteps):
                ref.step(taux, tauy)
            np.savez(os.path.join(out_dir, "obc_global.npz"), u=ref.u, v=ref.v, h=ref.h, eta=ref.arrs["eta"], uhtr=ref.uhtr, rx=OBCg.rx_normal,
                     ry=OBCg.ry_normal)
    finally:
        dist.destroy_process_group()


def phillips_layout_worker(rank, world, port, layout, out_dir, nsteps=2):
    """test.layout on BASELINE configs[3] (Phillips_2layer 480x320x2, LINEAR equation of state): the tiles of the layout
    against the one-tile oracle run; the barotropic subcycle has dozens of steps between its group passes."""
    import numpy as np
    import torch
    import exact_synth as xs
    from mom6_amd import _abi
    from mom6_amd.domains import Domain
    from mom6_amd.dynamics_split_rk2 import initialize_dyn_split_RK2, step_MOM_dyn_split_RK2
    from mom6_amd.tracer_advect import DeviceGrid
    from oracle import orc
    dist = _init(rank, world, port)
    try:
        gg, d, _ = xs.make_phillips()
        dt = 1800.0
        dom = Domain(gg.ni, gg.nj, layout, rank, gg.halo, True, False)
        tg = dom.tile_grid(gg)
        dg = DeviceGrid(tg)
        dg.set_domain(dom)
        H, U, V = _abi.POS_H, _abi.POS_U, _abi.POS_V
        T = lambda a, p: torch.from_numpy(dom.cut(a, p)).cuda()
        u, v, h, Tt, Ss = T(d["u"], U), T(d["v"], V), T(d["h"], H), T(d["T"], H), T(d["S"], H)
        Z = lambda p, k3=True: torch.zeros(tg.shape3(p) if k3 else tg.shape2(p), dtype=torch.float64, device="cuda")
        uh, vh, uhtr, vhtr, eta_av = Z(U), Z(V), Z(U), Z(V), Z(H, False)
        CS = initialize_dyn_split_RK2(u, v, h, uh, vh, dt, dg, EQN_OF_STATE="LINEAR", coriolis=dict(bound_coriolis=True))
        tx, ty = Z(U, False), Z(V, False)
        for n in range(nsteps):
            step_MOM_dyn_split_RK2(u, v, h, (Tt, Ss), None, None, dt, (tx, ty), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS,
                                   calc_dtbt=(n == 0))
        dg.sync()
        st = CS.barotropic_CSp.st
        np.savez(os.path.join(out_dir, f"ph_tile{rank}.npz"), ij=np.array([dom.i0, dom.j0, dom.ni, dom.nj]), u=u.cpu().numpy(),
                 v=v.cpu().numpy(), h=h.cpu().numpy(), eta=CS.eta.cpu().numpy(), uhtr=uhtr.cpu().numpy(), dtbt=st.dtbt, nstep=st.nstep_last)
        dg.close()
        if rank == 0:
            ref = orc.DynState(gg, d["u"], d["v"], d["h"], d["T"], d["S"], dt, eos_form="LINEAR")
            tz = (gg.zeros2(U), gg.zeros2(V))
            for n in range(nsteps):
                ref.step(tz[0], tz[1], calc_dtbt=(n == 0))
            np.savez(os.path.join(out_dir, "ph_global.npz"), u=ref.u, v=ref.v, h=ref.h, eta=ref.arrs["eta"], uhtr=ref.uhtr, dtbt=ref.bcs.dtbt,
                     nstep=ref.bcs.nstep_last)
    finally:
        dist.destroy_process_group()


def reproducing_sum_layout_worker(rank, world, port, layout, out_dir):
    """reproducing_sum / the chksum statistics of fields cut into tiles against the one-tile numbers"""
    import numpy as np
    import torch
    from mom6_amd import _abi, synth
    from mom6_amd.checksums import chksum, substats
    from mom6_amd.coms import reproducing_sum
    from mom6_amd.domains import Domain
    from mom6_amd.tracer_advect import DeviceGrid
    dist = _init(rank, world, port)
    try:
        NI, NJ, NK, halo = 70, 40, 3, 4
        gg = synth.make_grid(NI, NJ, NK, halo=halo, reentrant_x=True, reentrant_y=False, seed=78)
        d = synth.make_dynamics_state(gg, seed=3, umax=0.1, eta_amp=0.2)
        rng = np.random.default_rng(1)
        big = torch.from_numpy(rng.standard_normal(gg.shape3(_abi.POS_H)) * 10.0 ** rng.integers(-20, 20, gg.shape3(_abi.POS_H)))

        def numbers(dg, cut):
            out = {}
            for name, a, pos in (("h", d["h"], _abi.POS_H), ("u", d["u"], _abi.POS_U), ("v", d["v"], _abi.POS_V), ("big", big, _abi.POS_H)):
                da = cut(a, pos)
                r = reproducing_sum(da, pos, dg, by_layer=True)
                t = reproducing_sum(da, pos, dg)
                out[name + "_efp"] = np.array([e.v for e in r.EFP_lay_sums] + [r.EFP_sum.v, t.EFP_sum.v], dtype=np.int64)
                out[name + "_sums"] = np.array(r.sums + [r.sum, t.sum, float(t.npoints)])
                out[name + "_stats"] = np.array(substats(da, pos, dg) + (float(chksum(da, pos, dg)[0]),))
            return out

        dom = Domain(NI, NJ, layout, rank, halo, True, False)
        dg = DeviceGrid(dom.tile_grid(gg))
        dg.set_domain(dom)
        res = numbers(dg, lambda a, pos: dom.cut(a, pos).cuda())
        np.savez(os.path.join(out_dir, f"tile{rank}.npz"), **res)
        dg.close()
        if rank == 0:
            dg1 = DeviceGrid(gg)
            np.savez(os.path.join(out_dir, "global.npz"), **numbers(dg1, lambda a, pos: a.cuda()))
            dg1.close()
    finally:
        dist.destroy_process_group()


def write_energy_layout_worker(rank, world, port, layout, out_dir, ape=False):
    """write_energy's sums on tiles against the one-tile numbers (ape: with the depth list and the available potential energy)"""
    import numpy as np
    import torch
    from mom6_amd import _abi, synth
    from mom6_amd.domains import Domain
    from mom6_amd.sum_output import write_energy
    from mom6_amd.tracer_advect import DeviceGrid
    dist = _init(rank, world, port)
    try:
        NI, NJ, NK, halo = 70, 40, 3, 4
        gg = synth.make_grid(NI, NJ, NK, halo=halo, reentrant_x=True, reentrant_y=False, seed=78, land_frac=0.2)
        d = synth.make_dynamics_state(gg, seed=3, umax=0.3, eta_amp=0.2)

        def numbers(dg, cut):
            gp = None
            extra = {}
            if ape:
                from mom6_amd.sum_output import depth_list_setup
                gp = np.full(NK + 1, 9.8 * 2.0e-3); gp[0] = 9.8
                extra["depth_list"] = np.array([float(depth_list_setup(dg, domain=dom_of(dg)))])
            r = write_energy(cut(d["u"], _abi.POS_U), cut(d["v"], _abi.POS_V), cut(d["h"], _abi.POS_H),
                             (cut(d["T"], _abi.POS_H), cut(d["S"], _abi.POS_H)), dg, 900.0, g_prime=gp)
            if ape:
                extra["PE"] = np.array(r["PE"]); extra["Z_0APE"] = np.array(r["Z_0APE"])
            return dict(**extra, mass_EFP=np.array(r["mass_EFP"], dtype=np.int64), salt_EFP=np.array(r["salt_EFP"], dtype=np.int64),
                        heat_EFP=np.array(r["heat_EFP"], dtype=np.int64), mass_lay=np.array(r["mass_lay"]), KE_lay=np.array(r["KE_lay"]),
                        totals=np.array([r["mass_tot"], r["KE_tot"], r["toten"], float(r["npoints"])]), max_CFL=np.array(r["max_CFL"]))

        dom = Domain(NI, NJ, layout, rank, halo, True, False)
        dg = DeviceGrid(dom.tile_grid(gg))
        dg.set_domain(dom)
        doms = {id(dg): dom}
        dom_of = lambda g_: doms.get(id(g_))
        np.savez(os.path.join(out_dir, f"tile{rank}.npz"), **numbers(dg, lambda a, pos: dom.cut(a, pos).cuda()))
        dg.close()
        if rank == 0:
            dg1 = DeviceGrid(gg)
            np.savez(os.path.join(out_dir, "global.npz"), **numbers(dg1, lambda a, pos: a.cuda()))
            dg1.close()
    finally:
        dist.destroy_process_group()
