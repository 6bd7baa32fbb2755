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


def halo_worker(rank, world, port, layout, rx, ry, out_dir):
    """Every rank fills the halos of its tile of random global fields through Domain.pass_var; the result must
    equal the same window of the one-tile halo update (oracle/domains.c), bit for bit."""
    import numpy as np
    import torch
    from mom6_amd import _abi, synth
    from mom6_amd.domains import Domain
    from oracle import orc
    dist = _init(rank, world, port)
    try:
        NI, NJ, NK, halo = 23, 17, 3, 4
        gg = synth.make_grid(NI, NJ, NK, halo=halo, reentrant_x=rx, reentrant_y=ry)
        dom = Domain(NI, NJ, layout, rank, halo, rx, ry)
        rng = np.random.default_rng(5)
        ok = True
        fields, poss, expect = [], [], []
        for pos in (_abi.POS_H, _abi.POS_U, _abi.POS_V, _abi.POS_Q):
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
                orc.halo_update(gg, after, pos)
                tile = dom.cut(G, pos).copy()
                # poison the tile's halos
                h = halo
                xs = 1 if pos in (_abi.POS_U, _abi.POS_Q) else 0
                ys = 1 if pos in (_abi.POS_V, _abi.POS_Q) else 0
                keep = tile[..., h:h + dom.nj + ys, h:h + dom.ni + xs].copy()
                tile[:] = -777.0
                tile[..., h:h + dom.nj + ys, h:h + dom.ni + xs] = keep
                fields.append(torch.from_numpy(tile)); poss.append(pos); expect.append(dom.cut(after, pos))
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
