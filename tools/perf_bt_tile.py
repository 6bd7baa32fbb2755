"""One tile of an 8-GPU run, rehearsed on one GPU: a 1440x135x75 tile whose group passes go through the library's RCCL exchange
with the rank as its own neighbour (Domain(self_exchange=True)), so the multi-tile code path of btstep runs -- the subcycle
as one hipGraph per segment between two group passes.  MOM6HIP_BT_GRAPH=0 enqueues it kernel by kernel for comparison.
usage: python tools/perf_bt_tile.py [NIxNJxNK] [steps]"""
import json, os, sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import torch
from mom6_amd import synth, _abi
from mom6_amd.domains import Domain
from mom6_amd.tracer_advect import DeviceGrid
from mom6_amd.dynamics_split_rk2 import initialize_dyn_split_RK2, step_MOM_dyn_split_RK2

NI, NJ, NK = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1440x135x75").split('x')]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
g = synth.make_grid(NI, NJ, NK, seed=20241020, land_frac=0.3, reentrant_x=True, reentrant_y=True)
dom = Domain(g.ni, g.nj, (1, 1), 0, g.halo, g.reentrant_x, g.reentrant_y, self_exchange=True)
tg = dom.tile_grid(g)
dg = DeviceGrid(tg)
dom.attach_native(dg)
d = synth.make_dynamics_state(g, seed=1, umax=0.05, eta_amp=0.1, device="cuda")
U, V, H = _abi.POS_U, _abi.POS_V, _abi.POS_H
Z = lambda pos, k3=True: torch.zeros(g.shape3(pos) if k3 else g.shape2(pos), dtype=torch.float64, device="cuda")
u, v, h = d["u"], d["v"], d["h"]
uh, vh, uhtr, vhtr, eta_av = Z(U), Z(V), Z(U), Z(V), Z(H, False)
dt = 900.0
CS = initialize_dyn_split_RK2(u, v, h, uh, vh, dt, dg, coriolis=dict(bound_coriolis=True))
tx, ty = Z(U, False), Z(V, False)


def step(n):
    step_MOM_dyn_split_RK2(u, v, h, (d["T"], d["S"]), None, None, dt, (tx, ty), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS, calc_dtbt=(n == 0))


for n in range(2):
    step(n)
dg.sync()
c0, l0 = dg.bt_graph_stats()
t0 = time.perf_counter()
for n in range(K):
    step(2 + n)
dg.sync()
ms = (time.perf_counter() - t0) / K * 1e3
c1, l1 = dg.bt_graph_stats()
print(json.dumps({"tile": [NI, NJ, NK], "bt_graph": os.environ.get("MOM6HIP_BT_GRAPH", "1"), "ms_per_step": ms, "nstep": int(CS.barotropic_CSp.st.nstep_last),
                  "graph_launches_per_step": (l1 - l0) / K, "graphs_captured": c1}))
dg.close()
