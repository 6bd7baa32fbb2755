"""The cost of the open-boundary form of step_MOM_dyn_split_RK2 (step_with_obc: the plain sequence of the OBC entry points) beside the tuned
closed-domain step, on a regional grid with tc3's four FLATHER,ORLANSKI segments.  usage: python tools/perf_obc.py [ni nj nk] [steps]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from mom6_amd import _abi, synth  # noqa: E402
from mom6_amd.dynamics_split_rk2 import initialize_dyn_split_RK2, step_MOM_dyn_split_RK2  # noqa: E402
from mom6_amd.open_boundary import ocean_OBC_type  # noqa: E402
from mom6_amd.tracer_advect import DeviceGrid  # noqa: E402
from mom6_amd.vert_friction import vertvisc_type  # noqa: E402

U, V, H = _abi.POS_U, _abi.POS_V, _abi.POS_H
TC3 = ["J=N,I=N:0,FLATHER,ORLANSKI", "J=0,I=0:N,FLATHER,ORLANSKI", "I=N,J=0:N,FLATHER,ORLANSKI", "I=0,J=N:0,FLATHER,ORLANSKI"]
DT = 300.0


def run(ni, nj, nk, steps, with_obc):
    g = synth.make_grid(ni, nj, nk, land_frac=0.1, seed=304, reentrant_x=False, reentrant_y=False)
    OBC = None
    if with_obc:
        from test_continuity_obc import open_faces
        OBC = ocean_OBC_type(g, TC3, gamma_uv=0.3, rx_max=10.0, freeslip_vorticity=True, freeslip_strain=True, zero_biharmonic=True)
        open_faces(g, OBC)
        OBC.rx_normal, OBC.ry_normal = g.zeros3(U), g.zeros3(V)
        for s in OBC.segment:
            s.normal_vel_bt[:] = 0.0; s.SSH[:] = 0.0
        OBC.cuda()
    d = synth.make_dynamics_state(g, seed=4, umax=0.1, eta_amp=0.2)
    dg = DeviceGrid(g)
    u, v, h, T, S = (d[k].cuda() for k in ("u", "v", "h", "T", "S"))
    Z = lambda pos, k3=True: torch.zeros(g.shape3(pos) if k3 else g.shape2(pos), dtype=torch.float64, device="cuda")
    uh, vh, uhtr, vhtr, eta_av = Z(U), Z(V), Z(U), Z(V), Z(H, False)
    CS = initialize_dyn_split_RK2(u, v, h, uh, vh, DT, dg, coriolis=dict(bound_coriolis=True), vertvisc=dict(KV=1.0e-3, HBBL=10.0),
                                  hor_visc=dict(LAPLACIAN=True, KH_VEL_SCALE=0.01, AH_VEL_SCALE=0.05, SMAGORINSKY_AH=True, SMAG_BI_CONST=0.06), OBC=OBC)
    rng = np.random.default_rng(9)
    su, sv = g.shape2(U), g.shape2(V)
    X = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    visc = vertvisc_type(Kv_bbl_u=X(1.0e-3 * (0.5 + rng.random(su))), Kv_bbl_v=X(1.0e-3 * (0.5 + rng.random(sv))),
                         bbl_thick_u=X(2.0 + 8.0 * rng.random(su)), bbl_thick_v=X(2.0 + 8.0 * rng.random(sv)))
    tx, ty = X(0.0 * np.asarray(g.mask2dCu)), X(0.0 * np.asarray(g.mask2dCv))
    step = lambda n: step_MOM_dyn_split_RK2(u, v, h, (T, S), visc, None, DT, (tx, ty), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS, calc_dtbt=(n == 0))
    for n in range(3):
        step(n)
    dg.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for n in range(steps):
        step(n + 3)
    dg.sync(); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    ok = bool(torch.isfinite(u).all() and torch.isfinite(h).all())
    dg.close()
    return ms, ok


if __name__ == "__main__":
    only = None      # --only obc | closed: one of the two runs (for a kernel trace of that step alone)
    if "--only" in sys.argv:
        q = sys.argv.index("--only"); only = sys.argv[q + 1]; del sys.argv[q:q + 2]
    a = [int(x) for x in sys.argv[1:]]
    ni, nj, nk = (a + [720, 540, 20])[:3] if len(a) < 3 else a[:3]
    steps = a[3] if len(a) > 3 else 10
    out = {"grid": [ni, nj, nk], "steps": steps}
    for name, flag in (("closed_tuned_step_ms", False), ("obc_step_ms", True)):
        if only and (only == "obc") != flag:
            continue
        ms, ok = run(ni, nj, nk, steps, flag)
        out[name] = round(ms, 3); out[name.replace("_ms", "_finite")] = ok
        print(json.dumps(out), flush=True)
