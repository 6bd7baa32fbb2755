"""A regional domain with tc3's four FLATHER,ORLANSKI segments stepped for a while through the library: a disc of raised surface in the
middle of a flat basin at rest (tc3's THICKNESS_CONFIG = "circle_obcs") radiates out through the open boundaries.  Reports the kinetic
energy and the surface height through time: they have to drain, not reflect or grow.  usage: python tools/health_obc.py [ni nj nk steps]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from mom6_amd import _abi, synth  # noqa: E402
from mom6_amd.dynamics_split_rk2 import initialize_dyn_split_RK2, step_MOM_dyn_split_RK2  # noqa: E402
from mom6_amd.open_boundary import ocean_OBC_type  # noqa: E402
from mom6_amd.tracer_advect import DeviceGrid  # noqa: E402
from mom6_amd.vert_friction import vertvisc_type  # noqa: E402
from test_continuity_obc import open_faces  # noqa: E402

U, V, H = _abi.POS_U, _abi.POS_V, _abi.POS_H
TC3 = ["J=N,I=N:0,FLATHER,ORLANSKI", "J=0,I=0:N,FLATHER,ORLANSKI", "I=N,J=0:N,FLATHER,ORLANSKI", "I=0,J=N:0,FLATHER,ORLANSKI"]


def main(ni=160, nj=120, nk=4, steps=600, dt=300.0, closed=False):
    g = synth.make_grid(ni, nj, nk, land_frac=0.0, seed=304, reentrant_x=False, reentrant_y=False)
    OBC = None
    if not closed:
        OBC = ocean_OBC_type(g, TC3, gamma_uv=0.3, rx_max=10.0, freeslip_vorticity=True, freeslip_strain=True, zero_biharmonic=True)
        open_faces(g, OBC)
        OBC.rx_normal, OBC.ry_normal = g.zeros3(U), g.zeros3(V)
        for s in OBC.segment:
            s.normal_vel_bt[:] = 0.0; s.SSH[:] = 0.0
        OBC.cuda()
    d = synth.make_dynamics_state(g, seed=4, umax=0.0, eta_amp=0.0)
    h = d["h"].numpy().copy()
    tot = h.sum(0)
    depth = np.asarray(g.bathyT) * g.Z_to_H
    h *= np.where(tot > 0, depth / np.maximum(tot, 1e-30), 1.0)[None]      # at rest, flat surface
    jj, ii = np.meshgrid(np.arange(h.shape[1]), np.arange(h.shape[2]), indexing="ij")
    r2 = ((ii - h.shape[2] / 2) ** 2 + (jj - h.shape[1] / 2) ** 2) / (0.12 * min(ni, nj)) ** 2
    h[0] += 1.0 * np.exp(-r2) * np.asarray(g.mask2dT)                        # the disc: 1 m of extra surface height
    dg = DeviceGrid(g)
    X = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    u, v, hh = X(0 * d["u"].numpy()), X(0 * d["v"].numpy()), X(h)
    T, S = X(0 * h + 10.0), X(0 * h + 35.0)
    Z = lambda pos, k3=True: torch.zeros(g.shape3(pos) if k3 else g.shape2(pos), dtype=torch.float64, device="cuda")
    uh, vh, uhtr, vhtr, eta_av = Z(U), Z(V), Z(U), Z(V), Z(H, False)
    CS = initialize_dyn_split_RK2(u, v, hh, uh, vh, dt, dg, coriolis=dict(bound_coriolis=True), vertvisc=dict(KV=1.0e-4, HBBL=10.0),
                                  hor_visc=dict(LAPLACIAN=True, KH_VEL_SCALE=0.003, SMAGORINSKY_KH=True, SMAG_LAP_CONST=0.15), OBC=OBC)
    su, sv = g.shape2(U), g.shape2(V)
    visc = vertvisc_type(Kv_bbl_u=X(np.full(su, 1.0e-3)), Kv_bbl_v=X(np.full(sv, 1.0e-3)), bbl_thick_u=X(np.full(su, 5.0)), bbl_thick_v=X(np.full(sv, 5.0)))
    tx, ty = Z(U, False), Z(V, False)
    sj, si = slice(g.halo, g.halo + nj), slice(g.halo, g.halo + ni)
    dep = X(depth)
    out = []
    for n in range(steps):
        step_MOM_dyn_split_RK2(u, v, hh, (T, S), visc, None, dt, (tx, ty), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS, calc_dtbt=(n == 0))
        if n % (steps // 12) == 0 or n == steps - 1:
            dg.sync()
            eta = (hh.sum(0) - dep)[sj, si]
            ke = float(0.5 * ((u[:, sj, si] ** 2).mean() + (v[:, sj, si] ** 2).mean()))
            out.append(dict(step=n + 1, max_abs_eta=float(eta.abs().max()), mean_eta=float(eta.mean()), ke_per_mass=ke, max_u=float(u.abs().max()),
                            finite=bool(torch.isfinite(u).all() and torch.isfinite(hh).all())))
            print(json.dumps(out[-1]), flush=True)
    dg.close()
    return out


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    ni, nj, nk, steps = (a + [160, 120, 4, 600][len(a):])[:4]
    res = {"grid": [ni, nj, nk], "dt": 300.0, "open": main(ni, nj, nk, steps), "closed": main(ni, nj, nk, steps, closed=True)}
    o, c = res["open"][-1], res["closed"][-1]
    res["summary"] = {"open_final_max_abs_eta": o["max_abs_eta"], "closed_final_max_abs_eta": c["max_abs_eta"], "open_final_ke": o["ke_per_mass"],
                      "closed_final_ke": c["ke_per_mass"], "open_mean_eta": o["mean_eta"], "closed_mean_eta": c["mean_eta"]}
    print(json.dumps(res))
