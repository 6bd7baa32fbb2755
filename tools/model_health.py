"""Health of the time-stepping model of bench.py over a longer run (one GPU): max |u|, max |eta|, mean kinetic energy and
velocity truncations every few steps, for a given bathymetric roughness.
    python tools/model_health.py --steps 48 --rough 0.04 [--workload om4_025]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mom6_amd import synth
from mom6_amd.domains import Domain

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=48); ap.add_argument("--every", type=int, default=4)
ap.add_argument("--rough", type=float, default=None); ap.add_argument("--workload", default="om4_025")
ap.add_argument("--land", type=float, default=bench.LAND_FRAC)
ap.add_argument("--no-wind", action="store_true", help="zero wind stress: what the initial state does on its own")
ap.add_argument("--uniform-ts", action="store_true", help="T and S replaced by their layer means: no baroclinic pressure gradients")
ap.add_argument("--lateral", action="store_true", help="thickness_diffuse and mixedlayer_restrat in every thermodynamic cycle")
ap.add_argument("--neutral", action="store_true", help="with --lateral: tracer_hordiff with USE_NEUTRAL_DIFFUSION")
ap.add_argument("--ts-amp", type=float, default=None); ap.add_argument("--ts-decay", type=float, default=None)
ap.add_argument("--umax", type=float, default=None); ap.add_argument("--u-noise", type=float, default=None)
a = ap.parse_args()
for k, v in (("ts_amp", a.ts_amp), ("ts_decay", a.ts_decay), ("umax", a.umax), ("u_noise", a.u_noise)):      # the bench's state, varied
    if v is not None:
        bench.STATE[k] = v
NI, NJ, NK = bench.shape_of(a.workload)
grid = synth.make_grid(NI, NJ, NK, seed=20241020, land_frac=a.land, rough_noise=bench.rough_noise(NI) if a.rough is None else a.rough)
dom = Domain(NI, NJ, (1, 1), 0, grid.halo, grid.reentrant_x, grid.reentrant_y)
M = bench.Model(grid, dom, torch.device("cuda", 0), bench.SCHEME)
if a.lateral:
    M.enable_lateral(neutral=a.neutral)
if a.no_wind:
    M.taux.zero_(); M.tauy.zero_()
if a.uniform_ts:
    m = torch.as_tensor(grid.mask2dT, device=M.T.device)[None]
    for F in (M.T, M.S):
        F.copy_(((F * m).sum((1, 2)) / m.sum())[:, None, None].expand_as(F).contiguous())
print(json.dumps(dict(step=0, **M.health())), flush=True)
from mom6_amd.vert_friction import vertvisc_ntrunc
for n in range(a.steps):
    M.step()
    if (n + 1) % a.every == 0:
        h = M.health()
        au = M.u.abs(); idx = int(au.argmax()); k, r = divmod(idx, au.shape[1] * au.shape[2]); j, i = divmod(r, au.shape[2])
        h.update(step=n + 1, ntrunc=int(vertvisc_ntrunc(M.dg, M.CS.vertvisc_CSp)), umax_at=(k, j, i))
        print(json.dumps(h), flush=True)
        if h["nan"]:
            break
