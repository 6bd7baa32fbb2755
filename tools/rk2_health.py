"""Steps the bench model and prints per-step health numbers (where does a synthetic state go wrong?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mom6_amd import synth
from mom6_amd.domains import Domain

wl = sys.argv[1] if len(sys.argv) > 1 else "benchmark"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
NI, NJ, NK = bench.shape_of(wl)
grid = synth.make_grid(NI, NJ, NK, seed=20241020, rough_noise=0.0)     # smooth bathymetry: see Model
dom = Domain(NI, NJ, (1, 1), 0, grid.halo, grid.reentrant_x, grid.reentrant_y)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
M = bench.Model(grid, dom, dev, bench.SCHEME)
for n in range(nsteps):
    M.step()
    M.dg.sync()
    u, v, h = M.u, M.v, M.h
    ua = u.abs(); ku = int(ua.reshape(u.shape[0], -1).max(1).values.argmax())
    idx = int(ua.argmax()); k, r = divmod(idx, u.shape[1] * u.shape[2]); j, i = divmod(r, u.shape[2])
    hh = 0.5 * (h[k, j, max(i - 1, 0)] + h[k, j, min(i, h.shape[2] - 1)])
    print(f"step {n+1:3d} umax {float(ua.max()):10.3e} at k={k} j={j} i={i} (h there {float(hh):9.3e})  vmax {float(v.abs().max()):10.3e} "
          f"hmin {float(h.min()):9.2e} eta[{float(M.CS.eta.min()):8.3f},{float(M.CS.eta.max()):8.3f}] "
          f"accel_bt max {float(M.CS.u_accel_bt.abs().max()):9.2e} PFu max {float(M.CS.PFu.abs().max()):9.2e} CAu max {float(M.CS.CAu.abs().max()):9.2e} "
          f"nstep {M.CS.barotropic_CSp.st.nstep_last}", flush=True)
    if not torch.isfinite(ua.max()):
        break
