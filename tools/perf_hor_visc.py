"""Times horizontal_viscosity (the bench's options) on the benchmark grid, device-resident."""
import sys, json; sys.path.insert(0, '.')
import torch
import bench
from mom6_amd import synth
from mom6_amd.tracer_advect import DeviceGrid
from mom6_amd.hor_visc import hor_visc_init, horizontal_viscosity
NI, NJ, NK = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1440x1080x75").split('x')]
g = synth.make_grid(NI, NJ, NK, seed=20241020, land_frac=bench.LAND_FRAC, rough_noise=bench.rough_noise(NI))
d = synth.make_dynamics_state(g, seed=11, device="cuda", **bench.STATE)
dg = DeviceGrid(g)
CS = hor_visc_init(dg, bench.DT, **bench.HOR_VISC)
du, dv = torch.zeros_like(d["u"]), torch.zeros_like(d["v"])
f = lambda: horizontal_viscosity(d["u"], d["v"], d["h"], du, dv, None, None, dg, CS)
f(); torch.cuda.synchronize()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(n): f()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / n
cells = NI * NJ * NK
print(json.dumps({"horizontal_viscosity_ms": ms, "GBs_algorithmic_40B": 40.0 * cells / ms / 1e6}))
