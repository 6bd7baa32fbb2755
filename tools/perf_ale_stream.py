"""One call each of ALE_remap_tracers (4 fields) and ALE_remap_velocities with PPM_H4 on the benchmark grid (for counter passes:
short; MOM6HIP_ALE_STREAM selects the kernel form)."""
import sys, json; sys.path.insert(0, '.')
import torch
from mom6_amd import synth
from mom6_amd.tracer_advect import DeviceGrid
from mom6_amd.ale import ALE_remap_tracers, ALE_remap_velocities, ALE_remap_set_h_vel, initialize_remapping
NI, NJ, NK = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1440x1080x75").split('x')]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = synth.make_grid(NI, NJ, NK, seed=20241020)
st = synth.make_dynamics_state(g, seed=1, device="cuda")
dg = DeviceGrid(g)
w = torch.rand_like(st["h"]) * 0.1 + 0.95
hn = (w / (w * st["h"]).sum(0, keepdim=True) * st["h"] * st["h"].sum(0, keepdim=True)).contiguous()
R = initialize_remapping("PPM_H4")
tr = [st["T"].clone(), st["S"].clone(), st["T"].clone(), st["S"].clone()]
hu0 = torch.zeros_like(st["u"]); hv0 = torch.zeros_like(st["v"]); hu1 = torch.zeros_like(st["u"]); hv1 = torch.zeros_like(st["v"])
ALE_remap_set_h_vel(R, dg, st["h"], hu0, hv0); ALE_remap_set_h_vel(R, dg, hn, hu1, hv1)
u = st["u"].clone(); v = st["v"].clone()
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True); c = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps): ALE_remap_tracers(R, dg, st["h"], hn, tr)
b.record()
for _ in range(reps): ALE_remap_velocities(R, dg, hu0, hv0, hu1, hv1, u, v)
c.record(); torch.cuda.synchronize()
print(json.dumps({"shape": [NI, NJ, NK], "tracers4_ms": a.elapsed_time(b) / reps, "velocities_ms": b.elapsed_time(c) / reps}))
