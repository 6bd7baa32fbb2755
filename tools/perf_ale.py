"""Times the ALE remapping entries on the benchmark grid (device-resident).  MOM6HIP_ALE_STREAM = 0 / 1 / 2 selects the wave-per-column
kernel or the streaming kernel with one or two fields a launch for PPM_H4."""
import sys, json; sys.path.insert(0, '.')
import torch
from mom6_amd import synth, _abi
from mom6_amd.tracer_advect import DeviceGrid
from mom6_amd.ale import ALE_remap_tracers, ALE_remap_velocities, ALE_remap_set_h_vel, initialize_remapping

NI, NJ, NK = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1440x1080x75").split('x')]
g = synth.make_grid(NI, NJ, NK, seed=20241020)
st = synth.make_dynamics_state(g, seed=1, device="cuda")
dg = DeviceGrid(g)


def T(f, n=3):
    f(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


cells = NI * NJ * NK
w = torch.rand_like(st["h"]) * 0.1 + 0.95
hn = (w / (w * st["h"]).sum(0, keepdim=True) * st["h"] * st["h"].sum(0, keepdim=True)).contiguous()
out = {"shape": [NI, NJ, NK]}
for sch in ("PLM", "PPM_H4"):
    R = initialize_remapping(sch)
    tr = [st["T"].clone(), st["S"].clone(), st["T"].clone(), st["S"].clone()]
    ta = T(lambda: ALE_remap_tracers(R, dg, st["h"], hn, tr))
    out[sch + "_tracers4_ms"] = ta
    out[sch + "_tracers4_GBs_alg"] = (16 + 16 * 4) * cells / ta / 1e6
    hu0 = torch.zeros_like(st["u"]); hv0 = torch.zeros_like(st["v"]); hu1 = torch.zeros_like(st["u"]); hv1 = torch.zeros_like(st["v"])
    ALE_remap_set_h_vel(R, dg, st["h"], hu0, hv0); ALE_remap_set_h_vel(R, dg, hn, hu1, hv1)
    u = st["u"].clone(); v = st["v"].clone()
    tv = T(lambda: ALE_remap_velocities(R, dg, hu0, hv0, hu1, hv1, u, v))
    out[sch + "_velocities_ms"] = tv
    out[sch + "_velocities_GBs_alg"] = 2 * 32 * cells / tv / 1e6
print(json.dumps(out))
