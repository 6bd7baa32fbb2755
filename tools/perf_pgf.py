"""Times PressureForce_FV_Bouss (Wright, PLM reconstruction) on the benchmark grid, device-resident, per kernel (library HIP events)."""
import sys, json; sys.path.insert(0, '.')
import torch
from mom6_amd import synth
from mom6_amd.tracer_advect import DeviceGrid
from mom6_amd.pressure_force import EOS_init, PressureForce, PressureForce_init
NI, NJ, NK = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1440x1080x75").split('x')]
g = synth.make_grid(NI, NJ, NK, seed=20241020, land_frac=0.3)
d = synth.make_dynamics_state(g, seed=1, device="cuda", umax=0.1, eta_amp=0.2)
dg = DeviceGrid(g)
cs = PressureForce_init(g); eos = EOS_init("WRIGHT")
PFu, PFv, pbce = torch.zeros_like(d["u"]), torch.zeros_like(d["v"]), torch.zeros_like(d["h"])
eta = torch.zeros(d["h"].shape[1:], device="cuda", dtype=torch.float64)
def T(f, n=5):
    f(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print(json.dumps({"PressureForce_ms": T(lambda: PressureForce(d["h"], (d["T"], d["S"], eos), PFu, PFv, dg, cs, pbce=pbce, eta=eta))}))
