import sys, json; sys.path.insert(0,'.')
import torch
from mom6_amd import synth, _abi
from mom6_amd.tracer_advect import DeviceGrid
from mom6_amd.coriolis_adv import CorAdCalc, CoriolisAdv_init
g = synth.make_grid(1440,1080,75, seed=20241020)
st = synth.make_dynamics_state(g, seed=1, device="cuda")
dg = DeviceGrid(g)
uh = torch.rand_like(st["u"]); vh = torch.rand_like(st["v"])
CS = CoriolisAdv_init(bound_coriolis=True)
CAu=torch.zeros_like(st["u"]); CAv=torch.zeros_like(st["v"])
def T(f, n=5):
    f(); torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n
print("coradcalc ms", T(lambda: CorAdCalc(st["u"], st["v"], st["h"], uh, vh, CAu, CAv, None, dg, CS)))
