"""Times thickness_diffuse (KHTH alone, WRIGHT), mixedlayer_restrat (OM4 settings) and tracer_hordiff with USE_NEUTRAL_DIFFUSION (4 tracers,
KHTR = 50) on the benchmark grid, device-resident."""
import os, sys, json; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from mom6_amd import synth
from mom6_amd.tracer_advect import DeviceGrid
from mom6_amd.pressure_force import EOS_init
from mom6_amd.thickness_diffuse import thickness_diffuse, thickness_diffuse_init
from mom6_amd.mixedlayer_restrat import mixedlayer_restrat, mixedlayer_restrat_init
from mom6_amd.tracer_hor_diff import tracer_hordiff, tracer_hor_diff_init
NI, NJ, NK = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1440x1080x75").split('x')]
g = synth.make_grid(NI, NJ, NK, seed=20241020, land_frac=0.3)
d = synth.make_dynamics_state(g, seed=1, device="cuda", umax=0.1, eta_amp=0.2)
dg = DeviceGrid(g)
eos = EOS_init("WRIGHT")
sh2 = tuple(d["h"].shape[1:])
yy = torch.linspace(0.0, 1.0, sh2[0], device="cuda", dtype=torch.float64)[:, None].expand(sh2).contiguous()
td = thickness_diffuse_init(dg, THICKNESSDIFFUSE=True, KHTH=600.0)
mle = mixedlayer_restrat_init(dg, FOX_KEMPER_ML_RESTRAT_COEF=1.0, MLE_FRONT_LENGTH=500.0, MLE_USE_PBL_MLD=True, MLE_MLD_DECAY_TIME=345600.0,
                              MLE_MLD_DECAY_TIME2=5184000.0, FOX_KEMPER_ML_RESTRAT_COEF2=0.5, MLD_filtered=torch.zeros(sh2, device="cuda", dtype=torch.float64),
                              MLD_filtered_slow=torch.zeros(sh2, device="cuda", dtype=torch.float64))
ustar, h_MLD, Rd = 0.005 + 0.01 * yy, 20.0 + 80.0 * yy, (0.2 + 1.5 * yy).contiguous()
uq, vq = torch.zeros_like(d["u"]), torch.zeros_like(d["v"])
def T(f, n=4):
    f(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
hh = d["h"].clone()
nd = tracer_hor_diff_init(KHTR=50.0, USE_NEUTRAL_DIFFUSION=True)
trs = [d["T"].clone(), d["S"].clone(), torch.rand_like(d["T"]), torch.rand_like(d["T"])]
tv = dict(T=trs[0], S=trs[1], eqn_of_state=eos)
nd_sym = tracer_hor_diff_init(KHTR=50.0, USE_NEUTRAL_DIFFUSION=True, NDIFF_ANSWER_DATE=20250101)
print(json.dumps({"tracer_hordiff_neutral_4tr_symmetric_ms": T(lambda: tracer_hordiff(hh, 3600.0, None, None, None, dg, nd_sym, trs, tv=tv), n=2),
                  "tracer_hordiff_neutral_4tr_ms": T(lambda: tracer_hordiff(hh, 3600.0, None, None, None, dg, nd, trs, tv=tv), n=2),
                  "thickness_diffuse_ms": T(lambda: thickness_diffuse(hh, uq, vq, (d["T"], d["S"], eos), 3600.0, dg, None, None, None, td)),
                  "mixedlayer_restrat_ms": T(lambda: mixedlayer_restrat(hh, uq, vq, (d["T"], d["S"], eos), dict(ustar=ustar), 3600.0, None, h_MLD, None, dict(Rd_dx_h=Rd), dg, mle))}))
