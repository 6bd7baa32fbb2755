"""Per-variant duration of the two flux kernels of continuity_PPM on the benchmark grid (mom6hip_kernel_timing slots)."""
import sys, json; sys.path.insert(0, '.')
import torch
from mom6_amd import synth, _abi
from mom6_amd.tracer_advect import DeviceGrid
from mom6_amd.continuity import continuity, continuity_PPM_init, BT_cont_type
NI, NJ, NK = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1440x1080x75").split('x')]
g = synth.make_grid(NI, NJ, NK, seed=20241020)
st = synth.make_dynamics_state(g, seed=1, device="cuda")
dg = DeviceGrid(g)
cs = continuity_PPM_init(dg)
kk = (torch.arange(NK, device="cuda", dtype=torch.float64) + 0.5) / NK
vru = torch.clamp(1.0 - 0.8 * kk[:, None, None] ** 4 + 0 * st["u"], 0.05, 1.0).contiguous()
vrv = torch.clamp(1.0 - 0.8 * kk[:, None, None] ** 4 + 0 * st["v"], 0.05, 1.0).contiguous()
hp = st["h"].clone(); uh = torch.zeros_like(st["u"]); vh = torch.zeros_like(st["v"])
bt = BT_cont_type(**{n: torch.zeros(g.shape2(_abi.POS_U), device="cuda", dtype=torch.float64) for n in _abi.BT_CONT_U},
                  **{n: torch.zeros(g.shape2(_abi.POS_V), device="cuda", dtype=torch.float64) for n in _abi.BT_CONT_V})
continuity(st["u"], st["v"], st["h"], hp, uh, vh, 900.0, dg, cs, visc_rem_u=vru, visc_rem_v=vrv)
uhbt = (uh.sum(0) * 1.02).contiguous(); vhbt = (vh.sum(0) * 0.98).contiguous()
ucor = torch.zeros_like(st["u"]); vcor = torch.zeros_like(st["v"])
variants = {
    "plain": dict(),
    "bt_cont": dict(visc_rem_u=vru, visc_rem_v=vrv, BT_cont=bt),
    "uhbt+bt_cont": dict(uhbt=uhbt, vhbt=vhbt, visc_rem_u=vru, visc_rem_v=vrv, u_cor=ucor, v_cor=vcor, BT_cont=bt),
    "uhbt": dict(uhbt=uhbt, vhbt=vhbt, visc_rem_u=vru, visc_rem_v=vrv, u_cor=ucor, v_cor=vcor),
}
out = {}
for name, kw in variants.items():
    f = lambda: continuity(st["u"], st["v"], st["h"], hp, uh, vh, 900.0, dg, cs, **kw)
    f(); torch.cuda.synchronize()
    dg.kernel_timing(True)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t = dg.kernel_timing(False)
    out[name] = {"flux_x_ms": t[0][0] / max(t[0][1], 1), "flux_y_ms": t[1][0] / max(t[1][1], 1)}
print(json.dumps(out))
