"""Phase clock of cont_flux_coop_kernel (a -DFC_TRACE variant of the library: tools/build_variant.sh fctrace continuity.hip
-DFC_TRACE, run with MOM6HIP_LIB_PATH=variants/libmom6hip_fctrace.so): mean s_memtime ticks per block and phase."""
import ctypes as C, json, sys; sys.path.insert(0, '.')
import torch
from mom6_amd import synth, _abi
from mom6_amd._lib import lib
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
L = lib()
L.mom6hip_fc_trace.argtypes = [C.POINTER(C.c_uint64), C.c_int]
names = ["loads", "reconstruct", "first eval + sums", "brackets", "newton uhbt", "newton du0", "u_cor", "duR/duL chain", "three fits", "  newton: scalar logic", "  newton: evaluation", "  newton: sums", "", "", "passes", "blocks"]
kw = dict(uhbt=uhbt, vhbt=vhbt, visc_rem_u=vru, visc_rem_v=vrv, u_cor=ucor, v_cor=vcor, BT_cont=bt)
f = lambda: continuity(st["u"], st["v"], st["h"], hp, uh, vh, 900.0, dg, cs, **kw)
f(); torch.cuda.synchronize()
L.mom6hip_fc_trace(None, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
dg.kernel_timing(True)
f(); torch.cuda.synchronize()
t = dg.kernel_timing(False)
both = (C.c_uint64 * 128)()
L.mom6hip_fc_trace(both, 0)
for d in (0, 1):
    out = both[64 * d:64 * d + 64]
    nb = max(out[15], 1)
    tot = max(sum(out[i] for i in range(13)), 1)
    print(f"direction {'xy'[d]}: blocks {nb}  flux {t[d][0]:.2f} ms   ticks/block {tot / nb:.0f}   newton passes/block {out[14] / nb:.2f}")
    for i in range(13):
        print(f"  {names[i]:20s} {out[i] / nb:10.0f} ticks  {100.0 * out[i] / tot:5.1f} %")
    print("faces by passes alive :", [out[16 + n] for n in range(24)])
    print("solves by passes run  :", [out[40 + n] for n in range(24)])
