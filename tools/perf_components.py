import sys, time, json; sys.path.insert(0,'.')
import torch, numpy as np
from mom6_amd import synth, _abi
from mom6_amd.tracer_advect import DeviceGrid
from mom6_amd.continuity import continuity, continuity_PPM_init, BT_cont_type
from mom6_amd.coriolis_adv import CorAdCalc, CoriolisAdv_init
from mom6_amd.ale import ALE_remap_tracers, initialize_remapping
NI,NJ,NK = [int(x) for x in (sys.argv[1] if len(sys.argv)>1 else "1440x1080x75").split('x')]
g = synth.make_grid(NI,NJ,NK, seed=20241020)
st = synth.make_dynamics_state(g, seed=1, device="cuda")
dg = DeviceGrid(g)
cs = continuity_PPM_init(dg)
kk = (torch.arange(NK, device="cuda", dtype=torch.float64)+0.5)/NK
vru = torch.clamp(1.0-0.8*kk[:,None,None]**4 + 0*st["u"], 0.05, 1.0).contiguous()
vrv = torch.clamp(1.0-0.8*kk[:,None,None]**4 + 0*st["v"], 0.05, 1.0).contiguous()
hp = st["h"].clone(); uh = torch.zeros_like(st["u"]); vh = torch.zeros_like(st["v"])
def T(f, n=3):
    f(); torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n
cells=NI*NJ*NK
bt = BT_cont_type(**{n: torch.zeros(g.shape2(_abi.POS_U), device="cuda", dtype=torch.float64) for n in _abi.BT_CONT_U}, **{n: torch.zeros(g.shape2(_abi.POS_V), device="cuda", dtype=torch.float64) for n in _abi.BT_CONT_V})
t1 = T(lambda: continuity(st["u"], st["v"], st["h"], hp, uh, vh, 900.0, dg, cs, visc_rem_u=vru, visc_rem_v=vrv, BT_cont=bt))
uhbt = (uh.sum(0)*1.02).contiguous(); vhbt=(vh.sum(0)*0.98).contiguous()
ucor=torch.zeros_like(st["u"]); vcor=torch.zeros_like(st["v"])
t2 = T(lambda: continuity(st["u"], st["v"], st["h"], hp, uh, vh, 900.0, dg, cs, uhbt=uhbt, vhbt=vhbt, visc_rem_u=vru, visc_rem_v=vrv, u_cor=ucor, v_cor=vcor, BT_cont=bt))
t3 = T(lambda: continuity(st["u"], st["v"], st["h"], hp, uh, vh, 900.0, dg, cs, uhbt=uhbt, vhbt=vhbt, visc_rem_u=vru, visc_rem_v=vrv, u_cor=ucor, v_cor=vcor))
t0 = T(lambda: continuity(st["u"], st["v"], st["h"], hp, uh, vh, 900.0, dg, cs))
CS = CoriolisAdv_init(bound_coriolis=True)
CAu=torch.zeros_like(st["u"]); CAv=torch.zeros_like(st["v"])
tc = T(lambda: CorAdCalc(st["u"], st["v"], st["h"], uh, vh, CAu, CAv, None, dg, CS))
print(json.dumps({"shape":[NI,NJ,NK],"cont_plain_ms":t0,"cont_btcont_ms":t1,"cont_uhbt_btcont_ms":t2,"cont_uhbt_ms":t3,"coradcalc_ms":tc,
  "cont_GBs_alg": 96*cells/ t3/1e6, "cor_GBs_alg": 56*cells/tc/1e6}))
if NK<=80:
    tr=[st["T"].clone(), st["S"].clone()]
    w = torch.rand_like(st["h"])+0.5
    hn = (w/w.sum(0,keepdim=True)*st["h"].sum(0,keepdim=True)).contiguous()
    for sch in ("PLM","PPM_H4"):
        R = initialize_remapping(sch)
        ta = T(lambda: ALE_remap_tracers(R, dg, st["h"], hn, tr), n=1)
        print(sch, "ale_remap 2 tracers ms", ta, "GB/s alg", (16+16*2)*cells/ta/1e6)
