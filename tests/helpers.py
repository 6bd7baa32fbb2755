"""Shared helpers for the parity tests."""
import numpy as np

from mom6_amd import _abi, synth


def advect_case(ni=24, nj=20, nk=3, ntr=3, seed=3, land_frac=0.2, hot_frac=0.004, vanish_frac=0.05,
                reentrant_x=True, reentrant_y=False, halo=4, cfl=0.15):
    g = synth.make_grid(ni, nj, nk, halo=halo, land_frac=land_frac, seed=seed + 100,
                        reentrant_x=reentrant_x, reentrant_y=reentrant_y)
    st = synth.make_advection_state(g, ntr=ntr, seed=seed, hot_frac=hot_frac, vanish_frac=vanish_frac, cfl=cfl)
    case = {k: (v.numpy() if k != "tr" else [t.numpy() for t in v]) for k, v in st.items()}
    return g, case


def run_oracle(orc, g, case, scheme, dt=3600.0, cs_dt=900.0, x_first=None, use_vol=True,
               conc_underflow=None, max_iter=None):
    tr = [t.copy() for t in case["tr"]]
    vol = case["vol0"].copy() if use_vol else None
    uhr = g.zeros3(_abi.POS_U); vhr = g.zeros3(_abi.POS_V)
    st = orc.advect_tracer(g, case["h_end"], case["uhtr"], case["vhtr"], dt, cs_dt, scheme, tr,
                           conc_underflow=conc_underflow, x_first=x_first, vol_prev=vol,
                           update_vol_prev=use_vol, uhr_out=uhr, vhr_out=vhr, max_iter=max_iter)
    return {"tr": tr, "vol": vol, "uhr": uhr, "vhr": vhr, "stats": st}


def interior(g, a, pos=_abi.POS_H):
    sj, si = g.csl(pos)
    return a[..., sj, si]


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint64), np.ascontiguousarray(b).view(np.uint64))
