import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import tripolar as tp
from helpers import bits_equal
from mom6_amd import _abi
from oracle import orc
from test_native_domain import native_grid, H, U, V
from mom6_amd.dynamics_split_rk2 import initialize_dyn_split_RK2, step_MOM_dyn_split_RK2
g, g2 = tp.grids(ni=70, nj=10, nk=3)
dom, dg = native_grid(g)
d, _, (taux, tauy), _ = tp.states(g, g2)
dt = 900.0
ref = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], dt)
ref.bcs.dtbt = dt / 6.6
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
u, v, h, Tt, Ss = (T(d[k]) for k in ("u", "v", "h", "T", "S"))
Z = lambda pos, k3=True: torch.zeros(g.shape3(pos) if k3 else g.shape2(pos), dtype=torch.float64, device="cuda")
uh, vh, uhtr, vhtr, eta_av = Z(U), Z(V), Z(U), Z(V), Z(H, False)
CS = initialize_dyn_split_RK2(u, v, h, uh, vh, dt, dg, coriolis=dict(bound_coriolis=True))
CS.barotropic_CSp.st.dtbt = ref.bcs.dtbt
tx, ty = T(taux), T(tauy)
for n in range(3):
    ref.step(taux, tauy)
    step_MOM_dyn_split_RK2(u, v, h, (Tt, Ss), None, None, dt, (tx, ty), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS)
    dg.sync()
    for name, a, b in (("u", u, ref.u), ("v", v, ref.v), ("h", h, ref.h), ("uh", uh, ref.uh), ("vh", vh, ref.vh), ("uhtr", uhtr, ref.uhtr), ("eta", CS.eta, ref.arrs["eta"]),
                       ("u_av", CS.u_av, ref.arrs["u_av"]), ("v_av", CS.v_av, ref.arrs["v_av"]), ("h_av", CS.h_av, ref.arrs["h_av"]), ("CAu", CS.CAu, ref.arrs["CAu"]),
                       ("CAu_pred", CS.CAu_pred, ref.arrs["CAu_pred"]), ("visc_rem_u", CS.visc_rem_u, ref.arrs["visc_rem_u"]), ("PFu", CS.PFu, ref.arrs["PFu"]),
                       ("u_accel_bt", CS.u_accel_bt, ref.arrs["u_accel_bt"]), ("uhbt", CS.uhbt, ref.arrs["uhbt"])):
        an = a.cpu().numpy()
        if not bits_equal(an, b):
            w = np.argwhere(an.view(np.uint64) != b.view(np.uint64))
            print(n, name, "DIFF", len(w), "first", w[:4].tolist(), "shape", an.shape, float(np.nanmax(np.abs(an - b))))
print("done")
