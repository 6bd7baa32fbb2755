"""advect_tracer through the HOST memory space of the C ABI (what an unmodified Fortran caller with host arrays
gets) against the DEVICE memory space, same inputs: the PCIe-inclusive rate for DESIGN.md."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mom6_amd import synth
from mom6_amd.tracer_advect import DeviceGrid, advect_tracer, tracer_advect_init
ni, nj, nk = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "720x540x75").split("x"))
g = synth.make_grid(ni, nj, nk, seed=1)
st = synth.make_advection_state(g, ntr=4, seed=2, hot_frac=2e-5)
dg = DeviceGrid(g)
CS = tracer_advect_init(900.0, "PPM:H3")
cells = ni * nj * nk
for space in ("device", "host"):
    if space == "device":
        a = {k: (v.cuda() if k != "tr" else [t.cuda() for t in v]) for k, v in st.items() if k in ("h_end", "uhtr", "vhtr", "tr")}
    else:
        a = {k: (v.numpy().copy() if k != "tr" else [t.numpy().copy() for t in v]) for k, v in st.items() if k in ("h_end", "uhtr", "vhtr", "tr")}
    ts = []
    for r in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        advect_tracer(a["h_end"], a["uhtr"], a["vhtr"], None, 3600.0, dg, CS, a["tr"])
        dg.sync(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = min(ts)
    print(f"{space:6s} {ni}x{nj}x{nk}: {t*1e3:8.2f} ms per call  {cells/t/1e9:7.3f} Gcell/s  ({(4+3)*8*cells/1e9:.2f} GB in, {4*8*cells/1e9:.2f} GB out over the bus in host mode)")
