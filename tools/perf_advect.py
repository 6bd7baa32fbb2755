"""Times the dense (first-iteration) passes of advect_tracer on the benchmark grid, 4 tracers, PPM:H3 and PPM (library HIP events)."""
import sys, json; sys.path.insert(0, '.')
import torch
from mom6_amd import synth
from mom6_amd.tracer_advect import DeviceGrid, advect_tracer, tracer_advect_init
NI, NJ, NK = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1440x1080x75").split('x')]
g = synth.make_grid(NI, NJ, NK, seed=20241020, land_frac=0.3)
ad = synth.make_advection_state(g, ntr=4, seed=1, device="cuda", hot_frac=2.0e-5)
dg = DeviceGrid(g)
dg.set_timing(True)
out = {}
for scheme in ("PPM:H3", "PPM"):
    cs = tracer_advect_init(900.0, scheme)
    tx = ty = 0.0
    for q in range(4):
        advect_tracer(ad["h_end"], ad["uhtr"], ad["vhtr"], None, 3600.0, dg, cs, ad["tr"])
        t = dg.advect_timing()
        if q:
            tx += t.ms_x1 / 3; ty += t.ms_y1 / 3
    cells = NI * NJ * NK
    out[scheme] = {"adv_x_ms": tx, "adv_y_ms": ty, "adv_x_frac_of_8TBs": 6 * 16.0 * cells / (tx * 1e-3) / 8e12, "adv_y_frac_of_8TBs": 6 * 16.0 * cells / (ty * 1e-3) / 8e12}
print(json.dumps(out))
