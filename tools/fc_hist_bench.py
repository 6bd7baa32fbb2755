"""The Newton passes of cont_flux_coop_kernel in the bench's own step (a -DFC_TRACE variant: tools/build_variant.sh fctrace continuity.hip
-DFC_TRACE; MOM6HIP_LIB_PATH=variants/libmom6hip_fctrace.so): faces by the number of evaluation passes they were alive for, and the
solves of the blocks by the number of passes they ran, over model steps after a short spin-up.
    python tools/fc_hist_bench.py [spin-up steps] [measured steps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mom6_amd import synth
from mom6_amd._lib import lib
from mom6_amd.domains import Domain
nspin = int(sys.argv[1]) if len(sys.argv) > 1 else 12
nmeas = int(sys.argv[2]) if len(sys.argv) > 2 else 2
NI, NJ, NK = bench.shape_of("om4_025")
grid = synth.make_grid(NI, NJ, NK, seed=20241020, land_frac=bench.LAND_FRAC, rough_noise=bench.rough_noise(NI))
dom = Domain(NI, NJ, (1, 1), 0, grid.halo, grid.reentrant_x, grid.reentrant_y)
M = bench.Model(grid, dom, torch.device("cuda", 0), bench.SCHEME)
L = lib()
L.mom6hip_fc_trace.argtypes = [C.POINTER(C.c_uint64), C.c_int]
for n in range(nspin):
    M.step()
torch.cuda.synchronize()
L.mom6hip_fc_trace(None, 1)
for n in range(nmeas):
    M.step()
torch.cuda.synchronize()
out = (C.c_uint64 * 64)()
L.mom6hip_fc_trace(out, 0)
faces = [out[16 + n] for n in range(24)]; solves = [out[40 + n] for n in range(24)]
print("steps", nmeas, "blocks", out[15], "passes", out[14])
print("faces by passes alive :", faces)
print("solves by passes run  :", solves)
tf, ts = sum(faces), sum(solves)
print("mean passes per face %.3f, per solve %.3f" % (sum(n * c for n, c in enumerate(faces)) / max(tf, 1), sum(n * c for n, c in enumerate(solves)) / max(ts, 1)))
for cut in (1, 2, 3):
    left = sum(c for n, c in enumerate(faces) if n > cut)
    saved = sum((n - cut) * c for n, c in enumerate(solves) if n > cut)
    print(f"cut at {cut} passes: {100.0 * left / max(tf, 1):.2f} % of the face solves are left over; {saved} of {sum(n * c for n, c in enumerate(solves))} block passes saved")
