"""What restart / diagnostic staging costs the model: K steps of the bench model with u, v, h, T, S staged to pinned host arrays
every step (mom6hip_stage_to_host; one mom6hip_stage_wait at the end of each step's successor) against K steps without.
usage: python tools/perf_staging.py [NIxNJxNK] [steps]"""
import json, sys, time; sys.path.insert(0, '.')
import numpy as np
import torch
import bench
from mom6_amd.staging import host_register, stage_to_host, stage_wait

dims = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1440x1080x75").split('x')]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
from mom6_amd import synth
from mom6_amd.domains import Domain
torch.cuda.set_device(0)
grid = synth.make_grid(*dims, seed=20241020, land_frac=bench.LAND_FRAC, rough_noise=bench.rough_noise(dims[0]))
m = bench.Model(grid, Domain(dims[0], dims[1], (1, 1), 0, grid.halo, grid.reentrant_x, grid.reentrant_y), torch.device("cuda", 0), "PPM:H3")
names = ("u", "v", "h", "T", "S")
host = {n: np.empty(tuple(getattr(m, n).shape)) for n in names}
for n in names:
    host_register(host[n])
nbytes = sum(h.nbytes for h in host.values())
for _ in range(2):
    m.step()
m.dg.sync(); torch.cuda.synchronize()

def run(stage):
    t0 = time.perf_counter()
    for _ in range(K):
        if stage:
            stage_wait(m.dg)                  # the previous step's output has to have landed before its arrays are reused
            for n in names:
                stage_to_host(m.dg, host[n], getattr(m, n))
        m.step()
    if stage:
        stage_wait(m.dg)
    m.dg.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3

plain = run(False); staged = run(True); plain2 = run(False)
t0 = time.perf_counter()
for n in names:
    stage_to_host(m.dg, host[n], getattr(m, n))
stage_wait(m.dg)
alone = (time.perf_counter() - t0) * 1e3
print(json.dumps({"grid": dims, "steps": K, "staged_bytes_per_step": nbytes, "ms_per_step_plain": [plain, plain2], "ms_per_step_staged": staged,
                  "staging_alone_ms": alone, "d2h_GBps": nbytes / alone / 1e6}))
