"""Where the milliseconds of a step go.  Two uses:
  python3 tools/step_budget.py run [workload] [steps]     the bench's model, SPINUP + 4 warm-up steps, then `steps` steps and nothing else
                                                          (run it under `rocprofv3 --kernel-trace -d DIR -o NAME --`)
  python3 tools/step_budget.py report DB [steps]          per kernel: launches and milliseconds a step over the LAST `steps` steps of the
                                                          trace (the window opens at the pgf_face_kernel launch that starts them), the
                                                          sum of the kernels' durations against the window's length (what is left is
                                                          time with no kernel running: launch gaps, host work, waits)"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(workload, steps):
    import torch
    import bench
    from mom6_amd import synth
    from mom6_amd.domains import Domain
    NI, NJ, NK = bench.shape_of(workload)
    grid = synth.make_grid(NI, NJ, NK, seed=20241020, land_frac=bench.LAND_FRAC, rough_noise=bench.rough_noise(NI))
    dom = Domain(NI, NJ, (1, 1), 0, grid.halo, grid.reentrant_x, grid.reentrant_y)
    torch.cuda.set_device(0)
    M = bench.Model(grid, dom, torch.device("cuda", 0), bench.SCHEME)
    for n in range(bench.SPINUP + 4):
        M.step()
    M.dg.sync(); torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for n in range(steps):
        M.step()
    M.dg.sync(); torch.cuda.synchronize()
    print(f"step_budget: {steps} steps, {1e3 * (time.perf_counter() - t0) / steps:.3f} ms/step (under whatever tracer is attached)")


def report(path, steps):
    import sqlite3
    c = sqlite3.connect(path).cursor()
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else cols[0]
    rows = sorted(c.execute(f"select {name_col}, start, end from kernels").fetchall(), key=lambda r: r[1])
    starts = [s for n, s, e in rows if "pgf_face_kernel" in n]
    assert len(starts) >= steps, (len(starts), steps)
    t0 = starts[-steps]
    win = [(n, s, e) for n, s, e in rows if s >= t0]
    t1 = max(e for n, s, e in win)
    agg = {}
    for n, s, e in win:
        n = n.replace("(anonymous namespace)::", "").replace("void ", "")
        n = re.sub(r"mom6hip_[a-z_0-9]+::\{lambda", "{lambda", n)
        n = re.sub(r"\((?:[^()]|\([^()]*\))*\)( \[clone [^\]]*\])?$", "", n)[:110]
        a = agg.setdefault(n, [0, 0])
        a[0] += 1; a[1] += e - s
    # time covered by at least one kernel (kernels of side streams overlap the compute stream's)
    busy, cur_s, cur_e = 0, None, None
    for n, s, e in win:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot = sum(a[1] for a in agg.values())
    print(f"{'kernel':110s} {'calls/step':>10s} {'ms/step':>9s} {'avg us':>9s}")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:110s} {a[0] / steps:10.2f} {a[1] / 1e6 / steps:9.3f} {a[1] / a[0] / 1e3:9.2f}")
    print(f"window {1e-6 * (t1 - t0) / steps:.3f} ms/step; kernels' durations {1e-6 * tot / steps:.3f} ms/step; some kernel running "
          f"{1e-6 * busy / steps:.3f} ms/step; nothing running {1e-6 * ((t1 - t0) - busy) / steps:.3f} ms/step; launches {len(win) / steps:.0f} a step")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2] if len(sys.argv) > 2 else "om4_025", int(sys.argv[3]) if len(sys.argv) > 3 else 8)
    else:
        report(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 8)
