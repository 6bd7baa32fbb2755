#!/usr/bin/env python3
"""bench.py -- throughput of the MOM6 dynamical-core hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one baroclinic time step (DT) of the hot path on the synthetic global C-grid named in
`config.workload`; the hot-path components that run in a step are listed in `config.kernels`
(tracer advection runs every DT_THERM/DT-th step, as in src/core/MOM.F90:923-927).  Inputs are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md "Chip-level parameters")
DT = 900.0                 # baroclinic step [s]
DT_THERM = 3600.0          # tracer/thermodynamic step [s]
NTR = 4                    # T, S + 2 passive tracers
SCHEME = "PPM:H3"
HOT_FRAC = 2.0e-5


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="om4_025", help="om4_025 | benchmark | double_gyre | NIxNJxNK")
    ap.add_argument("--scheme", default=SCHEME)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def shape_of(name):
    from mom6_amd import synth
    if name in synth.CONFIGS:
        return synth.CONFIGS[name]
    return tuple(int(x) for x in name.lower().split("x"))


def cpu_baseline(grid, scheme, full_cells, steps_per_advect):
    """The CPU oracle (oracle/tracer_advect.c, a scalar C restatement of the reference routine; kind
    "port") timed on a bounded sample of the same workload: the same horizontal grid, 2 layers."""
    import numpy as np
    from mom6_amd import synth
    from oracle import orc
    nk_s = 2
    g = synth.make_grid(grid.ni, grid.nj, nk_s, halo=grid.halo, seed=20241020)
    st = synth.make_advection_state(g, ntr=NTR, seed=1, hot_frac=HOT_FRAC)
    tr = [t.numpy() for t in st["tr"]]
    h_end, uhtr, vhtr = st["h_end"].numpy(), st["uhtr"].numpy(), st["vhtr"].numpy()
    reps, t_used = 0, 0.0
    while t_used < 10.0 and reps < 20:
        t0 = time.perf_counter()
        orc.advect_tracer(g, h_end, uhtr, vhtr, DT_THERM, DT, scheme, tr)
        t_used += time.perf_counter() - t0
        reps += 1
    sec_per_cell_call = t_used / reps / (g.ni * g.nj * nk_s)
    sec_per_step = sec_per_cell_call * full_cells / steps_per_advect
    return {
        "value": DT / sec_per_step / 365.0, "unit": "SYPD", "cores": 1, "kind": "port",
        "ns_per_gridpoint_step": sec_per_step * 1e9 / full_cells,
        "sample": f"{reps} advect_tracer calls on {g.ni}x{g.nj}x{nk_s} (same horizontal grid, 2 of "
                  f"{grid.nk} layers), scaled per cell to the full grid; {t_used:.1f} s of CPU",
    }


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py: --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from mom6_amd import _abi, synth
    from mom6_amd.tracer_advect import DeviceGrid, advect_tracer, tracer_advect_init

    NI, NJ, NK = shape_of(a.workload)
    # N>1: independent replicas of the same tile per rank until the RCCL halo exchange lands
    # (DESIGN.md "Multi-GPU"); scaling is then weak by construction.
    grid = synth.make_grid(NI, NJ, NK, seed=20241020)
    # hot_frac: about one cell in 50 000 needs the flux limiter (and with it a second, sparse iteration)
    st = synth.make_advection_state(grid, ntr=NTR, seed=1 + rank, device=f"cuda:{local_rank}", hot_frac=HOT_FRAC)
    dg = DeviceGrid(grid, device=local_rank)
    CS = tracer_advect_init(DT, a.scheme)
    tr = st["tr"]
    steps_per_advect = int(round(DT_THERM / DT))
    cells = NI * NJ * NK

    def step(n):
        # tracer advection every DT_THERM/DT-th baroclinic step (MOM.F90:923-927)
        if (n + 1) % steps_per_advect == 0:
            return advect_tracer(st["h_end"], st["uhtr"], st["vhtr"], None, DT_THERM, dg, CS, tr)
        return None

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    last = None
    for n in range(a.warmup):
        last = step(n) or last
    dg.sync()
    barrier()
    t0 = time.perf_counter()
    for n in range(a.steps):
        last = step(n) or last
    dg.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    sec_per_step = elapsed / a.steps
    sypd = world * DT / sec_per_step / 365.0      # whole job: `world` replicas of the tile
    out = {
        "metric": "simulated-years/day (SYPD) of the implemented hot-path kernels",
        "value": sypd, "unit": "SYPD", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": sec_per_step * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "ns_per_gridpoint_step": sec_per_step * 1e9 / cells,
        "config": {
            "workload": f"{a.workload} {NI}x{NJ}x{NK} global C-grid, halo 4, reentrant-x, ~25% land, "
                        f"{NTR} tracers, DT={DT:.0f}s DT_THERM={DT_THERM:.0f}s",
            "kernels": {"advect_tracer": f"{a.scheme}, 1 call per {steps_per_advect} steps"},
            "not_yet_in_step": ["continuity_PPM", "CorAdCalc", "PressureForce_FV", "btstep", "ALE regrid/remap",
                                "RK2 momentum updates"],
            "advect_iterations_last_call": None if last is None else int(last.iterations),
            "parallelism": "1 tile per GPU" if world == 1 else f"{world} independent tile replicas",
        },
    }

    if rank == 0 and not a.no_roofline:
        # dominant kernel: measured per launch with HIP events on the library's own stream
        dg.set_timing(True)
        acc = {"x1": 0.0, "y1": 0.0, "x": 0.0, "y": 0.0, "nx": 0, "ny": 0, "total": 0.0, "calls": 0,
               "setup": 0.0, "halo": 0.0}
        for _ in range(3):
            advect_tracer(st["h_end"], st["uhtr"], st["vhtr"], None, DT_THERM, dg, CS, tr)
            t = dg.advect_timing()
            acc["x1"] += t.ms_x1; acc["y1"] += t.ms_y1; acc["x"] += t.ms_x; acc["y"] += t.ms_y
            acc["nx"] += t.n_x; acc["ny"] += t.n_y; acc["setup"] += t.ms_setup; acc["halo"] += t.ms_halo
            acc["total"] += t.ms_total; acc["calls"] += 1
        dg.set_timing(False)
        nc = acc["calls"]
        # the dense (first-iteration, every row active) launch of each pass: one per call
        ms_x, ms_y = acc["x1"] / nc, acc["y1"] / nc
        bytes_per_launch = (NTR + 2) * 16.0 * cells        # read+write Tr(ntr), hprev, uhr|vhr
        dom, ms_dom = ("adv_y_kernel<4,PPM:H3,first>", ms_y) if ms_y >= ms_x else ("adv_x_kernel<4,PPM:H3,first>", ms_x)
        achieved = bytes_per_launch / (ms_dom * 1e-3) / 1e9
        out["roofline"] = {
            "kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "avg_launch_ms": {"adv_x_kernel_first": ms_x, "adv_y_kernel_first": ms_y},
            "advect_tracer_call_ms": {"total": acc["total"] / nc, "setup+scan": acc["setup"] / nc,
                                      "halo": acc["halo"] / nc, "x_all_iterations": acc["x"] / nc,
                                      "y_all_iterations": acc["y"] / nc},
        }

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(grid, a.scheme, cells, steps_per_advect)

    if rank == 0:
        print(json.dumps(out))
    dg.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
