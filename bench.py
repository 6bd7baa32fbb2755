#!/usr/bin/env python3
"""bench.py -- throughput of the MOM6 dynamical-core hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one baroclinic time step (DT) of the hot path on the synthetic global C-grid named in
`config.workload`, in the order of step_MOM_dyn_split_RK2 (src/core/MOM_dynamics_split_RK2.F90:289-1176):
PressureForce (:495), continuity (:636, BT_cont), continuity (:757, uhbt + BT_cont), CorAdCalc (:869),
continuity (:1015, uhbt), CorAdCalc (:1061), and every DT_THERM/DT-th step advect_tracer
(src/core/MOM.F90:1438) and ALE_remap_tracers (:1662).  `config.kernels` lists what runs in the step and
`config.not_yet_in_step` what the reference's step also does but this build does not yet provide.
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
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
REMAP_SCHEME = "PPM_H4"    # OM4-class remapping scheme
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


class Step:
    """The hot-path step on one tile of the global grid, state resident in HBM.  `dom` is the tile's Domain
    (mom6_amd/domains.py); every rank generates the same global synthetic state on its own GPU, keeps the
    window of its tile and frees the rest, so N ranks step exactly the problem one rank steps."""

    def __init__(self, gg, dom, device, scheme):
        from mom6_amd import _abi, synth
        from mom6_amd.ale import initialize_remapping
        from mom6_amd.continuity import BT_cont_type, continuity_PPM_init
        from mom6_amd.coriolis_adv import CoriolisAdv_init
        from mom6_amd.pressure_force import EOS_init, PressureForce_init
        from mom6_amd.tracer_advect import DeviceGrid, tracer_advect_init
        self.dom = dom
        grid = self.g = dom.tile_grid(gg) if dom.nranks > 1 else gg
        self.dg = DeviceGrid(grid, device=device.index)
        if dom.nranks > 1:
            self.dg.set_domain(dom)
        dev = str(device)
        H, U, V = _abi.POS_H, _abi.POS_U, _abi.POS_V

        def tile(state, pos_of):
            out = {}
            for k, v in state.items():
                pos = pos_of(k)
                out[k] = [dom.cut(t, pos).clone() for t in v] if isinstance(v, list) else dom.cut(v, pos).clone()
            return out
        Z = lambda pos, k3=True: torch.zeros(grid.shape3(pos) if k3 else grid.shape2(pos), dtype=torch.float64, device=dev)
        # tracer-advection inputs (h_end, uhtr, vhtr, tracers): hot_frac = about one cell in 50 000 needs the
        # flux limiter and with it a second, sparse iteration
        adv = synth.make_advection_state(gg, ntr=NTR, seed=1, device=dev, hot_frac=HOT_FRAC)
        self.adv = tile(adv, lambda k: {"uhtr": U, "vhtr": V}.get(k, H))
        del adv
        dyn = synth.make_dynamics_state(gg, seed=11, device=dev)
        self.dyn = tile(dyn, lambda k: {"u": U, "uh": U, "v": V, "vh": V}.get(k, H))
        del dyn
        torch.cuda.empty_cache()
        d = self.dyn
        kk = (torch.arange(grid.nk, device=dev, dtype=torch.float64) + 0.5) / grid.nk
        self.vru = torch.clamp(1.0 - 0.8 * kk[:, None, None] ** 4 + 0 * d["u"], 0.05, 1.0).contiguous()
        self.vrv = torch.clamp(1.0 - 0.8 * kk[:, None, None] ** 4 + 0 * d["v"], 0.05, 1.0).contiguous()
        self.hp, self.h2 = d["h"].clone(), d["h"].clone()
        self.uh, self.vh = Z(_abi.POS_U), Z(_abi.POS_V)
        self.u_av, self.v_av = Z(_abi.POS_U), Z(_abi.POS_V)
        self.CAu, self.CAv = Z(_abi.POS_U), Z(_abi.POS_V)
        self.PFu, self.PFv, self.pbce = Z(_abi.POS_U), Z(_abi.POS_V), Z(_abi.POS_H)
        self.eta = Z(_abi.POS_H, False)
        # BT_THICK_SCHEME = FROM_BT_CONT (the default with USE_BT_CONT_TYPE): continuity also hands back h_u, h_v
        self.bt = BT_cont_type(**{n: Z(_abi.POS_U, False) for n in _abi.BT_CONT_U},
                               **{n: Z(_abi.POS_V, False) for n in _abi.BT_CONT_V}, h_u=Z(_abi.POS_U), h_v=Z(_abi.POS_V))
        self.cont_cs = continuity_PPM_init(self.dg)
        self.cor_cs = CoriolisAdv_init(bound_coriolis=True)
        self.pgf_cs = PressureForce_init(grid)
        self.eos = EOS_init("WRIGHT")
        self.adv_cs = tracer_advect_init(DT, scheme)
        self.remap_cs = initialize_remapping(REMAP_SCHEME)
        # barotropic transports for the corrector calls: the layer sums of a first continuity call, nudged
        from mom6_amd.continuity import continuity
        continuity(d["u"], d["v"], d["h"], self.hp, self.uh, self.vh, DT, self.dg, self.cont_cs,
                   visc_rem_u=self.vru, visc_rem_v=self.vrv)
        self.uhbt = (self.uh.sum(0) * 1.02).contiguous()
        self.vhbt = (self.vh.sum(0) * 0.98).contiguous()
        # barotropic solver: control structure, wind stress, and the baroclinic accelerations it is forced with
        # (u_bc_accel = CAu + PFu [+ diffu], MOM_dynamics_split_RK2.F90:557-564, evaluated once: the bench state is stationary)
        from mom6_amd.barotropic import barotropic_init, set_dtbt
        from mom6_amd.coriolis_adv import CorAdCalc
        from mom6_amd.pressure_force import PressureForce
        self.bt_cs = barotropic_init(self.dg)
        PressureForce(d["h"], (d["T"], d["S"], self.eos), self.PFu, self.PFv, self.dg, self.pgf_cs, pbce=self.pbce, eta=self.eta)
        CorAdCalc(d["u"], d["v"], d["h"], self.uh, self.vh, self.CAu, self.CAv, None, self.dg, self.cor_cs)
        mu = torch.as_tensor(grid.mask2dCu, device=dev); mv = torch.as_tensor(grid.mask2dCv, device=dev)
        # keep the synthetic forcing model-like: accelerations of at most a few 1e-5 m s-2
        self.bc_u = (torch.clamp(self.CAu + self.PFu, -3e-5, 3e-5) * mu).contiguous()
        self.bc_v = (torch.clamp(self.CAv + self.PFv, -3e-5, 3e-5) * mv).contiguous()
        yy = torch.linspace(0.0, 3.1416, grid.shape2(_abi.POS_U)[0], device=dev, dtype=torch.float64)
        self.taux = (0.1 * torch.cos(2 * yy)[:, None] * mu).contiguous()
        self.tauy = (0.0 * mv).contiguous()
        self.eta_bt = self.eta.clone()          # the barotropic solver's own free surface
        self.eta_pred, self.eta_av = Z(_abi.POS_H, False), Z(_abi.POS_H, False)
        self.al_u, self.al_v = Z(_abi.POS_U), Z(_abi.POS_V)
        self.uhbt_bt, self.vhbt_bt = Z(_abi.POS_U, False), Z(_abi.POS_V, False)
        from mom6_amd.barotropic import btcalc
        continuity(d["u"], d["v"], d["h"], self.hp, self.uh, self.vh, DT, self.dg, self.cont_cs, BT_cont=self.bt,
                   visc_rem_u=self.vru, visc_rem_v=self.vrv)
        btcalc(d["h"], self.dg, self.bt_cs, self.bt.arrays["h_u"], self.bt.arrays["h_v"])
        set_dtbt(self.dg, self.bt_cs, pbce=self.pbce, BT_cont=self.bt)      # calc_dtbt (:1110 of MOM_dynamics_split_RK2.F90)
        # a z*-like target grid for the remap: same column totals, slightly different partition
        w = 1.0 + 0.05 * torch.sin(6.2832 * kk)[:, None, None] + 0 * d["h"]
        self.h_new = (d["h"] * w / (d["h"] * w).sum(0, keepdim=True) * d["h"].sum(0, keepdim=True)).contiguous()
        self.steps_per_advect = int(round(DT_THERM / DT))
        self.last_adv = None
        self.dg.sync()

    def parts(self, n):
        """(name, callable) of the hot-path calls of baroclinic step n, in the reference's order."""
        from mom6_amd.ale import ALE_remap_tracers
        from mom6_amd.continuity import continuity
        from mom6_amd.coriolis_adv import CorAdCalc
        from mom6_amd.pressure_force import PressureForce
        from mom6_amd.tracer_advect import advect_tracer
        from mom6_amd.barotropic import bt_mass_source, btcalc, btstep
        d, dg = self.dyn, self.dg
        bt_args = lambda: (d["u"], d["v"], self.eta_bt, DT, self.bc_u, self.bc_v, (self.taux, self.tauy), self.pbce, self.eta,
                           self.u_av, self.v_av, self.al_u, self.al_v, self.eta_pred, self.uhbt_bt, self.vhbt_bt, dg, self.bt_cs,
                           self.vru, self.vrv)
        bt_kw = lambda: dict(BT_cont=self.bt, uh0=self.uh, vh0=self.vh, u_uh0=d["u"], v_vh0=d["v"])
        vr = dict(visc_rem_u=self.vru, visc_rem_v=self.vrv)
        H, U, V = 0, 1, 2      # _abi.POS_H, POS_U, POS_V

        def gp(fields, pos, halo):
            # do_group_pass: one tile = the library's own wrap kernels; N tiles = torch.distributed (RCCL)
            if self.dom.nranks == 1:
                return lambda: dg.halo_update(fields, pos)
            return lambda: self.dom.pass_var(fields, pos, halo=halo)

        out = [
            ("PressureForce", lambda: PressureForce(d["h"], (d["T"], d["S"], self.eos), self.PFu, self.PFv, dg, self.pgf_cs,
                                                    pbce=self.pbce, eta=self.eta)),
            ("pass_eta", gp([self.eta], [H], 1)),                                                        # :610
            ("continuity[BT_cont]", lambda: continuity(d["u"], d["v"], d["h"], self.hp, self.uh, self.vh, DT, dg,
                                                       self.cont_cs, BT_cont=self.bt, **vr)),
            ("btcalc+bt_mass_source", lambda: (btcalc(d["h"], dg, self.bt_cs, self.bt.arrays["h_u"], self.bt.arrays["h_v"]),
                                               bt_mass_source(d["h"], self.eta_bt, True, dg, self.bt_cs))),    # :629, :615
            ("btstep[pred]", lambda: btstep(*bt_args(), **bt_kw())),                                      # :655
            ("pass_visc_rem+uvp", gp([self.vru, self.vrv, d["u"], d["v"]], [U, V, U, V], 3)),           # :747,:751
            ("continuity[uhbt+BT_cont]", lambda: continuity(d["u"], d["v"], d["h"], self.hp, self.uh, self.vh, DT, dg,
                                                            self.cont_cs, uhbt=self.uhbt, vhbt=self.vhbt, u_cor=self.u_av,
                                                            v_cor=self.v_av, BT_cont=self.bt, **vr)),
            ("pass_hp_uv", gp([self.hp, self.u_av, self.v_av, self.uh, self.vh], [H, U, V, U, V], 2)),   # :763
            ("CorAdCalc", lambda: CorAdCalc(self.u_av, self.v_av, self.hp, self.uh, self.vh, self.CAu, self.CAv, None, dg,
                                            self.cor_cs)),
            ("btstep[corr]", lambda: btstep(*bt_args(), **bt_kw(), etaav=self.eta_av)),                    # :911
            ("pass_visc_rem+uv", gp([self.vru, self.vrv, d["u"], d["v"]], [U, V, U, V], 3)),            # :1004,:1008
            # the reference updates h in place here (:1015); a separate output keeps the bench state stationary
            ("continuity[uhbt]", lambda: continuity(d["u"], d["v"], d["h"], self.h2, self.uh, self.vh, DT, dg, self.cont_cs,
                                                    uhbt=self.uhbt, vhbt=self.vhbt, u_cor=self.u_av, v_cor=self.v_av, **vr)),
            ("pass_h+av_uvh", gp([self.h2, self.u_av, self.v_av, self.uh, self.vh], [H, U, V, U, V], 3)),  # :1018,:1027
            ("CorAdCalc[pred]", lambda: CorAdCalc(self.u_av, self.v_av, self.h2, self.uh, self.vh, self.CAu, self.CAv, None, dg,
                                                  self.cor_cs)),
        ]
        if (n + 1) % self.steps_per_advect == 0:
            a = self.adv

            def adv():
                self.last_adv = advect_tracer(a["h_end"], a["uhtr"], a["vhtr"], None, DT_THERM, dg, self.adv_cs, a["tr"])
            out.append(("advect_tracer", adv))
            out.append(("ALE_remap_tracers", lambda: ALE_remap_tracers(self.remap_cs, dg, d["h"], self.h_new, a["tr"])))
        return out

    def run(self, n):
        for _, f in self.parts(n):
            f()


def cpu_baseline(grid, scheme, full_cells, steps_per_advect):
    """The CPU oracle (oracle/*.c, a scalar C restatement of the reference routines; kind "port") timed on a
    bounded sample of the same workload: the same horizontal grid with 2 of the layers, one full cycle of
    steps_per_advect baroclinic steps.  The 3-D work is scaled per cell to the full grid; the barotropic
    subcycle is 2-D (independent of the layer count) and is counted as measured."""
    import ctypes as C
    import numpy as np
    from mom6_amd import _abi, synth
    from oracle import orc
    nk_s = 2
    g = synth.make_grid(grid.ni, grid.nj, nk_s, halo=grid.halo, seed=20241020)
    adv = synth.make_advection_state(g, ntr=NTR, seed=1, hot_frac=HOT_FRAC)
    dyn = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=11).items()}
    tr = [t.numpy() for t in adv["tr"]]
    h_end, uhtr, vhtr = adv["h_end"].numpy(), adv["uhtr"].numpy(), adv["vhtr"].numpy()
    cs = orc.continuity_cs(nk_s, g.Angstrom_H)
    vru = np.ones_like(dyn["u"]); vrv = np.ones_like(dyn["v"])
    hp = dyn["h"].copy(); uh = np.zeros_like(dyn["u"]); vh = np.zeros_like(dyn["v"])
    ucor, vcor = np.zeros_like(uh), np.zeros_like(vh)
    arrs, bt = orc.make_bt_cont(g, with_h=True)
    orc.continuity(g, cs, dyn["u"], dyn["v"], dyn["h"], hp, uh, vh, DT, visc_rem_u=vru, visc_rem_v=vrv, bt_cont=bt)
    uhbt = np.ascontiguousarray(uh.sum(0) * 1.02); vhbt = np.ascontiguousarray(vh.sum(0) * 0.98)
    E = orc.eos("WRIGHT"); pcs = orc.pressureforce_cs(g)
    PFu, PFv, pbce, eta = orc.pressureforce(g, pcs, E, dyn["h"], dyn["T"], dyn["S"])
    bcs, bcs_arrs = orc.barotropic_cs(g)
    orc.barotropic_init(g, bcs)
    orc.btcalc(g, bcs, dyn["h"], arrs["h_u"], arrs["h_v"])
    orc.set_dtbt(g, bcs, pbce=pbce, bt_cont=bt)
    bc_u = np.ascontiguousarray(np.clip(PFu, -3e-5, 3e-5) * g.mask2dCu); bc_v = np.ascontiguousarray(np.clip(PFv, -3e-5, 3e-5) * g.mask2dCv)
    taux = np.ascontiguousarray(0.1 * g.mask2dCu); tauy = np.ascontiguousarray(0.0 * g.mask2dCv)
    loop_s = C.c_double.in_dll(orc.lib(), "orc_btstep_loop_seconds")
    h_new = np.ascontiguousarray(dyn["h"] * 1.0)
    t_used, t_2d, cycles = 0.0, 0.0, 0

    def bts(etaav):
        nonlocal t_2d
        orc.btstep(g, bcs, dyn["u"], dyn["v"], eta, DT, bc_u, bc_v, taux, tauy, pbce, eta, ucor, vcor, vru, vrv, bt_cont=bt,
                   uh0=uh, vh0=vh, u_uh0=dyn["u"], v_vh0=dyn["v"], want_etaav=etaav)
        t_2d += loop_s.value

    while t_used < 12.0 and cycles < 3:
        t0 = time.perf_counter()
        for n in range(steps_per_advect):
            orc.pressureforce(g, pcs, E, dyn["h"], dyn["T"], dyn["S"])
            orc.continuity(g, cs, dyn["u"], dyn["v"], dyn["h"], hp, uh, vh, DT, visc_rem_u=vru, visc_rem_v=vrv, bt_cont=bt)
            orc.btcalc(g, bcs, dyn["h"], arrs["h_u"], arrs["h_v"])
            orc.bt_mass_source(g, bcs, dyn["h"], eta, True)
            bts(False)
            orc.continuity(g, cs, dyn["u"], dyn["v"], dyn["h"], hp, uh, vh, DT, uhbt=uhbt, vhbt=vhbt, visc_rem_u=vru,
                           visc_rem_v=vrv, u_cor=ucor, v_cor=vcor, bt_cont=bt)
            orc.coradcalc(g, ucor, vcor, hp, uh, vh, bound_coriolis=True)
            bts(True)
            orc.continuity(g, cs, dyn["u"], dyn["v"], dyn["h"], hp, uh, vh, DT, uhbt=uhbt, vhbt=vhbt, visc_rem_u=vru,
                           visc_rem_v=vrv, u_cor=ucor, v_cor=vcor)
            orc.coradcalc(g, ucor, vcor, hp, uh, vh, bound_coriolis=True)
        orc.advect_tracer(g, h_end, uhtr, vhtr, DT_THERM, DT, scheme, tr)
        orc.ale_remap_tracers(g, REMAP_SCHEME, dyn["h"], h_new, tr)
        t_used += time.perf_counter() - t0
        cycles += 1
    t_3d = t_used - t_2d
    sec_per_step = (t_3d / (g.ni * g.nj * nk_s) * full_cells + t_2d) / cycles / steps_per_advect
    return {
        "value": DT / sec_per_step / 365.0, "unit": "SYPD", "cores": 1, "kind": "port",
        "ns_per_gridpoint_step": sec_per_step * 1e9 / full_cells,
        "sample": f"{cycles} cycle(s) of {steps_per_advect} baroclinic steps (same calls as the GPU step) on "
                  f"{g.ni}x{g.nj}x{nk_s} (same horizontal grid, 2 of {grid.nk} layers); 3-D work ({t_3d:.1f} s) scaled per "
                  f"cell to the full grid, the 2-D barotropic subcycle ({t_2d:.1f} s, nstep={bcs.nstep_last}) counted as measured; "
                  f"{t_used:.1f} s of CPU",
    }


# algorithmic bytes per cell and call (SURVEY.md section 8d / DESIGN.md section 4)
ALG_BYTES = {
    "PressureForce": 64.0, "continuity[BT_cont]": 96.0, "continuity[uhbt+BT_cont]": 96.0, "continuity[uhbt]": 96.0,
    "CorAdCalc": 56.0, "CorAdCalc[pred]": 56.0, "ALE_remap_tracers": 16.0 + 16.0 * NTR,
}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py: --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    # MOM6HIP_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the N>1 path on a 1-GPU box
    backend = os.environ.get("MOM6HIP_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from mom6_amd import synth
    from mom6_amd.domains import Domain

    NI, NJ, NK = shape_of(a.workload)
    # N>1: the global grid is cut into `world` latitude bands (layout 1 x N, the x direction stays a local wrap);
    # halos travel between neighbouring GPUs through the reference's group passes (DESIGN.md "Multi-GPU").
    # The total work is fixed: strong scaling.
    grid = synth.make_grid(NI, NJ, NK, seed=20241020)
    dom = Domain(NI, NJ, (1, world), rank, grid.halo, grid.reentrant_x, grid.reentrant_y)
    S = Step(grid, dom, device, a.scheme)
    cells = NI * NJ * NK

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for n in range(a.warmup):
        S.run(n)
    S.dg.sync()
    barrier()
    t0 = time.perf_counter()
    for n in range(a.steps):
        S.run(n)
    S.dg.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    sec_per_step = elapsed / a.steps
    sypd = DT / sec_per_step / 365.0      # whole job: all ranks together advance the one global grid
    spa = S.steps_per_advect
    out = {
        "metric": "simulated-years/day (SYPD) of the hot-path kernels built so far",
        "value": sypd, "unit": "SYPD", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": sec_per_step * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "ns_per_gridpoint_step": sec_per_step * 1e9 / cells,
        "config": {
            "btstep_nstep": int(S.bt_cs.st.nstep_last), "dtbt_s": float(S.bt_cs.st.dtbt),
            "workload": f"{a.workload} {NI}x{NJ}x{NK} global C-grid, halo 4, reentrant-x, ~25% land, "
                        f"{NTR} tracers, DT={DT:.0f}s DT_THERM={DT_THERM:.0f}s",
            "kernels": {"PressureForce_FV_Bouss": "Wright EOS, PLM, 1/step", "continuity_PPM": "3/step (BT_cont; uhbt+BT_cont; uhbt)",
                        "CorAdCalc": "Sadourny75 energy + BOUND_CORIOLIS, 2/step",
                        "btstep": "BT_cont fits, layer fluxes, wide-halo march, 2/step (+ btcalc, bt_mass_source 1/step)",
                        "advect_tracer": f"{a.scheme}, 1 per {spa} steps",
                        "ALE_remap_tracers": f"{REMAP_SCHEME}, {NTR} tracers, 1 per {spa} steps"},
            "not_yet_in_step": ["RK2 momentum-update sweeps", "ALE regrid + velocity remap",
                                "vertvisc / horizontal_viscosity (SURVEY 8f)"],
            "advect_iterations_last_call": None if S.last_adv is None else int(S.last_adv.iterations),
            "parallelism": "1 tile" if world == 1 else f"layout 1x{world}: {world} latitude bands, one per GPU, "
                                                               "group passes over RCCL p2p",
        },
    }

    if world == 1 and not a.no_roofline:
        # per-call device time, HIP events on the stream the library launches on (the null stream, which is
        # also torch's current stream here)
        comp = {}
        for n in range(spa):
            for name, f in S.parts(n):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); f(); e1.record(); e1.synchronize()
                comp.setdefault(name, []).append(e0.elapsed_time(e1))
        comp_ms = {k: sum(v) / len(v) for k, v in comp.items()}
        per_step = {k: (sum(v) / spa) for k, v in comp.items()}
        out["components_ms_per_call"] = comp_ms
        out["components_ms_per_step"] = per_step
        # the dominant KERNEL: the dense (first-iteration) advect pass is measured per launch by the library's own
        # HIP events; the other operators are one-to-few kernels per call and are priced per call
        S.dg.set_timing(True)
        from mom6_amd.tracer_advect import advect_tracer
        ad = S.adv
        tx = ty = 0.0
        for _ in range(3):
            advect_tracer(ad["h_end"], ad["uhtr"], ad["vhtr"], None, DT_THERM, S.dg, S.adv_cs, ad["tr"])
            t = S.dg.advect_timing(); tx += t.ms_x1 / 3; ty += t.ms_y1 / 3
        S.dg.set_timing(False)
        cand = {k: (ALG_BYTES[k] * cells, ms) for k, ms in comp_ms.items() if k in ALG_BYTES}
        cand["adv_x_kernel<4,PPM:H3,first>"] = ((NTR + 2) * 16.0 * cells, tx)
        cand["adv_y_kernel<4,PPM:H3,first>"] = ((NTR + 2) * 16.0 * cells, ty)
        dom = max(per_step, key=per_step.get)
        dom_key = dom if dom in cand else ("adv_y_kernel<4,PPM:H3,first>" if ty >= tx else "adv_x_kernel<4,PPM:H3,first>")
        b, ms = cand[dom_key]
        out["roofline"] = {
            "kernel": dom_key, "bound": "hbm", "achieved": b / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_launch": b, "avg_launch_ms": ms,
            "all": {k: {"GBs": bb / (m * 1e-3) / 1e9, "frac": bb / (m * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms": m}
                    for k, (bb, m) in cand.items()},
        }

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(grid, a.scheme, cells, spa)

    if rank == 0:
        print(json.dumps(out))
    S.dg.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
