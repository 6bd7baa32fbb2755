#!/usr/bin/env python3
"""bench.py -- throughput of the MOM6 dynamical-core hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one baroclinic time step (DT) of the model on the synthetic global C-grid named in
`config.workload`: step_MOM_dyn_split_RK2 (src/core/MOM_dynamics_split_RK2.F90:289-1176) -- PressureForce,
continuity x3, btstep x2 (+ btcalc, bt_mass_source), CorAdCalc x2, horizontal_viscosity, vertvisc_coef x3 / vertvisc x2 /
vertvisc_remnant x3, the momentum sweeps and the group passes -- preceded every DT_THERM by set_viscous_BBL (MOM.F90:1205),
and every DT_THERM/DT-th step advect_tracer (src/core/MOM.F90:1438) and the ALE block (:1647-1700: ALE_regrid,
ALE_remap_tracers, ALE_remap_set_h_vel, ALE_remap_velocities).  The state
evolves: every step starts from the previous step's u, v, h, T, S.  Every operator of the reference's dynamic step on this configuration is
the library's own (nothing is prescribed or set to zero).  The state is resident in HBM before
the timed region.  Rank 0 prints ONE JSON line; `components_ms_per_call` times each operator on its own.
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
# vertvisc_init parameters: background viscosity, and the fixed-depth mixed-layer viscosity KV_ML_INVZ2 over HMIX_FIXED that
# carries the wind stress into the top 20 m (no boundary-layer scheme feeds visc%Kv_shear here)
VERTVISC = dict(KV=1.0e-4, HBBL=10.0, HMIX_FIXED=20.0, KV_ML_INVZ2=1.0e-2)
# hor_visc_init parameters: biharmonic Smagorinsky viscosity with a grid-scale background (the OM4_025 choice)
# tracer_hordiff after advect_tracer (step_MOM_tracer_dyn, MOM.F90:1438-1441): along-layer diffusion, KHTR = 50 m2 s-1
TRACER_HORDIFF = dict(KHTR=50.0, CHECK_DIFFUSIVE_CFL=True)
HOR_VISC = dict(BIHARMONIC=True, SMAGORINSKY_AH=True, SMAG_BI_CONST=0.06, AH_VEL_SCALE=0.01)
SET_VISC = dict(HBBL=10.0, KV=1.0e-4, CDRAG=0.003, BBL_USE_EOS=True)      # set_visc_init: the bottom boundary layer of set_viscous_BBL
HOT_FRAC = 2.0e-5
REGRID_OLD_WEIGHT = 0.0    # REGRID_TIME_SCALE = 0 (the reference's default): every ALE call regrids all the way to z*
PMC_PROFILE = "r05_c_pmc.json"
FP64_VECTOR_TFLOPS = 78.6            # 256 CUs x 4 SIMDs x 16 fp64 lanes per cycle x 2 (fma) x 2.4 GHz


def _pmc_kernel(prefix, source):
    """The counter record of a kernel from profiles/PMC_PROFILE, or None when `source` (a file of mom6_amd/csrc) has changed since the
    counters were taken (the profile carries the SHA-256 of every source).  Returns (record, sha)."""
    import hashlib
    prof = json.load(open(os.path.join(ROOT, "profiles", PMC_PROFILE)))
    sha = hashlib.sha256(open(os.path.join(ROOT, "mom6_amd", "csrc", source), "rb").read()).hexdigest()
    if prof.get("source_sha256", {}).get(source) != sha:
        return None, sha
    rec = [v for k, v in prof["kernels"].items() if k.startswith(prefix)]
    return (rec[0] if rec else None), sha      # the counter passes `roofline.traffic` is read from (profiles/)
LAND_FRAC = 0.30
# grid-scale bathymetric roughness (white noise, as a fraction of the depth range) with a fixed SLOPE: 0.04 on a 3-degree
# grid, 0.0025 (14 m rms) at 1/4 degree.  With the amplitude held at 0.04 the 1/4-degree bathymetry had 200 m steps between
# neighbouring cells and the run went to 5 m/s within 16 steps (tools/model_health.py).
def rough_noise(ni):
    return min(0.04, 3.6 / ni)


# the synthetic state (mom6_amd/synth.py): z* layers over the rough bathymetry -- every layer below the local bottom is
# vanished (Angstrom thick) --, a stratification that is a function of depth, so the state is close to rest balance
# Round 4: the horizontal T, S contrasts are a quarter of synth.py's defaults and confined to the upper ocean (e-folding depth 500 m):
# the full-depth contrasts of rounds 1-3 were far from thermal-wind balance with u, v and released potential energy for hundreds of
# steps (7 % per step of kinetic-energy growth in the timed window, 1.27 m/s and 2.96 m by step 240).  With these and SPINUP untimed
# steps before the warm-up the timed window grows by 0.3 % per step, and step 240 stands at 0.63 m/s and 2.0 m
# (profiles/r04_health_om4.json; tools/model_health.py --steps 240).
STATE = dict(umax=0.1, eta_amp=0.2, terrain_following=False, vanish_frac=0.0, h_noise=1.0e-3, ts_amp=0.25, ts_decay=500.0)
SPINUP = 48                # untimed steps before the warm-up (a multiple of DT_THERM / DT): the synthetic start adjusts to the wind and to itself


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="om4_025", help="om4_025 | benchmark | double_gyre | NIxNJxNK")
    ap.add_argument("--scheme", default=SCHEME)
    ap.add_argument("--spinup", type=int, default=None, help="untimed steps before the warm-up (default: SPINUP)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def attach_exchange(dom, dg, device, exchange):
    """The halo exchange of a multi-tile run: the library's own RCCL group passes ("rccl") or MOM6's group passes as callbacks
    into torch.distributed ("python").  attach_native is collective (broadcast of the unique id, ncclCommInitRank), so the
    ranks first agree, from a LOCAL probe, that every one of them can enter it; if any cannot, all use the callbacks."""
    if exchange == "rccl":
        import torch.distributed as dist
        ok = torch.tensor([1 if dom.native_available() else 0], device=str(device) if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            dom.attach_native(dg)
            return "rccl"
        print("bench.py: RCCL cannot be loaded on every rank; using torch.distributed callbacks", file=sys.stderr, flush=True)
    dg.set_domain(dom)
    return "python"


def exchange_preflight(dom, dg, device, exchange):
    """Before anything is timed on more than one GPU: one group pass of NaN-poisoned, rank-coded fields (h, u and v points, two layers)
    through the exchange the run is going to use, compared bit for bit with the same pass through the torch.distributed callbacks (the
    path the gloo layout tests cover).  Every rank takes the same decision (all_reduce of the verdict).  Returns a string for the
    JSON line: "ok", or what failed."""
    import torch.distributed as dist
    if os.environ.get("MOM6HIP_BENCH_PREFLIGHT", "1") == "0" or dom.nranks == 1:
        return "skipped"
    g = dg.grid
    dev = str(device)
    fields, pos = [], [0, 1, 2]
    for p_ in pos:
        shp = (2,) + g.shape2(p_)
        jj = torch.arange(shp[1], device=dev, dtype=torch.float64)[:, None] + dom.j0
        ii = torch.arange(shp[2], device=dev, dtype=torch.float64)[None, :] + dom.i0
        f = (1000.0 * jj + ii + 0.25 * p_)[None].repeat(2, 1, 1).contiguous()
        f[1] += 0.5
        h = g.halo
        poisoned = torch.full_like(f, float("nan"))
        xs = 1 if p_ == 1 else 0; ys = 1 if p_ == 2 else 0
        poisoned[:, h:h + dom.nj + ys, h:h + dom.ni + xs] = f[:, h:h + dom.nj + ys, h:h + dom.ni + xs]
        fields.append(poisoned)
    ref = [f.clone() for f in fields]
    was_native = getattr(dom, "native", False)
    try:
        dom.pass_var(fields, pos)                 # the exchange of the run
        dom.native = False
        dom.pass_var(ref, pos)                    # the callbacks
    finally:
        dom.native = was_native
    torch.cuda.synchronize()
    same = all(torch.equal(a.view(torch.int64), b.view(torch.int64)) for a, b in zip(fields, ref))
    filled = all(bool(torch.isfinite(a[:, g.halo:g.halo + dom.nj, :g.halo]).all()) for a in fields[:1]) if g.reentrant_x or dom.npi > 1 else True
    ok = torch.tensor([1 if (same and filled) else 0], device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    return "ok" if int(ok.item()) == 1 else f"FAILED on some rank (rank {dom.rank}: same={same}, filled={filled})"


def shape_of(name):
    from mom6_amd import synth
    if name in synth.CONFIGS:
        return synth.CONFIGS[name]
    return tuple(int(x) for x in name.lower().split("x"))


class Model:
    """The time-stepping model on one tile of the global grid: prognostic state, control structures, and step()."""

    def __init__(self, gg, dom, device, scheme, exchange="python"):
        from mom6_amd import _abi, synth
        from mom6_amd.ale import initialize_remapping
        from mom6_amd.dynamics_split_rk2 import initialize_dyn_split_RK2, initialize_dyn_split_RK2b
        from mom6_amd.tracer_advect import DeviceGrid, tracer_advect_init
        # MOM6HIP_BENCH_RK2B=1: SPLIT_RK2B = True (MOM_dynamics_split_RK2b.F90) instead of the default split scheme; not the default
        self.rk2b = os.environ.get("MOM6HIP_BENCH_RK2B", "0") == "1"
        if self.rk2b:
            initialize_dyn_split_RK2 = initialize_dyn_split_RK2b
        self.dom = dom
        grid = self.g = dom.tile_grid(gg) if dom.nranks > 1 else gg
        self.dg = DeviceGrid(grid, device=device.index)
        self.exchange = exchange
        if dom.nranks > 1:
            self.exchange = attach_exchange(dom, self.dg, device, exchange)
        dev = str(device)
        H, U, V = _abi.POS_H, _abi.POS_U, _abi.POS_V
        Z = lambda pos, k3=True: torch.zeros(grid.shape3(pos) if k3 else grid.shape2(pos), dtype=torch.float64, device=dev)
        # every rank generates the same global state on its own GPU and keeps its tile
        dyn = synth.make_dynamics_state(gg, seed=11, device=dev, **STATE)
        cut = lambda a, pos: dom.cut(a, pos).clone()
        self.u, self.v, self.h = cut(dyn["u"], U), cut(dyn["v"], V), cut(dyn["h"], H)
        self.T, self.S = cut(dyn["T"], H), cut(dyn["S"], H)
        del dyn
        if gg.tripolar_n:      # the generated state knows nothing of the fold: one face, one value on the fold line; folded halos
            if dom.on_fold or dom.nranks == 1:
                row = grid.halo + grid.nj
                self.v[:, row, :] = 0.5 * (self.v[:, row, :] - torch.flip(self.v[:, row, :], dims=(-1,)))
            fl, ps = [self.u, self.v, self.T, self.S, self.h], [1, 2, 0, 0, 0]
            (dom.pass_var(fl, ps) if dom.nranks > 1 else self.dg.halo_update(fl, ps))
        adv = synth.make_advection_state(gg, ntr=4, seed=1, device=dev, hot_frac=0.0)
        self.passive = [cut(t, H) for t in adv["tr"][2:4]]      # 2 passive tracers (a smooth blob and a step)
        del adv
        torch.cuda.empty_cache()
        self.uh, self.vh, self.uhtr, self.vhtr = Z(U), Z(V), Z(U), Z(V)
        self.eta_av = Z(H, False)
        mu = torch.as_tensor(grid.mask2dCu, device=dev)
        j0 = dom.j0 if dom.nranks > 1 else 0
        yy = (torch.arange(grid.shape2(U)[0], device=dev, dtype=torch.float64) + j0) / (gg.nj + 2 * gg.halo) * 3.1416
        self.taux = (0.1 * torch.cos(2 * yy)[:, None] * mu).contiguous()
        self.tauy = Z(V, False)
        self.CS = initialize_dyn_split_RK2(self.u, self.v, self.h, self.uh, self.vh, DT, self.dg, coriolis=dict(bound_coriolis=True),
                                           vertvisc=VERTVISC, hor_visc=HOR_VISC)
        # visc%Kv_bbl_[uv], visc%bbl_thick_[uv]: set by set_viscous_BBL at the start of every thermodynamic cycle (MOM.F90:1200-1208)
        from mom6_amd.pressure_force import EOS_init
        from mom6_amd.set_viscosity import set_visc_init
        from mom6_amd.vert_friction import vertvisc_type
        self.visc = vertvisc_type(Kv_bbl_u=Z(U, False), Kv_bbl_v=Z(V, False), bbl_thick_u=Z(U, False), bbl_thick_v=Z(V, False))
        self.set_visc_cs = set_visc_init(self.dg, **SET_VISC)
        self.eos = EOS_init("WRIGHT")
        self.adv_cs = tracer_advect_init(DT, scheme)
        from mom6_amd.tracer_hor_diff import tracer_hor_diff_init
        self.hordiff_cs = tracer_hor_diff_init(**TRACER_HORDIFF)
        self.remap_cs = initialize_remapping(REMAP_SCHEME)
        # z* target: the nominal layer thicknesses of the synthetic state (synth.make_dynamics_state)
        from mom6_amd.ale import initialize_regridding
        import numpy as _np
        kn = (_np.arange(grid.nk) + 0.5) / grid.nk
        dz_nom = 2.0 + 300.0 * kn ** 2
        self.regrid_cs = initialize_regridding(self.dg, coordinateResolution=dz_nom * (5500.0 / dz_nom.sum()),
                                               old_grid_weight=REGRID_OLD_WEIGHT)
        self.h_new = self.h.clone()
        self.dzRegrid = torch.zeros((grid.nk + 1,) + grid.shape2(H), dtype=torch.float64, device=dev)
        self.h_old_u, self.h_old_v, self.h_new_u, self.h_new_v = Z(U), Z(V), Z(U), Z(V)
        self.steps_per_advect = int(round(DT_THERM / DT))
        self.last_adv = None
        self.nstep = 0
        self.dg.sync()

    def step(self):
        from mom6_amd.ale import ALE_regrid, ALE_remap_set_h_vel, ALE_remap_tracers, ALE_remap_velocities
        from mom6_amd.dynamics_split_rk2 import step_MOM_dyn_split_RK2, step_MOM_dyn_split_RK2b
        from mom6_amd.tracer_advect import advect_tracer
        if self.rk2b:
            step_MOM_dyn_split_RK2 = step_MOM_dyn_split_RK2b
        n = self.nstep
        lat = getattr(self, "lateral", None)      # enable_lateral(): thickness_diffuse and mixedlayer_restrat in the cycle (not the default workload)
        if lat and n % self.steps_per_advect == 0:      # THICKNESSDIFFUSE_FIRST (MOM.F90:1149-1181)
            from mom6_amd.thickness_diffuse import thickness_diffuse
            thickness_diffuse(self.h, self.uhtr, self.vhtr, (self.T, self.S, self.eos), DT_THERM, self.dg, None, None, None, lat["td"])
            self.dg.halo_update([self.h], [0]) if self.dom.nranks == 1 else self.dom.pass_var([self.h], [0])
        if n % self.steps_per_advect == 0:      # bbl_time_int > 0: the first dynamic step of a thermodynamic cycle (MOM.F90:1200)
            from mom6_amd.set_viscosity import set_viscous_BBL
            set_viscous_BBL(self.u, self.v, self.h, (self.T, self.S, self.eos), self.visc, self.dg, self.set_visc_cs)
        step_MOM_dyn_split_RK2(self.u, self.v, self.h, (self.T, self.S), self.visc, None, DT, (self.taux, self.tauy), None, None,
                               self.uh, self.vh, self.uhtr, self.vhtr, self.eta_av, self.dg, self.CS, calc_dtbt=(n == 0))
        if lat and (n + 1) % self.steps_per_advect == 0:      # MOM.F90:1335-1338
            from mom6_amd.mixedlayer_restrat import mixedlayer_restrat
            mixedlayer_restrat(self.h, self.uhtr, self.vhtr, (self.T, self.S, self.eos), dict(ustar=lat["ustar"]), DT_THERM, None, None, None, None,
                               self.dg, lat["mle"])
            self.dg.halo_update([self.h], [0]) if self.dom.nranks == 1 else self.dom.pass_var([self.h], [0])
        if (n + 1) % self.steps_per_advect == 0:      # step_MOM_thermo / step_MOM_tracer_dyn (src/core/MOM.F90:1438, :1662)
            tr = [self.T, self.S] + self.passive
            self.last_adv = advect_tracer(self.h, self.uhtr, self.vhtr, None, DT_THERM, self.dg, self.adv_cs, tr)
            from mom6_amd.tracer_hor_diff import tracer_hordiff
            tv = dict(T=self.T, S=self.S, eqn_of_state=self.eos) if lat and lat.get("neutral") else None      # USE_NEUTRAL_DIFFUSION
            self.last_hordiff = tracer_hordiff(self.h, DT_THERM, None, None, None, self.dg, self.hordiff_cs, tr, tv=tv)      # MOM.F90:1441
            self.uhtr.zero_(); self.vhtr.zero_()
            # ALE (src/core/MOM.F90:1647-1700): regrid, remap tracers, remap velocities, adopt the new grid
            dg = self.dg
            ALE_regrid(dg, self.h, self.h_new, self.dzRegrid, None, self.regrid_cs)
            ALE_remap_tracers(self.remap_cs, dg, self.h, self.h_new, tr)
            ALE_remap_set_h_vel(None, dg, self.h, self.h_old_u, self.h_old_v)
            ALE_remap_set_h_vel(None, dg, self.h_new, self.h_new_u, self.h_new_v)
            ALE_remap_velocities(self.remap_cs, dg, self.h_old_u, self.h_old_v, self.h_new_u, self.h_new_v, self.u, self.v)
            self.h, self.h_new = self.h_new, self.h           # h(:,:,:) = h_new(:,:,:) on js-1:je+1 (:1697)
            fields, pos = [self.u, self.v, self.T, self.S, self.h], [1, 2, 0, 0, 0]      # pass_uv_T_S_h (:1713-1719)
            if self.dom.nranks == 1:
                dg.halo_update(fields, pos)
            else:
                self.dom.pass_var(fields, pos)
        self.nstep += 1

    def enable_lateral(self, KHTH=600.0, FOX_KEMPER_ML_RESTRAT_COEF=5.0, neutral=False):
        """thickness_diffuse (KHTH) before the first dynamic step and mixedlayer_restrat (MLE_DENSITY_DIFF) after the last one of every
        thermodynamic cycle: tools/model_health.py --lateral, MOM6HIP_BENCH_LATERAL=1; neutral (MOM6HIP_BENCH_LATERAL=2, --neutral): tracer_hordiff
        with USE_NEUTRAL_DIFFUSION as well"""
        from mom6_amd.mixedlayer_restrat import mixedlayer_restrat_init
        from mom6_amd.thickness_diffuse import thickness_diffuse_init
        rho0 = float(self.g.Rho0)
        tx = 0.5 * (self.taux[:, 1:] + self.taux[:, :-1]); ty = 0.5 * (self.tauy[1:, :] + self.tauy[:-1, :])
        ustar = torch.sqrt(torch.sqrt(tx * tx + ty * ty) / rho0).contiguous()      # forces%ustar from the wind stress at h points
        self.lateral = dict(td=thickness_diffuse_init(self.dg, THICKNESSDIFFUSE=True, KHTH=KHTH),
                            mle=mixedlayer_restrat_init(self.dg, FOX_KEMPER_ML_RESTRAT_COEF=FOX_KEMPER_ML_RESTRAT_COEF), ustar=ustar, neutral=bool(neutral))
        if neutral:
            from mom6_amd.tracer_hor_diff import tracer_hor_diff_init
            self.hordiff_cs = tracer_hor_diff_init(USE_NEUTRAL_DIFFUSION=True, **TRACER_HORDIFF)

    def health(self):
        """max |u|, max |v|, min h, max |eta|, kinetic energy per unit area, the vanished fraction, any NaN -- the bench
        refuses to report a number for a state that blew up or is on a growing trajectory"""
        bad = not bool(torch.isfinite(self.u).all() and torch.isfinite(self.h).all() and torch.isfinite(self.T).all())
        g = self.g
        sj, si = g.csl(0)
        mT = torch.as_tensor(g.mask2dT, device=self.u.device)[sj, si]
        hc = self.h[:, sj, si]
        ke = 0.25 * (hc * (self.u[:, sj, si.start:si.stop] ** 2 + self.u[:, sj, si.start + 1:si.stop + 1] ** 2
                           + self.v[:, sj.start:sj.stop, si] ** 2 + self.v[:, sj.start + 1:sj.stop + 1, si] ** 2)).sum(0)
        nocean = float(mT.sum())
        # z* regridding leaves the layers below the bottom at MIN_THICKNESS = 1e-3 (not at Angstrom): count h <= 2 MIN_THICKNESS
        vanished = float(((hc <= 2.0e-3) * mT[None]).sum()) / max(nocean * g.nk, 1.0)
        out = dict(umax=float(self.u.abs().max()), vmax=float(self.v.abs().max()), hmin=float(self.h.min()),
                   eta_max=float((self.CS.eta[sj, si] * mT).abs().max()), ke_mean=float((ke * mT).sum()) / max(nocean, 1.0),
                   vanished_layer_fraction=vanished, nan=bad)
        if not bad:
            # what MOM6 would write to ocean.stats for this state (write_energy, MOM_sum_output.F90:428), formed on the device as
            # order-invariant sums: the same numbers on any number of GPUs
            from mom6_amd.sum_output import write_energy
            e = write_energy(self.u, self.v, self.h, (self.T, self.S), self.dg, DT, H_to_kg_m2=float(self.g.Rho0))
            cp = 3991.86795711963
            out["ocean_stats"] = {"En_KE_per_mass": e["KE_tot"] / e["mass_tot"], "max_CFL": e["max_CFL"], "mass_kg": e["mass_tot"],
                                  "mean_salin": e["Salt"] / e["mass_tot"], "mean_temp": e["Heat"] / (e["mass_tot"] * cp),
                                  "mass_EFP": e["mass_EFP"]}
        return out


class Components:
    """The operators of the step called one by one on a stationary synthetic state (for the per-operator timings).  `dom` is the tile's Domain
    (mom6_amd/domains.py); every rank generates the same global synthetic state on its own GPU, keeps the
    window of its tile and frees the rest, so N ranks step exactly the problem one rank steps."""

    def __init__(self, gg, dom, device, scheme, exchange="python"):
        from mom6_amd import _abi, synth
        from mom6_amd.ale import initialize_remapping
        from mom6_amd.continuity import BT_cont_type, continuity_PPM_init
        from mom6_amd.coriolis_adv import CoriolisAdv_init
        from mom6_amd.pressure_force import EOS_init, PressureForce_init
        from mom6_amd.tracer_advect import DeviceGrid, tracer_advect_init
        self.dom = dom
        grid = self.g = dom.tile_grid(gg) if dom.nranks > 1 else gg
        self.dg = DeviceGrid(grid, device=device.index)
        self.exchange = exchange
        if dom.nranks > 1:
            self.exchange = attach_exchange(dom, self.dg, device, exchange)
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
        # the vertical viscosity of the step: vertvisc_coef + vertvisc_remnant (:598-600), then twice vertvisc_coef +
        # vertvisc + vertvisc_remnant (:717-744, :974-994); velocities are copies so the bench state stays stationary
        from mom6_amd.vert_friction import vertvisc, vertvisc_coef, vertvisc_remnant
        if not hasattr(self, "vv_cs"):
            from mom6_amd.vert_friction import vertvisc_init, vertvisc_type
            self.vv_cs = vertvisc_init(dg, **VERTVISC)
            g = self.g
            from mom6_amd.hor_visc import hor_visc_init
            from mom6_amd.set_viscosity import set_visc_init, set_viscous_BBL
            Zf = lambda pos: torch.zeros(g.shape2(pos), dtype=torch.float64, device=d["u"].device)
            self.vv_visc = vertvisc_type(Kv_bbl_u=Zf(1), Kv_bbl_v=Zf(2), bbl_thick_u=Zf(1), bbl_thick_v=Zf(2))
            self.sv_cs = set_visc_init(dg, **SET_VISC)
            set_viscous_BBL(d["u"], d["v"], d["h"], (d["T"], d["S"], self.eos), self.vv_visc, dg, self.sv_cs)
            self.hv_cs = hor_visc_init(dg, DT, **HOR_VISC)
            self.diffu, self.diffv = torch.zeros_like(d["u"]), torch.zeros_like(d["v"])
            self.vv_u, self.vv_v = d["u"].clone(), d["v"].clone()
            self.vv_ru, self.vv_rv = torch.zeros_like(d["u"]), torch.zeros_like(d["v"])

        # the fused form the RK2 step itself calls (mom6hip_vertvisc_step: vertvisc_coef, vertvisc, vertvisc_remnant of one seam in
        # one launch per direction); the velocity copies are part of the probe, not of the model step
        from mom6_amd.vert_friction import vertvisc_step

        def vv_full(dt):
            self.vv_u.copy_(d["u"]); self.vv_v.copy_(d["v"])
            vertvisc_step(self.vv_u, self.vv_v, d["h"], None, (self.taux, self.tauy), self.vv_visc, dt, dg, self.vv_cs, self.vv_ru, self.vv_rv)
        out.append(("vertvisc_coef+remnant", lambda: vertvisc_step(d["u"], d["v"], d["h"], None, None, self.vv_visc, DT, dg, self.vv_cs,
                                                                 self.vv_ru, self.vv_rv, update_velocities=False)))
        out.append(("vertvisc[pred]", lambda: vv_full(0.6 * DT)))
        out.append(("vertvisc[corr]", lambda: vv_full(DT)))
        from mom6_amd.hor_visc import horizontal_viscosity
        out.append(("horizontal_viscosity", lambda: horizontal_viscosity(d["u"], d["v"], d["h"], self.diffu, self.diffv, None, None, dg,
                                                                       self.hv_cs)))
        if n % self.steps_per_advect == 0:
            from mom6_amd.set_viscosity import set_viscous_BBL
            out.append(("set_viscous_BBL", lambda: set_viscous_BBL(d["u"], d["v"], d["h"], (d["T"], d["S"], self.eos), self.vv_visc, dg,
                                                                   self.sv_cs)))
        if (n + 1) % self.steps_per_advect == 0:
            a = self.adv

            def adv():
                self.last_adv = advect_tracer(a["h_end"], a["uhtr"], a["vhtr"], None, DT_THERM, dg, self.adv_cs, a["tr"])
            out.append(("advect_tracer", adv))

            def hordiff():
                from mom6_amd.tracer_hor_diff import tracer_hor_diff_init, tracer_hordiff
                if not hasattr(self, "hordiff_cs"):
                    self.hordiff_cs = tracer_hor_diff_init(**TRACER_HORDIFF)
                tracer_hordiff(a["h_end"], DT_THERM, None, None, None, dg, self.hordiff_cs, a["tr"])
            out.append(("tracer_hordiff", hordiff))
            out.append(("ALE_remap_tracers", lambda: ALE_remap_tracers(self.remap_cs, dg, d["h"], self.h_new, a["tr"])))
        return out

    def run(self, n):
        for _, f in self.parts(n):
            f()


def _cpu_sample(grid, scheme, nk_s, steps_per_advect, threads):
    """One timed sample of the CPU oracle on the benchmark's horizontal grid with nk_s layers: the first step (sets DTBT),
    then min(2, steps_per_advect) timed baroclinic steps and one thermodynamic block (advect_tracer + ALE regrid / remap)
    -- the calls of the GPU step.  Returns (seconds per dynamic step spent in 3-D work, in the 2-D barotropic subcycle,
    seconds of the thermodynamic block, nstep, threads used, CPU seconds)."""
    import ctypes as C
    import numpy as np
    from mom6_amd import _abi, synth
    from oracle import orc
    used = orc.set_threads(threads)
    g = synth.make_grid(grid.ni, grid.nj, nk_s, halo=grid.halo, seed=20241020, land_frac=LAND_FRAC, rough_noise=rough_noise(grid.ni))
    dyn = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=11, **STATE).items()}
    adv = synth.make_advection_state(g, ntr=4, seed=1, hot_frac=0.0)
    passive = [t.numpy() for t in adv["tr"][2:4]]
    del adv
    vv = orc.vertvisc_cs(g, Kv=VERTVISC["KV"], Hbbl=VERTVISC["HBBL"], Hmix=VERTVISC["HMIX_FIXED"], Kvml_invZ2=VERTVISC["KV_ML_INVZ2"])
    visc = orc.vertvisc_type(Kv_bbl_u=g.zeros2(_abi.POS_U), Kv_bbl_v=g.zeros2(_abi.POS_V), bbl_thick_u=g.zeros2(_abi.POS_U),
                             bbl_thick_v=g.zeros2(_abi.POS_V))
    svcs = orc.set_visc_cs(g, SET_VISC["HBBL"], SET_VISC["KV"], cdrag=SET_VISC["CDRAG"], BBL_use_EOS=SET_VISC["BBL_USE_EOS"])
    hvcs = orc.hor_visc_cs(g, DT, biharmonic=1, Smagorinsky_Ah=1, Smag_bi_const=HOR_VISC["SMAG_BI_CONST"], Ah_vel_scale=HOR_VISC["AH_VEL_SCALE"])
    orc.set_viscous_BBL(g, svcs, dyn["u"], dyn["v"], dyn["h"], dyn["T"], dyn["S"], orc.eos("WRIGHT"), visc)
    st = orc.DynState(g, dyn["u"], dyn["v"], dyn["h"], dyn["T"], dyn["S"], DT, vertvisc=vv, visc=visc, hor_visc=hvcs)
    yy = np.arange(g.shape2(_abi.POS_U)[0]) / (g.nj + 2 * g.halo) * 3.1416
    taux = np.ascontiguousarray(0.1 * np.cos(2 * yy)[:, None] * g.mask2dCu); tauy = g.zeros2(_abi.POS_V)
    loop_s = C.c_double.in_dll(orc.lib(), "orc_btstep_loop_seconds")
    st.step(taux, tauy, calc_dtbt=True)        # the first step sets DTBT, as on the GPU
    n_dyn = min(2, steps_per_advect)
    loop_s.value = 0.0
    t0 = time.perf_counter()
    orc.set_viscous_BBL(g, svcs, st.u, st.v, st.h, st.T, st.S, orc.eos("WRIGHT"), visc)      # once per thermodynamic cycle
    t_bbl = time.perf_counter() - t0
    t0 = time.perf_counter()
    for n in range(n_dyn):
        st.step(taux, tauy)
    t_dyn = time.perf_counter() - t0 + t_bbl * n_dyn / steps_per_advect
    t_2d = loop_s.value
    t0 = time.perf_counter()
    tr = [st.T, st.S] + passive
    orc.advect_tracer(g, st.h, st.uhtr, st.vhtr, DT_THERM, DT, scheme, tr)
    orc.tracer_hordiff(g, st.h, DT_THERM, tr, TRACER_HORDIFF["KHTR"], check_diffusive_CFL=TRACER_HORDIFF["CHECK_DIFFUSIVE_CFL"])
    kn = (np.arange(nk_s) + 0.5) / nk_s
    dz_nom = 2.0 + 300.0 * kn ** 2
    rcs = orc.regridding_cs(dz_nom * (5500.0 / dz_nom.sum()), old_grid_weight=REGRID_OLD_WEIGHT)
    h_new, dzr = orc.ale_regrid(g, rcs, st.h)
    orc.ale_remap_tracers(g, REMAP_SCHEME, st.h, h_new, tr)
    hou, hov = orc.ale_remap_set_h_vel(g, st.h); hnu, hnv = orc.ale_remap_set_h_vel(g, h_new)
    orc.ale_remap_velocities(g, REMAP_SCHEME, hou, hov, hnu, hnv, st.u, st.v)
    for f, ps in ((st.u, 1), (st.v, 2), (st.T, 0), (st.S, 0), (h_new, 0)):
        orc.halo_update(g, f, ps)
    t_thermo = time.perf_counter() - t0
    nstep = int(st.bcs.nstep_last)
    orc.set_threads(1)
    return dict(t3d=(t_dyn - t_2d) / n_dyn, t2d=t_2d / n_dyn, thermo=t_thermo, nstep=nstep, threads=used, cpu_s=t_dyn + t_thermo,
                n_dyn=n_dyn, cells=g.ni * g.nj * nk_s)


def _cpu_quota():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (a box whose mask lists every
    core of the host may still be entitled to a few of them)."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def _pick_threads(grid, limit):
    """The OpenMP thread count the all-cores sample runs on: the fastest of 4, 8, 16, ... <= limit for one continuity call
    on two layers of the benchmark grid (the sweep stops once a count is slower than the best so far; an over-subscribed
    box -- more threads than its CPU share -- is several times slower than one core in the 2-D subcycle)."""
    import numpy as np
    from mom6_amd import _abi, synth
    from oracle import orc
    g = synth.make_grid(grid.ni, grid.nj, 2, halo=grid.halo, seed=20241020, land_frac=LAND_FRAC, rough_noise=rough_noise(grid.ni))
    d = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=11, **STATE).items()}
    cs = orc.continuity_cs(2)
    cands = [n for n in (4, 8, 16, 32, 64, 128, 256, 512) if n < limit] + [limit]
    best, tbest, sweep = 1, None, {}
    for n in cands:
        orc.set_threads(n)
        h = np.zeros_like(d["h"]); uh = g.zeros3(_abi.POS_U); vh = g.zeros3(_abi.POS_V)
        orc.continuity(g, cs, d["u"], d["v"], d["h"], h, uh, vh, DT)          # warms the thread pool
        t0 = time.perf_counter()
        for _ in range(2):
            orc.continuity(g, cs, d["u"], d["v"], d["h"], h, uh, vh, DT)
        t = (time.perf_counter() - t0) / 2
        sweep[n] = round(t, 4)
        if tbest is None or t < tbest:
            best, tbest = n, t
        elif t > 1.3 * tbest:
            break
    orc.set_threads(1)
    return best, sweep


def _port_calibration():
    """How the port's time compares with the reference's own kernels (build container measurements, committed under profiles/):
    the unmodified MOM_continuity_PPM.F90, MOM_CoriolisAdv.F90, MOM_tracer_advect.F90 compiled in place with amdflang -O2 against the
    stand-ins of tests/fortran/stubs and timed on the port's inputs (tools/calibrate_ref_kernels.py; bitwise equal to the port), and
    PLM_reconstruction from oracle/_ref (tools/calibrate_ref.py).  port_over_reference > 1: the port is the slower of the two."""
    out = {"PLM_reconstruction": {"port_over_reference_1thr": 1.28, "source": "profiles/r02_calibrate_ref.json"}}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r05_calibrate_ref.json")
    try:
        with open(path) as f:
            cal = json.load(f)
        for k, v in cal["kernels"].items():
            out[k] = {n: round(x, 3) for n, x in v.items() if n.startswith("port_over_reference")}
            out[k]["source"] = "profiles/r05_calibrate_ref.json"
        out["grid"] = cal["grid"]; out["threads"] = cal["threads"]
        out["note"] = ("the reference files are compiled unmodified against hand-written stand-ins for MOM_grid / MOM_domains / MOM_file_parser "
                       "(FMS is not vendored): a calibration of the port, not a reference build; the bench's baseline stays kind = port")
    except (OSError, KeyError, ValueError):
        pass
    # the whole step: the reference's own dynamical core (MOM_dynamics_split_RK2.F90 and the 18 files it steps through, in place, amdflang -O2)
    # against DynState.step (tools/calibrate_ref_core.py; bitwise equal once the oracle takes the host's libm pow in btstep's one power)
    try:
        with open(os.path.join(os.path.dirname(path), "r05_calibrate_ref_core.json")) as f:
            core = json.load(f)
        out["step_MOM_dyn_split_RK2"] = {"port_over_reference_1thr": core["port_over_reference_1thr"], "grid": core["grid"],
                                         "ns_per_gridpoint_step": core["ns_per_gridpoint_step"],
                                         "reference_equals_port_bitwise_with_libm_pow": core["reference_O2_against_oracle_with_libm_pow"]["bitwise_equal"],
                                         "source": "profiles/r05_calibrate_ref_core.json"}
    except (OSError, KeyError, ValueError):
        pass
    return out


def cpu_baseline(grid, scheme, full_cells, steps_per_advect):
    """The CPU oracle (oracle/*.c, a C restatement of the reference routines; kind "port") timed on the GPU box's host cores
    on bounded samples of the same workload (the same horizontal grid with fewer layers; the GPU step's calls):
      * all cores: libmom6oracle_omp.so (OpenMP over the loops the reference marks !$OMP: k / j loops of continuity,
        CorAdCalc, PressureForce, btstep, advect_tracer, the ALE and vertvisc columns) on 16 of the layers -> `value`, on
        the thread count that a short sweep finds fastest within the box's CPU share (`cores`; `host_cores` is the
        affinity mask, `cpu_quota` the cgroup share);
      * one core: the scalar oracle on 2 of the layers -> `one_core`.
    The 3-D work is scaled per cell to the full grid; the barotropic subcycle is 2-D (independent of the layer count) and is
    counted as measured.  The reference Fortran itself cannot be built here (FMS is not vendored), so the calibration of this
    port against a flang build of the reference exists only for the PLM/PCM pieces of oracle/_ref."""
    host_cores = len(os.sched_getaffinity(0))
    quota = _cpu_quota()

    def rate(smp):
        scale = full_cells / smp["cells"]
        sec = smp["t3d"] * scale + smp["t2d"] + smp["thermo"] * scale / steps_per_advect
        return DT / sec / 365.0, sec

    one = _cpu_sample(grid, scheme, 2, steps_per_advect, 1)
    sy1, sec1 = rate(one)
    out = {"unit": "SYPD", "kind": "port", "host_cores": host_cores, "cpu_quota": quota,
           "compiler": "gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math (+ -fopenmp for the all-cores build)",
           "calibration_vs_reference_build": _port_calibration(),
           "one_core": {"value": sy1, "cores": 1, "ns_per_gridpoint_step": sec1 * 1e9 / full_cells,
                        "sample": f"{one['n_dyn']} baroclinic steps + one advect_tracer / ALE block on {grid.ni}x{grid.nj}x2 (2 of {grid.nk} "
                                  f"layers), 3-D work scaled per cell, the 2-D barotropic subcycle ({one['t2d']:.1f} s per step, "
                                  f"nstep={one['nstep']}) as measured; {one['cpu_s']:.1f} s of CPU"}}
    nk_all = min(16, grid.nk)
    if quota > 1:
        nthreads, sweep = _pick_threads(grid, quota)
        out["thread_sweep_seconds_per_continuity_call"] = sweep
        al = _cpu_sample(grid, scheme, nk_all, steps_per_advect, nthreads)
        sya, seca = rate(al)
        out.update({"value": sya, "cores": al["threads"], "ns_per_gridpoint_step": seca * 1e9 / full_cells,
                    "sample": f"{al['n_dyn']} baroclinic steps + one advect_tracer / ALE block on {grid.ni}x{grid.nj}x{nk_all} ({nk_all} of "
                              f"{grid.nk} layers) on {al['threads']} OpenMP threads, 3-D work scaled per cell, the 2-D barotropic "
                              f"subcycle ({al['t2d']:.1f} s per step, nstep={al['nstep']}) as measured; {al['cpu_s']:.1f} s of wall clock"})
    else:
        out.update({"value": sy1, "cores": 1, "ns_per_gridpoint_step": sec1 * 1e9 / full_cells, "sample": out["one_core"]["sample"]})
    return out


# algorithmic bytes per cell and call (SURVEY.md section 8d / DESIGN.md section 4)
ALG_BYTES = {
    "PressureForce": 64.0, "continuity[BT_cont]": 96.0, "continuity[uhbt+BT_cont]": 96.0, "continuity[uhbt]": 96.0,
    "CorAdCalc": 56.0, "CorAdCalc[pred]": 56.0, "ALE_remap_tracers": 16.0 + 16.0 * NTR, "horizontal_viscosity": 40.0, "tracer_hordiff": 8.0 + 16.0 * NTR,
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
    # MOM6HIP_BENCH_TRIPOLAR=1: the same shape with TRIPOLAR_N connectivity (the topology of OM4: the northern edge folds onto
    # itself, open ocean on the fold) instead of a closed northern edge; not the default, so that the line stays comparable
    TRIPOLAR = os.environ.get("MOM6HIP_BENCH_TRIPOLAR", "0") == "1"
    TOPO = ", TRIPOLAR_N" if TRIPOLAR else ""
    if TRIPOLAR:
        grid = synth.fold_of(synth.make_grid(NI, 2 * NJ, NK, seed=20241020, land_frac=LAND_FRAC, rough_noise=rough_noise(NI), fold_symmetric=True))
    else:
        grid = synth.make_grid(NI, NJ, NK, seed=20241020, land_frac=LAND_FRAC, rough_noise=rough_noise(NI))
    dom = Domain(NI, NJ, (1, world), rank, grid.halo, grid.reentrant_x, grid.reentrant_y, tripolar_n=TRIPOLAR)
    # the halo exchange of the N>1 run: "rccl" = the library's native group pass (mom6_amd/csrc/domain_rccl.hip; needs one GPU
    # per rank), "python" = callbacks into torch.distributed (the gloo rehearsal, or MOM6HIP_BENCH_EXCHANGE=python)
    exchange = os.environ.get("MOM6HIP_BENCH_EXCHANGE", "rccl" if backend == "nccl" else "python") if world > 1 else None
    M = Model(grid, dom, device, a.scheme, exchange=exchange)
    exchange = M.exchange
    preflight = exchange_preflight(dom, M.dg, device, exchange) if world > 1 else "one tile"
    if world > 1 and preflight not in ("ok", "skipped"):
        sys.exit(f"bench.py: the halo exchange ({exchange}) failed its preflight on {world} ranks: {preflight}")
    LATERAL = os.environ.get("MOM6HIP_BENCH_LATERAL", "0") in ("1", "2")      # not the default: the workload of BASELINE.json does not call them
    NEUTRAL = os.environ.get("MOM6HIP_BENCH_LATERAL", "0") == "2"
    if LATERAL:
        M.enable_lateral(neutral=NEUTRAL)
    cells = NI * NJ * NK

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    health0 = M.health()
    for n in range(SPINUP if a.spinup is None else a.spinup):
        M.step()
    for n in range(a.warmup):
        M.step()
    M.dg.sync()
    health_w = M.health()
    barrier()
    M.dg.kernel_timing(True)            # HIP events around the dominant kernel's launches, on the library's stream
    if exchange == "rccl":
        dom.exchange_timing(True)       # HIP events around every group pass, on the library's communication stream
    t0 = time.perf_counter()
    for n in range(a.steps):
        M.step()
    M.dg.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    ktime = M.dg.kernel_timing(False)
    exch = None
    if dist is not None:
        on = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], device=on, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if exchange == "rccl":          # per rank: milliseconds per step inside group passes (pack, send / recv, unpack), passes per step
            ms, npass = dom.exchange_timing(False)
            mine = torch.tensor([ms / a.steps, npass / a.steps], device=on, dtype=torch.float64)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            exch = {"kind": "native RCCL send/recv on the library's communication stream (mom6hip_domain_init_rccl)",
                    "ms_per_step_per_rank": [round(float(e[0]), 3) for e in every],
                    "group_passes_per_step": float(every[0][1])}
        else:
            exch = {"kind": f"torch.distributed callbacks ({backend})"}
    health = M.health()
    from mom6_amd.vert_friction import vertvisc_ntrunc
    ntrunc = int(vertvisc_ntrunc(M.dg, M.CS.vertvisc_CSp))      # CS%ntrunc: velocity truncations during the run
    # growth over the timed steps is refused as well as NaNs (a number from a run that is blowing up is not a benchmark): the
    # largest speed may not pass 1 m/s nor 4x its value after the warm-up, the free surface may not pass 3 m nor 3x its value after
    # the warm-up (+ 0.5 m), no velocity may have been truncated, and the kinetic energy per mass (write_energy) may not grow by more
    # than 3 % per step over the timed window nor pass 1e-2 m2 s-2.  What growth there is in this workload is the release of the
    # available potential energy of the synthetic T(y), S(x,y) fields while the state adjusts: 4 % per step in the first steps, 0.3 %
    # per step after the SPINUP untimed steps (profiles/r04_health_om4.json); a run that skips the spin-up (--spinup 0: the profiling
    # tools, which only want the kernels) is held to 12 % per step instead.
    speed = max(health["umax"], health["vmax"]); speed_w = max(health_w["umax"], health_w["vmax"])
    growing = speed > max(1.0, 4.0 * speed_w) or health["eta_max"] > min(3.0, 3.0 * health_w["eta_max"] + 0.5) or ntrunc > 0
    ke_growth = None
    if not health["nan"] and "ocean_stats" in health and "ocean_stats" in health_w:
        ke_w, ke_e = health_w["ocean_stats"]["En_KE_per_mass"], health["ocean_stats"]["En_KE_per_mass"]
        ke_growth = (ke_e / ke_w) ** (1.0 / max(a.steps, 1)) - 1.0 if ke_w > 0 else 0.0
        # (the rate gate is calibrated on the two global workloads; a 44 x 40 x 2 basin adjusts to its synthetic start within a few steps
        # at 20 % per step and is held to the absolute bounds only)
        # (the folded world of MOM6HIP_BENCH_TRIPOLAR=1 has its open ocean on the fold and adjusts faster in its first dozen steps: 14 %)
        rate_gate = 0.20 if TRIPOLAR else (0.03 if (a.spinup is None or a.spinup >= SPINUP) else 0.12)
        growing = growing or (ke_growth > rate_gate and cells >= 1000000) or ke_e > 1.0e-2
    if health["nan"] or health["hmin"] < 0.0 or growing:
        sys.exit(f"bench.py: the model state is not healthy after {M.nstep} steps (KE growth per step {ke_growth}): start {health0}, "
                 f"after warm-up {health_w}, at the end {health}")

    sec_per_step = elapsed / a.steps
    sypd = DT / sec_per_step / 365.0      # whole job: all ranks together advance the one global grid
    spa = M.steps_per_advect
    bcs = M.CS.barotropic_CSp.st
    out = {
        "metric": "simulated-years/day (SYPD) of the split-RK2 dynamical core + tracer advection + ALE remap",
        "value": sypd, "unit": "SYPD", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": sec_per_step * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "ns_per_gridpoint_step": sec_per_step * 1e9 / cells,
        "config": {
            "workload": f"{a.workload} {NI}x{NJ}x{NK} global C-grid, halo 4, reentrant-x{TOPO}, {100 * LAND_FRAC:.0f}% land, rough bathymetry, z* "
                        f"layers ({100 * health['vanished_layer_fraction']:.0f}% of the ocean cells are vanished layers below the bottom), "
                        f"T, S + 2 passive tracers, DT={DT:.0f}s DT_THERM={DT_THERM:.0f}s",
            "step": ("step_MOM_dyn_split_RK2b [SPLIT_RK2B: continuity_PPM x4, horizontal_viscosity x2] " if M.rk2b else "") +
                    "step_MOM_dyn_split_RK2 (1 library call: PressureForce_FV_Bouss [Wright, PLM], continuity_PPM x3, "
                    "btstep x2 + btcalc + bt_mass_source, CorAdCalc x2 [Sadourny75 energy, BOUND_CORIOLIS], horizontal_viscosity "
                    "[biharmonic Smagorinsky, BETTER_BOUND_AH], vertvisc_coef x3 + vertvisc x2 + vertvisc_remnant x3 [BOTTOMDRAGLAW], "
                    f"momentum sweeps, group passes); every {spa} steps set_viscous_BBL [BBL_USE_EOS], advect_tracer [{a.scheme}], "
                    f"tracer_hordiff [KHTR={TRACER_HORDIFF['KHTR']:.0f}] + ALE regrid/remap [{REMAP_SCHEME}]",
            "btstep_nstep": int(bcs.nstep_last), "dtbt_s": float(bcs.dtbt),
            "not_yet_in_step": [],
            "lateral_parameterizations_in_cycle": ("thickness_diffuse [KHTH=600] + mixedlayer_restrat [FOX_KEMPER_ML_RESTRAT_COEF=5]" +
                                                   (" + tracer_hordiff with USE_NEUTRAL_DIFFUSION" if NEUTRAL else "")) if LATERAL else None,
            "vertvisc": dict(VERTVISC, ntrunc=int(M.CS.vertvisc_CSp.ntrunc)), "hor_visc": HOR_VISC, "set_visc": SET_VISC,
            "ALE": f"z* regrid with old_grid_weight={REGRID_OLD_WEIGHT} (REGRID_TIME_SCALE = 0, the default), remap of T, S + 2 tracers "
                   f"and of u, v [{REMAP_SCHEME}]",
            "advect_iterations_last_call": None if M.last_adv is None else int(M.last_adv.iterations),
            "hordiff_iterations_last_call": None if getattr(M, "last_hordiff", None) is None else int(M.last_hordiff.num_itts),
            "state_at_start": health0, "state_after_warmup": health_w, "state_after_run": health, "model_steps_taken": M.nstep,
            "KE_growth_per_step_in_timed_window": ke_growth, "long_run_health_record": "profiles/r04_health_om4.json",
            "spinup_steps_before_warmup": SPINUP if a.spinup is None else a.spinup, "state_synthesis": STATE,
            "parallelism": "1 tile" if world == 1 else f"layout 1x{world}: {world} latitude bands, one per GPU, group passes "
                                                               + ("over RCCL send / recv inside the library (its communication stream)"
                                                                  if exchange == "rccl" else f"through torch.distributed callbacks ({backend})"),
            "exchange": exch, "exchange_preflight": preflight,
            # (row-split groups, continuity calls in two phases, passes completed before anything else ran, 0) of this rank, all steps
            "overlap_stats": list(M.dg.overlap_stats()),
        },
    }
    # what the device delivers to plain streaming kernels on THIS box (a = b, a = b + s*c over 4 GiB arrays), next to the nominal
    # peak the fractions are quoted against (report both)
    measured_bw = None
    if rank == 0 and not a.no_roofline:
        import ctypes as C
        from mom6_amd._lib import check, lib
        cp, tr = C.c_double(0.0), C.c_double(0.0)
        L = lib()
        L.mom6hip_stream_bandwidth.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        check(L.mom6hip_stream_bandwidth(M.dg.handle, 4 << 30, 10, C.byref(cp), C.byref(tr)), "mom6hip_stream_bandwidth")
        measured_bw = {"copy_GBs": cp.value, "triad_GBs": tr.value, "bytes_per_array": 4 << 30, "launches": 10}
    bt_floor = None
    if rank == 0 and world == 1:
        nodes = M.dg.bt_graph_nodes()
        if nodes > 0:
            pts = (NI + 2 * 4) * (NJ + 2 * 4)
            bt_floor = {"graph_nodes": nodes, "points": pts, "us_per_node": M.dg.graph_node_floor(nodes, pts, 20)}
    M.dg.close()
    del M
    torch.cuda.empty_cache()

    # roofline of the dominant KERNEL of the step: cont_flux_coop_kernel<1,10> (meridional_mass_flux: edge values,
    # flux_adjust, BT_cont fits; mom6_amd/csrc/continuity.hip; profiles/r01_f_bench_om4_kernel_stats.csv).  Duration: HIP
    # events around its launches inside the timed region.  Algorithmic bytes per cell and launch:
    # read v, visc_rem_v, h (24 B), write vh (8 B) [+ v_cor (8 B) with vhbt] [+ h_v (8 B) with BT_cont] -> 40 / 48 / 40 B
    # for the three calls of a step.
    if rank == 0:
        ms_y, n_y = ktime[1]
        ms_x, n_x = ktime[0]
        if n_y > 0:
            alg = (40.0 + 48.0 + 40.0) / 3.0 * cells / world
            avg_ms = ms_y / n_y
            # HBM bytes of the kernel from the committed counter passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, tools/
            # profile_round.sh): valid only for the kernel source they were taken with -- the profile carries the SHA-256 of
            # continuity.hip, and a different source (or workload) makes `traffic` null instead of stale
            traffic, traffic_from, valu = None, None, None
            try:
                pmc, sha = _pmc_kernel("cont_flux_coop_kernel<1", "continuity.hip")
                if a.workload == "om4_025" and world == 1 and pmc is not None:
                    traffic = pmc["fetch_bytes"] + pmc["write_bytes"]
                    traffic_from = f"profiles/{PMC_PROFILE} (continuity.hip sha256 {sha[:12]})"
                    if "SQ_INSTS_VALU" in pmc:      # what actually bounds this kernel: fp64 VALU issue (4 cycles per wave instruction)
                        issue_ms = pmc["SQ_INSTS_VALU"] * 4.0 / (1024 * 2.4e9) * 1e3
                        valu = {"valu_wave_instructions_per_launch": pmc["SQ_INSTS_VALU"], "issue_ms_at_full_rate": issue_ms,
                                "issue_frac": issue_ms / avg_ms, "instructions_per_cell": pmc["SQ_INSTS_VALU"] * 64.0 / (cells / world)}
            except Exception:
                pass
            out["roofline"] = {
                "kernel": "cont_flux_coop_kernel<1,10>", "bound": "hbm", "achieved": alg / (avg_ms * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_from": traffic_from, "fp64_valu": valu,
                "algorithmic_bytes_per_launch": alg, "avg_launch_ms": avg_ms, "launches_timed": int(n_y),
                # the zonal flux kernel is the round-5 one-cell-per-lane layout (3 waves/SIMD, neighbours by DPP); the meridional one
                # spills in that layout and stays on the round-3 kernel (profiles/r05_experiments.txt section 1)
                "also": {"cont_flux_coop3_kernel<0,5,8,4>": {"avg_launch_ms": ms_x / max(n_x, 1), "launches_timed": int(n_x)}},
                "measured_streaming_bandwidth": measured_bw,
                "frac_of_measured_triad": None if not measured_bw else alg / (avg_ms * 1e-3) / 1e9 / measured_bw["triad_GBs"],
            }

    if rank == 0:
        # microseconds per barotropic time step against the launch floor.  HIP events around the replay of the
        # subcycle's hipGraph (barotropic.hip), inside the timed region; 4 kernels + the x-halo wraps per barotropic step.
        ms_bt, n_bt = ktime[2]
        if n_bt > 0 and int(bcs.nstep_last) > 0 and bt_floor is not None:
            nstep, nodes, node_us = int(bcs.nstep_last), bt_floor["graph_nodes"], bt_floor["us_per_node"]
            per = ms_bt / n_bt / nstep * 1e3
            out["barotropic_subcycle"] = {
                "us_per_barotropic_step": per, "nstep": nstep, "graph_replays_timed": int(n_bt),
                "ms_per_btstep_call_in_subcycle": ms_bt / n_bt, "graph_nodes": nodes, "kernels_per_barotropic_step": nodes / nstep,
                # the launch floor measured here: the same number of dependent kernel nodes in a replayed hipGraph, each one
                # read-modify-write pass over a 2-D array of the tile's size (mom6hip_graph_node_floor)
                "launch_floor_us_per_node": node_us, "launch_floor_us_per_barotropic_step": node_us * nodes / nstep,
                "frac_of_launch_floor": node_us * nodes / nstep / per, "points_2d_per_tile": bt_floor["points"],
            }
        ms_pgf, n_pgf = ktime[3]
        if n_pgf > 0:
            # pgf_face_kernel against the FP64 vector roof: wave instructions from the committed counter pass (valid for the source
            # it was taken with) x 4 cycles each / (1024 SIMDs x 2.4 GHz), over the duration measured live
            fp64 = {"kernel": "pgf_face_kernel", "avg_launch_ms": ms_pgf / n_pgf, "launches_timed": int(n_pgf),
                    "peak_TFLOPs_fp64_vector": FP64_VECTOR_TFLOPS}
            try:
                pmc, sha = _pmc_kernel("pgf_face_kernel", "pressure_force.hip")
                if a.workload == "om4_025" and world == 1 and pmc is not None and "SQ_INSTS_VALU" in pmc:
                    issue_ms = pmc["SQ_INSTS_VALU"] * 4.0 / (1024 * 2.4e9) * 1e3
                    fp64.update({"valu_wave_instructions_per_launch": pmc["SQ_INSTS_VALU"], "issue_ms_at_full_rate": issue_ms,
                                 "frac_of_fp64_issue": issue_ms / (ms_pgf / n_pgf),
                                 "instructions_per_cell": pmc["SQ_INSTS_VALU"] * 64.0 / cells,
                                 "counters_from": f"profiles/{PMC_PROFILE} (pressure_force.hip sha256 {sha[:12]})"})
            except Exception:
                pass
            out["pressure_force_fp64"] = fp64

    if world == 1 and not a.no_roofline:
        # per-operator device time on a stationary synthetic state, HIP events on the stream the library launches on
        # (the null stream, which is also torch's current stream here)
        S = Components(grid, dom, device, a.scheme)
        comp = {}
        for n in range(spa):      # untimed pass: work-space allocations and first-launch costs stay out of the timings
            for name, f in S.parts(n):
                f()
        torch.cuda.synchronize()
        for n in range(spa):
            for name, f in S.parts(n):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); f(); e1.record(); e1.synchronize()
                comp.setdefault(name, []).append(e0.elapsed_time(e1))
        comp_ms = {k: sum(v) / len(v) for k, v in comp.items()}
        out["components_ms_per_call"] = comp_ms
        out["components_ms_per_step"] = {k: (sum(v) / spa) for k, v in comp.items()}
        # the dense (first-iteration) advect passes, measured per launch by the library's own HIP events
        S.dg.set_timing(True)
        from mom6_amd.tracer_advect import advect_tracer
        ad = S.adv
        tx = ty = 0.0
        for _ in range(3):
            advect_tracer(ad["h_end"], ad["uhtr"], ad["vhtr"], None, DT_THERM, S.dg, S.adv_cs, ad["tr"])
            t = S.dg.advect_timing(); tx += t.ms_x1 / 3; ty += t.ms_y1 / 3
        # TRACER_ADVECTION_SCHEME = PPM (the Colella-Woodward edge values, MOM_tracer_advect.F90:722-760) beside the configured
        # PPM:H3, on the same state (SURVEY.md 8d C2 asks for both)
        from mom6_amd.tracer_advect import tracer_advect_init
        cs_ppm = tracer_advect_init(DT, "PPM")
        px = py = pc = 0.0
        for q in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            advect_tracer(ad["h_end"], ad["uhtr"], ad["vhtr"], None, DT_THERM, S.dg, cs_ppm, ad["tr"])
            e1.record(); e1.synchronize()
            t = S.dg.advect_timing()
            if q:      # (the first call: work-space allocation)
                px += t.ms_x1 / 3; py += t.ms_y1 / 3; pc += e0.elapsed_time(e1) / 3
        S.dg.set_timing(False)
        cand = {k: (ALG_BYTES[k] * cells, ms) for k, ms in comp_ms.items() if k in ALG_BYTES}
        cand["adv_x_kernel<4,PPM:H3,first>"] = ((NTR + 2) * 16.0 * cells, tx)
        cand["adv_y_kernel<4,PPM:H3,first>"] = ((NTR + 2) * 16.0 * cells, ty)
        cand["adv_x_kernel<4,PPM,first>"] = ((NTR + 2) * 16.0 * cells, px)
        cand["adv_y_kernel<4,PPM,first>"] = ((NTR + 2) * 16.0 * cells, py)
        out["advect_tracer_PPM_ms_per_call"] = pc
        # the two lateral parameterisations of SURVEY.md 8f #4, beside the step (MOM.F90:1165, :1335), on the same state: thickness_diffuse
        # as .testing/tc4 sets it (KHTH alone), mixedlayer_restrat as OM4 sets it (the boundary-layer depth, both running means, the
        # frontal length scale).  Not part of `value`: the workload BASELINE.json names does not call them.
        try:
            from mom6_amd.mixedlayer_restrat import mixedlayer_restrat, mixedlayer_restrat_init
            from mom6_amd.thickness_diffuse import thickness_diffuse, thickness_diffuse_init
            d = S.dyn
            sh2 = tuple(d["h"].shape[1:])
            yy = torch.linspace(0.0, 1.0, sh2[0], device=d["h"].device, dtype=torch.float64)[:, None].expand(sh2).contiguous()
            td_cs = thickness_diffuse_init(S.dg, THICKNESSDIFFUSE=True, KHTH=600.0)
            mle_cs = mixedlayer_restrat_init(S.dg, FOX_KEMPER_ML_RESTRAT_COEF=1.0, MLE_FRONT_LENGTH=500.0, MLE_USE_PBL_MLD=True, MLE_MLD_DECAY_TIME=345600.0,
                                             MLE_MLD_DECAY_TIME2=5184000.0, FOX_KEMPER_ML_RESTRAT_COEF2=0.5,
                                             MLD_filtered=torch.zeros(sh2, device=d["h"].device, dtype=torch.float64),
                                             MLD_filtered_slow=torch.zeros(sh2, device=d["h"].device, dtype=torch.float64))
            ustar, h_MLD, Rd = 0.005 + 0.01 * yy, 20.0 + 80.0 * yy, (0.2 + 1.5 * yy).contiguous()
            lat = {"thickness_diffuse": 0.0, "mixedlayer_restrat": 0.0}
            for q in range(4):
                hh = d["h"].clone(); uq = torch.zeros_like(S.adv["uhtr"]); vq = torch.zeros_like(S.adv["vhtr"])
                for name, f in (("thickness_diffuse", lambda: thickness_diffuse(hh, uq, vq, (d["T"], d["S"], S.eos), DT_THERM, S.dg, None, None, None, td_cs)),
                                ("mixedlayer_restrat", lambda: mixedlayer_restrat(hh, uq, vq, (d["T"], d["S"], S.eos), dict(ustar=ustar), DT_THERM, None, h_MLD,
                                                                                  None, dict(Rd_dx_h=Rd), S.dg, mle_cs))):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); f(); e1.record(); e1.synchronize()
                    if q:
                        lat[name] += e0.elapsed_time(e1) / 3
                del hh, uq, vq
            # tracer_hordiff with USE_NEUTRAL_DIFFUSION (the continuous branch, NTR tracers of which two are T and S, KHTR = 50)
            from mom6_amd.tracer_hor_diff import tracer_hor_diff_init as _hd_init, tracer_hordiff as _hd
            nd_cs = _hd_init(KHTR=50.0, USE_NEUTRAL_DIFFUSION=True)
            trs = [d["T"].clone(), d["S"].clone()] + [torch.rand_like(d["T"]) for _ in range(max(NTR - 2, 0))]
            tvn = dict(T=trs[0], S=trs[1], eqn_of_state=S.eos)
            tn = 0.0
            for q in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); _hd(d["h"], DT_THERM, None, None, None, S.dg, nd_cs, trs, tv=tvn); e1.record(); e1.synchronize()
                if q:
                    tn += e0.elapsed_time(e1) / 2
            lat[f"tracer_hordiff[neutral,{len(trs)} tracers]"] = tn
            del trs
            out["lateral_parameterizations_ms_per_call"] = lat
        except Exception as exc:      # (reported, never fatal to the bench line)
            out["lateral_parameterizations_ms_per_call"] = {"error": repr(exc)}
        out.setdefault("roofline", {})["operators"] = {
            k: {"GBs": bb / (m * 1e-3) / 1e9, "frac": bb / (m * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms": m} for k, (bb, m) in cand.items()}
        S.dg.close()

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(grid, a.scheme, cells, spa)

    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
