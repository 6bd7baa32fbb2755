"""BASELINE.json configs[3], Phillips_2layer: a 1/8-degree-like re-entrant channel, 480 x 320 x 2, two layers in thermal-wind
balance across a jet, LINEAR equation of state, automatic DTBT -> dozens of barotropic steps per baroclinic step (the
hipGraph / wide-halo stress case).  The set-up follows src/user/Phillips_initialization.F90:37-128, :132-208
(tests/exact_synth.py: make_phillips).

CPU: the oracle steps a small copy of the channel 200 times from the balanced state -- kinetic energy and the free surface
stay bounded (the check ADVICE asked for: a transcription error in btstep's weights, the pbce / eta_PF coupling or the
h_av seams shared by oracle and library would show as growth here).
GPU: the library against the oracle at the full size, bit for bit, one tile and two tiles."""
import numpy as np
import pytest

import exact_synth as xs
from helpers import bits_equal
from mom6_amd import _abi
from oracle import orc

H, U, V = _abi.POS_H, _abi.POS_U, _abi.POS_V


def kinetic_energy(g, S):
    sj, si = g.csl(H)
    A = g.areaT[sj, si][None]
    hh = S.h[:, sj, si]
    ku = 0.5 * (S.u[:, sj, si.start:si.stop] ** 2 + S.u[:, sj, si.start + 1:si.stop + 1] ** 2)
    kv = 0.5 * (S.v[:, sj.start:sj.stop, si] ** 2 + S.v[:, sj.start + 1:sj.stop + 1, si] ** 2)
    return float((0.5 * hh * A * (ku + kv)).sum())


def test_oracle_balanced_channel_stays_bounded():
    g, st, g_prime = xs.make_phillips(96, 64)
    dt = 1200.0
    S = orc.DynState(g, st["u"], st["v"], st["h"], st["T"], st["S"], dt, eos_form="LINEAR")
    tz = (g.zeros2(U), g.zeros2(V))
    sj, si = g.csl(H)
    vol0 = float((S.h[:, sj, si] * g.areaT[sj, si][None]).sum())
    ke, eta = [], []
    for n in range(200):
        S.step(tz[0], tz[1], calc_dtbt=(n == 0))
        ke.append(kinetic_energy(g, S)); eta.append(float(np.abs(S.arrs["eta"][sj, si]).max()))
    assert S.bcs.nstep_last >= 20                       # a real subcycle
    assert np.all(np.isfinite(S.u)) and S.h.min() > 0
    # no sustained growth: the last 50 steps hold no more energy than the first 50 (inertial oscillations of the imperfect
    # discrete balance exchange kinetic and potential energy, so single steps go up and down)
    assert max(ke[150:]) <= 1.05 * max(ke[:50]), (max(ke[:50]), max(ke[150:]))
    assert max(eta[150:]) <= 1.1 * max(eta[:50]) and max(eta) < 1.0
    vol = float((S.h[:, sj, si] * g.areaT[sj, si][None]).sum())
    assert abs(vol - vol0) <= 1e-12 * vol0
    # the jet is still there (the run is not just damping everything away)
    assert np.abs(S.u[0]).max() > 0.2 * np.abs(st["u"][0]).max()


def layer_densities(g, g_prime, nk=2):
    """GV%Rlay and GV%g_prime of the layered Phillips_2layer run (set_coord_from_gprime, MOM_coord_initialization.F90:167-170: the
    densities follow from the reduced gravities): the densities the LINEAR equation of state gives the two layers' temperatures"""
    gp = np.zeros(nk + 1); gp[0] = g.g_Earth; gp[nk // 2] = g_prime
    Rlay = np.zeros(nk); Rlay[0] = 1000.0 - 0.2 * 20.0 + 0.8 * 35.0
    for k in range(1, nk):
        Rlay[k] = Rlay[k - 1] + gp[k] * (g.Rho0 / g.g_Earth)
    return Rlay, gp


def test_oracle_phillips_with_layer_densities_and_no_equation_of_state():
    """Phillips_2layer as the reference runs it: no equation of state (tv%eqn_of_state not associated), PressureForce_FV_Bouss on
    GV%Rlay and GV%g_prime (MOM_PressureForce_FV.F90:620-640, :868-905).  Bounded like the LINEAR-EOS emulation"""
    g, st, g_prime = xs.make_phillips(96, 64)
    dt = 1200.0
    Rlay, gp = layer_densities(g, g_prime)
    S = orc.DynState(g, st["u"], st["v"], st["h"], st["T"], st["S"], dt, eos_form=None, pressureforce=dict(use_ALE=False, Rlay=Rlay, g_prime=gp))
    L = orc.DynState(g, st["u"], st["v"], st["h"], st["T"], st["S"], dt, eos_form="LINEAR")
    tz = (g.zeros2(U), g.zeros2(V))
    sj, si = g.csl(H)
    ke = []
    for n in range(120):
        S.step(tz[0], tz[1], calc_dtbt=(n == 0))
        ke.append(kinetic_energy(g, S))
        if n == 0:
            # the same interface slopes and reduced gravity: the first pressure force is that of the LINEAR-EOS emulation to roundoff
            # (afterwards the two differ: the layered free surface feels g, the emulation g*Rlay(1)/Rho0, MOM_PressureForce_FV.F90:868-905)
            L.step(tz[0], tz[1], calc_dtbt=True)
            assert np.abs(S.arrs["PFu"] - L.arrs["PFu"]).max() < 1e-8 * np.abs(L.arrs["PFv"]).max() and not bits_equal(S.arrs["pbce"], L.arrs["pbce"])
            assert np.all(S.arrs["pbce"][0][sj, si] == g.g_Earth)
    assert S.bcs.nstep_last >= 20 and np.all(np.isfinite(S.u)) and S.h.min() > 0
    assert max(ke[90:]) <= 1.05 * max(ke[:30])
    assert np.abs(S.u[0]).max() > 0.2 * np.abs(st["u"][0]).max()


@pytest.mark.gpu
def test_phillips_with_layer_densities_matches_oracle_bitwise():
    """the full-size channel without an equation of state: library == oracle, bit for bit"""
    import torch
    from mom6_amd.dynamics_split_rk2 import initialize_dyn_split_RK2, step_MOM_dyn_split_RK2
    from mom6_amd.tracer_advect import DeviceGrid
    g, d, g_prime = xs.make_phillips()
    dt = 1800.0
    Rlay, gp = layer_densities(g, g_prime)
    ref = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], dt, eos_form=None, pressureforce=dict(use_ALE=False, Rlay=Rlay, g_prime=gp))
    dg = DeviceGrid(g)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    u, v, h = (T(d[k]) for k in ("u", "v", "h"))
    Z = lambda pos, k3=True: torch.zeros(g.shape3(pos) if k3 else g.shape2(pos), dtype=torch.float64, device="cuda")
    uh, vh, uhtr, vhtr, eta_av = Z(U), Z(V), Z(U), Z(V), Z(H, False)
    CS = initialize_dyn_split_RK2(u, v, h, uh, vh, dt, dg, EQN_OF_STATE=None, coriolis=dict(bound_coriolis=True),
                                  pressure_force=dict(use_ALE=False, Rlay=Rlay, g_prime=gp))
    tz = (g.zeros2(U), g.zeros2(V)); tx, ty = Z(U, False), Z(V, False)
    for n in range(3):
        ref.step(tz[0], tz[1], calc_dtbt=(n == 0))
        step_MOM_dyn_split_RK2(u, v, h, None, None, None, dt, (tx, ty), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS, calc_dtbt=(n == 0))
        dg.sync()
        assert CS.barotropic_CSp.st.nstep_last == ref.bcs.nstep_last >= 30
        for name, a, b in (("u", u, ref.u), ("v", v, ref.v), ("h", h, ref.h), ("uh", uh, ref.uh), ("eta_av", eta_av, ref.eta_av),
                           ("PFu", CS.PFu, ref.arrs["PFu"]), ("pbce", CS.pbce, ref.arrs["pbce"])):
            an = a.cpu().numpy()
            assert bits_equal(an, b), (n, name, float(np.abs(an - b).max()))
    dg.close()


@pytest.mark.gpu
def test_phillips_step_matches_oracle_bitwise():
    import torch
    from mom6_amd.dynamics_split_rk2 import initialize_dyn_split_RK2, step_MOM_dyn_split_RK2
    from mom6_amd.tracer_advect import DeviceGrid
    g, d, _ = xs.make_phillips()
    assert (g.ni, g.nj, g.nk) == (480, 320, 2)
    dt = 1800.0
    ref = orc.DynState(g, d["u"], d["v"], d["h"], d["T"], d["S"], dt, eos_form="LINEAR")
    dg = DeviceGrid(g)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    u, v, h, Tt, Ss = (T(d[k]) for k in ("u", "v", "h", "T", "S"))
    Z = lambda pos, k3=True: torch.zeros(g.shape3(pos) if k3 else g.shape2(pos), dtype=torch.float64, device="cuda")
    uh, vh, uhtr, vhtr, eta_av = Z(U), Z(V), Z(U), Z(V), Z(H, False)
    CS = initialize_dyn_split_RK2(u, v, h, uh, vh, dt, dg, EQN_OF_STATE="LINEAR", coriolis=dict(bound_coriolis=True))
    tz = (g.zeros2(U), g.zeros2(V)); tx, ty = Z(U, False), Z(V, False)
    for n in range(3):
        ref.step(tz[0], tz[1], calc_dtbt=(n == 0))
        step_MOM_dyn_split_RK2(u, v, h, (Tt, Ss), None, None, dt, (tx, ty), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS,
                               calc_dtbt=(n == 0))
        dg.sync()
        st = CS.barotropic_CSp.st
        assert st.dtbt == ref.bcs.dtbt and st.nstep_last == ref.bcs.nstep_last and st.nstep_last >= 30
        for name, a, b in (("u", u, ref.u), ("v", v, ref.v), ("h", h, ref.h), ("uh", uh, ref.uh), ("vh", vh, ref.vh),
                           ("uhtr", uhtr, ref.uhtr), ("eta_av", eta_av, ref.eta_av), ("eta", CS.eta, ref.arrs["eta"]),
                           ("u_av", CS.u_av, ref.arrs["u_av"]), ("CAu_pred", CS.CAu_pred, ref.arrs["CAu_pred"])):
            an = a.cpu().numpy()
            assert bits_equal(an, b), (n, name, float(np.abs(an - b).max()))
    cap, lau = dg.bt_graph_stats()      # every btstep replays the hipGraph of its subcycle
    assert lau == 6 and 1 <= cap <= 6, (cap, lau)
    dg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [(1, 2), (2, 1)])
def test_phillips_layout_independence(tmp_path, layout):
    import torch.multiprocessing as mp
    from mp_workers import phillips_layout_worker
    from test_domains import free_port
    mp.spawn(phillips_layout_worker, args=(2, free_port(), layout, str(tmp_path)), nprocs=2, join=True)
    glob = np.load(tmp_path / "ph_global.npz")
    hh = 4
    for r in range(2):
        t = np.load(tmp_path / f"ph_tile{r}.npz")
        i0, j0, ni, nj = t["ij"]
        assert float(t["dtbt"]) == float(glob["dtbt"]) and int(t["nstep"]) == int(glob["nstep"])
        for n, (sx, sy) in dict(h=(0, 0), eta=(0, 0), u=(1, 0), v=(0, 1), uhtr=(1, 0)).items():
            a = t[n][..., hh:hh + nj + sy, hh:hh + ni + sx]
            b = glob[n][..., hh + j0:hh + j0 + nj + sy, hh + i0:hh + i0 + ni + sx]
            assert np.array_equal(a.view(np.uint64), np.ascontiguousarray(b).view(np.uint64)), (layout, r, n)
