"""write_energy's global integrals (src/diagnostics/MOM_sum_output.F90:490-760; mom6hip_write_energy_sums): the oracle's totals
against exact rational arithmetic on the integrands formed with numpy, and the library against the oracle, bit for bit."""
from fractions import Fraction

import numpy as np
import pytest

from mom6_amd import _abi, synth
from oracle import orc

H, U, V = _abi.POS_H, _abi.POS_U, _abi.POS_V
C_P, RHO = 3991.86795711963, 1035.0


def case(ni=30, nj=22, nk=4, seed=5, **kw):
    g = synth.make_grid(ni, nj, nk, seed=seed, land_frac=0.25, **kw)
    d = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=seed + 1, umax=0.4, eta_amp=0.3).items()}
    return g, d


def efp_value(ints):
    return sum(Fraction(int(x)) * Fraction(2) ** (46 * (2 - i)) for i, x in enumerate(ints))


def test_oracle_sums_are_the_exact_sums_of_the_integrands():
    g, d = case()
    dt = 900.0
    r = orc.write_energy_sums(g, d["u"], d["v"], d["h"], d["T"], d["S"], dt)
    sj, si = g.csl(H)
    hl = g.halo
    areaTm = (g.mask2dT * g.areaT)[sj, si]
    h = d["h"][:, sj, si]
    mass = h * (RHO * areaTm)                                            # :506
    uW = d["u"][:, sj, hl:hl + g.ni]; uE = d["u"][:, sj, hl + 1:hl + g.ni + 1]
    vS = d["v"][:, hl:hl + g.nj, si]; vN = d["v"][:, hl + 1:hl + g.nj + 1, si]
    ke = (0.25 * RHO * (areaTm * h)) * ((uW * uW + uE * uE) + (vS * vS + vN * vN))      # :685
    exact = lambda a: sum(Fraction(float(x)) for x in a.ravel())
    assert efp_value(r["mass_EFP"]) == exact(mass)
    for k in range(g.nk):
        assert abs(Fraction(r["mass_lay"][k]) - exact(mass[k])) <= abs(exact(mass[k])) * Fraction(1, 2 ** 51)
        assert abs(Fraction(r["KE_lay"][k]) - exact(ke[k])) <= abs(exact(ke[k])) * Fraction(1, 2 ** 51)
    tot = 0.0
    for x in r["KE_lay"]:
        tot = tot + x
    assert r["KE_tot"] == tot and r["toten"] == r["KE_tot"] and r["PE_tot"] == 0.0
    salt = np.zeros_like(areaTm); heat = np.zeros_like(areaTm)
    for k in range(g.nk):                                                 # :695-700, the k order of the reference
        salt = salt + 1.0 * d["S"][k][sj, si] * (h[k] * (RHO * areaTm))
        heat = heat + (1.0 * C_P * d["T"][k][sj, si]) * (h[k] * (RHO * areaTm))
    assert efp_value(r["salt_EFP"]) == exact(salt) and efp_value(r["heat_EFP"]) == exact(heat)
    assert r["npoints"] == g.ni * g.nj * g.nk
    # the CFL numbers :718-744
    IaT = g.IareaT
    best = [0.0, 0.0]
    for k in range(g.nk):
        for j in range(hl, hl + g.nj):
            for I in range(hl, hl + g.ni + 1):                            # u faces isc-1 .. iec (array column I of the u array)
                uu = d["u"][k, j, I]
                ia = IaT[j, I] if uu < 0.0 else IaT[j, I - 1]
                best[0] = max(best[0], abs(uu * dt) * (g.dy_Cu[j, I] * ia)); best[1] = max(best[1], abs(uu * dt) * g.IdxCu[j, I])
        for J in range(hl, hl + g.nj + 1):
            for i in range(hl, hl + g.ni):
                vv = d["v"][k, J, i]
                ia = IaT[J, i] if vv < 0.0 else IaT[J - 1, i]
                best[0] = max(best[0], abs(vv * dt) * (g.dx_Cv[J, i] * ia)); best[1] = max(best[1], abs(vv * dt) * g.IdyCv[J, i])
    assert r["max_CFL"] == best


@pytest.mark.gpu
@pytest.mark.parametrize("space", ["device", "host"])
def test_write_energy_matches_the_oracle(space):
    import torch
    from mom6_amd.sum_output import write_energy
    from mom6_amd.tracer_advect import DeviceGrid
    for ni, nj, nk, kw in ((70, 45, 5, {}), (44, 40, 2, dict(reentrant_x=True, reentrant_y=True)), (300, 130, 3, dict(reentrant_x=True))):
        g, d = case(ni, nj, nk, seed=ni, **kw)
        dg = DeviceGrid(g)
        put = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if space == "device" else (lambda a: np.ascontiguousarray(a))
        want = orc.write_energy_sums(g, d["u"], d["v"], d["h"], d["T"], d["S"], 900.0)
        got = write_energy(put(d["u"]), put(d["v"]), put(d["h"]), (put(d["T"]), put(d["S"])), dg, 900.0)
        for key in want:
            a, b = want[key], got[key]
            if key.endswith("EFP") or key == "npoints":
                assert a == b, (ni, key, a, b)
            else:
                assert (np.array(a, dtype=np.float64).view(np.uint64) == np.array(b, dtype=np.float64).view(np.uint64)).all(), (ni, key, a, b)
        no_t = write_energy(put(d["u"]), put(d["v"]), put(d["h"]), None, dg, 900.0)
        assert no_t["Salt"] == 0.0 and no_t["KE_tot"] == want["KE_tot"] and no_t["mass_EFP"] == want["mass_EFP"]
        dg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [(1, 2), (2, 1)])
def test_write_energy_layout_independence(tmp_path, layout):
    """two tiles (gloo ranks sharing the card): the integers of every total, the by-layer sums and the CFL numbers of the one-tile
    run (the reals of salt and heat are EFP_to_real of integers that are only carried after the sum over PEs, as in the
    reference: equal as numbers)"""
    import torch.multiprocessing as mp
    from mp_workers import write_energy_layout_worker
    from test_domains import free_port
    mp.spawn(write_energy_layout_worker, args=(2, free_port(), layout, str(tmp_path)), nprocs=2, join=True)
    glob = np.load(tmp_path / "global.npz")
    for r in range(2):
        t = np.load(tmp_path / f"tile{r}.npz")
        for name in ("mass_EFP", "mass_lay", "KE_lay", "totals", "max_CFL"):
            assert np.array_equal(t[name].view(np.uint64), glob[name].view(np.uint64)), (layout, r, name)
        for name in ("salt_EFP", "heat_EFP"):
            assert efp_value(t[name]) == efp_value(glob[name]), (layout, r, name)


# ---- CALCULATE_APE: the depth list (:1109-1232), the reference heights (:610-630), the integrand (:633-645) ----------------
def g_prime_of(nk):
    gp = np.full(nk + 1, 9.8 * 2.0e-3); gp[0] = 9.8
    return gp


def test_depth_list_is_the_sorted_hypsometry():
    """create_depth_list against numpy: the depths in descending order with (near-)duplicates culled, the area at each depth the
    sum of the areas of all cells at least that deep, the volume below it the integral of that area; two closing entries"""
    rng = np.random.default_rng(3)
    D = np.round(rng.uniform(0.0, 4000.0, 500), 0); D[rng.random(500) < 0.2] = 0.0       # ties and land
    A = rng.uniform(1.0e8, 2.0e8, 500); A[D == 0.0] = 0.0
    depth, area, vol = orc.depth_list(D, A)
    uniq = np.unique(D)[::-1]                                                          # descending, unique
    n = len(uniq)
    assert len(depth) == n + 1 or len(depth) == n + 2
    assert np.array_equal(depth[:n], uniq) if len(depth) == n + 1 else np.array_equal(depth[:n], uniq)
    for k in range(n):
        deeper = D >= depth[k]
        assert abs(area[k] - A[deeper].sum()) <= 1e-9 * max(1.0, A.sum())
        assert abs(vol[k] - (A[deeper] * (D[deeper] - depth[k])).sum()) <= 1e-9 * max(1.0, (A * D).sum())
    assert vol[-1] == vol[-2] * 1000.0 and depth[-1] == depth[-2] and area[-1] == area[-2]


def test_ape_of_a_state_at_rest_is_the_reference_level_energy():
    """layers of uniform thickness over a flat bottom: every interface already sits at its reference height, so hint = Z_0APE + eta
    is the same everywhere and PE_pt = 1/2 area rho g' (hint^2 - hbot^2) with hbot = 0 below the bottom ... and a bump of one
    interface raises the total"""
    g = synth.make_grid(24, 16, 3, seed=2, land_frac=0.0)
    gflat = g
    gflat.metrics["bathyT"][...] = 1000.0
    h = np.zeros(g.shape3(H)); h[0] = 200.0; h[1] = 300.0; h[2] = 500.0
    u, v = np.zeros(g.shape3(U)), np.zeros(g.shape3(V))
    r = orc.write_energy_sums(gflat, u, v, h, h * 0 + 10.0, h * 0 + 35.0, 900.0)
    a = orc.write_energy_ape(gflat, h, r["mass_lay"], g_prime_of(3))
    # the reference DEPTHS of the interfaces are the interfaces' own depths (hint = Z_0APE + eta vanishes at rest)
    assert np.allclose(a["Z_0APE"][:3], [0.0, 200.0, 500.0], atol=1e-6) and abs(a["PE_tot"]) <= 1e-6 * 9.8 * 1035.0 * 1.0e12
    hb = h.copy(); hb[0, 8:12, 10:14] += 5.0; hb[1, 8:12, 10:14] -= 5.0      # an interface displaced, the column total kept
    rb = orc.write_energy_sums(gflat, u, v, hb, h * 0 + 10.0, h * 0 + 35.0, 900.0)
    b = orc.write_energy_ape(gflat, hb, rb["mass_lay"], g_prime_of(3))
    assert b["PE_tot"] > a["PE_tot"]


@pytest.mark.gpu
def test_write_energy_ape_matches_the_oracle():
    import torch
    from mom6_amd.sum_output import depth_list_setup, write_energy
    from mom6_amd.tracer_advect import DeviceGrid
    for ni, nj, nk, kw in ((70, 45, 5, {}), (44, 40, 2, dict(reentrant_x=True, reentrant_y=True)), (200, 90, 8, dict(reentrant_x=True))):
        g, d = case(ni, nj, nk, seed=ni, **kw)
        dg = DeviceGrid(g)
        gp = g_prime_of(nk)
        want = orc.write_energy_sums(g, d["u"], d["v"], d["h"], d["T"], d["S"], 900.0)
        wa = orc.write_energy_ape(g, d["h"], want["mass_lay"], gp)
        n = depth_list_setup(dg)
        assert n >= 3
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        for rep in range(2):      # the second call starts its searches from the remembered list positions (CS%lH)
            got = write_energy(T(d["u"]), T(d["v"]), T(d["h"]), (T(d["T"]), T(d["S"])), dg, 900.0, g_prime=gp)
            for key in ("PE", "Z_0APE"):
                assert (np.array(got[key]).view(np.uint64) == np.array(wa[key]).view(np.uint64)).all(), (ni, key, rep)
            assert got["PE_tot"] == wa["PE_tot"] and got["toten"] == want["KE_tot"] + wa["PE_tot"]
        host = write_energy(d["u"], d["v"], d["h"], (d["T"], d["S"]), dg, 900.0, g_prime=gp)
        assert host["PE"] == got["PE"]
        dg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [(1, 2), (2, 1)])
def test_write_energy_ape_layout_independence(tmp_path, layout):
    import torch.multiprocessing as mp
    from mp_workers import write_energy_layout_worker
    from test_domains import free_port
    mp.spawn(write_energy_layout_worker, args=(2, free_port(), layout, str(tmp_path), True), nprocs=2, join=True)
    glob = np.load(tmp_path / "global.npz")
    for r in range(2):
        t = np.load(tmp_path / f"tile{r}.npz")
        for name in ("PE", "Z_0APE", "depth_list"):
            assert np.array_equal(t[name].view(np.uint64), glob[name].view(np.uint64)), (layout, r, name)
