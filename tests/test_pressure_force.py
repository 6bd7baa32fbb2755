"""EOS + PressureForce_FV_Bouss: the oracle (oracle/pressure_force.c) against the reference's EOS check values
(tests/golden/eos_check_values.json, from EOS_unit_tests) and hydrostatic-consistency properties; GPU parity of
libmom6hip against the oracle."""
import json
import os

import numpy as np
import pytest

from mom6_amd import _abi, synth
from helpers import bits_equal, interior

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "eos_check_values.json")))
EPS = np.finfo(np.float64).eps


@pytest.mark.parametrize("c", GOLD["cases"], ids=lambda c: c["form"] + c["_line"][:4])
def test_eos_check_values(oracle, c):
    E = oracle.eos(c["form"], c.get("Rho_T0_S0", 1000.0), c.get("dRho_dT", -0.2), c.get("dRho_dS", 0.8))
    rho_ref = GOLD["rho_ref"]
    rho = oracle.eos_density(E, c["T"], c["S"], c["p"], rho_ref=rho_ref)
    assert abs(c["rho_check"] - (rho_ref + rho)) < GOLD["rel_tol_eps"] * EPS * (rho_ref + rho)
    # the anomaly form and the in-situ form agree (test_EOS_consistency's own cross-check)
    rho2 = oracle.eos_density(E, c["T"], c["S"], c["p"])
    assert abs((rho_ref + rho) - rho2) < 1e-9


@pytest.mark.parametrize("form", ["WRIGHT", "UNESCO", "WRIGHT_FULL", "WRIGHT_REDUCED"])
def test_eos_derivs_match_finite_differences(oracle, form):
    E = oracle.eos(form)
    T, S, p = 10.0, 34.0, 2.0e7
    dT, dS = oracle.eos_density_derivs(E, T, S, p)
    fdT = (oracle.eos_density(E, T + 1e-4, S, p) - oracle.eos_density(E, T - 1e-4, S, p)) / 2e-4
    fdS = (oracle.eos_density(E, T, S + 1e-4, p) - oracle.eos_density(E, T, S - 1e-4, p)) / 2e-4
    assert abs(dT - fdT) < 1e-6 * abs(dT) and abs(dS - fdS) < 1e-6 * abs(dS)


def test_unesco_matches_reference_build(oracle):
    """The restated UNESCO equation of state against the reference's own MOM_EOS_UNESCO.F90 (compiled unmodified into
    oracle/_ref): density, density anomaly and both derivatives, bit for bit, over the fit range and beyond it (S < 0 is
    clamped, :106)."""
    R = oracle.ref_lib()
    if R is None or not hasattr(R, "ref_unesco"):
        pytest.skip("oracle/_ref not built (needs /root/reference and amdflang)")
    rng = np.random.default_rng(5)
    n = 20000
    T = rng.uniform(-3.0, 41.0, n); S = rng.uniform(-1.0, 43.0, n); p = rng.uniform(0.0, 1.1e8, n)
    T[:4] = [25.0, 0.0, -2.0, 40.0]; S[:4] = [35.0, 0.0, -0.5, 42.0]; p[:4] = [1.0e7, 0.0, 0.0, 1.0e8]
    E = oracle.eos("UNESCO")
    P = lambda a: a.ctypes.data_as(oracle._dp)
    for use_ref, rho_ref in ((0, 0.0), (1, 1000.0), (1, 1035.0)):
        rho = np.empty(n); dT = np.empty(n); dS = np.empty(n)
        R.ref_unesco(n, P(T), P(S), P(p), rho_ref, use_ref, P(rho), P(dT), P(dS))
        mine = np.array([oracle.eos_density(E, T[m], S[m], p[m], rho_ref if use_ref else None) for m in range(n)])
        assert bits_equal(mine, rho), (use_ref, rho_ref, np.argwhere(mine != rho)[:3])
        d = np.array([oracle.eos_density_derivs(E, T[m], S[m], p[m]) for m in range(n)])
        assert bits_equal(d[:, 0].copy(), dT) and bits_equal(d[:, 1].copy(), dS)
    assert abs(rho[0] + 1035.0 - 1027.54345796120) < 1000 * EPS * 1027.5


def test_spec_vol_forms(oracle):
    """calculate_spec_vol with spv_ref: the reciprocal of the density of the same form (to roundoff), and for UNESCO the
    reference's own function bit for bit (oracle/_ref)"""
    rng = np.random.default_rng(6)
    n = 4000
    T = rng.uniform(-2.0, 32.0, n); S = rng.uniform(0.0, 40.0, n); p = rng.uniform(0.0, 6.0e7, n)
    spv_ref = 1.0 / 1035.0
    for form in ("WRIGHT", "UNESCO", "LINEAR", "WRIGHT_FULL", "WRIGHT_REDUCED"):
        E = oracle.eos(form)
        a = np.array([oracle.eos_spec_vol_anomaly(E, T[m], S[m], p[m], spv_ref) for m in range(n)])
        rho = np.array([oracle.eos_density(E, T[m], S[m], p[m]) for m in range(n)])
        assert np.max(np.abs((a + spv_ref) * rho - 1.0)) < 1e-13, form
    R = oracle.ref_lib()
    if R is not None and hasattr(R, "ref_unesco_spv"):
        P = lambda x: x.ctypes.data_as(oracle._dp)
        want = np.empty(n)
        R.ref_unesco_spv(n, P(T), P(S), P(p), spv_ref, P(want))
        E = oracle.eos("UNESCO")
        mine = np.array([oracle.eos_spec_vol_anomaly(E, T[m], S[m], p[m], spv_ref) for m in range(n)])
        assert bits_equal(mine, want)


def test_nonbouss_pressure_force_tracks_the_boussinesq_one(oracle):
    """PressureForce_FV_nonBouss on the same ocean expressed in mass per unit area: the two forms of the same physics differ by
    the Boussinesq error (a percent), not more; a resting stratified ocean feels no force in either"""
    g, st = pgf_case(40, 26, 8, seed=5)
    E = oracle.eos("WRIGHT")
    cs = oracle.pressureforce_cs(g)
    B = oracle.pressureforce(g, cs, E, st["h"], st["T"], st["S"])
    N = oracle.pressureforce_nonbouss(g, cs, E, st["h"] * g.Rho0, st["T"], st["S"], H_to_RZ=1.0)
    for q, pos, mk in ((0, _abi.POS_U, g.mask2dCu), (1, _abi.POS_V, g.mask2dCv)):
        m = interior(g, mk, pos) > 0
        a = interior(g, B[q], pos)[:, m]; b = interior(g, N[q], pos)[:, m]
        assert np.sqrt(((a - b) ** 2).mean()) < 0.03 * np.sqrt((a ** 2).mean())
    # eta of the non-Boussinesq form is the column mass per unit area
    assert np.allclose(interior(g, N[3]), interior(g, (st["h"] * g.Rho0).sum(0)), rtol=1e-13)
    assert np.all(interior(g, N[2]) > 0)
    # at rest
    g2 = synth.make_grid(20, 16, 5, seed=1, land_frac=0.0, max_depth=4000.0)
    g2.set_metric("bathyT", np.full_like(g2.bathyT, 4000.0))
    shp = g2.shape3(_abi.POS_H)
    h = np.empty(shp); T = np.empty(shp); S = np.empty(shp)
    for k, dz in enumerate([50.0, 150.0, 800.0, 1000.0, 2000.0]):
        h[k] = dz * 1035.0; T[k] = 20.0 - 4.0 * k; S[k] = 34.0 + 0.2 * k
    PFu, PFv, _, _ = oracle.pressureforce_nonbouss(g2, oracle.pressureforce_cs(g2), E, h, T, S)
    assert np.max(np.abs(interior(g2, PFu, _abi.POS_U))) < 1e-11 and np.max(np.abs(interior(g2, PFv, _abi.POS_V))) < 1e-11


def pgf_case(ni=30, nj=22, nk=6, seed=3, **kw):
    g = synth.make_grid(ni, nj, nk, seed=seed + 10, **kw)
    st = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=seed).items()}
    return g, st


def test_resting_stratified_ocean_has_no_pressure_force(oracle):
    """Flat interfaces and horizontally uniform T,S in every layer: PFu = PFv = 0 up to roundoff relative to
    g (the finite-volume form is exactly hydrostatically consistent, Adcroft et al. 2008)."""
    g = synth.make_grid(20, 16, 5, seed=1, land_frac=0.0, max_depth=4000.0)
    g.set_metric("bathyT", np.full_like(g.bathyT, 4000.0))
    shp = g.shape3(_abi.POS_H)
    h = np.empty(shp); T = np.empty(shp); S = np.empty(shp)
    dz = np.array([50.0, 150.0, 800.0, 1000.0, 2000.0])
    for k in range(5):
        h[k] = dz[k]; T[k] = 20.0 - 4.0 * k; S[k] = 34.0 + 0.2 * k
    E = oracle.eos("WRIGHT")
    PFu, PFv, pbce, eta = oracle.pressureforce(g, oracle.pressureforce_cs(g), E, h, T, S)
    assert np.max(np.abs(interior(g, PFu, _abi.POS_U))) < 1e-12
    assert np.max(np.abs(interior(g, PFv, _abi.POS_V))) < 1e-12
    assert np.allclose(interior(g, eta), 0.0, atol=1e-9)
    assert np.all(interior(g, pbce)[0] > 9.0) and np.all(np.diff(interior(g, pbce), axis=0) > 0)


def test_sea_surface_slope_gives_g_times_slope(oracle):
    g = synth.make_grid(20, 16, 3, seed=1, land_frac=0.0, reentrant_x=False)
    g.set_metric("bathyT", np.full_like(g.bathyT, 1000.0))
    shp = g.shape3(_abi.POS_H)
    x = np.arange(shp[2])[None, :] * np.ones((shp[1], 1))
    h = np.empty(shp); h[0] = 100.0 + 0.01 * x; h[1] = 300.0; h[2] = 600.0
    T = np.full(shp, 10.0); S = np.full(shp, 35.0)
    E = oracle.eos("LINEAR", 1035.0, 0.0, 0.0)
    PFu, PFv, _, eta = oracle.pressureforce(g, oracle.pressureforce_cs(g, Rho0=1035.0), E, h, T, S)
    expect = -g.g_Earth * 0.01 * interior(g, g.IdxCu, _abi.POS_U)
    got = interior(g, PFu, _abi.POS_U)[0]
    assert np.allclose(got[:, 1:-1], expect[:, 1:-1], rtol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["WRIGHT", "UNESCO", "LINEAR", "WRIGHT_FULL", "WRIGHT_REDUCED"])
@pytest.mark.parametrize("opts", [(True, False), (False, True)])
def test_gpu_parity(oracle, form, opts):
    import torch
    from mom6_amd.pressure_force import PressureForce, PressureForce_init, EOS_init
    from mom6_amd.tracer_advect import DeviceGrid
    extrap, massw = opts
    for (ni, nj, nk, topo) in [(70, 21, 5, (True, False)), (44, 40, 2, (True, True)), (10, 8, 8, (False, False)),
                               (130, 9, 3, (True, False))]:
        g, st = pgf_case(ni, nj, nk, seed=ni, reentrant_x=topo[0], reentrant_y=topo[1])
        E = oracle.eos(form, 1000.0, -0.2, 0.8)
        cs = oracle.pressureforce_cs(g, boundary_extrap=extrap, useMassWghtInterp=massw)
        rng = np.random.default_rng(ni)
        for p_atm in (None, np.ascontiguousarray(1.0e5 + 500.0 * rng.standard_normal(g.shape2(_abi.POS_H)))):
            ref = oracle.pressureforce(g, cs, E, st["h"], st["T"], st["S"], p_atm)
            dg = DeviceGrid(g)
            CS = PressureForce_init(g, boundary_extrap=extrap, useMassWghtInterp=massw)
            EOS = EOS_init(form, 1000.0, -0.2, 0.8)
            for resident in (False, True):
                X = (lambda a: None if a is None else torch.from_numpy(a.copy()).cuda()) if resident else \
                    (lambda a: None if a is None else a.copy())
                PFu, PFv = X(g.zeros3(_abi.POS_U)), X(g.zeros3(_abi.POS_V))
                pbce, eta = X(g.zeros3(_abi.POS_H)), X(g.zeros2(_abi.POS_H))
                PressureForce(X(st["h"]), (X(st["T"]), X(st["S"]), EOS), PFu, PFv, dg, CS, p_atm=X(p_atm), pbce=pbce, eta=eta)
                dg.sync()
                N = (lambda a: a.cpu().numpy()) if resident else (lambda a: a)
                for name, a, b in (("PFu", ref[0], PFu), ("PFv", ref[1], PFv), ("pbce", ref[2], pbce), ("eta", ref[3], eta)):
                    assert bits_equal(a, N(b)), (form, opts, (ni, nj, nk), p_atm is not None, resident, name,
                                                 np.argwhere(a != N(b))[:3])
            dg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("extrap", [False, True])
def test_gpu_ale_plm_edge_values(oracle, extrap):
    """ALE_PLM_edge_values / TS_PLM_edge_values (MOM_ALE.F90:1520, :1495) through the library == the oracle, host and device arrays"""
    import torch
    from mom6_amd.ale import ALE_PLM_edge_values, TS_PLM_edge_values
    from mom6_amd.tracer_advect import DeviceGrid
    for (ni, nj, nk) in [(70, 21, 5), (10, 8, 2), (33, 9, 12)]:
        g, st = pgf_case(ni, nj, nk, seed=3 * ni)
        want = oracle.ale_plm_edge_values(g, st["h"], st["T"], extrap)
        dg = DeviceGrid(g)
        for resident in (False, True):
            X = (lambda a: torch.from_numpy(a.copy()).cuda()) if resident else (lambda a: a.copy())
            N = (lambda a: a.cpu().numpy()) if resident else (lambda a: a)
            Qt, Qb = X(g.zeros3(_abi.POS_H)), X(g.zeros3(_abi.POS_H))
            ALE_PLM_edge_values(None, dg, X(st["h"]), X(st["T"]), extrap, Qt, Qb)
            dg.sync()
            sj, si = slice(g.halo - 1, g.halo + nj + 1), slice(g.halo - 1, g.halo + ni + 1)
            assert bits_equal(want[0][:, sj, si], N(Qt)[:, sj, si]) and bits_equal(want[1][:, sj, si], N(Qb)[:, sj, si]), (ni, nj, nk, resident)
        St, Sb, Tt, Tb = (torch.from_numpy(g.zeros3(_abi.POS_H)).cuda() for _ in range(4))
        TS_PLM_edge_values(None, St, Sb, Tt, Tb, dg, (torch.from_numpy(st["T"]).cuda(), torch.from_numpy(st["S"]).cuda()),
                           torch.from_numpy(st["h"]).cuda(), extrap)
        dg.sync()
        wS = oracle.ale_plm_edge_values(g, st["h"], st["S"], extrap)
        assert bits_equal(want[0][:, sj, si], Tt.cpu().numpy()[:, sj, si]) and bits_equal(wS[1][:, sj, si], Sb.cpu().numpy()[:, sj, si])
        dg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["WRIGHT", "UNESCO", "LINEAR", "WRIGHT_FULL", "WRIGHT_REDUCED"])
@pytest.mark.parametrize("opts", [(True, False), (False, True)])
def test_gpu_parity_nonbouss(oracle, form, opts):
    """PressureForce_FV_nonBouss: library == oracle, bit for bit"""
    import torch
    from mom6_amd.pressure_force import PressureForce, PressureForce_init, EOS_init
    from mom6_amd.tracer_advect import DeviceGrid
    extrap, massw = opts
    for (ni, nj, nk, topo) in [(70, 21, 5, (True, False)), (44, 40, 2, (True, True)), (10, 8, 8, (False, False)), (130, 9, 75, (True, False))]:
        g, st = pgf_case(ni, nj, nk, seed=ni, reentrant_x=topo[0], reentrant_y=topo[1])
        hm = np.ascontiguousarray(st["h"] * g.Rho0)
        E = oracle.eos(form, 1000.0, -0.2, 0.8)
        cs = oracle.pressureforce_cs(g, boundary_extrap=extrap, useMassWghtInterp=massw)
        rng = np.random.default_rng(ni)
        for p_atm in (None, np.ascontiguousarray(1.0e5 + 500.0 * rng.standard_normal(g.shape2(_abi.POS_H)))):
            ref = oracle.pressureforce_nonbouss(g, cs, E, hm, st["T"], st["S"], p_atm)
            dg = DeviceGrid(g)
            CS = PressureForce_init(g, boundary_extrap=extrap, useMassWghtInterp=massw)
            EOS = EOS_init(form, 1000.0, -0.2, 0.8)
            for resident in (False, True):
                X = (lambda a: None if a is None else torch.from_numpy(a.copy()).cuda()) if resident else \
                    (lambda a: None if a is None else a.copy())
                PFu, PFv = X(g.zeros3(_abi.POS_U)), X(g.zeros3(_abi.POS_V))
                pbce, eta = X(g.zeros3(_abi.POS_H)), X(g.zeros2(_abi.POS_H))
                PressureForce(X(hm), (X(st["T"]), X(st["S"]), EOS), PFu, PFv, dg, CS, p_atm=X(p_atm), pbce=pbce, eta=eta, Boussinesq=False,
                              H_to_RZ=1.0)
                dg.sync()
                N = (lambda a: a.cpu().numpy()) if resident else (lambda a: a)
                for name, a, b in (("PFu", ref[0], PFu), ("PFv", ref[1], PFv), ("pbce", ref[2], pbce), ("eta", ref[3], eta)):
                    assert bits_equal(a, N(b)), (form, opts, (ni, nj, nk), p_atm is not None, resident, name, np.argwhere(a != N(b))[:3])
            dg.close()


# ---- the branches without a reconstruction: int_density_dz (analytic LINEAR / WRIGHT), the bulk mixed layer's tv_tmp, no EOS ----------
def _columnwise_uniform(g, st):
    """T and S that vary in the horizontal but not in the vertical: every PLM slope is zero"""
    T = np.ascontiguousarray(np.broadcast_to(st["T"][0], st["T"].shape)); S = np.ascontiguousarray(np.broadcast_to(st["S"][0], st["S"].shape))
    return T, S


@pytest.mark.parametrize("massw", [False, True])
def test_analytic_integrals_agree_with_the_quadrature_branch(oracle, massw):
    """RECONSTRUCT_FOR_PRESSURE = False (.testing/tc4; MOM_PressureForce_FV.F90:765): with vertically uniform T and S in every
    column the PLM branch (Boole quadrature of the EOS, MOM_density_integrals.F90:369) and int_density_dz's analytic forms
    (MOM_EOS_linear.F90:259, MOM_EOS_Wright.F90:389) integrate the same density profile: LINEAR to roundoff, WRIGHT to the
    truncation of its series in eps"""
    g, st = pgf_case(40, 26, 6, seed=7)
    Tc, Sc = _columnwise_uniform(g, st)
    Tu, Su = np.full_like(Tc, 11.0), np.full_like(Sc, 34.5)
    # the WRIGHT analytic form interpolates its three coefficients (cubic in T) between the columns, the quadrature form T and S:
    # with T and S uniform the two differ by the series truncation only, with columnwise T and S by O(dT^2) of the face
    for form, T, S, rtol in (("LINEAR", Tc, Sc, 1e-11), ("WRIGHT", Tu, Su, 1e-9), ("WRIGHT", Tc, Sc, 1e-3)):
        E = oracle.eos(form, 1000.0, -0.2, 0.8)
        A = oracle.pressureforce(g, oracle.pressureforce_cs(g, useMassWghtInterp=massw), E, st["h"], T, S)
        B = oracle.pressureforce(g, oracle.pressureforce_cs(g, useMassWghtInterp=massw, reconstruct=False), E, st["h"], T, S)
        C = oracle.pressureforce(g, oracle.pressureforce_cs(g, useMassWghtInterp=massw, use_ALE=False), E, st["h"], T, S)
        for q, pos in ((0, _abi.POS_U), (1, _abi.POS_V)):
            a, b = interior(g, A[q], pos), interior(g, B[q], pos)
            assert np.max(np.abs(a - b)) < rtol * np.max(np.abs(a)), (form, q, rtol)
            assert bits_equal(B[q], C[q])          # reconstruct = False and "no ALE" are the same branch
        assert bits_equal(A[2], B[2]) and bits_equal(A[3], B[3])      # pbce and eta do not depend on the branch


def test_resting_ocean_without_reconstruction(oracle):
    g = synth.make_grid(20, 16, 5, seed=1, land_frac=0.0, max_depth=4000.0)
    g.set_metric("bathyT", np.full_like(g.bathyT, 4000.0))
    shp = g.shape3(_abi.POS_H)
    h = np.empty(shp); T = np.empty(shp); S = np.empty(shp)
    for k, dz in enumerate([50.0, 150.0, 800.0, 1000.0, 2000.0]):
        h[k] = dz; T[k] = 20.0 - 4.0 * k; S[k] = 34.0 + 0.2 * k
    for form in ("WRIGHT", "LINEAR"):
        PFu, PFv, pbce, eta = oracle.pressureforce(g, oracle.pressureforce_cs(g, reconstruct=False), oracle.eos(form), h, T, S)
        assert np.max(np.abs(interior(g, PFu, _abi.POS_U))) < 1e-12 and np.max(np.abs(interior(g, PFv, _abi.POS_V))) < 1e-12
    # layered, no equation of state
    Rlay = np.array([1026.0, 1027.0, 1027.5, 1027.8, 1028.0]); gp = np.r_[g.g_Earth, g.g_Earth * np.diff(Rlay) / g.Rho0, 0.0]
    PFu, PFv, pbce, eta = oracle.pressureforce(g, oracle.pressureforce_cs(g, use_ALE=False, Rlay=Rlay, g_prime=gp), None, h, None, None)
    assert np.max(np.abs(interior(g, PFu, _abi.POS_U))) < 1e-12 and np.max(np.abs(interior(g, PFv, _abi.POS_V))) < 1e-12
    assert np.all(np.diff(interior(g, pbce), axis=0) > 0) and np.allclose(interior(g, pbce)[0], g.g_Earth)


def test_layered_form_is_the_linear_eos_with_T_equal_to_Rlay(oracle):
    """use_EOS = False (:775-789, Phillips_2layer with layer densities): the same force as the analytic LINEAR form with
    rho = T and T(k) = GV%Rlay(k), to roundoff; the increments of pbce are those of Set_pbce_Bouss with the EOS"""
    g, st = pgf_case(36, 22, 4, seed=11)
    Rlay = np.array([1025.0, 1026.5, 1027.25, 1027.9])
    gp = np.r_[g.g_Earth, g.g_Earth * np.diff(Rlay) / g.Rho0, 0.0]
    T = np.ascontiguousarray(Rlay[:, None, None] + 0 * st["T"]); S = np.zeros_like(T)
    A = oracle.pressureforce(g, oracle.pressureforce_cs(g, use_ALE=False, Rlay=Rlay, g_prime=gp), None, st["h"], None, None)
    B = oracle.pressureforce(g, oracle.pressureforce_cs(g, use_ALE=False), oracle.eos("LINEAR", 0.0, 1.0, 0.0), st["h"], T, S)
    # (in vanished layers the EOS form takes dz from a difference of interface heights and loses digits the layered form keeps)
    hi = interior(g, st["h"])
    thick = {0: np.minimum(hi[:, :, :-1], hi[:, :, 1:]) > 1.0, 1: np.minimum(hi[:, :-1, :], hi[:, 1:, :]) > 1.0}
    for q, pos in ((0, _abi.POS_U), (1, _abi.POS_V)):
        a, b = interior(g, A[q], pos), interior(g, B[q], pos)
        a, b = (a[:, :, 1:-1], b[:, :, 1:-1]) if q == 0 else (a[:, 1:-1, :], b[:, 1:-1, :])
        assert thick[q].sum() > 100 and np.max(np.abs(a - b)[thick[q]]) < 1e-9 * np.max(np.abs(a))
    da, db = np.diff(interior(g, A[2]), axis=0), np.diff(interior(g, B[2]), axis=0)
    assert np.allclose(da, db, rtol=1e-12, atol=0) and bits_equal(A[3], B[3])


def test_bulk_mixed_layer_replaces_light_layers(oracle):
    """nkmb > 0 (:650-670, layered runs with a bulk mixed layer, .testing/tc1): layers whose target density GV%Rlay(k) is lighter
    than the buffer layer's coordinate density take the buffer layer's T and S"""
    g, st = pgf_case(30, 20, 8, seed=13)
    E = oracle.eos("WRIGHT")
    nkmb = 4
    heavy = np.full(8, 2000.0); light = np.full(8, 0.0)
    base = oracle.pressureforce(g, oracle.pressureforce_cs(g, use_ALE=False), E, st["h"], st["T"], st["S"])
    same = oracle.pressureforce(g, oracle.pressureforce_cs(g, use_ALE=False, nkmb=nkmb, Rlay=heavy), E, st["h"], st["T"], st["S"])
    for a, b in zip(base, same):
        assert bits_equal(a, b)
    T2, S2 = st["T"].copy(), st["S"].copy()
    T2[nkmb:] = st["T"][nkmb - 1]; S2[nkmb:] = st["S"][nkmb - 1]
    want = oracle.pressureforce(g, oracle.pressureforce_cs(g, use_ALE=False), E, st["h"], T2, S2)
    got = oracle.pressureforce(g, oracle.pressureforce_cs(g, use_ALE=False, nkmb=nkmb, Rlay=light), E, st["h"], st["T"], st["S"])
    for a, b in zip(want, got):
        assert bits_equal(a, b)
    # a realistic table: the coordinate densities of a few of the layers straddle the buffer layer's
    rho_bl = np.array([oracle.eos_density(E, t, s, 2.0e7) for t, s in zip(interior(g, st["T"])[nkmb - 1].ravel(), interior(g, st["S"])[nkmb - 1].ravel())])
    Rlay = np.linspace(rho_bl.min() - 1.0, rho_bl.max() + 1.0, 8)
    mixed = oracle.pressureforce(g, oracle.pressureforce_cs(g, use_ALE=False, nkmb=nkmb, Rlay=Rlay), E, st["h"], st["T"], st["S"])
    assert not bits_equal(mixed[0], base[0]) and not bits_equal(mixed[0], got[0])


PCM_MODES = [("LINEAR", dict(reconstruct=False)), ("WRIGHT", dict(reconstruct=False)), ("WRIGHT", dict(use_ALE=False, nkmb=3)),
             ("LINEAR", dict(use_ALE=False, nkmb=2)), (None, dict(use_ALE=False))]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", PCM_MODES, ids=lambda m: f"{m[0]}-{'-'.join(f'{k}{v}' for k, v in m[1].items())}")
@pytest.mark.parametrize("massw", [False, True])
def test_gpu_parity_without_reconstruction(oracle, mode, massw):
    """RECONSTRUCT_FOR_PRESSURE = False / no ALE / bulk mixed layer / no equation of state: library == oracle, bit for bit"""
    import torch
    from mom6_amd.pressure_force import PressureForce, PressureForce_init, EOS_init
    from mom6_amd.tracer_advect import DeviceGrid
    form, kw = mode
    for (ni, nj, nk, topo) in [(70, 21, 5, (True, False)), (14, 10, 2, (False, False)), (10, 8, 8, (True, False)), (130, 9, 75, (True, True))]:
        if kw.get("nkmb", 0) >= nk:
            continue
        g, st = pgf_case(ni, nj, nk, seed=ni, reentrant_x=topo[0], reentrant_y=topo[1])
        E = oracle.eos(form, 1000.0, -0.2, 0.8) if form else None
        Rlay = gp = None
        if form is None or kw.get("nkmb", 0) > 0:
            lo, hi = (1026.0, 1028.0) if form != "LINEAR" else (990.0, 1030.0)
            Rlay = np.linspace(lo, hi, nk); gp = np.r_[g.g_Earth, g.g_Earth * np.diff(Rlay) / g.Rho0, 0.0]
        cs = oracle.pressureforce_cs(g, useMassWghtInterp=massw, Rlay=Rlay, g_prime=gp, **kw)
        rng = np.random.default_rng(ni)
        dg = DeviceGrid(g)
        CS = PressureForce_init(g, useMassWghtInterp=massw, Rlay=Rlay, g_prime=gp, reconstruct=kw.get("reconstruct", True),
                                use_ALE=kw.get("use_ALE", True), nk_rho_varies=kw.get("nkmb", 0))
        EOS = EOS_init(form, 1000.0, -0.2, 0.8) if form else None
        for p_atm in (None, np.ascontiguousarray(1.0e5 + 500.0 * rng.standard_normal(g.shape2(_abi.POS_H)))):
            Tn, Sn = (st["T"], st["S"]) if form else (None, None)
            ref = oracle.pressureforce(g, cs, E, st["h"], Tn, Sn, p_atm)
            for resident in (False, True):
                X = (lambda a: None if a is None else torch.from_numpy(a.copy()).cuda()) if resident else \
                    (lambda a: None if a is None else a.copy())
                PFu, PFv = X(g.zeros3(_abi.POS_U)), X(g.zeros3(_abi.POS_V))
                pbce, eta = X(g.zeros3(_abi.POS_H)), X(g.zeros2(_abi.POS_H))
                PressureForce(X(st["h"]), (X(Tn), X(Sn), EOS), PFu, PFv, dg, CS, p_atm=X(p_atm), pbce=pbce, eta=eta)
                dg.sync()
                N = (lambda a: a.cpu().numpy()) if resident else (lambda a: a)
                for name, a, b in (("PFu", ref[0], PFu), ("PFv", ref[1], PFv), ("pbce", ref[2], pbce), ("eta", ref[3], eta)):
                    assert bits_equal(a, N(b)), (mode, massw, (ni, nj, nk), p_atm is not None, resident, name, np.argwhere(a != N(b))[:3])
        dg.close()


# ---- test_EOS_consistency (MOM_EOS.F90:2166-2524): the relations the reference's EOS_unit_tests hold every form to ----------------
EOS_POINTS = [(25.0, 35.0, 1.0e7),                                              # the point of EOS_unit_tests (:1917-1996)
              (2.0, 34.7, 4.0e7), (-1.5, 33.0, 0.0), (15.0, 36.5, 5.0e6), (28.0, 30.0, 1.0e5), (5.0, 20.0, 2.0e7)]


@pytest.mark.parametrize("form", ["WRIGHT", "WRIGHT_FULL", "WRIGHT_REDUCED", "UNESCO", "LINEAR"])
@pytest.mark.parametrize("pt", EOS_POINTS, ids=lambda p: f"T{p[0]}S{p[1]}p{p[2]:.0e}")
def test_EOS_consistency(oracle, form, pt):
    """the checks of test_EOS_consistency that concern the functions on the hot path, with its perturbations (dT = 0.1, dS = 0.5,
    dp = 1e5), its 4th-order differences (first_deriv :2443), its tolerances (tol = 1000 eps, r_tol = 50 * 10 eps) and its
    convergence criterion check_FD (:2494): |fd(1) - val| < 1.2 |fd(2) - val| / 2^4 + tol"""
    T0, S0, p0 = pt
    E = oracle.eos(form, 1000.0, -0.2, 0.8)
    dT, dS, order = 0.1, 0.5, 4
    tol = 1000.0 * EPS; r_tol = 50.0 * 10.0 * EPS
    rho_ref = 1000.0; spv_ref = 1.0 / rho_ref
    rho = lambda T, S, p: oracle.eos_density(E, T, S, p, rho_ref=rho_ref)
    r000 = rho(T0, S0, p0)
    # :2318-2326 rho and 1/spv agree; :2331-2341 with and without the reference value
    spv = oracle.eos_spec_vol_anomaly(E, T0, S0, p0, spv_ref)
    assert abs((rho_ref + r000) * (spv_ref + spv) - 1.0) < tol
    rho_nooff = oracle.eos_density(E, T0, S0, p0)
    assert abs(rho_nooff - (rho_ref + r000)) < tol * rho_nooff

    def first_deriv(R, dx):      # R(-2..2), 4th order :2451
        return (8.0 * (R[3] - R[1]) - (R[4] - R[0])) / (12.0 * dx)
    fdT = [first_deriv([rho(T0 + n * dT * i, S0, p0) for i in range(-2, 3)], n * dT) for n in (1, 2)]
    fdS = [first_deriv([rho(T0, S0 + n * dS * j, p0) for j in range(-2, 3)], n * dS) for n in (1, 2)]
    drho_dT, drho_dS = oracle.eos_density_derivs(E, T0, S0, p0)
    count_fac = 18.0 / 12.0
    for val, fd, d in ((drho_dT, fdT, dT), (drho_dS, fdS, dS)):
        tol_here = tol * abs(val) + count_fac * r_tol / d
        assert abs(fd[0] - val) < (1.2 * abs(fd[1] - val) / 2 ** order + abs(tol_here)), (form, pt, val, fd)
