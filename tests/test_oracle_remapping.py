"""The remapping oracle (oracle/remapping.c) against (1) the known-answer vectors of the reference's own
remapping_unit_tests (tests/golden/remapping_unit_tests.json, data from src/ALE/MOM_remapping.F90:1339-1569,
checked with the reference's tolerances) and (2) the reference PLM/PCM sources compiled unmodified
(oracle/_ref, bit-for-bit on random columns; only where oracle/_ref was built)."""
import json
import os

import numpy as np
import pytest

from helpers import bits_equal

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "remapping_unit_tests.json")))
EPS = GOLD["eps"]


def check(u, u_true, tol=0.0):
    # test_answer, MOM_remapping.F90:1683: fails if abs(u - u_true) > tol
    assert not np.any(np.abs(np.asarray(u) - np.asarray(u_true)) > tol), (u, u_true, tol)


def test_remapping_core_w_ppm_h4(oracle):
    c = GOLD["remapping_core_w"]
    dx = oracle.dz_from_h1h2(c["h0"], c["h1"])
    u1 = oracle.remapping_core_w(c["scheme"], c["h0"], c["u0"], dx, c["h_neglect"], c["h_neglect_edge"])
    check(u1, c["u1"], c["tol_eps"] * EPS)


def test_remap_via_sub_cells_ppm(oracle):
    c = GOLD["remap_via_sub_cells_ppm"]
    E = oracle.edge_values_explicit_h4(c["h0"], c["u0"], c["edge_h_neglect"])
    E, co = oracle.ppm_reconstruction(c["h0"], c["u0"], E, c["h_neglect"], extrapolate=True)
    u2, _ = oracle.remap_via_sub_cells(c["h0"], c["u0"], E, co, c["h2"], oracle.INT_PPM)
    check(u2, c["u2"], c["tol_eps"] * EPS)
    # the two further calls of the reference test (:1434-1440) must simply run and conserve
    for h in ([0.125] * 6, [2.25, 1.5, 1.0]):
        u, _ = oracle.remap_via_sub_cells(c["h0"], c["u0"], E, co, h, oracle.INT_PPM)
        assert np.all(np.isfinite(u))


def test_pcm(oracle):
    c = GOLD["pcm"]
    E, co = oracle.pcm_reconstruction(c["u"])
    check(E[0], c["left"]); check(E[1], c["right"]); check(co[0], c["P0"])


@pytest.mark.parametrize("c", GOLD["plm"], ids=lambda c: c["label"])
def test_plm_tables(oracle, c):
    E, co = oracle.plm_reconstruction(c["h"], c["u"], 1e-30)
    check(E[0], c["left"]); check(E[1], c["right"]); check(co[0], c["P0"]); check(co[1], c["P1"])


@pytest.mark.parametrize("c", GOLD["edge_values_explicit_h4"], ids=lambda c: c["label"])
def test_edge_values_explicit_h4(oracle, c):
    E = oracle.edge_values_explicit_h4(c["h"], c["u"], c["h_neglect"])
    check(E[0], c["left"], c["tol_left"]); check(E[1], c["right"], c["tol_right"])


@pytest.mark.parametrize("c", GOLD["ppm_reconstruction"], ids=lambda c: c["label"])
def test_ppm_tables(oracle, c):
    E0 = np.array([c["left_in"], c["right_in"]])
    E, co = oracle.ppm_reconstruction(c["h"], c["u"], E0)
    if "left" in c:
        check(E[0], c["left"]); check(E[1], c["right"])
    check(co[0], c["P0"]); check(co[1], c["P1"]); check(co[2], c["P2"])


def test_plm_vanished_layers_and_remap(oracle):
    c = GOLD["plm_vanished"]
    E, co = oracle.plm_reconstruction(c["h"], c["u"], 1e-30)
    check(E[0], c["left"]); check(E[1], c["right"])
    u1, _ = oracle.remap_via_sub_cells(c["h"], c["u"], E, co, c["h1"], oracle.INT_PLM)
    check(u1, c["u1"])


# ---- properties the scheme guarantees (conservative, bounded) on random columns ------------------------
@pytest.mark.parametrize("scheme", ["PCM", "PLM", "PPM_H4", "PPM_IH4", "PPM_CW"])
def test_remap_conserves_and_is_bounded(oracle, scheme):
    rng = np.random.default_rng(3)
    for trial in range(200):
        n0, n1 = int(rng.integers(5, 40)), int(rng.integers(1, 40))
        h0 = rng.random(n0) * 10.0
        h0[rng.random(n0) < 0.15] = 0.0          # vanished layers
        if h0.sum() == 0.0:
            h0[0] = 1.0
        w = rng.random(n1); w[rng.random(n1) < 0.15] = 0.0
        if w.sum() == 0.0:
            w[0] = 1.0
        h1 = w / w.sum() * h0.sum()
        u0 = rng.standard_normal(n0) * 5.0 + 10.0
        u1 = oracle.remapping_core_h(scheme, h0, u0, h1, 1e-30, 1e-10, boundary_extrapolation=False)
        tot0, tot1 = float(np.dot(h0, u0)), float(np.dot(h1, u1))
        assert abs(tot1 - tot0) <= 1e-12 * max(1.0, np.dot(h0, np.abs(u0))), (scheme, trial)
        assert u1.max() <= u0.max() + 1e-12 and u1.min() >= u0.min() - 1e-12


# ---- PPM_IH4 / PPM_CW edge values: the reference holds no known-answer vectors for edge_values_implicit_h4 and
# edge_values_explicit_h4cw (parity unpinned for these two); what pins the restatement is what the schemes guarantee
def _cell_means(poly, h):
    """exact cell averages and edge values of a polynomial (numpy poly1d) on the grid of widths h"""
    x = np.concatenate([[0.0], np.cumsum(h)])
    P = poly.integ()
    return (P(x[1:]) - P(x[:-1])) / h, poly(x)


def test_implicit_h4_is_exact_for_cubics(oracle):
    """edge_values_implicit_h4 is fourth-order: with the one-sided cubic fits that close the system (end_value_h4) it
    reproduces the edge values of any cubic from its cell averages, on any grid (regrid_edge_values.F90:470-490)"""
    rng = np.random.default_rng(5)
    for trial in range(50):
        n = int(rng.integers(5, 30))
        h = 0.5 + rng.random(n) * 3.0
        poly = np.poly1d(rng.standard_normal(4))
        ubar, edges = _cell_means(poly, h)
        L, R = oracle.edge_values("ih4", h, ubar)
        scale = np.abs(edges).max() + 1.0
        assert np.abs(L - edges[:-1]).max() <= 1e-9 * scale and np.abs(R - edges[1:]).max() <= 1e-9 * scale, trial
        assert np.array_equal(R[:-1], L[1:])      # one value per interface


def test_explicit_h4cw_on_linear_and_monotone_data(oracle):
    """edge_values_explicit_h4cw (Colella & Woodward eqs. 1.6-1.8): exact interior edges for linear data on a uniform grid,
    PCM at the two ends (:437-449), and after PPM_monotonicity every cell's parabola is monotone"""
    n = 12
    h = np.ones(n); u = 2.0 + 3.0 * (np.arange(n) + 0.5)
    L, R = oracle.edge_values("h4cw", h, u)
    x = np.arange(n + 1.0)
    assert np.allclose(L[2:n - 1], 2.0 + 3.0 * x[2:n - 1], rtol=0, atol=1e-12)
    assert L[0] == u[0] and R[0] == u[0] and L[1] == u[0] and R[n - 2] == u[n - 1] and L[n - 1] == u[n - 1] and R[n - 1] == u[n - 1]
    rng = np.random.default_rng(8)
    for trial in range(50):
        n = int(rng.integers(5, 30))
        h = 0.2 + rng.random(n) * 3.0
        u = np.cumsum(rng.random(n))      # monotone data: the remapped values stay within the neighbours
        h1 = np.full(n + 3, h.sum() / (n + 3))
        u1 = oracle.remapping_core_h("PPM_CW", h, u, h1, 1e-30, 1e-10, boundary_extrapolation=False)
        assert np.all(np.diff(u1) >= -1e-12), trial
        assert abs(np.dot(h1, u1) - np.dot(h, u)) <= 1e-11 * np.dot(h, np.abs(u))


def test_small_columns_fall_back_as_the_reference_does(oracle):
    """n0 <= 4: PPM_IH4 drops to PPM_H4 (identical results), PPM_CW is kept (MOM_remapping.F90:289-295)"""
    h = np.array([1.0, 2.0, 1.5, 0.5]); u = np.array([1.0, 3.0, 2.0, 5.0]); h1 = np.array([2.0, 1.0, 2.0])
    a = oracle.remapping_core_h("PPM_IH4", h, u, h1); b = oracle.remapping_core_h("PPM_H4", h, u, h1)
    assert np.array_equal(a, b)
    c = oracle.remapping_core_h("PPM_CW", h, u, h1)
    assert abs(np.dot(h1, c) - np.dot(h, u)) < 1e-12


# ---- bit-for-bit against the reference's own code where it compiles with no stand-ins ------------------
def test_plm_matches_reference_build(oracle):
    R = oracle.ref_lib()
    if R is None:
        pytest.skip("oracle/_ref not built (the reference sources only exist in the build container)")
    rng = np.random.default_rng(11)
    import ctypes as C
    dp = C.POINTER(C.c_double)
    P = lambda a: a.ctypes.data_as(dp)
    for trial in range(300):
        n = int(rng.integers(2, 60))
        h = rng.random(n) * 10.0 ** rng.integers(-3, 3)
        h[rng.random(n) < 0.2] = 0.0
        u = rng.standard_normal(n) * 10.0 ** rng.integers(-2, 3)
        if trial % 5 == 0:
            u = np.round(u)                      # ties / exact extrema
        for extrap in (0, 1):
            E, co = oracle.plm_reconstruction(h, u, 1e-30, extrapolate=bool(extrap))
            Er, cr = np.zeros((2, n)), np.zeros((2, n))
            R.ref_plm_reconstruction(n, P(h), P(u), P(Er), P(cr), 1e-30, extrap)
            assert bits_equal(E, Er) and bits_equal(co[:2], cr), (trial, extrap)
    for _ in range(2000):
        a = rng.standard_normal(7) * 10.0 ** rng.integers(-3, 3, 7)
        hh = np.abs(a[:3])
        assert oracle._remap_lib().orc_plm_slope_wa(hh[0], hh[1], hh[2], 1e-30, a[3], a[4], a[5]) == \
            R.ref_plm_slope_wa(hh[0], hh[1], hh[2], 1e-30, a[3], a[4], a[5])
        assert oracle._remap_lib().orc_plm_extrapolate_slope(hh[0], hh[1], 1e-30, a[3], a[4]) == \
            R.ref_plm_extrapolate_slope(hh[0], hh[1], 1e-30, a[3], a[4])


def test_hybgen_matches_reference_build(oracle):
    """hybgen_plm_coefs / hybgen_ppm_coefs / hybgen_weno_coefs (src/ALE/MOM_hybgen_remap.F90, compiled unmodified into oracle/_ref):
    the restatement gives the reference's bits on random columns with vanished layers, ties and extrema"""
    R = oracle.ref_lib()
    if R is None or not hasattr(R, "ref_hybgen_plm"):
        pytest.skip("oracle/_ref not built (the reference sources only exist in the build container)")
    import ctypes as C
    dp = C.POINTER(C.c_double)
    P = lambda a: a.ctypes.data_as(dp)
    rng = np.random.default_rng(23)
    for trial in range(400):
        n = int(rng.integers(5, 80))
        h = rng.random(n) * 10.0 ** rng.integers(-3, 3)
        h[rng.random(n) < 0.2] = 0.0
        s = rng.standard_normal(n) * 10.0 ** rng.integers(-2, 3)
        if trial % 4 == 0:
            s = np.round(s)                      # ties / exact extrema
        if trial % 7 == 0:
            s = np.sort(s)                       # monotone profiles: the limiters stay open
        thin = 1e-30 if trial % 3 else 1e-3
        sl = np.zeros(n); R.ref_hybgen_plm(n, P(s), P(h), P(sl), thin)
        assert bits_equal(oracle.hybgen_coefs("plm", s, h, thin), sl), ("plm", trial)
        for which, f in (("ppm", R.ref_hybgen_ppm), ("weno", R.ref_hybgen_weno)):
            Er = np.zeros((2, n)); f(n, P(s), P(h), P(Er), thin)
            assert bits_equal(oracle.hybgen_coefs(which, s, h, thin), Er), (which, trial)


@pytest.mark.parametrize("scheme", ["PLM_HYBGEN", "PPM_HYBGEN", "WENO_HYBGEN"])
def test_hybgen_schemes_conserve_and_stay_bounded(oracle, scheme):
    rng = np.random.default_rng(31)
    for trial in range(60):
        n0, n1 = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        h0 = rng.random(n0) + 0.01; h0[rng.random(n0) < 0.15] = 0.0
        if h0.sum() == 0.0:
            h0[0] = 1.0
        h1 = rng.random(n1) + 0.01; h1 *= h0.sum() / h1.sum()
        u0 = rng.standard_normal(n0)
        u1 = oracle.remapping_core_h(scheme, h0, u0, h1, boundary_extrapolation=bool(trial % 2))
        assert abs((u1 * h1).sum() - (u0 * h0).sum()) <= 1e-12 * max(1.0, np.abs(u0 * h0).sum())
        if not trial % 2:
            assert u1.min() >= u0.min() - 1e-12 and u1.max() <= u0.max() + 1e-12
    # PPM_HYBGEN is the scheme PPM_CW re-expresses (MOM_remapping.F90:317): the same values wherever no layer is thinner than `thin`
    h0 = rng.random(30) + 0.5; u0 = rng.standard_normal(30); h1 = rng.random(25) + 0.5; h1 *= h0.sum() / h1.sum()
    if scheme == "PPM_HYBGEN":
        a = oracle.remapping_core_h("PPM_HYBGEN", h0, u0, h1, h_neglect=1e-30, h_neglect_edge=1e-30)
        b = oracle.remapping_core_h("PPM_CW", h0, u0, h1, h_neglect=1e-30, h_neglect_edge=1e-30)
        assert np.allclose(a, b, rtol=0, atol=1e-13)


@pytest.mark.parametrize("scheme", ["PQM_IH4IH3", "PQM_IH6IH5"])
def test_pqm_conserves_and_keeps_constants(oracle, scheme):
    """PQM_IH4IH3 (edge_values_implicit_h4 + edge_slopes_implicit_h3 + PQM_limiter [+ PQM_boundary_extrapolation_v1], the quartic's integrals of
    average_value_ppoly) and PQM_IH6IH5 (edge_values_implicit_h6 + edge_slopes_implicit_h5 with their 6x6 systems): the column integral is kept, a constant stays the constant, and a column of four or fewer cells falls back to the
    scheme PPM_H4 / PLM / PCM gives (MOM_remapping.F90:288-294).  Bit-for-bit agreement with the reference's own build of these routines is
    tests/test_reference_kernels.py::test_reference_ale_regrid_and_remap_equal_the_oracle[PQM_IH4IH3-...]."""
    rng = np.random.default_rng(77)
    for trial in range(80):
        n0, n1 = int(rng.integers(6, 40)), int(rng.integers(1, 40))
        h0 = rng.random(n0) + 0.01
        if trial % 3 == 0:
            h0[rng.random(n0) < 0.2] = 1.0e-3
        h1 = rng.random(n1) + 0.01; h1 *= h0.sum() / h1.sum()
        u0 = rng.standard_normal(n0) * (10.0 if trial % 5 == 0 else 1.0)
        if trial % 4 == 0:
            u0 = np.round(u0)
        if trial % 7 == 0:
            u0 = np.sort(u0)
        u1 = oracle.remapping_core_h(scheme, h0, u0, h1, boundary_extrapolation=bool(trial % 2))
        assert np.isfinite(u1).all()
        assert abs((u1 * h1).sum() - (u0 * h0).sum()) <= 1e-12 * max(1.0, np.abs(u0 * h0).sum()), trial
        c = oracle.remapping_core_h(scheme, h0, np.full(n0, 3.25), h1, boundary_extrapolation=bool(trial % 2))
        assert np.abs(c - 3.25).max() <= (0.0 if scheme == "PQM_IH4IH3" else 1e-10)      # (the 6x6 systems do not reproduce a constant to the bit)
    for n0, same in ((4, "PPM_H4"), (3, "PLM"), (2, "PLM"), (1, "PCM")):
        h0 = rng.random(n0) + 0.1; u0 = rng.standard_normal(n0); h1 = rng.random(6) + 0.1; h1 *= h0.sum() / h1.sum()
        assert bits_equal(oracle.remapping_core_h(scheme, h0, u0, h1), oracle.remapping_core_h(same, h0, u0, h1))
