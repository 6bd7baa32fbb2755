"""The reference's own known-answer vectors pushed THROUGH THE HIP PATH (the C ABI on the GPU), with the reference's tolerances:
  * tests/golden/remapping_unit_tests.json (src/ALE/MOM_remapping.F90:1339-1569; test_answer :1683-1711) through
    mom6hip_ale_remap_tracers -- the whole-column cases (remapping_core_w with PPM_H4 :1390-1399; the PLM column with vanished
    layers :1559-1569);
  * tests/golden/eos_check_values.json (src/equation_of_state/MOM_EOS.F90:1917-1996; test_EOS_consistency :2166) through
    mom6hip_calculate_density, with and without rho_ref.
(tests/test_oracle_remapping.py and test_pressure_force.py::test_eos_check_values pin the ORACLE with the same vectors on the CPU;
the parity tests tie the library to the oracle on random inputs.  These tests close the chain directly.)"""
import json
import os

import numpy as np
import pytest

from mom6_amd import _abi, synth

HERE = os.path.dirname(__file__)
REMAP = json.load(open(os.path.join(HERE, "golden", "remapping_unit_tests.json")))
EOSV = json.load(open(os.path.join(HERE, "golden", "eos_check_values.json")))
EPS = np.finfo(np.float64).eps


def _columns(g, col, nk):
    """a 3-D h-point field whose every column is `col` (padded with zeros to nk layers)"""
    a = np.zeros(g.shape3(_abi.POS_H))
    for k, x in enumerate(col):
        a[k] = x
    return a


def _remap_through_library(scheme, h0, u0, h1, space):
    import torch
    from mom6_amd.ale import ALE_remap_tracers, initialize_remapping
    from mom6_amd.tracer_advect import DeviceGrid
    nk = max(len(h0), len(h1))
    g = synth.make_grid(12, 6, nk, seed=3, land_frac=0.0)
    g.H_subroundoff = 1.0e-30          # h_neglect = h_neglect_edge = 1e-30, as the reference's test sets them
    dg = DeviceGrid(g)
    T = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if space == "device" else (lambda a: np.ascontiguousarray(a).copy())
    N = lambda a: a if isinstance(a, np.ndarray) else a.cpu().numpy()
    h_old, h_new, u = T(_columns(g, h0, nk)), T(_columns(g, h1, nk)), T(_columns(g, u0, nk))
    ALE_remap_tracers(initialize_remapping(scheme), dg, h_old, h_new, [u])
    dg.sync()
    out = N(u)
    dg.close()
    sj, si = g.csl(0)
    cols = out[:, sj, si].reshape(nk, -1)
    assert np.all(cols == cols[:, :1])          # every column got the same answer
    return cols[:len(h1), 0]


@pytest.mark.gpu
@pytest.mark.parametrize("space", ["device", "host"])
def test_remapping_core_ppm_h4_vector_through_the_library(space):
    """MOM_remapping.F90:1390-1399: h0 = 4 x 0.75, u0 = 9, 3, -3, -9 remapped with PPM_H4 onto h1 = 3 x 1 gives 8, 0, -8 within 2 eps
    (the target column is padded with one vanished layer: ALE_remap_tracers keeps the number of layers)"""
    c = REMAP["remapping_core_w"]
    u1 = _remap_through_library(c["scheme"], c["h0"], c["u0"], c["h1"], space)
    assert not np.any(np.abs(u1 - np.array(c["u1"])) > c["tol_eps"] * EPS), u1


@pytest.mark.gpu
@pytest.mark.parametrize("space", ["device", "host"])
def test_plm_vanished_layers_vector_through_the_library(space):
    """MOM_remapping.F90:1559-1569: h = 0, 1, 1, 0 with u = 5, 4, 2, 1 remapped with PLM onto h1 = 1, 1 gives exactly 4, 2"""
    c = REMAP["plm_vanished"]
    u1 = _remap_through_library("PLM", c["h"], c["u"], c["h1"], space)
    assert np.array_equal(u1, np.array(c["u1"])), u1


@pytest.mark.gpu
@pytest.mark.parametrize("space", ["device", "host"])
def test_pcm_vector_through_the_library(space):
    """MOM_remapping.F90:1451-1458 (PCM: edge values and P0 equal the cell means) seen through the remap: a PCM remap onto the
    same grid returns the cell means exactly, and onto a grid of half cells returns each mean twice"""
    c = REMAP["pcm"]
    u = c["u"]
    same = _remap_through_library("PCM", [1.0] * 3 + [0.0] * 3, u + [0.0] * 3, [1.0] * 3 + [0.0] * 3, space)
    assert np.array_equal(same[:3], np.array(c["P0"]))
    halves = _remap_through_library("PCM", [1.0] * 3 + [0.0] * 3, u + [0.0] * 3, [0.5] * 6, space)
    assert np.array_equal(halves, np.repeat(np.array(c["P0"]), 2))


@pytest.mark.gpu
@pytest.mark.parametrize("c", EOSV["cases"], ids=lambda c: c["form"] + c["_line"][:4])
@pytest.mark.parametrize("space", ["device", "host"])
def test_eos_check_values_through_the_library(c, space):
    """abs(rho_check - (rho_ref + rho)) < 1000 eps (rho_ref + rho) (MOM_EOS.F90:2306) for the anomaly form, and the form
    without a reference value agrees with it within the same tolerance (:2333)"""
    import torch
    from mom6_amd.pressure_force import EOS_init, calculate_density
    from mom6_amd.tracer_advect import DeviceGrid
    g = synth.make_grid(12, 6, 2, seed=3)
    dg = DeviceGrid(g)
    E = EOS_init(c["form"], c.get("Rho_T0_S0", 1000.0), c.get("dRho_dT", -0.2), c.get("dRho_dS", 0.8))
    n = 130
    mk = (lambda v: torch.full((n,), v, dtype=torch.float64, device="cuda")) if space == "device" else (lambda v: np.full(n, v))
    N = lambda a: a if isinstance(a, np.ndarray) else a.cpu().numpy()
    T, S, p, rho, rho0 = mk(c["T"]), mk(c["S"]), mk(c["p"]), mk(0.0), mk(0.0)
    rho_ref = EOSV["rho_ref"]
    calculate_density(T, S, p, rho, E, dg, rho_ref=rho_ref)
    calculate_density(T, S, p, rho0, E, dg)
    dg.sync()
    r, r0 = N(rho), N(rho0)
    dg.close()
    tol = EOSV["rel_tol_eps"] * EPS
    assert np.all(np.abs(c["rho_check"] - (rho_ref + r)) < tol * (rho_ref + r)), r[:2]
    assert np.all(np.abs(r0 - (rho_ref + r)) < tol * r0)
