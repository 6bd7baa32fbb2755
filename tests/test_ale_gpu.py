"""GPU parity: libmom6hip's ALE_remap_tracers (C ABI) against the remapping oracle, which is itself pinned by
the reference's known-answer vectors (tests/test_oracle_remapping.py).  Bar: bit-exact fp64."""
import numpy as np
import pytest
import torch

from mom6_amd import _abi, synth
from mom6_amd.ale import ALE_remap_tracers, initialize_remapping
from mom6_amd.tracer_advect import DeviceGrid
from helpers import bits_equal, interior

pytestmark = pytest.mark.gpu


def remap_case(ni, nj, nk, ntr=3, seed=0, vanish=0.15):
    g = synth.make_grid(ni, nj, nk, seed=seed + 50, land_frac=0.2)
    rng = np.random.default_rng(seed)
    shp = g.shape3(_abi.POS_H)
    h_old = rng.random(shp) * 20.0 + 0.5
    h_old[rng.random(shp) < vanish] = 0.0
    h_old[0][h_old.sum(0) == 0.0] = 1.0
    w = rng.random(shp) + 0.05
    w[rng.random(shp) < vanish] = 0.0
    w[0][w.sum(0) == 0.0] = 1.0
    h_new = w / w.sum(0, keepdims=True) * h_old.sum(0, keepdims=True)
    # a few columns whose target is deeper / shallower than the source (remap_via_sub_cells :611-646)
    h_new[:, 5, 6] *= 1.25
    h_new[:, 6, 7] *= 0.8
    tr = [np.ascontiguousarray(rng.standard_normal(shp) * (m + 1) + 10.0 * m) for m in range(ntr)]
    tr[-1][:] = np.round(tr[-1])          # exact ties and extrema
    return g, np.ascontiguousarray(h_old), np.ascontiguousarray(h_new), tr


@pytest.mark.parametrize("scheme", ["PCM", "PLM", "PLM_HYBGEN", "PPM_H4", "PPM_IH4", "PPM_HYBGEN", "WENO_HYBGEN", "PPM_CW", "PQM_IH4IH3", "PQM_IH6IH5"])
@pytest.mark.parametrize("extrap", [False, True])
@pytest.mark.parametrize("nk", [2, 3, 4, 5, 8, 20, 75])
def test_remap_tracers_parity(oracle, scheme, extrap, nk):
    if scheme == "PQM_IH6IH5" and nk == 5:
        pytest.skip("the reference's edge_values_implicit_h6 reads six cells of a five-cell column")
    g, h_old, h_new, tr = remap_case(70, 12, nk, seed=nk)
    if scheme == "PQM_IH6IH5":
        # edge_slopes_implicit_h5 takes the six widths at either end as they are (regrid_edge_values.F90:1166): two vanished cells among them make
        # its system singular, a FATAL error in the reference
        h_old = np.maximum(h_old, 1.0e-3)
    if scheme.startswith("PQM") and extrap:
        # PQM_boundary_extrapolation_v1 divides by the widths of the two cells at the bottom (PQM_functions.F90:672, :682, :709): a vanished
        # cell there is 0/0 in the reference too, with a NaN whose sign depends on the machine; the layers keep a minimum thickness instead
        h_old = np.maximum(h_old, 1.0e-3)
    ref = [t.copy() for t in tr]
    oracle.ale_remap_tracers(g, scheme, h_old, h_new, ref, boundary_extrapolation=extrap)
    dg = DeviceGrid(g)
    CS = initialize_remapping(scheme, boundary_extrapolation=extrap)
    out = [t.copy() for t in tr]
    ALE_remap_tracers(CS, dg, h_old, h_new, out)                       # HOST memspace
    d = [torch.from_numpy(t.copy()).cuda() for t in tr]
    ALE_remap_tracers(CS, dg, torch.from_numpy(h_old).cuda(), torch.from_numpy(h_new).cuda(), d)   # DEVICE
    dg.sync()
    for m in range(len(tr)):
        assert bits_equal(ref[m], out[m]), (scheme, extrap, nk, m, np.argwhere(ref[m] != out[m])[:3])
        assert bits_equal(ref[m], d[m].cpu().numpy()), (scheme, extrap, nk, m)
    # land columns untouched, ocean columns changed
    assert not bits_equal(interior(g, ref[0]), interior(g, tr[0]))
    dg.close()


@pytest.mark.parametrize("form", ["0", "1", "2"])
@pytest.mark.parametrize("nk", [6, 7, 20, 75, 100])
def test_remap_stream_and_wave_kernels_agree(oracle, form, nk, monkeypatch):
    """PPM_H4 without boundary extrapolation on >= 6 layers takes the streaming kernel (MOM6HIP_ALE_STREAM = 1: a field a launch, 2: two
    fields a launch; 0: the wave-per-column kernel): the oracle's bits in every form -- random grids with vanished layers, columns whose
    new grid runs far ahead of the old one (the values written to the side array) or far behind it, identical grids, z*-like shifts,
    deeper and shallower targets, a conc_underflow"""
    monkeypatch.setenv("MOM6HIP_ALE_STREAM", form)
    g, h_old, h_new, tr = remap_case(70, 12, nk, ntr=5, seed=100 + nk)
    rng = np.random.default_rng(nk)
    H = h_old.sum(0)
    hl = g.halo
    # rows of special columns
    j = hl + 1      # everything in the top layer of the old grid, a uniform new grid: the targets run ahead of the sources
    h_old[:, j, :] = 0.0; h_old[0, j, :] = H[j, :]; h_new[:, j, :] = H[j, :] / nk
    j = hl + 2      # the reverse: a uniform old grid, everything in the bottom layer of the new one
    h_old[:, j, :] = H[j, :] / nk; h_new[:, j, :] = 0.0; h_new[nk - 1, j, :] = H[j, :]
    j = hl + 3      # identical grids
    h_new[:, j, :] = h_old[:, j, :]
    j = hl + 4      # z*-like: the old grid stretched by a few per cent, vanished layers at the bottom
    h_old[:, j, :] = np.linspace(2.0, 80.0, nk)[:, None]; h_old[nk - nk // 4:, j, :] = 1.0e-3
    h_new[:, j, :] = h_old[:, j, :] * (1.0 + 0.03 * rng.standard_normal(h_old.shape[2]))[None, :]; h_new[nk - nk // 4:, j, :] = 1.0e-3
    j = hl + 5      # thin targets in thick sources in the middle of the column, then the reverse
    h_old[:, j, :] = 1.0e-9; h_old[nk // 3, j, :] = 50.0; h_old[2 * nk // 3, j, :] = 50.0
    h_new[:, j, :] = (100.0 + nk * 1.0e-9) / nk
    ref = [t.copy() for t in tr]
    cu = [0.0, 0.0, 1.0e-2, 0.0, 0.3]
    oracle.ale_remap_tracers(g, "PPM_H4", h_old, h_new, ref, conc_underflow=cu)
    dg = DeviceGrid(g)
    CS = initialize_remapping("PPM_H4")
    d = [torch.from_numpy(t.copy()).cuda() for t in tr]
    ALE_remap_tracers(CS, dg, torch.from_numpy(h_old).cuda(), torch.from_numpy(h_new).cuda(), d, conc_underflow=cu)
    dg.sync()
    for m in range(len(tr)):
        got = d[m].cpu().numpy()
        assert bits_equal(ref[m], got), (form, nk, m, np.argwhere(ref[m] != got)[:5])
    dg.close()


def test_remap_conserves_on_gpu(oracle):
    g, h_old, h_new, tr = remap_case(40, 10, 30, ntr=2, seed=4)
    h_new = h_new.copy(); h_new[:, 5, 6] /= 1.25; h_new[:, 6, 7] /= 0.8     # equal column totals again
    dg = DeviceGrid(g)
    out = [t.copy() for t in tr]
    ALE_remap_tracers(initialize_remapping("PPM_H4"), dg, h_old, h_new, out)
    m = interior(g, g.mask2dT) > 0            # only ocean columns of the compute domain are remapped
    for a, b in zip(tr, out):
        c0 = interior(g, (a * h_old).sum(0))[m]; c1 = interior(g, (b * h_new).sum(0))[m]
        assert np.max(np.abs(c1 - c0) / np.maximum(1.0, interior(g, (np.abs(a) * h_old).sum(0))[m])) < 1e-13
    dg.close()


def test_remap_errors():
    from mom6_amd._lib import Mom6HipError
    with pytest.raises(Mom6HipError, match="REMAPPING_SCHEME"):
        initialize_remapping("P3M_IH4IH3")
    g, h_old, h_new, tr = remap_case(10, 8, 4, ntr=1)
    dg = DeviceGrid(g)
    with pytest.raises(Mom6HipError, match="ANSWER_DATE"):
        ALE_remap_tracers(initialize_remapping("PLM", answer_date=20181231), dg, h_old, h_new, tr)
    dg.close()
