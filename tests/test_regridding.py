"""z* regridding (ALE_regrid) and velocity remapping: the oracle (oracle/regridding.c) against the invariants of
the algorithm on the CPU; the HIP kernels against the oracle, bit for bit, on the GPU."""
import numpy as np
import pytest

from helpers import bits_equal, interior
from mom6_amd import _abi, synth
from oracle import orc


def make(ni=18, nj=14, nk=8, seed=2, terrain=False, **kw):
    g = synth.make_grid(ni, nj, nk, land_frac=0.2, seed=seed + 40, **kw)
    d = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=seed, eta_amp=0.3, terrain_following=terrain).items()}
    res = np.full(nk, 5500.0 / nk) * (0.3 + 1.4 * (np.arange(nk) + 0.5) / nk)      # stretched, sums to about 5500
    return g, d, res


@pytest.mark.parametrize("terrain", [False, True])
def test_zstar_regrid_invariants(terrain):
    g, d, res = make(terrain=terrain)
    cs = orc.regridding_cs(res)
    h_new, dz = orc.ale_regrid(g, cs, d["h"])
    h = d["h"]
    sj, si = slice(g.halo - 1, g.halo + g.nj + 1), slice(g.halo - 1, g.halo + g.ni + 1)
    ocean = g.mask2dT[sj, si] > 0
    # the column total is untouched, the surface and the bottom do not move
    assert np.allclose(h_new.sum(0)[sj, si][ocean], h.sum(0)[sj, si][ocean], rtol=1e-13)
    assert np.all(dz[0] == 0) and np.abs(dz[-1][sj, si][ocean]).max() < 1e-9
    assert h_new.min() >= 0
    # every new layer is at least MIN_THICKNESS thick (or the whole column is thinner than nk * MIN_THICKNESS)
    assert (h_new[:, sj, si][:, ocean] >= 1e-3 - 1e-9).all()        # up to roundoff of the 5 km column
    # z*: where the column is deep enough, the new thickness is the nominal one stretched by (D + eta) / D
    D = g.bathyT[sj, si]; tot = h.sum(0)[sj, si]
    k0 = 0
    expect = res[k0] * tot / np.maximum(D, 1e-30)
    deep = ocean & (D > res.sum() * 0.99)
    if deep.any():
        assert np.allclose(h_new[k0][sj, si][deep], expect[deep], rtol=1e-10)
    # regridding a z* grid again changes nothing (idempotence)
    h2, dz2 = orc.ale_regrid(g, cs, h_new)
    assert np.abs((h2 - h_new)[:, sj, si][:, ocean]).max() < 1e-8
    # land keeps its thicknesses
    assert np.array_equal(h_new[:, sj, si][:, ~ocean], h[:, sj, si][:, ~ocean])


def test_velocity_remap_conserves_transport():
    g, d, res = make()
    cs = orc.regridding_cs(res)
    h_new, dz = orc.ale_regrid(g, cs, d["h"])
    for a in (h_new,):
        orc.halo_update(g, a, _abi.POS_H)
    hou, hov = orc.ale_remap_set_h_vel(g, d["h"])
    hnu, hnv = orc.ale_remap_set_h_vel(g, h_new)
    u, v = d["u"].copy(), d["v"].copy()
    orc.ale_remap_velocities(g, "PPM_H4", hou, hov, hnu, hnv, u, v)
    mu = interior(g, g.mask2dCu, _abi.POS_U) > 0
    t0 = interior(g, (d["u"] * hou).sum(0), _abi.POS_U)[mu]; t1 = interior(g, (u * hnu).sum(0), _abi.POS_U)[mu]
    tot_o = interior(g, hou.sum(0), _abi.POS_U)[mu]; tot_n = interior(g, hnu.sum(0), _abi.POS_U)[mu]
    same = np.abs(tot_o - tot_n) < 1e-9 * tot_o           # faces whose column thickness is unchanged
    assert same.sum() > 10
    assert np.allclose(t0[same], t1[same], rtol=1e-9, atol=1e-9)
    assert np.abs(u).max() <= np.abs(d["u"]).max() * (1 + 1e-12)     # bounded remapping


def test_h_vel_via_dz_is_the_mean_of_the_new_thicknesses():
    """REMAP_UV_USING_OLD_ALG = True (.testing/tc2, tc4; MOM.F90:1666 -> ALE_remap_set_h_vel_via_dz, MOM_ALE.F90:912): the old
    thicknesses plus the interface movements are the new thicknesses (calc_h_new_by_dz), so where nothing is clipped the two
    routes to the velocity-point grid agree to roundoff; land faces are left alone by both"""
    g, d, res = make()
    h_new, dz = orc.ale_regrid(g, orc.regridding_cs(res), d["h"])
    orc.halo_update(g, h_new, _abi.POS_H)
    a_u, a_v = orc.ale_remap_set_h_vel(g, h_new)
    fill = -7.0
    b_u, b_v = orc.ale_remap_set_h_vel_via_dz(g, d["h"], dz, np.full_like(a_u, fill), np.full_like(a_v, fill))
    for a, b, pos, mk in ((a_u, b_u, _abi.POS_U, g.mask2dCu), (a_v, b_v, _abi.POS_V, g.mask2dCv)):
        m = np.broadcast_to(interior(g, mk, pos) > 0, interior(g, a, pos).shape)
        assert np.allclose(interior(g, a, pos)[m], interior(g, b, pos)[m], rtol=0, atol=1e-9 * interior(g, a, pos).max())
        assert np.all(interior(g, b, pos)[~m] == fill) and np.all(interior(g, b, pos)[m] >= 0.0)
    assert not bits_equal(a_u, b_u)          # ... but not the same bits: the switch changes answers


@pytest.mark.gpu
@pytest.mark.parametrize("space", ["device", "host"])
def test_h_vel_via_dz_matches_oracle_bitwise(space):
    import torch
    from mom6_amd.ale import ALE_remap_set_h_vel_via_dz
    from mom6_amd.tracer_advect import DeviceGrid
    for kw in (dict(), dict(nk=3, ni=70, nj=9), dict(reentrant_x=False)):
        g, d, res = make(**kw)
        h_new, dz = orc.ale_regrid(g, orc.regridding_cs(res, old_grid_weight=0.3, zs=50.0, zd=400.0), d["h"])
        w_u, w_v = orc.ale_remap_set_h_vel_via_dz(g, d["h"], dz)
        dg = DeviceGrid(g)
        T = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if space == "device" else (lambda a: np.ascontiguousarray(a).copy())
        N = lambda a: a if isinstance(a, np.ndarray) else a.cpu().numpy()
        hu, hv = T(np.zeros_like(w_u)), T(np.zeros_like(w_v))
        ALE_remap_set_h_vel_via_dz(None, dg, T(h_new), hu, hv, None, T(d["h"]), T(dz))
        dg.sync()
        assert bits_equal(N(hu), w_u) and bits_equal(N(hv), w_v)
        dg.close()


REGRID_CASES = [dict(), dict(terrain=True), dict(nk=3, ni=70, nj=9), dict(nk=40, seed=6), dict(reentrant_x=False),
                dict(old_grid_weight=0.4, zs=50.0, zd=400.0), dict(min_thickness=0.0)]


@pytest.mark.gpu
@pytest.mark.parametrize("space", ["device", "host"])
@pytest.mark.parametrize("kw", REGRID_CASES, ids=[",".join(f"{k}={v}" for k, v in c.items()) or "default" for c in REGRID_CASES])
def test_regrid_and_velocity_remap_match_oracle_bitwise(kw, space):
    import torch
    from mom6_amd.ale import (ALE_regrid, ALE_remap_set_h_vel, ALE_remap_velocities, initialize_regridding,
                              initialize_remapping)
    from mom6_amd.tracer_advect import DeviceGrid
    kw = dict(kw)
    rk = {k: kw.pop(k) for k in ("old_grid_weight", "zs", "zd", "min_thickness") if k in kw}
    if space == "host" and (kw or rk):
        pytest.skip("the staged path is covered on the default case")
    g, d, res = make(**kw)
    cs_o = orc.regridding_cs(res, **rk)
    h_new_o, dz_o = orc.ale_regrid(g, cs_o, d["h"])
    dg = DeviceGrid(g)
    T = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if space == "device" else (lambda a: np.ascontiguousarray(a).copy())
    N = lambda a: a if isinstance(a, np.ndarray) else a.cpu().numpy()
    CS = initialize_regridding(dg, coordinateResolution=res, min_thickness=rk.get("min_thickness", 1e-3),
                               old_grid_weight=rk.get("old_grid_weight", 0.0), depth_of_time_filter_shallow=rk.get("zs", 0.0),
                               depth_of_time_filter_deep=rk.get("zd", 0.0))
    h = T(d["h"]); h_new = T(np.zeros_like(d["h"])); dz = T(np.zeros_like(dz_o))
    ALE_regrid(dg, h, h_new, dz, None, CS)
    dg.sync()
    sj, si = slice(g.halo - 1, g.halo + g.nj + 1), slice(g.halo - 1, g.halo + g.ni + 1)
    assert bits_equal(N(h_new)[:, sj, si], h_new_o[:, sj, si]) and bits_equal(N(dz), dz_o)
    # velocity remap onto the new grid
    orc.halo_update(g, h_new_o, _abi.POS_H)
    hou_o, hov_o = orc.ale_remap_set_h_vel(g, d["h"]); hnu_o, hnv_o = orc.ale_remap_set_h_vel(g, h_new_o)
    u_o, v_o = d["u"].copy(), d["v"].copy()
    orc.ale_remap_velocities(g, "PPM_H4", hou_o, hov_o, hnu_o, hnv_o, u_o, v_o)
    hn = T(h_new_o)
    hou, hov, hnu, hnv = (T(np.zeros_like(a)) for a in (hou_o, hov_o, hnu_o, hnv_o))
    ALE_remap_set_h_vel(None, dg, h, hou, hov); ALE_remap_set_h_vel(None, dg, hn, hnu, hnv)
    u, v = T(d["u"]), T(d["v"])
    ALE_remap_velocities(initialize_remapping("PPM_H4"), dg, hou, hov, hnu, hnv, u, v)
    dg.sync()
    for name, a, b in (("h_old_u", hou, hou_o), ("h_new_v", hnv, hnv_o), ("u", u, u_o), ("v", v, v_o)):
        assert bits_equal(N(a), b), name
    dg.close()
