"""The OBC branches of MOM_vert_friction (src/parameterizations/vertical/MOM_vert_friction.F90 with an associated OBC: vertvisc_coef
projects the thicknesses, the depth, visc%Kv_shear and ustar outward across the faces of the open-boundary segments :1335-1355,
:1546-1566, :1901-1925, :2061-2110; vertvisc ends by storing the velocities of the specified segments :988-1006): the oracle against
what those branches state and against a quarter turn of the grid, on the CPU; the library against the oracle on the GPU, bit for bit.
(The reference holds no known-answer vectors for this module: parity unpinned.)"""
import numpy as np
import pytest

from helpers import bits_equal
from mom6_amd import _abi, synth
from mom6_amd.open_boundary import ocean_OBC_type
from oracle import orc
from rotation import rot, rot_vector, rotate_grid, unrot
from test_continuity_obc import TC3, open_faces, turned_segments
from test_vert_friction import VARIANTS

SEGS = TC3[:2] + ["I=N,J=0:N,SIMPLE", "I=0,J=N:0,FLATHER,ORLANSKI", "I=9,J=4:11,ORLANSKI", "J=7,I=15:3,SIMPLE"]


def vv_obc_case(segs, ni=22, nj=16, nk=6, seed=3, land_frac=0.1, with_ml=True):
    g = synth.make_grid(ni, nj, nk, seed=seed + 40, reentrant_x=False, reentrant_y=False, land_frac=land_frac)
    OBC = ocean_OBC_type(g, segs)
    open_faces(g, OBC)
    st = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=seed, umax=0.2).items()}
    rng = np.random.default_rng(seed)
    su, sv, sh = g.shape2(_abi.POS_U), g.shape2(_abi.POS_V), g.shape2(_abi.POS_H)
    st["u"] = np.ascontiguousarray(st["u"] + 0.05 * rng.standard_normal(st["u"].shape) * (OBC.segnum_u != 0)[None])
    st["v"] = np.ascontiguousarray(st["v"] + 0.05 * rng.standard_normal(st["v"].shape) * (OBC.segnum_v != 0)[None])
    arrs = dict(Kv_bbl_u=1.0e-3 * (0.5 + rng.random(su)), Kv_bbl_v=1.0e-3 * (0.5 + rng.random(sv)),
                bbl_thick_u=2.0 + 10.0 * rng.random(su), bbl_thick_v=2.0 + 10.0 * rng.random(sv),
                Kv_shear=1.0e-3 * rng.random((nk + 1,) + sh))
    if with_ml:
        arrs["nkml_visc_u"] = np.clip(nk * rng.random(su) ** 2, 0.0, nk); arrs["nkml_visc_v"] = np.clip(nk * rng.random(sv) ** 2, 0.0, nk)
        arrs["ustar"] = np.where(rng.random(sh) < 0.05, 0.0, 0.002 + 0.01 * rng.random(sh))
    taux = np.ascontiguousarray(0.1 * g.mask2dCu * rng.random(su))
    tauy = np.ascontiguousarray(0.05 * g.mask2dCv * rng.random(sv))
    for s in OBC.segment:
        if s.specified and s.on_pe:
            s.normal_vel[:] = 0.1 * rng.standard_normal(s.normal_vel.shape)
    return g, st, arrs, taux, tauy, OBC


def coef(g, st, arrs, OBC, dt=900.0, **kw):
    cs = orc.vertvisc_cs(g, Kv=1.0e-4, Hbbl=10.0, **kw)
    orc.vertvisc_coef(g, cs, st["u"], st["v"], st["h"], orc.vertvisc_type(**arrs), dt, OBC=OBC)
    return cs


def outside_cells(g, OBC):
    """h-point mask of the cells just outside the open-boundary faces"""
    out = np.zeros(g.shape2(_abi.POS_H), dtype=bool)
    for s in OBC.segment:
        if not s.on_pe:
            continue
        if s.is_E_or_W:      # u faces (j, I): cells I (west of the face) and I + 1; array index of u face I is I - isd + 1
            I = s.HI["IsdB"] - g.isd + 1
            js = slice(s.HI["jsd"] - g.jsd, s.HI["jed"] - g.jsd + 1)
            out[js, I if s.direction == _abi.OBC_DIRECTION_E else I - 1] = True
        else:
            J = s.HI["JsdB"] - g.jsd + 1
            is_ = slice(s.HI["isd"] - g.isd, s.HI["ied"] - g.isd + 1)
            out[J if s.direction == _abi.OBC_DIRECTION_N else J - 1, is_] = True
    return out


def test_no_segments_is_no_obc():
    g, st, arrs, taux, tauy, OBC = vv_obc_case([])
    a, b = coef(g, st, arrs, OBC, dynamic_viscous_ML=True), coef(g, st, arrs, None, dynamic_viscous_ML=True)
    for n in ("a_u", "a_v", "h_u", "h_v"):
        assert bits_equal(a._arrs[n], b._arrs[n])


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 8, 9])
def test_the_faces_of_a_segment_see_the_cell_inside_only(variant):
    """what lies outside the boundary (thickness, depth, Kv_shear, ustar) does not reach the coefficients of the segment's faces; faces off the
    segments are as without OBC; with the outside cell a copy of the inside one the two-cell means are the projections up to rounding"""
    kw = VARIANTS[variant]
    segs = TC3      # (segments along the edge: the cells outside touch no other face of the compute domain)
    g, st, arrs, taux, tauy, OBC = vv_obc_case(segs)
    a = coef(g, st, arrs, OBC, **kw)
    out = outside_cells(g, OBC)
    rng = np.random.default_rng(1)
    st2 = dict(st); arrs2 = dict(arrs)
    st2["h"] = np.where(out[None], 50.0 * rng.random(st["h"].shape), st["h"])
    arrs2["Kv_shear"] = np.where(out[None], rng.random(arrs["Kv_shear"].shape), arrs["Kv_shear"])
    arrs2["ustar"] = np.where(out, 1.0, arrs["ustar"])
    bathy = np.asarray(g.bathyT).copy()
    g.set_metric("bathyT", np.where(out, 1.0, bathy))
    b = coef(g, st2, arrs2, OBC, **kw)
    on_u, on_v = OBC.segnum_u != 0, OBC.segnum_v != 0
    assert on_u.any() and on_v.any()
    for n, on in (("a_u", on_u), ("h_u", on_u), ("a_v", on_v), ("h_v", on_v)):
        assert bits_equal(np.where(on[None], a._arrs[n], 0.0), np.where(on[None], b._arrs[n], 0.0)), n
    g.set_metric("bathyT", bathy)
    none = coef(g, st, arrs, None, **kw)
    for n, on in (("a_u", on_u), ("h_u", on_u), ("a_v", on_v), ("h_v", on_v)):
        assert bits_equal(np.where(on[None], 0.0, a._arrs[n]), np.where(on[None], 0.0, none._arrs[n])), n
        assert not bits_equal(a._arrs[n], none._arrs[n]), n


def turned_case(g, st, arrs, taux, tauy, OBC, segs):
    gr = rotate_grid(g)
    OBCr = ocean_OBC_type(gr, turned_segments(segs, g.ni, g.nj))
    for s, sr in zip(OBC.segment, OBCr.segment):
        if s.specified and s.on_pe:
            if s.is_E_or_W:
                sr.normal_vel[:] = -np.swapaxes(s.normal_vel, 1, 2)
            else:
                sr.normal_vel[:] = np.swapaxes(s.normal_vel, 1, 2)[:, ::-1, :]
    ur, vr = rot_vector(st["u"], st["v"])
    str_ = dict(u=ur, v=vr, h=rot(st["h"]))
    ar = {}
    for n, a in arrs.items():      # face scalars change places, cell fields turn
        if n.endswith("_u"):
            ar[n[:-2] + "_v"] = rot(a)
        elif n.endswith("_v"):
            ar[n[:-2] + "_u"] = rot(a)
        else:
            ar[n] = rot(a)
    tx, ty = rot_vector(taux, tauy)
    return gr, str_, ar, tx, ty, OBCr


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 6, 8, 10])
def test_oracle_turns_with_the_grid(variant):
    """the reference writes the u and v branches (and E/W, N/S) out separately: a quarter turn of the grid, the state and the segments gives
    the turned coefficients and velocities, to the bit"""
    kw = VARIANTS[variant]
    g, st, arrs, taux, tauy, OBC = vv_obc_case(SEGS)
    dt = 900.0
    cs = coef(g, st, arrs, OBC, dt, **kw)
    u, v = st["u"].copy(), st["v"].copy()
    orc.vertvisc(g, cs, u, v, st["h"], taux, tauy, orc.vertvisc_type(**arrs), dt, OBC=OBC)
    gr, str_, ar, tx, ty, OBCr = turned_case(g, st, arrs, taux, tauy, OBC, SEGS)
    csr = coef(gr, str_, ar, OBCr, dt, **kw)
    ur, vr = str_["u"].copy(), str_["v"].copy()
    orc.vertvisc(gr, csr, ur, vr, str_["h"], tx, ty, orc.vertvisc_type(**ar), dt, OBC=OBCr)
    # a_u' = rot(a_v), a_v' = rot(a_u) over the compute faces
    cu = (slice(None), slice(g.jsc - g.jsd, g.jec - g.jsd + 1), slice(g.isc - g.isd, g.iec - g.isd + 2))
    cv = (slice(None), slice(g.jsc - g.jsd, g.jec - g.jsd + 2), slice(g.isc - g.isd, g.iec - g.isd + 1))
    for n in ("a", "h"):
        assert bits_equal(unrot(csr._arrs[n + "_v"])[cu], cs._arrs[n + "_u"][cu]), n
        assert bits_equal(unrot(csr._arrs[n + "_u"])[cv], cs._arrs[n + "_v"][cv]), n
    assert np.array_equal((-unrot(vr))[cu], u[cu]) and np.array_equal(unrot(ur)[cv], v[cv])


def test_vertvisc_stores_the_velocities_of_the_specified_segments():
    g, st, arrs, taux, tauy, OBC = vv_obc_case(SEGS)
    dt = 900.0
    cs = coef(g, st, arrs, OBC, dt)
    u, v = st["u"].copy(), st["v"].copy()
    orc.vertvisc(g, cs, u, v, st["h"], taux, tauy, orc.vertvisc_type(**arrs), dt, OBC=OBC)
    u0, v0 = st["u"].copy(), st["v"].copy()
    orc.vertvisc(g, cs, u0, v0, st["h"], taux, tauy, orc.vertvisc_type(**arrs), dt)
    spec_u, spec_v = np.zeros_like(OBC.segnum_u, dtype=bool), np.zeros_like(OBC.segnum_v, dtype=bool)
    n_spec = 0
    for s in OBC.segment:
        if not (s.specified and s.on_pe):
            continue
        n_spec += 1
        if s.is_E_or_W:
            I = s.HI["IsdB"] - g.isd + 1; js = slice(s.HI["jsd"] - g.jsd, s.HI["jed"] - g.jsd + 1)
            spec_u[js, I] = True
            assert bits_equal(u[:, js, I], s.normal_vel[:, :, 0])
        else:
            J = s.HI["JsdB"] - g.jsd + 1; is_ = slice(s.HI["isd"] - g.isd, s.HI["ied"] - g.isd + 1)
            spec_v[J, is_] = True
            assert bits_equal(v[:, J, is_], s.normal_vel[:, 0, :])
    assert n_spec == 2
    assert bits_equal(np.where(spec_u[None], 0.0, u), np.where(spec_u[None], 0.0, u0))
    assert bits_equal(np.where(spec_v[None], 0.0, v), np.where(spec_v[None], 0.0, v0))


@pytest.mark.gpu
@pytest.mark.parametrize("variant", range(len(VARIANTS)))
@pytest.mark.parametrize("space", ["device", "host"])
def test_gpu_vertvisc_with_open_boundaries_matches_oracle_bitwise(variant, space):
    import torch
    from mom6_amd.tracer_advect import DeviceGrid
    from mom6_amd.vert_friction import vertvisc, vertvisc_coef, vertvisc_init, vertvisc_ntrunc, vertvisc_type
    kw = VARIANTS[variant]
    pk = dict(KV=1.0e-4, HBBL=10.0)
    names = dict(harmonic_visc="HARMONIC_VISC", harm_BL_val="HARMONIC_BL_SCALE", Kvml_invZ2="KV_ML_INVZ2", Hmix="HMIX_FIXED",
                 bottomdraglaw="BOTTOMDRAGLAW", Kv_extra_bbl="KV_EXTRA_BBL", direct_stress="DIRECT_STRESS",
                 CFL_based_trunc="CFL_BASED_TRUNCATIONS", maxvel="MAXVEL", vel_underflow="VEL_UNDERFLOW",
                 dynamic_viscous_ML="DYNAMIC_VISCOUS_ML", nkml="NKML")
    pk.update({names[k]: v for k, v in kw.items()})
    dt = 900.0
    for (ni, nj, nk) in [(22, 16, 6), (70, 20, 3)]:
        g, st, arrs, taux, tauy, OBC = vv_obc_case(SEGS, ni=ni, nj=nj, nk=nk, seed=ni)
        rcs = coef(g, st, arrs, OBC, dt, **kw)
        ru, rv = st["u"].copy(), st["v"].copy()
        rtbx, rtby = g.zeros2(_abi.POS_U), g.zeros2(_abi.POS_V)
        orc.vertvisc(g, rcs, ru, rv, st["h"], taux, tauy, orc.vertvisc_type(**arrs), dt, rtbx, rtby, OBC=OBC)
        none = coef(g, st, arrs, None, dt, **kw)
        assert not bits_equal(rcs._arrs["a_u"], none._arrs["a_u"])
        dg = DeviceGrid(g)
        resident = space == "device"
        X = (lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a).copy()).cuda()) if resident else \
            (lambda a: None if a is None else np.ascontiguousarray(a).copy())
        N = (lambda a: a.cpu().numpy()) if resident else (lambda a: a)
        if resident:      # the segments' own arrays in the memory space of the call
            for s in OBC.segment:
                for k in ("normal_vel", "normal_trans", "nudged_normal_vel", "tangential_vel", "tangential_grad"):
                    if s.on_pe and isinstance(getattr(s, k, None), np.ndarray):
                        setattr(s, k, X(getattr(s, k)))
        CS = vertvisc_init(dg, device_arrays=resident, **pk)
        visc = vertvisc_type(**{n: X(a) for n, a in arrs.items()})
        u, v, h = X(st["u"]), X(st["v"]), X(st["h"])
        vertvisc_coef(u, v, h, None, None, visc, None, dt, dg, CS, OBC=OBC)
        what = (variant, (ni, nj, nk), space)
        for n in ("a_u", "a_v", "h_u", "h_v"):
            assert bits_equal(rcs._arrs[n], N(CS.arrays[n])), (what, n, np.argwhere(rcs._arrs[n] != N(CS.arrays[n]))[:3])
        tbx, tby = X(g.zeros2(_abi.POS_U)), X(g.zeros2(_abi.POS_V))
        vertvisc(u, v, h, (X(taux), X(tauy)), visc, dt, OBC, None, None, dg, CS, tbx, tby)
        assert bits_equal(ru, N(u)), (what, "u", np.argwhere(ru != N(u))[:3])
        assert bits_equal(rv, N(v)), (what, "v", np.argwhere(rv != N(v))[:3])
        assert bits_equal(rtbx, N(tbx)) and bits_equal(rtby, N(tby)), (what, "tau_bot")
        assert vertvisc_ntrunc(dg, CS) == rcs.ntrunc, (what, "ntrunc")
        dg.close()
