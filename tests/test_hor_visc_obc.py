"""The OBC branches of horizontal_viscosity (src/parameterizations/lateral/MOM_hor_visc.F90 with an associated OBC on the PE): the strains at
the corner points of the segments :733-790 (OBC_ZERO_STRAIN, OBC_FREESLIP_STRAIN, OBC_COMPUTED_STRAIN), the thicknesses at and beside their
faces :791-849, OBC_ZERO_BIHARMONIC :889-903, the gradient of the Laplacian :1388-1409, the accelerations of the segments' own faces
:1751-1782.  The oracle against what those branches state and against a quarter turn of the grid, on the CPU; the library against the oracle
on the GPU, bit for bit.  (The reference holds no known-answer vectors for this module: parity unpinned.)"""
import numpy as np
import pytest

from helpers import bits_equal, interior
from mom6_amd import _abi, synth
from mom6_amd.open_boundary import ocean_OBC_type
from oracle import orc
from rotation import rot, rot_vector, rotate_grid, unrot_vector
from test_continuity_obc import TC3, open_faces, turned_segments
from test_hor_visc import DT, REF_NAMES, VARIANTS

U, V = _abi.POS_U, _abi.POS_V
SEGS = TC3 + ["I=9,J=4:11,ORLANSKI", "J=7,I=15:3,SIMPLE", "J=8,I=12:6,GRADIENT"]      # (the last two lie side by side: a chain of projections)
FLAGS = [dict(freeslip_strain=True, zero_biharmonic=True),      # .testing/tc3
         dict(zero_strain=True), dict(computed_strain=True, zero_biharmonic=True), dict(zero_biharmonic=True), dict()]
# (the reference projects the thicknesses across the corner points of a segment over IsdB:IedB for a northern segment and over isd:ied for the
# other three, :826-844: the ends of a segment that stops inside the domain do not turn with the grid; these segments span it)
SEGS_SPAN = TC3 + ["I=9,J=0:N,ORLANSKI", "J=7,I=N:0,SIMPLE", "J=8,I=0:N,GRADIENT"]
NAMES = ["smag_ah", "lap_plus_biharm", "legacy_bounds", "laplacian_noslip", "cont_thickness"]


def hv_obc_case(segs, flags, ni=22, nj=16, nk=3, seed=5):
    g = synth.make_grid(ni, nj, nk, seed=seed + 40, reentrant_x=False, reentrant_y=False, land_frac=0.1)
    OBC = ocean_OBC_type(g, segs, **flags)
    open_faces(g, OBC)
    d = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=seed, umax=0.3).items()}
    rng = np.random.default_rng(seed)
    d["u"] = np.ascontiguousarray(d["u"] + 0.05 * rng.standard_normal(d["u"].shape) * (OBC.segnum_u != 0)[None])
    d["v"] = np.ascontiguousarray(d["v"] + 0.05 * rng.standard_normal(d["v"].shape) * (OBC.segnum_v != 0)[None])
    for s in OBC.segment:
        if s.on_pe:
            s.tangential_vel[:] = 0.1 * rng.standard_normal(s.tangential_vel.shape)
    hu = np.zeros_like(d["u"]); hu[:, :, 1:-1] = 0.5 * (d["h"][:, :, :-1] + d["h"][:, :, 1:]) * 1.01
    hv = np.zeros_like(d["v"]); hv[:, 1:-1, :] = 0.5 * (d["h"][:, :-1, :] + d["h"][:, 1:, :]) * 0.99
    d["hu_cont"], d["hv_cont"] = hu, hv
    return g, d, OBC


def run_oracle(g, d, OBC, kw):
    cont = dict(hu_cont=d["hu_cont"], hv_cont=d["hv_cont"]) if kw.get("use_cont_thick") else {}
    return orc.horizontal_viscosity(g, orc.hor_visc_cs(g, DT, **kw), d["u"], d["v"], d["h"], DT, OBC=OBC, **cont)


def test_no_segments_is_no_obc():
    g, d, OBC = hv_obc_case([], FLAGS[0])
    a, b = run_oracle(g, d, OBC, VARIANTS["smag_ah"]), run_oracle(g, d, None, VARIANTS["smag_ah"])
    assert bits_equal(a[0], b[0]) and bits_equal(a[1], b[1])


@pytest.mark.parametrize("flags", FLAGS, ids=[",".join(f) or "none" for f in FLAGS])
def test_the_segments_own_faces_feel_no_viscosity_and_the_rest_changes_nearby_only(flags):
    g, d, OBC = hv_obc_case(SEGS, flags, ni=36, nj=28)
    a = run_oracle(g, d, OBC, VARIANTS["lap_plus_biharm"])
    b = run_oracle(g, d, None, VARIANTS["lap_plus_biharm"])
    on_u, on_v = OBC.segnum_u != 0, OBC.segnum_v != 0
    assert np.all(a[0][:, on_u] == 0.0) and np.all(a[1][:, on_v] == 0.0)
    near = np.zeros(g.shape2(_abi.POS_H), dtype=bool)
    near |= on_u[:, 1:] | on_u[:, :-1] | on_v[1:, :] | on_v[:-1, :]
    for _ in range(4):
        near[1:, :] |= near[:-1, :].copy(); near[:-1, :] |= near[1:, :].copy(); near[:, 1:] |= near[:, :-1].copy(); near[:, :-1] |= near[:, 1:].copy()
    far = ~near
    assert far.any()
    assert bits_equal(np.where(far[None], a[0][:, :, 1:], 0.0), np.where(far[None], b[0][:, :, 1:], 0.0))
    assert bits_equal(np.where(far[None], a[1][:, 1:, :], 0.0), np.where(far[None], b[1][:, 1:, :], 0.0))
    assert not bits_equal(a[0], b[0])


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("flags", FLAGS[:3], ids=[",".join(f) for f in FLAGS[:3]])
def test_oracle_turns_with_the_grid(name, flags):
    kw = VARIANTS[name]
    g, d, OBC = hv_obc_case(SEGS_SPAN, flags)
    du, dv = run_oracle(g, d, OBC, kw)
    gr = rotate_grid(g)
    OBCr = ocean_OBC_type(gr, turned_segments(SEGS_SPAN, g.ni, g.nj), **flags)
    for s, sr in zip(OBC.segment, OBCr.segment):
        if not s.on_pe:
            continue
        tv = np.swapaxes(s.tangential_vel, 1, 2)[:, ::-1, :]
        sr.tangential_vel[:] = tv if s.is_E_or_W else -tv      # v -> u' = v ; u -> v' = -u (tests/test_coriolis_obc.py)
    ur, vr = rot_vector(d["u"], d["v"])
    dr = dict(u=ur, v=vr, h=rot(d["h"]), hu_cont=rot(d["hv_cont"]), hv_cont=rot(d["hu_cont"]))
    dur, dvr = run_oracle(gr, dr, OBCr, kw)
    bu, bv = unrot_vector(dur, dvr)
    assert np.array_equal(interior(g, bu, U), interior(g, du, U)) and np.array_equal(interior(g, bv, V), interior(g, dv, V))
    assert np.abs(interior(g, du, U)).max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(VARIANTS))
@pytest.mark.parametrize("flags", FLAGS, ids=[",".join(f) or "none" for f in FLAGS])
@pytest.mark.parametrize("space", ["device", "host"])
def test_gpu_horizontal_viscosity_with_open_boundaries_matches_oracle_bitwise(name, flags, space):
    import torch
    from mom6_amd.hor_visc import hor_visc_init, horizontal_viscosity
    from mom6_amd.tracer_advect import DeviceGrid
    kw = VARIANTS[name]
    for (ni, nj, nk) in [(22, 16, 3), (150, 40, 2)]:
        g, d, OBC = hv_obc_case(SEGS, flags, ni=ni, nj=nj, nk=nk, seed=ni)
        ref = run_oracle(g, d, OBC, kw)
        dg = DeviceGrid(g)
        resident = space == "device"
        if resident:
            OBC.cuda()      # (segment%tangential_vel, read with OBC_COMPUTED_STRAIN, in the memory space of the fields)
        CS = hor_visc_init(dg, DT, device_arrays=resident, **{REF_NAMES[k]: v for k, v in kw.items()})
        X = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if resident else (lambda a: a.copy())
        N = (lambda a: a.cpu().numpy()) if resident else (lambda a: a)
        du, dv = X(np.zeros_like(d["u"])), X(np.zeros_like(d["v"]))
        cont = dict(hu_cont=X(d["hu_cont"]), hv_cont=X(d["hv_cont"])) if kw.get("use_cont_thick") else {}
        horizontal_viscosity(X(d["u"]), X(d["v"]), X(d["h"]), du, dv, None, None, dg, CS, OBC=OBC, **cont)
        dg.sync()
        assert bits_equal(N(du), ref[0]), (name, flags, (ni, nj, nk), "diffu", np.argwhere(N(du) != ref[0])[:3])
        assert bits_equal(N(dv), ref[1]), (name, flags, (ni, nj, nk), "diffv", np.argwhere(N(dv) != ref[1])[:3])
        dg.close()
