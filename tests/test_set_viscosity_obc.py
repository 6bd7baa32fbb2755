"""The OBC branches of set_viscous_BBL (src/parameterizations/vertical/MOM_set_viscosity.F90 with CS%OBC associated): the one-sided
depths and the masks of the faces at and beside the segments :374-413, the zero-gradient projection of the thicknesses, T and S across the
segments' faces :502-580, the weights of set_v_at_u / set_u_at_v :1829-1838, :1874-1883.  The oracle against what those branches state and
against a quarter turn of the grid, on the CPU; the library against the oracle on the GPU, bit for bit.  (The reference holds no
known-answer vectors for this module: parity unpinned.)"""
import numpy as np
import pytest

from helpers import bits_equal, interior
from mom6_amd import _abi, synth
from mom6_amd.open_boundary import ocean_OBC_type
from oracle import orc
from rotation import rot, rot_vector, rotate_grid, unrot
from test_continuity_obc import TC3, open_faces, turned_segments
from test_set_viscosity import REF, VARIANTS, rlay, visc_arrays
from test_vert_friction_obc import outside_cells

H, U, V = _abi.POS_H, _abi.POS_U, _abi.POS_V
SEGS = TC3[:2] + ["I=N,J=0:N,SIMPLE", "I=0,J=N:0,FLATHER,ORLANSKI", "I=9,J=4:11,ORLANSKI", "J=7,I=15:3,SIMPLE"]
NAMES = ["default", "rlay", "bg_vel", "body_force", "channel_tc", "channel_iterative", "channel_bounds"]


def bbl_obc_case(segs, ni=22, nj=16, nk=6, seed=3):
    g = synth.make_grid(ni, nj, nk, seed=seed + 40, reentrant_x=False, reentrant_y=False, land_frac=0.1)
    OBC = ocean_OBC_type(g, segs)
    open_faces(g, OBC)
    d = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=seed, umax=0.3).items()}
    rng = np.random.default_rng(seed)
    d["u"] = np.ascontiguousarray(d["u"] + 0.05 * rng.standard_normal(d["u"].shape) * (OBC.segnum_u != 0)[None])
    d["v"] = np.ascontiguousarray(d["v"] + 0.05 * rng.standard_normal(d["v"].shape) * (OBC.segnum_v != 0)[None])
    return g, d, OBC


def run_oracle(g, d, OBC, **kw):
    arrs = visc_arrays(g, d)
    visc = orc.vertvisc_type(**arrs)
    if not kw.get("BBL_use_EOS", True):
        kw = dict(kw, Rlay=rlay(g.nk))
    cs = orc.set_visc_cs(g, 10.0, 1.0e-4, **kw)
    orc.set_viscous_BBL(g, cs, d["u"], d["v"], d["h"], d["T"], d["S"], orc.eos("WRIGHT"), visc, OBC=OBC)
    return visc._keep


OUT = ("bbl_thick_u", "bbl_thick_v", "Kv_bbl_u", "Kv_bbl_v", "Ray_u", "Ray_v")


def test_no_segments_is_no_obc():
    g, d, OBC = bbl_obc_case([])
    a, b = run_oracle(g, d, OBC, **VARIANTS["channel_tc"]), run_oracle(g, d, None, **VARIANTS["channel_tc"])
    for n in OUT:
        assert bits_equal(a[n], b[n]), n


@pytest.mark.parametrize("name", ["default", "rlay", "body_force"])
def test_the_faces_of_a_segment_see_the_cell_inside_only(name):
    """without CHANNEL_DRAG the bottom boundary layer at a face of a segment along the edge of the domain does not depend on the thicknesses, T
    and S of the cell outside; away from the segments nothing changes"""
    kw = VARIANTS[name]
    g, d, OBC = bbl_obc_case(TC3)
    a = run_oracle(g, d, OBC, **kw)
    out = outside_cells(g, OBC)
    rng = np.random.default_rng(1)
    d2 = dict(d)
    d2["h"] = np.where(out[None], 50.0 * rng.random(d["h"].shape), d["h"])
    d2["T"] = np.where(out[None], 30.0 * rng.random(d["h"].shape), d["T"]); d2["S"] = np.where(out[None], 40.0 * rng.random(d["h"].shape), d["S"])
    b = run_oracle(g, d2, OBC, **kw)
    on_u, on_v = OBC.segnum_u != 0, OBC.segnum_v != 0
    for n in OUT:
        on = on_u if n.endswith("_u") else on_v
        on = on if a[n].ndim == 2 else on[None]
        assert bits_equal(np.where(on, a[n], 0.0), np.where(on, b[n], 0.0)), n
    none = run_oracle(g, d, None, **kw)
    near = np.zeros(g.shape2(H), dtype=bool)
    near |= on_u[:, 1:] | on_u[:, :-1] | on_v[1:, :] | on_v[:-1, :]
    for _ in range(2):
        near[1:, :] |= near[:-1, :].copy(); near[:-1, :] |= near[1:, :].copy(); near[:, 1:] |= near[:, :-1].copy(); near[:, :-1] |= near[:, 1:].copy()
    far = ~near
    assert not all(bits_equal(a[n], none[n]) for n in OUT)
    for n in OUT[:4]:
        sl = (slice(None), slice(1, None)) if n.endswith("_u") else (slice(1, None), slice(None))
        assert bits_equal(np.where(far, a[n][sl], 0.0), np.where(far, none[n][sl], 0.0)), n


@pytest.mark.parametrize("name", NAMES)
def test_oracle_turns_with_the_grid(name):
    """the reference writes the u and v halves (and E/W, N/S) out separately: a quarter turn of the grid, the state and the segments gives the
    turned boundary layer, to the bit"""
    kw = VARIANTS[name]
    g, d, OBC = bbl_obc_case(SEGS)
    a = run_oracle(g, d, OBC, **kw)
    gr = rotate_grid(g)
    OBCr = ocean_OBC_type(gr, turned_segments(SEGS, g.ni, g.nj))
    ur, vr = rot_vector(d["u"], d["v"])
    dr = dict(u=ur, v=vr, h=rot(d["h"]), T=rot(d["T"]), S=rot(d["S"]))
    b = run_oracle(gr, dr, OBCr, **kw)
    for n in OUT:
        other = n[:-1] + ("v" if n.endswith("u") else "u")
        pos = U if n.endswith("u") else V
        assert bits_equal(interior(g, unrot(b[other]), pos), interior(g, a[n], pos)), n


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(VARIANTS))
@pytest.mark.parametrize("space", ["device", "host"])
def test_gpu_set_viscous_BBL_with_open_boundaries_matches_oracle_bitwise(name, space):
    import torch
    from mom6_amd.pressure_force import EOS_init
    from mom6_amd.set_viscosity import set_visc_init, set_viscous_BBL
    from mom6_amd.tracer_advect import DeviceGrid
    from mom6_amd.vert_friction import vertvisc_type
    kw = VARIANTS[name]
    for (ni, nj, nk) in [(22, 16, 6), (70, 20, 3)]:
        g, d, OBC = bbl_obc_case(SEGS, ni=ni, nj=nj, nk=nk, seed=ni)
        ref = run_oracle(g, d, OBC, **kw)
        none = run_oracle(g, d, None, **kw)
        assert not bits_equal(ref["bbl_thick_u"], none["bbl_thick_u"])
        dg = DeviceGrid(g)
        pk = {REF[k]: v for k, v in kw.items()}
        if not kw.get("BBL_use_EOS", True):
            pk["Rlay"] = rlay(nk)
        CS = set_visc_init(dg, HBBL=10.0, KV=1.0e-4, OBC=OBC, **pk)
        resident = space == "device"
        X = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if resident else (lambda a: a.copy())
        N = (lambda a: a.cpu().numpy()) if resident else (lambda a: a)
        arrs = {n: X(a) for n, a in visc_arrays(g, d).items()}
        visc = vertvisc_type(**arrs)
        set_viscous_BBL(X(d["u"]), X(d["v"]), X(d["h"]), (X(d["T"]), X(d["S"]), EOS_init("WRIGHT")), visc, dg, CS)
        dg.sync()
        for n in OUT:
            assert bits_equal(N(arrs[n]), ref[n]), (name, (ni, nj, nk), space, n, np.argwhere(N(arrs[n]) != ref[n])[:3])
        dg.close()
