"""The OBC branches of continuity_PPM (src/core/MOM_continuity_PPM.F90 with an associated OBC: PPM_reconstruction_x/y :2385-2432, flux_layer
:956-971, the mass-flux blocks :629-634, :722-779, :782-805, flux_thickness :1058-1088) and the placement of segments from their
MOM_input strings (mom6_amd/open_boundary.py after MOM_open_boundary.F90:1211-1610).  The oracle against what those branches state,
on the CPU; the library against the oracle on the GPU, bit for bit.  (The reference holds no known-answer vectors: parity unpinned.)"""
import numpy as np
import pytest

from helpers import bits_equal, interior
from mom6_amd import _abi, synth
from mom6_amd.open_boundary import ocean_OBC_type, parse_segment_str
from oracle import orc
from rotation import rot, rot_vector, rotate_grid, unrot, unrot_vector

TC3 = ["J=N,I=N:0,FLATHER,ORLANSKI", "J=0,I=0:N,FLATHER,ORLANSKI", "I=N,J=0:N,FLATHER,ORLANSKI", "I=0,J=N:0,FLATHER,ORLANSKI"]      # .testing/tc3


def open_faces(g, OBC):
    """a regional grid's edge faces are walls in the synthetic grid: open the faces of the segments (face length and mask), as the
    reference's grid has them with OBCs along the edge of the domain"""
    m = dict(g.metrics)
    for n in ("mask2dCu", "dy_Cu", "mask2dCv", "dx_Cv"):
        m[n] = m[n].copy()
    su, sv = OBC.segnum_u != 0, OBC.segnum_v != 0
    m["mask2dCu"][su] = 1.0; m["dy_Cu"][su] = m["dyCu"][su]
    m["mask2dCv"][sv] = 1.0; m["dx_Cv"][sv] = m["dxCv"][sv]
    for n in ("mask2dCu", "dy_Cu", "mask2dCv", "dx_Cv"):
        g.set_metric(n, m[n])


def obc_case(segs, ni=22, nj=16, nk=4, seed=4, first_direction=0, land_frac=0.1, specified_data=True):
    g = synth.make_grid(ni, nj, nk, seed=seed + 30, first_direction=first_direction, reentrant_x=False, reentrant_y=False, land_frac=land_frac)
    OBC = ocean_OBC_type(g, segs)
    open_faces(g, OBC)
    st = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=seed).items()}
    rng = np.random.default_rng(seed)
    # flow through the boundary, thicknesses outside it that differ from the inside (the OBC branches must not read them)
    st["u"] = np.ascontiguousarray(st["u"] + 0.05 * rng.standard_normal(st["u"].shape) * (OBC.segnum_u != 0)[None])
    st["v"] = np.ascontiguousarray(st["v"] + 0.05 * rng.standard_normal(st["v"].shape) * (OBC.segnum_v != 0)[None])
    kk = (np.arange(nk) + 0.5) / nk
    st["visc_rem_u"] = np.ascontiguousarray(np.clip(1.0 - 0.8 * kk[:, None, None] ** 4 + 0.0 * st["u"], 0.05, 1.0) * (0.9 + 0.1 * rng.random(st["u"].shape)))
    st["visc_rem_v"] = np.ascontiguousarray(np.clip(1.0 - 0.8 * kk[:, None, None] ** 4 + 0.0 * st["v"], 0.05, 1.0) * (0.9 + 0.1 * rng.random(st["v"].shape)))
    if specified_data:
        for s in OBC.segment:
            if s.specified and s.on_pe:
                s.normal_vel[:] = 0.1 * rng.standard_normal(s.normal_vel.shape)
                s.normal_vel[rng.random(s.normal_vel.shape) < 0.2] = 0.0      # (the face area skips layers at rest :766)
                s.normal_trans[:] = s.normal_vel * (3.0e4 * (5.0 + 50.0 * rng.random(s.normal_vel.shape)))
    return g, st, OBC


def run(g, st, OBC, dt=900.0, uhbt=True, bt=True, **cskw):
    cs = orc.continuity_cs(g.nk, g.Angstrom_H, **cskw)
    out = dict(h=st["h"].copy(), uh=np.zeros_like(st["u"]), vh=np.zeros_like(st["v"]))
    kw = dict(visc_rem_u=st["visc_rem_u"], visc_rem_v=st["visc_rem_v"])
    if uhbt:
        uh0, vh0 = np.zeros_like(st["u"]), np.zeros_like(st["v"])
        orc.continuity(g, cs, st["u"], st["v"], st["h"].copy(), st["h"].copy(), uh0, vh0, dt, OBC=OBC, **kw)
        out["uhbt"] = np.ascontiguousarray(uh0.sum(0) * 1.05 + 0.02 * np.abs(uh0).sum(0))
        out["vhbt"] = np.ascontiguousarray(vh0.sum(0) * 0.95 - 0.02 * np.abs(vh0).sum(0))
        out.update(u_cor=np.zeros_like(st["u"]), v_cor=np.zeros_like(st["v"]), du_cor=g.zeros2(_abi.POS_U), dv_cor=g.zeros2(_abi.POS_V))
        kw.update(uhbt=out["uhbt"], vhbt=out["vhbt"], u_cor=out["u_cor"], v_cor=out["v_cor"], du_cor=out["du_cor"], dv_cor=out["dv_cor"])
    if bt:
        arrs, btst = orc.make_bt_cont(g, with_h=True)
        out["bt"] = arrs; kw["bt_cont"] = btst
    orc.continuity(g, cs, st["u"], st["v"], st["h"].copy(), out["h"], out["uh"], out["vh"], dt, OBC=OBC, **kw)
    return out


def test_segment_strings_are_placed_as_the_reference_places_them():
    g = synth.make_grid(10, 8, 2, reentrant_x=False, reentrant_y=False)
    assert parse_segment_str(10, 8, "I=N-2,J=3:N+1,ORLANSKI,NUDGED") == (8, 3, 9, ["ORLANSKI", "NUDGED"])
    OBC = ocean_OBC_type(g, TC3)
    d = [s.direction for s in OBC.segment]
    assert d == [_abi.OBC_DIRECTION_N, _abi.OBC_DIRECTION_S, _abi.OBC_DIRECTION_E, _abi.OBC_DIRECTION_W]
    assert all(s.open and s.Flather and s.radiation and s.on_pe for s in OBC.segment)
    assert OBC.open_u_BCs_exist_globally and OBC.Flather_v_BCs_exist_globally and not OBC.specified_u_BCs_exist_globally
    h = g.halo
    # the four sides of the 10 x 8 domain: v faces of the rows J = 0 and N, u faces of the columns I = 0 and N, over the cells 1 .. N
    assert (OBC.segnum_v[h + 8, h:h + 10] == 1).all() and (OBC.segnum_v[h, h:h + 10] == 2).all() and (OBC.segnum_v != 0).sum() == 20
    assert (OBC.segnum_u[h:h + 8, h + 10] == 3).all() and (OBC.segnum_u[h:h + 8, h] == 4).all() and (OBC.segnum_u != 0).sum() == 16
    N, S, E, W = OBC.segment
    assert (E.HI["IsdB"], E.HI["jsd"], E.HI["jed"]) == (h + 10, h + 1, h + 8) and (W.HI["IsdB"], W.HI["isd"]) == (h, h + 1)
    assert (N.HI["JsdB"], N.HI["isd"], N.HI["ied"]) == (h + 8, h + 1, h + 10) and (S.HI["JsdB"], S.HI["jsd"]) == (h, h + 1)
    with pytest.raises(Exception, match="cannot be used together"):
        ocean_OBC_type(g, ["I=N,J=0:N,ORLANSKI,OBLIQUE"])
    with pytest.raises(Exception, match="not understood"):
        ocean_OBC_type(g, ["I=N,J=0:N,SOMETHING"])


@pytest.mark.parametrize("first_direction", [0, 1])
def test_no_segments_is_no_obc(first_direction):
    g, st, OBC = obc_case([], first_direction=first_direction)
    a, b = run(g, st, OBC), run(g, st, None)
    for n in ("h", "uh", "vh", "u_cor", "v_cor", "du_cor", "dv_cor"):
        assert bits_equal(a[n], b[n]), n
    for n in a["bt"]:
        assert bits_equal(a["bt"][n], b["bt"][n]), n


@pytest.mark.parametrize("first_direction", [0, 1])
def test_open_faces_carry_the_interior_thickness(first_direction):
    segs = TC3 + ["I=9,J=4:11,ORLANSKI", "J=7,I=15:3,GRADIENT"]      # the four sides and two segments inside the domain
    g, st, OBC = obc_case(segs, first_direction=first_direction)
    o = run(g, st, OBC, uhbt=False)
    x_first = first_direction % 2 == 0
    hsrc = {True: st["h"], False: o["h"]}      # (the second direction sees the thicknesses the first one left; checked on faces of the first)
    for seg in OBC.segment:
        H = seg.HI
        if seg.is_E_or_W and x_first:
            I = H["IsdB"]; jj = slice(H["jsd"] - g.jsd, H["jed"] - g.jsd + 1)
            ci = (I - g.isd) if seg.direction == _abi.OBC_DIRECTION_E else (I + 1 - g.isd)
            hi = st["h"][:, jj, ci]; fI = I - (g.isd - 1)
            dy = g.dy_Cu[jj, fI]
            assert bits_equal(o["uh"][:, jj, fI], (dy * 1.0) * st["u"][:, jj, fI] * hi)
            assert bits_equal(o["bt"]["h_u"][:, jj, fI], hi * (st["visc_rem_u"][:, jj, fI] * 1.0))
            FA = np.zeros_like(dy)
            for k in range(g.nk):
                FA = FA + hi[k] * (dy * 1.0)
            for n in ("FA_u_W0", "FA_u_WW", "FA_u_E0", "FA_u_EE"):
                assert bits_equal(o["bt"][n][jj, fI], FA), n
            assert not o["bt"]["uBT_WW"][jj, fI].any() and not o["bt"]["uBT_EE"][jj, fI].any()
        if seg.is_N_or_S and not x_first:
            J = H["JsdB"]; ii = slice(H["isd"] - g.isd, H["ied"] - g.isd + 1)
            cj = (J - g.jsd) if seg.direction == _abi.OBC_DIRECTION_N else (J + 1 - g.jsd)
            hi = st["h"][:, cj, ii]; fJ = J - (g.jsd - 1)
            dx = g.dx_Cv[fJ, ii]
            assert bits_equal(o["vh"][:, fJ, ii], (dx * 1.0) * st["v"][:, fJ, ii] * hi)
            assert bits_equal(o["bt"]["h_v"][:, fJ, ii], hi * (st["visc_rem_v"][:, fJ, ii] * 1.0))


@pytest.mark.parametrize("first_direction", [0, 1])
def test_segments_act_locally(first_direction):
    """away from every segment nothing changes (the reconstruction's stencil, then the other direction's, then the convergence)"""
    g, st, OBC = obc_case(["I=0,J=N:0,FLATHER,ORLANSKI", "J=0,I=0:12,ORLANSKI"], ni=40, nj=30, first_direction=first_direction)
    o, n = run(g, st, OBC), run(g, st, None)
    near = np.zeros(g.shape2(_abi.POS_H), dtype=bool)
    near |= (OBC.segnum_u[:, 1:] != 0) | (OBC.segnum_u[:, :-1] != 0) | (OBC.segnum_v[1:, :] != 0) | (OBC.segnum_v[:-1, :] != 0)
    for _ in range(6):
        near[1:, :] |= near[:-1, :].copy(); near[:-1, :] |= near[1:, :].copy(); near[:, 1:] |= near[:, :-1].copy(); near[:, :-1] |= near[:, 1:].copy()
    far = ~near
    assert far[g.csl(_abi.POS_H)].sum() > 300
    assert bits_equal(np.where(far[None], o["h"], 0.0), np.where(far[None], n["h"], 0.0))
    assert bits_equal(np.where(far[None], o["uh"][:, :, 1:], 0.0), np.where(far[None], n["uh"][:, :, 1:], 0.0))
    assert not bits_equal(interior(g, o["h"]), interior(g, n["h"]))


@pytest.mark.parametrize("first_direction", [0, 1])
def test_specified_faces_take_the_external_transports(first_direction):
    segs = ["I=N,J=0:N,SIMPLE", "J=0,I=0:N,SIMPLE", "I=0,J=N:0,FLATHER,ORLANSKI"]
    g, st, OBC = obc_case(segs, first_direction=first_direction)
    o = run(g, st, OBC)
    E, S = OBC.segment[0], OBC.segment[1]
    fI = E.HI["IsdB"] - (g.isd - 1); jj = slice(E.HI["jsd"] - g.jsd, E.HI["jed"] - g.jsd + 1)
    assert bits_equal(o["uh"][:, jj, fI], E.normal_trans[:, :, 0]) and bits_equal(o["u_cor"][:, jj, fI], E.normal_vel[:, :, 0])
    assert not o["du_cor"][jj, fI].any()
    FA = g.H_subroundoff * g.dy_Cu[jj, fI]
    for k in range(g.nk):
        nv = E.normal_vel[k, :, 0]
        FA = np.where(np.abs(nv) > 0.0, FA + E.normal_trans[k, :, 0] / np.where(nv == 0.0, 1.0, nv), FA)
    for n in ("FA_u_W0", "FA_u_WW", "FA_u_E0", "FA_u_EE"):
        assert bits_equal(o["bt"][n][jj, fI], FA), n
    fJ = S.HI["JsdB"] - (g.jsd - 1); ii = slice(S.HI["isd"] - g.isd, S.HI["ied"] - g.isd + 1)
    assert bits_equal(o["vh"][:, fJ, ii], S.normal_trans[:, 0, :]) and bits_equal(o["v_cor"][:, fJ, ii], S.normal_vel[:, 0, :])
    # the barotropic transport is still matched on the faces that are solved for
    uh_sum = o["uh"].sum(0)
    solved = (OBC.segnum_u == 0) & (g.mask2dCu > 0)
    sj, si = g.csl(_abi.POS_U)
    err = np.abs(uh_sum - o["uhbt"])[sj, si][solved[sj, si]]
    assert err.max() <= 1e-6 * np.abs(o["uhbt"]).max()


def turned_segments(segs, ni, nj):
    """the segment strings of the grid turned by tests/rotation.py (x' = y, y' = -x): E -> S, N -> E, W -> N, S -> W"""
    out = []
    for s in segs:
        l, m, n, act = parse_segment_str(ni, nj, s)
        if s.replace(" ", "")[:2] == "I=":
            out.append(",".join([f"J={ni - l}", f"I={m}:{n}"] + act))
        else:
            out.append(",".join([f"I={l}", f"J={ni - m}:{ni - n}"] + act))
    return out


@pytest.mark.parametrize("first_direction", [0, 1])
@pytest.mark.parametrize("segs", [TC3 + ["I=9,J=4:11,ORLANSKI"], ["I=N,J=0:N,SIMPLE", "J=N,I=N:0,SIMPLE", "I=0,J=N:0,FLATHER", "J=5,I=3:14,GRADIENT"]],
                         ids=["tc3+inner", "simple"])
def test_oracle_turns_with_the_grid(segs, first_direction):
    """continuity_PPM's two directions are one text in the oracle and two in the reference: a quarter turn of the grid, the state and the
    segments gives the turned answers to the bit (the reference's rotational-reproducibility test applied to the OBC branches)"""
    g, st, OBC = obc_case(segs, first_direction=first_direction)
    o = run(g, st, OBC)
    gr = rotate_grid(g)
    OBCr = ocean_OBC_type(gr, turned_segments(segs, g.ni, g.nj))
    ur, vr = rot_vector(st["u"], st["v"])
    str_ = dict(u=ur, v=vr, h=rot(st["h"]), visc_rem_u=rot(st["visc_rem_v"]), visc_rem_v=rot(st["visc_rem_u"]))
    for s, sr in zip(OBC.segment, OBCr.segment):      # the external values of a u segment become those of the v' segment it turns into
        if s.specified and s.on_pe:
            if s.is_E_or_W:      # u (nk, j, 1) -> v' = -u at i' = j: (nk, 1, i')
                sr.normal_vel[:] = -np.swapaxes(s.normal_vel, 1, 2); sr.normal_trans[:] = -np.swapaxes(s.normal_trans, 1, 2)
            else:                # v (nk, 1, i) -> u' = v at j' = ni - 1 - i: (nk, j', 1)
                sr.normal_vel[:] = np.swapaxes(s.normal_vel, 1, 2)[:, ::-1, :]; sr.normal_trans[:] = np.swapaxes(s.normal_trans, 1, 2)[:, ::-1, :]
    cs = orc.continuity_cs(g.nk, g.Angstrom_H)
    hr = str_["h"].copy(); uhr = np.zeros_like(ur); vhr = np.zeros_like(vr)
    ubr, vbr = rot_vector(o["uhbt"], o["vhbt"])
    ucr, vcr = np.zeros_like(ur), np.zeros_like(vr)
    arrs, btst = orc.make_bt_cont(gr, with_h=True)
    orc.continuity(gr, cs, ur, vr, str_["h"].copy(), hr, uhr, vhr, 900.0, uhbt=ubr, vhbt=vbr, visc_rem_u=str_["visc_rem_u"],
                   visc_rem_v=str_["visc_rem_v"], u_cor=ucr, v_cor=vcr, bt_cont=btst, OBC=OBCr)
    assert bits_equal(interior(g, unrot(hr)), interior(g, o["h"]))
    same = np.array_equal      # (a vector component changes sign with the turn: zero transports come back as -0)
    uh_b, vh_b = unrot_vector(uhr, vhr)
    assert same(interior(g, uh_b, _abi.POS_U), interior(g, o["uh"], _abi.POS_U)) and same(interior(g, vh_b, _abi.POS_V), interior(g, o["vh"], _abi.POS_V))
    uc_b, vc_b = unrot_vector(ucr, vcr)
    assert same(interior(g, uc_b, _abi.POS_U), interior(g, o["u_cor"], _abi.POS_U)) and same(interior(g, vc_b, _abi.POS_V), interior(g, o["v_cor"], _abi.POS_V))
    # h_u of the original frame is h_v' turned back (a thickness: no sign)
    assert bits_equal(interior(g, unrot(arrs["h_v"]), _abi.POS_U), interior(g, o["bt"]["h_u"], _abi.POS_U))
    assert bits_equal(interior(g, unrot(arrs["h_u"]), _abi.POS_V), interior(g, o["bt"]["h_v"], _abi.POS_V))


OBC_GPU_CASES = [dict(segs=TC3 + ["I=9,J=4:11,ORLANSKI", "J=7,I=15:3,GRADIENT"]), dict(segs=TC3, cs=dict(simple_2nd=True)),
                 dict(segs=TC3 + ["I=9,J=4:11,ORLANSKI"], cs=dict(monotonic=True)), dict(segs=TC3, cs=dict(upwind_1st=True)),
                 dict(segs=["I=N,J=0:N,SIMPLE", "J=0,I=0:N,SIMPLE", "I=0,J=N:0,FLATHER,ORLANSKI", "J=N,I=N:0,SIMPLE,FLATHER"]),
                 dict(segs=["I=N,J=0:N,SIMPLE", "J=5,I=3:14,GRADIENT"], uhbt=False), dict(segs=TC3, bt=False),
                 dict(segs=TC3 + ["I=9,J=4:11,ORLANSKI"], cs=dict(aggress_adjust=True)), dict(segs=[], cs={}),
                 dict(segs=TC3 + ["J=7,I=15:3,GRADIENT", "J=7,I=12:5,SIMPLE"], nk=12, ni=70, nj=20)]      # (overlapping segments: the later one's number)


@pytest.mark.gpu
@pytest.mark.parametrize("case", OBC_GPU_CASES, ids=[str(n) for n in range(len(OBC_GPU_CASES))])
@pytest.mark.parametrize("first_direction", [0, 1])
@pytest.mark.parametrize("space", ["device", "host"])
def test_gpu_continuity_with_open_boundaries_matches_oracle_bitwise(case, first_direction, space):
    import torch
    from mom6_amd.continuity import BT_cont_type, continuity, continuity_PPM_init
    from mom6_amd.tracer_advect import DeviceGrid
    case = dict(case)
    cskw = case.pop("cs", {}); uhbt = case.pop("uhbt", True); bt = case.pop("bt", True)
    g, st, OBC = obc_case(case.pop("segs"), first_direction=first_direction, **case)
    want = run(g, st, OBC, uhbt=uhbt, bt=bt, **cskw)
    dg = DeviceGrid(g)
    put = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if space == "device" else (lambda a: np.ascontiguousarray(a).copy())
    get = (lambda a: a.cpu().numpy()) if space == "device" else (lambda a: a)
    CS = continuity_PPM_init(dg, **cskw)
    o = dict(h=put(st["h"]), uh=put(np.zeros_like(st["u"])), vh=put(np.zeros_like(st["v"])))
    kw = dict(visc_rem_u=put(st["visc_rem_u"]), visc_rem_v=put(st["visc_rem_v"]))
    if uhbt:
        o.update(u_cor=put(np.zeros_like(st["u"])), v_cor=put(np.zeros_like(st["v"])), du_cor=put(g.zeros2(_abi.POS_U)), dv_cor=put(g.zeros2(_abi.POS_V)))
        kw.update(uhbt=put(want["uhbt"]), vhbt=put(want["vhbt"]), u_cor=o["u_cor"], v_cor=o["v_cor"], du_cor=o["du_cor"], dv_cor=o["dv_cor"])
    if bt:
        arrs = {n: put(np.zeros_like(a)) for n, a in want["bt"].items()}
        kw["BT_cont"] = BT_cont_type(**arrs)
    continuity(put(st["u"]), put(st["v"]), put(st["h"]), o["h"], o["uh"], o["vh"], 900.0, dg, CS, OBC=OBC, **kw)
    dg.sync()
    for n, pos in (("h", _abi.POS_H), ("uh", _abi.POS_U), ("vh", _abi.POS_V)) + ((("u_cor", _abi.POS_U), ("v_cor", _abi.POS_V), ("du_cor", _abi.POS_U), ("dv_cor", _abi.POS_V)) if uhbt else ()):
        assert bits_equal(interior(g, get(o[n]), pos), interior(g, want[n], pos)), n
    if bt:
        for n in want["bt"]:
            pos = _abi.POS_U if (n in _abi.BT_CONT_U or n == "h_u") else _abi.POS_V
            assert bits_equal(interior(g, get(arrs[n]), pos), interior(g, want["bt"][n], pos)), n
    dg.close()


@pytest.mark.gpu
def test_gpu_continuity_refuses_a_broken_obc():
    import torch
    from mom6_amd._lib import Mom6HipError
    from mom6_amd.continuity import continuity, continuity_PPM_init
    from mom6_amd.tracer_advect import DeviceGrid
    g, st, OBC = obc_case(["I=N,J=0:N,SIMPLE"])
    dg = DeviceGrid(g)
    OBC.segment[0].normal_vel = None
    d = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in st.items()}
    with pytest.raises(Mom6HipError, match="normal_trans and normal_vel are required"):
        continuity(d["u"], d["v"], d["h"], d["h"].clone(), torch.zeros_like(d["u"]), torch.zeros_like(d["v"]), 900.0, dg, continuity_PPM_init(dg), OBC=OBC)
    dg.close()


def _write_obc_case(path, g, st, OBC, uhbt, vhbt, dt=900.0):
    """the input file of tests/fortran/obc_driver.F90"""
    with open(path, "wb") as f:
        np.array([g.ni, g.nj, g.nk, g.halo, int(g.reentrant_x), int(g.reentrant_y), g.first_direction, 0], dtype="<i4").tofile(f)
        np.array([g.Angstrom_H, g.H_subroundoff, g.dZ_subroundoff, g.H_to_Z, g.Z_to_H, g.g_Earth, g.Rho0, dt], dtype="<f8").tofile(f)
        for n in _abi.ALL_METRICS:
            np.ascontiguousarray(g.metrics[n], dtype="<f8").tofile(f)
        for a in (st["u"], st["v"], st["h"], uhbt, vhbt, st["visc_rem_u"], st["visc_rem_v"]):
            np.ascontiguousarray(a, dtype="<f8").tofile(f)
        np.array([OBC.number_of_segments, OBC.OBC_pe, OBC.open_u_BCs_exist_globally, OBC.open_v_BCs_exist_globally, OBC.specified_u_BCs_exist_globally,
                  OBC.specified_v_BCs_exist_globally, OBC.Flather_u_BCs_exist_globally, OBC.Flather_v_BCs_exist_globally], dtype="<i4").tofile(f)
        for s in OBC.segment:
            np.array([s.direction, s.open, s.specified, s.on_pe, s.is_E_or_W, s.is_N_or_S] +
                     [s.HI.get(k, 0) for k in ("IsdB", "IedB", "JsdB", "JedB", "isd", "ied", "jsd", "jed")], dtype="<i4").tofile(f)
        OBC.segnum_u.astype("<i4").tofile(f); OBC.segnum_v.astype("<i4").tofile(f)
        for s in OBC.segment:
            if s.specified and s.on_pe:
                np.ascontiguousarray(s.normal_trans, dtype="<f8").tofile(f); np.ascontiguousarray(s.normal_vel, dtype="<f8").tofile(f)
        np.array([OBC.zero_vorticity, OBC.freeslip_vorticity, OBC.computed_vorticity, OBC.specified_vorticity], dtype="<i4").tofile(f)
        for s in OBC.segment:
            if s.on_pe:
                np.ascontiguousarray(s.tangential_vel, dtype="<f8").tofile(f); np.ascontiguousarray(s.tangential_grad, dtype="<f8").tofile(f)
        np.array([OBC.zero_strain, OBC.freeslip_strain, OBC.computed_strain, OBC.zero_biharmonic], dtype="<i4").tofile(f)


def test_obc_driver_compiles(tmp_path):
    import os
    from test_fortran_abi import FC, _build_shims
    if not os.path.exists(FC):
        pytest.skip("amdflang not present")
    _build_shims(tmp_path, driver="obc_driver")


@pytest.mark.gpu
@pytest.mark.parametrize("segs", [TC3 + ["I=9,J=4:11,ORLANSKI"], ["I=N,J=0:N,SIMPLE", "J=0,I=0:N,SIMPLE", "I=0,J=N:0,FLATHER,ORLANSKI", "J=N,I=N:0,GRADIENT"]],
                         ids=["tc3+inner", "simple"])
def test_continuity_with_an_associated_OBC_from_fortran(tmp_path, segs):
    """continuity_PPM of the module shim with the reference's ocean_OBC_type (its segments, segnum arrays and external values) on host arrays:
    the oracle's bits"""
    import os, subprocess
    from test_fortran_abi import FC, _build_shims
    if not os.path.exists(FC):
        pytest.skip("amdflang not present")
    exe = _build_shims(tmp_path, driver="obc_driver")
    g, st, OBC = obc_case(segs)
    OBC.freeslip_vorticity = segs is not None and len(segs) == 5      # tc3's OBC_FREESLIP_VORTICITY; the other set: the computed vorticity
    OBC.computed_vorticity = not OBC.freeslip_vorticity
    OBC.freeslip_strain = OBC.zero_biharmonic = OBC.freeslip_vorticity      # tc3's OBC_FREESLIP_STRAIN, OBC_ZERO_BIHARMONIC; the other set: OBC_ZERO_STRAIN
    OBC.zero_strain = not OBC.freeslip_strain
    rng = np.random.default_rng(12)
    for s in OBC.segment:
        if s.on_pe:
            s.tangential_vel[:] = 0.1 * rng.standard_normal(s.tangential_vel.shape)
    want = run(g, st, OBC)
    want["CAu"], want["CAv"] = orc.coradcalc(g, st["u"], st["v"], st["h"], want["uh"], want["vh"], bound_coriolis=True, OBC=OBC)
    # set_viscous_BBL in layer mode (tc3's CDRAG, DRAG_BG_VEL, BBL_THICK_MIN), then vertvisc_coef / vertvisc with the wind stress the driver states
    m = g.metrics
    visc = orc.vertvisc_type(Kv_bbl_u=g.zeros2(_abi.POS_U), bbl_thick_u=g.zeros2(_abi.POS_U), Kv_bbl_v=g.zeros2(_abi.POS_V), bbl_thick_v=g.zeros2(_abi.POS_V))
    orc.set_viscous_BBL(g, orc.set_visc_cs(g, 10.0, 1.0e-4, cdrag=0.002, drag_bg_vel=0.05, BBL_thick_min=0.1, BBL_use_EOS=False,
                                           Rlay=1025.0 + 0.5 * np.arange(g.nk)), st["u"], st["v"], st["h"], None, None, None, visc, OBC=OBC)
    for n in ("bbl_thick_u", "bbl_thick_v", "Kv_bbl_u", "Kv_bbl_v"):
        want[n] = visc._keep[n]
    vcs = orc.vertvisc_cs(g, Kv=1.0e-4, Hbbl=10.0, Hmix=20.0)
    want["u1"], want["v1"] = st["u"].copy(), st["v"].copy()
    orc.vertvisc_coef(g, vcs, want["u1"], want["v1"], st["h"], visc, 900.0, OBC=OBC)
    orc.vertvisc(g, vcs, want["u1"], want["v1"], st["h"], np.ascontiguousarray(0.05 * m["mask2dCu"]), np.ascontiguousarray(-0.02 * m["mask2dCv"]),
                 visc, 900.0, OBC=OBC)
    # (BOUND_CORIOLIS = True, set for CoriolisAdv_init as in tc3, is the default of BOUND_CORIOLIS_BIHARM; BOUND_CORIOLIS_VEL defaults to MAXVEL)
    want["diffu"], want["diffv"] = orc.horizontal_viscosity(
        g, orc.hor_visc_cs(g, 900.0, Laplacian=1, Kh=25.0, Kh_vel_scale=0.003, Smagorinsky_Kh=1, Smag_Lap_const=0.15, Ah_vel_scale=0.003, Smagorinsky_Ah=1,
                           Smag_bi_const=0.06, bound_Coriolis=1, bound_Cor_vel=3.0e8), st["u"], st["v"], st["h"], 900.0, OBC=OBC)
    _write_obc_case(str(tmp_path / "in.bin"), g, st, OBC, want["uhbt"], want["vhbt"])
    # advect_tracer (PPM:H3) of two tracers with the transports of the continuity step, every segment with a registry; the values as the driver
    # states them (exact quotients of small integers of the Fortran indices)
    shp = g.shape3(_abi.POS_H)
    fk, fj, fi = (np.arange(n)[sl] + 1 for n, sl in zip(shp, ((slice(None), None, None), (None, slice(None), None), (None, None, slice(None)))))
    tr = [1.0 + ((3 * fi + 5 * fj + 7 * fk) % 11).astype(np.float64) / 11.0, ((2 * fi + 3 * fj + fk) % 7).astype(np.float64) / 7.0]
    for n, s in enumerate(OBC.segment):
        if s.on_pe:
            i0, j0 = (s.HI["IsdB"], s.HI["jsd"]) if s.is_E_or_W else (s.HI["isd"], s.HI["JsdB"])
            nk_, nj_, ni_ = s.normal_vel.shape
            si, sj, sk = i0 + np.arange(ni_)[None, None, :], j0 + np.arange(nj_)[None, :, None], 1 + np.arange(nk_)[:, None, None]
            s.tr_Reg = [dict(ntr_index=1, tres=5.0 + ((si + 2 * sj + 3 * sk) % 13).astype(np.float64) / 13.0),
                        dict(ntr_index=2, OBC_inflow_conc=0.25 + 0.125 * (n + 1))]
    orc.advect_tracer(g, want["h"], 900.0 * want["uh"], 900.0 * want["vh"], 900.0, 900.0, "PPM:H3", tr, OBC=OBC)
    want["tr1"], want["tr2"] = tr
    # update_segment_tracer_reservoirs with the same transports (the values as the driver states them)
    for n, s in enumerate(OBC.segment):
        if s.on_pe:
            s.Tr_InvLscale_in, s.Tr_InvLscale_out = 1.0e-4, (0.0 if (n + 1) % 2 == 0 else 3.0e-5)
            i0, j0 = (s.HI["IsdB"], s.HI["jsd"]) if s.is_E_or_W else (s.HI["isd"], s.HI["JsdB"])
            nk_, nj_, ni_ = s.normal_vel.shape
            si, sj, sk = i0 + np.arange(ni_)[None, None, :], j0 + np.arange(nj_)[None, :, None], 1 + np.arange(nk_)[:, None, None]
            s.tr_Reg[0]["t"] = np.ascontiguousarray(7.0 + ((2 * si + sj + sk) % 5).astype(np.float64) / 5.0)
    orc.update_segment_tracer_reservoirs(g, 900.0 * want["uh"], 900.0 * want["vh"], want["h"], OBC, 900.0, tr)
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0 and "obc_driver ok" in r.stdout, r.stderr[-800:]
    raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8")
    names = ["h", "uh", "vh", "u_cor", "v_cor", "FA_u_W0", "FA_u_WW", "FA_u_E0", "FA_u_EE", "uBT_WW", "uBT_EE", "FA_v_S0", "FA_v_SS", "FA_v_N0",
             "FA_v_NN", "vBT_SS", "vBT_NN", "h_u", "h_v", "CAu", "CAv", "bbl_thick_u", "bbl_thick_v", "Kv_bbl_u", "Kv_bbl_v", "u1", "v1", "diffu", "diffv", "tr1", "tr2"]
    arrs = [want[n] if n in want else want["bt"][n] for n in names]
    tres = [s.tr_Reg[0]["tres"] for s in OBC.segment if s.on_pe]
    raw, raw_tres = raw[:sum(a.size for a in arrs)], raw[sum(a.size for a in arrs):]
    assert raw_tres.size == sum(a.size for a in tres)
    for n, (a, w) in enumerate(zip(np.split(raw_tres, np.cumsum([a.size for a in tres])[:-1]), tres)):
        assert bits_equal(a.reshape(w.shape), w), ("tres", n, np.argwhere(a.reshape(w.shape) != w)[:4].tolist())
    got = np.split(raw, np.cumsum([a.size for a in arrs])[:-1])
    for n, a, w in zip(names, got, arrs):
        pos = _abi.POS_U if n in ("uh", "u_cor", "h_u", "CAu", "u1", "bbl_thick_u", "Kv_bbl_u", "diffu") or n.startswith(("FA_u", "uBT")) else (_abi.POS_V if n in ("vh", "v_cor", "h_v", "CAv", "v1", "bbl_thick_v", "Kv_bbl_v", "diffv") or n.startswith(("FA_v", "vBT")) else _abi.POS_H)
        ga, wa = interior(g, a.reshape(w.shape), pos), interior(g, w, pos)
        assert bits_equal(ga, wa), (n, np.argwhere(ga != wa)[:6].tolist(), ga[ga != wa][:4], wa[ga != wa][:4])
