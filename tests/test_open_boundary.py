"""radiation_open_bdry_conds for the normal component (src/core/MOM_open_boundary.F90:2196: Orlanski radiation, the gradient condition,
nudging), open_boundary_apply_normal_flow (:3337) and open_boundary_zero_normal_flow (:3374): the oracle against what the routines state
and against a quarter turn of the grid (the reference writes E, W, N, S out separately), on the CPU; the library against the oracle on the
GPU, bit for bit.  (The reference holds no known-answer vectors: parity unpinned.)"""
import numpy as np
import pytest

from helpers import bits_equal, interior
from mom6_amd import _abi, synth
from mom6_amd.open_boundary import ocean_OBC_type
from oracle import orc
from rotation import rot_vector, rotate_grid, unrot_vector
from test_continuity_obc import TC3, open_faces, turned_segments

SEGS = TC3 + ["I=9,J=4:11,ORLANSKI", "J=7,I=15:3,GRADIENT", "I=14,J=12:2,ORLANSKI,NUDGED"]


def rad_case(segs=SEGS, ni=22, nj=16, nk=3, seed=8, reentrant=(False, False)):
    g = synth.make_grid(ni, nj, nk, seed=seed + 50, reentrant_x=reentrant[0], reentrant_y=reentrant[1], land_frac=0.1)
    OBC = ocean_OBC_type(g, segs)
    open_faces(g, OBC)
    st = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=seed).items()}
    rng = np.random.default_rng(seed)
    d = dict(u_old=np.ascontiguousarray(st["u"] + 0.05 * rng.standard_normal(st["u"].shape)),
             v_old=np.ascontiguousarray(st["v"] + 0.05 * rng.standard_normal(st["v"].shape)))
    d["u_new"] = np.ascontiguousarray(d["u_old"] + 0.03 * rng.standard_normal(st["u"].shape))
    d["v_new"] = np.ascontiguousarray(d["v_old"] + 0.03 * rng.standard_normal(st["v"].shape))
    d["rx"] = np.ascontiguousarray(0.5 * rng.random(st["u"].shape)); d["ry"] = np.ascontiguousarray(0.5 * rng.random(st["v"].shape))
    for s in OBC.segment:
        if s.on_pe and s.nudged:
            s.nudged_normal_vel[:] = 0.2 * rng.standard_normal(s.nudged_normal_vel.shape)
            s.Velocity_nudging_timescale_in, s.Velocity_nudging_timescale_out = 3600.0, 86400.0
    return g, d, OBC


def run(g, d, OBC, gamma_uv=0.3, rx_max=1.0, dt=900.0):
    o = {k: v.copy() for k, v in d.items()}
    orc.radiation_open_bdry_conds(g, OBC, o["u_new"], o["u_old"], o["v_new"], o["v_old"], dt, gamma_uv=gamma_uv, rx_max=rx_max,
                                  rx_normal=o["rx"], ry_normal=o["ry"])
    o["normal_vel"] = [None if s.normal_vel is None else s.normal_vel.copy() for s in OBC.segment]
    return o


@pytest.mark.parametrize("gamma_uv", [0.3, 1.0])
def test_what_the_branches_state(gamma_uv):
    g, d, OBC = rad_case()
    o = run(g, d, OBC, gamma_uv=gamma_uv)
    touched_u = np.zeros(g.shape2(_abi.POS_U), dtype=bool); touched_v = np.zeros(g.shape2(_abi.POS_V), dtype=bool)
    for n, s in enumerate(OBC.segment):
        H = s.HI
        if s.is_E_or_W:
            I = H["IsdB"] - (g.isd - 1); jj = slice(H["jsd"] - g.jsd, H["jed"] - g.jsd + 1)
            d1 = -1 if s.direction == _abi.OBC_DIRECTION_E else 1
            touched_u[jj, I] = True
            if n == len(OBC.segment) - 1 or not s.nudged:      # (later segments may overwrite shared faces: none here)
                pass
            un = d["u_new"]
            if s.gradient:
                assert bits_equal(o["u_new"][:, jj, I], un[:, jj, I + d1])
            elif s.radiation and not s.nudged:
                dhdt = d["u_old"][:, jj, I + d1] - un[:, jj, I + d1]; dhdx = un[:, jj, I + d1] - un[:, jj, I + 2 * d1]
                with np.errstate(divide="ignore", invalid="ignore"):
                    rx_new = np.where(dhdt * dhdx > 0.0, np.minimum(dhdt / dhdx, 1.0), 0.0)
                rx_avg = (1.0 - gamma_uv) * d["rx"][:, jj, I] + gamma_uv * rx_new if gamma_uv < 1.0 else rx_new
                assert bits_equal(o["u_new"][:, jj, I], (un[:, jj, I] + rx_avg * un[:, jj, I + d1]) / (1.0 + rx_avg))
                if gamma_uv < 1.0:
                    assert bits_equal(o["rx"][:, jj, I], rx_avg)
                assert (rx_new >= 0.0).all() and (rx_new <= 1.0).all() and (rx_new > 0).any() and (rx_new == 0).any()
        else:
            J = H["JsdB"] - (g.jsd - 1); ii = slice(H["isd"] - g.isd, H["ied"] - g.isd + 1)
            touched_v[J, ii] = True
            if s.gradient:
                d1 = -1 if s.direction == _abi.OBC_DIRECTION_N else 1
                assert bits_equal(o["v_new"][:, J, ii], d["v_new"][:, J + d1, ii])
    # nothing but the segments' faces changes (the domain is closed: the halo update moves nothing)
    assert bits_equal(np.where(touched_u[None], 0.0, o["u_new"]), np.where(touched_u[None], 0.0, d["u_new"]))
    assert bits_equal(np.where(touched_v[None], 0.0, o["v_new"]), np.where(touched_v[None], 0.0, d["v_new"]))
    if gamma_uv >= 1.0:
        assert bits_equal(o["rx"], d["rx"]) and bits_equal(o["ry"], d["ry"])


def test_zero_normal_flow():
    g, d, OBC = rad_case()
    u, v = d["u_new"].copy(), d["v_new"].copy()
    orc.open_boundary_zero_normal_flow(g, OBC, u, v)
    su = np.zeros(g.shape2(_abi.POS_U), dtype=bool); sv = np.zeros(g.shape2(_abi.POS_V), dtype=bool)
    for s in OBC.segment:
        H = s.HI
        if s.is_E_or_W:
            su[H["jsd"] - g.jsd:H["jed"] - g.jsd + 1, H["IsdB"] - (g.isd - 1)] = True
        else:
            sv[H["JsdB"] - (g.jsd - 1), H["isd"] - g.isd:H["ied"] - g.isd + 1] = True
    assert not u[:, su].any() and not v[:, sv].any()
    assert bits_equal(u[:, ~su], d["u_new"][:, ~su]) and bits_equal(v[:, ~sv], d["v_new"][:, ~sv])


def test_oracle_turns_with_the_grid():
    g, d, OBC = rad_case()
    o = run(g, d, OBC)
    gr = rotate_grid(g)
    OBCr = ocean_OBC_type(gr, turned_segments(SEGS, g.ni, g.nj))
    for s, sr in zip(OBC.segment, OBCr.segment):
        sr.Velocity_nudging_timescale_in, sr.Velocity_nudging_timescale_out = s.Velocity_nudging_timescale_in, s.Velocity_nudging_timescale_out
        if s.on_pe and s.nudged:      # u (nk, j, 1) -> v' = -u at i' = j ; v (nk, 1, i) -> u' = v at j' = ni - 1 - i
            sr.nudged_normal_vel[:] = -np.swapaxes(s.nudged_normal_vel, 1, 2) if s.is_E_or_W else np.swapaxes(s.nudged_normal_vel, 1, 2)[:, ::-1, :]
    dr = {}
    dr["u_new"], dr["v_new"] = rot_vector(d["u_new"], d["v_new"]); dr["u_old"], dr["v_old"] = rot_vector(d["u_old"], d["v_old"])
    # the phase speeds are ratios of two differences of the same component: no sign; rx at u faces becomes ry' at v' faces and back
    from rotation import rot, unrot
    dr["rx"], dr["ry"] = rot(d["ry"]), rot(d["rx"])
    orr = run(gr, dr, OBCr)
    bu, bv = unrot_vector(orr["u_new"], orr["v_new"])
    assert np.array_equal(interior(g, bu, _abi.POS_U), interior(g, o["u_new"], _abi.POS_U))
    assert np.array_equal(interior(g, bv, _abi.POS_V), interior(g, o["v_new"], _abi.POS_V))
    assert np.array_equal(interior(g, unrot(orr["ry"]), _abi.POS_U), interior(g, o["rx"], _abi.POS_U))
    assert np.array_equal(interior(g, unrot(orr["rx"]), _abi.POS_V), interior(g, o["ry"], _abi.POS_V))


def test_every_form_of_the_radiation_is_provided():
    g, d, _ = rad_case()
    for segs in (["I=N,J=0:N,OBLIQUE,OBLIQUE_GRAD"], ["I=N,J=0:N,OBLIQUE,OBLIQUE_TAN"]):
        OBC = ocean_OBC_type(g, segs)
        for n in OBL_FIELDS:
            setattr(OBC, n, g.zeros3(_abi.POS_U if n.endswith("_u") else _abi.POS_V))
        run(g, d, OBC)


@pytest.mark.gpu
@pytest.mark.parametrize("gamma_uv", [0.3, 1.0])
@pytest.mark.parametrize("reentrant", [(False, False), (True, False)])
@pytest.mark.parametrize("space", ["device", "host"])
def test_gpu_radiation_matches_oracle_bitwise(gamma_uv, reentrant, space):
    import torch
    from mom6_amd.open_boundary import open_boundary_zero_normal_flow, radiation_open_bdry_conds
    from mom6_amd.tracer_advect import DeviceGrid
    segs = SEGS if not reentrant[0] else ["J=N,I=N:0,FLATHER,ORLANSKI", "J=0,I=0:N,FLATHER,ORLANSKI", "I=9,J=4:11,ORLANSKI", "J=7,I=15:3,GRADIENT",
                                           "I=14,J=12:2,ORLANSKI,NUDGED"]
    g, d, OBC = rad_case(segs, reentrant=reentrant)
    OBC.gamma_uv = gamma_uv
    want = run(g, d, OBC, gamma_uv=gamma_uv)
    g2, d2, OBC2 = rad_case(segs, reentrant=reentrant)      # (the oracle wrote into the first set's segments)
    OBC2.gamma_uv = gamma_uv
    dg = DeviceGrid(g2)
    put = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if space == "device" else (lambda a: np.ascontiguousarray(a).copy())
    get = (lambda a: a.cpu().numpy()) if space == "device" else (lambda a: a)
    o = {k: put(v) for k, v in d2.items()}
    OBC2.rx_normal, OBC2.ry_normal = (o["rx"], o["ry"]) if gamma_uv < 1.0 else (None, None)
    for s in OBC2.segment:
        if s.on_pe:
            s.normal_vel = put(s.normal_vel); s.nudged_normal_vel = put(s.nudged_normal_vel)
    radiation_open_bdry_conds(OBC2, o["u_new"], o["u_old"], o["v_new"], o["v_old"], dg, 900.0)
    dg.sync()
    assert bits_equal(get(o["u_new"]), want["u_new"]) and bits_equal(get(o["v_new"]), want["v_new"])      # (the halos as well: the pass is part of it)
    assert bits_equal(get(o["rx"]), want["rx"]) and bits_equal(get(o["ry"]), want["ry"])
    for s, w in zip(OBC2.segment, want["normal_vel"]):
        if s.on_pe and (s.radiation or s.gradient):
            assert bits_equal(get(s.normal_vel), w)
    u, v = put(d2["u_new"]), put(d2["v_new"])
    open_boundary_zero_normal_flow(OBC2, dg, u, v)
    dg.sync()
    uw, vw = d2["u_new"].copy(), d2["v_new"].copy()
    orc.open_boundary_zero_normal_flow(g, OBC, uw, vw)
    assert bits_equal(get(u), uw) and bits_equal(get(v), vw)
    dg.close()


@pytest.mark.gpu
def test_gpu_advect_tracer_with_segments_without_a_tracer_registry_is_the_closed_advection():
    """advect_x / advect_y read of an associated OBC only the tracer registries of its segments (segment%tr_Reg,
    MOM_tracer_advect.F90:442-477, :580-627): without one the answers are those of OBC => NULL() (with one: tests/test_advect_obc.py)"""
    from helpers import advect_case
    from mom6_amd._lib import Mom6HipError
    from mom6_amd.tracer_advect import DeviceGrid, advect_tracer, tracer_advect_init
    from test_continuity_obc import TC3
    g, case = advect_case(ni=24, nj=18, nk=3, ntr=2, reentrant_x=False, reentrant_y=False)
    OBC = ocean_OBC_type(g, TC3)
    dg = DeviceGrid(g)
    CS = tracer_advect_init(900.0, "PLM")
    out = []
    for obc in (None, OBC):
        tr = [t.copy() for t in case["tr"]]
        advect_tracer(case["h_end"], case["uhtr"], case["vhtr"], obc, 3600.0, dg, CS, tr)
        out.append(tr)
    assert all(bits_equal(a, b) for a, b in zip(*out))
    dg.close()


# ---- the tangential forms: ORLANSKI_TAN / ORLANSKI_GRAD / NUDGED_TAN / NUDGED_GRAD (:2403-2455 E, :2648-2700 W, :2892-2945 N, :3137-3190 S) ----
TAN_SEGS = ["J=N,I=N:0,ORLANSKI,ORLANSKI_TAN,ORLANSKI_GRAD", "J=0,I=0:N,ORLANSKI,ORLANSKI_TAN,NUDGED_TAN", "I=N,J=0:N,ORLANSKI,ORLANSKI_GRAD,NUDGED_GRAD",
            "I=0,J=N:0,ORLANSKI,NUDGED,ORLANSKI_TAN,ORLANSKI_GRAD,NUDGED_TAN,NUDGED_GRAD", "I=9,J=4:11,ORLANSKI,ORLANSKI_TAN,ORLANSKI_GRAD",
            "J=7,I=15:3,ORLANSKI,ORLANSKI_TAN,ORLANSKI_GRAD"]


def tan_case(seed=8, **kw):
    g, d, OBC = rad_case(TAN_SEGS, seed=seed, **kw)
    rng = np.random.default_rng(seed + 7)
    for s in OBC.segment:
        if not s.on_pe:
            continue
        shp = s.tangential_vel.shape
        s.tangential_vel[:] = 0.1 * rng.standard_normal(shp); s.tangential_grad[:] = 1e-5 * rng.standard_normal(shp)
        s.nudged_tangential_vel = 0.2 * rng.standard_normal(shp); s.nudged_tangential_grad = 2e-5 * rng.standard_normal(shp)
        s.Velocity_nudging_timescale_in, s.Velocity_nudging_timescale_out = 3600.0, 86400.0
    return g, d, OBC


def expected_tangential(g, d, s, r_normal, tv0, tg0, gamma_uv, rx_max, dt):
    """the four blocks of the reference written out one by one (Fortran indices; arrays here are [k, j - jsd, i - isd (+1 for face / corner arrays)])"""
    H = s.HI
    nk = g.nk
    tv, tg = tv0.copy(), tg0.copy()
    ew = s.is_E_or_W
    vn, vo = (d["v_new"], d["v_old"]) if ew else (d["u_new"], d["u_old"])
    Id = np.asarray(g.IdxBu if ew else g.IdyBu)
    V = lambda i, J, k: vn[k, J - (g.jsd - 1), i - g.isd]       # v(i, J, k)
    Vo = lambda i, J, k: vo[k, J - (g.jsd - 1), i - g.isd]
    Uu = lambda I, j, k: vn[k, j - g.jsd, I - (g.isd - 1)]      # u(I, j, k)
    Uo = lambda I, j, k: vo[k, j - g.jsd, I - (g.isd - 1)]
    Q = lambda I, J: Id[J - (g.jsd - 1), I - (g.isd - 1)]
    if ew:
        I = H["IsdB"]; R = lambda j, k: r_normal[k, j - g.jsd, I - (g.isd - 1)]
        east = s.direction == _abi.OBC_DIRECTION_E
        for k in range(nk):
            for J in range(H["JsdB"], H["JedB"] + 1):
                if gamma_uv < 1.0:
                    r = R(H["jsd"], k) if J == H["JsdB"] else (R(H["jed"], k) if J == H["JedB"] else 0.5 * (R(J, k) + R(J + 1, k)))
                else:
                    a, b = (I, I - 1) if east else (I + 1, I + 2)
                    dhdt = Vo(a, J, k) - V(a, J, k); dhdx = V(a, J, k) - V(b, J, k)
                    r = min(dhdt / dhdx, rx_max) if dhdt * dhdx > 0.0 else 0.0
                tau = s.Velocity_nudging_timescale_in if r <= 0.0 else s.Velocity_nudging_timescale_out
                g2 = dt / (tau + dt)
                jj = J - H["JsdB"]
                if s.radiation_tan:
                    tv[k, jj, 0] = ((V(I, J, k) + r * V(I - 1, J, k)) if east else (V(I + 1, J, k) + r * V(I + 2, J, k))) / (1.0 + r)
                if s.nudged_tan:
                    tv[k, jj, 0] = (1.0 - g2) * tv[k, jj, 0] + g2 * s.nudged_tangential_vel[k, jj, 0]
                if s.radiation_grad and max(H["JsdB"], g.jsd + 1) <= J <= min(H["JedB"], g.jed - 1):
                    if east:
                        tg[k, jj, 0] = ((V(I, J, k) - V(I - 1, J, k)) * Q(I - 1, J) + r * (V(I - 1, J, k) - V(I - 2, J, k)) * Q(I - 2, J)) / (1.0 + r)
                    else:
                        tg[k, jj, 0] = ((V(I + 2, J, k) - V(I + 1, J, k)) * Q(I + 1, J) + r * (V(I + 3, J, k) - V(I + 2, J, k)) * Q(I + 2, J)) / (1.0 + r)
                if s.nudged_grad:
                    tg[k, jj, 0] = (1.0 - g2) * tg[k, jj, 0] + g2 * s.nudged_tangential_grad[k, jj, 0]
    else:
        J = H["JsdB"]; R = lambda i, k: r_normal[k, J - (g.jsd - 1), i - g.isd]
        north = s.direction == _abi.OBC_DIRECTION_N
        for k in range(nk):
            for I in range(H["IsdB"], H["IedB"] + 1):
                if gamma_uv < 1.0:
                    r = R(H["isd"], k) if I == H["IsdB"] else (R(H["ied"], k) if I == H["IedB"] else 0.5 * (R(I, k) + R(I + 1, k)))
                else:
                    a, b = (J - 1, J - 2) if north else (J + 1, J + 2)
                    dhdt = Uo(I, a, k) - Uu(I, a, k); dhdy = Uu(I, a, k) - Uu(I, b, k)
                    r = min(dhdt / dhdy, rx_max) if dhdt * dhdy > 0.0 else 0.0
                tau = s.Velocity_nudging_timescale_in if r <= 0.0 else s.Velocity_nudging_timescale_out
                g2 = dt / (tau + dt)
                ii = I - H["IsdB"]
                if s.radiation_tan:
                    tv[k, 0, ii] = ((Uu(I, J, k) + r * Uu(I, J - 1, k)) if north else (Uu(I, J + 1, k) + r * Uu(I, J + 2, k))) / (1.0 + r)
                if s.nudged_tan:
                    tv[k, 0, ii] = (1.0 - g2) * tv[k, 0, ii] + g2 * s.nudged_tangential_vel[k, 0, ii]
                if s.radiation_grad and max(H["IsdB"], g.isd + 1) <= I <= min(H["IedB"], g.ied - 1):
                    if north:
                        tg[k, 0, ii] = ((Uu(I, J, k) - Uu(I, J - 1, k)) * Q(I, J - 1) + r * (Uu(I, J - 1, k) - Uu(I, J - 2, k)) * Q(I, J - 2)) / (1.0 + r)
                    else:
                        tg[k, 0, ii] = ((Uu(I, J + 2, k) - Uu(I, J + 1, k)) * Q(I, J + 1) + r * (Uu(I, J + 3, k) - Uu(I, J + 2, k)) * Q(I, J + 2)) / (1.0 + r)
                if s.nudged_grad:
                    tg[k, 0, ii] = (1.0 - g2) * tg[k, 0, ii] + g2 * s.nudged_tangential_grad[k, 0, ii]
    return tv, tg


@pytest.mark.parametrize("gamma_uv", [0.3, 1.0])
def test_oracle_tangential_forms_are_the_four_blocks_of_the_reference(gamma_uv):
    g, d, OBC = tan_case()
    before = [(s.tangential_vel.copy(), s.tangential_grad.copy()) for s in OBC.segment]
    o = run(g, d, OBC, gamma_uv=gamma_uv)
    # the normal part is what it is without the tangential forms
    OBCn = ocean_OBC_type(g, [",".join(w for w in t.split(",") if not w.endswith(("_TAN", "_GRAD"))) for t in TAN_SEGS])
    for s, sn in zip(OBC.segment, OBCn.segment):
        sn.Velocity_nudging_timescale_in, sn.Velocity_nudging_timescale_out = s.Velocity_nudging_timescale_in, s.Velocity_nudging_timescale_out
        if s.nudged:
            sn.nudged_normal_vel[:] = s.nudged_normal_vel
    on = run(g, d, OBCn, gamma_uv=gamma_uv)
    assert bits_equal(o["u_new"], on["u_new"]) and bits_equal(o["rx"], on["rx"]) and bits_equal(o["ry"], on["ry"])
    n_changed = 0
    for s, (tv0, tg0) in zip(OBC.segment, before):
        r_normal = o["rx"] if s.is_E_or_W else o["ry"]      # segment%rx_norm_rad as the normal part has just left it
        tv, tg = expected_tangential(g, d, s, r_normal, tv0, tg0, gamma_uv, 1.0, 900.0)
        assert bits_equal(s.tangential_vel, tv), (s.direction, "tangential_vel", np.argwhere(s.tangential_vel != tv)[:4].tolist())
        assert bits_equal(s.tangential_grad, tg), (s.direction, "tangential_grad", np.argwhere(s.tangential_grad != tg)[:4].tolist())
        n_changed += int((tv != tv0).sum() + (tg != tg0).sum())
    assert n_changed > 100


@pytest.mark.gpu
@pytest.mark.parametrize("gamma_uv", [0.3, 1.0])
@pytest.mark.parametrize("space", ["device", "host"])
def test_gpu_tangential_forms_match_oracle_bitwise(gamma_uv, space):
    import copy
    import torch
    from mom6_amd.open_boundary import radiation_open_bdry_conds
    from mom6_amd.tracer_advect import DeviceGrid
    for kw in (dict(), dict(ni=150, nj=40, nk=2, seed=5)):
        g, d, OBC = tan_case(**kw)
        OBC.gamma_uv, OBC.rx_max = gamma_uv, 1.0
        ref = copy.deepcopy(OBC)
        o = run(g, d, ref, gamma_uv=gamma_uv)
        dev = space == "device"
        X = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if dev else (lambda a: a.copy())
        N = (lambda a: a.cpu().numpy()) if dev else (lambda a: a)
        f = {k: X(v) for k, v in d.items()}
        OBC.rx_normal, OBC.ry_normal = f["rx"], f["ry"]
        if dev:
            OBC.cuda()
        dg = DeviceGrid(g)
        radiation_open_bdry_conds(OBC, f["u_new"], f["u_old"], f["v_new"], f["v_old"], dg, 900.0)
        dg.sync()
        assert bits_equal(N(f["u_new"]), o["u_new"]) and bits_equal(N(f["v_new"]), o["v_new"]) and bits_equal(N(f["rx"]), o["rx"])
        for n, (s, sr) in enumerate(zip(OBC.segment, ref.segment)):
            assert bits_equal(N(s.tangential_vel), sr.tangential_vel), (n, "tangential_vel", np.argwhere(N(s.tangential_vel) != sr.tangential_vel)[:4].tolist())
            assert bits_equal(N(s.tangential_grad), sr.tangential_grad), (n, "tangential_grad")
            assert bits_equal(N(s.normal_vel), sr.normal_vel), (n, "normal_vel")
        dg.close()


# ---- oblique radiation (OBLIQUE: :2349-2383 E, :2593-2628 W, :2838-2872 N, :3082-3117 S; gradient_at_q_points :3407) ----
OBL_SEGS = ["J=N,I=N:0,OBLIQUE", "J=0,I=0:N,OBLIQUE,NUDGED", "I=N,J=0:N,OBLIQUE", "I=0,J=N:0,OBLIQUE", "I=9,J=4:11,OBLIQUE", "J=7,I=15:3,OBLIQUE,NUDGED"]
OBL_FIELDS = ("rx_oblique_u", "ry_oblique_u", "cff_normal_u", "rx_oblique_v", "ry_oblique_v", "cff_normal_v")


def obl_case(seed=8, **kw):
    g, d, OBC = rad_case(OBL_SEGS, seed=seed, **kw)
    rng = np.random.default_rng(seed + 3)
    for n in OBL_FIELDS:
        shp = g.shape3(_abi.POS_U if n.endswith("_u") else _abi.POS_V)
        setattr(OBC, n, np.ascontiguousarray((1e-4 * rng.random(shp)) if n.startswith("cff") else 1e-4 * rng.standard_normal(shp)))
    return g, d, OBC


def expected_oblique(g, d, s, OBC0, gamma_uv, rx_max, dt):
    """the block of one direction written out with the reference's own indices; returns normal_vel and the three stored fields on the faces"""
    H = s.HI; nk = g.nk; eps = 1.0e-20
    mB = np.asarray(g.mask2dBu)
    U = lambda a, I, j, k: a[k, j - g.jsd, I - (g.isd - 1)]
    V = lambda a, i, J, k: a[k, J - (g.jsd - 1), i - g.isd]
    MB = lambda I, J: mB[J - (g.jsd - 1), I - (g.isd - 1)]
    un, uo, vn, vo = d["u_new"], d["u_old"], d["v_new"], d["v_old"]
    nv = s.normal_vel.copy()
    st = {n: getattr(OBC0, n).copy() for n in OBL_FIELDS}
    fmax = lambda a, b: a if a > b else b
    fmin = lambda a, b: a if a < b else b
    if s.is_E_or_W:
        I = H["IsdB"]; d1 = -1 if s.direction == _abi.OBC_DIRECTION_E else 1
        lo, hi = max(H["JsdB"], g.jsd), min(H["JedB"], g.jed - 1)
        GN = lambda J, m, k: 0.0 if not (lo <= J <= hi) else (U(un, I + (d1 if m == 1 else 0), J + 1, k) - U(un, I + (d1 if m == 1 else 0), J, k)) * MB(I + (d1 if m == 1 else 0), J)
        for k in range(nk):
            for j in range(H["jsd"], H["jed"] + 1):
                J = j
                dhdt = U(uo, I + d1, j, k) - U(un, I + d1, j, k); dhdx = U(un, I + d1, j, k) - U(un, I + 2 * d1, j, k)
                ssum = GN(J, 1, k) + GN(J - 1, 1, k)
                dhdy = GN(J - 1, 1, k) if dhdt * ssum > 0.0 else (0.0 if dhdt * ssum == 0.0 else GN(J, 1, k))
                if dhdt * dhdx < 0.0:
                    dhdt = 0.0
                cff = fmax(dhdx * dhdx + dhdy * dhdy, eps); rx = fmin(dhdt * dhdx, cff * rx_max); ry = fmin(cff, fmax(dhdt * dhdy, -cff))
                idx = (k, j - g.jsd, I - (g.isd - 1))
                if gamma_uv < 1.0:
                    rx = (1.0 - gamma_uv) * OBC0.rx_oblique_u[idx] + gamma_uv * rx; ry = (1.0 - gamma_uv) * OBC0.ry_oblique_u[idx] + gamma_uv * ry
                    cff = (1.0 - gamma_uv) * OBC0.cff_normal_u[idx] + gamma_uv * cff
                    st["rx_oblique_u"][idx], st["ry_oblique_u"][idx], st["cff_normal_u"][idx] = rx, ry, cff
                val = ((cff * U(un, I, j, k) + rx * U(un, I + d1, j, k)) - (fmax(ry, 0.0) * GN(J - 1, 2, k) + fmin(ry, 0.0) * GN(J, 2, k))) / (cff + rx)
                if s.nudged:
                    tau = s.Velocity_nudging_timescale_in if dhdt * dhdx <= 0.0 else s.Velocity_nudging_timescale_out
                    g2 = dt / (tau + dt)
                    val = (1.0 - g2) * val + g2 * s.nudged_normal_vel[k, j - H["jsd"], 0]
                nv[k, j - H["jsd"], 0] = val
    else:
        J = H["JsdB"]; d1 = -1 if s.direction == _abi.OBC_DIRECTION_N else 1
        lo, hi = max(H["IsdB"], g.isd), min(H["IedB"], g.ied - 1)
        GN = lambda I, m, k: 0.0 if not (lo <= I <= hi) else (V(vn, I + 1, J + (d1 if m == 1 else 0), k) - V(vn, I, J + (d1 if m == 1 else 0), k)) * MB(I, J + (d1 if m == 1 else 0))
        for k in range(nk):
            for i in range(H["isd"], H["ied"] + 1):
                I = i
                dhdt = V(vo, i, J + d1, k) - V(vn, i, J + d1, k); dhdy = V(vn, i, J + d1, k) - V(vn, i, J + 2 * d1, k)
                ssum = GN(I, 1, k) + GN(I - 1, 1, k)
                dhdx = GN(I - 1, 1, k) if dhdt * ssum > 0.0 else (0.0 if dhdt * ssum == 0.0 else GN(I, 1, k))
                if dhdt * dhdy < 0.0:
                    dhdt = 0.0
                cff = fmax(dhdx * dhdx + dhdy * dhdy, eps); ry = fmin(dhdt * dhdy, cff * rx_max); rx = fmin(cff, fmax(dhdt * dhdx, -cff))
                idx = (k, J - (g.jsd - 1), i - g.isd)
                if gamma_uv < 1.0:
                    rx = (1.0 - gamma_uv) * OBC0.rx_oblique_v[idx] + gamma_uv * rx; ry = (1.0 - gamma_uv) * OBC0.ry_oblique_v[idx] + gamma_uv * ry
                    cff = (1.0 - gamma_uv) * OBC0.cff_normal_v[idx] + gamma_uv * cff
                    st["rx_oblique_v"][idx], st["ry_oblique_v"][idx], st["cff_normal_v"][idx] = rx, ry, cff
                val = ((cff * V(vn, i, J, k) + ry * V(vn, i, J + d1, k)) - (fmax(rx, 0.0) * GN(I - 1, 2, k) + fmin(rx, 0.0) * GN(I, 2, k))) / (cff + ry)
                if s.nudged:
                    tau = s.Velocity_nudging_timescale_in if dhdt * dhdy <= 0.0 else s.Velocity_nudging_timescale_out
                    g2 = dt / (tau + dt)
                    val = (1.0 - g2) * val + g2 * s.nudged_normal_vel[k, 0, i - H["isd"]]
                nv[k, 0, i - H["isd"]] = val
    return nv, st


@pytest.mark.parametrize("gamma_uv", [0.3, 1.0])
def test_oracle_oblique_radiation_is_the_four_blocks_of_the_reference(gamma_uv):
    import copy
    g, d, OBC = obl_case()
    OBC0 = copy.deepcopy(OBC)
    o = run(g, d, OBC, gamma_uv=gamma_uv)
    for n, (s, s0) in enumerate(zip(OBC.segment, OBC0.segment)):
        nv, st = expected_oblique(g, d, s0, OBC0, gamma_uv, 1.0, 900.0)
        assert bits_equal(s.normal_vel, nv), (n, s.direction, np.argwhere(s.normal_vel != nv)[:4].tolist())
        H = s.HI
        for name in OBL_FIELDS:      # (on this segment's own faces; other segments own other faces)
            a, b = getattr(OBC, name), st[name]
            if s.is_E_or_W and name.endswith("_u"):
                sl = (slice(None), slice(H["jsd"] - g.jsd, H["jed"] - g.jsd + 1), H["IsdB"] - (g.isd - 1))
            elif s.is_N_or_S and name.endswith("_v"):
                sl = (slice(None), H["JsdB"] - (g.jsd - 1), slice(H["isd"] - g.isd, H["ied"] - g.isd + 1))
            else:
                continue
            assert bits_equal(a[sl], b[sl]), (n, name)
        # the faces of the segment carry its normal_vel afterwards (open_boundary_apply_normal_flow)
    assert not bits_equal(OBC.rx_oblique_u, OBC0.rx_oblique_u) or gamma_uv >= 1.0
    assert (gamma_uv < 1.0) or all(bits_equal(getattr(OBC, n), getattr(OBC0, n)) for n in OBL_FIELDS)


@pytest.mark.gpu
@pytest.mark.parametrize("gamma_uv", [0.3, 1.0])
@pytest.mark.parametrize("space", ["device", "host"])
def test_gpu_oblique_radiation_matches_oracle_bitwise(gamma_uv, space):
    import copy
    import torch
    from mom6_amd.open_boundary import radiation_open_bdry_conds
    from mom6_amd.tracer_advect import DeviceGrid
    for kw in (dict(), dict(ni=150, nj=40, nk=2, seed=5)):
        g, d, OBC = obl_case(**kw)
        OBC.gamma_uv, OBC.rx_max = gamma_uv, 1.0
        ref = copy.deepcopy(OBC)
        o = run(g, d, ref, gamma_uv=gamma_uv)
        dev = space == "device"
        X = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if dev else (lambda a: a.copy())
        N = (lambda a: a.cpu().numpy()) if dev else (lambda a: a)
        f = {k: X(v) for k, v in d.items()}
        OBC.rx_normal, OBC.ry_normal = f["rx"], f["ry"]
        if dev:
            OBC.cuda()
        dg = DeviceGrid(g)
        radiation_open_bdry_conds(OBC, f["u_new"], f["u_old"], f["v_new"], f["v_old"], dg, 900.0)
        dg.sync()
        assert bits_equal(N(f["u_new"]), o["u_new"]) and bits_equal(N(f["v_new"]), o["v_new"])
        for name in OBL_FIELDS:
            assert bits_equal(N(getattr(OBC, name)), getattr(ref, name)), name
        for n, (s, sr) in enumerate(zip(OBC.segment, ref.segment)):
            assert bits_equal(N(s.normal_vel), sr.normal_vel), (n, "normal_vel", np.argwhere(N(s.normal_vel) != sr.normal_vel)[:4].tolist())
        dg.close()


# ---- the tangential forms of the oblique radiation: OBLIQUE_TAN / OBLIQUE_GRAD (:2456-2556 E, :2701-2801 W, :2946-3046 N, :3191-3291 S) ----
OBLT_SEGS = ["J=N,I=N:0,OBLIQUE,OBLIQUE_TAN,OBLIQUE_GRAD", "J=0,I=0:N,OBLIQUE,OBLIQUE_TAN,NUDGED_TAN", "I=N,J=0:N,OBLIQUE,OBLIQUE_GRAD,NUDGED_GRAD",
             "I=0,J=N:0,OBLIQUE,NUDGED,OBLIQUE_TAN,OBLIQUE_GRAD,NUDGED_TAN,NUDGED_GRAD", "I=9,J=4:11,OBLIQUE,OBLIQUE_TAN,OBLIQUE_GRAD",
             "J=7,I=15:3,OBLIQUE,OBLIQUE_TAN,OBLIQUE_GRAD"]
# per direction, from the reference's text: the rows (offsets from the segment's index) of the tangential component in the rates and in
# tangential_vel [first inside, second], the row pairs (low row) and metric rows of the two gradient terms, the rows of grad_tan(., 1 | 2)
OBLT_ROWS = {_abi.OBC_DIRECTION_E: dict(r0=0, r1=-1, g1=-1, g2=-2, gt1=-1, gt2=0), _abi.OBC_DIRECTION_W: dict(r0=1, r1=2, g1=1, g2=2, gt1=2, gt2=1),
             _abi.OBC_DIRECTION_N: dict(r0=0, r1=-1, g1=-1, g2=-2, gt1=-1, gt2=0), _abi.OBC_DIRECTION_S: dict(r0=1, r1=2, g1=1, g2=2, gt1=2, gt2=1)}


def oblt_case(seed=8, **kw):
    g, d, OBC = rad_case(OBLT_SEGS, seed=seed, **kw)
    rng = np.random.default_rng(seed + 11)
    for n in OBL_FIELDS:
        shp = g.shape3(_abi.POS_U if n.endswith("_u") else _abi.POS_V)
        setattr(OBC, n, np.ascontiguousarray((1e-4 * rng.random(shp)) if n.startswith("cff") else 1e-4 * rng.standard_normal(shp)))
    for s in OBC.segment:
        if s.on_pe:
            shp = s.tangential_vel.shape
            s.tangential_vel[:] = 0.1 * rng.standard_normal(shp); s.tangential_grad[:] = 1e-5 * rng.standard_normal(shp)
            s.nudged_tangential_vel = 0.2 * rng.standard_normal(shp); s.nudged_tangential_grad = 2e-5 * rng.standard_normal(shp)
            s.Velocity_nudging_timescale_in, s.Velocity_nudging_timescale_out = 3600.0, 86400.0
    return g, d, OBC


def expected_oblique_tangential(g, d, s, OBCa, tv0, tg0, gamma_uv, rx_max, dt):
    """OBCa: the OBC after the normal part (its stored fields are the segment's rx_norm_obl, ry_norm_obl, cff_normal)"""
    H = s.HI; nk = g.nk; eps = 1.0e-20
    R = OBLT_ROWS[s.direction]
    ew = s.is_E_or_W
    tn, to = (d["v_new"], d["v_old"]) if ew else (d["u_new"], d["u_old"])
    A = H["IsdB"] if ew else H["JsdB"]
    mT = np.asarray(g.mask2dT); Id = np.asarray(g.IdxBu if ew else g.IdyBu); mC = np.asarray(g.mask2dCu if ew else g.mask2dCv)
    # the tangential component in the row t of cells at the corner point q; masks and metrics likewise (t: across the boundary, q / c: along it)
    Tn = (lambda t, q, k: tn[k, q - (g.jsd - 1), t - g.isd]) if ew else (lambda t, q, k: tn[k, t - g.jsd, q - (g.isd - 1)])
    To = (lambda t, q, k: to[k, q - (g.jsd - 1), t - g.isd]) if ew else (lambda t, q, k: to[k, t - g.jsd, q - (g.isd - 1)])
    MT = (lambda t, c: mT[c - g.jsd, t - g.isd]) if ew else (lambda t, c: mT[t - g.jsd, c - g.isd])
    IDM = (lambda t, q: Id[q - (g.jsd - 1), t - (g.isd - 1)]) if ew else (lambda t, q: Id[t - (g.jsd - 1), q - (g.isd - 1)])
    MC = (lambda t, c: mC[c - g.jsd, t - (g.isd - 1)]) if ew else (lambda t, c: mC[t - (g.jsd - 1), c - g.isd])
    c0, c1 = (H["jsd"], H["jed"]) if ew else (H["isd"], H["ied"])
    q0, q1 = (H["JsdB"], H["JedB"]) if ew else (H["IsdB"], H["IedB"])
    cd0, cd1 = (g.jsd, g.jed) if ew else (g.isd, g.ied)
    GT = lambda c, m, k: 0.0 if not (max(c0 - 1, cd0) <= c <= min(c1 + 1, cd1)) else \
        (Tn(A + (R["gt1"] if m == 1 else R["gt2"]), c, k) - Tn(A + (R["gt1"] if m == 1 else R["gt2"]), c - 1, k)) * MT(A + (R["gt1"] if m == 1 else R["gt2"]), c)
    lo1 = A + R["g1"]; lo2 = A + R["g2"]
    GG2 = lambda c, k: 0.0 if not (max(c0, cd0 + 1) <= c <= min(c1, cd1 - 1)) else \
        (((Tn(lo1 + 1, c, k) - Tn(lo1, c, k)) * IDM(lo1, c)) - (Tn(lo1 + 1, c - 1, k) - Tn(lo1, c - 1, k)) * IDM(lo1, c - 1)) * MC(lo1, c)
    names = ("rx_oblique_u", "ry_oblique_u", "cff_normal_u") if ew else ("ry_oblique_v", "rx_oblique_v", "cff_normal_v")      # (normal, along, cff)
    ST = [getattr(OBCa, n) for n in names]
    F = (lambda a, c, k: a[k, c - g.jsd, A - (g.isd - 1)]) if ew else (lambda a, c, k: a[k, A - (g.jsd - 1), c - g.isd])
    fmax = lambda a, b: a if a > b else b
    fmin = lambda a, b: a if a < b else b
    tv, tg = tv0.copy(), tg0.copy()
    for k in range(nk):
        for q in range(q0, q1 + 1):
            if gamma_uv < 1.0:
                rn, rt, cff = [(F(a, c0, k) if q == q0 else (F(a, c1, k) if q == q1 else 0.5 * (F(a, q, k) + F(a, q + 1, k)))) for a in ST]
            else:
                dhdt = To(A + R["r0"], q, k) - Tn(A + R["r0"], q, k); dhdn = Tn(A + R["r0"], q, k) - Tn(A + R["r1"], q, k)
                ssum = GT(q, 1, k) + GT(q + 1, 1, k)
                dhdl = GT(q, 1, k) if dhdt * ssum > 0.0 else (0.0 if dhdt * ssum == 0.0 else GT(q + 1, 1, k))
                if dhdt * dhdn < 0.0:
                    dhdt = 0.0
                cff = fmax(dhdn * dhdn + dhdl * dhdl, eps); rn = fmin(dhdt * dhdn, cff * rx_max); rt = fmin(cff, fmax(dhdt * dhdl, -cff))
            tau = s.Velocity_nudging_timescale_in if rn <= 0.0 else s.Velocity_nudging_timescale_out
            g2 = dt / (tau + dt)
            idx = (k, q - q0, 0) if ew else (k, 0, q - q0)
            if s.oblique_tan:
                tv[idx] = ((cff * Tn(A + R["r0"], q, k) + rn * Tn(A + R["r1"], q, k)) - (fmax(rt, 0.0) * GT(q, 2, k) + fmin(rt, 0.0) * GT(q + 1, 2, k))) / (cff + rn)
            if s.nudged_tan:
                tv[idx] = (1.0 - g2) * tv[idx] + g2 * s.nudged_tangential_vel[idx]
            if s.oblique_grad and q0 + 1 <= q <= q1 - 1:
                tg[idx] = ((cff * (Tn(lo1 + 1, q, k) - Tn(lo1, q, k)) * IDM(lo1, q) + rn * (Tn(lo2 + 1, q, k) - Tn(lo2, q, k)) * IDM(lo2, q)) -
                           (fmax(rt, 0.0) * GG2(q, k) + fmin(rt, 0.0) * GG2(q + 1, k))) / (cff + rn)
            if s.nudged_grad:
                tg[idx] = (1.0 - g2) * tg[idx] + g2 * s.nudged_tangential_grad[idx]
    return tv, tg


@pytest.mark.parametrize("gamma_uv", [0.3, 1.0])
def test_oracle_oblique_tangential_forms_are_the_blocks_of_the_reference(gamma_uv):
    g, d, OBC = oblt_case()
    before = [(s.tangential_vel.copy(), s.tangential_grad.copy()) for s in OBC.segment]
    run(g, d, OBC, gamma_uv=gamma_uv)
    n_changed = 0
    for s, (tv0, tg0) in zip(OBC.segment, before):
        tv, tg = expected_oblique_tangential(g, d, s, OBC, tv0, tg0, gamma_uv, 1.0, 900.0)
        assert bits_equal(s.tangential_vel, tv), (s.direction, "tangential_vel", np.argwhere(s.tangential_vel != tv)[:4].tolist())
        assert bits_equal(s.tangential_grad, tg), (s.direction, "tangential_grad", np.argwhere(s.tangential_grad != tg)[:4].tolist())
        n_changed += int((tv != tv0).sum() + (tg != tg0).sum())
    assert n_changed > 100


@pytest.mark.gpu
@pytest.mark.parametrize("gamma_uv", [0.3, 1.0])
@pytest.mark.parametrize("space", ["device", "host"])
def test_gpu_oblique_tangential_forms_match_oracle_bitwise(gamma_uv, space):
    import copy
    import torch
    from mom6_amd.open_boundary import radiation_open_bdry_conds
    from mom6_amd.tracer_advect import DeviceGrid
    for kw in (dict(), dict(ni=150, nj=40, nk=2, seed=5)):
        g, d, OBC = oblt_case(**kw)
        OBC.gamma_uv, OBC.rx_max = gamma_uv, 1.0
        ref = copy.deepcopy(OBC)
        o = run(g, d, ref, gamma_uv=gamma_uv)
        dev = space == "device"
        X = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if dev else (lambda a: a.copy())
        N = (lambda a: a.cpu().numpy()) if dev else (lambda a: a)
        f = {k: X(v) for k, v in d.items()}
        OBC.rx_normal, OBC.ry_normal = f["rx"], f["ry"]
        if dev:
            OBC.cuda()
        dg = DeviceGrid(g)
        radiation_open_bdry_conds(OBC, f["u_new"], f["u_old"], f["v_new"], f["v_old"], dg, 900.0)
        dg.sync()
        assert bits_equal(N(f["u_new"]), o["u_new"]) and bits_equal(N(f["v_new"]), o["v_new"])
        for n, (s, sr) in enumerate(zip(OBC.segment, ref.segment)):
            assert bits_equal(N(s.tangential_vel), sr.tangential_vel), (n, "tangential_vel", np.argwhere(N(s.tangential_vel) != sr.tangential_vel)[:4].tolist())
            assert bits_equal(N(s.tangential_grad), sr.tangential_grad), (n, "tangential_grad")
            assert bits_equal(N(s.normal_vel), sr.normal_vel), (n, "normal_vel")
        dg.close()
