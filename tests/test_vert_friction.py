"""MOM_vert_friction (vertvisc_coef / vertvisc / vertvisc_remnant): CPU checks of the oracle (oracle/vert_friction.c)
through what the implicit scheme guarantees, and GPU parity of libmom6hip against the oracle (bit-exact fp64).  The
reference holds no known-answer vectors for this module (parity unpinned)."""
import numpy as np
import pytest

from mom6_amd import _abi, synth
from helpers import bits_equal
from oracle import orc

VARIANTS = [
    dict(),                                                        # defaults: BOTTOMDRAGLAW, arithmetic thickness, no ML scheme
    dict(harmonic_visc=True),
    dict(harm_BL_val=1.0),
    dict(Kvml_invZ2=1.0e-2, Hmix=20.0),
    dict(bottomdraglaw=False, Kv_extra_bbl=5.0e-4),
    dict(bottomdraglaw=False),
    dict(direct_stress=True, Hmix=15.0),
    dict(CFL_based_trunc=False, maxvel=0.05, vel_underflow=1.0e-3),
    dict(dynamic_viscous_ML=True),                                 # DYNAMIC_VISCOUS_ML (.testing/tc1, tc2): visc%nkml_visc_u/v, forces%ustar
    dict(nkml=2),                                                  # the viscous mixed layer of a bulk mixed layer (tc1)
    dict(dynamic_viscous_ML=True, nkml=2, Kvml_invZ2=1.0e-2, Hmix=20.0),
]


def vv_case(ni=24, nj=18, nk=7, seed=5, with_shear=False, with_ray=False, umax=0.2, with_ml=False, **topo):
    g = synth.make_grid(ni, nj, nk, seed=seed + 40, **topo)
    st = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=seed, umax=umax).items()}
    rng = np.random.default_rng(seed)
    su, sv, sh = g.shape2(_abi.POS_U), g.shape2(_abi.POS_V), g.shape2(_abi.POS_H)
    arrs = dict(Kv_bbl_u=1.0e-3 * (0.5 + rng.random(su)), Kv_bbl_v=1.0e-3 * (0.5 + rng.random(sv)),
                bbl_thick_u=2.0 + 10.0 * rng.random(su), bbl_thick_v=2.0 + 10.0 * rng.random(sv))
    if with_shear:
        arrs["Kv_shear"] = 1.0e-3 * rng.random((nk + 1,) + sh)
    if with_ray:
        arrs["Ray_u"] = 1.0e-5 * rng.random(g.shape3(_abi.POS_U)); arrs["Ray_v"] = 1.0e-5 * rng.random(g.shape3(_abi.POS_V))
    taux = np.ascontiguousarray(0.1 * np.cos(np.linspace(0, 3, su[0]))[:, None] * g.mask2dCu)
    tauy = np.ascontiguousarray(0.05 * g.mask2dCv * rng.random(sv))
    if with_ml:      # what set_viscous_ML leaves in visc (a fractional number of layers) and forces%ustar, with a few calm cells
        arrs["nkml_visc_u"] = np.clip(nk * rng.random(su) ** 2, 0.0, nk); arrs["nkml_visc_v"] = np.clip(nk * rng.random(sv) ** 2, 0.0, nk)
        arrs["nkml_visc_u"][::3, ::4] = np.floor(arrs["nkml_visc_u"][::3, ::4])
        arrs["ustar"] = np.where(rng.random(sh) < 0.05, 0.0, 0.002 + 0.01 * rng.random(sh))
    return g, st, arrs, taux, tauy


def face_cols(g, a3, pos):
    """compute-range face columns of a face array, (nk, ncol), and the matching mask"""
    if pos == _abi.POS_U:
        sl = (slice(g.jsc - g.jsd, g.jec - g.jsd + 1), slice(g.isc - g.isd, g.iec - g.isd + 2))
        m = np.asarray(g.mask2dCu)[sl]
    else:
        sl = (slice(g.jsc - g.jsd, g.jec - g.jsd + 2), slice(g.isc - g.isd, g.iec - g.isd + 1))
        m = np.asarray(g.mask2dCv)[sl]
    return a3[(slice(None),) + sl].reshape(a3.shape[0], -1), m.reshape(-1) > 0


def test_momentum_budget_of_the_implicit_solve(oracle):
    """sum_k h_u u changes by the surface stress minus the bottom drag and Rayleigh drag of the NEW velocities"""
    g, st, arrs, taux, tauy = vv_case(with_ray=True)
    dt = 900.0
    cs = orc.vertvisc_cs(g, Kv=1.0e-4, Hbbl=10.0, CFL_based_trunc=False)
    visc = orc.vertvisc_type(**arrs)
    orc.vertvisc_coef(g, cs, st["u"], st["v"], st["h"], visc, dt)
    u, v = st["u"].copy(), st["v"].copy()
    tbx, tby = g.zeros2(_abi.POS_U), g.zeros2(_abi.POS_V)
    orc.vertvisc(g, cs, u, v, st["h"], taux, tauy, visc, dt, tbx, tby)
    for pos, new, old, tau, tb, hn, ray in ((_abi.POS_U, u, st["u"], taux, tbx, "h_u", "Ray_u"), (_abi.POS_V, v, st["v"], tauy, tby, "h_v", "Ray_v")):
        hv, m = face_cols(g, cs._arrs[hn], pos)
        un, _ = face_cols(g, new, pos); uo, _ = face_cols(g, old, pos)
        r, _ = face_cols(g, arrs[ray], pos)
        t, _ = face_cols(g, tau[None], pos); b, _ = face_cols(g, tb[None], pos)
        lhs = (hv * un).sum(0) - (hv * uo).sum(0)
        rhs = dt * (t[0] - b[0]) / cs.H_to_RZ      # taux_bot holds the bottom + Rayleigh drag of the new velocities
        scale = np.abs(hv * uo).sum(0) + np.abs(dt * t[0] / cs.H_to_RZ) + 1e-30
        assert np.all(np.abs(lhs - rhs)[m] <= 1e-11 * scale[m])
        assert np.abs(r).max() > 0


def test_visc_rem_is_the_response_to_a_unit_velocity(oracle):
    g, st, arrs, taux, tauy = vv_case(with_ray=True)
    dt = 600.0
    cs = orc.vertvisc_cs(g, Kv=1.0e-4, Hbbl=10.0, CFL_based_trunc=False)      # (no truncation of the unit velocity)
    visc = orc.vertvisc_type(**arrs)
    orc.vertvisc_coef(g, cs, st["u"], st["v"], st["h"], visc, dt)
    vru, vrv = g.zeros3(_abi.POS_U), g.zeros3(_abi.POS_V)
    orc.vertvisc_remnant(g, cs, visc, vru, vrv, dt)
    u1, v1 = np.ones_like(vru), np.ones_like(vrv)
    orc.vertvisc(g, cs, u1, v1, st["h"], 0.0 * taux, 0.0 * tauy, visc, dt)
    for pos, vr, x in ((_abi.POS_U, vru, u1), (_abi.POS_V, vrv, v1)):
        a, m = face_cols(g, vr, pos); b, _ = face_cols(g, x, pos)
        assert bits_equal(a[:, m], b[:, m])
        assert a[:, m].min() >= 0.0 and a[:, m].max() <= 1.0 + 1e-14


def test_strong_viscosity_with_a_no_slip_bottom_brings_the_column_to_rest(oracle):
    g, st, arrs, taux, tauy = vv_case()
    cs = orc.vertvisc_cs(g, Kv=1.0e5, Hbbl=10.0, bottomdraglaw=False, CFL_based_trunc=False)
    visc = orc.vertvisc_type(**arrs)
    orc.vertvisc_coef(g, cs, st["u"], st["v"], st["h"], visc, 900.0)
    u, v = st["u"].copy(), st["v"].copy()
    orc.vertvisc(g, cs, u, v, st["h"], 0.0 * taux, 0.0 * tauy, visc, 900.0)
    un, m = face_cols(g, u, _abi.POS_U); uo, _ = face_cols(g, st["u"], _abi.POS_U)
    assert np.abs(uo[:, m]).max() > 0.05 and np.abs(un[:, m]).max() < 0.1 * np.abs(uo[:, m]).max()


def test_truncation_counts_and_limits(oracle):
    g, st, arrs, taux, tauy = vv_case(umax=0.2)
    cs = orc.vertvisc_cs(g, Kv=1.0e-4, Hbbl=10.0, CFL_based_trunc=False, maxvel=0.05)
    visc = orc.vertvisc_type(**arrs)
    orc.vertvisc_coef(g, cs, st["u"], st["v"], st["h"], visc, 900.0)
    u, v = st["u"].copy(), st["v"].copy()
    orc.vertvisc(g, cs, u, v, st["h"], taux, tauy, visc, 900.0)
    un, m = face_cols(g, u, _abi.POS_U)
    assert np.abs(un[:, m]).max() <= 0.05 and cs.ntrunc > 0


@pytest.mark.gpu
@pytest.mark.parametrize("variant", range(len(VARIANTS)))
def test_gpu_parity(oracle, variant):
    import torch
    from mom6_amd.tracer_advect import DeviceGrid
    from mom6_amd.vert_friction import (vertvisc, vertvisc_and_remnant, vertvisc_coef, vertvisc_init, vertvisc_ntrunc, vertvisc_remnant,
                                        vertvisc_step, vertvisc_type)
    kw = VARIANTS[variant]
    pk = dict(KV=1.0e-4, HBBL=10.0)
    names = dict(harmonic_visc="HARMONIC_VISC", harm_BL_val="HARMONIC_BL_SCALE", Kvml_invZ2="KV_ML_INVZ2", Hmix="HMIX_FIXED",
                 bottomdraglaw="BOTTOMDRAGLAW", Kv_extra_bbl="KV_EXTRA_BBL", direct_stress="DIRECT_STRESS",
                 CFL_based_trunc="CFL_BASED_TRUNCATIONS", maxvel="MAXVEL", vel_underflow="VEL_UNDERFLOW",
                 dynamic_viscous_ML="DYNAMIC_VISCOUS_ML", nkml="NKML")
    pk.update({names[k]: v for k, v in kw.items()})
    ml = bool(kw.get("dynamic_viscous_ML") or kw.get("nkml"))
    for (ni, nj, nk, topo, extra) in [(24, 18, 7, dict(reentrant_x=True), dict()), (70, 9, 3, dict(reentrant_x=False), dict(with_shear=True, with_ray=True)),
                                      (12, 10, 75, dict(reentrant_x=True, reentrant_y=True), dict(with_shear=True))]:
        g, st, arrs, taux, tauy = vv_case(ni, nj, nk, seed=ni, with_ml=ml, **extra, **topo)
        dt = 900.0
        dz = np.ascontiguousarray(st["h"] * g.H_to_Z) if nk == 3 else None      # an explicit dz on one of the grids
        rcs = orc.vertvisc_cs(g, Kv=1.0e-4, Hbbl=10.0, **kw)
        rvisc = orc.vertvisc_type(**arrs)
        orc.vertvisc_coef(g, rcs, st["u"], st["v"], st["h"], rvisc, dt, dz=dz)
        rvru, rvrv = g.zeros3(_abi.POS_U), g.zeros3(_abi.POS_V)
        orc.vertvisc_remnant(g, rcs, rvisc, rvru, rvrv, dt)
        ru, rv = st["u"].copy(), st["v"].copy()
        rtbx, rtby = g.zeros2(_abi.POS_U), g.zeros2(_abi.POS_V)
        orc.vertvisc(g, rcs, ru, rv, st["h"], taux, tauy, rvisc, dt, rtbx, rtby)
        dg = DeviceGrid(g)
        for resident in (False, True):
            X = (lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a).copy()).cuda()) if resident else \
                (lambda a: None if a is None else np.ascontiguousarray(a).copy())
            N = (lambda a: a.cpu().numpy()) if resident else (lambda a: a)
            CS = vertvisc_init(dg, device_arrays=resident, **pk)
            visc = vertvisc_type(**{n: X(a) for n, a in arrs.items()})
            u, v, h = X(st["u"]), X(st["v"]), X(st["h"])
            vertvisc_coef(u, v, h, X(dz), None, visc, None, dt, dg, CS)
            what = (variant, (ni, nj, nk), resident)
            for n in ("a_u", "a_v", "h_u", "h_v"):
                assert bits_equal(rcs._arrs[n], N(CS.arrays[n])), (what, n, np.argwhere(rcs._arrs[n] != N(CS.arrays[n]))[:3])
            vru, vrv = X(g.zeros3(_abi.POS_U)), X(g.zeros3(_abi.POS_V))
            vertvisc_remnant(visc, vru, vrv, dt, dg, CS)
            assert bits_equal(rvru, N(vru)) and bits_equal(rvrv, N(vrv)), (what, "visc_rem")
            tbx, tby = X(g.zeros2(_abi.POS_U)), X(g.zeros2(_abi.POS_V))
            vertvisc(u, v, h, (X(taux), X(tauy)), visc, dt, None, None, None, dg, CS, tbx, tby)
            assert bits_equal(ru, N(u)), (what, "u", np.argwhere(ru != N(u))[:3])
            assert bits_equal(rv, N(v)), (what, "v", np.argwhere(rv != N(v))[:3])
            assert bits_equal(rtbx, N(tbx)) and bits_equal(rtby, N(tby)), (what, "tau_bot")
            assert vertvisc_ntrunc(dg, CS) == rcs.ntrunc, (what, "ntrunc", CS.ntrunc, rcs.ntrunc)
            # the fused pair gives what the two calls give
            u2, v2, r2u, r2v = X(st["u"]), X(st["v"]), X(g.zeros3(_abi.POS_U)), X(g.zeros3(_abi.POS_V))
            tbx, tby = X(g.zeros2(_abi.POS_U)), X(g.zeros2(_abi.POS_V))
            vertvisc_and_remnant(u2, v2, h, (X(taux), X(tauy)), visc, dt, dg, CS, r2u, r2v, tbx, tby)
            assert bits_equal(ru, N(u2)) and bits_equal(rv, N(v2)) and bits_equal(rvru, N(r2u)) and bits_equal(rvrv, N(r2v)), (what, "fused")
            assert bits_equal(rtbx, N(tbx)) and bits_equal(rtby, N(tby)), (what, "fused tau_bot")
            assert vertvisc_ntrunc(dg, CS) == 2 * rcs.ntrunc, (what, "fused ntrunc")
            # coefficients + solve in one kernel, on a fresh control structure
            CS3 = vertvisc_init(dg, device_arrays=resident, **pk)
            u3, v3, r3u, r3v = X(st["u"]), X(st["v"]), X(g.zeros3(_abi.POS_U)), X(g.zeros3(_abi.POS_V))
            tbx, tby = X(g.zeros2(_abi.POS_U)), X(g.zeros2(_abi.POS_V))
            vertvisc_step(u3, v3, h, X(dz), (X(taux), X(tauy)), visc, dt, dg, CS3, r3u, r3v, True, tbx, tby)
            for n in ("a_u", "a_v", "h_u", "h_v"):
                assert bits_equal(rcs._arrs[n], N(CS3.arrays[n])), (what, "step", n)
            assert bits_equal(ru, N(u3)) and bits_equal(rv, N(v3)) and bits_equal(rvru, N(r3u)) and bits_equal(rvrv, N(r3v)), (what, "step")
            assert bits_equal(rtbx, N(tbx)) and bits_equal(rtby, N(tby)), (what, "step tau_bot")
            assert vertvisc_ntrunc(dg, CS3) == rcs.ntrunc, (what, "step ntrunc")
            u4, v4 = X(st["u"]), X(st["v"])
            vertvisc_step(u4, v4, h, X(dz), None, visc, dt, dg, CS3, r3u, r3v, False)      # :598-600: velocities untouched
            assert bits_equal(st["u"], N(u4)) and bits_equal(rvru, N(r3u)) and bits_equal(rvrv, N(r3v)), (what, "step, remnant only")
        dg.close()


def test_surface_boundary_layer_viscosity(oracle):
    """DYNAMIC_VISCOUS_ML / a bulk mixed layer (find_coupling_coef :2047-2252): the coupling can only grow, it grows only above the
    base of the boundary layer (K <= ceil(nkml_visc)), not where the wind is calm, and more with a stronger wind"""
    g, st, arrs, taux, tauy = vv_case(with_ml=True, nk=12)
    base = {n: a for n, a in arrs.items() if n not in ("nkml_visc_u", "nkml_visc_v", "ustar")}
    cs0 = orc.vertvisc_cs(g, Kv=1.0e-4, Hbbl=10.0)
    orc.vertvisc_coef(g, cs0, st["u"], st["v"], st["h"], orc.vertvisc_type(**base), 900.0)
    cs1 = orc.vertvisc_cs(g, Kv=1.0e-4, Hbbl=10.0, dynamic_viscous_ML=True)
    orc.vertvisc_coef(g, cs1, st["u"], st["v"], st["h"], orc.vertvisc_type(**arrs), 900.0)
    cs2 = orc.vertvisc_cs(g, Kv=1.0e-4, Hbbl=10.0, dynamic_viscous_ML=True)
    orc.vertvisc_coef(g, cs2, st["u"], st["v"], st["h"], orc.vertvisc_type(**dict(arrs, ustar=2.0 * arrs["ustar"])), 900.0)
    a0, m = face_cols(g, cs0._arrs["a_u"], _abi.POS_U); a1, _ = face_cols(g, cs1._arrs["a_u"], _abi.POS_U); a2, _ = face_cols(g, cs2._arrs["a_u"], _abi.POS_U)
    nkv, _ = face_cols(g, arrs["nkml_visc_u"][None], _abi.POS_U)
    a0, a1, a2, nkv = a0[:, m], a1[:, m], a2[:, m], nkv[0, m]
    assert np.all(a1 >= a0) and np.any(a1 > a0) and np.all(a2 >= a1) and np.any(a2 > a1)
    K = np.arange(a0.shape[0])[:, None] + 1                       # interface number, one-based
    assert np.all((a1 == a0) | ((K >= 2) & (K <= np.ceil(nkv)[None])))
    assert bits_equal(cs0._arrs["h_u"], cs1._arrs["h_u"])
    # the bulk mixed layer: its nkml layers are the boundary layer
    cs3 = orc.vertvisc_cs(g, Kv=1.0e-4, Hbbl=10.0, nkml=3)
    orc.vertvisc_coef(g, cs3, st["u"], st["v"], st["h"], orc.vertvisc_type(**dict(base, ustar=arrs["ustar"])), 900.0)
    a3, _ = face_cols(g, cs3._arrs["a_u"], _abi.POS_U)
    a3 = a3[:, m]
    assert np.all((a3 == a0) | ((K >= 2) & (K <= 3))) and np.any(a3 > a0)


def test_unsupported_options_are_refused_by_name():
    from mom6_amd._lib import Mom6HipError
    from mom6_amd.vert_friction import vertvisc_CS
    with pytest.raises(Mom6HipError, match="unknown parameter"):
        vertvisc_CS.__init__(vertvisc_CS.__new__(vertvisc_CS), type("G", (), {"grid": synth.make_grid(8, 8, 2)})(), KV=1e-4, HBBL=1.0,
                             device_arrays=False, not_a_parameter=1)
