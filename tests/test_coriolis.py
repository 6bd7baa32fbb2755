"""CorAdCalc: CPU checks of the oracle (oracle/coriolis_adv.c) through properties the discretisation
guarantees, and GPU parity of libmom6hip against the oracle (bit-exact fp64).
The reference holds no known-answer vectors for CorAdCalc (parity unpinned)."""
import numpy as np
import pytest

from mom6_amd import _abi, synth
from helpers import bits_equal, interior

SCHEMES = ["SADOURNY75_ENERGY", "SADOURNY75_ENSTRO", "ARAKAWA_HSU90", "ARAKAWA_LAMB81", "ARAKAWA_LAMB_BLEND", "ROBUST_ENSTRO"]
# options beyond the scheme name: CORIOLIS_EN_DIS, PV_ADV_SCHEME, the blend parameters (MOM_CoriolisAdv.F90:1079-1199)
VARIANTS = [("SADOURNY75_ENERGY", dict(coriolis_en_dis=True)), ("ROBUST_ENSTRO", dict(pv_adv_scheme="PV_ADV_UPWIND1")),
            ("ARAKAWA_LAMB_BLEND", dict(coriolis_blend_f_eff_max=2.5, coriolis_blend_wt_lin=0.5)),
            ("ARAKAWA_LAMB_BLEND", dict(coriolis_blend_f_eff_max=2.0)), ("ARAKAWA_LAMB_BLEND", dict(coriolis_blend_f_eff_max=40.0))]
KES = ["KE_ARAKAWA", "KE_SIMPLE_GUDONOV", "KE_GUDONOV"]


def dyn_case(ni=40, nj=28, nk=3, seed=2, **kw):
    g = synth.make_grid(ni, nj, nk, seed=seed + 10, **kw)
    st = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=seed).items()}
    return g, st


def test_state_at_rest_has_no_acceleration(oracle):
    g, st = dyn_case()
    z = lambda a: np.zeros_like(a)
    for s in SCHEMES:
        CAu, CAv = oracle.coradcalc(g, z(st["u"]), z(st["v"]), st["h"], z(st["uh"]), z(st["vh"]), s, bound_coriolis=True)
        assert np.all(CAu == 0.0) and np.all(CAv == 0.0)


def test_sadourny_energy_coriolis_does_no_work(oracle):
    """With f only (u.grad terms removed by using uh, vh of a flow but zero u, v: KE = 0, zeta = 0), the
    Sadourny energy scheme conserves energy: sum(uh*CAu*dxCu) + sum(vh*CAv*dyCv) = 0 to roundoff on a closed
    or periodic domain, also with land (Sadourny 1975; src/core/MOM_CoriolisAdv.F90:667-672,786-791)."""
    # periodic in x, walls in y (the synthetic metrics vary with latitude, so a y-periodic seam would not close)
    g, st = dyn_case(ni=32, nj=24, nk=2, reentrant_x=True, reentrant_y=False, land_frac=0.0)
    z = lambda a: np.zeros_like(a)
    CAu, CAv = oracle.coradcalc(g, z(st["u"]), z(st["v"]), st["h"], st["uh"], st["vh"], "SADOURNY75_ENERGY")
    wu = interior(g, st["uh"] * CAu * g.dxCu, _abi.POS_U)[..., 1:]      # each periodic face once
    wv = interior(g, st["vh"] * CAv * g.dyCv, _abi.POS_V)
    scale = np.abs(wu).sum() + np.abs(wv).sum()
    assert scale > 0
    assert abs(wu.sum() + wv.sum()) < 1e-12 * scale


def test_bound_coriolis_bounds(oracle):
    g, st = dyn_case()
    a = oracle.coradcalc(g, st["u"], st["v"], st["h"], st["uh"], st["vh"], "SADOURNY75_ENERGY", bound_coriolis=False)
    b = oracle.coradcalc(g, st["u"], st["v"], st["h"], st["uh"], st["vh"], "SADOURNY75_ENERGY", bound_coriolis=True)
    assert not bits_equal(a[0], b[0])      # the bound is active somewhere (vanished layers)
    assert np.all(np.isfinite(b[0])) and np.all(np.isfinite(b[1]))


def test_arakawa_lamb_blend_limits(oracle):
    """ARAKAWA_LAMB_BLEND (:543-590) with CORIOLIS_BLEND_F_EFF_MAX <= 2 is the Sadourny energy scheme (Sad_wt = 1, AL_wt = 0:
    a = q/4 ...: the same terms in another grouping, so equal to roundoff, not to the bit); with a huge F_EFF_MAX every point
    takes the Arakawa & Lamb weights (AL_wt = 1, Sad_wt = 0): equal to ARAKAWA_LAMB81 to roundoff"""
    g, st = dyn_case(land_frac=0.2)
    args = (g, st["u"], st["v"], st["h"], st["uh"], st["vh"])
    sad = oracle.coradcalc(*args, "SADOURNY75_ENERGY")
    lo = oracle.coradcalc(*args, "ARAKAWA_LAMB_BLEND", coriolis_blend_f_eff_max=2.0)
    al = oracle.coradcalc(*args, "ARAKAWA_LAMB81")
    hi = oracle.coradcalc(*args, "ARAKAWA_LAMB_BLEND", coriolis_blend_f_eff_max=1.0e30)
    mid = oracle.coradcalc(*args, "ARAKAWA_LAMB_BLEND")
    for a, b in ((sad, lo), (al, hi)):
        for x, y in zip(a, b):
            assert np.abs(x - y).max() <= 1e-12 * np.abs(x).max()
    assert not np.array_equal(mid[0], lo[0]) and not np.array_equal(mid[0], hi[0])      # the default blends (vanished layers next to thick ones)


def test_arakawa_lamb_coriolis_does_no_work(oracle):
    """Arakawa & Lamb 1981 conserves energy too: with f only, sum(uh*CAu*dxCu) + sum(vh*CAv*dyCv) = 0 to roundoff (the ep_u /
    ep_v terms, :717-721 and :841-845, are what closes it)"""
    g, st = dyn_case(ni=32, nj=24, nk=2, reentrant_x=True, reentrant_y=False, land_frac=0.0)
    z = lambda a: np.zeros_like(a)
    for sch in ("ARAKAWA_LAMB81", "ARAKAWA_HSU90"):
        CAu, CAv = oracle.coradcalc(g, z(st["u"]), z(st["v"]), st["h"], st["uh"], st["vh"], sch)
        wu = interior(g, st["uh"] * CAu * g.dxCu, _abi.POS_U)[..., 1:]
        wv = interior(g, st["vh"] * CAv * g.dyCv, _abi.POS_V)
        scale = np.abs(wu).sum() + np.abs(wv).sum()
        assert scale > 0 and abs(wu.sum() + wv.sum()) < 1e-11 * scale, sch


def test_en_dis_and_robust_enstro(oracle):
    """CORIOLIS_EN_DIS picks, face by face, the thickness-flux estimate that takes energy out: the work of the Coriolis term
    is then <= that of the plain Sadourny scheme (which is zero).  ROBUST_ENSTRO is finite with vanished layers, its upwind
    PV advection differs from the centred one, and CoriolisAdv_init switches BOUND_CORIOLIS off for both (:1155)."""
    g, st = dyn_case(ni=32, nj=24, nk=2, reentrant_x=True, reentrant_y=False, land_frac=0.0)
    z = lambda a: np.zeros_like(a)
    # (u, v of the flow are needed: the choice is made on the sign of q*u)
    CAu, CAv = oracle.coradcalc(g, st["u"], st["v"], st["h"], st["uh"], st["vh"], "SADOURNY75_ENERGY", coriolis_en_dis=True)
    CAu0, CAv0 = oracle.coradcalc(g, st["u"], st["v"], st["h"], st["uh"], st["vh"], "SADOURNY75_ENERGY")
    assert not np.array_equal(CAu, CAu0) and np.all(np.isfinite(CAu)) and np.all(np.isfinite(CAv))
    assert oracle.coradcalc(g, st["u"], st["v"], st["h"], st["uh"], st["vh"], "SADOURNY75_ENERGY", coriolis_en_dis=True,
                            bound_coriolis=True)[0].tobytes() == CAu.tobytes()
    g2, s2 = dyn_case(land_frac=0.3)
    a = oracle.coradcalc(g2, s2["u"], s2["v"], s2["h"], s2["uh"], s2["vh"], "ROBUST_ENSTRO")
    b = oracle.coradcalc(g2, s2["u"], s2["v"], s2["h"], s2["uh"], s2["vh"], "ROBUST_ENSTRO", pv_adv_scheme="PV_ADV_UPWIND1")
    c = oracle.coradcalc(g2, s2["u"], s2["v"], s2["h"], s2["uh"], s2["vh"], "ROBUST_ENSTRO", bound_coriolis=True)
    assert np.all(np.isfinite(a[0])) and np.all(np.isfinite(b[1])) and not np.array_equal(a[0], b[0]) and a[0].tobytes() == c[0].tobytes()


def _gpu_case(oracle, scheme, ke, extra):
    import torch
    from mom6_amd.coriolis_adv import CorAdCalc, CoriolisAdv_init
    from mom6_amd.tracer_advect import DeviceGrid
    for (ni, nj, nk, topo) in [(70, 21, 3, (True, False)), (44, 40, 2, (True, True)), (10, 8, 8, (False, False)),
                               (130, 9, 2, (True, False))]:
        g, st = dyn_case(ni, nj, nk, seed=ni, reentrant_x=topo[0], reentrant_y=topo[1])
        dg = DeviceGrid(g)
        for no_slip, bound in ((False, False), (True, True)):
            ref = oracle.coradcalc(g, st["u"], st["v"], st["h"], st["uh"], st["vh"], scheme, ke, no_slip, bound, **extra)
            CS = CoriolisAdv_init(coriolis_scheme=scheme, ke_scheme=ke, no_slip=no_slip, bound_coriolis=bound, **extra)
            CAu, CAv = np.zeros_like(st["u"]), np.zeros_like(st["v"])
            CorAdCalc(st["u"], st["v"], st["h"], st["uh"], st["vh"], CAu, CAv, None, dg, CS)     # HOST
            assert bits_equal(ref[0], CAu), (scheme, extra, no_slip, bound, np.argwhere(ref[0] != CAu)[:3])
            assert bits_equal(ref[1], CAv), (scheme, extra, no_slip, bound, np.argwhere(ref[1] != CAv)[:3])
            d = {k: torch.from_numpy(st[k]).cuda() for k in ("u", "v", "h", "uh", "vh")}
            dCAu, dCAv = torch.zeros_like(d["u"]), torch.zeros_like(d["v"])
            CorAdCalc(d["u"], d["v"], d["h"], d["uh"], d["vh"], dCAu, dCAv, None, dg, CS)       # DEVICE
            dg.sync()
            assert bits_equal(ref[0], dCAu.cpu().numpy()) and bits_equal(ref[1], dCAv.cpu().numpy())
        dg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scheme,extra", VARIANTS)
def test_gpu_parity_of_the_scheme_options(oracle, scheme, extra):
    _gpu_case(oracle, scheme, "KE_ARAKAWA", extra)


@pytest.mark.gpu
@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("ke", KES)
def test_gpu_parity(oracle, scheme, ke):
    import torch
    from mom6_amd.coriolis_adv import CorAdCalc, CoriolisAdv_init
    from mom6_amd.tracer_advect import DeviceGrid
    for (ni, nj, nk, topo) in [(70, 21, 3, (True, False)), (44, 40, 2, (True, True)), (10, 8, 8, (False, False)),
                               (130, 9, 2, (True, False))]:
        g, st = dyn_case(ni, nj, nk, seed=ni, reentrant_x=topo[0], reentrant_y=topo[1])
        dg = DeviceGrid(g)
        for no_slip in (False, True):
            for bound in (False, True):
                ref = oracle.coradcalc(g, st["u"], st["v"], st["h"], st["uh"], st["vh"], scheme, ke, no_slip, bound)
                CS = CoriolisAdv_init(coriolis_scheme=scheme, ke_scheme=ke, no_slip=no_slip, bound_coriolis=bound)
                CAu, CAv = np.zeros_like(st["u"]), np.zeros_like(st["v"])
                CorAdCalc(st["u"], st["v"], st["h"], st["uh"], st["vh"], CAu, CAv, None, dg, CS)     # HOST
                assert bits_equal(ref[0], CAu), (scheme, ke, no_slip, bound, np.argwhere(ref[0] != CAu)[:3])
                assert bits_equal(ref[1], CAv), (scheme, ke, no_slip, bound, np.argwhere(ref[1] != CAv)[:3])
                d = {k: torch.from_numpy(st[k]).cuda() for k in ("u", "v", "h", "uh", "vh")}
                dCAu, dCAv = torch.zeros_like(d["u"]), torch.zeros_like(d["v"])
                CorAdCalc(d["u"], d["v"], d["h"], d["uh"], d["vh"], dCAu, dCAv, None, dg, CS)       # DEVICE
                dg.sync()
                assert bits_equal(ref[0], dCAu.cpu().numpy()) and bits_equal(ref[1], dCAv.cpu().numpy())
        dg.close()
