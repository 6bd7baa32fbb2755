"""CorAdCalc: CPU checks of the oracle (oracle/coriolis_adv.c) through properties the discretisation
guarantees, and GPU parity of libmom6hip against the oracle (bit-exact fp64).
The reference holds no known-answer vectors for CorAdCalc (parity unpinned)."""
import numpy as np
import pytest

from mom6_amd import _abi, synth
from helpers import bits_equal, interior

SCHEMES = ["SADOURNY75_ENERGY", "SADOURNY75_ENSTRO", "ARAKAWA_HSU90"]
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
