"""TRIPOLAR_N: the fold of the halo update and the polarity swaps of btstep, oracle and library, against the unfolded domain
(tests/tripolar.py)."""
import numpy as np
import pytest

import tripolar as tp
from helpers import bits_equal, interior
from mom6_amd import _abi
from oracle import orc

H, U, V, Q = _abi.POS_H, _abi.POS_U, _abi.POS_V, _abi.POS_Q


def test_fold_of_the_halo_update():
    """every staggering, vector and scalar pair: the halo beyond the fold == the rows of the unfolded field"""
    g, g2 = tp.grids(ni=12, nj=6, nk=2)
    rng = np.random.default_rng(1)
    for pos in (H, U, V, Q):
        for vec in ((False, True) if pos in (U, V) else (False,)):
            a2 = tp.symmetrize(g2, rng.standard_normal(g2.shape3(pos)), pos, vec)
            a = tp.folded(g, a2, pos)
            want = a.copy()
            sj, si = g.csl(pos)
            b = np.full_like(a, np.nan); b[:, sj, si] = a[:, sj, si]
            b[:, : g.halo] = a[:, : g.halo]                  # the closed southern halo is not touched
            orc.halo_update(g, b, pos | (0 if (vec or pos in (H, Q)) else _abi.PASS_SCALAR_PAIR))
            assert np.array_equal(b, want), (pos, vec, np.argwhere(b != want)[:4])


def run_pair(nsteps, viscous, rk2b=False, ni=20, nj=8, nk=3, use_bt_cont=True):
    g, g2 = tp.grids(ni=ni, nj=nj, nk=nk)
    d, d2, tau, tau2 = tp.states(g, g2)
    out = []
    for gg, dd, tt in ((g, d, tau), (g2, d2, tau2)):
        kw = dict(rk2b=rk2b, use_bt_cont=use_bt_cont)
        if viscous:
            visc = orc.vertvisc_type(Kv_bbl_u=1.0e-3 * gg.mask2dCu, Kv_bbl_v=1.0e-3 * gg.mask2dCv, bbl_thick_u=5.0 * gg.mask2dCu,
                                     bbl_thick_v=5.0 * gg.mask2dCv)
            kw.update(vertvisc=orc.vertvisc_cs(gg, Kv=1.0e-3, Hbbl=10.0, Hmix=20.0, Kvml_invZ2=1.0e-3), visc=visc,
                      hor_visc=orc.hor_visc_cs(gg, 900.0, biharmonic=True, Smagorinsky_Ah=True, Smag_bi_const=0.06, Ah_vel_scale=0.01))
        st = orc.DynState(gg, dd["u"], dd["v"], dd["h"], dd["T"], dd["S"], 900.0, **kw)
        st.bcs.dtbt = 900.0 / 6.6      # (the synthetic cells are hundreds of km wide: DTBT would exceed DT)
        for n in range(nsteps):
            st.step(tt[0], tt[1])
        out.append(st)
    return g, g2, out[0], out[1]


@pytest.mark.parametrize("case", [dict(viscous=False), dict(viscous=True), dict(viscous=False, rk2b=True), dict(viscous=True, use_bt_cont=False)],
                         ids=["inviscid", "viscous", "rk2b", "viscous-no_bt_cont"])
def test_oracle_folded_run_is_the_unfolded_run(case):
    g, g2, a, b = run_pair(3, **case)
    assert a.bcs.nstep_last == b.bcs.nstep_last and a.bcs.dtbt == b.bcs.dtbt and a.bcs.nstep_last > 3
    assert np.abs(a.u).max() > 1e-3 and np.abs(a.u).max() < 3.0
    for n, pos in (("h", H), ("u", U), ("v", V), ("uh", U), ("vh", V), ("uhtr", U), ("vhtr", V), ("eta_av", H)):
        x = interior(g, getattr(a, n), pos); y = interior(g, tp.folded(g, getattr(b, n), pos), pos)
        assert bits_equal(np.ascontiguousarray(x), np.ascontiguousarray(y)), (n, float(np.abs(x - y).max()), np.argwhere(x != y)[:3])
    # the flow crosses the fold: the meridional transport on the fold line is not zero
    sj, si = g.csl(V)
    assert np.abs(a.vh[:, sj, si][:, -1]).max() > 0
