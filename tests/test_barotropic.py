"""MOM_barotropic: the CPU oracle (oracle/barotropic.c) against the invariants of the algorithm -- the reference
holds no known-answer vectors for btstep (PARITY UNPINNED) -- and, on the GPU, the HIP path against the oracle."""
import math

import numpy as np
import pytest

from helpers import barotropic_case, bits_equal, btstep_weights, interior
from mom6_amd import _abi
from oracle import orc


def run_oracle(g, cs, case, **kw):
    c = {k: v for k, v in case.items() if k not in ("dt",)}
    dt = case["dt"]
    args = dict(U_in=c["U_in"], V_in=c["V_in"], eta_in=c["eta_in"], dt=dt, bc_accel_u=c["bc_accel_u"], bc_accel_v=c["bc_accel_v"],
                taux=c["taux"], tauy=c["tauy"], pbce=c["pbce"], eta_PF_in=c["eta_PF_in"], U_Cor=c["U_Cor"], V_Cor=c["V_Cor"],
                visc_rem_u=c["visc_rem_u"], visc_rem_v=c["visc_rem_v"], bt_cont=c["bt_cont"], uh0=c["uh0"], vh0=c["vh0"],
                u_uh0=c["u_uh0"], v_vh0=c["v_vh0"])
    args.update(kw)
    return orc.btstep(g, cs, **args)


def test_cr_pow_is_the_correctly_rounded_power():
    """bt_rem = av_rem ** (1/nstep): orc_cr_pow equals libm's pow except where libm is not correctly rounded
    (glibc documents < 1 ULP, not correct rounding); the disagreements are 1 ulp and rare."""
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(0.5, 1, 20000), 10 ** rng.uniform(-8, 0, 20000)])
    y = 1.0 / rng.integers(1, 120, x.size)
    mine = np.array([orc.cr_pow(a, b) for a, b in zip(x, y)])
    ref = np.array([math.pow(a, b) for a, b in zip(x, y)])
    d = np.abs(mine.view(np.int64) - ref.view(np.int64))
    assert d.max() <= 1 and np.mean(d != 0) < 5e-3
    assert orc.cr_pow(1.0, 0.25) == 1.0 and orc.cr_pow(0.0625, 0.25) == 0.5 and orc.cr_pow(0.25, 0.5) == 0.5
    try:
        import mpmath as mp
    except ImportError:
        return
    mp.mp.prec = 300
    for i in list(np.nonzero(d)[0][:50]) + list(range(300)):
        assert float(mp.power(mp.mpf(float(x[i])), mp.mpf(float(y[i])))) == mine[i]


def test_btcalc_weights_sum_to_the_mask():
    for scheme in ("FROM_BT_CONT", "HARMONIC", "ARITHMETIC", "HYBRID"):
        g, cs, case, keep = barotropic_case(orc, hvel_scheme=scheme)
        fu, fv = keep["cs_arrs"]["frhatu"], keep["cs_arrs"]["frhatv"]
        su, sv = interior(g, fu.sum(0), _abi.POS_U), interior(g, fv.sum(0), _abi.POS_V)
        mu, mv = interior(g, g.mask2dCu, _abi.POS_U), interior(g, g.mask2dCv, _abi.POS_V)
        assert np.all(fu >= 0) and np.all(fv >= 0)
        assert np.allclose(su[mu > 0], 1.0, atol=1e-12) and np.all(su[mu == 0] == 0)
        assert np.allclose(sv[mv > 0], 1.0, atol=1e-12) and np.all(sv[mv == 0] == 0)


def test_set_dtbt_is_the_gravity_wave_limit():
    g, cs, case, keep = barotropic_case(orc)
    # dtbt_max ~ 1/sqrt(g H (1/dx^2 + 1/dy^2)) up to the (1+2 bebt) and face-area factors
    H = g.bathyT.max()
    dx = interior(g, g.dxT).min(); dy = interior(g, g.dyT).min()
    est = 1.0 / math.sqrt(g.g_Earth * H * (1 / dx ** 2 + 1 / dy ** 2))
    assert 0.2 * est < cs.dtbt_max < 5.0 * est
    assert cs.dtbt <= 0.98 * cs.dtbt_max


def test_ocean_at_rest_stays_at_rest():
    g, cs, case, keep = barotropic_case(orc, rest=True, land_frac=0.0)
    assert np.abs(case["eta_in"]).max() < 1e-9
    out = run_oracle(g, cs, case, want_etaav=True)
    # the PGF of a homogeneous flat ocean and its barotropic part cancel to roundoff
    assert np.abs(out["uhbtav"]).max() < 1e-3 and np.abs(out["vhbtav"]).max() < 1e-3      # m3/s on ~1e9 m2 faces
    assert np.abs(interior(g, out["eta_out"])).max() < 1e-9
    assert np.abs(out["accel_layer_u"]).max() < 1e-12 and np.abs(out["accel_layer_v"]).max() < 1e-12


@pytest.mark.parametrize("use_bt_cont", [True, False])
@pytest.mark.parametrize("topo", [(True, False), (False, False), (True, True)])
def test_barotropic_mass_budget(use_bt_cont, topo):
    """eta_out - eta_in = n_eff * eta_src - dt_eff * IareaT * div(uhbtav, vhbtav): the time-filtered free surface and
    the time-filtered transports that continuity() will be asked to match describe the same mass budget
    (MOM_barotropic.F90:1770-1808 builds wt_trans from wt_eta for exactly this)."""
    g, cs, case, keep = barotropic_case(orc, use_bt_cont=use_bt_cont, reentrant_x=topo[0], reentrant_y=topo[1])
    out = run_oracle(g, cs, case, want_etaav=True)
    w = btstep_weights(cs, case["dt"])
    assert cs.nstep_last == w["nstep"] and w["nstep"] > 3
    sj, si = g.csl(_abi.POS_H)
    uh, vh = out["uhbtav"], out["vhbtav"]
    div = (uh[sj, si.start + 1:si.stop + 1] - uh[sj, si.start:si.stop]) + (vh[sj.start + 1:sj.stop + 1, si] - vh[sj.start:sj.stop, si])
    eta_src = (g.mask2dT * (keep["cs_arrs"]["eta_cor"] / w["nstep"]))[sj, si]
    lhs = out["eta_out"][sj, si] - case["eta_in"][sj, si]
    rhs = w["n_eff"] * eta_src - w["dt_eff"] * g.IareaT[sj, si] * div
    scale = np.abs(w["dt_eff"] * g.IareaT[sj, si] * np.abs(uh[sj, si.start:si.stop])).max() + 1e-3
    assert np.abs(lhs - rhs).max() < 1e-10 * max(scale, 1.0)
    assert w["dt_eff"] == pytest.approx(case["dt"], rel=0.2)
    # land stays dry, the filtered eta is finite everywhere
    assert np.all(np.isfinite(out["eta_out"])) and np.all(np.isfinite(out["accel_layer_u"]))
    assert np.all(out["uhbtav"][g.mask2dCu == 0] == 0) and np.all(out["vhbtav"][g.mask2dCv == 0] == 0)


def test_btstep_options_change_the_answer_where_they_should():
    g, cs, case, keep = barotropic_case(orc)
    base = run_oracle(g, cs, case)
    g2, cs2, case2, _k = barotropic_case(orc, strong_drag=1)
    sd = run_oracle(g2, cs2, case2)
    assert not bits_equal(base["uhbtav"], sd["uhbtav"])
    no_uh0 = run_oracle(g, cs, case, uh0=None, vh0=None, u_uh0=None, v_vh0=None)
    assert not bits_equal(base["uhbtav"], no_uh0["uhbtav"])
    # accel_layer differs between layers only through pbce - gtot
    a = base["accel_layer_u"]
    assert np.abs(a[0] - a[-1]).max() > 0
