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


def test_nonlinear_bt_continuity():
    """NONLINEAR_BT_CONTINUITY (.testing/tc1, tc2; MOM_barotropic.F90:752, :1137, :1852, :2871): without a BT_cont the open face
    areas follow the free surface.  (a) With a BT_cont argument the switch is ignored by btstep.  (b) NONLIN_BT_CONT_UPDATE_PERIOD
    = 0 and a flat initial surface give the face areas of the linear form, hence its bits.  (c) With updates every step (the
    stencil of the wide-halo march is 2 then) the answer changes a little, and the mass budget
    d(eta) = n_eff eta_src - dt_eff div(uhbtav) still closes.  (d) set_dtbt with eta uses the harmonic-mean depths."""
    g, cs, case, keep = barotropic_case(orc)
    g2, cs2, case2, _ = barotropic_case(orc, Nonlinear_continuity=1)
    a, b = run_oracle(g, cs, case), run_oracle(g2, cs2, case2)
    for n in ("uhbtav", "eta_out", "accel_layer_u"):
        assert bits_equal(a[n], b[n])                                                             # (a)
    g, cs, case, keep = barotropic_case(orc, use_bt_cont=False)
    flat = dict(case, eta_in=np.zeros_like(case["eta_in"]))
    lin = run_oracle(g, cs, flat, want_etaav=True)
    g0, cs0, case0, _ = barotropic_case(orc, use_bt_cont=False, Nonlinear_continuity=1, Nonlin_cont_update_period=0)
    cs0.dtbt = cs.dtbt
    p0 = run_oracle(g0, cs0, dict(case0, eta_in=flat["eta_in"]), want_etaav=True)
    for n in ("uhbtav", "vhbtav", "eta_out", "etaav", "accel_layer_u"):
        assert bits_equal(lin[n], p0[n]), n                                                       # (b)
    g1, cs1, case1, _ = barotropic_case(orc, use_bt_cont=False, Nonlinear_continuity=1)
    cs1.dtbt = cs.dtbt
    base = run_oracle(g, cs, case, want_etaav=True)
    p1 = run_oracle(g1, cs1, case1, want_etaav=True)
    d = np.abs(interior(g, p1["eta_out"]) - interior(g, base["eta_out"])).max()
    assert 0 < d < 0.05 * np.abs(interior(g, base["eta_out"])).max()                               # (c)
    w = btstep_weights(cs1, case1["dt"])
    sj, si = g.csl(_abi.POS_H)
    div = (p1["uhbtav"][sj, si.start + 1:si.stop + 1] - p1["uhbtav"][sj, si.start:si.stop]) + \
          (p1["vhbtav"][sj.start + 1:sj.stop + 1, si] - p1["vhbtav"][sj.start:sj.stop, si])
    eta_src = keep["cs_arrs"]["eta_cor"][sj, si] * 0   # (eta_cor enters through the source; the budget below uses the outputs only)
    m = g.mask2dT[sj, si] > 0
    # volume conservation over the basin: the area integral of d(eta_wtd) equals the filtered source minus nothing (closed or periodic basin)
    assert abs((div * m).sum()) < 1e-6 * np.abs(div).sum()
    e0 = np.zeros_like(case["eta_in"]); e1 = np.ascontiguousarray(case["eta_in"] + 50.0 * g.mask2dT)
    orc.halo_update(g, e1, _abi.POS_H)
    d0 = orc.set_dtbt(g1, cs1, pbce=case1["pbce"], eta=e0); d1 = orc.set_dtbt(g1, cs1, pbce=case1["pbce"], eta=e1)
    dlin = orc.set_dtbt(g, cs, pbce=case["pbce"])
    assert d1 < d0 and abs(d0 - dlin) < 0.2 * dlin and d0 != dlin                                  # (d) deeper water, faster waves


# ---- GPU: the HIP path against the oracle, bit for bit ----------------------------------------------------------
def _gpu_cs(g, cs_o, dg, device, hvel_scheme, **kw):
    """A barotropic_CS of the product with the oracle CS's parameters; state arrays filled by the product's own
    barotropic_init / btcalc / bt_mass_source / set_dtbt so that those are checked on the way."""
    from mom6_amd.barotropic import barotropic_init
    return barotropic_init(dg, device=device, BT_THICK_SCHEME=hvel_scheme, **kw)


def _to(device, a):
    import torch
    if a is None:
        return None
    return a.copy() if device == "cpu" else torch.from_numpy(a).cuda()


def _np(a):
    return a if isinstance(a, np.ndarray) else a.cpu().numpy()


BT_CASES = [
    dict(),                                                     # defaults: BT_cont fits, layer fluxes, pow drag
    dict(use_bt_cont=False),                                    # linear face areas, BT_THICK_SCHEME = HARMONIC
    dict(strong_drag=1),
    dict(reentrant_x=False),
    dict(reentrant_x=True, reentrant_y=True),
    dict(linearized_BT_PV=0),
    dict(Sadourny=0),
    dict(adjust_BT_cont=1),
    dict(visc_rem_u_uh0=1, vel_underflow=1e-9),
    dict(ni=70, nj=9, nk=3, seed=9),
    dict(hvel_scheme="HYBRID"), dict(hvel_scheme="ARITHMETIC"),
    dict(use_bt_cont=False, Nonlinear_continuity=1),                                  # NONLINEAR_BT_CONTINUITY, areas refreshed every step
    dict(use_bt_cont=False, Nonlinear_continuity=1, Nonlin_cont_update_period=3, reentrant_y=True),
    dict(use_bt_cont=False, Nonlinear_continuity=1, Nonlin_cont_update_period=0),
    dict(Nonlinear_continuity=1),                                                     # with a BT_cont: ignored by btstep
]


@pytest.mark.gpu
@pytest.mark.parametrize("space", ["device", "host"])
@pytest.mark.parametrize("kw", BT_CASES, ids=[",".join(f"{k}={v}" for k, v in c.items()) or "default" for c in BT_CASES])
def test_btstep_matches_oracle_bitwise(kw, space):
    import torch
    from mom6_amd.barotropic import barotropic_init, bt_mass_source, btcalc, btstep, set_dtbt
    from mom6_amd.continuity import BT_cont_type
    from mom6_amd.tracer_advect import DeviceGrid
    if space == "host" and kw not in (dict(), dict(use_bt_cont=False)):
        pytest.skip("the staged path is covered on two cases")
    device = "cuda" if space == "device" else "cpu"
    g, cs_o, case, keep = barotropic_case(orc, **kw)
    dg = DeviceGrid(g)
    T = lambda a: _to(device, a)
    cs_kw = {k: v for k, v in kw.items() if k in ("strong_drag", "linearized_BT_PV", "Sadourny", "adjust_BT_cont", "visc_rem_u_uh0",
                                                  "vel_underflow")}
    if "Nonlinear_continuity" in kw:
        cs_kw.update(NONLINEAR_BT_CONTINUITY=bool(kw["Nonlinear_continuity"]), NONLIN_BT_CONT_UPDATE_PERIOD=kw.get("Nonlin_cont_update_period", 1))
    use_bt = kw.get("use_bt_cont", True)
    scheme = kw.get("hvel_scheme") or ("FROM_BT_CONT" if use_bt else "HARMONIC")
    CS = barotropic_init(dg, device=device, BT_THICK_SCHEME=scheme, USE_BT_CONT_TYPE=True, **cs_kw)
    for n in ("IDatu", "IDatv", "q_D", "D_u_Cor", "D_v_Cor"):
        assert bits_equal(_np(CS.arrays[n]), keep["cs_arrs"][n]), n
    bt_arrs = {n: T(a) for n, a in keep["bt_arrs"].items()}
    BT = BT_cont_type(**bt_arrs)
    h = T(keep["h"])
    if scheme == "FROM_BT_CONT":
        btcalc(h, dg, CS, bt_arrs["h_u"], bt_arrs["h_v"])
    else:
        btcalc(h, dg, CS)
    assert bits_equal(_np(CS.frhatu), keep["cs_arrs"]["frhatu"]) and bits_equal(_np(CS.frhatv), keep["cs_arrs"]["frhatv"])
    bt_mass_source(h, T(case["eta_in"]), True, dg, CS)
    assert bits_equal(_np(CS.eta_cor), keep["cs_arrs"]["eta_cor"])
    if kw.get("Nonlinear_continuity") and not use_bt:      # set_dtbt with eta (:2871): the face areas of the nonlinear form
        want = orc.set_dtbt(g, cs_o, pbce=case["pbce"], eta=case["eta_in"])
        got = set_dtbt(dg, CS, eta=T(case["eta_in"]), pbce=T(case["pbce"]))
        assert got == want
        dt_keep = cs_o.dtbt
        orc.set_dtbt(g, cs_o, pbce=case["pbce"], bt_cont=None, gtot_est=g.g_Earth, SSH_add=10.0)
        cs_o.dtbt = dt_keep
    set_dtbt(dg, CS, pbce=T(case["pbce"]), BT_cont=BT if use_bt else None, gtot_est=g.g_Earth, SSH_add=10.0)
    assert CS.st.dtbt_max == cs_o.dtbt_max
    CS.st.dtbt = cs_o.dtbt

    ref = run_oracle(g, cs_o, case, want_etaav=True)
    out = dict(accel_layer_u=T(g.zeros3(_abi.POS_U)), accel_layer_v=T(g.zeros3(_abi.POS_V)), eta_out=T(g.zeros2(_abi.POS_H)),
               uhbtav=T(g.zeros2(_abi.POS_U)), vhbtav=T(g.zeros2(_abi.POS_V)), etaav=T(g.zeros2(_abi.POS_H)))
    c = {k: T(v) for k, v in case.items() if isinstance(v, np.ndarray)}
    btstep(c["U_in"], c["V_in"], c["eta_in"], case["dt"], c["bc_accel_u"], c["bc_accel_v"], (c["taux"], c["tauy"]), c["pbce"],
           c["eta_PF_in"], c["U_Cor"], c["V_Cor"], out["accel_layer_u"], out["accel_layer_v"], out["eta_out"], out["uhbtav"],
           out["vhbtav"], dg, CS, c["visc_rem_u"], c["visc_rem_v"], BT_cont=BT if use_bt else None, uh0=c["uh0"], vh0=c["vh0"],
           u_uh0=c["u_uh0"], v_vh0=c["v_vh0"], etaav=out["etaav"])
    dg.sync()
    assert CS.st.nstep_last == cs_o.nstep_last
    for n in ("uhbtav", "vhbtav", "eta_out", "etaav", "accel_layer_u", "accel_layer_v"):
        a, b = _np(out[n]), ref[n]
        assert bits_equal(a, b), (n, float(np.abs(a - b).max()), int((a != b).sum()))
    assert bits_equal(_np(CS.ubtav), keep["cs_arrs"]["ubtav"]) and bits_equal(_np(CS.vbtav), keep["cs_arrs"]["vbtav"])
    dg.close()


FUSED_CASES = [BT_CASES[0], BT_CASES[1], BT_CASES[3], BT_CASES[4], BT_CASES[8], BT_CASES[9]]


@pytest.mark.gpu
@pytest.mark.parametrize("kw", FUSED_CASES, ids=[",".join(f"{k}={v}" for k, v in c.items()) or "default" for c in FUSED_CASES])
def test_btstep_with_the_fused_step_kernel_matches_oracle_bitwise(kw, monkeypatch):
    """MOM6HIP_BT_FUSED=1: one kernel a barotropic step with LDS-staged halo tiles and alternating field sets (bt_step_fused_kernel,
    round 5: bit-exact, slower than the four kernels -- profiles/r05_experiments.txt -- and therefore not the default)"""
    monkeypatch.setenv("MOM6HIP_BT_FUSED", "1")
    test_btstep_matches_oracle_bitwise(kw, "device")


@pytest.mark.gpu
def test_btstep_optional_arguments_gpu():
    """Predictor-style call: no etaav, eta_PF_start given, bottom stress given, no layer fluxes; eta_out aliases eta_in."""
    import torch
    from mom6_amd.barotropic import barotropic_init, bt_mass_source, btcalc, btstep
    from mom6_amd.continuity import BT_cont_type
    from mom6_amd.tracer_advect import DeviceGrid
    g, cs_o, case, keep = barotropic_case(orc, seed=11)
    rng = np.random.default_rng(3)
    eps = np.ascontiguousarray(case["eta_PF_in"] + 0.01 * rng.standard_normal(case["eta_PF_in"].shape) * g.mask2dT)
    tbx = np.ascontiguousarray(0.01 * rng.standard_normal(g.shape2(_abi.POS_U)) * g.mask2dCu)
    tby = np.ascontiguousarray(0.01 * rng.standard_normal(g.shape2(_abi.POS_V)) * g.mask2dCv)
    ref = run_oracle(g, cs_o, case, eta_PF_start=eps, taux_bot=tbx, tauy_bot=tby, uh0=None, vh0=None, u_uh0=None, v_vh0=None)
    dg = DeviceGrid(g)
    T = lambda a: torch.from_numpy(a).cuda()
    CS = barotropic_init(dg, BT_THICK_SCHEME="FROM_BT_CONT")
    bt_arrs = {n: T(a) for n, a in keep["bt_arrs"].items()}
    BT = BT_cont_type(**bt_arrs)
    btcalc(T(keep["h"]), dg, CS, bt_arrs["h_u"], bt_arrs["h_v"])
    bt_mass_source(T(keep["h"]), T(case["eta_in"]), True, dg, CS)
    CS.st.dtbt = cs_o.dtbt
    c = {k: T(v) for k, v in case.items() if isinstance(v, np.ndarray)}
    alu, alv = T(g.zeros3(_abi.POS_U)), T(g.zeros3(_abi.POS_V))
    uhb, vhb = T(g.zeros2(_abi.POS_U)), T(g.zeros2(_abi.POS_V))
    eta = c["eta_in"].clone()
    btstep(c["U_in"], c["V_in"], eta, case["dt"], c["bc_accel_u"], c["bc_accel_v"], (c["taux"], c["tauy"]), c["pbce"], c["eta_PF_in"],
           c["U_Cor"], c["V_Cor"], alu, alv, eta, uhb, vhb, dg, CS, c["visc_rem_u"], c["visc_rem_v"], BT_cont=BT, eta_PF_start=T(eps),
           taux_bot=T(tbx), tauy_bot=T(tby))
    dg.sync()
    sj, si = g.csl(_abi.POS_H)
    assert bits_equal(eta.cpu().numpy()[sj, si], ref["eta_out"][sj, si])
    assert bits_equal(uhb.cpu().numpy(), ref["uhbtav"]) and bits_equal(alv.cpu().numpy(), ref["accel_layer_v"])
    dg.close()


@pytest.mark.gpu
def test_btstep_refuses_what_it_does_not_provide():
    from mom6_amd._lib import Mom6HipError
    from mom6_amd.barotropic import barotropic_init
    from mom6_amd.tracer_advect import DeviceGrid
    g, cs_o, case, keep = barotropic_case(orc)
    dg = DeviceGrid(g)
    with pytest.raises(Mom6HipError, match="INTEGRAL_BT_CONTINUITY"):
        barotropic_init(dg, INTEGRAL_BT_CONTINUITY=True)
    dg.close()
