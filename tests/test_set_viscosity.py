"""set_viscous_BBL / set_viscous_ML (second half): CPU checks of the oracle (oracle/set_viscosity.c) and GPU
parity of libmom6hip against it (bit-exact fp64).  The reference holds no known-answer vectors for MOM_set_viscosity (parity
unpinned)."""
import numpy as np
import pytest

import exact_synth as xs
from helpers import bits_equal, interior
from mom6_amd import _abi
from oracle import orc

H, U, V = _abi.POS_H, _abi.POS_U, _abi.POS_V
VARIANTS = {
    "default": dict(),
    "rlay": dict(BBL_use_EOS=False),
    "linear_drag": dict(linear_drag=True, drag_bg_vel=0.1),
    "bg_vel": dict(drag_bg_vel=0.05, BBL_thick_min=0.5),
    "body_force": dict(body_force_drag=True),
    "correct_bounds": dict(correct_BBL_bounds=True, BBL_thick_min=0.1, Kv_BBL_min=2.0e-3),
    "rino_mix": dict(RiNo_mix=True),
    # CHANNEL_DRAG (.testing/tc1, tc2 with USE_JACKSON_PARAM, DRAG_BG_VEL = 0.1, BBL_THICK_MIN = 0.1)
    "channel_tc": dict(Channel_drag=True, RiNo_mix=True, drag_bg_vel=0.1, BBL_thick_min=0.1),
    "channel": dict(Channel_drag=True),
    "channel_iterative": dict(Channel_drag=True, concave_trigonometric_L=False, BBL_thick_min=0.1),
    "channel_body_force": dict(Channel_drag=True, body_force_drag=True, drag_bg_vel=0.05),
    "channel_bounds": dict(Channel_drag=True, correct_BBL_bounds=True, BBL_thick_min=0.1, Kv_BBL_min=2.0e-3, Chan_drag_max_vol=3.0, c_Smag=0.06),
}
REF = dict(BBL_use_EOS="BBL_USE_EOS", linear_drag="LINEAR_DRAG", drag_bg_vel="DRAG_BG_VEL", BBL_thick_min="BBL_THICK_MIN",
           body_force_drag="DRAG_AS_BODY_FORCE", correct_BBL_bounds="CORRECT_BBL_BOUNDS", Kv_BBL_min="KV_BBL_MIN", RiNo_mix="USE_JACKSON_PARAM",
           Channel_drag="CHANNEL_DRAG", concave_trigonometric_L="TRIG_CHANNEL_DRAG_WIDTHS", Chan_drag_max_vol="CHANNEL_DRAG_MAX_BBL_THICK",
           c_Smag="SMAG_CONST_CHANNEL")


def visc_arrays(g, d):
    su, sv = g.shape2(U), g.shape2(V)
    return dict(Kv_bbl_u=np.zeros(su), Kv_bbl_v=np.zeros(sv), bbl_thick_u=np.zeros(su), bbl_thick_v=np.zeros(sv),
                Ray_u=np.zeros_like(d["u"]), Ray_v=np.zeros_like(d["v"]))


def rlay(nk):
    return 1025.0 + 0.5 * np.arange(nk)


def run_oracle(g, d, eos_form="WRIGHT", **kw):
    arrs = visc_arrays(g, d)
    visc = orc.vertvisc_type(**arrs)
    if not kw.get("BBL_use_EOS", True):
        kw = dict(kw, Rlay=rlay(g.nk))
    cs = orc.set_visc_cs(g, 10.0, 1.0e-4, **kw)
    orc.set_viscous_BBL(g, cs, d["u"], d["v"], d["h"], d["T"], d["S"], orc.eos(eos_form), visc)
    return visc._keep


def test_bbl_is_sane_and_scales_with_the_flow():
    g = xs.make_grid(40, 28, 12)
    d = xs.make_state(g, umax=0.3)
    a = run_oracle(g, d)
    mu = interior(g, g.mask2dCu, U) > 0
    bt, kv = interior(g, a["bbl_thick_u"], U)[mu], interior(g, a["Kv_bbl_u"], U)[mu]
    assert np.all(np.isfinite(bt)) and bt.min() >= 0.0 and bt.max() < 200.0
    assert kv.min() >= 1.0e-4 and kv.max() < 1.0      # never below KV_BBL_MIN
    # land faces are untouched
    assert np.all(interior(g, a["bbl_thick_u"], U)[~mu] == 0.0)
    # a faster flow has a larger friction velocity: kv = cdrag_sqrt*ustar*bbl_thick grows
    d2 = dict(d, u=d["u"] * 3.0, v=d["v"] * 3.0)
    b = run_oracle(g, d2)
    assert interior(g, b["Kv_bbl_u"], U)[mu].mean() > 1.5 * kv.mean()


def test_linear_drag_gives_the_analytic_ustar():
    """LINEAR_DRAG: ustar = sqrt(cdrag)*DRAG_BG_VEL everywhere; with an unstratified column (the layer densities all equal)
    and f = 0 the layer reaches the surface: bbl_thick = the column's thickness at the face (:815-822 with C2f = 0)"""
    g = xs.make_grid(24, 16, 6, land_frac=0.0, flat_bottom=True, max_depth=600.0, beta_plane=True)
    g.CoriolisBu[:] = 0.0
    g._struct = None
    d = xs.make_state(g, umax=0.1, vanish_frac=0.0)
    arrs = visc_arrays(g, d)
    visc = orc.vertvisc_type(**arrs)
    cs = orc.set_visc_cs(g, 10.0, 1.0e-4, linear_drag=True, drag_bg_vel=0.1, BBL_use_EOS=False, Rlay=np.full(6, 1030.0))
    orc.set_viscous_BBL(g, cs, d["u"], d["v"], d["h"], d["T"], d["S"], None, visc)
    a = visc._keep
    sj, si = g.csl(U)
    ustar = np.sqrt(0.003) * 0.1
    bt = a["bbl_thick_u"][sj, si]
    assert np.allclose(a["Kv_bbl_u"][sj, si], np.sqrt(0.003) * ustar * bt, rtol=1e-14)
    tot = d["h"].sum(0)
    col = 0.5 * (tot[:, :-1] + tot[:, 1:])[g.csl(H)[0], g.halo - 1:g.halo + g.ni]
    assert np.all(np.abs(bt - col) < 0.02 * col)      # (the upwind-biased harmonic mean differs a little from the arithmetic one)


def ml_inputs(g, d, seed=3):
    """forces%taux / tauy / ustar and the visc arrays set_viscous_ML reads and writes"""
    rng = np.random.default_rng(seed)
    su, sv, sh = g.shape2(U), g.shape2(V), g.shape2(H)
    taux = np.ascontiguousarray(0.1 * np.cos(np.linspace(0, 3, su[0]))[:, None] * g.mask2dCu)
    tauy = np.ascontiguousarray(0.05 * (rng.random(sv) - 0.3) * g.mask2dCv)
    ustar = np.ascontiguousarray(0.004 + 0.008 * rng.random(sh))
    return taux, tauy, dict(ustar=ustar, nkml_visc_u=np.full(su, -1.0), nkml_visc_v=np.full(sv, -1.0))


ML_VARIANTS = {
    "tc_values": dict(bulk_Ri_ML=0.05, TKE_decay=10.0, omega_frac=1.0),              # .testing/tc1, tc2: BULK_RI_ML, TKE_DECAY, ML_OMEGA_FRAC
    "wright": dict(bulk_Ri_ML=2.0, TKE_decay=1.0),                                   # (the rest: values that spread the answers over the layers)
    "omega_frac_1": dict(bulk_Ri_ML=2.0, TKE_decay=1.0, omega_frac=1.0),
    "omega_blend": dict(bulk_Ri_ML=4.0, TKE_decay=0.5, omega_frac=0.4),
    "nkml2": dict(bulk_Ri_ML=2.0, TKE_decay=1.0, omega_frac=1.0, nkml=2),            # a bulk mixed layer (tc1)
    "rlay": dict(bulk_Ri_ML=2.0, TKE_decay=1.0, eos=None),                           # no equation of state: GV%Rlay
    "no_decay": dict(bulk_Ri_ML=0.5, TKE_decay=0.0),
}
ML_DT = 7200.0


def run_oracle_ml(g, d, taux, tauy, arrs, dt=ML_DT, eos="WRIGHT", **kw):
    visc = orc.vertvisc_type(**arrs)
    cs = orc.set_visc_cs(g, 10.0, 1.0e-4, dynamic_viscous_ML=True, Rlay=rlay(g.nk) if eos is None else None, **kw)
    orc.set_viscous_ML(g, cs, d["u"], d["v"], d["h"], d["T"], d["S"], None if eos is None else orc.eos(eos), taux, tauy, visc, dt)
    return visc._keep


def test_set_viscous_ML_without_the_switch_and_refusals():
    """without DYNAMIC_VISCOUS_ML set_viscous_ML returns at once (MOM_set_viscosity.F90:2043); BBL_USE_TIDAL_BG is refused"""
    g = xs.make_grid(12, 10, 3)
    d = xs.make_state(g)
    taux, tauy, arrs = ml_inputs(g, d)
    visc = orc.vertvisc_type(**arrs)
    orc.set_viscous_ML(g, orc.set_visc_cs(g, 10.0, 1.0e-4), d["u"], d["v"], d["h"], d["T"], d["S"], orc.eos("WRIGHT"), taux, tauy, visc, 900.0)
    assert np.all(visc._keep["nkml_visc_u"] == -1.0)
    cs3 = orc.set_visc_cs(g, 10.0, 1.0e-4, BBL_use_tidal_bg=True)
    with pytest.raises(RuntimeError):
        orc.set_viscous_BBL(g, cs3, d["u"], d["v"], d["h"], d["T"], d["S"], orc.eos("WRIGHT"), orc.vertvisc_type(**visc_arrays(g, d)))


def test_cr_exp_is_the_correctly_rounded_exponential():
    """the decay of the bulk Richardson number, exp(-htot*Idecay_len_TKE) (:2178), is evaluated correctly rounded on both sides:
    against 200-bit arithmetic it is exact; glibc's exp (< 1 ulp, not correctly rounded) differs from it rarely and by one ulp"""
    import math
    mp = pytest.importorskip("mpmath")
    mp.mp.prec = 200
    rng = np.random.default_rng(2)
    ts = np.concatenate([-rng.random(3000) * 30.0, -10.0 ** rng.uniform(-12, 2.8, 3000), [-0.0, -1e-300, -699.9, -745.0, -1000.0]])
    nlibm = 0
    for t in ts:
        want = float(mp.exp(mp.mpf(float(t)))) if t > -700.0 else 0.0
        got = orc.cr_exp(t)
        assert got == want, (t, got, want)
        nlibm += got != math.exp(t) and t > -700.0
    assert nlibm < 0.01 * ts.size


def test_cr_acos_and_cr_cos_are_correctly_rounded():
    """find_L_open_concave_trigonometric (:1213-1225) calls cos(acos(x)/3 - 2 pi/3): both functions are evaluated correctly rounded
    on both sides (oracle/cr_trig.c, csrc/cr_math.hpp): exact against 300-bit arithmetic; libm differs from them rarely, by one ulp"""
    import math
    mp = pytest.importorskip("mpmath")
    mp.mp.prec = 300
    rng = np.random.default_rng(5)
    xa = np.concatenate([rng.uniform(-1, 1, 2000), 1 - 10.0 ** rng.uniform(-16, 0, 800), -1 + 10.0 ** rng.uniform(-16, 0, 800),
                         10.0 ** rng.uniform(-300, -1, 100), [0.0, 1.0, -1.0, 0.5, -0.5]])
    nl = 0
    for x in xa:
        got = orc.cr_acos(x)
        assert got == float(mp.acos(mp.mpf(float(x)))), x
        nl += got != math.acos(x)
    assert nl < 0.01 * xa.size
    xc = np.concatenate([rng.uniform(-math.pi, math.pi, 2000), rng.uniform(-2.0944, -1.0472, 2000), 10.0 ** rng.uniform(-300, 0, 100),
                         [0.0, math.pi / 2, -math.pi / 2, math.pi]])
    nl = 0
    for x in xc:
        got = orc.cr_cos(x)
        assert got == float(mp.cos(mp.mpf(float(x)))), x
        nl += got != math.cos(x)
    assert nl < 0.01 * xc.size
    assert np.isnan(orc.cr_acos(1.0000001)) and np.isnan(orc.cr_cos(4.0))


def channel_geometry(g):
    """the sign of the bottom curvature across each u face (:872-889): 0 uniform slope, +1 concave, -1 convex"""
    D = 0.5 * (g.bathyT[:, :-1] + g.bathyT[:, 1:])      # D_u at I = isd .. ied-1 (columns 1 .. nih-1 of the u arrays)
    m = g.mask2dCu[:, 1:-1]
    Dc = D[1:-1]
    with np.errstate(invalid="ignore", divide="ignore"):
        tp = m[2:] * D[2:]; Dp = 2.0 * Dc * tp / (Dc + tp)
        tm = m[:-2] * D[:-2]; Dm = 2.0 * Dc * tm / (Dc + tm)
    lo, hi = np.minimum(Dp, Dm), np.maximum(Dp, Dm)
    crv = 3.0 * (hi + lo - 2.0 * Dc)
    kind = np.sign(np.where(np.abs(crv) < 1e-2 * ((hi - lo) + 0.1), 0.0, crv))
    out = np.full(g.shape2(U), np.nan)
    out[1:-1, 1:-1] = kind
    return out


@pytest.mark.parametrize("name", [n for n in VARIANTS if n.startswith("channel")])
def test_channel_drag_is_sane(name):
    """CHANNEL_DRAG (.testing/tc1, tc2; :863-1002): the Rayleigh drag is non-negative, zero on land and in layers that do not
    touch the sloping bottom; the fraction of the bottom drag left to the viscous boundary layer is at most one, so Kv_bbl
    does not exceed the value without the option; all three bottom shapes occur on the test grid"""
    kw = VARIANTS[name]
    g = xs.make_grid(40, 28, 12)
    d = xs.make_state(g, umax=0.3)
    a = run_oracle(g, d, **kw)
    plain = run_oracle(g, d, **{k: v for k, v in kw.items() if k not in ("Channel_drag", "concave_trigonometric_L", "Chan_drag_max_vol", "c_Smag")})
    mu = interior(g, g.mask2dCu, U) > 0
    Ru = interior(g, a["Ray_u"], U)
    assert np.all(np.isfinite(Ru)) and Ru.min() >= 0.0 and np.all(Ru[:, ~mu] == 0.0)
    assert Ru[:, mu].max() > 0.0 and (Ru[:, mu] > 0).mean() < 0.9
    if not kw.get("body_force_drag") and not kw.get("correct_BBL_bounds"):
        kv, kv0 = interior(g, a["Kv_bbl_u"], U)[mu], interior(g, plain["Kv_bbl_u"], U)[mu]
        assert np.all(kv <= kv0 * (1 + 1e-12)) and np.any(kv < 0.99 * kv0)
        assert bits_equal(a["bbl_thick_u"], plain["bbl_thick_u"])
    kinds = interior(g, channel_geometry(g), U)[mu]
    assert {-1.0, 1.0} <= set(np.unique(kinds))      # concave and convex bottoms (a uniform slope: the next test)


def sloping_grid(ni=24, nj=16, nk=6, flat=False, **kw):
    """a bottom that is a plane (flat: level): CHANNEL_DRAG takes find_L_open_uniform_slope everywhere"""
    g = xs.make_grid(ni, nj, nk, land_frac=0.0, flat_bottom=True, max_depth=600.0, **kw)
    if not flat:
        jj, ii = np.meshgrid(np.arange(g.bathyT.shape[0]), np.arange(g.bathyT.shape[1]), indexing="ij")
        g.bathyT[:] = 300.0 + 2.0 * jj + 1.0 * ii
        g._struct = None
    return g


def test_channel_drag_over_a_plane_bottom():
    """a level bottom: every interface is fully open, the only step in L is at the bottom itself, where the whole drag goes to the
    viscous boundary layer: no Rayleigh drag and the Kv_bbl of the plain scheme.  A tilted plane: uniform-slope widths, some drag"""
    g = sloping_grid(flat=True)
    d = xs.make_state(g, umax=0.2, vanish_frac=0.0)
    a, plain = run_oracle(g, d, Channel_drag=True), run_oracle(g, d)
    # (next to the closed northern and southern walls the neighbouring face is land: its depth counts as zero, :873-876)
    assert np.all(interior(g, a["Ray_u"], U)[:, 1:-1] == 0.0) and interior(g, a["Ray_u"], U)[:, 0].max() > 0.0
    assert bits_equal(interior(g, a["Kv_bbl_u"], U)[1:-1], interior(g, plain["Kv_bbl_u"], U)[1:-1])
    g = sloping_grid()
    assert set(np.unique(interior(g, channel_geometry(g), U)[1:-1])) == {0.0}
    d = xs.make_state(g, umax=0.2, vanish_frac=0.0)
    a, plain = run_oracle(g, d, Channel_drag=True), run_oracle(g, d)
    Ru = interior(g, a["Ray_u"], U)
    assert Ru.max() > 0.0 and Ru.min() >= 0.0
    assert np.all(interior(g, a["Kv_bbl_u"], U) <= interior(g, plain["Kv_bbl_u"], U))


def test_channel_drag_trigonometric_and_iterative_widths_agree():
    """the reference's own debugging check (:900-925): the two concave solutions are mathematically equivalent; here their Rayleigh
    drags agree to a relative 1e-8 where there is any"""
    g = xs.make_grid(40, 28, 12)
    d = xs.make_state(g, umax=0.3)
    a = run_oracle(g, d, Channel_drag=True, BBL_thick_min=0.1)
    b = run_oracle(g, d, Channel_drag=True, BBL_thick_min=0.1, concave_trigonometric_L=False)
    for n in ("Ray_u", "Ray_v", "Kv_bbl_u", "Kv_bbl_v"):
        assert np.allclose(a[n], b[n], rtol=1e-8, atol=1e-14), n
    assert not bits_equal(a["Ray_u"], b["Ray_u"])


@pytest.mark.parametrize("name", list(ML_VARIANTS))
def test_dynamic_viscous_ML_is_sane(name):
    """DYNAMIC_VISCOUS_ML (.testing/tc1, tc2; :2111-2230, :2400-2506): the fractional number of layers in the viscous mixed layer is
    between nkml and nz on ocean faces and nkml on land; a more efficient conversion (a larger bulk Richardson number) cannot
    make the layer shallower; with no conversion at all (BULK_RI_ML = 0) the search stops at the first stratified layer."""
    kw = dict(ML_VARIANTS[name])
    g = xs.make_grid(40, 28, 12)
    d = xs.make_state(g, umax=0.3)
    taux, tauy, arrs = ml_inputs(g, d)
    a = run_oracle_ml(g, d, taux, tauy, arrs, **kw)
    nkml = kw.get("nkml", 0)
    for n, pos, mk in (("nkml_visc_u", U, g.mask2dCu), ("nkml_visc_v", V, g.mask2dCv)):
        x = interior(g, a[n], pos); m = interior(g, mk, pos) > 0
        assert np.all(x[~m] == nkml) and np.all(x[m] >= nkml) and np.all(x[m] <= g.nk)
        if name != "tc_values":
            assert np.any(x[m] != np.round(x[m])) and len(np.unique(np.ceil(x[m]))) >= 2      # columns end inside different layers
    _, _, arrs2 = ml_inputs(g, d)
    b = run_oracle_ml(g, d, taux, tauy, arrs2, **dict(kw, bulk_Ri_ML=10.0 * kw["bulk_Ri_ML"]))
    _, _, arrs3 = ml_inputs(g, d)
    c = run_oracle_ml(g, d, taux, tauy, arrs3, **dict(kw, bulk_Ri_ML=0.0))
    mu = interior(g, g.mask2dCu, U) > 0
    xa, xb, xc = (interior(g, q["nkml_visc_u"], U)[mu] for q in (a, b, c))
    assert np.all(xb >= xa) and np.all(xc <= xa) and np.all(xc == np.round(xc))
    if name != "tc_values":
        assert np.any(xb > xa)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(VARIANTS))
def test_gpu_parity(name):
    import torch
    from mom6_amd.pressure_force import EOS_init
    from mom6_amd.set_viscosity import set_visc_init, set_viscous_BBL, set_viscous_ML
    from mom6_amd.tracer_advect import DeviceGrid
    from mom6_amd.vert_friction import vertvisc_type
    kw = VARIANTS[name]
    for (ni, nj, nk, topo) in [(70, 21, 8, (True, False)), (44, 40, 2, (True, True)), (10, 8, 30, (False, False)), (200, 9, 75, (True, False))]:
        g = xs.make_grid(ni, nj, nk, reentrant_x=topo[0], reentrant_y=topo[1])
        d = xs.make_state(g, umax=0.3)
        ref = run_oracle(g, d, **kw)
        dg = DeviceGrid(g)
        pk = {REF[k]: v for k, v in kw.items()}
        if not kw.get("BBL_use_EOS", True):
            pk["Rlay"] = rlay(nk)
        CS = set_visc_init(dg, HBBL=10.0, KV=1.0e-4, **pk)
        for resident in (True, False):
            X = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if resident else (lambda a: a.copy())
            N = (lambda a: a.cpu().numpy()) if resident else (lambda a: a)
            arrs = {n: X(a) for n, a in visc_arrays(g, d).items()}
            visc = vertvisc_type(**arrs)
            set_viscous_BBL(X(d["u"]), X(d["v"]), X(d["h"]), (X(d["T"]), X(d["S"]), EOS_init("WRIGHT")), visc, dg, CS)
            set_viscous_ML(None, None, None, None, None, visc, 900.0, dg, CS)      # no DYNAMIC_VISCOUS_ML: the early return
            dg.sync()
            for n in ("bbl_thick_u", "bbl_thick_v", "Kv_bbl_u", "Kv_bbl_v", "Ray_u", "Ray_v"):
                assert bits_equal(N(arrs[n]), ref[n]), (name, (ni, nj, nk), resident, n, np.argwhere(N(arrs[n]) != ref[n])[:3])
        dg.close()


@pytest.mark.gpu
def test_gpu_parity_unesco_eos():
    """BBL_USE_EOS with EQN_OF_STATE = UNESCO: the density derivatives at the bottom pressure come from MOM_EOS_UNESCO."""
    import torch
    from mom6_amd.pressure_force import EOS_init
    from mom6_amd.set_viscosity import set_visc_init, set_viscous_BBL
    from mom6_amd.tracer_advect import DeviceGrid
    from mom6_amd.vert_friction import vertvisc_type
    g = xs.make_grid(70, 21, 20)
    d = xs.make_state(g, umax=0.3)
    ref = run_oracle(g, d, eos_form="UNESCO")
    assert not bits_equal(ref["bbl_thick_u"], run_oracle(g, d)["bbl_thick_u"])
    dg = DeviceGrid(g)
    CS = set_visc_init(dg, HBBL=10.0, KV=1.0e-4)
    X = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    arrs = {n: X(a) for n, a in visc_arrays(g, d).items()}
    set_viscous_BBL(X(d["u"]), X(d["v"]), X(d["h"]), (X(d["T"]), X(d["S"]), EOS_init("UNESCO")), vertvisc_type(**arrs), dg, CS)
    dg.sync()
    for n in ("bbl_thick_u", "bbl_thick_v", "Kv_bbl_u", "Kv_bbl_v", "Ray_u", "Ray_v"):
        assert bits_equal(arrs[n].cpu().numpy(), ref[n]), n
    dg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(ML_VARIANTS))
def test_gpu_parity_dynamic_viscous_ML(name):
    """set_viscous_ML with DYNAMIC_VISCOUS_ML: library == oracle, bit for bit (the exponential included)"""
    import torch
    from mom6_amd.pressure_force import EOS_init
    from mom6_amd.set_viscosity import set_visc_init, set_viscous_ML
    from mom6_amd.tracer_advect import DeviceGrid
    from mom6_amd.vert_friction import vertvisc_type
    kw = dict(ML_VARIANTS[name])
    eos = kw.pop("eos", "WRIGHT")
    for (ni, nj, nk, topo) in [(70, 21, 8, (True, False)), (44, 40, 3, (True, True)), (10, 8, 30, (False, False)), (200, 9, 75, (True, False))]:
        if kw.get("nkml", 0) >= nk:
            continue
        g = xs.make_grid(ni, nj, nk, reentrant_x=topo[0], reentrant_y=topo[1])
        d = xs.make_state(g, umax=0.3)
        taux, tauy, arrs = ml_inputs(g, d, seed=ni)
        ref = run_oracle_ml(g, d, taux, tauy, {n: a.copy() for n, a in arrs.items()}, eos=eos, **kw)
        dg = DeviceGrid(g)
        CS = set_visc_init(dg, HBBL=10.0, KV=1.0e-4, DYNAMIC_VISCOUS_ML=True, BULK_RI_ML=kw["bulk_Ri_ML"], TKE_DECAY=kw["TKE_decay"],
                           ML_OMEGA_FRAC=kw.get("omega_frac", 0.0), NKML=kw.get("nkml", 0), Rlay=rlay(nk) if eos is None else None)
        for resident in (True, False):
            X = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if resident else (lambda a: a.copy())
            N = (lambda a: a.cpu().numpy()) if resident else (lambda a: a)
            va = {n: X(a) for n, a in arrs.items()}
            visc = vertvisc_type(**va)
            tv = (X(d["T"]), X(d["S"]), EOS_init(eos)) if eos else None
            set_viscous_ML(X(d["u"]), X(d["v"]), X(d["h"]), tv, (X(taux), X(tauy)), visc, ML_DT, dg, CS)
            dg.sync()
            for n in ("nkml_visc_u", "nkml_visc_v"):
                assert bits_equal(N(va[n]), ref[n]), (name, (ni, nj, nk), resident, n, np.argwhere(N(va[n]) != ref[n])[:3])
        dg.close()
