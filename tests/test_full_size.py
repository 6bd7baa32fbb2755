"""The hot path at BASELINE.json's full sizes (benchmark 360x180x75, OM4_025-shaped 1440x1080x75), checked through
size-independent properties -- the oracle needs minutes at these sizes, the properties need none: conservation (tracer
mass, volume), bounds / monotonicity, uniform-field preservation, the barotropic transport match, the momentum budget of
the implicit viscous solve, idempotence of remapping onto the same grid.  Everything runs device-resident through the
C ABI; reductions are fp64 torch sums on the device (tolerances are those of the summation, not of the kernels)."""
import numpy as np
import pytest

from mom6_amd import _abi, synth

SIZES = [(360, 180, 75), (1440, 1080, 75)]
H, U, V = _abi.POS_H, _abi.POS_U, _abi.POS_V


def _inner(g, a, pos=H):
    sj, si = g.csl(pos)
    return a[..., sj, si]


@pytest.fixture(scope="module", params=SIZES, ids=[f"{a}x{b}x{c}" for a, b, c in SIZES])
def world(request):
    import torch
    from mom6_amd.tracer_advect import DeviceGrid
    ni, nj, nk = request.param
    g = synth.make_grid(ni, nj, nk, seed=20241020, rough_noise=0.0)
    dg = DeviceGrid(g)
    dyn = synth.make_dynamics_state(g, seed=11, device="cuda", umax=0.1, eta_amp=0.2, terrain_following=True)
    T = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64, device="cuda")
    w = dict(g=g, dg=dg, dyn=dyn, areaT=T(g.areaT), mT=T(g.mask2dT), mu=T(g.mask2dCu), mv=T(g.mask2dCv))
    yield w
    dg.close()
    torch.cuda.empty_cache()


@pytest.mark.gpu
def test_advect_tracer_conserves_and_stays_bounded(world):
    import torch
    from mom6_amd.tracer_advect import advect_tracer, tracer_advect_init
    g, dg = world["g"], world["dg"]
    st = synth.make_advection_state(g, ntr=4, seed=1, device="cuda", hot_frac=0.0)
    tr = [t.clone() for t in st["tr"]] + [torch.full_like(st["tr"][0], 3.25)]      # + a uniform tracer
    A = _inner(g, world["areaT"] * world["mT"])[None]
    # the tracer amount before: concentrations times the volume the scheme starts from, h_end + dt*div(uhtr, vhtr)
    sj, si = g.csl(H)
    uh, vh = st["uhtr"], st["vhtr"]
    div = (uh[:, sj, si.start + 1:si.stop + 1] - uh[:, sj, si.start:si.stop]) + (vh[:, sj.start + 1:sj.stop + 1, si] - vh[:, sj.start:sj.stop, si])
    vol0 = _inner(g, st["h_end"]) * A + div
    vol1 = _inner(g, st["h_end"]) * A
    before = [float((_inner(g, t) * vol0).sum()) for t in tr]
    lo = [float(_inner(g, t)[:, _inner(g, world["mT"]) > 0].min()) for t in tr]
    hi = [float(_inner(g, t)[:, _inner(g, world["mT"]) > 0].max()) for t in tr]
    stats = advect_tracer(st["h_end"], st["uhtr"], st["vhtr"], None, 3600.0, dg, tracer_advect_init(900.0, "PPM:H3"), tr)
    dg.sync()
    assert stats.iterations >= 1
    for m, t in enumerate(tr):
        after = float((_inner(g, t) * vol1).sum())
        assert abs(after - before[m]) <= 1e-9 * abs(before[m]), (m, before[m], after)
        x = _inner(g, t)[:, _inner(g, world["mT"]) > 0]
        assert float(x.min()) >= lo[m] - 1e-12 * abs(lo[m]) and float(x.max()) <= hi[m] + 1e-12 * abs(hi[m]), m
    # a uniform tracer stays uniform (flux form: to roundoff of (c*vol + c*in - c*out)/vol_new)
    assert float((_inner(g, tr[-1])[:, _inner(g, world["mT"]) > 0] - 3.25).abs().max()) <= 1e-12


@pytest.mark.gpu
def test_continuity_conserves_volume_and_matches_the_barotropic_transport(world):
    import torch
    from mom6_amd.continuity import continuity, continuity_PPM_init
    g, dg, d = world["g"], world["dg"], world["dyn"]
    cs = continuity_PPM_init(dg)
    h1 = d["h"].clone(); uh = torch.zeros_like(d["u"]); vh = torch.zeros_like(d["v"])
    continuity(d["u"], d["v"], d["h"], h1, uh, vh, 900.0, dg, cs)
    A = _inner(g, world["areaT"] * world["mT"])[None]
    v0, v1 = float((_inner(g, d["h"]) * A).sum()), float((_inner(g, h1) * A).sum())
    # flux form: what leaves one cell enters its neighbour; the closed N/S edges and the re-entrant E/W edge lose nothing
    # (only the Angstrom floor of vanished layers can add volume)
    assert abs(v1 - v0) <= 1e-10 * v0
    assert float(_inner(g, h1).min()) >= g.Angstrom_H
    # with uhbt the layer transports are adjusted until their sum matches it to ETA_TOLERANCE
    uhbt = (uh.sum(0) * 1.03).contiguous(); vhbt = (vh.sum(0) * 0.97).contiguous()
    ucor, vcor = torch.zeros_like(d["u"]), torch.zeros_like(d["v"])
    h2 = d["h"].clone()
    continuity(d["u"], d["v"], d["h"], h2, uh, vh, 900.0, dg, cs, uhbt=uhbt, vhbt=vhbt, u_cor=ucor, v_cor=vcor)
    sj, si = g.csl(U)
    err = (uh.sum(0) - uhbt)[sj, si].abs() * 900.0
    tol = 0.5 * g.nk * g.Angstrom_H * float(world["areaT"].max())      # dt*|uh_err| < tol_eta*area (:1162)
    assert float(err.max()) <= 2.0 * tol + 1e-9 * float(uhbt.abs().max()) * 900.0


@pytest.mark.gpu
def test_remapping_conserves_and_is_the_identity_on_the_same_grid(world):
    import torch
    from mom6_amd.ale import ALE_remap_tracers, initialize_remapping
    g, dg, d = world["g"], world["dg"], world["dyn"]
    R = initialize_remapping("PPM_H4")
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    w = torch.rand(d["h"].shape, generator=gen, dtype=torch.float64, device="cuda") * 0.2 + 0.9
    hn = (w * d["h"]); hn = (hn * (d["h"].sum(0, keepdim=True) / hn.sum(0, keepdim=True))).contiguous()
    Tn, Sn = d["T"].clone(), d["S"].clone()
    ALE_remap_tracers(R, dg, d["h"], hn, [Tn, Sn])
    m = _inner(g, world["mT"]) > 0
    for old, new in ((d["T"], Tn), (d["S"], Sn)):
        c0 = (_inner(g, old) * _inner(g, d["h"])).sum(0)[m]; c1 = (_inner(g, new) * _inner(g, hn)).sum(0)[m]
        assert float(((c1 - c0).abs() / c0.abs().clamp(min=1e-30)).max()) <= 1e-11      # column inventories
        lo, hi = _inner(g, old).amin(0)[m], _inner(g, old).amax(0)[m]
        x = _inner(g, new)[:, m]
        assert bool((x >= lo[None] - 1e-12).all()) and bool((x <= hi[None] + 1e-12).all())
    Ti = d["T"].clone()
    ALE_remap_tracers(R, dg, d["h"], d["h"], [Ti])
    assert float((_inner(g, Ti) - _inner(g, d["T"]))[:, m].abs().max()) <= 1e-12 * float(d["T"].abs().max())


@pytest.mark.gpu
def test_vertical_viscosity_budget_and_remnant_bounds(world):
    import torch
    from mom6_amd.vert_friction import vertvisc_init, vertvisc_step, vertvisc_type
    g, dg, d = world["g"], world["dg"], world["dyn"]
    CS = vertvisc_init(dg, KV=1.0e-4, HBBL=10.0, HMIX_FIXED=20.0, KV_ML_INVZ2=1.0e-2, CFL_BASED_TRUNCATIONS=False)
    mu, mv = world["mu"], world["mv"]
    visc = vertvisc_type(Kv_bbl_u=(3.0e-3 * mu).contiguous(), Kv_bbl_v=(3.0e-3 * mv).contiguous(),
                         bbl_thick_u=(10.0 * mu).contiguous(), bbl_thick_v=(10.0 * mv).contiguous())
    taux = (0.1 * mu).contiguous(); tauy = (-0.05 * mv).contiguous()
    u, v = d["u"].clone(), d["v"].clone()
    ru, rv = torch.zeros_like(u), torch.zeros_like(v)
    tbx, tby = torch.zeros_like(mu), torch.zeros_like(mv)
    dt = 900.0
    vertvisc_step(u, v, d["h"], None, (taux, tauy), visc, dt, dg, CS, ru, rv, True, tbx, tby)
    dg.sync()
    for pos, new, old, tau, tb, hn, rem, mask in ((U, u, d["u"], taux, tbx, "h_u", ru, mu), (V, v, d["v"], tauy, tby, "h_v", rv, mv)):
        sj, si = g.csl(pos)
        m = mask[sj, si] > 0
        hv = CS.arrays[hn][:, sj, si]
        lhs = ((hv * new[:, sj, si]).sum(0) - (hv * old[:, sj, si]).sum(0))[m]
        rhs = (dt * (tau[sj, si] - tb[sj, si]) / CS.st.H_to_RZ)[m]
        scale = ((hv * old[:, sj, si]).abs().sum(0) + (dt * tau[sj, si] / CS.st.H_to_RZ).abs())[m] + 1e-30
        assert float(((lhs - rhs).abs() / scale).max()) <= 1e-10
        r = rem[:, sj, si][:, m]
        assert float(r.min()) >= 0.0 and float(r.max()) <= 1.0 + 1e-13


@pytest.mark.gpu
def test_rk2_step_conserves_volume_and_stays_sane(world):
    import torch
    from mom6_amd.dynamics_split_rk2 import initialize_dyn_split_RK2, step_MOM_dyn_split_RK2
    from mom6_amd.vert_friction import vertvisc_type
    g, dg, d = world["g"], world["dg"], world["dyn"]
    u, v, h, Tt, Ss = (d[k].clone() for k in ("u", "v", "h", "T", "S"))
    Z = lambda pos, k3=True: torch.zeros(g.shape3(pos) if k3 else g.shape2(pos), dtype=torch.float64, device="cuda")
    uh, vh, uhtr, vhtr, eta_av = Z(U), Z(V), Z(U), Z(V), Z(H, False)
    mu, mv = world["mu"], world["mv"]
    CS = initialize_dyn_split_RK2(u, v, h, uh, vh, 900.0, dg, coriolis=dict(bound_coriolis=True),
                                  vertvisc=dict(KV=1.0e-4, HBBL=10.0, HMIX_FIXED=20.0, KV_ML_INVZ2=1.0e-2))
    visc = vertvisc_type(Kv_bbl_u=(3.0e-3 * mu).contiguous(), Kv_bbl_v=(3.0e-3 * mv).contiguous(),
                         bbl_thick_u=(10.0 * mu).contiguous(), bbl_thick_v=(10.0 * mv).contiguous())
    taux, tauy = (0.1 * mu).contiguous(), Z(V, False)
    A = _inner(g, world["areaT"] * world["mT"])[None]
    v0 = float((_inner(g, h) * A).sum())
    for n in range(2):
        step_MOM_dyn_split_RK2(u, v, h, (Tt, Ss), visc, None, 900.0, (taux, tauy), None, None, uh, vh, uhtr, vhtr, eta_av, dg, CS,
                               calc_dtbt=(n == 0))
    dg.sync()
    assert bool(torch.isfinite(u).all() and torch.isfinite(v).all() and torch.isfinite(h).all())
    assert float(u.abs().max()) < 3.0 and float(_inner(g, h).min()) >= g.Angstrom_H
    assert abs(float((_inner(g, h) * A).sum()) - v0) <= 1e-10 * v0
    # the barotropic free surface stays the layer sum to the solver's tolerance (bt_mass_source feeds the difference back)
    eta_h = _inner(g, h.sum(0) - torch.as_tensor(np.asarray(g.bathyT), device="cuda") * g.Z_to_H)
    m = _inner(g, world["mT"]) > 0
    assert float((eta_h - _inner(g, CS.eta))[m].abs().max()) < 0.05


@pytest.mark.gpu
def test_lateral_parameterizations_are_overturnings_that_conserve_volume(world):
    """thickness_diffuse and mixedlayer_restrat at full size: every face column's transports sum to zero (pure overturnings), the volume
    of every water column and of the ocean is conserved, land faces carry nothing, uhtr / vhtr are the transports times dt, and the
    restratification moves nothing below the deepest mixed layer"""
    import torch
    from mom6_amd.mixedlayer_restrat import mixedlayer_restrat, mixedlayer_restrat_init
    from mom6_amd.pressure_force import EOS_init
    from mom6_amd.thickness_diffuse import thickness_diffuse, thickness_diffuse_init
    g, dg, d = world["g"], world["dg"], world["dyn"]
    eos = EOS_init("WRIGHT")
    dt = 3600.0
    A = _inner(g, world["areaT"])[None]
    sh2 = tuple(d["h"].shape[1:])
    yy = torch.linspace(0.0, 1.0, sh2[0], device="cuda", dtype=torch.float64)[:, None].expand(sh2).contiguous()
    h0 = d["h"]
    col0 = (_inner(g, h0) * A).sum(0)
    # ---- thickness_diffuse (KHTH with a maximum, MEKE%Kh)
    h, uhtr, vhtr = h0.clone(), torch.zeros_like(d["u"]), torch.zeros_like(d["v"])
    uhGM, vhGM = torch.zeros_like(d["u"]), torch.zeros_like(d["v"])
    td = thickness_diffuse_init(dg, THICKNESSDIFFUSE=True, KHTH=300.0, KHTH_MAX=900.0, MEKE_KHTH_FAC=0.5)
    thickness_diffuse(h, uhtr, vhtr, (d["T"], d["S"], eos), dt, dg, dict(Kh=(600.0 * yy * world["mT"]).contiguous()), None, dict(uhGM=uhGM, vhGM=vhGM), td)
    dg.sync()
    for q, m, pos in ((uhGM, world["mu"], U), (vhGM, world["mv"], V)):
        qi = _inner(g, q, pos)
        assert bool(torch.isfinite(qi).all()) and float(qi.abs().max()) > 0.0
        assert float((qi.sum(0).abs() / (qi.abs().sum(0) + 1e-30)).max()) < 1e-10      # the column's transports cancel
        assert float((qi * (1.0 - _inner(g, m, pos))[None]).abs().max()) == 0.0           # nothing through land
    assert torch.equal(uhtr, uhGM * dt) and torch.equal(vhtr, vhGM * dt)
    col1 = (_inner(g, h) * A).sum(0)
    assert float(((col1 - col0).abs() / (col0.abs() + 1.0)).max()) < 1e-12 and float(_inner(g, h).min()) >= g.Angstrom_H
    assert abs(float(col1.sum()) - float(col0.sum())) <= 1e-13 * float(col0.sum())
    # ---- mixedlayer_restrat (OM4's settings: the boundary-layer depth, both running means, the frontal length scale)
    mld = (20.0 + 80.0 * yy).contiguous()
    mle = mixedlayer_restrat_init(dg, FOX_KEMPER_ML_RESTRAT_COEF=1.0, FOX_KEMPER_ML_RESTRAT_COEF2=0.5, MLE_FRONT_LENGTH=500.0, MLE_USE_PBL_MLD=True,
                                  MLE_MLD_DECAY_TIME=345600.0, MLE_MLD_DECAY_TIME2=5184000.0, MLD_filtered=torch.zeros_like(yy),
                                  MLD_filtered_slow=torch.zeros_like(yy))
    h2, u2, v2 = h.clone(), torch.zeros_like(d["u"]), torch.zeros_like(d["v"])
    uhml, vhml = torch.zeros_like(d["u"]), torch.zeros_like(d["v"])
    mixedlayer_restrat(h2, u2, v2, (d["T"], d["S"], eos), dict(ustar=(0.005 + 0.01 * yy).contiguous()), dt, None, mld, None,
                       dict(Rd_dx_h=(0.2 + 1.5 * yy).contiguous()), dg, mle, uhml, vhml)
    dg.sync()
    for q, pos in ((uhml, U), (vhml, V)):
        qi = _inner(g, q, pos)
        assert bool(torch.isfinite(qi).all()) and float(qi.abs().max()) > 0.0
        assert float((qi.sum(0).abs() / (qi.abs().sum(0) + 1e-30)).max()) < 1e-10
    assert torch.equal(u2, uhml * dt) and torch.equal(v2, vhml * dt)
    col2 = (_inner(g, h2) * A).sum(0)
    assert float(((col2 - col1).abs() / (col1.abs() + 1.0)).max()) < 1e-12
    # the running means took the boundary-layer depth (they start from zero) and nothing moved where every layer lies below it
    assert torch.equal(mle.MLD_filtered[g.csl(H)], mld[g.csl(H)])
    # (a layer's position in the mixed layer is taken at the FACES, from the mean of the two columns: it lies below the deepest mixed
    # layer, 100 m, at all four faces of a cell if its top does in the cell and in its four neighbours)
    ztop = torch.cumsum(_inner(g, h), 0) - _inner(g, h)      # depth of the top of each layer
    zmin = ztop.clone()
    for sh, ax in ((1, -1), (-1, -1), (1, -2), (-1, -2)):
        zmin = torch.minimum(zmin, torch.roll(ztop, sh, ax))
    deep = zmin > 1.05 * 100.0 + 1.0
    assert float(deep.double().mean()) > 0.3
    assert float(((_inner(g, h2) - _inner(g, h)).abs() * deep).max()) == 0.0


@pytest.mark.gpu
def test_neutral_diffusion_conserves_keeps_constants_and_mixes_nothing_across_density(world):
    """tracer_hordiff with USE_NEUTRAL_DIFFUSION at full size: the inventory of every tracer over (h + H_subroundoff) * area is conserved
    (a flux leaves one cell and enters its neighbour), a constant tracer keeps its bits (no differences, no fluxes), a tracer that is a
    function of density alone -- here temperature itself under a linear equation of state without a salinity term -- has nothing to
    flux along neutral surfaces beyond the interpolation's residue, while a passive tracer unrelated to density is mixed"""
    import torch
    from mom6_amd.pressure_force import EOS_init
    from mom6_amd.tracer_hor_diff import tracer_hor_diff_init, tracer_hordiff
    g, dg, d = world["g"], world["dg"], world["dyn"]
    h = d["h"]
    A = _inner(g, world["areaT"])[None]
    hh = _inner(g, h) + g.H_subroundoff
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    T, S = d["T"].clone(), d["S"].clone()
    passive = (torch.rand(T.shape, device="cuda", dtype=torch.float64, generator=gen) * world["mT"][None]).contiguous()
    const = torch.full_like(T, 2.75)
    trs = [T, S, passive, const]
    before = [t.clone() for t in trs]
    CS = tracer_hor_diff_init(KHTR=1000.0, USE_NEUTRAL_DIFFUSION=True)
    st = tracer_hordiff(h, 7200.0, None, None, None, dg, CS, trs, tv=dict(T=T, S=S, eqn_of_state=EOS_init("LINEAR", dRho_dT=-0.2, dRho_dS=0.0)))
    dg.sync()
    assert st.num_itts == 1 and st.halo_updates == 1
    for m, (a, b) in enumerate(zip(before, trs)):
        ai, bi = _inner(g, a), _inner(g, b)
        assert bool(torch.isfinite(bi).all())
        s0, s1 = float((hh * A * ai).sum()), float((hh * A * bi).sum())
        assert abs(s1 - s0) <= 1e-11 * max(abs(s0), 1.0), (m, s0, s1)
    assert torch.equal(_inner(g, trs[3]), _inner(g, before[3]))
    dT = float((_inner(g, T) - _inner(g, before[0])).abs().max())
    dP = float((_inner(g, passive) - _inner(g, before[2])).abs().max())
    assert dP > 1e-3 and dT < 0.05 * dP, (dT, dP)      # density's own tracer barely moves along its own surfaces; the passive one mixes
