"""tracer_epipycnal_ML_diff (src/tracer/MOM_tracer_hor_diff.F90:700-1621; DIFFUSE_ML_TO_INTERIOR, .testing/tc1) and the branches of
tracer_hordiff it changes (:544-550): the oracle against what the scheme guarantees on the CPU (the reference holds no known-answer
vectors for it: parity unpinned), the library against the oracle on the GPU, bit for bit."""
import numpy as np
import pytest

from helpers import bits_equal, interior
from mom6_amd import _abi, synth
from oracle import orc

NKML, NKBL = 2, 1
NKMB = NKML + NKBL


def layered_case(ni=26, nj=18, nk=9, seed=3, reentrant=(True, False), land_frac=0.2, exact=False, eos_form="wright"):
    """A bulk-mixed-layer state: NKMB variable-density layers over isopycnal layers with target densities Rlay; interior layers
    lighter than the local mixed layer are (mostly) vanished, as in a layered run.  exact: a linear equation of state that depends
    on T only and whole-number temperatures, so that the coordinate densities of neighbouring columns and the target densities
    coincide to the bit (the branches for equal densities, :986-1008)."""
    g = synth.make_grid(ni, nj, nk, land_frac=land_frac, seed=seed + 700, reentrant_x=reentrant[0], reentrant_y=reentrant[1], halo=4)
    rng = np.random.default_rng(seed)
    shp = (nk, g.njh, g.nih)
    jj, ii = np.meshgrid(np.arange(g.njh), np.arange(g.nih), indexing="ij")
    if exact:
        eos = orc.eos("LINEAR", 1030.0, -1.0, 0.0)
        Rlay = 1030.0 - np.arange(nk + 3, 3, -1.0)[:nk]                 # whole numbers, increasing with k
        T = np.empty(shp); S = np.full(shp, 35.0)
        Tml = np.floor(4.0 + 6.0 * (0.5 + 0.5 * np.sin(0.5 * ii + 0.3 * jj)) + rng.integers(0, 2, (g.njh, g.nih)))
        for k in range(NKMB):
            T[k] = Tml - k * rng.integers(0, 2, (g.njh, g.nih))
        for k in range(NKMB, nk):
            T[k] = 1030.0 - Rlay[k]
    else:
        eos = orc.eos("WRIGHT" if eos_form == "wright" else "LINEAR")
        T = np.empty(shp); S = np.empty(shp)
        Tml = 4.0 + 18.0 * (0.5 + 0.5 * np.sin(0.45 * ii + 0.2) * np.cos(0.3 * jj)) + rng.normal(0, 0.5, (g.njh, g.nih))
        for k in range(NKMB):
            T[k] = Tml - 0.7 * k * rng.random((g.njh, g.nih)); S[k] = 34.5 + 0.3 * rng.random((g.njh, g.nih)) + 0.05 * k
        rho_ml = np.array([[orc.eos_density(eos, T[NKMB - 1, j, i], S[NKMB - 1, j, i], 2.0e7) for i in range(g.nih)] for j in range(g.njh)])
        lo, hi = np.percentile(rho_ml, 15), rho_ml.max() + 1.5
        Rlay = np.linspace(lo - 1.0, hi, nk)
        for k in range(NKMB, nk):
            T[k] = 10.0 - 1.2 * (k - NKMB) + rng.normal(0, 0.2, (g.njh, g.nih)); S[k] = 35.0 + 0.05 * k + rng.normal(0, 0.02, (g.njh, g.nih))
    rho_top = np.array([[max(orc.eos_density(eos, T[k, j, i], S[k, j, i], 2.0e7) for k in range(NKMB)) for i in range(g.nih)] for j in range(g.njh)])
    h = np.empty(shp)
    ang = g.Angstrom_H
    for k in range(nk):
        hk = (5.0 + 40.0 * rng.random((g.njh, g.nih))) if k < NKMB else (2.0 + 120.0 * rng.random((g.njh, g.nih)))
        if k >= NKMB:      # lighter than the mixed layer: vanished, but for a few columns (unstable columns are sorted)
            light = (Rlay[k] < rho_top) & (rng.random((g.njh, g.nih)) > 0.06)
            hk = np.where(light, ang, hk)
        hk = np.where(rng.random((g.njh, g.nih)) < 0.07, ang * rng.choice([1.0, 5.0, 30.0], (g.njh, g.nih)), hk)      # around h_exclude
        h[k] = hk
    h[0] = np.maximum(h[0], 1.0)
    mT = g.mask2dT[None]
    tr = [np.ascontiguousarray(T), np.ascontiguousarray(S), np.ascontiguousarray(rng.random(shp) * mT),
          np.ascontiguousarray(np.where(rng.random(shp) > 0.8, 1.0, 0.0) * mT)]
    for t in tr + [h]:
        orc.halo_update(g, t, _abi.POS_H)
    return g, np.ascontiguousarray(h), tr, eos, Rlay


def epi(eos, Rlay, **kw):
    return dict(eos=eos, Rlay=Rlay, nkml=NKML, nk_rho_varies=NKMB, idx_T=0, idx_S=1, **kw)


def inventory(g, h, t):
    return float((interior(g, h) * interior(g, g.areaT)[None] * interior(g, t)).sum())


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("answer_date", [20240101, 20240401])
def test_oracle_conserves_and_stays_within_the_range(exact, answer_date):
    g, h, tr, eos, Rlay = layered_case(exact=exact)
    const = np.full_like(tr[0], 7.25)
    tr = [t.copy() for t in tr] + [const]
    before = [t.copy() for t in tr]
    st = orc.tracer_hordiff(g, h, 3600.0 * 24, tr, 800.0, epipycnal=epi(eos, Rlay, ML_KhTr_scale=0.0, answer_date=answer_date))
    assert st.num_itts == 1 and st.halo_updates == 2      # one pass in the along-layer loop, one in tracer_epipycnal_ML_diff
    for m, (t0, t1) in enumerate(zip(before, tr)):
        a, b = inventory(g, h, t0), inventory(g, h, t1)
        # the along-layer part conserves over h + h_neglect, the epipycnal part over h: both to roundoff here
        assert abs(a - b) <= 1e-10 * max(1.0, abs(a)), (m, a, b)
        assert interior(g, t1).max() <= interior(g, t0).max() + 1e-9 and interior(g, t1).min() >= interior(g, t0).min() - 1e-9, m
    assert np.array_equal(interior(g, tr[-1]), interior(g, before[-1]))      # a constant stays, to the bit
    # with ML_KHTR_SCALE = 0 the along-layer diffusion leaves the variable-density layers alone (:544-550): what changes them is the
    # epipycnal exchange with the interior
    chg = [float(np.abs(interior(g, tr[2])[k] - interior(g, before[2])[k]).max()) for k in range(h.shape[0])]
    assert all(c > 0 for c in chg[:NKMB]) and max(chg[NKMB:]) > 0


def test_oracle_equal_answers_when_no_layer_is_split():
    """with every pairing between whole layers the two answer dates differ only in the order of the sums; and without the
    exchange (no diffusivity) nothing moves"""
    g, h, tr, eos, Rlay = layered_case(exact=True)
    a = [t.copy() for t in tr]; b = [t.copy() for t in tr]
    orc.tracer_hordiff(g, h, 3600.0, a, 300.0, epipycnal=epi(eos, Rlay, answer_date=20240101))
    orc.tracer_hordiff(g, h, 3600.0, b, 300.0, epipycnal=epi(eos, Rlay, answer_date=20240401))
    for x, y in zip(a, b):
        assert np.allclose(interior(g, x), interior(g, y), rtol=0, atol=1e-11)
    c = [t.copy() for t in tr]
    st = orc.tracer_hordiff(g, h, 3600.0, c, 0.0, epipycnal=epi(eos, Rlay))
    assert st.num_itts == 0 and all(np.array_equal(x, y) for x, y in zip(c, tr))


def test_oracle_limit_bug_and_iterations():
    g, h, tr, eos, Rlay = layered_case()
    a = [t.copy() for t in tr]; b = [t.copy() for t in tr]; c = [t.copy() for t in tr]
    orc.tracer_hordiff(g, h, 86400.0 * 5, a, 2000.0, epipycnal=epi(eos, Rlay, limit_bug=True))
    orc.tracer_hordiff(g, h, 86400.0 * 5, b, 2000.0, epipycnal=epi(eos, Rlay, limit_bug=False))
    assert any(not np.array_equal(x, y) for x, y in zip(a, b))      # HOR_DIFF_LIMIT_BUG changes the range of a face (:1293-1297)
    st = orc.tracer_hordiff(g, h, 86400.0 * 5, c, 2.0e5, check_diffusive_CFL=True, epipycnal=epi(eos, Rlay))
    assert st.num_itts > 1 and st.halo_updates == 2 * st.num_itts      # the epipycnal part iterates as often (:1257-1261)
    assert all(np.isfinite(x).all() for x in c)


EPI_CASES = [dict(), dict(exact=True), dict(answer_date=20240401), dict(exact=True, answer_date=20240401), dict(limit_bug=False),
             dict(KhTr=2.0e5, check=True, dt=86400.0 * 5), dict(ML_KhTr_scale=0.0, conc_underflow=[0.0, 0.0, 1.0e-2, 0.5]),
             dict(ML_KhTr_scale=0.4, eos_form="linear", reentrant=(True, True)), dict(reentrant=(False, False), ni=70, nj=9, nk=6),
             dict(exact=True, nk=14, ni=40, nj=12, seed=11)]


@pytest.mark.gpu
@pytest.mark.parametrize("kw", EPI_CASES, ids=[",".join(f"{k}={v}" for k, v in c.items()) or "default" for c in EPI_CASES])
@pytest.mark.parametrize("space", ["device", "host"])
def test_epipycnal_diffusion_matches_oracle_bitwise(kw, space):
    import torch
    from mom6_amd.tracer_advect import DeviceGrid
    from mom6_amd.tracer_hor_diff import tracer_hor_diff_init, tracer_hordiff
    kw = dict(kw)
    gk = {k: kw.pop(k) for k in ("reentrant", "ni", "nj", "nk", "exact", "eos_form", "seed") if k in kw}
    cu = kw.pop("conc_underflow", None)
    KhTr, check, dt = kw.pop("KhTr", 800.0), kw.pop("check", False), kw.pop("dt", 86400.0)
    g, h, tr, eos, Rlay = layered_case(**gk)
    ref = [t.copy() for t in tr]
    rs = orc.tracer_hordiff(g, h, dt, ref, KhTr, check_diffusive_CFL=check, conc_underflow=cu, epipycnal=epi(eos, Rlay, **kw))
    dg = DeviceGrid(g)
    put = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if space == "device" else (lambda a: np.ascontiguousarray(a).copy())
    dtr = [put(t) for t in tr]
    names = {"ML_KhTr_scale": "ML_KHTR_SCALE", "answer_date": "HOR_DIFF_ANSWER_DATE", "limit_bug": "HOR_DIFF_LIMIT_BUG"}
    CS = tracer_hor_diff_init(KHTR=KhTr, CHECK_DIFFUSIVE_CFL=check, DIFFUSE_ML_TO_INTERIOR=True, **{names[k]: v for k, v in kw.items()})
    tv = dict(T=dtr[0], S=dtr[1], eqn_of_state=eos, P_Ref=2.0e7)
    st = tracer_hordiff(put(h), dt, None, None, None, dg, CS, dtr, tv=tv, conc_underflow=cu, GV=dict(Rlay=Rlay, nkml=NKML, nk_rho_varies=NKMB))
    dg.sync()
    assert (st.num_itts, st.halo_updates) == (rs.num_itts, rs.halo_updates) and st.max_CFL == rs.max_CFL
    changed = False
    for m, (a, b) in enumerate(zip(dtr, ref)):
        an = a.cpu().numpy() if space == "device" else a
        assert bits_equal(interior(g, an), interior(g, b)), m
        changed = changed or not np.array_equal(interior(g, b), interior(g, tr[m]))
    assert changed
    dg.close()


@pytest.mark.gpu
def test_epipycnal_entry_points_refuse_what_they_cannot_take():
    import torch
    from mom6_amd._lib import Mom6HipError
    from mom6_amd.tracer_advect import DeviceGrid
    from mom6_amd.tracer_hor_diff import tracer_hor_diff_init, tracer_hordiff
    g, h, tr, eos, Rlay = layered_case()
    dg = DeviceGrid(g)
    dtr = [torch.from_numpy(t).cuda() for t in tr]
    with pytest.raises(Mom6HipError, match="mutually exclusive"):
        tracer_hor_diff_init(KHTR=10.0, DIFFUSE_ML_TO_INTERIOR=True, USE_NEUTRAL_DIFFUSION=True)
    CS = tracer_hor_diff_init(KHTR=10.0, DIFFUSE_ML_TO_INTERIOR=True)
    tv = dict(T=dtr[0], S=dtr[1], eqn_of_state=eos, P_Ref=2.0e7)
    with pytest.raises(Mom6HipError, match="GV%Rlay"):
        tracer_hordiff(torch.from_numpy(h).cuda(), 3600.0, None, None, None, dg, CS, dtr, tv=tv)
    with pytest.raises(Mom6HipError, match="nk_rho_varies"):      # not a layered run
        tracer_hordiff(torch.from_numpy(h).cuda(), 3600.0, None, None, None, dg, CS, dtr, tv=tv, GV=dict(Rlay=Rlay, nkml=0, nk_rho_varies=0))
    dg.close()


def _write_layered_tracer_case(tmp, g, h, tr, eos_name, Rlay, params, resident):
    """the input and parameter files of tests/fortran/tracer_driver.F90 for a layered state at rest (no transports: advect_tracer is
    called and has nothing to move)"""
    zu = np.zeros(g.shape3(_abi.POS_U)); zv = np.zeros(g.shape3(_abi.POS_V)); z2 = {n: np.zeros(g.shape2(p)) for n, p in
        (("Kh", _abi.POS_H), ("L2u", _abi.POS_U), ("L2v", _abi.POS_V), ("SN_u", _abi.POS_U), ("SN_v", _abi.POS_V), ("Res_fn_h", _abi.POS_H), ("Rd_dx_h", _abi.POS_H))}
    opt = [len(tr), 0, 0, 0, 0, NKMB, NKML, 0]
    dt = 86400.0
    with open(tmp / "in.bin", "wb") as fh:
        np.array([g.ni, g.nj, g.nk, g.halo, int(g.reentrant_x), int(g.reentrant_y), g.first_direction, 0], dtype="<i4").tofile(fh)
        np.array([g.Angstrom_H, g.H_subroundoff, g.dZ_subroundoff, g.H_to_Z, g.Z_to_H, g.g_Earth, g.Rho0, 900.0], dtype="<f8").tofile(fh)
        np.array(opt, dtype="<i4").tofile(fh)
        for n in _abi.ALL_METRICS:
            np.ascontiguousarray(g.metrics[n], dtype="<f8").tofile(fh)
        np.array([dt, 1.0], dtype="<f8").tofile(fh)
        for a in [h, zu, zv] + tr + [z2[n] for n in ("Kh", "L2u", "L2v", "SN_u", "SN_v", "Res_fn_h", "Rd_dx_h")]:
            np.ascontiguousarray(a, dtype="<f8").tofile(fh)
        np.ascontiguousarray(Rlay, dtype="<f8").tofile(fh)
        np.array([2.0e7], dtype="<f8").tofile(fh)
    with open(tmp / "params.txt", "w") as fh:
        fh.write(f"TRACER_ADVECTION_SCHEME = PPM:H3\nDT = 900.0\nGPU_RESIDENT_DYNAMICS = {resident}\nEQN_OF_STATE = {eos_name}\n")
        fh.write("DIFFUSE_ML_TO_INTERIOR = True\n")
        for k, v in params.items():
            fh.write(f"{k} = {v}\n")
    return dt


@pytest.mark.gpu
def test_epipycnal_diffusion_from_fortran(tmp_path):
    """tracer_hor_diff_init / tracer_hordiff of the shim with DIFFUSE_ML_TO_INTERIOR as .testing/tc1 sets it (ML_KHTR_SCALE = 0), and with the
    later answer date: GV%Rlay, GV%nkml, GV%nk_rho_varies and tv%P_Ref come from the reference's own types; the oracle's bits"""
    import os, subprocess
    from test_fortran_abi import FC, _build_shims
    if not os.path.exists(FC):
        pytest.skip("amdflang not present")
    exe = _build_shims(tmp_path, driver="tracer_driver")
    g, h, tr, eos, Rlay = layered_case(ni=30, nj=16, nk=8)
    for params, resident in [(dict(KHTR=800.0, ML_KHTR_SCALE=0.0), False), (dict(KHTR=800.0, ML_KHTR_SCALE=0.0), True),
                             (dict(KHTR=2.0e5, CHECK_DIFFUSIVE_CFL=True, HOR_DIFF_ANSWER_DATE=20240401), False)]:
        dt = _write_layered_tracer_case(tmp_path, g, h, tr, "WRIGHT", Rlay, params, resident)
        ref = [t.copy() for t in tr]
        orc.advect_tracer(g, h, np.zeros(g.shape3(_abi.POS_U)), np.zeros(g.shape3(_abi.POS_V)), dt, 900.0, "PPM:H3", ref)
        for t in ref:
            orc.halo_update(g, t, _abi.POS_H)
        orc.tracer_hordiff(g, h, dt, ref, params["KHTR"], check_diffusive_CFL=params.get("CHECK_DIFFUSIVE_CFL", False),
                           epipycnal=epi(eos, Rlay, ML_KhTr_scale=params.get("ML_KHTR_SCALE", 1.0), answer_date=params.get("HOR_DIFF_ANSWER_DATE", 20240101)))
        r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "params.txt")], capture_output=True, text=True)
        assert r.returncode == 0 and "tracer_driver ok" in r.stdout, (params, r.stderr[-600:])
        raw = np.fromfile(str(tmp_path / "out.bin"), dtype="<f8").reshape((len(tr),) + tr[0].shape)
        for m, w in enumerate(ref):
            assert bits_equal(interior(g, raw[m]), interior(g, w)), (params, m)
        assert not bits_equal(interior(g, raw[2]), interior(g, tr[2]))
