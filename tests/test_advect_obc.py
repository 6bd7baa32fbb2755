"""advect_tracer with an associated OBC whose segments carry tracer registries (src/tracer/MOM_tracer_advect.F90:441-477, :580-627 in
advect_x, :823-861, :965-1014 in advect_y): a registered tracer takes its reservoir value (or its inflow concentration) in the cell outside a
segment, the slopes of the three cells about the segment's face are formed again, and an inflow through the face carries the reservoir value
with the whole remaining transport.  The oracle against what those lines state and against a quarter turn of the grid, on the CPU; the
library against the oracle on the GPU, bit for bit.  (The reference holds no known-answer vectors for this routine: parity unpinned.)"""
import numpy as np
import pytest

from helpers import bits_equal, interior
from mom6_amd import _abi, synth
from mom6_amd.open_boundary import ocean_OBC_type
from oracle import orc
from rotation import rot, rot_vector, rotate_grid, unrot
from test_continuity_obc import open_faces, turned_segments

SCHEMES = ["PLM", "PPM:H3", "PPM"]
# the four sides (the cells outside lie in the halo, masked), two specified segments inside the domain
SEGS = ["J=N,I=N:0,FLATHER,ORLANSKI", "J=0,I=0:N,FLATHER,ORLANSKI", "I=N,J=0:N,SIMPLE", "I=0,J=N:0,FLATHER,ORLANSKI", "I=9,J=0:N,SIMPLE",
        "J=7,I=N:0,SIMPLE"]


def adv_obc_case(segs, ni=22, nj=16, nk=3, ntr=3, seed=3, registry=True, land_frac=0.1):
    g = synth.make_grid(ni, nj, nk, halo=4, land_frac=land_frac, seed=seed + 100, reentrant_x=False, reentrant_y=False)
    OBC = ocean_OBC_type(g, segs)
    open_faces(g, OBC)
    st = synth.make_advection_state(g, ntr=ntr, seed=seed, hot_frac=0.004, vanish_frac=0.05, cfl=0.15)
    case = {k: (v.numpy() if k != "tr" else [t.numpy() for t in v]) for k, v in st.items()}
    rng = np.random.default_rng(seed)      # (the faces of the segments are open when the transports are made: they carry flow in and out)
    if registry:
        for n, s in enumerate(OBC.segment):
            if not s.on_pe:
                continue
            # tracer 1 with a reservoir, tracer 3 with an inflow concentration; tracer 2 is not registered
            s.tr_Reg = [dict(ntr_index=1, tres=5.0 + rng.random(s.normal_vel.shape)), dict(ntr_index=3, OBC_inflow_conc=0.25 + 0.1 * n)]
    return g, case, OBC


def run(g, case, OBC, scheme, x_first=None, **kw):
    tr = [t.copy() for t in case["tr"]]
    uhr = g.zeros3(_abi.POS_U); vhr = g.zeros3(_abi.POS_V)
    st = orc.advect_tracer(g, case["h_end"], case["uhtr"], case["vhtr"], 3600.0, 900.0, scheme, tr, x_first=x_first, uhr_out=uhr, vhr_out=vhr,
                           OBC=OBC, **kw)
    return dict(tr=tr, uhr=uhr, vhr=vhr, stats=st)


@pytest.mark.parametrize("scheme", SCHEMES)
def test_segments_without_a_registry_change_nothing(scheme):
    g, case, OBC = adv_obc_case(SEGS, registry=False)
    a, b = run(g, case, OBC, scheme), run(g, case, None, scheme)
    assert all(bits_equal(x, y) for x, y in zip(a["tr"], b["tr"])) and bits_equal(a["uhr"], b["uhr"])


@pytest.mark.parametrize("scheme", SCHEMES)
def test_an_unregistered_tracer_and_the_far_field_are_untouched_and_inflow_brings_the_reservoir(scheme):
    g, case, OBC = adv_obc_case(SEGS)
    a, b = run(g, case, OBC, scheme), run(g, case, None, scheme)
    near = np.zeros(g.shape2(_abi.POS_H), dtype=bool)
    on_u, on_v = OBC.segnum_u != 0, OBC.segnum_v != 0
    near |= on_u[:, 1:] | on_u[:, :-1] | on_v[1:, :] | on_v[:-1, :]
    for _ in range(4 * 4):      # (a few passes of the advection, a stencil of at most 3 cells each)
        near[1:, :] |= near[:-1, :].copy(); near[:-1, :] |= near[1:, :].copy(); near[:, 1:] |= near[:, :-1].copy(); near[:, :-1] |= near[:, 1:].copy()
    assert not bits_equal(a["tr"][0], b["tr"][0]) and not bits_equal(a["tr"][2], b["tr"][2])
    # a uniform tracer stays uniform in a closed advection; with a reservoir of another value the cells inside an inflow face change
    g2, case2, OBC2 = adv_obc_case(SEGS)
    case2["tr"][0][:] = 1.0
    for s in OBC2.segment:
        if s.on_pe:
            s.tr_Reg = [dict(ntr_index=1, OBC_inflow_conc=3.0)]
    c = run(g2, case2, OBC2, scheme)
    t = interior(g2, c["tr"][0]); wet = interior(g2, np.asarray(g2.mask2dT)) > 0
    # (no bound holds: an inflow takes the whole remaining transport of its face at once, past the limiter of the cell's volume :593)
    assert np.all(np.isfinite(t)) and t[:, wet].max() > 1.0 + 1e-6


# The reference is not symmetric under the quarter turn for every segment: the three cells whose slopes it forms again are I-1 .. I+1 and
# J-1 .. J+1 whatever the side the segment opens to (:466, :850; each with the masks of its own two faces), so an eastern segment and the
# southern one it turns into refresh different cells; and a non-specified segment inside the domain takes its inflow value in advect_x only
# (:586-599 against :969).  Northern and southern segments turn into eastern and western ones that do the same arithmetic: each of the four
# branches (E, W in advect_x; N, S in advect_y) is on one side of such a pair.
SEGS_TURN = ["J=N,I=N:0,FLATHER,ORLANSKI", "J=0,I=0:N,FLATHER,ORLANSKI", "J=7,I=N:0,SIMPLE", "J=11,I=0:N,SIMPLE"]


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("x_first", [True, False])
def test_oracle_turns_with_the_grid(scheme, x_first):
    g, case, OBC = adv_obc_case(SEGS_TURN)
    a = run(g, case, OBC, scheme, x_first=x_first)
    gr = rotate_grid(g)
    OBCr = ocean_OBC_type(gr, turned_segments(SEGS_TURN, g.ni, g.nj))
    assert sorted(s.direction for s in OBCr.segment) == [_abi.OBC_DIRECTION_E] * 2 + [_abi.OBC_DIRECTION_W] * 2
    for s, sr in zip(OBC.segment, OBCr.segment):
        if s.on_pe:      # a scalar on the faces of a u segment (nk, j, 1) -> on those of the v' segment (nk, 1, i' = j); of a v segment -> (nk, j' = ni - 1 - i, 1)
            tres = s.tr_Reg[0]["tres"]
            tr_ = np.swapaxes(tres, 1, 2) if s.is_E_or_W else np.swapaxes(tres, 1, 2)[:, ::-1, :]
            sr.tr_Reg = [dict(ntr_index=1, tres=np.ascontiguousarray(tr_)), dict(s.tr_Reg[1])]
    ur, vr = rot_vector(case["uhtr"], case["vhtr"])
    cr = dict(h_end=rot(case["h_end"]), uhtr=ur, vhtr=vr, tr=[rot(t) for t in case["tr"]])
    b = run(gr, cr, OBCr, scheme, x_first=not x_first)      # x' = y: the turned grid does y first where this one does x first
    for m in range(3):
        assert bits_equal(interior(g, unrot(b["tr"][m])), interior(g, a["tr"][m])), m


# ---- the reference's own lines, written out once more for one row (PLM) ----
# The reference holds no vectors for advect_x; what can be checked without the oracle's author reading the same lines the same way twice is a
# second, separate transcription of the row loop of advect_x (:408-662) in Python scalars, for the branches this case reaches (the transports
# stay below the limiter of :488-514, which the test asserts).  It pins the point ADVICE r04 found: Fortran does not tell `I` from `i`, so
# `do i=segment%HI%IsdB-1,segment%HI%IsdB+1` (:466) runs the I of `G%mask2dCu(I,j)*G%mask2dCu(I-1,j)` (:472) with it -- the three slopes
# formed again about a segment take the masks of their own faces; with land outside a western segment the first cell inside keeps its slope.
def _sign(a, b):
    return abs(a) if b >= 0.0 else -abs(a)


def ref_advect_x_row_plm(g, OBC, j, k, T, hprev, uhr, uh_neglect):
    """one (j, k) row of advect_x, usePPM = False; T: list of 1-D rows indexed by i - isd, updated in place; hprev, uhr rows likewise"""
    isd, ied, is_, ie = g.isd, g.ied, g.isc, g.iec
    X = lambda i: i - isd          # h-point column of cell i
    U = lambda I: I - isd + 1      # u-point column of face I
    mCu = np.asarray(g.mask2dCu)[j - g.jsd]; areaT = np.asarray(g.areaT)[j - g.jsd]
    min_h = 0.1 * g.Angstrom_H; tiny_h = np.finfo(np.float64).tiny; h_neglect = g.H_subroundoff
    ntr = len(T); stencil = 1
    slope = [dict() for _ in range(ntr)]
    for m in range(ntr):                                              # :409-436
        for i in range(is_ - stencil, ie + stencil + 1):
            Tp, Tc, Tm = T[m][X(i + 1)], T[m][X(i)], T[m][X(i - 1)]
            dMx = max(Tp, Tc, Tm) - Tc; dMn = Tc - min(Tp, Tc, Tm)
            slope[m][i] = mCu[U(i)] * mCu[U(i - 1)] * _sign(min(0.5 * abs(Tp - Tm), 2.0 * dMx, 2.0 * dMn), Tp - Tm)
    T_tmp = [t.copy() for t in T]                                     # :439-443
    for s in OBC.segment:                                             # :445-481
        if not s.tr_Reg or not s.is_E_or_W or not (s.HI['jsd'] <= j <= s.HI['jed']):
            continue
        I = s.HI['IsdB']
        for reg in s.tr_Reg:
            m = reg["ntr_index"] - 1
            val = reg["tres"][k, j - s.HI['jsd'], 0] if reg.get("tres") is not None else reg["OBC_inflow_conc"]
            if s.direction == _abi.OBC_DIRECTION_W:
                T_tmp[m][X(I)] = val
            else:
                T_tmp[m][X(I + 1)] = val
        for m in range(ntr):
            for i in range(s.HI['IsdB'] - 1, s.HI['IsdB'] + 2):                   # (the loop variable is the I of the masks below)
                I = i
                Tp, Tc, Tm = T_tmp[m][X(i + 1)], T_tmp[m][X(i)], T_tmp[m][X(i - 1)]
                dMx = max(Tp, Tc, Tm) - Tc; dMn = Tc - min(Tp, Tc, Tm)
                slope[m][i] = mCu[U(I)] * mCu[U(I - 1)] * _sign(min(0.5 * abs(Tp - Tm), 2.0 * dMx, 2.0 * dMn), Tp - Tm)
    uhh, CFL = {}, {}
    for I in range(is_ - 1, ie + 1):                                  # :485-514
        i = I; u = uhr[U(I)]
        if u == 0.0 or (u < 0.0 and hprev[X(i + 1)] <= tiny_h) or (u > 0.0 and hprev[X(i)] <= tiny_h):
            uhh[I] = 0.0; CFL[I] = 0.0
        elif u < 0.0:
            hup = hprev[X(i + 1)] - areaT[X(i + 1)] * min_h; hlos = max(0.0, uhr[U(I + 1)])
            assert not ((((hup - hlos) + u) < 0.0) and ((0.5 * hup + u) < 0.0)), "the case must stay below the limiter"
            uhh[I] = u; CFL[I] = -uhh[I] / hprev[X(i + 1)]
        else:
            hup = hprev[X(i)] - areaT[X(i)] * min_h; hlos = max(0.0, -uhr[U(I - 1)])
            assert not ((((hup - hlos) - u) < 0.0) and ((0.5 * hup - u) < 0.0)), "the case must stay below the limiter"
            uhh[I] = u; CFL[I] = uhh[I] / hprev[X(i)]
    flux = [dict() for _ in range(ntr)]
    for m in range(ntr):                                              # :560-578
        for I in range(is_ - 1, ie + 1):
            i = I
            if uhh[I] >= 0.0:
                Tc = T_tmp[m][X(i)]; flux[m][I] = uhh[I] * (Tc + 0.5 * slope[m][i] * (1. - CFL[I]))
            else:
                Tc = T_tmp[m][X(i + 1)]; flux[m][I] = uhh[I] * (Tc - 0.5 * slope[m][i + 1] * (1. - CFL[I]))
    if OBC.specified_u_BCs_exist_globally or OBC.open_u_BCs_exist_globally:      # :581-603
        for s in OBC.segment:
            if not s.tr_Reg or not s.is_E_or_W or not (s.HI['jsd'] <= j <= s.HI['jed']):
                continue
            I = s.HI['IsdB']
            if (uhr[U(I)] > 0.0 and s.direction == _abi.OBC_DIRECTION_W) or (uhr[U(I)] < 0.0 and s.direction == _abi.OBC_DIRECTION_E):
                uhh[I] = uhr[U(I)]
                for reg in s.tr_Reg:
                    m = reg["ntr_index"] - 1
                    flux[m][I] = uhh[I] * (reg["tres"][k, j - s.HI['jsd'], 0] if reg.get("tres") is not None else reg["OBC_inflow_conc"])
    if OBC.open_u_BCs_exist_globally:                                 # :605-627
        mT = np.asarray(g.mask2dT)[j - g.jsd]
        for s in OBC.segment:
            I = s.HI['IsdB']; i = I
            if s.is_E_or_W and (s.HI['jsd'] <= j <= s.HI['jed']):
                if s.specified or not s.tr_Reg:
                    continue
                if (uhr[U(I)] > 0.0 and mT[X(i)] < 0.5) or (uhr[U(I)] < 0.0 and mT[X(i + 1)] < 0.5):
                    uhh[I] = uhr[U(I)]
                    for reg in s.tr_Reg:
                        m = reg["ntr_index"] - 1
                        flux[m][I] = uhh[I] * (reg["tres"][k, j - s.HI['jsd'], 0] if reg.get("tres") is not None else reg["OBC_inflow_conc"])
    for I in range(is_ - 1, ie + 1):                                  # :632-635
        uhr[U(I)] = uhr[U(I)] - uhh[I]
        if abs(uhr[U(I)]) < uh_neglect[U(I)]:
            uhr[U(I)] = 0.0
    for i in range(is_, ie + 1):                                      # :636-662
        I = i
        if uhh[I] != 0.0 or uhh[I - 1] != 0.0:
            do_i = True
            hlst = hprev[X(i)]
            hprev[X(i)] = hprev[X(i)] - (uhh[I] - uhh[I - 1])
            if hprev[X(i)] <= 0.0:
                do_i = False
            elif hprev[X(i)] < h_neglect * areaT[X(i)]:
                hlst = hlst + (h_neglect * areaT[X(i)] - hprev[X(i)]); Ihnew = 1.0 / (h_neglect * areaT[X(i)])
            else:
                Ihnew = 1.0 / hprev[X(i)]
            if do_i and Ihnew > 0.0:
                for m in range(ntr):
                    T[m][X(i)] = (T[m][X(i)] * hlst - (flux[m][I] - flux[m][I - 1])) * Ihnew
    return slope


@pytest.mark.parametrize("side", ["W", "E"])
def test_oracle_advect_x_row_equals_a_second_transcription_of_the_reference_with_land_outside_a_segment(side):
    """a segment two columns inside the domain with LAND in the column outside it (the case of ADVICE r04): vhtr = 0, one pass of advect_x"""
    ni, nj, nk = 14, 6, 2
    g = synth.make_grid(ni, nj, nk, halo=4, land_frac=0.0, seed=77, reentrant_x=False, reentrant_y=False)
    A = 3 if side == "W" else ni - 3          # the segment's face I (global index = Fortran I of an unshifted domain)
    segs = [f"I={A},J=N:0,FLATHER,ORLANSKI" if side == "W" else f"I={A},J=0:N,FLATHER,ORLANSKI"]
    OBC = ocean_OBC_type(g, segs)
    s = OBC.segment[0]
    assert s.direction == (_abi.OBC_DIRECTION_W if side == "W" else _abi.OBC_DIRECTION_E)
    # land outside the segment: the columns beyond the outside cell are masked and the face beyond the outside cell is a wall
    m = {n: np.asarray(g.metrics[n]).copy() for n in ("mask2dT", "mask2dCu", "dy_Cu", "mask2dCv", "dx_Cv")}
    Ic = s.HI['IsdB'] - g.isd            # h-column of cell I
    out = slice(0, Ic + 1) if side == "W" else slice(Ic + 1, None)
    m["mask2dT"][:, out] = 0.0; m["mask2dCv"][:, out] = 0.0; m["dx_Cv"][:, out] = 0.0
    uc = s.HI['IsdB'] - g.isd + 1        # u-column of face I
    if side == "W":
        m["mask2dCu"][:, :uc] = 0.0; m["dy_Cu"][:, :uc] = 0.0
    else:
        m["mask2dCu"][:, uc + 1:] = 0.0; m["dy_Cu"][:, uc + 1:] = 0.0
    for n, a in m.items():
        g.set_metric(n, a)
    open_faces(g, OBC)
    assert np.asarray(g.mask2dCu)[4, uc] == 1.0 and np.asarray(g.mask2dCu)[4, uc + (-1 if side == "W" else 1)] == 0.0
    rng = np.random.default_rng(5)
    h = np.ascontiguousarray(50.0 + 10.0 * rng.random(g.shape3(_abi.POS_H)))
    T1 = np.ascontiguousarray(10.0 + np.cumsum(rng.random(g.shape3(_abi.POS_H)), axis=2))      # monotone in i: the slopes do not vanish
    T2 = np.ascontiguousarray(rng.random(g.shape3(_abi.POS_H)))
    areaT = np.asarray(g.areaT)
    uhtr = np.ascontiguousarray((0.02 + 0.01 * rng.random(g.shape3(_abi.POS_U))) * (1.0 if side == "W" else -1.0) * areaT.mean() * 50.0
                                * np.asarray(g.mask2dCu)[None])
    vhtr = g.zeros3(_abi.POS_V)
    # the reservoir lies between the tracer of the cells either side of the outside cell: the profile stays monotone across the segment, so
    # a slope there vanishes only through its masks
    xo = s.HI['IsdB'] - g.isd + (0 if side == "W" else 1)      # h-column of the outside cell
    tres = 0.5 * (T1[:, s.HI['jsd'] - g.jsd:s.HI['jed'] - g.jsd + 1, xo - 1] + T1[:, s.HI['jsd'] - g.jsd:s.HI['jed'] - g.jsd + 1, xo + 1])
    s.tr_Reg = [dict(ntr_index=1, tres=np.ascontiguousarray(tres[:, :, None])), dict(ntr_index=2, OBC_inflow_conc=0.7)]
    assert s.tr_Reg[0]["tres"].shape == s.normal_vel.shape
    tr = [T1.copy(), T2.copy()]
    uhr_o = g.zeros3(_abi.POS_U); vhr_o = g.zeros3(_abi.POS_V)
    st = orc.advect_tracer(g, h, uhtr, vhtr, 3600.0, 900.0, "PLM", tr, x_first=True, uhr_out=uhr_o, vhr_out=vhr_o, OBC=OBC, max_iter=1)
    # the second transcription: hprev (:160-164), uh_neglect (:175-177), then the row loop for every (j, k)
    want = [T1.copy(), T2.copy()]
    first_inside = s.HI['IsdB'] + 1 if side == "W" else s.HI['IsdB']
    slopes_seen = []
    for k in range(nk):
        for j in range(g.jsc, g.jec + 1):
            jj = j - g.jsd
            hprev = np.zeros(g.nih); uhr = uhtr[k, jj].copy()
            for i in range(g.isc, g.iec + 1):
                x = i - g.isd
                hp = max(0.0, areaT[jj, x] * h[k, jj, x] + ((uhr[x + 1] - uhr[x]) + (0.0 - 0.0)))
                hprev[x] = hp + max(0.0, 1.0e-13 * hp - areaT[jj, x] * h[k, jj, x])
            uh_neglect = np.zeros(g.nih + 1)
            for I in range(g.isd, g.ied):
                uh_neglect[I - g.isd + 1] = g.H_subroundoff * min(areaT[jj, I - g.isd], areaT[jj, I - g.isd + 1])
            rows = [want[0][k, jj], want[1][k, jj]]
            sl = ref_advect_x_row_plm(g, OBC, j, k, rows, hprev, uhr, uh_neglect)
            slopes_seen.append((sl[0][first_inside], sl[0][s.HI['IsdB'] + (0 if side == "W" else 1)]))
            cols = slice(g.isc - 1 - g.isd + 1, g.iec - g.isd + 2)      # the faces I = is-1 .. ie the row loop owns
            assert bits_equal(uhr[cols], uhr_o[k, jj][cols]), (k, j)
    # the first cell inside keeps its slope although the face beyond the outside cell is land, and the outside cell has none (the reading
    # ADVICE r04 corrected: with the masks of the faces IsdB, IsdB-1 for all three cells a western segment lost the first and an eastern
    # one gave the second)
    assert all(a != 0.0 and b == 0.0 for a, b in slopes_seen)
    for mtr in range(2):
        assert bits_equal(want[mtr], tr[mtr]), (side, mtr, np.argwhere(want[mtr] != tr[mtr])[:4])
    assert not bits_equal(tr[0], T1)


# ---- the library against the oracle, on the GPU ----

def run_hip(g, case, OBC, scheme, space, x_first=None, max_iter=None, conc_underflow=None):
    import torch
    from mom6_amd.tracer_advect import DeviceGrid, advect_tracer, tracer_advect_init
    dev = space == "device"
    X = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if dev else (lambda a: a.copy())
    N = (lambda a: a.cpu().numpy()) if dev else (lambda a: a)
    if OBC is not None and dev:      # the reservoirs in the memory space of the call
        for s in OBC.segment:
            for t in (s.tr_Reg or []):
                if t.get("tres") is not None and not hasattr(t["tres"], "data_ptr"):
                    t["tres"] = X(t["tres"])
    dg = DeviceGrid(g)
    CS = tracer_advect_init(900.0, scheme)
    tr = [X(t) for t in case["tr"]]
    uhr, vhr = X(g.zeros3(_abi.POS_U)), X(g.zeros3(_abi.POS_V))
    st = advect_tracer(X(case["h_end"]), X(case["uhtr"]), X(case["vhtr"]), OBC, 3600.0, dg, CS, tr, x_first_in=x_first, uhr_out=uhr, vhr_out=vhr,
                       max_iter_in=max_iter, conc_underflow=conc_underflow)
    dg.sync()
    out = dict(tr=[N(t) for t in tr], uhr=N(uhr), vhr=N(vhr), stats=st)
    dg.close()
    return out


def same(a, b, what):
    for m, (x, y) in enumerate(zip(a["tr"], b["tr"])):
        assert bits_equal(x, y), (what, "tracer", m + 1, np.argwhere(x != y)[:4])
    assert bits_equal(a["uhr"], b["uhr"]), (what, "uhr", np.argwhere(a["uhr"] != b["uhr"])[:4])
    assert bits_equal(a["vhr"], b["vhr"]), (what, "vhr", np.argwhere(a["vhr"] != b["vhr"])[:4])
    assert a["stats"].iterations == b["stats"].iterations and a["stats"].domore_remaining == b["stats"].domore_remaining, what


@pytest.mark.gpu
@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("x_first", [True, False])
def test_gpu_general_kernels_match_oracle_in_a_closed_domain(scheme, x_first, monkeypatch):
    """the general kernels (a thread a face, a thread a cell) without any OBC: the arithmetic of the marching ones"""
    monkeypatch.setenv("MOM6HIP_ADV_GENERIC", "1")
    for (ni, nj, nk, seed) in [(22, 16, 3, 3), (150, 40, 2, 5)]:
        g, case, _ = adv_obc_case([], ni=ni, nj=nj, nk=nk, seed=seed, registry=False)
        ref = run(g, case, None, scheme, x_first=x_first, conc_underflow=[1e-3, 0.0, 0.0])
        for space in ("device", "host"):
            same(run_hip(g, case, None, scheme, space, x_first=x_first, conc_underflow=[1e-3, 0.0, 0.0]), ref, (scheme, x_first, ni, space))


@pytest.mark.gpu
@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("x_first", [True, False])
@pytest.mark.parametrize("space", ["device", "host"])
def test_gpu_advect_tracer_with_segment_registries_matches_oracle_bitwise(scheme, x_first, space):
    for (ni, nj, nk, seed) in [(22, 16, 3, 3), (150, 40, 2, 5), (64, 48, 4, 9)]:
        g, case, OBC = adv_obc_case(SEGS, ni=ni, nj=nj, nk=nk, seed=seed)
        ref = run(g, case, OBC, scheme, x_first=x_first)
        closed = run(g, case, None, scheme, x_first=x_first)
        assert not bits_equal(ref["tr"][0], closed["tr"][0])
        same(run_hip(g, case, OBC, scheme, space, x_first=x_first), ref, (scheme, x_first, ni, space))


@pytest.mark.gpu
def test_gpu_advect_tracer_registries_few_iterations_and_no_registry():
    g, case, OBC = adv_obc_case(SEGS)
    ref = run(g, case, OBC, "PPM", max_iter=1)
    same(run_hip(g, case, OBC, "PPM", "host", max_iter=1), ref, "max_iter=1")
    g, case, OBC = adv_obc_case(SEGS, registry=False)
    same(run_hip(g, case, OBC, "PPM", "device"), run(g, case, None, "PPM"), "no registry")


@pytest.mark.gpu
def test_gpu_advect_tracer_refuses_bad_registries():
    from mom6_amd._lib import Mom6HipError
    g, case, OBC = adv_obc_case(SEGS)
    OBC.segment[0].tr_Reg[0]["ntr_index"] = 7
    with pytest.raises(Mom6HipError, match="names tracer"):
        run_hip(g, case, OBC, "PLM", "host")
    g, case, OBC = adv_obc_case(["I=9,J=0:N,SIMPLE", "I=11,J=0:N,SIMPLE"])
    with pytest.raises(Mom6HipError, match="closer than four cells"):
        run_hip(g, case, OBC, "PLM", "host")
    g, case, OBC = adv_obc_case(SEGS)
    OBC.segment[0].tr_Reg[0]["tres"] = np.zeros((3, 2, 2))
    with pytest.raises(Mom6HipError, match="shape"):
        run_hip(g, case, OBC, "PLM", "host")


# ---- update_segment_tracer_reservoirs (src/core/MOM_open_boundary.F90:5373), called after advect_tracer in step_MOM_tracer_dyn (MOM.F90:1447) ----

def reservoir_case(InvL=(0.0, 0.0), seed=3, segs=SEGS, **kw):
    g, case, OBC = adv_obc_case(segs, seed=seed, **kw)
    rng = np.random.default_rng(seed + 50)
    for s in OBC.segment:
        if not s.on_pe:
            continue
        s.Tr_InvLscale_in, s.Tr_InvLscale_out = InvL
        s.tr_Reg = [dict(ntr_index=1, tres=5.0 + rng.random(s.normal_vel.shape), t=7.0 + rng.random(s.normal_vel.shape)),
                    dict(ntr_index=3, OBC_inflow_conc=0.25),      # (no reservoir: not updated)
                    dict(ntr_index=2, tres=1.0 + rng.random(s.normal_vel.shape), t=2.0 + rng.random(s.normal_vel.shape), resrv_lfac_in=0.5, resrv_lfac_out=2.0)]
    return g, case, OBC


def seg_faces(g, s, a3):
    """the values of a 3-D face field on a segment's faces, in the layout of its own arrays"""
    hi = s.HI
    if s.is_E_or_W:
        return a3[:, hi["jsd"] - 1:hi["jed"], hi["IsdB"]:hi["IsdB"] + 1]
    return a3[:, hi["JsdB"]:hi["JsdB"] + 1, hi["isd"] - 1:hi["ied"]]


def seg_inside(g, s, a3):
    """the values of a 3-D cell field in the cells inside a segment"""
    hi = s.HI
    plus = s.direction in (_abi.OBC_DIRECTION_E, _abi.OBC_DIRECTION_N)
    if s.is_E_or_W:
        i = hi["IsdB"] - 1 + (0 if plus else 1)      # cell I (E) or I + 1 (W), 1-based -> 0-based
        return a3[:, hi["jsd"] - 1:hi["jed"], i:i + 1]
    j = hi["JsdB"] - 1 + (0 if plus else 1)
    return a3[:, j:j + 1, hi["isd"] - 1:hi["ied"]]


@pytest.mark.parametrize("InvL", [(0.0, 0.0), (1.0e-4, 3.0e-5), (0.0, 3.0e-5)])
def test_oracle_reservoirs_take_the_inside_value_on_outflow_and_the_external_one_on_inflow(InvL):
    g, case, OBC = reservoir_case(InvL)
    before = [[None if t.get("tres") is None else t["tres"].copy() for t in s.tr_Reg] for s in OBC.segment]
    orc.update_segment_tracer_reservoirs(g, case["uhtr"], case["vhtr"], case["h_end"], OBC, 3600.0, case["tr"])
    changed = 0
    for s, b in zip(OBC.segment, before):
        xr = seg_faces(g, s, case["uhtr"] if s.is_E_or_W else case["vhtr"])
        out = (xr > 0) if s.direction in (_abi.OBC_DIRECTION_E, _abi.OBC_DIRECTION_N) else (xr < 0)      # flow out of the domain, into the reservoir
        wet = seg_inside(g, s, np.broadcast_to(np.asarray(g.mask2dT)[None], case["h_end"].shape)) > 0
        for q, t in enumerate(s.tr_Reg):
            if t.get("tres") is None:
                continue
            inside = seg_inside(g, s, case["tr"][t["ntr_index"] - 1])
            new, old = t["tres"], b[q]
            assert bits_equal(new[~wet], old[~wet])      # (a land cell inside: skipped :5426)
            changed += int((new != old).sum())
            lo, hi_ = np.minimum(old, np.minimum(inside, t["t"])), np.maximum(old, np.maximum(inside, t["t"]))
            assert np.all(new >= lo - 1e-12) and np.all(new <= hi_ + 1e-12)      # a backward-Euler blend of the three
            if InvL[1] == 0.0:      # no length scale outwards: the reservoir takes the value inside at once
                m = wet & out & (xr != 0)
                assert np.allclose(new[m], inside[m], rtol=0, atol=1e-12)
            if InvL[0] == 0.0:      # none inwards: the external value at once
                m = wet & ~out & (xr != 0)
                assert np.allclose(new[m], t["t"][m], rtol=0, atol=1e-12)
    assert changed > 0


def test_oracle_reservoirs_turn_with_the_grid():
    g, case, OBC = reservoir_case((1.0e-4, 3.0e-5), segs=SEGS_TURN)
    gr = rotate_grid(g)
    OBCr = ocean_OBC_type(gr, turned_segments(SEGS_TURN, g.ni, g.nj))
    T_ = lambda a, s: np.ascontiguousarray(np.swapaxes(a, 1, 2) if s.is_E_or_W else np.swapaxes(a, 1, 2)[:, ::-1, :])
    for s, sr in zip(OBC.segment, OBCr.segment):
        sr.Tr_InvLscale_in, sr.Tr_InvLscale_out = s.Tr_InvLscale_in, s.Tr_InvLscale_out
        sr.tr_Reg = [{k: (T_(v, s) if isinstance(v, np.ndarray) else v) for k, v in t.items()} for t in s.tr_Reg]
    ur, vr = rot_vector(case["uhtr"], case["vhtr"])
    orc.update_segment_tracer_reservoirs(g, case["uhtr"], case["vhtr"], case["h_end"], OBC, 3600.0, case["tr"])
    orc.update_segment_tracer_reservoirs(gr, ur, vr, rot(case["h_end"]), OBCr, 3600.0, [rot(t) for t in case["tr"]])
    for s, sr in zip(OBC.segment, OBCr.segment):
        for t, tr_ in zip(s.tr_Reg, sr.tr_Reg):
            if t.get("tres") is not None:
                assert bits_equal(T_(t["tres"], s), tr_["tres"])


@pytest.mark.gpu
@pytest.mark.parametrize("InvL", [(0.0, 0.0), (1.0e-4, 3.0e-5), (0.0, 3.0e-5)])
@pytest.mark.parametrize("space", ["device", "host"])
def test_gpu_update_segment_tracer_reservoirs_matches_oracle_bitwise(InvL, space):
    import copy
    import torch
    from mom6_amd.open_boundary import update_segment_tracer_reservoirs
    from mom6_amd.tracer_advect import DeviceGrid
    for (ni, nj, nk, seed) in [(22, 16, 3, 3), (150, 40, 2, 5)]:
        g, case, OBC = reservoir_case(InvL, seed=seed, ni=ni, nj=nj, nk=nk)
        ref = copy.deepcopy(OBC)
        orc.update_segment_tracer_reservoirs(g, case["uhtr"], case["vhtr"], case["h_end"], ref, 3600.0, case["tr"])
        dev = space == "device"
        X = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if dev else (lambda a: a)
        if dev:
            OBC.cuda()
        dg = DeviceGrid(g)
        update_segment_tracer_reservoirs(dg, X(case["uhtr"]), X(case["vhtr"]), X(case["h_end"]), OBC, 3600.0, [X(t) for t in case["tr"]])
        dg.sync()
        n = 0
        for s, sr in zip(OBC.segment, ref.segment):
            for t, tr_ in zip(s.tr_Reg, sr.tr_Reg):
                if t.get("tres") is not None:
                    got = t["tres"].cpu().numpy() if dev else t["tres"]
                    assert bits_equal(got, tr_["tres"]), (InvL, space, ni, np.argwhere(got != tr_["tres"])[:4])
                    n += 1
        assert n == 2 * len(OBC.segment)
        dg.close()
