"""Machine-independent synthetic inputs for the golden-digest parity tests (TEST INFRASTRUCTURE).

The digests under tests/golden/ are SHA-256 sums of the oracle's output bits, made in the build container
(tools/make_golden_digests.py); the GPU box has to regenerate *bit-identical* inputs.  libm functions (sin, exp,
normal deviates through log) may differ in the last bit between CPUs, so everything here uses only operations IEEE-754
defines exactly -- + - * / sqrt floor min max compare -- applied element-wise in numpy (no fused contraction, no
reductions whose order depends on the SIMD width: vertical sums are explicit k loops), and a counter-based
integer hash (splitmix64 finaliser) for noise.  Grids and states have the same structure as mom6_amd/synth.py.
"""
from __future__ import annotations

import numpy as np

from mom6_amd import _abi
from mom6_amd.grid import Grid

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def hash01(shape, seed):
    """uniform [0,1) doubles, a pure function of (flat index, seed)"""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) + np.uint64((int(seed) * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)).reshape(shape)


def noise(shape, seed):
    """uniform in [-1, 1)"""
    return 2.0 * hash01(shape, seed) - 1.0


def psin(t):
    """a period-1 sine look-alike from parabolas: exact arithmetic, C1, range [-1, 1]"""
    f = t - np.floor(t)
    return np.where(f < 0.5, 16.0 * f * (0.5 - f), -16.0 * (f - 0.5) * (1.0 - f))


def pcos(t):
    return psin(t + 0.25)


def ksum(a):
    """sum over the leading (layer) axis in k order"""
    out = np.zeros(a.shape[1:], dtype=np.float64)
    for k in range(a.shape[0]):
        out = out + a[k]
    return out


def make_grid(ni, nj, nk, halo=4, land_frac=0.3, seed=7, reentrant_x=True, reentrant_y=False, max_depth=5500.0,
              first_direction=0, rough_noise=0.04, beta_plane=False, uniform=False, flat_bottom=False, spacing=None) -> Grid:
    """Mercator-like C-grid (dx shrinking polewards), bowl bathymetry with ridges and noise, about land_frac land."""
    g = Grid(ni=ni, nj=nj, nk=nk, halo=halo, reentrant_x=reentrant_x, reentrant_y=reentrant_y, first_direction=first_direction)
    nih, njh, h = g.nih, g.njh, halo
    dy = 27000.0 * 1080.0 / max(nj, 270)
    dx0 = 27800.0 * 1440.0 / max(ni, 360)
    if spacing is not None:
        dx0 = dy = float(spacing)
    # latitude fraction in [-1, 1] of h rows and of v/q rows (clipped in the halos)
    yh = np.clip(((np.arange(njh) - h) + 0.5) / nj * 2.0 - 1.0, -1.0, 1.0)
    yv = np.clip(((np.arange(njh + 1) - h)) / nj * 2.0 - 1.0, -1.0, 1.0)
    cosl = (lambda y: 1.0 + 0.0 * y) if uniform else (lambda y: 1.0 - 0.78 * y * y)
    dx_h, dx_v = dx0 * cosl(yh), dx0 * cosl(yv)
    bc = lambda col, n: np.repeat(col[:, None], n, axis=1)
    dxT, dyT = bc(dx_h, nih), np.full((njh, nih), dy)
    dxCu, dyCu = bc(dx_h, nih + 1), np.full((njh, nih + 1), dy)
    dxCv, dyCv = bc(dx_v, nih), np.full((njh + 1, nih), dy)
    dxBu, dyBu = bc(dx_v, nih + 1), np.full((njh + 1, nih + 1), dy)

    X = ((np.arange(ni) + 0.5) / ni)[None, :] + np.zeros((nj, 1))
    Y = ((np.arange(nj) + 0.5) / nj)[:, None] + np.zeros((1, ni))
    bowl = np.sqrt(4.0 * Y * (1.0 - Y)) * (0.65 + 0.35 * pcos(X) * (1.0 - 2.0 * Y))
    rough = 0.12 * psin(3.0 * X + 0.16) * psin(2.0 * Y) + rough_noise * noise((nj, ni), seed)
    field = bowl + rough
    if land_frac > 0:
        thr = np.sort(field.ravel())[int(land_frac * field.size)]
    else:
        thr = field.min() - 0.05 * (field.max() - field.min())
    ocean = field > thr
    depth_c = np.where(ocean, max_depth * np.clip((field - thr) / (field.max() - thr), 0.02, 1.0), 0.0)
    if flat_bottom:
        depth_c = np.where(ocean, max_depth, 0.0)

    def embed_h(a_c, fill=0.0):
        a = np.full((njh, nih), fill, dtype=np.float64)
        a[h:h + nj, h:h + ni] = a_c
        if reentrant_x:
            a[h:h + nj, :h] = a_c[:, ni - h:]
            a[h:h + nj, h + ni:] = a_c[:, :h]
        if reentrant_y:
            a[:h, :] = a[nj:nj + h, :]
            a[h + nj:, :] = a[h:2 * h, :]
        return a

    mT = embed_h(ocean.astype(np.float64))
    bathyT = embed_h(depth_c)
    mCu = np.zeros((njh, nih + 1)); mCu[:, 1:nih] = mT[:, :-1] * mT[:, 1:]
    mCv = np.zeros((njh + 1, nih)); mCv[1:njh, :] = mT[:-1, :] * mT[1:, :]
    mBu = np.zeros((njh + 1, nih + 1))
    mBu[1:njh, 1:nih] = mT[:-1, :-1] * mT[:-1, 1:] * mT[1:, :-1] * mT[1:, 1:]
    if reentrant_x:
        for m in (mCu, mBu):
            m[:, 0] = m[:, ni]; m[:, -1] = m[:, -1 - ni]
    if reentrant_y:
        for m in (mCv, mBu):
            m[0, :] = m[nj, :]; m[-1, :] = m[-1 - nj, :]

    def inv(a):
        out = np.zeros_like(a)
        np.divide(1.0, a, out=out, where=a > 0)
        return out

    S = g.set_metric
    S("mask2dT", mT); S("bathyT", bathyT)
    S("dxT", dxT); S("dyT", dyT); S("IdxT", inv(dxT)); S("IdyT", inv(dyT)); S("areaT", dxT * dyT); S("IareaT", inv(dxT * dyT))
    S("mask2dCu", mCu); S("dxCu", dxCu); S("dyCu", dyCu); S("dy_Cu", dyCu * mCu); S("IdxCu", inv(dxCu)); S("IdyCu", inv(dyCu))
    S("areaCu", dxCu * dyCu); S("IareaCu", inv(dxCu * dyCu))
    S("mask2dCv", mCv); S("dxCv", dxCv); S("dyCv", dyCv); S("dx_Cv", dxCv * mCv); S("IdxCv", inv(dxCv)); S("IdyCv", inv(dyCv))
    S("areaCv", dxCv * dyCv); S("IareaCv", inv(dxCv * dyCv))
    S("mask2dBu", mBu); S("dxBu", dxBu); S("dyBu", dyBu); S("areaBu", dxBu * dyBu); S("IareaBu", inv(dxBu * dyBu))
    S("IdxBu", inv(dxBu)); S("IdyBu", inv(dyBu))
    f = (1.0e-4 + 2.0e-5 * yv) if beta_plane else 1.4e-4 * yv * (1.5 - 0.5 * yv * yv)
    S("CoriolisBu", bc(f, nih + 1))
    return g


def embed(g, a_c, pos):
    out = np.zeros(g.shape3(pos, a_c.shape[0]) if a_c.ndim == 3 else g.shape2(pos), dtype=np.float64)
    sj, si = g.csl(pos)
    out[..., sj, si] = a_c
    return out


def fill_halo(g, a, pos):
    """pass_var on the one-tile domain, in place"""
    xs = 1 if pos in (_abi.POS_U, _abi.POS_Q) else 0
    ys = 1 if pos in (_abi.POS_V, _abi.POS_Q) else 0
    h, ni, nj = g.halo, g.ni, g.nj
    if g.reentrant_x:
        a[..., h:h + nj + ys, :h] = a[..., h:h + nj + ys, ni:ni + h]
        a[..., h:h + nj + ys, h + ni + xs:] = a[..., h:h + nj + ys, h + xs:2 * h + xs]
    if g.reentrant_y:
        a[..., :h, :] = a[..., nj:nj + h, :]
        a[..., h + nj + ys:, :] = a[..., h + ys:2 * h + ys, :]
    return a


def nominal_dz(nk, total=5500.0):
    kk = (np.arange(nk) + 0.5) / nk
    dz = 2.0 + 300.0 * kk * kk
    s = 0.0
    for x in dz:
        s = s + float(x)
    return dz * (total / s)


def make_state(g: Grid, seed=1, umax=0.1, eta_amp=0.2, vanish_frac=0.05, terrain_following=False, ntr=2):
    """h, u, v, T, S (+ ntr passive tracers) with valid halos: z*-like layers that vanish below the topography and in
    blobs (or terrain-following layers), filling the column up to a smooth free surface; smooth flow + noise."""
    nk, ni, nj = g.nk, g.ni, g.nj
    sjh, sih = g.csl(_abi.POS_H); sju, siu = g.csl(_abi.POS_U); sjv, siv = g.csl(_abi.POS_V)
    mT = g.mask2dT[sjh, sih]; depth = g.bathyT[sjh, sih]
    X = ((np.arange(ni) + 0.5) / ni)[None, None, :]
    Y = ((np.arange(nj) + 0.5) / nj)[None, :, None]
    K = ((np.arange(nk) + 0.5) / nk)[:, None, None]
    shp = (nk, nj, ni)
    dzn = nominal_dz(nk)[:, None, None]
    ztop = np.zeros((nk, 1, 1))
    for k in range(1, nk):
        ztop[k] = ztop[k - 1] + dzn[k - 1]
    if terrain_following:
        h0 = (dzn + np.zeros(shp)) * (depth[None] / 5500.0)
        h0 = np.maximum(h0 * (1.0 + 0.01 * noise(shp, seed + 1)), 0.0)
    else:
        h0 = np.maximum(np.minimum(dzn + np.zeros(shp), depth[None] - ztop), 0.0)
        if vanish_frac > 0:
            blob = psin(4.5 * X + 0.48 * K) * psin(3.5 * Y - 0.32 * K) + 0.3 * noise(shp, seed + 2)
            q = np.sort(blob.ravel()[:: max(1, blob.size // 200000)])
            q = q[min(len(q) - 1, int((1.0 - vanish_frac) * len(q)))]
            h0 = np.where(blob > q, 0.0, h0)
        h0 = np.maximum(h0 * (1.0 + 0.05 * noise(shp, seed + 1)), 0.0)
    h0 = np.where(h0 < 1.0e-3, g.Angstrom_H, h0)
    h0 = np.where(mT[None] > 0, h0, g.Angstrom_H)
    if eta_amp is not None:
        eta0 = eta_amp * psin(X[0]) * psin(0.5 * Y[0])
        tot = ksum(h0)
        h0 = np.where(mT[None] > 0, h0 * ((depth * g.Z_to_H + eta0) / np.maximum(tot, 1e-30))[None], h0)
    h = fill_halo(g, embed(g, h0, _abi.POS_H), _abi.POS_H)
    h = np.where(h <= 0, g.Angstrom_H, h)

    amp = umax / (1.0 + 20.0 * K * K)
    ue = amp * (0.7 * psin(Y + 0.3 * K) * pcos(X) + 0.3 * noise(shp, seed + 3))
    vn = amp * (0.7 * pcos(X - 0.2 * K) * psin(Y) + 0.3 * noise(shp, seed + 4))
    mCu = g.mask2dCu[sju, siu]; mCv = g.mask2dCv[sjv, siv]
    u_c = np.concatenate([ue[:, :, -1:], ue], 2) * mCu[None]
    v_c = np.concatenate([vn[:, -1:, :], vn], 1) * mCv[None]
    u = fill_halo(g, embed(g, u_c, _abi.POS_U), _abi.POS_U)
    v = fill_halo(g, embed(g, v_c, _abi.POS_V), _abi.POS_V)
    zmid = ztop + 0.5 * dzn
    if terrain_following:
        zmid = zmid * (depth[None] / 5500.0)
    T = 20.0 / ((1.0 + zmid / 1400.0) * (1.0 + zmid / 1400.0)) + 2.0 * pcos(0.5 * Y) + 0.01 * noise(shp, seed + 5)
    S = 35.0 + 0.5 * psin(X) * psin(0.5 * Y) + 0.01 * noise(shp, seed + 6) + np.zeros(shp)
    out = {"h": np.ascontiguousarray(h), "u": np.ascontiguousarray(u), "v": np.ascontiguousarray(v)}
    out["T"] = np.ascontiguousarray(fill_halo(g, embed(g, T * mT[None], _abi.POS_H), _abi.POS_H))
    out["S"] = np.ascontiguousarray(fill_halo(g, embed(g, S * mT[None], _abi.POS_H), _abi.POS_H))
    tr = []
    for m in range(ntr):
        if m % 2 == 0:      # a smooth blob
            r2 = (X - 0.4) * (X - 0.4) + (Y - 0.5) * (Y - 0.5)
            a = np.maximum(1.0 - r2 / 0.06, 0.0) * np.maximum(1.0 - r2 / 0.06, 0.0) + np.zeros(shp)
        else:               # a step
            a = ((X > 0.3) & (X < 0.6) & (Y > 0.2) & (Y < 0.7)).astype(np.float64) + np.zeros(shp)
        tr.append(np.ascontiguousarray(fill_halo(g, embed(g, a * mT[None], _abi.POS_H), _abi.POS_H)))
    out["tr"] = tr
    return out


def wind_stress(g: Grid, amp=0.1):
    ny = g.shape2(_abi.POS_U)[0]
    yy = (np.arange(ny) - g.halo + 0.5) / g.nj
    taux = np.ascontiguousarray(amp * pcos(yy)[:, None] * g.mask2dCu)
    return taux, g.zeros2(_abi.POS_V)


def bbl_arrays(g: Grid, seed=9):
    su, sv = g.shape2(_abi.POS_U), g.shape2(_abi.POS_V)
    return dict(Kv_bbl_u=np.ascontiguousarray(1.0e-3 * (0.5 + hash01(su, seed))), Kv_bbl_v=np.ascontiguousarray(1.0e-3 * (0.5 + hash01(sv, seed + 1))),
                bbl_thick_u=np.ascontiguousarray(2.0 + 8.0 * hash01(su, seed + 2)), bbl_thick_v=np.ascontiguousarray(2.0 + 8.0 * hash01(sv, seed + 3)))


def make_advection_inputs(g: Grid, h, seed=21, cfl=0.15, hot_frac=2.0e-4, hot_cfl=0.65):
    """uhtr, vhtr, h_end for advect_tracer on the thicknesses h: transports c * min(neighbouring volumes), |c| <= cfl, plus
    a fraction hot_frac of cells that diverge strongly in x (the flux limiter postpones part of their transport to a later
    iteration, src/tracer/MOM_tracer_advect.F90:494-497)."""
    nk, ni, nj = g.nk, g.ni, g.nj
    sjh, sih = g.csl(_abi.POS_H); sju, siu = g.csl(_abi.POS_U); sjv, siv = g.csl(_abi.POS_V)
    area = g.areaT[sjh, sih]
    mCu = g.mask2dCu[sju, siu]; mCv = g.mask2dCv[sjv, siv]
    X = ((np.arange(ni) + 0.5) / ni)[None, None, :]
    Y = ((np.arange(nj) + 0.5) / nj)[None, :, None]
    K = ((np.arange(nk) + 0.5) / nk)[:, None, None]
    shp = (nk, nj, ni)
    vol0 = area[None] * h[:, sjh, sih]
    cu = cfl * (0.7 * psin(Y + 0.3 * K) * pcos(X) + 0.3 * noise(shp, seed))
    cv = cfl * (0.7 * pcos(X - 0.2 * K) * psin(Y) + 0.3 * noise(shp, seed + 1))
    if hot_frac > 0:
        hc = hash01(shp, seed + 2) < hot_frac
        hw = np.roll(hc, -1, 2); hs = np.roll(hc, -1, 1)
        cu = np.where(hc, hot_cfl, cu); cu = np.where(hw & ~hc, -hot_cfl, cu)
        cv = np.where(hc, -0.5 * hot_cfl, cv); cv = np.where(hs & ~hc, 0.5 * hot_cfl, cv)
    vol_e = np.roll(vol0, -1, 2); vol_n = np.roll(vol0, -1, 1)
    uh_e = cu * np.minimum(vol0, vol_e) * mCu[None, :, 1:]
    vh_n = cv * np.minimum(vol0, vol_n) * mCv[None, 1:, :]
    for _ in range(4):
        uh_w = np.roll(uh_e, 1, 2); vh_s = np.roll(vh_n, 1, 1)
        out = np.maximum(uh_e, 0.0) + np.maximum(-uh_w, 0.0) + np.maximum(vh_n, 0.0) + np.maximum(-vh_s, 0.0)
        inn = np.maximum(-uh_e, 0.0) + np.maximum(uh_w, 0.0) + np.maximum(-vh_n, 0.0) + np.maximum(vh_s, 0.0)
        with np.errstate(over="ignore"):
            sc = np.minimum((0.8 * vol0 + inn) / np.maximum(out, 1e-300), 1.0)
        uh_e = np.where(uh_e >= 0, uh_e * sc, uh_e * np.roll(sc, -1, 2))
        vh_n = np.where(vh_n >= 0, vh_n * sc, vh_n * np.roll(sc, -1, 1))
    uh = np.concatenate([uh_e[:, :, -1:], uh_e], 2) * mCu[None]
    vh = np.concatenate([vh_n[:, -1:, :], vh_n], 1) * mCv[None]
    div = (uh[:, :, 1:] - uh[:, :, :-1]) + (vh[:, 1:, :] - vh[:, :-1, :])
    h_end = np.maximum(vol0 - div, 0.0) / area[None]
    h_end = np.where(h_end < g.Angstrom_H, g.Angstrom_H, h_end)
    return dict(h_end=np.ascontiguousarray(embed(g, h_end, _abi.POS_H)), uhtr=np.ascontiguousarray(embed(g, uh, _abi.POS_U)),
                vhtr=np.ascontiguousarray(embed(g, vh, _abi.POS_V)))


def make_phillips(ni=480, nj=320, nk=2, halo=4, max_depth=2000.0, jet_height=200.0, jet_width=0.08, noise_amp=1.0e-3, seed=31,
                  dT=10.0):
    """BASELINE configs[3], Phillips_2layer: a re-entrant zonal channel on a beta plane with a flat bottom, two layers
    (nk layers: the upper and the lower half of the column), an interface that rises northwards across a jet, the upper
    layer in thermal-wind balance with it, and millimetre-per-second noise (the set-up of
    src/user/Phillips_initialization.F90:37-128, :132-208, with tanh(y/L) replaced by y/sqrt(L^2 + y^2), which has the
    same shape and is exact arithmetic).  The reference runs it in layered mode with layer densities; here the layers
    carry T = 20 and 20 - dT under a LINEAR equation of state, which gives the same reduced gravity
    g' = g * 0.2 * dT / Rho0.  Returns (grid, state, g_prime)."""
    g = make_grid(ni, nj, nk, halo=halo, land_frac=0.0, seed=seed, reentrant_x=True, reentrant_y=False, max_depth=max_depth,
                  beta_plane=True, uniform=True, flat_bottom=True, rough_noise=0.0, spacing=12500.0)
    sjh, sih = g.csl(_abi.POS_H); sju, siu = g.csl(_abi.POS_U); sjv, siv = g.csl(_abi.POS_V)
    Y = ((np.arange(nj) + 0.5) / nj)[None, :, None] - 0.5
    shp = (nk, nj, ni)
    t = Y / jet_width
    prof = t / np.sqrt(1.0 + t * t)                       # the tanh look-alike
    dprof = 1.0 / ((1.0 + t * t) * np.sqrt(1.0 + t * t))    # its derivative with respect to t
    # interfaces: nk/2 layers above the jet interface, nk/2 below (:96-101)
    eta = np.zeros((nk + 1, nj, ni))
    eta[nk] = -max_depth
    for K in range(1, nk):
        e0 = -max_depth * (K / nk)
        eta[K] = np.minimum(np.maximum(e0 + jet_height * prof[0] * (1.0 - abs(2.0 * K / nk - 1.0)) + np.zeros((nj, ni)), -max_depth), 0.0)
    rho_T = 0.2
    g_prime = g.g_Earth * rho_T * dT / g.Rho0
    fh = 1.0e-4 + 2.0e-5 * (2.0 * Y)                       # Coriolis parameter of the h rows (make_grid's beta plane)
    dy = g.dyT[0, 0]
    # thermal wind (:176-190): u(k) = u(k+1) + g'/f * d(eta)/dy for the layers above the mid-depth interface
    shear = g_prime / fh * (jet_height * dprof / (jet_width * (nj * dy)))
    ue = np.zeros(shp)
    for k in range(nk // 2):
        ue[k] = shear[0] + np.zeros((nj, ni))
    # the free surface in geostrophic balance with the upper-layer flow: d(eta)/dy = -f u / g, summed row by row
    es = np.zeros(nj)
    for j in range(1, nj):
        es[j] = es[j - 1] - 0.5 * (fh[0, j, 0] * shear[0, j, 0] + fh[0, j - 1, 0] * shear[0, j - 1, 0]) * dy / g.g_Earth
    es = es - 0.5 * (es.min() + es.max())
    eta[0] = es[:, None] + np.zeros((nj, ni))
    h0 = np.maximum(eta[:-1] - eta[1:], g.Angstrom_H)
    h0 = h0 * (1.0 + 1.0e-7 * noise(shp, seed + 1))
    h = fill_halo(g, embed(g, h0, _abi.POS_H), _abi.POS_H)
    h = np.where(h <= 0, g.Angstrom_H, h)
    K = ((np.arange(nk) + 0.5) / nk)[:, None, None]
    ue = ue + noise_amp * K * noise(shp, seed + 2)
    vn = noise_amp * K * noise(shp, seed + 3)
    mCu = g.mask2dCu[sju, siu]; mCv = g.mask2dCv[sjv, siv]
    u = fill_halo(g, embed(g, np.concatenate([ue[:, :, -1:], ue], 2) * mCu[None], _abi.POS_U), _abi.POS_U)
    v = fill_halo(g, embed(g, np.concatenate([vn[:, -1:, :], vn], 1) * mCv[None], _abi.POS_V), _abi.POS_V)
    Tk = np.where(np.arange(nk) < nk // 2, 20.0, 20.0 - dT)[:, None, None]
    T = fill_halo(g, embed(g, Tk + np.zeros(shp), _abi.POS_H), _abi.POS_H)
    S = fill_halo(g, embed(g, 35.0 + np.zeros(shp), _abi.POS_H), _abi.POS_H)
    st = {"h": np.ascontiguousarray(h), "u": np.ascontiguousarray(u), "v": np.ascontiguousarray(v), "T": np.ascontiguousarray(T),
          "S": np.ascontiguousarray(S)}
    return g, st, g_prime
