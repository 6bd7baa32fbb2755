"""TEST INFRASTRUCTURE for TRIPOLAR_N: a fold-symmetric doubled domain and its folded half.

A tripolar grid is its own northern neighbour turned by half a turn: cell (i, nj+m) is cell (ni+1-i, nj+1-m).  The same
physics can therefore be run without any fold on the UNFOLDED domain of 2*nj rows whose northern half is the southern half
turned about the centre -- provided metrics, state and forcing have that symmetry, which every operator then preserves bit
for bit (tests/test_properties.py: the operators turn with the grid exactly).  The folded run must reproduce the southern half
of the unfolded run: this pins the halo rules of the fold (which FMS owns and the reference does not hold) and the polarity
swaps of MOM_barotropic.F90:1471-1475, :4036-4064 to the geometry of the fold."""
import numpy as np

from mom6_amd import _abi, synth


def grids(ni=20, nj=8, nk=3, seed=3, land_frac=0.2, **kw):
    g2 = synth.make_grid(ni, 2 * nj, nk, land_frac=land_frac, seed=seed, reentrant_x=True, fold_symmetric=True, **kw)
    return synth.fold_of(g2), g2


def symmetrize(g2, a, pos, vector=False):
    """Give a data-domain field of the unfolded grid the symmetry of the fold (northern half := image of the southern half,
    the rows on the centre line (anti)symmetrised), then refill its halos."""
    import torch
    a = np.array(a, copy=True)
    sj, si = g2.csl(pos)
    c = a[..., sj, si]
    nj = g2.nj // 2
    s = -1.0 if vector else 1.0
    if pos in (_abi.POS_V, _abi.POS_Q):      # rows J = 0..2nj, centre line J = nj
        c[..., nj + 1:, :] = s * c[..., :nj, :][..., ::-1, ::-1]
        c[..., nj, :] = 0.5 * (c[..., nj, :] + s * c[..., nj, ::-1])
    else:
        c[..., nj:, :] = s * c[..., :nj, :][..., ::-1, ::-1]
    a[..., sj, si] = c
    return synth.fill_halo(g2, torch.from_numpy(a), pos).numpy()


def folded(g, a2, pos):
    """the folded grid's data-domain array of an unfolded field: the southern rows, the halo beyond the fold included"""
    return np.ascontiguousarray(a2[..., :g.shape2(pos)[0], :])


def states(g, g2, seed=4, umax=0.1):
    """A fold-symmetric dynamical state + wind stress on the unfolded grid and its folded half."""
    d2 = {k: v.numpy() for k, v in synth.make_dynamics_state(g2, seed=seed, umax=umax, eta_amp=0.2).items()}
    P = {"u": (_abi.POS_U, True), "v": (_abi.POS_V, True), "uh": (_abi.POS_U, True), "vh": (_abi.POS_V, True)}
    for k in list(d2):
        pos, vec = P.get(k, (_abi.POS_H, False))
        d2[k] = symmetrize(g2, d2[k], pos, vec)
    d2["v"] = d2["v"] * g2.mask2dCv[None]; d2["u"] = d2["u"] * g2.mask2dCu[None]
    yy = np.linspace(0.0, np.pi, g2.shape2(_abi.POS_U)[0])
    taux2 = symmetrize(g2, 0.1 * np.cos(2 * yy)[:, None] * g2.mask2dCu, _abi.POS_U, True)
    tauy2 = symmetrize(g2, 0.02 * np.sin(3 * yy)[:, None][: g2.shape2(_abi.POS_V)[0]] * np.ones(g2.shape2(_abi.POS_V)) * g2.mask2dCv
                       if False else 0.02 * g2.mask2dCv * np.sin(np.linspace(0.0, 3.0, g2.shape2(_abi.POS_V)[1]))[None, :], _abi.POS_V, True)
    d = {k: folded(g, v, P.get(k, (_abi.POS_H, False))[0]) for k, v in d2.items()}
    return d, d2, (folded(g, taux2, _abi.POS_U), folded(g, tauy2, _abi.POS_V)), (taux2, tauy2)
