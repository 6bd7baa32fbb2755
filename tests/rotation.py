"""A quarter turn of a grid and of the fields on it (the reference's rotational-reproducibility test, ROTATE_INDEX /
.testing `make test.rotate`: MOM6's expressions are parenthesised so that answers reproduce to the bit under index
rotation).  New axes: x' = y, y' = -x, so on arrays indexed [j, i]

    rot(A) = A.T[::-1, :]          R[j', i'] = A[j = i', i = n_i - 1 - j']

maps h -> h, q -> q, u-point arrays onto v'-point arrays and v-point arrays onto u'-point arrays; a vector (u, v) becomes
u' = rot(v), v' = -rot(u); dx and dy metrics swap."""
import numpy as np

from mom6_amd import _abi
from mom6_amd.grid import Grid


def rot(a):
    a = np.asarray(a)
    return np.ascontiguousarray(np.swapaxes(a, -1, -2)[..., ::-1, :])


def rot_vector(u, v):
    """(u', v') of the turned frame"""
    return rot(v), np.ascontiguousarray(-rot(u))


def unrot(a):
    a = np.asarray(a)
    return np.ascontiguousarray(np.swapaxes(a[..., ::-1, :], -1, -2))


def unrot_vector(up, vp):
    """(u, v) of the original frame from (u', v')"""
    return np.ascontiguousarray(-unrot(vp)), unrot(up)


_SWAP = {"dx": "dy", "dy": "dx", "Idx": "Idy", "Idy": "Idx"}


def _turned_name(n):
    """name of the metric of the turned grid that rot(metric n) is"""
    for stem, pos_from, pos_to in (("T", "T", "T"), ("Cu", "Cu", "Cv"), ("Cv", "Cv", "Cu"), ("Bu", "Bu", "Bu")):
        if n.endswith(stem):
            head = n[: -len(stem)]
            if head in ("dy_", "dx_"):      # dy_Cu <-> dx_Cv
                return ("dx_" if head == "dy_" else "dy_") + pos_to
            return _SWAP.get(head, head) + pos_to
    return n      # bathyT is caught above ("T"); CoriolisBu ("Bu")


def rotate_grid(g: Grid) -> Grid:
    r = Grid(ni=g.nj, nj=g.ni, nk=g.nk, halo=g.halo, reentrant_x=g.reentrant_y, reentrant_y=g.reentrant_x,
             first_direction=(g.first_direction + 1) % 2, Angstrom_H=g.Angstrom_H, H_to_Z=g.H_to_Z, Z_to_H=g.Z_to_H, g_Earth=g.g_Earth,
             Rho0=g.Rho0)
    for n, a in g.metrics.items():
        r.set_metric(_turned_name(n), rot(a))
    return r
