"""mom6hip_chksum: MOM_checksums' bit-count checksum (MOM_checksums.F90:1387-1401, :2407-2414) of fields on the GPU against
the same sum formed with numpy on the host."""
import numpy as np
import pytest

from mom6_amd import _abi, synth

POP = np.array([bin(i).count("1") for i in range(256)], dtype=np.int64)


def ref_chksum(g, a, pos, di=0, dj=0, symmetric=False, scale=1.0):
    """subchk: sum of popcnt(transfer(abs(scale*x), 1_8)) over (isc+di : iec+di, jsc+dj : jec+dj), mod 1e9; min / max there"""
    xs = 1 if pos in (_abi.POS_U, _abi.POS_Q) else 0
    ys = 1 if pos in (_abi.POS_V, _abi.POS_Q) else 0
    h = g.halo
    i0, i1 = h + xs + di - (1 if symmetric and xs else 0), h + xs + g.ni + di
    j0, j1 = h + ys + dj - (1 if symmetric and ys else 0), h + ys + g.nj + dj
    sub = np.ascontiguousarray(a[..., j0:j1, i0:i1])
    bits = np.abs(scale * sub).view(np.uint8)
    return int(POP[bits].sum() % 1000000000), float(sub.min()), float(sub.max())


def test_reference_bitcount_of_known_values():
    """popcnt of the IEEE bits: 1.0 = 0x3FF0..., 2.0 = 0x4000..., 0.0, -1.0 (abs first), 0.1"""
    g = synth.make_grid(4, 3, 1, halo=2, land_frac=0.0)
    a = np.zeros(g.shape2(_abi.POS_H)); sj, si = g.csl(_abi.POS_H)
    a[sj, si] = np.array([[1.0, 2.0, 0.0, -1.0], [0.1, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0]])
    want = 10 + 1 + 0 + 10 + bin(np.float64(0.1).view(np.uint64)).count("1")
    assert ref_chksum(g, a, _abi.POS_H)[0] == want


@pytest.mark.gpu
@pytest.mark.parametrize("space", ["device", "host"])
def test_chksum_matches_host_sum(space):
    import torch
    from mom6_amd.checksums import Bchksum, chksum, hchksum, uchksum, vchksum
    from mom6_amd.tracer_advect import DeviceGrid
    g = synth.make_grid(70, 33, 5, land_frac=0.2, seed=3)
    d = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=4, umax=0.3, eta_amp=0.2).items()}
    q = np.random.default_rng(1).standard_normal(g.shape3(_abi.POS_Q))
    dg = DeviceGrid(g)
    put = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if space == "device" else (lambda a: np.ascontiguousarray(a))
    for name, a, pos in (("h", d["h"], _abi.POS_H), ("u", d["u"], _abi.POS_U), ("v", d["v"], _abi.POS_V), ("q", q, _abi.POS_Q),
                         ("h2d", d["h"][0], _abi.POS_H)):
        da = put(a)
        for di, dj, sym, scale in ((0, 0, False, 1.0), (1, -1, False, 1.0), (-2, 2, False, 0.5), (0, 0, True, 1.0), (2, 0, True, 3.0)):
            want = ref_chksum(g, a, pos, di, dj, sym, scale)
            got = chksum(da, pos, dg, di, dj, sym, scale)
            assert got == want, (name, di, dj, sym, scale, got, want)
    # the reference's calls: hchksum(h, "h", HI, haloshift=1) prints bc0 and the four corner shifts
    got = hchksum(put(d["h"]), "h", dg, haloshift=1)
    assert got == {n: ref_chksum(g, d["h"], _abi.POS_H, di, dj)[0]
                   for n, (di, dj) in (("bc0", (0, 0)), ("bcSW", (-1, -1)), ("bcSE", (1, -1)), ("bcNW", (-1, 1)), ("bcNE", (1, 1)))}
    got = uchksum(put(d["u"]), "u", dg, haloshift=2, symmetric=True, omit_corners=True)
    assert got["bcW"] == ref_chksum(g, d["u"], _abi.POS_U, -2, 0, True)[0] and set(got) == {"bc0", "bcS", "bcE", "bcW", "bcN"}
    assert vchksum(put(d["v"]), "v", dg)["bc0"] == ref_chksum(g, d["v"], _abi.POS_V)[0]
    assert Bchksum(put(q), "q", dg, symmetric=True)["bc0"] == ref_chksum(g, q, _abi.POS_Q, 0, 0, True)[0]
    dg.close()
