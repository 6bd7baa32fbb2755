"""reproducing_sum (src/framework/MOM_coms.F90:219, :318): the oracle against exact integer arithmetic -- an extended-fixed-point
sum of doubles whose bits all lie at or above 2^-138 is EXACT, so Python's rationals give the answer independently of any
restatement -- and mom6hip_reproducing_sum against the oracle, bit for bit, on one tile, through RCCL and on two tiles."""
from fractions import Fraction
from types import SimpleNamespace

import numpy as np
import pytest

from mom6_amd import _abi, synth
from oracle import orc

H, U, V, Q = _abi.POS_H, _abi.POS_U, _abi.POS_V, _abi.POS_Q
PREC = 1 << 46


def efp_value(ints):
    """the rational an EFP number stands for: sum ints(i) * 2^(46*(3-i)), i = 1..6"""
    return sum(Fraction(int(x)) * Fraction(2) ** (46 * (2 - i)) for i, x in enumerate(ints))


def ints_to_real(ints):
    r = 0.0
    for i, x in enumerate(ints):      # ints_to_real :545
        r = r + float(2.0 ** (46 * (2 - i))) * float(x)
    return r


def regular(ints):
    """regularize_ints' form (:643): limbs 2..6 below prec in magnitude, every non-zero limb of the sign of the value"""
    nz = [x for x in ints if x != 0]
    return all(abs(x) < PREC for x in ints[1:]) and (all(x > 0 for x in nz) or all(x < 0 for x in nz))


def plain(ni, nj):
    return SimpleNamespace(halo=0, ni=ni, nj=nj)


@pytest.mark.parametrize("shape,scale", [((1, 7, 9), 1.0), ((3, 40, 50), 1.0e8), ((2, 33, 17), 1.0e-12), ((4, 12, 31), 1.0e20)])
def test_oracle_sum_is_the_exact_sum(shape, scale):
    rng = np.random.default_rng(sum(shape))
    a = rng.standard_normal(shape) * scale * np.exp(rng.uniform(-20, 20, shape))
    a[:, ::3, ::4] = 0.0
    g = plain(shape[2], shape[1])
    exact = sum(Fraction(float(x)) for x in a.ravel())
    r = orc.reproducing_sum(g, a, H)
    assert efp_value(r["efp"]) == exact and regular(r["efp"])
    assert r["sum"] == ints_to_real(r["efp"])
    assert abs(Fraction(r["sum"]) - exact) <= abs(exact) * Fraction(1, 2 ** 51)
    b = orc.reproducing_sum(g, a, H, by_layer=True)
    for k in range(shape[0]):
        assert efp_value(b["efp_lay"][k]) == sum(Fraction(float(x)) for x in a[k].ravel()) and regular(b["efp_lay"][k])
        assert b["sums"][k] == ints_to_real(b["efp_lay"][k])
    assert efp_value(b["efp"]) == exact
    tot = 0.0
    for v in b["sums"]:
        tot = tot + v
    assert b["sum"] == tot            # :421-427: the floating-point sum of the layer values, k ascending


def test_oracle_sum_does_not_depend_on_the_order():
    rng = np.random.default_rng(5)
    a = rng.standard_normal((2, 30, 44)) * 10.0 ** rng.integers(-9, 9, (2, 30, 44))
    g = plain(44, 30)
    r = orc.reproducing_sum(g, a, H)
    p = rng.permutation(a.size)
    assert orc.reproducing_sum(g, a.ravel()[p].reshape(a.shape), H) == r
    assert orc.reproducing_sum(plain(30, 44), np.ascontiguousarray(a.transpose(0, 2, 1)), H) == r
    # the cancellation a naive sum loses: 1e16 + 1 - 1e16
    c = np.zeros((1, 1, 3)); c[0, 0] = (1.0e16, 1.0, -1.0e16)
    assert orc.reproducing_sum(plain(3, 1), c, H)["sum"] == 1.0 and float(np.sum(c)) != 1.0


def test_oracle_carries_rows_and_single_terms():
    """more than max_count_prec = 2^17 - 1 terms: by rows (:399 / :457), and a row too long for that one term at a time (:405)"""
    rng = np.random.default_rng(9)
    a = np.abs(rng.standard_normal((1, 400, 400))) * 2.0 ** 45          # every limb-3 term near 2^45: the row sums must carry
    r = orc.reproducing_sum(plain(400, 400), a, H)
    assert efp_value(r["efp"]) == sum(Fraction(float(x)) for x in a.ravel()) and regular(r["efp"])
    w = 1 << 17
    b = -np.abs(rng.standard_normal((1, 1, w))) * 2.0 ** 45
    r = orc.reproducing_sum(plain(w, 1), b, H)
    assert efp_value(r["efp"]) == sum(Fraction(float(x)) for x in b.ravel()) and regular(r["efp"])


def test_oracle_error_codes():
    g = plain(4, 2)
    a = np.ones((1, 2, 4))
    assert orc.reproducing_sum(g, a, H, return_err=True)["err"] == 0
    b = a.copy(); b[0, 1, 2] = np.nan
    r = orc.reproducing_sum(g, b, H, return_err=True)
    assert r["err"] == 2 and r["efp"] == [0] * 6 and r["sum"] == 0.0
    c = a.copy(); c[0, 0, 0] = 1.0e300                                   # no EFP representation: +1 (too large) +2 (overflow)
    assert orc.reproducing_sum(g, c, H, return_err=True)["err"] == 3
    for bad in (b, c):
        with pytest.raises(RuntimeError):
            orc.reproducing_sum(g, bad, H)


def fields(g, seed=1):
    rng = np.random.default_rng(seed)
    d = {k: v.numpy() for k, v in synth.make_dynamics_state(g, seed=seed, umax=0.3, eta_amp=0.2).items()}
    q = rng.standard_normal(g.shape3(Q)) * 10.0 ** rng.integers(-12, 12, g.shape3(Q))
    return (("h", d["h"], H), ("u", d["u"], U), ("v", d["v"], V), ("q", q, Q), ("h2d", d["h"][0] * 1.0e9, H), ("u2d", d["u"][1], U))


@pytest.mark.gpu
@pytest.mark.parametrize("space", ["device", "host"])
def test_reproducing_sum_matches_the_oracle(space):
    import torch
    from mom6_amd.checksums import substats
    from mom6_amd.coms import EFP_to_real, reproducing_sum
    from mom6_amd.tracer_advect import DeviceGrid
    g = synth.make_grid(70, 45, 5, land_frac=0.2, seed=3)
    dg = DeviceGrid(g)
    put = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()) if space == "device" else (lambda a: np.ascontiguousarray(a))
    for name, a, pos in fields(g):
        da = put(a)
        want = orc.reproducing_sum(g, a, pos)
        got = reproducing_sum(da, pos, dg)
        assert got.EFP_sum.v == want["efp"] and got.sum == want["sum"] and got.err == 0, name
        assert got.npoints == g.ni * g.nj * (1 if a.ndim == 2 else a.shape[0])
        assert EFP_to_real(got.EFP_sum) == got.sum
        wl = orc.reproducing_sum(g, a, pos, by_layer=True)
        gl = reproducing_sum(da, pos, dg, by_layer=True)
        assert [e.v for e in gl.EFP_lay_sums] == wl["efp_lay"] and gl.sums == wl["sums"] and gl.sum == wl["sum"] and gl.EFP_sum.v == wl["efp"], name
        # subStats: the mean MOM_checksums prints
        sj, si = g.csl(H)
        xs, ys = (1 if pos in (U, Q) else 0), (1 if pos in (V, Q) else 0)
        sub = a[..., g.halo + ys:g.halo + ys + g.nj, g.halo + xs:g.halo + xs + g.ni]
        mean, mn, mx = substats(da, pos, dg)
        assert mean == want["sum"] / float(sub.size) and mn == sub.min() and mx == sub.max(), name
    dg.close()


@pytest.mark.gpu
def test_reproducing_sum_of_many_large_terms_and_error_codes():
    """a layer of 2^18 points whose limbs all sit near 2^45 (the device's split accumulators must carry what the reference
    carries row by row); NaN / unrepresentable terms give the reference's codes, or raise without `err`"""
    import torch
    from mom6_amd._lib import Mom6HipError
    from mom6_amd.coms import reproducing_sum
    from mom6_amd.tracer_advect import DeviceGrid
    g = synth.make_grid(1024, 256, 2, land_frac=0.0, seed=1)
    rng = np.random.default_rng(2)
    a = rng.standard_normal(g.shape3(H)) * 2.0 ** 45 * (1.0 + 2.0 ** -30 * rng.standard_normal(g.shape3(H)))
    a[1] = -np.abs(a[1])
    dg = DeviceGrid(g)
    da = torch.from_numpy(a).cuda()
    want = orc.reproducing_sum(g, a, H, by_layer=True)
    got = reproducing_sum(da, H, dg, by_layer=True)
    assert [e.v for e in got.EFP_lay_sums] == want["efp_lay"] and got.sums == want["sums"] and got.sum == want["sum"]
    assert reproducing_sum(da, H, dg).EFP_sum.v == orc.reproducing_sum(g, a, H)["efp"]
    for val, code in ((np.nan, 2), (1.0e300, 3), (-np.inf, 3)):
        b = a.copy(); b[1, g.halo + 5, g.halo + 7] = val
        db = torch.from_numpy(b).cuda()
        r = reproducing_sum(db, H, dg, return_err=True)
        o = orc.reproducing_sum(g, b, H, return_err=True)
        assert r.err == o["err"] == code and r.EFP_sum.v == [0] * 6 and r.sum == 0.0, val
        with pytest.raises(Mom6HipError):
            reproducing_sum(db, H, dg)
    halo_nan = a.copy(); halo_nan[:, 0, :] = np.nan                     # halo values are not summed
    assert reproducing_sum(torch.from_numpy(halo_nan).cuda(), H, dg).EFP_sum.v == orc.reproducing_sum(g, a, H)["efp"]
    dg.close()


@pytest.mark.gpu
def test_reproducing_sum_through_the_rccl_exchange():
    """the limbs cross sum_across_PEs as 16-bit pieces of 32-bit integers: the library's own all-reduce with the rank as its
    only peer (tests/test_native_domain.py) must hand back the same integers"""
    import torch
    from mom6_amd.coms import reproducing_sum
    from test_native_domain import native_grid
    import exact_synth as xs
    g = xs.make_grid(37, 23, 3, reentrant_x=True, reentrant_y=True)
    dom, dg = native_grid(g)
    rng = np.random.default_rng(4)
    a = rng.standard_normal(g.shape3(U)) * 10.0 ** rng.integers(-15, 15, g.shape3(U))
    a[0] = -np.abs(a[0]) * 1.0e30                                        # a negative first limb
    want = orc.reproducing_sum(g, a, U, by_layer=True)
    got = reproducing_sum(torch.from_numpy(a).cuda(), U, dg, by_layer=True)
    assert [e.v for e in got.EFP_lay_sums] == want["efp_lay"] and got.sum == want["sum"] and got.npoints == 37 * 23 * 3
    assert reproducing_sum(torch.from_numpy(a).cuda(), U, dg).EFP_sum.v == orc.reproducing_sum(g, a, U)["efp"]
    dg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [(1, 2), (2, 1)])
def test_reproducing_sum_layout_independence(tmp_path, layout):
    """two tiles (gloo ranks sharing the card): every rank gets the one-tile integers"""
    import torch.multiprocessing as mp
    from mp_workers import reproducing_sum_layout_worker
    from test_domains import free_port
    mp.spawn(reproducing_sum_layout_worker, args=(2, free_port(), layout, str(tmp_path)), nprocs=2, join=True)
    glob = np.load(tmp_path / "global.npz")
    for r in range(2):
        t = np.load(tmp_path / f"tile{r}.npz")
        for name in glob.files:
            assert np.array_equal(t[name].view(np.uint64), glob[name].view(np.uint64)), (layout, r, name)
