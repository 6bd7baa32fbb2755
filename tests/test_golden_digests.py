"""Bitwise parity at full size through golden digests.

tests/golden/digests_<size>.json hold the SHA-256 of the bits of every output of the CPU oracle for the scenarios of
tests/digest_scenarios.py (made by tools/make_golden_digests.py in the build container).  On the GPU the same
scenarios run through libmom6hip and every output must hash to the same value: this is the tolerance-0 comparison at
360x180x75 (BASELINE configs[2]), on a 1440-wide band of the OM4_025 grid and on a 1080-row strip, where the launch
geometry has many blocks per row, several J segments and the barotropic hipGraph at its real step count.

On the CPU: the input generator is machine-independent (its digests are part of the fixture), and the oracle
reproduces a sample of its own digests (the fixture is not stale)."""
import json
import os

import numpy as np
import pytest

import digest_scenarios as ds

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(size):
    with open(os.path.join(GOLD, f"digests_{size}.json")) as f:
        return json.load(f)["fields"]


@pytest.mark.parametrize("size", list(ds.SIZES))
def test_inputs_are_reproduced_bit_for_bit(size):
    gold = load(size)
    g, d, dm, taux, tauy, bbl = ds.make_inputs(size)
    for name, a in (("input.h", d["h"]), ("input.u", d["u"]), ("input.T", d["T"]), ("input.bathyT", g.bathyT), ("input.model_h", dm["h"])):
        assert ds.digest(a)["sha256"] == gold[name]["sha256"], name


def test_oracle_reproduces_its_digests():
    """the cheapest operators of the tall strip, on the CPU: the fixture belongs to this oracle"""
    gold = load("tall")
    got = ds.run(ds.OracleOps, "tall", parts=("operators",), only=("coradcalc", "pressureforce", "vertvisc", "hor_visc", "set_viscous_BBL", "hordiff"))
    assert len(got) > 10
    for name, dg in got.items():
        assert dg["sha256"] == gold[name]["sha256"], name


@pytest.mark.gpu
@pytest.mark.parametrize("size", list(ds.SIZES))
def test_hip_matches_golden_digests(size):
    gold = load(size)
    got = ds.run(ds.HipOps, size)
    bad = []
    for name, want in gold.items():
        have = got.get(name)
        if have is None:
            bad.append((name, "missing"))
        elif have["sha256"] != want["sha256"]:
            bad.append((name, dict(nonzero=(have["nonzero"], want["nonzero"]), first_nonzero=(have["first_nonzero"], want["first_nonzero"]),
                                   first8=(have["first8"][:3], want["first8"][:3]))))
    assert not bad, f"{len(bad)} of {len(gold)} fields differ from the oracle's digests; first: {bad[:4]}"
    assert len(got) == len(gold)
