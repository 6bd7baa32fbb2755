"""The all-cores build of the oracle (libmom6oracle_omp.so: the same sources with -fopenmp, used only by bench.py's
CPU baseline) gives the bits of the scalar oracle: every operator and the time-stepping model of the golden-digest
scenarios on the tall strip and on a small benchmark-shaped grid, with 2, 3 and all threads."""
import os

import pytest

import digest_scenarios as ds
from oracle import orc


@pytest.fixture
def scalar_again():
    yield
    orc.set_threads(1)


@pytest.mark.parametrize("size", [(40, 24, 6), (23, 31, 9)])
def test_openmp_oracle_matches_scalar_oracle_bitwise(size, scalar_again):
    orc.set_threads(1)
    ref = ds.run(ds.OracleOps, size)
    for n in (2, 3, 0):
        used = orc.set_threads(n)
        if n and used != n:
            continue
        got = ds.run(ds.OracleOps, size)
        bad = [k for k in ref if got[k]["sha256"] != ref[k]["sha256"]]
        assert not bad, (n, bad[:5])
