"""The Fortran side of the boundary: mom6_amd/fortran/mom6hip_c_api.F90 (ISO_C_BINDING interfaces)
compiles with amdflang, and a Fortran host program that owns plain Fortran arrays drives the HIP
advect_tracer through it (HOST memspace: the drop-in path) with bit-identical results to the oracle."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from mom6_amd import _abi
from mom6_amd.grid import Grid
from helpers import bits_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FC = shutil.which("amdflang") or "/opt/rocm/bin/amdflang"
API = os.path.join(ROOT, "mom6_amd", "fortran", "mom6hip_c_api.F90")
DRV = os.path.join(ROOT, "tests", "fortran", "advect_driver.F90")


def _build(tmp):
    flags = ["-O0", "-ffp-contract=off"]
    subprocess.run([FC, *flags, "-c", API, "-o", str(tmp / "api.o"), "-J", str(tmp)], check=True)
    subprocess.run([FC, *flags, f"-I{tmp}", "-c", DRV, "-o", str(tmp / "drv.o"), "-J", str(tmp)], check=True)
    libdir = os.path.join(ROOT, "mom6_amd")
    subprocess.run([FC, str(tmp / "drv.o"), str(tmp / "api.o"), f"-L{libdir}", "-lmom6hip",
                    f"-Wl,-rpath,{libdir}", "-o", str(tmp / "advect_driver")], check=True)
    return str(tmp / "advect_driver")


@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")
def test_fortran_binding_compiles_and_fails_loudly_without_gpu(tmp_path):
    exe = _build(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([exe, str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode != 0
    assert "no HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")
def test_fortran_driver_matches_oracle(tmp_path, oracle):
    exe = _build(tmp_path)
    out = tmp_path / "out.bin"
    r = subprocess.run([exe, str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "advect_driver ok" in r.stdout
    ni, nj, nk, halo = 20, 12, 3, 4
    g = Grid(ni=ni, nj=nj, nk=nk, halo=halo, reentrant_x=True, reentrant_y=True)
    g.H_subroundoff = 1.0e-30
    nih, njh = g.nih, g.njh
    raw = np.fromfile(str(out) + ".in", dtype="<f8")
    sizes = [njh * nih, nk * njh * nih, nk * njh * (nih + 1), nk * (njh + 1) * nih, nk * njh * nih, nk * njh * nih]
    assert raw.size == sum(sizes)
    parts = np.split(raw, np.cumsum(sizes)[:-1])
    areaT = parts[0].reshape(njh, nih)
    h_end = parts[1].reshape(nk, njh, nih); uhtr = parts[2].reshape(nk, njh, nih + 1)
    vhtr = parts[3].reshape(nk, njh + 1, nih)
    t1 = parts[4].reshape(nk, njh, nih).copy(); t2 = parts[5].reshape(nk, njh, nih).copy()
    g.set_metric("areaT", areaT); g.set_metric("IareaT", 1.0 / areaT)
    g.set_metric("mask2dT", np.ones((njh, nih)))
    g.set_metric("mask2dCu", np.ones((njh, nih + 1))); g.set_metric("mask2dCv", np.ones((njh + 1, nih)))
    st = oracle.advect_tracer(g, np.ascontiguousarray(h_end), np.ascontiguousarray(uhtr),
                              np.ascontiguousarray(vhtr), 3600.0, 900.0, "PPM:H3", [t1, t2])
    res = np.fromfile(str(out), dtype="<f8")
    r1, r2 = np.split(res, 2)
    assert bits_equal(t1.ravel(), r1) and bits_equal(t2.ravel(), r2)
    assert f"iterations={st.iterations}" in r.stdout
