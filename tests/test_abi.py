"""The C-ABI library loads without a GPU, exports every symbol include/mom6hip.h declares, the
ctypes mirror matches the compiled structs, and the product fails loudly without a device."""
import ctypes as C
import os
import re

import pytest

from mom6_amd import _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "mom6hip.h")).read()
    declared = set(re.findall(r"\b(mom6hip_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations found"
    L = _lib.lib()
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    assert declared == set(_lib.EXPORTS)


def test_struct_layout_matches():
    L = _lib.lib()
    assert L.mom6hip_abi_sizeof_grid() == C.sizeof(_abi.GridStruct)
    assert L.mom6hip_abi_offsetof_grid_mask2dT() == _abi.GridStruct.mask2dT.offset
    assert L.mom6hip_abi_sizeof_tracer_advect_cs() == C.sizeof(_abi.TracerAdvectCS)
    assert L.mom6hip_abi_sizeof_advect_stats() == C.sizeof(_abi.AdvectStats)
    assert L.mom6hip_abi_sizeof_advect_timing() == C.sizeof(_abi.AdvectTiming)
    for n, t in (("remapping_cs", _abi.RemappingCS), ("regridding_cs", _abi.RegriddingCS), ("coriolisadv_cs", _abi.CoriolisAdvCS),
                 ("continuity_cs", _abi.ContinuityCS), ("bt_cont", _abi.BTCont), ("eos", _abi.EOS),
                 ("pressureforce_cs", _abi.PressureForceCS), ("barotropic_cs", _abi.BarotropicCS),
                 ("dyn_split_rk2_cs", _abi.DynSplitRK2CS), ("vertvisc_cs", _abi.VertviscCS), ("vertvisc_type", _abi.VertviscType),
                 ("hor_visc_cs", _abi.HorViscCS), ("set_visc_cs", _abi.SetViscCS),
                 ("tracer_hor_diff_cs", _abi.TracerHorDiffCS), ("hordiff_stats", _abi.HorDiffStats), ("epipycnal_cs", _abi.EpipycnalCS), ("energy_sums", _abi.EnergySums)):
        f = getattr(L, f"mom6hip_abi_sizeof_{n}"); f.restype = C.c_uint64
        assert f() == C.sizeof(t), n


def test_no_cpu_fallback():
    """Without a GPU the product must raise, not compute."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mom6_amd import synth
    from mom6_amd.tracer_advect import DeviceGrid
    g = synth.make_grid(8, 8, 1)
    with pytest.raises(_lib.Mom6HipError, match="no HIP device"):
        DeviceGrid(g)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mom6_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".F90")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower().replace("the oracle", "").replace("oracle/domains.c", "") \
                    or f in ("tracer_advect.hip", "Makefile"), f"{f} mentions the oracle"
