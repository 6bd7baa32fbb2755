"""Restart / diagnostic staging (mom6hip_stage_to_host / _wait, include/mom6hip.h): a staged field is a snapshot -- what the
host array receives is the field as it was when it was staged, whatever the model has done to it since."""
import numpy as np
import pytest

from mom6_amd import _abi, synth


@pytest.mark.gpu
@pytest.mark.parametrize("pinned", [True, False])
def test_staged_fields_are_snapshots(pinned):
    import torch
    from mom6_amd.staging import host_register, host_unregister, stage_query, stage_to_host, stage_wait
    from mom6_amd.tracer_advect import DeviceGrid
    g = synth.make_grid(300, 200, 20, land_frac=0.2, seed=3)
    dg = DeviceGrid(g)
    d = synth.make_dynamics_state(g, seed=4, umax=0.3, eta_amp=0.2, device="cuda")
    names = ("u", "v", "h", "T", "S")
    host = {n: np.full(tuple(d[n].shape), np.nan) for n in names}
    if pinned:
        for n in names:
            host_register(host[n])
    for batch in range(3):      # the slots of a batch are reused by the next
        want = {n: d[n].cpu().numpy().copy() for n in names}
        for n in names:
            stage_to_host(dg, host[n], d[n])
        for n in names:         # the model moves on at once
            d[n].mul_(1.5).add_(float(batch))
        assert 0 <= stage_query(dg) <= len(names)
        stage_wait(dg)
        assert stage_query(dg) == 0
        for n in names:
            assert np.array_equal(host[n].view(np.uint64), want[n].view(np.uint64)), (batch, n)
    if pinned:
        for n in names:
            host_unregister(host[n])
    dg.close()


@pytest.mark.gpu
def test_staging_refuses_bad_arguments():
    import torch
    from mom6_amd._lib import Mom6HipError
    from mom6_amd.staging import stage_to_host, stage_wait
    from mom6_amd.tracer_advect import DeviceGrid
    g = synth.make_grid(20, 12, 3, seed=1)
    dg = DeviceGrid(g)
    a = torch.zeros(g.shape3(_abi.POS_H), dtype=torch.float64, device="cuda")
    with pytest.raises(Mom6HipError):
        stage_to_host(dg, np.zeros(5), a)                         # wrong size
    with pytest.raises(Mom6HipError):
        stage_to_host(dg, np.zeros(g.shape3(_abi.POS_H)), np.zeros(g.shape3(_abi.POS_H)))      # not a device field
    stage_wait(dg)                                                # nothing staged: returns
    dg.close()


@pytest.mark.gpu
def test_stream_bandwidth_reports_plausible_numbers():
    """mom6hip_stream_bandwidth: the measured companion of the nominal HBM peak (bench.py `roofline.measured_streaming_bandwidth`)"""
    import ctypes as C
    from mom6_amd._lib import check, lib
    from mom6_amd.tracer_advect import DeviceGrid
    g = synth.make_grid(20, 12, 3, seed=1)
    dg = DeviceGrid(g)
    L = lib()
    L.mom6hip_stream_bandwidth.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    cp, tr = C.c_double(0.0), C.c_double(0.0)
    check(L.mom6hip_stream_bandwidth(dg.handle, 256 << 20, 5, C.byref(cp), C.byref(tr)), "mom6hip_stream_bandwidth")
    assert 500.0 < cp.value < 9000.0 and 500.0 < tr.value < 9000.0      # GB/s: above PCIe by far, below the nominal 8 TB/s + slack
    dg.close()
