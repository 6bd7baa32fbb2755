"""Restart / diagnostic staging: device-resident fields into the host arrays MOM6 registered for them
(register_restart_field, src/framework/MOM_restart.F90; the arrays post_data reads, src/framework/MOM_diag_mediator.F90)
while the model keeps stepping -- mom6hip_stage_to_host / mom6hip_stage_wait."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import Mom6HipError, check, lib
from .tracer_advect import DeviceGrid, _ptr_space
from . import _abi


def _setup():
    L = lib()
    if not getattr(L, "_stage_ready", False):
        L.mom6hip_host_register.argtypes = [C.c_void_p, C.c_uint64]
        L.mom6hip_host_unregister.argtypes = [C.c_void_p]
        L.mom6hip_stage_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.mom6hip_stage_query.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.mom6hip_stage_wait.argtypes = [C.c_void_p]
        L._stage_ready = True
    return L


def host_register(a: np.ndarray):
    """page-lock a host array once, so that its staged copies run asynchronously at the PCIe rate"""
    if not (isinstance(a, np.ndarray) and a.flags.c_contiguous):
        raise Mom6HipError("host_register: a C-contiguous numpy array is needed")
    check(_setup().mom6hip_host_register(C.c_void_p(a.ctypes.data), a.nbytes), "host_register")


def host_unregister(a: np.ndarray):
    check(_setup().mom6hip_host_unregister(C.c_void_p(a.ctypes.data)), "host_unregister")


def stage_to_host(G: DeviceGrid, host: np.ndarray, field):
    """snapshot `field` (a device tensor) now and start its copy into `host`; returns at once"""
    p, space = _ptr_space(field)
    if space != _abi.MEM_DEVICE:
        raise Mom6HipError("stage_to_host: the field must live on the device")
    nbytes = field.numel() * field.element_size()
    if not (isinstance(host, np.ndarray) and host.flags.c_contiguous and host.nbytes == nbytes):
        raise Mom6HipError("stage_to_host: the host array must be C-contiguous and of the field's size")
    check(_setup().mom6hip_stage_to_host(G.handle, C.c_void_p(host.ctypes.data), C.c_void_p(p), nbytes), "stage_to_host")


def stage_query(G: DeviceGrid) -> int:
    n = C.c_int32(0)
    check(_setup().mom6hip_stage_query(G.handle, C.byref(n)), "stage_query")
    return int(n.value)


def stage_wait(G: DeviceGrid):
    check(_setup().mom6hip_stage_wait(G.handle), "stage_wait")
