"""ctypes front-end of the CPU oracle (oracle/libmom6oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by mom6_amd."""
import ctypes as C
import os
import subprocess

import numpy as np

from mom6_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_dp = C.POINTER(C.c_double)


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmom6oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_halo_update.argtypes = [C.POINTER(_abi.GridStruct), _dp, C.c_int, C.c_int]
        L.orc_halo_update.restype = None
        L.orc_advect_tracer.argtypes = [
            C.POINTER(_abi.GridStruct), _dp, _dp, _dp, C.c_double, C.POINTER(_abi.TracerAdvectCS),
            C.POINTER(_dp), _dp, C.c_int, C.c_int, _dp, C.c_int, C.c_int, _dp, _dp,
            C.POINTER(_abi.AdvectStats)]
        L.orc_advect_tracer.restype = C.c_int
        _LIB = L
    return _LIB


def _p(a):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(_dp)


def halo_update(grid, f, pos):
    nk = 1 if f.ndim == 2 else f.shape[0]
    lib().orc_halo_update(C.byref(grid.struct()), _p(f), pos, nk)


def advect_tracer(grid, h_end, uhtr, vhtr, dt, cs_dt, scheme, tr, conc_underflow=None,
                  x_first=None, vol_prev=None, max_iter=None, update_vol_prev=False,
                  uhr_out=None, vhr_out=None, use_huynh_stencil_bug=False):
    """advect_tracer on numpy arrays (tracers updated in place).  Returns AdvectStats."""
    cs = _abi.TracerAdvectCS(float(cs_dt), _abi.ADV_SCHEMES[scheme], int(use_huynh_stencil_bug))
    ntr = len(tr)
    trp = (_dp * ntr)(*[_p(t) for t in tr])
    cu = None if conc_underflow is None else np.ascontiguousarray(conc_underflow, dtype=np.float64)
    st = _abi.AdvectStats()
    rc = lib().orc_advect_tracer(
        C.byref(grid.struct()), _p(h_end), _p(uhtr), _p(vhtr), float(dt), C.byref(cs), trp, _p(cu),
        ntr, -1 if x_first is None else int(bool(x_first)), _p(vol_prev),
        0 if max_iter is None else int(max_iter), int(bool(update_vol_prev)), _p(uhr_out),
        _p(vhr_out), C.byref(st))
    if rc != 0:
        raise RuntimeError(f"orc_advect_tracer failed rc={rc}")
    return st
