"""Host-side mirror of MOM_checksums (reference: src/framework/MOM_checksums.F90): hchksum / uchksum / vchksum / Bchksum of
fields that live on the GPU, through mom6hip_chksum -- the bit-count numbers MOM6 prints in debugging runs, without a
device-to-host copy of the field."""
from __future__ import annotations

import ctypes as C

from . import _abi
from ._lib import check, lib
from .tracer_advect import DeviceGrid, _ptr_space


def _setup():
    L = lib()
    if not getattr(L, "_chk_ready", False):
        L.mom6hip_chksum.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int32] * 5 + [C.c_double, C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                                                                 C.POINTER(C.c_double), C.c_int32]
        L._chk_ready = True
    return L


def chksum(array, pos, G: DeviceGrid, di=0, dj=0, symmetric=False, scale=1.0):
    """(bitcount, min, max) of `array` (staggering `pos`) over the compute domain shifted by (di, dj): subchk / subStats."""
    p, space = _ptr_space(array)
    nk = 1 if array.ndim == 2 else int(array.shape[0])
    bc, mn, mx = C.c_int64(0), C.c_double(0.0), C.c_double(0.0)
    check(_setup().mom6hip_chksum(G.handle, C.c_void_p(p), int(pos), nk, int(di), int(dj), int(bool(symmetric)), float(scale), C.byref(bc),
                                  C.byref(mn), C.byref(mx), space), "chksum")
    return int(bc.value), float(mn.value), float(mx.value)


def substats(array, pos, G: DeviceGrid, symmetric=False):
    """(aMean, aMin, aMax) of subStats (:1403 for h points, :1061 / :1247 / :1582 for the staggered ones): the extrema over the
    computational domain (the symmetric one with sym_stats) and the mean over the h-point computational domain, from the
    order-invariant sum of MOM_coms -- the three numbers chk_sum_msg prints beside the bit counts."""
    from .coms import reproducing_sum
    _, mn, mx = chksum(array, pos, G, 0, 0, symmetric)
    rs = reproducing_sum(array, pos, G)
    return rs.sum / float(rs.npoints), mn, mx


def _shifted(array, pos, G, haloshift, symmetric, omit_corners, scale):
    out = {"bc0": chksum(array, pos, G, 0, 0, symmetric, scale)[0]}
    h = int(haloshift)
    if h == 0:
        return out
    if not omit_corners:      # chksum_h_3d :1368-1376
        for n, (di, dj) in (("bcSW", (-h, -h)), ("bcSE", (h, -h)), ("bcNW", (-h, h)), ("bcNE", (h, h))):
            out[n] = chksum(array, pos, G, di, dj, symmetric, scale)[0]
    else:                     # :1377-1383
        for n, (di, dj) in (("bcS", (0, -h)), ("bcE", (h, 0)), ("bcW", (-h, 0)), ("bcN", (0, h))):
            out[n] = chksum(array, pos, G, di, dj, symmetric, scale)[0]
    return out


def hchksum(array, mesg, G: DeviceGrid, haloshift=0, omit_corners=False, scale=1.0):
    """hchksum(array, mesg, HI, haloshift, omit_corners, scale) -- chksum_h_2d / chksum_h_3d (:1277): the checksums the
    reference would print for `mesg`, as a dict (bc0 and, with a halo shift, the four shifted ones)."""
    return _shifted(array, _abi.POS_H, G, haloshift, False, omit_corners, scale)


def uchksum(array, mesg, G: DeviceGrid, haloshift=0, symmetric=False, omit_corners=False, scale=1.0):
    """chksum_u_3d (:905)"""
    return _shifted(array, _abi.POS_U, G, haloshift, symmetric, omit_corners, scale)


def vchksum(array, mesg, G: DeviceGrid, haloshift=0, symmetric=False, omit_corners=False, scale=1.0):
    """chksum_v_3d (:1091)"""
    return _shifted(array, _abi.POS_V, G, haloshift, symmetric, omit_corners, scale)


def Bchksum(array, mesg, G: DeviceGrid, haloshift=0, symmetric=False, omit_corners=False, scale=1.0):
    """chksum_B_3d (:1433)"""
    return _shifted(array, _abi.POS_Q, G, haloshift, symmetric, omit_corners, scale)
