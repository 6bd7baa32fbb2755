"""Host-side mirror of MOM_coms' order-invariant sums (reference: src/framework/MOM_coms.F90): reproducing_sum of a field
that lives on the GPU through mom6hip_reproducing_sum -- the extended-fixed-point (EFP) sum whose bits do not depend on the
domain decomposition (Hallberg & Adcroft 2014) -- without a device-to-host copy of the field."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional

from ._lib import Mom6HipError, check, lib
from .tracer_advect import DeviceGrid, _ptr_space

NI = 6                   # MOM_coms.F90:36
PREC = 1 << 46           # :28


@dataclass
class EFP_type:
    """type EFP_type (:75): the six integers of an extended-fixed-point number."""
    v: List[int]

    def to_real(self) -> float:
        return EFP_to_real(self)


def EFP_to_real(e: EFP_type) -> float:
    """EFP_to_real (:790) = ints_to_real (:545): sum of pr(i)*ints(i), i ascending, in doubles."""
    r = 0.0
    for i, x in enumerate(e.v):
        r = r + float(2.0 ** (46 * (2 - i))) * float(x)
    return r


@dataclass
class ReproducingSum:
    sum: float
    EFP_sum: EFP_type
    npoints: int
    err: int
    sums: Optional[List[float]] = None
    EFP_lay_sums: Optional[List[EFP_type]] = None


def _setup():
    L = lib()
    if not getattr(L, "_coms_ready", False):
        L.mom6hip_reproducing_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                              C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_int32]
        L._coms_ready = True
    return L


def reproducing_sum(array, pos, G: DeviceGrid, by_layer=False, return_err=False) -> ReproducingSum:
    """reproducing_sum(array(isc:iec,jsc:jec[,:]), sums, EFP_sum, EFP_lay_sums, err) -- reproducing_sum_2d (:219) /
    reproducing_sum_3d (:318) over the h-point computational domain of a field of staggering `pos`.  by_layer asks for the
    by-layer sums too (and, as in the reference, makes the total the floating-point sum of those).  Without return_err an
    unrepresentable term, an overflow or a NaN raises, as the reference's FATAL does."""
    if G is None:
        raise Mom6HipError("MOM_coms: reproducing_sum needs the grid the field lives on.")
    p, space = _ptr_space(array)
    nk = 1 if array.ndim == 2 else int(array.shape[0])
    s, n, e = C.c_double(0.0), C.c_int64(0), C.c_int32(0)
    tot = (C.c_int64 * NI)()
    lay = (C.c_double * nk)() if by_layer else None
    elay = (C.c_int64 * (NI * nk))() if by_layer else None
    check(_setup().mom6hip_reproducing_sum(G.handle, C.c_void_p(p), int(pos), nk, C.byref(s), lay, tot, elay, C.byref(n),
                                           C.byref(e) if return_err else None, space), "reproducing_sum")
    out = ReproducingSum(float(s.value), EFP_type([int(x) for x in tot]), int(n.value), int(e.value))
    if by_layer:
        out.sums = [float(x) for x in lay]
        out.EFP_lay_sums = [EFP_type([int(elay[NI * k + i]) for i in range(NI)]) for k in range(nk)]
    return out
