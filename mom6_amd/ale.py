"""Host-side mirror of the ALE remapping entry points (reference: src/ALE/MOM_ALE.F90,
src/ALE/MOM_remapping.F90); the work is done by libmom6hip (mom6_amd/csrc/ale_remap.hip)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi
from ._lib import Mom6HipError, check, lib
from .tracer_advect import DeviceGrid, _ptr_space

_dp = C.POINTER(C.c_double)


class RemappingCS:
    """remapping_CS (MOM_remapping.F90:25-41) as set by initialize_remapping (:1259)."""

    def __init__(self, remapping_scheme="PLM", boundary_extrapolation=False, force_bounds_in_subcell=False,
                 answer_date=99991231):
        if remapping_scheme not in _abi.REMAP_SCHEMES:
            # setReconstructionType, MOM_remapping.F90:1282-1326
            raise Mom6HipError("setReconstructionType: Unrecognized choice for REMAPPING_SCHEME ("
                               + str(remapping_scheme) + "); libmom6hip provides PCM, PLM, PPM_H4")
        self.remapping_scheme = remapping_scheme
        self.boundary_extrapolation = bool(boundary_extrapolation)
        self.force_bounds_in_subcell = bool(force_bounds_in_subcell)
        self.answer_date = int(answer_date)

    def struct(self):
        return _abi.RemappingCS(_abi.REMAP_SCHEMES[self.remapping_scheme], int(self.boundary_extrapolation),
                                int(self.force_bounds_in_subcell), self.answer_date)


def initialize_remapping(remapping_scheme="PLM", boundary_extrapolation=False, force_bounds_in_subcell=False,
                         answer_date=99991231):
    return RemappingCS(remapping_scheme, boundary_extrapolation, force_bounds_in_subcell, answer_date)


def ALE_remap_tracers(CS: RemappingCS, G: DeviceGrid, h_old, h_new, Reg, conc_underflow=None):
    """ALE_remap_tracers(CS, G, GV, h_old, h_new, Reg, ...) -- MOM_ALE.F90:737.  `Reg` is the list of tracer
    arrays, remapped in place from the grid h_old to the grid h_new."""
    ntr = 0 if Reg is None else len(Reg)
    if ntr == 0:
        return
    shape = G.grid.shape3(_abi.POS_H)
    spaces = set()

    def P(a, name):
        if tuple(a.shape) != shape:
            raise Mom6HipError(f"ALE_remap_tracers: {name} has shape {tuple(a.shape)}, expected {shape}")
        p, s = _ptr_space(a)
        spaces.add(s)
        return C.c_void_p(p)

    ph0, ph1 = P(h_old, "h_old"), P(h_new, "h_new")
    trp = (C.c_void_p * ntr)(*[P(t, f"Reg%Tr({m+1})%t") for m, t in enumerate(Reg)])
    if len(spaces) != 1:
        raise Mom6HipError("ALE_remap_tracers: all fields must be in the same memory space")
    cu = None
    if conc_underflow is not None:
        cu = np.ascontiguousarray(conc_underflow, dtype=np.float64)
    cs = CS.struct()
    check(lib().mom6hip_ale_remap_tracers(G.handle, C.byref(cs), ph0, ph1, trp,
                                          None if cu is None else cu.ctypes.data_as(_dp), ntr, spaces.pop()),
          "ALE_remap_tracers")
