"""Host-side horizontal grid: the members of ocean_grid_type / hor_index_type / verticalGrid_type
that the hot path reads (reference: src/core/MOM_grid.F90, src/framework/MOM_hor_index.F90,
src/core/MOM_verticalGrid.F90), laid out as the reference lays them out (symmetric memory).

numpy arrays are C-ordered with the LAST axis = i, so `a[j - jsd, i - isd]` is Fortran `a(i,j)`;
3-D fields are `(nk, nj, ni)`.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _abi


@dataclass
class Grid:
    ni: int                      # compute-domain size in i
    nj: int
    nk: int
    halo: int = 4                # NIHALO = NJHALO (reference .testing and OM4-class runs use 4)
    reentrant_x: bool = True     # REENTRANT_X default (src/framework/MOM_domains.F90:184)
    reentrant_y: bool = False
    tripolar_n: bool = False     # TRIPOLAR_N (src/framework/MOM_domains.F90:189)
    first_direction: int = 0
    Angstrom_H: float = 1.0e-10  # GV%Angstrom_H default ANGSTROM=1e-10 m (MOM_verticalGrid.F90)
    H_subroundoff: float = 1.0e-20 * max(1.0e-10, 1.0e-17)  # overwritten in __post_init__
    dZ_subroundoff: float = 1.0e-20 * 1.0e-10
    H_to_Z: float = 1.0
    Z_to_H: float = 1.0
    g_Earth: float = 9.80
    Rho0: float = 1035.0
    metrics: dict = field(default_factory=dict)

    def __post_init__(self):
        # GV%H_subroundoff = 1e-20 * max(GV%Angstrom_H, GV%m_to_H*1e-17)   (MOM_verticalGrid.F90:165)
        self.H_subroundoff = 1.0e-20 * max(self.Angstrom_H, 1.0e-17)
        self.dZ_subroundoff = 1.0e-20 * max(self.Angstrom_H * self.H_to_Z, 1.0e-17)
        # hor_index_type: local numbering with isd = 1 (MOM_hor_index.F90)
        self.isd, self.jsd = 1, 1
        self.isc, self.jsc = 1 + self.halo, 1 + self.halo
        self.iec, self.jec = self.isc + self.ni - 1, self.jsc + self.nj - 1
        self.ied, self.jed = self.iec + self.halo, self.jec + self.halo
        self._struct = None

    # ---- shapes -----------------------------------------------------------------------------
    @property
    def nih(self): return self.ied - self.isd + 1
    @property
    def njh(self): return self.jed - self.jsd + 1

    def shape2(self, pos):
        xs = 1 if pos in (_abi.POS_U, _abi.POS_Q) else 0
        ys = 1 if pos in (_abi.POS_V, _abi.POS_Q) else 0
        return (self.njh + ys, self.nih + xs)

    def shape3(self, pos, nk=None):
        return ((self.nk if nk is None else nk),) + self.shape2(pos)

    def zeros2(self, pos): return np.zeros(self.shape2(pos), dtype=np.float64)
    def zeros3(self, pos): return np.zeros(self.shape3(pos), dtype=np.float64)

    # slices of the compute domain inside a data-domain array of staggering `pos`
    def csl(self, pos):
        xs = 1 if pos in (_abi.POS_U, _abi.POS_Q) else 0
        ys = 1 if pos in (_abi.POS_V, _abi.POS_Q) else 0
        h = self.halo
        # u: I = isc-1..iec  -> offsets (isc-1)-(isd-1) = h .. ; h-points: isc-isd = h
        return (slice(h, h + self.nj + ys), slice(h, h + self.ni + xs))

    def pos_of(self, name):
        if name in _abi.H_METRICS: return _abi.POS_H
        if name in _abi.U_METRICS: return _abi.POS_U
        if name in _abi.V_METRICS: return _abi.POS_V
        if name in _abi.Q_METRICS: return _abi.POS_Q
        raise KeyError(name)

    def set_metric(self, name, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        if arr.shape != self.shape2(self.pos_of(name)):
            raise ValueError(f"{name}: shape {arr.shape} != {self.shape2(self.pos_of(name))}")
        self.metrics[name] = arr
        self._struct = None

    def __getattr__(self, name):
        m = self.__dict__.get("metrics")
        if m is not None and name in m:
            return m[name]
        raise AttributeError(name)

    # ---- C struct ---------------------------------------------------------------------------
    def struct(self) -> _abi.GridStruct:
        if self._struct is None:
            s = _abi.GridStruct()
            for n in ("isc", "iec", "jsc", "jec", "isd", "ied", "jsd", "jed", "nk"):
                setattr(s, n, int(getattr(self, n)))
            s.symmetric = 1
            s.reentrant_x = int(self.reentrant_x)
            s.reentrant_y = int(self.reentrant_y)
            s.tripolar_n = int(self.tripolar_n)
            s.first_direction = int(self.first_direction)
            for n in ("Angstrom_H", "H_subroundoff", "dZ_subroundoff", "H_to_Z", "Z_to_H",
                      "g_Earth", "Rho0"):
                setattr(s, n, float(getattr(self, n)))
            for n in _abi.ALL_METRICS:
                a = self.metrics.get(n)
                if a is not None:
                    setattr(s, n, a.ctypes.data_as(C.POINTER(C.c_double)))
            self._struct = s
        return self._struct
