"""mom6_amd -- MI355X (gfx950) implementation of the MOM6 split-RK2 dynamical-core hot path.

The product is libmom6hip.so (hand-written HIP kernels behind the C ABI of include/mom6hip.h);
this package is the thin host-side mirror of the reference's Fortran procedures for that path.
"""
from . import _abi
from .grid import Grid

__all__ = ["Grid", "_abi"]
