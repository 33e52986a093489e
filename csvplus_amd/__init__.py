"""csvplus_amd — MI355X (gfx950) implementation of csvplus's Index-build + Join hot path.

The product is the C-ABI library `lib/libcsvplus_hip.so` (hand-written HIP kernels,
include/csvplus_hip.h).  This package is its Python host side: the ctypes binding,
the SoA column staging and the multi-GPU sharding.  There is no CPU fallback.
"""
from . import _native as native  # noqa: F401
from ._native import CphError, NativeLibraryMissing, Context, DeviceIndex, Matches, Chain, join_chain  # noqa: F401
from .columns import StrCol  # noqa: F401

__all__ = ["native", "CphError", "NativeLibraryMissing", "Context", "DeviceIndex", "Matches", "Chain", "join_chain", "StrCol"]
