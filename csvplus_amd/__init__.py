"""csvplus_amd — MI355X (gfx950) implementation of csvplus's Index-build + Join hot path.

The product is the C-ABI library `lib/libcsvplus_hip.so` (hand-written HIP kernels,
include/csvplus_hip.h).  This package is its Python host side: the ctypes binding,
the SoA column staging and the multi-GPU sharding.  There is no CPU fallback.

Modules: `engine` (one GPU: index_on / join / chained_join on torch device memory), `dist` (probe-row sharding +
allgatherv over RCCL), `streaming` (host -> device pipeline of join chunks), `ingest` (CSV text -> columns),
`materialize` (gather through row ids, ToCsv), `dedup` (ResolveDuplicates over the device index),
`pipeline` (CSV -> indices -> chained join -> CSV, all in HBM), `datagen` (deterministic synthetic tables).
"""
from . import _native as native  # noqa: F401
from ._native import CphError, NativeLibraryMissing, Context, DeviceIndex, Matches, Chain, join_chain  # noqa: F401
from .columns import StrCol  # noqa: F401

__all__ = ["native", "CphError", "NativeLibraryMissing", "Context", "DeviceIndex", "Matches", "Chain", "join_chain", "StrCol"]
